"""GPU probe: the IPA generator fold at full size (2^19 pairs = the first round of a 2^20 opening), with the
closed form G'_i = [a + b] G0 + [a i + b (m + i)] D on sampled indices for generators G0 + j D."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plonky_amd import device as dev, synth, lib
from plonky_amd.selfcheck import _mul, _add
from plonky_amd.synth import MODULI
dev.init(0)
L = lib.load()
p, r = MODULI[0], MODULI[1]
G = (p - 1, 2)
D = _mul(p, 0xC0FFEE, G)
g0 = np.stack([synth.mont(0, G[0]), synth.mont(0, G[1])]); dd = np.stack([synth.mont(0, D[0]), synth.mont(0, D[1])])
for log_m in (12, 16, 19):
    m = 1 << log_m
    bases = dev.gen_bases_dev(0, 2 * m, g0, dd)
    lo, hi = bases[:m].contiguous(), bases[m:].contiguous()
    u = synth.to_int(synth.rand_field(1, 77, 1)[0])  # Montgomery limbs of some scalar
    a_can = synth.from_mont(1, synth.rand_field(1, 77, 1)[0]); b_can = pow(a_can, -1, r)
    sa = np.array(synth.mont(1, a_can), dtype=np.uint64); sb = np.array(synth.mont(1, b_can), dtype=np.uint64)
    out = torch.empty((m, 2, 4), dtype=torch.int64, device="cuda"); oz = torch.empty((m,), dtype=torch.uint8, device="cuda")
    def run():
        lib.check(L.plk_curve_fold_pairs_dev(0, m, ctypes.c_void_p(lo.data_ptr()), None, ctypes.c_void_p(hi.data_ptr()), None,
                                             sa.ctypes.data_as(ctypes.c_void_p), sb.ctypes.data_as(ctypes.c_void_p),
                                             ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(oz.data_ptr()), None))
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): run()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 3
    h = dev.to_host(out)
    ok = int(oz.sum().item()) == 0
    for i in (0, 1, m // 3, m - 1):
        exp = _add(p, _mul(p, (a_can + b_can) % r, G), _mul(p, (a_can * i + b_can * (m + i)) % r, D))
        ok = ok and (synth.from_mont(0, h[i, 0]), synth.from_mont(0, h[i, 1])) == exp
    print("fold 2^%d pairs: %.3f ms  (%.1f M pairs/s)  closed-form ok: %s" % (log_m, t * 1e3, m / t / 1e6, ok), flush=True)
