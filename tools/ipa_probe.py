"""GPU probe: all log2(n) rounds of the inner-product argument of one opening (halo.rs:63-124) through the C ABI (plk_halo_*:
device-resident vectors, two persistent table-free contexts for the long rounds, frozen generators + tabled MSMs for the
short ones) -- timed as a whole and round by round (correctness: tests/test_gpu_halo.py).
Usage: python tools/ipa_probe.py [log_n] [freeze_log ...] [tabled]
tabled: the plain argument first, then the one that starts over the caller's commitment tables (plk_halo_begin_tabled_dev; PLK_HALO_LEAD
lead rounds, default 3) - same results required."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plonky_amd import device as dev, synth
from plonky_amd.selfcheck import GENERATORS, _mul
from plonky_amd.synth import MODULI
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
TABLED = "tabled" in sys.argv
freezes = [int(v) for v in sys.argv[2:] if v not in ("tabled", "beside")] or [0]
CURVE, BASE, SCAL = 0, 0, 1
p, r = MODULI[BASE], MODULI[SCAL]
n = 1 << log_n
G = GENERATORS[CURVE]
D = _mul(p, 7, G)
m = lambda f, v: np.array(synth.mont(f, v), dtype=np.uint64)
dev.init(0)
A, B = dev.to_device(synth.rand_field(SCAL, 1, n)), dev.to_device(synth.rand_field(SCAL, 2, n))
g0 = np.stack([m(BASE, G[0]), m(BASE, G[1])]); dd = np.stack([m(BASE, D[0]), m(BASE, D[1])])
Gd = dev.gen_bases_dev(CURVE, n, g0, dd)
X_INT = 0x1F3D5B79A2C4E6081F3D5B79A2C4E6081F3D5B79A2C4E608 % r      # u_prime = [x] U_base (halo.rs:46-47: x = halo_n(u_scaling bits))
UB = _mul(p, 13, G)
H, U = _mul(p, 11, G), _mul(p, X_INT, UB)
Hm, Um = np.stack([m(BASE, H[0]), m(BASE, H[1])]), np.stack([m(BASE, U[0]), m(BASE, U[1])])
# full-size challenges (a small u would make the generator fold of its round unrealistically cheap)
us = [synth.to_int(row) % r or 1 for row in synth.rand_field(SCAL, 3, log_n)]
ums = [(m(SCAL, u), m(SCAL, pow(u, -1, r))) for u in us]
bl = [(m(SCAL, 100 + j), m(SCAL, 200 + j)) for j in range(log_n)]
tables = None
if TABLED:
    # the prover's tables over the circuit's fixed generators [pedersen_g, pedersen_h, U]: they exist before the opening
    UBm = np.stack([m(BASE, UB[0]), m(BASE, UB[1])])
    tables = dev.msm_precompute_dev(CURVE, torch.cat([Gd, dev.to_device(Hm[None]), dev.to_device(UBm[None])])); torch.cuda.synchronize()
    TKW = dict(h_index=n, u_index=n + 1, u_prime_scalar=m(SCAL, X_INT)) if "beside" not in sys.argv else {}
def run(freeze_log, per_round=None, tab=None):
    t0 = time.perf_counter()
    arg = dev.HaloArgument(CURVE, A, B, Gd, Hm, Um, freeze_log=freeze_log, tables=tab, **(TKW if tab is not None else {}))
    torch.cuda.synchronize()
    t_begin = time.perf_counter() - t0
    outs = []
    for j in range(log_n):
        t1 = time.perf_counter()
        outs.append(arg.round_lr(*bl[j]))
        t2 = time.perf_counter()
        arg.round_fold(*ums[j])
        if per_round is not None:
            torch.cuda.synchronize()
            per_round.append((len(arg) * 2, (t2 - t1) * 1e3, (time.perf_counter() - t2) * 1e3, arg.frozen))
    fin = arg.read()
    t = time.perf_counter() - t0
    arg.free()
    return t, t_begin, outs, fin
ref = None
for fz in freezes:
    run(fz)
    t, tb, outs, fin = run(fz)
    if ref is None:
        ref = (outs, fin)
    same = all(np.array_equal(a[0], b[0]) for a, b in zip(outs, ref[0])) and all(np.array_equal(x, y) for x, y in zip(fin, ref[1]))
    print("IPA at n = 2^%d, freeze_log %d: %d rounds, %.2f ms in all (begin %.2f ms incl. copies / contexts); same results as the first setting: %s"
          % (log_n, fz, log_n, t * 1e3, tb * 1e3, same), flush=True)
if TABLED:
    run(freezes[0], tab=tables)
    t, tb, outs, fin = run(freezes[0], tab=tables)
    same = all(np.array_equal(a[0], b[0]) for a, b in zip(outs, ref[0])) and all(np.array_equal(x, y) for x, y in zip(fin, ref[1]))
    print("IPA at n = 2^%d over the caller's tables (PLK_HALO_LEAD=%s), freeze_log %d: %.2f ms in all (begin %.2f ms); same results as the plain argument: %s"
          % (log_n, os.environ.get("PLK_HALO_LEAD", "default"), freezes[0], t * 1e3, tb * 1e3, same), flush=True)
pr = []
run(freezes[0], pr, tab=tables)
for ln, t_lr, t_fold, frozen in pr:
    print("  round at length %8d: L/R %.3f ms  fold %.3f ms  %s" % (ln, t_lr, t_fold, "frozen generators" if frozen else ""))
