"""The hot-path calls of one Plonk proof (src/plonk.rs:85-200) strung together on the device, data resident in HBM
between them: 9 wire iNTTs (values_to_polynomials), 9 LDEs to 8n (polynomials_to_values_padded), 9 blinded commitments
(commit_polynomials), Z: one iNTT + one commitment, the quotient t = vanishing / Z_H (divide_by_z_h) and its 7 chunk
commitments.  Constraint evaluation (vanishing_poly, permutation_polynomial) is not part of this repository: the numerator
is a synthetic multiple of Z_H of the right degree.  Usage (GPU box): python tools/prover_pipeline_probe.py [log_n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plonky_amd import api, device as dev, synth
from plonky_amd.selfcheck import _mul
from plonky_amd.synth import MODULI

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
F, CURVE = 1, 0   # wires live in the scalar field of Tweedledee = TweedledumBase
dev.init(0)
p = MODULI[0]
G = (p - 1, 2)
D = _mul(p, 0xC0FFEE, G)
g0 = np.stack([synth.mont(0, G[0]), synth.mont(0, G[1])]); dd = np.stack([synth.mont(0, D[0]), synth.mont(0, D[1])])
gens = dev.gen_bases_dev(CURVE, n + 1, g0, dd)            # pedersen_g (n) followed by pedersen_h
t0 = time.perf_counter(); pre = dev.msm_precompute_dev(CURVE, gens); torch.cuda.synchronize()
print("setup: msm_precompute of %d generators %.1f ms (once per circuit)" % (n + 1, (time.perf_counter() - t0) * 1e3))
wires = dev.to_device(synth.rand_field(F, 1, 9 * n)).reshape(9, n, 4)
blind = dev.to_device(synth.rand_field(F, 2, 9 + 1 + 7)).reshape(-1, 1, 4)
zvals = dev.to_device(synth.rand_field(F, 3, n))
q0 = synth.rand_field(F, 4, 7 * n)
zpad = np.zeros((n, 4), dtype=np.uint64)
numer = dev.to_device(api.field_op(F, "sub", np.concatenate([zpad, q0]), np.concatenate([q0, zpad])))   # q0 * (X^n - 1)
t_out = torch.empty((8 * n, 4), dtype=torch.int64, device="cuda")
ev8 = torch.empty((9, 8 * n, 4), dtype=torch.int64, device="cuda")


def run():
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
    marks[0].record()
    polys = dev.ntt_dev(F, wires, inverse=True)                                     # values_to_polynomials
    marks[1].record()
    dev.ntt_padded_dev(F, polys, log_n + 3, out=ev8)                                # polynomials_to_values_padded
    marks[2].record()
    c_wires = dev.msm_execute_dev(pre, torch.cat([polys, blind[:9]], dim=1).contiguous())   # commit_polynomials (+ [r] H)
    marks[3].record()
    zpoly = dev.ntt_dev(F, zvals, inverse=True)
    c_z = dev.msm_execute_dev(pre, torch.cat([zpoly, blind[9]], dim=0).contiguous())
    marks[4].record()
    t = dev.divide_by_z_h_dev(F, numer, n, out=t_out)                               # quotient
    marks[5].record()
    chunks = t[: 7 * n].reshape(7, n, 4)                                            # pad to 7n, split
    c_t = dev.msm_execute_dev(pre, torch.cat([chunks, blind[10:17]], dim=1).contiguous())
    marks[6].record()
    torch.cuda.synchronize()
    names = ["9 wire iNTT (n)", "9 LDE n -> 8n", "9 wire commitments", "Z: iNTT + commitment", "divide_by_z_h (8n domain)", "7 quotient-chunk commitments"]
    ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(6)]
    return names, ms, t


run()
names, ms, t = run()
ok = bool(np.array_equal(dev.to_host(t[: 7 * n]), q0))
for nm, v in zip(names, ms):
    print("  %-32s %8.3f ms" % (nm, v))
print("hot-path device time per proof at n = 2^%d: %.2f ms   (quotient check: %s)" % (log_n, sum(ms), ok))
