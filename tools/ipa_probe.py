"""GPU probe: all log2(n) rounds of the inner-product argument of one opening (halo.rs:63-124) on the device -- per round
the two L / R terms (one table-free MSM each, blinding and inner-product terms folded in), the two scalar folds and the
generator fold -- timed as a whole (correctness: tests/test_gpu_halo.py, one round against the oracle and a whole argument
against the closed form of the folded vectors).
Usage: python tools/ipa_probe.py [log_n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plonky_amd import device as dev, synth
from plonky_amd.selfcheck import GENERATORS, _mul
from plonky_amd.synth import MODULI
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
CURVE, BASE, SCAL = 0, 0, 1
p, r = MODULI[BASE], MODULI[SCAL]
n = 1 << log_n
G = GENERATORS[CURVE]
D = _mul(p, 7, G)
m = lambda f, v: np.array(synth.mont(f, v), dtype=np.uint64)
A, B = dev.to_device(synth.rand_field(SCAL, 1, n)), dev.to_device(synth.rand_field(SCAL, 2, n))
g0 = np.stack([m(BASE, G[0]), m(BASE, G[1])]); dd = np.stack([m(BASE, D[0]), m(BASE, D[1])])
Gd = dev.gen_bases_dev(CURVE, n, g0, dd)
H, U = _mul(p, 11, G), _mul(p, 13, G)
Hm, Um = np.stack([m(BASE, H[0]), m(BASE, H[1])]), np.stack([m(BASE, U[0]), m(BASE, U[1])])
us = [1 + 17 * j for j in range(log_n)]
def run():
    a, b, g, gz = A, B, Gd, None
    outs = []
    for j in range(log_n):
        lr, lrz = dev.halo_round_lr_dev(CURVE, a, b, g, Hm, Um, m(SCAL, 100 + j), m(SCAL, 200 + j), g_zero=gz)
        a, b, g, gz = dev.halo_round_fold_dev(CURVE, a, b, g, m(SCAL, us[j]), m(SCAL, pow(us[j], -1, r)), g_zero=gz)
        outs.append(lr)
    return a, b, g, gz, outs
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); a, b, g, gz, outs = run(); torch.cuda.synchronize(); t = time.perf_counter() - t0
print("IPA at n = 2^%d: %d rounds, %.2f ms in all (%.2f ms per round on average); final vectors of length %d"
      % (log_n, log_n, t * 1e3, t * 1e3 / log_n, a.shape[0]))
