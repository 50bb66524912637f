"""GPU probe: one rank's share of the nine-commitment batch at N ranks under two decompositions -
  (a) the product's plan (parallel.BatchPlan: whole vectors per rank + the remainder sharded by base range), via bench.py --emulate-rank;
  (b) EVERY vector sharded by base range: nine MSMs over the rank's 2^20 / N generators (a context of its own, its own window), one batched call.
Prints the time of (b) for N = 2, 4, 8 and the one-GPU time of nine full MSMs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plonky_amd import device as dev, synth
from plonky_amd.selfcheck import _mul, CURVE_BASE, CURVE_SCALAR, GENERATORS
from plonky_amd.synth import MODULI
dev.init(0)
curve, log_n = 0, 20
bf, sf = CURVE_BASE[curve], CURVE_SCALAR[curve]
p = MODULI[bf]
G = GENERATORS[curve]
D = _mul(p, 0xC0FFEE1234567, G)
g0 = np.stack([synth.mont(bf, G[0]), synth.mont(bf, G[1])]); dd = np.stack([synth.mont(bf, D[0]), synth.mont(bf, D[1])])
for N in (1, 2, 4, 8):
    n = (1 << log_n) // N
    bases = dev.gen_bases_dev(curve, n, g0, dd)
    pre = dev.msm_precompute_dev(curve, bases)
    s = dev.to_device(np.stack([synth.rand_field(sf, 0x900 + k, n) for k in range(9)]))
    oxy, oz = dev.msm_execute_dev(pre, s)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        K = 10
        for _ in range(K):
            dev.msm_execute_dev(pre, s, oxy, oz)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K)
    print("N = %d: nine MSMs over %d generators each (window %d) in one call: %.3f ms" % (N, n, pre.window, best * 1e3), flush=True)
    pre.free(); del bases, s
