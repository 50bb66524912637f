"""GPU probe: what an IPA round costs when L_j / R_j are taken over the CALLER's commitment tables (2^20 generators, window 20)
instead of over folded generators: a batch of two scalar vectors with disjoint supports (each half zeros), for several
block patterns (round k: blocks of n / 2^k).  Usage: python tools/gpu/lead_probe.py [log_n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from plonky_amd import device as dev, synth
from plonky_amd.selfcheck import _mul
from plonky_amd.synth import MODULI
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
dev.init(0)
p = MODULI[0]
G = (p - 1, 2); D = _mul(p, 0xC0FFEE, G)
g0 = np.stack([synth.mont(0, G[0]), synth.mont(0, G[1])]); dd = np.stack([synth.mont(0, D[0]), synth.mont(0, D[1])])
gens = dev.gen_bases_dev(0, n, g0, dd)
pre = dev.msm_precompute_dev(0, gens); torch.cuda.synchronize()
print("window", pre.window)
full = synth.rand_field(1, 77, n)
def timed(sc, K=5):
    s = dev.to_device(sc)
    oxy, oz = dev.msm_execute_dev(pre, s); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): dev.msm_execute_dev(pre, s, oxy, oz)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
print("one full vector            %.3f ms" % timed(full[None]))
print("two full vectors           %.3f ms" % timed(np.stack([full, full[::-1].copy()])))
for k in (1, 2, 3, 4, 6):
    blk = n >> k
    idx = np.arange(n)
    hi = ((idx // (blk)) & 1).astype(bool)       # the upper half of every block of 2 blk ... round k splits blocks of n / 2^(k-1)
    L = full.copy(); L[~hi] = 0
    R = full.copy(); R[hi] = 0
    print("round %d (blocks of %7d): L + R as a batch of two %.3f ms" % (k, blk, timed(np.stack([L, R]))))
