"""Does replaying the MSM's ~15 launches from a hipGraph shorten the step?  (inter-kernel gaps of a 1.7 ms call)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from plonky_amd import device as dev, synth
from plonky_amd.selfcheck import GENERATORS, _mul
from plonky_amd.synth import MODULI
dev.init(0)
n = 1 << 20
G = GENERATORS[0]; p = MODULI[0]; D = _mul(p, 424242, G)
g0 = np.stack([synth.mont(0, G[0]), synth.mont(0, G[1])]); dd = np.stack([synth.mont(0, D[0]), synth.mont(0, D[1])])
bases = dev.gen_bases_dev(0, n, g0, dd)
pre = dev.msm_precompute_dev(0, bases)
s = dev.to_device(synth.rand_field(1, 5, n))
oxy = torch.empty((1, 2, 4), dtype=torch.int64, device="cuda"); oz = torch.empty((1,), dtype=torch.uint8, device="cuda")
def loop(fn, k=30):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
print("stream launches: %.4f ms" % loop(lambda: dev.msm_execute_dev(pre, s, oxy, oz)))
ref = oxy.clone()
try:
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        dev.msm_execute_dev(pre, s, oxy, oz)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        dev.msm_execute_dev(pre, s, oxy, oz)
    oxy.zero_()
    print("graph replay:    %.4f ms" % loop(g.replay), "same result:", bool(torch.equal(oxy, ref)))
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
x = dev.to_device(synth.rand_field(0, 6, n)); y = torch.empty_like(x)
print("ntt stream launches: %.4f ms" % loop(lambda: dev.ntt_dev(0, x, out=y), 100))
try:
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        dev.ntt_dev(0, x, out=y)
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=st):
        dev.ntt_dev(0, x, out=y)
    print("ntt graph replay:    %.4f ms" % loop(g2.replay, 100))
except Exception as e:
    print("ntt graph capture failed:", repr(e)[:300])
