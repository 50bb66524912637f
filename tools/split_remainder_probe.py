"""GPU probe: one rank's share of the nine-commitment batch at 8 ranks (one whole 2^20 vector + the 2^17-generator share of the ninth) -
  (a) the product's form: ONE batched call over the full tables, the share passed with its base range (plk_msm_execute_parts_dev);
  (b) two calls on two streams: the whole vector over the full tables, the share over a context of its own (its own window, tables of
      2^17 generators), joined by an event."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plonky_amd import device as dev, synth
from plonky_amd.selfcheck import _mul, CURVE_BASE, CURVE_SCALAR, GENERATORS
from plonky_amd.synth import MODULI
dev.init(0)
curve, log_n, N = 0, 20, int(sys.argv[1]) if len(sys.argv) > 1 else 8
bf, sf = CURVE_BASE[curve], CURVE_SCALAR[curve]
p = MODULI[bf]
G = GENERATORS[curve]
D = _mul(p, 0xC0FFEE1234567, G)
g0 = np.stack([synth.mont(bf, G[0]), synth.mont(bf, G[1])]); dd = np.stack([synth.mont(bf, D[0]), synth.mont(bf, D[1])])
n = 1 << log_n
m = n // N
bases = dev.gen_bases_dev(curve, n, g0, dd)
pre = dev.msm_precompute_dev(curve, bases)
pre_share = dev.msm_precompute_dev(curve, bases[:m].contiguous())
s_whole = dev.to_device(synth.rand_field(sf, 0x901, n))
s_part = dev.to_device(synth.rand_field(sf, 0x902, m))
parts = [(0, s_whole), (0, s_part)]
oxy, oz = dev.msm_execute_parts_dev(pre, parts)
o1, z1 = dev.msm_execute_dev(pre, s_whole)
o2, z2 = dev.msm_execute_dev(pre_share, s_part)
torch.cuda.synchronize()
assert np.array_equal(dev.to_host(oxy)[0], dev.to_host(o1)[0]) and np.array_equal(dev.to_host(oxy)[1], dev.to_host(o2)[0])
side = torch.cuda.Stream()
def form_a():
    dev.msm_execute_parts_dev(pre, parts, oxy, oz)
def form_b():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        dev.msm_execute_dev(pre_share, s_part, o2, z2)
    dev.msm_execute_dev(pre, s_whole, o1, z1)
    main.wait_stream(side)
def form_c():
    dev.msm_execute_dev(pre_share, s_part, o2, z2)
    dev.msm_execute_dev(pre, s_whole, o1, z1)
for name, f in (("one batched call over the full tables (parts)", form_a), ("two calls, two streams, share context", form_b), ("two calls, one stream, share context", form_c)):
    for _ in range(3): f()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        for _ in range(20): f()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20)
    print("N = %d  %-50s %.3f ms (share window %d)" % (N, name, best * 1e3, pre_share.window), flush=True)
