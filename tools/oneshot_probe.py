"""GPU probe: msm_parallel (one-shot, table-free) at 2^log_n on Tweedledee, ITERS calls; for rocprofv3 traces of that path alone.
usage: python tools/oneshot_probe.py [log_n=20] [iters=10]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plonky_amd import device as dev, synth
from plonky_amd.synth import MODULI
from plonky_amd.selfcheck import _mul
dev.init(0)
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 1 << log_n
p = MODULI[0]
G = (p - 1, 2)
D = _mul(p, 0xC0FFEE1234567, G)
g0 = np.stack([synth.mont(0, G[0]), synth.mont(0, G[1])]); dd = np.stack([synth.mont(0, D[0]), synth.mont(0, D[1])])
bases = dev.gen_bases_dev(0, n, g0, dd)
s = dev.to_device(synth.rand_field(1, 0x350022, n))
oxy = torch.empty((1, 2, 4), dtype=torch.int64, device="cuda"); oz = torch.empty((1,), dtype=torch.uint8, device="cuda")
for k in range(iters + 2):
    if k == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    pre = dev.msm_precompute_dev(0, bases, table_free=True)
    dev.msm_execute_dev(pre, s, oxy, oz)
    torch.cuda.synchronize()
    pre.free()
print("one-shot 2^%d: %.3f ms per call" % (log_n, (time.perf_counter() - t0) / iters * 1e3))
