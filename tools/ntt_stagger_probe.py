"""GPU probe: a single device-resident NTT (one round of workgroups per pass) with the staggered start of k_ntt_pass, PLK_NTT_STAGGER
from the environment (one process per setting: tools/gpu/r05_ntt_stagger.sh).  Steady-state loop time + the round trip as a check."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from plonky_amd import device as dev, synth
dev.init(0)
out = []
for log_n in (20, 19, 18):
    xh = synth.rand_field(0, 0xF70020, 1 << log_n)
    x = dev.to_device(xh)
    y = torch.empty_like(x)
    for _ in range(20):
        dev.ntt_dev(0, x, out=y)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        K = 200
        for _ in range(K):
            dev.ntt_dev(0, x, out=y)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K)
    ok = np.array_equal(dev.to_host(dev.ntt_dev(0, y, inverse=True)), xh)
    out.append("2^%d %.1f us%s" % (log_n, best * 1e6, "" if ok else " ROUNDTRIP-MISMATCH"))
print("PLK_NTT_STAGGER=%s  " % os.environ.get("PLK_NTT_STAGGER", "(default)") + "   ".join(out), flush=True)
