"""GPU probe: NTT time per transform vs batch size and size (device-resident), HIP-event timed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plonky_amd import device as dev, synth
dev.init(0)
for log_n, batch in ((20, 1), (20, 4), (20, 9), (14, 64), (16, 16), (18, 4), (22, 1), (23, 1)):
    x = torch.randint(0, 1 << 60, (batch, 1 << log_n, 4), dtype=torch.int64, device="cuda")
    x[..., 3] &= (1 << 61) - 1
    y = torch.empty_like(x)
    for _ in range(3):
        dev.ntt_dev(0, x, out=y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        dev.ntt_dev(0, x, out=y)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / K
    print("log_n", log_n, "batch", batch, "us/transform %.1f" % (t / batch * 1e6), "Gelem/s %.2f" % (batch * (1 << log_n) / t / 1e9), flush=True)
