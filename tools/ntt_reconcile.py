"""Why did tools/ntt_probe.py report 10.4-10.6 G elements/s for nine 2^20 transforms in one call where bench.py reports 8.7-9.0 G?
(round-3 review, "what's weak" 5).  Same entry point (plk_ntt_dev), same clock.  The two harnesses differ in
  (a) the DATA: the probe drew torch.randint(0, 2^60) limbs (the top four bits of every 64-bit limb zero, values far below p in the
      top limb), bench.py draws uniform field elements (synth.rand_field);
  (b) the batch layout: bench.py times nine COPIES of one vector, the probe nine different vectors;
  (c) the loop: the probe 3 warm-up + 20 timed calls, bench.py 1 + 10.
This script times every combination with HIP events AND the wall clock, alternating the variants so that they share the GPU's
clock / temperature state, and prints the shader clock rocm-smi reports under each."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from plonky_amd import device as dev, synth

dev.init(0)
LOG_N, B = 20, 9
n = 1 << LOG_N


def sclk():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=20).stdout
        for line in out.splitlines():
            if "sclk" in line:
                return line.split(":")[-1].strip()
    except Exception:
        pass
    return "?"


def data(kind):
    if kind == "randint60":
        x = torch.randint(0, 1 << 60, (B, n, 4), dtype=torch.int64, device="cuda")
        x[..., 3] &= (1 << 61) - 1
        return x
    if kind == "uniform9":
        return dev.to_device(np.stack([synth.rand_field(0, 0xF70020 + v, n) for v in range(B)]))
    if kind == "uniform_copies":
        return dev.to_device(synth.rand_field(0, 0xF70020, n)).unsqueeze(0).repeat(B, 1, 1).contiguous()
    if kind == "zeros":
        return torch.zeros((B, n, 4), dtype=torch.int64, device="cuda")
    raise ValueError(kind)


def timed(x, y, warm, iters):
    for _ in range(warm):
        dev.ntt_dev(0, x, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(iters):
        dev.ntt_dev(0, x, out=y)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters
    return e0.elapsed_time(e1) / iters * 1e-3, wall


kinds = ["randint60", "uniform9", "uniform_copies", "zeros"]
xs = {k: data(k) for k in kinds}
y = torch.empty_like(xs["uniform9"])
print("# nine 2^20 TweedledeeBase transforms in one plk_ntt_dev call; G elements/s by HIP events (wall clock in brackets)")
for rnd in range(3):
    for warm, iters in ((3, 20), (1, 10), (3, 100)):
        row = []
        for k in (kinds if rnd % 2 == 0 else kinds[::-1]):
            ev, wall = timed(xs[k], y, warm, iters)
            row.append("%s %.2f (%.2f)" % (k, B * n / ev / 1e9, B * n / wall / 1e9))
        print("round %d warm %d iters %3d: %s | sclk %s" % (rnd, warm, iters, "; ".join(sorted(row)), sclk()), flush=True)
# the single transform, both data kinds
x1u, x1r = xs["uniform9"][0].contiguous(), xs["randint60"][0].contiguous()
y1 = torch.empty_like(x1u)
for rnd in range(3):
    a, _ = timed(x1u, y1, 3, 50)
    b, _ = timed(x1r, y1, 3, 50)
    print("single 2^20 transform: uniform %.1f us, randint60 %.1f us" % (a * 1e6, b * 1e6), flush=True)
