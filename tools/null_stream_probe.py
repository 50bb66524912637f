"""GPU probe: what the stream a caller launches on costs.  torch's default stream is HIP's NULL stream; a launch there has to look at
every other stream the process has used.  The headline step (2^20 NTT + 2^20 MSM, one stream) on the NULL stream and on a created stream,
before and after a second stream has been used.   usage: python tools/null_stream_probe.py [steps=200]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plonky_amd import device as dev, synth
from plonky_amd.synth import MODULI
from plonky_amd.selfcheck import _mul
dev.init(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = 1 << 20
p = MODULI[0]
G = (p - 1, 2)
D = _mul(p, 0xC0FFEE1234567, G)
g0 = np.stack([synth.mont(0, G[0]), synth.mont(0, G[1])]); dd = np.stack([synth.mont(0, D[0]), synth.mont(0, D[1])])
bases = dev.gen_bases_dev(0, n, g0, dd)
pre = dev.msm_precompute_dev(0, bases)
s = dev.to_device(synth.rand_field(1, 0x350022, n))
x = dev.to_device(synth.rand_field(0, 0x77, n)); y = torch.empty_like(x)
oxy = torch.empty((1, 2, 4), dtype=torch.int64, device="cuda"); oz = torch.empty((1,), dtype=torch.uint8, device="cuda")


def run(label, stream, ntt=True, msm=True):
    def step():
        if ntt:
            dev.ntt_dev(0, x, out=y)
        if msm:
            dev.msm_execute_dev(pre, s, oxy, oz)
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.default_stream())
    with ctx:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    print("%-64s %.4f ms" % (label, best), flush=True)


run("step on the NULL stream (no other stream used yet)", None)
run("NTT alone on the NULL stream", None, msm=False)
a = torch.cuda.Stream()
run("step on a created stream", a)
run("NTT alone on a created stream", a, msm=False)
run("step on the NULL stream, after a created stream was used", None)
run("NTT alone on the NULL stream, after a created stream was used", None, msm=False)
b = torch.cuda.Stream()
run("step on a second created stream", b)
run("step on the first created stream again", a)

# ---- what bench.py saw: after steps with the NTT on a second stream BESIDE the MSM, the transform's own loop on the first stream
side = torch.cuda.Stream()
y2 = torch.empty_like(x)
for _ in range(60):
    with torch.cuda.stream(side):
        dev.ntt_dev(0, x, out=y2)
    dev.msm_execute_dev(pre, s, oxy, oz)
torch.cuda.synchronize()
run("NTT alone on the NULL stream, after two-stream steps", None, msm=False)
run("NTT alone on that second stream", side, msm=False)
run("MSM alone on the NULL stream", None, ntt=False)
for _ in range(5):
    dev.ntt_dev(0, x, out=y)
torch.cuda.synchronize()
run("NTT alone on the NULL stream, once more", None, msm=False)
