"""GPU probe: the headline step (one 2^20 NTT + one 2^20 MSM, independent inputs) with both calls on ONE stream against the NTT on a second
stream (the prover's transforms and commitments are independent tasks: plonk.rs runs them under Rayon).  Results are checked equal.
usage: python tools/overlap_probe.py [steps=200]"""
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
x = dev.to_device(synth.rand_field(0, 0x77, n)); y = torch.empty_like(x); y2 = torch.empty_like(x)
oxy = torch.empty((1, 2, 4), dtype=torch.int64, device="cuda"); oz = torch.empty((1,), dtype=torch.uint8, device="cuda")
oxy2 = torch.empty_like(oxy); oz2 = torch.empty_like(oz)


def run(label, side, out_y, out_xy, out_z, ntt=True, msm=True):
    main = torch.cuda.current_stream()
    def step():
        if ntt:
            if side is None:
                dev.ntt_dev(0, x, out=out_y)
            else:
                with torch.cuda.stream(side):
                    dev.ntt_dev(0, x, out=out_y)
        if msm:
            dev.msm_execute_dev(pre, s, out_xy, out_z)
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
    print("%-44s %.4f ms per step" % (label, best), flush=True)
    return best


run("NTT alone", None, y, oxy, oz, msm=False)
run("MSM alone", None, y, oxy, oz, ntt=False)
a = run("one stream (NTT, then MSM)", None, y, oxy, oz)
b = run("NTT on a second stream", torch.cuda.Stream(), y2, oxy2, oz2)
c = run("NTT on a second stream, high priority", torch.cuda.Stream(priority=-1), y2, oxy2, oz2)
d = run("NTT on a second stream, low priority", torch.cuda.Stream(priority=0), y2, oxy2, oz2)
assert torch.equal(y, y2) and torch.equal(oxy, oxy2) and torch.equal(oz, oz2)
print("same results; step %.4f -> %.4f ms (%.1f %%)" % (a, min(b, c), (a / min(b, c) - 1) * 100))
