"""table-free (one-shot) MSM at 2^20: where the time goes (precompute / execute / free), per window size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from plonky_amd import device as dev, synth
from plonky_amd.selfcheck import GENERATORS
CURVE, BASE, SCALAR = 0, 0, 1
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
sync = torch.cuda.synchronize
G = GENERATORS[CURVE]
g0 = np.stack([synth.mont(BASE, G[0]), synth.mont(BASE, G[1])])
bases = dev.gen_bases_dev(CURVE, n, g0, g0)
s = dev.to_device(synth.rand_field(SCALAR, 77, n))
oxy = torch.empty((1, 2, 4), dtype=torch.int64, device="cuda"); oz = torch.empty((1,), dtype=torch.uint8, device="cuda")
ref = None
for c in ["", "12", "13", "14", "15", "16"]:
    if c: os.environ["PLK_MSM_WINDOW_TF"] = c
    ts = np.zeros(3)
    reps = 6
    for r in range(reps + 1):
        sync(); t0 = time.perf_counter()
        pre = dev.msm_precompute_dev(CURVE, bases, table_free=True); sync(); t1 = time.perf_counter()
        dev.msm_execute_dev(pre, s, oxy, oz); sync(); t2 = time.perf_counter()
        pre.free(); sync(); t3 = time.perf_counter()
        if r: ts += [t1 - t0, t2 - t1, t3 - t2]
    ts *= 1e3 / reps
    if ref is None: ref = oxy.clone()
    print("window %-3s precompute %.3f ms  execute %.3f ms  free %.3f ms  total %.3f ms  same result %s" % (c or "def", ts[0], ts[1], ts[2], ts.sum(), bool(torch.equal(ref, oxy))))
