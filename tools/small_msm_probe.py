"""GPU probe: two-vector tabled MSMs over few generators (the frozen rounds of the inner-product argument: 2^14 + 2 generators), time per
execution and per stage.  usage: python tools/small_msm_probe.py [log_n ...]   (default 10 12 14 16)"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plonky_amd import device as dev, synth, lib
from plonky_amd.synth import MODULI
from plonky_amd.selfcheck import _mul
dev.init(0)
L = lib.load()
STAGES = ["order_count", "order_scatter", "order_buckets", "accumulate", "assemble_lines", "planes", "final"]
p = MODULI[0]
G = (p - 1, 2)
D = _mul(p, 0xC0FFEE1234567, G)
g0 = np.stack([synth.mont(0, G[0]), synth.mont(0, G[1])]); dd = np.stack([synth.mont(0, D[0]), synth.mont(0, D[1])])
iters = int(os.environ.get("PROBE_ITERS", "200"))
wins = [int(w) for w in os.environ.get("PROBE_WINDOWS", "0").split(",")]
for log_n, win in [(int(a), w) for a in (sys.argv[1:] or [10, 12, 14, 16]) for w in wins]:
    n = (1 << log_n) + 2
    if win:
        os.environ["PLK_MSM_WINDOW"] = str(win)   # read at every precompute (msm.hip choose_window)
    bases = dev.gen_bases_dev(0, n, g0, dd)
    pre = dev.msm_precompute_dev(0, bases)
    s = dev.to_device(np.stack([synth.rand_field(1, 0x350022 + k, n) for k in range(2)]))
    oxy, oz = dev.msm_execute_dev(pre, s)
    for _ in range(10):
        dev.msm_execute_dev(pre, s, oxy, oz)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        dev.msm_execute_dev(pre, s, oxy, oz)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / iters * 1e3
    line = "2 x (2^%d + 2) pairs, window %d: %.4f ms per call" % (log_n, pre.window, t)
    if not os.environ.get("PROBE_NO_STAGES"):
        L.plk_msm_set_profiling(pre._ctx, 1)
        for _ in range(20):
            dev.msm_execute_dev(pre, s, oxy, oz)
        torch.cuda.synchronize()
        arr = (ctypes.c_double * 7)(); calls = ctypes.c_uint(0)
        L.plk_msm_get_timings(pre._ctx, arr, ctypes.byref(calls))
        L.plk_msm_set_profiling(pre._ctx, 0)
        line += "   stages (us, per profiled MSM): " + "  ".join("%s %.0f" % (k, v / max(1, calls.value) * 1e3) for k, v in zip(STAGES, arr))
    print(line, flush=True)
    pre.free(); del bases, s
