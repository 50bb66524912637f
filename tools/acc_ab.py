"""Same-lease A/B of several builds of libplonky_hip.so: one 2^20 Tweedledee MSM, per-stage HIP-event timings
(plk_msm_get_timings), the builds ALTERNATING inside one process so that clocks, temperature and the physical GPU are shared.

    python tools/acc_ab.py [--log-n 20] [--reps 30] [--inner 5] name=path/to/lib.so name2=path2 ...

Only entry points that exist since round 2 are used, so a library built from an old commit loads too.  Prints one line per
build: median / min of the accumulation stage and of the whole execution, and the GPU's UUID.
"""
import argparse
import ctypes
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from plonky_amd import synth
from plonky_amd.selfcheck import closed_form_msm, _mul
from plonky_amd.synth import MODULI

vp = ctypes.c_void_p


class Build:
    def __init__(self, name, path):
        self.name, self.path = name, path
        L = ctypes.CDLL(os.path.abspath(path), mode=ctypes.RTLD_LOCAL)
        L.plk_last_error.restype = ctypes.c_char_p
        L.plk_curve_gen_bases_dev.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_uint64, vp, vp, vp, vp]
        L.plk_msm_precompute_dev.argtypes = [ctypes.c_int, ctypes.c_size_t, vp, vp, ctypes.c_uint, vp, vp]
        L.plk_msm_execute_dev.argtypes = [vp, ctypes.c_uint, vp, ctypes.c_size_t, vp, vp, vp]
        L.plk_msm_set_profiling.argtypes = [vp, ctypes.c_int]
        L.plk_msm_get_timings.argtypes = [vp, vp, vp]
        L.plk_msm_free.argtypes = [vp]
        self.L = L
        self.ctx = vp()

    def check(self, rc):
        if rc != 0:
            raise RuntimeError("%s: %d %s" % (self.name, rc, self.L.plk_last_error().decode()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--inner", type=int, default=5)
    ap.add_argument("builds", nargs="+")
    a = ap.parse_args()
    torch.cuda.init()
    props = torch.cuda.get_device_properties(0)
    uuid = getattr(props, "uuid", "?")
    n = 1 << a.log_n
    p = MODULI[0]
    G = (p - 1, 2)
    D = _mul(p, 0xC0FFEE1234567, G)
    g0 = np.stack([synth.mont(0, G[0]), synth.mont(0, G[1])])
    dd = np.stack([synth.mont(0, D[0]), synth.mont(0, D[1])])
    s_host = synth.rand_field(1, 0x350020, n)
    s = torch.from_numpy(s_host.view(np.int64)).cuda()
    exp = closed_form_msm(0, s_host, G, D)
    stream = vp(torch.cuda.current_stream().cuda_stream)
    builds = [Build(*b.split("=", 1)) for b in a.builds]
    bases = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    oxy = torch.zeros(8, dtype=torch.int64, device="cuda")
    oz = torch.zeros(1, dtype=torch.uint8, device="cuda")
    for b in builds:
        b.check(b.L.plk_init(0))
        b.check(b.L.plk_curve_gen_bases_dev(0, n, 0, g0.ctypes.data_as(vp), dd.ctypes.data_as(vp), vp(bases.data_ptr()), stream))
        b.check(b.L.plk_msm_precompute_dev(0, n, vp(bases.data_ptr()), None, 0, stream, ctypes.byref(b.ctx)))
        b.check(b.L.plk_msm_execute_dev(b.ctx, 1, vp(s.data_ptr()), n, vp(oxy.data_ptr()), vp(oz.data_ptr()), stream))
        torch.cuda.synchronize()
        got = oxy.cpu().numpy().view(np.uint64).reshape(2, 4)
        ok = (synth.from_mont(0, got[0]), synth.from_mont(0, got[1])) == exp
        print("# %s: %s closed form %s" % (b.name, b.path, "ok" if ok else "MISMATCH"), flush=True)
        b.acc, b.wall, b.stages = [], [], []
    for rep in range(a.reps):
        order = builds if rep % 2 == 0 else builds[::-1]
        for b in order:
            # wall time of `inner` executions without the per-stage events, then the same with them
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.inner):
                b.check(b.L.plk_msm_execute_dev(b.ctx, 1, vp(s.data_ptr()), n, vp(oxy.data_ptr()), vp(oz.data_ptr()), stream))
            torch.cuda.synchronize()
            b.wall.append((time.perf_counter() - t0) / a.inner * 1e3)
            b.check(b.L.plk_msm_set_profiling(b.ctx, 1))
            for _ in range(a.inner):
                b.check(b.L.plk_msm_execute_dev(b.ctx, 1, vp(s.data_ptr()), n, vp(oxy.data_ptr()), vp(oz.data_ptr()), stream))
            torch.cuda.synchronize()
            ms = (ctypes.c_double * 7)()
            calls = ctypes.c_uint()
            b.check(b.L.plk_msm_get_timings(b.ctx, ms, ctypes.byref(calls)))
            b.check(b.L.plk_msm_set_profiling(b.ctx, 0))
            b.acc.append(ms[3] / max(1, calls.value))
            b.stages.append([v / max(1, calls.value) for v in ms])
    print("gpu %s uuid %s, 2^%d Tweedledee, %d alternating reps x %d executions" % (props.name, uuid, a.log_n, a.reps, a.inner))
    for b in builds:
        st = [statistics.median(x[i] for x in b.stages) for i in range(7)]
        print("%-14s accumulate median %.4f min %.4f ms | execution median %.4f min %.4f ms | stages %s" % (
            b.name, statistics.median(b.acc), min(b.acc), statistics.median(b.wall), min(b.wall), " ".join("%.4f" % v for v in st)))
    for b in builds:
        b.L.plk_msm_free(b.ctx)


if __name__ == "__main__":
    main()
