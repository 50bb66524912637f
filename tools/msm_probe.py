"""GPU probe: MSM timing + closed-form check for any curve / size (device-resident)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plonky_amd import device as dev, synth, lib
from plonky_amd.selfcheck import closed_form_msm, _mul, CURVE_BASE, CURVE_SCALAR
from plonky_amd.synth import MODULI
dev.init(0)
GEN = {0: (MODULI[0] - 1, 2),
       2: (81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
           241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030)}
for curve, log_n in [(int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1:]] or [(2, 16), (2, 20)]:
    n = 1 << log_n
    bf, sf = CURVE_BASE[curve], CURVE_SCALAR[curve]
    p = MODULI[bf]
    G = GEN[curve]
    D = _mul(p, 0xC0FFEE1234567, G)
    g0 = np.stack([synth.mont(bf, G[0]), synth.mont(bf, G[1])]); dd = np.stack([synth.mont(bf, D[0]), synth.mont(bf, D[1])])
    t0 = time.perf_counter(); bases = dev.gen_bases_dev(curve, n, g0, dd); torch.cuda.synchronize(); t_gen = time.perf_counter() - t0
    s_host = synth.rand_field(sf, 0x350022, n)
    s = dev.to_device(s_host)
    t0 = time.perf_counter(); pre = dev.msm_precompute_dev(curve, bases); torch.cuda.synchronize(); t_pre = time.perf_counter() - t0
    oxy, oz = dev.msm_execute_dev(pre, s); torch.cuda.synchronize()
    K = int(os.environ.get("PROBE_ITERS", "5"))
    t0 = time.perf_counter()
    for _ in range(K): dev.msm_execute_dev(pre, s, oxy, oz)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / K
    L = 6 if curve == 2 else 4
    got = dev.to_host(oxy).reshape(2, L)
    exp = closed_form_msm(curve, s_host, G, D)
    ok = (synth.from_mont(bf, got[0]), synth.from_mont(bf, got[1])) == exp and int(oz.cpu()[0]) == 0
    print("curve", curve, "log_n", log_n, "window", pre.window, "gen %.1f ms precompute %.1f ms execute %.3f ms  %.1f Mpairs/s  closed-form ok: %s" % (t_gen * 1e3, t_pre * 1e3, t * 1e3, n / t / 1e6, ok), flush=True)
    pre.free(); del bases, s
