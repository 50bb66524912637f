"""GPU probe (tuning build -DPLK_FINAL_TRACE, ab_libs/libplonky_hip_trace.so): where k_msm_final's time goes.  Runs a 2^log_n tabled MSM a
few times, then prints the shader-clock stamps thread 0 of the two blocks (column sums, row sums) left along the chain, as differences."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plonky_amd import device as dev, synth, lib
from plonky_amd.selfcheck import _mul, CURVE_BASE, CURVE_SCALAR, GENERATORS
from plonky_amd.synth import MODULI
dev.init(0)
L = lib.load()
curve = 0
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
bf, sf = CURVE_BASE[curve], CURVE_SCALAR[curve]
p = MODULI[bf]
G = GENERATORS[curve]
D = _mul(p, 0xC0FFEE1234567, G)
g0 = np.stack([synth.mont(bf, G[0]), synth.mont(bf, G[1])]); dd = np.stack([synth.mont(bf, D[0]), synth.mont(bf, D[1])])
bases = dev.gen_bases_dev(curve, n, g0, dd)
s = dev.to_device(synth.rand_field(sf, 0x350022, n))
pre = dev.msm_precompute_dev(curve, bases)
oxy, oz = dev.msm_execute_dev(pre, s)
torch.cuda.synchronize()
names = ["start", "loaded + parts added", "plane doubled into place", "barrier 1", "tree over the planes", "barrier 2", "shifted into place",
         "arrival counted", "windows added", "normalised (inversion)"]
for rep in range(3):
    dev.msm_execute_dev(pre, s, oxy, oz)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    assert L.plk_debug_final_trace(buf) == 0
    t = np.array(list(buf), dtype=np.uint64).reshape(2, 16).astype(np.int64)
    for b in range(2):
        row = t[b]
        out = []
        for i in range(1, 10):
            if row[i] and row[i - 1] and row[i] >= row[i - 1]:
                out.append("%s +%d" % (names[i], row[i] - row[i - 1]))
        if row[10] and row[11] and row[12] and row[8]:
            out.append("inside the normalisation: ZZZ to words +%d, inversion +%d, products + stores +%d" % (row[10] - row[8], row[11] - row[10], row[12] - row[11]))
            if row[15]:
                out.append("inside the inversion: %d iterations, low-word steps %d, matrix application %d" % (row[15], row[13], row[14]))
        print("window", pre.window, "block", b, "total", int(max(row[:10]) - row[0]), "ticks |", " | ".join(out), flush=True)
