"""Samples the shader clock (clock64 against the 100 MHz wall clock) on 8 probe waves while NTT variants run on another
stream.  Needs: hipcc -O2 --offload-arch=gfx950 -shared -fPIC -o build_exp/libclockprobe.so tools/lab/clockprobe.hip and the
variant libraries of tools/ntt_experiments.sh (EXPS="3 6").  Run on the GPU box from the repo root."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
os.environ["PLK_NTT_NO_PIPE"] = "1"
from plonky_amd import synth
vp, i, u = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint
def load(path):
    L = ctypes.CDLL(path)
    L.plk_init.argtypes = [i]; L.plk_ntt_dev.argtypes = [i, u, i, u, vp, vp, vp]
    assert L.plk_init(0) == 0
    return L
root = os.getcwd()
libs = {"compute-only": load(root + "/build_exp/libplonky_hip_e6.so"), "memory-only": load(root + "/build_exp/libplonky_hip_e3.so"),
        "full": load(root + "/plonky_amd/csrc/libplonky_hip.so")}
P = ctypes.CDLL(root + "/build_exp/libclockprobe.so"); P.probe_launch.argtypes = [vp, i, i, i, vp]
x = torch.from_numpy(synth.rand_field(0, 1, 9 << 20).view(np.int64)).cuda()
y = torch.empty_like(x)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
W, BL = 10, 8
buf = torch.zeros((BL * W * 2,), dtype=torch.int64, device="cuda")
def probe(label, L):
    torch.cuda.synchronize()
    if L is not None:
        for _ in range(40):
            L.plk_ntt_dev(0, 20, 0, 9, vp(x.data_ptr()), vp(y.data_ptr()), vp(s1.cuda_stream))
    P.probe_launch(vp(buf.data_ptr()), BL, W, 1000, vp(s2.cuda_stream))
    torch.cuda.synchronize()
    b = buf.cpu().numpy().reshape(BL, W, 2)
    mhz = b[:, :, 1] / (b[:, :, 0] / 100.0)
    print("%-14s shader clock MHz per probe block (mean over windows): %s" % (label, np.round(mhz.mean(axis=1)).astype(int)))
probe("idle", None)
for k, L in libs.items():
    probe(k, L); probe(k, L)
