"""GPU probe: nine 2^20 transforms through the HOST-pointer entry point plk_ntt_batch (pageable numpy buffers, PCIe inside), the round-4
form (transform b on stream b mod 3: PLK_NTT_HOST_PIPE=0) against one stream per direction (round 5), alternating in one process."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from plonky_amd import lib, synth
L = lib.load()
lib.check(L.plk_init(0))
vp = ctypes.c_void_p
log_n, n = 20, 1 << 20
hin = [np.ascontiguousarray(synth.rand_field(0, 0xF70020 + b, n)) for b in range(9)]
hout = [np.zeros_like(hin[0]) for _ in range(9)]
ins = (vp * 9)(*[a.ctypes.data for a in hin])
outs = (vp * 9)(*[a.ctypes.data for a in hout])
res = {"0": [], "1": []}
ref = None
for rep in range(12):
    for mode in ("0", "1"):
        os.environ["PLK_NTT_HOST_PIPE"] = mode
        lib.check(L.plk_ntt_batch(0, log_n, 0, 9, ins, outs))
        t0 = time.perf_counter()
        for _ in range(3):
            lib.check(L.plk_ntt_batch(0, log_n, 0, 9, ins, outs))
        res[mode].append((time.perf_counter() - t0) / 3 * 1e3)
        cur = [o.copy() for o in hout]
        if ref is None:
            ref = cur
        assert all(np.array_equal(a, b) for a, b in zip(ref, cur))
for mode, name in (("0", "transform b on stream b mod 3 (round 4)"), ("1", "one stream per direction (round 5)")):
    v = sorted(res[mode])
    print("host_ntt9  %-42s median %.2f ms  min %.2f ms" % (name, v[len(v) // 2], v[0]))
print("one-way PCIe floor at 56 GB/s: %.2f ms" % (9 * n * 32 / 56e9 * 1e3))
