"""GPU debugging helper: compares plk_ntt with the oracle for several forced pass plans."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plonky_amd import lib, synth
from oracle import oracle_lib as ol

L = lib.load()
cases = [(12, "6,6"), (12, "7,5"), (12, "5,7"), (13, "7,6"), (13, "6,7"), (13, "5,8"), (13, "8,5"), (14, "7,7"), (14, "5,5,4"), (16, "6,5,5"), (11, "7,4"), (11,"4,7"), (20, "7,7,6")]
for log_n, plan in cases:
    os.environ["PLK_NTT_PLAN"] = plan
    L.plk_ntt_clear_cache()
    n = 1 << log_n
    x = synth.rand_field(0, 0xF70000 + log_n, n)
    out = np.empty_like(x)
    rc = L.plk_ntt(0, log_n, 0, x.ctypes.data, out.ctypes.data)
    exp = ol.FftPrecomputation(0, n).fft_with_precomputation_power_of_2(x, threads=8)
    bad = np.nonzero((out != exp).any(axis=1))[0]
    print(log_n, plan, "rc", rc, "mismatches", len(bad), "first", bad[:8].tolist(), flush=True)
