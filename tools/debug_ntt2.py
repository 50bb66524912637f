"""GPU debugging helper: replicate the pytest order for test_ntt_matches_oracle and report mismatch patterns."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonky_amd as pa
from plonky_amd import lib, synth
from oracle import oracle_lib as ol

for log_n in list(range(0, 15)) + [16]:
    for field in (0, 1, 2):
        n = 1 << log_n
        x = synth.rand_field(field, 0xF70000 + log_n, n)
        pre = pa.fft_precompute(field, n)
        opre = ol.FftPrecomputation(field, n)
        fwd = pa.fft_with_precomputation_power_of_2(x, pre)
        exp = opre.fft_with_precomputation_power_of_2(x, threads=4)
        bad = np.nonzero((fwd != exp).any(axis=1))[0]
        fwd2 = pa.fft_with_precomputation_power_of_2(x, pre)
        bad2 = np.nonzero((fwd2 != exp).any(axis=1))[0]
        inv = pa.ifft_with_precomputation_power_of_2(x, pre)
        iexp = opre.ifft_with_precomputation_power_of_2(x, threads=4)
        badi = np.nonzero((inv != iexp).any(axis=1))[0]
        if len(bad) or len(bad2) or len(badi):
            print(log_n, field, "fwd mism", len(bad), bad[:6].tolist(), bad[-3:].tolist(), "| 2nd call", len(bad2), "| inv", len(badi), flush=True)
print("done")
