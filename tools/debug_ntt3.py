"""GPU debugging helper: field-op traffic first (as the pytest file does), then NTT 2^13; mismatch pattern."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonky_amd as pa
from plonky_amd import api, lib, synth
from oracle import oracle_lib as ol

mode = sys.argv[1] if len(sys.argv) > 1 else "field"
if mode == "field":
    for field in (0, 1, 2, 3):
        x = synth.rand_field(field, 1, 91204)
        y = synth.rand_field(field, 2, 91204)
        for op in ("add", "sub", "mul"):
            api.field_op(field, op, x, y)
        api.field_op(field, "inverse", x[:4096])
for log_n in (11, 12, 13, 13, 14):
    for field in (0, 1, 2):
        n = 1 << log_n
        x = synth.rand_field(field, 0xF70000 + log_n, n)
        pre = pa.fft_precompute(field, n)
        exp = ol.FftPrecomputation(field, n).fft_with_precomputation_power_of_2(x, threads=4)
        for rep in range(3):
            fwd = pa.fft_with_precomputation_power_of_2(x, pre)
            bad = np.nonzero((fwd != exp).any(axis=1))[0]
            print(log_n, field, rep, "mism", len(bad), bad[:10].tolist(), bad[-4:].tolist(), flush=True)
