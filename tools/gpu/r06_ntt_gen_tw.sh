#!/bin/bash
# inter-pass twiddles formed from a two-level geometric table (PLK_NTT_GEN_TW=1: one more product per element, no 36 MiB table stream)
# against the streamed outer table: parity slice first, then the probe, alternating, one lease
O=gpurun_out/r6gt; mkdir -p $O
PLK_NTT_GEN_TW=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_poly.py -x -q -m gpu -k "test_fft_and_ifft or test_ntt_matches_oracle or test_ntt_linearity or test_ntt_padding or test_polynomials_to_values_padded or test_ntt_2p20_full_size" > $O/parity.txt 2>&1; tail -2 $O/parity.txt
(for rep in 1 2; do echo "# streamed outer table (default)"; python tools/ntt_probe.py 2>/dev/null; echo "# PLK_NTT_GEN_TW=1"; PLK_NTT_GEN_TW=1 python tools/ntt_probe.py 2>/dev/null; done) > $O/ntt_gen_tw.txt 2>&1
cat $O/ntt_gen_tw.txt
