#!/bin/bash
# round 5: NTT tuning builds (ab_libs/libplonky_hip_ntt<bits>.so: bit 0 batched tile loads, bit 1 batched inter-pass twiddles, bit 2 four waves,
# bit 3 bank skew of the tile in LDS) against the product library, alternating, one process per measurement
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/r05_ntt_variants2.txt
( timeout 600 env PLK_HIP_LIB=$PWD/ab_libs/libplonky_hip_ntt13.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_poly.py -x -q -m gpu -k "ntt or fft or poly or lde or divide" 2>&1 | tail -3 ) >> gpurun_out/r05_ntt_variants2.txt
for rep in 1 2 3; do
  for lib in ab_libs/libplonky_hip_ntt*.so plonky_amd/csrc/libplonky_hip.so; do
    echo "== $lib" >> gpurun_out/r05_ntt_variants2.txt
    PLK_HIP_LIB=$PWD/$lib timeout 300 python tools/ntt_stagger_probe.py 2>/dev/null >> gpurun_out/r05_ntt_variants2.txt
    PLK_HIP_LIB=$PWD/$lib timeout 300 python tools/ntt_probe.py 2>/dev/null | grep -E "batch 9|log_n 23|batch 64" >> gpurun_out/r05_ntt_variants2.txt
  done
done
cat gpurun_out/r05_ntt_variants2.txt
