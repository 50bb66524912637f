#!/bin/bash
# round 5: NTT tuning builds (ab_libs/libplonky_hip_ntt<bits>.so: bit 0 batched tile loads, bit 1 batched inter-pass twiddles, bit 2 four waves)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/r05_ntt_variants.txt
for rep in 1 2; do
  for lib in ab_libs/libplonky_hip_nttold.so ab_libs/libplonky_hip_ntt1.so ab_libs/libplonky_hip_ntt2.so ab_libs/libplonky_hip_ntt3.so ab_libs/libplonky_hip_ntt5.so ab_libs/libplonky_hip_ntt6.so plonky_amd/csrc/libplonky_hip.so; do
    echo "== $lib" >> gpurun_out/r05_ntt_variants.txt
    PLK_HIP_LIB=$PWD/$lib timeout 300 python tools/ntt_stagger_probe.py 2>/dev/null >> gpurun_out/r05_ntt_variants.txt
    PLK_HIP_LIB=$PWD/$lib timeout 300 python tools/ntt_probe.py 2>/dev/null | grep -E "batch 9|log_n 23|batch 64" >> gpurun_out/r05_ntt_variants.txt
  done
done
cat gpurun_out/r05_ntt_variants.txt
