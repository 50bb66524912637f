#!/bin/bash
# round 5: wave-owned columns in k_ntt_pass (PLK_NTT_VARIANT bit 4: column-major tile in LDS, every wave runs the stages of its own columns,
# wave_sync() instead of workgroup barriers between the steps; bit 5: the global phases by wave as well, no workgroup barrier at all)
# ab_libs/libplonky_hip_nttv{5,21,53}.so = the same objects with ntt.hip at VARIANT 5 (the product form) / 21 / 53
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
O=gpurun_out/r05_ntt_wave.txt
: > $O
for v in 53 21; do
  echo "== parity, variant $v" >> $O
  ( timeout 500 env PLK_HIP_LIB=$PWD/ab_libs/libplonky_hip_nttv$v.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_poly.py -x -q -m gpu -k "ntt or fft or poly or lde or divide" 2>&1 | tail -3 ) >> $O
done
for rep in 1 2 3; do
  for v in 5 21 53; do
    echo "== variant $v" >> $O
    PLK_HIP_LIB=$PWD/ab_libs/libplonky_hip_nttv$v.so timeout 300 python tools/ntt_stagger_probe.py 2>/dev/null >> $O
    PLK_HIP_LIB=$PWD/ab_libs/libplonky_hip_nttv$v.so timeout 300 python tools/ntt_probe.py 2>/dev/null | grep -E "batch 9|log_n 23|batch 64" >> $O
  done
done
cat $O
