#!/bin/bash
# round 5: the GPU suite on the ordering kernels with batched loads; MSM stage A/B against round 4; NTT old (loads one by one) against new (batched)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r05_suite.log
timeout 900 python tools/acc_ab.py --reps 14 r04=ab_libs/libplonky_hip_r04.so head=plonky_amd/csrc/libplonky_hip.so > gpurun_out/r05_msm_ab.txt 2>&1
: > gpurun_out/r05_ntt_ab.txt
for rep in 1 2; do
  for lib in ab_libs/libplonky_hip_nttold.so plonky_amd/csrc/libplonky_hip.so; do
    echo "== $lib" >> gpurun_out/r05_ntt_ab.txt
    PLK_HIP_LIB=$PWD/$lib timeout 300 python tools/ntt_stagger_probe.py 2>/dev/null >> gpurun_out/r05_ntt_ab.txt
    PLK_HIP_LIB=$PWD/$lib timeout 300 python tools/ntt_probe.py 2>/dev/null >> gpurun_out/r05_ntt_ab.txt
  done
done
tail -3 gpurun_out/r05_suite.log; cat gpurun_out/r05_msm_ab.txt; cat gpurun_out/r05_ntt_ab.txt
