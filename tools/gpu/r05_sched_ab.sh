#!/bin/bash
# round 5: the backend's scheduling strategy for the two hot translation units (msm_acc.hip, ntt.hip): -mllvm -amdgpu-sched-strategy=max-ilp /
# iterative-ilp (accumulation only: the compiler crashes on ntt.hip) / max-memory-clause, -amdgpu-schedule-metric-bias=0; the other objects unchanged
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
O=gpurun_out/r05_sched_ab.txt
: > $O
timeout 600 python tools/acc_ab.py --reps 20 product=plonky_amd/csrc/libplonky_hip.so maxilp=ab_libs/libplonky_hip_maxilp.so iterilp=ab_libs/libplonky_hip_iterilp.so bias0=ab_libs/libplonky_hip_bias0.so maxmem=ab_libs/libplonky_hip_maxmem.so >> $O 2>&1
for rep in 1 2; do
  for v in "" _maxilp _bias0 _maxmem; do
    lib=$PWD/ab_libs/libplonky_hip$v.so; [ -z "$v" ] && lib=$PWD/plonky_amd/csrc/libplonky_hip.so
    echo "== ntt ${v:-product}" >> $O
    PLK_HIP_LIB=$lib timeout 300 python tools/ntt_stagger_probe.py 2>/dev/null >> $O
    PLK_HIP_LIB=$lib timeout 300 python tools/ntt_probe.py 2>/dev/null | grep -E "batch 9|log_n 23" >> $O
  done
done
cat $O
