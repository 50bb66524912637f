#!/bin/bash
# kernel-trace of the MSM at a given window: tools/gpu/prof_msm.sh <window> <tag>
W=${1:-20}; TAG=${2:-msm}
export TMPDIR=/tmp; R="${GRAFT_REPO_ROOT:-/root/repo}"; cd /tmp
rm -rf $R/gpurun_out/prof_$TAG
PLK_MSM_WINDOW=$W timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --workload msm --steps 10 --warmup 2 --no-cpu-baseline --timed-only > $R/gpurun_out/prof_$TAG.json 2> $R/gpurun_out/prof_$TAG.err
cd $R
F=$(find gpurun_out/prof_$TAG -name "*.db" | head -1); python tools/rocpd_summary.py $F > gpurun_out/r02_kstats_$TAG.txt 2>&1
rm -rf gpurun_out/prof_$TAG
