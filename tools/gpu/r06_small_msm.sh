#!/bin/bash
# the fixed cost of a small tabled two-vector MSM (the IPA's frozen rounds): stages + kernel trace at 2^14 + 2
mkdir -p gpurun_out/r6sm
python tools/small_msm_probe.py 10 12 14 16 > gpurun_out/r6sm/small_msm.txt 2>&1
cd /tmp && export TMPDIR=/tmp
PROBE_NO_STAGES=1 PROBE_ITERS=50 rocprofv3 --kernel-trace --stats -d /tmp/sm -o sm -- python $GRAFT_REPO_ROOT/tools/small_msm_probe.py 14 > /tmp/sm.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/sm -name "*.db" | head -1) > gpurun_out/r6sm/kernel_stats_2p14.txt 2>&1
cat gpurun_out/r6sm/small_msm.txt; head -40 gpurun_out/r6sm/kernel_stats_2p14.*
