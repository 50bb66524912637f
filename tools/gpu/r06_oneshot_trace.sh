#!/bin/bash
# kernel trace of the one-shot (table-free) msm_parallel at 2^20 on the final tree
O=gpurun_out/r6os; mkdir -p $O
python tools/oneshot_probe.py 20 20 > $O/oneshot.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/os -o os -- python $GRAFT_REPO_ROOT/tools/oneshot_probe.py 20 10 > /tmp/os.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/os -name "*.db" | head -1) > $O/kernel_stats_oneshot.txt 2>&1
cat $O/oneshot.txt; head -30 $O/kernel_stats_oneshot.txt | cut -c1-130
