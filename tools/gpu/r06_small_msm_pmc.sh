#!/bin/bash
# counters of one frozen IPA round's MSM (two vectors over 2^14 + 2 generators): waves, wave cycles, VALU instructions per kernel -
# how much of the GPU each link of the chain occupies (separate --pmc pass, kernel trace only)
O=gpurun_out/r6sm; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PROBE_NO_STAGES=1 PROBE_ITERS=50 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU -d /tmp/smp -o smp -- python $GRAFT_REPO_ROOT/tools/small_msm_probe.py 14 > /tmp/smp.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/smp -name "*.db" | head -1) --pmc > $O/pmc_small_msm_2p14.txt 2>&1
head -60 $O/pmc_small_msm_2p14.txt | cut -c1-220
