#!/bin/bash
# kernel profile of the frozen rounds of the opening argument (every round over 2^14 frozen generators): where a 0.46 ms round goes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r4o; mkdir -p $O
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf $O/prof
rocprofv3 --kernel-trace -d $O/prof -o f -- python $R/tools/ipa_probe.py 14 14 tabled > $O/probe.txt 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/prof -name "*.db" | head -1) > $O/r04_ipa_frozen_round_kernels.txt 2>&1
head -40 $O/r04_ipa_frozen_round_kernels.txt | cut -c1-160
rm -rf $O/prof
