#!/bin/bash
export TMPDIR=/tmp; R="${GRAFT_REPO_ROOT:-$PWD}"; O=$R/gpurun_out/r3t; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace --stats -d $O/p14 -o ipa -- python $R/tools/ipa_probe.py 14 14 > $O/ipa14.txt 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/p14 -name "*.db" | head -1) > $O/ipa14_kernels.txt 2>&1
rm -rf $O/p14
cut -c1-150 $O/ipa14_kernels.txt | head -12
