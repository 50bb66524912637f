#!/bin/bash
# kernel profiles of (a) an argument of frozen rounds only (2^14) and (b) a 2^20 argument over the caller's tables
export TMPDIR=/tmp; R="${GRAFT_REPO_ROOT:-$PWD}"; O=$R/gpurun_out/r3i; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace --stats -d $O/p14 -o ipa -- python $R/tools/ipa_probe.py 14 14 > $O/ipa14.txt 2>&1
PLK_HALO_LEAD=2 rocprofv3 --kernel-trace --stats -d $O/p20 -o ipa -- python $R/tools/ipa_probe.py 20 14 tabled > $O/ipa20.txt 2>&1
cd $R
db() { find $1 -name "*.db" | head -1; }
python tools/rocpd_summary.py $(db $O/p14) > $O/ipa14_kernels.txt 2>&1
python tools/rocpd_summary.py $(db $O/p20) > $O/ipa20_kernels.txt 2>&1
rm -rf $O/p14 $O/p20
cut -c1-150 $O/ipa14_kernels.txt | head -24; cut -c1-150 $O/ipa20_kernels.txt | head -40
