#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt; grep -E "^FAILED|Error" $O/pytest.txt | head
timeout 600 python tools/ipa_probe.py 20 14 > $O/ipa20.txt 2>&1; cat $O/ipa20.txt | head -12
timeout 300 python tools/ipa_probe.py 16 14 > $O/ipa16.txt 2>&1; head -2 $O/ipa16.txt
timeout 400 python tools/fuzz_gpu.py 240 > $O/fuzz.log 2>&1; tail -9 $O/fuzz.log
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/$O/prof_ipa -o ipa -- python $R/tools/ipa_probe.py 14 14 > $R/$O/ipa14_profiled.txt 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/prof_ipa -name "*.db" | head -1) > $O/ipa14_kernel_stats.txt 2>&1; head -24 $O/ipa14_kernel_stats.txt; rm -rf $O/prof_ipa
