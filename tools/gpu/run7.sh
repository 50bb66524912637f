#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
timeout 900 python bench.py --no-cpu-baseline --workload msm > $O/bench_msm.json 2> $O/bench_msm.err; python - <<PY
import json; r=json.load(open("$O/bench_msm.json")); c=r["components"]; print("msm", c["msm_ms"], c["msm_batch9_ms"], c["msm_stage_ms"], r["checks"])
PY
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/$O/prof_ipa -o ipa -- python $R/tools/ipa_probe.py 14 14 > $R/$O/ipa14_profiled.txt 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/prof_ipa -name "*.db" | head -1) > $O/ipa14_kernel_stats.txt 2>&1; head -40 $O/ipa14_kernel_stats.txt; rm -rf $O/prof_ipa
head -3 $O/ipa14_profiled.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "msm or halo or checked or fullsize" > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt
