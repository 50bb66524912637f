#!/bin/bash
# closing pass over the final tree on ONE lease: the quotient-numerator and bench-form tests, the rocprofv3 passes (kernel stats + PMC
# traffic, tied to this tree's device-source hash), then the two bench lines with those PMC summaries in place
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4r; mkdir -p $O
(rocm-smi --showuniqueid 2>/dev/null | grep "^GPU\[") > $O/box.txt
( timeout 1500 python -m pytest tests/test_gpu_plonk.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 ) | tee $O/tests.txt
bash tools/profile_round.sh r04 > $O/profile_round.log 2>&1; tail -4 $O/profile_round.log | cut -c1-200
cp gpurun_out/r04_pmc_traffic.json gpurun_out/r04_pmc_traffic_quotient.json profiles/
timeout 1500 python bench.py > $O/r04_bench.json 2> $O/r04_bench.err
timeout 600 python bench.py --workload quotient --log-n 20 --steps 6 --warmup 2 > $O/r04_bench_quotient.json 2> $O/bench_quotient.err
python - <<PY
import json
r=json.load(open("$O/r04_bench.json")); print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['traffic'], all(r['checks'].values()))
q=json.load(open("$O/r04_bench_quotient.json")); print(q['components']['vanishing_points_ms'], q['rooflines']['vanishing_points']['traffic'])
PY
