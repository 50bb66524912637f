#!/bin/bash
# the two bench lines again once profiles/ holds the PMC summaries of this tree (bench.py ties `traffic` to the device-source hash)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4n; mkdir -p $O
(rocm-smi --showuniqueid 2>/dev/null | grep "^GPU\[") > $O/box.txt
( timeout 1500 python -m pytest tests/test_gpu_plonk.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 ) > $O/tests.txt
timeout 1500 python bench.py > $O/r04_bench.json 2> $O/r04_bench.err
timeout 600 python bench.py --workload quotient --log-n 20 --steps 6 --warmup 2 > $O/r04_bench_quotient.json 2> $O/bench_quotient.err
