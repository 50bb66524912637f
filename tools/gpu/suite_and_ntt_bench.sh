#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -x -q -m gpu ) 2>&1 | grep -E 'passed|failed|error' | tail -5 > gpurun_out/r2_tfull.log
python bench.py --workload ntt --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r2_b_ntt.json 2> gpurun_out/r2_b_ntt.err
