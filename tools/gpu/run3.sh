#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -x -q -m gpu ) 2>&1 | tail -8 > gpurun_out/r2_tfull.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_b_e.json 2> gpurun_out/r2_b_e.err
