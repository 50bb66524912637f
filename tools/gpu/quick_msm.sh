#!/bin/bash
# quick check after an MSM kernel change: the GPU suite and the MSM-only bench, twice
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 ) > gpurun_out/r2_tfull.log
python bench.py --workload msm --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/r2_b_msm.json 2> gpurun_out/r2_b_msm.err
python bench.py --workload msm --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/r2_b_msm2.json 2> gpurun_out/r2_b_msm2.err
