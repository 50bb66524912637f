#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_gpu_plonk.py tests/test_gpu_parity.py -x -q -m gpu ) 2>&1 | tail -12 > gpurun_out/r2_tplonk.log
python tools/prover_pipeline_probe.py 20 > gpurun_out/r2_probe20.log 2>&1
python bench.py --workload msm --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2_b_msm.json 2> gpurun_out/r2_b_msm.err
