#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "msm or fold or commit" 2>&1 | tail -5 > gpurun_out/r2_t1.log
bash tools/gpu/prof_msm.sh 20 w20d
timeout 200 python bench.py --workload msm --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_b_d.json 2> gpurun_out/r2_b_d.err
