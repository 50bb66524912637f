#!/bin/bash
# round-2 first pass: MSM parity + window sweep
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "msm or fold or commit" 2>&1 | tail -25 > gpurun_out/r2_t1.log
for w in 16 18 19 20; do
  PLK_MSM_WINDOW=$w timeout 200 python bench.py --workload msm --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_b_w$w.json 2> gpurun_out/r2_b_w$w.err
done
echo done
