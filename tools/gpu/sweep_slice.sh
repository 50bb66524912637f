#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for sl in 18 24 30 36 40 72; do
  PLK_MSM_WINDOW=20 PLK_MSM_SLICE=$sl timeout 200 python bench.py --workload msm --steps 10 --warmup 3 --no-cpu-baseline --timed-only > gpurun_out/r2_sl$sl.json 2> gpurun_out/r2_sl$sl.err
done
