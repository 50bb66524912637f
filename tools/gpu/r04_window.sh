#!/bin/bash
# the window sweep redone on the round-4 tree (the accumulation got 6 % cheaper, the bucket reduction did not), then the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
for rep in 1 2; do
  for w in 18 19 20 21; do
    PLK_MSM_WINDOW=$w PROBE_ITERS=40 python tools/msm_probe.py 0:20 2>&1 | grep "log_n"
  done
done
} > gpurun_out/r04_window_sweep.txt 2>&1
./ab_libs/tail_lab > gpurun_out/r04_tail_lab.txt 2>&1
python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err
