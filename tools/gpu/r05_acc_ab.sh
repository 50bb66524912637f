#!/bin/bash
# round 5: the GPU suite on the new accumulation kernel, the same-lease A/B against the round-4 library (ab_libs/, not tracked), one bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -25 ) > gpurun_out/r05_suite.log
libs=""
for f in ab_libs/libplonky_hip_*.so; do n=$(basename $f .so); libs="$libs ${n#libplonky_hip_}=$f"; done
timeout 600 python tools/acc_ab.py --reps 20 $libs head=plonky_amd/csrc/libplonky_hip.so > gpurun_out/r05_acc_ab.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_try.json 2> gpurun_out/r05_bench_try.err
tail -12 gpurun_out/r05_suite.log; cat gpurun_out/r05_acc_ab.txt; tail -3 gpurun_out/r05_bench_try.err; cut -c1-1500 gpurun_out/r05_bench_try.json
