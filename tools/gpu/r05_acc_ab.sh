#!/bin/bash
# round 5: the GPU suite, the same-lease A/B of the accumulation against the libraries in ab_libs/ (not tracked), the crossover sweep
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r05_suite.log
libs=""
for f in ab_libs/libplonky_hip_*.so; do n=$(basename $f .so); libs="$libs ${n#libplonky_hip_}=$f"; done
timeout 900 python tools/acc_ab.py --reps 16 $libs head=plonky_amd/csrc/libplonky_hip.so > gpurun_out/r05_acc_ab.txt 2>&1
timeout 1500 python bench.py --workload crossover --log-n 20 > gpurun_out/r05_crossover.json 2> gpurun_out/r05_crossover.txt
tail -4 gpurun_out/r05_suite.log; cat gpurun_out/r05_acc_ab.txt; cat gpurun_out/r05_crossover.txt | tail -20
