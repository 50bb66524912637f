#!/bin/bash
# the quotient numerator after the algebraic diet of its gate evaluation: parity, the quotient bench twice, per-pass durations
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4k; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_plonk.py -x -q 2>&1 | tail -5 ) > $O/plonk_tests.txt
python bench.py --workload quotient --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_quotient_1.json 2> $O/bench_quotient_1.err
python bench.py --workload quotient --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_quotient_2.json 2> $O/bench_quotient_2.err
bash tools/gpu/r04_vanish_passes.sh > $O/vanish_passes.txt 2>&1
