#!/bin/bash
# full GPU evidence of the current tree: suite, profiles (kernel stats + PMC traffic), fuzz, the default bench line, probes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 ) > gpurun_out/r2_tfull.log
bash tools/gpu/profile_and_fuzz.sh
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
python tools/prover_pipeline_probe.py 17 > gpurun_out/r2_probe17.log 2>&1
python tools/prover_pipeline_probe.py 20 > gpurun_out/r2_probe20.log 2>&1
python bench.py --workload commit9 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_commit9.json 2> gpurun_out/r02_bench_commit9.err
python bench.py --workload msm --curve bls12_377 --log-n 22 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_bls.json 2> gpurun_out/r02_bench_bls.err
python tools/fold_probe.py > gpurun_out/r2_fold.log 2>&1
python tools/gpu/tf_probe.py 20 > gpurun_out/r2_tf_probe.log 2>&1
python tools/gpu/tf_probe.py 16 >> gpurun_out/r2_tf_probe.log 2>&1
