#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/r2_bench_default.json 2> gpurun_out/r2_bench_default.err
python bench.py --workload commit9 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2_bench_commit9.json 2> gpurun_out/r2_bench_commit9.err
python bench.py --workload msm --curve bls12_377 --log-n 22 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2_bench_bls22.json 2> gpurun_out/r2_bench_bls22.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --workload commit9 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_commit9_trun.json 2> gpurun_out/r2_bench_commit9_trun.err
