#!/bin/bash
mkdir -p gpurun_out/r3n
timeout 900 python -m pytest tests/test_gpu_halo.py -x -q 2>&1 | tail -5 > gpurun_out/r3n/pytest_halo.txt; cat gpurun_out/r3n/pytest_halo.txt
python bench.py --workload quotient --steps 4 --warmup 1 > gpurun_out/r3n/bench_quotient.json 2> gpurun_out/r3n/bench_quotient.err; tail -3 gpurun_out/r3n/bench_quotient.err
python -c "
import json; r=json.load(open('gpurun_out/r3n/bench_quotient.json')); print(r['components']); print(r['checks']); print({k:(round(v['achieved'],1), round(v['frac'],3) if v['frac'] else None) for k,v in r['rooflines'].items()})"
timeout 400 python tools/fuzz_gpu.py 240 > gpurun_out/r3n/fuzz.log 2>&1; tail -12 gpurun_out/r3n/fuzz.log
