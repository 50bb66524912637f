#!/bin/bash
mkdir -p gpurun_out/r3q
timeout 900 python -m pytest tests/test_gpu_halo.py -x -q 2>&1 | tail -3
timeout 1200 python bench.py > gpurun_out/r3q/r03_bench.json 2> gpurun_out/r3q/bench.err; python -c "
import json; r=json.load(open('gpurun_out/r3q/r03_bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['traffic'], r['config'].get('kernel_source_sha'))"
timeout 600 python bench.py --workload quotient --log-n 20 --steps 6 --warmup 2 > gpurun_out/r3q/r03_bench_quotient.json 2> gpurun_out/r3q/bench_quotient.err; python -c "
import json; q=json.load(open('gpurun_out/r3q/r03_bench_quotient.json')); print({k:(round(v['achieved'],1), round(v['frac'],3), v['traffic']) for k,v in q['rooflines'].items()})"
