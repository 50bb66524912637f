#!/bin/bash
mkdir -p gpurun_out/r3u
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3u/bench.json 2> gpurun_out/r3u/bench.err; python -c "
import json; r=json.load(open('gpurun_out/r3u/bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['frac']); c=r['components']; print({k:c[k] for k in ('ntt_ms','msm_ms','msm_batch9_ms','msm_parallel_one_shot_ms','msm_stage_ms')})"
python tools/ipa_probe.py 20 14 tabled 2>&1 | grep -v amdgpu.ids | head -12
python bench.py --workload quotient --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['components'])"
