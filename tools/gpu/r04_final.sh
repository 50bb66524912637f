#!/bin/bash
# last pass over the final tree: the device-group tests, the full-size plan on eight virtual devices, then the rocprofv3 passes
# (kernel stats + PMC traffic tied to the final device-source hash)
O=gpurun_out/r4k; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_multi.py tests/test_gpu_fullsize.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 900 python bench.py --gpus 8 --single-process --virtual-devices --steps 8 2>/dev/null | grep "^{" > $O/r04_bench_single_process_virtual8.json; python -c "
import json; r=json.load(open('$O/r04_bench_single_process_virtual8.json')); print(r['checks'], r['components']['group'], r['components']['one_device'])"
bash tools/profile_round.sh r04 > $O/profile_round.log 2>&1; tail -6 $O/profile_round.log | cut -c1-200
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | grep "^{" > $O/bench_check.json; python -c "
import json; r=json.load(open('$O/bench_check.json')); print(r['value'], r['ms_per_step'], r['rooflines']['msm_accumulate']['traffic'], r['rooflines']['ntt_pass']['traffic'], all(r['checks'].values()))"
