#!/bin/bash
# Round 4: the bench line in its new form (headline region without per-kernel events, steady-state component loops, GPU identity,
# the opening argument as a component), the two-rank form with the strong-scaling components, the single-process device group.
O=gpurun_out/r4c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q --durations=8 > $O/pytest_fullsize.txt 2>&1; tail -14 $O/pytest_fullsize.txt
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python - <<PY
import json
r = json.load(open("$O/bench.json"))
c = r["components"]
print("value", r["value"], "ms/step", r["ms_per_step"], "profiled", r["ms_per_step_profiled"], r["checks"])
print({k: c[k] for k in c if k.endswith("_ms") or k.endswith("per_s")}); print(c["msm_stage_ms"]); print(r["config"]["gpu"]); print(c["host_pointer"])
PY
timeout 900 python bench.py --gpus 2 --single-process --virtual-devices --steps 20 > $O/bench_single_process_virtual2.json 2> $O/bench_sp.err; tail -2 $O/bench_sp.err; cat $O/bench_single_process_virtual2.json | cut -c1-1500
timeout 900 python bench.py --gpus 2 --same-device --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_2ranks_same_device.json 2> $O/bench_2r.err; tail -2 $O/bench_2r.err; python - <<PY
import json
r = json.load(open("$O/bench_2ranks_same_device.json")); print(r["components"]["multi_gpu"]); print(r["checks"])
PY
