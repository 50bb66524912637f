#!/bin/bash
# the whole GPU suite, the same-lease A/B of the accumulation against the round-3 build, the default bench line
O=gpurun_out/r4e; mkdir -p $O
(rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID:" | head -1) > $O/box.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
timeout 900 python tools/acc_ab.py --reps 20 --inner 5 r03=ab_libs/libplonky_hip_r03.so head=plonky_amd/csrc/libplonky_hip.so 2>&1 | grep -v amdgpu.ids > $O/acc_ab.txt; tail -4 $O/acc_ab.txt
timeout 1200 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<PY
import json
r = json.load(open("$O/bench.json")); c = r["components"]
print("value", r["value"], "ms/step", r["ms_per_step"], "profiled", r["ms_per_step_profiled"], all(r["checks"].values()))
print({k: round(c[k], 4) for k in c if k.endswith("_ms") or k.endswith("per_s")}); print(c["msm_stage_ms"]); print(r["config"]["gpu"]); print(r["rooflines"]["msm_accumulate"]["frac"], r["rooflines"]["ntt_pass"]["frac"])
PY
