#!/bin/bash
# round 6: the whole GPU suite, the default bench line, and the same-lease A/B of k_msm_final_pair against k_msm_final (PLK_MSM_FINAL_V1=1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -16 ) > gpurun_out/r06_suite.log
tail -14 gpurun_out/r06_suite.log
for rep in 1 2; do
  for v in pair fin1; do
    unset PLK_MSM_FINAL_V1; [ $v = fin1 ] && export PLK_MSM_FINAL_V1=1
    timeout 600 python bench.py --workload msm --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r06_final_${v}_${rep}.json 2> gpurun_out/r06_final_${v}_${rep}.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r06_final_${v}_${rep}.json"))
c=d["components"]
print("${v} ${rep}", "msm_ms %.4f" % c.get("msm_ms"), "batch9 %.3f" % c.get("msm_batch9_ms"), "stages", c.get("msm_stage_ms"), all(d.get("checks").values()))
PY
  done
done
unset PLK_MSM_FINAL_V1
timeout 900 python bench.py > gpurun_out/r06_bench_first.json 2> gpurun_out/r06_bench_first.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_bench_first.json"))
print("value", d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "checks", d["checks"])
c=d["components"]
for k in ("ntt_ms","ntt_batch9_melems_per_s","msm_ms","msm_batch9_mpairs_per_s","msm_parallel_one_shot_ms","ipa_ms","lde9_ms","divide_by_z_h_ms","msm_stage_ms"):
    print(" ", k, c.get(k))
print(" roofline", json.dumps(d["roofline"])[:600])
PY
