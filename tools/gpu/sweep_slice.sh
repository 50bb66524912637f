#!/bin/bash
# tuning sweeps of the tabled 2^20 MSM: entries per accumulation lane (PLK_MSM_SLICE; default = one round of lanes, 70-71) and the
# group size of the row / column sums (PLK_MSM_GLOG; default 3)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/r2_sweeps.log
run() { # tag env
  env "$2" timeout 200 python bench.py --workload msm --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r2_sw_$1.json 2> gpurun_out/r2_sw_$1.err
  python - <<PY >> gpurun_out/r2_sweeps.log
import json
try:
    d = json.load(open("gpurun_out/r2_sw_$1.json")); c = d["components"]; s = c["msm_stage_ms"]
    print("%-10s msm %.4f ms  batch9 %.0f M/s  accumulate %.4f  assemble+lines %.4f  checks %s" % ("$1", c["msm_ms"], c["msm_batch9_mpairs_per_s"], s["accumulate"], s["assemble_lines"], all(d["checks"].values())))
except Exception as e:
    print("$1 FAILED", open("gpurun_out/r2_sw_$1.err").read()[-200:])
PY
}
run default X=1
for sl in 48 60 66 76 84 96; do run slice$sl PLK_MSM_SLICE=$sl; done
for g in 2 4; do run glog$g PLK_MSM_GLOG=$g; done
