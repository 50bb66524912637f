#!/bin/bash
# round 6: k_ord_bin_sort with the records of pass 1 kept in registers against the re-reading form (PLK_MSM_BINSORT_NOCACHE=1), same lease
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_msm_order.py tests/test_gpu_checked.py -x -q -m gpu 2>&1 | tail -4 ) > gpurun_out/r06_t5.log; tail -2 gpurun_out/r06_t5.log
for rep in 1 2; do
  for v in cache nocache; do
    unset PLK_MSM_BINSORT_NOCACHE; [ $v = nocache ] && export PLK_MSM_BINSORT_NOCACHE=1
    timeout 600 python bench.py --workload msm --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r06_binsort_${v}_${rep}.json 2> /dev/null
    python - <<PY
import json
d=json.load(open("gpurun_out/r06_binsort_${v}_${rep}.json")); c=d["components"]
print("${v} ${rep}", "msm_ms %.4f" % c["msm_ms"], "batch9 %.3f" % c["msm_batch9_ms"], c["msm_stage_ms"], all(d["checks"].values()))
PY
  done
done
