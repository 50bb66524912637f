#!/bin/bash
# round 5, final tree: entries per accumulation lane (PLK_MSM_SLICE; default 70 at 2^20 = one round of lanes) and the group size of the row / column
# sums (PLK_MSM_GLOG; default 3) once more, one process per setting, one lease
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
O=gpurun_out/r05_slice_glog.txt
: > $O
run() {
  env "$@" python bench.py --workload msm --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['components']
print('$*'.ljust(28), 'ms_per_step %.4f' % d['ms_per_step'], 'stages', c.get('msm_stage_ms'))" >> $O
}
for rep in 1 2; do
run PLK_X=default
for s in 35 48 60 66 76 84 96; do run PLK_MSM_SLICE=$s; done
for g in 2 4; do run PLK_MSM_GLOG=$g; done
done
cat $O
