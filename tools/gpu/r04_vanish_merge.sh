#!/bin/bash
# the two Rescue gates in one launch (ab_libs/libplonky_hip_vm.so: -DPLK_VANISH_MERGE_RESCUE=1) against the product (five launches)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4l; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_plonk.py -x -q 2>&1 | tail -3 ) | tee $O/plonk_tests.txt
for rep in 1 2; do
for L in plonky_amd/csrc/libplonky_hip.so ab_libs/libplonky_hip_vm.so; do
  PLK_HIP_LIB=$L timeout 600 python bench.py --workload quotient --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$L vanishing_points_ms %.3f checks %s' % (r['components']['vanishing_points_ms'], all(r['checks'].values())))
"
done
done | tee $O/vanish_merge.txt
PLK_HIP_LIB=ab_libs/libplonky_hip_vm.so timeout 600 python -m pytest tests/test_gpu_plonk.py -x -q 2>&1 | tail -2 | tee -a $O/vanish_merge.txt
