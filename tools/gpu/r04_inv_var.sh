#!/bin/bash
# the one-lane (data-dependent) division-step inversion at the end of an MSM: parity, then the MSM bench against the library before it
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4m; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_msm_large.py -x -q 2>&1 | tail -3 ) | tee $O/tests.txt
for rep in 1 2 3; do
for L in plonky_amd/csrc/libplonky_hip.so ab_libs/libplonky_hip_vm.so; do
  PLK_HIP_LIB=$L timeout 600 python bench.py --workload msm --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); c=r['components']; print('$L msm_ms %.4f final %.4f checks %s' % (c['msm_ms'], c['msm_stage_ms'].get('final', -1) if isinstance(c.get('msm_stage_ms'), dict) else -1, all(r['checks'].values())))
"
done
done | tee $O/inv_var.txt
