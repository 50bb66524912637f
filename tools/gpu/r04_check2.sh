#!/bin/bash
# after the accumulation's boundary handling changed (no register clears) and the slab knob of the vanishing points: suite subset,
# A/B against round 3, the quotient bench with and without slabs
O=gpurun_out/r4f; mkdir -p $O
(rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID:" | head -1) > $O/box.txt
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plonk.py tests/test_gpu_checked.py tests/test_gpu_msm_large.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 900 python tools/acc_ab.py --reps 20 --inner 5 r03=ab_libs/libplonky_hip_r03.so head=plonky_amd/csrc/libplonky_hip.so 2>&1 | grep -v amdgpu.ids > $O/acc_ab.txt; tail -3 $O/acc_ab.txt
for S in 0 16 17 18 19; do
  PLK_VANISH_SLAB_LOG=$S timeout 600 python bench.py --workload quotient --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('slab_log $S vanishing_points_ms %.3f frac %.3f checks %s' % (r['components']['vanishing_points_ms'], r['rooflines']['vanishing_points']['frac'] or 0, all(r['checks'].values())))
"
done | tee $O/vanish_slabs.txt
