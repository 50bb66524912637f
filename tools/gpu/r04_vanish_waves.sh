#!/bin/bash
# the quotient numerator held to 3 waves per SIMD (ab_libs/libplonky_hip_vw3.so: -DPLK_VANISH_WAVES=3) against the product (2 waves);
# then the parity tests of the comb opt-in and the opening argument back on the bucket method
O=gpurun_out/r4i; mkdir -p $O
for L in plonky_amd/csrc/libplonky_hip.so ab_libs/libplonky_hip_vw3.so; do
  for rep in 1 2; do
  PLK_HIP_LIB=$L timeout 600 python bench.py --workload quotient --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$L vanishing_points_ms %.3f checks %s' % (r['components']['vanishing_points_ms'], all(r['checks'].values())))
"
  done
done | tee $O/vanish_waves.txt
PLK_HIP_LIB=ab_libs/libplonky_hip_vw3.so timeout 600 python -m pytest tests/test_gpu_plonk.py -x -q 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "comb or msm" 2>&1 | tail -2
(timeout 600 python tools/ipa_probe.py 20 14 tabled 2>/dev/null) | head -2
