#!/bin/bash
# the inner-product argument after (a) window 16 at 2^14 generators (b) L_j / R_j normalised on the host: tests, then A/B on one lease
O=gpurun_out/r6ipa; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_halo.py -x -q -m gpu > $O/halo_tests.txt 2>&1; tail -3 $O/halo_tests.txt
F='^IPA at n = 2\^20 over|round at length +(1048576|16384|2):'
(for rep in 1 2; do
 echo "# default (window 16 at 2^14, host to_affine of L / R)"; timeout 300 python tools/ipa_probe.py 20 14 tabled 2>/dev/null | grep -E "$F"
 echo "# PLK_MSM_WINDOW_2P14=13"; PLK_MSM_WINDOW_2P14=13 timeout 300 python tools/ipa_probe.py 20 14 tabled 2>/dev/null | grep -E "$F"
 echo "# PLK_HALO_DEVICE_AFFINE=1"; PLK_HALO_DEVICE_AFFINE=1 timeout 300 python tools/ipa_probe.py 20 14 tabled 2>/dev/null | grep -E "$F"
 echo "# PLK_HALO_DEVICE_AFFINE=1 PLK_MSM_WINDOW_2P14=13 (round 6 before this change)"; PLK_HALO_DEVICE_AFFINE=1 PLK_MSM_WINDOW_2P14=13 timeout 300 python tools/ipa_probe.py 20 14 tabled 2>/dev/null | grep -E "$F"
done) > $O/ipa_ab.txt 2>&1
cat $O/ipa_ab.txt
