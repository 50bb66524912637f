#!/bin/bash
# Round 4, first GPU job: the device group behind the C ABI (virtual devices on the one GPU), the large seeded MSMs against the
# oracle, and the same-lease A/B of the accumulation kernel (round-2 build, round-3 build, HEAD, HEAD without the waves hint).
O=gpurun_out/r4a; mkdir -p $O
(rocm-smi --showuniqueid --showclocks 2>/dev/null | grep -E "Unique|sclk|mclk" | head -6; nproc) > $O/box.txt 2>&1
timeout 600 python tests/multi_device_worker.py 2 14 $O/multi_2_14.npz > $O/multi_worker.log 2>&1; tail -5 $O/multi_worker.log
timeout 1500 python -m pytest tests/test_gpu_multi.py -x -q > $O/pytest_multi.txt 2>&1; tail -15 $O/pytest_multi.txt
timeout 1500 python -m pytest tests/test_gpu_msm_large.py -x -q --durations=5 > $O/pytest_large.txt 2>&1; tail -12 $O/pytest_large.txt
timeout 900 python tools/acc_ab.py --reps 30 --inner 5 r02=ab_libs/libplonky_hip_r02.so r03=ab_libs/libplonky_hip_r03.so head=plonky_amd/csrc/libplonky_hip.so w0=ab_libs/libplonky_hip_w0.so > $O/acc_ab.txt 2>&1; tail -8 $O/acc_ab.txt
