#!/bin/bash
# after the raw-piece store change: parity subset, same-lease A/B, then the fuzz tool over a device group of three and on one device
O=gpurun_out/r4g; mkdir -p $O
(rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID:" | head -1) > $O/box.txt
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_checked.py tests/test_gpu_msm_large.py tests/test_gpu_multi.py tests/test_gpu_halo.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 900 python tools/acc_ab.py --reps 20 --inner 5 r03=ab_libs/libplonky_hip_r03.so head=plonky_amd/csrc/libplonky_hip.so 2>&1 | grep -v amdgpu.ids > $O/acc_ab.txt; tail -3 $O/acc_ab.txt
FUZZ_DEVICES=3 timeout 600 python tools/fuzz_gpu.py 150 2>&1 | grep -v amdgpu.ids | tail -12 > $O/fuzz_group3.txt; head -3 $O/fuzz_group3.txt | cut -c1-200
FUZZ_SEED=424242 timeout 600 python tools/fuzz_gpu.py 150 2>&1 | grep -v amdgpu.ids | tail -12 > $O/fuzz_one.txt; head -3 $O/fuzz_one.txt | cut -c1-200
