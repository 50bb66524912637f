#!/bin/bash
mkdir -p gpurun_out/r3k
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3k/pytest.txt
cat gpurun_out/r3k/pytest.txt
PLK_HALO_LEAD=2 python tools/ipa_probe.py 20 14 tabled 2>&1 | grep -v amdgpu.ids | head -30 > gpurun_out/r3k/ipa.txt
PLK_MSM_NO_FORK=1 PLK_HALO_LEAD=2 python tools/ipa_probe.py 20 14 tabled 2>&1 | grep -v amdgpu.ids | head -3 >> gpurun_out/r3k/ipa.txt
cat gpurun_out/r3k/ipa.txt
