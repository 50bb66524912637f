#!/bin/bash
mkdir -p gpurun_out/r3h
timeout 900 python -m pytest tests/test_gpu_halo.py -x -q 2>&1 | tail -15 > gpurun_out/r3h/pytest_halo.txt
cat gpurun_out/r3h/pytest_halo.txt
for st in 2 0 3; do PLK_HALO_STAGE=$st python tools/ipa_probe.py 20 14 tabled 2>&1 | grep -v amdgpu.ids | head -30; done > gpurun_out/r3h/ipa_lead.txt
cat gpurun_out/r3h/ipa_lead.txt | grep "IPA at\|round at length  *[0-9]\{5,7\}:"
