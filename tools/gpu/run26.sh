#!/bin/bash
mkdir -p gpurun_out/r3v
for ml in 17 16 14; do echo "PLK_HALO_STAGE_MIN_LOG=$ml"; PLK_HALO_STAGE_MIN_LOG=$ml python tools/ipa_probe.py 20 14 tabled 2>&1 | grep -v amdgpu.ids | head -9; done > gpurun_out/r3v/stage_min.txt
cat gpurun_out/r3v/stage_min.txt
