#!/bin/bash
mkdir -p gpurun_out/r3l
for w in 0 9 10 11 12 13 14; do
  if [ $w = 0 ]; then unset PLK_MSM_WINDOW; else export PLK_MSM_WINDOW=$w; fi
  echo "PLK_MSM_WINDOW=$w"; python tools/ipa_probe.py 14 14 2>&1 | grep -v amdgpu.ids | head -4
done > gpurun_out/r3l/frozen_window.txt
unset PLK_MSM_WINDOW
PLK_HALO_STAGE_MIN_LOG=1 python -m pytest tests/test_gpu_halo.py -x -q 2>&1 | tail -3 >> gpurun_out/r3l/frozen_window.txt
python tools/ipa_probe.py 20 14 tabled 2>&1 | grep -v amdgpu.ids | head -3 >> gpurun_out/r3l/frozen_window.txt
cat gpurun_out/r3l/frozen_window.txt
