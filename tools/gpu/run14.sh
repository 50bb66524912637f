#!/bin/bash
mkdir -p gpurun_out/r3j
for hu in 0 1; do
  if [ $hu = 1 ]; then export PLK_HALO_DEBUG_NO_HU=1; fi
  PLK_HALO_LEAD=2 python tools/ipa_probe.py 20 14 tabled 2>&1 | grep -v amdgpu.ids | head -12
done > gpurun_out/r3j/no_hu.txt
cat gpurun_out/r3j/no_hu.txt
