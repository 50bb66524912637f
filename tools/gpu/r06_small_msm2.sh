#!/bin/bash
# window sweep of the small tabled two-vector MSM (2^14 + 2, 2^12 + 2 generators) on the round-6 reduction
mkdir -p gpurun_out/r6sm
PROBE_WINDOWS=11,12,13,14,15,16,17,18 python tools/small_msm_probe.py 14 > gpurun_out/r6sm/window_sweep.txt 2>&1
PROBE_WINDOWS=6,7,8,9,10,11,12,13,14 python tools/small_msm_probe.py 12 >> gpurun_out/r6sm/window_sweep.txt 2>&1
PLK_MSM_SLICE=16 PROBE_WINDOWS=13,16 python tools/small_msm_probe.py 14 >> gpurun_out/r6sm/window_sweep.txt 2>&1
cat gpurun_out/r6sm/window_sweep.txt
