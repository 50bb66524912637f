#!/bin/bash
mkdir -p gpurun_out/r3g
python tools/gpu/lead_probe.py 20 > gpurun_out/r3g/lead_probe.txt 2>&1
PLK_MSM_WINDOW=19 python tools/gpu/lead_probe.py 20 >> gpurun_out/r3g/lead_probe.txt 2>&1
tail -30 gpurun_out/r3g/lead_probe.txt
