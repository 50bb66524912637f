#!/bin/bash
# headline step with the NTT on a second stream (tools/overlap_probe.py)
mkdir -p gpurun_out/r6ov
python tools/overlap_probe.py 200 > gpurun_out/r6ov/overlap.txt 2>&1
cat gpurun_out/r6ov/overlap.txt
