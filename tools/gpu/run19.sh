#!/bin/bash
mkdir -p gpurun_out/r3o
timeout 500 python tools/fuzz_gpu.py 300 > gpurun_out/r3o/fuzz.log 2>&1; grep -v "amdgpu.ids" gpurun_out/r3o/fuzz.log | tail -12 | cut -c1-250
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
