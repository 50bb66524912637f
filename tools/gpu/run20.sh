#!/bin/bash
mkdir -p gpurun_out/r3p
python tools/prover_pipeline_probe.py 20 ipa 2>&1 | grep -v amdgpu.ids > gpurun_out/r3p/pipeline_ipa.txt; cat gpurun_out/r3p/pipeline_ipa.txt
python tools/prover_pipeline_probe.py 17 ipa 2>&1 | grep -v amdgpu.ids | tail -4
