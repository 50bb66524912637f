#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 ) > gpurun_out/r2_tfull.log
python tools/gpu/tf_probe.py 20 > gpurun_out/r2_tf_probe.log 2>&1
python tools/gpu/tf_probe.py 18 >> gpurun_out/r2_tf_probe.log 2>&1
python tools/gpu/tf_probe.py 14 >> gpurun_out/r2_tf_probe.log 2>&1
python tools/fuzz_gpu.py 60 > gpurun_out/r2_fuzz_short.log 2>&1
