#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_halo.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 ) > gpurun_out/r2_thalo.log
python tools/ipa_probe.py 20 > gpurun_out/r2_ipa.log 2>&1
python tools/ipa_probe.py 16 >> gpurun_out/r2_ipa.log 2>&1
