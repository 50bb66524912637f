#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python tools/gpu/tf_probe.py 20 > gpurun_out/r2_tf_probe.log 2>&1
python tools/gpu/tf_probe.py 19 >> gpurun_out/r2_tf_probe.log 2>&1
