#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -x -q -m gpu ) 2>&1 | tail -12 > gpurun_out/r2_tfull.log
python tools/prover_pipeline_probe.py 17 > gpurun_out/r2_probe17.log 2>&1; python tools/prover_pipeline_probe.py 20 > gpurun_out/r2_probe20.log 2>&1
