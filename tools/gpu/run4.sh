#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_plonk.py -x -q ) 2>&1 | tail -30 > gpurun_out/r2_tplonk.log
