#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( PLK_MSM_NO_GLV=1 PLK_FOLD_NO_GLV=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_halo.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 ) > gpurun_out/r2_tnoglv.log
( FUZZ_SEED=777 timeout 400 python tools/fuzz_gpu.py 300 2>&1 | tail -2 ) > gpurun_out/r2_fuzz2.log
