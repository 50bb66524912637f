#!/bin/bash
# a longer randomised parity run on the final tree (other seeds than r06_fuzz.sh), plus the ordering fuzz with a longer budget
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( rocm-smi --showuniqueid 2>/dev/null | grep "^GPU\["; echo "# FUZZ_GEOMETRY=1 FUZZ_SEED=20261111 python tools/fuzz_gpu.py 900 (the final tree)"; FUZZ_GEOMETRY=1 FUZZ_SEED=20261111 timeout 1300 python tools/fuzz_gpu.py 900 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-400 ) > gpurun_out/r06_fuzz_long.log
( echo "# FUZZ_BUDGET=600 python tools/fuzz_order_gpu.py (the final tree)"; FUZZ_BUDGET=600 FUZZ_SEED=7171 timeout 900 python tools/fuzz_order_gpu.py 600 2>&1 | grep -v amdgpu.ids | tail -4 ) > gpurun_out/r06_fuzz_order_long.log
tail -9 gpurun_out/r06_fuzz_long.log | cut -c1-200; tail -3 gpurun_out/r06_fuzz_order_long.log
