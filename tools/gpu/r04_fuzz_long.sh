#!/bin/bash
# a longer run of the randomised parity stress with fresh seeds (one device 900 s, a device group of two 500 s)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4s; mkdir -p $O
( rocm-smi --showuniqueid 2>/dev/null | grep "^GPU\["; echo "# python tools/fuzz_gpu.py 900 (FUZZ_SEED=77001)"; FUZZ_SEED=77001 timeout 1200 python tools/fuzz_gpu.py 900 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tail -12 ) > $O/r04_fuzz_long.log
( echo "# FUZZ_DEVICES=2 python tools/fuzz_gpu.py 500 (FUZZ_SEED=77002)"; FUZZ_DEVICES=2 FUZZ_SEED=77002 timeout 800 python tools/fuzz_gpu.py 500 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tail -12 ) >> $O/r04_fuzz_long.log
grep "fuzz" $O/r04_fuzz_long.log
