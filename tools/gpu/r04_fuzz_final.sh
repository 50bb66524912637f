#!/bin/bash
# the randomised parity stress on the final tree: one device, then a device group of three virtual devices
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4p; mkdir -p $O
( rocm-smi --showuniqueid 2>/dev/null | grep "^GPU\["; echo "# python tools/fuzz_gpu.py 600 (FUZZ_SEED=20260929)"; FUZZ_SEED=20260929 timeout 900 python tools/fuzz_gpu.py 600 2>&1 | tail -25 ) > $O/r04_fuzz_final.log
( echo "# FUZZ_DEVICES=3 python tools/fuzz_gpu.py 300 (FUZZ_SEED=20260930)"; FUZZ_DEVICES=3 FUZZ_SEED=20260930 timeout 600 python tools/fuzz_gpu.py 300 2>&1 | tail -25 ) >> $O/r04_fuzz_final.log
tail -5 $O/r04_fuzz_final.log
