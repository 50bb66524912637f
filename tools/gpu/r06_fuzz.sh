#!/bin/bash
# the randomised parity stress with the geometries drawn per case as well (FUZZ_GEOMETRY=1: PLK_MSM_SLICE / PLK_MSM_GLOG / PLK_NTT_PLAN), final tree:
# one device, then a device group of three virtual devices
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( rocm-smi --showuniqueid 2>/dev/null | grep "^GPU\["; echo "# FUZZ_GEOMETRY=1 python tools/fuzz_gpu.py 420 (FUZZ_SEED=20260931, the final tree)"; FUZZ_GEOMETRY=1 FUZZ_SEED=20260931 timeout 700 python tools/fuzz_gpu.py 420 2>&1 | tail -25 ) > gpurun_out/r06_fuzz_geometry.log
( echo "# FUZZ_GEOMETRY=1 FUZZ_DEVICES=3 python tools/fuzz_gpu.py 200 (FUZZ_SEED=20261002, the final tree)"; FUZZ_GEOMETRY=1 FUZZ_DEVICES=3 FUZZ_SEED=20261002 timeout 400 python tools/fuzz_gpu.py 200 2>&1 | tail -25 ) > gpurun_out/r06_fuzz_geometry_devices3.log
tail -12 gpurun_out/r06_fuzz_geometry.log; tail -12 gpurun_out/r06_fuzz_geometry_devices3.log
