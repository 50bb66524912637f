#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash tools/profile_round.sh r02 > gpurun_out/r02_profile_round.log 2>&1
( echo "# python tools/fuzz_gpu.py 240   (1 x MI355X, FUZZ_SEED=12345; every case bit-exact against the oracle or the run aborts)"; python tools/fuzz_gpu.py 240 ) > gpurun_out/r02_fuzz.log 2>&1
