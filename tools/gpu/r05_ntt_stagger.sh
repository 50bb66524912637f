#!/bin/bash
# round 5: NTT with batched loads - parity (every suite that transforms), the staggered-start sweep, the host-pointer pipeline A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_poly.py tests/test_gpu_plonk.py tests/test_gpu_fullsize.py tests/test_gpu_halo.py -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r05_suite_ntt.log
: > gpurun_out/r05_ntt_stagger.txt
for rep in 1 2; do for s in 0 2 4 6; do PLK_NTT_STAGGER=$s timeout 300 python tools/ntt_stagger_probe.py 2>/dev/null >> gpurun_out/r05_ntt_stagger.txt; done; done
timeout 300 python tools/ntt_probe.py 2>/dev/null >> gpurun_out/r05_ntt_stagger.txt
timeout 600 python tools/host_ntt9_probe.py > gpurun_out/r05_host_ntt9.txt 2>&1
tail -4 gpurun_out/r05_suite_ntt.log; cat gpurun_out/r05_ntt_stagger.txt; tail -4 gpurun_out/r05_host_ntt9.txt
