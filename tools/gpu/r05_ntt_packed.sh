#!/bin/bash
# round 5: the 36-byte limb form between passes (tables.cuh, PLK_LIMB_PACKED) against the padded 48-byte form it replaces
# (ab_libs/libplonky_hip_pad48.so = the same tree built with -DPLK_LIMB_PACKED=0): parity of every caller first, then alternating timings,
# one process per measurement
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
O=gpurun_out/r05_ntt_packed.txt
: > $O
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_poly.py tests/test_gpu_plonk.py -x -q -m gpu -k "ntt or fft or poly or lde or divide or vanish or quotient or constraint or padded" 2>&1 | tail -3 ) >> $O
for rep in 1 2 3; do
  for lib in ab_libs/libplonky_hip_pad48.so plonky_amd/csrc/libplonky_hip.so; do
    echo "== $lib" >> $O
    PLK_HIP_LIB=$PWD/$lib timeout 300 python tools/ntt_stagger_probe.py 2>/dev/null >> $O
    PLK_HIP_LIB=$PWD/$lib timeout 300 python tools/ntt_probe.py 2>/dev/null | grep -E "batch 9|log_n 23|batch 64|batch 4" >> $O
  done
done
for lib in ab_libs/libplonky_hip_pad48.so plonky_amd/csrc/libplonky_hip.so; do
  echo "== $lib quotient" >> $O
  PLK_HIP_LIB=$PWD/$lib timeout 600 python bench.py --workload quotient --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    try: d = json.loads(line)
    except Exception: continue
    c = d.get('components', {})
    print({k: c[k] for k in c if isinstance(c[k], (int, float))})
" >> $O
done
cat $O
