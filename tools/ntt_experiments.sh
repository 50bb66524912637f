#!/bin/bash
# Builds timing-only variants of the NTT pass kernel (results are wrong by construction) to split its
# run time into multiplications / LDS stages / memory: see PLK_NTT_EXP in plonky_amd/csrc/ntt.hip.
# Usage (container): tools/ntt_experiments.sh build ; (GPU box): tools/ntt_experiments.sh run
set -e
cd "$(dirname "$0")/../plonky_amd/csrc"
if [ "$1" = build ]; then
  make -j4 >/dev/null; mkdir -p ../../build_exp
  for e in ${EXPS:-1 2 3}; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DPLK_NTT_EXP=$e $EXTRA -c ntt.hip -o /tmp/ntt_e$e.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_exp/libplonky_hip_e$e.so capi.o /tmp/ntt_e$e.o msm.o fieldops.o poly.o fold.o
  done
else
  cd ../..
  for e in 0 1 2 3; do
    lib=""; [ $e != 0 ] && lib="$PWD/build_exp/libplonky_hip_e$e.so"
    echo "== PLK_NTT_EXP=$e"
    PLK_HIP_LIB=$lib python bench.py --workload ntt --no-check --no-cpu-baseline --steps 50 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); c = d['components']; print({k: round(v, 4) for k, v in c.items() if 'ntt' in k and 'ms' in k})"
  done
fi
