#!/bin/bash
# NTT plan variants at 2^20: 3-pass (7,7,6) with 1024-element tiles (the product), and 4096-element tiles
# (147 KB of LDS, one workgroup of 1024 lanes per CU, stage twiddles read through L1 when they do not fit): 2-pass (10,10), (7,7,6).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() { # tag lib plan
  PLK_HIP_LIB=$2 PLK_NTT_PLAN=$3 timeout 200 python bench.py --workload ntt --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r2_nttv_$1.json 2> gpurun_out/r2_nttv_$1.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r2_nttv_$1.json"))
    c = d["components"]; r = d["roofline"]
    print("%-28s single %.4f ms  batch9 %.0f Melems/s  lde9 %.3f ms  launches %.1f x %.2f us  checks %s" % ("$1 ($3)", c["ntt_ms"], c["ntt_batch9_melems_per_s"], c["lde9_ms"], r["launches_per_transform"], r["launch_ms"] * 1e3, all(d["checks"].values())))
except Exception as e:
    print("%-28s FAILED: %s" % ("$1 ($3)", open("gpurun_out/r2_nttv_$1.err").read()[-300:].replace("\n", " ")))
PY
}
run t10_776 "" ""
run t12_1010 build_exp/libplonky_hip_t12.so "10,10"
run t12_776 build_exp/libplonky_hip_t12.so "7,7,6"
run t12_884 build_exp/libplonky_hip_t12.so "8,8,4"
