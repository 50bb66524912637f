#!/bin/bash
# round 6: k_ntt_pass with two tiles per workgroup (the second tile's loads under the first one's stages) for one-round launches, against
# the one-tile form (PLK_NTT_PIPE=0), same lease; transform parity first
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ntt_plans.py tests/test_gpu_poly.py -x -q -m gpu -k "ntt or fft or poly or division or values" 2>&1 | tail -4 ) > gpurun_out/r06_t6.log; tail -2 gpurun_out/r06_t6.log
for rep in 1 2 3; do
  for v in pipe nopipe; do
    unset PLK_NTT_PIPE; [ $v = nopipe ] && export PLK_NTT_PIPE=0
    timeout 600 python bench.py --workload ntt --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r06_ntt_${v}_${rep}.json 2> /dev/null
    python - <<PY
import json
d=json.load(open("gpurun_out/r06_ntt_${v}_${rep}.json")); c=d["components"]
print("${v} ${rep}", "ms_per_step %.4f" % d["ms_per_step"], "ntt_ms %.4f" % c["ntt_ms"], "batch9 G/s %.2f" % (c["ntt_batch9_melems_per_s"]/1e3), "lde9 %.3f" % c.get("lde9_ms",0), all(d["checks"].values()))
PY
  done
done
