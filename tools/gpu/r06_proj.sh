#!/bin/bash
# round 6: parity of the projective-output entry points and the MSM bench with the msm_projective_ms component
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_msm_order.py tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_halo.py -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r06_t4.log
tail -4 gpurun_out/r06_t4.log
timeout 600 python bench.py --workload msm --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r06_proj.json 2> gpurun_out/r06_proj.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_proj.json")); c=d["components"]
print("msm_ms %.4f projective %.4f batch9 %.3f" % (c["msm_ms"], c["msm_projective_ms"], c["msm_batch9_ms"]), d["checks"])
PY
tail -3 gpurun_out/r06_proj.err
