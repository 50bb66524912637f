#!/bin/bash
# round 6: the tile-major level-1 ordering (k_ord_tiles + gathered level 2) against round 5's kernels (PLK_MSM_ORDER_V1=1), same lease;
# parity first; then where the one-shot (table-free) MSM spends its time under the new reduction launches
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_msm_geometry.py tests/test_gpu_checked.py tests/test_gpu_halo.py tests/test_gpu_multi.py tests/test_gpu_msm_order.py -x -q -m gpu --durations=12 2>&1 | tail -30 ) > gpurun_out/r06_t1.log
tail -22 gpurun_out/r06_t1.log
for rep in 1 2; do
  for v in ord2 ord1; do
    unset PLK_MSM_ORDER_V1; [ $v = ord1 ] && export PLK_MSM_ORDER_V1=1
    timeout 600 python bench.py --workload msm --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r06_order_${v}_${rep}.json 2> gpurun_out/r06_order_${v}_${rep}.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r06_order_${v}_${rep}.json"))
c=d["components"]
print("${v} ${rep}", "msm_ms %.4f" % c.get("msm_ms"), "batch9 %.3f" % c.get("msm_batch9_ms"), "one-shot %.3f" % c.get("msm_parallel_one_shot_ms", 0), "ipa %.2f" % c.get("ipa_ms", 0), "stages", c.get("msm_stage_ms"), all(d.get("checks").values()))
PY
  done
done
unset PLK_MSM_ORDER_V1
( timeout 900 python -m pytest tests/test_gpu_msm_large.py tests/test_gpu_knobs.py -x -q -m gpu -k "2p16 or 2p18 or n349525 or knob" 2>&1 | tail -5 ) > gpurun_out/r06_t2.log
tail -3 gpurun_out/r06_t2.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ord -o ord -- python $GRAFT_REPO_ROOT/bench.py --workload msm --steps 20 --warmup 3 --no-cpu-baseline --timed-only > /dev/null 2> /tmp/prof_ord.err
cd "$GRAFT_REPO_ROOT"
python tools/rocpd_summary.py $(find /tmp/prof_ord -name "*.db" | head -1) > gpurun_out/r06_order_kernel_stats.txt 2>&1
grep "k_msm\|k_ord" gpurun_out/r06_order_kernel_stats.txt | cut -c1-120
