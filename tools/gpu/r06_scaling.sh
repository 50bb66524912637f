#!/bin/bash
# round 6: emulated strong scaling (one rank's share of the N-rank problem alone on one GPU, incl. the record copy that stands in for the
# all-gather and the combine) for the nine-commitment batch and the single BLS12-377 2^22 MSM: base-range sharding against BUCKET-range
# sharding (--bucket-shard) of the sharded vectors.  Parity of the bucket ranges first.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_msm_order.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r06_t3.log
tail -4 gpurun_out/r06_t3.log
O=gpurun_out
for N in 1 2 4 8; do
  [ "$SKIP_BLS" = 1 ] && true
  for mode in base bucket; do
    flag=""; [ $mode = bucket ] && flag="--bucket-shard"
    [ $N = 1 ] && [ $mode = bucket ] && continue
    timeout 400 python bench.py --workload commit9 --emulate-rank 0/$N $flag --steps 10 --warmup 2 2>/dev/null | grep "^{" > $O/r06_commit9_emu_${mode}_$N.json
    timeout 900 python bench.py --workload msm --shard --curve bls12_377 --log-n 22 --emulate-rank 0/$N $flag --steps 5 --warmup 2 2>/dev/null | grep "^{" > $O/r06_bls22_emu_${mode}_$N.json
  done
done
python - <<'PY' | tee gpurun_out/r06_commit9_scaling.txt
import json
print("# bench.py --workload commit9 --emulate-rank 0/N and --workload msm --shard --curve bls12_377 --log-n 22 --emulate-rank 0/N (round 6, one box):")
print("# one rank's share of the N-rank problem alone on one GPU incl. the copy that stands in for the exchange and the combine; efficiency = T_1 / (N T_N)")
print("# base = sharded vectors shared by contiguous BASE range (rounds 2-5); bucket = by BUCKET range (--bucket-shard, plk_msm_execute_parts_buckets_dev)")
for wl in ("commit9", "bls22"):
    t1 = None
    for mode in ("base", "bucket"):
        for N in (1, 2, 4, 8):
            if N == 1 and mode == "bucket": continue
            try:
                d = json.load(open("gpurun_out/r06_%s_emu_%s_%d.json" % (wl, mode, N)))
            except Exception as e:
                print("%s %s N = %d: FAILED %r" % (wl, mode, N, e)); continue
            t = d["ms_per_step"]
            if N == 1: t1 = t
            print("%s %-6s N = %d: %.3f ms  efficiency %.2f  checks %s" % (wl, mode, N, t, (t1 / (N * t)) if t1 else 0, all(d["checks"].values())))
PY
