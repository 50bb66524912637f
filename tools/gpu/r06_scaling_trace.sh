#!/bin/bash
# round 6: per-kernel times of ONE rank's step of the nine-commitment batch at 8 ranks (a whole vector + its share of the ninth), the share
# taken by base range and by bucket range
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for mode in base bucket; do
  flag=""; [ $mode = bucket ] && flag="--bucket-shard"
  cd /tmp && rm -rf /tmp/prof_sc_$mode && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_sc_$mode -o sc -- python $GRAFT_REPO_ROOT/bench.py --workload commit9 --emulate-rank 0/8 $flag --steps 10 --warmup 2 --timed-only > /dev/null 2>&1
  cd "$GRAFT_REPO_ROOT"
  python tools/rocpd_summary.py $(find /tmp/prof_sc_$mode -name "*.db" | head -1) > gpurun_out/r06_commit9_rank_of_8_kernels_$mode.txt 2>&1
  echo "== $mode"; grep "k_" gpurun_out/r06_commit9_rank_of_8_kernels_$mode.txt | cut -c1-100 | head -22
done
