#!/bin/bash
# nine 2^20 vectors in one call: ordering + accumulation of consecutive vectors on k streams in turn (PLK_MSM_FORK_LARGE=k) against one stream
O=gpurun_out/r6fl; mkdir -p $O
run() { python bench.py --workload commit9 --steps 20 --warmup 5 --timed-only --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('%.3f ms per step, %.1f M pairs/s, checks %s' % (r['ms_per_step'], r['value'], all(r['checks'].values())))"; }
(for rep in 1 2; do
 echo "# default"; run
 for k in 2 3 9; do echo "# PLK_MSM_FORK_LARGE=$k"; PLK_MSM_FORK_LARGE=$k run; done
done) > $O/fork_large.txt 2>&1
cat $O/fork_large.txt
