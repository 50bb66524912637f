#!/bin/bash
O=gpurun_out/r3j; mkdir -p $O
for W in 16 17 18 19 20; do
  PLK_MSM_WINDOW=$W timeout 600 python bench.py --workload msm --shard --curve bls12_377 --log-n 22 --emulate-rank 0/8 --steps 5 --warmup 2 --no-check > $O/bls22_8_w$W.json 2>/dev/null
  python -c "import json; r=json.load(open('$O/bls22_8_w$W.json')); print('bls 2^22 0/8 window $W: %.3f ms/step'%r['ms_per_step'], r['components']['msm_stage_ms'])" 2>/dev/null || echo "window $W failed"
done
for W in 16 17 18 19 20; do
  PLK_MSM_WINDOW=$W timeout 600 python bench.py --workload msm --shard --curve bls12_377 --log-n 22 --emulate-rank 0/4 --steps 5 --warmup 2 --no-check > $O/bls22_4_w$W.json 2>/dev/null
  python -c "import json; r=json.load(open('$O/bls22_4_w$W.json')); print('bls 2^22 0/4 window $W: %.3f ms/step'%r['ms_per_step'], r['components']['msm_stage_ms'])" 2>/dev/null || echo "window $W failed"
done
for W in 16 18 19 20; do
  PLK_MSM_WINDOW=$W timeout 600 python bench.py --workload msm --shard --log-n 20 --emulate-rank 0/4 --steps 8 --warmup 2 --no-check > $O/tw20_4_w$W.json 2>/dev/null
  python -c "import json; r=json.load(open('$O/tw20_4_w$W.json')); print('tweedledee 2^20 0/4 (2^18) window $W: %.3f ms/step'%r['ms_per_step'], r['components']['msm_stage_ms'])" 2>/dev/null || echo "window $W failed"
done
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.txt 2>&1; grep -E "passed|failed|s call|s setup" $O/pytest.txt | head -12
