#!/bin/bash
# round-3 GPU run 1: suite, ceilings, baseline bench, emulated-rank shard timings with window sweeps
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python tools/measure_ceilings.py $O > $O/ceilings.log 2>&1; tail -12 $O/ceilings.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
for N in 2 4 8; do
  timeout 300 python bench.py --workload commit9 --emulate-rank 0/$N --steps 10 --warmup 2 > $O/commit9_emu_$N.json 2> $O/commit9_emu_$N.err
  python - <<PY
import json; r=json.load(open("$O/commit9_emu_$N.json")); print("commit9 emu 0/$N", "ms/step %.3f"%r["ms_per_step"], "window", r["components"]["msm_window_bits"], r["components"]["msm_stage_ms"])
PY
done
timeout 300 python bench.py --workload commit9 --steps 10 --warmup 2 --no-cpu-baseline > $O/commit9_1.json 2> $O/commit9_1.err
python -c "import json; r=json.load(open('$O/commit9_1.json')); print('commit9 N=1 ms/step %.3f'%r['ms_per_step'])"
# window sweep for the shard sizes of N = 8 / 4 / 2 (2^17 / 2^18 / 2^19 generators)
for N in 8 4 2; do for W in 14 15 16 17 18 19 20; do
  PLK_MSM_WINDOW=$W timeout 300 python bench.py --workload commit9 --emulate-rank 0/$N --steps 6 --warmup 2 --no-check > $O/c9_${N}_w$W.json 2>/dev/null
  python -c "import json; r=json.load(open('$O/c9_${N}_w$W.json')); print('commit9 0/$N window $W: %.3f ms/step'%r['ms_per_step'], r['components']['msm_stage_ms'])" 2>/dev/null || echo "commit9 0/$N window $W failed"
done; done
for N in 1 2 4 8; do
  timeout 600 python bench.py --workload msm --shard --curve bls12_377 --log-n 22 --emulate-rank 0/$N --steps 5 --warmup 2 > $O/bls22_emu_$N.json 2> $O/bls22_emu_$N.err
  python -c "import json; r=json.load(open('$O/bls22_emu_$N.json')); print('bls 2^22 emu 0/$N %.3f ms/step'%r['ms_per_step'], 'window', r['components']['msm_window_bits'], r['components']['msm_stage_ms'])" || tail -3 $O/bls22_emu_$N.err
done
