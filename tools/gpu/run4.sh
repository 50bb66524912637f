#!/bin/bash
O=gpurun_out/r3d; mkdir -p $O build
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
timeout 600 python tools/prover_pipeline_probe.py 20 > $O/pipeline20.txt 2>&1; grep -E "vanishing|hot-path|commit" $O/pipeline20.txt
/opt/rocm/bin/hipcc -O2 -std=c++17 -pthread --offload-arch=gfx950 -o build/h2d_probe tools/h2d_probe.cpp 2>/dev/null && timeout 300 build/h2d_probe > $O/h2d_probe.txt 2>&1; grep "3 streams" $O/h2d_probe.txt
timeout 900 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; python - <<PY
import json; r=json.load(open("$O/bench_default.json")); c=r["components"]; print("value", r["value"], "ms/step", r["ms_per_step"]); print({k:v for k,v in c.items() if k!="host_pointer"}); print(c.get("host_pointer")); print(r["checks"])
PY
for N in 1 2 4 8; do
  timeout 300 python bench.py --workload commit9 --emulate-rank 0/$N --steps 10 --warmup 2 > $O/commit9_emu_$N.json 2> $O/commit9_emu_$N.err
  python -c "import json; r=json.load(open('$O/commit9_emu_$N.json')); print('commit9 emu 0/$N ms/step %.3f'%r['ms_per_step'], r['checks'], r['components']['msm_stage_ms'])" || tail -3 $O/commit9_emu_$N.err
  PLK_MSM_NO_TAIL_PIPELINE=1 timeout 300 python bench.py --workload commit9 --emulate-rank 0/$N --steps 10 --warmup 2 --no-check > $O/commit9_emu_${N}_nopipe.json 2>/dev/null
  python -c "import json; r=json.load(open('$O/commit9_emu_${N}_nopipe.json')); print('  without pipelined tails: %.3f'%r['ms_per_step'])"
done
for N in 1 2 4 8; do
  timeout 600 python bench.py --workload msm --shard --curve bls12_377 --log-n 22 --emulate-rank 0/$N --steps 5 --warmup 2 > $O/bls22_emu_$N.json 2> $O/bls22_emu_$N.err
  python -c "import json; r=json.load(open('$O/bls22_emu_$N.json')); print('bls 2^22 emu 0/$N %.3f ms/step'%r['ms_per_step'], 'window', r['components']['msm_window_bits'], r['components']['msm_stage_ms'])" || tail -3 $O/bls22_emu_$N.err
done
timeout 600 python tools/ipa_probe.py 20 14 > $O/ipa20.txt 2>&1; head -3 $O/ipa20.txt
