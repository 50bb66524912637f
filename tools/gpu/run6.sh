#!/bin/bash
O=gpurun_out/r3f; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 600 python tools/ipa_probe.py 20 14 > $O/ipa20.txt 2>&1; cat $O/ipa20.txt | head -24
PLK_HALO_NO_GRAPH=1 timeout 600 python tools/ipa_probe.py 20 14 > $O/ipa20_nograph.txt 2>&1; head -2 $O/ipa20_nograph.txt
timeout 900 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; python - <<PY
import json; r=json.load(open("$O/bench_default.json")); c=r["components"]; print("value", r["value"], "ms/step", r["ms_per_step"]); print({k:v for k,v in c.items() if k!="host_pointer"}); print(c.get("host_pointer")); print(r["checks"])
PY
for N in 1 8; do
  timeout 300 python bench.py --workload commit9 --emulate-rank 0/$N --steps 10 --warmup 2 > $O/commit9_emu_$N.json 2> $O/commit9_emu_$N.err
  python -c "import json; r=json.load(open('$O/commit9_emu_$N.json')); print('commit9 emu 0/$N ms/step %.3f'%r['ms_per_step'], r['checks'], r['components']['msm_stage_ms'])" || tail -3 $O/commit9_emu_$N.err
done
