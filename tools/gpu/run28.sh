#!/bin/bash
O=gpurun_out/r3final; mkdir -p $O
(echo "# python tools/prover_pipeline_probe.py 17 ipa / 20 ipa (1 x MI355X, final tree of round 3)"; timeout 600 python tools/prover_pipeline_probe.py 17 ipa 2>/dev/null; timeout 600 python tools/prover_pipeline_probe.py 20 ipa 2>/dev/null) > $O/r03_pipeline.txt; tail -4 $O/r03_pipeline.txt
timeout 1200 python bench.py > $O/r03_bench.json 2> $O/r03_bench.err; python -c "
import json; r=json.load(open('$O/r03_bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['traffic'], r['config'].get('kernel_source_sha'))"
timeout 600 python bench.py --workload quotient --log-n 20 --steps 6 --warmup 2 > $O/r03_bench_quotient.json 2> $O/bench_quotient.err; python -c "
import json; q=json.load(open('$O/r03_bench_quotient.json')); print({k:(round(v['achieved'],1), round(v['frac'],3), v['traffic']) for k,v in q['rooflines'].items()})"
