#!/bin/bash
# Round 3: everything profiles/r03_* is made from, on the FINAL sources (bench.py ties the PMC traffic to their hash).
O=gpurun_out/r3final; mkdir -p $O build
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt
timeout 300 python tools/measure_ceilings.py $O > $O/ceilings.log 2>&1; tail -3 $O/ceilings.log
timeout 1200 python bench.py > $O/r03_bench.json 2> $O/r03_bench.err; python -c "
import json; r=json.load(open('$O/r03_bench.json')); print('bench value', r['value'], 'ms/step', r['ms_per_step'], r['checks']); print(r['cpu_baseline'])" || tail -5 $O/r03_bench.err
timeout 600 python bench.py --workload quotient --log-n 20 --steps 6 --warmup 2 > $O/r03_bench_quotient.json 2> $O/bench_quotient.err
(echo "# python tools/ipa_probe.py 20 14 tabled / 16 14 tabled (1 x MI355X, final tree of round 3): the plain argument, then the one over the prover's tables"; timeout 600 python tools/ipa_probe.py 20 14 tabled 2>/dev/null; timeout 300 python tools/ipa_probe.py 16 14 tabled 2>/dev/null | head -2) > $O/r03_ipa.txt; head -3 $O/r03_ipa.txt
(echo "# python tools/prover_pipeline_probe.py 17 ipa / 20 ipa (1 x MI355X, final tree of round 3)"; timeout 600 python tools/prover_pipeline_probe.py 17 ipa 2>/dev/null; timeout 600 python tools/prover_pipeline_probe.py 20 ipa 2>/dev/null) > $O/r03_pipeline.txt; tail -3 $O/r03_pipeline.txt
(export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf $R/$O/pipa; rocprofv3 --kernel-trace --stats -d $R/$O/pipa -o ipa -- python $R/tools/ipa_probe.py 20 14 tabled > /dev/null 2>&1; cd $R
 (echo "# rocprofv3 --kernel-trace --stats -- python tools/ipa_probe.py 20 14 tabled: three plain arguments and three over the prover's tables (2^20, 20 rounds each)"; python tools/rocpd_summary.py $(find $O/pipa -name "*.db" | head -1)) > $O/r03_ipa_kernels.txt 2>&1; rm -rf $O/pipa; head -8 $O/r03_ipa_kernels.txt | cut -c1-150)
(export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf $R/$O/ppipe; rocprofv3 --kernel-trace --stats -d $R/$O/ppipe -o pipe -- python $R/tools/prover_pipeline_probe.py 20 ipa > /dev/null 2>&1; cd $R
 (echo "# rocprofv3 --kernel-trace --stats -- python tools/prover_pipeline_probe.py 20 ipa: every kernel of the hot path of a 2^20-gate proof (setup, two passes of the pipeline, two openings)"; python tools/rocpd_summary.py $(find $O/ppipe -name "*.db" | head -1)) > $O/r03_pipeline_kernels.txt 2>&1; rm -rf $O/ppipe; head -6 $O/r03_pipeline_kernels.txt | cut -c1-150)
for N in 1 2 4 8; do
  timeout 300 python bench.py --workload commit9 --emulate-rank 0/$N --steps 10 --warmup 2 > $O/commit9_emu_$N.json 2> $O/commit9_emu_$N.err
  timeout 600 python bench.py --workload msm --shard --curve bls12_377 --log-n 22 --emulate-rank 0/$N --steps 5 --warmup 2 > $O/bls22_emu_$N.json 2> $O/bls22_emu_$N.err
done
python - <<PY
import json
for wl in ("commit9", "bls22"):
    t1 = None
    for N in (1, 2, 4, 8):
        r = json.load(open("$O/%s_emu_%d.json" % (wl, N))); t = r["ms_per_step"]; t1 = t1 or t
        print("%s N = %d: %.3f ms  efficiency %.2f  checks %s" % (wl, N, t, t1 / N / t, all(r["checks"].values())))
PY
bash tools/profile_round.sh r03 > $O/profile_round.log 2>&1; tail -25 $O/profile_round.log
