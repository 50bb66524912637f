#!/bin/bash
# Round 6: everything profiles/r06_* is made from, on ONE lease and ONE tree: the GPU suite, the rocprofv3 passes (kernel stats, PMC traffic,
# wave cycles, instruction counts - bench.py ties the traffic to the device-source hash and prints the rocprof average beside its live event
# time), THEN the bench lines (default, quotient, single process over 2 / 8 virtual devices), the opening argument, the proof pipeline, the
# emulated strong-scaling sweeps and the field-operation table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6ev; mkdir -p $O build
(rocm-smi --showuniqueid --showclocks --showpower 2>/dev/null | grep "^GPU\["; nproc; lscpu | grep "Model name") > $O/r06_gpu_box_info.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -q --durations=10 > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt
bash tools/profile_round.sh r06 > $O/profile_round.log 2>&1; tail -6 $O/profile_round.log | cut -c1-200
cp gpurun_out/r06_pmc_traffic.json gpurun_out/r06_pmc_traffic_quotient.json gpurun_out/r06_rocprofv3_kernel_stats.txt gpurun_out/r06_rocprofv3_kernel_stats_quotient.txt profiles/
timeout 300 python tools/measure_ceilings.py $O r06 > $O/ceilings.log 2>&1; tail -3 $O/ceilings.log
timeout 1500 python bench.py > $O/r06_bench.json 2> $O/r06_bench.err; python - <<PY
import json
r=json.load(open("$O/r06_bench.json")); c=r["components"]
print('bench value', r['value'], 'ms/step', r['ms_per_step'], 'profiled', r['ms_per_step_profiled'], all(r['checks'].values()))
print({k: round(c[k], 4) for k in c if (k.endswith('_ms') or k.endswith('per_s')) and isinstance(c[k], float)}); print(c['msm_stage_ms']); print(c.get('host_pointer'))
a=r['rooflines']['msm_accumulate']; print('acc frac', a['frac'], 'nominal', a['frac_nominal'], 'own', a['frac_own'], 'rocprof', a.get('rocprof_avg_ms'), 'traffic', a['traffic'], 'ntt', r['rooflines']['ntt_pass']['frac']); print(r['cpu_baseline'])
PY
timeout 600 python bench.py --workload quotient --log-n 20 --steps 6 --warmup 2 > $O/r06_bench_quotient.json 2> $O/bench_quotient.err
timeout 900 python bench.py --gpus 2 --single-process --virtual-devices --steps 20 2>/dev/null | grep "^{" > $O/r06_bench_single_process_virtual2.json
timeout 900 python bench.py --gpus 8 --single-process --virtual-devices --steps 20 2>/dev/null | grep "^{" > $O/r06_bench_single_process_virtual8.json
(echo "# python tools/ipa_probe.py 20 14 tabled / 16 14 tabled (1 x MI355X, round 6): the plain argument, then the one over the prover's tables"; timeout 600 python tools/ipa_probe.py 20 14 tabled 2>/dev/null; timeout 300 python tools/ipa_probe.py 16 14 tabled 2>/dev/null | head -2) > $O/r06_ipa.txt; head -3 $O/r06_ipa.txt
(echo "# python tools/prover_pipeline_probe.py 17 ipa / 20 ipa (1 x MI355X, round 6)"; timeout 600 python tools/prover_pipeline_probe.py 17 ipa 2>/dev/null; timeout 600 python tools/prover_pipeline_probe.py 20 ipa 2>/dev/null) > $O/r06_pipeline.txt; tail -3 $O/r06_pipeline.txt
bash tools/gpu/r06_scaling.sh > $O/scaling.log 2>&1; cp gpurun_out/r06_commit9_scaling.txt $O/r06_commit9_scaling_rerun.txt; tail -16 $O/scaling.log
timeout 1800 python bench.py --workload crossover --log-n 20 > $O/r06_crossover.json 2> $O/r06_crossover_host_pointer_vs_cpu.txt; tail -4 $O/r06_crossover_host_pointer_vs_cpu.txt | cut -c1-200
