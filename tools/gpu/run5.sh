#!/bin/bash
# round 3: NTT shuffle variant, parity of it, profiles, final-default scaling emulation
O=gpurun_out/r3e; mkdir -p $O build
# the wave-shuffle variant must give the same bits: the NTT parity tests with it switched on
PLK_NTT_SHUFFLE=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_poly.py -m gpu -x -q -k "ntt or fft or poly or lde or divide" > $O/pytest_shuffle.txt 2>&1; tail -3 $O/pytest_shuffle.txt
for V in 0 1; do
  echo "PLK_NTT_SHUFFLE=$V" >> $O/ntt_variants.txt
  for rep in 1 2 3; do PLK_NTT_SHUFFLE=$V timeout 300 python tools/ntt_probe.py 2>/dev/null | grep -E "log_n 20|log_n 23" >> $O/ntt_variants.txt; done
done
cat $O/ntt_variants.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python tools/ipa_probe.py 20 14 > $O/ipa20.txt 2>&1; head -3 $O/ipa20.txt
timeout 600 python bench.py --workload quotient --log-n 20 --steps 6 --warmup 2 > $O/bench_quotient.json 2> $O/bench_quotient.err; python -c "
import json; r=json.load(open('$O/bench_quotient.json')); print(r['components'], r['checks']); print({k:(v['achieved'], v['frac'], v['hbm']['frac']) for k,v in r['rooflines'].items()})" || tail -5 $O/bench_quotient.err
for N in 1 2 4 8; do
  timeout 300 python bench.py --workload commit9 --emulate-rank 0/$N --steps 10 --warmup 2 > $O/commit9_emu_$N.json 2> $O/commit9_emu_$N.err
  python -c "import json; r=json.load(open('$O/commit9_emu_$N.json')); print('commit9 emu 0/$N ms/step %.3f'%r['ms_per_step'], r['checks'])" || tail -3 $O/commit9_emu_$N.err
done
timeout 900 python bench.py --gpus 2 --same-device --workload commit9 --steps 5 --warmup 2 --no-cpu-baseline > $O/commit9_2ranks_one_gpu.json 2> $O/commit9_2ranks_one_gpu.err; python -c "
import json; r=json.load(open('$O/commit9_2ranks_one_gpu.json')); print('2 ranks on one GPU', r['ms_per_step'], r['checks'])"
bash tools/profile_round.sh r03 > $O/profile_round.log 2>&1; tail -40 $O/profile_round.log
echo "512-element tiles (PLK_NTT_TILE_LOG=9), plans (7,7,6) and (5,5,5,5) / (6,6,4,4)" >> $O/ntt_variants.txt
for PLAN in "" "5,5,5,5" "6,5,5,4"; do
  PLK_HIP_LIB=$PWD/variants/libplonky_hip_t9.so PLK_NTT_PLAN=$PLAN timeout 300 python tools/ntt_probe.py 2>&1 | grep -E "log_n 20|Error|error" | sed "s/^/t9 plan '$PLAN': /" >> $O/ntt_variants.txt
done
tail -12 $O/ntt_variants.txt
