#!/bin/bash
# The rocprofv3 passes behind profiles/rNN_*: kernel trace of bench.py's timed region, then FETCH_SIZE and WRITE_SIZE PMC passes
# (separate runs, kernel trace only, as MI355X_MICROARCH.md prescribes), for the headline workload AND for --workload quotient
# (the vanishing points and the generator fold), plus one SQ_INSTS_VALU pass (instructions per element of the NTT pass kernel and
# per addition of the accumulation).  Run on the GPU box from the repo root; results land in gpurun_out/.
# usage: tools/profile_round.sh [round tag, default r04]
RN=${1:-r04}
export TMPDIR=/tmp; R="${GRAFT_REPO_ROOT:-$PWD}"; cd /tmp
O=$R/gpurun_out
rm -rf $O/prof_$RN $O/pmc_fetch $O/pmc_write $O/pmc_valu $O/prof_q $O/pmc_fetch_q $O/pmc_write_q
rocprofv3 --kernel-trace --stats -d $O/prof_$RN -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --timed-only > $O/${RN}_bench_profiled_run.json 2> $O/prof_bench.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --timed-only > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --timed-only > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace -d $O/pmc_valu -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --timed-only > /dev/null 2>&1
# where the cycles of a wave go (one SQ pass, 8 slots): WAIT_ANY (parked on s_waitcnt / a barrier) + WAIT_INST_ANY (issue stall; WAIT_INST_LDS is
# its LDS part) + ACTIVE_INST_ANY ~ WAVE_CYCLES, all in quad-cycles; ACTIVE_INST_VALU and the instruction counts beside them
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d $O/pmc_sq -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --timed-only > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_q -o bench -- python $R/bench.py --workload quotient --steps 4 --warmup 1 > $O/${RN}_bench_quotient_profiled_run.json 2> $O/prof_q.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_q -o bench -- python $R/bench.py --workload quotient --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_q -o bench -- python $R/bench.py --workload quotient --steps 2 --warmup 1 > /dev/null 2>&1
cd $R
db() { find $1 -name "*.db" | head -1; }
python tools/rocpd_summary.py $(db $O/prof_$RN) > $O/${RN}_rocprofv3_kernel_stats.txt 2>&1; head -30 $O/${RN}_rocprofv3_kernel_stats.txt
python tools/rocpd_summary.py $(db $O/prof_q) > $O/${RN}_rocprofv3_kernel_stats_quotient.txt 2>&1; head -12 $O/${RN}_rocprofv3_kernel_stats_quotient.txt
python tools/pmc_traffic.py $(db $O/pmc_fetch) $(db $O/pmc_write) 20 $O/${RN}_pmc_traffic.json
python tools/pmc_traffic.py $(db $O/pmc_fetch_q) $(db $O/pmc_write_q) 20 $O/${RN}_pmc_traffic_quotient.json quotient
python tools/rocpd_summary.py $(db $O/pmc_fetch) --pmc > $O/${RN}_rocprofv3_pmc_fetch_size.txt 2>&1; python tools/rocpd_summary.py $(db $O/pmc_write) --pmc > $O/${RN}_rocprofv3_pmc_write_size.txt 2>&1
python tools/rocpd_summary.py $(db $O/pmc_fetch_q) --pmc > $O/${RN}_rocprofv3_pmc_fetch_size_quotient.txt 2>&1; python tools/rocpd_summary.py $(db $O/pmc_write_q) --pmc > $O/${RN}_rocprofv3_pmc_write_size_quotient.txt 2>&1
python tools/rocpd_summary.py $(db $O/pmc_sq) --pmc > $O/${RN}_rocprofv3_pmc_sq_wave_cycles.txt 2>&1; head -40 $O/${RN}_rocprofv3_pmc_sq_wave_cycles.txt | cut -c1-200
python tools/rocpd_summary.py $(db $O/pmc_valu) --pmc > $O/${RN}_rocprofv3_pmc_sq_insts_valu.txt 2>&1; grep -A8 "^PMC" $O/${RN}_rocprofv3_pmc_sq_insts_valu.txt | head -12
(rocm-smi --showproductname; rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8; lscpu | grep -E "Model name|^CPU\(s\)") > $O/${RN}_gpu_box_info.txt 2>&1
rm -rf $O/prof_$RN $O/pmc_fetch $O/pmc_write $O/pmc_valu $O/pmc_sq $O/prof_q $O/pmc_fetch_q $O/pmc_write_q
tail -2 $O/prof_bench.err; tail -2 $O/prof_q.err
