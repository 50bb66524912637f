#!/bin/bash
# round 5: the GPU suite, then where the wave cycles of the MSM's kernels go (one SQ counter pass, kernel trace only) and their durations
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( timeout 1700 python -m pytest tests -q -m gpu --durations=6 2>&1 | tail -25 ) > $O/r05_suite.log
export TMPDIR=/tmp; cd /tmp
rm -rf $O/pmc_sq $O/prof_msm
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d $O/pmc_sq -o bench -- python $R/bench.py --workload msm --steps 3 --warmup 1 --no-cpu-baseline --no-check --timed-only > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_msm -o bench -- python $R/bench.py --workload msm --steps 10 --warmup 2 --no-cpu-baseline --timed-only > /dev/null 2> $O/prof_msm.err
cd $R
db() { find $1 -name "*.db" | head -1; }
python tools/rocpd_summary.py $(db $O/pmc_sq) --pmc > $O/r05a_pmc_sq_wave_cycles_msm.txt 2>&1
python tools/rocpd_summary.py $(db $O/prof_msm) > $O/r05a_kernel_stats_msm.txt 2>&1
rm -rf $O/pmc_sq $O/prof_msm
tail -12 $O/r05_suite.log; head -24 $O/r05a_kernel_stats_msm.txt; grep -A3 "k_msm_accumulate\|k_ord_scatter\|k_msm_gsum\|k_msm_final" $O/r05a_pmc_sq_wave_cycles_msm.txt | cut -c1-220 | head -40
