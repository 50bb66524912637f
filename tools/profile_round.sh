#!/bin/bash
# The rocprofv3 passes behind profiles/rNN_*: kernel trace of bench.py's timed region, then FETCH_SIZE and WRITE_SIZE
# PMC passes (separate runs, kernel trace only).  Run on the GPU box from the repo root; results land in gpurun_out/.
# usage: tools/profile_round.sh [round tag, default r02]
RN=${1:-r02}
export TMPDIR=/tmp; R="${GRAFT_REPO_ROOT:-$PWD}"; cd /tmp
rm -rf $R/gpurun_out/prof_$RN $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$RN -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --timed-only > $R/gpurun_out/${RN}_bench_profiled_run.json 2> $R/gpurun_out/prof_bench.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --timed-only > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --timed-only > /dev/null 2>&1
cd $R
F=$(find gpurun_out/prof_$RN -name "*.db" | head -1); python tools/rocpd_summary.py $F > gpurun_out/${RN}_rocprofv3_kernel_stats.txt 2>&1; head -40 gpurun_out/${RN}_rocprofv3_kernel_stats.txt
FF=$(find gpurun_out/pmc_fetch -name "*.db" | head -1); FW=$(find gpurun_out/pmc_write -name "*.db" | head -1)
python tools/pmc_traffic.py $FF $FW 20 gpurun_out/${RN}_pmc_traffic.json
python tools/rocpd_summary.py $FF --pmc > gpurun_out/${RN}_rocprofv3_pmc_fetch_size.txt 2>&1; python tools/rocpd_summary.py $FW --pmc > gpurun_out/${RN}_rocprofv3_pmc_write_size.txt 2>&1
(rocm-smi --showproductname; rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8; lscpu | grep -E "Model name|^CPU\(s\)") > gpurun_out/${RN}_gpu_box_info.txt 2>&1
rm -rf gpurun_out/prof_$RN gpurun_out/pmc_fetch gpurun_out/pmc_write
tail -2 gpurun_out/prof_bench.err
