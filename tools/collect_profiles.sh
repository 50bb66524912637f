#!/bin/bash
# after tools/gpu/r04_evidence.sh came back: the files the round's profiles/ are made of (gpurun_out/ is scratch, profiles/ is tracked)
# usage: tools/collect_profiles.sh [round tag, default r04] [evidence directory, default gpurun_out/r4ev]
R=${1:-r04}; E=${2:-gpurun_out/r4ev}
cd "$(dirname "$0")/.."
for f in $E/${R}_*; do cp "$f" profiles/; done
cp $E/pytest.txt profiles/${R}_gpu_suite.txt
for f in gpurun_out/${R}_rocprofv3_* gpurun_out/${R}_pmc_traffic*.json gpurun_out/${R}_bench_profiled_run.json gpurun_out/${R}_bench_quotient_profiled_run.json; do
  [ -f "$f" ] && cp "$f" profiles/
done
ls -la profiles/ | grep "${R}_" | wc -l
