#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python tools/ipa_probe.py 20 > gpurun_out/r2_ipa.log 2>&1
python tools/ipa_probe.py 16 >> gpurun_out/r2_ipa.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke.log 2>&1
python bench.py --no-cpu-baseline > gpurun_out/r2_bench_nocpu.json 2> gpurun_out/r2_bench_nocpu.err
