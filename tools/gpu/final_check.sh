#!/bin/bash
# last check of a tree: the GPU suite, smoke(), the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 ) > gpurun_out/r2_tfull.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke.log 2>&1
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
