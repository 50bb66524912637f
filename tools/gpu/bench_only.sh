#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
