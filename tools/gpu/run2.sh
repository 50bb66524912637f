#!/bin/bash
O=gpurun_out/r3b; mkdir -p $O build
timeout 900 python -m pytest tests/test_gpu_halo.py tests/test_gpu_parity.py -m gpu -x -q -k "halo or fft_precompute_table" > $O/pytest_halo.txt 2>&1; tail -15 $O/pytest_halo.txt
timeout 600 python tools/ipa_probe.py 20 14 12 16 17 18 > $O/ipa20.txt 2>&1; cat $O/ipa20.txt
timeout 300 python tools/ipa_probe.py 16 14 > $O/ipa16.txt 2>&1; head -3 $O/ipa16.txt
/opt/rocm/bin/hipcc -O2 -std=c++17 -pthread --offload-arch=gfx950 -o build/h2d_probe tools/h2d_probe.cpp 2>/dev/null && timeout 300 build/h2d_probe > $O/h2d_probe.txt 2>&1; cat $O/h2d_probe.txt
timeout 300 python tools/gpu/graph_probe.py > $O/graph_probe.txt 2>&1; cat $O/graph_probe.txt
for N in 1 8; do
timeout 300 python bench.py --workload commit9 --emulate-rank 0/$N --steps 10 --warmup 2 --no-check > $O/commit9_emu_$N.json 2> $O/commit9_emu_$N.err
python -c "import json; r=json.load(open('$O/commit9_emu_$N.json')); print('commit9 emu 0/$N ms/step %.3f'%r['ms_per_step'], r['components']['msm_stage_ms'])"
done
