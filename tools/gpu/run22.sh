#!/bin/bash
mkdir -p gpurun_out/r3r
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5
python tools/ipa_probe.py 20 14 tabled 2>&1 | grep -v amdgpu.ids | head -10
python tools/ipa_probe.py 14 14 2>&1 | grep "IPA at"
PLK_MSM_TABLE_FUSED=1 python tools/ipa_probe.py 14 14 2>&1 | grep "IPA at"
