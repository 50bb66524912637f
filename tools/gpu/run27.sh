#!/bin/bash
PROBE_ROUNDS=1 python tools/prover_pipeline_probe.py 20 ipa 2>&1 | grep -v amdgpu.ids | tail -48
