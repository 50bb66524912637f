#!/bin/bash
mkdir -p gpurun_out/r3m
for fz in 13 14 15 16; do python tools/ipa_probe.py 20 $fz tabled 2>&1 | grep "IPA at"; done > gpurun_out/r3m/freeze_sweep.txt
cat gpurun_out/r3m/freeze_sweep.txt
