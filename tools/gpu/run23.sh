#!/bin/bash
mkdir -p gpurun_out/r3s
build/mul_latency > gpurun_out/r3s/mul_latency.txt 2>&1; cat gpurun_out/r3s/mul_latency.txt
