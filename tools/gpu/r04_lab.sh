#!/bin/bash
O=gpurun_out/r4d; mkdir -p $O
(rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID:" | head -1; ./ab_libs/madd_lab) > $O/madd_lab.txt 2>&1; cat $O/madd_lab.txt
