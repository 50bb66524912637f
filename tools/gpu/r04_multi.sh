#!/bin/bash
# Round 4: the device group behind the C ABI on virtual devices (tests/test_gpu_multi.py incl. tests/capi_host multi), then the NTT
# harness reconciliation (tools/ntt_reconcile.py).
O=gpurun_out/r4b; mkdir -p $O
(rocm-smi --showuniqueid 2>/dev/null | grep -E "Unique" | head -2) > $O/box.txt 2>&1
timeout 600 python tests/multi_device_worker.py 2 14 $O/multi_2_14.npz > $O/multi_worker.log 2>&1; tail -5 $O/multi_worker.log
timeout 1800 python -m pytest tests/test_gpu_multi.py -x -q --durations=6 > $O/pytest_multi.txt 2>&1; tail -25 $O/pytest_multi.txt
timeout 600 python tools/ntt_reconcile.py > $O/ntt_reconcile.txt 2>&1; cat $O/ntt_reconcile.txt
