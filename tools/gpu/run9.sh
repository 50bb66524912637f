#!/bin/bash
O=gpurun_out/r3i; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt; grep -E "^FAILED|Error" $O/pytest.txt | head
timeout 600 python tools/ipa_probe.py 20 14 > $O/ipa20.txt 2>&1; cat $O/ipa20.txt | head -14
for spec in "0:14" "0:16" "0:17"; do
  echo "default window:"; timeout 200 python tools/msm_probe.py $spec 2>/dev/null | grep curve
done
echo "old choices:"; PLK_MSM_WINDOW=11 timeout 200 python tools/msm_probe.py 0:14 2>/dev/null | grep curve; PLK_MSM_WINDOW=14 timeout 200 python tools/msm_probe.py 0:16 2>/dev/null | grep curve; PLK_MSM_WINDOW=15 timeout 200 python tools/msm_probe.py 0:17 2>/dev/null | grep curve
