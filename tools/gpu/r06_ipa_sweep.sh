#!/bin/bash
# IPA at 2^20 over the caller's tables: rounds per stage (PLK_HALO_LEAD over the tables, PLK_HALO_STAGE over explicit generators), smallest stage output
O=gpurun_out/r6ipa; mkdir -p $O
F='^IPA at n = 2\^20 over'
(for lead in 1 2 3 4; do for stage in 1 2 3; do
  echo -n "LEAD=$lead STAGE=$stage: "; PLK_HALO_LEAD=$lead PLK_HALO_STAGE=$stage timeout 200 python tools/ipa_probe.py 20 14 tabled 2>/dev/null | grep -E "$F" | sed 's/.*freeze_log 14: //; s/ in all.*//'
done; done
for ml in 15 16 17; do echo -n "default LEAD / STAGE, STAGE_MIN_LOG=$ml: "; PLK_HALO_STAGE_MIN_LOG=$ml timeout 200 python tools/ipa_probe.py 20 14 tabled 2>/dev/null | grep -E "$F" | sed 's/.*freeze_log 14: //; s/ in all.*//'; done
echo -n "default: "; timeout 200 python tools/ipa_probe.py 20 14 tabled 2>/dev/null | grep -E "$F" | sed 's/.*freeze_log 14: //; s/ in all.*//') > $O/ipa_sweep.txt 2>&1
cat $O/ipa_sweep.txt
