#!/bin/bash
# round 5: the opening argument against the length at which the generators are frozen, with the size rule of the comb (<= 2^12 generators)
# and with the comb forced for every frozen set (PLK_MSM_COMB=1: up to 2^15 generators)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
O=gpurun_out/r05_ipa_freeze.txt
: > $O
echo "== comb by size (the product)" >> $O
timeout 400 python tools/ipa_probe.py 20 10 11 12 13 14 15 16 tabled 2>&1 | grep -E "^IPA" >> $O
echo "== PLK_MSM_COMB=1 (comb for every tabled context up to 2^15 generators)" >> $O
PLK_MSM_COMB=1 timeout 400 python tools/ipa_probe.py 20 11 12 13 14 15 tabled 2>&1 | grep -E "^IPA" >> $O
cat $O
