#!/bin/bash
# first_8gpu.sh -- everything the first lease on a multi-GPU node should produce, in ONE command:
#   1. the two tests of the GPU suite that skip below two physical GPUs (real peer copies inside the single-process device group,
#      the RCCL backend at world size = device count);
#   2. the SCALE curve: bench.py --gpus 1 / 2 / 4 / 8 for the headline step (`both`, weak), the nine-commitment batch (`commit9`, strong,
#      BASELINE configs[3]) and the single sharded MSM (`msm --shard`, BLS12-377 2^22, strong, BASELINE configs[4]) - one JSON line each,
#      rank 0's components.multi_gpu carries the one-GPU time of the same run and the efficiency;
#   3. the single-process device group (the untouched-plonk.rs form) over all devices with PLK_VERBOSE=1:
#      which pairs have peer access, how many copies went peer to peer / through the host (plk_group_copy_stats).
# usage: bash tools/gpu/first_8gpu.sh [max_gpus]   -> gpurun_out/first8_*.{json,log}; nothing here needs the network or /root/reference
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
NG=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
MAXG=${1:-$NG}
echo "visible GPUs: $NG (running up to $MAXG)" | tee gpurun_out/first8_summary.log
rocm-smi --showtopo > gpurun_out/first8_topology.log 2>&1 || true
# 1. the tests that need >= 2 physical GPUs (they skip, and say so, on a one-GPU box)
( timeout 1800 python -m pytest tests/test_gpu_multi.py tests/test_gpu_fullsize.py -q -m gpu -rs -k "real_devices or nccl_every_visible_gpu" 2>&1 | tail -15 ) | tee gpurun_out/first8_tests.log
# 2. the scaling curve, one process per GPU over RCCL (bench.py spawns its ranks itself; the driver's torchrun form is equivalent)
for N in 1 2 4 8; do
  [ "$N" -gt "$MAXG" ] && continue
  for W in "both" "commit9" "msm --shard --curve bls12_377 --log-n 22"; do
    tag=$(echo "$W" | awk '{print $1}'); [ "$tag" = msm ] && tag=msm_shard_bls12_377_2p22
    timeout 1500 python bench.py --gpus $N --workload $W --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/first8_${tag}_n$N.json 2> gpurun_out/first8_${tag}_n$N.err
    python - "$tag" "$N" <<'PY' | tee -a gpurun_out/first8_summary.log
import json, sys
tag, n = sys.argv[1], sys.argv[2]
try:
    d = json.load(open("gpurun_out/first8_%s_n%s.json" % (tag, n)))
    mg = d.get("components", {}).get("multi_gpu", {})
    print("%-28s N=%s value %.1f %s  ms_per_step %.4f  scaling %s  multi_gpu %s" % (tag, n, d["value"], d["unit"], d["ms_per_step"], d.get("scaling"), json.dumps(mg)[:400]))
except Exception as e:
    print("%-28s N=%s FAILED: %r (see gpurun_out/first8_%s_n%s.err)" % (tag, n, e, tag, n))
PY
  done
done
# 3. one process, all devices behind the C ABI (what an untouched plonk.rs would run), peer-path statistics
if [ "$MAXG" -ge 2 ]; then
  PLK_VERBOSE=1 timeout 1500 python bench.py --gpus $MAXG --single-process --workload commit9 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/first8_single_process_n$MAXG.json 2> gpurun_out/first8_single_process_n$MAXG.err
  grep -i "peer\|copies\|staged" gpurun_out/first8_single_process_n$MAXG.err | head -20 | tee -a gpurun_out/first8_summary.log
  PLK_VERBOSE=1 PLK_PEER_MODE=host timeout 1500 python bench.py --gpus $MAXG --single-process --workload commit9 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/first8_single_process_hoststaged_n$MAXG.json 2> gpurun_out/first8_single_process_hoststaged_n$MAXG.err
fi
echo "done: gpurun_out/first8_summary.log"
