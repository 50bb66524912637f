#!/bin/bash
# round 5: pass sizes of the 2^20 transform once more on the final kernel (PLK_NTT_PLAN), including two passes of 2^10 on 1024-element tiles
# (one column per tile: strided 32-byte accesses)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
O=gpurun_out/r05_ntt_plans.txt
: > $O
for rep in 1 2; do
for plan in "" "6,7,7" "7,6,7" "8,6,6" "6,6,8" "10,10" "9,9,2" "8,8,4"; do
  echo "== plan '${plan}'" >> $O
  PLK_NTT_PLAN=$plan timeout 200 python - >> $O 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from plonky_amd import device as dev, synth
dev.init(0)
xh = synth.rand_field(0, 0xF70020, 1 << 20)
x = dev.to_device(xh); y = torch.empty_like(x)
for _ in range(20): dev.ntt_dev(0, x, out=y)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(200): dev.ntt_dev(0, x, out=y)
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 200)
ok = np.array_equal(dev.to_host(dev.ntt_dev(0, y, inverse=True)), xh)
xb = torch.randint(0, 1 << 60, (9, 1 << 20, 4), dtype=torch.int64, device="cuda"); xb[..., 3] &= (1 << 61) - 1
yb = torch.empty_like(xb)
for _ in range(3): dev.ntt_dev(0, xb, out=yb)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): dev.ntt_dev(0, xb, out=yb)
torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / 20
print("lone 2^20 %.1f us%s   nine %.2f G elements/s" % (best * 1e6, "" if ok else " ROUNDTRIP-MISMATCH", 9 * (1 << 20) / tb / 1e9))
PY
done
done
cat $O
