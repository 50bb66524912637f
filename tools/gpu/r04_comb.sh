#!/bin/bash
# the comb mode of small tabled contexts: its own test, the MSM / halo / multi-device parity tests that now run through it, the
# opening argument's time, a quick timing of small MSMs comb against buckets
O=gpurun_out/r4h; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "comb or msm" > $O/pytest_comb.txt 2>&1; tail -5 $O/pytest_comb.txt
timeout 1800 python -m pytest tests/test_gpu_halo.py tests/test_gpu_multi.py tests/test_gpu_parity.py tests/test_gpu_checked.py -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
(timeout 600 python tools/ipa_probe.py 20 14 tabled 2>/dev/null) > $O/ipa.txt; head -2 $O/ipa.txt; tail -16 $O/ipa.txt
python - <<'PY' 2>&1 | grep -v amdgpu
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from plonky_amd import device as dev, synth
from plonky_amd.selfcheck import GENERATORS, _mul
from plonky_amd.synth import MODULI
dev.init(0)
p = MODULI[0]; G = GENERATORS[0]; D = _mul(p, 77, G)
g0 = np.stack([synth.mont(0, G[0]), synth.mont(0, G[1])]); dd = np.stack([synth.mont(0, D[0]), synth.mont(0, D[1])])
for lg in (10, 12, 14, 15):
    n = 1 << lg
    bases = dev.gen_bases_dev(0, n, g0, dd)
    s2 = dev.to_device(np.stack([synth.rand_field(1, 5 + k, n) for k in range(2)]))
    res = {}
    for name, kw in (("comb", {}), ("buckets", {"device_window": 13 if lg >= 14 else 10})):
        t0 = time.perf_counter(); pre = dev.msm_precompute_dev(0, bases, **kw); torch.cuda.synchronize(); tp = time.perf_counter() - t0
        for batch in (1, 2):
            s = s2[:batch].contiguous()
            oxy, oz = dev.msm_execute_dev(pre, s); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50): dev.msm_execute_dev(pre, s, oxy, oz)
            torch.cuda.synchronize()
            res[(name, batch)] = ((time.perf_counter() - t0) / 50 * 1e3, dev.to_host(oxy).copy())
        res[(name, "pre")] = tp * 1e3
        pre.free()
    same = all(np.array_equal(res[("comb", b)][1], res[("buckets", b)][1]) for b in (1, 2))
    print("n = 2^%d: comb %.3f ms (batch 2: %.3f), buckets %.3f ms (batch 2: %.3f); precompute %.2f / %.2f ms; same points: %s" % (
        lg, res[("comb", 1)][0], res[("comb", 2)][0], res[("buckets", 1)][0], res[("buckets", 2)][0], res[("comb", "pre")], res[("buckets", "pre")], same), flush=True)
PY
