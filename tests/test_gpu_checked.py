"""The checked build (libplonky_hip_checked.so, -DPLK_CHECKED: bounds guards on every index the MSM's ordering and accumulation
kernels compute - SURVEY.md section 5, the device-side stand-in for the reference's debug assertions) runs a reduced set of MSMs
- tabled and table-free, uniform, sparse and skewed scalars, two curves, several windows - with bit-exact results and ZERO
guard violations.  The library is selected through PLK_HIP_LIB in a subprocess (one HIP library per process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECKED = os.path.join(ROOT, "plonky_amd", "csrc", "libplonky_hip_checked.so")

SCRIPT = r'''
import ctypes, sys
import numpy as np
import torch
from plonky_amd import device as dev, lib, synth
from plonky_amd.selfcheck import GENERATORS, closed_form_msm, _mul
from plonky_amd.synth import MODULI
L = lib.load()
assert L.plk_checked_build() == 1, "not the checked build"
dev.init(0)
cases = 0
for curve, bf, sf in ((0, 0, 1), (2, 3, 2)):
    p = MODULI[bf]
    G = GENERATORS[curve]
    D = _mul(p, 987654321, G)
    g0 = np.stack([synth.mont(bf, G[0]), synth.mont(bf, G[1])]); dd = np.stack([synth.mont(bf, D[0]), synth.mont(bf, D[1])])
    for n in (1, 37, 4096, 1 << 16):
        bases = dev.gen_bases_dev(curve, n, g0, dd)
        s = synth.rand_field(sf, 11 + n, n)
        variants = {"uniform": s}
        sp = s.copy(); sp[::3] = 0; variants["sparse"] = sp                      # a third of the scalars zero
        sk = s.copy(); sk[:, 1:] = 0; sk[:, 0] &= np.uint64(0xF); variants["skewed"] = sk     # sixteen distinct scalars: every window has a few very hot buckets (the heavy-bucket path)
        for table_free in (False, True):
            for window in ((0, 8, 16) if n >= 4096 else (0,)):
                if table_free and window > 16:
                    continue
                pre = dev.msm_precompute_dev(curve, bases, device_window=window, table_free=table_free)
                for name, sv in variants.items():
                    xy, z = dev.msm_execute_dev(pre, dev.to_device(sv))
                    torch.cuda.synchronize()
                    exp = closed_form_msm(curve, sv, G, D)
                    got = dev.to_host(xy)[0]
                    if exp is None:
                        assert int(z[0]) == 1, (curve, n, table_free, window, name)
                    else:
                        assert int(z[0]) == 0 and (synth.from_mont(bf, got[0]), synth.from_mont(bf, got[1])) == exp, (curve, n, table_free, window, name)
                    cases += 1
                pre.free()
    # a batch through the shared reduction
    n = 1 << 14
    bases = dev.gen_bases_dev(curve, n, g0, dd)
    pre = dev.msm_precompute_dev(curve, bases)
    sv = np.stack([synth.rand_field(sf, 500 + k, n) for k in range(3)])
    xy, z = dev.msm_execute_dev(pre, dev.to_device(sv))
    torch.cuda.synchronize()
    for k in range(3):
        got = dev.to_host(xy)[k]
        assert (synth.from_mont(bf, got[0]), synth.from_mont(bf, got[1])) == closed_form_msm(curve, sv[k], G, D)
    cases += 3
counts = (ctypes.c_uint * 8)()
lib.check(L.plk_checked_failures(counts))
print("CHECKED cases", cases, "violations", list(counts))
assert not any(counts), list(counts)
'''


def test_checked_build_msm_subset_has_no_violations():
    assert os.path.exists(CHECKED), "libplonky_hip_checked.so is missing: python -c 'import __graft_entry__ as g; g.build()'"
    env = dict(os.environ, PLK_HIP_LIB=CHECKED, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "CHECKED cases" in out.stdout and "violations [0, 0, 0, 0, 0, 0, 0, 0]" in out.stdout


def test_normal_build_reports_itself():
    import ctypes
    from plonky_amd import lib
    L = lib.load()
    assert L.plk_checked_build() == 0
    counts = (ctypes.c_uint * 8)()
    lib.check(L.plk_checked_failures(counts))
    assert not any(counts)
