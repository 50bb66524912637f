"""CPU-only: ProjectivePoint::to_affine as the library's host side computes it for the L_j / R_j of an inner-product-argument round
(plonky_amd/csrc/hostnorm.cpp; curve.rs:206-214: x / z, y / z, zero for z = 0), against Python integers on every curve's base field."""
import ctypes
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import bigint_ref as br
from tests.util import array_to_ints, ints_to_array

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# PLK_CURVE_* -> the field of its coordinates (include/plonky_hip.h)
CURVE_BASE = [(0, br.TWEEDLEDEE_BASE), (1, br.TWEEDLEDUM_BASE), (2, br.BLS12_377_BASE), (3, br.PALLAS_BASE), (4, br.VESTA_BASE)]


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hostnorm") / "hostnorm.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "hostnorm_harness.cpp")])
    L = ctypes.CDLL(so)
    L.hostnorm_to_affine.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return L


@pytest.mark.parametrize("curve,f", CURVE_BASE, ids=[f.name for _, f in CURVE_BASE])
def test_host_to_affine_matches_big_integers(lib, curve, f):
    p, nl = f.p, f.n_limbs
    R = 1 << (64 * nl)
    rng = random.Random(0x6A11 + curve)
    edge = [1, 2, p - 1, p - 2, (p - 1) // 2, (1 << (64 * nl - 1)) % p, R % p, (R * R) % p]
    pts, zero = [], []
    for k in range(64):
        x, y = rng.randrange(p), rng.randrange(p)
        z = edge[k % len(edge)] if k < 16 else rng.randrange(1, p)
        is_zero = k in (5, 40)          # ProjectivePoint::ZERO as emit_projective writes it: flag set, coordinates zero
        z_only = k in (7, 41)           # z = 0 with the flag clear: still the identity (curve.rs:207)
        if is_zero:
            x = y = z = 0
        if z_only:
            z = 0
        pts.append((x, y, z))
        zero.append(1 if is_zero else 0)
    mont = lambda v: v * R % p
    xyz = ints_to_array([mont(c) for pt in pts for c in pt], nl)
    flags = np.array(zero, dtype=np.uint8)
    out = np.zeros((len(pts) * 2, nl), dtype=np.uint64)
    assert lib.hostnorm_to_affine(curve, len(pts), xyz.ctypes.data, flags.ctypes.data, out.ctypes.data) == 0
    got = array_to_ints(out)
    for k, (x, y, z) in enumerate(pts):
        if z == 0:
            assert got[2 * k] == 0 and got[2 * k + 1] == 0
            continue
        zi = pow(z, -1, p)
        assert got[2 * k] == mont(x * zi % p), (f.name, k)
        assert got[2 * k + 1] == mont(y * zi % p), (f.name, k)


def test_host_to_affine_rejects_unknown_curve(lib):
    buf = np.zeros(64, dtype=np.uint64)
    assert lib.hostnorm_to_affine(9, 1, buf.ctypes.data, buf.ctypes.data, buf.ctypes.data) == -1
