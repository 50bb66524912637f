"""CPU-only: the device field arithmetic of plonky_amd/csrc/fp.cuh + fp29.cuh, compiled for the host
with g++ (it is plain C++ outside hipcc), swept against Python integers on the reference's own
edge-value generator (src/field/field.rs:498-615) and on seeded random values.  This pins the
exact code the GPU runs (same header, same template instantiations) before it ever reaches a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import bigint_ref as br
from plonky_amd import synth
from tests.test_oracle_kats import reference_test_inputs
from tests.util import array_to_ints, ints_to_array

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = [br.TWEEDLEDEE_BASE, br.TWEEDLEDUM_BASE, br.BLS12_377_SCALAR, br.BLS12_377_BASE]


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("fp") / "fp_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "fp_host_harness.cpp")])
    L = ctypes.CDLL(so)
    L.fp_host_op.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    return L


def run(L, f, op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = a if b is None else np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    assert L.fp_host_op(f.field_id, op, a.ctypes.data, b.ctypes.data, out.ctypes.data, a.shape[0]) == 0
    return array_to_ints(out)


@pytest.mark.parametrize("f", FIELDS, ids=lambda f: f.name)
def test_device_header_arithmetic_on_host(host_lib, f):
    p, n = f.p, f.n_limbs
    inputs = reference_test_inputs(p)
    inputs += [synth.to_int(r) for r in synth.rand_field(f.field_id, 99, 200)]
    m = len(inputs)
    x = ints_to_array(inputs, n)
    ia = np.repeat(np.arange(m), m)
    ib = np.tile(np.arange(m), m)
    a, b = x[ia], x[ib]
    Rinv = f.Rinv
    got_add = run(host_lib, f, 0, a, b)
    got_sub = run(host_lib, f, 1, a, b)
    got_mul = run(host_lib, f, 2, a, b)
    got_cios = run(host_lib, f, 3, a, b)
    k = 0
    for i in range(m):
        ai = inputs[i]
        for j in range(m):
            bj = inputs[j]
            assert got_add[k] == (ai + bj) % p
            assert got_sub[k] == (ai - bj) % p
            e = ai * bj * Rinv % p
            assert got_mul[k] == e, (hex(ai), hex(bj))
            assert got_cios[k] == e
            k += 1
    assert run(host_lib, f, 4, x) == [v * v * Rinv % p for v in inputs]
    assert run(host_lib, f, 7, x) == [(-v) % p for v in inputs]
    assert run(host_lib, f, 6, x) == [v * pow(2, -1, p) % p for v in inputs]
    small = ints_to_array([f.to_mont(v) for v in range(0, 25)], n)
    assert run(host_lib, f, 5, small) == [0] + [f.to_mont(pow(v, -1, p)) for v in range(1, 25)]
    # Euclidean inversion (the reference's algorithm): edge values + random, Montgomery in / Montgomery out
    nz = [v for v in inputs if v != 0][:120] + inputs[-60:]
    assert run(host_lib, f, 8, ints_to_array(nz, n)) == [pow(v * Rinv % p, -1, p) * f.R % p for v in nz]
    assert run(host_lib, f, 8, ints_to_array([0], n)) == [0]
