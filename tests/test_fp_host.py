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
FIELDS = [br.TWEEDLEDEE_BASE, br.TWEEDLEDUM_BASE, br.BLS12_377_SCALAR, br.BLS12_377_BASE, br.PALLAS_BASE, br.VESTA_BASE]


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
    # the division-step inversion (fe_inv_safegcd): same contract, plus the values that stress its sign handling
    edge = [1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, 1 << 30, (1 << 30) - 1, 1 << 60, (1 << 255) % p, f.R % p, f.R2 % p]
    edge = [v for v in edge if v % p]
    assert run(host_lib, f, 18, ints_to_array(nz + edge, n)) == [pow(v * Rinv % p, -1, p) * f.R % p for v in nz + edge]
    assert run(host_lib, f, 18, ints_to_array([0], n)) == [0]
    # its data-dependent form for one lane (fe_inv_safegcd_var: runs of division steps at once): every input of this test + powers of two
    allv = [v for v in inputs if v % p] + edge + [(1 << k) % p for k in range(0, 32 * n, 7)] + [(p - (1 << k)) % p for k in range(0, 32 * n - 2, 11)]
    allv = [v for v in allv if v % p]
    assert run(host_lib, f, 28, ints_to_array(allv, n)) == [pow(v * Rinv % p, -1, p) * f.R % p for v in allv]
    assert run(host_lib, f, 28, ints_to_array([0], n)) == [0]


@pytest.mark.parametrize("f", FIELDS, ids=lambda f: f.name)
def test_lazy_29bit_arithmetic_on_host(host_lib, f):
    """fz.cuh (the working form of the hot kernels): 29-bit limbs, R' = 2^(29 NZ), lazily reduced."""
    p, n = f.p, f.n_limbs
    nz = (64 * n + 28) // 29
    Rp_inv = pow(1 << (29 * nz), -1, p)
    inputs = reference_test_inputs(p)
    inputs += [synth.to_int(r) for r in synth.rand_field(f.field_id, 77, 150)]
    m = len(inputs)
    x = ints_to_array(inputs, n)
    assert run(host_lib, f, 10, x) == inputs
    assert run(host_lib, f, 12, x) == [v * v * Rp_inv % p for v in inputs]
    assert run(host_lib, f, 14, x) == [v * pow(2, 2 * 29 * nz - 64 * n, p) * Rp_inv % p for v in inputs]
    assert run(host_lib, f, 15, x) == [v * pow(2, 64 * n, p) * Rp_inv % p for v in inputs]
    ia = np.repeat(np.arange(m), m)
    ib = np.tile(np.arange(m), m)
    a, b = x[ia], x[ib]
    got_mul = run(host_lib, f, 11, a, b)
    got_lin = run(host_lib, f, 13, a, b)
    got_chain = run(host_lib, f, 16, a, b)
    got_zero = run(host_lib, f, 17, a, b)
    got_ma2 = run(host_lib, f, 24, a, b)   # fz_mul_add2 at the edge of its column bound
    got_ms2 = run(host_lib, f, 25, a, b)   # fz_mul_sub2 on ordinary operands
    got_red = run(host_lib, f, 19, a, b) if n == 4 else None  # fz_reduce_small: the NTT fields
    got_nc = [run(host_lib, f, 20 + t, a, b) for t in range(4)] if n == 4 else None  # carry-free radix-4 steps of the NTT tile
    k = 0
    for i in range(m):
        ai = inputs[i]
        for j in range(m):
            bj = inputs[j]
            assert got_mul[k] == ai * bj * Rp_inv % p
            assert got_lin[k] == (ai - bj) % p
            u = (ai - bj) ** 2 * Rp_inv
            v = (ai + bj) * bj * Rp_inv
            assert got_chain[k] == (u - 2 * v) * (ai - bj) * Rp_inv % p
            assert got_zero[k] == (1 if ai * bj % p == 0 else 0)
            assert got_ms2[k] == (ai * bj - bj * (ai + bj)) * Rp_inv % p
            # the limb patterns of harness op 24, rebuilt from the same input words
            xw = [(ai >> (32 * t)) & 0xFFFFFFFF for t in range(3)]
            yw = [(bj >> (32 * t)) & 0xFFFFFFFF for t in range(2)]
            small = nz <= 9
            la = (1 << 29) + 7 if small else (1 << 29) + (1 << 27)
            lb, lc, ld, top = (1 << 31 if small else la), (1 << 30 if small else la), (1 << 29) - 1, (1 << 25 if small else 3)
            def val(limb, tl):
                return sum(limb << (29 * t) for t in range(nz - 1)) + (tl << (29 * (nz - 1)))
            A = val(la - (xw[0] & 7), top + (xw[1] & 0xFFFF))
            B = val(lb - (yw[0] & 0xFF), top)
            C = val(lc - (xw[2] & 0xFF), top)
            Dv = val(ld - (yw[1] & 0xFF), (1 << 22) if small else 1)
            assert got_ma2[k] == (A * B + C * Dv) * Rp_inv % p, (hex(ai), hex(bj))
            if got_red is not None:
                assert got_red[k] == (ai + 13 * bj) % p
            if got_nc is not None:
                y3 = (bj - ai) * bj * Rp_inv
                o = [2 * (ai + bj), (ai - bj) + y3, 0, (ai - bj) - y3]
                x1, x3 = o[2] * bj * Rp_inv, o[0] * bj * Rp_inv
                z0, z1 = o[3] + x1, o[3] - x1
                z2, z3 = (o[1] + x3) * bj * Rp_inv, (o[1] - x3) * ai * Rp_inv
                want = [z0 + z2, z1 + z3, z0 - z2, z1 - z3]
                for t in range(4):
                    assert got_nc[t][k] == want[t] % p, (t, hex(ai), hex(bj))
            k += 1
    # the value p itself (== 0 mod p) as an operand exercises the second branch of the zero test
    pw = ints_to_array([p] * 3, n)
    yw = ints_to_array([1, 12345, p - 1], n)
    assert run(host_lib, f, 17, pw, yw) == [1, 1, 1]


@pytest.mark.parametrize("c", [br.TWEEDLEDEE, br.TWEEDLEDUM, br.BLS12_377], ids=lambda c: c.name)
def test_lazy_xyzz_accumulation_on_host(host_lib, c):
    """ecz.cuh: the bucket-accumulation inner loop (mixed XYZZ additions on lazy coordinates), incl. the
    exceptional cases of curve_adds.rs:50-90 (identity accumulator, P + P, P + (-P)) and long chains."""
    import random
    f = c.base
    n = f.n_limbs
    nz = (64 * n + 28) // 29
    Rp = pow(2, 29 * nz, f.p)
    host_lib.ecz_host_sum.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    G = (c.gx, c.gy)
    rng = random.Random(5)
    pts = [br.ec_mul(c, rng.randrange(1, 1 << 64), G) for _ in range(40)]

    def run_sum(points, negs, fn="ecz_host_sum"):
        getattr(host_lib, fn).argtypes = host_lib.ecz_host_sum.argtypes
        xs = ints_to_array([P[0] * Rp % f.p for P in points], n)
        ys = ints_to_array([P[1] * Rp % f.p for P in points], n)
        ng = np.array(negs, dtype=np.uint8)
        out = np.zeros(4 * n + 1, dtype=np.uint64)
        out32 = np.zeros(4 * 2 * n + 2, dtype=np.uint32)
        assert getattr(host_lib, fn)(f.field_id, len(points), xs.ctypes.data, ys.ctypes.data, ng.ctypes.data, out32.ctypes.data) == 0
        inf = int(out32[4 * 2 * n])
        assert inf in (0, 1), "a limb bound of the lazy accumulation was exceeded (flag %d)" % inf
        if inf:
            return None
        w = out32[: 4 * 2 * n].reshape(4, 2 * n)
        X, Y, ZZ, ZZZ = [sum(int(v) << (32 * i) for i, v in enumerate(row)) for row in w]
        assert ZZ % f.p != 0 and pow(ZZ, 3, f.p) == ZZZ * ZZZ * Rp % f.p  # ZZ^3 = ZZZ^2 (in R'-form: extra R')
        return (X * pow(ZZ, -1, f.p) % f.p, Y * pow(ZZZ, -1, f.p) % f.p)

    def expect(points, negs):
        acc = None
        for P, ng in zip(points, negs):
            acc = br.ec_add(c, acc, br.ec_neg(c, P) if ng else P)
        return acc

    cases = [
        ([pts[0]], [0]), ([pts[0]], [1]),
        ([pts[0], pts[0]], [0, 0]),                     # doubling branch
        ([pts[0], pts[0]], [0, 1]),                     # P + (-P) = identity
        ([pts[0], pts[0], pts[1]], [0, 1, 0]),          # identity accumulator picks up the next point
        ([pts[0], pts[1], br.ec_add(c, pts[0], pts[1])], [0, 0, 0]),   # acc == operand after two adds -> double
        ([pts[0], pts[1], br.ec_add(c, pts[0], pts[1])], [0, 0, 1]),   # acc == -operand -> identity
        (pts, [rng.randrange(2) for _ in pts]),
        (pts * 8, [rng.randrange(2) for _ in range(len(pts) * 8)]),    # 320-long chain: bounds hold
    ]
    for points, negs in cases:
        assert run_sum(points, negs) == expect(points, negs)
        # balanced tree of full XYZZ additions (the reduction kernels): same group element
        assert run_sum(points, negs, "ecz_host_tree") == expect(points, negs)
    # trees that hit a + a (doubling) and a + (-a) at inner nodes
    assert run_sum([pts[0], pts[1], pts[0], pts[1]], [0, 0, 0, 0], "ecz_host_tree") == expect([pts[0], pts[1]] * 2, [0] * 4)
    assert run_sum([pts[0], pts[1], pts[0], pts[1]], [0, 0, 1, 1], "ecz_host_tree") is None


@pytest.mark.parametrize("c", [br.TWEEDLEDEE, br.TWEEDLEDUM, br.PALLAS, br.VESTA], ids=lambda c: c.name)
def test_glv_split_on_host(host_lib, c):
    """glv.cuh: k = k1 + k2 lambda (mod r) with half-length k1, k2, and [k] P = [k1] P + [k2] phi(P), phi((x, y)) = (beta x, y)
    (curve.rs:140-149 endomorphism; the reference's own test: *_curve.rs test_endomorphism_*)."""
    import random, re
    r, p = c.scalar.p, c.base.p
    txt = open(os.path.join(os.path.dirname(__file__), "..", "plonky_amd", "csrc", "glv_params.cuh")).read()
    blk = txt[txt.index("struct %sGlv" % c.name):]
    blk = blk[:blk.index("\n};")]

    def const(name, n):
        m = re.search(r"%s\[%d\] = \{([^}]*)\}" % (name, n), blk)
        return sum(int(w.strip().rstrip("u"), 16) << (32 * i) for i, w in enumerate(m.group(1).split(",")))

    beta, lam = const("BETA", 8), const("LAMBDA", 8)
    assert pow(beta, 3, p) == 1 and beta != 1 and pow(lam, 3, r) == 1 and lam != 1
    G = (c.gx, c.gy)
    assert br.ec_mul(c, lam, G) == (beta * G[0] % p, G[1])
    rng = random.Random(99)
    ks = [0, 1, 2, r - 1, r - 2, r // 2, r // 2 + 1, lam, r - lam] + [rng.randrange(r) for _ in range(3000)]
    k = ints_to_array(ks, 4)
    k1 = np.zeros_like(k)
    k2 = np.zeros_like(k)
    host_lib.glv_host_split.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    assert host_lib.glv_host_split(c.curve_id, k.ctypes.data, k1.ctypes.data, k2.ctypes.data, len(ks)) == 0

    def signed(row):
        v = synth.to_int(row)
        return -(v & ((1 << 255) - 1)) if v >> 255 else v

    P = br.ec_mul(c, 424242, G)
    phiP = (beta * P[0] % p, P[1])
    neg = lambda Q: (Q[0], (-Q[1]) % p)
    for i, kv in enumerate(ks):
        a, b = signed(k1[i]), signed(k2[i])
        assert abs(a) < 1 << 130 and abs(b) < 1 << 130
        assert (a + b * lam - kv) % r == 0
        if i < 40:
            t1 = br.ec_mul(c, abs(a), P if a >= 0 else neg(P)) if a else None
            t2 = br.ec_mul(c, abs(b), phiP if b >= 0 else neg(phiP)) if b else None
            assert br.ec_add(c, t1, t2) == (br.ec_mul(c, kv, P) if kv else None)
