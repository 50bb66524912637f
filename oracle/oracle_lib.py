"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
Arrays are numpy uint64, shape (count, n_limbs) little-endian limbs, Montgomery form unless
stated otherwise - the in-memory representation of the reference's field types
(src/field/tweedledee_base.rs:14-18).
"""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "liboracle.so")

FIELD_LIMBS = {0: 4, 1: 4, 2: 4, 3: 6, 4: 4, 5: 4}
CURVE_BASE_FIELD = {0: 0, 1: 1, 2: 3, 3: 4, 4: 5}
CURVE_SCALAR_FIELD = {0: 1, 1: 0, 2: 2, 3: 5, 4: 4}
CURVE_SCALAR_BITS = {0: 255, 1: 255, 2: 253, 3: 255, 4: 255}


def build(force=False):
    srcs = [os.path.join(_DIR, x) for x in ("plk_oracle.cpp", "plonk_gates.inc", "serialization.inc")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(x) for x in srcs):
        subprocess.check_call(["make", "-C", _DIR, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.orc_fft_precompute.restype = ctypes.c_void_p
        L.orc_fft_precompute.argtypes = [ctypes.c_int, ctypes.c_size_t]
        L.orc_fft_free.argtypes = [ctypes.c_void_p]
        L.orc_fft_table_size.restype = ctypes.c_long
        L.orc_fft_table_size.argtypes = [ctypes.c_void_p]
        L.orc_fft_table_layer.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p]
        L.orc_fft.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
        L.orc_msm_precompute.restype = ctypes.c_void_p
        L.orc_msm_precompute.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_int]
        L.orc_msm_free.argtypes = [ctypes.c_void_p]
        L.orc_msm_table_entry.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_msm_execute.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_field_binop.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.orc_field_unop.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.orc_field_const.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.orc_root_of_unity.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.orc_batch_inverse.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.orc_div2.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_reverse_bits.restype = ctypes.c_uint64
        L.orc_reverse_bits.argtypes = [ctypes.c_uint64, ctypes.c_uint]
        L.orc_curve_generator.argtypes = [ctypes.c_int, ctypes.c_void_p]
        L.orc_curve_op.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint8, ctypes.c_void_p, ctypes.c_uint8,
                                   ctypes.c_void_p, ctypes.c_void_p]
        L.orc_scalar_mul_batch.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_affine_summation.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p]
        L.orc_to_digits.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_gen_bases.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_rand_field.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_void_p]
        L.orc_poly_divide_by_z_h.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_int]
        L.orc_poly_mul.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_int]
        L.orc_poly_to_values_padded.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
        L.orc_gate_constraints.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 8
        L.orc_eval_l_1.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_mds.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
        L.orc_vanishing_points.argtypes = [ctypes.c_int, ctypes.c_size_t] + [ctypes.c_void_p] * 11 + [ctypes.c_int]
        L.orc_field_to_bytes.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.orc_field_from_bytes.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.orc_field_sqrt.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_point_to_bytes.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.orc_point_from_bytes.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


BINOPS = {"add": 0, "sub": 1, "mul": 2}
UNOPS = {"neg": 0, "square": 1, "inverse": 2, "to_canonical": 3, "from_canonical": 4, "double": 5, "triple": 6}
CONSTS = {"ORDER": 0, "R": 1, "R2": 2, "R3": 3, "MU": 4, "TWO": 5, "THREE": 6, "GENERATOR": 7, "T": 8, "NEG_ONE": 9}


def field_binop(field, op, a, b):
    a, b = _u64(a), _u64(b)
    out = np.empty_like(a)
    rc = lib().orc_field_binop(field, BINOPS[op], _p(a), _p(b), _p(out), a.shape[0])
    assert rc == 0
    return out


def field_unop(field, op, a):
    a = _u64(a)
    out = np.empty_like(a)
    rc = lib().orc_field_unop(field, UNOPS[op], _p(a), _p(out), a.shape[0])
    assert rc == 0
    return out


def field_const(field, which):
    out = np.zeros(FIELD_LIMBS[field], dtype=np.uint64)
    assert lib().orc_field_const(field, CONSTS[which], _p(out)) == 0
    return out


def root_of_unity(field, n_power):
    out = np.zeros(FIELD_LIMBS[field], dtype=np.uint64)
    assert lib().orc_root_of_unity(field, n_power, _p(out)) == 0
    return out


def batch_inverse(field, a):
    a = _u64(a)
    out = np.empty_like(a)
    assert lib().orc_batch_inverse(field, _p(a), _p(out), a.shape[0]) == 0
    return out


def div2(limbs):
    a = _u64(limbs)
    out = np.empty_like(a)
    assert lib().orc_div2(a.shape[0], _p(a), _p(out)) == 0
    return out


def reverse_bits(n, num_bits):
    return int(lib().orc_reverse_bits(n, num_bits))


class FftPrecomputation:
    """fft_precompute (src/fft.rs:47-59)."""

    def __init__(self, field, degree):
        self.field = field
        self.h = lib().orc_fft_precompute(field, degree)
        assert self.h

    def size(self):
        return int(lib().orc_fft_table_size(self.h))

    def layer(self, i):
        out = np.zeros((1 << i, FIELD_LIMBS[self.field]), dtype=np.uint64)
        assert lib().orc_fft_table_layer(self.h, i, _p(out)) == 0
        return out

    def _run(self, mode, x, threads):
        x = _u64(x)
        n = x.shape[0]
        n_out = 1 << max(0, (n - 1).bit_length()) if mode == 0 else n
        out = np.zeros((n_out, x.shape[1]), dtype=np.uint64)
        assert lib().orc_fft(self.h, mode, _p(x), n, _p(out), threads) == 0
        return out

    def fft_with_precomputation(self, x, threads=1):
        return self._run(0, x, threads)

    def fft_with_precomputation_power_of_2(self, x, threads=1):
        return self._run(1, x, threads)

    def ifft_with_precomputation_power_of_2(self, x, threads=1):
        return self._run(2, x, threads)

    def __del__(self):
        try:
            lib().orc_fft_free(self.h)
        except Exception:
            pass


def _pow2_ceil(n):
    return 1 << max(0, (n - 1).bit_length())


def poly_divide_by_z_h(field, coeffs, n, threads=1):
    """Polynomial::divide_by_z_h (src/polynomial.rs:330-380)."""
    a = _u64(coeffs).reshape(-1, FIELD_LIMBS[field])
    out = np.zeros((max(a.shape[0], _pow2_ceil(max(a.shape[0], 1))), FIELD_LIMBS[field]), dtype=np.uint64)
    out_len = ctypes.c_size_t(0)
    assert lib().orc_poly_divide_by_z_h(field, _p(a), a.shape[0], n, _p(out), ctypes.byref(out_len), threads) == 0
    return out[: out_len.value].copy()


def poly_mul(field, a, b, threads=1):
    """Polynomial::mul (src/polynomial.rs:208-226)."""
    a = _u64(a).reshape(-1, FIELD_LIMBS[field])
    b = _u64(b).reshape(-1, FIELD_LIMBS[field])
    out = np.zeros((_pow2_ceil(max(a.shape[0] + b.shape[0], 1)), FIELD_LIMBS[field]), dtype=np.uint64)
    out_len = ctypes.c_size_t(0)
    assert lib().orc_poly_mul(field, _p(a), a.shape[0], _p(b), b.shape[0], _p(out), ctypes.byref(out_len), threads) == 0
    return out[: out_len.value].copy()


def poly_to_values_padded(pre, coeffs, threads=1):
    """One polynomial of polynomials_to_values_padded (src/plonk_util.rs:179-190) against an FftPrecomputation."""
    a = _u64(coeffs).reshape(-1, FIELD_LIMBS[pre.field])
    out = np.zeros((pre.size(), FIELD_LIMBS[pre.field]), dtype=np.uint64)
    rc = lib().orc_poly_to_values_padded(pre.h, _p(a), a.shape[0], _p(out), threads)
    if rc != 0:
        raise ValueError("polynomial of length %d does not fit the domain (8x blow-up)" % a.shape[0])
    return out


def curve_generator(curve):
    L = FIELD_LIMBS[CURVE_BASE_FIELD[curve]]
    out = np.zeros((2, L), dtype=np.uint64)
    assert lib().orc_curve_generator(curve, _p(out)) == 0
    return out


def _curve_op(curve, op, a_xy, a_zero, b, b_zero=0):
    L = FIELD_LIMBS[CURVE_BASE_FIELD[curve]]
    a_xy, b = _u64(a_xy), _u64(b)
    out = np.zeros((2, L), dtype=np.uint64)
    oz = np.zeros(1, dtype=np.uint8)
    assert lib().orc_curve_op(curve, op, _p(a_xy), int(a_zero), _p(b), int(b_zero), _p(out), _p(oz)) == 0
    return out, int(oz[0])


def affine_add(curve, a_xy, a_zero, b_xy, b_zero):
    return _curve_op(curve, 0, a_xy, a_zero, b_xy, b_zero)


def affine_double(curve, a_xy, a_zero=0):
    return _curve_op(curve, 1, a_xy, a_zero, a_xy, 0)


def scalar_mul(curve, scalar_mont, a_xy, a_zero=0):
    return _curve_op(curve, 2, a_xy, a_zero, scalar_mont)


def scalar_mul_batch(curve, scalars_mont, base_xy, threads=1):
    """[s_i] base for every scalar (curve_multiplication.rs:5-70 once per scalar) -> ((n, 2, L) points, (n,) zero flags)."""
    L = FIELD_LIMBS[CURVE_BASE_FIELD[curve]]
    s = _u64(scalars_mont).reshape(-1, 4)
    out = np.zeros((s.shape[0], 2, L), dtype=np.uint64)
    oz = np.zeros(s.shape[0], dtype=np.uint8)
    assert lib().orc_scalar_mul_batch(curve, s.shape[0], _p(s), _p(_u64(base_xy)), threads, _p(out), _p(oz)) == 0
    return out, oz


def mul_naive(curve, scalar_mont, a_xy, a_zero=0):
    return _curve_op(curve, 3, a_xy, a_zero, scalar_mont)


def affine_summation(curve, mode, pts_xy, zero=None):
    L = FIELD_LIMBS[CURVE_BASE_FIELD[curve]]
    pts = _u64(pts_xy).reshape(-1, 2, L)
    n = pts.shape[0]
    z = np.zeros(n, dtype=np.uint8) if zero is None else np.ascontiguousarray(zero, dtype=np.uint8)
    out = np.zeros((2, L), dtype=np.uint64)
    oz = np.zeros(1, dtype=np.uint8)
    m = {"pairwise": 0, "batch_inversion": 1, "best": 2}[mode]
    assert lib().orc_affine_summation(curve, m, n, _p(pts), _p(z), _p(out), _p(oz)) == 0
    return out, int(oz[0])


def to_digits(curve, scalar_mont, w):
    s = _u64(scalar_mont)
    out = np.zeros(512, dtype=np.uint64)
    n = ctypes.c_size_t(0)
    assert lib().orc_to_digits(curve, _p(s), w, _p(out), ctypes.byref(n)) == 0
    return [int(v) for v in out[: n.value]]


def gen_bases(curve, n, g0_xy, d_xy):
    L = FIELD_LIMBS[CURVE_BASE_FIELD[curve]]
    out = np.zeros((n, 2, L), dtype=np.uint64)
    assert lib().orc_gen_bases(curve, n, _p(_u64(g0_xy)), _p(_u64(d_xy)), _p(out)) == 0
    return out


def rand_field(field, seed, count):
    out = np.zeros((count, FIELD_LIMBS[field]), dtype=np.uint64)
    assert lib().orc_rand_field(field, seed, count, _p(out)) == 0
    return out


class MsmPrecomputation:
    """msm_precompute (src/curve/curve_msm.rs:27-38): per-generator power tables."""

    def __init__(self, curve, bases_xy, w, zero=None, threads=1):
        self.curve = curve
        self.L = FIELD_LIMBS[CURVE_BASE_FIELD[curve]]
        b = _u64(bases_xy).reshape(-1, 2, self.L)
        self.n = b.shape[0]
        z = np.zeros(self.n, dtype=np.uint8) if zero is None else np.ascontiguousarray(zero, dtype=np.uint8)
        self.h = lib().orc_msm_precompute(curve, self.n, _p(b), _p(z), w, threads)
        assert self.h

    def table_entry(self, i, j):
        out = np.zeros((2, self.L), dtype=np.uint64)
        oz = np.zeros(1, dtype=np.uint8)
        assert lib().orc_msm_table_entry(self.h, i, j, _p(out), _p(oz)) == 0
        return out, int(oz[0])

    def execute(self, scalars, parallel=True, threads=1, want_projective=False):
        """msm_execute (:63) / msm_execute_parallel (:102); returns to_affine() of the result."""
        s = _u64(scalars)
        out = np.zeros((2, self.L), dtype=np.uint64)
        oz = np.zeros(1, dtype=np.uint8)
        proj = np.zeros((3, self.L), dtype=np.uint64)
        rc = lib().orc_msm_execute(self.h, _p(s), s.shape[0], 1 if parallel else 0, threads, _p(out), _p(oz), _p(proj))
        if rc == -2:
            raise AssertionError("powers_per_generator.len() != scalars.len()")  # curve_msm.rs:67,106
        assert rc == 0
        if want_projective:
            return out, int(oz[0]), proj
        return out, int(oz[0])

    def __del__(self):
        try:
            lib().orc_msm_free(self.h)
        except Exception:
            pass


# ---- the Plonk quotient numerator (plonk_gates.inc) ----
def gate_constraints(field, gate, k, l, r, b, zeta, a, unfiltered=False):
    """gate < 0: evaluate_all_constraints (gates/mod.rs:46-125); else Gate::evaluate_filtered / evaluate_unfiltered of that gate."""
    k, l, r, b, zeta, a = (_u64(x) for x in (k, l, r, b, zeta, a))
    out = np.zeros((8, 4), dtype=np.uint64)
    cnt = ctypes.c_size_t(0)
    assert lib().orc_gate_constraints(field, gate, 1 if unfiltered else 0, _p(k), _p(l), _p(r), _p(b), _p(zeta), _p(a), _p(out), ctypes.byref(cnt)) == 0
    return out[: cnt.value]


def eval_l_1(field, n, x):
    x = _u64(x)
    out = np.zeros(4, dtype=np.uint64)
    assert lib().orc_eval_l_1(field, n, _p(x), _p(out)) == 0
    return out


def mds(field, n, r, c):
    out = np.zeros(4, dtype=np.uint64)
    assert lib().orc_mds(field, n, r, c, _p(out)) == 0
    return out


def vanishing_points(field, degree, constants, wires, s_sigma, z, k_is, alpha, beta, gamma, zeta, a, threads=1):
    """plonk.rs:392-453: constants (6, 8n, 4), wires (9, 8n, 4), s_sigma (6, 8n, 4), z (8n, 4), k_is (6, 4) -> (8n, 4)."""
    arrs = [_u64(x) for x in (constants, wires, s_sigma, z, k_is, alpha, beta, gamma, zeta, a)]
    out = np.zeros((8 * degree, 4), dtype=np.uint64)
    assert lib().orc_vanishing_points(field, degree, *[_p(x) for x in arrs], _p(out), threads) == 0
    return out


# ---- canonical byte encodings (serialization.inc) ----
def field_to_bytes(field, x):
    x = _u64(x)
    L = lib().orc_field_limbs(field)
    n = x.size // L
    out = np.zeros(n * L * 8, dtype=np.uint8)
    assert lib().orc_field_to_bytes(field, _p(x), n, _p(out)) == 0
    return out.reshape(n, L * 8)


def field_from_bytes(field, b):
    """returns (elements, number of `Out of range` records)"""
    b = np.ascontiguousarray(b, dtype=np.uint8)
    L = lib().orc_field_limbs(field)
    n = b.size // (L * 8)
    out = np.zeros((n, L), dtype=np.uint64)
    bad = lib().orc_field_from_bytes(field, _p(b), n, _p(out))
    assert bad >= 0
    return out, bad


def field_sqrt(field, x):
    x = _u64(x)
    out = np.zeros_like(x)
    rc = lib().orc_field_sqrt(field, _p(x), _p(out))
    assert rc >= 0
    return out if rc else None


def point_to_bytes(curve, xy, zero=None):
    xy = _u64(xy)
    L = xy.shape[-1]
    n = xy.size // (2 * L)
    z = np.ascontiguousarray(zero, dtype=np.uint8) if zero is not None else np.zeros(n, dtype=np.uint8)
    out = np.zeros(n * (1 + L * 8), dtype=np.uint8)
    assert lib().orc_point_to_bytes(curve, _p(xy), _p(z), n, _p(out)) == 0
    return out.reshape(n, 1 + L * 8)


def point_from_bytes(curve, b, limbs):
    """returns (xy (n, 2, L), zero flags, status: 0 ok / 1 Out of range / 2 Invalid x coordinate)"""
    b = np.ascontiguousarray(b, dtype=np.uint8)
    n = b.size // (1 + limbs * 8)
    xy = np.zeros((n, 2, limbs), dtype=np.uint64)
    zero = np.zeros(n, dtype=np.uint8)
    status = np.zeros(n, dtype=np.uint8)
    assert lib().orc_point_from_bytes(curve, _p(b), n, _p(xy), _p(zero), _p(status)) == 0
    return xy, zero, status


# ---- one round of the inner-product argument, composed from the restated primitives (halo.rs:63-124) ----
def _ip(field, a, b):
    """Field::inner_product (field.rs:213-221)"""
    acc = np.zeros(a.shape[1], dtype=np.uint64)
    prod = field_binop(field, "mul", a, b)
    for row in prod:
        acc = field_binop(field, "add", acc.reshape(1, -1), row.reshape(1, -1))[0]
    return acc


def halo_round_lr(curve, scalar_field, halo_a, halo_b, halo_g, pedersen_h, u_prime, l_blinding, r_blinding, threads=4):
    """L_j = msm_parallel(a_lo, g_hi, 8) + [l_j] H + [<a_lo, b_hi>] U' and R_j likewise (halo.rs:86-93): ((2, 2, L), [zero, zero])."""
    m = halo_a.shape[0] // 2
    outs, zeros = [], []
    for a_half, b_half, g_half, blind in ((halo_a[:m], halo_b[m:], halo_g[m:], l_blinding), (halo_a[m:], halo_b[:m], halo_g[:m], r_blinding)):
        xy, z = MsmPrecomputation(curve, g_half, 8, threads=threads).execute(a_half, parallel=True, threads=threads)
        t1, z1 = scalar_mul(curve, blind, pedersen_h)
        t2, z2 = scalar_mul(curve, _ip(scalar_field, a_half, b_half), u_prime)
        xy, z = affine_add(curve, xy, z, t1, z1)
        xy, z = affine_add(curve, xy, z, t2, z2)
        outs.append(xy)
        zeros.append(z)
    return np.stack(outs), zeros


def halo_round_fold(curve, scalar_field, halo_a, halo_b, halo_g, u_j, u_j_inv):
    """halo.rs:117-123: the folded (halo_a, halo_b, halo_g, zero flags)."""
    m = halo_a.shape[0] // 2
    scale = lambda s, v: field_binop(scalar_field, "mul", np.tile(_u64(s), (v.shape[0], 1)), v)
    a2 = field_binop(scalar_field, "add", scale(u_j_inv, halo_a[m:]), scale(u_j, halo_a[:m]))
    b2 = field_binop(scalar_field, "add", scale(u_j_inv, halo_b[:m]), scale(u_j, halo_b[m:]))
    g2, gz = [], []
    for i in range(m):
        p1, z1 = scalar_mul(curve, u_j_inv, halo_g[i])
        p2, z2 = scalar_mul(curve, u_j, halo_g[m + i])
        xy, z = affine_add(curve, p1, z1, p2, z2)
        g2.append(xy)
        gz.append(z)
    return a2, b2, np.stack(g2), np.array(gz, dtype=np.uint8)
