"""Host-side mirror of the reference's hot-path interface, same names and argument meaning.

  reference (Rust, generic over F / C)                        here
  ------------------------------------------------------------------------------------------
  fft_precompute::<F>(degree)              fft.rs:47      fft_precompute(field, degree)
  fft_with_precomputation(c, &pre)         fft.rs:61      fft_with_precomputation(c, pre)
  fft_with_precomputation_power_of_2       fft.rs:103     fft_with_precomputation_power_of_2(c, pre)
  ifft_with_precomputation_power_of_2      fft.rs:82      ifft_with_precomputation_power_of_2(p, pre)
  fft(c)                                   fft.rs:42      fft(field, c)
  msm_precompute::<C>(generators, w)       curve_msm.rs:27   msm_precompute(curve, generators, w)
  msm_execute(&pre, scalars)               curve_msm.rs:63   msm_execute(pre, scalars)
  msm_execute_parallel(&pre, scalars)      curve_msm.rs:102  msm_execute_parallel(pre, scalars)
  msm_parallel(scalars, generators, w)     curve_msm.rs:54   msm_parallel(curve, scalars, generators, w)

The generic type parameter becomes an explicit field / curve id.  Field elements are numpy
uint64 arrays of shape (n, 4) -- exactly the reference's `limbs` arrays (Montgomery form);
generators are (n, 2, L) arrays (x, y) plus an optional zero-flag vector.  MSM results are the
affine point `(xy, zero)` = ProjectivePoint::to_affine() of the reference's return value.

Error behaviour follows the reference: contract violations raise (the reference panics):
length mismatch (curve_msm.rs:67,106) -> AssertionError; non power of two (util.rs:17) ->
AssertionError; beyond the 2-adicity (field.rs:430) -> AssertionError.

Every function calls the HIP library through the C ABI; there is no CPU implementation here.
"""
import ctypes

import numpy as np

from . import lib as _lib

TWEEDLEDEE_BASE, TWEEDLEDUM_BASE, BLS12_377_SCALAR, BLS12_377_BASE, PALLAS_BASE, VESTA_BASE = 0, 1, 2, 3, 4, 5
TWEEDLEDEE, TWEEDLEDUM, BLS12_377, PALLAS, VESTA = 0, 1, 2, 3, 4

_FIELD_LIMBS = {0: 4, 1: 4, 2: 4, 3: 6, 4: 4, 5: 4}
_CURVE_LIMBS = {0: 4, 1: 4, 2: 6, 3: 4, 4: 4}
_TWO_ADICITY = {0: 34, 1: 33, 2: 47, 3: 46, 4: 32, 5: 32}
CURVE_SCALAR_FIELD = {0: 1, 1: 0, 2: 2, 3: 5, 4: 4}
CURVE_BASE_FIELD = {0: 0, 1: 1, 2: 3, 3: 4, 4: 5}


def log2_ceil(n):  # util.rs:2-9
    r = 0
    while (1 << r) < n:
        r += 1
    return r


def log2_strict(n):  # util.rs:12-19 (panics when n is not a power of two)
    r = log2_ceil(n)
    assert n == 1 << r, "Not a power of two"
    return r


def init_devices(n_devices=0):
    """Run the host-pointer entry points over several GPUs from this one process (plk_init_devices): 0 = PLK_NGPU or every
    visible device; with PLK_VIRTUAL_DEVICES=k, k logical devices on one physical GPU.  Returns the number of logical devices."""
    L = _lib.load()
    _lib.check(L.plk_init_devices(int(n_devices)))
    return int(L.plk_device_count())


def device_count():
    return int(_lib.load().plk_device_count())


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _elems(field, a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    L = _FIELD_LIMBS[field]
    return a.reshape(-1, L)


class FftPrecomputation:
    """fft.rs:28-40.  The reference stores host tables (subgroups_rev); the device tables live
    in the library's cache keyed by (device, field, log n), so this object only records the
    size -- it stays plain data, as the reference's serde-able struct is."""

    def __init__(self, field, degree_pow):
        self.field = field
        self.degree_pow = degree_pow

    def size(self):
        return 1 << self.degree_pow


def fft_precompute(field, degree):
    degree_pow = log2_ceil(degree)
    assert degree_pow <= _TWO_ADICITY[field], "n_power <= TWO_ADICITY"  # field.rs:430
    _lib.check(_lib.load().plk_ntt_precompute(field, degree_pow))
    return FftPrecomputation(field, degree_pow)


def fft_precompute_table(field, degree):
    """The CONTENTS of the reference's FftPrecomputation (fft.rs:28-59), built on the device: the list subgroups_rev[i],
    i = 0 ..= log2_ceil(degree), each a (2^i, L) array of the bit-reversed powers of primitive_root_of_unity(i) (L = 4 limbs; 6 for
    Bls12377Base)."""
    degree_pow = log2_ceil(degree)
    assert degree_pow <= _TWO_ADICITY[field], "n_power <= TWO_ADICITY"  # field.rs:430
    flat = np.empty(((2 << degree_pow) - 1, _FIELD_LIMBS[field]), dtype=np.uint64)
    _lib.check(_lib.load().plk_ntt_precompute_table(field, degree_pow, _ptr(flat)))
    return [flat[(1 << i) - 1: (2 << i) - 1] for i in range(degree_pow + 1)]


def _ntt(field, log_n, inverse, x):
    out = np.empty_like(x)
    _lib.check(_lib.load().plk_ntt(field, log_n, 1 if inverse else 0, _ptr(x), _ptr(out)))
    return out


def fft_with_precomputation_power_of_2(coefficients, precomputation):
    x = _elems(precomputation.field, coefficients)
    degree_pow = log2_strict(x.shape[0])
    # fft.rs:107-111 debug_assert_eq!: release builds accept any table at least as large (Appendix A.7)
    assert degree_pow <= _TWO_ADICITY[precomputation.field]
    return _ntt(precomputation.field, degree_pow, False, x)


def ifft_with_precomputation_power_of_2(points, precomputation):
    x = _elems(precomputation.field, points)
    degree_pow = log2_strict(x.shape[0])
    assert degree_pow <= _TWO_ADICITY[precomputation.field]
    return _ntt(precomputation.field, degree_pow, True, x)


def fft_with_precomputation(coefficients, precomputation):
    x = _elems(precomputation.field, coefficients)
    degree = x.shape[0]
    log_n = log2_ceil(degree)
    if degree == 1 << log_n:
        return fft_with_precomputation_power_of_2(x, precomputation)
    out = np.empty((1 << log_n, x.shape[1]), dtype=np.uint64)
    _lib.check(_lib.load().plk_ntt_padded(precomputation.field, log_n, _ptr(x), degree, _ptr(out)))
    return out


def fft(field, coefficients):
    x = _elems(field, coefficients)
    return fft_with_precomputation(x, fft_precompute(field, x.shape[0]))


def fft_batch(field, polys, inverse=False):
    """`batch` independent power-of-two transforms in one call (the par_iter over the 9 wire
    polynomials, plonk_util.rs:169-190).  polys: (batch, n, 4)."""
    polys = np.ascontiguousarray(polys, dtype=np.uint64)
    batch, n = polys.shape[0], polys.shape[1]
    log_n = log2_strict(n)
    out = np.empty_like(polys)
    ins = (ctypes.c_void_p * batch)(*[polys[b].ctypes.data for b in range(batch)])
    outs = (ctypes.c_void_p * batch)(*[out[b].ctypes.data for b in range(batch)])
    _lib.check(_lib.load().plk_ntt_batch(field, log_n, 1 if inverse else 0, batch, ins, outs))
    return out


# ---- polynomial callers of the NTT (src/polynomial.rs, src/plonk_util.rs) ----
def _pow2_ceil(n):
    return 1 << log2_ceil(max(n, 1))


def polynomial_divide_by_z_h(field, coeffs, n):
    """Polynomial::divide_by_z_h (polynomial.rs:330-380): coeffs / (X^n - 1).  Returns the
    2^ceil(log2(degree + 1)) untrimmed coefficients the reference returns; the zero polynomial
    comes back unchanged."""
    x = _elems(field, coeffs)
    assert n >= 1
    cap = max(x.shape[0], _pow2_ceil(x.shape[0]))
    out = np.zeros((cap, x.shape[1]), dtype=np.uint64)
    out_len = ctypes.c_size_t(0)
    _lib.check(_lib.load().plk_poly_divide_by_z_h(field, _ptr(x), x.shape[0], n, _ptr(out), cap, ctypes.byref(out_len)))
    return out[: out_len.value].copy()


def polynomial_mul(field, a, b):
    """Polynomial::mul (polynomial.rs:208-226)."""
    x, y = _elems(field, a), _elems(field, b)
    cap = _pow2_ceil(x.shape[0] + y.shape[0])
    out = np.zeros((cap, x.shape[1]), dtype=np.uint64)
    out_len = ctypes.c_size_t(0)
    _lib.check(_lib.load().plk_poly_mul(field, _ptr(x), x.shape[0], _ptr(y), y.shape[0], _ptr(out), cap, ctypes.byref(out_len)))
    return out[: out_len.value].copy()


def polynomials_to_values_padded(polys, precomputation):
    """plonk_util.rs:179-190: every polynomial padded to 8x its length, then evaluated on the
    precomputation's domain (eval_domain pads further when the domain is larger).  polys: sequence of
    (len_b, 4) arrays; returns (batch, domain, 4)."""
    field = precomputation.field
    ps = [_elems(field, p) for p in polys]
    n = precomputation.size()
    for p in ps:
        assert p.shape[0] * 8 <= n, "fft.rs:107-111: the table is too small for the padded polynomial"
    batch = len(ps)
    out = np.empty((batch, n, 4), dtype=np.uint64)
    if batch == 0:
        return out
    ins = (ctypes.c_void_p * batch)(*[p.ctypes.data for p in ps])
    lens = (ctypes.c_size_t * batch)(*[p.shape[0] for p in ps])
    outs = (ctypes.c_void_p * batch)(*[out[b].ctypes.data for b in range(batch)])
    _lib.check(_lib.load().plk_ntt_padded_batch(field, precomputation.degree_pow, batch, ins, lens, outs))
    return out


def values_to_polynomials(values_vec, precomputation):
    """plonk_util.rs:169-177: Polynomial::from_evaluations over a batch."""
    return fft_batch(precomputation.field, values_vec, inverse=True)


class MsmPrecomputation:
    """curve_msm.rs:16-25.  Owns a device context holding the generators and the window tables
    [2^(c j)] G_i.  `w` is kept because it is part of the reference struct; the device window c
    is a tuning choice (the result does not depend on it)."""

    def __init__(self, curve, ctx, n, w):
        self.curve = curve
        self._ctx = ctx
        self.n = n
        self.w = w

    @property
    def window(self):
        return int(_lib.load().plk_msm_ctx_window(self._ctx))

    def __len__(self):
        return self.n

    def free(self):
        if self._ctx:
            _lib.load().plk_msm_free(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _points(curve, generators):
    L = _CURVE_LIMBS[curve]
    g = np.ascontiguousarray(generators, dtype=np.uint64).reshape(-1, 2, L)
    return g


MSM_TABLE_FREE = 1  # PLK_MSM_TABLE_FREE


def msm_precompute(curve, generators, w, zero=None, device_window=0, table_free=False):
    """generators: (n, 2, L) affine x,y Montgomery limbs; zero: optional n flags (AffinePoint.zero).
    The reference takes ProjectivePoints and only ever passes normalised generators
    (circuit_builder.rs:1127-1133); the shim in INTEGRATION.md reads powers_per_generator[i][0].
    table_free: no window tables on the device (generators used once: msm_parallel, the IPA rounds)."""
    g = _points(curve, generators)
    n = g.shape[0]
    z = None
    if zero is not None:
        z = np.ascontiguousarray(zero, dtype=np.uint8)
        assert z.shape[0] == n
    ctx = ctypes.c_void_p()
    _lib.check(_lib.load().plk_msm_precompute_ex(curve, n, _ptr(g), _ptr(z) if z is not None else None, device_window,
                                                 MSM_TABLE_FREE if table_free else 0, ctypes.byref(ctx)))
    return MsmPrecomputation(curve, ctx, n, w)


def msm_execute_parallel(precomputation, scalars):
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    # assert_eq!(precomputation.powers_per_generator.len(), scalars.len())  curve_msm.rs:106
    assert s.shape[0] == precomputation.n, "powers_per_generator.len() != scalars.len()"
    L = _CURVE_LIMBS[precomputation.curve]
    out = np.zeros((2, L), dtype=np.uint64)
    oz = np.zeros(1, dtype=np.uint8)
    _lib.check(_lib.load().plk_msm_execute(precomputation._ctx, _ptr(s), s.shape[0], _ptr(out), _ptr(oz)))
    return out, int(oz[0])


# msm_execute (serial, curve_msm.rs:63) computes the same group element as msm_execute_parallel.
msm_execute = msm_execute_parallel


def msm_execute_parallel_projective(precomputation, scalars):
    """msm_execute_parallel with its OWN return type (curve_msm.rs:102-157 returns the ProjectivePoint `y`, not normalised):
    ((3, L) x | y | z Montgomery limbs, zero flag).  ProjectivePoint::to_affine / batch_to_affine (curve.rs:206-232) gives the unique point."""
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    assert s.shape[0] == precomputation.n, "powers_per_generator.len() != scalars.len()"
    L = _CURVE_LIMBS[precomputation.curve]
    out = np.zeros((3, L), dtype=np.uint64)
    oz = np.zeros(1, dtype=np.uint8)
    _lib.check(_lib.load().plk_msm_execute_projective(precomputation._ctx, _ptr(s), s.shape[0], _ptr(out), _ptr(oz)))
    return out, int(oz[0])


def msm_execute_batch(precomputation, scalar_vectors):
    """commit_polynomials (plonk_util.rs:215-231): several scalar vectors, same generators."""
    sv = np.ascontiguousarray(scalar_vectors, dtype=np.uint64)
    batch = sv.shape[0]
    sv = sv.reshape(batch, -1, 4)
    assert sv.shape[1] == precomputation.n, "powers_per_generator.len() != scalars.len()"
    L = _CURVE_LIMBS[precomputation.curve]
    out = np.zeros((batch, 2, L), dtype=np.uint64)
    oz = np.zeros(batch, dtype=np.uint8)
    ptrs = (ctypes.c_void_p * batch)(*[sv[b].ctypes.data for b in range(batch)])
    _lib.check(_lib.load().plk_msm_execute_batch(precomputation._ctx, batch, ptrs, sv.shape[1], _ptr(out), _ptr(oz)))
    return out, oz


def commitment_precompute(curve, generators, blinding_point, w, device_window=0):
    """Device context for PolynomialCommitment::coeffs_vec_to_commitments (poly_commit.rs:31-66): the blinding term
    [r] H is the (n+1)-th term of the same MSM, so the context simply holds H after the n generators."""
    g = _points(curve, generators)
    h = _points(curve, blinding_point)
    assert h.shape[0] == 1
    return msm_precompute(curve, np.concatenate([g, h]), w, device_window=device_window)


def coeffs_vec_to_commitments(precomputation, coefficients_vec, blinding_factors):
    """poly_commit.rs:51-66: one commitment per coefficient vector, pedersen_hash(coeffs) + [r] H, already normalised
    (the reference's batch_to_affine).  blinding_factors: one Montgomery scalar per vector (zeros when blinding is off).
    precomputation: from commitment_precompute.  Returns (points (k, 2, L), zero flags (k,))."""
    cv = np.ascontiguousarray(coefficients_vec, dtype=np.uint64)
    k = cv.shape[0]
    cv = cv.reshape(k, -1, 4)
    r = np.ascontiguousarray(blinding_factors, dtype=np.uint64).reshape(k, 1, 4)
    assert cv.shape[1] + 1 == precomputation.n, "coefficients.len() must equal the number of generators (curve_msm.rs:106)"
    return msm_execute_batch(precomputation, np.concatenate([cv, r], axis=1))


def msm_parallel(curve, scalars, generators, w, zero=None):
    """curve_msm.rs:54-61: precompute + execute for generators that are used once -> no device tables."""
    pre = msm_precompute(curve, generators, w, zero=zero, table_free=True)
    try:
        return msm_execute_parallel(pre, scalars)
    finally:
        pre.free()


def msm_precompute_table(curve, generators, w, zero=None):
    """The CONTENTS of the reference's MsmPrecomputation (curve_msm.rs:16-52): powers_per_generator[i][j] = [2^(w j)] G_i,
    j < ceil(ScalarField::BITS / w).  Returns (table (n, digits, 2, L), zero flags (n, digits))."""
    g = _points(curve, generators)
    n = g.shape[0]
    L = _CURVE_LIMBS[curve]
    digits = int(_lib.load().plk_msm_table_digits(curve, w))
    assert digits > 0
    z = None if zero is None else np.ascontiguousarray(zero, dtype=np.uint8)
    out = np.zeros((n, digits, 2, L), dtype=np.uint64)
    oz = np.zeros((n, digits), dtype=np.uint8)
    _lib.check(_lib.load().plk_msm_precompute_table(curve, n, _ptr(g), _ptr(z) if z is not None else None, w, _ptr(out), _ptr(oz)))
    return out, oz


def fold_generators(curve, g_lo, g_hi, scalar_lo, scalar_hi, lo_zero=None, hi_zero=None):
    """The generator fold of an IPA round (halo.rs:119-123): out_i = [scalar_lo] g_lo_i + [scalar_hi] g_hi_i, where the
    reference calls msm_parallel(&[u_inv, u], &[g_lo_i, g_hi_i], 4) per pair.  Points (m, 2, L) affine Montgomery
    limbs, scalars (4,) Montgomery limbs in the curve's scalar field.  Returns (out (m, 2, L), zero flags (m,))."""
    lo, hi = _points(curve, g_lo), _points(curve, g_hi)
    m = lo.shape[0]
    assert hi.shape[0] == m
    a = np.ascontiguousarray(scalar_lo, dtype=np.uint64).reshape(4)
    b = np.ascontiguousarray(scalar_hi, dtype=np.uint64).reshape(4)
    lz = None if lo_zero is None else np.ascontiguousarray(lo_zero, dtype=np.uint8)
    hz = None if hi_zero is None else np.ascontiguousarray(hi_zero, dtype=np.uint8)
    out = np.zeros_like(lo)
    oz = np.zeros(m, dtype=np.uint8)
    _lib.check(_lib.load().plk_curve_fold_pairs(curve, m, _ptr(lo), _ptr(lz) if lz is not None else None, _ptr(hi),
                                                _ptr(hz) if hz is not None else None, _ptr(a), _ptr(b), _ptr(out), _ptr(oz)))
    return out, oz


def curve_sum_affine(curve, points, zero=None):
    """Adds k affine points (per-GPU partial MSM results after the all-gather)."""
    p = _points(curve, points)
    k = p.shape[0]
    z = None if zero is None else np.ascontiguousarray(zero, dtype=np.uint8)
    L = _CURVE_LIMBS[curve]
    out = np.zeros((2, L), dtype=np.uint64)
    oz = np.zeros(1, dtype=np.uint8)
    _lib.check(_lib.load().plk_curve_sum_affine(curve, k, _ptr(p), _ptr(z) if z is not None else None, _ptr(out), _ptr(oz)))
    return out, int(oz[0])


def affine_summation_best(curve, summation, zero=None):
    """curve_summations.rs:18-22 (and the _pairwise / _batch_inversion forms it chooses between, :39-58 / :60-68: one group element,
    whatever the form): the sum of a list of affine points, identity operands, P = Q and P = -Q included.  On the device the list is
    one workgroup's XYZZ sum (k_sum_affine); returns ((2, L) affine limbs, zero flag)."""
    return curve_sum_affine(curve, summation, zero)


def affine_multisummation_best(curve, summations, zeros=None):
    """curve_summations.rs:24-35: k independent sums of affine points -> k points, one device sum per list (inside the MSM the
    reference's call site, curve_msm.rs:131-145, is replaced by the bucket accumulation - DESIGN.md section 5)."""
    L = _CURVE_LIMBS[curve]
    out = []
    for k, pts in enumerate(summations):
        arr = np.asarray(pts, dtype=np.uint64).reshape(-1, 2, L)
        out.append(curve_sum_affine(curve, arr, None if zeros is None else zeros[k]))
    return out


def field_op(field, op, a, b=None):
    """Element-wise device field arithmetic (parity tests of the HIP field code)."""
    ops = {"add": 0, "sub": 1, "mul": 2, "neg": 3, "square": 4, "inverse": 5, "to_canonical": 6, "from_canonical": 7,
           "inverse_euclid": 8, "inverse_divsteps": 9, "inverse_divsteps_var": 10, "inverse_divsteps_one_lane": 11,
           "mul_add2_edge": 12, "wide_sum": 13}
    a = _elems(field, a)
    out = np.empty_like(a)
    bb = _elems(field, b) if b is not None else a
    _lib.check(_lib.load().plk_field_op(field, ops[op], _ptr(a), _ptr(bb), _ptr(out), a.shape[0]))
    return out


def msm_debug_digits(curve, scalars, window_bits):
    """The MSM's digit recoding on its own (to_digits, curve_msm.rs:159-180): scalars (n, 4) Montgomery limbs in the curve's scalar field
    -> (n, ceil((BITS + 1) / w)) signed digits as the ordering kernels form them, and the reference's unsigned digits rebuilt from them
    (u_j = d_j - carry_j + 2^w carry_(j+1), carry_(j+1) = [d_j - carry_j < 0]), and the carry out of the top window (always 0: it has room)."""
    import ctypes
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    nd = ctypes.c_uint()
    _lib.check(_lib.load().plk_msm_debug_digits(curve, window_bits, 0, None, None, ctypes.byref(nd)))
    d = np.zeros((s.shape[0], nd.value), dtype=np.int32)
    _lib.check(_lib.load().plk_msm_debug_digits(curve, window_bits, s.shape[0], _ptr(s), _ptr(d), ctypes.byref(nd)))
    # carry_(j+1) = [d_j - carry_j < 0]: a negative digit borrowed from the next window, and so did a ZERO digit that stands for
    # 2^w - 1 + carry_j = 2^w (the device has no entry for it: magnitude 0, carry out)
    unsigned = np.zeros(d.shape, dtype=np.int64)
    carry = np.zeros(s.shape[0], dtype=np.int64)
    for j in range(nd.value):
        v = d[:, j].astype(np.int64) - carry
        carry = (v < 0).astype(np.int64)
        unsigned[:, j] = v + (carry << window_bits)
    return d, unsigned, carry


# ---- the Plonk quotient numerator (plonk.rs:375-456, gates/) ----
NUM_WIRES, NUM_ROUTED_WIRES, NUM_CONSTANTS, GRID_WIDTH = 9, 6, 6, 65  # plonk.rs:21-25


def evaluate_all_constraints(field, local_constant_values, local_wire_values, right_wire_values, below_wire_values, inner_zeta, inner_a):
    """gates/mod.rs:46-125 at `count` points: constants (count, 6, 4), wires (count, 9, 4) x 3 -> (count, 8, 4).
    inner_zeta / inner_a: InnerC::ZETA and InnerC::A (4 limbs, Montgomery) - how InnerC enters the curve gates."""
    k = np.ascontiguousarray(local_constant_values, dtype=np.uint64).reshape(-1, NUM_CONSTANTS, 4)
    l, r, b = (np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, NUM_WIRES, 4) for x in (local_wire_values, right_wire_values, below_wire_values))
    count = k.shape[0]
    assert l.shape[0] == r.shape[0] == b.shape[0] == count
    zeta, a = (np.ascontiguousarray(x, dtype=np.uint64).reshape(4) for x in (inner_zeta, inner_a))
    out = np.empty((count, 8, 4), dtype=np.uint64)
    _lib.check(_lib.load().plk_plonk_evaluate_all_constraints(field, count, _ptr(k), _ptr(l), _ptr(r), _ptr(b), _ptr(zeta), _ptr(a), _ptr(out)))
    return out


def vanishing_points(field, degree, constants_8n, wire_values_8n, s_sigma_values_8n, plonk_z_points_8n, k_is, alpha, beta, gamma, inner_zeta, inner_a):
    """The 8n-point loop of Prover::vanishing_poly (plonk.rs:392-453); Polynomial::from_evaluations of the result
    (ifft_with_precomputation_power_of_2) is the vanishing polynomial.  Tables: (6, 8n, 4), (9, 8n, 4), (6, 8n, 4), (8n, 4)."""
    log_degree = log2_strict(degree)
    n8 = 8 * degree
    c = np.ascontiguousarray(constants_8n, dtype=np.uint64).reshape(NUM_CONSTANTS, n8, 4)
    w = np.ascontiguousarray(wire_values_8n, dtype=np.uint64).reshape(NUM_WIRES, n8, 4)
    s = np.ascontiguousarray(s_sigma_values_8n, dtype=np.uint64).reshape(NUM_ROUTED_WIRES, n8, 4)
    z = np.ascontiguousarray(plonk_z_points_8n, dtype=np.uint64).reshape(n8, 4)
    ks = np.ascontiguousarray(k_is, dtype=np.uint64).reshape(NUM_ROUTED_WIRES, 4)
    sc = [np.ascontiguousarray(x, dtype=np.uint64).reshape(4) for x in (alpha, beta, gamma, inner_zeta, inner_a)]
    out = np.empty((n8, 4), dtype=np.uint64)
    _lib.check(_lib.load().plk_plonk_vanishing_points(field, log_degree, _ptr(c), _ptr(w), _ptr(s), _ptr(z), _ptr(ks), *[_ptr(x) for x in sc], _ptr(out)))
    return out


# ---- batch inversion (field.rs:223-278, curve.rs:216-232) ----
def batch_multiplicative_inverse(field, x):
    """Field::batch_multiplicative_inverse (field.rs:251-278): panics ("No inverse") on a zero element -> AssertionError here."""
    a = _elems(field, x)
    out = np.empty_like(a)
    rc = _lib.load().plk_field_batch_inverse(field, _ptr(a), _ptr(out), a.shape[0])
    if rc == _lib.PLK_ERR_INVALID_ARG and _lib.load().plk_last_error().decode("utf-8", "replace").startswith("No inverse"):
        raise AssertionError("No inverse")  # the reference panics (field.rs:266); explicit, so it survives python -O
    _lib.check(rc)
    return out


def batch_multiplicative_inverse_opt(field, x):
    """Field::batch_multiplicative_inverse_opt (field.rs:223-249): (inverses, is_none) - zero elements have no inverse."""
    a = _elems(field, x)
    out = np.empty_like(a)
    none = np.zeros(a.shape[0], dtype=np.uint8)
    _lib.check(_lib.load().plk_field_batch_inverse_opt(field, _ptr(a), _ptr(out), _ptr(none), a.shape[0]))
    return out, none


def batch_to_affine(curve, proj_xyz, zero=None):
    """ProjectivePoint::batch_to_affine (curve.rs:216-232): (n, 3, L) homogeneous projective limbs (+ zero flags) -> ((n, 2, L), zero flags)."""
    L = _CURVE_LIMBS[curve]
    p = np.ascontiguousarray(proj_xyz, dtype=np.uint64).reshape(-1, 3, L)
    n = p.shape[0]
    z = np.ascontiguousarray(zero, dtype=np.uint8) if zero is not None else None
    out = np.empty((n, 2, L), dtype=np.uint64)
    oz = np.zeros(n, dtype=np.uint8)
    _lib.check(_lib.load().plk_curve_batch_to_affine(curve, n, _ptr(p), _ptr(z) if z is not None else None, _ptr(out), _ptr(oz)))
    return out, oz


# ---- canonical byte encodings (serialization.rs:17-72) ----
def field_to_bytes(field, x):
    """ToBytes for field elements: (n, BYTES) uint8, little-endian canonical value."""
    a = _elems(field, x)
    out = np.empty((a.shape[0], a.shape[1] * 8), dtype=np.uint8)
    _lib.check(_lib.load().plk_field_to_bytes(field, _ptr(a), a.shape[0], _ptr(out)))
    return out


def field_from_bytes(field, b):
    """FromBytes for field elements; "Out of range" (field.rs:100) raises ValueError."""
    L = _FIELD_LIMBS[field]
    bb = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1, L * 8)
    out = np.empty((bb.shape[0], L), dtype=np.uint64)
    rc = _lib.load().plk_field_from_bytes(field, _ptr(bb), bb.shape[0], _ptr(out))
    if rc == _lib.PLK_ERR_INVALID_ARG:
        raise ValueError(_lib.load().plk_last_error().decode())
    _lib.check(rc)
    return out


def point_to_bytes(curve, xy, zero=None):
    """ToBytes for AffinePoint: (n, 1 + BYTES) uint8: mask = zero | (y odd) << 1, then x."""
    L = _CURVE_LIMBS[curve]
    p = np.ascontiguousarray(xy, dtype=np.uint64).reshape(-1, 2, L)
    z = np.ascontiguousarray(zero, dtype=np.uint8) if zero is not None else None
    out = np.empty((p.shape[0], 1 + L * 8), dtype=np.uint8)
    _lib.check(_lib.load().plk_curve_point_to_bytes(curve, _ptr(p), _ptr(z) if z is not None else None, p.shape[0], _ptr(out)))
    return out


def point_from_bytes(curve, b, with_status=False):
    """FromBytes for AffinePoint: ((n, 2, L), zero flags); an undecodable record raises ValueError unless with_status."""
    L = _CURVE_LIMBS[curve]
    bb = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1, 1 + L * 8)
    n = bb.shape[0]
    out = np.empty((n, 2, L), dtype=np.uint64)
    oz = np.zeros(n, dtype=np.uint8)
    st = np.zeros(n, dtype=np.uint8)
    rc = _lib.load().plk_curve_point_from_bytes(curve, _ptr(bb), n, _ptr(out), _ptr(oz), _ptr(st))
    if with_status and rc in (_lib.PLK_OK, _lib.PLK_ERR_INVALID_ARG):
        return out, oz, st
    if rc == _lib.PLK_ERR_INVALID_ARG:
        raise ValueError(_lib.load().plk_last_error().decode())
    _lib.check(rc)
    return out, oz
