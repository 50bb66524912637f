"""GPU parity tests: the HIP path (through the C ABI of include/plonky_hip.h) against the oracle.

Bit-exact on integer limbs everywhere.  The tests read like the reference's own unit tests
(src/fft.rs:164-232, src/curve/curve_msm.rs:186-241, src/curve/curve_summations.rs:164-184,
src/field/field.rs:618-780) with the reference call replaced by the plonky_amd mirror.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import plonky_amd as pa
from plonky_amd import api, synth
from oracle import bigint_ref as br
from oracle import oracle_lib as ol
from tests.test_oracle_kats import (MSM_KAT_X_MONT, MSM_KAT_Y_MONT, _bls_test_msm_inputs, from_mont_arr, mont_arr,
                                    reference_test_inputs)
from tests.util import ints_to_array, limbs_to_int

FIELDS = [br.TWEEDLEDEE_BASE, br.TWEEDLEDUM_BASE, br.BLS12_377_SCALAR, br.BLS12_377_BASE, br.PALLAS_BASE, br.VESTA_BASE]
NTT_FIELDS = FIELDS  # all six: Bls12377Base (14 working limbs) since round 5
CURVES = [br.TWEEDLEDEE, br.TWEEDLEDUM, br.BLS12_377, br.PALLAS, br.VESTA]


# ---------------- field arithmetic: the reference's test_arithmetic! sweep on the device ----------------
@pytest.mark.parametrize("f", FIELDS, ids=lambda f: f.name)
def test_device_field_arithmetic_sweep(f):
    inputs = reference_test_inputs(f.p)
    m = len(inputs)
    x = mont_arr(f, inputs)
    for op in ("neg", "square", "to_canonical", "from_canonical"):
        assert np.array_equal(api.field_op(f.field_id, op, x), ol.field_unop(f.field_id, op, x)), op
    a = x[np.repeat(np.arange(m), m)]
    b = x[np.tile(np.arange(m), m)]
    for op in ("add", "sub", "mul"):
        assert np.array_equal(api.field_op(f.field_id, op, a, b), ol.field_binop(f.field_id, op, a, b)), op


@pytest.mark.parametrize("f", FIELDS, ids=lambda f: f.name)
def test_device_field_inverse_and_random(f):
    x = synth.rand_field(f.field_id, 0x1234, 4096)
    y = synth.rand_field(f.field_id, 0x5678, 4096)
    for op in ("add", "sub", "mul"):
        assert np.array_equal(api.field_op(f.field_id, op, x, y), ol.field_binop(f.field_id, op, x, y))
    small = mont_arr(f, list(range(0, 25)))
    assert np.array_equal(api.field_op(f.field_id, "inverse", small), ol.field_unop(f.field_id, "inverse", small))
    assert np.array_equal(api.field_op(f.field_id, "inverse", x[:256]), ol.field_unop(f.field_id, "inverse", x[:256]))
    # the two other inversion routines of the device code (Euclid as in the reference; division steps, used by the kernels)
    edge = mont_arr(f, [1, 2, f.p - 1, f.p - 2, (f.p - 1) // 2, (f.p + 1) // 2, 1 << 30, (1 << 60) + 1])
    for arr in (small, x, edge):
        want = ol.field_unop(f.field_id, "inverse", arr)
        assert np.array_equal(api.field_op(f.field_id, "inverse_euclid", arr), want)
        assert np.array_equal(api.field_op(f.field_id, "inverse_divsteps", arr), want)
        assert np.array_equal(api.field_op(f.field_id, "inverse_divsteps_var", arr), want)
        assert np.array_equal(api.field_op(f.field_id, "inverse_divsteps_one_lane", arr[:200]), want[:200])  # one lane per wave, scalar low words


@pytest.mark.parametrize("f", FIELDS, ids=lambda f: f.name)
def test_device_shared_reductions(f):
    """fz_mul_add2 and the column accumulators (FzWide) on the DEVICE: sums of products through one Montgomery reduction, with every
    limb below the top one at the largest value the kernels pass (the limb patterns of tests/fp_host_harness.cpp op 24) and on ordinary
    operands, against Python integers."""
    p, n = f.p, f.n_limbs
    nz = (64 * n + 28) // 29
    Rp_inv = pow(1 << (29 * nz), -1, p)
    inputs = reference_test_inputs(p)[:40] + [synth.to_int(r) for r in synth.rand_field(f.field_id, 0x5EED, 60)]
    m = len(inputs)
    x = ints_to_array(inputs, n)
    a, b = x[np.repeat(np.arange(m), m)], x[np.tile(np.arange(m), m)]
    got_edge = [limbs_to_int(r) for r in api.field_op(f.field_id, "mul_add2_edge", a, b)]
    got_wide = [limbs_to_int(r) for r in api.field_op(f.field_id, "wide_sum", a, b)]
    small = nz <= 9
    la = (1 << 29) + 7 if small else (1 << 29) + (1 << 27)
    lb, lc, ld, top = (1 << 31 if small else la), (1 << 30 if small else la), (1 << 29) - 1, (1 << 25 if small else 3)

    def val(limb, tl):
        return sum(limb << (29 * t) for t in range(nz - 1)) + (tl << (29 * (nz - 1)))
    k = 0
    for ai in inputs:
        xw = [(ai >> (32 * t)) & 0xFFFFFFFF for t in range(3)]
        for bj in inputs:
            yw = [(bj >> (32 * t)) & 0xFFFFFFFF for t in range(2)]
            A, B = val(la - (xw[0] & 7), top + (xw[1] & 0xFFFF)), val(lb - (yw[0] & 0xFF), top)
            C, Dv = val(lc - (xw[2] & 0xFF), top), val(ld - (yw[1] & 0xFF), (1 << 22) if small else 1)
            assert got_edge[k] == (A * B + C * Dv) * Rp_inv % p, (hex(ai), hex(bj))
            assert got_wide[k] == ((ai * bj + (ai + bj) * bj) * Rp_inv + ai) % p, (hex(ai), hex(bj))
            k += 1


# ---------------- NTT ----------------
def test_fft_and_ifft():
    """fft.rs:164-185 verbatim: degree 200, coeffs i*1337 % 100 in Bls12377Scalar."""
    f = br.BLS12_377_SCALAR
    degree = 200
    degree_padded = 1 << pa.log2_ceil(degree)
    coefficients = mont_arr(f, [(i * 1337) % 100 for i in range(degree)])
    precomputation = pa.fft_precompute(pa.BLS12_377_SCALAR, degree)
    assert precomputation.size() == 256
    points = pa.fft_with_precomputation(coefficients, precomputation)
    expected = br.ntt_naive(f, [(i * 1337) % 100 for i in range(degree)] + [0] * (degree_padded - degree))
    assert from_mont_arr(f, points) == expected
    interpolated = pa.ifft_with_precomputation_power_of_2(points, precomputation)
    assert np.array_equal(interpolated[:degree], coefficients)
    assert not interpolated[degree:].any()
    # and limb-for-limb the oracle's (= the reference algorithm's) output
    opre = ol.FftPrecomputation(2, degree)
    assert np.array_equal(points, opre.fft_with_precomputation(coefficients))
    assert np.array_equal(interpolated, opre.ifft_with_precomputation_power_of_2(points))


@pytest.mark.parametrize("f", NTT_FIELDS, ids=lambda f: f.name)
@pytest.mark.parametrize("log_n", list(range(0, 15)) + [16])
def test_ntt_matches_oracle(f, log_n):
    n = 1 << log_n
    x = synth.rand_field(f.field_id, 0xF70000 + log_n, n)
    pre = pa.fft_precompute(f.field_id, n)
    opre = ol.FftPrecomputation(f.field_id, n)
    fwd = pa.fft_with_precomputation_power_of_2(x, pre)
    assert np.array_equal(fwd, opre.fft_with_precomputation_power_of_2(x, threads=4))
    inv = pa.ifft_with_precomputation_power_of_2(x, pre)
    assert np.array_equal(inv, opre.ifft_with_precomputation_power_of_2(x, threads=4))
    assert np.array_equal(pa.ifft_with_precomputation_power_of_2(fwd, pre), x)


def test_fft_and_ifft_bls12_377_base():
    """The unit-test shape of fft.rs:164-185 over Bls12377Base (a `Field`, bls12_377_base.rs:18-262, so fft::<Bls12377Base> type-checks
    in the reference): zero-padded transform of 200 and 1000 coefficients against the big-integer transform and the oracle, inverse,
    a batch from host pointers, device tensors (plain and zero-padded), a three-pass size by linearity and round trip."""
    import torch
    from plonky_amd import device as dev
    f = br.BLS12_377_BASE
    for degree in (200, 1000):
        n = 1 << pa.log2_ceil(degree)
        ints = [(i * 1337 + (i * i) % 89) % f.p for i in range(degree)]
        coefficients = mont_arr(f, ints)
        pre = pa.fft_precompute(pa.BLS12_377_BASE, degree)
        assert pre.size() == n
        points = pa.fft_with_precomputation(coefficients, pre)
        assert points.shape == (n, 6)
        assert from_mont_arr(f, points) == br.ntt(f, ints + [0] * (n - degree))
        opre = ol.FftPrecomputation(3, degree)
        assert np.array_equal(points, opre.fft_with_precomputation(coefficients))
        back = pa.ifft_with_precomputation_power_of_2(points, pre)
        assert np.array_equal(back[:degree], coefficients) and not back[degree:].any()
        assert np.array_equal(back, opre.ifft_with_precomputation_power_of_2(points))
        # device-resident forms
        d = dev.to_device(coefficients)
        assert np.array_equal(dev.to_host(dev.ntt_padded_dev(3, d, pa.log2_ceil(degree))), points)
        assert np.array_equal(dev.to_host(dev.ntt_dev(3, dev.to_device(points), inverse=True)), back)
    n = 1 << 12
    polys = np.stack([synth.rand_field(3, 0xB377 + k, n) for k in range(5)])
    outs = api.fft_batch(3, polys)
    opre = ol.FftPrecomputation(3, n)
    for k in range(5):
        assert np.array_equal(outs[k], opre.fft_with_precomputation_power_of_2(polys[k], threads=4)), k
    assert np.array_equal(api.fft_batch(3, outs, inverse=True), polys)
    assert np.array_equal(dev.to_host(dev.ntt_dev(3, dev.to_device(polys))), outs)
    # 2^20 (three passes; 48 MiB per operand): linearity and the round trip, and a slice of outputs against direct evaluation
    n = 1 << 20
    a, b = synth.rand_field(3, 21, n), synth.rand_field(3, 22, n)
    fa, fb, fs = api.fft_batch(3, np.stack([a, b, api.field_op(3, "add", a, b)]))
    assert np.array_equal(api.field_op(3, "add", fa, fb), fs)
    assert np.array_equal(pa.ifft_with_precomputation_power_of_2(fa, pa.fft_precompute(3, n)), a)
    sparse = np.zeros((n, 6), dtype=np.uint64)
    idx = [0, 1, 5, 1 << 10, (1 << 19) + 3, n - 1]
    vals = [3, 5, f.p - 2, 7, 11, 13]
    sparse[idx] = mont_arr(f, vals)
    got = from_mont_arr(f, pa.fft_with_precomputation_power_of_2(sparse, pa.fft_precompute(3, n)))
    g = f.primitive_root_of_unity(20)
    for j in (0, 1, 2, 12345, (1 << 19) + 77, n - 1):
        assert got[j] == sum(v * pow(g, (i * j) % n, f.p) for i, v in zip(idx, vals)) % f.p, j
    with pytest.raises(Exception):  # the polynomial callers stay with the circuit's scalar fields
        pa.polynomial_divide_by_z_h(3, synth.rand_field(3, 1, 64), 16)


def test_ntt_config1_2p14_golden():
    """BASELINE config 1 shape (benches/fft.rs: TweedledeeBase, 2^14), seed 0xF70014."""
    x = synth.rand_field(0, 0xF70014, 1 << 14)
    pre = pa.fft_precompute(0, 1 << 14)
    out = pa.fft_with_precomputation_power_of_2(x, pre)
    assert np.array_equal(out, ol.FftPrecomputation(0, 1 << 14).fft_with_precomputation_power_of_2(x, threads=4))


def test_ntt_2p20_full_size():
    """BASELINE config 2: 2^20 TweedledeeBase, forward + inverse, bit-exact vs the oracle."""
    n = 1 << 20
    x = synth.rand_field(0, 0xF70020, n)
    pre = pa.fft_precompute(0, n)
    opre = ol.FftPrecomputation(0, n)
    fwd = pa.fft_with_precomputation_power_of_2(x, pre)
    assert np.array_equal(fwd, opre.fft_with_precomputation_power_of_2(x, threads=8))
    inv = pa.ifft_with_precomputation_power_of_2(fwd, pre)
    assert np.array_equal(inv, x)


def test_ntt_linearity_and_batch_2p18():
    """size-independent properties: NTT(a + b) = NTT(a) + NTT(b); batched == one by one."""
    f = br.TWEEDLEDUM_BASE
    n = 1 << 18
    a = synth.rand_field(1, 11, n)
    b = synth.rand_field(1, 12, n)
    s = api.field_op(1, "add", a, b)
    out = api.fft_batch(1, np.stack([a, b, s]))
    assert np.array_equal(api.field_op(1, "add", out[0], out[1]), out[2])
    pre = pa.fft_precompute(1, n)
    assert np.array_equal(out[0], pa.fft_with_precomputation_power_of_2(a, pre))
    back = api.fft_batch(1, out, inverse=True)
    assert np.array_equal(back[0], a) and np.array_equal(back[1], b)


def test_ntt_padding_and_errors():
    f = br.TWEEDLEDEE_BASE
    x = synth.rand_field(0, 5, 37)
    pre = pa.fft_precompute(0, 37)
    out = pa.fft_with_precomputation(x, pre)
    padded = np.zeros((64, 4), dtype=np.uint64)
    padded[:37] = x
    assert np.array_equal(out, pa.fft_with_precomputation_power_of_2(padded, pre))
    with pytest.raises(AssertionError):  # log2_strict panics (util.rs:17)
        pa.fft_with_precomputation_power_of_2(x, pre)
    with pytest.raises(AssertionError):  # field.rs:430
        pa.fft_precompute(1, 1 << 34)
    # the raw C ABI reports codes instead of aborting
    from plonky_amd import lib
    L = lib.load()
    assert L.plk_ntt(6, 4, 0, x.ctypes.data, x.ctypes.data) == lib.PLK_ERR_INVALID_ARG  # no such field
    assert L.plk_ntt(0, 4, 0, None, None) == lib.PLK_ERR_INVALID_ARG
    assert L.plk_ntt(0, 31, 0, x.ctypes.data, x.ctypes.data) == lib.PLK_ERR_TWO_ADICITY
    assert len(L.plk_last_error()) > 0


# ---------------- MSM ----------------
def _bases(c, pts):
    return np.array([[c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])] for P in pts], dtype=np.uint64).reshape(-1, 2, c.base.n_limbs)


def test_msm():
    """curve_msm.rs:218-241: BLS12-377, generators G, 2G, 3G, the three fixed scalars, w = 5."""
    c, pts, scal = _bls_test_msm_inputs()
    generators = _bases(c, pts)
    scalars = mont_arr(c.scalar, scal)
    precomputation = pa.msm_precompute(pa.BLS12_377, generators, 5)
    result_msm, zero = pa.msm_execute(precomputation, scalars)
    assert zero == 0
    assert list(result_msm[0]) == MSM_KAT_X_MONT and list(result_msm[1]) == MSM_KAT_Y_MONT
    # result_naive: sum of naive scalar multiplications (oracle's mul_naive)
    exp = br.msm(c, scal, pts)
    assert tuple(from_mont_arr(c.base, result_msm)) == exp
    for win in (2, 3, 5, 8, 13):
        pre = pa.msm_precompute(pa.BLS12_377, generators, 5, device_window=win)
        r, z = pa.msm_execute_parallel(pre, scalars)
        assert z == 0 and np.array_equal(r, result_msm), win


def test_to_digits_reference_vector_on_device():
    """curve_msm.rs:186-216 (test_to_digits): the device never stores digits - it recodes them, signed, inside the ordering kernels
    (ord_digit, msm.hip).  plk_msm_debug_digits runs exactly that code; the reference's unsigned digits follow from the signed ones
    by u_j = d_j - carry_j + 2^w carry_(j+1), and must be the reference's vector.  Then the same identity on random scalars of every
    curve and several windows against the oracle's to_digits, plus the defining property sum_j d_j 2^(w j) = the canonical scalar."""
    x_canonical = [
        0b1010101010101010101010101010101010101010101010101010101010101010,
        0b1100110011001100110011001100110011001100110011001100110011001100,
        0b1111000011110000111100001111000011110000111100001111000011110000,
        0b0000111111111111111111111111111111111111111111111111111111111111,
    ]
    expected = [
        0b01010101010101010, 0b10101010101010101, 0b01010101010101010, 0b11001010101010101,
        0b01100110011001100, 0b00110011001100110, 0b10011001100110011, 0b11110000110011001,
        0b01111000011110000, 0b00111100001111000, 0b00011110000111100, 0b11111111111111110,
        0b11111111111111111, 0b11111111111111111, 0b00011111111111111,
    ]
    x = ol.field_unop(2, "from_canonical", np.array([x_canonical], dtype=np.uint64))
    signed, unsigned, top_carry = pa.msm_debug_digits(pa.BLS12_377, x, 17)
    assert signed.shape == (1, 15) and top_carry[0] == 0
    assert list(unsigned[0]) == expected
    assert np.abs(signed).max() <= 1 << 16
    rng = np.random.default_rng(0xD161)
    for curve, c in ((pa.TWEEDLEDEE, br.TWEEDLEDEE), (pa.TWEEDLEDUM, br.TWEEDLEDUM), (pa.BLS12_377, br.BLS12_377)):
        r = c.scalar.p
        vals = [0, 1, 2, r - 1, r - 2, (1 << 200) - 1, 1 << 200, (r - 1) // 2] + [int(v) ** 4 % r for v in rng.integers(0, 2 ** 63, 56)]
        sc = mont_arr(c.scalar, vals)
        for w in (3, 5, 11, 13, 16, 17, 20, 21):
            signed, unsigned, top_carry = pa.msm_debug_digits(curve, sc, w)
            assert not top_carry.any() and np.abs(signed).max() <= 1 << (w - 1)
            for i, v in enumerate(vals):
                assert sum(int(d) << (w * j) for j, d in enumerate(signed[i])) == v, (curve, w, i)
                ref = ol.to_digits(curve, sc[i], w)    # the reference's digit count: ceil(BITS / w)
                got = [int(u) for u in unsigned[i]]
                assert got[:len(ref)] == ref and not any(got[len(ref):]), (curve, w, i)


def test_msm_tweedledee_mini_kat():
    c = br.TWEEDLEDEE
    G = (c.gx, c.gy)
    pts = [G, br.ec_mul(c, 2, G), br.ec_mul(c, 3, G)]
    pre = pa.msm_precompute(pa.TWEEDLEDEE, _bases(c, pts), 11)
    out, zero = pa.msm_execute(pre, mont_arr(c.scalar, [1, 2, 3]))
    assert zero == 0 and tuple(from_mont_arr(c.base, out)) == br.ec_mul(c, 14, G)
    assert list(synth.to_limbs(br.ec_mul(c, 14, G)[0], 4)) == [14704193998986281273, 16806378273837548806, 1462211411981937090, 3084878336663070809]


def test_msm_length_mismatch():
    c, pts, scal = _bls_test_msm_inputs()
    pre = pa.msm_precompute(pa.BLS12_377, _bases(c, pts), 5)
    with pytest.raises(AssertionError):  # assert_eq! curve_msm.rs:67,106
        pa.msm_execute(pre, mont_arr(c.scalar, scal[:2]))
    from plonky_amd import lib
    s = mont_arr(c.scalar, scal[:2])
    out = np.zeros((2, 6), dtype=np.uint64)
    oz = np.zeros(1, dtype=np.uint8)
    assert lib.load().plk_msm_execute(pre._ctx, s.ctypes.data, 2, out.ctypes.data, oz.ctypes.data) == lib.PLK_ERR_SIZE_MISMATCH


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_msm_summation_edge_cases(c):
    """curve_summations.rs:164-184 shapes through the MSM: {G,G}, {G,2G}, {G,G,G}, {}, G + (-G), identity operands."""
    G = (c.gx, c.gy)
    G2, G3 = br.ec_mul(c, 2, G), br.ec_mul(c, 3, G)
    one = mont_arr(c.scalar, [1])[0]

    def run(pts, scal=None, zero=None):
        s = np.array([one] * len(pts), dtype=np.uint64).reshape(-1, 4) if scal is None else mont_arr(c.scalar, scal)
        return pa.msm_parallel(c.curve_id, s, _bases(c, pts) if pts else np.zeros((0, 2, c.base.n_limbs), dtype=np.uint64), 4, zero=zero)

    out, z = run([G, G])
    assert z == 0 and tuple(from_mont_arr(c.base, out)) == G2
    out, z = run([G, G2])
    assert z == 0 and tuple(from_mont_arr(c.base, out)) == G3
    out, z = run([G, G, G])
    assert z == 0 and tuple(from_mont_arr(c.base, out)) == G3
    out, z = run([])
    assert z == 1
    out, z = run([G, br.ec_neg(c, G)])
    assert z == 1 and not out.any()
    out, z = run([G, G2, G3], zero=[0, 1, 0])
    assert z == 0 and tuple(from_mont_arr(c.base, out)) == br.ec_mul(c, 4, G)
    out, z = run([G, G2], scal=[0, 0])
    assert z == 1
    r = c.scalar.p
    out, z = run([G, G2, G3], scal=[r - 1, 1, 0])  # -G + 2G
    assert z == 0 and tuple(from_mont_arr(c.base, out)) == G


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n,win", [(1, 0), (10, 0), (24, 3), (24, 7), (300, 0), (1000, 6), (1000, 14), (4096, 0), (4096, 11), (4096, 13), (4096, 16), (300, 20)])
def test_msm_matches_oracle(c, n, win):
    """msm_execute_parallel on seeded inputs vs the oracle's restatement of the reference (w = 8 tables)."""
    G = (c.gx, c.gy)
    d = limbs_to_int(ol.rand_field(c.scalar.field_id, 7, 1)[0]) % c.scalar.p
    D = br.ec_mul(c, d, G)
    bases = ol.gen_bases(c.curve_id, n, _bases(c, [G])[0], _bases(c, [D])[0])
    scalars = synth.rand_field(c.scalar.field_id, 0x350020 + n, n)
    if n >= 10:  # edge scalars 0, 1, r-1 and a duplicated base (doubling branch)
        scalars[0] = mont_arr(c.scalar, [0])[0]
        scalars[1] = mont_arr(c.scalar, [1])[0]
        scalars[2] = mont_arr(c.scalar, [c.scalar.p - 1])[0]
        bases[5] = bases[4]
        scalars[5] = scalars[4]
        # values around which the endomorphism split of the table-free mode turns: a cube root of unity, r / 2, 2^128
        lam = pow(5, (c.scalar.p - 1) // 3, c.scalar.p)
        scalars[6] = mont_arr(c.scalar, [lam])[0]
        scalars[7] = mont_arr(c.scalar, [c.scalar.p - lam])[0]
        scalars[8] = mont_arr(c.scalar, [c.scalar.p // 2])[0]
        scalars[9] = mont_arr(c.scalar, [1 << 128])[0]
    expected, ez = ol.MsmPrecomputation(c.curve_id, bases, 8, threads=8).execute(scalars, parallel=True, threads=8)
    pre = pa.msm_precompute(c.curve_id, bases, 8, device_window=win)
    got, gz = pa.msm_execute_parallel(pre, scalars)
    assert gz == ez and np.array_equal(got, expected)
    # the table-free mode (every window its own buckets, doubled into place at the end; GLV-split scalars on the prime-order
    # curves; windows above 12 bits reduce every window through its own row / column sums): same point
    if win <= 16:
        pre_tf = pa.msm_precompute(c.curve_id, bases, 8, device_window=win, table_free=True)
        got, gz = pa.msm_execute_parallel(pre_tf, scalars)
        assert gz == ez and np.array_equal(got, expected)
        got, gz = pa.msm_execute_parallel(pre_tf, scalars)  # the context is reusable
        assert gz == ez and np.array_equal(got, expected)

@pytest.mark.parametrize("c", [br.TWEEDLEDEE, br.BLS12_377], ids=lambda c: c.name)
def test_msm_execute_parts_matches_oracle(c):
    """plk_msm_execute_parts_dev: vectors that cover only a sub-range of the precomputed generators (a rank's share of a sharded
    commitment next to its whole vectors) - one batched call; every result against the oracle's MSM over exactly those generators.
    Ranges at the start, in the middle, at the end, of length 0 and 1, and the whole list; errors for a range past the end and for
    a table-free context."""
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    from plonky_amd.lib import PlonkyHipError
    dev.init(0)
    n = 3000
    G = (c.gx, c.gy)
    pt = lambda P: np.array([c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])], dtype=np.uint64)
    bases = ol.gen_bases(c.curve_id, n, pt(G), pt(br.ec_mul(c, 0x5150, G)))
    pre = dev.msm_precompute_dev(c.curve_id, dev.to_device(bases.reshape(n, 2, -1)))
    ranges = [(0, n), (0, 700), (1234, 1000), (n - 17, 17), (5, 0), (2999, 1), (0, n)]
    vecs = [ol.rand_field(c.scalar.field_id, 900 + k, cnt) if cnt else np.zeros((0, 4), dtype=np.uint64) for k, (_, cnt) in enumerate(ranges)]
    parts = [(first, dev.to_device(v).reshape(-1, 4)) for (first, _), v in zip(ranges, vecs)]
    oxy, oz = dev.msm_execute_parts_dev(pre, parts)
    got, gz = dev.to_host(oxy), oz.cpu().numpy()
    for k, ((first, cnt), v) in enumerate(zip(ranges, vecs)):
        if cnt == 0:
            assert gz[k] == 1
            continue
        exp, ez = ol.MsmPrecomputation(c.curve_id, bases[first:first + cnt], 8, threads=8).execute(v, parallel=True, threads=8)
        assert int(gz[k]) == ez and (ez or np.array_equal(got[k], exp)), (k, first, cnt)
    with pytest.raises(PlonkyHipError, match="covers generators"):
        dev.msm_execute_parts_dev(pre, [(n - 5, dev.to_device(vecs[1][:6]))])
    tf = dev.msm_precompute_dev(c.curve_id, dev.to_device(bases.reshape(n, 2, -1)), table_free=True)
    with pytest.raises(PlonkyHipError, match="tabled context"):
        dev.msm_execute_parts_dev(tf, [(0, dev.to_device(vecs[1]))])



def test_msm_parallel_one_shot():
    """msm_parallel (curve_msm.rs:54-61; the call of the IPA rounds, halo.rs:87-91, window 8 there): fresh generators,
    one execution -> the table-free path, against the oracle's msm_parallel at the reference's window."""
    c = br.TWEEDLEDEE
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 99991, G)
    for n in (2, 64, 1 << 12):
        bases = ol.gen_bases(0, n, _bases(c, [G])[0], _bases(c, [D])[0])
        scalars = synth.rand_field(c.scalar.field_id, 0x1A0000 + n, n)
        expected, ez = ol.MsmPrecomputation(0, bases, 8, threads=8).execute(scalars, parallel=True, threads=8)
        got, gz = pa.msm_parallel(0, scalars, bases, 8)
        assert gz == ez and np.array_equal(got, expected)
    # skewed digits and identity inputs through the table-free buckets
    n = 1 << 12
    bases = ol.gen_bases(0, n, _bases(c, [G])[0], _bases(c, [D])[0])
    scalars = mont_arr(c.scalar, [(i % 3) for i in range(n)])
    zero = np.zeros(n, dtype=np.uint8)
    zero[7] = 1
    expected, ez = ol.MsmPrecomputation(0, bases, 8, zero=zero, threads=8).execute(scalars, parallel=True, threads=8)
    got, gz = pa.msm_parallel(0, scalars, bases, 8, zero=zero)
    assert gz == ez and np.array_equal(got, expected)


def test_msm_skewed_digits():
    """A witness-like scalar vector (mostly 0 / 1 / small values): a handful of huge buckets."""
    c = br.TWEEDLEDEE
    n = 1 << 13
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 12345, G)
    bases = ol.gen_bases(0, n, _bases(c, [G])[0], _bases(c, [D])[0])
    vals = [(i % 3) for i in range(n)]
    vals[17] = c.scalar.p - 1
    scalars = mont_arr(c.scalar, vals)
    expected, ez = ol.MsmPrecomputation(0, bases, 8, threads=8).execute(scalars, parallel=True, threads=8)
    got, gz = pa.msm_execute_parallel(pa.msm_precompute(0, bases, 11), scalars)
    assert gz == ez and np.array_equal(got, expected)


def test_msm_batch_and_sum_affine():
    """9-wire commit shape (plonk_util.rs:215-231) at small n + combining per-shard partial results."""
    c = br.TWEEDLEDEE
    n = 2048
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 99, G)
    bases = ol.gen_bases(0, n, _bases(c, [G])[0], _bases(c, [D])[0])
    sv = np.stack([synth.rand_field(1, 0x350920 + k, n) for k in range(9)])
    pre = pa.msm_precompute(0, bases, 11)
    outs, zs = api.msm_execute_batch(pre, sv)
    opre = ol.MsmPrecomputation(0, bases, 8, threads=8)
    for k in range(9):
        e, ez = opre.execute(sv[k], parallel=True, threads=8)
        assert zs[k] == ez and np.array_equal(outs[k], e)
    # base-range sharding: 4 shards, partial sums added by plk_curve_sum_affine
    parts, pz = [], []
    for s in range(4):
        lo, hi = s * n // 4, (s + 1) * n // 4
        o, z = pa.msm_parallel(0, sv[0][lo:hi], bases[lo:hi], 11)
        parts.append(o)
        pz.append(z)
    tot, tz = api.curve_sum_affine(0, np.stack(parts), pz)
    assert tz == 0 and np.array_equal(tot, outs[0])


def test_gen_bases_dev_matches_oracle():
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    dev.init(0)
    for c in CURVES:
        G = (c.gx, c.gy)
        D = br.ec_mul(c, 0xABCDEF, G)
        g0, dd = _bases(c, [G])[0], _bases(c, [D])[0]
        got = dev.to_host(dev.gen_bases_dev(c.curve_id, 500, g0, dd)).reshape(500, 2, -1)
        assert np.array_equal(got, ol.gen_bases(c.curve_id, 500, g0, dd))


def test_msm_2p20_closed_form():
    """BASELINE config 3 at full size: bases G0 + i D admit the closed form
    sum s_i (G0 + i D) = [sum s_i] G0 + [sum i s_i] D   -- no CPU MSM needed."""
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    from plonky_amd.selfcheck import closed_form_msm
    dev.init(0)
    c = br.TWEEDLEDEE
    n = 1 << 20
    G = (c.gx, c.gy)
    d = limbs_to_int(synth.rand_field(1, 0x350020, 1)[0]) % c.scalar.p
    D = br.ec_mul(c, d, G)
    bases = dev.gen_bases_dev(0, n, _bases(c, [G])[0], _bases(c, [D])[0])
    scal = synth.rand_field(1, 0x350020, n)
    pre = dev.msm_precompute_dev(0, bases)
    oxy, oz = dev.msm_execute_dev(pre, dev.to_device(scal))
    torch.cuda.synchronize()
    got = dev.to_host(oxy).reshape(2, 4)
    exp = closed_form_msm(0, scal, G, D)
    assert int(oz.cpu()[0]) == 0 and tuple(from_mont_arr(c.base, got)) == exp
    # the same MSM without tables (one-shot mode)
    pre.free()
    pre_tf = dev.msm_precompute_dev(0, bases, table_free=True)
    oxy, oz = dev.msm_execute_dev(pre_tf, dev.to_device(scal))
    torch.cuda.synchronize()
    got = dev.to_host(oxy).reshape(2, 4)
    assert int(oz.cpu()[0]) == 0 and tuple(from_mont_arr(c.base, got)) == exp


def test_golden_vectors():
    """The HIP path against the committed fixtures of tests/golden/ (no oracle involved)."""
    import glob
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for path in sorted(glob.glob(os.path.join(gold, "poly_*.npz"))):
        g = np.load(path)
        assert np.array_equal(pa.polynomial_divide_by_z_h(int(g["field"]), g["numerator"], int(g["n"])), g["quotient"]), path
        assert np.array_equal(pa.polynomial_mul(int(g["field"]), g["a"], g["b"]), g["product"]), path
    for path in sorted(glob.glob(os.path.join(gold, "ntt_*.npz"))):
        g = np.load(path)
        pre = pa.fft_precompute(int(g["field"]), g["input"].shape[0])
        assert np.array_equal(pa.fft_with_precomputation_power_of_2(g["input"], pre), g["forward"]), path
        assert np.array_equal(pa.ifft_with_precomputation_power_of_2(g["input"], pre), g["inverse"]), path
    for path in sorted(glob.glob(os.path.join(gold, "msm_*.npz"))):
        g = np.load(path)
        for win in (0, 5):
            pre = pa.msm_precompute(int(g["curve"]), g["bases"], 11, device_window=win)
            xy, z = pa.msm_execute_parallel(pre, g["scalars"])
            assert z == int(g["expected_zero"]) and np.array_equal(xy, g["expected_xy"]), (path, win)


def test_msm_heavy_buckets_closed_form():
    """Extremely skewed scalars (most of them identical, the rest tiny): a few buckets receive ~10^5
    entries each - the heavy-bucket path of the reduction (multi-chunk) - checked with the closed form."""
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    from plonky_amd.selfcheck import closed_form_msm
    dev.init(0)
    c = br.TWEEDLEDEE
    n = 1 << 17
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 987654321, G)
    bases = dev.gen_bases_dev(0, n, _bases(c, [G])[0], _bases(c, [D])[0])
    hot = synth.rand_field(1, 4242, 1)[0]
    scal = np.tile(hot, (n, 1))
    small = mont_arr(c.scalar, [0, 1, 2, 3])
    idx = np.arange(n)
    sel = idx % 10 >= 7
    scal[sel] = small[idx[sel] % 4]
    pre = dev.msm_precompute_dev(0, bases)
    oxy, oz = dev.msm_execute_dev(pre, dev.to_device(scal))
    torch.cuda.synchronize()
    got = dev.to_host(oxy).reshape(2, 4)
    exp = closed_form_msm(0, scal, G, D)
    assert int(oz.cpu()[0]) == 0 and tuple(from_mont_arr(c.base, got)) == exp


def test_concurrent_callers():
    """the reference's callers invoke these functions concurrently from Rayon workers
    (plonk_util.rs:173-189, halo.rs:119-123).  Eight host threads hammer NTTs, a shared MSM context, their own
    one-shot MSMs and divide_by_z_h at the same time; every result must be the single-threaded one."""
    import threading

    f = br.TWEEDLEDEE_BASE
    c = br.TWEEDLEDEE
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 424242, G)
    n = 1 << 10
    bases = ol.gen_bases(0, n, _bases(c, [G])[0], _bases(c, [D])[0])
    shared = pa.msm_precompute(0, bases, 8)
    xs = [synth.rand_field(f.field_id, 0x7000 + t, 1 << (8 + t % 4)) for t in range(8)]
    ss = [synth.rand_field(c.scalar.field_id, 0x7100 + t, n) for t in range(8)]
    qs = [synth.rand_field(f.field_id, 0x7200 + t, 300 + 17 * t) for t in range(8)]
    want_ntt = [pa.fft_with_precomputation_power_of_2(x, pa.fft_precompute(f.field_id, x.shape[0])) for x in xs]
    want_msm = [pa.msm_execute_parallel(shared, s) for s in ss]
    from tests.test_gpu_poly import mul_by_z_h_mont
    ms = [mul_by_z_h_mont(f, q, 64) for q in qs]
    errors = []

    def worker(t):
        try:
            for _ in range(6):
                x = xs[t]
                assert np.array_equal(pa.fft_with_precomputation_power_of_2(x, pa.fft_precompute(f.field_id, x.shape[0])), want_ntt[t])
                got, gz = pa.msm_execute_parallel(shared, ss[t])
                assert gz == want_msm[t][1] and np.array_equal(got, want_msm[t][0])
                got, gz = pa.msm_parallel(0, ss[t], bases, 8)
                assert gz == want_msm[t][1] and np.array_equal(got, want_msm[t][0])
                q = pa.polynomial_divide_by_z_h(f.field_id, ms[t], 64)
                assert np.array_equal(q[: qs[t].shape[0]], qs[t]) and not q[qs[t].shape[0]:].any()
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


# ---------------- IPA generator fold (halo.rs:119-123) ----------------
@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_fold_generators(c):
    """G'_i = [u_inv] G_lo_i + [u] G_hi_i against the oracle's scalar multiplication (curve_multiplication.rs:5-85) and
    addition, pair by pair; includes equal / opposite / identity inputs and the scalars 0, 1, r - 1."""
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 0xF01D + c.curve_id, G)
    m = 20
    pts = ol.gen_bases(c.curve_id, 2 * m, _bases(c, [G])[0], _bases(c, [D])[0])
    lo, hi = pts[:m].copy(), pts[m:].copy()
    hi[3] = lo[3]                                   # P == Q
    neg = list(br.ec_neg(c, tuple(from_mont_arr(c.base, lo[4]))))
    hi[4] = mont_arr(c.base, neg)                   # Q == -P
    lz = np.zeros(m, dtype=np.uint8)
    hz = np.zeros(m, dtype=np.uint8)
    lz[5] = 1
    hz[6] = 1
    lz[7] = hz[7] = 1
    r = c.scalar.p
    u = limbs_to_int(synth.rand_field(c.scalar.field_id, 0xF01D, 1)[0]) % r
    u = c.scalar.from_mont(u) or 3
    lam = pow(5, (r - 1) // 3, r)  # a cube root of unity of the scalar field: the value the endomorphism split turns around
    cases = [(pow(u, -1, r), u), (0, 1), (1, 0), (r - 1, 1), (1, r - 1), (5, 5), (0, 0), (lam, r - lam), (r // 2, r // 2 + 1), (lam * lam % r, 1 << 128)]
    for a, b in cases:
        sa, sb = mont_arr(c.scalar, [a])[0], mont_arr(c.scalar, [b])[0]
        got, gz = pa.fold_generators(c.curve_id, lo, hi, sa, sb, lo_zero=lz, hi_zero=hz)
        for i in range(m):
            p1, z1 = ol.scalar_mul(c.curve_id, sa, lo[i], int(lz[i]))
            p2, z2 = ol.scalar_mul(c.curve_id, sb, hi[i], int(hz[i]))
            exp, ez = ol.affine_add(c.curve_id, p1, z1, p2, z2)
            assert int(gz[i]) == ez, (a, b, i)
            if not ez:
                assert np.array_equal(got[i], exp), (a, b, i)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_msm_precompute_table_is_the_reference_table(c):
    """msm_precompute (curve_msm.rs:27-52): the device-built powers_per_generator[i][j] = [2^(w j)] G_i against the
    oracle's restatement, entry by entry, for the reference's window sizes 4, 8 and 11 and an identity generator."""
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 77, G)
    n = 6
    bases = ol.gen_bases(c.curve_id, n, _bases(c, [G])[0], _bases(c, [D])[0])
    zero = np.zeros(n, dtype=np.uint8)
    zero[2] = 1
    for w in (4, 8, 11):
        tab, tz = pa.msm_precompute_table(c.curve_id, bases, w, zero=zero)
        digits = (c.scalar.bits + w - 1) // w
        assert tab.shape == (n, digits, 2, c.base.n_limbs)
        pre = ol.MsmPrecomputation(c.curve_id, bases, w, zero=zero)
        for i in range(n):
            for j in range(digits):
                exp, ez = pre.table_entry(i, j)
                assert int(tz[i, j]) == ez, (w, i, j)
                if not ez:
                    assert np.array_equal(tab[i, j], exp), (w, i, j)
                else:
                    assert not tab[i, j].any()


def test_coeffs_vec_to_commitments():
    """PolynomialCommitment::coeffs_vec_to_commitments (poly_commit.rs:51-66): pedersen_hash(coeffs) + [r] H for several
    polynomials, normalised; against the oracle's MSM, scalar multiplication and addition."""
    c = br.TWEEDLEDEE
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 31337, G)
    n = 200
    gens = ol.gen_bases(0, n + 1, _bases(c, [G])[0], _bases(c, [D])[0])
    g, h = gens[:n], gens[n:]
    pre = pa.commitment_precompute(0, g, h, 8)
    k = 5
    coeffs = np.stack([synth.rand_field(c.scalar.field_id, 0xC0 + t, n) for t in range(k)])
    blind = synth.rand_field(c.scalar.field_id, 0xB1, k)
    blind[1] = 0  # blinding off for one of them
    got, gz = pa.coeffs_vec_to_commitments(pre, coeffs, blind)
    opre = ol.MsmPrecomputation(0, g, 8)
    for t in range(k):
        m, mz = opre.execute(coeffs[t])
        b, bz = ol.scalar_mul(0, blind[t], h[0], 0)
        exp, ez = ol.affine_add(0, m, mz, b, bz)
        assert int(gz[t]) == ez and np.array_equal(got[t], exp), t


@pytest.mark.parametrize("table_free", [False, True])
def test_msm_batch_larger_than_one_group(table_free):
    """20 scalar vectors in one call: more than one group of the shared reduction (16 per group), with tables and
    table-free, skewed vectors included; every result equals the single-vector execution and the oracle."""
    c = br.TWEEDLEDUM
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 271828, G)
    n = 300
    bases = ol.gen_bases(c.curve_id, n, _bases(c, [G])[0], _bases(c, [D])[0])
    vecs = [synth.rand_field(c.scalar.field_id, 0xBA7C0 + t, n) for t in range(20)]
    vecs[3] = mont_arr(c.scalar, [7] * n)          # one bucket takes everything
    vecs[11] = mont_arr(c.scalar, [0] * n)         # identity result
    pre = pa.msm_precompute(c.curve_id, bases, 8, table_free=table_free)
    out, oz = pa.msm_execute_batch(pre, np.stack(vecs))
    opre = ol.MsmPrecomputation(c.curve_id, bases, 8, threads=8)
    for t in range(20):
        exp, ez = opre.execute(vecs[t], parallel=True, threads=8)
        assert int(oz[t]) == ez, t
        if not ez:
            assert np.array_equal(out[t], exp), t
        one, z1 = pa.msm_execute_parallel(pre, vecs[t])
        assert z1 == ez and (ez or np.array_equal(one, exp)), t


# ---------------- batch inversion (field.rs:223-278, curve.rs:216-232) ----------------
@pytest.mark.parametrize("f", [br.TWEEDLEDEE_BASE, br.TWEEDLEDUM_BASE, br.BLS12_377_SCALAR, br.BLS12_377_BASE], ids=lambda f: f.name)
def test_batch_multiplicative_inverse(f):
    """bls12_377_base.rs:336-361 (inverses of 1..24 through the batch path) + seeded inputs of awkward lengths vs the oracle's
    Montgomery-trick restatement; a zero element is `No inverse` (field.rs:266) / None in the _opt variant."""
    from plonky_amd import api
    small = ints_to_array([f.to_mont(v) for v in range(1, 25)], f.n_limbs)
    got = api.batch_multiplicative_inverse(f.field_id, small)
    assert [f.from_mont(limbs_to_int(r)) for r in got] == [pow(v, -1, f.p) for v in range(1, 25)]
    for n in (1, 7, 8, 9, 1000, 4097):
        x = ol.rand_field(f.field_id, 0xBA7C + n, n)
        assert np.array_equal(api.batch_multiplicative_inverse(f.field_id, x), ol.batch_inverse(f.field_id, x)), n
    x = ol.rand_field(f.field_id, 5, 100)
    x[17] = 0
    x[99] = 0
    with pytest.raises(AssertionError):
        api.batch_multiplicative_inverse(f.field_id, x)
    inv, none = api.batch_multiplicative_inverse_opt(f.field_id, x)
    assert list(np.nonzero(none)[0]) == [17, 99] and not inv[17].any() and not inv[99].any()
    keep = np.ones(100, dtype=bool)
    keep[[17, 99]] = False
    assert np.array_equal(inv[keep], ol.batch_inverse(f.field_id, x[keep]))
    assert api.batch_multiplicative_inverse(f.field_id, x[:0]).shape[0] == 0


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_batch_to_affine(c):
    """curve.rs:216-232: random projective representatives (X, Y, Z) = (x z, y z, z) of known affine points, identity flags."""
    from plonky_amd import api
    G = (c.gx, c.gy)
    p = c.base.p
    pts, proj, zero = [], [], []
    for i in range(37):
        P = br.ec_mul(c, 1000 + 17 * i, G)
        z = (0x1234567 * (i + 1) + 99) % p
        pts.append(P)
        proj.append([P[0] * z % p, P[1] * z % p, z])
        zero.append(1 if i % 11 == 5 else 0)
    arr = np.array([[c.base.mont_limbs(v) for v in row] for row in proj], dtype=np.uint64)
    out, oz = api.batch_to_affine(c.curve_id, arr, zero)
    for i in range(37):
        if zero[i]:
            assert oz[i] == 1 and not out[i].any()
        else:
            assert oz[i] == 0 and tuple(from_mont_arr(c.base, out[i])) == pts[i]


def test_msm_sparse_scalars_many_buckets():
    """Scalar vectors that are almost all zero (Z = 1, zero-padded quotient chunks) against a large window: most buckets are
    empty and the entries of a lane are far apart in the bucket order."""
    c = br.TWEEDLEDEE
    n = 4096
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 31337, G)
    bases = ol.gen_bases(0, n, _bases(c, [G])[0], _bases(c, [D])[0])
    vals = [0] * n
    vals[0], vals[5], vals[n - 1] = 1, c.scalar.p - 2, 0x123456789ABCDEF0123456789
    scalars = mont_arr(c.scalar, vals)
    expected, ez = ol.MsmPrecomputation(0, bases, 8, threads=8).execute(scalars, parallel=True, threads=8)
    for win in (16, 20):
        got, gz = pa.msm_execute_parallel(pa.msm_precompute(0, bases, 11, device_window=win), scalars)
        assert gz == ez and np.array_equal(got, expected), win


# ---------------- canonical byte encodings (serialization.rs:17-72) ----------------
def test_bytes_match_golden_and_oracle():
    import glob, os
    from plonky_amd import api
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    paths = sorted(glob.glob(os.path.join(gold, "bytes_*.npz")))
    assert len(paths) == 3
    for path in paths:
        g = np.load(path)
        f, c = int(g["field"]), int(g["curve"])
        assert np.array_equal(api.field_to_bytes(f, g["elems"]), g["elem_bytes"]), path
        assert np.array_equal(api.field_from_bytes(f, g["elem_bytes"]), g["elems"]), path
        assert np.array_equal(api.point_to_bytes(c, g["points"], g["points_zero"]), g["point_bytes"]), path
        xy, zero = api.point_from_bytes(c, g["point_bytes"])
        assert np.array_equal(zero, g["points_zero"]) and np.array_equal(xy, g["points"]), path
    for f in (br.TWEEDLEDEE_BASE, br.TWEEDLEDUM_BASE, br.BLS12_377_SCALAR, br.BLS12_377_BASE):
        x = ol.rand_field(f.field_id, 0x5E71A2, 333)
        b = api.field_to_bytes(f.field_id, x)
        assert np.array_equal(b, ol.field_to_bytes(f.field_id, x)) and np.array_equal(api.field_from_bytes(f.field_id, b), x)
        raw = np.array([list(v.to_bytes(8 * f.n_limbs, "little")) for v in (5, f.p, f.p - 1)], dtype=np.uint8)
        with pytest.raises(ValueError, match="Out of range"):
            api.field_from_bytes(f.field_id, raw)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_point_bytes_round_trip_and_errors(c):
    """test_curve_serialization! (serialization.rs:177-210) on the device, decompression (Tonelli-Shanks) included."""
    from plonky_amd import api
    f, L = c.base, c.base.n_limbs
    G = (c.gx, c.gy)
    pts = [br.ec_mul(c, 3 + 977 * k, G) for k in range(40)]
    xy = np.stack([ints_to_array([f.to_mont(P[0]), f.to_mont(P[1])], L) for P in pts] + [np.zeros((2, L), dtype=np.uint64)])
    zero = np.array([0] * len(pts) + [1], dtype=np.uint8)
    b = api.point_to_bytes(c.curve_id, xy, zero)
    assert np.array_equal(b, ol.point_to_bytes(c.curve_id, xy, zero))
    back, bz = api.point_from_bytes(c.curve_id, b)
    assert np.array_equal(bz, zero) and np.array_equal(back, xy)
    bad_x = next(x for x in range(2, 100) if pow((x ** 3 + c.b) % f.p, (f.p - 1) // 2, f.p) == f.p - 1)
    rec = np.array([[0] + list(bad_x.to_bytes(8 * L, "little")), [2] + list(f.p.to_bytes(8 * L, "little")), list(b[0])], dtype=np.uint8)
    with pytest.raises(ValueError):
        api.point_from_bytes(c.curve_id, rec)
    _, _, status = api.point_from_bytes(c.curve_id, rec, with_status=True)
    assert list(status) == [2, 1, 0]


def test_capi_host_nine_threads():
    """tests/capi_host.cpp: a C++ program above the C ABI only, nine host threads calling plk_ntt / plk_msm_precompute /
    plk_msm_execute at once as the reference's Rayon workers do (plonk_util.rs:173-189)."""
    import os, subprocess
    from plonky_amd import lib
    exe = lib.build_host_harness()
    assert exe and os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "capi_host: OK" in r.stdout, r.stdout + r.stderr


def test_msm_context_shared_by_two_streams():
    """One device context, executions enqueued back to back on two different HIP streams (the `_dev` entry points are
    asynchronous): the workspace of the context is handed from one stream to the other through its event, so the second
    execution may not start before the first one's last kernel.  Every result must be the single-stream one."""
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    dev.init(0)
    c = br.TWEEDLEDEE
    n = 1 << 15
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 777, G)
    bases = dev.gen_bases_dev(0, n, _bases(c, [G])[0], _bases(c, [D])[0])
    pre = dev.msm_precompute_dev(0, bases)
    svs = [dev.to_device(synth.rand_field(1, 0x5700 + k, n)) for k in range(6)]
    want = []
    for s in svs:
        xy, z = dev.msm_execute_dev(pre, s)
        torch.cuda.synchronize()
        want.append((dev.to_host(xy).copy(), int(z.cpu()[0])))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(3):
        for k, s in enumerate(svs):
            with torch.cuda.stream(s1 if (k + rep) % 2 == 0 else s2):
                outs.append((k, dev.msm_execute_dev(pre, s)))
    torch.cuda.synchronize()
    for k, (xy, z) in outs:
        assert int(z.cpu()[0]) == want[k][1] and np.array_equal(dev.to_host(xy), want[k][0]), k
    # zero-length polynomials of a batch: every output row is written (ntt_padded_dev)
    ev = dev.ntt_padded_dev(0, torch.empty((3, 0, 4), dtype=torch.int64, device="cuda"), 6)
    assert ev.shape == (3, 64, 4) and not dev.to_host(ev).any()


@pytest.mark.parametrize("field", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("log_n", [0, 1, 5, 12])
def test_fft_precompute_table_matches_oracle(field, log_n):
    """plk_ntt_precompute_table: the reference's FftPrecomputation::subgroups_rev (fft.rs:28-59), layer by layer against the
    oracle's restatement of fft_precompute (orc_fft_table_layer)."""
    import plonky_amd as pa
    from plonky_amd import api
    layers = api.fft_precompute_table(field, 1 << log_n)
    pre = ol.FftPrecomputation(field, 1 << log_n)
    assert len(layers) == log_n + 1
    for i, layer in enumerate(layers):
        assert np.array_equal(layer, pre.layer(i)), (field, log_n, i)


def test_fft_precompute_table_2p20_spot_layers():
    """At the size of the Plonk prover's tables (fft_precompute(8n) at 2^23 is 2 x 256 MiB; here 2^20): first / last layers."""
    from plonky_amd import api
    layers = api.fft_precompute_table(0, 1 << 20)
    pre = ol.FftPrecomputation(0, 1 << 20)
    for i in (0, 1, 2, 10, 19, 20):
        assert np.array_equal(layers[i], pre.layer(i)), i
    with pytest.raises(AssertionError):
        api.fft_precompute_table(5, 1 << 33)  # beyond VestaBase's 2-adicity (field.rs:430)


# ---------------- small fixed-base MSMs without buckets (comb.hip) ----------------
@pytest.mark.parametrize("c", [br.TWEEDLEDEE, br.BLS12_377, br.PALLAS], ids=lambda c: c.name)
def test_msm_comb_small_contexts(c):
    """A tabled context over few generators with an automatic window is a COMB (by size up to 2^12 generators since round 5; PLK_MSM_COMB=1, set
    here, takes it to 2^15; comb.hip: a table of the multiples 1 .. 8 of
    every window's point, executions = mixed additions + a tree): against the oracle and against the bucket method (explicit
    window) - sizes around the lane / block geometry, identity generators, duplicate and opposite generators, edge scalars, batches
    larger than one launch pair takes, sub-ranges of the generators."""
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 0xC0FFEE, G)
    L = c.base.n_limbs
    r = c.scalar.p
    os.environ["PLK_MSM_COMB"] = "1"  # read at every precompute; the bucket contexts below name their window and are unaffected
    try:
        _comb_cases(c, G, D, L, r, torch, dev)
    finally:
        del os.environ["PLK_MSM_COMB"]


def _comb_cases(c, G, D, L, r, torch, dev):
    for n in (1, 2, 3, 31, 32, 33, 257, 2050):
        bases = ol.gen_bases(c.curve_id, n, _bases(c, [G])[0], _bases(c, [D])[0]).reshape(n, 2, L)
        zero = np.zeros(n, dtype=np.uint8)
        scalars = synth.rand_field(c.scalar.field_id, 0xC0B + n, n)
        if n >= 31:
            zero[3] = 1                                                    # AffinePoint::ZERO among the generators
            bases[7] = bases[6]                                            # the doubling branch inside the tree
            scalars[7] = scalars[6]
            bases[9] = np.array([c.base.mont_limbs(G[0]), c.base.mont_limbs(c.base.p - G[1])], dtype=np.uint64)  # -G next to G: cancels
            bases[8] = np.array([c.base.mont_limbs(G[0]), c.base.mont_limbs(G[1])], dtype=np.uint64)
            scalars[9] = scalars[8]
            for k, v in enumerate([0, 1, r - 1, 8, 9, 16, (1 << 252) + 8, r // 2]):
                scalars[10 + k] = mont_arr(c.scalar, [v])[0]
        opre = ol.MsmPrecomputation(c.curve_id, bases, 8, zero=zero, threads=8)
        exp, ez = opre.execute(scalars, parallel=True, threads=8)
        pre = pa.msm_precompute(c.curve_id, bases, 8, zero=zero)          # automatic window: the comb
        assert pre.window == 4
        got, gz = pa.msm_execute_parallel(pre, scalars)
        assert gz == ez and (ez or np.array_equal(got, exp)), (c.name, n)
        buck = pa.msm_precompute(c.curve_id, bases, 8, zero=zero, device_window=7)  # the bucket method on the same data
        b_xy, b_z = pa.msm_execute_parallel(buck, scalars)
        assert b_z == gz and (gz or np.array_equal(b_xy, got)), (c.name, n)
        # a batch of 19 vectors (more than one launch pair of the comb takes), among them an all-zero one
        vecs = np.stack([synth.rand_field(c.scalar.field_id, 0xBA7 + v, n) for v in range(19)])
        vecs[4] = 0
        vecs[11] = scalars
        bxy, bz = pa.msm_execute_batch(pre, vecs)
        assert bz[4] == 1 and int(bz[11]) == ez and (ez or np.array_equal(bxy[11], exp))
        for v in (0, 18):
            e2, z2 = opre.execute(vecs[v], parallel=True, threads=8)
            assert int(bz[v]) == z2 and (z2 or np.array_equal(bxy[v], e2)), (c.name, n, v)
        if n >= 33:
            # sub-ranges of the generators (plk_msm_execute_parts_dev) over a device-resident comb context
            dpre = dev.msm_precompute_dev(c.curve_id, dev.to_device(bases), zero=torch.from_numpy(zero).cuda())
            sd = dev.to_device(scalars)
            parts = [(0, sd), (5, sd[5:n - 2].contiguous()), (n - 1, sd[n - 1:].contiguous()), (12, sd[12:12].contiguous())]
            pxy, pz = dev.msm_execute_parts_dev(dpre, parts)
            torch.cuda.synchronize()
            pxy, pz = dev.to_host(pxy).reshape(4, 2, L), pz.cpu().numpy()
            for k, (f, cnt) in enumerate([(0, n), (5, n - 7), (n - 1, 1), (12, 0)]):
                e3, z3 = ol.MsmPrecomputation(c.curve_id, bases[f:f + cnt], 8, zero=zero[f:f + cnt], threads=8).execute(scalars[f:f + cnt], parallel=True, threads=8) if cnt else (None, 1)
                assert int(pz[k]) == z3 and (z3 or np.array_equal(pxy[k], e3)), (c.name, n, k)
            dpre.free()
        pre.free()
        buck.free()
