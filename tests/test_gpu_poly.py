"""GPU parity tests of the polynomial callers of the NTT: the HIP path through the
C ABI against the oracle's restatement of src/polynomial.rs / src/plonk_util.rs, bit for bit, and against
exact big-int division.  Shapes follow the reference's own tests (src/polynomial.rs:405-500).
"""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import plonky_amd as pa
from plonky_amd import synth
from oracle import bigint_ref as br
from oracle import oracle_lib as ol
from tests.test_oracle_poly import from_mont_arr, mont_arr, times_z_h

NTT_FIELDS = [br.TWEEDLEDEE_BASE, br.TWEEDLEDUM_BASE, br.BLS12_377_SCALAR, br.PALLAS_BASE, br.VESTA_BASE]


def mul_by_z_h_mont(f, a, n):
    """a (Montgomery limbs) times X^n - 1 with the oracle's field ops"""
    ln = a.shape[0]
    out = np.zeros((ln + n, f.n_limbs), dtype=np.uint64)
    out[n:] = a
    lo = ol.field_binop(f.field_id, "sub", out[:ln].copy(), a)
    out[:ln] = lo
    return out


# test_division_by_z_h (polynomial.rs:469-490): n random, not a power of two -> every denominator distinct
@pytest.mark.parametrize("f", NTT_FIELDS, ids=lambda f: f.name)
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_division_by_z_h(f, seed):
    rng = random.Random(seed * 104729 + f.field_id)
    a_deg = rng.randrange(1, 10_000)
    n = rng.randrange(1, a_deg) if a_deg > 1 else 1
    a = synth.rand_field(f.field_id, 0xD1F0000 + seed, a_deg)
    m = mul_by_z_h_mont(f, a, n)
    got = pa.polynomial_divide_by_z_h(f.field_id, m, n)
    want = ol.poly_divide_by_z_h(f.field_id, m, n)
    assert got.shape == want.shape and np.array_equal(got, want)
    # the reference's own assertion: trimmed quotient == a
    k = got.shape[0]
    while k and not got[k - 1].any():
        k -= 1
    assert np.array_equal(got[:k], a)


# the Plonk shape (plonk.rs:388-391): degree < 8n, Z_H of a power-of-two n: 8 distinct denominators
@pytest.mark.parametrize("f", NTT_FIELDS, ids=lambda f: f.name)
@pytest.mark.parametrize("log_n", [3, 7, 10, 13])
def test_division_by_z_h_plonk_shape(f, log_n):
    n = 1 << log_n
    a = synth.rand_field(f.field_id, 0xD2F0000 + log_n, 7 * n - 5)
    m = np.concatenate([mul_by_z_h_mont(f, a, n), np.zeros((5, f.n_limbs), dtype=np.uint64)])  # trailing zeros get trimmed
    got = pa.polynomial_divide_by_z_h(f.field_id, m, n)
    assert got.shape[0] == 8 * n
    assert np.array_equal(got, ol.poly_divide_by_z_h(f.field_id, m, n, threads=8))
    assert np.array_equal(got[: a.shape[0]], a) and not got[a.shape[0]:].any()


def test_division_by_z_h_small_and_odd_cases():
    f = br.TWEEDLEDEE_BASE
    # degree 0 numerator is not divisible, the reference still returns a/(g^n - 1): compare with the oracle
    for coeffs, n in ([7], 3), ([0, 0, 5], 1), ([1, 2, 3, 4, 5], 2), (list(range(1, 18)), 16), (list(range(1, 18)), 32), ([3] * 1025, 1024):
        m = mont_arr(f, coeffs)
        assert np.array_equal(pa.polynomial_divide_by_z_h(f.field_id, m, n), ol.poly_divide_by_z_h(f.field_id, m, n)), (coeffs[:4], n)
    # n half the domain size: two denominators
    a = synth.rand_field(f.field_id, 77, 40)
    m = mul_by_z_h_mont(f, a, 128)
    assert np.array_equal(pa.polynomial_divide_by_z_h(f.field_id, m, 128), ol.poly_divide_by_z_h(f.field_id, m, 128))


# divide_zero_poly_by_z_h (polynomial.rs:492-496)
def test_divide_zero_poly_by_z_h():
    f = br.TWEEDLEDEE_BASE
    assert pa.polynomial_divide_by_z_h(f.field_id, np.zeros((0, 4), dtype=np.uint64), 16).shape == (0, 4)
    z = pa.polynomial_divide_by_z_h(f.field_id, np.zeros((5, 4), dtype=np.uint64), 16)
    assert z.shape == (5, 4) and not z.any()


# test_polynomial_multiplication (polynomial.rs:405-419)
@pytest.mark.parametrize("f", NTT_FIELDS, ids=lambda f: f.name)
@pytest.mark.parametrize("seed", [1, 2])
def test_polynomial_multiplication(f, seed):
    rng = random.Random(seed * 31 + f.field_id)
    a = synth.rand_field(f.field_id, 0xA000 + seed, rng.randrange(1, 10_000))
    b = synth.rand_field(f.field_id, 0xB000 + seed, rng.randrange(1, 10_000))
    got = pa.polynomial_mul(f.field_id, a, b)
    want = ol.poly_mul(f.field_id, a, b, threads=8)
    assert got.shape == want.shape and np.array_equal(got, want)
    ai, bi, gi = from_mont_arr(f, a), from_mont_arr(f, b), from_mont_arr(f, got)
    for _ in range(5):
        x = rng.randrange(f.p)
        assert br.poly_eval(f, gi, x) == br.poly_eval(f, ai, x) * br.poly_eval(f, bi, x) % f.p


def test_polynomial_multiplication_edge_cases():
    f = br.TWEEDLEDUM_BASE
    a = synth.rand_field(f.field_id, 5, 9)
    z = pa.polynomial_mul(f.field_id, a, np.zeros((3, 4), dtype=np.uint64))
    assert z.shape == (1, 4) and not z.any()
    z = pa.polynomial_mul(f.field_id, np.zeros((0, 4), dtype=np.uint64), a)
    assert z.shape == (1, 4) and not z.any()
    one = mont_arr(f, [1])
    assert np.array_equal(pa.polynomial_mul(f.field_id, a, one), ol.poly_mul(f.field_id, a, one))
    c = mont_arr(f, [5, 0, 0])  # trailing zeros: degree 0
    assert np.array_equal(pa.polynomial_mul(f.field_id, c, c), ol.poly_mul(f.field_id, c, c))


# polynomials_to_values_padded (plonk_util.rs:179-190): the 9 wire polynomials on the 8n domain
@pytest.mark.parametrize("f", NTT_FIELDS, ids=lambda f: f.name)
@pytest.mark.parametrize("log_n", [4, 9, 12])
def test_polynomials_to_values_padded(f, log_n):
    n = 1 << log_n
    pre = pa.fft_precompute(f.field_id, 8 * n)
    opre = ol.FftPrecomputation(f.field_id, 8 * n)
    polys = [synth.rand_field(f.field_id, 0xC000 + 16 * log_n + k, n if k % 3 else n - k) for k in range(9)]
    got = pa.polynomials_to_values_padded(polys, pre)
    assert got.shape == (9, 8 * n, 4)
    for k in range(9):
        assert np.array_equal(got[k], ol.poly_to_values_padded(opre, polys[k], threads=8)), k
    # values_to_polynomials (plonk_util.rs:169-177) brings the padded coefficients back
    back = pa.values_to_polynomials(got, pre)
    for k in range(9):
        assert np.array_equal(back[k][: polys[k].shape[0]], polys[k]) and not back[k][polys[k].shape[0]:].any()


def test_padded_domain_too_small_is_rejected():
    f = br.TWEEDLEDEE_BASE
    pre = pa.fft_precompute(f.field_id, 64)
    with pytest.raises(AssertionError):
        pa.polynomials_to_values_padded([synth.rand_field(f.field_id, 1, 9)], pre)


# device-resident variants, 2^20 quotient shape: divide_by_z_h of a degree < 8n polynomial with n = 2^17,
# checked through the algebra (q * Z_H == m) and against the host-pointer entry point
def test_device_resident_divide_by_z_h_large():
    import torch
    from plonky_amd import device as dv

    f = br.TWEEDLEDEE_BASE
    dv.init()
    log_n = 17
    n = 1 << log_n
    q = synth.rand_field(f.field_id, 0xE000, 7 * n)
    m = mul_by_z_h_mont(f, q, n)
    d_m = dv.to_device(m)
    d_q = dv.divide_by_z_h_dev(f.field_id, d_m, n)
    torch.cuda.synchronize()
    got = dv.to_host(d_q)
    assert got.shape[0] == 8 * n
    assert np.array_equal(got[: 7 * n], q) and not got[7 * n:].any()
    # second call hits the cached tables and may alias input and output
    d_buf = torch.zeros((8 * n, 4), dtype=torch.int64, device="cuda")
    d_buf[: m.shape[0]] = d_m
    d_q2 = dv.divide_by_z_h_dev(f.field_id, d_buf, n, out=d_buf)
    torch.cuda.synchronize()
    assert np.array_equal(dv.to_host(d_q2), got)
    # LDE on device: 9 wires of n coefficients -> 8n evaluations, against the plain transform of the padded data
    w = synth.rand_field(f.field_id, 0xE100, 9 * n).reshape(9, n, 4)
    d_w = dv.to_device(w)
    ev = dv.ntt_padded_dev(f.field_id, d_w, log_n + 3)
    padded = torch.zeros((9, 8 * n, 4), dtype=torch.int64, device="cuda")
    padded[:, :n] = d_w
    ref = dv.ntt_dev(f.field_id, padded)
    torch.cuda.synchronize()
    assert torch.equal(ev, ref)
    # product on device
    a, b = synth.rand_field(f.field_id, 1, 3000), synth.rand_field(f.field_id, 2, 5000)
    d_p = dv.poly_mul_dev(f.field_id, dv.to_device(a), dv.to_device(b))
    torch.cuda.synchronize()
    assert np.array_equal(dv.to_host(d_p), pa.polynomial_mul(f.field_id, a, b))


def test_error_behaviour_of_the_new_entry_points():
    """Contract violations come back as error codes with a message, never as aborts (include/plonky_hip.h)."""
    import ctypes

    from plonky_amd import lib

    L = lib.load()
    f = br.TWEEDLEDEE_BASE
    a = synth.rand_field(f.field_id, 1, 40)
    out = np.zeros((64, 4), dtype=np.uint64)
    n_out = ctypes.c_size_t(0)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    # Z_H of n = 0 is the zero polynomial
    assert L.plk_poly_divide_by_z_h(f.field_id, p(a), 40, 0, p(out), 64, ctypes.byref(n_out)) == lib.PLK_ERR_INVALID_ARG
    assert b"X^0" in L.plk_last_error()
    # output too small for the 2^6 result
    assert L.plk_poly_divide_by_z_h(f.field_id, p(a), 40, 3, p(out), 8, ctypes.byref(n_out)) == lib.PLK_ERR_INVALID_ARG
    # a field without NTT entry points (Bls12377Base) and a bad id
    assert L.plk_poly_mul(3, p(a), 4, p(a), 4, p(out), 64, ctypes.byref(n_out)) == lib.PLK_ERR_INVALID_ARG
    assert L.plk_poly_mul(9, p(a), 4, p(a), 4, p(out), 64, ctypes.byref(n_out)) == lib.PLK_ERR_INVALID_ARG
    # padded transform: more coefficients than the domain holds
    ins = (ctypes.c_void_p * 1)(a.ctypes.data)
    lens = (ctypes.c_size_t * 1)(40)
    outs = (ctypes.c_void_p * 1)(out.ctypes.data)
    assert L.plk_ntt_padded_batch(f.field_id, 5, 1, ins, lens, outs) == lib.PLK_ERR_INVALID_ARG
    assert L.plk_ntt_padded_batch(f.field_id, 6, 1, ins, lens, outs) == lib.PLK_OK
    # table-free MSM: a window whose bucket ranges do not fit the partition
    c = br.TWEEDLEDEE
    g = mont_arr(c.base, [c.gx, c.gy]).reshape(1, 2, 4)
    ctx = ctypes.c_void_p()
    assert L.plk_msm_precompute_ex(0, 1, p(g), None, 17, 1, ctypes.byref(ctx)) == lib.PLK_ERR_INVALID_ARG  # table-free: windows <= 16 bits
    assert L.plk_msm_precompute_ex(0, 1, p(g), None, 40, 0, ctypes.byref(ctx)) == lib.PLK_ERR_INVALID_ARG
    # reference-layout table: window 0
    assert L.plk_msm_table_digits(0, 0) == lib.PLK_ERR_INVALID_ARG
    assert L.plk_msm_table_digits(0, 11) == 24 and L.plk_msm_table_digits(2, 11) == 23  # ceil(255 / 11), ceil(253 / 11)
    # fold: null scalar
    oz = np.zeros(1, dtype=np.uint8)
    assert L.plk_curve_fold_pairs(0, 1, p(g), None, p(g), None, None, None, p(g.copy()), p(oz)) == lib.PLK_ERR_INVALID_ARG
