"""Pin the oracle's polynomial callers of the NTT (Polynomial::divide_by_z_h,
Polynomial::mul, polynomials_to_values_padded) against exact big-int arithmetic and against the
reference's own unit tests for them (src/polynomial.rs:405-500).  CPU only.
"""
import random

import numpy as np
import pytest

from oracle import bigint_ref as br
from oracle import oracle_lib as ol
from tests.util import array_to_ints, ints_to_array

NTT_FIELDS = [br.TWEEDLEDEE_BASE, br.TWEEDLEDUM_BASE, br.BLS12_377_SCALAR, br.PALLAS_BASE, br.VESTA_BASE]


def mont_arr(f, vals):
    return ints_to_array([f.to_mont(v % f.p) for v in vals], f.n_limbs)


def from_mont_arr(f, arr):
    return [f.from_mont(v) for v in array_to_ints(arr)]


def times_z_h(f, a, n):
    """a * (X^n - 1), canonical ints"""
    out = [0] * (len(a) + n)
    for i, c in enumerate(a):
        out[i + n] = (out[i + n] + c) % f.p
        out[i] = (out[i] - c) % f.p
    return out


# test_division_by_z_h (src/polynomial.rs:469-490): a random, m = a * Z_H, m.divide_by_z_h(n) trimmed == a.
# n is NOT a power of two there, so root^n has full order and every denominator is distinct.
@pytest.mark.parametrize("f", NTT_FIELDS, ids=lambda f: f.name)
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_division_by_z_h(f, seed):
    rng = random.Random(seed * 7919 + f.field_id)
    a_deg = rng.randrange(1, 1500)
    n = rng.randrange(1, a_deg) if a_deg > 1 else 1
    a = br.poly_trim([rng.randrange(f.p) for _ in range(a_deg)])
    m = times_z_h(f, a, n)
    got = ol.poly_divide_by_z_h(f.field_id, mont_arr(f, m), n)
    size = 1 << (len(br.poly_trim(m)) - 1).bit_length()
    assert got.shape[0] == size  # the ifft output is not trimmed (polynomial.rs:369)
    q = from_mont_arr(f, got)
    assert br.poly_trim(q) == a
    assert br.poly_trim(q) == br.poly_divide_by_z_h_exact(f, m, n)


# the Plonk shape: quotient of degree < 8n by Z_H of a power-of-two n (plonk.rs:388-391): only 8 distinct denominators
@pytest.mark.parametrize("f", NTT_FIELDS, ids=lambda f: f.name)
def test_division_by_z_h_power_of_two_n(f):
    rng = random.Random(99 + f.field_id)
    n = 64
    a = [rng.randrange(f.p) for _ in range(7 * n - 3)] + [0, 0, 0]  # trailing zeros are trimmed first
    m = times_z_h(f, a, n)
    got = from_mont_arr(f, ol.poly_divide_by_z_h(f.field_id, mont_arr(f, m), n))
    assert len(got) == 8 * n
    assert br.poly_trim(got) == br.poly_trim(a)


# divide_zero_poly_by_z_h (src/polynomial.rs:492-496): the zero polynomial comes back as it is, untrimmed
def test_divide_zero_poly_by_z_h():
    f = br.TWEEDLEDEE_BASE
    assert ol.poly_divide_by_z_h(f.field_id, np.zeros((0, 4), dtype=np.uint64), 16).shape == (0, 4)
    z = ol.poly_divide_by_z_h(f.field_id, np.zeros((5, 4), dtype=np.uint64), 16)
    assert z.shape == (5, 4) and not z.any()


# test_polynomial_multiplication (src/polynomial.rs:405-419), against the schoolbook product
@pytest.mark.parametrize("f", NTT_FIELDS, ids=lambda f: f.name)
def test_polynomial_multiplication(f):
    rng = random.Random(5 + f.field_id)
    a = [rng.randrange(f.p) for _ in range(rng.randrange(1, 200))]
    b = [rng.randrange(f.p) for _ in range(rng.randrange(1, 200))] + [0, 0]
    got = from_mont_arr(f, ol.poly_mul(f.field_id, mont_arr(f, a), mont_arr(f, b)))
    want = br.poly_mul_schoolbook(f, br.poly_trim(a), br.poly_trim(b))
    assert len(got) == 1 << (len(want) - 1).bit_length()
    assert br.poly_trim(got) == br.poly_trim(want)
    for _ in range(10):
        x = rng.randrange(f.p)
        assert br.poly_eval(f, got, x) == br.poly_eval(f, a, x) * br.poly_eval(f, b, x) % f.p
    # a zero operand gives Polynomial::zero(1) (polynomial.rs:209-211)
    z = ol.poly_mul(f.field_id, mont_arr(f, a), np.zeros((3, f.n_limbs), dtype=np.uint64))
    assert z.shape == (1, f.n_limbs) and not z.any()


# polynomials_to_values_padded (src/plonk_util.rs:179-190): pad to 8x, evaluate on the 8n domain
@pytest.mark.parametrize("f", NTT_FIELDS, ids=lambda f: f.name)
def test_polynomials_to_values_padded(f):
    rng = random.Random(17 + f.field_id)
    n = 32
    pre = ol.FftPrecomputation(f.field_id, 8 * n)
    for length in (n, n - 5, 1):
        a = [rng.randrange(f.p) for _ in range(length)]
        got = from_mont_arr(f, ol.poly_to_values_padded(pre, mont_arr(f, a)))
        want = br.ntt(f, a + [0] * (8 * n - length))
        assert got == want
    with pytest.raises(ValueError):
        ol.poly_to_values_padded(pre, mont_arr(f, [1] * (n + 1)))
