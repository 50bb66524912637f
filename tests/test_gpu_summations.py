"""GPU parity: affine (multi)summation (curve_summations.rs) directly, not only through the MSM that replaces its call site.

The reference's two unit tests (curve_summations.rs:164-184) read as they do there, with the library's mirror in place of
affine_summation_pairwise / affine_summation_batch_inversion; then seeded lists of every length around the reference's switch
between its two forms (70 pair sums, curve_summations.rs:29), with identity operands, repeated points (the doubling branch,
:86-92) and opposite points (:113-141), against all three forms of the oracle - which must agree with each other - and against
big-integer group arithmetic; k independent lists through affine_multisummation_best.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import plonky_amd as pa
from oracle import bigint_ref as br
from oracle import oracle_lib as ol
from tests.test_oracle_kats import from_mont_arr

CURVES = [br.TWEEDLEDEE, br.TWEEDLEDUM, br.BLS12_377, br.PALLAS, br.VESTA]


def _arr(c, pts):
    return np.array([[c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])] for P in pts], dtype=np.uint64).reshape(-1, 2, c.base.n_limbs)


def _same(c, got, P):
    """(limbs, zero flag) against an affine big-integer point or None for the identity."""
    out, z = got
    if P is None:
        return z == 1 and not out.any()
    return z == 0 and tuple(from_mont_arr(c.base, out)) == P


def test_pairwise_affine_summation():
    """curve_summations.rs:164-175."""
    c = br.BLS12_377
    g = (c.gx, c.gy)
    g2, g3 = br.ec_mul(c, 2, g), br.ec_mul(c, 3, g)
    assert _same(c, pa.affine_summation_best(pa.BLS12_377, _arr(c, [g, g])), g2)
    assert _same(c, pa.affine_summation_best(pa.BLS12_377, _arr(c, [g, g2])), g3)
    assert _same(c, pa.affine_summation_best(pa.BLS12_377, _arr(c, [g, g, g])), g3)
    assert _same(c, pa.affine_summation_best(pa.BLS12_377, _arr(c, [])), None)


def test_pairwise_affine_summation_batch_inversion():
    """curve_summations.rs:177-184."""
    c = br.BLS12_377
    g = (c.gx, c.gy)
    assert _same(c, pa.affine_summation_best(pa.BLS12_377, _arr(c, [g, g])), br.ec_mul(c, 2, g))
    assert _same(c, pa.affine_summation_best(pa.BLS12_377, _arr(c, [g, g, g])), br.ec_mul(c, 3, g))
    assert _same(c, pa.affine_summation_best(pa.BLS12_377, _arr(c, [])), None)


def _seeded_list(c, n, seed):
    """n points k_i G with small seeded k_i (so that big-integer arithmetic can name the sum), every fifth one a repeat of its
    neighbour, every seventh the opposite of its neighbour, every eleventh flagged as the identity."""
    rng = np.random.default_rng(seed)
    G = (c.gx, c.gy)
    ks = [int(v) for v in rng.integers(1, 1 << 20, size=n)]
    zero = np.zeros(n, dtype=np.uint8)
    for i in range(1, n):
        if i % 5 == 0:
            ks[i] = ks[i - 1]
        elif i % 7 == 0:
            ks[i] = -ks[i - 1]
        if i % 11 == 0:
            zero[i] = 1
    pts = [br.ec_mul(c, abs(k), G) if k > 0 else br.ec_neg(c, br.ec_mul(c, abs(k), G)) for k in ks]
    total = sum(k for k, z in zip(ks, zero) if not z) % c.scalar.p
    return _arr(c, pts), zero, (br.ec_mul(c, total, G) if total else None)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_seeded_summations_match_oracle_and_big_integers(c):
    for n in (0, 1, 2, 3, 4, 7, 63, 64, 65, 69, 70, 71, 72, 139, 140, 141, 142, 300, 1000):
        pts, zero, expected = _seeded_list(c, n, 0x5A00 + n)
        got = pa.affine_summation_best(c.curve_id, pts, zero)
        assert _same(c, got, expected), n
        for mode in ("pairwise", "batch_inversion", "best"):  # the reference's three forms: one group element
            out, z = ol.affine_summation(c.curve_id, mode, pts, zero=zero)
            assert z == got[1] and np.array_equal(out, got[0]), (n, mode)
    # the whole list cancels: P + (-P) pairs only
    G = (c.gx, c.gy)
    P = [br.ec_mul(c, 3 + i, G) for i in range(40)]
    pts = _arr(c, [q for p in P for q in (p, br.ec_neg(c, p))])
    assert _same(c, pa.affine_summation_best(c.curve_id, pts), None)


@pytest.mark.parametrize("c", [br.TWEEDLEDEE, br.BLS12_377], ids=lambda c: c.name)
def test_affine_multisummation_best(c):
    """curve_summations.rs:24-35: k lists -> k sums (lengths either side of the switch, an empty list among them)."""
    lens = [0, 1, 5, 69, 70, 71, 200, 2]
    cases = [_seeded_list(c, n, 0x5B00 + 17 * n) for n in lens]
    got = pa.affine_multisummation_best(c.curve_id, [p for p, _, _ in cases], zeros=[z for _, z, _ in cases])
    assert len(got) == len(lens)
    for n, g, (p, z, e) in zip(lens, got, cases):
        assert _same(c, g, e), n
        out, oz = ol.affine_summation(c.curve_id, "best", p, zero=z)
        assert oz == g[1] and np.array_equal(out, g[0]), n
