"""GPU parity of one round of the inner-product argument (src/halo.rs:63-124): the L / R terms, the
scalar folds and the generator fold on the device against the oracle's composition of the restated primitives, and a whole
argument (log n rounds) against the closed form G_final = <s, G>, a_final = <s', a> of the folded vectors.
The challenges u_j and the blinding factors are inputs (the transcript and the RNG are outside the hot path)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bigint_ref as br
from oracle import oracle_lib as ol
from tests.test_oracle_kats import from_mont_arr, mont_arr
from tests.util import limbs_to_int

CURVES = [br.TWEEDLEDEE, br.TWEEDLEDUM, br.BLS12_377]


def _setup(c, n, seed):
    G = (c.gx, c.gy)
    L = c.base.n_limbs
    pt = lambda P: np.array([c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])], dtype=np.uint64)
    g = ol.gen_bases(c.curve_id, n, pt(G), pt(br.ec_mul(c, 0x9E3779B9 + seed, G)))
    h, up = pt(br.ec_mul(c, 0xABCDEF01 + seed, G)), pt(br.ec_mul(c, 0x13579BDF + seed, G))
    a = ol.rand_field(c.scalar.field_id, seed + 1, n)
    b = ol.rand_field(c.scalar.field_id, seed + 2, n)
    sc = ol.rand_field(c.scalar.field_id, seed + 3, 3)
    return g.reshape(n, 2, L), h, up, a, b, sc


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [2, 16, 256])
def test_halo_round_matches_oracle(c, n):
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    dev.init(0)
    g, h, up, a, b, sc = _setup(c, n, 1000 + n)
    l_blind, r_blind, u = sc
    u_inv = ol.field_unop(c.scalar.field_id, "inverse", u.reshape(1, 4))[0]
    da, db, dg = dev.to_device(a), dev.to_device(b), dev.to_device(g)
    lr, lrz = dev.halo_round_lr_dev(c.curve_id, da, db, dg, h, up, l_blind, r_blind)
    exp, ez = ol.halo_round_lr(c.curve_id, c.scalar.field_id, a, b, g, h, up, l_blind, r_blind)
    assert list(lrz.cpu().numpy()) == ez and np.array_equal(dev.to_host(lr), exp)
    a2, b2, g2, gz2 = dev.halo_round_fold_dev(c.curve_id, da, db, dg, u, u_inv)
    ea, eb, eg, egz = ol.halo_round_fold(c.curve_id, c.scalar.field_id, a, b, g, u, u_inv)
    assert np.array_equal(dev.to_host(a2), ea) and np.array_equal(dev.to_host(b2), eb)
    assert np.array_equal(gz2.cpu().numpy(), egz) and np.array_equal(dev.to_host(g2), eg)


def test_halo_whole_argument_closed_form():
    """log n rounds on the device; then G_final = sum_i s_i G_i and a_final = sum_i s'_i a_i where s_i = prod_j u_j^(+-1) by the
    bits of i (a fold keeps lo with u^-1 for G / u for a, hi with u for G / u^-1 for a) - checked with Python integers."""
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    dev.init(0)
    c = br.TWEEDLEDEE
    f = c.scalar
    n = 1 << 10
    g, h, up, a, b, _ = _setup(c, n, 7)
    us = ol.rand_field(f.field_id, 99, 10)
    u_ints = [f.from_mont(limbs_to_int(r)) for r in us]
    da, db, dg, dz = dev.to_device(a), dev.to_device(b), dev.to_device(g), None
    for j in range(10):
        u_inv = ol.field_unop(f.field_id, "inverse", us[j].reshape(1, 4))[0]
        da, db, dg, dz = dev.halo_round_fold_dev(c.curve_id, da, db, dg, us[j], u_inv, dz)
    assert da.shape[0] == 1 and int(dz.cpu()[0]) == 0
    # closed form: after round j (halving from the top bit down), index bit (9 - j) selects hi
    s_g, s_a = [], []
    for i in range(n):
        sg = sa = 1
        for j in range(10):
            hi = (i >> (9 - j)) & 1
            uj, uinv = u_ints[j], pow(u_ints[j], -1, f.p)
            sg = sg * (uj if hi else uinv) % f.p
            sa = sa * (uinv if hi else uj) % f.p
        s_g.append(sg)
        s_a.append(sa)
    a_ints = [f.from_mont(limbs_to_int(r)) for r in a]
    assert f.from_mont(limbs_to_int(dev.to_host(da)[0])) == sum(x * y for x, y in zip(s_a, a_ints)) % f.p
    exp, ez = ol.MsmPrecomputation(c.curve_id, g, 8, threads=8).execute(mont_arr(f, s_g), parallel=True, threads=8)
    assert ez == 0 and np.array_equal(dev.to_host(dg)[0], exp)
