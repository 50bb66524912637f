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


# ---- the argument behind the C ABI (plk_halo_*): device-resident vectors, frozen generators for the short rounds ----
def _run_argument(c, n, freeze_log, seed, tabled=False, lead_rounds=0, extra_generators=0, inside=False):
    """All rounds through plk_halo_*; returns the per-round (L_j, R_j) and the final (a, b, g).  tabled: the first rounds run over
    the caller's commitment tables (plk_halo_begin_tabled_dev), which may hold more generators than the argument uses; inside:
    pedersen_h and the fixed generator U are among them, u_prime = [x] U."""
    from plonky_amd import device as dev
    g, h, up, a, b, _ = _setup(c, n, seed)
    f = c.scalar
    rounds = n.bit_length() - 1
    us = ol.rand_field(f.field_id, seed + 50, max(rounds, 1))
    blind = ol.rand_field(f.field_id, seed + 60, 2 * max(rounds, 1))
    tables, kw = None, {}
    if tabled:
        L = c.base.n_limbs
        more = ol.gen_bases(c.curve_id, extra_generators, h, up).reshape(extra_generators, 2, L) if extra_generators else np.zeros((0, 2, L), dtype=np.uint64)
        gens = [g, more]
        if inside:
            G = (c.gx, c.gy)
            pt = lambda P: np.array([c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])], dtype=np.uint64)
            ubase = br.ec_mul(c, 0x2468ACE + seed, G)
            x_int = (0x1234567 + seed) * 0x9E3779B97F4A7C15F39CC0605CEDC835 % f.p
            up = pt(br.ec_mul(c, x_int, ubase))
            gens += [h.reshape(1, 2, L), pt(ubase).reshape(1, 2, L)]
            kw = dict(h_index=n + extra_generators, u_index=n + extra_generators + 1, u_prime_scalar=np.array(f.mont_limbs(x_int), dtype=np.uint64))
        tables = dev.msm_precompute_dev(c.curve_id, dev.to_device(np.concatenate(gens)))
    arg = dev.HaloArgument(c.curve_id, dev.to_device(a), dev.to_device(b), dev.to_device(g), h, up, freeze_log=freeze_log, tables=tables,
                           lead_rounds=lead_rounds, **kw)
    lrs, states = [], []
    for j in range(rounds):
        lr, z = arg.round_lr(blind[2 * j], blind[2 * j + 1])
        lr2, z2 = arg.round_lr(blind[2 * j], blind[2 * j + 1])   # the retry of halo.rs:83-114 must not disturb the state
        assert np.array_equal(lr, lr2) and np.array_equal(z, z2)
        lrs.append((lr, z))
        u_inv = ol.field_unop(f.field_id, "inverse", us[j].reshape(1, 4))[0]
        arg.round_fold(us[j], u_inv)
        assert len(arg) == n >> (j + 1)
        # the state after the fold: halo_a, halo_b always; halo_g while it is still folded explicitly (kept scaled inside)
        states.append((arg.frozen, arg.read(with_g=not arg.frozen)))
    fa, fb, fg, fgz = arg.read()
    arg.free()
    return (g, h, up, a, b, us, blind), lrs, states, (fa, fb, fg, fgz)


def _oracle_argument(c, inputs):
    g, h, up, a, b, us, blind = inputs
    f = c.scalar
    n = a.shape[0]
    lrs, states = [], []
    gz = np.zeros(n, dtype=np.uint8)
    for j in range(n.bit_length() - 1):
        exp, ez = ol.halo_round_lr(c.curve_id, f.field_id, a, b, g, h, up, blind[2 * j], blind[2 * j + 1])
        lrs.append((exp, ez))
        u_inv = ol.field_unop(f.field_id, "inverse", us[j].reshape(1, 4))[0]
        a, b, g, gz = ol.halo_round_fold(c.curve_id, f.field_id, a, b, g, us[j], u_inv)
        states.append((a, b, g, gz))
    return lrs, states, (a, b, g, gz)


@pytest.fixture(params=[None, "1"], ids=["stages_default", "stages_small"])
def stage_floor(request, monkeypatch):
    """PLK_HALO_STAGE_MIN_LOG (read when an argument begins): by default a stage of virtual rounds over the explicit generator set
    needs 2^17 outputs - out of reach of an oracle-checked size; "1" lets the short vectors of these tests run in stages too."""
    if request.param is not None:
        monkeypatch.setenv("PLK_HALO_STAGE_MIN_LOG", request.param)
    return request.param


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n,freeze_log", [(2, 0), (64, 0), (64, 3), (256, 5), (256, 63), (1, 0)])
def test_halo_argument_capi_matches_oracle(c, n, freeze_log, stage_floor):
    """Every round's L_j / R_j and the final (halo_a, halo_b, halo_g) against the oracle's composition of the reference's
    primitives - with the generators folded explicitly all the way (freeze_log 63: never frozen... until 2 are left), frozen
    from the start (default 2^14 >= n) and frozen part-way (2^3, 2^5)."""
    pytest.importorskip("torch")
    from plonky_amd import device as dev
    dev.init(0)
    inputs, lrs, states, (fa, fb, fg, fgz) = _run_argument(c, n, freeze_log if freeze_log != 63 else 1, 4000 + n)
    exp_lrs, exp_states, (ea, eb, eg, egz) = _oracle_argument(c, inputs)
    for j, ((lr, z), (elr, ez)) in enumerate(zip(lrs, exp_lrs)):
        assert list(z) == list(ez) and np.array_equal(lr, elr), "round %d" % j
    for j, ((frozen, got), exp) in enumerate(zip(states, exp_states)):
        assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]), "halo_a / halo_b after round %d" % j
        if not frozen:
            assert np.array_equal(got[2], exp[2]) and np.array_equal(got[3], np.asarray(exp[3], dtype=np.uint8)), "halo_g after round %d" % j
    assert np.array_equal(fa, ea) and np.array_equal(fb, eb)
    assert np.array_equal(fgz, np.asarray(egz, dtype=np.uint8)) and np.array_equal(fg, eg)
    if n >= 4 and freeze_log in (3, 5):
        assert states[-1][0]   # frozen generators at the end
        # before that: pairwise folds (halo_g after every round); with the small stage floor: stages of two virtual rounds on the
        # curves with the endomorphism
        assert states[0][0] == (c is not br.BLS12_377 and stage_floor is not None) and not states[1][0]


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("inside", [False, True], ids=["hu_beside", "hu_inside"])
@pytest.mark.parametrize("n,freeze_log,lead,extra", [(256, 3, 0, 0), (256, 3, 4, 1), (256, 5, 0, 3), (64, 1, 1, 0), (128, 2, 2, 0), (16, 63, 0, 0), (4, 0, 0, 0)])
def test_halo_argument_over_the_callers_tables(c, n, freeze_log, lead, extra, inside, stage_floor):
    """plk_halo_begin_tabled_dev: the first rounds over the caller's commitment tables (one batched MSM with challenge-expanded
    scalars + the H / U' terms), the generators of those rounds folded at once (2^r-to-1), then the usual regimes - every
    L_j / R_j, halo_a / halo_b after every round, halo_g whenever it exists and the final triple, bit for bit against the oracle.
    BLS12-377 (no endomorphism) and short vectors fall back to the plain path.  hu_inside: pedersen_h and U are generators of the
    tables (u_prime = [x] U), their terms are two more scalars of the same MSM."""
    pytest.importorskip("torch")
    from plonky_amd import device as dev
    dev.init(0)
    inputs, lrs, states, (fa, fb, fg, fgz) = _run_argument(c, n, freeze_log if freeze_log != 63 else 1, 6000 + n + lead, tabled=True, lead_rounds=lead,
                                                           extra_generators=extra, inside=inside)
    exp_lrs, exp_states, (ea, eb, eg, egz) = _oracle_argument(c, inputs)
    for j, ((lr, z), (elr, ez)) in enumerate(zip(lrs, exp_lrs)):
        assert list(z) == list(ez) and np.array_equal(lr, elr), "round %d" % j
    seen_g = 0
    for j, ((frozen, got), exp) in enumerate(zip(states, exp_states)):
        assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]), "halo_a / halo_b after round %d" % j
        if not frozen:
            seen_g += 1
            assert np.array_equal(got[2], exp[2]) and np.array_equal(got[3], np.asarray(exp[3], dtype=np.uint8)), "halo_g after round %d" % j
    assert np.array_equal(fa, ea) and np.array_equal(fb, eb)
    assert np.array_equal(fgz, np.asarray(egz, dtype=np.uint8)) and np.array_equal(fg, eg)


@pytest.mark.parametrize("c", [br.TWEEDLEDEE, br.TWEEDLEDUM], ids=lambda c: c.name)
@pytest.mark.parametrize("r,n_out", [(1, 1), (1, 37), (2, 5), (3, 64), (4, 3)])
def test_fold_multi_matches_big_integers(c, r, n_out):
    """plk_curve_fold_multi_dev: out_i = g_i + sum_t [s_t] g_{i + t n_out} against Python integers - random points, identity
    inputs, the same point in every slot (additions that meet doublings and inverses) and the edge scalars 0, 1, r - 1."""
    pytest.importorskip("torch")
    from plonky_amd import device as dev
    dev.init(0)
    f = c.scalar
    T = 1 << r
    n = T * n_out
    G = (c.gx, c.gy)
    L = c.base.n_limbs
    rng = np.random.default_rng(900 + 17 * r + n_out)
    ks = [int(x) for x in rng.integers(1, 1 << 62, size=n)]
    for i in range(0, n, 7):
        ks[i] = ks[0]                       # repeated points
    pts = [br.ec_mul(c, k, G) for k in ks]
    zero = np.zeros(n, dtype=np.uint8)
    zero[3 % n] = 1
    if n > 8:
        zero[n - 1] = 1
    g = np.array([[c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])] for P in pts], dtype=np.uint64).reshape(n, 2, L)
    sc_ints = [int.from_bytes(rng.bytes(32), "little") % f.p for _ in range(T)]
    for t, v in zip(range(1, T), [0, 1, f.p - 1]):
        sc_ints[t] = v
    rev = lambda t: int(format(t, "0%db" % r)[::-1], 2)
    sc = np.zeros((T, 4), dtype=np.uint64)
    for t in range(T):
        sc[rev(t)] = f.mont_limbs(sc_ints[t])
    out, oz = dev.fold_generators_multi_dev(c.curve_id, dev.to_device(g), dev.to_device(sc), r, g_zero=__import__("torch").from_numpy(zero).cuda())
    out, oz = dev.to_host(out), oz.cpu().numpy()
    for i in range(n_out):
        acc = None if zero[i] else pts[i]
        for t in range(1, T):
            j = i + t * n_out
            if zero[j] or sc_ints[t] == 0:
                continue
            term = br.ec_mul(c, sc_ints[t], pts[j])
            acc = br.ec_add(c, acc, term)
        if acc is None:
            assert oz[i] == 1, i
        else:
            assert oz[i] == 0 and tuple(from_mont_arr(c.base, out[i])) == acc, i


@pytest.mark.parametrize("c", [br.PALLAS, br.VESTA], ids=lambda c: c.name)
def test_halo_argument_stages_on_pallas_and_vesta(c, monkeypatch):
    """The two other curves with the endomorphism through the same stages: a tabled first stage with H / U inside the tables, a second
    stage over the explicit generators (small stage floor), pairwise folds, frozen generators - against the oracle, round by round."""
    pytest.importorskip("torch")
    from plonky_amd import device as dev
    dev.init(0)
    monkeypatch.setenv("PLK_HALO_STAGE_MIN_LOG", "1")
    n = 128
    inputs, lrs, states, (fa, fb, fg, fgz) = _run_argument(c, n, 2, 7100, tabled=True, lead_rounds=2, extra_generators=1, inside=True)
    exp_lrs, exp_states, (ea, eb, eg, egz) = _oracle_argument(c, inputs)
    for j, ((lr, z), (elr, ez)) in enumerate(zip(lrs, exp_lrs)):
        assert list(z) == list(ez) and np.array_equal(lr, elr), "round %d" % j
    for j, ((frozen, got), exp) in enumerate(zip(states, exp_states)):
        assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]), "halo_a / halo_b after round %d" % j
        if not frozen:
            assert np.array_equal(got[2], exp[2]) and np.array_equal(got[3], np.asarray(exp[3], dtype=np.uint8)), "halo_g after round %d" % j
    assert np.array_equal(fa, ea) and np.array_equal(fb, eb)
    assert np.array_equal(fgz, np.asarray(egz, dtype=np.uint8)) and np.array_equal(fg, eg)


def test_tabled_begin_rejects_what_it_cannot_use():
    """plk_halo_begin_tabled_dev: a table-free context, tables of another curve, tables over fewer generators than the argument,
    H / U positions inside halo_g, beyond the tables or equal - all PLK_ERR_INVALID_ARG with a text, nothing half-built."""
    pytest.importorskip("torch")
    from plonky_amd import device as dev
    from plonky_amd.lib import PlonkyHipError
    dev.init(0)
    c = br.TWEEDLEDEE
    n = 64
    g, h, up, a, b, _ = _setup(c, n, 31)
    L = c.base.n_limbs
    more = ol.gen_bases(c.curve_id, 2, h, up).reshape(2, 2, L)
    dg = dev.to_device(g)
    args = (c.curve_id, dev.to_device(a), dev.to_device(b), dg, h, up)
    full = dev.msm_precompute_dev(c.curve_id, dev.to_device(np.concatenate([g, more])))
    x = np.array(c.scalar.mont_limbs(5), dtype=np.uint64)
    with pytest.raises(PlonkyHipError, match="tabled context"):
        dev.HaloArgument(*args, tables=dev.msm_precompute_dev(c.curve_id, dg, table_free=True))
    with pytest.raises(PlonkyHipError, match="tabled context"):
        dev.HaloArgument(*args, tables=dev.msm_precompute_dev(c.curve_id, dg[: n // 2].contiguous()))
    g2 = _setup(br.TWEEDLEDUM, n, 31)[0]
    with pytest.raises(PlonkyHipError, match="tabled context"):
        dev.HaloArgument(*args, tables=dev.msm_precompute_dev(br.TWEEDLEDUM.curve_id, dev.to_device(g2)))
    for hi, ui in ((n - 1, n), (n, n), (n, n + 2), (n + 5, n + 1)):
        with pytest.raises(PlonkyHipError, match="generators"):
            dev.HaloArgument(*args, tables=full, freeze_log=3, h_index=hi, u_index=ui, u_prime_scalar=x)
    # ... and the same call with proper positions goes through
    arg = dev.HaloArgument(*args, tables=full, freeze_log=3, h_index=n, u_index=n + 1, u_prime_scalar=x)
    assert len(arg) == n and arg.frozen
    arg.free()


def test_fold_multi_rejects_bad_depths():
    pytest.importorskip("torch")
    import torch
    from plonky_amd import device as dev, lib
    from plonky_amd.lib import PlonkyHipError
    dev.init(0)
    c = br.TWEEDLEDEE
    g = torch.zeros((32, 2, c.base.n_limbs), dtype=torch.int64, device="cuda")
    for r in (0, 5):
        sc = torch.zeros((1 << r, 4), dtype=torch.int64, device="cuda")
        with pytest.raises(PlonkyHipError, match="1 <= r <= 4"):
            dev.fold_generators_multi_dev(c.curve_id, g, sc, r)


def test_fold_multi_needs_the_endomorphism():
    pytest.importorskip("torch")
    import torch
    from plonky_amd import device as dev, lib
    dev.init(0)
    c = br.BLS12_377
    g = torch.zeros((4, 2, c.base.n_limbs), dtype=torch.int64, device="cuda")
    sc = torch.zeros((2, 4), dtype=torch.int64, device="cuda")
    with pytest.raises(Exception, match="endomorphism"):
        dev.fold_generators_multi_dev(c.curve_id, g, sc, 1)


@pytest.mark.parametrize("tabled,freeze_log", [(False, 0), (True, 12)])
def test_halo_argument_2p16_closed_form(tabled, freeze_log):
    """A whole 2^16 argument through both regimes (explicit folds down to 2^14, frozen below): the final generator is
    <s, G> with s_i = prod_j u_j^(+-1) by the bits of i, G_i = G0 + i D: [sum s_i] G0 + [sum i s_i] D on Python integers."""
    pytest.importorskip("torch")
    from plonky_amd import device as dev, synth
    from plonky_amd.selfcheck import _add, _mul
    dev.init(0)
    c = br.TWEEDLEDEE
    f = c.scalar
    log_n = 16
    n = 1 << log_n
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 0x9E3779B9 + 5, G)
    pt = lambda P: np.array([c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])], dtype=np.uint64)
    g = dev.gen_bases_dev(c.curve_id, n, pt(G), pt(D))
    a = synth.rand_field(f.field_id, 11, n)
    b = synth.rand_field(f.field_id, 12, n)
    us = synth.rand_field(f.field_id, 13, log_n)
    bl = synth.rand_field(f.field_id, 14, 2)
    tables = dev.msm_precompute_dev(c.curve_id, g) if tabled else None   # three lead rounds, explicit folds 2^13 -> 2^12, frozen below
    arg = dev.HaloArgument(c.curve_id, dev.to_device(a), dev.to_device(b), g, pt(br.ec_mul(c, 77, G)), pt(br.ec_mul(c, 78, G)), freeze_log=freeze_log,
                           tables=tables)
    u_ints = [f.from_mont(limbs_to_int(r)) for r in us]
    for j in range(log_n):
        lr, z = arg.round_lr(bl[0], bl[1])
        assert not z.any()
        arg.round_fold(us[j], np.array(f.mont_limbs(pow(u_ints[j], -1, f.p)), dtype=np.uint64))
    assert arg.frozen
    fa, fb, fg, fgz = arg.read()
    arg.free()
    # s_i by the bits of i (bit log_n - 1 - j selects hi in round j): vectorised over i with Python integers per bit pattern
    s = np.ones(1, dtype=object)
    for j in range(log_n):  # build from the top bit down: s = [s * uinv, s * u] concatenated by halves
        uj, uinv = u_ints[j], pow(u_ints[j], -1, f.p)
        s = np.concatenate([s * uinv % f.p, s * uj % f.p]) if j == 0 else np.concatenate([np.concatenate([blk * uinv % f.p, blk * uj % f.p]) for blk in np.split(s, 1 << j)])
    # after the loop s is ordered with round 0's bit as the most significant: index i = sum bit_j 2^(log_n-1-j)
    sum_s = int(sum(int(v) for v in s) % f.p)
    sum_is = int(sum(i * int(v) for i, v in enumerate(s)) % f.p)
    p = c.base.p
    exp = _add(p, _mul(p, sum_s, G), _mul(p, sum_is, D))
    assert int(fgz[0]) == 0 and tuple(from_mont_arr(c.base, fg[0])) == exp
    a_ints = None  # halo_a's closed form is covered at 2^10 above; here the generator path through both regimes is the point
