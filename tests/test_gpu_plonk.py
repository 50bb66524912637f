"""GPU parity tests of the Plonk quotient numerator: plk_plonk_evaluate_all_constraints and
plk_plonk_vanishing_points through the C ABI against the oracle's restatement of src/gates/ and plonk.rs:392-453, bit for
bit, plus the size-independent property of an honest witness: the vanishing polynomial divides by Z_H exactly.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import plonky_amd as pa
from plonky_amd import api
from oracle import bigint_ref as br
from oracle import oracle_lib as ol
from tests.test_oracle_plonk import ZETA_MONT, _random_tables, mont

FIELDS = [br.TWEEDLEDEE_BASE, br.TWEEDLEDUM_BASE, br.BLS12_377_SCALAR]


@pytest.mark.parametrize("f", FIELDS, ids=lambda f: f.name)
def test_evaluate_all_constraints_matches_oracle(f):
    """gates/mod.rs:46-125 at random points; half of them with binary selector constants (exactly one gate active)."""
    count = 300
    k = ol.rand_field(f.field_id, 11, count * 6).reshape(count, 6, 4)
    l, r, b = (ol.rand_field(f.field_id, 12 + i, count * 9).reshape(count, 9, 4) for i in range(3))
    zeta, a = ol.rand_field(f.field_id, 20, 2)
    one, zero = mont(f, [1])[0], mont(f, [0])[0]
    for i in range(count // 2):
        bits = br.PLONK_GATES[i % len(br.PLONK_GATES)][0]
        for j, c in enumerate(bits):
            k[i, j] = one if c == "1" else zero
    got = api.evaluate_all_constraints(f.field_id, k, l, r, b, zeta, a)
    for i in range(count):
        exp = ol.gate_constraints(f.field_id, -1, k[i], l[i], r[i], b[i], zeta, a)
        assert np.array_equal(got[i], exp), i


@pytest.mark.parametrize("f", FIELDS, ids=lambda f: f.name)
@pytest.mark.parametrize("degree", [1, 2, 8, 128, 2048])
def test_vanishing_points_match_oracle(f, degree):
    """plonk.rs:392-453 on random tables: every gate contributes to every point; degree 1, 2 exercise the wrap-around of
    the right / below indices, 2048 the two-level power table."""
    consts, wires, sigma, z, k_is, alpha, beta, gamma, zeta, a = _random_tables(f, degree, 0xBEEF + degree)
    got = api.vanishing_points(f.field_id, degree, consts, wires, sigma, z, k_is, alpha, beta, gamma, zeta, a)
    exp = ol.vanishing_points(f.field_id, degree, consts, wires, sigma, z, k_is, alpha, beta, gamma, zeta, a, threads=16)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("f", FIELDS, ids=lambda f: f.name)
def test_vanishing_points_extreme_words(f):
    """The same comparison with half of every table (and then the challenges) replaced by words at the edges of the input conversion
    and of the lazy limb arithmetic: largest limbs, zero operands, every row of the top-part table (tests/test_oracle_plonk.py pins
    the oracle to the big-integer restatement on such tables)."""
    from tests.test_oracle_plonk import extreme_tables
    degree = 64
    (consts, wires, sigma, z, k_is, alpha, beta, gamma, zeta, a), ext = extreme_tables(f, degree, 0xE47)
    rng = np.random.default_rng(7)
    for round_ in range(3):
        if round_ == 1:
            alpha, beta, gamma = ext[-1].copy(), ext[-2].copy(), ext[0].copy()  # p - 1, p - 2, 0
        if round_ == 2:
            k_is = ext[rng.integers(0, len(ext), 6)].copy()
            zeta, a = ext[-1].copy(), ext[-1].copy()
        got = api.vanishing_points(f.field_id, degree, consts, wires, sigma, z, k_is, alpha, beta, gamma, zeta, a)
        exp = ol.vanishing_points(f.field_id, degree, consts, wires, sigma, z, k_is, alpha, beta, gamma, zeta, a, threads=16)
        assert np.array_equal(got, exp), round_


def honest_tables(f, degree, seed):
    """A satisfied circuit of `degree` gates cycling through Arithmetic / Constant / Base4Sum / Buffer / CurveDbl gates with
    the identity wiring (s_sigma_j = k_j x, hence Z = 1): selector and wire columns on the n-subgroup, Montgomery limbs."""
    import random
    p = f.p
    rng = random.Random(seed)
    G = (br.TWEEDLEDUM.gx, br.TWEEDLEDUM.gy)
    consts = [[0] * degree for _ in range(6)]
    wires = [[0] * degree for _ in range(9)]
    for i in range(degree):
        kind = i % 5
        w = [rng.randrange(p) for _ in range(9)]
        if kind == 0:  # ArithmeticGate 1001 + (c0, c1)
            k = [1, 0, 0, 1, rng.randrange(p), rng.randrange(p)]
            w[3] = (k[4] * w[0] * w[1] + k[5] * w[2]) % p
        elif kind == 1:  # ConstantGate 10110 + c
            k = [1, 0, 1, 1, 0, rng.randrange(p)]
            w[0] = k[5]
        elif kind == 2:  # Base4SumGate 1000
            k = [1, 0, 0, 0, rng.randrange(p), rng.randrange(p)]
            limbs = [rng.randrange(4) for _ in range(7)]
            acc = w[0]
            for v in limbs:
                acc = (4 * acc + v) % p
            w = [w[0], acc] + limbs
        elif kind == 3:  # BufferGate 101010 would also switch CurveAddGate on (its prefix 10101, see test_oracle_plonk): use RescueB-free rows
            k = [1, 0, 1, 1, 0, rng.randrange(p)]
            w[0] = k[5]
        else:  # CurveDblGate 10111 (only meaningful over the inner curve's base field; any field satisfies the equations)
            k = [1, 0, 1, 1, 1, rng.randrange(p)]
            x, y = rng.randrange(1, p), rng.randrange(1, p)
            inv = pow(2 * y, -1, p)
            lam = 3 * x * x * inv % p
            xn = (lam * lam - 2 * x) % p
            yn = (lam * (x - xn) - y) % p
            w = [x, y, xn, yn, inv, lam] + w[6:]
        for j in range(6):
            consts[j][i] = k[j]
        for j in range(9):
            wires[j][i] = w[j]
    to = lambda rows: np.stack([mont(f, r) for r in rows])
    return to(consts), to(wires)


@pytest.mark.parametrize("log_degree", [7, 12])
def test_honest_witness_quotient_divides_exactly(log_degree):
    """End to end on the device: LDE of the selector / wire columns (plonk_util.rs:169-190), vanishing points, inverse
    transform (plonk.rs:455), divide_by_z_h (plonk.rs:178-181): for a satisfied circuit the division is exact, i.e.
    q * Z_H reproduces the vanishing polynomial coefficient for coefficient, and the quotient has degree < 7n."""
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    dev.init(0)
    f = br.TWEEDLEDUM_BASE
    degree = 1 << log_degree
    n8 = 8 * degree
    consts_n, wires_n = honest_tables(f, degree, 4242 + log_degree)
    k_is = ol.rand_field(f.field_id, 5, 6)
    alpha, beta, gamma = ol.rand_field(f.field_id, 6, 3)
    zeta, a = np.array(ZETA_MONT, dtype=np.uint64), mont(f, [0])[0]
    # columns -> coefficients (ifft over the n-subgroup) -> values over the 8n domain
    cols = dev.to_device(np.concatenate([consts_n, wires_n]))                      # (15, n, 4)
    coeffs = dev.ntt_dev(f.field_id, cols, inverse=True)
    lde = dev.ntt_padded_dev(f.field_id, coeffs, log_degree + 3)                  # (15, 8n, 4)
    consts_8n, wires_8n = lde[:6].contiguous(), lde[6:].contiguous()
    # identity wiring: S_sigma_j(x) = k_j x on the whole 8n domain; Z = 1
    g8 = f.primitive_root_of_unity(log_degree + 3)
    xs, x = [], 1
    for _ in range(n8):
        xs.append(x)
        x = x * g8 % f.p
    kc = [f.from_mont(br.limbs_to_int(k_is[j])) for j in range(6)]
    sigma_8n = dev.to_device(np.stack([mont(f, [kc[j] * v % f.p for v in xs]) for j in range(6)]))
    z_8n = dev.to_device(np.tile(mont(f, [1]), (n8, 1)))
    pts = dev.vanishing_points_dev(f.field_id, log_degree, consts_8n, wires_8n, sigma_8n, z_8n, k_is, alpha, beta, gamma, zeta, a)
    # spot-check against the oracle on the same tables
    if log_degree <= 7:
        exp = ol.vanishing_points(f.field_id, degree, dev.to_host(consts_8n), dev.to_host(wires_8n), dev.to_host(sigma_8n), dev.to_host(z_8n), k_is,
                                  alpha, beta, gamma, zeta, a, threads=16)
        assert np.array_equal(dev.to_host(pts), exp)
    vanishing = dev.ntt_dev(f.field_id, pts, inverse=True)                          # Polynomial::from_evaluations
    v_host = dev.to_host(vanishing)
    assert v_host.any(), "the vanishing polynomial of random gates is not identically zero"
    q = pa.polynomial_divide_by_z_h(f.field_id, v_host, degree)
    q_ints = [f.from_mont(br.limbs_to_int(r)) for r in q]
    while q_ints and q_ints[-1] == 0:
        q_ints.pop()
    assert len(q_ints) <= 7 * degree                                                 # plonk.rs:182 pads t to 7n
    # q * (X^n - 1) == vanishing, exactly
    v_ints = [f.from_mont(br.limbs_to_int(r)) for r in v_host]
    back = [0] * n8
    for i, c in enumerate(q_ints):
        back[i + degree] = (back[i + degree] + c) % f.p
        back[i] = (back[i] - c) % f.p
    assert back == v_ints
    # and a broken witness is caught: flip one wire value -> the division leaves a remainder
    wires_bad = wires_n.copy()
    wires_bad[3, 0] = mont(f, [12345])[0]
    cols = dev.to_device(np.concatenate([consts_n, wires_bad]))
    lde = dev.ntt_padded_dev(f.field_id, dev.ntt_dev(f.field_id, cols, inverse=True), log_degree + 3)
    pts = dev.vanishing_points_dev(f.field_id, log_degree, lde[:6].contiguous(), lde[6:].contiguous(), sigma_8n, z_8n, k_is, alpha, beta, gamma, zeta, a)
    v_bad = dev.to_host(dev.ntt_dev(f.field_id, pts, inverse=True))
    q_bad = [f.from_mont(br.limbs_to_int(r)) for r in pa.polynomial_divide_by_z_h(f.field_id, v_bad, degree)]
    back = [0] * (len(q_bad) + degree)
    for i, c in enumerate(q_bad):
        back[i + degree] = (back[i + degree] + c) % f.p
        back[i] = (back[i] - c) % f.p
    assert back[:n8] != [f.from_mont(br.limbs_to_int(r)) for r in v_bad]
