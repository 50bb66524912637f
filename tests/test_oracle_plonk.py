"""Pin the oracle's restatement of the Plonk quotient numerator (the gates of src/gates/,
evaluate_all_constraints, eval_l_1, Prover::vanishing_poly's 8n-point loop).  CPU only.

The reference holds no known-answer vectors for the gates; what it holds is (a) the gate equations themselves,
(b) the low-degree test of every gate (test_gate_low_degree!, gates/mod.rs:336-443) and (c) the fact that an honest
witness satisfies them.  So the C++ restatement (oracle/plonk_gates.inc, statement by statement) is pinned against
an independent big-int restatement written from the equations (oracle/bigint_ref.py), against honest witnesses
built from real curve / Rescue arithmetic (every constraint must vanish), and against the degree bound.
"""
import random

import numpy as np
import pytest

from oracle import bigint_ref as br
from oracle import oracle_lib as ol
from tests.util import array_to_ints, ints_to_array

F = br.TWEEDLEDUM_BASE          # C = Tweedledee: C::ScalarField = TweedledumBase = InnerC::BaseField, InnerC = Tweedledum
INNER = br.TWEEDLEDUM
# InnerC::ZETA, tweedledum_curve.rs:37-44 (Montgomery limbs); InnerC::A = 0 (tweedledum_curve.rs:11)
ZETA_MONT = [7605997034305223424, 3132214451552427455, 3308921103222877309, 2709928666517121162]
ZETA = F.from_mont(br.limbs_to_int(ZETA_MONT))
FIELDS = [br.TWEEDLEDEE_BASE, br.TWEEDLEDUM_BASE, br.BLS12_377_SCALAR]
N_GATES = len(br.PLONK_GATES)


def mont(f, vals):
    return ints_to_array([f.to_mont(v % f.p) for v in vals], 4)


def unmont(f, arr):
    return [f.from_mont(v) for v in array_to_ints(arr)]


def oracle_gate(f, gate, k, l, r, b, zeta, a, unfiltered=False):
    return unmont(f, ol.gate_constraints(f.field_id, gate, mont(f, k), mont(f, l), mont(f, r), mont(f, b), mont(f, [zeta])[0], mont(f, [a])[0], unfiltered))


def test_zeta_is_a_cube_root_of_unity():
    assert ZETA != 1 and pow(ZETA, 3, F.p) == 1


@pytest.mark.parametrize("f", FIELDS, ids=lambda f: f.name)
def test_gates_match_bigint_restatement(f):
    rng = random.Random(1234 + f.field_id)
    for trial in range(6):
        k = [rng.randrange(f.p) for _ in range(6)]
        if trial >= 3:  # binary selectors, as in a real circuit: exactly one filter is 1
            bits = br.PLONK_GATES[rng.randrange(N_GATES)][0]
            k[: len(bits)] = [int(c) for c in bits]
        l, r, b = ([rng.randrange(f.p) for _ in range(9)] for _ in range(3))
        zeta, a = rng.randrange(f.p), rng.randrange(f.p)
        for g in range(N_GATES):
            for unf in (False, True):
                assert oracle_gate(f, g, k, l, r, b, zeta, a, unf) == br.plonk_gate_filtered(f, g, k, l, r, b, zeta, a, unf), (g, unf)
        assert oracle_gate(f, -1, k, l, r, b, zeta, a) == br.plonk_all_constraints(f, k, l, r, b, zeta, a)


def test_constraint_counts():
    """curve_add 6, curve_dbl 4, curve_endo 7, base_4_sum 1 + 7, public_input 3, buffer 0, constant 1, arithmetic 1, rescue_a 8, rescue_b 4."""
    z = [0] * 9
    assert [len(oracle_gate(F, g, z[:6], z, z, z, ZETA, 0)) for g in range(N_GATES)] == [6, 4, 7, 8, 3, 0, 1, 1, 8, 4]
    assert len(oracle_gate(F, -1, z[:6], z, z, z, ZETA, 0)) == 8


def test_prefixes_as_in_the_reference_code():
    """The doc comment of gates/mod.rs:1-16 lists CurveAddGate as 101000 and BufferGate as 101010; the CODE gives CurveAddGate
    10101 (curve_add.rs:39) and BufferGate 101000 (buffer.rs:27).  The restatements follow the code, whose prefixes are prefix-free
    (the doc comment's pair would not be: 10101 is a prefix of 101010)."""
    names = ["curve_add", "curve_dbl", "curve_endo", "base_4_sum", "public_input", "buffer", "constant", "arithmetic", "rescue_a", "rescue_b"]
    by_name = dict(zip(names, (gb for gb, _ in br.PLONK_GATES)))
    assert by_name["curve_add"] == "10101" and by_name["buffer"] == "101000" and by_name["public_input"] == "101001"
    clashes = set()
    for g, (gb, _) in enumerate(br.PLONK_GATES):
        for h, (hb, _) in enumerate(br.PLONK_GATES):
            if g != h and hb.startswith(gb):
                clashes.add((names[g], names[h]))
    assert clashes == set()
    assert "101010".startswith(by_name["curve_add"])  # what the doc comment's BufferGate prefix would have collided with


def _prefix_consts(gate, extra):
    bits = br.PLONK_GATES[gate][0]
    return ([int(c) for c in bits] + list(extra) + [0] * 6)[:6]


def test_honest_witnesses_satisfy_the_gates():
    """Every constraint vanishes on a witness built from the real arithmetic (and one flipped value breaks it)."""
    p = F.p
    rng = random.Random(77)
    G = (INNER.gx, INNER.gy)
    P1, P2 = br.ec_mul(INNER, 1234567, G), br.ec_mul(INNER, 7654321, G)
    z9 = [0] * 9
    # CurveAddGate (curve_add.rs:108-149 generates exactly this): bit = 1 -> P1 + P2, bit = 0 -> P1
    for bit in (0, 1):
        P3 = br.ec_add(INNER, P1, P2)
        inv = pow(P1[0] - P2[0], -1, p)
        lam = (P1[1] - P2[1]) * inv % p
        out = P3 if bit else P1
        acc_old = rng.randrange(p)
        l = [P1[0], P1[1], acc_old, (2 * acc_old + bit) % p, P2[0], P2[1], bit, inv, lam]
        r = [out[0], out[1]] + z9[2:]
        assert oracle_gate(F, 0, _prefix_consts(0, []), l, r, z9, ZETA, 0) == [0] * 6
        l[8] = (lam + 1) % p
        assert any(oracle_gate(F, 0, _prefix_consts(0, []), l, r, z9, ZETA, 0))
    # CurveDblGate
    D = br.ec_add(INNER, P1, P1)
    inv = pow(2 * P1[1], -1, p)
    lam = 3 * P1[0] * P1[0] * inv % p
    l = [P1[0], P1[1], D[0], D[1], inv, lam, 0, 0, 0]
    assert oracle_gate(F, 1, _prefix_consts(1, []), l, z9, z9, ZETA, 0) == [0] * 4
    # CurveEndoGate: the addend is (zeta^b1 x, (2 b0 - 1) y) (curve_endo.rs:98-141)
    for b0 in (0, 1):
        for b1 in (0, 1):
            Q = ((ZETA if b1 else 1) * P2[0] % p, (2 * b0 - 1) * P2[1] % p)
            assert br.ec_on_curve(INNER, Q)
            S = br.ec_add(INNER, P1, Q)
            inv = pow(P1[0] - Q[0], -1, p)
            un_old, sg_old = rng.randrange(p), rng.randrange(p)
            l = [P1[0], P1[1], un_old, sg_old, P2[0], P2[1], b0, b1, inv]
            r = [S[0], S[1]] + z9[2:]
            limb = (2 * b0 - 1) * ((ZETA - 1) * b1 + 1)
            b = [0, 0, (4 * un_old + 2 * b1 + b0) % p, (2 * sg_old + limb) % p] + z9[4:]
            assert oracle_gate(F, 2, _prefix_consts(2, []), l, r, b, ZETA, 0) == [0] * 7
    # Base4SumGate
    limbs = [rng.randrange(4) for _ in range(7)]
    acc = acc_old = rng.randrange(p)
    for v in limbs:
        acc = (4 * acc + v) % p
    assert oracle_gate(F, 3, _prefix_consts(3, []), [acc_old, acc] + limbs, z9, z9, ZETA, 0) == [0] * 8
    assert oracle_gate(F, 3, _prefix_consts(3, []), [acc_old, acc] + [4] + limbs[1:], z9, z9, ZETA, 0)[1] != 0
    # ArithmeticGate / ConstantGate / PublicInputGate
    c0, c1, m0, m1, ad = (rng.randrange(p) for _ in range(5))
    assert oracle_gate(F, 7, _prefix_consts(7, [c0, c1]), [m0, m1, ad, (c0 * m0 * m1 + c1 * ad) % p] + z9[4:], z9, z9, ZETA, 0) == [0]
    assert oracle_gate(F, 6, _prefix_consts(6, [c0]), [c0] + z9[1:], z9, z9, ZETA, 0) == [0]
    adv = [rng.randrange(p) for _ in range(3)]
    assert oracle_gate(F, 4, _prefix_consts(4, []), z9[:6] + adv, adv + z9[3:], z9, ZETA, 0) == [0] * 3
    # Rescue steps: roots = ins^(1/5) (step A), outs = MDS * (...) + round constants
    mds = [[pow(4 + i - j, -1, p) for j in range(4)] for i in range(4)]
    assert mds[1][2] == unmont(F, ol.mds(F.field_id, 4, 1, 2).reshape(1, 4))[0]
    roots = [rng.randrange(p) for _ in range(4)]
    ins = [pow(x, 5, p) for x in roots]
    rc = [rng.randrange(p) for _ in range(4)]
    outs = [(rc[i] + sum(mds[i][j] * roots[j] for j in range(4))) % p for i in range(4)]
    assert oracle_gate(F, 8, [0, 0] + rc, ins + roots + [0], outs + z9[4:], z9, ZETA, 0) == [0] * 8
    outs_b = [(rc[i] + sum(mds[i][j] * pow(ins[j], 5, p) for j in range(4))) % p for i in range(4)]
    assert oracle_gate(F, 9, [0, 1] + rc, ins + z9[4:], outs_b + z9[4:], z9, ZETA, 0) == [0] * 4


@pytest.mark.parametrize("f", FIELDS, ids=lambda f: f.name)
def test_eval_l_1(f):
    """plonk_util.rs:14-24: L_1 is the Lagrange basis of the order-n subgroup at 1."""
    n = 16
    g = f.primitive_root_of_unity(4)
    for kk in range(n):
        got = unmont(f, ol.eval_l_1(f.field_id, n, mont(f, [pow(g, kk, f.p)])[0]).reshape(1, 4))[0]
        assert got == (1 if kk == 0 else 0)
    x = 0xABCDEF123456789
    assert unmont(f, ol.eval_l_1(f.field_id, n, mont(f, [x])[0]).reshape(1, 4))[0] == br.plonk_eval_l_1(f, n, x)


def _random_tables(f, degree, seed):
    n8 = 8 * degree
    mk = lambda rows, s: ol.rand_field(f.field_id, seed + s, rows * n8).reshape(rows, n8, 4)
    scal = ol.rand_field(f.field_id, seed + 99, 6 + 5)
    return mk(6, 1), mk(9, 2), mk(6, 3), mk(1, 4)[0], scal[:6], scal[6], scal[7], scal[8], scal[9], scal[10]


@pytest.mark.parametrize("f", FIELDS, ids=lambda f: f.name)
@pytest.mark.parametrize("degree", [1, 8, 128])
def test_vanishing_points_match_bigint_restatement(f, degree):
    """plonk.rs:392-453 on random tables (random selector constants: every gate contributes to every point)."""
    consts, wires, sigma, z, k_is, alpha, beta, gamma, zeta, a = _random_tables(f, degree, 0x5EED + degree)
    got = unmont(f, ol.vanishing_points(f.field_id, degree, consts, wires, sigma, z, k_is, alpha, beta, gamma, zeta, a, threads=4))
    rows = lambda t: [unmont(f, t[j]) for j in range(t.shape[0])]
    one = lambda v: unmont(f, v.reshape(1, 4))[0]
    exp = br.plonk_vanishing_points(f, degree, rows(consts), rows(wires), rows(sigma), unmont(f, z), unmont(f, k_is), one(alpha), one(beta),
                                    one(gamma), one(zeta), one(a))
    assert got == exp


def extreme_words(f):
    """Stored (Montgomery) words at the edges of a 256-bit word and of the device's input conversion (it cuts a word at bit
    s = floor(log2 p) - 5, plonk.hip lz_from_rform): 0, 1, p - 1, all-ones low parts, every cut boundary +- 1."""
    p = f.p
    s = p.bit_length() - 1 - 5
    vals = {0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << s) - 1, 1 << s, (1 << s) + 1, ((p - 1) >> s) << s, (((p - 1) >> s) << s) - 1, (1 << 29) - 1, 1 << 29,
            (1 << 232) - 1, 1 << 232, (1 << 227) - 1, 1 << 227}
    for t in range(1, ((p - 1) >> s) + 1):
        vals.update({t << s, (t << s) - 1, (t << s) | ((1 << 29) - 1)})
    return sorted(v for v in vals if 0 <= v < p)


def extreme_tables(f, degree, seed):
    """_random_tables with half of every table replaced by extreme_words; returns the tables and the word list (as arrays)."""
    consts, wires, sigma, z, k_is, alpha, beta, gamma, zeta, a = _random_tables(f, degree, seed)
    ext = ints_to_array(extreme_words(f), 4)
    rng = np.random.default_rng(seed + f.field_id)
    for t in (consts, wires, sigma, z):
        flat = t.reshape(-1, 4)
        idx = rng.random(flat.shape[0]) < 0.5
        flat[idx] = ext[rng.integers(0, len(ext), int(idx.sum()))]
    return (consts, wires, sigma, z, k_is, alpha, beta, gamma, zeta, a), ext


@pytest.mark.parametrize("f", FIELDS, ids=lambda f: f.name)
def test_vanishing_points_extreme_words_match_bigint_restatement(f):
    """The oracle itself at the edge words (the GPU test of the same name compares the device with the oracle there)."""
    degree = 4
    (consts, wires, sigma, z, k_is, alpha, beta, gamma, zeta, a), ext = extreme_tables(f, degree, 0xE47)
    alpha, beta, gamma = ext[-1].copy(), ext[-2].copy(), ext[0].copy()  # p - 1, p - 2, 0
    got = unmont(f, ol.vanishing_points(f.field_id, degree, consts, wires, sigma, z, k_is, alpha, beta, gamma, zeta, a, threads=4))
    rows = lambda t: [unmont(f, t[j]) for j in range(t.shape[0])]
    one = lambda v: unmont(f, v.reshape(1, 4))[0]
    exp = br.plonk_vanishing_points(f, degree, rows(consts), rows(wires), rows(sigma), unmont(f, z), unmont(f, k_is), one(alpha), one(beta),
                                    one(gamma), one(zeta), one(a))
    assert got == exp


@pytest.mark.parametrize("gate", range(N_GATES), ids=[g[0] for g in br.PLONK_GATES])
def test_gate_low_degree(gate):
    """test_gate_low_degree! (gates/mod.rs:336-443) at n = 16 instead of 256: random degree < n constant and wire
    polynomials, extended to 16n points; every filtered constraint interpolates to degree + 1 <= 8n."""
    f, n = F, 16
    pre_n, pre_16n = ol.FftPrecomputation(f.field_id, n), ol.FftPrecomputation(f.field_id, 16 * n)

    def lde(rows, seed):
        vals = ol.rand_field(f.field_id, seed, rows * n).reshape(rows, n, 4)
        out = []
        for j in range(rows):
            coeffs = pre_n.ifft_with_precomputation_power_of_2(vals[j])
            padded = np.zeros((16 * n, 4), dtype=np.uint64)
            padded[:n] = coeffs
            out.append(pre_16n.fft_with_precomputation_power_of_2(padded))
        return np.stack(out)

    consts, wires = lde(6, 1000 + gate), lde(9, 2000 + gate)
    zeta, a = mont(f, [ZETA])[0], mont(f, [0])[0]
    cols = None
    for i in range(16 * n):
        ir, ib = (i + 16) % (16 * n), (i + 16 * br.GRID_WIDTH) % (16 * n)
        c = ol.gate_constraints(f.field_id, gate, consts[:, i], wires[:, i], wires[:, ir], wires[:, ib], zeta, a)
        if cols is None:
            cols = [[] for _ in range(len(c))]
        for j in range(len(c)):
            cols[j].append(c[j])
    for j, col in enumerate(cols or []):
        coeffs = pre_16n.ifft_with_precomputation_power_of_2(np.stack(col))
        nz = np.nonzero(coeffs.any(axis=1))[0]
        deg_plus_1 = int(nz[-1]) + 1 if len(nz) else 0
        assert deg_plus_1 <= 8 * n, (gate, j, deg_plus_1)
