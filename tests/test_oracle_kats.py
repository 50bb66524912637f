"""Pin the oracle (oracle/plk_oracle.cpp + oracle/bigint_ref.py) against every constant, KAT and
unit-test vector the reference holds for the NTT/MSM path.  CPU only.

Each test names the reference test / constant it reproduces (file:line into /root/reference).
"""
import numpy as np
import pytest

from oracle import bigint_ref as br
from oracle import oracle_lib as ol
from tests.util import array_to_ints, int_to_limbs, ints_to_array, limbs_to_int

FIELDS = [br.TWEEDLEDEE_BASE, br.TWEEDLEDUM_BASE, br.BLS12_377_SCALAR, br.BLS12_377_BASE, br.PALLAS_BASE, br.VESTA_BASE]
CURVES = [br.TWEEDLEDEE, br.TWEEDLEDUM, br.BLS12_377, br.PALLAS, br.VESTA]


def mont_arr(f, vals):
    return ints_to_array([f.to_mont(v % f.p) for v in vals], f.n_limbs)


def from_mont_arr(f, arr):
    return [f.from_mont(v) for v in array_to_ints(arr)]


# ---- 1. constants as KATs (tweedledee_base.rs:22-171, tweedledum_base.rs, bls12_377_{base,scalar}.rs) ----
@pytest.mark.parametrize("f", FIELDS, ids=lambda f: f.name)
def test_field_constants_rederived(f):
    n = f.n_limbs
    assert limbs_to_int(ol.field_const(f.field_id, "ORDER")) == f.p
    assert limbs_to_int(ol.field_const(f.field_id, "R")) == f.R
    assert limbs_to_int(ol.field_const(f.field_id, "R2")) == f.R2
    assert limbs_to_int(ol.field_const(f.field_id, "R3")) == f.R3
    assert int(ol.field_const(f.field_id, "MU")[0]) == f.mu
    assert limbs_to_int(ol.field_const(f.field_id, "TWO")) == f.to_mont(2)
    assert limbs_to_int(ol.field_const(f.field_id, "THREE")) == f.to_mont(3)
    assert limbs_to_int(ol.field_const(f.field_id, "GENERATOR")) == f.to_mont(f.generator)
    assert limbs_to_int(ol.field_const(f.field_id, "T")) == f.to_mont(f.T)
    assert limbs_to_int(ol.field_const(f.field_id, "NEG_ONE")) == f.to_mont(f.p - 1)
    # from_canonical_u64(2).limbs == TWO.limbs
    two = ol.field_unop(f.field_id, "from_canonical", ints_to_array([2], n))
    assert list(two[0]) == list(ol.field_const(f.field_id, "TWO"))
    assert f.p & 0xFFFFFFFF == 1  # every in-scope modulus is 1 mod 2^32 (used by the HIP reduction)


def test_neg_one_literals():
    # tweedledee_base.rs:150-152, tweedledum_base.rs:150-152
    assert list(ol.field_const(0, "NEG_ONE")) == [1203234400779632644, 1020774078174862118, 0, 0]
    assert list(ol.field_const(1, "NEG_ONE")) == [9584103694345961476, 1020774078366967526, 0, 0]
    # bls12_377_scalar.rs:163
    assert list(ol.field_const(2, "NEG_ONE")) == [10157024534604021774, 16668528035959406606, 5322190058819395602, 387181115924875961]
    # bls12_377_base.rs:193-196
    assert list(ol.field_const(3, "NEG_ONE")) == [9384023879812382873, 14252412606051516495, 9184438906438551565,
                                                  11444845376683159689, 8738795276227363922, 81297770384137296]


def test_pallas_vesta_literals():
    """The reference's literals for the Pasta fields and curves (pallas_base.rs, vesta_base.rs, pallas_curve.rs, vesta_curve.rs)."""
    # pallas_base.rs:21-60: ORDER, R, R2, R3, MU; :153-155 NEG_ONE
    assert list(ol.field_const(4, "ORDER")) == [0x992d30ed00000001, 0x224698fc094cf91b, 0x0, 0x4000000000000000]
    assert list(ol.field_const(4, "R")) == [0x34786d38fffffffd, 0x992c350be41914ad, 0xffffffffffffffff, 0x3fffffffffffffff]
    assert list(ol.field_const(4, "R2")) == [0x8c78ecb30000000f, 0xd7d30dbd8b0de0e7, 0x7797a99bc3c95d18, 0x96d41af7b9cb714]
    assert list(ol.field_const(4, "R3")) == [0xf185a5993a9e10f9, 0xf6a68f3b6ac5b1d1, 0xdf8d1014353fd42c, 0x2ae309222d2d9910]
    assert int(ol.field_const(4, "MU")[0]) == 0x992d30ecffffffff
    assert list(ol.field_const(4, "NEG_ONE")) == [0x64b4c3b400000004, 0x891a63f02533e46e, 0, 0]
    # vesta_base.rs:22-27 ORDER
    assert list(ol.field_const(5, "ORDER")) == [0x8c46eb2100000001, 0x224698fc0994a8dd, 0x0, 0x4000000000000000]
    # the two curves form a cycle: each one's scalar field is the other's base field, and the group orders match
    for c in (br.PALLAS, br.VESTA):
        G = (c.gx, c.gy)
        assert br.ec_on_curve(c, G) and br.ec_mul(c, c.scalar.p, G) is None
    # pallas_curve.rs:24-35 / vesta_curve.rs:22-33: ZETA is a primitive cube root of unity of the base field, ZETA_SCALAR of the
    # scalar field, and the endomorphism (x, y) -> (ZETA x, y) is multiplication by ZETA_SCALAR
    zp = br.PALLAS_BASE.from_mont(limbs_to_int([0xfbdfd7aa9e65eac8, 0x0cd4d654e50025fb, 0xd59892a33785b99a, 0x2a27fb62585e8789]))
    zv = br.VESTA_BASE.from_mont(limbs_to_int([0x410e7d207feeeee3, 0x6afdf14fd8fa2279, 0xfd3d8a04eca4d4d7, 0x2de2d60777dba4ef]))
    assert zp != 1 and pow(zp, 3, br.PALLAS_BASE.p) == 1 and zv != 1 and pow(zv, 3, br.VESTA_BASE.p) == 1
    G = (br.PALLAS.gx, br.PALLAS.gy)
    assert br.ec_mul(br.PALLAS, zv, G) == (zp * G[0] % br.PALLAS_BASE.p, G[1])
    G = (br.VESTA.gx, br.VESTA.gy)
    assert br.ec_mul(br.VESTA, zp, G) == (zv * G[0] % br.VESTA_BASE.p, G[1])
    _check_glv_params(br.PALLAS, zp, zv)
    _check_glv_params(br.VESTA, zv, zp)


# ---- 2. test_to_digits (curve_msm.rs:186-216) ----
def test_to_digits_reference_vector():
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
    assert list(ol.field_unop(2, "to_canonical", x)[0]) == x_canonical
    assert ol.to_digits(2, x[0], 17) == expected
    assert br.to_digits(br.BLS12_377, limbs_to_int(x_canonical), 17) == expected


# ---- 3. test_msm (curve_msm.rs:218-241): BLS12-377, bases G,2G,3G, w=5, serial msm_execute ----
MSM_KAT_X = [12143235104262876081, 18188420512310322855, 3827143886156141214, 7291472954475080594, 15086800798470119505, 88226617263656672]
MSM_KAT_Y = [11939431015848900540, 9079896797298381533, 9490893446675808244, 8383328383104571926, 13148821763957202392, 64683294495841426]
MSM_KAT_X_MONT = [1660187596251774888, 7706514383371098576, 2044973863172090, 5805746842974987156, 11290425522310945819, 82008346954468233]
MSM_KAT_Y_MONT = [12907969589755861313, 1230281974771094326, 3495780339883022908, 3849941327073121838, 1722009066098801759, 73986680728869869]


def _bls_test_msm_inputs():
    c = br.BLS12_377
    G = (c.gx, c.gy)
    pts = [G, br.ec_mul(c, 2, G), br.ec_mul(c, 3, G)]
    scal = [limbs_to_int([11111111, 22222222, 33333333, 44444444]),
            limbs_to_int([22222222, 22222222, 33333333, 44444444]),
            limbs_to_int([33333333, 22222222, 33333333, 44444444])]
    return c, pts, scal


def test_msm_reference_unit_test():
    c, pts, scal = _bls_test_msm_inputs()
    # expected value from independent big-int arithmetic == the literal in SURVEY.md 8(c).3
    exp = br.msm(c, scal, pts)
    assert int_to_limbs(exp[0], 6) == MSM_KAT_X and int_to_limbs(exp[1], 6) == MSM_KAT_Y
    assert br.msm_yao(c, scal, pts, 5) == exp
    bases = np.array([[c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])] for P in pts], dtype=np.uint64)
    scalars = mont_arr(c.scalar, scal)
    pre = ol.MsmPrecomputation(2, bases, 5)
    for parallel in (False, True):
        out, zero = pre.execute(scalars, parallel=parallel)
        assert zero == 0
        assert list(out[0]) == MSM_KAT_X_MONT and list(out[1]) == MSM_KAT_Y_MONT
    # result_naive of the reference test: sum of CurveScalar * generator
    acc = None
    for s, P in zip(scal, pts):
        xy, z = ol.scalar_mul(2, mont_arr(c.scalar, [s])[0], np.array([c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])], dtype=np.uint64))
        pt = tuple(from_mont_arr(c.base, xy))
        acc = br.ec_add(c, acc, pt)
    assert acc == exp


def test_msm_length_mismatch_is_an_error():
    # assert_eq!(precomputation.powers_per_generator.len(), scalars.len())  curve_msm.rs:67,106
    c, pts, scal = _bls_test_msm_inputs()
    bases = np.array([[c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])] for P in pts], dtype=np.uint64)
    pre = ol.MsmPrecomputation(2, bases, 5)
    with pytest.raises(AssertionError):
        pre.execute(mont_arr(c.scalar, scal[:2]))


# ---- 4. Tweedledee mini-KAT ----
def test_tweedledee_mini_kat():
    c = br.TWEEDLEDEE
    G = (c.gx, c.gy)
    pts = [G, br.ec_mul(c, 2, G), br.ec_mul(c, 3, G)]
    exp = br.ec_mul(c, 14, G)
    assert int_to_limbs(exp[0], 4) == [14704193998986281273, 16806378273837548806, 1462211411981937090, 3084878336663070809]
    assert int_to_limbs(exp[1], 4) == [1150583325350264410, 10604125093259843816, 12810708021793073644, 4268371192981446443]
    bases = np.array([[c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])] for P in pts], dtype=np.uint64)
    for w in (4, 8, 11):
        pre = ol.MsmPrecomputation(0, bases, w)
        out, zero = pre.execute(mont_arr(c.scalar, [1, 2, 3]))
        assert zero == 0 and tuple(from_mont_arr(c.base, out)) == exp


# ---- 5. roots of unity (field.rs:429-435) ----
def test_roots_of_unity():
    r = ol.root_of_unity(0, 20)
    assert list(r) == [11373224964941166251, 12592263464888098129, 15702808136517317952, 2975761563284842327]
    assert int_to_limbs(br.TWEEDLEDEE_BASE.from_mont(limbs_to_int(r)), 4) == \
        [11965472966415935403, 5272490094479267563, 14706214652162365581, 2527472860114563183]
    r = ol.root_of_unity(1, 20)
    assert int_to_limbs(br.TWEEDLEDUM_BASE.from_mont(limbs_to_int(r)), 4) == \
        [12875609724145928807, 14157279445705010331, 5382916167020292923, 2653439448752205256]
    for f in FIELDS:
        for k in (0, 1, 2, 9, 14, 20, f.two_adicity):
            assert f.from_mont(limbs_to_int(ol.root_of_unity(f.field_id, k))) == f.primitive_root_of_unity(k)


@pytest.mark.parametrize("f", FIELDS[:2], ids=lambda f: f.name)
def test_primitive_root_order(f):
    # tweedledee_base.rs:246-253: generator_order(root) == 2^n_power for n_power < 10
    for n_power in range(10):
        root = f.from_mont(limbs_to_int(ol.root_of_unity(f.field_id, n_power)))
        order, cur = 1, root
        while cur != 1:
            cur = cur * root % f.p
            order += 1
        assert order == 1 << n_power


# ---- 6. fft_and_ifft (fft.rs:164-185) + test_reverse_bits (:187-195) ----
def test_fft_and_ifft_reference_unit_test():
    f = br.BLS12_377_SCALAR
    degree = 200
    coeffs = [(i * 1337) % 100 for i in range(degree)]
    pre = ol.FftPrecomputation(2, degree)
    assert pre.size() == 256
    points = pre.fft_with_precomputation(mont_arr(f, coeffs))
    assert points.shape[0] == 256
    expected = br.ntt_naive(f, coeffs + [0] * 56)
    assert from_mont_arr(f, points) == expected
    assert br.ntt(f, coeffs + [0] * 56) == expected
    back = pre.ifft_with_precomputation_power_of_2(points)
    got = from_mont_arr(f, back)
    assert got[:degree] == coeffs and got[degree:] == [0] * 56
    assert br.intt(f, expected) == coeffs + [0] * 56
    # multithreaded layer loop gives the same limbs
    assert np.array_equal(pre.fft_with_precomputation_power_of_2(points, threads=3),
                          pre.fft_with_precomputation_power_of_2(points, threads=1))


def test_fft_and_ifft_shape_on_bls12_377_base():
    """The same unit-test shape over the sixth field (fft::<Bls12377Base> type-checks in the reference, bls12_377_base.rs:18-262; no caller
    uses it): the oracle's generic restatement against the naive DFT and the big-integer radix-2 transform, two sizes, table layers."""
    f = br.BLS12_377_BASE
    for degree in (200, 1000):
        n = 1 << (degree - 1).bit_length()
        coeffs = [(i * 1337 + (i * i) % 89) % f.p for i in range(degree)]
        pre = ol.FftPrecomputation(3, degree)
        assert pre.size() == n
        points = pre.fft_with_precomputation(mont_arr(f, coeffs))
        assert points.shape == (n, 6)
        expected = br.ntt(f, coeffs + [0] * (n - degree))
        if degree == 200:
            assert br.ntt_naive(f, coeffs + [0] * (n - degree)) == expected
        assert from_mont_arr(f, points) == expected
        back = from_mont_arr(f, pre.ifft_with_precomputation_power_of_2(points))
        assert back[:degree] == coeffs and not any(back[degree:])
    pre = ol.FftPrecomputation(3, 16)
    for i in range(5):
        g = f.primitive_root_of_unity(i)
        assert from_mont_arr(f, pre.layer(i)) == [pow(g, ol.reverse_bits(k, i), f.p) for k in range(1 << i)]


def test_reverse_bits():
    assert ol.reverse_bits(0b00110101, 8) == 0b10101100
    # reverse_index_bits([a,b,c,d]) == [a,c,b,d]
    assert [ol.reverse_bits(i, 2) for i in range(4)] == [0, 2, 1, 3]
    assert [ol.reverse_bits(i, 1) for i in range(2)] == [0, 1]


def test_fft_table_contents():
    # FftPrecomputation.subgroups_rev[i] = bit-reversed powers of primitive_root_of_unity(i) (fft.rs:47-59)
    f = br.TWEEDLEDEE_BASE
    pre = ol.FftPrecomputation(0, 16)
    for i in range(5):
        g = f.primitive_root_of_unity(i)
        layer = from_mont_arr(f, pre.layer(i))
        assert layer == [pow(g, ol.reverse_bits(k, i), f.p) for k in range(1 << i)]


# ---- 7. div2 KAT (bigint_arithmetic.rs:134-156), BLS Montgomery KATs (bls12_377_base.rs:290-361) ----
def test_div2_kat():
    assert list(ol.div2([40, 0, 0, 0, 0, 0])) == [20, 0, 0, 0, 0, 0]
    assert list(ol.div2([15668009436471190370, 3102040391300197453, 4166322749169705801, 3518225024268476800,
                         11231577158546850254, 226224965816356276])) == \
        [17057376755090370993, 10774392232504874534, 2083161374584852900, 1759112512134238400,
         5615788579273425127, 113112482908178138]


def test_bls12base_to_and_from_canonical_and_mul():
    f = br.BLS12_377_BASE
    a = [1, 2, 3, 4, 0, 0]
    b = [3, 4, 5, 6, 0, 0]
    am = ol.field_unop(3, "from_canonical", np.array([a], dtype=np.uint64))
    bm = ol.field_unop(3, "from_canonical", np.array([b], dtype=np.uint64))
    assert limbs_to_int(am[0]) == limbs_to_int(a) * f.R % f.p
    assert list(ol.field_unop(3, "to_canonical", am)[0]) == a
    prod = ol.field_unop(3, "to_canonical", ol.field_binop(3, "mul", am, bm))
    assert limbs_to_int(prod[0]) == limbs_to_int(a) * limbs_to_int(b) % f.p


@pytest.mark.parametrize("f", FIELDS, ids=lambda f: f.name)
def test_negation_inverse_batch_inverse_small(f):
    n = f.n_limbs
    one = mont_arr(f, [1])
    xs = mont_arr(f, list(range(25)))
    neg = ol.field_unop(f.field_id, "neg", xs)
    assert array_to_ints(ol.field_binop(f.field_id, "add", xs, neg)) == [0] * 25
    inv = ol.field_unop(f.field_id, "inverse", xs[1:])
    prod = ol.field_binop(f.field_id, "mul", xs[1:], inv)
    assert all(list(r) == list(one[0]) for r in prod)
    binv = ol.batch_inverse(f.field_id, xs[1:])
    assert np.array_equal(binv, inv)
    assert from_mont_arr(f, inv) == [pow(i, -1, f.p) for i in range(1, 25)]


# ---- 8. test_arithmetic! sweep (field.rs:498-615, 618-780), WORD_BITS = 32 ----
def reference_test_inputs(modulus, word_bits=32):
    modwords = -(-modulus.bit_length() // word_bits)
    smalls = list(range(10))
    word_max = (1 << word_bits) - 1
    bigs = [word_max - x for x in smalls]
    one_words = smalls + bigs
    multiple_words = [x << (word_bits * i) for i in range(1, modwords) for x in one_words]
    basic = one_words + multiple_words
    maxval = (1 << (modwords * word_bits)) - 1
    diff_max = [maxval - x for x in basic if maxval - x < modulus]
    diff_mod = [modulus - x for x in basic if x < modulus and x != 0]
    basics = [x for x in basic if x < modulus]
    return basics + diff_max + diff_mod


def test_reference_test_inputs_count():
    assert len(reference_test_inputs(br.TWEEDLEDEE_BASE.p)) == 302


@pytest.mark.parametrize("f", FIELDS, ids=lambda f: f.name)
def test_arithmetic_sweep(f):
    inputs = reference_test_inputs(f.p)
    m = len(inputs)
    x = mont_arr(f, inputs)
    # unary: neg, square on every input
    assert from_mont_arr(f, ol.field_unop(f.field_id, "neg", x)) == [(-v) % f.p for v in inputs]
    assert from_mont_arr(f, ol.field_unop(f.field_id, "square", x)) == [v * v % f.p for v in inputs]
    assert from_mont_arr(f, ol.field_unop(f.field_id, "double", x)) == [2 * v % f.p for v in inputs]
    assert from_mont_arr(f, ol.field_unop(f.field_id, "triple", x)) == [3 * v % f.p for v in inputs]
    # binary: all rotations == all ordered pairs
    idx_a = np.repeat(np.arange(m), m)
    idx_b = np.tile(np.arange(m), m)
    a, b = x[idx_a], x[idx_b]
    R, p = f.R, f.p
    am = [f.to_mont(inputs[i]) for i in range(m)]
    Rinv = f.Rinv
    got_add = array_to_ints(ol.field_binop(f.field_id, "add", a, b))
    got_sub = array_to_ints(ol.field_binop(f.field_id, "sub", a, b))
    got_mul = array_to_ints(ol.field_binop(f.field_id, "mul", a, b))
    k = 0
    for i in range(m):
        ai = am[i]
        for j in range(m):
            bj = am[j]
            assert got_add[k] == (ai + bj) % p
            assert got_sub[k] == (ai - bj) % p
            assert got_mul[k] == ai * bj * Rinv % p
            k += 1


# ---- 9. summation edge cases (curve_summations.rs:164-184) ----
@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_affine_summation_edge_cases(c):
    G = (c.gx, c.gy)
    G2 = br.ec_mul(c, 2, G)
    G3 = br.ec_mul(c, 3, G)

    def arr(pts):
        return np.array([[c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])] for P in pts], dtype=np.uint64).reshape(-1, 2, c.base.n_limbs)

    for mode in ("pairwise", "batch_inversion", "best"):
        out, z = ol.affine_summation(c.curve_id, mode, arr([G, G]))
        assert z == 0 and tuple(from_mont_arr(c.base, out)) == G2
        out, z = ol.affine_summation(c.curve_id, mode, arr([G, G2]))
        assert z == 0 and tuple(from_mont_arr(c.base, out)) == G3
        out, z = ol.affine_summation(c.curve_id, mode, arr([G, G, G]))
        assert z == 0 and tuple(from_mont_arr(c.base, out)) == G3
        out, z = ol.affine_summation(c.curve_id, mode, arr([]))
        assert z == 1
        out, z = ol.affine_summation(c.curve_id, mode, arr([G, br.ec_neg(c, G)]))
        assert z == 1
        # identity operands through the zero flag
        out, z = ol.affine_summation(c.curve_id, mode, arr([G, G2, G3]), zero=[0, 1, 0])
        assert z == 0 and tuple(from_mont_arr(c.base, out)) == br.ec_mul(c, 4, G)


# ---- 10. group law (bls12_377_curve.rs:40-63, tweedledee_curve.rs:64-74) ----
@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_group_law(c):
    gen = ol.curve_generator(c.curve_id)
    G = tuple(from_mont_arr(c.base, gen))
    assert G == (c.gx, c.gy) and br.ec_on_curve(c, G)
    # affine double vs projective double on a chain of points
    P, Pxy = G, gen
    for _ in range(20):
        d, z = ol.affine_double(c.curve_id, Pxy)
        P = br.ec_add(c, P, P)
        assert z == 0 and tuple(from_mont_arr(c.base, d)) == P
        Pxy = d
    # 10 * G == G + ... + G
    ten = mont_arr(c.scalar, [10])[0]
    acc, accxy, accz = None, gen * 0, 1
    for _ in range(10):
        accxy, accz = ol.affine_add(c.curve_id, accxy, accz, gen, 0)
    m1, z1 = ol.mul_naive(c.curve_id, ten, gen)
    m2, z2 = ol.scalar_mul(c.curve_id, ten, gen)
    assert np.array_equal(m1, accxy) and np.array_equal(m2, accxy) and z1 == z2 == accz == 0
    # fixed scalar (bls12_377_curve.rs:57-62) and boundary scalars
    r = c.scalar.p
    for s in (limbs_to_int([11111111, 22222222, 33333333, 44444444]) % r, 0, 1, r - 1):
        sm = mont_arr(c.scalar, [s])[0]
        a, za = ol.scalar_mul(c.curve_id, sm, gen)
        b, zb = ol.mul_naive(c.curve_id, sm, gen)
        exp = br.ec_mul(c, s, G)
        assert za == zb == (1 if exp is None else 0)
        if exp is not None:
            assert tuple(from_mont_arr(c.base, a)) == exp and tuple(from_mont_arr(c.base, b)) == exp


def test_endomorphism_constants():
    # [zeta_q] P == (zeta_p * x, y)  (tweedledee_curve.rs:22-37,64-74 ; tweedledum_curve.rs:38-52,79-89)
    zetas = {
        0: ([1444470991491022206, 3301226169728360777, 72516509137424193, 708688398506307241],
            [13597504620482004229, 16590497220115833568, 15137822970486674306, 1901757351910266741]),
        1: ([7605997034305223424, 3132214451552427455, 3308921103222877309, 2709928666517121162],
            [9282944046338294407, 16421485501699768486, 18374227564572127422, 3902997619921080662]),
    }
    for cid, (zp, zq) in zetas.items():
        c = br.CURVES[cid]
        G = (c.gx, c.gy)
        zeta_p = c.base.from_mont(limbs_to_int(zp))
        zeta_q = c.scalar.from_mont(limbs_to_int(zq))
        assert br.ec_mul(c, zeta_q, G) == (zeta_p * G[0] % c.base.p, G[1])
        _check_glv_params(c, zeta_p, zeta_q)


def _check_glv_params(c, zeta_p, zeta_q):
    """plonky_amd/csrc/glv_params.cuh (derived from the moduli by tools/gen_glv_params.py) holds the reference's endomorphism:
    (BETA, LAMBDA) is (ZETA, ZETA_SCALAR) or the other primitive pair (ZETA^2, ZETA_SCALAR^2)."""
    import os, re
    txt = open(os.path.join(os.path.dirname(__file__), "..", "plonky_amd", "csrc", "glv_params.cuh")).read()
    blk = txt[txt.index("struct %sGlv" % c.name):]
    blk = blk[:blk.index("\n};")]

    def const(name):
        m = re.search(r"%s\[8\] = \{([^}]*)\}" % name, blk)
        return sum(int(w.strip().rstrip("u"), 16) << (32 * i) for i, w in enumerate(m.group(1).split(",")))

    p, r = c.base.p, c.scalar.p
    assert (const("BETA"), const("LAMBDA")) in ((zeta_p, zeta_q), (zeta_p * zeta_p % p, zeta_q * zeta_q % r))


# ---- seeded generator: C++ and Python agree; MSM oracle (reference algorithm) vs big-int maths ----
@pytest.mark.parametrize("f", FIELDS, ids=lambda f: f.name)
def test_rand_field_agrees(f):
    a = ol.rand_field(f.field_id, 0xF70014, 64)
    b = br.rand_field_limbs(f, 0xF70014, 64)
    assert [list(map(int, r)) for r in a] == b
    assert all(limbs_to_int(r) < f.p for r in a)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_msm_oracle_vs_bigint_random(c):
    n = 24
    G = (c.gx, c.gy)
    d = limbs_to_int(ol.rand_field(c.scalar.field_id, 7, 1)[0])
    D = br.ec_mul(c, d, G)
    gxy = np.array([c.base.mont_limbs(G[0]), c.base.mont_limbs(G[1])], dtype=np.uint64)
    dxy = np.array([c.base.mont_limbs(D[0]), c.base.mont_limbs(D[1])], dtype=np.uint64)
    bases = ol.gen_bases(c.curve_id, n, gxy, dxy)
    pts, P = [], G
    for i in range(n):
        pts.append(P)
        P = br.ec_add(c, P, D)
    assert [tuple(from_mont_arr(c.base, b)) for b in bases] == pts
    scal_m = ol.rand_field(c.scalar.field_id, 0x350020, n)
    # edge scalars: 0, 1, r-1 + a duplicate base
    scal_m[0] = mont_arr(c.scalar, [0])[0]
    scal_m[1] = mont_arr(c.scalar, [1])[0]
    scal_m[2] = mont_arr(c.scalar, [c.scalar.p - 1])[0]
    bases[5] = bases[4]
    pts[5] = pts[4]
    scal = from_mont_arr(c.scalar, scal_m)
    exp = br.msm(c, scal, pts)
    for w in (5, 11):
        pre = ol.MsmPrecomputation(c.curve_id, bases, w)
        # table contents: powers[i][j] = [2^(w j)] G_i
        e, z = pre.table_entry(3, 2)
        assert tuple(from_mont_arr(c.base, e)) == br.ec_mul(c, 1 << (2 * w), pts[3])
        o1, z1, p1 = pre.execute(scal_m, parallel=False, want_projective=True)
        o2, z2, p2 = pre.execute(scal_m, parallel=True, threads=2, want_projective=True)
        assert z1 == z2 == 0
        assert tuple(from_mont_arr(c.base, o1)) == exp and np.array_equal(o1, o2)
