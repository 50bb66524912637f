"""Independent big-integer restatement of the Plonky NTT/MSM hot path (TEST INFRASTRUCTURE ONLY).

This file is part of the *oracle*: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  It is never on the product path.

It restates, with plain Python integers, what the reference computes:
  * prime fields in Montgomery form          (reference src/field/monty.rs:21-178,
                                              src/field/tweedledee_base.rs:20-64,
                                              src/field/tweedledum_base.rs:20-64,
                                              src/field/bls12_377_base.rs:23-98,
                                              src/field/bls12_377_scalar.rs:23-96)
  * roots of unity                            (src/field/field.rs:429-435)
  * the natural-order forward / inverse DFT   (src/fft.rs:82-156)
  * short-Weierstrass group law, a = 0        (src/curve/curve.rs, src/curve/curve_adds.rs)
  * MSM digit convention and result           (src/curve/curve_msm.rs:63-180)

It is deliberately written from the mathematics (canonical integers mod p), not from the
limb-level algorithms, so it is independent of oracle/plk_oracle.cpp (which follows the
reference's limb algorithms line by line) and of the HIP kernels.  The three must agree.

Parity pinning: every constant below is a KAT from the reference source (file:line in the
comments); tests/test_oracle_kats.py re-derives each of them from the modulus alone.
"""

from dataclasses import dataclass

MASK64 = (1 << 64) - 1


def limbs_to_int(limbs):
    v = 0
    for i, l in enumerate(limbs):
        v |= int(l) << (64 * i)
    return v


def int_to_limbs(v, n):
    return [(v >> (64 * i)) & MASK64 for i in range(n)]


@dataclass(frozen=True)
class FieldSpec:
    name: str
    field_id: int          # id used by include/plonky_hip.h (PLK_FIELD_*)
    n_limbs: int           # u64 limbs
    p: int
    bits: int              # Field::BITS
    two_adicity: int
    generator: int         # MULTIPLICATIVE_SUBGROUP_GENERATOR (canonical)

    @property
    def r_bits(self):
        return 64 * self.n_limbs

    @property
    def R(self):
        return (1 << self.r_bits) % self.p

    @property
    def R2(self):
        return pow(1 << self.r_bits, 2, self.p)

    @property
    def R3(self):
        return pow(1 << self.r_bits, 3, self.p)

    @property
    def Rinv(self):
        return pow(self.R, -1, self.p)

    @property
    def mu(self):
        # -p^-1 mod 2^64  (monty.rs:35)
        return (-pow(self.p, -1, 1 << 64)) % (1 << 64)

    @property
    def T(self):
        return (self.p - 1) >> self.two_adicity

    def to_mont(self, x):
        return (x * self.R) % self.p

    def from_mont(self, m):
        return (m * self.Rinv) % self.p

    def mont_limbs(self, x):
        return int_to_limbs(self.to_mont(x % self.p), self.n_limbs)

    def primitive_root_of_unity(self, n_power):
        # field.rs:429-435:  g^T, then raised to 2^(adicity - n_power)
        assert n_power <= self.two_adicity
        base_root = pow(self.generator, self.T, self.p)
        return pow(base_root, 1 << (self.two_adicity - n_power), self.p)


# tweedledee_base.rs:22 ORDER (decimal in the doc comment), :117 BITS, :161 TWO_ADICITY, :155 generator 5
TWEEDLEDEE_BASE = FieldSpec(
    "TweedledeeBase", 0, 4,
    28948022309329048855892746252171976963322203655954433126947083963168578338817, 255, 34, 5)
# tweedledum_base.rs:22, :117, :161
TWEEDLEDUM_BASE = FieldSpec(
    "TweedledumBase", 1, 4,
    28948022309329048855892746252171976963322203655955319056773317069363642105857, 255, 33, 5)
# bls12_377_scalar.rs:25, :155, :171; generator limbs at :165 are the Montgomery form of 11
BLS12_377_SCALAR = FieldSpec(
    "Bls12377Scalar", 2, 4,
    8444461749428370424248824938781546531375899335154063827935233455917409239041, 253, 47, 11)
# bls12_377_base.rs:26, :170, :202; MULTIPLICATIVE_SUBGROUP_GENERATOR = FIVE (:199)
BLS12_377_BASE = FieldSpec(
    "Bls12377Base", 3, 6,
    258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177,
    377, 46, 5)

# pallas_base.rs:21-26 / vesta_base.rs:21-27 ORDER, :117 BITS, :161 TWO_ADICITY, :157 generator FIVE
PALLAS_BASE = FieldSpec("PallasBase", 4, 4, 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001, 255, 32, 5)
VESTA_BASE = FieldSpec("VestaBase", 5, 4, 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001, 255, 32, 5)

FIELDS = {f.field_id: f for f in (TWEEDLEDEE_BASE, TWEEDLEDUM_BASE, BLS12_377_SCALAR, BLS12_377_BASE, PALLAS_BASE, VESTA_BASE)}


@dataclass(frozen=True)
class CurveSpec:
    name: str
    curve_id: int
    base: FieldSpec
    scalar: FieldSpec
    b: int                 # y^2 = x^3 + b   (A = 0 for all three in-scope curves)
    gx: int
    gy: int


# tweedledee_curve.rs:11-18  A=0, B=5, G=(-1, 2)
TWEEDLEDEE = CurveSpec("Tweedledee", 0, TWEEDLEDEE_BASE, TWEEDLEDUM_BASE, 5,
                       TWEEDLEDEE_BASE.p - 1, 2)
# tweedledum_curve.rs:11-33  A=0, B=7, G=(1, y) with y given as Montgomery limbs
_TDUM_GY = TWEEDLEDUM_BASE.from_mont(limbs_to_int(
    [12815994359195135157, 12442237869110527732, 9256472484777506843, 1114242145010923164]))
TWEEDLEDUM = CurveSpec("Tweedledum", 1, TWEEDLEDUM_BASE, TWEEDLEDEE_BASE, 7, 1, _TDUM_GY)
# bls12_377_curve.rs:14-33  A=0, B=1, generator decimal in the doc comments
BLS12_377 = CurveSpec(
    "Bls12377", 2, BLS12_377_BASE, BLS12_377_SCALAR, 1,
    81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
    241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030)

# pallas_curve.rs:7-19, vesta_curve.rs:7-19: A = 0, B = 5, G = (-1, 2)
PALLAS = CurveSpec("Pallas", 3, PALLAS_BASE, VESTA_BASE, 5, PALLAS_BASE.p - 1, 2)
VESTA = CurveSpec("Vesta", 4, VESTA_BASE, PALLAS_BASE, 5, VESTA_BASE.p - 1, 2)

CURVES = {c.curve_id: c for c in (TWEEDLEDEE, TWEEDLEDUM, BLS12_377, PALLAS, VESTA)}


# ---------------------------------------------------------------------------------------------
# NTT  (src/fft.rs)
# ---------------------------------------------------------------------------------------------

def ntt_naive(f: FieldSpec, coeffs):
    """out[j] = sum_k c_k g^(jk), g = primitive_root_of_unity(log n)  (fft.rs:197-232 evaluate_naive)."""
    n = len(coeffs)
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    g = f.primitive_root_of_unity(log_n)
    out = []
    for j in range(n):
        x = pow(g, j, f.p)
        acc, xp = 0, 1
        for c in coeffs:
            acc = (acc + c * xp) % f.p
            xp = (xp * x) % f.p
        out.append(acc)
    return out


def ntt(f: FieldSpec, coeffs):
    """Same function as ntt_naive, O(n log n) recursive radix-2 (any correct algorithm is
    bit-exact because field elements are uniquely represented)."""
    n = len(coeffs)
    if n == 1:
        return list(coeffs)
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    g = f.primitive_root_of_unity(log_n)
    p = f.p

    def rec(a, w):
        m = len(a)
        if m == 1:
            return a
        e = rec(a[0::2], w * w % p)
        o = rec(a[1::2], w * w % p)
        out = [0] * m
        t = 1
        h = m // 2
        for k in range(h):
            x = t * o[k] % p
            out[k] = (e[k] + x) % p
            out[k + h] = (e[k] - x) % p
            t = t * w % p
        return out

    return rec(list(coeffs), g)


def intt(f: FieldSpec, points):
    """ifft_with_precomputation_power_of_2 (fft.rs:82-101): forward DFT, index reversal i<->n-i, times n^-1."""
    n = len(points)
    r = ntt(f, points)
    n_inv = pow(n, -1, f.p)
    out = [0] * n
    for i in range(n):
        out[i] = r[(n - i) % n] * n_inv % f.p
    return out


# ---------------------------------------------------------------------------------------------
# Curve arithmetic, affine with None = identity  (src/curve/curve.rs, curve_adds.rs)
# ---------------------------------------------------------------------------------------------

def ec_add(c: CurveSpec, P, Q):
    p = c.base.p
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    y3 = (lam * (x1 - x3) - y1) % p
    return (x3, y3)


def ec_neg(c: CurveSpec, P):
    if P is None:
        return None
    return (P[0], (-P[1]) % c.base.p)


def ec_mul(c: CurveSpec, k, P):
    """Plain double-and-add on the integer k >= 0 (mul_naive, bls12_377_curve.rs:65-85)."""
    acc = None
    add = P
    while k:
        if k & 1:
            acc = ec_add(c, acc, add)
        add = ec_add(c, add, add)
        k >>= 1
    return acc


def ec_on_curve(c: CurveSpec, P):
    if P is None:
        return True
    x, y = P
    return (y * y - x * x * x - c.b) % c.base.p == 0


def to_digits(c: CurveSpec, s_canonical, w):
    """curve_msm.rs:159-180: unsigned w-bit digits, LSB first, ceil(BITS/w) of them."""
    bits = c.scalar.bits
    num = (bits + w - 1) // w
    s = s_canonical & ((1 << bits) - 1)
    return [(s >> (i * w)) & ((1 << w) - 1) for i in range(num)]


def msm(c: CurveSpec, scalars_canonical, points):
    """Group element sum_i s_i * P_i with s_i taken as integers in [0, r)  (curve_msm.rs:63-100)."""
    acc = None
    for s, P in zip(scalars_canonical, points):
        acc = ec_add(c, acc, ec_mul(c, s, P))
    return acc


def msm_yao(c: CurveSpec, scalars_canonical, points, w):
    """The reference's Yao formulation (curve_msm.rs:63-100), used to cross-check msm()."""
    num = (c.scalar.bits + w - 1) // w
    powers = []
    for P in points:
        row = [P]
        for _ in range(1, num):
            q = row[-1]
            for _ in range(w):
                q = ec_add(c, q, q)
            row.append(q)
        powers.append(row)
    occ = [[] for _ in range(1 << w)]
    for i, s in enumerate(scalars_canonical):
        for j, d in enumerate(to_digits(c, s, w)):
            occ[d].append((i, j))
    y = None
    u = None
    for d in range((1 << w) - 1, 0, -1):
        for (i, j) in occ[d]:
            u = ec_add(c, u, powers[i][j])
        y = ec_add(c, y, u)
    return y


# ---------------------------------------------------------------------------------------------
# Seeded synthetic inputs: SplitMix64 + rejection sampling exactly like
# rand_range_from_rng (src/bigint/bigint_arithmetic.rs:98-117).  plonky_amd/synth.py implements
# the same generator vectorised; tests check the two agree.
# ---------------------------------------------------------------------------------------------

def splitmix64_stream(seed):
    state = seed & MASK64
    while True:
        state = (state + 0x9E3779B97F4A7C15) & MASK64
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        yield z ^ (z >> 31)


def rand_field_limbs(f: FieldSpec, seed, count):
    """count elements, each the raw limbs of a uniform value < p (interpreted by callers as
    Montgomery limbs, which is what F::rand() does - tweedledee_base.rs:195-199)."""
    strip = 64 - (f.p >> (64 * (f.n_limbs - 1))).bit_length()
    gen = splitmix64_stream(seed)
    out = []
    while len(out) < count:
        limbs = [next(gen) for _ in range(f.n_limbs)]
        limbs[-1] >>= strip
        if limbs_to_int(limbs) < f.p:
            out.append(limbs)
    return out


# ---------------------------------------------------------------------------------------------
# polynomial callers of the NTT (src/polynomial.rs) -- independent of the NTT: exact arithmetic
# ---------------------------------------------------------------------------------------------
def poly_trim(coeffs):
    """Polynomial::trim (src/polynomial.rs:178-180)."""
    k = len(coeffs)
    while k and coeffs[k - 1] == 0:
        k -= 1
    return list(coeffs[:k])


def poly_mul_schoolbook(f: FieldSpec, a, b):
    """Plain O(n m) product of canonical-int coefficient lists (what Polynomial::mul computes)."""
    if not a or not b:
        return []
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % f.p
    return out


def poly_divide_by_z_h_exact(f: FieldSpec, a, n):
    """a / (X^n - 1) by synthetic division, canonical ints; raises if the division is not exact.
    This is the mathematical meaning of Polynomial::divide_by_z_h (src/polynomial.rs:329-330)."""
    a = poly_trim(a)
    if not a:
        return []
    if len(a) <= n:
        raise ValueError("not divisible by Z_H")
    q = [0] * (len(a) - n)
    rem = list(a)
    for i in range(len(a) - 1, n - 1, -1):
        c = rem[i]
        q[i - n] = c
        rem[i] = 0
        rem[i - n] = (rem[i - n] + c) % f.p
    if any(rem):
        raise ValueError("not divisible by Z_H")
    return q


def poly_eval(f: FieldSpec, a, x):
    """Polynomial::eval (src/polynomial.rs:124-126), Horner."""
    acc = 0
    for c in reversed(a):
        acc = (acc * x + c) % f.p
    return acc


# ---------------------------------------------------------------------------------------------
# The Plonk quotient numerator, from the mathematics of the gates:
#   gates/mod.rs:46-125,289-300, gates/*.rs, mds.rs:63-77, plonk_util.rs:7-33, plonk.rs:392-453.
# Canonical integers mod p, written table-driven (a gate = prefix bits + a function returning its constraint
# list) so that it shares no structure with oracle/plonk_gates.inc or plonky_amd/csrc/plonk.hip.
# ---------------------------------------------------------------------------------------------
NUM_WIRES, NUM_ROUTED_WIRES, NUM_CONSTANTS, GRID_WIDTH = 9, 6, 6, 65  # plonk.rs:21-25


def _mds4(p, r, c):
    return pow(4 + r - c, -1, p)  # Cauchy matrix, x_r = 4 + r, y_c = c (mds.rs:63-77)


def _g_curve_add(p, k, l, r, b, zeta, a):
    x1, y1, acc_old, acc_new, x2, y2, bit, inv, lam = l
    x4, y4 = r[0], r[1]
    x3 = lam * lam - x1 - x2
    y3 = lam * (x1 - x4) - y1
    return [(y1 - y2) * inv - lam, bit * x3 + (1 - bit) * x1 - x4, bit * y3 + (1 - bit) * y1 - y4, acc_new - (2 * acc_old + bit),
            bit * (1 - bit), inv * (x1 - x2) - 1]


def _g_curve_dbl(p, k, l, r, b, zeta, a):
    xo, yo, xn, yn, inv, lam = l[:6]
    return [(3 * xo * xo + a) * inv - lam, lam * lam - 2 * xo - xn, lam * (xo - xn) - yo - yn, 2 * yo * inv - 1]


def _g_curve_endo(p, k, l, r, b, zeta, a):
    x1, y1, un_old, sg_old, x_in, y_in, b0, b1, inv = l
    x3, y3 = r[0], r[1]
    mult = (zeta - 1) * b1 + 1
    x2, y2 = mult * x_in, (2 * b0 - 1) * y_in
    lam = (y1 - y2) * inv
    return [lam * lam - x1 - x2 - x3, lam * (x1 - x3) - y1 - y3, b[2] - (4 * un_old + 2 * b1 + b0), b[3] - (2 * sg_old + (2 * b0 - 1) * mult),
            b0 * (b0 - 1), b1 * (b1 - 1), inv * (x1 - x2) - 1]


def _g_base4(p, k, l, r, b, zeta, a):
    acc = l[0]
    for limb in l[2:]:
        acc = 4 * acc + limb
    return [acc - l[1]] + [limb * (limb - 1) * (limb - 2) * (limb - 3) for limb in l[2:]]


def _g_rescue_a(p, k, l, r, b, zeta, a):
    out = []
    for i in range(4):
        out.append(pow(l[4 + i], 5, p) - l[i])
        out.append(k[2 + i] + sum(_mds4(p, i, j) * l[4 + j] for j in range(4)) - r[i])
    return out


def _g_rescue_b(p, k, l, r, b, zeta, a):
    return [k[2 + i] + sum(_mds4(p, i, j) * pow(l[j], 5, p) for j in range(4)) - r[i] for i in range(4)]


# (prefix bits, constraint function) in the order of evaluate_all_constraints (gates/mod.rs:52-113)
PLONK_GATES = [
    ("10101", _g_curve_add), ("10111", _g_curve_dbl), ("11", _g_curve_endo), ("1000", _g_base4),
    ("101001", lambda p, k, l, r, b, zeta, a: [l[6 + i] - r[i] for i in range(3)]),       # PublicInputGate
    ("101000", lambda p, k, l, r, b, zeta, a: []),                                           # BufferGate (buffer.rs:27)
    ("10110", lambda p, k, l, r, b, zeta, a: [k[5] - l[0]]),                                 # ConstantGate
    ("1001", lambda p, k, l, r, b, zeta, a: [k[4] * l[0] * l[1] + k[5] * l[2] - l[3]]),      # ArithmeticGate
    ("00", _g_rescue_a), ("01", _g_rescue_b),
]


def plonk_gate_filtered(f: FieldSpec, gate, k, l, r, b, zeta, a, unfiltered=False):
    p = f.p
    prefix, fn = PLONK_GATES[gate]
    filt = 1
    for bit, c in zip(prefix, k):
        filt = filt * (c if bit == "1" else 1 - c) % p
    return [(v if unfiltered else filt * v) % p for v in fn(p, k, l, r, b, zeta, a)]


def plonk_all_constraints(f: FieldSpec, k, l, r, b, zeta, a):
    out = []
    for g in range(len(PLONK_GATES)):
        cs = plonk_gate_filtered(f, g, k, l, r, b, zeta, a)
        out += [0] * (len(cs) - len(out))
        for i, v in enumerate(cs):
            out[i] = (out[i] + v) % f.p
    return out


def plonk_eval_l_1(f: FieldSpec, n, x):
    if x % f.p == 1:
        return 1
    return (pow(x, n, f.p) - 1) * pow(n * (x - 1), -1, f.p) % f.p


def plonk_vanishing_points(f: FieldSpec, degree, constants, wires, s_sigma, z, k_is, alpha, beta, gamma, zeta, a):
    """plonk.rs:392-453 on canonical integers; tables are lists of rows of 8 * degree values."""
    p, n8 = f.p, 8 * degree
    g = f.primitive_root_of_unity(n8.bit_length() - 1)
    out = []
    x = 1
    for i in range(n8):
        ir, ib = (i + 8) % n8, (i + 8 * GRID_WIDTH) % n8
        k = [constants[j][i] for j in range(NUM_CONSTANTS)]
        l = [wires[j][i] for j in range(NUM_WIRES)]
        r = [wires[j][ir] for j in range(NUM_WIRES)]
        b = [wires[j][ib] for j in range(NUM_WIRES)]
        terms = [plonk_eval_l_1(f, degree, x) * (z[i] - 1) % p]
        fp = gp = 1
        for j in range(NUM_ROUTED_WIRES):
            fp = fp * (l[j] + beta * k_is[j] * x + gamma) % p
            gp = gp * (l[j] + beta * s_sigma[j][i] + gamma) % p
        terms.append((fp * z[i] - gp * z[ir]) % p)
        terms += plonk_all_constraints(f, k, l, r, b, zeta, a)
        acc = 0
        for t in reversed(terms):
            acc = (acc * alpha + t) % p
        out.append(acc)
        x = x * g % p
    return out
