"""Closed-form self-check for the synthetic MSM workload (harness code, not the product path).

With bases B_i = G0 + i*D the MSM has the closed form
    sum_i s_i B_i = [sum_i s_i] G0 + [sum_i i s_i] D
so a full-size (2^20 / 2^22) run can be verified with two scalar multiplications on Python
integers, without any CPU MSM.  Scalars arrive as Montgomery limbs m_i = s_i R mod r, and the two
sums are linear, so sum s_i = R^-1 sum m_i.  Curve data: tweedledee_curve.rs:11-18,
tweedledum_curve.rs:11-33, bls12_377_curve.rs:14-33 (a = 0 for all three).
"""
import numpy as np

from .synth import MODULI

CURVE_BASE = {0: 0, 1: 1, 2: 3, 3: 4, 4: 5}
CURVE_SCALAR = {0: 1, 1: 0, 2: 2, 3: 5, 4: 4}
# GENERATOR_AFFINE of the curves the harness uses, canonical integers: tweedledee_curve.rs:14-18 (NEG_ONE, TWO) and
# bls12_377_curve.rs:16-33 (the decimal values of its doc comments)
GENERATORS = {
    0: (MODULI[0] - 1, 2),
    2: (81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
        241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030),
}


def _add(p, P, Q):
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
    return (x3, (lam * (x1 - x3) - y1) % p)


def _mul(p, k, P):
    acc, add = None, P
    while k:
        if k & 1:
            acc = _add(p, acc, add)
        add = _add(p, add, add)
        k >>= 1
    return acc


def limb_sums(scalars_mont, first=0):
    """(sum m_i, sum (first + i) m_i) as Python integers, from (n, 4) uint64 Montgomery limbs."""
    a = np.ascontiguousarray(scalars_mont, dtype=np.uint64).reshape(-1, 4)
    n = a.shape[0]
    # split limbs into 32-bit halves so the column sums stay exact in Python ints via float-free numpy
    lo = (a & np.uint64(0xFFFFFFFF)).astype(np.uint64)
    hi = (a >> np.uint64(32)).astype(np.uint64)
    idx = np.arange(first, first + n, dtype=np.uint64)
    s_plain = 0
    s_weight = 0
    for k in range(4):
        # sums of up to 2^22 values below 2^32 fit in uint64; weighted sums are done in two 16-bit index halves
        s_plain += (int(lo[:, k].sum(dtype=np.uint64)) + (int(hi[:, k].sum(dtype=np.uint64)) << 32)) << (64 * k)
        il, ih = idx & np.uint64(0xFFFF), idx >> np.uint64(16)
        w = 0
        for part, sh in ((lo[:, k], 0), (hi[:, k], 32)):
            pl, ph = part & np.uint64(0xFFFF), part >> np.uint64(16)
            for pv, ps in ((pl, 0), (ph, 16)):
                for iv, ish in ((il, 0), (ih, 16)):
                    w += int((pv * iv).sum(dtype=np.uint64)) << (sh + ps + ish)
        s_weight += w << (64 * k)
    return s_plain, s_weight


def closed_form_msm(curve, scalars_mont, G0, D, first=0):
    """Expected affine point (canonical integers) or None for the identity."""
    p = MODULI[CURVE_BASE[curve]]
    r = MODULI[CURVE_SCALAR[curve]]
    rinv = pow(1 << 256, -1, r)
    sp, sw = limb_sums(scalars_mont, first)
    a = sp % r * rinv % r
    b = sw % r * rinv % r
    return _add(p, _mul(p, a, G0), _mul(p, b, D))
