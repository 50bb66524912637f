#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz: small committed input/expected-output vectors for the NTT and
MSM path.  The reference is Rust and cannot run here, so the expected values come from the
independent big-integer maths of oracle/bigint_ref.py (NOT from the C++ restatement and NOT from
the HIP code): both of those are then checked against these files
(tests/test_golden.py on CPU, tests/test_gpu_parity.py::test_golden_vectors on the GPU).
Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bigint_ref as br  # noqa: E402
from plonky_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def arr(vals, n):
    return np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)] for v in vals], dtype=np.uint64)


def main():
    # NTT: 2^6 points per field, seeded Montgomery-limb inputs, forward and inverse
    for f in (br.TWEEDLEDEE_BASE, br.TWEEDLEDUM_BASE, br.BLS12_377_SCALAR):
        x = synth.rand_field(f.field_id, 0x601D0000 + f.field_id, 64)
        xi = [f.from_mont(synth.to_int(r)) for r in x]
        fwd = arr([f.to_mont(v) for v in br.ntt(f, xi)], 4)
        inv = arr([f.to_mont(v) for v in br.intt(f, xi)], 4)
        np.savez(os.path.join(OUT, "ntt_%s_2p6.npz" % f.name), field=f.field_id, input=x, forward=fwd, inverse=inv)
    # MSM: 24 generators G0 + i D, seeded scalars with the edge values 0, 1, r-1 and a duplicated base
    for c in (br.TWEEDLEDEE, br.TWEEDLEDUM, br.BLS12_377):
        n = 24
        G = (c.gx, c.gy)
        D = br.ec_mul(c, 0x5EED + c.curve_id, G)
        pts, P = [], G
        for _ in range(n):
            pts.append(P)
            P = br.ec_add(c, P, D)
        pts[5] = pts[4]
        s = synth.rand_field(c.scalar.field_id, 0x601D1000 + c.curve_id, n)
        sc = [c.scalar.from_mont(synth.to_int(r)) for r in s]
        sc[0], sc[1], sc[2] = 0, 1, c.scalar.p - 1
        s = arr([c.scalar.to_mont(v) for v in sc], 4)
        res = br.msm(c, sc, pts)
        L = c.base.n_limbs
        bases = np.stack([arr([c.base.to_mont(P[0]), c.base.to_mont(P[1])], L) for P in pts])
        expect = arr([c.base.to_mont(res[0]), c.base.to_mont(res[1])], L)
        np.savez(os.path.join(OUT, "msm_%s_24.npz" % c.name), curve=c.curve_id, bases=bases, scalars=s, expected_xy=expect, expected_zero=0)
    # polynomial callers: m = q * (X^n - 1) with n = 12 (not a power of two: 64 distinct denominators) and the
    # product of two short polynomials, per NTT field; expected values by exact synthetic division / schoolbook
    for f in (br.TWEEDLEDEE_BASE, br.TWEEDLEDUM_BASE, br.BLS12_377_SCALAR):
        q = [f.from_mont(synth.to_int(r)) for r in synth.rand_field(f.field_id, 0x601D2000 + f.field_id, 37)]
        n = 12
        m = [0] * (len(q) + n)
        for i, cq in enumerate(q):
            m[i + n] = (m[i + n] + cq) % f.p
            m[i] = (m[i] - cq) % f.p
        quot = br.poly_divide_by_z_h_exact(f, m, n)
        assert quot == q
        size = 1 << (len(m) - 1).bit_length()
        a = [f.from_mont(synth.to_int(r)) for r in synth.rand_field(f.field_id, 0x601D3000 + f.field_id, 21)]
        b = [f.from_mont(synth.to_int(r)) for r in synth.rand_field(f.field_id, 0x601D4000 + f.field_id, 30)]
        prod = br.poly_mul_schoolbook(f, a, b)
        psize = 1 << (len(prod) - 1).bit_length()
        np.savez(os.path.join(OUT, "poly_%s.npz" % f.name), field=f.field_id, n=n,
                 numerator=arr([f.to_mont(v) for v in m], 4), quotient=arr([f.to_mont(v) for v in quot + [0] * (size - len(quot))], 4),
                 a=arr([f.to_mont(v) for v in a], 4), b=arr([f.to_mont(v) for v in b], 4),
                 product=arr([f.to_mont(v) for v in prod + [0] * (psize - len(prod))], 4))
    # byte encodings (serialization.rs:17-72): canonical little-endian field bytes; points as mask byte + x, from big-int maths
    for c in (br.TWEEDLEDEE, br.TWEEDLEDUM, br.BLS12_377):
        f, L = c.base, c.base.n_limbs
        G = (c.gx, c.gy)
        vals = [0, 1, 2, f.p - 1, f.p // 2, 0x0123456789ABCDEF << 64 | 0xFEDCBA9876543210] + [f.from_mont(synth.to_int(r)) for r in synth.rand_field(f.field_id, 0x601D5000 + f.field_id, 6)]
        fbytes = np.array([list(v.to_bytes(8 * L, "little")) for v in vals], dtype=np.uint8)
        pts = [G, br.ec_neg(c, G)] + [br.ec_mul(c, k, G) for k in (2, 3, 0xDEADBEEF, c.scalar.p - 5)]
        zero = [0] * len(pts) + [1]
        pts.append((0, 0))
        pbytes = np.array([[(1 if z else 0) | (2 if P[1] & 1 else 0)] + list(P[0].to_bytes(8 * L, "little")) for P, z in zip(pts, zero)], dtype=np.uint8)
        np.savez(os.path.join(OUT, "bytes_%s.npz" % c.name), curve=c.curve_id, field=f.field_id, elems=arr([f.to_mont(v) for v in vals], L), elem_bytes=fbytes,
                 points=np.stack([arr([f.to_mont(P[0]), f.to_mont(P[1])], L) for P in pts]), points_zero=np.array(zero, dtype=np.uint8), point_bytes=pbytes)
    msm_seeded_generators()
    print("golden vectors written to", OUT)


# (curve, n, device window the test asks for: 0 = the library's own choice).  Round 6 added the reference's own benchmark curve
# (src/bin/msms.rs:12 is Tweedledum) at 2^18 / 2^20, a Pasta curve at 2^18, and two lengths that are NOT powers of two at the c = 20
# geometry: 1 000 003 (the reference takes any length, curve_msm.rs:102-107) and 349 525 = a third of 2^20, what a rank of three holds.
SEEDED_MSM = [(br.TWEEDLEDEE, 1 << 16, 0), (br.TWEEDLEDEE, 1 << 18, 0), (br.TWEEDLEDEE, 1 << 20, 0),
              (br.BLS12_377, 1 << 16, 0), (br.BLS12_377, 1 << 18, 0), (br.BLS12_377, 1 << 20, 0),
              (br.TWEEDLEDUM, 1 << 18, 0), (br.TWEEDLEDUM, 1 << 20, 0), (br.PALLAS, 1 << 18, 0),
              (br.TWEEDLEDEE, 1000003, 0), (br.TWEEDLEDUM, 349525, 20)]


def seeded_name(c, n):
    return "seeded_msm_%s_%s.npz" % (c.name, "2p%d" % (n.bit_length() - 1) if n & (n - 1) == 0 else "n%d" % n)


def msm_seeded_generators(only_missing=False):
    """MSMs at production geometry over generators WITHOUT structure (curve_msm.rs:218-241 tests arbitrary generators): G_i = [h_i] G
    for seeded h_i, seeded scalars s_i - the fixture holds only the seeds and the expected affine point [sum s_i h_i mod r] G, one
    scalar multiplication on Python integers (the generators lie in the r-subgroup).  The tests rebuild
    the generators from the seeds with the oracle's scalar multiplication (curve_multiplication.rs)."""
    for c, n, window in SEEDED_MSM:
        path = os.path.join(OUT, seeded_name(c, n))
        if only_missing and os.path.exists(path):
            continue
        f, r = c.scalar, c.scalar.p
        log_n = n.bit_length() - 1  # the seeds of the power-of-two cases are those of rounds 4 / 5; the ragged lengths mix n in
        tag = log_n if n & (n - 1) == 0 else 0x40 + (n % 61)
        seed_h, seed_s = 0x601D6000 + 0x100 * c.curve_id + tag, 0x601D7000 + 0x100 * c.curve_id + tag
        h = synth.rand_field(f.field_id, seed_h, n)
        sv = synth.rand_field(f.field_id, seed_s, n)
        hb, sb = h.tobytes(), sv.tobytes()
        acc = 0
        for i in range(n):  # Montgomery limbs as integers: (s R)(h R) summed, R^-2 taken off once
            acc += int.from_bytes(sb[32 * i:32 * i + 32], "little") * int.from_bytes(hb[32 * i:32 * i + 32], "little")
        rinv = pow(1 << 256, -1, r)
        k = acc % r * rinv % r * rinv % r
        P = br.ec_mul(c, k, (c.gx, c.gy))
        L = c.base.n_limbs
        np.savez(path, curve=c.curve_id, log_n=log_n, n=n, device_window=window, seed_h=seed_h, seed_s=seed_s,
                 expected_xy=arr([c.base.to_mont(P[0]), c.base.to_mont(P[1])], L), expected_zero=0)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "seeded-missing":
        msm_seeded_generators(only_missing=True)  # adds fixtures without touching the committed ones
    else:
        main()
