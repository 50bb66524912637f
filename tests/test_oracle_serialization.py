"""Pin the oracle's restatement of src/serialization.rs:17-72 (+ Field::square_root, field.rs:440-472): the reference's
own round-trip tests (serialization.rs:157-210) and error cases, against Python integers.  CPU only."""
import numpy as np
import pytest

from oracle import bigint_ref as br
from oracle import oracle_lib as ol
from tests.util import array_to_ints, ints_to_array

FIELDS = [br.TWEEDLEDEE_BASE, br.TWEEDLEDUM_BASE, br.BLS12_377_SCALAR, br.BLS12_377_BASE, br.PALLAS_BASE, br.VESTA_BASE]
CURVES = [br.TWEEDLEDEE, br.TWEEDLEDUM, br.BLS12_377, br.PALLAS, br.VESTA]


@pytest.mark.parametrize("f", FIELDS, ids=lambda f: f.name)
def test_field_serialization_round_trip(f):
    """test_field_serialization! (serialization.rs:157-175): x.write then read gives x; bytes are the canonical value."""
    x = ol.rand_field(f.field_id, 0x5E71A1, 50)
    b = ol.field_to_bytes(f.field_id, x)
    assert b.shape == (50, 8 * f.n_limbs)
    assert [int.from_bytes(bytes(r), "little") for r in b] == [f.from_mont(v) for v in array_to_ints(x)]
    back, bad = ol.field_from_bytes(f.field_id, b)
    assert bad == 0 and np.array_equal(back, x)
    # "Out of range" (field.rs:100): the modulus itself and all-ones are rejected, p - 1 is accepted
    raw = np.array([list(v.to_bytes(8 * f.n_limbs, "little")) for v in (f.p, (1 << (64 * f.n_limbs)) - 1, f.p - 1)], dtype=np.uint8)
    _, bad = ol.field_from_bytes(f.field_id, raw)
    assert bad == 2


@pytest.mark.parametrize("f", FIELDS, ids=lambda f: f.name)
def test_square_root(f):
    """field.rs:720-745: the square root of a square is +-the number; a non-residue has none."""
    for v in (0, 1, 4, 0xABCDEF, f.p - 1):
        sq = v * v % f.p
        r = ol.field_sqrt(f.field_id, ints_to_array([f.to_mont(sq)], f.n_limbs)[0])
        assert r is not None and f.from_mont(array_to_ints(r.reshape(1, -1))[0]) in (v % f.p, (-v) % f.p)
    nr = f.generator  # the multiplicative generator is a non-residue
    assert pow(nr, (f.p - 1) // 2, f.p) == f.p - 1
    assert ol.field_sqrt(f.field_id, ints_to_array([f.to_mont(nr)], f.n_limbs)[0]) is None


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_curve_serialization_round_trip(c):
    """test_curve_serialization! (serialization.rs:177-210): points and the point at infinity survive write / read."""
    f, L = c.base, c.base.n_limbs
    G = (c.gx, c.gy)
    pts = [br.ec_mul(c, k, G) for k in (1, 2, 5, 0xC0FFEE, c.scalar.p - 1)]
    xy = np.stack([ints_to_array([f.to_mont(P[0]), f.to_mont(P[1])], L) for P in pts] + [np.zeros((2, L), dtype=np.uint64)])
    zero = np.array([0] * len(pts) + [1], dtype=np.uint8)
    b = ol.point_to_bytes(c.curve_id, xy, zero)
    for i, P in enumerate(pts):
        assert b[i, 0] == (2 if P[1] & 1 else 0) and int.from_bytes(bytes(b[i, 1:]), "little") == P[0]
    assert b[-1, 0] == 1
    back, bz, status = ol.point_from_bytes(c.curve_id, b, L)
    assert not status.any() and np.array_equal(bz, zero) and np.array_equal(back, xy)
    # an x with no point on the curve -> "Invalid x coordinate"; an x >= p -> "Out of range"
    bad_x = next(x for x in range(2, 100) if pow((x ** 3 + c.b) % f.p, (f.p - 1) // 2, f.p) == f.p - 1)
    rec = np.array([[0] + list(bad_x.to_bytes(8 * L, "little")), [2] + list(f.p.to_bytes(8 * L, "little"))], dtype=np.uint8)
    _, _, status = ol.point_from_bytes(c.curve_id, rec, L)
    assert list(status) == [2, 1]
