"""Seeded synthetic inputs for the harness.

The reference draws from OsRng (benches/fft.rs:23, src/bin/msms.rs:31) and is irreproducible, so
the harness uses SplitMix64 + the reference's own rejection sampling (rand_range_from_rng,
src/bigint/bigint_arithmetic.rs:98-117): draw N u64 limbs, shift the top limb right by the
modulus' leading zeros, retry while >= p.  The limbs are used as MONTGOMERY limbs, exactly what
F::rand() does (tweedledee_base.rs:195-199).  Vectorised numpy; oracle/bigint_ref.py and
oracle/plk_oracle.cpp hold the scalar versions and the tests check all three agree.
"""
import numpy as np

MODULI = {
    0: 28948022309329048855892746252171976963322203655954433126947083963168578338817,
    1: 28948022309329048855892746252171976963322203655955319056773317069363642105857,
    2: 8444461749428370424248824938781546531375899335154063827935233455917409239041,
    3: 258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177,
    4: 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001,  # PallasBase
    5: 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001,  # VestaBase
}
LIMBS = {0: 4, 1: 4, 2: 4, 3: 6, 4: 4, 5: 4}

_GAMMA = np.uint64(0x9E3779B97F4A7C15)


def _splitmix64(seed, count, start=0):
    """outputs start .. start+count-1 of the SplitMix64 stream seeded with `seed`."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _less_than(cand, p_limbs):
    """row-wise cand < p for little-endian u64 limb rows."""
    n = cand.shape[1]
    lt = np.zeros(cand.shape[0], dtype=bool)
    eq = np.ones(cand.shape[0], dtype=bool)
    for i in range(n - 1, -1, -1):
        pi = np.uint64(p_limbs[i])
        lt |= eq & (cand[:, i] < pi)
        eq &= cand[:, i] == pi
    return lt


def rand_field(field, seed, count):
    """(count, L) uint64: the first `count` accepted candidates of the seeded stream."""
    p = MODULI[field]
    L = LIMBS[field]
    p_limbs = [(p >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(L)]
    strip = 64 - (p >> (64 * (L - 1))).bit_length()
    out = np.empty((count, L), dtype=np.uint64)
    have, start = 0, 0
    while have < count:
        want = max(1024, int((count - have) * 2.3) + 64)
        raw = _splitmix64(seed, want * L, start * L).reshape(want, L).copy()
        raw[:, L - 1] >>= np.uint64(strip)
        acc = raw[_less_than(raw, p_limbs)]
        take = min(count - have, acc.shape[0])
        if take < acc.shape[0]:
            # stop exactly after the candidate that produced the last accepted element
            pass
        out[have:have + take] = acc[:take]
        have += take
        start += want
    return out


def to_int(limbs):
    v = 0
    for i, l in enumerate(limbs):
        v |= int(l) << (64 * i)
    return v


def to_limbs(v, L):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(L)]


def mont(field, v):
    L = LIMBS[field]
    p = MODULI[field]
    return np.array(to_limbs(v % p * (1 << (64 * L)) % p, L), dtype=np.uint64)


def from_mont(field, limbs):
    L = LIMBS[field]
    p = MODULI[field]
    return to_int(limbs) * pow(1 << (64 * L), -1, p) % p
