"""GPU parity: the round-6 bucket ordering (tile-major level 1, bins from the low bits of the bucket number, whole bins ordered in LDS,
segmented kernels for hot bins) and reduction (list-driven assembly, tree levels inside the row / column sums, transposed bucket grid,
one workgroup for both windows of the end) at the geometry they are built for - the 20-bit window - against the oracle.

The default window reaches 20 bits only from 2^19 generators on, where the oracle takes a minute per vector; the new kernels are selected
by the WINDOW and the tile count, not by the size, so `device_window=20` over 2^16 .. 2^17 generators runs exactly the code a 2^20 MSM
runs (msm_configure: OrdCfg::perm needs >= 2^16 scalars) at sizes the oracle answers in seconds.  What the uniform vectors of the other
suites never reach (curve_msm.rs:102-157 takes ANY scalars; a witness is full of zeros, ones and small values):
  * every scalar the same, 16 distinct scalars: a few buckets hold everything - hot bins (more than 32768 entries: the segmented
    level-2 kernels, many workgroups per bin), heavy buckets (k_msm_heavy_*), thousands of live head pieces in the list of k_msm_heads;
  * scalars below 2^10 / 2^20: only the lowest window has entries - with the bins taken from the LOW bits they still spread over bins;
  * the top window alone (multiples of 2^240), zeros everywhere else: empty bins that still own bucket offsets;
  * a sub-range of the generators (plk_msm_execute_parts_dev: ent_first > 0, fewer tiles than the context was laid out for);
  * lengths that are not a multiple of the 1024-scalar tile.
Each against the oracle's msm_execute_parallel; the same vectors through round 5's kernels (PLK_MSM_ORDER_V1 / PLK_MSM_TAIL_V1) run in
tests/test_gpu_knobs.py's processes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import plonky_amd as pa
from plonky_amd import synth
from oracle import bigint_ref as br
from oracle import oracle_lib as ol
from tests.test_oracle_kats import mont_arr

THREADS = min(64, os.cpu_count() or 1)


def _bases(c, n, seed):
    G = (c.gx, c.gy)
    pt = lambda P: np.array([c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])], dtype=np.uint64)
    return ol.gen_bases(c.curve_id, n, pt(G), pt(br.ec_mul(c, 0xB1A5 + seed, G)))


def _vectors(c, n, seed):
    f = c.scalar
    rnd = synth.rand_field(f.field_id, 0x6F0000 + seed, n)
    rng = np.random.default_rng(seed)
    m = lambda vals: mont_arr(f, [int(v) % f.p for v in vals])
    out = {"uniform": rnd}
    out["all_equal"] = np.repeat(rnd[7:8], n, axis=0)
    out["all_one"] = np.repeat(m([1]), n, axis=0)
    pick = rnd[:16]
    out["sixteen_distinct"] = pick[rng.integers(0, 16, size=n)]
    small = m(rng.integers(0, 1 << 10, size=n))
    out["below_2p10"] = small
    out["below_2p20"] = m(rng.integers(0, 1 << 20, size=n))
    out["top_window_only"] = m([(int(v) % 16000) << 240 for v in rng.integers(1, 1 << 30, size=n)])
    half = rnd.copy()
    half[rng.random(n) < 0.5] = 0
    out["half_zero"] = half
    mixed = rnd.copy()
    mixed[: n // 3] = m([1])[0]
    mixed[n // 3: n // 2] = m([f.p - 1])[0]
    out["ones_minus_ones_random"] = mixed
    return out


@pytest.mark.parametrize("c,n", [(br.TWEEDLEDEE, (1 << 16) + 37), (br.TWEEDLEDUM, 1 << 17), (br.BLS12_377, (1 << 16) + 1024)],
                         ids=["Tweedledee_2p16+37", "Tweedledum_2p17", "Bls12377_2p16+1024"])
def test_window20_orderings_and_reduction_match_oracle(c, n):
    bases = _bases(c, n, n & 0xFF)
    pre = pa.msm_precompute(c.curve_id, bases, 11, device_window=20)
    assert pre.window == 20
    opre = ol.MsmPrecomputation(c.curve_id, bases, 13, threads=THREADS)
    vecs = _vectors(c, n, n & 0xFFFF)
    names = list(vecs)
    expected = {name: opre.execute(vecs[name], parallel=True, threads=THREADS) for name in names}
    for name in names:
        exp, ez = expected[name]
        got, gz = pa.msm_execute_parallel(pre, vecs[name])
        assert gz == ez and np.array_equal(got, exp), name
    # all of them in ONE batched call (shared reduction over nine slots, a workspace each) and once more (workspaces reused)
    stack = np.stack([vecs[k] for k in names])
    for _ in range(2):
        bxy, bz = pa.msm_execute_batch(pre, stack)
        for k, name in enumerate(names):
            exp, ez = expected[name]
            assert int(bz[k]) == ez and np.array_equal(bxy[k], exp), ("batch", name)
    pre.free()


def test_window20_sub_ranges_of_the_generators():
    """plk_msm_execute_parts_dev over a 20-bit-window context: vector b covers generators first[b] .. first[b] + count[b] - 1 only (a rank's
    share of a sharded commitment); entry ids start at first[b], the tiles are those of count[b] scalars."""
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    c = br.TWEEDLEDEE
    n = (1 << 17) + 513
    dev.init(0)
    bases = _bases(c, n, 3)
    s = synth.rand_field(c.scalar.field_id, 0x6F7001, n)
    s[1000:3000] = mont_arr(c.scalar, [1])[0]
    db = dev.to_device(bases.reshape(n, 2, -1))
    pre = dev.msm_precompute_dev(c.curve_id, db, device_window=20)
    parts = [(0, n), (0, 70001), (70001, n - 70001), (12345, 1 << 16), (n - 1025, 1025), (5, 0)]
    oxy, oz = dev.msm_execute_parts_dev(pre, [(f, dev.to_device(np.ascontiguousarray(s[f:f + cnt]).reshape(cnt, 4))) for f, cnt in parts])
    got, gz = dev.to_host(oxy), oz.cpu().numpy()
    opre = ol.MsmPrecomputation(c.curve_id, bases, 13, threads=THREADS)
    for k, (f, cnt) in enumerate(parts):
        v = np.zeros_like(s)
        v[f:f + cnt] = s[f:f + cnt]
        exp, ez = opre.execute(v, parallel=True, threads=THREADS)
        assert int(gz[k]) == ez and np.array_equal(got[k].reshape(exp.shape), exp), (k, f, cnt)


@pytest.mark.parametrize("n,window", [((1 << 16) + 37, 20), (6000, 13), (3000, 9), (1 << 16, 16)], ids=["2p16+37_w20", "6000_w13", "3000_w9_one_level", "2p16_w16"])
def test_bucket_ranges_add_up_to_the_msm(n, window):
    """plk_msm_execute_parts_buckets_dev: the part-th of `parts` ranges of the coarse bucket bins per vector (a rank's BUCKET share of a sharded
    vector).  The shares of all the ranks add up to the vector's MSM (the oracle's), for 2, 3 and 8 ranks, through both orderings (the
    tile-major one at window 20 over >= 2^16 scalars, round 5's kernels elsewhere, the one-level form included), with a generator sub-range
    and a whole vector in the same batched call."""
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    c = br.TWEEDLEDEE
    dev.init(0)
    bases = _bases(c, n, 11)
    s = synth.rand_field(c.scalar.field_id, 0x6F8000 + n, n)
    s[10:200] = mont_arr(c.scalar, [1])[0]
    s[200:300] = 0
    opre = ol.MsmPrecomputation(c.curve_id, bases, 13, threads=THREADS)
    exp, ez = opre.execute(s, parallel=True, threads=THREADS)
    sub = np.zeros_like(s)
    sub[n // 3: n // 2] = s[n // 3: n // 2]
    exp_sub, ez_sub = opre.execute(sub, parallel=True, threads=THREADS)
    db, ds = dev.to_device(bases.reshape(n, 2, -1)), dev.to_device(s)
    pre = dev.msm_precompute_dev(c.curve_id, db, device_window=window)
    for world in (2, 3, 8):
        # every rank's call: a whole vector, its bucket share of the sharded one, its bucket share of a generator sub-range
        pts, zs, pts_sub, zs_sub = [], [], [], []
        for r in range(world):
            oxy, oz = dev.msm_execute_parts_dev(pre, [(0, ds), (0, ds), (n // 3, ds[n // 3: n // 2].contiguous())], buckets=[(0, 1), (r, world), (r, world)])
            got, gz = dev.to_host(oxy), oz.cpu().numpy()
            assert int(gz[0]) == ez and np.array_equal(got[0].reshape(exp.shape), exp), (world, r, "whole vector beside the shares")
            pts.append(got[1]); zs.append(int(gz[1])); pts_sub.append(got[2]); zs_sub.append(int(gz[2]))
        tot, tz = pa.curve_sum_affine(c.curve_id, np.stack(pts), np.array(zs, dtype=np.uint8))
        assert tz == ez and np.array_equal(tot.reshape(exp.shape), exp), (world, "bucket shares do not add up")
        tot, tz = pa.curve_sum_affine(c.curve_id, np.stack(pts_sub), np.array(zs_sub, dtype=np.uint8))
        assert tz == ez_sub and np.array_equal(tot.reshape(exp_sub.shape), exp_sub), (world, "bucket shares of a generator sub-range")
    with pytest.raises(Exception):
        dev.msm_execute_parts_dev(pre, [(0, ds)], buckets=[(3, 3)])


@pytest.mark.parametrize("c,n,window", [(br.TWEEDLEDEE, 5000, 0), (br.TWEEDLEDEE, (1 << 16) + 37, 20), (br.BLS12_377, 3000, 13), (br.TWEEDLEDUM, 300, 0)],
                         ids=["Tweedledee_5000", "Tweedledee_2p16+37_w20", "Bls12377_3000_w13", "Tweedledum_300_comb"])
def test_projective_result_is_the_same_point(c, n, window):
    """plk_msm_execute_projective[_dev]: msm_execute_parallel's own return type (curve_msm.rs:102-157 returns a ProjectivePoint, not
    normalised).  Whatever representative comes back, ProjectivePoint::to_affine (curve.rs:206-214: x / z, y / z, on Python integers here)
    is the affine point of the affine entry points = the oracle's; the all-zero vector gives ProjectivePoint::ZERO; a batch on the device."""
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    dev.init(0)
    f = c.base
    bases = _bases(c, n, 21)
    s = synth.rand_field(c.scalar.field_id, 0x6F9000 + n, n)
    pre = pa.msm_precompute(c.curve_id, bases, 11, device_window=window)
    axy, az = pa.msm_execute_parallel(pre, s)
    exp, ez = ol.MsmPrecomputation(c.curve_id, bases, 8, threads=THREADS).execute(s, parallel=True, threads=THREADS)
    assert az == ez and np.array_equal(axy, exp)

    def to_affine(xyz):
        x, y, z = (f.from_mont(synth.to_int(xyz[k])) for k in range(3))
        zi = pow(z, -1, f.p)
        return x * zi % f.p, y * zi % f.p

    want = (f.from_mont(synth.to_int(axy[0])), f.from_mont(synth.to_int(axy[1])))
    xyz, z = pa.msm_execute_parallel_projective(pre, s)
    assert z == 0 and to_affine(xyz) == want
    xyz0, z0 = pa.msm_execute_parallel_projective(pre, np.zeros_like(s))
    assert z0 == 1 and not xyz0.any()
    pre.free()
    # device-resident, three vectors in one call (one of them all zero)
    db = dev.to_device(bases.reshape(n, 2, -1))
    dpre = dev.msm_precompute_dev(c.curve_id, db, device_window=window)
    vecs = np.stack([s, np.zeros_like(s), np.roll(s, 3, axis=0)])
    oxyz, oz = dev.msm_execute_dev(dpre, dev.to_device(vecs), projective=True)
    oxy, oz2 = dev.msm_execute_dev(dpre, dev.to_device(vecs))
    got, gz, aff, afz = dev.to_host(oxyz), oz.cpu().numpy(), dev.to_host(oxy), oz2.cpu().numpy()
    assert list(gz) == [0, 1, 0] and list(afz) == [0, 1, 0]
    for k in (0, 2):
        assert to_affine(got[k]) == (f.from_mont(synth.to_int(aff[k][0])), f.from_mont(synth.to_int(aff[k][1]))), k


@pytest.mark.parametrize("c", [br.TWEEDLEDEE, br.TWEEDLEDUM, br.BLS12_377, br.PALLAS, br.VESTA], ids=lambda c: c.name)
@pytest.mark.parametrize("n", [(1 << 14) + 2, 20011], ids=["2p14+2", "20011"])
def test_default_window_at_2p14_generators_matches_oracle(c, n):
    """2^14 <= n < 2^15 takes the 16-bit window since round 6 (msm.hip choose_window: the frozen generators of an inner-product argument are
    2^14 + 2) - one bucket piece per lane, the list-driven assembly: uniform and skewed vectors, one by one and as one batched call, against
    the oracle's msm_execute_parallel on every curve."""
    bases = _bases(c, n, 0x2E)
    pre = pa.msm_precompute(c.curve_id, bases, 11)
    assert pre.window == 16
    opre = ol.MsmPrecomputation(c.curve_id, bases, 11, threads=THREADS)
    vecs = _vectors(c, n, 0x2E14 + n)
    names = ["uniform", "all_equal", "sixteen_distinct", "below_2p20", "half_zero", "ones_minus_ones_random"]
    expected = {name: opre.execute(vecs[name], parallel=True, threads=THREADS) for name in names}
    for name in names:
        exp, ez = expected[name]
        got, gz = pa.msm_execute_parallel(pre, vecs[name])
        assert gz == ez and np.array_equal(got, exp), name
    bxy, bz = pa.msm_execute_batch(pre, np.stack([vecs[k] for k in names[:2]]))   # two vectors: the shape of an IPA round
    for k, name in enumerate(names[:2]):
        exp, ez = expected[name]
        assert int(bz[k]) == ez and np.array_equal(bxy[k], exp), ("batch", name)
    pre.free()
