"""GPU parity: the MSM's lane and reduction geometries against the oracle.

The entries a lane of k_msm_accumulate walks (msm_configure_lanes: 8 .. 96, from the size) and the group size of the row / column sums
(the tail geometry: 2^2 or 2^3 buckets per partial) follow from the problem size, so the default suite sees each size with ONE geometry.
PLK_MSM_SLICE / PLK_MSM_GLOG (read when a context is built) choose them; the result is sum_i s_i B_i (curve_msm.rs:102-180) whatever
the partition of the sorted entry list into lanes and of a line of buckets into groups: chunks of 2 entries (almost every piece a head
piece, heavy-bucket lists everywhere), chunks far longer than a bucket, groups from one bucket to a whole line - bit-identical to the
oracle, with edge scalars, a duplicated generator, a batch of three vectors, one- and two-level reductions and the table-free mode.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import plonky_amd as pa
from plonky_amd import synth
from oracle import bigint_ref as br
from oracle import oracle_lib as ol
from tests.test_oracle_kats import mont_arr

KNOBS = ("PLK_MSM_SLICE", "PLK_MSM_GLOG")


@pytest.fixture(autouse=True)
def _restore_knobs():
    yield
    for k in KNOBS:
        os.environ.pop(k, None)


def _case(c, n, seed):
    G = (c.gx, c.gy)
    pt = lambda P: np.array([c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])], dtype=np.uint64)
    bases = ol.gen_bases(c.curve_id, n, pt(G), pt(br.ec_mul(c, 0xD15EA5E + seed, G)))
    scalars = synth.rand_field(c.scalar.field_id, 0x6E0000 + seed, n)
    scalars[0] = mont_arr(c.scalar, [0])[0]
    scalars[1] = mont_arr(c.scalar, [1])[0]
    scalars[2] = mont_arr(c.scalar, [c.scalar.p - 1])[0]
    bases[5] = bases[4]
    scalars[5] = scalars[4]
    scalars[100:140] = scalars[99]  # forty equal scalars: every window's digit of theirs lands in one bucket (a heavy bucket at small windows)
    return bases, scalars


@pytest.mark.parametrize("c", [br.TWEEDLEDEE, br.BLS12_377], ids=lambda c: c.name)
@pytest.mark.parametrize("win", [9, 13, 16])
def test_lane_and_group_geometries_match_oracle(c, win):
    n = 6000
    bases, scalars = _case(c, n, win)
    opre = ol.MsmPrecomputation(c.curve_id, bases, 8, threads=8)
    expected, ez = opre.execute(scalars, parallel=True, threads=8)
    settings = [{}] + [{"PLK_MSM_SLICE": str(s)} for s in (2, 3, 5, 8, 13, 31, 64, 96, 257, 4096)]
    if win >= 13:  # two-level reduction: the row / column sums exist
        settings += [{"PLK_MSM_GLOG": str(g)} for g in (0, 1, 2, 3, 4, 5, 6)]
        settings += [{"PLK_MSM_SLICE": "3", "PLK_MSM_GLOG": "1"}, {"PLK_MSM_SLICE": "96", "PLK_MSM_GLOG": "5"}]
    for st in settings:
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(st)
        pre = pa.msm_precompute(c.curve_id, bases, 8, device_window=win)
        got, gz = pa.msm_execute_parallel(pre, scalars)
        assert gz == ez and np.array_equal(got, expected), st
        got, gz = pa.msm_execute_parallel(pre, scalars)  # workspaces reused
        assert gz == ez and np.array_equal(got, expected), st
    # the table-free mode under the extreme lane settings
    for st in ({"PLK_MSM_SLICE": "2"}, {"PLK_MSM_SLICE": "96", "PLK_MSM_GLOG": "1"}):
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(st)
        pre_tf = pa.msm_precompute(c.curve_id, bases, 8, device_window=win, table_free=True)
        got, gz = pa.msm_execute_parallel(pre_tf, scalars)
        assert gz == ez and np.array_equal(got, expected), ("table-free", st)


def test_geometries_on_a_batch_2p16():
    """Three vectors in one call over 2^16 generators at window 16 (the window the product picks there), lanes of 8 / 24 / 96 entries:
    every vector against the oracle."""
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    c = br.TWEEDLEDEE
    n = 1 << 16
    dev.init(0)
    bases, s0 = _case(c, n, 77)
    vecs = [s0, synth.rand_field(c.scalar.field_id, 0x6E1001, n), synth.rand_field(c.scalar.field_id, 0x6E1002, n)]
    vecs[2][: n // 2] = 0  # a half-empty vector
    opre = ol.MsmPrecomputation(c.curve_id, bases, 8, threads=8)
    exp = [opre.execute(v, parallel=True, threads=8) for v in vecs]
    db, ds = dev.to_device(bases.reshape(n, 2, -1)), dev.to_device(np.stack(vecs))
    for sl in (None, "8", "24", "96"):
        os.environ.pop("PLK_MSM_SLICE", None)
        if sl:
            os.environ["PLK_MSM_SLICE"] = sl
        pre = dev.msm_precompute_dev(c.curve_id, db, device_window=16)
        oxy, oz = dev.msm_execute_dev(pre, ds)
        got, gz = dev.to_host(oxy), oz.cpu().numpy()
        for k, (e, ezk) in enumerate(exp):
            assert int(gz[k]) == ezk and np.array_equal(got[k].reshape(e.shape), e), (sl, k)
