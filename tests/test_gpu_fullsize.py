"""BASELINE.json's full-size configurations through the C ABI, checked by the closed form of the arithmetic-progression
generators (sum s_i (G0 + i D) = [sum s_i] G0 + [sum i s_i] D, two scalar multiplications on Python integers):
config 4 (the 9-wire commit batch at 2^20), config 5 (a 2^22 BLS12-377 G1 MSM), and the sharded MSM of the multi-GPU path over
the HIP code with the real RCCL backend at world size 1."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bigint_ref as br
from plonky_amd import synth
from plonky_amd.selfcheck import closed_form_msm
from tests.test_oracle_kats import from_mont_arr
from tests.util import limbs_to_int


def _pt(c, P):
    return np.array([c.base.mont_limbs(P[0]), c.base.mont_limbs(P[1])], dtype=np.uint64)


def test_commit9_2p20_closed_form():
    """BASELINE config 4: nine 2^20 scalar vectors against the same 2^20 generators, one batched call; every result is
    checked against its own closed form."""
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    dev.init(0)
    c = br.TWEEDLEDEE
    n = 1 << 20
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 0x350920, G)
    bases = dev.gen_bases_dev(0, n, _pt(c, G), _pt(c, D))
    pre = dev.msm_precompute_dev(0, bases)
    sv = np.stack([synth.rand_field(1, 0x350920 + k, n) for k in range(9)])
    oxy, oz = dev.msm_execute_dev(pre, dev.to_device(sv))
    torch.cuda.synchronize()
    got = dev.to_host(oxy)
    assert not oz.cpu().numpy().any()
    for k in range(9):
        assert tuple(from_mont_arr(c.base, got[k])) == closed_form_msm(0, sv[k], G, D), k
    pre.free()


def test_bls12_377_2p22_closed_form():
    """BASELINE config 5 (BLS12-377 G1, the curve the reference actually has): one 2^22-pair MSM."""
    torch = pytest.importorskip("torch")
    from plonky_amd import device as dev
    dev.init(0)
    c = br.BLS12_377
    n = 1 << 22
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 0x350022, G)
    bases = dev.gen_bases_dev(2, n, _pt(c, G), _pt(c, D))
    pre = dev.msm_precompute_dev(2, bases)
    s = synth.rand_field(2, 0x350022, n)
    oxy, oz = dev.msm_execute_dev(pre, dev.to_device(s))
    torch.cuda.synchronize()
    assert int(oz.cpu()[0]) == 0
    assert tuple(from_mont_arr(c.base, dev.to_host(oxy).reshape(2, 6))) == closed_form_msm(2, s, G, D)
    pre.free()


def test_sharded_msm_hip_nccl_world_size_1():
    """plonky_amd.parallel.msm_sharded_hip over the HIP path with backend nccl (= RCCL): base range of this rank, one packed
    all-gather, plk_curve_sum_affine.  One rank here; the CPU suite runs the same plumbing at world size 2 on gloo."""
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    from plonky_amd import device as dev, parallel
    dev.init(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        c = br.TWEEDLEDEE
        n = 1 << 14
        G = (c.gx, c.gy)
        D = br.ec_mul(c, 4242, G)
        lo, hi = parallel.shard_bounds(n, dist.get_rank(), dist.get_world_size())
        bases = dev.gen_bases_dev(0, hi - lo, _pt(c, G), _pt(c, D), first=lo)
        pre = dev.msm_precompute_dev(0, bases)
        sv = np.stack([synth.rand_field(1, 0x77 + k, n) for k in range(3)])
        xy, zero = parallel.msm_sharded_hip(pre, dev.to_device(sv[:, lo:hi]))
        for k in range(3):
            assert zero[k] == 0 and tuple(from_mont_arr(c.base, xy[k])) == closed_form_msm(0, sv[k], G, D), k
    finally:
        dist.destroy_process_group()
