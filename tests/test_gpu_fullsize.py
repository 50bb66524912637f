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


# ---- N > 1 over the HIP path: two ranks share the one GPU of the box (gloo, the ~200-byte payload staged through the host) ----
def _rank_hip(rank, world, port, n, batch, q):
    import torch
    import torch.distributed as dist
    from plonky_amd import device as dev, parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev.init(0)
        c = br.TWEEDLEDEE
        G = (c.gx, c.gy)
        D = br.ec_mul(c, 4242, G)
        lo, hi = parallel.shard_bounds(n, rank, world)
        bases = dev.gen_bases_dev(0, hi - lo, _pt(c, G), _pt(c, D), first=lo)
        pre = dev.msm_precompute_dev(0, bases)
        sv = np.stack([synth.rand_field(1, 0x77 + k, n) for k in range(batch)])
        ex = parallel.PartialExchange(0, batch, "cuda")
        for _ in range(2):  # the buffers are reused from step to step
            xy, zero = parallel.msm_sharded_hip(pre, dev.to_device(np.ascontiguousarray(sv[:, lo:hi])), exchange=ex)
        part_xy, part_z = ex.partials()
        # the same three vectors through the batch plan: one whole vector per rank, the third sharded by base range
        plan = parallel.BatchPlan(batch, world, rank, n)
        bases_all = dev.gen_bases_dev(0, n, _pt(c, G), _pt(c, D))
        pre_all = dev.msm_precompute_dev(0, bases_all)
        ex2 = parallel.PartialExchange(0, batch, "cuda", whole_per_rank=plan.whole)
        dev.msm_execute_dev(pre_all, dev.to_device(plan.local_scalars(sv)), ex2.out_xy, ex2.out_zero)
        ex2.gather()
        hxy, hz = ex2.combine()
        q.put((rank, xy, zero, dev.to_host(part_xy), part_z.cpu().numpy(), dev.to_host(hxy), hz.cpu().numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_msm_hip_world_size_2_one_gpu():
    """parallel.msm_sharded_hip at world size 2: both ranks run the HIP kernels on GPU 0 over their own base range, ONE
    all-gather of the packed records, plk_msm_combine_partials_dev; every rank must hold the closed form of the GLOBAL MSM."""
    pytest.importorskip("torch")
    import socket
    import torch.multiprocessing as mp
    n, batch, world = 1 << 14, 3, 2
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_hip, args=(r, world, port, n, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    c = br.TWEEDLEDEE
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 4242, G)
    sv = np.stack([synth.rand_field(1, 0x77 + k, n) for k in range(batch)])
    from plonky_amd import parallel
    for rank, xy, zero, part_xy, part_z, hxy, hz in res:
        assert part_xy.shape == (world, batch, 2, 4) and not part_z.any()
        for k in range(batch):
            assert zero[k] == 0 and tuple(from_mont_arr(c.base, xy[k])) == closed_form_msm(0, sv[k], G, D), (rank, k)
            # whole vectors handed over by their owner + the sharded remainder: the same global results
            assert int(hz[k]) == 0 and np.array_equal(hxy[k], xy[k]), (rank, k)
            for r in range(world):  # the gathered records hold every rank's partial result, in rank order
                lo, hi = parallel.shard_bounds(n, r, world)
                assert tuple(from_mont_arr(c.base, part_xy[r, k])) == closed_form_msm(0, sv[k, lo:hi], G, D, first=lo), (rank, k, r)


@pytest.mark.parametrize("flags", [["--workload", "commit9", "--log-n", "15"],
                                   ["--workload", "msm", "--shard", "--curve", "bls12_377", "--log-n", "14"],
                                   ["--workload", "both", "--log-n", "14"]])
def test_bench_spawns_its_ranks(flags):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: spawns two ranks (here both on GPU 0), prints ONE JSON
    line, every self-check (per-rank closed form, device point sum against the host sum, global closed form) true."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--same-device", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"] + flags, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["checks"] and all(r["checks"].values()), r["checks"]
    assert r["scaling"] == ("weak" if "both" in flags else "strong")
    if r["scaling"] == "strong":
        assert r["checks"]["msm_global_sum_closed_form"]


def test_bench_emulated_rank():
    """--emulate-rank r/N: one rank's shard of the strong-scaling problem alone (the per-rank time behind DESIGN.md section 6)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "commit9", "--log-n", "16", "--emulate-rank", "3/8", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    e = r["emulated_rank"]
    # nine vectors on eight ranks: one whole vector and an eighth of the ninth per rank (parallel.BatchPlan)
    assert e["rank"] == 3 and e["whole_vectors"] == 1 and e["sharded_vectors"] == 1 and e["pairs_local"] == (1 << 16) + (1 << 16) // 8
    assert all(r["checks"].values())
