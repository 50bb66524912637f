"""CPU, world size 2, gloo: the N > 1 path of bench.py / plonky_amd.parallel - base-range sharding,
the all-gather of the partial points, the final point sum - with the oracle standing in for the
per-rank device MSM (there is no GPU here).  The sharded result must equal the unsharded MSM."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import bigint_ref as br
from oracle import oracle_lib as ol
from plonky_amd import parallel, synth
from tests.util import array_to_ints


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(n, batch):
    c = br.TWEEDLEDEE
    G = (c.gx, c.gy)
    D = br.ec_mul(c, 31337, G)
    g0 = np.array([c.base.mont_limbs(G[0]), c.base.mont_limbs(G[1])], dtype=np.uint64)
    dd = np.array([c.base.mont_limbs(D[0]), c.base.mont_limbs(D[1])], dtype=np.uint64)
    bases = ol.gen_bases(0, n, g0, dd)
    scalars = np.stack([synth.rand_field(1, 0x350920 + k, n) for k in range(batch)])
    return c, bases, scalars


def _worker(rank, world, port, n, batch, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c, bases, scalars = _inputs(n, batch)
    lo, hi = parallel.shard_bounds(n, rank, world)
    pre = ol.MsmPrecomputation(0, bases[lo:hi], 8)

    def execute_local(sv):
        xs, zs = [], []
        for b in range(sv.shape[0]):
            xy, z = pre.execute(sv[b])
            xs.append(xy)
            zs.append(z)
        return torch.from_numpy(np.stack(xs).view(np.int64)), torch.tensor(zs, dtype=torch.uint8)

    def combine(points, zeros):
        acc = None
        for p, z in zip(points.numpy().view(np.uint64), zeros.numpy()):
            if z:
                continue
            acc = br.ec_add(c, acc, tuple(c.base.from_mont(v) for v in array_to_ints(p)))
        return acc

    res = parallel.msm_sharded(execute_local, combine, scalars[:, lo:hi])
    dist.barrier()
    if rank == 0:
        out_q.put(res)
    dist.destroy_process_group()


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_sharded_msm_world_size_2_gloo():
    n, batch, world = 300, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    c, bases, scalars = _inputs(n, batch)
    full = ol.MsmPrecomputation(0, bases, 8)
    for b in range(batch):
        xy, z = full.execute(scalars[b])
        assert z == 0
        assert res[b] == tuple(c.base.from_mont(v) for v in array_to_ints(xy))


def test_round_robin_covers_every_unit_once():
    for n in (0, 1, 9, 19):
        for world in (1, 2, 8):
            got = sorted(i for r in range(world) for i in parallel.round_robin(n, r, world))
            assert got == list(range(n))
