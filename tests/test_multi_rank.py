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


def _exchange_worker(rank, world, port, batch, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = parallel.PartialExchange(0, batch, "cpu")  # record layout of plk_msm_partials_bytes; the point sum itself needs the GPU
    ex.out_xy.copy_(torch.arange(batch * 8, dtype=torch.int64).view(batch, 2, 4) + 1000 * rank)
    ex.out_zero.copy_(torch.tensor([(rank + k) % 2 for k in range(batch)], dtype=torch.uint8))
    ex.gather()
    xy, z = ex.partials()
    dist.barrier()
    if rank == 1:
        out_q.put((ex.rec, xy.numpy(), z.numpy()))
    dist.destroy_process_group()


def test_partial_exchange_record_layout_world_size_2_gloo():
    """The preallocated exchange of the sharded MSM: the MSM's output pointers are views into the send record, one
    all-gather, rank-ordered records on every rank (CPU tensors here; -m gpu runs it over the HIP path)."""
    batch, world = 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    rec, xy, z = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert rec == (batch * 64 + batch + 15) // 16 * 16
    for r in range(world):
        assert np.array_equal(xy[r], np.arange(batch * 8).reshape(batch, 2, 4) + 1000 * r)
        assert list(z[r]) == [(r + k) % 2 for k in range(batch)]


def test_batch_plan_parts_cover_every_pair_once():
    """parallel.BatchPlan: whole vectors per rank + the remainder sharded by base range; the (first, scalars) parts a rank hands to
    plk_msm_execute_parts_dev are exactly its whole vectors and its slice of every sharded one - over all ranks every
    (vector, generator) pair appears once, and the slices are views of the rank's local array (no copy)."""
    import numpy as np
    import torch
    from plonky_amd import parallel
    for batch, world, n in ((9, 8, 64), (9, 4, 50), (9, 2, 33), (1, 4, 40), (3, 3, 17), (5, 8, 16)):
        vectors = np.arange(batch * n * 4, dtype=np.uint64).reshape(batch, n, 4) + 1
        seen = np.zeros((batch, n), dtype=np.int64)
        for rank in range(world):
            plan = parallel.BatchPlan(batch, world, rank, n)
            local = torch.from_numpy(plan.local_scalars(vectors).view(np.int64))
            parts = plan.parts(local)
            assert len(parts) == plan.slots
            for k, (first, sc) in enumerate(parts):
                v = plan.own[k] if k < plan.whole else plan.rem[k - plan.whole]
                cnt = sc.shape[0]
                if plan.full_context:
                    assert (first, cnt) == ((0, n) if k < plan.whole else (plan.lo, plan.hi - plan.lo))
                    assert sc.data_ptr() == local[k, first:].data_ptr()       # a view into the rank's array
                    g0 = first
                else:
                    assert first == 0 and cnt == plan.n_local                 # the context itself only holds the rank's base range
                    g0 = plan.lo
                assert np.array_equal(sc.numpy().view(np.uint64), vectors[v, g0:g0 + cnt])
                seen[v, g0:g0 + cnt] += 1
        assert (seen == 1).all(), (batch, world, n)


def test_batch_plan_bucket_shard_hands_every_rank_the_whole_vector():
    """parallel.BatchPlan(bucket_shard=True) (round 6): a sharded vector goes to EVERY rank whole, with the rank's range of the coarse bucket
    bins (plk_msm_execute_parts_buckets_dev); whole vectors are dealt as before with every bucket; the bin ranges of the ranks tile the
    bins exactly as the library cuts them (msm_execute_t: the larger ranges first, like shard_bounds)."""
    import numpy as np
    import torch
    from plonky_amd import parallel
    for batch, world, n in ((9, 8, 64), (9, 4, 50), (1, 4, 40), (5, 3, 17)):
        vectors = np.arange(batch * n * 4, dtype=np.uint64).reshape(batch, n, 4) + 1
        for rank in range(world):
            plan = parallel.BatchPlan(batch, world, rank, n, bucket_shard=True)
            assert plan.full_context and plan.n_local == n and plan.first == 0
            local = torch.from_numpy(plan.local_scalars(vectors).view(np.int64))
            parts, buckets = plan.parts(local), plan.buckets()
            assert len(parts) == len(buckets) == plan.slots
            for k, ((first, sc), (bp, bn)) in enumerate(zip(parts, buckets)):
                v = plan.own[k] if k < plan.whole else plan.rem[k - plan.whole]
                assert first == 0 and np.array_equal(sc.numpy().view(np.uint64), vectors[v])
                assert (bp, bn) == ((0, 1) if k < plan.whole else (rank, world))
            assert plan.pairs_local() == plan.whole * n + plan.sharded * (n // world)
        assert parallel.BatchPlan(batch, world, 0, n).buckets() is None
    # the ranks' bin ranges: contiguous, disjoint, covering - the arithmetic of msm_execute_t
    for nbins, world in ((512, 8), (512, 3), (64, 5)):
        cuts = [parallel.shard_bounds(nbins, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == nbins and all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))


def test_bench_without_gpus_fails_loudly():
    """`python bench.py --gpus 2` as the driver types it: no assert on WORLD_SIZE - it spawns its ranks itself, and on a box
    without GPUs it exits non-zero naming the missing devices (there is no CPU path)."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 3 and "needs 2 GPU(s), 0 visible" in out.stderr


def test_bench_single_process_child_never_raises():
    """bench.py runs the in-library multi-GPU case of the driver's N > 1 line in a child process with a time limit: a child that
    cannot run (no GPU here), one that is cut off by the limit - both come back as {"error": ...}, neither as an exception."""
    import importlib.util
    if torch.cuda.is_available():
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "2"  # what a launcher leaves behind must not reach the child
    try:
        r = bench.single_process_child(2, 10, 4, timeout_s=300)
        assert "error" in r and "no JSON line" in r["error"] and "there is no CPU path" in r["error"], r
        r = bench.single_process_child(2, 10, 4, timeout_s=0.05)
        assert "error" in r and "child stopped" in r["error"], r
    finally:
        del os.environ["RANK"], os.environ["WORLD_SIZE"]


def _ntt_worker(rank, world, port, batch, log_n, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1 << log_n
    polys = torch.from_numpy(np.stack([synth.rand_field(0, 0xF70020 + v, n) for v in range(batch)]).view(np.int64))
    pre = ol.FftPrecomputation(0, n)
    calls = []

    def transform(rows):  # the oracle stands in for device.ntt_dev: one batched call per rank
        calls.append(rows.shape[0])
        return torch.from_numpy(np.stack([pre.fft_with_precomputation_power_of_2(r.numpy().view(np.uint64)) for r in rows]).view(np.int64))

    gathered = parallel.ntt_batch_sharded(transform, polys, n, gather=True)
    mine, local = parallel.ntt_batch_sharded(transform, polys, n, gather=False)
    dist.barrier()
    out_q.put((rank, calls, mine, local.numpy().view(np.uint64), gathered.numpy().view(np.uint64) if rank == 1 else None))
    dist.destroy_process_group()


def test_ntt_batch_sharded_world_size_2_gloo():
    """The transform batch of a proof (plonk_util.rs:169-190) dealt out over two ranks: each rank transforms its own rows in one
    batched call, one all-gather returns all nine in batch order on every rank; without the gather the rows stay with their rank."""
    batch, world, log_n = 9, 2, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ntt_worker, args=(r, world, port, batch, log_n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 1 << log_n
    pre = ol.FftPrecomputation(0, n)
    exp = np.stack([pre.fft_with_precomputation_power_of_2(synth.rand_field(0, 0xF70020 + v, n)) for v in range(batch)])
    for rank, calls, mine, local, gathered in got:
        assert mine == list(range(rank, batch, world)) and calls == [len(mine), len(mine)]
        assert np.array_equal(local, exp[mine])
        if gathered is not None:
            assert np.array_equal(gathered, exp)


def test_library_plan_equals_batch_plan():
    """plk_multi_plan - the plan the in-library fan-out over a device group follows (multi.hip, pure arithmetic: runs without a GPU) -
    against parallel.BatchPlan, the plan of the one-process-per-GPU form: the same vectors whole, the same base ranges, and every
    (vector, generator) pair exactly once over the devices."""
    import ctypes
    from plonky_amd import lib
    L = lib.load()
    for batch, world, n in ((9, 8, 1 << 20), (9, 4, 50), (9, 2, 33), (1, 4, 40), (3, 3, 17), (5, 8, 16), (16, 3, 1000), (2, 7, 7), (0, 4, 10)):
        seen = np.zeros((max(batch, 1), n), dtype=np.int64) if n <= 4096 else None
        pairs = 0
        for d in range(world):
            slots = ctypes.c_uint(0)
            vec = (ctypes.c_uint * max(batch, 1))()
            first = (ctypes.c_uint64 * max(batch, 1))()
            count = (ctypes.c_uint64 * max(batch, 1))()
            assert L.plk_multi_plan(world, batch, n, d, ctypes.byref(slots), vec, first, count) == 0
            plan = parallel.BatchPlan(batch, world, d, n)
            assert slots.value == plan.slots
            for s in range(slots.value):
                if s < plan.whole:
                    assert (vec[s], first[s], count[s]) == (plan.own[s], 0, n)
                else:
                    assert (vec[s], first[s], count[s]) == (plan.rem[s - plan.whole], plan.lo, plan.hi - plan.lo)
                pairs += count[s]
                if seen is not None:
                    seen[vec[s], first[s]:first[s] + count[s]] += 1
        assert pairs == batch * n
        if seen is not None and batch:
            assert (seen == 1).all(), (batch, world, n)
    slots = ctypes.c_uint(0)
    assert L.plk_multi_plan(0, 1, 1, 0, ctypes.byref(slots), None, None, None) != 0      # no devices
    assert L.plk_multi_plan(4, 9, 100, 4, ctypes.byref(slots), None, None, None) != 0    # device out of range
    assert L.plk_multi_plan(4, 9, 100, 1, ctypes.byref(slots), None, None, None) == 0 and slots.value == 3  # the count alone
