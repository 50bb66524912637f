"""One-process-per-GPU plumbing for the sharded MSM: contiguous base ranges per
rank, ONE all-gather of the packed partial results, point sum on the device.  Independent NTTs need no
collective.  Backend-agnostic: nccl (= RCCL over xGMI) on the GPUs; gloo in the CPU tests and when several
ranks share one GPU (the payload - a few hundred bytes - is then staged through host memory)."""
import ctypes

import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous, balanced [lo, hi) of rank `rank` among `world` ranks (sizes differ by at most 1)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def round_robin(n_items, rank, world):
    """Independent units (whole transforms: the 9 wire LDEs, the iNTTs of a proof) are dealt out rank by rank - no collective."""
    return list(range(rank, n_items, world))


def _backend():
    return dist.get_backend() if dist.is_initialized() else None


class BatchPlan:
    """How `batch` scalar vectors against the same n generators are dealt out over `world` ranks (include/plonky_hip.h):
    floor(batch / world) WHOLE vectors per rank (vector v belongs to rank v mod world, slot v // world there) - they need no
    exchange beyond handing the result over and keep the window a full-size MSM deserves - and the remaining batch mod world
    vectors SHARDED by contiguous base range (slot whole + j on every rank).  9 wire polynomials on 8 GPUs: one whole vector
    per rank and an eighth of the ninth.  A single MSM on N GPUs: the sharded case alone."""

    def __init__(self, batch, world, rank, n, bucket_shard=False):
        """bucket_shard (round 6): a sharded vector is shared by BUCKET range instead of by base range - every rank holds all the
        generators and reads the whole vector, but keeps only the entries whose bucket falls into its world-th of the coarse bins
        (plk_msm_execute_parts_buckets_dev): an N-th of the additions over an N-th of the buckets at the full-size window."""
        self.bucket_shard = bool(bucket_shard)
        self.batch, self.world, self.rank, self.n = batch, world, rank, n
        self.whole = batch // world
        self.sharded = batch - self.whole * world
        self.slots = self.whole + self.sharded
        self.own = [rank + k * world for k in range(self.whole)]
        self.rem = list(range(self.whole * world, batch))
        self.lo, self.hi = shard_bounds(n, rank, world)
        # whole vectors need every generator on every rank, and so does a bucket range; a job sharded by base range only its own range
        self.full_context = self.whole > 0 or self.bucket_shard
        self.n_local = n if self.full_context else self.hi - self.lo
        self.first = 0 if self.full_context else self.lo

    def local_scalars(self, vectors):
        """vectors: (batch, n, 4) uint64 host array of ALL scalar vectors -> this rank's (slots, n_local, 4) array: its whole
        vectors, then its share of the sharded ones (over the full context: the vector with zeros outside [lo, hi) - a zero
        scalar has no digits, so it costs an MSM nothing)."""
        import numpy as np
        out = np.zeros((self.slots, self.n_local, 4), dtype=np.uint64)
        for k, v in enumerate(self.own):
            out[k] = vectors[v]
        for j, v in enumerate(self.rem):
            if self.bucket_shard:
                out[self.whole + j] = vectors[v]
            elif self.full_context:
                out[self.whole + j, self.lo:self.hi] = vectors[v, self.lo:self.hi]
            else:
                out[self.whole + j] = vectors[v, self.lo:self.hi]
        return out

    def parts(self, local):
        """local: this rank's (slots, n_local, 4) CUDA tensor (local_scalars) -> the (first, scalars) list of
        device.msm_execute_parts_dev: whole vectors over all generators, sharded ones over their base range only (the slice of
        the zero-padded row: the ordering kernels then read hi - lo scalars instead of n mostly zero ones)."""
        out = [(0, local[k]) for k in range(self.whole)]
        for j in range(self.sharded):
            row = local[self.whole + j]
            if self.bucket_shard:
                out.append((0, row))
            else:
                out.append((self.lo, row[self.lo:self.hi]) if self.full_context else (0, row))
        return out

    def buckets(self):
        """the (part, parts) list that goes with parts() under bucket_shard (device.msm_execute_parts_dev(..., buckets=...)), else None"""
        if not self.bucket_shard:
            return None
        return [(0, 1)] * self.whole + [(self.rank, self.world)] * self.sharded

    def pairs_local(self):
        """scalar-point pairs this rank reduces per step (a bucket range: its share of a vector's additions)"""
        if self.bucket_shard:
            return self.whole * self.n + self.sharded * (self.n // self.world)
        return self.whole * self.n + self.sharded * (self.hi - self.lo)


class PartialExchange:
    """The exchange step of the sharded MSM with every buffer allocated once.

    `send` is this rank's record in the layout of plk_msm_partials_bytes (include/plonky_hip.h): `slots` affine points then
    `slots` identity flags.  `out_xy` / `out_zero` are views INTO it, so plk_msm_execute_dev writes its results straight
    into the send buffer of the collective; gather() is the one all_gather_into_tensor; combine() produces the `batch` results
    on the device (plk_msm_combine_partials_dev: whole vectors copied from their owner, sharded ones summed over the ranks).
    No packing, no allocation, no host round trip inside a step (except the host staging gloo needs for device tensors)."""

    def __init__(self, curve, batch, device="cuda", whole_per_rank=0, world=None, rank=None, solo=False):
        """solo: this process works alone even though a process group exists (rank 0 timing the one-GPU form of a problem
        inside a multi-rank run): no collective, the record is copied into its slot like an emulated rank's."""
        from . import lib as _lib
        from .api import _CURVE_LIMBS
        self.curve, self.batch, self.whole = curve, batch, whole_per_rank
        self.L = _CURVE_LIMBS[curve]
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.real_world = 1 if solo else (dist.get_world_size() if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)
        self.slots = whole_per_rank + (batch - whole_per_rank * self.world)
        assert 0 < self.slots <= batch
        self.rec = int(_lib.load().plk_msm_partials_bytes(curve, self.slots))
        assert self.rec > 0 and self.rec % 16 == 0
        dev = torch.device(device)
        self.send = torch.zeros(self.rec, dtype=torch.uint8, device=dev)
        self.recv = torch.zeros(self.world * self.rec, dtype=torch.uint8, device=dev)
        pts = self.slots * 2 * self.L * 8
        self.out_xy = self.send[:pts].view(torch.int64).view(self.slots, 2, self.L)
        self.out_zero = self.send[pts:pts + self.slots]
        self.sum_xy = torch.empty((batch, 2, self.L), dtype=torch.int64, device=dev)
        self.sum_zero = torch.empty((batch,), dtype=torch.uint8, device=dev)
        self.recv.view(self.world, self.rec)[:, pts:pts + self.slots] = 1  # until a record arrives its points are the identity
        self._host = None
        if dev.type == "cuda" and _backend() == "gloo":
            self._host = (torch.zeros(self.rec, dtype=torch.uint8).pin_memory(), torch.zeros(self.world * self.rec, dtype=torch.uint8).pin_memory())

    def gather(self):
        """ONE collective: every rank's record to every rank.  (An emulated rank - world > the real world size - fills its own
        slot of the gathered buffer only: the copy stands in for the collective.)"""
        if self.real_world == 1:
            self.recv[self.rank * self.rec:(self.rank + 1) * self.rec].copy_(self.send)
        elif self._host is not None:
            hs, hr = self._host
            hs.copy_(self.send)  # synchronises the current stream
            dist.all_gather_into_tensor(hr, hs)
            self.recv.copy_(hr, non_blocking=True)
        else:
            dist.all_gather_into_tensor(self.recv, self.send)
        return self.recv

    def combine(self):
        """The `batch` results in vector order -> (sum_xy (batch, 2, L), sum_zero (batch,)) on the device."""
        from . import lib as _lib
        assert self.recv.is_cuda, "the point sum runs on the GPU (there is no CPU path)"
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.load().plk_msm_combine_partials_dev(self.curve, self.world, self.batch, self.whole, ctypes.c_void_p(self.recv.data_ptr()),
                                                            ctypes.c_void_p(self.sum_xy.data_ptr()), ctypes.c_void_p(self.sum_zero.data_ptr()), st))
        return self.sum_xy, self.sum_zero

    def partials(self):
        """Views of the gathered records: (world, slots, 2, L) int64 and (world, slots) uint8."""
        rows = self.recv.view(self.world, self.rec)
        pts = self.slots * 2 * self.L * 8
        xy = rows[:, :pts].contiguous().view(torch.int64).view(self.world, self.slots, 2, self.L)
        return xy, rows[:, pts:pts + self.slots].contiguous()


def ntt_batch_sharded(transform, inputs, n_out, gather=True):
    """A batch of INDEPENDENT transforms over the ranks (plonk_util.rs:169-190: the nine iNTTs / LDEs of a proof):
    transform b belongs to rank b mod world (round_robin); computing it needs no collective.
    inputs: (batch, n, 4) tensor (a rank only touches its own rows); transform(rows (k, n, 4)) -> (k, n_out, 4) runs this rank's
    rows in ONE batched call (device.ntt_dev / device.ntt_padded_dev on the GPUs, the oracle in the CPU test).
    gather=False: the results stay where they were computed - returns (indices, results) of this rank; a downstream stage dealt
    out the same way needs nothing else.  gather=True: ONE all-gather hands every rank all results in batch order, (batch, n_out, 4)
    (equal-sized blocks of ceil(batch / world) rows: a rank with one row fewer sends a zero row that is dropped)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    batch = inputs.shape[0]
    mine = round_robin(batch, rank, world)
    out = transform(inputs[mine]) if mine else None
    if not gather:
        return mine, out
    if world == 1:
        return out
    per = (batch + world - 1) // world
    block = torch.zeros((per, n_out, inputs.shape[2]), dtype=inputs.dtype, device=inputs.device)
    if out is not None:
        block[:len(mine)] = out
    staged = block.is_cuda and _backend() == "gloo"
    src = block.cpu() if staged else block
    recv = torch.empty((world * per, n_out, inputs.shape[2]), dtype=block.dtype, device=src.device)
    dist.all_gather_into_tensor(recv.view(-1), src.view(-1))
    recv = recv.to(inputs.device).view(world, per, n_out, inputs.shape[2])
    res = torch.empty((batch, n_out, inputs.shape[2]), dtype=inputs.dtype, device=inputs.device)
    for r in range(world):
        idx = round_robin(batch, r, world)
        if idx:
            res[idx] = recv[r, :len(idx)]
    return res


def all_gather_points(xy, zero):
    """xy: (batch, 2, L) int64, zero: (batch,) uint8 - this rank's partial results (any device).
    Returns (world, batch, 2, L) and (world, batch) on every rank.  ONE collective: the identity flags travel behind the
    coordinate limbs in the same record.  Allocating convenience form of PartialExchange (CPU tests, one-off calls)."""
    world = dist.get_world_size()
    batch, L = xy.shape[0], xy.shape[2]
    pts = batch * 2 * L * 8
    rec = (pts + batch + 15) & ~15
    send = torch.zeros(rec, dtype=torch.uint8, device=xy.device)
    send[:pts] = xy.contiguous().view(torch.uint8).reshape(-1)
    send[pts:pts + batch] = zero
    staged = xy.is_cuda and _backend() == "gloo"
    src = send.cpu() if staged else send
    recv = torch.empty(world * rec, dtype=torch.uint8, device=src.device)
    dist.all_gather_into_tensor(recv, src)
    rows = recv.to(xy.device).view(world, rec)
    g_xy = rows[:, :pts].contiguous().view(torch.int64).view(world, batch, 2, L)
    return g_xy, rows[:, pts:pts + batch].contiguous()


def msm_sharded(execute_local, combine, scalars_local):
    """execute_local(scalars) -> (xy (batch,2,L), zero (batch,)) partial results of this rank's base range;
    combine(points (world,2,L), zeros (world,)) -> (xy, zero) sums the ranks' partial points.
    Returns the list of global results, one per scalar vector."""
    xy, zero = execute_local(scalars_local)
    g_xy, g_z = all_gather_points(xy, zero)
    out = []
    for b in range(xy.shape[0]):
        out.append(combine(g_xy[:, b], g_z[:, b]))
    return out


def msm_sharded_hip(pre, scalars_local, exchange=None):
    """The sharded MSM over the HIP path: `pre` is this rank's device MsmPrecomputation over its base range
    (plonky_amd.device.msm_precompute_dev), scalars_local the matching slice of every scalar vector ((batch, n_local, 4) int64
    CUDA tensor).  Per-rank partial results written into the exchange record -> one all-gather -> point sums on the device.
    Returns (xy (batch, 2, L) uint64 numpy, zero list) on every rank."""
    from . import device as dev
    batch = scalars_local.shape[0] if scalars_local.dim() == 3 else 1
    ex = exchange if exchange is not None else PartialExchange(pre.curve, batch, scalars_local.device)
    dev.msm_execute_dev(pre, scalars_local, ex.out_xy, ex.out_zero)
    ex.gather()
    xy, z = ex.combine()
    return dev.to_host(xy), [int(v) for v in z.cpu().numpy()]
