"""One-process-per-GPU plumbing for the sharded MSM: contiguous base ranges per
rank, one all-gather of the affine partial results, local point sum.  Independent NTTs need no
collective.  Backend-agnostic (nccl = RCCL on the GPUs, gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous, balanced [lo, hi) of rank `rank` among `world` ranks (sizes differ by at most 1)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_points(xy, zero):
    """xy: (batch, 2, L) int64, zero: (batch,) uint8 - this rank's partial results.
    Returns (world, batch, 2, L) and (world, batch) on every rank."""
    world = dist.get_world_size()
    # output = the ranks' tensors concatenated along dim 0 (the layout both nccl and gloo accept)
    g_xy = torch.empty((world * xy.shape[0],) + tuple(xy.shape[1:]), dtype=xy.dtype, device=xy.device)
    g_z = torch.empty((world * zero.shape[0],) + tuple(zero.shape[1:]), dtype=zero.dtype, device=zero.device)
    dist.all_gather_into_tensor(g_xy, xy.contiguous())
    dist.all_gather_into_tensor(g_z, zero.contiguous())
    return g_xy.view((world,) + tuple(xy.shape)), g_z.view((world,) + tuple(zero.shape))


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
