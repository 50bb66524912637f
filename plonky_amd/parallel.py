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
    Returns (world, batch, 2, L) and (world, batch) on every rank.  ONE collective: the identity flag travels as an extra
    64-bit word behind the 2L coordinate limbs of its point (the payload is a few hundred bytes: latency, not bandwidth)."""
    world = dist.get_world_size()
    batch = xy.shape[0]
    words = xy.shape[1] * xy.shape[2]
    packed = torch.empty((batch, words + 1), dtype=torch.int64, device=xy.device)
    packed[:, :words] = xy.reshape(batch, words)
    packed[:, words] = zero.to(torch.int64)
    gathered = torch.empty((world * batch, words + 1), dtype=torch.int64, device=xy.device)
    dist.all_gather_into_tensor(gathered, packed)
    gathered = gathered.view(world, batch, words + 1)
    g_xy = gathered[:, :, :words].reshape((world,) + tuple(xy.shape)).contiguous()
    g_z = gathered[:, :, words].to(torch.uint8).contiguous()
    return g_xy, g_z


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


def msm_sharded_hip(pre, scalars_local):
    """The sharded MSM over the HIP path: `pre` is this rank's device MsmPrecomputation over its base range
    (plonky_amd.device.msm_precompute_dev), scalars_local the matching slice of every scalar vector ((batch, n_local, 4) int64
    CUDA tensor).  Per-rank partial results -> one all-gather -> plk_curve_sum_affine per vector.
    Returns (xy (batch, 2, L) uint64 numpy, zero list) on every rank."""
    import numpy as np
    from . import api, device as dev

    def execute_local(sv):
        xy, z = dev.msm_execute_dev(pre, sv)
        return xy, z

    def combine(points, zeros):
        return api.curve_sum_affine(pre.curve, dev.to_host(points), zeros.cpu().numpy())

    res = msm_sharded(execute_local, combine, scalars_local)
    return np.stack([r[0] for r in res]), [r[1] for r in res]


def round_robin(n_items, rank, world):
    """Independent units (whole transforms: the 9 wire LDEs, the iNTTs of a proof) are dealt out rank by rank - no collective."""
    return list(range(rank, n_items, world))
