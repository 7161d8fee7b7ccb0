"""Multi-GPU plumbing for the registration path: one process per GPU, scan points sharded contiguously, map replicated,
ONE all-reduce (sum, float64) of the packed normal equations of the whole batch per ICP iteration.

Layout of the packed record per scan (ELM_PACKED_SUMS = 32 doubles, written by K1/K2 and consumed by K2):
  [0..20]  upper triangle of JtJ, row-major (i <= j)      [21..26] Jtr
  [27]     sum of residual norms (|r.n| for GICP)           [28]     number of correspondences
  [29..31] work counters (candidates, occupied voxels, tested + 2^40 * fall-back workgroups)
"""
import numpy as np

PACKED_SUMS = 32
TRI = [(i, j) for i in range(6) for j in range(i, 6)]


def shard_bounds(n, rank, world_size):
    """Contiguous shard [lo, hi) of an n-point scan owned by `rank` (every point belongs to exactly one rank)."""
    return n * rank // world_size, n * (rank + 1) // world_size


def pack_sums(JTJ, JTr, residual_sum, n_corr, counters=(0.0, 0.0, 0.0)):
    v = np.zeros(PACKED_SUMS)
    for k, (i, j) in enumerate(TRI):
        v[k] = JTJ[i, j]
    v[21:27] = JTr
    v[27] = residual_sum
    v[28] = n_corr
    v[29:32] = counters
    return v


def unpack_sums(v):
    H = np.zeros((6, 6))
    for k, (i, j) in enumerate(TRI):
        H[i, j] = H[j, i] = v[k]
    return H, np.array(v[21:27]), float(v[27]), float(v[28])


def init_rccl(ctx, rank=None, world_size=None):
    """Create the context's RCCL communicator; the unique id travels over an already initialised torch.distributed
    process group (any backend)."""
    import torch.distributed as dist
    from .registration import Context
    rank = dist.get_rank() if rank is None else rank
    world_size = dist.get_world_size() if world_size is None else world_size
    ids = [Context.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    ctx.comm_init(rank, world_size, ids[0])
