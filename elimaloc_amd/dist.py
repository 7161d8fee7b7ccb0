"""Multi-GPU plumbing for the registration path: one process per GPU, scan points sharded contiguously, map replicated,
ONE all-reduce (sum, float64) of the packed normal equations of the whole batch per ICP iteration.

Layout of the packed record per scan (ELM_PACKED_SUMS = 32 doubles, written by K1/K2 and consumed by K2):
  [0..20]  upper triangle of JtJ, row-major (i <= j)      [21..26] Jtr
  [27]     sum of residual norms (|r.n| for GICP)           [28]     number of correspondences
  [29..31] work counters (candidates, occupied voxels, tested + 2^40 * fall-back workgroups) when the instrumented kernels run;
           in production (counters off) and under an exchange: the RANK-AGREEMENT check (1, id, id^2), id = 16 (registration + 1) +
           (iteration & 15) of the slot -- after the sum all-reduce every rank verifies sum(id) == n id and sum(id^2) == n id^2 (exact
           integers in doubles): all ranks iterate the same registration in the slot, else the call ends with ELM_ERR_COMM
           (k_solve modes 1 / 2, RegParams::rank_check)
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


def rank_check_id(registration, iteration):
    """id of a slot's (registration, iteration) as the solve packs it; an idle slot (registration -1) has id = iteration & 15"""
    return 16.0 * (registration + 1) + (iteration & 15)


def rank_check_values(registration, iteration):
    i = rank_check_id(registration, iteration)
    return (1.0, i, i * i)


def rank_check_ok(record, registration, iteration):
    """record: one scan's 32 all-reduced doubles; True iff every contributing rank packed this rank's id"""
    n, a1, a2 = float(record[29]), float(record[30]), float(record[31])
    i = rank_check_id(registration, iteration)
    return n >= 1.0 and a1 == n * i and a2 == n * i * i


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
