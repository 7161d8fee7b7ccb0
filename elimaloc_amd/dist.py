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
    """id of a slot's (registration, iteration) as the solve packs it; an idle slot (registration -1) has id = iteration & 15.  The
    registration enters modulo 2^19: n id^2 stays an exact integer in a double for any number of registrations per call (n <= 64 ranks)"""
    return 16.0 * ((registration & 0x7FFFF if registration >= 0 else -1) + 1) + (iteration & 15)


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


# ---- locality-aware sharding (round 6) ----------------------------------------------------------------------------------------------
# A rank's shard should be a spatially compact part of the scan, not a thinned-out copy of the whole of it: the search index is addressed
# by position, so a shard that covers 1/W of the scan's footprint at full density touches 1/W of the index entries the scan touches, while
# every W-th point of the whole footprint touches nearly all of them (measured on one MI355X, VGICP, 32 768-point shards of 262 144-point
# scans against the 50 M-point map: see DESIGN.md section 6).  The points' order is not contractual (the reference's own VoxelDownsample
# emits unordered_map order, vhm.hpp:278-280) -- a LiDAR driver's azimuth order has the same property for free; a scan in arbitrary order
# is sorted along the Hilbert curve the device's ordering kernel uses (2 m sensor-frame cells) before it is cut into contiguous shards.
_HILBERT_LUT = None


def _hilbert_lut():
    global _HILBERT_LUT
    if _HILBERT_LUT is None:
        lut = np.zeros((64, 64), np.int64)
        ys, xs = np.meshgrid(np.arange(64), np.arange(64), indexing="ij")
        x, y, d = xs.copy(), ys.copy(), np.zeros((64, 64), np.int64)
        s = 32
        while s > 0:  # the textbook xy2d, vectorised over the 64 x 64 cells (elm_api.cpp: hilbert_xy2d)
            rx = (x & s) != 0
            ry = (y & s) != 0
            d += s * s * ((3 * rx.astype(np.int64)) ^ ry.astype(np.int64))
            flip = (~ry) & rx
            x = np.where(flip, s - 1 - x, x)
            y = np.where(flip, s - 1 - y, y)
            swap = ~ry
            x, y = np.where(swap, y, x), np.where(swap, x, y)
            s >>= 1
        lut[:, :] = d
        _HILBERT_LUT = lut
    return _HILBERT_LUT


def spatial_order(xyz):
    """permutation that sorts a scan (n x 3 float32, sensor frame) along the Hilbert curve over 2 m cells -- the stable sort
    k_scan_order performs on the device (tests/test_hostfed.py)"""
    xyz = np.asarray(xyz, np.float32)
    cx = np.clip(np.floor(xyz[:, 0] * np.float32(0.5)).astype(np.int64) + 32, 0, 63)
    cy = np.clip(np.floor(xyz[:, 1] * np.float32(0.5)).astype(np.int64) + 32, 0, 63)
    return np.argsort(_hilbert_lut()[cy, cx], kind="stable")


def spatial_shards(xyz, world_size):
    """the world_size contiguous shards of the spatially ordered scan (every point in exactly one shard, sizes as shard_bounds)"""
    xyz = np.asarray(xyz, np.float32)
    o = xyz[spatial_order(xyz)]
    return [np.ascontiguousarray(o[slice(*shard_bounds(len(o), r, world_size))]) for r in range(world_size)]
