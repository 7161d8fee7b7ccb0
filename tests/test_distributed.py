"""N > 1 path on CPU: world_size-2 gloo processes run the sharded registration PROTOCOL -- contiguous scan shards, the
32-double packed record per scan, ONE all-reduce per ICP iteration, gates evaluated after the reduce against n_total,
pose recomputed redundantly and identically on every rank -- with the CPU oracle standing in for the per-shard
accumulate kernel (the HIP kernels need a GPU; their own sharded path is covered by test_gpu_parity.py::
test_sharded_hook_sums and by bench.py --gpus N on the MI355X node)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world_size, port, method, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from elimaloc_amd import synth
    from elimaloc_amd.dist import shard_bounds, pack_sums, unpack_sums, PACKED_SUMS
    from oracle import oracle as O
    import np_ref

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    world = synth.make_world(20000, seed=5)
    scans, T0s = [], []
    for b in range(2):  # a batch of two scans, reduced together
        sc, Tt = synth.make_scan(world, 3001 + 500 * b, seed=60 + b)
        scans.append(sc)
        T0s.append(synth.perturb(Tt, seed=70 + b, max_trans=0.2 + 0.2 * b, max_rot_deg=1.0))
    m = O.Map(1.0, 30)
    m.add_points(world)
    if method in (2, 3):
        m.cal_voxel_cov_all(2)
    if method == 1:
        m.cal_point_cov_all(0.4, 2)
    cfg1 = O.default_config(method, max_iteration=1, min_overlap_ratio=0.0, max_fitness_score=1e300, max_thread=2)
    full = O.default_config(method, max_thread=2)
    B = len(scans)
    T = [t.copy() for t in T0s]
    done = [False] * B
    iters = [0] * B
    fitness = [0.0] * B
    for it in range(full.max_iteration):
        buf = np.zeros(B * PACKED_SUMS)
        for b in range(B):
            if done[b]:
                continue
            lo, hi = shard_bounds(len(scans[b]), rank, world_size)
            if hi > lo:
                r = O.register(m, scans[b][lo:hi], T[b], cfg1)["iters"][0]
                if "JTJ" in r and r["n_corr"] > 0:
                    buf[b * PACKED_SUMS:(b + 1) * PACKED_SUMS] = pack_sums(r["JTJ"], r["JTr"], r["residual_sum"], r["n_corr"])
        t = torch.from_numpy(buf)
        dist.all_reduce(t)  # ONE collective per ICP iteration for the whole batch
        for b in range(B):
            if done[b]:
                continue
            H, g, rs, n_corr = unpack_sums(buf[b * PACKED_SUMS:(b + 1) * PACKED_SUMS])
            iters[b] += 1
            if np.float32(n_corr) / np.float32(len(scans[b])) < full.min_overlap_ratio:  # gate on the WHOLE scan
                done[b] = True
                continue
            fitness[b] = rs / n_corr
            x = np.linalg.solve(H + full.lm_lambda * np.diag(np.diag(H)), g)
            dT = np.eye(4); dT[:3, :3] = np_ref.exp_so3(x[3:]); dT[:3, 3] = x[:3]
            T[b] = T[b] @ dT
            if np_ref.rot_angle(dT[:3, :3]) + np.linalg.norm(x[:3]) < full.icp_termination_threshold_m:
                done[b] = True
        if all(done):
            break
    # every rank holds the same poses (bitwise: the all-reduced buffer is identical everywhere)
    gathered = [None] * world_size
    dist.all_gather_object(gathered, [t.tobytes() for t in T])
    same = all(g == gathered[0] for g in gathered)
    ref = [O.register(m, scans[b], T0s[b], full) for b in range(B)] if rank == 0 else None
    if rank == 0:
        out_q.put(dict(same=same, iters=iters, T=T, fitness=fitness,
                       ref_iters=[r["iterations"] for r in ref], ref_T=[r["T"] for r in ref],
                       ref_fitness=[r["fitness"] for r in ref]))
    dist.destroy_process_group()


@pytest.mark.parametrize("method", [0, 2])
def test_sharded_protocol_world_size_2_gloo(method):
    import torch.multiprocessing as mp
    from elimaloc_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400) + method
    procs = [ctx.Process(target=_worker, args=(r, 2, port, method, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res["same"]
    assert res["iters"] == res["ref_iters"]
    for T, Tr, f, fr in zip(res["T"], res["ref_T"], res["fitness"], res["ref_fitness"]):
        dt, dr = synth.pose_error(Tr, T)
        assert dt < 1e-9 and dr < 1e-10
        assert abs(f - fr) < 1e-9


def test_shard_bounds_partition():
    from elimaloc_amd.dist import shard_bounds
    for n in (0, 1, 7, 131072, 262144 + 5):
        for w in (1, 2, 3, 8):
            cuts = [shard_bounds(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in cuts) - min(hi - lo for lo, hi in cuts) <= 1


def test_pack_unpack_round_trip():
    from elimaloc_amd.dist import pack_sums, unpack_sums, PACKED_SUMS
    from elimaloc_amd import _lib
    assert PACKED_SUMS == _lib.PACKED_SUMS == 32
    rng = np.random.default_rng(0)
    A = rng.normal(size=(6, 6)); H = A + A.T
    g = rng.normal(size=6)
    v = pack_sums(H, g, 3.5, 17.0)
    H2, g2, rs, n = unpack_sums(v)
    assert np.array_equal(H, H2) and np.array_equal(g, g2) and rs == 3.5 and n == 17.0


def _stream_worker(rank, world_size, port, out_q):
    """The multi-rank STREAM protocol of elm_register_stream (elm_k_solve.hip: claim_registration with StreamArgs::stride): S slots,
    slot s serves the registrations s, s + S, s + 2S, ...; whether a registration finishes at an iteration is decided on sums that
    every rank holds after the all-reduce of the slots' records -- so every rank takes the same refill decisions without any
    exchange about them, issues the same number of collectives, and stops at the same iteration."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    count, S, max_iter = 23, 5, 10
    rng = np.random.default_rng(1000 + rank)  # every rank contributes DIFFERENT partial sums (its shard of the points)
    slot_reg = [s if s < count else -1 for s in range(S)]  # initial fill (k_stream_refill, first = 1)
    slot_iter = [0] * S
    log, completed, n_collectives = [], 0, 0
    finished_at = {}
    while completed < count:
        buf = np.zeros(S * 32)
        for s in range(S):
            if slot_reg[s] >= 0:
                buf[s * 32] = rng.uniform(0.0, 1.0)  # this rank's part of a "step size" that only the sum defines
        t = torch.from_numpy(buf)
        dist.all_reduce(t)
        n_collectives += 1
        for s in range(S):
            r = slot_reg[s]
            if r < 0:
                continue
            slot_iter[s] += 1
            if buf[s * 32] < 0.45 * world_size or slot_iter[s] >= max_iter:  # termination on the all-reduced value
                finished_at[r] = slot_iter[s]
                completed += 1
                nxt = r + S  # static queue of the slot: a function of the slot alone
                slot_reg[s] = nxt if nxt < count else -1
                slot_iter[s] = 0
                log.append((n_collectives, s, r, slot_reg[s]))
    gathered = [None] * world_size
    dist.all_gather_object(gathered, (log, n_collectives, sorted(finished_at.items())))
    if rank == 0:
        out_q.put(dict(same=all(g == gathered[0] for g in gathered), served=sorted(finished_at), n_collectives=n_collectives,
                       iters=[finished_at[r] for r in sorted(finished_at)]))
    dist.destroy_process_group()


def test_stream_static_slot_queue_world_size_2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + (os.getpid() % 40)
    procs = [ctx.Process(target=_stream_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res["same"]                       # identical refill decisions, collective counts and iteration counts on both ranks
    assert res["served"] == list(range(23))  # every registration was served exactly once
    assert len(set(res["iters"])) > 1        # and they finished at different iterations (the refills really interleave)


def _refill_worker(rank, world_size, port, corrupt, out_q):
    """The DEFAULT multi-rank stream protocol (k_solve mode 1 -> all-reduce -> mode 2 -> k_stream_refill): free slots take the pending
    registrations in SLOT order (a prefix count over the finished flags, which derive from the all-reduced sums), and slots 29..31 of
    every exchanged record carry the rank-agreement check (elimaloc_amd.dist.rank_check_*).  corrupt: rank 1 sees another termination
    value for one slot once -- its slots drift apart from rank 0's, which the check must report on the very next exchange."""
    import torch
    import torch.distributed as dist
    from elimaloc_amd.dist import rank_check_values, rank_check_ok
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    count, S, max_iter = 29, 6, 10
    rng = np.random.default_rng(2000 + rank)
    slot_reg = [s if s < count else -1 for s in range(S)]
    slot_iter = [0] * S
    nxt, completed, n_coll, fault_at, corrupted = min(S, count), 0, 0, None, None
    order = []
    for _ in range(((count + S - 1) // S + 1) * max_iter):  # the host's hard limit: every rank issues the same number of collectives
        buf = np.zeros(S * 32)
        for s in range(S):
            if slot_reg[s] >= 0:
                buf[s * 32] = rng.uniform(0.0, 1.0)
            buf[s * 32 + 29:s * 32 + 32] = rank_check_values(slot_reg[s], slot_iter[s])
        t = torch.from_numpy(buf)
        dist.all_reduce(t)
        n_coll += 1
        free = []
        for s in range(S):
            if not rank_check_ok(buf[s * 32:(s + 1) * 32], slot_reg[s], slot_iter[s]) and fault_at is None:
                fault_at = n_coll
            if slot_reg[s] < 0:
                continue
            slot_iter[s] += 1
            v = buf[s * 32]
            if corrupt and rank == 1 and corrupted is None and n_coll >= 2 and v >= 0.45 * world_size and slot_iter[s] < max_iter:
                v = 0.0  # this rank alone believes the slot has converged
                corrupted = n_coll
            if v < 0.45 * world_size or slot_iter[s] >= max_iter:
                completed += 1
                order.append(slot_reg[s])
                free.append(s)
        for s in free:  # slot order
            slot_reg[s] = nxt if nxt < count else -1
            slot_iter[s] = 0
            nxt += 1 if slot_reg[s] >= 0 else 0
        if fault_at is not None and corrupt:
            break
        if completed >= count and not corrupt:
            break
    gathered = [None] * world_size
    dist.all_gather_object(gathered, (order, n_coll, fault_at, corrupted))
    if rank == 0:
        out_q.put(dict(orders=[g[0] for g in gathered], n_coll=[g[1] for g in gathered], fault_at=[g[2] for g in gathered], corrupted=gathered[1][3]))
    dist.destroy_process_group()


@pytest.mark.parametrize("corrupt", [False, True])
def test_stream_slot_order_refill_and_rank_check_world_size_2_gloo(corrupt):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 200) + (50 if corrupt else 0)
    procs = [ctx.Process(target=_refill_worker, args=(r, 2, port, corrupt, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if not corrupt:
        assert res["orders"][0] == res["orders"][1] and sorted(res["orders"][0]) == list(range(29))  # same refill decisions, every registration once
        assert res["n_coll"][0] == res["n_coll"][1] and res["fault_at"] == [None, None]
    else:
        # rank 1 freed a slot one exchange early and refilled it: at the NEXT exchange both ranks see sum(id) != n id for that slot
        assert res["corrupted"] is not None and res["fault_at"] == [res["corrupted"] + 1] * 2, res


def test_spatial_shards_partition_the_scan_compactly():
    """locality-aware sharding (dist.spatial_shards): a partition of the scan into world_size contiguous parts of its Hilbert order --
    every point in exactly one shard, shard sizes as shard_bounds, and a shard's footprint (occupied 2 m cells) is about 1 / W of the
    scan's, where a strided shard occupies nearly all of them"""
    from elimaloc_amd import dist as D
    from elimaloc_amd import synth
    world = synth.make_world(200000, seed=1001)
    sc, _ = synth.make_scan(world, 20001, seed=77)
    for W in (1, 2, 3, 8):
        parts = D.spatial_shards(sc, W)
        assert [len(p) for p in parts] == [D.shard_bounds(len(sc), r, W)[1] - D.shard_bounds(len(sc), r, W)[0] for r in range(W)]
        allp = np.concatenate(parts)
        assert sorted(map(tuple, allp)) == sorted(map(tuple, sc))
    cells = lambda p: len({(int(np.floor(x / 2)), int(np.floor(y / 2))) for x, y, _ in p})  # noqa: E731
    total = cells(sc)
    parts = D.spatial_shards(sc, 8)
    assert max(cells(p) for p in parts) < 0.3 * total          # compact: ~1/8 of the footprint (+ boundary cells)
    assert cells(sc[0::8]) > 0.6 * total                        # a thinned copy touches most of it
    # the order is the device's: stable sort by the Hilbert index of the 2 m cell
    o = D.spatial_order(sc)
    assert sorted(o) == list(range(len(sc)))


def test_rank_check_id_stays_exact_for_any_registration_count():
    """n id^2 must be an exact integer in a double whatever the registration index (the index enters modulo 2^19)"""
    from elimaloc_amd.dist import rank_check_id, rank_check_values, rank_check_ok
    for reg in (-1, 0, 1, 2**19 - 1, 2**19, 3 * 2**19 + 5, 2**31 - 1):
        for it in (0, 9, 31):
            i = rank_check_id(reg, it)
            assert i < 2**23 + 16 and float(int(i)) == i
            for n in (1, 2, 8, 64):
                rec = np.zeros(32)
                for _ in range(n):  # the all-reduce: n ranks add the same triple (any summation tree: every partial sum is exact)
                    rec[29:32] += rank_check_values(reg, it)
                assert rec[31] == n * i * i < 2**53 and rank_check_ok(rec, reg, it)
                rec[30] += 16.0  # one rank iterated the next registration in this slot
                assert not rank_check_ok(rec, reg, it)
    assert rank_check_id(-1, 3) == 3.0 and rank_check_id(0, 0) == 16.0
