#!/usr/bin/env python3
"""bench.py -- ICP registrations/sec of the MI355X-native pcm_matching hot path.

Contract: `python bench.py --gpus N --steps K --warmup W`.  N > 1: one rank per GPU -- either launched by the caller
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, the driver's form) or, when no rank environment is set, by
bench.py itself (it re-executes under torch.distributed.run on 127.0.0.1); `--gpus` must equal the number of ranks, a box with fewer GPUs is
refused with a non-zero exit, and the line carries `rccl_ranks` (ncclCommCount of the communicator the timed region all-reduced over).
One STEP = one pass of the hot path over one batch of synthetic input: `--batch` (per GPU) full RunRegister-equivalent
registrations (initial transform -> iterate to the reference's own termination rule or max_iteration), all scans
already resident in HBM -- uploaded as packed float32 xyz and ordered ON THE DEVICE (k_scan_order) -- when the timed region
starts.  What `value` does NOT time: the H2D upload and the ordering kernel of every scan.  The `host_fed` object (N = 1) times
exactly that as well: every scan starts in (page-locked) HOST memory and is uploaded, ordered and registered inside the timed
region (elm_register_stream_host), reported beside the PCIe rate it sits under.

Workload (BASELINE.json configs[1]): P2P ICP, 131072-pt synthetic scans vs a 10M-pt voxel-hashed map, defaults of
config/localization.ini.  N>1: every scan is sharded point-wise over the N GPUs (map replicated), ONE RCCL all-reduce
of the packed normal equations of the whole batch per ICP iteration; the batch grows with N (weak scaling: per-GPU
points per launch fixed); rank r generates only the scans i = r (mod N) and one all-to-all per 32 scans hands out the shards
(bit-identical inputs to the one-rank generation).  With N>1 the same registrations are also timed in replica mode (whole
registrations per GPU, no collective) and reported under "replica" (SURVEY.md 8e asks for both).
The timed launches carry no instrumentation: the work counters (C, V, tested candidates) come from one untimed pass of the same step.

Prints ONE JSON line on rank 0 with
  * `inputs`: SHA-1 of every uploaded scan + initial guess (the workload is bit-reproducible: seeded, BLAS-free),
  * `roofline`: dominant kernel and the unit that binds it -- VALU issue for the grid kernel: SQ_INSTS_VALU per SIMD-cycle of the
    committed rocprofv3 --pmc pass of this command AT THIS batch / slots (refused otherwise) x the kernel's mean issue cost
    (tools/probes/valu_probe + tools/valu_mix.py), with the hipEvent-measured launch time it belongs to; under
    `hbm` the HBM stream (measured fabric bytes/unit x this run's units per launch / launch time against 8 TB/s), the compulsory
    stream, the bytes the points REQUEST from the kernel's own structures and the SURVEY 8(d) figure of the REFERENCE's walk,
  * at N=1 `cpu_baseline`: the CPU oracle on the host cores, SURVEY 8(d) protocol (3 warm-ups, >= 20 timed
    registrations, median / p10 / p90, correspondence-vs-total split, 10 threads and all cores, a full-map sample),
    `pose_err_vs_cpu`, `reference_api` (RunRegister on host buffers, one call at a time) and `hard_guess` (the 0.5 m /
    2 deg initial-guess set timed the same way).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
METHOD_NAMES = {0: "P2P", 1: "GICP", 2: "VGICP", 3: "AVGICP"}
SCAN_RANGE_M = 60.0
SCAN_NOISE_M = 0.01


def b_alg_reference(method, C, V):
    """Algorithmic bytes per scan point per ICP iteration of the REFERENCE's walk, compact-layout model of SURVEY.md 8(d)."""
    if method == 0:
        return 444.0 + 12.0 * C
    if method == 1:
        return 480.0 + 12.0 * C
    if method == 2:
        return 468.0 + 12.0 * V
    return 124.0 + 36.0 * V


def kernel_bytes_model(method, tested, V, pairs, grid=True):
    """Bytes one scan point REQUESTS from this kernel's own data structures in one ICP iteration.

    P2P/GICP on the dense cell grid (k_accumulate_grid): 16 scan point (float4) + 4 walk-statistics word + 4 x 12 column
    offset triples + 12 per distance-tested candidate + 12 winner re-read (+ 4 bucket index + 128 covariance record for
    GICP) + 1 (256-byte partial record per 256-point workgroup).  On the neighbourhood lists (k_accumulate_cell): 32 hash
    slot + 4 x 16 column records instead of the statistics word and the offset triples.
    VGICP (k_accumulate_vnbr): 16 + 32 slot + 32 per voxel-mean record (V = occupied neighbours) + 72 winner covariance + 1.
    AVGICP: 16 + 32 + 32 V + 72 per emitted pair + 1."""
    index = (4.0 + 48.0) if grid else (32.0 + 64.0)
    if method == 0:
        return 16.0 + index + 12.0 * tested + 12.0 + 1.0
    if method == 1:
        return 16.0 + index + 12.0 * tested + 12.0 + 4.0 + 128.0 + 1.0
    if method == 2:
        return 16.0 + 32.0 + 32.0 * V + 72.0 + 1.0
    return 16.0 + 32.0 + 32.0 * V + 72.0 * pairs + 1.0


def percentile(v, q):
    return float(np.percentile(np.asarray(v, dtype=np.float64), q))


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed.run environment: become the launcher -- one rank per GPU of this
    node, rendezvous on 127.0.0.1 (the form the driver itself uses for N > 1).  A box with fewer than N GPUs is refused HERE, loudly:
    a run labelled N GPUs never silently measures one."""
    if not args.dry_launch:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: this box has {have} GPU(s); one rank per GPU is the only mode "
                             f"(no oversubscription, no CPU fallback) -- refusing to run")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    sys.stdout.flush()
    sys.stderr.flush()
    os.execvpe(cmd[0], cmd, dict(os.environ))


def generate_inputs(world, args, rank, world_size, dist, guess, n_keep, keep_own, consume):
    """The step's registrations: n_batch = batch x world_size seeded scans + initial guesses; `consume(i, shard, n)` receives, in order
    of i, THIS rank's contiguous shard [n rank / W, n (rank + 1) / W) of scan i (packed float32 xyz, the caller's point order).

    One rank: every scan is generated here.  W ranks: rank r generates only the scans i = r (mod W) -- the host work per rank does not
    grow with W -- and one all-to-all per round of 32 scans (host tensors over gloo, outside every timed region) hands every rank its
    shard of every scan; the per-scan metadata (true pose, initial guess, SHA-1, extent) is all-gathered.  Bit-identical to the
    one-rank generation (same seeds; `inputs.sha1` covers all of it).
    Returns dict(T_true, T0s, digests, rmaxs, kept = {i: full scan} for i < n_keep on rank 0 (+ this rank's own scans if keep_own))."""
    from concurrent.futures import ThreadPoolExecutor
    from elimaloc_amd import synth
    n_batch = args.batch * world_size
    npts = args.scan_points

    def gen(i):
        sc, Tt = synth.make_scan(world, npts, seed=2002 + i, max_range=SCAN_RANGE_M, noise=SCAN_NOISE_M)
        T0 = synth.perturb(Tt, seed=3003 + i, **guess)
        h = hashlib.sha1(sc.tobytes())
        h.update(np.ascontiguousarray(T0).tobytes())
        rmax = float(np.sqrt((sc.astype(np.float64) ** 2).sum(axis=1).max())) if sc.shape[0] else 0.0
        return sc, Tt, T0, h.digest(), rmax

    synth.make_scan(world, 16, seed=1)  # builds the (cached) tile index of the world before the threads start
    meta = [None] * n_batch  # (T_true, T0, digest, rmax)
    kept = {}
    workers = max(2, min(16, (os.cpu_count() or 16) // max(world_size, 1)))
    with ThreadPoolExecutor(max_workers=workers) as pool:  # numpy releases the GIL in the heavy parts; make_scan calls no BLAS
        if world_size == 1:
            for i, (sc, Tt, T0, dg, rmax) in enumerate(pool.map(gen, range(n_batch))):
                meta[i] = (Tt, T0, dg, rmax)
                if i < n_keep or keep_own:
                    kept[i] = sc
                consume(i, sc, sc.shape[0])
        else:
            import torch
            bounds = [npts * d // world_size for d in range(world_size + 1)]
            mine = bounds[rank + 1] - bounds[rank]
            ROUND = 32
            for m0 in range(0, args.batch, ROUND):
                ms = list(range(m0, min(m0 + ROUND, args.batch)))
                cnt = len(ms)
                own = list(pool.map(gen, [rank + world_size * m for m in ms]))
                if any(o[0].shape[0] != npts for o in own):
                    raise SystemExit("make_scan returned a scan of another size")
                # send buffer: destination-major, then scan; receive buffer: source-major, then scan
                inp = torch.from_numpy(np.concatenate([o[0][bounds[d]:bounds[d + 1]].reshape(-1) for d in range(world_size) for o in own]))
                out = torch.empty(world_size * cnt * mine * 3, dtype=torch.float32)
                dist.all_to_all_single(out, inp, [cnt * mine * 3] * world_size, [cnt * (bounds[d + 1] - bounds[d]) * 3 for d in range(world_size)])
                metas = [None] * world_size
                dist.all_gather_object(metas, [(o[1], o[2], o[3], o[4]) for o in own])
                got = out.numpy().reshape(world_size, cnt, mine, 3)
                for j, m in enumerate(ms):
                    for src in range(world_size):
                        i = src + world_size * m
                        meta[i] = metas[src][j]
                        consume(i, np.ascontiguousarray(got[src, j]), npts)
                    if keep_own:
                        kept[rank + world_size * m] = own[j][0]
                    elif rank == 0 and world_size * m < n_keep:
                        kept[world_size * m] = own[j][0]
        # the pooled generation must equal a sequential one (round 1's pool corrupted rows through concurrent OpenBLAS calls)
        mine_idx = list(range(n_batch)) if world_size == 1 else list(range(rank, n_batch, world_size))
        for i in sorted(set([mine_idx[0], mine_idx[len(mine_idx) // 3], mine_idx[-1]])):
            if gen(i)[3] != meta[i][2]:
                raise SystemExit(f"input generation is not deterministic (scan {i})")
    return dict(T_true=[m[0] for m in meta], T0s=[m[1] for m in meta], digests=[m[2] for m in meta], rmaxs=[m[3] for m in meta], kept=kept)


def dry_launch(args, rank, world_size, local_rank, dist):
    """--dry-launch: the launcher, the rendezvous and the sharded input generation + exchange WITHOUT a GPU (gloo only) -- what the CPU
    test suite runs at N = 2.  Prints one JSON line (no `value`: nothing is measured)."""
    from elimaloc_amd import synth
    world = synth.make_world(args.map_points, seed=1001)
    guess = dict(max_trans=0.15, max_rot_deg=0.5) if args.guess == "easy" else dict(max_trans=0.5, max_rot_deg=2.0)
    shard_sha = hashlib.sha1()
    shard_points = [0]

    def consume(i, shard, n):
        shard_sha.update(shard.tobytes())
        shard_points[0] += shard.shape[0]

    g = generate_inputs(world, args, rank, world_size, dist, guess, 0, False, consume)
    who = [None] * world_size
    mine = dict(rank=rank, local_rank=local_rank, pid=os.getpid(), shard_points=shard_points[0], shard_sha1=shard_sha.hexdigest())
    if world_size > 1:
        dist.all_gather_object(who, mine)
    else:
        who = [mine]
    return {"dry_launch": True, "metric": "ICP registrations/sec, 128k-pt scan vs 10M-pt map; pose err vs CPU ref", "value": None,
            "n_gpus": world_size, "rccl_ranks": None, "ranks": who,
            "inputs": {"sha1": hashlib.sha1(b"".join(g["digests"])).hexdigest(), "registrations": len(g["digests"])}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="registrations per GPU per step (0 = 4096 on one GPU -- a timed region of ~1 s at 20 steps, "
                    "five draining launches per ~110 -- and 1024 per GPU on several)")
    ap.add_argument("--hostfed-batch", type=int, default=2048, help="registrations per step of the host-fed leg (N = 1; 0 = skip)")
    ap.add_argument("--hostfed-steps", type=int, default=8)
    ap.add_argument("--hostfed-slots", type=int, default=128, help="device slots of the host-fed leg (PCIe-bound: a third of them is busy)")
    ap.add_argument("--slots", type=int, default=256, help="registrations iterating concurrently per GPU (measured 128 / 256 / 512: 87.6 / 90.7 / 90.3 k/s; continuous batching: "
                    "finished slots take the next pending registration on the device); 0 = lockstep batch of --batch")
    ap.add_argument("--scan-points", type=int, default=131072)
    ap.add_argument("--map-points", type=int, default=10_000_000)
    ap.add_argument("--method", type=int, default=0, help="0 P2P (configs[1]), 1 GICP, 2 VGICP, 3 AVGICP")
    ap.add_argument("--guess", choices=("easy", "hard"), default="easy", help="initial-guess set of SURVEY 8(d): easy = 0.15 m / 0.5 deg "
                    "(the headline workload), hard = 0.5 m / 2 deg")
    ap.add_argument("--cpu-sample", type=int, default=20, help="registrations timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-full-sample", type=int, default=3, help="of those, registrations repeated on the un-cropped map (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip latency / reference-API / hard-guess / replica legs (profiling passes)")
    ap.add_argument("--no-latency", action="store_true", help="alias of --no-extras")
    ap.add_argument("--asym-triples", type=int, default=0, help="append this many isolated collinear point triples to the world (synth.collinear_triples): "
                    "rank-1 neighbourhoods whose regularised covariance is not symmetric (layout bits 7 / 8) -- the covariance methods then "
                    "carry the antisymmetric side records")
    ap.add_argument("--dry-launch", action="store_true", help="launcher + rendezvous + sharded input generation only, gloo, no GPU (CPU test of the N > 1 path)")
    args = ap.parse_args()
    extras = not (args.no_extras or args.no_latency)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(args, sys.argv[1:])  # does not return

    # stdout carries exactly ONE JSON line: everything libraries print (RCCL banners, gloo notices) goes to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    # --gpus IS the number of ranks, whoever launched them (the driver's torch.distributed.run for N > 1, launch_ranks above, plain
    # python for N = 1): a mismatch is an error, never a silently smaller run
    if args.gpus != world_size:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world_size}: launch one rank per GPU (python bench.py --gpus N does it itself)")
    if args.batch <= 0:
        args.batch = 4096 if world_size == 1 else 1024
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # launched by torch.distributed.run (RANK set): take the collective path even for one rank, so that a 1-GPU box
    # exercises exactly the code the 8-GPU node runs
    distributed = world_size > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if args.dry_launch:
        if distributed:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", rank=rank, world_size=world_size)
        line = dry_launch(args, rank, world_size, local_rank, dist)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if distributed:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback")
    if torch.cuda.device_count() < world_size or local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py --gpus {args.gpus}: {torch.cuda.device_count()} GPU(s) visible to rank {rank} (local rank {local_rank}); one rank per GPU is the only mode")
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="cpu:gloo,cuda:nccl", rank=rank, world_size=world_size)

    from elimaloc_amd import synth
    from elimaloc_amd.registration import (Context, VoxelHashMap, Registration, RegistrationConfig, IcpMethod, Scan, PinnedBuffer,
                                           results_from_raw)

    method = IcpMethod(args.method)
    ctx = Context(local_rank)
    rccl_ranks, rank_table = None, None
    if distributed:
        ids = [Context.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ctx.comm_init(rank, world_size, ids[0])
        # what the RCCL communicator itself reports: the run is an N-GPU run only if N ranks joined it, each on a device of its own
        crank, rccl_ranks = ctx.comm_info()
        if rccl_ranks != world_size or crank != rank:
            raise SystemExit(f"RCCL communicator reports rank {crank} of {rccl_ranks}, expected rank {rank} of {world_size}")
        props = torch.cuda.get_device_properties(local_rank)
        rank_table = [None] * world_size
        dist.all_gather_object(rank_table, dict(rank=rank, local_rank=local_rank, pid=os.getpid(), device=props.name,
                                                gpu_uuid=str(getattr(props, "uuid", "")), rccl_rank=crank))
        if len({(r["local_rank"], r["gpu_uuid"]) for r in rank_table}) != world_size or len({r["pid"] for r in rank_table}) != world_size:
            raise SystemExit(f"ranks do not sit on {world_size} distinct GPUs / processes: {rank_table}")

    # ---------------- synthetic inputs (seeded, BLAS-free arithmetic: bit-identical whoever generates them) ----------------
    t0 = time.time()
    world = synth.make_world(args.map_points, seed=1001)
    if args.asym_triples > 0:
        world = np.ascontiguousarray(np.concatenate([world, synth.collinear_triples(world, args.asym_triples, seed=4004)]))
    vm = VoxelHashMap(1.0, 30, ctx)
    vm.AddPoints(world)
    if method in (IcpMethod.VGICP, IcpMethod.AVGICP):
        vm.CalVoxelCovAll()
    if method == IcpMethod.GICP:
        vm.CalPointCovAll(0.4)
    info = vm.info()
    t_map = time.time() - t0
    n_batch = args.batch * world_size  # weak scaling: per-GPU points per launch fixed
    n_keep = max(args.cpu_sample, 8) if (rank == 0 and world_size == 1) else 0  # full host copies kept for the CPU / reference-API legs (N = 1)
    guess = dict(max_trans=0.15, max_rot_deg=0.5) if args.guess == "easy" else dict(max_trans=0.5, max_rot_deg=2.0)
    want_replica = extras and distributed and args.slots > 0 and (world_size > 1 or bool(os.environ.get("ELM_BENCH_FORCE_REPLICA")))

    scans = []
    # host-fed leg (N = 1): the first `n_fed` scans once more in ONE page-locked buffer, back to back (what a driver's DMA ring holds)
    n_fed = min(args.hostfed_batch, n_batch) if (world_size == 1 and not distributed and extras and args.slots > 0) else 0  # (a host-fed stream runs on one rank without a communicator)
    pin = PinnedBuffer(max(1, n_fed * args.scan_points * 3)) if n_fed else None
    fed_sizes = []

    def consume(i, shard, n):
        if i < n_fed:
            o = sum(fed_sizes) * 3
            pin.array[o:o + shard.size] = shard.ravel()
            fed_sizes.append(shard.shape[0])
        scans.append(Scan(ctx, shard, n_total=n))  # H2D upload + device-side ordering of THIS rank's shard

    gin = generate_inputs(world, args, rank, world_size, dist, guess, n_keep, want_replica, consume)
    T_true, T0s, digests, rmaxs = gin["T_true"], gin["T0s"], gin["digests"], gin["rmaxs"]
    scans_host = [gin["kept"][i] for i in range(n_keep)] if n_keep else []
    # every scan point lies within the sensor range (+ 5 sigma of the noise on every axis)
    if max(rmaxs) > SCAN_RANGE_M + 5.0 * SCAN_NOISE_M * 3 ** 0.5:
        raise SystemExit(f"corrupted scan: max |p| = {max(rmaxs):.3f} m")
    inputs_sha1 = hashlib.sha1(b"".join(digests)).hexdigest()
    t_in = time.time() - t0 - t_map
    cfg = RegistrationConfig(icp_method=method)
    reg = Registration(cfg, ctx)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    n_slots = args.slots * world_size  # per-GPU points per launch stay fixed as ranks are added
    packed = reg.pack_inputs(scans, T0s)  # handle array + column-major guesses, marshalled once

    def step():
        if args.slots > 0:
            return reg.RunRegisterStream(packed[0], vm, packed[1], slots=n_slots, raw=True)  # results complete in host memory; dicts later
        return reg.RunRegisterBatch(scans, vm, T0s)

    for _ in range(args.warmup):
        out = step()
    ctx.set_profiling(True)
    ctx.get_profile(reset=True)
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t_start
    prof = ctx.get_profile(reset=True)
    ctx.set_profiling(False)
    if args.slots > 0:
        out = results_from_raw(out)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    regs = n_batch * args.steps
    value = regs / elapsed
    iters = np.array([r["iterations"] for r in out])
    pt_iters = float(sum(r["point_iterations"] for r in out))          # whole batch, all ranks (all-reduced sums)
    # The timed launches carry no instrumentation.  The work counters (candidates C / occupied voxels V of the reference's walk, candidates
    # this kernel distance-tests) come from ONE untimed pass of the same step with the counters compiled in: same poses, bit for bit.
    ctx.set_work_counters(True)
    out_c = results_from_raw(step()) if args.slots > 0 else step()
    ctx.set_work_counters(False)
    if not all(np.array_equal(a["T"], b["T"]) and a["iterations"] == b["iterations"] for a, b in zip(out_c, out)):
        raise SystemExit("the instrumented pass does not reproduce the timed one")
    C = float(sum(r["n_cand_total"] for r in out_c)) / max(pt_iters, 1)  # candidates of the reference's walk per point-iteration
    V = float(sum(r["n_occ_total"] for r in out_c)) / max(pt_iters, 1)   # occupied neighbour voxels per point-iteration
    tested = float(sum(r["n_tested_total"] for r in out_c)) / max(pt_iters, 1)  # candidates this kernel distance-tests
    bytes_ref = b_alg_reference(int(method), C, V)
    info = vm.info()
    grid = int(info.nbr_entries) == int(info.n_points)  # the dense cell grid holds every map point once (the lists: 27 times)
    bytes_unit = kernel_bytes_model(int(method), tested, V, C if int(method) == 3 else 0.0, grid)
    forced = os.environ.get("ELM_KERNEL", "")  # developer switch: "direct" = the plain 27-probe walk for every method
    kernel_name = (f"k_accumulate_direct<{METHOD_NAMES[int(method)]}>" if forced == "direct" else
                   f"k_accumulate_vnbr<{METHOD_NAMES[int(method)]}>" if int(method) in (2, 3) else
                   f"k_accumulate_{'grid' if grid else 'cell'}<{METHOD_NAMES[int(method)]}>")
    # dominant kernel: k_accumulate. Units one launch processes ON THIS GPU = its shard of the batch's live points.
    launches = max(prof["accumulate_launches"], 1)
    acc_ms_avg = prof["accumulate_ms"] / launches
    units_per_launch = (pt_iters / world_size) * args.steps / launches
    sec = acc_ms_avg * 1e-3
    requested_gbs = bytes_unit * units_per_launch / sec / 1e9 if sec > 0 else 0.0
    ref_gbs = bytes_ref * units_per_launch / sec / 1e9 if sec > 0 else 0.0
    # COMPULSORY HBM bytes of one launch: every scan point once (16 B) + its share of the partial records (1 B) + the GICP payload
    # of its match (128 B: a map record is matched by ~0.3 scan points, no reuse) + the search index at most once (it is
    # cache-resident: the 4 MB L2s and the 256 MB Infinity Cache serve the re-reads)
    index_once = float(info.index_bytes)
    compulsory_unit = 17.0 + (128.0 if int(method) == 1 else 0.0) + index_once / max(units_per_launch, 1.0)
    compulsory_gbs = compulsory_unit * units_per_launch / sec / 1e9 if sec > 0 else 0.0
    # MEASURED HBM traffic: bytes / unit of the rocprofv3 --pmc passes of this command committed under profiles/ (2 x FETCH_SIZE +
    # WRITE_SIZE, gfx950 correction of MI355X_MICROARCH.md), scaled by this run's units per launch
    traffic, traffic_src, pmc_extra = None, None, {}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        try:
            pm = json.load(open(pmc_path)).get(kernel_name + ("@hard" if args.guess == "hard" else ""))
            # a counter pass speaks for THIS run only if it was taken at this operating point: same scan / map size, guess set, registrations
            # per step and slots (the launch mix -- live slots per launch, draining launches -- follows from those); otherwise the line falls
            # back to the compulsory HBM stream and says so
            if (pm and pm.get("scan_points") == args.scan_points and args.map_points == 10_000_000 and pm.get("guess", "easy") == args.guess
                    and pm.get("batch") == args.batch and pm.get("slots") == args.slots and world_size == 1):
                traffic = pm.get("hbm_bytes_per_unit") * units_per_launch
                traffic_src = (f"profiles/pmc_latest.json[{kernel_name}]: HBM bytes/unit of separate rocprofv3 --pmc passes of this command "
                               f"({pm.get('source', 'see profiles/README.md')}) x this run's units per launch; counters cannot be read in the timed run")
                pmc_extra = {k: pm[k] for k in ("valu_busy", "valu_insts_per_simd_cycle", "ta_busy", "l1_line_accesses_per_cu_cycle", "l1_hit", "l2_hit", "valu_insts_per_wave",
                                                "resident_waves_per_simd", "ps_per_unit_traced", "batch", "slots", "source") if k in pm}
        except Exception:  # noqa: BLE001
            traffic = None
    measured_gbs = (traffic / sec / 1e9) if (traffic and sec > 0) else None
    # `achieved`: the measured HBM stream when a matching counter pass is committed, else the compulsory stream
    achieved_gbs = measured_gbs if measured_gbs is not None else compulsory_gbs

    hbm = {
        "achieved": achieved_gbs,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved_gbs / HBM_PEAK_GBS,
        "achieved_is": "measured HBM traffic (committed counter pass: 2 x FETCH_SIZE + WRITE_SIZE -- an UPPER bound on DRAM bytes, FETCH_SIZE also "
                       "counts Infinity-Cache hits; tools/probes/gather_probe calibrates the x2 for this gather pattern) / hipEvent launch time"
                       if measured_gbs is not None else "compulsory HBM bytes / hipEvent launch time (no matching counter pass committed)",
        "traffic": traffic,
        "traffic_source": traffic_src,
        "compulsory_bytes_per_unit": compulsory_unit,
        "compulsory_gbs": compulsory_gbs,
        "compulsory_frac": compulsory_gbs / HBM_PEAK_GBS,
        # what the points ask the memory hierarchy for (L1 / L2 / Infinity Cache serve most of it: the index is cache-resident)
        "requested_bytes_per_unit": bytes_unit,
        "requested_gbs": requested_gbs,
        "requested_over_hbm_peak": requested_gbs / HBM_PEAK_GBS,
        "bytes_model": "requested: scan point + index words (cell offsets / hash slot + column records) + 12 B x tested candidate slots + "
                       "winner (+ payload) + partial record; compulsory: scan point + partial record (+ GICP record) + the index once per "
                       "launch; DESIGN.md section 4",
        # the SURVEY 8(d) model of the REFERENCE's 27-voxel walk (all C candidates): a speed-up-vs-model figure, not a utilisation
        "algorithmic_ref_bytes_per_unit": bytes_ref,
        "algorithmic_ref_gbs": ref_gbs,
        "algorithmic_ref_over_peak": ref_gbs / HBM_PEAK_GBS,
    }
    # The binding roof.  The committed counter passes of THIS command (same batch / slots: checked above) say which unit the kernel keeps
    # busiest.  VALU issue: wave64 VALU instructions per SIMD per shader cycle (SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)) x the
    # kernel's mean issue cost in cycles.  The cost model is measured, not assumed: tools/probes/valu_probe (profiles/r04_valu_probe.txt)
    # times every instruction class at 8 waves / SIMD -- ~2.4 cycles (v_fma_f32, v_add_f32, v_add_u32, v_and_b32, v_mov_b32), ~4.3 (packed
    # float32, every float64 op, v_and_or / v_med3 / v_min / v_max / shifts / v_cmp / v_cndmask / conversions), ~8.2 (rcp, sqrt) -- and
    # shows that SQ_ACTIVE_INST_VALU ticks ONCE per instruction whatever its class (twice for transcendentals), so round 3's
    # "4 x SQ_ACTIVE_INST_VALU / SIMD cycles" over-counted the 2.4-cycle class; tools/valu_mix.py weighs the kernel's own opcode mix
    # (profiles/r04_valu_mix.txt: 29 % fast, 70 % slow, 1 % transcendental for the P2P grid kernel -> 3.78 cycles).  TA busy is the
    # vector-memory front end.  Without a matching pass the line falls back to the HBM stream.
    clock_ghz = 2.4  # MI355X_MICROARCH.md: max clock
    simd_cycles = 1024 * clock_ghz  # G SIMD-cycles/s over the chip (256 CUs x 4 SIMDs)
    issue_cycles = {"k_accumulate_grid<P2P>": 3.78, "k_accumulate_grid<GICP>": 3.78, "k_accumulate_vnbr<VGICP>": 3.85, "k_accumulate_vnbr<AVGICP>": 3.86}  # profiles/r04_valu_mix.txt
    roofline = dict(hbm)
    roofline["bound"] = "hbm"
    vb = None
    if pmc_extra.get("valu_insts_per_simd_cycle") is not None and kernel_name in issue_cycles:
        vb = float(pmc_extra["valu_insts_per_simd_cycle"]) * issue_cycles[kernel_name]
    elif pmc_extra.get("valu_busy") is not None:  # a pass summarised before the probe: the x4 figure rescaled to the mix's mean cost
        vb = float(pmc_extra["valu_busy"]) / 4.0 * issue_cycles.get(kernel_name, 4.0)
    if vb is not None:
        tb = float(pmc_extra.get("ta_busy", 0.0))
        if vb >= tb and vb > hbm["frac"]:
            roofline = {"bound": "valu_issue", "achieved": vb * simd_cycles, "peak": simd_cycles, "unit": "G SIMD-cycles/s (VALU issue)", "frac": vb,
                        "achieved_is": f"SQ_INSTS_VALU per SIMD-cycle ({pmc_extra.get('valu_insts_per_simd_cycle')}) x {issue_cycles.get(kernel_name, 4.0)} cycles mean issue cost "
                                       "of this kernel's opcode mix (profiles/r04_valu_mix.txt, per-class costs measured by tools/probes/valu_probe: "
                                       "profiles/r04_valu_probe.txt); counters: committed rocprofv3 --pmc pass of this command at this batch / slots "
                                       "(profiles/); the launch time it belongs to is measured live below",
                        "frac_bounds": {"every_instruction_2.4_cycles": float(pmc_extra.get("valu_insts_per_simd_cycle", 0.0)) * 2.4,
                                        "every_instruction_4.3_cycles": float(pmc_extra.get("valu_insts_per_simd_cycle", 0.0)) * 4.3},
                        "traffic": traffic}
        elif tb > hbm["frac"]:
            roofline = {"bound": "vector_memory_issue", "achieved": tb * 256 * clock_ghz, "peak": 256 * clock_ghz, "unit": "G CU-cycles/s (TA-busy)", "frac": tb,
                        "achieved_is": "TA_TA_BUSY / (256 CUs x kernel cycles), committed rocprofv3 --pmc pass of this command (profiles/)",
                        "traffic": traffic}
        roofline["hbm"] = hbm
        roofline["valu_issue_frac"] = vb
        if pmc_extra.get("ps_per_unit_traced") is not None and units_per_launch > 0:
            roofline["counter_pass_ps_per_unit"] = pmc_extra["ps_per_unit_traced"]
            roofline["this_run_ps_per_unit"] = 1e9 * acc_ms_avg / units_per_launch
    roofline.update({
        "kernel": kernel_name,
        "counters": pmc_extra,
        "search_index": "dense cell grid" if grid else "neighbourhood lists",
        "index_bytes": int(info.index_bytes),
        "map_device_bytes": int(info.device_bytes),
        "tested_candidates_per_point": tested,
        "units_per_launch": units_per_launch,
        "avg_launch_ms": acc_ms_avg,
        "launches": prof["accumulate_launches"],
        "accumulate_ms_per_step": prof["accumulate_ms"] / args.steps,
        "solve_ms_per_step": prof["solve_ms"] / args.steps,
        "timed_region_s": elapsed,
    })

    result = {
        "metric": "ICP registrations/sec, 128k-pt scan vs 10M-pt map; pose err vs CPU ref",
        "value": value,
        "unit": "registrations/s",
        "n_gpus": world_size,
        "rccl_ranks": rccl_ranks,  # ncclCommCount of the communicator the timed region all-reduced over (null: one process, no communicator)
        "ranks": rank_table,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"{METHOD_NAMES[int(method)]} ICP, {args.scan_points}-pt scan vs {args.map_points}-pt voxel-hashed map "
                        f"(BASELINE configs[1] when P2P/131072/10M), localization.ini defaults, full convergence, "
                        f"initial guess {guess['max_trans']} m / {guess['max_rot_deg']} deg ({args.guess})",
            "batch_per_gpu": args.batch,
            "registrations_per_step": n_batch,
            "slots_per_gpu": args.slots,
            "scheduling": ("continuous batching: every ICP iteration is one launch over the slots, finished slots are refilled on "
                           "the device from the step's queue" if args.slots > 0 else "lockstep batch"),
            "parallelism": "1 GPU" if world_size == 1 else f"scan points sharded over {world_size} GPUs, map replicated, "
                           "one RCCL all-reduce (32 doubles/scan) per ICP iteration",
            "iterations_mean": float(iters.mean()),
            "iterations_min": int(iters.min()),
            "iterations_max": int(iters.max()),
            "success_rate": float(np.mean([r["is_success"] for r in out])),
            "map_points_retained": int(info.n_points),
            "map_voxels": int(info.n_voxels),
            "map_layout_flags": int(info.layout_flags),  # include/elimaloc_hip.h; bits 7 / 8 set would mean the per-pair (strict) path ran
            "candidates_per_point_C": C,
            "occupied_voxels_per_point_V": V,
            "map_build_s": t_map,
            "input_gen_s": t_in,
            "input_generation": ("every scan generated on this rank" if world_size == 1 else
                                 f"rank r generates the scans i = r mod {world_size}; one all-to-all per 32 scans (gloo, host) hands out the shards"),
        },
        "inputs": {
            "sha1": inputs_sha1,
            "what": f"sha1 over, per registration in order, sha1(float32 scan bytes + float64 row-major initial guess); {n_batch} registrations; "
                    "seeds 1001 / 2002+i / 3003+i; pooled generation verified against sequential generation",
            "max_abs_scan_m": max(rmaxs),
        },
        "inputs_timed": "in `value`: every ICP iteration of every registration (correspondence search, accumulation, solve, slot refill) and the "
                        "download of the results; NOT in `value`: the H2D upload and the device-side ordering of the scans (resident when the timed "
                        "region starts).  `host_fed` times those too.",
        "roofline": roofline,
    }

    # ---------------- extras outside the timed region ----------------
    if extras and args.slots > 0:
        # single-registration latency on resident scans (B = 1)
        lat = []
        for _ in range(5):
            t1 = time.perf_counter()
            reg.RunRegisterBatch(scans[:1], vm, T0s[:1])
            lat.append(time.perf_counter() - t1)
        result["config"]["latency_ms_batch1"] = 1e3 * float(np.median(lat))

    if extras and rank == 0 and world_size == 1 and scans_host:
        # the reference's API: Registration::RunRegister on host buffers (pcm.cpp:280-282), one call at a time
        k = min(8, len(scans_host))
        for i in range(min(2, k)):
            reg.RunRegister(scans_host[i], vm, T0s[i])
        tt = []
        same = True
        for i in range(k):
            t1 = time.perf_counter()
            pose, ok, fit, cov = reg.RunRegister(scans_host[i], vm, T0s[i])
            tt.append(time.perf_counter() - t1)
            same = same and float(np.abs(pose - out[i]["T"]).max()) < 1e-9  # caller's point order vs the Hilbert-ordered resident scan
        result["reference_api"] = {
            "what": "Registration::RunRegister-equivalent elm_register on pageable HOST buffers (numpy arrays: pipelined staging through the "
                    "context's pinned buffer + H2D in the caller's point order + all iterations + result download per call), sequential calls "
                    "on one context",
            "registrations_per_s": 1.0 / float(np.median(tt)),
            "ms_per_call_median": 1e3 * float(np.median(tt)),
            "n_calls": k,
            "pose_equals_stream_to_1e-9": same,
        }
        if n_fed >= k:
            # the same calls on page-locked sources (a LiDAR driver's DMA buffer): the copy engine reads the caller's memory directly
            views = [pin.array[3 * sum(fed_sizes[:i]):3 * sum(fed_sizes[:i + 1])].reshape(-1, 3) for i in range(k)]
            reg.RunRegister(views[0], vm, T0s[0])
            tp = []
            for i in range(k):
                t1 = time.perf_counter()
                pose, ok, fit, cov = reg.RunRegister(views[i], vm, T0s[i])
                tp.append(time.perf_counter() - t1)
                same = same and float(np.abs(pose - out[i]["T"]).max()) < 1e-9
            result["reference_api"]["page_locked_source"] = {"registrations_per_s": 1.0 / float(np.median(tp)), "ms_per_call_median": 1e3 * float(np.median(tp)),
                                                             "pose_equals_stream_to_1e-9": same}

    if n_fed and rank == 0:
        # The WHOLE registration inside the timed region: every scan starts in page-locked host memory, is uploaded (DMA, groups of
        # ~32 MB on a copy stream), ordered on the device and registered -- elm_register_stream_host, RunRegister's per-call contract
        # (reg.cpp:274-290) at stream rate.  Bit-identical to the resident stream.
        ptrs = [pin.ptr + 12 * sum(fed_sizes[:i]) for i in range(n_fed)]
        import ctypes as C
        packed_f = ((C.c_void_p * n_fed)(*ptrs), (C.c_uint32 * n_fed)(*fed_sizes),
                    np.ascontiguousarray(np.asarray(T0s[:n_fed], dtype=np.float64).reshape(-1, 4, 4).transpose(0, 2, 1)).reshape(-1), [pin])
        fed = reg.RunRegisterStreamHost(packed_f, vm, slots=args.hostfed_slots, raw=True)  # warm-up: staging sets, arena, side streams
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.hostfed_steps):
            fed = reg.RunRegisterStreamHost(packed_f, vm, slots=args.hostfed_slots, raw=True)
        barrier()
        tf_ = time.perf_counter() - t1
        fed = results_from_raw(fed)
        fed_bytes = 12.0 * float(sum(fed_sizes)) / n_fed
        fed_rate = n_fed * args.hostfed_steps / tf_
        h2d_gbs = ctx.measure_h2d(pin.ptr, int(12 * sum(fed_sizes)), reps=3)  # the WHOLE buffer the stream reads (its placement on the host's NUMA nodes matters)
        h2d_head_gbs = ctx.measure_h2d(pin.ptr, int(12 * sum(fed_sizes[:min(n_fed, 64)])), reps=5)
        result["host_fed"] = {
            "what": "elm_register_stream_host: every scan in page-locked HOST memory when the timed region starts; H2D upload (packed float32 xyz, "
                    "12 B/pt), device-side ordering (k_scan_order) and all ICP iterations inside the timed region, overlapped on three HIP streams",
            "value": fed_rate,
            "unit": "registrations/s",
            "registrations_per_step": n_fed,
            "slots": args.hostfed_slots,
            "steps": args.hostfed_steps,
            "timed_region_s": tf_,
            "bytes_per_registration": fed_bytes,
            "pcie_achieved_gbs": fed_rate * fed_bytes / 1e9,
            "pcie_h2d_probe_gbs": h2d_gbs,
            "pcie_h2d_probe_first_100MB_gbs": h2d_head_gbs,
            "pcie_spec_gbs": 63.0,
            "pcie_roof_registrations_per_s": h2d_gbs * 1e9 / fed_bytes,
            "frac_of_pcie_probe": fed_rate * fed_bytes / 1e9 / h2d_gbs if h2d_gbs > 0 else None,
            "bit_identical_to_resident": bool(all(np.array_equal(a["T"], b["T"]) and a["iterations"] == b["iterations"] for a, b in zip(fed, out))),
            "max_abs_pose_diff_vs_resident": float(max(np.abs(a["T"] - b["T"]).max() for a, b in zip(fed, out))),
        }

    if extras and args.slots > 0 and args.guess == "easy":
        # the harder initial-guess set of SURVEY 8(d) through the same entry point (iteration-count dependence)
        T0h = [synth.perturb(Tt, seed=3003 + i, max_trans=0.5, max_rot_deg=2.0) for i, Tt in enumerate(T_true)]
        packed_h = reg.pack_inputs(scans, T0h)
        reg.RunRegisterStream(packed_h[0], vm, packed_h[1], slots=n_slots, raw=True)
        barrier()
        t1 = time.perf_counter()
        hsteps = 3
        for _ in range(hsteps):
            outh = reg.RunRegisterStream(packed_h[0], vm, packed_h[1], slots=n_slots, raw=True)
        barrier()
        th = time.perf_counter() - t1
        if distributed:
            t = torch.tensor([th], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            th = float(t.item())
        outh = results_from_raw(outh)
        ih = np.array([r["iterations"] for r in outh])
        result["hard_guess"] = {
            "workload": f"same scans and map, initial guess 0.5 m / 2 deg, {hsteps} steps of {n_batch} registrations",
            "value": n_batch * hsteps / th,
            "unit": "registrations/s",
            "iterations_mean": float(ih.mean()),
            "iterations_max": int(ih.max()),
            "success_rate": float(np.mean([r["is_success"] for r in outh])),
        }
    else:
        T0h, outh = None, None

    if want_replica:
        # replica mode: whole registrations per GPU, no collective (the comparison SURVEY 8e asks for)
        rctx = Context(local_rank)
        rvm = VoxelHashMap(1.0, 30, rctx)
        rvm.AddPoints(world)
        if method in (IcpMethod.VGICP, IcpMethod.AVGICP):
            rvm.CalVoxelCovAll()
        if method == IcpMethod.GICP:
            rvm.CalPointCovAll(0.4)
        mine = list(range(rank, n_batch, world_size))
        rscans = [Scan(rctx, gin["kept"][i]) for i in mine]  # the scans this rank generated (whole), the same registrations as above
        rreg = Registration(cfg, rctx)
        rp = rreg.pack_inputs(rscans, [T0s[i] for i in mine])
        rreg.RunRegisterStream(rp[0], rvm, rp[1], slots=args.slots, raw=True)
        barrier(); rctx.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            rout = rreg.RunRegisterStream(rp[0], rvm, rp[1], slots=args.slots, raw=True)
        rctx.synchronize(); barrier()
        tr = time.perf_counter() - t1
        t = torch.tensor([tr], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tr = float(t.item())
        # replica results must equal the sharded ones up to the summation tree
        rout = results_from_raw(rout)
        dmax = max(float(np.abs(r["T"] - out[i]["T"]).max()) for r, i in zip(rout, mine))
        result["replica"] = {"what": "whole registrations per GPU, no collective; same registrations as `value`",
                             "value": n_batch * args.steps / tr, "unit": "registrations/s",
                             "max_abs_pose_diff_vs_sharded": dmax}
        del rscans, rvm
        rctx.close()

    # ---------------- CPU baseline + pose error vs the CPU reference (rank 0, N = 1 only) ----------------
    if rank == 0 and world_size == 1 and not args.no_cpu and args.cpu_sample > 0:
        from oracle import oracle as O
        threads = 10  # the reference's shipped max_thread (config/localization.ini:95)
        ncpu = os.cpu_count() or 1
        cpu_model = "unknown"
        phys = ncpu
        try:
            with open("/proc/cpuinfo") as f:
                txt = f.read()
            cpu_model = next(line.split(":", 1)[1].strip() for line in txt.splitlines() if line.startswith("model name"))
            cores = set()
            pid = cid = None
            for line in txt.splitlines():
                if line.startswith("physical id"):
                    pid = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    cid = line.split(":")[1].strip()
                elif not line.strip():
                    if pid is not None and cid is not None:
                        cores.add((pid, cid))
                    pid = cid = None
            if cores:
                phys = len(cores)
        except Exception:  # noqa: BLE001
            pass
        all_threads = max(1, min(phys, ncpu))
        n_s = min(args.cpu_sample, len(scans_host))

        def crop_map(i, full=False):
            """the oracle's AoS / unordered_map map over the part of the world scan i can reach: everything within the scan's
            own extent + 15 m of the sensor (correspondences look at most 2 voxels + the initial-guess error away)."""
            if full:
                pts = world
            else:
                r = float(np.sqrt((scans_host[i].astype(np.float64) ** 2).sum(axis=1).max())) + 15.0
                d = world[:, :2].astype(np.float64) - T_true[i][:2, 3]
                pts = world[(d * d).sum(axis=1) < r * r]
            om = O.Map(1.0, 30)
            om.add_points(pts)
            if method in (IcpMethod.VGICP, IcpMethod.AVGICP):
                om.cal_voxel_cov_all(threads)
            if method == IcpMethod.GICP:
                om.cal_point_cov_all(0.4, threads)
            return om

        ocfg10 = O.default_config(int(method), max_thread=min(threads, ncpu))
        ocfg_all = O.default_config(int(method), max_thread=all_threads)
        om = crop_map(0)
        for _ in range(3):  # warm-ups
            O.register(om, scans_host[0], T0s[0], ocfg10)
        t10, c10, tall, call, errs, it_match = [], [], [], [], [], []
        refs = []
        for i in range(n_s):
            if i:
                om = crop_map(i)
            ref = O.register(om, scans_host[i], T0s[i], ocfg10)
            refs.append(ref)
            t10.append(ref["elapsed_ms"] * 1e-3)
            c10.append(ref["correspondence_ms"] * 1e-3)
            errs.append(synth.pose_error(ref["T"], out[i]["T"]))
            it_match.append(ref["iterations"] == out[i]["iterations"] and ref["is_success"] == out[i]["is_success"])
            ra = O.register(om, scans_host[i], T0s[i], ocfg_all)  # every physical core for the correspondence search; accumulation stays serial
            tall.append(ra["elapsed_ms"] * 1e-3)
            call.append(ra["correspondence_ms"] * 1e-3)
            del om
        full_info = None
        if args.cpu_full_sample > 0:
            tb = time.perf_counter()
            omf = crop_map(0, full=True)
            t_build = time.perf_counter() - tb
            O.register(omf, scans_host[0], T0s[0], ocfg10)  # warm-up
            tf, same = [], True
            for i in range(min(args.cpu_full_sample, n_s)):
                rf = O.register(omf, scans_host[i], T0s[i], ocfg10)
                tf.append(rf["elapsed_ms"] * 1e-3)
                same = same and bool(np.array_equal(rf["T"], refs[i]["T"])) and rf["iterations"] == refs[i]["iterations"]
            del omf
            full_info = {"n": len(tf), "seconds_per_registration_median": float(np.median(tf)),
                         "cropped_seconds_same_scans_median": float(np.median(t10[:len(tf)])),
                         "pose_bit_identical_to_cropped_map": same, "map_build_s": t_build,
                         "what": f"the same registrations on the un-cropped {args.map_points}-point oracle map (std::unordered_map locality)"}
        med10 = float(np.median(t10))
        result["cpu_baseline"] = {
            "value": 1.0 / med10,
            "unit": "registrations/s",
            "cores": min(threads, ncpu),
            "kind": "port",
            "sample": f"{n_s} registrations of the same batch (scans 0..{n_s - 1}) after 3 warm-ups, by the CPU oracle (faithful restatement: "
                      f"168-B AoS points, std::unordered_map, {min(threads, ncpu)}-thread correspondence search, serial accumulation), span of "
                      f"reg.cpp:307-394; map = world within the scan's extent + 15 m of the sensor; value = 1 / median; host has {ncpu} logical / "
                      f"{phys} physical CPUs",
            "seconds_per_registration": med10,
            "seconds_p10": percentile(t10, 10),
            "seconds_p90": percentile(t10, 90),
            "correspondence_seconds_median": float(np.median(c10)),
            "correspondence_fraction": float(np.median(np.array(c10) / np.array(t10))),
            "cpu_model": cpu_model,
            "value_all_cores": 1.0 / float(np.median(tall)),
            "all_cores_threads": all_threads,
            "all_cores_seconds_p10": percentile(tall, 10),
            "all_cores_seconds_p90": percentile(tall, 90),
            "all_cores_correspondence_fraction": float(np.median(np.array(call) / np.array(tall))),
            "full_map": full_info,
        }
        result["pose_err_vs_cpu"] = {
            "max_trans_m": float(max(e[0] for e in errs)),
            "max_rot_rad": float(max(e[1] for e in errs)),
            "n_checked": len(errs),
            "iterations_and_flags_match": bool(all(it_match)),
            "tolerance": "1e-4 m / 1e-5 rad",
        }
        result["gpu_over_cpu"] = value / (1.0 / med10)
        if outh is not None:
            # pose parity on the hard set as well (a few: up to 10 iterations each on the CPU)
            eh, mh = [], []
            for i in range(min(4, n_s)):
                om = crop_map(i)
                ref = O.register(om, scans_host[i], T0h[i], ocfg10)
                eh.append(synth.pose_error(ref["T"], outh[i]["T"]))
                mh.append(ref["iterations"] == outh[i]["iterations"] and ref["is_success"] == outh[i]["is_success"])
                del om
            result["hard_guess"]["pose_err_vs_cpu"] = {"max_trans_m": float(max(e[0] for e in eh)), "max_rot_rad": float(max(e[1] for e in eh)),
                                                       "n_checked": len(eh), "iterations_and_flags_match": bool(all(mh))}

    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if distributed:
        ctx.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
