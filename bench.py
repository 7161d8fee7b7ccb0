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

Headline workload (BASELINE.json configs[1]): P2P ICP, 131072-pt synthetic scans vs a 10M-pt voxel-hashed map, defaults of
config/localization.ini.  N>1: every scan is sharded point-wise over the N GPUs (map replicated), ONE RCCL all-reduce
of the packed normal equations of the whole batch per ICP iteration; the SAME per-GPU operating point at every N (4096 registrations per
GPU through 256 slots per GPU = 16 registrations per slot; weak scaling: per-GPU points per launch fixed); rank r generates only the
scans i = r (mod N) and one all-to-all per 32 scans hands out the shards (bit-identical inputs to the one-rank generation).  With N>1 the
same registrations are also timed in replica mode (whole registrations per GPU, no collective) and reported under "replica".
The timed launches carry no instrumentation: the work counters (C, V, tested candidates) come from one untimed pass of the same step.

At N = 1 the default command also times, in the same process and under `configs`, the other single-GPU BASELINE configurations:
  C3_gicp      GICP 131072 / 10M (configs[2]; the reference's shipped default method), + vgicp / avgicp at the same sizes,
  C4_shard     VGICP, 32768-point scans (the per-rank shard of a 262144-point scan at N = 8) vs the 50M-point map, 2048 registrations
               through 256 slots: what ONE rank of `bench.py --gpus 8 --method 2 --scan-points 262144 --map-points 50000000 --batch 256
               --slots 32` launches per ICP iteration (without the exchange),
  C5_stream    deskew(131072) + VGICP + 27-state EKF update in closed loop through the node callback: ms per scan, sustained Hz,
each with its own `roofline` (from a counter pass committed AT THAT leg's operating point, else the compulsory stream) and its own
`pose_err_vs_cpu` on >= 4 registrations.  `--legs none` skips them (the profiling passes do).

Prints ONE JSON line on rank 0 with
  * `inputs`: SHA-1 of every uploaded scan + initial guess (the workload is bit-reproducible: seeded, BLAS-free),
  * `roofline`: dominant kernel and the unit that binds it -- VALU issue for the grid kernel: SQ_INSTS_VALU per SIMD-cycle of the
    committed rocprofv3 --pmc pass of this command AT THIS batch / slots (refused otherwise) x the kernel's mean issue cost
    (tools/probes/valu_probe + tools/valu_mix.py), with the hipEvent-measured launch time it belongs to; under
    `hbm` the HBM stream (measured fabric bytes/unit x this run's units per launch / launch time against 8 TB/s), the compulsory
    stream, the bytes the points REQUEST from the kernel's own structures and the SURVEY 8(d) figure of the REFERENCE's walk.  No
    fraction above 1 is ever printed: the model charges a launch at most the part of the index its scans can touch.
  * at N=1 `cpu_baseline`: the CPU oracle on the host cores, SURVEY 8(d) protocol (3 warm-ups, >= 20 timed
    registrations, median / p10 / p90, correspondence-vs-total split, 10 threads and all cores, a full-map sample),
    `pose_err_vs_cpu`, `reference_api` (RunRegister on host buffers, one call at a time) and `hard_guess` (the 0.5 m /
    2 deg initial-guess set timed the same way).
"""
import argparse
import hashlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on this pool: the host driver only supports dmabuf IPC (RCCL / hipIpc* fail with "invalid argument" otherwise).
# Exported by the image already; set here as well so that a launcher with a scrubbed environment still forms its communicator.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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
    VGICP (k_accumulate_vnbr, round 6 layout): 12 scan point + 4 dense-table word + 64 per block of four list slots (V = occupied
    neighbours: ceil(V / 4) blocks) + 64 winner record from the per-voxel table + 1.
    AVGICP: 12 + 4 + 48 per face record read (pairs) + 1."""
    index = (4.0 + 48.0) if grid else (32.0 + 64.0)
    if method == 0:
        return 16.0 + index + 12.0 * tested + 12.0 + 1.0
    if method == 1:
        return 16.0 + index + 12.0 * tested + 12.0 + 4.0 + 128.0 + 1.0
    if method == 2:
        return 12.0 + 4.0 + 64.0 * math.ceil(V / 4.0) + 64.0 + 1.0
    return 12.0 + 4.0 + 48.0 * pairs + 1.0


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
        if have < args.gpus and not (os.environ.get("ELM_BENCH_SHARED_GPU") and have >= 1):  # (the rehearsal mode: ranks share devices)
            raise SystemExit(f"bench.py --gpus {args.gpus}: this box has {have} GPU(s); one rank per GPU is the only mode "
                             f"(no oversubscription, no CPU fallback) -- refusing to run")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    sys.stdout.flush()
    sys.stderr.flush()
    os.execvpe(cmd[0], cmd, dict(os.environ))


def generate_inputs(world, args, rank, world_size, dist, guess, n_keep, keep_own, consume, stats=None):
    """The step's registrations: n_batch = batch x world_size seeded scans + initial guesses; `consume(i, shard, n)` receives, in order
    of i, THIS rank's shard of scan i (packed float32 xyz).  One rank: the whole scan in the generator's order.  W ranks: the contiguous
    part [n rank / W, n (rank + 1) / W) of the scan sorted along the ordering kernel's Hilbert curve (elimaloc_amd.dist.spatial_order:
    locality-aware sharding -- a rank holds a compact sector of the scan at full density, not a thinned copy of all of it).

    One rank: every scan is generated here.  W ranks: rank r generates only the scans i = r (mod W) -- the host work per rank does not
    grow with W -- and one all-to-all per round of 32 scans (host tensors over gloo, outside every timed region) hands every rank its
    shard of every scan; the per-scan metadata (true pose, initial guess, SHA-1, extent) is all-gathered.  Bit-identical to the
    one-rank generation (same seeds; `inputs.sha1` covers all of it).
    Returns dict(T_true, T0s, digests, rmaxs, kept = {i: full scan} for i < n_keep on rank 0 (+ this rank's own scans if keep_own))."""
    from concurrent.futures import ThreadPoolExecutor
    from elimaloc_amd import synth
    from elimaloc_amd import dist as elm_dist
    n_batch = args.batch * world_size * max(1, int(getattr(args, "group_ranks", 1)))  # (a device group: one process generates every scan whole)
    npts = args.scan_points

    def gen(i):
        K = max(1, int(getattr(args, "shard_of", 1)))
        if K > 1:  # --shard-of K: registration i is shard i mod K of a K x npts-point scan (the C4_shard leg's shape, for its counter passes)
            full, Tt = synth.make_scan(world, npts * K, seed=2002 + i, max_range=SCAN_RANGE_M, noise=SCAN_NOISE_M)
            sc = elm_dist.spatial_shards(full, K)[i % K]
        else:
            sc, Tt = synth.make_scan(world, npts, seed=2002 + i, max_range=SCAN_RANGE_M, noise=SCAN_NOISE_M)
        T0 = synth.perturb(Tt, seed=3003 + i, **guess)
        pre = os.environ.get("ELM_BENCH_POINT_ORDER", "")  # developer A/B of the CALLER's point order (not the default workload)
        if pre == "shuffle":  # no locality at all inside the ordering kernel's 2 m cells (a spinning LiDAR's firing order)
            sc = np.ascontiguousarray(sc[np.random.default_rng(4004 + i).permutation(sc.shape[0])])
        elif pre:  # rows sorted by the sensor-frame cell of that size (x-major) before the upload
            c = float(pre)
            key = np.floor(sc[:, 0] / c).astype(np.int64) * 65536 + np.floor(sc[:, 1] / c).astype(np.int64)
            sc = np.ascontiguousarray(sc[np.argsort(key, kind="stable")])
        h = hashlib.sha1(sc.tobytes())
        h.update(np.ascontiguousarray(T0).tobytes())
        rmax = float(np.sqrt((sc.astype(np.float64) ** 2).sum(axis=1).max())) if sc.shape[0] else 0.0
        # what the ranks receive their shards of (W > 1): the spatially ordered scan; `kept` and the SHA-1 stay the generator's
        so = sc[elm_dist.spatial_order(sc)] if (world_size > 1 and not os.environ.get("ELM_BENCH_PLAIN_SHARDS")) else sc
        return sc, Tt, T0, h.digest(), rmax, so

    synth.make_scan(world, 16, seed=1)  # builds the (cached) tile index of the world before the threads start
    meta = [None] * n_batch  # (T_true, T0, digest, rmax)
    kept = {}
    workers = max(2, min(16, (os.cpu_count() or 16) // max(world_size, 1)))
    with ThreadPoolExecutor(max_workers=workers) as pool:  # numpy releases the GIL in the heavy parts; make_scan calls no BLAS
        if world_size == 1:
            for i, (sc, Tt, T0, dg, rmax, _) in enumerate(pool.map(gen, range(n_batch))):
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
                t_g = time.perf_counter()
                own = list(pool.map(gen, [rank + world_size * m for m in ms]))
                t_x = time.perf_counter()
                if any(o[0].shape[0] != npts for o in own):
                    raise SystemExit("make_scan returned a scan of another size")
                # send buffer: destination-major, then scan; receive buffer: source-major, then scan
                inp = torch.from_numpy(np.concatenate([o[5][bounds[d]:bounds[d + 1]].reshape(-1) for d in range(world_size) for o in own]))
                out = torch.empty(world_size * cnt * mine * 3, dtype=torch.float32)
                dist.all_to_all_single(out, inp, [cnt * mine * 3] * world_size, [cnt * (bounds[d + 1] - bounds[d]) * 3 for d in range(world_size)])
                metas = [None] * world_size
                dist.all_gather_object(metas, [(o[1], o[2], o[3], o[4]) for o in own])
                if stats is not None:  # the launch budget (--dry-launch): where the host time of the N > 1 input path goes
                    stats["generate_s"] = stats.get("generate_s", 0.0) + (t_x - t_g)
                    stats["exchange_s"] = stats.get("exchange_s", 0.0) + (time.perf_counter() - t_x)
                    stats["bytes_sent"] = stats.get("bytes_sent", 0) + int(inp.numel()) * 4 * (world_size - 1) // world_size
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

    stats = {}
    t_all = time.perf_counter()
    g = generate_inputs(world, args, rank, world_size, dist, guess, 0, False, consume, stats)
    import resource
    who = [None] * world_size
    mine = dict(rank=rank, local_rank=local_rank, pid=os.getpid(), shard_points=shard_points[0], shard_sha1=shard_sha.hexdigest(),
                # launch budget of this rank: wall seconds generating its scans / in the all-to-all (+ metadata gather), bytes it sent, peak RSS
                budget=dict(generate_s=stats.get("generate_s"), exchange_s=stats.get("exchange_s"), total_s=time.perf_counter() - t_all,
                            bytes_sent=stats.get("bytes_sent"), peak_rss_mb=resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0,
                            host_cpus=os.cpu_count()))
    if world_size > 1:
        dist.all_gather_object(who, mine)
    else:
        who = [mine]
    return {"dry_launch": True, "metric": "ICP registrations/sec, 128k-pt scan vs 10M-pt map; pose err vs CPU ref", "value": None,
            "n_gpus": world_size, "rccl_ranks": None, "ranks": who,
            "inputs": {"sha1": hashlib.sha1(b"".join(g["digests"])).hexdigest(), "registrations": len(g["digests"])}}




# ------------------------------------------------------------------ roofline model (plain functions: tests/test_bench_model.py) --------
CLOCK_GHZ = 2.4  # MI355X_MICROARCH.md: max clock
SIMD_GCYCLES = 1024 * CLOCK_GHZ  # G SIMD-cycles/s over the chip (256 CUs x 4 SIMDs)
L1_HIT_LINES_PER_CYCLE, L1_MISS_LINES_PER_CYCLE = 1.6, 0.40  # distinct 128-byte lines a CU's vector-memory pipeline serves per cycle (tools/probes/l1_probe)
ISSUE_CYCLES = {"k_accumulate_grid<P2P>": 3.80, "k_accumulate_grid<GICP>": 3.81, "k_accumulate_vnbr<VGICP>": 3.88,
                "k_accumulate_vnbr<AVGICP>": 3.91}  # mean issue cost of the kernels' opcode mixes (tools/valu_mix.py), profiles/r06_valu_mix.txt
L2_TOTAL_BYTES = 8 * 4 * 1024 * 1024  # MI355X_MICROARCH.md: 4 MB of L2 per XCD
WORLD_PTS_PER_M2 = 27.5  # synth.make_world: ~25 ground points + the walls' share per square metre of map


def kernel_name_for(method, grid, forced=""):
    m = METHOD_NAMES[int(method)]
    if forced == "direct":  # developer switch ELM_KERNEL=direct: the plain 27-probe walk for every method
        return f"k_accumulate_direct<{m}>"
    if int(method) in (2, 3):
        return f"k_accumulate_vnbr<{m}>"
    return f"k_accumulate_{'grid' if grid else 'cell'}<{m}>"


def pmc_key(kernel_name, scan_points, map_points, guess, world="lattice"):
    """key of a committed counter pass in profiles/pmc_latest.json: the kernel, the workload sizes when they are not the headline's,
    the initial-guess set when it is the hard one, the world when it is not the lattice (tools/merge_pmc.py writes the same key)"""
    k = kernel_name
    if int(scan_points) != 131072 or int(map_points) != 10_000_000:
        k += f"@{int(scan_points)}/{int(map_points)}"
    if guess == "hard":
        k += "@hard"
    if world != "lattice":
        k += f"@{world}"
    return k


def index_touch_bound(index_bytes, units_per_launch, requested_index_bytes_per_unit, live_scans, map_points, scan_range_m=SCAN_RANGE_M, search_m=5.0, shard_of=1):
    """The UNIQUE bytes of the search index one launch has to touch at least once: the smallest of the index itself, what its points
    request, and the share of the (spatially organised) index that lies under the launch's scans -- a scan reaches scan_range_m + the search
    radius from its sensor (a Hilbert-contiguous shard of it: 1 / shard_of of that footprint), the map covers map_points /
    WORLD_PTS_PER_M2 square metres.  `index_bytes` is the caller's figure for the bytes the leg's kernel is CERTAIN to read under that
    footprint (bench.py: leg_roofline), so that the result is a lower bound of the measured traffic."""
    area = max(float(map_points) / WORLD_PTS_PER_M2, 1.0)
    # k footprints of a fraction f of the map each, placed independently: they cover 1 - (1 - f)^k ~ 1 - exp(-k f) of it (NOT min(1, k f):
    # footprints overlap long before they tile the map)
    share = 1.0 - math.exp(-float(live_scans) * math.pi * (scan_range_m + search_m) ** 2 / max(1, int(shard_of)) / area)
    once = min(float(index_bytes), float(requested_index_bytes_per_unit) * float(units_per_launch), share * float(index_bytes))
    # consecutive launches iterate the SAME scans: what of their index footprint still sits in the 8 x 4 MB of L2 from the previous launch
    # is not fetched again (FETCH_SIZE counts L2 misses) -- a lower bound must not charge it
    return max(once - L2_TOTAL_BYTES, 0.0)


def hbm_object(method, index_bytes, units_per_launch, sec, bytes_unit, bytes_ref, live_scans, map_points, traffic=None, traffic_src=None, shard_of=1):
    """The HBM stream of one accumulate launch: measured (a committed counter pass) or compulsory (scan point + partial record + GICP
    payload + the touched part of the index once), plus the requested bytes and the SURVEY 8(d) figure of the reference's walk."""
    payload = 64.0 if int(method) == 1 else 0.0  # the GICP match's compact record: one 64-byte sector per pair (128 with ELM_CHECK=full_records)
    stream_unit = 12.0 + 1.0 + payload           # scan point (packed xyz) + its share of the 256-byte partial record + payload
    requested_index_unit = max(bytes_unit - 17.0 - (132.0 if int(method) == 1 else 0.0), 0.0)
    index_once = index_touch_bound(index_bytes, units_per_launch, requested_index_unit, live_scans, map_points, shard_of=shard_of)
    compulsory_unit = stream_unit + index_once / max(units_per_launch, 1.0)
    gbs = lambda b: (b * units_per_launch / sec / 1e9) if sec > 0 else 0.0  # noqa: E731
    compulsory_gbs, requested_gbs, ref_gbs = gbs(compulsory_unit), gbs(bytes_unit), gbs(bytes_ref)
    if compulsory_gbs > 1.05 * HBM_PEAK_GBS:  # cannot be: fall back to the stream that is moved whatever the caches do
        compulsory_unit, compulsory_gbs, index_once = stream_unit, gbs(stream_unit), 0.0
    measured_gbs = (traffic / sec / 1e9) if (traffic and sec > 0) else None
    achieved_gbs = measured_gbs if measured_gbs is not None else compulsory_gbs
    return {
        "achieved": achieved_gbs,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved_gbs / HBM_PEAK_GBS,
        "achieved_is": "measured HBM traffic (committed counter pass: FETCH_SIZE + WRITE_SIZE with the gather calibration of "
                       "profiles/r03_gather_probe.txt -- an UPPER bound on DRAM bytes, FETCH_SIZE also counts Infinity-Cache hits) / hipEvent launch time"
                       if measured_gbs is not None else "compulsory HBM bytes / hipEvent launch time (no matching counter pass committed)",
        "traffic": traffic,
        "traffic_source": traffic_src,
        "compulsory_bytes_per_unit": compulsory_unit,
        "compulsory_gbs": compulsory_gbs,
        "compulsory_frac": compulsory_gbs / HBM_PEAK_GBS,
        "index_bytes_charged_per_launch": index_once,
        # what the points ask the memory hierarchy for (L1 / L2 / Infinity Cache serve most of it: the index is cache-resident)
        "requested_bytes_per_unit": bytes_unit,
        "requested_gbs": requested_gbs,
        "requested_over_hbm_peak": requested_gbs / HBM_PEAK_GBS,
        "bytes_model": "requested: scan point + index words (cell offsets / hash slot + column records) + 12 B x tested candidate slots + "
                       "winner (+ payload) + partial record; compulsory: scan point + partial record (+ GICP record) + the part of the index "
                       "under the launch's scans once; DESIGN.md section 4",
        # the SURVEY 8(d) model of the REFERENCE's 27-voxel walk (all C candidates): a speed-up-vs-model figure, not a utilisation
        "algorithmic_ref_bytes_per_unit": bytes_ref,
        "algorithmic_ref_gbs": ref_gbs,
        "algorithmic_ref_over_peak": ref_gbs / HBM_PEAK_GBS,
    }


def load_counter_pass(kernel_name, scan_points, map_points, guess, batch, slots, units_per_launch, world="lattice", shard_of=1):
    """The committed counter pass that speaks for a run: same kernel, sizes, guess set, registrations per step and slots PER GPU (the launch
    mix -- live slots per launch, draining launches -- follows from those) and, within 10 %, the same units per launch on this GPU.  A rank of
    an N-GPU run qualifies with the N = 1 pass of the same per-GPU operating point (more, smaller shards: the same units per launch)."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(path):
        return None
    try:
        pm = json.load(open(path)).get(pmc_key(kernel_name, scan_points, map_points, guess, world))
    except Exception:  # noqa: BLE001
        return None
    if not pm or pm.get("batch") != batch or pm.get("slots") != slots or pm.get("guess", "easy") != guess or pm.get("world", "lattice") != world:
        return None
    if int(pm.get("scan_points", 131072)) != int(scan_points) or int(pm.get("map_points", 10_000_000)) != int(map_points) or int(pm.get("shard_of", 1)) != int(shard_of):
        return None
    up = float(pm.get("units_per_launch_profiled", 0.0))
    if up > 0 and abs(units_per_launch - up) > 0.10 * up:
        return None
    return pm


def build_roofline(method, kernel_name, hbm, pm, units_per_launch, acc_ms_avg):
    """The binding roof of one leg.  With a matching counter pass: VALU issue = SQ_INSTS_VALU per SIMD-cycle x the kernel's mean issue cost
    (tools/probes/valu_probe: ~2.4 cycles for v_fma_f32 / v_add / v_mov, ~4.3 for packed float32, float64, med3 / min / max / shifts / cmp /
    cndmask / conversions, ~8.2 for rcp / sqrt; SQ_ACTIVE_INST_VALU ticks once per instruction whatever its class), TA busy = the
    vector-memory front end, else the HBM stream.  Without one: the compulsory HBM stream."""
    roofline = dict(hbm)
    roofline["bound"] = "hbm"
    extra = {}
    if pm:
        extra = {k: pm[k] for k in ("valu_busy", "valu_insts_per_simd_cycle", "ta_busy", "l1_line_accesses_per_cu_cycle", "l1_hit", "l2_hit", "valu_insts_per_wave",
                                    "resident_waves_per_simd", "ps_per_unit_traced", "batch", "slots", "source") if k in pm}
    vb = None
    if extra.get("valu_insts_per_simd_cycle") is not None and kernel_name in ISSUE_CYCLES:
        vb = float(extra["valu_insts_per_simd_cycle"]) * ISSUE_CYCLES[kernel_name]
    elif extra.get("valu_busy") is not None:  # a pass summarised before the probe: the x4 figure rescaled to the mix's mean cost
        vb = float(extra["valu_busy"]) / 4.0 * ISSUE_CYCLES.get(kernel_name, 4.0)
    if vb is not None:
        tb = float(extra.get("ta_busy", 0.0))
        if vb >= tb and vb > hbm["frac"]:
            roofline = {"bound": "valu_issue", "achieved": vb * SIMD_GCYCLES, "peak": SIMD_GCYCLES, "unit": "G SIMD-cycles/s (VALU issue)", "frac": vb,
                        "achieved_is": f"SQ_INSTS_VALU per SIMD-cycle ({extra.get('valu_insts_per_simd_cycle')}) x {ISSUE_CYCLES.get(kernel_name, 4.0)} cycles mean issue cost "
                                       "of this kernel's opcode mix (profiles/r06_valu_mix.txt, per-class costs measured by tools/probes/valu_probe: "
                                       "profiles/r04_valu_probe.txt); counters: committed rocprofv3 --pmc pass of this command at this batch / slots "
                                       "(profiles/); the launch time it belongs to is measured live below",
                        "frac_bounds": {"every_instruction_2.4_cycles": float(extra.get("valu_insts_per_simd_cycle", 0.0)) * 2.4,
                                        "every_instruction_4.3_cycles": float(extra.get("valu_insts_per_simd_cycle", 0.0)) * 4.3},
                        "traffic": hbm["traffic"]}
        elif tb > hbm["frac"]:
            roofline = {"bound": "vector_memory_issue", "achieved": tb * 256 * CLOCK_GHZ, "peak": 256 * CLOCK_GHZ, "unit": "G CU-cycles/s (TA-busy)", "frac": tb,
                        "achieved_is": "TA_TA_BUSY / (256 CUs x kernel cycles), committed rocprofv3 --pmc pass of this command (profiles/)",
                        "traffic": hbm["traffic"]}
        roofline["hbm"] = hbm
        roofline["valu_issue_frac"] = vb
        if extra.get("l1_line_accesses_per_cu_cycle") is not None and extra.get("l1_hit") is not None:
            # tools/probes/l1_probe (profiles/r05_l1_probe.txt): the vector-memory pipeline serves 1.6 distinct lines per CU-cycle when they hit
            # the L1 and 0.40 when they miss it, whatever the width of the load -- a model of the unit beside its busy counter
            acc_, hit_ = float(extra["l1_line_accesses_per_cu_cycle"]), float(extra["l1_hit"])
            roofline["vector_memory_model"] = {"utilisation": acc_ * (hit_ / L1_HIT_LINES_PER_CYCLE + (1.0 - hit_) / L1_MISS_LINES_PER_CYCLE),
                                               "what": f"TCP_TOTAL_CACHE_ACCESSES per CU-cycle ({acc_:.3f}) x (L1 hit {hit_:.3f} / {L1_HIT_LINES_PER_CYCLE} + miss / "
                                                       f"{L1_MISS_LINES_PER_CYCLE} lines per cycle, profiles/r05_l1_probe.txt); TA_BUSY reads {tb:.3f}"}
        if extra.get("ps_per_unit_traced") is not None and units_per_launch > 0:
            roofline["counter_pass_ps_per_unit"] = extra["ps_per_unit_traced"]
            roofline["this_run_ps_per_unit"] = 1e9 * acc_ms_avg / units_per_launch
    roofline["counters"] = extra
    return roofline


def assert_fractions(obj, where="line"):
    """`Never print a roofline above 1`: every `frac` / `compulsory_frac` of the line is checked before it is printed -- and a compulsory
    bound is a LOWER bound: measured traffic below 0.95 of it means the byte model charges something the kernel does not read."""
    if isinstance(obj, dict):
        if isinstance(obj.get("traffic"), (int, float)) and isinstance(obj.get("compulsory_gbs"), (int, float)) and isinstance(obj.get("achieved"), (int, float)) \
                and obj.get("unit") == "GB/s" and obj["traffic"] and obj["achieved"] < 0.95 * obj["compulsory_gbs"]:
            raise AssertionError(f"{where}: measured traffic ({obj['achieved']:.1f} GB/s) below the compulsory bound ({obj['compulsory_gbs']:.1f} GB/s): the bound is not a bound")
        for k, v in obj.items():
            if k in ("frac", "compulsory_frac", "valu_issue_frac") and isinstance(v, (int, float)) and v > 1.05:
                raise AssertionError(f"{where}.{k} = {v}: a utilisation above 1 means the byte / cycle model is wrong for this operating point")
            assert_fractions(v, where + "." + str(k))
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            assert_fractions(v, f"{where}[{i}]")


LINE_LIMIT = 6000  # the driver parses the line out of an ~8 KB tail of stdout (round 5's 45 KB line was not parsed): hard limit, asserted


def _num(x, digits=6):
    """numbers of the driver line: 6 significant digits (the full record keeps every bit)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float(f"{float(x):.{digits}g}")
    except (TypeError, ValueError):
        return None


def _pick(src, keys):
    return {k: _num(src[k]) for k in keys if isinstance(src, dict) and k in src and not isinstance(src[k], (dict, list))}


def compact_roofline(rf):
    """The driver line's `roofline`: the HBM stream of the dominant kernel (north_star: `achieved fraction of the HBM roofline`) --
    achieved = counter-measured fabric bytes per launch / hipEvent launch time, peak 8 TB/s -- with the unit the counter passes show busiest
    beside it as `limiter` / `limiter_frac`.  Numbers only; what every figure means is DESIGN.md section 5."""
    if not isinstance(rf, dict):
        return None
    hv = rf.get("hbm") if isinstance(rf.get("hbm"), dict) else (rf if rf.get("unit") == "GB/s" else None)
    if hv is None:  # a leg without a byte roofline (the config-5 loop: launch latency)
        return {"bound": rf.get("bound")}
    out = {"bound": "hbm", "achieved": _num(hv.get("achieved")), "peak": _num(hv.get("peak")), "unit": "GB/s", "frac": _num(hv.get("frac")),
           "hbm_frac": _num(hv.get("frac")), "traffic": _num(hv.get("traffic")),
           "traffic_measured": hv.get("traffic") is not None,
           "compulsory_bytes_per_unit": _num(hv.get("compulsory_bytes_per_unit")),
           "traffic_over_compulsory": _num(hv["traffic"] / (hv["compulsory_bytes_per_unit"] * rf["units_per_launch"]))
           if hv.get("traffic") and hv.get("compulsory_bytes_per_unit") and rf.get("units_per_launch") else None,
           "algorithmic_ref_bytes_per_unit": _num(hv.get("algorithmic_ref_bytes_per_unit")),
           "limiter": rf.get("bound") if rf.get("bound") != "hbm" else "hbm", "limiter_frac": _num(rf.get("frac")),
           "valu_issue_frac": _num(rf.get("valu_issue_frac")), "ta_busy": _num((rf.get("counters") or {}).get("ta_busy")),
           "l2_hit": _num((rf.get("counters") or {}).get("l2_hit"))}
    out.update(_pick(rf, ("kernel", "avg_launch_ms", "units_per_launch", "launches", "index_bytes", "tested_candidates_per_point",
                          "accumulate_ms_per_step", "solve_ms_per_step", "timed_region_s", "counter_pass_ps_per_unit", "this_run_ps_per_unit")))
    return out


def _compact_pose(pe):
    if not isinstance(pe, dict):
        return None
    out = _pick(pe, ("max_trans_m", "max_rot_rad", "n_checked", "iterations_and_flags_match"))
    fp = pe.get("first_iteration_pairs")
    if isinstance(fp, dict):
        out["pair_points"], out["pair_mismatches"] = fp.get("points"), fp.get("mismatches")
    return out


def _compact_leg(leg, brief=False):
    """one `configs` leg as ~200 bytes of numbers (brief: the field-world sub-legs, ~150)"""
    if not isinstance(leg, dict):
        return None
    out = {k: _num(v, 5) for k, v in _pick(leg, ("value", "iterations_mean", "vs_lattice_world") if brief else
                                           ("value", "ms_per_step", "steps", "iterations_mean", "success_rate", "sustained_hz")).items()}
    rf = compact_roofline(leg.get("roofline"))
    if rf:
        out.update({k: _num(rf[k], 4) for k in (("limiter_frac", "hbm_frac") if brief else ("limiter", "limiter_frac", "hbm_frac", "traffic_over_compulsory", "avg_launch_ms"))
                    if rf.get(k) is not None})
        if "limiter" not in out and "limiter_frac" not in out and rf.get("bound"):
            out["limiter"] = rf["bound"]
    pe = _compact_pose(leg.get("pose_err_vs_cpu"))
    if pe:
        out["pose_max_m"] = _num(pe.get("max_trans_m"), 3)
        if not brief:
            out["pose_max_rad"] = _num(pe.get("max_rot_rad"), 3)
        out["flags_match"] = pe.get("iterations_and_flags_match")
        if pe.get("pair_mismatches") is not None:
            out["pair_mismatches"] = pe["pair_mismatches"]
    return out


def driver_line(full):
    """The ONE stdout line: the contract's keys + numbers only.  Every prose string of the full record lives in
    DESIGN.md section 5; the full record goes to bench_full.json beside bench.py and to stderr."""
    line = {k: _num(full.get(k)) for k in ("metric", "value", "unit", "n_gpus", "rehearsal", "rccl_ranks", "steps", "warmup", "ms_per_step", "higher_is_better",
                                           "scaling", "vs_baseline", "dtype", "data")}
    cfg = full.get("config", {})
    line["config"] = {"workload": str(cfg.get("workload", "")).split(" (BASELINE")[0][:120]}
    line["config"].update(_pick(cfg, ("scan_points", "map_points", "guess", "world", "shard_of", "batch_per_gpu", "registrations_per_step", "slots_per_gpu", "iterations_mean", "iterations_max",
                                      "success_rate", "map_points_retained", "map_voxels", "candidates_per_point_C", "occupied_voxels_per_point_V",
                                      "latency_ms_batch1")))
    line["config"]["parallelism"] = str(cfg.get("parallelism", ""))[:120]
    line["config"]["process_model"] = cfg.get("process_model")
    line["roofline"] = compact_roofline(full.get("roofline"))
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "seconds_per_registration", "correspondence_fraction", "cpu_model", "value_all_cores",
                                          "all_cores_threads"))
        line["cpu_baseline"]["sample"] = str(cb.get("sample", "")).split(" after")[0][:80]
        if isinstance(cb.get("full_map"), dict):
            line["cpu_baseline"]["full_map_seconds_per_registration"] = _num(cb["full_map"].get("seconds_per_registration_median"))
    pe = _compact_pose(full.get("pose_err_vs_cpu"))
    if pe:
        line["pose_err_vs_cpu"] = pe
    if full.get("gpu_over_cpu") is not None:
        line["gpu_over_cpu"] = _num(full["gpu_over_cpu"])
    if isinstance(full.get("inputs"), dict):
        line["inputs_sha1"] = full["inputs"].get("sha1")
    for k, keys in (("hard_guess", ("value", "iterations_mean", "success_rate")),
                    ("host_fed", ("value", "pcie_achieved_gbs", "pcie_h2d_probe_gbs", "frac_of_pcie_probe", "bit_identical_to_resident")),
                    ("reference_api", ("registrations_per_s", "ms_per_call_median")),
                    ("replica", ("value", "max_abs_pose_diff_vs_sharded")),
                    ("single_process", ("value", "devices", "batch_per_gpu", "rccl_ranks", "max_abs_pose_diff_vs_ranks", "iterations_match", "wall_s"))):
        if isinstance(full.get(k), dict):
            line[k] = _pick(full[k], keys)
    if isinstance(full.get("single_process"), dict) and full["single_process"].get("error"):
        line["single_process"]["error"] = str(full["single_process"]["error"])[-120:]
    if isinstance(full.get("hard_guess"), dict) and isinstance(full["hard_guess"].get("pose_err_vs_cpu"), dict):
        hp = _compact_pose(full["hard_guess"]["pose_err_vs_cpu"])
        line["hard_guess"].update({"pose_max_m": hp.get("max_trans_m"), "flags_match": hp.get("iterations_and_flags_match"), "pair_mismatches": hp.get("pair_mismatches")})
    if isinstance(full.get("configs"), dict):
        legs = {}
        for name, leg in full["configs"].items():
            if name == "field_world":
                legs[name] = {m: _compact_leg(leg[m], brief=True) for m in ("P2P", "GICP", "VGICP", "AVGICP") if m in leg}
            else:
                legs[name] = _compact_leg(leg)
        line["configs"] = legs
    cc = full.get("c_caller")
    if isinstance(cc, dict):
        out = {}
        for k, v in (cc.get("register") or {}).items():
            if isinstance(v, dict) and "ms_median" in v:
                out["_".join(k.replace(",", "").split()[:3])[:40]] = _num(v["ms_median"])
        if isinstance(cc.get("config5"), dict) and "ms_per_scan_median" in cc["config5"]:
            out["config5_ms_per_scan"] = _num(cc["config5"]["ms_per_scan_median"])
        line["c_caller_ms"] = out
    if full.get("model_errors"):
        line["model_errors"] = len(full["model_errors"])
    line["process_wall_s"] = _num(full.get("process_wall_s"))
    line["full_record"] = "bench_full.json"
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:  # never break the contract with the driver: drop the optional blocks, last first
        for k in ("c_caller_ms", "configs", "reference_api", "host_fed", "hard_guess", "replica"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) < LINE_LIMIT:
                break
    assert len(text) < LINE_LIMIT, len(text)
    return text


def emit(full, path=None):
    """full record -> bench_full.json (beside bench.py) + stderr; the driver line (numbers only, < LINE_LIMIT bytes) -> stdout"""
    text = driver_line(full)
    blob = json.dumps(full)
    try:
        with open(path or os.path.join(ROOT, "bench_full.json"), "w") as f:
            f.write(blob + "\n")
    except OSError as e:  # a read-only checkout must not cost the line
        print(f"bench.py: could not write bench_full.json: {e}", file=sys.stderr)
    print(blob, file=sys.stderr, flush=True)
    print(text, flush=True)
    return text


def single_process_leg(args, n, out, timeout_s=120, devices=""):
    """The same N GPUs driven by ONE process (a device group, `bench.py --single-process`) -- the reference node's process model -- as a
    CHILD process of rank 0 after this job's own timed region, bounded by a timeout: a smaller batch at the same slots, its poses compared
    with this job's for the same registrations.  Any failure is reported in the record, never raised (the job's own line must not depend
    on it)."""
    import subprocess
    import tempfile
    d = tempfile.mkdtemp(prefix="elm_sp_")
    poses = os.path.join(d, "poses.npz")
    per_gpu = min(256, args.batch)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(n), "--single-process", "--batch", str(per_gpu), "--steps", "3", "--warmup", "1",
           "--method", str(args.method), "--scan-points", str(args.scan_points), "--map-points", str(args.map_points), "--slots", str(args.slots),
           "--guess", args.guess, "--world", args.world, "--dump-poses", poses] + (["--devices", devices] if devices else [])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                                                            "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "GROUP_WORLD_SIZE", "ROLE_NAME")}
    t0 = time.time()
    rec = {"batch_per_gpu": per_gpu, "devices": n}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            rec["error"] = (r.stderr or r.stdout)[-300:]
            rec["returncode"] = r.returncode
        else:
            line = json.loads(lines[-1])
            rec.update({"value": line.get("value"), "ms_per_step": line.get("ms_per_step"), "rccl_ranks": line.get("rccl_ranks"),
                        "process_model": (line.get("config") or {}).get("process_model"),
                        "avg_launch_ms": (line.get("roofline") or {}).get("avg_launch_ms")})
            z = np.load(poses)
            k = min(len(z["T"]), len(out))
            rec["registrations_compared"] = int(k)
            rec["max_abs_pose_diff_vs_ranks"] = float(max(np.abs(z["T"][i] - out[i]["T"]).max() for i in range(k))) if k else None
            rec["iterations_match"] = bool(all(int(z["iterations"][i]) == out[i]["iterations"] for i in range(k)))
    except Exception as e:  # noqa: BLE001
        rec["error"] = repr(e)[-300:]
    rec["wall_s"] = time.time() - t0
    return rec


def sanitize_fractions(result):
    """What main() does with assert_fractions' verdict: a model error must never cost the driver its line.  Every roofline object that breaks
    a rule (a utilisation above 1.05, measured traffic below the compulsory bound) loses the offending figures and says so; the errors are
    listed at the top level (`model_errors`).  The CPU tests call assert_fractions itself."""
    errors = []

    def walk(obj, where):
        if isinstance(obj, dict):
            try:
                assert_fractions({k: v for k, v in obj.items() if not isinstance(v, (dict, list))}, where)
            except AssertionError as e:
                errors.append(str(e)[:200])
                for k in ("frac", "compulsory_frac", "valu_issue_frac", "compulsory_gbs", "compulsory_bytes_per_unit"):
                    if k in obj and k != "frac":
                        obj[k] = None
                if isinstance(obj.get("frac"), (int, float)) and obj["frac"] > 1.05:
                    obj["frac"] = None
                obj["model_error"] = str(e)[:200]
            for k, v in obj.items():
                walk(v, where + "." + str(k))
        elif isinstance(obj, list):
            for i, v in enumerate(obj):
                walk(v, f"{where}[{i}]")
    walk(result, "line")
    if errors:
        result["model_errors"] = errors
    return errors


def host_cpu_info():
    ncpu = os.cpu_count() or 1
    cpu_model, phys = "unknown", ncpu
    try:
        with open("/proc/cpuinfo") as f:
            txt = f.read()
        cpu_model = next(line.split(":", 1)[1].strip() for line in txt.splitlines() if line.startswith("model name"))
        cores, pid, cid = set(), None, None
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
    return ncpu, phys, cpu_model


def c_caller_numbers(timeout_s=120):
    """What a plain-C caller of the C ABI sees (no Python): examples/register_latency.c (one RunRegister-equivalent call: pageable /
    page-locked / resident) and examples/stream_harness.c at config-5 size (node callback + EKF update per scan).  Compiled with gcc into a
    temporary directory and run as child processes AFTER every timed region; any failure is reported, never raised."""
    import re
    import shutil
    import subprocess
    import tempfile
    out = {"what": "examples/register_latency.c and examples/stream_harness.c: child processes, plain C, 131 072-point scans against a 9 M-point map "
                   "(ground lattice + wall; these registrations take 2 iterations)"}
    if not shutil.which("gcc"):
        out["error"] = "no gcc"
        return out
    libdir = os.path.join(ROOT, "elimaloc_amd")
    d = tempfile.mkdtemp(prefix="elm_c_")
    try:
        for src, key, env in (("register_latency.c", "register", {"ELM_LAT_CALLS": "100"}),
                              ("stream_harness.c", "config5", {"ELM_HARNESS_GRID": "3000", "ELM_HARNESS_SCAN": "131072", "ELM_HARNESS_SCANS": "60"})):
            exe = os.path.join(d, src[:-2])
            try:
                subprocess.check_call(["gcc", "-O2", "-std=c11", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", src), "-L", libdir,
                                       "-lelimaloc_hip", "-lm", "-Wl,-rpath," + libdir, "-o", exe], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=60)
                r = subprocess.run([exe], capture_output=True, text=True, timeout=timeout_s, env=dict(os.environ, **env))
                if key == "register":
                    got = {}
                    for line in r.stdout.splitlines():
                        m = re.match(r"(elm_register\w*), (.*?)\s+median ([0-9.]+) ms\s+p10 ([0-9.]+)\s+p90 ([0-9.]+)\s+\((\d+) calls, ([0-9.]+) iterations", line)
                        if m:
                            got[(m.group(1) + " " + m.group(2)).strip()] = {"ms_median": float(m.group(3)), "ms_p10": float(m.group(4)), "ms_p90": float(m.group(5)),
                                                                            "calls": int(m.group(6)), "iterations_mean": float(m.group(7))}
                    out[key] = got if got else {"error": (r.stdout + r.stderr)[-300:], "returncode": r.returncode}
                else:
                    m = re.search(r"per scan \(.*?\): mean ([0-9.]+) ms\s+median ([0-9.]+) ms\s+max ([0-9.]+) ms\s+\(n = (\d+)\)", r.stdout)
                    out[key] = ({"ms_per_scan_mean": float(m.group(1)), "ms_per_scan_median": float(m.group(2)), "ms_per_scan_max": float(m.group(3)), "scans": int(m.group(4)),
                                 "returncode": r.returncode} if m else {"error": (r.stdout + r.stderr)[-300:], "returncode": r.returncode})
            except Exception as e:  # noqa: BLE001
                out[key] = {"error": repr(e)}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="registrations per GPU per step (0 = 4096 at every N: a timed region of ~1 s at 20 steps, five draining "
                    "launches per ~110; 16 registrations per slot at the default 256 slots)")
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
    ap.add_argument("--world", choices=("lattice", "field"), default="lattice", help="lattice: the SURVEY 8(d) world (jittered planes, the headline); field: "
                    "height-field terrain + boxes + clutter + a voxel-centre lattice patch + collinear poles (synth.make_field_world)")
    ap.add_argument("--cpu-sample", type=int, default=20, help="registrations timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-full-sample", type=int, default=3, help="of those, registrations repeated on the un-cropped map (0 = skip)")
    ap.add_argument("--pose-sample", type=int, default=64, help="registrations of the headline batch whose pose, iteration count and flags are compared with the "
                    "CPU oracle (the first --cpu-sample of them are the timed baseline; the rest run with every core and are not timed)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip latency / reference-API / hard-guess / replica legs and the `configs` legs (profiling passes)")
    ap.add_argument("--no-latency", action="store_true", help="alias of --no-extras")
    ap.add_argument("--legs", default="all", help="`configs` legs timed after the headline at N = 1: all | none | comma list of gicp,vgicp,avgicp,c4,c5,field")
    ap.add_argument("--leg-steps", type=int, default=5, help="timed steps of every `configs` leg (after one warm-up step)")
    ap.add_argument("--asym-triples", type=int, default=0, help="append this many isolated collinear point triples to the world (synth.collinear_triples): "
                    "rank-1 neighbourhoods whose regularised covariance is not symmetric (layout bits 7 / 8) -- the covariance methods then "
                    "carry the antisymmetric side records")
    ap.add_argument("--shard-of", type=int, default=1, help="K > 1: every registration is shard i mod K (dist.spatial_shards) of a K x --scan-points scan: one rank's "
                    "launches of a K-GPU run as registrations of their own (the C4_shard leg's shape; N = 1 only)")
    ap.add_argument("--single-process", action="store_true", help="the N GPUs as ONE process: a device group (elm_ctx_create_multi) instead of one rank per GPU -- "
                    "the process model of the reference's node (one process calling RunRegister); same per-GPU operating point; no CPU / extras legs")
    ap.add_argument("--devices", default="", help="--single-process: comma list of device ids (default 0..N-1; an id may repeat: ranks sharing a GPU exchange "
                    "through host memory -- the one-GPU test form)")
    ap.add_argument("--dump-poses", default="", help="write the final poses / iteration counts of the last step to this .npz (rank 0)")
    ap.add_argument("--dry-launch", action="store_true", help="launcher + rendezvous + sharded input generation only, gloo, no GPU (CPU test of the N > 1 path)")
    args = ap.parse_args()
    extras = not (args.no_extras or args.no_latency)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.single_process:
        extras = False
        args.no_cpu = True
    if args.gpus > 1 and "RANK" not in os.environ and not args.single_process:
        launch_ranks(args, sys.argv[1:])  # does not return

    # stdout carries exactly ONE JSON line: everything libraries print (RCCL banners, gloo notices) goes to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    t_process = time.time()

    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    devices = [int(x) for x in args.devices.split(",") if x.strip()] if args.devices else list(range(args.gpus))
    if args.single_process:
        if world_size != 1 or len(devices) != args.gpus:
            raise SystemExit(f"--single-process: one process drives the {args.gpus} GPUs (WORLD_SIZE {world_size}, --devices {devices})")
    # --gpus IS the number of ranks, whoever launched them (the driver's torch.distributed.run for N > 1, launch_ranks above, plain
    # python for N = 1): a mismatch is an error, never a silently smaller run
    elif args.gpus != world_size:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world_size}: launch one rank per GPU (python bench.py --gpus N does it itself)")
    if args.batch <= 0:
        args.batch = 4096  # the SAME per-GPU operating point at every N (registrations per slot do not change along the scaling curve)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # launched by torch.distributed.run (RANK set): take the collective path even for one rank, so that a 1-GPU box
    # exercises exactly the code the 8-GPU node runs
    distributed = (world_size > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)) and not args.single_process
    ranks = len(devices) if args.single_process else world_size  # GPUs that share every registration
    if args.dry_launch:
        if distributed:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", rank=rank, world_size=world_size)
        line = dry_launch(args, rank, world_size, local_rank, dist)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        if rank == 0:
            print(json.dumps(line, separators=(",", ":")), flush=True)
        if distributed:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback")
    if args.single_process and max(devices) >= torch.cuda.device_count():
        raise SystemExit(f"bench.py --single-process --devices {devices}: {torch.cuda.device_count()} GPU(s) visible")
    # ELM_BENCH_SHARED_GPU=1 (developer REHEARSAL of the N > 1 flow on a box with fewer GPUs than ranks: the ranks share devices and exchange
    # their sums through torch.distributed / gloo on the host instead of RCCL, which refuses two ranks on one device).  It exercises the
    # launcher, the sharded input path, the sharded registrations, the replica and single-process legs and the line at N -- it measures
    # nothing: the line says so (`config.process_model`) and carries `rehearsal: true`.
    shared_gpu = bool(os.environ.get("ELM_BENCH_SHARED_GPU")) and world_size > 1
    if shared_gpu:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    elif torch.cuda.device_count() < world_size or local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py --gpus {args.gpus}: {torch.cuda.device_count()} GPU(s) visible to rank {rank} (local rank {local_rank}); one rank per GPU is the only mode")
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if shared_gpu else "cpu:gloo,cuda:nccl", rank=rank, world_size=world_size)

    from concurrent.futures import ThreadPoolExecutor
    from elimaloc_amd import synth
    from elimaloc_amd.registration import (Context, VoxelHashMap, Registration, RegistrationConfig, IcpMethod, Scan, PinnedBuffer,
                                           results_from_raw)

    method = IcpMethod(args.method)
    ctx = Context.multi(devices) if args.single_process else Context(local_rank)
    rccl_ranks, rank_table = None, None
    group_exchange = None
    if args.single_process:
        g_ranks, g_ex, g_dev = ctx.group_info()
        if g_ranks != len(devices):
            raise SystemExit(f"device group reports {g_ranks} ranks, expected {len(devices)}")
        group_exchange = {0: "none", 1: "rccl", 2: "host"}[g_ex]
        rccl_ranks = g_ranks if g_ex == 1 else None  # (elm_ctx_create_multi verified ncclCommCount / ncclCommUserRank of every rank's communicator)
    if distributed and shared_gpu:
        import ctypes as C_
        hip_ = C_.CDLL("libamdhip64.so")
        hip_.hipMemcpy.argtypes = [C_.c_void_p, C_.c_void_p, C_.c_size_t, C_.c_int]
        hip_.hipStreamSynchronize.argtypes = [C_.c_void_p]

        def gloo_exchange(ptr, n, hip_stream):
            if hip_.hipStreamSynchronize(C_.c_void_p(hip_stream)) != 0:
                return 1
            buf = torch.empty(n, dtype=torch.float64)
            if hip_.hipMemcpy(buf.data_ptr(), ptr, n * 8, 2) != 0:
                return 1
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            return 0 if hip_.hipMemcpy(ptr, buf.data_ptr(), n * 8, 1) == 0 else 1
        ctx.set_allreduce_hook(gloo_exchange)
    elif distributed:
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

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    def max_over_ranks(seconds):
        if not distributed:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def ensure_covariances(vmap, m):
        info_ = vmap.info()
        if m in (IcpMethod.VGICP, IcpMethod.AVGICP) and not info_.has_voxel_cov:
            vmap.CalVoxelCovAll()
        if m == IcpMethod.GICP and not info_.has_point_cov:
            vmap.CalPointCovAll(0.4)

    def time_stream(reg_, vmap, packed_, slots_, steps_, warmup_):
        """`warmup_` untimed + `steps_` timed steps of RunRegisterStream over the packed registrations: (raw results of the last step,
        seconds (max over ranks), kernel profile, the step callable)."""
        def step_():
            return reg_.RunRegisterStream(packed_[0], vmap, packed_[1], slots=slots_, raw=True)
        o = None
        for _ in range(warmup_):
            o = step_()
        ctx.set_profiling(True)
        ctx.get_profile(reset=True)
        barrier()
        t1 = time.perf_counter()
        for _ in range(steps_):
            o = step_()
        barrier()
        el = max_over_ranks(time.perf_counter() - t1)
        pr = ctx.get_profile(reset=True)
        ctx.set_profiling(False)
        return o, el, pr, step_

    def work_counters(step_, out_):
        """C / V of the reference's walk and the candidates this kernel tests, from ONE untimed pass with the counters compiled in; the pass
        must reproduce the timed poses bit for bit."""
        ctx.set_work_counters(True)
        oc = results_from_raw(step_())
        ctx.set_work_counters(False)
        if not all(np.array_equal(a["T"], b["T"]) and a["iterations"] == b["iterations"] for a, b in zip(oc, out_)):
            raise SystemExit("the instrumented pass does not reproduce the timed one")
        pt = max(float(sum(r["point_iterations"] for r in out_)), 1.0)
        return (float(sum(r["n_cand_total"] for r in oc)) / pt, float(sum(r["n_occ_total"] for r in oc)) / pt,
                float(sum(r["n_tested_total"] for r in oc)) / pt, float(sum(r["fallback_blocks"] for r in oc)) / pt)

    def leg_roofline(m, vmap, prof_, steps_, out_, cvt, op, elapsed_):
        """roofline object of one timed leg; op = dict(scan_points, map_points, guess, batch, slots): its operating point per GPU"""
        Cc, Vv, tested_, undecided_ = cvt
        info_ = vmap.info()
        pt_iters_ = float(sum(r["point_iterations"] for r in out_))  # whole batch, all ranks (all-reduced sums)
        grid_ = int(info_.nbr_entries) == int(info_.n_points)  # the dense cell grid holds every map point once (the lists: 27 times)
        bytes_ref_ = b_alg_reference(int(m), Cc, Vv)
        bytes_unit_ = kernel_bytes_model(int(m), tested_, Vv, Cc if int(m) == 3 else 0.0, grid_)
        kname = kernel_name_for(m, grid_, os.environ.get("ELM_KERNEL", ""))
        launches_ = max(prof_["accumulate_launches"], 1)
        acc_ms_ = prof_["accumulate_ms"] / launches_
        upl = (pt_iters_ / ranks) * steps_ / launches_  # units one launch processes ON THIS GPU = its shard of the batch's live points
        sec_ = acc_ms_ * 1e-3
        live_scans_ = upl / max(op["scan_points"] / ranks, 1.0)
        pm_ = load_counter_pass(kname, op["scan_points"], op["map_points"], op["guess"], op["batch"], op["slots"], upl, op.get("world", "lattice"), int(op.get("shard_of", 1)))
        traffic_ = pm_.get("hbm_bytes_per_unit") * upl if (pm_ and pm_.get("hbm_bytes_per_unit") is not None) else None
        src_ = (f"profiles/pmc_latest.json[{pmc_key(kname, op['scan_points'], op['map_points'], op['guess'], op.get('world', 'lattice'))}]: HBM bytes/unit of separate rocprofv3 --pmc passes of this "
                f"leg's command ({pm_.get('source', 'see profiles/README.md')}) x this run's units per launch; counters cannot be read in the timed run") if traffic_ else None
        # the index bytes this leg's kernel is CERTAIN to read under its scans' footprint (the compulsory bound must stay below the measured
        # traffic): P2P / GICP -- the cell grid's blocks and offsets (every block holds map points, every map point is a candidate of the
        # scan points around it); VGICP / AVGICP -- the lists of the voxels that HOLD map points (their own query list: word + ~3 blocks of
        # four slots + their record; AVGICP: word + ~3.5 face records of 48 bytes read); query voxels that are merely adjacent to the map are
        # reached only by misaligned points and are not counted
        parts_ = [int(x) for x in info_.index_part_bytes]
        if int(m) in (0, 1):
            certain_ = parts_[0] if grid_ else parts_[3]
        elif int(m) == 2:
            # blocks per list: the map's own average over ALL its query lists (the lists of voxels that hold points are longer than those of
            # the fringe: the average is on the safe side of a lower bound), at most the 3 of a fully surrounded ground voxel
            nq_ = max(int(info_.n_list_voxels), 1)
            blocks_ = min(3.0, max(1.0, (parts_[1] - 64.0 * int(info_.n_voxels)) / (64.0 * nq_)))
            certain_ = min(parts_[1], int(info_.n_voxels) * (4 + 64 * blocks_ + 64))
        else:
            certain_ = min(parts_[2] if parts_[2] else parts_[1], int(info_.n_voxels) * (4 + 48 * 3))
        part_bytes_ = (parts_[0] if grid_ else parts_[3]) if int(m) in (0, 1) else (parts_[1] if int(m) == 2 or not parts_[2] else parts_[2])
        shard_of_ = int(op.get("shard_of", 1))
        hbm_ = hbm_object(int(m), certain_, upl, sec_, bytes_unit_, bytes_ref_, live_scans_, op["map_points"], traffic_, src_, shard_of=shard_of_)
        roof = build_roofline(int(m), kname, hbm_, pm_, upl, acc_ms_)
        if pm_ and ranks > 1:
            roof["counter_pass_shape"] = (f"taken at N = 1 ({op['slots']} slots x {op['scan_points']}-point scans); this rank runs {op['slots'] * ranks} slots x "
                                          f"{op['scan_points'] // ranks}-point shards: the same per-GPU registrations, slots and units per launch")
        roof.update({
            "kernel": kname,
            "search_index": "voxel-mean lists" if int(m) in (2, 3) else ("dense cell grid" if grid_ else "neighbourhood lists"),
            "index_bytes": int(part_bytes_),            # the structures THIS leg's kernel reads (elm_map_info.index_part_bytes)
            "index_bytes_certain": int(certain_),
            "map_device_bytes": int(info_.device_bytes),
            "candidates_per_point_C": Cc,
            "occupied_voxels_per_point_V": Vv,
            "tested_candidates_per_point": tested_,
            "undecided_share_after_stage1": undecided_,  # P2P / GICP: points the per-lane 2x2x2 block could not decide (served by stage 2)
            "units_per_launch": upl,
            "live_scans_per_launch": live_scans_,
            "avg_launch_ms": acc_ms_,
            "launches": prof_["accumulate_launches"],
            "accumulate_ms_per_step": prof_["accumulate_ms"] / steps_,
            "solve_ms_per_step": prof_["solve_ms"] / steps_,
            "timed_region_s": elapsed_,
        })
        return roof

    ncpu, phys, cpu_model = host_cpu_info()
    threads = 10  # the reference's shipped max_thread (config/localization.ini:95)
    O = None
    want_cpu = rank == 0 and world_size == 1 and not args.no_cpu and args.cpu_sample > 0
    if want_cpu:
        from oracle import oracle as O  # noqa: N811  (the checker: CPU baseline + pose parity only)

    def oracle_map(pts, m, vs=1.0, cap=30):
        om_ = O.Map(vs, cap)
        om_.add_points(pts)
        if m in (IcpMethod.VGICP, IcpMethod.AVGICP):
            om_.cal_voxel_cov_all(threads)
        if m == IcpMethod.GICP:
            om_.cal_point_cov_all(0.4, threads)
        return om_

    def crop_world(wpts, scan_host, T_true_):
        """the part of the world scan i can reach: everything within the scan's own extent + 15 m of the sensor (correspondences look at
        most 2 voxels + the initial-guess error away)"""
        r = float(np.sqrt((scan_host.astype(np.float64) ** 2).sum(axis=1).max())) + 15.0
        d = wpts[:, :2].astype(np.float64) - T_true_[:2, 3]
        return wpts[(d * d).sum(axis=1) < r * r]

    _cloud_cache = {}

    def pairs_check(vmap_, m, om_, scan_h, T0_, acc_):
        """The first iteration's PAIRS, point by point: the product's search (elm_map_get_correspondences -- the QUERY instantiation of the
        kernel the registration runs, on the whole map) against the oracle's walk (reg.cpp:317-334 / vhm.cpp:31-206): the same source
        points paired, every target bit-identical.  acc_: [points, pairs, mismatching points or pairs] summed over the checked registrations."""
        try:
            _pairs_check(vmap_, m, om_, scan_h, T0_, acc_)
        except Exception as e:  # noqa: BLE001  (an auxiliary check must not cost the line its timed numbers)
            acc_[2] += 1
            acc_.append(repr(e))

    def _pairs_check(vmap_, m, om_, scan_h, T0_, acc_):
        th_ = 5.0  # RegistrationConfig::max_search_dist (localization.ini)
        x, y, z = (scan_h[:, k].astype(np.float64) for k in range(3))
        T_ = np.asarray(T0_, dtype=np.float64)
        g = np.stack([((T_[r, 0] * x + T_[r, 1] * y) + T_[r, 2] * z) + T_[r, 3] for r in range(3)], 1)  # TransformPoints (reg.hpp:141-146)
        key = (id(vmap_), int(m) in (0, 1))
        if int(m) in (0, 1):
            a_, t_, _ = om_.nearest_points(g, th_, min(threads, ncpu))
            _, si, ti = vmap_._correspondences(0, g, th_)
            if key not in _cloud_cache:
                _cloud_cache.clear()
                _cloud_cache[key] = vmap_.Pointcloud()
            tp = np.where((ti >= 0)[:, None], _cloud_cache[key][np.maximum(ti, 0)], 0.0)
            want_src, want_tgt = np.flatnonzero(a_), t_[a_]
        else:
            if int(m) == 2:
                a_, t_, _c = om_.nearest_voxel(g, th_, min(threads, ncpu))
                want_src, want_tgt = np.flatnonzero(a_), t_[a_]
            else:
                want_src, want_tgt, _c = om_.all_cov_pairs(g, th_)
            _, si, ti = vmap_._correspondences(int(m) - 1, g, th_)
            if key not in _cloud_cache:
                _cloud_cache.clear()
                _cloud_cache[key] = vmap_.Voxels()[3]
            tp = np.where((ti >= 0)[:, None], _cloud_cache[key][np.maximum(ti, 0)], 0.0)
        if np.array_equal(si, want_src):
            bad = int((tp != want_tgt).any(axis=1).sum())
        else:
            bad = int(len(np.setxor1d(si, want_src))) + 1
        acc_[0] += int(g.shape[0]); acc_[1] += int(len(want_src)); acc_[2] += bad

    def pose_check(wpts, m, scans_h, T_true_, T0_, out_, idx, vmap_=None):
        """pose_err_vs_cpu of the registrations `idx`: the CPU oracle on the same scan / guess against the part of the map the scan reaches"""
        errs_, match_ = [], []
        pairs_ = [0, 0, 0]
        ocfg_ = O.default_config(int(m), max_thread=min(threads, ncpu))
        for i in idx:
            om_ = oracle_map(crop_world(wpts, scans_h[i], T_true_[i]), m)
            ref_ = O.register(om_, scans_h[i], T0_[i], ocfg_)
            errs_.append(synth.pose_error(ref_["T"], out_[i]["T"]))
            match_.append(ref_["iterations"] == out_[i]["iterations"] and ref_["is_success"] == out_[i]["is_success"])
            if vmap_ is not None:
                pairs_check(vmap_, m, om_, scans_h[i], T0_[i], pairs_)
            del om_
        res_ = {"max_trans_m": float(max(e[0] for e in errs_)), "max_rot_rad": float(max(e[1] for e in errs_)), "n_checked": len(errs_),
                "iterations_and_flags_match": bool(all(match_)), "tolerance": "1e-4 m / 1e-5 rad"}
        if vmap_ is not None:
            res_["first_iteration_pairs"] = {"points": pairs_[0], "pairs": pairs_[1], "mismatches": pairs_[2], **({"errors": pairs_[3:]} if len(pairs_) > 3 else {}),
                                             "what": "elm_map_get_correspondences (the registration kernel's own search, pairs written out) against the oracle's walk: same source points, bit-identical targets"}
        return res_

    # ---------------- synthetic inputs (seeded, BLAS-free arithmetic: bit-identical whoever generates them) ----------------
    t0 = time.time()
    world = synth.make_world(args.map_points, seed=1001) if args.world == "lattice" else synth.make_field_world(args.map_points, seed=1001)
    if args.asym_triples > 0:
        world = np.ascontiguousarray(np.concatenate([world, synth.collinear_triples(world, args.asym_triples, seed=4004)]))
    vm = VoxelHashMap(1.0, 30, ctx)
    vm.AddPoints(world)
    ensure_covariances(vm, method)
    info = vm.info()
    t_map = time.time() - t0
    n_batch = args.batch * ranks  # weak scaling: per-GPU points per launch fixed
    n_keep = max(args.cpu_sample, 0 if args.no_cpu else min(args.pose_sample, args.batch), 8) if (rank == 0 and world_size == 1) else 0  # full host copies kept for the CPU / reference-API legs (N = 1)
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

    args.group_ranks = ranks if args.single_process else 1
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

    n_slots = args.slots * ranks  # per-GPU points per launch stay fixed as ranks are added
    packed = reg.pack_inputs(scans, T0s)  # handle array + column-major guesses, marshalled once
    op_point = dict(scan_points=args.scan_points, map_points=args.map_points, guess=args.guess, batch=args.batch, slots=args.slots, world=args.world,
                    shard_of=max(1, args.shard_of))

    if args.slots > 0:
        out_raw, elapsed, prof, step = time_stream(reg, vm, packed, n_slots, args.steps, args.warmup)
        out = results_from_raw(out_raw)
    else:  # lockstep batch (developer A/B)
        def step():
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
        elapsed = max_over_ranks(time.perf_counter() - t_start)
        prof = ctx.get_profile(reset=True)
        ctx.set_profiling(False)

    regs = n_batch * args.steps
    value = regs / elapsed
    iters = np.array([r["iterations"] for r in out])
    # The timed launches carry no instrumentation.  The work counters (candidates C / occupied voxels V of the reference's walk, candidates
    # this kernel distance-tests) come from ONE untimed pass of the same step with the counters compiled in: same poses, bit for bit.
    if args.slots > 0:
        cvt = work_counters(step, out)
    else:
        ctx.set_work_counters(True)
        out_c = step()
        ctx.set_work_counters(False)
        pt = max(float(sum(r["point_iterations"] for r in out)), 1.0)
        cvt = (float(sum(r["n_cand_total"] for r in out_c)) / pt, float(sum(r["n_occ_total"] for r in out_c)) / pt, float(sum(r["n_tested_total"] for r in out_c)) / pt,
               float(sum(r["fallback_blocks"] for r in out_c)) / pt)
    info = vm.info()
    roofline = leg_roofline(method, vm, prof, args.steps, out, cvt, op_point, elapsed)

    result = {
        "metric": "ICP registrations/sec, 128k-pt scan vs 10M-pt map; pose err vs CPU ref",
        "value": value,
        "unit": "registrations/s",
        "n_gpus": ranks,
        "rehearsal": bool(shared_gpu),
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
                        f"initial guess {guess['max_trans']} m / {guess['max_rot_deg']} deg ({args.guess})" + ("" if args.world == "lattice" else f", world `{args.world}`"),
            "scan_points": args.scan_points,
            "map_points": args.map_points,
            "guess": args.guess, "world": args.world, "shard_of": max(1, args.shard_of),
            "batch_per_gpu": args.batch,
            "registrations_per_step": n_batch,
            "slots_per_gpu": args.slots,
            "registrations_per_slot": (args.batch / args.slots) if args.slots > 0 else None,  # the same at every N: the scaling curve compares one operating point
            "scheduling": ("continuous batching: every ICP iteration is one launch over the slots, finished slots are refilled on "
                           "the device from the step's queue" if args.slots > 0 else "lockstep batch"),
            "parallelism": "1 GPU" if ranks == 1 else f"scan points sharded over {ranks} GPUs, map replicated, "
                           "one RCCL all-reduce (32 doubles/scan) per ICP iteration",
            "process_model": (f"one process, device group of {ranks} ({group_exchange} exchange), devices {devices}" if args.single_process else
                              ("REHEARSAL: ranks share GPUs, sums exchanged over gloo on the host" if shared_gpu else
                               ("one process per GPU" if ranks > 1 else "one process, one GPU"))),
            "iterations_mean": float(iters.mean()),
            "iterations_min": int(iters.min()),
            "iterations_max": int(iters.max()),
            "success_rate": float(np.mean([r["is_success"] for r in out])),
            "map_points_retained": int(info.n_points),
            "map_voxels": int(info.n_voxels),
            "map_layout_flags": int(info.layout_flags),  # include/elimaloc_hip.h; bits 7 / 8: the map holds asymmetric covariances (side records)
            "candidates_per_point_C": cvt[0],
            "occupied_voxels_per_point_V": cvt[1],
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
        hsteps = 3
        outh_raw, th, _, _ = time_stream(reg, vm, packed_h, n_slots, hsteps, 1)
        outh = results_from_raw(outh_raw)
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
        tr = max_over_ranks(time.perf_counter() - t1)
        # replica results must equal the sharded ones up to the summation tree
        rout = results_from_raw(rout)
        dmax = max(float(np.abs(r["T"] - out[i]["T"]).max()) for r, i in zip(rout, mine))
        result["replica"] = {"what": "whole registrations per GPU, no collective; same registrations as `value`",
                             "value": n_batch * args.steps / tr, "unit": "registrations/s",
                             "max_abs_pose_diff_vs_sharded": dmax}
        del rscans, rvm
        rctx.close()

    # ---------------- CPU baseline + pose error vs the CPU reference (rank 0, N = 1 only) ----------------
    if want_cpu:
        all_threads = max(1, min(phys, ncpu))
        n_s = min(args.cpu_sample, len(scans_host))

        def crop_map(i, full=False):
            return oracle_map(world if full else crop_world(world, scans_host[i], T_true[i]), method)

        ocfg10 = O.default_config(int(method), max_thread=min(threads, ncpu))
        ocfg_all = O.default_config(int(method), max_thread=all_threads)
        om = crop_map(0)
        for _ in range(3):  # warm-ups
            O.register(om, scans_host[0], T0s[0], ocfg10)
        t10, c10, tall, call, errs, it_match = [], [], [], [], [], []
        head_pairs = [0, 0, 0]
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
            if i < 8:  # (after the timed runs of this registration: the oracle reports its own elapsed time)
                pairs_check(vm, method, om, scans_host[i], T0s[i], head_pairs)
            del om
        # pose parity on more of the batch than the baseline times (every core; these runs are not part of `cpu_baseline`)
        for i in range(n_s, min(args.pose_sample, len(scans_host)) if n_s else 0):
            om_i = crop_map(i)
            ref = O.register(om_i, scans_host[i], T0s[i], ocfg_all)
            errs.append(synth.pose_error(ref["T"], out[i]["T"]))
            it_match.append(ref["iterations"] == out[i]["iterations"] and ref["is_success"] == out[i]["is_success"])
            del om_i
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
            "first_iteration_pairs": {"points": head_pairs[0], "pairs": head_pairs[1], "mismatches": head_pairs[2], **({"errors": head_pairs[3:]} if len(head_pairs) > 3 else {}),
                                      "what": "elm_map_get_correspondences (the registration kernel's own search, pairs written out) against the oracle's walk on the first 8 checked registrations: same source points, bit-identical targets"},
        }
        result["gpu_over_cpu"] = value / (1.0 / med10)
        if outh is not None:
            # pose parity on the hard set as well (a few: up to 10 iterations each on the CPU)
            result["hard_guess"]["pose_err_vs_cpu"] = pose_check(world, method, scans_host, T_true, T0h, outh, list(range(min(4, n_s))), vm)

    # ---------------- the other single-GPU BASELINE configurations, timed in the same process (N = 1) ----------------
    legs = [] if (args.legs == "none" or not extras) else (["gicp", "vgicp", "avgicp", "c5", "c4", "field"] if args.legs == "all" else [x for x in args.legs.split(",") if x])
    headline_shape = (world_size == 1 and not distributed and args.slots > 0 and int(method) == 0 and args.scan_points == 131072
                      and args.map_points == 10_000_000 and args.guess == "easy" and args.world == "lattice")
    if legs and headline_shape:
        configs = {}
        n_chk = list(range(min(4, len(scans_host)))) if want_cpu else []

        def method_leg(m, vmap, packed_, n_regs, wpts, scans_h, Ttrue_, T0_, op, what):
            tl = time.time()
            ensure_covariances(vmap, m)
            reg_m = Registration(RegistrationConfig(icp_method=m), ctx)
            o_raw, el, pr, st_ = time_stream(reg_m, vmap, packed_, op["slots"], args.leg_steps, 1)
            o = results_from_raw(o_raw)
            cv = work_counters(st_, o)
            it = np.array([r["iterations"] for r in o])
            leg = {
                "workload": what,
                "value": n_regs * args.leg_steps / el,
                "unit": "registrations/s",
                "steps": args.leg_steps, "warmup": 1, "ms_per_step": 1e3 * el / args.leg_steps,
                "registrations_per_step": n_regs, "slots": op["slots"],
                "iterations_mean": float(it.mean()), "iterations_max": int(it.max()),
                "success_rate": float(np.mean([r["is_success"] for r in o])),
                "map_layout_flags": int(vmap.info().layout_flags),
                "roofline": leg_roofline(m, vmap, pr, args.leg_steps, o, cv, op, el),
            }
            if want_cpu and scans_h:
                leg["pose_err_vs_cpu"] = pose_check(wpts, m, scans_h, Ttrue_, T0_, o, list(range(min(4, len(scans_h)))), vmap)
            leg["leg_wall_s"] = time.time() - tl
            return leg

        for name, m in (("gicp", IcpMethod.GICP), ("vgicp", IcpMethod.VGICP), ("avgicp", IcpMethod.AVGICP)):
            if name in legs:
                key = "C3_gicp" if name == "gicp" else name
                configs[key] = method_leg(m, vm, packed, n_batch, world, scans_host, T_true, T0s, op_point,
                                          f"{METHOD_NAMES[int(m)]} ICP, the headline's scans and map (131072-pt scans vs 10000000-pt map"
                                          + ("; BASELINE configs[2], the reference's shipped icp_method" if name == "gicp" else "") + ")")

        if "c5" in legs:
            configs["C5_stream"] = c5_stream_leg(ctx, world, O if want_cpu else None, threads)

        if "field" in legs:
            # a second world: are the kernels over-fitted to the jittered planes?  Height-field terrain, boxes, clutter,
            # a patch of exact voxel-centre lattice, collinear poles, density falling with range; 1024 registrations per method
            tl = time.time()
            fworld = synth.make_field_world(args.map_points, seed=1001)
            fvm = VoxelHashMap(1.0, 30, ctx)
            fvm.AddPoints(fworld)
            n_f = 1024
            pool_n = max(2, min(16, ncpu))

            def fgen(i):
                sc, Tt = synth.make_scan(fworld, args.scan_points, seed=2002 + i, max_range=SCAN_RANGE_M, noise=SCAN_NOISE_M)
                return sc, Tt, synth.perturb(Tt, seed=3003 + i, **guess)
            synth.make_scan(fworld, 16, seed=1)
            with ThreadPoolExecutor(max_workers=pool_n) as pool:
                fg = list(pool.map(fgen, range(n_f)))
            fscans = [Scan(ctx, g[0]) for g in fg]
            fT, fT0 = [g[1] for g in fg], [g[2] for g in fg]
            fhost = [g[0] for g in fg[:4]]
            fpacked = reg.pack_inputs(fscans, fT0)
            fop = dict(scan_points=args.scan_points, map_points=args.map_points, guess=args.guess, batch=n_f, slots=args.slots, world="field")
            fl = {"what": "synth.make_field_world: height-field terrain + boxes + Poisson clutter + a patch of exact voxel-centre lattice + collinear poles, "
                          f"density falling with range from the centre, negative coordinates; {n_f} registrations of {args.scan_points}-point scans through {args.slots} slots",
                  "map_points_retained": int(fvm.info().n_points), "map_voxels": int(fvm.info().n_voxels)}
            for m in (IcpMethod.P2P, IcpMethod.GICP, IcpMethod.VGICP, IcpMethod.AVGICP):
                leg = method_leg(m, fvm, fpacked, n_f, fworld, fhost, fT, fT0, fop, f"{METHOD_NAMES[int(m)]} on the field world")
                # the same 1024 registrations on the lattice world = the like-for-like comparison (the headline's 4096 drain less)
                lat_raw, lat_el, _, _ = time_stream(Registration(RegistrationConfig(icp_method=m), ctx), vm, reg.pack_inputs(scans[:n_f], T0s[:n_f]), args.slots, args.leg_steps, 1)
                leg["lattice_world_same_shape"] = n_f * args.leg_steps / lat_el
                leg["vs_lattice_world"] = leg["value"] / leg["lattice_world_same_shape"]
                # registrations/s = iterations/s over iterations per registration: the second factor belongs to the method and the world (the
                # oracle needs the same count), the first to the kernels
                lat_iters = float(np.mean([r["iterations"] for r in results_from_raw(lat_raw)]))
                leg["lattice_world_iterations_mean"] = lat_iters
                leg["iteration_rate_vs_lattice_world"] = (leg["value"] * leg["iterations_mean"]) / (leg["lattice_world_same_shape"] * lat_iters)
                fl[METHOD_NAMES[int(m)]] = leg
            fl["leg_wall_s"] = time.time() - tl
            configs["field_world"] = fl
            del fscans, fvm, fworld, fg

        if "c4" in legs:
            tl = time.time()
            n4, pts4, map4, slots4 = 2048, 32768, 50_000_000, 256
            del packed, scans
            world4 = synth.make_world(map4, seed=1001)
            vm4 = VoxelHashMap(1.0, 30, ctx)
            vm4.AddPoints(world4)
            pool_n = max(2, min(16, ncpu))

            from elimaloc_amd import dist as elm_dist
            plain4 = bool(os.environ.get("ELM_BENCH_PLAIN_SHARDS"))  # developer A/B: rounds 2-5's shape (a 32768-point scan of the whole footprint)

            def gen4(i):
                if plain4:
                    sc, Tt = synth.make_scan(world4, pts4, seed=2002 + i, max_range=SCAN_RANGE_M, noise=SCAN_NOISE_M)
                else:  # shard i mod 8 of a 262144-point scan, cut the way `bench.py --gpus 8` cuts it (dist.spatial_shards)
                    full, Tt = synth.make_scan(world4, 8 * pts4, seed=2002 + i, max_range=SCAN_RANGE_M, noise=SCAN_NOISE_M)
                    sc = elm_dist.spatial_shards(full, 8)[i % 8]
                return sc, Tt, synth.perturb(Tt, seed=3003 + i, **guess)
            synth.make_scan(world4, 16, seed=1)
            with ThreadPoolExecutor(max_workers=pool_n) as pool:
                g4 = list(pool.map(gen4, range(n4)))
            scans4 = [Scan(ctx, g[0]) for g in g4]
            T4, T04 = [g[1] for g in g4], [g[2] for g in g4]
            host4 = [g[0] for g in g4[:4]]
            op4 = dict(scan_points=pts4, map_points=map4, guess=args.guess, batch=n4, slots=slots4, world="lattice", shard_of=1 if plain4 else 8)
            leg = method_leg(IcpMethod.VGICP, vm4, reg.pack_inputs(scans4, T04), n4, world4, host4, T4, T04, op4,
                             f"VGICP, {pts4}-point shards (one of the 8 Hilbert-contiguous shards of BASELINE configs[3]'s 262144-point scans, shard i mod 8 of "
                             f"scan i) vs the {map4}-point map, {n4} registrations through {slots4} slots: the launches ONE rank of the 8-GPU run issues per ICP "
                             "iteration, without the exchange (there all 8 ranks hold a shard of the same registration; here every shard is a registration of "
                             "its own, so the value is also the whole-job rate 8 such ranks would reach if the collective were free)")
            leg["shards"] = "plain" if plain4 else "spatial"
            leg["setup_s"] = time.time() - tl - leg["leg_wall_s"]
            configs["C4_shard"] = leg
            del scans4, vm4, world4, g4
        result["configs"] = configs
        if legs and want_cpu and args.world == "lattice" and not os.environ.get("ELM_BENCH_NO_C_CALLER"):
            # the drop-in numbers as a C caller sees them (child processes, after every timed region of this one)
            tcc = time.time()
            result["c_caller"] = c_caller_numbers()
            result["c_caller"]["wall_s"] = time.time() - tcc
    if distributed and (world_size > 1 or os.environ.get("ELM_BENCH_FORCE_SINGLE_PROCESS")) and extras and not os.environ.get("ELM_BENCH_NO_SINGLE_PROCESS"):
        # both process models on the same GPUs in one driver run: the N ranks wait while rank 0's child drives all N devices by itself
        if rank == 0 and time.time() - t_process > 240.0:  # (the job's own line comes first: no extra leg in a run that is already long)
            result["single_process"] = {"skipped": "process wall above 240 s"}
        elif rank == 0:
            result["single_process"] = single_process_leg(args, world_size, out, devices=",".join(str(r % torch.cuda.device_count()) for r in range(world_size)) if shared_gpu else "")
        dist.barrier()
    result["process_wall_s"] = time.time() - t_process

    if args.dump_poses and rank == 0:
        np.savez(args.dump_poses, T=np.stack([r["T"] for r in out]), iterations=np.array([r["iterations"] for r in out]),
                 success=np.array([r["is_success"] for r in out]))
    for e_ in sanitize_fractions(result):
        print("bench.py: roofline model error: " + e_, file=sys.stderr)
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    if rank == 0:
        emit(result)
    if distributed:
        ctx.comm_destroy()
        dist.destroy_process_group()


def c5_stream_leg(ctx, world, O, threads, n_scans=30, n_pts=131072):
    """BASELINE configs[4]: deskew(131072) + VGICP vs the 10 M-point map + 27-state EKF update at 10 Hz LiDAR / 200 Hz IMU in closed loop
    through the node callback (elm_pcm_callback_point_cloud: ONE C-ABI call per scan) -- pcm.cpp:198-324 + ekfl.cpp:147-220.  ms per scan
    (callback + EKF update, host wall clock), the rate the loop could sustain, and the poses of four scans against the CPU oracle chain
    (deskew -> pose sync -> downsample -> register) fed the same inputs."""
    from elimaloc_amd import synth
    from elimaloc_amd.ekf import EkfAlgorithm, EkfConfig
    from elimaloc_amd.pcm_matching import PcmMatching, PcmMatchingConfig
    from elimaloc_amd.registration import IcpMethod, RegistrationConfig
    from elimaloc_amd.stream import LocalizationStream, rot_to_quat_xyzw
    tl = time.time()
    tf = np.eye(4)
    tf[:3, :3] = synth.rot_zyx(0.0, 0.01, 0.02)
    tf[:3, 3] = [1.2, 0.0, 1.6]
    cfg = PcmMatchingConfig(tf_ego_to_lidar=tf, registration=RegistrationConfig(icp_method=IcpMethod.VGICP))
    node = PcmMatching(cfg, ctx)
    node.Init(world)
    st = LocalizationStream(node, EkfAlgorithm(EkfConfig()), native=True)
    drive = synth.Drive()
    rng = np.random.default_rng(42)
    imu_hz, t0 = 200, 500.0
    P0 = drive.ego_pose(0.0)
    # the scans' candidate points: within 75 m of the drive's start (the drive moves a few metres, the scans reach 60 m)
    x0, y0, *_ = drive.at(0.0)
    d = world[:, :2].astype(np.float64) - np.array([float(x0), float(y0)])
    near = np.ascontiguousarray(world[(d * d).sum(axis=1) < 75.0 * 75.0])
    totals, errs, nsrc, kept = [], [], [], []
    for k in range(int(n_scans * imu_hz / 10) + 1):
        t = k / imu_hz
        g, f = drive.imu(t, rng)
        if k == 2:
            st.ekf.CallbackPcmInitOdom(t0 + t, P0[:3, 3], rot_to_quat_xyzw(P0[:3, :3]))
        st.CallbackImu(t0 + t, g, f)
        if k > 10 and k % (imu_hz // 10) == 0:
            t_end = t - cfg.d_lidar_time_delay - 0.005
            raw, rel = drive.scan(near, n_pts, t_end, tf, seed=7000 + k, max_range=60.0)
            stamp = t0 + t_end + cfg.d_lidar_time_delay
            imu_w, odom_w = st._windows(stamp) if st.deq_odom_ else (None, None)
            c0 = time.perf_counter()
            out = st.CallbackPointCloud(raw, rel, stamp)
            c1 = time.perf_counter()
            if out is None:
                continue
            totals.append((c1 - c0) * 1e3)
            nsrc.append(out["n_source"])
            errs.append(synth.pose_error(drive.ego_pose(t_end), out["pose_ego"]))
            if O is not None and len(totals) in (4, 11, 18, 25):
                kept.append((raw, rel, stamp, imu_w.copy(), odom_w.copy(), out["pose_lidar"].copy()))
    errs = np.array(errs)
    med = float(np.median(totals[3:]))
    leg = {
        "workload": f"deskew({n_pts}) + VGICP vs the {world.shape[0]}-pt map + 27-state EKF update, closed loop, 10 Hz LiDAR / 200 Hz IMU, shipped "
                    f"input_voxel_ds_m = 1.5 (the registration sees {int(np.median(nsrc))} points per scan), {len(totals)} of {st.n_scan} scans published",
        "value": med, "unit": "ms per scan (callback + EKF update)", "higher_is_better": False,
        "max_scan_ms": float(np.max(totals[3:])),
        "sustained_hz": 1000.0 / med, "required_hz": 10.0, "period_fraction": med / 100.0,
        "truth_err_m_median": float(np.median(errs[:, 0])), "truth_err_rad_max": float(errs[:, 1].max()),
        "roofline": {"bound": "launch latency", "note": "per scan: k_deskew 8 us, k_ds_* 27 us, one or two (accumulate 9 us + solve 8 us) iterations over ~10 k points "
                                                        "(profiles/r02_c5_kernel_stats.csv): microseconds of kernel time per 100 ms period -- no roofline applies; the "
                                                        "number that matters is the period fraction"},
    }
    if O is not None and kept:
        om = O.Map(cfg.d_pcm_voxel_size, cfg.i_pcm_voxel_max_point)
        d = world[:, :2].astype(np.float64) - np.array([float(x0), float(y0)])
        om.add_points(world[(d * d).sum(axis=1) < 90.0 * 90.0])
        om.cal_voxel_cov_all(threads)
        es, match = [], []
        for raw, rel, stamp, imu_w, od, pose_lidar in kept:
            scan_end = stamp - cfg.d_lidar_time_delay
            keep = O.filter_points_by_distance(raw, cfg.d_input_max_dist)
            xyz, tt = raw[keep], rel[keep]
            front = float(tt[0])
            s_cur = scan_end + front
            iok, itime, irot = O.imu_deskew_info(imu_w[:, 0].copy(), imu_w[:, 1:].copy(), s_cur, scan_end)
            ook, inc = O.odom_deskew_info(od, s_cur, scan_end)
            und = O.deskew_points(xyz, tt - np.float32(front), itime, irot, s_cur, scan_end, inc)
            pok, sync_ego = O.get_interpolated_pose(od, scan_end)
            src = und[O.voxel_downsample(und, cfg.d_input_voxel_ds_m)]
            ref = O.register(om, src, sync_ego.astype(np.float64) @ cfg.tf_ego_to_lidar, O.default_config(2))
            es.append(synth.pose_error(ref["T"], pose_lidar))
            match.append(bool(iok and ook and pok and ref["is_success"]))
        leg["pose_err_vs_cpu"] = {"max_trans_m": float(max(e[0] for e in es)), "max_rot_rad": float(max(e[1] for e in es)), "n_checked": len(es),
                                  "iterations_and_flags_match": bool(all(match)), "tolerance": "1e-4 m / 1e-5 rad",
                                  "what": "the oracle chain (deskew -> pose sync -> VoxelDownsample -> register) on the inputs of four of the loop's callbacks"}
    leg["leg_wall_s"] = time.time() - tl
    return leg


if __name__ == "__main__":
    main()
