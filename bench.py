#!/usr/bin/env python3
"""bench.py -- ICP registrations/sec of the MI355X-native pcm_matching hot path.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1: launched by torch.distributed.run, one rank per GPU).
One STEP = one pass of the hot path over one batch of synthetic input: `--batch` (per GPU) full RunRegister-equivalent
registrations (initial transform -> iterate to the reference's own termination rule or max_iteration), all scans
already resident in HBM when the timed region starts.

Workload (BASELINE.json configs[1]): P2P ICP, 131072-pt synthetic scans vs a 10M-pt voxel-hashed map, defaults of
config/localization.ini.  N>1: every scan is sharded point-wise over the N GPUs (map replicated), ONE RCCL all-reduce
of the packed normal equations of the whole batch per ICP iteration; the batch grows with N (weak scaling: per-GPU
points per launch fixed).

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel k_accumulate, algorithmic bytes / hipEvent-measured
kernel time, against 8 TB/s HBM) and, at N=1, `cpu_baseline` (the CPU oracle timed on the host cores on a bounded
sample of the same workload, reference's shipped max_thread = 10).
"""
import argparse
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


def b_alg(method, C, V):
    """Algorithmic bytes per scan point per ICP iteration, compact-layout model of SURVEY.md 8(d)."""
    if method == 0:
        return 444.0 + 12.0 * C
    if method == 1:
        return 480.0 + 12.0 * C
    if method == 2:
        return 468.0 + 12.0 * V
    return 124.0 + 36.0 * V


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="registrations per GPU per step")
    ap.add_argument("--slots", type=int, default=128, help="registrations iterating concurrently per GPU (continuous batching: "
                    "finished slots take the next pending registration on the device); 0 = lockstep batch of --batch")
    ap.add_argument("--scan-points", type=int, default=131072)
    ap.add_argument("--map-points", type=int, default=10_000_000)
    ap.add_argument("--method", type=int, default=0, help="0 P2P (configs[1]), 1 GICP, 2 VGICP, 3 AVGICP")
    ap.add_argument("--cpu-sample", type=int, default=16, help="registrations timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency runs (profiling passes)")
    args = ap.parse_args()

    # stdout carries exactly ONE JSON line: everything libraries print (RCCL banners, gloo notices) goes to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # launched by torch.distributed.run (RANK set): take the collective path even for one rank, so that a 1-GPU box
    # exercises exactly the code the 8-GPU node runs
    distributed = world_size > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if args.gpus != world_size and distributed:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world_size}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="cpu:gloo,cuda:nccl", rank=rank, world_size=world_size)

    from elimaloc_amd import synth
    from elimaloc_amd.registration import (Context, VoxelHashMap, Registration, RegistrationConfig, IcpMethod, Scan)

    method = IcpMethod(args.method)
    ctx = Context(local_rank)
    if distributed:
        ids = [Context.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ctx.comm_init(rank, world_size, ids[0])

    # ---------------- synthetic inputs (identical on every rank: seeded) ----------------
    t0 = time.time()
    world = synth.make_world(args.map_points, seed=1001)
    vm = VoxelHashMap(1.0, 30, ctx)
    vm.AddPoints(world)
    if method in (IcpMethod.VGICP, IcpMethod.AVGICP):
        vm.CalVoxelCovAll()
    if method == IcpMethod.GICP:
        vm.CalPointCovAll(0.4)
    info = vm.info()
    t_map = time.time() - t0
    n_batch = args.batch * world_size  # weak scaling: per-GPU points per launch fixed
    scans_host, T_true, T0s, scans = [], [], [], []

    def gen(i):
        sc, Tt = synth.make_scan(world, args.scan_points, seed=2002 + i)
        n = sc.shape[0]
        lo, hi = n * rank // world_size, n * (rank + 1) // world_size  # contiguous shard of every scan
        return (sc if i < max(args.cpu_sample, 1) else None), Tt, synth.perturb(Tt, seed=3003 + i), np.ascontiguousarray(sc[lo:hi]), n

    synth.make_scan(world, 16, seed=1)  # builds the (cached) tile index of the world before the threads start
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=16) as pool:  # numpy releases the GIL in the heavy parts
        for full, Tt, T0, shard, n in pool.map(gen, range(n_batch)):
            if full is not None:
                scans_host.append(full)
            T_true.append(Tt)
            T0s.append(T0)
            scans.append(Scan(ctx, shard, n_total=n))
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
        from elimaloc_amd.registration import results_from_raw
        out = results_from_raw(out)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    regs = n_batch * args.steps
    value = regs / elapsed
    iters = np.array([r["iterations"] for r in out])
    pt_iters = float(sum(r["point_iterations"] for r in out))          # whole batch, all ranks (all-reduced sums)
    C = float(sum(r["n_cand_total"] for r in out)) / max(pt_iters, 1)  # candidates tested per point-iteration
    V = float(sum(r["n_occ_total"] for r in out)) / max(pt_iters, 1)   # occupied neighbour voxels per point-iteration
    bytes_unit = b_alg(int(method), C, V)
    kmode = os.environ.get("ELM_KERNEL", "cell")
    kernel_name = (f"k_accumulate_vnbr<{METHOD_NAMES[int(method)]}>" if (kmode in ("cell", "nbr") and int(method) in (2, 3)) else
                   f"k_accumulate_cell<{METHOD_NAMES[int(method)]}>" if (kmode == "cell" and int(method) in (0, 1)) else
                   f"k_accumulate_nbr<{METHOD_NAMES[int(method)]}>" if (kmode == "nbr" and int(method) in (0, 1)) else
                   f"k_accumulate_direct<{METHOD_NAMES[int(method)]}>" if kmode == "direct" else f"k_accumulate<{METHOD_NAMES[int(method)]}>")
    # dominant kernel: k_accumulate. Units one launch processes ON THIS GPU = its shard of the batch's live points.
    launches = max(prof["accumulate_launches"], 1)
    acc_ms_avg = prof["accumulate_ms"] / launches
    units_per_launch = (pt_iters / world_size) * args.steps / launches
    achieved_gbs = bytes_unit * units_per_launch / (acc_ms_avg * 1e-3) / 1e9 if acc_ms_avg > 0 else 0.0
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        try:
            pm = json.load(open(pmc_path))
            if (pm.get("method") == int(method) and pm.get("batch") == args.batch and pm.get("scan_points") == args.scan_points
                    and pm.get("kernel") == kernel_name):
                traffic = pm.get("hbm_bytes_per_unit") * units_per_launch  # measured HBM bytes per unit x this run's units
        except Exception:
            traffic = None

    # single-registration latency (B = 1), outside the timed region
    lat = []
    for _ in range(0 if args.no_latency else 5):
        t1 = time.perf_counter()
        reg.RunRegisterBatch(scans[:1], vm, T0s[:1])
        lat.append(time.perf_counter() - t1)
    latency_ms = 1e3 * float(np.median(lat)) if lat else None

    result = {
        "metric": "ICP registrations/sec, 128k-pt scan vs 10M-pt map; pose err vs CPU ref",
        "value": value,
        "unit": "registrations/s",
        "n_gpus": world_size,
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
                        f"(BASELINE configs[1] when P2P/131072/10M), localization.ini defaults, full convergence",
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
            "candidates_per_point_C": C,
            "occupied_voxels_per_point_V": V,
            "latency_ms_batch1": latency_ms,
            "map_build_s": t_map,
            "input_gen_s": t_in,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": kernel_name,
            "achieved": achieved_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS,
            "traffic": traffic,
            # the index lets the kernel touch a fraction of the bytes the reference's walk reads, so `frac` (algorithmic bytes of
            # the REFERENCE algorithm) exceeds 1; the measured HBM stream is the physical utilisation
            "measured_hbm_gbs": (traffic / (acc_ms_avg * 1e-3) / 1e9) if (traffic and acc_ms_avg > 0) else None,
            "measured_hbm_frac": (traffic / (acc_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and acc_ms_avg > 0) else None,
            "tested_candidates_per_point": float(sum(r["n_tested_total"] for r in out)) / max(pt_iters, 1),
            "bytes_per_unit": bytes_unit,
            "units_per_launch": units_per_launch,
            "avg_launch_ms": acc_ms_avg,
            "launches": prof["accumulate_launches"],
            "accumulate_ms_per_step": prof["accumulate_ms"] / args.steps,
            "solve_ms_per_step": prof["solve_ms"] / args.steps,
        },
    }

    # ---------------- CPU baseline + pose error vs the CPU reference (rank 0, N = 1 only) ----------------
    if rank == 0 and world_size == 1 and not args.no_cpu and args.cpu_sample > 0:
        from oracle import oracle as O
        threads = 10  # the reference's shipped max_thread (config/localization.ini:95)
        ncpu = os.cpu_count() or 1
        t_cpu, t_cpu_all, errs, it_match = [], [], [], []
        all_threads = min(ncpu, 128)
        cpu_model = "unknown"
        try:
            with open("/proc/cpuinfo") as f:
                cpu_model = next(line.split(":", 1)[1].strip() for line in f if line.startswith("model name"))
        except Exception:  # noqa: BLE001
            pass
        for i in range(min(args.cpu_sample, len(scans_host))):
            # the oracle's AoS/unordered_map map over the part of the world this scan can reach (75 m around the
            # sensor; the 60 m scan cannot see further, results are identical to the full map)
            Tt = T_true[i]
            near = world[np.linalg.norm(world[:, :2].astype(np.float64) - Tt[:2, 3], axis=1) < 75.0]
            om = O.Map(1.0, 30)
            om.add_points(near)
            if method in (IcpMethod.VGICP, IcpMethod.AVGICP):
                om.cal_voxel_cov_all(threads)
            if method == IcpMethod.GICP:
                om.cal_point_cov_all(0.4, threads)
            ref = O.register(om, scans_host[i], T0s[i], O.default_config(int(method), max_thread=min(threads, ncpu)))
            t_cpu.append(ref["elapsed_ms"] * 1e-3)
            dt, dr = synth.pose_error(ref["T"], out[i]["T"])
            errs.append((dt, dr))
            it_match.append(ref["iterations"] == out[i]["iterations"] and ref["is_success"] == out[i]["is_success"])
            # the same registration with every core the host offers to the correspondence search; the
            # accumulation stays serial, as in the reference
            t_cpu_all.append(O.register(om, scans_host[i], T0s[i], O.default_config(int(method), max_thread=all_threads))["elapsed_ms"] * 1e-3)
            del om
        cpu_rate = 1.0 / float(np.mean(t_cpu))
        result["cpu_baseline"] = {
            "value": cpu_rate,
            "unit": "registrations/s",
            "cores": min(threads, ncpu),
            "kind": "port",
            "sample": f"{len(t_cpu)} registrations of the same batch (scans 0..{len(t_cpu) - 1}) by the CPU oracle "
                      f"(faithful restatement: 168-B AoS points, std::unordered_map, {min(threads, ncpu)}-thread "
                      f"correspondence search, serial accumulation), span of reg.cpp:307-394, host has {ncpu} logical CPUs; "
                      f"map = world within 75 m of the sensor",
            "seconds_per_registration": float(np.mean(t_cpu)),
            "cpu_model": cpu_model,
            "value_all_cores": 1.0 / float(np.mean(t_cpu_all)),
            "all_cores_threads": all_threads,
        }
        result["pose_err_vs_cpu"] = {
            "max_trans_m": float(max(e[0] for e in errs)),
            "max_rot_rad": float(max(e[1] for e in errs)),
            "n_checked": len(errs),
            "iterations_and_flags_match": bool(all(it_match)),
            "tolerance": "1e-4 m / 1e-5 rad",
        }
        result["gpu_over_cpu"] = value / cpu_rate

    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if distributed:
        ctx.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
