#!/usr/bin/env python3
"""developer statistics: undecided-point share (stage 2) and candidates tested per point, easy vs hard initial guesses"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from elimaloc_amd import synth
from elimaloc_amd.registration import Context, Registration, RegistrationConfig, IcpMethod, Scan, VoxelHashMap, results_from_raw
ctx = Context(0)
ctx.set_work_counters(True)
world = synth.make_world(10_000_000, seed=1001)
vm = VoxelHashMap(1.0, 30, ctx); vm.AddPoints(world)
reg = Registration(RegistrationConfig(icp_method=IcpMethod.P2P), ctx)
B = int(os.environ.get("HS_BATCH", "256"))
scans, Tts = [], []
for i in range(B):
    sc, Tt = synth.make_scan(world, 131072, seed=2002 + i)
    scans.append(Scan(ctx, sc)); Tts.append(Tt)
for name, g in (("easy", dict(max_trans=0.15, max_rot_deg=0.5)), ("hard", dict(max_trans=0.5, max_rot_deg=2.0))):
    T0s = [synth.perturb(Tt, seed=3003 + i, **g) for i, Tt in enumerate(Tts)]
    packed = reg.pack_inputs(scans, T0s)
    reg.RunRegisterStream(packed[0], vm, packed[1], slots=128, raw=True)
    ctx.set_profiling(True); ctx.get_profile(reset=True)
    t0 = time.perf_counter()
    out = results_from_raw(reg.RunRegisterStream(packed[0], vm, packed[1], slots=128, raw=True))
    el = time.perf_counter() - t0
    p = ctx.get_profile(reset=True); ctx.set_profiling(False)
    pt = sum(r["point_iterations"] for r in out)
    print(name, "reg/s %.0f" % (B / el), "iters %.2f" % np.mean([r["iterations"] for r in out]),
          "undecided share %.4f" % (sum(r["fallback_blocks"] for r in out) / pt), "tested/pt %.1f" % (sum(r["n_tested_total"] for r in out) / pt),
          "acc ms/launch %.4f launches %d" % (p["accumulate_ms"] / max(p["accumulate_launches"], 1), p["accumulate_launches"]),
          "ps/unit %.1f" % (1e9 * p["accumulate_ms"] / pt), flush=True)
    # per-iteration picture: one lockstep batch of 32 with traces off, iteration count forced
    for it in (1, 2, 3, 5):
        cfg = RegistrationConfig(icp_method=IcpMethod.P2P, max_iteration=it, icp_termination_threshold_m=0.0)
        o = Registration(cfg, ctx).RunRegisterBatch(scans[:32], vm, T0s[:32])
        pt2 = sum(r["point_iterations"] for r in o)
        print("   first %d iterations (no termination): undecided %.4f tested/pt %.1f" % (it, sum(r["fallback_blocks"] for r in o) / pt2, sum(r["n_tested_total"] for r in o) / pt2), flush=True)
