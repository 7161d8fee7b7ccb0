#!/usr/bin/env python3
"""Developer micro-bench of the accumulate kernel: prints per-step accumulate/solve ms, fall-back fraction."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from elimaloc_amd import synth
from elimaloc_amd.registration import Context, VoxelHashMap, Registration, RegistrationConfig, IcpMethod, Scan

ap = argparse.ArgumentParser()
ap.add_argument("--method", type=int, default=0)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--scan-points", type=int, default=131072)
ap.add_argument("--map-points", type=int, default=10_000_000)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--term", type=float, default=0.02)
ap.add_argument("--slots", type=int, default=0, help=">0: continuous batching (elm_register_stream) with that many slots")
a = ap.parse_args()
ctx = Context(0)
ctx.set_work_counters(True)
world = synth.make_world(a.map_points, seed=1001)
vm = VoxelHashMap(1.0, 30, ctx); vm.AddPoints(world)
m = IcpMethod(a.method)
if m in (IcpMethod.VGICP, IcpMethod.AVGICP): vm.CalVoxelCovAll()
if m == IcpMethod.GICP: vm.CalPointCovAll(0.4)
scans, T0s = [], []
for i in range(a.batch):
    sc, Tt = synth.make_scan(world, a.scan_points, seed=2002 + i)
    scans.append(Scan(ctx, sc)); T0s.append(synth.perturb(Tt, seed=3003 + i))
reg = Registration(RegistrationConfig(icp_method=m, max_iteration=a.iters, icp_termination_threshold_m=a.term), ctx)
run = (lambda: reg.RunRegisterStream(scans, vm, T0s, slots=a.slots)) if a.slots > 0 else (lambda: reg.RunRegisterBatch(scans, vm, T0s))
for _ in range(2): out = run()
ctx.set_profiling(True); ctx.get_profile(reset=True)
t0 = time.perf_counter()
for _ in range(a.steps): out = run()
el = time.perf_counter() - t0
p = ctx.get_profile()
pt_it = sum(r["point_iterations"] for r in out)
blocks = pt_it / 256.0
fb = sum(r["fallback_blocks"] for r in out)
print(f"kernel={os.environ.get('ELM_KERNEL','cell')} method={m.name} B={a.batch} iters={[r['iterations'] for r in out]}")
print(f"  step {1e3*el/a.steps:.3f} ms  accumulate {p['accumulate_ms']/a.steps:.3f} ms  solve {p['solve_ms']/a.steps:.3f} ms  "
      f"ns/point-iter {1e6*p['accumulate_ms']/a.steps/pt_it:.3f}  us/scan-iter(131072) {1e3*p['accumulate_ms']/a.steps/pt_it*131072:.1f}  "
      f"fallback {fb:.0f}/{blocks:.0f} = {fb/blocks:.3f}  C={sum(r['n_cand_total'] for r in out)/pt_it:.1f} tested={sum(r['n_tested_total'] for r in out)/pt_it:.1f}")
