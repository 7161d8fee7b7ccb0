import os, sys
sys.path.insert(0, '.')
import numpy as np
from elimaloc_amd import synth
from elimaloc_amd.registration import Context, Registration, RegistrationConfig, IcpMethod, Scan, VoxelHashMap
ctx = Context(0)
ctx.set_work_counters(True)
world = synth.make_world(10_000_000, seed=1001)
vm = VoxelHashMap(1.0, 30, ctx); vm.AddPoints(world)
scans, Tts = [], []
for i in range(32):
    sc, Tt = synth.make_scan(world, 131072, seed=2002 + i)
    scans.append(Scan(ctx, sc)); Tts.append(Tt)
for name, g in (("easy", dict(max_trans=0.15, max_rot_deg=0.5)), ("hard", dict(max_trans=0.5, max_rot_deg=2.0))):
    T0s = [synth.perturb(Tt, seed=3003 + i, **g) for i, Tt in enumerate(Tts)]
    for it in (1, 2, 3, 10):
        cfg = RegistrationConfig(icp_method=IcpMethod.P2P, max_iteration=it)
        o = Registration(cfg, ctx).RunRegisterBatch(scans, vm, T0s)
        pt = sum(r["point_iterations"] for r in o)
        print(name, "first %2d iterations: undecided %.4f  of which stage-1 block empty %.4f (share of all points %.4f)  tested/pt %.1f" % (
            it, sum(r["fallback_blocks"] for r in o) / pt, sum(r["n_occ_total"] for r in o) / max(sum(r["fallback_blocks"] for r in o), 1),
            sum(r["n_occ_total"] for r in o) / pt, sum(r["n_tested_total"] for r in o) / pt), flush=True)
