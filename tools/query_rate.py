#!/usr/bin/env python3
"""Developer tool (MI355X): what the correspondence / step calls cost as calls of their own -- elm_map_get_correspondences and
elm_align_clouds_local on a 131 072-point scan against the 10 M-point bench map (host buffers in and out, one call at a time).
    python tools/query_rate.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elimaloc_amd import synth  # noqa: E402
from elimaloc_amd.registration import Context, VoxelHashMap, Registration, RegistrationConfig, IcpMethod  # noqa: E402

ctx = Context(0)
world = synth.make_world(10_000_000, seed=1001)
vm = VoxelHashMap(1.0, 30, ctx)
vm.AddPoints(world)
vm.CalVoxelCovAll()
vm.BuildNeighbourhoods()
scan, Tt = synth.make_scan(world, 131072, seed=2002)
T0 = synth.perturb(Tt, seed=3003)
g = scan.astype(np.float64) @ T0[:3, :3].T + T0[:3, 3]
for what, name in ((0, "GetCorrespondencePoints"), (1, "GetCorrespondencesCov"), (2, "GetCorrespondencesAllCov")):
    vm._correspondences(what, g, 5.0)
    t = []
    for _ in range(20):
        t0 = time.perf_counter()
        q, si, ti = vm._correspondences(what, g, 5.0)
        t.append(time.perf_counter() - t0)
    print(f"{name}: {len(si)} pairs of {len(g)} points, {1e3 * np.median(t):.3f} ms per call (host arrays in, index arrays out)")
reg = Registration(RegistrationConfig(icp_method=IcpMethod.VGICP), ctx)
_, tm, tc, si, ti = vm.GetCorrespondencesCov(g, 5.0, indices=True)
local = scan.astype(np.float64)[si]
reg.AlignCloudsLocalVoxelCov(local, tm, tc, T0, 5.0)
t = []
for _ in range(20):
    t0 = time.perf_counter()
    reg.AlignCloudsLocalVoxelCov(local, tm, tc, T0, 5.0)
    t.append(time.perf_counter() - t0)
print(f"AlignCloudsLocalVoxelCov: {len(si)} pairs, {1e3 * np.median(t):.3f} ms per call")
t = []
for _ in range(20):
    t0 = time.perf_counter()
    reg.AlignCloudsLocal(local, tm, T0, 5.0)
    t.append(time.perf_counter() - t0)
print(f"AlignCloudsLocal: {len(si)} pairs, {1e3 * np.median(t):.3f} ms per call")
