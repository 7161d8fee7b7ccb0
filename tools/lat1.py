#!/usr/bin/env python3
"""Resident single-registration latency (131 072-point scan vs the 10 M-point map): median / p10 / p90 wall time of RunRegisterBatch([scan])
over `--calls` calls cycling through eight scans, per method.  Environment switches (ELM_KERNEL, ELM_GRID, ELM_CHECK ...) are read by the library
at context creation: run once per setting.      python tools/lat1.py [--method 0] [--calls 80]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from elimaloc_amd import synth
from elimaloc_amd.registration import Context, VoxelHashMap, Registration, RegistrationConfig, IcpMethod, Scan
ap = argparse.ArgumentParser()
ap.add_argument("--method", type=int, default=0)
ap.add_argument("--calls", type=int, default=80)
ap.add_argument("--map-points", type=int, default=10_000_000)
a = ap.parse_args()
c = Context(0)
world = synth.make_world(a.map_points, seed=1001)
vm = VoxelHashMap(1.0, 30, c); vm.AddPoints(world)
m = IcpMethod(a.method)
if m == IcpMethod.GICP: vm.CalPointCovAll(0.4)
elif m != IcpMethod.P2P: vm.CalVoxelCovAll()
reg = Registration(RegistrationConfig(icp_method=m), c)
scans, T0s = [], []
for i in range(8):
    sc, Tt = synth.make_scan(world, 131072, seed=2002 + i)
    scans.append(Scan(c, sc)); T0s.append(synth.perturb(Tt, seed=3003 + i))
for i in range(16): reg.RunRegisterBatch([scans[i % 8]], vm, [T0s[i % 8]])
lat, its = [], []
for i in range(a.calls):
    t = time.perf_counter(); out = reg.RunRegisterBatch([scans[i % 8]], vm, [T0s[i % 8]]); lat.append(time.perf_counter() - t); its.append(out[0]["iterations"])
lat = np.array(lat) * 1e3
sw = {k: v for k, v in os.environ.items() if k.startswith("ELM_")}
print("method %d %s: median %.4f ms  p10 %.4f  p90 %.4f  iterations mean %.2f" % (a.method, sw, np.median(lat), np.percentile(lat, 10), np.percentile(lat, 90), np.mean(its)), flush=True)
