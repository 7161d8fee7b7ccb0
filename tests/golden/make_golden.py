#!/usr/bin/env python3
"""Generates tests/golden/golden_r01.npz: small seeded inputs and the CPU oracle's outputs for them.

The reference ships no golden vectors and cannot be built or run here, so these vectors are produced by
the oracle (oracle/elm_oracle.cpp), which is cross-checked against an independent numpy re-derivation in
tests/test_oracle.py.  They pin the oracle against regressions and give the GPU tests a fixture that does not need the
oracle at all.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from elimaloc_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

out = {}
world = synth.make_world(20000, seed=1001)
scan, T_true = synth.make_scan(world, 2048, seed=2002)
T0 = synth.perturb(T_true, seed=3003, max_trans=0.3, max_rot_deg=1.0)
out["world"] = world
out["scan"] = scan
out["T0"] = T0
out["T_true"] = T_true
m = O.Map(1.0, 30)
m.add_points(world)
pts = m.pointcloud()[0]
out["map_points_sorted"] = pts[np.lexsort(pts.T[::-1])].astype(np.float32)
key, npts, _, _ = m.voxels()
kn = np.concatenate([key, npts[:, None]], axis=1)
out["map_voxels_sorted"] = kn[np.lexsort(kn.T[::-1])]
for method, name in ((0, "p2p"), (1, "gicp"), (2, "vgicp"), (3, "avgicp")):
    mm = O.Map(1.0, 30)
    mm.add_points(world)
    if method in (2, 3):
        mm.cal_voxel_cov_all(1)
    if method == 1:
        mm.cal_point_cov_all(0.4, 1)
    r = O.register(mm, scan, T0, O.default_config(method, max_thread=1))
    out[f"{name}_T"] = r["T"]
    out[f"{name}_flags"] = np.array([r["is_success"], r["iterations"], r["gate"]], np.int64)
    out[f"{name}_fitness"] = np.array([r["fitness"]])
    out[f"{name}_local_cov"] = r["local_cov"]
    out[f"{name}_ncorr"] = np.array([it["n_corr"] for it in r["iters"]], np.int64)
    out[f"{name}_JTJ"] = np.array([it["JTJ"] for it in r["iters"]])
    out[f"{name}_JTr"] = np.array([it["JTr"] for it in r["iters"]])
    out[f"{name}_res"] = np.array([it["residual_sum"] for it in r["iters"]])
    out[f"{name}_Titer"] = np.array([it["T"] for it in r["iters"]])
# C1: P2P, termination threshold 0 -> exactly 10 iterations
mm = O.Map(1.0, 30)
mm.add_points(world)
r = O.register(mm, scan, T0, O.default_config(0, icp_termination_threshold_m=0.0, max_thread=1))
out["c1_T"] = r["T"]
out["c1_iterations"] = np.array([r["iterations"]])
# deskew
st = synth.make_deskew_stream(1024, seed=41)
front = float(st["time"][0])
scan_end = st["stamp"]; scan_cur = scan_end + front
ok_i, itime, irot = O.imu_deskew_info(st["imu_t"], st["imu_w"], scan_cur, scan_end)
ok_o, inc = O.odom_deskew_info(st["odom"], scan_cur, scan_end)
rel = st["time"] - np.float32(front)
out.update(dk_xyz=st["xyz"], dk_time=st["time"], dk_stamp=np.array([st["stamp"]]), dk_imu_t=st["imu_t"], dk_imu_w=st["imu_w"],
           dk_odom=st["odom"], dk_tab_time=itime, dk_tab_rot=irot, dk_incre=inc,
           dk_out=O.deskew_points(st["xyz"], rel, itime, irot, scan_cur, scan_end, inc))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden_r01.npz"), **out)
print("wrote golden_r01.npz with", len(out), "arrays")
