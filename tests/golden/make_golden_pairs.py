#!/usr/bin/env python3
"""Generates tests/golden/golden_r05_pairs.npz: the first iteration's PAIRS and the step of the reference's public calls
(GetCorrespondencePoints / GetCorrespondencesCov / GetCorrespondencesAllCov, vhm.cpp:31-206; AlignCloudsLocal / PointCov / VoxelCov,
reg.cpp:15-225) on the inputs of golden_r01.npz, as the CPU oracle computes them (see make_golden.py for what that pins and what it does
not).  Run from the repo root:  python tests/golden/make_golden_pairs.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "golden_r01.npz"))
world, scan, T0 = G["world"], G["scan"], G["T0"]
x, y, z = (scan[:, k].astype(np.float64) for k in range(3))
g = np.stack([((T0[r, 0] * x + T0[r, 1] * y) + T0[r, 2] * z) + T0[r, 3] for r in range(3)], 1)  # TransformPoints (reg.hpp:141-146)
# a few points that pair with the default target at the origin, with nothing, and far outside the map
g = np.concatenate([g, [[0.3, -0.2, 40.0], [0.3, -0.2, 3.0], [500.0, 500.0, 1.0]]])
local = np.concatenate([scan.astype(np.float64), np.zeros((3, 3))])
out = {"queries": g}
m = O.Map(1.0, 30)
m.add_points(world)
m.cal_voxel_cov_all(1)
m.cal_point_cov_all(0.4, 1)
th = 5.0
acc, tgt, _ = m.nearest_points(g, th, 1)
out["points_src"] = np.flatnonzero(acc)
out["points_tgt"] = tgt[acc]
acc_v, mean, cov = m.nearest_voxel(g, th, 1)
out["cov_src"] = np.flatnonzero(acc_v)
out["cov_mean"] = mean[acc_v]
out["cov_cov"] = cov[acc_v]
src, amean, acov = m.all_cov_pairs(g, th)
out["allcov_src"] = src
out["allcov_mean"] = amean
out["allcov_cov"] = acov
# the steps around T0 from those pairs
r = O.align_clouds_local(0, local[acc], tgt[acc], None, T0, th, O.default_config(0))
out["step_p2p"] = r["T"]; out["fit_p2p"] = np.array([r["fitness"]])
r = O.align_clouds_local(2, local[acc_v], mean[acc_v], cov[acc_v], T0, th, O.default_config(2))
out["step_vgicp"] = r["T"]; out["fit_vgicp"] = np.array([r["fitness"]])
r = O.align_clouds_local(3, local[src], amean, acov, T0, th, O.default_config(3))
out["step_avgicp"] = r["T"]; out["fit_avgicp"] = np.array([r["fitness"]])
# GICP: the matched points' own neighbourhood means / covariances (Pointcloud order of the oracle differs from the product's: the pairs are
# given by position, the targets' records by value)
pxyz, pcov, pmean = m.pointcloud()
order = {tuple(p): i for i, p in enumerate(pxyz.astype(np.float32).tolist())}
idx = np.array([order.get(tuple(t), -1) for t in tgt[acc].astype(np.float32).tolist()])
gm = np.where((idx >= 0)[:, None], pmean[np.maximum(idx, 0)], 0.0)
gc = np.where((idx >= 0)[:, None, None], pcov[np.maximum(idx, 0)], np.eye(3))
out["gicp_mean"] = gm
out["gicp_cov"] = gc
r = O.align_clouds_local(1, local[acc], gm, gc, T0, th, O.default_config(1))
out["step_gicp"] = r["T"]; out["fit_gicp"] = np.array([r["fitness"]]); out["cov_gicp"] = r["local_cov"]
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden_r05_pairs.npz"), **out)
print({k: np.asarray(v).shape for k, v in out.items()})
