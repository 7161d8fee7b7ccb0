"""BASELINE config 5 at full size: closed-loop stream (deskew of a 131 072-point raw scan + VGICP against a 10 M-point map
on the GPU, 27-state EKF update on the CPU) at 10 Hz LiDAR / 200 Hz IMU, simulated drive.  Prints one JSON line with the
per-stage latency per scan and the fraction of the 100 ms scan period used.

    python tools/stream_c5.py [--map 10000000] [--scan 131072] [--scans 40] [--ds 1.5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elimaloc_amd import synth  # noqa: E402
from elimaloc_amd.ekf import EkfAlgorithm, EkfConfig  # noqa: E402
from elimaloc_amd.pcm_matching import PcmMatching, PcmMatchingConfig  # noqa: E402
from elimaloc_amd.registration import Context, IcpMethod, RegistrationConfig  # noqa: E402
from elimaloc_amd.stream import LocalizationStream, rot_to_quat_xyzw  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map", type=int, default=10_000_000)
    ap.add_argument("--scan", type=int, default=131072)
    ap.add_argument("--scans", type=int, default=40)
    ap.add_argument("--ds", type=float, default=1.5, help="input_voxel_ds_m (loc.ini:90); 0.05 keeps almost every point")
    ap.add_argument("--method", type=int, default=2)
    ap.add_argument("--native", action="store_true", help="CallbackPointCloud as one C-ABI call (no Python between the stages)")
    a = ap.parse_args()
    world = synth.make_world(a.map, seed=1001)
    tf = np.eye(4)
    tf[:3, :3] = synth.rot_zyx(0.0, 0.01, 0.02)
    tf[:3, 3] = [1.2, 0.0, 1.6]
    cfg = PcmMatchingConfig(tf_ego_to_lidar=tf, d_input_voxel_ds_m=a.ds, registration=RegistrationConfig(icp_method=IcpMethod(a.method)))
    node = PcmMatching(cfg, Context(0))
    t_build = time.perf_counter()
    node.Init(world)
    t_build = time.perf_counter() - t_build
    st = LocalizationStream(node, EkfAlgorithm(EkfConfig()), native=a.native)
    drive = synth.Drive()
    rng = np.random.default_rng(42)
    imu_hz, t0 = 200, 500.0
    P0 = drive.ego_pose(0.0)
    stages, totals, errs, ekf_ms, imu_ms = [], [], [], [], []
    for k in range(int(a.scans * imu_hz / 10) + 1):
        t = k / imu_hz
        g, f = drive.imu(t, rng)
        if k == 2:
            st.ekf.CallbackPcmInitOdom(t0 + t, P0[:3, 3], rot_to_quat_xyzw(P0[:3, :3]))
        c0 = time.perf_counter()
        st.CallbackImu(t0 + t, g, f)
        imu_ms.append((time.perf_counter() - c0) * 1e3)
        if k > 10 and k % (imu_hz // 10) == 0:
            t_end = t - cfg.d_lidar_time_delay - 0.005
            raw, rel = drive.scan(world, a.scan, t_end, tf, seed=7000 + k, max_range=60.0)
            c0 = time.perf_counter()
            out = st.CallbackPointCloud(raw, rel, t0 + t_end + cfg.d_lidar_time_delay)
            c1 = time.perf_counter()
            if out is None:
                continue
            tm = {} if a.native else dict(node.timings_)
            tm["ekf_update_ms" if not a.native else "callback_plus_ekf_ms"] = (c1 - c0) * 1e3 - sum(tm.values())
            tm["n_source"] = out["n_source"]
            stages.append(tm)
            totals.append((c1 - c0) * 1e3)
            errs.append(synth.pose_error(drive.ego_pose(t_end), out["pose_ego"]))
    keys = [k for k in stages[0]]
    med = {k: float(np.median([s[k] for s in stages[3:]])) for k in keys}
    errs = np.array(errs)
    print(json.dumps({
        "workload": f"C5 stream: deskew({a.scan}) + {IcpMethod(a.method).name} vs {a.map}-pt map + EKF update, 10 Hz LiDAR / 200 Hz IMU",
        "native_callback": bool(a.native), "scans_ok": len(stages), "scans": st.n_scan, "input_voxel_ds_m": a.ds, "median_stage_ms": med,
        "median_scan_ms": float(np.median(totals[3:])), "max_scan_ms": float(np.max(totals[3:])),
        "period_fraction": float(np.median(totals[3:])) / 100.0, "imu_callback_ms_median": float(np.median(imu_ms)),
        "map_build_s": t_build, "truth_err_m_median": float(np.median(errs[:, 0])), "truth_err_rad_max": float(errs[:, 1].max())}))


if __name__ == "__main__":
    main()
