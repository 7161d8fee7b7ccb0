"""Caller glue either side of the device path: host functions of the C ABI against the
oracle on CPU; the whole CallbackPointCloud / CallbackInitialPose sequence against an oracle-built pipeline on GPU."""
import numpy as np
import pytest

from elimaloc_amd import synth


def test_filter_and_downsample_match_oracle(oracle):
    from elimaloc_amd.pcm_matching import filter_points_by_distance, voxel_downsample
    rng = np.random.default_rng(1)
    xyz = (rng.normal(size=(20000, 3)) * [60, 60, 5]).astype(np.float32)
    xyz[:5] = [[100, 0, 0], [60, 80, 0], [0, 0, 100.00001], [70.7107, 70.7107, 0], [57.735, 57.735, 57.735]]  # ~ the radius
    t = rng.uniform(-0.1, 0, 20000).astype(np.float32)
    out, tout = filter_points_by_distance(xyz, t, 100.0)
    keep = oracle.filter_points_by_distance(xyz, 100.0)
    assert np.array_equal(out, xyz[keep]) and np.array_equal(tout, t[keep]) and 0 < len(keep) < len(xyz)
    ds, idx = voxel_downsample(out, 1.5)
    assert np.array_equal(idx, oracle.voxel_downsample(out, 1.5)) and np.array_equal(ds, out[idx])
    for vs in (0.05, 0.3, 1e-5):  # 1e-5: keys beyond +-2^20 take the generic (node-based) path
        _, idx = voxel_downsample(out, vs)
        assert np.array_equal(idx, oracle.voxel_downsample(out, vs))
    e, ei = voxel_downsample(np.zeros((0, 3), np.float32), 1.5)
    assert e.shape == (0, 3) and ei.size == 0


def test_filter_threshold_is_exact_at_the_boundary(oracle):
    """The root-free form of FilterPointsByDistance (s > s_max instead of sqrtf(s) > max_dist) takes the reference's decision for
    every float around the threshold, for thresholds that are and are not float squares, zero, negative, huge and NaN."""
    from elimaloc_amd.pcm_matching import filter_points_by_distance
    rng = np.random.default_rng(3)
    for max_dist in (100.0, 60.0, 0.1, 1e-3, 37.123456789, 1.0, 0.0, -1.0, 3e19, 1.8446743e19, 1e300, float("inf"), float("nan")):
        pts = []
        if max_dist == max_dist and 0 < max_dist < 1e18:
            s0 = np.float32(max_dist * max_dist)
            s_list = [s0]
            for _ in range(40):
                s_list.append(np.nextafter(s_list[-1], np.float32(np.inf), dtype=np.float32))
            lo = s0
            for _ in range(40):
                lo = np.nextafter(lo, np.float32(0), dtype=np.float32)
                s_list.append(lo)
            for s_ in s_list:  # points on one axis: x*x must reproduce the wanted float exactly often enough; add random directions too
                pts.append([np.sqrt(np.float64(s_)), 0.0, 0.0])
                d = rng.normal(size=3); d /= np.linalg.norm(d)
                pts.append(list(d * np.sqrt(np.float64(s_))))
        pts += [[0, 0, 0], [1e-20, 0, 0], [1e10, 1e10, 1e10], [np.nan, 0, 0], [3e19, 0, 0]]
        xyz = np.asarray(pts, dtype=np.float32)
        out, _ = filter_points_by_distance(xyz, np.zeros(len(xyz), np.float32), max_dist)
        keep = oracle.filter_points_by_distance(xyz, max_dist)
        assert np.array_equal(out, xyz[keep], equal_nan=True), max_dist


def test_interpolated_pose_matches_oracle(oracle):
    from elimaloc_amd.pcm_matching import get_interpolated_pose
    st = synth.make_deskew_stream(10, seed=3)
    od = st["odom"]
    for t in (od[3, 0] + 0.004, od[5, 0], od[-1, 0] + 0.03, od[0, 0] - 1.0):
        ok, T = get_interpolated_pose(od, t)
        rok, RT = oracle.get_interpolated_pose(od, t)
        assert ok == rok
        if ok:
            assert np.array_equal(T, RT) and T.dtype == np.float32
    # between two samples the translation is the linear interpolation (float32 accuracy)
    ok, T = get_interpolated_pose(od, 0.5 * (od[3, 0] + od[4, 0]))
    np.testing.assert_allclose(T[:3, 3], 0.5 * (od[3, 1:4] + od[4, 1:4]), atol=2e-4)
    # beyond the last sample: twist extrapolation, whose odom_after stamp stays 0 in the reference (dt_trans < 0)
    ok, T = get_interpolated_pose(od, od[-1, 0] + 0.03)
    assert ok and np.all(np.isfinite(T))


def test_covariance_shaping_matches_oracle(oracle):
    from elimaloc_amd.pcm_matching import shape_odom_covariance
    rng = np.random.default_rng(2)
    for k in range(6):
        A = rng.normal(size=(6, 8)); cov = A @ A.T * (1e-12 if k == 0 else 1e-3)
        if k == 1:
            cov = np.eye(6)  # non-GICP methods leave local_cov = I
        T = synth.make_pose(np.zeros((2, 3), np.float32) + 10, seed=k)
        fit = [0.01, 0.3, 0.6, 0.25, 0.05, 1.0][k]
        out = shape_odom_covariance(cov, T, fit)
        ref = oracle.shape_odom_covariance(cov, T, fit)
        assert np.array_equal(out, ref)
        assert out[:3, :3].max() <= 5.0 * max(fit, 0.25) ** 2 + 1e-12 and np.all(out[:3, 3:] == 0)
    I = shape_odom_covariance(np.eye(6), np.eye(4), 0.1)
    assert np.allclose(np.diag(I)[:3], 0.0625) and np.allclose(np.diag(I)[3:], (0.25 * np.pi / 180) ** 2)


@pytest.mark.gpu
def test_pcm_pipeline_matches_oracle_pipeline(oracle):
    """Config-5-shaped stream step (without the EKF, which stays on the CPU and is round-2 work): raw scan with per-point
    times + IMU + odometry -> filter -> deskew kernel -> pose sync -> downsample -> VGICP registration -> ego pose +
    covariance, product vs the same sequence assembled from oracle calls."""
    from elimaloc_amd.pcm_matching import PcmMatching, PcmMatchingConfig
    from elimaloc_amd.registration import Context, RegistrationConfig, IcpMethod
    world = synth.make_world(100000, seed=1001)
    ctx = Context(0)
    tf = np.eye(4); tf[:3, :3] = synth.rot_zyx(0.0, 0.01, 0.02); tf[:3, 3] = [1.2, 0.0, 1.6]
    cfg = PcmMatchingConfig(tf_ego_to_lidar=tf, registration=RegistrationConfig(icp_method=IcpMethod.VGICP))
    node = PcmMatching(cfg, ctx)
    node.Init(world)
    # a moving sensor: odometry (ego frame) along +x at 10 m/s with a slow yaw, IMU with that yaw rate
    st = synth.make_deskew_stream(30000, seed=9, stamp=2000.0)
    ego0 = np.eye(4); ego0[:3, 3] = [-8.0, 3.0, 0.3]
    od = st["odom"].copy()
    od[:, 1:4] += ego0[:3, 3]
    scan_end = st["stamp"] - cfg.d_lidar_time_delay
    rok, ego_end = oracle.get_interpolated_pose(od, scan_end)
    lidar_end = ego_end.astype(np.float64) @ tf
    # raw points: world points seen from the lidar pose at scan end, then "skewed" is not needed for parity -- the deskew
    # kernel simply gets these points with the stream's per-point times
    near = world[np.linalg.norm(world - lidar_end[:3, 3].astype(np.float32), axis=1) < 40.0]
    pick = near[np.random.default_rng(4).choice(len(near), 30000, replace=False)].astype(np.float64)
    raw = ((pick - lidar_end[:3, 3]) @ lidar_end[:3, :3]).astype(np.float32)
    imu = np.concatenate([st["imu_t"][:, None], st["imu_w"]], axis=1)
    out = node.CallbackPointCloud(raw, st["time"], st["stamp"], imu, od)
    # ---- the same sequence from oracle pieces
    keep = oracle.filter_points_by_distance(raw, cfg.d_input_max_dist)
    xyz, tt = raw[keep], st["time"][keep]
    front = float(tt[0]); s_end = scan_end; s_cur = s_end + front
    iok, itime, irot = oracle.imu_deskew_info(st["imu_t"], st["imu_w"], s_cur, s_end)
    ook, inc = oracle.odom_deskew_info(od, s_cur, s_end)
    assert iok and ook
    und = oracle.deskew_points(xyz, tt - np.float32(front), itime, irot, s_cur, s_end, inc)
    pok, sync_ego = oracle.get_interpolated_pose(od, s_end)
    src_idx = oracle.voxel_downsample(und, cfg.d_input_voxel_ds_m)
    om = oracle.Map(1.0, 30); om.add_points(world); om.cal_voxel_cov_all()
    ref = oracle.register(om, und[src_idx], sync_ego.astype(np.float64) @ tf, oracle.default_config(2))
    assert (out is not None) == ref["is_success"]
    assert out is not None
    ref_ego = ref["T"] @ np.linalg.inv(tf)
    dt, dr = synth.pose_error(ref_ego, out["pose_ego"])
    # the deskew kernel may differ from the oracle by 2e-6 m on a few points (float32 sin/cos rounding): the pose follows
    assert dt <= 1e-4 and dr <= 1e-5
    assert out["n_source"] == len(src_idx)
    np.testing.assert_allclose(out["covariance"], oracle.shape_odom_covariance(ref["local_cov"], ref_ego, ref["fitness"]),
                               rtol=1e-6, atol=1e-12)
    # ---- init-pose flow (pcm.cpp:356-447) on the raw scan
    rviz = np.eye(4); rviz[:3, :3] = ego_end[:3, :3].astype(np.float64); rviz[:3, 3] = ego_end[:3, 3].astype(np.float64) + [0.2, -0.1, 5.0]
    ip = node.CallbackInitialPose(rviz, raw)
    gok, gz = om.find_ground_height(rviz[0, 3], rviz[1, 3])
    assert gok and ip is not None and ip["ground_z"] == gz
    g = rviz.copy(); g[2, 3] = gz
    iref = oracle.register(om, raw[oracle.voxel_downsample(raw, 1.5)], g @ tf, oracle.default_config(2))
    dt, dr = synth.pose_error(iref["T"] @ np.linalg.inv(tf), ip["pose_ego"])
    assert iref["is_success"] and dt <= 1e-4 and dr <= 1e-5
    ctx.close()


@pytest.mark.gpu
def test_native_callback_equals_staged_python_callback(oracle):
    """elm_pcm_callback_point_cloud (the whole CallbackPointCloud in one C-ABI call) against the stage-by-stage Python
    mirror that test_pcm_pipeline_matches_oracle_pipeline pins to the oracle: same source size, pose within 1e-10 m."""
    from elimaloc_amd.pcm_matching import PcmMatching, PcmMatchingConfig
    from elimaloc_amd.registration import Context, RegistrationConfig, IcpMethod
    world = synth.make_world(100000, seed=1001)
    ctx = Context(0)
    tf = np.eye(4); tf[:3, :3] = synth.rot_zyx(0.0, 0.01, 0.02); tf[:3, 3] = [1.2, 0.0, 1.6]
    # 1.5 m: the shipped downsample; 0.05 m: nearly every point kept; 2e-5 m: voxel keys beyond the device table's packing
    # range -> the callback falls back to the host stages
    for method, ds in ((IcpMethod.VGICP, 1.5), (IcpMethod.GICP, 1.5), (IcpMethod.P2P, 0.05), (IcpMethod.VGICP, 2e-5)):
        cfg = PcmMatchingConfig(tf_ego_to_lidar=tf, d_input_voxel_ds_m=ds, registration=RegistrationConfig(icp_method=method))
        node = PcmMatching(cfg, ctx)
        node.Init(world)
        st = synth.make_deskew_stream(30000, seed=9, stamp=2000.0)
        od = st["odom"].copy()
        od[:, 1:4] += [-8.0, 3.0, 0.3]
        rok, ego_end = oracle.get_interpolated_pose(od, st["stamp"] - cfg.d_lidar_time_delay)
        lidar_end = ego_end.astype(np.float64) @ tf
        near = world[np.linalg.norm(world - lidar_end[:3, 3].astype(np.float32), axis=1) < 40.0]
        pick = near[np.random.default_rng(4).choice(len(near), 30000, replace=False)].astype(np.float64)
        raw = ((pick - lidar_end[:3, 3]) @ lidar_end[:3, :3]).astype(np.float32)
        imu = np.concatenate([st["imu_t"][:, None], st["imu_w"]], axis=1)
        a = node.CallbackPointCloud(raw, st["time"], st["stamp"], imu, od)
        b = node.CallbackPointCloudNative(raw, st["time"], st["stamp"], imu, od)
        assert a is not None and b is not None and a["n_source"] == b["n_source"] and a["time"] == b["time"]
        np.testing.assert_allclose(b["pose_ego"], a["pose_ego"], rtol=0, atol=1e-10)
        np.testing.assert_allclose(b["covariance"], a["covariance"], rtol=1e-8, atol=1e-14)
        # silent returns: no odometry at all -> deskew data missing -> nothing published
        assert node.CallbackPointCloudNative(raw, st["time"], st["stamp"], imu, np.zeros((0, 14))) is None
        assert node.CallbackPointCloudNative(np.zeros((0, 3), np.float32), np.zeros(0, np.float32), st["stamp"], imu, od) is None
    ctx.close()
