"""CPU EKF counterpart: csrc/elm_ekf.cpp against the independent numpy re-derivation tests/np_ekf.py.

Runs without a GPU: the EKF is plain host code inside libelimaloc_hip.so (no device calls).  Tolerances: the two differ
only in floating-point association order (K(HP) vs (KH)P, Gauss-Jordan vs LAPACK inverse), so 1e-9 relative on state
and covariance over a several-thousand-step stream is a tight bar.
"""
import ctypes as C
import math

import numpy as np
import pytest

from elimaloc_amd import _lib
from elimaloc_amd.ekf import EkfAlgorithm, EkfConfig, GnssSource

import np_ekf


def cfg_dict(cfg):
    return {n: getattr(cfg.c, n) for n, _ in _lib.EkfConfig._fields_}


def compare(e, r, tol=1e-9):
    s = e.State()
    x = s["x"]
    ref = np.concatenate([r.pos, np.zeros(3), r.vel, r.gyro, r.acc, r.bg, r.ba, r.grav, np.zeros(3)])
    np.testing.assert_allclose(x, ref, rtol=tol, atol=tol)
    q = s["rot_xyzw"]
    np.testing.assert_allclose([q[3], q[0], q[1], q[2]], r.rot, rtol=0, atol=tol)
    q = s["imu_rot_xyzw"]
    np.testing.assert_allclose([q[3], q[0], q[1], q[2]], r.imu_rot, rtol=0, atol=tol)
    np.testing.assert_allclose(s["P"], r.P, rtol=tol, atol=1e-12)
    assert (s["state_initialized"], s["yaw_initialized"], s["rotation_stabilized"], s["state_stabilized"], s["pcm_init_on_going"]) == (
        r.state_init, r.yaw_init, r.rot_stab, r.state_stab, r.pcm_init)


def trajectory(t):
    """Smooth planar drive with gentle roll/pitch: returns position, ZYX euler, body gyro, body specific force."""
    yaw = 0.3 * math.sin(0.2 * t)
    pitch = 0.02 * math.sin(0.5 * t)
    roll = 0.015 * math.cos(0.4 * t)
    return yaw, pitch, roll


def euler_quat_xyzw(roll, pitch, yaw):
    q = np_ekf.quat_mul(np_ekf.quat_mul(np_ekf.aa_quat(yaw, [0, 0, 1]), np_ekf.aa_quat(pitch, [0, 1, 0])), np_ekf.aa_quat(roll, [1, 0, 0]))
    return np.array([q[1], q[2], q[3], q[0]])


def drive(n_sec=12.0, imu_hz=200, pcm_hz=10, seed=7, use_ckf=1, via_odom=True, lag=0.03):
    """Init -> PCM_INIT -> IMU predictions + PCM pose updates; yields after every call for comparison."""
    rng = np.random.default_rng(seed)
    cfg = EkfConfig(use_complementary_filter=use_ckf)
    e, r = EkfAlgorithm(cfg), np_ekf.NpEkf(cfg_dict(cfg))
    dt = 1.0 / imu_hz
    speed = 8.0
    pos = np.array([3.0, -2.0, 0.5])
    t0 = 100.0
    steps = int(n_sec * imu_hz)
    hist = []
    for k in range(steps):
        t = t0 + k * dt
        yaw, pitch, roll = trajectory(t - t0)
        R = np_ekf.quat_R(np.array([euler_quat_xyzw(roll, pitch, yaw)[3], *euler_quat_xyzw(roll, pitch, yaw)[:3]]))
        vel = R @ np.array([speed, 0, 0])
        pos = pos + vel * dt
        hist.append((t, pos.copy(), roll, pitch, yaw))
        y2, p2, r2 = trajectory(t - t0 + dt)
        gyro = np.array([(r2 - roll) / dt, (p2 - pitch) / dt, (y2 - yaw) / dt]) + rng.normal(0, 1e-3, 3)
        acc = R.T @ np.array([0, 0, 9.81]) + np.array([0, speed * gyro[2], 0]) + rng.normal(0, 1e-2, 3)
        if k == 3:
            q0 = euler_quat_xyzw(roll, pitch, yaw)
            assert e.CallbackPcmInitOdom(t, pos, q0) and r.update_pcm_odom(t, pos, q0, np.eye(6) * 1e-9, np_ekf.PCM_INIT)
            yield e, r
        a, b = e.RunPredictionImu(t, gyro, acc), r.predict_imu(t, gyro, acc)
        assert a == b
        ego, ego_r = e.GetCurrentState(), r.publish()
        for key, v in ego_r.items():
            assert abs(ego[key] - v) <= 1e-9 * max(1.0, abs(v)), (key, ego[key], v)
        yield e, r
        if k > 3 and k % (imu_hz // pcm_hz) == 0:
            # the ICP pose arrives `lag` seconds late and carries the scan's stamp
            kk = max(0, k - int(lag * imu_hz))
            ts, ps, rs, pis, ys = hist[kk]
            meas_p = ps + rng.normal(0, 0.02, 3)
            meas_q = euler_quat_xyzw(rs + rng.normal(0, 1e-3), pis + rng.normal(0, 1e-3), ys + rng.normal(0, 1e-3))
            cov = np.diag([4e-4, 4e-4, 4e-4, 1e-6, 1e-6, 1e-6]) + 1e-7
            if via_odom:
                a, b = e.CallbackPcmOdom(ts, meas_p, meas_q, cov), r.update_pcm_odom(ts, meas_p, meas_q, cov, np_ekf.PCM)
            else:
                a = e.RunGnssUpdate(ts, meas_p, meas_q, cov[:3, :3], cov[3:, 3:], GnssSource.PCM)
                b = r.update_pose(ts, meas_p, meas_q, cov[:3, :3], cov[3:, 3:], np_ekf.PCM)
            assert a == b
            yield e, r


@pytest.mark.parametrize("use_ckf,via_odom", [(1, True), (0, True), (1, False)])
def test_stream_matches_numpy(use_ckf, via_odom):
    n = 0
    for e, r in drive(use_ckf=use_ckf, via_odom=via_odom):
        n += 1
        if n % 7 == 0 or n < 50:
            compare(e, r)
    compare(e, r)
    s = e.State()
    assert s["state_initialized"] and not s["pcm_init_on_going"]
    # the filter tracks the simulated drive: final position within a few cm of the last measurement
    assert n > 2000


def test_filter_converges_to_truth():
    last = None
    for e, r in drive(n_sec=10.0):
        last = e
    s = last.State()
    assert math.sqrt(s["P"][0, 0]) < 0.05 and math.sqrt(s["P"][5, 5]) < 0.2 * math.pi / 180
    assert s["rotation_stabilized"] and s["state_stabilized"]
    assert abs(np.linalg.norm(s["x"][6:9]) - 8.0) < 0.3  # speed recovered from pose updates + IMU


def test_init_state_and_flags():
    e = EkfAlgorithm()
    s = e.State()
    c = e.cfg
    assert s["x"][0] == c.ekf_init_x_m and s["x"][23] == c.imu_gravity
    d = np.diag(s["P"])
    assert (d[:15] == 100.0).all() and (d[15:] == 1e-4).all()
    assert not s["state_initialized"]
    # first IMU call only latches the timestamp (b_reset_for_init_prediction_)
    assert e.RunPredictionImu(5.0, [0, 0, 0], [0, 0, 9.81]) is False
    # not initialised -> no prediction, state untouched
    assert e.RunPredictionImu(5.005, [0, 0, 0.1], [0, 0, 9.81]) is False
    np.testing.assert_array_equal(e.State()["x"], s["x"])


def test_pcm_init_resets_and_blocks_prediction():
    e = EkfAlgorithm()
    e.RunPredictionImu(1.0, [0, 0, 0], [0, 0, 9.81])
    assert e.CallbackPcmInitOdom(1.0, [10, 20, 30], [0, 0, math.sin(0.25), math.cos(0.25)])
    s = e.State()
    np.testing.assert_array_equal(s["x"][:3], [10, 20, 30])
    assert s["pcm_init_on_going"] and s["state_initialized"] and s["yaw_initialized"]
    assert e.RunPredictionImu(1.005, [0, 0, 0], [0, 0, 9.81]) is False  # PCM init on going
    # 12 PCM updates end the warm-up (count > 10 checked before the increment)
    for i in range(12):
        assert e.State()["pcm_init_on_going"]
        e.RunGnssUpdate(1.1 + 0.1 * i, [10, 20, 30], [0, 0, math.sin(0.25), math.cos(0.25)], np.eye(3) * 1e-3, np.eye(3) * 1e-5)
    assert not e.State()["pcm_init_on_going"]


def test_position_only_sources_leave_attitude():
    cfg = EkfConfig()
    e, r = EkfAlgorithm(cfg), np_ekf.NpEkf(cfg_dict(cfg))
    q = [0, 0, 0, 1]
    for src in (GnssSource.NAVSATFIX, GnssSource.BESTPOS, GnssSource.NOVATEL):
        assert e.RunGnssUpdate(2.0, [1, 2, 3], q, np.eye(3) * 0.01, np.eye(3) * 1e-4, src)
        assert r.update_pose(2.0, [1, 2, 3], q, np.eye(3) * 0.01, np.eye(3) * 1e-4, int(src))
        compare(e, r)


def test_time_compensation_rejects_old_and_empty():
    e = EkfAlgorithm()
    cov = np.eye(6) * 1e-4
    assert e.CallbackPcmOdom(1.0, [0, 0, 0], [0, 0, 0, 1], cov) is False  # empty state history
    e.RunPredictionImu(10.0, [0, 0, 0], [0, 0, 9.81])
    e.GetCurrentState()
    assert e.CallbackPcmOdom(9.0, [0, 0, 0], [0, 0, 0, 1], cov) is False  # older than the whole history


def test_constant_velocity_prediction_and_can_updates():
    """use_imu = 0 / use_can = 1 (ekf_algorithm.cpp:81-165, 434-506, 567-587): timer-driven CV prediction, CAN speed + yaw rate updates
    (scale factor, rotated measurement covariance, yaw-rate bias learnt at standstill by ZuptCan), PCM pose updates in between."""
    rng = np.random.default_rng(11)
    cfg = EkfConfig(can_vel_scale_factor=1.02, ekf_can_meas_uncertainty_vel_mps=0.5, ekf_can_meas_uncertainty_yaw_rate_deg=2.0)
    e, r = EkfAlgorithm(cfg), np_ekf.NpEkf(cfg_dict(cfg))
    t0, n_can_zupt = 50.0, 0
    q0 = euler_quat_xyzw(0.01, -0.02, 0.7)
    assert e.RunPrediction(t0) is False and r.predict(t0) is False  # the first call only latches the time
    assert e.CallbackPcmInitOdom(t0, [5, 6, 0.3], q0) and r.update_pcm_odom(t0, [5, 6, 0.3], q0, np.eye(6) * 1e-9, np_ekf.PCM_INIT)
    assert e.RunPrediction(t0 + 0.01) is False and r.predict(t0 + 0.01) is False  # PCM initialisation on going
    pos = np.array([5.0, 6.0, 0.3])
    for k in range(1, 1500):
        t = t0 + 0.01 * k
        speed = 0.0 if k > 1100 else 6.0  # the vehicle stops at the end: ZuptCan
        yaw = 0.7 + 0.1 * math.sin(0.01 * k)
        pos = pos + 0.01 * speed * np.array([math.cos(yaw), math.sin(yaw), 0.0])
        if k % 10 == 0:
            q = euler_quat_xyzw(0.01, -0.02, yaw)
            cov = np.diag([4e-4, 4e-4, 4e-4, 1e-6, 1e-6, 1e-6])
            mp = pos + rng.normal(0, 0.01, 3)
            assert e.RunGnssUpdate(t, mp, q, cov[:3, :3], cov[3:, 3:], GnssSource.PCM) == r.update_pose(t, mp, q, cov[:3, :3], cov[3:, 3:], np_ekf.PCM)
        a, b = e.RunPrediction(t), r.predict(t)
        assert a == b
        if k % 2 == 0:
            vel = [speed + rng.normal(0, 0.02) if speed else 0.0, 0.0, 0.0]
            gyro = [0.0, 0.0, 0.1 * 0.01 * math.cos(0.01 * k) / 0.01 + 0.004]  # a constant yaw-rate bias of 4 mrad/s
            a, b = e.RunCanUpdate(t, vel, gyro), r.update_can(t, vel, gyro)
            assert a == b and a
            t_can = t
            n_can_zupt += speed == 0.0
        if k % 5 == 0 or k < 40:
            compare(e, r)
        ego, ego_r = e.GetCurrentState(), r.publish()
        for key, v in ego_r.items():
            assert abs(ego[key] - v) <= 1e-9 * max(1.0, abs(v)), (key, ego[key], v)
    compare(e, r)
    assert n_can_zupt > 100 and abs(r.can_bias) > 1e-4  # the bias estimator ran
    assert e.RunCanUpdate(t_can + 0.005, [0, 0, 0], [0, 0, 0]) is False and r.update_can(t_can + 0.005, [0, 0, 0], [0, 0, 0]) is False  # closer than 10 ms to the previous message: ignored
    s = e.State()
    assert np.linalg.norm(s["x"][6:9]) < 0.2  # standstill (6 m/s before)


def test_zupt_and_mount_calibration_modes():
    """use_zupt = 1 and imu_estimate_calibration = 1 (ekf_algorithm.cpp:508-565, 703-776) on the simulated drive, then parked."""
    rng = np.random.default_rng(5)
    cfg = EkfConfig(use_zupt=1, imu_estimate_calibration=1)
    e, r = EkfAlgorithm(cfg), np_ekf.NpEkf(cfg_dict(cfg))
    dt, t0 = 0.005, 10.0
    pos = np.array([0.0, 0.0, 0.2])
    imu_bias_g, imu_bias_a = np.array([2e-3, -1e-3, 3e-3]), np.array([0.02, -0.03, 0.01])
    calibrated = zupted = 0
    for k in range(6000):
        t = t0 + k * dt
        speed = 8.0 if k < 3600 else 0.0
        yaw = 0.2 * math.sin(0.002 * k) if k < 3600 else 0.2 * math.sin(7.2)
        q = euler_quat_xyzw(0.0, 0.0, yaw)
        Rw = np_ekf.quat_R(np.array([q[3], q[0], q[1], q[2]]))
        pos = pos + Rw @ np.array([speed, 0, 0]) * dt
        yaw_rate = 0.2 * 0.002 * math.cos(0.002 * k) / dt if k < 3600 else 0.0
        gyro = np.array([0, 0, yaw_rate]) + imu_bias_g + rng.normal(0, 1e-4, 3)
        acc = Rw.T @ np.array([0, 0, 9.81]) + np.array([0, speed * yaw_rate, 0]) + imu_bias_a + rng.normal(0, 1e-3, 3)
        if k == 2:
            assert e.CallbackPcmInitOdom(t, pos, q) and r.update_pcm_odom(t, pos, q, np.eye(6) * 1e-9, np_ekf.PCM_INIT)
        imu_rot_before, bg_before = r.imu_rot.copy(), r.bg.copy()
        a, b = e.RunPredictionImu(t, gyro, acc), r.predict_imu(t, gyro, acc)
        assert a == b
        calibrated += bool(np.any(r.imu_rot != imu_rot_before))
        zupted += bool(b and k >= 3600 and np.any(r.bg != bg_before))
        e.GetCurrentState(); r.publish()
        if k > 2 and k % 20 == 0:
            cov = np.diag([1e-4, 1e-4, 1e-4, 1e-6, 1e-6, 1e-6])
            mp = pos + rng.normal(0, 0.005, 3)
            assert e.RunGnssUpdate(t, mp, q, cov[:3, :3], cov[3:, 3:], GnssSource.PCM) == r.update_pose(t, mp, q, cov[:3, :3], cov[3:, 3:], np_ekf.PCM)
        if k % 11 == 0:
            compare(e, r)
    compare(e, r)
    assert calibrated > 500 and zupted > 500  # both modes were live, not just enabled


def _np_enu(ref, lat, lon, h):
    """Independent statement of the geodetic -> local tangent plane conversion (textbook form with e^2 and the prime-vertical radius,
    rotation written as R_x(90 deg - lat0) R_z(90 deg + lon0))."""
    a, f = 6378137.0, 1.0 / 298.257223563
    e2 = 2 * f - f * f

    def ecef(la, lo, hh):
        la, lo = math.radians(la), math.radians(lo)
        N = a / math.sqrt(1 - e2 * math.sin(la) ** 2)
        return np.array([(N + hh) * math.cos(la) * math.cos(lo), (N + hh) * math.cos(la) * math.sin(lo), (N * (1 - e2) + hh) * math.sin(la)])
    la0, lo0 = math.radians(ref[0]), math.radians(ref[1])
    Rz = np.array([[-math.sin(lo0), math.cos(lo0), 0], [-math.cos(lo0), -math.sin(lo0), 0], [0, 0, 1]])
    Rx = np.array([[1, 0, 0], [0, math.sin(la0), math.cos(la0)], [0, -math.cos(la0), math.sin(la0)]])
    return Rx @ Rz @ (ecef(lat, lon, h) - ecef(*ref))


def test_gps_projection_and_navsatfix_front_end():
    """ProjectGpsPoint = GeographicLib LocalCartesian.Forward (ekf_localization.cpp:643-648) and CallbackNavsatFix (:92-125)."""
    from elimaloc_amd.ekf import ProjectGpsPoint
    ref = (37.5407, 127.0793, 35.0)
    np.testing.assert_allclose(ProjectGpsPoint(ref, *ref), 0.0, atol=1e-9)
    np.testing.assert_allclose(ProjectGpsPoint(ref, ref[0], ref[1], ref[2] + 12.5), [0, 0, 12.5], atol=1e-9)
    rng = np.random.default_rng(3)
    for _ in range(200):
        lat, lon, h = ref[0] + rng.uniform(-0.2, 0.2), ref[1] + rng.uniform(-0.2, 0.2), ref[2] + rng.uniform(-50, 200)
        np.testing.assert_allclose(ProjectGpsPoint(ref, lat, lon, h), _np_enu(ref, lat, lon, h), rtol=0, atol=2e-8)
    # one arc second to the north: the meridional radius of curvature at the reference latitude
    a, f = 6378137.0, 1.0 / 298.257223563
    e2 = 2 * f - f * f
    Mr = a * (1 - e2) / (1 - e2 * math.sin(math.radians(ref[0])) ** 2) ** 1.5
    n1 = ProjectGpsPoint(ref, ref[0] + 1.0 / 3600.0, ref[1], ref[2])
    assert abs(n1[1] - (Mr + ref[2]) * math.radians(1.0 / 3600.0)) < 1e-3 and abs(n1[0]) < 1e-9
    # the front end: squared standard deviations, the use_gps switch, the uncertainty gate, position rows only
    cfg = EkfConfig()
    e, r = EkfAlgorithm(cfg), np_ekf.NpEkf(cfg_dict(cfg))
    lat, lon, h = ref[0] + 1e-4, ref[1] - 2e-4, ref[2] + 1.0
    cov = np.diag([0.3, 0.4, 0.8])
    ok, pos = e.CallbackNavsatFix(5.0, lat, lon, h, cov, ref, use_gps=False)
    assert not ok and np.allclose(pos, _np_enu(ref, lat, lon, h), atol=2e-8)
    ok, _ = e.CallbackNavsatFix(5.0, lat, lon, h, np.diag([1.5, 0.4, 0.8]), ref)  # 1.5^2 > gnss_uncertainy_max_m
    assert not ok
    np.testing.assert_array_equal(e.State()["P"], r.P)
    ok, pos = e.CallbackNavsatFix(5.0, lat, lon, h, cov, ref)
    assert ok and r.update_pose(5.0, pos, [0, 0, 0, 1], np.diag([0.09, 0.16, 0.64]), np.zeros((3, 3)), np_ekf.NAVSATFIX)
    compare(e, r)
