"""Independent numpy float64 re-derivation of the reference's 27-state EKF (checker for csrc/elm_ekf.cpp).

TEST INFRASTRUCTURE.  Written from the reference's formulas (ekf_algorithm.cpp / .hpp, localization_functions.hpp) with
dense numpy algebra (explicit H matrices, np.linalg.inv, matrix exponentials by Rodrigues) rather than the index tricks
the product uses, so an indexing or ordering slip in either shows up as a mismatch.  PARITY UNPINNED: the reference has
no EKF tests or golden vectors and cannot be built here (Eigen / ROS / GeographicLib absent).
"""
import math

import numpy as np

N = 27
S_X, S_ROLL, S_VX, S_RR, S_AX, S_BG, S_BA, S_G, S_IMU = 0, 3, 6, 9, 12, 15, 18, 21, 24
NOVATEL, NAVSATFIX, BESTPOS, PCM, PCM_INIT = range(5)
DEG = math.pi / 180.0


def quat_mul(a, b):  # (w, x, y, z)
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw])


def quat_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R_quat(R):  # Eigen's Quaternion(Matrix3) branch structure
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        return np.array([w, (R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s])
    i = 0
    if R[1, 1] > R[0, 0]:
        i = 1
    if R[2, 2] > R[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    v = np.zeros(3)
    v[i] = 0.5 * s
    s = 0.5 / s
    w = (R[k, j] - R[j, k]) * s
    v[j] = (R[j, i] + R[i, j]) * s
    v[k] = (R[k, i] + R[i, k]) * s
    return np.array([w, *v])


def aa_quat(angle, axis):
    return np.array([math.cos(angle / 2), *(math.sin(angle / 2) * np.asarray(axis, float))])


def rotvec_quat(v):
    n = np.linalg.norm(v)
    return aa_quat(n, v / n if n > 0 else v)


def skew(a):
    return np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])


def so3_exp(w):
    th = np.linalg.norm(w)
    if th < 1e-5:
        return np.eye(3)
    K = skew(w / th)
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K


def rot_jac(gyro, dt):
    w = gyro * dt
    th = np.linalg.norm(w)
    if th < 1e-5:
        return np.zeros((3, 3))
    K = skew(w / th)
    return dt * (np.eye(3) + (1 - math.cos(th)) / th**2 * K + (th - math.sin(th)) / th**3 * K @ K)


def rot_to_vec(R):
    if abs(R[2, 0]) > 0.998:
        a = np.array([0.0, math.pi / 2 * (1 if R[2, 0] >= 0 else -1), math.atan2(-R[1, 2], R[1, 1])])
    else:
        p = math.asin(-R[2, 0])
        a = np.array([math.atan2(R[2, 1] / math.cos(p), R[2, 2] / math.cos(p)), p,
                      math.atan2(R[1, 0] / math.cos(p), R[0, 0] / math.cos(p))])
    return np.array([math.fmod(v + math.pi, 2 * math.pi) - math.pi for v in a])


def norm_angle(a):
    while a > math.pi:
        a -= 2 * math.pi
    while a < -math.pi:
        a += 2 * math.pi
    return a


def g2l(g, roll, pitch, yaw):
    cy, sy, cp, sp, cr, sr = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch), math.cos(roll), math.sin(roll)
    M = np.array([[cy * cp, sy * cp, -sp], [cy * sp * sr - sy * cr, sy * sp * sr + cy * cr, cp * sr],
                  [cy * sp * cr + sy * sr, sy * sp * cr - cy * sr, cp * cr]])
    return M @ np.asarray(g, float)


class NpEkf:
    def __init__(self, cfg):
        self.c = cfg  # dict of the elm_ekf_config fields
        c = cfg
        self.pos = np.array([c["ekf_init_x_m"], c["ekf_init_y_m"], c["ekf_init_z_m"]])
        self.rot = quat_mul(quat_mul(aa_quat(c["ekf_init_yaw_deg"] * DEG, [0, 0, 1]), aa_quat(c["ekf_init_pitch_deg"] * DEG, [0, 1, 0])),
                            aa_quat(c["ekf_init_roll_deg"] * DEG, [1, 0, 0]))
        self.imu_rot = np.array([1.0, 0, 0, 0])
        self.vel, self.gyro, self.acc, self.bg, self.ba = (np.zeros(3) for _ in range(5))
        self.grav = np.array([0, 0, c["imu_gravity"]])
        d = np.full(N, 100.0)
        d[S_BG:S_BG + 3] = c["ekf_imu_bias_cov_gyro"]
        d[S_BA:S_BA + 3] = c["ekf_imu_bias_cov_acc"]
        d[S_G:S_G + 3] = c["ekf_imu_bias_cov_acc"]
        d[S_IMU:S_IMU + 3] = c["ekf_imu_bias_cov_gyro"]
        self.P = np.diag(d)
        self.reset = True
        self.state_init = self.yaw_init = self.rot_stab = self.state_stab = self.pcm_init = False
        self.pcm_count = 0
        self.prev_t = 0.0
        self.ckf = None
        self.prev_ego = dict(timestamp=0.0)
        self.deq = []
        self.prev_can_t = 0.0
        self.can_bias = 0.0

    # -- flags
    def _sd(self, i):
        return math.sqrt(self.P[i, i])

    def _chk_rot(self):
        self.rot_stab = all(self._sd(i) < 0.2 * DEG for i in (3, 4, 5))

    def _update(self, H, R, Y):
        S = H @ self.P @ H.T + R
        K = self.P @ H.T @ np.linalg.inv(S)
        dx = K @ Y
        self.pos = self.pos + dx[0:3]
        self.vel = self.vel + dx[6:9]
        self.gyro = self.gyro + dx[9:12]
        self.acc = self.acc + dx[12:15]
        self.bg = self.bg + dx[15:18]
        self.ba = self.ba + dx[18:21]
        self.grav = self.grav + dx[21:24]
        q = quat_mul(self.rot, rotvec_quat(dx[3:6]))
        self.rot = q / np.linalg.norm(q)
        q = quat_mul(self.imu_rot, rotvec_quat(dx[24:27]))
        self.imu_rot = q / np.linalg.norm(q)
        self.P = self.P - K @ H @ self.P

    def _ckf(self, t, acc_in):
        acc_meas = acc_in - self.ba
        vel_local = quat_R(self.rot).T @ self.vel  # rot is unit: inverse rotation = transpose
        cen = vel_local[0] * self.gyro[2]
        if self.ckf is None:
            self.ckf = [vel_local[0], t]
        dt = t - self.ckf[1]
        if dt < 1e-6:
            return
        est_ax = (vel_local[0] - self.ckf[0]) / dt
        self.ckf = [vel_local[0], t]
        comp = acc_meas - np.array([0, cen, 0])
        if self.rot_stab:
            comp = comp - np.array([est_ax, 0, 0])
        diff = np.linalg.norm(acc_meas) - np.linalg.norm(self.grav)
        n = np.linalg.norm(comp)
        gd = comp / n if n > 0 else comp
        z = np.array([math.atan2(gd[1], gd[2]), -math.asin(gd[0])])
        rpy = rot_to_vec(quat_R(self.rot))
        inn = np.array([norm_angle(z[0] - rpy[0]), norm_angle(z[1] - rpy[1])])
        H = np.zeros((2, N))
        H[0, 3] = H[1, 4] = 1.0
        base = (1.0 if self.state_init else 10.0) * DEG
        cu, lu, du = abs(cen) / 9.81 * 10, abs(est_ax) / 9.81 * 10, abs(diff) / 9.81 * 10
        R = np.diag([max((base * (1 + du + cu)) ** 2, DEG**2), max((base * (1 + du + lu)) ** 2, DEG**2)])
        self._update(H, R, inn)

    def predict_imu(self, t, gyro_in, acc_in):
        c = self.c
        gyro_in, acc_in = np.asarray(gyro_in, float), np.asarray(acc_in, float)
        if self.reset:
            self.prev_t, self.reset = t, False
            return False
        if self.pcm_init:
            self.prev_t = t
            return False
        self._chk_rot()
        use_ckf = c["gps_type"] == 1 or c["use_complementary_filter"]
        if not self.state_init:
            self.prev_t = t
            if self.yaw_init and use_ckf:
                self._ckf(t, acc_in)
            return False
        if abs(t - self.prev_t) < 1e-6:
            return False
        dt = t - self.prev_t
        G = quat_R(self.rot)
        cg = gyro_in - self.bg
        q = quat_mul(self.rot, R_quat(so3_exp(cg * dt)))
        self.rot = q / np.linalg.norm(q)
        ag = G @ (acc_in - self.ba) - self.grav
        self.pos = self.pos + self.vel * dt + 0.5 * ag * dt * dt
        self.vel = self.vel + ag * dt
        self.gyro, self.acc = cg, ag
        Q = np.zeros((N, N))
        for s0, sd in ((S_X, c["state_std_pos_m"]), (S_ROLL, c["state_std_rot_deg"] * DEG), (S_VX, c["state_std_vel_mps"]),
                       (S_RR, c["imu_std_gyro_dps"] * DEG), (S_AX, c["imu_std_acc_mps"]), (S_BG, c["ekf_imu_bias_cov_gyro"]),
                       (S_BA, c["ekf_imu_bias_cov_acc"]), (S_G, c["ekf_imu_bias_cov_acc"]), (S_IMU, c["state_std_rot_deg"] * DEG)):
            Q[s0:s0 + 3, s0:s0 + 3] = np.eye(3) * sd**2 * dt * dt
        F = np.eye(N)
        F[S_X:S_X + 3, S_VX:S_VX + 3] = np.eye(3) * dt
        F[S_X:S_X + 3, S_BA:S_BA + 3] = -0.5 * G * dt * dt
        F[S_ROLL:S_ROLL + 3, S_BG:S_BG + 3] = -rot_jac(cg, dt)
        F[S_VX:S_VX + 3, S_BA:S_BA + 3] = -G * dt
        F[S_RR:S_RR + 3, S_BG:S_BG + 3] = -np.eye(3)
        F[S_AX:S_AX + 3, S_BA:S_BA + 3] = -G
        if c["imu_estimate_gravity"]:
            F[2, S_G + 2] = -0.5 * dt * dt
            F[S_VX + 2, S_G + 2] = -dt
            F[S_AX + 2, S_G + 2] = -1.0
        self.P = F @ self.P @ F.T + Q
        self.prev_t = t
        if c["use_zupt"]:
            self._zupt_imu(gyro_in, acc_in)
        if use_ckf:
            self._ckf(t, acc_in)
        if c["imu_estimate_calibration"]:
            self._calibrate()
        return True

    # -- side modes (off in the shipped localization.ini)
    def _zupt_imu(self, gyro_in, acc_in):  # ekf_algorithm.cpp:508-565
        Rw = quat_R(self.rot / np.linalg.norm(self.rot))
        vl = Rw.T @ self.vel
        if abs(vl[0]) > 0.1:
            return
        self.vel = self.vel + (0.1 - abs(vl[0])) / 0.1 * 0.1 * (-self.vel)
        if np.linalg.norm(self.gyro) > 0.1 or np.linalg.norm(self.acc[:2]) > 0.1:
            return
        ba0 = self.ba.copy()
        self.bg = self.bg + 0.01 * (gyro_in - self.bg)
        self.ba = ba0 + 0.01 * (acc_in - (Rw.T @ self.grav + ba0))
        if self.c["imu_estimate_gravity"]:
            self.grav = self.grav + np.array([0, 0, 0.01 * (Rw @ (acc_in - ba0) - self.grav)[2]])

    def _calibrate(self):  # ekf_algorithm.cpp:703-776
        if np.linalg.norm(self.vel) < 3.0 or not self.rot_stab:
            return
        Rrel = quat_R(self.rot / np.linalg.norm(self.rot)) @ quat_R(self.imu_rot / np.linalg.norm(self.imu_rot)).T
        d = Rrel.T @ self.vel
        d = d / np.linalg.norm(d)
        inn = np.array([0.0, math.asin(d[2]), -math.atan2(d[1], d[0])])
        H = np.zeros((3, N))
        H[0, S_IMU] = H[1, S_IMU + 1] = H[2, S_IMU + 2] = 1.0
        self._update(H, np.eye(3) * DEG**2, inn)

    def predict(self, t):  # RunPrediction, ekf_algorithm.cpp:81-165
        c = self.c
        if self.reset:
            self.prev_t, self.reset = t, False
            return False
        if self.pcm_init:
            self.prev_t = t
            return False
        if abs(t - self.prev_t) < 1e-6:
            return False
        dt = t - self.prev_t
        v0, a0, g0 = self.vel.copy(), self.acc.copy(), self.gyro.copy()
        self.pos = self.pos + v0 * dt + 0.5 * a0 * dt * dt
        q = quat_mul(self.rot, R_quat(so3_exp(g0 * dt)))
        self.rot = q / np.linalg.norm(q)
        self.vel = v0 + a0 * dt
        Q = np.zeros((N, N))
        for s0, sd in ((S_X, c["state_std_pos_m"]), (S_ROLL, c["state_std_rot_deg"] * DEG), (S_VX, c["state_std_vel_mps"]),
                       (S_RR, c["state_std_gyro_dps"]), (S_AX, c["state_std_acc_mps"])):
            Q[s0:s0 + 3, s0:s0 + 3] = np.eye(3) * sd**2 * dt * dt
        F = np.eye(N)
        F[S_X:S_X + 3, S_VX:S_VX + 3] = np.eye(3) * dt
        F[S_ROLL:S_ROLL + 3, S_RR:S_RR + 3] = np.eye(3) * dt
        F[S_X:S_X + 3, S_AX:S_AX + 3] = np.eye(3) * 0.5 * dt * dt
        F[S_VX:S_VX + 3, S_AX:S_AX + 3] = np.eye(3) * dt
        self.P = F @ self.P @ F.T + Q
        self.prev_t = t
        return True

    def update_can(self, t, vel, gyro):  # RunCanUpdate + ZuptCan, ekf_algorithm.cpp:434-506, 567-587
        c = self.c
        vel, gyro = np.asarray(vel, float), np.asarray(gyro, float)
        if abs(t - self.prev_can_t) < 0.01:
            return False
        v = vel.copy()
        v[0] *= c["can_vel_scale_factor"]
        Rw = quat_R(self.rot)
        H = np.zeros((4, N))
        H[0, S_VX] = H[1, S_VX + 1] = H[2, S_VX + 2] = H[3, S_RR + 2] = 1.0
        z = np.concatenate([Rw @ v, [gyro[2] - self.can_bias]])
        zs = np.concatenate([self.vel, [self.gyro[2]]])
        s = c["ekf_can_meas_uncertainty_vel_mps"]
        R = np.zeros((4, 4))
        R[:3, :3] = Rw @ np.diag([s**2, (2 * s) ** 2, (2 * s) ** 2]) @ Rw.T
        R[3, 3] = (c["ekf_can_meas_uncertainty_yaw_rate_deg"] * DEG) ** 2
        self._update(H, R, z - zs)
        self.prev_can_t = t
        if not np.linalg.norm(vel) > 0.05:
            self.can_bias = 0.05 * gyro[2] + 0.95 * self.can_bias
            self.vel = 0.95 * self.vel
        return True

    def update_pose(self, t, pos, quat_xyzw, pos_cov, rot_cov, source):
        c = self.c
        mq = np.array([quat_xyzw[3], quat_xyzw[0], quat_xyzw[1], quat_xyzw[2]], float)
        pos = np.asarray(pos, float)
        if source == PCM_INIT:
            self.pos, self.rot = pos.copy(), mq
            self.vel, self.gyro, self.acc, self.bg, self.ba = (np.zeros(3) for _ in range(5))
            self.grav = np.array([0, 0, c["imu_gravity"]])
            self.P[:15, :15] = np.eye(15) * 100.0
            self.state_init = self.yaw_init = self.pcm_init = True
            return True
        self.yaw_init = self._sd(5) < 5 * DEG
        self.state_init = all(self._sd(i) < 5 * DEG for i in (3, 4, 5)) and self._sd(0) < 1.0 and self._sd(1) < 1.0
        self._chk_rot()
        self.state_stab = self.rot_stab and self._sd(0) < 0.5 and self._sd(1) < 0.5
        if self.pcm_init and source == PCM:
            if self.pcm_count > 10:
                self.pcm_init = False
            self.pcm_count += 1
        H = np.zeros((6, N))
        H[:, :6] = np.eye(6)
        R = np.zeros((6, 6))
        R[:3, :3] = np.asarray(pos_cov, float).reshape(3, 3)
        R[3:, 3:] = np.asarray(rot_cov, float).reshape(3, 3)
        if source in (NOVATEL, BESTPOS, NAVSATFIX):
            R += np.diag([c["gnss_min_cov_x_m"], c["gnss_min_cov_y_m"], c["gnss_min_cov_z_m"], c["gnss_min_cov_roll_deg"] * DEG,
                          c["gnss_min_cov_pitch_deg"] * DEG, c["gnss_min_cov_yaw_deg"] * DEG])
        sa = rot_to_vec(quat_R(self.rot / np.linalg.norm(self.rot)))
        ma = rot_to_vec(quat_R(mq / np.linalg.norm(mq)))
        Y = np.concatenate([pos - self.pos, [norm_angle(ma[i] - sa[i]) for i in range(3)]])
        if source in (NAVSATFIX, BESTPOS):
            if not self.yaw_init:
                R[0, 0] += 3.0
                R[1, 1] += 3.0
            self._update(H[:3], R[:3, :3], Y[:3])
        else:
            self._update(H, R, Y)
        return True

    def publish(self):
        s = dict(timestamp=self.prev_t)
        if s["timestamp"] - self.prev_ego["timestamp"] < 1e-6:
            s = self.prev_ego
        else:
            eu = rot_to_vec(quat_R(self.rot))
            v, a = g2l(self.vel, *eu), g2l(self.acc, *eu)
            pc = np.abs(g2l([self.P[0, 0], self.P[1, 1], self.P[2, 2]], *eu))
            s.update(x_m=self.pos[0], y_m=self.pos[1], z_m=self.pos[2], roll_rad=eu[0], pitch_rad=eu[1], yaw_rad=eu[2],
                     roll_vel=self.gyro[0], pitch_vel=self.gyro[1], yaw_vel=self.gyro[2], vx=v[0], vy=v[1], vz=v[2],
                     ax=a[0], ay=a[1], az=a[2], x_cov_m=pc[0], y_cov_m=pc[1], z_cov_m=pc[2], roll_cov_rad=self.P[3, 3],
                     pitch_cov_rad=self.P[4, 4], yaw_cov_rad=self.P[5, 5])
            self.prev_ego = s
        if not self.deq or self.deq[-1]["timestamp"] + 1e-5 < s["timestamp"]:
            self.deq.append(s)
        if self.deq[-1]["timestamp"] > s["timestamp"]:
            self.deq = []
        self.deq = self.deq[-1000:]
        return s

    def update_pcm_odom(self, stamp, pos, quat_xyzw, cov36, source):
        cov = np.asarray(cov36, float).reshape(6, 6)
        if source == PCM_INIT:
            return self.update_pose(stamp, pos, quat_xyzw, cov[:3, :3], cov[3:, 3:], source)
        if not self.deq or self.deq[0]["timestamp"] > stamp:
            return False
        cur = self.deq[-1]
        closest = self.deq[0]
        for s in self.deq:
            closest = s
            if s["timestamp"] > stamp:
                break
        p = np.asarray(pos, float).copy()
        q = np.array([quat_xyzw[3], quat_xyzw[0], quat_xyzw[1], quat_xyzw[2]], float)
        t_out = stamp
        d = cur["timestamp"] - stamp
        if d > 0:
            dp, da = np.zeros(3), np.zeros(3)
            if abs(cur["timestamp"] - closest["timestamp"]) > 1e-5:
                ratio = d / (cur["timestamp"] - closest["timestamp"])
                dp = np.array([cur[k] - closest[k] for k in ("x_m", "y_m", "z_m")]) * ratio
                da = np.array([norm_angle(cur[k] - closest[k]) for k in ("roll_rad", "pitch_rad", "yaw_rad")]) * ratio
            t_out = cur["timestamp"]
            p = p + dp
            dq = quat_mul(quat_mul(aa_quat(da[2], [0, 0, 1]), aa_quat(da[1], [0, 1, 0])), aa_quat(da[0], [1, 0, 0]))
            q = quat_mul(q, dq)
            q = q / np.linalg.norm(q)
        return self.update_pose(t_out, p, [q[1], q[2], q[3], q[0]], cov[:3, :3], cov[3:, 3:], source)
