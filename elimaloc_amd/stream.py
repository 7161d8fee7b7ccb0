"""The launch file's topology without ROS (launch/ELiMaLoc.launch: pcm_matching <-> ekf_localization), BASELINE config 5:
every LiDAR scan is deskewed and registered on the GPU, its pose goes to the CPU EKF as a PCM update, and the EKF's
IMU-rate odometry is what the next scan is deskewed with and seeded from.

    imu  -> EkfLocalization::CallbackImu   (ekfl.cpp:137-145)  RunPredictionImu + PublishInThread -> odometry message
         -> PcmMatching::CallbackImu       (pcm.cpp:326-336)   deq_imu_
    odom -> PcmMatching::CallbackEkfState  (pcm.cpp:338-354)   deq_odom_ (poses at |x| or |y| < 1e-9 are refused)
    scan -> PcmMatching::CallbackPointCloud(pcm.cpp:198-324)   -> PublishPcmOdom -> EkfLocalization::CallbackPcmOdom
    init -> PcmMatching::CallbackInitialPose (pcm.cpp:356-447) -> CallbackPcmInitOdom (ekfl.cpp:181-203)
"""
import collections
import math

import numpy as np

from .ekf import EkfAlgorithm, GnssSource

QUEUE_LENGTH = 2000  # i_queue_length_ (pcm.hpp:112)


def rot_to_quat_xyzw(R):
    """Rotation matrix -> quaternion (x, y, z, w), Shepperd's method."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    else:
        i = int(np.argmax([R[0, 0], R[1, 1], R[2, 2]]))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = [0.0, 0.0, 0.0, (R[k, j] - R[j, k]) / s]
        q[i], q[j], q[k] = 0.25 * s, (R[j, i] + R[i, j]) / s, (R[k, i] + R[i, k]) / s
    return np.array(q)


def euler_zyx_quat_xyzw(roll, pitch, yaw):  # UpdateEkfOdom (ekfl.cpp:523-525)
    cr, sr, cp, sp, cy, sy = (math.cos(roll / 2), math.sin(roll / 2), math.cos(pitch / 2), math.sin(pitch / 2),
                              math.cos(yaw / 2), math.sin(yaw / 2))
    return np.array([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
                     cr * cp * cy + sr * sp * sy])


class LocalizationStream:
    def __init__(self, pcm_node, ekf=None, native=False):
        self.pcm = pcm_node
        self.native = native  # True: the whole CallbackPointCloud is one C-ABI call
        self.ekf = ekf or EkfAlgorithm()
        self.deq_imu_ = collections.deque(maxlen=QUEUE_LENGTH * 4)
        self.deq_odom_ = collections.deque(maxlen=QUEUE_LENGTH * 4)
        self.last_raw_scan = None
        self.n_scan, self.n_ok = 0, 0

    # ---- ekf_localization side
    def CallbackImu(self, t, gyro, acc):
        """gyro / acc already in the ego frame (ImuStructConverter with the calibration rotation applied by the caller)."""
        self.ekf.RunPredictionImu(t, gyro, acc)
        ego = self.ekf.GetCurrentState()
        q = euler_zyx_quat_xyzw(ego["roll_rad"], ego["pitch_rad"], ego["yaw_rad"])
        row = np.array([ego["timestamp"], ego["x_m"], ego["y_m"], ego["z_m"], q[0], q[1], q[2], q[3], ego["vx"], ego["vy"], ego["vz"],
                        ego["roll_vel"], ego["pitch_vel"], ego["yaw_vel"]])
        # ---- pcm_matching side
        if self.deq_imu_ and self.deq_imu_[-1][0] > t:
            self.deq_imu_.clear()
        self.deq_imu_.append(np.array([t, gyro[0], gyro[1], gyro[2]]))
        if abs(row[1]) < 1e-9 or abs(row[2]) < 1e-9:
            return ego
        if self.deq_odom_ and self.deq_odom_[-1][0] > row[0]:
            self.deq_odom_.clear()
        self.deq_odom_.append(row)
        return ego

    def _windows(self, stamp):
        # the node pops entries older than scan_start - 0.01 s (IMU) / - 0.1 s (odom) (pcm.cpp:536-541, 590-596); hand over a
        # window that safely contains them
        while self.deq_imu_ and self.deq_imu_[0][0] < stamp - 1.0:
            self.deq_imu_.popleft()
        while self.deq_odom_ and self.deq_odom_[0][0] < stamp - 1.0:
            self.deq_odom_.popleft()
        return np.array(self.deq_imu_).reshape(-1, 4), np.array(self.deq_odom_).reshape(-1, 14)

    def CallbackPointCloud(self, xyz, point_time, stamp):
        self.n_scan += 1
        self.last_raw_scan = xyz
        if not self.deq_odom_:  # b_get_first_odom_ == false (pcm.cpp:208-211)
            return None
        imu, odom = self._windows(stamp)
        out = (self.pcm.CallbackPointCloudNative if self.native else self.pcm.CallbackPointCloud)(xyz, point_time, stamp, imu, odom)
        if out is None:
            return None
        self.n_ok += 1
        P = out["pose_ego"]
        out["ekf_updated"] = self.ekf.CallbackPcmOdom(out["time"], P[:3, 3], rot_to_quat_xyzw(P[:3, :3]), out["covariance"],
                                                      GnssSource.PCM)
        return out

    def CallbackInitialPose(self, rviz_pose, stamp):
        out = self.pcm.CallbackInitialPose(rviz_pose, self.last_raw_scan)
        if out is None:
            return None
        P = out["pose_ego"]
        self.ekf.CallbackPcmInitOdom(stamp, P[:3, 3], rot_to_quat_xyzw(P[:3, :3]))
        return out
