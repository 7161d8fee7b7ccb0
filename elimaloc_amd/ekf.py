"""Host-side mirror of the reference's EkfAlgorithm / EkfLocalization pose-update interface.

The filter itself is plain CPU C++ in libelimaloc_hip.so (csrc/elm_ekf.cpp) -- north_star keeps the EKF on the CPU.
Names follow ekf_localization/include/ekf_algorithm.hpp:72-104 (RunPredictionImu, RunPrediction, RunCanUpdate, RunGnssUpdate,
GetCurrentState) and
ekf_localization.cpp:147-220 (CallbackPcmOdom, CallbackPcmInitOdom).
"""
import ctypes as C
import enum

import numpy as np

from . import _lib


class GnssSource(enum.IntEnum):  # localization_struct.hpp:28
    NOVATEL = 0
    NAVSATFIX = 1
    BESTPOS = 2
    PCM = 3
    PCM_INIT = 4


class EkfConfig:
    """[ekf_localization] keys (config/localization.ini); defaults from elm_ekf_config_default."""

    def __init__(self, **kw):
        self.c = _lib.EkfConfig()
        _lib.lib().elm_ekf_config_default(C.byref(self.c))
        for k, v in kw.items():
            if not hasattr(self.c, k):
                raise AttributeError(f"unknown EKF config key {k}")
            setattr(self.c, k, v)

    def __getattr__(self, k):
        return getattr(self.__dict__["c"], k)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def ProjectGpsPoint(ref_lla, lat, lon, alt):
    """ekfl.cpp:643-648: east / north / up metres of (lat, lon, alt) at the reference (lat, lon, alt), WGS84."""
    out = np.empty(3)
    _lib.check(_lib.lib().elm_gps_project(float(ref_lla[0]), float(ref_lla[1]), float(ref_lla[2]), float(lat), float(lon), float(alt), _dp(out)))
    return out


class EkfAlgorithm:
    def __init__(self, cfg=None):
        self.cfg = cfg or EkfConfig()
        self._h = C.c_void_p()
        _lib.check(_lib.lib().elm_ekf_create(C.byref(self.cfg.c), C.byref(self._h)), None, "elm_ekf_create")

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.lib().elm_ekf_destroy(self._h)
            self._h = None

    def RunPredictionImu(self, timestamp, gyro, acc):
        g = np.ascontiguousarray(gyro, np.float64)
        a = np.ascontiguousarray(acc, np.float64)
        out = C.c_int(0)
        _lib.check(_lib.lib().elm_ekf_predict_imu(self._h, float(timestamp), _dp(g), _dp(a), C.byref(out)))
        return bool(out.value)

    def RunPrediction(self, timestamp):
        """ekfa.cpp:81-165: the constant-velocity model (the node's timer path when use_imu = 0)."""
        out = C.c_int(0)
        _lib.check(_lib.lib().elm_ekf_predict(self._h, float(timestamp), C.byref(out)))
        return bool(out.value)

    def RunCanUpdate(self, timestamp, vel, gyro):
        """ekfa.cpp:434-506 (+ ZuptCan): vehicle-frame velocity and angular rate of the CAN message."""
        v = np.ascontiguousarray(vel, np.float64)
        g = np.ascontiguousarray(gyro, np.float64)
        out = C.c_int(0)
        _lib.check(_lib.lib().elm_ekf_update_can(self._h, float(timestamp), _dp(v), _dp(g), C.byref(out)))
        return bool(out.value)

    def CallbackNavsatFix(self, stamp, lat, lon, alt, position_covariance, ref_lla, use_gps=True, gnss_uncertainty_max_m=1.0):
        """ekfl.cpp:92-125.  Returns (updated, projected position)."""
        pc = np.ascontiguousarray(position_covariance, np.float64).reshape(9)
        pos = np.empty(3)
        out = C.c_int(0)
        _lib.check(_lib.lib().elm_ekf_update_navsatfix(self._h, float(stamp), float(lat), float(lon), float(alt), _dp(pc), float(ref_lla[0]),
                                                       float(ref_lla[1]), float(ref_lla[2]), int(bool(use_gps)), float(gnss_uncertainty_max_m),
                                                       _dp(pos), C.byref(out)))
        return bool(out.value), pos

    def RunGnssUpdate(self, timestamp, pos, quat_xyzw, pos_cov, rot_cov, source=GnssSource.PCM):
        p = np.ascontiguousarray(pos, np.float64)
        q = np.ascontiguousarray(quat_xyzw, np.float64)
        pc = np.ascontiguousarray(pos_cov, np.float64).reshape(9)
        rc = np.ascontiguousarray(rot_cov, np.float64).reshape(9)
        out = C.c_int(0)
        _lib.check(_lib.lib().elm_ekf_update_pose(self._h, float(timestamp), _dp(p), _dp(q), _dp(pc), _dp(rc), int(source),
                                                  C.byref(out)))
        return bool(out.value)

    def CallbackPcmOdom(self, stamp, pos, quat_xyzw, covariance36, source=GnssSource.PCM):
        p = np.ascontiguousarray(pos, np.float64)
        q = np.ascontiguousarray(quat_xyzw, np.float64)
        cv = np.ascontiguousarray(covariance36, np.float64).reshape(36)
        out = C.c_int(0)
        _lib.check(_lib.lib().elm_ekf_update_pcm_odom(self._h, float(stamp), _dp(p), _dp(q), _dp(cv), int(source),
                                                      C.byref(out)))
        return bool(out.value)

    def CallbackPcmInitOdom(self, stamp, pos, quat_xyzw):
        return self.CallbackPcmOdom(stamp, pos, quat_xyzw, np.eye(6) * 1e-9, GnssSource.PCM_INIT)

    def State(self):
        s = _lib.EkfStateC()
        _lib.check(_lib.lib().elm_ekf_get_state(self._h, C.byref(s)))
        return {
            "x": np.array(s.x), "rot_xyzw": np.array(s.rot_xyzw), "imu_rot_xyzw": np.array(s.imu_rot_xyzw),
            "P": np.array(s.P).reshape(27, 27), "timestamp": s.timestamp,
            "state_initialized": bool(s.b_state_initialized), "yaw_initialized": bool(s.b_yaw_initialized),
            "rotation_stabilized": bool(s.b_rotation_stabilized), "state_stabilized": bool(s.b_state_stabilized),
            "pcm_init_on_going": bool(s.b_pcm_init_on_going),
        }

    def GetCurrentState(self):
        """GetCurrentState + PublishInThread's state-history upkeep; returns the EgoState fields as a dict."""
        s = _lib.EgoStateC()
        _lib.check(_lib.lib().elm_ekf_publish(self._h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in _lib.EgoStateC._fields_}
