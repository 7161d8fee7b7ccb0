"""Host-side mirror of the deskew part of PcmMatching (pcm_matching.cpp:467-824).

DeskewPointCloud keeps the reference's call shape: raw scan with per-point time + message stamp in, undistorted
points + scan-end time out, False when IMU/odometry tables are unavailable (pcm.cpp:494-496).  The IMU / odometry
queues of the node (deq_imu_, deq_odom_) are passed as arrays.  The per-point loop runs as a HIP kernel behind
elm_deskew; the table preparation (ImuDeskewInfo / OdomDeskewInfo) is the C-ABI's host function
elm_deskew_prepare.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import DeskewTables, check
from .registration import default_context

I_QUEUE_LENGTH = 2000  # pcm.hpp:113


class PcmDeskew:
    def __init__(self, ctx=None, b_lidar_scan_time_end=True, b_run_deskew=True):
        self.ctx = ctx or default_context()
        self.b_lidar_scan_time_end = bool(b_lidar_scan_time_end)  # loc.ini:5
        self.b_run_deskew = bool(b_run_deskew)                    # loc.ini:86
        self.d_time_scan_cur_ = 0.0
        self.d_time_scan_end_ = 0.0
        self._tabs = [np.zeros(I_QUEUE_LENGTH) for _ in range(4)]
        self.tables = None

    def prepare(self, imu, odom, stamp, front_time, back_time):
        """ImuDeskewInfo + OdomDeskewInfo (pcm.cpp:533-729).
        imu: (k,4) rows (t, wx, wy, wz) in the ego frame; odom: (m,14) rows
        (t, px,py,pz, qx,qy,qz,qw, vx,vy,vz, wx,wy,wz)."""
        imu = np.ascontiguousarray(imu, dtype=np.float64).reshape(-1, 4)
        odom = np.ascontiguousarray(odom, dtype=np.float64).reshape(-1, 14)
        tab = DeskewTables()
        dp = C.POINTER(C.c_double)
        check(_lib.lib().elm_deskew_prepare(imu.ctypes.data_as(dp), imu.shape[0], odom.ctypes.data_as(dp),
                                            odom.shape[0], float(stamp), float(front_time), float(back_time),
                                            int(self.b_lidar_scan_time_end), int(self.b_run_deskew),
                                            self._tabs[0].ctypes.data_as(dp), self._tabs[1].ctypes.data_as(dp),
                                            self._tabs[2].ctypes.data_as(dp), self._tabs[3].ctypes.data_as(dp),
                                            I_QUEUE_LENGTH, C.byref(tab)), None, "elm_deskew_prepare")
        self.tables = tab
        self.d_time_scan_cur_ = tab.d_time_scan_cur
        self.d_time_scan_end_ = tab.d_time_scan_end
        return tab

    def DeskewPointCloud(self, xyz, point_time, timestamp, imu, odom):
        """pcm.cpp:467-531.  xyz (n,3) float32, point_time (n,) float32 (time field of PointXYZIT).
        Returns (ok, undistorted (n,3) float32 or None)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(point_time, dtype=np.float32).copy()
        front, back = float(t[0]), float(t[-1])
        if self.b_lidar_scan_time_end:
            t -= np.float32(front)  # pcm.cpp:483-485
        tab = self.prepare(imu, odom, timestamp, front, back)
        out = np.empty_like(xyz)
        ok = C.c_int(0)
        fp = C.POINTER(C.c_float)
        check(_lib.lib().elm_deskew(self.ctx._h, xyz.ctypes.data_as(fp), t.ctypes.data_as(fp), xyz.shape[0],
                                    C.byref(tab), out.ctypes.data_as(fp), C.byref(ok)), self.ctx._h, "elm_deskew")
        if not ok.value:
            return False, None
        return True, out

    def DeskewDownsample(self, xyz, point_time, timestamp, imu, odom, voxel_size):
        """DeskewPointCloud + VoxelHashMap::VoxelDownsample (pcm.cpp:238 + 257-258) fused on the device
        (elm_deskew_downsample): returns (ok, kept undistorted points (k,3) float32 in input order)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(point_time, dtype=np.float32).copy()
        front, back = float(t[0]), float(t[-1])
        if self.b_lidar_scan_time_end:
            t -= np.float32(front)
        tab = self.prepare(imu, odom, timestamp, front, back)
        scan, ok = C.c_void_p(), C.c_int(0)
        fp = C.POINTER(C.c_float)
        L = _lib.lib()
        check(L.elm_deskew_downsample(self.ctx._h, xyz.ctypes.data_as(fp), t.ctypes.data_as(fp), xyz.shape[0], C.byref(tab),
                                      float(voxel_size), C.byref(scan), C.byref(ok)), self.ctx._h, "elm_deskew_downsample")
        if not ok.value:
            return False, None
        n = int(L.elm_scan_size(scan))
        out = np.empty((n, 3), np.float32)
        try:
            check(L.elm_scan_download(scan, out.ctypes.data_as(fp), n), self.ctx._h, "elm_scan_download")
        finally:
            L.elm_scan_destroy(scan)
        return True, out
