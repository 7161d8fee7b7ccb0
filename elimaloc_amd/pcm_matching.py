"""ROS-free mirror of the PcmMatching node's per-scan path (pcm_matching.cpp:22-115, 198-324, 356-447): the call
sequence either side of the device kernels, with the node's message queues passed as arrays.

    node = PcmMatching(cfg);  node.Init(map_xyz)
    out = node.CallbackPointCloud(xyz, point_time, stamp, imu, odom)      # deskew -> pose sync -> downsample -> ICP
    out = node.CallbackInitialPose(rviz_pose, last_raw_scan_xyz)          # ground height -> ICP on the raw scan

Everything numerical is behind the C ABI (HIP kernels for deskew / registration, host C++ for the float32 glue)."""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import check
from .deskew import PcmDeskew
from .registration import (Context, IcpMethod, Registration, RegistrationConfig, VoxelHashMap, default_context)


@dataclass
class PcmMatchingConfig:
    """[common_variable] / [pcm_matching] keys of config/localization.ini (loc.ini:2-9, 80-105) + calibration."""
    s_lidar_type: str = "velodyne"       # lidar_type ("ouster" -> OusterCloudmsg2cloud)
    i_input_index_sampling: int = 5      # input_index_sampling (applied on the Ouster path only, pcm.cpp:910)
    b_lidar_scan_time_end: bool = True   # lidar_scan_time_end
    d_lidar_time_delay: float = 0.03     # lidar_time_delay
    d_pcm_voxel_size: float = 1.0
    i_pcm_voxel_max_point: int = 30
    b_run_deskew: bool = True
    d_input_max_dist: float = 100.0
    d_input_voxel_ds_m: float = 1.5
    tf_ego_to_lidar: np.ndarray = field(default_factory=lambda: np.eye(4))
    registration: object = None          # RegistrationConfig (elm_reg_config)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def filter_points_by_distance(xyz, point_time, max_dist):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    t = None if point_time is None else np.ascontiguousarray(point_time, dtype=np.float32)
    out = np.empty_like(xyz)
    tout = None if t is None else np.empty_like(t)
    n = C.c_size_t(0)
    check(_lib.lib().elm_filter_points_by_distance(_fp(xyz), None if t is None else _fp(t), xyz.shape[0], float(max_dist),
                                                   _fp(out), None if t is None else _fp(tout), C.byref(n)), None,
          "elm_filter_points_by_distance")
    return out[:n.value], (None if t is None else tout[:n.value])


def voxel_downsample(xyz, voxel_size):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    keep = np.empty(xyz.shape[0], np.int64)
    n = C.c_size_t(0)
    check(_lib.lib().elm_voxel_downsample(_fp(xyz), xyz.shape[0], float(voxel_size), keep.ctypes.data_as(C.POINTER(C.c_int64)),
                                          C.byref(n)), None, "elm_voxel_downsample")
    return xyz[keep[:n.value]], keep[:n.value].copy()


def get_interpolated_pose(odom, t):
    od = np.ascontiguousarray(odom, dtype=np.float64).reshape(-1, 14)
    T = np.zeros(16, np.float32)
    ok = C.c_int(0)
    check(_lib.lib().elm_get_interpolated_pose(_dp(od), od.shape[0], float(t), _fp(T), C.byref(ok)), None,
          "elm_get_interpolated_pose")
    return bool(ok.value), T.reshape(4, 4).T.copy()


def shape_odom_covariance(local_cov, icp_ego_pose, d_icp_pose_std_m):
    lc = np.ascontiguousarray(np.asarray(local_cov, dtype=np.float64).T).ravel()  # column-major
    ps = np.ascontiguousarray(np.asarray(icp_ego_pose, dtype=np.float64).T).ravel()
    out = np.zeros(36)
    check(_lib.lib().elm_shape_odom_covariance(_dp(lc), _dp(ps), float(d_icp_pose_std_m), _dp(out)), None,
          "elm_shape_odom_covariance")
    return out.reshape(6, 6)


class PcmMatching:
    def __init__(self, cfg=None, ctx=None):
        self.cfg_ = cfg or PcmMatchingConfig()
        if self.cfg_.registration is None:
            self.cfg_.registration = RegistrationConfig()
        self.ctx = ctx or default_context()
        self.registration_ = Registration(self.cfg_.registration, self.ctx)
        self.local_map_ = VoxelHashMap(self.cfg_.d_pcm_voxel_size, self.cfg_.i_pcm_voxel_max_point, self.ctx)
        self.deskew_ = PcmDeskew(self.ctx, self.cfg_.b_lidar_scan_time_end, self.cfg_.b_run_deskew)
        self.icp_local_cov_ = np.eye(6)
        self.d_icp_pose_std_m = 0.0
        self.d_time_scan_end_ = 0.0

    @classmethod
    def FromFiles(cls, localization_ini, calibration_ini, map_pcd, ctx=None):
        """Init() the way the node does it (pcm.cpp:22-101): ProcessINI on the two ini files, load the PCD map, build."""
        from .formats import LoadPcdXyz, LoadPcmMatchingConfig
        node = cls(LoadPcmMatchingConfig(localization_ini, calibration_ini), ctx)
        node.Init(LoadPcdXyz(map_pcd))
        return node

    def Init(self, map_xyz):  # pcm.cpp:81-101
        self.registration_.Init(self.cfg_.registration)
        self.local_map_.Init(self.cfg_.d_pcm_voxel_size, self.cfg_.i_pcm_voxel_max_point)
        self.local_map_.AddPoints(map_xyz)
        m = IcpMethod(self.cfg_.registration.icp_method)
        if m in (IcpMethod.VGICP, IcpMethod.AVGICP):
            self.local_map_.CalVoxelCovAll()
        elif m == IcpMethod.GICP:
            self.local_map_.CalPointCovAll(self.cfg_.registration.gicp_cov_search_dist)

    def CallbackPointCloud(self, xyz, point_time, stamp, imu, odom):
        """pcm.cpp:198-324.  Returns None when the reference would publish nothing (deskew / pose sync / ICP failure),
        else dict(pose_ego 4x4 float64, covariance 6x6 row-major, fitness, time)."""
        import time
        tm = self.timings_ = {}
        t0 = time.perf_counter()

        def lap(name):
            nonlocal t0
            t1 = time.perf_counter()
            tm[name] = (t1 - t0) * 1e3
            t0 = t1

        stamp = float(stamp) - self.cfg_.d_lidar_time_delay                                   # :216-217
        xyz, point_time = filter_points_by_distance(xyz, point_time, self.cfg_.d_input_max_dist)  # :235
        lap("filter_ms")
        if xyz.shape[0] == 0:
            return None
        ok, undistorted = self.deskew_.DeskewPointCloud(xyz, point_time, stamp, imu, odom)    # :238
        lap("deskew_ms")
        if not ok:
            return None
        self.d_time_scan_end_ = self.deskew_.d_time_scan_end_
        ok, sync_ego_affine = get_interpolated_pose(odom, self.d_time_scan_end_)              # :248-251
        if not ok:
            return None
        src, _ = voxel_downsample(undistorted, self.cfg_.d_input_voxel_ds_m)                  # :257-258
        lap("sync_downsample_ms")
        sync_lidar_pose = sync_ego_affine.astype(np.float64) @ self.cfg_.tf_ego_to_lidar      # :266
        pose, ok, fit, cov = self.registration_.RunRegister(src, self.local_map_, sync_lidar_pose)  # :280-282
        lap("register_ms")
        self.icp_local_cov_ = cov
        if not ok:
            return None                                                                       # :289-292
        self.d_icp_pose_std_m = fit                                                           # :295
        icp_ego_pose = pose @ np.linalg.inv(self.cfg_.tf_ego_to_lidar)                        # :298
        out = dict(pose_ego=icp_ego_pose, pose_lidar=pose, fitness=fit, time=self.d_time_scan_end_,
                   covariance=shape_odom_covariance(cov, icp_ego_pose, fit), n_source=src.shape[0])
        lap("publish_ms")
        return out

    def _node_config(self):
        c = self.cfg_
        nc = _lib.PcmNodeConfig()
        _lib.lib().elm_pcm_node_config_default(C.byref(nc))
        nc.lidar_type = c.s_lidar_type.encode()
        nc.lidar_scan_time_end, nc.run_deskew = int(c.b_lidar_scan_time_end), int(c.b_run_deskew)
        nc.pcm_voxel_max_point, nc.input_index_sampling = c.i_pcm_voxel_max_point, c.i_input_index_sampling
        nc.lidar_time_delay, nc.pcm_voxel_size = c.d_lidar_time_delay, c.d_pcm_voxel_size
        nc.input_max_dist, nc.input_voxel_ds_m = c.d_input_max_dist, c.d_input_voxel_ds_m
        nc.tf_ego_to_lidar = (C.c_double * 16)(*np.asarray(c.tf_ego_to_lidar, dtype=np.float64).T.ravel())
        return nc

    def CallbackPointCloudNative(self, xyz, point_time, stamp, imu, odom):
        """The same callback through ONE C-ABI call (elm_pcm_callback_point_cloud): no Python between the stages."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(point_time, dtype=np.float32)
        imu = np.ascontiguousarray(imu, dtype=np.float64).reshape(-1, 4)
        odom = np.ascontiguousarray(odom, dtype=np.float64).reshape(-1, 14)
        if not hasattr(self, "_nc"):
            self._nc = self._node_config()
        out, pub = _lib.PcmScanOutput(), C.c_int(0)
        check(_lib.lib().elm_pcm_callback_point_cloud(self.ctx._h, self.local_map_._h, C.byref(self._nc), C.byref(self.cfg_.registration),
                                                      _fp(xyz), _fp(t), xyz.shape[0], float(stamp), _dp(imu), imu.shape[0], _dp(odom),
                                                      odom.shape[0], C.byref(out), C.byref(pub)), self.ctx._h, "elm_pcm_callback_point_cloud")
        if not pub.value:
            return None
        self.d_time_scan_end_ = out.time_scan_end
        self.d_icp_pose_std_m = out.fitness_score
        return dict(pose_ego=np.array(out.pose_ego).reshape(4, 4).T.copy(), pose_lidar=np.array(out.pose_lidar).reshape(4, 4).T.copy(),
                    fitness=out.fitness_score, time=out.time_scan_end, covariance=np.array(out.covariance).reshape(6, 6),
                    n_source=int(out.n_source))

    def CallbackInitialPose(self, rviz_pose, raw_scan_xyz):
        """pcm.cpp:356-447: ground height under the clicked pose, then RunRegister on the last RAW scan."""
        rviz_pose = np.asarray(rviz_pose, dtype=np.float64)
        found, z_ground = self.local_map_.FindGroundHeight(rviz_pose[:2, 3])
        if not found:
            return None
        ground_pose = rviz_pose.copy()
        ground_pose[2, 3] = z_ground
        init_lidar_pose = ground_pose @ self.cfg_.tf_ego_to_lidar
        src, _ = voxel_downsample(raw_scan_xyz, self.cfg_.d_input_voxel_ds_m)
        pose, ok, fit, cov = self.registration_.RunRegister(src, self.local_map_, init_lidar_pose)
        self.icp_local_cov_ = cov
        final_pose = pose @ np.linalg.inv(self.cfg_.tf_ego_to_lidar)
        if not ok:
            return None
        self.d_icp_pose_std_m = fit
        return dict(pose_ego=final_pose, pose_lidar=pose, fitness=fit, ground_z=z_ground)
