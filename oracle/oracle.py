"""ctypes binding of the CPU ORACLE (oracle/libelm_oracle.so).

TEST INFRASTRUCTURE, NOT PRODUCT: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module, and only as the checker / reported CPU baseline.
PARITY UNPINNED: see oracle/elm_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

P2P, GICP, VGICP, AVGICP = 0, 1, 2, 3
MAX_ITER_TRACE = 64


class Config(C.Structure):
    _fields_ = [
        ("icp_method", C.c_int32),
        ("max_iteration", C.c_int32),
        ("max_thread", C.c_int32),
        ("use_radar_cov", C.c_int32),
        ("max_search_dist", C.c_double),
        ("lm_lambda", C.c_double),
        ("icp_termination_threshold_m", C.c_double),
        ("min_overlap_ratio", C.c_double),
        ("max_fitness_score", C.c_double),
        ("gicp_cov_search_dist", C.c_double),
        ("range_variance_m", C.c_double),
        ("azimuth_variance_deg", C.c_double),
        ("elevation_variance_deg", C.c_double),
    ]


class IterTrace(C.Structure):
    _fields_ = [
        ("n_corr", C.c_int64),
        ("JTJ", C.c_double * 36),
        ("JTr", C.c_double * 6),
        ("residual_sum", C.c_double),
        ("x", C.c_double * 6),
        ("step_norm", C.c_double),
        ("T", C.c_double * 16),
        ("n_cand", C.c_int64),
        ("n_occ", C.c_int64),
    ]


class Result(C.Structure):
    _fields_ = [
        ("T", C.c_double * 16),
        ("is_success", C.c_int32),
        ("iterations", C.c_int32),
        ("gate", C.c_int32),
        ("_pad", C.c_int32),
        ("fitness", C.c_double),
        ("local_cov", C.c_double * 36),
        ("elapsed_ms", C.c_double),
        ("correspondence_ms", C.c_double),
        ("iters", IterTrace * MAX_ITER_TRACE),
    ]


class DeskewTables(C.Structure):
    _fields_ = [
        ("time_scan_cur", C.c_double),
        ("time_scan_end", C.c_double),
        ("imu_pointer_cur", C.c_int32),
        ("run_deskew", C.c_int32),
        ("odom_incre_x", C.c_float),
        ("odom_incre_y", C.c_float),
        ("odom_incre_z", C.c_float),
        ("odom_available", C.c_int32),
        ("imu_time", C.POINTER(C.c_double)),
        ("imu_rot_x", C.POINTER(C.c_double)),
        ("imu_rot_y", C.POINTER(C.c_double)),
        ("imu_rot_z", C.POINTER(C.c_double)),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libelm_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("elm_oracle.cpp", "elm_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "libelm_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("ELM_ORACLE_LIB") or os.path.join(_HERE, "libelm_oracle.so")  # (override: the sanitizer flavour, tools/sanitize.sh)
        if not os.path.exists(so):
            so = build()
        L = C.CDLL(so)
        dp = C.POINTER(C.c_double)
        fp = C.POINTER(C.c_float)
        L.orc_map_create.restype = C.c_void_p
        L.orc_map_create.argtypes = [C.c_double, C.c_int]
        L.orc_map_destroy.argtypes = [C.c_void_p]
        L.orc_map_add_points.argtypes = [C.c_void_p, fp, C.c_size_t]
        L.orc_map_cal_voxel_cov_all.argtypes = [C.c_void_p, C.c_int]
        L.orc_map_cal_point_cov_all.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.orc_map_num_points.restype = C.c_size_t
        L.orc_map_num_points.argtypes = [C.c_void_p]
        L.orc_map_num_voxels.restype = C.c_size_t
        L.orc_map_num_voxels.argtypes = [C.c_void_p]
        L.orc_map_empty.argtypes = [C.c_void_p]
        L.orc_map_pointcloud.restype = C.c_size_t
        L.orc_map_pointcloud.argtypes = [C.c_void_p, dp, dp, dp, C.c_size_t]
        L.orc_map_voxels.restype = C.c_size_t
        L.orc_map_voxels.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp, dp, C.c_size_t]
        L.orc_map_find_ground_height.argtypes = [C.c_void_p, C.c_double, C.c_double, dp]
        L.orc_nearest_points.argtypes = [C.c_void_p, dp, C.c_size_t, C.c_double, C.c_int,
                                         C.POINTER(C.c_uint8), dp, dp]
        L.orc_nearest_voxel.argtypes = [C.c_void_p, dp, C.c_size_t, C.c_double, C.c_int,
                                        C.POINTER(C.c_uint8), dp, dp]
        L.orc_register.argtypes = [C.c_void_p, fp, C.c_size_t, dp, C.POINTER(Config), C.POINTER(Result)]
        L.orc_voxel_downsample.restype = C.c_size_t
        L.orc_voxel_downsample.argtypes = [fp, C.c_size_t, C.c_double, C.POINTER(C.c_int64)]
        L.orc_deskew_points.argtypes = [fp, fp, C.c_size_t, C.POINTER(DeskewTables), fp]
        L.orc_imu_deskew_info.argtypes = [dp, dp, C.c_size_t, C.c_double, C.c_double, dp, dp, dp, dp,
                                          C.POINTER(C.c_int32)]
        L.orc_odom_deskew_info.argtypes = [dp, C.c_size_t, C.c_double, C.c_double, fp]
        L.orc_filter_points_by_distance.restype = C.c_size_t
        L.orc_filter_points_by_distance.argtypes = [fp, C.c_size_t, C.c_double, C.POINTER(C.c_int64)]
        L.orc_get_interpolated_pose.argtypes = [dp, C.c_size_t, C.c_double, fp]
        L.orc_shape_odom_covariance.argtypes = [dp, dp, C.c_double, dp]
        L.orc_ldlt_solve6.argtypes = [dp, dp, dp]
        L.orc_inverse6.argtypes = [dp, dp]
        L.orc_jacobi_svd3.argtypes = [dp, dp, dp, dp]
        L.orc_smallest_eigenvector.argtypes = [dp, dp]
        L.orc_angle_axis_to_matrix.argtypes = [dp, dp]
        L.orc_matrix_to_angle.restype = C.c_double
        L.orc_matrix_to_angle.argtypes = [dp]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def default_config(method=GICP, **kw):
    """Defaults = config/localization.ini:83-105 of the reference."""
    c = Config(icp_method=method, max_iteration=10, max_thread=10, use_radar_cov=0, max_search_dist=5.0,
               lm_lambda=0.5, icp_termination_threshold_m=0.02, min_overlap_ratio=0.4, max_fitness_score=0.5,
               gicp_cov_search_dist=0.4, range_variance_m=1.0, azimuth_variance_deg=0.4, elevation_variance_deg=0.4)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


class Map:
    """VoxelHashMap restatement handle."""

    def __init__(self, voxel_size=1.0, max_points=30):
        self._h = lib().orc_map_create(voxel_size, max_points)
        self.voxel_size = voxel_size
        self.max_points = max_points

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_map_destroy(self._h)
            self._h = None

    def add_points(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        lib().orc_map_add_points(self._h, _fp(xyz), xyz.shape[0])

    def cal_voxel_cov_all(self, threads=8):
        lib().orc_map_cal_voxel_cov_all(self._h, threads)

    def cal_point_cov_all(self, dist=0.4, threads=8):
        lib().orc_map_cal_point_cov_all(self._h, dist, threads)

    @property
    def num_points(self):
        return lib().orc_map_num_points(self._h)

    @property
    def num_voxels(self):
        return lib().orc_map_num_voxels(self._h)

    def empty(self):
        return bool(lib().orc_map_empty(self._h))

    def pointcloud(self):
        n = self.num_points
        xyz = np.empty((n, 3)); cov = np.empty((n, 9)); mean = np.empty((n, 3))
        lib().orc_map_pointcloud(self._h, _dp(xyz), _dp(cov), _dp(mean), n)
        # cov rows are column-major 3x3
        return xyz, cov.reshape(n, 3, 3).transpose(0, 2, 1).copy(), mean

    def voxels(self):
        n = self.num_voxels
        key = np.empty((n, 3), np.int32); npts = np.empty(n, np.int32)
        cov = np.empty((n, 9)); mean = np.empty((n, 3))
        lib().orc_map_voxels(self._h, key.ctypes.data_as(C.POINTER(C.c_int32)),
                             npts.ctypes.data_as(C.POINTER(C.c_int32)), _dp(cov), _dp(mean), n)
        return key, npts, cov.reshape(n, 3, 3).transpose(0, 2, 1).copy(), mean

    def find_ground_height(self, x, y):
        z = C.c_double(0.0)
        ok = lib().orc_map_find_ground_height(self._h, x, y, C.byref(z))
        return bool(ok), z.value

    def nearest_points(self, q, max_dist=5.0, threads=8):
        q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, 3)
        n = q.shape[0]
        acc = np.empty(n, np.uint8); tgt = np.empty((n, 3)); d2 = np.empty(n)
        lib().orc_nearest_points(self._h, _dp(q), n, max_dist, threads,
                                 acc.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(tgt), _dp(d2))
        return acc.astype(bool), tgt, d2

    def nearest_voxel(self, q, max_dist=5.0, threads=8):
        q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, 3)
        n = q.shape[0]
        acc = np.empty(n, np.uint8); mean = np.empty((n, 3)); cov = np.empty((n, 9))
        lib().orc_nearest_voxel(self._h, _dp(q), n, max_dist, threads,
                                acc.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(mean), _dp(cov))
        return acc.astype(bool), mean, cov.reshape(n, 3, 3).transpose(0, 2, 1).copy()


    def all_cov_pairs(self, q, max_dist=5.0):
        """GetCorrespondencesAllCov (vhm.cpp:153-206): (source index, target mean, target covariance) per pair, input order."""
        q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, 3)
        n = q.shape[0]
        cap = 7 * n
        src = np.empty(max(cap, 1), np.uint32); mean = np.empty((max(cap, 1), 3)); cov = np.empty((max(cap, 1), 9))
        f = lib().orc_all_cov_pairs
        f.restype = C.c_size_t
        f.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.c_double, C.POINTER(C.c_uint32), C.POINTER(C.c_double),
                      C.POINTER(C.c_double), C.c_size_t]
        k = f(self._h, _dp(q), n, max_dist, src.ctypes.data_as(C.POINTER(C.c_uint32)), _dp(mean), _dp(cov), cap)
        return src[:k].astype(np.int64), mean[:k].copy(), cov[:k].reshape(k, 3, 3).transpose(0, 2, 1).copy()


def register(m, scan_xyz, T0, cfg):
    """RunRegister restatement. T0: 4x4 numpy (row/col indexed normally). Returns dict."""
    scan = np.ascontiguousarray(scan_xyz, dtype=np.float32).reshape(-1, 3)
    T0c = np.asfortranarray(np.asarray(T0, dtype=np.float64)).ravel(order="F").copy()
    res = Result()
    lib().orc_register(m._h, _fp(scan), scan.shape[0], _dp(T0c), C.byref(cfg), C.byref(res))
    iters = []
    for k in range(min(res.iterations, MAX_ITER_TRACE)):
        it = res.iters[k]
        iters.append(dict(
            n_corr=it.n_corr,
            JTJ=np.array(it.JTJ).reshape(6, 6, order="F"),
            JTr=np.array(it.JTr),
            residual_sum=it.residual_sum,
            x=np.array(it.x),
            step_norm=it.step_norm,
            T=np.array(it.T).reshape(4, 4, order="F"),
            n_cand=it.n_cand, n_occ=it.n_occ))
    return dict(T=np.array(res.T).reshape(4, 4, order="F"), is_success=bool(res.is_success),
                iterations=res.iterations, gate=res.gate, fitness=res.fitness,
                local_cov=np.array(res.local_cov).reshape(6, 6, order="F"),
                elapsed_ms=res.elapsed_ms, correspondence_ms=res.correspondence_ms, iters=iters)


def align_clouds_local(method, src_local, tgt_xyz, tgt_cov, last_icp_pose, trans_th, cfg, src_cov=None):
    """Registration::AlignCloudsLocal (method 0) / AlignCloudsLocalPointCov (1) / AlignCloudsLocalVoxelCov (2, 3), reg.cpp:15-225, on
    explicit pairs -> dict(T 4x4, local_cov 6x6, fitness, JTJ 6x6, JTr)."""
    src = np.ascontiguousarray(src_local, dtype=np.float64).reshape(-1, 3)
    tgt = np.ascontiguousarray(tgt_xyz, dtype=np.float64).reshape(-1, 3)
    n = src.shape[0]
    colmajor = lambda c: None if c is None else np.ascontiguousarray(np.asarray(c, dtype=np.float64).reshape(n, 3, 3).transpose(0, 2, 1)).reshape(n, 9)
    tc, sc = colmajor(tgt_cov), colmajor(src_cov)
    T = np.asfortranarray(np.asarray(last_icp_pose, dtype=np.float64)).ravel(order="F").copy()
    Tout = np.empty(16); cov = np.empty(36); fit = C.c_double(0.0); JTJ = np.empty(36); JTr = np.empty(6)
    f = lib().orc_align_clouds_local
    f.restype = None
    dp = C.POINTER(C.c_double)
    f.argtypes = [C.c_int, dp, dp, dp, dp, C.c_size_t, dp, C.c_double, C.POINTER(Config), dp, dp, C.POINTER(C.c_double), dp, dp]
    f(int(method), _dp(src), _dp(tgt), None if tc is None else _dp(tc), None if sc is None else _dp(sc), n, _dp(T), float(trans_th),
      C.byref(cfg), _dp(Tout), _dp(cov), C.byref(fit), _dp(JTJ), _dp(JTr))
    return dict(T=Tout.reshape(4, 4).T.copy(), local_cov=cov.reshape(6, 6).T.copy(), fitness=fit.value, JTJ=JTJ.reshape(6, 6).T.copy(), JTr=JTr)


def cal_frame_point_cov(xyz, range_var_m, azim_var_deg, ele_var_deg):
    """Registration::CalFramePointCov (reg.hpp:186-217): the R S term per point -> [n, 3, 3]."""
    q = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    n = q.shape[0]
    cov = np.empty((max(n, 1), 9))
    f = lib().orc_cal_frame_point_cov
    f.restype = None
    f.argtypes = [C.POINTER(C.c_double), C.c_size_t, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double)]
    f(_dp(q), n, float(range_var_m), float(azim_var_deg), float(ele_var_deg), _dp(cov))
    return cov[:n].reshape(n, 3, 3).transpose(0, 2, 1).copy()


def voxel_downsample(xyz, voxel_size):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    keep = np.empty(xyz.shape[0], np.int64)
    k = lib().orc_voxel_downsample(_fp(xyz), xyz.shape[0], voxel_size, keep.ctypes.data_as(C.POINTER(C.c_int64)))
    return keep[:k].copy()


def deskew_points(xyz, rel_time, imu_time, imu_rot, scan_cur, scan_end, odom_incre, run_deskew=True,
                  odom_available=True):
    """DeskewPoint loop. imu_rot: (k,3) integrated rotation table; imu_pointer_cur = k-1."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    rt = np.ascontiguousarray(rel_time, dtype=np.float32)
    it = np.ascontiguousarray(imu_time, dtype=np.float64)
    rx = np.ascontiguousarray(imu_rot[:, 0], dtype=np.float64)
    ry = np.ascontiguousarray(imu_rot[:, 1], dtype=np.float64)
    rz = np.ascontiguousarray(imu_rot[:, 2], dtype=np.float64)
    tab = DeskewTables(scan_cur, scan_end, len(it) - 1, int(run_deskew), odom_incre[0], odom_incre[1],
                       odom_incre[2], int(odom_available), _dp(it), _dp(rx), _dp(ry), _dp(rz))
    out = np.empty_like(xyz)
    lib().orc_deskew_points(_fp(xyz), _fp(rt), xyz.shape[0], C.byref(tab), _fp(out))
    return out


def imu_deskew_info(imu_t, imu_w, scan_cur, scan_end):
    imu_t = np.ascontiguousarray(imu_t, dtype=np.float64)
    imu_w = np.ascontiguousarray(imu_w, dtype=np.float64).reshape(-1, 3)
    tt = np.zeros(2000); rx = np.zeros(2000); ry = np.zeros(2000); rz = np.zeros(2000)
    cur = C.c_int32(0)
    ok = lib().orc_imu_deskew_info(_dp(imu_t), _dp(imu_w), len(imu_t), scan_cur, scan_end, _dp(tt), _dp(rx),
                                   _dp(ry), _dp(rz), C.byref(cur))
    k = cur.value + 1
    return bool(ok), tt[:k].copy(), np.stack([rx[:k], ry[:k], rz[:k]], axis=1)


def odom_deskew_info(odom14, scan_cur, scan_end):
    od = np.ascontiguousarray(odom14, dtype=np.float64).reshape(-1, 14)
    inc = np.zeros(3, np.float32)
    ok = lib().orc_odom_deskew_info(_dp(od), od.shape[0], scan_cur, scan_end, _fp(inc))
    return bool(ok), inc


def ldlt_solve6(A, b):
    A = np.asarray(A, dtype=np.float64).ravel(order="F").copy(); b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.empty(6)
    lib().orc_ldlt_solve6(_dp(A), _dp(b), _dp(x))
    return x


def inverse6(A):
    A = np.asarray(A, dtype=np.float64).ravel(order="F").copy()
    out = np.empty(36)
    lib().orc_inverse6(_dp(A), _dp(out))
    return out.reshape(6, 6, order="F")


def jacobi_svd3(A):
    A = np.asarray(A, dtype=np.float64).ravel(order="F").copy()
    U = np.empty(9); S = np.empty(3); V = np.empty(9)
    lib().orc_jacobi_svd3(_dp(A), _dp(U), _dp(S), _dp(V))
    return U.reshape(3, 3, order="F"), S, V.reshape(3, 3, order="F")


def smallest_eigenvector(C):
    C = np.asarray(C, dtype=np.float64).ravel(order="F").copy()
    n = np.empty(3)
    lib().orc_smallest_eigenvector(_dp(C), _dp(n))
    return n


def angle_axis_to_matrix(v):
    v = np.ascontiguousarray(v, dtype=np.float64)
    R = np.empty(9)
    lib().orc_angle_axis_to_matrix(_dp(v), _dp(R))
    return R.reshape(3, 3, order="F")


def matrix_to_angle(R):
    R = np.asarray(R, dtype=np.float64).ravel(order="F").copy()
    return lib().orc_matrix_to_angle(_dp(R))


def filter_points_by_distance(xyz, max_dist):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    keep = np.empty(xyz.shape[0], np.int64)
    k = lib().orc_filter_points_by_distance(_fp(xyz), xyz.shape[0], max_dist, keep.ctypes.data_as(C.POINTER(C.c_int64)))
    return keep[:k].copy()


def get_interpolated_pose(odom14, t):
    od = np.ascontiguousarray(odom14, dtype=np.float64).reshape(-1, 14)
    T = np.zeros(16, np.float32)
    ok = lib().orc_get_interpolated_pose(_dp(od), od.shape[0], t, _fp(T))
    return bool(ok), T.reshape(4, 4).T.copy()


def shape_odom_covariance(local_cov, pose, std_m):
    lc = np.asarray(local_cov, dtype=np.float64).ravel(order="F").copy()
    ps = np.asarray(pose, dtype=np.float64).ravel(order="F").copy()
    out = np.zeros(36)
    lib().orc_shape_odom_covariance(_dp(lc), _dp(ps), std_m, _dp(out))
    return out.reshape(6, 6)
