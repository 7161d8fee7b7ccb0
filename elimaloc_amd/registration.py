"""Host-side mirror of the reference's operator interface for the registration hot path.

Same names, argument meaning and error behaviour as
  pcm_matching/include/registration.hpp      (IcpMethod :60, RegistrationConfig :62-85, Registration :101-230)
  pcm_matching/include/voxel_hash_map.hpp    (VoxelHashMap :89-335)
so that parity tests read like calls into the reference.  Everything numerical happens in the C-ABI library
(HIP kernels); this file only marshals numpy arrays.  Out-params of the C++ signatures are returned as tuples.
"""
import ctypes as C
import enum

import numpy as np

from . import _lib
from ._lib import ElmError, RegConfig, RegResult, IterTrace, MapInfo, check


class IcpMethod(enum.IntEnum):  # reg.hpp:60
    P2P = 0
    GICP = 1
    VGICP = 2
    AVGICP = 3


def RegistrationConfig(**kw):
    """RegistrationConfig with the shipped defaults of config/localization.ini:80-105."""
    cfg = RegConfig()
    _lib.lib().elm_reg_config_default(C.byref(cfg))
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise AttributeError(f"RegistrationConfig has no field {k}")
        setattr(cfg, k, int(v) if k in ("icp_method", "max_iteration", "i_max_thread", "use_radar_cov") else v)
    return cfg


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _colmajor16(T):
    return np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(4, 4).T).ravel()


class Context:
    """One GPU, one HIP stream -- or, Context.multi([0, 1, ...]), the LEAD context of a device group: N GPUs inside this process, maps
    replicated, scans sharded, one all-reduce of the packed sums per ICP iteration (elm_ctx_create_multi).  elm_ctx_create fails without a
    gfx950 device."""

    def __init__(self, device_id=0, _devices=None):
        self._h = C.c_void_p()
        if _devices is None:
            check(_lib.lib().elm_ctx_create(device_id, C.byref(self._h)), None, "elm_ctx_create")
        else:
            ids = (C.c_int * len(_devices))(*[int(d) for d in _devices])
            check(_lib.lib().elm_ctx_create_multi(ids, len(_devices), C.byref(self._h)), None, "elm_ctx_create_multi")
            device_id = int(_devices[0])
        self.device_id = device_id
        self._hook_ref = None

    @classmethod
    def multi(cls, device_ids):
        """the lead context of a device group over device_ids (an id may repeat: ranks sharing a GPU exchange through host memory)"""
        return cls(_devices=list(device_ids))

    def group_info(self):
        """(ranks, exchange, device ids): exchange 0 = a plain context, 1 = RCCL, 2 = host memory"""
        n, ex = C.c_int(0), C.c_int(0)
        ids = (C.c_int * 64)()
        check(_lib.lib().elm_ctx_group_info(self._h, C.byref(n), C.byref(ex), ids, 64), self._h, "elm_ctx_group_info")
        return n.value, ex.value, [ids[i] for i in range(n.value)]

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().elm_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(_lib.lib().elm_ctx_synchronize(self._h), self._h, "elm_ctx_synchronize")

    @property
    def stream(self):
        return _lib.lib().elm_ctx_stream(self._h)

    def set_work_counters(self, on=True):
        """n_cand_total / n_occ_total / n_tested_total / fallback_blocks of the results: off by default (they read 0), on = instrumented kernels."""
        check(_lib.lib().elm_ctx_set_work_counters(self._h, int(bool(on))), self._h, "elm_ctx_set_work_counters")

    def set_profiling(self, on=True):
        check(_lib.lib().elm_ctx_set_profiling(self._h, int(bool(on))), self._h, "elm_ctx_set_profiling")

    def get_profile(self, reset=False):
        p = _lib.Profile()
        check(_lib.lib().elm_ctx_get_profile(self._h, C.byref(p), int(bool(reset))), self._h, "elm_ctx_get_profile")
        return dict(accumulate_launches=int(p.accumulate_launches), solve_steps=int(p.solve_steps),
                    accumulate_ms=float(p.accumulate_ms), solve_ms=float(p.solve_ms))

    def measure_h2d(self, host_ptr, nbytes, reps=5):
        """GB/s of a plain host-to-device copy from host_ptr on this box (the PCIe rate a host-fed stream sits under)."""
        g = C.c_double(0.0)
        check(_lib.lib().elm_ctx_measure_h2d(self._h, C.c_void_p(host_ptr), int(nbytes), int(reps), C.byref(g)), self._h, "elm_ctx_measure_h2d")
        return g.value

    # ---- multi-GPU (RCCL over xGMI): one small all-reduce of the packed normal equations per iteration
    @staticmethod
    def comm_unique_id():
        buf = (C.c_char * _lib.COMM_ID_BYTES)()
        check(_lib.lib().elm_comm_get_unique_id(C.cast(buf, C.c_void_p)), None, "elm_comm_get_unique_id")
        return bytes(buf)

    def comm_init(self, rank, nranks, id_bytes):
        buf = (C.c_char * _lib.COMM_ID_BYTES).from_buffer_copy(id_bytes)
        check(_lib.lib().elm_comm_init(self._h, rank, nranks, C.cast(buf, C.c_void_p)), self._h, "elm_comm_init")

    def comm_destroy(self):
        check(_lib.lib().elm_comm_destroy(self._h), self._h, "elm_comm_destroy")

    def comm_info(self):
        """(rank, nranks) as the RCCL communicator reports them (ncclCommUserRank / ncclCommCount); (0, 0) without one."""
        r, n = C.c_int(0), C.c_int(0)
        check(_lib.lib().elm_comm_info(self._h, C.byref(r), C.byref(n)), self._h, "elm_comm_info")
        return r.value, n.value

    def set_allreduce_hook(self, fn):
        """fn(dev_ptr:int, n_doubles:int, hip_stream:int) -> 0 on success; None removes the hook."""
        if fn is None:
            self._hook_ref = _lib.ALLREDUCE_FN(0)
        else:
            self.hook_error = None

            def call(p, n, s, u):
                # an exception must not escape a ctypes callback (it would be printed and swallowed, the exchange reported as done): the
                # call that is exchanging ends with ELM_ERR_COMM "allreduce hook failed" and the exception is kept in hook_error
                try:
                    return int(fn(p, n, s))
                except BaseException as e:  # noqa: BLE001
                    self.hook_error = e
                    return 1
            self._hook_ref = _lib.ALLREDUCE_FN(call)
        check(_lib.lib().elm_comm_set_hook(self._h, self._hook_ref, None), self._h, "elm_comm_set_hook")


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class VoxelHashMap:
    """vhm.hpp:89-335.  Points are (n,3) float32 arrays (the PCD map and the LiDAR are float32, pcm.hpp:205-215)."""

    def __init__(self, voxel_size=1.0, max_points_per_voxel=30, ctx=None):
        self.ctx = ctx or default_context()
        self.voxel_size_ = float(voxel_size)
        self.max_points_per_voxel_ = int(max_points_per_voxel)
        self._pending = []
        self._h = None
        self._want_voxel_cov = False
        self._want_point_cov = None

    def Init(self, voxel_size, max_points_per_voxel):  # vhm.cpp:26-29
        self.voxel_size_ = float(voxel_size)
        self.max_points_per_voxel_ = int(max_points_per_voxel)

    def Clear(self):  # vhm.hpp:324
        self._release()
        self._pending = []

    def _release(self):
        if self._h is not None:
            _lib.lib().elm_map_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def AddPoints(self, points):  # vhm.cpp:270-285
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        if pts.shape[0] == 0:
            return
        self._pending.append(pts)
        self._release()  # rebuilt on next use; sequential AddPoints == one AddPoints of the concatenation

    def Update(self, points, origin=None):  # vhm.cpp:268
        self.AddPoints(points)

    def _handle(self):
        if self._h is None:
            allp = (np.concatenate(self._pending, axis=0) if len(self._pending) > 1 else
                    (self._pending[0] if self._pending else np.zeros((0, 3), np.float32)))
            self._pending = [allp] if allp.shape[0] else []
            h = C.c_void_p()
            check(_lib.lib().elm_map_build(self.ctx._h, _fp(allp), allp.shape[0], self.voxel_size_,
                                           self.max_points_per_voxel_, C.byref(h)), self.ctx._h, "elm_map_build")
            self._h = h
            if self._want_voxel_cov:
                check(_lib.lib().elm_map_cal_voxel_cov_all(self._h), self.ctx._h, "elm_map_cal_voxel_cov_all")
            if self._want_point_cov is not None:
                check(_lib.lib().elm_map_cal_point_cov_all(self._h, self._want_point_cov), self.ctx._h,
                      "elm_map_cal_point_cov_all")
        return self._h

    def CalVoxelCovAll(self):  # vhm.hpp:183-193
        self._want_voxel_cov = True
        if self._h is not None:
            check(_lib.lib().elm_map_cal_voxel_cov_all(self._h), self.ctx._h, "elm_map_cal_voxel_cov_all")
        else:
            self._handle()

    def CalPointCovAll(self, d_search_dist):  # vhm.hpp:252-257
        self._want_point_cov = float(d_search_dist)
        if self._h is not None:
            check(_lib.lib().elm_map_cal_point_cov_all(self._h, float(d_search_dist)), self.ctx._h,
                  "elm_map_cal_point_cov_all")
        else:
            self._handle()

    def BuildNeighbourhoods(self):
        """Pay the neighbourhood-list build (P2P/GICP streaming layout) now instead of on the first registration."""
        check(_lib.lib().elm_map_build_neighbourhoods(self._handle()), self.ctx._h, "elm_map_build_neighbourhoods")

    def Empty(self):  # vhm.hpp:325
        return bool(_lib.lib().elm_map_empty(self._handle()))

    def info(self):
        mi = MapInfo()
        check(_lib.lib().elm_map_get_info(self._handle(), C.byref(mi)), self.ctx._h, "elm_map_get_info")
        return mi

    def Pointcloud(self, with_cov=False):  # vhm.cpp:245-255
        n = int(self.info().n_points)
        xyz = np.empty((n, 3))
        if not with_cov:
            check(_lib.lib().elm_map_download_points(self._handle(), _dp(xyz), None, None, n), self.ctx._h,
                  "elm_map_download_points")
            return xyz
        cov = np.empty((n, 9)); mean = np.empty((n, 3))
        check(_lib.lib().elm_map_download_points(self._handle(), _dp(xyz), _dp(cov), _dp(mean), n), self.ctx._h,
              "elm_map_download_points")
        return xyz, cov.reshape(n, 3, 3).transpose(0, 2, 1).copy(), mean

    def Voxels(self):
        """(stored keys, point counts, covs, means) of every voxel."""
        n = int(self.info().n_voxels)
        key = np.empty((n, 3), np.int32); npts = np.empty(n, np.int32)
        cov = np.empty((n, 9)); mean = np.empty((n, 3))
        check(_lib.lib().elm_map_download_voxels(self._handle(), key.ctypes.data_as(C.POINTER(C.c_int32)),
                                                 npts.ctypes.data_as(C.POINTER(C.c_int32)), _dp(cov), _dp(mean), n),
              self.ctx._h, "elm_map_download_voxels")
        return key, npts, cov.reshape(n, 3, 3).transpose(0, 2, 1).copy(), mean

    def Covariances(self):  # vhm.cpp:257-265: voxels with more than 2 points
        key, npts, cov, mean = self.Voxels()
        sel = npts > 2
        return cov[sel], mean[sel]

    def FindGroundHeight(self, position):  # vhm.hpp:285-322 -> (found, ground_z)
        z = C.c_double(0.0); found = C.c_int(0)
        check(_lib.lib().elm_map_find_ground_height(self._handle(), float(position[0]), float(position[1]),
                                                    C.byref(z), C.byref(found)), self.ctx._h,
              "elm_map_find_ground_height")
        return bool(found.value), z.value

    def _correspondences(self, what, points, max_dist):
        q = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        n = q.shape[0]
        cap = n * (7 if what == 2 else 1)
        src = np.empty(max(cap, 1), np.uint32); tgt = np.empty(max(cap, 1), np.int32)
        k = C.c_size_t(0)
        check(_lib.lib().elm_map_get_correspondences(self.ctx._h, self._handle(), int(what), _dp(q), n, float(max_dist),
                                                     src.ctypes.data_as(C.POINTER(C.c_uint32)), tgt.ctypes.data_as(C.POINTER(C.c_int32)),
                                                     cap, C.byref(k)), self.ctx._h, "elm_map_get_correspondences")
        return q, src[:k.value].astype(np.int64), tgt[:k.value].astype(np.int64)

    def GetCorrespondencePoints(self, points, max_correspondence_dist, indices=False):
        """vhm.cpp:31-88 -> (source points [k, 3], target points [k, 3]) in input order; a point with no neighbour bucket at all pairs
        with the reference's default PointStruct at the origin when that is within range (QUIRK, vhm.cpp:37).  indices=True also returns
        (source index, target index in Pointcloud() order or -1)."""
        q, src, tgt = self._correspondences(0, points, max_correspondence_dist)
        mp = self.Pointcloud()
        target = np.where((tgt >= 0)[:, None], mp[np.maximum(tgt, 0)], 0.0) if tgt.size else np.zeros((0, 3))
        return (q[src], target, src, tgt) if indices else (q[src], target)

    def GetCorrespondencesCov(self, points, max_correspondence_dist, indices=False):
        """vhm.cpp:90-151 -> (source points, target means, target covariances [k, 3, 3]): the nearest voxel MEAN among the occupied
        neighbours; none at all: the default CovStruct (mean 0, covariance I) when the origin is within range."""
        return self._cov_pairs(1, points, max_correspondence_dist, indices)

    def GetCorrespondencesAllCov(self, points, max_correspondence_dist, indices=False):
        """vhm.cpp:153-206 -> one pair per occupied FACE neighbour (and the point's own voxel) whose mean is within range, in the
        reference's neighbour order (vhm.cpp:224-230)."""
        return self._cov_pairs(2, points, max_correspondence_dist, indices)

    def _cov_pairs(self, what, points, max_dist, indices):
        q, src, tgt = self._correspondences(what, points, max_dist)
        _, _, cov, mean = self.Voxels()
        ok = tgt >= 0
        tm = np.where(ok[:, None], mean[np.maximum(tgt, 0)], 0.0) if tgt.size else np.zeros((0, 3))
        tc = np.where(ok[:, None, None], cov[np.maximum(tgt, 0)], np.eye(3)) if tgt.size else np.zeros((0, 3, 3))
        return (q[src], tm, tc, src, tgt) if indices else (q[src], tm, tc)

    def GetAdjacentVoxels(self, point, search_range):  # vhm.cpp:208-243: keys only, whether or not such voxels exist
        v = self.PointToVoxel(point, self.voxel_size_).astype(np.int64)
        if search_range == 0:
            return v[None, :].copy()
        if search_range == 1:
            off = np.array([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]])
        else:  # any other range: the 27 of the 3 x 3 x 3 block, x slowest (voxel_neighbor = 1 whatever `range` says)
            r = np.arange(-1, 2)
            off = np.stack(np.meshgrid(r, r, r, indexing="ij"), -1).reshape(-1, 3)
        return v[None, :] + off

    @staticmethod
    def PointToVoxel(point, voxel_size):  # vhm.hpp:176-180
        return np.floor(np.asarray(point, dtype=np.float64) / voxel_size).astype(np.int32)

    @staticmethod
    def VoxelDownsample(points, voxel_size):  # vhm.hpp:260-283: first point of every floor-keyed voxel
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        keys = np.floor(pts.astype(np.float64) / voxel_size).astype(np.int64)
        _, first = np.unique(keys, axis=0, return_index=True)
        first.sort()
        return pts[first]


class Scan:
    """A device-resident source scan (sensor frame)."""

    def __init__(self, ctx, xyz, n_total=None):
        self.ctx = ctx
        pts = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        self.n = pts.shape[0]
        self._h = C.c_void_p()
        check(_lib.lib().elm_scan_upload(ctx._h, _fp(pts), self.n, self.n if n_total is None else int(n_total),
                                         C.byref(self._h)), ctx._h, "elm_scan_upload")

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().elm_scan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PinnedBuffer:
    """Page-locked host memory (elm_host_alloc) viewed as a float32 numpy array."""

    def __init__(self, n_floats):
        self.ptr = _lib.lib().elm_host_alloc(4 * int(n_floats))
        if not self.ptr:
            raise ElmError("elm_host_alloc failed")
        self.array = np.ctypeslib.as_array((C.c_float * int(n_floats)).from_address(self.ptr))

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            _lib.lib().elm_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _result_dict(r, trace=None):
    d = dict(T=np.array(r.T).reshape(4, 4).T.copy(), is_success=bool(r.is_success), iterations=int(r.iterations),
             gate=int(r.gate), path=int(r.path), fitness_score=float(r.fitness_score), d_fitness=float(r.d_fitness),
             local_cov=np.array(r.local_cov).reshape(6, 6).T.copy(), n_corr_last=float(r.n_corr_last),
             point_iterations=float(r.point_iterations), n_cand_total=float(r.n_cand_total),
             n_occ_total=float(r.n_occ_total), fallback_blocks=float(r.fallback_blocks), n_tested_total=float(r.n_tested_total))
    if trace is not None:
        its = []
        for k in range(min(r.iterations, _lib.MAX_ITER_TRACE)):
            t = trace[k]
            its.append(dict(n_corr=t.n_corr, JTJ=np.array(t.JTJ).reshape(6, 6).T.copy(), JTr=np.array(t.JTr),
                            residual_sum=t.residual_sum, x=np.array(t.x), step_norm=t.step_norm,
                            T=np.array(t.T).reshape(4, 4).T.copy()))
        d["iters"] = its
    return d


def results_from_raw(res):
    return [_result_dict(r) for r in res]


class Registration:
    """reg.hpp:101-230."""

    def __init__(self, config=None, ctx=None):
        self.ctx = ctx or default_context()
        self.config_ = config if config is not None else RegistrationConfig()
        self.d_fitness_score_ = 0.0

    def Init(self, config):  # reg.hpp:104
        self.config_ = config

    def RunRegister(self, source_local, voxel_map, initial_guess, m_config=None, trace=False):
        """reg.cpp:274-418.  Returns (pose 4x4, is_success, fitness_score, local_cov 6x6[, details]).

        fitness_score is None on failure (the reference leaves its out-param untouched, reg.cpp:415)."""
        cfg = m_config if m_config is not None else self.config_
        scan = np.ascontiguousarray(source_local, dtype=np.float32).reshape(-1, 3)
        T0 = _colmajor16(initial_guess)
        Tout = np.empty(16); ok = C.c_int(0); fit = C.c_double(float("nan")); cov = np.empty(36)
        res = RegResult()
        tr = (IterTrace * _lib.MAX_ITER_TRACE)() if trace else None
        check(_lib.lib().elm_register(self.ctx._h, voxel_map._handle(), _fp(scan), scan.shape[0], _dp(T0),
                                      C.byref(cfg), _dp(Tout), C.byref(ok), C.byref(fit), _dp(cov), C.byref(res),
                                      tr), self.ctx._h, "elm_register")
        self.d_fitness_score_ = res.d_fitness
        pose = Tout.reshape(4, 4).T.copy()
        out = (pose, bool(ok.value), (fit.value if ok.value else None), cov.reshape(6, 6).T.copy())
        if trace:
            return out + (_result_dict(res, tr),)
        return out

    def _align(self, method, source_local, target_xyz, target_cov, last_icp_pose, trans_th, m_config, source_cov=None):
        cfg = m_config if m_config is not None else self.config_
        src = np.ascontiguousarray(source_local, dtype=np.float64).reshape(-1, 3)
        tgt = np.ascontiguousarray(target_xyz, dtype=np.float64).reshape(-1, 3)
        n = src.shape[0]
        colmajor = lambda c: None if c is None else np.ascontiguousarray(np.asarray(c, dtype=np.float64).reshape(n, 3, 3).transpose(0, 2, 1)).reshape(n, 9)
        tc, sc = colmajor(target_cov), colmajor(source_cov)
        T = _colmajor16(last_icp_pose)
        Tout = np.empty(16); cov = np.zeros(36); fit = C.c_double(float("nan"))
        check(_lib.lib().elm_align_clouds_local(self.ctx._h, int(method), _dp(src), _dp(tgt), None if tc is None else _dp(tc),
                                                None if sc is None else _dp(sc), n, _dp(T), float(trans_th), C.byref(cfg), _dp(Tout),
                                                _dp(cov), C.byref(fit)), self.ctx._h, "elm_align_clouds_local")
        self.d_fitness_score_ = fit.value
        return Tout.reshape(4, 4).T.copy(), cov.reshape(6, 6)

    @staticmethod
    def CalFramePointCov(points, range_var_m, azim_var_deg, ele_var_deg):
        """reg.hpp:186-217: the covariance term R S of every point (from its position: the MAP frame under the initial guess at the
        reference's call site, reg.cpp:302-305) -> [n, 3, 3], not symmetric.  Host arithmetic (elm_cal_frame_point_cov)."""
        q = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        n = q.shape[0]
        cov = np.empty((max(n, 1), 9))
        rc = _lib.lib().elm_cal_frame_point_cov(_dp(q), n, float(range_var_m), float(azim_var_deg), float(ele_var_deg), _dp(cov))
        if rc != 0:
            raise ElmError("elm_cal_frame_point_cov failed")
        return cov[:n].reshape(n, 3, 3).transpose(0, 2, 1).copy()

    def AlignCloudsLocal(self, source_local, target_pose, last_icp_pose, trans_th, m_config=None):
        """reg.cpp:15-66 on explicit pairs (source PointStruct::local, target PointStruct::pose) -> the step as a 4x4."""
        return self._align(IcpMethod.P2P, source_local, target_pose, None, last_icp_pose, trans_th, m_config)[0]

    def AlignCloudsLocalPointCov(self, source_local, target_mean, target_cov, last_icp_pose, trans_th, m_config=None, source_cov=None):
        """reg.cpp:68-152 (targets' covariance.mean / covariance.cov) -> (step 4x4, local_cov 6x6)."""
        return self._align(IcpMethod.GICP, source_local, target_mean, target_cov, last_icp_pose, trans_th, m_config, source_cov)

    def AlignCloudsLocalVoxelCov(self, source_local, target_mean, target_cov, last_icp_pose, trans_th, m_config=None, source_cov=None):
        """reg.cpp:154-225 (CovStruct targets) -> the step as a 4x4."""
        return self._align(IcpMethod.VGICP, source_local, target_mean, target_cov, last_icp_pose, trans_th, m_config, source_cov)[0]

    def RunRegisterBatch(self, scans, voxel_map, initial_guesses, m_config=None, trace=False):
        """Many resident scans against one map, iterated together. Returns a list of result dicts."""
        cfg = m_config if m_config is not None else self.config_
        B = len(scans)
        arr, T0 = self.pack_inputs(scans, initial_guesses)
        res = (RegResult * B)()
        tr = (IterTrace * (_lib.MAX_ITER_TRACE * B))() if trace else None
        check(_lib.lib().elm_register_batch(self.ctx._h, voxel_map._handle(), arr, B, _dp(T0), C.byref(cfg), res, tr), self.ctx._h, "elm_register_batch")
        return [_result_dict(res[b], tr[b * _lib.MAX_ITER_TRACE:(b + 1) * _lib.MAX_ITER_TRACE] if trace else None) for b in range(B)]

    @staticmethod
    def pack_inputs(scans, initial_guesses):
        """(ctypes array of scan handles, contiguous column-major float64 guesses) as the C ABI takes them."""
        arr = (C.c_void_p * len(scans))(*[s._h for s in scans])
        T0 = np.ascontiguousarray(np.asarray(initial_guesses, dtype=np.float64).reshape(-1, 4, 4).transpose(0, 2, 1)).reshape(-1)
        return arr, T0

    def RunRegisterStream(self, scans, voxel_map, initial_guesses, slots=32, m_config=None, trace=False, raw=False):
        """Continuous batching (elm_register_stream): len(scans) registrations through `slots` device slots that are
        refilled on the device as registrations finish.  Same results as RunRegisterBatch, in input order."""
        cfg = m_config if m_config is not None else self.config_
        arr, T0 = scans, initial_guesses
        if not isinstance(scans, C.Array):  # pack_inputs() lets a caller marshal once and call many times
            arr, T0 = self.pack_inputs(scans, initial_guesses)
        B = len(arr)
        res = (RegResult * B)()
        tr = (IterTrace * (_lib.MAX_ITER_TRACE * B))() if trace else None
        check(_lib.lib().elm_register_stream(self.ctx._h, voxel_map._handle(), arr, B, _dp(T0), C.byref(cfg), int(slots), res, tr),
              self.ctx._h, "elm_register_stream")
        if raw:  # the elm_reg_result array as the library filled it; results_from_raw() turns it into dicts later
            return res
        return [_result_dict(res[b], tr[b * _lib.MAX_ITER_TRACE:(b + 1) * _lib.MAX_ITER_TRACE] if trace else None) for b in range(B)]

    @staticmethod
    def pack_host_inputs(scans_host, initial_guesses, pinned=None):
        """Marshal HOST scans for RunRegisterStreamHost once: (pointer array, point counts, column-major guesses, keep-alive).

        pinned: optional PinnedBuffer holding all scans back to back (page-locked: the DMA engines read it directly)."""
        B = len(scans_host)
        keep = [np.ascontiguousarray(s, dtype=np.float32).reshape(-1, 3) for s in scans_host]
        npts = (C.c_uint32 * B)(*[k.shape[0] for k in keep])
        if pinned is not None:
            o = 0
            ptrs = []
            for k in keep:
                pinned.array[o:o + k.size] = k.ravel()
                ptrs.append(pinned.ptr + 4 * o)
                o += k.size
            keep = [pinned]
        else:
            ptrs = [k.ctypes.data for k in keep]
        arr = (C.c_void_p * B)(*ptrs)
        T0 = np.ascontiguousarray(np.asarray(initial_guesses, dtype=np.float64).reshape(-1, 4, 4).transpose(0, 2, 1)).reshape(-1)
        return arr, npts, T0, keep

    def RunRegisterStreamHost(self, packed, voxel_map, slots=32, m_config=None, trace=False, raw=False):
        """elm_register_stream_host: the scans are in host memory when the call starts; uploads, device-side ordering and the
        iterations of earlier registrations overlap.  packed = pack_host_inputs(...)."""
        cfg = m_config if m_config is not None else self.config_
        arr, npts, T0, _keep = packed
        B = len(arr)
        res = (RegResult * B)()
        tr = (IterTrace * (_lib.MAX_ITER_TRACE * B))() if trace else None
        check(_lib.lib().elm_register_stream_host(self.ctx._h, voxel_map._handle(), arr, npts, B, _dp(T0), C.byref(cfg), int(slots), res, tr),
              self.ctx._h, "elm_register_stream_host")
        if raw:
            return res
        return [_result_dict(res[b], tr[b * _lib.MAX_ITER_TRACE:(b + 1) * _lib.MAX_ITER_TRACE] if trace else None) for b in range(B)]

    def EnqueueBatch(self, scans, voxel_map, initial_guesses, m_config=None, trace=False):
        cfg = m_config if m_config is not None else self.config_
        B = len(scans)
        arr = (C.c_void_p * B)(*[s._h for s in scans])
        T0 = np.concatenate([_colmajor16(T) for T in initial_guesses])
        self._pending = (B, trace)
        check(_lib.lib().elm_register_batch_enqueue(self.ctx._h, voxel_map._handle(), arr, B, _dp(T0),
                                                    C.byref(cfg), int(bool(trace))), self.ctx._h,
              "elm_register_batch_enqueue")

    def FinishBatch(self):
        B, trace = self._pending
        res = (RegResult * B)()
        tr = (IterTrace * (_lib.MAX_ITER_TRACE * B))() if trace else None
        check(_lib.lib().elm_register_batch_finish(self.ctx._h, res, tr), self.ctx._h, "elm_register_batch_finish")
        out = []
        for b in range(B):
            t = tr[b * _lib.MAX_ITER_TRACE:(b + 1) * _lib.MAX_ITER_TRACE] if trace else None
            out.append(_result_dict(res[b], t))
        return out

    @staticmethod
    def TransformPoints(T, points):  # reg.hpp:126-148 (host-side convenience, float64)
        p = np.asarray(points, dtype=np.float64).reshape(-1, 3)
        T = np.asarray(T, dtype=np.float64)
        return p @ T[:3, :3].T + T[:3, 3]
