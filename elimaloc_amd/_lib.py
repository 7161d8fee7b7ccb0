"""ctypes loader of the C-ABI library (include/elimaloc_hip.h -> elimaloc_amd/libelimaloc_hip.so).

There is no CPU fallback: a missing library or a missing gfx950 device raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ELM_LIB") or os.path.join(_HERE, "libelimaloc_hip.so")  # ELM_LIB: developer A/B of prebuilt variants

ELM_OK = 0
P2P, GICP, VGICP, AVGICP = 0, 1, 2, 3
MAX_ITER_TRACE = 64
PACKED_SUMS = 32
COMM_ID_BYTES = 128


class ElmError(RuntimeError):
    pass


class RegConfig(C.Structure):
    """POD mirror of RegistrationConfig (reg.hpp:62-85)."""
    _fields_ = [
        ("i_max_thread", C.c_int32),
        ("icp_method", C.c_int32),
        ("voxel_search_method", C.c_int32),
        ("use_radar_cov", C.c_int32),
        ("max_iteration", C.c_int32),
        ("b_debug_print", C.c_int32),
        ("gicp_cov_search_dist", C.c_double),
        ("max_search_dist", C.c_double),
        ("lm_lambda", C.c_double),
        ("icp_termination_threshold_m", C.c_double),
        ("min_overlap_ratio", C.c_double),
        ("max_fitness_score", C.c_double),
        ("doppler_trans_lambda", C.c_double),
        ("range_variance_m", C.c_double),
        ("azimuth_variance_deg", C.c_double),
        ("elevation_variance_deg", C.c_double),
        ("ego_to_lidar_trans", C.c_double * 3),
        ("ego_to_lidar_rot", C.c_double * 9),
        ("ego_to_imu_rot", C.c_double * 9),
    ]


class IterTrace(C.Structure):
    _fields_ = [
        ("JTJ", C.c_double * 36),
        ("JTr", C.c_double * 6),
        ("residual_sum", C.c_double),
        ("n_corr", C.c_double),
        ("x", C.c_double * 6),
        ("step_norm", C.c_double),
        ("T", C.c_double * 16),
    ]


class RegResult(C.Structure):
    _fields_ = [
        ("T", C.c_double * 16),
        ("fitness_score", C.c_double),
        ("d_fitness", C.c_double),
        ("local_cov", C.c_double * 36),
        ("is_success", C.c_int32),
        ("iterations", C.c_int32),
        ("gate", C.c_int32),
        ("path", C.c_int32),
        ("n_corr_last", C.c_double),
        ("point_iterations", C.c_double),
        ("n_cand_total", C.c_double),
        ("n_occ_total", C.c_double),
        ("fallback_blocks", C.c_double),
        ("n_tested_total", C.c_double),
    ]


class Profile(C.Structure):
    _fields_ = [
        ("accumulate_launches", C.c_uint64),
        ("solve_steps", C.c_uint64),
        ("accumulate_ms", C.c_double),
        ("solve_ms", C.c_double),
    ]


class MapInfo(C.Structure):
    _fields_ = [
        ("n_input_points", C.c_uint64),
        ("n_points", C.c_uint64),
        ("n_voxels", C.c_uint64),
        ("hash_capacity", C.c_uint64),
        ("voxel_size", C.c_double),
        ("max_points_per_voxel", C.c_int32),
        ("has_voxel_cov", C.c_int32),
        ("has_point_cov", C.c_int32),
        ("layout_flags", C.c_int32),
        ("device_bytes", C.c_uint64),
        ("n_query_voxels", C.c_uint64),
        ("nbr_entries", C.c_uint64),
        ("index_bytes", C.c_uint64),
        ("index_part_bytes", C.c_uint64 * 4),
        ("n_list_voxels", C.c_uint64),
    ]


class DeskewTables(C.Structure):
    _fields_ = [
        ("d_time_scan_cur", C.c_double),
        ("d_time_scan_end", C.c_double),
        ("i_imu_pointer_cur", C.c_int32),
        ("b_run_deskew", C.c_int32),
        ("b_is_imu_available", C.c_int32),
        ("b_is_odom_available", C.c_int32),
        ("f_odom_incre_x", C.c_float),
        ("f_odom_incre_y", C.c_float),
        ("f_odom_incre_z", C.c_float),
        ("_pad", C.c_float),
        ("vec_d_imu_time", C.POINTER(C.c_double)),
        ("vec_d_imu_rot_x", C.POINTER(C.c_double)),
        ("vec_d_imu_rot_y", C.POINTER(C.c_double)),
        ("vec_d_imu_rot_z", C.POINTER(C.c_double)),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)

# every symbol include/elimaloc_hip.h declares (checked by the CPU test-suite)
EXPORTS = [
    "elm_reg_config_default", "elm_ctx_create", "elm_ctx_create_multi", "elm_ctx_group_info", "elm_register_shard", "elm_ctx_destroy", "elm_last_error", "elm_strerror",
    "elm_ctx_synchronize", "elm_ctx_stream", "elm_ctx_set_profiling", "elm_ctx_set_work_counters", "elm_ctx_get_profile", "elm_map_build", "elm_map_destroy", "elm_map_cal_voxel_cov_all",
    "elm_map_cal_point_cov_all", "elm_map_build_neighbourhoods", "elm_map_get_info", "elm_map_empty", "elm_map_download_points",
    "elm_map_download_voxels", "elm_map_find_ground_height", "elm_map_get_correspondences", "elm_align_clouds_local", "elm_scan_upload", "elm_scan_destroy",
    "elm_scan_size", "elm_scan_download", "elm_register", "elm_format_register_log", "elm_register_batch", "elm_register_stream", "elm_register_stream_host", "elm_host_alloc", "elm_host_free", "elm_ctx_measure_h2d", "elm_register_batch_enqueue",
    "elm_register_batch_finish", "elm_deskew", "elm_deskew_downsample", "elm_deskew_prepare", "elm_comm_get_unique_id", "elm_comm_init",
    "elm_comm_destroy", "elm_comm_info", "elm_comm_set_hook", "elm_filter_points_by_distance", "elm_voxel_downsample",
    "elm_get_interpolated_pose", "elm_shape_odom_covariance", "elm_cal_frame_point_cov",
    "elm_ekf_config_default", "elm_ekf_create", "elm_ekf_destroy", "elm_ekf_predict_imu", "elm_ekf_predict", "elm_ekf_update_can", "elm_gps_project", "elm_ekf_update_navsatfix",
    "elm_ekf_update_pose", "elm_ekf_update_pcm_odom", "elm_ekf_get_state", "elm_ekf_publish",
    "elm_ini_load", "elm_ini_destroy", "elm_ini_get_string", "elm_ini_get_int", "elm_ini_get_bool", "elm_ini_get_double",
    "elm_ini_get_array", "elm_pcm_node_config_default", "elm_load_pcm_config", "elm_load_ekf_config", "elm_pcd_load_xyz",
    "elm_free", "elm_scan_from_cloud", "elm_pcm_callback_point_cloud",
]


class PcmNodeConfig(C.Structure):
    """elm_pcm_node_config (include/elimaloc_hip.h)."""
    _fields_ = [("lidar_type", C.c_char * 32), ("lidar_scan_time_end", C.c_int32), ("pcm_voxel_max_point", C.c_int32),
                ("run_deskew", C.c_int32), ("input_index_sampling", C.c_int32), ("lidar_time_delay", C.c_double),
                ("pcm_voxel_size", C.c_double), ("input_max_dist", C.c_double), ("input_voxel_ds_m", C.c_double),
                ("tf_ego_to_lidar", C.c_double * 16)]


class PcmScanOutput(C.Structure):
    """elm_pcm_scan_output (include/elimaloc_hip.h)."""
    _fields_ = [("pose_ego", C.c_double * 16), ("pose_lidar", C.c_double * 16), ("covariance", C.c_double * 36),
                ("fitness_score", C.c_double), ("time_scan_end", C.c_double), ("n_filtered", C.c_uint64), ("n_source", C.c_uint64),
                ("result", RegResult)]


class CloudField(C.Structure):
    _fields_ = [("name", C.c_char * 24), ("offset", C.c_uint32), ("datatype", C.c_int32)]


FIELD_UINT16, FIELD_UINT32, FIELD_FLOAT32 = 4, 6, 7


class EkfConfig(C.Structure):
    """elm_ekf_config (include/elimaloc_hip.h)."""
    _fields_ = [("imu_gravity", C.c_double)] + [(n, C.c_int32) for n in (
        "imu_estimate_gravity", "imu_estimate_calibration", "use_zupt", "use_complementary_filter", "gps_type", "_pad")] + [
        (n, C.c_double) for n in (
            "ekf_init_x_m", "ekf_init_y_m", "ekf_init_z_m", "ekf_init_roll_deg", "ekf_init_pitch_deg", "ekf_init_yaw_deg",
            "state_std_pos_m", "state_std_rot_deg", "state_std_vel_mps", "state_std_gyro_dps", "state_std_acc_mps",
            "imu_std_gyro_dps", "imu_std_acc_mps", "ekf_imu_bias_cov_gyro", "ekf_imu_bias_cov_acc",
            "gnss_min_cov_x_m", "gnss_min_cov_y_m", "gnss_min_cov_z_m", "gnss_min_cov_roll_deg", "gnss_min_cov_pitch_deg",
            "gnss_min_cov_yaw_deg", "can_vel_scale_factor", "ekf_can_meas_uncertainty_vel_mps",
            "ekf_can_meas_uncertainty_yaw_rate_deg")]


class EkfStateC(C.Structure):
    _fields_ = [("x", C.c_double * 27), ("rot_xyzw", C.c_double * 4), ("imu_rot_xyzw", C.c_double * 4),
                ("P", C.c_double * 729), ("timestamp", C.c_double)] + [(n, C.c_int32) for n in (
                    "b_state_initialized", "b_yaw_initialized", "b_rotation_stabilized", "b_state_stabilized",
                    "b_pcm_init_on_going", "_pad")]


class EgoStateC(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "timestamp", "x_m", "y_m", "z_m", "roll_rad", "pitch_rad", "yaw_rad", "roll_vel", "pitch_vel", "yaw_vel",
        "vx", "vy", "vz", "ax", "ay", "az", "x_cov_m", "y_cov_m", "z_cov_m", "roll_cov_rad", "pitch_cov_rad", "yaw_cov_rad")]


_LIB = None


def lib():
    """Load libelimaloc_hip.so (fails loudly when it has not been built)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ElmError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, dp, fp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int)
    L.elm_reg_config_default.argtypes = [C.POINTER(RegConfig)]
    L.elm_reg_config_default.restype = None
    L.elm_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.elm_ctx_destroy.argtypes = [vp]
    L.elm_ctx_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.elm_ctx_group_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.elm_register_shard.argtypes = [vp, vp, fp, C.c_size_t, C.c_size_t, dp, C.POINTER(RegConfig), C.POINTER(RegResult), C.POINTER(IterTrace), C.c_int]
    L.elm_ctx_destroy.restype = None
    L.elm_last_error.argtypes = [vp]
    L.elm_last_error.restype = C.c_char_p
    L.elm_strerror.argtypes = [C.c_int]
    L.elm_strerror.restype = C.c_char_p
    L.elm_ctx_synchronize.argtypes = [vp]
    L.elm_ctx_stream.argtypes = [vp]
    L.elm_ctx_stream.restype = vp
    L.elm_format_register_log.argtypes = [C.POINTER(RegConfig), C.POINTER(RegResult), C.c_size_t, C.POINTER(IterTrace), C.POINTER(C.c_double),
                                          C.c_double, C.c_char_p, C.c_size_t]
    L.elm_format_register_log.restype = C.c_size_t
    L.elm_ctx_set_profiling.argtypes = [vp, C.c_int]
    L.elm_ctx_set_work_counters.argtypes = [vp, C.c_int]
    L.elm_ctx_get_profile.argtypes = [vp, C.POINTER(Profile), C.c_int]
    L.elm_map_build.argtypes = [vp, fp, C.c_size_t, C.c_double, C.c_int, C.POINTER(vp)]
    L.elm_map_destroy.argtypes = [vp]
    L.elm_map_destroy.restype = None
    L.elm_map_cal_voxel_cov_all.argtypes = [vp]
    L.elm_map_cal_point_cov_all.argtypes = [vp, C.c_double]
    L.elm_map_build_neighbourhoods.argtypes = [vp]
    L.elm_map_get_info.argtypes = [vp, C.POINTER(MapInfo)]
    L.elm_map_empty.argtypes = [vp]
    L.elm_map_download_points.argtypes = [vp, dp, dp, dp, C.c_size_t]
    L.elm_map_download_voxels.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp, dp, C.c_size_t]
    L.elm_map_find_ground_height.argtypes = [vp, C.c_double, C.c_double, dp, ip]
    L.elm_cal_frame_point_cov.argtypes = [dp, C.c_size_t, C.c_double, C.c_double, C.c_double, dp]
    L.elm_align_clouds_local.argtypes = [vp, C.c_int, dp, dp, dp, dp, C.c_size_t, dp, C.c_double, C.POINTER(RegConfig), dp, dp, dp]
    L.elm_map_get_correspondences.argtypes = [vp, vp, C.c_int, dp, C.c_size_t, C.c_double, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.c_size_t, C.POINTER(C.c_size_t)]
    L.elm_scan_upload.argtypes = [vp, fp, C.c_size_t, C.c_size_t, C.POINTER(vp)]
    L.elm_scan_destroy.argtypes = [vp]
    L.elm_scan_destroy.restype = None
    L.elm_scan_size.argtypes = [vp]
    L.elm_scan_size.restype = C.c_size_t
    L.elm_scan_download.argtypes = [vp, fp, C.c_size_t]
    L.elm_deskew_downsample.argtypes = [vp, fp, fp, C.c_size_t, C.POINTER(DeskewTables), C.c_double, C.POINTER(vp), ip]
    L.elm_register.argtypes = [vp, vp, fp, C.c_size_t, dp, C.POINTER(RegConfig), dp, ip, dp, dp,
                               C.POINTER(RegResult), C.POINTER(IterTrace)]
    L.elm_register_stream.argtypes = [vp, vp, C.POINTER(vp), C.c_int, dp, C.POINTER(RegConfig), C.c_int, C.POINTER(RegResult),
                                      C.POINTER(IterTrace)]
    L.elm_register_stream_host.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(C.c_uint32), C.c_int, dp, C.POINTER(RegConfig), C.c_int,
                                           C.POINTER(RegResult), C.POINTER(IterTrace)]
    L.elm_ctx_measure_h2d.argtypes = [vp, vp, C.c_size_t, C.c_int, dp]
    L.elm_host_alloc.argtypes = [C.c_size_t]
    L.elm_host_alloc.restype = vp
    L.elm_host_free.argtypes = [vp]
    L.elm_host_free.restype = None
    L.elm_register_batch.argtypes = [vp, vp, C.POINTER(vp), C.c_int, dp, C.POINTER(RegConfig),
                                     C.POINTER(RegResult), C.POINTER(IterTrace)]
    L.elm_register_batch_enqueue.argtypes = [vp, vp, C.POINTER(vp), C.c_int, dp, C.POINTER(RegConfig), C.c_int]
    L.elm_register_batch_finish.argtypes = [vp, C.POINTER(RegResult), C.POINTER(IterTrace)]
    L.elm_deskew.argtypes = [vp, fp, fp, C.c_size_t, C.POINTER(DeskewTables), fp, ip]
    L.elm_deskew_prepare.argtypes = [dp, C.c_size_t, dp, C.c_size_t, C.c_double, C.c_float, C.c_float, C.c_int,
                                     C.c_int, dp, dp, dp, dp, C.c_size_t, C.POINTER(DeskewTables)]
    L.elm_filter_points_by_distance.argtypes = [fp, fp, C.c_size_t, C.c_double, fp, fp, C.POINTER(C.c_size_t)]
    L.elm_voxel_downsample.argtypes = [fp, C.c_size_t, C.c_double, C.POINTER(C.c_int64), C.POINTER(C.c_size_t)]
    L.elm_get_interpolated_pose.argtypes = [dp, C.c_size_t, C.c_double, fp, ip]
    L.elm_shape_odom_covariance.argtypes = [dp, dp, C.c_double, dp]
    L.elm_ekf_config_default.argtypes = [C.POINTER(EkfConfig)]
    L.elm_ekf_config_default.restype = None
    L.elm_ekf_create.argtypes = [C.POINTER(EkfConfig), C.POINTER(vp)]
    L.elm_ekf_destroy.argtypes = [vp]
    L.elm_ekf_destroy.restype = None
    L.elm_ekf_predict_imu.argtypes = [vp, C.c_double, dp, dp, ip]
    L.elm_ekf_predict.argtypes = [vp, C.c_double, ip]
    L.elm_ekf_update_can.argtypes = [vp, C.c_double, dp, dp, ip]
    L.elm_gps_project.argtypes = [C.c_double] * 6 + [dp]
    L.elm_ekf_update_navsatfix.argtypes = [vp] + [C.c_double] * 4 + [dp] + [C.c_double] * 3 + [C.c_int, C.c_double, dp, ip]
    L.elm_ekf_update_pose.argtypes = [vp, C.c_double, dp, dp, dp, dp, C.c_int, ip]
    L.elm_ekf_update_pcm_odom.argtypes = [vp, C.c_double, dp, dp, dp, C.c_int, ip]
    L.elm_ekf_get_state.argtypes = [vp, C.POINTER(EkfStateC)]
    L.elm_ekf_publish.argtypes = [vp, C.POINTER(EgoStateC)]
    cp, szp = C.c_char_p, C.POINTER(C.c_size_t)
    L.elm_ini_load.argtypes = [cp, C.POINTER(vp)]
    L.elm_ini_destroy.argtypes = [vp]
    L.elm_ini_destroy.restype = None
    L.elm_ini_get_string.argtypes = [vp, cp, cp, C.c_char_p, C.c_size_t]
    L.elm_ini_get_int.argtypes = [vp, cp, cp, ip]
    L.elm_ini_get_bool.argtypes = [vp, cp, cp, ip]
    L.elm_ini_get_double.argtypes = [vp, cp, cp, dp]
    L.elm_ini_get_array.argtypes = [vp, cp, cp, dp, C.c_size_t, szp]
    L.elm_pcm_node_config_default.argtypes = [C.POINTER(PcmNodeConfig)]
    L.elm_pcm_node_config_default.restype = None
    L.elm_load_pcm_config.argtypes = [cp, cp, C.POINTER(PcmNodeConfig), C.POINTER(RegConfig)]
    L.elm_load_ekf_config.argtypes = [cp, C.POINTER(EkfConfig)]
    L.elm_pcd_load_xyz.argtypes = [cp, C.POINTER(fp), szp]
    L.elm_free.argtypes = [vp]
    L.elm_free.restype = None
    L.elm_scan_from_cloud.argtypes = [vp, C.c_size_t, C.c_size_t, C.POINTER(CloudField), C.c_int, C.c_int, C.c_int, fp, fp, fp,
                                      C.c_size_t, szp]
    L.elm_pcm_callback_point_cloud.argtypes = [vp, vp, C.POINTER(PcmNodeConfig), C.POINTER(RegConfig), fp, fp, C.c_size_t, C.c_double,
                                               dp, C.c_size_t, dp, C.c_size_t, C.POINTER(PcmScanOutput), ip]
    L.elm_comm_get_unique_id.argtypes = [vp]
    L.elm_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
    L.elm_comm_destroy.argtypes = [vp]
    L.elm_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.elm_comm_set_hook.argtypes = [vp, ALLREDUCE_FN, vp]
    _LIB = L
    return L


def check(status, ctx=None, what=""):
    if status != ELM_OK:
        L = lib()
        msg = L.elm_strerror(status).decode()
        if ctx is not None:
            detail = L.elm_last_error(ctx)
            if detail:
                msg += ": " + detail.decode()
        raise ElmError(f"{what} failed ({status}): {msg}")
