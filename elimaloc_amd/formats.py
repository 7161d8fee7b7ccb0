"""File / wire formats either side of the registration path: thin mirrors over the C ABI.

  IniParser             bsw/system/ini_parser/ini_parser.h:7-31 (Init, ParseConfig)
  LoadPcmMatchingConfig PcmMatching::ProcessINI (pcm_matching.cpp:121-196)
  LoadEkfConfig         EkfLocalization::ProcessINI (ekf_localization.cpp:218-316)
  LoadPcdXyz            pcl::io::loadPCDFile<PointXYZINormal> at pcm_matching.cpp:72-79
  Cloudmsg2cloud / OusterCloudmsg2cloud   pcm_matching.cpp:900-930
All parsing happens in libelimaloc_hip.so (csrc/elm_io.cpp, host code).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


class IniParser:
    def __init__(self):
        self._h = None

    def Init(self, path):
        self.close()
        h = C.c_void_p()
        st = _lib.lib().elm_ini_load(str(path).encode(), C.byref(h))
        if st != _lib.ELM_OK:
            return False
        self._h = h
        return True

    def close(self):
        if self._h:
            _lib.lib().elm_ini_destroy(self._h)
        self._h = None

    __del__ = close

    def ParseConfig(self, section, key, kind=float):
        """Returns the value, or None when the key is missing (IniParser::ParseConfig returning false).
        kind: str | bool | int | float | list (vector<double>)."""
        L, s, k = _lib.lib(), section.encode(), key.encode()
        if kind is str:
            buf = C.create_string_buffer(1024)
            r = L.elm_ini_get_string(self._h, s, k, buf, 1024)
            out = buf.value.decode()
        elif kind in (bool, int):
            v = C.c_int(0)
            r = (L.elm_ini_get_bool if kind is bool else L.elm_ini_get_int)(self._h, s, k, C.byref(v))
            out = kind(v.value)
        elif kind is float:
            v = C.c_double(0)
            r = L.elm_ini_get_double(self._h, s, k, C.byref(v))
            out = v.value
        elif kind is list:
            arr = np.zeros(64)
            n = C.c_size_t(0)
            r = L.elm_ini_get_array(self._h, s, k, arr.ctypes.data_as(C.POINTER(C.c_double)), 64, C.byref(n))
            out = arr[:min(n.value, 64)].tolist()
        else:
            raise TypeError(kind)
        if r < 0:
            check(r, None, f"ini [{section}] {key}")
        return out if r == 1 else None


def LoadPcmMatchingConfig(localization_ini=None, calibration_ini=None):
    """-> PcmMatchingConfig (with .registration) filled the way ProcessINI fills cfg_ / registration_config_."""
    from .pcm_matching import PcmMatchingConfig
    from .registration import RegistrationConfig
    L = _lib.lib()
    node, reg = _lib.PcmNodeConfig(), RegistrationConfig()
    L.elm_pcm_node_config_default(C.byref(node))
    check(L.elm_load_pcm_config(None if localization_ini is None else str(localization_ini).encode(),
                                None if calibration_ini is None else str(calibration_ini).encode(), C.byref(node), C.byref(reg)),
          None, "elm_load_pcm_config")
    cfg = PcmMatchingConfig(
        b_lidar_scan_time_end=bool(node.lidar_scan_time_end), d_lidar_time_delay=node.lidar_time_delay,
        d_pcm_voxel_size=node.pcm_voxel_size, i_pcm_voxel_max_point=node.pcm_voxel_max_point, b_run_deskew=bool(node.run_deskew),
        d_input_max_dist=node.input_max_dist, d_input_voxel_ds_m=node.input_voxel_ds_m,
        tf_ego_to_lidar=np.array(node.tf_ego_to_lidar).reshape(4, 4).T.copy(), registration=reg,
        s_lidar_type=node.lidar_type.decode(), i_input_index_sampling=node.input_index_sampling)
    return cfg


def LoadEkfConfig(localization_ini):
    from .ekf import EkfConfig
    cfg = EkfConfig()
    check(_lib.lib().elm_load_ekf_config(str(localization_ini).encode(), C.byref(cfg.c)), None, "elm_load_ekf_config")
    return cfg


def LoadPcdXyz(path):
    """-> float32 [n, 3] map points in file order."""
    p = C.POINTER(C.c_float)()
    n = C.c_size_t(0)
    check(_lib.lib().elm_pcd_load_xyz(str(path).encode(), C.byref(p), C.byref(n)), None, f"elm_pcd_load_xyz({path})")
    try:
        return np.ctypeslib.as_array(p, shape=(n.value * 3,)).reshape(-1, 3).copy() if n.value else np.zeros((0, 3), np.float32)
    finally:
        _lib.lib().elm_free(p)


def _unpack(data, point_step, fields, is_ouster, index_sampling):
    raw = np.frombuffer(bytes(data), np.uint8)
    n_pts = raw.size // point_step if point_step else 0
    F = (_lib.CloudField * len(fields))()
    for i, (name, offset, datatype) in enumerate(fields):
        F[i].name, F[i].offset, F[i].datatype = name.encode(), offset, datatype
    cap = n_pts + 1
    xyz, inten, t = np.zeros((cap, 3), np.float32), np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    n = C.c_size_t(0)
    fp = C.POINTER(C.c_float)
    check(_lib.lib().elm_scan_from_cloud(raw.ctypes.data_as(C.c_void_p), n_pts, point_step, F, len(fields), int(is_ouster),
                                         int(index_sampling), xyz.ctypes.data_as(fp), inten.ctypes.data_as(fp), t.ctypes.data_as(fp),
                                         cap, C.byref(n)), None, "elm_scan_from_cloud")
    return xyz[:n.value], inten[:n.value], t[:n.value]


def Cloudmsg2cloud(data, point_step, fields):
    """PointXYZIT records (x y z intensity time) -> xyz [n,3], intensity [n], rel_time [n].  fields: (name, offset, datatype)."""
    return _unpack(data, point_step, fields, False, 1)


def OusterCloudmsg2cloud(data, point_step, fields, input_index_sampling):
    return _unpack(data, point_step, fields, True, input_index_sampling)
