"""On-disk / wire formats: csrc/elm_io.cpp against independent Python writers / parsers.

All fixtures are written by this file (no reference file is copied): the INI text uses the reference's key names with
other values, the PCD files are produced by a small writer below (ascii, binary, binary_compressed with an LZF encoder
that emits both literal runs and back-references), the scan records are packed with numpy structured dtypes.
Host-only code: runs without a GPU.
"""
import math
import struct

import numpy as np
import pytest

from elimaloc_amd import _lib, formats
from elimaloc_amd.formats import (Cloudmsg2cloud, IniParser, LoadEkfConfig, LoadPcdXyz, LoadPcmMatchingConfig,
                                  OusterCloudmsg2cloud)

LOC_INI = """; fixture written for this test-suite
# hash comments are skipped too
[common_variable]
lidar_type = ouster
lidar_scan_time_end = 0 ; 0: stamp is the first point
lidar_time_delay = 0.125 ; seconds
stray line without an equals sign

[pcm_matching]
debug_print = 0
pcm_voxel_size = 0.8 ; metres
pcm_voxel_max_point = 20 ; cap
run_deskew = 2 ; anything > 0 is true
input_max_dist = 80
input_index_sampling = 3
input_voxel_ds_m = 1.25 ; one per voxel
ICP_METHOD = 2 ; keys are case-insensitive
voxel_search_method = 2
gicp_cov_search_dist = 0.35
max_thread = 6
max_iteration = 7 ; first value
max_iteration = 12 ; a repeated key replaces the earlier value
max_search_dist = 4.5 ; metres
lm_lambda = 0.25
icp_termination_threshold_m = 1e-2 ; exponent form
min_overlap_ratio = .3
max_fitness_score = 0.75abc
use_radar_cov = -1 ; negative -> false
doppler_trans_lambda = 0.5

[EKF_LOCALIZATION]
imu_gravity = 9.79
imu_estimate_gravity = 0
use_complementary_filter = 0
gps_type = 1
ekf_init_x_m = 12.5
ekf_init_yaw_deg = -33.0
ekf_state_uncertainty_pos_m = 0.04 ; metres
ekf_imu_bias_cov_gyro = 0.0002
ekf_gnss_min_cov_z_m = 0.9
empty_value =
"""

CAL_INI = """[Rear To Imu]
transform_xyz_m = 0.0 0.0 0.1
rotation_rpy_deg = 1.0 -2.0 30.0

[Rear To Main LiDAR]
; transform_xyz_m = 9 9 9
transform_xyz_m = 1.1, -0.2, 1.7
rotation_rpy_deg = 0.5 -1.5 12.0"""  # no trailing newline on purpose


@pytest.fixture
def ini_files(tmp_path):
    loc, cal = tmp_path / "localization.ini", tmp_path / "calibration.ini"
    loc.write_text(LOC_INI)
    cal.write_bytes(CAL_INI.replace("\n", "\r\n").encode())  # CRLF line ends
    return loc, cal


def test_ini_parse_rules(ini_files):
    loc, _ = ini_files
    p = IniParser()
    assert p.Init(loc)
    assert p.ParseConfig("common_variable", "lidar_type", str) == "ouster"
    assert p.ParseConfig("common_variable", "lidar_time_delay", str) == "0.125 ; seconds"  # SimpleIni keeps inline text
    assert p.ParseConfig("common_variable", "lidar_time_delay", float) == 0.125            # atof stops at the blank
    assert p.ParseConfig("pcm_matching", "max_fitness_score", float) == 0.75
    assert p.ParseConfig("pcm_matching", "min_overlap_ratio", float) == 0.3
    assert p.ParseConfig("pcm_matching", "icp_termination_threshold_m", float) == 0.01
    assert p.ParseConfig("pcm_matching", "icp_method", int) == 2
    assert p.ParseConfig("PCM_MATCHING", "Icp_Method", int) == 2
    assert p.ParseConfig("pcm_matching", "max_iteration", int) == 12
    assert p.ParseConfig("pcm_matching", "run_deskew", bool) is True
    assert p.ParseConfig("pcm_matching", "use_radar_cov", bool) is False
    assert p.ParseConfig("pcm_matching", "pcm_voxel_size", int) == 0  # atoi("0.8 ...")
    assert p.ParseConfig("pcm_matching", "no_such_key", float) is None
    assert p.ParseConfig("no_such_section", "max_iteration", int) is None
    assert p.ParseConfig("common_variable", "stray line without an equals sign", str) is None
    assert p.ParseConfig("ekf_localization", "empty_value", str) == ""
    assert p.ParseConfig("ekf_localization", "empty_value", float) == 0.0
    assert not IniParser().Init(str(loc) + ".missing")


def test_ini_arrays(ini_files, tmp_path):
    _, cal = ini_files
    p = IniParser()
    assert p.Init(cal)
    assert p.ParseConfig("Rear To Main LiDAR", "transform_xyz_m", list) == [1.1, -0.2, 1.7]
    assert p.ParseConfig("Rear To Imu", "rotation_rpy_deg", list) == [1.0, -2.0, 30.0]
    f = tmp_path / "a.ini"
    f.write_text("[s]\nv = 1 inf -inf 2.5e0\nbad = 1 x 3\n")
    q = IniParser()
    q.Init(f)
    assert q.ParseConfig("s", "v", list) == [1.0, math.inf, -math.inf, 2.5]
    with pytest.raises(_lib.ElmError):
        q.ParseConfig("s", "bad", list)  # std::stod throws in the reference


def zyx(rpy_deg):
    r, p, y = np.radians(rpy_deg)
    Rx = np.array([[1, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]])
    Ry = np.array([[math.cos(p), 0, math.sin(p)], [0, 1, 0], [-math.sin(p), 0, math.cos(p)]])
    Rz = np.array([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def test_load_pcm_config(ini_files):
    loc, cal = ini_files
    cfg = LoadPcmMatchingConfig(loc, cal)
    assert cfg.s_lidar_type == "ouster" and cfg.i_input_index_sampling == 3
    assert cfg.b_lidar_scan_time_end is False and cfg.d_lidar_time_delay == 0.125
    assert (cfg.d_pcm_voxel_size, cfg.i_pcm_voxel_max_point, cfg.b_run_deskew) == (0.8, 20, True)
    assert (cfg.d_input_max_dist, cfg.d_input_voxel_ds_m) == (80.0, 1.25)
    r = cfg.registration
    assert (r.icp_method, r.i_max_thread, r.max_iteration, r.use_radar_cov) == (2, 6, 12, 0)
    assert (r.max_search_dist, r.lm_lambda, r.icp_termination_threshold_m) == (4.5, 0.25, 0.01)
    assert (r.min_overlap_ratio, r.max_fitness_score, r.gicp_cov_search_dist) == (0.3, 0.75, 0.35)
    assert r.range_variance_m == 1.0  # key absent from the fixture -> shipped default kept
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = zyx([0.5, -1.5, 12.0]), [1.1, -0.2, 1.7]
    np.testing.assert_allclose(cfg.tf_ego_to_lidar, T, rtol=0, atol=1e-15)
    np.testing.assert_allclose(np.array(r.ego_to_lidar_rot).reshape(3, 3).T, T[:3, :3], rtol=0, atol=1e-15)
    np.testing.assert_allclose(np.array(r.ego_to_imu_rot).reshape(3, 3).T, zyx([1.0, -2.0, 30.0]), rtol=0, atol=1e-15)
    np.testing.assert_array_equal(np.array(r.ego_to_lidar_trans), [1.1, -0.2, 1.7])


def test_defaults_when_files_absent_or_partial(tmp_path):
    cfg = LoadPcmMatchingConfig(None, None)
    assert (cfg.d_pcm_voxel_size, cfg.i_pcm_voxel_max_point, cfg.d_input_voxel_ds_m, cfg.i_input_index_sampling) == (1.0, 30, 1.5, 5)
    assert cfg.registration.icp_method == 1 and cfg.registration.max_iteration == 10
    np.testing.assert_array_equal(cfg.tf_ego_to_lidar, np.eye(4))
    bad = tmp_path / "cal.ini"
    bad.write_text("[Rear To Main LiDAR]\ntransform_xyz_m = 1 2\nrotation_rpy_deg = 0 0 0\n[Rear To Imu]\nrotation_rpy_deg = 0 0 0\n")
    with pytest.raises(_lib.ElmError):  # "Invalid Calibration!" (pcm.cpp:144-147)
        LoadPcmMatchingConfig(None, bad)
    with pytest.raises(_lib.ElmError):
        LoadPcmMatchingConfig(tmp_path / "missing.ini", None)


def test_load_ekf_config(ini_files):
    loc, _ = ini_files
    c = LoadEkfConfig(loc)
    assert (c.imu_gravity, c.imu_estimate_gravity, c.use_complementary_filter, c.gps_type) == (9.79, 0, 0, 1)
    assert (c.ekf_init_x_m, c.ekf_init_yaw_deg, c.state_std_pos_m) == (12.5, -33.0, 0.04)
    assert (c.ekf_imu_bias_cov_gyro, c.gnss_min_cov_z_m) == (0.0002, 0.9)
    assert c.state_std_vel_mps == 2.0 and c.gnss_min_cov_x_m == 0.2  # untouched defaults


# ---------------------------------------------------------------------------------------------- PCD
def lzf_compress(data: bytes) -> bytes:
    """Greedy LZF encoder (independent of the product's decoder): 3-byte hash matches, literal runs <= 32."""
    out, lit = bytearray(), bytearray()
    table, i, n = {}, 0, len(data)

    def flush():
        nonlocal lit
        while lit:
            chunk, lit = lit[:32], lit[32:]
            out.append(len(chunk) - 1)
            out.extend(chunk)

    while i < n:
        best = 0
        if i + 2 < n:
            key = data[i:i + 3]
            j = table.get(key)
            table[key] = i
            if j is not None and 0 < i - j <= 8192:
                m = 0
                while i + m < n and data[j + m] == data[i + m] and m < 264:
                    m += 1
                if m >= 3:
                    best, off = m, i - j - 1
        if best:
            flush()
            ln = best - 2
            if ln < 7:
                out.append((ln << 5) | (off >> 8))
            else:
                out.append((7 << 5) | (off >> 8))
                out.append(ln - 7)
            out.append(off & 0xFF)
            i += best
        else:
            lit.append(data[i])
            i += 1
    flush()
    return bytes(out)


def pcd_cloud(n, seed):
    rng = np.random.default_rng(seed)
    rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("_", "u1", 4), ("intensity", "<f4"), ("normal_x", "<f4"),
                             ("normal_y", "<f4"), ("normal_z", "<f4"), ("_2", "u1", 4), ("curvature", "<f4"), ("_3", "u1", 12)])
    xyz = (rng.uniform(-50, 50, (n, 3))).astype(np.float32)
    xyz[: n // 3] = np.round(xyz[: n // 3])  # repeated byte patterns so the LZF stream holds back-references
    rec["x"], rec["y"], rec["z"] = xyz.T
    rec["intensity"] = rng.uniform(0, 255, n).astype(np.float32)
    rec["normal_z"] = 1.0
    return rec, xyz


HEADER = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z _ intensity normal_x normal_y normal_z _ curvature _\n"
          "SIZE 4 4 4 1 4 4 4 4 1 4 1\nTYPE F F F U F F F F U F U\nCOUNT 1 1 1 4 1 1 1 1 4 1 12\nWIDTH {n}\nHEIGHT 1\n"
          "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA {kind}\n")


def write_pcd(path, rec, kind):
    n = rec.shape[0]
    head = HEADER.format(n=n, kind=kind).encode()
    if kind == "binary":
        body = rec.tobytes()
    elif kind == "binary_compressed":
        soa = b"".join(np.ascontiguousarray(rec[name]).tobytes() for name in rec.dtype.names)
        comp = lzf_compress(soa)
        body = struct.pack("<II", len(comp), len(soa)) + comp
    else:
        lines = []
        for r in rec:
            vals = [repr(float(r["x"])), repr(float(r["y"])), repr(float(r["z"]))] + ["0"] * 4 + [repr(float(r["intensity"])), "0", "0", "1"] + \
                   ["0"] * 4 + ["0"] + ["0"] * 12
            lines.append(" ".join(vals))
        body = ("\n".join(lines) + "\n").encode()
    path.write_bytes(head + body)


@pytest.mark.parametrize("kind", ["ascii", "binary", "binary_compressed"])
@pytest.mark.parametrize("n", [0, 1, 1000])
def test_pcd_xyz_bit_exact(tmp_path, kind, n):
    rec, xyz = pcd_cloud(n, 11 + n)
    f = tmp_path / f"m_{kind}.pcd"
    write_pcd(f, rec, kind)
    got = LoadPcdXyz(f)
    assert got.dtype == np.float32 and got.shape == (n, 3)
    assert got.tobytes() == xyz.tobytes()  # repr(float(float32)) round-trips through strtof exactly


def test_pcd_compressed_stream_has_backrefs():
    rec, _ = pcd_cloud(1000, 5)
    soa = b"".join(np.ascontiguousarray(rec[name]).tobytes() for name in rec.dtype.names)
    assert len(lzf_compress(soa)) < len(soa) // 2


def test_pcd_xyz_only_and_reordered_fields(tmp_path):
    xyz = np.arange(30, dtype=np.float32).reshape(10, 3) - 7.5
    f = tmp_path / "a.pcd"
    f.write_text("VERSION .7\nFIELDS z x y\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 5\nHEIGHT 2\nDATA ascii\n" +
                 "".join(f"{p[2]} {p[0]} {p[1]}\n" for p in xyz))
    np.testing.assert_array_equal(LoadPcdXyz(f), xyz)  # POINTS absent -> WIDTH * HEIGHT
    g = tmp_path / "nan.pcd"
    g.write_text("FIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA ascii\nnan 1 -2\n")
    got = LoadPcdXyz(g)
    assert np.isnan(got[0, 0]) and got[0, 1] == 1 and got[0, 2] == -2


def test_pcd_errors(tmp_path):
    with pytest.raises(_lib.ElmError):
        LoadPcdXyz(tmp_path / "missing.pcd")
    f = tmp_path / "d.pcd"
    f.write_text("FIELDS x y z\nSIZE 8 8 8\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA ascii\n1 2 3\n")
    with pytest.raises(_lib.ElmError):  # float64 coordinates do not match PointXYZINormal's float32 fields
        LoadPcdXyz(f)
    g = tmp_path / "t.pcd"
    g.write_bytes(b"FIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA binary\n" + b"\0" * 40)
    with pytest.raises(_lib.ElmError):  # truncated body
        LoadPcdXyz(g)
    h = tmp_path / "nofields.pcd"
    h.write_text("FIELDS x y\nSIZE 4 4\nTYPE F F\nCOUNT 1 1\nWIDTH 1\nHEIGHT 1\nDATA ascii\n1 2\n")
    with pytest.raises(_lib.ElmError):
        LoadPcdXyz(h)
    # hostile headers: POINTS * point_step wraps 64 bits (2^62 * 16), or POINTS far beyond what the file can hold -- refused
    # before any allocation or copy, for every DATA kind
    for kind in ("ascii", "binary", "binary_compressed"):
        for pts in (1 << 62, 1 << 40, 10**12):
            w = tmp_path / f"wrap_{kind}_{pts}.pcd"
            w.write_bytes(f"FIELDS x y z w\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH {pts}\nHEIGHT 1\nPOINTS {pts}\nDATA {kind}\n".encode()
                          + b"\0" * 64)
            with pytest.raises(_lib.ElmError):
                LoadPcdXyz(w)


# ---------------------------------------------------------------------------------------------- scan records
XYZIT = np.dtype({"names": ["x", "y", "z", "intensity", "time"], "formats": ["<f4"] * 5, "offsets": [0, 4, 8, 16, 20], "itemsize": 32})
OUSTER = np.dtype({"names": ["x", "y", "z", "intensity", "t", "reflectivity", "ring", "ambient", "range"],
                   "formats": ["<f4", "<f4", "<f4", "<f4", "<u4", "<u2", "<u2", "<u2", "<u4"], "offsets": [0, 4, 8, 16, 20, 24, 26, 28, 32],
                   "itemsize": 48})
F32, U16, U32 = _lib.FIELD_FLOAT32, _lib.FIELD_UINT16, _lib.FIELD_UINT32


def test_xyzit_records():
    rng = np.random.default_rng(3)
    n = 257
    rec = np.zeros(n, XYZIT)
    for k in XYZIT.names:
        rec[k] = rng.normal(size=n).astype(np.float32)
    fields = [("x", 0, F32), ("y", 4, F32), ("z", 8, F32), ("intensity", 16, F32), ("time", 20, F32)]
    xyz, inten, t = Cloudmsg2cloud(rec.tobytes(), 32, fields)
    np.testing.assert_array_equal(xyz, np.stack([rec["x"], rec["y"], rec["z"]], 1))
    np.testing.assert_array_equal(inten, rec["intensity"])
    np.testing.assert_array_equal(t, rec["time"])
    # a cloud without a time field: fromROSMsg leaves the member at 0
    xyz2, _, t2 = Cloudmsg2cloud(rec.tobytes(), 32, fields[:4])
    assert (t2 == 0).all() and (xyz2 == xyz).all()
    with pytest.raises(_lib.ElmError):
        Cloudmsg2cloud(rec.tobytes(), 32, fields[1:])
    e = Cloudmsg2cloud(b"", 32, fields)
    assert e[0].shape == (0, 3)


@pytest.mark.parametrize("n,sampling", [(10, 5), (11, 5), (9, 5), (1000, 3), (4, 1), (0, 5)])
def test_ouster_records_sampling_and_trailing_default_point(n, sampling):
    rng = np.random.default_rng(n)
    rec = np.zeros(n, OUSTER)
    rec["x"], rec["y"], rec["z"] = rng.normal(size=(3, n)).astype(np.float32)
    rec["t"] = rng.integers(0, 100_000_000, n)
    rec["reflectivity"] = rng.integers(0, 65535, n)
    rec["intensity"] = 7.0
    fields = [("x", 0, F32), ("y", 4, F32), ("z", 8, F32), ("intensity", 16, F32), ("t", 20, U32), ("reflectivity", 24, U16),
              ("ring", 26, U16), ("ambient", 28, U16), ("range", 32, U32)]
    xyz, inten, t = OusterCloudmsg2cloud(rec.tobytes(), 48, fields, sampling)
    total = n // sampling + 1                     # resize(size / sampling + 1)   (pcm.cpp:908)
    sel = rec[::sampling]
    assert xyz.shape[0] == total and sel.shape[0] in (total, total - 1)
    k = sel.shape[0]
    np.testing.assert_array_equal(xyz[:k], np.stack([sel["x"], sel["y"], sel["z"]], 1))
    np.testing.assert_array_equal(inten[:k], sel["reflectivity"].astype(np.float32))
    np.testing.assert_array_equal(t[:k], sel["t"].astype(np.float32) * np.float32(1e-9))
    if k < total:                                  # the untouched slot is a default point at the sensor origin
        assert (xyz[k:] == 0).all() and (inten[k:] == 0).all() and (t[k:] == 0).all()


@pytest.mark.gpu
def test_node_from_reference_style_files(tmp_path, oracle):
    """Init() the way the node does it (pcm.cpp:22-101): localization.ini + calibration.ini + a binary_compressed PCD map
    -> ProcessINI -> loadPCDFile -> map build -> one registration, against the oracle fed the same file contents."""
    from elimaloc_amd import synth
    from elimaloc_amd.pcm_matching import PcmMatching
    world = synth.make_world(60000, seed=1001)
    rec = np.zeros(world.shape[0], dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4")])
    rec["x"], rec["y"], rec["z"] = world.T
    soa = b"".join(np.ascontiguousarray(rec[n]).tobytes() for n in rec.dtype.names)
    comp = lzf_compress(soa)
    pcd = tmp_path / "map.pcd"
    pcd.write_bytes((f"VERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH {len(rec)}\nHEIGHT 1\n"
                     f"POINTS {len(rec)}\nDATA binary_compressed\n").encode() + struct.pack("<II", len(comp), len(soa)) + comp)
    loc, cal = tmp_path / "localization.ini", tmp_path / "calibration.ini"
    loc.write_text("[pcm_matching]\nicp_method = 2 ; VGICP\npcm_voxel_size = 1.0\npcm_voxel_max_point = 30\nmax_iteration = 12 ; more than shipped\n")
    cal.write_text("[Rear To Main LiDAR]\ntransform_xyz_m = 1.0 0.1 1.5\nrotation_rpy_deg = 0.2 -0.4 1.0\n[Rear To Imu]\nrotation_rpy_deg = 0 0 0\n")
    node = PcmMatching.FromFiles(loc, cal, pcd)
    assert node.local_map_.info().n_input_points == world.shape[0]
    assert node.cfg_.registration.icp_method == 2 and node.cfg_.registration.max_iteration == 12
    scan, T_true = synth.make_scan(world, 8000, seed=77)
    T0 = synth.perturb(T_true, seed=78)
    pose, ok, fit, cov = node.registration_.RunRegister(scan, node.local_map_, T0)
    om = oracle.Map(1.0, 30)
    om.add_points(formats.LoadPcdXyz(pcd))
    om.cal_voxel_cov_all()
    ref = oracle.register(om, scan, T0, oracle.default_config(2, max_iteration=12))
    dt, dr = synth.pose_error(ref["T"], pose)
    assert ok == ref["is_success"] and dt <= 1e-4 and dr <= 1e-5
    node.ctx.close()
