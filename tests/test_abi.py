"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/elimaloc_hip.h
declares, refuses to run without a gfx950 device (no CPU fallback), and its GPU-free host functions work."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "elimaloc_hip.h")


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as g
    from elimaloc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    return _lib.lib()


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(elm_[a-z0-9_]+)\s*\(", src))
    names.discard("elm_allreduce_fn")
    return names


def test_header_symbols_all_exported(L):
    from elimaloc_amd import _lib
    decl = _declared_symbols()
    assert decl, "no declarations parsed"
    assert decl == set(_lib.EXPORTS), (decl ^ set(_lib.EXPORTS))
    for name in decl:
        assert hasattr(L, name), f"{name} declared in include/elimaloc_hip.h but not exported"


def test_struct_layouts_match_header(L):
    """ctypes mirrors vs the C structs: compile a tiny C probe that prints sizeof/offsetof."""
    import subprocess
    import tempfile
    from elimaloc_amd import _lib
    probe = r'''
#include <stdio.h>
#include <stddef.h>
#include "elimaloc_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu\n", sizeof(elm_reg_config), sizeof(elm_iter_trace), sizeof(elm_reg_result),
         sizeof(elm_map_info), sizeof(elm_deskew_tables));
  printf("%zu %zu %zu %zu\n", offsetof(elm_reg_config, gicp_cov_search_dist), offsetof(elm_reg_config, ego_to_lidar_trans),
         offsetof(elm_reg_result, is_success), offsetof(elm_deskew_tables, vec_d_imu_time));
  printf("%zu %zu %zu %zu %zu\n", sizeof(elm_ekf_config), sizeof(elm_ekf_state), sizeof(elm_ego_state),
         offsetof(elm_ekf_config, ekf_init_x_m), offsetof(elm_ekf_state, b_state_initialized));
  return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "p.c")
        open(src, "w").write(probe)
        exe = os.path.join(d, "p")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    sizes = [int(x) for x in out]
    assert sizes[:5] == [C.sizeof(_lib.RegConfig), C.sizeof(_lib.IterTrace), C.sizeof(_lib.RegResult),
                         C.sizeof(_lib.MapInfo), C.sizeof(_lib.DeskewTables)]
    assert sizes[5:9] == [_lib.RegConfig.gicp_cov_search_dist.offset, _lib.RegConfig.ego_to_lidar_trans.offset,
                          _lib.RegResult.is_success.offset, _lib.DeskewTables.vec_d_imu_time.offset]
    assert sizes[9:] == [C.sizeof(_lib.EkfConfig), C.sizeof(_lib.EkfStateC), C.sizeof(_lib.EgoStateC),
                         _lib.EkfConfig.ekf_init_x_m.offset, _lib.EkfStateC.b_state_initialized.offset]


def test_cpp_shims_compile_and_link(L):
    """The drop-in C++ shims (reference class names over the C ABI) and the ROS-free harness build with g++ and link
    against the library (running them needs a GPU)."""
    import subprocess
    import tempfile
    from elimaloc_amd import _lib
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "pcm_harness")
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "examples", "pcm_harness.cpp"), "-L", os.path.dirname(_lib.LIB_PATH),
                               "-lelimaloc_hip", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-o", exe])
        assert os.path.exists(exe)


def test_defaults_are_localization_ini(L):
    from elimaloc_amd.registration import RegistrationConfig, IcpMethod
    c = RegistrationConfig()
    # config/localization.ini:83-105
    assert (c.icp_method, c.max_iteration, c.i_max_thread, c.use_radar_cov) == (IcpMethod.GICP, 10, 10, 0)
    assert (c.max_search_dist, c.lm_lambda, c.icp_termination_threshold_m) == (5.0, 0.5, 0.02)
    assert (c.min_overlap_ratio, c.max_fitness_score, c.gicp_cov_search_dist) == (0.4, 0.5, 0.4)


def test_no_device_no_fallback(L):
    """Without a gfx950 device the context cannot be created: the product never computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = L.elm_ctx_create(0, C.byref(h))
    assert rc != 0 and not h.value
    from elimaloc_amd.registration import Context
    from elimaloc_amd._lib import ElmError
    with pytest.raises(ElmError):
        Context(0)
    # the device group's constructor: bad arguments are ELM_ERR_INVALID, a box without a GPU has no group either
    ids = (C.c_int * 2)(0, 0)
    assert L.elm_ctx_create_multi(None, 2, C.byref(h)) == -1 and L.elm_ctx_create_multi(ids, 0, C.byref(h)) == -1
    assert L.elm_ctx_create_multi(ids, 65, C.byref(h)) == -1 and L.elm_ctx_create_multi(ids, 2, None) == -1
    assert L.elm_ctx_create_multi(ids, 2, C.byref(h)) != 0 and not h.value
    with pytest.raises(ElmError):
        Context.multi([0, 0])


def test_missing_library_fails_loudly(monkeypatch):
    from elimaloc_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libelimaloc_hip.so")
    with pytest.raises(_lib.ElmError):
        _lib.lib()


def test_product_does_not_touch_the_oracle():
    """Nothing under elimaloc_amd/ or include/ may import, link or dlopen oracle/ (it is test infrastructure)."""
    bad = []
    for base in ("elimaloc_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", "Makefile")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"#include[^\n]*oracle|libelm_oracle|elm_oracle\.h|from oracle|import oracle|dlopen[^\n]*oracle", txt):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_deskew_prepare_matches_oracle(L, oracle):
    """elm_deskew_prepare is a host function (no GPU): ImuDeskewInfo / OdomDeskewInfo against the oracle."""
    from elimaloc_amd import synth, _lib
    for seed, end_mode in ((1, True), (2, False)):
        st = synth.make_deskew_stream(1000, seed=seed)
        odom = st["odom"] if end_mode else st["odom"][st["odom"][:, 0] < st["stamp"] - 0.02]  # forces twist extrapolation
        imu = np.ascontiguousarray(np.concatenate([st["imu_t"][:, None], st["imu_w"]], axis=1))
        tabs = [np.zeros(2000) for _ in range(4)]
        tab = _lib.DeskewTables()
        dp = C.POINTER(C.c_double)
        od = np.ascontiguousarray(odom)
        rc = L.elm_deskew_prepare(imu.ctypes.data_as(dp), imu.shape[0], od.ctypes.data_as(dp), od.shape[0],
                                  st["stamp"], float(st["time"][0]), float(st["time"][-1]), 1, 1,
                                  *[t.ctypes.data_as(dp) for t in tabs], 2000, C.byref(tab))
        assert rc == 0
        scan_end = st["stamp"]; scan_cur = scan_end + float(st["time"][0])
        assert (tab.d_time_scan_cur, tab.d_time_scan_end) == (scan_cur, scan_end)
        iok, itime, irot = oracle.imu_deskew_info(st["imu_t"], st["imu_w"], scan_cur, scan_end)
        ook, inc = oracle.odom_deskew_info(odom, scan_cur, scan_end)
        assert bool(tab.b_is_imu_available) == iok and bool(tab.b_is_odom_available) == ook
        k = tab.i_imu_pointer_cur + 1
        assert k == len(itime)
        assert np.array_equal(tabs[0][:k], itime)
        assert np.array_equal(np.stack([tabs[1][:k], tabs[2][:k], tabs[3][:k]], axis=1), irot)
        assert (tab.f_odom_incre_x, tab.f_odom_incre_y, tab.f_odom_incre_z) == tuple(inc)
        assert abs(tab.f_odom_incre_x) > 0.5  # 10 m/s over a 0.1 s scan
    # unavailable cases (pcm.cpp:544-547, 598-607)
    tab = _lib.DeskewTables()
    rc = L.elm_deskew_prepare(imu.ctypes.data_as(dp), 0, od.ctypes.data_as(dp), 0, st["stamp"], -0.1, 0.0, 1, 1,
                              *[t.ctypes.data_as(dp) for t in tabs], 2000, C.byref(tab))
    assert rc == 0 and not tab.b_is_imu_available and not tab.b_is_odom_available


def test_voxel_downsample_matches_oracle(oracle):
    from elimaloc_amd.registration import VoxelHashMap
    rng = np.random.default_rng(3)
    pts = rng.uniform(-20, 20, size=(5000, 3)).astype(np.float32)
    keep = oracle.voxel_downsample(pts, 1.5)
    out = VoxelHashMap.VoxelDownsample(pts, 1.5)
    assert np.array_equal(out, pts[keep])
    assert 0 < len(keep) < 5000


def _build_c_harness(d):
    import subprocess
    from elimaloc_amd import _lib
    exe = os.path.join(d, "stream_harness")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "stream_harness.c"), "-L", os.path.dirname(_lib.LIB_PATH), "-lelimaloc_hip",
                           "-lm", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-o", exe])
    return exe


def test_plain_c_caller_compiles_and_links(L):
    """The header is C (not only C++): the closed-loop example (callback + EKF + config loaders) builds with gcc -std=c11."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        assert os.path.exists(_build_c_harness(d))


@pytest.mark.gpu
def test_plain_c_closed_loop_runs(tmp_path):
    """examples/stream_harness.c on the GPU: every scan published, the EKF ends within 5 cm of the parked pose; also with
    configuration files in the reference's format driving both nodes."""
    import subprocess
    exe = _build_c_harness(str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    loc, cal = tmp_path / "localization.ini", tmp_path / "calibration.ini"
    loc.write_text("[pcm_matching]\nicp_method = 1 ; GICP, the shipped default\ninput_voxel_ds_m = 1.0\n[ekf_localization]\nuse_complementary_filter = 1\n")
    cal.write_text("[Rear To Main LiDAR]\ntransform_xyz_m = 1.0 0.0 1.6\nrotation_rpy_deg = 0.0 0.5 1.0\n[Rear To Imu]\nrotation_rpy_deg = 0 0 0\n")
    r = subprocess.run([exe, str(loc), str(cal)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_register_log_lines_are_the_references():
    """debug_print (loc.ini `debug_print`, reg.cpp:343-347, 396-413): the text elm_register writes to stdout, built by
    elm_format_register_log from a result, against the reference's own `std::cout <<` sequences (host-only: no GPU)."""
    import ctypes as C
    from elimaloc_amd import _lib
    L = _lib.lib()
    RESET, GREEN, YELLOW = "\033[0m", "\033[32m", "\033[33m"

    def fmt(cfg, res, n, trace=None, corr=None, total=0.0):
        buf = C.create_string_buffer(4096)
        need = L.elm_format_register_log(C.byref(cfg), C.byref(res), n, trace, corr, total, buf, len(buf))
        assert need == len(buf.value)
        return buf.value.decode()

    cfg = _lib.RegConfig()
    L.elm_reg_config_default(C.byref(cfg))
    res = _lib.RegResult()
    res.iterations, res.gate, res.is_success, res.n_corr_last, res.d_fitness = 2, 0, 1, 7000.0, 0.0123456789
    assert cfg.b_debug_print == 0 and fmt(cfg, res, 8000) == ""  # the default prints nothing on success
    cfg.b_debug_print = 1
    tr = (_lib.IterTrace * _lib.MAX_ITER_TRACE)()
    tr[0].n_corr, tr[1].n_corr = 6990.0, 7000.0
    corr = (C.c_double * 2)(0.125, 0.25)
    want = (f"[Registration] Total Correspondence Time for: 1 in 0.125 ms, and cores num: 6990{RESET}\n"
            f"[Registration] Total Correspondence Time for: 2 in 0.25 ms, and cores num: 7000{RESET}\n"
            f"[Registration] Total Correspondence Time: 0.375 ms{RESET}\n"
            f"[Registration] RunRegister: iteration 2 executed in 1.5 ms{RESET}\n"
            f"{GREEN}[RunRegister] Corresponding ratio 0.875{RESET}\n"
            f"{GREEN}[RunRegister] ICP Fitness Score 0.0123457{RESET}\n")
    assert fmt(cfg, res, 8000, tr, corr, 1.5) == want
    # the warnings do not depend on debug_print
    cfg.b_debug_print = 0
    res.gate, res.is_success, res.d_fitness = 3, 0, 0.75
    assert fmt(cfg, res, 8000) == f"{YELLOW}[RunRegister] ICP Fitness Score Low 0.75{RESET}\n"
    res.gate, res.n_corr_last, res.iterations = 2, 1000.0, 1
    assert fmt(cfg, res, 8000) == f"{YELLOW}[RunRegister] Small corresponding  ratio. 0.125{RESET}\n"
    res.gate, res.iterations = 1, 0
    assert fmt(cfg, res, 8000) == f"{YELLOW}VOXEL MAP EMPTY!{RESET}\n"
    # a short buffer is truncated, the needed length still reported
    small = C.create_string_buffer(8)
    assert L.elm_format_register_log(C.byref(cfg), C.byref(res), 8000, None, None, 0.0, small, 8) == len(f"{YELLOW}VOXEL MAP EMPTY!{RESET}\n")
    assert len(small.value) == 7


@pytest.mark.gpu
def test_debug_print_costs_nothing_when_off(capfd):
    """b_debug_print = 0 (the shipped default): elm_register records no event, keeps no trace and prints nothing on success;
    = 1: the reference's lines appear on stdout, one per executed iteration + the totals, and the caller's profile is untouched."""
    import numpy as np
    from elimaloc_amd import synth
    from elimaloc_amd.registration import Context, VoxelHashMap, Registration, RegistrationConfig, IcpMethod
    ctx = Context(0)
    world = synth.make_world(30000, seed=11)
    scan, Tt = synth.make_scan(world, 4096, seed=12)
    T0 = synth.perturb(Tt, seed=13)
    vm = VoxelHashMap(1.0, 30, ctx)
    vm.AddPoints(world)
    reg = Registration(RegistrationConfig(icp_method=IcpMethod.P2P), ctx)
    ctx.get_profile(reset=True)
    pose0, ok0, _, _ = reg.RunRegister(scan, vm, T0)
    out = capfd.readouterr().out
    assert ok0 and "[Registration]" not in out and "[RunRegister]" not in out
    assert ctx.get_profile()["accumulate_launches"] == 0  # no events were recorded
    pose1, ok1, _, _, det = Registration(RegistrationConfig(icp_method=IcpMethod.P2P, b_debug_print=1), ctx).RunRegister(scan, vm, T0, trace=True)
    out = capfd.readouterr().out
    assert np.array_equal(pose0, pose1) and ok1
    assert out.count("[Registration] Total Correspondence Time for: ") == det["iterations"]
    assert f"[Registration] RunRegister: iteration {det['iterations']} executed in " in out
    assert "[RunRegister] Corresponding ratio " in out and "[RunRegister] ICP Fitness Score " in out
    assert ctx.get_profile()["accumulate_launches"] == 0  # ... and the caller's profile totals are as they were
    ctx.close()


def test_runtime_switches_are_the_documented_ones():
    """the library reads a handful of environment variables and include/elimaloc_hip.h lists every one of them; a new
    getenv() in the sources without a line there fails here.  (ELM_GRID / ELM_CHECK hold tokens: every token the sources test is listed too.)"""
    import glob
    import re
    src = "".join(open(p).read() for p in sorted(glob.glob(os.path.join(ROOT, "elimaloc_amd", "csrc", "*.[ch]*"))))
    header = open(os.path.join(ROOT, "include", "elimaloc_hip.h")).read()
    shims = "".join(open(p).read() for p in sorted(glob.glob(os.path.join(ROOT, "include", "elimaloc", "*.hpp"))))
    read = set(re.findall(r'getenv\("([A-Z_0-9]+)"\)', src)) | set(re.findall(r'env_token\("([A-Z_0-9]+)"', src)) | {"ELM_CHECK"}
    assert read == {"ELM_KERNEL", "ELM_GRID", "ELM_CHECK", "ELM_SCAN_ORDER", "ELM_GROUP_EXCHANGE"}, read
    assert set(re.findall(r'getenv\("([A-Z_0-9]+)"\)', shims)) == {"ELM_DEVICES"}
    table = header[header.index("run-time switches"):header.index("/* ---------------------------------------------------------------- context")]
    for name in sorted(read | {"ELM_DEVICES"}):
        assert name in table, name
    tokens = set(re.findall(r'check_mode\("([a-z_]+)"\)', src)) | set(re.findall(r'env_token\("ELM_GRID", "([a-z_]+)"', src))
    assert tokens and all(t in table for t in tokens - {"avg_skip"}), tokens  # (avg_skip: a test-only token, drops pairs on purpose)
    # no compile-time A/B macro is left in the kernels
    kern = "".join(open(p).read() for p in sorted(glob.glob(os.path.join(ROOT, "elimaloc_amd", "csrc", "elm_k_*.hip")) + glob.glob(os.path.join(ROOT, "elimaloc_amd", "csrc", "elm_dev_*.hpp"))))
    assert not re.findall(r"^#\s*ifn?def\s+ELM_|^#\s*if\s+!?ELM_", kern, flags=re.M)
