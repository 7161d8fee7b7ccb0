"""Device-side scan ordering (k_scan_order) and the host-fed stream (elm_register_stream_host): GPU tests through the C ABI.

The ordering is a pure permutation of the caller's points -- deterministic, stable inside a cell -- so registrations on ordered scans
agree with the caller's order up to the summation order; the host-fed stream must be bit-identical to the resident stream.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from elimaloc_amd import synth  # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    from elimaloc_amd.registration import Context
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def world100k():
    return synth.make_world(100000, seed=1001)


def _hilbert(order, x, y):
    """index of cell (x, y) along a Hilbert curve over a 2^order square (the textbook xy2d)"""
    d = 0
    s = 1 << (order - 1)
    while s > 0:
        rx = 1 if (x & s) else 0
        ry = 1 if (y & s) else 0
        d += s * s * ((3 * rx) ^ ry)
        if ry == 0:
            if rx == 1:
                x, y = s - 1 - x, s - 1 - y
            x, y = y, x
        s >>= 1
    return d


def _keys(xyz):
    cx = np.clip(np.floor(xyz[:, 0] * np.float32(0.5)).astype(np.int64) + 32, 0, 63)
    cy = np.clip(np.floor(xyz[:, 1] * np.float32(0.5)).astype(np.int64) + 32, 0, 63)
    lut = np.array([[_hilbert(6, x, y) for x in range(64)] for y in range(64)])
    return lut[cy, cx]


def _download(scan):
    from elimaloc_amd import _lib
    out = np.empty((scan.n, 3), np.float32)
    _lib.check(_lib.lib().elm_scan_download(scan._h, out.ctypes.data_as(C.POINTER(C.c_float)), scan.n), scan.ctx._h, "elm_scan_download")
    return out


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000, 4097, 16383, 16384, 16385, 40000, 131072, 200001, 262144, 523776, 523777])
def test_device_order_is_the_stable_sort_by_cell(ctx, n, monkeypatch):
    """elm_scan_upload: the resident scan is exactly the caller's points stably sorted by the Hilbert index of their 2 m cell
    (points beyond +-64 m are clamped into the border cells) -- the same bytes at every upload, from the one-workgroup kernel
    (k_scan_order: scans below 16 384 points; host-fed streams use it for every size and must reproduce the resident stream bit for bit:
    test_stream_host_equals_resident_stream) and from the many-workgroup form a larger scan takes when it is uploaded on its own.
    Beyond 523 776 points (the one-workgroup kernel's 16-bit run offsets) both keep the caller's order."""
    from elimaloc_amd.registration import Scan
    rng = np.random.default_rng(n + 7)
    xyz = (rng.standard_normal((n, 3)) * np.array([40.0, 40.0, 3.0])).astype(np.float32)  # |x|, |y| beyond 64 m occur
    a = _download(Scan(ctx, xyz))
    b = _download(Scan(ctx, xyz))
    assert np.array_equal(a, b)
    expect = xyz[np.argsort(_keys(xyz), kind="stable")] if 0 < n <= 523776 else xyz
    assert np.array_equal(a, expect)


def test_device_order_degenerate_scan_keeps_the_callers_order(ctx):
    """More than 65535 points in one cell do not fit the kernel's 16-bit run offsets: the scan keeps the caller's order."""
    from elimaloc_amd.registration import Scan
    rng = np.random.default_rng(3)
    xyz = (rng.random((90000, 3)) * 1.5).astype(np.float32)  # one 2 m cell
    xyz[::7, 0] += 10.0
    assert np.array_equal(_download(Scan(ctx, xyz)), xyz)


@pytest.mark.parametrize("method", [0, 1, 2, 3])
def test_stream_host_equals_resident_stream(ctx, oracle, world100k, method):
    """elm_register_stream_host (uploads in groups on a copy stream, ordering kernel, arrival published to the solve kernel's
    refill) against elm_register_stream on elm_scan_upload'ed scans: ragged sizes incl. an empty scan, more registrations than
    slots and several upload groups, page-locked and pageable sources -- bit-identical results and traces; the oracle agrees."""
    from elimaloc_amd.registration import Registration, RegistrationConfig, IcpMethod, Scan, VoxelHashMap, PinnedBuffer
    m = IcpMethod(method)
    vm = VoxelHashMap(1.0, 30, ctx)
    vm.AddPoints(world100k)
    if m in (IcpMethod.VGICP, IcpMethod.AVGICP):
        vm.CalVoxelCovAll()
    if m == IcpMethod.GICP:
        vm.CalPointCovAll(0.4)
    reg = Registration(RegistrationConfig(icp_method=m), ctx)
    sizes = [6000, 1000, 0, 257, 5000, 3000, 256, 4097, 1, 2500, 7000] * 13  # 143 registrations: three upload groups of 64
    hosts, T0s = [], []
    for i, n in enumerate(sizes):
        sc, Tt = synth.make_scan(world100k, max(n, 1), seed=500 + i)
        hosts.append(sc[:n])
        T0s.append(synth.perturb(Tt, seed=600 + i, max_trans=0.05 + 0.004 * i, max_rot_deg=0.02 * (i + 1)))
    scans = [Scan(ctx, h) for h in hosts]
    want = reg.RunRegisterStream(scans, vm, T0s, slots=9, trace=True)
    assert len({r["iterations"] for r in want}) > 2
    pin = PinnedBuffer(max(1, sum(h.size for h in hosts)))
    for packed, slots in ((reg.pack_host_inputs(hosts, T0s, pinned=pin), 9), (reg.pack_host_inputs(hosts, T0s), 5),
                          (reg.pack_host_inputs(hosts, T0s, pinned=pin), 200)):
        got = reg.RunRegisterStreamHost(packed, vm, slots=slots, trace=True)
        for k, (a, b) in enumerate(zip(got, want)):
            assert (a["iterations"], a["is_success"], a["gate"]) == (b["iterations"], b["is_success"], b["gate"]), k
            assert np.array_equal(a["T"], b["T"]) and np.array_equal(a["local_cov"], b["local_cov"]), k
            assert a["n_corr_last"] == b["n_corr_last"]
            for ia, ib in zip(a["iters"], b["iters"]):
                assert np.array_equal(ia["JTJ"], ib["JTJ"]) and np.array_equal(ia["T"], ib["T"])
    pin.close()
    om = oracle.Map(1.0, 30)
    om.add_points(world100k)
    if m in (IcpMethod.VGICP, IcpMethod.AVGICP):
        om.cal_voxel_cov_all()
    if m == IcpMethod.GICP:
        om.cal_point_cov_all(0.4)
    for k in (0, 4, 10):
        ref = oracle.register(om, hosts[k], T0s[k], oracle.default_config(method))
        dt, dr = synth.pose_error(ref["T"], want[k]["T"])
        assert ref["iterations"] == want[k]["iterations"] and dt <= 1e-4 and dr <= 1e-5


def test_stream_host_uniform_contiguous_group_and_misuse(ctx, world100k):
    """Uniform scans back to back in one page-locked buffer take the one-DMA-per-group path; a communicator hook is refused."""
    from elimaloc_amd import _lib
    from elimaloc_amd.registration import Registration, RegistrationConfig, IcpMethod, Scan, VoxelHashMap, PinnedBuffer
    vm = VoxelHashMap(1.0, 30, ctx)
    vm.AddPoints(world100k)
    reg = Registration(RegistrationConfig(icp_method=IcpMethod.P2P), ctx)
    hosts, T0s = [], []
    for i in range(70):
        sc, Tt = synth.make_scan(world100k, 2048, seed=900 + i)
        hosts.append(sc); T0s.append(synth.perturb(Tt, seed=950 + i))
    pin = PinnedBuffer(sum(h.size for h in hosts))
    packed = reg.pack_host_inputs(hosts, T0s, pinned=pin)
    got = reg.RunRegisterStreamHost(packed, vm, slots=16)
    want = reg.RunRegisterStream([Scan(ctx, h) for h in hosts], vm, T0s, slots=16)
    for a, b in zip(got, want):
        assert np.array_equal(a["T"], b["T"]) and a["iterations"] == b["iterations"]
    ctx.set_allreduce_hook(lambda p, n, s: 0)
    try:
        with pytest.raises(_lib.ElmError):
            reg.RunRegisterStreamHost(packed, vm, slots=16)
    finally:
        ctx.set_allreduce_hook(None)
    again = reg.RunRegisterStreamHost(packed, vm, slots=16)  # the context is still usable
    assert all(np.array_equal(a["T"], b["T"]) for a, b in zip(again, want))
    pin.close()
