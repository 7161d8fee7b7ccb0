"""Device groups (elm_ctx_create_multi; SURVEY.md 8(b): one process, N GPUs): the registration path with the scan sharded over the ranks of
a group inside ONE process.  The one-GPU box runs the group with device_ids = {0, 0}: two real ranks (two contexts, two streams, two worker
threads) that exchange their packed sums through host memory in rank order -- the arithmetic of the RCCL path.  Bars: the group's pose is
bit-identical to the process-per-GPU form (two contexts + the exchange hook, tests/test_gpu_parity.py::_run_two_ranks) on the same shards,
within 1e-9 of the unsharded run (the summation tree), within the north_star tolerance of the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from elimaloc_amd import synth  # noqa: E402
from elimaloc_amd import dist as D  # noqa: E402

POSE_TOL_M, POSE_TOL_RAD = 1e-4, 1e-5


@pytest.fixture(scope="module")
def world100k():
    return synth.make_world(100000, seed=1001)


def _inputs(world, n=7):
    full, T0s = [], []
    for i in range(n):
        sc, Tt = synth.make_scan(world, 3000 + 700 * i, seed=900 + i)
        full.append(sc)
        T0s.append(synth.perturb(Tt, seed=950 + i, max_trans=0.05 + 0.05 * i, max_rot_deg=0.3 + 0.2 * i))
    return full, T0s


def _prepare(vm, m):
    from elimaloc_amd.registration import IcpMethod
    if m in (IcpMethod.VGICP, IcpMethod.AVGICP):
        vm.CalVoxelCovAll()
    if m == IcpMethod.GICP:
        vm.CalPointCovAll(0.4)


def _hook_ranks(world, shards_of, T0s, m, slots):
    """the process-per-GPU form on one GPU: two contexts in two threads, the exchange hook adds the ranks' sums in rank order"""
    import ctypes as C
    import threading
    from elimaloc_amd.registration import Context, Registration, RegistrationConfig, Scan, VoxelHashMap
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    barrier = threading.Barrier(2)
    bufs, results, errors = [None, None], [None, None], []

    def rank_main(r):
        try:
            ctx = Context(0)
            vmr = VoxelHashMap(1.0, 30, ctx)
            vmr.AddPoints(world)
            _prepare(vmr, m)
            scans = [Scan(ctx, sh[r], n_total=sum(len(q) for q in sh)) for sh in shards_of]

            def hook(ptr, n, hip_stream):
                assert hip.hipStreamSynchronize(C.c_void_p(hip_stream)) == 0
                mine = np.empty(n, np.float64)
                assert hip.hipMemcpy(mine.ctypes.data, ptr, n * 8, 2) == 0
                bufs[r] = mine
                barrier.wait(timeout=120)
                total = bufs[0] + bufs[1]
                barrier.wait(timeout=120)
                assert hip.hipMemcpy(ptr, total.ctypes.data, n * 8, 1) == 0
                return 0

            ctx.set_allreduce_hook(hook)
            reg = Registration(RegistrationConfig(icp_method=m), ctx)
            results[r] = reg.RunRegisterStream(scans, vmr, T0s, slots=slots) if slots else reg.RunRegisterBatch(scans, vmr, T0s)
            ctx.set_allreduce_hook(None)
            del scans, vmr
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
            barrier.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(timeout=600) for t in th]
    assert not errors, errors
    return results


@pytest.mark.parametrize("method,slots", [(0, 3), (1, 3), (2, 0), (3, 2)])
def test_group_of_two_ranks_on_one_gpu(oracle, world100k, method, slots):
    from elimaloc_amd.registration import Context, Registration, RegistrationConfig, IcpMethod, Scan, VoxelHashMap
    m = IcpMethod(method)
    full, T0s = _inputs(world100k)
    # unsharded, one plain context
    c = Context(0)
    vm = VoxelHashMap(1.0, 30, c)
    vm.AddPoints(world100k)
    _prepare(vm, m)
    single = Registration(RegistrationConfig(icp_method=m), c).RunRegisterBatch([Scan(c, s) for s in full], vm, T0s)
    del vm
    c.close()
    # the group: ONE process-level object, everything through the lead context
    g = Context.multi([0, 0])
    assert g.group_info() == (2, 2, [0, 0])  # two ranks, host-memory exchange (RCCL refuses two ranks on one device)
    gvm = VoxelHashMap(1.0, 30, g)
    gvm.AddPoints(world100k)
    _prepare(gvm, m)
    gscans = [Scan(g, s) for s in full]
    reg = Registration(RegistrationConfig(icp_method=m), g)
    grp = reg.RunRegisterStream(gscans, gvm, T0s, slots=slots) if slots else reg.RunRegisterBatch(gscans, gvm, T0s)
    again = reg.RunRegisterStream(gscans, gvm, T0s, slots=slots) if slots else reg.RunRegisterBatch(gscans, gvm, T0s)
    # the same shards through two contexts + the exchange hook
    hook = _hook_ranks(world100k, [D.spatial_shards(s, 2) for s in full], T0s, m, slots)
    for a, b, h0, h1, s in zip(grp, again, hook[0], hook[1], single):
        assert np.array_equal(a["T"], b["T"]) and a["iterations"] == b["iterations"]                 # deterministic
        assert np.array_equal(a["T"], h0["T"]) and np.array_equal(a["T"], h1["T"])                    # = the process-per-GPU form, bit for bit
        assert a["iterations"] == h0["iterations"] == s["iterations"] and a["is_success"] == s["is_success"]
        np.testing.assert_allclose(a["T"], s["T"], rtol=0, atol=1e-9)                                 # sharding changes the summation tree only
        assert a["point_iterations"] == s["point_iterations"] and a["n_corr_last"] == s["n_corr_last"]
    om = oracle.Map(1.0, 30)
    om.add_points(world100k)
    if method in (2, 3):
        om.cal_voxel_cov_all()
    if method == 1:
        om.cal_point_cov_all(0.4)
    ref = oracle.register(om, full[3], T0s[3], oracle.default_config(method))
    dt, dr = synth.pose_error(ref["T"], grp[3]["T"])
    assert ref["iterations"] == grp[3]["iterations"] and dt <= POSE_TOL_M and dr <= POSE_TOL_RAD
    del gscans, gvm
    g.close()


def test_group_run_register_on_host_buffers(oracle, world100k):
    """Registration::RunRegister (elm_register) on the lead context: the caller's points cut into contiguous shards, one per rank"""
    from elimaloc_amd.registration import Context, Registration, RegistrationConfig, IcpMethod, VoxelHashMap
    full, T0s = _inputs(world100k, 4)
    c = Context(0)
    vm = VoxelHashMap(1.0, 30, c)
    vm.AddPoints(world100k)
    plain = [Registration(RegistrationConfig(icp_method=IcpMethod.P2P), c).RunRegister(s, vm, T0, trace=True) for s, T0 in zip(full, T0s)]
    del vm
    c.close()
    g = Context.multi([0, 0, 0])  # three ranks: ragged shard bounds
    assert g.group_info()[0] == 3
    gvm = VoxelHashMap(1.0, 30, g)
    gvm.AddPoints(world100k)
    reg = Registration(RegistrationConfig(icp_method=IcpMethod.P2P), g)
    for s, T0, p in zip(full, T0s, plain):
        pose, ok, fit, cov, det = reg.RunRegister(s, gvm, T0, trace=True)
        assert ok == p[1] and det["iterations"] == p[4]["iterations"]
        np.testing.assert_allclose(pose, p[0], rtol=0, atol=1e-9)
        np.testing.assert_allclose(fit, p[2], rtol=1e-9)
        assert det["point_iterations"] == p[4]["point_iterations"]
    # read-backs and the pairs work on the lead's replica
    assert gvm.info().n_points == len(gvm.Pointcloud())
    found, z = gvm.FindGroundHeight(np.array([0.0, 0.0, 0.0]))
    assert found
    del gvm
    g.close()


def test_group_misuse_fails_loudly(world100k):
    from elimaloc_amd.registration import Context, Registration, RegistrationConfig, IcpMethod, Scan, VoxelHashMap
    from elimaloc_amd import _lib
    full, T0s = _inputs(world100k, 2)
    one = Context.multi([0])
    assert one.group_info()[0] == 1  # a group of one is a plain context
    g = Context.multi([0, 0])
    gvm = VoxelHashMap(1.0, 30, g)
    gvm.AddPoints(world100k)
    reg = Registration(RegistrationConfig(icp_method=IcpMethod.P2P), g)
    with pytest.raises(_lib.ElmError):  # a scan of another (plain) context
        reg.RunRegisterBatch([Scan(one, full[0])], gvm, T0s[:1])
    pvm = VoxelHashMap(1.0, 30, one)
    pvm.AddPoints(world100k)
    with pytest.raises(_lib.ElmError):  # a map of another context
        reg.RunRegisterBatch([Scan(g, full[0])], pvm, T0s[:1])
    with pytest.raises(_lib.ElmError):  # a shard of a shard
        Scan(g, full[0], n_total=2 * len(full[0]))
    # the group still works afterwards
    out = reg.RunRegisterBatch([Scan(g, s) for s in full], gvm, T0s)
    assert all(r["iterations"] >= 1 for r in out)
    del gvm, pvm
    g.close()
    one.close()


def test_group_with_tiny_and_empty_shards(oracle, world100k):
    """ragged inputs: scans of 0, 1, 2 and 5 points on a group of three ranks (ranks without a single point of a scan still take part in
    every exchange), through RunRegister (host buffers) and through resident scans; the same iterations, gates and poses as one context"""
    from elimaloc_amd.registration import Context, Registration, RegistrationConfig, IcpMethod, Scan, VoxelHashMap
    base, Tt = synth.make_scan(world100k, 400, seed=4321)
    T0 = synth.perturb(Tt, seed=4322, max_trans=0.05, max_rot_deg=0.2)
    sizes = [0, 1, 2, 5, 400]
    c = Context(0)
    vm = VoxelHashMap(1.0, 30, c)
    vm.AddPoints(world100k)
    reg = Registration(RegistrationConfig(icp_method=IcpMethod.P2P), c)
    plain = [reg.RunRegister(base[:n], vm, T0, trace=True) for n in sizes]
    plain_b = reg.RunRegisterBatch([Scan(c, base[:n]) for n in sizes], vm, [T0] * len(sizes))
    del vm
    c.close()
    g = Context.multi([0, 0, 0])
    gvm = VoxelHashMap(1.0, 30, g)
    gvm.AddPoints(world100k)
    greg = Registration(RegistrationConfig(icp_method=IcpMethod.P2P), g)
    for n, p in zip(sizes, plain):
        pose, ok, fit, cov, det = greg.RunRegister(base[:n], gvm, T0, trace=True)
        assert (ok, det["iterations"], det["gate"]) == (p[1], p[4]["iterations"], p[4]["gate"]), n
        np.testing.assert_allclose(pose, p[0], rtol=0, atol=1e-9)
    gb = greg.RunRegisterBatch([Scan(g, base[:n]) for n in sizes], gvm, [T0] * len(sizes))
    for n, a, b in zip(sizes, gb, plain_b):
        assert (a["is_success"], a["iterations"], a["gate"]) == (b["is_success"], b["iterations"], b["gate"]), n
        if n >= 5:  # (fewer points than unknowns: a singular system -- any step)
            np.testing.assert_allclose(a["T"], b["T"], rtol=0, atol=1e-9)
    del gvm
    g.close()


@pytest.mark.parametrize("method", [1, 2, 3])
def test_group_covariance_methods_ragged_stream(oracle, world100k, method):
    """the covariance methods through a stream on a three-rank group: more slots than registrations, scans of very different sizes (one of
    them smaller than the group), one that fails the overlap gate, use_radar_cov on one pass -- results as on one context"""
    from elimaloc_amd.registration import Context, Registration, RegistrationConfig, IcpMethod, Scan, VoxelHashMap
    m = IcpMethod(method)
    scans, T0s = [], []
    for i, n in enumerate([2, 700, 5000, 64, 1300]):
        sc, Tt = synth.make_scan(world100k, max(n, 8), seed=7700 + i)
        sc = sc[:n]
        if i == 3:
            sc = sc + np.float32(400.0)  # nowhere near the map: overlap gate in the first iteration
        scans.append(sc)
        T0s.append(synth.perturb(Tt, seed=7800 + i, max_trans=0.08, max_rot_deg=0.4))
    runs = {}
    for name, devs in (("plain", None), ("group", [0, 0, 0])):
        c = Context(0) if devs is None else Context.multi(devs)
        vm = VoxelHashMap(1.0, 30, c)
        vm.AddPoints(world100k)
        _prepare(vm, m)
        res = [Scan(c, s) for s in scans]
        out = {}
        for radar in (0, 1):
            reg = Registration(RegistrationConfig(icp_method=m, use_radar_cov=radar, range_variance_m=0.7, azimuth_variance_deg=1.5, elevation_variance_deg=0.9), c)
            out[radar] = reg.RunRegisterStream(res, vm, T0s, slots=16)
        runs[name] = out
        del res, vm
        c.close()
    for radar in (0, 1):
        for k, (a, b) in enumerate(zip(runs["group"][radar], runs["plain"][radar])):
            assert (a["is_success"], a["iterations"], a["gate"]) == (b["is_success"], b["iterations"], b["gate"]), (radar, k)
            if len(scans[k]) >= 64:
                np.testing.assert_allclose(a["T"], b["T"], rtol=0, atol=1e-8 if radar else 1e-9)
    assert runs["plain"][0][3]["gate"] == 2


def test_one_rank_group_over_rccl(oracle, world100k, monkeypatch):
    """What a one-GPU box can run of the group's RCCL path: ELM_GROUP_EXCHANGE=rccl makes elm_ctx_create_multi({0}, 1) a real group of ONE
    rank -- its worker thread loads RCCL, forms a one-rank communicator (ncclCommInitRank from that thread), enqueues one ncclAllReduce per
    iteration from that thread and destroys the communicator there.  Bit-identical to the plain context (a sum over one rank)."""
    from elimaloc_amd.registration import Context, Registration, RegistrationConfig, IcpMethod, Scan, VoxelHashMap
    full, T0s = _inputs(world100k, 5)
    c = Context(0)
    vm = VoxelHashMap(1.0, 30, c)
    vm.AddPoints(world100k)
    vm.CalVoxelCovAll()
    plain = {m: Registration(RegistrationConfig(icp_method=IcpMethod(m)), c).RunRegisterStream([Scan(c, s) for s in full], vm, T0s, slots=3) for m in (0, 2)}
    plain_one = Registration(RegistrationConfig(icp_method=IcpMethod.P2P), c).RunRegister(full[0], vm, T0s[0])
    del vm
    c.close()
    monkeypatch.setenv("ELM_GROUP_EXCHANGE", "rccl")
    g = Context.multi([0])
    assert g.group_info() == (1, 1, [0])  # one rank, RCCL exchange
    gvm = VoxelHashMap(1.0, 30, g)
    gvm.AddPoints(world100k)
    gvm.CalVoxelCovAll()
    for m in (0, 2):
        out = Registration(RegistrationConfig(icp_method=IcpMethod(m)), g).RunRegisterStream([Scan(g, s) for s in full], gvm, T0s, slots=3)
        for a, b in zip(out, plain[m]):
            assert np.array_equal(a["T"], b["T"]) and a["iterations"] == b["iterations"] and a["is_success"] == b["is_success"]
    pose, ok, fit, cov = Registration(RegistrationConfig(icp_method=IcpMethod.P2P), g).RunRegister(full[0], gvm, T0s[0])
    assert ok == plain_one[1] and np.array_equal(pose, plain_one[0])
    del gvm
    g.close()
