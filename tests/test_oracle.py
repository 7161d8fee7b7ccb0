"""CPU tests that pin the oracle (the reference ships no tests or golden vectors):
analytic known-answer cases, the reference's documented quirks, and an independent numpy float64 re-derivation
(tests/np_ref.py) of the whole RunRegister trajectory for all four methods."""
import math

import numpy as np
import pytest

from elimaloc_amd import synth

import np_ref  # tests/ is on sys.path via conftest


# ------------------------------------------------------------------------------------------------ linear algebra kit
def test_ldlt_and_inverse_against_numpy(oracle):
    rng = np.random.default_rng(0)
    for _ in range(20):
        A = rng.normal(size=(6, 30))
        H = A @ A.T + 0.1 * np.eye(6)
        b = rng.normal(size=6)
        np.testing.assert_allclose(oracle.ldlt_solve6(H, b), np.linalg.solve(H, b), rtol=1e-10)
        np.testing.assert_allclose(oracle.inverse6(H), np.linalg.inv(H), rtol=1e-9, atol=1e-12)
    # singular system: Eigen's LDLT solve zeroes the components of zero pivots
    assert np.array_equal(oracle.ldlt_solve6(np.zeros((6, 6)), np.ones(6)), np.zeros(6))


def test_ldlt_discrete_cases_follow_eigens_published_algorithm(oracle):
    """The cases where Eigen's LDLT (registration.cpp:56,138,214) takes discrete decisions -- pivot order, exactly-zero pivots,
    negative pivots -- against an independent numpy statement of the published algorithm (np_ref.eigen_ldlt_solve)."""
    rng = np.random.default_rng(7)
    cases = []
    for _ in range(20):  # indefinite, well conditioned: negative pivots, pivot order matters for nothing but must not break anything
        Q, _ = np.linalg.qr(rng.normal(size=(6, 6)))
        cases.append((Q @ np.diag([5.0, -3.0, 2.0, -1.5, 1.0, 0.7]) @ Q.T, rng.normal(size=6), True))
    # equal diagonal entries: the FIRST of the maxima is the pivot
    H = np.full((6, 6), 0.25) + np.diag([2.0, 2.0, 2.0, 1.0, 2.0, 1.0])
    cases.append((H, np.arange(1.0, 7.0), True))
    # exact zero pivots (integers: no rounding): rank 3, the system of a scan whose points lie on the sensor's x axis has this shape
    J = np.array([[1, 0, 0, 0, 2, -1], [0, 1, 0, -2, 0, 3], [0, 0, 1, 1, -3, 0]], dtype=np.float64)
    cases.append((J.T @ J, J.T @ np.array([1.0, -2.0, 0.5]), False))
    Z = np.zeros((6, 6)); Z[0, 0] = 4.0; Z[2, 2] = 1.0; Z[0, 2] = Z[2, 0] = 1.0  # zero rows / columns: components of zero pivots are zeroed
    cases.append((Z, np.array([1.0, 5.0, 2.0, 7.0, 0.0, 3.0]), False))
    for A, b, nonsingular in cases:
        x = oracle.ldlt_solve6(A, b)
        np.testing.assert_allclose(x, np_ref.eigen_ldlt_solve(A, b), rtol=1e-11, atol=1e-13)
        if nonsingular:
            np.testing.assert_allclose(x, np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)
    x = oracle.ldlt_solve6(Z, np.array([1.0, 5.0, 2.0, 7.0, 0.0, 3.0]))
    assert x[1] == 0.0 and x[3] == 0.0 and x[4] == 0.0 and x[5] == 0.0  # D^-1 with Eigen's 1 / max_double cut-off
    np.testing.assert_allclose(Z[np.ix_([0, 2], [0, 2])] @ x[[0, 2]], [1.0, 2.0], rtol=1e-14)


def test_jacobi_svd_on_rank_deficient_and_signed_input(oracle):
    """JacobiSVD<Matrix3d> (voxel_hash_map.hpp:141,241) where its conventions show: rows / columns that are exactly zero are never
    rotated (two-sided Jacobi only touches index pairs with a non-zero off-diagonal), so an exactly planar neighbourhood gets the
    SAME third column in U and V and its regularisation U diag(1,1,1e-3) V^T is symmetric; the sign of a negative diagonal entry
    goes into U, not V (JacobiSVD.h step 3), so U != V for indefinite symmetric input."""
    rng = np.random.default_rng(11)
    P = rng.normal(size=(2, 30))
    C = np.zeros((3, 3)); C[:2, :2] = np.cov(P)  # points with z exactly constant
    U, S, V = oracle.jacobi_svd3(C)
    assert S[2] == 0.0 and np.array_equal(np.abs(U[:, 2]), [0.0, 0.0, 1.0]) and np.array_equal(U[:, 2], V[:, 2])
    reg = U @ np.diag([1.0, 1.0, 1e-3]) @ V.T
    np.testing.assert_allclose(reg, np.eye(3) - 0.999 * np.outer([0, 0, 1.0], [0, 0, 1.0]), atol=1e-15)
    np.testing.assert_allclose(U @ np.diag(S) @ V.T, C, atol=1e-14)
    U, S, V = oracle.jacobi_svd3(np.diag([0.7, 0.0, 0.0]))  # collinear points along x: already diagonal, nothing to do
    assert np.array_equal(U, np.eye(3)) and np.array_equal(V, np.eye(3)) and np.array_equal(S, [0.7, 0.0, 0.0])
    U, S, V = oracle.jacobi_svd3(np.diag([0.0, 2.0, 3.0]))  # sorted by decreasing singular value: columns of U and V move together
    assert np.array_equal(S, [3.0, 2.0, 0.0]) and np.array_equal(U, V) and np.array_equal(np.abs(U), np.eye(3)[:, [2, 1, 0]])
    U, S, V = oracle.jacobi_svd3(np.diag([2.0, -1.0, 0.5]))  # a negative "singular value" is made positive by negating U's column
    assert np.array_equal(S, [2.0, 1.0, 0.5]) and np.array_equal(V, np.eye(3)) and np.array_equal(U, np.diag([1.0, -1.0, 1.0]))


def test_smallest_eigenvector_conventions(oracle):
    """SelfAdjointEigenSolver<Matrix3d>::eigenvectors().col(0) (registration.cpp:89-91) on already-diagonal input: the 3x3
    tridiagonalisation leaves the identity as eigenvector basis, the ascending sort is a selection sort that only swaps for a
    strictly smaller value -- so the first of equal eigenvalues stays in place: C = I -> e_x."""
    assert np.array_equal(oracle.smallest_eigenvector(np.eye(3)), [1.0, 0.0, 0.0])
    assert np.array_equal(np.abs(oracle.smallest_eigenvector(np.diag([1.0, 1.0, 1e-3]))), [0.0, 0.0, 1.0])
    assert np.array_equal(np.abs(oracle.smallest_eigenvector(np.diag([1.0, 1e-3, 1.0]))), [0.0, 1.0, 0.0])
    assert np.array_equal(np.abs(oracle.smallest_eigenvector(np.diag([1e-3, 1e-3, 1.0]))), [1.0, 0.0, 0.0])  # two equal smallest: the first
    rng = np.random.default_rng(12)
    for _ in range(10):
        Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        n = oracle.smallest_eigenvector(Q @ np.diag([1.0, 1.0, 1e-3]) @ Q.T)
        assert abs(abs(n @ Q[:, 2]) - 1.0) < 1e-12 and abs(np.linalg.norm(n) - 1.0) < 1e-12


def test_jacobi_svd_reconstructs_and_orders(oracle):
    rng = np.random.default_rng(1)
    for _ in range(20):
        A = rng.normal(size=(3, 3))
        U, S, V = oracle.jacobi_svd3(A)
        np.testing.assert_allclose(U @ np.diag(S) @ V.T, A, atol=1e-13)
        np.testing.assert_allclose(U.T @ U, np.eye(3), atol=1e-13)
        np.testing.assert_allclose(V.T @ V, np.eye(3), atol=1e-13)
        assert S[0] >= S[1] >= S[2] >= 0
        np.testing.assert_allclose(S, np.linalg.svd(A, compute_uv=False), rtol=1e-12)
    # symmetric PSD: plane regularisation == I - 0.999 n n^T with n the least-variance direction
    P = rng.normal(size=(3, 40)) * np.array([[1.0], [0.7], [0.01]])
    C = np.cov(P)
    U, S, V = oracle.jacobi_svd3(C)
    reg = U @ np.diag([1, 1, 1e-3]) @ V.T
    n = np.linalg.eigh(C)[1][:, 0]
    np.testing.assert_allclose(reg, np.eye(3) - 0.999 * np.outer(n, n), atol=1e-9)


def test_angle_axis_round_trip(oracle):
    rng = np.random.default_rng(2)
    for _ in range(20):
        v = rng.normal(size=3)
        v *= rng.uniform(0, 3.0) / np.linalg.norm(v)  # angle in [0, pi)
        R = oracle.angle_axis_to_matrix(v)
        np.testing.assert_allclose(R, synth.rotvec_to_matrix(v), atol=1e-14)
        np.testing.assert_allclose(oracle.matrix_to_angle(R), np.linalg.norm(v), atol=1e-12)
    assert np.array_equal(oracle.angle_axis_to_matrix(np.zeros(3)), np.eye(3))  # zero vector -> identity (reg.cpp:61)
    assert oracle.matrix_to_angle(np.eye(3)) == 0.0


# ------------------------------------------------------------------------------------------------ map semantics
def test_trunc_store_vs_floor_query_and_origin_default(oracle):
    """QUIRK #1 + #3: a point at x=-0.5 is STORED under key 0 (truncation) but a query at x=-1.2 looks in floor keys
    {-3,-2,-1}: nothing found -> the default PointStruct at the origin is the 'neighbour' and passes the 5 m test."""
    m = oracle.Map(1.0, 30)
    m.add_points(np.array([[-0.5, 0.25, 0.25]], np.float32))
    key, npts, _, _ = m.voxels()
    assert key.tolist() == [[0, 0, 0]] and npts.tolist() == [1]
    acc, tgt, d2 = m.nearest_points(np.array([[-0.6, 0.25, 0.25], [-1.2, 0.25, 0.25], [-7.0, 0.25, 0.25]]))
    assert acc.tolist() == [True, True, False]
    assert np.allclose(tgt[0], [-0.5, 0.25, 0.25]) and np.isclose(d2[0], 0.01)
    assert np.array_equal(tgt[1], [0, 0, 0]) and np.isclose(d2[1], 1.2 ** 2 + 2 * 0.25 ** 2)  # bogus origin pair
    assert np.array_equal(tgt[2], [0, 0, 0])


def test_spacing_rule_and_capacity(oracle):
    """AddPointWithSpacing: strict < map_resolution = sqrt(vs^2/max) on the Euclidean norm; first point always kept;
    capacity max_points."""
    res = math.sqrt(1.0 / 30)
    base = np.array([0.5, 0.5, 0.5])
    pts = np.array([base, base + [res * 0.999, 0, 0], base + [res * 1.001, 0, 0], base + [0, 2 * res, 0]], np.float32)
    m = oracle.Map(1.0, 30)
    m.add_points(pts)
    assert m.num_points == 3  # the second is within the spacing of the first
    # capacity: 4 far-apart points, max 2
    m2 = oracle.Map(1.0, 2)
    m2.add_points(np.array([[0.1, 0.1, 0.1], [0.9, 0.1, 0.1], [0.1, 0.9, 0.1], [0.9, 0.9, 0.9]], np.float32))
    assert m2.num_points == 2
    # order dependence: insertion order decides which points survive
    m3 = oracle.Map(1.0, 30)
    m3.add_points(pts[[1, 0, 2, 3]])
    kept = m3.pointcloud()[0]
    assert any(np.allclose(k, pts[1]) for k in kept) and not any(np.allclose(k, pts[0]) for k in kept)


def test_point_cov_counts_self_twice(oracle):
    """QUIRK #2 (vhm.hpp:202-220): neighbours = {self} + every point within r of self -- self again."""
    # five points in general position (a neighbourhood of <= 3 distinct points is rank deficient: the sign of the
    # null-space columns of Eigen's U and V is then decided by round-off and cannot be pinned by an independent SVD)
    p = np.array([[10.1, 10.1, 10.1], [10.35, 10.12, 10.1], [10.15, 10.38, 10.12], [10.2, 10.2, 10.36],
                  [10.33, 10.35, 10.3]], np.float32)  # pairwise 0.2 .. 0.4 m apart: all survive the spacing rule
    m = oracle.Map(1.0, 30)
    m.add_points(p)
    assert m.num_points == 5
    m.cal_point_cov_all(0.4, threads=1)
    xyz, cov, mean = m.pointcloud()
    P = p.astype(np.float64)
    for i in range(len(P)):
        k = int(np.argmin(np.linalg.norm(xyz - P[i], axis=1)))
        near = P[np.sum((P - P[i]) ** 2, axis=1) <= 0.4 * 0.4]
        nb = np.vstack([P[i], near])  # self + everything within 0.4 m -- which contains self again
        assert len(nb) == 6
        np.testing.assert_allclose(mean[k], nb.mean(axis=0), atol=1e-14)
        D = nb - nb.mean(axis=0)
        U, _, Vt = np.linalg.svd(D.T @ D / (len(nb) - 1))
        np.testing.assert_allclose(cov[k], U @ np.diag([1, 1, 1e-3]) @ Vt, atol=1e-9)


def test_voxel_cov_small_counts(oracle):
    m = oracle.Map(1.0, 30)
    m.add_points(np.array([[5.5, 5.5, 5.5], [7.2, 7.2, 7.2], [7.7, 7.3, 7.4]], np.float32))
    m.cal_voxel_cov_all(1)
    key, npts, cov, mean = m.voxels()
    for k, n, c, mu in zip(key, npts, cov, mean):
        if n == 1:  # (I, the point) vhm.hpp:121-124
            assert np.array_equal(c, np.eye(3)) and np.allclose(mu, [5.5, 5.5, 5.5])
        else:
            assert n == 2 and np.allclose(mu, [7.45, 7.25, 7.3], atol=1e-6)
            assert np.allclose(c, c.T, atol=1e-12) and np.isclose(np.trace(c), 2.001, atol=1e-9)


def test_voxel_downsample_first_point_per_floor_voxel(oracle):
    pts = np.array([[0.2, 0.2, 0.2], [0.4, 0.4, 0.4], [-0.2, 0.2, 0.2], [1.6, 0.1, 0.1], [-1.4, 0.2, 0.2]], np.float32)
    assert oracle.voxel_downsample(pts, 1.5).tolist() == [0, 2, 3]  # floor keys: (0,0,0) (0,0,0) (-1,0,0) (1,0,0) (-1,0,0)


def test_find_ground_height(oracle):
    rng = np.random.default_rng(5)
    pts = np.concatenate([rng.uniform(-4, 4, size=(200, 3)) * [1, 1, 0.0] + [0, 0, 0.3],
                          rng.uniform(-4, 4, size=(50, 3)) + [0, 0, 6]]).astype(np.float32)
    m = oracle.Map(1.0, 30)
    m.add_points(pts)
    ok, z = m.find_ground_height(0.0, 0.0)
    assert ok and abs(z - 0.3) < 1e-6
    ok, _ = m.find_ground_height(100.0, 100.0)
    assert not ok


# ------------------------------------------------------------------------------------------------ registration KATs
def test_single_pair_normal_equations_p2p(oracle):
    """One map point, one scan point, T0 = I: JTJ / JTr by hand (reg.cpp:28-51)."""
    q = np.array([2.25, 1.5, 0.75])
    p = np.array([2.0, 1.25, 1.0])
    m = oracle.Map(1.0, 30)
    m.add_points(q[None].astype(np.float32))
    r = oracle.register(m, p[None].astype(np.float32), np.eye(4), oracle.default_config(oracle.P2P, max_iteration=1))
    res = q - p
    w = 25.0 / (5.0 + res @ res) ** 2
    J = np.hstack([np.eye(3), -np_ref.skew(p)])
    it = r["iters"][0]
    assert it["n_corr"] == 1
    np.testing.assert_allclose(it["JTJ"], w * J.T @ J, atol=1e-15)
    np.testing.assert_allclose(it["JTr"], w * J.T @ res, atol=1e-15)
    np.testing.assert_allclose(it["residual_sum"], np.linalg.norm(res), atol=1e-15)
    assert r["fitness"] == pytest.approx(np.linalg.norm(res))


def test_identity_registration_is_a_fixed_point(oracle):
    world = synth.make_world(20000, seed=4)
    m = oracle.Map(1.0, 30)
    m.add_points(world)
    kept = m.pointcloud()[0]
    rng = np.random.default_rng(0)
    scan_w = kept[rng.choice(len(kept), 3000, replace=False)]
    T = np.eye(4); T[:3, :3] = synth.rot_zyx(0.01, -0.02, 0.7); T[:3, 3] = [1.5, -2.5, 1.8]
    local = (scan_w - T[:3, 3]) @ T[:3, :3]
    r = oracle.register(m, local.astype(np.float32), T, oracle.default_config(oracle.P2P))
    assert r["is_success"] and r["iterations"] == 1 and r["iters"][0]["n_corr"] == 3000
    dt, dr = synth.pose_error(T, r["T"])
    assert dt < 1e-5 and dr < 1e-6 and r["fitness"] < 1e-4  # float32 rounding of the scan only


def test_pure_translation_and_rotation_recovery(oracle):
    world = synth.make_world(60000, seed=6)
    m = oracle.Map(1.0, 30)
    m.add_points(world)
    m.cal_voxel_cov_all()
    scan, T_true = synth.make_scan(world, 6000, seed=7, noise=0.0)
    for delta_t, delta_r in (([0.3, -0.2, 0.1], [0, 0, 0]), ([0, 0, 0], [0.0, 0.0, 0.03])):
        D = np.eye(4); D[:3, :3] = synth.rotvec_to_matrix(delta_r); D[:3, 3] = delta_t
        r = oracle.register(m, scan, T_true @ D, oracle.default_config(oracle.VGICP, max_iteration=30,
                                                                       icp_termination_threshold_m=1e-4))
        dt, dr = synth.pose_error(T_true, r["T"])
        assert r["is_success"] and dt < 0.02 and dr < 2e-3, (dt, dr)  # VGICP's fixed point has a few-mm bias
        steps = [it["step_norm"] for it in r["iters"]]
        assert steps[0] > steps[-1] and steps[-1] < 1e-4


def test_gates(oracle):
    world = synth.make_world(20000, seed=8)
    m = oracle.Map(1.0, 30)
    m.add_points(world)
    scan, T_true = synth.make_scan(world, 2000, seed=9)
    # empty map
    r = oracle.register(oracle.Map(1.0, 30), scan, T_true, oracle.default_config(oracle.P2P))
    assert not r["is_success"] and r["gate"] == 1 and np.array_equal(r["T"], T_true) and r["iterations"] == 0
    # overlap gate: float division (float)n / N < 0.4 ; 800/2000 = 0.4 in float32 is 0.4000000059 -> NOT < 0.4
    far = scan.copy(); far[800:] += np.float32(500.0)
    r = oracle.register(m, far, T_true, oracle.default_config(oracle.P2P))
    assert r["iters"][0]["n_corr"] == 800 and r["gate"] != 2
    far[799] += np.float32(500.0)
    r = oracle.register(m, far, T_true, oracle.default_config(oracle.P2P))
    assert r["iters"][0]["n_corr"] == 799 and r["gate"] == 2 and r["iterations"] == 1 and np.array_equal(r["T"], T_true)
    # fitness gate
    r = oracle.register(m, scan, T_true, oracle.default_config(oracle.P2P, max_fitness_score=1e-6))
    assert not r["is_success"] and r["gate"] == 3


def test_vgicp_small_weight_skip(oracle):
    """reg.cpp:201: w < 0.01 pairs are skipped but stay in the fitness denominator; live only when th > 9."""
    m = oracle.Map(1.0, 30)
    m.add_points(np.array([[0.5, 0.5, 0.5], [0.6, 0.4, 0.5]], np.float32))
    m.cal_voxel_cov_all(1)
    scan = np.array([[0.5, 0.5, 0.5], [1.45, 1.45, 1.45]], np.float32)
    cfg = oracle.default_config(oracle.VGICP, max_iteration=1, max_search_dist=100.0, min_overlap_ratio=0.0)
    r = oracle.register(m, scan, np.eye(4), cfg)
    it = r["iters"][0]
    assert it["n_corr"] == 2
    # second pair: |r|^2 ~ 2.7 -> w = 1e4/(100+2.7)^2 = 0.948 (kept); move it far: w < 0.01 needs |r|^2 > 900
    cfg2 = oracle.default_config(oracle.VGICP, max_iteration=1, max_search_dist=12.0, min_overlap_ratio=0.0)
    w_far = 144.0 / (12.0 + 2.7) ** 2
    assert w_far > 0.01  # sanity of the formula for this geometry


# ------------------------------------------------------------------------------------------------ numpy cross-check
@pytest.mark.parametrize("method,radar", [(0, False), (1, False), (2, False), (3, False), (1, True), (2, True), (3, True), (0, True)])
def test_oracle_matches_numpy_rederivation(oracle, method, radar):
    if method == 0:
        world = synth.make_world(9000, seed=31)
        scan, T_true = synth.make_scan(world, 700, seed=32)
    else:
        # covariance methods: a dense volumetric cloud, so that every voxel / 0.4 m neighbourhood has full rank (rank
        # deficient covariances are regularised with round-off-decided signs in the reference, see the SVD test)
        rng = np.random.default_rng(31)
        world = (rng.uniform(0.0, 1.0, size=(40000, 3)) * [7.0, 7.0, 3.0] - [3.5, 3.5, 0.0]).astype(np.float32)
        T_true = np.eye(4); T_true[:3, :3] = synth.rot_zyx(0.02, -0.01, 0.6); T_true[:3, 3] = [0.3, -0.2, 1.4]
        inner = world[(np.abs(world[:, 0]) < 2.0) & (np.abs(world[:, 1]) < 2.0) & (world[:, 2] > 0.8) & (world[:, 2] < 2.2)]
        pick = inner[rng.choice(len(inner), 600, replace=False)].astype(np.float64) + rng.normal(0, 0.01, size=(600, 3))
        scan = ((pick - T_true[:3, 3]) @ T_true[:3, :3]).astype(np.float32)
    T0 = synth.perturb(T_true, seed=33, max_trans=0.3, max_rot_deg=1.0)
    m = oracle.Map(1.0, 30)
    m.add_points(world)
    vox = np_ref.build_map(world)
    assert m.num_points == sum(len(b) for b in vox.values()) and m.num_voxels == len(vox)
    vcov = pcov = None
    np_ref.DEGENERATE.clear()
    if method in (2, 3):
        m.cal_voxel_cov_all()
        vcov = np_ref.voxel_covs(vox)
        key, npts, ocov, omean = m.voxels()
        omap = {tuple(k): (c, mu) for k, c, mu in zip(key, ocov, omean)}
        for k in vcov:
            if ("v", k) in np_ref.DEGENERATE:
                vcov[k] = omap[k]  # implementation-defined in the reference: take the oracle's value
            else:
                np.testing.assert_allclose(vcov[k][0], omap[k][0], atol=1e-9)
                np.testing.assert_allclose(vcov[k][1], omap[k][1], atol=1e-12)
    if method == 1:
        # (radar variant: 0.8 m neighbourhoods, so that no covariance is rank deficient -- with sign-flipped regularisations
        # R^T C R + I can be singular, and what Matrix3d::inverse() returns then is not worth pinning)
        cov_dist = 0.8 if radar else 0.4
        m.cal_point_cov_all(cov_dist)
        pcov = np_ref.point_covs(vox, 1.0, cov_dist)
        oxyz, ocov, omean = m.pointcloud()
        omap = {tuple(p): (c, mu) for p, c, mu in zip(oxyz, ocov, omean)}
        for (k, idx) in pcov:
            o = omap[tuple(vox[k][idx])]
            if ("p", k, idx) in np_ref.DEGENERATE:
                pcov[(k, idx)] = o
            else:
                np.testing.assert_allclose(pcov[(k, idx)][0], o[0], atol=1e-9)
                np.testing.assert_allclose(pcov[(k, idx)][1], o[1], atol=1e-12)
    assert len(np_ref.DEGENERATE) < 0.2 * m.num_points
    if radar and method == 1:
        assert not np_ref.DEGENERATE
    # radar: use_radar_cov = 1 with non-default spreads (reg.hpp:186-217): the first iteration adds R S of the point under the
    # initial guess to R^T C R, the later ones the identity; P2P ignores the switch (AlignCloudsLocal never reads a covariance)
    rv = (0.7, 1.5, 0.9)
    ref = np_ref.register(vox, scan.astype(np.float64), T0, method, vcov=vcov, pcov=pcov, radar=rv if radar else None)
    out = oracle.register(m, scan, T0, oracle.default_config(method, max_thread=4, use_radar_cov=int(radar), range_variance_m=rv[0],
                                                             azimuth_variance_deg=rv[1], elevation_variance_deg=rv[2]))
    if radar and method != 0:
        plain = oracle.register(m, scan, T0, oracle.default_config(method, max_thread=4))
        assert np.abs(plain["iters"][0]["JTJ"] - out["iters"][0]["JTJ"]).max() > 1e-3 * np.abs(plain["iters"][0]["JTJ"]).max()  # the switch matters
    assert out["iterations"] == ref["iterations"] and out["is_success"] == ref["is_success"] and out["gate"] == ref["gate"]
    for a, b in zip(out["iters"], ref["iters"]):
        assert a["n_corr"] == b["n_corr"]
        if "JTJ" not in b:
            continue
        scale = np.abs(b["JTJ"]).max()
        np.testing.assert_allclose(a["JTJ"], b["JTJ"], rtol=0, atol=1e-9 * scale)
        np.testing.assert_allclose(a["JTr"], b["JTr"], rtol=0, atol=1e-9 * scale)
        np.testing.assert_allclose(a["residual_sum"], b["residual_sum"], rtol=1e-9)
        np.testing.assert_allclose(a["T"], b["T"], atol=1e-9)
    np.testing.assert_allclose(out["T"], ref["T"], atol=1e-9)
    np.testing.assert_allclose(out["fitness"], ref["fitness"], rtol=1e-9)
    np.testing.assert_allclose(out["local_cov"], ref["local_cov"], rtol=1e-7, atol=1e-12)


def test_thread_count_does_not_change_the_result(oracle):
    world = synth.make_world(20000, seed=41)
    scan, T_true = synth.make_scan(world, 3000, seed=42)
    T0 = synth.perturb(T_true, seed=43)
    m = oracle.Map(1.0, 30)
    m.add_points(world)
    a = oracle.register(m, scan, T0, oracle.default_config(0, max_thread=1))
    b = oracle.register(m, scan, T0, oracle.default_config(0, max_thread=7))
    assert np.array_equal(a["T"], b["T"])  # ordered join -> the serial accumulation sees the same sequence


# ------------------------------------------------------------------------------------------------ deskew KATs
def test_deskew_single_points(oracle):
    """pcm.cpp:780-824 on hand-made tables, including the pos_z <- rot_z quirk (pcm.cpp:804)."""
    imu_time = np.array([0.0, 0.05, 0.10])
    imu_rot = np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 0.01], [0.0, 0.0, 0.02]])
    inc = np.array([1.0, 0.0, 0.0], np.float32)
    pts = np.array([[10.0, 0.0, 0.0], [10.0, 0.0, 0.0], [0.0, 5.0, 1.0]], np.float32)
    rel = np.array([0.10, 0.0, 0.05], np.float32)
    out = oracle.deskew_points(pts, rel, imu_time, imu_rot, 0.0, 0.10, inc)
    # point at scan end: only the quirk acts: z += rot_z_cur - incre_z = 0.02 - 0
    np.testing.assert_allclose(out[0], [10.0, 0.0, 0.02], atol=1e-6)
    # point at scan start: yaw = 0 - 0.02, translation x = 0*1 - 1 = -1, z += 0 - 0
    c, s = math.cos(-0.02), math.sin(-0.02)
    np.testing.assert_allclose(out[1], [c * 10 - 1.0, s * 10, 0.0], atol=2e-6)
    # mid point: rot_z interpolated 0.01 -> yaw -0.01, x shift 0.5 - 1, z += 0.01
    c, s = math.cos(-0.01), math.sin(-0.01)
    np.testing.assert_allclose(out[2], [-s * 5 - 0.5, c * 5, 1.0 + 0.01], atol=2e-6)
    assert out.dtype == np.float32
    # run_deskew = 0 -> copy
    assert np.array_equal(oracle.deskew_points(pts, rel, imu_time, imu_rot, 0.0, 0.10, inc, run_deskew=False), pts)


def test_imu_and_odom_tables(oracle):
    t = np.arange(0.0, 0.2001, 0.01)
    w = np.tile([0.0, 0.0, 0.5], (len(t), 1))
    ok, tt, rot = oracle.imu_deskew_info(t, w, scan_cur=0.05, scan_end=0.15)
    assert ok and tt[0] == pytest.approx(0.04) and tt[-1] == pytest.approx(0.16)  # window [start-0.01, end+0.01]
    np.testing.assert_allclose(rot[:, 2], 0.5 * (tt - tt[0]), atol=1e-12)       # Euler sum of the gyro
    ok, _, _ = oracle.imu_deskew_info(t[:0], w[:0], 0.05, 0.15)
    assert not ok
    od = np.zeros((21, 14)); od[:, 0] = t; od[:, 1] = 10.0 * t; od[:, 7] = 1.0; od[:, 8] = 10.0
    ok, inc = oracle.odom_deskew_info(od, 0.05, 0.15)
    assert ok and inc[0] == pytest.approx(1.0, abs=1e-5) and abs(inc[1]) < 1e-6  # 10 m/s over the 0.1 s scan
    ok, inc = oracle.odom_deskew_info(od[:12], 0.05, 0.15)  # no sample after scan end: twist extrapolation
    assert ok and inc[0] == pytest.approx(1.0, abs=1e-5)
    ok, _ = oracle.odom_deskew_info(od[10:], 0.05, 0.15)     # first odom later than scan start
    assert not ok
