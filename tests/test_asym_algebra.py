"""The algebra behind the antisymmetric side sums (elm_dev_reduce.hpp: asym_side_store; elm_k_solve.hip: k_solve), checked in plain numpy on the CPU.

The reference forms, per pair, J^T M J with M = (R^-1 C R^-T)^-1 and J = [I | -[p]x] (reg.cpp:100-125, 178-200) -- for an asymmetric
"covariance" C all 36 entries matter (LDLT reads the lower triangle, reg.cpp:136-138).  The device accumulates in the world frame:
the 21 upper entries of H_w = J_w^T A J_w (A = w C^-1, J_w = [I | -[a]x], a = R p) plus, on maps with an asymmetric record, the 15
entries D of the strict lower triangle of H_w - H_w^T, in closed form from the axial vector of A - A^T."""
import numpy as np


def skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def side_record(A, a):
    """the fifteen values asym_side_store writes, in its slot order"""
    n1, n2, n3 = A[2, 1] - A[1, 2], A[0, 2] - A[2, 0], A[1, 0] - A[0, 1]
    s = a[0] * n1 + a[1] * n2 + a[2] * n3
    nu = np.array([n1, n2, n3])
    d = np.zeros(15)
    d[0], d[1], d[2] = n3, -n2, n1
    d[3:12] = (np.outer(nu, a) - s * np.eye(3)).ravel()
    d[12], d[13], d[14] = s * a[2], -s * a[1], s * a[0]
    return d


def restore(upper, d):
    """k_solve: all 36 entries from the packed upper triangle + the side record"""
    H = np.zeros((6, 6))
    for i in range(6):
        for j in range(6):
            if i <= j:
                H[i, j] = upper[i, j]
            else:
                idx = (i - 1 + j) if i < 3 else (3 + (i - 3) * 3 + j) if j < 3 else (9 + i + j - 3 - 1)
                H[i, j] = upper[j, i] + d[idx]
    return H


def test_side_record_restores_the_lower_triangle():
    rng = np.random.default_rng(0)
    for _ in range(50):
        A = rng.normal(size=(3, 3))  # any 3x3, not symmetric
        a = rng.normal(size=3) * 10
        J = np.hstack([np.eye(3), -skew(a)])
        H = J.T @ A @ J
        got = restore(np.triu(H), side_record(A, a))
        np.testing.assert_allclose(got, H, rtol=0, atol=1e-12 * np.abs(H).max())
    S = rng.normal(size=(3, 3))
    assert not side_record(S + S.T, a).any()  # a symmetric A writes exact zeros


def test_world_frame_sums_with_side_record_equal_the_reference_form():
    """P^T H_w P == J_l^T (R^-1 C R^-T)^-1 J_l for an asymmetric C, all 36 entries; the sum over pairs is linear in both forms."""
    rng = np.random.default_rng(1)
    Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    R = Q * np.sign(np.linalg.det(Q))
    P = np.zeros((6, 6)); P[:3, :3] = R; P[3:, 3:] = R
    Hl = np.zeros((6, 6)); up = np.zeros((6, 6)); d = np.zeros(15)
    for _ in range(20):
        U, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        V, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        C = U @ np.diag([1.0, 1.0, 1e-3]) @ V.T  # vhm.hpp:141-146 with U != V
        p = rng.normal(size=3) * 5
        w = rng.uniform(0.1, 1.0)
        Rinv = np.linalg.inv(R)
        M = np.linalg.inv(Rinv @ C @ Rinv.T)
        Jl = np.hstack([np.eye(3), -skew(p)])
        Hl += w * Jl.T @ M @ Jl
        A = w * np.linalg.inv(C)
        a = R @ p
        Jw = np.hstack([np.eye(3), -skew(a)])
        up += np.triu(Jw.T @ A @ Jw)
        d += side_record(A, a)
    got = P.T @ restore(up, d) @ P
    assert np.abs(Hl - Hl.T).max() > 1e-3 * np.abs(Hl).max()
    np.testing.assert_allclose(got, Hl, rtol=0, atol=1e-10 * np.abs(Hl).max())
