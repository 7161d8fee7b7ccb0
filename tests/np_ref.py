"""Independent numpy float64 re-derivation of one RunRegister, used to cross-check the C++
oracle.  Written from the math, not from oracle/elm_oracle.cpp: dict-of-lists voxel map, numpy SVD / solve / eigh,
Rodrigues via the closed form.  Pure-Python loops -> small cases only (<= a few thousand scan points).
"""
import math

import numpy as np


DEGENERATE = set()  # keys of rank-deficient covariances met by voxel_covs / point_covs


def build_map(points, voxel_size=1.0, max_points=30):
    """AddPoints: truncation keys, first point always kept, later ones iff < max_points stored and none within
    sqrt(vs^2/max_points) (strict <)."""
    res = math.sqrt(voxel_size * voxel_size / max_points)
    vox = {}
    for p in np.asarray(points, dtype=np.float64):
        key = tuple(int(c) for c in np.trunc(p / voxel_size))
        b = vox.get(key)
        if b is None:
            vox[key] = [p]
        elif len(b) < max_points and all(np.linalg.norm(q - p) >= res for q in b):
            b.append(p)
    return vox


def regularize(cov):
    U, _, Vt = np.linalg.svd(cov)
    return U @ np.diag([1.0, 1.0, 1e-3]) @ Vt


def is_rank_deficient(cov):
    """Sample covariances of <= 3 distinct points: the reference's U diag(1,1,1e-3) V^T then depends on round-off
    decided signs inside Eigen's Jacobi SVD and cannot be pinned by an independent SVD."""
    sv = np.linalg.svd(cov, compute_uv=False)
    return sv[2] <= 1e-9 * max(sv[0], 1e-300)


def voxel_covs(vox):
    out = {}
    for k, b in vox.items():
        n = len(b)
        if n == 1:
            out[k] = (np.eye(3), b[0].copy())
        else:
            P = np.array(b)
            mean = P.mean(axis=0)
            D = P - mean
            out[k] = (regularize(D.T @ D / (n - 1)), mean)
            if is_rank_deficient(D.T @ D / (n - 1)):
                DEGENERATE.add(("v", k))
    return out


def neighbours27(g, voxel_size):
    v = np.floor(g / voxel_size).astype(int)
    for i in (-1, 0, 1):
        for j in (-1, 0, 1):
            for k in (-1, 0, 1):
                yield (v[0] + i, v[1] + j, v[2] + k)


def point_covs(vox, voxel_size, dist):
    """per stored point: {itself} + every stored point of the 27 floor-keyed voxels within dist (itself again)."""
    out = {}
    for k, b in vox.items():
        for idx, p in enumerate(b):
            nb = [p]
            for nk in neighbours27(p, voxel_size):
                for q in vox.get(nk, ()):
                    if np.sum((q - p) ** 2) <= dist * dist:
                        nb.append(q)
            if len(nb) == 1:
                out[(k, idx)] = (np.eye(3), p.copy())
            else:
                P = np.array(nb)
                mean = P.mean(axis=0)
                D = P - mean
                out[(k, idx)] = (regularize(D.T @ D / (len(nb) - 1)), mean)
                if is_rank_deficient(D.T @ D / (len(nb) - 1)):
                    DEGENERATE.add(("p", k, idx))
    return out


def skew(p):
    return np.array([[0, -p[2], p[1]], [p[2], 0, -p[0]], [-p[1], p[0], 0]])


def exp_so3(w):
    th = np.linalg.norm(w)
    if th == 0:
        return np.eye(3)
    K = skew(w / th)
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K


def rot_angle(R):
    c = (np.trace(R) - 1) / 2
    s = 0.5 * np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return math.atan2(s, c)


def radar_point_cov(g, range_var, az_var_deg, el_var_deg):
    """Registration::CalPointCov (reg.hpp:186-208) from its definition: R S with R = Rz(azimuth) Ry(elevation) written as plain
    rotation matrices and S = diag(range spread, cross-range spreads floored at 0.1) -- not symmetric, built from the MAP-frame point."""
    dist = math.hypot(g[0], g[1])
    S = np.diag([range_var, max(0.1, dist * math.sin(math.radians(az_var_deg))), max(0.1, dist * math.sin(math.radians(el_var_deg)))])
    el, az = math.atan2(g[2], dist), math.atan2(g[1], g[0])
    Rz = np.array([[math.cos(az), -math.sin(az), 0.0], [math.sin(az), math.cos(az), 0.0], [0.0, 0.0, 1.0]])
    Ry = np.array([[math.cos(el), 0.0, math.sin(el)], [0.0, 1.0, 0.0], [-math.sin(el), 0.0, math.cos(el)]])
    return Rz @ Ry @ S


def register(vox, scan, T0, method, voxel_size=1.0, max_iteration=10, th=5.0, lam=0.5, term=0.02, min_overlap=0.4,
             max_fitness=0.5, vcov=None, pcov=None, radar=None):
    """method: 0 P2P, 1 GICP, 2 VGICP, 3 AVGICP.  radar = (range_variance_m, azimuth_variance_deg, elevation_variance_deg) switches
    use_radar_cov on: the source covariance added to R^T C R is CalPointCov of the point under the INITIAL guess in the first
    iteration and the identity afterwards (reg.cpp:302-305 attaches it to source_global once, reg.cpp:390 rebuilds source_global
    from source_local, whose points carry the default identity)."""
    scan = np.asarray(scan, dtype=np.float64)
    T = np.array(T0, dtype=np.float64)
    T_init = T.copy()
    N = len(scan)
    trace = []
    fitness = 0.0
    local_cov = np.eye(6)
    if not vox:
        return dict(T=T, is_success=False, iterations=0, gate=1, iters=[], fitness=0.0, local_cov=local_cov)
    iters = 0
    for _ in range(max_iteration):
        iters += 1
        R, t = T[:3, :3], T[:3, 3]
        pairs = []  # (p_local, target_pos, cov or None, normal or None)
        for p in scan:
            g = R @ p + t
            if method in (0, 1):
                best, bd = None, np.inf
                for nk in neighbours27(g, voxel_size):
                    for idx, q in enumerate(vox.get(nk, ())):
                        d = np.sum((q - g) ** 2)
                        if d < bd:
                            bd, best = d, (nk, idx, q)
                if best is None:
                    tgt, C, mean = np.zeros(3), np.eye(3), np.zeros(3)
                    d = np.sum(g ** 2)
                else:
                    tgt = best[2]
                    d = bd
                    if method == 1:
                        C, mean = pcov[(best[0], best[1])]
                if d < th * th:
                    if method == 0:
                        pairs.append((p, tgt, None, None))
                    else:
                        w, V = np.linalg.eigh((C + C.T) / 2)
                        n = V[:, 0] if not np.allclose(C, np.eye(3)) else np.array([1.0, 0, 0])
                        pairs.append((p, mean, C, n))
            elif method == 2:
                best, bd = None, np.inf
                for nk in neighbours27(g, voxel_size):
                    if nk in vox:
                        C, mean = vcov[nk]
                        d = np.sum((mean - g) ** 2)
                        if d < bd:
                            bd, best = d, (C, mean)
                if best is None:
                    best, bd = (np.eye(3), np.zeros(3)), np.sum(g ** 2)
                if bd < th * th:
                    pairs.append((p, best[1], best[0], None))
            else:
                v = np.floor(g / voxel_size).astype(int)
                for off in ((0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
                    nk = (v[0] + off[0], v[1] + off[1], v[2] + off[2])
                    if nk in vox:
                        C, mean = vcov[nk]
                        if np.sum((mean - g) ** 2) < th * th:
                            pairs.append((p, mean, C, None))
        n_corr = len(pairs)
        rec = dict(n_corr=n_corr)
        if np.float32(n_corr) / np.float32(N) < min_overlap:
            trace.append(rec)
            return dict(T=T, is_success=False, iterations=iters, gate=2, iters=trace, fitness=fitness,
                        local_cov=local_cov)
        H = np.zeros((6, 6)); b = np.zeros(6); rs = 0.0
        Rt = R.T
        for p, m, C, n in pairs:
            r = Rt @ (m - t) - p
            J = np.hstack([np.eye(3), -skew(p)])
            w = th * th / (th + r @ r) ** 2
            if method == 1:
                w = w * 0.8 + 0.2
            if C is None:
                M = np.eye(3)
            else:
                RCR = Rt @ C @ R
                if radar is not None:
                    RCR = RCR + (radar_point_cov(T_init[:3, :3] @ p + T_init[:3, 3], *radar) if iters == 1 else np.eye(3))
                M = np.linalg.inv(RCR)
            if method >= 2 and w < 0.01:
                continue
            H += w * J.T @ M @ J
            b += w * J.T @ M @ r
            if method == 1:
                nl = Rt @ n
                nl = nl / np.linalg.norm(nl)
                rs += abs(r @ nl)
            else:
                rs += np.linalg.norm(r)
        fitness = rs / n_corr
        Hd = H + lam * np.diag(np.diag(H))
        # with the radar covariances the metric (and so H) is not symmetric; JTJ.ldlt() reads the lower triangle only
        x = eigen_ldlt_solve(Hd, b) if radar is not None else np.linalg.solve(Hd, b)
        if method == 1:
            local_cov = np.linalg.inv(Hd)
        dT = np.eye(4)
        dT[:3, :3] = exp_so3(x[3:])
        dT[:3, 3] = x[:3]
        T = T @ dT
        step = rot_angle(dT[:3, :3]) + np.linalg.norm(x[:3])
        rec.update(JTJ=H, JTr=b, residual_sum=rs, x=x, step_norm=step, T=T.copy())
        trace.append(rec)
        if step < term:
            break
    ok = not (fitness > max_fitness)
    return dict(T=T, is_success=ok, iterations=iters, gate=0 if ok else 3, iters=trace, fitness=fitness,
                local_cov=local_cov)


def eigen_ldlt_solve(A, b):
    """Independent numpy statement of Eigen 3.3.7's LDLT<MatrixXd, Lower>::compute + solve, written from the published routine
    (Eigen/src/Cholesky/LDLT.h: ldlt_inplace<Lower>::unblocked and LDLT::_solve_impl).  The factorisation is LEFT-looking: at step
    k only column k is updated (A[k,k] -= A10 (D A10^T), A21 -= A20 (D A10^T)), so the pivot search
    `mat.diagonal().tail(size-k).cwiseAbs().maxCoeff()` sees the not-yet-updated (original, permuted) diagonal entries of the
    rows below k; the FIRST of equal maxima wins; the symmetric exchange touches the lower triangle only; the column is divided by
    the pivot unless the pivot is exactly zero; the solve divides by D only where |d| > 1 / max_double (else the component is 0)."""
    M = np.array(A, dtype=np.float64)
    M = np.tril(M)  # only the lower triangle is read and written
    n = M.shape[0]
    trans = np.zeros(n, dtype=int)
    for k in range(n):
        p = k + int(np.argmax(np.abs(np.diag(M)[k:])))
        trans[k] = p
        if p != k:
            s_ = n - p - 1
            M[[k, p], :k] = M[[p, k], :k]                      # row(k).head(k) <-> row(p).head(k)
            if s_ > 0:
                M[p + 1:, [k, p]] = M[p + 1:, [p, k]]          # col(k).tail(s) <-> col(p).tail(s)
            M[k, k], M[p, p] = M[p, p], M[k, k]
            for i in range(k + 1, p):                          # the part between, transposed
                M[i, k], M[p, i] = M[p, i], M[i, k]
        rs = n - k - 1
        if k > 0:
            temp = np.diag(M)[:k] * M[k, :k]
            M[k, k] -= M[k, :k] @ temp
            if rs > 0:
                M[k + 1:, k] -= M[k + 1:, :k] @ temp
        if rs > 0 and abs(M[k, k]) > 0.0:
            M[k + 1:, k] /= M[k, k]
    y = np.asarray(b, dtype=np.float64).copy()
    for k in range(n):
        y[k], y[trans[k]] = y[trans[k]], y[k]
    for i in range(n):
        y[i] -= M[i, :i] @ y[:i]
    tol = 1.0 / np.finfo(np.float64).max
    d = np.diag(M)
    y = np.where(np.abs(d) > tol, y / np.where(d == 0.0, 1.0, d), 0.0)
    for i in range(n - 1, -1, -1):
        y[i] -= M[i + 1:, i] @ y[i + 1:]
    for k in range(n - 1, -1, -1):
        y[k], y[trans[k]] = y[trans[k]], y[k]
    return y
