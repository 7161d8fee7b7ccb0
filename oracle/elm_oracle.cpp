/*
 * elm_oracle.cpp -- CPU ORACLE: restatement of the reference's pcm_matching hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see elm_oracle.h).  PARITY UNPINNED (no reference
 * goldens exist; the reference cannot be built here).
 *
 * Every function cites the reference file:line it follows.  Paths are relative to
 * /root/reference/src/app/localization/:
 *   vhm.hpp = pcm_matching/include/voxel_hash_map.hpp   vhm.cpp = pcm_matching/src/voxel_hash_map.cpp
 *   reg.hpp = pcm_matching/include/registration.hpp     reg.cpp = pcm_matching/src/registration.cpp
 *   pcm.cpp = pcm_matching/src/pcm_matching.cpp         lf.hpp  = localization_interface/localization_functions.hpp
 *
 * Data structures are kept deliberately faithful (168-byte AoS PointStruct, node-based
 * std::unordered_map<Voxel, VoxelBlock>, a heap-allocated neighbour-voxel vector per query,
 * whole-struct copies on every NN improvement, a serial residual/Jacobian loop, a full
 * re-transform of the scan per iteration) because this file is also the timed CPU baseline.
 *
 * Third-party arithmetic that is NOT under /root/reference (Eigen 3.3.x as shipped by
 * Ubuntu 20.04's libeigen3-dev, PCL 1.10, tf) is restated from its published algorithms:
 * cofactor inverses (3x3, 4x4), pivoted LDLT, partial-pivot LU inverse (6x6), two-sided
 * Jacobi SVD (3x3), Rodrigues (AngleAxis -> matrix), matrix -> quaternion -> angle,
 * pcl::getTransformation (ZYX Euler, float), tf getRPY / setRPY.
 *
 * Build: g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared -pthread (see Makefile).
 * No -march / -ffast-math: the reference builds with plain -std=c++14 on x86-64 (SSE2, no FMA).
 */
#include "elm_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------
// Tiny fixed-size linear algebra kit.  Matrices are stored column-major (m[c*R + r]) like Eigen.
// ---------------------------------------------------------------------------------------------
struct V3 {
    double x, y, z;
};
inline V3 sub(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline double sqn(const V3& a) { return (a.x * a.x + a.y * a.y) + a.z * a.z; } // Eigen redux order for size 3
inline double nrm(const V3& a) { return std::sqrt(sqn(a)); }

struct M3 {
    double m[9]; // column-major
    double& operator()(int r, int c) { return m[c * 3 + r]; }
    double operator()(int r, int c) const { return m[c * 3 + r]; }
};
struct M4 {
    double m[16];
    double& operator()(int r, int c) { return m[c * 4 + r]; }
    double operator()(int r, int c) const { return m[c * 4 + r]; }
};
struct M6 {
    double m[36];
    double& operator()(int r, int c) { return m[c * 6 + r]; }
    double operator()(int r, int c) const { return m[c * 6 + r]; }
};

M3 m3_identity() {
    M3 r{};
    r(0, 0) = r(1, 1) = r(2, 2) = 1.0;
    return r;
}
M4 m4_identity() {
    M4 r{};
    for (int i = 0; i < 4; ++i) r(i, i) = 1.0;
    return r;
}
M3 m3_mul(const M3& a, const M3& b) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r(i, j) = (a(i, 0) * b(0, j) + a(i, 1) * b(1, j)) + a(i, 2) * b(2, j);
    return r;
}
M3 m3_transpose(const M3& a) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r(i, j) = a(j, i);
    return r;
}
V3 m3_mulv(const M3& a, const V3& v) {
    return {(a(0, 0) * v.x + a(0, 1) * v.y) + a(0, 2) * v.z, (a(1, 0) * v.x + a(1, 1) * v.y) + a(1, 2) * v.z,
            (a(2, 0) * v.x + a(2, 1) * v.y) + a(2, 2) * v.z};
}
M4 m4_mul(const M4& a, const M4& b) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r(i, j) = ((a(i, 0) * b(0, j) + a(i, 1) * b(1, j)) + a(i, 2) * b(2, j)) + a(i, 3) * b(3, j);
    return r;
}

// Eigen Matrix3d::inverse(): cofactors times 1/det (Eigen/src/LU/InverseImpl.h, size-3 helper).
M3 m3_inverse(const M3& a) {
    auto cof = [&](int i, int j) {
        int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return a(i1, j1) * a(i2, j2) - a(i1, j2) * a(i2, j1);
    };
    double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
    double det = (c00 * a(0, 0) + c10 * a(1, 0)) + c20 * a(2, 0);
    double invdet = 1.0 / det;
    M3 r;
    // inverse(i,j) = cofactor(j,i) * invdet
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r(i, j) = cof(j, i) * invdet;
    r(0, 0) = c00 * invdet;
    r(0, 1) = c10 * invdet;
    r(0, 2) = c20 * invdet;
    return r;
}

// Eigen Matrix4d::inverse(): adjugate / determinant (size-4 cofactor form).
M4 m4_inverse(const M4& a) {
    auto det3 = [&](int r0, int r1, int r2, int c0, int c1, int c2) {
        return a(r0, c0) * (a(r1, c1) * a(r2, c2) - a(r1, c2) * a(r2, c1)) -
               a(r0, c1) * (a(r1, c0) * a(r2, c2) - a(r1, c2) * a(r2, c0)) +
               a(r0, c2) * (a(r1, c0) * a(r2, c1) - a(r1, c1) * a(r2, c0));
    };
    M4 cofm;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) {
            int r[3], c[3], ri = 0, ci = 0;
            for (int k = 0; k < 4; ++k) {
                if (k != i) r[ri++] = k;
                if (k != j) c[ci++] = k;
            }
            double d = det3(r[0], r[1], r[2], c[0], c[1], c[2]);
            cofm(i, j) = ((i + j) & 1) ? -d : d;
        }
    }
    double det = ((a(0, 0) * cofm(0, 0) + a(0, 1) * cofm(0, 1)) + a(0, 2) * cofm(0, 2)) + a(0, 3) * cofm(0, 3);
    double invdet = 1.0 / det;
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r(i, j) = cofm(j, i) * invdet;
    return r;
}

// Eigen LDLT<Matrix6d>::compute + solve (Eigen/src/Cholesky/LDLT.h, ldlt_inplace<Lower>::unblocked,
// LDLT::_solve_impl).  Pivot = largest |diagonal| of the remaining block, symmetric swap.
void ldlt_solve6(const M6& A, const double b[6], double x[6]) {
    const int n = 6;
    M6 mat = A;
    int transpositions[6];
    for (int k = 0; k < n; ++k) {
        // biggest diagonal entry in the remaining block
        int piv = k;
        double best = std::fabs(mat(k, k));
        for (int i = k + 1; i < n; ++i) {
            double v = std::fabs(mat(i, i));
            if (v > best) {
                best = v;
                piv = i;
            }
        }
        transpositions[k] = piv;
        if (k != piv) {
            // symmetric swap working on the lower triangle only
            int s = n - piv - 1;
            for (int c = 0; c < k; ++c) std::swap(mat(k, c), mat(piv, c));           // row(k).head(k) <-> row(piv).head(k)
            for (int i = 0; i < s; ++i) std::swap(mat(piv + 1 + i, k), mat(piv + 1 + i, piv)); // col(k).tail(s) <-> col(piv).tail(s)
            std::swap(mat(k, k), mat(piv, piv));
            for (int i = k + 1; i < piv; ++i) std::swap(mat(i, k), mat(piv, i));     // the "middle" part, transposed
        }
        int rs = n - k - 1;
        // A10 = row(k).head(k), A20 = block below, A21 = col(k).tail(rs)
        if (k > 0) {
            double temp[6];
            for (int c = 0; c < k; ++c) temp[c] = mat(c, c) * mat(k, c); // temp = real(diag.head(k)) .* A10^*
            double acc = 0.0;
            for (int c = 0; c < k; ++c) acc += mat(k, c) * temp[c];
            mat(k, k) -= acc;
            if (rs > 0) {
                for (int i = 0; i < rs; ++i) {
                    double a2 = 0.0;
                    for (int c = 0; c < k; ++c) a2 += mat(k + 1 + i, c) * temp[c];
                    mat(k + 1 + i, k) -= a2;
                }
            }
        }
        double realAkk = mat(k, k);
        bool pivot_is_valid = (std::fabs(realAkk) > 0.0);
        if (rs > 0 && pivot_is_valid)
            for (int i = 0; i < rs; ++i) mat(k + 1 + i, k) /= realAkk;
    }
    // solve: dst = P b
    double d[6];
    for (int i = 0; i < n; ++i) d[i] = b[i];
    for (int k = 0; k < n; ++k) std::swap(d[k], d[transpositions[k]]);
    // dst = L^-1 (P b)   (unit lower)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) d[i] -= mat(i, j) * d[j];
    // dst = D^-1 ...  with Eigen's pseudo-inverse tolerance
    const double tolerance = 1.0 / std::numeric_limits<double>::max();
    for (int i = 0; i < n; ++i) {
        if (std::fabs(mat(i, i)) > tolerance)
            d[i] /= mat(i, i);
        else
            d[i] = 0.0;
    }
    // dst = L^-T ...
    for (int i = n - 1; i >= 0; --i)
        for (int j = i + 1; j < n; ++j) d[i] -= mat(j, i) * d[j];
    // dst = P^-1 ...
    for (int k = n - 1; k >= 0; --k) std::swap(d[k], d[transpositions[k]]);
    for (int i = 0; i < n; ++i) x[i] = d[i];
}

// Eigen Matrix6d::inverse(): PartialPivLU, then solve against the identity.
M6 m6_inverse(const M6& A) {
    const int n = 6;
    M6 lu = A;
    int perm[6];
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double best = std::fabs(lu(k, k));
        for (int i = k + 1; i < n; ++i) {
            double v = std::fabs(lu(i, k));
            if (v > best) {
                best = v;
                piv = i;
            }
        }
        if (piv != k) {
            for (int c = 0; c < n; ++c) std::swap(lu(k, c), lu(piv, c));
            std::swap(perm[k], perm[piv]);
        }
        if (lu(k, k) != 0.0) {
            for (int i = k + 1; i < n; ++i) lu(i, k) /= lu(k, k);
        }
        for (int c = k + 1; c < n; ++c)
            for (int i = k + 1; i < n; ++i) lu(i, c) -= lu(i, k) * lu(k, c);
    }
    M6 inv;
    for (int c = 0; c < n; ++c) {
        double y[6];
        for (int i = 0; i < n; ++i) y[i] = (perm[i] == c) ? 1.0 : 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j) y[i] -= lu(i, j) * y[j];
        for (int i = n - 1; i >= 0; --i) {
            for (int j = i + 1; j < n; ++j) y[i] -= lu(i, j) * y[j];
            y[i] /= lu(i, i);
        }
        for (int i = 0; i < n; ++i) inv(i, c) = y[i];
    }
    return inv;
}

// ---- Eigen JacobiSVD<Matrix3d>(ComputeFullU|ComputeFullV), two-sided Jacobi (Eigen/src/SVD/JacobiSVD.h,
//      Eigen/src/Jacobi/Jacobi.h) --------------------------------------------------------------------
struct JRot {
    double c, s;
};
// JacobiRotation::makeJacobi(x, y, z) for real symmetric 2x2 [[x y][y z]]
JRot make_jacobi(double x, double y, double z) {
    JRot j;
    double deno = 2.0 * std::fabs(y);
    if (deno < std::numeric_limits<double>::min()) {
        j.c = 1.0;
        j.s = 0.0;
        return j;
    }
    double tau = (x - z) / deno;
    double w = std::sqrt(tau * tau + 1.0);
    double t = (tau > 0.0) ? 1.0 / (tau + w) : 1.0 / (tau - w);
    double sign_t = t > 0.0 ? 1.0 : -1.0;
    double n = 1.0 / std::sqrt(t * t + 1.0);
    j.s = -sign_t * (y / std::fabs(y)) * std::fabs(t) * n;
    j.c = n;
    return j;
}
// apply_rotation_in_the_plane on rows p,q (applyOnTheLeft) : x' = c x + s y ; y' = -s x + c y
void rot_left(M3& m, int p, int q, JRot j) {
    for (int c = 0; c < 3; ++c) {
        double xi = m(p, c), yi = m(q, c);
        m(p, c) = j.c * xi + j.s * yi;
        m(q, c) = -j.s * xi + j.c * yi;
    }
}
// applyOnTheRight(p,q,j) == rotation j.transpose() in the plane of columns p,q : x' = c x - s y ; y' = s x + c y
void rot_right(M3& m, int p, int q, JRot j) {
    for (int r = 0; r < 3; ++r) {
        double xi = m(r, p), yi = m(r, q);
        m(r, p) = j.c * xi - j.s * yi;
        m(r, q) = j.s * xi + j.c * yi;
    }
}
void real_2x2_jacobi_svd(const M3& W, int p, int q, JRot* j_left, JRot* j_right) {
    double m00 = W(p, p), m01 = W(p, q), m10 = W(q, p), m11 = W(q, q);
    JRot rot1;
    double t = m00 + m11;
    double d = m10 - m01;
    if (std::fabs(d) < std::numeric_limits<double>::min()) {
        rot1.s = 0.0;
        rot1.c = 1.0;
    } else {
        double u = t / d;
        double tmp = std::sqrt(1.0 + u * u);
        rot1.s = 1.0 / tmp;
        rot1.c = u / tmp;
    }
    // m.applyOnTheLeft(0,1,rot1)
    double n00 = rot1.c * m00 + rot1.s * m10, n01 = rot1.c * m01 + rot1.s * m11;
    double n11 = -rot1.s * m01 + rot1.c * m11;
    *j_right = make_jacobi(n00, n01, n11);
    // *j_left = rot1 * j_right->transpose()
    JRot jt{j_right->c, -j_right->s};
    j_left->c = rot1.c * jt.c - rot1.s * jt.s;
    j_left->s = rot1.c * jt.s + rot1.s * jt.c;
}
void jacobi_svd3(const M3& A, M3& U, double S[3], M3& V) {
    const double considerAsZero = std::numeric_limits<double>::min();
    const double precision = 2.0 * std::numeric_limits<double>::epsilon();
    double scale = 0.0;
    for (int i = 0; i < 9; ++i) scale = std::max(scale, std::fabs(A.m[i]));
    if (scale == 0.0) scale = 1.0;
    M3 W;
    for (int i = 0; i < 9; ++i) W.m[i] = A.m[i] / scale;
    U = m3_identity();
    V = m3_identity();
    double maxDiag = std::max(std::fabs(W(0, 0)), std::max(std::fabs(W(1, 1)), std::fabs(W(2, 2))));
    bool finished = false;
    int guard = 0;
    while (!finished && guard++ < 1000) {
        finished = true;
        for (int p = 1; p < 3; ++p) {
            for (int q = 0; q < p; ++q) {
                double threshold = std::max(considerAsZero, precision * maxDiag);
                if (std::fabs(W(p, q)) > threshold || std::fabs(W(q, p)) > threshold) {
                    finished = false;
                    JRot jl, jr;
                    real_2x2_jacobi_svd(W, p, q, &jl, &jr);
                    rot_left(W, p, q, jl);
                    JRot jlt{jl.c, -jl.s};
                    rot_right(U, p, q, jlt); // U.applyOnTheRight(p,q,j_left.transpose())
                    rot_right(W, p, q, jr);
                    rot_right(V, p, q, jr);
                    maxDiag = std::max(maxDiag, std::max(std::fabs(W(p, p)), std::fabs(W(q, q))));
                }
            }
        }
    }
    for (int i = 0; i < 3; ++i) {
        double a = std::fabs(W(i, i));
        S[i] = a;
        if (a != 0.0) {
            double sgn = W(i, i) / a;
            for (int r = 0; r < 3; ++r) U(r, i) *= sgn;
        }
    }
    for (int i = 0; i < 3; ++i) S[i] *= scale;
    // sort descending, swapping columns (selection sort as Eigen does)
    for (int i = 0; i < 3; ++i) {
        int pos = i;
        double big = S[i];
        for (int k = i + 1; k < 3; ++k)
            if (S[k] > big) {
                big = S[k];
                pos = k;
            }
        if (big == 0.0) break;
        if (pos != i) {
            std::swap(S[i], S[pos]);
            for (int r = 0; r < 3; ++r) {
                std::swap(U(r, i), U(r, pos));
                std::swap(V(r, i), V(r, pos));
            }
        }
    }
}

// "Plane regularization" used by CalVoxelCov and ProcessVoxelBlock (vhm.hpp:140-145, 240-245):
// cov <- U diag(1,1,1e-3) V^T
M3 plane_regularize(const M3& cov) {
    M3 U, V;
    double S[3];
    jacobi_svd3(cov, U, S, V);
    const double values[3] = {1.0, 1.0, 1e-3};
    M3 UD;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) UD(r, c) = U(r, c) * values[c];
    return m3_mul(UD, m3_transpose(V));
}

// Eigenvector of the smallest eigenvalue of a symmetric 3x3 (reg.cpp:89-91).  The reference uses
// SelfAdjointEigenSolver::compute (tridiagonal QL); only col(0) is used and only through fabs(dot),
// so any accurate symmetric eigen-solver gives the same answer when the smallest eigenvalue is simple.
// cov == Identity (points whose neighbourhood is just themselves, vhm.hpp:223-226): the reference's
// solver returns eigenvectors == Identity for an already-diagonal input, hence col(0) = e_x.
V3 smallest_eigenvector(const M3& Cin) {
    M3 A = Cin;
    // symmetric: only the lower triangle is referenced by Eigen
    A(0, 1) = A(1, 0);
    A(0, 2) = A(2, 0);
    A(1, 2) = A(2, 1);
    M3 Vv = m3_identity();
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = std::fabs(A(1, 0)) + std::fabs(A(2, 0)) + std::fabs(A(2, 1));
        if (off == 0.0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A(p, q) == 0.0) continue;
                double theta = (A(q, q) - A(p, p)) / (2.0 * A(p, q));
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    double akp = A(k, p), akq = A(k, q);
                    A(k, p) = c * akp - s * akq;
                    A(k, q) = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    double apk = A(p, k), aqk = A(q, k);
                    A(p, k) = c * apk - s * aqk;
                    A(q, k) = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = Vv(k, p), vkq = Vv(k, q);
                    Vv(k, p) = c * vkp - s * vkq;
                    Vv(k, q) = s * vkp + c * vkq;
                }
            }
    }
    int best = 0; // first minimum (Eigen's sort keeps the first of equal eigenvalues in place)
    for (int i = 1; i < 3; ++i)
        if (A(i, i) < A(best, best)) best = i;
    return {Vv(0, best), Vv(1, best), Vv(2, best)};
}

// Eigen AngleAxisd(angle, axis).toRotationMatrix() (Eigen/src/Geometry/AngleAxis.h)
M3 angle_axis_matrix(double angle, const V3& axis) {
    M3 res;
    double s = std::sin(angle), c = std::cos(angle);
    V3 sin_axis{s * axis.x, s * axis.y, s * axis.z};
    double c1 = 1.0 - c;
    V3 cos1_axis{c1 * axis.x, c1 * axis.y, c1 * axis.z};
    double tmp;
    tmp = cos1_axis.x * axis.y;
    res(0, 1) = tmp - sin_axis.z;
    res(1, 0) = tmp + sin_axis.z;
    tmp = cos1_axis.x * axis.z;
    res(0, 2) = tmp + sin_axis.y;
    res(2, 0) = tmp - sin_axis.y;
    tmp = cos1_axis.y * axis.z;
    res(1, 2) = tmp - sin_axis.x;
    res(2, 1) = tmp + sin_axis.x;
    res(0, 0) = cos1_axis.x * axis.x + c;
    res(1, 1) = cos1_axis.y * axis.y + c;
    res(2, 2) = cos1_axis.z * axis.z + c;
    return res;
}
// rotation_vector -> matrix exactly as reg.cpp:58-61: AngleAxisd(v.norm(), v.normalized())
M3 rotvec_to_matrix(const V3& v) {
    double n2 = sqn(v);
    V3 axis = v;
    if (n2 > 0.0) { // MatrixBase::normalized(): divide only when squaredNorm > 0
        double n = std::sqrt(n2);
        axis = {v.x / n, v.y / n, v.z / n};
    }
    return angle_axis_matrix(nrm(v), axis);
}
// Eigen AngleAxisd(Matrix3d).angle(): matrix -> quaternion -> angle (reg.cpp:381-382)
double matrix_to_angle(const M3& mat) {
    double qw, qx, qy, qz;
    double t = mat(0, 0) + mat(1, 1) + mat(2, 2);
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
        qw = 0.5 * t;
        t = 0.5 / t;
        qx = (mat(2, 1) - mat(1, 2)) * t;
        qy = (mat(0, 2) - mat(2, 0)) * t;
        qz = (mat(1, 0) - mat(0, 1)) * t;
    } else {
        int i = 0;
        if (mat(1, 1) > mat(0, 0)) i = 1;
        if (mat(2, 2) > mat(i, i)) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(mat(i, i) - mat(j, j) - mat(k, k) + 1.0);
        double q[3];
        q[i] = 0.5 * t;
        t = 0.5 / t;
        qw = (mat(k, j) - mat(j, k)) * t;
        q[j] = (mat(j, i) + mat(i, j)) * t;
        q[k] = (mat(k, i) + mat(i, k)) * t;
        qx = q[0];
        qy = q[1];
        qz = q[2];
    }
    double n = std::sqrt((qx * qx + qy * qy) + qz * qz);
    if (n < std::numeric_limits<double>::epsilon()) {
        // stableNorm(): same value up to scaling safety
        double mx = std::max(std::fabs(qx), std::max(std::fabs(qy), std::fabs(qz)));
        n = (mx == 0.0) ? 0.0 : mx * std::sqrt((qx / mx) * (qx / mx) + (qy / mx) * (qy / mx) + (qz / mx) * (qz / mx));
    }
    if (n != 0.0) return 2.0 * std::atan2(n, std::fabs(qw));
    return 0.0;
}

// ---------------------------------------------------------------------------------------------
// Reference data structures (vhm.hpp:41-155)
// ---------------------------------------------------------------------------------------------
struct CovStruct { // vhm.hpp:41-53
    M3 cov;
    V3 mean;
    CovStruct() : cov(m3_identity()), mean{0, 0, 0} {}
};
struct PointStruct { // vhm.hpp:55-87, sizeof == 168
    V3 pose;
    V3 local;
    CovStruct covariance;
    float vel, azi_angle, ele_angle;
    double intensity;
    PointStruct() : pose{0, 0, 0}, local{0, 0, 0}, covariance(), vel(0), azi_angle(0), ele_angle(0), intensity(0) {}
};
static_assert(sizeof(PointStruct) == 168, "PointStruct layout must match the reference's 168 bytes");

struct Voxel {
    int32_t x, y, z;
    bool operator==(const Voxel& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct VoxelHash { // vhm.hpp:150-155
    size_t operator()(const Voxel& v) const {
        const uint32_t* vec = reinterpret_cast<const uint32_t*>(&v.x);
        return ((1 << 20) - 1) & (vec[0] * 73856093 ^ vec[1] * 19349669 ^ vec[2] * 83492791);
    }
};
struct VoxelBlock { // vhm.hpp:94-149
    std::vector<PointStruct> points;
    CovStruct covariance;
    int num_points;
    double map_resolution;

    void AddPointWithSpacing(const PointStruct& point) { // vhm.hpp:106-113
        if (points.size() < static_cast<size_t>(num_points) &&
            std::none_of(points.cbegin(), points.cend(), [&](const PointStruct& vp) {
                return nrm(sub(vp.pose, point.pose)) < map_resolution;
            })) {
            points.push_back(point);
        }
    }
    void CalVoxelCov() { // vhm.hpp:114-148
        int n = static_cast<int>(points.size());
        covariance.cov = m3_identity();
        covariance.mean = {0, 0, 0};
        if (n == 0) return;
        if (n == 1) {
            covariance.mean = points[0].pose;
            return;
        }
        // rowwise().mean(): sum in index order, divided by n
        V3 mean{0, 0, 0};
        for (int j = 0; j < n; ++j) {
            mean.x += points[j].pose.x;
            mean.y += points[j].pose.y;
            mean.z += points[j].pose.z;
        }
        mean = {mean.x / n, mean.y / n, mean.z / n};
        M3 cov{};
        for (int j = 0; j < n; ++j) {
            double d[3] = {points[j].pose.x - mean.x, points[j].pose.y - mean.y, points[j].pose.z - mean.z};
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) cov(r, c) += d[r] * d[c];
        }
        for (int i = 0; i < 9; ++i) cov.m[i] /= (n - 1);
        covariance.cov = plane_regularize(cov);
        covariance.mean = mean;
    }
};

} // namespace

struct orc_map {
    double voxel_size_;
    int max_points_per_voxel_;
    std::unordered_map<Voxel, VoxelBlock, VoxelHash> map_;
};

namespace {

inline Voxel PointToVoxel(const V3& p, double voxel_size) { // vhm.hpp:176-180 (floor)
    return Voxel{static_cast<int>(std::floor(p.x / voxel_size)), static_cast<int>(std::floor(p.y / voxel_size)),
                 static_cast<int>(std::floor(p.z / voxel_size))};
}

std::vector<Voxel> GetAdjacentVoxels(const orc_map& m, const V3& pose, int range) { // vhm.cpp:208-243
    std::vector<Voxel> voxels;
    Voxel voxel = PointToVoxel(pose, m.voxel_size_);
    int vx = voxel.x, vy = voxel.y, vz = voxel.z;
    const int voxel_neighbor = 1;
    if (range == 0) return std::vector<Voxel>{voxel};
    if (range == 1) {
        return std::vector<Voxel>{{vx, vy, vz},     {vx + 1, vy, vz}, {vx - 1, vy, vz}, {vx, vy + 1, vz},
                                  {vx, vy - 1, vz}, {vx, vy, vz + 1}, {vx, vy, vz - 1}};
    }
    voxels.reserve(27);
    for (int i = vx - voxel_neighbor; i < vx + voxel_neighbor + 1; ++i)
        for (int j = vy - voxel_neighbor; j < vy + voxel_neighbor + 1; ++j)
            for (int k = vz - voxel_neighbor; k < vz + voxel_neighbor + 1; ++k) voxels.push_back(Voxel{i, j, k});
    return voxels;
}

void AddPoints(orc_map& m, const std::vector<PointStruct>& points) { // vhm.cpp:270-285
    if (points.empty()) return;
    const double map_resolution = std::sqrt(m.voxel_size_ * m.voxel_size_ / m.max_points_per_voxel_);
    for (const auto& point : points) {
        // (point.pose / voxel_size_).cast<int>() : truncation toward zero, NOT floor (QUIRK #1)
        Voxel voxel{static_cast<int>(point.pose.x / m.voxel_size_), static_cast<int>(point.pose.y / m.voxel_size_),
                    static_cast<int>(point.pose.z / m.voxel_size_)};
        auto search = m.map_.find(voxel);
        if (search != m.map_.end()) {
            search->second.AddPointWithSpacing(point);
        } else {
            VoxelBlock vb;
            vb.points.push_back(point);
            vb.num_points = m.max_points_per_voxel_;
            vb.map_resolution = map_resolution;
            m.map_.insert({voxel, std::move(vb)});
        }
    }
}

void ProcessVoxelBlock(const orc_map& m, VoxelBlock& voxel_block, double d2max) { // vhm.hpp:195-250
    for (auto& point : voxel_block.points) {
        std::vector<V3> neighbors;
        neighbors.push_back(point.pose); // the point itself ... (QUIRK #2: it is found again below)
        std::vector<Voxel> adjacent = GetAdjacentVoxels(m, point.pose, 2);
        for (const auto& nv : adjacent) {
            auto it = m.map_.find(nv);
            if (it == m.map_.end()) continue;
            for (const auto& np : it->second.points) {
                if (sqn(sub(np.pose, point.pose)) <= d2max) neighbors.push_back(np.pose);
            }
        }
        if (neighbors.size() == 1) {
            point.covariance.cov = m3_identity();
            point.covariance.mean = point.pose;
        } else {
            size_t n = neighbors.size();
            V3 mean{0, 0, 0};
            for (size_t i = 0; i < n; ++i) {
                mean.x += neighbors[i].x;
                mean.y += neighbors[i].y;
                mean.z += neighbors[i].z;
            }
            mean = {mean.x / n, mean.y / n, mean.z / n};
            M3 cov{};
            for (size_t i = 0; i < n; ++i) {
                double d[3] = {neighbors[i].x - mean.x, neighbors[i].y - mean.y, neighbors[i].z - mean.z};
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) cov(r, c) += d[r] * d[c];
            }
            for (int i = 0; i < 9; ++i) cov.m[i] /= static_cast<double>(n - 1);
            point.covariance.cov = plane_regularize(cov);
            point.covariance.mean = mean;
        }
    }
}

template <class F>
void parallel_chunks(size_t n, int threads, F&& f) { // stand-in for the TBB arena: contiguous ordered chunks
    if (threads <= 1 || n < 2) {
        f(0, size_t(0), n);
        return;
    }
    size_t T = std::min<size_t>(threads, n);
    std::vector<std::thread> pool;
    for (size_t t = 0; t < T; ++t) {
        size_t b = n * t / T, e = n * (t + 1) / T;
        pool.emplace_back([&, t, b, e] { f(static_cast<int>(t), b, e); });
    }
    for (auto& th : pool) th.join();
}

struct SearchStats {
    int64_t n_cand = 0, n_occ = 0;
};

// GetCorrespondencePoints (vhm.cpp:31-88)
void GetCorrespondencePoints(const orc_map& m, const std::vector<PointStruct>& vec_points, double max_dist,
                             int threads, std::vector<PointStruct>& out_src, std::vector<PointStruct>& out_tgt,
                             SearchStats* stats) {
    const double d_max_dist_squared = max_dist * max_dist;
    int T = std::max(1, threads);
    std::vector<std::vector<PointStruct>> src(T), tgt(T);
    std::vector<SearchStats> st(T);
    parallel_chunks(vec_points.size(), T, [&](int t, size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            const PointStruct& point = vec_points[i];
            std::vector<Voxel> vec_voxels = GetAdjacentVoxels(m, point.pose, 2);
            PointStruct closest_neighbor; // default: pose (0,0,0), cov I  (QUIRK #3)
            double d_closest = std::numeric_limits<double>::max();
            for (const auto& voxel : vec_voxels) {
                auto search = m.map_.find(voxel);
                if (search != m.map_.end()) {
                    st[t].n_occ++;
                    for (const auto& neighbor : search->second.points) {
                        st[t].n_cand++;
                        double d2 = sqn(sub(neighbor.pose, point.pose));
                        if (d2 < d_closest) {
                            closest_neighbor = neighbor;
                            d_closest = d2;
                        }
                    }
                }
            }
            if (sqn(sub(closest_neighbor.pose, point.pose)) < d_max_dist_squared) {
                src[t].emplace_back(point);
                tgt[t].emplace_back(closest_neighbor);
            }
        }
    });
    out_src.clear();
    out_tgt.clear();
    for (int t = 0; t < T; ++t) { // ordered join (parallel_reduce join is left-to-right)
        out_src.insert(out_src.end(), src[t].begin(), src[t].end());
        out_tgt.insert(out_tgt.end(), tgt[t].begin(), tgt[t].end());
        if (stats) {
            stats->n_cand += st[t].n_cand;
            stats->n_occ += st[t].n_occ;
        }
    }
}

// GetCorrespondencesCov (vhm.cpp:90-151)
void GetCorrespondencesCov(const orc_map& m, const std::vector<PointStruct>& vec_points, double max_dist,
                           int threads, std::vector<PointStruct>& out_src, std::vector<CovStruct>& out_tgt,
                           SearchStats* stats) {
    const double d_max_dist_squared = max_dist * max_dist;
    int T = std::max(1, threads);
    std::vector<std::vector<PointStruct>> src(T);
    std::vector<std::vector<CovStruct>> tgt(T);
    std::vector<SearchStats> st(T);
    parallel_chunks(vec_points.size(), T, [&](int t, size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            const PointStruct& point = vec_points[i];
            std::vector<Voxel> vec_voxels = GetAdjacentVoxels(m, point.pose, 2);
            std::vector<CovStruct> vec_neighbors_cov;
            vec_neighbors_cov.reserve(vec_voxels.size());
            for (const auto& voxel : vec_voxels) {
                auto search = m.map_.find(voxel);
                if (search != m.map_.end() && search->second.points.size() > 0) {
                    vec_neighbors_cov.push_back(search->second.covariance);
                }
            }
            st[t].n_occ += static_cast<int64_t>(vec_neighbors_cov.size());
            CovStruct closest_cov; // default (I, 0)
            double d_closest = std::numeric_limits<double>::max();
            for (const auto& cov : vec_neighbors_cov) {
                st[t].n_cand++;
                double d2 = sqn(sub(cov.mean, point.pose));
                if (d2 < d_closest) {
                    closest_cov = cov;
                    d_closest = d2;
                }
            }
            if (sqn(sub(closest_cov.mean, point.pose)) < d_max_dist_squared) {
                src[t].emplace_back(point);
                tgt[t].emplace_back(closest_cov);
            }
        }
    });
    out_src.clear();
    out_tgt.clear();
    for (int t = 0; t < T; ++t) {
        out_src.insert(out_src.end(), src[t].begin(), src[t].end());
        out_tgt.insert(out_tgt.end(), tgt[t].begin(), tgt[t].end());
        if (stats) {
            stats->n_cand += st[t].n_cand;
            stats->n_occ += st[t].n_occ;
        }
    }
}

// GetCorrespondencesAllCov (vhm.cpp:153-206)
void GetCorrespondencesAllCov(const orc_map& m, const std::vector<PointStruct>& vec_points, double max_dist,
                              int threads, std::vector<PointStruct>& out_src, std::vector<CovStruct>& out_tgt,
                              SearchStats* stats) {
    const double d_max_dist_squared = max_dist * max_dist;
    int T = std::max(1, threads);
    std::vector<std::vector<PointStruct>> src(T);
    std::vector<std::vector<CovStruct>> tgt(T);
    std::vector<SearchStats> st(T);
    parallel_chunks(vec_points.size(), T, [&](int t, size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            const PointStruct& point = vec_points[i];
            std::vector<Voxel> vec_voxels = GetAdjacentVoxels(m, point.pose, 1);
            std::vector<CovStruct> vec_neighbors_cov;
            vec_neighbors_cov.reserve(vec_voxels.size());
            for (const auto& voxel : vec_voxels) {
                auto search = m.map_.find(voxel);
                if (search != m.map_.end() && !search->second.points.empty()) {
                    vec_neighbors_cov.emplace_back(search->second.covariance);
                }
            }
            st[t].n_occ += static_cast<int64_t>(vec_neighbors_cov.size());
            for (const auto& cov : vec_neighbors_cov) {
                st[t].n_cand++;
                if (sqn(sub(cov.mean, point.pose)) < d_max_dist_squared) {
                    src[t].emplace_back(point);
                    tgt[t].emplace_back(cov);
                }
            }
        }
    });
    out_src.clear();
    out_tgt.clear();
    for (int t = 0; t < T; ++t) {
        out_src.insert(out_src.end(), src[t].begin(), src[t].end());
        out_tgt.insert(out_tgt.end(), tgt[t].begin(), tgt[t].end());
        if (stats) {
            stats->n_cand += st[t].n_cand;
            stats->n_occ += st[t].n_occ;
        }
    }
}

// Registration::TransformPoints (reg.hpp:136-148): pose <- T * [pose,1], every other field copied.
void TransformPoints(const M4& T, const std::vector<PointStruct>& points, std::vector<PointStruct>& o_points) {
    o_points.resize(points.size());
    std::transform(points.cbegin(), points.cend(), o_points.begin(), [&](const PointStruct& point) {
        double p[4] = {point.pose.x, point.pose.y, point.pose.z, 1.0};
        double q[3];
        for (int r = 0; r < 3; ++r) q[r] = ((T(r, 0) * p[0] + T(r, 1) * p[1]) + T(r, 2) * p[2]) + T(r, 3) * p[3];
        PointStruct tp = point;
        tp.pose = {q[0], q[1], q[2]};
        return tp;
    });
}

inline double square(double x) { return x * x; } // reg.hpp:219

struct AlignOut {
    M6 JTJ;
    double JTr[6];
    double residual_sum;
    double x[6];
    M4 transformation;
};

// J_g = [ I3 | -[local]x ] (reg.cpp:36-41; skew from reg.hpp:221-225)
inline void make_Jg(const V3& l, double J[3][6]) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 6; ++c) J[r][c] = 0.0;
    J[0][0] = J[1][1] = J[2][2] = 1.0;
    // -1.0 * skew: skew = [0 -z y; z 0 -x; -y x 0]
    J[0][3] = -1.0 * 0.0;
    J[0][4] = -1.0 * -l.z;
    J[0][5] = -1.0 * l.y;
    J[1][3] = -1.0 * l.z;
    J[1][4] = -1.0 * 0.0;
    J[1][5] = -1.0 * -l.x;
    J[2][3] = -1.0 * -l.y;
    J[2][4] = -1.0 * l.x;
    J[2][5] = -1.0 * 0.0;
}

// JTJ += (w J^T) M J ; JTr += (w J^T) M r   (reg.cpp:47-48, 124-125, 204-205). M == nullptr -> identity metric
inline void accumulate(M6& JTJ, double JTr[6], double w, const double J[3][6], const M3* M, const V3& r) {
    double wJt[6][3];
    for (int i = 0; i < 6; ++i)
        for (int k = 0; k < 3; ++k) wJt[i][k] = w * J[k][i];
    double A[6][3];
    if (M) {
        for (int i = 0; i < 6; ++i)
            for (int l = 0; l < 3; ++l)
                A[i][l] = (wJt[i][0] * (*M)(0, l) + wJt[i][1] * (*M)(1, l)) + wJt[i][2] * (*M)(2, l);
    } else {
        for (int i = 0; i < 6; ++i)
            for (int l = 0; l < 3; ++l) A[i][l] = wJt[i][l];
    }
    const double rr[3] = {r.x, r.y, r.z};
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) JTJ(i, j) += (A[i][0] * J[0][j] + A[i][1] * J[1][j]) + A[i][2] * J[2][j];
        JTr[i] += (A[i][0] * rr[0] + A[i][1] * rr[1]) + A[i][2] * rr[2];
    }
}

// the shared tail: LM damping, LDLT solve, exp (reg.cpp:55-65 / 136-151 / 213-224)
inline void solve_and_exp(AlignOut& o, double lm_lambda, M6* regularized_out) {
    M6 reg = o.JTJ;
    for (int i = 0; i < 6; ++i) reg(i, i) = o.JTJ(i, i) + lm_lambda * o.JTJ(i, i);
    if (regularized_out) *regularized_out = reg;
    ldlt_solve6(reg, o.JTr, o.x);
    V3 rotation_vector{o.x[3], o.x[4], o.x[5]};
    M3 R = rotvec_to_matrix(rotation_vector);
    o.transformation = m4_identity();
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o.transformation(r, c) = R(r, c);
    o.transformation(0, 3) = o.x[0];
    o.transformation(1, 3) = o.x[1];
    o.transformation(2, 3) = o.x[2];
}

inline V3 to_local(const M4& Tinv, const V3& g) { // last_icp_pose_inv * [g,1], head<3>
    return {((Tinv(0, 0) * g.x + Tinv(0, 1) * g.y) + Tinv(0, 2) * g.z) + Tinv(0, 3) * 1.0,
            ((Tinv(1, 0) * g.x + Tinv(1, 1) * g.y) + Tinv(1, 2) * g.z) + Tinv(1, 3) * 1.0,
            ((Tinv(2, 0) * g.x + Tinv(2, 1) * g.y) + Tinv(2, 2) * g.z) + Tinv(2, 3) * 1.0};
}

// Eigen::AngleAxisd * Eigen::AngleAxisd assigned to a Matrix3d (reg.hpp:197-200): each AngleAxis becomes a quaternion (w = cos(a/2),
// vec = sin(a/2) axis), the quaternions are multiplied (Eigen/src/Geometry/Quaternion.h, quat_product) and the product is converted
// with QuaternionBase::toRotationMatrix.
M3 yaw_pitch_matrix(double azi_angle, double ele_angle) {
    const double yw = std::cos(azi_angle / 2.0), yz = std::sin(azi_angle / 2.0); // about UnitZ: (w, 0, 0, z)
    const double pw = std::cos(ele_angle / 2.0), py = std::sin(ele_angle / 2.0); // about UnitY: (w, 0, y, 0)
    // a * b with a = (yw; 0, 0, yz), b = (pw; 0, py, 0)
    const double w = yw * pw - 0.0 * 0.0 - 0.0 * py - yz * 0.0;
    const double x = yw * 0.0 + 0.0 * pw + 0.0 * 0.0 - yz * py;
    const double y = yw * py + 0.0 * pw + yz * 0.0 - 0.0 * 0.0;
    const double z = yw * 0.0 + yz * pw + 0.0 * py - 0.0 * 0.0;
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    M3 R;
    R(0, 0) = 1.0 - (tyy + tzz); R(0, 1) = txy - twz; R(0, 2) = txz + twy;
    R(1, 0) = txy + twz; R(1, 1) = 1.0 - (txx + tzz); R(1, 2) = tyz - twx;
    R(2, 0) = txz - twy; R(2, 1) = tyz + twx; R(2, 2) = 1.0 - (txx + tyy);
    return R;
}

// Registration::CalPointCov / CalFramePointCov (reg.hpp:186-217): "covariance" of a radar return from its range / azimuth /
// elevation spreads -- R * S with S = diag(s_x, s_y, s_z), NOT R S R^T: the matrix is not symmetric, and it is built from
// point.pose, which at the call site (reg.cpp:302-305) is the point in the MAP frame under the initial guess.
void CalFramePointCov(std::vector<PointStruct>& points, double range_var_m, double azim_var_deg, double ele_var_deg) {
    for (auto& point : points) {
        const double dist = std::sqrt(point.pose.x * point.pose.x + point.pose.y * point.pose.y);
        const double s_x = range_var_m;
        const double s_y = std::max(0.1, dist * std::sin(azim_var_deg / 180 * M_PI));
        const double s_z = std::max(0.1, dist * std::sin(ele_var_deg / 180 * M_PI));
        const double ele_angle = std::atan2(point.pose.z, dist);
        const double azi_angle = std::atan2(point.pose.y, point.pose.x);
        const M3 R = yaw_pitch_matrix(azi_angle, ele_angle);
        M3 cov;
        for (int r = 0; r < 3; ++r) {
            cov(r, 0) = R(r, 0) * s_x;
            cov(r, 1) = R(r, 1) * s_y;
            cov(r, 2) = R(r, 2) * s_z;
        }
        point.covariance.cov = cov;
    }
}

// AlignCloudsLocal -- P2P (reg.cpp:15-66)
AlignOut AlignCloudsLocal(const std::vector<PointStruct>& source_global, const std::vector<PointStruct>& target_global,
                          const M4& last_icp_pose, double trans_th, const orc_config& cfg, double* fitness) {
    AlignOut o{};
    M4 inv = m4_inverse(last_icp_pose);
    double d_residual_sum = 0.0;
    for (size_t i = 0; i < source_global.size(); ++i) {
        V3 target_local = to_local(inv, target_global[i].pose);
        V3 residual_local = sub(target_local, source_global[i].local);
        double J[3][6];
        make_Jg(source_global[i].local, J);
        double weight_g = square(trans_th) / square(trans_th + sqn(residual_local));
        accumulate(o.JTJ, o.JTr, weight_g, J, nullptr, residual_local);
        d_residual_sum += nrm(residual_local);
    }
    *fitness = d_residual_sum / source_global.size();
    o.residual_sum = d_residual_sum;
    solve_and_exp(o, cfg.lm_lambda, nullptr);
    return o;
}

// AlignCloudsLocalPointCov -- GICP (reg.cpp:68-152)
AlignOut AlignCloudsLocalPointCov(const std::vector<PointStruct>& source_global,
                                  const std::vector<PointStruct>& target_global, M6& local_cov,
                                  const M4& last_icp_pose, double trans_th, const orc_config& cfg, double* fitness) {
    AlignOut o{};
    M3 sensor_rot;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) sensor_rot(r, c) = last_icp_pose(r, c);
    M3 sensor_rot_inv = m3_inverse(sensor_rot);
    M4 inv = m4_inverse(last_icp_pose);
    double d_residual_sum = 0.0;
    for (size_t i = 0; i < source_global.size(); ++i) {
        const CovStruct& target_cov = target_global[i].covariance;
        V3 vec_normal_global = smallest_eigenvector(target_cov.cov);
        V3 vec_normal_local = m3_mulv(sensor_rot_inv, vec_normal_global);
        double nn = nrm(vec_normal_local);
        if (sqn(vec_normal_local) > 0.0) vec_normal_local = {vec_normal_local.x / nn, vec_normal_local.y / nn, vec_normal_local.z / nn};
        V3 target_local = to_local(inv, target_cov.mean); // the neighbourhood MEAN, not the matched point (reg.cpp:97)
        V3 residual_local = sub(target_local, source_global[i].local);
        M3 RCR = m3_mul(m3_mul(sensor_rot_inv, target_cov.cov), m3_transpose(sensor_rot_inv));
        if (cfg.use_radar_cov) // reg.cpp:109-111
            for (int k = 0; k < 9; ++k) RCR.m[k] += source_global[i].covariance.cov.m[k];
        M3 mahalanobis_local = m3_inverse(RCR);
        double J[3][6];
        make_Jg(source_global[i].local, J);
        double weight_g = square(trans_th) / square(trans_th + sqn(residual_local)) * 0.8 + 0.2;
        accumulate(o.JTJ, o.JTr, weight_g, J, &mahalanobis_local, residual_local);
        double d_point_to_plane_dist = std::fabs((residual_local.x * vec_normal_local.x + residual_local.y * vec_normal_local.y) +
                                                 residual_local.z * vec_normal_local.z);
        d_residual_sum += d_point_to_plane_dist;
    }
    *fitness = d_residual_sum / source_global.size();
    o.residual_sum = d_residual_sum;
    M6 regularized;
    solve_and_exp(o, cfg.lm_lambda, &regularized);
    local_cov = m6_inverse(regularized); // reg.cpp:141-142
    return o;
}

// AlignCloudsLocalVoxelCov -- VGICP / AVGICP (reg.cpp:154-225)
AlignOut AlignCloudsLocalVoxelCov(const std::vector<PointStruct>& source_global,
                                  const std::vector<CovStruct>& target_cov_global, const M4& last_icp_pose,
                                  double trans_th, const orc_config& cfg, double* fitness) {
    AlignOut o{};
    M3 sensor_rot;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) sensor_rot(r, c) = last_icp_pose(r, c);
    M3 sensor_rot_inv = m3_inverse(sensor_rot);
    M4 inv = m4_inverse(last_icp_pose);
    double d_residual_sum = 0.0;
    for (size_t i = 0; i < source_global.size(); ++i) {
        const CovStruct& target_cov = target_cov_global[i];
        V3 target_local = to_local(inv, target_cov.mean);
        V3 residual_local = sub(target_local, source_global[i].local);
        M3 RCR = m3_mul(m3_mul(sensor_rot_inv, target_cov.cov), m3_transpose(sensor_rot_inv));
        if (cfg.use_radar_cov) // reg.cpp:188-190
            for (int k = 0; k < 9; ++k) RCR.m[k] += source_global[i].covariance.cov.m[k];
        M3 mahalanobis_local = m3_inverse(RCR);
        double J[3][6];
        make_Jg(source_global[i].local, J);
        double weight_g = square(trans_th) / square(trans_th + sqn(residual_local));
        if (weight_g < 0.01) continue; // reg.cpp:201
        accumulate(o.JTJ, o.JTr, weight_g, J, &mahalanobis_local, residual_local);
        d_residual_sum += nrm(residual_local);
    }
    *fitness = d_residual_sum / source_global.size();
    o.residual_sum = d_residual_sum;
    solve_and_exp(o, cfg.lm_lambda, nullptr);
    return o;
}

using Clock = std::chrono::steady_clock;
inline double ms_since(Clock::time_point a, Clock::time_point b) {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() * 1e-6;
}

// Pcl2PointStruct (pcm.hpp:205-220): float -> double, local = pose
void Pcl2PointStruct(const float* xyz, size_t n, std::vector<PointStruct>& vec_points) {
    vec_points.clear();
    vec_points.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        PointStruct ps;
        ps.pose = {static_cast<double>(xyz[3 * i]), static_cast<double>(xyz[3 * i + 1]), static_cast<double>(xyz[3 * i + 2])};
        ps.local = ps.pose;
        vec_points.emplace_back(std::move(ps));
    }
}

} // namespace

// =================================================================================================
// C interface
// =================================================================================================
extern "C" {

orc_map* orc_map_create(double voxel_size, int max_points_per_voxel) { // VoxelHashMap::Init vhm.cpp:26-29
    orc_map* m = new orc_map();
    m->voxel_size_ = voxel_size;
    m->max_points_per_voxel_ = max_points_per_voxel;
    return m;
}
void orc_map_destroy(orc_map* m) { delete m; }

void orc_map_add_points(orc_map* m, const float* xyz, size_t n) {
    std::vector<PointStruct> pts;
    Pcl2PointStruct(xyz, n, pts);
    AddPoints(*m, pts);
}

void orc_map_cal_voxel_cov_all(orc_map* m, int threads) { // vhm.hpp:183-193
    std::vector<VoxelBlock*> blocks;
    blocks.reserve(m->map_.size());
    for (auto& kv : m->map_) blocks.push_back(&kv.second);
    parallel_chunks(blocks.size(), threads, [&](int, size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) blocks[i]->CalVoxelCov();
    });
}

void orc_map_cal_point_cov_all(orc_map* m, double dist, int threads) { // vhm.hpp:252-257
    double d2 = dist * dist;
    std::vector<VoxelBlock*> blocks;
    blocks.reserve(m->map_.size());
    for (auto& kv : m->map_) blocks.push_back(&kv.second);
    parallel_chunks(blocks.size(), threads, [&](int, size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) ProcessVoxelBlock(*m, *blocks[i], d2);
    });
}

size_t orc_map_num_points(const orc_map* m) {
    size_t n = 0;
    for (const auto& kv : m->map_) n += kv.second.points.size();
    return n;
}
size_t orc_map_num_voxels(const orc_map* m) { return m->map_.size(); }
int orc_map_empty(const orc_map* m) { return m->map_.empty() ? 1 : 0; }

size_t orc_map_pointcloud(const orc_map* m, double* xyz, double* cov9, double* mean3, size_t cap) {
    size_t n = 0;
    for (const auto& kv : m->map_) {
        for (const auto& p : kv.second.points) {
            if (n < cap) {
                if (xyz) {
                    xyz[3 * n] = p.pose.x;
                    xyz[3 * n + 1] = p.pose.y;
                    xyz[3 * n + 2] = p.pose.z;
                }
                if (cov9) std::memcpy(cov9 + 9 * n, p.covariance.cov.m, 9 * sizeof(double));
                if (mean3) {
                    mean3[3 * n] = p.covariance.mean.x;
                    mean3[3 * n + 1] = p.covariance.mean.y;
                    mean3[3 * n + 2] = p.covariance.mean.z;
                }
            }
            ++n;
        }
    }
    return n;
}

size_t orc_map_voxels(const orc_map* m, int32_t* key3, int32_t* npts, double* cov9, double* mean3, size_t cap) {
    size_t n = 0;
    for (const auto& kv : m->map_) {
        if (n < cap) {
            if (key3) {
                key3[3 * n] = kv.first.x;
                key3[3 * n + 1] = kv.first.y;
                key3[3 * n + 2] = kv.first.z;
            }
            if (npts) npts[n] = static_cast<int32_t>(kv.second.points.size());
            if (cov9) std::memcpy(cov9 + 9 * n, kv.second.covariance.cov.m, 9 * sizeof(double));
            if (mean3) {
                mean3[3 * n] = kv.second.covariance.mean.x;
                mean3[3 * n + 1] = kv.second.covariance.mean.y;
                mean3[3 * n + 2] = kv.second.covariance.mean.z;
            }
        }
        ++n;
    }
    return n;
}

int orc_map_find_ground_height(const orc_map* m, double px, double py, double* ground_z) { // vhm.hpp:285-322
    const double r2 = 5.0 * 5.0;
    std::vector<double> zs;
    for (const auto& kv : m->map_)
        for (const auto& p : kv.second.points) {
            double dx = p.pose.x - px, dy = p.pose.y - py;
            if (dx * dx + dy * dy <= r2) zs.push_back(p.pose.z);
        }
    if (zs.size() <= 3) return 0;
    size_t N = std::min<size_t>(5, zs.size());
    std::partial_sort(zs.begin(), zs.begin() + N, zs.end());
    double s = 0.0;
    for (size_t i = 0; i < N; ++i) s += zs[i];
    *ground_z = s / N;
    return 1;
}

void orc_nearest_points(const orc_map* m, const double* q, size_t n, double max_dist, int threads,
                        uint8_t* accepted, double* tgt_xyz, double* d2out) {
    const double d2max = max_dist * max_dist;
    parallel_chunks(n, threads, [&](int, size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            V3 pose{q[3 * i], q[3 * i + 1], q[3 * i + 2]};
            std::vector<Voxel> vox = GetAdjacentVoxels(*m, pose, 2);
            V3 best{0, 0, 0};
            double dbest = std::numeric_limits<double>::max();
            for (const auto& v : vox) {
                auto it = m->map_.find(v);
                if (it == m->map_.end()) continue;
                for (const auto& nb : it->second.points) {
                    double d2 = sqn(sub(nb.pose, pose));
                    if (d2 < dbest) {
                        best = nb.pose;
                        dbest = d2;
                    }
                }
            }
            double dfin = sqn(sub(best, pose));
            accepted[i] = dfin < d2max ? 1 : 0;
            tgt_xyz[3 * i] = best.x;
            tgt_xyz[3 * i + 1] = best.y;
            tgt_xyz[3 * i + 2] = best.z;
            if (d2out) d2out[i] = dfin;
        }
    });
}

void orc_nearest_voxel(const orc_map* m, const double* q, size_t n, double max_dist, int threads, uint8_t* accepted,
                       double* mean_xyz, double* cov9) {
    const double d2max = max_dist * max_dist;
    parallel_chunks(n, threads, [&](int, size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            V3 pose{q[3 * i], q[3 * i + 1], q[3 * i + 2]};
            std::vector<Voxel> vox = GetAdjacentVoxels(*m, pose, 2);
            CovStruct best;
            double dbest = std::numeric_limits<double>::max();
            for (const auto& v : vox) {
                auto it = m->map_.find(v);
                if (it == m->map_.end() || it->second.points.empty()) continue;
                double d2 = sqn(sub(it->second.covariance.mean, pose));
                if (d2 < dbest) {
                    best = it->second.covariance;
                    dbest = d2;
                }
            }
            accepted[i] = sqn(sub(best.mean, pose)) < d2max ? 1 : 0;
            mean_xyz[3 * i] = best.mean.x;
            mean_xyz[3 * i + 1] = best.mean.y;
            mean_xyz[3 * i + 2] = best.mean.z;
            if (cov9) std::memcpy(cov9 + 9 * i, best.cov.m, 9 * sizeof(double));
        }
    });
}

// GetCorrespondencesAllCov (vhm.cpp:153-206) as a call of its own: the pairs in input order (one range, joined in order) as
// (source index, target mean, target covariance); returns the number of pairs (at most cap are written)
size_t orc_all_cov_pairs(const orc_map* m, const double* q, size_t n, double max_dist, uint32_t* src_index, double* mean_xyz,
                         double* cov9, size_t cap) {
    const double d_max_dist_squared = max_dist * max_dist;
    size_t k = 0;
    for (size_t i = 0; i < n; ++i) {
        V3 pose{q[3 * i], q[3 * i + 1], q[3 * i + 2]};
        std::vector<Voxel> vec_voxels = GetAdjacentVoxels(*m, pose, 1);
        std::vector<CovStruct> vec_neighbors_cov;
        vec_neighbors_cov.reserve(vec_voxels.size());
        for (const auto& voxel : vec_voxels) {
            auto search = m->map_.find(voxel);
            if (search != m->map_.end() && !search->second.points.empty()) vec_neighbors_cov.emplace_back(search->second.covariance);
        }
        for (const auto& cov : vec_neighbors_cov) {
            if (sqn(sub(cov.mean, pose)) < d_max_dist_squared) {
                if (k < cap) {
                    src_index[k] = static_cast<uint32_t>(i);
                    mean_xyz[3 * k] = cov.mean.x; mean_xyz[3 * k + 1] = cov.mean.y; mean_xyz[3 * k + 2] = cov.mean.z;
                    if (cov9) std::memcpy(cov9 + 9 * k, cov.cov.m, 9 * sizeof(double));
                }
                ++k;
            }
        }
    }
    return k;
}

// Registration::AlignCloudsLocal / AlignCloudsLocalPointCov / AlignCloudsLocalVoxelCov (reg.cpp:15-225) as calls of their own.
// method 0 / 1 / 2,3; src_local = PointStruct::local, tgt_xyz = target pose (P2P) or covariance.mean, covariances column-major
void orc_align_clouds_local(int method, const double* src_local, const double* tgt_xyz, const double* tgt_cov9, const double* src_cov9,
                            size_t n, const double last_icp_pose[16], double trans_th, const orc_config* cfg, double T_out[16],
                            double local_cov[36], double* fitness, double JTJ_out[36], double JTr_out[6]) {
    std::vector<PointStruct> src(n), tgt(n);
    std::vector<CovStruct> tcov(n);
    for (size_t i = 0; i < n; ++i) {
        src[i].local = {src_local[3 * i], src_local[3 * i + 1], src_local[3 * i + 2]};
        src[i].pose = src[i].local;
        if (src_cov9) std::memcpy(src[i].covariance.cov.m, src_cov9 + 9 * i, 9 * sizeof(double));
        tgt[i].pose = {tgt_xyz[3 * i], tgt_xyz[3 * i + 1], tgt_xyz[3 * i + 2]};
        if (tgt_cov9) {
            std::memcpy(tcov[i].cov.m, tgt_cov9 + 9 * i, 9 * sizeof(double));
            tcov[i].mean = tgt[i].pose;
            tgt[i].covariance = tcov[i];
        }
    }
    M4 T;
    std::memcpy(T.m, last_icp_pose, 16 * sizeof(double));
    AlignOut o{};
    M6 cov;
    for (int k = 0; k < 36; ++k) cov.m[k] = (k % 7 == 0) ? 1.0 : 0.0;
    if (method == 0) o = AlignCloudsLocal(src, tgt, T, trans_th, *cfg, fitness);
    else if (method == 1) o = AlignCloudsLocalPointCov(src, tgt, cov, T, trans_th, *cfg, fitness);
    else o = AlignCloudsLocalVoxelCov(src, tcov, T, trans_th, *cfg, fitness);
    std::memcpy(T_out, o.transformation.m, 16 * sizeof(double));
    if (local_cov) std::memcpy(local_cov, cov.m, 36 * sizeof(double));
    if (JTJ_out) std::memcpy(JTJ_out, o.JTJ.m, 36 * sizeof(double));
    if (JTr_out) std::memcpy(JTr_out, o.JTr, 6 * sizeof(double));
}

// Registration::CalFramePointCov (reg.hpp:211-217) as a call of its own: cov9 = n column-major 3x3
void orc_cal_frame_point_cov(const double* xyz, size_t n, double range_var_m, double azim_var_deg, double ele_var_deg, double* cov9) {
    std::vector<PointStruct> pts(n);
    for (size_t i = 0; i < n; ++i) pts[i].pose = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    CalFramePointCov(pts, range_var_m, azim_var_deg, ele_var_deg);
    for (size_t i = 0; i < n; ++i) std::memcpy(cov9 + 9 * i, pts[i].covariance.cov.m, 9 * sizeof(double));
}

// RunRegister (reg.cpp:274-418)
void orc_register(const orc_map* mp, const float* scan_xyz, size_t n, const double T0[16], const orc_config* cfgp,
                  orc_result* out) {
    const orc_map& voxel_map = *mp;
    const orc_config& cfg = *cfgp;
    std::memset(out, 0, sizeof(*out));
    std::vector<PointStruct> source_local;
    Pcl2PointStruct(scan_xyz, n, source_local);

    M4 initial_guess;
    std::memcpy(initial_guess.m, T0, sizeof(initial_guess.m));

    std::vector<PointStruct> source_c_global, target_c_global;
    std::vector<CovStruct> target_cov_c_global;
    M6 local_cov{};
    for (int i = 0; i < 6; ++i) local_cov(i, i) = 1.0; // reg.cpp:280
    auto finish = [&](const M4& T, bool ok, int gate, int iters, double fit) {
        std::memcpy(out->T, T.m, sizeof(out->T));
        out->is_success = ok ? 1 : 0;
        out->gate = gate;
        out->iterations = iters;
        out->fitness = fit;
        std::memcpy(out->local_cov, local_cov.m, sizeof(out->local_cov));
    };

    int i_source_total_num = static_cast<int>(source_local.size());
    int i_source_corr_num = 0;
    double corres_ratio = 0.0;
    double d_fitness_score_ = 0.0;

    std::vector<PointStruct> source_global;
    source_global.resize(source_local.size());
    TransformPoints(initial_guess, source_local, source_global); // reg.cpp:289

    if (voxel_map.map_.empty()) { // reg.cpp:291-295
        finish(initial_guess, false, 1, 0, 0.0);
        return;
    }

    M4 last_icp_pose = initial_guess;
    // reg.cpp:302-305: the radar covariances are attached to source_global ONCE, from the poses under the initial guess.  The
    // re-transform at the end of every iteration (reg.cpp:390) rebuilds source_global from source_local and copies ITS covariance
    // (the default identity of Pcl2PointStruct's points, pcm.hpp:205-220), so only the first iteration sees them.
    if (cfg.use_radar_cov)
        CalFramePointCov(source_global, cfg.range_variance_m, cfg.azimuth_variance_deg, cfg.elevation_variance_deg);
    auto start = Clock::now();
    int i_iteration = 0;
    double total_correspondence_time_ms = 0.0;
    for (int j = 0; j < cfg.max_iteration; ++j) {
        i_iteration++;
        SearchStats st;
        auto c0 = Clock::now();
        switch (cfg.icp_method) {
        case ORC_P2P:
        case ORC_GICP:
            GetCorrespondencePoints(voxel_map, source_global, cfg.max_search_dist, cfg.max_thread, source_c_global,
                                    target_c_global, &st);
            break;
        case ORC_VGICP:
            GetCorrespondencesCov(voxel_map, source_global, cfg.max_search_dist, cfg.max_thread, source_c_global,
                                  target_cov_c_global, &st);
            break;
        case ORC_AVGICP:
            GetCorrespondencesAllCov(voxel_map, source_global, cfg.max_search_dist, cfg.max_thread, source_c_global,
                                     target_cov_c_global, &st);
            break;
        }
        total_correspondence_time_ms += ms_since(c0, Clock::now());

        i_source_corr_num = static_cast<int>(source_c_global.size());
        corres_ratio = (float)i_source_corr_num / i_source_total_num; // reg.cpp:351 (float division)
        orc_iter_trace* tr = (i_iteration <= ORC_MAX_ITER_TRACE) ? &out->iters[i_iteration - 1] : nullptr;
        if (tr) {
            tr->n_corr = i_source_corr_num;
            tr->n_cand = st.n_cand;
            tr->n_occ = st.n_occ;
        }
        if (corres_ratio < cfg.min_overlap_ratio) { // reg.cpp:352-356
            out->elapsed_ms = ms_since(start, Clock::now());
            out->correspondence_ms = total_correspondence_time_ms;
            finish(last_icp_pose, false, 2, i_iteration, d_fitness_score_);
            return;
        }

        AlignOut est;
        switch (cfg.icp_method) {
        case ORC_P2P:
            est = AlignCloudsLocal(source_c_global, target_c_global, last_icp_pose, cfg.max_search_dist, cfg,
                                   &d_fitness_score_);
            break;
        case ORC_GICP:
            est = AlignCloudsLocalPointCov(source_c_global, target_c_global, local_cov, last_icp_pose,
                                           cfg.max_search_dist, cfg, &d_fitness_score_);
            break;
        default:
            est = AlignCloudsLocalVoxelCov(source_c_global, target_cov_c_global, last_icp_pose, cfg.max_search_dist,
                                           cfg, &d_fitness_score_);
            break;
        }

        last_icp_pose = m4_mul(last_icp_pose, est.transformation); // reg.cpp:378

        M3 dR;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) dR(r, c) = est.transformation(r, c);
        double rot_norm = matrix_to_angle(dR);
        V3 dt{est.transformation(0, 3), est.transformation(1, 3), est.transformation(2, 3)};
        double transform_norm = rot_norm + nrm(dt);
        if (tr) {
            std::memcpy(tr->JTJ, est.JTJ.m, sizeof(tr->JTJ));
            std::memcpy(tr->JTr, est.JTr, sizeof(tr->JTr));
            tr->residual_sum = est.residual_sum;
            std::memcpy(tr->x, est.x, sizeof(tr->x));
            tr->step_norm = transform_norm;
            std::memcpy(tr->T, last_icp_pose.m, sizeof(tr->T));
        }
        if (transform_norm < cfg.icp_termination_threshold_m) break; // reg.cpp:385-387

        TransformPoints(last_icp_pose, source_local, source_global); // reg.cpp:390
    }
    out->elapsed_ms = ms_since(start, Clock::now());
    out->correspondence_ms = total_correspondence_time_ms;

    if (d_fitness_score_ > cfg.max_fitness_score) { // reg.cpp:405-409
        finish(last_icp_pose, false, 3, i_iteration, d_fitness_score_);
        return;
    }
    finish(last_icp_pose, true, 0, i_iteration, d_fitness_score_);
}

size_t orc_voxel_downsample(const float* xyz, size_t n, double voxel_size, int64_t* keep_idx) { // vhm.hpp:260-283
    std::unordered_map<Voxel, int64_t, VoxelHash> grid;
    grid.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        V3 p{(double)xyz[3 * i], (double)xyz[3 * i + 1], (double)xyz[3 * i + 2]};
        Voxel v = PointToVoxel(p, voxel_size);
        if (grid.find(v) == grid.end()) grid.insert({v, (int64_t)i});
    }
    size_t k = 0;
    for (const auto& kv : grid) keep_idx[k++] = kv.second;
    std::sort(keep_idx, keep_idx + k);
    return k;
}

// ------------------------------------------------------------------------------------------------
// deskew (pcm.cpp:467-824) -- float32 semantics
// ------------------------------------------------------------------------------------------------
namespace {
// pcl::getTransformation(x,y,z,roll,pitch,yaw) with Scalar = float (pcl/common/impl/eigen.hpp)
struct Aff3f {
    float m[3][4];
};
Aff3f pcl_getTransformation(float x, float y, float z, float roll, float pitch, float yaw) {
    float A = std::cos(yaw), B = std::sin(yaw), C = std::cos(pitch), D = std::sin(pitch), E = std::cos(roll),
          F = std::sin(roll), DE = D * E, DF = D * F;
    Aff3f t;
    t.m[0][0] = A * C;
    t.m[0][1] = A * DF - B * E;
    t.m[0][2] = B * F + A * DE;
    t.m[0][3] = x;
    t.m[1][0] = B * C;
    t.m[1][1] = A * E + B * DF;
    t.m[1][2] = B * DE - A * F;
    t.m[1][3] = y;
    t.m[2][0] = -D;
    t.m[2][1] = C * F;
    t.m[2][2] = C * E;
    t.m[2][3] = z;
    return t;
}

// FindRotation (pcm.cpp:731-762)
void FindRotation(const orc_deskew_tables& tb, double d_point_time, float* rx, float* ry, float* rz) {
    *rx = 0;
    *ry = 0;
    *rz = 0;
    int front = 0;
    while (front < tb.imu_pointer_cur) {
        if (d_point_time < tb.imu_time[front]) break;
        ++front;
    }
    if (d_point_time > tb.imu_time[front] || front == 0) {
        *rx = tb.imu_rot_x[front];
        *ry = tb.imu_rot_y[front];
        *rz = tb.imu_rot_z[front];
    } else {
        int back = front - 1;
        double ratio_front = (d_point_time - tb.imu_time[back]) / (tb.imu_time[front] - tb.imu_time[back]);
        double ratio_back = (tb.imu_time[front] - d_point_time) / (tb.imu_time[front] - tb.imu_time[back]);
        *rx = tb.imu_rot_x[front] * ratio_front + tb.imu_rot_x[back] * ratio_back;
        *ry = tb.imu_rot_y[front] * ratio_front + tb.imu_rot_y[back] * ratio_back;
        *rz = tb.imu_rot_z[front] * ratio_front + tb.imu_rot_z[back] * ratio_back;
    }
}
// FindPosition (pcm.cpp:764-778)
void FindPosition(const orc_deskew_tables& tb, double d_rel_time, float* px, float* py, float* pz) {
    *px = 0;
    *py = 0;
    *pz = 0;
    if (!tb.odom_available) return;
    float f_ratio = d_rel_time / (tb.time_scan_end - tb.time_scan_cur);
    *px = f_ratio * tb.odom_incre_x;
    *py = f_ratio * tb.odom_incre_y;
    *pz = f_ratio * tb.odom_incre_z;
}
} // namespace

void orc_deskew_points(const float* xyz_in, const float* rel_time, size_t n, const orc_deskew_tables* tab,
                       float* xyz_out) {
    const orc_deskew_tables& tb = *tab;
    for (size_t i = 0; i < n; ++i) {
        float x = xyz_in[3 * i], y = xyz_in[3 * i + 1], z = xyz_in[3 * i + 2];
        if (!tb.run_deskew) { // pcm.cpp:513-525
            xyz_out[3 * i] = x;
            xyz_out[3 * i + 1] = y;
            xyz_out[3 * i + 2] = z;
            continue;
        }
        // DeskewPoint (pcm.cpp:780-824); d_rel_time is the float point time promoted to double
        double d_rel_time = rel_time[i];
        double d_point_time = tb.time_scan_cur + d_rel_time;
        float f_rot_x_end = tb.imu_rot_x[tb.imu_pointer_cur];
        float f_rot_y_end = tb.imu_rot_y[tb.imu_pointer_cur];
        float f_rot_z_end = tb.imu_rot_z[tb.imu_pointer_cur];
        float rxc, ryc, rzc;
        FindRotation(tb, d_point_time, &rxc, &ryc, &rzc);
        float pxc, pyc, pzc;
        FindPosition(tb, d_rel_time, &pxc, &pyc, &pzc);
        float f_rot_x_from_end = rxc - f_rot_x_end;
        float f_rot_y_from_end = ryc - f_rot_y_end;
        float f_rot_z_from_end = rzc - f_rot_z_end;
        float f_pos_x_from_end = pxc - tb.odom_incre_x;
        float f_pos_y_from_end = pyc - tb.odom_incre_y;
        float f_pos_z_from_end = rzc - tb.odom_incre_z; // QUIRK #4 (pcm.cpp:804): rot_z, not pos_z
        (void)pzc;
        Aff3f t = pcl_getTransformation(f_pos_x_from_end, f_pos_y_from_end, f_pos_z_from_end, f_rot_x_from_end,
                                        f_rot_y_from_end, f_rot_z_from_end);
        xyz_out[3 * i] = t.m[0][0] * x + t.m[0][1] * y + t.m[0][2] * z + t.m[0][3];
        xyz_out[3 * i + 1] = t.m[1][0] * x + t.m[1][1] * y + t.m[1][2] * z + t.m[1][3];
        xyz_out[3 * i + 2] = t.m[2][0] * x + t.m[2][1] * y + t.m[2][2] * z + t.m[2][3];
    }
}

int orc_imu_deskew_info(const double* imu_t, const double* imu_w, size_t n_imu, double scan_cur, double scan_end,
                        double* tab_time, double* tab_rx, double* tab_ry, double* tab_rz, int32_t* imu_pointer_cur) {
    // pcm.cpp:533-585.  The deque pops (samples older than scan_cur - 0.01) are applied here as a skip.
    size_t first = 0;
    while (first < n_imu && imu_t[first] < scan_cur - 0.01) ++first;
    *imu_pointer_cur = 0;
    if (first >= n_imu) return 0;
    int cur = 0;
    for (size_t i = first; i < n_imu; ++i) {
        double t = imu_t[i];
        if (t > scan_end + 0.01) break;
        if (cur >= 2000) break; // the reference has no bounds check on its 2000-entry arrays (pcm.hpp:113)
        if (cur == 0) {
            tab_rx[0] = 0;
            tab_ry[0] = 0;
            tab_rz[0] = 0;
            tab_time[0] = t;
            ++cur;
            continue;
        }
        double dt = t - tab_time[cur - 1];
        tab_rx[cur] = tab_rx[cur - 1] + imu_w[3 * i] * dt;
        tab_ry[cur] = tab_ry[cur - 1] + imu_w[3 * i + 1] * dt;
        tab_rz[cur] = tab_rz[cur - 1] + imu_w[3 * i + 2] * dt;
        tab_time[cur] = t;
        ++cur;
    }
    --cur;
    *imu_pointer_cur = cur;
    if (cur <= 0) return 0;
    return 1;
}

namespace {
// tf::Matrix3x3(q).getRPY (tf/LinearMath/Matrix3x3.h: setRotation + getEulerYPR, solution 1)
void tf_quat_to_rpy(double x, double y, double z, double w, double* roll, double* pitch, double* yaw) {
    double d = x * x + y * y + z * z + w * w;
    double s = 2.0 / d;
    double xs = x * s, ys = y * s, zs = z * s;
    double wx = w * xs, wy = w * ys, wz = w * zs;
    double xx = x * xs, xy = x * ys, xz = x * zs;
    double yy = y * ys, yz = y * zs, zz = z * zs;
    double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
    if (std::fabs(m20) >= 1) { // gimbal lock (never reached by the synthetic streams)
        *yaw = 0;
        double delta = std::atan2(m21, m22);
        *pitch = (m20 < 0) ? M_PI / 2.0 : -M_PI / 2.0;
        *roll = delta;
    } else {
        *pitch = -std::asin(m20);
        *roll = std::atan2(m21 / std::cos(*pitch), m22 / std::cos(*pitch));
        *yaw = std::atan2(m10 / std::cos(*pitch), m00 / std::cos(*pitch));
    }
}
struct M4f {
    float m[4][4];
};
M4f aff_to_m4(const Aff3f& a) {
    M4f r{};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[i][j];
    r.m[3][3] = 1.0f;
    return r;
}
} // namespace

int orc_odom_deskew_info(const double* od, size_t n_odom, double scan_cur, double scan_end, float* incre_xyz) {
    // pcm.cpp:587-729.  The deque pops (odom older than scan_cur - 0.1) are applied here as a skip.
    incre_xyz[0] = incre_xyz[1] = incre_xyz[2] = 0.0f;
    size_t first = 0;
    while (first < n_odom && od[14 * first] < scan_cur - 0.1) ++first;
    if (first >= n_odom) return 0;
    if (od[14 * first] > scan_cur) return 0;
    auto row = [&](size_t i) { return od + 14 * i; };
    // start odometry: first sample with stamp >= scan_cur (or the last one)
    size_t si = first;
    for (size_t i = first; i < n_odom; ++i) {
        si = i;
        if (row(i)[0] < scan_cur) continue;
        break;
    }
    const double* so = row(si);
    double roll, pitch, yaw;
    tf_quat_to_rpy(so[4], so[5], so[6], so[7], &roll, &pitch, &yaw);
    Aff3f begin = pcl_getTransformation((float)so[1], (float)so[2], (float)so[3], (float)roll, (float)pitch, (float)yaw);

    double end_stamp, ex, ey, ez, eqx, eqy, eqz, eqw;
    const double* lo = row(n_odom - 1);
    if (lo[0] > scan_end) {
        size_t ei = first;
        for (size_t i = first; i < n_odom; ++i) {
            ei = i;
            if (row(i)[0] < scan_end) continue;
            break;
        }
        const double* eo = row(ei);
        end_stamp = eo[0];
        ex = eo[1];
        ey = eo[2];
        ez = eo[3];
        eqx = eo[4];
        eqy = eo[5];
        eqz = eo[6];
        eqw = eo[7];
    } else {
        double dt = scan_end - lo[0];
        end_stamp = scan_end; // ros::Time(d_time_scan_end_) (ns rounding of ros::Time ignored)
        double r2, p2, y2;
        tf_quat_to_rpy(lo[4], lo[5], lo[6], lo[7], &r2, &p2, &y2);
        // R = Rz(yaw) Ry(pitch) Rx(roll) (Eigen AngleAxisd products, double)
        double cy = std::cos(y2), sy = std::sin(y2), cp = std::cos(p2), sp = std::sin(p2), cr = std::cos(r2),
               sr = std::sin(r2);
        double R[3][3] = {{cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr},
                          {sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr},
                          {-sp, cp * sr, cp * cr}};
        double gv[3];
        for (int i = 0; i < 3; ++i) gv[i] = R[i][0] * lo[8] + R[i][1] * lo[9] + R[i][2] * lo[10];
        ex = lo[1] + gv[0] * dt;
        ey = lo[2] + gv[1] * dt;
        ez = lo[3] + gv[2] * dt;
        r2 += lo[11] * dt;
        p2 += lo[12] * dt;
        y2 += lo[13] * dt;
        // tf::Quaternion::setRPY
        double hy = y2 * 0.5, hp = p2 * 0.5, hr = r2 * 0.5;
        double cY = std::cos(hy), sY = std::sin(hy), cP = std::cos(hp), sP = std::sin(hp), cR = std::cos(hr),
               sR = std::sin(hr);
        eqx = sR * cP * cY - cR * sP * sY;
        eqy = cR * sP * cY + sR * cP * sY;
        eqz = cR * cP * sY - sR * sP * cY;
        eqw = cR * cP * cY + sR * sP * sY;
    }
    tf_quat_to_rpy(eqx, eqy, eqz, eqw, &roll, &pitch, &yaw);
    Aff3f end = pcl_getTransformation((float)ex, (float)ey, (float)ez, (float)roll, (float)pitch, (float)yaw);

    // affine_trans_begin.inverse() * affine_trans_end (Eigen Affine3f: linear().inverse(), -linv * t), float
    float L[3][3], Linv[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) L[i][j] = begin.m[i][j];
    auto cof = [&](int i, int j) {
        int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return L[i1][j1] * L[i2][j2] - L[i1][j2] * L[i2][j1];
    };
    float det = (cof(0, 0) * L[0][0] + cof(1, 0) * L[1][0]) + cof(2, 0) * L[2][0];
    float invdet = 1.0f / det;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Linv[i][j] = cof(j, i) * invdet;
    float tinv[3];
    for (int i = 0; i < 3; ++i)
        tinv[i] = -((Linv[i][0] * begin.m[0][3] + Linv[i][1] * begin.m[1][3]) + Linv[i][2] * begin.m[2][3]);
    // translation of the product = Linv * t_end + tinv
    float bt[3];
    for (int i = 0; i < 3; ++i)
        bt[i] = ((Linv[i][0] * end.m[0][3] + Linv[i][1] * end.m[1][3]) + Linv[i][2] * end.m[2][3]) + tinv[i] * 1.0f;

    // InterpolateTfWithTime (lf.hpp:219-241): only the translation reaches the deskew
    double dt_scan = scan_end - scan_cur;
    double dt_trans = end_stamp - so[0];
    if (dt_trans == 0.0) {
        incre_xyz[0] = incre_xyz[1] = incre_xyz[2] = 0.0f;
        return 1;
    }
    double ratio = dt_scan / dt_trans;
    float fr = (float)ratio; // Vector3f * double: the scalar is converted to the vector's Scalar
    incre_xyz[0] = bt[0] * fr;
    incre_xyz[1] = bt[1] * fr;
    incre_xyz[2] = bt[2] * fr;
    (void)aff_to_m4;
    return 1;
}


// ------------------------------------------------------------------------------------------------
// caller glue around RunRegister (pcm.cpp:451-465, 933-1101; pcm.hpp:222-290; lf.hpp:219-241)
// ------------------------------------------------------------------------------------------------
namespace {
struct Q4f {
    float w, x, y, z;
};
struct A3f { // Eigen::Affine3f as 4x4 float, row-major
    float m[4][4];
};
A3f a3_identity() {
    A3f a{};
    for (int i = 0; i < 4; ++i) a.m[i][i] = 1.0f;
    return a;
}
void q_to_rot(const Q4f& q, A3f& a) { // Quaternionf::toRotationMatrix into the linear block
    const float tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;
    const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    a.m[0][0] = 1.0f - (tyy + tzz); a.m[0][1] = txy - twz; a.m[0][2] = txz + twy;
    a.m[1][0] = txy + twz; a.m[1][1] = 1.0f - (txx + tzz); a.m[1][2] = tyz - twx;
    a.m[2][0] = txz - twy; a.m[2][1] = tyz + twx; a.m[2][2] = 1.0f - (txx + tyy);
}
Q4f rot_to_q(const A3f& a) { // Quaternionf(Matrix3f)
    Q4f q;
    float t = a.m[0][0] + a.m[1][1] + a.m[2][2];
    if (t > 0.0f) {
        t = std::sqrt(t + 1.0f);
        q.w = 0.5f * t;
        t = 0.5f / t;
        q.x = (a.m[2][1] - a.m[1][2]) * t;
        q.y = (a.m[0][2] - a.m[2][0]) * t;
        q.z = (a.m[1][0] - a.m[0][1]) * t;
    } else {
        int i = 0;
        if (a.m[1][1] > a.m[0][0]) i = 1;
        if (a.m[2][2] > a.m[i][i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(a.m[i][i] - a.m[j][j] - a.m[k][k] + 1.0f);
        float v[3];
        v[i] = 0.5f * t;
        t = 0.5f / t;
        q.w = (a.m[k][j] - a.m[j][k]) * t;
        v[j] = (a.m[j][i] + a.m[i][j]) * t;
        v[k] = (a.m[k][i] + a.m[i][k]) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}
A3f a3_inverse(const A3f& a) {
    A3f r = a3_identity();
    auto cof = [&](int i, int j) {
        int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return a.m[i1][j1] * a.m[i2][j2] - a.m[i1][j2] * a.m[i2][j1];
    };
    float det = (cof(0, 0) * a.m[0][0] + cof(1, 0) * a.m[1][0]) + cof(2, 0) * a.m[2][0];
    float invdet = 1.0f / det;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = cof(j, i) * invdet;
    for (int i = 0; i < 3; ++i)
        r.m[i][3] = -((r.m[i][0] * a.m[0][3] + r.m[i][1] * a.m[1][3]) + r.m[i][2] * a.m[2][3]);
    return r;
}
A3f a3_mul(const A3f& a, const A3f& b) {
    A3f r = a3_identity();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = ((a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j]) + a.m[i][3] * b.m[3][j];
    return r;
}
// InterpolateTfWithTime (lf.hpp:219-241)
A3f InterpolateTfWithTime(const A3f& between, double dt_scan, double dt_trans) {
    if (dt_trans == 0.0) return a3_identity();
    double ratio = dt_scan / dt_trans;
    float fr = static_cast<float>(ratio);
    A3f out = a3_identity();
    Q4f rot = rot_to_q(between);
    const float one = 1.0f - std::numeric_limits<float>::epsilon();
    float d = ((0.0f * rot.x + 0.0f * rot.y) + 0.0f * rot.z) + 1.0f * rot.w; // Identity().dot(rotation)
    float absD = std::fabs(d);
    float scale0, scale1;
    if (absD >= one) {
        scale0 = 1.0f - fr;
        scale1 = fr;
    } else {
        float theta = std::acos(absD);
        float sinTheta = std::sin(theta);
        scale0 = std::sin((1.0f - fr) * theta) / sinTheta;
        scale1 = std::sin(fr * theta) / sinTheta;
    }
    if (d < 0.0f) scale1 = -scale1;
    Q4f qi{scale0 * 1.0f + scale1 * rot.w, scale0 * 0.0f + scale1 * rot.x, scale0 * 0.0f + scale1 * rot.y,
           scale0 * 0.0f + scale1 * rot.z};
    for (int i = 0; i < 3; ++i) out.m[i][3] = between.m[i][3] * fr;
    q_to_rot(qi, out);
    return out;
}
} // namespace

size_t orc_filter_points_by_distance(const float* xyz, size_t n, double max_dist, int64_t* keep_idx) { // pcm.cpp:451-465
    size_t k = 0;
    for (size_t i = 0; i < n; ++i) {
        float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
        double distance = std::sqrt(px * px + py * py + pz * pz);
        if (!(distance > max_dist)) keep_idx[k++] = (int64_t)i;
    }
    return k;
}

int orc_get_interpolated_pose(const double* od, size_t n_odom, double d_cur_time, float T_out[16]) { // pcm.cpp:933-1045
    bool b_found_before = false, b_found_after = false;
    double before[14], after[14];
    std::memset(before, 0, sizeof(before));
    std::memset(after, 0, sizeof(after));
    for (size_t i = 0; i < n_odom; ++i) {
        if (od[14 * i] <= d_cur_time) {
            std::memcpy(before, od + 14 * i, sizeof(before));
            b_found_before = true;
        }
        if (od[14 * i] > d_cur_time) {
            std::memcpy(after, od + 14 * i, sizeof(after));
            b_found_after = true;
            break;
        }
    }
    if (!b_found_before) return 0;
    if (!b_found_after) {
        const double* latest = od + 14 * (n_odom - 1);
        double dt = d_cur_time - latest[0];
        double roll, pitch, yaw;
        tf_quat_to_rpy(latest[4], latest[5], latest[6], latest[7], &roll, &pitch, &yaw);
        double cy = std::cos(yaw), sy = std::sin(yaw), cp = std::cos(pitch), sp = std::sin(pitch), cr = std::cos(roll),
               sr = std::sin(roll);
        double R[3][3] = {{cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr},
                          {sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr},
                          {-sp, cp * sr, cp * cr}};
        for (int i = 0; i < 3; ++i) after[1 + i] = latest[1 + i] + (R[i][0] * latest[8] + R[i][1] * latest[9] + R[i][2] * latest[10]) * dt;
        roll += latest[11] * dt;
        pitch += latest[12] * dt;
        yaw += latest[13] * dt;
        double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
        double cY = std::cos(hy), sY = std::sin(hy), cP = std::cos(hp), sP = std::sin(hp), cR = std::cos(hr), sR = std::sin(hr);
        after[4] = sR * cP * cY - cR * sP * sY;
        after[5] = cR * sP * cY + sR * cP * sY;
        after[6] = cR * cP * sY - sR * sP * cY;
        after[7] = cR * cP * cY + sR * sP * sY;
        after[0] = 0.0; // header.stamp of odom_after is left default on this branch
    }
    double dt_scan = d_cur_time - before[0];
    double dt_trans = after[0] - before[0];
    A3f pose_before = a3_identity(), pose_after = a3_identity();
    for (int i = 0; i < 3; ++i) {
        pose_before.m[i][3] = static_cast<float>(before[1 + i]);
        pose_after.m[i][3] = static_cast<float>(after[1 + i]);
    }
    q_to_rot(Q4f{(float)before[7], (float)before[4], (float)before[5], (float)before[6]}, pose_before);
    q_to_rot(Q4f{(float)after[7], (float)after[4], (float)after[5], (float)after[6]}, pose_after);
    A3f between = a3_mul(a3_inverse(pose_before), pose_after);
    A3f interp = InterpolateTfWithTime(between, dt_scan, dt_trans);
    A3f res = a3_mul(pose_before, interp);
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) T_out[c * 4 + r] = res.m[r][c];
    return 1;
}

void orc_shape_odom_covariance(const double local_cov[36], const double pose[16], double icp_pose_std_m, double cov_out[36]) {
    // PublishPcmOdom (pcm.cpp:1082-1098), NormalizeCovariance (pcm.hpp:248-268), UpdateCovarianceField (pcm.hpp:270-290)
    double d_icp_pose_std_m = std::max(icp_pose_std_m, 0.25);
    M3 R, Ctt, Crr;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            R(r, c) = pose[c * 4 + r];
            Ctt(r, c) = local_cov[c * 6 + r];
            Crr(r, c) = local_cov[(c + 3) * 6 + (r + 3)];
        }
    M3 translation_covariance = m3_mul(m3_mul(R, Ctt), m3_transpose(R));
    auto normalize = [](const M3& in) {
        M3 c = in;
        double min_diag = std::min({c(0, 0), c(1, 1), c(2, 2)});
        const double min_threshold = 1e-9;
        if (min_diag <= min_threshold) {
            for (int i = 0; i < 9; ++i) c.m[i] *= 1e9;
            min_diag = std::min({c(0, 0), c(1, 1), c(2, 2)});
            if (min_diag < min_threshold) min_diag = min_threshold;
        }
        M3 o;
        for (int i = 0; i < 9; ++i) o.m[i] = std::min(c.m[i] / min_diag, 5.0);
        return o;
    };
    double angle_std = d_icp_pose_std_m * M_PI / 180.0;
    M3 tn = normalize(translation_covariance), rn = normalize(Crr);
    std::memset(cov_out, 0, 36 * sizeof(double));
    for (int row = 0; row < 3; ++row)
        for (int col = 0; col < 3; ++col) {
            cov_out[row * 6 + col] = tn(row, col) * d_icp_pose_std_m * d_icp_pose_std_m;
            cov_out[(row + 3) * 6 + (col + 3)] = rn(row, col) * angle_std * angle_std;
        }
}

// ---- exported helpers --------------------------------------------------------------------------
void orc_ldlt_solve6(const double A[36], const double b[6], double x[6]) {
    M6 a;
    std::memcpy(a.m, A, sizeof(a.m));
    ldlt_solve6(a, b, x);
}
void orc_inverse6(const double A[36], double Ainv[36]) {
    M6 a;
    std::memcpy(a.m, A, sizeof(a.m));
    M6 r = m6_inverse(a);
    std::memcpy(Ainv, r.m, sizeof(r.m));
}
void orc_jacobi_svd3(const double A[9], double U[9], double S[3], double V[9]) {
    M3 a, u, v;
    std::memcpy(a.m, A, sizeof(a.m));
    jacobi_svd3(a, u, S, v);
    std::memcpy(U, u.m, sizeof(u.m));
    std::memcpy(V, v.m, sizeof(v.m));
}
void orc_smallest_eigenvector(const double C[9], double n[3]) {
    M3 c;
    std::memcpy(c.m, C, sizeof(c.m));
    const V3 v = smallest_eigenvector(c);
    n[0] = v.x; n[1] = v.y; n[2] = v.z;
}
void orc_angle_axis_to_matrix(const double rotvec[3], double R[9]) {
    M3 r = rotvec_to_matrix(V3{rotvec[0], rotvec[1], rotvec[2]});
    std::memcpy(R, r.m, sizeof(r.m));
}
double orc_matrix_to_angle(const double R[9]) {
    M3 r;
    std::memcpy(r.m, R, sizeof(r.m));
    return matrix_to_angle(r);
}

} // extern "C"
