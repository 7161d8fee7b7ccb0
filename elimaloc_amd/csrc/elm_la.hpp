// elm_la.hpp -- small fixed-size fp64 linear algebra used by the HIP kernels and the host side of the product.
// Row-major arrays.  Written for the product; the test oracle under oracle/ has its own, separate kit.
#pragma once
#include <math.h>

#if defined(__HIPCC__) && defined(__forceinline__) // (a host-only translation unit that never included the HIP runtime header takes the plain form)
#define ELM_HD __host__ __device__ __forceinline__
#else
#define ELM_HD inline
#endif

namespace elm {

// sinf / cosf as glibc >= 2.28 computes them on x86-64 (sysdeps/ieee754/flt-32/s_sincosf.h, the ARM optimized-routines algorithm, in its
// FMA-contracted multiarch variant that every FMA-capable CPU selects): float64 polynomial on the reduced argument, one rounding to
// float32.  pcl::getTransformation calls std::cos / std::sin on floats (pcm.cpp:806), so the deskewed points depend on these exact results.
// Restated from the published routine; tests/test_sincosf.py compiles this header for the host and sweeps it against the C library's sinf / cosf
// (320 M arguments in (-120, 120) were checked once: bit for bit).
// (|x| >= 120 takes the correctly rounded float64 function: a deskew rotation never gets there.)
ELM_HD float glibc_sincos_poly(double x, double x2, bool flip, int n) {
    const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
    const double sg = flip ? -1.0 : 1.0; // the second table negates the cosine coefficients
    const double c0 = sg * 0x1p0, c1c = sg * -0x1.ffffffd0c621cp-2, c2c = sg * 0x1.55553e1068f19p-5, c3c = sg * -0x1.6c087e89a359dp-10,
                 c4c = sg * 0x1.99343027bf8c3p-16;
    if ((n & 1) == 0) {
        const double x3 = x * x2, s1 = __builtin_fma(x2, s3c, s2c), x7 = x3 * x2, s = __builtin_fma(x3, s1c, x);
        return (float)__builtin_fma(x7, s1, s);
    }
    const double x4 = x2 * x2, c2 = __builtin_fma(x2, c4c, c3c), c1 = __builtin_fma(x2, c1c, c0), x6 = x4 * x2, c = __builtin_fma(x4, c2c, c1);
    return (float)__builtin_fma(x6, c2, c);
}
ELM_HD unsigned glibc_abstop12(float f) {
    unsigned u;
    __builtin_memcpy(&u, &f, 4);
    return (u >> 20) & 0x7ffu;
}
template <bool COS>
ELM_HD float glibc_sincosf(float y) {
    const unsigned top = glibc_abstop12(y);
    double x = (double)y;
    if (top < glibc_abstop12(0x1.921FB6p-1f)) { // |y| < pi / 4
        if (top < glibc_abstop12(0x1p-12f)) return COS ? 1.0f : y;
        return glibc_sincos_poly(x, x * x, false, COS ? 1 : 0);
    }
    if (top < glibc_abstop12(120.0f)) {
        const double r = x * 0x1.45F306DC9C883p+23;
        const int n = ((int)r + 0x800000) >> 24;
        x = __builtin_fma(-(double)n, 0x1.921FB54442D18p0, x);
        const int q = COS ? n + 1 : n;
        const double sgn = ((q & 3) == 1 || (q & 3) == 2) ? -1.0 : 1.0; // sign[q & 3] = {1, -1, -1, 1}
        return glibc_sincos_poly(x * sgn, x * x, (q & 2) != 0, COS ? (n ^ 1) : n);
    }
    return COS ? (float)cos(x) : (float)sin(x);
}

// ---- 3x3 ------------------------------------------------------------------------------------------
// inverse by cofactors times 1/det (the form Eigen uses for Matrix3d::inverse(), reg.cpp:79,113)
ELM_HD void inv3(const double a[9], double r[9]) {
    double c00 = a[4] * a[8] - a[5] * a[7];
    double c01 = a[5] * a[6] - a[3] * a[8];
    double c02 = a[3] * a[7] - a[4] * a[6];
    double det = (c00 * a[0] + c01 * a[1]) + c02 * a[2];
    double id = 1.0 / det;
    r[0] = c00 * id;
    r[3] = c01 * id;
    r[6] = c02 * id;
    r[1] = (a[2] * a[7] - a[1] * a[8]) * id;
    r[4] = (a[0] * a[8] - a[2] * a[6]) * id;
    r[7] = (a[1] * a[6] - a[0] * a[7]) * id;
    r[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    r[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    r[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}
ELM_HD void mul3(const double a[9], const double b[9], double r[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r[i * 3 + j] = (a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j]) + a[i * 3 + 2] * b[6 + j];
}
// r = a * b^T
ELM_HD void mul3_bt(const double a[9], const double b[9], double r[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r[i * 3 + j] = (a[i * 3] * b[j * 3] + a[i * 3 + 1] * b[j * 3 + 1]) + a[i * 3 + 2] * b[j * 3 + 2];
}

// Cyclic Jacobi eigen-decomposition of a symmetric 3x3 (row-major a, overwritten by its diagonal form).
// Returns eigenvalues w[3] sorted DESCENDING and the matching eigenvectors as COLUMNS of v (row-major).
ELM_HD void eig3_sym_desc(const double ain[9], double w[3], double v[9]) {
    double a00 = ain[0], a01 = ain[1], a02 = ain[2], a11 = ain[4], a12 = ain[5], a22 = ain[8];
    double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
    for (int sweep = 0; sweep < 24; ++sweep) {
        double off = fabs(a01) + fabs(a02) + fabs(a12);
        double dia = fabs(a00) + fabs(a11) + fabs(a22);
        if (off <= 1e-300 || off <= 1e-22 * dia) break;
        // (0,1)
        if (a01 != 0.0) {
            double th = (a11 - a00) / (2.0 * a01);
            double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            double n00 = a00 - t * a01, n11 = a11 + t * a01;
            double n02 = c * a02 - s * a12, n12 = s * a02 + c * a12;
            a00 = n00; a11 = n11; a01 = 0.0; a02 = n02; a12 = n12;
            double x;
            x = c * v00 - s * v01; v01 = s * v00 + c * v01; v00 = x;
            x = c * v10 - s * v11; v11 = s * v10 + c * v11; v10 = x;
            x = c * v20 - s * v21; v21 = s * v20 + c * v21; v20 = x;
        }
        // (0,2)
        if (a02 != 0.0) {
            double th = (a22 - a00) / (2.0 * a02);
            double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            double n00 = a00 - t * a02, n22 = a22 + t * a02;
            double n01 = c * a01 - s * a12, n12 = s * a01 + c * a12;
            a00 = n00; a22 = n22; a02 = 0.0; a01 = n01; a12 = n12;
            double x;
            x = c * v00 - s * v02; v02 = s * v00 + c * v02; v00 = x;
            x = c * v10 - s * v12; v12 = s * v10 + c * v12; v10 = x;
            x = c * v20 - s * v22; v22 = s * v20 + c * v22; v20 = x;
        }
        // (1,2)
        if (a12 != 0.0) {
            double th = (a22 - a11) / (2.0 * a12);
            double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            double n11 = a11 - t * a12, n22 = a22 + t * a12;
            double n01 = c * a01 - s * a02, n02 = s * a01 + c * a02;
            a11 = n11; a22 = n22; a12 = 0.0; a01 = n01; a02 = n02;
            double x;
            x = c * v01 - s * v02; v02 = s * v01 + c * v02; v01 = x;
            x = c * v11 - s * v12; v12 = s * v11 + c * v12; v11 = x;
            x = c * v21 - s * v22; v22 = s * v21 + c * v22; v21 = x;
        }
    }
    double e0 = a00, e1 = a11, e2 = a22;
    // sort descending (3-element network), swapping eigenvector columns
#define ELM_SWAPCOL(A0, A1, A2, B0, B1, B2) { double q; q = A0; A0 = B0; B0 = q; q = A1; A1 = B1; B1 = q; q = A2; A2 = B2; B2 = q; }
    if (e0 < e1) { double q = e0; e0 = e1; e1 = q; ELM_SWAPCOL(v00, v10, v20, v01, v11, v21) }
    if (e1 < e2) { double q = e1; e1 = e2; e2 = q; ELM_SWAPCOL(v01, v11, v21, v02, v12, v22) }
    if (e0 < e1) { double q = e0; e0 = e1; e1 = q; ELM_SWAPCOL(v00, v10, v20, v01, v11, v21) }
#undef ELM_SWAPCOL
    w[0] = e0; w[1] = e1; w[2] = e2;
    v[0] = v00; v[1] = v01; v[2] = v02; v[3] = v10; v[4] = v11; v[5] = v12; v[6] = v20; v[7] = v21; v[8] = v22;
}

// Two-sided Jacobi SVD of a real 3x3 (row-major), the algorithm of Eigen's JacobiSVD<Matrix3d> (sweeps over
// (p,q) = (1,0),(2,0),(2,1); each step diagonalises the 2x2 sub-block with a left and a right rotation; signs moved
// into U; singular values sorted descending).  Rank-deficient inputs (a map point with one or two neighbours) leave
// the signs of the null-space columns of U and V to round-off, and the reference's covariances inherit that, so the
// same operation order is followed here instead of an eigen-decomposition shortcut.
struct Rot2 { double c, s; };
ELM_HD Rot2 make_jacobi2(double x, double y, double z) { // JacobiRotation::makeJacobi for [[x y][y z]]
    Rot2 j;
    const double deno = 2.0 * fabs(y);
    if (deno < 2.2250738585072014e-308) { j.c = 1.0; j.s = 0.0; return j; }
    const double tau = (x - z) / deno;
    const double w = sqrt(tau * tau + 1.0);
    const double t = (tau > 0.0) ? 1.0 / (tau + w) : 1.0 / (tau - w);
    const double sign_t = t > 0.0 ? 1.0 : -1.0;
    const double n = 1.0 / sqrt(t * t + 1.0);
    j.s = -sign_t * (y / fabs(y)) * fabs(t) * n;
    j.c = n;
    return j;
}
// rows p,q of m: x' = c x + s y ; y' = -s x + c y
ELM_HD void rot_rows(double m[9], int p, int q, double c, double s) {
    for (int k = 0; k < 3; ++k) {
        const double xi = m[p * 3 + k], yi = m[q * 3 + k];
        m[p * 3 + k] = c * xi + s * yi;
        m[q * 3 + k] = -s * xi + c * yi;
    }
}
// columns p,q of m: x' = c x - s y ; y' = s x + c y
ELM_HD void rot_cols(double m[9], int p, int q, double c, double s) {
    for (int k = 0; k < 3; ++k) {
        const double xi = m[k * 3 + p], yi = m[k * 3 + q];
        m[k * 3 + p] = c * xi - s * yi;
        m[k * 3 + q] = s * xi + c * yi;
    }
}
ELM_HD void svd3_jacobi(const double A[9], double U[9], double S[3], double V[9]) {
    const double tiny = 2.2250738585072014e-308;    // numeric_limits<double>::min()
    const double precision = 2.0 * 2.220446049250313e-16;
    double scale = 0.0;
    for (int i = 0; i < 9; ++i) scale = fmax(scale, fabs(A[i]));
    if (scale == 0.0) scale = 1.0;
    double W[9];
    for (int i = 0; i < 9; ++i) W[i] = A[i] / scale;
    for (int i = 0; i < 9; ++i) { U[i] = (i % 4 == 0) ? 1.0 : 0.0; V[i] = U[i]; }
    double maxDiag = fmax(fabs(W[0]), fmax(fabs(W[4]), fabs(W[8])));
    bool finished = false;
    for (int guard = 0; !finished && guard < 1000; ++guard) {
        finished = true;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                const double threshold = fmax(tiny, precision * maxDiag);
                if (fabs(W[p * 3 + q]) > threshold || fabs(W[q * 3 + p]) > threshold) {
                    finished = false;
                    // real_2x2_jacobi_svd on [[W(p,p) W(p,q)][W(q,p) W(q,q)]]
                    const double m00 = W[p * 3 + p], m01 = W[p * 3 + q], m10 = W[q * 3 + p], m11 = W[q * 3 + q];
                    double r1c, r1s;
                    const double t = m00 + m11, d = m10 - m01;
                    if (fabs(d) < tiny) { r1s = 0.0; r1c = 1.0; }
                    else {
                        const double u = t / d;
                        const double tmp = sqrt(1.0 + u * u);
                        r1s = 1.0 / tmp;
                        r1c = u / tmp;
                    }
                    const double n00 = r1c * m00 + r1s * m10, n01 = r1c * m01 + r1s * m11, n11 = -r1s * m01 + r1c * m11;
                    const Rot2 jr = make_jacobi2(n00, n01, n11);
                    // j_left = rot1 * j_right^T
                    const double jlc = r1c * jr.c - r1s * (-jr.s);
                    const double jls = r1c * (-jr.s) + r1s * jr.c;
                    rot_rows(W, p, q, jlc, jls);       // W.applyOnTheLeft(p,q,j_left)
                    rot_cols(U, p, q, jlc, -jls);      // U.applyOnTheRight(p,q,j_left.transpose())
                    rot_cols(W, p, q, jr.c, jr.s);     // W.applyOnTheRight(p,q,j_right)
                    rot_cols(V, p, q, jr.c, jr.s);     // V.applyOnTheRight(p,q,j_right)
                    maxDiag = fmax(maxDiag, fmax(fabs(W[p * 3 + p]), fabs(W[q * 3 + q])));
                }
            }
    }
    for (int i = 0; i < 3; ++i) {
        const double a = fabs(W[i * 4]);
        S[i] = a;
        if (a != 0.0) {
            const double sg = W[i * 4] / a;
            for (int r = 0; r < 3; ++r) U[r * 3 + i] *= sg;
        }
    }
    for (int i = 0; i < 3; ++i) S[i] *= scale;
    for (int i = 0; i < 3; ++i) { // selection sort, descending, columns of U and V follow
        int pos = i;
        double big = S[i];
        for (int k = i + 1; k < 3; ++k)
            if (S[k] > big) { big = S[k]; pos = k; }
        if (big == 0.0) break;
        if (pos != i) {
            double q = S[i]; S[i] = S[pos]; S[pos] = q;
            for (int r = 0; r < 3; ++r) {
                q = U[r * 3 + i]; U[r * 3 + i] = U[r * 3 + pos]; U[r * 3 + pos] = q;
                q = V[r * 3 + i]; V[r * 3 + i] = V[r * 3 + pos]; V[r * 3 + pos] = q;
            }
        }
    }
}

// "Plane regularization" of a sample covariance (vhm.hpp:140-145, 240-245): U diag(1,1,1e-3) V^T of its SVD.
// cov_out row-major.  normal = eigenvector of the smallest eigenvalue of the REGULARISED matrix, which is what
// reg.cpp:89-91 extracts from it (only used through |r . n| in the GICP fitness term).
ELM_HD void plane_regularize(const double cov[9], double cov_out[9], double normal[3]) {
    double U[9], S[3], V[9];
    svd3_jacobi(cov, U, S, V);
    const double s[3] = {1.0, 1.0, 1e-3};
    double UD[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) UD[i * 3 + j] = U[i * 3 + j] * s[j];
    mul3_bt(UD, V, cov_out); // (U diag) V^T
    double w[3], ev[9];
    double sym[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) sym[i * 3 + j] = (j <= i) ? cov_out[i * 3 + j] : cov_out[j * 3 + i]; // lower triangle, as Eigen reads it
    eig3_sym_desc(sym, w, ev);
    normal[0] = ev[2];
    normal[1] = ev[5];
    normal[2] = ev[8];
}

// Registration::CalPointCov (reg.hpp:186-209): the "covariance" the reference attaches to a source point under use_radar_cov -- R S with
// R = AngleAxisd(azimuth, UnitZ) * AngleAxisd(elevation, UnitY) of the point's position (quaternion product, then
// Quaternion::toRotationMatrix) and S = diag(range spread, max(0.1, d sin(azimuth spread)), max(0.1, d sin(elevation spread))), d the
// horizontal range: a product, not R S R^T -- the matrix is not symmetric.  Row-major.
ELM_HD void radar_point_cov(double gx, double gy, double gz, double range_var_m, double azim_var_deg, double ele_var_deg, double* Cs) {
    const double kPi = 3.14159265358979323846;
    const double dist = sqrt(gx * gx + gy * gy);
    const double s_x = range_var_m;
    const double s_y = fmax(0.1, dist * sin(azim_var_deg / 180 * kPi));
    const double s_z = fmax(0.1, dist * sin(ele_var_deg / 180 * kPi));
    const double ele = atan2(gz, dist), azi = atan2(gy, gx);
    const double yw = cos(azi / 2.0), yz = sin(azi / 2.0), pw = cos(ele / 2.0), py = sin(ele / 2.0);
    const double w = yw * pw, x = -(yz * py), y = yw * py, z = yz * pw;
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    const double R[9] = {1.0 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.0 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1.0 - (txx + tyy)};
    for (int r = 0; r < 3; ++r) {
        Cs[r * 3 + 0] = R[r * 3 + 0] * s_x;
        Cs[r * 3 + 1] = R[r * 3 + 1] * s_y;
        Cs[r * 3 + 2] = R[r * 3 + 2] * s_z;
    }
}

// ---- 6x6 ------------------------------------------------------------------------------------------
// Solve A x = b, A symmetric (row-major, only the lower triangle is read), by LDL^T with diagonal pivoting
// (largest |diagonal| of the trailing block, symmetric row/column swap) as Eigen's LDLT does (reg.cpp:56).
ELM_HD void ldlt_solve6_ws(const double* A, const double* b, double* x, double* m, double* misc) {
    // m: 36 doubles, misc: >= 24 doubles of caller-provided workspace (LDS on the device)
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) m[i * 6 + j] = (j <= i) ? A[i * 6 + j] : A[j * 6 + i];
    double* d = misc;      // [6]
    double* y = misc + 6;  // [6]
    double* permd = misc + 12; // [6] permutation kept as doubles in the same workspace
    for (int i = 0; i < 6; ++i) permd[i] = (double)i;
    for (int k = 0; k < 6; ++k) {
        // pivot = largest |diagonal| of the trailing Schur complement
        int p = k;
        double best = -1.0;
        for (int i = k; i < 6; ++i) {
            double di = m[i * 6 + i];
            for (int c = 0; c < k; ++c) di -= m[i * 6 + c] * m[i * 6 + c] * d[c];
            double c2 = fabs(di);
            if (c2 > best) { best = c2; p = i; }
        }
        if (p != k) { // symmetric swap of rows/cols k and p on the full matrix
            for (int j = 0; j < 6; ++j) { double q = m[k * 6 + j]; m[k * 6 + j] = m[p * 6 + j]; m[p * 6 + j] = q; }
            for (int i = 0; i < 6; ++i) { double q = m[i * 6 + k]; m[i * 6 + k] = m[i * 6 + p]; m[i * 6 + p] = q; }
            double qd = permd[k]; permd[k] = permd[p]; permd[p] = qd;
        }
        // m[k][k] -= sum_{c<k} L[k][c]^2 d[c] ; L[i][k] = (m[i][k] - sum_c L[i][c] d[c] L[k][c]) / d[k]
        double dk = m[k * 6 + k];
        for (int c = 0; c < k; ++c) dk -= m[k * 6 + c] * m[k * 6 + c] * d[c];
        d[k] = dk;
        for (int i = k + 1; i < 6; ++i) {
            double s = m[i * 6 + k];
            for (int c = 0; c < k; ++c) s -= m[i * 6 + c] * d[c] * m[k * 6 + c];
            m[i * 6 + k] = (dk != 0.0) ? s / dk : s;
        }
    }
    for (int i = 0; i < 6; ++i) y[i] = b[(int)permd[i]];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < i; ++j) y[i] -= m[i * 6 + j] * y[j];
    for (int i = 0; i < 6; ++i) y[i] = (fabs(d[i]) > 5.6e-309) ? y[i] / d[i] : 0.0;
    for (int i = 5; i >= 0; --i)
        for (int j = i + 1; j < 6; ++j) y[i] -= m[j * 6 + i] * y[j];
    for (int i = 0; i < 6; ++i) x[(int)permd[i]] = y[i];
}

ELM_HD void ldlt_solve6(const double A[36], const double b[6], double x[6]) {
    double m[36], misc[24];
    ldlt_solve6_ws(A, b, x, m, misc);
}

// General 6x6 inverse by Gauss-Jordan with partial pivoting (reg.cpp:141 uses PartialPivLU).
ELM_HD void inv6_ws(const double* A, double* R, double* m) {
    for (int i = 0; i < 36; ++i) m[i] = A[i];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) R[i * 6 + j] = (i == j) ? 1.0 : 0.0;
    for (int k = 0; k < 6; ++k) {
        int p = k;
        double best = fabs(m[k * 6 + k]);
        for (int i = k + 1; i < 6; ++i) {
            double c = fabs(m[i * 6 + k]);
            if (c > best) { best = c; p = i; }
        }
        if (p != k)
            for (int j = 0; j < 6; ++j) {
                double q = m[k * 6 + j]; m[k * 6 + j] = m[p * 6 + j]; m[p * 6 + j] = q;
                q = R[k * 6 + j]; R[k * 6 + j] = R[p * 6 + j]; R[p * 6 + j] = q;
            }
        double piv = 1.0 / m[k * 6 + k];
        for (int j = 0; j < 6; ++j) { m[k * 6 + j] *= piv; R[k * 6 + j] *= piv; }
        for (int i = 0; i < 6; ++i) {
            if (i == k) continue;
            double f = m[i * 6 + k];
            if (f == 0.0) continue;
            for (int j = 0; j < 6; ++j) { m[i * 6 + j] -= f * m[k * 6 + j]; R[i * 6 + j] -= f * R[k * 6 + j]; }
        }
    }
}

ELM_HD void inv6(const double A[36], double R[36]) {
    double m[36];
    inv6_ws(A, R, m);
}

// ---- SO(3) ----------------------------------------------------------------------------------------
// AngleAxisd(|v|, v/|v|).toRotationMatrix() (reg.cpp:58-61), Rodrigues; zero vector -> identity. Row-major.
ELM_HD void rotvec_to_matrix(const double v[3], double R[9]) {
    double n2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    double ang = sqrt(n2);
    double ax = v[0], ay = v[1], az = v[2];
    if (n2 > 0.0) { ax /= ang; ay /= ang; az /= ang; }
    double s = sin(ang), c = cos(ang), c1 = 1.0 - c;
    double sx = s * ax, sy = s * ay, sz = s * az;
    double cx = c1 * ax, cy = c1 * ay, cz = c1 * az;
    double t;
    t = cx * ay; R[1] = t - sz; R[3] = t + sz;
    t = cx * az; R[2] = t + sy; R[6] = t - sy;
    t = cy * az; R[5] = t - sx; R[7] = t + sx;
    R[0] = cx * ax + c; R[4] = cy * ay + c; R[8] = cz * az + c;
}
// AngleAxisd(Matrix3d).angle() (reg.cpp:381-382): rotation matrix -> unit quaternion -> 2 atan2(|vec|, |w|)
ELM_HD double matrix_to_angle(const double R[9]) {
    double tr = R[0] + R[4] + R[8];
    double qw, qx, qy, qz;
    if (tr > 0.0) {
        double t = sqrt(tr + 1.0);
        qw = 0.5 * t; t = 0.5 / t;
        qx = (R[7] - R[5]) * t; qy = (R[2] - R[6]) * t; qz = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 4]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        double t = sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
        double q[3];
        q[i] = 0.5 * t; t = 0.5 / t;
        qw = (R[k * 3 + j] - R[j * 3 + k]) * t;
        q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
        qx = q[0]; qy = q[1]; qz = q[2];
    }
    double n = sqrt((qx * qx + qy * qy) + qz * qz);
    if (n == 0.0) return 0.0;
    return 2.0 * atan2(n, fabs(qw));
}

} // namespace elm
