// elm_glue.cpp -- host-side caller glue around the registration path: the steps of
// PcmMatching::CallbackPointCloud either side of RunRegister, kept in the float32 / float64 arithmetic the reference uses.
// Plain C++ (no device code); exported through the same C ABI.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "elm_la.hpp"

#include "../../include/elimaloc_hip.h"
#include "elm_hostapi.hpp"

namespace {

struct Aff3f { // Eigen::Affine3f: 3x3 linear part (row-major here) + translation
    float l[9];
    float t[3];
};
Aff3f aff_identity() {
    Aff3f a;
    for (int i = 0; i < 9; ++i) a.l[i] = (i % 4 == 0) ? 1.f : 0.f;
    a.t[0] = a.t[1] = a.t[2] = 0.f;
    return a;
}
// Eigen::Quaternionf::toRotationMatrix()
void quat_to_matrix(float w, float x, float y, float z, float r[9]) {
    const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    r[0] = 1.f - (tyy + tzz); r[1] = txy - twz; r[2] = txz + twy;
    r[3] = txy + twz; r[4] = 1.f - (txx + tzz); r[5] = tyz - twx;
    r[6] = txz - twy; r[7] = tyz + twx; r[8] = 1.f - (txx + tyy);
}
// Eigen::Quaternionf(Matrix3f)
void matrix_to_quat(const float m[9], float* w, float* x, float* y, float* z) {
    float t = m[0] + m[4] + m[8];
    if (t > 0.f) {
        t = sqrtf(t + 1.f);
        *w = 0.5f * t;
        t = 0.5f / t;
        *x = (m[7] - m[5]) * t; *y = (m[2] - m[6]) * t; *z = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrtf(m[i * 4] - m[j * 4] - m[k * 4] + 1.f);
        float q[3];
        q[i] = 0.5f * t;
        t = 0.5f / t;
        *w = (m[k * 3 + j] - m[j * 3 + k]) * t;
        q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
        *x = q[0]; *y = q[1]; *z = q[2];
    }
}
Aff3f aff_inverse(const Aff3f& a) { // Transform::inverse(Affine): linear().inverse(), -linv * t
    const float* L = a.l;
    Aff3f r;
    const float c00 = L[4] * L[8] - L[5] * L[7], c01 = L[5] * L[6] - L[3] * L[8], c02 = L[3] * L[7] - L[4] * L[6];
    const float det = (c00 * L[0] + c01 * L[1]) + c02 * L[2];
    const float id = 1.f / det;
    r.l[0] = c00 * id; r.l[3] = c01 * id; r.l[6] = c02 * id;
    r.l[1] = (L[2] * L[7] - L[1] * L[8]) * id; r.l[4] = (L[0] * L[8] - L[2] * L[6]) * id; r.l[7] = (L[1] * L[6] - L[0] * L[7]) * id;
    r.l[2] = (L[1] * L[5] - L[2] * L[4]) * id; r.l[5] = (L[2] * L[3] - L[0] * L[5]) * id; r.l[8] = (L[0] * L[4] - L[1] * L[3]) * id;
    for (int i = 0; i < 3; ++i) r.t[i] = -((r.l[i * 3] * a.t[0] + r.l[i * 3 + 1] * a.t[1]) + r.l[i * 3 + 2] * a.t[2]);
    return r;
}
Aff3f aff_mul(const Aff3f& a, const Aff3f& b) {
    Aff3f r;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) r.l[i * 3 + j] = (a.l[i * 3] * b.l[j] + a.l[i * 3 + 1] * b.l[3 + j]) + a.l[i * 3 + 2] * b.l[6 + j];
        r.t[i] = ((a.l[i * 3] * b.t[0] + a.l[i * 3 + 1] * b.t[1]) + a.l[i * 3 + 2] * b.t[2]) + a.t[i] * 1.f;
    }
    return r;
}
// InterpolateTfWithTime (localization_functions.hpp:219-241)
Aff3f interpolate_tf(const Aff3f& between, double dt_scan, double dt_trans) {
    if (dt_trans == 0.0) return aff_identity();
    const double ratio = dt_scan / dt_trans;
    const float fr = (float)ratio;
    Aff3f out = aff_identity();
    float qw, qx, qy, qz;
    matrix_to_quat(between.l, &qw, &qx, &qy, &qz); // rotation() of an Affine3f of a rigid motion == its linear part
    // Quaternionf::Identity().slerp(ratio, rotation)
    const float one = 1.f - 1.1920929e-07f;
    const float d = ((0.f * qx + 0.f * qy) + 0.f * qz) + 1.f * qw;
    const float absD = fabsf(d);
    float scale0, scale1;
    if (absD >= one) {
        scale0 = 1.f - fr;
        scale1 = fr;
    } else {
        const float theta = acosf(absD), sinTheta = sinf(theta);
        scale0 = sinf((1.f - fr) * theta) / sinTheta;
        scale1 = sinf(fr * theta) / sinTheta;
    }
    if (d < 0.f) scale1 = -scale1;
    const float ix = scale0 * 0.f + scale1 * qx, iy = scale0 * 0.f + scale1 * qy, iz = scale0 * 0.f + scale1 * qz,
                iw = scale0 * 1.f + scale1 * qw;
    for (int i = 0; i < 3; ++i) out.t[i] = between.t[i] * fr; // translate() on an identity transform
    quat_to_matrix(iw, ix, iy, iz, out.l);                    // rotate() on an identity linear part
    return out;
}
void quat_to_rpy(const double q[4], double* roll, double* pitch, double* yaw) { // tf::Matrix3x3(q).getRPY
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double s = 2.0 / (x * x + y * y + z * z + w * w);
    const double xs = x * s, ys = y * s, zs = z * s;
    const double wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
    const double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
    if (fabs(m20) >= 1.0) {
        *yaw = 0.0;
        *roll = atan2(m21, m22);
        *pitch = (m20 < 0) ? M_PI / 2.0 : -M_PI / 2.0;
    } else {
        *pitch = -asin(m20);
        const double cp = cos(*pitch);
        *roll = atan2(m21 / cp, m22 / cp);
        *yaw = atan2(m10 / cp, m00 / cp);
    }
}

struct Key3 {
    int32_t x, y, z;
    bool operator==(const Key3& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct Key3Hash {
    size_t operator()(const Key3& k) const {
        uint64_t h = (uint32_t)k.x * 0x9E3779B97F4A7C15ull;
        h ^= ((uint32_t)k.y + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
        h ^= ((uint32_t)k.z + 0x165667B1ull) * 0x9E3779B185EBCA87ull;
        return (size_t)(h ^ (h >> 29));
    }
};

// NormalizeDiagonalCovariance is computed by the reference (pcm.cpp:1087,1091) but its result is unused; only
// NormalizeCovariance (pcm.hpp:248-268) reaches the message.
void normalize_covariance(const double in[9], double out[9]) {
    double c[9];
    memcpy(c, in, sizeof(c));
    double min_diag = std::min({c[0], c[4], c[8]});
    const double min_threshold = 1e-9;
    if (min_diag <= min_threshold) {
        for (double& v : c) v *= 1e9;
        min_diag = std::min({c[0], c[4], c[8]});
        if (min_diag < min_threshold) min_diag = min_threshold;
    }
    for (int i = 0; i < 9; ++i) out[i] = std::min(c[i] / min_diag, 5.0);
}

} // namespace

extern "C" int elm_filter_points_by_distance(const float* xyz, const float* time, size_t n, double max_dist, float* xyz_out,
                                             float* time_out, size_t* n_out) {
    if (!n_out || (n && (!xyz || !xyz_out))) return ELM_ERR_INVALID;
    // pcm.cpp:456 drops a point when (double)sqrtf(x*x + y*y + z*z) > max_dist (float arithmetic, float sqrt, widened).  sqrtf is
    // monotone, so that is s > s_max with s_max = the largest float whose float root does not exceed max_dist: the same decisions
    // without a root per point.
    float s_max;
    if (!(max_dist >= 0.0)) {
        s_max = -1.0f; // every point is dropped (NaN: every comparison of the reference is false -> handled below)
    } else if (std::isinf(max_dist)) { // `distance > inf` is false for every point
        s_max = __builtin_inff();
    } else {
        s_max = (float)(max_dist * max_dist);
        while (s_max > 0.0f && (double)sqrtf(s_max) > max_dist) s_max = nextafterf(s_max, 0.0f);
        while ((double)sqrtf(nextafterf(s_max, __builtin_inff())) <= max_dist) s_max = nextafterf(s_max, __builtin_inff());
    }
    size_t k = 0;
    if (max_dist != max_dist) { // NaN threshold: `distance > NaN` is false, nothing is dropped
        s_max = __builtin_inff();
    }
    for (size_t i = 0; i < n; ++i) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        const float s2 = x * x + y * y + z * z;
        if (s2 > s_max) continue; // (a NaN coordinate compares false, as in the reference: the point is kept)
        xyz_out[3 * k] = x; xyz_out[3 * k + 1] = y; xyz_out[3 * k + 2] = z;
        if (time && time_out) time_out[k] = time[i];
        ++k;
    }
    *n_out = k;
    return ELM_OK;
}

extern "C" int elm_voxel_downsample(const float* xyz, size_t n, double voxel_size, int64_t* keep_idx, size_t* n_keep) {
    if (!n_keep || !(voxel_size > 0.0) || (n && (!xyz || !keep_idx))) return ELM_ERR_INVALID;
    size_t k = 0;
    // fast path: keys within +-2^20 per axis pack into one 64-bit word -> flat open-addressing set (no node allocations).
    // Scratch is kept per thread (no page faults per scan); table slots are prefetched 16 points ahead.
    static thread_local std::vector<uint64_t> packed, table;
    static thread_local std::vector<uint32_t> slot;
    if (packed.size() < n) { packed.resize(n); slot.resize(n); }
    bool fits = true;
    const int32_t lim = 1 << 20;
    size_t cap = 64;
    int cap_log2 = 6;
    while (cap < 2 * n) { cap <<= 1; ++cap_log2; }
    // floor(x / vs) by truncation + fix-up (no libm call): identical for every finite quotient inside the packing range
    auto fl = [](double q) { const int32_t kk = (int32_t)q; return ((double)kk > q) ? kk - 1 : kk; };
    for (size_t i = 0; i < n; ++i) {
        const double qx = (double)xyz[3 * i] / voxel_size, qy = (double)xyz[3 * i + 1] / voxel_size, qz = (double)xyz[3 * i + 2] / voxel_size;
        if (!(qx > -lim && qx < lim && qy > -lim && qy < lim && qz > -lim && qz < lim)) { fits = false; break; }
        const uint64_t key = ((uint64_t)(fl(qx) + lim) << 42) | ((uint64_t)(fl(qy) + lim) << 21) | (uint64_t)(fl(qz) + lim);
        packed[i] = key;
        slot[i] = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - cap_log2)); // top bits: they depend on every key bit
    }
    if (fits) {
        table.assign(cap, ~0ull);
        for (size_t i = 0; i < n; ++i) {
            if (i + 16 < n) __builtin_prefetch(&table[slot[i + 16]], 1);
            const uint64_t key = packed[i];
            size_t h = slot[i];
            while (table[h] != ~0ull && table[h] != key) h = (h + 1) & (cap - 1);
            if (table[h] == ~0ull) { table[h] = key; keep_idx[k++] = (int64_t)i; } // first point of every voxel, input order
        }
        *n_keep = k;
        return ELM_OK;
    }
    std::unordered_map<Key3, int64_t, Key3Hash> grid;
    grid.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        const Key3 key{(int32_t)floor((double)xyz[3 * i] / voxel_size), (int32_t)floor((double)xyz[3 * i + 1] / voxel_size),
                       (int32_t)floor((double)xyz[3 * i + 2] / voxel_size)};
        if (grid.emplace(key, (int64_t)i).second) keep_idx[k++] = (int64_t)i;
    }
    *n_keep = k;
    return ELM_OK;
}

extern "C" int elm_get_interpolated_pose(const double* odom14, size_t n_odom, double d_cur_time, float T_out[16], int* ok) {
    if (!T_out || !ok || (n_odom && !odom14)) return ELM_ERR_INVALID;
    *ok = 0;
    const double* before = nullptr;
    const double* after = nullptr;
    for (size_t i = 0; i < n_odom; ++i) {
        const double* o = odom14 + 14 * i;
        if (o[0] <= d_cur_time) before = o;
        if (o[0] > d_cur_time) { after = o; break; }
    }
    if (!before) return ELM_OK; // "Pose before not exist"
    double after_buf[14];
    if (!after) { // extrapolate the last odometry with its twist (pcm.cpp:956-1011); the header stamp stays default (0)
        const double* lo = odom14 + 14 * (n_odom - 1);
        const double dt = d_cur_time - lo[0];
        double r2, p2, y2;
        quat_to_rpy(lo + 4, &r2, &p2, &y2);
        const double cy = cos(y2), sy = sin(y2), cp = cos(p2), sp = sin(p2), cr = cos(r2), sr = sin(r2);
        const double R[3][3] = {{cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr},
                                {sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr},
                                {-sp, cp * sr, cp * cr}};
        memset(after_buf, 0, sizeof(after_buf));
        for (int k = 0; k < 3; ++k) after_buf[1 + k] = lo[1 + k] + (R[k][0] * lo[8] + R[k][1] * lo[9] + R[k][2] * lo[10]) * dt;
        r2 += lo[11] * dt; p2 += lo[12] * dt; y2 += lo[13] * dt;
        const double cY = cos(y2 * 0.5), sY = sin(y2 * 0.5), cP = cos(p2 * 0.5), sP = sin(p2 * 0.5), cR = cos(r2 * 0.5), sR = sin(r2 * 0.5);
        after_buf[4] = sR * cP * cY - cR * sP * sY;
        after_buf[5] = cR * sP * cY + sR * cP * sY;
        after_buf[6] = cR * cP * sY - sR * sP * cY;
        after_buf[7] = cR * cP * cY + sR * sP * sY;
        after_buf[0] = 0.0; // odom_after.header.stamp is never set on this branch -> toSec() == 0
        after = after_buf;
    }
    const double dt_scan = d_cur_time - before[0], dt_trans = after[0] - before[0];
    Aff3f pb = aff_identity(), pa = aff_identity();
    for (int k = 0; k < 3; ++k) { pb.t[k] = (float)before[1 + k]; pa.t[k] = (float)after[1 + k]; }
    quat_to_matrix((float)before[7], (float)before[4], (float)before[5], (float)before[6], pb.l);
    quat_to_matrix((float)after[7], (float)after[4], (float)after[5], (float)after[6], pa.l);
    const Aff3f between = aff_mul(aff_inverse(pb), pa);
    const Aff3f res = aff_mul(pb, interpolate_tf(between, dt_scan, dt_trans));
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T_out[c * 4 + r] = res.l[r * 3 + c];
        T_out[12 + r] = res.t[r];
    }
    T_out[3] = T_out[7] = T_out[11] = 0.f;
    T_out[15] = 1.f;
    *ok = 1;
    return ELM_OK;
}

// Registration::CalFramePointCov (reg.hpp:211-217; called at reg.cpp:302-305 under use_radar_cov on the points in the MAP frame under the
// initial guess): per point the R S term of CalPointCov.  Plain host arithmetic, the function the radar kernel itself evaluates.
extern "C" int elm_cal_frame_point_cov(const double* xyz, size_t n, double range_var_m, double azim_var_deg, double ele_var_deg, double* cov9) {
    if (n && (!xyz || !cov9)) return ELM_ERR_INVALID;
    for (size_t i = 0; i < n; ++i) {
        double C[9];
        elm::radar_point_cov(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], range_var_m, azim_var_deg, ele_var_deg, C);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) cov9[9 * i + c * 3 + r] = C[r * 3 + c]; // column-major out
    }
    return ELM_OK;
}

extern "C" int elm_shape_odom_covariance(const double local_cov[36], const double icp_ego_pose[16], double d_icp_pose_std_m,
                                         double cov_out[36]) {
    if (!local_cov || !icp_ego_pose || !cov_out) return ELM_ERR_INVALID;
    const double std_m = std::max(d_icp_pose_std_m, 0.25); // pcm.cpp:1082
    double R[9], A[9], RA[9], tcov[9], rcov[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            R[r * 3 + c] = icp_ego_pose[c * 4 + r];
            A[r * 3 + c] = local_cov[c * 6 + r];             // block<3,3>(0,0), column-major 6x6
            rcov[r * 3 + c] = local_cov[(c + 3) * 6 + (r + 3)]; // block<3,3>(3,3)
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) RA[i * 3 + j] = (R[i * 3] * A[j] + R[i * 3 + 1] * A[3 + j]) + R[i * 3 + 2] * A[6 + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) tcov[i * 3 + j] = (RA[i * 3] * R[j * 3] + RA[i * 3 + 1] * R[j * 3 + 1]) + RA[i * 3 + 2] * R[j * 3 + 2];
    double tn[9], rn[9];
    normalize_covariance(tcov, tn);
    normalize_covariance(rcov, rn);
    const double angle_std = std_m * M_PI / 180.0;
    memset(cov_out, 0, 36 * sizeof(double));
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { // UpdateCovarianceField: row-major boost::array<double,36> (pcm.hpp:270-290)
            cov_out[r * 6 + c] = tn[r * 3 + c] * std_m * std_m;
            cov_out[(r + 3) * 6 + (c + 3)] = rn[r * 3 + c] * angle_std * angle_std;
        }
    return ELM_OK;
}


// ------------------------------------------------------------------------------------------------------
// PcmMatching::CallbackPointCloud as one call (pcm.cpp:198-324)
// ------------------------------------------------------------------------------------------------------
static void mul4_cm(const double* A, const double* B, double* C) { // column-major 4x4
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += A[k * 4 + r] * B[c * 4 + k];
            C[c * 4 + r] = s;
        }
}
// general inverse of a column-major 4x4 by Gauss-Jordan with partial pivoting (the reference calls Matrix4d::inverse())
static bool inv4_cm(const double* A, double* R) {
    double a[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) { a[r][c] = A[c * 4 + r]; a[r][4 + c] = (r == c) ? 1.0 : 0.0; }
    for (int k = 0; k < 4; ++k) {
        int p = k;
        for (int r = k + 1; r < 4; ++r)
            if (fabs(a[r][k]) > fabs(a[p][k])) p = r;
        if (a[p][k] == 0.0) return false;
        if (p != k)
            for (int c = 0; c < 8; ++c) std::swap(a[k][c], a[p][c]);
        const double piv = 1.0 / a[k][k];
        for (int c = 0; c < 8; ++c) a[k][c] *= piv;
        for (int r = 0; r < 4; ++r) {
            if (r == k) continue;
            const double f = a[r][k];
            for (int c = 0; c < 8; ++c) a[r][c] -= f * a[k][c];
        }
    }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) R[c * 4 + r] = a[r][4 + c];
    return true;
}

extern "C" int elm_pcm_callback_point_cloud(elm_ctx* ctx, const elm_map* map, const elm_pcm_node_config* node, const elm_reg_config* reg,
                                            const float* xyz, const float* point_time, size_t n, double stamp, const double* imu4,
                                            size_t n_imu, const double* odom14, size_t n_odom, elm_pcm_scan_output* out, int* published) {
    if (!ctx || !map || !node || !reg || !out || !published || (n && (!xyz || !point_time))) return ELM_ERR_INVALID;
    *published = 0;
    memset(out, 0, sizeof(*out));
    stamp -= node->lidar_time_delay; // pcm.cpp:216-217
    if (n == 0) return ELM_OK;       // "Input Empty!" (pcm.cpp:226-229)
    // The filtered cloud and the deskew tables are written straight into the context's page-locked staging buffer, laid out as
    // the device wants them ([tables | xyz] then the per-point times): two DMAs per scan, no pageable copies.  Scratch only grows:
    // no allocation on the per-scan path after warm-up.
    char* stage = (char*)elm_host::callback_staging(ctx, elm_host::kCbTableBytes + n * 4 * sizeof(float) + 64);
    if (!stage) return ELM_ERR_ALLOC;
    double* tab = (double*)stage;
    float* fx = (float*)(stage + elm_host::kCbTableBytes);
    float* ft = fx + 3 * n;
    const size_t TR = elm_host::kCbTableRows;
    size_t nf = 0;
    int rc = elm_filter_points_by_distance(xyz, point_time, n, node->input_max_dist, fx, ft, &nf); // :235
    if (rc != ELM_OK) return rc;
    out->n_filtered = nf;
    if (nf == 0) return ELM_OK;
    // DeskewPointCloud (pcm.cpp:467-531)
    const float front = ft[0], back = ft[nf - 1];
    if (node->lidar_scan_time_end)
        for (size_t i = 0; i < nf; ++i) ft[i] -= front; // :483-485
    elm_deskew_tables tabs;
    rc = elm_deskew_prepare(imu4, n_imu, odom14, n_odom, stamp, front, back, node->lidar_scan_time_end, node->run_deskew, tab,
                            tab + TR, tab + 2 * TR, tab + 3 * TR, TR, &tabs);
    if (rc != ELM_OK) return rc;
    // Deskew + VoxelDownsample + RunRegister in one device pass (the undistorted cloud stays in HBM and becomes the registration
    // source, the host waits once); voxel keys that do not pack, or a device group (the registration is sharded over its ranks), take the
    // stage-by-stage host path with identical results.
    int ok = 0, success = 0;
    double fit = 0.0, cov6[36], syncd[16], T0[16];
    float sync[16];
    if (!tabs.b_is_imu_available || !tabs.b_is_odom_available) return ELM_OK; // "Deskew fail!" (pcm.cpp:494-496, 238-241)
    out->time_scan_end = tabs.d_time_scan_end;
    rc = elm_get_interpolated_pose(odom14, n_odom, tabs.d_time_scan_end, sync, &ok); // :248-251 (a function of the odometry queue only)
    if (rc != ELM_OK || !ok) return rc;
    for (int i = 0; i < 16; ++i) syncd[i] = (double)sync[i];
    mul4_cm(syncd, node->tf_ego_to_lidar, T0); // sync_lidar_pose (pcm.cpp:266)
    int unpackable = 0;
    rc = elm_host::callback_register(ctx, map, stage, ft, nf, &tabs, node->input_voxel_ds_m, T0, reg, &out->result, &out->n_source, &unpackable);
    if (rc == ELM_OK && unpackable) rc = ELM_ERR_UNSUPPORTED;
    if (rc == ELM_OK) {
        memcpy(out->pose_lidar, out->result.T, sizeof(out->pose_lidar));
        memcpy(cov6, out->result.local_cov, sizeof(cov6));
        success = out->result.is_success;
        fit = out->result.fitness_score;
    } else if (rc == ELM_ERR_UNSUPPORTED) {
        // the staging buffer is about to be reused by elm_deskew / elm_register: take the inputs out of it first
        std::vector<float> hx(fx, fx + 3 * nf), ht(ft, ft + nf);
        std::vector<double> htab(tab, tab + 4 * TR);
        tabs.vec_d_imu_time = htab.data(); tabs.vec_d_imu_rot_x = htab.data() + TR; tabs.vec_d_imu_rot_y = htab.data() + 2 * TR; tabs.vec_d_imu_rot_z = htab.data() + 3 * TR;
        std::vector<float> und(3 * nf);
        rc = elm_deskew(ctx, hx.data(), ht.data(), nf, &tabs, und.data(), &ok);
        if (rc != ELM_OK) return rc;
        if (!ok) return ELM_OK; // "Deskew fail!" (pcm.cpp:238-241)
        std::vector<int64_t> keep(nf);
        size_t nk = 0;
        rc = elm_voxel_downsample(und.data(), nf, node->input_voxel_ds_m, keep.data(), &nk); // :257-258
        if (rc != ELM_OK) return rc;
        std::vector<float> src(3 * std::max<size_t>(nk, 1));
        for (size_t k = 0; k < nk; ++k) {
            const size_t i = (size_t)keep[k];
            src[3 * k] = und[3 * i]; src[3 * k + 1] = und[3 * i + 1]; src[3 * k + 2] = und[3 * i + 2];
        }
        out->n_source = nk;
        rc = elm_register(ctx, map, src.data(), nk, T0, reg, out->pose_lidar, &success, &fit, cov6, &out->result, nullptr); // :280-282
        if (rc != ELM_OK) return rc;
    } else {
        return rc;
    }
    if (!success) return ELM_OK; // pcm.cpp:289-292
    out->fitness_score = fit;
    double tinv[16];
    if (!inv4_cm(node->tf_ego_to_lidar, tinv)) return ELM_ERR_INVALID;
    mul4_cm(out->pose_lidar, tinv, out->pose_ego); // :298
    rc = elm_shape_odom_covariance(cov6, out->pose_ego, fit, out->covariance);
    if (rc != ELM_OK) return rc;
    *published = 1;
    return ELM_OK;
}


// ---- the reference's stdout lines of one RunRegister (reg.cpp:291-295, 343-356, 393-413) ------------------------------------------------
extern "C" size_t elm_format_register_log(const elm_reg_config* cfg, const elm_reg_result* res, size_t n_points, const elm_iter_trace* trace,
                                          const double* corr_ms, double total_ms, char* buf, size_t cap) {
    static const char* RESET = "\033[0m"; // localization_functions.hpp:78-85
    static const char* GREEN = "\033[32m";
    static const char* YELLOW = "\033[33m";
    std::ostringstream o; // std::cout's default formatting (%g, six significant digits)
    if (cfg && res) {
        const bool dbg = cfg->b_debug_print != 0;
        const float ratio = n_points ? (float)res->n_corr_last / (float)n_points : 0.f; // reg.cpp:351: (float) corr / total
        if (res->gate == 1) {
            o << YELLOW << "VOXEL MAP EMPTY!" << RESET << "\n"; // reg.cpp:291-295
        } else {
            double corr_total = 0.0;
            for (int i = 0; i < res->iterations; ++i) {
                if (corr_ms) corr_total += corr_ms[i];
                if (dbg && trace && corr_ms && i < ELM_MAX_ITER_TRACE) // reg.cpp:343-347
                    o << "[Registration] Total Correspondence Time for: " << (i + 1) << " in " << corr_ms[i] << " ms, and cores num: "
                      << (long long)trace[i].n_corr << RESET << "\n";
            }
            if (res->gate == 2) {
                o << YELLOW << "[RunRegister] Small corresponding  ratio. " << ratio << RESET << "\n"; // reg.cpp:352-356 (returns here)
            } else {
                if (dbg) { // reg.cpp:396-403
                    o << "[Registration] Total Correspondence Time: " << corr_total << " ms" << RESET << "\n";
                    o << "[Registration] RunRegister: iteration " << res->iterations << " executed in " << total_ms << " ms" << RESET << "\n";
                    o << GREEN << "[RunRegister] Corresponding ratio " << ratio << RESET << "\n";
                }
                if (res->gate == 3) o << YELLOW << "[RunRegister] ICP Fitness Score Low " << res->d_fitness << RESET << "\n"; // reg.cpp:405-409
                else if (dbg) o << GREEN << "[RunRegister] ICP Fitness Score " << res->d_fitness << RESET << "\n";            // reg.cpp:411-413
            }
        }
    }
    const std::string t = o.str();
    if (buf && cap) {
        const size_t n = std::min(t.size(), cap - 1);
        memcpy(buf, t.data(), n);
        buf[n] = 0;
    }
    return t.size();
}
