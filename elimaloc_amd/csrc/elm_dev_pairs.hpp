// elm_dev_pairs.hpp -- probes, the pair arithmetic of the four methods, partial-record publish
// Device-side helpers shared by the kernel translation units (every function is inline / a template: no symbol is emitted by itself).
#pragma once
#include <float.h>
#include <hip/hip_runtime.h>

#include "elm_internal.hpp"
#include "elm_la.hpp"

namespace elm {

// The kernel library (elm_k_*.hip) -- hand-written HIP kernels for gfx950 (CDNA4 / MI355X).
//
//   K1  k_accumulate_*        fused  T*p -> floor key -> neighbourhood probe -> nearest point / voxel mean
//                             -> residual + Jacobian -> block reduction of the packed normal equations
//                             (replaces TransformPoints reg.hpp:136-148, GetCorrespondence* vhm.cpp:31-206 and
//                             the serial loops of AlignCloudsLocal* reg.cpp:28-51, 85-132, 171-208):
//                             k_accumulate_cell (P2P / GICP), k_accumulate_vnbr (VGICP / AVGICP),
//                             k_accumulate_direct (the plain 27-probe walk: in-kernel reference and fall-back)
//   K2  k_solve               deterministic final reduction, overlap gate (reg.cpp:349-356), LM-damped LDLT solve,
//                             exp, pose composition, termination and fitness gates (reg.cpp:55-65, 378-387, 405-417)
//   K3  k_voxel_cov           VoxelBlock::CalVoxelCov for every voxel (vhm.hpp:114-148, 183-193)
//   K4  k_point_cov           ProcessVoxelBlock for every map point (vhm.hpp:195-257)
//   K0  k_deskew              DeskewPoint / FindRotation / FindPosition (pcm.cpp:731-824), float32 semantics
//
// Compiled with -ffp-contract=off: every discrete decision (voxel key, strict-< nearest neighbour, range test,
// gates) is taken on fp64 values computed in the same operation order as the reference's scalar code; fused
// multiply-adds are used only where written explicitly (fma()).



// ------------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nb) {
    // blocks are dispatched round-robin over the 8 XCDs; give every XCD one contiguous range of logical
    // blocks so that its private L2 sees one spatially compact part of the (cell-ordered) scans.
    unsigned q = nb >> 3, r = nb & 7u, xcd = bid & 7u, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

struct Probe {
    int vid;
    unsigned start, cnt;
};
__device__ __forceinline__ Probe probe_voxel(const DevMap& m, int kx, int ky, int kz) {
    unsigned h = hash3(kx, ky, kz) & m.mask;
    Probe p;
    p.vid = -1;
    p.start = 0;
    p.cnt = 0;
    for (;;) {
        const int4 key = *reinterpret_cast<const int4*>(&m.slots[h]);
        if (key.w < 0) break;
        if (key.x == kx && key.y == ky && key.z == kz) {
            const uint2 rg = *reinterpret_cast<const uint2*>(&m.slots[h].start);
            p.vid = key.w;
            p.start = rg.x;
            p.cnt = rg.y;
            break;
        }
        h = (h + 1) & m.mask;
    }
    return p;
}

__device__ __forceinline__ int floor_key(double g, double vs) { return (int)floor(g / vs); } // vhm.hpp:176-180
// same value without the float64 division when the voxel size is a power of two (uniform branch)
__device__ __forceinline__ int floor_key(double g, const DevMap& m) {
    return (m.inv_vs_exact != 0.0) ? (int)floor(g * m.inv_vs_exact) : (int)floor(g / m.voxel_size);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

constexpr double kCompactK = 999.0; // 1 / 1e-3 - 1: the k of U diag(1, 1, 1e-3) U^T (vhm.hpp:143, 243)
// inverse covariance I + k n n^T from the compact records (DevMap::grid_gicp8, VoxRec)
__device__ __forceinline__ void compact_cinv(double nx, double ny, double nz, double k, double* Ci) {
    const double kx = k * nx, ky = k * ny, kz = k * nz;
    Ci[0] = 1.0 + kx * nx; Ci[1] = kx * ny; Ci[2] = kx * nz;
    Ci[3] = Ci[1]; Ci[4] = 1.0 + ky * ny; Ci[5] = ky * nz;
    Ci[6] = Ci[2]; Ci[7] = Ci[5]; Ci[8] = 1.0 + kz * nz;
}

// upper-triangle packing of the symmetric 6x6: idx(i,j), i <= j
__host__ __device__ constexpr int tri(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

// sqrt of a squared distance (0 or a normal number well inside the exponent range): the core of the compiler's own expansion
// (v_rsq_f64 + the same seven fused steps) without its scaling / class handling for denormals, infinities and NaNs
__device__ __forceinline__ double sqrt_dist2(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    double d = __builtin_fma(-g, g, x);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return (x > 0.0) ? g : 0.0;
}
// a / b for normal b well inside the exponent range: reciprocal + two Newton steps + the final residual correction (the compiler's
// expansion without v_div_scale / v_div_fixup)
__device__ __forceinline__ double div_normal(double a, double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    const double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}
// the same without the final residual correction: within an ulp or two of a / b (the weights of the fused pair forms, which are
// not bit-for-bit restatements of the reference's arithmetic anyway)
__device__ __forceinline__ double div_close(double a, double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    return a * y;
}

// Adds one (source point, target) pair to the thread's packed sums.
//   acc[0..20] upper JTJ, acc[21..26] JTr, acc[27] residual sum, acc[28] pair count
// p = source point in the sensor frame, (mx,my,mz) = target position in the world frame,
// C = world-frame covariance of the target (row-major) or nullptr for the identity metric.
template <int METHOD>
__device__ __forceinline__ void add_pair(double* acc, const double* Rinv, const double* tinv, double px, double py,
                                         double pz, double mx, double my, double mz, const double* C,
                                         const double* nfit, const RegParams& rp) {
    // target_local = T^-1 * [m,1]  (reg.cpp:31 / 98 / 177)
    const double lx = ((Rinv[0] * mx + Rinv[1] * my) + Rinv[2] * mz) + tinv[0];
    const double ly = ((Rinv[3] * mx + Rinv[4] * my) + Rinv[5] * mz) + tinv[1];
    const double lz = ((Rinv[6] * mx + Rinv[7] * my) + Rinv[8] * mz) + tinv[2];
    const double rx = lx - px, ry = ly - py, rz = lz - pz; // residual_local
    const double r2 = (rx * rx + ry * ry) + rz * rz;
    const double den = rp.th + r2;
    double w = rp.th2 / (den * den); // square(th) / square(th + |r|^2)
    if (METHOD == ELM_GICP) w = w * 0.8 + 0.2;
    acc[28] += 1.0;
    if (METHOD == ELM_VGICP || METHOD == ELM_AVGICP) {
        if (w < 0.01) return; // reg.cpp:201 -- skipped pairs stay in the fitness denominator
    }
    // A = w * M, M = (Rinv C Rinv^T)^-1 (reg.cpp:107-113, 187-191) or I
    double A[9];
    if (METHOD == ELM_P2P) {
        A[0] = w; A[1] = 0; A[2] = 0; A[3] = 0; A[4] = w; A[5] = 0; A[6] = 0; A[7] = 0; A[8] = w;
    } else {
        double RC[9], RCR[9], M[9];
        mul3(Rinv, C, RC);
        mul3_bt(RC, Rinv, RCR);
        inv3(RCR, M);
#pragma unroll
        for (int i = 0; i < 9; ++i) A[i] = w * M[i];
    }
    // B = -[p]x
    //     [  0   pz  -py ]
    //     [ -pz  0    px ]
    //     [  py -px   0  ]
    double AB[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        AB[i * 3 + 0] = A[i * 3 + 2] * py - A[i * 3 + 1] * pz;
        AB[i * 3 + 1] = A[i * 3 + 0] * pz - A[i * 3 + 2] * px;
        AB[i * 3 + 2] = A[i * 3 + 1] * px - A[i * 3 + 0] * py;
    }
    // translation block (upper triangle of A)
    acc[tri(0, 0)] += A[0]; acc[tri(0, 1)] += A[1]; acc[tri(0, 2)] += A[2];
    acc[tri(1, 1)] += A[4]; acc[tri(1, 2)] += A[5]; acc[tri(2, 2)] += A[8];
    // translation x rotation block: all nine entries of A B
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[tri(i, 3 + j)] += AB[i * 3 + j];
    // rotation block: B^T (A B), upper triangle.  B^T rows: (0,-pz,py) (pz,0,-px) (-py,px,0)
    acc[tri(3, 3)] += py * AB[6] - pz * AB[3];
    acc[tri(3, 4)] += py * AB[7] - pz * AB[4];
    acc[tri(3, 5)] += py * AB[8] - pz * AB[5];
    acc[tri(4, 4)] += pz * AB[1] - px * AB[7];
    acc[tri(4, 5)] += pz * AB[2] - px * AB[8];
    acc[tri(5, 5)] += px * AB[5] - py * AB[2];
    // J^T (w M) r
    const double ax = (A[0] * rx + A[1] * ry) + A[2] * rz;
    const double ay = (A[3] * rx + A[4] * ry) + A[5] * rz;
    const double az = (A[6] * rx + A[7] * ry) + A[8] * rz;
    acc[21] += ax; acc[22] += ay; acc[23] += az;
    acc[24] += py * az - pz * ay;
    acc[25] += pz * ax - px * az;
    acc[26] += px * ay - py * ax;
    if (METHOD == ELM_GICP) {
        // |r . n_l|, n_l = normalised Rinv * n (reg.cpp:91-95, 128)
        double nx = (Rinv[0] * nfit[0] + Rinv[1] * nfit[1]) + Rinv[2] * nfit[2];
        double ny = (Rinv[3] * nfit[0] + Rinv[4] * nfit[1]) + Rinv[5] * nfit[2];
        double nz = (Rinv[6] * nfit[0] + Rinv[7] * nfit[1]) + Rinv[8] * nfit[2];
        const double nn2 = (nx * nx + ny * ny) + nz * nz;
        if (nn2 > 0.0) {
            const double nn = sqrt(nn2);
            nx /= nn; ny /= nn; nz /= nn;
        }
        acc[27] += fabs((rx * nx + ry * ny) + rz * nz);
    } else {
        acc[27] += sqrt(r2);
    }
}

// ---- use_radar_cov = 1 (reg.hpp:186-217, reg.cpp:109-111 / 188-190 / 302-305) ----------------------------------------------------
// The reference attaches a "covariance" R S to every source point -- R = Rz(azimuth) Ry(elevation) of the point's MAP-frame position
// under the initial guess, S = diag(range spread, max(0.1, d sin(azimuth spread)), max(0.1, d sin(elevation spread))), d the
// horizontal range; a product, not R S R^T: the matrix is not symmetric -- and adds it to R^-1 C R^-T before the inversion.  The
// re-transform at the end of an iteration replaces it by the identity (see oracle/elm_oracle.cpp, orc_register), so the first
// iteration sees R S and every later one I.  With a non-symmetric metric J^T M J is not symmetric either: JTJ.ldlt() reads its lower
// triangle, GICP's covariance output is the inverse of the full matrix, so all 36 entries are accumulated -- in the sensor frame, with
// the reference's own sequence of 3x3 products (the world-frame form of add_pair_world needs a symmetric C^-1 computed in advance).
//   acc[0..35] J^T M J row-major, acc[36..41] J^T M r, acc[42] residual sum, acc[43] pair count, acc[44..46] search statistics
constexpr int kRadarAcc = 47;
constexpr int kRadarSums = 64; // doubles per partial record of the radar kernel
__device__ __forceinline__ void radar_source_cov(double gx, double gy, double gz, const RegParams& rp, double* Cs) {
    radar_point_cov(gx, gy, gz, rp.radar_var[0], rp.radar_var[1], rp.radar_var[2], Cs); // (elm_la.hpp: the same code serves elm_cal_frame_point_cov)
}
// p = source point in the sensor frame, (mx, my, mz) = target position and C = target covariance (row-major) in the map frame,
// Cs = the source point's covariance term, nfit = GICP's plane normal in the map frame
template <int METHOD>
__device__ __forceinline__ void add_pair_radar(double* acc, const double* Rinv, const double* tinv, double px, double py, double pz,
                                               double mx, double my, double mz, const double* C, const double* Cs, const double* nfit,
                                               const RegParams& rp) {
    const double lx = ((Rinv[0] * mx + Rinv[1] * my) + Rinv[2] * mz) + tinv[0];
    const double ly = ((Rinv[3] * mx + Rinv[4] * my) + Rinv[5] * mz) + tinv[1];
    const double lz = ((Rinv[6] * mx + Rinv[7] * my) + Rinv[8] * mz) + tinv[2];
    const double rx = lx - px, ry = ly - py, rz = lz - pz; // residual_local
    const double r2 = (rx * rx + ry * ry) + rz * rz;
    const double den = rp.th + r2;
    double w = rp.th2 / (den * den);
    if (METHOD == ELM_GICP) w = w * 0.8 + 0.2;
    acc[43] += 1.0;
    if (METHOD == ELM_VGICP || METHOD == ELM_AVGICP) {
        if (w < 0.01) return; // reg.cpp:201
    }
    double RC[9], RCR[9], M[9], A[9];
    mul3(Rinv, C, RC);
    mul3_bt(RC, Rinv, RCR);
#pragma unroll
    for (int i = 0; i < 9; ++i) RCR[i] += Cs[i]; // reg.cpp:109-111 / 188-190
    inv3(RCR, M);
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = w * M[i];
    // J = [I | B], B = -[p]x = rows (0, pz, -py) (-pz, 0, px) (py, -px, 0); B^T = rows (0, -pz, py) (pz, 0, -px) (-py, px, 0)
    double AB[9], BtA[9], BtAB[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        AB[i * 3 + 0] = A[i * 3 + 2] * py - A[i * 3 + 1] * pz;
        AB[i * 3 + 1] = A[i * 3 + 0] * pz - A[i * 3 + 2] * px;
        AB[i * 3 + 2] = A[i * 3 + 1] * px - A[i * 3 + 0] * py;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        BtA[0 * 3 + j] = py * A[6 + j] - pz * A[3 + j];
        BtA[1 * 3 + j] = pz * A[0 + j] - px * A[6 + j];
        BtA[2 * 3 + j] = px * A[3 + j] - py * A[0 + j];
        BtAB[0 * 3 + j] = py * AB[6 + j] - pz * AB[3 + j];
        BtAB[1 * 3 + j] = pz * AB[0 + j] - px * AB[6 + j];
        BtAB[2 * 3 + j] = px * AB[3 + j] - py * AB[0 + j];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            acc[i * 6 + j] += A[i * 3 + j];
            acc[i * 6 + 3 + j] += AB[i * 3 + j];
            acc[(3 + i) * 6 + j] += BtA[i * 3 + j];
            acc[(3 + i) * 6 + 3 + j] += BtAB[i * 3 + j];
        }
    const double ax = (A[0] * rx + A[1] * ry) + A[2] * rz;
    const double ay = (A[3] * rx + A[4] * ry) + A[5] * rz;
    const double az = (A[6] * rx + A[7] * ry) + A[8] * rz;
    acc[36] += ax; acc[37] += ay; acc[38] += az;
    acc[39] += py * az - pz * ay;
    acc[40] += pz * ax - px * az;
    acc[41] += px * ay - py * ax;
    if (METHOD == ELM_GICP) {
        double nx = (Rinv[0] * nfit[0] + Rinv[1] * nfit[1]) + Rinv[2] * nfit[2];
        double ny = (Rinv[3] * nfit[0] + Rinv[4] * nfit[1]) + Rinv[5] * nfit[2];
        double nz = (Rinv[6] * nfit[0] + Rinv[7] * nfit[1]) + Rinv[8] * nfit[2];
        const double nn2 = (nx * nx + ny * ny) + nz * nz;
        if (nn2 > 0.0) {
            const double nn = sqrt(nn2);
            nx /= nn; ny /= nn; nz /= nn;
        }
        acc[42] += fabs((rx * nx + ry * ny) + rz * nz);
    } else {
        acc[42] += sqrt(r2);
    }
}

// The covariance-weighted methods in the WORLD frame.  With a = R p = g - t and the world residual e = m - g:
//   r_l = R^-1 e,   M_l = (R^-1 C R^-T)^-1 = R^T C^-1 R,   R [p]x R^T = [a]x
//   =>  J_l^T M_l J_l = P^T (J_w^T C^-1 J_w) P,   J_l^T M_l r_l = P^T (J_w^T C^-1 e),   J_w = [I | -[a]x],  P = diag(R, R)
// so the per-pair 3x3 products and the 3x3 inverse of reg.cpp:107-113 / 187-191 disappear: C^-1 is computed once per map point /
// voxel at map build, the pairs are accumulated in the world frame and k_solve applies the one congruence with P per
// iteration.  Equal to the reference's sums up to rounding (R orthonormal to ~1e-16).  |r_l| = |e|.
//   acc[0..20] upper J^T M J, acc[21..26] J^T M r, acc[27] residual sum, acc[28] pair count  -- world frame
// ASSIGN: the caller adds exactly one pair to all-zero sums (plain stores instead of additions: 0.0 + x cannot be folded by the
// compiler, and the zeros cost registers)
template <int METHOD, bool ASSIGN>
__device__ __forceinline__ void add_pair_world(double* acc, double ax, double ay, double az, double ex, double ey, double ez,
                                               const double* Cinv, const double* nfit, const RegParams& rp) {
#define ELM_ACC(k, x) do { if (ASSIGN) acc[k] = (x); else acc[k] += (x); } while (0)
    const double r2 = (ex * ex + ey * ey) + ez * ez;
    const double den = rp.th + r2;
    double w = rp.th2 / (den * den); // square(th) / square(th + |r|^2)
    if (METHOD == ELM_GICP) w = w * 0.8 + 0.2;
    ELM_ACC(28, 1.0);
    if (METHOD == ELM_VGICP || METHOD == ELM_AVGICP) {
        if (w < 0.01) return; // reg.cpp:201 -- skipped pairs stay in the fitness denominator
    }
    double A[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = w * Cinv[i];
    // B = -[a]x
    double AB[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        AB[i * 3 + 0] = A[i * 3 + 2] * ay - A[i * 3 + 1] * az;
        AB[i * 3 + 1] = A[i * 3 + 0] * az - A[i * 3 + 2] * ax;
        AB[i * 3 + 2] = A[i * 3 + 1] * ax - A[i * 3 + 0] * ay;
    }
    ELM_ACC(tri(0, 0), A[0]); ELM_ACC(tri(0, 1), A[1]); ELM_ACC(tri(0, 2), A[2]);
    ELM_ACC(tri(1, 1), A[4]); ELM_ACC(tri(1, 2), A[5]); ELM_ACC(tri(2, 2), A[8]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) ELM_ACC(tri(i, 3 + j), AB[i * 3 + j]);
    ELM_ACC(tri(3, 3), ay * AB[6] - az * AB[3]);
    ELM_ACC(tri(3, 4), ay * AB[7] - az * AB[4]);
    ELM_ACC(tri(3, 5), ay * AB[8] - az * AB[5]);
    ELM_ACC(tri(4, 4), az * AB[1] - ax * AB[7]);
    ELM_ACC(tri(4, 5), az * AB[2] - ax * AB[8]);
    ELM_ACC(tri(5, 5), ax * AB[5] - ay * AB[2]);
    const double bx = (A[0] * ex + A[1] * ey) + A[2] * ez;
    const double by = (A[3] * ex + A[4] * ey) + A[5] * ez;
    const double bz = (A[6] * ex + A[7] * ey) + A[8] * ez;
    ELM_ACC(21, bx); ELM_ACC(22, by); ELM_ACC(23, bz);
    ELM_ACC(24, ay * bz - az * by);
    ELM_ACC(25, az * bx - ax * bz);
    ELM_ACC(26, ax * by - ay * bx);
    if (METHOD == ELM_GICP) {
        // |r_l . n_l| with n_l the normalised R^-1 n (reg.cpp:91-95, 128) = |e . n| for the unit normal the map build stores
        ELM_ACC(27, fabs((ex * nfit[0] + ey * nfit[1]) + ez * nfit[2]));
    } else {
        ELM_ACC(27, sqrt(r2));
    }
}
#undef ELM_ACC

// AVGICP forms up to seven pairs per scan point, all with the same Jacobian J_w = [I | -[a]x] (a = R p): sum_v J^T A_v J =
// J^T (sum_v A_v) J and sum_v J^T A_v e_v = J^T sum_v (A_v e_v) with A_v = w_v C_v^-1.  The pairs are therefore gathered into
// one 3x3 + one 3-vector per point (AvgPairSum, 27 registers) and expanded ONCE (by the block reduction, PairSum below) -- the
// same sums as seven add_pair_world calls up to the order of the additions.
struct AvgPairSum {
    double A[9], b[3], rsum, n;
};
__device__ __forceinline__ void avg_pair_init(AvgPairSum& P) {
#pragma unroll
    for (int i = 0; i < 9; ++i) P.A[i] = 0.0;
    P.b[0] = P.b[1] = P.b[2] = 0.0;
    P.rsum = 0.0;
    P.n = 0.0;
}
__device__ __forceinline__ void avg_pair_add(AvgPairSum& P, double ex, double ey, double ez, const double* __restrict__ Cinv, const RegParams& rp) {
    const double r2 = (ex * ex + ey * ey) + ez * ez;
    const double den = rp.th + r2;
    const double w = div_normal(rp.th2, den * den); // square(th) / square(th + |r|^2)
    P.n += 1.0;
    if (w < 0.01) return; // reg.cpp:201 -- skipped pairs stay in the fitness denominator
    double A[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        A[i] = w * Cinv[i];
        P.A[i] += A[i];
    }
    P.b[0] += (A[0] * ex + A[1] * ey) + A[2] * ez;
    P.b[1] += (A[3] * ex + A[4] * ey) + A[5] * ez;
    P.b[2] += (A[6] * ex + A[7] * ey) + A[8] * ez;
    P.rsum += sqrt_dist2(r2);
}

// ------------------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------------------
// Nearest bucket point straight from global memory (one thread walks its 27 voxels): the "direct" kernel.
// GetCorrespondencePoints (vhm.cpp:31-88): strict-< minimum over every bucket point of the 27 voxels, voxels
// visited x-major .. z-minor (vhm.cpp:234-240), bucket in insertion order.
__device__ __forceinline__ void nearest_point_direct(const DevMap& m, int vx, int vy, int vz, double gx, double gy,
                                                     double gz, double& bd2, float& bx, float& by, float& bz,
                                                     int& bidx, double& n_cand, double& n_occ) {
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
                const Probe pr = probe_voxel(m, vx + dx, vy + dy, vz + dz);
                if (pr.vid < 0) continue;
                n_occ += 1.0;
                n_cand += (double)pr.cnt;
                for (unsigned j = 0; j < pr.cnt; ++j) {
                    const float4 q = m.pts[pr.start + j];
                    const double ex = (double)q.x - gx, ey = (double)q.y - gy, ez = (double)q.z - gz;
                    const double d2 = (ex * ex + ey * ey) + ez * ez;
                    if (d2 < bd2) {
                        bd2 = d2;
                        bx = q.x; by = q.y; bz = q.z;
                        bidx = (int)(pr.start + j);
                    }
                }
            }
}
// GetCorrespondencesCov (vhm.cpp:90-151): nearest voxel MEAN among the existing neighbours, from global memory
__device__ __forceinline__ void nearest_voxel_direct(const DevMap& m, int vx, int vy, int vz, double gx, double gy,
                                                     double gz, double& bd2, int& bvid, double& bmx, double& bmy,
                                                     double& bmz, double& n_cand, double& n_occ) {
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
                const Probe pr = probe_voxel(m, vx + dx, vy + dy, vz + dz);
                if (pr.vid < 0 || pr.cnt == 0) continue;
                n_occ += 1.0;
                n_cand += 1.0;
                const double cx = m.vox_mean[(size_t)pr.vid * 3], cy = m.vox_mean[(size_t)pr.vid * 3 + 1], cz = m.vox_mean[(size_t)pr.vid * 3 + 2];
                const double ex = cx - gx, ey = cy - gy, ez = cz - gz;
                const double d2 = (ex * ex + ey * ey) + ez * ez;
                if (d2 < bd2) { bd2 = d2; bvid = pr.vid; bmx = cx; bmy = cy; bmz = cz; }
            }
}

// pair payloads shared by the kernels
template <int METHOD, bool ASSIGN = false>
__device__ __forceinline__ void finish_point_pair(double* acc, const DevMap& m, const ScanState& S, const RegParams& rp,
                                                  double px, double py, double pz, double gx, double gy, double gz,
                                                  double bd2, float bx, float by, float bz, int bidx, const double* __restrict__ payload) {
    // no bucket at all: the reference's default PointStruct at the origin with cov I (vhm.cpp:37, QUIRK)
    const double dfin = (bidx >= 0) ? bd2 : (gx * gx + gy * gy) + gz * gz;
    if (!(dfin < rp.th2)) return;
    if (METHOD == ELM_P2P) {
        if (bidx < 0) { bx = 0.f; by = 0.f; bz = 0.f; }
        add_pair<ELM_P2P>(acc, S.Rinv, S.tinv, px, py, pz, (double)bx, (double)by, (double)bz, nullptr, nullptr, rp);
    } else {
        double Ci[9], mean[3], nf[3];
        if (bidx >= 0) {
            const double* __restrict__ rec = payload + (size_t)bidx * 16;
#pragma unroll
            for (int k = 0; k < 9; ++k) Ci[k] = rec[3 + k];
#pragma unroll
            for (int k = 0; k < 3; ++k) { mean[k] = rec[k]; nf[k] = rec[12 + k]; }
        } else {
            Ci[0] = 1; Ci[1] = 0; Ci[2] = 0; Ci[3] = 0; Ci[4] = 1; Ci[5] = 0; Ci[6] = 0; Ci[7] = 0; Ci[8] = 1;
            mean[0] = mean[1] = mean[2] = 0.0;
            nf[0] = 1.0; nf[1] = 0.0; nf[2] = 0.0;
        }
        // GICP's target position is the neighbourhood MEAN of the matched point (reg.cpp:97)
        add_pair_world<ELM_GICP, ASSIGN>(acc, gx - S.T[12], gy - S.T[13], gz - S.T[14], mean[0] - gx, mean[1] - gy, mean[2] - gz, Ci, nf, rp);
    }
}
template <int METHOD, bool ASSIGN = false>
__device__ __forceinline__ void voxel_pair(double* acc, const DevMap& m, const ScanState& S, const RegParams& rp, double gx, double gy,
                                           double gz, int vid, double mx, double my, double mz) {
    double Ci[9];
    if (vid >= 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Ci[k] = m.vox_cinv[(size_t)vid * 9 + k];
    } else {
        Ci[0] = 1; Ci[1] = 0; Ci[2] = 0; Ci[3] = 0; Ci[4] = 1; Ci[5] = 0; Ci[6] = 0; Ci[7] = 0; Ci[8] = 1;
    }
    add_pair_world<METHOD, ASSIGN>(acc, gx - S.T[12], gy - S.T[13], gz - S.T[14], mx - gx, my - gy, mz - gz, Ci, nullptr, rp);
}
template <bool ASSIGN = false>
__device__ __forceinline__ void finish_voxel_pair(double* acc, const DevMap& m, const ScanState& S, const RegParams& rp,
                                                  double px, double py, double pz, double gx, double gy, double gz,
                                                  double bd2, int bvid, double bmx, double bmy, double bmz) {
    const double dfin = (bvid >= 0) ? bd2 : (gx * gx + gy * gy) + gz * gz;
    if (!(dfin < rp.th2)) return;
    if (bvid < 0) bmx = bmy = bmz = 0.0;
    voxel_pair<ELM_VGICP, ASSIGN>(acc, m, S, rp, gx, gy, gz, bvid, bmx, bmy, bmz);
}

// block -> (scan, first point) ; returns false when the scan is finished
__device__ __forceinline__ int find_scan(const ScanDesc* __restrict__ scans, int batch, unsigned L, const RegParams& rp) {
    if (rp.uniform_blocks) return (int)(L / rp.uniform_blocks); // no dependent descriptor loads at the head of the workgroup
    int lo = 0, hi = batch - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (scans[mid].blk_begin <= L) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// wave reduction (64 lanes) of the 31 packed sums, then the four waves of the block through LDS
__device__ __forceinline__ void block_reduce_store(double* acc, double (*red)[32], double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const double v = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 32)
        out[threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}


// A workgroup's 32 block sums -> its partial record; k_solve reduces the scan's records in a fixed order.  (Rounds 2-5 also carried a fused
// form -- ticket counter per scan, the last workgroup reduces: -3 % on one rank, measured twice -- removed in round 6:
// profiles/r06_removed_fused_reduce.patch.)
__device__ __forceinline__ void publish_and_reduce(double value, unsigned L, int, unsigned, unsigned, double* __restrict__ partials, const RegParams&, double*) {
    if (threadIdx.x < (unsigned)kSums) partials[(size_t)L * kSums + threadIdx.x] = value;
}

// DPP row operations (quad permutes, row rotations / mirrors) instead of ds_bpermute (__shfl), which goes through the LDS pipeline
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false));
}

// slot `slot` of the voxel-mean lists: its word (voxel id | position code << kVidBits; -1: padding) / its voxel id
__device__ __forceinline__ int vnbr_vc(const DevMap& m, unsigned slot) { return m.vnbr_blk[slot >> 2].vc[slot & 3u]; }
__device__ __forceinline__ unsigned vnbr_vid(const DevMap& m, unsigned slot) { return (unsigned)vnbr_vc(m, slot) & kVidMask; }

} // namespace elm
