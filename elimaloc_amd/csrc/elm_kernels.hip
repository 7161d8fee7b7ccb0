// elm_kernels.hip -- hand-written HIP kernels for gfx950 (CDNA4 / MI355X).
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
#include <float.h>
#include <algorithm>
#include <hip/hip_runtime.h>

#include "elm_internal.hpp"
#include "elm_la.hpp"

namespace elm {


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

// ---- K1a: direct kernel (first correct version; kept for A/B measurements, ELM_KERNEL=direct) --------------
template <int METHOD>
__global__ __launch_bounds__(kBlock) void k_accumulate_direct(const DevMap m, const ScanDesc* __restrict__ scans,
                                                              int batch, unsigned total_blocks,
                                                              const ScanState* __restrict__ st,
                                                              double* __restrict__ partials, const RegParams rp) {
    const unsigned L = xcd_remap(blockIdx.x, total_blocks);
    const int s = find_scan(scans, batch, L, rp);
    const ScanState& S = st[s];
    if (S.done) return; // uniform: the whole block leaves; k_solve skips this scan too
    const ScanDesc sd = scans[s];
    if (L >= sd.blk_end) return; // a scan whose size was only known on the device owns fewer workgroups than were launched for it
    const unsigned i = (L - sd.blk_begin) * kBlock + threadIdx.x;
    const bool valid = i < sd.n;
    double acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
    if (valid) {
        const Pt3 pf = sd.pts[i];
        const double px = pf.x, py = pf.y, pz = pf.z;
        // g = T * [p,1]  (reg.hpp:141-146), same association as the reference's scalar product
        const double gx = ((S.T[0] * px + S.T[4] * py) + S.T[8] * pz) + S.T[12];
        const double gy = ((S.T[1] * px + S.T[5] * py) + S.T[9] * pz) + S.T[13];
        const double gz = ((S.T[2] * px + S.T[6] * py) + S.T[10] * pz) + S.T[14];
        const int vx = floor_key(gx, m.voxel_size), vy = floor_key(gy, m.voxel_size), vz = floor_key(gz, m.voxel_size);
        double n_cand = 0.0, n_occ = 0.0;
        if (METHOD == ELM_P2P || METHOD == ELM_GICP) {
            double bd2 = DBL_MAX;
            float bx = 0.f, by = 0.f, bz = 0.f;
            int bidx = -1;
            nearest_point_direct(m, vx, vy, vz, gx, gy, gz, bd2, bx, by, bz, bidx, n_cand, n_occ);
            finish_point_pair<METHOD>(acc, m, S, rp, px, py, pz, gx, gy, gz, bd2, bx, by, bz, bidx, m.pt_gicp);
        } else if (METHOD == ELM_VGICP) {
            double bd2 = DBL_MAX, bmx = 0.0, bmy = 0.0, bmz = 0.0;
            int bvid = -1;
            nearest_voxel_direct(m, vx, vy, vz, gx, gy, gz, bd2, bvid, bmx, bmy, bmz, n_cand, n_occ);
            finish_voxel_pair(acc, m, S, rp, px, py, pz, gx, gy, gz, bd2, bvid, bmx, bmy, bmz);
        } else {
            // GetCorrespondencesAllCov (vhm.cpp:153-206): every existing face-neighbour voxel within range is a pair,
            // order (0, +x, -x, +y, -y, +z, -z) (vhm.cpp:224-230)
            const int ox[7] = {0, 1, -1, 0, 0, 0, 0}, oy[7] = {0, 0, 0, 1, -1, 0, 0}, oz[7] = {0, 0, 0, 0, 0, 1, -1};
#pragma unroll
            for (int k7 = 0; k7 < 7; ++k7) {
                const Probe pr = probe_voxel(m, vx + ox[k7], vy + oy[k7], vz + oz[k7]);
                if (pr.vid < 0 || pr.cnt == 0) continue;
                n_occ += 1.0;
                n_cand += 1.0;
                const double cx = m.vox_mean[(size_t)pr.vid * 3], cy = m.vox_mean[(size_t)pr.vid * 3 + 1], cz = m.vox_mean[(size_t)pr.vid * 3 + 2];
                const double ex = cx - gx, ey = cy - gy, ez = cz - gz;
                const double d2 = (ex * ex + ey * ey) + ez * ez;
                if (d2 < rp.th2) voxel_pair<ELM_AVGICP>(acc, m, S, rp, gx, gy, gz, pr.vid, cx, cy, cz);
            }
        }
        if (rp.stats) { // the work counters read 0 unless elm_ctx_set_work_counters(ctx, 1), whatever the search index
            acc[29] = n_cand;
            acc[30] = n_occ;
            acc[31] = n_cand; // every candidate is distance-tested on this path
        }
    }
    __shared__ double red[kBlock / 64][32];
    __shared__ double s_scr[8 * kSums + 2];
    double sum = 0.0;
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const double v = wave_sum(acc[k]);
            if (lane == 0) red[wave][k] = v;
        }
        __syncthreads();
        if (threadIdx.x < 32) sum = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    }
    publish_and_reduce(sum, L, s, sd.blk_begin, sd.blk_end, partials, rp, s_scr);
}

// ---- K1r: the radar-covariance variant of the direct kernel (use_radar_cov = 1, methods with covariances) ----------------------------
// One thread per scan point, the plain 27-probe (7-probe) walk, 64-double partial records (see add_pair_radar).  A configuration for
// radar sensors with a few hundred returns per scan: not a throughput path.
template <int METHOD>
__global__ __launch_bounds__(kBlock) void k_accumulate_radar(const DevMap m, const ScanDesc* __restrict__ scans, int batch, unsigned total_blocks,
                                                             const ScanState* __restrict__ st, double* __restrict__ partials, const RegParams rp) {
    const unsigned L = xcd_remap(blockIdx.x, total_blocks);
    const int s = find_scan(scans, batch, L, rp);
    const ScanState& S = st[s];
    if (S.done) return;
    const ScanDesc sd = scans[s];
    if (L >= sd.blk_end) return;
    const unsigned i = (L - sd.blk_begin) * kBlock + threadIdx.x;
    double acc[kRadarAcc];
#pragma unroll
    for (int k = 0; k < kRadarAcc; ++k) acc[k] = 0.0;
    if (i < sd.n) {
        const Pt3 pf = sd.pts[i];
        const double px = pf.x, py = pf.y, pz = pf.z;
        const double gx = ((S.T[0] * px + S.T[4] * py) + S.T[8] * pz) + S.T[12];
        const double gy = ((S.T[1] * px + S.T[5] * py) + S.T[9] * pz) + S.T[13];
        const double gz = ((S.T[2] * px + S.T[6] * py) + S.T[10] * pz) + S.T[14];
        double Cs[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (rp.radar == 2) { // ELM_CHECK=strict_pairs: the reference's arithmetic of use_radar_cov = 0 -- no source term at all
#pragma unroll
            for (int k = 0; k < 9; ++k) Cs[k] = 0.0;
        } else if (S.iters == 0) radar_source_cov(gx, gy, gz, rp, Cs); // S.T is still the initial guess: g is the pose CalFramePointCov reads
        const int vx = floor_key(gx, m.voxel_size), vy = floor_key(gy, m.voxel_size), vz = floor_key(gz, m.voxel_size);
        const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        double n_cand = 0.0, n_occ = 0.0;
        if (METHOD == ELM_GICP) {
            double bd2 = DBL_MAX;
            float bx = 0.f, by = 0.f, bz = 0.f;
            int bidx = -1;
            nearest_point_direct(m, vx, vy, vz, gx, gy, gz, bd2, bx, by, bz, bidx, n_cand, n_occ);
            const double dfin = (bidx >= 0) ? bd2 : (gx * gx + gy * gy) + gz * gz; // no bucket: the default PointStruct at the origin (vhm.cpp:37)
            if (dfin < rp.th2) {
                double C[9], mean[3] = {0.0, 0.0, 0.0}, nf[3] = {1.0, 0.0, 0.0};
#pragma unroll
                for (int k = 0; k < 9; ++k) C[k] = ident[k];
                if (bidx >= 0) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) C[k] = m.pt_cov[(size_t)bidx * 9 + k];
#pragma unroll
                    for (int k = 0; k < 3; ++k) { mean[k] = m.pt_gicp[(size_t)bidx * 16 + k]; nf[k] = m.pt_gicp[(size_t)bidx * 16 + 12 + k]; }
                }
                add_pair_radar<ELM_GICP>(acc, S.Rinv, S.tinv, px, py, pz, mean[0], mean[1], mean[2], C, Cs, nf, rp);
            }
        } else if (METHOD == ELM_VGICP) {
            double bd2 = DBL_MAX, bmx = 0.0, bmy = 0.0, bmz = 0.0;
            int bvid = -1;
            nearest_voxel_direct(m, vx, vy, vz, gx, gy, gz, bd2, bvid, bmx, bmy, bmz, n_cand, n_occ);
            const double dfin = (bvid >= 0) ? bd2 : (gx * gx + gy * gy) + gz * gz;
            if (dfin < rp.th2) {
                double C[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) C[k] = (bvid >= 0) ? m.vox_cov[(size_t)bvid * 9 + k] : ident[k];
                if (bvid < 0) bmx = bmy = bmz = 0.0;
                add_pair_radar<ELM_VGICP>(acc, S.Rinv, S.tinv, px, py, pz, bmx, bmy, bmz, C, Cs, nullptr, rp);
            }
        } else {
            const int ox[7] = {0, 1, -1, 0, 0, 0, 0}, oy[7] = {0, 0, 0, 1, -1, 0, 0}, oz[7] = {0, 0, 0, 0, 0, 1, -1};
            for (int k7 = 0; k7 < 7; ++k7) {
                const Probe pr = probe_voxel(m, vx + ox[k7], vy + oy[k7], vz + oz[k7]);
                if (pr.vid < 0 || pr.cnt == 0) continue;
                n_occ += 1.0;
                n_cand += 1.0;
                const double cx = m.vox_mean[(size_t)pr.vid * 3], cy = m.vox_mean[(size_t)pr.vid * 3 + 1], cz = m.vox_mean[(size_t)pr.vid * 3 + 2];
                const double ex = cx - gx, ey = cy - gy, ez = cz - gz;
                const double d2 = (ex * ex + ey * ey) + ez * ez;
                if (d2 < rp.th2) {
                    double C[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k) C[k] = m.vox_cov[(size_t)pr.vid * 9 + k];
                    add_pair_radar<ELM_AVGICP>(acc, S.Rinv, S.tinv, px, py, pz, cx, cy, cz, C, Cs, nullptr, rp);
                }
            }
        }
        if (rp.stats) {
            acc[44] = n_cand;
            acc[45] = n_occ;
            acc[46] = n_cand;
        }
    }
    __shared__ double red[kBlock / 64][kRadarSums];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kRadarAcc; ++k) {
        const double v = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)kRadarSums) {
        const int k = threadIdx.x;
        partials[(size_t)L * kRadarSums + k] = (k < kRadarAcc) ? ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k] : 0.0;
    }
}

// ---- K1 on the neighbourhood lists (fall-back search index) ---------------------------------------------------------
// The candidate list of a query voxel (the points of its 27 buckets) is kept sorted by half-voxel cells: a 6x6x6 grid with origin
// (v - 1) * voxel_size (per axis), edge h = voxel_size / 2, indices clamped to 0..5 so the outermost cells are unbounded outward
// (truncated keys make the bucket of voxel key 0 two voxels wide).  Cells are ordered (ix, iy, iz)-major, so the cells iz0..iz1
// of one (ix, iy) column are one contiguous range; a 16-byte record per column holds its seven cell boundaries.  The two stages
// are those of k_accumulate_grid (below), on one list per query voxel instead of the global grid.
constexpr int kCellAxis = 6;
constexpr int kCells = kCellAxis * kCellAxis * kCellAxis; // 216
constexpr int kCellCols = kCellAxis * kCellAxis;          // 36 (ix, iy) columns
constexpr int kCellStride = kCellCols * 8;                // uint16 entries per query voxel: one 16-byte record per column =
                                                          // offsets of its cells iz = 0..5, the column end, one pad -> 576 B

struct __attribute__((aligned(4))) Vec4u { // 16 bytes at dword alignment (global loads of 128 bits only need that)
    unsigned x, y, z, w;
};

// entry i (0..7) of a column record
__device__ __forceinline__ int col_entry(const uint4 r, int i) {
    const unsigned w = (i & 4) ? ((i & 2) ? r.w : r.z) : ((i & 2) ? r.y : r.x);
    return (int)((i & 1) ? (w >> 16) : (w & 0xFFFFu));
}

__device__ __forceinline__ int cell_of(double a, double o, double inv_h) {
    const int c = (int)floor((a - o) * inv_h);
    return c < 0 ? 0 : (c > kCellAxis - 1 ? kCellAxis - 1 : c);
}

// first cell c0 of the two-cell span [c0, c0 + 1] that g leans into along one axis, and the distance from g to the open
// faces of that span (faces of the clamped outermost cells do not exist: those cells are unbounded outward)
__device__ __forceinline__ int lean_span(double g, double o, double h, double inv_h, double& rho) {
    const double u = (g - o) * inv_h;
    const double fl = floor(u);
    int c = (int)fl;
    c = c < 0 ? 0 : (c > kCellAxis - 1 ? kCellAxis - 1 : c);
    int c0 = (u - fl >= 0.5) ? c : c - 1;
    c0 = c0 < 0 ? 0 : (c0 > kCellAxis - 2 ? kCellAxis - 2 : c0);
    const double lo = (c0 == 0) ? DBL_MAX : g - (o + (double)c0 * h);
    const double hi = (c0 == kCellAxis - 2) ? DBL_MAX : (o + (double)(c0 + 2) * h) - g;
    rho = fmin(rho, fmin(lo, hi));
    return c0;
}


constexpr double kFallbackUnit = 1099511627776.0; // 2^40: slot 31 carries tested candidates + 2^40 * points that took the exact search

// ---- block reduction of N packed sums through an LDS transpose, PP values per pass (PP * kBlock doubles of LDS) ----------
// Every thread stores PP of its values column-wise, then kBlock / PP lanes per value add PP strided columns each and finish
// with DPP row operations (+ one cross-row exchange when a value owns 32 lanes).  Fixed summation order: deterministic.
// Leaves the N block sums in red[0..N) (LDS), visible to every thread on return.
template <int N, int PP>
__device__ __forceinline__ void block_reduce_to_lds(const double* v, double* buf, double* red) {
    static_assert(PP == 8 || PP == 16, "8 or 16 values per pass");
    constexpr int LPV = kBlock / PP; // lanes per value: 32 or 16
    const int tid = threadIdx.x;
    const int k = tid / LPV, seg = tid % LPV;
#pragma unroll
    for (int h = 0; h * PP < N; ++h) {
        if (h) __syncthreads();
#pragma unroll
        for (int q = 0; q < PP; ++q)
            if (h * PP + q < N) buf[q * kBlock + tid] = v[h * PP + q];
        __syncthreads();
        if (h * PP + k < N) {
            double a = 0.0;
#pragma unroll
            for (int i = 0; i < PP; ++i) a += buf[k * kBlock + i * LPV + seg];
            a += dpp_move<0x128>(a); // row_ror:8
            a += dpp_move<0x124>(a); // row_ror:4
            a += dpp_move<0x4E>(a);  // quad_perm [2,3,0,1]
            a += dpp_move<0xB1>(a);  // quad_perm [1,0,3,2]   -> every lane holds the sum of its row of 16
            if (LPV == 32) a += __shfl_xor(a, 16, 64);
            if (seg == 0) red[h * PP + k] = a;
        }
    }
    __syncthreads();
}

// ---- the 29 world-frame sums of one scan point, kept factored until the reduction needs them ---------------------------------
// add_pair_world's sums are functions of A = sum w C^-1 (3x3), b = sum A e (3), a = R p (3), the residual sum and the pair count
// (one pair: VGICP / GICP; up to seven with one common a: AVGICP).  Holding those 17 numbers and expanding EIGHT sums at a time,
// right before each pass of the block reduction writes them to LDS, keeps ~20 double registers alive instead of 32 + temporaries
// (k_accumulate_vnbr<VGICP>: 92 -> VGPRs of a 7-wave kernel).  The formulas and their operand order are add_pair_world's.
struct PairSum {
    double A[9], b[3], ax, ay, az, rsum, n, c29, c30, c31;
};
__device__ __forceinline__ void pair_sum_zero(PairSum& P) {
#pragma unroll
    for (int i = 0; i < 9; ++i) P.A[i] = 0.0;
    P.b[0] = P.b[1] = P.b[2] = 0.0;
    P.ax = P.ay = P.az = 0.0;
    P.rsum = 0.0; P.n = 0.0; P.c29 = 0.0; P.c30 = 0.0; P.c31 = 0.0;
}
// entry (i, j) of A * (-[a]x)
template <int I, int J>
__device__ __forceinline__ double pair_ab(const PairSum& P) {
    // (x y - z w as fma(x, y, -(z w)): one rounding less than add_pair_world's two products and a difference, one instruction less)
    if (J == 0) return __builtin_fma(P.A[I * 3 + 2], P.ay, -(P.A[I * 3 + 1] * P.az));
    if (J == 1) return __builtin_fma(P.A[I * 3 + 0], P.az, -(P.A[I * 3 + 2] * P.ax));
    return __builtin_fma(P.A[I * 3 + 1], P.ax, -(P.A[I * 3 + 0] * P.ay));
}
template <int K>
__device__ __forceinline__ double pair_sum_value(const PairSum& P) {
    if (K == tri(0, 0)) return P.A[0];
    if (K == tri(0, 1)) return P.A[1];
    if (K == tri(0, 2)) return P.A[2];
    if (K == tri(1, 1)) return P.A[4];
    if (K == tri(1, 2)) return P.A[5];
    if (K == tri(2, 2)) return P.A[8];
    if (K == tri(0, 3)) return pair_ab<0, 0>(P);
    if (K == tri(0, 4)) return pair_ab<0, 1>(P);
    if (K == tri(0, 5)) return pair_ab<0, 2>(P);
    if (K == tri(1, 3)) return pair_ab<1, 0>(P);
    if (K == tri(1, 4)) return pair_ab<1, 1>(P);
    if (K == tri(1, 5)) return pair_ab<1, 2>(P);
    if (K == tri(2, 3)) return pair_ab<2, 0>(P);
    if (K == tri(2, 4)) return pair_ab<2, 1>(P);
    if (K == tri(2, 5)) return pair_ab<2, 2>(P);
    if (K == tri(3, 3)) return __builtin_fma(P.ay, pair_ab<2, 0>(P), -(P.az * pair_ab<1, 0>(P)));
    if (K == tri(3, 4)) return __builtin_fma(P.ay, pair_ab<2, 1>(P), -(P.az * pair_ab<1, 1>(P)));
    if (K == tri(3, 5)) return __builtin_fma(P.ay, pair_ab<2, 2>(P), -(P.az * pair_ab<1, 2>(P)));
    if (K == tri(4, 4)) return __builtin_fma(P.az, pair_ab<0, 1>(P), -(P.ax * pair_ab<2, 1>(P)));
    if (K == tri(4, 5)) return __builtin_fma(P.az, pair_ab<0, 2>(P), -(P.ax * pair_ab<2, 2>(P)));
    if (K == tri(5, 5)) return __builtin_fma(P.ax, pair_ab<1, 2>(P), -(P.ay * pair_ab<0, 2>(P)));
    if (K == 21) return P.b[0];
    if (K == 22) return P.b[1];
    if (K == 23) return P.b[2];
    if (K == 24) return __builtin_fma(P.ay, P.b[2], -(P.az * P.b[1]));
    if (K == 25) return __builtin_fma(P.az, P.b[0], -(P.ax * P.b[2]));
    if (K == 26) return __builtin_fma(P.ax, P.b[1], -(P.ay * P.b[0]));
    if (K == 27) return P.rsum;
    if (K == 28) return P.n;
    if (K == 29) return P.c29;
    if (K == 30) return P.c30;
    if (K == 31) return P.c31;
    return 0.0;
}
template <int H, int PP, int Q, int NVAL>
struct PairPassWriter {
    static __device__ __forceinline__ void run(const PairSum& P, double* buf, int tid) {
        if (H * PP + Q < NVAL) buf[Q * kBlock + tid] = pair_sum_value<H * PP + Q>(P);
        PairPassWriter<H, PP, Q + 1, NVAL>::run(P, buf, tid);
    }
};
template <int H, int PP, int NVAL>
struct PairPassWriter<H, PP, PP, NVAL> {
    static __device__ __forceinline__ void run(const PairSum&, double*, int) {}
};
template <int PP, int H, int NVAL>
struct PairReducePass {
    static __device__ __forceinline__ void run(const PairSum& P, double* buf, double* red) {
        constexpr int LPV = kBlock / PP;
        const int tid = threadIdx.x;
        const int k = tid / LPV, seg = tid % LPV;
        if (H) __syncthreads();
        PairPassWriter<H, PP, 0, NVAL>::run(P, buf, tid);
        __syncthreads();
        if (H * PP + k < NVAL) {
            double a = 0.0;
#pragma unroll
            for (int i = 0; i < PP; ++i) a += buf[k * kBlock + i * LPV + seg];
            a += dpp_move<0x128>(a); // row_ror:8
            a += dpp_move<0x124>(a); // row_ror:4
            a += dpp_move<0x4E>(a);  // quad_perm [2,3,0,1]
            a += dpp_move<0xB1>(a);  // quad_perm [1,0,3,2]
            if (LPV == 32) a += __shfl_xor(a, 16, 64);
            if (seg == 0) red[H * PP + k] = a;
        }
        PairReducePass<PP, H + 1, NVAL>::run(P, buf, red);
    }
};
template <int PP, int NVAL>
struct PairReducePass<PP, (kSums + PP - 1) / PP, NVAL> {
    static __device__ __forceinline__ void run(const PairSum&, double*, double*) {}
};
// block_reduce_to_lds<NVAL, PP> on the factored sums (NVAL = kSums, or kSums - 3 without the work counters): same passes, same tree,
// same values; red[NVAL .. kSums) is left untouched
template <int PP, int NVAL = kSums>
__device__ __forceinline__ void block_reduce_pair_sum(const PairSum& P, double* buf, double* red) {
    PairReducePass<PP, 0, NVAL>::run(P, buf, red);
    __syncthreads();
}
// ---- the antisymmetric side sums (maps with a flagged covariance whose stored inverse is NOT symmetric) -------------------------
// A rank-deficient neighbourhood whose SVD returns U != V gives the reference a "covariance" U diag(1, 1, 1e-3) V^T that is not
// symmetric; J^T M J is then not symmetric either, JTJ.ldlt() reads its LOWER triangle and GICP's covariance output inverts the full
// matrix (reg.cpp:107-113, 136-142; vhm.hpp:141-146, 241-247).  The packed record holds the 21 entries of the UPPER triangle of the
// world-frame sum H_w = sum J_w^T A J_w (exact for any A: every entry of A is used).  What is missing is D = strict lower triangle of
// H_w - H_w^T = sum J_w^T (A - A^T) J_w.  With A - A^T = [nu]x (nu = the axial vector of A's antisymmetric part), J_w = [I | -[a]x]
// and s = a . nu:
//     J^T [nu]x J = [ [nu]x            .        ]          ([a]x [nu]x = nu a^T - s I,   [a]x [nu]x (-[a]x) = s [a]x)
//                   [ nu a^T - s I     s [a]x   ]
// -- fifteen numbers per pair, all zero for a symmetric A.  They travel in a SIDE record of 16 doubles per workgroup
// (RegParams::asym, written by every workgroup of a launch on such a map: zeros unless one of its pairs is asymmetric) and k_solve
// restores all 36 entries of H_w before the congruence with P.  Slot order: D(1,0) D(2,0) D(2,1) | D(3+i, j) row-major | D(4,3) D(5,3) D(5,4).
constexpr int kAsymSums = kAsymRecord;
// before the block reduction: does this wavefront hold a pair with an asymmetric A?  One flag per wavefront in LDS (every wavefront writes
// its own, so nothing has to be cleared); the reduction's barriers publish them.
__device__ __forceinline__ void asym_mark(const double* A, const RegParams& rp, unsigned* s_hitw) {
    if (!rp.asym) return; // (uniform: the map holds no asymmetric record)
    const bool hit = (A[7] != A[5]) || (A[2] != A[6]) || (A[3] != A[1]); // never for the compact and the symmetric stored inverses
    const unsigned long long any = __ballot(hit);
    if ((threadIdx.x & 63u) == 0u) s_hitw[threadIdx.x >> 6] = any ? 1u : 0u;
}
// after the block reduction (its last barrier has passed; `buf` is free again): the workgroup's side record
__device__ __forceinline__ void asym_side_store(const double* A, double ax, double ay, double az, unsigned L, const RegParams& rp, double* buf, double* red16,
                                                const unsigned* s_hitw) {
    if (!rp.asym) return;
    const unsigned t = threadIdx.x;
    if (s_hitw[0] | s_hitw[1] | s_hitw[2] | s_hitw[3]) { // (uniform)
        const double n1 = A[7] - A[5], n2 = A[2] - A[6], n3 = A[3] - A[1];
        const double s = (ax * n1 + ay * n2) + az * n3;
        double d[15];
        d[0] = n3; d[1] = -n2; d[2] = n1;
        d[3] = n1 * ax - s; d[4] = n1 * ay; d[5] = n1 * az;
        d[6] = n2 * ax; d[7] = n2 * ay - s; d[8] = n2 * az;
        d[9] = n3 * ax; d[10] = n3 * ay; d[11] = n3 * az - s;
        d[12] = s * az; d[13] = -(s * ay); d[14] = s * ax;
        block_reduce_to_lds<15, 8>(d, buf, red16); // two passes of eight values through the (by now free) reduction buffer
        if (t < (unsigned)kAsymSums) rp.asym[(size_t)L * kAsymSums + t] = (t < 15u) ? red16[t] : 0.0;
    } else if (t < (unsigned)kAsymSums) {
        rp.asym[(size_t)L * kAsymSums + t] = 0.0;
    }
}

// one pair into the factored form (add_pair_world's weight, threshold and residual rules)
template <int METHOD>
__device__ __forceinline__ void pair_sum_single(PairSum& P, double ex, double ey, double ez, const double* Cinv, const double* nfit, const RegParams& rp) {
    const double r2 = (ex * ex + ey * ey) + ez * ez;
    const double den = rp.th + r2;
    double w = div_normal(rp.th2, den * den); // square(th) / square(th + |r|^2)
    if (METHOD == ELM_GICP) w = w * 0.8 + 0.2;
    P.n = 1.0;
    if (METHOD == ELM_VGICP || METHOD == ELM_AVGICP) {
        if (w < 0.01) return; // reg.cpp:201 -- skipped pairs stay in the fitness denominator
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) P.A[i] = w * Cinv[i];
    P.b[0] = (P.A[0] * ex + P.A[1] * ey) + P.A[2] * ez;
    P.b[1] = (P.A[3] * ex + P.A[4] * ey) + P.A[5] * ez;
    P.b[2] = (P.A[6] * ex + P.A[7] * ey) + P.A[8] * ez;
    if (METHOD == ELM_GICP) P.rsum = fabs((ex * nfit[0] + ey * nfit[1]) + ez * nfit[2]);
    else P.rsum = sqrt_dist2(r2);
}

// pair_sum_single on a compact record: C^-1 = I + k n n^T is never formed -- A = w I + (w k) n n^T and b = w e + (w k)(n . e) n,
// fused (the same values up to the rounding of the last bit; 20 float64 operations less).  Maps whose every covariance is of the
// compact form only (DevMap::gicp_compact / vox_compact == 2): there is no full-record fallback in the kernels that use it.
template <int METHOD>
__device__ __forceinline__ void pair_sum_compact(PairSum& P, double ex, double ey, double ez, double nx, double ny, double nz, double k, const RegParams& rp) {
    const double r2 = (ex * ex + ey * ey) + ez * ez;
    const double den = rp.th + r2;
    double w = div_close(rp.th2, den * den); // square(th) / square(th + |r|^2)
    if (METHOD == ELM_GICP) w = __builtin_fma(w, 0.8, 0.2);
    P.n = 1.0;
    if (METHOD == ELM_VGICP || METHOD == ELM_AVGICP) {
        if (w < 0.01) return; // reg.cpp:201 -- skipped pairs stay in the fitness denominator
    }
    const double wk = w * k;
    const double ne = __builtin_fma(nz, ez, __builtin_fma(ny, ey, nx * ex));
    const double sn = wk * ne;
    const double ux = wk * nx, uy = wk * ny, uz = wk * nz;
    P.A[0] = __builtin_fma(ux, nx, w); P.A[1] = ux * ny; P.A[2] = ux * nz;
    P.A[3] = P.A[1]; P.A[4] = __builtin_fma(uy, ny, w); P.A[5] = uy * nz;
    P.A[6] = P.A[2]; P.A[7] = P.A[5]; P.A[8] = __builtin_fma(uz, nz, w);
    P.b[0] = __builtin_fma(sn, nx, w * ex);
    P.b[1] = __builtin_fma(sn, ny, w * ey);
    P.b[2] = __builtin_fma(sn, nz, w * ez);
    if (METHOD == ELM_GICP) P.rsum = fabs(ne); // |r_l . n_l| (reg.cpp:91-95, 128)
    else P.rsum = sqrt_dist2(r2);
}

// ---- P2P pair in 18 sums ---------------------------------------------------------------------------------------------
// AlignCloudsLocal (reg.cpp:28-51) has M = I and J = [I | -[p]x], so J^T w J and J^T w r are functions of
//   w, w p (3), w p p^T (6 unique), w r (3), w (p x r) (3), |r|, pair count            (18 sums instead of 29)
// with r = R^-1 (q - g): the reference's T^-1 q - p written on the world-frame residual e = q - g the search already holds
// (equal up to the rounding of an orthonormal R, ~1e-16 relative; |r|^2 = |e|^2 = the search's float64 distance).
constexpr int kP2PVals = 21; // 18 sums + the three work counters
__device__ __forceinline__ void pair_p2p(double* v, const double* Rinv, double px, double py, double pz, double ex, double ey, double ez,
                                         double d2, const RegParams& rp) {
    // (the rotation of e and the cross product with fused multiply-adds, the weight without the division's last correction: eleven
    // float64 instructions less per pair, the sums the same to the last bit or two)
    const double rx = __builtin_fma(Rinv[2], ez, __builtin_fma(Rinv[1], ey, Rinv[0] * ex));
    const double ry = __builtin_fma(Rinv[5], ez, __builtin_fma(Rinv[4], ey, Rinv[3] * ex));
    const double rz = __builtin_fma(Rinv[8], ez, __builtin_fma(Rinv[7], ey, Rinv[6] * ex));
    const double den = rp.th + d2;
    const double w = div_close(rp.th2, den * den); // square(th) / square(th + |r|^2)  (reg.cpp:38-39)
    const double wx = w * px, wy = w * py, wz = w * pz;
    const double ax = w * rx, ay = w * ry, az = w * rz;
    v[0] = w;
    v[1] = wx; v[2] = wy; v[3] = wz;
    v[4] = wx * px; v[5] = wx * py; v[6] = wx * pz; v[7] = wy * py; v[8] = wy * pz; v[9] = wz * pz;
    v[10] = ax; v[11] = ay; v[12] = az;
    v[13] = __builtin_fma(py, az, -(pz * ay)); v[14] = __builtin_fma(pz, ax, -(px * az)); v[15] = __builtin_fma(px, ay, -(py * ax));
    v[16] = sqrt_dist2(d2);
    v[17] = 1.0;
}
// slot k of the packed 32-sum record (21 upper JTJ, 6 JTr, residual, count, 3 counters) from the 21 reduced P2P values
__device__ __forceinline__ double p2p_expand(const double* s, int k) {
    switch (k) {
    case tri(0, 0): case tri(1, 1): case tri(2, 2): return s[0];
    case tri(0, 4): return s[3];       // -w [p]x, translation x rotation block
    case tri(0, 5): return -s[2];
    case tri(1, 3): return -s[3];
    case tri(1, 5): return s[1];
    case tri(2, 3): return s[2];
    case tri(2, 4): return -s[1];
    case tri(3, 3): return s[7] + s[9]; // w (|p|^2 I - p p^T), rotation block
    case tri(3, 4): return -s[5];
    case tri(3, 5): return -s[6];
    case tri(4, 4): return s[4] + s[9];
    case tri(4, 5): return -s[8];
    case tri(5, 5): return s[4] + s[7];
    case 21: return s[10];
    case 22: return s[11];
    case 23: return s[12];
    case 24: return s[13];
    case 25: return s[14];
    case 26: return s[15];
    case 27: return s[16];
    case 28: return s[17];
    case 29: return s[18];
    case 30: return s[19];
    case 31: return s[20];
    default: return 0.0; // tri(0,1), tri(0,2), tri(1,2), tri(0,3), tri(1,4), tri(2,5)
    }
}

// query-voxel probe of the neighbourhood-list table: linear probing, two slots per round trip (load <= 0.5)
struct QProbe {
    unsigned start, cnt, nocc;
    int qid;
};
__device__ __forceinline__ QProbe probe_query(const DevMap& m, int vx, int vy, int vz) {
    QProbe r;
    r.start = 0; r.cnt = 0; r.nocc = 0; r.qid = -1;
    unsigned h = hash3(vx, vy, vz) & m.qmask;
    for (;;) {
        const unsigned h2 = (h + 1) & m.qmask;
        const int4 key = *reinterpret_cast<const int4*>(&m.qslots[h]);
        const uint4 rg = *reinterpret_cast<const uint4*>(&m.qslots[h].start);
        const int4 key2 = *reinterpret_cast<const int4*>(&m.qslots[h2]);
        const uint4 rg2 = *reinterpret_cast<const uint4*>(&m.qslots[h2].start);
        if (key.w < 0) break;
        if (key.x == vx && key.y == vy && key.z == vz) { r.start = rg.x; r.cnt = rg.y; r.nocc = rg.z; r.qid = key.w; break; }
        if (key2.w < 0) break;
        if (key2.x == vx && key2.y == vy && key2.z == vz) { r.start = rg2.x; r.cnt = rg2.y; r.nocc = rg2.z; r.qid = key2.w; break; }
        h = (h + 2) & m.qmask;
    }
    return r;
}

// rank of a stored point's bucket in the reference's visiting order of the 27 neighbours of query voxel (vx, vy, vz):
// x-major .. z-minor (vhm.cpp:234-240); the bucket key is the truncated one (vhm.cpp:275)
__device__ __forceinline__ unsigned visit_rank(const Pt3 q, int vx, int vy, int vz, double voxel_size) {
    const int kx = (int)((double)q.x / voxel_size), ky = (int)((double)q.y / voxel_size), kz = (int)((double)q.z / voxel_size);
    return (unsigned)(((kx - vx + 1) * 3 + (ky - vy + 1)) * 3 + (kz - vz + 1));
}
// minimum / integer sum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ double row_min(double v) {
    v = fmin(v, dpp_move<0xB1>(v));  // quad_perm [1,0,3,2]
    v = fmin(v, dpp_move<0x4E>(v));  // quad_perm [2,3,0,1]
    v = fmin(v, dpp_move<0x124>(v)); // row_ror:4
    v = fmin(v, dpp_move<0x128>(v)); // row_ror:8
    return v;
}
// the same over aligned groups of LPI = 1, 2, 4, 8 or 16 lanes
template <unsigned LPI>
__device__ __forceinline__ double group_min(double v) {
    if (LPI >= 2) v = fmin(v, dpp_move<0xB1>(v));   // quad_perm [1,0,3,2]
    if (LPI >= 4) v = fmin(v, dpp_move<0x4E>(v));   // quad_perm [2,3,0,1]: quads done
    if (LPI >= 8) v = fmin(v, dpp_move<0x141>(v));  // row_half_mirror: lane i <-> 7 - i inside each half row
    if (LPI >= 16) v = fmin(v, dpp_move<0x140>(v)); // row_mirror: lane i <-> 15 - i
    return v;
}
template <unsigned LPI>
__device__ __forceinline__ unsigned group_min_u32(unsigned v) {
    if (LPI >= 2) v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false));
    if (LPI >= 4) v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false));
    if (LPI >= 8) v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false));
    if (LPI >= 16) v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false));
    return v;
}
template <unsigned LPI>
__device__ __forceinline__ int group_sum_int(int v) {
    if (LPI >= 2) v += __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);
    if (LPI >= 4) v += __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);
    if (LPI >= 8) v += __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false);
    if (LPI >= 16) v += __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false);
    return v;
}
__device__ __forceinline__ int row_sum_int(int v) {
    v += __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false);
    return v;
}

constexpr int kRedPass = 8;    // values per pass of the block reduction (16 KB of LDS per workgroup)

constexpr int kBlocksPerTrip = 2;
constexpr int kHardLanes = 4; // measured: 16 -> 56.9k, 8 -> 55.6k, 4 -> 63.0k, 2 -> 60.3k, 1 -> 55.2k registrations/s
constexpr int kCellWaves = 5; // minimum waves per SIMD: caps the kernel at 96 VGPRs (measured: 5 -> 42.0k, unconstrained 4 -> 39.4k, 6 spills -> 36.5k registrations/s)

// an undecided point handed to the workgroup-cooperative exact stage of k_accumulate_cell
struct HardRec {
    double gx, gy, gz;   // the transformed point
    unsigned start, cnt; // its candidate list
    int qid;             // its query voxel (cell offset table)
    float r2;            // upper bound of its squared nearest-neighbour distance (inf: nothing found yet)
};

// K1 (default for P2P / GICP).
//   stage 1, per lane: probe the query voxel -> the 4 column records of the 2x2x2 block of half-voxel cells the point leans
//     into -> its ~25 candidates in float32 (blocks of 4, two blocks = six 16-byte loads per round trip) -> decided when the winner
//     is clear of the runner-up by the float32 error margin AND closer than rho, the distance to the block's open faces
//     (nothing outside the block can win or tie).
//   stage 2, per workgroup: the undecided points (pose still far off, isolated points, near ties: ~3 %, ~5 % in a first
//     iteration) are compacted into LDS in thread order and served 16 at a time, 16 lanes (one DPP row) per point: the lanes
//     take the columns of the cells that intersect the ball around the point with the stage-1 distance as radius (the whole list
//     when stage 1 found nothing), walk them with the reference's float64 distances and reduce (distance, visiting rank,
//     insertion order) lexicographically -- exactly the reference's first strict minimum in its visiting order (vhm.cpp:208-243).
//   then every lane adds its pair and the workgroup reduces the packed sums.
template <int METHOD>
__global__ __launch_bounds__(kBlock, (METHOD == ELM_P2P ? kCellWaves : kCellWaves - 1)) void k_accumulate_cell(const DevMap m, const ScanDesc* __restrict__ scans, int batch,
                                                                            unsigned total_blocks, const ScanState* __restrict__ st,
                                                                            double* __restrict__ partials, const RegParams rp) {
    constexpr int NV = (METHOD == ELM_P2P) ? kP2PVals : kSums;
    __shared__ double s_buf[kRedPass * kBlock]; // stage 2: the HardRec queue; afterwards the transpose buffer of the reduction
    __shared__ double s_red[kSums];
    __shared__ int s_res[kBlock];               // stage 2 results: winning list index per queued point
    __shared__ int s_tst[kBlock];               //                  candidates walked for it
    __shared__ unsigned s_cnt[kBlock / 64];
    static_assert(sizeof(HardRec) * kBlock <= sizeof(double) * kRedPass * kBlock, "HardRec queue must fit the reduction buffer");
    const unsigned L = xcd_remap(blockIdx.x, total_blocks);
    const int s = find_scan(scans, batch, L, rp);
    const ScanState& S = st[s];
    if (S.done) return;
    const ScanDesc sd = scans[s];
    if (L >= sd.blk_end) return; // a scan whose size was only known on the device owns fewer workgroups than were launched for it
    const unsigned i = (L - sd.blk_begin) * kBlock + threadIdx.x;
    const bool valid = i < sd.n;
    double v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = 0.0;
    double px = 0.0, py = 0.0, pz = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
    QProbe qp;
    qp.start = 0; qp.cnt = 0; qp.nocc = 0; qp.qid = -1;
    int bj = -1;
    int n_tested = 0;
    float hr2 = __builtin_inff();
    bool hard = false;
    if (valid) {
        const Pt3 pf = sd.pts[i];
        px = pf.x; py = pf.y; pz = pf.z;
        gx = ((S.T[0] * px + S.T[4] * py) + S.T[8] * pz) + S.T[12];
        gy = ((S.T[1] * px + S.T[5] * py) + S.T[9] * pz) + S.T[13];
        gz = ((S.T[2] * px + S.T[6] * py) + S.T[10] * pz) + S.T[14];
        const int vx = floor_key(gx, m), vy = floor_key(gy, m), vz = floor_key(gz, m);
        qp = probe_query(m, vx, vy, vz);
        if (qp.cnt) {
            const Pt3* __restrict__ lp = m.nbr_pts + qp.start;
            const uint16_t* __restrict__ co = m.nbr_cell_off + (size_t)qp.qid * kCellStride;
            const double hc = 0.5 * m.voxel_size, inv_h = 2.0 / m.voxel_size;
            const double ox = (double)(vx - 1) * m.voxel_size, oy = (double)(vy - 1) * m.voxel_size, oz = (double)(vz - 1) * m.voxel_size;
            double rho = DBL_MAX;
            const int c0x = lean_span(gx, ox, hc, inv_h, rho), c0y = lean_span(gy, oy, hc, inv_h, rho), c0z = lean_span(gz, oz, hc, inv_h, rho);
            // the four (ix, iy) columns of the block: one contiguous range [cell c0z, cell c0z + 2) each
            int sb[4], se[4], cb[5];
            cb[0] = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint4 rec = *reinterpret_cast<const uint4*>(co + ((c0x + (k >> 1)) * kCellAxis + (c0y + (k & 1))) * 8);
                sb[k] = col_entry(rec, c0z);
                se[k] = col_entry(rec, c0z + 2);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) cb[k + 1] = cb[k] + ((se[k] - sb[k] + 3) >> 2); // blocks of 4 candidates
            const int nblk = cb[4];
            n_tested = ((se[0] - sb[0]) + (se[1] - sb[1])) + ((se[2] - sb[2]) + (se[3] - sb[3]));
            // float32 filter: g = gh + gl (float32 each, gh + gl == g to ~2^-48), so (q - gh) - gl reproduces q - g to a few float32
            // ulps of |q - g| and the float32 distance is within 2^-20 relative (+ slack / 2) of the reference's float64 one
            const float ghx = (float)gx, ghy = (float)gy, ghz = (float)gz;
            const float glx = (float)(gx - (double)ghx), gly = (float)(gy - (double)ghy), glz = (float)(gz - (double)ghz);
            float m1 = __builtin_inff(), m2 = __builtin_inff();
            int j1 = -1;
            for (int t0 = 0; t0 < nblk; t0 += 2) { // two blocks = 8 candidates per round trip
                int pp[2], pe[2];
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const int t = t0 + w;
                    int b_ = sb[3] - 4 * cb[3], e_ = se[3];
#pragma unroll
                    for (int k = 2; k >= 0; --k) {
                        const bool lt = t < cb[k + 1];
                        b_ = lt ? (sb[k] - 4 * cb[k]) : b_;
                        e_ = lt ? se[k] : e_;
                    }
                    pp[w] = (t < nblk) ? b_ + 4 * t : 0;
                    pe[w] = (t < nblk) ? e_ : 0; // empty block when past the end
                }
                // a block = 4 consecutive 12-byte candidates = 48 contiguous bytes: three 16-byte loads (dword aligned) with one
                // address computation.  Slots past the segment end hold the next cell's candidates (the arrays are padded
                // at the very end) and are masked below.
                float qf[2][12];
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const Vec4u* bp = reinterpret_cast<const Vec4u*>(lp + pp[w]);
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const Vec4u r = bp[u];
                        qf[w][4 * u] = __uint_as_float(r.x); qf[w][4 * u + 1] = __uint_as_float(r.y);
                        qf[w][4 * u + 2] = __uint_as_float(r.z); qf[w][4 * u + 3] = __uint_as_float(r.w);
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int id = pp[u >> 2] + (u & 3);
                    const float qx = qf[u >> 2][3 * (u & 3)], qy = qf[u >> 2][3 * (u & 3) + 1], qz = qf[u >> 2][3 * (u & 3) + 2];
                    const float ex = (qx - ghx) - glx, ey = (qy - ghy) - gly, ez = (qz - ghz) - glz;
                    const float dd = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
                    const float d = (id < pe[u >> 2]) ? dd : __builtin_inff();
                    m2 = fminf(m2, fmaxf(d, m1));
                    const bool c = d < m1;
                    m1 = c ? d : m1;
                    j1 = c ? id : j1;
                }
            }
            hard = true;
            if (j1 >= 0) {
                const float slack = 4e-11f * (fabsf(ghx) + fabsf(ghy) + fabsf(ghz) + 1.0f);
                const float r2 = m1 + m1 * 1.9073486328125e-06f + slack; // 2^-19: no float64 distance of a block candidate's rival is below this
                hr2 = r2;
                // sqrt(r2) * 1.000001 + 1e-6 < rho, without the square root
                const double rr = (rho - 1e-6) * 0.999999;
                if (m2 > r2 && rr > 0.0 && (double)r2 < rr * rr) {
                    bj = j1;
                    hard = false;
                }
            }
        }
    }
    // ---- stage 2: queue the undecided points in thread order
    const unsigned long long hm = __ballot(hard);
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wave] = (unsigned)__popcll(hm);
    __syncthreads();
    unsigned n_hard = 0, my_slot = 0;
#pragma unroll
    for (unsigned w = 0; w < kBlock / 64; ++w) {
        my_slot += (w < wave) ? s_cnt[w] : 0u;
        n_hard += s_cnt[w];
    }
    my_slot += (unsigned)__popcll(hm & ((1ull << lane) - 1ull));
    if (n_hard) { // uniform
        HardRec* __restrict__ s_rec = reinterpret_cast<HardRec*>(s_buf);
        if (hard) {
            HardRec r;
            r.gx = gx; r.gy = gy; r.gz = gz; r.start = qp.start; r.cnt = qp.cnt; r.qid = qp.qid; r.r2 = hr2;
            s_rec[my_slot] = r;
        }
        __syncthreads();
        const unsigned rl = threadIdx.x & 15u, row = threadIdx.x >> 4; // 16 rows of 16 lanes
        const double hc = 0.5 * m.voxel_size, inv_h = 2.0 / m.voxel_size;
        (void)hc;
        for (unsigned it = row; it < n_hard; it += kBlock / 16) {
            const HardRec R = s_rec[it];
            const int hvx = floor_key(R.gx, m), hvy = floor_key(R.gy, m), hvz = floor_key(R.gz, m);
            const Pt3* __restrict__ hp = m.nbr_pts + R.start;
            int sb = 0, se = 0;
            bool ball = false;
            if (R.r2 < __builtin_inff()) {
                // every candidate within sqrt(r2) of g -- the nearest one and whatever ties with it -- has its cell inside the
                // per-axis cell range of [g - r, g + r] (cell_of is monotonic; the lists were sorted with the same expression)
                const double r = sqrt((double)R.r2) * 1.000001 + 1e-6;
                const double ox = (double)(hvx - 1) * m.voxel_size, oy = (double)(hvy - 1) * m.voxel_size, oz = (double)(hvz - 1) * m.voxel_size;
                const int lox = cell_of(R.gx - r, ox, inv_h), hix = cell_of(R.gx + r, ox, inv_h);
                const int loy = cell_of(R.gy - r, oy, inv_h), hiy = cell_of(R.gy + r, oy, inv_h);
                const int loz = cell_of(R.gz - r, oz, inv_h), hiz = cell_of(R.gz + r, oz, inv_h);
                const int ny = hiy - loy + 1, ncol = (hix - lox + 1) * ny;
                if (ncol <= 16) {
                    ball = true;
                    if ((int)rl < ncol) {
                        const int cx = lox + (int)rl / ny, cy = loy + (int)rl % ny;
                        const uint4 rec = *reinterpret_cast<const uint4*>(m.nbr_cell_off + (size_t)R.qid * kCellStride + (cx * kCellAxis + cy) * 8);
                        sb = col_entry(rec, loz);
                        se = col_entry(rec, hiz + 1);
                    }
                }
            }
            if (!ball) { // nothing found in stage 1 (or a ball wider than 16 columns): the whole list, split 16 ways
                sb = (int)((R.cnt * rl) >> 4);
                se = (int)((R.cnt * (rl + 1u)) >> 4);
            }
            // the reference's float64 walk over this lane's share; equal distances are settled by its visiting order: bucket
            // rank (vhm.cpp:234-240), then insertion order (= global index)
            double bd = DBL_MAX;
            int bk = -1;
            unsigned brank = 0xFFFFFFFFu, bgi = 0xFFFFFFFFu;
            for (int k0 = sb; k0 < se; k0 += 4) {
                Pt3 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = hp[min(k0 + u, se - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + u;
                    const double ex = (double)q[u].x - R.gx, ey = (double)q[u].y - R.gy, ez = (double)q[u].z - R.gz;
                    const double d2 = (ex * ex + ey * ey) + ez * ez;
                    if (k >= se) continue;
                    if (d2 < bd) {
                        bd = d2; bk = k; brank = 0xFFFFFFFFu;
                    } else if (d2 == bd) {
                        if (brank == 0xFFFFFFFFu) {
                            const Pt3 b = hp[bk];
                            brank = visit_rank(b, hvx, hvy, hvz, m.voxel_size);
                            bgi = m.nbr_idx[(size_t)R.start + bk];
                        }
                        const unsigned rk = visit_rank(q[u], hvx, hvy, hvz, m.voxel_size), gi = m.nbr_idx[(size_t)R.start + k];
                        if (rk < brank || (rk == brank && gi < bgi)) { bk = k; brank = rk; bgi = gi; }
                    }
                }
            }
            const double dmin = row_min(bd);
            const unsigned at = (unsigned)((__ballot(bk >= 0 && bd == dmin) >> (16u * ((threadIdx.x >> 4) & 3u))) & 0xFFFFull);
            int win;
            if (__popc(at) <= 1) {
                win = __shfl(bk, (int)((threadIdx.x & 48u) + (unsigned)(__ffs((int)at) - 1)), 64);
                if (at == 0u) win = -1;
            } else { // the same float64 distance in several lanes: visiting order decides
                if (bk >= 0 && bd == dmin) {
                    if (brank == 0xFFFFFFFFu) {
                        const Pt3 b = hp[bk];
                        brank = visit_rank(b, hvx, hvy, hvz, m.voxel_size);
                        bgi = m.nbr_idx[(size_t)R.start + bk];
                    }
                } else {
                    brank = 0xFFFFFFFFu; bgi = 0xFFFFFFFFu; bk = -1;
                }
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) {
                    const unsigned orank = (unsigned)__shfl_xor((int)brank, off, 64), og = (unsigned)__shfl_xor((int)bgi, off, 64);
                    const int ok = __shfl_xor(bk, off, 64);
                    if (orank < brank || (orank == brank && og < bgi)) { brank = orank; bgi = og; bk = ok; }
                }
                win = bk;
            }
            const int walked = row_sum_int(se - sb);
            if (rl == 0) { s_res[it] = win; s_tst[it] = walked; }
        }
        __syncthreads();
        if (hard) { bj = s_res[my_slot]; n_tested += s_tst[my_slot]; }
        __syncthreads(); // the queue is dead: the reduction may overwrite it
    }
    if (valid) {
        // the winner's float64 distance in the reference's arithmetic (range test, weight); no bucket at all: the reference's
        // default PointStruct at the origin (vhm.cpp:37, QUIRK)
        float bx = 0.f, by = 0.f, bz = 0.f;
        int bidx = -1;
        if (bj >= 0) {
            const Pt3 q = m.nbr_pts[(size_t)qp.start + bj];
            bx = q.x; by = q.y; bz = q.z;
            bidx = (METHOD == ELM_GICP) ? (int)m.nbr_idx[(size_t)qp.start + bj] : 0;
        }
        const double ex = (double)bx - gx, ey = (double)by - gy, ez = (double)bz - gz;
        const double bd2 = (ex * ex + ey * ey) + ez * ez;
        if (METHOD == ELM_P2P) {
            if (bd2 < rp.th2) pair_p2p(v, S.Rinv, px, py, pz, ex, ey, ez, bd2, rp);
        } else {
            finish_point_pair<METHOD, true>(v, m, S, rp, px, py, pz, gx, gy, gz, bd2, bx, by, bz, bidx, m.pt_gicp);
        }
        if (rp.stats) { // (as in the grid / voxel-list kernels: 0 unless the work counters are switched on)
            v[NV - 3] = (double)qp.cnt;  // candidates of the reference's walk
            v[NV - 2] = (double)qp.nocc; // occupied neighbour voxels
            v[NV - 1] = (double)n_tested + (hard ? kFallbackUnit : 0.0); // high part: points served by stage 2
        }
    }
    block_reduce_to_lds<NV, kRedPass>(v, s_buf, s_red);
    publish_and_reduce((threadIdx.x < kSums) ? ((METHOD == ELM_P2P) ? p2p_expand(s_red, (int)threadIdx.x) : s_red[threadIdx.x]) : 0.0, L, s, sd.blk_begin,
                       sd.blk_end, partials, rp, s_buf);
}

// ---- K1g: dense cell grid (default search index for P2P / GICP) ---------------------------------------------------------
// The same two stages as k_accumulate_cell on DevMap::grid_*: the map points stored ONCE, sorted by half-voxel cell, addressed
// without a hash probe -- the query's cell coordinates ARE the address of its column's offsets -- so a point costs its scan
// record, four 12-byte offset triples and its ~25 candidates, and neighbouring queries share every line they touch.
// The reference's candidate set is not the geometric neighbourhood: a query with floor key f sees the buckets with STORED
// (truncated) keys f-1..f+1 (vhm.hpp:176-180 vs vhm.cpp:275), i.e. (f-2, f+1] voxel sizes on a negative axis.  The grid's cells
// follow the stored keys, so that set is a cell range [alo, ahi] per axis: blocks and balls are clipped to it, and a clipped
// face does not bound rho (nothing eligible lies beyond it).

// candidate k = block k / 4, slot k % 4 of the grid's structure-of-arrays blocks
__device__ __forceinline__ Pt3 blk_point(const GridBlk* __restrict__ blk, int k) {
    const float* b = reinterpret_cast<const float*>(blk + (k >> 2)) + (k & 3);
    Pt3 q;
    q.x = b[0]; q.y = b[4]; q.z = b[8];
    return q;
}

// per axis: the query's floor key f, the allowed cell range of the reference's walk, and 2 g / voxel_size (cell coordinate)
struct GridAxis {
    int f, alo, ahi;
    int cg;   // floor(t): the point's own half-voxel cell
    float fr; // t - floor(t): its position inside that cell, [0, 1)
};
__device__ __forceinline__ GridAxis grid_axis(double g, const DevMap& m) {
    GridAxis a;
    const double q = (m.inv_vs_exact != 0.0) ? g * m.inv_vs_exact : g / m.voxel_size; // == g / voxel_size bit for bit
    const double t = q + q; // exact: the cell coordinate, floor(t) in {2f, 2f+1}
    const double fl = floor(t);
    a.cg = (int)fl;
    a.fr = (float)(t - fl);
    a.f = a.cg >> 1; // floor(q) == floor(floor(2 q) / 2): PointToVoxel (vhm.hpp:176-180) from the ONE floor the cell needs anyway
    // stored keys f-1 .. f+1 -> cells: key k > 0 owns cells {2k, 2k+1}, key 0 owns {-2 .. 1}, key k < 0 owns {2k-2, 2k-1}
    const int kl = a.f - 1, kh = a.f + 1;
    a.alo = 2 * kl - ((kl <= 0) ? 2 : 0);
    a.ahi = 2 * kh + ((kh < 0) ? -1 : 1);
    return a;
}
// the (clipped) two-cell span the query leans into and the distance to its open faces IN CELL UNITS (float: the fraction of g in
// its own cell plus small integers; the 3e-8 m of rounding sit inside the 1e-6 m margin of the decision)
__device__ __forceinline__ void grid_lean(const GridAxis& a, int& blo, int& bhi, float& rho_u, int& own, float& d_other) {
    const int cg = a.cg;
    const float fr = a.fr; // position inside the own cell, [0, 1)
    const int c0 = (fr >= 0.5f) ? cg : cg - 1;
    blo = max(c0, a.alo);
    bhi = min(c0 + 1, a.ahi);
    const float dlo = (blo == a.alo) ? 3e38f : (float)(cg - blo) + fr;
    const float dhi = (bhi == a.ahi) ? 3e38f : (float)(bhi + 1 - cg) - fr;
    rho_u = fminf(rho_u, fminf(dlo, dhi));
    own = cg - blo;                                                  // the own cell is the span's first (0) or second (1) cell
    d_other = (bhi > blo) ? ((own == 0) ? 1.0f - fr : fr) : 3e18f;   // distance to the span's other cell, cell units
}

struct GridHardRec {
    double gx, gy, gz;
    float r2; // upper bound of the squared nearest-neighbour distance (inf: nothing found yet)
    float _pad;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
// a - splat(b.x) / a - splat(b.y) as ONE packed instruction: the op_sel bits broadcast one half of the second operand, so the six
// per-point scalars (gh, gl) live in three register pairs instead of six (the compiler does not fold the splat by itself)
__device__ __forceinline__ f32x2 pk_sub_lo(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_sub_hi(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// squared float32 distances of a block's four candidates to g = gh + gl (gxy = (ghx, ghy), gzl = (ghz, glx), gl2 = (gly, glz)):
// packed two-wide arithmetic (v_pk_add/mul/fma_f32)
__device__ __forceinline__ void blk_dist(const GridBlk& B, f32x2 gxy, f32x2 gzl, f32x2 gl2, f32x2& d01, f32x2& d23) {
    const f32x2 x01 = {B.x[0], B.x[1]}, x23 = {B.x[2], B.x[3]}, y01 = {B.y[0], B.y[1]}, y23 = {B.y[2], B.y[3]}, z01 = {B.z[0], B.z[1]}, z23 = {B.z[2], B.z[3]};
    const f32x2 ex01 = pk_sub_hi(pk_sub_lo(x01, gxy), gzl), ex23 = pk_sub_hi(pk_sub_lo(x23, gxy), gzl);
    const f32x2 ey01 = pk_sub_lo(pk_sub_hi(y01, gxy), gl2), ey23 = pk_sub_lo(pk_sub_hi(y23, gxy), gl2);
    const f32x2 ez01 = pk_sub_hi(pk_sub_lo(z01, gzl), gl2), ez23 = pk_sub_hi(pk_sub_lo(z23, gzl), gl2);
    d01 = __builtin_elementwise_fma(ez01, ez01, __builtin_elementwise_fma(ey01, ey01, ex01 * ex01));
    d23 = __builtin_elementwise_fma(ez23, ez23, __builtin_elementwise_fma(ey23, ey23, ex23 * ex23));
}
// the same on gh = float32(g) alone (gzz = (ghz, -)): six subtractions fewer; |g - gh| then has to be part of the caller's margin
__device__ __forceinline__ void blk_dist_h(const GridBlk& B, f32x2 gxy, f32x2 gzz, f32x2& d01, f32x2& d23) {
    const f32x2 x01 = {B.x[0], B.x[1]}, x23 = {B.x[2], B.x[3]}, y01 = {B.y[0], B.y[1]}, y23 = {B.y[2], B.y[3]}, z01 = {B.z[0], B.z[1]}, z23 = {B.z[2], B.z[3]};
    const f32x2 ex01 = pk_sub_lo(x01, gxy), ex23 = pk_sub_lo(x23, gxy);
    const f32x2 ey01 = pk_sub_hi(y01, gxy), ey23 = pk_sub_hi(y23, gxy);
    const f32x2 ez01 = pk_sub_lo(z01, gzz), ez23 = pk_sub_lo(z23, gzz);
    d01 = __builtin_elementwise_fma(ez01, ez01, __builtin_elementwise_fma(ey01, ey01, ex01 * ex01));
    d23 = __builtin_elementwise_fma(ez23, ez23, __builtin_elementwise_fma(ey23, ey23, ex23 * ex23));
}
// (m1, m2) = the two smallest candidate keys seen so far (m1 <= m2): one new distance from slot u of its block.  A key is the
// distance's bit pattern (non-negative floats order like unsigned integers) with the two lowest mantissa bits replaced by the
// slot, so the winner's slot rides along for free: and_or + med3 + min per candidate
__device__ __forceinline__ void two_smallest(float d, unsigned u, unsigned& m1, unsigned& m2) {
    const unsigned key = (__float_as_uint(d) & ~3u) | u;
    unsigned med;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(med) : "v"(m1), "v"(m2), "v"(key)); // the median of (m1 <= m2, key) is the new runner-up
    m2 = med;
    m1 = min(m1, key);
}

// The offsets of column (cx, cy) (grid-relative cell coordinates inside the grid) for the cells zlo .. zhi that the grid stores:
// e[i] .. e[i + 1] = the blocks of cell zc0 + i, i < nzc (nzc = 0: nothing stored there), e[0] .. e[nzc] = the whole run.
// Dense grid: one table over the bounding box.  Two-level grid: the column's tile first (DevMap::grid_tiles), then its own offsets.
template <int TILED>
__device__ __forceinline__ const uint32_t* col_cells(const DevMap& m, int cx, int cy, int zlo, int zhi, int& zc0, int& nzc) {
    if (!TILED) {
        zc0 = zlo;
        nzc = zhi - zlo + 1;
        return m.grid_start + (((unsigned)cx * (unsigned)m.gny + (unsigned)cy) * (unsigned)m.gnz + (unsigned)zlo);
    }
    const uint2 te = m.grid_tiles[(unsigned)(cx >> kTileShift) * (unsigned)m.gtny + (unsigned)(cy >> kTileShift)];
    const int z0 = (int)(te.y & 0xFFFFu), nz = (int)(te.y >> 16);
    const int lo = min(max(zlo - z0, 0), nz), hi = min(max(zhi + 1 - z0, lo), nz);
    zc0 = z0 + lo;
    nzc = hi - lo;
    return m.grid_start + te.x + (unsigned)(((cx & (kTile - 1)) << kTileShift) | (cy & (kTile - 1))) * (unsigned)(nz + 1) + (unsigned)lo;
}

constexpr int kGridWaves = 8; // minimum waves per SIMD of the P2P kernel = a 64-VGPR cap.  Round 4, after the work counters left the production kernels (2 spilled
                         // VGPRs, 26 spilled SGPRs at the cap): 6 -> 89.0 k, 7 -> 94.0-94.3 k, 8 -> 95.8-96.4 k registrations/s (hard guesses 20.5 -> 20.8 k);
                         // round 2, with the counters: 5 -> 59.8k, 6 -> 63.9k, 7 -> 65.6k, 8 (35 spills) -> 54.2k.  profiles/r04_sweep.txt
constexpr int kGicpWaves = 7; // GICP: 5 (the old cap; the kernel used 72 VGPRs anyway) -> 71.2-71.6 k, 7 (one spill) -> 72.5-73.3 k, 8 (8 spills) -> 67.2 k
// The exact search of ONE undecided point by its group of LPI lanes (stage 2 of k_accumulate_grid): the ball of radius sqrt(R.r2) around g (seeded first when stage 1 found nothing) intersected with the reference's allowed cell
// range and the grid, float32 keys first, the reference's float64 distances and visiting order on a near tie.  win: block * 4 + slot of the
// nearest neighbour (-1: none), the same value in every lane of the group; walked: candidate slots this lane tested (instrumented builds).
template <int TILED, unsigned LPI>
__device__ __forceinline__ void grid_ball_walk(const DevMap& m, const GridBlk* __restrict__ lp, const GridHardRec& R, const bool live, const unsigned rl,
                                               const unsigned lane, int& win_out, int& walked_out) {
    const GridAxis ax = grid_axis(R.gx, m), ay = grid_axis(R.gy, m), az = grid_axis(R.gz, m);
    int lox = ax.alo, hix = ax.ahi, loy = ay.alo, hiy = ay.ahi, loz = az.alo, hiz = az.ahi;
    int seeded = 0;
    float r2s = R.r2;
    // A point whose stage-1 block came back EMPTY (a third of the undecided points under a poor initial guess: the surface is
    // two cells below the point) has no ball: it would walk all 36 columns of its 27 voxels, ~240 candidates.  Seed it first:
    // the group's lanes probe the 2 x 2 columns nearest to the point over the whole allowed z-range; the nearest candidate found
    // there bounds the nearest neighbour, and the walk below is confined to that ball like any other undecided point's.
    if (__any(live && !(r2s < __builtin_inff()))) { // wave-uniform
        float best = __builtin_inff();
        if (live && !(r2s < __builtin_inff())) {
            const double inv_h = 2.0 / m.voxel_size;
            const int cx0 = (int)floor(R.gx * inv_h - 0.5), cy0 = (int)floor(R.gy * inv_h - 0.5);
            const float shx = (float)R.gx, shy = (float)R.gy, shz = (float)R.gz;
            const f32x2 sxy = {shx, shy}, szl = {shz, (float)(R.gx - (double)shx)}, sl2 = {(float)(R.gy - (double)shy), (float)(R.gz - (double)shz)};
            const int zlo = max(az.alo - m.gz0, 0), zhi = min(az.ahi - m.gz0, m.gnz - 1);
            for (unsigned q = rl; q < 4u; q += LPI) {
                const int cxa = cx0 + (int)(q & 1u), cya = cy0 + (int)(q >> 1);
                const int cx = cxa - m.gx0, cy = cya - m.gy0;
                if (cxa < ax.alo || cxa > ax.ahi || cya < ay.alo || cya > ay.ahi || cx < 0 || cx >= m.gnx || cy < 0 || cy >= m.gny || zlo > zhi) continue;
                int zc0, nzc;
                const uint32_t* e = col_cells<TILED>(m, cx, cy, zlo, zhi, zc0, nzc);
                const int b0 = (int)e[0], b1 = (int)e[nzc];
                seeded += 4 * (b1 - b0);
                for (int b = b0; b < b1; ++b) {
                    f32x2 da, db;
                    blk_dist(lp[b], sxy, szl, sl2, da, db);
                    best = fminf(best, fminf(fminf(da.x, da.y), fminf(db.x, db.y)));
                }
            }
        }
        best = __uint_as_float(group_min_u32<LPI>(__float_as_uint(best))); // non-negative floats order like their bit patterns
        if (!(r2s < __builtin_inff()) && best < 1e30f) // (padding slots sit at 1e18: their squares are not candidates)
            r2s = best + best * 4e-6f + 4e-11f * (fabsf((float)R.gx) + fabsf((float)R.gy) + fabsf((float)R.gz) + 1.0f);
    }
    if (r2s < __builtin_inff()) {
        // every candidate within sqrt(r2) of g -- the nearest one and whatever ties with it -- has its cell inside the
        // per-axis cell range of [g - r, g + r] (1e-6 of margin: a stored coordinate exactly on a cell face counts to the
        // cell further from zero, grid_cell_of)
        const double r = (double)(__builtin_sqrtf(r2s) * 1.000001f) + 1e-6; // float32 root (1 ulp) inside the margin
        const double inv_h = 2.0 / m.voxel_size;
        lox = max(lox, (int)floor((R.gx - r) * inv_h)); hix = min(hix, (int)floor((R.gx + r) * inv_h));
        loy = max(loy, (int)floor((R.gy - r) * inv_h)); hiy = min(hiy, (int)floor((R.gy + r) * inv_h));
        loz = max(loz, (int)floor((R.gz - r) * inv_h)); hiz = min(hiz, (int)floor((R.gz + r) * inv_h));
    }
    lox = max(lox - m.gx0, 0); hix = min(hix - m.gx0, m.gnx - 1);
    loy = max(loy - m.gy0, 0); hiy = min(hiy - m.gy0, m.gny - 1);
    loz = max(loz - m.gz0, 0); hiz = min(hiz - m.gz0, m.gnz - 1);
    const int nx = hix - lox + 1, ny = hiy - loy + 1, nz = hiz - loz + 1;
    const int ncol = (live && nx > 0 && ny > 0 && nz > 0) ? nx * ny : 0;
    // float32 pass over this lane's columns first (the arithmetic of stage 1): a winner that leads the runner-up of the
    // whole ball by the margin is the float64 winner as well
    const unsigned gsh = threadIdx.x & 63u & ~(LPI - 1u);
    const unsigned long long gmask = ((1ull << LPI) - 1ull) << gsh;
    int win = -1, walked = seeded;
    bool need64 = false;
    {
        // distances to gh = float32(g) alone (six packed subtractions fewer per block, as in stage 1): an exact distance to g differs
        // from the one to gh by at most eg = |g - gh|_1 in the ROOT, which the decision below pays for
        const float ghx = (float)R.gx, ghy = (float)R.gy, ghz = (float)R.gz;
        const float glx = (float)(R.gx - (double)ghx), gly = (float)(R.gy - (double)ghy), glz = (float)(R.gz - (double)ghz);
        const f32x2 gxy = {ghx, ghy}, gzl = {ghz, glx}, gl2 = {gly, glz};
        unsigned m1 = 0x7F800000u, m2 = 0x7F800000u;
        int jb = 0;
        const float rny = __builtin_amdgcn_rcpf((float)ny); // c / ny for the few dozen columns of a ball: exact via float32
        // software-pipelined walk: the offsets of this lane's NEXT column are requested before the current column's blocks
        // are walked, and block b + 1 before block b is evaluated -- the walk is a chain of dependent round trips (offsets ->
        // blocks, column after column) that the other wavefronts only partly hide when many points are undecided
        auto col_run = [&](int c, int& r0, int& r1) {
            const int qx = (int)(((float)c + 0.5f) * rny);
            const int cx = lox + qx, cy = loy + (c - qx * ny);
            int zc0, nzc;
            const uint32_t* e = col_cells<TILED>(m, cx, cy, loz, hiz, zc0, nzc);
            r0 = (int)e[0]; r1 = (int)e[nzc];
        };
        int c = (int)rl, nb0 = 0, nb1 = 0;
        if (c < ncol) col_run(c, nb0, nb1);
        while (c < ncol) {
            const int b0 = nb0, b1 = nb1;
            c += (int)LPI;
            if (c < ncol) col_run(c, nb0, nb1);
            walked += 4 * (b1 - b0);
            GridBlk Bn = lp[(b0 < b1) ? b0 : 0];
            for (int b = b0; b < b1; ++b) {
                const GridBlk B = Bn;
                Bn = lp[(b + 1 < b1) ? b + 1 : 0]; // (block 0: the padding block, always resident)
                f32x2 da, db;
                blk_dist(B, gxy, gzl, gl2, da, db);
                const unsigned was = m1;
                two_smallest(da.x, 0u, m1, m2);
                two_smallest(da.y, 1u, m1, m2);
                two_smallest(db.x, 2u, m1, m2);
                two_smallest(db.y, 3u, m1, m2);
                jb = (m1 != was) ? b : jb;
            }
        }
        const unsigned m1g = group_min_u32<LPI>(m1);
        const unsigned long long holders = __ballot(m1 == m1g) & gmask;
        const unsigned hl = (unsigned)__ffsll((long long)holders) - 1u; // first lane of the group that holds the minimum
        const unsigned m2g = group_min_u32<LPI>((lane == hl) ? m2 : m1); // a second holder of the same key counts as a tie
        const int jw = __shfl(jb * 4 + (int)(m1 & 3u), (int)hl, 64);
        const float d1 = __uint_as_float(m1g & ~3u), d2 = __uint_as_float(m2g & ~3u);
        // clear float32 winner (2^-18: float32 arithmetic + key bits, see stage 1; the distances are to g itself -- gh + the low parts --
        // so only the rounding of the low parts is left for the slack; padding slots at 1e36 never win)
        const float slack = 4e-11f * (fabsf(ghx) + fabsf(ghy) + fabsf(ghz) + 1.0f);
        if (jw >= 4 && d2 > d1 + d1 * 3.814697265625e-06f + slack) win = jw; // (2^-18 on one side covers both, as in stage 1)
        else need64 = live && jw >= 4;                                         // near tie: the float64 walk below decides
    }
    if (__any(need64)) { // wave-uniform; practically never taken
        // near tie in float32: the reference's float64 arithmetic decides.  First the float64 minimum over the ball, then,
        // among the candidates that meet it (usually one), the one the reference meets first -- bucket visiting rank
        // (vhm.cpp:234-240: x-major .. z-minor over the stored keys f-1..f+1), then insertion order (= bucket-order index).
        // The bucket of a cell: c >= 2 -> c >> 1, -2 <= c <= 1 -> 0, c <= -3 -> (c + 2) >> 1.
        const int ncol64 = need64 ? ncol : 0;
        double bd = DBL_MAX;
#pragma unroll 1
        for (int c = (int)rl; c < ncol64; c += (int)LPI) {
            const int cx = lox + c / ny, cy = loy + c % ny;
            int zc0, nzc;
            const uint32_t* e = col_cells<TILED>(m, cx, cy, loz, hiz, zc0, nzc);
#pragma unroll 1
            for (int k = 4 * (int)e[0]; k < 4 * (int)e[nzc]; ++k) {
                const Pt3 q = blk_point(lp, k);
                const double ex = (double)q.x - R.gx, ey = (double)q.y - R.gy, ez = (double)q.z - R.gz;
                bd = fmin((ex * ex + ey * ey) + ez * ez, bd);
            }
        }
        const double dmin = group_min<LPI>(bd);
        unsigned brank = 0xFFFFFFFFu, bgi = 0xFFFFFFFFu;
        int bk = -1;
#pragma unroll 1
        for (int c = (int)rl; c < ncol64; c += (int)LPI) {
            const int cx = lox + c / ny, cy = loy + c % ny;
            const int ccx = cx + m.gx0, ccy = cy + m.gy0;
            const int kx = ccx >= 2 ? ccx >> 1 : (ccx >= -2 ? 0 : (ccx + 2) >> 1), ky = ccy >= 2 ? ccy >> 1 : (ccy >= -2 ? 0 : (ccy + 2) >> 1);
            int zc0, nzc;
            const uint32_t* e = col_cells<TILED>(m, cx, cy, loz, hiz, zc0, nzc);
#pragma unroll 1
            for (int z = 0; z < nzc; ++z) {
                const int ccz = zc0 + z + m.gz0;
                const int kz = ccz >= 2 ? ccz >> 1 : (ccz >= -2 ? 0 : (ccz + 2) >> 1);
                const unsigned rank = (unsigned)(((kx - ax.f + 1) * 3 + (ky - ay.f + 1)) * 3 + (kz - az.f + 1));
#pragma unroll 1
                for (int k = 4 * (int)e[z]; k < 4 * (int)e[z + 1]; ++k) {
                    const Pt3 q = blk_point(lp, k);
                    const double ex = (double)q.x - R.gx, ey = (double)q.y - R.gy, ez = (double)q.z - R.gz;
                    if ((ex * ex + ey * ey) + ez * ez != dmin) continue;
                    const unsigned gi = m.grid_idx[k];
                    if (rank < brank || (rank == brank && gi < bgi)) { brank = rank; bgi = gi; bk = k; }
                }
            }
        }
#pragma unroll
        for (int off = (int)LPI / 2; off > 0; off >>= 1) {
            const unsigned orank = (unsigned)__shfl_xor((int)brank, off, 64), og = (unsigned)__shfl_xor((int)bgi, off, 64);
            const int ok = __shfl_xor(bk, off, 64);
            if (orank < brank || (orank == brank && og < bgi)) { brank = orank; bgi = og; bk = ok; }
        }
        win = need64 ? bk : win;
    }
    win_out = win;
    walked_out = walked;
}

// STATS = 1 (elm_ctx_set_work_counters): the launch also sums the three work counters (candidates / occupied buckets of the reference's
// walk from the dense statistics box, candidates this kernel tested + points served by stage 2).  The production launches run with
// STATS = 0: no statistics load, 18 (P2P) / 29 reduced values, no per-point bookkeeping in stage 2.
// WIDE = 1: the block array does not fit 32-bit byte offsets (4 GB = ~275 M map points): stage 1 carries offsets in 16-byte units
// (three per block) and forms the 64-bit address per block with one shift-add; everything else addresses blocks by index already.
template <int METHOD, int COMPACT, int TILED, int STATS, int WIDE>
__global__ __launch_bounds__(kBlock, (METHOD == ELM_P2P ? (STATS ? 7 : kGridWaves) : kGicpWaves)) void k_accumulate_grid( // (the instrumented P2P build holds 20.7 KB of LDS: 7 workgroups per CU)
       const DevMap m, const ScanDesc* __restrict__ scans, int batch,
                                                                            unsigned total_blocks, const ScanState* __restrict__ st,
                                                                            double* __restrict__ partials, const RegParams rp) {
    constexpr int kStats = (STATS == 1) ? 1 : 0;   // the instrumented build (work counters)
    constexpr bool QUERY = STATS == 2;         // elm_map_get_correspondences: the search alone, on float64 GLOBAL-frame points (RegParams::query)
    constexpr int NV = (METHOD == ELM_P2P) ? (kStats ? kP2PVals : kP2PVals - 3) : kSums;
    __shared__ double s_buf[kRedPass * kBlock]; // stage 2: the queue of undecided points; afterwards the reduction's transpose buffer
    __shared__ double s_red[kSums];
    __shared__ int s_res[kBlock];
    __shared__ int s_tst[kStats ? kBlock : 1];
    __shared__ float s_pz[METHOD == ELM_P2P ? kBlock : 1];  // P2P: the point's z (x and y ride in the stash's spare 8 bytes)
    __shared__ unsigned s_st[kStats ? kBlock : 1];           // instrumented builds: the walk statistics of the query voxel
    __shared__ unsigned s_cnt[kBlock / 64];
    const unsigned L = xcd_remap(blockIdx.x, total_blocks);
    const int s = find_scan(scans, batch, L, rp);
    const ScanState& S = st[s];
    if (S.done) return;
    const ScanDesc sd = scans[s];
    if (L >= sd.blk_end) return; // a scan whose size was only known on the device owns fewer workgroups than were launched for it
    const unsigned i = (L - sd.blk_begin) * kBlock + threadIdx.x;
    const bool valid = i < sd.n;
    double v[(METHOD == ELM_P2P) ? NV : 1]; // P2P: its 18 sums + 3 counters; GICP: the factored form P below
#pragma unroll
    for (int k = 0; k < ((METHOD == ELM_P2P) ? NV : 1); ++k) v[k] = 0.0;
    PairSum P;
    if (METHOD != ELM_P2P) pair_sum_zero(P);
    int bj = -1; // winning candidate: block * 4 + slot
    int n_tested = 0;
    float hr2 = __builtin_inff();
    bool hard = false;
    const double h = 0.5 * m.voxel_size;
    const GridBlk* __restrict__ lp = m.grid_blk;
    constexpr unsigned kBlkStep = WIDE ? (unsigned)(sizeof(GridBlk) / 16) : (unsigned)sizeof(GridBlk); // stage-1 offsets: 16-byte units / bytes
    // The transformed point g (three doubles) and the walk statistics of its query voxel are NOT kept in registers across the candidate
    // loop and the cooperative stage (the kernel lives on occupancy): they wait in LDS, in the upper half of the reduction buffer (the queue
    // of stage 2 takes at most the lower half), in the 32 bytes this thread's own wavefront overwrites first in the reduction (values 4..7
    // of the first pass): no barrier is needed between the last read of the stash and the reduction.  (Rounds 2-3 stashed the point and
    // redid the 18-operation float64 transform at both later uses -- in the epilogue and, for a wavefront with an undecided point, before
    // stage 2: 36 half-rate instructions per point.  The P2P pair also needs the point itself: x and y as floats in the stash's last 8
    // bytes, z in a 1 KB array of its own -- re-reading it from global memory at the epilogue cost 4.7 %.)
    // (round 5) the thread's OWN slots of values 4..7 of the reduction's first pass: doubles (4 + j) * kBlock + tid -- the address is
    // tid * 8 plus immediate offsets (two ds_write2st64_b64 / ds_read2st64_b64), and the thread itself overwrites them first
    auto stash_w = [&](double a, double b, double c, double d) {
        double* p = s_buf + threadIdx.x;
        p[4 * kBlock] = a; p[5 * kBlock] = b; p[6 * kBlock] = c; p[7 * kBlock] = d;
    };
    auto load_g = [&](double& gx, double& gy, double& gz, float& pxf, float& pyf) {
        const double* p = s_buf + threadIdx.x;
        gx = p[4 * kBlock]; gy = p[5 * kBlock]; gz = p[6 * kBlock];
        const double w = p[7 * kBlock];
        pxf = __int_as_float(__double2loint(w)); pyf = __int_as_float(__double2hiint(w));
    };
    auto transform = [&](const float4 pf, double& px, double& py, double& pz, double& gx, double& gy, double& gz) {
        px = pf.x; py = pf.y; pz = pf.z;
        gx = ((S.T[0] * px + S.T[4] * py) + S.T[8] * pz) + S.T[12]; // g = T * [p, 1] (reg.hpp:141-146), the reference's association
        gy = ((S.T[1] * px + S.T[5] * py) + S.T[9] * pz) + S.T[13];
        gz = ((S.T[2] * px + S.T[6] * py) + S.T[10] * pz) + S.T[14];
    };
    if (valid) {
        double px, py, pz, gx, gy, gz;
        float4 pf = make_float4(0.f, 0.f, 0.f, 0.f);
        if (QUERY) { // the point as the caller holds it: already in the map's frame
            gx = rp.query[3 * (size_t)i]; gy = rp.query[3 * (size_t)i + 1]; gz = rp.query[3 * (size_t)i + 2];
            px = py = pz = 0.0;
        } else {
            const Pt3 p3 = sd.pts[i]; // 12 bytes per point: one global_load_dwordx3
            pf = make_float4(p3.x, p3.y, p3.z, 0.f);
            transform(pf, px, py, pz, gx, gy, gz);
        }
        const GridAxis ax = grid_axis(gx, m), ay = grid_axis(gy, m), az = grid_axis(gz, m);
        // statistics of the reference's walk for this query voxel: candidates and occupied buckets among the 27
        unsigned stat = 0;
        if (kStats && !TILED) { // (the two-level grid keeps no dense statistics box: its work counters read 0)
            const int ux = ax.f - m.vx0, uy = ay.f - m.vy0, uz = az.f - m.vz0;
            const bool in_box = (unsigned)ux < (unsigned)m.vnx && (unsigned)uy < (unsigned)m.vny && (unsigned)uz < (unsigned)m.vnz;
            const unsigned sidx = in_box ? ((unsigned)ux * (unsigned)m.vny + (unsigned)uy) * (unsigned)m.vnz + (unsigned)uz : 0u;
            stat = m.vox_stat[sidx];
            stat = in_box ? stat : 0u;
        }
        float rho_u = 3e38f; // distance to the block's open faces, cell units
        int bx0, bx1, by0, by1, bz0, bz1;
        int ox, oy, oz;
        float dxo, dyo, dzo;
        grid_lean(ax, bx0, bx1, rho_u, ox, dxo);
        grid_lean(ay, by0, by1, rho_u, oy, dyo);
        grid_lean(az, bz0, bz1, rho_u, oz, dzo);
        (void)oz; (void)dzo;
        // block cells relative to the grid; a block that leaves the grid (or came out empty) goes to stage 2, which clamps
        const int rx0 = bx0 - m.gx0, rx1 = bx1 - m.gx0, ry0 = by0 - m.gy0, ry1 = by1 - m.gy0, rz0 = bz0 - m.gz0, rz1 = bz1 - m.gz0;
        const bool inside = rx0 >= 0 && rx1 < m.gnx && rx0 <= rx1 && ry0 >= 0 && ry1 < m.gny && ry0 <= ry1 && rz0 >= 0 && rz1 < m.gnz && rz0 <= rz1;
        // the four (ix, iy) columns of the block: one contiguous run of candidate blocks [cell bz0, cell bz1] each
        unsigned sb[4]; // byte offsets modulo 2^32 (build_cell_grid keeps the block array below 4 GB)
        int cb[5];
        cb[0] = 0;
        // all four 12-byte loads are issued before the first is used (a column that is clipped away or outside reads cell 0 and is
        // masked afterwards); cell indices fit 32 bits (build_cell_grid)
        // The columns are requested in VISITING order straight away -- own, the nearer of the x / y neighbour, the other, the diagonal one
        // (lower bounds of the squared distance from the point's position in its cell: 1e-6 m off each face distance for the float32
        // cell coordinate) -- so nothing has to be permuted once the offsets are back.  ox / oy: the own cell is the span's first (0) or
        // second (1) cell; a span clipped to one cell has no neighbour on that axis.
        const float ex_ = fmaxf(dxo * (float)h - 1e-6f, 0.f), ey_ = fmaxf(dyo * (float)h - 1e-6f, 0.f);
        const float Bx = fminf(ex_ * ex_, 1e36f), By = fminf(ey_ * ey_, 1e36f);
        const bool xfirst = Bx <= By, has_x = rx1 != rx0, has_y = ry1 != ry0;
        unsigned s0[4], s1[4], s2[4];
        bool ok[4];
        ok[0] = inside;
        ok[1] = inside && (xfirst ? has_x : has_y);
        ok[2] = inside && (xfirst ? has_y : has_x);
        ok[3] = inside && has_x && has_y;
        if (!TILED) {
            // cell index of the own column, the neighbours one column step away (x: gny * gnz cells, y: gnz), towards the other cell of the span
            const int own_x = ox ? rx1 : rx0, own_y = oy ? ry1 : ry0;
            const unsigned base = ((unsigned)own_x * (unsigned)m.gny + (unsigned)own_y) * (unsigned)m.gnz + (unsigned)rz0;
            const unsigned xs = (unsigned)m.gny * (unsigned)m.gnz, ys = (unsigned)m.gnz; // (scalar)
            const unsigned cx_n = ox ? base - xs : base + xs, cy_n = oy ? base - ys : base + ys, cd = ox ? cy_n - xs : cy_n + xs;
            unsigned cellv[4];
            cellv[0] = base; cellv[1] = xfirst ? cx_n : cy_n; cellv[2] = xfirst ? cy_n : cx_n; cellv[3] = cd;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t* e = m.grid_start + (ok[k] ? cellv[k] : 0u);
                s0[k] = e[0]; s1[k] = e[1]; s2[k] = e[2];
            }
        } else { // the column's tile, then the two ends of its run (a masked column reads tile 0 / entry 0)
            const int own_x = ox ? rx1 : rx0, oth_x = ox ? rx0 : rx1, own_y = oy ? ry1 : ry0, oth_y = oy ? ry0 : ry1;
            int cxv[4], cyv[4];
            cxv[0] = own_x; cyv[0] = own_y;
            cxv[1] = xfirst ? oth_x : own_x; cyv[1] = xfirst ? own_y : oth_y;
            cxv[2] = xfirst ? own_x : oth_x; cyv[2] = xfirst ? oth_y : own_y;
            cxv[3] = oth_x; cyv[3] = oth_y;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int zc0, nzc;
                const uint32_t* e = col_cells<1>(m, ok[k] ? cxv[k] : 0, ok[k] ? cyv[k] : 0, rz0, rz1, zc0, nzc);
                s0[k] = e[0]; s1[k] = e[nzc]; s2[k] = s1[k];
            }
        }
        // everything that does not need the loads goes HERE, in their shadow (the asm is a scheduling barrier: left alone the
        // compiler waits for the offsets first and does this arithmetic on the critical path -- 7 % of the kernel)
        float rr = (rho_u < 1e30f) ? (rho_u * (float)h - 1.1e-6f) * 0.999998f : 1e18f;
        float rr2 = (rr > 0.f) ? rr * rr * 0.999999f : -1.f;
        // float32 filter on gh = float32(g): a float32 distance to gh is within 2^-20 relative of the exact one, and exact distances
        // to gh and to g differ by at most eg = |g - gh|_1.  Everything the decision compares lies below rr (a winner at rr or
        // beyond is undecided anyway), so (sqrt(d) + eg)^2 <= d + egrr with egrr = 2 eg rr + eg^2: margins without a root.
        float ghx = (float)gx, ghy = (float)gy, ghz = (float)gz;
        // (|g - gh| <= half an ulp of gh per axis = 2^-24 |gh|: the bound instead of the three float64 differences)
        const float eg = (fabsf(ghx) + fabsf(ghy) + fabsf(ghz)) * 5.9604652e-08f;
        float egrr = (2.0f * eg * fmaxf(rr, 0.f) + eg * eg) * 1.000001f;
        // for the ball of an undecided point: any block candidate is within 3.5 h of g
        float egblk = (7.0f * eg * (float)h + eg * eg) * 1.000001f;
        // a lane stops at the first column (in visiting order) that lies farther than its current winner
        float L1 = fminf(Bx, By), L2 = fmaxf(Bx, By), L3 = (Bx + By) * 0.999999f;
        asm volatile("" : "+v"(rr2), "+v"(ghx), "+v"(ghy), "+v"(ghz), "+v"(egrr), "+v"(egblk), "+v"(L1), "+v"(L2), "+v"(L3));
        stash_w(gx, gy, gz, __hiloint2double(__float_as_int(pf.y), __float_as_int(pf.x)));
        if (METHOD == ELM_P2P) s_pz[threadIdx.x] = pf.z;
        if (kStats) s_st[threadIdx.x] = stat;
        {
            int b0v[4], b1v[4]; // (already in visiting order)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                b0v[k] = ok[k] ? (int)s0[k] : 0;
                b1v[k] = ok[k] ? (int)((rz1 > rz0 || TILED) ? s2[k] : s1[k]) : 0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                sb[k] = (unsigned)(b0v[k] - cb[k]) * kBlkStep; // block t of the flattened sequence lives at byte (unit) sb[k] + 48 (3) t for cb[k] <= t < cb[k + 1]
                cb[k + 1] = cb[k] + (b1v[k] - b0v[k]);
            }
        }
        {
            int nblk = cb[4];
            const f32x2 gxy = {ghx, ghy}, gzz = {ghz, 0.f};
            unsigned m1 = 0x7F800000u, m2 = 0x7F800000u; // +inf
            unsigned jb = 0;                             // byte offset of m1's block (block 0 = padding = none yet)
            for (int t0 = 0; t0 < nblk; t0 += kBlocksPerTrip) { // kBlocksPerTrip blocks (three 16-byte loads each) per round trip
                unsigned pb[kBlocksPerTrip]; // byte offsets (32-bit: the loads take the scalar base + this lane's offset)
#pragma unroll
                for (int w = 0; w < kBlocksPerTrip; ++w) {
                    const int t = t0 + w;
                    unsigned b_ = sb[3];
#pragma unroll
                    for (int k = 2; k >= 0; --k) b_ = (t < cb[k + 1]) ? sb[k] : b_;
                    pb[w] = (t < nblk) ? b_ + (unsigned)t * kBlkStep : 0u; // past the end: block 0, four padding slots
                }
                GridBlk B[kBlocksPerTrip];
#pragma unroll
                for (int w = 0; w < kBlocksPerTrip; ++w)
                    B[w] = *reinterpret_cast<const GridBlk*>(reinterpret_cast<const char*>(lp) + (WIDE ? ((size_t)pb[w] << 4) : (size_t)pb[w]));
                __builtin_amdgcn_sched_barrier(0); // all six loads are in flight before the first is waited for (the scheduler otherwise
                                                   // sometimes starts on the first block between the two blocks' loads)
#pragma unroll
                for (int w = 0; w < kBlocksPerTrip; ++w) {
                    f32x2 da, db;
                    blk_dist_h(B[w], gxy, gzz, da, db);
                    const unsigned was = m1;
                    two_smallest(da.x, 0u, m1, m2);
                    two_smallest(da.y, 1u, m1, m2);
                    two_smallest(db.x, 2u, m1, m2);
                    two_smallest(db.y, 3u, m1, m2);
                    jb = (m1 != was) ? pb[w] : jb;
                }
                // the first column that lies beyond the current winner (2^-17 relative: outside the margins of the decision below)
                // ends this lane's sequence: it and the columns after it cannot win or tie
                const float dbest = __uint_as_float(m1 & ~3u);
                const float best = (dbest < rr2) ? (dbest + dbest * 7.62939453125e-06f + 2.0f * egrr) * 1.000001f : __builtin_inff();
                nblk = (L1 > best) ? cb[1] : ((L2 > best) ? cb[2] : ((L3 > best) ? cb[3] : nblk));
            }
            n_tested = 4 * nblk;
            hard = true;
            if (jb > 0) { // a real candidate (block 0 is padding)
                // the keys drop two mantissa bits (< 3.6e-7 relative, downwards) on top of the float32 distance's 2^-20: 2^-18 covers
                // both sides of the comparison
                const float d1 = __uint_as_float(m1 & ~3u), d2 = __uint_as_float(m2 & ~3u);
                const float r2 = d1 + d1 * 3.814697265625e-06f + egrr; // >= the winner's exact squared distance when it lies below rr
                hr2 = d1 + d1 * 3.814697265625e-06f + egblk;           // the same bound for any block candidate: stage 2's ball
                if (d2 - d2 * 3.814697265625e-06f > r2 + egrr && r2 < rr2) {
                    bj = (int)(jb / kBlkStep) * 4 + (int)(m1 & 3u);
                    hard = false;
                }
            }
        }
    }
    // ---- stage 2: queue the undecided points in thread order, kHardLanes lanes per point
    const unsigned long long hm = __ballot(hard);
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wave] = (unsigned)__popcll(hm);
    __syncthreads();
    unsigned n_hard = 0, my_slot = 0;
#pragma unroll
    for (unsigned w = 0; w < kBlock / 64; ++w) {
        my_slot += (w < wave) ? s_cnt[w] : 0u;
        n_hard += s_cnt[w];
    }
    my_slot += (unsigned)__popcll(hm & ((1ull << lane) - 1ull));
    if (n_hard) { // uniform
        GridHardRec* __restrict__ s_rec = reinterpret_cast<GridHardRec*>(s_buf);
        if (hard) {
            GridHardRec r;
            float pxf_, pyf_;
            load_g(r.gx, r.gy, r.gz, pxf_, pyf_);
            r.r2 = hr2; r._pad = 0.f;
            s_rec[my_slot] = r;
        }
        __syncthreads();
        constexpr unsigned LPI = kHardLanes; // lanes per undecided point (a power of two <= 16: one DPP row holds 16 / LPI points)
        const unsigned rl = threadIdx.x & (LPI - 1u), row = threadIdx.x / LPI;
        for (unsigned it0 = 0; it0 < n_hard; it0 += kBlock / LPI) {
            const unsigned it = it0 + row;
            if (it0 + (threadIdx.x & ~63u) / LPI >= n_hard) break; // wave-uniform: this wavefront has no point in this pass
            const bool live = it < n_hard;
            const GridHardRec R = s_rec[live ? it : 0];
            int win, walked;
            grid_ball_walk<TILED, LPI>(m, lp, R, live, rl, lane, win, walked);
            if (kStats) walked = group_sum_int<LPI>(walked);
            if (rl == 0 && live) {
                s_res[it] = win;
                if (kStats) s_tst[it] = walked;
            }
        }
        __syncthreads();
        if (hard) {
            bj = s_res[my_slot];
            if (kStats) n_tested += s_tst[my_slot];
        }
        __syncthreads(); // the queue is dead: the reduction may overwrite it
    }
    if (QUERY) { // the pair of GetCorrespondencePoints (vhm.cpp:31-88): the nearest point of the 27 buckets when it lies within max_dist
        if (valid) {
            double gx, gy, gz;
            float pxf, pyf;
            load_g(gx, gy, gz, pxf, pyf);
            int out = -2;
            if (bj >= 0) {
                const Pt3 q = blk_point(lp, bj);
                const double ex = (double)q.x - gx, ey = (double)q.y - gy, ez = (double)q.z - gz;
                if ((ex * ex + ey * ey) + ez * ez < rp.th2) out = (int)m.grid_idx[bj];
            } else if ((gx * gx + gy * gy) + gz * gz < rp.th2) {
                out = -1; // no bucket at all: the default PointStruct at the origin (QUIRK, vhm.cpp:37)
            }
            rp.q_out[i] = out;
        }
        return; // (uniform)
    }
    if (valid) {
        double gx, gy, gz;
        float pxf, pyf;
        load_g(gx, gy, gz, pxf, pyf);
        const double px = pxf, py = pyf, pz = (METHOD == ELM_P2P) ? (double)s_pz[threadIdx.x] : 0.0; // (the pair of P2P: J = [I | -[p]x])
        const unsigned stat = kStats ? s_st[threadIdx.x] : 0u;
        // the winner's float64 distance in the reference's arithmetic (range test, weight); no bucket at all (the search came
        // back empty): the reference's default PointStruct at the origin (vhm.cpp:37, QUIRK)
        float bx = 0.f, by = 0.f, bz = 0.f;
        int bidx = -1;
        // GICP never uses the matched point itself (its target is the neighbourhood mean, reg.cpp:97) except in the range test
        // d^2 < max_search_dist^2: a winner that stage 1 decided is a candidate of the point's own 2 x 2 x 2 block of cells, hence
        // within 3.5 cell edges of it, so with a search radius beyond that the test is known to pass and the three loads of the
        // winner's coordinates are skipped
        const bool range_known = METHOD != ELM_P2P && !hard && bj >= 0 && (12.25 * h * h) * 1.0001 < rp.th2;
        if (bj >= 0) {
            if (!range_known) {
                const Pt3 q = blk_point(lp, bj);
                bx = q.x; by = q.y; bz = q.z;
            }
            bidx = bj; // GICP: the payload records are stored in slot order (DevMap::grid_gicp)
        }
        const double ex = (double)bx - gx, ey = (double)by - gy, ez = (double)bz - gz;
        const double bd2 = range_known ? 0.0 : (ex * ex + ey * ey) + ez * ez;
        const double c_cand = (double)(stat & 0xFFFFu); // candidates of the reference's walk
        const double c_occ = (double)(stat >> 16);     // occupied neighbour voxels
        const double c_tested = (double)n_tested + (hard ? kFallbackUnit : 0.0); // high part: points served by stage 2
        if (METHOD == ELM_P2P) {
            if (bd2 < rp.th2) pair_p2p(v, S.Rinv, px, py, pz, ex, ey, ez, bd2, rp);
            if (kStats) { v[NV - 3] = c_cand; v[NV - 2] = c_occ; v[NV - 1] = c_tested; }
        } else {
            // finish_point_pair: no bucket at all -> the reference's default PointStruct at the origin with covariance I (QUIRK);
            // GICP's target position is the neighbourhood MEAN of the matched point (reg.cpp:97)
            const double dfin = (bidx >= 0) ? bd2 : (gx * gx + gy * gy) + gz * gz;
            if (COMPACT == 2) { // every record of this map is compact: mean + unit normal, k implied (0 with n.x = 2: identity)
                if (dfin < rp.th2) {
                    double mean[3] = {0.0, 0.0, 0.0}, nf[3] = {1.0, 0.0, 0.0}, k = 0.0;
                    if (bidx >= 0) {
                        const double* __restrict__ rec = m.grid_gicp8 + (size_t)bidx * 8;
#pragma unroll
                        for (int q = 0; q < 3; ++q) { mean[q] = rec[q]; nf[q] = rec[3 + q]; }
                        const bool ident = nf[0] == 2.0;
                        nf[0] = ident ? 1.0 : nf[0];
                        k = ident ? 0.0 : kCompactK;
                    }
                    pair_sum_compact<ELM_GICP>(P, mean[0] - gx, mean[1] - gy, mean[2] - gz, nf[0], nf[1], nf[2], k, rp);
                    P.ax = gx - S.T[12]; P.ay = gy - S.T[13]; P.az = gz - S.T[14];
                }
            } else if (dfin < rp.th2) {
                double Ci[9], mean[3], nf[3];
                if (bidx >= 0 && COMPACT) { // 48 of the record's 64 bytes: mean + unit normal -- the inverse covariance is I + 999 n n^T
                    const double* __restrict__ rec = m.grid_gicp8 + (size_t)bidx * 8;
#pragma unroll
                    for (int k = 0; k < 3; ++k) { mean[k] = rec[k]; nf[k] = rec[3 + k]; }
                    if (nf[0] == 2.0) { // identity covariance (a neighbourhood of the point alone): eigenvector e_x (reg.cpp:89-91)
                        nf[0] = 1.0;
                        compact_cinv(1.0, 0.0, 0.0, 0.0, Ci);
                    } else if (nf[0] == nf[0]) {
                        compact_cinv(nf[0], nf[1], nf[2], kCompactK, Ci);
                    } else { // outside the compact form (rank-deficient neighbourhood, U != V in its SVD): the stored record, by its index
                        const double* __restrict__ full = m.pt_gicp + (size_t)(unsigned)rec[7] * 16;
#pragma unroll
                        for (int k = 0; k < 9; ++k) Ci[k] = full[3 + k];
#pragma unroll
                        for (int k = 0; k < 3; ++k) nf[k] = full[12 + k];
                    }
                } else if (bidx >= 0) {
                    const double* __restrict__ rec = m.grid_gicp + (size_t)bidx * 16;
#pragma unroll
                    for (int k = 0; k < 9; ++k) Ci[k] = rec[3 + k];
#pragma unroll
                    for (int k = 0; k < 3; ++k) { mean[k] = rec[k]; nf[k] = rec[12 + k]; }
                } else {
                    Ci[0] = 1; Ci[1] = 0; Ci[2] = 0; Ci[3] = 0; Ci[4] = 1; Ci[5] = 0; Ci[6] = 0; Ci[7] = 0; Ci[8] = 1;
                    mean[0] = mean[1] = mean[2] = 0.0;
                    nf[0] = 1.0; nf[1] = 0.0; nf[2] = 0.0;
                }
                pair_sum_single<ELM_GICP>(P, mean[0] - gx, mean[1] - gy, mean[2] - gz, Ci, nf, rp);
                P.ax = gx - S.T[12]; P.ay = gy - S.T[13]; P.az = gz - S.T[14];
            }
            if (kStats) { P.c29 = c_cand; P.c30 = c_occ; P.c31 = c_tested; }
        }
    }
    // maps with an asymmetric flagged covariance (only the instantiations that read stored inverses can meet one): the side record
    __shared__ double s_asym[(METHOD != ELM_P2P && COMPACT != 2) ? kAsymSums : 1];
    __shared__ unsigned s_hitw[kBlock / 64];
    if (METHOD != ELM_P2P && COMPACT != 2) asym_mark(P.A, rp, s_hitw);
    if (METHOD == ELM_P2P) block_reduce_to_lds<NV, kRedPass>(v, s_buf, s_red);
    else
        block_reduce_pair_sum<kRedPass, kStats ? kSums : kSums - 3>(P, s_buf, s_red);
    if (METHOD != ELM_P2P && COMPACT != 2) asym_side_store(P.A, P.ax, P.ay, P.az, L, rp, s_buf, s_asym, s_hitw);
    const int tk = (int)threadIdx.x;
    publish_and_reduce((tk < kSums && (kStats || tk < kSums - 3)) ? ((METHOD == ELM_P2P) ? p2p_expand(s_red, tk) : s_red[tk]) : 0.0, L, s, sd.blk_begin,
                       sd.blk_end, partials, rp, s_buf);
}

// map build: the GICP payload records in grid slot order (16 lanes per record, one 8-byte word each); COMPACT: the 64-byte form
// {mean[3], unit normal[3], k, 0} (8 lanes per record)
template <int COMPACT>
__global__ __launch_bounds__(256) void k_gather_gicp(const DevMap m, size_t n_slots, double* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    constexpr unsigned W = COMPACT ? 8u : 16u;
    const size_t slot = t / W;
    if (slot >= n_slots) return;
    const unsigned src = m.grid_idx[slot];
    const unsigned w = (unsigned)(t % W);
    const unsigned from = COMPACT ? (w < 3u ? w : (w < 7u ? w + 9u : 15u)) : w; // mean 0..2, normal 12..14, k 15 (NaN: not of the compact form)
    // word 7 of a compact record: the record's bucket-order index, where the full 128-byte record of a non-conforming point is found
    double v = (src == 0xFFFFFFFFu) ? 0.0 : (COMPACT && w == 7u) ? (double)src : m.pt_gicp[(size_t)src * 16 + from];
    if (COMPACT && w == 3u && src != 0xFFFFFFFFu) {
        // the first 48 bytes must tell the three kinds of record apart (the kernel loads only those in the common case):
        //   k = kCompactK (regularised covariance): the unit normal as it is;  k = 0 (identity): n.x = 2 (not a unit vector);
        //   anything else (k = NaN: outside the compact form, or another k): n.x = NaN -> the full record is read
        const double k = m.pt_gicp[(size_t)src * 16 + 15];
        if (k == 0.0) v = 2.0;
        else if (!(fabs(k - kCompactK) <= 1e-7)) v = __builtin_nan("");
    }
    out[t] = v;
}

// map build: cnt27 | nocc27 << 16 for every voxel of the dense floor-key box (see DevMap::vox_stat)
__global__ __launch_bounds__(256) void k_vox_stat(const DevMap m, uint32_t* __restrict__ out) {
    const size_t n = (size_t)m.vnx * m.vny * m.vnz;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int uz = (int)(idx % m.vnz), uy = (int)((idx / m.vnz) % m.vny), ux = (int)(idx / ((size_t)m.vnz * m.vny));
    const int vx = ux + m.vx0, vy = uy + m.vy0, vz = uz + m.vz0;
    unsigned c = 0, o = 0;
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
                const Probe pr = probe_voxel(m, vx + dx, vy + dy, vz + dz);
                if (pr.vid >= 0 && pr.cnt > 0) { c += pr.cnt; ++o; }
            }
    out[idx] = (c > 0xFFFFu ? 0xFFFFu : c) | (o << 16);
}

// map build: sort every neighbourhood list by cell (stable: key = cell << 16 | position) and write its offset table.
// One 64-lane workgroup per query voxel, bitonic sort of <= 1024 keys in LDS.
__global__ __launch_bounds__(64) void k_nbr_cellsort(const DevMap m, const int32_t* __restrict__ qkeys, unsigned n_q,
                                                     const unsigned* __restrict__ offsets, const unsigned* __restrict__ counts,
                                                     Pt3* __restrict__ pts, unsigned* __restrict__ idx, uint16_t* __restrict__ cell_off) {
    __shared__ unsigned s_key[1024];
    __shared__ Pt3 s_pt[1024];
    __shared__ unsigned s_idx[1024];
    const unsigned q = blockIdx.x;
    if (q >= n_q) return;
    const unsigned l = threadIdx.x;
    const unsigned n = counts[q], o = offsets[q];
    uint16_t* co = cell_off + (size_t)q * kCellStride;
    if (n == 0 || n > 1024) { // (n > 1024 cannot happen: 27 buckets of <= 30 points; the host refuses larger voxel caps)
        for (unsigned c = l; c < (unsigned)kCellStride; c += 64) co[c] = 0;
        return;
    }
    const int vx = qkeys[3 * q], vy = qkeys[3 * q + 1], vz = qkeys[3 * q + 2];
    const double inv_h = 2.0 / m.voxel_size;
    const double ox = (double)(vx - 1) * m.voxel_size, oy = (double)(vy - 1) * m.voxel_size, oz = (double)(vz - 1) * m.voxel_size;
    unsigned np2 = 64;
    while (np2 < n) np2 <<= 1;
    for (unsigned j = l; j < np2; j += 64) {
        if (j < n) {
            const Pt3 p = pts[(size_t)o + j];
            s_pt[j] = p;
            s_idx[j] = idx[(size_t)o + j];
            const unsigned cell = (unsigned)((cell_of((double)p.x, ox, inv_h) * kCellAxis + cell_of((double)p.y, oy, inv_h)) * kCellAxis +
                                             cell_of((double)p.z, oz, inv_h));
            s_key[j] = (cell << 16) | j;
        } else {
            s_key[j] = 0xFFFFFFFFu;
        }
    }
    __syncthreads();
    for (unsigned k = 2; k <= np2; k <<= 1)
        for (unsigned jj = k >> 1; jj > 0; jj >>= 1) {
            for (unsigned t = l; t < np2; t += 64) {
                const unsigned p = t ^ jj;
                if (p > t) {
                    const unsigned a = s_key[t], b = s_key[p];
                    const bool up = (t & k) == 0;
                    if ((a > b) == up) { s_key[t] = b; s_key[p] = a; }
                }
            }
            __syncthreads();
        }
    __shared__ uint16_t s_bnd[kCells + 1]; // s_bnd[c] = first sorted position whose cell is >= c
    for (unsigned j = l; j < n; j += 64) {
        const unsigned src = s_key[j] & 0xFFFFu;
        pts[(size_t)o + j] = s_pt[src];
        idx[(size_t)o + j] = s_idx[src];
        const int c = (int)(s_key[j] >> 16);
        const int cprev = j ? (int)(s_key[j - 1] >> 16) : -1;
        for (int cc = cprev + 1; cc <= c; ++cc) s_bnd[cc] = (uint16_t)j;
        if (j == n - 1)
            for (int cc = c + 1; cc <= kCells; ++cc) s_bnd[cc] = (uint16_t)n;
    }
    __syncthreads();
    for (unsigned e = l; e < (unsigned)kCellStride; e += 64) {
        const unsigned col = e >> 3, z = e & 7;
        co[e] = (z <= (unsigned)kCellAxis) ? s_bnd[col * kCellAxis + z] : (uint16_t)0;
    }
}

// slot `slot` of the voxel-mean lists: its word (voxel id | position code << kVidBits; -1: padding) / its voxel id
__device__ __forceinline__ int vnbr_vc(const DevMap& m, unsigned slot) { return m.vnbr_blk[slot >> 2].vc[slot & 3u]; }
__device__ __forceinline__ unsigned vnbr_vid(const DevMap& m, unsigned slot) { return (unsigned)vnbr_vc(m, slot) & kVidMask; }

// ---- K1e: voxel-mean lists (VGICP) ------------------------------------------------------------------------
// GetCorrespondencesCov (vhm.cpp:90-151) visits the 27 neighbour voxels of the point's floor-keyed voxel and keeps the
// nearest voxel MEAN (strict <, first met wins).  Here the occupied ones (~10 of 27) are precomputed per query voxel in
// that visiting order as 64-byte blocks of four {float32 means, voxel id | position code}: one probe, then <= 7 contiguous blocks for
// the float32 filter, then ONE float64 record of the winner from the per-voxel table DevMap::vox_rec (round 6: a voxel's record is
// stored once, 64 B x n_vox -- cache resident -- instead of once per query list it appears in, 27 x) -- no staging, no barriers before
// the block reduction.  A float32 near-tie walks the list's records in the reference's order, float64.
constexpr int kVnbrRecs = 3; // VGICP: list records per round trip (measured 2 / 3 / 4 / 6 / 8: 105.9 / 110.5 / 102.3 / 99.3 / 92.3 k registrations/s)
constexpr int kAvgRecs = 1; // AVGICP: records per round trip: 1 -> 68 VGPRs, 7 waves, 89-91k registrations/s; 2 -> 99 VGPRs, 4 waves, 77.9k; 3 -> 76.9k;
                       // 4 -> 56.1k (round 3, 64-byte self-contained records; accumulating the pairs' w (I + k n n^T) in symmetric form
                       // without forming the 3x3 per pair: the same 88-89k -- the walk is a chain of dependent record loads, not arithmetic)
constexpr int kVnbrBlks = 1; // VGICP filter: float32 blocks of four means per round trip (1 / 2 / 3: 123.4 / 121.0 / 120.5 k registrations/s)
constexpr int kVnbrWaves = 1;
// FACES (AVGICP on maps with the dense face-sublist table; the walk reads the face sublists, 48 bytes per record):
//   1  nine entries of w C^-1 per pair, flagged voxels read their stored inverse in line
//   2  the fused walk (sum w, sum (w k) n n^T, b) on a map without a flagged voxel
//   4  the fused walk on a map WITH flagged voxels: their pairs are skipped, the workgroup is marked (RegParams::flagged)
//   3  the fix-up launch after 4: marked workgroups only, flagged records only (form 1's arithmetic), ADDED to the partial record
template <int METHOD, int COMPACT, int STATS, int FACES>
__global__ __launch_bounds__(kBlock, kVnbrWaves) void k_accumulate_vnbr(const DevMap m, const ScanDesc* __restrict__ scans, int batch,
                                                            unsigned total_blocks, const ScanState* __restrict__ st,
                                                            double* __restrict__ partials, const RegParams rp) {
    constexpr int kStats = (STATS == 1) ? 1 : 0;
    constexpr bool QUERY = STATS == 2; // elm_map_get_correspondences (GetCorrespondencesCov / GetCorrespondencesAllCov): RegParams::query in, q_out out
    __shared__ double s_buf[kRedPass * kBlock];
    __shared__ double s_red[kSums];
    const unsigned L = xcd_remap(blockIdx.x, total_blocks);
    const int s = find_scan(scans, batch, L, rp);
    const ScanState& S = st[s];
    if (S.done) return;
    const ScanDesc sd = scans[s];
    if (L >= sd.blk_end) return; // a scan whose size was only known on the device owns fewer workgroups than were launched for it
    const unsigned i = (L - sd.blk_begin) * kBlock + threadIdx.x;
    const bool valid = i < sd.n;
    if (FACES == 3) { // the fix-up launch: only workgroups whose fused walk met a flagged record have anything to add
        if (rp.flagged[L] == 0u) return; // (uniform)
        __syncthreads();
        if (threadIdx.x == 0) rp.flagged[L] = 0u; // ready for the next iteration
    }
    PairSum P;
    pair_sum_zero(P);
    if (valid) {
        double px = 0.0, py = 0.0, pz = 0.0, gx, gy, gz;
        if (QUERY) {
            gx = rp.query[3 * (size_t)i]; gy = rp.query[3 * (size_t)i + 1]; gz = rp.query[3 * (size_t)i + 2];
        } else {
            const Pt3 pf = sd.pts[i];
            px = pf.x; py = pf.y; pz = pf.z;
            gx = ((S.T[0] * px + S.T[4] * py) + S.T[8] * pz) + S.T[12];
            gy = ((S.T[1] * px + S.T[5] * py) + S.T[9] * pz) + S.T[13];
            gz = ((S.T[2] * px + S.T[6] * py) + S.T[10] * pz) + S.T[14];
        }
        const int vx = floor_key(gx, m), vy = floor_key(gy, m), vz = floor_key(gz, m);
        unsigned start = 0, cnt = 0;
        if (m.vq_dense) { // the dense floor-key box: no probe
            const int ux = vx - m.vq_x0, uy = vy - m.vq_y0, uz = vz - m.vq_z0;
            if ((unsigned)ux < (unsigned)m.vq_nx && (unsigned)uy < (unsigned)m.vq_ny && (unsigned)uz < (unsigned)m.vq_nz) {
                const size_t vidx = ((size_t)ux * m.vq_ny + uy) * m.vq_nz + uz;
                if (METHOD == ELM_AVGICP && m.vqf_dense) { // the face neighbours only
                    const unsigned w = m.vqf_dense[vidx];
                    start = w >> 3; cnt = w & 7u;
                } else {
                    const unsigned w = m.vq_dense[vidx];
                    start = (w >> 5) << 2; cnt = w & 31u; // lists start at multiples of four records
                }
            }
        } else {
            unsigned h = hash3(vx, vy, vz) & m.vqmask;
            for (;;) {
                const unsigned h2 = (h + 1) & m.vqmask;
                const int4 key = *reinterpret_cast<const int4*>(&m.vqslots[h]);
                const uint4 rg = *reinterpret_cast<const uint4*>(&m.vqslots[h].start);
                const int4 key2 = *reinterpret_cast<const int4*>(&m.vqslots[h2]);
                const uint4 rg2 = *reinterpret_cast<const uint4*>(&m.vqslots[h2].start);
                if (key.w < 0) break;
                if (key.x == vx && key.y == vy && key.z == vz) { start = rg.x; cnt = rg.y; break; }
                if (key2.w < 0) break;
                if (key2.x == vx && key2.y == vy && key2.z == vz) { start = rg2.x; cnt = rg2.y; break; }
                h = (h + 2) & m.vqmask;
            }
        }
        // AVGICP with the face table: lp = the face sublist (whole records); everything else addresses the list's slots through vnbr_vc
        const VoxRec* __restrict__ lp = m.vface + ((METHOD == ELM_AVGICP && m.vq_dense && m.vqf_dense) ? start : 0u);
        if (METHOD == ELM_VGICP) {
            double bd2 = DBL_MAX, bmx = 0.0, bmy = 0.0, bmz = 0.0;
            double bn[4] = {1.0, 0.0, 0.0, 0.0}; // the winner's plane normal and k (compact records)
            int bvid = -1;
            unsigned bj = 0;
            // float32 filter over the means (blocks of four, three 16-byte loads each instead of eight for the float64 records):
            // a winner that is clear of the runner-up by the rounding of the stored means and of the arithmetic is the strict
            // float64 minimum as well; its float64 record is read afterwards.  Near ties (practically never) take the float64 walk.
            bool exact = cnt != 0u;
            int wv0 = -1, wv1 = -1, wv2 = -1, wv3 = -1, wvc = -1; // slot words of the block that holds the current winner; the winner's
            if (cnt != 0u) {
                const unsigned nblk = (cnt + 3u) >> 2, blk0 = start >> 2;
                // distances to gh = float32(g): |g - gh| joins the stored means' rounding in the margin of the decision
                const float ghx = (float)gx, ghy = (float)gy, ghz = (float)gz;
                const float eg = (fabsf((float)(gx - (double)ghx)) + fabsf((float)(gy - (double)ghy)) + fabsf((float)(gz - (double)ghz))) * 1.000001f;
                const f32x2 gxy = {ghx, ghy}, gzz = {ghz, 0.f};
                unsigned m1 = 0x7F800000u, m2 = 0x7F800000u;
                for (unsigned b0 = 0; b0 < nblk; b0 += kVnbrBlks) {
                    VoxBlk B[kVnbrBlks];
#pragma unroll
                    for (int u = 0; u < kVnbrBlks; ++u) B[u] = m.vnbr_blk[(b0 + u < nblk) ? blk0 + b0 + u : m.vnbr_pad_blk];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < kVnbrBlks; ++u) {
                        f32x2 da, db;
                        blk_dist_h(B[u].g, gxy, gzz, da, db);
                        const unsigned was = m1;
                        two_smallest(da.x, 0u, m1, m2);
                        two_smallest(da.y, 1u, m1, m2);
                        two_smallest(db.x, 2u, m1, m2);
                        two_smallest(db.y, 3u, m1, m2);
                        const bool ch = m1 != was;
                        wv0 = ch ? B[u].vc[0] : wv0; wv1 = ch ? B[u].vc[1] : wv1; wv2 = ch ? B[u].vc[2] : wv2; wv3 = ch ? B[u].vc[3] : wv3;
                    }
                }
                // |float32(mean) - mean| <= 2^-24 |mean|_1 <= 6.5e-8 (|g|_1 + 6 voxel sizes); float32 arithmetic + key bits: 2^-18
                const float em = 6.5e-8f * (fabsf(ghx) + fabsf(ghy) + fabsf(ghz) + 6.0f * (float)m.voxel_size) + eg;
                const float s1 = __builtin_sqrtf(__uint_as_float(m1 & ~3u)), s2 = __builtin_sqrtf(__uint_as_float(m2 & ~3u));
                if (s2 - s2 * 3.814697265625e-06f - em > s1 + s1 * 3.814697265625e-06f + em) {
                    const unsigned sl = m1 & 3u; // the winner's slot word rode along with its block
                    wvc = sl == 0u ? wv0 : sl == 1u ? wv1 : sl == 2u ? wv2 : wv3;
                    exact = false;
                }
            }
            if (exact) {
                for (unsigned j = 0; j < cnt; j += kVnbrRecs) { // kVnbrRecs records per round trip (slot word, then the voxel's record)
                    VoxRec r[kVnbrRecs];
#pragma unroll
                    for (int u = 0; u < kVnbrRecs; ++u) r[u] = m.vox_rec[vnbr_vid(m, start + min(j + u, cnt - 1))]; // past the end: the last record again (never < itself)
#pragma unroll
                    for (int u = 0; u < kVnbrRecs; ++u) {
                        const double ex = r[u].mx - gx, ey = r[u].my - gy, ez = r[u].mz - gz;
                        const double d2 = (ex * ex + ey * ey) + ez * ez;
                        const bool c = d2 < bd2; // strict: the first met keeps a tie (vhm.cpp:128)
                        bj = c ? j + u : bj;
                        bd2 = c ? d2 : bd2;
                    }
                }
            }
            if (cnt) { // the winner's float64 record, from the per-voxel table (after the float64 walk: its slot word is an L1 hit)
                const VoxRec w = m.vox_rec[exact ? vnbr_vid(m, start + min(bj, cnt - 1)) : ((unsigned)wvc & kVidMask)];
                bvid = w.vid; bmx = w.mx; bmy = w.my; bmz = w.mz;
                if (COMPACT) { bn[0] = w.nx; bn[1] = w.ny; bn[2] = w.nz; bn[3] = w.k; }
                const double ex = w.mx - gx, ey = w.my - gy, ez = w.mz - gz;
                bd2 = (ex * ex + ey * ey) + ez * ez; // the walk's own arithmetic for this record
            }
            // finish_voxel_pair: no voxel at all -> the reference's default VoxelStruct at the origin with covariance I (QUIRK)
            const double dfin = (bvid >= 0) ? bd2 : (gx * gx + gy * gy) + gz * gz;
            if (QUERY) { // the pair of GetCorrespondencesCov (vhm.cpp:90-151); no occupied neighbour at all: the default CovStruct at the origin
                rp.q_out[i] = (dfin < rp.th2) ? bvid : -2;
            } else
            if (COMPACT == 2) { // every voxel of this map is compact (no voxel at all: the default at the origin, covariance I: k = 0)
                if (dfin < rp.th2) {
                    if (bvid < 0) bmx = bmy = bmz = 0.0;
                    pair_sum_compact<ELM_VGICP>(P, bmx - gx, bmy - gy, bmz - gz, bn[0], bn[1], bn[2], bn[3], rp);
                    P.ax = gx - S.T[12]; P.ay = gy - S.T[13]; P.az = gz - S.T[14];
                }
            } else if (dfin < rp.th2) {
                if (bvid < 0) bmx = bmy = bmz = 0.0;
                double Ci[9];
                if (bvid >= 0 && COMPACT && bn[3] == bn[3]) { // the record carried the normal and k: no second fetch
                    compact_cinv(bn[0], bn[1], bn[2], bn[3], Ci);
                } else if (bvid >= 0) { // (k = NaN: a voxel whose inverse is not of the compact form)
#pragma unroll
                    for (int k = 0; k < 9; ++k) Ci[k] = m.vox_cinv[(size_t)bvid * 9 + k];
                } else {
                    Ci[0] = 1; Ci[1] = 0; Ci[2] = 0; Ci[3] = 0; Ci[4] = 1; Ci[5] = 0; Ci[6] = 0; Ci[7] = 0; Ci[8] = 1;
                }
                pair_sum_single<ELM_VGICP>(P, bmx - gx, bmy - gy, bmz - gz, Ci, nullptr, rp);
                P.ax = gx - S.T[12]; P.ay = gy - S.T[13]; P.az = gz - S.T[14];
            }
            (void)px; (void)py; (void)pz;
            if (kStats) { P.c29 = (double)cnt; P.c30 = (double)cnt; P.c31 = (double)cnt; }
        } else {
            // AVGICP, GetCorrespondencesAllCov (vhm.cpp:153-206): every existing FACE neighbour (and the voxel itself) whose
            // mean is within range is a pair of its own.  The records carry the neighbour's position code (dx+1)*9+(dy+1)*3+
            // (dz+1); the seven wanted ones are met in list order (-x, -y, -z, 0, +z, +y, +x) instead of the reference's
            // (0, +x, -x, +y, -y, +z, -z): the same pairs, added in another order.
            double n_pairs = 0.0;
            AvgPairSum Q;
            avg_pair_init(Q);
            const bool faces_only = FACES != 0; // lp is a face sublist (the launcher checks m.vq_dense && m.vqf_dense)
            const bool via_faces = FACES != 0 || (m.vq_dense && m.vqf_dense); // (the QUERY instantiation reads the face sublists too when the map has them)
            if (COMPACT && (FACES == 2 || FACES == 4)) { // (4: on a map with flagged voxels -- their pairs are left to the fix-up launch)
                // Face sublists of a map whose every voxel is of the compact form (DevMap::vface_plain): 48 bytes per record as below, and
                // A_v = w (I + k n n^T) is never formed -- the point gathers sum w, sum (w k) n n^T (six entries) and
                // b = sum w e + (w k)(n . e) n, fused: 25 float64 operations per pair less than the nine-entry form, the same sums up
                // to the rounding of the last bit (the pair test d^2 < th^2 keeps the reference's arithmetic).
                double W = 0.0, M00 = 0.0, M01 = 0.0, M02 = 0.0, M11 = 0.0, M12 = 0.0, M22 = 0.0;
                for (unsigned j = 0; j < cnt; ++j) {
                    const double2* __restrict__ rp16 = reinterpret_cast<const double2*>(lp + j);
                    const double2 r0 = rp16[0], r1 = rp16[1], r2 = rp16[2]; // (mx, my), (mz, nx), (ny, nz)
                    n_pairs += 1.0;
                    const double ex = r0.x - gx, ey = r0.y - gy, ez = r1.x - gz;
                    const double d2 = (ex * ex + ey * ey) + ez * ez;
                    if (d2 < rp.th2) {
                        if (FACES == 4 && r1.y != r1.y) { // a flagged voxel (NaN normal): the fix-up launch adds this pair, with the stored inverse
                            rp.flagged[L] = 1u; // (only maps with such voxels meet this; they always come with the array)
                            continue;
                        }
                        const double den = rp.th + d2;
                        const double w = div_close(rp.th2, den * den); // square(th) / square(th + |r|^2)
                        Q.n += 1.0;
                        if (!(w < 0.01)) { // reg.cpp:201 -- skipped pairs stay in the fitness denominator
                            const double nx = r1.y, ny = r2.x, nz = r2.y; // (identity covariance: the zero normal, k_vface)
                            const double wk = w * kCompactK;
                            const double sn = wk * __builtin_fma(nz, ez, __builtin_fma(ny, ey, nx * ex));
                            Q.b[0] = __builtin_fma(sn, nx, __builtin_fma(w, ex, Q.b[0]));
                            Q.b[1] = __builtin_fma(sn, ny, __builtin_fma(w, ey, Q.b[1]));
                            Q.b[2] = __builtin_fma(sn, nz, __builtin_fma(w, ez, Q.b[2]));
                            const double ux = wk * nx, uy = wk * ny, uz = wk * nz;
                            M00 = __builtin_fma(ux, nx, M00); M01 = __builtin_fma(ux, ny, M01); M02 = __builtin_fma(ux, nz, M02);
                            M11 = __builtin_fma(uy, ny, M11); M12 = __builtin_fma(uy, nz, M12); M22 = __builtin_fma(uz, nz, M22);
                            W += w;
                            Q.rsum += sqrt_dist2(d2);
                        }
                    }
                }
                Q.A[0] = W + M00; Q.A[1] = M01; Q.A[2] = M02;
                Q.A[3] = M01; Q.A[4] = W + M11; Q.A[5] = M12;
                Q.A[6] = M02; Q.A[7] = M12; Q.A[8] = W + M22;
            } else if (COMPACT && FACES) { // (FACES = 3, the fix-up launch: the flagged records alone)
                // Face sublists, compact records: 48 of the record's 64 bytes -- mean and unit normal; k = kCompactK is implied, the other two
                // kinds are flagged in the normal's first word by k_vface (2: identity covariance, NaN: outside the compact form -> the stored
                // inverse by the record's voxel id).  Three 16-byte loads per pair instead of four: the walk is a chain of record loads.
                for (unsigned j = 0; j < cnt; ++j) {
                    const double2* __restrict__ rp16 = reinterpret_cast<const double2*>(lp + j);
                    const double2 r0 = rp16[0], r1 = rp16[1], r2 = rp16[2]; // (mx, my), (mz, nx), (ny, nz)
                    double Ci[9];
                    if (FACES == 3 && r1.y == r1.y) continue;
                    if (r1.y == 2.0) {
                        compact_cinv(1.0, 0.0, 0.0, 0.0, Ci);
                    } else if (r1.y == r1.y) {
                        compact_cinv(r1.y, r2.x, r2.y, kCompactK, Ci);
                    } else {
                        const double* __restrict__ cp = m.vox_cinv + (size_t)lp[j].vid * 9;
#pragma unroll
                        for (int k = 0; k < 9; ++k) Ci[k] = cp[k];
                    }
                    n_pairs += 1.0;
                    const double ex = r0.x - gx, ey = r0.y - gy, ez = r1.x - gz;
                    const double d2 = (ex * ex + ey * ey) + ez * ez;
                    if (d2 < rp.th2) avg_pair_add(Q, ex, ey, ez, Ci, rp);
                }
            } else
            for (unsigned j = 0; j < cnt; j += kAvgRecs) { // kAvgRecs records (two 16-byte loads each) per round trip
                VoxRec r[kAvgRecs];
                if (via_faces) {
#pragma unroll
                    for (int u = 0; u < kAvgRecs; ++u) r[u] = lp[min(j + u, cnt - 1)];
                } else { // the whole list: the slot word carries the position code, the record comes from the per-voxel table
#pragma unroll
                    for (int u = 0; u < kAvgRecs; ++u) {
                        const unsigned vc = (unsigned)vnbr_vc(m, start + min(j + u, cnt - 1));
                        r[u] = m.vox_rec[vc & kVidMask];
                        r[u].pad = (int32_t)(vc >> kVidBits);
                    }
                }
                // the face neighbours among them: their inverse covariances are requested together, before the first is used
                bool use[kAvgRecs];
                double Ci[kAvgRecs][9];
#pragma unroll
                for (int u = 0; u < kAvgRecs; ++u) {
                    // the face neighbours (and the voxel itself): position codes 4, 10, 12, 13, 14, 16, 22 -- one shift of a 27-bit mask; the
                    // face sublists hold nothing else (uniform branch: no test at all)
                    constexpr unsigned kFaceMask = (1u << 4) | (1u << 10) | (1u << 12) | (1u << 13) | (1u << 14) | (1u << 16) | (1u << 22);
                    if (faces_only) use[u] = j + u < cnt;
                    else use[u] = j + u < cnt && ((kFaceMask >> ((unsigned)r[u].pad & 31u)) & 1u) != 0u;
                    if (COMPACT && r[u].k == r[u].k) {
                        compact_cinv(r[u].nx, r[u].ny, r[u].nz, r[u].k, Ci[u]);
                    } else {
                        const double* __restrict__ cp = m.vox_cinv + (size_t)(use[u] ? r[u].vid : 0) * 9;
#pragma unroll
                        for (int k = 0; k < 9; ++k) Ci[u][k] = cp[k];
                    }
                }
#pragma unroll
                for (int u = 0; u < kAvgRecs; ++u) {
                    if (!use[u]) continue;
                    n_pairs += 1.0;
                    const double ex = r[u].mx - gx, ey = r[u].my - gy, ez = r[u].mz - gz;
                    const double d2 = (ex * ex + ey * ey) + ez * ez;
                    if (QUERY) { // GetCorrespondencesAllCov (vhm.cpp:153-206): the pair's place in the reference's order (0, +x, -x, +y, -y, +z, -z)
                        // position codes (dx+1)*9 + (dy+1)*3 + (dz+1): 13, 22, 4, 16, 10, 14, 12
                        const unsigned code = (unsigned)r[u].pad & 31u;
                        const int rank = code == 13u ? 0 : code == 22u ? 1 : code == 4u ? 2 : code == 16u ? 3 : code == 10u ? 4 : code == 14u ? 5 : 6;
                        if (d2 < rp.th2) rp.q_out[8 * (size_t)i + rank] = r[u].vid;
                    } else
                    if (d2 < rp.th2) avg_pair_add(Q, ex, ey, ez, Ci[u], rp);
                }
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) P.A[k] = Q.A[k];
            P.b[0] = Q.b[0]; P.b[1] = Q.b[1]; P.b[2] = Q.b[2];
            P.rsum = Q.rsum; P.n = Q.n;
            if (Q.n > 0.0) { // (a point without a pair -- a NaN / infinite return among them -- contributes zeros, not 0 x NaN)
                P.ax = gx - S.T[12]; P.ay = gy - S.T[13]; P.az = gz - S.T[14];
            }
            if (kStats && FACES != 3) { P.c29 = n_pairs; P.c30 = n_pairs; P.c31 = n_pairs; } // (the fused walk has counted every record)
        }
    }
    if (QUERY) return; // (uniform) the pairs are written, there are no sums
    // The side record of maps with an asymmetric flagged covariance (asym_side_store): every instantiation that reads stored inverses
    // computes it; the fused walk on a map with flagged voxels (FACES = 4) has skipped those pairs and writes zeros, which its fix-up
    // launch (FACES = 3, marked workgroups only) overwrites.  Clean maps (COMPACT = 2 / FACES = 2) never carry the pointer.
    __shared__ double s_asym[(COMPACT == 2 || FACES == 2) ? 1 : kAsymSums];
    __shared__ unsigned s_hitw[kBlock / 64];
    constexpr bool kAsymHere = COMPACT != 2 && FACES != 2 && FACES != 4;
    if (kAsymHere) asym_mark(P.A, rp, s_hitw);
    block_reduce_pair_sum<kRedPass, kStats ? kSums : kSums - 3>(P, s_buf, s_red);
    if (FACES == 4) {
        if (rp.asym && threadIdx.x < (unsigned)kAsymSums) rp.asym[(size_t)L * kAsymSums + threadIdx.x] = 0.0;
    } else if (kAsymHere) {
        asym_side_store(P.A, P.ax, P.ay, P.az, L, rp, s_buf, s_asym, s_hitw);
    }
    if (FACES == 3) { // added to the record the fused walk of this workgroup wrote earlier on the stream (the solve's reduction comes after both)
        if (threadIdx.x < (unsigned)kSums - 3u) partials[(size_t)L * kSums + threadIdx.x] += s_red[threadIdx.x];
        return;
    }
    publish_and_reduce((threadIdx.x < (kStats ? kSums : kSums - 3)) ? s_red[threadIdx.x] : 0.0, L, s, sd.blk_begin, sd.blk_end, partials, rp, s_buf);
}

__global__ __launch_bounds__(256) void k_vnbr_fill(const DevMap m, const int32_t* __restrict__ qkeys, unsigned n_q,
                                                   const unsigned* __restrict__ offsets, VoxBlk* __restrict__ out_blk) {
    const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_q) return;
    const int vx = qkeys[3 * q], vy = qkeys[3 * q + 1], vz = qkeys[3 * q + 2];
    unsigned o = offsets[q]; // a multiple of four
    const unsigned o0 = o;
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
                const Probe pr = probe_voxel(m, vx + dx, vy + dy, vz + dz);
                if (pr.vid < 0 || pr.cnt == 0) continue;
                VoxBlk& B = out_blk[o >> 2]; // slot o % 4 of block o / 4: the float32 mean (the filter) + voxel id | position code << 26
                B.g.x[o & 3u] = (float)m.vox_mean[(size_t)pr.vid * 3];
                B.g.y[o & 3u] = (float)m.vox_mean[(size_t)pr.vid * 3 + 1];
                B.g.z[o & 3u] = (float)m.vox_mean[(size_t)pr.vid * 3 + 2];
                B.vc[o & 3u] = (int32_t)((unsigned)pr.vid | ((unsigned)(((dx + 1) * 3 + (dy + 1)) * 3 + (dz + 1)) << kVidBits)); // (AVGICP picks the face codes)
                ++o;
            }
    for (; ((o - o0) & 3u) != 0u; ++o) { // padding slots of the last block: never the nearest
        VoxBlk& B = out_blk[o >> 2];
        B.g.x[o & 3u] = 1e18f; B.g.y[o & 3u] = 1e18f; B.g.z[o & 3u] = 1e18f;
        B.vc[o & 3u] = -1;
    }
}
// the per-voxel float64 record of the voxel-mean search (DevMap::vox_rec): mean, plane normal, k -- ONE copy per voxel
__global__ __launch_bounds__(256) void k_vox_rec_fill(const DevMap m, VoxRec* __restrict__ out) {
    const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m.n_vox) return;
    VoxRec r;
    r.mx = m.vox_mean[(size_t)v * 3]; r.my = m.vox_mean[(size_t)v * 3 + 1]; r.mz = m.vox_mean[(size_t)v * 3 + 2];
    r.nx = m.vox_nk[(size_t)v * 4]; r.ny = m.vox_nk[(size_t)v * 4 + 1]; r.nz = m.vox_nk[(size_t)v * 4 + 2];
    r.k = m.vox_nk[(size_t)v * 4 + 3];
    r.vid = (int32_t)v;
    r.pad = 0;
    out[v] = r;
}

__global__ __launch_bounds__(256) void k_nbr_count(const DevMap m, const int32_t* __restrict__ qkeys, unsigned n_q,
                                                   unsigned* __restrict__ counts, unsigned* __restrict__ nocc) {
    const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_q) return;
    const int vx = qkeys[3 * q], vy = qkeys[3 * q + 1], vz = qkeys[3 * q + 2];
    unsigned c = 0, o = 0;
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
                const Probe pr = probe_voxel(m, vx + dx, vy + dy, vz + dz);
                if (pr.vid >= 0 && pr.cnt > 0) { c += pr.cnt; ++o; }
            }
    counts[q] = c;
    nocc[q] = o;
}
__global__ __launch_bounds__(256) void k_nbr_fill(const DevMap m, const int32_t* __restrict__ qkeys, unsigned n_q,
                                                  const unsigned* __restrict__ offsets, Pt3* __restrict__ out,
                                                  unsigned* __restrict__ out_idx) {
    const unsigned q = (blockIdx.x * blockDim.x + threadIdx.x) >> 5; // one 32-lane group per query voxel
    const unsigned l = threadIdx.x & 31;
    if (q >= n_q) return;
    const int vx = qkeys[3 * q], vy = qkeys[3 * q + 1], vz = qkeys[3 * q + 2];
    unsigned o = offsets[q];
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
                const Probe pr = probe_voxel(m, vx + dx, vy + dy, vz + dz);
                if (pr.vid < 0) continue;
                for (unsigned j = l; j < pr.cnt; j += 32) {
                    const float4 p = m.pts[pr.start + j];
                    Pt3 t; t.x = p.x; t.y = p.y; t.z = p.z;
                    out[(size_t)o + j] = t;
                    out_idx[(size_t)o + j] = pr.start + j;
                }
                o += pr.cnt;
            }
}

// ------------------------------------------------------------------------------------------------------
// K2
// ------------------------------------------------------------------------------------------------------
__device__ void update_inverse(ScanState& S) {
    double R[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[r * 3 + c] = S.T[c * 4 + r];
    inv3(R, S.Rinv);
    for (int r = 0; r < 3; ++r)
        S.tinv[r] = -((S.Rinv[r * 3] * S.T[12] + S.Rinv[r * 3 + 1] * S.T[13]) + S.Rinv[r * 3 + 2] * S.T[14]);
}

__device__ void init_scan_state(ScanState& S, const double* __restrict__ T0, int reg, int map_empty) {
    for (int k = 0; k < 16; ++k) S.T[k] = T0[k];
    update_inverse(S);
    S.fitness = 0.0;
    for (int k = 0; k < 36; ++k) S.local_cov[k] = (k % 7 == 0) ? 1.0 : 0.0; // reg.cpp:280
    S.n_corr_last = 0.0;
    S.pt_iters = 0.0; S.cand_total = 0.0; S.occ_total = 0.0; S.fallback_blocks = 0.0; S.tested_total = 0.0;
    S.done = map_empty ? 1 : 0; // VOXEL MAP EMPTY (reg.cpp:291-295): is_success = false, return initial_guess
    S.success = 0;
    S.gate = map_empty ? 1 : 0;
    S.iters = 0;
    S.reg = reg;
    S._pad = 0;
}

__global__ __launch_bounds__(64) void k_init_state(ScanState* st, const double* __restrict__ T0, int batch, int map_empty,
                                                   int* active) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= batch) return;
    if (!map_empty) atomicAdd(active, 1); // scans still iterating (the host zeroed the counter)
    init_scan_state(st[s], T0 + (size_t)s * 16, s, map_empty);
}

// Small batches (a single RunRegister above all): descriptors and initial guesses travel as kernel arguments -- no H2D copies, no
// memset of the counter.  n_dev != nullptr: the scan's size is only known on the device (the deskew + downsample kernels have just
// produced it): the descriptor takes n from there, so the host never waits for it.
__global__ __launch_bounds__(64) void k_init_pack(ScanDesc* scans, ScanState* st, const InitPack pack, int batch, int map_empty, int* active,
                                                  const unsigned* __restrict__ n_dev) {
    const int s = threadIdx.x;
    if (s == 0) { active[0] = map_empty ? 0 : batch; active[1] = 0; } // [1]: the rank-agreement fault word (RegParams::rank_check)
    if (s >= batch) return;
    ScanDesc d = pack.d[s];
    if (n_dev) {
        d.n = *n_dev;
        d.n_total = d.n;
        d.blk_end = d.blk_begin + (d.n + kBlock - 1) / kBlock;
    }
    scans[s] = d;
    init_scan_state(st[s], pack.T0[s], s, map_empty);
}
void launch_init_pack(hipStream_t s, ScanDesc* scans, ScanState* st, const InitPack& pack, int batch, int map_empty, int* active, const unsigned* n_dev) {
    hipLaunchKernelGGL(k_init_pack, dim3(1), dim3(64), 0, s, scans, st, pack, batch, map_empty, active, n_dev);
}

// Continuous batching: after the solve of an iteration, every slot whose registration has finished saves its final state
// and takes the next pending registration (descriptor + initial guess), so every accumulate launch stays full until the
// queue runs dry.  Slots are served in slot order by one thread: the assignment is deterministic (identical on every rank).
constexpr int kMaxSlots = 4096; // 32 KB of LDS for the two slot tables
__global__ __launch_bounds__(1024) void k_stream_refill(ScanDesc* scans, ScanState* st, int slots, const QueueItem* __restrict__ queue,
                                                       const double* __restrict__ qT0, ScanState* out_state, StreamCtrl* ctrl, int first, int save) {
    __shared__ int s_assign[kMaxSlots]; // registration to start in the slot, -1 = slot keeps going, -2 = slot goes idle
    __shared__ int s_save[kMaxSlots];   // registration whose final state is copied out, -1 = none
    // the slots' flags are fetched by all threads at once; the serial part below only touches LDS
    for (int s = threadIdx.x; s < slots; s += blockDim.x) s_save[s] = (!first && st[s].done && st[s].reg >= 0) ? st[s].reg : -1;
    __syncthreads();
    {
        // free slots take the pending registrations in SLOT ORDER: an exclusive prefix count of the free flags (every thread owns a
        // contiguous run of slots; wave scan + the 16 wave totals) instead of one thread walking the slots (that walk, a chain of
        // dependent LDS accesses, took 12 us of this launch's 24 at 256 slots)
        __shared__ int s_wtot[16];
        const int per = (slots + (int)blockDim.x - 1) / (int)blockDim.x, s0 = (int)threadIdx.x * per, s1 = min(slots, s0 + per);
        int mine = 0;
        for (int s = s0; s < s1; ++s) mine += (first || s_save[s] >= 0) ? 1 : 0;
        int inc = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(inc, off, 64);
            if ((int)(threadIdx.x & 63u) >= off) inc += o;
        }
        if ((threadIdx.x & 63u) == 63u) s_wtot[threadIdx.x >> 6] = inc;
        __syncthreads();
        int before = inc - mine, all = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
            before += (w < (int)(threadIdx.x >> 6)) ? s_wtot[w] : 0;
            all += s_wtot[w];
        }
        const int next0 = first ? 0 : ctrl->next, total = ctrl->total;
        int r = next0 + before;
        for (int s = s0; s < s1; ++s) {
            const bool free_slot = first || s_save[s] >= 0; // (otherwise: still iterating, or already idle)
            s_assign[s] = free_slot ? ((r < total) ? r : -2) : -1;
            r += free_slot ? 1 : 0;
        }
        __syncthreads(); // every thread has read ctrl->next
        if (threadIdx.x == 0) {
            ctrl->next = min(total, next0 + all);
            if (first) { ctrl->completed = 0; ctrl->done_iter = -1; }
            else if (save) ctrl->completed += all; // (save = 0: the solve has saved and counted them)
        }
    }
    __syncthreads();
    constexpr int W = (int)(sizeof(ScanState) / sizeof(double));
    for (int s = (int)(threadIdx.x >> 6); save && s < slots; s += (int)(blockDim.x >> 6)) { // one wavefront per slot
        const int r = s_save[s];
        if (r < 0) continue;
        const double* src = reinterpret_cast<const double*>(&st[s]);
        double* dst = reinterpret_cast<double*>(&out_state[r]);
        for (int k = (int)(threadIdx.x & 63); k < W; k += 64) dst[k] = src[k];
    }
    __syncthreads();
    for (int s = threadIdx.x; s < slots; s += blockDim.x) {
        const int r = s_assign[s];
        if (r >= 0) {
            const QueueItem q = queue[r];
            scans[s].pts = q.pts;
            scans[s].n = q.n;
            scans[s].n_total = q.n_total;
            init_scan_state(st[s], qT0 + (size_t)r * 16, r, 0);
        } else if (r == -2) {
            st[s].done = 1;
            st[s].reg = -1;
        }
    }
}

// Wave-parallel 6x6 LDL^T with Eigen's diagonal pivoting (Eigen/src/Cholesky/LDLT.h, ldlt_inplace<Lower>::unblocked): that routine
// is left-looking -- at step k only column k has been updated -- so its pivot search `mat.diagonal().tail(size - k).cwiseAbs()
// .maxCoeff()` sees the ORIGINAL diagonal entries of the rows not yet eliminated, in their current positions (every symmetric
// exchange k <-> p moves row k to position p), and keeps the first of equal maxima.  The elimination itself runs right-looking here
// (same L and D up to rounding), on a matrix spread over lanes: lane l < 36 holds A[l/6][l%6].
// All 64 lanes execute it with uniform control flow.  Returns x = A^-1 b (uniform in every lane) and, when want_inv,
// leaves A^-1[l/6][l%6] in `inv_elem` of lane l < 36.  ~3 us instead of ~25 us for the single-lane version.
__device__ __forceinline__ void wave_ldlt6(double a, const double* b, double x[6], bool want_inv, double& inv_elem) {
    const int lane = threadIdx.x & 63;
    const int li = (lane < 36) ? lane / 6 : 0, lj = (lane < 36) ? lane % 6 : 0;
    unsigned active = 0x3Fu;
    int order[6];
    double piv[6];
    int rank_i = 6, rank_j = 6, rank_l = 6; // elimination step of row li / column lj / index `lane` (lane < 6)
    double d0[6]; // |original diagonal|
    int pos[6];   // pos[j] = the row that Eigen's exchanges have moved to position j
#pragma unroll
    for (int i = 0; i < 6; ++i) { d0[i] = fabs(__shfl(a, i * 7, 64)); pos[i] = i; }
    auto d0_of = [&](int r) { return r == 0 ? d0[0] : r == 1 ? d0[1] : r == 2 ? d0[2] : r == 3 ? d0[3] : r == 4 ? d0[4] : d0[5]; };
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int jb = k;
        double best = d0_of(pos[k]);
#pragma unroll
        for (int j = k + 1; j < 6; ++j) {
            const double v = d0_of(pos[j]);
            if (v > best) { best = v; jb = j; } // strict: the first maximum stays
        }
        int p = pos[k];
#pragma unroll
        for (int j = k + 1; j < 6; ++j)
            if (j == jb) { p = pos[j]; pos[j] = pos[k]; }
        pos[k] = p;
        const double dp = __shfl(a, p * 7, 64);
        const double aip = __shfl(a, li * 6 + p, 64), apj = __shfl(a, p * 6 + lj, 64);
        const bool ai = ((active >> li) & 1u) && li != p, aj = ((active >> lj) & 1u) && lj != p;
        if (dp != 0.0) {
            const double lip = aip / dp;
            if (ai && aj) a -= lip * apj;      // Schur complement of the remaining block
            else if (ai && lj == p) a = lip;   // column p below/right of the pivot now holds L[i][p]
            else if (li == p && aj) a = apj / dp;
        } else {
            if ((ai && lj == p) || (li == p && aj)) a = 0.0;
        }
        order[k] = p;
        piv[k] = dp;
        active &= ~(1u << p);
        if (li == p) rank_i = k;
        if (lj == p) rank_j = k;
        if (lane == p) rank_l = k;
    }
    (void)rank_j;
    // ---- solve A x = b with lanes 0..5 holding the vector
    double y = (lane < 6) ? b[lane] : 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { // forward: L y' = P b
        const int p = order[k];
        const double yp = __shfl(y, p, 64);
        const double lip = __shfl(a, (lane < 6 ? lane : 0) * 6 + p, 64);
        if (lane < 6 && rank_l > k) y -= lip * yp;
    }
    {
        double d = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) d = (rank_l == k) ? piv[k] : d;
        y = (fabs(d) > 5.6e-309) ? y / d : 0.0; // Eigen zeroes the components of (numerically) zero pivots
    }
#pragma unroll
    for (int k = 5; k >= 0; --k) { // backward: L^T x = y
        const int p = order[k];
        const double lip = __shfl(a, (lane < 6 ? lane : 0) * 6 + p, 64);
        double c = (lane < 6 && rank_l > k) ? lip * y : 0.0;
        c += __shfl_xor(c, 1, 64);
        c += __shfl_xor(c, 2, 64);
        c += __shfl_xor(c, 4, 64);
        const double tot = __shfl(c, 0, 64);
        if (lane == p) y -= tot;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = __shfl(y, i, 64);
    // ---- inverse: the same substitutions on the six unit vectors, one matrix element per lane
    inv_elem = 0.0;
    if (want_inv) {
        double Y = (lane < 36 && li == lj) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int p = order[k];
            const double ypc = __shfl(Y, p * 6 + lj, 64);
            const double lip = __shfl(a, li * 6 + p, 64);
            if (lane < 36 && rank_i > k) Y -= lip * ypc;
        }
        {
            double d = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) d = (rank_i == k) ? piv[k] : d;
            Y = (fabs(d) > 5.6e-309) ? Y / d : 0.0;
        }
#pragma unroll
        for (int k = 5; k >= 0; --k) {
            const int p = order[k];
            const double lip = __shfl(a, li * 6 + p, 64);
            const double c = (lane < 36 && rank_i > k) ? lip * Y : 0.0;
            double colsum = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) colsum += __shfl(c, i * 6 + lj, 64);
            if (lane < 36 && li == p) Y -= colsum;
        }
        inv_elem = Y;
    }
}

// Matrix<double, 6, 6>::inverse() as Eigen computes it (PartialPivLU: row exchanges on the first largest |entry| of the column, then the
// solve against the identity), on the first wavefront: lane l < 36 holds A[l / 6][l % 6] and receives inverse[l / 6][l % 6].  Every
// element sees the operations of the textbook one-thread loop in the same order (the eliminations of different elements are independent,
// the substitutions run row by row) -- ~120 instructions instead of the ~1 500 of round 3's one-lane version with LDS operands (on a map
// with asymmetric covariances nearly every GICP solve takes this path: 1.3 ms per bench step at 256 slots).  All 64 lanes execute it.
__device__ __forceinline__ double wave_inverse6_partial_piv(double a) {
    const int lane = threadIdx.x & 63;
    const int li = (lane < 36) ? lane / 6 : 0, lj = (lane < 36) ? lane % 6 : 0;
    int perm[6] = {0, 1, 2, 3, 4, 5};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int piv = k;
        double best = fabs(__shfl(a, k * 6 + k, 64));
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            const double v = fabs(__shfl(a, i * 6 + k, 64));
            if (v > best) { best = v; piv = i; } // the first largest |entry| of the column
        }
        if (piv != k) { // (uniform) exchange rows k and piv
            const int src = (li == k) ? piv : ((li == piv) ? k : li);
            a = __shfl(a, src * 6 + lj, 64);
#pragma unroll
            for (int i = k + 1; i < 6; ++i)
                if (i == piv) { const int t = perm[k]; perm[k] = perm[i]; perm[i] = t; }
        }
        const double d = __shfl(a, k * 6 + k, 64);
        if (d != 0.0 && li > k && lj == k) a = a / d;
        const double mult = __shfl(a, li * 6 + k, 64), ukc = __shfl(a, k * 6 + lj, 64);
        if (li > k && lj > k) a -= mult * ukc;
    }
    // the solve against the (row-permuted) identity: lane (i, c) holds y_i of column c
    double Y = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) Y = (li == i && perm[i] == lj) ? 1.0 : Y;
#pragma unroll
    for (int i = 1; i < 6; ++i) { // forward: unit lower triangle
        double acc = Y;
#pragma unroll
        for (int j = 0; j < i; ++j) {
            const double lij = __shfl(a, i * 6 + j, 64), yj = __shfl(Y, j * 6 + lj, 64);
            acc -= lij * yj;
        }
        if (li == i) Y = acc;
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) { // backward: upper triangle
        double acc = Y;
#pragma unroll
        for (int j = i + 1; j < 6; ++j) {
            const double uij = __shfl(a, i * 6 + j, 64), yj = __shfl(Y, j * 6 + lj, 64);
            acc -= uij * yj;
        }
        const double uii = __shfl(a, i * 6 + i, 64);
        if (li == i) Y = acc / uii;
    }
    return Y;
}

// queue position for a free slot, or -1 when nothing is pending.  Plain streams: every registration is there from the start, one
// atomicAdd hands them out.  Host-fed streams: only registrations whose scan has landed in HBM (ctrl->ready, published by the upload
// stream after the scan's ordering kernel) may start, so the counter advances by compare-and-swap and never overshoots.
__device__ __forceinline__ int claim_registration(const StreamArgs& sa, int prev) {
    if (sa.stride > 0) { // static queue per slot (every rank takes the same decision)
        const int r = prev + sa.stride;
        return (r < sa.ctrl->total) ? r : -1;
    }
    if (!sa.hostfed) {
        const int r = atomicAdd(&sa.ctrl->next, 1);
        return (r < sa.ctrl->total) ? r : -1;
    }
    int old = __hip_atomic_load(&sa.ctrl->next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        const int ready = __hip_atomic_load(&sa.ctrl->ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (old >= ready) return -1;
        const int seen = atomicCAS(&sa.ctrl->next, old, old + 1);
        if (seen == old) return old;
        old = seen;
    }
}
// The slot takes the next pending registration (descriptor + initial state: init_scan_state's arithmetic) or goes idle.
// All 64 lanes of the solve's first wavefront call it (uniform).
__device__ __forceinline__ void start_slot(const StreamArgs& sa, ScanState& S, int s, int prev) {
    const int lane = threadIdx.x & 63;
    int r = 0;
    if (lane == 0) r = claim_registration(sa, prev);
    r = __shfl(r, 0, 64);
    if (r >= 0) {
        const double* T0 = sa.qT0 + (size_t)r * 16;
        if (lane >= 1 && lane < 37) S.local_cov[lane - 1] = ((lane - 1) % 7 == 0) ? 1.0 : 0.0; // reg.cpp:280
        if (lane == 0) { // the scalar part of init_scan_state, same arithmetic
            const QueueItem q = sa.queue[r];
            sa.scans[s].pts = q.pts;
            sa.scans[s].n = q.n;
            sa.scans[s].n_total = q.n_total;
            for (int k = 0; k < 16; ++k) S.T[k] = T0[k];
            update_inverse(S);
            S.fitness = 0.0;
            S.n_corr_last = 0.0;
            S.pt_iters = 0.0; S.cand_total = 0.0; S.occ_total = 0.0; S.fallback_blocks = 0.0; S.tested_total = 0.0;
            S.done = 0;
            S.success = 0;
            S.gate = 0;
            S.iters = 0;
            S.reg = r;
            S._pad = 0;
        }
    } else if (lane == 0) {
        S.done = 1; // idle slot
        S.reg = -1;
    }
}
// Continuous batching inside the solve (single-rank streams): the wavefront that has just finished a registration saves its final
// state and takes the next pending registration for the slot -- what k_stream_refill does, without the extra launch.  The
// queue position comes from an atomic counter, so WHICH slot serves a registration depends on the order the workgroups get
// here; a registration's arithmetic does not depend on its slot (uniform slot sizes, partial sums in block order), so every
// result is unchanged.  Multi-rank streams must assign identically on every rank: there slot s serves the registrations s, s + S,
// s + 2 S, ... (StreamArgs::stride), also from inside the solve; k_stream_refill only does the initial fill.
// The lead lane's plain stores to S are ordered against the other lanes' loads by a workgroup-scope release / acquire fence pair
// (the wavefront is the only writer and the only reader of S inside this launch) and re-read with agent-scope loads.
__device__ __forceinline__ void finish_slot(const StreamArgs& sa, ScanState& S, int s) {
#if !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__) && defined(__HIP_DEVICE_COMPILE__)
#error "finish_slot relies on gfx9 memory ordering (one vmcnt for loads and stores, write-through vector L1)"
#endif
    constexpr int W = (int)(sizeof(ScanState) / sizeof(double));
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int reg_old = __hip_atomic_load(&S.reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double* src = reinterpret_cast<const double*>(&S);
    double* dst = reinterpret_cast<double*>(&sa.out_state[reg_old]);
    for (int k = lane; k < W; k += 64) dst[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0 && atomicAdd(&sa.ctrl->completed, 1) + 1 == sa.ctrl->total) sa.ctrl->done_iter = sa.iter; // the stream's last registration
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0); // the copy's loads are done before the state is overwritten
    __builtin_amdgcn_wave_barrier();
    if (sa.save_only) return; // the refill launch hands the slot its next registration (S.done = 1 and S.reg >= 0 mark it free)
    start_slot(sa, S, s, reg_old);
}

constexpr int kSolveThreads = 1024;
// NT = 1024 threads (default) or 256 (half-set streams: a workgroup of 4 waves and <= 80 VGPRs fits the slot a retiring accumulate
// workgroup frees, so the solve can run beside the other half's accumulate launch).  The reduction always adds the partial records in
// the order of 32 groups of 32 lanes -- 256 threads walk four of those groups each -- so the sums do not depend on NT.
// id of a slot's (registration, iteration) in the exchanged record (RegParams::rank_check).  The registration index enters modulo 2^19 so
// that n id^2 stays an exact integer in a double for any number of registrations per call (id < 2^23 + 16, id^2 < 2^47, n <= 64 ranks:
// below 2^53); an idle slot (registration -1) has id = iteration & 15.  Two ranks that swap registrations r and r + 2^19 k in one slot, or
// that swap two registrations between two slots while each slot agrees with itself across the ranks, are not told apart: the check sees
// ranks that DISAGREE about a slot, which is what a broken collective or a non-deterministic solve produces.
__device__ __forceinline__ double rank_check_id(int reg, int iters) {
    return 16.0 * (double)((reg < 0 ? -1 : (reg & 0x7FFFF)) + 1) + (double)(iters & 15);
}
template <int NT>
__global__ __launch_bounds__(NT, NT == 256 ? 6 : 1) void k_solve(const ScanDesc* scans, ScanState* st,
                                                         const double* __restrict__ partials, double* sums,
                                                         const RegParams rp, elm_iter_trace* trace, int mode, int* active,
                                                         const StreamArgs sa) {
    const int s = blockIdx.x;
    ScanState& S = st[s];
    const int t = threadIdx.x;
    __shared__ double tot[kSums];
    __shared__ double part[kSolveThreads / 32][kSums];
    static_assert(NT == kSolveThreads || NT == 256, "solve workgroup size");
    const bool done = S.done != 0;
    const bool radar = rp.radar != 0;         // k_accumulate_radar's records: 64 doubles, all 36 entries of J^T M J (single GPU, unfused)
    __shared__ double full[36];               // radar / asymmetric side sums: J^T M J row-major, all 36 entries
    // a map with an asymmetric flagged covariance: the accumulate kernels also wrote 16-double side records (asym_side_store)
    const bool asym = rp.asym != nullptr && !radar && rp.method != ELM_P2P;
    __shared__ double dsum[kAsymSums];        // the scan's side sums: the strict lower triangle of H_w - H_w^T
    __shared__ double apart[kSolveThreads / kAsymSums][kAsymSums];
    if (radar) {
        // multi-rank: mode 1 leaves the 64 sums of the scan in sums[s][64] for the all-reduce, mode 2 (one wavefront) reads them back
        double a = 0.0;
        if (mode != 2) {
            const int k = t & 63, g = t >> 6; // sixteen groups of 64 lanes, one lane per sum, fixed order
            double* part64 = &part[0][0];
            double v = 0.0;
            if (!done) {
                const ScanDesc sd = scans[s];
                for (unsigned b = sd.blk_begin + g; b < sd.blk_end; b += kSolveThreads / 64) v += partials[(size_t)b * kRadarSums + k];
            }
            part64[g * 64 + k] = v;
            __syncthreads();
            if (t < 64) {
                a = part64[t];
#pragma unroll
                for (int q = 1; q < kSolveThreads / 64; ++q) a += part64[q * 64 + t];
                if (mode == 1) sums[(size_t)s * kRadarSums + t] = (t < kRadarAcc) ? a : 0.0; // (zeros for finished scans keep the buffer defined)
            }
            if (mode == 1) return;
        } else if (t < 64) {
            a = (scans[s].blk_end > scans[s].blk_begin) ? sums[(size_t)s * kRadarSums + t] : 0.0;
        }
        if (t < 64) {
            if (t < 36) full[t] = a;
            else if (t < kRadarAcc) tot[21 + (t - 36)] = a; // J^T M r, residual sum, pair count, statistics: the slots of the 32-sum layout
        }
    } else if (mode != 2) {
        // deterministic reduction of this scan's per-workgroup partial sums: 32 strided groups of 32 lanes read whole
        // 256-byte records (four independent loads in flight per lane), then the group sums are added in a fixed order
        const int k = t & 31;
        constexpr unsigned G = kSolveThreads / 32;
        auto group_sum = [&](int g) -> double {
            double v = 0.0;
            if (!done) {
                const ScanDesc sd = scans[s];
                unsigned b = sd.blk_begin + g;
                double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
                for (; b + 15 * G < sd.blk_end; b += 16 * G) { // sixteen loads in flight, summed in the order of the loop below
                    double a[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) a[q] = partials[(size_t)(b + q * G) * kSums + k];
#pragma unroll
                    for (int q = 0; q < 16; q += 4) { v0 += a[q]; v1 += a[q + 1]; v2 += a[q + 2]; v3 += a[q + 3]; }
                }
                for (; b + 3 * G < sd.blk_end; b += 4 * G) {
                    const double a0 = partials[(size_t)b * kSums + k], a1 = partials[(size_t)(b + G) * kSums + k];
                    const double a2 = partials[(size_t)(b + 2 * G) * kSums + k], a3 = partials[(size_t)(b + 3 * G) * kSums + k];
                    v0 += a0; v1 += a1; v2 += a2; v3 += a3;
                }
                // the tail keeps the accumulator rotation of the unrolled loop, so trailing all-zero records (slots of a stream
                // are sized for the largest scan) leave every sum bit-identical to the exact-size layout
                if (b < sd.blk_end) { v0 += partials[(size_t)b * kSums + k]; b += G; }
                if (b < sd.blk_end) { v1 += partials[(size_t)b * kSums + k]; b += G; }
                if (b < sd.blk_end) { v2 += partials[(size_t)b * kSums + k]; b += G; }
                v = (v0 + v1) + (v2 + v3);
            }
            return v;
        };
        if (NT == kSolveThreads) { // one group per 32 lanes
            part[t >> 5][k] = group_sum(t >> 5);
        } else { // 256 threads: four of the 32 groups each, the same sums
#pragma unroll 1
            for (int g = t >> 5; g < (int)G; g += NT / 32) part[g][k] = group_sum(g);
        }
        if (asym) {
            // the side records, 64 groups of 16 lanes, one running sum per group (trailing all-zero records of a slot sized for a larger
            // scan change nothing), the groups added in a fixed order below
            constexpr int GA = kSolveThreads / kAsymSums;
            const int k2 = t & (kAsymSums - 1);
            auto side_sum = [&](int g) -> double {
                double v = 0.0;
                if (!done) {
                    const ScanDesc sd = scans[s];
                    for (unsigned b = sd.blk_begin + (unsigned)g; b < sd.blk_end; b += (unsigned)GA) v += rp.asym[(size_t)b * kAsymSums + k2];
                }
                return v;
            };
            if (NT == kSolveThreads) {
                apart[t / kAsymSums][k2] = side_sum(t / kAsymSums);
            } else {
#pragma unroll 1
                for (int g = t / kAsymSums; g < GA; g += NT / kAsymSums) apart[g][k2] = side_sum(g);
            }
        }
        __syncthreads();
        if (t < 32) {
            double a = part[0][t];
#pragma unroll
            for (int q = 1; q < kSolveThreads / 32; ++q) a += part[q][t];
            if (mode == 1) {
                if (rp.rank_check && t >= 29) { // (1, id, id^2) in the slots of the work counters (zero in production): see RegParams::rank_check
                    const double id = rank_check_id(S.reg, S.iters);
                    a = (t == 29) ? 1.0 : (t == 30) ? id : id * id;
                }
                sums[(size_t)s * kSums + t] = a; // zeros for finished scans keep the all-reduce buffer defined
            } else tot[t] = a;
        } else if (asym && t < 32 + kAsymSums) {
            const int k2 = t - 32;
            double a = apart[0][k2];
#pragma unroll 1
            for (int q = 1; q < kSolveThreads / kAsymSums; ++q) a += apart[q][k2];
            if (mode == 1) rp.asym_sums[(size_t)s * kAsymSums + k2] = a;
            else dsum[k2] = a;
        }
        if (mode == 1) return;
    } else {
        // (a scan without a single workgroup -- no points on this rank -- has no record and no sums: zeros)
        if (t < 32) tot[t] = (scans[s].blk_end > scans[s].blk_begin) ? sums[(size_t)s * kSums + t] : 0.0;
        else if (asym && t < 32 + kAsymSums)
            dsum[t - 32] = (rp.asym_sums && scans[s].blk_end > scans[s].blk_begin) ? rp.asym_sums[(size_t)s * kAsymSums + (t - 32)] : 0.0;
    }
    __syncthreads();
    if (t >= 64) return; // the first wave does the rest with uniform control flow; lane 0 owns the state
    if (mode == 2 && rp.rank_check && !radar) {
        // every rank must be iterating the same registration (and iteration) in this slot: exact integer arithmetic in doubles
        const double n = sums[(size_t)s * kSums + 29], a1 = sums[(size_t)s * kSums + 30], a2 = sums[(size_t)s * kSums + 31];
        const double id = rank_check_id(S.reg, S.iters);
        if (t == 0 && !(n >= 1.0 && a1 == n * id && a2 == n * id * id)) atomicOr(active + 1, 1);
        if (t >= 29 && t < 32) tot[t] = 0.0; // (they are not work counters)
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
    }
    if (done) {
        // host-fed stream: an idle slot (nothing was pending when it last looked) takes a registration whose scan has arrived since
        if (sa.ctrl && sa.hostfed && __hip_atomic_load(&S.reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) start_slot(sa, S, s, -1);
        return;
    }
    const bool lead = (t == 0);

    // Side sums that are not all zero: this iteration's J^T M J is not symmetric (some pair met an asymmetric stored inverse) -- all 36
    // entries are carried from here on, the factorisation reads the lower triangle and GICP's covariance output inverts the full matrix,
    // as for use_radar_cov.  All zero (every iteration of every ordinary registration on such a map): the symmetric path, bit for bit.
    bool nonsym = false;
    if (asym) {
#pragma unroll 1
        for (int k = 0; k < 15; ++k) nonsym = nonsym || (dsum[k] != 0.0);
    }

    if (rp.method != ELM_P2P && !radar) {
        // the covariance-weighted kernels accumulate in the world frame (add_pair_world): H_l = P^T H_w P, b_l = P^T b_w with
        // P = diag(R, R), R the rotation the pairs were formed with (S.T is updated further down)
        __shared__ double hw[36], bw[6];
        if (t < 36) {
            const int i = t / 6, j = t % 6;
            double hv = tot[tri(i < j ? i : j, i < j ? j : i)];
            if (nonsym && i > j) // lower triangle = upper triangle + D (slot order of asym_side_store)
                hv += dsum[(i < 3) ? (i - 1 + j) : (j < 3) ? (3 + (i - 3) * 3 + j) : (9 + i + j - 3 - 1)];
            hw[t] = hv;
        }
        if (t < 6) bw[t] = tot[21 + t];
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        // R(r, c) = S.T[c * 4 + r].  H_l(i, j) = sum_{k, l} P(k, i) H_w(k, l) P(l, j): only the 3x3 block of (i, j) contributes
        if (t < 36) {
            const int i = t / 6, j = t % 6, bi = (i / 3) * 3, bj = (j / 3) * 3, ii = i % 3, jj = j % 3;
            double h = 0.0;
            for (int k = 0; k < 3; ++k) {
                double row = 0.0;
                for (int l = 0; l < 3; ++l) row += hw[(bi + k) * 6 + (bj + l)] * S.T[jj * 4 + l];
                h += S.T[ii * 4 + k] * row;
            }
            if (i <= j) tot[tri(i, j)] = h;
            if (nonsym) full[t] = h;
        }
        if (t < 6) {
            const int bi = (t / 3) * 3, ii = t % 3;
            tot[21 + t] = (S.T[ii * 4 + 0] * bw[bi] + S.T[ii * 4 + 1] * bw[bi + 1]) + S.T[ii * 4 + 2] * bw[bi + 2];
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
    }

    const ScanDesc sd = scans[s];
    const int iter = S.iters + 1; // i_iteration++ (reg.cpp:311)
    const double n_corr = tot[28];
    if (lead) {
        S.iters = iter;
        S.n_corr_last = n_corr;
        S.pt_iters += (double)sd.n_total;
        S.cand_total += tot[29];
        S.occ_total += tot[30];
        const double fb = floor(tot[31] / 1099511627776.0);
        S.fallback_blocks += fb;
        S.tested_total += tot[31] - fb * 1099511627776.0;
    }
    elm_iter_trace* tr = (trace && iter <= ELM_MAX_ITER_TRACE) ? &trace[(size_t)S.reg * ELM_MAX_ITER_TRACE + (iter - 1)] : nullptr;

    // corres_ratio = (float)i_source_corr_num / i_source_total_num (reg.cpp:351): float division, compared as double
    const float ratio_f = (float)n_corr / (float)sd.n_total;
    if ((double)ratio_f < rp.min_overlap) { // reg.cpp:352-356: fail, return the current pose, fitness untouched
        if (lead) {
            if (tr) {
                for (int k = 0; k < 36; ++k) tr->JTJ[k] = 0.0;
                for (int k = 0; k < 6; ++k) { tr->JTr[k] = 0.0; tr->x[k] = 0.0; }
                tr->residual_sum = tot[27];
                tr->n_corr = n_corr;
                tr->step_norm = 0.0;
                for (int k = 0; k < 16; ++k) tr->T[k] = S.T[k];
            }
            S.done = 1;
            atomicSub(active, 1);
            S.success = 0;
            S.gate = 2;
        }
        if (sa.ctrl) finish_slot(sa, S, s); // uniform: every lane took this branch
        return;
    }
    const double fitness = tot[27] / n_corr; // d_fitness_score_ = d_residual_sum / source_global.size()

    // JTJ + lambda * diag(JTJ), one element per lane (reg.cpp:55-56 / 136-138 / 213-214)
    const int li = (t < 36) ? t / 6 : 0, lj = (t < 36) ? t % 6 : 0;
    // (radar: JTJ is not symmetric and JTJ.ldlt() reads its lower triangle -- the factorisation of the symmetric matrix with that triangle)
    const bool full36 = radar || nonsym;
    const double hij = full36 ? full[(li < lj ? lj : li) * 6 + (li < lj ? li : lj)] : tot[tri(li < lj ? li : lj, li < lj ? lj : li)];
    const double a = (li == lj) ? hij + rp.lm_lambda * hij : hij;
    double x[6], inv_elem;
    wave_ldlt6(a, &tot[21], x, rp.method == ELM_GICP && !full36, inv_elem);
    if (rp.method == ELM_GICP && !full36 && t < 36) S.local_cov[t] = inv_elem; // reg.cpp:141-142 (symmetric: layout-free)
    if (rp.method == ELM_GICP && full36) {
        // JTJ_regularized.inverse() of the FULL matrix (reg.cpp:141-142): Eigen's PartialPivLU + solve against the identity; column-major out
        const double el = (t < 36) ? ((li == lj) ? full[t] + rp.lm_lambda * full[t] : full[t]) : 0.0;
        const double inv_el = wave_inverse6_partial_piv(el);
        const double inv_t = __shfl(inv_el, lj * 6 + li, 64);
        if (t < 36) S.local_cov[t] = inv_t;
    }

    double dR[9];
    rotvec_to_matrix(&x[3], dR);
    // T <- T * [dR | dt]  (reg.cpp:378), column-major T
    double Tn[16];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
            Tn[c * 4 + r] = (S.T[0 * 4 + r] * dR[0 * 3 + c] + S.T[1 * 4 + r] * dR[1 * 3 + c]) + S.T[2 * 4 + r] * dR[2 * 3 + c];
        Tn[12 + r] = ((S.T[0 * 4 + r] * x[0] + S.T[1 * 4 + r] * x[1]) + S.T[2 * 4 + r] * x[2]) + S.T[12 + r];
    }
    Tn[3] = 0.0; Tn[7] = 0.0; Tn[11] = 0.0; Tn[15] = 1.0;
    const double step = matrix_to_angle(dR) + sqrt((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]); // reg.cpp:381-384
    if (tr && t < 36) tr->JTJ[t] = full36 ? full[lj * 6 + li] : hij; // column-major (symmetric unless radar / asymmetric side sums)
    bool fin = false;
    if (lead) {
        S.fitness = fitness;
        for (int k = 0; k < 16; ++k) S.T[k] = Tn[k];
        update_inverse(S);
        if (tr) {
            for (int k = 0; k < 6; ++k) { tr->JTr[k] = tot[21 + k]; tr->x[k] = x[k]; }
            tr->residual_sum = tot[27];
            tr->n_corr = n_corr;
            tr->step_norm = step;
            for (int k = 0; k < 16; ++k) tr->T[k] = Tn[k];
        }
        if (step < rp.term_thr || iter >= rp.max_iter) { // reg.cpp:385-387 / loop end
            S.done = 1;
            atomicSub(active, 1);
            const bool bad = fitness > rp.max_fitness; // reg.cpp:405-409 (NaN compares false, like the reference)
            S.success = bad ? 0 : 1;
            S.gate = bad ? 3 : 0;
            fin = true;
        }
    }
    if (sa.ctrl && __shfl((int)fin, 0, 64)) finish_slot(sa, S, s);
}

// ------------------------------------------------------------------------------------------------------
// K3 / K4: map covariances
// ------------------------------------------------------------------------------------------------------
// k of the compact form I + k n n^T of an inverse covariance (n = unit plane normal): k = trace - 3; *ok = false when the matrix is
// not of that form to 1e-10 relative (a rank-deficient neighbourhood whose SVD returned U != V: the full matrix is then kept in use)
__device__ __forceinline__ double compact_k(const double Ci[9], const double n[3], bool* ok) {
    const double k = ((Ci[0] + Ci[4]) + Ci[8]) - 3.0;
    double err = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) err = fmax(err, fabs(Ci[i * 3 + j] - (((i == j) ? 1.0 : 0.0) + k * n[i] * n[j])));
    *ok = err <= 1e-10 * (1.0 + fabs(k)); // (NaN compares false)
    return k;
}

// A flagged covariance whose stored inverse is not symmetric (U != V in the SVD of a rank-deficient neighbourhood):
// the packed 21-sum forms cannot carry its antisymmetric part, so the host routes such a map to the per-pair kernels (bad[1] counts them).
__device__ inline bool inv_asymmetric(const double Ci[9]) {
    double mx = 0.0;
    for (int k = 0; k < 9; ++k) mx = fmax(mx, fabs(Ci[k]));
    const double d = fmax(fmax(fabs(Ci[1] - Ci[3]), fabs(Ci[2] - Ci[6])), fabs(Ci[5] - Ci[7]));
    return d > 1e-12 * mx; // (false on NaN / inf entries: a map with non-finite points keeps the path it always had)
}

__global__ __launch_bounds__(256) void k_voxel_cov(const DevMap m, const uint2* __restrict__ ranges, double* vox_mean,
                                                   double* vox_cov, double* vox_cinv, double* vox_nk, unsigned* bad) {
    const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m.n_vox) return;
    const uint2 rg = ranges[v];
    const unsigned n = rg.y;
    double mean[3] = {0, 0, 0};
    double nrm[3] = {1, 0, 0};
    double C[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (n == 1) {
        const float4 q = m.pts[rg.x];
        mean[0] = q.x; mean[1] = q.y; mean[2] = q.z;
    } else if (n >= 2) {
        double sx = 0, sy = 0, sz = 0;
        for (unsigned j = 0; j < n; ++j) {
            const float4 q = m.pts[rg.x + j];
            sx += (double)q.x; sy += (double)q.y; sz += (double)q.z;
        }
        mean[0] = sx / (double)n; mean[1] = sy / (double)n; mean[2] = sz / (double)n;
        double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (unsigned j = 0; j < n; ++j) {
            const float4 q = m.pts[rg.x + j];
            const double d[3] = {(double)q.x - mean[0], (double)q.y - mean[1], (double)q.z - mean[2]};
            for (int a = 0; a < 3; ++a)
                for (int bq = 0; bq < 3; ++bq) c[a * 3 + bq] += d[a] * d[bq];
        }
        for (int k = 0; k < 9; ++k) c[k] /= (double)(n - 1);
        plane_regularize(c, C, nrm);
        const double nn = sqrt((nrm[0] * nrm[0] + nrm[1] * nrm[1]) + nrm[2] * nrm[2]);
        if (nn > 0.0) { nrm[0] /= nn; nrm[1] /= nn; nrm[2] /= nn; }
    }
    for (int k = 0; k < 3; ++k) vox_mean[(size_t)v * 3 + k] = mean[k];
    for (int k = 0; k < 9; ++k) vox_cov[(size_t)v * 9 + k] = C[k];
    double Ci[9];
    inv3(C, Ci); // the inverse the registration needs (add_pair_world), by the cofactor form Eigen uses for Matrix3d::inverse()
    for (int k = 0; k < 9; ++k) vox_cinv[(size_t)v * 9 + k] = Ci[k];
    bool ok;
    const double kk = compact_k(Ci, nrm, &ok);
    // (k is 0 -- identity -- or 1 / 1e-3 - 1 for every regularised covariance; the face sublists imply it, so anything else is `bad`)
    ok = ok && (kk == 0.0 || fabs(kk - kCompactK) <= 1e-7);
    if (!ok) atomicAdd(bad, 1u);
    if (!ok && inv_asymmetric(Ci)) atomicAdd(bad + 1, 1u);
    for (int k = 0; k < 3; ++k) vox_nk[(size_t)v * 4 + k] = nrm[k];
    vox_nk[(size_t)v * 4 + 3] = ok ? kk : __builtin_nan(""); // NaN: the pairs of this voxel read vox_cinv[vid]
}

__global__ __launch_bounds__(256) void k_point_cov(const DevMap m, double d2max, double* pt_gicp, double* pt_cov, unsigned* bad) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m.n_pts) return;
    const float4 pf = m.pts[i];
    const double px = pf.x, py = pf.y, pz = pf.z;
    const int vx = floor_key(px, m.voxel_size), vy = floor_key(py, m.voxel_size), vz = floor_key(pz, m.voxel_size);
    // pass 1: neighbours = {self} + every bucket point of the 27 floor-keyed voxels with d^2 <= r^2 -- the point
    // itself is found again there (vhm.hpp:202-220), so it is counted twice
    double sx = px, sy = py, sz = pz;
    unsigned n = 1;
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
                const Probe pr = probe_voxel(m, vx + dx, vy + dy, vz + dz);
                if (pr.vid < 0) continue;
                for (unsigned j = 0; j < pr.cnt; ++j) {
                    const float4 q = m.pts[pr.start + j];
                    const double ex = (double)q.x - px, ey = (double)q.y - py, ez = (double)q.z - pz;
                    if ((ex * ex + ey * ey) + ez * ez <= d2max) {
                        sx += (double)q.x; sy += (double)q.y; sz += (double)q.z;
                        ++n;
                    }
                }
            }
    double mean[3] = {px, py, pz};
    double C[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double nf[3] = {1, 0, 0}; // eigenvectors of the identity are the identity: col(0) = e_x
    if (n > 1) {
        mean[0] = sx / (double)n; mean[1] = sy / (double)n; mean[2] = sz / (double)n;
        double c[9];
        {
            const double d[3] = {px - mean[0], py - mean[1], pz - mean[2]};
            for (int a = 0; a < 3; ++a)
                for (int bq = 0; bq < 3; ++bq) c[a * 3 + bq] = d[a] * d[bq];
        }
        for (int dx = -1; dx <= 1; ++dx)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dz = -1; dz <= 1; ++dz) {
                    const Probe pr = probe_voxel(m, vx + dx, vy + dy, vz + dz);
                    if (pr.vid < 0) continue;
                    for (unsigned j = 0; j < pr.cnt; ++j) {
                        const float4 q = m.pts[pr.start + j];
                        const double ex = (double)q.x - px, ey = (double)q.y - py, ez = (double)q.z - pz;
                        if ((ex * ex + ey * ey) + ez * ez <= d2max) {
                            const double d[3] = {(double)q.x - mean[0], (double)q.y - mean[1], (double)q.z - mean[2]};
                            for (int a = 0; a < 3; ++a)
                                for (int bq = 0; bq < 3; ++bq) c[a * 3 + bq] += d[a] * d[bq];
                        }
                    }
                }
        for (int k = 0; k < 9; ++k) c[k] /= (double)(n - 1);
        plane_regularize(c, C, nf);
    }
    double Ci[9];
    inv3(C, Ci); // what the registration needs (add_pair_world); the covariance itself goes to pt_cov for the read-backs
    {
        // the fitness normal as a unit vector (the reference normalises R^-1 n per pair, reg.cpp:93-95)
        const double nn = sqrt((nf[0] * nf[0] + nf[1] * nf[1]) + nf[2] * nf[2]);
        if (nn > 0.0) { nf[0] /= nn; nf[1] /= nn; nf[2] /= nn; }
    }
    double* rec = pt_gicp + (size_t)i * 16;
    for (int k = 0; k < 3; ++k) { rec[k] = mean[k]; rec[12 + k] = nf[k]; }
    for (int k = 0; k < 9; ++k) rec[3 + k] = Ci[k];
    bool ok;
    double kk = compact_k(Ci, nf, &ok); // k of Cinv = I + k n n^T (the compact 64-byte records, DevMap::grid_gicp8)
    ok = ok && (kk == 0.0 || fabs(kk - kCompactK) <= 1e-7); // (the 48-byte reads imply k: 0 or 1 / 1e-3 - 1, nothing else is compact)
    rec[15] = ok ? kk : __builtin_nan(""); // NaN: this point's inverse is not of that form -- its pairs read the full record
    if (!ok) atomicAdd(bad, 1u);
    if (!ok && inv_asymmetric(Ci)) atomicAdd(bad + 1, 1u);
    for (int k = 0; k < 9; ++k) pt_cov[(size_t)i * 9 + k] = C[k];
}

// ------------------------------------------------------------------------------------------------------
// K0: deskew (float32 semantics of pcm.cpp:780-824; sin/cos evaluated in fp64 and rounded once to float32,
// which reproduces glibc's correctly-rounded sinf/cosf results)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_deskew(const float* __restrict__ xyz, const float* __restrict__ rel_time,
                                                unsigned n, const DeskewDev d, float* __restrict__ out) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const double d_rel_time = (double)rel_time[i];
    const double d_point_time = d.time_scan_cur + d_rel_time;
    const int cur = d.imu_pointer_cur;
    const float f_rot_x_end = (float)d.rot_x[cur], f_rot_y_end = (float)d.rot_y[cur], f_rot_z_end = (float)d.rot_z[cur];
    // FindRotation (pcm.cpp:731-762)
    int front = 0;
    while (front < cur) {
        if (d_point_time < d.imu_time[front]) break;
        ++front;
    }
    float rxc, ryc, rzc;
    if (d_point_time > d.imu_time[front] || front == 0) {
        rxc = (float)d.rot_x[front]; ryc = (float)d.rot_y[front]; rzc = (float)d.rot_z[front];
    } else {
        const int back = front - 1;
        const double tf = d.imu_time[front], tb = d.imu_time[back];
        const double ratio_front = (d_point_time - tb) / (tf - tb);
        const double ratio_back = (tf - d_point_time) / (tf - tb);
        rxc = (float)(d.rot_x[front] * ratio_front + d.rot_x[back] * ratio_back);
        ryc = (float)(d.rot_y[front] * ratio_front + d.rot_y[back] * ratio_back);
        rzc = (float)(d.rot_z[front] * ratio_front + d.rot_z[back] * ratio_back);
    }
    // FindPosition (pcm.cpp:764-778)
    float pxc = 0.f, pyc = 0.f;
    if (d.odom_available) {
        const float f_ratio = (float)(d_rel_time / (d.time_scan_end - d.time_scan_cur));
        pxc = f_ratio * d.incre_x;
        pyc = f_ratio * d.incre_y;
    }
    const float roll = rxc - f_rot_x_end, pitch = ryc - f_rot_y_end, yaw = rzc - f_rot_z_end;
    const float tx = pxc - d.incre_x, ty = pyc - d.incre_y;
    const float tz = rzc - d.incre_z; // pcm.cpp:804 uses f_rot_z_cur here (kept: drop-in parity)
    // pcl::getTransformation(x, y, z, roll, pitch, yaw), Scalar = float
    const float A = glibc_sincosf<true>(yaw), B = glibc_sincosf<false>(yaw), Cc = glibc_sincosf<true>(pitch),
                D = glibc_sincosf<false>(pitch), E = glibc_sincosf<true>(roll), F = glibc_sincosf<false>(roll);
    const float DE = D * E, DF = D * F;
    const float t00 = A * Cc, t01 = A * DF - B * E, t02 = B * F + A * DE;
    const float t10 = B * Cc, t11 = A * E + B * DF, t12 = B * DE - A * F;
    const float t20 = -D, t21 = Cc * F, t22 = Cc * E;
    out[3 * i] = t00 * x + t01 * y + t02 * z + tx;
    out[3 * i + 1] = t10 * x + t11 * y + t12 * z + ty;
    out[3 * i + 2] = t20 * x + t21 * y + t22 * z + tz;
}

// ------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------
void launch_stream_refill(hipStream_t s, ScanDesc* scans, ScanState* st, int slots, const QueueItem* queue, const double* qT0,
                          ScanState* out_state, StreamCtrl* ctrl, int first, int save) {
    hipLaunchKernelGGL(k_stream_refill, dim3(1), dim3(1024), 0, s, scans, st, slots, queue, qT0, out_state, ctrl, first, save);
}
void launch_init_state(hipStream_t s, ScanState* st, const double* T0, int batch, int map_empty, int* active) {
    hipLaunchKernelGGL(k_init_state, dim3((batch + 63) / 64), dim3(64), 0, s, st, T0, batch, map_empty, active);
}

void launch_accumulate_radar(hipStream_t s, const DevMap& m, const ScanDesc* scans, int batch, int total_blocks, ScanState* st, double* partials,
                             const RegParams& rp) {
    if (total_blocks <= 0) return;
    const dim3 grid((unsigned)total_blocks), block(kBlock);
    switch (rp.method) {
    case ELM_GICP: hipLaunchKernelGGL(k_accumulate_radar<ELM_GICP>, grid, block, 0, s, m, scans, batch, (unsigned)total_blocks, st, partials, rp); break;
    case ELM_VGICP: hipLaunchKernelGGL(k_accumulate_radar<ELM_VGICP>, grid, block, 0, s, m, scans, batch, (unsigned)total_blocks, st, partials, rp); break;
    default: hipLaunchKernelGGL(k_accumulate_radar<ELM_AVGICP>, grid, block, 0, s, m, scans, batch, (unsigned)total_blocks, st, partials, rp); break;
    }
}
// ---- Registration::AlignCloudsLocal* on explicit pairs (elm_align_clouds_local) ---------------------------------------------------------
// The reference's public step functions (reg.cpp:15-66 P2P, :68-152 GICP, :154-225 VGICP / AVGICP): given the pairs -- source points in the
// SENSOR frame, targets and their covariances in the map frame -- and last_icp_pose, the LM-damped Gauss-Newton step as a 4x4 transform.
// RunRegister never calls them here (its kernels pair and accumulate at once); a caller that holds pairs of its own does.  The pairs are
// accumulated with the reference's per-pair arithmetic in the sensor frame (add_pair / add_pair_radar: all 36 entries for the covariance
// methods, so a non-symmetric covariance behaves as in the reference: LDLT on the lower triangle, the full inverse for local_cov).
// SelfAdjointEigenSolver(cov).eigenvectors().col(0) (reg.cpp:89-91): the eigenvector of the smallest eigenvalue; only used through
// |r . n| (the fitness score).  Cyclic Jacobi on the lower triangle, the FIRST minimum of equal eigenvalues (identity covariance: e_x).
__device__ __forceinline__ void smallest_eigenvector3(const double C[9], double n[3]) {
    double A[9] = {C[0], C[3], C[6], C[3], C[4], C[7], C[6], C[7], C[8]}; // (row-major; the lower triangle mirrored)
    double V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = fabs(A[3]) + fabs(A[6]) + fabs(A[7]);
        if (off == 0.0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p * 3 + q] == 0.0) continue;
                const double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * A[p * 3 + q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) { const double akp = A[k * 3 + p], akq = A[k * 3 + q]; A[k * 3 + p] = c * akp - s * akq; A[k * 3 + q] = s * akp + c * akq; }
                for (int k = 0; k < 3; ++k) { const double apk = A[p * 3 + k], aqk = A[q * 3 + k]; A[p * 3 + k] = c * apk - s * aqk; A[q * 3 + k] = s * apk + c * aqk; }
                for (int k = 0; k < 3; ++k) { const double vkp = V[k * 3 + p], vkq = V[k * 3 + q]; V[k * 3 + p] = c * vkp - s * vkq; V[k * 3 + q] = s * vkp + c * vkq; }
            }
    }
    int best = 0;
    for (int i = 1; i < 3; ++i)
        if (A[i * 4] < A[best * 4]) best = i;
    n[0] = V[best]; n[1] = V[3 + best]; n[2] = V[6 + best];
}
template <int METHOD>
__global__ __launch_bounds__(256) void k_align_pairs(const double* __restrict__ src_local, const double* __restrict__ tgt_xyz, const double* __restrict__ tgt_cov,
                                                     const double* __restrict__ src_cov, size_t n, const AlignArgs a, double* __restrict__ partials) {
    RegParams rp{};
    rp.th = a.th; rp.th2 = a.th2;
    double acc[kRadarAcc];
#pragma unroll
    for (int k = 0; k < kRadarAcc; ++k) acc[k] = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double px = src_local[3 * i], py = src_local[3 * i + 1], pz = src_local[3 * i + 2];
        const double mx = tgt_xyz[3 * i], my = tgt_xyz[3 * i + 1], mz = tgt_xyz[3 * i + 2];
        if (METHOD == ELM_P2P) {
            double a32[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) a32[k] = 0.0;
            add_pair<ELM_P2P>(a32, a.Rinv, a.tinv, px, py, pz, mx, my, mz, nullptr, nullptr, rp);
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c) acc[r * 6 + c] += a32[r <= c ? tri(r, c) : tri(c, r)];
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[36 + k] += a32[21 + k];
            acc[42] += a32[27]; acc[43] += a32[28];
        } else {
            double C[9], Cs[9], nf[3] = {1.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 9; ++k) { C[k] = tgt_cov[9 * i + k]; Cs[k] = (a.use_src_cov && src_cov) ? src_cov[9 * i + k] : 0.0; }
            if (METHOD == ELM_GICP) smallest_eigenvector3(C, nf);
            add_pair_radar<METHOD>(acc, a.Rinv, a.tinv, px, py, pz, mx, my, mz, C, Cs, nf, rp);
        }
    }
    __shared__ double red[256 / 64][kRadarAcc];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kRadarAcc; ++k) {
        const double v = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)kRadarSums)
        partials[(size_t)blockIdx.x * kRadarSums + threadIdx.x] =
            threadIdx.x < (unsigned)kRadarAcc ? ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x] : 0.0;
}
// the step from the sums (reg.cpp:52-65 / 134-151 / 211-224): fitness, J^T M J + lambda diag, LDLT, (GICP) the inverse as local_cov, exp
__global__ __launch_bounds__(64) void k_align_solve(const double* __restrict__ partials, int n_blocks, size_t n, const AlignArgs a, double* __restrict__ out) {
    __shared__ double sums[kRadarSums];
    const int t = threadIdx.x;
    double v = 0.0;
    for (int b = 0; b < n_blocks; ++b) v += partials[(size_t)b * kRadarSums + t]; // fixed order
    sums[t] = v;
    __syncthreads();
    if (t != 0) return;
    double JTJ[36], JTr[6], A[36], x[6], cov[36], R[9];
    for (int k = 0; k < 36; ++k) JTJ[k] = sums[k];
    for (int k = 0; k < 6; ++k) JTr[k] = sums[36 + k];
    for (int k = 0; k < 36; ++k) A[k] = JTJ[k];
    for (int k = 0; k < 6; ++k) A[k * 7] = JTJ[k * 7] + a.lm_lambda * JTJ[k * 7];
    ldlt_solve6(A, JTr, x);
    for (int k = 0; k < 36; ++k) cov[k] = (k % 7 == 0) ? 1.0 : 0.0;
    if (a.method == ELM_GICP) inv6(A, cov);
    rotvec_to_matrix(x + 3, R);
    double* T = out; // column-major
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T[c * 4 + r] = R[r * 3 + c];
        T[12 + r] = x[r];
        T[r * 4 + 3] = 0.0;
    }
    T[15] = 1.0;
    for (int k = 0; k < 36; ++k) out[16 + k] = cov[k];
    out[52] = sums[42] / (double)n; // d_fitness_score_ = d_residual_sum / source_global.size()
    for (int k = 0; k < 6; ++k) out[53 + k] = x[k];
    for (int k = 0; k < 36; ++k) out[59 + k] = JTJ[k];
    for (int k = 0; k < 6; ++k) out[95 + k] = JTr[k];
    out[101] = sums[43];
}
void launch_align_pairs(hipStream_t s, const double* src_local, const double* tgt_xyz, const double* tgt_cov, const double* src_cov, size_t n,
                        const AlignArgs& a, double* partials, double* out) {
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 1024);
    if (blocks) {
        if (a.method == ELM_P2P) hipLaunchKernelGGL(k_align_pairs<ELM_P2P>, dim3(blocks), dim3(256), 0, s, src_local, tgt_xyz, tgt_cov, src_cov, n, a, partials);
        else if (a.method == ELM_GICP) hipLaunchKernelGGL(k_align_pairs<ELM_GICP>, dim3(blocks), dim3(256), 0, s, src_local, tgt_xyz, tgt_cov, src_cov, n, a, partials);
        else hipLaunchKernelGGL(k_align_pairs<ELM_VGICP>, dim3(blocks), dim3(256), 0, s, src_local, tgt_xyz, tgt_cov, src_cov, n, a, partials);
    }
    hipLaunchKernelGGL(k_align_solve, dim3(1), dim3(64), 0, s, partials, blocks, n, a, out);
}
// The query form of the plain walk (elm_map_get_correspondences without a search index, or with ELM_CHECK=query_direct as the in-product
// checker of the production search): GetCorrespondencePoints / GetCorrespondencesCov / GetCorrespondencesAllCov (vhm.cpp:31-206) on
// float64 GLOBAL-frame points -- 27 (7) hash probes per point, every bucket point, the reference's arithmetic.  q_out as RegParams::q_out.
template <int WHAT>
__global__ __launch_bounds__(256) void k_query_direct(const DevMap m, const double* __restrict__ query, size_t n, double th2, int32_t* __restrict__ q_out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double gx = query[3 * i], gy = query[3 * i + 1], gz = query[3 * i + 2];
    const int vx = floor_key(gx, m.voxel_size), vy = floor_key(gy, m.voxel_size), vz = floor_key(gz, m.voxel_size);
    double n_cand = 0.0, n_occ = 0.0;
    if (WHAT == 0) {
        double bd2 = DBL_MAX;
        float bx = 0.f, by = 0.f, bz = 0.f;
        int bidx = -1;
        if (m.n_vox) nearest_point_direct(m, vx, vy, vz, gx, gy, gz, bd2, bx, by, bz, bidx, n_cand, n_occ);
        const double dfin = (bidx >= 0) ? bd2 : (gx * gx + gy * gy) + gz * gz;
        q_out[i] = (dfin < th2) ? bidx : -2;
    } else if (WHAT == 1) {
        double bd2 = DBL_MAX, bmx = 0.0, bmy = 0.0, bmz = 0.0;
        int bvid = -1;
        if (m.n_vox) nearest_voxel_direct(m, vx, vy, vz, gx, gy, gz, bd2, bvid, bmx, bmy, bmz, n_cand, n_occ);
        const double dfin = (bvid >= 0) ? bd2 : (gx * gx + gy * gy) + gz * gz;
        q_out[i] = (dfin < th2) ? bvid : -2;
    } else {
        const int ox[7] = {0, 1, -1, 0, 0, 0, 0}, oy[7] = {0, 0, 0, 1, -1, 0, 0}, oz[7] = {0, 0, 0, 0, 0, 1, -1}; // vhm.cpp:224-230
#pragma unroll
        for (int k7 = 0; k7 < 7; ++k7) {
            int out = -2;
            if (m.n_vox) {
                const Probe pr = probe_voxel(m, vx + ox[k7], vy + oy[k7], vz + oz[k7]);
                if (pr.vid >= 0 && pr.cnt != 0) {
                    const double cx = m.vox_mean[(size_t)pr.vid * 3], cy = m.vox_mean[(size_t)pr.vid * 3 + 1], cz = m.vox_mean[(size_t)pr.vid * 3 + 2];
                    const double ex = cx - gx, ey = cy - gy, ez = cz - gz;
                    if ((ex * ex + ey * ey) + ez * ez < th2) out = pr.vid;
                }
            }
            q_out[8 * i + k7] = out;
        }
        q_out[8 * i + 7] = -2;
    }
}
void launch_query_direct(hipStream_t s, const DevMap& m, int what, const double* query, size_t n, double th2, int32_t* q_out) {
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    if (what == 0) hipLaunchKernelGGL(k_query_direct<0>, g, b, 0, s, m, query, n, th2, q_out);
    else if (what == 1) hipLaunchKernelGGL(k_query_direct<1>, g, b, 0, s, m, query, n, th2, q_out);
    else hipLaunchKernelGGL(k_query_direct<2>, g, b, 0, s, m, query, n, th2, q_out);
}
void launch_accumulate_direct(hipStream_t s, const DevMap& m, const ScanDesc* scans, int batch, int total_blocks,
                              ScanState* st, double* partials, const RegParams& rp) {
    dim3 g(total_blocks), b(kBlock);
#define ELM_LAUNCH(K, M) hipLaunchKernelGGL((K<M>), g, b, 0, s, m, scans, batch, (unsigned)total_blocks, st, partials, rp)
    switch (rp.method) {
    case ELM_P2P: ELM_LAUNCH(k_accumulate_direct, ELM_P2P); break;
    case ELM_GICP: ELM_LAUNCH(k_accumulate_direct, ELM_GICP); break;
    case ELM_VGICP: ELM_LAUNCH(k_accumulate_direct, ELM_VGICP); break;
    default: ELM_LAUNCH(k_accumulate_direct, ELM_AVGICP); break;
    }
#undef ELM_LAUNCH
}

void launch_accumulate_cell(hipStream_t s, const DevMap& m, const ScanDesc* scans, int batch, int total_blocks,
                            ScanState* st, double* partials, const RegParams& rp) {
    dim3 g(total_blocks), b(kBlock);
    if (rp.method == ELM_P2P)
        hipLaunchKernelGGL((k_accumulate_cell<ELM_P2P>), g, b, 0, s, m, scans, batch, (unsigned)total_blocks, st, partials, rp);
    else
        hipLaunchKernelGGL((k_accumulate_cell<ELM_GICP>), g, b, 0, s, m, scans, batch, (unsigned)total_blocks, st, partials, rp);
}
void launch_accumulate_grid(hipStream_t s, const DevMap& m, const ScanDesc* scans, int batch, int total_blocks,
                            ScanState* st, double* partials, const RegParams& rp) {
    dim3 g(total_blocks), b(kBlock);
#define ELM_LAUNCH_GW(M, C, T, S_, W_) hipLaunchKernelGGL((k_accumulate_grid<M, C, T, S_, W_>), g, b, 0, s, m, scans, batch, (unsigned)total_blocks, st, partials, rp)
#define ELM_LAUNCH_G(M, C, T)                                   \
    do {                                                        \
        if (m.grid_wide) {                                      \
            if (rp.stats) ELM_LAUNCH_GW(M, C, T, 1, 1);         \
            else ELM_LAUNCH_GW(M, C, T, 0, 1);                  \
        } else {                                                \
            if (rp.stats) ELM_LAUNCH_GW(M, C, T, 1, 0);         \
            else ELM_LAUNCH_GW(M, C, T, 0, 0);                  \
        }                                                       \
    } while (0)
    // index form (template parameter TILED): 0 dense grid, 1 two-level grid
#define ELM_LAUNCH_GT(M, C)                                                        \
    do {                                                                           \
        if (m.grid_tiled) ELM_LAUNCH_G(M, C, 1);                                   \
        else ELM_LAUNCH_G(M, C, 0);                                                \
    } while (0)
    if (rp.query) { // elm_map_get_correspondences: the search of the P2P kernel alone (STATS = 2)
        if (m.grid_wide) {
            if (m.grid_tiled) ELM_LAUNCH_GW(ELM_P2P, 0, 1, 2, 1); else ELM_LAUNCH_GW(ELM_P2P, 0, 0, 2, 1);
        } else {
            if (m.grid_tiled) ELM_LAUNCH_GW(ELM_P2P, 0, 1, 2, 0); else ELM_LAUNCH_GW(ELM_P2P, 0, 0, 2, 0);
        }
    }
    else if (rp.method == ELM_P2P) ELM_LAUNCH_GT(ELM_P2P, 0);
    else if (m.gicp_compact == 2) ELM_LAUNCH_GT(ELM_GICP, 2);
    else if (m.gicp_compact) ELM_LAUNCH_GT(ELM_GICP, 1);
    else ELM_LAUNCH_GT(ELM_GICP, 0);
#undef ELM_LAUNCH_GT
#undef ELM_LAUNCH_G
#undef ELM_LAUNCH_GW
}
void launch_gather_gicp(hipStream_t s, const DevMap& m, size_t n_slots, double* out, int compact) {
    const size_t threads = n_slots * (compact ? 8 : 16);
    if (compact) hipLaunchKernelGGL(k_gather_gicp<1>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, m, n_slots, out);
    else hipLaunchKernelGGL(k_gather_gicp<0>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, m, n_slots, out);
}
void launch_vox_stat(hipStream_t s, const DevMap& m, uint32_t* out) {
    const size_t n = (size_t)m.vnx * m.vny * m.vnz;
    hipLaunchKernelGGL(k_vox_stat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, m, out);
}
int stream_max_slots() { return kMaxSlots; }
void launch_accumulate_vnbr(hipStream_t s, const DevMap& m, const ScanDesc* scans, int batch, int total_blocks,
                            ScanState* st, double* partials, const RegParams& rp) {
#define ELM_LAUNCH_VF(M, C, S_, F_) hipLaunchKernelGGL((k_accumulate_vnbr<M, C, S_, F_>), dim3(total_blocks), dim3(kBlock), 0, s, m, scans, batch, (unsigned)total_blocks, st, partials, rp)
#define ELM_LAUNCH_V(M, C)                                                         \
    do {                                                                           \
        const bool faces_ = (M) == ELM_AVGICP && m.vq_dense && m.vqf_dense;        \
        /* the fused walk; on a map with flagged voxels it needs the workgroup flags and partial records it can add to */ \
        if (faces_ && (C) && m.vface_plain && (!m.vface_flagged || rp.flagged)) { \
            if (!m.vface_flagged) {                                                \
                if (rp.stats) ELM_LAUNCH_VF(M, C, 1, ((M) == ELM_AVGICP && (C) ? 2 : 0)); \
                else ELM_LAUNCH_VF(M, C, 0, ((M) == ELM_AVGICP && (C) ? 2 : 0));   \
            } else {                                                               \
                if (rp.stats) ELM_LAUNCH_VF(M, C, 1, ((M) == ELM_AVGICP && (C) ? 4 : 0)); \
                else ELM_LAUNCH_VF(M, C, 0, ((M) == ELM_AVGICP && (C) ? 4 : 0));   \
                if (m.vface_flagged == 1) ELM_LAUNCH_VF(M, C, 0, ((M) == ELM_AVGICP && (C) ? 3 : 0)); \
            }                                                                      \
        } else if (faces_) {                                                       \
            if (rp.stats) ELM_LAUNCH_VF(M, C, 1, ((M) == ELM_AVGICP ? 1 : 0));     \
            else ELM_LAUNCH_VF(M, C, 0, ((M) == ELM_AVGICP ? 1 : 0));              \
        } else {                                                                   \
            if (rp.stats) ELM_LAUNCH_VF(M, C, 1, 0);                               \
            else ELM_LAUNCH_VF(M, C, 0, 0);                                        \
        }                                                                          \
    } while (0)
    if (rp.query) { // elm_map_get_correspondences: the walk alone (STATS = 2), full records (the voxel ids and position codes)
        if (rp.method == ELM_VGICP) ELM_LAUNCH_VF(ELM_VGICP, 0, 2, 0);
        else ELM_LAUNCH_VF(ELM_AVGICP, 0, 2, 0);
    }
    else if (rp.method == ELM_VGICP) {
        if (m.vox_compact == 2) ELM_LAUNCH_V(ELM_VGICP, 2);
        else if (m.vox_compact) ELM_LAUNCH_V(ELM_VGICP, 1);
        else ELM_LAUNCH_V(ELM_VGICP, 0);
    }
    else { if (m.vox_compact) ELM_LAUNCH_V(ELM_AVGICP, 1); else ELM_LAUNCH_V(ELM_AVGICP, 0); }
#undef ELM_LAUNCH_V
#undef ELM_LAUNCH_VF
}
// map build: the face neighbours (and the voxel itself) of every voxel-mean list, in list order (AVGICP's pairs)
__device__ __forceinline__ bool is_face_code(int code) {
    return code == 13 || code == 22 || code == 4 || code == 16 || code == 10 || code == 14 || code == 12;
}
__global__ __launch_bounds__(256) void k_vface(const DevMap m, const uint32_t* __restrict__ offsets,
                                               const uint32_t* __restrict__ counts, unsigned n_q, uint32_t* __restrict__ face_cnt,
                                               const uint32_t* __restrict__ face_off, VoxRec* __restrict__ out, int plain) {
    const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_q) return;
    unsigned n = 0;
    const unsigned o = out ? face_off[q] : 0u;
    for (unsigned j = 0; j < counts[q]; ++j) {
        const unsigned vc = (unsigned)vnbr_vc(m, offsets[q] + j);
        if (!is_face_code((int)(vc >> kVidBits))) continue;
        VoxRec r = m.vox_rec[vc & kVidMask];
        r.pad = (int32_t)(vc >> kVidBits);
        // the first 48 bytes tell the three kinds of record apart (k_accumulate_vnbr<AVGICP> loads only those in the common case):
        //   k = kCompactK (regularised covariance): the unit normal as it is;  k = 0 (identity): n.x = 2 (not a unit vector);
        //   anything else (k = NaN: outside the compact form, or another k): n.x = NaN -> the stored inverse is read
        // plain (DevMap::vface_plain: no voxel outside the compact form, the fused walk): an identity covariance is written as the ZERO
        // normal -- I + 999 * 0 0^T -- so that walk treats every record alike
        if (r.k == 0.0) { r.nx = plain ? 0.0 : 2.0; if (plain) r.ny = r.nz = 0.0; }
        else if (!(fabs(r.k - kCompactK) <= 1e-7)) r.nx = __builtin_nan("");
        if (out) out[o + n] = r;
        ++n;
    }
    if (!out) face_cnt[q] = n;
}
void launch_vface(hipStream_t s, const DevMap& m, const uint32_t* offsets, const uint32_t* counts, uint32_t n_q, uint32_t* face_cnt,
                  const uint32_t* face_off, VoxRec* out, int plain) {
    if (n_q) hipLaunchKernelGGL(k_vface, dim3((n_q + 255) / 256), dim3(256), 0, s, m, offsets, counts, n_q, face_cnt, face_off, out, plain);
}
void launch_vnbr_fill(hipStream_t s, const DevMap& m, const int32_t* qkeys, uint32_t n_q, const uint32_t* offsets, VoxBlk* out_blk) {
    hipLaunchKernelGGL(k_vnbr_fill, dim3((n_q + 255) / 256), dim3(256), 0, s, m, qkeys, n_q, offsets, out_blk);
}
void launch_vox_rec_fill(hipStream_t s, const DevMap& m, VoxRec* out) {
    if (m.n_vox) hipLaunchKernelGGL(k_vox_rec_fill, dim3((m.n_vox + 255) / 256), dim3(256), 0, s, m, out);
}
void launch_nbr_cellsort(hipStream_t s, const DevMap& m, const int32_t* qkeys, uint32_t n_q, const uint32_t* offsets,
                         const uint32_t* counts, Pt3* pts, uint32_t* idx, uint16_t* cell_off) {
    hipLaunchKernelGGL(k_nbr_cellsort, dim3(n_q), dim3(64), 0, s, m, qkeys, n_q, offsets, counts, pts, idx, cell_off);
}
size_t nbr_cell_stride() { return (size_t)kCellStride; }
void launch_nbr_count(hipStream_t s, const DevMap& m, const int32_t* qkeys, uint32_t n_q, uint32_t* counts, uint32_t* nocc) {
    hipLaunchKernelGGL(k_nbr_count, dim3((n_q + 255) / 256), dim3(256), 0, s, m, qkeys, n_q, counts, nocc);
}
void launch_nbr_fill(hipStream_t s, const DevMap& m, const int32_t* qkeys, uint32_t n_q, const uint32_t* offsets, Pt3* out,
                     uint32_t* out_idx) {
    const uint64_t threads = (uint64_t)n_q * 32;
    hipLaunchKernelGGL(k_nbr_fill, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, m, qkeys, n_q, offsets, out, out_idx);
}

void launch_solve(hipStream_t s, const ScanDesc* scans, int batch, ScanState* st, const double* partials,
                  double* sums, const RegParams& rp, elm_iter_trace* trace, int mode, int* active, const StreamArgs* refill) {
    StreamArgs sa = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
    if (refill) sa = *refill;
    // one wavefront per scan when the sums are already reduced (the second half of a multi-rank iteration)
    const bool small = rp.solve_small != 0 && rp.radar == 0;
    const int threads = mode == 2 ? 64 : (small ? 256 : kSolveThreads);
    if (small) hipLaunchKernelGGL(k_solve<256>, dim3(batch), dim3(threads), 0, s, scans, st, partials, sums, rp, trace, mode, active, sa);
    else hipLaunchKernelGGL(k_solve<kSolveThreads>, dim3(batch), dim3(threads), 0, s, scans, st, partials, sums, rp, trace, mode, active, sa);
}

void launch_voxel_cov(hipStream_t s, const DevMap& m, const uint2* ranges, double* vox_mean, double* vox_cov, double* vox_cinv, double* vox_nk, unsigned* bad) {
    hipLaunchKernelGGL(k_voxel_cov, dim3((m.n_vox + 255) / 256), dim3(256), 0, s, m, ranges, vox_mean, vox_cov, vox_cinv, vox_nk, bad);
}

void launch_point_cov(hipStream_t s, const DevMap& m, double d2max, double* pt_gicp, double* pt_cov, unsigned* bad) {
    hipLaunchKernelGGL(k_point_cov, dim3((m.n_pts + 255) / 256), dim3(256), 0, s, m, d2max, pt_gicp, pt_cov, bad);
}

// ------------------------------------------------------------------------------------------------------
// VoxelDownsample on the device (vhm.hpp:260-283): the first point (smallest input index) of every floor-keyed voxel, kept
// in input order.  Used by the node callback so that the deskewed scan never leaves HBM before it is registered.
//   k_ds_insert: packed 64-bit key per point -> open-addressing table (CAS), atomicMin of the index per voxel
//   k_ds_count / k_ds_offsets / k_ds_scatter: ordered compaction (block counts -> exclusive scan -> scatter)
// ------------------------------------------------------------------------------------------------------
constexpr int kDsBlock = 1024;
__device__ __forceinline__ bool ds_key(const float* __restrict__ xyz, unsigned i, double vs, unsigned long long& key) {
    const double qx = (double)xyz[3 * i] / vs, qy = (double)xyz[3 * i + 1] / vs, qz = (double)xyz[3 * i + 2] / vs;
    const double lim = 1048576.0; // 2^20: three 21-bit fields
    if (!(qx > -lim && qx < lim && qy > -lim && qy < lim && qz > -lim && qz < lim)) return false;
    const long long kx = (long long)floor(qx) + 1048576, ky = (long long)floor(qy) + 1048576, kz = (long long)floor(qz) + 1048576;
    key = ((unsigned long long)kx << 42) | ((unsigned long long)ky << 21) | (unsigned long long)kz;
    return true;
}
__global__ __launch_bounds__(256) void k_ds_insert(const float* __restrict__ xyz, unsigned n, double vs, unsigned long long* table,
                                                   unsigned* first, unsigned cap_log2, unsigned* __restrict__ slot, int* overflow) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long key;
    if (!ds_key(xyz, i, vs, key)) { atomicExch(overflow, 1); slot[i] = 0; return; }
    const unsigned mask = (1u << cap_log2) - 1u;
    unsigned h = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> (64 - cap_log2));
    for (;;) {
        const unsigned long long prev = atomicCAS(&table[h], ~0ull, key);
        if (prev == ~0ull || prev == key) break;
        h = (h + 1) & mask;
    }
    atomicMin(&first[h], i);
    slot[i] = h;
}
__global__ __launch_bounds__(kDsBlock) void k_ds_count(const unsigned* __restrict__ first, const unsigned* __restrict__ slot, unsigned n,
                                                       unsigned* __restrict__ block_count) {
    __shared__ unsigned s_cnt[kDsBlock / 64];
    const unsigned i = blockIdx.x * kDsBlock + threadIdx.x;
    const bool keep = i < n && first[slot[i]] == i;
    const unsigned long long b = __ballot(keep);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = (unsigned)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int w = 0; w < kDsBlock / 64; ++w) t += s_cnt[w];
        block_count[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(1024) void k_ds_offsets(unsigned* block_count, unsigned n_blocks, unsigned* total) { // in-place exclusive scan
    __shared__ unsigned s[1024];
    unsigned carry = 0;
    for (unsigned base = 0; base < n_blocks; base += 1024) {
        const unsigned j = base + threadIdx.x;
        const unsigned v = j < n_blocks ? block_count[j] : 0u;
        s[threadIdx.x] = v;
        __syncthreads();
        for (unsigned off = 1; off < 1024; off <<= 1) {
            const unsigned t = threadIdx.x >= off ? s[threadIdx.x - off] : 0u;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (j < n_blocks) block_count[j] = carry + s[threadIdx.x] - v;
        const unsigned chunk_total = s[1023];
        __syncthreads();
        carry += chunk_total;
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(kDsBlock) void k_ds_scatter(const float* __restrict__ xyz, const unsigned* __restrict__ first,
                                                         const unsigned* __restrict__ slot, unsigned n,
                                                         const unsigned* __restrict__ block_offset, Pt3* __restrict__ out) {
    __shared__ unsigned s_cnt[kDsBlock / 64];
    const unsigned i = blockIdx.x * kDsBlock + threadIdx.x;
    const bool keep = i < n && first[slot[i]] == i;
    const unsigned long long b = __ballot(keep);
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wave] = (unsigned)__popcll(b);
    __syncthreads();
    unsigned pos = block_offset[blockIdx.x];
    for (unsigned w = 0; w < wave; ++w) pos += s_cnt[w];
    pos += (unsigned)__popcll(b & ((1ull << lane) - 1ull));
    if (keep) {
        Pt3 q;
        q.x = xyz[3 * i]; q.y = xyz[3 * i + 1]; q.z = xyz[3 * i + 2];
        out[pos] = q;
    }
}

// leaves the table as it was found (all ones): only the slots this scan touched are rewritten, instead of a memset of the whole
// table (3 MB for a 131 072-point scan) before every scan
__global__ __launch_bounds__(256) void k_ds_clear(const unsigned* __restrict__ slot, unsigned n, unsigned long long* table, unsigned* first) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned h = slot[i];
    table[h] = ~0ull;
    first[h] = ~0u;
}
void launch_voxel_downsample(hipStream_t s, const float* xyz, uint32_t n, double vs, unsigned long long* table, unsigned* first,
                             unsigned cap_log2, unsigned* slot, unsigned* block_count, unsigned* total, int* overflow, Pt3* out) {
    const unsigned nb = (n + kDsBlock - 1) / kDsBlock;
    hipLaunchKernelGGL(k_ds_insert, dim3((n + 255) / 256), dim3(256), 0, s, xyz, n, vs, table, first, cap_log2, slot, overflow);
    hipLaunchKernelGGL(k_ds_count, dim3(nb), dim3(kDsBlock), 0, s, first, slot, n, block_count);
    hipLaunchKernelGGL(k_ds_offsets, dim3(1), dim3(1024), 0, s, block_count, nb, total);
    hipLaunchKernelGGL(k_ds_scatter, dim3(nb), dim3(kDsBlock), 0, s, xyz, first, slot, n, block_count, out);
    hipLaunchKernelGGL(k_ds_clear, dim3((n + 255) / 256), dim3(256), 0, s, slot, n, table, first);
}

// ------------------------------------------------------------------------------------------------------
// Scan ordering on the device: the points of a scan along a Hilbert curve over 2 m x 2 m sensor-frame cells (all heights of a cell
// together), so that every 256-point workgroup of the accumulate kernels -- and, through the XCD-aware block mapping, every XCD's
// L2 -- touches a few adjacent map cells.  Source order is not contractual (the reference's own VoxelDownsample emits
// unordered_map order, vhm.hpp:278-280); the result must only be DETERMINISTIC (the summation tree follows the point order).
// One workgroup per scan (kOrderWaves wavefronts), a counting sort without atomics:
//   A  wave w owns the contiguous points [w C, (w + 1) C); 64 consecutive points per step (coalesced 12-byte loads).  Lanes with the
//      same 12-bit key find each other with 12 ballots; a point's rank inside its (wave, key) run = the wave's counter for that key
//      (16-bit, LDS) + the number of lower lanes of its group; the group's last lane writes the counter back.  The 32-bit word
//      key | rank << 12 goes to scratch.
//   -  exclusive prefix of the counters over (key, wave): start of every (key, wave) run
//   B  every point moves to start[key] + offset[wave][key] + rank.
// Within a key the order is (wave, step, lane) = the caller's order: the sort is stable.  A (degenerate) scan with more than 65535
// points in one cell or per wave keeps the caller's order.
constexpr int kOrderWaves = 8; // 512 threads, 80 KB of LDS (64 KB of counters + run starts): co-resides with the accumulate workgroups of the compute stream;
                           // host-fed stream: 8 -> 32.4k, 16 (144 KB: waits for an empty CU) -> 31.0k registrations/s
constexpr int kOrderThreads = kOrderWaves * 64;
constexpr int kOrderBins = kOrderCells * kOrderCells; // 4096 keys: kOrderWaves x 8 KB of 16-bit counters + 16 KB of run starts
__device__ __forceinline__ unsigned order_key(const Pt3 p, const unsigned short* lut) {
    const int cx = (int)floorf(p.x * 0.5f) + kOrderCells / 2, cy = (int)floorf(p.y * 0.5f) + kOrderCells / 2;
    const int ux = min(max(cx, 0), kOrderCells - 1), uy = min(max(cy, 0), kOrderCells - 1);
    return lut[uy * kOrderCells + ux];
}
__global__ __launch_bounds__(kOrderThreads) void k_scan_order(const OrderJob* __restrict__ jobs, const uint16_t* __restrict__ hilbert_lut) {
    __shared__ unsigned short s_cnt[kOrderWaves][kOrderBins];
    __shared__ unsigned s_start[kOrderBins]; // phase A: the Hilbert table (16-bit entries) lives here
    __shared__ unsigned s_wsum[kOrderWaves];
    __shared__ int s_over;
    const OrderJob job = jobs[blockIdx.x];
    const unsigned n = job.n;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    unsigned short* s_lut = reinterpret_cast<unsigned short*>(s_start);
    for (unsigned k = tid; k < (unsigned)kOrderBins; k += kOrderThreads) {
        s_lut[k] = hilbert_lut[k];
#pragma unroll
        for (int w = 0; w < kOrderWaves; ++w) s_cnt[w][k] = 0;
    }
    if (tid == 0) s_over = 0;
    __syncthreads();
    const unsigned chunk = ((n + kOrderThreads - 1) / kOrderThreads) * 64u; // points per wave, a multiple of 64
    const unsigned w0 = min(n, wave * chunk), w1 = min(n, w0 + chunk);
    const bool too_long = chunk > 65535u; // uniform: the 16-bit counters cannot hold a wave's run
    if (!too_long) {
        Pt3 nxt;
        nxt.x = nxt.y = nxt.z = 0.f;
        if (w0 + lane < w1) nxt = job.src[w0 + lane];
        for (unsigned base = w0; base < w1; base += 64u) {
            const unsigned i = base + lane;
            const bool valid = i < w1;
            const Pt3 p = nxt;
            if (i + 64u < w1) nxt = job.src[i + 64u]; // next step's point is in flight while this one is ranked
            const unsigned key = valid ? order_key(p, s_lut) : 0u;
            unsigned long long grp = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 12; ++b) {
                const bool bit = (key >> b) & 1u;
                const unsigned long long bal = __ballot(bit);
                grp &= bit ? bal : ~bal;
            }
            const unsigned below = (unsigned)__popcll(grp & ((1ull << lane) - 1ull)), size = (unsigned)__popcll(grp);
            const unsigned c = s_cnt[wave][key]; // every lane of the group reads the counter before its last lane writes it back
            if (valid && below + 1u == size) s_cnt[wave][key] = (unsigned short)(c + size);
            if (valid) job.tmp[i] = key | ((c + below) << 12);
        }
    }
    __syncthreads();
    // exclusive prefix over (key, wave): every thread takes KPT consecutive keys
    constexpr int KPT = kOrderBins / kOrderThreads;
    static_assert(KPT * kOrderThreads == kOrderBins, "keys per thread");
    unsigned tot = 0;
    int over = too_long ? 1 : 0;
    unsigned t4[KPT];
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
        const unsigned key = tid * (unsigned)KPT + (unsigned)q;
        unsigned run = 0;
#pragma unroll
        for (int w = 0; w < kOrderWaves; ++w) {
            const unsigned c = s_cnt[w][key];
            s_cnt[w][key] = (unsigned short)run;
            run += c;
        }
        if (run > 65535u) over = 1; // a cell with more than 65535 points: the scan keeps the caller's order (k_order_prefix: the same rule)
        t4[q] = tot;
        tot += run;
    }
    unsigned inc = tot; // inclusive scan over the wavefront
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = (unsigned)__shfl_up((int)inc, off, 64);
        if (lane >= (unsigned)off) inc += o;
    }
    if (lane == 63u) s_wsum[wave] = inc;
    if (over) s_over = 1;
    __syncthreads(); // also: the last read of the Hilbert table is behind us
    unsigned wbase = 0;
#pragma unroll
    for (int w = 0; w < kOrderWaves; ++w) wbase += ((unsigned)w < wave) ? s_wsum[w] : 0u;
    const unsigned ex = wbase + inc - tot;
#pragma unroll
    for (int q = 0; q < KPT; ++q) s_start[tid * (unsigned)KPT + (unsigned)q] = ex + t4[q];
    __syncthreads();
    const bool identity = s_over != 0; // uniform
    if (identity) {
        for (unsigned i = tid; i < n; i += kOrderThreads) job.dst[i] = job.src[i];
        return;
    }
    for (unsigned base = w0; base < w1; base += 128u) { // two steps per trip: four loads in flight per lane
        const unsigned i0 = base + lane, i1 = base + 64u + lane;
        const bool v0 = i0 < w1, v1 = i1 < w1;
        Pt3 p0, p1;
        unsigned d0 = 0, d1 = 0;
        if (v0) { p0 = job.src[i0]; d0 = job.tmp[i0]; }
        if (v1) { p1 = job.src[i1]; d1 = job.tmp[i1]; }
        if (v0) job.dst[s_start[d0 & 4095u] + s_cnt[wave][d0 & 4095u] + (d0 >> 12)] = p0;
        if (v1) job.dst[s_start[d1 & 4095u] + s_cnt[wave][d1 & 4095u] + (d1 >> 12)] = p1;
    }
}
void launch_scan_order(hipStream_t s, const OrderJob* jobs, int n_jobs, const uint16_t* hilbert_lut) {
    if (n_jobs > 0) hipLaunchKernelGGL(k_scan_order, dim3(n_jobs), dim3(kOrderThreads), 0, s, jobs, hilbert_lut);
}

// ---- the same ordering of ONE scan by many workgroups (a scan uploaded on its own: k_scan_order's single workgroup takes 0.29 ms for
// 131 072 points on one CU).  Four launches, the same stable counting sort, hence the same bytes as k_scan_order:
//   k_order_rank     workgroup g owns the contiguous points [g C, (g + 1) C) (4 wavefronts, contiguous quarters): rank of every point
//                    inside its (workgroup, key) run -- per-wave ballot ranking as above, then the exclusive prefix over the four
//                    waves -- into tmp[i] = key | rank << 12, the workgroup's per-key totals into hist[g][key]
//   k_order_prefix   start[g][key] = points of this key in workgroups before g (16 workgroups x 256 keys), tot[key]; flag = a key with more
//                    than 65535 points (the scan keeps the caller's order)
//   k_order_base     one workgroup: base[key] = points of smaller keys; flag also for a scan beyond k_scan_order's size limit
//   k_order_scatter  dst[start[g][key] + base[key] + rank] = src[i]
constexpr int kWideWaves = 4, kWideThreads = kWideWaves * 64;
__global__ __launch_bounds__(kWideThreads) void k_order_rank(const OrderJob* __restrict__ jobs, const uint16_t* __restrict__ hilbert_lut, unsigned chunk,
                                                            uint16_t* __restrict__ hist) {
    __shared__ unsigned short s_cnt[kWideWaves][kOrderBins];
    __shared__ unsigned short s_lut[kOrderBins];
    const OrderJob job = jobs[0];
    const unsigned n = job.n;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, g = blockIdx.x;
    for (unsigned k = tid; k < (unsigned)kOrderBins; k += kWideThreads) {
        s_lut[k] = hilbert_lut[k];
#pragma unroll
        for (int w = 0; w < kWideWaves; ++w) s_cnt[w][k] = 0;
    }
    __syncthreads();
    const unsigned sub = chunk / kWideWaves; // a multiple of 64
    const unsigned w0 = min(n, g * chunk + wave * sub), w1 = min(n, w0 + sub);
    for (unsigned base = w0; base < w1; base += 64u) {
        const unsigned i = base + lane;
        const bool valid = i < w1;
        Pt3 p;
        p.x = p.y = p.z = 0.f;
        if (valid) p = job.src[i];
        const unsigned key = valid ? order_key(p, s_lut) : 0u;
        unsigned long long grp = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 12; ++b) {
            const bool bit = (key >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            grp &= bit ? bal : ~bal;
        }
        const unsigned below = (unsigned)__popcll(grp & ((1ull << lane) - 1ull)), size = (unsigned)__popcll(grp);
        const unsigned c = s_cnt[wave][key];
        if (valid && below + 1u == size) s_cnt[wave][key] = (unsigned short)(c + size);
        if (valid) job.tmp[i] = key | ((c + below) << 12);
    }
    __syncthreads();
    for (unsigned key = tid; key < (unsigned)kOrderBins; key += kWideThreads) { // exclusive prefix over the waves; totals out
        unsigned run = 0;
#pragma unroll
        for (int w = 0; w < kWideWaves; ++w) {
            const unsigned c = s_cnt[w][key];
            s_cnt[w][key] = (unsigned short)run;
            run += c;
        }
        hist[(size_t)g * kOrderBins + key] = (uint16_t)run;
    }
    __syncthreads();
    for (unsigned i = w0 + lane; i < w1; i += 64u) { // (the wave re-reads what it wrote itself)
        const unsigned d = job.tmp[i];
        job.tmp[i] = (d & 4095u) | (((d >> 12) + s_cnt[wave][d & 4095u]) << 12);
    }
}
// (a) per key, over the workgroups: 16 workgroups x 256 keys, the G loads of a key sixteen at a time
__global__ __launch_bounds__(256) void k_order_prefix(const uint16_t* __restrict__ hist, unsigned G, unsigned* __restrict__ start, unsigned* __restrict__ tot,
                                                     int* __restrict__ flag) {
    const unsigned key = blockIdx.x * 256u + threadIdx.x;
    unsigned run = 0;
    for (unsigned g0 = 0; g0 < G; g0 += 16u) { // sixteen independent loads in flight, then the running sum
        unsigned c[16];
#pragma unroll
        for (unsigned u = 0; u < 16u; ++u) c[u] = (g0 + u < G) ? hist[(size_t)(g0 + u) * kOrderBins + key] : 0u;
#pragma unroll
        for (unsigned u = 0; u < 16u; ++u) {
            if (g0 + u < G) start[(size_t)(g0 + u) * kOrderBins + key] = run;
            run += c[u];
        }
    }
    tot[key] = run;
    if (run > 65535u) atomicOr(flag, 1); // a cell with more than 65535 points: the scan keeps the caller's order (k_scan_order's rule)
}
// (b) over the keys: base[key] = points of smaller keys (one workgroup, four consecutive keys per thread)
__global__ __launch_bounds__(1024) void k_order_base(const OrderJob* __restrict__ jobs, const unsigned* __restrict__ tot, unsigned* __restrict__ base,
                                                    int* __restrict__ flag) {
    __shared__ unsigned s_wsum[16];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint4 t = reinterpret_cast<const uint4*>(tot)[tid];
    const unsigned t4[4] = {0u, t.x, t.x + t.y, t.x + t.y + t.z};
    const unsigned sum = t.x + t.y + t.z + t.w;
    unsigned inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = (unsigned)__shfl_up((int)inc, off, 64);
        if (lane >= (unsigned)off) inc += o;
    }
    if (lane == 63u) s_wsum[wave] = inc;
    __syncthreads();
    unsigned wbase = 0;
#pragma unroll
    for (unsigned w = 0; w < 16u; ++w) wbase += (w < wave) ? s_wsum[w] : 0u;
    const unsigned ex = wbase + inc - sum;
    reinterpret_cast<uint4*>(base)[tid] = make_uint4(ex + t4[0], ex + t4[1], ex + t4[2], ex + t4[3]);
    if (tid == 0 && ((jobs[0].n + kOrderThreads - 1) / kOrderThreads) * 64u > 65535u) atomicOr(flag, 1); // k_scan_order's own size limit (its `too_long`)
}
__global__ __launch_bounds__(kWideThreads) void k_order_scatter(const OrderJob* __restrict__ jobs, unsigned chunk, const unsigned* __restrict__ start,
                                                               const unsigned* __restrict__ base, const int* __restrict__ flag) {
    __shared__ unsigned s_start[kOrderBins];
    const OrderJob job = jobs[0];
    const unsigned n = job.n, tid = threadIdx.x, g = blockIdx.x;
    const unsigned i0 = min(n, g * chunk), i1 = min(n, i0 + chunk);
    if (*flag) { // degenerate scan: the caller's order
        for (unsigned i = i0 + tid; i < i1; i += kWideThreads) job.dst[i] = job.src[i];
        return;
    }
    for (unsigned k = tid; k < (unsigned)kOrderBins; k += kWideThreads) s_start[k] = start[(size_t)g * kOrderBins + k] + base[k];
    __syncthreads();
    for (unsigned i = i0 + tid; i < i1; i += kWideThreads) {
        const unsigned d = job.tmp[i];
        job.dst[s_start[d & 4095u] + (d >> 12)] = job.src[i];
    }
}
// scratch: order_wide_scratch_bytes(n) behind the n words of job.tmp
unsigned order_wide_groups(unsigned n) {
    unsigned G = (n + 2047u) / 2048u;
    return G < 2u ? 2u : (G > 64u ? 64u : G);
}
size_t order_wide_scratch_bytes(unsigned n) { return (size_t)order_wide_groups(n) * kOrderBins * 6 + (size_t)kOrderBins * 8 + 64; }
void launch_scan_order_wide(hipStream_t s, const OrderJob* job, unsigned n, const uint16_t* hilbert_lut, void* scratch) {
    const unsigned G = order_wide_groups(n);
    unsigned chunk = (n + G - 1) / G;
    chunk = ((chunk + kWideThreads - 1) / kWideThreads) * kWideThreads; // whole 64-point steps per wave
    unsigned* start = (unsigned*)scratch;
    unsigned* tot = start + (size_t)G * kOrderBins;
    unsigned* base = tot + kOrderBins;
    uint16_t* hist = (uint16_t*)(base + kOrderBins);
    int* flag = (int*)((char*)scratch + (size_t)G * kOrderBins * 6 + (size_t)kOrderBins * 8);
    (void)hipMemsetAsync(flag, 0, sizeof(int), s);
    hipLaunchKernelGGL(k_order_rank, dim3(G), dim3(kWideThreads), 0, s, job, hilbert_lut, chunk, hist);
    hipLaunchKernelGGL(k_order_prefix, dim3(kOrderBins / 256), dim3(256), 0, s, (const uint16_t*)hist, G, start, tot, flag);
    hipLaunchKernelGGL(k_order_base, dim3(1), dim3(1024), 0, s, job, (const unsigned*)tot, base, flag);
    hipLaunchKernelGGL(k_order_scatter, dim3(G), dim3(kWideThreads), 0, s, job, chunk, (const unsigned*)start, (const unsigned*)base, (const int*)flag);
}
// host-fed streams: the upload stream publishes how many scans have landed (after their ordering kernel, same stream)
__global__ void k_publish_ready(StreamCtrl* ctrl, int ready) {
    __hip_atomic_store(&ctrl->ready, ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_publish_ready(hipStream_t s, StreamCtrl* ctrl, int ready) { hipLaunchKernelGGL(k_publish_ready, dim3(1), dim3(1), 0, s, ctrl, ready); }
// host-fed streams start with every slot idle: the solve hands out registrations as their scans arrive
__global__ void k_slots_idle(ScanDesc* scans, ScanState* st, int slots, unsigned cap_blocks) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= slots) return;
    scans[s].pts = nullptr; scans[s].n = 0; scans[s].n_total = 0;
    scans[s].blk_begin = cap_blocks * (unsigned)s; scans[s].blk_end = cap_blocks * (unsigned)(s + 1); // every slot owns cap_blocks workgroups
    st[s].done = 1;
    st[s].reg = -1;
}
void launch_slots_idle(hipStream_t s, ScanDesc* scans, ScanState* st, int slots, unsigned cap_blocks) {
    hipLaunchKernelGGL(k_slots_idle, dim3((slots + 255) / 256), dim3(256), 0, s, scans, st, slots, cap_blocks);
}

void launch_deskew(hipStream_t s, const float* xyz, const float* rel_time, uint32_t n, const DeskewDev& d, float* xyz_out) {
    hipLaunchKernelGGL(k_deskew, dim3((n + 255) / 256), dim3(256), 0, s, xyz, rel_time, n, d, xyz_out);
}

} // namespace elm
