// elm_dev_reduce.hpp -- block reductions, factored pair sums, side sums, the P2P pair, lane-group operations
// Device-side helpers shared by the kernel translation units (every function is inline / a template: no symbol is emitted by itself).
#pragma once
#include <float.h>
#include <hip/hip_runtime.h>

#include "elm_internal.hpp"
#include "elm_la.hpp"
#include "elm_dev_pairs.hpp"

namespace elm {

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

} // namespace elm
