// elm_kernels.hip -- hand-written HIP kernels for gfx950 (CDNA4 / MI355X).
//
//   K1  k_accumulate<METHOD>  fused  T*p -> floor key -> 27(7)-voxel hash probe -> nearest point / voxel mean
//                             -> residual + Jacobian -> wave + block reduction of the packed normal equations
//                             (replaces TransformPoints reg.hpp:136-148, GetCorrespondence* vhm.cpp:31-206 and
//                             the serial loops of AlignCloudsLocal* reg.cpp:28-51, 85-132, 171-208)
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
#include <hip/hip_runtime.h>

#include "elm_internal.hpp"
#include "elm_la.hpp"

namespace elm {

// Optional per-phase cycle accounting of the staged accumulate kernel (build with -DELM_PHASE_TIMING; thread 0 of
// every workgroup adds its clock64() deltas).  Diagnostic builds only.
#ifdef ELM_PHASE_TIMING
__device__ unsigned long long g_phase[16];
#define ELM_PHASE_BEGIN unsigned long long ph_t0_ = clock64();
#define ELM_PHASE(k)                                                              \
    if (threadIdx.x == 0) {                                                       \
        const unsigned long long ph_t1_ = clock64();                              \
        atomicAdd(&g_phase[k], ph_t1_ - ph_t0_);                                  \
        ph_t0_ = ph_t1_;                                                          \
    }
#else
#define ELM_PHASE_BEGIN
#define ELM_PHASE(k)
#endif

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

// upper-triangle packing of the symmetric 6x6: idx(i,j), i <= j
__host__ __device__ constexpr int tri(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

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

// ------------------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------------------
// Nearest bucket point straight from global memory (one thread walks its 27 voxels).  Used by the "direct" kernel
// and as the fall-back of the staged kernel for workgroups whose points are too spread out to stage.
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

// pair payloads shared by both kernels
template <int METHOD>
__device__ __forceinline__ void finish_point_pair(double* acc, const DevMap& m, const ScanState& S, const RegParams& rp,
                                                  double px, double py, double pz, double gx, double gy, double gz,
                                                  double bd2, float bx, float by, float bz, int bidx) {
    // no bucket at all: the reference's default PointStruct at the origin with cov I (vhm.cpp:37, QUIRK)
    const double dfin = (bidx >= 0) ? bd2 : (gx * gx + gy * gy) + gz * gz;
    if (!(dfin < rp.th2)) return;
    if (METHOD == ELM_P2P) {
        if (bidx < 0) { bx = 0.f; by = 0.f; bz = 0.f; }
        add_pair<ELM_P2P>(acc, S.Rinv, S.tinv, px, py, pz, (double)bx, (double)by, (double)bz, nullptr, nullptr, rp);
    } else {
        double C[9], mean[3], nf[3];
        if (bidx >= 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) C[k] = m.pt_gicp[(size_t)bidx * 16 + 3 + k];
#pragma unroll
            for (int k = 0; k < 3; ++k) { mean[k] = m.pt_gicp[(size_t)bidx * 16 + k]; nf[k] = m.pt_gicp[(size_t)bidx * 16 + 12 + k]; }
        } else {
            C[0] = 1; C[1] = 0; C[2] = 0; C[3] = 0; C[4] = 1; C[5] = 0; C[6] = 0; C[7] = 0; C[8] = 1;
            mean[0] = mean[1] = mean[2] = 0.0;
            nf[0] = 1.0; nf[1] = 0.0; nf[2] = 0.0;
        }
        // GICP's target position is the neighbourhood MEAN of the matched point (reg.cpp:97)
        add_pair<ELM_GICP>(acc, S.Rinv, S.tinv, px, py, pz, mean[0], mean[1], mean[2], C, nf, rp);
    }
}
__device__ __forceinline__ void finish_voxel_pair(double* acc, const DevMap& m, const ScanState& S, const RegParams& rp,
                                                  double px, double py, double pz, double gx, double gy, double gz,
                                                  double bd2, int bvid, double bmx, double bmy, double bmz) {
    const double dfin = (bvid >= 0) ? bd2 : (gx * gx + gy * gy) + gz * gz;
    if (!(dfin < rp.th2)) return;
    double C[9];
    if (bvid >= 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) C[k] = m.vox_cov[(size_t)bvid * 9 + k];
    } else {
        C[0] = 1; C[1] = 0; C[2] = 0; C[3] = 0; C[4] = 1; C[5] = 0; C[6] = 0; C[7] = 0; C[8] = 1;
        bmx = bmy = bmz = 0.0;
    }
    add_pair<ELM_VGICP>(acc, S.Rinv, S.tinv, px, py, pz, bmx, bmy, bmz, C, nullptr, rp);
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

// cross-lane helpers for the wave-cooperative stage: v_readlane for wave-uniform sources and DPP row operations instead of
// ds_bpermute (__shfl), which goes through the LDS pipeline and costs ~100 cycles per dependent step
__device__ __forceinline__ int lane_bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ unsigned lane_bcast(unsigned v, int src) { return (unsigned)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ double lane_bcast(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ double wave_min(double v) { // minimum over the 64 lanes, in every lane
    v = fmin(v, dpp_move<0xB1>(v));  // quad_perm [1,0,3,2]
    v = fmin(v, dpp_move<0x4E>(v));  // quad_perm [2,3,0,1]
    v = fmin(v, dpp_move<0x124>(v)); // row_ror:4
    v = fmin(v, dpp_move<0x128>(v)); // row_ror:8 -> every lane holds the minimum of its row of 16
    return fmin(fmin(lane_bcast(v, 0), lane_bcast(v, 16)), fmin(lane_bcast(v, 32), lane_bcast(v, 48)));
}

// Block reduction of the 32 packed sums through an LDS transpose (two halves of 16 values, 32 KB each): every
// thread stores its values column-wise, then 16 lanes per value add 16 strided columns each and finish with four
// shuffle steps.  ~40 LDS/shuffle operations per thread instead of 192 dependent shuffles.  Deterministic.
// buf must hold 16 * kBlock doubles and must not be in use by anybody (callers sync before).
__device__ __forceinline__ void block_reduce_store_lds(const double* acc, double* buf, double* __restrict__ out) {
    const int tid = threadIdx.x;
    const int k = tid >> 4, seg = tid & 15;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h) __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) buf[q * kBlock + tid] = acc[h * 16 + q];
        __syncthreads();
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) v += buf[k * kBlock + i * 16 + seg];
        // the 16 lanes of a value are one DPP row: rotations and quad permutes instead of four ds_bpermute round trips
        v += dpp_move<0x128>(v); // row_ror:8
        v += dpp_move<0x124>(v); // row_ror:4
        v += dpp_move<0x4E>(v);  // quad_perm [2,3,0,1]
        v += dpp_move<0xB1>(v);  // quad_perm [1,0,3,2]
        if (seg == 0) out[h * 16 + k] = v;
    }
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
    const unsigned i = (L - sd.blk_begin) * kBlock + threadIdx.x;
    const bool valid = i < sd.n;
    double acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
    if (valid) {
        const float4 pf = sd.pts[i];
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
            finish_point_pair<METHOD>(acc, m, S, rp, px, py, pz, gx, gy, gz, bd2, bx, by, bz, bidx);
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
                if (d2 < rp.th2) {
                    double C[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k) C[k] = m.vox_cov[(size_t)pr.vid * 9 + k];
                    add_pair<ELM_AVGICP>(acc, S.Rinv, S.tinv, px, py, pz, cx, cy, cz, C, nullptr, rp);
                }
            }
        }
        acc[29] = n_cand;
        acc[30] = n_occ;
        acc[31] = n_cand; // every candidate is distance-tested on this path
    }
    __shared__ double red[kBlock / 64][32];
    block_reduce_store(acc, red, partials + (size_t)L * kSums);
}

// ---- K1b: staged kernel -----------------------------------------------------------------------------------
// A workgroup owns 256 consecutive scan points; scans are stored along a Hilbert curve over 2 m sensor-frame cells,
// so those points occupy a few adjacent map voxels.  The workgroup
//   1. transforms its points and reduces the bounding box of their voxel keys (+1 voxel halo),
//   2. probes every cell of the box ONCE (4 cells per thread; ~10x fewer hash probes than 27 per point),
//   3. prefix-sums the bucket sizes and copies the buckets into LDS with coalesced 16-byte loads, in cell order
//      (x-major .. z-minor) and insertion order inside a bucket, i.e. exactly the reference's visiting order,
//   4. lets every thread scan its 3x3 columns as 9 contiguous LDS ranges (the three z-neighbours of a column are
//      adjacent in the staged order), all lanes of a voxel reading the same LDS address (broadcast),
//   5. reduces the packed normal equations.
// Workgroups whose box is too large (> kMaxCell cells or > kMaxStage points) take the direct path instead; the
// result is identical either way (same candidates, same order, same fp64 arithmetic).
constexpr int kMaxCell = 2048;  // box cells (incl. halo) a workgroup may cover
constexpr int kCellsPerThread = kMaxCell / kBlock;
constexpr int kMaxStage = 2944; // bucket points a workgroup may stage (34.5 KB as SoA floats; 3 workgroups per CU fit in 160 KB)
constexpr int kMaxList = 768;   // non-empty cells a workgroup may stage
constexpr double kFallbackUnit = 1099511627776.0; // 2^40: slot 31 carries tested candidates + 2^40 * fall-back workgroups
// bucket sizes are packed into 8 bits of the LDS cell table: maps with max_points_per_voxel > 255 use the direct kernel

// marks the 3x3x3 (or the 7 face-neighbour) cells of this thread's voxel in the workgroup's box
template <bool kSeven>
__device__ __forceinline__ void mark_needed(unsigned char* s_need, int bx, int by, int bz, int ny, int nz) {
    if (kSeven) {
        const int c = (bx * ny + by) * nz + bz;
        s_need[c] = 1; s_need[c + 1] = 1; s_need[c - 1] = 1;
        s_need[c + nz] = 1; s_need[c - nz] = 1;
        s_need[c + ny * nz] = 1; s_need[c - ny * nz] = 1;
    } else {
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int c0 = ((bx + dx) * ny + (by + dy)) * nz + (bz - 1);
                s_need[c0] = 1; s_need[c0 + 1] = 1; s_need[c0 + 2] = 1;
            }
    }
}

// first-slot loads of up to kCellsPerThread probes are issued back to back, then resolved (collisions walk on)
struct SlotLoad {
    int4 key;
    uint2 rg;
    unsigned h;
};
__device__ __forceinline__ SlotLoad slot_load(const DevMap& m, unsigned h) {
    SlotLoad r;
    r.h = h;
    r.key = *reinterpret_cast<const int4*>(&m.slots[h]);
    r.rg = *reinterpret_cast<const uint2*>(&m.slots[h].start);
    return r;
}
__device__ __forceinline__ Probe slot_resolve(const DevMap& m, SlotLoad sl, int kx, int ky, int kz) {
    Probe p;
    p.vid = -1; p.start = 0; p.cnt = 0;
    for (;;) {
        if (sl.key.w < 0) break;
        if (sl.key.x == kx && sl.key.y == ky && sl.key.z == kz) {
            p.vid = sl.key.w; p.start = sl.rg.x; p.cnt = sl.rg.y;
            break;
        }
        sl = slot_load(m, (sl.h + 1) & m.mask);
    }
    return p;
}

template <int METHOD>
__global__ __launch_bounds__(kBlock, 3) void k_accumulate(const DevMap m, const ScanDesc* __restrict__ scans, int batch,
                                                          unsigned total_blocks, const ScanState* __restrict__ st,
                                                          double* __restrict__ partials, const RegParams rp) {
    constexpr bool kPoints = (METHOD == ELM_P2P || METHOD == ELM_GICP);
    // LDS (<= 53 KB -> 3 workgroups per CU): bucket points as SoA floats (36 KB) or voxel means; cell tables; scratch
    __shared__ __attribute__((aligned(16))) float s_xyz[kPoints ? 3 * kMaxStage : 4];
    __shared__ unsigned s_cell[kPoints ? kMaxCell : 1];   // (LDS offset << 8) | count; offsets are the running prefix
    __shared__ uint2 s_list[kPoints ? kMaxList : 1];      // compacted non-empty cells: (cell, global start of its bucket), sorted by cell
    __shared__ double s_mean[kPoints ? 1 : kMaxList][3];  // means of the non-empty needed cells (compact slots)
    __shared__ int s_mvid[kPoints ? 1 : kMaxList];        // their voxel ids
    __shared__ short s_slot[kPoints ? 1 : kMaxCell];      // cell -> slot, -1 = no voxel
    __shared__ int s_nslot;
    __shared__ unsigned char s_need[kMaxCell];
    __shared__ double red[kBlock / 64][32];
    __shared__ int s_bb[6];
    __shared__ unsigned long long s_wsum[kBlock / 64];
    float* const s_x = s_xyz;
    float* const s_y = s_xyz + (kPoints ? kMaxStage : 0);
    float* const s_z = s_xyz + (kPoints ? 2 * kMaxStage : 0);
    uint2* const s_tmp = reinterpret_cast<uint2*>(s_xyz); // (global start, count) per cell while probing (16 KB, aliases s_x/s_y)

    const unsigned L = xcd_remap(blockIdx.x, total_blocks);
    const int s = find_scan(scans, batch, L, rp);
    const ScanState& S = st[s];
    if (S.done) return;
    const ScanDesc sd = scans[s];
    const unsigned tid = threadIdx.x;
    const unsigned i = (L - sd.blk_begin) * kBlock + tid;
    const bool valid = i < sd.n;
    const int lane = tid & 63, wave = tid >> 6;
    ELM_PHASE_BEGIN

    double acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;

    double px = 0, py = 0, pz = 0, gx = 0, gy = 0, gz = 0;
    int vx = 0, vy = 0, vz = 0;
    if (valid) {
        const float4 pf = sd.pts[i];
        px = pf.x; py = pf.y; pz = pf.z;
        gx = ((S.T[0] * px + S.T[4] * py) + S.T[8] * pz) + S.T[12];
        gy = ((S.T[1] * px + S.T[5] * py) + S.T[9] * pz) + S.T[13];
        gz = ((S.T[2] * px + S.T[6] * py) + S.T[10] * pz) + S.T[14];
        vx = floor_key(gx, m.voxel_size); vy = floor_key(gy, m.voxel_size); vz = floor_key(gz, m.voxel_size);
    }
    // ---- 1. bounding box of the voxel keys; clear the need map
    if (tid < 3) s_bb[tid] = INT_MAX;
    else if (tid < 6) s_bb[tid] = INT_MIN;
    for (int k = tid; k < kMaxCell / 4; k += kBlock) reinterpret_cast<unsigned*>(s_need)[k] = 0u;
    __syncthreads();
    {
        int mnx = valid ? vx : INT_MAX, mny = valid ? vy : INT_MAX, mnz = valid ? vz : INT_MAX;
        int mxx = valid ? vx : INT_MIN, mxy = valid ? vy : INT_MIN, mxz = valid ? vz : INT_MIN;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mnx = min(mnx, __shfl_xor(mnx, off, 64)); mny = min(mny, __shfl_xor(mny, off, 64)); mnz = min(mnz, __shfl_xor(mnz, off, 64));
            mxx = max(mxx, __shfl_xor(mxx, off, 64)); mxy = max(mxy, __shfl_xor(mxy, off, 64)); mxz = max(mxz, __shfl_xor(mxz, off, 64));
        }
        if (lane == 0) {
            atomicMin(&s_bb[0], mnx); atomicMin(&s_bb[1], mny); atomicMin(&s_bb[2], mnz);
            atomicMax(&s_bb[3], mxx); atomicMax(&s_bb[4], mxy); atomicMax(&s_bb[5], mxz);
        }
    }
    __syncthreads();
    const int lox = s_bb[0] - 1, loy = s_bb[1] - 1, loz = s_bb[2] - 1;
    const long long ex_ = (long long)s_bb[3] - s_bb[0] + 3, ey_ = (long long)s_bb[4] - s_bb[1] + 3, ez_ = (long long)s_bb[5] - s_bb[2] + 3;
    bool staged = (ex_ <= kMaxCell && ey_ <= kMaxCell && ez_ <= kMaxCell && ex_ * ey_ * ez_ <= kMaxCell);
    const int ny = (int)ey_, nz = (int)ez_;
    const int ncell = staged ? (int)(ex_ * ey_ * ez_) : 0;
    double n_cand = 0.0, n_occ = 0.0, n_tested = 0.0;
    // cell index -> (cx, cy, cz) without integer division: floor(c / d) == (c * M) >> 22 for c < 2048, d <= 2048
    const unsigned mg_nz = (1u << 22) / (unsigned)max(nz, 1) + 1u, mg_ny = (1u << 22) / (unsigned)max(ny, 1) + 1u;
    // ---- 2a. mark the cells some point of the workgroup will visit
    if (staged) {
        if (valid) mark_needed<METHOD == ELM_AVGICP>(s_need, vx - lox, vy - loy, vz - loz, ny, nz);
        __syncthreads();
    }
    ELM_PHASE(0)

    if (kPoints) {
        double bd2 = DBL_MAX;
        float bx = 0.f, by = 0.f, bz = 0.f;
        int bidx = -1;
        if (staged) {
            // ---- 2b. probe the needed cells: thread t takes cells t, t+256, ... (needed cells are clustered, the
            //          stride spreads them over the threads); all first-slot loads are in flight together
            {
                SlotLoad sl[kCellsPerThread];
                int kx[kCellsPerThread], ky[kCellsPerThread], kz[kCellsPerThread];
                bool need[kCellsPerThread];
#pragma unroll
                for (int k = 0; k < kCellsPerThread; ++k) {
                    const int c = (int)tid + k * kBlock;
                    need[k] = (c < ncell) && s_need[c];
                    const int cxy = (int)(((unsigned)c * mg_nz) >> 22), cz = c - cxy * nz;
                    const int cx = (int)(((unsigned)cxy * mg_ny) >> 22), cy = cxy - cx * ny;
                    kx[k] = lox + cx; ky[k] = loy + cy; kz[k] = loz + cz;
                    if (need[k]) sl[k] = slot_load(m, hash3(kx[k], ky[k], kz[k]) & m.mask);
                }
#pragma unroll
                for (int k = 0; k < kCellsPerThread; ++k) {
                    const int c = (int)tid + k * kBlock;
                    uint2 r = make_uint2(0u, 0u);
                    if (need[k]) {
                        const Probe pr = slot_resolve(m, sl[k], kx[k], ky[k], kz[k]);
                        if (pr.vid >= 0) r = make_uint2(pr.start, pr.cnt);
                    }
                    if (c < ncell) s_tmp[c] = r;
                }
            }
            __syncthreads();
            ELM_PHASE(1)
            // ---- 3a. block-wide exclusive prefix of (points, non-empty cells) in cell order
            unsigned cstart[kCellsPerThread], ccnt[kCellsPerThread];
            unsigned lsum = 0, lne = 0;
#pragma unroll
            for (int k = 0; k < kCellsPerThread; ++k) {
                const int c = (int)tid * kCellsPerThread + k;
                uint2 r = make_uint2(0u, 0u);
                if (c < ncell) r = s_tmp[c];
                cstart[k] = r.x; ccnt[k] = r.y;
                lsum += r.y;
                lne += (r.y > 0) ? 1u : 0u;
            }
            unsigned long long v = ((unsigned long long)lne << 32) | lsum, inc = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned long long t = __shfl_up(inc, off, 64);
                if (lane >= off) inc += t;
            }
            if (lane == 63) s_wsum[wave] = inc;
            __syncthreads(); // also: every s_tmp read is done before s_x/s_y are overwritten below
            unsigned long long wbase = 0, total = 0;
#pragma unroll
            for (int w = 0; w < kBlock / 64; ++w) {
                const unsigned long long t = s_wsum[w];
                if (w < wave) wbase += t;
                total += t;
            }
            const unsigned tot_pts = (unsigned)(total & 0xffffffffu), tot_ne = (unsigned)(total >> 32);
            if (tot_pts > (unsigned)kMaxStage || tot_ne > (unsigned)kMaxList) staged = false; // uniform
            if (staged) {
                const unsigned long long excl = wbase + inc - v;
                unsigned off = (unsigned)(excl & 0xffffffffu), lpos = (unsigned)(excl >> 32);
#pragma unroll
                for (int k = 0; k < kCellsPerThread; ++k) {
                    const int c = (int)tid * kCellsPerThread + k;
                    if (c < ncell) {
                        s_cell[c] = (off << 8) | ccnt[k];
                        if (ccnt[k] > 0) s_list[lpos++] = make_uint2((unsigned)c, cstart[k]);
                        off += ccnt[k];
                    }
                }
                __syncthreads();
                ELM_PHASE(2)
                // ---- 3b. copy the buckets: 16 lanes per non-empty cell (2 points per lane), 4 cells in flight per group
                const unsigned grp = tid >> 4, l = tid & 15;
                for (unsigned e0 = grp; e0 < tot_ne; e0 += 64) {
                    float4 qa[4], qb[4];
                    unsigned da[4];
                    bool oka[4], okb[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const unsigned e = e0 + 16u * u;
                        oka[u] = false; okb[u] = false; da[u] = 0;
                        if (e < tot_ne) {
                            const uint2 le = s_list[e];
                            const unsigned ci = s_cell[le.x];
                            const unsigned cnt = ci & 0xffu, o = ci >> 8;
                            da[u] = o + l;
                            if (l < cnt) { qa[u] = m.pts[le.y + l]; oka[u] = true; }
                            if (l + 16 < cnt) { qb[u] = m.pts[le.y + l + 16]; okb[u] = true; }
                            for (unsigned j = l + 32; j < cnt; j += 16) { // buckets larger than 32 points
                                const float4 qq = m.pts[le.y + j];
                                s_x[o + j] = qq.x; s_y[o + j] = qq.y; s_z[o + j] = qq.z;
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (oka[u]) { s_x[da[u]] = qa[u].x; s_y[da[u]] = qa[u].y; s_z[da[u]] = qa[u].z; }
                        if (okb[u]) { s_x[da[u] + 16] = qb[u].x; s_y[da[u] + 16] = qb[u].y; s_z[da[u] + 16] = qb[u].z; }
                    }
                }
                __syncthreads();
                ELM_PHASE(3)
                // ---- 4. nearest bucket point of the thread's 27 cells, staged in the reference's visiting order.
                //      The thread's own cell is scanned first; every other cell is skipped when even the closest
                //      point its key range could hold is STRICTLY farther than the best so far (it can neither win
                //      nor tie), and ties are resolved towards the smaller staged index, i.e. the candidate the
                //      reference's x-major / insertion-order walk meets first.  Same winner as the full walk.
                int bj = -1, bcell = 0;
                if (valid) {
                    const int cown = ((vx - lox) * ny + (vy - loy)) * nz + (vz - loz);
                    auto scan_cell = [&](int c, unsigned info) {
                        const int beg = (int)(info >> 8), end = beg + (int)(info & 0xffu);
                        for (int j = beg; j < end; j += 4) {
                            const int j1 = min(j + 1, end - 1), j2 = min(j + 2, end - 1), j3 = min(j + 3, end - 1);
                            const float x0 = s_x[j], y0 = s_y[j], z0 = s_z[j];
                            const float x1 = s_x[j1], y1 = s_y[j1], z1 = s_z[j1];
                            const float x2 = s_x[j2], y2 = s_y[j2], z2 = s_z[j2];
                            const float x3 = s_x[j3], y3 = s_y[j3], z3 = s_z[j3];
                            const double e0x = (double)x0 - gx, e0y = (double)y0 - gy, e0z = (double)z0 - gz;
                            const double e1x = (double)x1 - gx, e1y = (double)y1 - gy, e1z = (double)z1 - gz;
                            const double e2x = (double)x2 - gx, e2y = (double)y2 - gy, e2z = (double)z2 - gz;
                            const double e3x = (double)x3 - gx, e3y = (double)y3 - gy, e3z = (double)z3 - gz;
                            const double d0 = (e0x * e0x + e0y * e0y) + e0z * e0z;
                            const double d1 = (e1x * e1x + e1y * e1y) + e1z * e1z;
                            const double d2 = (e2x * e2x + e2y * e2y) + e2z * e2z;
                            const double d3 = (e3x * e3x + e3y * e3y) + e3z * e3z;
                            if (d0 < bd2 || (d0 == bd2 && j < bj)) { bd2 = d0; bj = j; bcell = c; }
                            if (d1 < bd2 || (d1 == bd2 && j1 < bj)) { bd2 = d1; bj = j1; bcell = c; }
                            if (d2 < bd2 || (d2 == bd2 && j2 < bj)) { bd2 = d2; bj = j2; bcell = c; }
                            if (d3 < bd2 || (d3 == bd2 && j3 < bj)) { bd2 = d3; bj = j3; bcell = c; }
                        }
                    };
                    const unsigned iown = s_cell[cown];
                    scan_cell(cown, iown);
                    n_tested += (double)(iown & 0xffu);
                    const double vs = m.voxel_size, slack = 1e-9 * vs;
                    for (int dx = -1; dx <= 1; ++dx) {
                        // stored keys truncate toward zero (vhm.cpp:275): key k > 0 holds [k, k+1) vs, k < 0 holds (k-1, k] vs,
                        // k == 0 holds (-1, 1) vs
                        const int kx = vx + dx;
                        const double lx = (double)(kx <= 0 ? kx - 1 : kx) * vs - slack, hx = (double)(kx >= 0 ? kx + 1 : kx) * vs + slack;
                        const double ax = fmax(fmax(lx - gx, gx - hx), 0.0);
                        for (int dy = -1; dy <= 1; ++dy) {
                            const int ky = vy + dy;
                            const double ly = (double)(ky <= 0 ? ky - 1 : ky) * vs - slack, hy = (double)(ky >= 0 ? ky + 1 : ky) * vs + slack;
                            const double ay = fmax(fmax(ly - gy, gy - hy), 0.0);
                            const double axy = ax * ax + ay * ay;
                            const int c0 = ((kx - lox) * ny + (ky - loy)) * nz + (vz - 1 - loz);
#pragma unroll
                            for (int dz = 0; dz < 3; ++dz) {
                                const int c = c0 + dz;
                                const unsigned info = s_cell[c];
                                const unsigned cnt = info & 0xffu;
                                n_occ += (cnt > 0) ? 1.0 : 0.0;
                                n_cand += (double)cnt;
                                if (cnt == 0 || c == cown) continue;
                                const int kz = vz - 1 + dz;
                                const double lz = (double)(kz <= 0 ? kz - 1 : kz) * vs - slack, hz = (double)(kz >= 0 ? kz + 1 : kz) * vs + slack;
                                const double az = fmax(fmax(lz - gz, gz - hz), 0.0);
                                const double lb = (axy + az * az) * (1.0 - 1e-12);
                                if (lb > bd2) continue; // every point of the cell is strictly farther than the current best
                                n_tested += (double)cnt;
                                scan_cell(c, info);
                            }
                        }
                    }
                    if (bj >= 0) {
                        bx = s_x[bj]; by = s_y[bj]; bz = s_z[bj];
                        bidx = 0;
                        if (METHOD == ELM_GICP) { // global index of the winner (s_list is sorted by cell)
                            int lo_ = 0, hi_ = (int)tot_ne - 1;
                            while (lo_ < hi_) {
                                const int mid = (lo_ + hi_) >> 1;
                                if ((int)s_list[mid].x < bcell) lo_ = mid + 1; else hi_ = mid;
                            }
                            bidx = (int)(s_list[lo_].y + ((unsigned)bj - (s_cell[bcell] >> 8)));
                        }
                    }
                }
                ELM_PHASE(4)
            }
        }
        if (!staged) {
            if (valid) { nearest_point_direct(m, vx, vy, vz, gx, gy, gz, bd2, bx, by, bz, bidx, n_cand, n_occ); n_tested = n_cand; }
            if (tid == 0) acc[31] = kFallbackUnit; // fall-back workgroups are counted (high part of slot 31)
            ELM_PHASE(7)
        }
        if (valid) finish_point_pair<METHOD>(acc, m, S, rp, px, py, pz, gx, gy, gz, bd2, bx, by, bz, bidx);
        ELM_PHASE(5)
    } else {
        // VGICP / AVGICP: stage the voxel MEANS of the needed cells (GetCorrespondencesCov / AllCov, vhm.cpp:90-206)
        if (staged) {
            if (tid == 0) s_nslot = 0;
            __syncthreads();
            {
                SlotLoad sl[kCellsPerThread];
                int kx[kCellsPerThread], ky[kCellsPerThread], kz[kCellsPerThread];
                bool need[kCellsPerThread];
#pragma unroll
                for (int k = 0; k < kCellsPerThread; ++k) {
                    const int c = (int)tid + k * kBlock;
                    need[k] = (c < ncell) && s_need[c];
                    const int cxy = (int)(((unsigned)c * mg_nz) >> 22), cz = c - cxy * nz;
                    const int cx = (int)(((unsigned)cxy * mg_ny) >> 22), cy = cxy - cx * ny;
                    kx[k] = lox + cx; ky[k] = loy + cy; kz[k] = loz + cz;
                    if (need[k]) sl[k] = slot_load(m, hash3(kx[k], ky[k], kz[k]) & m.mask);
                }
#pragma unroll
                for (int k = 0; k < kCellsPerThread; ++k) {
                    const int c = (int)tid + k * kBlock;
                    int slot = -1;
                    if (need[k]) {
                        const Probe pr = slot_resolve(m, sl[k], kx[k], ky[k], kz[k]);
                        if (pr.vid >= 0 && pr.cnt > 0) {
                            slot = atomicAdd(&s_nslot, 1);
                            if (slot < kMaxList) {
                                s_mean[slot][0] = m.vox_mean[(size_t)pr.vid * 3];
                                s_mean[slot][1] = m.vox_mean[(size_t)pr.vid * 3 + 1];
                                s_mean[slot][2] = m.vox_mean[(size_t)pr.vid * 3 + 2];
                                s_mvid[slot] = pr.vid;
                            }
                        }
                    }
                    if (c < ncell) s_slot[c] = (short)slot;
                }
            }
            __syncthreads();
            if (s_nslot > kMaxList) staged = false; // uniform
        }
        if (!staged && tid == 0) acc[31] = kFallbackUnit;
        ELM_PHASE(1)
        if (valid) {
            if (METHOD == ELM_VGICP) {
                double bd2 = DBL_MAX, bmx = 0.0, bmy = 0.0, bmz = 0.0;
                int bvid = -1;
                if (staged) {
                    for (int dx = -1; dx <= 1; ++dx)
                        for (int dy = -1; dy <= 1; ++dy) {
                            const int c0 = ((vx + dx - lox) * ny + (vy + dy - loy)) * nz + (vz - 1 - loz);
#pragma unroll
                            for (int dz = 0; dz < 3; ++dz) {
                                const int sl = s_slot[c0 + dz];
                                if (sl < 0) continue;
                                const int vid = s_mvid[sl];
                                n_occ += 1.0;
                                n_cand += 1.0;
                                const double cx = s_mean[sl][0], cy = s_mean[sl][1], cz = s_mean[sl][2];
                                const double ex = cx - gx, ey = cy - gy, ez = cz - gz;
                                const double d2 = (ex * ex + ey * ey) + ez * ez;
                                if (d2 < bd2) { bd2 = d2; bvid = vid; bmx = cx; bmy = cy; bmz = cz; }
                            }
                        }
                } else {
                    nearest_voxel_direct(m, vx, vy, vz, gx, gy, gz, bd2, bvid, bmx, bmy, bmz, n_cand, n_occ);
                }
                ELM_PHASE(4)
                finish_voxel_pair(acc, m, S, rp, px, py, pz, gx, gy, gz, bd2, bvid, bmx, bmy, bmz);
            } else {
                const int ox[7] = {0, 1, -1, 0, 0, 0, 0}, oy[7] = {0, 0, 0, 1, -1, 0, 0}, oz[7] = {0, 0, 0, 0, 0, 1, -1};
#pragma unroll
                for (int k7 = 0; k7 < 7; ++k7) {
                    int vid;
                    double cx, cy, cz;
                    if (staged) {
                        const int c = ((vx + ox[k7] - lox) * ny + (vy + oy[k7] - loy)) * nz + (vz + oz[k7] - loz);
                        const int sl = s_slot[c];
                        if (sl < 0) continue;
                        vid = s_mvid[sl];
                        cx = s_mean[sl][0]; cy = s_mean[sl][1]; cz = s_mean[sl][2];
                    } else {
                        const Probe pr = probe_voxel(m, vx + ox[k7], vy + oy[k7], vz + oz[k7]);
                        if (pr.vid < 0 || pr.cnt == 0) continue;
                        vid = pr.vid;
                        cx = m.vox_mean[(size_t)vid * 3]; cy = m.vox_mean[(size_t)vid * 3 + 1]; cz = m.vox_mean[(size_t)vid * 3 + 2];
                    }
                    n_occ += 1.0;
                    n_cand += 1.0;
                    const double ex = cx - gx, ey = cy - gy, ez = cz - gz;
                    const double d2 = (ex * ex + ey * ey) + ez * ez;
                    if (d2 < rp.th2) {
                        double C[9];
#pragma unroll
                        for (int k = 0; k < 9; ++k) C[k] = m.vox_cov[(size_t)vid * 9 + k];
                        add_pair<ELM_AVGICP>(acc, S.Rinv, S.tinv, px, py, pz, cx, cy, cz, C, nullptr, rp);
                    }
                }
            }
        }
        ELM_PHASE(5)
    }
    if (valid) {
        acc[29] = n_cand;
        acc[30] = n_occ;
        acc[31] += n_tested; // low part of slot 31: candidates actually distance-tested after pruning
    }
    // ---- 5. reduce
    if (kPoints) {
        __syncthreads(); // every thread is done with the staged points: reuse their LDS
        block_reduce_store_lds(acc, reinterpret_cast<double*>(s_xyz), partials + (size_t)L * kSums);
    } else {
        block_reduce_store(acc, red, partials + (size_t)L * kSums);
    }
    ELM_PHASE(6)
}

// ---- K1c: neighbourhood-list kernel -----------------------------------------------------------------------
// One thread per scan point: ONE hash probe (floor key of the transformed point) finds the precomputed candidate
// list of that query voxel -- the 27 neighbour buckets already concatenated in the reference's visiting order --
// and the thread streams it with independent 16-byte loads (8 in flight).  Lanes of the same voxel read the same
// addresses.  No workgroup cooperation, no barriers before the final reduction.
template <int METHOD>
__global__ __launch_bounds__(kBlock) void k_accumulate_nbr(const DevMap m, const ScanDesc* __restrict__ scans, int batch,
                                                           unsigned total_blocks, const ScanState* __restrict__ st,
                                                           double* __restrict__ partials, const RegParams rp) {
    __shared__ double s_buf[16 * kBlock]; // 32 KB for the transpose reduction
    const unsigned L = xcd_remap(blockIdx.x, total_blocks);
    const int s = find_scan(scans, batch, L, rp);
    const ScanState& S = st[s];
    if (S.done) return;
    const ScanDesc sd = scans[s];
    const unsigned i = (L - sd.blk_begin) * kBlock + threadIdx.x;
    const bool valid = i < sd.n;
    double acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
    if (valid) {
        const float4 pf = sd.pts[i];
        const double px = pf.x, py = pf.y, pz = pf.z;
        const double gx = ((S.T[0] * px + S.T[4] * py) + S.T[8] * pz) + S.T[12];
        const double gy = ((S.T[1] * px + S.T[5] * py) + S.T[9] * pz) + S.T[13];
        const double gz = ((S.T[2] * px + S.T[6] * py) + S.T[10] * pz) + S.T[14];
        const int vx = floor_key(gx, m.voxel_size), vy = floor_key(gy, m.voxel_size), vz = floor_key(gz, m.voxel_size);
        // query-voxel probe
        unsigned start = 0, cnt = 0, nocc = 0;
        double n_exact = 0.0;
        {
            unsigned h = hash3(vx, vy, vz) & m.qmask;
            for (;;) {
                const int4 key = *reinterpret_cast<const int4*>(&m.qslots[h]);
                const uint4 rg = *reinterpret_cast<const uint4*>(&m.qslots[h].start);
                if (key.w < 0) break;
                if (key.x == vx && key.y == vy && key.z == vz) { start = rg.x; cnt = rg.y; nocc = rg.z; break; }
                h = (h + 1) & m.qmask;
            }
        }
        const Pt3* __restrict__ lp = m.nbr_pts + start;
        double bd2 = DBL_MAX;
        int bj = -1;
        const int n = (int)cnt;
        // Pass 1, float32: best and second-best squared distance.  g is split into gh + gl (float32 each, gh + gl == g
        // to ~2^-48), so (q - gh) - gl reproduces q - g to a few float32 ulps of |q - g| and the float32 distance is
        // within ~5e-7 relative (+ a tiny absolute term) of the reference's float64 one.  When the runner-up is
        // farther than that margin the float32 winner IS the float64 winner (it cannot even tie); otherwise -- exact
        // or near ties, i.e. practically never -- the lane falls back to the full float64 walk below.
        const float ghx = (float)gx, ghy = (float)gy, ghz = (float)gz;
        const float glx = (float)(gx - (double)ghx), gly = (float)(gy - (double)ghy), glz = (float)(gz - (double)ghz);
        float m1 = __builtin_inff(), m2 = __builtin_inff();
        int j1 = -1;
        int j = 0;
        for (; j + 8 <= n; j += 8) {
            Pt3 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = lp[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float ex = (q[u].x - ghx) - glx, ey = (q[u].y - ghy) - gly, ez = (q[u].z - ghz) - glz;
                const float d = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
                m2 = fminf(m2, fmaxf(d, m1));
                const bool c = d < m1;
                m1 = c ? d : m1;
                j1 = c ? (j + u) : j1;
            }
        }
        for (; j < n; ++j) {
            const Pt3 q = lp[j];
            const float ex = (q.x - ghx) - glx, ey = (q.y - ghy) - gly, ez = (q.z - ghz) - glz;
            const float d = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
            m2 = fminf(m2, fmaxf(d, m1));
            const bool c = d < m1;
            m1 = c ? d : m1;
            j1 = c ? j : j1;
        }
        if (j1 >= 0) {
            const float slack = 4e-11f * (fabsf(ghx) + fabsf(ghy) + fabsf(ghz) + 1.0f);
            const bool clear_winner = m2 > m1 + m1 * 1.9073486328125e-06f + slack; // 2^-19
            if (clear_winner) {
                const Pt3 q = lp[j1];
                const double ex = (double)q.x - gx, ey = (double)q.y - gy, ez = (double)q.z - gz;
                bd2 = (ex * ex + ey * ey) + ez * ez; // the reference's float64 value for the range test
                bj = j1;
            } else {
                // exact float64 walk; the list is cell-sorted, so equal distances are resolved explicitly towards the
                // candidate the reference meets first (bucket visiting rank, then insertion order = global index)
                unsigned brank = 0xFFFFFFFFu, bgi = 0xFFFFFFFFu;
                for (int k = 0; k < n; ++k) {
                    const Pt3 q = lp[k];
                    const double ex = (double)q.x - gx, ey = (double)q.y - gy, ez = (double)q.z - gz;
                    const double d2 = (ex * ex + ey * ey) + ez * ez;
                    if (d2 > bd2) continue;
                    const int kx = (int)((double)q.x / m.voxel_size), ky = (int)((double)q.y / m.voxel_size), kz = (int)((double)q.z / m.voxel_size);
                    const unsigned rank = (unsigned)(((kx - vx + 1) * 3 + (ky - vy + 1)) * 3 + (kz - vz + 1));
                    const unsigned gi = m.nbr_idx[(size_t)start + k];
                    if (d2 < bd2 || rank < brank || (rank == brank && gi < bgi)) { bd2 = d2; bj = k; brank = rank; bgi = gi; }
                }
                n_exact = 1.0;
            }
        }
        float bx = 0.f, by = 0.f, bz = 0.f;
        int bidx = -1;
        if (bj >= 0) {
            const Pt3 q = lp[bj];
            bx = q.x; by = q.y; bz = q.z;
            bidx = (METHOD == ELM_GICP) ? (int)m.nbr_idx[(size_t)start + bj] : 0; // payload lookup only for GICP
        }
        finish_point_pair<METHOD>(acc, m, S, rp, px, py, pz, gx, gy, gz, bd2, bx, by, bz, bidx);
        acc[29] = (double)cnt;
        acc[30] = (double)nocc;
        acc[31] = (double)cnt + n_exact * kFallbackUnit; // high part: points that needed the exact float64 walk
    }
    block_reduce_store_lds(acc, s_buf, partials + (size_t)L * kSums);
}

// ---- K1d: cell-indexed neighbourhood lists --------------------------------------------------------------
// The candidate list of a query voxel is kept sorted by half-voxel cells: a 6x6x6 grid with origin (v - 1) * voxel_size
// (per axis), edge h = voxel_size / 2, indices clamped to 0..5 so the outermost cells are unbounded outward (truncated
// keys make the bucket of voxel key 0 two voxels wide).  Cells are ordered (ix, iy, iz)-major, so the cells iz0..iz1 of
// one (ix, iy) column are one contiguous range; a 16-byte record per column holds its seven cell boundaries.
//   stage 1, per lane: probe the query voxel (as K1c), then scan the 2x2x2 block of cells that g leans into (own cell +
//     the neighbours on the nearer side of every axis; 4 column ranges, ~20 candidates) in float32 with the best /
//     runner-up logic of K1c.  Every point of space within rho = distance(g, open faces of that block) >= h/2 lies in the
//     block, so when the float32 winner is a clear one AND its distance (+ error margin) is below rho, no candidate outside
//     the block can win or tie: the lane is done after three round trips (probe, 4 records, candidates).
//   stage 2, per wave: the few lanes that could not decide (pose still far off, isolated points, near ties) are served one
//     after the other by the whole wave: 64 lanes share the lane's complete list, compute the reference's float64
//     distances and reduce (distance, visiting rank, global index) lexicographically -- exactly the reference's first
//     strict minimum in its visiting order (vhm.cpp:208-243 + insertion order), whatever the order of the list.
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

template <int METHOD>
__global__ __launch_bounds__(kBlock) void k_accumulate_cell(const DevMap m, const ScanDesc* __restrict__ scans, int batch,
                                                            unsigned total_blocks, const ScanState* __restrict__ st,
                                                            double* __restrict__ partials, const RegParams rp) {
    __shared__ double s_buf[16 * kBlock];
    ELM_PHASE_BEGIN
    const unsigned L = xcd_remap(blockIdx.x, total_blocks);
    const int s = find_scan(scans, batch, L, rp);
    const ScanState& S = st[s];
    if (S.done) return;
    const ScanDesc sd = scans[s];
    const unsigned i = (L - sd.blk_begin) * kBlock + threadIdx.x;
    const bool valid = i < sd.n;
    double acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
    double px = 0.0, py = 0.0, pz = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
    int vx = 0, vy = 0, vz = 0;
    unsigned start = 0, cnt = 0, nocc = 0;
    const Pt3* __restrict__ lp = m.nbr_pts;
    double bd2 = DBL_MAX;
    int bj = -1;
    int n_tested = 0;
    bool hard = false;
    if (valid) {
        const float4 pf = sd.pts[i];
        px = pf.x; py = pf.y; pz = pf.z;
        gx = ((S.T[0] * px + S.T[4] * py) + S.T[8] * pz) + S.T[12];
        gy = ((S.T[1] * px + S.T[5] * py) + S.T[9] * pz) + S.T[13];
        gz = ((S.T[2] * px + S.T[6] * py) + S.T[10] * pz) + S.T[14];
        vx = floor_key(gx, m); vy = floor_key(gy, m); vz = floor_key(gz, m);
        int qid = -1;
        ELM_PHASE(8)
        {
            // linear probing, two slots per round trip (the table is kept at load <= 0.5: a third slot is rarely needed)
            unsigned h = hash3(vx, vy, vz) & m.qmask;
            for (;;) {
                const unsigned h2 = (h + 1) & m.qmask;
                const int4 key = *reinterpret_cast<const int4*>(&m.qslots[h]);
                const uint4 rg = *reinterpret_cast<const uint4*>(&m.qslots[h].start);
                const int4 key2 = *reinterpret_cast<const int4*>(&m.qslots[h2]);
                const uint4 rg2 = *reinterpret_cast<const uint4*>(&m.qslots[h2].start);
                if (key.w < 0) break;
                if (key.x == vx && key.y == vy && key.z == vz) { start = rg.x; cnt = rg.y; nocc = rg.z; qid = key.w; break; }
                if (key2.w < 0) break;
                if (key2.x == vx && key2.y == vy && key2.z == vz) { start = rg2.x; cnt = rg2.y; nocc = rg2.z; qid = key2.w; break; }
                h = (h + 2) & m.qmask;
            }
        }
        ELM_PHASE(9)
        lp = m.nbr_pts + start;
#ifdef ELM_SKIP_STAGE1
        if (false) {
#else
        if (cnt) {
#endif
            const uint16_t* __restrict__ co = m.nbr_cell_off + (size_t)qid * kCellStride;
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
#ifdef ELM_SKIP_CANDS
            const int nblk = cb[4] > 100000 ? 1 : 0; // ablation: records only
#else
            const int nblk = cb[4];
#endif
            n_tested = ((se[0] - sb[0]) + (se[1] - sb[1])) + ((se[2] - sb[2]) + (se[3] - sb[3]));
            const float ghx = (float)gx, ghy = (float)gy, ghz = (float)gz;
            const float glx = (float)(gx - (double)ghx), gly = (float)(gy - (double)ghy), glz = (float)(gz - (double)ghz);
            float m1 = __builtin_inff(), m2 = __builtin_inff();
            int j1 = -1;
            for (int t0 = 0; t0 < nblk; t0 += 2) { // two blocks = 8 independent loads per round trip
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
                    for (int v = 0; v < 3; ++v) {
                        const Vec4u r = bp[v];
                        qf[w][4 * v] = __uint_as_float(r.x); qf[w][4 * v + 1] = __uint_as_float(r.y);
                        qf[w][4 * v + 2] = __uint_as_float(r.z); qf[w][4 * v + 3] = __uint_as_float(r.w);
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
            // float32 distances are within 2^-20 relative (+ slack / 2) of the reference's float64 ones (see K1c)
            hard = true;
            if (j1 >= 0) {
                const float slack = 4e-11f * (fabsf(ghx) + fabsf(ghy) + fabsf(ghz) + 1.0f);
                const float r2 = m1 + m1 * 1.9073486328125e-06f + slack; // 2^-19
                // sqrt(r2) * 1.000001 + 1e-6 < rho, without the square root
                const double rr = (rho - 1e-6) * 0.999999;
                if (m2 > r2 && rr > 0.0 && (double)r2 < rr * rr) {
                    const Pt3 q = lp[j1];
                    const double ex = (double)q.x - gx, ey = (double)q.y - gy, ez = (double)q.z - gz;
                    bd2 = (ex * ex + ey * ey) + ez * ez; // the reference's float64 value for the range test
                    bj = j1;
                    hard = false;
                }
            }
        }
        ELM_PHASE(10)
    }
    // stage 2: the wave serves its undecided lanes one by one
    double n_exact = 0.0;
    {
#ifdef ELM_SKIP_HARD
        unsigned long long todo = 0;
#else
        unsigned long long todo = __ballot(hard);
#endif
        const int lane = (int)(threadIdx.x & 63);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const double hgx = lane_bcast(gx, src), hgy = lane_bcast(gy, src), hgz = lane_bcast(gz, src);
            const unsigned hstart = lane_bcast(start, src), hcnt = lane_bcast(cnt, src);
            const Pt3* __restrict__ hp = m.nbr_pts + hstart;
            // pass 1: the float64 minimum (the reference's arithmetic).  Equal distances are almost never seen; when one
            // is, pass 2 below settles it by the reference's visiting order.
            double d_best = DBL_MAX;
            unsigned k_best = 0xFFFFFFFFu;
            bool tie = false;
            for (unsigned base = 0; base < hcnt; base += 256) { // four independent coalesced loads per lane and round trip
                Pt3 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = hp[min(base + (unsigned)lane + 64u * u, hcnt - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned k = base + (unsigned)lane + 64u * u;
                    const double ex = (double)q[u].x - hgx, ey = (double)q[u].y - hgy, ez = (double)q[u].z - hgz;
                    const double dd = (ex * ex + ey * ey) + ez * ez;
                    const double d2 = (k < hcnt) ? dd : DBL_MAX;
                    tie = (d2 == d_best && k < hcnt) ? true : ((d2 < d_best) ? false : tie);
                    k_best = (d2 < d_best) ? k : k_best;
                    d_best = fmin(d2, d_best);
                }
            }
            const double d_min = wave_min(d_best);
            const unsigned long long at_min = __ballot(d_best == d_min);
            const bool contested = (__popcll(at_min) != 1) || (__ballot(tie && d_best == d_min) != 0ull);
            if (!contested) {
                k_best = lane_bcast(k_best, __ffsll((long long)at_min) - 1);
            } else {
                const int hvx = lane_bcast(vx, src), hvy = lane_bcast(vy, src), hvz = lane_bcast(vz, src);
                unsigned r_best = 0xFFFFFFFFu, g_best = 0xFFFFFFFFu;
                k_best = 0xFFFFFFFFu;
                for (unsigned k = (unsigned)lane; k < hcnt; k += 64) {
                    const Pt3 q = hp[k];
                    const double ex = (double)q.x - hgx, ey = (double)q.y - hgy, ez = (double)q.z - hgz;
                    const double d2 = (ex * ex + ey * ey) + ez * ez;
                    if (d2 != d_min) continue;
                    const unsigned gi = m.nbr_idx[(size_t)hstart + k];
                    const int kx = (int)((double)q.x / m.voxel_size), ky = (int)((double)q.y / m.voxel_size), kz = (int)((double)q.z / m.voxel_size);
                    const unsigned rank = (unsigned)(((kx - hvx + 1) * 3 + (ky - hvy + 1)) * 3 + (kz - hvz + 1));
                    if (rank < r_best || (rank == r_best && gi < g_best)) { r_best = rank; g_best = gi; k_best = k; }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const unsigned orank = (unsigned)__shfl_xor((int)r_best, off, 64), og = (unsigned)__shfl_xor((int)g_best, off, 64),
                                   ok = (unsigned)__shfl_xor((int)k_best, off, 64);
                    if (orank < r_best || (orank == r_best && og < g_best)) { r_best = orank; g_best = og; k_best = ok; }
                }
            }
            d_best = d_min;
            if (lane == src) { bd2 = d_best; bj = (int)k_best; n_exact = 1.0; n_tested += (int)hcnt; }
        }
    }
    ELM_PHASE(12)
    if (valid) {
        float bx = 0.f, by = 0.f, bz = 0.f;
        int bidx = -1;
        if (bj >= 0) {
            const Pt3 q = lp[bj];
            bx = q.x; by = q.y; bz = q.z;
            bidx = (METHOD == ELM_GICP) ? (int)m.nbr_idx[(size_t)start + bj] : 0;
        }
#ifndef ELM_SKIP_PAIR
        finish_point_pair<METHOD>(acc, m, S, rp, px, py, pz, gx, gy, gz, bd2, bx, by, bz, bidx);
#else
        acc[0] = bd2 + bx + by + bz + bidx;
#endif
        acc[29] = (double)cnt;  // candidates of the reference's walk
        acc[30] = (double)nocc;
        acc[31] = (double)n_tested + n_exact * kFallbackUnit;
        ELM_PHASE(13)
    }
#ifdef ELM_SKIP_REDUCE
    if (acc[0] + acc[29] + acc[31] == -1.0) partials[(size_t)L * kSums + threadIdx.x] = acc[1] + s_buf[threadIdx.x];
#else
    block_reduce_store_lds(acc, s_buf, partials + (size_t)L * kSums);
#endif
    ELM_PHASE(14)
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

// ---- K1e: voxel-mean lists (VGICP) ------------------------------------------------------------------------
// GetCorrespondencesCov (vhm.cpp:90-151) visits the 27 neighbour voxels of the point's floor-keyed voxel and keeps the
// nearest voxel MEAN (strict <, first met wins).  Here the occupied ones (~10 of 27) are precomputed per query voxel in
// that visiting order as 32-byte (mean, id) records: one probe, then <= 27 contiguous records, float64 distances in the
// reference's order -- no staging, no barriers before the block reduction.
template <int METHOD>
__global__ __launch_bounds__(kBlock) void k_accumulate_vnbr(const DevMap m, const ScanDesc* __restrict__ scans, int batch,
                                                            unsigned total_blocks, const ScanState* __restrict__ st,
                                                            double* __restrict__ partials, const RegParams rp) {
    __shared__ double s_buf[16 * kBlock];
    const unsigned L = xcd_remap(blockIdx.x, total_blocks);
    const int s = find_scan(scans, batch, L, rp);
    const ScanState& S = st[s];
    if (S.done) return;
    const ScanDesc sd = scans[s];
    const unsigned i = (L - sd.blk_begin) * kBlock + threadIdx.x;
    const bool valid = i < sd.n;
    double acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
    if (valid) {
        const float4 pf = sd.pts[i];
        const double px = pf.x, py = pf.y, pz = pf.z;
        const double gx = ((S.T[0] * px + S.T[4] * py) + S.T[8] * pz) + S.T[12];
        const double gy = ((S.T[1] * px + S.T[5] * py) + S.T[9] * pz) + S.T[13];
        const double gz = ((S.T[2] * px + S.T[6] * py) + S.T[10] * pz) + S.T[14];
        const int vx = floor_key(gx, m), vy = floor_key(gy, m), vz = floor_key(gz, m);
        unsigned start = 0, cnt = 0;
        {
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
        const VoxRec* __restrict__ lp = m.vnbr + start;
        if (METHOD == ELM_VGICP) {
            double bd2 = DBL_MAX, bmx = 0.0, bmy = 0.0, bmz = 0.0;
            int bvid = -1;
            for (unsigned j = 0; j < cnt; j += 4) { // four records (eight 16-byte loads) per round trip
                VoxRec r[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) r[u] = lp[min(j + u, cnt - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double ex = r[u].mx - gx, ey = r[u].my - gy, ez = r[u].mz - gz;
                    const double d2 = (ex * ex + ey * ey) + ez * ez;
                    if (j + u < cnt && d2 < bd2) { bd2 = d2; bvid = r[u].vid; bmx = r[u].mx; bmy = r[u].my; bmz = r[u].mz; }
                }
            }
            finish_voxel_pair(acc, m, S, rp, px, py, pz, gx, gy, gz, bd2, bvid, bmx, bmy, bmz);
            acc[29] = (double)cnt;
            acc[30] = (double)cnt;
            acc[31] = (double)cnt;
        } else {
            // AVGICP, GetCorrespondencesAllCov (vhm.cpp:153-206): every existing FACE neighbour (and the voxel itself) whose
            // mean is within range is a pair of its own.  The records carry the neighbour's position code (dx+1)*9+(dy+1)*3+
            // (dz+1); the seven wanted ones are met in list order (-x, -y, -z, 0, +z, +y, +x) instead of the reference's
            // (0, +x, -x, +y, -y, +z, -z): the same pairs, added in another order.
            double n_pairs = 0.0;
            for (unsigned j = 0; j < cnt; ++j) {
                const VoxRec r = lp[j];
                const int code = r.pad;
                if (!(code == 13 || code == 22 || code == 4 || code == 16 || code == 10 || code == 14 || code == 12)) continue;
                n_pairs += 1.0;
                const double ex = r.mx - gx, ey = r.my - gy, ez = r.mz - gz;
                const double d2 = (ex * ex + ey * ey) + ez * ez;
                if (d2 < rp.th2) {
                    double C[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k) C[k] = m.vox_cov[(size_t)r.vid * 9 + k];
                    add_pair<ELM_AVGICP>(acc, S.Rinv, S.tinv, px, py, pz, r.mx, r.my, r.mz, C, nullptr, rp);
                }
            }
            acc[29] = n_pairs;
            acc[30] = n_pairs;
            acc[31] = n_pairs;
        }
    }
    block_reduce_store_lds(acc, s_buf, partials + (size_t)L * kSums);
}

__global__ __launch_bounds__(256) void k_vnbr_fill(const DevMap m, const int32_t* __restrict__ qkeys, unsigned n_q,
                                                   const unsigned* __restrict__ offsets, VoxRec* __restrict__ out) {
    const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_q) return;
    const int vx = qkeys[3 * q], vy = qkeys[3 * q + 1], vz = qkeys[3 * q + 2];
    unsigned o = offsets[q];
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
                const Probe pr = probe_voxel(m, vx + dx, vy + dy, vz + dz);
                if (pr.vid < 0 || pr.cnt == 0) continue;
                VoxRec r;
                r.mx = m.vox_mean[(size_t)pr.vid * 3]; r.my = m.vox_mean[(size_t)pr.vid * 3 + 1]; r.mz = m.vox_mean[(size_t)pr.vid * 3 + 2];
                r.vid = pr.vid;
                r.pad = ((dx + 1) * 3 + (dy + 1)) * 3 + (dz + 1); // position code of this neighbour (AVGICP picks the face ones)
                out[o++] = r;
            }
}

// map build: size and content of the neighbourhood list of every query voxel (init time)
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

// Continuous batching: after the solve of an iteration, every slot whose registration has finished saves its final state
// and takes the next pending registration (descriptor + initial guess), so every accumulate launch stays full until the
// queue runs dry.  Slots are served in slot order by one thread: the assignment is deterministic (identical on every rank).
constexpr int kMaxSlots = 1024;
__global__ __launch_bounds__(1024) void k_stream_refill(ScanDesc* scans, ScanState* st, int slots, const QueueItem* __restrict__ queue,
                                                       const double* __restrict__ qT0, ScanState* out_state, StreamCtrl* ctrl, int first) {
    __shared__ int s_assign[kMaxSlots]; // registration to start in the slot, -1 = slot keeps going, -2 = slot goes idle
    __shared__ int s_save[kMaxSlots];   // registration whose final state is copied out, -1 = none
    // the slots' flags are fetched by all threads at once; the serial part below only touches LDS
    for (int s = threadIdx.x; s < slots; s += blockDim.x) s_save[s] = (!first && st[s].done && st[s].reg >= 0) ? st[s].reg : -1;
    __syncthreads();
    if (threadIdx.x == 0) {
        int next = first ? 0 : ctrl->next, completed = first ? 0 : ctrl->completed;
        const int total = ctrl->total;
        for (int s = 0; s < slots; ++s) {
            s_assign[s] = -1;
            if (!first && s_save[s] < 0) continue; // still iterating, or already idle
            if (!first) ++completed;
            s_assign[s] = (next < total) ? next++ : -2;
        }
        ctrl->next = next;
        ctrl->completed = completed;
    }
    __syncthreads();
    constexpr int W = (int)(sizeof(ScanState) / sizeof(double));
    for (int s = (int)(threadIdx.x >> 6); s < slots; s += (int)(blockDim.x >> 6)) { // one wavefront per slot
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

// Wave-parallel 6x6 LDL^T with diagonal pivoting (the pivot rule of Eigen's LDLT: largest |diagonal| of the trailing
// Schur complement, symmetric exchange), right-looking, on a matrix spread over lanes: lane l < 36 holds A[l/6][l%6].
// All 64 lanes execute it with uniform control flow.  Returns x = A^-1 b (uniform in every lane) and, when want_inv,
// leaves A^-1[l/6][l%6] in `inv_elem` of lane l < 36.  ~3 us instead of ~25 us for the single-lane version.
__device__ __forceinline__ void wave_ldlt6(double a, const double* b, double x[6], bool want_inv, double& inv_elem) {
    const int lane = threadIdx.x & 63;
    const int li = (lane < 36) ? lane / 6 : 0, lj = (lane < 36) ? lane % 6 : 0;
    unsigned active = 0x3Fu;
    int order[6];
    double piv[6];
    int rank_i = 6, rank_j = 6, rank_l = 6; // elimination step of row li / column lj / index `lane` (lane < 6)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int p = 0;
        double best = -1.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const double di = __shfl(a, i * 7, 64);
            if (((active >> i) & 1u) && fabs(di) > best) { best = fabs(di); p = i; }
        }
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

constexpr int kSolveThreads = 1024;
__global__ __launch_bounds__(kSolveThreads) void k_solve(const ScanDesc* __restrict__ scans, ScanState* st,
                                                         const double* __restrict__ partials, double* sums,
                                                         const RegParams rp, elm_iter_trace* trace, int mode, int* active) {
    const int s = blockIdx.x;
    ScanState& S = st[s];
    const int t = threadIdx.x;
    __shared__ double tot[kSums];
    __shared__ double part[kSolveThreads / 32][kSums];
    const bool done = S.done != 0;
    if (mode != 2) {
        // deterministic reduction of this scan's per-workgroup partial sums: 32 strided groups of 32 lanes read whole
        // 256-byte records (four independent loads in flight per lane), then the group sums are added in a fixed order
        const int k = t & 31, g = t >> 5;
        constexpr unsigned G = kSolveThreads / 32;
        double v = 0.0;
        if (!done) {
            const ScanDesc sd = scans[s];
            unsigned b = sd.blk_begin + g;
            double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
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
        part[g][k] = v;
        __syncthreads();
        if (t < 32) {
            double a = part[0][t];
#pragma unroll
            for (int q = 1; q < kSolveThreads / 32; ++q) a += part[q][t];
            if (mode == 1) sums[(size_t)s * kSums + t] = a; // zeros for finished scans keep the all-reduce buffer defined
            else tot[t] = a;
        }
        if (mode == 1) return;
    } else {
        if (t < 32) tot[t] = sums[(size_t)s * kSums + t];
    }
    __syncthreads();
    if (done || t >= 64) return; // the first wave does the rest with uniform control flow; lane 0 owns the state
    const bool lead = (t == 0);

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
        return;
    }
    const double fitness = tot[27] / n_corr; // d_fitness_score_ = d_residual_sum / source_global.size()

    // JTJ + lambda * diag(JTJ), one element per lane (reg.cpp:55-56 / 136-138 / 213-214)
    const int li = (t < 36) ? t / 6 : 0, lj = (t < 36) ? t % 6 : 0;
    const double hij = tot[tri(li < lj ? li : lj, li < lj ? lj : li)];
    const double a = (li == lj) ? hij + rp.lm_lambda * hij : hij;
    double x[6], inv_elem;
    wave_ldlt6(a, &tot[21], x, rp.method == ELM_GICP, inv_elem);
    if (rp.method == ELM_GICP && t < 36) S.local_cov[t] = inv_elem; // reg.cpp:141-142 (symmetric: layout-free)

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
    if (tr && t < 36) tr->JTJ[t] = hij; // symmetric
    if (!lead) return;
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
    }
}

// ------------------------------------------------------------------------------------------------------
// K3 / K4: map covariances
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_voxel_cov(const DevMap m, const uint2* __restrict__ ranges, double* vox_mean,
                                                   double* vox_cov) {
    const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m.n_vox) return;
    const uint2 rg = ranges[v];
    const unsigned n = rg.y;
    double mean[3] = {0, 0, 0};
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
        double nrm[3];
        plane_regularize(c, C, nrm);
    }
    for (int k = 0; k < 3; ++k) vox_mean[(size_t)v * 3 + k] = mean[k];
    for (int k = 0; k < 9; ++k) vox_cov[(size_t)v * 9 + k] = C[k];
}

__global__ __launch_bounds__(256) void k_point_cov(const DevMap m, double d2max, double* pt_gicp) {
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
    double* rec = pt_gicp + (size_t)i * 16;
    for (int k = 0; k < 3; ++k) { rec[k] = mean[k]; rec[12 + k] = nf[k]; }
    for (int k = 0; k < 9; ++k) rec[3 + k] = C[k];
    rec[15] = 0.0;
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
    const float A = (float)cos((double)yaw), B = (float)sin((double)yaw), Cc = (float)cos((double)pitch),
                D = (float)sin((double)pitch), E = (float)cos((double)roll), F = (float)sin((double)roll);
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
int debug_phase_cycles(unsigned long long* out16, int reset) {
#ifdef ELM_PHASE_TIMING
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 1;
#else
    (void)out16; (void)reset;
    return 0;
#endif
}

void launch_stream_refill(hipStream_t s, ScanDesc* scans, ScanState* st, int slots, const QueueItem* queue, const double* qT0,
                          ScanState* out_state, StreamCtrl* ctrl, int first) {
    hipLaunchKernelGGL(k_stream_refill, dim3(1), dim3(1024), 0, s, scans, st, slots, queue, qT0, out_state, ctrl, first);
}
void launch_init_state(hipStream_t s, ScanState* st, const double* T0, int batch, int map_empty, int* active) {
    hipLaunchKernelGGL(k_init_state, dim3((batch + 63) / 64), dim3(64), 0, s, st, T0, batch, map_empty, active);
}

void launch_accumulate(hipStream_t s, const DevMap& m, const ScanDesc* scans, int batch, int total_blocks,
                       ScanState* st, double* partials, const RegParams& rp, int direct) {
    dim3 g(total_blocks), b(kBlock);
#define ELM_LAUNCH(K, M) hipLaunchKernelGGL((K<M>), g, b, 0, s, m, scans, batch, (unsigned)total_blocks, st, partials, rp)
    if (direct) {
        switch (rp.method) {
        case ELM_P2P: ELM_LAUNCH(k_accumulate_direct, ELM_P2P); break;
        case ELM_GICP: ELM_LAUNCH(k_accumulate_direct, ELM_GICP); break;
        case ELM_VGICP: ELM_LAUNCH(k_accumulate_direct, ELM_VGICP); break;
        default: ELM_LAUNCH(k_accumulate_direct, ELM_AVGICP); break;
        }
    } else {
        switch (rp.method) {
        case ELM_P2P: ELM_LAUNCH(k_accumulate, ELM_P2P); break;
        case ELM_GICP: ELM_LAUNCH(k_accumulate, ELM_GICP); break;
        case ELM_VGICP: ELM_LAUNCH(k_accumulate, ELM_VGICP); break;
        default: ELM_LAUNCH(k_accumulate, ELM_AVGICP); break;
        }
    }
#undef ELM_LAUNCH
}

void launch_accumulate_nbr(hipStream_t s, const DevMap& m, const ScanDesc* scans, int batch, int total_blocks,
                           ScanState* st, double* partials, const RegParams& rp) {
    dim3 g(total_blocks), b(kBlock);
    if (rp.method == ELM_P2P)
        hipLaunchKernelGGL((k_accumulate_nbr<ELM_P2P>), g, b, 0, s, m, scans, batch, (unsigned)total_blocks, st, partials, rp);
    else
        hipLaunchKernelGGL((k_accumulate_nbr<ELM_GICP>), g, b, 0, s, m, scans, batch, (unsigned)total_blocks, st, partials, rp);
}
void launch_accumulate_cell(hipStream_t s, const DevMap& m, const ScanDesc* scans, int batch, int total_blocks,
                            ScanState* st, double* partials, const RegParams& rp) {
    dim3 g(total_blocks), b(kBlock);
    if (rp.method == ELM_P2P)
        hipLaunchKernelGGL((k_accumulate_cell<ELM_P2P>), g, b, 0, s, m, scans, batch, (unsigned)total_blocks, st, partials, rp);
    else
        hipLaunchKernelGGL((k_accumulate_cell<ELM_GICP>), g, b, 0, s, m, scans, batch, (unsigned)total_blocks, st, partials, rp);
}
void launch_accumulate_vnbr(hipStream_t s, const DevMap& m, const ScanDesc* scans, int batch, int total_blocks,
                            ScanState* st, double* partials, const RegParams& rp) {
    if (rp.method == ELM_VGICP)
        hipLaunchKernelGGL((k_accumulate_vnbr<ELM_VGICP>), dim3(total_blocks), dim3(kBlock), 0, s, m, scans, batch, (unsigned)total_blocks, st, partials, rp);
    else
        hipLaunchKernelGGL((k_accumulate_vnbr<ELM_AVGICP>), dim3(total_blocks), dim3(kBlock), 0, s, m, scans, batch, (unsigned)total_blocks, st, partials, rp);
}
void launch_vnbr_fill(hipStream_t s, const DevMap& m, const int32_t* qkeys, uint32_t n_q, const uint32_t* offsets, VoxRec* out) {
    hipLaunchKernelGGL(k_vnbr_fill, dim3((n_q + 255) / 256), dim3(256), 0, s, m, qkeys, n_q, offsets, out);
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
                  double* sums, const RegParams& rp, elm_iter_trace* trace, int mode, int* active) {
    hipLaunchKernelGGL(k_solve, dim3(batch), dim3(kSolveThreads), 0, s, scans, st, partials, sums, rp, trace, mode, active);
}

void launch_voxel_cov(hipStream_t s, const DevMap& m, const uint2* ranges, double* vox_mean, double* vox_cov) {
    hipLaunchKernelGGL(k_voxel_cov, dim3((m.n_vox + 255) / 256), dim3(256), 0, s, m, ranges, vox_mean, vox_cov);
}

void launch_point_cov(hipStream_t s, const DevMap& m, double d2max, double* pt_gicp) {
    hipLaunchKernelGGL(k_point_cov, dim3((m.n_pts + 255) / 256), dim3(256), 0, s, m, d2max, pt_gicp);
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
                                                         const unsigned* __restrict__ block_offset, float4* __restrict__ out) {
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
    if (keep) out[pos] = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 0.f);
}

void launch_voxel_downsample(hipStream_t s, const float* xyz, uint32_t n, double vs, unsigned long long* table, unsigned* first,
                             unsigned cap_log2, unsigned* slot, unsigned* block_count, unsigned* total, int* overflow, float4* out) {
    const unsigned nb = (n + kDsBlock - 1) / kDsBlock;
    hipLaunchKernelGGL(k_ds_insert, dim3((n + 255) / 256), dim3(256), 0, s, xyz, n, vs, table, first, cap_log2, slot, overflow);
    hipLaunchKernelGGL(k_ds_count, dim3(nb), dim3(kDsBlock), 0, s, first, slot, n, block_count);
    hipLaunchKernelGGL(k_ds_offsets, dim3(1), dim3(1024), 0, s, block_count, nb, total);
    hipLaunchKernelGGL(k_ds_scatter, dim3(nb), dim3(kDsBlock), 0, s, xyz, first, slot, n, block_count, out);
}

void launch_deskew(hipStream_t s, const float* xyz, const float* rel_time, uint32_t n, const DeskewDev& d, float* xyz_out) {
    hipLaunchKernelGGL(k_deskew, dim3((n + 255) / 256), dim3(256), 0, s, xyz, rel_time, n, d, xyz_out);
}

} // namespace elm
