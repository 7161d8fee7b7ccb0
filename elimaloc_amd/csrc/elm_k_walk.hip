// elm_k_walk.hip -- k_accumulate_direct (the plain 27-probe walk), k_accumulate_radar (per-pair arithmetic), k_query_direct
// (one translation unit of the kernel library: see elm_kernels.md / DESIGN.md section 4; split from the former elm_kernels.hip in round 6)
#include <float.h>
#include <algorithm>
#include <hip/hip_runtime.h>

#include "elm_internal.hpp"
#include "elm_la.hpp"
#include "elm_dev_pairs.hpp"

namespace elm {

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

} // namespace elm
