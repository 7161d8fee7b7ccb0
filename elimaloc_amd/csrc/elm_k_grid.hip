// elm_k_grid.hip -- k_accumulate_grid: the two-stage search on the dense / two-level cell grid (P2P, GICP; the headline kernel)
// (one translation unit of the kernel library: see elm_kernels.md / DESIGN.md section 4; split from the former elm_kernels.hip in round 6)
#include <float.h>
#include <algorithm>
#include <hip/hip_runtime.h>

#include "elm_internal.hpp"
#include "elm_la.hpp"
#include "elm_dev_grid.hpp"

namespace elm {

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

} // namespace elm
