// elm_k_cell.hip -- k_accumulate_cell: the two-stage search on per-query-voxel neighbourhood lists (fall-back index)
// (one translation unit of the kernel library: see elm_kernels.md / DESIGN.md section 4; split from the former elm_kernels.hip in round 6)
#include <float.h>
#include <algorithm>
#include <hip/hip_runtime.h>

#include "elm_internal.hpp"
#include "elm_la.hpp"
#include "elm_dev_grid.hpp"

namespace elm {

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
    // maps with an asymmetric stored inverse (GICP reads stored inverses here): the workgroup's antisymmetric side record (round 6: the
    // fall-back index carries it like the grid kernel, so such a map no longer drops to the per-pair kernels)
    __shared__ double s_asym[(METHOD == ELM_GICP) ? kAsymSums : 1];
    __shared__ unsigned s_hitw[kBlock / 64];
    PairSum PA; // this lane's pair once more in the factored form -- filled only on such maps (rp.asym, uniform): A = w C^-1 feeds the side sums
    pair_sum_zero(PA);
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
            if (METHOD == ELM_GICP && rp.asym && bidx >= 0 && bd2 < rp.th2) { // (no bucket at all: the identity covariance -- symmetric)
                const double* __restrict__ rec = m.pt_gicp + (size_t)bidx * 16;
                double Ci[9], nf[3];
#pragma unroll
                for (int k = 0; k < 9; ++k) Ci[k] = rec[3 + k];
                nf[0] = rec[12]; nf[1] = rec[13]; nf[2] = rec[14];
                pair_sum_single<ELM_GICP>(PA, rec[0] - gx, rec[1] - gy, rec[2] - gz, Ci, nf, rp); // finish_point_pair's weight and target (reg.cpp:97)
            }
        }
        if (rp.stats) { // (as in the grid / voxel-list kernels: 0 unless the work counters are switched on)
            v[NV - 3] = (double)qp.cnt;  // candidates of the reference's walk
            v[NV - 2] = (double)qp.nocc; // occupied neighbour voxels
            v[NV - 1] = (double)n_tested + (hard ? kFallbackUnit : 0.0); // high part: points served by stage 2
        }
    }
    if (METHOD == ELM_GICP) asym_mark(PA.A, rp, s_hitw);
    block_reduce_to_lds<NV, kRedPass>(v, s_buf, s_red);
    if (METHOD == ELM_GICP) {
        const double red_keep = (threadIdx.x < kSums) ? s_red[threadIdx.x] : 0.0; // (the side record's reduction reuses the buffers)
        asym_side_store(PA.A, gx - S.T[12], gy - S.T[13], gz - S.T[14], L, rp, s_buf, s_asym, s_hitw);
        publish_and_reduce(red_keep, L, s, sd.blk_begin, sd.blk_end, partials, rp, s_buf);
        return;
    }
    publish_and_reduce((threadIdx.x < kSums) ? ((METHOD == ELM_P2P) ? p2p_expand(s_red, (int)threadIdx.x) : s_red[threadIdx.x]) : 0.0, L, s, sd.blk_begin,
                       sd.blk_end, partials, rp, s_buf);
}

void launch_accumulate_cell(hipStream_t s, const DevMap& m, const ScanDesc* scans, int batch, int total_blocks,
                            ScanState* st, double* partials, const RegParams& rp) {
    dim3 g(total_blocks), b(kBlock);
    if (rp.method == ELM_P2P)
        hipLaunchKernelGGL((k_accumulate_cell<ELM_P2P>), g, b, 0, s, m, scans, batch, (unsigned)total_blocks, st, partials, rp);
    else
        hipLaunchKernelGGL((k_accumulate_cell<ELM_GICP>), g, b, 0, s, m, scans, batch, (unsigned)total_blocks, st, partials, rp);
}

} // namespace elm
