// elm_k_vnbr.hip -- k_accumulate_vnbr: nearest voxel mean / face neighbours on the voxel-mean lists (VGICP, AVGICP)
// (one translation unit of the kernel library: see elm_kernels.md / DESIGN.md section 4; split from the former elm_kernels.hip in round 6)
#include <float.h>
#include <algorithm>
#include <hip/hip_runtime.h>

#include "elm_internal.hpp"
#include "elm_la.hpp"
#include "elm_dev_grid.hpp"

namespace elm {


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
            if (cnt != 0u) {
                const unsigned nblk = (cnt + 3u) >> 2, blk0 = start >> 2;
                // distances to gh = float32(g): |g - gh| joins the stored means' rounding in the margin of the decision
                const float ghx = (float)gx, ghy = (float)gy, ghz = (float)gz;
                const float eg = (fabsf((float)(gx - (double)ghx)) + fabsf((float)(gy - (double)ghy)) + fabsf((float)(gz - (double)ghz))) * 1.000001f;
                const f32x2 gxy = {ghx, ghy}, gzz = {ghz, 0.f};
                unsigned m1 = 0x7F800000u, m2 = 0x7F800000u, jb = 0u;
                for (unsigned b0 = 0; b0 < nblk; b0 += kVnbrBlks) {
                    GridBlk B[kVnbrBlks]; // the means of the block's four slots (48 of its 64 bytes: the slot words are read for the winner alone)
#pragma unroll
                    for (int u = 0; u < kVnbrBlks; ++u) B[u] = m.vnbr_blk[(b0 + u < nblk) ? blk0 + b0 + u : m.vnbr_pad_blk].g;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < kVnbrBlks; ++u) {
                        f32x2 da, db;
                        blk_dist_h(B[u], gxy, gzz, da, db);
                        const unsigned was = m1;
                        two_smallest(da.x, 0u, m1, m2);
                        two_smallest(da.y, 1u, m1, m2);
                        two_smallest(db.x, 2u, m1, m2);
                        two_smallest(db.y, 3u, m1, m2);
                        jb = (m1 != was) ? b0 + (unsigned)u : jb;
                    }
                }
                // |float32(mean) - mean| <= 2^-24 |mean|_1 <= 6.5e-8 (|g|_1 + 6 voxel sizes); float32 arithmetic + key bits: 2^-18
                const float em = 6.5e-8f * (fabsf(ghx) + fabsf(ghy) + fabsf(ghz) + 6.0f * (float)m.voxel_size) + eg;
                const float s1 = __builtin_sqrtf(__uint_as_float(m1 & ~3u)), s2 = __builtin_sqrtf(__uint_as_float(m2 & ~3u));
                if (s2 - s2 * 3.814697265625e-06f - em > s1 + s1 * 3.814697265625e-06f + em) {
                    bj = jb * 4u + (m1 & 3u); // the winner's slot of the list
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
            if (cnt) { // the winner's slot word (the line its block's means came from: an L1 hit), then its float64 record from the per-voxel table
                const VoxRec w = m.vox_rec[vnbr_vid(m, start + min(bj, cnt - 1))];
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

} // namespace elm
