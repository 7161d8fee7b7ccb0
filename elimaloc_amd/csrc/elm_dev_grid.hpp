// elm_dev_grid.hpp -- cell-grid addressing, float32 block distances, the stage-2 ball walk
// Device-side helpers shared by the kernel translation units (every function is inline / a template: no symbol is emitted by itself).
#pragma once
#include <float.h>
#include <hip/hip_runtime.h>

#include "elm_internal.hpp"
#include "elm_la.hpp"
#include "elm_dev_reduce.hpp"

namespace elm {

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

} // namespace elm
