// elm_k_map.hip -- map build: covariances, GICP payload gather, walk statistics, neighbourhood / voxel-mean list fills, face sublists
// (one translation unit of the kernel library: see elm_kernels.md / DESIGN.md section 4; split from the former elm_kernels.hip in round 6)
#include <float.h>
#include <algorithm>
#include <hip/hip_runtime.h>

#include "elm_internal.hpp"
#include "elm_la.hpp"
#include "elm_dev_reduce.hpp"

namespace elm {

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

void launch_gather_gicp(hipStream_t s, const DevMap& m, size_t n_slots, double* out, int compact) {
    const size_t threads = n_slots * (compact ? 8 : 16);
    if (compact) hipLaunchKernelGGL(k_gather_gicp<1>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, m, n_slots, out);
    else hipLaunchKernelGGL(k_gather_gicp<0>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, m, n_slots, out);
}

void launch_vox_stat(hipStream_t s, const DevMap& m, uint32_t* out) {
    const size_t n = (size_t)m.vnx * m.vny * m.vnz;
    hipLaunchKernelGGL(k_vox_stat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, m, out);
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

void launch_nbr_count(hipStream_t s, const DevMap& m, const int32_t* qkeys, uint32_t n_q, uint32_t* counts, uint32_t* nocc) {
    hipLaunchKernelGGL(k_nbr_count, dim3((n_q + 255) / 256), dim3(256), 0, s, m, qkeys, n_q, counts, nocc);
}

void launch_nbr_fill(hipStream_t s, const DevMap& m, const int32_t* qkeys, uint32_t n_q, const uint32_t* offsets, Pt3* out,
                     uint32_t* out_idx) {
    const uint64_t threads = (uint64_t)n_q * 32;
    hipLaunchKernelGGL(k_nbr_fill, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, m, qkeys, n_q, offsets, out, out_idx);
}

void launch_voxel_cov(hipStream_t s, const DevMap& m, const uint2* ranges, double* vox_mean, double* vox_cov, double* vox_cinv, double* vox_nk, unsigned* bad) {
    hipLaunchKernelGGL(k_voxel_cov, dim3((m.n_vox + 255) / 256), dim3(256), 0, s, m, ranges, vox_mean, vox_cov, vox_cinv, vox_nk, bad);
}

void launch_point_cov(hipStream_t s, const DevMap& m, double d2max, double* pt_gicp, double* pt_cov, unsigned* bad) {
    hipLaunchKernelGGL(k_point_cov, dim3((m.n_pts + 255) / 256), dim3(256), 0, s, m, d2max, pt_gicp, pt_cov, bad);
}

size_t nbr_cell_stride() { return (size_t)kCellStride; }

} // namespace elm
