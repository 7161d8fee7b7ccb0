// elm_k_scan.hip -- per-scan kernels: k_deskew, device VoxelDownsample (k_ds_*), scan ordering (k_scan_order, k_order_*)
// (one translation unit of the kernel library: see elm_kernels.md / DESIGN.md section 4; split from the former elm_kernels.hip in round 6)
#include <float.h>
#include <algorithm>
#include <hip/hip_runtime.h>

#include "elm_internal.hpp"
#include "elm_la.hpp"
#include "elm_dev_pairs.hpp"

namespace elm {

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
void launch_deskew(hipStream_t s, const float* xyz, const float* rel_time, uint32_t n, const DeskewDev& d, float* xyz_out) {
    hipLaunchKernelGGL(k_deskew, dim3((n + 255) / 256), dim3(256), 0, s, xyz, rel_time, n, d, xyz_out);
}

} // namespace elm
