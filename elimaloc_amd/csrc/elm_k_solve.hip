// elm_k_solve.hip -- k_init_*, k_stream_refill, k_solve (reduction, gates, LDLT, exp, compose), k_align_* (AlignCloudsLocal* on explicit pairs)
// (one translation unit of the kernel library: see elm_kernels.md / DESIGN.md section 4; split from the former elm_kernels.hip in round 6)
#include <float.h>
#include <algorithm>
#include <hip/hip_runtime.h>

#include "elm_internal.hpp"
#include "elm_la.hpp"
#include "elm_dev_reduce.hpp"

namespace elm {

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
// NT = 1024 threads.  The reduction adds the partial records in the order of 32 groups of 32 lanes.
// id of a slot's (registration, iteration) in the exchanged record (RegParams::rank_check).  The registration index enters modulo 2^19 so
// that n id^2 stays an exact integer in a double for any number of registrations per call (id < 2^23 + 16, id^2 < 2^47, n <= 64 ranks:
// below 2^53); an idle slot (registration -1) has id = iteration & 15.  Two ranks that swap registrations r and r + 2^19 k in one slot, or
// that swap two registrations between two slots while each slot agrees with itself across the ranks, are not told apart: the check sees
// ranks that DISAGREE about a slot, which is what a broken collective or a non-deterministic solve produces.
__device__ __forceinline__ double rank_check_id(int reg, int iters) {
    return 16.0 * (double)((reg < 0 ? -1 : (reg & 0x7FFFF)) + 1) + (double)(iters & 15);
}
template <int NT>
__global__ __launch_bounds__(NT, 1) void k_solve(const ScanDesc* scans, ScanState* st,
                                                         const double* __restrict__ partials, double* sums,
                                                         const RegParams rp, elm_iter_trace* trace, int mode, int* active,
                                                         const StreamArgs sa) {
    const int s = blockIdx.x;
    ScanState& S = st[s];
    const int t = threadIdx.x;
    __shared__ double tot[kSums];
    __shared__ double part[kSolveThreads / 32][kSums];
    static_assert(NT == kSolveThreads, "solve workgroup size");
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
            a = sums[(size_t)s * kRadarSums + t]; // (mode 2: the all-reduced sums -- also of a scan that has no point on THIS rank)
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
        // mode 2: the all-reduced sums mode 1 left in `sums` on every rank.  A scan without a single workgroup HERE (no point of it on this
        // rank) reads them like any other: the other ranks' pairs are in there.  (Round 6: rounds 2-5 zeroed them for such a scan -- a
        // leftover of the fused reduction -- so a rank without points saw n_corr = 0 and failed the overlap gate while the others went on;
        // found by tests/test_group.py::test_group_with_tiny_and_empty_shards.)
        if (t < 32) tot[t] = sums[(size_t)s * kSums + t];
        else if (asym && t < 32 + kAsymSums) dsum[t - 32] = rp.asym_sums ? rp.asym_sums[(size_t)s * kAsymSums + (t - 32)] : 0.0;
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

// host-fed streams start with every slot idle: the solve hands out registrations as their scans arrive
__global__ void k_slots_idle(ScanDesc* scans, ScanState* st, int slots, unsigned cap_blocks) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= slots) return;
    scans[s].pts = nullptr; scans[s].n = 0; scans[s].n_total = 0;
    scans[s].blk_begin = cap_blocks * (unsigned)s; scans[s].blk_end = cap_blocks * (unsigned)(s + 1); // every slot owns cap_blocks workgroups
    st[s].done = 1;
    st[s].reg = -1;
}

// host-fed streams: the upload stream publishes how many scans have landed (after their ordering kernel, same stream)
__global__ void k_publish_ready(StreamCtrl* ctrl, int ready) {
    __hip_atomic_store(&ctrl->ready, ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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

void launch_solve(hipStream_t s, const ScanDesc* scans, int batch, ScanState* st, const double* partials,
                  double* sums, const RegParams& rp, elm_iter_trace* trace, int mode, int* active, const StreamArgs* refill) {
    StreamArgs sa = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
    if (refill) sa = *refill;
    // one wavefront per scan when the sums are already reduced (the second half of a multi-rank iteration)
    const int threads = mode == 2 ? 64 : kSolveThreads;
    hipLaunchKernelGGL(k_solve<kSolveThreads>, dim3(batch), dim3(threads), 0, s, scans, st, partials, sums, rp, trace, mode, active, sa);
}

void launch_publish_ready(hipStream_t s, StreamCtrl* ctrl, int ready) { hipLaunchKernelGGL(k_publish_ready, dim3(1), dim3(1), 0, s, ctrl, ready); }

void launch_slots_idle(hipStream_t s, ScanDesc* scans, ScanState* st, int slots, unsigned cap_blocks) {
    hipLaunchKernelGGL(k_slots_idle, dim3((slots + 255) / 256), dim3(256), 0, s, scans, st, slots, cap_blocks);
}

int stream_max_slots() { return kMaxSlots; }

} // namespace elm
