// Developer probe (MI355X): what does ONE wave64 VALU instruction cost a SIMD, per instruction class, and what do the SQ counters the
// roofline is quoted from (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES) read for a kernel whose issue rate is KNOWN?
//
// Every kernel is a loop over one block of 32 instructions of ONE class on 8 independent accumulators (no dependent pair closer than
// 8 instructions), bracketed by s_memtime.  (s_memtime is a CONSTANT-rate counter on this chip, not the shader clock: the first run read
// 2.0 shader cycles -- GRBM_GUI_ACTIVE / 8 of the same launch -- per tick at the ~2.05 GHz the kernels ran at; the columns below are in
// ticks, the counter table of run_valu_probe.sh is in shader cycles.)  It is launched so that every SIMD of the chip holds exactly W waves
// (W = 1, 2, 4, 8: workgroups of 256 threads = one wave per SIMD, W workgroups per CU held apart by their LDS allocation) and prints
//   cyc/inst/wave  = elapsed ticks of one wave / its instructions                  (what a lone wave sees: issue + dependency)
//   cyc/inst/SIMD  = elapsed ticks / (W x instructions)                                  (the SIMD's issue cost per wave64 instruction
//                                                                                    once enough waves hide the dependencies)
// Under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE` the same launches calibrate
// the counters (tools/probes/run_valu_probe.sh -> profiles/r04_valu_probe.txt).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_probe.hip -o tools/probes/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

// 32 instructions: 4 rounds over 8 accumulators.  I(n) expands to the instruction text for accumulator operand %n.
#define R8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define R32(I) R8(I) R8(I) R8(I) R8(I)
#define ACC8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

#define I_FMA_F32(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_PK_FMA_F32(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_PK_ADD_F32(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
#define I_PK_MUL_F32(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define I_FMA_F64(n) "v_fma_f64 %" #n ", %" #n ", %8, %9\n"
#define I_ADD_F64(n) "v_add_f64 %" #n ", %" #n ", %8\n"
#define I_MUL_F64(n) "v_mul_f64 %" #n ", %" #n ", %8\n"
#define I_AND_OR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define I_MED3(n) "v_med3_u32 %" #n ", %" #n ", %8, %9\n"
#define I_MIN_U32(n) "v_min_u32 %" #n ", %" #n ", %8\n"
#define I_ADD_U32(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_LSHL_ADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 1, %8\n"
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_MUL_LO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define I_RCP_F32(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define I_SQRT_F32(n) "v_sqrt_f32 %" #n ", %" #n "\n"
#define I_CVT_F64_F32(n) "v_cvt_f32_f64 %" #n ", %8\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_ADD_F32(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define I_MAX_F32(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define I_AND_B32(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_LSHLREV(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define I_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 1, 30\n"
#define I_CMP(n) "v_cmp_lt_u32 vcc, %" #n ", %8\n"
#define I_CNDMASK_S(n) "v_cndmask_b32 %" #n ", %" #n ", %8, s[10:11]\n"
#define I_CMP_CND(n) "v_cmp_lt_u32 vcc, %" #n ", %9\n v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"

enum Kind { FMA_F32, PK_FMA_F32, PK_ADD_F32, PK_MUL_F32, FMA_F64, ADD_F64, MUL_F64, AND_OR, MED3, MIN_U32, ADD_U32, LSHL_ADD, CNDMASK, MUL_LO, RCP_F32,
            SQRT_F32, CVT_F32_F64, MOV, MIX_GRID, ADD_F32, MAX_F32, AND_B32, LSHLREV, BFE, CMP, CNDMASK_S, CMP_CND, N_KIND };
static const char* kKindName[N_KIND] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_and_or_b32",
                                        "v_med3_u32", "v_min_u32", "v_add_u32", "v_lshl_add_u32", "v_cndmask_b32", "v_mul_lo_u32", "v_rcp_f32", "v_sqrt_f32",
                                        "v_cvt_f32_f64", "v_mov_b32", "mix: 12 pk_f32 + 12 int (and_or/med3/min) + 8 fma_f64", "v_add_f32", "v_max_f32", "v_and_b32",
                                        "v_lshlrev_b32", "v_bfe_u32", "v_cmp_lt_u32 (vcc)", "v_cndmask_b32 (mask in s[10:11])",
                                        "v_cmp_lt_u32 + v_cndmask_b32 (per PAIR)"};

__device__ __forceinline__ uint64_t now() {
    uint64_t t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

template <int KIND>
__global__ __launch_bounds__(256) void k_valu(uint64_t* __restrict__ out, int iters, float seed) {
    extern __shared__ char lds_pad[]; // sized by the host so that exactly W workgroups fit a CU
    if (seed == 12345.f) lds_pad[threadIdx.x] = 1;
    uint64_t t0, t1;
    if (KIND == FMA_F32 || KIND == RCP_F32 || KIND == SQRT_F32 || KIND == MOV || KIND == ADD_F32 || KIND == MAX_F32) {
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
        const float b = 0.999f, c = 0.001f;
        t0 = now();
        for (int i = 0; i < iters; ++i) {
            if (KIND == FMA_F32) asm volatile(R32(I_FMA_F32) : ACC8 : "v"(b), "v"(c));
            if (KIND == RCP_F32) asm volatile(R32(I_RCP_F32) : ACC8 : "v"(b), "v"(c));
            if (KIND == SQRT_F32) asm volatile(R32(I_SQRT_F32) : ACC8 : "v"(b), "v"(c));
            if (KIND == MOV) asm volatile(R32(I_MOV) : ACC8 : "v"(b), "v"(c));
            if (KIND == ADD_F32) asm volatile(R32(I_ADD_F32) : ACC8 : "v"(c), "v"(b));
            if (KIND == MAX_F32) asm volatile(R32(I_MAX_F32) : ACC8 : "v"(c), "v"(b));
        }
        t1 = now();
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345f) out[0] = 1;
    } else if (KIND == PK_FMA_F32 || KIND == PK_ADD_F32 || KIND == PK_MUL_F32) {
        f32x2 a0 = {seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
        const f32x2 b = {0.999f, 0.998f}, c = {0.001f, 0.002f};
        t0 = now();
        for (int i = 0; i < iters; ++i) {
            if (KIND == PK_FMA_F32) asm volatile(R32(I_PK_FMA_F32) : ACC8 : "v"(b), "v"(c));
            if (KIND == PK_ADD_F32) asm volatile(R32(I_PK_ADD_F32) : ACC8 : "v"(c), "v"(b));
            if (KIND == PK_MUL_F32) asm volatile(R32(I_PK_MUL_F32) : ACC8 : "v"(b), "v"(c));
        }
        t1 = now();
        if (a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y == 1.2345f) out[0] = 1;
    } else if (KIND == FMA_F64 || KIND == ADD_F64 || KIND == MUL_F64) {
        double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
        const double b = 0.999, c = 0.001;
        t0 = now();
        for (int i = 0; i < iters; ++i) {
            if (KIND == FMA_F64) asm volatile(R32(I_FMA_F64) : ACC8 : "v"(b), "v"(c));
            if (KIND == ADD_F64) asm volatile(R32(I_ADD_F64) : ACC8 : "v"(c), "v"(b));
            if (KIND == MUL_F64) asm volatile(R32(I_MUL_F64) : ACC8 : "v"(b), "v"(c));
        }
        t1 = now();
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345) out[0] = 1;
    } else if (KIND == CVT_F32_F64) {
        float a0 = seed, a1 = seed, a2 = seed, a3 = seed, a4 = seed, a5 = seed, a6 = seed, a7 = seed;
        const double b = 0.999 + seed, c = 0.001;
        t0 = now();
        for (int i = 0; i < iters; ++i) asm volatile(R32(I_CVT_F64_F32) : ACC8 : "v"(b), "v"(c));
        t1 = now();
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345f) out[0] = 1;
    } else if (KIND == MIX_GRID) {
        // the candidate loop's mix: packed float32 distance arithmetic, integer key bookkeeping, float64 pair arithmetic
        f32x2 p0 = {seed, seed}, p1 = p0 + 1.f, p2 = p0 + 2.f;
        unsigned u0 = (unsigned)seed, u1 = u0 + 1, u2 = u0 + 2;
        double d0 = seed, d1 = seed + 1.0;
        const f32x2 pb = {0.999f, 0.998f}, pc = {0.001f, 0.002f};
        const unsigned ub = 0x7fffffffu, uc = 3u;
        const double db = 0.999, dc = 0.001;
        t0 = now();
        for (int i = 0; i < iters; ++i) {
            asm volatile(
                "v_pk_fma_f32 %0, %0, %8, %9\n v_and_or_b32 %3, %3, %10, %11\n v_pk_add_f32 %1, %1, %9\n v_med3_u32 %4, %4, %10, %11\n"
                "v_pk_mul_f32 %2, %2, %8\n v_min_u32 %5, %5, %10\n v_fma_f64 %6, %6, %12, %13\n v_fma_f64 %7, %7, %12, %13\n"
                "v_pk_fma_f32 %0, %0, %8, %9\n v_and_or_b32 %3, %3, %10, %11\n v_pk_add_f32 %1, %1, %9\n v_med3_u32 %4, %4, %10, %11\n"
                "v_pk_mul_f32 %2, %2, %8\n v_min_u32 %5, %5, %10\n v_fma_f64 %6, %6, %12, %13\n v_fma_f64 %7, %7, %12, %13\n"
                "v_pk_fma_f32 %0, %0, %8, %9\n v_and_or_b32 %3, %3, %10, %11\n v_pk_add_f32 %1, %1, %9\n v_med3_u32 %4, %4, %10, %11\n"
                "v_pk_mul_f32 %2, %2, %8\n v_min_u32 %5, %5, %10\n v_fma_f64 %6, %6, %12, %13\n v_fma_f64 %7, %7, %12, %13\n"
                "v_pk_fma_f32 %0, %0, %8, %9\n v_and_or_b32 %3, %3, %10, %11\n v_pk_add_f32 %1, %1, %9\n v_med3_u32 %4, %4, %10, %11\n"
                "v_pk_mul_f32 %2, %2, %8\n v_min_u32 %5, %5, %10\n v_fma_f64 %6, %6, %12, %13\n v_fma_f64 %7, %7, %12, %13\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(d0), "+v"(d1)
                : "v"(pb), "v"(pc), "v"(ub), "v"(uc), "v"(db), "v"(dc));
        }
        t1 = now();
        if (p0.x + p1.x + p2.y + (float)(u0 + u1 + u2) + (float)(d0 + d1) == 1.2345f) out[0] = 1;
    } else {
        unsigned a0 = (unsigned)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        const unsigned b = 0x7ffffffbu, c = 3u;
        t0 = now();
        for (int i = 0; i < iters; ++i) {
            if (KIND == AND_OR) asm volatile(R32(I_AND_OR) : ACC8 : "v"(b), "v"(c));
            if (KIND == MED3) asm volatile(R32(I_MED3) : ACC8 : "v"(b), "v"(c));
            if (KIND == MIN_U32) asm volatile(R32(I_MIN_U32) : ACC8 : "v"(b), "v"(c));
            if (KIND == ADD_U32) asm volatile(R32(I_ADD_U32) : ACC8 : "v"(c), "v"(b));
            if (KIND == LSHL_ADD) asm volatile(R32(I_LSHL_ADD) : ACC8 : "v"(c), "v"(b));
            if (KIND == CNDMASK) asm volatile(R32(I_CNDMASK) : ACC8 : "v"(c), "v"(b) : "vcc");
            if (KIND == MUL_LO) asm volatile(R32(I_MUL_LO) : ACC8 : "v"(c), "v"(b));
            if (KIND == AND_B32) asm volatile(R32(I_AND_B32) : ACC8 : "v"(b), "v"(c));
            if (KIND == LSHLREV) asm volatile(R32(I_LSHLREV) : ACC8 : "v"(b), "v"(c));
            if (KIND == BFE) asm volatile(R32(I_BFE) : ACC8 : "v"(b), "v"(c));
            if (KIND == CMP) asm volatile(R32(I_CMP) : ACC8 : "v"(b), "v"(c) : "vcc");
            if (KIND == CNDMASK_S) asm volatile("s_mov_b64 s[10:11], 0x5555\n" R32(I_CNDMASK_S) : ACC8 : "v"(c), "v"(b) : "s10", "s11");
            if (KIND == CMP_CND) asm volatile(R8(I_CMP_CND) R8(I_CMP_CND) : ACC8 : "v"(c), "v"(b) : "vcc"); // 16 pairs = 32 instructions
        }
        t1 = now();
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345u) out[0] = 1;
    }
    if ((threadIdx.x & 63) == 0) out[1 + (size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kern_t)(uint64_t*, int, float);
template <int K>
static void fill(kern_t* tab) {
    tab[K] = k_valu<K>;
    if constexpr (K + 1 < N_KIND) fill<K + 1>(tab);
}

int main(int argc, char** argv) {
    kern_t tab[N_KIND];
    fill<0>(tab);
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iters = 2000; // x 32 instructions
    const int only_w = argc > 1 ? atoi(argv[1]) : 0; // counter passes: one occupancy per process keeps the per-kernel averages clean
    uint64_t* d = nullptr;
    const size_t n_out = 1 + (size_t)cus * 8 * 4;
    CHK(hipMalloc(&d, n_out * 8));
    std::vector<uint64_t> h(n_out);
    printf("# %s, %d CUs, clock %d MHz; %d x 32 instructions per wave\n", prop.name, cus, prop.clockRate / 1000, iters);
    printf("%-58s %3s %14s %14s %10s\n", "instruction", "W", "tick/inst/wave", "tick/inst/SIMD", "wall_us");
    for (int k = 0; k < N_KIND; ++k) {
        for (int W : {1, 2, 4, 8}) {
            if (only_w && W != only_w) continue;
            // W workgroups (4 waves: one per SIMD) per CU: each takes 1 / W of the 160 KB of LDS (minus a little), so no CU holds W + 1
            const size_t lds = std::min<size_t>((size_t)(160 * 1024) / W - 1024, 64 * 1024);
            CHK(hipFuncSetAttribute((const void*)tab[k], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CHK(hipMemset(d, 0, n_out * 8));
            hipEvent_t e0, e1;
            CHK(hipEventCreate(&e0));
            CHK(hipEventCreate(&e1));
            hipLaunchKernelGGL(tab[k], dim3(cus * W), dim3(256), lds, 0, d, 16, 1.5f); // warm-up (code cache, clocks)
            CHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(tab[k], dim3(cus * W), dim3(256), lds, 0, d, iters, 1.5f);
            CHK(hipEventRecord(e1, 0));
            CHK(hipDeviceSynchronize());
            float ms = 0.f;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            CHK(hipMemcpy(h.data(), d, n_out * 8, hipMemcpyDeviceToHost));
            std::vector<uint64_t> dt(h.begin() + 1, h.begin() + 1 + (size_t)cus * W * 4);
            std::sort(dt.begin(), dt.end());
            const double med = (double)dt[dt.size() / 2];
            const double n_inst = (double)iters * 32.0;
            printf("%-58s %3d %14.3f %14.3f %10.1f\n", kKindName[k], W, med / n_inst, med / n_inst / W, ms * 1e3);
            CHK(hipEventDestroy(e0));
            CHK(hipEventDestroy(e1));
        }
    }
    return 0;
}
