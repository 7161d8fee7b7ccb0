// Developer probe (MI355X): the rate at which a CU's vector L1 serves gathers that HIT it -- the unit the grid kernels sit on
// (1.0 TCP_TOTAL_CACHE_ACCESSES per CU-cycle at the bench's operating point).  Every lane issues independent loads (no dependent
// chain) inside a 16 KB window that every CU holds in its L1; 8 waves per SIMD.  Patterns per wave-level load instruction:
//   0  dwordx4, every lane its own 128-byte line            1  dword, every lane its own line
//   2  dwordx3, every lane its own line                     3  three dword loads of ONE line per lane (x[k], y[k], z[k] of a block)
//   4  dwordx4, four lanes per 64 bytes (one line per quad) 5  dwordx4, all lanes the same 16 bytes
//   6  dwordx4, two lanes per 128-byte line (own 64-byte half)   7  three dwordx4 rows of ONE 48-byte block per lane (the grid kernel)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/l1_probe.hip -o /tmp/l1_probe && /tmp/l1_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f3 __attribute__((ext_vector_type(3)));

// the load instruction is written out (the compiler would narrow a 16-byte load to the dwords that are used)
#define LD4(dst, off, imm) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #imm : "=v"(dst) : "v"(off), "s"(base))
#define LD3(dst, off, imm) asm volatile("global_load_dwordx3 %0, %1, %2 offset:" #imm : "=v"(dst) : "v"(off), "s"(base))
#define LD1(dst, off, imm) asm volatile("global_load_dword %0, %1, %2 offset:" #imm : "=v"(dst) : "v"(off), "s"(base))

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_l1(const float* __restrict__ base, int iters, float* out) {
    const unsigned lane = threadIdx.x & 63u;
    unsigned o; // byte offset inside a 16 KB window
    if (MODE == 0 || MODE == 1 || MODE == 2 || MODE == 3) o = lane * 128u;            // own 128-byte line
    else if (MODE == 4) o = (lane >> 2) * 128u + (lane & 3u) * 16u;                     // quad = 64 contiguous bytes
    else if (MODE == 5) o = 0u;
    else if (MODE == 6) o = (lane >> 1) * 128u + (lane & 1u) * 64u;
    else o = lane * 144u;                                                               // 48-byte blocks 144 bytes apart (one or two lines)
    o += (threadIdx.x >> 6) * 1024u; // the four waves of a workgroup start in different parts of the window
    unsigned oa = o & 16383u, ob = (o + 4096u) & 16383u, oc = (o + 2048u) & 16383u, od = (o + 6144u) & 16383u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) { // four independent groups of loads in flight per wave, 8 waves per SIMD
        if (MODE == 0 || MODE == 4 || MODE == 5 || MODE == 6) {
            f4 a, b, c, d;
            LD4(a, oa, 0); LD4(b, ob, 0); LD4(c, oc, 0); LD4(d, od, 0);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            acc += a.x + b.y + c.z + d.w;
        } else if (MODE == 1) {
            float a, b, c, d;
            LD1(a, oa, 0); LD1(b, ob, 0); LD1(c, oc, 0); LD1(d, od, 0);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            acc += a + b + c + d;
        } else if (MODE == 2) {
            f3 a, b, c, d;
            LD3(a, oa, 0); LD3(b, ob, 0); LD3(c, oc, 0); LD3(d, od, 0);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            acc += a.x + b.y + c.z + d.x;
        } else if (MODE == 3) { // x[k], y[k], z[k] of a 48-byte block: three dwords 16 bytes apart
            float a0, a1, a2, b0, b1, b2, c0, c1, c2, d0, d1, d2;
            LD1(a0, oa, 4); LD1(a1, oa, 20); LD1(a2, oa, 36); LD1(b0, ob, 4); LD1(b1, ob, 20); LD1(b2, ob, 36);
            LD1(c0, oc, 4); LD1(c1, oc, 20); LD1(c2, oc, 36); LD1(d0, od, 4); LD1(d1, od, 20); LD1(d2, od, 36);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(d0), "+v"(d1), "+v"(d2));
            acc += a0 + a1 + a2 + b0 + b1 + b2 + c0 + c1 + c2 + d0 + d1 + d2;
        } else { // the three 16-byte rows of a 48-byte block
            f4 a0, a1, a2, b0, b1, b2, c0, c1, c2, d0, d1, d2;
            LD4(a0, oa, 0); LD4(a1, oa, 16); LD4(a2, oa, 32); LD4(b0, ob, 0); LD4(b1, ob, 16); LD4(b2, ob, 32);
            LD4(c0, oc, 0); LD4(c1, oc, 16); LD4(c2, oc, 32); LD4(d0, od, 0); LD4(d1, od, 16); LD4(d2, od, 32);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(d0), "+v"(d1), "+v"(d2));
            acc += a0.x + a1.y + a2.z + b0.x + b1.y + b2.z + c0.x + c1.y + c2.z + d0.x + d1.y + d2.z;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// ---- second part: RANDOM addresses (no two lanes on one bank by construction of a stride), the patterns of the grid kernel --------------
//   R0 three dwordx4 rows of a random 48-byte block   R1 dwordx3 at a random dword (the column offsets)   R2 three dwords 16 bytes apart
//   of a random block (the winner's x[k], y[k], z[k])   R3 one dword   R4 one dwordx4 (16-byte aligned)   R5 one dwordx2
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_rand(const float* __restrict__ base, const unsigned* __restrict__ offs, int iters, float* out) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    unsigned oa = offs[(t * 4u) & 0xFFFFu], ob = offs[(t * 4u + 1u) & 0xFFFFu], oc = offs[(t * 4u + 2u) & 0xFFFFu], od = offs[(t * 4u + 3u) & 0xFFFFu];
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            f4 a0, a1, a2, b0, b1, b2, c0, c1, c2, d0, d1, d2;
            LD4(a0, oa, 0); LD4(a1, oa, 16); LD4(a2, oa, 32); LD4(b0, ob, 0); LD4(b1, ob, 16); LD4(b2, ob, 32);
            LD4(c0, oc, 0); LD4(c1, oc, 16); LD4(c2, oc, 32); LD4(d0, od, 0); LD4(d1, od, 16); LD4(d2, od, 32);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(d0), "+v"(d1), "+v"(d2));
            acc += a0.x + a1.y + a2.z + b0.x + b1.y + b2.z + c0.x + c1.y + c2.z + d0.x + d1.y + d2.z;
        } else if (MODE == 1) {
            f3 a, b, c, d;
            LD3(a, oa, 0); LD3(b, ob, 0); LD3(c, oc, 0); LD3(d, od, 0);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            acc += a.x + b.y + c.z + d.x;
        } else if (MODE == 2) {
            float a0, a1, a2, b0, b1, b2, c0, c1, c2, d0, d1, d2;
            LD1(a0, oa, 4); LD1(a1, oa, 20); LD1(a2, oa, 36); LD1(b0, ob, 4); LD1(b1, ob, 20); LD1(b2, ob, 36);
            LD1(c0, oc, 4); LD1(c1, oc, 20); LD1(c2, oc, 36); LD1(d0, od, 4); LD1(d1, od, 20); LD1(d2, od, 36);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(d0), "+v"(d1), "+v"(d2));
            acc += a0 + a1 + a2 + b0 + b1 + b2 + c0 + c1 + c2 + d0 + d1 + d2;
        } else if (MODE == 3) {
            float a, b, c, d;
            LD1(a, oa, 0); LD1(b, ob, 0); LD1(c, oc, 0); LD1(d, od, 0);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            acc += a + b + c + d;
        } else if (MODE == 4) {
            f4 a, b, c, d;
            LD4(a, oa, 0); LD4(b, ob, 0); LD4(c, oc, 0); LD4(d, od, 0);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            acc += a.x + b.y + c.z + d.w;
        } else {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 a, b, c, d;
            asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(a) : "v"(oa), "s"(base));
            asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(b) : "v"(ob), "s"(base));
            asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(c) : "v"(oc), "s"(base));
            asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(d) : "v"(od), "s"(base));
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            acc += a.x + b.y + c.x + d.y;
        }
    }
    out[t] = acc;
}

static int run_random(float* dout) {
    const int blocks = 256 * 8, iters = 1000;
    const char* names[] = {"R0 3 x dwordx4 rows of a random 48 B block", "R1 dwordx3 at a random dword", "R2 3 x dword of a random block",
                           "R3 dword, random", "R4 dwordx4, random (16 B aligned)", "R5 dwordx2, random (8 B aligned)"};
    const int loads_per_iter[] = {12, 4, 12, 4, 4, 4};
    const size_t windows[] = {16u << 10, 1u << 20, 64u << 20}; // L1-resident, L2-resident, beyond the L2s (Infinity Cache)
    float* d;
    CHK(hipMalloc(&d, windows[2] + 4096)); CHK(hipMemset(d, 0, windows[2] + 4096));
    unsigned* doffs; CHK(hipMalloc(&doffs, 65536 * 4));
    for (size_t w : windows) {
        for (int mode = 0; mode < 6; ++mode) {
            std::vector<unsigned> offs(65536);
            unsigned long long x = 88172645463325252ull;
            for (unsigned i = 0; i < 65536; ++i) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                const unsigned r = (unsigned)(x >> 20);
                if (mode == 0 || mode == 2) offs[i] = (r % (unsigned)(w / 48)) * 48u;
                else if (mode == 4) offs[i] = (r % (unsigned)(w / 16)) * 16u;
                else if (mode == 5) offs[i] = (r % (unsigned)(w / 8)) * 8u;
                else offs[i] = (r % (unsigned)(w / 4)) * 4u;
            }
            CHK(hipMemcpy(doffs, offs.data(), 65536 * 4, hipMemcpyHostToDevice));
            auto launch = [&]() {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k_rand<0>, dim3(blocks), dim3(256), 0, 0, d, doffs, iters, dout); break;
                    case 1: hipLaunchKernelGGL(k_rand<1>, dim3(blocks), dim3(256), 0, 0, d, doffs, iters, dout); break;
                    case 2: hipLaunchKernelGGL(k_rand<2>, dim3(blocks), dim3(256), 0, 0, d, doffs, iters, dout); break;
                    case 3: hipLaunchKernelGGL(k_rand<3>, dim3(blocks), dim3(256), 0, 0, d, doffs, iters, dout); break;
                    case 4: hipLaunchKernelGGL(k_rand<4>, dim3(blocks), dim3(256), 0, 0, d, doffs, iters, dout); break;
                    default: hipLaunchKernelGGL(k_rand<5>, dim3(blocks), dim3(256), 0, 0, d, doffs, iters, dout); break;
                }
            };
            hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
            launch(); CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0)); launch(); CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            const double wave_instr = (double)blocks * 4 * iters * loads_per_iter[mode];
            printf("window %6zu KB  %-44s %8.3f ms   %6.2f CU-cycles per wave load instruction (2.4 GHz)\n", w >> 10, names[mode], ms,
                   ms * 1e-3 * 2.4e9 / (wave_instr / 256.0));
        }
    }
    return 0;
}

int main() {
    const size_t nfl = 4096 + 4096;
    std::vector<float> h(nfl, 1.f);
    float* d; float* dout;
    CHK(hipMalloc(&d, nfl * 4)); CHK(hipMemcpy(d, h.data(), nfl * 4, hipMemcpyHostToDevice));
    const int blocks = 256 * 8, iters = 2000;
    CHK(hipMalloc(&dout, (size_t)blocks * 256 * 4));
    const char* names[] = {"dwordx4, own 128 B line per lane", "dword, own line per lane", "dwordx3, own line per lane", "3 x dword, one line per lane",
                           "dwordx4, one line per quad", "dwordx4, all lanes one line", "dwordx4, two lanes per 128 B line", "3 x dwordx4 rows of a 48 B block per lane"};
    const int loads_per_iter[] = {4, 4, 4, 12, 4, 4, 4, 12};
    for (int mode = 0; mode < 8; ++mode) {
        auto launch = [&]() {
            switch (mode) {
                case 0: hipLaunchKernelGGL(k_l1<0>, dim3(blocks), dim3(256), 0, 0, d, iters, dout); break;
                case 1: hipLaunchKernelGGL(k_l1<1>, dim3(blocks), dim3(256), 0, 0, d, iters, dout); break;
                case 2: hipLaunchKernelGGL(k_l1<2>, dim3(blocks), dim3(256), 0, 0, d, iters, dout); break;
                case 3: hipLaunchKernelGGL(k_l1<3>, dim3(blocks), dim3(256), 0, 0, d, iters, dout); break;
                case 4: hipLaunchKernelGGL(k_l1<4>, dim3(blocks), dim3(256), 0, 0, d, iters, dout); break;
                case 5: hipLaunchKernelGGL(k_l1<5>, dim3(blocks), dim3(256), 0, 0, d, iters, dout); break;
                case 6: hipLaunchKernelGGL(k_l1<6>, dim3(blocks), dim3(256), 0, 0, d, iters, dout); break;
                default: hipLaunchKernelGGL(k_l1<7>, dim3(blocks), dim3(256), 0, 0, d, iters, dout); break;
            }
        };
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        launch(); CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0)); launch(); CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        const double wave_instr = (double)blocks * 4 * iters * loads_per_iter[mode];
        const double per_cu_cycles = ms * 1e-3 * 2.4e9;
        printf("mode %d  %-44s %8.3f ms   %6.2f CU-cycles per wave load instruction (2.4 GHz)   wave_instr %.4g\n", mode, names[mode], ms,
               per_cu_cycles / (wave_instr / 256.0), wave_instr);
    }
    if (getenv("L1_PROBE_FIXED_ONLY")) return 0;
    return run_random(dout);
}
