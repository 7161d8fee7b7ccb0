// Developer probe (MI355X): how the vector-memory front end prices 16-byte gathers that hit L1 --
// distinct lines per lane, the same line for all lanes, groups of lanes sharing a line, and lanes masked off by exec.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/ta_probe.hip -o /tmp/ta_probe && /tmp/ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// every lane does `iters` rounds of three dependent-free 16-byte loads at base + off[lane] (+16, +32), like one grid block
template <int MODE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ base, const unsigned* __restrict__ offs, int iters, float* out) {
    const unsigned lane = threadIdx.x & 63u;
    unsigned o = offs[(blockIdx.x * 256 + threadIdx.x) & 0xFFFFu];
    float acc = 0.f;
    const bool on = (MODE != 3) || (lane & 1u) == 0u;          // mode 3: odd lanes masked off
    const bool on4 = (MODE != 4) || (lane & 3u) == 0u;          // mode 4: one lane in four
    const bool on16 = (MODE != 5) || (lane < 32u);              // mode 5: upper half of the wave off
    for (int it = 0; it < iters; ++it) {
        if (on && on4 && on16) {
            const float4 a = base[o], b = base[o + 1], c = base[o + 2];
            acc += a.x + b.y + c.z;
            o = (o + 3u * 64u + (__float_as_uint(a.w) & 1u)) & 0x3FFFu; // stays inside a 256 KB window
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const size_t n = 1u << 14; // float4 elements = 256 KB window
    std::vector<float> h(n * 4, 0.f);
    float4* d; unsigned* doffs; float* dout;
    CHK(hipMalloc(&d, n * 16 + 4096)); CHK(hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice));
    CHK(hipMalloc(&doffs, 65536 * 4)); CHK(hipMalloc(&dout, 256 * 2048 * 4 * 8));
    const int blocks = 256 * 8, iters = 2000;
    const char* names[] = {"distinct line per lane (48 B apart x 3 -> own line)", "all lanes the same line", "groups of 4 lanes share a line",
                           "distinct, odd lanes exec-masked", "distinct, 1 lane in 4 active", "distinct, lanes 32..63 masked", "groups of 16 lanes share a line"};
    for (int mode = 0; mode < 7; ++mode) {
        std::vector<unsigned> offs(65536);
        for (unsigned i = 0; i < 65536; ++i) {
            const unsigned lane = i & 63u;
            unsigned v;
            if (mode == 1) v = 0;
            else if (mode == 2) v = (lane >> 2) * 8u + (i >> 6) % 7u * 8u;    // 128-byte line = 8 float4
            else if (mode == 6) v = (lane >> 4) * 8u;
            else v = lane * 8u * 3u + (i >> 6) % 5u * 8u;                     // every lane its own lines
            offs[i] = v & 0x3FFFu;
        }
        CHK(hipMemcpy(doffs, offs.data(), 65536 * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        auto launch = [&]() {
            switch (mode) {
                case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, doffs, iters, dout); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, doffs, iters, dout); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, d, doffs, iters, dout); break;
                default: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, doffs, iters, dout); break;
            }
        };
        launch(); CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0)); launch(); CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        const double wave_instr = (double)blocks * 4 * iters * 3; // 16-byte load instructions
        const double per_cu_cycles = ms * 1e-3 * 2.4e9;
        printf("mode %d  %-58s %8.3f ms   %6.1f CU-cycles per wave load instruction\n", mode, names[mode], ms, per_cu_cycles / (wave_instr / 256.0));
    }
    return 0;
}
