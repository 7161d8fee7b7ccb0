// Developer probe (MI355X): calibrates rocprofv3's FETCH_SIZE for the access pattern of the grid kernel -- three 16-byte loads of one
// 48-byte block per lane at a pseudo-random block index -- against a byte count that is KNOWN from the address stream:
// every block touches one or two 128-byte lines (or 64-byte half lines), counted on the host with the same index function.
// Kernels (each launched once, named so that the counter CSV can be matched):
//   k_stream_16B        coalesced 16 B / lane streaming read (the guide's calibrated case: FETCH_SIZE = bytes / 2)
//   k_gather_48B<F>     random 48-byte blocks out of a footprint of F MB: 64 (inside the 256 MB Infinity Cache, > 8 x 4 MB L2),
//                       1024 and 4096 (far beyond it)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/gather_probe.hip -o tools/probes/gather_probe
//   tools/probes/run_gather_probe.sh   (rocprofv3 --pmc FETCH_SIZE, then TCC_EA0_RDREQ_sum / TCC_MISS_sum; prints the ratios)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <unordered_set>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__host__ __device__ inline uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
struct __attribute__((aligned(16))) Blk { float x[4], y[4], z[4]; };

__global__ __launch_bounds__(256) void k_stream_16B(const float4* __restrict__ p, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int FOOT_MB>
__global__ __launch_bounds__(256) void k_gather_48B(const Blk* __restrict__ b, uint32_t n_blk, int rounds, float* out) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        const uint32_t j = mix(t * 131u + (uint32_t)r * 2654435761u) % n_blk;
        const Blk B = b[j];
        acc += B.x[0] + B.y[1] + B.z[2];
    }
    out[t] = acc;
}

int main() {
    const size_t foot_mb[3] = {64, 1024, 4096};
    const size_t max_bytes = foot_mb[2] << 20;
    void* d = nullptr;
    float* dout = nullptr;
    CHK(hipMalloc(&d, max_bytes));
    CHK(hipMemset(d, 0, max_bytes));
    const int blocks = 256 * 16, rounds = 16;
    const bool host_count = getenv("GP_SKIP_HOST") == nullptr; // the counter passes only need the launches
    CHK(hipMalloc(&dout, (size_t)blocks * 256 * 4));
    CHK(hipDeviceSynchronize());
    {
        const size_t n = (size_t)2048 << 20 >> 4; // 2 GB of float4
        hipLaunchKernelGGL(k_stream_16B, dim3(blocks), dim3(256), 0, 0, (const float4*)d, n, dout);
        CHK(hipDeviceSynchronize());
        printf("k_stream_16B requested_bytes %zu unique_128B_line_bytes %zu\n", n * 16, n * 16);
    }
    for (int f = 0; f < 3; ++f) {
        const uint32_t n_blk = (uint32_t)((foot_mb[f] << 20) / sizeof(Blk));
        if (f == 0) hipLaunchKernelGGL(k_gather_48B<64>, dim3(blocks), dim3(256), 0, 0, (const Blk*)d, n_blk, rounds, dout);
        if (f == 1) hipLaunchKernelGGL(k_gather_48B<1024>, dim3(blocks), dim3(256), 0, 0, (const Blk*)d, n_blk, rounds, dout);
        if (f == 2) hipLaunchKernelGGL(k_gather_48B<4096>, dim3(blocks), dim3(256), 0, 0, (const Blk*)d, n_blk, rounds, dout);
        CHK(hipDeviceSynchronize());
        if (!host_count) continue;
        // the same address stream on the host: line visits (every load instruction of a wave re-requests its lines) and unique lines
        size_t visits128 = 0, visits64 = 0;
        std::unordered_set<uint64_t> u128, u64;
        const size_t threads = (size_t)blocks * 256;
        u128.reserve(threads * rounds * 2);
        u64.reserve(threads * rounds * 2);
        for (size_t t = 0; t < threads; ++t)
            for (int r = 0; r < rounds; ++r) {
                const uint64_t a = (uint64_t)(mix((uint32_t)t * 131u + (uint32_t)r * 2654435761u) % n_blk) * sizeof(Blk);
                for (uint64_t l = a / 128; l <= (a + 47) / 128; ++l) { ++visits128; u128.insert(l); }
                for (uint64_t l = a / 64; l <= (a + 47) / 64; ++l) { ++visits64; u64.insert(l); }
            }
        printf("k_gather_48B<%zu> requested_bytes %zu line128_visit_bytes %zu unique_128B_line_bytes %zu line64_visit_bytes %zu unique_64B_bytes %zu\n",
               foot_mb[f], threads * rounds * sizeof(Blk), visits128 * 128, u128.size() * 128, visits64 * 64, u64.size() * 64);
    }
    return 0;
}
