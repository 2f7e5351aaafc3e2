// micro_sector_store.hip -- dev tool: does a FULL 32-byte sector per scattered record store run faster than
// the 16-byte half sector the scan writes today, alone and behind the reservation atomic?  2^30 records into 2^22
// sequentially filling regions (the scan's pattern at config 3), four records in flight per lane.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31); }
constexpr int LOGP = 22, U = 4;
template <int BYTES, bool ATOMIC>
__global__ void k(uint4* dst, uint32_t* fill, uint64_t n, uint64_t cap) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * U;
    for (uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * U; i0 < n; i0 += stride) {
        uint64_t p[U]; uint32_t j[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { p[u] = mix64(i0 + u) >> (64 - LOGP); j[u] = ATOMIC ? atomicAdd(&fill[p[u]], 1u) : (uint32_t)((i0 + u) >> LOGP); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (j[u] >= cap) continue;
            uint4 v; v.x = (uint32_t)i0; v.y = 1; v.z = 2; v.w = j[u];
            uint4* q = dst + (p[u] * cap + j[u]) * (BYTES / 16);
            q[0] = v;
            if (BYTES >= 32) q[1] = v;
            if (BYTES >= 64) { q[2] = v; q[3] = v; }
        }
    }
}
template <int BYTES, bool ATOMIC>
void run(const char* what, uint4* buf, uint32_t* fill, uint64_t N, uint64_t cap) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float ms;
    for (int r = 0; r < 2; ++r) {
        hipMemset(fill, 0, (1u << LOGP) * 4);
        hipEventRecord(a); hipLaunchKernelGGL((k<BYTES, ATOMIC>), dim3(256 * 8), dim3(256), 0, 0, buf, fill, N, cap); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    }
    printf("%-34s %7.2f ms  %5.1f G records/s  %6.0f GB/s\n", what, ms, N / ms / 1e6, N * (double)BYTES / ms / 1e6);
}
int main() {
    const uint64_t N = 1ull << 30;
    const uint64_t cap = (N >> LOGP) + (N >> (LOGP + 2));
    void* buf = nullptr; uint32_t* fill = nullptr;
    const uint64_t bytes = ((uint64_t)cap << LOGP) * 64 + 4096;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&fill, (1u << LOGP) * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    run<16, false>("16-byte records, no atomic", (uint4*)buf, fill, N, cap);
    run<32, false>("32-byte records, no atomic", (uint4*)buf, fill, N, cap);
    run<64, false>("64-byte records, no atomic", (uint4*)buf, fill, N, cap);
    run<16, true>("16-byte records, returning atomic", (uint4*)buf, fill, N, cap);
    run<32, true>("32-byte records, returning atomic", (uint4*)buf, fill, N, cap);
    run<64, true>("64-byte records, returning atomic", (uint4*)buf, fill, N, cap);
    return 0;
}
