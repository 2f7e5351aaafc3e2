// micro_part_store.hip -- dev tool: rate of 16-byte record stores into P partition regions that fill sequentially (the
// scan's pattern) as a function of P: do partial-sector stores merge in L2 / Infinity Cache when the ACTIVE lines fit?
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31); }
__global__ void part_store(uint4* dst, uint64_t n, int logp, uint64_t cap) {            // slot = time / P: no atomics
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t p = mix64(i) >> (64 - logp);
        uint4 v; v.x = (uint32_t)i; v.y = 1; v.z = 2; v.w = 3;
        dst[p * cap + (i >> logp)] = v;
    }
}
__global__ void part_store_atomic(uint4* dst, uint32_t* fill, uint64_t n, int logp, uint64_t cap) {   // slot from a returning atomic
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t p = mix64(i) >> (64 - logp);
        const uint32_t j = atomicAdd(&fill[p], 1u);
        uint4 v; v.x = (uint32_t)i; v.y = 1; v.z = 2; v.w = 3;
        if (j < cap) dst[p * cap + j] = v;
    }
}
// XCD-private regions: workgroup b runs on XCD b % 8; partition p of XCD x has its own region -> a line is filled by one L2
__global__ void part_store_xcd(uint4* dst, uint64_t n, int logp, uint64_t cap) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t x = blockIdx.x & 7u;
    uint64_t t = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride, ++t) {
        const uint64_t p = mix64(i) >> (64 - logp);
        uint4 v; v.x = (uint32_t)i; v.y = 1; v.z = 2; v.w = 3;
        dst[(p * 8 + x) * (cap / 8 + 1) + ((i >> logp) >> 3)] = v;
    }
}
int main() {
    const uint64_t N = 1ull << 30;
    void* buf = nullptr; uint32_t* fill = nullptr;
    const uint64_t bytes = (N + (N >> 2)) * 16 + (64ull << 20) * 16;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&fill, (1u << 23) * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float ms;
    const int grid = 256 * 8, block = 256;
    for (int logp = 12; logp <= 22; logp += 2) {
        const uint64_t cap = (N >> logp) + (N >> (logp + 2));
        hipEventRecord(a); hipLaunchKernelGGL(part_store, dim3(grid), dim3(block), 0, 0, (uint4*)buf, N, logp, cap); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("P=2^%d  store only          %.2f ms  %.1f G/s\n", logp, ms, N / ms / 1e6);
        hipMemset(fill, 0, (1u << 23) * 4);
        hipEventRecord(a); hipLaunchKernelGGL(part_store_atomic, dim3(grid), dim3(block), 0, 0, (uint4*)buf, fill, N, logp, cap); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("P=2^%d  atomic + store      %.2f ms  %.1f G/s\n", logp, ms, N / ms / 1e6);
        hipEventRecord(a); hipLaunchKernelGGL(part_store_xcd, dim3(grid), dim3(block), 0, 0, (uint4*)buf, N, logp, cap); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("P=2^%d  store, XCD-private  %.2f ms  %.1f G/s\n", logp, ms, N / ms / 1e6);
    }
    return 0;
}
