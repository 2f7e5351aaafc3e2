// micro_atomics.hip -- design-input microbenchmarks (MI355X): random device atomics,
// scattered 16-B stores, LDS 64-bit CAS, streaming copy.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31);
}
__global__ void k_atomic_add32(uint32_t* t, uint64_t mask, uint64_t n_per_thread) {
    uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    for (uint64_t i = 0; i < n_per_thread; ++i) atomicAdd(&t[mix(g * n_per_thread + i) & mask], 1u);
}
__global__ void k_atomic_add64_ret(unsigned long long* t, uint64_t mask, uint64_t n_per_thread, unsigned long long* sink) {
    uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; unsigned long long acc = 0;
    for (uint64_t i = 0; i < n_per_thread; ++i) acc += atomicAdd(&t[mix(g * n_per_thread + i) & mask], 1ull);
    if (acc == 0x1234567) *sink = acc;
}
__global__ void k_atomic_cas64(unsigned long long* t, uint64_t mask, uint64_t n_per_thread, unsigned long long* sink) {
    uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; unsigned long long acc = 0;
    for (uint64_t i = 0; i < n_per_thread; ++i) { uint64_t h = mix(g * n_per_thread + i); acc += atomicCAS(&t[h & mask], 0ull, h | 1); }
    if (acc == 0x1234567) *sink = acc;
}
__global__ void k_scatter16(uint4* t, uint64_t mask, uint64_t n_per_thread) {
    uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    for (uint64_t i = 0; i < n_per_thread; ++i) { uint64_t h = mix(g * n_per_thread + i); t[h & mask] = make_uint4((uint32_t)h, 1, 2, 3); }
}
__global__ void k_gather8(const uint64_t* t, uint64_t mask, uint64_t n_per_thread, unsigned long long* sink) {
    uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; unsigned long long acc = 0;
    for (uint64_t i = 0; i < n_per_thread; ++i) acc += t[mix(g * n_per_thread + i) & mask];
    if (acc == 0x1234567) *sink = acc;
}
__global__ void k_copy(const uint4* a, uint4* b, uint64_t n) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, s = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += s) b[i] = a[i];
}
template <int SLOTS>
__global__ void k_lds_cas(uint64_t n_per_thread, unsigned long long* sink) {
    __shared__ unsigned long long keys[SLOTS]; __shared__ uint32_t cnt[SLOTS];
    for (int i = threadIdx.x; i < SLOTS; i += blockDim.x) { keys[i] = ~0ull; cnt[i] = 0; }
    __syncthreads();
    uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    for (uint64_t i = 0; i < n_per_thread; ++i) {
        uint64_t key = mix((g * n_per_thread + i) % (SLOTS / 4 * 3 / 2)) >> 2;   // ~37% load, many repeats
        uint32_t s = (uint32_t)mix(key) & (SLOTS - 1);
        for (;;) {
            unsigned long long old = atomicCAS(&keys[s], ~0ull, (unsigned long long)key);
            if (old == ~0ull || old == key) { atomicAdd(&cnt[s], 1u); break; }
            s = (s + 1) & (SLOTS - 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && cnt[0] == 0x7fffffff) *sink = cnt[1];
}
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d mem=%.1f GB\n", p.name, p.multiProcessorCount, p.totalGlobalMem / 1e9);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long* sink; CK(hipMalloc(&sink, 8));
    const int BLK = 256, GRID = 256 * 8; const uint64_t NPT = 256; const double NOPS = (double)BLK * GRID * NPT;
    size_t sizes[] = { 1ull << 20, 64ull << 20, 1ull << 30, 16ull << 30, 64ull << 30 };
    for (size_t sz : sizes) {
        void* buf; if (hipMalloc(&buf, sz) != hipSuccess) { printf("alloc %zu failed\n", sz); continue; }
        CK(hipMemset(buf, 0, sz));
        float ms;
#define RUN(name, launch) do { launch; CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) { launch; } CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); \
    printf("%-18s table %8.1f MB : %8.2f Gops/s\n", name, sz / 1048576.0, 3 * NOPS / (ms * 1e-3) / 1e9); } while (0)
        RUN("atomicAdd32", (k_atomic_add32<<<GRID, BLK>>>((uint32_t*)buf, sz / 4 - 1, NPT)));
        RUN("atomicAdd64_ret", (k_atomic_add64_ret<<<GRID, BLK>>>((unsigned long long*)buf, sz / 8 - 1, NPT, sink)));
        CK(hipMemset(buf, 0, sz));
        RUN("atomicCAS64", (k_atomic_cas64<<<GRID, BLK>>>((unsigned long long*)buf, sz / 8 - 1, NPT, sink)));
        RUN("scatter16B", (k_scatter16<<<GRID, BLK>>>((uint4*)buf, sz / 16 - 1, NPT)));
        RUN("gather8B", (k_gather8<<<GRID, BLK>>>((const uint64_t*)buf, sz / 8 - 1, NPT, sink)));
        CK(hipFree(buf));
    }
    {   size_t sz = 4ull << 30; void *a, *b; CK(hipMalloc(&a, sz)); CK(hipMalloc(&b, sz)); CK(hipMemset(a, 1, sz));
        float ms; k_copy<<<2048, 256>>>((uint4*)a, (uint4*)b, sz / 16); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) k_copy<<<2048, 256>>>((uint4*)a, (uint4*)b, sz / 16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy 4GB: %.2f TB/s (read+write)\n", 5 * 2.0 * sz / (ms * 1e-3) / 1e12); }
    {   float ms; const uint64_t N2 = 4096; 
        k_lds_cas<4096><<<GRID, BLK>>>(N2, sink); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); k_lds_cas<4096><<<GRID, BLK>>>(N2, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("LDS cas64+add (4096 slots, 256 thr/WG): %.2f Ginserts/s\n", (double)BLK * GRID * N2 / (ms * 1e-3) / 1e9); }
    return 0;
}
