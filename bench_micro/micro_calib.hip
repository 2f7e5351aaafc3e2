// micro_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE per ACCESS PATTERN (dev tool).
// MI355X_MICROARCH.md (HBM): FETCH_SIZE = TCC_EA0_RDREQ x 64 B reports half the bytes of a wide coalesced read; "other access
// widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".  The pipeline's kernels
// are not streaming kernels: the scan issues far atomics + scattered 16-byte stores, the glue random 8-byte gathers.  Each
// kernel below performs a KNOWN number of accesses of one pattern over a footprint far beyond L2 + Infinity Cache (16 GiB),
// so counter bytes / access gives the factor to apply to that pattern:  bench_micro/calib.sh runs it under two --pmc passes.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31); }
constexpr uint64_t FOOT = 16ull << 30;                    // bytes
__global__ void calib_stream_read16(const uint4* src, uint64_t n, uint64_t* sink) {     // n 16-byte loads, coalesced
    uint64_t acc = 0; const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x1234567) *sink = acc;
}
__global__ void calib_stream_write16(uint4* dst, uint64_t n) {                           // n 16-byte stores, coalesced
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { uint4 v; v.x = (uint32_t)i; v.y = 1; v.z = 2; v.w = 3; dst[i] = v; }
}
__global__ void calib_gather8(const uint64_t* src, uint64_t n, uint64_t* sink) {         // n random 8-byte loads
    uint64_t acc = 0; const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += src[mix64(i) % (FOOT / 8)];
    if (acc == 0x1234567) *sink = acc;
}
__global__ void calib_scatter16(uint4* dst, uint64_t n) {                                // n random 16-byte stores
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { uint4 v; v.x = (uint32_t)i; v.y = 1; v.z = 2; v.w = 3; dst[mix64(i) % (FOOT / 16)] = v; }
}
__global__ void calib_scatter8(uint64_t* dst, uint64_t n) {                              // n random 8-byte stores
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[mix64(i) % (FOOT / 8)] = i;
}
__global__ void calib_atomic32_far(uint32_t* ctr, uint64_t n, uint64_t foot_words) {     // n device atomics on random words
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) atomicAdd(&ctr[mix64(i) % foot_words], 1u);
}
__global__ void calib_atomic32_ret(uint32_t* ctr, uint64_t n, uint64_t foot_words, uint64_t* sink) {   // returning atomics (the scan's reservation)
    uint64_t acc = 0; const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += atomicAdd(&ctr[mix64(i) % foot_words], 1u);
    if (acc == 0x1234567) *sink = acc;
}
int main() {
    void* buf = nullptr; uint64_t* sink = nullptr;
    if (hipMalloc(&buf, FOOT) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, FOOT);
    const uint64_t N = 1ull << 30;                        // accesses per kernel
    const int grid = 256 * 16, block = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float ms;
#define RUN(name, useful, ...) do { hipEventRecord(a); hipLaunchKernelGGL(name, dim3(grid), dim3(block), 0, 0, __VA_ARGS__); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); \
        printf("%-24s accesses %llu useful_bytes %llu ms %.3f  (%.1f G accesses/s, %.1f GB/s useful)\n", #name, (unsigned long long)N, (unsigned long long)(useful), ms, N / ms / 1e6, (useful) / ms / 1e6); } while (0)
    RUN(calib_stream_read16, N * 16, (const uint4*)buf, N, sink);
    RUN(calib_stream_write16, N * 16, (uint4*)buf, N);
    RUN(calib_gather8, N * 8, (const uint64_t*)buf, N, sink);
    RUN(calib_scatter16, N * 16, (uint4*)buf, N);
    RUN(calib_scatter8, N * 8, (uint64_t*)buf, N);
    RUN(calib_atomic32_far, N * 4, (uint32_t*)buf, N, FOOT / 4);
    RUN(calib_atomic32_ret, N * 4, (uint32_t*)buf, N, (16ull << 20) / 4, sink);          // 16 MB of counters: the scan's part_fill array
    RUN(calib_atomic32_far, N * 4, (uint32_t*)buf, N, (16ull << 20) / 4);
    return 0;
}
