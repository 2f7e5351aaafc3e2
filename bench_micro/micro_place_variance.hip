// micro_place_variance.hip -- dev tool: the rate of scattered 16-byte stores into a 82 GB region is bimodal between PROCESSES
// (34 ms <-> 47 ms for 2^30 stores, alternating from one process to the next on the same box: profiles/r04_micro_place_variance.log).
// Which way of obtaining the region avoids the slow mode?   micro_place_variance <variant>
//   0  one hipMalloc (what libcdbg does)
//   1  a 200 GB hipMalloc + hipFree first, then as 0
//   2  virtual-memory API: one address range, 1 GB physical chunks mapped in ascending order
//   3  as 2, chunks of 64 MB
//   4  two hipMallocs of the region, the faster one (by a 2^26-store probe) is kept
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31); }
constexpr int LOGP = 22;
__global__ void k(uint4* dst, uint64_t n, uint64_t cap) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t p = mix64(i) >> (64 - LOGP);
        uint4 v; v.x = (uint32_t)i; v.y = 1; v.z = 2; v.w = 3;
        dst[p * cap * 4 + (i >> LOGP)] = v;                    // (region slots of 64 bytes, as micro_sector_store: 82 GB)
    }
}
static float run(uint4* buf, uint64_t n, uint64_t cap) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); float ms = 0;
    for (int r = 0; r < 2; ++r) { (void)hipEventRecord(a); hipLaunchKernelGGL(k, dim3(256 * 8), dim3(256), 0, 0, buf, n, cap); (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b); }
    return ms;
}
static int vmm_region(void** out, uint64_t bytes, uint64_t chunk) {
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (chunk % gran) chunk = (chunk / gran + 1) * gran;
    const uint64_t n = (bytes + chunk - 1) / chunk;
    hipDeviceptr_t base; CK(hipMemAddressReserve(&base, n * chunk, 0, 0, 0));
    for (uint64_t i = 0; i < n; ++i) {
        hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap((hipDeviceptr_t)((char*)base + i * chunk), chunk, 0, h, 0));
    }
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(base, n * chunk, &acc, 1));
    *out = (void*)base; return 0;
}
int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const uint64_t N = 1ull << 30, cap = (N >> LOGP) + (N >> (LOGP + 2));
    const uint64_t bytes = ((uint64_t)cap << LOGP) * 64 + 4096;
    void* buf = nullptr;
    if (variant == 1) { void* d = nullptr; CK(hipMalloc(&d, 200ull << 30)); CK(hipMemset(d, 0, 1 << 20)); CK(hipFree(d)); }
    if (variant == 2) { if (vmm_region(&buf, bytes, 1ull << 30)) return 1; }
    else if (variant == 3) { if (vmm_region(&buf, bytes, 64ull << 20)) return 1; }
    else if (variant == 4) {
        void* a = nullptr; void* b = nullptr; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        const float ta = run((uint4*)a, 1ull << 26, cap), tb = run((uint4*)b, 1ull << 26, cap);
        printf("probe %.2f %.2f ms; ", ta, tb);
        if (ta <= tb) { buf = a; CK(hipFree(b)); } else { buf = b; CK(hipFree(a)); }
    } else CK(hipMalloc(&buf, bytes));
    const float ms = run((uint4*)buf, N, cap);
    printf("variant %d: %.2f ms  %.1f G records/s\n", variant, ms, N / ms / 1e6);
    return 0;
}
