// dev microbenchmark (MI355X): throughput of random returning 32-bit atomic adds on a counter array in HBM,
// by memory scope (agent = what atomicAdd() emits; workgroup = performed in the XCD's own L2) and by whether the
// counters an XCD touches are private to it.  Question behind it: the scan pays one device atomic per record on
// 4 M partition cursors (~33 ms of its 68 ms) -- would XCD-private cursors with narrow-scope atomics be cheaper?
//   hipcc --offload-arch=gfx950 -O3 bench_micro/micro_atomic_scope.hip -o bench_micro/micro_atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xF; }

// SCOPE 0 agent, 1 workgroup; PRIV 0: any counter, 1: counters of this XCD's 1/8 slice (by HW XCC id), 2: slice by blockIdx % 8
template <int SCOPE, int PRIV, bool RTN>
__global__ void __launch_bounds__(256) k_atom(uint32_t* ctr, uint32_t n_mask, uint32_t iters, uint32_t* sink, uint32_t* xcc_seen) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t x = PRIV == 1 ? xcc_id() : (blockIdx.x & 7u);
    if (threadIdx.x == 0 && xcc_seen) atomicOr(&xcc_seen[blockIdx.x & 7u], 1u << xcc_id());
    uint32_t acc = 0, h = gid * 2654435761u;
    for (uint32_t i = 0; i < iters; ++i) {
        h = mix(h + i);
        uint32_t idx = h & n_mask;
        if (PRIV) idx = (idx & (n_mask >> 3)) | (x * ((n_mask + 1) >> 3));
        uint32_t r;
        if (SCOPE == 0) r = __hip_atomic_fetch_add(&ctr[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else r = __hip_atomic_fetch_add(&ctr[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (RTN) acc += r;
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;
}
__global__ void k_sum(const uint32_t* ctr, uint32_t n, unsigned long long* out) {
    unsigned long long s = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) s += ctr[i];
    atomicAdd(out, s);
}
int main() {
    const uint32_t GRID = 256 * 8, IT = 2048;
    uint32_t *ctr, *sink, *seen; unsigned long long* tot;
    CK(hipMalloc(&ctr, 64u << 20)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&seen, 32)); CK(hipMalloc(&tot, 8));
    CK(hipMemset(seen, 0, 32));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double total = (double)GRID * 256 * IT;
#define RUN(SC, PR, RT, NCTR, label) do { \
    CK(hipMemset(ctr, 0, 64u << 20)); CK(hipMemset(tot, 0, 8)); \
    k_atom<SC, PR, RT><<<GRID, 256>>>(ctr, (NCTR) - 1, 8, sink, seen); CK(hipMemset(ctr, 0, 64u << 20)); CK(hipDeviceSynchronize()); \
    CK(hipEventRecord(e0)); k_atom<SC, PR, RT><<<GRID, 256>>>(ctr, (NCTR) - 1, IT, sink, nullptr); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
    k_sum<<<1024, 256>>>(ctr, (NCTR), tot); unsigned long long t; CK(hipMemcpy(&t, tot, 8, hipMemcpyDeviceToHost)); \
    printf("%-58s %2u M counters: %7.1f G atomics/s   sum %s\n", label, (unsigned)((NCTR) >> 20), total / (ms * 1e-3) / 1e9, t == (unsigned long long)total ? "exact" : "LOST UPDATES"); } while (0)
    for (uint32_t n : { 1u << 20, 4u << 20, 16u << 20 }) {
        RUN(0, 0, true, n, "agent scope, any counter, returning");
        RUN(0, 0, false, n, "agent scope, any counter, no return");
        RUN(0, 1, true, n, "agent scope, XCD-private slice (XCC_ID), returning");
        RUN(1, 1, true, n, "workgroup scope, XCD-private slice (XCC_ID), returning");
        RUN(1, 2, true, n, "workgroup scope, slice by blockIdx % 8, returning");
        RUN(1, 1, false, n, "workgroup scope, XCD-private slice (XCC_ID), no return");
        RUN(1, 0, true, n, "workgroup scope, ANY counter (expected to lose updates)");
    }
    uint32_t h[8]; CK(hipMemcpy(h, seen, 32, hipMemcpyDeviceToHost));
    printf("XCC ids seen per blockIdx %% 8:"); for (int i = 0; i < 8; ++i) printf(" %x", h[i]); printf("\n");
    return 0;
}
