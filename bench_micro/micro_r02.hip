// micro_r02.hip -- round-2 design-input microbenchmarks (MI355X).  Not part of the product.
//   A  VALU issue cost of the integer ops the scan / count kernels are made of
//   B  LDS atomic throughput (ds_cmpst_rtn_b64, ds_add_u32, ...) at k_count's occupancy
//   C  HBM scatter of 16 B records vs 64 / 128 / 256 B lines (two-level record placement)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// ---------------- A: VALU ----------------
enum { OP_ADD32, OP_MULLO, OP_MULHI, OP_SHL64, OP_MAD64, OP_ADD64, OP_BFREV, OP_MIN, OP_XORSHIFT, OP_BPERM, OP_MUL64, OP_CMPSEL64, OP_PERM };
template <int OP>
__global__ void __launch_bounds__(256) k_valu(uint64_t iters, uint64_t* sink) {
    uint64_t a0 = threadIdx.x + 1, a1 = a0 * 3 + blockIdx.x, a2 = a0 * 5 + 7, a3 = a0 * 7 + 11;
    uint32_t s = (uint32_t)(threadIdx.x & 31) + 1;
    for (uint64_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == OP_ADD32) { a0 = (uint32_t)a0 + (uint32_t)a1; a1 = (uint32_t)a1 + (uint32_t)a2; a2 = (uint32_t)a2 + (uint32_t)a3; a3 = (uint32_t)a3 + (uint32_t)a0; }
            if (OP == OP_MULLO) { a0 = (uint32_t)a0 * (uint32_t)a1; a1 = (uint32_t)a1 * (uint32_t)a2; a2 = (uint32_t)a2 * (uint32_t)a3; a3 = (uint32_t)a3 * (uint32_t)a0; }
            if (OP == OP_MULHI) { a0 = __umulhi((uint32_t)a0, (uint32_t)a1); a1 = __umulhi((uint32_t)a1, (uint32_t)a2) | 3; a2 = __umulhi((uint32_t)a2, (uint32_t)a3) | 5; a3 = __umulhi((uint32_t)a3, (uint32_t)a0) | 9; }
            if (OP == OP_SHL64) { a0 = (a0 << s) | 1; a1 = (a1 >> s) | (1ull << 63); a2 = (a2 << s) | 1; a3 = (a3 >> s) | (1ull << 63); }
            if (OP == OP_MAD64) { a0 = (uint64_t)(uint32_t)a0 * (uint32_t)a1 + a2; a1 = (uint64_t)(uint32_t)a1 * (uint32_t)a2 + a3; a2 = (uint64_t)(uint32_t)a2 * (uint32_t)a3 + a0; a3 = (uint64_t)(uint32_t)a3 * (uint32_t)a0 + a1; }
            if (OP == OP_ADD64) { a0 += a1; a1 += a2; a2 += a3; a3 += a0; }
            if (OP == OP_BFREV) { a0 = __builtin_bitreverse32((uint32_t)a0) + 1; a1 = __builtin_bitreverse32((uint32_t)a1) + 1; a2 = __builtin_bitreverse32((uint32_t)a2) + 1; a3 = __builtin_bitreverse32((uint32_t)a3) + 1; }
            if (OP == OP_MIN) { a0 = min((uint32_t)a0, (uint32_t)a1) + 1; a1 = min((uint32_t)a1, (uint32_t)a2) + 1; a2 = min((uint32_t)a2, (uint32_t)a3) + 1; a3 = min((uint32_t)a3, (uint32_t)a0) + 1; }
            if (OP == OP_XORSHIFT) { uint32_t x = (uint32_t)a0; x ^= x << 13; x ^= x >> 17; x ^= x << 5; a0 = x; uint32_t y = (uint32_t)a1; y ^= y << 13; y ^= y >> 17; y ^= y << 5; a1 = y; }
            if (OP == OP_BPERM) { a0 = __shfl((uint32_t)a0, (int)(a1 & 63)) + 1; a1 = __shfl((uint32_t)a1, (int)(a0 & 63)) + 1; a2 = __shfl((uint32_t)a2, (int)(a3 & 63)) + 1; a3 = __shfl((uint32_t)a3, (int)(a2 & 63)) + 1; }
            if (OP == OP_MUL64) { a0 = a0 * 0x9E3779B97F4A7C15ULL + a1; a1 = a1 * 0xBF58476D1CE4E5B9ULL + a2; a2 = a2 * 0x94D049BB133111EBULL + a3; a3 = a3 * 0x9E3779B97F4A7C15ULL + a0; }
            if (OP == OP_CMPSEL64) { a0 = (a0 < a1 ? a0 : a1) + 1; a1 = (a1 < a2 ? a1 : a2) + 3; a2 = (a2 < a3 ? a2 : a3) + 5; a3 = (a3 < a0 ? a3 : a0) + 7; }
            if (OP == OP_PERM) { a0 = __builtin_amdgcn_perm((uint32_t)a0, (uint32_t)a1, 0x02010003u) + 1; a1 = __builtin_amdgcn_perm((uint32_t)a1, (uint32_t)a2, 0x02010003u) + 1; a2 = __builtin_amdgcn_perm((uint32_t)a2, (uint32_t)a3, 0x02010003u) + 1; a3 = __builtin_amdgcn_perm((uint32_t)a3, (uint32_t)a0, 0x02010003u) + 1; }
        }
    }
    if ((a0 ^ a1 ^ a2 ^ a3) == 0x123456789ull) *sink = a0;
}

// ---------------- B: LDS atomics ----------------
enum { L_CAS64, L_ADD32, L_CAS64_ADD32, L_ADD32_RTN, L_WRITE64, L_READ64, L_CAS32, L_ADD64_RTN, L_CAS64_SAME };
template <int OP, int SLOTS, int NT>
__global__ void __launch_bounds__(NT) k_lds(uint64_t iters, uint64_t* sink) {
    __shared__ unsigned long long keys[SLOTS]; __shared__ uint32_t cnt[SLOTS];
    for (int i = threadIdx.x; i < SLOTS; i += NT) { keys[i] = ~0ull; cnt[i] = 0; }
    __syncthreads();
    uint32_t x = (blockIdx.x * NT + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long acc = 0;
    for (uint64_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            const uint32_t s = x & (SLOTS - 1);
            if (OP == L_CAS64) acc += atomicCAS(&keys[s], ~0ull, (unsigned long long)s);
            if (OP == L_CAS64_SAME) acc += atomicCAS(&keys[s], 12345ull, (unsigned long long)s);   // never succeeds after init: pure compare
            if (OP == L_ADD32) atomicAdd(&cnt[s], 1u);
            if (OP == L_CAS64_ADD32) { acc += atomicCAS(&keys[s], ~0ull, (unsigned long long)s); atomicAdd(&cnt[s], 1u); }
            if (OP == L_ADD32_RTN) acc += atomicAdd(&cnt[s], 1u);
            if (OP == L_WRITE64) keys[s] = x;
            if (OP == L_READ64) acc += keys[s];
            if (OP == L_CAS32) acc += atomicCAS(&cnt[s], 0u, s);
            if (OP == L_ADD64_RTN) acc += atomicAdd(&keys[s], 1ull);
        }
    }
    __syncthreads();
    if (acc == 0x123456789ull || cnt[threadIdx.x & (SLOTS - 1)] == 0x7fffffff) *sink = acc;
}

// ---------------- C: HBM scatter by line size ----------------
// every wave-store writes 1 KB: 64 lanes x 16 B, in groups of LINE/16 lanes to the same random LINE-byte line
template <int LINE>
__global__ void __launch_bounds__(256) k_scatter(uint4* buf, uint64_t nlines_mask, uint64_t iters) {
    const uint32_t lane = threadIdx.x & 63;
    constexpr int LPL = LINE / 16;                       // lanes per line
    uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPL;
    uint64_t x = g * 0x9E3779B97F4A7C15ULL + 77;
    for (uint64_t i = 0; i < iters; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const uint64_t line = x & nlines_mask;
        buf[line * LPL + (lane % LPL)] = make_uint4((uint32_t)x, lane, 2, 3);
    }
}
// same, but with one device atomic per line first (reservation) -- what a flush costs
template <int LINE>
__global__ void __launch_bounds__(256) k_scatter_atomic(uint4* buf, uint32_t* ctr, uint64_t nlines_mask, uint64_t iters) {
    const uint32_t lane = threadIdx.x & 63;
    constexpr int LPL = LINE / 16;
    uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPL;
    uint64_t x = g * 0x9E3779B97F4A7C15ULL + 77;
    for (uint64_t i = 0; i < iters; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const uint64_t line = x & nlines_mask;
        uint32_t r = 0;
        if (lane % LPL == 0) r = atomicAdd(&ctr[line & 0xFFFFF], 1u);
        r = __shfl(r, (lane / LPL) * LPL);
        buf[line * LPL + (lane % LPL)] = make_uint4((uint32_t)x, r, 2, 3);
    }
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint64_t* sink; CK(hipMalloc(&sink, 8));
    float ms;
    const double SIMDS = 4.0 * p.multiProcessorCount, HZ = 2.4e9;
    {   // A: 8 waves per SIMD: 256 CUs x 8 WGs of 256 threads
        const int GRID = p.multiProcessorCount * 8; const uint64_t IT = 2000;
#define RUNA(name, OP, nops) do { k_valu<OP><<<GRID, 256>>>(10, sink); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); k_valu<OP><<<GRID, 256>>>(IT, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); \
    const double waveops = (double)GRID * 4 * IT * 8 * (nops); printf("VALU %-10s : %7.2f cycles per wave-op per SIMD (%.1f ms)\n", name, ms * 1e-3 * HZ * SIMDS / waveops, ms); } while (0)
        RUNA("add32", OP_ADD32, 4); RUNA("mul_lo", OP_MULLO, 4); RUNA("mul_hi", OP_MULHI, 4); RUNA("shl64(var)", OP_SHL64, 4);
        RUNA("mad_u64_u32", OP_MAD64, 4); RUNA("add64", OP_ADD64, 4); RUNA("bfrev", OP_BFREV, 4); RUNA("min_u32", OP_MIN, 4);
        RUNA("xorshift32(6ops)", OP_XORSHIFT, 2); RUNA("bpermute", OP_BPERM, 4); RUNA("mul64+add64", OP_MUL64, 4); RUNA("cmp+sel64", OP_CMPSEL64, 4); RUNA("v_perm", OP_PERM, 4);
    }
    {   // B: k_count geometry: 4096 slots (48 KB) -> 3 WGs x 512 threads per CU
        const uint64_t IT = 1024;
#define RUNB(name, OP, SLOTS, NT, wgs_per_cu) do { const int GRID = p.multiProcessorCount * (wgs_per_cu); k_lds<OP, SLOTS, NT><<<GRID, NT>>>(4, sink); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); k_lds<OP, SLOTS, NT><<<GRID, NT>>>(IT, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); \
    const double ops = (double)GRID * NT * IT * 4; printf("LDS %-16s slots %5d NT %4d wg/cu %d : %8.1f Gops/s  (%.2f lane-ops per clk per CU)\n", name, SLOTS, NT, wgs_per_cu, ops / (ms * 1e-3) / 1e9, ops / (ms * 1e-3) / HZ / p.multiProcessorCount); } while (0)
        RUNB("xorshift-only", L_WRITE64 + 100, 4096, 512, 3);
        RUNB("cas64_rtn", L_CAS64, 4096, 512, 3);
        RUNB("cas64_rtn(cmp)", L_CAS64_SAME, 4096, 512, 3);
        RUNB("add32", L_ADD32, 4096, 512, 3);
        RUNB("cas64+add32", L_CAS64_ADD32, 4096, 512, 3);
        RUNB("add32_rtn", L_ADD32_RTN, 4096, 512, 3);
        RUNB("cas32_rtn", L_CAS32, 4096, 512, 3);
        RUNB("add64_rtn", L_ADD64_RTN, 4096, 512, 3);
        RUNB("write64", L_WRITE64, 4096, 512, 3);
        RUNB("read64", L_READ64, 4096, 512, 3);
        RUNB("cas64+add32", L_CAS64_ADD32, 4096, 256, 3);
        RUNB("cas64+add32", L_CAS64_ADD32, 1024, 256, 8);
        RUNB("cas64+add32", L_CAS64_ADD32, 512, 64, 16);
        RUNB("cas64_rtn", L_CAS64, 1024, 256, 8);
    }
    {   // C
        const size_t sz = 32ull << 30; uint4* buf; CK(hipMalloc(&buf, sz)); CK(hipMemset(buf, 0, sz));
        uint32_t* ctr; CK(hipMalloc(&ctr, 4u << 20)); CK(hipMemset(ctr, 0, 4u << 20));
        const int GRID = p.multiProcessorCount * 8; const uint64_t IT = 512;
#define RUNC(LINE) do { k_scatter<LINE><<<GRID, 256>>>(buf, sz / LINE - 1, 4); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); k_scatter<LINE><<<GRID, 256>>>(buf, sz / LINE - 1, IT); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); \
    const double bytes = (double)GRID * 256 * IT * 16; printf("scatter %4d-B lines over 32 GB        : %8.1f GB/s useful (%.1f G lines/s)\n", LINE, bytes / (ms * 1e-3) / 1e9, bytes / LINE / (ms * 1e-3) / 1e9); \
    k_scatter_atomic<LINE><<<GRID, 256>>>(buf, ctr, sz / LINE - 1, 4); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); k_scatter_atomic<LINE><<<GRID, 256>>>(buf, ctr, sz / LINE - 1, IT); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); \
    printf("scatter %4d-B lines + 1 atomic per line : %8.1f GB/s useful\n", LINE, bytes / (ms * 1e-3) / 1e9); } while (0)
        RUNC(16); RUNC(32); RUNC(64); RUNC(128); RUNC(256); RUNC(1024);
        CK(hipFree(buf));
    }
    return 0;
}
