// ab_overlap_place.h -- dev tool (VERDICT r5 "next" #1, step 0): does a pure record-placement kernel (one returning 32-bit atomic + two
// 8-byte stores per record, records and partition ids read from a coalesced stream; no LDS, < 32 VGPRs) overlap with k_count_fast<1>?
// Lab note.  It was compiled into a VARIANT library only (-DCDBG_AB_OVERLAP): host_count.h included this file behind that macro and called
// ab_overlap_before_count() / ab_overlap_after_count() around the launch of the one-pass count tier (capped layout).  The hooks left the product header
// when deferred placement was built from the result (profiles/r06_ab_overlap_place_vs_count.log); to repeat the measurement put the two calls back.
// Even steps: the placement kernel alone, then the count alone.  Odd steps: both at once on two streams.  One line per step on stderr.
#pragma once

namespace {

__global__ void __launch_bounds__(256) k_ab_place_init(uint4* src, uint32_t* lp, uint64_t n, int log_np) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t x = i + 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; x ^= x >> 31;
        uint4 v; v.x = (uint32_t)i; v.y = (uint32_t)(i >> 32); v.z = (uint32_t)x; v.w = 7u; src[i] = v;
        lp[i] = (uint32_t)(x >> (64 - log_np));
    }
}
// U records per lane and iteration: U atomics in flight before the first store
template <int U>
__global__ void __launch_bounds__(256) k_ab_place(const uint4* __restrict__ src, const uint32_t* __restrict__ lp, uint64_t n, uint32_t* fill, uint64_t* region, uint32_t cap) {
    const uint64_t per = (uint64_t)blockDim.x * U, stride = (uint64_t)gridDim.x * per;
    for (uint64_t b = (uint64_t)blockIdx.x * per; b < n; b += stride) {
        uint4 v[U]; uint32_t p[U], j[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const uint64_t i = b + (uint64_t)u * blockDim.x + threadIdx.x; p[u] = 0xFFFFFFFFu; if (i < n) { v[u] = src[i]; p[u] = lp[i]; } }
#pragma unroll
        for (int u = 0; u < U; ++u) if (p[u] != 0xFFFFFFFFu) j[u] = atomicAdd(&fill[p[u]], 1u);
#pragma unroll
        for (int u = 0; u < U; ++u) if (p[u] != 0xFFFFFFFFu && j[u] < cap) {
            // TWO 8-byte stores, as the scan issues them (k_scan.h scan_finish_record: one 16-byte store per lane behind a returning atomic runs at half the rate;
            // left alone the compiler merges the pair into global_store_dwordx4 -- the first run of this A/B measured that: 0.8 G records in 53 ms)
            volatile uint64_t* d = region + ((uint64_t)p[u] * cap + j[u]) * 2;
            d[1] = ((uint64_t)v[u].w << 32) | v[u].z; d[0] = ((uint64_t)v[u].y << 32) | v[u].x;
        }
    }
}

// Where do the placement workgroups land?  One-wave workgroups that register on their CU (XCC_ID | SE / SH / CU of HW_ID) and -- quota > 0 -- leave
// when the CU has its share already; the survivors take batches of 64 x U x 4 records from one cursor, so any number of them finishes the job.
template <int U>
__global__ void __launch_bounds__(64) k_ab_place_q(const uint4* __restrict__ src, const uint32_t* __restrict__ lp, uint64_t n, uint32_t* fill, uint64_t* region, uint32_t cap,
                                                   uint32_t* cu_count, uint32_t quota, unsigned long long* cursor, uint32_t prio) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const uint32_t key = ((xcc & 15u) << 8) | ((hw >> 8) & 0xFFu);
    uint32_t r = 0;
    if (threadIdx.x == 0) r = atomicAdd(&cu_count[key], 1u);
    r = __builtin_amdgcn_readfirstlane(r);
    if (quota && r >= quota) return;
    if (prio) __builtin_amdgcn_s_setprio(3);
    constexpr uint64_t BATCH = 64ull * U * 4;
    for (;;) {
        unsigned long long b0 = 0;
        if (threadIdx.x == 0) b0 = atomicAdd(cursor, 1ull);
        b0 = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(b0 >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)b0);
        if (b0 * BATCH >= n) return;
        for (int q = 0; q < 4; ++q) {
            const uint64_t b = b0 * BATCH + (uint64_t)q * 64 * U;
            uint4 v[U]; uint32_t p[U], j[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const uint64_t i = b + (uint64_t)u * 64 + threadIdx.x; p[u] = 0xFFFFFFFFu; if (i < n) { v[u] = src[i]; p[u] = lp[i]; } }
#pragma unroll
            for (int u = 0; u < U; ++u) if (p[u] != 0xFFFFFFFFu) j[u] = atomicAdd(&fill[p[u]], 1u);
#pragma unroll
            for (int u = 0; u < U; ++u) if (p[u] != 0xFFFFFFFFu && j[u] < cap) {
                volatile uint64_t* d = region + ((uint64_t)p[u] * cap + j[u]) * 2;
                d[1] = ((uint64_t)v[u].w << 32) | v[u].z; d[0] = ((uint64_t)v[u].y << 32) | v[u].x;
            }
        }
    }
}

struct AbOverlap {
    DBuf<uint32_t> cu_count; DBuf<unsigned long long> cursor; int quota = -1, prio = 0;
    DBuf<uint4> src; DBuf<uint32_t> lp, fill; DBuf<uint64_t> region;
    hipStream_t sb{}; hipEvent_t e0{}, e1{}, ea{}, c0{}, c1{};
    uint64_t n = 0; uint32_t cap = 0; uint64_t npl = 0; int step = 0; int grid = 0, unroll = 4;
    bool on = false, both = false;
    float ms_alone = 0;
};
inline AbOverlap& ab_overlap() { static AbOverlap a; return a; }

inline void ab_place_launch(AbOverlap& a, hipStream_t s) {
    if (a.quota >= 0) {                                      // one-wave workgroups with a per-CU quota (0: no quota, placement only logged)
        (void)hipMemsetAsync(a.cu_count.p, 0, 4096 * sizeof(uint32_t), s); (void)hipMemsetAsync(a.cursor.p, 0, sizeof(unsigned long long), s);
        if (a.unroll == 1) CDBG_LAUNCH((k_ab_place_q<1>), a.grid, 64, s, (const uint4*)a.src.p, (const uint32_t*)a.lp.p, a.n, a.fill.p, a.region.p, a.cap, a.cu_count.p, (uint32_t)a.quota, a.cursor.p, (uint32_t)a.prio);
        else CDBG_LAUNCH((k_ab_place_q<4>), a.grid, 64, s, (const uint4*)a.src.p, (const uint32_t*)a.lp.p, a.n, a.fill.p, a.region.p, a.cap, a.cu_count.p, (uint32_t)a.quota, a.cursor.p, (uint32_t)a.prio);
        return;
    }
    if (a.unroll == 1) CDBG_LAUNCH((k_ab_place<1>), a.grid, 256, s, (const uint4*)a.src.p, (const uint32_t*)a.lp.p, a.n, a.fill.p, a.region.p, a.cap);
    else if (a.unroll == 2) CDBG_LAUNCH((k_ab_place<2>), a.grid, 256, s, (const uint4*)a.src.p, (const uint32_t*)a.lp.p, a.n, a.fill.p, a.region.p, a.cap);
    else CDBG_LAUNCH((k_ab_place<4>), a.grid, 256, s, (const uint4*)a.src.p, (const uint32_t*)a.lp.p, a.n, a.fill.p, a.region.p, a.cap);
}
// before the one-pass count kernel is launched on stream s
inline int ab_overlap_before_count(cdbg_ctx* c, hipStream_t s, uint64_t NPL, uint32_t part_cap, int log_npl) {
    AbOverlap& a = ab_overlap();
    const char* e = getenv("CDBG_AB_OVERLAP");
    a.on = e != nullptr && NPL > 1024;
    if (!a.on) return CDBG_OK;
    if (!a.sb) {
        a.n = strtoull(e, nullptr, 10); a.cap = part_cap; a.npl = NPL;
        if (const char* g = getenv("CDBG_AB_PLACE_GRID")) a.grid = atoi(g); else a.grid = 256 * 8;
        if (const char* u = getenv("CDBG_AB_PLACE_UNROLL")) a.unroll = atoi(u);
        if (const char* q = getenv("CDBG_AB_PLACE_QUOTA")) a.quota = atoi(q);
        if (const char* q = getenv("CDBG_AB_PLACE_PRIO")) a.prio = atoi(q);
        CK(a.cu_count.alloc(4096, true)); CK(a.cursor.alloc(1, true));
        HIPCK(hipStreamCreateWithFlags(&a.sb, hipStreamNonBlocking));
        HIPCK(hipEventCreate(&a.e0)); HIPCK(hipEventCreate(&a.e1)); HIPCK(hipEventCreate(&a.ea)); HIPCK(hipEventCreate(&a.c0)); HIPCK(hipEventCreate(&a.c1));
        CK(a.src.alloc(a.n, false)); CK(a.lp.alloc(a.n, false)); CK(a.fill.alloc(NPL, true)); CK(a.region.alloc(NPL * (uint64_t)part_cap * 2, false));
        CDBG_LAUNCH(k_ab_place_init, 256 * 16, 256, s, a.src.p, a.lp.p, a.n, log_npl);
        HIPCK(hipMemsetAsync(a.region.p, 0, NPL * (uint64_t)part_cap * 16, s));       // (touch the region once: first-touch pages are not what is measured)
        HIPCK(hipStreamSynchronize(s));
    }
    a.both = (a.step++ & 1) != 0;
    HIPCK(hipMemsetAsync(a.fill.p, 0, a.npl * sizeof(uint32_t), s));
    if (!a.both) {                                           // alone, on the count's own stream, before it
        HIPCK(hipEventRecord(a.e0, s)); ab_place_launch(a, s); HIPCK(hipEventRecord(a.e1, s));
    } else {                                                 // at once: stream B starts where the count starts
        HIPCK(hipEventRecord(a.ea, s)); HIPCK(hipStreamWaitEvent(a.sb, a.ea, 0));
        if (!getenv("CDBG_AB_COUNT_FIRST")) { HIPCK(hipEventRecord(a.e0, a.sb)); ab_place_launch(a, a.sb); HIPCK(hipEventRecord(a.e1, a.sb)); }
    }
    HIPCK(hipEventRecord(a.c0, s));
    return CDBG_OK;
}
// after the one-pass count kernel was launched on stream s (the caller synchronises s next)
inline int ab_overlap_after_count(cdbg_ctx*, hipStream_t s) {
    AbOverlap& a = ab_overlap();
    if (!a.on) return CDBG_OK;
    if (a.both && getenv("CDBG_AB_COUNT_FIRST")) { HIPCK(hipEventRecord(a.e0, a.sb)); ab_place_launch(a, a.sb); HIPCK(hipEventRecord(a.e1, a.sb)); }   // (the count's workgroups are on the chip first)
    HIPCK(hipEventRecord(a.c1, s));
    HIPCK(hipEventSynchronize(a.c1)); HIPCK(hipEventSynchronize(a.e1));
    char where[160] = "";
    if (a.quota >= 0) {                                      // CUs that took placement waves, the fullest one, waves that stayed
        std::vector<uint32_t> cc(4096); HIPCK(hipMemcpy(cc.data(), a.cu_count.p, 4096 * sizeof(uint32_t), hipMemcpyDeviceToHost));
        uint32_t cus = 0, mx = 0, stay = 0, xccs = 0;
        for (int x = 0; x < 16; ++x) { bool any = false; for (int i = 0; i < 256; ++i) { const uint32_t v = cc[x * 256 + i]; if (v) { ++cus; any = true; mx = v > mx ? v : mx; stay += a.quota ? (v < (uint32_t)a.quota ? v : (uint32_t)a.quota) : v; } } xccs += any; }
        snprintf(where, sizeof where, " | waves landed on %u CUs of %u XCDs, at most %u on one, %u stayed (quota %d, prio %d)", cus, xccs, mx, stay, a.quota, a.prio);
    }
    float ms_place = 0, ms_count = 0, ms_pair = 0;
    HIPCK(hipEventElapsedTime(&ms_place, a.e0, a.e1)); HIPCK(hipEventElapsedTime(&ms_count, a.c0, a.c1));
    if (a.both) {
        // wall of the pair: from the common start to the later of the two ends
        float x = 0, y = 0; HIPCK(hipEventElapsedTime(&x, a.ea, a.e1)); HIPCK(hipEventElapsedTime(&y, a.ea, a.c1)); ms_pair = x > y ? x : y;
        fprintf(stderr, "[ab-overlap] AT ONCE: place %.2f ms (%.3f G records, %d WGs, %d per lane) | k_count_fast %.2f ms | pair %.2f ms%s\n", ms_place, (double)a.n / 1e9, a.grid, a.unroll, ms_count, ms_pair, where);
    } else fprintf(stderr, "[ab-overlap] ALONE  : place %.2f ms (%.3f G records, %d WGs, %d per lane) | k_count_fast %.2f ms | sum  %.2f ms%s\n", ms_place, (double)a.n / 1e9, a.grid, a.unroll, ms_count, ms_place + ms_count, where);
    return CDBG_OK;
}

}  // namespace
