// devrt.h -- the one place that knows whether we compile for the GPU (hipcc, gfx950)
// or for the kernel-logic simulator used by the CPU test-suite (tests/hostsim).
//
// PRODUCT BUILD: hipcc --offload-arch=gfx950.  There is no CPU fallback in the
// product: libcdbg.so refuses to create a context without a HIP device.
//
// CDBG_HOSTSIM build: compiled by tests/hostsim/build.sh with g++.  Every kernel
// body is the same source; threads of a workgroup run as cooperative fibers
// (tests/hostsim/hostsim.h, found through the include path of tests/hostsim/build.sh).  It exists so that `pytest -m "not gpu"` can exercise the real
// kernel logic (record packing, junction rules, chain walks, glue, ranking) in a
// container without a GPU.  It is test infrastructure and is never loaded by
// bcalm_amd/.
#pragma once
#include <stdint.h>

#ifdef CDBG_HOSTSIM
#include "hostsim.h"
#else
#include <hip/hip_runtime.h>
#define CDBG_HD __host__ __device__ __forceinline__
#define CDBG_DEV __device__ __forceinline__
#define CDBG_SPIN_YIELD() __builtin_amdgcn_s_sleep(1)
#define CDBG_LAUNCH(kern, grid, block, stream, ...) \
    hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), 0, stream, __VA_ARGS__)
#define CDBG_SHARED __shared__
// wave-level rendezvous between an LDS write and reads of it by other lanes of the SAME wave:
// the hardware executes a wave's LDS instructions in order, so only the compiler needs a barrier
#define CDBG_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
// Pins a 64-bit value in registers as an opaque SSA value.  A select chain over the elements of a small register
// array ("element idx without a runtime index") is otherwise folded back by the optimiser into a load with a
// selected ADDRESS, which forces the whole array into scratch memory (seen in the ISA of the two- and four-word
// kernels as scratch_store/scratch_load in the inner loops).
#define CDBG_PIN64(x) asm volatile("" : "+v"(x))
#endif

namespace cdbg {

// uint64_t is `unsigned long` on LP64 while the HIP atomics are declared for
// `unsigned long long`; same width, so cast once here.
CDBG_DEV uint64_t atomic_add_u64(uint64_t* p, uint64_t v) {
    return (uint64_t)atomicAdd((unsigned long long*)p, (unsigned long long)v);
}
CDBG_DEV uint64_t atomic_cas_u64(uint64_t* p, uint64_t cmp, uint64_t v) {
    return (uint64_t)atomicCAS((unsigned long long*)p, (unsigned long long)cmp, (unsigned long long)v);
}
CDBG_DEV uint64_t atomic_exch_u64(uint64_t* p, uint64_t v) {
    return (uint64_t)atomicExch((unsigned long long*)p, (unsigned long long)v);
}
// the compiler may not move memory operations across this point (the hardware executes a wave's LDS operations in order)
#define CDBG_COMPILER_BARRIER() asm volatile("" ::: "memory")
// L1-bypassing (agent-scope) load / volatile LDS flag read
CDBG_DEV uint64_t ld_agent_u64(const uint64_t* p) {
#ifdef CDBG_HOSTSIM
    return *p;
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
CDBG_DEV uint32_t ld_volatile_u32(const uint32_t* p) { return *(const volatile uint32_t*)p; }
CDBG_DEV uint32_t atomic_add_u32(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
CDBG_DEV uint32_t atomic_sub_u32(uint32_t* p, uint32_t v) { return atomicSub(p, v); }
CDBG_DEV uint32_t atomic_or_u32(uint32_t* p, uint32_t v) { return atomicOr(p, v); }
CDBG_DEV uint32_t atomic_cas_u32(uint32_t* p, uint32_t cmp, uint32_t v) { return atomicCAS(p, cmp, v); }
CDBG_DEV uint32_t atomic_max_u32(uint32_t* p, uint32_t v) { return atomicMax(p, v); }
CDBG_DEV uint32_t atomic_min_u32(uint32_t* p, uint32_t v) { return atomicMin(p, v); }
CDBG_DEV void atomic_or_u64(uint64_t* p, uint64_t v) { (void)atomicOr((unsigned long long*)p, (unsigned long long)v); }
// unaligned global accesses (gfx950 global_load/store take any byte address; one request instead of 8 / 4 / 2)
CDBG_DEV uint64_t ld_unaligned_u64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
CDBG_DEV void st_unaligned_u64(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
CDBG_DEV void st_unaligned_u32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
CDBG_DEV void st_unaligned_u16(uint8_t* p, uint16_t v) { __builtin_memcpy(p, &v, 2); }

// ---- wave64 primitives that do not go through the LDS crossbar ----
// (measured on MI355X, bench_micro/micro_r02: one ds_bpermute costs ~19 SIMD cycles per wave, a DPP-modified
// VALU op ~3; the wave scans / broadcasts of the per-partition kernels therefore use DPP and v_readlane)
#ifdef CDBG_HOSTSIM
CDBG_DEV uint32_t wave_incl_sum_u32(uint32_t v) {
    const int lane = (int)(threadIdx.x & 63);
    for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(v, (unsigned)d); if (lane >= d) v += x; }
    return v;
}
CDBG_DEV uint32_t wave_readlane_u32(uint32_t v, int src_lane) { return __shfl(v, src_lane); }   // src_lane wave-uniform
CDBG_DEV uint32_t alignbit_u32(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31u)); }
#else
CDBG_DEV uint32_t wave_incl_sum_u32(uint32_t v) {
    // row_shr:1,2,4,8 inside rows of 16 lanes, then row_bcast:15 / row_bcast:31 across rows; lanes without a source add 0
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
CDBG_DEV uint32_t wave_readlane_u32(uint32_t v, int src_lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src_lane); }
CDBG_DEV uint32_t alignbit_u32(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
#endif
// value known to be identical in every lane -> scalar register (frees VGPRs, makes the address math scalar)
#ifdef CDBG_HOSTSIM
CDBG_DEV uint32_t uni_u32(uint32_t v) { return __shfl(v, 0); }   // (called by all lanes in wave-uniform control flow only)
#define CDBG_NOINLINE
#define CDBG_DEV_NOINL inline
#define CDBG_LDS_BARRIER() __syncthreads()
#define CDBG_LDS_FENCE() do { } while (0)
#else
CDBG_DEV uint32_t uni_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
#define CDBG_NOINLINE __attribute__((noinline))
#define CDBG_DEV_NOINL __device__
// Workgroup barrier for threads that communicate through LDS ONLY.  __syncthreads() carries a workgroup-scope fence
// over every address space, which on gfx950 becomes s_waitcnt vmcnt(0): each barrier then drains the wave's global
// loads and stores -- prefetches stop being prefetches and a barrier behind a burst of stores waits for their
// acknowledgements.  The "local" fences order LDS traffic only (s_waitcnt lgkmcnt(0)); global accesses stay in flight.
#define CDBG_LDS_BARRIER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); \
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)
// Release fence for a slot-claim protocol that lives in LDS ONLY (write the key words, then publish the claim word).
// __threadfence_block() orders every address space: it waits for the wave's global loads too (vmcnt(0)) -- in the
// counting kernels that is the prefetch of the NEXT partition's records, issued a moment earlier.
#define CDBG_LDS_FENCE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")
#endif
CDBG_DEV uint64_t uni_u64(uint64_t v) { return ((uint64_t)uni_u32((uint32_t)(v >> 32)) << 32) | uni_u32((uint32_t)v); }
// One slot of an append-only list for every lane that calls this TOGETHER (divergent control flow allowed): the active lanes of the wave advance
// the cursor (an LDS word of their workgroup) with ONE atomic and take consecutive slots in lane order, so that their 16-byte stores behind it coalesce.
#ifdef CDBG_HOSTSIM
CDBG_DEV uint32_t wave_append_slots(uint32_t* cursor) { return atomic_add_u32(cursor, 1u); }   // (the simulator's wave rendezvous needs every live lane: one atomic per lane there)
#else
CDBG_DEV uint32_t wave_append_slots(uint32_t* cursor) {
    const uint64_t act = __ballot(1);                                                              // the lanes that are here (exec mask)
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
    uint32_t base = 0;
    if (rank == 0) base = atomic_add_u32(cursor, (uint32_t)__popcll(act));
    return uni_u32(base) + rank;                                                                   // (readfirstlane reads the first ACTIVE lane: the one of rank 0)
}
#endif
// v in the lanes whose bit of a wave-uniform 64-bit mask (from uni_u64) is set, 0 in the others.  A mask in a scalar
// register pair IS a lane predicate on CDNA: one v_cndmask, where (mask >> lane) & 1 costs a 64-bit shift per lane.
#ifdef CDBG_HOSTSIM
CDBG_DEV uint32_t lane_pick_u32(uint64_t mask, uint32_t v, int lane) { return ((mask >> lane) & 1ULL) ? v : 0u; }
#else
CDBG_DEV uint32_t lane_pick_u32(uint64_t mask, uint32_t v, int) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(mask));
    return r;
}
#endif
CDBG_DEV uint64_t wave_sum_u64(uint64_t v) {        // all lanes -> the wave total (kernel epilogues only)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

}  // namespace cdbg
