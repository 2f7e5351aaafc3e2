// hostsim.h -- minimal SIMT simulator for the CPU test-suite (CDBG_HOSTSIM builds only).
//
// TEST INFRASTRUCTURE.  Lets tests/ run the *same kernel source* that hipcc compiles
// for gfx950, inside a container without a GPU: each workgroup's threads are
// cooperative fibers (ucontext) on one OS thread, __syncthreads() and the 64-lane
// wave intrinsics (__ballot/__shfl*) are rendezvous points, atomics are plain
// read-modify-writes.  It finds logic errors, out-of-bounds and divergent-barrier
// bugs; it does NOT model memory races, so GPU parity tests (-m gpu) remain the gate.
// Workgroups run one after another; `static` stands in for __shared__.
#pragma once
#include <ucontext.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define CDBG_HD inline
#define CDBG_DEV inline
#define CDBG_SHARED static
#define CDBG_SPIN_YIELD() ::hostsim::spin_yield()
#define CDBG_WAVE_SYNC() ::hostsim::wave_rendezvous(0)
#define CDBG_PIN64(x) do { } while (0)
#define CDBG_LAUNCH(kern, grid, block, stream, ...) \
    (::hostsim::g_kernel_name = #kern, ::hostsim::launch((unsigned)(grid), (unsigned)(block), [=]() { kern(__VA_ARGS__); }))

struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace hostsim {

struct Idx { unsigned x = 0, y = 0, z = 0; };
inline Idx g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

enum { RUNNABLE = 0, AT_BLOCK = 1, AT_WAVE = 2, DONE = 3 };
inline const char* g_kernel_name = "?";
constexpr int kStack = 256 * 1024;
constexpr int kMaxThreads = 1024;

struct Block {
    int n = 0;
    ucontext_t sched;
    ucontext_t ctx[kMaxThreads];
    int state[kMaxThreads];
    uint64_t posted[kMaxThreads];
    uint64_t snap[kMaxThreads];
    uint64_t active[kMaxThreads / 64];      // lanes that took part in the last wave rendezvous
    int cur = -1;
    const std::function<void()>* body = nullptr;
    char* stacks = nullptr;
};
inline Block g_blk;

inline void trampoline() {
    (*g_blk.body)();
    g_blk.state[g_blk.cur] = DONE;
    swapcontext(&g_blk.ctx[g_blk.cur], &g_blk.sched);
}
inline void to_sched() { swapcontext(&g_blk.ctx[g_blk.cur], &g_blk.sched); }
inline void spin_yield() { to_sched(); }
inline void block_barrier() { g_blk.state[g_blk.cur] = AT_BLOCK; to_sched(); }
inline void wave_rendezvous(uint64_t v) {
    g_blk.posted[g_blk.cur] = v;
    g_blk.state[g_blk.cur] = AT_WAVE;
    to_sched();
}
inline int lane() { return g_blk.cur & 63; }
inline int wave_base() { return g_blk.cur & ~63; }

inline void run_block(unsigned bx, unsigned nthreads, const std::function<void()>& body) {
    Block& B = g_blk;
    if (!B.stacks) B.stacks = (char*)malloc((size_t)kStack * kMaxThreads);
    if (nthreads > (unsigned)kMaxThreads) { fprintf(stderr, "hostsim: block too large\n"); abort(); }
    B.n = (int)nthreads; B.body = &body;
    g_blockIdx.x = bx;
    for (int i = 0; i < B.n; ++i) {
        getcontext(&B.ctx[i]);
        B.ctx[i].uc_stack.ss_sp = B.stacks + (size_t)i * kStack;
        B.ctx[i].uc_stack.ss_size = kStack;
        B.ctx[i].uc_link = &B.sched;
        makecontext(&B.ctx[i], (void (*)())trampoline, 0);
        B.state[i] = RUNNABLE;
    }
    long idle_passes = 0;
    for (;;) {
        bool ran = false; int done = 0;
        for (int i = 0; i < B.n; ++i) {
            if (B.state[i] == DONE) { ++done; continue; }
            if (B.state[i] != RUNNABLE) continue;
            B.cur = i; g_threadIdx.x = (unsigned)i;
            swapcontext(&B.sched, &B.ctx[i]);
            ran = true;
            if (B.state[i] == DONE) ++done;
        }
        if (done == B.n) break;
        bool released = false;
        // block barrier: every live thread waits at it
        {
            int at = 0, live = 0;
            for (int i = 0; i < B.n; ++i) { if (B.state[i] != DONE) { ++live; if (B.state[i] == AT_BLOCK) ++at; } }
            if (live && at == live) { for (int i = 0; i < B.n; ++i) if (B.state[i] == AT_BLOCK) B.state[i] = RUNNABLE; released = true; }
        }
        // wave rendezvous: every live lane of the wave waits at it
        for (int w = 0; w * 64 < B.n; ++w) {
            int lo = w * 64, hi = lo + 64 < B.n ? lo + 64 : B.n, at = 0, live = 0;
            for (int i = lo; i < hi; ++i) { if (B.state[i] != DONE) { ++live; if (B.state[i] == AT_WAVE) ++at; } }
            if (live && at == live) {
                uint64_t act = 0;
                for (int i = lo; i < hi; ++i) {
                    if (B.state[i] == AT_WAVE) { B.snap[i] = B.posted[i]; act |= 1ull << (i - lo); B.state[i] = RUNNABLE; }
                    else B.snap[i] = 0;
                }
                B.active[w] = act; released = true;
            }
        }
        bool any_runnable = false;
        for (int i = 0; i < B.n; ++i) if (B.state[i] == RUNNABLE) any_runnable = true;
        if (!any_runnable && !released) {
            fprintf(stderr, "hostsim: DEADLOCK in %s block %u (divergent __syncthreads / wave intrinsic?) states:", g_kernel_name, bx);
            for (int i = 0; i < B.n && i < 64; ++i) fprintf(stderr, " %d", B.state[i]);
            fprintf(stderr, "\n"); abort();
        }
        if (!released && ran) { if (++idle_passes > 50000000L) { fprintf(stderr, "hostsim: livelock (spin loop never satisfied)\n"); abort(); } }
        else idle_passes = 0;
    }
}

template <class F>
inline void launch(unsigned grid, unsigned block, F f) {
    std::function<void()> body = f;
    g_gridDim.x = grid; g_blockDim.x = block;
    for (unsigned b = 0; b < grid; ++b) run_block(b, block, body);
}

template <class T> inline uint64_t to_bits(T v) { uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }

}  // namespace hostsim

#define threadIdx ::hostsim::g_threadIdx
#define blockIdx ::hostsim::g_blockIdx
#define blockDim ::hostsim::g_blockDim
#define gridDim ::hostsim::g_gridDim

inline void __syncthreads() { ::hostsim::block_barrier(); }
inline void __threadfence() {}
inline void __threadfence_block() {}

inline unsigned long long __ballot(int pred) {
    ::hostsim::wave_rendezvous(pred ? 1 : 0);
    unsigned long long m = 0; int b = ::hostsim::wave_base();
    for (int l = 0; l < 64 && b + l < ::hostsim::g_blk.n; ++l) if (::hostsim::g_blk.snap[b + l]) m |= 1ull << l;
    return m;
}
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int __all(int pred) {
    unsigned long long m = __ballot(pred);
    return m == ::hostsim::g_blk.active[::hostsim::wave_base() / 64];
}
template <class T> inline T __shfl(T v, int src, int width = 64) {
    ::hostsim::wave_rendezvous(::hostsim::to_bits(v));
    int l = ::hostsim::lane(); int base = l & ~(width - 1);
    int s = base + (src & (width - 1));
    return ::hostsim::from_bits<T>(::hostsim::g_blk.snap[::hostsim::wave_base() + s]);
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
    ::hostsim::wave_rendezvous(::hostsim::to_bits(v));
    int l = ::hostsim::lane(); int base = l & ~(width - 1);
    int s = l - (int)d; if (s < base) s = l;
    return ::hostsim::from_bits<T>(::hostsim::g_blk.snap[::hostsim::wave_base() + s]);
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
    ::hostsim::wave_rendezvous(::hostsim::to_bits(v));
    int l = ::hostsim::lane(); int base = l & ~(width - 1);
    int s = l + (int)d; if (s >= base + width) s = l;
    return ::hostsim::from_bits<T>(::hostsim::g_blk.snap[::hostsim::wave_base() + s]);
}
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) {
    ::hostsim::wave_rendezvous(::hostsim::to_bits(v));
    int l = ::hostsim::lane(); int s = l ^ mask; (void)width;
    return ::hostsim::from_bits<T>(::hostsim::g_blk.snap[::hostsim::wave_base() + s]);
}

inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }

template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = (T)(o + v); return o; }
template <class T> inline T atomicSub(T* p, T v) { T o = *p; *p = (T)(o - v); return o; }
template <class T> inline T atomicOr(T* p, T v) { T o = *p; *p = (T)(o | v); return o; }
template <class T> inline T atomicAnd(T* p, T v) { T o = *p; *p = (T)(o & v); return o; }
template <class T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }

// ---- the sliver of the HIP runtime API the host orchestration uses ----
typedef int hipError_t;
typedef int hipStream_t;
struct hostsim_event { double t; };
typedef hostsim_event* hipEvent_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hostsim error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t = 0) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = 0) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = 0; return hipSuccess; }
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = 1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hostsim_event{0}; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = 0) {
    e->t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)8 << 30; *t = (size_t)8 << 30; return hipSuccess; }
