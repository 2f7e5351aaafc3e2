// comm.h -- multi-GPU transport of libcdbg (include/cdbg.h, "Multi-GPU").
//
// The reference has no counterpart (one shared-memory call, /root/reference/src/bcalm_1.cpp:57); this is the
// exchange layer of SURVEY.md section 8(e): RCCL over xGMI, used directly from the library.  xGMI is point to
// point (a dedicated link per GPU pair), so the all-to-all-v and the variable-size all-gather are one grouped
// ncclSend / ncclRecv per peer -- link-parallel, no ring.  RCCL is bound at run time (dlopen of librccl.so.1:
// a process that already carries PyTorch's copy gets that one), so the library has no link-time dependency and
// the CPU simulator build has none at all.  The device-buffer collectives are STREAM-ORDERED on the context's stream:
// no host synchronisation before or after them (the library skips its own as well: cdbg_ctx::tr_ordered); only the small
// host-side all-gather of counts returns values to the host and waits.
#pragma once
#include "../../include/cdbg.h"

#ifndef CDBG_HOSTSIM
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#endif
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace cdbg {

#ifndef CDBG_HOSTSIM
// the slice of the RCCL API in use (rccl.h: ncclUint8 = 1, ncclInt32 = 2, ncclMax = 2, NCCL_UNIQUE_ID_BYTES = 128)
struct RcclApi {
    typedef struct { char internal[128]; } UniqueId;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    void* lib = nullptr;                                // published only after every symbol has resolved
    std::mutex mu;
    // (several host threads may initialise their contexts at once -- one thread per GPU in the CLI)
    bool load(std::string& err) {
        std::lock_guard<std::mutex> g(mu);
        if (lib) return true;
        void* h = nullptr;
        const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) { err = "cannot load librccl.so.1"; return false; }
#define CDBG_RCCL_SYM(field, name) do { *(void**)(&field) = dlsym(h, name); if (!field) { err = std::string("librccl lacks ") + name; return false; } } while (0)
        CDBG_RCCL_SYM(GetUniqueId, "ncclGetUniqueId"); CDBG_RCCL_SYM(CommInitRank, "ncclCommInitRank"); CDBG_RCCL_SYM(CommDestroy, "ncclCommDestroy");
        CDBG_RCCL_SYM(GroupStart, "ncclGroupStart"); CDBG_RCCL_SYM(GroupEnd, "ncclGroupEnd"); CDBG_RCCL_SYM(Send, "ncclSend"); CDBG_RCCL_SYM(Recv, "ncclRecv");
        CDBG_RCCL_SYM(AllGather, "ncclAllGather"); CDBG_RCCL_SYM(AllReduce, "ncclAllReduce"); CDBG_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef CDBG_RCCL_SYM
        lib = h;
        return true;
    }
};
inline RcclApi& rccl_api() { static RcclApi a; return a; }

// one RCCL communicator bound to a context's device and stream
struct RcclComm {
    void* comm = nullptr; int world = 1, rank = 0; hipStream_t stream{};
    uint64_t* d_small = nullptr;                        // staging of the host-side all-gather
    std::string err;
    bool ok(int rc, const char* what) { if (rc != 0) { err = std::string(what) + ": " + rccl_api().GetErrorString(rc); return false; } return true; }
    bool hip_ok(hipError_t e, const char* what) { if (e != hipSuccess) { err = std::string(what) + ": " + hipGetErrorString(e); return false; } return true; }

    static int all_gather_u64(void* u, const uint64_t* send, uint64_t* recv, int n) {
        RcclComm* c = (RcclComm*)u; RcclApi& A = rccl_api();
        const size_t nb = (size_t)n * 8;
        if ((size_t)n * (c->world + 1) > 4096) { c->err = "all_gather_u64: too many words"; return -1; }
        if (!c->hip_ok(hipMemcpyAsync(c->d_small, send, nb, hipMemcpyHostToDevice, c->stream), "H2D")) return -1;
        if (!c->ok(A.AllGather(c->d_small, c->d_small + n, nb, 1 /* ncclUint8 */, c->comm, c->stream), "ncclAllGather")) return -1;
        if (!c->hip_ok(hipMemcpyAsync(recv, c->d_small + n, nb * c->world, hipMemcpyDeviceToHost, c->stream), "D2H")) return -1;
        return c->hip_ok(hipStreamSynchronize(c->stream), "sync") ? 0 : -1;
    }
    static int all_to_all_v(void* u, const void* send, const uint64_t* soff, const uint64_t* scnt, void* recv, const uint64_t* roff, const uint64_t* rcnt) {
        RcclComm* c = (RcclComm*)u; RcclApi& A = rccl_api();
        if (!c->ok(A.GroupStart(), "ncclGroupStart")) return -1;
        bool good = true;
        for (int r = 0; r < c->world && good; ++r) {
            if (r == c->rank) continue;                  // the own block is a device-to-device copy
            if (scnt[r] && !c->ok(A.Send((const char*)send + soff[r], scnt[r], 1, r, c->comm, c->stream), "ncclSend")) good = false;
            if (good && rcnt[r] && !c->ok(A.Recv((char*)recv + roff[r], rcnt[r], 1, r, c->comm, c->stream), "ncclRecv")) good = false;
        }
        if (!good) { (void)A.GroupEnd(); return -1; }    // never leave the group open
        if (!c->ok(A.GroupEnd(), "ncclGroupEnd")) return -1;
        if (scnt[c->rank] && !c->hip_ok(hipMemcpyAsync((char*)recv + roff[c->rank], (const char*)send + soff[c->rank], scnt[c->rank], hipMemcpyDeviceToDevice, c->stream), "D2D")) return -1;
        return 0;                                        // (stream-ordered: what consumes the bytes is enqueued on the same stream)
    }
    static int all_gather_v(void* u, const void* send, uint64_t nbytes, void* recv, const uint64_t* roff, const uint64_t* rcnt) {
        RcclComm* c = (RcclComm*)u; RcclApi& A = rccl_api();
        if (!c->ok(A.GroupStart(), "ncclGroupStart")) return -1;
        bool good = true;
        for (int r = 0; r < c->world && good; ++r) {
            if (r == c->rank) continue;
            if (nbytes && !c->ok(A.Send(send, nbytes, 1, r, c->comm, c->stream), "ncclSend")) good = false;
            if (good && rcnt[r] && !c->ok(A.Recv((char*)recv + roff[r], rcnt[r], 1, r, c->comm, c->stream), "ncclRecv")) good = false;
        }
        if (!good) { (void)A.GroupEnd(); return -1; }
        if (!c->ok(A.GroupEnd(), "ncclGroupEnd")) return -1;
        if (nbytes && !c->hip_ok(hipMemcpyAsync((char*)recv + roff[c->rank], send, nbytes, hipMemcpyDeviceToDevice, c->stream), "D2D")) return -1;
        return 0;
    }
    static int all_reduce_max_i32(void* u, void* dev, uint64_t n) {
        RcclComm* c = (RcclComm*)u; RcclApi& A = rccl_api();
        if (n && !c->ok(A.AllReduce(dev, dev, n, 2 /* ncclInt32 */, 2 /* ncclMax */, c->comm, c->stream), "ncclAllReduce")) return -1;
        return 0;
    }
    bool init(const void* uid, int world_, int rank_, hipStream_t s) {
        RcclApi& A = rccl_api();
        if (!A.load(err)) return false;
        world = world_; rank = rank_; stream = s;
        RcclApi::UniqueId id; memcpy(&id, uid, sizeof id);
        if (!ok(A.CommInitRank(&comm, world, id, rank), "ncclCommInitRank")) return false;
        return hip_ok(hipMalloc(&d_small, 4096 * sizeof(uint64_t)), "hipMalloc");
    }
    void destroy() { if (comm) { (void)rccl_api().CommDestroy(comm); comm = nullptr; } if (d_small) { (void)hipFree(d_small); d_small = nullptr; } }
    cdbg_transport transport() { return cdbg_transport{ this, &all_gather_u64, &all_to_all_v, &all_gather_v, &all_reduce_max_i32 }; }
};
#endif  // !CDBG_HOSTSIM

}  // namespace cdbg
