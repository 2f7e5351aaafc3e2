// cdbg_impl.cpp -- libcdbg.so: host orchestration behind the C ABI of include/cdbg.h.
//
// This is the MI355X replacement for what GraphUnitigsTemplate<span>::create() does
// behind /root/reference/src/bcalm_1.cpp:57 (configure -> count -> bcalm -> bglue):
// pick the k-mer width (the Integer::apply dispatch of bcalm_1.cpp:95 becomes a
// template switch on W = 1, 2, 3, 4), pick minimizer size / partition count from the input
// volume (the job of DSK's configuration step), then drive
// the hand-written HIP kernels on one stream with everything resident in HBM.
//
// Built two ways from this same source:
//   hipcc --offload-arch=gfx950  -> bcalm_amd/_build/libcdbg.so   (the product)
//   g++ -DCDBG_HOSTSIM           -> tests/hostsim/_build/...      (kernel-logic simulator, tests only)
// The product has NO CPU fallback: without a HIP device cdbg_create() fails.
#include "../../include/cdbg.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <string>
#include <vector>

#include "k_links.h"
#include "k_verify.h"
#include "k_dglue.h"
#include "comm.h"
#include "k_count_fast.h"
#include "k_compact_wave.h"
#include "k_split.h"

using namespace cdbg;

namespace {

thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCK(call)                                                                                   \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail(e_ == hipErrorOutOfMemory ? CDBG_E_NOMEM : CDBG_E_NODEVICE, "%s failed: %s (%s:%d)", \
                        #call, hipGetErrorString(e_), __FILE__, __LINE__);                            \
    } while (0)
#define CK(expr) do { int rc_ = (expr); if (rc_ != CDBG_OK) return rc_; } while (0)

template <class T>
struct DBuf {                                   // owned device array
    T* p = nullptr; size_t n = 0, cap = 0;
    // (re)size to `count` elements; an existing allocation that is large enough is kept,
    // so that a context can be re-run (cdbg_reset) without touching the allocator
    // (floor_cap: never end up smaller than this -- buffers that are swapped with another one every step)
    int alloc(size_t count, bool zero, size_t floor_cap = 0) {
        size_t want = std::max<size_t>(count, 1);
        if (!p || cap < std::max(want, floor_cap)) {
            release();
            // Sizes that follow device-side reservations (piece ids, glue records: chunk tails stay unused) differ by a
            // fraction of a percent from one run of the same input to the next; without headroom every new maximum
            // re-allocated gigabytes in the middle of a step (measured: +230 ms in 3 of 26 steps at config 3).
            if (want > (1u << 16)) want += want / 32;
            want = std::max(want, floor_cap);
            hipError_t e = hipMalloc(&p, want * sizeof(T));
            if (e != hipSuccess) { p = nullptr; return fail(CDBG_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", want * sizeof(T), hipGetErrorString(e)); }
            cap = want;
        }
        n = count;
        if (zero) { hipError_t e = hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T)); if (e != hipSuccess) return fail(CDBG_E_NODEVICE, "hipMemset failed"); }
        return CDBG_OK;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; cap = 0; } }
    void swap(DBuf& o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); }
    DBuf() = default;
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    ~DBuf() { release(); }
};

// The record region outlives its context.  The scan's 1.6 G scattered 16-byte stores are sensitive to WHERE the 75 GB region
// lies physically: every free + re-allocation handed back a less contiguous set of pages, and five contexts created and
// destroyed in one process scanned in 66.7 -> 70.8 -> 68.9 -> 75.3 -> 77.9 ms (profiles/r03_scan_variance_by_allocation.log).
// A destroyed context therefore leaves its region with the process (one per device); the next context on that device adopts
// it when it is large enough.  cdbg_release_cached() gives it back to the driver.
struct RegionStash { std::mutex mu; uint64_t* p[64] = {}; size_t cap[64] = {}; };
RegionStash& region_stash() { static RegionStash st; return st; }
void stash_region(int dev, DBuf<uint64_t>& b) {
    if (dev < 0 || dev >= 64 || !b.p || b.cap < (1u << 24)) return;          // (small regions are not worth keeping)
    RegionStash& st = region_stash();
    std::lock_guard<std::mutex> g(st.mu);
    if (st.cap[dev] >= b.cap) return;                    // (the larger one stays; the caller's buffer is freed by its destructor)
    if (st.p[dev]) (void)hipFree(st.p[dev]);
    st.p[dev] = b.p; st.cap[dev] = b.cap; b.p = nullptr; b.n = 0; b.cap = 0;
}
void adopt_region(int dev, DBuf<uint64_t>& b, size_t want) {
    if (dev < 0 || dev >= 64 || b.cap >= want) return;
    RegionStash& st = region_stash();
    std::lock_guard<std::mutex> g(st.mu);
    if (st.cap[dev] < want) return;
    b.release(); b.p = st.p[dev]; b.cap = st.cap[dev]; b.n = 0; st.p[dev] = nullptr; st.cap[dev] = 0;
}

#ifndef CDBG_TSC1
#define CDBG_TSC1 4096
#endif
#ifndef CDBG_NTC1
#define CDBG_NTC1 512
#endif
#ifndef CDBG_TSC2
#define CDBG_TSC2 2048
#endif
#ifndef CDBG_TSC4
#define CDBG_TSC4 2048
#endif
constexpr int TS_COUNT_1 = CDBG_TSC1, TS_COUNT_2 = CDBG_TSC2, TS_COUNT_4 = CDBG_TSC4;      // LDS table slots per W
#ifndef CDBG_TSK1
#define CDBG_TSK1 1024
#endif
// compaction runs in up to two LDS tiers: the bucket table of TSK slots for buckets with <= TSK/2 entries, then a
// table twice the size for the deferred ones; only buckets beyond that use the HBM-resident tables.  Measured at
// config 3 / config 4 shapes: W = 2 gains from the small first tier (7 instead of 3 workgroups per CU: 133 -> 84 ms),
// W = 1 does not (its kernel is VALU bound and the denser table costs probes: 84 -> 96 ms), so W = 1 starts at 1024
constexpr int TS_COMPACT_1 = CDBG_TSK1, TS_COMPACT_2 = 512, TS_COMPACT_4 = 512;
template <int W> struct Cfg;
// TSW: slots of the wave-per-bucket compaction tier (buckets of at most TSW / 2 entries; k_compact_wave.h)
template <> struct Cfg<1> { static constexpr int TSC = TS_COUNT_1, TSK = TS_COMPACT_1, TSK2 = 2 * TS_COMPACT_1, NTC = CDBG_NTC1, TSW = 512, TSW2 = 512; };   // (TSW2 == TSW: no second wave tier)
#ifndef CDBG_TSW2
#define CDBG_TSW2 256
#endif
#ifndef CDBG_NTC2
#define CDBG_NTC2 512
#endif
template <> struct Cfg<2> { static constexpr int TSC = TS_COUNT_2, TSK = TS_COMPACT_2, TSK2 = 2 * TS_COMPACT_2, NTC = CDBG_NTC2, TSW = CDBG_TSW2, TSW2 = 2 * CDBG_TSW2; };
// (W >= 3: 512 threads with member-balanced wave shares and 8-record batches: 2 x 8 waves per CU instead of 2 x 4;
//  config-5 share: count tier 1 257 -> 214 ms, tier 2 87 -> 62 ms.  Before the balanced shares 512 threads LOST: 341 -> 464 ms)
#ifndef CDBG_NTC4
#define CDBG_NTC4 512
#endif
#ifndef CDBG_TSW4
#define CDBG_TSW4 256
#endif
template <> struct Cfg<4> { static constexpr int TSC = TS_COUNT_4, TSK = TS_COMPACT_4, TSK2 = 2 * TS_COMPACT_4, NTC = CDBG_NTC4, TSW = CDBG_TSW4, TSW2 = 2 * CDBG_TSW4; };
// three-word k-mers (64 <= k <= 95, the span-96 entry of the reference's KSIZE_LIST, README.md:93-99): the four-word geometry
// with 3/4 of the key bytes (count table 56 KB instead of 72)
template <> struct Cfg<3> { static constexpr int TSC = TS_COUNT_4, TSK = TS_COMPACT_4, TSK2 = 2 * TS_COMPACT_4, NTC = CDBG_NTC4, TSW = CDBG_TSW4, TSW2 = 2 * CDBG_TSW4; };

#ifndef CDBG_PGRID
#define CDBG_PGRID (256 * 12)
#endif
constexpr uint64_t PERSISTENT_GRID = CDBG_PGRID;     // persistent workgroups for the per-partition kernels (256 CUs)
// workgroups of `kern` that are resident at once on the whole device: the grid of a persistent kernel whose
// workgroups stride over equal work items must be exactly this (a partial extra generation would run alone)
template <class K>
uint64_t resident_grid(K kern, int threads, uint64_t fallback) {
#ifndef CDBG_HOSTSIM
    int occ = 0, dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, threads, 0) == hipSuccess && occ > 0 && cus > 0)
        return (uint64_t)occ * (uint64_t)cus;
#else
    (void)kern; (void)threads;
#endif
    return fallback;
}
// every persistent workgroup of every launch of a stage may leave one partly used output chunk behind: a stage has up to
// eight launches (count: one-pass, second tier, multi-pass retry, spill repair, HBM fallback; compact: the workgroup tiers and the HBM
// fallback, once over the buckets and once over the sub-buckets of the second-level split)
constexpr uint64_t CHUNK_SLACK_WGS = 8 * (PERSISTENT_GRID + 1);
constexpr uint64_t MAX_GRID = 1u << 22;          // workgroups per launch (grid * block must stay < 2^32)
uint64_t pow2_at_least(uint64_t x) { uint64_t p = 1; while (p < x) p <<= 1; return p; }
// slots of a junction table: 32-bit slot indices
int glue_table_slots(uint64_t want, uint32_t* out) {
    const uint64_t p = pow2_at_least(want);
    if (p > (1ull << 31)) return fail(CDBG_E_INTERNAL, "junction table of %llu slots exceeds 32-bit slot indices: shard the input over more GPUs", (unsigned long long)p);
    *out = (uint32_t)p; return CDBG_OK;
}

}  // namespace

struct cdbg_ctx {
    cdbg_params prm{};
    int W = 1, k = 0, m = 0, log_np = 0, rank_bits = 0;
    uint64_t n_local_parts = 1;
    hipStream_t stream{};
    int stage = 0;                               // 0 input, 1 counted, 2 compacted, 3 glued
    cdbg_stats_t st{};

    // Ingest: pushed bytes go through two pinned staging buffers and are copied to the device asynchronously on
    // their own stream while the caller parses the next chunk; the device text grows by doubling.
    static constexpr uint64_t STAGE_BYTES = 32ull << 20;
    uint64_t stage_bytes = STAGE_BYTES;          // (CDBG_STAGE_BYTES: smaller staging chunks, tests of the streaming scan)
    uint8_t* pin[2] = { nullptr, nullptr }; hipEvent_t pin_ev[2] = {}; bool pin_busy[2] = { false, false };
    int pin_cur = 0; uint64_t pin_fill = 0; hipStream_t copy_stream{};
    uint64_t n_dev = 0;                          // bytes of text already on (or on their way to) the device
    bool reads_final = false;                    // text complete, padded, nbytes set
    // streaming scan (cdbg_expect_input): tiles already scanned while the input was still arriving
    uint64_t expect_bytes = 0, ss_done = 0, ss_spill_cap = 0; uint32_t ss_part_cap = 0; bool ss_on = false;
    int log_np_override = -1;                    // set when a first count showed buckets too full for the LDS compaction tiers
    DBuf<uint8_t> reads; uint64_t nbytes = 0, nbytes_padded = 0;

    DBuf<uint32_t> part_count, spill_part; DBuf<uint64_t> part_off, part_cursor, records, exscan_tmp, spill_recs;
    DBuf<uint64_t> dstats; DBuf<uint32_t> derr;
    DBuf<uint64_t> solid_keys; DBuf<uint32_t> solid_cnt; DBuf<uint64_t> solid_cursor, seg_off; DBuf<uint32_t> seg_n;
    DBuf<uint32_t> big_list, big_count, big_list2, big_count2, retry_list;
    uint64_t n_solid_entries = 0;                // home + traveller solid entries

    DBuf<uint32_t> piece_n; DBuf<uint64_t> piece_kc, piece_boff; DBuf<uint8_t> piece_bases; DBuf<uint64_t> cursors;
    DBuf<uint64_t> glue_keys; DBuf<uint32_t> glue_a, glue_b, glue_conf; uint32_t glue_cap = 0;   // (fallback junction table)
    DBuf<uint32_t> retry_list2;                              // partitions that did not fit the second count tier either
    DBuf<uint32_t> var_cap; DBuf<uint64_t> var_pairs;        // single-pass layout of skewed inputs: region capacities, begin / end of every partition's records
    DBuf<uint64_t> split_keys, vseg_off, split_cur; DBuf<uint32_t> split_cnt, vseg_n, vlist_a, vlist_b;   // second-level bucket split (k_split.h)
    DBuf<uint64_t> repair_recs, repair_off, rp_idx; DBuf<uint32_t> repair_part, rp_flag, rp_size, rp_fill;   // capped-scan spill repair (kept: no allocation per step)
    DBuf<uint32_t> jfill; DBuf<uint64_t> jrecs;              // join buckets
    bool direct_join = false; int join_log_jb = 0;           // the compaction kernels filled the join buckets themselves (no junction log)
    DBuf<uint64_t> glog_keys; DBuf<uint32_t> glog_tag; uint64_t glog_cap = 0, n_glog = 0;
    uint64_t n_pieces = 0, n_piece_bases = 0;

    // multi-GPU merge staging (xchg_*)
    DBuf<uint32_t> mg_n, mg_gtag; DBuf<uint64_t> mg_kc, mg_boff, mg_gkeys; DBuf<uint8_t> mg_bases;
    DBuf<uint32_t> mg_ab, xp_ab;                         // -all-abundance-counts: merged per-base abundances / this rank's gap-free stream
    DBuf<uint64_t> xp_aoff, xr_aoff; uint64_t xp_nab = 0; bool xp_ab_ready = false;
    uint64_t last_add_np = 0, last_add_nb = 0, last_add_pieces = 0;   // where the latest xchg_add_packed put its pieces
    uint64_t mg_np = 0, mg_nb = 0, mg_nl = 0, mg_cap_p = 0, mg_cap_b = 0, mg_cap_l = 0; bool mg_open = false;
    // junction join result (cdbg_glue_join / first half of cdbg_glue): partner end of every piece end
    DBuf<uint32_t> link; bool joined = false; uint64_t n_join_local = 0;
    // packed exchange (xchg_sizes_packed / _add_packed): this rank's piece bases, 4 per byte, no gaps
    DBuf<uint8_t> xp_bases, xp_dense; DBuf<uint32_t> xp_lens; DBuf<uint64_t> xp_uoff; uint64_t xp_bytes = 0, xp_unpacked = 0;
    DBuf<uint32_t> xr_lens; DBuf<uint64_t> xr_uoff;      // receiver-side scratch of xchg_add_packed

    DBuf<uint64_t> unitig_off; DBuf<uint32_t> unitig_len; DBuf<uint64_t> unitig_kc; DBuf<uint8_t> unitig_bases, unitig_packed;   // packed: the same arena at 2 bits per base
    uint64_t n_unitigs = 0, unitig_total = 0;
    DBuf<uint32_t> piece_ab, unitig_ab;          // -all-abundance-counts
    DBuf<uint64_t> link_off; DBuf<uint32_t> link_to; uint64_t n_links = 0; bool linked = false;
    DBuf<uint4> rank_a, rank_b; DBuf<uint32_t> rank_flag;
    // multi-GPU: transport (RCCL or caller-supplied) and the record exchange buffers
    cdbg_transport tr{}; bool have_tr = false; uint64_t comm_bytes = 0;
    bool tr_ordered = false;                             // the transport enqueues on the context's stream (built-in RCCL): no host sync around a device-buffer collective
    bool force_multi = false;                            // CDBG_FORCE_MULTI: run the multi-rank code path with one rank (tests)
#ifndef CDBG_HOSTSIM
    RcclComm* rccl = nullptr;
#endif
    DBuf<uint32_t> xcnt; DBuf<uint64_t> xoff, xbase, xrecs;
    uint64_t piece_lo = 0, piece_hi = 0;                 // this rank's piece ids inside the merged arrays (owner-sharded emission)
    bool xchg_done = false;                              // the glue exchange of this run has happened
    DBuf<uint8_t> xg[5], xsend;    // list-ranking state (kept: a step must not allocate once the first step's buffers exist)
    // sharded glue (k_dglue.h): routing scratch, wire buffers, received pieces
    DBuf<uint8_t> dg_dest, dg_dense, dg_packed, dg_rpacked, dg_rdense; DBuf<uint64_t> dg_pos, dg_cnt, dg_wire_s, dg_wire_r, dg_meta_s, dg_meta_r, dg_boff, dg_uoff, dg_rboff, dg_rkc, dg_aoff, dg_raoff;
    DBuf<uint2> dg_pairs, dg_pair_s, dg_pair_r, dg_rs, dg_rr; DBuf<uint32_t> dg_qs, dg_qsrc, dg_qr, dg_lens, dg_alen, dg_rn, dg_rlens, dg_ab_s, dg_ab_r, dg_rab; DBuf<uint4> dg_rst;
};

namespace {

int stream_scan_dispatch(cdbg_ctx* c);
// ---- streaming ingest ----
int ingest_init(cdbg_ctx* c) {
    if (c->pin[0]) return CDBG_OK;
    if (const char* e = getenv("CDBG_STAGE_BYTES")) c->stage_bytes = std::min<uint64_t>(cdbg_ctx::STAGE_BYTES, std::max<uint64_t>(64, strtoull(e, nullptr, 10)));
    HIPCK(hipStreamCreate(&c->copy_stream));
    for (int i = 0; i < 2; ++i) {
        if (hipHostMalloc((void**)&c->pin[i], cdbg_ctx::STAGE_BYTES) != hipSuccess) return fail(CDBG_E_NOMEM, "pinned staging buffer (%llu bytes)", (unsigned long long)cdbg_ctx::STAGE_BYTES);
        HIPCK(hipEventCreate(&c->pin_ev[i]));
    }
    return CDBG_OK;
}
void ingest_release(cdbg_ctx* c) {
    for (int i = 0; i < 2; ++i) {
        if (c->pin[i]) { (void)hipHostFree(c->pin[i]); (void)hipEventDestroy(c->pin_ev[i]); c->pin[i] = nullptr; }
    }
    if (c->copy_stream) { (void)hipStreamDestroy(c->copy_stream); c->copy_stream = hipStream_t{}; }
}
// device text with room for `need` bytes: grows by doubling (device-to-device copy of what is already there)
int ingest_reserve(cdbg_ctx* c, uint64_t need) {
    if (c->reads.p && c->reads.cap >= need) return CDBG_OK;
    uint64_t cap = std::max<uint64_t>(c->reads.cap * 2, 256ull << 20);
    if (c->expect_bytes) cap = std::max<uint64_t>(cap, c->expect_bytes + c->expect_bytes / 64 + (8ull << 20));   // announced: one allocation
    while (cap < need) cap *= 2;
    if (c->ss_on) HIPCK(hipStreamSynchronize(c->stream));    // a streaming scan may be reading the old buffer
    DBuf<uint8_t> bigger;
    CK(bigger.alloc(cap, false));
    HIPCK(hipStreamSynchronize(c->copy_stream));             // copies into the old buffer have landed
    if (c->n_dev) HIPCK(hipMemcpy(bigger.p, c->reads.p, c->n_dev, hipMemcpyDeviceToDevice));
    c->reads.swap(bigger);
    return CDBG_OK;
}
// send the current staging buffer on its way and switch to the other one
int ingest_flush(cdbg_ctx* c) {
    if (!c->pin_fill) return CDBG_OK;
    CK(ingest_reserve(c, c->n_dev + c->pin_fill));
    const int b = c->pin_cur;
    HIPCK(hipMemcpyAsync(c->reads.p + c->n_dev, c->pin[b], c->pin_fill, hipMemcpyHostToDevice, c->copy_stream));
    HIPCK(hipEventRecord(c->pin_ev[b], c->copy_stream));
    c->pin_busy[b] = true;
    c->n_dev += c->pin_fill; c->pin_fill = 0;
    c->pin_cur = b ^ 1;
    if (c->pin_busy[b ^ 1]) { HIPCK(hipEventSynchronize(c->pin_ev[b ^ 1])); c->pin_busy[b ^ 1] = false; }   // its copy must be done before reuse
    if (c->expect_bytes && c->prm.world_size == 1 && !c->force_multi) CK(stream_scan_dispatch(c));
    return CDBG_OK;
}
int ingest_append(cdbg_ctx* c, const char* src, uint64_t n) {
    CK(ingest_init(c));
    while (n) {
        const uint64_t room = c->stage_bytes - c->pin_fill, take = std::min(room, n);
        memcpy(c->pin[c->pin_cur] + c->pin_fill, src, take);
        c->pin_fill += take; src += take; n -= take;
        if (c->pin_fill == c->stage_bytes) CK(ingest_flush(c));
    }
    return CDBG_OK;
}
// text complete: last partial buffer out, all copies done, tail padded with separators
int upload_pending(cdbg_ctx* c) {
    if (c->reads_final) return CDBG_OK;
    if (!c->pin[0] || (c->n_dev == 0 && c->pin_fill == 0)) {                   // nothing was pushed
        // a rank of a multi-GPU job may receive no reads at all (a small input dealt out in chunks): it still takes part
        // in every collective, with an empty text of separators
        if (c->prm.world_size > 1 || c->force_multi) {
            CK(c->reads.alloc(512, false));
            HIPCK(hipMemset(c->reads.p, '\n', 512));
            c->nbytes = 0; c->nbytes_padded = 256; c->reads_final = true;
        }
        return CDBG_OK;
    }
    CK(ingest_flush(c));
    const uint64_t n = c->n_dev;
    const uint64_t np = ((n + 15) / 16) * 16 + 256;
    CK(ingest_reserve(c, np));
    HIPCK(hipStreamSynchronize(c->copy_stream));
    HIPCK(hipMemset(c->reads.p + n, '\n', np - n));
    c->nbytes = n; c->nbytes_padded = np; c->reads_final = true;
    ingest_release(c);
    return CDBG_OK;
}

int read_u64(const uint64_t* dptr, uint64_t* out, size_t n = 1) {
    HIPCK(hipMemcpy(out, dptr, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return CDBG_OK;
}
int read_u32(const uint32_t* dptr, uint32_t* out, size_t n = 1) {
    HIPCK(hipMemcpy(out, dptr, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return CDBG_OK;
}
int check_device_error(cdbg_ctx* c, const char* where) {
    uint32_t e = 0; CK(read_u32(c->derr.p, &e));
    if (e) return fail(CDBG_E_INTERNAL, "%s: device reported error %u (1 solid overflow, 2 scratch sizing, 3 piece overflow, 4 unitig overflow, 5 glue log overflow, 9 junction-ownership flag of a k-mer wrong [simulator build only])", where, e);
    HIPCK(hipGetLastError());
    return CDBG_OK;
}
// dev aid (CDBG_HOST_MARKS=1): wall-clock marks on stderr between the host-side phases of a stage, to find time that no
// stage timer covers (allocations, host sorts, synchronous copies)
struct HostMarks {
    bool on = getenv("CDBG_HOST_MARKS") != nullptr; double t0 = now();
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
    void mark(const char* what) { if (!on) return; (void)hipDeviceSynchronize(); const double t = now(); fprintf(stderr, "[host] %-28s %8.2f ms\n", what, t - t0); t0 = t; }
};
struct Timer {
    hipEvent_t a{}, b{}; hipStream_t s{};
    int start(hipStream_t st) { s = st; HIPCK(hipEventCreate(&a)); HIPCK(hipEventCreate(&b)); HIPCK(hipEventRecord(a, s)); return CDBG_OK; }
    int stop(float* ms) { HIPCK(hipEventRecord(b, s)); HIPCK(hipEventSynchronize(b)); HIPCK(hipEventElapsedTime(ms, a, b)); (void)hipEventDestroy(a); (void)hipEventDestroy(b); return CDBG_OK; }
};

// ---------------------------------------------------------------------------------------
// configuration (DSK's "configure" role, row a5): partitions and minimizer length from volume
// ---------------------------------------------------------------------------------------
void configure(cdbg_ctx* c, uint64_t total_bytes) {
    const int W = c->W;
    const int ts = W == 1 ? TS_COUNT_1 : W == 2 ? TS_COUNT_2 : TS_COUNT_4;   // (W = 3 shares the four-word geometry)
    // mean k-mer occurrences per partition: ~0.3 distinct per occurrence at sequencing depth fills the
    // LDS table to ~45 %; inputs with more distinct k-mers per occurrence take several LDS passes
#ifndef CDBG_OCC_NUM
#define CDBG_OCC_NUM 3
#define CDBG_OCC_DEN 2
#endif
    // (four-word k-mers: at k = 127 three quarters of the k-mers of reads with 1 % errors are distinct, so a partition
    //  must hold fewer occurrences for its distinct k-mers to fit the one-pass table.  0.3 tables' worth until the count
    //  tiers learned to send a partition that will not fit straight to the bigger table; with that, twice the partition size
    //  halves the per-partition fixed costs for less than it adds to the second tier: 316 -> 307 ms at the config-5 share)
    const uint64_t target_occ = W >= 3 ? (uint64_t)ts * 6 / 10 : (uint64_t)ts * CDBG_OCC_NUM / CDBG_OCC_DEN;
    int log_np = c->log_np_override >= 0 ? c->log_np_override : c->prm.log2_partitions;
    if (log_np < 0) {
        log_np = 0;
        while (log_np < 24 && ((uint64_t)1 << log_np) * target_occ < total_bytes) ++log_np;
    }
    if (log_np < c->rank_bits) log_np = c->rank_bits;
    if (log_np > 26) log_np = 26;
    int m = c->prm.minimizer_size;
    if (m <= 0) m = std::min(16, std::max(6, (log_np + 10) / 2 + 1));
    m = std::max(1, std::min(m, std::min(16, c->k - 1)));
    c->log_np = log_np; c->m = m;
    c->n_local_parts = ((uint64_t)1 << log_np) >> c->rank_bits;
    c->st.minimizer_size = m; c->st.log2_partitions = log_np; c->st.kmer_words = W;
}

// exclusive prefix sum of n uint32 counts into n + 1 uint64 offsets (off[n] = total), on the context's stream
int exscan_u32(cdbg_ctx* c, const uint32_t* counts, uint64_t* off, uint64_t n) {
    hipStream_t s = c->stream;
    const uint64_t nb = (n + EXSCAN_BLOCK - 1) / EXSCAN_BLOCK;
    CK(c->exscan_tmp.alloc(nb + 1, false));
    if (n == 0) { HIPCK(hipMemsetAsync(off, 0, sizeof(uint64_t), s)); return CDBG_OK; }
    CDBG_LAUNCH(k_exscan_sums, nb, EXSCAN_THREADS, s, counts, c->exscan_tmp.p, n);
    CDBG_LAUNCH(k_exscan_top, 1, EXSCAN_THREADS, s, c->exscan_tmp.p, nb, off + n);
    CDBG_LAUNCH(k_exscan_apply, nb, EXSCAN_THREADS, s, counts, (const uint64_t*)c->exscan_tmp.p, off, n);
    return CDBG_OK;
}

// the scan kernel for this k / m / mode on the context's stream (persistent grid: resident workgroups)
template <int W, int MODE>
void launch_scan_mode(cdbg_ctx* c, ScanParams& sp, uint64_t grid) {
    hipStream_t s = c->stream;
    const bool fast_scan = c->k <= 63 && (c->k - c->m) <= SCANF_WNMAX;
    sp.n_tiles = grid;
    if (grid == 0) return;                                   // (a rank without reads)
    // compile-time minimizer windows (k - m): the k = 31 family m = 16 .. 12 and k = 55, m = 16 (config 4)
#define CDBG_SCAN_WNT(WW, WNT_)                                                                                                  \
    if (fast_scan && W == WW && c->k - c->m == WNT_) {                                                                           \
        CDBG_LAUNCH((k_scan_fast<W, MODE, W == WW ? WNT_ : 0>), std::min<uint64_t>(grid, resident_grid(k_scan_fast<W, MODE, W == WW ? WNT_ : 0>, SCAN_THREADS, SCANF_GRID)), SCAN_THREADS, s, sp); \
        return;                                                                                                                  \
    }
    CDBG_SCAN_WNT(1, 15) CDBG_SCAN_WNT(1, 16) CDBG_SCAN_WNT(1, 17) CDBG_SCAN_WNT(1, 18) CDBG_SCAN_WNT(1, 19) CDBG_SCAN_WNT(2, 39)
#undef CDBG_SCAN_WNT
    if (fast_scan) CDBG_LAUNCH((k_scan_fast<W, MODE, 0>), std::min<uint64_t>(grid, resident_grid(k_scan_fast<W, MODE, 0>, SCAN_THREADS, SCANF_GRID)), SCAN_THREADS, s, sp);
    else CDBG_LAUNCH((k_scan<W, MODE>), std::min<uint64_t>(grid, resident_grid(k_scan<W, MODE>, SCAN_THREADS, SCAN_GRID)), SCAN_THREADS, s, sp);
}
inline uint64_t scan_tile_bytes(const cdbg_ctx* c) { return (c->k <= 63 && (c->k - c->m) <= SCANF_WNMAX) ? (uint64_t)SCANF_TILE : (uint64_t)SCAN_TILE; }
void scan_params_base(cdbg_ctx* c, ScanParams& sp) {
    sp.reads = c->reads.p; sp.nbytes = c->nbytes; sp.nbytes_padded = c->nbytes_padded;
    sp.k = c->k; sp.m = c->m; sp.log_np = c->log_np; sp.rank_bits = c->rank_bits; sp.rank = c->prm.rank;
    sp.part_count = c->part_count.p; sp.part_cursor = c->part_cursor.p; sp.records = nullptr; sp.stats = c->dstats.p;
    sp.tile_stride = 1; sp.tile_offset = 0; sp.error = c->derr.p;
}
// capacity of a partition region from a sampled histogram (single-pass capped layout)
void capped_capacities(double mean, uint64_t NPL, uint32_t& part_cap, uint64_t& spill_cap) {
    part_cap = (uint32_t)(mean * 2.5 + 8.0 * std::sqrt(mean + 1.0) + 16.0);
    part_cap = (part_cap + 7u) & ~7u;
    if (const char* e = getenv("CDBG_PART_CAP")) part_cap = (uint32_t)std::max(1, atoi(e));   // test knob: force spills
    spill_cap = std::max<uint64_t>((uint64_t)(mean * (double)NPL / 32.0), 65536);
}

// ---------------------------------------------------------------------------------------
// Streaming scan: with cdbg_expect_input() the library knows the input volume before the last byte has
// arrived, so partitioning and region capacities are fixed from the first ~128 MB that landed and the single-pass scan
// runs on the tiles that are complete while the host is still parsing / copying the rest.
// ---------------------------------------------------------------------------------------
template <int W>
int stream_scan_advance(cdbg_ctx* c) {
    constexpr int RW = RecFmt<W>::RW;
    hipStream_t s = c->stream;
    const uint64_t landed = c->n_dev & ~15ull;
    if (!c->ss_on) {
        // (test knobs: CDBG_STREAM_MIN_BYTES / CDBG_STREAM_BATCH_TILES shrink the thresholds to simulator sizes)
        const char* emin = getenv("CDBG_STREAM_MIN_BYTES");
        const uint64_t min_bytes = emin ? strtoull(emin, nullptr, 10) : (128ull << 20);
        if (landed < std::min<uint64_t>(c->expect_bytes / 2, min_bytes)) return CDBG_OK;
        configure(c, c->expect_bytes);
        const uint64_t TB = scan_tile_bytes(c);
        const uint64_t tiles_now = landed > TB + 8192 ? (landed - 8192) / TB : 0;
        const uint64_t tiles_exp = (c->expect_bytes + TB - 1) / TB;
        if (!emin && (tiles_now < 1024 || tiles_exp <= 8192)) return CDBG_OK;    // small input: count decides
        if (tiles_now < 1) return CDBG_OK;
        const uint64_t NPL = c->n_local_parts;
        CK(c->part_count.alloc(NPL, true)); CK(c->part_off.alloc(NPL + 1, false)); CK(c->part_cursor.alloc(NPL, false));
        CK(c->dstats.alloc(32, true)); CK(c->derr.alloc(4, true)); CK(c->cursors.alloc(8, true));
        HIPCK(hipStreamSynchronize(c->copy_stream));                               // the sample reads what has landed
        ScanParams sp{}; c->nbytes = landed; c->nbytes_padded = landed; scan_params_base(c, sp);
        const uint64_t stride = std::min<uint64_t>(64, std::max<uint64_t>(1, tiles_now / 2048));
        const uint64_t ns = (tiles_now + stride - 1) / stride;
        sp.tile_stride = (uint32_t)stride;
        launch_scan_mode<W, SCAN_HIST>(c, sp, ns);
        CK(exscan_u32(c, c->part_count.p, c->part_off.p, NPL));
        uint64_t sample_records = 0; CK(read_u64(c->part_off.p + NPL, &sample_records));
        const double mean = (double)sample_records * (double)tiles_exp / (double)ns / (double)NPL;
        capped_capacities(mean, NPL, c->ss_part_cap, c->ss_spill_cap);
        if ((double)c->ss_part_cap * (double)NPL * RW * 8.0 > 200e9) { c->expect_bytes = 0; return CDBG_OK; }   // would not fit: no streaming
        CK(c->records.alloc((uint64_t)c->ss_part_cap * NPL * RW, false));
        CK(c->spill_recs.alloc(c->ss_spill_cap * RW, false)); CK(c->spill_part.alloc(c->ss_spill_cap, false));
        HIPCK(hipMemsetAsync(c->part_count.p, 0, NPL * sizeof(uint32_t), s));
        HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
        c->ss_on = true; c->ss_done = 0;
    }
    const uint64_t TB = scan_tile_bytes(c);
    const uint64_t tiles_now = landed > TB + 8192 ? (landed - 8192) / TB : 0;      // tiles whose halo has landed as well
    const char* ebt = getenv("CDBG_STREAM_BATCH_TILES");
    if (tiles_now < c->ss_done + (ebt ? strtoull(ebt, nullptr, 10) : 32768ull)) return CDBG_OK;   // batches of >= 128 MB
    // the kernel must see the bytes: order the compute stream behind the copies enqueued so far
    hipEvent_t ev; HIPCK(hipEventCreate(&ev));
    HIPCK(hipEventRecord(ev, c->copy_stream)); HIPCK(hipStreamWaitEvent(s, ev, 0)); (void)hipEventDestroy(ev);
    ScanParams sp{}; c->nbytes = landed; c->nbytes_padded = landed; scan_params_base(c, sp);
    sp.records = c->records.p; sp.part_cap = c->ss_part_cap; sp.part_fill = c->part_count.p;
    sp.spill_recs = c->spill_recs.p; sp.spill_part = c->spill_part.p; sp.spill_cursor = c->cursors.p + 6; sp.spill_cap = c->ss_spill_cap;
    sp.tile_offset = (uint32_t)c->ss_done;
    launch_scan_mode<W, SCAN_EMIT_CAPPED>(c, sp, tiles_now - c->ss_done);
    c->ss_done = tiles_now;
    return CDBG_OK;
}
int stream_scan_dispatch(cdbg_ctx* c) {
    switch (c->W) { case 1: return stream_scan_advance<1>(c); case 2: return stream_scan_advance<2>(c); case 3: return stream_scan_advance<3>(c); default: return stream_scan_advance<4>(c); }
}

// Several ranks: a rank-local failure between two collectives (out of memory, a device error, a bad input) must not leave
// the other ranks waiting inside the transport.  Before each collective stage the ranks exchange a status word; if any
// rank failed, every rank returns an error together.
int agree(cdbg_ctx* c, int rc, const char* where) {
    if (!(c->prm.world_size > 1 || c->force_multi) || !c->have_tr) return rc;
    const std::string mine = rc != CDBG_OK ? g_err : std::string();
    std::vector<uint64_t> all(c->prm.world_size); const uint64_t st = (uint64_t)(int64_t)rc;
    if (c->tr.all_gather_u64(c->tr.user, &st, all.data(), 1) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_u64 failed (%s)", where);
    if (rc != CDBG_OK) { g_err = mine; return rc; }
    for (int r = 0; r < c->prm.world_size; ++r)
        if (all[r] != 0) return fail(CDBG_E_INTERNAL, "%s: rank %d reported error %lld; all ranks stop", where, r, (long long)(int64_t)all[r]);
    return CDBG_OK;
}

template <int W>
int count_impl(cdbg_ctx* c) {
    constexpr int RW = RecFmt<W>::RW;
    constexpr int TS = Cfg<W>::TSC;
    const bool multi_ctx = c->prm.world_size > 1 || c->force_multi;
    const int world = c->prm.world_size;
    if (multi_ctx && !c->have_tr) return fail(CDBG_E_STATE, "world_size %d but no transport: call cdbg_comm_init_rccl or cdbg_set_transport first", world);
    int rc_in = upload_pending(c);
    if (rc_in == CDBG_OK && (!c->reads.p || (c->nbytes == 0 && !multi_ctx))) rc_in = fail(CDBG_E_STATE, "no reads: call cdbg_push_reads/cdbg_push_text/cdbg_generate_reads first");
    if (!multi_ctx) CK(rc_in);
    // multi-GPU, reads SHARDED over the ranks (X1): the scan fills the partitions of every rank and the records travel to
    // their owners; every rank must choose the same partitioning, so the input volume that drives configure() is the sum
    // over the ranks.  Reads REPLICATED (X0): every rank scans the whole text for its own partitions, nothing travels.
    const bool multi = multi_ctx && !c->prm.reads_replicated;
    uint64_t total_bytes = c->nbytes;
    if (multi_ctx) {
        // one small all-gather: every rank's input status (a rank-local failure stops all ranks together) and byte count
        const std::string mine_err = rc_in != CDBG_OK ? g_err : std::string();
        std::vector<uint64_t> all(2 * (size_t)world); const uint64_t mine[2] = { (uint64_t)(int64_t)rc_in, c->nbytes };
        if (c->tr.all_gather_u64(c->tr.user, mine, all.data(), 2) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_u64 failed");
        if (rc_in != CDBG_OK) { g_err = mine_err; return rc_in; }
        for (int r = 0; r < world; ++r) if (all[2 * r]) return fail(CDBG_E_INTERNAL, "count: input: rank %d reported error %lld; all ranks stop", r, (long long)(int64_t)all[2 * r]);
        if (multi) { total_bytes = 0; for (int r = 0; r < world; ++r) total_bytes += all[2 * r + 1]; }
        else for (int r = 0; r < world; ++r) if (all[2 * r + 1] != mine[1]) return fail(CDBG_E_PARAM, "reads_replicated: the ranks hold different texts (%llu vs %llu bytes)", (unsigned long long)mine[1], (unsigned long long)all[2 * r + 1]);
        if (total_bytes == 0) return fail(CDBG_E_STATE, "no reads on any rank");
    }
    if (!c->ss_on) configure(c, total_bytes);             // (a streaming scan fixed the partitioning from the announced volume)
    const uint64_t NPL = c->n_local_parts;
    const uint64_t NPS = multi ? (NPL << c->rank_bits) : NPL;    // partition slots the scan fills: all of them when the reads are sharded
    hipStream_t s = c->stream;
    HostMarks hm;
    Timer t_total; CK(t_total.start(s));

    if (!c->ss_on) {
        CK(c->part_count.alloc(NPS, true));
        CK(c->part_off.alloc(NPS + 1, false));
        CK(c->part_cursor.alloc(NPS, false));
        CK(c->dstats.alloc(32, true));
        CK(c->derr.alloc(4, true));
        CK(c->cursors.alloc(8, true));
    }

    hm.mark("count: allocations");
    ScanParams sp{};
    scan_params_base(c, sp);
    sp.emit_all = multi ? 1u : 0u; sp.npl = (uint32_t)NPL;
    // instruction-lean scan when the window fits registers; generic LDS-doubling scan otherwise
    const bool fast_scan = c->k <= 63 && (c->k - c->m) <= SCANF_WNMAX;
    const uint64_t tiles = fast_scan ? (c->nbytes + SCANF_TILE - 1) / SCANF_TILE : (c->nbytes + SCAN_TILE - 1) / SCAN_TILE;
    c->st.n_launch_scan = tiles;
#define LAUNCH_SCAN(MODE, GRID) launch_scan_mode<W, MODE>(c, sp, (GRID))
    auto exscan = [&](const uint32_t* counts) -> int {       // counts -> part_off (exclusive), part_off[NPS] = total
        return exscan_u32(c, counts, c->part_off.p, NPS);
    };

    // Record placement.  exact : histogram pass + emit pass at exact offsets (two scans, zero slack).
    //                    capped: ONE scan into fixed-capacity partition regions sized from a sampled
    //                            histogram; the rare records that do not fit go to a spill list and their
    //                            partitions are repaired (gathered contiguously) before counting.
    // (sharded reads: the exact layout is what travels -- no slack on the wire; a single-pass scan into capped regions is
    //  squeezed into it by k_pack_regions, which costs one pass over the rank's records instead of a second pass over its reads)
    bool capped = tiles > 8192;
    if (const char* e = getenv("CDBG_SCAN_MODE")) { if (!strcmp(e, "exact")) capped = false; else if (!strcmp(e, "capped")) capped = true; }
    if (tiles == 0) capped = false;                          // (a rank without reads: nothing to sample)
    // var: ONE pass into regions of their own size per partition, estimated from a denser sample -- what a skewed input gets instead
    // of the exact two-pass layout (CDBG_SCAN_MODE=var: test knob)
    bool var = false;
    if (const char* e = getenv("CDBG_SCAN_MODE")) { if (!strcmp(e, "var") && tiles && !multi) { var = true; capped = false; } }
    uint64_t n_records = 0, hs[2] = {0, 0};
    uint32_t part_cap = 0; uint64_t n_spill = 0; bool packed_exact = false;
    uint64_t n_spilled_parts = 0;                            // partitions whose region overflowed (capped mode): counted from gathered copies
    Timer t;
    uint64_t spill_cap = 0;
    if (c->ss_on) capped = true;                             // tiles [0, ss_done) were scanned while the input was arriving
    if (capped) {
        bool fits = true;
        if (c->ss_on) { part_cap = c->ss_part_cap; spill_cap = c->ss_spill_cap; }
        else {
            CK(t.start(s));
            const uint64_t stride = std::min<uint64_t>(64, std::max<uint64_t>(1, tiles / 4096));
            const uint64_t ns = (tiles + stride - 1) / stride;
            sp.tile_stride = (uint32_t)stride;
            LAUNCH_SCAN(SCAN_HIST, ns);
            CK(exscan(c->part_count.p));
            uint64_t sample_records = 0; CK(read_u64(c->part_off.p + NPS, &sample_records));
            // the fullest partition of the sample: a skewed input (repeats, low complexity, coverage peaks) puts far more
            // into some partitions than any capacity covers; the capped pass would then hammer a few fill counters and spill
            // (145 ms at the hostile config-3 line before falling back) -- such inputs go straight to the exact two-pass layout
            HIPCK(hipMemsetAsync(c->dstats.p + 31, 0, sizeof(uint64_t), s));
            CDBG_LAUNCH(k_max_u32, std::min<uint64_t>((NPS + 255) / 256, 4096), 256, s, (const uint32_t*)c->part_count.p, NPS, c->dstats.p + 31);
            uint64_t sample_max = 0; CK(read_u64(c->dstats.p + 31, &sample_max));
            CK(t.stop(&c->st.ms_scan_hist));
            const double mean = (double)sample_records * (double)tiles / (double)ns / (double)NPS;
            capped_capacities(mean, NPS, part_cap, spill_cap);
            if ((double)part_cap * (double)NPS * RW * 8.0 > 200e9) fits = false;       // would not fit: use the exact layout
            // (at least 32 sampled records in that partition: with a mean of a few records per partition -- long reads, k = 127 --
            //  the sampled maximum is Poisson noise, and scaling it up sent the config-5 share through two passes: 598 -> 662 ms)
            else if (sample_max >= 32 && (double)sample_max * (double)tiles / (double)ns > 8.0 * (double)part_cap && getenv("CDBG_SCAN_MODE") == nullptr) { fits = false; var = !multi; }
            if (fits) {
                if (c->xrecs.cap > c->records.cap) c->records.swap(c->xrecs);   // (sharded reads: the previous step left the region buffer there)
                adopt_region(c->prm.device_id, c->records, (uint64_t)part_cap * NPS * RW);   // (what an earlier context of this process left behind)
                CK(c->records.alloc((uint64_t)part_cap * NPS * RW, false));
                CK(c->spill_recs.alloc(spill_cap * RW, false)); CK(c->spill_part.alloc(spill_cap, false));
                HIPCK(hipMemsetAsync(c->part_count.p, 0, NPS * sizeof(uint32_t), s));
                HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
            }
        }
        if (!fits) capped = false;
        else {
            sp.tile_stride = 1; sp.records = c->records.p; sp.part_cap = part_cap; sp.part_fill = c->part_count.p;
            sp.spill_recs = c->spill_recs.p; sp.spill_part = c->spill_part.p; sp.spill_cursor = c->cursors.p + 6; sp.spill_cap = spill_cap;
            CK(t.start(s));
            const uint64_t done = c->ss_on ? std::min<uint64_t>(c->ss_done, tiles) : 0;
            sp.tile_offset = (uint32_t)done;
            if (tiles > done) LAUNCH_SCAN(SCAN_EMIT_CAPPED, tiles - done);
            sp.tile_offset = 0;
            c->st.n_tiles_overlapped = done;
            CK(exscan(c->part_count.p));                     // only for the total number of records
            CK(t.stop(&c->st.ms_scan_emit));
            hm.mark("count: sample + capped scan");
            CK(read_u64(c->part_off.p + NPS, &n_records));
            CK(read_u64(c->dstats.p, hs, 2));
            CK(read_u64(c->cursors.p + 6, &n_spill));
            uint32_t derr = 0; CK(read_u32(c->derr.p, &derr));
            c->ss_on = false;                                // (the streamed part is accounted for; a re-count scans everything)
            if (derr == 6 || n_spill > spill_cap || (multi && n_spill)) {   // estimate was off (very skewed input): exact layout instead
                capped = false; HIPCK(hipMemset(c->derr.p, 0, 4 * sizeof(uint32_t))); HIPCK(hipMemset(c->cursors.p + 6, 0, sizeof(uint64_t)));
            } else if (multi) {
                // what travels is the exact owner-major layout: squeeze the regions (part_off = exclusive scan of the fills)
                CK(c->xrecs.alloc(std::max<uint64_t>(n_records, 1) * RW, false));
                PackRegionParams pk{ c->records.p, c->part_count.p, c->part_off.p, NPS, part_cap, RW, c->xrecs.p };
                CDBG_LAUNCH(k_pack_regions, std::min<uint64_t>((NPS + 3) / 4, 256 * 16), 256, s, pk);
                c->records.swap(c->xrecs);
                capped = false; packed_exact = true;
            } else if (n_spill) {
                // repair: gather region + spilled records of each spilled partition into one contiguous run (k_count.h)
                RepairParams rp{};
                rp.records = c->records.p; rp.spill_recs = c->spill_recs.p; rp.spill_part = c->spill_part.p; rp.n_spill = n_spill;
                rp.part_fill = c->part_count.p; rp.npl = NPL; rp.part_cap = part_cap; rp.RW = RW;
                CK(c->rp_flag.alloc(NPL, false)); CK(c->rp_idx.alloc(NPL + 1, false));
                rp.flag = c->rp_flag.p; rp.ridx = c->rp_idx.p;
                CDBG_LAUNCH(k_repair_flag, (NPL + 255) / 256, 256, s, rp);
                CK(exscan_u32(c, c->rp_flag.p, c->rp_idx.p, NPL));
                CK(read_u64(c->rp_idx.p + NPL, &n_spilled_parts));
                const uint64_t nsp = n_spilled_parts;
                CK(c->repair_part.alloc(nsp, false)); CK(c->rp_size.alloc(nsp, false)); CK(c->repair_off.alloc(nsp + 1, false)); CK(c->rp_fill.alloc(nsp, true));
                rp.item_part = c->repair_part.p; rp.item_size = c->rp_size.p; rp.item_off = c->repair_off.p; rp.item_fill = c->rp_fill.p;
                CDBG_LAUNCH(k_repair_list, (NPL + 255) / 256, 256, s, rp);
                CK(exscan_u32(c, c->rp_size.p, c->repair_off.p, nsp));
                uint64_t total = 0; CK(read_u64(c->repair_off.p + nsp, &total));
                CK(c->repair_recs.alloc(total * RW, false));
                rp.out = c->repair_recs.p;
                CDBG_LAUNCH(k_repair_gather, nsp, 256, s, rp);
                CDBG_LAUNCH(k_repair_scatter, std::min<uint64_t>((n_spill + 255) / 256, 1u << 16), 256, s, rp);
            }
        }
    }
    if (var && !capped && !packed_exact) {
        const float ms_sample1 = c->st.ms_scan_hist;
        CK(t.start(s));
        HIPCK(hipMemsetAsync(c->part_count.p, 0, NPS * sizeof(uint32_t), s));
        HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
        const uint64_t stride = tiles >= 64 ? 4 : 1;         // a quarter of the tiles: +- 10 % on a partition of 400 records
        const uint64_t ns = (tiles + stride - 1) / stride;
        sp.tile_stride = (uint32_t)stride; sp.tile_offset = 0; sp.part_cap = 0; sp.var_limit = nullptr;
        LAUNCH_SCAN(SCAN_HIST, ns);
        CK(c->var_cap.alloc(NPS, false)); CK(c->var_pairs.alloc(2 * NPS, false));
        uint64_t sample_records = 0;
        CK(exscan(c->part_count.p)); CK(read_u64(c->part_off.p + NPS, &sample_records));
        const double scale = (double)tiles / (double)ns, mean = (double)sample_records * scale / (double)NPS;
        uint32_t cap_min = 0; capped_capacities(mean, NPS, cap_min, spill_cap);
        VarParams vp{ c->part_count.p, c->var_cap.p, NPS, (float)scale, cap_min, c->part_off.p, c->part_cursor.p, c->var_pairs.p, c->dstats.p + 30 };
        if (const char* e = getenv("CDBG_VAR_SCALE")) vp.scale = (float)atof(e);   // test knob (with CDBG_PART_CAP): regions far too small, so that partitions spill
        CDBG_LAUNCH(k_var_caps, (NPS + 255) / 256, 256, s, vp);
        CK(exscan_u32(c, c->var_cap.p, c->part_off.p, NPS));
        uint64_t total_cap = 0; CK(read_u64(c->part_off.p + NPS, &total_cap));
        float ms2 = 0; CK(t.stop(&ms2)); c->st.ms_scan_hist = ms_sample1 + ms2;
        if ((double)total_cap * RW * 8.0 > 200e9) var = false;                     // would not fit: the exact layout
        else {
            spill_cap = std::max<uint64_t>(total_cap / 32, 65536);
            CK(c->records.alloc(total_cap * RW, false));
            CK(c->spill_recs.alloc(spill_cap * RW, false)); CK(c->spill_part.alloc(spill_cap, false));
            HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
            HIPCK(hipMemsetAsync(c->cursors.p + 6, 0, sizeof(uint64_t), s));
            CK(t.start(s));
            CDBG_LAUNCH(k_copy_u64, (NPS + 255) / 256, 256, s, (const uint64_t*)c->part_off.p, c->part_cursor.p, NPS);
            sp.tile_stride = 1; sp.records = c->records.p; sp.var_limit = c->part_off.p + 1;
            sp.spill_recs = c->spill_recs.p; sp.spill_part = c->spill_part.p; sp.spill_cursor = c->cursors.p + 6; sp.spill_cap = spill_cap;
            LAUNCH_SCAN(SCAN_EMIT, tiles);
            sp.var_limit = nullptr;
            CDBG_LAUNCH(k_var_finish, (NPS + 255) / 256, 256, s, vp);
            CK(t.stop(&c->st.ms_scan_emit));
            hm.mark("count: samples + single-pass scan into estimated regions");
            CK(read_u64(c->dstats.p + 30, &n_records));
            CK(read_u64(c->dstats.p, hs, 2));
            CK(read_u64(c->cursors.p + 6, &n_spill));
            uint32_t derr = 0; CK(read_u32(c->derr.p, &derr));
            if (derr == 6 || n_spill > spill_cap) {          // the estimate was off by more than the spill list holds: exact layout
                var = false; HIPCK(hipMemset(c->derr.p, 0, 4 * sizeof(uint32_t))); HIPCK(hipMemset(c->cursors.p + 6, 0, sizeof(uint64_t)));
            } else if (n_spill) {
                RepairParams rp{};
                rp.records = c->records.p; rp.spill_recs = c->spill_recs.p; rp.spill_part = c->spill_part.p; rp.n_spill = n_spill;
                rp.part_fill = nullptr; rp.npl = NPL; rp.part_cap = 0; rp.RW = RW; rp.var_off = c->part_off.p; rp.var_cursor = c->part_cursor.p;
                CK(c->rp_flag.alloc(NPL, false)); CK(c->rp_idx.alloc(NPL + 1, false));
                rp.flag = c->rp_flag.p; rp.ridx = c->rp_idx.p;
                CDBG_LAUNCH(k_repair_flag, (NPL + 255) / 256, 256, s, rp);
                CK(exscan_u32(c, c->rp_flag.p, c->rp_idx.p, NPL));
                CK(read_u64(c->rp_idx.p + NPL, &n_spilled_parts));
                const uint64_t nsp = n_spilled_parts;
                CK(c->repair_part.alloc(nsp, false)); CK(c->rp_size.alloc(nsp, false)); CK(c->repair_off.alloc(nsp + 1, false)); CK(c->rp_fill.alloc(nsp, true));
                rp.item_part = c->repair_part.p; rp.item_size = c->rp_size.p; rp.item_off = c->repair_off.p; rp.item_fill = c->rp_fill.p;
                CDBG_LAUNCH(k_repair_list, (NPL + 255) / 256, 256, s, rp);
                CK(exscan_u32(c, c->rp_size.p, c->repair_off.p, nsp));
                uint64_t total = 0; CK(read_u64(c->repair_off.p + nsp, &total));
                CK(c->repair_recs.alloc(total * RW, false));
                rp.out = c->repair_recs.p;
                CDBG_LAUNCH(k_repair_gather, nsp, 256, s, rp);
                CDBG_LAUNCH(k_repair_scatter, std::min<uint64_t>((n_spill + 255) / 256, 1u << 16), 256, s, rp);
            }
        }
    }
    if (!capped && !packed_exact && !var) {
        sp.tile_stride = 1; sp.part_cap = 0;
        HIPCK(hipMemsetAsync(c->part_count.p, 0, NPS * sizeof(uint32_t), s));
        HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
        // pass 1: histogram of records per partition
        CK(t.start(s));
        LAUNCH_SCAN(SCAN_HIST, tiles);
        CK(exscan(c->part_count.p));
        CK(t.stop(&c->st.ms_scan_hist));
        CK(read_u64(c->part_off.p + NPS, &n_records));
        CK(read_u64(c->dstats.p, hs, 2));
        // pass 2: emit records at exact offsets
        CK(c->records.alloc(std::max<uint64_t>(n_records, 1) * RW, false));
        CK(t.start(s));
        CDBG_LAUNCH(k_copy_u64, (NPS + 255) / 256, 256, s, (const uint64_t*)c->part_off.p, c->part_cursor.p, NPS);
        sp.records = c->records.p;
        LAUNCH_SCAN(SCAN_EMIT, tiles);
        CK(t.stop(&c->st.ms_scan_emit));
    }
    if (multi) {
        // ---- record exchange: block r of the record array (the partitions rank r owns) goes to rank r ----
        Timer tx; CK(tx.start(s));
        CK(c->xcnt.alloc((uint64_t)world * NPL, false)); CK(c->xoff.alloc((uint64_t)world * (NPL + 1), false)); CK(c->xbase.alloc(world, false));
        std::vector<uint64_t> so(world), sc(world), ro(world), rc(world);
        // per-partition counts first (equal blocks of NPL counts)
        for (int r = 0; r < world; ++r) { so[r] = (uint64_t)r * NPL * 4; sc[r] = NPL * 4; ro[r] = so[r]; rc[r] = sc[r]; }
        if (!c->tr_ordered) HIPCK(hipStreamSynchronize(s));
        if (c->tr.all_to_all_v(c->tr.user, c->part_count.p, so.data(), sc.data(), c->xcnt.p, ro.data(), rc.data()) != 0) return fail(CDBG_E_INTERNAL, "transport all_to_all_v (counts) failed");
        c->comm_bytes += 2 * (uint64_t)(world - 1) * NPL * 4;
        // where the records of sender s start inside its block, per partition; block sizes
        std::vector<uint64_t> xb(world + 1, 0);
        for (int r = 0; r < world; ++r) {
            CK(exscan_u32(c, c->xcnt.p + (uint64_t)r * NPL, c->xoff.p + (uint64_t)r * (NPL + 1), NPL));
            uint64_t tot = 0; CK(read_u64(c->xoff.p + (uint64_t)r * (NPL + 1) + NPL, &tot));
            xb[r + 1] = xb[r] + tot;
        }
        HIPCK(hipMemcpy(c->xbase.p, xb.data(), world * sizeof(uint64_t), hipMemcpyHostToDevice));
        CK(c->xrecs.alloc(std::max<uint64_t>(xb[world], 1) * RW, false));
        for (int r = 0; r < world; ++r) {
            uint64_t b[1], e[1]; CK(read_u64(c->part_off.p + (uint64_t)r * NPL, b)); CK(read_u64(c->part_off.p + (uint64_t)(r + 1) * NPL, e));
            so[r] = b[0] * RW * 8; sc[r] = (e[0] - b[0]) * RW * 8;
            ro[r] = xb[r] * RW * 8; rc[r] = (xb[r + 1] - xb[r]) * RW * 8;
            if (r != c->prm.rank) c->comm_bytes += sc[r] + rc[r];
        }
        if (c->tr.all_to_all_v(c->tr.user, c->records.p, so.data(), sc.data(), c->xrecs.p, ro.data(), rc.data()) != 0) return fail(CDBG_E_INTERNAL, "transport all_to_all_v (records) failed");
        // merge the blocks: partition lp = its segments in sender order
        CK(c->part_count.alloc(NPL, false)); CK(c->part_off.alloc(NPL + 1, false));
        SumCountParams scp{ c->xcnt.p, world, NPL, c->part_count.p };
        CDBG_LAUNCH(k_sum_counts, (NPL + 255) / 256, 256, s, scp);
        CK(exscan_u32(c, c->part_count.p, c->part_off.p, NPL));
        n_records = xb[world];
        CK(c->records.alloc(std::max<uint64_t>(n_records, 1) * RW, false));
        HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
        MergeRecParams mp{ c->xcnt.p, c->xoff.p, c->xbase.p, world, NPL, RW, c->xrecs.p, c->part_off.p, c->records.p, c->dstats.p };
        CDBG_LAUNCH(k_merge_records, std::min<uint64_t>((NPL + 3) / 4, 8192), 256, s, mp);
        HIPCK(hipStreamSynchronize(s));
        CK(read_u64(c->dstats.p, hs, 2));
        float msx = 0; CK(tx.stop(&msx)); c->st.ms_exchange += msx;
    }
#undef LAUNCH_SCAN
#ifdef CDBG_PROFILE_PHASES
    { uint64_t ph[6]; CK(read_u64(c->dstats.p + 16, ph, 6)); fprintf(stderr, "k_scan phase ticks: load %llu keys %llu winmin %llu flags %llu collect %llu emit %llu\n", (unsigned long long)ph[0], (unsigned long long)ph[1], (unsigned long long)ph[2], (unsigned long long)ph[3], (unsigned long long)ph[4], (unsigned long long)ph[5]); }
#endif
    c->st.n_records = n_records; c->st.n_member_kmers = hs[0];
    hm.mark("count: spill repair/exchange");

    // count
    // (slack: every persistent workgroup of every launch of the stage may strand one partly used chunk)
    // (launches of the stage: one-pass tier 1, tier 2 of at most 256 workgroups, multi-pass retry, spill repair, HBM tables)
    const uint64_t solid_cap = hs[0] / (uint64_t)std::max(1, c->prm.abundance_min) + 4096 + (4 * std::min<uint64_t>(NPL, PERSISTENT_GRID) + 256 + 5) * (uint64_t)COUNT_CHUNK;
    CK(c->solid_keys.alloc(solid_cap * W, false));
    CK(c->solid_cnt.alloc(solid_cap, false));
    CK(c->solid_cursor.alloc(4, true));
    CK(c->seg_off.alloc(NPL, true));
    CK(c->seg_n.alloc(NPL, true));
    CK(c->big_list.alloc(NPL, false));
    CK(c->retry_list.alloc(NPL, false));
    CK(c->big_count.alloc(4, true));                         // [0] partitions for the HBM pass, [1] partitions for the multi-pass kernel
    HIPCK(hipMemset(c->dstats.p, 0, 32 * sizeof(uint64_t)));

    CountParams cp{};
    cp.records = c->records.p; cp.part_off = c->part_off.p; cp.part_list = nullptr;
    if (capped) { cp.part_stride = part_cap; cp.part_fill = c->part_count.p; }
    if (var) { cp.part_off = c->var_pairs.p; cp.part_pairs = 1u; }   // (regions of estimated size: begin / end per partition)
    cp.k = c->k; cp.amin = (uint32_t)c->prm.abundance_min;
    cp.solid_keys = c->solid_keys.p; cp.solid_cnt = c->solid_cnt.p; cp.solid_cap = solid_cap; cp.solid_cursor = c->solid_cursor.p;
    cp.seg_off = c->seg_off.p; cp.seg_n = c->seg_n.p; cp.stats = c->dstats.p;
    cp.big_list = c->big_list.p; cp.big_count = c->big_count.p; cp.error = c->derr.p;
    hm.mark("count: solid buffers");
    CK(t.start(s));
    cp.n_items = (uint32_t)NPL; cp.max_passes = 64;
    // one-pass kernel over all partitions; the ones whose distinct k-mers do not fit the LDS table at once come back on
    // the retry list and go through the multi-pass kernel
    {
        // (admission by predicted fill: k_count_fast.h; one-word k-mers: off -- their second tier runs one workgroup per CU against three)
        CountFastParams fp{ cp, c->retry_list.p, c->big_count.p + 1, count_fast_record_limit<W>(c->k), W == 1 ? 0u : W == 2 ? 177u : 200u };
        if (const char* e = getenv("CDBG_FAST_SKIP_Q8")) fp.skip_fill_q8 = (uint32_t)std::max(0, atoi(e));   // dev knob
        if (const char* e = getenv("CDBG_FAST_MAX_RECORDS")) fp.fast_max_records = std::min<uint32_t>(count_fast_record_limit<W>(c->k), (uint32_t)std::max(1, atoi(e)));   // dev knob
        if (capped) CDBG_LAUNCH((k_count_fast<W, TS, Cfg<W>::NTC, true>), std::min<uint64_t>(NPL, PERSISTENT_GRID), Cfg<W>::NTC, s, fp);
        else CDBG_LAUNCH((k_count_fast<W, TS, Cfg<W>::NTC, false>), std::min<uint64_t>(NPL, PERSISTENT_GRID), Cfg<W>::NTC, s, fp);
    }
    c->st.n_launch_count = NPL;
    uint32_t nretry = 0;
    HIPCK(hipStreamSynchronize(s));
    hm.mark("count: tier 1");
    CK(read_u32(c->big_count.p + 1, &nretry));
    const uint32_t* retry_ptr = c->retry_list.p;
    if (nretry && getenv("CDBG_NO_COUNT_TIER2") == nullptr) {
        // second tier: the same one-pass kernel with a table twice the size (one workgroup per CU) over the retry list; at the
        // config-5 share 6 % of the partitions -- a minimizer locus of long reads -- cost 250 of 590 ms in the multi-pass kernel
        CK(c->retry_list2.alloc(nretry, false));
        HIPCK(hipMemsetAsync(c->big_count.p + 2, 0, sizeof(uint32_t), s));
        CountParams c2 = cp; c2.part_list = c->retry_list.p; c2.n_items = nretry;
        // (three- and four-word k-mers: the admission rule here as well -- a partition predicted beyond 0.68 of the 4096 slots goes to
        //  the multi-pass kernel untried: count 216 -> 201 ms at the config-5 share; two-word k-mers: no difference, off)
        CountFastParams fp2{ c2, c->retry_list2.p, c->big_count.p + 2, count_fast_record_limit<W>(c->k), W >= 3 ? 175u : 0u };
        if (const char* e = getenv("CDBG_FAST_SKIP2_Q8")) fp2.skip_fill_q8 = (uint32_t)std::max(0, atoi(e));   // dev knob
        // (multi-word k-mers: 1024 threads -- the table fills the CU's LDS either way, so the workgroup size IS the occupancy: 16
        //  waves per CU instead of 8, second tier 81 -> 67 ms at the config-5 share, 24 -> 18 at the config-4 share)
#ifndef CDBG_NT_TIER2
#define CDBG_NT_TIER2 1024
#endif
        constexpr int NT2 = W == 1 ? Cfg<W>::NTC : CDBG_NT_TIER2;
        if (capped) CDBG_LAUNCH((k_count_fast<W, 2 * TS, NT2, 3>), std::min<uint64_t>(nretry, 256), NT2, s, fp2);
        else CDBG_LAUNCH((k_count_fast<W, 2 * TS, NT2, 2>), std::min<uint64_t>(nretry, 256), NT2, s, fp2);
        HIPCK(hipStreamSynchronize(s));
        CK(read_u32(c->big_count.p + 2, &nretry));
        retry_ptr = c->retry_list2.p;
    }
    c->st.n_multipass_partitions = nretry;
    if (nretry) {
        CountParams rp1 = cp;
        rp1.part_list = retry_ptr; rp1.n_items = nretry;
        // (multi-word k-mers: the table of the second tier and 1024 threads -- half the passes at 16 waves per CU: 45 -> 39 ms at the config-5 share)
        constexpr int TSG = W == 1 ? TS : 2 * TS, NTG = W == 1 ? Cfg<W>::NTC : 1024;
        CDBG_LAUNCH((k_count<W, TSG, NTG, false>), std::min<uint64_t>(nretry, W == 1 ? PERSISTENT_GRID : 256), NTG, s, rp1);
    }
    if (n_spilled_parts) {                                   // spilled partitions: count their gathered copies
        CountParams rp2 = cp;
        rp2.records = c->repair_recs.p; rp2.item_off = c->repair_off.p; rp2.part_list = c->repair_part.p; rp2.part_stride = 0;
        rp2.n_items = (uint32_t)n_spilled_parts; rp2.max_passes = 4096;
        if (const char* ev = getenv("CDBG_REPAIR_MAX_PASSES")) rp2.max_passes = (uint32_t)std::max(1, atoi(ev));   // (tests: a spilled partition that is deferred as well)
        constexpr int TSG = W == 1 ? TS : 2 * TS, NTG = W == 1 ? Cfg<W>::NTC : 1024;
        CDBG_LAUNCH((k_count<W, TSG, NTG, false>), std::min<uint64_t>(rp2.n_items, W == 1 ? PERSISTENT_GRID : 256), NTG, s, rp2);
    }
    uint32_t nbig = 0;
    HIPCK(hipStreamSynchronize(s));
    hm.mark("count: tier 2 + multi-pass");
    CK(read_u32(c->big_count.p, &nbig));
    DBuf<uint64_t> g_keys, big_off; DBuf<uint32_t> g_cnt;
    if (nbig) {                                              // partitions whose distinct k-mers overflow LDS
        std::vector<uint32_t> bl(nbig); CK(read_u32(c->big_list.p, bl.data(), nbig));
        std::sort(bl.begin(), bl.end());
        std::vector<uint64_t> offs(nbig + 1, 0);
        const uint64_t nmax = (uint64_t)RecFmt<W>::CAPB - c->k + 1;
        // (records of every listed partition: one bulk copy of the fill / offset array when the list is long -- a skewed input
        //  lists 10^4 partitions, and a synchronous 4-byte copy each cost 77 ms per step at the hostile config-3 line)
        std::vector<uint32_t> h_fill; std::vector<uint64_t> h_off;
        if (nbig > 64) {
            if (capped) { h_fill.resize(NPL); CK(read_u32(c->part_count.p, h_fill.data(), NPL)); }
            else if (var) { h_off.resize(2 * NPL); CK(read_u64(c->var_pairs.p, h_off.data(), 2 * NPL)); }
            else { h_off.resize(NPL + 1); CK(read_u64(c->part_off.p, h_off.data(), NPL + 1)); }
        }
        for (uint32_t i = 0; i < nbig; ++i) {
            uint64_t nrec_p;
            if (!h_fill.empty()) nrec_p = h_fill[bl[i]];
            else if (!h_off.empty()) nrec_p = var ? h_off[2 * (size_t)bl[i] + 1] - h_off[2 * (size_t)bl[i]] : h_off[bl[i] + 1] - h_off[bl[i]];
            else if (capped) { uint32_t f = 0; CK(read_u32(c->part_count.p + bl[i], &f)); nrec_p = f; }
            else if (var) { uint64_t po[2]; CK(read_u64(c->var_pairs.p + 2 * (size_t)bl[i], po, 2)); nrec_p = po[1] - po[0]; }
            else { uint64_t po[2]; CK(read_u64(c->part_off.p + bl[i], po, 2)); nrec_p = po[1] - po[0]; }
            const uint64_t occ = nrec_p * nmax;
            offs[i + 1] = offs[i] + pow2_at_least(2 * occ + 4 * 256);
        }
        CK(g_keys.alloc(offs[nbig] * W, false)); CK(g_cnt.alloc(offs[nbig], false));
        CK(big_off.alloc(nbig + 1, false));
        HIPCK(hipMemcpy(big_off.p, offs.data(), (nbig + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
        HIPCK(hipMemcpy(c->big_list.p, bl.data(), nbig * sizeof(uint32_t), hipMemcpyHostToDevice));
        CountParams bp = cp;
        bp.part_list = c->big_list.p; bp.g_keys = g_keys.p; bp.g_cnt = g_cnt.p; bp.big_off = big_off.p;
        bp.n_items = nbig; bp.max_passes = 1;
        // (grid bounded: every workgroup reserves whole output chunks, the slack is sized for PERSISTENT_GRID)
        CDBG_LAUNCH((k_count<W, TS, 256, true>), std::min<uint64_t>(nbig, PERSISTENT_GRID), 256, s, bp);
        c->st.n_big_partitions += nbig;
    }
    CK(t.stop(&c->st.ms_count));
    hm.mark("count: HBM-table partitions");
    CK(check_device_error(c, "count"));
    uint64_t cs[4]; CK(read_u64(c->dstats.p, cs, 4));
#ifdef CDBG_PROFILE_PHASES
    { uint64_t ph[16]; CK(read_u64(c->dstats.p + 8, ph, 16));
      for (int w = 0; w < 2; ++w) fprintf(stderr, "k_count_fast phase cycles, %s wave, summed over WGs: loop-end->top %llu | wait own records + stage %llu | insert %llu | barrier A %llu | sweep %llu | barrier B %llu\n", w ? "last" : "first",
          (unsigned long long)ph[8 * w + 0], (unsigned long long)ph[8 * w + 1], (unsigned long long)ph[8 * w + 2], (unsigned long long)ph[8 * w + 3], (unsigned long long)ph[8 * w + 4], (unsigned long long)ph[8 * w + 5]); }
#endif
    c->st.n_distinct = cs[0]; c->st.n_occurrences = cs[1]; c->st.n_solid = cs[2]; c->st.n_solid_travellers = cs[3];
    CK(read_u64(c->solid_cursor.p, &c->n_solid_entries));
    if (getenv("CDBG_DEBUG_SEGHIST")) {                      // dev aid: solid entries per bucket, log2 bins (stderr)
        std::vector<uint32_t> sn(NPL); CK(read_u32(c->seg_n.p, sn.data(), NPL));
        uint64_t nb[32] = {0}, ne[32] = {0};
        for (uint64_t p = 0; p < NPL; ++p) { int b = 0; while ((1u << b) <= sn[p] && b < 31) ++b; ++nb[b]; ne[b] += sn[p]; }
        for (int b = 0; b < 32; ++b) if (nb[b]) fprintf(stderr, "[seghist] entries < 2^%-2d : %10llu buckets %12llu entries\n", b, (unsigned long long)nb[b], (unsigned long long)ne[b]);
    }
    c->st.input_bytes = c->nbytes;
    float ms = 0; CK(t_total.stop(&ms)); c->st.ms_total = ms;
    c->stage = 1;
    return CDBG_OK;
}

template <int W>
int compact_impl(cdbg_ctx* c) {
    constexpr int TS = Cfg<W>::TSK;
    if (c->stage < 1) return fail(CDBG_E_STATE, "cdbg_compact before cdbg_count");
    hipStream_t s = c->stream;
    const uint64_t NPL = c->n_local_parts;
    const uint64_t S = c->st.n_solid;
    HostMarks hm;
    Timer t; CK(t.start(s));
    // (the junction join works from the glue LOG; its tables are built in cdbg_glue)
    CK(c->cursors.alloc(8, false));

    // Single-rank contexts: the glue records go straight into the join buckets of the glue stage (k_glue.h) instead of a
    // sequential log that a scatter pass re-reads; contexts that exchange the log with other ranks, and the global-table
    // join (CDBG_GLUE_TABLE), keep the log.  Observed 1.8-2.0 records per solid traveller (bound: 3): buckets sized for
    // a mean fill of at most 96 of JB_CAP = 256 at 2.2, 131 at the bound.
    bool direct = !(c->prm.world_size > 1 || c->force_multi) && getenv("CDBG_GLUE_TABLE") == nullptr && getenv("CDBG_GLUE_LOG") == nullptr;
    int log_jb = 0;
    { const uint64_t est = c->st.n_solid_travellers * 22 / 10 + 1024; while ((96ull << log_jb) < est && log_jb < 26) ++log_jb; }
    if (const char* ev = getenv("CDBG_JOIN_LOG_JB")) log_jb = std::max(0, std::min(26, atoi(ev)));   // (tests: force the overflow fallback)
    for (int attempt = 0; attempt < 2; ++attempt) {
        // glue log: <= 2 open ends + 1 confirm per junction, one junction per solid traveller at most; the tail of a
        // chunk that the next bucket does not fit into is abandoned, hence the generous second attempt
        // (every persistent wave of tier 0 may strand one partly used chunk of each output array as well)
        // (... in the wave tiers over the buckets and over the sub-buckets of the second-level split)
        const uint64_t wave_slack = 2 * std::min<uint64_t>(NPL, 256ull * 32) + 2 * std::min<uint64_t>(c->n_solid_entries / 32 + 4, 256ull * 32);
        c->glog_cap = (attempt == 0 ? 3 : 8) * c->st.n_solid_travellers + (attempt + 1) * (CHUNK_SLACK_WGS * (uint64_t)GLOG_CHUNK + wave_slack * CW_GLOG_CHUNK) + 64;
        if (direct) {
            c->glog_cap = ~0ull >> 2;                        // (the log cursor only counts)
            CK(c->jfill.alloc(1ull << log_jb, false)); CK(c->jrecs.alloc((JB_CAP << log_jb) * (uint64_t)(W + 1), false));
            HIPCK(hipMemsetAsync(c->jfill.p, 0, sizeof(uint32_t) << log_jb, s));
        } else {
            CK(c->glog_keys.alloc(c->glog_cap * W, false)); CK(c->glog_tag.alloc(c->glog_cap, false));
        }
        const uint64_t pslack = CHUNK_SLACK_WGS * (uint64_t)PIECE_CHUNK + wave_slack * CW_PIECE_CHUNK, bslack = CHUNK_SLACK_WGS * (uint64_t)BASES_CHUNK + wave_slack * CW_BASES_CHUNK;
        const uint64_t pcap = (attempt == 0 ? std::min<uint64_t>(S, S / 3 + 4096) + 16 : S + 16) + pslack;
        const uint64_t bcap = (attempt == 0 ? S + (pcap - pslack) * (uint64_t)(c->k - 1) + 64 : S * (uint64_t)c->k + 64) + bslack;
        CK(c->piece_n.alloc(pcap, false)); HIPCK(hipMemsetAsync(c->piece_n.p, 0, pcap * sizeof(uint32_t), s));
        CK(c->piece_kc.alloc(pcap, false)); CK(c->piece_boff.alloc(pcap, false));
        CK(c->piece_bases.alloc(bcap, false));
        if (c->prm.all_abundance_counts) CK(c->piece_ab.alloc(bcap, false));
        HIPCK(hipMemsetAsync(c->cursors.p, 0, 8 * sizeof(uint64_t), s));
        if (!direct) HIPCK(hipMemsetAsync(c->glog_tag.p, 0xFF, c->glog_cap * sizeof(uint32_t), s));
        HIPCK(hipMemsetAsync(c->big_count.p, 0, 4 * sizeof(uint32_t), s));
        HIPCK(hipMemsetAsync(c->derr.p, 0, 4 * sizeof(uint32_t), s));
        HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));

        CompactParams kp{};
        kp.solid_keys = c->solid_keys.p; kp.solid_cnt = c->solid_cnt.p; kp.seg_off = c->seg_off.p; kp.seg_n = c->seg_n.p;
        kp.part_list = nullptr; kp.k = c->k; kp.m = c->m; kp.log_np = c->log_np; kp.rank_bits = c->rank_bits; kp.rank = c->prm.rank;
        kp.piece_n = c->piece_n.p; kp.piece_kc = c->piece_kc.p; kp.piece_boff = c->piece_boff.p; kp.piece_bases = c->piece_bases.p;
        kp.piece_ab = c->prm.all_abundance_counts ? c->piece_ab.p : nullptr;
        kp.piece_cap = pcap; kp.bases_cap = bcap; kp.piece_cursor = c->cursors.p; kp.bases_cursor = c->cursors.p + 1;
        kp.glue_keys = nullptr; kp.glue_a = nullptr; kp.glue_b = nullptr; kp.glue_conf = nullptr; kp.glue_mask = 0;
        kp.glog_keys = c->glog_keys.p; kp.glog_tag = c->glog_tag.p; kp.glog_cap = c->glog_cap; kp.glog_cursor = c->cursors.p + 4;
        kp.jfill = direct ? c->jfill.p : nullptr; kp.jrecs = direct ? c->jrecs.p : nullptr; kp.log_jb = log_jb;
        kp.big_list = c->big_list.p; kp.big_count = c->big_count.p; kp.error = c->derr.p; kp.stats = c->dstats.p;
        kp.n_items = (uint32_t)NPL;
        // The LDS tiers over the buckets 0 .. n of `base` (their segments: base.seg_off / seg_n): one wave per bucket
        // (k_compact_wave.h), W >= 2: again one wave each with a table twice the size, then a workgroup per bucket with an LDS
        // table of TS and of 2 TS slots (k_compact.h).  Every tier hands the buckets beyond its table to the next on a list;
        // la / lb: the two lists (>= n entries each).  Returns the survivors (count, and which list holds them).
        auto lds_tiers = [&](const CompactParams& base, uint32_t n, DBuf<uint32_t>& la, DBuf<uint32_t>& lb, uint32_t& nleft, const uint32_t*& left) -> int {
            CK(c->big_count.alloc(4, false)); CK(c->big_count2.alloc(4, false));
            HIPCK(hipMemsetAsync(c->big_count.p, 0, 4 * sizeof(uint32_t), s)); HIPCK(hipMemsetAsync(c->big_count2.p, 0, 4 * sizeof(uint32_t), s));
            uint32_t nbig = 0;
            {   // tier 0
                CompactParams k0 = base; k0.part_list = nullptr; k0.n_items = n; k0.big_list = la.p; k0.big_count = c->big_count.p;
                HIPCK(hipMemsetAsync(c->cursors.p + 5, 0, sizeof(uint64_t), s));       // the bucket queue (re)starts
                CompactWaveParams wp{ k0, n, reinterpret_cast<uint32_t*>(c->cursors.p + 5) };
                const uint64_t wgrid = resident_grid(k_compact_wave<W, Cfg<W>::TSW>, CW_THREADS, 256 * 3);
                CDBG_LAUNCH((k_compact_wave<W, Cfg<W>::TSW>), std::min<uint64_t>(((uint64_t)n + CW_THREADS / 64 - 1) / (CW_THREADS / 64), wgrid), CW_THREADS, s, wp);
                HIPCK(hipStreamSynchronize(s));
                CK(read_u32(c->big_count.p, &nbig));
            }
            DBuf<uint32_t>* cur = &la; DBuf<uint32_t>* oth = &lb; uint32_t* cnt_cur = c->big_count.p; uint32_t* cnt_oth = c->big_count2.p;
            auto next_tier = [&](CompactParams& kt) { kt = base; kt.part_list = cur->p; kt.n_items = nbig; kt.big_list = oth->p; kt.big_count = cnt_oth; };
            auto flip = [&]() -> int { HIPCK(hipStreamSynchronize(s)); CK(read_u32(cnt_oth, &nbig)); std::swap(cur, oth); std::swap(cnt_cur, cnt_oth);
                                       HIPCK(hipMemsetAsync(cnt_oth, 0, 4 * sizeof(uint32_t), s)); return CDBG_OK; };
            if (nbig && Cfg<W>::TSW2 > Cfg<W>::TSW) {
                // tier 0b: the deferred buckets again one wave each, with a table twice the size (fewer waves per CU, but no
                // workgroup barriers: at the config-4 share the workgroup tier below spent 44 ms on the 129..256-entry buckets)
                CompactParams k0; next_tier(k0);
                HIPCK(hipMemsetAsync(c->cursors.p + 5, 0, sizeof(uint64_t), s));
                CompactWaveParams wp{ k0, nbig, reinterpret_cast<uint32_t*>(c->cursors.p + 5) };
                const uint64_t wgrid = resident_grid(k_compact_wave<W, Cfg<W>::TSW2>, CW_THREADS, 256 * 2);
                CDBG_LAUNCH((k_compact_wave<W, Cfg<W>::TSW2>), std::min<uint64_t>(((uint64_t)nbig + CW_THREADS / 64 - 1) / (CW_THREADS / 64), wgrid), CW_THREADS, s, wp);
                CK(flip());
            }
            if (nbig) {                                      // tier 1: a workgroup per bucket, LDS table of TS slots
                CompactParams k1; next_tier(k1);
                CDBG_LAUNCH((k_compact<W, TS, false>), std::min<uint64_t>(nbig, PERSISTENT_GRID), COMPACT_THREADS, s, k1);
                CK(flip());
            }
            if (nbig) {                                      // tier 2: the deferred buckets with a table twice the size
                CompactParams k2; next_tier(k2);
                CDBG_LAUNCH((k_compact<W, Cfg<W>::TSK2, false>), std::min<uint64_t>(nbig, PERSISTENT_GRID), COMPACT_THREADS, s, k2);
                CK(flip());
            }
            nleft = nbig; left = cur->p;
            return CDBG_OK;
        };
        uint32_t nbig = 0; const uint32_t* left = nullptr;
        CK(c->big_list.alloc(NPL, false)); CK(c->big_list2.alloc(NPL, false));
        CK(lds_tiers(kp, (uint32_t)NPL, c->big_list, c->big_list2, nbig, left));
        c->st.n_launch_compact = NPL;
        hm.mark("compact: buffers + LDS tiers");
        CompactParams kh = kp; uint64_t n_src = NPL;         // what the HBM tier below reads: the buckets themselves, or their sub-buckets
        if (nbig && getenv("CDBG_NO_SPLIT") == nullptr) {
            // Second-level split (k_split.h): what no LDS tier could take is re-bucketed by junction into sub-buckets of ~100 entries,
            // which go through the same tiers again (the hostile config-3 line spent 76 ms walking 17 K such buckets through HBM tables)
            CK(c->split_cur.alloc(4, true));
            SplitParams sp{ kp.solid_keys, kp.solid_cnt, kp.seg_off, kp.seg_n, left, nbig, c->k, c->m, nullptr, nullptr, 0, nullptr, nullptr, 0, c->split_cur.p, c->derr.p };
            CDBG_LAUNCH(k_split_measure, (nbig + 255) / 256, 256, s, sp);
            uint64_t need[2] = {0, 0}; CK(read_u64(c->split_cur.p + 2, need, 2));
            if (need[1] >= (1ull << 31)) return fail(CDBG_E_INTERNAL, "bucket split: %llu sub-buckets exceed 31-bit ids", (unsigned long long)need[1]);
            CK(c->split_keys.alloc(2 * need[0] * W + W, false)); CK(c->split_cnt.alloc(2 * need[0] + 1, false));
            CK(c->vseg_off.alloc(need[1] + 1, false)); CK(c->vseg_n.alloc(need[1] + 1, false));
            CK(c->vlist_a.alloc(need[1] + 1, false)); CK(c->vlist_b.alloc(need[1] + 1, false));
            sp.out_keys = c->split_keys.p; sp.out_cnt = c->split_cnt.p; sp.out_cap = 2 * need[0]; sp.vseg_off = c->vseg_off.p; sp.vseg_n = c->vseg_n.p; sp.vcap = (uint32_t)need[1];
            CDBG_LAUNCH((k_split_buckets<W>), std::min<uint64_t>(nbig, PERSISTENT_GRID), SPLIT_THREADS, s, sp);
            kh = kp; kh.solid_keys = c->split_keys.p; kh.solid_cnt = c->split_cnt.p; kh.seg_off = c->vseg_off.p; kh.seg_n = c->vseg_n.p; kh.split = 1u;
            n_src = need[1];
            c->st.n_split_buckets += nbig;
            CK(lds_tiers(kh, (uint32_t)need[1], c->vlist_a, c->vlist_b, nbig, left));
            hm.mark("compact: split + LDS tiers");
        }
        DBuf<uint64_t> g_keys, big_off; DBuf<uint32_t> g_cnt, g_lnk, g_aux, hb_list;
        if (nbig) {                                          // buckets with more entries than fit LDS
            std::vector<uint32_t> bl(nbig); CK(read_u32(left, bl.data(), nbig));
            std::sort(bl.begin(), bl.end());
            std::vector<uint64_t> offs(nbig + 1, 0);
            std::vector<uint32_t> h_segn;
            if (nbig > 64) { h_segn.resize(n_src); CK(read_u32(kh.seg_n, h_segn.data(), n_src)); }   // (one bulk copy, not one per bucket)
            for (uint32_t i = 0; i < nbig; ++i) {
                uint32_t e = 0;
                if (!h_segn.empty()) e = h_segn[bl[i]]; else CK(read_u32(kh.seg_n + bl[i], &e));
                offs[i + 1] = offs[i] + pow2_at_least(2 * (uint64_t)e + 16);
            }
            CK(g_keys.alloc(offs[nbig] * W, false)); CK(g_cnt.alloc(offs[nbig], false));
            CK(g_lnk.alloc(2 * offs[nbig], false)); CK(g_aux.alloc(3 * offs[nbig], false));
            CK(big_off.alloc(nbig + 1, false)); CK(hb_list.alloc(nbig, false));
            HIPCK(hipMemcpy(big_off.p, offs.data(), (nbig + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
            HIPCK(hipMemcpy(hb_list.p, bl.data(), nbig * sizeof(uint32_t), hipMemcpyHostToDevice));
            CompactParams bp = kh;
            bp.part_list = hb_list.p; bp.g_keys = g_keys.p; bp.g_cnt = g_cnt.p;
            bp.g_lnk = g_lnk.p; bp.g_aux = g_aux.p; bp.big_off = big_off.p;
            bp.n_items = nbig;
            CDBG_LAUNCH((k_compact<W, TS, true>), std::min<uint64_t>(nbig, PERSISTENT_GRID), COMPACT_THREADS, s, bp);
            HIPCK(hipStreamSynchronize(s));
        }
        uint32_t e = 0; CK(read_u32(c->derr.p, &e));
        if (e == 8 && direct) { direct = false; --attempt; continue; }   // a join bucket overflowed (cannot happen with a sound hash): through the log instead
        if ((e == 3 || e == 5) && attempt == 0) continue;    // piece arrays / glue log too small: retry with the safe bounds
        if (nbig) c->st.n_big_partitions += nbig;
        break;
    }
    CK(t.stop(&c->st.ms_compact));
    hm.mark("compact: workgroup tiers");
    CK(check_device_error(c, "compact"));
    uint64_t cur[5]; CK(read_u64(c->cursors.p, cur, 5));
    c->n_pieces = cur[0]; c->n_piece_bases = cur[1]; c->n_glog = cur[4];
    c->direct_join = direct; c->join_log_jb = log_jb;
    uint64_t ks[4]; CK(read_u64(c->dstats.p, ks, 4));
#ifdef CDBG_PROFILE_PHASES
    { uint64_t ph[9]; CK(read_u64(c->dstats.p + 8, ph, 9)); fprintf(stderr, "k_compact_wave phase cycles (summed over waves): between buckets %llu | load+mins %llu | classify %llu | mutual+terminals %llu | walk1(+cycles) %llu | reserve+confirms %llu | walk2+prefix bases %llu | last bases+glog %llu | reset %llu\n",
        (unsigned long long)ph[0], (unsigned long long)ph[1], (unsigned long long)ph[2], (unsigned long long)ph[3], (unsigned long long)ph[4], (unsigned long long)ph[5], (unsigned long long)ph[6], (unsigned long long)ph[7], (unsigned long long)ph[8]); }
#endif
    c->st.n_pieces = ks[3]; c->st.n_glue_open_ends = ks[0]; c->st.n_cycles = ks[2];
    c->st.ms_total += c->st.ms_compact;
    c->stage = 2; c->joined = false;
    return CDBG_OK;
}

// Glue, first half: hash-join the piece ends on their junction (k-1)-mers -> link[end] = partner end.
// sharded (multi-GPU, after xchg_*): this rank joins only the junctions whose key hash selects it --
// 1/world of the device atomics -- and leaves the other ends at NONE; the caller combines the link arrays of all
// ranks with an element-wise MAX all-reduce (every end is set by exactly one rank) before cdbg_glue.
template <int W>
int glue_join_impl(cdbg_ctx* c, bool sharded) {
    if (c->stage < 2) return fail(CDBG_E_STATE, "cdbg_glue before cdbg_compact");
    hipStream_t s = c->stream;
    const uint64_t NP = c->n_pieces;
    if (2 * NP >= 0x7FFFFFF0ULL) return fail(CDBG_E_INTERNAL, "too many pieces for 31-bit end ids (%llu)", (unsigned long long)NP);
    const uint32_t NS = (uint32_t)(2 * NP);
    Timer t; CK(t.start(s));
    CK(c->link.alloc(NS, false));
    HIPCK(hipMemsetAsync(c->link.p, 0xFF, (size_t)std::max<uint32_t>(NS, 1) * sizeof(uint32_t), s));
    HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
    HIPCK(hipMemsetAsync(c->derr.p, 0, 4 * sizeof(uint32_t), s));
    const uint32_t world = sharded ? (uint32_t)c->prm.world_size : 1u;
    const uint64_t n_mine = c->n_glog / world + (world > 1 ? (c->n_glog >> 6) + 1024 : 0);      // records this rank joins (estimate when sharded)
    bool bucketed = (getenv("CDBG_GLUE_TABLE") == nullptr || c->direct_join) && c->n_glog > 0;
    if (bucketed && c->direct_join) {                        // the buckets were filled by the compaction kernels
        const uint64_t JB = 1ull << c->join_log_jb;
        JoinBucketParams bp{ c->jfill.p, c->jrecs.p, (uint32_t)JB, c->link.p, c->dstats.p, nullptr, nullptr, 0, nullptr };
        CDBG_LAUNCH((k_join_bucket<W>), std::min<uint64_t>((JB + 3) / 4, 256 * 16), JB_THREADS, s, bp);
    } else if (bucketed) {
        // bucketed join (k_glue.h): scatter the log into buckets of ~JB_CAP / 2 records, one wave joins a bucket in LDS
        int log_jb = 0; while (((uint64_t)(JB_CAP / 2) << log_jb) < n_mine && log_jb < 26) ++log_jb;
        const uint64_t JB = 1ull << log_jb;
        CK(c->jfill.alloc(JB, false)); CK(c->jrecs.alloc(JB * JB_CAP * (W + 1), false));
        HIPCK(hipMemsetAsync(c->jfill.p, 0, JB * sizeof(uint32_t), s));
        JoinScatterParams sp{ c->glog_keys.p, c->glog_tag.p, c->n_glog, log_jb, c->jfill.p, c->jrecs.p, c->derr.p,
                              world - 1, world > 1 ? (uint32_t)c->prm.rank : 0u };
        CDBG_LAUNCH((k_join_scatter<W>), std::min<uint64_t>((c->n_glog + GLUE_THREADS - 1) / GLUE_THREADS, 1u << 16), GLUE_THREADS, s, sp);
        JoinBucketParams bp{ c->jfill.p, c->jrecs.p, (uint32_t)JB, c->link.p, c->dstats.p, nullptr, nullptr, 0, nullptr };
        CDBG_LAUNCH((k_join_bucket<W>), std::min<uint64_t>((JB + 3) / 4, 256 * 16), JB_THREADS, s, bp);
        HIPCK(hipStreamSynchronize(s));
        uint32_t e = 0; CK(read_u32(c->derr.p, &e));
        if (e == 8) {                                        // a bucket overflowed (cannot happen with a sound hash): global table instead
            bucketed = false;
            HIPCK(hipMemsetAsync(c->link.p, 0xFF, (size_t)std::max<uint32_t>(NS, 1) * sizeof(uint32_t), s));
            HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
            HIPCK(hipMemsetAsync(c->derr.p, 0, 4 * sizeof(uint32_t), s));
        }
    }
    if (!bucketed && c->n_glog) {
        // fallback: one junction table in HBM (at most one junction per glue record)
        CK(glue_table_slots(n_mine + n_mine / 4 + 1024, &c->glue_cap));
        CK(c->glue_keys.alloc((uint64_t)c->glue_cap * W, false));
        CK(c->glue_a.alloc(c->glue_cap, false)); CK(c->glue_b.alloc(c->glue_cap, false)); CK(c->glue_conf.alloc(c->glue_cap, false));
        HIPCK(hipMemsetAsync(c->glue_keys.p, 0xFF, (uint64_t)c->glue_cap * W * sizeof(uint64_t), s));
        HIPCK(hipMemsetAsync(c->glue_a.p, 0, (uint64_t)c->glue_cap * sizeof(uint32_t), s));
        HIPCK(hipMemsetAsync(c->glue_b.p, 0, (uint64_t)c->glue_cap * sizeof(uint32_t), s));
        HIPCK(hipMemsetAsync(c->glue_conf.p, 0, (uint64_t)c->glue_cap * sizeof(uint32_t), s));
        GlueBuildParams bp{ c->glog_keys.p, c->glog_tag.p, c->n_glog, c->glue_keys.p, c->glue_a.p, c->glue_b.p, c->glue_conf.p, c->glue_cap - 1,
                            world - 1, world > 1 ? (uint32_t)c->prm.rank : 0u };
        CDBG_LAUNCH((k_glue_build<W>), std::min<uint64_t>((c->n_glog + GLUE_THREADS - 1) / GLUE_THREADS, MAX_GRID), GLUE_THREADS, s, bp);
        GlueResolveParams gp{};
        gp.keys = c->glue_keys.p; gp.a = c->glue_a.p; gp.b = c->glue_b.p; gp.conf = c->glue_conf.p;
        gp.cap = c->glue_cap; gp.W = W; gp.link = c->link.p; gp.stats = c->dstats.p;
        CDBG_LAUNCH(k_glue_resolve, std::min<uint64_t>((c->glue_cap + GLUE_THREADS - 1) / GLUE_THREADS, GLUE_RESOLVE_GRID), GLUE_THREADS, s, gp);
    }
    float ms = 0; CK(t.stop(&ms));
    CK(check_device_error(c, "glue join"));
    uint64_t gs = 0; CK(read_u64(c->dstats.p, &gs));
    c->n_join_local = gs; c->st.ms_glue = ms; c->joined = true;
    return CDBG_OK;
}

// ---- the replicated glue exchange, step by step (internal; the caller-driven variant of this API was removed in round 3:
// a context with world_size > 1 always exchanges through its transport) ----
// ---- multi-GPU exchange: the pieces and glue records of every rank are gathered (RCCL all-gather
// driven by the caller through torch.distributed; this library only copies device-to-device into / out
// of caller-provided device buffers) and merged in rank order, after which cdbg_glue runs on the union ----
int xchg_export(cdbg_ctx* c, int what, void* dst_dev, uint64_t nbytes) {
    if (!c || !dst_dev) return fail(CDBG_E_PARAM, "null argument");
    if (c->stage != 2) return fail(CDBG_E_STATE, "xchg_export before cdbg_compact");
    if (c->direct_join && what >= 4) return fail(CDBG_E_STATE, "this single-rank context keeps no junction log (CDBG_GLUE_LOG=1 keeps it)");
    const void* src = nullptr; uint64_t have = 0;
    switch (what) {
        case 0: src = c->piece_n.p; have = c->n_pieces * sizeof(uint32_t); break;
        case 1: src = c->piece_kc.p; have = c->n_pieces * sizeof(uint64_t); break;
        case 2: src = c->piece_boff.p; have = c->n_pieces * sizeof(uint64_t); break;
        case 3: src = c->piece_bases.p; have = c->n_piece_bases; break;
        case 4: src = c->glog_keys.p; have = c->n_glog * (uint64_t)c->W * sizeof(uint64_t); break;
        case 5: src = c->glog_tag.p; have = c->n_glog * sizeof(uint32_t); break;
        default: return fail(CDBG_E_PARAM, "unknown export kind %d", what);
    }
    if (nbytes < have) return fail(CDBG_E_PARAM, "export buffer too small (%llu < %llu)", (unsigned long long)nbytes, (unsigned long long)have);
    if (have) HIPCK(hipMemcpyAsync(dst_dev, src, have, hipMemcpyDeviceToDevice, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    return CDBG_OK;
}
int xchg_begin(cdbg_ctx* c, uint64_t total_pieces, uint64_t total_bases, uint64_t total_glog) {
    if (!c) return fail(CDBG_E_PARAM, "null context");
    if (c->stage != 2) return fail(CDBG_E_STATE, "xchg_begin before cdbg_compact");
    // the merged arrays are swapped with the context's own in xchg_end: give them at least the same
    // capacity, so that a re-run after cdbg_reset finds arrays that are large enough and never reallocates
    CK(c->mg_n.alloc(total_pieces, false, c->piece_n.cap)); CK(c->mg_kc.alloc(total_pieces, false, c->piece_kc.cap));
    CK(c->mg_boff.alloc(total_pieces, false, c->piece_boff.cap));
    const uint64_t bases_slack = 64ull * 4096;               // the packed exchange starts every rank's bases on a 64-byte boundary
    CK(c->mg_bases.alloc(total_bases + bases_slack, false, c->piece_bases.cap));
    CK(c->mg_gkeys.alloc(total_glog * c->W, false, c->glog_keys.cap)); CK(c->mg_gtag.alloc(total_glog, false, c->glog_tag.cap));
    if (c->prm.all_abundance_counts) CK(c->mg_ab.alloc(total_bases + bases_slack, false, c->piece_ab.cap));
    c->mg_np = c->mg_nb = c->mg_nl = 0; c->mg_cap_p = total_pieces; c->mg_cap_b = total_bases + bases_slack; c->mg_cap_l = total_glog; c->mg_open = true;
    return CDBG_OK;
}
// ---- packed variant of the exchange: bases travel as 2 bits (pieces padded to whole bytes, reservation gaps squeezed
// out) and the per-piece base offsets do not travel at all -- the receiver recomputes them from the piece lengths ----
int xchg_sizes_packed(cdbg_ctx* c, uint64_t out[4]) {
    if (!c || !out) return fail(CDBG_E_PARAM, "null argument");
    if (c->stage != 2) return fail(CDBG_E_STATE, "xchg_* needs a compacted, not yet glued context");
    if (c->direct_join) return fail(CDBG_E_STATE, "this single-rank context joined its junction records in place and keeps no log to exchange (create it with world_size > 1, or set CDBG_GLUE_LOG=1)");
    hipStream_t s = c->stream;
    const uint64_t NP = c->n_pieces;
    CK(c->xp_lens.alloc(NP, false)); CK(c->xp_uoff.alloc(NP + 1, false));
    if (NP) {
        PackLenParams lp{ NP, c->k, c->piece_n.p, c->xp_lens.p };
        CDBG_LAUNCH(k_pack_lens, (NP + 255) / 256, 256, s, lp);
    }
    CK(exscan_u32(c, c->xp_lens.p, c->xp_uoff.p, NP));
    HIPCK(hipStreamSynchronize(s));
    CK(read_u64(c->xp_uoff.p + NP, &c->xp_unpacked));
    const uint64_t chunks = (c->xp_unpacked + 63) / 64;      // 64 bases -> 16 bytes per lane
    c->xp_bytes = chunks * 16;
    CK(c->xp_dense.alloc(chunks * 64 + 64, false)); CK(c->xp_bases.alloc(c->xp_bytes + 16, false));
    if (chunks) HIPCK(hipMemsetAsync(c->xp_dense.p + (chunks - 1) * 64, 'A', 64, s));     // tail padding of the last chunk
    if (NP) {
        SqueezeParams sq{ NP, c->xp_lens.p, c->xp_uoff.p, c->piece_boff.p, c->piece_bases.p, c->xp_dense.p };
        CDBG_LAUNCH(k_squeeze_bases, (NP + 255) / 256, 256, s, sq);
    }
    if (chunks) {
        StreamPackParams pp{ chunks, c->xp_dense.p, c->xp_bases.p, c->xp_unpacked };
        CDBG_LAUNCH(k_pack_stream, (chunks + 255) / 256, 256, s, pp);
    }
    HIPCK(hipStreamSynchronize(s));
    out[0] = NP; out[1] = c->xp_unpacked; out[2] = c->n_glog; out[3] = c->xp_bytes;
    return CDBG_OK;
}
int xchg_export_packed(cdbg_ctx* c, void* dst_dev, uint64_t nbytes) {
    if (!c || !dst_dev) return fail(CDBG_E_PARAM, "null argument");
    if (c->stage != 2 || !c->xp_bases.p) return fail(CDBG_E_STATE, "xchg_export_packed before xchg_sizes_packed");
    if (nbytes < c->xp_bytes) return fail(CDBG_E_PARAM, "export buffer too small (%llu < %llu)", (unsigned long long)nbytes, (unsigned long long)c->xp_bytes);
    if (c->xp_bytes) HIPCK(hipMemcpyAsync(dst_dev, c->xp_bases.p, c->xp_bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    return CDBG_OK;
}
int xchg_add_packed(cdbg_ctx* c, uint64_t n_pieces, uint64_t n_bases, uint64_t n_packed, uint64_t n_glog, const void* piece_n, const void* piece_kc,
                             const void* packed_bases, const void* glog_keys, const void* glog_tag) {
    if (!c) return fail(CDBG_E_PARAM, "null context");
    if (!c->mg_open) return fail(CDBG_E_STATE, "xchg_add_packed without xchg_begin");
    c->mg_nb = (c->mg_nb + 63) / 64 * 64;                    // 16-byte stores of the streaming unpack
    if (c->mg_np + n_pieces > c->mg_cap_p || c->mg_nb + n_bases > c->mg_cap_b || c->mg_nl + n_glog > c->mg_cap_l)
        return fail(CDBG_E_PARAM, "xchg_add_packed exceeds the totals given to xchg_begin");
    if (2 * (c->mg_np + n_pieces) >= 0x7FFFFFF0ULL) return fail(CDBG_E_INTERNAL, "too many pieces for 31-bit end ids");
    if (n_packed != (n_bases + 63) / 64 * 16) return fail(CDBG_E_PARAM, "packed size %llu does not match %llu bases", (unsigned long long)n_packed, (unsigned long long)n_bases);
    hipStream_t s = c->stream;
    // offsets of the source rank's pieces inside its gap-free stream, recomputed here from its piece_n
    // (context members: a step must not allocate or free device memory once the buffers of the first step exist)
    DBuf<uint32_t>& lens = c->xr_lens; DBuf<uint64_t>& uoff = c->xr_uoff;
    CK(lens.alloc(n_pieces, false)); CK(uoff.alloc(n_pieces + 1, false));
    if (n_pieces) {
        PackLenParams lp{ n_pieces, c->k, (const uint32_t*)piece_n, lens.p };
        CDBG_LAUNCH(k_pack_lens, (n_pieces + 255) / 256, 256, s, lp);
    }
    CK(exscan_u32(c, lens.p, uoff.p, n_pieces));
    HIPCK(hipStreamSynchronize(s));
    uint64_t tu = 0; CK(read_u64(uoff.p + n_pieces, &tu));
    if (tu != n_bases) return fail(CDBG_E_PARAM, "piece lengths (%llu bases) do not match the packed stream (%llu bases)", (unsigned long long)tu, (unsigned long long)n_bases);
    const uint64_t chunks = (n_bases + 63) / 64;
    if (chunks) {
        StreamUnpackParams up{ chunks, (const uint8_t*)packed_bases, c->mg_bases.p + c->mg_nb, n_bases };
        CDBG_LAUNCH(k_unpack_stream, (chunks + 255) / 256, 256, s, up);
    }
    MergeParams mp{ n_pieces, n_glog, c->mg_np, c->mg_nb, c->mg_nl, c->W,
                    (const uint32_t*)piece_n, (const uint64_t*)piece_kc, uoff.p, (const uint64_t*)glog_keys, (const uint32_t*)glog_tag,
                    c->mg_n.p, c->mg_kc.p, c->mg_boff.p, c->mg_gkeys.p, c->mg_gtag.p };
    const uint64_t work = std::max(n_pieces, n_glog);
    if (work) CDBG_LAUNCH(k_merge_append, std::min<uint64_t>((work + 255) / 256, MAX_GRID), 256, s, mp);
    HIPCK(hipStreamSynchronize(s));
    c->last_add_np = c->mg_np; c->last_add_nb = c->mg_nb; c->last_add_pieces = n_pieces;
    c->mg_np += n_pieces; c->mg_nb += n_bases; c->mg_nl += n_glog;
    return CDBG_OK;
}
// -all-abundance-counts: the abundances of this rank's pieces as a gap-free stream, one u32 per k-mer, in the piece
// order of xchg_sizes_packed
int xchg_abundance_values(cdbg_ctx* c, uint64_t* n_values) {
    if (!c || !n_values) return fail(CDBG_E_PARAM, "null argument");
    if (!c->prm.all_abundance_counts) return fail(CDBG_E_STATE, "context was created without all_abundance_counts");
    if (c->stage != 2) return fail(CDBG_E_STATE, "xchg_* needs a compacted, not yet glued context");
    CK(c->xp_aoff.alloc(c->n_pieces + 1, false));
    CK(exscan_u32(c, c->piece_n.p, c->xp_aoff.p, c->n_pieces));
    HIPCK(hipStreamSynchronize(c->stream));
    CK(read_u64(c->xp_aoff.p + c->n_pieces, &c->xp_nab));
    *n_values = c->xp_nab; c->xp_ab_ready = true;
    return CDBG_OK;
}
int xchg_export_abundances(cdbg_ctx* c, void* dst_dev, uint64_t nbytes) {
    if (!c || !dst_dev) return fail(CDBG_E_PARAM, "null argument");
    if (c->stage != 2 || !c->xp_ab_ready) return fail(CDBG_E_STATE, "xchg_export_abundances before xchg_abundance_values");
    const uint64_t NP = c->n_pieces;
    if (nbytes < c->xp_nab * sizeof(uint32_t)) return fail(CDBG_E_PARAM, "export buffer too small (%llu < %llu)", (unsigned long long)nbytes, (unsigned long long)(c->xp_nab * sizeof(uint32_t)));
    if (NP) {
        AbStreamParams ap{ NP, c->k, 0, c->piece_n.p, c->xp_aoff.p, nullptr, c->piece_boff.p, 0, c->piece_ab.p, (uint32_t*)dst_dev };
        CDBG_LAUNCH(k_ab_stream, (NP + 255) / 256, 256, c->stream, ap);
    }
    HIPCK(hipStreamSynchronize(c->stream));
    return CDBG_OK;
}
// ... and the stream of the rank whose pieces the latest xchg_add_packed appended
int xchg_add_abundances(cdbg_ctx* c, const void* ab_stream, uint64_t n_values) {
    if (!c || !ab_stream) return fail(CDBG_E_PARAM, "null argument");
    if (!c->prm.all_abundance_counts) return fail(CDBG_E_STATE, "context was created without all_abundance_counts");
    if (!c->mg_open) return fail(CDBG_E_STATE, "xchg_add_abundances without xchg_begin");
    const uint64_t NP = c->last_add_pieces;
    CK(c->xr_aoff.alloc(NP + 1, false));
    CK(exscan_u32(c, c->mg_n.p + c->last_add_np, c->xr_aoff.p, NP));
    HIPCK(hipStreamSynchronize(c->stream));
    uint64_t tot = 0; CK(read_u64(c->xr_aoff.p + NP, &tot));
    if (n_values != tot) return fail(CDBG_E_PARAM, "abundance stream of %llu values does not match the %llu k-mers of the pieces added last", (unsigned long long)n_values, (unsigned long long)tot);
    if (NP) {
        AbStreamParams ap{ NP, c->k, 1, c->mg_n.p + c->last_add_np, c->xr_aoff.p, c->xr_uoff.p, nullptr, c->last_add_nb, c->mg_ab.p, (uint32_t*)ab_stream };
        CDBG_LAUNCH(k_ab_stream, (NP + 255) / 256, 256, c->stream, ap);
    }
    HIPCK(hipStreamSynchronize(c->stream));
    return CDBG_OK;
}
int xchg_end(cdbg_ctx* c) {
    if (!c) return fail(CDBG_E_PARAM, "null context");
    if (!c->mg_open) return fail(CDBG_E_STATE, "xchg_end without xchg_begin");
    c->piece_n.swap(c->mg_n); c->piece_kc.swap(c->mg_kc); c->piece_boff.swap(c->mg_boff);
    c->piece_bases.swap(c->mg_bases); c->glog_keys.swap(c->mg_gkeys); c->glog_tag.swap(c->mg_gtag);
    if (c->prm.all_abundance_counts) c->piece_ab.swap(c->mg_ab);
    c->n_pieces = c->mg_np; c->n_piece_bases = c->mg_nb; c->n_glog = c->mg_nl; c->glog_cap = c->mg_cap_l;
    c->mg_open = false;
    HIPCK(hipStreamSynchronize(c->stream));
    return CDBG_OK;
}

// ---- multi-GPU: sharded junction join.  After xchg_end every rank holds the union of the glue records;
// instead of every rank joining all of them, xchg_glue_join joins this rank's share of the junctions, the caller
// MAX-all-reduces the int32 link arrays (cdbg_glue_links_export / _import) and cdbg_glue then ranks and emits ----
int xchg_glue_join(cdbg_ctx* c, uint64_t* n_ends) {
    if (!c || !n_ends) return fail(CDBG_E_PARAM, "null argument");
    if (c->stage != 2) return fail(CDBG_E_STATE, "xchg_glue_join needs a compacted, not yet glued context");
    int rc;
    switch (c->W) { case 1: rc = glue_join_impl<1>(c, true); break; case 2: rc = glue_join_impl<2>(c, true); break; case 3: rc = glue_join_impl<3>(c, true); break; default: rc = glue_join_impl<4>(c, true); }
    if (rc == CDBG_OK) *n_ends = 2 * c->n_pieces;
    return rc;
}

// ---- multi-GPU glue exchange, driven by the library through the context's transport: every rank's pieces (lengths,
// abundance sums, bases packed 4 per byte, no offsets) and junction log are all-gathered and merged in rank order
// (xchg_*); the junction hash-join is sharded by key hash and its result, one partner id per piece end, is
// combined with ONE MAX all-reduce (every end is set by exactly one rank) ----
int glue_exchange(cdbg_ctx* c) {
    const int world = c->prm.world_size, W = c->W;
    hipStream_t s = c->stream;
    Timer t; CK(t.start(s));
    uint64_t mine[4]; CK(xchg_sizes_packed(c, mine));          // pieces, bases once unpacked, glue-log records, packed bytes
    std::vector<uint64_t> all((size_t)world * 4);
    if (c->tr.all_gather_u64(c->tr.user, mine, all.data(), 4) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_u64 failed");
    auto col = [&](int r, int j) { return all[(size_t)r * 4 + j]; };
    // the five arrays: piece_n (u32), piece_kc (u64), packed bases, glue keys (u64 x W), glue tags (u32)
    const uint64_t item[5] = { 4, 8, 1, 8ull * W, 4 }; const int which[5] = { 0, 0, 3, 2, 2 };
    std::vector<std::vector<uint64_t>> roff(5, std::vector<uint64_t>(world)), rcnt(5, std::vector<uint64_t>(world));
    DBuf<uint8_t>& sendbuf = c->xsend;
    for (int a = 0; a < 5; ++a) {
        uint64_t tot = 0;
        for (int r = 0; r < world; ++r) { rcnt[a][r] = col(r, which[a]) * item[a]; roff[a][r] = tot; tot += (rcnt[a][r] + 15) / 16 * 16; }
        CK(c->xg[a].alloc(tot + 16, false));
        const uint64_t nb = rcnt[a][c->prm.rank];
        CK(sendbuf.alloc(nb + 16, false));
        if (a == 2) CK(xchg_export_packed(c, sendbuf.p, nb + 16));
        else CK(xchg_export(c, a == 0 ? 0 : a == 1 ? 1 : a == 3 ? 4 : 5, sendbuf.p, nb + 16));
        if (c->tr.all_gather_v(c->tr.user, sendbuf.p, nb, c->xg[a].p, roff[a].data(), rcnt[a].data()) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_v failed");
        for (int r = 0; r < world; ++r) if (r != c->prm.rank) c->comm_bytes += nb + rcnt[a][r];
    }
    uint64_t tp = 0, tb = 0, tl = 0;
    for (int r = 0; r < world; ++r) { if (r == c->prm.rank) c->piece_lo = tp; tp += col(r, 0); if (r == c->prm.rank) c->piece_hi = tp; tb += col(r, 1); tl += col(r, 2); }
    // -all-abundance-counts: a sixth array, one u32 per k-mer of the rank's pieces
    std::vector<uint64_t> aoff(world), acnt(world);
    if (c->prm.all_abundance_counts) {
        uint64_t nv = 0; CK(xchg_abundance_values(c, &nv));
        std::vector<uint64_t> allv(world);
        if (c->tr.all_gather_u64(c->tr.user, &nv, allv.data(), 1) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_u64 failed");
        uint64_t tot = 0;
        for (int r = 0; r < world; ++r) { acnt[r] = allv[r] * 4; aoff[r] = tot; tot += (acnt[r] + 15) / 16 * 16; }
        CK(c->xp_ab.alloc(tot / 4 + 4, false));
        const uint64_t nb = acnt[c->prm.rank];
        CK(sendbuf.alloc(nb + 16, false));
        CK(xchg_export_abundances(c, sendbuf.p, nb + 16));
        if (c->tr.all_gather_v(c->tr.user, sendbuf.p, nb, c->xp_ab.p, aoff.data(), acnt.data()) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_v failed");
        for (int r = 0; r < world; ++r) if (r != c->prm.rank) c->comm_bytes += nb + acnt[r];
    }
    CK(xchg_begin(c, tp, tb, tl));
    for (int r = 0; r < world; ++r) {
        CK(xchg_add_packed(c, col(r, 0), col(r, 1), col(r, 3), col(r, 2), c->xg[0].p + roff[0][r], c->xg[1].p + roff[1][r],
                                    c->xg[2].p + roff[2][r], c->xg[3].p + roff[3][r], c->xg[4].p + roff[4][r]));
        if (c->prm.all_abundance_counts) CK(xchg_add_abundances(c, (const uint8_t*)c->xp_ab.p + aoff[r], acnt[r] / 4));
    }
    CK(xchg_end(c));
    // sharded junction join
    uint64_t n_ends = 0; CK(xchg_glue_join(c, &n_ends));
    if (c->tr.all_reduce_max_i32(c->tr.user, c->link.p, n_ends) != 0) return fail(CDBG_E_INTERNAL, "transport all_reduce_max_i32 failed");
    c->comm_bytes += 2 * n_ends * 4 * (uint64_t)(world - 1) / (uint64_t)world;          // (ring all-reduce volume per rank)
    float ms = 0; CK(t.stop(&ms)); c->st.ms_exchange += ms;
    c->xchg_done = true;
    return CDBG_OK;
}


// =======================================================================================
// Sharded glue across ranks (k_dglue.h): join by key owner, ranking by piece owner, emission by head owner.
// Returns DG_FALLBACK (every rank, together) when the distributed ranking does not converge -- closed chains that cross
// ranks -- and the caller then runs the replicated exchange, which can cut cycles.
// =======================================================================================
// The stage's end state: the unitig arena at 2 bits per base next to the ASCII one (one streaming pass:
// 64 bases -> 16 bytes per lane; base i of the arena = bits [2 (i & 3), 2 (i & 3) + 2) of byte i >> 2, A0 C1 G2 T3)
int pack_unitigs(cdbg_ctx* c) {
    const uint64_t chunks = (c->unitig_total + 63) / 64;
    CK(c->unitig_packed.alloc(chunks * 16 + 16, false));
    if (chunks) { StreamPackParams pp{ chunks, c->unitig_bases.p, c->unitig_packed.p, c->unitig_total }; CDBG_LAUNCH(k_pack_stream, (chunks + 255) / 256, 256, c->stream, pp); }
    return CDBG_OK;
}
constexpr int DG_FALLBACK = 1;
struct DgRoute { std::vector<uint64_t> scnt, soff, rcnt, roff, all; uint64_t n_send = 0, n_recv = 0, n_all = 0; };
// positions of the n items whose destinations are in c->dg_dest: send-block counts / offsets, receive counts / offsets
// (extra: n_extra more words of this rank ride in the same small all-gather -- extra_all[r * n_extra + i] = word i of rank r: the
//  piece counts and the status words that used to cost a host-synchronous collective of their own)
int dg_route(cdbg_ctx* c, uint64_t n, DgRoute& R, const uint64_t* extra = nullptr, int n_extra = 0, std::vector<uint64_t>* extra_all = nullptr) {
    const int world = c->prm.world_size, me = c->prm.rank; hipStream_t s = c->stream;
    CK(c->dg_cnt.alloc(3 * DG_MAX_WORLD, true)); CK(c->dg_pos.alloc(n, false));
    RouteParams rp{ n, c->dg_dest.p, c->dg_cnt.p, c->dg_cnt.p + DG_MAX_WORLD, c->dg_cnt.p + 2 * DG_MAX_WORLD, c->dg_pos.p, world };
    if (n) CDBG_LAUNCH(k_route_count, std::min<uint64_t>((n + DG_THREADS - 1) / DG_THREADS, 256 * 8), DG_THREADS, s, rp);
    R.scnt.assign(world, 0); R.soff.assign(world + 1, 0); R.rcnt.assign(world, 0); R.roff.assign(world + 1, 0); R.all.assign((size_t)world * world, 0);
    CK(read_u64(c->dg_cnt.p, R.scnt.data(), world));
    for (int d = 0; d < world; ++d) R.soff[d + 1] = R.soff[d] + R.scnt[d];
    R.n_send = R.soff[world];
    HIPCK(hipMemcpy(c->dg_cnt.p + DG_MAX_WORLD, R.soff.data(), world * sizeof(uint64_t), hipMemcpyHostToDevice));
    if (n) CDBG_LAUNCH(k_route_place, std::min<uint64_t>((n + DG_THREADS * DG_ITEMS - 1) / (DG_THREADS * DG_ITEMS), 256 * 8), DG_THREADS, s, rp);
    {
        const int row = world + n_extra;
        std::vector<uint64_t> mine(row), got((size_t)row * world);
        for (int d = 0; d < world; ++d) mine[d] = R.scnt[d];
        for (int i = 0; i < n_extra; ++i) mine[world + i] = extra[i];
        if (c->tr.all_gather_u64(c->tr.user, mine.data(), got.data(), row) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_u64 failed");
        for (int r = 0; r < world; ++r) {
            for (int d = 0; d < world; ++d) R.all[(size_t)r * world + d] = got[(size_t)r * row + d];
            if (extra_all) for (int i = 0; i < n_extra; ++i) (*extra_all)[(size_t)r * n_extra + i] = got[(size_t)r * row + world + i];
        }
    }
    R.n_all = 0;
    for (int r = 0; r < world; ++r) { R.rcnt[r] = R.all[(size_t)r * world + me]; for (int d = 0; d < world; ++d) R.n_all += R.all[(size_t)r * world + d]; }
    for (int r = 0; r < world; ++r) R.roff[r + 1] = R.roff[r] + R.rcnt[r];
    R.n_recv = R.roff[world];
    return CDBG_OK;
}
// all-to-all-v of fixed-size items laid out by dg_route (reverse = true: the replies travel back along the same blocks)
int dg_a2a(cdbg_ctx* c, const void* send, void* recv, const DgRoute& R, uint64_t item, bool reverse = false) {
    const int world = c->prm.world_size, me = c->prm.rank;
    std::vector<uint64_t> so(world), sc(world), ro(world), rc(world);
    for (int r = 0; r < world; ++r) {
        const uint64_t* sof = reverse ? R.roff.data() : R.soff.data(); const uint64_t* scn = reverse ? R.rcnt.data() : R.scnt.data();
        const uint64_t* rof = reverse ? R.soff.data() : R.roff.data(); const uint64_t* rcn = reverse ? R.scnt.data() : R.rcnt.data();
        so[r] = sof[r] * item; sc[r] = scn[r] * item; ro[r] = rof[r] * item; rc[r] = rcn[r] * item;
        if (r != me) c->comm_bytes += sc[r] + rc[r];
    }
    if (!c->tr_ordered) HIPCK(hipStreamSynchronize(c->stream));   // (a caller-supplied transport reads the buffers from the host side)
    if (c->tr.all_to_all_v(c->tr.user, send, so.data(), sc.data(), recv, ro.data(), rc.data()) != 0) return fail(CDBG_E_INTERNAL, "transport all_to_all_v failed");
    return CDBG_OK;
}
// all-to-all-v of variable-size blocks: send_bytes[d] at send_off[d]; the receive sizes are exchanged first
int dg_a2a_blocks(cdbg_ctx* c, const void* send, const std::vector<uint64_t>& send_off, const std::vector<uint64_t>& send_bytes,
                  std::vector<uint64_t>& recv_off, std::vector<uint64_t>& recv_bytes, DBuf<uint8_t>& recv, uint64_t align) {
    const int world = c->prm.world_size, me = c->prm.rank;
    std::vector<uint64_t> all((size_t)world * world);
    if (c->tr.all_gather_u64(c->tr.user, send_bytes.data(), all.data(), world) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_u64 failed");
    recv_off.assign(world + 1, 0); recv_bytes.assign(world, 0);
    for (int r = 0; r < world; ++r) { recv_bytes[r] = all[(size_t)r * world + me]; recv_off[r + 1] = recv_off[r] + (recv_bytes[r] + align - 1) / align * align; if (r != me) c->comm_bytes += send_bytes[r] + recv_bytes[r]; }
    CK(recv.alloc(recv_off[world] + align, false));
    if (!c->tr_ordered) HIPCK(hipStreamSynchronize(c->stream));   // (a caller-supplied transport reads the buffers from the host side)
    if (c->tr.all_to_all_v(c->tr.user, send, send_off.data(), send_bytes.data(), recv.p, recv_off.data(), recv_bytes.data()) != 0) return fail(CDBG_E_INTERNAL, "transport all_to_all_v failed");
    return CDBG_OK;
}

template <int W>
int glue_sharded(cdbg_ctx* c) {
    const int world = c->prm.world_size, me = c->prm.rank, k = c->k;
    hipStream_t s = c->stream;
    if (world > DG_MAX_WORLD) return fail(CDBG_E_PARAM, "sharded glue supports up to %d ranks", DG_MAX_WORLD);
    Timer t; CK(t.start(s));
    const uint64_t NP = c->n_pieces;
    auto grid = [](uint64_t n) { return (uint32_t)std::min<uint64_t>(std::max<uint64_t>((n + 255) / 256, 1), 256 * 16); };
    DgOwners own{}; own.world = world;
    const uint32_t NSl = (uint32_t)(2 * NP);
    uint32_t end_base = 0;
    HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
    HIPCK(hipMemsetAsync(c->derr.p, 0, 4 * sizeof(uint32_t), s));
    // ---- 1a. junction records to their key owners (the piece id ranges of the ranks ride in the routing's count exchange) ----
    DgRoute R;
    {
        const uint64_t n = c->n_glog;
        CK(c->dg_dest.alloc(std::max<uint64_t>(std::max<uint64_t>(n, NSl), NP) + 1, false));
        LogRouteParams lp{ c->glog_keys.p, c->glog_tag.p, n, world, 0, c->dg_dest.p, nullptr, nullptr };
        if (n) CDBG_LAUNCH((k_log_dest<W>), grid(n), 256, s, lp);
        std::vector<uint64_t> all(world);
        CK(dg_route(c, n, R, &NP, 1, &all));
        uint64_t tot = 0; for (int r = 0; r < world; ++r) { own.b[r] = (uint32_t)tot; tot += all[r]; }
        if (2 * tot >= 0x7FFFFFF0ULL) return fail(CDBG_E_INTERNAL, "too many pieces for 31-bit end ids (%llu): use more partitions per GPU or fewer reads", (unsigned long long)tot);   // (every rank sees the same total)
        own.b[world] = (uint32_t)tot; c->piece_lo = own.b[me]; c->piece_hi = own.b[me + 1];
        end_base = 2u * own.b[me]; lp.end_base = end_base;
        CK(c->dg_wire_s.alloc(R.n_send * (W + 1) + 1, false)); CK(c->dg_wire_r.alloc(R.n_recv * (W + 1) + 1, false));
        lp.pos = c->dg_pos.p; lp.wire = c->dg_wire_s.p;
        if (n) CDBG_LAUNCH((k_log_write<W>), grid(n), 256, s, lp);
        CK(dg_a2a(c, c->dg_wire_s.p, c->dg_wire_r.p, R, (uint64_t)(W + 1) * 8));
    }
    // ---- join this rank's keys; joined pairs to the end owners ----
    uint64_t n_pairs = 0; uint32_t join_err = 0;
    {
        const uint64_t n = R.n_recv;
        int log_jb = 0; while (((uint64_t)(JB_CAP / 2) << log_jb) < n && log_jb < 26) ++log_jb;
        const uint64_t JB = 1ull << log_jb;
        CK(c->jfill.alloc(JB, false)); CK(c->jrecs.alloc(JB * JB_CAP * (W + 1), false));
        HIPCK(hipMemsetAsync(c->jfill.p, 0, JB * sizeof(uint32_t), s));
        HIPCK(hipMemsetAsync(c->cursors.p + 7, 0, sizeof(uint64_t), s));
        WireScatterParams wp{ c->dg_wire_r.p, n, log_jb, c->jfill.p, c->jrecs.p, c->derr.p };
        if (n) CDBG_LAUNCH((k_join_scatter_wire<W>), grid(n), 256, s, wp);
        // (pair list: <= n entries used, in per-wave chunks whose tails stay unused -- pre-filled with the 'no pair' marker)
        // (+ n / 8: a wave also abandons the rest of its current chunk whenever a bucket's pairs do not fit into it)
        const uint64_t pair_cap = n + n / 8 + 2 + (uint64_t)JB_PAIR_CHUNK * std::min<uint64_t>((JB + 3) / 4, 256 * 16) * (JB_THREADS / 64);
        CK(c->dg_pairs.alloc(pair_cap, false));
        HIPCK(hipMemsetAsync(c->dg_pairs.p, 0xFF, pair_cap * sizeof(uint2), s));
        JoinBucketParams bp{ c->jfill.p, c->jrecs.p, (uint32_t)JB, nullptr, c->dstats.p, c->dg_pairs.p, c->cursors.p + 7, pair_cap, c->derr.p };
        CDBG_LAUNCH((k_join_bucket<W>), std::min<uint64_t>((JB + 3) / 4, 256 * 16), JB_THREADS, s, bp);
        CK(read_u32(c->derr.p, &join_err));
        CK(read_u64(c->cursors.p + 7, &n_pairs));
        if (join_err || n_pairs > pair_cap) n_pairs = 0;     // (nothing of a failed join travels; the ranks agree on what happens below)
        uint64_t gs = 0; CK(read_u64(c->dstats.p, &gs)); c->n_join_local = gs;
    }
    {
        PairRouteParams pp{ c->dg_pairs.p, n_pairs, own, c->dg_dest.p, nullptr, nullptr };
        CK(c->dg_dest.alloc(std::max<uint64_t>(std::max<uint64_t>(n_pairs, NSl), NP) + 1, false)); pp.dest = c->dg_dest.p;
        if (n_pairs) CDBG_LAUNCH(k_pair_dest, grid(n_pairs), 256, s, pp);
        // the status of every rank's join rides in the routing's count exchange: a bucket overflow (8) stops all ranks together, a
        // pair list that did not fit (9: nearly every record joined and the chunk tails ate the headroom) sends all of them to
        // the replicated exchange
        const uint64_t stw = join_err; std::vector<uint64_t> sts(world);
        CK(dg_route(c, n_pairs, R, &stw, 1, &sts));
        bool any9 = false;
        for (int r = 0; r < world; ++r) {
            if (sts[r] == 9) any9 = true;
            else if (sts[r]) return fail(CDBG_E_INTERNAL, "sharded junction join: rank %d reported device error %llu (8 bucket overflow); all ranks stop", r, (unsigned long long)sts[r]);
        }
        if (any9) { HIPCK(hipMemsetAsync(c->derr.p, 0, 4 * sizeof(uint32_t), s)); float ms = 0; CK(t.stop(&ms)); c->st.ms_exchange += ms; return DG_FALLBACK; }
        CK(c->dg_pair_s.alloc(R.n_send + 1, false)); CK(c->dg_pair_r.alloc(R.n_recv + 1, false));
        pp.pos = c->dg_pos.p; pp.wire = c->dg_pair_s.p;
        if (n_pairs) CDBG_LAUNCH(k_pair_write, grid(n_pairs), 256, s, pp);
        CK(dg_a2a(c, c->dg_pair_s.p, c->dg_pair_r.p, R, sizeof(uint2)));
        CK(c->link.alloc(NSl, false));
        HIPCK(hipMemsetAsync(c->link.p, 0xFF, (size_t)std::max<uint32_t>(NSl, 1) * sizeof(uint32_t), s));
        PairApplyParams ap{ c->dg_pair_r.p, R.n_recv, end_base, c->link.p };
        if (R.n_recv) CDBG_LAUNCH(k_pair_apply, grid(R.n_recv), 256, s, ap);
    }
    // ---- 2. distributed list ranking ----
    CK(c->rank_a.alloc(NSl, false)); CK(c->rank_b.alloc(NSl, false));
    uint2* const st = reinterpret_cast<uint2*>(c->rank_b.p);
    DRankParams dp{}; dp.n_local = NSl; dp.base = end_base; dp.own = own; dp.me = me; dp.link = c->link.p; dp.piece_n = c->piece_n.p; dp.st = st; dp.dest = c->dg_dest.p;
    const uint32_t gridS = std::max<uint32_t>((NSl + 255) / 256, 1);
    if (NSl) CDBG_LAUNCH(k_dr_init, gridS, 256, s, dp);
    {
        uint64_t total_states = 2ull * own.b[world];
        int max_rounds = 4; while ((1ull << (max_rounds - 3)) < total_states) ++max_rounds;
        bool done = false;
        for (int round = 0; round < max_rounds; ++round) {
            if (NSl) CDBG_LAUNCH(k_dr_jump, gridS, 256, s, dp);
            CK(dg_route(c, NSl, R));
            if (R.n_all == 0) { done = true; break; }                       // no rank has an unfinished state
            ++c->st.n_glue_rounds;
            CK(c->dg_qs.alloc(R.n_send + 1, false)); CK(c->dg_qsrc.alloc(R.n_send + 1, false)); CK(c->dg_qr.alloc(R.n_recv + 1, false));
            CK(c->dg_rs.alloc(R.n_recv + 1, false)); CK(c->dg_rr.alloc(R.n_send + 1, false));
            dp.pos = c->dg_pos.p; dp.q_send = c->dg_qs.p; dp.q_src = c->dg_qsrc.p; dp.q_recv = c->dg_qr.p; dp.r_send = c->dg_rs.p; dp.n_recv = R.n_recv;
            dp.r_recv = c->dg_rr.p; dp.n_sent = R.n_send;
            if (NSl) CDBG_LAUNCH(k_dr_query, gridS, 256, s, dp);
            CK(dg_a2a(c, c->dg_qs.p, c->dg_qr.p, R, sizeof(uint32_t)));
            if (R.n_recv) CDBG_LAUNCH(k_dr_reply, grid(R.n_recv), 256, s, dp);
            CK(dg_a2a(c, c->dg_rs.p, c->dg_rr.p, R, sizeof(uint2), true));
            if (R.n_send) CDBG_LAUNCH(k_dr_apply, grid(R.n_send), 256, s, dp);
        }
        if (!done) { float ms = 0; CK(t.stop(&ms)); c->st.ms_exchange += ms; c->st.n_glue_rounds = 0; return DG_FALLBACK; }   // closed chains across ranks
    }
    // ---- 3. heads of this rank, then every piece to the owner of its head ----
    uint64_t hm[2] = {0, 0};
    {
        HIPCK(hipMemsetAsync(c->dstats.p + 8, 0, 2 * sizeof(uint64_t), s));
        HeadMeasureParams mp{ NSl, k, c->link.p, st, c->dstats.p + 8 };
        if (NSl) CDBG_LAUNCH(k_heads_measure, grid(NSl), 256, s, mp);
        CK(read_u64(c->dstats.p + 8, hm, 2));
    }
    const uint64_t ucap = std::max<uint64_t>(hm[0], 1), ocap = std::max<uint64_t>(hm[1], 1);
    CK(c->unitig_off.alloc(ucap, false)); CK(c->unitig_len.alloc(ucap, false)); CK(c->unitig_kc.alloc(ucap, false));
    CK(c->unitig_bases.alloc(ocap + 64, false));
    if (c->prm.all_abundance_counts) CK(c->unitig_ab.alloc(ocap + 64, false));
    HIPCK(hipMemsetAsync(c->cursors.p + 2, 0, 2 * sizeof(uint64_t), s));
    HeadParams hp{};
    hp.n_states = NSl; hp.k = k; hp.link = c->link.p; hp.st = st; hp.hinfo = c->rank_a.p;
    hp.unitig_off = c->unitig_off.p; hp.unitig_len = c->unitig_len.p; hp.unitig_kc = c->unitig_kc.p;
    hp.unitig_cap = ucap; hp.out_cap = ocap; hp.n_unitigs = c->cursors.p + 2; hp.out_cursor = c->cursors.p + 3; hp.error = c->derr.p; hp.own_lo = 0; hp.own_hi = NSl;
    if (NSl) CDBG_LAUNCH(k_unitig_heads, (NSl + HEADS_PER_WG - 1) / HEADS_PER_WG, GLUE_THREADS, s, hp);
    uint64_t NR = 0;
    {
        PieceRouteParams pr{}; pr.n_pieces = (uint32_t)NP; pr.k = k; pr.own = own; pr.st = st; pr.piece_n = c->piece_n.p; pr.piece_kc = c->piece_kc.p; pr.piece_boff = c->piece_boff.p; pr.dest = c->dg_dest.p;
        const uint32_t gridP = std::max<uint32_t>((uint32_t)((NP + 255) / 256), 1);
        if (NP) CDBG_LAUNCH(k_piece_dest, gridP, 256, s, pr);
        CK(dg_route(c, NP, R));
        const uint64_t ns = R.n_send; NR = R.n_recv;
        CK(c->dg_meta_s.alloc(3 * ns + 3, false)); CK(c->dg_lens.alloc(ns + 1, false)); CK(c->dg_boff.alloc(ns + 1, false)); CK(c->dg_alen.alloc(ns + 1, false));
        pr.pos = c->dg_pos.p; pr.meta = c->dg_meta_s.p; pr.lens = c->dg_lens.p; pr.boff = c->dg_boff.p; pr.alen = c->dg_alen.p;
        if (NP) CDBG_LAUNCH(k_piece_write, gridP, 256, s, pr);
        CK(c->dg_meta_r.alloc(3 * NR + 3, false));
        CK(dg_a2a(c, c->dg_meta_s.p, c->dg_meta_r.p, R, 24));
        // the bases of every destination's pieces as one gap-free stream, 2 bits per base (whole 64-base chunks)
        CK(c->dg_uoff.alloc(ns + world + 1, false));
        std::vector<uint64_t> dbase(world + 1, 0), dtot(world, 0), sbytes(world), soffb(world);
        for (int d = 0; d < world; ++d) {
            CK(exscan_u32(c, c->dg_lens.p + R.soff[d], c->dg_uoff.p + R.soff[d] + d, R.scnt[d]));
            CK(read_u64(c->dg_uoff.p + R.soff[d] + d + R.scnt[d], &dtot[d]));
            dbase[d + 1] = dbase[d] + (dtot[d] + 63) / 64 * 64;
        }
        CK(c->dg_dense.alloc(dbase[world] + 64, false)); CK(c->dg_packed.alloc(dbase[world] / 4 + 16, false));
        if (dbase[world]) HIPCK(hipMemsetAsync(c->dg_dense.p, 'A', dbase[world], s));
        for (int d = 0; d < world; ++d) {
            if (!R.scnt[d]) { sbytes[d] = 0; soffb[d] = dbase[d] / 4; continue; }
            SqueezeParams sq{ R.scnt[d], c->dg_lens.p + R.soff[d], c->dg_uoff.p + R.soff[d] + d, c->dg_boff.p + R.soff[d], c->piece_bases.p, c->dg_dense.p + dbase[d] };
            CDBG_LAUNCH(k_squeeze_bases, (R.scnt[d] + 255) / 256, 256, s, sq);
            sbytes[d] = (dtot[d] + 63) / 64 * 16; soffb[d] = dbase[d] / 4;
        }
        const uint64_t chunks = dbase[world] / 64;
        if (chunks) { StreamPackParams pp{ chunks, c->dg_dense.p, c->dg_packed.p, dbase[world] }; CDBG_LAUNCH(k_pack_stream, (chunks + 255) / 256, 256, s, pp); }
        std::vector<uint64_t> roffb, rbytes;
        CK(dg_a2a_blocks(c, c->dg_packed.p, soffb, sbytes, roffb, rbytes, c->dg_rpacked, 16));
        // receiver: metas -> piece arrays, packed streams -> ASCII
        CK(c->dg_rn.alloc(NR + 1, false)); CK(c->dg_rkc.alloc(NR + 1, false)); CK(c->dg_rst.alloc(NR + 1, false)); CK(c->dg_rlens.alloc(NR + 1, false)); CK(c->dg_rboff.alloc(NR + 2, false));
        PieceRecvParams rv{ NR, k, end_base, c->dg_meta_r.p, c->dg_rn.p, c->dg_rkc.p, c->dg_rst.p, c->dg_rlens.p };
        if (NR) CDBG_LAUNCH(k_piece_recv, grid(NR), 256, s, rv);
        std::vector<uint64_t> rbase(world + 1, 0), rtot(world, 0);
        for (int r = 0; r < world; ++r) {
            CK(exscan_u32(c, c->dg_rlens.p + R.roff[r], c->dg_rboff.p + R.roff[r], R.rcnt[r]));
            CK(read_u64(c->dg_rboff.p + R.roff[r] + R.rcnt[r], &rtot[r]));
            if ((rtot[r] + 63) / 64 * 16 != rbytes[r]) return fail(CDBG_E_INTERNAL, "sharded glue: rank %d sent %llu packed bytes for %llu bases", r, (unsigned long long)rbytes[r], (unsigned long long)rtot[r]);
            rbase[r + 1] = rbase[r] + (rtot[r] + 63) / 64 * 64;
        }
        CK(c->dg_rdense.alloc(rbase[world] + 64, false));
        for (int r = 0; r < world; ++r) {
            if (!R.rcnt[r]) continue;
            if (rbase[r]) CDBG_LAUNCH(k_add_u64, (R.rcnt[r] + 255) / 256, 256, s, c->dg_rboff.p + R.roff[r], R.rcnt[r], rbase[r]);
            const uint64_t ch = (rtot[r] + 63) / 64;
            if (ch) { StreamUnpackParams up{ ch, c->dg_rpacked.p + roffb[r], c->dg_rdense.p + rbase[r], rtot[r] }; CDBG_LAUNCH(k_unpack_stream, (ch + 255) / 256, 256, s, up); }
        }
        // -all-abundance-counts: one u32 per k-mer of every piece, same routing
        if (c->prm.all_abundance_counts) {
            CK(c->dg_aoff.alloc(ns + world + 1, false));
            std::vector<uint64_t> abase(world + 1, 0), atot(world, 0), ab_sb(world), ab_so(world);
            for (int d = 0; d < world; ++d) {
                CK(exscan_u32(c, c->dg_alen.p + R.soff[d], c->dg_aoff.p + R.soff[d] + d, R.scnt[d]));
                CK(read_u64(c->dg_aoff.p + R.soff[d] + d + R.scnt[d], &atot[d]));
                abase[d + 1] = abase[d] + (atot[d] + 3) / 4 * 4;
            }
            CK(c->dg_ab_s.alloc(abase[world] + 4, false));
            for (int d = 0; d < world; ++d) {
                ab_sb[d] = atot[d] * 4; ab_so[d] = abase[d] * 4;
                if (!R.scnt[d]) continue;
                AbStreamParams ap{ R.scnt[d], k, 0, c->dg_alen.p + R.soff[d], c->dg_aoff.p + R.soff[d] + d, nullptr, c->dg_boff.p + R.soff[d], 0, c->piece_ab.p, c->dg_ab_s.p + abase[d] };
                CDBG_LAUNCH(k_ab_stream, (R.scnt[d] + 255) / 256, 256, s, ap);
            }
            std::vector<uint64_t> ab_ro, ab_rb; DBuf<uint8_t>& rbuf = c->xsend;
            CK(dg_a2a_blocks(c, c->dg_ab_s.p, ab_so, ab_sb, ab_ro, ab_rb, rbuf, 16));
            CK(c->dg_rab.alloc(rbase[world] + 64, false)); CK(c->dg_raoff.alloc(NR + 2, false));
            for (int r = 0; r < world; ++r) {
                if (!R.rcnt[r]) continue;
                CK(exscan_u32(c, c->dg_rn.p + R.roff[r], c->dg_raoff.p + R.roff[r], R.rcnt[r]));
                uint64_t tot = 0; CK(read_u64(c->dg_raoff.p + R.roff[r] + R.rcnt[r], &tot));
                if (tot * 4 != ab_rb[r]) return fail(CDBG_E_INTERNAL, "sharded glue: abundance stream of rank %d does not match its pieces", r);
                AbStreamParams ap{ R.rcnt[r], k, 1, c->dg_rn.p + R.roff[r], c->dg_raoff.p + R.roff[r], nullptr, c->dg_rboff.p + R.roff[r], 0, c->dg_rab.p, reinterpret_cast<uint32_t*>(rbuf.p + ab_ro[r]) };
                CDBG_LAUNCH(k_ab_stream, (R.rcnt[r] + 255) / 256, 256, s, ap);
                HIPCK(hipStreamSynchronize(s));                             // (the next source's prefix sums reuse dg_raoff's boundary word)
            }
        }
    }
    // ---- emit what this rank owns ----
    if (NR) {
        EmitParams ep{};
        ep.n_pieces = (uint32_t)NR; ep.k = k; ep.st = reinterpret_cast<const uint2*>(c->dg_rst.p); ep.hinfo = c->rank_a.p;
        ep.piece_n = c->dg_rn.p; ep.piece_kc = c->dg_rkc.p; ep.piece_boff = c->dg_rboff.p; ep.piece_bases = c->dg_rdense.p;
        ep.unitig_kc = c->unitig_kc.p; ep.out = c->unitig_bases.p;
        ep.piece_ab = c->prm.all_abundance_counts ? c->dg_rab.p : nullptr; ep.unitig_ab = c->unitig_ab.p;
        CDBG_LAUNCH(k_emit, (uint32_t)((NR + GLUE_THREADS - 1) / GLUE_THREADS), GLUE_THREADS, s, ep);
    }
    { uint64_t cur[2]; CK(read_u64(c->cursors.p + 2, cur, 2)); c->n_unitigs = cur[0]; c->unitig_total = cur[1]; }
    CK(pack_unitigs(c));
    float ms = 0; CK(t.stop(&ms));
    c->st.ms_glue = ms;
    CK(agree(c, check_device_error(c, "sharded glue"), "glue: emit"));
    c->joined = false; c->xchg_done = true;
    c->st.n_glue_joined = c->n_join_local; c->st.n_unitigs = c->n_unitigs; c->st.unitig_bases = c->unitig_total;
    c->st.ms_total += c->st.ms_glue;
    c->stage = 3;
    return CDBG_OK;
}

template <int W>
int glue_impl(cdbg_ctx* c) {
    if (c->stage < 2) return fail(CDBG_E_STATE, "cdbg_glue before cdbg_compact");
    if ((c->prm.world_size > 1 || c->force_multi) && c->have_tr && !c->xchg_done && !c->joined) {
        // multi-GPU.  Every rank emits its own unitigs (emit_replicated = 0): the sharded glue of k_dglue.h -- every record and
        // every piece travels once.  emit_replicated = 1 (the CLI's rank 0 writes one file and needs the whole graph for the
        // links), or closed chains across ranks: the replicated exchange -- pieces + junction log of all ranks to every rank.
        if (!c->prm.emit_replicated && getenv("CDBG_GLUE_REPLICATED") == nullptr) {
            const int rc = glue_sharded<W>(c);
            if (rc != DG_FALLBACK) return rc;
        }
        CK(glue_exchange(c));
    }
    if (!c->joined) CK(glue_join_impl<W>(c, false));         // (cdbg_glue_join ran it already in the sharded flow)
    hipStream_t s = c->stream;
    const uint64_t NP = c->n_pieces;
    const uint32_t NS = (uint32_t)(2 * NP);
    const float ms_join = c->st.ms_glue;
    HostMarks hm;
    Timer t; CK(t.start(s));
    DBuf<uint32_t>& flag = c->rank_flag; DBuf<uint4>& st_a = c->rank_a; DBuf<uint4>& st_b = c->rank_b;
    CK(st_a.alloc(NS, false)); CK(st_b.alloc(NS, false));
    CK(flag.alloc(4, true));
    HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
    HIPCK(hipMemsetAsync(c->derr.p, 0, 4 * sizeof(uint32_t), s));
    uint32_t* const link_p = c->link.p;

    uint64_t n_cycles_cut = 0;
    const uint32_t gridS = (NS + GLUE_THREADS - 1) / GLUE_THREADS;
    RankParams rp{};
    uint4* fa_st = nullptr;                                  // final state array of the 16-byte ranking
    const uint2* st8 = nullptr; uint4* hinfo = nullptr;      // what heads / emit read: 8-byte states, and the array for the per-head records
    bool ranked = false;
    if (NS) {
        // usual case (no closed chains): doubling on 8-byte states, expanded once at the end.  The 8-byte
        // state array (updated in place) lives in st_b; st_a takes the per-head records of heads / emit.
        int max_rounds8 = 2; while ((1ull << (max_rounds8 - 1)) < NS) ++max_rounds8;
        Rank8Params r8{ NS, link_p, c->piece_n.p, reinterpret_cast<uint2*>(st_b.p), flag.p };
        CDBG_LAUNCH(k_rank8_init, gridS, GLUE_THREADS, s, r8);
        for (int r = 0; r < max_rounds8 && !ranked; ++r) {
            HIPCK(hipMemsetAsync(flag.p, 0, sizeof(uint32_t), s));
            CDBG_LAUNCH(k_rank8_jump, gridS, GLUE_THREADS, s, r8);
            HIPCK(hipStreamSynchronize(s));
            uint32_t ch = 0; CK(read_u32(flag.p, &ch));
            if (!ch) ranked = true;
        }
        if (ranked) { st8 = r8.a; hinfo = st_a.p; }
    }
    if (NS && !ranked) {                                     // closed chains: the 16-byte version elects cut points
        int max_rounds = 2; while ((1ull << (max_rounds - 1)) < NS) ++max_rounds;
        for (int pass = 0; pass < 2; ++pass) {
            rp.n_states = NS; rp.link = link_p; rp.piece_n = c->piece_n.p;
            rp.st_a = st_a.p; rp.st_b = st_b.p; rp.changed = flag.p;
            CDBG_LAUNCH(k_rank_init, gridS, GLUE_THREADS, s, rp);
            bool converged = false;
            for (int r = 0; r < max_rounds; ++r) {
                HIPCK(hipMemsetAsync(flag.p, 0, sizeof(uint32_t), s));
                CDBG_LAUNCH(k_rank_jump, gridS, GLUE_THREADS, s, rp);
                std::swap(rp.st_a, rp.st_b);
                HIPCK(hipStreamSynchronize(s));
                uint32_t ch = 0; CK(read_u32(flag.p, &ch));
                if (!ch) { converged = true; break; }
            }
            fa_st = rp.st_a;
            if (converged) break;
            if (pass == 1) return fail(CDBG_E_INTERNAL, "list ranking did not converge after cutting cycles");
            // closed chains: cut each at its smallest piece, then rank again
            HIPCK(hipMemsetAsync(flag.p, 0, sizeof(uint32_t), s));
            CutParams cu{ NS, rp.st_a, link_p, flag.p };
            CDBG_LAUNCH(k_cut_cycles, gridS, GLUE_THREADS, s, cu);
            HIPCK(hipStreamSynchronize(s));
            uint32_t nc = 0; CK(read_u32(flag.p, &nc)); n_cycles_cut += nc;
        }
    }
    if (NS && !st8) {                                        // (16-byte ranking: narrow its final states into the other buffer)
        uint4* const other = (fa_st == st_a.p) ? st_b.p : st_a.p;
        RankNarrowParams np{ NS, fa_st, reinterpret_cast<uint2*>(other) };
        CDBG_LAUNCH(k_rank_narrow, gridS, GLUE_THREADS, s, np);
        st8 = reinterpret_cast<const uint2*>(other); hinfo = fa_st;
    }
    // unitig heads + emission
    const uint64_t ucap = std::max<uint64_t>(NP, 1);
    const uint64_t ocap = std::max<uint64_t>(c->n_piece_bases, 1);
    CK(c->unitig_off.alloc(ucap, false)); CK(c->unitig_len.alloc(ucap, false)); CK(c->unitig_kc.alloc(ucap, false));
    CK(c->unitig_bases.alloc(ocap + 64, false));             // (+ 64: the 2-bit packing pass reads whole 64-base chunks)
    if (c->prm.all_abundance_counts) CK(c->unitig_ab.alloc(ocap, false));
    HIPCK(hipMemsetAsync(c->cursors.p + 2, 0, 2 * sizeof(uint64_t), s));
    if (NS) {
        HeadParams hp{};
        hp.n_states = NS; hp.k = c->k; hp.link = link_p; hp.st = st8; hp.hinfo = hinfo;
        hp.unitig_off = c->unitig_off.p; hp.unitig_len = c->unitig_len.p; hp.unitig_kc = c->unitig_kc.p;
        hp.unitig_cap = ucap; hp.out_cap = ocap; hp.n_unitigs = c->cursors.p + 2; hp.out_cursor = c->cursors.p + 3; hp.error = c->derr.p;
        hp.own_lo = 0; hp.own_hi = NS;
        if (c->prm.world_size > 1 && !c->prm.emit_replicated && c->xchg_done) { hp.own_lo = (uint32_t)(2 * c->piece_lo); hp.own_hi = (uint32_t)(2 * c->piece_hi); }
        CDBG_LAUNCH(k_unitig_heads, (NS + HEADS_PER_WG - 1) / HEADS_PER_WG, GLUE_THREADS, s, hp);
        EmitParams ep{};
        ep.n_pieces = (uint32_t)NP; ep.k = c->k; ep.st = st8; ep.hinfo = hp.hinfo;
        ep.piece_n = c->piece_n.p; ep.piece_kc = c->piece_kc.p; ep.piece_boff = c->piece_boff.p; ep.piece_bases = c->piece_bases.p;
        ep.unitig_kc = c->unitig_kc.p; ep.out = c->unitig_bases.p;
        ep.piece_ab = c->prm.all_abundance_counts ? c->piece_ab.p : nullptr; ep.unitig_ab = c->unitig_ab.p;
        CDBG_LAUNCH(k_emit, (uint32_t)((NP + GLUE_THREADS - 1) / GLUE_THREADS), GLUE_THREADS, s, ep);
    }
    { uint64_t cur[2]; CK(read_u64(c->cursors.p + 2, cur, 2)); c->n_unitigs = cur[0]; c->unitig_total = cur[1]; }
    CK(pack_unitigs(c));
    float ms_fin = 0; CK(t.stop(&ms_fin));
    hm.mark("glue: rank + heads + emit");
    c->st.ms_glue = ms_join + ms_fin;
    CK(check_device_error(c, "glue"));
    c->joined = false;
    c->st.n_glue_joined = c->n_join_local; c->st.n_unitigs = c->n_unitigs; c->st.unitig_bases = c->unitig_total; c->st.n_cycles += n_cycles_cut;
    c->st.ms_total += c->st.ms_glue;
    c->stage = 3;
    return CDBG_OK;
}

template <int W>
int link_impl(cdbg_ctx* c) {
    if (c->stage < 3) return fail(CDBG_E_STATE, "cdbg_link before cdbg_glue");
    hipStream_t s = c->stream;
    const uint64_t U = c->n_unitigs, NE = 2 * U;
    // (k_links.h packs a slot index with a flag in bit 30: the table may have at most 2^30 slots)
    if (pow2_at_least(4 * U + 64) > (1ull << 30)) return fail(CDBG_E_INTERNAL, "too many unitigs (%llu) for the 30-bit slots of the link table", (unsigned long long)U);
    const uint32_t cap = (uint32_t)pow2_at_least(4 * U + 64);
    DBuf<uint64_t> lk_keys; DBuf<uint32_t> lk_cnt, lk_ends, end_slot, deg;
    CK(lk_keys.alloc((uint64_t)cap * W, false)); CK(lk_cnt.alloc((uint64_t)cap * 2, true));
    CK(lk_ends.alloc((uint64_t)cap * 2 * LINK_PER_FLAG, false)); CK(end_slot.alloc(NE, false)); CK(deg.alloc(NE, false));
    HIPCK(hipMemsetAsync(lk_keys.p, 0xFF, (uint64_t)cap * W * sizeof(uint64_t), s));
    CK(c->link_off.alloc(NE + 1, true));
    LinkParams lp{};
    lp.n_unitigs = U; lp.k = c->k; lp.unitig_off = c->unitig_off.p; lp.unitig_len = c->unitig_len.p; lp.bases = c->unitig_bases.p;
    lp.lk_keys = lk_keys.p; lp.lk_cnt = lk_cnt.p; lp.lk_ends = lk_ends.p; lp.lk_mask = cap - 1;
    lp.end_slot = end_slot.p; lp.deg = deg.p;
    c->n_links = 0;
    if (NE) {
        const uint64_t grid = (NE + LINK_THREADS - 1) / LINK_THREADS;
        CDBG_LAUNCH((k_link_insert<W>), grid, LINK_THREADS, s, lp);
        CDBG_LAUNCH(k_link_count, grid, LINK_THREADS, s, lp);
        const uint64_t nb = (NE + EXSCAN_BLOCK - 1) / EXSCAN_BLOCK;
        CK(c->exscan_tmp.alloc(nb + 1, false));
        const uint32_t* degp = deg.p;                        // (plain pointer: launch arguments are captured by value)
        CDBG_LAUNCH(k_exscan_sums, nb, EXSCAN_THREADS, s, degp, c->exscan_tmp.p, NE);
        CDBG_LAUNCH(k_exscan_top, 1, EXSCAN_THREADS, s, c->exscan_tmp.p, nb, c->link_off.p + NE);
        CDBG_LAUNCH(k_exscan_apply, nb, EXSCAN_THREADS, s, degp, (const uint64_t*)c->exscan_tmp.p, c->link_off.p, NE);
        CK(read_u64(c->link_off.p + NE, &c->n_links));
        CK(c->link_to.alloc(c->n_links, false));
        lp.link_off = c->link_off.p; lp.link_to = c->link_to.p;
        CDBG_LAUNCH(k_link_fill, grid, LINK_THREADS, s, lp);
        HIPCK(hipStreamSynchronize(s));
    }
    c->linked = true;
    return CDBG_OK;
}

// the unitig definition checked on the resident result (k_verify.h)
template <int W>
int verify_impl(cdbg_ctx* c, uint64_t* out) {
    if (c->stage < 3) return fail(CDBG_E_STATE, "cdbg_verify before cdbg_glue");
    hipStream_t s = c->stream;
    const bool sharded_set = (c->prm.world_size > 1 || c->force_multi) && !c->prm.emit_replicated;   // this rank holds a share of the unitigs: no links
    if (!sharded_set && !c->linked) CK(link_impl<W>(c));
    DBuf<uint64_t> d; CK(d.alloc(8, true));
    VerifyParams vp{ c->n_unitigs, c->k, c->unitig_off.p, c->unitig_len.p, c->unitig_bases.p,
                     c->seg_off.p, c->seg_n.p, c->solid_keys.p, c->solid_cnt.p, c->n_local_parts, c->link_off.p, c->link_to.p, d.p };
    if (c->n_unitigs) CDBG_LAUNCH((k_verify_unitig_kmers<W>), std::min<uint64_t>((c->n_unitigs + 255) / 256, 1u << 16), 256, s, vp);
    CDBG_LAUNCH((k_verify_solid<W>), std::min<uint64_t>((c->n_local_parts + 255) / 256, 1u << 16), 256, s, vp);
    if (!sharded_set && c->n_unitigs) CDBG_LAUNCH(k_verify_maximal, (2 * c->n_unitigs + 255) / 256, 256, s, vp);
    HIPCK(hipStreamSynchronize(s));
    CK(read_u64(d.p, out, 8));
    if (sharded_set) out[6] = out[7] = ~0ull;
    return CDBG_OK;
}

}  // namespace

// =======================================================================================
// C ABI
// =======================================================================================
extern "C" {

const char* cdbg_last_error(void) { return g_err.c_str(); }

int cdbg_create(const cdbg_params* p, cdbg_ctx** out) {
    if (!p || !out) return fail(CDBG_E_PARAM, "null argument");
    *out = nullptr;
    if (p->k < 3 || p->k > 127) return fail(CDBG_E_PARAM, "kmer-size %d out of range (3..127)", p->k);
    if (p->abundance_min < 1) return fail(CDBG_E_PARAM, "abundance-min must be >= 1");
    const int ws = p->world_size <= 0 ? 1 : p->world_size;
    if (ws & (ws - 1)) return fail(CDBG_E_PARAM, "world_size must be a power of two");
    if (p->rank < 0 || p->rank >= ws) return fail(CDBG_E_PARAM, "rank out of range");
    if (p->minimizer_size > 16) return fail(CDBG_E_PARAM, "minimizer-size must be <= 16");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(CDBG_E_NODEVICE, "no HIP device available (%s): libcdbg has no CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (p->device_id < 0 || p->device_id >= ndev) return fail(CDBG_E_PARAM, "device_id %d out of range (%d devices)", p->device_id, ndev);
    HIPCK(hipSetDevice(p->device_id));
    cdbg_ctx* c = new cdbg_ctx();
    c->prm = *p; c->prm.world_size = ws;
    // words per k-mer: the reference's span rule k < 32 W (README.md:91-99, Integer::apply at src/bcalm_1.cpp:95); the top word of a
    // multi-word key therefore always keeps its two top bits free for the slot-claim protocol (k_count.h), for even k as well
    c->k = p->k; c->W = p->k <= 31 ? 1 : p->k <= 63 ? 2 : p->k <= 95 ? 3 : 4;
    c->rank_bits = 0; while ((1 << c->rank_bits) < ws) ++c->rank_bits;
    if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return fail(CDBG_E_NODEVICE, "hipStreamCreate failed"); }
    *out = c;
    return CDBG_OK;
}

void cdbg_destroy(cdbg_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    ingest_release(c);
#ifndef CDBG_HOSTSIM
    if (c->rccl) { c->rccl->destroy(); delete c->rccl; c->rccl = nullptr; }
#endif
    (void)hipStreamSynchronize(c->stream);
    stash_region(c->prm.device_id, c->records);
    (void)hipStreamDestroy(c->stream);
    delete c;
}
int cdbg_release_cached(void) {
    RegionStash& st = region_stash();
    std::lock_guard<std::mutex> g(st.mu);
    int cur = 0; (void)hipGetDevice(&cur);
    for (int d = 0; d < 64; ++d) if (st.p[d]) { (void)hipSetDevice(d); (void)hipFree(st.p[d]); st.p[d] = nullptr; st.cap[d] = 0; }
    (void)hipSetDevice(cur);
    return CDBG_OK;
}

int cdbg_push_reads(cdbg_ctx* c, const char* bases, const uint64_t* offsets, uint64_t n_reads) {
    if (!c || !bases || !offsets) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    if (c->stage != 0 || c->reads_final) return fail(CDBG_E_STATE, "reads must be pushed before the first stage");
    for (uint64_t i = 0; i < n_reads; ++i) {
        if (offsets[i + 1] < offsets[i]) return fail(CDBG_E_PARAM, "offsets not monotone at read %llu", (unsigned long long)i);
        CK(ingest_append(c, bases + offsets[i], offsets[i + 1] - offsets[i]));
        CK(ingest_append(c, "\n", 1));
    }
    return CDBG_OK;
}
int cdbg_push_text(cdbg_ctx* c, const char* text, uint64_t nbytes) {
    if (!c || (!text && nbytes)) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    if (c->stage != 0 || c->reads_final) return fail(CDBG_E_STATE, "reads must be pushed before the first stage");
    CK(ingest_append(c, text, nbytes));
    CK(ingest_append(c, "\n", 1));
    return CDBG_OK;
}
int cdbg_expect_input(cdbg_ctx* c, uint64_t text_bytes) {
    if (!c) return fail(CDBG_E_PARAM, "null context");
    if (c->stage != 0 || c->n_dev || c->pin_fill) return fail(CDBG_E_STATE, "cdbg_expect_input must precede the first push");
    c->expect_bytes = text_bytes;
    return CDBG_OK;
}
int cdbg_generate_reads(cdbg_ctx* c, uint64_t first_read, uint64_t n_reads, uint64_t total_reads, uint64_t read_len, int cfg) {
    if (!c) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    if (c->stage != 0 || c->reads_final || c->n_dev || c->pin_fill) return fail(CDBG_E_STATE, "reads already present");
    if (!n_reads || !read_len || total_reads < n_reads) return fail(CDBG_E_PARAM, "bad synthetic read set");
    const uint64_t n = n_reads * (read_len + 1);
    const uint64_t np = ((n + 15) / 16) * 16 + 256;
    CK(c->reads.alloc(np, false));
    HIPCK(hipMemsetAsync(c->reads.p + n, '\n', np - n, c->stream));
    GenParams g{ c->reads.p, first_read, n_reads, total_reads, read_len, cfg };
    const uint64_t blocks = std::min<uint64_t>((n + 255) / 256, MAX_GRID);
    CDBG_LAUNCH(k_gen_reads, blocks, 256, c->stream, g);
    HIPCK(hipStreamSynchronize(c->stream));
    c->nbytes = n; c->nbytes_padded = np; c->reads_final = true;
    return CDBG_OK;
}
int cdbg_read_text(cdbg_ctx* c, uint64_t first_byte, uint64_t nbytes, char* out) {
    if (!c || !out) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    CK(upload_pending(c));
    if (first_byte + nbytes > c->nbytes) return fail(CDBG_E_PARAM, "range beyond the resident text");
    HIPCK(hipMemcpy(out, c->reads.p + first_byte, nbytes, hipMemcpyDeviceToHost));
    return CDBG_OK;
}

#define DISPATCH_W(fn)                                              \
    switch (c->W) {                                                  \
        case 1: return fn<1>(c);                                     \
        case 2: return fn<2>(c);                                     \
        case 3: return fn<3>(c);                                     \
        default: return fn<4>(c);                                    \
    }
static int count_dispatch(cdbg_ctx* c) { DISPATCH_W(count_impl) }
int cdbg_count(cdbg_ctx* c) {
    if (!c) return fail(CDBG_E_PARAM, "null context");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    if (c->stage != 0) return fail(CDBG_E_STATE, "cdbg_count called twice");
    CK(count_dispatch(c));
    // Compaction handles a bucket in LDS up to 1024 entries (2 x TSK / 2); beyond that it falls back to tables in HBM,
    // which is 20-50 x slower per bucket.  The partition count was chosen for the COUNT table (occurrences per
    // partition); with few k-mers filtered out (abundance-min 1, deep coverage without errors) the solid entries per
    // bucket can be several times what the compaction tiers hold.  One look at the exact number after counting:
    // if the mean is above 300 entries, count again with enough partitions for ~150 per bucket (scan + count are
    // cheap next to the fallback), unless the caller fixed the partition count.
    // (several ranks: the decision is taken on the sum over the ranks -- every rank must use the same partitioning)
    uint64_t entries = c->st.n_solid + c->st.n_solid_travellers, parts = std::max<uint64_t>(c->n_local_parts, 1);
    if (c->prm.world_size > 1 || c->force_multi) {
        std::vector<uint64_t> all(c->prm.world_size);
        if (c->tr.all_gather_u64(c->tr.user, &entries, all.data(), 1) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_u64 failed");
        entries = 0; for (uint64_t v : all) entries += v;
        parts *= (uint64_t)c->prm.world_size;
    }
    const uint64_t per_bucket = entries / parts;
    if (c->prm.log2_partitions < 0 && c->log_np_override < 0 && per_bucket > 300 && c->log_np < 26) {
        int extra = 1; while ((per_bucket >> extra) > 150 && c->log_np + extra < 26) ++extra;
        const float first_ms = c->st.ms_total;
        c->log_np_override = c->log_np + extra;
        c->stage = 0; c->st.n_big_partitions = 0; c->st.n_multipass_partitions = 0;       // (the first attempt's fallbacks are not part of the result)
        CK(count_dispatch(c));
        c->st.ms_total += first_ms;                          // the first attempt is part of the stage's time
    }
    return CDBG_OK;
}
int cdbg_compact(cdbg_ctx* c) { if (!c) return fail(CDBG_E_PARAM, "null context"); (void)hipSetDevice(c->prm.device_id); DISPATCH_W(compact_impl) }
int cdbg_glue(cdbg_ctx* c) { if (!c) return fail(CDBG_E_PARAM, "null context"); (void)hipSetDevice(c->prm.device_id); DISPATCH_W(glue_impl) }
int cdbg_run(cdbg_ctx* c) { CK(cdbg_count(c)); CK(cdbg_compact(c)); return cdbg_glue(c); }
int cdbg_link(cdbg_ctx* c) { if (!c) return fail(CDBG_E_PARAM, "null context"); (void)hipSetDevice(c->prm.device_id); DISPATCH_W(link_impl) }
int cdbg_num_links(cdbg_ctx* c, uint64_t* n) {
    if (!c || !n) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    if (!c->linked) return fail(CDBG_E_STATE, "cdbg_num_links before cdbg_link");
    *n = c->n_links; return CDBG_OK;
}
int cdbg_fetch_links(cdbg_ctx* c, uint64_t* end_off, uint32_t* link_to) {
    if (!c || !end_off) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    if (!c->linked) return fail(CDBG_E_STATE, "cdbg_fetch_links before cdbg_link");
    HIPCK(hipMemcpy(end_off, c->link_off.p, (2 * c->n_unitigs + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (c->n_links && link_to) HIPCK(hipMemcpy(link_to, c->link_to.p, c->n_links * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return CDBG_OK;
}
int cdbg_reset(cdbg_ctx* c) {
    if (!c) return fail(CDBG_E_PARAM, "null context");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    c->stage = 0; c->st = cdbg_stats_t{};
    c->n_solid_entries = c->n_pieces = c->n_piece_bases = c->n_unitigs = c->unitig_total = 0; c->linked = false; c->n_links = 0; c->joined = false;
    c->xchg_done = false; c->xp_ab_ready = false; c->comm_bytes = 0; c->piece_lo = c->piece_hi = 0; c->ss_on = false; c->expect_bytes = 0;
    return CDBG_OK;                                  // reads and every device buffer stay resident
}

int cdbg_num_solid(cdbg_ctx* c, uint64_t* n) {
    if (!c || !n) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    if (c->stage < 1) return fail(CDBG_E_STATE, "cdbg_num_solid before cdbg_count");
    *n = c->st.n_solid; return CDBG_OK;
}
int cdbg_fetch_solid(cdbg_ctx* c, char* kmers, uint32_t* counts, uint64_t capacity, uint64_t* n_written) {
    if (!c || !kmers || !counts || !n_written) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    if (c->stage < 1) return fail(CDBG_E_STATE, "cdbg_fetch_solid before cdbg_count");
    if (capacity < c->st.n_solid) return fail(CDBG_E_PARAM, "capacity %llu < %llu solid k-mers", (unsigned long long)capacity, (unsigned long long)c->st.n_solid);
    const uint64_t S = c->st.n_solid, E = c->n_solid_entries;
    *n_written = 0;
    if (!S) return CDBG_OK;
    DBuf<uint8_t> dk; DBuf<uint32_t> dc; DBuf<uint64_t> dn;
    CK(dk.alloc(S * (uint64_t)(c->k + 1), false)); CK(dc.alloc(S, false)); CK(dn.alloc(1, true));
    (void)E;
    DecodeParams dp{ c->solid_keys.p, c->solid_cnt.p, c->seg_off.p, c->seg_n.p, c->n_local_parts, c->k, c->W, dk.p, dc.p, dn.p, S };
    CDBG_LAUNCH(k_decode_solid, (c->n_local_parts + 255) / 256, 256, c->stream, dp);
    HIPCK(hipStreamSynchronize(c->stream));
    uint64_t nw = 0; CK(read_u64(dn.p, &nw));
    if (nw != S) return fail(CDBG_E_INTERNAL, "solid k-mer bookkeeping mismatch: %llu decoded vs %llu counted", (unsigned long long)nw, (unsigned long long)S);
    HIPCK(hipMemcpy(kmers, dk.p, S * (uint64_t)(c->k + 1), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(counts, dc.p, S * sizeof(uint32_t), hipMemcpyDeviceToHost));
    *n_written = S;
    return CDBG_OK;
}
int cdbg_num_unitigs(cdbg_ctx* c, uint64_t* n, uint64_t* total_bases) {
    if (!c || !n) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    if (c->stage < 3) return fail(CDBG_E_STATE, "cdbg_num_unitigs before cdbg_glue");
    *n = c->n_unitigs; if (total_bases) *total_bases = c->unitig_total; return CDBG_OK;
}
int cdbg_fetch_unitigs(cdbg_ctx* c, uint64_t first, uint64_t n, char* seq_buf, uint64_t* seq_off, uint64_t* kc) {
    if (!c || !seq_buf || !seq_off || !kc) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    if (c->stage < 3) return fail(CDBG_E_STATE, "cdbg_fetch_unitigs before cdbg_glue");
    if (first + n > c->n_unitigs) return fail(CDBG_E_PARAM, "unitig range out of bounds");
    if (!n) { seq_off[0] = 0; return CDBG_OK; }
    std::vector<uint64_t> off(n); std::vector<uint32_t> len(n);
    HIPCK(hipMemcpy(off.data(), c->unitig_off.p + first, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(len.data(), c->unitig_len.p + first, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(kc, c->unitig_kc.p + first, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    // unitigs are laid out in allocation order, not id order: copy the whole arena once when
    // the request covers everything, otherwise one copy per unitig
    uint64_t w = 0;
    if (first == 0 && n == c->n_unitigs) {
        std::vector<char> arena(c->unitig_total);
        HIPCK(hipMemcpy(arena.data(), c->unitig_bases.p, c->unitig_total, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; ++i) { seq_off[i] = w; memcpy(seq_buf + w, arena.data() + off[i], len[i]); w += len[i]; }
    } else {                                                 // a sub-range: gathered gap-free on the device, one copy back
        for (uint64_t i = 0; i < n; ++i) { seq_off[i] = w; w += len[i]; }
        DBuf<uint64_t> doff; DBuf<uint8_t> dense;
        CK(doff.alloc(n, false)); CK(dense.alloc(w, false));
        HIPCK(hipMemcpy(doff.p, seq_off, n * sizeof(uint64_t), hipMemcpyHostToDevice));
        GatherUnitigParams gp{ n, c->unitig_off.p + first, c->unitig_len.p + first, doff.p, c->unitig_bases.p, dense.p };
        CDBG_LAUNCH(k_gather_unitigs, (uint32_t)((n * 64 + 255) / 256), 256, c->stream, gp);
        HIPCK(hipStreamSynchronize(c->stream));
        HIPCK(hipMemcpy(seq_buf, dense.p, w, hipMemcpyDeviceToHost));
    }
    seq_off[n] = w;
    return CDBG_OK;
}
int cdbg_fetch_unitigs_packed(cdbg_ctx* c, uint8_t* packed, uint64_t packed_capacity, uint64_t* base_off, uint32_t* len, uint64_t* kc) {
    if (!c || !packed || !base_off || !len || !kc) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    if (c->stage < 3) return fail(CDBG_E_STATE, "cdbg_fetch_unitigs_packed before cdbg_glue");
    const uint64_t nbytes = (c->unitig_total + 3) / 4;
    if (packed_capacity < nbytes) return fail(CDBG_E_PARAM, "packed buffer too small (%llu < %llu bytes)", (unsigned long long)packed_capacity, (unsigned long long)nbytes);
    const uint64_t n = c->n_unitigs;
    if (nbytes) HIPCK(hipMemcpy(packed, c->unitig_packed.p, nbytes, hipMemcpyDeviceToHost));
    if (n) {
        HIPCK(hipMemcpy(base_off, c->unitig_off.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(len, c->unitig_len.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(kc, c->unitig_kc.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    return CDBG_OK;
}
int cdbg_fetch_unitig_abundances(cdbg_ctx* c, uint64_t first, uint64_t n, uint32_t* ab, uint64_t* ab_off) {
    if (!c || !ab || !ab_off) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    if (c->stage < 3) return fail(CDBG_E_STATE, "cdbg_fetch_unitig_abundances before cdbg_glue");
    if (!c->prm.all_abundance_counts) return fail(CDBG_E_STATE, "context was created without all_abundance_counts");
    if (first + n > c->n_unitigs) return fail(CDBG_E_PARAM, "unitig range out of bounds");
    if (!n) { ab_off[0] = 0; return CDBG_OK; }
    std::vector<uint64_t> off(n); std::vector<uint32_t> len(n);
    HIPCK(hipMemcpy(off.data(), c->unitig_off.p + first, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(len.data(), c->unitig_len.p + first, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<uint32_t> arena(c->unitig_total);
    HIPCK(hipMemcpy(arena.data(), c->unitig_ab.p, c->unitig_total * sizeof(uint32_t), hipMemcpyDeviceToHost));
    uint64_t w = 0;
    for (uint64_t i = 0; i < n; ++i) {
        ab_off[i] = w;
        const uint32_t nk = len[i] - (uint32_t)c->k + 1u;
        memcpy(ab + w, arena.data() + off[i] + (c->k - 1), nk * sizeof(uint32_t)); w += nk;
    }
    ab_off[n] = w;
    return CDBG_OK;
}
int cdbg_set_transport(cdbg_ctx* c, const cdbg_transport* t) {
    if (!c || !t || !t->all_gather_u64 || !t->all_to_all_v || !t->all_gather_v || !t->all_reduce_max_i32) return fail(CDBG_E_PARAM, "null transport");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    c->tr = *t; c->have_tr = true; c->tr_ordered = false; c->force_multi = getenv("CDBG_FORCE_MULTI") != nullptr; return CDBG_OK;
}
int cdbg_comm_unique_id(void* out) {
    if (!out) return fail(CDBG_E_PARAM, "null argument");
#ifdef CDBG_HOSTSIM
    return fail(CDBG_E_NODEVICE, "no RCCL in the simulator build");
#else
    std::string err; RcclApi& A = rccl_api();
    if (!A.load(err)) return fail(CDBG_E_NODEVICE, "%s", err.c_str());
    RcclApi::UniqueId id; const int rc = A.GetUniqueId(&id);
    if (rc != 0) return fail(CDBG_E_NODEVICE, "ncclGetUniqueId: %s", A.GetErrorString(rc));
    memcpy(out, &id, sizeof id); return CDBG_OK;
#endif
}
int cdbg_comm_init_rccl(cdbg_ctx* c, const void* uid) {
    if (!c || !uid) return fail(CDBG_E_PARAM, "null argument");
#ifdef CDBG_HOSTSIM
    return fail(CDBG_E_NODEVICE, "no RCCL in the simulator build");
#else
    HIPCK(hipSetDevice(c->prm.device_id));
    if (c->rccl) { c->rccl->destroy(); delete c->rccl; c->rccl = nullptr; }
    c->rccl = new RcclComm();
    if (!c->rccl->init(uid, c->prm.world_size, c->prm.rank, c->stream)) { const std::string e = c->rccl->err; c->rccl->destroy(); delete c->rccl; c->rccl = nullptr; return fail(CDBG_E_NODEVICE, "RCCL: %s", e.c_str()); }
    c->tr = c->rccl->transport(); c->have_tr = true; c->tr_ordered = true; c->force_multi = getenv("CDBG_FORCE_MULTI") != nullptr;
    return CDBG_OK;
#endif
}
int cdbg_comm_bytes(cdbg_ctx* c, uint64_t* out) { if (!c || !out) return fail(CDBG_E_PARAM, "null argument"); *out = c->comm_bytes; return CDBG_OK; }
int cdbg_digest(cdbg_ctx* c, uint64_t out[4]) {
    if (!c || !out) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    if (c->stage < 3) return fail(CDBG_E_STATE, "cdbg_digest before cdbg_glue");
    DBuf<uint64_t> d; CK(d.alloc(4, true));
    DigestParams dp{ c->n_unitigs, c->k, c->unitig_off.p, c->unitig_len.p, c->unitig_kc.p, c->unitig_bases.p,
                     c->seg_off.p, c->seg_n.p, c->solid_cnt.p, c->n_local_parts, d.p };
    if (c->n_unitigs) CDBG_LAUNCH(k_digest_unitigs, std::min<uint64_t>((c->n_unitigs + 255) / 256, 1u << 16), 256, c->stream, dp);
    CDBG_LAUNCH(k_digest_solid, std::min<uint64_t>((c->n_local_parts + 255) / 256, 1u << 16), 256, c->stream, dp);
    HIPCK(hipStreamSynchronize(c->stream));
    CK(read_u64(d.p, out, 4));
    return CDBG_OK;
}
int cdbg_verify(cdbg_ctx* c, uint64_t out[8]) {
    if (!c || !out) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);
    switch (c->W) { case 1: return verify_impl<1>(c, out); case 2: return verify_impl<2>(c, out); case 3: return verify_impl<3>(c, out); default: return verify_impl<4>(c, out); }
}
int cdbg_stats(cdbg_ctx* c, cdbg_stats_t* out) {
    if (!c || !out) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    *out = c->st; return CDBG_OK;
}

}  // extern "C"
