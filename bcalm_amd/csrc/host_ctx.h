// host_ctx.h -- libcdbg.so, host side: error plumbing, owned device buffers, the context (everything a job keeps in HBM), the
// streaming ingest, DSK's "configure" role (row a5) and the small helpers every stage uses.
// Included by cdbg_impl.cpp only (one translation unit; see there for the two ways this source is built).
#pragma once

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

// Device buffers outlive their context.  The scan's 1.6 G scattered 16-byte stores and as many far atomics are sensitive to WHERE
// the 75 GB record region and the 16 MB of fill counters lie physically: every free + re-allocation handed back a less contiguous
// set of pages, and five contexts created and destroyed in one process scanned in 66.7 -> 70.8 -> 68.9 -> 75.3 -> 77.9 ms
// (profiles/r03_scan_variance_by_allocation.log).  Buffers of 1 MB and more therefore go back to a per-device POOL of the process
// instead of the driver, and an allocation takes the smallest pooled block that is large enough (and not more than twice as large):
// a second context of the same shape runs on the very pages of the first.  cdbg_release_cached() empties the pool; a pooled
// volume beyond DevPool::limit() -- a fraction of the card -- is freed at once.
struct DevPool {
    static constexpr size_t MIN_BYTES = 1u << 20;
    struct Block { void* p; size_t bytes; };
    std::mutex mu; std::vector<Block> blocks[64]; size_t held[64] = {}; size_t limit_[64] = {};
    static int device() { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0; return d; }
    // what the pool may keep on device d: 55 % of the card (160 GB of an MI355X's 288: one config-3 context), nothing with
    // CDBG_NO_POOL set -- torch tensors, RCCL buffers and pinned staging of the same process cannot drain the pool, so it must
    // leave them room whatever the card's size.  Read once per device (callers hold mu).
    size_t limit(int d) {
        if (!limit_[d]) {
            size_t fr = 0, tot = 0;
            if (getenv("CDBG_NO_POOL")) limit_[d] = 1;
            else if (hipMemGetInfo(&fr, &tot) == hipSuccess && tot) limit_[d] = std::max<size_t>(tot / 100 * 55, 1);
            else limit_[d] = 1;
        }
        return limit_[d];
    }
    void* take(size_t want, size_t* got) {
        const int d = device();
        std::lock_guard<std::mutex> g(mu);
        int best = -1;
        for (int i = 0; i < (int)blocks[d].size(); ++i)
            if (blocks[d][i].bytes >= want && blocks[d][i].bytes <= 2 * want && (best < 0 || blocks[d][i].bytes < blocks[d][best].bytes)) best = i;
        if (best < 0) return nullptr;
        void* p = blocks[d][best].p; *got = blocks[d][best].bytes; held[d] -= *got;
        blocks[d].erase(blocks[d].begin() + best);
        return p;
    }
    void give(void* p, size_t bytes) {
        const int d = device();
        (void)hipDeviceSynchronize();                    // (what hipFree did: the block may be handed to a context that works on another stream)
        {
            std::lock_guard<std::mutex> g(mu);
            if (bytes >= MIN_BYTES && held[d] + bytes <= limit(d)) { blocks[d].push_back({ p, bytes }); held[d] += bytes; return; }
        }
        (void)hipFree(p);
    }
    // everything back to the driver (an allocation failed, or the caller asked).  One drain at a time, and a second caller waits for the
    // first one's hipFree calls: contexts on several host threads run out of memory together, and the thread that found the pool already
    // emptied retried its hipMalloc before the blocks were back with the driver -- "out of memory" with 150 GB about to be free (found by
    // the round-5 fuzz of eight ranks on one device)
    std::mutex drain_mu;
    void drain(int d) {
        std::lock_guard<std::mutex> dg(drain_mu);
        std::vector<Block> bl;
        { std::lock_guard<std::mutex> g(mu); bl.swap(blocks[d]); held[d] = 0; }
        for (const Block& b : bl) (void)hipFree(b.p);
    }
};
inline DevPool& dev_pool() { static DevPool p; return p; }

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
            size_t got = 0;
            if (want * sizeof(T) >= DevPool::MIN_BYTES) p = (T*)dev_pool().take(want * sizeof(T), &got);
            if (p) cap = got / sizeof(T);
            else {
                static const bool dbg = getenv("CDBG_DEBUG_ALLOC") != nullptr;   // dev aid (read once per process): every fresh allocation of 256 MB and more, with what the card has free
                if (dbg && want * sizeof(T) >= (256u << 20)) { size_t fr = 0, tot = 0; (void)hipMemGetInfo(&fr, &tot); fprintf(stderr, "[alloc] %8.2f GB (free %7.2f of %7.2f GB, pool %7.2f GB)\n", (double)(want * sizeof(T)) / 1e9, (double)fr / 1e9, (double)tot / 1e9, (double)dev_pool().held[DevPool::device()] / 1e9); }
                hipError_t e = hipMalloc(&p, want * sizeof(T));
                // (pooled blocks of other shapes may be in the way.  The failed call leaves its error as the thread's LAST error: read it away, or the
                //  next launch check -- hipGetLastError() -- reports an out-of-memory that was dealt with here; found by the round-5 fuzz session)
                if (e != hipSuccess) { (void)hipGetLastError(); dev_pool().drain(DevPool::device()); e = hipMalloc(&p, want * sizeof(T)); if (e != hipSuccess) (void)hipGetLastError(); }
                if (e != hipSuccess) { p = nullptr; return fail(CDBG_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", want * sizeof(T), hipGetErrorString(e)); }
                cap = want;
            }
        }
        n = count;
        if (zero) { hipError_t e = hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T)); if (e != hipSuccess) return fail(CDBG_E_NODEVICE, "hipMemset failed"); }
        return CDBG_OK;
    }
    void release() { if (p) { dev_pool().give(p, cap * sizeof(T)); p = nullptr; n = 0; cap = 0; } }
    void swap(DBuf& o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); }
    DBuf() = default;
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    ~DBuf() { release(); }
};

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
// k-mers wider than four words (k = 128 .. 255: the reference's KSIZE_LIST is open-ended, README.md:91-99 -- "must contain 32", larger
// spans are a build option there as here: CDBG_MAX_W).  Same kernels, tables a quarter / half the slots of the four-word geometry so that
// keys of 40 - 64 bytes still fit the CU's LDS; not tuned (no BASELINE config lives there), parity-tested at k = 128, 191 and 255.
#ifndef CDBG_MAX_W
#define CDBG_MAX_W 8
#endif
template <int W> struct Cfg { static constexpr int TSC = 1024, TSK = 256, TSK2 = 512, NTC = 512, TSW = 128, TSW2 = 256; };
// TSW: slots of the wave-per-bucket compaction tier (buckets of at most TSW / 2 entries; k_compact_wave.h)
template <> struct Cfg<1> { static constexpr int TSC = TS_COUNT_1, TSK = TS_COMPACT_1, TSK2 = 2 * TS_COMPACT_1, NTC = CDBG_NTC1, TSW = 512, TSW2 = 1024; };   // (TSW2: the second wave tier, buckets of 257 .. 512 entries: round 5)
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
// the one-pass count kernel: 768 workgroups are resident (3 per CU); the more take turns, the shorter the tail in which the last ones
// run alone -- count at config 3: 63.8 / 61.2 / 58.7 / 57.7 / 57.1 / 56.6 ms with 768 / 1536 / 3072 / 6144 / 12288 / 24576 workgroups
// (profiles/r04_ab_cfg3_count_grid.log); every workgroup may strand one output chunk, hence the smaller COUNT_CHUNK (k_count.h)
#ifndef CDBG_COUNT_GRID
#define CDBG_COUNT_GRID (256 * 48)
#endif
constexpr uint64_t COUNT_GRID = CDBG_COUNT_GRID;
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

// The environment knobs (test switches that force the rare paths, dev switches of the A/B scripts) are read ONCE, when the
// context is created: no getenv on the per-step host path, and a job cannot change behaviour half way.
struct Knobs {
    std::vector<std::pair<std::string, std::string>> kv;
    void snapshot() {
        static const char* const NAMES[] = { "CDBG_MAX_PASSES", "CDBG_CW_TIER3", "CDBG_CW_TIER2", "CDBG_SCAN_TWO_LEVEL", "CDBG_COUNT_MAX_SUB", "CDBG_VAR_RESAMPLE", "CDBG_EXACT_NO_CUR32", "CDBG_SOLID_FIRST_TINY", "CDBG_PREWARM_MIN_BYTES", "CDBG_NO_PREWARM", "CDBG_DEBUG_SEGHIST", "CDBG_FAST_MAX_RECORDS", "CDBG_FAST_SKIP2_Q8", "CDBG_FAST_SKIP_Q8", "CDBG_FORCE_MULTI", "CDBG_GENERIC_SCAN", "CDBG_GLUE_LOG", "CDBG_GLUE_RANK", "CDBG_GLUE_REPLICATED", "CDBG_GLUE_TABLE", "CDBG_JOIN_LOG_JB", "CDBG_NO_COUNT_TIER2", "CDBG_NO_SIFT", "CDBG_NO_SPLIT", "CDBG_PART_CAP", "CDBG_REPAIR_MAX_PASSES", "CDBG_SCAN_MODE", "CDBG_STAGE_BYTES", "CDBG_STREAM_BATCH_TILES", "CDBG_STREAM_MIN_BYTES", "CDBG_VAR_SCALE", "CDBG_WALK_MAX", "CDBG_DEFER_SLICES", "CDBG_DEFER_CAP", "CDBG_PLACE_GRID", "CDBG_BIG_ONE_WG" };
        for (const char* n : NAMES) if (const char* e = getenv(n)) kv.emplace_back(n, e);
    }
    const char* get(const char* name) const { for (const auto& p : kv) if (p.first == name) return p.second.c_str(); return nullptr; }
};

struct cdbg_ctx {
    cdbg_params prm{};
    Knobs knobs;
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
    // Zero-copy ingest for multi-threaded parsers (cdbg_stage_acquire / cdbg_stage_commit): pinned buffers handed out to the caller's
    // threads, filled in place, committed in any order -- every commit appends to the device text; one mutex orders the appends
    // (and every other push).  state: 0 free, 1 held by a caller, 2 copy in flight
    struct Stage { uint8_t* p = nullptr; hipEvent_t ev{}; int state = 0; };
    static constexpr int MAX_STAGES = 64;
    std::mutex ingest_mu; std::vector<Stage> stages;
    // Pre-warm (cdbg_expect_input on a large input): fresh device memory costs 40 - 70 ms per GB to obtain, seconds for the text, the
    // record region and the solid arrays of a job -- a background thread obtains them, sized from the announced volume, while the
    // caller parses.  prewarm_reads / prewarm_region: 1 while the thread is still about to publish c->reads / c->records (the ingest
    // path waits for the text buffer instead of allocating a second one; the streaming scan starts once the region is there).
    std::thread prewarm; std::atomic<int> prewarm_reads{0}, prewarm_region{0};
    uint64_t n_dev = 0;                          // bytes of text already on (or on their way to) the device
    bool reads_final = false;                    // text complete, padded, nbytes set
    // streaming scan (cdbg_expect_input): tiles already scanned while the input was still arriving
    uint64_t expect_bytes = 0, ss_done = 0, ss_spill_cap = 0; uint32_t ss_part_cap = 0; bool ss_on = false;
    int log_np_override = -1;                    // set when a first count showed buckets too full for the LDS compaction tiers
    DBuf<uint8_t> reads; uint64_t nbytes = 0, nbytes_padded = 0;

    DBuf<uint32_t> part_count, spill_part; DBuf<uint64_t> part_off, part_cursor, records, exscan_tmp, spill_recs;
    DBuf<uint64_t> dstats; DBuf<uint32_t> derr;
    // deferred record placement (host_count.h): the record streams of the slices the scan does not place itself, their cursors, the stream the
    // placement kernels run on beside the count stage, and one event per slice (+ [0]: where that stream's work of this step begins)
    DBuf<uint64_t> defer_recs; DBuf<uint32_t> defer_part, defer_count;
    hipStream_t place_stream{}; hipEvent_t place_ev[17] = {}; hipEvent_t scan_ev{}; bool defer_off_once = false;
    DBuf<uint64_t> solid_keys; DBuf<uint32_t> solid_cnt; DBuf<uint64_t> solid_cursor, seg_off; DBuf<uint32_t> seg_n;
    DBuf<uint32_t> big_list, big_count, big_list2, big_count2, retry_list;
    uint64_t n_solid_entries = 0;                // home + traveller solid entries

    DBuf<uint32_t> piece_n; DBuf<uint64_t> piece_kc, piece_boff; DBuf<uint8_t> piece_bases; DBuf<uint64_t> cursors;
    DBuf<uint64_t> glue_keys; DBuf<uint32_t> glue_a, glue_b, glue_conf; uint32_t glue_cap = 0;   // (fallback junction table)
    DBuf<uint32_t> retry_list2;                              // partitions that did not fit the second count tier either
    DBuf<uint32_t> var_cap; DBuf<uint64_t> var_pairs, ovf_words;        // single-pass layout of skewed inputs: region capacities, begin / end of every partition's records
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
    uint64_t unitig_id_base = 0, unitig_id_total = 0;       // job-wide unitig ids of a sharded set (cdbg_link): this rank's first id, the job's unitigs
    DBuf<uint4> rank_a, rank_b; DBuf<uint32_t> rank_flag;
    DBuf<uint4> walk_rec; DBuf<uint32_t> walk_heads, walk_hlen; DBuf<uint64_t> walk_hoff; bool walk_off = false;   // chains walked from their heads (k_walk.h); walk_off: a run of this context had a chain the walk does not take
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
    if (const char* e = c->knobs.get("CDBG_STAGE_BYTES")) c->stage_bytes = std::min<uint64_t>(cdbg_ctx::STAGE_BYTES, std::max<uint64_t>(64, strtoull(e, nullptr, 10)));
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
    for (auto& sg : c->stages) if (sg.p) { (void)hipHostFree(sg.p); (void)hipEventDestroy(sg.ev); }
    c->stages.clear();
    if (c->copy_stream) { (void)hipStreamDestroy(c->copy_stream); c->copy_stream = hipStream_t{}; }
}
// device text with room for `need` bytes: grows by doubling (device-to-device copy of what is already there)
int ingest_reserve(cdbg_ctx* c, uint64_t need) {
    while (c->prewarm_reads.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(200));   // (the pre-warm thread is obtaining the text buffer)
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
// ---- zero-copy staging (include/cdbg.h cdbg_stage_acquire / cdbg_stage_commit); callers hold c->ingest_mu ----
int stage_acquire(cdbg_ctx* c, char** buf, uint64_t* cap) {
    CK(ingest_init(c));
    for (;;) {
        int oldest = -1;
        for (int i = 0; i < (int)c->stages.size(); ++i) {
            cdbg_ctx::Stage& sg = c->stages[i];
            if (sg.state == 2 && hipEventQuery(sg.ev) == hipSuccess) sg.state = 0;       // its copy has landed
            if (sg.state == 0) { sg.state = 1; *buf = (char*)sg.p; *cap = c->stage_bytes; return CDBG_OK; }
            if (sg.state == 2 && oldest < 0) oldest = i;
        }
        // (every buffer is with a caller or on its way.  Beyond 32 buffers -- 1 GB pinned, twice the CLI's parser threads -- a copy that is about to land is waited for
        //  instead of pinning 32 MB more on the ingest path; a caller that HOLDS them all still gets more, up to MAX_STAGES)
        if ((int)c->stages.size() >= 32 && oldest >= 0) { HIPCK(hipEventSynchronize(c->stages[oldest].ev)); continue; }
        if ((int)c->stages.size() < cdbg_ctx::MAX_STAGES) {
            cdbg_ctx::Stage sg;
            if (hipHostMalloc((void**)&sg.p, cdbg_ctx::STAGE_BYTES) != hipSuccess) return fail(CDBG_E_NOMEM, "pinned staging buffer (%llu bytes)", (unsigned long long)cdbg_ctx::STAGE_BYTES);
            if (hipEventCreate(&sg.ev) != hipSuccess) { (void)hipHostFree(sg.p); return fail(CDBG_E_NODEVICE, "hipEventCreate failed (staging buffer)"); }
            sg.state = 1; c->stages.push_back(sg);
            *buf = (char*)sg.p; *cap = c->stage_bytes; return CDBG_OK;
        }
        if (oldest < 0) return fail(CDBG_E_STATE, "cdbg_stage_acquire: all %d staging buffers are held by the caller", cdbg_ctx::MAX_STAGES);
        HIPCK(hipEventSynchronize(c->stages[oldest].ev));
    }
}
int ingest_flush(cdbg_ctx* c);
int stage_commit(cdbg_ctx* c, char* buf, uint64_t n) {
    cdbg_ctx::Stage* sg = nullptr;
    for (auto& x : c->stages) if ((char*)x.p == buf) sg = &x;
    if (!sg || sg->state != 1) return fail(CDBG_E_PARAM, "cdbg_stage_commit: not a buffer handed out by cdbg_stage_acquire");
    if (n > c->stage_bytes) return fail(CDBG_E_PARAM, "cdbg_stage_commit: %llu bytes in a buffer of %llu", (unsigned long long)n, (unsigned long long)c->stage_bytes);
    if (n && base_valid((uint8_t)buf[n - 1])) {                                             // every commit ends a sequence: what follows it in the text is another thread's
        if (n == c->stage_bytes) return fail(CDBG_E_PARAM, "cdbg_stage_commit: a full buffer must end with a separator");
        buf[n++] = '\n';
    }
    if (!n) { sg->state = 0; return CDBG_OK; }
    CK(ingest_flush(c));                                                                    // (bytes of an earlier cdbg_push_text come first)
    CK(ingest_reserve(c, c->n_dev + n));
    HIPCK(hipMemcpyAsync(c->reads.p + c->n_dev, buf, n, hipMemcpyHostToDevice, c->copy_stream));
    HIPCK(hipEventRecord(sg->ev, c->copy_stream));
    sg->state = 2;
    c->n_dev += n;
    if (c->expect_bytes && c->prm.world_size == 1 && !c->force_multi) CK(stream_scan_dispatch(c));
    return CDBG_OK;
}
// text complete: last partial buffer out, all copies done, tail padded with separators
int upload_pending(cdbg_ctx* c) {
    if (c->reads_final) return CDBG_OK;
    if (c->n_dev == 0 && c->pin_fill == 0) {                                   // nothing was pushed
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
    bool on = enabled(); double t0 = now();
    static bool enabled() { static const bool e = getenv("CDBG_HOST_MARKS") != nullptr; return e; }   // (read once per process)
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
    //  Round 4: behind the sifting tier (k_count_fast.h: abundance-min >= 2) an overfull partition costs two reads of its records
    //  instead of the 4096-slot tier and the multi-pass kernel, and twice the partition size again wins: the config-5 share holds
    //  3.7 M minimizer loci, i.e. 0.44 per partition at 2^23 (64 % of the partitions empty) -- 2^22: count 175.8 -> 162.8 ms, step
    //  231.6 -> 222.9 ms; 2^21: 281.6 ms, 2^24: 279.0 ms (profiles/r04_ab_cfg5_partitions.log))
    const bool sifted = W >= 3 && c->prm.abundance_min >= 2;
    const uint64_t target_occ = W >= 3 ? (uint64_t)ts * (sifted ? 12 : 6) / 10 : (uint64_t)ts * CDBG_OCC_NUM / CDBG_OCC_DEN;
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


}  // namespace
