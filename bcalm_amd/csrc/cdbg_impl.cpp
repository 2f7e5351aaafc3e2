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
#include <thread>
#include <atomic>
#include <chrono>
#include <unordered_set>
#include <vector>

#include "k_links.h"
#include "k_verify.h"
#include "k_dglue.h"
#include "k_walk.h"
#include "comm.h"
#include "k_count_fast.h"
#include "k_compact_wave.h"
#include "k_split.h"

using namespace cdbg;

#include "host_ctx.h"
#include "host_count.h"
#include "host_compact.h"
#include "host_exchange.h"
#include "host_glue_sharded.h"
#include "host_glue.h"


// =======================================================================================
// C ABI
// =======================================================================================
extern "C" {

const char* cdbg_last_error(void) { return g_err.c_str(); }

int cdbg_create(const cdbg_params* p, cdbg_ctx** out) {
    if (!p || !out) return fail(CDBG_E_PARAM, "null argument");
    *out = nullptr;
    if (p->k < 3 || p->k > 32 * CDBG_MAX_W - 1) return fail(CDBG_E_PARAM, "kmer-size %d out of range (3..%d)", p->k, 32 * CDBG_MAX_W - 1);
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
    c->knobs.snapshot();
    // words per k-mer: the reference's span rule k < 32 W (README.md:91-99, Integer::apply at src/bcalm_1.cpp:95); the top word of a
    // multi-word key therefore always keeps its two top bits free for the slot-claim protocol (k_count.h), for even k as well
    c->k = p->k; c->W = p->k / 32 + 1;
    c->rank_bits = 0; while ((1 << c->rank_bits) < ws) ++c->rank_bits;
    if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return fail(CDBG_E_NODEVICE, "hipStreamCreate failed"); }
    *out = c;
    return CDBG_OK;
}

void cdbg_destroy(cdbg_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    prewarm_join(c);
    ingest_release(c);
#ifndef CDBG_HOSTSIM
    if (c->rccl) { c->rccl->destroy(); delete c->rccl; c->rccl = nullptr; }
#endif
    (void)hipStreamSynchronize(c->stream);               // (the buffers go back to the process's pool: nothing may still be writing them)
    if (c->place_stream) { (void)hipStreamSynchronize(c->place_stream); (void)hipStreamDestroy(c->place_stream); }
    for (hipEvent_t& e : c->place_ev) if (e) (void)hipEventDestroy(e);
    if (c->scan_ev) (void)hipEventDestroy(c->scan_ev);
    (void)hipStreamDestroy(c->stream);
    delete c;
}
int cdbg_abi_version(void) { return CDBG_ABI_VERSION; }
uint64_t cdbg_stats_sizeof(void) { return sizeof(cdbg_stats_t); }
int cdbg_release_cached(void) {
    int cur = 0; (void)hipGetDevice(&cur);
    for (int d = 0; d < 64; ++d) {
        bool any; { std::lock_guard<std::mutex> g(dev_pool().mu); any = !dev_pool().blocks[d].empty(); }
        if (any) { (void)hipSetDevice(d); dev_pool().drain(d); }
    }
    (void)hipSetDevice(cur);
    return CDBG_OK;
}

int cdbg_push_reads(cdbg_ctx* c, const char* bases, const uint64_t* offsets, uint64_t n_reads) {
    if (!c || !bases || !offsets) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    if (c->stage != 0 || c->reads_final) return fail(CDBG_E_STATE, "reads must be pushed before the first stage");
    std::lock_guard<std::mutex> lk(c->ingest_mu);
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
    std::lock_guard<std::mutex> lk(c->ingest_mu);
    CK(ingest_append(c, text, nbytes));
    CK(ingest_append(c, "\n", 1));
    return CDBG_OK;
}
int cdbg_stage_acquire(cdbg_ctx* c, char** buf, uint64_t* capacity) {
    if (!c || !buf || !capacity) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);                 // (any host thread: one parser thread per slice of the input)
    if (c->stage != 0 || c->reads_final) return fail(CDBG_E_STATE, "reads must be pushed before the first stage");
    std::lock_guard<std::mutex> lk(c->ingest_mu);
    return stage_acquire(c, buf, capacity);
}
int cdbg_stage_commit(cdbg_ctx* c, char* buf, uint64_t nbytes) {
    if (!c || !buf) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);
    if (c->stage != 0 || c->reads_final) return fail(CDBG_E_STATE, "reads must be pushed before the first stage");
    std::lock_guard<std::mutex> lk(c->ingest_mu);
    return stage_commit(c, buf, nbytes);
}
int cdbg_expect_input(cdbg_ctx* c, uint64_t text_bytes) {
    if (!c) return fail(CDBG_E_PARAM, "null context");
    // (a text that is complete already -- cdbg_generate_reads -- must not be announced again: the pre-warm thread would swap the buffer that holds it)
    if (c->stage != 0 || c->n_dev || c->pin_fill || c->reads_final) return fail(CDBG_E_STATE, "cdbg_expect_input must precede the first push");
    c->expect_bytes = text_bytes;
    // a large input on one GPU: obtain the text buffer, the record region and the solid arrays in the background (host_count.h prewarm_run)
    uint64_t min_bytes = 1ull << 30;
    if (const char* e = c->knobs.get("CDBG_PREWARM_MIN_BYTES")) min_bytes = strtoull(e, nullptr, 10);
    if (c->prm.world_size == 1 && !c->force_multi && text_bytes >= min_bytes && !c->knobs.get("CDBG_NO_PREWARM") && !c->prewarm.joinable()) {
        (void)hipSetDevice(c->prm.device_id);
        configure(c, text_bytes);                          // (what the streaming scan will choose from the same number)
        c->prewarm_reads = 1; c->prewarm_region = 1;
        // (no exception may cross the C ABI, and the flags must not stay set without a thread behind them: ingest_reserve waits on them)
        try { c->prewarm = std::thread(prewarm_run, c); } catch (...) { c->prewarm_reads = 0; c->prewarm_region = 0; }
    }
    return CDBG_OK;
}
int cdbg_generate_reads(cdbg_ctx* c, uint64_t first_read, uint64_t n_reads, uint64_t total_reads, uint64_t read_len, int cfg) {
    if (!c) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    if (c->stage != 0 || c->reads_final || c->n_dev || c->pin_fill) return fail(CDBG_E_STATE, "reads already present");
    if (!n_reads || !read_len || total_reads < n_reads) return fail(CDBG_E_PARAM, "bad synthetic read set");
    prewarm_join(c);                                        // (an announced input that is generated after all: the pre-warm thread owns c->reads until it is done)
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

#if CDBG_MAX_W >= 8
#define DISPATCH_WIDE(fn, ...) case 5: return fn<5>(__VA_ARGS__); case 6: return fn<6>(__VA_ARGS__); case 7: return fn<7>(__VA_ARGS__); case 8: return fn<8>(__VA_ARGS__);
#else
#define DISPATCH_WIDE(fn, ...)
#endif
#define DISPATCH_WA(fn, ...)                                         \
    switch (c->W) {                                                  \
        case 1: return fn<1>(__VA_ARGS__);                           \
        case 2: return fn<2>(__VA_ARGS__);                           \
        case 3: return fn<3>(__VA_ARGS__);                           \
        case 4: return fn<4>(__VA_ARGS__);                           \
        DISPATCH_WIDE(fn, __VA_ARGS__)                               \
        default: return fail(CDBG_E_PARAM, "k-mers of %d words: rebuild with CDBG_MAX_W", c->W);   \
    }
#define DISPATCH_W(fn) DISPATCH_WA(fn, c)
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
int cdbg_unitig_id_base(cdbg_ctx* c, uint64_t* first_id, uint64_t* total) {
    if (!c || !first_id) return fail(CDBG_E_PARAM, "null argument");
    if (!c->linked) return fail(CDBG_E_STATE, "cdbg_unitig_id_base before cdbg_link");
    *first_id = c->unitig_id_base; if (total) *total = c->unitig_id_total; return CDBG_OK;
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
    prewarm_join(c);
    if (c->place_stream) (void)hipStreamSynchronize(c->place_stream);   // (a count stage that returned with an error may have left its placement stream busy)
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
    c->tr = *t; c->have_tr = true; c->tr_ordered = false; c->force_multi = c->knobs.get("CDBG_FORCE_MULTI") != nullptr; return CDBG_OK;
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
    c->tr = c->rccl->transport(); c->have_tr = true; c->tr_ordered = true; c->force_multi = c->knobs.get("CDBG_FORCE_MULTI") != nullptr;
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
    DISPATCH_WA(verify_impl, c, out)
}
int cdbg_verify_edges(cdbg_ctx* c, uint64_t out[4]) {
    if (!c || !out) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);
    DISPATCH_WA(verify_edges_impl, c, out)
}
int cdbg_verify_unitigs(cdbg_ctx* c, const char* bases, const uint64_t* offsets, uint64_t n_unitigs, uint64_t out[12]) {
    if (!c || !out || !offsets || (!bases && n_unitigs)) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);
    DISPATCH_WA(verify_unitigs_impl, c, bases, offsets, n_unitigs, out)
}
int cdbg_stats(cdbg_ctx* c, cdbg_stats_t* out) {
    if (!c || !out) return fail(CDBG_E_PARAM, "null argument");
    (void)hipSetDevice(c->prm.device_id);                 // (the caller may be any host thread: one thread per GPU in the CLI)
    *out = c->st; return CDBG_OK;
}

}  // extern "C"
