// host_count.h -- libcdbg.so, host side of stage 1: record placement (capped / estimated /
// exact regions), the streaming scan, the record exchange of sharded reads, the count tiers.  Included by cdbg_impl.cpp only.
#pragma once

namespace {

// the register-window scan (k_scan_fast.h): k <= 63 with a window of at most SCANF_WNMAX keys, or k <= 127 through its two-level window
// minimum (windows of 17 .. 111 keys); the LDS-doubling scan of k_scan.h otherwise
inline bool scan_is_fast(const cdbg_ctx* c) {
    const int wn = c->k - c->m;
    if (c->knobs.get("CDBG_GENERIC_SCAN")) return false;          // (test knob: the generic kernel for every shape)
    return (c->k <= 63 && wn <= SCANF_WNMAX) || (c->k <= 127 && wn > SCANF_WNMAX && wn >= 17);
}
#ifndef CDBG_SCAN_GEN
#define CDBG_SCAN_GEN 8                                   // workgroups per resident place of the register-window scan: 1 -> 4 / 16: scan 69.3 -> 66.0 / 65.6 ms at config 3 (profiles/r04_ab_cfg3_scan_grid.log)
#endif
// the scan kernel for this k / m / mode on the context's stream (persistent grid: resident workgroups)
// (dry: nothing is launched; the return value is the number of workgroups the launch would have -- the segments of a deferred record stream)
template <int W, int MODE>
uint64_t launch_scan_mode(cdbg_ctx* c, ScanParams& sp, uint64_t grid, bool dry = false) {
    hipStream_t s = c->stream;
    const bool fast_scan = scan_is_fast(c);
    sp.n_tiles = grid;
    if (grid == 0) return 0;                                 // (a rank without reads)
#define CDBG_SCAN_GO(KERN, FALLBACK, GEN)                                                                                        \
    { const uint64_t g_ = std::min<uint64_t>(grid, resident_grid(KERN, SCAN_THREADS, FALLBACK) * (GEN));                         \
      if (!dry) CDBG_LAUNCH(KERN, g_, SCAN_THREADS, s, sp);                                                                      \
      return g_; }
    if (fast_scan && (c->k - c->m > SCANF_WNMAX || (c->k - c->m >= 17 && c->knobs.get("CDBG_SCAN_TWO_LEVEL"))))   // the two-level window minimum (k_scan_fast.h, WNT = -1): k <= 127 with a long minimizer window (dev knob: any window of 17 keys and more)
        CDBG_SCAN_GO((k_scan_fast<W, MODE, -1>), SCANF_GRID, CDBG_SCAN_GEN)
    // compile-time minimizer windows (k - m): the k = 31 family m = 16 .. 12 and k = 55, m = 16 (config 4)
#define CDBG_SCAN_WNT(WW, WNT_)                                                                                                  \
    if (fast_scan && W == WW && c->k - c->m == WNT_) CDBG_SCAN_GO((k_scan_fast<W, MODE, W == WW ? WNT_ : 0>), SCANF_GRID, CDBG_SCAN_GEN)
    CDBG_SCAN_WNT(1, 15) CDBG_SCAN_WNT(1, 16) CDBG_SCAN_WNT(1, 17) CDBG_SCAN_WNT(1, 18) CDBG_SCAN_WNT(1, 19) CDBG_SCAN_WNT(2, 39)
#undef CDBG_SCAN_WNT
    if (fast_scan) CDBG_SCAN_GO((k_scan_fast<W, MODE, 0>), SCANF_GRID, CDBG_SCAN_GEN)
    CDBG_SCAN_GO((k_scan<W, MODE>), SCAN_GRID, 1)
#undef CDBG_SCAN_GO
}
inline uint64_t scan_tile_bytes(const cdbg_ctx* c) { return scan_is_fast(c) ? (uint64_t)SCANF_TILE : (uint64_t)SCAN_TILE; }
void scan_params_base(cdbg_ctx* c, ScanParams& sp) {
    sp.reads = c->reads.p; sp.nbytes = c->nbytes; sp.nbytes_padded = c->nbytes_padded;
    sp.k = c->k; sp.m = c->m; sp.log_np = c->log_np; sp.rank_bits = c->rank_bits; sp.rank = c->prm.rank;
    sp.part_count = c->part_count.p; sp.part_cursor = c->part_cursor.p; sp.records = nullptr; sp.stats = c->dstats.p;
    sp.tile_stride = 1; sp.tile_offset = 0; sp.error = c->derr.p;
}
// capacity of a partition region from a sampled histogram (single-pass capped layout)
// (streaming: the scan that runs while a FILE is still being read -- the CLI's first and only job, whose region is fresh device memory at 40 - 70 ms per GB: there
//  the capacity is mean + 12 sqrt(mean), + 2.2 sigma: 43 GB instead of 53 at config 3, and the 1.4 % of the partitions that spill are repaired and counted from
//  their gathered copies -- a few ms of GPU time against half a second of hipMalloc)
void capped_capacities(const cdbg_ctx* c, double mean, uint64_t NPL, uint32_t& part_cap, uint64_t& spill_cap, bool streaming = false) {
    // A partition's records are the sum over its minimizer loci (~ coverage records each): the spread is ~ 5.5 sqrt(mean), so
    // mean + 20 sqrt(mean) is + 3.6 sigma -- a few hundred of config 3's 4 M partitions spill and are repaired.  Round 5: was
    // 2.5 mean + 8 sqrt(mean) (75 GB at config 3, now 53 GB): the scan does not care (same-box A/B, CDBG_PART_CAP 1128 / 768 / 640:
    // scan 67.4 / 65.3 / 69.1 ms), and fresh device memory costs the CLI 40 ms per GB (profiles/r05_micro_alloc.log).
    part_cap = (uint32_t)(mean + (streaming ? 12.0 : 20.0) * std::sqrt(mean + 1.0) + 16.0);
    part_cap = (part_cap + 7u) & ~7u;
    if (const char* e = c->knobs.get("CDBG_PART_CAP")) part_cap = (uint32_t)std::max(1, atoi(e));   // test knob: force spills
    spill_cap = std::max<uint64_t>((uint64_t)(mean * (double)NPL / 32.0), 65536);
}

// ---- grids of the count stage's launches (every persistent workgroup of every launch may strand one partly used output chunk) ----
#ifndef CDBG_SIFT_GRID
#define CDBG_SIFT_GRID (256 * 16)
#endif
#ifndef CDBG_T2_GRID
#define CDBG_T2_GRID 256
#endif
#ifndef CDBG_MP_GRID_W
#define CDBG_MP_GRID_W 256
#endif
#ifndef CDBG_MP_GRID_1
#define CDBG_MP_GRID_1 (256 * 48)                          // (one-word multi-pass kernel on the hostile line: count 92.4 -> 86.9 ms with 12288 instead of 3072 workgroups;
#endif                                                   //  2048 instead of 256 for the wider kernels: neutral at k = 55, + 3 ms at k = 127 -- profiles/r04_ab_cfg3_count_grid.log)
#ifndef CDBG_MP1_WIDE
#define CDBG_MP1_WIDE 1                                    // (round 5: the multi-pass kernel of one-word k-mers with the 8192-slot table and 1024 threads as well, as the wider k-mers have it -- half the passes at 16 waves per CU: hostile line count 89.9 -> 80.6 ms; 0: 4096 slots, 512 threads)
#endif
constexpr uint64_t SIFT_GRID = CDBG_SIFT_GRID, T2_GRID = CDBG_T2_GRID, MP_GRID_W = CDBG_MP_GRID_W, MP_GRID_1 = CDBG_MP_GRID_1;
// (launches of the stage: one-pass tier 1 of COUNT_GRID workgroups, tier 2 of at most SIFT_GRID, multi-pass retry, spill repair, HBM tables)
inline uint64_t count_solid_slack(uint64_t NPL, uint32_t slices = 1) {   // slices: launches of the one-pass tier (deferred placement: one per slice of the partition space)
    // (no launch has more workgroups than partitions: a small input -- a test, a shard of few partitions -- reserves megabytes, not the 3 - 15 GB
    //  that the grids' constants alone came to; fresh device memory costs 40 - 70 ms per GB)
    auto lim = [NPL](uint64_t grid) { return std::min<uint64_t>(NPL, grid); };
    return 4096 + ((uint64_t)slices * std::min<uint64_t>(NPL / slices, COUNT_GRID) + 3 * lim(PERSISTENT_GRID) + lim(SIFT_GRID) + lim(T2_GRID) + 2 * lim(MP_GRID_1 > MP_GRID_W ? MP_GRID_1 : MP_GRID_W) + lim(768) + 5) * (uint64_t)COUNT_CHUNK;
}
// first attempt at the solid arrays' size (see count_impl): a third of the bound of one entry per abundance-min member k-mers
inline uint64_t count_solid_first_cap(uint64_t members, int amin, uint64_t NPL, uint32_t slices = 1) {
    return std::max<uint64_t>(members / (uint64_t)std::max(1, amin) / 3, 1u << 16) + count_solid_slack(NPL, slices);
}

// ---------------------------------------------------------------------------------------
// Pre-warm thread (cdbg_expect_input; host_ctx.h): the text buffer, the record region and the solid arrays, sized from the announced
// volume a little above what the stages will ask for (a DBuf keeps an allocation that is large enough), obtained while the caller parses
// ---------------------------------------------------------------------------------------
void prewarm_run(cdbg_ctx* c) {
    (void)hipSetDevice(c->prm.device_id);
    const uint64_t bytes = c->expect_bytes;
    { DBuf<uint8_t> b; const uint64_t cap = std::max<uint64_t>(256ull << 20, bytes + bytes / 64 + (8ull << 20));
      if (c->reads.cap < cap && !c->n_dev && b.alloc(cap, false) == CDBG_OK) c->reads.swap(b); }   // (a context that is re-run keeps what it has)
    c->prewarm_reads.store(0, std::memory_order_release);
    const int RW = 2 * c->W;
    const uint64_t NPL = c->n_local_parts;
    {   // records: ~2 runs of junctions per window of k - m + 1 minimizer positions
        // (the density of random minimizers, 2 per window + 1; sequencing reads stay ~10 % below it -- separators, records cut at read ends.  An input that
        //  exceeds it makes the streaming scan obtain a larger region itself: slower, not wrong)
        const double rec = (double)bytes * 2.0 / (double)(c->k - c->m + 2);
        uint32_t part_cap = 0; uint64_t spill_cap = 0; capped_capacities(c, rec / (double)NPL, NPL, part_cap, spill_cap, true);
        DBuf<uint64_t> r;
        size_t fr = 0, tot = 0;
        const bool fits = hipMemGetInfo(&fr, &tot) == hipSuccess && (double)part_cap * (double)NPL * RW * 8.0 < 0.6 * (double)fr;
        if (fits && c->records.cap < (uint64_t)part_cap * NPL * RW && r.alloc((uint64_t)part_cap * NPL * RW, false) == CDBG_OK) c->records.swap(r);
    }
    c->prewarm_region.store(0, std::memory_order_release);
    {   // solid arrays (touched by cdbg_count only, which joins this thread first)
        const uint64_t cap = count_solid_first_cap(bytes + bytes / 8, c->prm.abundance_min, NPL);
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && (double)cap * (8.0 * c->W + 4.0) < 0.5 * (double)fr) {
            (void)c->solid_keys.alloc(cap * (uint64_t)c->W, false); (void)c->solid_cnt.alloc(cap, false);
        }
    }
}
void prewarm_join(cdbg_ctx* c) { if (c->prewarm.joinable()) c->prewarm.join(); c->prewarm_reads = 0; c->prewarm_region = 0; }

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
    if (!c->ss_on && c->prewarm_region.load(std::memory_order_acquire)) return CDBG_OK;   // (the pre-warm thread is obtaining the record region: a later push starts the scan)
    if (!c->ss_on) {
        // (test knobs: CDBG_STREAM_MIN_BYTES / CDBG_STREAM_BATCH_TILES shrink the thresholds to simulator sizes)
        const char* emin = c->knobs.get("CDBG_STREAM_MIN_BYTES");
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
        capped_capacities(c, mean, NPL, c->ss_part_cap, c->ss_spill_cap, true);
        // (the same budget as count_impl -- what the card has free, what this context and the pool would hand back, less a reserve for the
        //  stages that follow; a region that cannot be had switches streaming off instead of failing the push)
        { size_t fr = 0, tot = 0;
          const double budget = (hipMemGetInfo(&fr, &tot) == hipSuccess && tot) ? (double)fr + (double)c->records.cap * 8.0 + (double)dev_pool().held[DevPool::device()] - 0.15 * (double)tot : 200e9;
          if ((double)c->ss_part_cap * (double)NPL * RW * 8.0 > budget) { c->expect_bytes = 0; return CDBG_OK; } }   // would not fit: no streaming
        if (c->records.alloc((uint64_t)c->ss_part_cap * NPL * RW, false) != CDBG_OK || c->spill_recs.alloc(c->ss_spill_cap * RW, false) != CDBG_OK ||
            c->spill_part.alloc(c->ss_spill_cap, false) != CDBG_OK) { c->expect_bytes = 0; return CDBG_OK; }
        HIPCK(hipMemsetAsync(c->part_count.p, 0, NPL * sizeof(uint32_t), s));
        HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
        c->ss_on = true; c->ss_done = 0;
    }
    const uint64_t TB = scan_tile_bytes(c);
    const uint64_t tiles_now = landed > TB + 8192 ? (landed - 8192) / TB : 0;      // tiles whose halo has landed as well
    const char* ebt = c->knobs.get("CDBG_STREAM_BATCH_TILES");
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
    switch (c->W) {
        case 1: return stream_scan_advance<1>(c); case 2: return stream_scan_advance<2>(c); case 3: return stream_scan_advance<3>(c); case 4: return stream_scan_advance<4>(c);
#if CDBG_MAX_W >= 8
        case 5: return stream_scan_advance<5>(c); case 6: return stream_scan_advance<6>(c); case 7: return stream_scan_advance<7>(c); case 8: return stream_scan_advance<8>(c);
#endif
        default: return fail(CDBG_E_PARAM, "k-mers of %d words: rebuild with CDBG_MAX_W", c->W);
    }
}

// ---------------------------------------------------------------------------------------
// Deferred record placement (round 6).  The scan is bound by its memory REQUESTS (two per record: 1.6 G returning atomics beside 1.6 G partial-sector
// stores, 66 ms at config 3 with a third of the VALU busy), the one-pass count kernel by VALU issue (57 ms, little traffic) -- and the two ran one
// after the other.  Now the partition space is cut into S slices: the scan places slice 0's records as before and APPENDS the others to S - 1 streams
// (coalesced 16-byte stores into one segment per scan workgroup, cursors in LDS: no device atomic); k_place (k_scan.h) scatters stream q on a second HIP
// stream while k_count_fast counts slice q - 1 on the first (events between them).  The placement kernel needs no LDS and two wave slots per CU: the
// count's three workgroups per CU leave eight.
// Measured before it was built (profiles/r06_ab_overlap_place_vs_count.log): 0.8 G records placed beside the count cost the pair 70 ms against
// 57 + 35 = 92 one after the other -- with ONE-WAVE workgroups spread evenly over the CUs: larger workgroups or more of them land unevenly once the count's
// workgroups are on the chip, and a CU with 12 placement waves is the step's tail.
// ---------------------------------------------------------------------------------------
#ifndef CDBG_PLACE_GRID
#define CDBG_PLACE_GRID (256 * 2)                          // (one-wave workgroups: two per CU.  256: too few requests in flight; 384, 768: land unevenly; 1024: fine for one launch on an idle chip, uneven for the launches behind it)
#endif
// The slices of this step and their record streams.  w[q]: sixteenths of the partition space in slice q (slice 0: placed by the scan); cap[q]: records a
// SEGMENT of stream q holds (one segment per workgroup of the scan); base[q]: first slot of stream q.  n == 1: no deferral.
struct DeferPlan { uint32_t n = 1; uint32_t w[16] = {}; uint32_t cap = 0; uint64_t slots = 0; uint64_t nseg = 0; };
// Measured at config 3 (profiles/r06_ab_defer_slices.log; same box, no deferral: 170.5 ms with the scan at 71): two halves 154.5 ms (scan 46, count stage 66.5 with
// the placement's 36 ms inside it); four quarters 155.6 (41 + 72.5: the placement of 1.2 G records, 58 ms, is what the last quarter's count waits for); patterns with a
// shrinking tail ("4,4,4,2,1,1") lost: every further placement launch lands on a chip full of count workgroups, unevenly, and runs at half its rate.
// With the scan's loads pipelined over the tiles (k_scan_fast.h; profiles/r06_ab_scan_pipelined.log) the scan of a quarter takes 36 ms instead of 41 and single
// steps of "4" or "6,6,4" reach 150 ms -- but over the 25 steps of bench.py, in one process: "8,8" 150.0 ms, "4" 153.7, "6,6,4" 154.4: one placement launch,
// onto a chip the scan has just left, lands evenly every time; a second and third one do not.
#ifndef CDBG_DEFER_PATTERN
#define CDBG_DEFER_PATTERN "8,8"
#endif
inline DeferPlan defer_plan(cdbg_ctx* c, int W, double mean, uint64_t NPS, int RW, uint64_t nseg, double budget_left) {
    DeferPlan d; d.nseg = nseg;
    // one-word k-mers only: the scans of wider k-mers are bound by their instructions, not by their requests (config-4 / -5 shares: 232 -> 239 / 198 -> 204 ms with two slices)
    const char* pat = c->knobs.get("CDBG_DEFER_SLICES");
    if (pat == nullptr) pat = W == 1 ? CDBG_DEFER_PATTERN : "0";
    if (c->defer_off_once) { c->defer_off_once = false; return d; }
    uint32_t n = 0, sum = 0, w[16];
    if (strchr(pat, ',') != nullptr) {                       // sixteenths per slice
        for (const char* q = pat; *q && n < 16; ) { w[n] = (uint32_t)std::max(0, atoi(q)); sum += w[n]; ++n; while (*q && *q != ',') ++q; if (*q == ',') ++q; }
    } else {                                                 // a number: that many equal slices (2, 4, 8, 16)
        uint32_t S = (uint32_t)std::max(0, atoi(pat)); while (S & (S - 1)) S &= S - 1;
        if (S >= 2 && S <= 16) { n = S; for (uint32_t q = 0; q < S; ++q) w[q] = 16 / S; sum = 16; }
    }
    for (uint32_t q = 0; q < n; ++q) if (w[q] == 0) return d;
    if (n < 2 || sum != 16 || NPS < 16 * 64 || nseg == 0) return d;
    const double est = mean * (double)NPS;                   // records of the step
    uint32_t wmax = 0; for (uint32_t q = 1; q < n; ++q) wmax = std::max(wmax, w[q]);
    // (the persistent workgroups take equal shares of the tiles: a segment's share is est * w / 16 / nseg with a spread of a few sqrt;
    //  a segment that fills up loses nothing -- the scan places the record itself)
    const double share = est * (double)wmax / 16.0 / (double)nseg;
    uint64_t cap = (uint64_t)(share * 1.05 + 8.0 * std::sqrt(share + 1.0)) + 64;
    if (const char* e = c->knobs.get("CDBG_DEFER_CAP")) cap = (uint64_t)std::max(1, atoi(e));   // test knob: segments that fill up
    cap = (cap + 3) & ~3ull;
    if (cap >= (1ull << 31)) return d;
    const uint64_t slots = (uint64_t)(n - 1) * nseg * cap;
    if ((double)slots * (RW * 8.0 + 4.0) > budget_left) return d;
    d.cap = (uint32_t)cap;
    d.n = n; d.slots = slots; for (uint32_t q = 0; q < n; ++q) d.w[q] = w[q];
    return d;
}
template <int W>
int defer_launch_places(cdbg_ctx* c, const ScanParams& sp) {
    hipStream_t s = c->stream;
    if (!c->place_stream) HIPCK(hipStreamCreateWithFlags(&c->place_stream, hipStreamNonBlocking));   // (non-blocking: the host's reads behind the scan must not wait for the placement)
    if (!c->scan_ev) HIPCK(hipEventCreate(&c->scan_ev));
    for (uint32_t q = 0; q < sp.defer_slices; ++q) if (!c->place_ev[q]) HIPCK(hipEventCreate(&c->place_ev[q]));
    HIPCK(hipEventRecord(c->scan_ev, s));
    HIPCK(hipStreamWaitEvent(c->place_stream, c->scan_ev, 0));
    HIPCK(hipEventRecord(c->place_ev[0], c->place_stream));
    uint64_t grid = CDBG_PLACE_GRID;
    if (const char* e = c->knobs.get("CDBG_PLACE_GRID")) grid = (uint64_t)std::max(1, atoi(e));
    for (uint32_t q = 1; q < sp.defer_slices; ++q) {
        PlaceParams pp{ sp, q };
        CDBG_LAUNCH(k_place<W>, grid, 64, c->place_stream, pp);
        HIPCK(hipEventRecord(c->place_ev[q], c->place_stream));
    }
    return CDBG_OK;
}

template <int W>
int count_impl(cdbg_ctx* c) {
    constexpr int RW = RecFmt<W>::RW;
    constexpr int TS = Cfg<W>::TSC;
    const bool multi_ctx = c->prm.world_size > 1 || c->force_multi;
    const int world = c->prm.world_size;
    if (multi_ctx && !c->have_tr) return fail(CDBG_E_STATE, "world_size %d but no transport: call cdbg_comm_init_rccl or cdbg_set_transport first", world);
    prewarm_join(c);
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
    const bool fast_scan = scan_is_fast(c);
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
    // bytes a record region may take: what the card has free, plus what this context's region of the step before and the process's
    // pool would hand back, less a reserve for the stages that follow (the card's size is asked, not assumed)
    auto region_budget = [&]() -> double {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess || !tot) return 200e9;
        return (double)fr + (double)c->records.cap * 8.0 + (double)dev_pool().held[DevPool::device()] - 0.15 * (double)tot;
    };
    bool capped = tiles > 8192;
    if (const char* e = c->knobs.get("CDBG_SCAN_MODE")) { if (!strcmp(e, "exact")) capped = false; else if (!strcmp(e, "capped")) capped = true; }
    if (tiles == 0) capped = false;                          // (a rank without reads: nothing to sample)
    // var: ONE pass into regions of their own size per partition, estimated from a denser sample -- what a skewed input gets instead
    // of the exact two-pass layout (CDBG_SCAN_MODE=var: test knob)
    bool var = false;
    if (const char* e = c->knobs.get("CDBG_SCAN_MODE")) { if (!strcmp(e, "var") && tiles && !multi) { var = true; capped = false; } }
    uint64_t n_records = 0, hs[2] = {0, 0};
    uint32_t part_cap = 0; uint64_t n_spill = 0; bool packed_exact = false;
    uint64_t n_spilled_parts = 0;                            // partitions whose region overflowed (capped mode): counted from gathered copies
    Timer t;
    uint64_t spill_cap = 0;
    DeferPlan dp; uint32_t& defer_S = dp.n;                   // deferred placement: slices of the partition space (1: off) and their streams
    OvfParams var_op{};                                      // skewed inputs: the overflow-region layout (its finishing pass runs slice by slice under deferred placement)
    bool deferred_open = false;                              // placement kernels of this step are (or may be) still running: the record count and the spill list are not final
    uint64_t sample_ns = 0, sample_stride = 0, sample_recs = 0;   // the capped path's sampled histogram, when it ran (still in part_count)
    if (c->ss_on) capped = true;                             // tiles [0, ss_done) were scanned while the input was arriving
    auto defer_fill_params = [&](ScanParams& q) {            // the slices and streams of dp as the kernels see them
        uint32_t lg = 0; while ((1ull << lg) < NPS) ++lg;
        q.defer_slices = defer_S; q.defer_shift = lg - 4; q.defer_nseg = (uint32_t)dp.nseg; q.defer_map = 0;
        for (uint32_t sl = 0, x = 0; sl < defer_S; ++sl) for (uint32_t i = 0; i < dp.w[sl]; ++i, ++x) q.defer_map |= (uint64_t)sl << (4 * x);
        q.defer_seg_cap = dp.cap;
        q.defer_recs = c->defer_recs.p; q.defer_part = c->defer_part.p; q.defer_count = c->defer_count.p;
    };
    // repair: gather region + spilled records of each spilled partition into one contiguous run (k_count.h); the one-pass count kernels
    // leave such a partition alone (fill > capacity) and the repair launch of the count stage counts the gathered copy
    auto repair_spills = [&](const uint64_t* ovf) -> int {
        RepairParams rp{};
        rp.records = c->records.p; rp.spill_recs = c->spill_recs.p; rp.spill_part = c->spill_part.p; rp.n_spill = n_spill;
        rp.part_fill = c->part_count.p; rp.npl = NPL; rp.part_cap = part_cap; rp.RW = RW; rp.ovf = ovf;
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
        return CDBG_OK;
    };
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
            sample_ns = ns; sample_stride = stride; sample_recs = sample_records;
            // the fullest partition of the sample: a skewed input (repeats, low complexity, coverage peaks) puts far more
            // into some partitions than any capacity covers; the capped pass would then hammer a few fill counters and spill
            // (145 ms at the hostile config-3 line before falling back) -- such inputs go straight to the exact two-pass layout
            HIPCK(hipMemsetAsync(c->dstats.p + 31, 0, sizeof(uint64_t), s));
            CDBG_LAUNCH(k_max_u32, std::min<uint64_t>((NPS + 255) / 256, 4096), 256, s, (const uint32_t*)c->part_count.p, NPS, c->dstats.p + 31);
            uint64_t sample_max = 0; CK(read_u64(c->dstats.p + 31, &sample_max));
            CK(t.stop(&c->st.ms_scan_hist));
            const double mean = (double)sample_records * (double)tiles / (double)ns / (double)NPS;
            capped_capacities(c, mean, NPS, part_cap, spill_cap);
            if ((double)part_cap * (double)NPS * RW * 8.0 > region_budget()) fits = false;   // would not fit: use the exact layout
            // (at least 32 sampled records in that partition: with a mean of a few records per partition -- long reads, k = 127 --
            //  the sampled maximum is Poisson noise, and scaling it up sent the config-5 share through two passes: 598 -> 662 ms)
            else if (sample_max >= 32 && (double)sample_max * (double)tiles / (double)ns > 8.0 * (double)part_cap && c->knobs.get("CDBG_SCAN_MODE") == nullptr) { fits = false; var = !multi; }
            if (fits) {
                if (c->xrecs.cap > c->records.cap) c->records.swap(c->xrecs);   // (sharded reads: the previous step left the region buffer there)
                // deferred placement (above): one GPU's own partitions, the whole text at hand
                if (!multi) {
                    sp.tile_stride = 1;
                    const uint64_t nseg = launch_scan_mode<W, SCAN_EMIT_CAPPED>(c, sp, tiles, true);   // (dry run: the workgroups of the emit launch)
                    dp = defer_plan(c, W, mean, NPS, RW, nseg, region_budget() - (double)part_cap * (double)NPS * RW * 8.0 + (double)(c->defer_recs.cap + c->defer_part.cap / 2) * 8.0);
                }
                CK(c->records.alloc((uint64_t)part_cap * NPS * RW, false));
                CK(c->spill_recs.alloc(spill_cap * RW, false)); CK(c->spill_part.alloc(spill_cap, false));
                if (defer_S > 1) {
                    if (c->defer_recs.alloc(dp.slots * RW, false) != CDBG_OK || c->defer_part.alloc(dp.slots, false) != CDBG_OK) defer_S = 1;   // (no room after all: every record placed by the scan)
                    else { CK(c->defer_count.alloc((uint64_t)(defer_S - 1) * dp.nseg, false)); HIPCK(hipMemsetAsync(c->defer_count.p, 0, (uint64_t)(defer_S - 1) * dp.nseg * sizeof(uint32_t), s)); }
                }
                HIPCK(hipMemsetAsync(c->part_count.p, 0, NPS * sizeof(uint32_t), s));
                HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
            }
        }
        if (!fits) capped = false;
        else {
            sp.tile_stride = 1; sp.records = c->records.p; sp.part_cap = part_cap; sp.part_fill = c->part_count.p;
            sp.spill_recs = c->spill_recs.p; sp.spill_part = c->spill_part.p; sp.spill_cursor = c->cursors.p + 6; sp.spill_cap = spill_cap;
            if (defer_S > 1) defer_fill_params(sp);
            CK(t.start(s));
            const uint64_t done = c->ss_on ? std::min<uint64_t>(c->ss_done, tiles) : 0;
            sp.tile_offset = (uint32_t)done;
            if (tiles > done) {
                const uint64_t g = LAUNCH_SCAN(SCAN_EMIT_CAPPED, tiles - done);
                if (defer_S > 1 && g != dp.nseg) return fail(CDBG_E_INTERNAL, "deferred placement: the scan ran with %llu workgroups, its streams have %llu segments", (unsigned long long)g, (unsigned long long)dp.nseg);
            }
            sp.tile_offset = 0;
            c->st.n_tiles_overlapped = done;
            if (defer_S > 1) { CK(defer_launch_places<W>(c, sp)); deferred_open = true; }   // (second stream, behind the scan; the count's slices wait for its events)
            else CK(exscan(c->part_count.p));                // only for the total number of records
            CK(t.stop(&c->st.ms_scan_emit));
            hm.mark("count: sample + capped scan");
            if (!deferred_open) CK(read_u64(c->part_off.p + NPS, &n_records));
            CK(read_u64(c->dstats.p, hs, 2));
            CK(read_u64(c->cursors.p + 6, &n_spill));        // (deferred placement: what the scan itself spilled so far -- looked at again when the last stream is placed)
            uint32_t derr = 0; CK(read_u32(c->derr.p, &derr));
            c->ss_on = false;                                // (the streamed part is accounted for; a re-count scans everything)
            if (derr == 6 || n_spill > spill_cap) {          // estimate was off (very skewed input): exact layout instead
                if (deferred_open) { HIPCK(hipStreamSynchronize(c->place_stream)); deferred_open = false; defer_S = 1; }   // (the placement kernels append to the same lists)
                capped = false; HIPCK(hipMemset(c->derr.p, 0, 4 * sizeof(uint32_t))); HIPCK(hipMemset(c->cursors.p + 6, 0, sizeof(uint64_t)));
            } else if (multi) {
                // what travels is the exact owner-major layout: squeeze the regions (part_off = exclusive scan of the fills)
                CK(c->xrecs.alloc(std::max<uint64_t>(n_records, 1) * RW, false));
                PackRegionParams pk{ c->records.p, c->part_count.p, c->part_off.p, NPS, part_cap, RW, c->xrecs.p };
                CDBG_LAUNCH(k_pack_regions, std::min<uint64_t>((NPS + 3) / 4, 256 * 16), 256, s, pk);
                if (n_spill) {                               // the few records that overflowed their region go behind it (round 5: a spill used to send the whole step through the exact two-pass layout)
                    HIPCK(hipMemsetAsync(c->part_cursor.p, 0, NPS * sizeof(uint64_t), s));
                    PackSpillParams ps{ c->spill_recs.p, c->spill_part.p, n_spill, c->part_off.p, c->part_cursor.p, part_cap, RW, c->xrecs.p };
                    CDBG_LAUNCH(k_pack_spills, std::min<uint64_t>((n_spill + 255) / 256, 1u << 16), 256, s, ps);
                }
                c->records.swap(c->xrecs);
                capped = false; packed_exact = true;
            } else if (n_spill && !deferred_open) {
                CK(repair_spills(nullptr));
            }
        }
    }
    if (var && !capped && !packed_exact) {
        // Skewed input, ONE pass: the capped layout (uniform regions, 32-bit fill counters: two memory requests per record) plus an overflow
        // region for every partition the sample finds heavy (k_count.h, k_ovf_*).  Round 5: sized from the FIRST sample (1 tile in 64, still
        // in part_count) -- a partition heavy enough to matter has dozens of sampled records (+ 4 sigma is in its capacity), the light ones keep
        // the uniform capacity, and what is misjudged spills and is repaired.  Round 4 scanned a quarter of the tiles again for it (17 ms at
        // the hostile config-3 line) and gave EVERY partition a region of its own, whose bounds every record looked up (scan 86 ms for 66).
        // (CDBG_VAR_RESAMPLE = stride: a second sample of that density, for A/B; CDBG_SCAN_MODE=var without a first sample: taken here)
        const float ms_sample1 = c->st.ms_scan_hist;
        CK(t.start(s));
        uint64_t stride = sample_ns ? sample_stride : std::min<uint64_t>(64, std::max<uint64_t>(1, tiles / 4096));
        bool resample = sample_ns == 0;
        if (const char* e = c->knobs.get("CDBG_VAR_RESAMPLE")) { stride = std::max(1, atoi(e)); resample = true; }
        const uint64_t ns = (tiles + stride - 1) / stride;
        uint64_t sample_records = sample_recs;
        if (resample) {
            HIPCK(hipMemsetAsync(c->part_count.p, 0, NPS * sizeof(uint32_t), s));
            HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
            sp.tile_stride = (uint32_t)stride; sp.tile_offset = 0; sp.part_cap = 0; sp.var_limit = nullptr; sp.ovf = nullptr;
            LAUNCH_SCAN(SCAN_HIST, ns);
            CK(exscan(c->part_count.p)); CK(read_u64(c->part_off.p + NPS, &sample_records));
        }
        CK(c->var_cap.alloc(NPS, false)); CK(c->var_pairs.alloc(2 * NPS, false)); CK(c->ovf_words.alloc(NPS, false));
        const double scale = (double)tiles / (double)ns, mean = (double)sample_records * scale / (double)NPS;
        capped_capacities(c, mean, NPS, part_cap, spill_cap);
        // uniform capacity of a skewed input: at least 3 x the mean, so that a partition the sample does NOT flag (fewer than twice the mean sample + 2
        // sampled records) fits with near certainty; the flagged ones get their overflow region from their own count + 4 sigma
        if (c->knobs.get("CDBG_PART_CAP") == nullptr) part_cap = std::max<uint32_t>(part_cap, ((uint32_t)(3.0 * mean) + 7u) & ~7u);
        const float heavy_min = c->knobs.get("CDBG_PART_CAP") ? 0.0f : (float)(2.0 * mean / scale + 2.0);
        OvfParams& op = var_op;
        op = OvfParams{ c->part_count.p, c->var_cap.p, NPS, (float)scale, heavy_min, part_cap, c->part_off.p, NPS * (uint64_t)part_cap, c->ovf_words.p,
                        c->part_count.p, nullptr, RW, c->var_pairs.p, c->dstats.p + 28, 0, NPS };
        if (const char* e = c->knobs.get("CDBG_VAR_SCALE")) op.scale = (float)atof(e);   // test knob (with CDBG_PART_CAP): overflow regions far too small, so that partitions spill
        CDBG_LAUNCH(k_ovf_caps, (NPS + 255) / 256, 256, s, op);
        CK(exscan_u32(c, c->var_cap.p, c->part_off.p, NPS));
        uint64_t ovf_total = 0; CK(read_u64(c->part_off.p + NPS, &ovf_total));
        const uint64_t total_cap = NPS * (uint64_t)part_cap + ovf_total;
        float ms2 = 0; CK(t.stop(&ms2)); c->st.ms_scan_hist = ms_sample1 + ms2;
        // does it fit?  What the card has free, plus what this context (the region of the step before) and the process's pool would hand
        // back, less a reserve for the stages that follow; an allocation that fails all the same falls back to the exact layout as well
        if ((double)total_cap * RW * 8.0 > region_budget()) var = false;           // would not fit: the exact layout
        else {
            // deferred placement (above) on this layout as well: k_place looks a heavy partition's overflow word up like the scan does
            sp.tile_stride = 1; sp.tile_offset = 0;
            const uint64_t nseg = launch_scan_mode<W, SCAN_EMIT_CAPPED>(c, sp, tiles, true);
            dp = defer_plan(c, W, mean, NPS, RW, nseg, region_budget() - (double)total_cap * RW * 8.0 + (double)(c->defer_recs.cap + c->defer_part.cap / 2) * 8.0);
            if (c->records.alloc(total_cap * RW, false) != CDBG_OK) var = false;
        }
        if (var) {
            c->ss_on = false;                                // (as on the capped path: a re-count scans everything)
            spill_cap = std::max<uint64_t>(total_cap / 32, 65536);
            CK(c->spill_recs.alloc(spill_cap * RW, false)); CK(c->spill_part.alloc(spill_cap, false));
            if (defer_S > 1) {
                if (c->defer_recs.alloc(dp.slots * RW, false) != CDBG_OK || c->defer_part.alloc(dp.slots, false) != CDBG_OK) defer_S = 1;
                else { CK(c->defer_count.alloc((uint64_t)(defer_S - 1) * dp.nseg, false)); HIPCK(hipMemsetAsync(c->defer_count.p, 0, (uint64_t)(defer_S - 1) * dp.nseg * sizeof(uint32_t), s)); }
            }
            CK(t.start(s));
            CDBG_LAUNCH(k_ovf_words, (NPS + 255) / 256, 256, s, op);
            HIPCK(hipMemsetAsync(c->part_count.p, 0, NPS * sizeof(uint32_t), s));      // the sample is spent: fill counters
            HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
            HIPCK(hipMemsetAsync(c->cursors.p + 6, 0, sizeof(uint64_t), s));
            sp.tile_stride = 1; sp.tile_offset = 0; sp.records = c->records.p; sp.part_cap = part_cap; sp.part_fill = c->part_count.p; sp.ovf = c->ovf_words.p;
            sp.spill_recs = c->spill_recs.p; sp.spill_part = c->spill_part.p; sp.spill_cursor = c->cursors.p + 6; sp.spill_cap = spill_cap;
            if (defer_S > 1) defer_fill_params(sp);
            const uint64_t g = LAUNCH_SCAN(SCAN_EMIT_CAPPED, tiles);
            if (defer_S > 1 && g != dp.nseg) return fail(CDBG_E_INTERNAL, "deferred placement: the scan ran with %llu workgroups, its streams have %llu segments", (unsigned long long)g, (unsigned long long)dp.nseg);
            op.records = c->records.p;
            if (defer_S > 1) { CK(defer_launch_places<W>(c, sp)); deferred_open = true; }   // (k_ovf_finish then runs slice by slice, in front of every slice's count)
            else CDBG_LAUNCH(k_ovf_finish, std::min<uint64_t>((NPS + 3) / 4, 256 * 16), 256, s, op);
            sp.ovf = nullptr;
            CK(t.stop(&c->st.ms_scan_emit));
            hm.mark("count: sample + single-pass scan into capped regions with overflow regions");
            if (!deferred_open) CK(read_u64(c->dstats.p + 28, &n_records));
            CK(read_u64(c->dstats.p, hs, 2));
            CK(read_u64(c->cursors.p + 6, &n_spill));
            uint32_t derr = 0; CK(read_u32(c->derr.p, &derr));
            if (derr == 6 || n_spill > spill_cap) {          // the estimate was off by more than the spill list holds: exact layout
                if (deferred_open) { HIPCK(hipStreamSynchronize(c->place_stream)); deferred_open = false; defer_S = 1; }
                var = false; HIPCK(hipMemset(c->derr.p, 0, 4 * sizeof(uint32_t))); HIPCK(hipMemset(c->cursors.p + 6, 0, sizeof(uint64_t)));
            } else if (n_spill && !deferred_open) {
                CK(repair_spills(c->ovf_words.p));
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
        // the histogram is spent (part_off holds its scan): its words become the running indices, and end at the same counts.  Fewer than
        // 2^32 records: pre-loaded with the offsets, so that a record costs one atomic and one store (k_scan.h)
        const bool cur32 = n_records < (1ull << 32) && c->knobs.get("CDBG_EXACT_NO_CUR32") == nullptr;
        if (cur32) CDBG_LAUNCH(k_cursor32_load, (NPS + 255) / 256, 256, s, (const uint64_t*)c->part_off.p, c->part_count.p, NPS);
        else HIPCK(hipMemsetAsync(c->part_count.p, 0, NPS * sizeof(uint32_t), s));
        sp.records = c->records.p; sp.part_fill = c->part_count.p; sp.part_off = cur32 ? nullptr : c->part_off.p; sp.var_limit = nullptr;
        LAUNCH_SCAN(SCAN_EMIT, tiles);
        if (cur32) CDBG_LAUNCH(k_cursor32_counts, (NPS + 255) / 256, 256, s, (const uint64_t*)c->part_off.p, c->part_count.p, NPS);
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
    c->st.n_records = n_records; c->st.n_member_kmers = hs[0];   // (deferred placement: the record count follows when the last stream is placed)
    c->st.count_slices = (int)defer_S;
    hm.mark("count: spill repair/exchange");

    // Solid entries: at most one per abundance-min member k-mers -- a bound that is 12 x the need at sequencing depth (config 3: 6.7 G
    // entries, 81 GB, for 0.65 G), and fresh device memory costs 40 - 70 ms per GB to obtain (profiles/r05_micro_alloc.log: the CLI's
    // first and only job paid seconds for it).  First attempt: a third of the bound; the kernels report an overflow (device error 1)
    // without writing out of bounds, and the stage then runs once more with the bound itself (an input of mostly distinct k-mers).
    const uint64_t solid_slack = count_solid_slack(NPL, defer_S);
    const uint64_t solid_bound = hs[0] / (uint64_t)std::max(1, c->prm.abundance_min) + solid_slack;
    uint64_t solid_cap = std::min(solid_bound, count_solid_first_cap(hs[0], c->prm.abundance_min, NPL, defer_S));
    if (c->knobs.get("CDBG_SOLID_FIRST_TINY")) solid_cap = 1u << 15;   // (tests: force the second attempt; not below one partition's entries -- the kernels park an overflowing partition at offset 0)
    if (c->solid_keys.cap >= solid_bound * W && c->solid_cnt.cap >= solid_bound) solid_cap = solid_bound;   // (a context that was re-run keeps what it has)
    float ms_count_first = 0;
    for (int solid_attempt = 0;; ++solid_attempt) {
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
    cp.n_items = (uint32_t)NPL; cp.max_passes = 16;                // (round 5: was 64 -- beyond 16 LDS passes the HBM-table pass is cheaper: hostile k = 127 share 4.2 -> 2.0 s, k = 55 1.03 -> 0.85 s)
    if (const char* e = c->knobs.get("CDBG_MAX_PASSES")) cp.max_passes = (uint32_t)std::max(1, atoi(e));   // dev knob: LDS passes before a partition goes to the HBM-table pass
    if (const char* e = c->knobs.get("CDBG_COUNT_MAX_SUB")) { uint32_t v = (uint32_t)std::max(1, atoi(e)); while (v & (v - 1)) v &= v - 1; cp.max_sub = std::min(v, 16u); }   // dev knob: 1, 2, 4, 8, 16
    // one-pass kernel over all partitions; the ones whose distinct k-mers do not fit the LDS table at once come back on
    // the retry list and go through the multi-pass kernel
    {
        // (admission by predicted fill: k_count_fast.h; one-word k-mers: off -- their second tier runs one workgroup per CU against three)
        CountFastParams fp{ cp, c->retry_list.p, c->big_count.p + 1, count_fast_record_limit<W>(c->k), W == 1 ? 0u : W == 2 ? 177u : 200u };
        if (const char* e = c->knobs.get("CDBG_FAST_SKIP_Q8")) fp.skip_fill_q8 = (uint32_t)std::max(0, atoi(e));   // dev knob
        if (const char* e = c->knobs.get("CDBG_FAST_MAX_RECORDS")) fp.fast_max_records = std::min<uint32_t>(count_fast_record_limit<W>(c->k), (uint32_t)std::max(1, atoi(e)));   // dev knob
        if (defer_S > 1) {
            // deferred placement: the tier runs once per slice of the partition space, slice q as soon as its stream has been placed (stream q is
            // placed on the second HIP stream while slice q - 1 is counted here; slice 0 was placed by the scan)
            for (uint32_t q = 0, x = 0; q < defer_S; x += dp.w[q], ++q) {
                if (q) HIPCK(hipStreamWaitEvent(s, c->place_ev[q], 0));
                const uint64_t per = NPL / 16 * dp.w[q];
                fp.c.item_base = (uint32_t)(NPL / 16 * x); fp.c.n_items = (uint32_t)per;
                if (var) {                                   // (skewed inputs: the slice's overflowed partitions become one run each, begin / end of every partition written)
                    var_op.p0 = NPL / 16 * x; var_op.p1 = var_op.p0 + per;
                    CDBG_LAUNCH(k_ovf_finish, std::min<uint64_t>((per + 3) / 4, 256 * 16), 256, s, var_op);
                    CDBG_LAUNCH((k_count_fast<W, TS, Cfg<W>::NTC, false>), std::min<uint64_t>(per, COUNT_GRID), Cfg<W>::NTC, s, fp);
                } else
                CDBG_LAUNCH((k_count_fast<W, TS, Cfg<W>::NTC, true>), std::min<uint64_t>(per, COUNT_GRID), Cfg<W>::NTC, s, fp);
            }
        }
        else if (capped) CDBG_LAUNCH((k_count_fast<W, TS, Cfg<W>::NTC, true>), std::min<uint64_t>(NPL, COUNT_GRID), Cfg<W>::NTC, s, fp);
        else CDBG_LAUNCH((k_count_fast<W, TS, Cfg<W>::NTC, false>), std::min<uint64_t>(NPL, COUNT_GRID), Cfg<W>::NTC, s, fp);
    }
    c->st.n_launch_count = NPL;
    uint32_t nretry = 0;
    HIPCK(hipStreamSynchronize(s));
    hm.mark("count: tier 1");
    if (deferred_open) {
        // every stream has been placed (the last slice's launch waited for it): the record count and the spill list are final now
        deferred_open = false;
        HIPCK(hipStreamSynchronize(c->place_stream));
        if (var) CK(read_u64(c->dstats.p + 28, &n_records));   // (k_ovf_finish summed the fills of its slices; the count kernels leave that word alone)
        else { CK(exscan(c->part_count.p)); CK(read_u64(c->part_off.p + NPS, &n_records)); }
        c->st.n_records = n_records;
        { std::vector<uint32_t> dc((size_t)(defer_S - 1) * dp.nseg); CK(read_u32(c->defer_count.p, dc.data(), dc.size())); uint64_t nd = 0;
          for (uint32_t v : dc) nd += std::min(v, dp.cap);
          c->st.n_deferred_records = nd; }
        HIPCK(hipEventElapsedTime(&c->st.ms_place, c->place_ev[0], c->place_ev[defer_S - 1]));
        CK(read_u64(c->cursors.p + 6, &n_spill));
        if (n_spill > spill_cap) {
            // the capacity estimate was off by more than the spill list holds (the scan alone had not shown it): this step once more, every record placed by
            // the scan, which then falls back to the exact layout by itself
            HIPCK(hipMemset(c->derr.p, 0, 4 * sizeof(uint32_t))); HIPCK(hipMemset(c->cursors.p + 6, 0, sizeof(uint64_t)));
            c->defer_off_once = true;
            return count_impl<W>(c);
        }
        if (n_spill) CK(repair_spills(var ? c->ovf_words.p : nullptr));   // (the tier above left the spilled partitions alone; their gathered copies are counted below)
    }
    CK(read_u32(c->big_count.p + 1, &nretry));
    const uint32_t* retry_ptr = c->retry_list.p;
    // second tier, k-mers of three words and more under an abundance filter: the sifting tier (k_count_fast.h) -- fingerprints first,
    // exact counts only for what was seen again; two workgroups per CU instead of one, and partitions of up to 6000 distinct k-mers
    // fit: config-5 share 67 (4096-slot tier) + 52 (multi-pass) -> 82 + 17 ms.  Two-word k-mers keep the 4096-slot tier (k = 55: more
    // than half of the occurrences are of k-mers seen again, and reading them twice costs more than the small table saves: 141 -> 149 ms)
    const bool sift = W >= 3 && c->prm.abundance_min >= 2 && c->knobs.get("CDBG_NO_SIFT") == nullptr;
    if constexpr (W >= 3) if (nretry && sift && c->knobs.get("CDBG_NO_COUNT_TIER2") == nullptr) {
        CK(c->retry_list2.alloc(nretry, false));
        HIPCK(hipMemsetAsync(c->big_count.p + 2, 0, sizeof(uint32_t), s));
        CountParams c2 = cp; c2.part_list = c->retry_list.p; c2.n_items = nretry;
        CountFastParams fp2{ c2, c->retry_list2.p, c->big_count.p + 2, count_fast_record_limit<W>(c->k), 192u };   // (admission: predicted distinct k-mers beyond 3/4 of the fingerprint words -> multi-pass kernel untried)
        if (const char* e = c->knobs.get("CDBG_FAST_SKIP2_Q8")) fp2.skip_fill_q8 = (uint32_t)std::max(0, atoi(e));   // dev knob
        constexpr int TSS = 512, FSS = 8192, NTS = 512;       // (exact table: a quarter of tier 1's slots)
        const uint64_t grid = std::min<uint64_t>(nretry, SIFT_GRID);    // (two workgroups per CU are resident; the rest take turns: COUNT_GRID, host_ctx.h)
        if (capped) CDBG_LAUNCH((k_count_fast<W, TSS, NTS, 3, FSS>), grid, NTS, s, fp2);
        else CDBG_LAUNCH((k_count_fast<W, TSS, NTS, 2, FSS>), grid, NTS, s, fp2);
        HIPCK(hipStreamSynchronize(s));
        CK(read_u32(c->big_count.p + 2, &nretry));
        retry_ptr = c->retry_list2.p;
    }
    if constexpr (W <= 4) if (nretry && !sift && c->knobs.get("CDBG_NO_COUNT_TIER2") == nullptr) {   // (wider keys: a table twice the size does not fit the LDS)
        // second tier: the same one-pass kernel with a table twice the size (one workgroup per CU) over the retry list; at the
        // config-5 share 6 % of the partitions -- a minimizer locus of long reads -- cost 250 of 590 ms in the multi-pass kernel
        CK(c->retry_list2.alloc(nretry, false));
        HIPCK(hipMemsetAsync(c->big_count.p + 2, 0, sizeof(uint32_t), s));
        CountParams c2 = cp; c2.part_list = c->retry_list.p; c2.n_items = nretry;
        // (three- and four-word k-mers: the admission rule here as well -- a partition predicted beyond 0.68 of the 4096 slots goes to
        //  the multi-pass kernel untried: count 216 -> 201 ms at the config-5 share; two-word k-mers: no difference, off)
        CountFastParams fp2{ c2, c->retry_list2.p, c->big_count.p + 2, count_fast_record_limit<W>(c->k), W >= 3 ? 175u : 0u };
        if (const char* e = c->knobs.get("CDBG_FAST_SKIP2_Q8")) fp2.skip_fill_q8 = (uint32_t)std::max(0, atoi(e));   // dev knob
        // (multi-word k-mers: 1024 threads -- the table fills the CU's LDS either way, so the workgroup size IS the occupancy: 16
        //  waves per CU instead of 8, second tier 81 -> 67 ms at the config-5 share, 24 -> 18 at the config-4 share)
#ifndef CDBG_NT_TIER2
#define CDBG_NT_TIER2 1024
#endif
        constexpr int NT2 = W == 1 ? Cfg<W>::NTC : CDBG_NT_TIER2;
        if (capped) CDBG_LAUNCH((k_count_fast<W, 2 * TS, NT2, 3>), std::min<uint64_t>(nretry, T2_GRID), NT2, s, fp2);
        else CDBG_LAUNCH((k_count_fast<W, 2 * TS, NT2, 2>), std::min<uint64_t>(nretry, T2_GRID), NT2, s, fp2);
        HIPCK(hipStreamSynchronize(s));
        CK(read_u32(c->big_count.p + 2, &nretry));
        retry_ptr = c->retry_list2.p;
    }
    c->st.n_multipass_partitions = nretry;
    if (nretry) {
        CountParams rp1 = cp;
        rp1.part_list = retry_ptr; rp1.n_items = nretry;
        // (multi-word k-mers: the table of the second tier and 1024 threads -- half the passes at 16 waves per CU: 45 -> 39 ms at the config-5 share)
        constexpr int TSG = (W == 1 && !CDBG_MP1_WIDE) ? TS : 2 * TS, NTG = (W == 1 && !CDBG_MP1_WIDE) ? Cfg<W>::NTC : 1024;
        CDBG_LAUNCH((k_count<W, TSG, NTG, false>), std::min<uint64_t>(nretry, W == 1 ? MP_GRID_1 : MP_GRID_W), NTG, s, rp1);
    }
    if (n_spilled_parts) {                                   // spilled partitions: count their gathered copies
        CountParams rp2 = cp;
        rp2.records = c->repair_recs.p; rp2.item_off = c->repair_off.p; rp2.part_list = c->repair_part.p; rp2.part_stride = 0;
        rp2.n_items = (uint32_t)n_spilled_parts; rp2.max_passes = 4096;
        if (const char* ev = c->knobs.get("CDBG_REPAIR_MAX_PASSES")) rp2.max_passes = (uint32_t)std::max(1, atoi(ev));   // (tests: a spilled partition that is deferred as well)
        constexpr int TSG = (W == 1 && !CDBG_MP1_WIDE) ? TS : 2 * TS, NTG = (W == 1 && !CDBG_MP1_WIDE) ? Cfg<W>::NTC : 1024;
        CDBG_LAUNCH((k_count<W, TSG, NTG, false>), std::min<uint64_t>(rp2.n_items, W == 1 ? MP_GRID_1 : MP_GRID_W), NTG, s, rp2);
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
        // table of a listed partition: twice its member k-mers (an upper bound of its distinct ones), counted on the device (k_big_members).
        // Round 4 took records x the largest record of the format: 3 - 5 x too many slots, which one workgroup clears and sweeps
        HIPCK(hipMemcpy(c->big_list.p, bl.data(), nbig * sizeof(uint32_t), hipMemcpyHostToDevice));
        DBuf<uint64_t> d_members; CK(d_members.alloc(nbig, true));
        DBuf<uint32_t> d_nrec; CK(d_nrec.alloc(nbig, false));
        { CountParams mp = cp; mp.part_list = c->big_list.p; mp.n_items = nbig;
          BigMembersParams bm{ mp, RW, d_members.p, d_nrec.p };
          CDBG_LAUNCH(k_big_members, nbig, 256, s, bm); }
        std::vector<uint64_t> h_members(nbig); CK(read_u64(d_members.p, h_members.data(), nbig));
        for (uint32_t i = 0; i < nbig; ++i) offs[i + 1] = offs[i] + pow2_at_least(2 * h_members[i] + 4 * 256);
        CK(g_keys.alloc(offs[nbig] * W, false)); CK(g_cnt.alloc(offs[nbig], false));
        CK(big_off.alloc(nbig + 1, false));
        HIPCK(hipMemcpy(big_off.p, offs.data(), (nbig + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
        CountParams bp = cp;
        bp.part_list = c->big_list.p; bp.g_keys = g_keys.p; bp.g_cnt = g_cnt.p; bp.big_off = big_off.p;
        bp.n_items = nbig; bp.max_passes = 1;
        if (c->knobs.get("CDBG_BIG_ONE_WG") != nullptr) {
            // (dev knob: round 5's pass -- one 256-thread workgroup per listed partition; the grid bounded: every workgroup reserves whole output chunks, the slack is sized for PERSISTENT_GRID)
            CDBG_LAUNCH((k_count<W, TS, 256, true>), std::min<uint64_t>(nbig, PERSISTENT_GRID), 256, s, bp);
        } else {
            // grid-wide (k_count.h): all workgroups share the listed partitions' records, clear and sweep the tables slot by slot; the segments of the
            // solid arrays are reserved between the two sweeps -- every other launch of the stage has finished, the cursor is the host's to advance
            DBuf<uint64_t> rec_pref, seg_pref; DBuf<uint32_t> nsolid, wr;
            CK(rec_pref.alloc(nbig + 1, false)); CK(seg_pref.alloc(nbig + 1, false)); CK(nsolid.alloc(nbig, true)); CK(wr.alloc(nbig, true));
            CK(exscan_u32(c, d_nrec.p, rec_pref.p, nbig));
            BigGridParams bg{ bp, RW, rec_pref.p, nsolid.p, seg_pref.p, 0, wr.p };
            const uint64_t grid = resident_grid(k_big_insert<W>, 256, 256 * 4);
            CDBG_LAUNCH(k_big_clear<W>, std::min<uint64_t>((offs[nbig] + 255) / 256, 256 * 16), 256, s, bg);
            CDBG_LAUNCH(k_big_insert<W>, grid, 256, s, bg);
            CDBG_LAUNCH((k_big_sweep<W, 0>), std::min<uint64_t>((offs[nbig] + 255) / 256, 256 * 16), 256, s, bg);
            CK(exscan_u32(c, nsolid.p, seg_pref.p, nbig));
            uint64_t n_out = 0, cur = 0; CK(read_u64(seg_pref.p + nbig, &n_out)); CK(read_u64(c->solid_cursor.p, &cur));
            if (cur + n_out > solid_cap) { const uint32_t one = 1; HIPCK(hipMemcpy(c->derr.p, &one, sizeof one, hipMemcpyHostToDevice)); n_out = 0; }   // (as the kernels report it: the attempt with the bound follows)
            else {
                const uint64_t upd = cur + n_out; HIPCK(hipMemcpy(c->solid_cursor.p, &upd, sizeof upd, hipMemcpyHostToDevice));
                bg.seg_base = cur;
                CDBG_LAUNCH(k_big_segments, (nbig + 255) / 256, 256, s, bg);
                CDBG_LAUNCH((k_big_sweep<W, 1>), std::min<uint64_t>((offs[nbig] + 255) / 256, 256 * 16), 256, s, bg);
            }
            HIPCK(hipStreamSynchronize(s));                  // (the prefix arrays are this block's)
        }
        c->st.n_big_partitions += nbig;
    }
    CK(t.stop(&c->st.ms_count)); c->st.ms_count += ms_count_first;
    hm.mark("count: HBM-table partitions");
    { uint32_t de = 0; CK(read_u32(c->derr.p, &de));
      if (de == 1 && solid_cap < solid_bound && solid_attempt == 0) {                       // the first attempt's estimate was too small: once more with the bound
          ms_count_first = c->st.ms_count; solid_cap = solid_bound; c->st.n_big_partitions = 0; c->st.n_multipass_partitions = 0;
          HIPCK(hipMemset(c->derr.p, 0, 4 * sizeof(uint32_t)));
          continue;
      } }
    break;
    }
    CK(check_device_error(c, "count"));
    uint64_t cs[4]; CK(read_u64(c->dstats.p, cs, 4));
#ifdef CDBG_PROFILE_PHASES
    { uint64_t ph[16]; CK(read_u64(c->dstats.p + 8, ph, 16));
      for (int w = 0; w < 2; ++w) fprintf(stderr, "k_count_fast phase cycles, %s wave, summed over WGs: loop-end->top %llu | wait own records + stage %llu | insert %llu | barrier A %llu | sweep %llu | barrier B %llu\n", w ? "last" : "first",
          (unsigned long long)ph[8 * w + 0], (unsigned long long)ph[8 * w + 1], (unsigned long long)ph[8 * w + 2], (unsigned long long)ph[8 * w + 3], (unsigned long long)ph[8 * w + 4], (unsigned long long)ph[8 * w + 5]); }
#endif
    c->st.n_distinct = cs[0]; c->st.n_occurrences = cs[1]; c->st.n_solid = cs[2]; c->st.n_solid_travellers = cs[3];
    CK(read_u64(c->solid_cursor.p, &c->n_solid_entries));
    if (c->knobs.get("CDBG_DEBUG_SEGHIST")) {                      // dev aid: solid entries per bucket, log2 bins (stderr)
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

}  // namespace
