// k_scan.h -- stage 1a: reads -> minimizer-partitioned super-k-mer records.
//
// Replaces (new design, not a translation) the read scan / super-k-mer splitter of
// gatb-core's DSK "fill partitions" step that GraphUnitigsTemplate<span>::create
// drives (/root/reference/src/bcalm_1.cpp:57), and
// folds BCALM 2's "doubled k-mer" routing of bcalm_algo (row a7) into the same pass.
//
// MI355X-first formulation: the whole read set is ONE byte stream in HBM in which
// any non-ACGT byte ('\n' between reads, 'N', ...) breaks the sequence, so the scan
// is position-parallel: one lane per junction ((k-1)-mer start), 4096 junctions per
// workgroup tile staged once through LDS as 2-bit codes.
//
//   key(m-mer)  = mix32(canonical m-mer)                    (hash-ordered minimizers)
//   g(j)        = min key over the k-m m-mers of junction j  (LDS doubling window-min)
//   run         = maximal stretch of valid junctions with equal g
//   record(run) = the k-mers touching the run's junctions: j_lo-1 .. j_hi.  Interior
//                 k-mers have both junctions in the run (home here).  The first/last
//                 k-mer has its other junction elsewhere: it is HOME in the partition
//                 of the smaller g and a TRAVELLER (flagged copy) in the other, so each
//                 bucket later sees every k-mer adjacent to each of its junctions
//                 without any exchange step.
//   partition   = part_of(g); each k-mer is counted exactly once per partition.
//
// Record layout (RW = 2W uint64 words, r[0] least significant):
//   bits [0,8)  n = number of member k-mers       bit 8  member 0 is a traveller
//   bit 9  member n-1 is a traveller               bits [16, 64*RW): bases, 2 bit each,
//   base i at bits [64*RW-2(i+1), 64*RW-2i)  (first base on top), n+k-1 bases.
//   bit 10 the junction member 0 shares with the run BEFORE belongs to another bucket ("foreign"; implied by bit 8)
//   bit 11 the junction member n-1 shares with the run AFTER is foreign (implied by bit 9)
//   Every other junction of a record lies in the run, i.e. is owned by the record's bucket.  The count stage carries the
//   two bits with the k-mer (k_count.h, KEY_FOREIGN_*) so that the compaction does not have to recompute the minimizers
//   of both junctions of every solid k-mer (k - m + 1 hashed windows each: a quarter of k_compact_wave<2> at k = 55).
#pragma once
#include "kmer.h"

namespace cdbg {

#if defined(CDBG_PROFILE_PHASES) && !defined(CDBG_HOSTSIM)
#define CDBG_SPH(i) do { if (threadIdx.x == 0) { const uint64_t t_ = wall_clock64(); sph[i] += t_ - st_prev; st_prev = t_; } } while (0)
#else
#define CDBG_SPH(i) do { } while (0)
#endif
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_GRID = 256 * 2;                     // fallback grid of the persistent launch
constexpr int SCAN_TILE = 4096;                       // junctions per workgroup
constexpr int SCAN_PKW = (SCAN_TILE + 16 + 256) / 16 + 11;   // packed words (16 bases each) incl. halo/over-read; EVEN
static_assert(SCAN_PKW % 2 == 0, "validity halves must fill whole 32-bit words");
constexpr int SCAN_NKEY = SCAN_TILE + 256;                  // keys of a tile: k - m <= 254 (k <= 255)

template <int W> struct RecFmt {
    static constexpr int RW = 2 * W;
    static constexpr int CAPB = 32 * RW - 8;          // bases that fit beside the 16 meta bits
};

struct ScanParams {
    const uint8_t* reads;        // padded to a multiple of 16 with separator bytes
    uint64_t nbytes;             // valid bytes
    uint64_t nbytes_padded;      // allocated/initialised bytes (multiple of 16)
    int k, m, log_np;            // global partition count = 1 << log_np
    int rank_bits, rank;         // this GPU owns partitions p with (p & ((1<<rank_bits)-1)) == rank
    uint32_t* part_count;        // HIST pass: records per local partition
    uint64_t* part_cursor;       // EMIT pass: running cursors, pre-loaded with exclusive offsets
    uint64_t* records;           // EMIT pass: RW words per record
    uint64_t* stats;             // [0] member k-mers emitted (incl. travellers), [1] traveller members
    uint64_t n_tiles;            // tiles of this launch (k_scan_fast: persistent workgroups stride over them)
    uint32_t tile_stride, tile_offset;   // tile = blockIdx.x * tile_stride + tile_offset (sampling for the capacity estimate)
    // SCAN_EMIT_CAPPED (single pass, no histogram): partition p owns records [p*part_cap, (p+1)*part_cap);
    // part_fill[p] counts the records offered; records beyond the capacity go to the spill list
    uint32_t part_cap; uint32_t* part_fill;
    uint64_t* spill_recs; uint32_t* spill_part; uint64_t* spill_cursor; uint64_t spill_cap; uint32_t* error;
    uint32_t emit_all, npl;      // multi-GPU with sharded reads: emit every partition, slot = owner * npl + local partition
    const uint64_t* var_limit;   // SCAN_EMIT with estimated regions: end of partition p's region (records beyond it go to the spill list); nullptr = exact offsets
    const uint64_t* ovf;         // SCAN_EMIT_CAPPED, skewed inputs: per partition 0 or (first record of its overflow region << OVF_CAP_BITS) | records the partition
                                 // may hold in all (record j >= part_cap goes to slot j of that region; its first part_cap slots receive the uniform region's records afterwards)
    const uint64_t* part_off;    // SCAN_EMIT: first record of partition p's region, added to the 32-BIT running index part_fill[p] (zeroed by the host);
                                 // nullptr: part_fill[p] was pre-loaded with the offset itself (exact layout of fewer than 2^32 records: no second access)
    // SCAN_EMIT_CAPPED, DEFERRED PLACEMENT (round 6; host_count.h): the partition space is cut into defer_slices ranges of whole SIXTEENTHS (slice of partition
    // p = nibble p >> defer_shift of defer_map: slices need not be equal -- the last ones are small, so that little counting is left when the last stream has
    // been placed).  The scan places the records of slice 0 itself; a record of slice q > 0 is APPENDED to stream q -- coalesced 16-byte stores -- and k_place
    // scatters stream q + 1 on a second HIP stream while the count stage counts slice q: the placement is bound by memory requests (two per record), the count
    // by VALU issue.  A stream is one SEGMENT per workgroup of the scan (the persistent workgroups take equal shares of the tiles), so that appending needs no
    // device atomic at all: the workgroup's cursors live in LDS and are written to defer_count[(q - 1) * defer_nseg + workgroup] when it is done (one
    // cursor word for the whole chip admits 88 M atomics/s: 25 M wave-level reservations took the scan from 66 to 318 ms).  Segment g of stream q holds
    // records [((q - 1) * defer_nseg + g) * defer_seg_cap, + defer_seg_cap) of defer_recs / defer_part (one capacity for all streams: that of the largest
    // slice -- nothing but LDS cursors and registers in the scan's emit loop); a record that finds its segment full is placed directly (always correct).
    // defer_slices <= 1: off.
    uint32_t defer_slices, defer_shift, defer_nseg, defer_seg_cap;
    uint64_t defer_map;
    uint64_t* defer_recs; uint32_t* defer_part; uint32_t* defer_count;
};
// Round 5: placing a record is ONE returning 32-bit atomic and one store wherever the layout allows it.  The exact layout used to
// advance a 64-bit cursor per partition: the same 1.6 G atomics on the same addresses took its emit pass from 66.8 to 89.1 ms at
// config 3 (profiles/r05_ab_scan_atomic_width.log), and a 32-bit index plus a load of the region's offset costs the same 86 - 89 ms:
// the pass pays per memory REQUEST of a record (capped layout: two), whatever its kind.
constexpr int SCAN_DEFER_MAX = 16;                       // slices of the partition space at most
constexpr int SCAN_HIST = 0, SCAN_EMIT = 1, SCAN_EMIT_CAPPED = 2;
constexpr int OVF_CAP_BITS = 28;                         // (a partition of more than 2^28 records spills its excess, and the step then falls back to the exact layout)
constexpr uint64_t OVF_CAP_MASK = (1ULL << OVF_CAP_BITS) - 1ULL;

CDBG_DEV uint64_t scan_get64(const uint32_t* pk, int bitoff) {
    const int w = bitoff >> 5, sh = bitoff & 31;
    uint64_t x = ((uint64_t)pk[w] << 32) | pk[w + 1];
    return sh ? ((x << sh) | ((uint64_t)pk[w + 2] >> (32 - sh))) : x;
}
// all bases [q, q+len) valid?  vm: one bit per base, LSB-first in 32-bit words
CDBG_DEV bool scan_all_valid(const uint32_t* vm, int q, int len) {
    while (len > 0) {
        const int w = q >> 5, sh = q & 31;
        uint64_t x = ((uint64_t)vm[w + 1] << 32) | vm[w];
        uint32_t bits = (uint32_t)(x >> sh);
        const int c = len < 32 ? len : 32;
        const uint32_t msk = c == 32 ? 0xFFFFFFFFu : ((1u << c) - 1u);
        if ((bits & msk) != msk) return false;
        q += c; len -= c;
    }
    return true;
}

struct alignas(16) RecPair { uint64_t lo, hi; };      // two words of a record, stored at once (global_store_dwordx4)
// place one record: count it (HIST), write it at its exact offset (EMIT) or into the partition's
// fixed-capacity region / the spill list (EMIT_CAPPED) -- one device atomic either way.
// bitoff = bit offset of the record's first base in the tile's packed 2-bit stream.
#ifndef CDBG_REC16_W1
#define CDBG_REC16_W1 0                                  // (A/B: one 16-byte store for the record of a one-word k-mer as well)
#endif
// the placement atomic of a record (its raw return value: scan_finish_record turns it into the record's place)
template <int MODE>
CDBG_DEV uint32_t scan_reserve_record(const ScanParams& P, uint32_t lpart) {
    if (MODE == SCAN_HIST) { atomic_add_u32(&P.part_count[lpart], 1u); return 0u; }
    return atomic_add_u32(&P.part_fill[lpart], 1u);
}
// where record j of partition lpart goes (nullptr: nowhere -- HIST pass, or the spill list is full and the step falls back to the exact layout)
template <int W, int MODE>
CDBG_DEV uint64_t* scan_record_dst(const ScanParams& P, uint32_t lpart, uint32_t j) {
    constexpr int RW = RecFmt<W>::RW;
    if (MODE == SCAN_HIST) return nullptr;
    uint64_t* dst;
    bool fits;
    if (MODE == SCAN_EMIT) {
        // exact layout (var_limit == nullptr: the histogram pass sized every region), or ESTIMATED regions of their own size per
        // partition (var_limit[p] = end of p's region, sized from a sampled histogram: the single-pass layout of skewed inputs)
        uint64_t pos = (uint64_t)j;
        fits = true;
        if (P.part_off != nullptr) {                         // (nullptr: the counters were pre-loaded with the regions' offsets -- exact layout of fewer than 2^32 records)
            pos += P.part_off[lpart];
            fits = P.var_limit == nullptr || pos < P.var_limit[lpart];
        }
        dst = P.records + pos * RW;
    } else {
        fits = j < P.part_cap;
        dst = P.records + ((uint64_t)lpart * P.part_cap + j) * RW;
        if (!fits && P.ovf != nullptr) {                     // skewed input: a partition the sample found heavy has an overflow region of its own
            const uint64_t d = P.ovf[lpart];                 // (one more load for the records beyond the uniform capacity only)
            if ((uint64_t)j < (d & OVF_CAP_MASK)) { fits = true; dst = P.records + ((d >> OVF_CAP_BITS) + j) * RW; }
        }
    }
    if (!fits) {
        const uint64_t o = atomic_add_u64(P.spill_cursor, 1ULL);
        if (o >= P.spill_cap) { *P.error = 6; return nullptr; }
        P.spill_part[o] = lpart;
        dst = P.spill_recs + o * RW;
    }
    return dst;
}
// the RW words of a record (w[0] least significant: the meta word) to dst
// Records of 32 bytes and more leave as 16-byte stores (RW is even, a record is 16-byte aligned): config-4 share scan 53.8 -> 48.8 ms, config-5 share
// 19.0 -> 17.9.  The 16-byte record of a ONE-word k-mer stays TWO 8-byte stores: the L2 then counts 3.2 G write requests for 1.6 G records (4.95 G requests
// in all) -- and still runs the pass in 66 ms, while ONE 16-byte store per record (3.34 G requests) takes 132 - 150 ms: a lane's scattered 16-byte store
// behind its returning atomic runs at the 11 G/s of round 4's micro-benchmark (profiles/r04_micro_sector_store.log), two 8-byte stores -- the second
// one hits the sector the first opened -- at more than twice that (profiles/r05_scan_request_counters.log).
template <int W>
CDBG_DEV void scan_store_words(uint64_t* dst, const uint64_t (&w)[RecFmt<W>::RW]) {
    constexpr int RW = RecFmt<W>::RW;
    static_assert(RW % 2 == 0, "records are pairs of words");
    if (RW == 2 && !CDBG_REC16_W1) {
        // (the compiler barrier keeps the pair apart: with both words in registers at once it merges them into one global_store_dwordx4 -- round 6's first
        //  placement kernel ran 0.8 G records in 53 ms that way, 35 ms with two stores: profiles/r06_ab_overlap_place_vs_count.log)
        dst[1] = w[1]; CDBG_COMPILER_BARRIER(); dst[0] = w[0];
        return;
    }
#pragma unroll
    for (int i = RW - 2; i >= 0; i -= 2) {
        RecPair v; v.lo = w[i]; v.hi = w[i + 1];
        *reinterpret_cast<RecPair*>(&dst[i]) = v;
    }
}
// the RW words of the record whose first base sits at bit offset bitoff of the tile's packed 2-bit stream
template <int W>
CDBG_DEV void scan_record_words(const uint32_t* pk, int bitoff, uint32_t meta, uint64_t (&w)[RecFmt<W>::RW]) {
    constexpr int RW = RecFmt<W>::RW;
#pragma unroll
    for (int wv = 0; wv < RW; ++wv) {
        uint64_t x = scan_get64(pk, bitoff + 64 * wv);
        if (wv == RW - 1) x = (x & ~0xFFFFULL) | meta;
        w[RW - 1 - wv] = x;
    }
}
template <int W, int MODE>
CDBG_DEV void scan_finish_record(const ScanParams& P, const uint32_t* pk, int bitoff, uint32_t meta, uint32_t lpart, uint32_t j) {
    constexpr int RW = RecFmt<W>::RW;
    if (MODE == SCAN_HIST) return;
    uint64_t* dst = scan_record_dst<W, MODE>(P, lpart, j);
    if (dst == nullptr) return;
    // (words cut out of the tile's packed stream and stored one by one / pair by pair: for the 16-byte record the two 8-byte stores stay apart this way)
    if (RW == 2 && !CDBG_REC16_W1) {
#pragma unroll
        for (int wv = 0; wv < RW; ++wv) {
            uint64_t x = scan_get64(pk, bitoff + 64 * wv);
            if (wv == RW - 1) x = (x & ~0xFFFFULL) | meta;
            dst[RW - 1 - wv] = x;
        }
        return;
    }
#pragma unroll
    for (int wv = 0; wv < RW; wv += 2) {
        const uint64_t hi = scan_get64(pk, bitoff + 64 * wv);
        uint64_t lo = scan_get64(pk, bitoff + 64 * (wv + 1));
        if (wv + 1 == RW - 1) lo = (lo & ~0xFFFFULL) | meta;
        RecPair v; v.lo = lo; v.hi = hi;
        *reinterpret_cast<RecPair*>(&dst[RW - 2 - wv]) = v;
    }
}
// sdefer: the workgroup's stream cursors (LDS, SCAN_DEFER_MAX words, zeroed by the kernel when it starts)
template <int W, int MODE>
CDBG_DEV void scan_emit_record(const ScanParams& P, const uint32_t* pk, int bitoff, uint32_t meta, uint32_t lpart, uint32_t* sdefer) {
    if (MODE == SCAN_EMIT_CAPPED && P.defer_slices > 1u) {
        // deferred placement (ScanParams): a record of slice q > 0 joins the workgroup's segment of stream q.  The lanes of the wave that hold a record of
        // the same slice take consecutive slots behind ONE LDS atomic (wave_append_slots, devrt.h) and write 16 bytes each side by side; divergent control flow
        const uint32_t sl = (uint32_t)(P.defer_map >> (4u * (lpart >> P.defer_shift))) & 15u;
        if (sl) {
            constexpr int RW = RecFmt<W>::RW;
            uint32_t slot = P.defer_seg_cap;
            for (uint32_t q = 1; q < P.defer_slices; ++q)     // (uniform trip count; the body runs for the lanes of slice q)
                if (sl == q) slot = wave_append_slots(&sdefer[q]);
            if (slot < P.defer_seg_cap) {
                const uint64_t at = ((uint64_t)(sl - 1u) * P.defer_nseg + blockIdx.x) * P.defer_seg_cap + slot;
                uint64_t w[RW];
                scan_record_words<W>(pk, bitoff, meta, w);
                uint64_t* dst = P.defer_recs + at * RW;
#pragma unroll
                for (int i = 0; i < RW; i += 2) { RecPair v; v.lo = w[i]; v.hi = w[i + 1]; *reinterpret_cast<RecPair*>(&dst[i]) = v; }
                P.defer_part[at] = lpart;
                return;
            }                                                // (segment full: placed here and now)
        }
    }
    scan_finish_record<W, MODE>(P, pk, bitoff, meta, lpart, scan_reserve_record<MODE>(P, lpart));
}
// the workgroup's stream cursors: cleared when the kernel starts, published when it ends (both behind / before a workgroup barrier of the caller)
template <int MODE>
CDBG_DEV void scan_defer_init(const ScanParams& P, uint32_t* sdefer) { if (MODE == SCAN_EMIT_CAPPED && P.defer_slices > 1u && threadIdx.x < (unsigned)SCAN_DEFER_MAX) sdefer[threadIdx.x] = 0; }
template <int MODE>
CDBG_DEV void scan_defer_publish(const ScanParams& P, const uint32_t* sdefer) {
    if (MODE == SCAN_EMIT_CAPPED && P.defer_slices > 1u && threadIdx.x >= 1u && threadIdx.x < P.defer_slices)
        P.defer_count[(uint64_t)(threadIdx.x - 1u) * P.defer_nseg + blockIdx.x] = sdefer[threadIdx.x];
}

// ---- deferred placement, second half: stream `slice` of a scan with ScanParams::defer_slices > 1 into the partition regions ----
// One-wave workgroups, up to four records per lane and round: that many returning atomics in flight before the first store.  No LDS and few registers: the
// waves fit beside the count stage's workgroups (which fill the CU's LDS but only 24 of its 32 wave slots), and 4 waves per CU saturate the memory side
// (profiles/r06_ab_overlap_place_vs_count.log: 0.8 G records alone 35 ms; beside k_count_fast<1> 40 ms, and the count 57 -> 70 ms instead of 57 + 35).
struct PlaceParams { ScanParams sp; uint32_t slice; };
template <int W>
__global__ void __launch_bounds__(64) k_place(PlaceParams Q) {
    constexpr int RW = RecFmt<W>::RW;
    constexpr int PLACE_U = RW <= 2 ? 4 : RW <= 4 ? 2 : 1;   // (records in flight per lane: bounded by registers for the wider records)
    const ScanParams& P = Q.sp;
    for (uint32_t seg = blockIdx.x; seg < P.defer_nseg; seg += gridDim.x) {
        const uint64_t sidx = (uint64_t)(Q.slice - 1u) * P.defer_nseg + seg;
        const uint32_t n0 = uni_u32(P.defer_count[sidx]);
        const uint32_t n = n0 < P.defer_seg_cap ? n0 : P.defer_seg_cap;   // (appends beyond the capacity were placed by the scan)
        const uint64_t base = sidx * P.defer_seg_cap;
        for (uint32_t b = 0; b < n; b += 64u * PLACE_U) {
            uint64_t w[PLACE_U][RW]; uint32_t p[PLACE_U], j[PLACE_U];
#pragma unroll
            for (int u = 0; u < PLACE_U; ++u) {
                const uint32_t i = b + (uint32_t)u * 64u + threadIdx.x;
                p[u] = 0xFFFFFFFFu;
                if (i < n) {
                    p[u] = P.defer_part[base + i];
                    const uint64_t* src = P.defer_recs + (base + i) * RW;
#pragma unroll
                    for (int q = 0; q < RW; q += 2) { const RecPair v = *reinterpret_cast<const RecPair*>(&src[q]); w[u][q] = v.lo; w[u][q + 1] = v.hi; }
                }
            }
#pragma unroll
            for (int u = 0; u < PLACE_U; ++u) if (p[u] != 0xFFFFFFFFu) j[u] = scan_reserve_record<SCAN_EMIT_CAPPED>(P, p[u]);
#pragma unroll
            for (int u = 0; u < PLACE_U; ++u) if (p[u] != 0xFFFFFFFFu) {
                uint64_t* dst = scan_record_dst<W, SCAN_EMIT_CAPPED>(P, p[u], j[u]);
                if (dst != nullptr) scan_store_words<W>(dst, w[u]);
            }
        }
    }
}

template <int W, int MODE>
__global__ void __launch_bounds__(SCAN_THREADS) k_scan(ScanParams P) {
    constexpr int RW = RecFmt<W>::RW;
    constexpr int CAPB = RecFmt<W>::CAPB;
    CDBG_SHARED uint32_t pk[SCAN_PKW];
    CDBG_SHARED uint32_t vm[SCAN_PKW / 2 + 4];
    CDBG_SHARED uint32_t ka[SCAN_NKEY];
    CDBG_SHARED uint32_t kb[SCAN_NKEY];
    CDBG_SHARED uint64_t brk[SCAN_TILE / 64 + 2];
    CDBG_SHARED uint64_t stt[SCAN_TILE / 64 + 2];
    CDBG_SHARED uint32_t s_members, s_trav, s_nrec, s_nstart;
    CDBG_SHARED uint32_t s_defer[SCAN_DEFER_MAX];            // deferred placement: this workgroup's stream cursors
    constexpr uint32_t LIST_CAP = (SCAN_NKEY - SCAN_TILE / 2) / 2;     // staged records (two words each) behind the run-start list

    const int tid = threadIdx.x;
    const int k = P.k, m = P.m;
#if defined(CDBG_PROFILE_PHASES) && !defined(CDBG_HOSTSIM)
    uint64_t sph[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t st_prev = wall_clock64();
#endif
    uint64_t n_members = 0, n_trav = 0;
    if (tid == 0) { s_members = 0; s_trav = 0; }
    scan_defer_init<MODE>(P, s_defer);                       // (the first tile's barriers come before any record)
    // persistent workgroups stride over the tiles (a workgroup launch per ~30 us tile costs more than the tile)
    for (uint64_t tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
    const int64_t t0 = ((int64_t)tile * P.tile_stride + P.tile_offset) * SCAN_TILE;
    const int64_t base = t0 - 16;                     // byte offset of tile-local base index 0

    // ---- 1. load 16 bytes per lane-iteration, encode to 2 bit + validity ----
    for (int w = tid; w < SCAN_PKW; w += SCAN_THREADS) {
        const int64_t off = base + 16 * (int64_t)w;
        uint32_t packed = 0, vbits = 0;
        if (off >= 0 && off < (int64_t)P.nbytes_padded) {
            const uint4 v = *reinterpret_cast<const uint4*>(P.reads + off);
            const uint32_t wd[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t c = (wd[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                const uint32_t ok = ((c & 0xC0u) == 0x40u) & ((0x10008Au >> (c & 0x1Fu)) & 1u);
                packed |= base_code(c) << (30 - 2 * j);
                vbits |= ok << j;
            }
        }
        pk[w] = packed;
        reinterpret_cast<uint16_t*>(vm)[w] = (uint16_t)vbits;
    }
    if (tid < 4) vm[SCAN_PKW / 2 + tid] = 0;           // over-read words of scan_all_valid
    if (tid == 0) s_nrec = 0;
    __syncthreads();
    CDBG_SPH(0);

    // ---- 2. m-mer ordering keys; index i <-> tile base index q = 15 + i ----
    const int WN = k - m;                              // m-mers per junction
    const int nkey = SCAN_TILE + WN + 1;
    for (int i = tid; i < SCAN_NKEY; i += SCAN_THREADS) {
        uint32_t key = 0xFFFFFFFFu;
        if (i < nkey) {
            const int q = 15 + i;
            if (scan_all_valid(vm, q, m)) {
                const uint64_t x = ((uint64_t)pk[q >> 4] << 32) | pk[(q >> 4) + 1];
                const uint32_t v = (uint32_t)((x << (2 * (q & 15))) >> (64 - 2 * m));
                key = mmer_key(v, m);
            }
        }
        ka[i] = key;
    }
    __syncthreads();
    CDBG_SPH(1);

    // ---- 3. sliding-window minimum of width WN by doubling (ping-pong in LDS) ----
    uint32_t* cur = ka; uint32_t* oth = kb;
    int width = 1;
    // (radix 4 first: a window of 111 keys is 3 passes of four reads + the final combine instead of 6 doubling passes + combine --
    //  the same reads, half the writes and barriers; this phase is half of the kernel at k = 127 since the run structure went bit-parallel)
    while (4 * width <= WN) {
        for (int i = tid; i < SCAN_NKEY; i += SCAN_THREADS) {
            uint32_t a = cur[i];
#pragma unroll
            for (int j = 1; j < 4; ++j) { const uint32_t b = (i + j * width < SCAN_NKEY) ? cur[i + j * width] : 0xFFFFFFFFu; a = a < b ? a : b; }
            oth[i] = a;
        }
        __syncthreads();
        uint32_t* t = cur; cur = oth; oth = t;
        width *= 4;
    }
    while (2 * width <= WN) {
        for (int i = tid; i < SCAN_NKEY; i += SCAN_THREADS) {
            const uint32_t a = cur[i];
            const uint32_t b = (i + width < SCAN_NKEY) ? cur[i + width] : 0xFFFFFFFFu;
            oth[i] = a < b ? a : b;
        }
        __syncthreads();
        uint32_t* t = cur; cur = oth; oth = t;
        width *= 2;
    }
    if (width != WN) {
        const int d = WN - width;
        for (int i = tid; i < SCAN_NKEY; i += SCAN_THREADS) {
            const uint32_t a = cur[i];
            const uint32_t b = (i + d < SCAN_NKEY) ? cur[i + d] : 0xFFFFFFFFu;
            oth[i] = a < b ? a : b;
        }
        __syncthreads();
        uint32_t* t = cur; cur = oth; oth = t;
    }
    const uint32_t* g = cur;                           // g[jq], jq in [0, TILE+1]; junction jq <-> q = 15 + jq
    uint32_t* lst = oth + SCAN_TILE / 2;               // the idle ping-pong buffer: run-start list (u16 x TILE), then this tile's staged records
    CDBG_SPH(2);

    // ---- 4. run structure, bit-parallel: brk bit (jq-1) = junction jq does NOT continue the previous run ----
    // (a) every lane: is g equal to the junction before?  One compare per junction, gathered into 64-bit words by ballot.
    for (int it = 0; it < SCAN_TILE / SCAN_THREADS; ++it) {
        const int jq = 1 + it * SCAN_THREADS + tid;
        const unsigned long long em = __ballot(g[jq] == g[jq - 1]);
        if ((tid & 63) == 0) brk[(jq - 1) >> 6] = em;
    }
    __syncthreads();
    // (b) one wave, lane w = the 64 junctions jq = 64 w + 1 .. 64 w + 64: validity of a junction = its k-1 bases all valid, for 64
    // junctions at once by AND-doubling over a 320-bit window of the validity bits (a 126-bit window test per junction and lane was
    // 29 % of the kernel at k = 127); then continue / break / start words, and the run starts compacted into a list (walking all
    // junctions for the ~70 runs of a tile kept 98 % of the lanes idle through 16 divergent iterations: another 35 %).
    uint16_t* const sl = reinterpret_cast<uint16_t*>(oth);                       // [SCAN_TILE] run starts (jq - 1), ascending
    if (tid < 64) {
        const int K1 = k - 1, q0 = 16 + 64 * tid;                               // junction jq <-> tile base index q = 15 + jq
        constexpr int NXW = 5;                                                   // 320-bit window: 64 junctions + k - 1 <= 254 further bases
        uint64_t X[NXW];
#pragma unroll
        for (int j = 0; j < NXW; ++j) {
            const int bit = q0 + 64 * j, w = bit >> 5, sh = bit & 31;
            const uint64_t lo = ((uint64_t)vm[w + 1] << 32) | vm[w];
            X[j] = sh ? ((lo >> sh) | ((uint64_t)vm[w + 2] << (64 - sh))) : lo;
        }
        auto xw = [&](int i) -> uint64_t {                                      // X[i] without a runtime index (a select chain keeps the words in registers)
            uint64_t r = 0;
#pragma unroll
            for (int j = 0; j < NXW; ++j) r = (i == j) ? X[j] : r;
            return r;
        };
        auto shr_and = [&](int sft) {                                           // X &= X >> sft (320-bit, zero fill), sft wave-uniform, 1 .. 254
            const int ws = sft >> 6, bs = sft & 63;
#pragma unroll
            for (int i = 0; i < NXW; ++i) {                                      // ascending: X[i] only reads words at or above i
                const uint64_t a0 = xw(i + ws), a1 = xw(i + ws + 1);
                X[i] &= bs ? ((a0 >> bs) | (a1 << (64 - bs))) : a0;
            }
        };
        { int len = 1; while (2 * len <= K1) { shr_and(len); len *= 2; } if (K1 > len) shr_and(K1 - len); }
        const uint64_t v = X[0];
        // validity of the junction before each of mine: my word shifted up, with the top bit of the previous lane's word
        const uint32_t top = (uint32_t)(v >> 63);
        const uint32_t below = __shfl_up(top, 1u);
        const uint64_t vp = (v << 1) | (tid ? (uint64_t)below : 0ULL);
        uint64_t cont = v & vp & brk[tid];
        if (tid == 0) cont &= ~1ULL;                                            // the tile's first junction never continues a run
        const uint64_t starts = v & ~cont;
        brk[tid] = ~cont; stt[tid] = starts;
        const uint32_t cnt = (uint32_t)__popcll(starts);
        uint32_t off = wave_incl_sum_u32(cnt) - cnt;
        unsigned long long bits = starts;
        while (bits) { const int j = __ffsll((long long)bits) - 1; bits &= bits - 1; sl[off++] = (uint16_t)(tid * 64 + j); }
        if (tid == 63) s_nstart = off;
    }
    if (tid == 64) { brk[SCAN_TILE / 64] = ~0ULL; brk[SCAN_TILE / 64 + 1] = ~0ULL; }
    __syncthreads();
    CDBG_SPH(3);

    // ---- 5. one lane per run start: find the run end, apply the boundary rules, emit ----
    const int NMAX = CAPB - k + 1 < 255 ? CAPB - k + 1 : 255;   // member k-mers per record (the count is an 8-bit field of the meta word: eight-word records have room for more)
    const uint32_t rank_mask = (1u << P.rank_bits) - 1u;
    const int nstart = (int)s_nstart;
    for (int si = tid; si < nstart; si += SCAN_THREADS) {                       // one lane per run
        const int bit = sl[si];
        const int jq = bit + 1;
        const uint32_t gq = g[jq];
        const uint32_t part = part_of(gq, P.log_np);
        // own partitions only (every rank scans the same text), or -- reads sharded over the ranks -- all partitions, laid
        // out owner-major ([owner][local partition]) so that each owner's block of the record array is contiguous
        if (!P.emit_all && (part & rank_mask) != (uint32_t)P.rank) continue;
        const uint32_t lpart = P.emit_all ? (part & rank_mask) * P.npl + (part >> P.rank_bits) : part >> P.rank_bits;
        // run end: next break bit after `bit`
        int e;
        {
            int nb = bit + 1;
            unsigned long long wv = brk[nb >> 6] >> (nb & 63);
            if (wv) e = nb + __ffsll((long long)wv) - 1;
            else {
                int wi = (nb >> 6) + 1;
                while (brk[wi] == 0) ++wi;
                e = wi * 64 + __ffsll((long long)brk[wi]) - 1;
            }
            // e = bit index of the next break => last junction of the run is jq index e (bit e-1)
            if (e > SCAN_TILE) e = SCAN_TILE;
        }
        const int s = jq;                              // run = junctions [s, e]
        // boundary members
        bool first_incl = false, first_trav = false, last_incl = false, last_trav = false, first_foreign = false, last_foreign = false;
        if (scan_all_valid(vm, 15 + s - 1, k)) {       // k-mer s-1 (its right junction is s)
            const uint32_t g2 = g[s - 1];
            first_foreign = part_of(g2, P.log_np) != part;   // the junction this k-mer shares with the run before belongs to another bucket
            if (gq < g2) first_incl = true;
            else if (first_foreign) { first_incl = true; first_trav = true; }
        }
        if (scan_all_valid(vm, 15 + e, k)) {           // k-mer e (its left junction is e)
            const uint32_t g2 = g[e + 1];
            last_foreign = part_of(g2, P.log_np) != part;
            if (gq < g2) last_incl = true;
            else if (last_foreign) { last_incl = true; last_trav = true; }
            else if (gq == g2) last_incl = true;       // artificial split (tile edge): keep it here
        }
        // chunk the run into records of at most NMAX members
        int c = s; bool firstchunk = true;
        while (c <= e) {
            const int ms = (firstchunk && first_incl) ? c - 1 : c;
            int ce = ms + NMAX - 1; if (ce > e) ce = e;
            const int me = (ce == e && !last_incl) ? e - 1 : ce;
            const int n = me - ms + 1;
            if (n > 0) {
                uint32_t meta = (uint32_t)n | (sub_of(gq, P.log_np) << 12);   // (bits 12-15: the minimizer's sub-partition, for the multi-pass count)
                const bool ft = firstchunk && first_incl && first_trav;
                const bool lt = (ce == e) && last_incl && last_trav;
                if (ft) meta |= 0x100u;
                if (lt) meta |= 0x200u;
                if (firstchunk && first_incl && first_foreign) meta |= 0x400u;
                if ((ce == e) && last_incl && last_foreign) meta |= 0x800u;
                // stage the record in LDS; all lanes emit together afterwards, so a tile costs two
                // rounds of device-atomic latency instead of one per loop iteration
                const uint32_t li = atomic_add_u32(&s_nrec, 1u);
                if (li < LIST_CAP) { lst[2 * li] = (uint32_t)ms | (meta << 16); lst[2 * li + 1] = lpart; }
                else scan_emit_record<W, MODE>(P, pk, 2 * (15 + ms), meta, lpart, s_defer);   // list full (low-complexity tile)
                n_members += (uint64_t)n;
                n_trav += (ft ? 1 : 0) + (lt ? 1 : 0);
            }
            firstchunk = false;
            c = ce + 1;
        }
    }
    __syncthreads();
    CDBG_SPH(4);
    {
        const uint32_t nl = s_nrec < LIST_CAP ? s_nrec : LIST_CAP;
        for (uint32_t i = tid; i < nl; i += SCAN_THREADS)
            scan_emit_record<W, MODE>(P, pk, 2 * (15 + (int)(lst[2 * i] & 0xFFFFu)), lst[2 * i] >> 16, lst[2 * i + 1], s_defer);
    }
    __syncthreads();                                    // the next tile reuses the LDS arrays
    CDBG_SPH(5);
    }
#if defined(CDBG_PROFILE_PHASES) && !defined(CDBG_HOSTSIM)
    if (threadIdx.x == 0) for (int i = 0; i < 6; ++i) atomic_add_u64(&P.stats[16 + i], sph[i]);
#endif
    scan_defer_publish<MODE>(P, s_defer);                    // (behind the last tile's closing barrier)
    if (MODE != SCAN_EMIT || P.var_limit) {             // one device atomic per workgroup, not per lane (the exact layout's histogram pass counted already)
        uint32_t nm = (uint32_t)n_members, nt = (uint32_t)n_trav;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { nm += __shfl_xor(nm, d); nt += __shfl_xor(nt, d); }
        if ((tid & 63) == 0) { if (nm) atomic_add_u32(&s_members, nm); if (nt) atomic_add_u32(&s_trav, nt); }
        __syncthreads();
        if (tid == 0 && s_members) { atomic_add_u64(&P.stats[0], (uint64_t)s_members); atomic_add_u64(&P.stats[1], (uint64_t)s_trav); }
    }
}

// ---- exclusive prefix sum of n uint32 counts into n+1 uint64 offsets ----
// three launches: per-block sums, scan of the block sums (one workgroup), block-local scan + base
constexpr int EXSCAN_THREADS = 1024;
constexpr int EXSCAN_ITEMS = 4;                          // consecutive counts per lane
constexpr int EXSCAN_BLOCK = EXSCAN_THREADS * EXSCAN_ITEMS;

// inclusive scan of one value per lane over the workgroup; returns the lane's inclusive value and the block total
CDBG_DEV uint64_t exscan_block_incl(uint64_t v, uint64_t* wsum /* [EXSCAN_THREADS/64] LDS */, uint64_t& total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint64_t x = __shfl_up(incl, d); if (lane >= d) incl += x; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint64_t basew = 0, tot = 0;
    for (int w = 0; w < EXSCAN_THREADS / 64; ++w) { const uint64_t x = wsum[w]; if (w < wave) basew += x; tot += x; }
    total = tot;
    __syncthreads();
    return incl + basew;
}
__global__ void __launch_bounds__(EXSCAN_THREADS) k_exscan_sums(const uint32_t* cnt, uint64_t* bsum, uint64_t n) {
    CDBG_SHARED uint64_t wsum[EXSCAN_THREADS / 64];
    const uint64_t i0 = (uint64_t)blockIdx.x * EXSCAN_BLOCK + (uint64_t)threadIdx.x * EXSCAN_ITEMS;
    uint64_t v = 0;
    for (int j = 0; j < EXSCAN_ITEMS; ++j) if (i0 + j < n) v += cnt[i0 + j];
    uint64_t total; exscan_block_incl(v, wsum, total);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}
__global__ void __launch_bounds__(EXSCAN_THREADS) k_exscan_top(uint64_t* bsum, uint64_t nb, uint64_t* off_n) {
    CDBG_SHARED uint64_t wsum[EXSCAN_THREADS / 64];
    uint64_t carry = 0;
    for (uint64_t b0 = 0; b0 < nb; b0 += EXSCAN_THREADS) {  // uniform trip count
        const uint64_t i = b0 + threadIdx.x;
        const uint64_t v = i < nb ? bsum[i] : 0;
        uint64_t total; const uint64_t incl = exscan_block_incl(v, wsum, total);
        if (i < nb) bsum[i] = carry + incl - v;              // exclusive
        carry += total;
    }
    if (threadIdx.x == 0) *off_n = carry;
}
__global__ void __launch_bounds__(EXSCAN_THREADS) k_exscan_apply(const uint32_t* cnt, const uint64_t* bsum, uint64_t* off, uint64_t n) {
    CDBG_SHARED uint64_t wsum[EXSCAN_THREADS / 64];
    const uint64_t i0 = (uint64_t)blockIdx.x * EXSCAN_BLOCK + (uint64_t)threadIdx.x * EXSCAN_ITEMS;
    uint32_t c[EXSCAN_ITEMS]; uint64_t v = 0;
    for (int j = 0; j < EXSCAN_ITEMS; ++j) { c[j] = (i0 + j < n) ? cnt[i0 + j] : 0u; v += c[j]; }
    uint64_t total; const uint64_t incl = exscan_block_incl(v, wsum, total);
    uint64_t acc = bsum[blockIdx.x] + incl - v;
    for (int j = 0; j < EXSCAN_ITEMS; ++j) { if (i0 + j < n) off[i0 + j] = acc; acc += c[j]; }
}
__global__ void k_max_u32(const uint32_t* a, uint64_t n, uint64_t* out) {        // *out = max(*out, max a[i]) (out zeroed by the host)
    uint32_t m = 0; const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) m = a[i] > m ? a[i] : m;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(m, d); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m) atomic_max_u32(reinterpret_cast<uint32_t*>(out), m);
}
__global__ void k_copy_u64(const uint64_t* src, uint64_t* dst, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// exact layout of fewer than 2^32 records: the 32-bit running indices start at the regions' offsets; afterwards back to record counts
__global__ void k_cursor32_load(const uint64_t* off, uint32_t* cur, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cur[i] = (uint32_t)off[i];
}
__global__ void k_cursor32_counts(const uint64_t* off, uint32_t* cur, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cur[i] -= (uint32_t)off[i];
}

// ---- counter-based synthetic reads, resident in HBM ----
// identical bit-for-bit to oracle/cdbg_oracle.c:orc_synth_reads
//   cfg 0 .. 255       the plain generator: uniform random genome at 30x, uniform read starts, 1 % substitutions
//   cfg | GEN_HOSTILE  the same reads over a HOSTILE genome (what uniform random data never exercises): 1 block in 20 of
//                      4096 bases is two-letter low complexity, up to 1000 exact copies of one 5 kbp repeat, 50 homopolymer
//                      runs of 600 bases, and a third of the reads start inside 1/40 of the genome (~20x coverage skew)
//   cfg < 0            every base 'A' (test hook: ONE k-mer seen n_reads * (read_len - k + 1) times, count saturation)
constexpr int GEN_HOSTILE = 0x100;
CDBG_HD uint32_t gen_genome_base(uint64_t gp, uint64_t G, uint64_t seed_g, bool hostile) {
    const uint32_t plain = (uint32_t)(mix64(seed_g + gp) >> 62);
    if (!hostile) return plain;
    const uint64_t hp_stride = G / 50;                    // homopolymer runs
    if (hp_stride >= 2400) {
        const uint64_t off = gp % hp_stride;
        if (off >= hp_stride / 2 && off < hp_stride / 2 + 600) return (uint32_t)((gp / hp_stride) & 3u);
    }
    uint64_t nc = G / 50000; if (nc > 1000) nc = 1000;    // copies of the 5 kbp repeat
    if (nc) {
        const uint64_t rs = G / nc, off = gp % rs;
        if (off >= rs / 4 && off < rs / 4 + 5000) return (uint32_t)(mix64((seed_g ^ 0x5EED0000ULL) + (off - rs / 4)) >> 62);
    }
    const uint64_t bh = mix64(seed_g + 0x10C0000000ULL + (gp >> 12));   // two-letter blocks
    if (bh % 20 == 0) {
        const uint32_t a = (uint32_t)(bh >> 8) & 3u, b2 = (a + 1u + (uint32_t)((bh >> 16) % 3)) & 3u;
        return (plain & 2u) ? a : b2;
    }
    return plain;
}
CDBG_HD uint64_t gen_read_start(uint64_t u, uint64_t G, uint64_t L, bool hostile) {
    if (hostile && (u >> 32) % 3 == 0) {
        uint64_t hot = G / 40; if (hot < L) hot = L;
        return G / 2 - hot / 2 + u % (hot - L + 1);
    }
    return u % (G - L + 1);
}
struct GenParams {
    uint8_t* out; uint64_t first_read, n_reads, total_reads, read_len; int cfg;
};
__global__ void k_gen_reads(GenParams P) {
    const uint64_t L1 = P.read_len + 1;
    const uint64_t total = P.n_reads * L1, stride = (uint64_t)gridDim.x * blockDim.x;
    const bool hostile = P.cfg >= 0 && (P.cfg & GEN_HOSTILE);
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const uint64_t i = idx / L1, j = idx % L1;
    if (j == P.read_len) { P.out[idx] = '\n'; continue; }
    if (P.cfg < 0) { P.out[idx] = 'A'; continue; }
    const uint64_t SEED_G = 0xBCA10000ULL + (uint64_t)P.cfg, SEED_R = 0xBCA11000ULL + (uint64_t)P.cfg,
                   SEED_E = 0xBCA12000ULL + (uint64_t)P.cfg;
    uint64_t G = (P.total_reads * P.read_len + 29) / 30;
    if (G < P.read_len) G = P.read_len;
    const uint64_t r = P.first_read + i;
    const uint64_t start = gen_read_start(mix64(SEED_R + 2 * r), G, P.read_len, hostile);
    const int strand = (int)(mix64(SEED_R + 2 * r + 1) & 1ULL);
    const uint64_t gp = strand ? start + P.read_len - 1 - j : start + j;
    uint32_t b = gen_genome_base(gp, G, SEED_G, hostile);
    if (strand) b = 3u - b;
    const uint64_t x = mix64(SEED_E + r * P.read_len + j);
    if (x % 10000 < 100) b = (b + 1 + (uint32_t)((x >> 32) % 3)) & 3u;
    P.out[idx] = (uint8_t)("ACGT"[b]);
    }
}

}  // namespace cdbg
