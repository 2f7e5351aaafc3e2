// k_count.h -- stage 1b: per-partition k-mer counting in an LDS open-address table.
//
// New MI355X design for the job gatb-core's DSK partition counting does on CPUs
// (hot path entered at /root/reference/src/bcalm_1.cpp:57;
// abundance filter semantics /root/reference/README.md:23-25: keep count >= min).
//
// One workgroup owns one minimizer partition: its super-k-mer records are streamed
// from HBM exactly once (coalesced), expanded to canonical k-mers in registers and
// counted in a table that lives in the CU's LDS (ds_cmpst_b64 claims the slot,
// ds_add bumps the count) -- LDS sustains ~10x the rate of device-scope atomics on
// MI355X (bench_micro/: 274 G inserts/s vs 17-27 G atomics/s).  Solid k-mers
// (count >= abundance_min; home AND traveller copies) are appended as one contiguous
// segment per partition.  A partition whose distinct k-mers do not fit LDS is put on
// the `big` list and re-run by the same code with its table in HBM scratch.
#pragma once
#include "k_scan_fast.h"

namespace cdbg {

#if defined(CDBG_PROFILE_PHASES) && !defined(CDBG_HOSTSIM)
#define CDBG_PH(i) do { if (threadIdx.x == 0) { const uint64_t t_ = wall_clock64(); ph[i] += t_ - t_prev; t_prev = t_; } } while (0)
#else
#define CDBG_PH(i) do { } while (0)
#endif
// per-wave member map entries (records of a batch x members per record).  Kept small on purpose: with
// 416 B per wave the W=1 kernel stays under 53 KB of LDS = 3 workgroups (24 waves) per CU, which
// measured 13 % faster than 64-record batches at 2 workgroups per CU.
constexpr uint32_t COUNT_CHUNK = 8192;                 // solid entries a workgroup reserves per device atomic (>= 3/4 of the largest one-pass table)
constexpr uint32_t TRAV_FLAG = 0x80000000u;          // in a count word: this entry is a traveller copy
// Abundances are 31-bit and SATURATE (gatb-core's own ceiling is its `-abundance-max` default 2147483647 [UPSTREAM-RECALL]):
// exact below COUNT_SAT, reported as COUNT_MAX = 2^31 - 1 from there on.  The one-pass kernel (k_count_fast.h) cannot
// reach the ceiling -- it defers every partition of 2^23 records (>= 2^31 - 2^23 member k-mers) or more to the kernels
// of this file -- so its hot add stays a plain ds_add_u32.  Here an add that finds the count at or above COUNT_SAT takes
// its increment back: the stored value never gets within 4096 in-flight increments of bit 31 (the traveller flag) and
// ends at exactly min(count, COUNT_SAT) whatever the interleaving.
constexpr uint32_t COUNT_MAX = 0x7FFFFFFFu, COUNT_SAT = COUNT_MAX - 4095u;
#ifndef CDBG_COUNT_MAX_SUB
#define CDBG_COUNT_MAX_SUB 1
#endif
constexpr uint32_t COUNT_MAX_SUB = CDBG_COUNT_MAX_SUB;  // multi-pass kernel: passes that divide a partition by its records' sub-partition (1: by k-mer hash only, as in round 4)
constexpr uint32_t COUNT_FAST_MAX_RECORDS = 1u << 23;   // 255 members x (2^23 - 1) records < COUNT_SAT
CDBG_DEV void count_add_sat(uint32_t* p) {
    const uint32_t old = atomic_add_u32(p, 1u);
    if ((old & COUNT_MAX) >= COUNT_SAT) atomic_sub_u32(p, 1u);
}
CDBG_HD uint32_t count_value(uint32_t word) { const uint32_t n = word & COUNT_MAX; return n >= COUNT_SAT ? COUNT_MAX : n; }   // abundance of a count word
CDBG_HD uint32_t count_word_out(uint32_t word) { return (word & TRAV_FLAG) | count_value(word); }                            // what the solid array holds
// Multi-word keys (W > 1) are claimed through their TOP word: a k-mer or (k-1)-mer of k < 32 W (the span rule) leaves
// the two top bits of that word clear.  The join tables of the glue (k_glue.h) use all-ones for "empty" and bit 63 for
// "claimed, lower words not written yet".  The k-mer tables of this stage put two facts about the k-mer into those
// bits (KEY_FOREIGN_*: never both), so there BOTH bits set mean a state: all-ones = empty, anything else = "claimed,
// lower words not written yet" with the low 62 bits of the claimer's top word beside it (key_pending): a lane that meets
// the claim of ANOTHER key moves on at once, only a lane with the same 62 bits looks again.
// One 64-bit compare-and-swap per probe, no separate state array.
constexpr uint64_t KEY_EMPTY = ~0ULL, KEY_PENDING = 1ULL << 63;
// A solid k-mer travels from the count tables to the compaction with two flags in the top bits of its top word: the
// junction at the LEFT / RIGHT end of its canonical label belongs to ANOTHER bucket.  The scan knows (record meta bits
// 10 / 11, k_scan.h); the flags are a function of the k-mer and of the bucket, so every occurrence of a k-mer in a
// bucket forms the same 64 W-bit key and the tables need no extra operation for them.  A home k-mer has at least one
// junction in its bucket, a traveller exactly one: both flags together do not occur.
constexpr uint64_t KEY_FOREIGN_L = 1ULL << 62, KEY_FOREIGN_R = 1ULL << 63, KEY_FLAGS = KEY_FOREIGN_L | KEY_FOREIGN_R;
// the claim word of a key while its lower words are being written (62 ones would read as EMPTY: one less -- a lane whose
// own 62 bits are that value then waits for a key that is not its own, which ends when that key is published)
CDBG_HD uint64_t key_pending(uint64_t top) {
    const uint64_t low = top & ~KEY_FLAGS;
    return KEY_FLAGS | (low == ~KEY_FLAGS ? low - 1ULL : low);
}
// flags of a member k-mer: it is the first / last member of its record, it was stored in read orientation or reversed
CDBG_HD uint64_t key_flags(bool left_foreign_in_read, bool right_foreign_in_read, bool reversed) {
    const bool l = reversed ? right_foreign_in_read : left_foreign_in_read, r = reversed ? left_foreign_in_read : right_foreign_in_read;
    return (l ? KEY_FOREIGN_L : 0ULL) | (r ? KEY_FOREIGN_R : 0ULL);
}

// ---------------------------------------------------------------------------
// Open-address table of W-word keys.  W == 1: the key word itself is claimed with
// one 64-bit CAS (EMPTY = all ones, unreachable for k <= 31).  W > 1: a 32-bit state
// word per slot is claimed (EMPTY -> BUSY), the key words are written, then the state
// becomes the key's tag (bit 31 set); readers of a BUSY slot retry.
// ---------------------------------------------------------------------------
template <int W>
struct KTable {
    uint64_t* keys;      // [cap * W]
    uint32_t mask;       // cap - 1
};

template <int W>
CDBG_DEV void ktable_clear(const KTable<W>& t, int tid, int nthreads) {
    const uint32_t cap = t.mask + 1;
    for (uint32_t i = tid; i < cap; i += nthreads) t.keys[(uint64_t)i * W + (W - 1)] = KEY_EMPTY;   // the claim word only
}
template <int W>
CDBG_DEV bool ktable_used(const KTable<W>& t, uint32_t s) {
    return t.keys[(uint64_t)s * W + (W - 1)] != KEY_EMPTY;
}
template <int W>
CDBG_DEV Kmer<W> ktable_key(const KTable<W>& t, uint32_t s) {
    Kmer<W> r;
    for (int i = 0; i < W; ++i) r.w[i] = t.keys[(uint64_t)s * W + i];
    return r;
}
// Workgroup barrier for tables that may live in HBM scratch (GLOBAL): the CU's vector L1 is not
// guaranteed to reflect other waves' global stores / L2 atomics, so drop it after the barrier
// (agent-scope fence = buffer_inv; MI355X_MICROARCH "inter-workgroup visibility" applies to the
// stale-line case within a CU as well).  LDS tables need only the barrier.
template <bool GLOBAL>
CDBG_DEV void block_sync() {
    if (GLOBAL) { __threadfence(); __syncthreads(); __threadfence(); }
    else CDBG_LDS_BARRIER();                             // LDS tables: no need to drain the wave's global loads / stores
}
// find-or-insert; returns slot, sets is_new.  GLOBAL selects the fence flavour.
// Gives up (returns 0xFFFFFFFF) after max_probe occupied slots, so a full table cannot hang a lane.
template <int W, bool GLOBAL>
CDBG_DEV uint32_t ktable_insert(const KTable<W>& t, const Kmer<W>& key, bool& is_new, uint32_t max_probe = 0xFFFFFFFFu) {
    const uint32_t h = key.hash();
    uint32_t s = h & t.mask;
    is_new = false;
    if (W == 1) {
        // single-exit loop (one compare-and-swap, a few selects, one back edge): the early-return form costs twice
        // the instructions per probe once the compiler has structurised and unrolled its three exits
        uint32_t probes = 0; bool hit; uint64_t old;
#pragma clang loop unroll(disable)
        do {
            old = atomic_cas_u64(&t.keys[s], ~0ULL, key.w[0]);
            hit = (old == ~0ULL) | (old == key.w[0]);
            s = hit ? s : ((s + 1) & t.mask);
            ++probes;
        } while (!hit && probes < max_probe);
        is_new = old == ~0ULL;
        return hit ? s : 0xFFFFFFFFu;
    } else {
        // Single-exit loop with the publish INSIDE the iteration that claimed the slot: lanes of one wave that insert
        // the same key must see the claimer finish before the back edge (a wave has no independent thread scheduling,
        // so a claimer parked behind the loop exit while a sibling lane spins on the pending claim would never run again).
        const uint64_t top = key.w[W - 1], ptop = key_pending(top);
        uint32_t probes = 0, res = 0xFFFFFFFFu; bool done = false;
#pragma clang loop unroll(disable)
        do {
            uint64_t* const claim = &t.keys[(uint64_t)s * W + (W - 1)];
            const uint64_t old = atomic_cas_u64(claim, KEY_EMPTY, ptop);
            // LDS tables: the lower words are requested right behind the claim, whatever it returns -- one round trip for a
            // hit instead of two (a wave's LDS operations execute in order: a published top word seen by the compare-and-swap
            // means that the lower words read after it are the published ones).  HBM tables read them only when the top matches.
            // (two-word keys only: measured at k = 55, count 169 -> 154 ms; with three lower words to fetch per probe it lost, k = 127: 352 -> 355 ms)
            uint64_t low[W > 1 ? W - 1 : 1];
            constexpr bool SPEC = !GLOBAL && W == 2;
            if (SPEC) {
                CDBG_COMPILER_BARRIER();
#pragma unroll
                for (int i = 0; i < W - 1; ++i) low[i] = t.keys[(uint64_t)s * W + i];
            }
            // (selects instead of an if / else chain: the structurised chain cost a dozen scalar mask operations per probe)
            const bool mine = old == KEY_EMPTY, wait = old == ptop;
            bool same = old == top;
            if (SPEC) same &= (low[0] == key.w[0]);
            else if (same) {
#pragma unroll
                for (int i = 0; i < W - 1; ++i) same &= ((GLOBAL ? ld_agent_u64(&t.keys[(uint64_t)s * W + i]) : t.keys[(uint64_t)s * W + i]) == key.w[i]);
            }
            if (mine) {                                      // claimed: write the lower words, then publish the top word
#pragma unroll
                for (int i = 0; i < W - 1; ++i) t.keys[(uint64_t)s * W + i] = key.w[i];
                if (GLOBAL) __threadfence(); else CDBG_LDS_FENCE();
                atomic_exch_u64(claim, top);
            }
            if (wait) CDBG_SPIN_YIELD();                     // being written by another lane (this key, as far as one can tell): look again
            is_new = is_new | mine; done = mine | same; res = done ? s : res;
            const bool advance = !(done | wait);
            s = advance ? ((s + 1) & t.mask) : s; probes += advance ? 1u : 0u;
        } while (!done && probes < max_probe);
        return res;
    }
}
// lookup only (table no longer being modified); returns slot or 0xFFFFFFFF
template <int W>
CDBG_DEV uint32_t ktable_find(const KTable<W>& t, const Kmer<W>& key) {
    const uint32_t h = key.hash();
    uint32_t s = h & t.mask;
    // single-exit loops (see ktable_insert)
    if (W == 1) {
        uint64_t v; bool stop;
#pragma clang loop unroll(disable)
        do {
            v = t.keys[s];
            stop = (v == key.w[0]) | (v == ~0ULL);
            s = stop ? s : ((s + 1) & t.mask);
        } while (!stop);
        return v == key.w[0] ? s : 0xFFFFFFFFu;
    } else {
        const uint64_t top = key.w[W - 1];
        bool found = false, stop;
#pragma clang loop unroll(disable)
        do {
            const uint64_t v = t.keys[(uint64_t)s * W + (W - 1)];
            if (v == top) {
                bool eq = true;
                for (int i = 0; i < W - 1; ++i) eq &= (t.keys[(uint64_t)s * W + i] == key.w[i]);
                found = eq;
            }
            stop = found | (v == KEY_EMPTY);
            s = stop ? s : ((s + 1) & t.mask);
        } while (!stop);
        return found ? s : 0xFFFFFFFFu;
    }
}

// ---- record decoding ----
template <int N>
CDBG_DEV uint64_t sel_word(const uint64_t (&r)[N], int idx) {      // r[idx] without dynamic register indexing
    uint64_t w = r[0];
    CDBG_PIN64(w);
#pragma unroll
    for (int j = 1; j < N; ++j) { uint64_t e = r[j]; CDBG_PIN64(e); w = (idx == j) ? e : w; }
    return idx < N ? w : 0ULL;
}
template <int W>
struct RecView {
    uint64_t r[RecFmt<W>::RW];
    // member k-mer t (bases [t, t+k)) as a number: one funnel shift of the RW-word record
    CDBG_DEV Kmer<W> kmer(int t, int k) const {
        if (W == 1) {                                    // 128-bit record r[1]:r[0], k-mer = bits [sh, sh + 2k), sh >= 16
            const int sh1 = 128 - 2 * (t + k);
            const uint64_t lo = sh1 >= 64 ? r[1] : r[0], hi = sh1 >= 64 ? 0ULL : r[1];
            const int b1 = sh1 & 63;
            Kmer<W> x1;
            x1.w[0] = ((lo >> b1) | (b1 ? (hi << (64 - b1)) : 0ULL)) & (~0ULL >> (64 - 2 * k));
            return x1;
        }
        const int sh = 64 * RecFmt<W>::RW - 2 * (t + k);
        const int ws = sh >> 6, bs = sh & 63;
        Kmer<W> x;
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const uint64_t lo = sel_word<RecFmt<W>::RW>(r, i + ws), hi = sel_word<RecFmt<W>::RW>(r, i + ws + 1);
            x.w[i] = bs ? ((lo >> bs) | (hi << (64 - bs))) : lo;
        }
        x.mask(k);
        return x;
    }
    CDBG_DEV int n() const { return (int)(r[0] & 0xFFu); }
    CDBG_DEV uint32_t sub() const { return (uint32_t)(r[0] >> 12) & 15u; }   // sub-partition of the record's minimizer (sub_of, kmer.h)
    CDBG_DEV bool first_trav() const { return (r[0] >> 8) & 1u; }
    CDBG_DEV bool last_trav() const { return (r[0] >> 9) & 1u; }
    CDBG_DEV bool first_foreign() const { return (r[0] >> 10) & 1u; }
    CDBG_DEV bool last_foreign() const { return (r[0] >> 11) & 1u; }
    CDBG_DEV uint32_t base(int i) const {
        const int pos = 64 * RecFmt<W>::RW - 2 * (i + 1);
        const int wi = pos >> 6;
        uint64_t w = r[0];                               // select chain: keeps r[] in registers (no scratch)
        CDBG_PIN64(w);
#pragma unroll
        for (int j = 1; j < RecFmt<W>::RW; ++j) { uint64_t e = r[j]; CDBG_PIN64(e); w = (wi == j) ? e : w; }
        return (uint32_t)(w >> (pos & 63)) & 3u;
    }
};

struct CountParams {
    const uint64_t* records;       // RW words per record
    const uint64_t* part_off;      // [n_parts + 1] record offsets (exact two-pass layout); part_pairs: [2 n_parts] begin / end of every partition's records
    uint32_t part_pairs;           //       (regions of estimated size filled in one pass: a region has slack behind its records; a spilled partition: begin == end)
    uint32_t part_stride;          // != 0: capped single-pass layout, partition p = records [p*stride, p*stride + fill[p])
    const uint32_t* part_fill;     //       records offered to p; fill > stride => spilled, handled by the repair launch
    const uint64_t* item_off;      // != null: work item i = records [item_off[i], item_off[i+1]) (repair launch)
    const uint32_t* part_list;     // optional: partitions to process (big pass); else blockIdx.x
    int k; uint32_t amin;
    // outputs
    uint64_t* solid_keys;          // W words per solid entry
    uint32_t* solid_cnt;           // count | TRAV_FLAG
    uint64_t solid_cap;            // entries allocated
    uint64_t* solid_cursor;
    uint64_t* seg_off; uint32_t* seg_n;   // per partition
    uint64_t* stats;               // [0] distinct home k-mers [1] home occurrences [2] solid home [3] solid travellers
    uint32_t* big_list; uint32_t* big_count;   // partitions that overflowed LDS
    uint32_t* error;               // set to 1 on output overflow
    // HBM scratch table (GLOBAL variant): slot i of the big pass uses [big_off[i], big_off[i+1]) slots
    uint64_t* g_keys; uint32_t* g_cnt; const uint64_t* big_off;
    uint32_t n_items;              // partitions (or part_list entries) to process
    uint32_t item_base;            // one-pass tier without a list: work item i is partition item_base + i (the tier runs once per SLICE of the partition space
                                   // while the records of the next slice are still being placed: host_count.h, deferred placement)
    uint32_t max_passes;           // LDS multi-pass limit before a partition is deferred to the HBM pass
    uint32_t max_sub;              // multi-pass kernel: passes that divide a partition by its records' sub-partition (a power of two <= 16; 0: COUNT_MAX_SUB)
};

// ---------------------------------------------------------------------------
// One partition, processed by the whole workgroup.
//
// Insert phase, member-parallel: each wave loads a batch of 64 records (one per lane) and
// builds a prefix sum of their member counts with wave shuffles; then every lane takes ONE
// member k-mer per step: it finds the owning record by a 6-step binary search over the
// lanes' prefix sums (ds_bpermute), fetches that record from its lane (ds_bpermute) and
// extracts the k-mer with a funnel shift, so all 64 lanes insert on every step regardless
// of how many k-mers each record holds.
//
// npass == 1: build the table once, sweep for statistics + solid count, reserve the
// output segment, sweep again to write it.
// npass  > 1 (the distinct k-mers do not fit the LDS table): the partition's records are
// streamed npass times and pass j only inserts the k-mers whose hash selects j; phase 0
// counts, then the segment is reserved, phase 1 repeats the passes and writes.  The records
// of one partition are a few hundred KB, so the re-reads are served by L2/MALL.
// `start_np` / `strikes` adapt the starting number of passes per workgroup.
// ---------------------------------------------------------------------------
template <int W, int TS, int NT, bool GLOBAL>
CDBG_DEV void count_partition(const CountParams& P, const uint32_t item, uint64_t (&acc)[4],
                              uint32_t& start_np, uint32_t& strikes, bool& clean, uint64_t& chunk_base, uint32_t& chunk_left,
                              uint64_t (&ph)[8], uint64_t& t_prev) {
    constexpr int RW = RecFmt<W>::RW;
    constexpr int NW = NT / 64;
    CDBG_SHARED uint64_t l_keys[GLOBAL ? 1 : TS * W];
    CDBG_SHARED uint32_t l_cnt[GLOBAL ? 1 : TS];
    // slots that received a new key, in insertion order: the single-pass sweep visits (and resets) only these
    // instead of all TS slots, most of which are empty at the usual ~20-45 % load
    constexpr uint32_t LIST_CAP = GLOBAL ? 1 : TS / 2;
    CDBG_SHARED uint16_t l_used[LIST_CAP];
    CDBG_SHARED uint32_t s_fill, s_over, s_nsolid, s_wr, s_members;
    CDBG_SHARED uint64_t s_base;
    CDBG_SHARED uint32_t s_stat[4];
    CDBG_SHARED uint64_t s_occ;                          // home occurrences of the partition (64 bit: one k-mer may hold 2^31 - 1 of them)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t p = P.part_list ? P.part_list[item] : item;
    uint64_t rec0, rec1;
    if (P.item_off) { rec0 = P.item_off[item]; rec1 = P.item_off[item + 1]; }
    else if (P.part_stride) {
        const uint32_t f = P.part_fill[p];
        if (f > P.part_stride) return;                           // spilled: counted by the repair launch
        rec0 = (uint64_t)p * P.part_stride; rec1 = rec0 + f;
    } else if (P.part_pairs) { rec0 = P.part_off[2ull * p]; rec1 = P.part_off[2ull * p + 1]; }
    else { rec0 = P.part_off[p]; rec1 = P.part_off[p + 1]; }
    CDBG_PH(0);
    if (rec1 == rec0) { if (tid == 0) { P.seg_off[p] = 0; P.seg_n[p] = 0; } return; }

    KTable<W> T; uint32_t* cnt; uint32_t cap;
    if (GLOBAL) {
        const uint64_t o0 = P.big_off[item]; cap = (uint32_t)(P.big_off[item + 1] - o0);
        T.keys = P.g_keys + o0 * W; cnt = P.g_cnt + o0;
    } else {
        cap = TS; T.keys = l_keys; cnt = l_cnt;
    }
    T.mask = cap - 1;
    const uint32_t maxfill = cap - cap / 4;                       // load limit; inserts also give up after 64 probes
    const int k = P.k;

    uint32_t npass = GLOBAL ? 1u : start_np;
    // Multi-pass partitions of moderate size run ONE phase: the segment is reserved up front for members / amin
    // entries (an upper bound of the solid entries) from the workgroup's chunk, every pass writes its solid entries
    // straight behind those of the passes before it, and the unused tail goes back to the chunk.  Only partitions
    // whose bound exceeds a chunk count first and write in a second round of passes.
    bool have_ub = false, onephase = false, reserved = false; uint32_t ub = 0;
    for (;;) {                                                    // attempts with npass, 2 npass, ...
        if (tid == 0) { s_nsolid = 0; s_wr = 0; s_over = 0; }
        if (tid < 4) s_stat[tid] = 0;
        if (tid == 4) s_occ = 0;
        if (!GLOBAL && npass > 1 && !have_ub) {                   // member k-mers of the partition (first byte of every record)
            if (tid == 0) s_members = 0;
            block_sync<GLOBAL>();
            uint32_t mine = 0;
            for (uint64_t i = rec0 + tid; i < rec1; i += NT) mine += (uint32_t)(P.records[i * RW] & 0xFFu);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
            if (lane == 0 && mine) atomic_add_u32(&s_members, mine);
            block_sync<GLOBAL>();
            ub = s_members / (P.amin ? P.amin : 1u);
            onephase = ub <= COUNT_CHUNK; have_ub = true;
        }
        if (onephase && npass > 1 && !reserved) {
            if (tid == 0) {
                if (ub > chunk_left) { chunk_base = atomic_add_u64(P.solid_cursor, (uint64_t)COUNT_CHUNK); chunk_left = COUNT_CHUNK; }
                uint64_t b = chunk_base; chunk_base += ub; chunk_left -= ub;
                if (b + ub > P.solid_cap) { *P.error = 1; b = 0; s_over = 2; }
                s_base = b;
            }
            reserved = true;
        }
        bool overflow = false;
        const uint32_t max_sub = P.max_sub ? P.max_sub : COUNT_MAX_SUB;
        const uint32_t nsub = npass < max_sub ? npass : max_sub, nhash = npass / nsub;   // passes = sub-partitions x hash classes
        const int nphase = (npass == 1 || onephase) ? 1 : 2;
        for (int phase = 0; phase < nphase && !overflow; ++phase) {
            for (uint32_t pass = 0; pass < npass; ++pass) {
                // ---- build the table of this pass ----
                if (tid == 0) s_fill = 0;
                if (!clean) {                                     // (the list sweep of the previous partition left it clean)
                    ktable_clear<W>(T, tid, NT);
                    for (uint32_t i = tid; i < cap; i += NT) cnt[i] = 0;
                }
                clean = false;
                block_sync<GLOBAL>();
                CDBG_PH(1);
                // the partition's records are split evenly over the waves; a wave takes its share 64 records at a time
                const uint64_t per_wave = (rec1 - rec0 + NW - 1) / NW;
                const uint64_t w0 = rec0 + (uint64_t)wave * per_wave, w1 = (w0 + per_wave < rec1) ? w0 + per_wave : rec1;
                for (uint64_t b0 = w0; b0 < w1; b0 += 64) {      // wave-uniform
                    if (__any((int)ld_volatile_u32(&s_over))) break;
                    const int nrec = (int)((w1 - b0) < 64 ? (w1 - b0) : 64);
                    RecView<W> R; int n = 0;
#pragma unroll
                    for (int i = 0; i < RW; ++i) R.r[i] = 0;
                    if (lane < nrec) {
#pragma unroll
                        for (int i = 0; i < RW; ++i) R.r[i] = P.records[(b0 + lane) * RW + i];
                        n = R.n();
                        // multi-pass: a pass takes the RECORDS of its sub-partition(s) -- all occurrences of a k-mer in this bucket
                        // come with the same minimizer, hence the same sub-partition (record meta bits 12-15) -- so a member k-mer is
                        // extracted in one pass only; beyond four sub-partitions (a single hot minimizer locus cannot be split this
                        // way) the passes divide the k-mers of a sub-partition by their hash as before
                        if (npass > 1 && (R.sub() & (nsub - 1u)) != (pass & (nsub - 1u))) n = 0;
                    }
                    int incl = n;                                  // inclusive prefix sum over the wave
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
                    const int excl = incl - n;                     // lanes past the batch: excl == total
                    const int total = __shfl(incl, 63);
                    for (int g0 = 0; g0 < total; g0 += 64) {       // wave-uniform trip count
                        const int g = g0 + lane;
                        const bool active = g < total;
                        // owning record of member g: the last lane whose first member is <= g (binary search over
                        // the lanes' prefix sums with ds_bpermute; no member->record map, so a batch is 64 records)
                        int ri = 0, ex = 0;
#pragma unroll
                        for (int step = 32; step >= 1; step >>= 1) {
                            const int cand = ri + step;
                            const int e = __shfl(excl, cand & 63);
                            if (e <= g) { ri = cand; ex = e; }     // ri stays a multiple of 2*step, so cand <= 63
                        }
                        RecView<W> Q;
#pragma unroll
                        for (int i = 0; i < RW; ++i) Q.r[i] = __shfl(R.r[i], ri);
                        if (active) {
                            const int t = g - ex, qn = Q.n();
                            const Kmer<W> fw = Q.kmer(t, k);
                            const Kmer<W> rc = fw.rc(k);
                            const bool rev = rc < fw;
                            const Kmer<W>& can = rev ? rc : fw;
                            if (nhash == 1 || ((can.hash() >> 20) & (nhash - 1)) == pass / nsub) {
                                bool is_new;
                                Kmer<W> ck = can;                     // (the key carries the foreign-junction flags: KEY_FOREIGN_*)
                                ck.w[W - 1] |= key_flags(t == 0 && Q.first_foreign(), t == qn - 1 && Q.last_foreign(), rev);
                                const uint32_t s = ktable_insert<W, GLOBAL>(T, ck, is_new, 64u);
                                if (s == 0xFFFFFFFFu) s_over = 1;   // table (nearly) full: this pass is void
                                else {
                                    if (is_new) {
                                        const uint32_t fi = atomic_add_u32(&s_fill, 1u);
                                        if (fi >= maxfill) s_over = 1;
                                        if (!GLOBAL && fi < LIST_CAP) l_used[fi] = (uint16_t)s;
                                    }
                                    const bool trav = (t == 0 && Q.first_trav()) || (t == qn - 1 && Q.last_trav());
                                    count_add_sat(&cnt[s]);
                                    if (trav) atomic_or_u32(&cnt[s], TRAV_FLAG);
                                }
                            }
                        }
                    }
                }
                block_sync<GLOBAL>();
                CDBG_PH(2);
                if (s_over) { overflow = true; break; }
                // ---- single pass: ONE sweep.  The segment is reserved for the table's fill (an upper bound
                // of the solid entries), solid entries are written compactly, the unused tail goes back to the
                // workgroup's chunk.
                if (npass == 1) {
                    if (GLOBAL) {                                  // an HBM table's fill can exceed the solid capacity
                        uint32_t my_solid = 0;                    // (sized for members / amin): count the solid entries first
                        for (uint32_t s = tid; s < cap; s += NT)
                            if (ktable_used<W>(T, s) && count_value(cnt[s]) >= P.amin) ++my_solid;
#pragma unroll
                        for (int d = 32; d >= 1; d >>= 1) my_solid += __shfl_xor(my_solid, d);
                        if (lane == 0 && my_solid) atomic_add_u32(&s_nsolid, my_solid);
                        block_sync<GLOBAL>();
                    }
                    if (tid == 0) {
                        const uint32_t need = GLOBAL ? s_nsolid : s_fill;
                        uint64_t b;
                        if (need > COUNT_CHUNK) b = atomic_add_u64(P.solid_cursor, (uint64_t)need);
                        else {
                            if (need > chunk_left) { chunk_base = atomic_add_u64(P.solid_cursor, (uint64_t)COUNT_CHUNK); chunk_left = COUNT_CHUNK; }
                            b = chunk_base; chunk_base += need; chunk_left -= need;
                        }
                        if (b + need > P.solid_cap) { *P.error = 1; b = ~0ull; }   // (not through s_over: slower waves may still be reading it)
                        s_base = b;
                    }
                    block_sync<GLOBAL>();
                    const bool wr_ok = s_base != ~0ull; const uint64_t obase = wr_ok ? s_base : 0;
                    uint32_t st_dist = 0, st_sh = 0, st_st = 0; uint64_t st_occ = 0;
                    const uint32_t nfill = s_fill;
                    const bool by_list = !GLOBAL && nfill <= LIST_CAP;
                    for (uint32_t i = tid; i < (by_list ? nfill : cap); i += NT) {
                        const uint32_t s = by_list ? (uint32_t)l_used[i] : i;
                        if (!by_list && !ktable_used<W>(T, s)) continue;
                        const uint32_t c = count_word_out(cnt[s]), n = c & ~TRAV_FLAG; const bool trav = c & TRAV_FLAG;
                        if (!trav) { ++st_dist; st_occ += n; }
                        if (n >= P.amin) {
                            if (trav) ++st_st; else ++st_sh;
                            if (wr_ok) {
                                const uint64_t o = obase + atomic_add_u32(&s_wr, 1u);
                                for (int i2 = 0; i2 < W; ++i2) P.solid_keys[o * W + i2] = T.keys[(uint64_t)s * W + i2];
                                P.solid_cnt[o] = c;
                            }
                        }
                        if (by_list) {                             // hand the slot back empty
                            T.keys[(uint64_t)s * W + (W - 1)] = KEY_EMPTY;
                            cnt[s] = 0;
                        }
                    }
                    clean = by_list;
                    // (full 32-bit sums: a giant partition -- one bucket for the whole input -- overflows 16-bit fields)
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) { st_dist += __shfl_xor(st_dist, d); st_sh += __shfl_xor(st_sh, d); st_st += __shfl_xor(st_st, d); }
                    st_occ = wave_sum_u64(st_occ);
                    if (lane == 0) {
                        if (st_dist) atomic_add_u32(&s_stat[0], st_dist);
                        if (st_occ) atomic_add_u64(&s_occ, st_occ);
                        if (st_sh) atomic_add_u32(&s_stat[2], st_sh);
                        if (st_st) atomic_add_u32(&s_stat[3], st_st);
                    }
                    block_sync<GLOBAL>();
                    if (tid == 0) {
                        const uint32_t used = s_wr, need = GLOBAL ? s_nsolid : s_fill;
                        P.seg_off[p] = obase; P.seg_n[p] = used;
                        if (need <= COUNT_CHUNK && wr_ok) { chunk_base -= (need - used); chunk_left += (need - used); }   // return the tail
                    }
                    CDBG_PH(3);
                    block_sync<GLOBAL>();
                    CDBG_PH(4);
                    continue;
                }
                // ---- multi-pass, one phase: statistics and solid entries of this pass in one sweep ----
                if (onephase) {
                    const uint64_t obase = s_base; const bool wr_ok = s_over == 0;
                    uint32_t st_dist = 0, st_sh = 0, st_st = 0; uint64_t st_occ = 0;
                    for (uint32_t s = tid; s < cap; s += NT) {
                        if (!ktable_used<W>(T, s)) continue;
                        const uint32_t c = count_word_out(cnt[s]), n = c & ~TRAV_FLAG; const bool trav = c & TRAV_FLAG;
                        if (!trav) { ++st_dist; st_occ += n; }
                        if (n >= P.amin) {
                            if (trav) ++st_st; else ++st_sh;
                            if (wr_ok) {
                                const uint64_t o = obase + atomic_add_u32(&s_wr, 1u);
                                for (int i = 0; i < W; ++i) P.solid_keys[o * W + i] = T.keys[(uint64_t)s * W + i];
                                P.solid_cnt[o] = c;
                            }
                        }
                    }
                    // (full 32-bit sums: a giant partition -- one bucket for the whole input -- overflows 16-bit fields)
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) { st_dist += __shfl_xor(st_dist, d); st_sh += __shfl_xor(st_sh, d); st_st += __shfl_xor(st_st, d); }
                    st_occ = wave_sum_u64(st_occ);
                    if (lane == 0) {
                        if (st_dist) atomic_add_u32(&s_stat[0], st_dist);
                        if (st_occ) atomic_add_u64(&s_occ, st_occ);
                        if (st_sh) atomic_add_u32(&s_stat[2], st_sh);
                        if (st_st) atomic_add_u32(&s_stat[3], st_st);
                    }
                    block_sync<GLOBAL>();
                    CDBG_PH(3);
                    continue;
                }
                // ---- sweep: statistics + number of solid entries (phase 0) ----
                if (phase == 0) {
                    uint32_t my_solid = 0, st_dist = 0, st_sh = 0, st_st = 0; uint64_t st_occ = 0;
                    for (uint32_t s = tid; s < cap; s += NT) {
                        if (!ktable_used<W>(T, s)) continue;
                        const uint32_t c = count_word_out(cnt[s]), n = c & ~TRAV_FLAG; const bool trav = c & TRAV_FLAG;
                        if (!trav) { ++st_dist; st_occ += n; }
                        if (n >= P.amin) { ++my_solid; if (trav) ++st_st; else ++st_sh; }
                    }
                    // wave-level tree reduction first (an LDS atomic with per-lane values is serialised
                    // lane by lane by the compiler), then one LDS atomic per wave and counter
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) {
                        my_solid += __shfl_xor(my_solid, d); st_dist += __shfl_xor(st_dist, d);
                        st_sh += __shfl_xor(st_sh, d); st_st += __shfl_xor(st_st, d);
                    }
                    st_occ = wave_sum_u64(st_occ);
                    if (lane == 0) {
                        if (my_solid) atomic_add_u32(&s_nsolid, my_solid);
                        if (st_dist) atomic_add_u32(&s_stat[0], st_dist);
                        if (st_occ) atomic_add_u64(&s_occ, st_occ);
                        if (st_sh) atomic_add_u32(&s_stat[2], st_sh);
                        if (st_st) atomic_add_u32(&s_stat[3], st_st);
                    }
                    block_sync<GLOBAL>();
                    if (pass == npass - 1 && tid == 0) {          // everything counted: reserve the segment
                        // sub-allocate from this workgroup's chunk: one device atomic per COUNT_CHUNK entries
                        uint64_t b;
                        if (s_nsolid > COUNT_CHUNK) b = atomic_add_u64(P.solid_cursor, (uint64_t)s_nsolid);
                        else {
                            if (s_nsolid > chunk_left) { chunk_base = atomic_add_u64(P.solid_cursor, (uint64_t)COUNT_CHUNK); chunk_left = COUNT_CHUNK; }
                            b = chunk_base; chunk_base += s_nsolid; chunk_left -= s_nsolid;
                        }
                        if (b + s_nsolid > P.solid_cap) { *P.error = 1; b = 0; s_nsolid = 0; }
                        s_base = b;
                        P.seg_off[p] = b; P.seg_n[p] = s_nsolid;
                    }
                    if (pass == npass - 1) block_sync<GLOBAL>();
                    CDBG_PH(3);
                }
                // ---- sweep: write solid entries (multi-pass: phase 1) ----
                if (phase == 1) {
                    const uint64_t obase = s_base;
                    if (s_nsolid) {
                        for (uint32_t s = tid; s < cap; s += NT) {
                            if (!ktable_used<W>(T, s)) continue;
                            const uint32_t c = count_word_out(cnt[s]);
                            if ((c & ~TRAV_FLAG) < P.amin) continue;
                            const uint64_t o = obase + atomic_add_u32(&s_wr, 1u);
                            for (int i = 0; i < W; ++i) P.solid_keys[o * W + i] = T.keys[(uint64_t)s * W + i];
                            P.solid_cnt[o] = c;
                        }
                    }
                }
                block_sync<GLOBAL>();
                CDBG_PH(4);
            }
        }
        if (!overflow) {
            if (onephase && npass > 1 && tid == 0) {              // (the last pass ended with a barrier: s_wr is final)
                const uint32_t used = s_wr;
                P.seg_off[p] = s_base; P.seg_n[p] = s_over ? 0u : used;
                chunk_base -= (ub - used); chunk_left += (ub - used);   // the reservation is the newest of this chunk: return its tail
            }
            break;
        }
        block_sync<GLOBAL>();
        if (GLOBAL) { if (tid == 0) *P.error = 2; return; }     // scratch sizing bug: cannot happen by construction
        npass *= 2;
        if (npass > P.max_passes) {                               // hopeless in LDS: defer to the HBM pass
            if (tid == 0) {
                if (reserved) { chunk_base -= ub; chunk_left += ub; }   // hand the whole reservation back
                // (a partition of the repair launch has no region the HBM pass could read: report instead of dropping it)
                if (P.item_off) *P.error = 7;
                const uint32_t i = atomic_add_u32(P.big_count, 1u); P.big_list[i] = p; P.seg_off[p] = 0; P.seg_n[p] = 0;
            }
            return;
        }
    }
    if (tid == 0) { for (int i = 0; i < 4; ++i) acc[i] += (uint64_t)s_stat[i]; acc[1] += s_occ; }
    if (!GLOBAL) {                                                // adapt the starting pass count of this workgroup
        if (npass > start_np) { if (++strikes >= 2) { start_np = npass; strikes = 0; } }
        else if (strikes) --strikes;
    }
}

// persistent workgroups, grid-stride over partitions (HIP limits grid*block to < 2^32 work-items);
// statistics are accumulated in registers and published with one atomic per workgroup
// waves per SIMD to promise the register allocator: the target, or what the workgroup's LDS leaves room for (160 KB per CU)
constexpr int lds_waves_per_simd(size_t lds_bytes, int nt, int target) {
    const int wgs = (int)(163840 / (lds_bytes ? lds_bytes : 1)), w = wgs * (nt / 64) / 4;
    return w < 1 ? 1 : (w < target ? w : target);
}
template <int W, int TS, int NT, bool GLOBAL>
__global__ void __launch_bounds__(NT, lds_waves_per_simd(GLOBAL ? 1024 : (size_t)TS * (8 * W + 4 + 1) + 1024, NT, (W == 1 && !GLOBAL) ? 6 : 4)) k_count(CountParams P) {   // waves per SIMD that the LDS tables allow: 3 workgroups x 2 waves (W = 1), 4 otherwise
    uint64_t acc[4] = {0, 0, 0, 0};
    uint32_t start_np = 1, strikes = 0; bool clean = false;
    uint64_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t t_prev = 0;
    uint64_t chunk_base = 0; uint32_t chunk_left = 0;
#if defined(CDBG_PROFILE_PHASES) && !defined(CDBG_HOSTSIM)
    t_prev = wall_clock64();
#endif
    for (uint32_t item = blockIdx.x; item < P.n_items; item += gridDim.x) {
        count_partition<W, TS, NT, GLOBAL>(P, item, acc, start_np, strikes, clean, chunk_base, chunk_left, ph, t_prev);
        block_sync<GLOBAL>();                            // LDS is reused by the next partition
        CDBG_PH(5);
    }
#if defined(CDBG_PROFILE_PHASES) && !defined(CDBG_HOSTSIM)
    if (threadIdx.x == 0) for (int i = 0; i < 8; ++i) if (ph[i]) atomic_add_u64(&P.stats[8 + i], ph[i]);
#endif
    if (threadIdx.x == 0) for (int i = 0; i < 4; ++i) if (acc[i]) atomic_add_u64(&P.stats[i], acc[i]);
}

// member k-mers of every LISTED partition (the HBM-table pass sizes its tables from them: round 5 -- records x the largest record a format
// allows over-sized them 3 - 5 x, and one workgroup clears and sweeps a table: at k = 127 a partition of a repeat's locus had 600 MB of it)
struct BigMembersParams { CountParams c; int RW; uint64_t* members; uint32_t* nrec; };   // RW: words per record; nrec: records of every listed partition (the grid-wide pass below)
// records [rec0, rec1) of partition p in the layout the stage runs with (capped regions, begin / end pairs, exact offsets)
CDBG_DEV void count_part_range(const CountParams& P, uint32_t p, uint64_t& rec0, uint64_t& rec1) {
    if (P.part_stride) { const uint32_t f = P.part_fill[p]; rec0 = (uint64_t)p * P.part_stride; rec1 = rec0 + (f > P.part_stride ? 0u : f); }
    else if (P.part_pairs) { rec0 = P.part_off[2ull * p]; rec1 = P.part_off[2ull * p + 1]; }
    else { rec0 = P.part_off[p]; rec1 = P.part_off[p + 1]; }
}
__global__ void k_big_members(BigMembersParams B) {
    const CountParams& P = B.c;
    const uint32_t item = blockIdx.x;
    const uint32_t p = P.part_list[item];
    uint64_t rec0, rec1; count_part_range(P, p, rec0, rec1);
    if (threadIdx.x == 0) B.nrec[item] = (uint32_t)(rec1 - rec0);
    uint64_t mine = 0;
    for (uint64_t i = rec0 + threadIdx.x; i < rec1; i += blockDim.x) mine += P.records[i * B.RW] & 0xFFu;
    mine = wave_sum_u64(mine);
    if ((threadIdx.x & 63) == 0 && mine) atomic_add_u64(&B.members[item], mine);
}

// ---------------------------------------------------------------------------
// HBM-table pass, GRID-WIDE (round 6).  The partitions that no LDS tier takes -- a repeat's locus at long k brings 10^5 - 10^6 distinct
// k-mers into one partition -- used to get ONE 256-thread workgroup each (count_partition<.., GLOBAL>): the largest partition was the
// stage's tail, and sending more partitions here (fewer LDS passes) made it worse (hostile k = 127 share: 2.0 s at 16 LDS passes, 8.4 s
// at 4; profiles/r05_hostile_long_k.log).  Now the listed partitions' records are ONE list that all workgroups share 64 records at a time
// (k_big_insert: a lane finds its record's partition in the prefix of the record counts), the tables are cleared and swept slot by slot
// by the whole grid (k_big_clear / k_big_sweep: a wave's 64 slots lie in one table -- every table is a power of two >= 1024), and a
// partition's segment of the solid arrays is reserved between the two sweeps from the exact number of its solid entries.
// ---------------------------------------------------------------------------
struct BigGridParams {
    CountParams c;                 // part_list: the listed partitions; big_off / g_keys / g_cnt: their tables; n_items
    int RW;
    const uint64_t* rec_pref;      // [n_items + 1] exclusive prefix of the partitions' record counts
    uint32_t* nsolid;              // [n_items] solid entries of every partition (sweep 0; zeroed by the host)
    const uint64_t* seg_pref;      // [n_items + 1] exclusive prefix of nsolid (sweep 1)
    uint64_t seg_base;             // first entry of the listed partitions' segments in the solid arrays
    uint32_t* wr;                  // [n_items] write cursors (sweep 1; zeroed by the host)
};
// largest i < n with pref[i] <= x (pref[0] = 0 <= x)
CDBG_DEV uint32_t big_find(const uint64_t* pref, uint32_t n, uint64_t x) {
    uint32_t lo = 0, hi = n;                               // invariant: pref[lo] <= x < pref[hi]
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (pref[mid] <= x) lo = mid; else hi = mid; }
    return lo;
}
template <int W>
__global__ void __launch_bounds__(256) k_big_clear(BigGridParams B) {
    const CountParams& P = B.c;
    const uint64_t total = P.big_off[P.n_items], stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += stride) { P.g_keys[s * W + (W - 1)] = KEY_EMPTY; P.g_cnt[s] = 0; }
}
template <int W>
__global__ void __launch_bounds__(256) k_big_insert(BigGridParams B) {
    constexpr int RW = RecFmt<W>::RW;
    const CountParams& P = B.c;
    const int lane = threadIdx.x & 63;
    const uint64_t total = B.rec_pref[P.n_items];
    const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6), gw = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int k = P.k;
    for (uint64_t b0 = gw * 64; b0 < total; b0 += nwaves * 64) {   // wave-uniform
        RecView<W> R; int n = 0; uint32_t item = 0;
#pragma unroll
        for (int i = 0; i < RW; ++i) R.r[i] = 0;
        if (b0 + (uint64_t)lane < total) {
            const uint64_t r = b0 + (uint64_t)lane;
            item = big_find(B.rec_pref, P.n_items, r);
            uint64_t rec0, rec1; count_part_range(P, P.part_list[item], rec0, rec1);
            const uint64_t at = rec0 + (r - B.rec_pref[item]);
#pragma unroll
            for (int i = 0; i < RW; ++i) R.r[i] = P.records[at * RW + i];
            n = R.n();
        }
        int incl = n;                                      // inclusive prefix sum over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
        const int excl = incl - n;
        const int total_m = __shfl(incl, 63);
        for (int g0 = 0; g0 < total_m; g0 += 64) {         // wave-uniform trip count: one member k-mer per lane and step (count_partition's scheme)
            const int g = g0 + lane;
            const bool active = g < total_m;
            int ri = 0, ex = 0;
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                const int cand = ri + step;
                const int e = __shfl(excl, cand & 63);
                if (e <= g) { ri = cand; ex = e; }
            }
            RecView<W> Q;
#pragma unroll
            for (int i = 0; i < RW; ++i) Q.r[i] = __shfl(R.r[i], ri);
            const uint32_t it = __shfl(item, ri);
            if (active) {
                const uint64_t o0 = P.big_off[it];
                KTable<W> T; T.keys = P.g_keys + o0 * W; T.mask = (uint32_t)(P.big_off[it + 1] - o0) - 1u;
                uint32_t* const cnt = P.g_cnt + o0;
                const int t = g - ex, qn = Q.n();
                const Kmer<W> fw = Q.kmer(t, k);
                const Kmer<W> rc = fw.rc(k);
                const bool rev = rc < fw;
                Kmer<W> ck = rev ? rc : fw;
                ck.w[W - 1] |= key_flags(t == 0 && Q.first_foreign(), t == qn - 1 && Q.last_foreign(), rev);
                bool is_new;
                const uint32_t s = ktable_insert<W, true>(T, ck, is_new, T.mask);   // (tables hold twice the partition's member k-mers: never full)
                if (s == 0xFFFFFFFFu) *P.error = 2;
                else {
                    const bool trav = (t == 0 && Q.first_trav()) || (t == qn - 1 && Q.last_trav());
                    count_add_sat(&cnt[s]);
                    if (trav) atomic_or_u32(&cnt[s], TRAV_FLAG);
                }
            }
        }
    }
}
// PHASE 0: statistics and the number of solid entries of every partition; PHASE 1: the solid entries into the partitions' segments
template <int W, int PHASE>
__global__ void __launch_bounds__(256) k_big_sweep(BigGridParams B) {
    const CountParams& P = B.c;
    const int lane = threadIdx.x & 63;
    const uint64_t total = P.big_off[P.n_items];
    const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6), gw = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint64_t st_dist = 0, st_occ = 0, st_sh = 0, st_st = 0;
    for (uint64_t s0 = gw * 64; s0 < total; s0 += nwaves * 64) {   // wave-uniform; the 64 slots lie in one table
        const uint32_t item = big_find(P.big_off, P.n_items, s0);
        const uint64_t s = s0 + (uint64_t)lane;
        const bool used = P.g_keys[s * W + (W - 1)] != KEY_EMPTY;
        const uint32_t c = count_word_out(P.g_cnt[s]), n = c & ~TRAV_FLAG; const bool trav = c & TRAV_FLAG;
        const bool solid = used && n >= P.amin;
        if (PHASE == 0 && used) {
            if (!trav) { ++st_dist; st_occ += n; }
            if (solid) { if (trav) ++st_st; else ++st_sh; }
        }
        const uint64_t m = __ballot(solid);                // (every lane of the wave is here)
        if (m == 0) continue;
        const uint32_t cntm = (uint32_t)__popcll(m);
        if (PHASE == 0) { if (lane == 0) atomic_add_u32(&B.nsolid[item], cntm); }
        else {
            uint32_t base = 0;
            if (lane == 0) base = atomic_add_u32(&B.wr[item], cntm);
            base = __shfl(base, 0);
            if (solid) {
                const uint64_t o = B.seg_base + B.seg_pref[item] + base + (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
                for (int i = 0; i < W; ++i) P.solid_keys[o * W + i] = P.g_keys[s * W + i];
                P.solid_cnt[o] = c;
            }
        }
    }
    if (PHASE == 0) {
        st_dist = wave_sum_u64(st_dist); st_occ = wave_sum_u64(st_occ); st_sh = wave_sum_u64(st_sh); st_st = wave_sum_u64(st_st);
        if (lane == 0) {
            if (st_dist) atomic_add_u64(&P.stats[0], st_dist);
            if (st_occ) atomic_add_u64(&P.stats[1], st_occ);
            if (st_sh) atomic_add_u64(&P.stats[2], st_sh);
            if (st_st) atomic_add_u64(&P.stats[3], st_st);
        }
    }
}
// the listed partitions' segments: [seg_base + seg_pref[i], + nsolid[i])
__global__ void k_big_segments(BigGridParams B) {
    const CountParams& P = B.c;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_items) return;
    const uint32_t p = P.part_list[i];
    P.seg_off[p] = B.seg_base + B.seg_pref[i]; P.seg_n[p] = B.nsolid[i];
}

// ---- capped-layout repair: gather a spilled partition's region + spill records contiguously ----
// Entirely on the device (the host only reads two totals): partitions whose fill exceeds the region capacity are
// flagged and compacted into a list (two prefix sums), their regions are copied into one array of back-to-back runs, and
// every spilled record finds its run through the partition -> list index map and takes the next free place in it with
// one device atomic.  (The first version read the spill list back, sorted it on the host and uploaded an order array:
// 66 ms per step at the config-5 share, where 1 % of the partitions hold two minimizer loci and overflow their region.)
struct RepairParams {
    const uint64_t* records; const uint64_t* spill_recs; const uint32_t* spill_part; uint64_t n_spill;
    const uint32_t* part_fill; uint64_t npl; uint32_t part_cap; int RW;
    const uint64_t* ovf;         // != null (skewed inputs): overflow-region words of the partitions (k_scan.h ScanParams::ovf); a partition with one holds
                                 // part_cap records in its uniform region and the rest of its capacity in slots [part_cap, capacity) of the overflow region
    uint32_t* flag;              // [npl]  1 = spilled
    const uint64_t* ridx;        // [npl + 1] exclusive scan of flag: list index of a spilled partition
    uint32_t* item_part;         // [nsp]  spilled partitions, ascending
    uint32_t* item_size;         // [nsp]  records of the gathered run (= fill)
    const uint64_t* item_off;    // [nsp + 1] exclusive scan of item_size
    uint32_t* item_fill;         // [nsp]  spilled records placed so far (zeroed)
    uint64_t* out;               // gathered runs
};
CDBG_DEV uint64_t repair_capacity(const RepairParams& P, uint64_t p) {       // records of partition p that did NOT go to the spill list, at most
    const uint64_t t = P.ovf ? (P.ovf[p] & OVF_CAP_MASK) : 0ULL;
    return t ? t : (uint64_t)P.part_cap;
}
__global__ void k_repair_flag(RepairParams P) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < P.npl) P.flag[p] = (uint64_t)P.part_fill[p] > repair_capacity(P, p) ? 1u : 0u;
}
__global__ void k_repair_list(RepairParams P) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.npl || !P.flag[p]) return;
    const uint64_t i = P.ridx[p];
    P.item_part[i] = (uint32_t)p; P.item_size[i] = P.part_fill[p];
}
__global__ void k_repair_gather(RepairParams P) {        // one workgroup per spilled partition: its region(s)
    const uint32_t it = blockIdx.x;
    const uint64_t o0 = P.item_off[it] * P.RW, p = P.item_part[it];
    const uint64_t nreg = (uint64_t)P.part_cap * P.RW;
    const uint64_t src = p * nreg;
    for (uint64_t i = threadIdx.x; i < nreg; i += blockDim.x) P.out[o0 + i] = P.records[src + i];
    const uint64_t cap = repair_capacity(P, p);
    if (cap > P.part_cap) {                              // the slots [part_cap, capacity) of its overflow region
        const uint64_t hb = (P.ovf[p] >> OVF_CAP_BITS) * P.RW, n2 = cap * P.RW;
        for (uint64_t i = nreg + threadIdx.x; i < n2; i += blockDim.x) P.out[o0 + i] = P.records[hb + i];
    }
}
__global__ void k_repair_scatter(RepairParams P) {       // one thread per spilled record
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; o < P.n_spill; o += stride) {
        const uint64_t sp = P.spill_part[o], it = P.ridx[sp];
        const uint64_t dst = (P.item_off[it] + repair_capacity(P, sp) + atomic_add_u32(&P.item_fill[it], 1u)) * P.RW;
        for (int w = 0; w < P.RW; ++w) P.out[dst + w] = P.spill_recs[o * P.RW + w];
    }
}


// ---- single-pass record layout for SKEWED inputs: the capped layout plus an OVERFLOW REGION for every partition the sample finds heavy ----
// (the uniform capacity of the capped layout cannot hold a coverage peak or a repeat; the exact layout costs a second pass over the
//  reads: 60 ms of histogram at config-3 size.  Round 4 gave every partition a region of its own estimated size -- begin and end looked
//  up per record: the scan pays per memory request of a record, 86 instead of 66 ms.  Now only the records BEYOND the uniform capacity
//  pay the look-up.)  k_ovf_caps: sampled count s -> total capacity scale * (s + 4 sqrt(s) + 2) of a heavy partition, 0 for the others;
//  the host scans the capacities into offsets behind the uniform regions; k_ovf_words builds the words the scan reads; k_ovf_finish
//  (after the scan) moves the uniform region's records of an overflowed partition to the front of its overflow region -- the partition
//  is one run again -- and writes begin / end of every partition's records for the count kernels (part_pairs; spilled: empty, the
//  repair launch counts them).
struct OvfParams {
    const uint32_t* sample; uint32_t* tcap; uint64_t n; float scale; float heavy_min; uint32_t part_cap;
    const uint64_t* toff; uint64_t area0; uint64_t* words;
    const uint32_t* fill; uint64_t* records; int RW; uint64_t* pairs; uint64_t* stats;   // stats[0] += records offered, stats[1] += partitions that used their overflow region
    uint64_t p0, p1;             // k_ovf_finish: the partitions [p0, p1) (deferred placement finishes a slice of the partition space when its stream has been placed)
};
__global__ void k_ovf_caps(OvfParams P) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.n) return;
    const float s = (float)P.sample[p];
    float est = P.scale * (s + 4.0f * sqrtf(s) + 2.0f) + 8.0f;
    if (est > (float)(OVF_CAP_MASK - 15ULL)) est = (float)(OVF_CAP_MASK - 15ULL);
    const uint32_t c = ((uint32_t)est + 7u) & ~7u;
    // heavy: sampled well above what a partition of the uniform part of the input shows (heavy_min: twice the mean sample + 2).  The margins of a
    // 1-in-64 sample are wide -- without this bar the typical partition (6 sampled records at config 3: up to 1139 with + 4 sigma) counted as heavy
    // and the region came to 2.4 x the uniform layout's (k = 55, 125 M reads: 189 GB, and the step ran out of memory).
    P.tcap[p] = (s >= P.heavy_min && c > P.part_cap) ? c : 0u;
}
__global__ void k_ovf_words(OvfParams P) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.n) return;
    const uint64_t t = P.tcap[p];
    P.words[p] = t ? (((P.area0 + P.toff[p]) << OVF_CAP_BITS) | t) : 0ULL;
}
__global__ void k_ovf_finish(OvfParams P) {              // one wave per partition, grid-stride
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint64_t offered = 0, used = 0;
    for (uint64_t p = P.p0 + wave; p < P.p1; p += n_waves) {
        const uint64_t f = P.fill[p], d = P.words[p], t = d & OVF_CAP_MASK, hb = d >> OVF_CAP_BITS, b = p * P.part_cap;
        offered += f;
        uint64_t r0 = b, r1 = b + f;
        if (f > P.part_cap) {
            if (f <= t) {
                const uint64_t n = (uint64_t)P.part_cap * P.RW;
                for (uint64_t i = lane; i < n; i += 64) P.records[hb * P.RW + i] = P.records[b * P.RW + i];
                r0 = hb; r1 = hb + f; ++used;
            } else r1 = b;                               // spilled: counted from its gathered copy
        }
        if (lane == 0) { P.pairs[2 * p] = r0; P.pairs[2 * p + 1] = r1; }
    }
    if (lane == 0) { if (offered) atomic_add_u64(&P.stats[0], offered); if (used) atomic_add_u64(&P.stats[1], used); }
}

// ---- multi-GPU, single-pass scan: squeeze the capped regions into the exact owner-major layout that travels ----
// (slot = owner * npl + local partition; off = exclusive scan of the fill counts; one wave per slot, grid-stride)
struct PackRegionParams { const uint64_t* regions; const uint32_t* fill; const uint64_t* off; uint64_t n_slots; uint32_t part_cap; int RW; uint64_t* out; };
__global__ void k_pack_regions(PackRegionParams P) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t sl = wave; sl < P.n_slots; sl += n_waves) {
        const uint64_t n = (uint64_t)(P.fill[sl] < P.part_cap ? P.fill[sl] : P.part_cap) * P.RW;   // (fill counts the records OFFERED: the rest is on the spill list)
        const uint64_t* src = P.regions + sl * P.part_cap * P.RW;
        uint64_t* dst = P.out + P.off[sl] * P.RW;
        for (uint64_t i = lane; i < n; i += 64) dst[i] = src[i];
    }
}

// the spilled records of overfull regions behind their region's records (off[] was scanned from the OFFERED counts, so the room is there);
// fill_extra: zeroed per-slot counters
struct PackSpillParams { const uint64_t* spill_recs; const uint32_t* spill_part; uint64_t n_spill; const uint64_t* off; uint64_t* fill_extra; uint32_t part_cap; int RW; uint64_t* out; };
__global__ void k_pack_spills(PackSpillParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; o < P.n_spill; o += stride) {
        const uint64_t sl = P.spill_part[o];
        const uint64_t dst = (P.off[sl] + P.part_cap + atomic_add_u64(&P.fill_extra[sl], 1ULL)) * P.RW;
        for (int w = 0; w < P.RW; ++w) P.out[dst + w] = P.spill_recs[o * P.RW + w];
    }
}

// ---- multi-GPU (reads sharded over the ranks): merge the record blocks received from every rank ----
// Sender s delivered, for each of this rank's partitions lp, xcnt[s][lp] records, sorted by lp, at record index
// xbase[s] + xoff[s][lp] of the receive buffer.  Partition lp of the merged array = its segments in sender order.
struct SumCountParams { const uint32_t* xcnt; int world; uint64_t npl; uint32_t* total; };
__global__ void k_sum_counts(SumCountParams P) {
    const uint64_t lp = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lp >= P.npl) return;
    uint32_t t = 0;
    for (int s = 0; s < P.world; ++s) t += P.xcnt[(uint64_t)s * P.npl + lp];
    P.total[lp] = t;
}
struct MergeRecParams {
    const uint32_t* xcnt; const uint64_t* xoff; const uint64_t* xbase; int world; uint64_t npl; int RW;
    const uint64_t* xrecs; const uint64_t* part_off; uint64_t* out; uint64_t* stats;   // stats[0] member k-mers, [1] traveller members
};
__global__ void k_merge_records(MergeRecParams P) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint64_t members = 0, trav = 0;
    for (uint64_t lp = wave; lp < P.npl; lp += n_waves) {
        uint64_t dst = P.part_off[lp] * P.RW;
        for (int s = 0; s < P.world; ++s) {
            const uint64_t n = (uint64_t)P.xcnt[(uint64_t)s * P.npl + lp] * P.RW;
            const uint64_t* src = P.xrecs + (P.xbase[s] + P.xoff[(uint64_t)s * (P.npl + 1) + lp]) * P.RW;
            for (uint64_t i = lane; i < n; i += 64) {
                const uint64_t w = src[i];
                P.out[dst + i] = w;
                if (i % P.RW == 0) { members += w & 0xFFu; trav += ((w >> 8) & 1u) + ((w >> 9) & 1u); }
            }
            dst += n;
        }
    }
    members = wave_sum_u64(members); trav = wave_sum_u64(trav);
    if (lane == 0) { if (members) atomic_add_u64(&P.stats[0], members); if (trav) atomic_add_u64(&P.stats[1], trav); }
}

}  // namespace cdbg
