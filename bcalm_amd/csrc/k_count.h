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
#include "k_scan.h"

namespace cdbg {

constexpr int COUNT_THREADS = 256;
constexpr uint32_t TRAV_FLAG = 0x80000000u;          // in a count word: this entry is a traveller copy
constexpr uint32_t ST_EMPTY = 0u, ST_BUSY = 1u;      // slot states for multi-word keys (W > 1)

// ---------------------------------------------------------------------------
// Open-address table of W-word keys.  W == 1: the key word itself is claimed with
// one 64-bit CAS (EMPTY = all ones, unreachable for k <= 31).  W > 1: a 32-bit state
// word per slot is claimed (EMPTY -> BUSY), the key words are written, then the state
// becomes the key's tag (bit 31 set); readers of a BUSY slot retry.
// ---------------------------------------------------------------------------
template <int W>
struct KTable {
    uint64_t* keys;      // [cap * W]
    uint32_t* state;     // [cap]   (W > 1 only)
    uint32_t mask;       // cap - 1
};

template <int W>
CDBG_DEV void ktable_clear(const KTable<W>& t, int tid, int nthreads) {
    const uint32_t cap = t.mask + 1;
    if (W == 1) { for (uint32_t i = tid; i < cap; i += nthreads) t.keys[i] = ~0ULL; }
    else { for (uint32_t i = tid; i < cap; i += nthreads) t.state[i] = ST_EMPTY; }
}
template <int W>
CDBG_DEV bool ktable_used(const KTable<W>& t, uint32_t s) {
    if (W == 1) return t.keys[s] != ~0ULL;
    return t.state[s] > ST_BUSY;
}
template <int W>
CDBG_DEV Kmer<W> ktable_key(const KTable<W>& t, uint32_t s) {
    Kmer<W> r;
    for (int i = 0; i < W; ++i) r.w[i] = t.keys[(uint64_t)s * W + i];
    return r;
}
// find-or-insert; returns slot, sets is_new.  GLOBAL selects the fence flavour.
template <int W, bool GLOBAL>
CDBG_DEV uint32_t ktable_insert(const KTable<W>& t, const Kmer<W>& key, bool& is_new) {
    const uint32_t h = key.hash();
    uint32_t s = h & t.mask;
    is_new = false;
    if (W == 1) {
        for (;;) {
            const uint64_t old = atomic_cas_u64(&t.keys[s], ~0ULL, key.w[0]);
            if (old == ~0ULL) { is_new = true; return s; }
            if (old == key.w[0]) return s;
            s = (s + 1) & t.mask;
        }
    } else {
        const uint32_t tag = (h >> 1) | 0x80000000u;
        for (;;) {
            const uint32_t st = atomic_cas_u32(&t.state[s], ST_EMPTY, ST_BUSY);
            if (st == ST_EMPTY) {
                for (int i = 0; i < W; ++i) t.keys[(uint64_t)s * W + i] = key.w[i];
                if (GLOBAL) __threadfence(); else __threadfence_block();
                atomicExch(&t.state[s], tag);
                is_new = true; return s;
            }
            if (st == ST_BUSY) { CDBG_SPIN_YIELD(); continue; }
            if (st == tag) {
                bool eq = true;
                for (int i = 0; i < W; ++i) eq &= ((GLOBAL ? ld_agent_u64(&t.keys[(uint64_t)s * W + i]) : t.keys[(uint64_t)s * W + i]) == key.w[i]);
                if (eq) return s;
            }
            s = (s + 1) & t.mask;
        }
    }
}
// lookup only (table no longer being modified); returns slot or 0xFFFFFFFF
template <int W>
CDBG_DEV uint32_t ktable_find(const KTable<W>& t, const Kmer<W>& key) {
    const uint32_t h = key.hash();
    uint32_t s = h & t.mask;
    if (W == 1) {
        for (;;) {
            const uint64_t v = t.keys[s];
            if (v == key.w[0]) return s;
            if (v == ~0ULL) return 0xFFFFFFFFu;
            s = (s + 1) & t.mask;
        }
    } else {
        const uint32_t tag = (h >> 1) | 0x80000000u;
        for (;;) {
            const uint32_t st = t.state[s];
            if (st == ST_EMPTY) return 0xFFFFFFFFu;
            if (st == tag) {
                bool eq = true;
                for (int i = 0; i < W; ++i) eq &= (t.keys[(uint64_t)s * W + i] == key.w[i]);
                if (eq) return s;
            }
            s = (s + 1) & t.mask;
        }
    }
}

// ---- record decoding ----
template <int W>
struct RecView {
    uint64_t r[RecFmt<W>::RW];
    CDBG_DEV int n() const { return (int)(r[0] & 0xFFu); }
    CDBG_DEV bool first_trav() const { return (r[0] >> 8) & 1u; }
    CDBG_DEV bool last_trav() const { return (r[0] >> 9) & 1u; }
    CDBG_DEV uint32_t base(int i) const {
        const int pos = 64 * RecFmt<W>::RW - 2 * (i + 1);
        return (uint32_t)(r[pos >> 6] >> (pos & 63)) & 3u;
    }
};

struct CountParams {
    const uint64_t* records;       // RW words per record
    const uint64_t* part_off;      // [n_parts + 1] record offsets
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
    uint64_t* g_keys; uint32_t* g_state; uint32_t* g_cnt; const uint64_t* big_off;
    uint32_t n_items;              // partitions (or part_list entries) to process
};

// one partition, processed by the whole workgroup; `item` = index of the work item (== partition
// unless a part_list is given)
template <int W, int TS, bool GLOBAL>
CDBG_DEV void count_partition(const CountParams& P, const uint32_t item) {
    constexpr int RW = RecFmt<W>::RW;
    CDBG_SHARED uint64_t l_keys[GLOBAL ? 1 : TS * W];
    CDBG_SHARED uint32_t l_state[(GLOBAL || W == 1) ? 1 : TS];
    CDBG_SHARED uint32_t l_cnt[GLOBAL ? 1 : TS];
    CDBG_SHARED uint32_t s_fill, s_over, s_nsolid, s_wr;
    CDBG_SHARED uint64_t s_base;
    CDBG_SHARED uint32_t s_stat[4];

    const int tid = threadIdx.x;
    const uint32_t p = P.part_list ? P.part_list[item] : item;
    const uint64_t rec0 = P.part_off[p], rec1 = P.part_off[p + 1];

    KTable<W> T; uint32_t* cnt; uint32_t cap;
    if (GLOBAL) {
        const uint64_t o0 = P.big_off[item]; cap = (uint32_t)(P.big_off[item + 1] - o0);
        T.keys = P.g_keys + o0 * W; T.state = P.g_state + o0; cnt = P.g_cnt + o0;
    } else {
        cap = TS; T.keys = l_keys; T.state = l_state; cnt = l_cnt;
    }
    T.mask = cap - 1;
    const uint32_t maxfill = cap - cap / 4 - COUNT_THREADS;       // leave room for in-flight claims

    if (tid == 0) { s_fill = 0; s_over = 0; s_nsolid = 0; s_wr = 0; }
    if (tid < 4) s_stat[tid] = 0;
    if (rec1 == rec0) { if (tid == 0) { P.seg_off[p] = 0; P.seg_n[p] = 0; } return; }
    ktable_clear<W>(T, tid, COUNT_THREADS);
    for (uint32_t i = tid; i < cap; i += COUNT_THREADS) cnt[i] = 0;
    __syncthreads();

    // ---- count: one record per lane-iteration ----
    const int k = P.k;
    for (uint64_t r = rec0 + tid; r < rec1; r += COUNT_THREADS) {
        if (ld_volatile_u32(&s_over)) break;
        RecView<W> R;
#pragma unroll
        for (int i = 0; i < RW; ++i) R.r[i] = P.records[r * RW + i];
        const int n = R.n();
        Kmer<W> fw = Kmer<W>::zero(), rc = Kmer<W>::zero();
        for (int i = 0; i < k - 1; ++i) { const uint32_t b = R.base(i); fw.push_right(k, b); rc.push_left(k, 3u - b); }
        for (int t = 0; t < n; ++t) {
            const uint32_t b = R.base(t + k - 1);
            fw.push_right(k, b); rc.push_left(k, 3u - b);
            const Kmer<W>& can = (rc < fw) ? rc : fw;
            bool is_new;
            const uint32_t s = ktable_insert<W, GLOBAL>(T, can, is_new);
            if (is_new) { if (atomic_add_u32(&s_fill, 1u) >= maxfill) s_over = 1; }
            atomic_add_u32(&cnt[s], 1u);
            const bool trav = (t == 0 && R.first_trav()) || (t == n - 1 && R.last_trav());
            if (trav && !(cnt[s] & TRAV_FLAG)) atomic_or_u32(&cnt[s], TRAV_FLAG);
            if (ld_volatile_u32(&s_over)) break;
        }
    }
    __syncthreads();
    if (s_over) {                                            // does not fit: defer to the big pass
        if (tid == 0) {
            if (GLOBAL) *P.error = 2;                        // scratch sizing bug: cannot happen by construction
            else { const uint32_t i = atomic_add_u32(P.big_count, 1u); P.big_list[i] = p; P.seg_off[p] = 0; P.seg_n[p] = 0; }
        }
        return;
    }

    // ---- sweep 1: statistics + number of solid entries ----
    uint32_t my_solid = 0, st_dist = 0, st_occ = 0, st_sh = 0, st_st = 0;
    for (uint32_t s = tid; s < cap; s += COUNT_THREADS) {
        if (!ktable_used<W>(T, s)) continue;
        const uint32_t c = cnt[s], n = c & ~TRAV_FLAG; const bool trav = c & TRAV_FLAG;
        if (!trav) { ++st_dist; st_occ += n; }
        if (n >= P.amin) { ++my_solid; if (trav) ++st_st; else ++st_sh; }
    }
    if (my_solid) atomic_add_u32(&s_nsolid, my_solid);
    if (st_dist) atomic_add_u32(&s_stat[0], st_dist);
    if (st_occ) atomic_add_u32(&s_stat[1], st_occ);
    if (st_sh) atomic_add_u32(&s_stat[2], st_sh);
    if (st_st) atomic_add_u32(&s_stat[3], st_st);
    __syncthreads();
    if (tid == 0) {
        uint64_t b = atomic_add_u64(P.solid_cursor, (uint64_t)s_nsolid);
        if (b + s_nsolid > P.solid_cap) { *P.error = 1; b = 0; s_nsolid = 0; }
        s_base = b;
        P.seg_off[p] = b; P.seg_n[p] = s_nsolid;
        for (int i = 0; i < 4; ++i) if (s_stat[i]) atomic_add_u64(&P.stats[i], (uint64_t)s_stat[i]);
    }
    __syncthreads();
    if (s_nsolid == 0) return;

    // ---- sweep 2: write the partition's solid segment ----
    const uint64_t obase = s_base;
    for (uint32_t s = tid; s < cap; s += COUNT_THREADS) {
        if (!ktable_used<W>(T, s)) continue;
        const uint32_t c = cnt[s];
        if ((c & ~TRAV_FLAG) < P.amin) continue;
        const uint64_t o = obase + atomic_add_u32(&s_wr, 1u);
        for (int i = 0; i < W; ++i) P.solid_keys[o * W + i] = T.keys[(uint64_t)s * W + i];
        P.solid_cnt[o] = c;
    }
}

// grid-stride over partitions (HIP limits grid*block to < 2^32 work-items)
template <int W, int TS, bool GLOBAL>
__global__ void __launch_bounds__(COUNT_THREADS) k_count(CountParams P) {
    for (uint32_t item = blockIdx.x; item < P.n_items; item += gridDim.x) {
        count_partition<W, TS, GLOBAL>(P, item);
        __syncthreads();                                 // LDS is reused by the next partition
    }
}

}  // namespace cdbg
