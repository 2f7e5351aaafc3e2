// k_count_fast.h -- stage 1b, the one-pass kernel: a minimizer partition counted in ONE LDS pass.
//
// Same job and same outputs as k_count (k_count.h; SURVEY.md section 8 row a6, abundance filter of
// /root/reference/README.md:23-25) for the partitions whose distinct k-mers fit the LDS table at once -- at
// sequencing depth that is all of them.  A partition that does not fit is appended to `retry_list` and counted
// by the generic multi-pass kernel k_count in a second launch.
//
// What the round-1 profile said (k_count VALU bound: 1044 SIMD cycles per 64 member k-mers) and what
// bench_micro/micro_r02 measured on MI355X: one ds_bpermute costs ~19 SIMD cycles per wave, a 64-bit multiply
// 18, while LDS CAS64 + add32 sustain 2.2 lane-inserts per clock and CU (28 CU cycles per wave-insert, far from
// being the limit).  Hence:
//   * owner search without shuffles: every record of a batch sets the bit of its first member position in a
//     per-wave LDS mask (ds_or_b64); lane j of step g0 reads the step's mask word (one broadcast read) and
//     popcounts the bits at or below j -> owning record.  No binary search over ds_bpermute;
//   * the record is read from the wave's LDS stage at a runtime index (the three dwords that hold the k-mer,
//     two v_alignbit) instead of four ds_bpermute + a 128-bit funnel shift built from selects;
//   * hash_lds(): two 32-bit multiplies instead of a 64-bit product;
//   * statistics live in per-thread registers until the kernel ends (no per-partition wave reductions);
//   * the output segment is carved from the workgroup's chunk by every thread redundantly (uniform
//     registers), so a partition costs two workgroup barriers: inserts done / sweep done;
//   * the wave's records for the NEXT partition and the fill count of the one after are requested right after
//     the first batch of the current partition has been staged -- behind the only s_waitcnt vmcnt the
//     iteration needs, so the requests have a whole partition's time to land.
#pragma once
#include "k_count.h"

namespace cdbg {

#ifndef CDBG_CB_WIDE
#define CDBG_CB_WIDE 8
#endif
// records per wave batch (W >= 3: CDBG_CB_WIDE -- records of up to 122-153 members fill the 64-lane steps with far fewer of them)
// Member-capped batches (one-word k-mers).  A batch of 16 records of k = 31 holds 120 +- 20 members: a third of the batches
// spilled a handful of members into a third 64-lane step, and the wave's steps ran at 80 % of their lanes (the kernel is
// bound by VALU issue).  A batch is now the longest run of records with at most CF_BATCH_MEMBERS members (and at most
// COUNT_CB = 24 records): whole steps, the last one nearly full; the position masks shrink from 14 words to 3 per wave, which
// pays for the larger stage.  128 members (two steps): count 67.4 -> 61.5 ms at config 3; 192 (three steps, a wave's ~400
// members in 3 batches instead of 4): 64.1 -> 59.0 ms on one box, 63.8 -> 58.7 on another (profiles/r04_ab_cfg3_count_batch_members.log;
// 256: 59.7; 20 / 26 / 28 / 32 records per batch: 60.9 / 70.8 / 70.5 / 70.8 -- beyond 24 the workgroup's LDS no longer fits
// three to a CU).
#ifndef CDBG_CB1
#define CDBG_CB1 24
#endif
template <int W> struct CountCap { static constexpr bool ON = W == 1; };
#ifndef CDBG_CF_BATCH
#define CDBG_CF_BATCH 192
#endif
constexpr uint32_t CF_BATCH_MEMBERS = CDBG_CF_BATCH;
static_assert(CF_BATCH_MEMBERS % 64 == 0, "whole 64-lane steps (and mask words)");
template <int W> struct CountCb { static constexpr int V = W >= 3 ? CDBG_CB_WIDE : W == 1 ? CDBG_CB1 : 16; };
// Used-slot list (one-word k-mers).  The sweep of a 4096-slot table for the ~960 keys of a config-3 partition was 19 of the
// kernel's 67 ms.  Every wave now notes the slots its own lanes claimed and sweeps exactly those.  The list costs no LDS: a
// count word holds the count in its LOWER half (15 bits + the traveller flag; a partition of the one-pass kernel cannot
// carry a count beyond that: count_fast_record_limit) and, in its UPPER half, one list entry -- the wave's j-th new slot
// lives in the upper half of word wave * (TS / waves) + j, written and read as 16 bits.  The halves never interfere: the
// count atomics (ds_add_u32 / ds_or_b32 on the whole word) never carry out of bit 15, and an LDS bank executes a word's
// atomic and a half-word write one after the other.  Upper halves are never cleared: entries beyond a wave's count are not
// read.  A wave that claims more than its TS / waves entries flags the partition for the next tier (at three quarters of
// the slots the partition is refused anyway).  Multi-word k-mers keep the full sweep: 4 slots per thread there.
template <int W> struct CountList { static constexpr bool ON = W == 1; };
// Shared stage (two-word k-mers; CDBG_COUNT_SHARED).  With a share of the RECORDS per wave (87 records of 1 .. 66 members over 8
// waves at the config-4 share) a wave holds 194 +- 46 member k-mers: 3.03 steps of 64 lanes on average -- four steps, the last
// one nearly empty -- and the workgroup waits for its fullest wave: the wait at the insert barrier was 28 - 32 % of a wave's time
// (CDBG_PROFILE_PHASES, round 4).  A partition of at most COUNT_SHARED_RECORDS records is instead staged ONCE for the whole
// workgroup (the per-wave stages, masks and record words are one pool of exactly that size), member positions run through the
// partition, and the 64-member steps are dealt out round robin: every step full but the last, every wave within one step of
// the others, for two more LDS barriers per partition: count 141.5 -> 137.2 ms at the config-4 share.  Larger partitions take the per-wave path below.
#ifndef CDBG_COUNT_SHARED
#define CDBG_COUNT_SHARED 1
#endif
template <int W> struct CountShared { static constexpr bool ON = W == 2 && CDBG_COUNT_SHARED; };
constexpr uint32_t COUNT_SHARED_MEMBERS = 4096;          // 64 mask words: one lane each in the prefix popcount
constexpr uint32_t CF_TRAV16 = 0x8000u;                  // the traveller flag in a 16-bit count
// (host) partitions of at least this many records are not tried by the one-pass kernel: a k-mer occurs at most once per member
// position, so below it no count reaches 2^15 (CountList) / the saturation threshold of the 31-bit counts
template <int W> inline uint32_t count_fast_record_limit(int k) {
    if (!CountList<W>::ON) return COUNT_FAST_MAX_RECORDS;
    return 0x7FFFu / (uint32_t)(RecFmt<W>::CAPB - k + 1) + 1u;
}
template <int W> struct CountGeom {
    // member positions of one batch: COUNT_CB records of at most CAPB - k + 1 members, k >= 3 (W = 1), 32, 64, 96
    static constexpr int KMIN = W == 1 ? 3 : 32 * (W - 1);
    static constexpr int NMAX = RecFmt<W>::CAPB - KMIN + 1;
    static constexpr int MASKW = CountCap<W>::ON ? (int)(CF_BATCH_MEMBERS / 64) : (CountCb<W>::V * NMAX + 63) / 64;
    static_assert(!CountCap<W>::ON || NMAX <= (int)CF_BATCH_MEMBERS, "a record must fit a batch");
};
// LDS of one workgroup:
//   keys / cnt   the open-address table (claim word EMPTY <=> slot free)
//   stage        per wave: the COUNT_CB records of the batch being expanded
//   rinfo        per wave: 128 - 2k + 2 * (first member position) of each staged record (bit offset of member g = rinfo - 2g),
//                plus the record's four boundary facts in the top bits (see where it is written)
//   smask        per wave: bit g set <=> a record of the batch starts at member position g
//   emask        per wave: bit g set <=> the member at position g is the last of a record that has something to say about it
//   fp           sifting tier only (FS > 0, below): FS fingerprint words
// (sifting tier: members per wave and partition whose fingerprint word is remembered from pass 0; the rest look theirs up again)
#ifdef CDBG_HOSTSIM
constexpr uint32_t SIFT_MS_CAP = 96;                     // (simulator: the small test partitions run through both kinds of member)
#else
constexpr uint32_t SIFT_MS_CAP = 1280;
#endif
template <int W, int TS, int NT, int FS = 0>
struct CountFastLds {
    uint64_t keys[TS * W];
    uint32_t cnt[TS];
    uint32_t fp[FS > 0 ? FS : 1]; uint32_t fpfill;       // fpfill: distinct fingerprints of the partition
    uint16_t mslot[FS > 0 ? (NT / 64) * SIFT_MS_CAP : 2];   // sifting tier: per wave, the fingerprint word of its j-th member k-mer (pass 0 -> pass 1)
    uint16_t ring[FS > 0 ? (NT / 64) * 128 : 2];            // sifting tier, pass 1: per wave, the members seen again that wait for a full 64-lane step
    uint64_t stage[(NT / 64) * CountCb<W>::V * RecFmt<W>::RW + 2];   // + 2: the dword window of the last record may over-read (up to 4 dwords)
    uint64_t smask[(NT / 64) * CountGeom<W>::MASKW];
    uint64_t emask[(NT / 64) * CountGeom<W>::MASKW];
    uint32_t rinfo[(NT / 64) * CountCb<W>::V];
    uint32_t wsum[NT / 64];                              // (shared stage) member k-mers of every wave's records
    uint32_t fill[2], wr[2];                             // per partition parity: new keys / solid entries written
    uint32_t over;
    uint64_t cbase;                                      // chunk hand-out broadcast
};

struct CountFastParams {
    CountParams c;                                       // item_off / g_* unused here; part_list only by the LISTED tier
    uint32_t* retry_list; uint32_t* retry_count;         // partitions that need the multi-pass kernel
    uint32_t fast_max_records;                           // partitions with more records are not tried (<= COUNT_FAST_MAX_RECORDS)
    uint32_t skip_fill_q8;                               // 0: off.  Else a partition whose PREDICTED distinct k-mers (records x the workgroup's
                                                         // running distinct-per-record average) exceed skip_fill_q8 / 256 of the table goes to the next tier untried
};

// raw words of a partition's record range as loaded: resolved one partition later, so that no load is waited for
// in the partition that issues it (unconditional loads from a clamped index: a select on a loaded value would
// put an s_waitcnt right behind the load).  CAPPED: fixed-capacity regions + fill counts; else exact offsets.
// (CAPPED is a small bit set: bit 0 = capped layout, bit 1 = LISTED: the items are entries of P.part_list -- the second
//  tier that takes the first tier's retry list with a table twice the size; the list entry is one more dependent load)
template <int CAPPED> struct CountRaw;
template <> struct CountRaw<1> { uint32_t f; };
template <> struct CountRaw<0> { uint64_t a, b; };
template <> struct CountRaw<3> { uint32_t f, p; };
template <> struct CountRaw<2> { uint64_t a, b; uint32_t p; };
template <int CAPPED>
CDBG_DEV CountRaw<CAPPED> count_raw_load(const CountParams& P, uint32_t item) {
    const uint32_t i = item < P.n_items ? item : P.n_items - 1u;
    CountRaw<CAPPED> r;
    uint32_t p = i + P.item_base;
    if constexpr (CAPPED & 2) { p = P.part_list[i]; r.p = p; }
    if constexpr (CAPPED & 1) r.f = P.part_fill[p];
    else if (P.part_pairs) { r.a = P.part_off[2ull * p]; r.b = P.part_off[2ull * p + 1]; }
    else { r.a = P.part_off[p]; r.b = P.part_off[p + 1]; }
    return r;
}
struct CountRange { uint64_t rec0; uint32_t n; uint32_t p; };      // records [rec0, rec0 + n) of partition p (n == 0: nothing to do here); wave-uniform
template <int CAPPED>
CDBG_DEV CountRange count_raw_resolve(const CountParams& P, uint32_t item, const CountRaw<CAPPED>& w) {
    CountRange r; r.p = item + P.item_base; r.rec0 = 0; r.n = 0;
    if (item >= P.n_items) return r;
    if constexpr (CAPPED & 2) r.p = uni_u32(w.p);
    if constexpr (CAPPED & 1) { const uint32_t f = uni_u32(w.f); r.rec0 = (uint64_t)r.p * P.part_stride; r.n = f > P.part_stride ? 0u : f; }   // spilled: counted by the repair launch
    else { const uint64_t a = uni_u64(w.a), b = uni_u64(w.b); r.rec0 = a; r.n = (uint32_t)(b - a); }
    return r;
}
// this wave's share of a partition's records: [w0, w1)
template <int NW>
CDBG_DEV void count_wave_share(const CountRange& rg, int wave, uint64_t& w0, uint64_t& w1) {
    const uint64_t per_wave = ((uint64_t)rg.n + NW - 1) / NW;
    const uint64_t end = rg.rec0 + rg.n;
    w0 = rg.rec0 + (uint64_t)wave * per_wave;
    if (w0 > end) w0 = end;
    w1 = (w0 + per_wave < end) ? w0 + per_wave : end;
}
// Three- and four-word k-mers (k >= 64): a partition holds few records (~20) of very unequal size (1 .. CAPB - k + 1
// members), so equal RECORD shares left the last wave of a workgroup 40 % of its time at the barrier (k = 127: the first
// wave of four took ceil(n / 4) records, the last what remained).  There every wave walks ALL records of the partition and
// takes an equal share of the MEMBERS of each 64-record chunk: a record may be split between two waves (a wave stages it
// with the index of its first member).  Measured at the config-5 share: count 526 -> 481 ms.  Not for two-word k-mers
// (k = 55: ~68 records per partition over 8 waves): balancing every chunk by itself costs every wave a fourth, nearly
// empty step for the 4 records of the second chunk (count 180 -> 205 ms), and balancing the whole partition (meta words
// of the next chunks prefetched with the first) was no better than the record shares (200 ms): eight waves reading and
// prefix-summing all records eat what the shorter barrier wait gives.  One-word k-mers: ~48 records per wave even out.
template <int W> struct CountBal { static constexpr bool ON = W > 2; };
template <int W, int NW>
CDBG_DEV void count_share(const CountRange& rg, int wave, uint64_t& w0, uint64_t& w1) {
    if (CountBal<W>::ON) { w0 = rg.rec0; w1 = rg.rec0 + rg.n; }
    else count_wave_share<NW>(rg, wave, w0, w1);
}
template <int W>
CDBG_DEV void count_load_chunk(const CountParams& P, uint64_t first, uint64_t end, int lane, RecView<W>& R) {
    constexpr int RW = RecFmt<W>::RW;
#pragma unroll
    for (int i = 0; i < RW; ++i) R.r[i] = 0;
    if (first + (uint64_t)lane < end) {
#pragma unroll
        for (int i = 0; i < RW; ++i) R.r[i] = P.records[(first + lane) * RW + i];
    }
}
struct CountAcc { uint32_t dist, sh, st, last_fill; uint64_t occ;
#ifdef CDBG_PROFILE_PHASES
    uint64_t ph[8], t_prev;
#endif
};
#if defined(CDBG_PROFILE_PHASES) && !defined(CDBG_HOSTSIM)
#define CDBG_FPH(i) do { const uint64_t t_ = clock64(); acc.ph[i] += t_ - acc.t_prev; acc.t_prev = t_; } while (0)
#else
#define CDBG_FPH(i) do { } while (0)
#endif
// what the kernel's loop carries one partition ahead
// The loop is unrolled twice over two register sets (ping-pong): a register COPY of a requested value would be its
// first use and put the wait for it at the end of the partition that issued the request.
template <int W, int CAPPED>
struct CountSet { RecView<W> R; CountRaw<CAPPED> raw; };    // R: this wave's records of a partition; raw: range words of the partition after it    // R: this wave's records of a partition; raw: range words of the partition after it
template <int W, int CAPPED>
struct CountAhead { CountSet<W, CAPPED>* cur; CountSet<W, CAPPED>* nxt; CountRange rg_nxt; uint32_t item_nxt, item_nn; bool issued; };
// resolve the next partition's range from the words requested one partition ago, request this wave's share of its
// records and the range words of the partition after it
template <int W, int NW, int CAPPED>
CDBG_DEV void count_issue_ahead(const CountParams& P, CountAhead<W, CAPPED>& A, int wave, int lane) {
    A.rg_nxt = count_raw_resolve<CAPPED>(P, A.item_nxt, A.cur->raw);
    uint64_t w0, w1; count_share<W, NW>(A.rg_nxt, wave, w0, w1);
    count_load_chunk<W>(P, w0, w1, lane, A.nxt->R);
    A.nxt->raw = count_raw_load<CAPPED>(P, A.item_nn);
    A.issued = true;
}

// Sifting tier (FS > 0; multi-word k-mers, abundance-min >= 2).  At k = 127 and 1 % errors 95 % of the distinct k-mers of a
// partition occur once and are dropped by the abundance filter -- yet each of them takes a slot of 8 W + 4 bytes, and the
// partitions they overfill went through a 4096-slot tier that leaves room for ONE workgroup per CU and then through the
// multi-pass kernel (config-5 share: 67 + 52 of 201 ms of counting for 7 % of the partitions).  Here a partition is read
// twice: the first pass leaves a 30-bit fingerprint of every k-mer in a table of FS 4-byte words, with "seen again" in bit 1;
// the second pass gives the exact table only the k-mers whose fingerprint was seen again.  Exact all the same: a fingerprint
// seen once IS one occurrence of one k-mer (counted as a distinct k-mer of abundance 1 on the spot), and k-mers that share
// a fingerprint are told apart by their keys in the exact table as before.  The exact table is a quarter of tier 1's.
// Round 5: pass 1 no longer extracts every member a second time.  Pass 0 leaves, per member, the fingerprint WORD it ended at
// (13 bits, L.mslot: the wave's j-th member); pass 1 runs a light step over the members -- slot -> "seen again" bit, nothing else:
// 71 % of them at the config-5 share learn here that their k-mer occurred once -- and gathers the others in a small ring; a heavy
// step (cut the k-mer out of the record, reverse complement, hash, exact insert) runs whenever 64 of them wait: every heavy
// lane works.  Members beyond SIFT_MS_CAP per wave take the heavy step and look their fingerprint up by its tag as before.
// returns false when the partition did not fit one pass (table left dirty)
template <int W, int TS, int NT, int CAPPED, int FS = 0>
CDBG_DEV bool count_partition_fast(const CountParams& P, CountFastLds<W, TS, NT, FS>& L, const CountRange& rg, CountAhead<W, CAPPED>& A,
                                   const uint32_t par, uint64_t& chunk_base, uint32_t& chunk_left, CountAcc& acc) {
    constexpr bool SIFT = FS > 0;
    static_assert(!SIFT || W > 1, "the sifting tier is written for multi-word k-mers");
    constexpr int LOG_FS = FS == 16384 ? 14 : FS == 8192 ? 13 : FS == 4096 ? 12 : FS == 2048 ? 11 : 0;
    static_assert(!SIFT || LOG_FS > 0, "fingerprint table size");
    constexpr int RW = RecFmt<W>::RW;
    constexpr int NW = NT / 64;
    constexpr int MASKW = CountGeom<W>::MASKW;
    constexpr int LOG_TS = TS == 8192 ? 13 : TS == 4096 ? 12 : TS == 2048 ? 11 : TS == 1024 ? 10 : TS == 512 ? 9 : -1;
    static_assert(LOG_TS > 0, "table size");
    const int tid = threadIdx.x, lane = tid & 63, wave = (int)uni_u32((uint32_t)tid >> 6);
    const int k = P.k;
    constexpr int COUNT_CB = CountCb<W>::V;
    uint64_t* const stage = L.stage + (size_t)wave * COUNT_CB * RW;
    uint64_t* const smask = L.smask + (size_t)wave * MASKW;
    uint64_t* const emask = L.emask + (size_t)wave * MASKW;
    uint32_t* const rinfo = L.rinfo + (size_t)wave * COUNT_CB;
    const uint64_t lane_le = lane == 63 ? ~0ULL : ((2ULL << lane) - 1ULL);      // bits 0 .. lane
    const uint32_t RBITS = 64u * RW;                                            // bits of a record
    const uint64_t kmask1 = ~0ULL >> (64 - 2 * (k < 32 ? k : 31));             // (W == 1 only)

    uint64_t w0, w1; count_share<W, NW>(rg, wave, w0, w1);
    CDBG_FPH(0);
    uint32_t n_new = 0, n_fp = 0;                                             // wave-uniform: keys / fingerprints this wave added
    uint32_t n_once = 0;                                                      // (sifting tier) this lane's home k-mers of abundance 1: counted when the partition has fitted
    constexpr uint32_t LCAP = (uint32_t)(TS / NW);                             // list entries of a wave (CountList)
    const uint32_t lbase = (uint32_t)wave * LCAP;
    bool shared_done = false;
    if constexpr (CountShared<W>::ON && FS == 0) {
        constexpr uint32_t SREC = (uint32_t)(NW * COUNT_CB);                   // records the pooled stage holds
        static_assert(NW * MASKW >= (int)(COUNT_SHARED_MEMBERS / 64), "the pooled masks must cover COUNT_SHARED_MEMBERS positions");
        if (rg.n <= SREC) {                                                   // uniform; every wave's share is in its prefetched registers (<= 64 records)
            const RecView<W>& R = A.cur->R;
            const uint32_t nrec_w = (uint32_t)(w1 - w0);
            const uint32_t nfull = (uint32_t)lane < nrec_w ? (uint32_t)R.n() : 0u;
            const uint32_t incl = wave_incl_sum_u32(nfull);
            const bool odd = __any((uint32_t)lane < nrec_w && nfull == 0u);     // (a record without members would shift the ranks: per-wave path)
            const uint32_t wtot = wave_readlane_u32(incl, 63);
            if (lane == 0) L.wsum[wave] = odd ? COUNT_SHARED_MEMBERS + 1u : wtot;
            CDBG_LDS_BARRIER();                                                   // ---- every wave's member count is known ----
            uint32_t base = 0, total = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { const uint32_t v = uni_u32(L.wsum[w]); if (w < wave) base += v; total += v; }
            if (total <= COUNT_SHARED_MEMBERS) {                              // uniform
                shared_done = true;
                if ((uint32_t)lane < nrec_w) {
                    const uint32_t gi = (uint32_t)(w0 - rg.rec0) + (uint32_t)lane;   // the record's index in the partition = its rank among the start bits
#pragma unroll
                    for (int i = 0; i < RW; ++i) L.stage[gi * RW + i] = R.r[i];
                    const uint32_t excl = base + incl - nfull, meta = (uint32_t)R.r[0];
                    const uint32_t f0 = (meta >> 10) & 1u, f1 = (meta >> 11) & 1u, ft = (meta >> 8) & 1u, lt = (meta >> 9) & 1u;
                    L.rinfo[gi] = (RBITS - 2u * (uint32_t)k + 2u * excl) | (f1 << 31) | (f0 << 30) | (f0 << 29) | (f1 << 28) | (ft << 27) | (lt << 26);   // (low 14 bits: < 256 + 2 * 4096)
                    atomic_or_u64(&L.smask[excl >> 6], 1ULL << (excl & 63u));
                    const uint32_t last = excl + nfull - 1u;
                    if (f1 | lt) atomic_or_u64(&L.emask[last >> 6], 1ULL << (last & 63u));
                }
                if (!A.issued) { CDBG_FPH(1); count_issue_ahead<W, NW, CAPPED>(P, A, wave, lane); }
                CDBG_LDS_BARRIER();                                               // ---- stage, record words and masks are complete ----
                // records that start before each 64-member word: prefix popcount, one mask word per lane
                const uint32_t words = (total + 63u) >> 6;
                const uint32_t pc = (uint32_t)lane < words ? (uint32_t)__popcll(L.smask[lane]) : 0u;
                const uint32_t before_word = wave_incl_sum_u32(pc) - pc;
                // (steps TAKEN one by one from an LDS counter instead of dealt round robin: 138.2 against 137.2 ms at the config-4 share)
                for (uint32_t g0 = 64u * (uint32_t)wave; g0 < total; g0 += 64u * (uint32_t)NW) {   // wave-uniform
                    const uint32_t wi = g0 >> 6;
                    const uint64_t M = uni_u64(L.smask[wi]), E = uni_u64(L.emask[wi]);
                    const uint32_t started = wave_readlane_u32(before_word, (int)wi);
                    const uint32_t g = g0 + (uint32_t)lane;
                    bool is_new = false;
                    if (g < total) {
                        const uint32_t slot = started + (uint32_t)__popcll(M & lane_le) - 1u;
                        const uint32_t ri = L.rinfo[slot];
                        const uint32_t sh = (ri & 0x3FFFu) - 2u * g;
                        const uint32_t as_first = lane_pick_u32(M, ri, lane), as_last = lane_pick_u32(E, ri, lane);
                        const uint32_t facts = (as_first & 0x68000000u) | (as_last & ~0x68000000u);
                        const bool trav = (facts & 0x0C000000u) != 0u;
                        const uint32_t* dw = reinterpret_cast<const uint32_t*>(L.stage + slot * RW) + (sh >> 5);
                        uint32_t in[2 * W + 1];
#pragma unroll
                        for (int j = 0; j < 2 * W + 1; ++j) in[j] = dw[j];
                        Kmer<W> fw;
#pragma unroll
                        for (int i = 0; i < W; ++i)
                            fw.w[i] = ((uint64_t)alignbit_u32(in[2 * i + 2], in[2 * i + 1], sh) << 32) | alignbit_u32(in[2 * i + 1], in[2 * i], sh);
                        fw.mask(k);
                        const Kmer<W> rc = fw.rc(k);
                        const bool rev = rc < fw;
                        const Kmer<W>& can = rev ? rc : fw;
                        const uint64_t ktop = can.w[W - 1] | ((uint64_t)((rev ? facts << 2 : facts) & 0xC0000000u) << 32);
                        uint32_t s = can.hash_lds() >> (32 - LOG_TS);
                        const uint64_t top = ktop, ptop = key_pending(ktop);
                        uint32_t probes = 0;
                        bool hit = false;
#pragma clang loop unroll(disable)
                        do {                                                   // (the claim protocol of the per-wave path below)
                            uint64_t* const slotp = &L.keys[(uint64_t)s * W];
                            const uint64_t old = atomic_cas_u64(slotp + (W - 1), KEY_EMPTY, ptop);
                            bool eq = true;
                            if (W == 2) {
                                CDBG_COMPILER_BARRIER();
                                eq = slotp[0] == can.w[0];
                            } else if (old == top) {
#pragma unroll
                                for (int i = 0; i < W - 1; ++i) eq &= (slotp[i] == can.w[i]);
                            }
                            const bool mine = old == KEY_EMPTY, same = (old == top) & eq, wait = old == ptop;
                            if (wait) CDBG_SPIN_YIELD();
                            if (mine) {
#pragma unroll
                                for (int i = 0; i < W - 1; ++i) slotp[i] = can.w[i];
                                CDBG_LDS_FENCE();
                                atomic_exch_u64(slotp + (W - 1), top);
                            }
                            is_new = is_new | mine; hit = mine | same;
                            const bool advance = !(hit | wait);
                            s = advance ? ((s + 1) & (TS - 1)) : s; probes += advance ? 1u : 0u;
                        } while (!hit && probes < 64u);
                        if (!hit) L.over = 1;
                        else {
                            atomic_add_u32(&L.cnt[s], 1u);
                            if (trav) atomic_or_u32(&L.cnt[s], TRAV_FLAG);
                        }
                    }
                    n_new += (uint32_t)__popcll(__ballot(is_new));
                }
            }
        }
    }
    if (!shared_done)
#pragma clang loop unroll(disable)
    for (int phase = SIFT ? 0 : 1; phase < 2; ++phase) {                       // (sifting tier: 0 = fingerprints, 1 = exact counts of what was seen again)
    RecView<W> R = A.cur->R;
    uint32_t mbase = 0;                                                       // (sifting tier) members of this wave's earlier batches: the same numbering in both passes
    for (uint64_t c0 = w0; c0 < w1; c0 += 64) {                               // wave-uniform
        if (c0 != w0) count_load_chunk<W>(P, c0, w1, lane, R);                 // (the first 64 records came prefetched)
        const int nrec = (int)((w1 - c0) < 64 ? (w1 - c0) : 64);
        const uint32_t nfull = lane < nrec ? (uint32_t)R.n() : 0u;             // members of the lane's record
        uint32_t n = nfull, first = 0;                                        // ... of which this wave takes [first, first + n)
        if (CountBal<W>::ON) {
            const uint32_t gi = wave_incl_sum_u32(nfull), mc = wave_readlane_u32(gi, 63), ge = gi - nfull;
            const uint32_t m_lo = mc * (uint32_t)wave / (uint32_t)NW, m_hi = mc * ((uint32_t)wave + 1u) / (uint32_t)NW;   // (mc <= 64 * 248)
            const uint32_t sb = ge > m_lo ? ge : m_lo, se = gi < m_hi ? gi : m_hi;
            n = se > sb ? se - sb : 0u; first = sb - ge;
        }
        const uint32_t incl = wave_incl_sum_u32(n);
        // (the wave's records with members are consecutive lanes; the stage index of a record is its rank among them)
        const int fa = CountBal<W>::ON ? (int)uni_u32((uint32_t)__builtin_ctzll(__ballot(n != 0u) | (1ULL << 63))) : 0;
        for (int lo = 0, nb = COUNT_CB; lo < nrec; lo += nb) {                 // batches of at most COUNT_CB records
            const uint32_t before = lo ? wave_readlane_u32(incl, lo - 1) : 0u;
            if (CountCap<W>::ON) {
                // the records from `lo` on whose members end within CF_BATCH_MEMBERS positions (prefix sums are monotone: a run; >= 1 record)
                const uint64_t fit = __ballot(lane >= lo && lane < lo + COUNT_CB && lane < nrec && incl - before <= CF_BATCH_MEMBERS);
                nb = (int)uni_u32((uint32_t)__popcll(fit));
            }
            const uint32_t total = wave_readlane_u32(incl, lo + nb - 1 < 63 ? lo + nb - 1 : 63) - before;
            const uint32_t excl = incl - n - before;
            if (CountBal<W>::ON && total == 0u && A.issued) continue;        // (uniform) none of these 16 records has members for this wave
            if (lane >= lo && lane < lo + nb && n) {
                const int si = lane - (fa > lo ? fa : lo);
#pragma unroll
                for (int i = 0; i < RW; ++i) stage[si * RW + i] = R.r[i];
                // Bit base of the record's members (even, < 2^13) and, in the bits above, the record's four boundary facts (meta
                // bits 8..11, k_scan.h) laid out for the member loop: a bit-field merge of the word as seen by the FIRST member
                // and as seen by the LAST one leaves each fact in the lanes it applies to, and the two foreign-junction facts
                // land where the key wants them (bits 31 / 30 = bits 63 / 62 of the top word) -- twice, so that a reversed
                // member shifts the swapped pair in.  A wave that holds only a part of the record keeps the facts of its ends.
                const uint32_t meta = (uint32_t)R.r[0];
                const uint32_t head = first == 0u ? meta : 0u, tail = first + n == nfull ? meta : 0u;
                const uint32_t f0 = (head >> 10) & 1u, f1 = (tail >> 11) & 1u, ft = (head >> 8) & 1u, lt = (tail >> 9) & 1u;
                rinfo[si] = (RBITS - 2u * (uint32_t)k - 2u * first + 2u * excl) | (f1 << 31) | (f0 << 30) | (f0 << 29) | (f1 << 28) | (ft << 27) | (lt << 26);
                atomic_or_u64(&smask[excl >> 6], 1ULL << (excl & 63u));
                const uint32_t last = excl + n - 1u;
                if (f1 | lt) atomic_or_u64(&emask[last >> 6], 1ULL << (last & 63u));
            }
            CDBG_WAVE_SYNC();
            if (!A.issued) { CDBG_FPH(1); count_issue_ahead<W, NW, CAPPED>(P, A, wave, lane); }        // behind the wait for this partition's own records
            if constexpr (SIFT) if (phase == 1) {
                // ---- sifting tier, pass 1: light steps over all members, heavy steps over the ones seen again ----
                static_assert(!SIFT || (COUNT_CB <= 8 && COUNT_CB * (CountGeom<W>::NMAX < 255 ? CountGeom<W>::NMAX : 255) <= 2048), "ring entry: 11 bits of member position (a record holds at most 255 members: its 8-bit field), 3 bits of record");
                uint16_t* const ring = L.ring + (size_t)wave * 128;
                const uint16_t* const ms = L.mslot + (size_t)wave * SIFT_MS_CAP;
                uint32_t rn = 0, rhead = 0;                                   // wave-uniform: entries waiting / first of them
                auto heavy = [&](const uint32_t cnt) {                        // the first cnt (<= 64) waiting members, one per lane
                    bool is_new = false;
                    if ((uint32_t)lane < cnt) {
                        const uint32_t e = ring[(rhead + (uint32_t)lane) & 127u];
                        const uint32_t g = e & 0x7FFu, slot = (e >> 11) & 7u;
                        const uint32_t ri = rinfo[slot];
                        const uint32_t sh = (ri & 0x1FFFu) - 2u * g;
                        const uint32_t isf = (uint32_t)(smask[g >> 6] >> (g & 63u)) & 1u, isl = (uint32_t)(emask[g >> 6] >> (g & 63u)) & 1u;
                        const uint32_t facts = ((isf ? ri : 0u) & 0x68000000u) | ((isl ? ri : 0u) & ~0x68000000u);
                        const bool trav = (facts & 0x0C000000u) != 0u;
                        const uint32_t* dw = reinterpret_cast<const uint32_t*>(stage + slot * RW) + (sh >> 5);
                        uint32_t in[2 * W + 1];
#pragma unroll
                        for (int j = 0; j < 2 * W + 1; ++j) in[j] = dw[j];
                        Kmer<W> fw;
#pragma unroll
                        for (int i = 0; i < W; ++i)
                            fw.w[i] = ((uint64_t)alignbit_u32(in[2 * i + 2], in[2 * i + 1], sh) << 32) | alignbit_u32(in[2 * i + 1], in[2 * i], sh);
                        fw.mask(k);
                        const Kmer<W> rc = fw.rc(k);
                        const bool rev = rc < fw;
                        const Kmer<W>& can = rev ? rc : fw;
                        const uint64_t ktop = can.w[W - 1] | ((uint64_t)((rev ? facts << 2 : facts) & 0xC0000000u) << 32);
                        const uint32_t hh = can.hash_lds();
                        uint32_t s = hh >> (32 - LOG_TS);
                        bool go = true;
                        if (e & 0x8000u) {                                     // beyond the remembered members: the fingerprint is found by its tag
                            uint32_t fs = hh >> (32 - LOG_FS), probes = 0, old;
                            const uint32_t tag = (((hh ^ (hh >> 15)) * 0x2C1B3C6Du) & ~3u) | 1u;
#pragma clang loop unroll(disable)
                            for (;;) {                                         // (the tag is there: pass 0 put it)
                                old = L.fp[fs];
                                if ((old & ~2u) == tag || old == 0u || ++probes == 64u) break;
                                fs = (fs + 1) & (FS - 1);
                            }
                            if (!(old & 2u)) { go = false; if (!trav) ++n_once; }
                        }
                        if (go) {
                            const uint64_t top = ktop, ptop = key_pending(ktop);
                            uint32_t probes = 0;
                            bool hit = false;
#pragma clang loop unroll(disable)
                            do {                                               // (single exit, publish inside the iteration: see ktable_insert)
                                uint64_t* const slotp = &L.keys[(uint64_t)s * W];
                                const uint64_t old = atomic_cas_u64(slotp + (W - 1), KEY_EMPTY, ptop);
                                bool eq = true;
                                if (old == top) {
#pragma unroll
                                    for (int i = 0; i < W - 1; ++i) eq &= (slotp[i] == can.w[i]);
                                }
                                const bool mine = old == KEY_EMPTY, same = (old == top) & eq, wait = old == ptop;
                                if (wait) CDBG_SPIN_YIELD();
                                if (mine) {
#pragma unroll
                                    for (int i = 0; i < W - 1; ++i) slotp[i] = can.w[i];
                                    CDBG_LDS_FENCE();
                                    atomic_exch_u64(slotp + (W - 1), top);
                                }
                                is_new = is_new | mine; hit = mine | same;
                                const bool advance = !(hit | wait);
                                s = advance ? ((s + 1) & (TS - 1)) : s; probes += advance ? 1u : 0u;
                            } while (!hit && probes < 64u);
                            if (!hit) L.over = 1;
                            else {
                                atomic_add_u32(&L.cnt[s], 1u);
                                if (trav) atomic_or_u32(&L.cnt[s], TRAV_FLAG);
                            }
                        }
                    }
                    n_new += (uint32_t)__popcll(__ballot(is_new));
                };
                uint32_t started1 = 0;
                for (uint32_t g0 = 0; g0 < total; g0 += 64) {                 // wave-uniform trip count
                    const uint64_t M = uni_u64(smask[g0 >> 6]), E = uni_u64(emask[g0 >> 6]);
                    const uint32_t g = g0 + (uint32_t)lane;
                    const uint32_t slot = started1 + (uint32_t)__popcll(M & lane_le) - 1u;
                    started1 += (uint32_t)__popcll(M);
                    bool want = false; uint32_t entry = 0;
                    if (g < total) {
                        const uint32_t mi = mbase + g;
                        entry = g | (slot << 11);
                        if (mi < SIFT_MS_CAP) {
                            const uint32_t old = L.fp[ms[mi]];
                            want = (old & 2u) != 0u;
                            if (!want) {                                       // seen once: one k-mer of abundance 1 (a traveller copy is not counted)
                                const uint32_t ri = rinfo[slot];
                                const uint32_t as_first = lane_pick_u32(M, ri, lane), as_last = lane_pick_u32(E, ri, lane);
                                if (!(((as_first & 0x68000000u) | (as_last & ~0x68000000u)) & 0x0C000000u)) ++n_once;
                            }
                        } else { want = true; entry |= 0x8000u; }
                    }
                    const uint64_t wb = __ballot(want);
                    if (want) ring[(rhead + rn + (uint32_t)__popcll(wb & (lane_le >> 1))) & 127u] = (uint16_t)entry;
                    rn += (uint32_t)__popcll(wb);
                    CDBG_WAVE_SYNC();
                    if (rn >= 64u) { heavy(64u); rhead = (rhead + 64u) & 127u; rn -= 64u; CDBG_WAVE_SYNC(); }
                }
                if (rn) { heavy(rn); }
                mbase += total;
                CDBG_WAVE_SYNC();                                              // every lane has read the stage and the masks
                if (lane < MASKW && lane <= (int)((total + 63) >> 6)) { smask[lane] = 0; emask[lane] = 0; }
                CDBG_WAVE_SYNC();
                continue;
            }
            uint32_t started = 0;                                             // records of the batch that start before the step's window
            for (uint32_t g0 = 0; g0 < total; g0 += 64) {                     // wave-uniform trip count
                const uint64_t M = uni_u64(smask[g0 >> 6]), E = uni_u64(emask[g0 >> 6]);   // (scalar: used as lane predicates below)
                const uint32_t g = g0 + (uint32_t)lane;
                const bool active = g < total;
                uint32_t slot_new = 0;
                const uint32_t slot = started + (uint32_t)__popcll(M & lane_le) - 1u;   // active lanes: >= 0 (member 0 starts record 0)
                started += (uint32_t)__popcll(M);
                bool is_new = false;
                if (active) {
                    const uint32_t ri = rinfo[slot];
                    const uint32_t sh = (ri & 0x1FFFu) - 2u * g;              // the member's k-mer = record bits [sh, sh + 2k)
                    // the record's facts about its first / last member, in the lanes that hold that member
                    const uint32_t as_first = lane_pick_u32(M, ri, lane), as_last = lane_pick_u32(E, ri, lane);
                    const uint32_t facts = (as_first & 0x68000000u) | (as_last & ~0x68000000u);   // (v_bfi_b32)
                    const bool trav = (facts & 0x0C000000u) != 0u;
                    Kmer<W> fw;
                    if (W == 1) {
                        // 128-bit record, sh >= 16: the three dwords from bit sh on, two funnel shifts
                        const uint32_t* dw = reinterpret_cast<const uint32_t*>(stage + slot * RW) + (sh >> 5);
                        const uint32_t d0 = dw[0], d1 = dw[1], d2 = dw[2];
                        const uint32_t lo32 = alignbit_u32(d1, d0, sh), hi32 = alignbit_u32(d2, d1, sh);
                        fw.w[0] = (((uint64_t)hi32 << 32) | lo32) & kmask1;
                    } else {
                        // the same dword window for W words: 2 W + 1 dwords from bit sh on, 2 W funnel shifts (a select chain
                        // over the RW record words per output word before: ~110 v_cndmask per k-mer at W = 4)
                        const uint32_t* dw = reinterpret_cast<const uint32_t*>(stage + slot * RW) + (sh >> 5);
                        uint32_t in[2 * W + 1];
#pragma unroll
                        for (int j = 0; j < 2 * W + 1; ++j) in[j] = dw[j];
#pragma unroll
                        for (int i = 0; i < W; ++i)
                            fw.w[i] = ((uint64_t)alignbit_u32(in[2 * i + 2], in[2 * i + 1], sh) << 32) | alignbit_u32(in[2 * i + 1], in[2 * i], sh);
                        fw.mask(k);
                    }
                    const Kmer<W> rc = fw.rc(k);
                    const bool rev = rc < fw;
                    const Kmer<W>& can = rev ? rc : fw;
                    // the key carries the foreign-junction flags (KEY_FOREIGN_*, k_count.h): the same for every occurrence
                    // (bits 31 / 30 of `facts`: right / left junction foreign in read orientation; bits 29 / 28: the same two swapped)
                    const uint64_t ktop = can.w[W - 1] | ((uint64_t)((rev ? facts << 2 : facts) & 0xC0000000u) << 32);
                    const uint32_t hh = can.hash_lds();
                    uint32_t s = hh >> (32 - LOG_TS);
                    bool hit = true, go = true;
                    if constexpr (SIFT) {
                        // the fingerprint's slot: the hash's top bits; its 30-bit tag: a mix of all of them (bit 0 set: never 0 = free)
                        uint32_t fs = hh >> (32 - LOG_FS);
                        const uint32_t tag = (((hh ^ (hh >> 15)) * 0x2C1B3C6Du) & ~3u) | 1u;
                        uint32_t probes = 0;
                        {                                                  // (pass 0 only: pass 1 is the branch above)
                            go = false;
#pragma clang loop unroll(disable)
                            for (;;) {
                                const uint32_t old = atomic_cas_u32(&L.fp[fs], 0u, tag);
                                if (old == 0u) { is_new = true; break; }
                                if ((old & ~2u) == tag) { if (!(old & 2u)) atomic_or_u32(&L.fp[fs], 2u); break; }
                                fs = (fs + 1) & (FS - 1);
                                if (++probes == 64u) { hit = false; break; }
                            }
                            if (mbase + g < SIFT_MS_CAP) L.mslot[(size_t)wave * SIFT_MS_CAP + mbase + g] = (uint16_t)fs;   // where pass 1 finds this member's "seen again" bit
                        }
                    }
                    if (!go) { if (!hit) L.over = 1; }
                    else {
                    if (W == 1) {
                        // only `old` and `s` live out of the probe loop: every flag carried across its back edge costs three
                        // scalar mask operations per probe, and the longest probe sequence of the 64 lanes sets the trip count
                        uint32_t probes = 0; uint64_t old;
#pragma clang loop unroll(disable)
                        for (;;) {
                            old = atomic_cas_u64(&L.keys[s], ~0ULL, ktop);
                            if ((old == ~0ULL) | (old == ktop) | (++probes == 64u)) break;
                            s = (s + 1) & (TS - 1);
                        }
                        is_new = old == ~0ULL;
                        hit = is_new | (old == ktop);
                    } else {
                        const uint64_t top = ktop, ptop = key_pending(ktop);
                        uint32_t probes = 0;
                        hit = false;
#pragma clang loop unroll(disable)
                        do {                                                   // (single exit, publish inside the iteration: see ktable_insert)
                            uint64_t* const slotp = &L.keys[(uint64_t)s * W];
                            const uint64_t old = atomic_cas_u64(slotp + (W - 1), KEY_EMPTY, ptop);
                            // the lower words are requested right behind the claim, whatever it returns: one LDS round trip for a
                            // hit instead of two (LDS operations of a wave execute in order: a published top word seen by the
                            // compare-and-swap means the lower words read after it are the published ones)
                            // (two-word keys; with more lower words to fetch per probe the speculation gains nothing: k = 127, 352 -> 350 ms)
                            bool eq = true;
                            if (W == 2) {
                                CDBG_COMPILER_BARRIER();                        // (the reads must be issued AFTER the compare-and-swap)
                                eq = slotp[0] == can.w[0];
                            } else if (old == top) {
#pragma unroll
                                for (int i = 0; i < W - 1; ++i) eq &= (slotp[i] == can.w[i]);
                            }
                            const bool mine = old == KEY_EMPTY, same = (old == top) & eq, wait = old == ptop;
                            if (wait) CDBG_SPIN_YIELD();
                            if (mine) {
#pragma unroll
                                for (int i = 0; i < W - 1; ++i) slotp[i] = can.w[i];
                                CDBG_LDS_FENCE();
                                atomic_exch_u64(slotp + (W - 1), top);
                            }
                            is_new = is_new | mine; hit = mine | same;
                            const bool advance = !(hit | wait);
                            s = advance ? ((s + 1) & (TS - 1)) : s; probes += advance ? 1u : 0u;
                        } while (!hit && probes < 64u);
                    }
                    if (!hit) L.over = 1;                                      // table (nearly) full: not a one-pass partition
                    else {
                        atomic_add_u32(&L.cnt[s], 1u);
                        if (trav) atomic_or_u32(&L.cnt[s], CountList<W>::ON ? CF_TRAV16 : TRAV_FLAG);
                    }
                    if (CountList<W>::ON) slot_new = s;
                    }
                }
                const uint64_t nb = __ballot(is_new);
                if (SIFT && phase == 0) { n_fp += (uint32_t)__popcll(nb); continue; }
                if (CountList<W>::ON) {
                    // used-slot list: the wave's j-th new key leaves its slot in the upper half of count word wave * LCAP + j
                    const uint32_t pos = n_new + (uint32_t)__popcll(nb & (lane_le >> 1));
                    if (is_new && pos < LCAP) reinterpret_cast<uint16_t*>(L.cnt)[2u * (lbase + pos) + 1u] = (uint16_t)slot_new;
                }
                n_new += (uint32_t)__popcll(nb);
            }
            CDBG_WAVE_SYNC();                                                  // every lane has read the stage and the mask
            if (lane < MASKW && lane <= (int)((total + 63) >> 6)) { smask[lane] = 0; emask[lane] = 0; }   // hand the masks back clean
            CDBG_WAVE_SYNC();
            mbase += total;
        }
    }
    if (SIFT && phase == 0) {
        if (!A.issued) count_issue_ahead<W, NW, CAPPED>(P, A, wave, lane);
        if (lane == 0 && n_fp) atomic_add_u32(&L.fpfill, n_fp);
        CDBG_LDS_BARRIER();                                                       // ---- every fingerprint is in ----
        if (uni_u32(L.over) || uni_u32(L.fpfill) > (uint32_t)(FS - FS / 4)) { acc.last_fill = uni_u32(L.fpfill); return false; }   // uniform
    }
    }
    if (!A.issued) count_issue_ahead<W, NW, CAPPED>(P, A, wave, lane);                // (a wave without records of this partition)
    if (lane == 0 && n_new) atomic_add_u32(&L.fill[par], n_new);
    if (CountList<W>::ON && n_new > LCAP && lane == 0) L.over = 1;             // (more new keys than the wave's list holds: next tier)
    CDBG_FPH(2);
    CDBG_LDS_BARRIER();                                                           // ---- barrier A: all inserts done ----
    CDBG_FPH(3);
    const uint32_t need = uni_u32(L.fill[par]);
    if (W > 1) acc.last_fill = SIFT ? uni_u32(L.fpfill) : need;                // (what the admission rule of the caller learns from)
    if (uni_u32(L.over) || need > (uint32_t)(TS - TS / 4)) return false;       // uniform
    if (SIFT) { acc.dist += n_once; acc.occ += n_once; }
    if (need > chunk_left) {                                                   // uniform: new chunk (one device atomic per COUNT_CHUNK entries)
        if (tid == 0) L.cbase = atomic_add_u64(P.solid_cursor, (uint64_t)COUNT_CHUNK);
        CDBG_LDS_BARRIER();
        chunk_base = uni_u64(L.cbase); chunk_left = COUNT_CHUNK;
    }
    const uint64_t obase = chunk_base;
    const bool wr_ok = obase + need <= P.solid_cap;
    if (!wr_ok && tid == 0) *P.error = 1;
    if constexpr (CountList<W>::ON) {
        // sweep of the used slots only: every wave walks the list of the slots it claimed, 64 per round
        uint16_t* const h16 = reinterpret_cast<uint16_t*>(L.cnt);
        for (uint32_t j0 = 0; j0 < n_new; j0 += 64u) {                        // wave-uniform
            const uint32_t j = j0 + (uint32_t)lane;
            const bool valid = j < n_new;
            const uint32_t sl = valid ? (uint32_t)h16[2u * (lbase + j) + 1u] : 0u;
            const uint64_t key = L.keys[sl];
            const uint32_t cv = h16[2u * sl];
            const uint32_t cn = cv & ~CF_TRAV16; const bool trav = cv & CF_TRAV16;
            if (valid && !trav) { ++acc.dist; acc.occ += cn; }
            const bool solid = valid && cn >= P.amin;
            if (solid) { if (trav) ++acc.st; else ++acc.sh; }
            const uint64_t sb = __ballot(solid);
            uint32_t wbase = 0;
            if (sb) {                                                          // uniform
                if (lane == 0) wbase = atomic_add_u32(&L.wr[par], (uint32_t)__popcll(sb));
                wbase = wave_readlane_u32(wbase, 0);
            }
            if (solid && wr_ok) {
                const uint64_t o = obase + wbase + (uint32_t)__popcll(sb & (lane_le >> 1));
                P.solid_keys[o] = key;
                P.solid_cnt[o] = cn | (trav ? TRAV_FLAG : 0u);
            }
            if (valid) { L.keys[sl] = KEY_EMPTY; h16[2u * sl] = 0; }
        }
    } else
    // sweep: statistics, solid entries out, slots back to EMPTY.  Every thread owns TS / NT slots; all of their LDS reads
    // are issued first (one wait), the wave's solid entries get their places from ONE returning LDS atomic.
    {
        constexpr int SPT = TS / NT;
        static_assert(TS % NT == 0, "slots per thread");
        uint64_t topw[SPT]; uint32_t cv[SPT];
#pragma unroll
        for (int j = 0; j < SPT; ++j) { const uint32_t sl = (uint32_t)tid + (uint32_t)j * NT; topw[j] = L.keys[(uint64_t)sl * W + (W - 1)]; cv[j] = L.cnt[sl]; }
        uint32_t my_solid = 0;
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const bool used_j = topw[j] != KEY_EMPTY;
            const uint32_t cn = cv[j] & ~TRAV_FLAG; const bool trav = cv[j] & TRAV_FLAG;
            if (used_j && !trav) { ++acc.dist; acc.occ += cn; }
            if (used_j && cn >= P.amin) { ++my_solid; if (trav) ++acc.st; else ++acc.sh; }
        }
        const uint32_t incl = wave_incl_sum_u32(my_solid);
        const uint32_t wave_total = wave_readlane_u32(incl, 63);
        uint32_t wbase = 0;
        if (lane == 0 && wave_total) wbase = atomic_add_u32(&L.wr[par], wave_total);
        wbase = wave_readlane_u32(wbase, 0);
        uint64_t o = obase + wbase + (incl - my_solid);
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const uint32_t sl = (uint32_t)tid + (uint32_t)j * NT;
            if (topw[j] == KEY_EMPTY) continue;
            if ((cv[j] & ~TRAV_FLAG) >= P.amin && wr_ok) {
#pragma unroll
                for (int i = 0; i < W - 1; ++i) P.solid_keys[o * W + i] = L.keys[(uint64_t)sl * W + i];
                P.solid_keys[o * W + (W - 1)] = topw[j];
                P.solid_cnt[o] = cv[j];
                ++o;
            }
            L.keys[(uint64_t)sl * W + (W - 1)] = KEY_EMPTY;
            L.cnt[sl] = 0;
        }
    }
    if constexpr (SIFT) {                                                      // the fingerprints go: 16 bytes per store
        uint4 z; z.x = 0; z.y = 0; z.z = 0; z.w = 0;
        for (int i = tid; i < FS / 4; i += NT) reinterpret_cast<uint4*>(L.fp)[i] = z;
        if (tid == 0) L.fpfill = 0;
    }
    if (CountShared<W>::ON && FS == 0 && shared_done) {                        // (uniform) the pooled masks go back clean
        for (int i = tid; i < (int)(COUNT_SHARED_MEMBERS / 64) + 1 && i < NW * MASKW; i += NT) { L.smask[i] = 0; L.emask[i] = 0; }
    }
    if (tid == 0) { L.fill[par ^ 1u] = 0; L.wr[par ^ 1u] = 0; }               // the next partition's counters (nobody touches them now)
    CDBG_FPH(4);
    CDBG_LDS_BARRIER();                                                           // ---- barrier B: sweep done, table clean ----
    CDBG_FPH(5);
    const uint32_t used = uni_u32(L.wr[par]);
    chunk_base += used; chunk_left -= used;                                    // the unused tail of the reservation stays in the chunk
    if (tid == 0) { P.seg_off[rg.p] = obase; P.seg_n[rg.p] = used; }
    return true;
}

// the rare path's table reset, kept out of line so that its address arithmetic is not hoisted into (and spilled
// around) the partition loop
template <int W, int TS, int NT, int FS>
CDBG_NOINLINE CDBG_DEV_NOINL void count_fast_clear(CountFastLds<W, TS, NT, FS>& L) {
    const int tid = threadIdx.x;
    for (uint32_t i = tid; i < (uint32_t)(FS > 0 ? FS : 1); i += NT) L.fp[i] = 0;
    if (tid == 0) L.fpfill = 0;
    for (uint32_t i = tid; i < (uint32_t)TS; i += NT) { L.keys[(uint64_t)i * W + (W - 1)] = KEY_EMPTY; L.cnt[i] = 0; }
    for (uint32_t i = tid; i < (uint32_t)((NT / 64) * CountGeom<W>::MASKW); i += NT) { L.smask[i] = 0; L.emask[i] = 0; }
    if (tid == 0) { L.fill[0] = L.fill[1] = 0; L.wr[0] = L.wr[1] = 0; L.over = 0; }
}

// persistent workgroups, grid-stride over partitions; statistics are accumulated in registers and published once
// per wave when the kernel ends
// (two-word k-mers: promised 4 waves per SIMD the kernel took 90 VGPRs and only two of the three workgroups the LDS has
//  room for were resident; promised 6 it takes 80 and no scratch: count 218 -> 180 ms at the config-4 share)
#ifndef CDBG_CF_WAVES2
#define CDBG_CF_WAVES2 6
#endif
template <int W, int TS, int NT, int CAPPED, int FS = 0>
__global__ void __launch_bounds__(NT, lds_waves_per_simd(sizeof(CountFastLds<W, TS, NT, FS>), NT, W == 1 ? 6 : W == 2 ? CDBG_CF_WAVES2 : 3)) k_count_fast(CountFastParams FP) {   // waves per SIMD that the LDS tables allow: 3 workgroups x 2 waves (W = 1)
    CDBG_SHARED CountFastLds<W, TS, NT, FS> L;
    const CountParams& P = FP.c;
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = (int)uni_u32((uint32_t)tid >> 6);
    CountAcc ca; ca.dist = 0; ca.sh = 0; ca.st = 0; ca.occ = 0; ca.last_fill = 0;
#if defined(CDBG_PROFILE_PHASES) && !defined(CDBG_HOSTSIM)
    for (int i = 0; i < 8; ++i) ca.ph[i] = 0;
    ca.t_prev = clock64();
#endif
    uint64_t chunk_base = 0; uint32_t chunk_left = 0;
    count_fast_clear<W, TS, NT, FS>(L);
    CDBG_LDS_BARRIER();
    // software pipeline: range words two partitions ahead, this wave's records one partition ahead
    const uint32_t stride = gridDim.x;
    CountRange rg_cur = count_raw_resolve<CAPPED>(P, blockIdx.x, count_raw_load<CAPPED>(P, blockIdx.x));
    CountSet<W, CAPPED> S0, S1;
    S0.raw = count_raw_load<CAPPED>(P, blockIdx.x + stride);
    { uint64_t w0, w1; count_share<W, NW>(rg_cur, wave, w0, w1); count_load_chunk<W>(P, w0, w1, lane, S0.R); }
    uint32_t par = 0, misses = 0, deferred = 0;          // misses: consecutive partitions that did not fit one pass; deferred: partitions sent on untried since
    // Admission by predicted fill.  A partition that overflows the table costs its whole insert phase for nothing, and well before
    // that the probe sequences of a table more than half full are long, while the next tier's table is twice the size: measured at the
    // config-5 share (k = 127: 6 records of 53 members per partition on average, a heavy tail of partitions with two minimizer loci)
    // count 332 -> 256 ms when partitions of >= 24 records skip this tier, at the config-4 share (k = 55) 154 -> 145 ms at >= 175 records;
    // both are the same rule in units of the table: predicted distinct k-mers > 0.47 / 0.69 of the slots.  The prediction: the
    // partition's size -- its member k-mers where every wave sees all records anyway (CountBal, k >= 64: 1 .. 122 members per
    // record), its records otherwise -- times the distinct k-mers per unit of size that this workgroup has seen so far (running
    // average over its partitions, 8 fractional bits).
    uint32_t dpu_q8 = 0;
    auto one_partition = [&](CountSet<W, CAPPED>& cur, CountSet<W, CAPPED>& nxt, const uint32_t item) {
        CountAhead<W, CAPPED> A;
        A.cur = &cur; A.nxt = &nxt; A.item_nxt = item + stride; A.item_nn = item + 2 * stride; A.issued = false;
        if (rg_cur.n) {                                  // uniform
            bool done = false;
            // after four misses in a row (an input of mostly distinct k-mers) the workgroup stops trying -- but looks again every
            // 16th partition: one heavy locus must not send the rest of the workgroup's stride to the slower tiers
            bool try_fast = misses < 4;
            if (!try_fast && ++deferred >= 16u) { try_fast = true; deferred = 0; }
            uint32_t size = rg_cur.n;
            if (W > 1 && FP.skip_fill_q8) {              // (uniform; one-word k-mers never use the rule)
                if (CountBal<W>::ON && rg_cur.n <= 64u) size = wave_readlane_u32(wave_incl_sum_u32((uint32_t)cur.R.n()), 63);   // (lanes without a record hold zeros)
                else if (CountBal<W>::ON) size = rg_cur.n * 64u;   // (more than one chunk of records: big; any large number will do)
                if ((((uint64_t)size * dpu_q8) >> 8) * 256u > (uint64_t)(FS > 0 ? FS : TS) * FP.skip_fill_q8) try_fast = false;
            }
            if (try_fast && rg_cur.n < FP.fast_max_records) {   // (a partition that could carry a count to the 31-bit ceiling goes to the saturating kernels)
                done = count_partition_fast<W, TS, NT, CAPPED, FS>(P, L, rg_cur, A, par, chunk_base, chunk_left, ca);
                if (W > 1 && FP.skip_fill_q8) {          // (a partition that did not fit still says: at least this many distinct k-mers)
                    const uint32_t sample = (uint32_t)((float)ca.last_fill * 256.0f / (float)(size ? size : 1u));
                    dpu_q8 = dpu_q8 ? (uint32_t)((int32_t)dpu_q8 + (((int32_t)sample - (int32_t)dpu_q8) >> 4)) : sample;
                }
                if (done) { par ^= 1u; misses = 0; deferred = 0; }
                else {                                   // leave a clean table and known counters behind
                    CDBG_LDS_BARRIER();
                    count_fast_clear<W, TS, NT, FS>(L);
                    par = 0; ++misses;
                    CDBG_LDS_BARRIER();
                }
            }
            if (!done && tid == 0) {
                const uint32_t i = atomic_add_u32(FP.retry_count, 1u);
                FP.retry_list[i] = rg_cur.p;
            }
        }
        if (!A.issued) count_issue_ahead<W, NW, CAPPED>(P, A, wave, lane);
        rg_cur = A.rg_nxt;
    };
    for (uint32_t item = blockIdx.x; item < P.n_items;) {
        one_partition(S0, S1, item); item += stride;
        if (item >= P.n_items) break;
        one_partition(S1, S0, item); item += stride;
    }
#if defined(CDBG_PROFILE_PHASES) && !defined(CDBG_HOSTSIM)
    if (lane == 0 && (wave == 0 || wave == NW - 1)) for (int i = 0; i < 6; ++i) atomic_add_u64(&P.stats[8 + (wave ? 8 : 0) + i], ca.ph[i]);
#endif
    {   // statistics: one atomic per wave and counter
        uint64_t d = wave_sum_u64(ca.dist), o = wave_sum_u64(ca.occ), h = wave_sum_u64(ca.sh), t = wave_sum_u64(ca.st);
        if (lane == 0) {
            if (d) atomic_add_u64(&P.stats[0], d);
            if (o) atomic_add_u64(&P.stats[1], o);
            if (h) atomic_add_u64(&P.stats[2], h);
            if (t) atomic_add_u64(&P.stats[3], t);
        }
    }
}

}  // namespace cdbg
