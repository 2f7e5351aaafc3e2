// k_compact_wave.h -- stage 2, first tier: ONE WAVE per minimizer bucket.
//
// Same job, same outputs and same junction rules as k_compact (k_compact.h; bcalm2 / graph3 of gatb-core's
// bcalm_algo; unitig definition
// /root/reference/bidirected-graphs-in-bcalm2/bidirected-graphs-in-bcalm2.md:83-88).
//
// Why a second formulation: at sequencing depth a bucket holds ~150 solid entries.  k_compact gives it a
// 256-thread workgroup and 13 workgroup barriers; its phase profile (profiles/r02_*) is latency, not work:
// most lanes idle in every phase and every barrier waits for the slowest wave.  Here a bucket belongs to one
// wavefront: a phase is 3-5 wave iterations, phases are separated by wave-level ordering only (the LDS executes a
// wave's instructions in order), two buckets are in flight per workgroup and 16-24 per CU.  Per-bucket state is
// entry-indexed (<= TSW/2 entries): the k-mers in entry order, 16-bit link / piece words and a table of 4-byte
// slots in which every k-mer end registers at its junction (cw_jt_*: the classification is junction-centric, the
// workgroup tiers of k_compact.h still probe for successors) -- 9 KB of LDS per wave for one-word k-mers.
// Buckets with more entries than TSW/2 (or more junctions than the table takes) are deferred: to a second launch of
// this kernel with a table twice the size (W >= 2), then to the workgroup-per-bucket tiers of k_compact.
#pragma once
#include "k_compact.h"

namespace cdbg {

constexpr int CW_THREADS = 128;                         // 2 independent waves per workgroup (LDS granularity: more waves per CU)
constexpr uint32_t CW_PIECE_CHUNK = 256, CW_BASES_CHUNK = 1u << 14, CW_GLOG_CHUNK = 512;   // per-WAVE reservations (one device atomic each)
constexpr uint32_t CW_BATCH = 8;                         // buckets handed out per queue ticket
constexpr uint16_t CWL_CONF = 1u << 14, CWL_POSTED = 1u << 15;
constexpr uint16_t CWN_NONE = 0xFFFFu, CWN_FOREIGN = 0xFFFEu;

template <int TSW> struct CwSlot {                      // field layout of a junction-table slot (see cw_jt_register)
    static constexpr uint32_t EB = TSW > 512 ? 10u : 9u, SIDE_BIT = 6u + 2u * EB, TAG_MASK = ~((2u << SIDE_BIT) - 1u);
};
template <int W, int TSW>
struct CompactWaveLds {                                 // one per wave
    static constexpr int EMAX = TSW / 2;
    static_assert(TSW <= 512 || W <= 4, "the 1024-slot wave tier: k <= 127 (16-bit base offsets: EMAX * k = 512 * 127 < 65536)");
    uint64_t ekeys[EMAX * W];                           // the bucket's k-mers in entry order: word i of entry e at [i * EMAX + e]
    uint32_t jt[TSW];                                   // junction table: one slot per junction that an end of the bucket registers at (cw_jt_*);
                                                        // once the links are known, its memory holds the terminal ends of home entries (walk 1 work list)
    uint32_t cnt[EMAX];                                 // count | TRAV_FLAG; after walk 2: (byte offset << 1) | strand
    uint16_t lnk[2 * EMAX];                             // per end (2 * entry + end): note, then link word
    uint16_t pdesc[EMAX], pn[EMAX], pb[EMAX];           // pieces: start end (bit 15: cyclic), k-mers, relative base offset
    uint8_t vis[EMAX];                                  // bit 0 visited, bit 1 traveller, bit 2 / 3 owns the junction at the left / right end
    uint32_t np, nb, nb2, nopen, nconf, lw, ncyc, pad, ncov;   // pad: number of terminals
};

template <int W, int TSW>
CDBG_DEV Kmer<W> cw_key(const CompactWaveLds<W, TSW>& L, uint32_t e) {
    Kmer<W> r;
#pragma unroll
    for (int i = 0; i < W; ++i) r.w[i] = L.ekeys[i * (TSW / 2) + e];
    return r;
}
// ---- the junction table -------------------------------------------------------------------------------------------
// Whether a junction is 1-in/1-out is a fact about the JUNCTION: it holds exactly when one end of the bucket's k-mers
// reaches it from each side.  Every end whose junction the bucket owns registers there once (one hash insert of the
// canonical (k-1)-mer) instead of probing the k-mer table for its four possible successors (four canonical k-mers, four
// hashes, four probe sequences per end: a third of this kernel's time).  A slot does not hold the (k-1)-mer: it holds a
// 7-bit tag of its hash and the end that claimed the slot, and a later end with the same tag recomputes the claimer's
// junction from the entry-ordered k-mers and compares all of it -- exact, at 4 bytes per slot for any k.
//   bits [0,3) / [3,6)   ends registered on side 0 / 1 (side 0: the end's outgoing (k-1)-suffix IS the canonical junction)
//   bits [6,15) / [15,24) the first end registered on side 0 / 1      bit 24  the side of the end that claimed the slot
//   bits [25,32)          tag            (a claimed slot is never 0: the claimer's side counts 1)
// (a table of 1024 slots -- buckets of up to 512 entries, round 5: the second one-wave tier of one-word k-mers -- has 10-bit end ids:
//  [6,16) / [16,26), side in bit 26, a 5-bit tag: CwSlot)
// Even k: a k-mer that is its own reverse complement reaches the junction with both of its ends from the same side, so
// that side counts 2 and the junction is never 1-in/1-out (the two edges (s,+) / (s,-) of the overlap table, .md:41-46).
// Odd k: a junction that is its own reverse complement has one side only -- never 1-in/1-out either (every end there
// sees its own node's reverse complement among its successors).
// j: the junction as the end leaves through it; r: its reverse complement; returns the canonical one of the two and the side
template <int W>
CDBG_DEV Kmer<W> cw_junction_of(const Kmer<W>& x, uint32_t end, int k, uint32_t& side, Kmer<W>& j, Kmer<W>& r) {
    Kmer<W> u, ur; orient_pair<W>(x, end, k, u, ur);
    j = suffix_km1<W>(u, k); r = ur.shr(2);
    const bool rev = r < j;
    side = rev ? 1u : 0u;
    return rev ? r : j;
}
// registers end `it` (side known) at junction jc; returns the slot, or NONE32 when the table is too full (the caller defers the bucket)
template <int W, int TSW>
CDBG_DEV uint32_t cw_jt_register(CompactWaveLds<W, TSW>& L, const Kmer<W>& jc, const Kmer<W>& j, const Kmer<W>& r, uint32_t side, uint32_t it, int k) {
    constexpr int LOG = TSW == 1024 ? 10 : TSW == 512 ? 9 : TSW == 256 ? 8 : TSW == 128 ? 7 : -1;
    static_assert(LOG > 0, "wave table size (end ids are 9- or 10-bit fields of a slot)");
    constexpr uint32_t EB = CwSlot<TSW>::EB, SIDE_BIT = CwSlot<TSW>::SIDE_BIT, TAG_MASK = CwSlot<TSW>::TAG_MASK;
    const uint32_t h = jc.hash_lds();
    uint32_t s = h >> (32 - LOG);
    const uint32_t tag = (h << LOG) & TAG_MASK;                  // the 7 (5: 1024 slots) hash bits below the slot index
    const uint32_t mine = tag | (side << SIDE_BIT) | (it << (6u + EB * side)) | (1u << (3u * side));
    uint32_t probes = 0, res = NONE32; bool done = false;
#pragma clang loop unroll(disable)
    do {
        const uint32_t old = atomic_cas_u32(&L.jt[s], 0u, mine);
        bool same = false;
        if (old != 0u && (old & TAG_MASK) == tag) {               // same tag: is it the same junction?  ask the end that claimed the slot
            // (no reverse complement needed: the stored k-mer x' of that end touches its junction with its (k-1)-suffix -- right
            //  end -- or its (k-1)-prefix -- left end -- as written, and the junctions are equal exactly when that is j or rc(j))
            const uint32_t cs = (old >> SIDE_BIT) & 1u, rid = (old >> (6u + EB * cs)) & ((1u << EB) - 1u);
            const Kmer<W> xr = cw_key<W, TSW>(L, rid >> 1);
            const Kmer<W> c = (rid & 1u) == END_RIGHT ? suffix_km1<W>(xr, k) : prefix_km1<W>(xr, k);
            same = (c == j) | (c == r);
        }
        if (same) {
            const uint32_t before = atomic_add_u32(&L.jt[s], 1u << (3u * side));
            if (((before >> (3u * side)) & 7u) == 0u) atomic_or_u32(&L.jt[s], it << (6u + EB * side));   // the first end on this side
        }
        done = (old == 0u) | same;
        res = done ? s : res;
        s = done ? s : ((s + 1) & (TSW - 1)); probes += done ? 0u : 1u;
    } while (!done && probes < 48u);
    return res;
}

struct CompactWaveParams {
    CompactParams c;
    uint32_t n_buckets;
    uint32_t* queue;                                    // next bucket to hand out (zeroed by the host)
};
// per-wave chunk of one output array (wave-uniform registers)
struct CwChunk { uint64_t base; uint32_t left; };
CDBG_DEV uint64_t cw_reserve(CwChunk& ch, uint64_t* cursor, uint32_t need, uint32_t chunk, int lane) {
    if (need > chunk) {                                  // (cannot happen for the wave tier's bounds; kept for safety)
        uint64_t b = 0;
        if (lane == 0) b = atomic_add_u64(cursor, (uint64_t)need);
        return uni_u64(b);
    }
    if (need > ch.left) {
        uint64_t b = 0;
        if (lane == 0) b = atomic_add_u64(cursor, (uint64_t)chunk);
        ch.base = uni_u64(b); ch.left = chunk;
    }
    const uint64_t r = ch.base; ch.base += need; ch.left -= need;
    return r;
}

// the solid entries of one bucket as loaded, NPER per lane (entry e = lane + 64 j)
// (measured and discarded in round 3: unconditional loads from a clamped index, so that the waits for the current bucket's
//  entries become s_waitcnt vmcnt(N) instead of vmcnt(0) and the request for the next bucket stays in flight through the
//  load and classify phases -- the ISA showed exactly that, the times did not move: 40.0 / 67.3 / 50.1 -> 40.5 / 67.5 / 49.0 ms
//  at k = 31 / 55 / 127.  16 - 20 waves per CU cover a bucket's exposed memory latency; the waves wait on LDS round trips.)
template <int W, int TSW>
struct CwEntries {
    static constexpr int NPER = TSW / 2 / 64 > 0 ? TSW / 2 / 64 : 1;
    uint64_t key[NPER * W]; uint32_t cnt[NPER];
};
template <int W, int TSW>
CDBG_DEV void cw_load_entries(const CompactParams& P, uint64_t so, uint32_t E, int lane, CwEntries<W, TSW>& X) {
    constexpr int NPER = CwEntries<W, TSW>::NPER;
#pragma unroll
    for (int j = 0; j < NPER; ++j) {
        const uint32_t e = (uint32_t)lane + 64u * (uint32_t)j;
#pragma unroll
        for (int i = 0; i < W; ++i) X.key[j * W + i] = 0;
        X.cnt[j] = 0;
        if (e < E && E <= (uint32_t)(TSW / 2)) {
#pragma unroll
            for (int i = 0; i < W; ++i) X.key[j * W + i] = P.solid_keys[(so + e) * W + i];
            X.cnt[j] = P.solid_cnt[so + e];
        }
    }
}
// One bucket.  X: its entries (requested one bucket ago); E, so: its segment.
#if defined(CDBG_PROFILE_PHASES) && !defined(CDBG_HOSTSIM)
#define CDBG_WPH(i) do { const uint64_t t_ = clock64(); acc[4 + (i)] += t_ - acc[15]; acc[15] = t_; } while (0)
#else
#define CDBG_WPH(i) do { } while (0)
#endif
template <int W, int TSW>
CDBG_DEV void compact_bucket_wave(const CompactParams& P, CompactWaveLds<W, TSW>& L, const uint32_t p, const uint32_t E, const int lane,
                                  const CwEntries<W, TSW>& X, CwChunk& pc, CwChunk& bc, CwChunk& lc, uint64_t (&acc)[16]) {
    CDBG_WPH(0);
    constexpr int NPER = CwEntries<W, TSW>::NPER;
    const int k = P.k;
    const uint32_t pg = (p << P.rank_bits) | (uint32_t)P.rank;
    if (lane == 0) { L.np = 0; L.nb = 0; L.nb2 = 0; L.nopen = 0; L.nconf = 0; L.lw = 0; L.ncyc = 0; L.pad = 0; L.ncov = 0; }

    // ---- load the bucket: home + traveller solid k-mers (the table is empty: the previous bucket emptied its slots) ----
    uint32_t n_home = 0;
#pragma unroll
    for (int j = 0; j < NPER; ++j) {
        const uint32_t e = (uint32_t)lane + 64u * (uint32_t)j;
        if (e < E) {
            Kmer<W> x;
#pragma unroll
            for (int i = 0; i < W; ++i) x.w[i] = X.key[j * W + i];
            // which of the two junctions this bucket owns came along in the top bits of the key (KEY_FOREIGN_*, k_count.h:
            // the scan knows; recomputing both junction minimizers here took a quarter of this kernel at k = 55)
            const uint64_t fl = x.w[W - 1] & KEY_FLAGS;
            x.w[W - 1] &= ~KEY_FLAGS;
            const uint32_t cv = X.cnt[j];
#pragma unroll
            for (int i = 0; i < W; ++i) L.ekeys[i * (TSW / 2) + e] = x.w[i];
            L.cnt[e] = cv;
            L.vis[e] = (uint8_t)(((cv & TRAV_FLAG) ? 2u : 0u) | ((fl & KEY_FOREIGN_L) ? 0u : 4u) | ((fl & KEY_FOREIGN_R) ? 0u : 8u));
#ifdef CDBG_HOSTSIM
            {   // (the simulator build checks every flag against the definition)
                uint32_t gl, gr; kmer_junction_mins<W>(x, k, P.m, gl, gr);
                if (!P.split && ((part_of(gl, P.log_np) == pg) != !(fl & KEY_FOREIGN_L) || (part_of(gr, P.log_np) == pg) != !(fl & KEY_FOREIGN_R))) *P.error = 9;   // (sub-buckets of k_split.h own junctions by sub-minimizer)
            }
#endif
            if (!(cv & TRAV_FLAG)) ++n_home;
        }
    }
    n_home = wave_readlane_u32(wave_incl_sum_u32(n_home), 63);
    CDBG_WAVE_SYNC();
    CDBG_WPH(1);

    // ---- classify: every end whose junction this bucket owns registers at the junction (cw_jt_*) ... ----
    for (uint32_t it = lane; it < 2 * E; it += 64) {
        const uint32_t e = it >> 1, end = it & 1u;
        uint32_t where = 0xFFFFu;                        // junction owned elsewhere: glue decides
        if ((L.vis[e] >> (2 + end)) & 1u) {
            uint32_t side; Kmer<W> j, r;
            const Kmer<W> jc = cw_junction_of<W>(cw_key<W, TSW>(L, e), end, k, side, j, r);
            const uint32_t s = cw_jt_register<W, TSW>(L, jc, j, r, side, it, k);
            if (s == NONE32) L.pad = 1u;                 // (table too full: the bucket goes to the next tier)
            where = s == NONE32 ? 0xFFFFu : (s | (side << 15));
        }
        L.lnk[it] = (uint16_t)where;
    }
    CDBG_WAVE_SYNC();
    if (uni_u32(L.pad) != 0u) {                          // (uniform) some end found no slot within the probe bound: nothing was written yet
        for (uint32_t i = lane; i < (uint32_t)TSW; i += 64) L.jt[i] = 0u;
        if (lane == 0) { const uint32_t i_ = atomic_add_u32(P.big_count, 1u); P.big_list[i_] = p; }
        CDBG_WAVE_SYNC();
        return;
    }
    CDBG_WPH(2);
    // ---- ... and reads there whether it is the only end on its side and which end is the only one on the other side: a
    // junction is 1-in/1-out exactly then (both ends read the same counts, so the partnership is mutual by construction).
    // Terminal ends of home entries go on a list. ----
    for (uint32_t it = lane; it < 2 * E; it += 64) {
        const uint32_t e = it >> 1;
        const uint32_t w = L.lnk[it];
        const bool home = !(L.cnt[e] & TRAV_FLAG);
        uint32_t link = LNK_DEAD; bool conf = false;
        if (w == 0xFFFFu) link = LNK_OPEN;
        else {
            const uint32_t side = w >> 15;
            const uint32_t v = L.jt[w & 0x7FFFu];
            const uint32_t own = (v >> (3u * side)) & 7u, opp = (v >> (3u * (1u - side))) & 7u;
            const uint32_t yy = (v >> (6u + CwSlot<TSW>::EB * (1u - side))) & ((1u << CwSlot<TSW>::EB) - 1u), y = yy >> 1, ye = yy & 1u;
            if (own == 1u && opp == 1u && y != e) {      // (never through the node's own other end)
                const bool yhome = !(L.cnt[y] & TRAV_FLAG);
                if (home && yhome) link = LNK_INTERNAL | (ye << 2) | (y << 3);
                else {
                    // 1-1 junction with a traveller on at least one side: confirm it for glue, once (a home end is open and
                    // posted anyway: its confirmation rides on that record)
                    if (home || (!yhome && e < y)) { conf = true; if (!home) atomic_add_u32(&L.nconf, 1u); }
                    if (home) link = LNK_OPEN;
                }
            }
        }
        L.lnk[it] = (uint16_t)((home ? link : LNK_DEAD) | (conf ? CWL_CONF : 0u));
    }
    CDBG_WAVE_SYNC();
    // terminal ends of home entries: the walk-1 work list, in the memory of the junction table (dead from here on: 1 KB per wave
    // less is a fifth more waves per CU for one-word k-mers)
    uint16_t* const term = reinterpret_cast<uint16_t*>(L.jt);
    for (uint32_t it = lane; it < 2 * E; it += 64) {
        if (!(L.cnt[it >> 1] & TRAV_FLAG) && (L.lnk[it] & 3u) != LNK_INTERNAL) {
            const uint32_t ti = atomic_add_u32(&L.pad, 1u);
            term[ti] = (uint16_t)it;
        }
    }
    CDBG_WAVE_SYNC();

    CDBG_WPH(3);
    // ---- walk 1: every terminal end measures its piece; the smaller terminal id registers it ----
    const uint32_t nterm = uni_u32(L.pad);
    for (uint32_t ti = lane; ti < nterm; ti += 64) {
        const uint32_t it = term[ti];
        uint32_t cur = it >> 1, ex = (it & 1u) ^ 1u, n = 1;
        for (;;) {
            const uint32_t l = L.lnk[cur * 2 + ex];
            if ((l & 3u) != LNK_INTERNAL) break;
            cur = (l >> 3) & 0x3FFu; ex = ((l >> 2) & 1u) ^ 1u; ++n;
        }
        const uint32_t other = cur * 2 + ex;
        if (it <= other) {
            const uint32_t li = atomic_add_u32(&L.np, 1u);
            L.pdesc[li] = (uint16_t)it; L.pn[li] = (uint16_t)n;
            atomic_add_u32(&L.nb, n + (uint32_t)k - 1u);
            atomic_add_u32(&L.ncov, n);
            const uint32_t no = ((L.lnk[it] & 3u) == LNK_OPEN ? 1u : 0u) + ((L.lnk[other] & 3u) == LNK_OPEN ? 1u : 0u);
            if (no) atomic_add_u32(&L.nopen, no);
        }
    }
    CDBG_WAVE_SYNC();
    for (uint32_t i = lane; i < (uint32_t)TSW; i += 64) L.jt[i] = 0u;   // the work list is consumed: the table goes back empty, for the next bucket
    // ---- closed chains entirely inside the bucket (isolated cycles): only when the linear pieces do not cover every
    // home entry (rare).  Mark what the pieces cover, then cut each remaining cycle at its smallest entry. ----
    if (uni_u32(L.ncov) != n_home) {
        const uint32_t np0 = uni_u32(L.np);
        for (uint32_t li = lane; li < np0; li += 64) {
            const uint32_t start = L.pdesc[li];
            uint32_t cur = start >> 1, ex = (start & 1u) ^ 1u;
            for (uint32_t t = 0, n = L.pn[li]; t < n; ++t) {
                L.vis[cur] = L.vis[cur] | 1u;
                if (t + 1 < n) { const uint32_t l = L.lnk[cur * 2 + ex]; cur = (l >> 3) & 0x3FFu; ex = ((l >> 2) & 1u) ^ 1u; }
            }
        }
        CDBG_WAVE_SYNC();
        for (uint32_t e = lane; e < E; e += 64) {
            if (L.vis[e] & 3u) continue;                 // traveller, or part of a linear piece
            uint32_t cur = e, ex = END_RIGHT, n = 0; bool is_min = true;
            do {
                const uint32_t l = L.lnk[cur * 2 + ex];
                cur = (l >> 3) & 0x3FFu; ex = ((l >> 2) & 1u) ^ 1u; ++n;
                if (cur < e) is_min = false;
            } while (cur != e);
            if (is_min) {
                const uint32_t li = atomic_add_u32(&L.np, 1u);
                L.pdesc[li] = (uint16_t)((e * 2 + END_LEFT) | 0x8000u);   // cyclic piece starting at e, walking right
                L.pn[li] = (uint16_t)n;
                atomic_add_u32(&L.nb, n + (uint32_t)k - 1u);
                atomic_add_u32(&L.ncyc, 1u);
            }
        }
        CDBG_WAVE_SYNC();
    }

    CDBG_WPH(4);
    // ---- output space: piece ids, base bytes, glue-log records from this wave's chunks ----
    uint32_t np = uni_u32(L.np);
    const uint32_t nb = uni_u32(L.nb), nconf = uni_u32(L.nconf), nlog = nconf + uni_u32(L.nopen);
    const uint64_t pbase = cw_reserve(pc, P.piece_cursor, np, CW_PIECE_CHUNK, lane);
    const uint64_t bbase = cw_reserve(bc, P.bases_cursor, nb, CW_BASES_CHUNK, lane);
    uint64_t lbase = cw_reserve(lc, P.glog_cursor, nlog, CW_GLOG_CHUNK, lane);
    bool log_ok = true;
    if (pbase + np > P.piece_cap || bbase + nb > P.bases_cap) { if (lane == 0) *P.error = 3; np = 0; }
    if (lbase + nlog > P.glog_cap) { if (lane == 0) *P.error = 5; np = 0; lbase = 0; log_ok = false; }   // never write past the log

    // ---- glue log, part 1: CONFIRM records of junctions whose two k-mers are both travellers here ----
    if (log_ok && nconf) {
        for (uint32_t it = lane; it < 2 * E; it += 64) {
            const uint32_t l = L.lnk[it];
            if (!(l & CWL_CONF) || (l & 3u) == LNK_OPEN) continue;   // open home ends carry their confirmation themselves
            const uint64_t o = lbase + atomic_add_u32(&L.lw, 1u);
            const Kmer<W> jc = canon_junction_at<W>(cw_key<W, TSW>(L, it >> 1), it & 1u, k);
            glue_record_put<W>(P, o, jc, GTAG_CONFIRM);
        }
    }
    CDBG_WAVE_SYNC();                                    // walk 2 rewrites the link words this pass reads
    CDBG_WPH(5);
    // ---- walk 2: one lane per piece hands every k-mer its byte offset + strand (stored over the k-mer's count, which
    // is summed here), marks the open ends and writes the piece's first k-1 bases ----
    uint8_t* const out = P.piece_bases + bbase;
    for (uint32_t li = lane; li < np; li += 64) {
        const uint32_t d = L.pdesc[li];
        const bool cyclic = d & 0x8000u;
        const uint32_t start = d & 0x7FFFu;
        const uint32_t s0 = start >> 1, e0 = start & 1u, n = L.pn[li];
        const uint32_t rel = atomic_add_u32(&L.nb2, n + (uint32_t)k - 1u);
        uint64_t kc = 0;
        uint32_t cur = s0, ex = e0 ^ 1u;
        for (uint32_t t = 0; t < n; ++t) {
            const uint32_t ab = L.cnt[cur] & ~TRAV_FLAG;
            const uint32_t l = L.lnk[cur * 2 + ex];
            kc += (uint64_t)ab;
            if (P.piece_ab) P.piece_ab[bbase + rel + (uint32_t)k - 1u + t] = ab;
            L.cnt[cur] = ((rel + (uint32_t)k - 1u + t) << 1) | (ex == END_RIGHT ? 0u : 1u);
            if (t + 1 < n) { cur = (l >> 3) & 0x3FFu; ex = ((l >> 2) & 1u) ^ 1u; }
        }
        const uint64_t pid = pbase + li;
        P.piece_n[pid] = n; P.piece_kc[pid] = kc; P.piece_boff[pid] = bbase + rel;
        if (!cyclic) {
            const uint32_t il = s0 * 2 + e0, ir = cur * 2 + ex;
            const uint32_t ll = L.lnk[il], lr = L.lnk[ir];
            if ((ll & 3u) == LNK_OPEN) L.lnk[il] = (uint16_t)(CWL_POSTED | (ll & CWL_CONF) | (li * 2u + 0u));
            if ((lr & 3u) == LNK_OPEN) L.lnk[ir] = (uint16_t)(CWL_POSTED | (lr & CWL_CONF) | (li * 2u + 1u));
        }
        L.pb[li] = (uint16_t)rel;                        // (bucket-relative: at most EMAX * k bytes)
        if (k < 9) {                                     // (fewer than 8 prefix bases: byte stores, here)
            const Kmer<W> x0 = cw_key<W, TSW>(L, s0);
            const Kmer<W> xo = ((e0 ^ 1u) == END_RIGHT) ? x0 : x0.rc(k);
            kmer_prefix_ascii<W>(out + rel, xo, k, k - 1);
        }
    }
    CDBG_WAVE_SYNC();
    // the first k-1 bases of every piece -- the start k-mer, read leaving through the far end of the start terminal -- one lane
    // per 8 bases (an unaligned 8-byte store; the last chunk of a piece overlaps the one before): with one lane per piece the
    // 16 stores of a 127-mer's prefix were serial work of a few lanes (walk 2: a third of this kernel at k = 127)
    if (k >= 9) {
        const uint32_t nchunk = ((uint32_t)k - 1u + 7u) >> 3;
        for (uint32_t item = lane; item < np * nchunk; item += 64) {
            const uint32_t li = item / nchunk, c = item - li * nchunk;
            const uint32_t d = L.pdesc[li], start = d & 0x7FFFu;
            const Kmer<W> x0 = cw_key<W, TSW>(L, start >> 1);
            const int i = (int)(8u * c + 8u > (uint32_t)k - 1u ? (uint32_t)k - 9u : 8u * c);
            const uint64_t v = ((start & 1u) ^ 1u) == END_RIGHT ? kmer_ascii8<W>(x0, k, i) : kmer_rc_ascii8<W>(x0, k, i);
            st_unaligned_u64(out + L.pb[li] + i, v);
        }
    }
    CDBG_WPH(6);
    if (np) {
        // last base of every home k-mer (one lane per k-mer; no reverse complement needed:
        // the last base of rc(x) is the complement of the first base of x)
        for (uint32_t e = lane; e < E; e += 64) {
            if (L.vis[e] & 2u) continue;                 // traveller copy: belongs to another bucket's piece
            const uint32_t v = L.cnt[e];
            const Kmer<W> x = cw_key<W, TSW>(L, e);
            const uint32_t b = (v & 1u) ? 3u - x.base(k, 0) : x.base(k, k - 1);
            out[v >> 1] = (uint8_t)("ACGT"[b]);
        }
        // ---- glue log, part 2: one record per open piece end ----
        for (uint32_t it = lane; it < 2 * E; it += 64) {
            const uint32_t l = L.lnk[it];
            if (!(l & CWL_POSTED)) continue;
            const uint64_t o = lbase + atomic_add_u32(&L.lw, 1u);
            const Kmer<W> jc = canon_junction_at<W>(cw_key<W, TSW>(L, it >> 1), it & 1u, k);
            glue_record_put<W>(P, o, jc, (uint32_t)(pbase * 2 + (l & 0x3FFu)) | ((l & CWL_CONF) ? GTAG_CONFBIT : 0u));
        }
    }
    CDBG_WAVE_SYNC();
#ifdef CDBG_WAVE_DEBUG
    if (lane == 0) { for (uint32_t it = 0; it < 2 * E; ++it) fprintf(stderr, " [%u cnt %x vis %x lnk %04x]", it, L.cnt[it >> 1], L.vis[it >> 1], L.lnk[it]); fprintf(stderr, "\n"); }
    if (lane == 0) fprintf(stderr, "bucket %u E %u np %u nconf %u nopen %u lw %u pbase %llu lbase %llu nterm %u ncov %u nhome %u\n", p, E, np, nconf, L.nopen, L.lw, (unsigned long long)pbase, (unsigned long long)lbase, nterm, L.ncov, n_home);
#endif
    const uint32_t nopen_posted = np ? uni_u32(L.lw) - (log_ok ? nconf : 0u) : 0u;
    acc[0] += nopen_posted; acc[1] += log_ok ? nconf : 0u; acc[2] += uni_u32(L.ncyc); acc[3] += np;
    CDBG_WPH(7);
    CDBG_WAVE_SYNC();
    CDBG_WPH(8);
}

// Persistent waves pulling batches of buckets from ONE device-wide queue (a returning atomic per batch, requested one
// batch ahead): the launch stays balanced whatever number of workgroups the hardware really admits per CU (the occupancy
// query can be one high, MI355X_MICROARCH.md "Residency": a partial second generation would run alone).  Pipeline per
// wave: queue ticket -> segment descriptor -> entries -> compaction, one bucket apart each; the entries live in two
// register sets (ping-pong, as in k_count_fast).
template <int W, int TSW>
// waves per SIMD promised to the register allocator.  With the junction table (4-byte slots, no k-mer hash table; the walk-1
// work list in its memory once the links are known) a workgroup of two waves needs 16 / 10 / 14 KB of LDS (W = 1 / 2 / 4), i.e. the
// LDS has room for 20 / 32 / 22 waves per CU, and the kernel waits on dependent LDS round trips: occupancy is what it is short of.
// Left alone the compiler takes 132 / 83 / 110 VGPRs = 12 / 20 / 16 waves per CU.  Measured: W = 1 promised 4 (16 waves per CU)
// 32.2 -> 26.7 ms at config 3, promised 5 with the smaller LDS (92 VGPRs, 20 waves per CU) 25.8 -> 24.4 ms; W = 2 promised 6 (80
// VGPRs, 24 waves per CU) 49.3 -> 47.7 ms at the config-4 share; W = 4 promised 4 or 5: no change (36 ms).
#ifndef CDBG_CW_WAVES1
#define CDBG_CW_WAVES1 5
#endif
#ifndef CDBG_CW_WAVES2
#define CDBG_CW_WAVES2 6
#endif
#ifndef CDBG_CW_WAVES4
#define CDBG_CW_WAVES4 3
#endif
__global__ void __launch_bounds__(CW_THREADS, W == 1 ? (TSW > 512 ? 3 : CDBG_CW_WAVES1) : W == 2 ? (TSW > 256 ? 2 : CDBG_CW_WAVES2) : (TSW > 256 ? 2 : CDBG_CW_WAVES4)) k_compact_wave(CompactWaveParams WP) {
    CDBG_SHARED CompactWaveLds<W, TSW> Ls[CW_THREADS / 64];
    const CompactParams& P = WP.c;
    const int tid = threadIdx.x, lane = tid & 63, wave = (int)uni_u32((uint32_t)tid >> 6);
    CompactWaveLds<W, TSW>& L = Ls[wave];
    for (uint32_t i = lane; i < (uint32_t)TSW; i += 64) L.jt[i] = 0u;
    CDBG_WAVE_SYNC();
    uint64_t acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#if defined(CDBG_PROFILE_PHASES) && !defined(CDBG_HOSTSIM)
    acc[15] = clock64();
#endif
    CwChunk pc{0, 0}, bc{0, 0}, lc{0, 0};
    const uint32_t nb = WP.n_buckets;
    // (no lambdas that capture the parameter block by reference: taking its address sends the pointers it holds through
    //  memory, the compiler then loses their address space and every global access becomes a FLAT one -- which counts
    //  against lgkmcnt as well, so each LDS wait would also wait for the global loads and stores in flight)
    uint32_t* const queue = WP.queue;
    const uint32_t* const seg_n = P.seg_n; const uint64_t* const seg_off = P.seg_off;
    // One word saturates at ~88 M returning atomics per second (MI355X_MICROARCH.md, "dequeue"): a ticket per bucket
    // would cap the stage at 4.2 M buckets / 48 ms.  A ticket therefore hands out CW_BATCH consecutive buckets, and the
    // next ticket is requested when a batch starts.
    uint32_t q_batch, q_j = 0, q_next_raw;              // id stream: current batch, position in it, next batch (raw: lane 0 only)
#define CW_TICKET(dst) do { uint32_t t_ = 0; if (lane == 0) t_ = atomic_add_u32(queue, 1u); (dst) = t_; } while (0)
#define CW_NEXT_ID(dst) do { if (q_j == CW_BATCH) { q_batch = uni_u32(q_next_raw); CW_TICKET(q_next_raw); q_j = 0; } (dst) = q_batch * CW_BATCH + q_j; ++q_j; } while (0)
    // (second wave tier: the items are entries of part_list -- the buckets the first tier deferred; one more dependent load)
    const uint32_t* const plist = P.part_list;
#define CW_SEG_LOAD(p_, n_, off_, id_) do { uint32_t q_ = (p_) < nb ? (p_) : nb - 1u; if (plist) q_ = plist[q_]; (id_) = q_; (n_) = seg_n[q_]; (off_) = seg_off[q_]; } while (0)
    CwEntries<W, TSW> X0, X1;
    uint32_t p_cur, p_nxt, p_nn;                        // buckets: being compacted, entries requested, descriptor requested
    { uint32_t t; CW_TICKET(t); q_batch = uni_u32(t); CW_TICKET(q_next_raw); }
    CW_NEXT_ID(p_cur); CW_NEXT_ID(p_nxt); CW_NEXT_ID(p_nn);
    uint32_t E_cur, segn_nxt, id_cur, id_nxt; uint64_t sego_nxt;
    {
        uint32_t n0; uint64_t o0; CW_SEG_LOAD(p_cur, n0, o0, id_cur);
        E_cur = p_cur < nb ? uni_u32(n0) : 0u; id_cur = uni_u32(id_cur);
        cw_load_entries<W, TSW>(P, uni_u64(o0), E_cur, lane, X0);
    }
    CW_SEG_LOAD(p_nxt, segn_nxt, sego_nxt, id_nxt);
#define CW_ONE_BUCKET(Xc, Xn) do {                                                                                         \
        uint32_t segn_nn, id_nn; uint64_t sego_nn; CW_SEG_LOAD(p_nn, segn_nn, sego_nn, id_nn);  /* descriptor two buckets ahead */ \
        const uint32_t E_nxt = p_nxt < nb ? uni_u32(segn_nxt) : 0u;                                                         \
        cw_load_entries<W, TSW>(P, uni_u64(sego_nxt), E_nxt, lane, Xn);                     /* entries one bucket ahead */     \
        if (E_cur > (uint32_t)(TSW / 2)) {                  /* more entries than this wave tier holds: the next tier */         \
            if (lane == 0) { const uint32_t i_ = atomic_add_u32(P.big_count, 1u); P.big_list[i_] = id_cur; }                \
        } else if (E_cur) compact_bucket_wave<W, TSW>(P, L, id_cur, E_cur, lane, Xc, pc, bc, lc, acc);                      \
        p_cur = p_nxt; E_cur = E_nxt; id_cur = uni_u32(id_nxt); p_nxt = p_nn; segn_nxt = segn_nn; sego_nxt = sego_nn; id_nxt = id_nn; \
        CW_NEXT_ID(p_nn);                                                                                                   \
    } while (0)
    while (p_cur < nb) {
        CW_ONE_BUCKET(X0, X1);
        if (p_cur >= nb) break;
        CW_ONE_BUCKET(X1, X0);
    }
#undef CW_ONE_BUCKET
#undef CW_SEG_LOAD
#undef CW_TICKET
#undef CW_NEXT_ID
    if (lane == 0) for (int i = 0; i < 4; ++i) if (acc[i]) atomic_add_u64(&P.stats[i], acc[i]);
#if defined(CDBG_PROFILE_PHASES) && !defined(CDBG_HOSTSIM)
    if (lane == 0) for (int i = 4; i < 13; ++i) atomic_add_u64(&P.stats[4 + i], acc[i]);
#endif
}

}  // namespace cdbg
