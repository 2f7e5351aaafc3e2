// k_compact.h -- stage 2: per-minimizer-bucket unitig compaction in LDS.
//
// New MI355X design for the job of bcalm2<span>() / graph3 in gatb-core's
// bcalm_algo (entered via
// /root/reference/src/bcalm_1.cpp:57).  The correctness contract is the unitig
// definition of /root/reference/bidirected-graphs-in-bcalm2/
// bidirected-graphs-in-bcalm2.md:83-88.
//
// A bucket = one minimizer partition = the solid k-mers counted there: HOME k-mers
// plus TRAVELLER copies (k-mers homed elsewhere that touch one of this bucket's
// junctions).  By construction the bucket that owns junction J (the (k-1)-mer whose
// minimizer hashes here) holds EVERY solid k-mer adjacent to J, so whether J is a
// 1-in/1-out junction is decided locally by probing the LDS table (4 successors +
// 4 back-probes), with the spec's exclusions (self-loop, hairpin, palindromic
// junction => x == y => never merged).
//
//   end status   DEAD      junction not 1-1 (or excluded)           -> unitig ends here
//                INTERNAL  1-1 and the neighbour is HOME here       -> merged in LDS
//                OPEN      junction owned by another bucket, or the 1-1 neighbour is a
//                          traveller                                  -> resolved by glue
// Each piece (maximal chain of INTERNAL links) is written once, with its k-mer count,
// summed abundance and, for OPEN ends, a glue-table insert keyed by the canonical
// junction (k-1)-mer; the bucket that owns a 1-1 junction with a traveller on either
// side also posts CONFIRM(J).  Glue joins the two ends of J iff J is confirmed.
#pragma once
#include "k_count.h"

namespace cdbg {

constexpr int COMPACT_THREADS = 256;
constexpr uint32_t PIECE_CHUNK = 1024, BASES_CHUNK = 1u << 16;   // per-workgroup reservations (one device atomic each)
constexpr uint32_t LNK_DEAD = 0, LNK_INTERNAL = 1, LNK_OPEN = 2;
constexpr uint32_t LNK_CONF = 1u << 30;                 // this end's junction is 1-1 with a traveller: post CONFIRM
constexpr uint32_t LNK_POSTED = 1u << 31;               // open piece end; low bits = 2 * (piece index inside the bucket) + side, LNK_CONF kept
constexpr uint32_t GLOG_CHUNK = 2048;                   // glue-log records a workgroup reserves per device atomic
constexpr uint32_t GTAG_EMPTY = 0xFFFFFFFFu, GTAG_CONFIRM = 0xFFFFFFFEu;
constexpr uint32_t GTAG_CONFBIT = 0x80000000u;          // end record that also confirms its junction (end ids stay below 0x7FFFFFF0)
constexpr uint32_t NONE32 = 0xFFFFFFFFu;
constexpr uint32_t END_LEFT = 0, END_RIGHT = 1;

// ---- glue table: canonical junction (k-1)-mer -> (end a, end b, confirmed) in HBM ----
template <int W>
struct GlueTable {
    KTable<W> t;
    uint32_t* a; uint32_t* b;      // piece-end id + 1 (0 = empty)
    uint32_t* conf;                // 1 = junction confirmed 1-1 by its owning bucket
};
// conf_bit = GTAG_CONFBIT when the end's own bucket confirmed the junction 1-1: the flag rides in the top bit of
// the a/b word, so an end + confirmation costs the same two device atomics as a plain end
template <int W>
CDBG_DEV void glue_post_end(const GlueTable<W>& G, const Kmer<W>& jc, uint32_t end_id, uint32_t conf_bit) {
    bool nw; const uint32_t s = ktable_insert<W, true>(G.t, jc, nw);
    const uint32_t v = (end_id + 1u) | conf_bit;
    if (atomic_cas_u32(&G.a[s], 0u, v) != 0u) atomic_cas_u32(&G.b[s], 0u, v);
}
template <int W>
CDBG_DEV void glue_post_confirm(const GlueTable<W>& G, const Kmer<W>& jc) {
    bool nw; const uint32_t s = ktable_insert<W, true>(G.t, jc, nw);
    atomic_or_u32(&G.conf[s], 1u);
}

struct CompactParams {
    const uint64_t* solid_keys; const uint32_t* solid_cnt;
    const uint64_t* seg_off; const uint32_t* seg_n;
    const uint32_t* part_list;     // optional (big pass)
    int k, m, log_np, rank_bits, rank;
    // pieces
    uint32_t* piece_n; uint64_t* piece_kc; uint64_t* piece_boff; uint8_t* piece_bases;
    uint32_t* piece_ab;            // optional (-all-abundance-counts): abundance of the k-mer ending at each base offset
    uint64_t piece_cap, bases_cap;
    uint64_t* piece_cursor; uint64_t* bases_cursor;
    // glue table (HBM)
    uint64_t* glue_keys; uint32_t* glue_a; uint32_t* glue_b; uint32_t* glue_conf; uint32_t glue_mask;
    // glue log: (junction key, tag) records; tag = piece-end id, GTAG_CONFIRM, or GTAG_EMPTY (pre-filled)
    uint64_t* glog_keys; uint32_t* glog_tag; uint64_t glog_cap; uint64_t* glog_cursor;
    // ... or, single-rank contexts, straight into the join buckets of the glue stage (k_glue.h): jrecs != nullptr.
    // The compaction kernels wait on LDS most of the time; their memory pipes take the one counter atomic + one
    // 16-byte store per record that a separate scatter pass over the log paid 11 ms for (config 3).
    uint32_t* jfill; uint64_t* jrecs; int log_jb;
    uint32_t* big_list; uint32_t* big_count; uint32_t* error;
    uint64_t* stats;               // [0] open ends posted [1] confirms posted [2] in-bucket cycles [3] pieces written
    // HBM scratch (GLOBAL variant)
    uint64_t* g_keys; uint32_t* g_cnt; uint32_t* g_lnk; uint32_t* g_aux; const uint64_t* big_off;
    uint32_t n_items;              // buckets (or part_list entries) to process
    uint32_t split;                // the buckets are sub-buckets of the second-level split (k_split.h): ownership is NOT by minimizer partition
};

// join buckets of the glue stage: <= JB_CAP records each, chosen by a hash of the junction key; a record = W key words
// + the tag word, ONE scattered store
constexpr uint32_t JB_CAP = 256;
template <int W>
CDBG_DEV void join_bucket_put(uint32_t* jfill, uint64_t* jrecs, int log_jb, uint32_t* error, const Kmer<W>& jc, uint32_t tag) {
    const uint32_t b = log_jb ? jc.hash_lds() >> (32 - log_jb) : 0u;
    const uint32_t pos = atomic_add_u32(&jfill[b], 1u);
    if (pos >= JB_CAP) { *error = 8; return; }           // (the host falls back: log + global table)
    const uint64_t o = (uint64_t)b * JB_CAP + pos;
    if (W == 1) { uint4 r; r.x = (uint32_t)jc.w[0]; r.y = (uint32_t)(jc.w[0] >> 32); r.z = tag; r.w = 0; reinterpret_cast<uint4*>(jrecs)[o] = r; }
    else { for (int j = 0; j < W; ++j) jrecs[o * (W + 1) + j] = jc.w[j]; jrecs[o * (W + 1) + W] = tag; }
}
// one glue record: into its join bucket, or at position o of the sequential log
template <int W>
CDBG_DEV void glue_record_put(const CompactParams& P, uint64_t o, const Kmer<W>& jc, uint32_t tag) {
    if (P.jrecs) { join_bucket_put<W>(P.jfill, P.jrecs, P.log_jb, P.error, jc, tag); return; }
    for (int i = 0; i < W; ++i) P.glog_keys[o * W + i] = jc.w[i];
    P.glog_tag[o] = tag;
}

// bases [i, i + 8) of a k-mer as 8 ASCII letters in memory order: one 16-bit field (the word index is a select chain once
// per 8 bases, not per base), 2-bit codes spread to bytes, letters by arithmetic (0 A 0x41, 1 C 0x43, 2 G 0x47, 3 T 0x54)
template <int W>
CDBG_DEV uint32_t kmer_bases8(const Kmer<W>& x, int k, int i) {   // bases i .. i + 7 as 16 bits: base i in bits 15:14 ... base i + 7 in bits 1:0
    const int pos = 2 * (k - i - 8);                       // bit of base i + 7; requires 0 <= i, i + 8 <= k
    uint64_t v = x.word_z(pos >> 6) >> (pos & 63);
    if (W > 1 && (pos & 63) > 48) v |= x.word_z((pos >> 6) + 1) << (64 - (pos & 63));
    return (uint32_t)v & 0xFFFFu;
}
CDBG_DEV uint64_t ascii8_of_bases(uint32_t f) {
    const uint32_t a = f >> 8, b = f & 0xFFu;
    const uint32_t ta = ((a >> 6) & 3u) | (((a >> 4) & 3u) << 8) | (((a >> 2) & 3u) << 16) | ((a & 3u) << 24);
    const uint32_t tb = ((b >> 6) & 3u) | (((b >> 4) & 3u) << 8) | (((b >> 2) & 3u) << 16) | ((b & 3u) << 24);
    const uint32_t ea = ((ta & 0x01010101u) * 2u) + ((ta & 0x02020202u) * 3u) + 0x41414141u + (((ta >> 1) & ta & 0x01010101u) * 11u);
    const uint32_t eb = ((tb & 0x01010101u) * 2u) + ((tb & 0x02020202u) * 3u) + 0x41414141u + (((tb >> 1) & tb & 0x01010101u) * 11u);
    return (uint64_t)ea | ((uint64_t)eb << 32);
}
template <int W>
CDBG_DEV uint64_t kmer_ascii8(const Kmer<W>& x, int k, int i) { return ascii8_of_bases(kmer_bases8<W>(x, k, i)); }
// bases i .. i + 7 of rc(x) without forming rc(x): the complement of bases k - 8 - i .. k - 1 - i of x, in reverse order
template <int W>
CDBG_DEV uint64_t kmer_rc_ascii8(const Kmer<W>& x, int k, int i) {
    uint32_t f = kmer_bases8<W>(x, k, k - 8 - i);
    f = ((f >> 8) | (f << 8)) & 0xFFFFu;                   // reverse the eight 2-bit groups: bytes, nibbles, pairs
    f = ((f >> 4) & 0x0F0Fu) | ((f & 0x0F0Fu) << 4);
    f = ((f >> 2) & 0x3333u) | ((f & 0x3333u) << 2);
    return ascii8_of_bases(f ^ 0xFFFFu);
}
// the first n bases of x to dst (any alignment): 8 per store, the last 8 overlapping; bytes when n < 8
template <int W>
CDBG_DEV void kmer_prefix_ascii(uint8_t* dst, const Kmer<W>& x, int k, int n) {
    if (n >= 8) {
        for (int i = 0; ; i += 8) {
            if (i + 8 > n) { if (i == n) break; i = n - 8; }
            st_unaligned_u64(dst + i, kmer_ascii8<W>(x, k, i));
            if (i + 8 == n) break;
        }
    } else {
        for (int i = 0; i < n; ++i) dst[i] = (uint8_t)("ACGT"[x.base(k, i)]);
    }
}

// orient stored label x so that `end` is on the right (we leave x through `end`)
template <int W>
CDBG_DEV Kmer<W> orient_out(const Kmer<W>& x, uint32_t end, int k) { return end == END_RIGHT ? x : x.rc(k); }

// A stored (canonical) k-mer x seen from one of its ends: u reads out of that end, ur = rc(u).  One reverse complement
// serves both (x.rc for the left end IS u, for the right end it is rc(u)); orient_out followed by u.rc(k) took two.
template <int W>
CDBG_DEV void orient_pair(const Kmer<W>& x, uint32_t end, int k, Kmer<W>& u, Kmer<W>& ur) {
    const Kmer<W> xr = x.rc(k);
    const bool right = end == END_RIGHT;
#pragma unroll
    for (int i = 0; i < W; ++i) { u.w[i] = right ? x.w[i] : xr.w[i]; ur.w[i] = right ? xr.w[i] : x.w[i]; }
}
// canonical junction (k-1)-mer at that end: the suffix of u, or its reverse complement = rc(u) without its last base
template <int W>
CDBG_DEV Kmer<W> canon_junction_at(const Kmer<W>& x, uint32_t end, int k) {
    Kmer<W> u, ur; orient_pair<W>(x, end, k, u, ur);
    const Kmer<W> j = suffix_km1<W>(u, k), r = ur.shr(2);
    return (r < j) ? r : j;
}

// successors of oriented k-mer u present in the table; returns count, last hit in (slot, enter_end)
template <int W>
CDBG_DEV int probe_succ(const KTable<W>& T, const Kmer<W>& u, int k, uint32_t& slot, uint32_t& enter_end) {
    int n = 0;
    // the four successors u[1:]+c share everything but one base: v = (u << 2) | c, and its reverse complement is
    // comp(c) in front of rc(u) without its last base -- one rc() for the four probes
    Kmer<W> vb = u; vb.push_right(k, 0);
    const Kmer<W> rb = u.rc(k).shr(2);
    const int pos = 2 * (k - 1);
    for (uint32_t c = 0; c < 4; ++c) {
        Kmer<W> v = vb; v.w[0] |= (uint64_t)c;
        Kmer<W> r = rb; r.or_word(pos >> 6, (uint64_t)(3u - c) << (pos & 63));
        const bool fwd = !(r < v);                       // v is the canonical label
        const uint32_t f = ktable_find<W>(T, fwd ? v : r);
        // (even k: a successor that is its own reverse complement is reached by two edges, .md:41-46: never a unique successor)
        if (f != NONE32) { n += (r == v) ? 2 : 1; slot = f; enter_end = fwd ? END_LEFT : END_RIGHT; }
    }
    return n;
}

template <int W, int TS, bool GLOBAL>
CDBG_DEV void compact_bucket(const CompactParams& P, const uint32_t item, uint64_t (&acc)[4], bool& clean,
                             uint64_t& pc_base, uint32_t& pc_left, uint64_t& bc_base, uint32_t& bc_left,
                             uint64_t& lc_base, uint32_t& lc_left, uint64_t (&ph)[8], uint64_t& t_prev) {
    CDBG_SHARED uint64_t l_keys[GLOBAL ? 1 : TS * W];
    CDBG_SHARED uint32_t l_cnt[GLOBAL ? 1 : TS];
    CDBG_SHARED uint32_t l_lnk[GLOBAL ? 1 : 2 * TS];
    // aux words: [0, cap/4) visited bytes | piece starts | entry slots | piece lengths | piece base offsets (cap/2 each)
    CDBG_SHARED uint32_t l_aux[GLOBAL ? 1 : 2 * TS + TS / 4];
    CDBG_SHARED uint32_t s_np, s_nb, s_stat[4];
    CDBG_SHARED uint64_t s_pbase, s_bbase, s_lbase;
    CDBG_SHARED uint32_t s_nopen, s_lw;

    const int tid = threadIdx.x;
    const int k = P.k;
    const uint32_t p = P.part_list ? P.part_list[item] : item;
    const uint32_t E = P.seg_n[p];
    CDBG_PH(0);
    if (E == 0) return;

    KTable<W> T; uint32_t *cnt, *lnk, *pdesc, *slots, *pn, *pb; uint8_t* vis; uint32_t cap;
    if (GLOBAL) {
        const uint64_t o0 = P.big_off[item]; cap = (uint32_t)(P.big_off[item + 1] - o0);
        T.keys = P.g_keys + o0 * W; cnt = P.g_cnt + o0;
        lnk = P.g_lnk + 2 * o0; vis = reinterpret_cast<uint8_t*>(P.g_aux + 3 * o0); pdesc = P.g_aux + 3 * o0 + cap / 4;
    } else {
        if (E > (uint32_t)TS / 2) {                      // does not fit LDS: defer to the big pass
            if (tid == 0) { const uint32_t i = atomic_add_u32(P.big_count, 1u); P.big_list[i] = p; }
            return;
        }
        cap = TS; T.keys = l_keys; cnt = l_cnt; lnk = l_lnk; vis = reinterpret_cast<uint8_t*>(l_aux); pdesc = l_aux + TS / 4;
    }
    T.mask = cap - 1;
    slots = pdesc + cap / 2; pn = slots + cap / 2; pb = pn + cap / 2;   // E <= cap/2 entries, <= cap/2 pieces
    const uint32_t pg = (p << P.rank_bits) | (uint32_t)P.rank;     // global partition id of this bucket

    if (tid == 0) { s_np = 0; s_nb = 0; s_nopen = 0; s_lw = 0; s_stat[0] = s_stat[1] = s_stat[2] = s_stat[3] = 0; }
    if (GLOBAL || !clean) {                                  // (the previous bucket emptied the slots it had used)
        ktable_clear<W>(T, tid, COMPACT_THREADS);
        for (uint32_t i = tid; i < cap; i += COMPACT_THREADS) vis[i] = 0;
    }
    block_sync<GLOBAL>();

    // ---- load the bucket: home + traveller solid k-mers ----
    const uint64_t so = P.seg_off[p];
    for (uint32_t e = tid; e < E; e += COMPACT_THREADS) {
        Kmer<W> x;
        for (int i = 0; i < W; ++i) x.w[i] = P.solid_keys[(so + e) * W + i];
        const uint64_t fl = x.w[W - 1] & KEY_FLAGS;           // (KEY_FOREIGN_*, k_count.h: the junctions this bucket does NOT own)
        x.w[W - 1] &= ~KEY_FLAGS;
        bool nw; const uint32_t s = ktable_insert<W, GLOBAL>(T, x, nw);
        const uint32_t cv = P.solid_cnt[so + e];
        cnt[s] = cv;
        // vis byte: bit 0 visited (walk 1), bit 1 traveller (cnt[] is recycled for byte offsets later),
        // bit 2 / 3: the junction at the LEFT / RIGHT end of the label is owned by this bucket
        vis[s] = (uint8_t)(((cv & TRAV_FLAG) ? 2u : 0u) | ((fl & KEY_FOREIGN_L) ? 0u : 4u) | ((fl & KEY_FOREIGN_R) ? 0u : 8u));
#ifdef CDBG_HOSTSIM
        {   // (the simulator build checks every flag against the definition)
            uint32_t gl, gr; kmer_junction_mins<W>(x, k, P.m, gl, gr);
            if (!P.split && ((part_of(gl, P.log_np) == pg) != !(fl & KEY_FOREIGN_L) || (part_of(gr, P.log_np) == pg) != !(fl & KEY_FOREIGN_R))) *P.error = 9;   // (sub-buckets of k_split.h own junctions by sub-minimizer)
        }
#endif
        slots[e] = s;
    }
    block_sync<GLOBAL>();
    CDBG_PH(1);

    // ---- classify both ends of every entry ----
    // Step 1: every end whose junction this bucket owns probes its successors ONCE and notes its unique partner
    // end (or none).  Step 2: a junction is 1-in/1-out exactly when two ends name each other, so the second
    // round of probes (from the partner back) is replaced by one LDS read.  Final link words are staged in
    // the piece-length/offset arrays (unused until walk 1) because step 2 still reads the notes in lnk[].
    constexpr uint32_t NOTE_NONE = 0xFFFFFFFFu, NOTE_FOREIGN = 0xFFFFFFFEu;
    uint32_t* const fin = pn;                                // 2E <= cap words (pn and pb are contiguous)
    for (uint32_t it = tid; it < 2 * E; it += COMPACT_THREADS) {
        const uint32_t s = slots[it >> 1], end = it & 1u, idx = s * 2 + end;
        uint32_t note = NOTE_FOREIGN;                        // junction owned elsewhere: glue decides
        if ((vis[s] >> (2 + end)) & 1u) {
            const Kmer<W> u = orient_out<W>(ktable_key<W>(T, s), end, k);
            uint32_t y = 0, ye = 0;
            note = (probe_succ<W>(T, u, k, y, ye) == 1 && y != s) ? (y * 2 + ye) : NOTE_NONE;
        }
        lnk[idx] = note;
    }
    block_sync<GLOBAL>();
    for (uint32_t it = tid; it < 2 * E; it += COMPACT_THREADS) {
        const uint32_t s = slots[it >> 1], end = it & 1u, idx = s * 2 + end;
        const bool home = !(cnt[s] & TRAV_FLAG);
        const uint32_t note = lnk[idx];
        uint32_t link = LNK_DEAD; bool conf = false;
        if (note == NOTE_FOREIGN) link = LNK_OPEN;
        else if (note != NOTE_NONE && lnk[note] == idx) {    // the partner end has exactly one successor too: this one
            const uint32_t y = note >> 1, ye = note & 1u;
            const bool yhome = !(cnt[y] & TRAV_FLAG);
            if (home && yhome) link = LNK_INTERNAL | (ye << 2) | (y << 3);
            else {
                // 1-1 junction with a traveller on at least one side: confirm it for glue (once)
                // (a home end is open and posted anyway: its confirmation rides on that record)
                if (home || (!yhome && s < y)) { conf = true; if (!home) atomic_add_u32(&s_stat[1], 1u); }
                if (home) link = LNK_OPEN;
            }
        }
        fin[it] = (home ? link : LNK_DEAD) | (conf ? LNK_CONF : 0u);
    }
    block_sync<GLOBAL>();
    for (uint32_t it = tid; it < 2 * E; it += COMPACT_THREADS) lnk[slots[it >> 1] * 2 + (it & 1u)] = fin[it];
    block_sync<GLOBAL>();
    CDBG_PH(2);

    // ---- walk 1: every terminal end measures its piece; the smaller terminal id registers it ----
    for (uint32_t it = tid; it < 2 * E; it += COMPACT_THREADS) {
        const uint32_t s = slots[it >> 1], end = it & 1u, idx = s * 2 + end;
        if (cnt[s] & TRAV_FLAG) continue;
        if ((lnk[idx] & 3u) == LNK_INTERNAL) continue;   // not a terminal
        uint32_t cur = s, ex = end ^ 1u, n = 1;
        vis[cur] = vis[cur] | 1u;
        for (;;) {
            const uint32_t l = lnk[cur * 2 + ex];
            if ((l & 3u) != LNK_INTERNAL) break;
            cur = l >> 3; ex = ((l >> 2) & 1u) ^ 1u; ++n;
            vis[cur] = vis[cur] | 1u;
        }
        const uint32_t other = cur * 2 + ex;
        if (idx <= other) {
            const uint32_t li = atomic_add_u32(&s_np, 1u);
            pdesc[li] = idx;                             // start terminal (bit 31 clear: linear piece)
            pn[li] = n;
            atomic_add_u32(&s_nb, n + (uint32_t)k - 1u);
            const uint32_t no = ((lnk[idx] & 3u) == LNK_OPEN ? 1u : 0u) + ((lnk[other] & 3u) == LNK_OPEN ? 1u : 0u);
            if (no) atomic_add_u32(&s_nopen, no);
        }
    }
    block_sync<GLOBAL>();
    CDBG_PH(3);
    // ---- closed chains entirely inside the bucket (isolated cycles): cut at the smallest slot ----
    for (uint32_t it = tid; it < E; it += COMPACT_THREADS) {
        const uint32_t s = slots[it];
        if (vis[s] & 3u) continue;                       // traveller, or already part of a piece
        uint32_t cur = s, ex = END_RIGHT, n = 0; bool is_min = true;
        do {
            const uint32_t l = lnk[cur * 2 + ex];
            cur = l >> 3; ex = ((l >> 2) & 1u) ^ 1u; ++n;
            if (cur < s) is_min = false;
        } while (cur != s);
        if (is_min) {
            const uint32_t li = atomic_add_u32(&s_np, 1u);
            pdesc[li] = (s * 2 + END_LEFT) | 0x80000000u;   // cyclic piece starting at s, walking right
            pn[li] = n;
            atomic_add_u32(&s_nb, n + (uint32_t)k - 1u);
            atomic_add_u32(&s_stat[2], 1u);
        }
    }
    block_sync<GLOBAL>();
    if (tid == 0) {
        s_stat[3] = s_np;                                // pieces really written (ids also cover reservation gaps)
        // sub-allocate piece ids and base bytes from this workgroup's chunks
        uint64_t pb, bb;
        if (s_np > PIECE_CHUNK) pb = atomic_add_u64(P.piece_cursor, (uint64_t)s_np);
        else { if (s_np > pc_left) { pc_base = atomic_add_u64(P.piece_cursor, (uint64_t)PIECE_CHUNK); pc_left = PIECE_CHUNK; }
               pb = pc_base; pc_base += s_np; pc_left -= s_np; }
        if (s_nb > BASES_CHUNK) bb = atomic_add_u64(P.bases_cursor, (uint64_t)s_nb);
        else { if (s_nb > bc_left) { bc_base = atomic_add_u64(P.bases_cursor, (uint64_t)BASES_CHUNK); bc_left = BASES_CHUNK; }
               bb = bc_base; bc_base += s_nb; bc_left -= s_nb; }
        const uint32_t nlog = s_stat[1] + s_nopen;       // confirms + open piece ends
        uint64_t lb;
        if (nlog > GLOG_CHUNK) lb = atomic_add_u64(P.glog_cursor, (uint64_t)nlog);
        else { if (nlog > lc_left) { lc_base = atomic_add_u64(P.glog_cursor, (uint64_t)GLOG_CHUNK); lc_left = GLOG_CHUNK; }
               lb = lc_base; lc_base += nlog; lc_left -= nlog; }
        if (pb + s_np > P.piece_cap || bb + s_nb > P.bases_cap) { *P.error = 3; s_np = 0; }
        if (lb + nlog > P.glog_cap) { *P.error = 5; s_np = 0; lb = 0; s_stat[1] = 0; }   // never write past the log
        s_pbase = pb; s_bbase = bb; s_lbase = lb; s_nb = 0;
    }
    block_sync<GLOBAL>();

    CDBG_PH(4);
    // ---- glue log, part 1: CONFIRM records of junctions whose two k-mers are both travellers here ----
    const uint32_t np = s_np;
    if (np || s_stat[1]) {
        for (uint32_t it = tid; it < 2 * E; it += COMPACT_THREADS) {
            const uint32_t s = slots[it >> 1], end = it & 1u;
            const uint32_t l = lnk[s * 2 + end];
            if (!(l & LNK_CONF) || (l & 3u) == LNK_OPEN) continue;   // open home ends carry their confirmation themselves
            const uint64_t o = s_lbase + atomic_add_u32(&s_lw, 1u);
            const Kmer<W> jc = canon_junction_at<W>(ktable_key<W>(T, s), end, k);
            glue_record_put<W>(P, o, jc, GTAG_CONFIRM);
        }
    }
    block_sync<GLOBAL>();
    // ---- walk 2: one lane per piece walks its chain ONCE more, only to hand every k-mer its byte
    // offset + strand (stored over the k-mer's count, which is summed here) and to mark open ends;
    // the bases themselves are written by the dense phases below, one lane per k-mer ----
    for (uint32_t li = tid; li < np; li += COMPACT_THREADS) {
        const uint32_t d = pdesc[li];
        const bool cyclic = d & 0x80000000u;
        const uint32_t start = d & 0x7FFFFFFFu;
        const uint32_t s0 = start >> 1, e0 = start & 1u, n = pn[li];
        const uint32_t rel = atomic_add_u32(&s_nb, n + (uint32_t)k - 1u);
        pb[li] = rel;
        uint64_t kc = 0;
        uint32_t cur = s0, ex = e0 ^ 1u;
        for (uint32_t t = 0; t < n; ++t) {
            const uint32_t ab = cnt[cur] & ~TRAV_FLAG;
            kc += (uint64_t)ab;
            if (P.piece_ab) P.piece_ab[s_bbase + rel + (uint32_t)k - 1u + t] = ab;
            cnt[cur] = ((rel + (uint32_t)k - 1u + t) << 1) | (ex == END_RIGHT ? 0u : 1u);
            if (t + 1 < n) { const uint32_t l = lnk[cur * 2 + ex]; cur = l >> 3; ex = ((l >> 2) & 1u) ^ 1u; }
        }
        const uint64_t pid = s_pbase + li;
        P.piece_n[pid] = n; P.piece_kc[pid] = kc; P.piece_boff[pid] = s_bbase + rel;
        if (!cyclic) {
            // left end of the piece = start terminal (s0, e0); right end = (cur, ex)
            const uint32_t il = s0 * 2 + e0, ir = cur * 2 + ex;
            const bool ol = (lnk[il] & 3u) == LNK_OPEN, orr = (lnk[ir] & 3u) == LNK_OPEN;
            if (ol) lnk[il] = LNK_POSTED | (lnk[il] & LNK_CONF) | (li * 2u + 0u);
            if (orr) lnk[ir] = LNK_POSTED | (lnk[ir] & LNK_CONF) | (li * 2u + 1u);
        }
    }
    block_sync<GLOBAL>();
    if (np) {
        uint8_t* const out = P.piece_bases + s_bbase;
        // last base of every home k-mer (one lane per k-mer; no reverse complement needed:
        // the last base of rc(x) is the complement of the first base of x)
        for (uint32_t it = tid; it < E; it += COMPACT_THREADS) {
            const uint32_t s = slots[it];
            if (vis[s] & 2u) continue;                           // traveller copy: belongs to another bucket's piece
            const uint32_t v = cnt[s];
            const Kmer<W> x = ktable_key<W>(T, s);
            const uint32_t b = (v & 1u) ? 3u - x.base(k, 0) : x.base(k, k - 1);
            out[v >> 1] = (uint8_t)("ACGT"[b]);
        }
        // the first k-1 bases of every piece (one lane per base)
        const uint32_t k1 = (uint32_t)k - 1u;
        for (uint32_t xx = tid; xx < np * k1; xx += COMPACT_THREADS) {
            const uint32_t li = xx / k1, i = xx - li * k1;
            const uint32_t start = pdesc[li] & 0x7FFFFFFFu;
            const Kmer<W> x = ktable_key<W>(T, start >> 1);
            const bool fwd = ((start & 1u) ^ 1u) == END_RIGHT;   // leaving through the right end: label as is
            const uint32_t b = fwd ? x.base(k, (int)i) : 3u - x.base(k, k - 1 - (int)i);
            out[pb[li] + i] = (uint8_t)("ACGT"[b]);
        }
    }
    // ---- glue log, part 2: one record per open piece end (dense over all ends) ----
    if (np) {
        uint32_t my_open = 0;
        for (uint32_t it = tid; it < 2 * E; it += COMPACT_THREADS) {
            const uint32_t s = slots[it >> 1], end = it & 1u;
            const uint32_t l = lnk[s * 2 + end];
            if (!(l & LNK_POSTED)) continue;
            const uint64_t o = s_lbase + atomic_add_u32(&s_lw, 1u);
            const Kmer<W> jc = canon_junction_at<W>(ktable_key<W>(T, s), end, k);
            glue_record_put<W>(P, o, jc, (uint32_t)(s_pbase * 2 + (l & 0x3FFFFFFFu)) | ((l & LNK_CONF) ? GTAG_CONFBIT : 0u));
            ++my_open;
        }
        if (my_open) atomic_add_u32(&s_stat[0], my_open);
    }
    block_sync<GLOBAL>();
    if (!GLOBAL) {                                           // hand the LDS table back empty: E slots instead of all TS
        for (uint32_t e = tid; e < E; e += COMPACT_THREADS) {
            const uint32_t s = slots[e];
            T.keys[(uint64_t)s * W + (W - 1)] = KEY_EMPTY; vis[s] = 0;
        }
        clean = true;                                        // (the kernel's per-bucket barrier orders this before the next load)
    }
    CDBG_PH(5);
    if (tid == 0) for (int i = 0; i < 4; ++i) acc[i] += (uint64_t)s_stat[i];
}

template <int W, int TS, bool GLOBAL>
__global__ void __launch_bounds__(COMPACT_THREADS, (W == 2 && TS <= 512 && !GLOBAL) ? 6 : 1) k_compact(CompactParams P) {   // two-word small tier: <= 80 VGPRs so that 6 workgroups per CU fit (the others are LDS-limited below that)
    uint64_t acc[4] = {0, 0, 0, 0};
    uint64_t pc_base = 0, bc_base = 0, lc_base = 0; uint32_t pc_left = 0, bc_left = 0, lc_left = 0;
    uint64_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t t_prev = 0;
#if defined(CDBG_PROFILE_PHASES) && !defined(CDBG_HOSTSIM)
    t_prev = wall_clock64();
#endif
    bool clean = false;
    for (uint32_t item = blockIdx.x; item < P.n_items; item += gridDim.x) {
        compact_bucket<W, TS, GLOBAL>(P, item, acc, clean, pc_base, pc_left, bc_base, bc_left, lc_base, lc_left, ph, t_prev);
        block_sync<GLOBAL>();                            // LDS is reused by the next bucket
    }
    if (threadIdx.x == 0) for (int i = 0; i < 4; ++i) if (acc[i]) atomic_add_u64(&P.stats[i], acc[i]);
#if defined(CDBG_PROFILE_PHASES) && !defined(CDBG_HOSTSIM)
    if (threadIdx.x == 0) for (int i = 0; i < 8; ++i) if (ph[i]) atomic_add_u64(&P.stats[8 + i], ph[i]);
#endif
}

// ---- glue table construction from the log: one lane per record, every lane busy, so the device
// atomics run at throughput instead of paying their latency inside the per-bucket kernel ----
struct GlueBuildParams {
    const uint64_t* glog_keys; const uint32_t* glog_tag; uint64_t n_records;
    uint64_t* glue_keys; uint32_t* glue_a; uint32_t* glue_b; uint32_t* glue_conf; uint32_t glue_mask;
    uint32_t shard_mask, shard_rank;   // multi-GPU sharded join: this rank takes the junctions with (mix32(hash) & mask) == rank; mask 0 = all
};
template <int W>
__global__ void k_glue_build(GlueBuildParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const GlueTable<W> G{ { P.glue_keys, P.glue_mask }, P.glue_a, P.glue_b, P.glue_conf };
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n_records; i += stride) {
        const uint32_t tag = P.glog_tag[i];
        if (tag == GTAG_EMPTY) continue;
        Kmer<W> jc;
        for (int j = 0; j < W; ++j) jc.w[j] = P.glog_keys[i * W + j];
        if (P.shard_mask && (mix32(jc.hash()) & P.shard_mask) != P.shard_rank) continue;
        if (tag == GTAG_CONFIRM) glue_post_confirm<W>(G, jc);
        else glue_post_end<W>(G, jc, tag & ~GTAG_CONFBIT, tag & GTAG_CONFBIT);
    }
}

}  // namespace cdbg
