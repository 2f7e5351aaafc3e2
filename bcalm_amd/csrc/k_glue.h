// k_glue.h -- stage 3: glue pieces into maximal unitigs.
//
// New MI355X design for the job of bglue<span>() + unionFind in gatb-core's
// bglue_algo (UF API hinted by
// /root/reference/example/uf/testUF.cpp:9-57) and of the unitig record writer
// (/root/reference/README.md:62-72: LN, KC, km).
//
// Every piece has at most one partner per end, so "gluing" is list ranking, not a
// general union-find: (1) hash-join of open piece ends on the canonical junction
// (k-1)-mer (the table was filled by k_compact; a junction is joined iff its owning
// bucket confirmed it 1-in/1-out), (2) pointer jumping over traversal states
// (state e = "enter piece e>>1 through end e&1"; succ(e) = link[e^1]) computing for
// every state its tail and the number of k-mers from it to the tail, (3) closed
// chains (isolated circular unitigs, /root/reference/example/circular_unitigs_unittests)
// are cut at their smallest piece, (4) one lane per piece copies its bases to its
// place inside the unitig, reverse-complemented when the chosen direction enters the
// piece from the right.  KC is the sum of the pieces' abundances.
#pragma once
#include "k_compact.h"

namespace cdbg {

constexpr int GLUE_THREADS = 256;

// ---- (1) sweep the glue table: confirmed junction with two ends -> mutual links ----
struct GlueResolveParams {
    const uint64_t* keys; const uint32_t* a; const uint32_t* b; const uint32_t* conf;
    uint32_t cap; int W;
    uint32_t* link;                // [2 * n_pieces], NONE32 = no partner
    uint64_t* stats;               // [0] junctions joined
};
// grid-stride, no LDS and no barrier: a workgroup that only lives for 256 table slots costs more to launch
// than to run (the table has 2^28 slots); joined junctions are counted per lane, then one atomic per wave
constexpr uint32_t GLUE_RESOLVE_GRID = 8192;
__global__ void k_glue_resolve(GlueResolveParams P) {
    uint32_t joined = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < P.cap; s += stride) {
        // confirmed 1-1 by the owning bucket: a CONFIRM record, or the flag riding on one of the two ends
        const uint32_t a = P.a[s];
        if (a == 0) continue;                                // no end posted
        const uint32_t b = P.b[s];
        if (b == 0) continue;
        if (!(((a | b) & 0x80000000u) || P.conf[s])) continue;
        const uint32_t ea = (a & 0x7FFFFFFFu) - 1u, eb = (b & 0x7FFFFFFFu) - 1u;
        P.link[ea] = eb;
        P.link[eb] = ea;
        ++joined;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) joined += __shfl_xor(joined, d);
    if ((threadIdx.x & 63) == 0 && joined) atomic_add_u64(&P.stats[0], (uint64_t)joined);
}

// ---- (1') bucketed junction join: the default path ----
// k_glue_build pays ~2.2 device atomics per log record into a table of 2^28 slots spread over 5 GB: every atomic is an
// HBM read-modify-write (25 GB of traffic for a 1.7 GB log; profiles/r02a_*).  Here the log is first scattered into
// JOIN BUCKETS of <= JB_CAP records by a hash of the junction key (one device atomic on a bucket counter -- an
// L2-resident array -- and one plain store per record), then ONE WAVE joins each bucket in an LDS table of 2 JB_CAP slots
// and writes the mutual links.  The global-table kernels above remain the fallback (a bucket that overflows).
constexpr int JB_THREADS = 256;                         // 4 independent waves per workgroup
constexpr uint32_t JB_PAIR_CHUNK = 2048;                // pairs a wave reserves per device atomic (sharded glue: pair output)
struct JoinScatterParams {
    const uint64_t* glog_keys; const uint32_t* glog_tag; uint64_t n_records; int log_jb;
    uint32_t* jfill; uint64_t* jrecs; uint32_t* error;   // a bucket record = W key words + the tag word: ONE scattered store
    uint32_t shard_mask, shard_rank;                    // multi-GPU sharded join (see GlueBuildParams)
};
template <int W>
__global__ void k_join_scatter(JoinScatterParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n_records; i += stride) {
        const uint32_t tag = P.glog_tag[i];
        if (tag == GTAG_EMPTY) continue;
        Kmer<W> jc;
        for (int j = 0; j < W; ++j) jc.w[j] = P.glog_keys[i * W + j];
        if (P.shard_mask && (mix32(jc.hash()) & P.shard_mask) != P.shard_rank) continue;
        join_bucket_put<W>(P.jfill, P.jrecs, P.log_jb, P.error, jc, tag);
    }
}
template <int W>
struct JoinWaveLds { uint64_t keys[2 * JB_CAP * W]; uint32_t a[2 * JB_CAP], b[2 * JB_CAP], conf[2 * JB_CAP]; };
struct JoinBucketParams {
    const uint32_t* jfill; const uint64_t* jrecs; uint32_t n_buckets;
    uint32_t* link; uint64_t* stats;                    // stats[0] junctions joined
    // multi-GPU sharded glue (k_dglue.h): the joined ends belong to other ranks -- instead of link[] writes, every joined
    // junction appends the two pairs (end, partner), (partner, end) to a list (one reservation per bucket)
    uint2* pairs; uint64_t* pair_cursor; uint64_t pair_cap; uint32_t* error;
};
template <int W>
__global__ void __launch_bounds__(JB_THREADS) k_join_bucket(JoinBucketParams P) {
    CDBG_SHARED JoinWaveLds<W> Ls[JB_THREADS / 64];
    constexpr uint32_t TSJ = 2 * JB_CAP;
    const int tid = threadIdx.x, lane = tid & 63, wave = (int)uni_u32((uint32_t)tid >> 6);
    JoinWaveLds<W>& L = Ls[wave];
    for (uint32_t i = lane; i < TSJ; i += 64) { L.keys[(uint64_t)i * W + (W - 1)] = KEY_EMPTY; L.a[i] = 0; L.b[i] = 0; L.conf[i] = 0; }
    CDBG_WAVE_SYNC();
    uint32_t joined = 0;
    uint64_t pchunk_base = 0; uint32_t pchunk_left = 0;  // (pair output: this wave's reservation -- one device atomic per JB_PAIR_CHUNK pairs,
                                                         //  not per bucket: 2 M buckets on ONE cursor word ran at the single-address rate, 25 ms)
    const uint32_t n_waves = gridDim.x * (JB_THREADS / 64);
    for (uint32_t bk = blockIdx.x * (JB_THREADS / 64) + (uint32_t)wave; bk < P.n_buckets; bk += n_waves) {
        uint32_t n = uni_u32(P.jfill[bk]); if (n > JB_CAP) n = JB_CAP;
        const uint64_t base = (uint64_t)bk * JB_CAP;
        // insert the bucket's records: find-or-insert the junction, then post the end / the confirmation
        for (uint32_t i = lane; i < n; i += 64) {
            Kmer<W> jc; uint32_t tag;
            if (W == 1) { const uint4 r = reinterpret_cast<const uint4*>(P.jrecs)[base + i]; jc.w[0] = (uint64_t)r.x | ((uint64_t)r.y << 32); tag = r.z; }
            else { for (int j = 0; j < W; ++j) jc.w[j] = P.jrecs[(base + i) * (W + 1) + j]; tag = (uint32_t)P.jrecs[(base + i) * (W + 1) + W]; }
            const uint64_t top = jc.w[W - 1];
            uint32_t s = (jc.hash() >> 7) & (TSJ - 1);   // (bits other than the ones that chose the bucket)
            bool done = false;
#pragma clang loop unroll(disable)
            do {
                // the same junction arrives in 2-3 records, possibly in sibling lanes of one iteration.  Multi-word keys:
                // the claimer marks the slot PENDING, writes the lower words and publishes inside the same iteration; a
                // sibling that meets PENDING looks again (single exit: see ktable_insert)
                uint64_t* const claim = &L.keys[(uint64_t)s * W + (W - 1)];
                const uint64_t old = atomic_cas_u64(claim, KEY_EMPTY, W == 1 ? top : (top | KEY_PENDING));
                bool advance = true;
                if (old == KEY_EMPTY) {
                    if (W > 1) {
                        for (int j = 0; j < W - 1; ++j) L.keys[(uint64_t)s * W + j] = jc.w[j];
                        CDBG_LDS_FENCE();
                        atomic_exch_u64(claim, top);
                    }
                    done = true; advance = false;
                } else if ((old & ~KEY_PENDING) == top) {
                    if (W > 1 && (old & KEY_PENDING)) { CDBG_SPIN_YIELD(); advance = false; }
                    else {
                        bool eq = true;
                        for (int j = 0; j < W - 1; ++j) eq &= (L.keys[(uint64_t)s * W + j] == jc.w[j]);
                        if (eq) { done = true; advance = false; }
                    }
                }
                if (advance) s = (s + 1) & (TSJ - 1);
            } while (!done);
            if (tag == GTAG_CONFIRM) atomic_or_u32(&L.conf[s], 1u);
            else {
                const uint32_t v = ((tag & ~GTAG_CONFBIT) + 1u) | (tag & GTAG_CONFBIT);
                if (atomic_cas_u32(&L.a[s], 0u, v) != 0u) atomic_cas_u32(&L.b[s], 0u, v);
            }
        }
        CDBG_WAVE_SYNC();
        // a junction with two ends, confirmed 1-1 by its owning bucket (a CONFIRM record, or the flag riding on an end)
        uint64_t pbase = 0;
        if (P.pairs) {                                   // pair output: count the bucket's joins first, ONE reservation
            uint32_t nj = 0;
            for (uint32_t s = lane; s < TSJ; s += 64) {
                if (L.keys[(uint64_t)s * W + (W - 1)] == KEY_EMPTY) continue;
                const uint32_t a = L.a[s], b = L.b[s];
                if (a && b && (((a | b) & 0x80000000u) || L.conf[s])) ++nj;
            }
            const uint32_t incl = wave_incl_sum_u32(nj), tot = wave_readlane_u32(incl, 63);
            if (2u * tot > pchunk_left) {                // (uniform) a new chunk; the tail of the old one stays unused: the list has gaps
                uint32_t lo = 0, hi = 0;
                const uint32_t want = 2u * tot > JB_PAIR_CHUNK ? 2u * tot : JB_PAIR_CHUNK;
                if (lane == 0) { const uint64_t o = atomic_add_u64(P.pair_cursor, (uint64_t)want); lo = (uint32_t)o; hi = (uint32_t)(o >> 32); }
                pchunk_base = ((uint64_t)wave_readlane_u32(hi, 0) << 32) | wave_readlane_u32(lo, 0); pchunk_left = want;
            }
            pbase = pchunk_base + 2ull * (incl - nj);
            pchunk_base += 2ull * tot; pchunk_left -= 2u * tot;
            if (tot && pbase + 2ull * nj > P.pair_cap) { *P.error = 9; pbase = ~0ull; }
        }
        for (uint32_t s = lane; s < TSJ; s += 64) {
            if (L.keys[(uint64_t)s * W + (W - 1)] == KEY_EMPTY) continue;
            const uint32_t a = L.a[s], b = L.b[s];
            if (a && b && (((a | b) & 0x80000000u) || L.conf[s])) {
                const uint32_t ea = (a & 0x7FFFFFFFu) - 1u, eb = (b & 0x7FFFFFFFu) - 1u;
                if (P.pairs) {
                    if (pbase != ~0ull) { uint2 p0; p0.x = ea; p0.y = eb; uint2 p1; p1.x = eb; p1.y = ea; P.pairs[pbase] = p0; P.pairs[pbase + 1] = p1; pbase += 2; }
                } else { P.link[ea] = eb; P.link[eb] = ea; }
                ++joined;
            }
            L.keys[(uint64_t)s * W + (W - 1)] = KEY_EMPTY; L.a[s] = 0; L.b[s] = 0; L.conf[s] = 0;
        }
        CDBG_WAVE_SYNC();
    }
    const uint64_t j64 = wave_sum_u64(joined);
    if (lane == 0 && j64) atomic_add_u64(&P.stats[0], j64);
}

// ---- (2) pointer jumping ----
// per traversal state one 16-byte record {nxt, acc, tail, minp}: a jump is ONE random 16-byte gather
// instead of four 4-byte gathers from four arrays
struct RankParams {
    uint32_t n_states;             // 2 * n_pieces
    const uint32_t* link; const uint32_t* piece_n;
    uint4* st_a; uint4* st_b;      // ping-pong: x = successor state, y = k-mers from this state up to (excluding) x,
                                   //            z = last state reached, w = smallest piece id seen (cycle leader election)
    uint32_t* changed;
};
__global__ void k_rank_init(RankParams P) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.n_states) return;
    uint4 v; v.x = P.link[e ^ 1u]; v.y = P.piece_n[e >> 1]; v.z = e; v.w = e >> 1;
    P.st_a[e] = v;
}
// one doubling round a -> b
__global__ void k_rank_jump(RankParams P) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.n_states) return;
    uint4 v = P.st_a[e];
    if (v.x != NONE32) {
        const uint4 t = P.st_a[v.x];
        v.y += t.y; v.z = t.z; v.x = t.x; v.w = t.w < v.w ? t.w : v.w;
        *P.changed = 1u;
    }
    P.st_b[e] = v;
}
// ---- (2b) the same doubling on 8-byte states, for the usual case without closed chains ----
// {x, y}: x = successor state, or RANK_TAIL | tail state once the end of the chain is known; y as above.  Half the
// bytes per round of the 16-byte version (the rounds stream every state, so they are bandwidth bound); the
// smallest-piece field that elects a cut point on closed chains is not carried: if this version does not
// converge, the caller falls back to the 16-byte version, which can cut cycles.
constexpr uint32_t RANK_TAIL = 0x80000000u;
struct Rank8Params { uint32_t n_states; const uint32_t* link; const uint32_t* piece_n; uint2* a; uint32_t* changed; };
__global__ void k_rank8_init(Rank8Params P) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.n_states) return;
    const uint32_t nxt = P.link[e ^ 1u];
    uint2 v; v.x = nxt == NONE32 ? (RANK_TAIL | e) : nxt; v.y = P.piece_n[e >> 1];
    P.a[e] = v;
}
// IN PLACE and asynchronous: a state is one aligned 8-byte word, so whatever a jump reads -- the successor's state of
// before or after its own update in this launch -- is a consistent {x, y} pair ("y k-mers ahead lies state x"), and
// jumping over either keeps the invariant.  Finished states cost one streaming read per launch and no write; a launch
// makes up to RANK8_JUMPS jumps per state (9 ping-pong rounds of 2.5 GB each before: 13.8 ms at config 3).
#ifndef CDBG_RANK8_JUMPS
#define CDBG_RANK8_JUMPS 5
#endif
constexpr int RANK8_JUMPS = CDBG_RANK8_JUMPS;
__global__ void k_rank8_jump(Rank8Params P) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.n_states) return;
    uint2 v = P.a[e];
    if (v.x & RANK_TAIL) return;
#pragma unroll 1
    for (int j = 0; j < RANK8_JUMPS && !(v.x & RANK_TAIL); ++j) {
        const uint2 t = P.a[v.x];
        v.y += t.y; v.x = t.x;
    }
    P.a[e] = v;
    if (!(v.x & RANK_TAIL)) *P.changed = 1u;
}
// k_unitig_heads / k_emit read the converged 8-byte states {RANK_TAIL | tail state, k-mers to the tail}; the 16-byte
// version (closed chains) is narrowed to that layout once it has converged
struct RankNarrowParams { uint32_t n_states; const uint4* st; uint2* out; };
__global__ void k_rank_narrow(RankNarrowParams P) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.n_states) return;
    const uint4 v = P.st[e];
    uint2 o; o.x = RANK_TAIL | v.z; o.y = v.y;
    P.out[e] = o;
}

// states still unresolved after ceil(log2(n_states))+1 rounds lie on closed chains:
// cut the chain at the left end of its smallest piece
struct CutParams { uint32_t n_states; const uint4* st; uint32_t* link; uint32_t* n_cycles; };
__global__ void k_cut_cycles(CutParams P) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.n_states) return;
    const uint4 v = P.st[e];
    if (v.x == NONE32) return;
    if ((e & 1u) || v.w != (e >> 1)) return;              // only the leader piece's left-end state acts
    const uint32_t partner = P.link[e];
    P.link[e] = NONE32;
    if (partner != NONE32) P.link[partner] = NONE32;
    atomic_add_u32(P.n_cycles, 1u);
}

// ---- (3) unitig heads: allocate id and output space ----
struct HeadParams {
    uint32_t n_states; int k;
    const uint32_t* link; const uint2* st;               // st[e] = {RANK_TAIL | tail state, k-mers to the tail}
    uint4* hinfo;                  // per state, written for head states only: {unitig id, k-mers of the unitig, output offset lo, hi}
                                   // (the spare ping-pong buffer of the ranking: one 16-byte gather for k_emit)
    uint64_t* unitig_off; uint32_t* unitig_len; uint64_t* unitig_kc;
    uint64_t unitig_cap, out_cap;
    uint64_t* n_unitigs; uint64_t* out_cursor; uint32_t* error;
    uint32_t own_lo, own_hi;       // multi-GPU, owner-sharded emission: only heads e with own_lo <= e < own_hi get a unitig here
};
// One workgroup per HEADS_PER_WG consecutive states (HEADS_ITEMS per lane): the heads of the workgroup get
// consecutive unitig ids and output space from ONE device reservation; positions inside the batch come from a
// workgroup-wide exclusive scan of the per-lane (count, length) sums -- no per-head atomics.
constexpr int HEADS_ITEMS = 8;
constexpr int HEADS_PER_WG = GLUE_THREADS * HEADS_ITEMS;
__global__ void k_unitig_heads(HeadParams P) {
    CDBG_SHARED uint32_t s_wn[GLUE_THREADS / 64]; CDBG_SHARED uint64_t s_wl[GLUE_THREADS / 64];
    CDBG_SHARED uint64_t s_ubase, s_obase;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * HEADS_PER_WG;
    // head of the chosen direction: no predecessor, and the larger tail of the two directions
    // (piece ids inside unused reservation gaps have piece_n == 0 => acc == 0: not a unitig)
    uint32_t len[HEADS_ITEMS]; uint32_t cnt = 0; uint64_t sum = 0;
#pragma unroll
    for (int i = 0; i < HEADS_ITEMS; ++i) {
        const uint64_t e = base + (uint64_t)i * GLUE_THREADS + tid;
        len[i] = 0;
        if (e < P.n_states && P.link[e] == NONE32) {
            const uint2 v = P.st[e];
            if (v.y != 0 && v.x > P.st[e ^ 1u].x) {
                if (e >= P.own_lo && e < P.own_hi) { len[i] = v.y + (uint32_t)P.k - 1u; ++cnt; sum += len[i]; }
                else { uint4 h; h.x = NONE32; h.y = 0; h.z = 0; h.w = 0; P.hinfo[e] = h; }   // another rank's unitig: its pieces are skipped by k_emit
            }
        }
    }
    // exclusive scan of (cnt, sum) over the workgroup: wave shuffles, then the wave totals through LDS
    uint32_t icnt = cnt; uint64_t isum = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t c = __shfl_up(icnt, d); const uint64_t l = __shfl_up(isum, d);
        if (lane >= d) { icnt += c; isum += l; }
    }
    if (lane == 63) { s_wn[wave] = icnt; s_wl[wave] = isum; }
    __syncthreads();
    uint32_t my_i = icnt - cnt, tot_n = 0; uint64_t my_o = isum - sum, tot_l = 0;
    for (int w = 0; w < GLUE_THREADS / 64; ++w) {
        if (w < wave) { my_i += s_wn[w]; my_o += s_wl[w]; }
        tot_n += s_wn[w]; tot_l += s_wl[w];
    }
    if (tid == 0 && tot_n) {                               // one reservation per workgroup
        s_ubase = atomic_add_u64(P.n_unitigs, (uint64_t)tot_n);
        s_obase = atomic_add_u64(P.out_cursor, tot_l);
    }
    __syncthreads();
    if (!cnt) return;
    uint64_t uid = s_ubase + my_i, off = s_obase + my_o;
#pragma unroll
    for (int i = 0; i < HEADS_ITEMS; ++i) {
        if (!len[i]) continue;
        const uint64_t e = base + (uint64_t)i * GLUE_THREADS + tid;
        uint4 h; h.x = NONE32; h.y = len[i] - ((uint32_t)P.k - 1u); h.z = (uint32_t)off; h.w = (uint32_t)(off >> 32);
        if (uid >= P.unitig_cap || off + len[i] > P.out_cap) *P.error = 4;
        else { h.x = (uint32_t)uid; P.unitig_off[uid] = off; P.unitig_len[uid] = len[i]; P.unitig_kc[uid] = 0; }
        P.hinfo[e] = h;
        ++uid; off += len[i];
    }
}

// ---- (4) emit: one lane per piece ----
struct EmitParams {
    uint32_t n_pieces; int k;
    const uint2* st; const uint4* hinfo;
    const uint32_t* piece_n; const uint64_t* piece_kc; const uint64_t* piece_boff; const uint8_t* piece_bases;
    uint64_t* unitig_kc; uint8_t* out;
    const uint32_t* piece_ab; uint32_t* unitig_ab;   // optional per-k-mer abundances, indexed like the bases (k-mer ending at that base)
};
CDBG_DEV uint8_t comp_ascii(uint8_t c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; }
// 8 ASCII bases: reversed and complemented.  A 0x41 <-> T 0x54 differ by 0x15, C 0x43 <-> G 0x47 by 0x04; bit 1 tells the pairs apart
CDBG_DEV uint64_t comp_ascii8_rev(uint64_t w) {
    const uint64_t m = (w >> 1) & 0x0101010101010101ull;
    return __builtin_bswap64(w ^ 0x1515151515151515ull ^ (m * 0x11ull));
}
__global__ void k_emit(EmitParams P) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.n_pieces) return;
    const uint32_t e0 = 2 * p, e1 = 2 * p + 1;
    // direction d visits this piece in state e; its reverse visits it in e^1; chosen: larger tail
    const uint4 s01 = reinterpret_cast<const uint4*>(P.st)[p];               // both states of the piece: one 16-byte load
    const uint32_t z0 = s01.x & ~RANK_TAIL, y0 = s01.y, z1 = s01.z & ~RANK_TAIL, y1 = s01.w;
    const uint32_t e = (z0 > z1) ? e0 : e1;
    const uint32_t head = ((z0 > z1) ? z1 : z0) ^ 1u;                     // head of d = mirror of the tail of the reverse direction
    const uint32_t n = P.piece_n[p];
    if (n == 0) return;                                    // reservation gap
    const uint4 h = P.hinfo[head];                          // ONE gather: unitig id, its k-mer count, its output offset
    const uint32_t uid = h.x;
    if (uid == NONE32) return;
    const uint32_t koff = h.y - ((z0 > z1) ? y0 : y1);                     // k-mers before this piece
    const uint32_t nb = n + (uint32_t)P.k - 1u;
    const uint8_t* src = P.piece_bases + P.piece_boff[p];
    const uint64_t uoff = (uint64_t)h.z | ((uint64_t)h.w << 32);
    uint8_t* dst = P.out + uoff + koff;
    const uint32_t skip = koff ? (uint32_t)P.k - 1u : 0u;  // the overlap was written by the previous piece
    // The fragment [skip, nb) of a lane lands at its own byte address of the output (7.5 bytes on average for a
    // piece behind the head): byte stores made every byte a 32-byte sector write (9.4 GB written for 0.9 GB of
    // unitigs).  8 bases per load / store at any alignment, the last 8 overlapping; short fragments as 4 + 2 + 1.
    const bool fwd = (e & 1u) == END_LEFT;
    uint32_t i = skip;
    if (nb - skip >= 8u) {
        for (;;) {
            if (i + 8u > nb) { if (i == nb) break; i = nb - 8u; }
            const uint64_t w = fwd ? ld_unaligned_u64(src + i) : comp_ascii8_rev(ld_unaligned_u64(src + (nb - 8u - i)));
            st_unaligned_u64(dst + i, w);
            i += 8u;
        }
    } else {
        uint64_t w = 0; const uint32_t r = nb - skip;
        for (uint32_t j = 0; j < r; ++j) w |= (uint64_t)(fwd ? src[skip + j] : comp_ascii(src[nb - 1 - skip - j])) << (8 * j);
        if (r & 4u) { st_unaligned_u32(dst + i, (uint32_t)w); w >>= 32; i += 4u; }
        if (r & 2u) { st_unaligned_u16(dst + i, (uint16_t)w); w >>= 16; i += 2u; }
        if (r & 1u) dst[i] = (uint8_t)w;
    }
    if (P.piece_ab) {                                      // -all-abundance-counts: k-mer t of the piece -> k-mer koff+t (or mirrored)
        const uint32_t* sa = P.piece_ab + P.piece_boff[p] + (P.k - 1);
        uint32_t* da = P.unitig_ab + uoff + koff + (P.k - 1);
        if ((e & 1u) == END_LEFT) { for (uint32_t t = 0; t < n; ++t) da[t] = sa[t]; }
        else { for (uint32_t t = 0; t < n; ++t) da[t] = sa[n - 1 - t]; }
    }
    atomic_add_u64(&P.unitig_kc[uid], P.piece_kc[p]);     // (77 M device atomics at config 3: 0.5 ms of the kernel, measured by leaving them out)
}

// ---- fetch helper: unitigs [first, first + n) gathered gap-free into one buffer (one wave per unitig), so that a partial
// cdbg_fetch_unitigs is ONE device-to-host copy instead of one per unitig ----
struct GatherUnitigParams { uint64_t n; const uint64_t* src_off; const uint32_t* len; const uint64_t* dst_off; const uint8_t* src; uint8_t* dst; };
__global__ void k_gather_unitigs(GatherUnitigParams P) {
    const uint64_t u = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (u >= P.n) return;
    const uint32_t lane = threadIdx.x & 63u, len = P.len[u];
    const uint8_t* s = P.src + P.src_off[u]; uint8_t* d = P.dst + P.dst_off[u];
    for (uint32_t i = lane; i < len; i += 64) d[i] = s[i];
}

// ---- fetch helpers: solid k-mers as ASCII (stage-1 parity surface); one lane per partition segment ----
struct DecodeParams { const uint64_t* keys; const uint32_t* cnt; const uint64_t* seg_off; const uint32_t* seg_n; uint64_t n_parts;
                      int k, W; uint8_t* out_kmers; uint32_t* out_cnt; uint64_t* n_out; uint64_t cap; };
__global__ void k_decode_solid(DecodeParams P) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.n_parts) return;
    const uint64_t so = P.seg_off[p];
    for (uint32_t e = 0; e < P.seg_n[p]; ++e) {
        const uint64_t i = so + e;
        const uint32_t c = P.cnt[i];
        if (c & TRAV_FLAG) continue;                       // traveller copies are not part of the k-mer set
        const uint64_t o = atomic_add_u64(P.n_out, 1ULL);
        if (o >= P.cap) continue;
        for (int b = 0; b < P.k; ++b) {
            const int pos = 2 * (P.k - 1 - b);
            P.out_kmers[o * (uint64_t)(P.k + 1) + b] = (uint8_t)("ACGT"[(P.keys[i * P.W + (pos >> 6)] >> (pos & 63)) & 3u]);
        }
        P.out_kmers[o * (uint64_t)(P.k + 1) + P.k] = 0;
        P.out_cnt[o] = c;
    }
}

// ---- multi-GPU merge: append one rank's pieces / glue log to the merged arrays, renumbering ----
struct MergeParams {
    uint64_t n_pieces, n_glog, piece_base, bases_base, glog_base; int W;
    const uint32_t* src_n; const uint64_t* src_kc; const uint64_t* src_boff; const uint64_t* src_gkeys; const uint32_t* src_gtag;   // src_boff: offsets inside the source's base array (packed exchange: inside its gap-free stream)
    uint32_t* dst_n; uint64_t* dst_kc; uint64_t* dst_boff; uint64_t* dst_gkeys; uint32_t* dst_gtag;
};
// ---- packed piece bases for the wire: the pieces' bases as ONE gap-free stream, 4 bases per byte ----
// lens[i] = bases of piece i (0 for ids inside reservation gaps)
struct PackLenParams { uint64_t n_pieces; int k; const uint32_t* piece_n; uint32_t* lens; };
__global__ void k_pack_lens(PackLenParams P) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_pieces) return;
    const uint32_t n = P.piece_n[i];
    P.lens[i] = n ? n + (uint32_t)P.k - 1u : 0u;
}
// sender, step 1: squeeze the reservation gaps out (one lane per piece; only this rank's own pieces)
struct SqueezeParams { uint64_t n_pieces; const uint32_t* lens; const uint64_t* uoff; const uint64_t* boff; const uint8_t* bases; uint8_t* dense; };
__global__ void k_squeeze_bases(SqueezeParams P) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_pieces) return;
    const uint32_t len = P.lens[i];
    const uint8_t* src = P.bases + P.boff[i];
    uint8_t* dst = P.dense + P.uoff[i];
    uint32_t j = 0;
    for (; j + 8 <= len; j += 8) st_unaligned_u64(dst + j, ld_unaligned_u64(src + j));     // (a byte per store was 13 ms for 0.9 GB)
    for (; j < len; ++j) dst[j] = src[j];
}
// sender, step 2 / receiver: streaming 64 ASCII bases <-> 16 packed bytes per lane (dense is padded to 64 bytes)
// -all-abundance-counts across ranks: the abundances of a rank's pieces as one gap-free u32 stream, n values per piece
// (aoff = exclusive scan of piece_n).  dir 0: piece_ab (indexed like the bases) -> stream; dir 1: stream -> piece_ab
// of the merged numbering (bases of the piece at boff_base + uoff[p])
struct AbStreamParams { uint64_t n_pieces; int k; int dir; const uint32_t* piece_n; const uint64_t* aoff; const uint64_t* uoff; const uint64_t* boff; uint64_t boff_base;
                        uint32_t* piece_ab; uint32_t* stream; };
__global__ void k_ab_stream(AbStreamParams P) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.n_pieces) return;
    const uint32_t n = P.piece_n[p];
    if (!n) return;                                        // reservation gap
    uint32_t* const st = P.stream + P.aoff[p];
    uint32_t* const ab = P.piece_ab + (P.boff ? P.boff[p] : P.boff_base + P.uoff[p]) + (P.k - 1);
    if (P.dir == 0) { for (uint32_t t = 0; t < n; ++t) st[t] = ab[t]; }
    else { for (uint32_t t = 0; t < n; ++t) ab[t] = st[t]; }
}
struct StreamPackParams { uint64_t n_chunks; const uint8_t* ascii; uint8_t* packed; uint64_t n_bases; };
__global__ void k_pack_stream(StreamPackParams P) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P.n_chunks) return;
    uint32_t out[4] = {0, 0, 0, 0};
    const uint4* src = reinterpret_cast<const uint4*>(P.ascii + 64 * c);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 v = src[q];
        const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t x = w[t], code = ((x >> 1) ^ (x >> 2)) & 0x03030303u;      // 2-bit code of each of 4 ASCII bases
            out[q] |= ((code & 3u) | ((code >> 6) & 0xCu) | ((code >> 12) & 0x30u) | ((code >> 18) & 0xC0u)) << (8 * t);
        }
    }
    uint4 o; o.x = out[0]; o.y = out[1]; o.z = out[2]; o.w = out[3];
    reinterpret_cast<uint4*>(P.packed)[c] = o;
}
struct StreamUnpackParams { uint64_t n_chunks; const uint8_t* packed; uint8_t* ascii; uint64_t n_bases; };
__global__ void k_unpack_stream(StreamUnpackParams P) {        // the last chunk may be partial
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P.n_chunks) return;
    const uint4 v = reinterpret_cast<const uint4*>(P.packed)[c];
    const uint32_t in[4] = { v.x, v.y, v.z, v.w };
    uint8_t* dst = P.ascii + 64 * c;
    const uint64_t left = P.n_bases - 64 * c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t w[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t b = (in[q] >> (8 * t)) & 0xFFu;
            uint32_t x = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) x |= (uint32_t)(uint8_t)("ACGT"[(b >> (2 * u)) & 3u]) << (8 * u);
            w[t] = x;
        }
        if (left >= (uint64_t)(16 * q + 16)) { uint4 o; o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3]; *reinterpret_cast<uint4*>(dst + 16 * q) = o; }
        else { for (int j = 0; j < 16; ++j) if ((uint64_t)(16 * q + j) < left) dst[16 * q + j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3))); }
    }
}

__global__ void k_merge_append(MergeParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t i = i0; i < P.n_pieces; i += stride) {
        P.dst_n[P.piece_base + i] = P.src_n[i];
        P.dst_kc[P.piece_base + i] = P.src_kc[i];
        P.dst_boff[P.piece_base + i] = P.src_boff[i] + P.bases_base;
    }
    for (uint64_t i = i0; i < P.n_glog; i += stride) {
        const uint32_t t = P.src_gtag[i];
        // piece-end ids move with their piece: end = 2 * piece + side
        P.dst_gtag[P.glog_base + i] = (t == GTAG_EMPTY || t == GTAG_CONFIRM) ? t : (((t & ~GTAG_CONFBIT) + (uint32_t)(2 * P.piece_base)) | (t & GTAG_CONFBIT));
        for (int j = 0; j < P.W; ++j) P.dst_gkeys[(P.glog_base + i) * P.W + j] = P.src_gkeys[i * P.W + j];
    }
}

}  // namespace cdbg
