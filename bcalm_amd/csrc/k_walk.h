// k_walk.h -- stage 3, single-GPU fast path: the chains of pieces are WALKED from their heads instead of ranked.
//
// Same job as k_rank8_* + k_unitig_heads + k_emit (k_glue.h; SURVEY.md section 8 row a9, the record writer of
// /root/reference/README.md:62-72), taken when the chains are short -- which they are whenever the reads carry
// errors: 7.5 pieces per unitig at BASELINE config 3.  Pointer jumping pays log2(chain) random gathers for EVERY
// traversal state of BOTH directions, a per-head record, and then one gather + one scattered partial-sector store per
// piece (profiles/r03_pmc_hbm_traffic_per_kernel_cfg3.csv: 52 GB of traffic for 2.3 GB of algorithmic bytes).  A
// walk pays one gather per piece and direction to measure the chain, and one more (+ the piece's bases) for the
// direction that is kept; the lanes of a wave hold unitigs of one span of heads, so what they write lands in one
// window of the arena instead of 77 M places.
//
//   piece record (32 bytes, one aligned line half): w0 = successor state when the piece is entered through its LEFT
//   end, w1 = ... through its RIGHT end (NONE32: the chain ends here), w2 = k-mers, w3/w4 = KC, w5/w6 = offset of
//   its bases.  k_walk_init builds the records from the piece arrays and link[] in one streaming pass and lists the
//   head states; k_walk_measure / k_walk_place / k_walk_copy do the rest.
//
// First version (one workgroup per 1024 pieces, heads compacted in LDS, one chain per lane): 24.6 ms at config 3 against
// 18.5 ms of ranking + heads + emit -- a quarter of the lanes busy (the longest of 64 chains sets the pace) at 4 waves per
// SIMD.  This version keeps every lane on a chain and needs no LDS: 3.2 + 0.6 + 6.8 ms.
//
// What a walk cannot do -- chains longer than max_steps (error-free reads: one unitig of 10^5 pieces), closed chains
// (no head: the visited-piece count falls short) -- is reported to the host, which runs the ranking path of k_glue.h on
// the same link[]; the context then stays on that path.
#pragma once
#include "k_glue.h"

namespace cdbg {

constexpr int WALK_THREADS = 256;
constexpr int WALK_PPT = 8;                                   // consecutive pieces per thread in the head search
constexpr int WALK_PIECES = WALK_THREADS * WALK_PPT;          // pieces per workgroup of the head search
constexpr int WALK_SPAN = 1024;                               // head slots one wave walks (k_walk_measure / k_walk_copy); one workgroup places (k_walk_place)
constexpr int WALK_SPT = WALK_SPAN / WALK_THREADS;

// ---- (A) piece records and head states (in state order inside a workgroup's pieces; the workgroups' runs in the order of their reservations) ----
struct WalkInitParams { uint32_t n_pieces; const uint32_t* piece_n; const uint64_t* piece_kc; const uint64_t* piece_boff; const uint32_t* link;
                        uint4* rec; uint32_t* heads; uint64_t* n_heads; uint64_t* n_live; };
__global__ void __launch_bounds__(WALK_THREADS) k_walk_init(WalkInitParams P) {
    CDBG_SHARED uint32_t s_wn[WALK_THREADS / 64], s_wl[WALK_THREADS / 64]; CDBG_SHARED uint64_t s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a wave takes 64 * WALK_PPT consecutive pieces, 64 at a time: coalesced loads and record stores, heads in state order
    const uint32_t p0 = blockIdx.x * (uint32_t)WALK_PIECES + (uint32_t)wave * (64u * WALK_PPT) + (uint32_t)lane;
    uint32_t hs[2 * WALK_PPT], at[WALK_PPT]; uint32_t run = 0, live = 0;
#pragma unroll
    for (int i = 0; i < WALK_PPT; ++i) {
        const uint32_t p = p0 + 64u * (uint32_t)i;
        hs[2 * i] = NONE32; hs[2 * i + 1] = NONE32;
        uint32_t cnt = 0;
        if (p < P.n_pieces) {
            const uint32_t n = P.piece_n[p];                   // 0: a gap between two reservations of the compaction kernels
            uint4 a, b; a.x = NONE32; a.y = NONE32; a.z = n; a.w = 0; b.x = 0; b.y = 0; b.z = 0; b.w = 0;
            if (n) {
                const uint64_t kc = P.piece_kc[p], bo = P.piece_boff[p];
                const uint2 l = reinterpret_cast<const uint2*>(P.link)[p];      // partners of the left / right end
                a.x = l.y; a.y = l.x;                          // entered through the left end, a walk leaves through the right one
                a.w = (uint32_t)kc; b.x = (uint32_t)(kc >> 32); b.y = (uint32_t)bo; b.z = (uint32_t)(bo >> 32);
                ++live;
                if (l.x == NONE32) { hs[2 * i] = 2 * p; ++cnt; }          // nothing joined at the left end: entering there starts a chain
                if (l.y == NONE32) { hs[2 * i + 1] = 2 * p + 1; ++cnt; }
            }
            P.rec[2 * (uint64_t)p] = a; P.rec[2 * (uint64_t)p + 1] = b;
        }
        const uint32_t incl = wave_incl_sum_u32(cnt);
        at[i] = run + incl - cnt;
        run += wave_readlane_u32(incl, 63);
    }
    const uint32_t lsum = wave_readlane_u32(wave_incl_sum_u32(live), 63);
    if (lane == 0) { s_wn[wave] = run; s_wl[wave] = lsum; }
    __syncthreads();
    uint32_t wbase = 0, tot = 0, ltot = 0;
    for (int w = 0; w < WALK_THREADS / 64; ++w) { if (w < wave) wbase += s_wn[w]; tot += s_wn[w]; ltot += s_wl[w]; }
    if (tid == 0) {                                        // (two device atomics per workgroup: per wave they ran at the single-address rate, 4 ms)
        s_base = tot ? atomic_add_u64(P.n_heads, (uint64_t)tot) : 0ull;
        if (ltot) atomic_add_u64(P.n_live, (uint64_t)ltot);
    }
    __syncthreads();
    uint32_t* const out = P.heads + s_base + wbase;
#pragma unroll
    for (int i = 0; i < WALK_PPT; ++i) {
        uint32_t j = at[i];
        if (hs[2 * i] != NONE32) out[j++] = hs[2 * i];
        if (hs[2 * i + 1] != NONE32) out[j] = hs[2 * i + 1];
    }
}

// bases [skip, nb) of a piece to their place in the unitig: 8 bases per load / store at any alignment, the last 8
// overlapping; short fragments as 4 + 2 + 1 (k_emit's copy)
CDBG_DEV void walk_copy(const uint8_t* src, uint8_t* dst, uint32_t nb, uint32_t skip, bool fwd) {
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
}

// The two walking kernels keep every lane of a wave on a chain: a wave owns WALK_SPAN consecutive head slots, and a lane that
// has finished its chain takes the next unclaimed slot of the wave (a ballot and a popcount: no atomics, no LDS).  With one
// chain per lane the longest of 64 chains -- 4 times the average -- would set the pace of the wave.
CDBG_DEV uint32_t walk_lane_rank(uint64_t mask, int lane) { return (uint32_t)__popcll(mask & ((1ULL << lane) - 1ULL)); }

// ---- (B) measure: tail and k-mers of the chain behind every head; hlen = k-mers when this direction is the one that is kept
// (the rule of k_unitig_heads: the larger tail; the output does not depend on which path glued it), else 0 ----
struct WalkMeasureParams { const uint4* rec; const uint32_t* heads; const uint64_t* n_heads; uint32_t* hlen; uint32_t max_steps; uint32_t* giveup; };
__global__ void __launch_bounds__(WALK_THREADS) k_walk_measure(WalkMeasureParams P) {
    const int lane = threadIdx.x & 63;
    const uint64_t n_heads = *P.n_heads;
    const uint64_t wv = ((uint64_t)blockIdx.x * WALK_THREADS + threadIdx.x) >> 6;
    uint64_t next = wv * WALK_SPAN; const uint64_t end = next + WALK_SPAN < n_heads ? next + WALK_SPAN : n_heads;
    bool active = false; uint32_t e = 0, h = 0, len = 0, steps = 0; uint64_t idx = 0;
    for (;;) {
        const uint64_t need = __ballot(!active);
        if (next < end) {                                  // (uniform)
            const uint64_t my = next + walk_lane_rank(need, lane);
            if (!active && my < end) { idx = my; h = P.heads[my]; e = h; len = 0; steps = 0; active = true; }
            next += (uint64_t)__popcll(need);
        }
        if (!__any(active)) break;
        if (active) {
            const uint4 a = P.rec[2 * (uint64_t)(e >> 1)];
            len += a.z;
            const uint32_t nx = (e & 1u) ? a.y : a.x;
            if (nx == NONE32) { P.hlen[idx] = e > (h ^ 1u) ? len : 0u; active = false; }       // (an isolated piece: the walk from its right end is kept)
            else if (++steps > P.max_steps) { *P.giveup = 1u; P.hlen[idx] = 0u; active = false; }
            else e = nx;
        }
    }
}

// ---- (C) place: the kept heads of a span get consecutive unitig ids and consecutive room in the arena from ONE device
// reservation per span; hlen becomes unitig id + 1 (0: not kept), hoff the unitig's offset ----
struct WalkPlaceParams { const uint64_t* n_heads; uint32_t* hlen; uint64_t* hoff; int k; uint64_t* n_unitigs; uint64_t* out_cursor; uint64_t unitig_cap, out_cap; uint32_t* error; };
__global__ void __launch_bounds__(WALK_THREADS) k_walk_place(WalkPlaceParams P) {
    CDBG_SHARED uint32_t s_wn[WALK_THREADS / 64]; CDBG_SHARED uint64_t s_wl[WALK_THREADS / 64];
    CDBG_SHARED uint64_t s_ubase, s_obase;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t n_heads = *P.n_heads;
    if ((uint64_t)blockIdx.x * WALK_SPAN >= n_heads) return;   // (uniform; the launch covers the upper bound of two heads per piece)
    const uint64_t i0 = (uint64_t)blockIdx.x * WALK_SPAN + (uint64_t)tid * WALK_SPT;
    uint32_t kl[WALK_SPT]; uint32_t kn = 0; uint64_t ksum = 0;
#pragma unroll
    for (int i = 0; i < WALK_SPT; ++i) {
        kl[i] = i0 + i < n_heads ? P.hlen[i0 + i] : 0u;
        if (kl[i]) { ++kn; ksum += (uint64_t)kl[i] + (uint64_t)P.k - 1u; }
    }
    uint32_t icnt = kn; uint64_t isum = ksum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t c = __shfl_up(icnt, (unsigned)d); const uint64_t l = __shfl_up(isum, (unsigned)d);
        if (lane >= d) { icnt += c; isum += l; }
    }
    if (lane == 63) { s_wn[wave] = icnt; s_wl[wave] = isum; }
    __syncthreads();
    uint32_t my_i = icnt - kn, tot_n = 0; uint64_t my_o = isum - ksum, tot_l = 0;
    for (int w = 0; w < WALK_THREADS / 64; ++w) {
        if (w < wave) { my_i += s_wn[w]; my_o += s_wl[w]; }
        tot_n += s_wn[w]; tot_l += s_wl[w];
    }
    if (tid == 0 && tot_n) {
        s_ubase = atomic_add_u64(P.n_unitigs, (uint64_t)tot_n);
        s_obase = atomic_add_u64(P.out_cursor, tot_l);
        if (s_ubase + tot_n > P.unitig_cap || s_obase + tot_l > P.out_cap) *P.error = 4;
    }
    __syncthreads();
    if (!tot_n) return;
    const bool fits = s_ubase + tot_n <= P.unitig_cap && s_obase + tot_l <= P.out_cap;
    uint64_t uid = s_ubase + my_i, off = s_obase + my_o;
#pragma unroll
    for (int i = 0; i < WALK_SPT; ++i) {
        if (!kl[i]) continue;
        P.hlen[i0 + i] = fits ? (uint32_t)uid + 1u : 0u; P.hoff[i0 + i] = off;
        ++uid; off += (uint64_t)kl[i] + (uint64_t)P.k - 1u;
    }
}

// ---- (D) copy: the kept heads walk again; lanes of a wave hold unitigs of one span, i.e. of one window of the arena ----
struct WalkCopyParams {
    int k; const uint4* rec; const uint8_t* piece_bases;
    const uint32_t* heads; const uint64_t* n_heads; const uint32_t* hlen; const uint64_t* hoff;
    uint64_t* unitig_off; uint32_t* unitig_len; uint64_t* unitig_kc; uint8_t* out; uint64_t* visited;
    const uint32_t* piece_ab; uint32_t* unitig_ab;          // optional per-k-mer abundances, indexed like the bases
};
__global__ void __launch_bounds__(WALK_THREADS) k_walk_copy(WalkCopyParams P) {
    const int lane = threadIdx.x & 63;
    const uint64_t n_heads = *P.n_heads;
    const uint64_t wv = ((uint64_t)blockIdx.x * WALK_THREADS + threadIdx.x) >> 6;
    uint64_t next = wv * WALK_SPAN; const uint64_t end = next + WALK_SPAN < n_heads ? next + WALK_SPAN : n_heads;
    bool active = false; uint32_t e = 0, koff = 0, uid = 0, seen = 0; uint64_t kc = 0, uoff = 0;
    for (;;) {
        const uint64_t need = __ballot(!active);
        if (next < end) {                                  // (uniform)
            const uint64_t my = next + walk_lane_rank(need, lane);
            if (!active && my < end) {
                const uint32_t u1 = P.hlen[my];
                if (u1) { uid = u1 - 1u; uoff = P.hoff[my]; e = P.heads[my]; koff = 0; kc = 0; active = true; }
            }
            next += (uint64_t)__popcll(need);
        }
        if (!__any(active)) { if (next < end) continue; break; }   // (a claim may have found only heads of the other direction)
        if (active) {
            const uint32_t p = e >> 1;
            const uint4 a = P.rec[2 * (uint64_t)p], b = P.rec[2 * (uint64_t)p + 1];
            const uint32_t n = a.z, nb = n + (uint32_t)P.k - 1u;
            const uint64_t boff = (uint64_t)b.y | ((uint64_t)b.z << 32);
            kc += (uint64_t)a.w | ((uint64_t)b.x << 32);
            const bool fwd = (e & 1u) == END_LEFT;
            walk_copy(P.piece_bases + boff, P.out + uoff + koff, nb, koff ? (uint32_t)P.k - 1u : 0u, fwd);   // (the overlap was written by the previous piece)
            if (P.piece_ab) {                              // -all-abundance-counts: k-mer t of the piece -> k-mer koff + t (or mirrored)
                const uint32_t* sa = P.piece_ab + boff + (P.k - 1);
                uint32_t* da = P.unitig_ab + uoff + koff + (P.k - 1);
                if (fwd) { for (uint32_t t = 0; t < n; ++t) da[t] = sa[t]; }
                else { for (uint32_t t = 0; t < n; ++t) da[t] = sa[n - 1 - t]; }
            }
            koff += n; ++seen;
            const uint32_t nx = (e & 1u) ? a.y : a.x;
            if (nx == NONE32) { P.unitig_off[uid] = uoff; P.unitig_len[uid] = koff + (uint32_t)P.k - 1u; P.unitig_kc[uid] = kc; active = false; }
            else e = nx;
        }
    }
    const uint64_t s64 = wave_sum_u64((uint64_t)seen);
    if (lane == 0 && s64) atomic_add_u64(P.visited, s64);
}

}  // namespace cdbg
