// k_verify.h -- the unitig definition checked on the resident result, at any size, without the oracle's code.
//
// /root/reference/bidirected-graphs-in-bcalm2/bidirected-graphs-in-bcalm2.md:64 (nodes = the distinct canonical k-mers
// that pass the abundance filter, README.md:23-25), :83-88 (a unitig is a path that repeats no vertex and whose inner
// junctions are 1-in / 1-out), :92 (maximal unitigs).  cdbg_verify (include/cdbg.h) runs three passes over what the
// stages left in HBM:
//   k_verify_unitig_kmers  every k-mer position of every unitig -> canonical k-mer -> two independent 64-bit mixes,
//                          summed (commutative) together with the number of positions
//   k_verify_solid         the same two sums over the HOME entries of the solid table the count stage wrote
//   k_verify_maximal       from the link table (k_links.h): ends e != f of two DIFFERENT unitigs that are each other's
//                          only link -- a junction that is 1-in / 1-out and was left unglued
// Equal position counts and equal sums <=> the multiset of k-mers spelled by the unitigs IS the solid set (no k-mer
// twice, none missing, none invented; a difference survives both 64-bit sums with probability ~2^-128); no mergeable
// pair <=> every unitig is maximal.  With the conservation sums of cdbg_digest (KC, lengths) that is the definition.
// The unitig pass reads the sequences the caller would fetch, the solid pass the keys as counted: neither shares code
// with the compaction or the glue (only Kmer<W>::rc and the mixer).
//
// EDGE CONSERVATION (round 5; cdbg_verify_edges) closes the inner-junction half of the definition (.md:85: "for every
// 0 < i < n the only edges incident on v_i are e_{i-1}, e_i and their mirrors") at any size.  The maximality pass above only
// sees unitig ENDS: a unitig that runs THROUGH a branching node passes it.  So the edges of the solid k-mer graph are counted
// from the count stage's keys, with nothing of the scan's routing or the compaction in between:
//   k_verify_edge_insert   both ends of every HOME solid k-mer register at the canonical (k-1)-mer they leave through, in one
//                          global open-address table (exact keys): a ends on the key's strand, b on the other, c at a key
//                          that is its own reverse complement
//   k_verify_edge_sum      D = sum over the junctions of 2 a b + c^2 = sum over all k-mer ends of the ends they see across
// and compared with what the unitig set accounts for: every inner adjacency of a unitig is one junction seen from both sides
// (2 per adjacency = 2 sum(LN - k)), every unitig end sees its links (L = the link table's size, built over unitig ENDS only).
// With i adjacencies of unitigs running through a junction of (a, b) ends the unitig side counts 2 (a - i)(b - i) + 2 i, i.e.
// 2 i (a + b - i - 1) less than 2 a b: zero exactly for i = 0 or a = b = i = 1, positive otherwise (a self-complementary
// junction: (c - 2 i)^2 + 2 i against c^2, less for every i >= 1) -- the differences cannot cancel, so
//   D == L + 2 sum(LN - k)   <=>   every inner junction of every unitig is 1-in / 1-out.
#pragma once
#include "k_links.h"

namespace cdbg {

struct VerifyParams {
    uint64_t n_unitigs; int k;
    const uint64_t* unitig_off; const uint32_t* unitig_len; const uint8_t* bases;
    const uint64_t* seg_off; const uint32_t* seg_n; const uint64_t* solid_keys; const uint32_t* solid_cnt; uint64_t n_parts;
    const uint64_t* link_off; const uint32_t* link_to;
    uint64_t* out;     // [0] k-mer positions in unitigs [1] sum mixA [2] sum mixB   [3..5] the same over the solid table
                       // [6] mergeable end pairs (each pair counted from both ends)  [7] unitigs whose two ends are each other's only link (closed chains, cut open)
};

template <int W>
CDBG_DEV void verify_mix(const Kmer<W>& c, uint64_t& a, uint64_t& b) {
    uint64_t h = 0x243F6A8885A308D3ULL;
#pragma unroll
    for (int i = 0; i < W; ++i) h = mix64(h ^ c.w[i]);
    a = h; b = mix64(h ^ 0xA4093822299F31D0ULL);
}

template <int W>
__global__ void k_verify_unitig_kmers(VerifyParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t n = 0, sa = 0, sb = 0;
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < P.n_unitigs; u += stride) {
        const uint8_t* s = P.bases + P.unitig_off[u];
        const uint32_t len = P.unitig_len[u];
        Kmer<W> x = Kmer<W>::zero();
        for (uint32_t i = 0; i < len; ++i) {
            x.push_right(P.k, base_code(s[i]));
            if (i + 1 >= (uint32_t)P.k) {
                uint64_t a, b; verify_mix<W>(x.canonical(P.k), a, b);
                ++n; sa += a; sb += b;
            }
        }
    }
    n = wave_sum_u64(n); sa = wave_sum_u64(sa); sb = wave_sum_u64(sb);
    if ((threadIdx.x & 63) == 0 && n) { atomic_add_u64(&P.out[0], n); atomic_add_u64(&P.out[1], sa); atomic_add_u64(&P.out[2], sb); }
}

template <int W>
__global__ void k_verify_solid(VerifyParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t n = 0, sa = 0, sb = 0;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P.n_parts; p += stride) {
        const uint64_t so = P.seg_off[p];
        for (uint32_t e = 0, ne = P.seg_n[p]; e < ne; ++e) {
            if (P.solid_cnt[so + e] & TRAV_FLAG) continue;          // traveller copies are not part of the k-mer set
            Kmer<W> c;
#pragma unroll
            for (int i = 0; i < W; ++i) c.w[i] = P.solid_keys[(so + e) * W + i];
            c.w[W - 1] &= ~KEY_FLAGS;                               // (junction-ownership flags ride in the top bits)
            uint64_t a, b; verify_mix<W>(c, a, b);
            ++n; sa += a; sb += b;
        }
    }
    n = wave_sum_u64(n); sa = wave_sum_u64(sa); sb = wave_sum_u64(sb);
    if ((threadIdx.x & 63) == 0 && n) { atomic_add_u64(&P.out[3], n); atomic_add_u64(&P.out[4], sa); atomic_add_u64(&P.out[5], sb); }
}

__global__ void k_verify_maximal(VerifyParams P) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t merge = 0, cyc = 0;
    if (e < 2 * P.n_unitigs) {
        const uint64_t o = P.link_off[e];
        if (P.link_off[e + 1] - o == 1) {
            const uint64_t f = P.link_to[o];
            if (f != e && P.link_off[f + 1] - P.link_off[f] == 1) {   // (links are symmetric: f's only link is e)
                if ((f >> 1) != (e >> 1)) merge = 1;
                else if (e & 1) cyc = 1;
            }
        }
    }
    merge = wave_sum_u64(merge); cyc = wave_sum_u64(cyc);
    if ((threadIdx.x & 63) == 0) { if (merge) atomic_add_u64(&P.out[6], merge); if (cyc) atomic_add_u64(&P.out[7], cyc); }
}

// ---- edge conservation ----
struct VerifyEdgeParams {
    int k;
    const uint64_t* seg_off; const uint32_t* seg_n; const uint64_t* solid_keys; const uint32_t* solid_cnt; uint64_t n_parts;
    uint64_t* jt_keys; uint32_t* jt_cnt; uint32_t jt_mask;      // junction table: W words per key; count word = a | b << 10 | c << 20
    uint64_t* out;                                               // [0] D  [1] distinct junctions  [2] table overflow (must be 0)
};
constexpr uint32_t VE_A = 1u, VE_B = 1u << 10, VE_C = 1u << 20;

template <int W>
__global__ void k_verify_edge_insert(VerifyEdgeParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const KTable<W> T{ P.jt_keys, P.jt_mask };
    uint64_t lost = 0;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P.n_parts; p += stride) {
        const uint64_t so = P.seg_off[p];
        for (uint32_t e = 0, ne = P.seg_n[p]; e < ne; ++e) {
            if (P.solid_cnt[so + e] & TRAV_FLAG) continue;          // (a traveller is a copy: its home entry registers both ends)
            Kmer<W> c;
#pragma unroll
            for (int i = 0; i < W; ++i) c.w[i] = P.solid_keys[(so + e) * W + i];
            c.w[W - 1] &= ~KEY_FLAGS;
            const Kmer<W> r = c.rc(P.k);
            for (int side = 0; side < 2; ++side) {                  // leaving through the last k-mer / through the reverse complement of the first
                const Kmer<W> j = suffix_km1<W>(side ? c : r, P.k);
                const Kmer<W> jr = j.rc(P.k - 1);
                const bool pal = (jr == j);
                const bool other = !pal && jr < j;
                bool nw; const uint32_t s = ktable_insert<W, true>(T, other ? jr : j, nw, 1u << 20);
                if (s == 0xFFFFFFFFu) { ++lost; continue; }
                atomic_add_u32(&P.jt_cnt[s], pal ? VE_C : other ? VE_B : VE_A);
            }
        }
    }
    lost = wave_sum_u64(lost);
    if ((threadIdx.x & 63) == 0 && lost) atomic_add_u64(&P.out[2], lost);
}
__global__ void k_verify_edge_sum(VerifyEdgeParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t d = 0, nj = 0;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s <= (uint64_t)P.jt_mask; s += stride) {
        const uint32_t w = P.jt_cnt[s];
        if (!w) continue;
        const uint64_t a = w & 1023u, b = (w >> 10) & 1023u, c = w >> 20;
        d += 2 * a * b + c * c; ++nj;
    }
    d = wave_sum_u64(d); nj = wave_sum_u64(nj);
    if ((threadIdx.x & 63) == 0 && nj) { atomic_add_u64(&P.out[0], d); atomic_add_u64(&P.out[1], nj); }
}

}  // namespace cdbg
