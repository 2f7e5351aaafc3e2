// k_links.h -- edges between unitigs.
//
// New MI355X design for what gatb-core's link_tigs does after bglue (the `L:<+/->:<id>:<+/->`
// tokens of /root/reference/README.md:62-72 and the GFA1 `L` lines of
// /root/reference/scripts/convertToGFA.py:103-112; edge semantics:
// /root/reference/bidirected-graphs-in-bcalm2/bidirected-graphs-in-bcalm2.md:39-46,100-103).
//
// Only the first and last k-mer of a unitig can have edges to other unitigs, and an edge is a
// (k-1)-overlap, so this is one more hash-join on junction (k-1)-mers: every unitig END (2 per
// unitig) is keyed by the canonical (k-1)-mer it reaches when LEAVING the unitig through that end,
// plus a flag telling on which strand of the key it leaves.  Leaving through end e enters every
// end e' with the same key and the opposite flag (same flag when the key is its own reverse
// complement: a palindromic junction, where the edge back into e itself is the self-mirror edge of
// .md:30).  Link of unitig u:  from-sign '+' = leaving through its last k-mer, '-' = through the
// reverse complement of its first k-mer;  to-sign '+' = entering v at its first k-mer, '-' = at the
// reverse complement of its last k-mer.
#pragma once
#include "k_glue.h"

namespace cdbg {

constexpr int LINK_THREADS = 256;
// unitig ends that can share one junction key and strand flag: the four k-mers c + J, plus one more at even k, where a
// k-mer that is its own reverse complement is a unitig of its own whose two ends both leave through J (.md:30)
constexpr uint32_t LINK_PER_FLAG = 6;

template <int W>
struct LinkTable {
    KTable<W> t;
    uint32_t* cnt;        // [cap * 2]  ends seen per flag
    uint32_t* ends;       // [cap * 2 * LINK_PER_FLAG]  end ids per flag
};
struct LinkParams {
    uint64_t n_unitigs; int k;
    const uint64_t* unitig_off; const uint32_t* unitig_len; const uint8_t* bases;
    uint64_t* lk_keys; uint32_t* lk_cnt; uint32_t* lk_ends; uint32_t lk_mask;
    uint32_t* end_slot;   // [2U] table slot of each end (bit 31 = flag, bit 30 = palindromic key)
    uint32_t* deg;        // [2U] out-degree of each end
    const uint64_t* link_off; uint32_t* link_to;   // fill pass
    // a unitig set SHARDED over the ranks of a multi-GPU job (round 5): every rank describes its ends -- the canonical junction key
    // and a flag word -- all ranks gather all descriptions (rank order: the position of an end is its job-wide id, 2 x unitig id + side,
    // with unitig ids numbered rank after rank), every rank joins ALL ends in its own table and keeps the links of its own ends
    // [e0, e0 + 2 n_unitigs).  Ends are ~3 % of a graph's bytes: this replicates a 6 ms kernel, not the graph.
    uint64_t* end_keys; uint32_t* end_meta;          // k_link_describe: [2U * W], [2U]  (bit 0 = flag, bit 1 = palindromic key)
    const uint64_t* all_keys; const uint32_t* all_meta; uint64_t n_all_ends, e0;
};

// out-going oriented k-mer of end e of a unitig held as ASCII
template <int W>
CDBG_DEV Kmer<W> link_end_kmer(const uint8_t* s, uint32_t len, int k, uint32_t side) {
    Kmer<W> x = Kmer<W>::zero();
    const uint8_t* p = side ? s + (len - (uint32_t)k) : s;
    for (int i = 0; i < k; ++i) x.push_right(k, base_code(p[i]));
    return side ? x : x.rc(k);
}

template <int W>
__global__ void k_link_insert(LinkParams P) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * P.n_unitigs) return;
    const uint64_t u = e >> 1; const uint32_t side = (uint32_t)(e & 1);
    const Kmer<W> x = link_end_kmer<W>(P.bases + P.unitig_off[u], P.unitig_len[u], P.k, side);
    Kmer<W> j = suffix_km1<W>(x, P.k);
    const Kmer<W> r = j.rc(P.k - 1);
    const bool pal = (r == j);
    const uint32_t flag = (!pal && r < j) ? 1u : 0u;
    const Kmer<W> jc = flag ? r : j;
    const KTable<W> T{ P.lk_keys, P.lk_mask };
    bool nw; const uint32_t s = ktable_insert<W, true>(T, jc, nw);
    const uint32_t idx = atomic_add_u32(&P.lk_cnt[s * 2 + flag], 1u);
    if (idx < LINK_PER_FLAG) P.lk_ends[((uint64_t)s * 2 + flag) * LINK_PER_FLAG + idx] = (uint32_t)e;
    P.end_slot[e] = s | (flag << 31) | (pal ? (1u << 30) : 0u);
}
// (sharded set) the junction key and flags of every local end, as k_link_insert forms them
template <int W>
__global__ void k_link_describe(LinkParams P) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * P.n_unitigs) return;
    const uint64_t u = e >> 1; const uint32_t side = (uint32_t)(e & 1);
    const Kmer<W> x = link_end_kmer<W>(P.bases + P.unitig_off[u], P.unitig_len[u], P.k, side);
    Kmer<W> j = suffix_km1<W>(x, P.k);
    const Kmer<W> r = j.rc(P.k - 1);
    const bool pal = (r == j);
    const uint32_t flag = (!pal && r < j) ? 1u : 0u;
    const Kmer<W> jc = flag ? r : j;
    for (int i = 0; i < W; ++i) P.end_keys[e * W + i] = jc.w[i];
    P.end_meta[e] = flag | (pal ? 2u : 0u);
}
// (sharded set) every end of the job into this rank's table; end_slot is indexed by the job-wide end id
template <int W>
__global__ void k_link_insert_described(LinkParams P) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= P.n_all_ends) return;
    Kmer<W> jc; for (int i = 0; i < W; ++i) jc.w[i] = P.all_keys[g * W + i];
    const uint32_t m = P.all_meta[g], flag = m & 1u, pal = (m >> 1) & 1u;
    const KTable<W> T{ P.lk_keys, P.lk_mask };
    bool nw; const uint32_t s = ktable_insert<W, true>(T, jc, nw);
    const uint32_t idx = atomic_add_u32(&P.lk_cnt[s * 2 + flag], 1u);
    if (idx < LINK_PER_FLAG) P.lk_ends[((uint64_t)s * 2 + flag) * LINK_PER_FLAG + idx] = (uint32_t)g;
    P.end_slot[g] = s | (flag << 31) | (pal ? (1u << 30) : 0u);
}
__global__ void k_link_count(LinkParams P) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * P.n_unitigs) return;
    const uint32_t v = P.end_slot[P.e0 + e], s = v & 0x3FFFFFFFu, flag = v >> 31, pal = (v >> 30) & 1u;
    const uint32_t other = pal ? flag : flag ^ 1u;
    const uint32_t c = P.lk_cnt[s * 2 + other];
    P.deg[e] = c < LINK_PER_FLAG ? c : LINK_PER_FLAG;
}
__global__ void k_link_fill(LinkParams P) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * P.n_unitigs) return;
    const uint32_t v = P.end_slot[P.e0 + e], s = v & 0x3FFFFFFFu, flag = v >> 31, pal = (v >> 30) & 1u;
    const uint32_t other = pal ? flag : flag ^ 1u;
    const uint32_t c = P.deg[e];
    const uint64_t o = P.link_off[e];
    for (uint32_t i = 0; i < c; ++i) P.link_to[o + i] = P.lk_ends[((uint64_t)s * 2 + other) * LINK_PER_FLAG + i];
}


// ---- result digests (cdbg_digest): size-independent checks at sizes no oracle can follow ----
// Per unitig an orientation-independent 64-bit hash: polynomial hashes of the sequence and of its reverse complement,
// combined symmetrically, mixed with KC; the set digest is the SUM over unitigs (order independent).  tests/ hold the
// same formula in Python and pin it against the oracle's unitigs on small inputs.
struct DigestParams {
    uint64_t n_unitigs; int k;
    const uint64_t* unitig_off; const uint32_t* unitig_len; const uint64_t* unitig_kc; const uint8_t* bases;
    const uint64_t* seg_off; const uint32_t* seg_n; const uint32_t* solid_cnt; uint64_t n_parts;
    uint64_t* out;                 // [0] sum KC  [1] sum of solid (home) counts  [2] set digest  [3] sum (LN - k + 1)
};
CDBG_DEV uint32_t digest_code(uint8_t c) { return ((c >> 1) ^ (c >> 2)) & 3u; }   // A0 C1 G2 T3 (kmer.h base_code)
__global__ void k_digest_unitigs(DigestParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t kc_sum = 0, dig = 0, km_sum = 0;
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < P.n_unitigs; u += stride) {
        const uint8_t* b = P.bases + P.unitig_off[u];
        const uint32_t n = P.unitig_len[u];
        const uint64_t B = 0x100000001B3ULL;
        uint64_t hf = 0, hr = 0;
        for (uint32_t i = 0; i < n; ++i) {
            hf = hf * B + (uint64_t)(digest_code(b[i]) + 1u);
            hr = hr * B + (uint64_t)(3u - digest_code(b[n - 1 - i]) + 1u);
        }
        const uint64_t kc = P.unitig_kc[u];
        dig += mix64((hf + hr) ^ mix64(hf * hr + kc));
        kc_sum += kc; km_sum += (uint64_t)n - (uint64_t)P.k + 1u;
    }
    kc_sum = wave_sum_u64(kc_sum); dig = wave_sum_u64(dig); km_sum = wave_sum_u64(km_sum);
    if ((threadIdx.x & 63) == 0) { atomic_add_u64(&P.out[0], kc_sum); atomic_add_u64(&P.out[2], dig); atomic_add_u64(&P.out[3], km_sum); }
}
__global__ void k_digest_solid(DigestParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t s = 0;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P.n_parts; p += stride) {
        const uint64_t so = P.seg_off[p];
        for (uint32_t e = 0, n = P.seg_n[p]; e < n; ++e) { const uint32_t c = P.solid_cnt[so + e]; if (!(c & TRAV_FLAG)) s += c; }
    }
    s = wave_sum_u64(s);
    if ((threadIdx.x & 63) == 0 && s) atomic_add_u64(&P.out[1], s);
}

}  // namespace cdbg
