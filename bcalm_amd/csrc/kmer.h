// kmer.h -- multi-word 2-bit k-mers for host and device code.
//
// Plays the role of gatb-core's Integer / LargeInt<N> selected by Integer::apply
// (/root/reference/src/bcalm_1.cpp:95, KSIZE_LIST in /root/reference/README.md:91-99):
// W 64-bit words hold a k-mer of up to 32*W-1 bases (the reference's span rule); W = 1, 2, 3, 4 cover
// k <= 31, 63, 95, 127 (BASELINE configs: k = 31, 55, 127).  New code, MI355X-first: plain structs of
// uint64_t that live in VGPRs, no virtual dispatch, everything __forceinline__.
//
// Encoding: A=0 C=1 G=2 T=3 (numeric order == lexicographic order, the canonical
// convention of /root/reference/scripts/unitigEvaluator.cpp:64-66,70-82), first
// base in the MOST significant position, w[0] = least significant word.
#pragma once
#include <stdint.h>

#include "devrt.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define CDBG_PIN(x) CDBG_PIN64(x)
#else
#define CDBG_PIN(x) do { } while (0)
#endif

namespace cdbg {

// ASCII -> 2-bit code for ACGTacgt; validity must be checked separately.
CDBG_HD uint32_t base_code(uint32_t c) { return ((c >> 1) ^ (c >> 2)) & 3u; }
CDBG_HD bool base_valid(uint32_t c) {
    c &= 0xDFu;                                   // fold case
    return c == 'A' || c == 'C' || c == 'G' || c == 'T';
}

// reverse the order of the 32 2-bit groups of x (no complement)
CDBG_HD uint64_t rev2(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    // gfx950: two v_bfrev_b32 reverse all 64 bits, one swap of neighbouring bits restores each 2-bit code
    x = __builtin_bitreverse64(x);
    return ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
#endif
    x = __builtin_bswap64(x);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    return x;
}
CDBG_HD uint32_t rev2_32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    x = __builtin_bitreverse32(x);
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
#endif
    x = __builtin_bswap32(x);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    return x;
}

// 32-bit bijective mixer (lowbias32): orders m-mers for minimizer selection.
CDBG_HD uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
CDBG_HD uint64_t mix64(uint64_t x) {          // splitmix64 finaliser (with increment)
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

template <int W>
struct Kmer {
    uint64_t w[W];

    CDBG_HD static Kmer zero() { Kmer r; for (int i = 0; i < W; ++i) r.w[i] = 0; return r; }
    CDBG_HD static Kmer ones() { Kmer r; for (int i = 0; i < W; ++i) r.w[i] = ~0ULL; return r; }

    CDBG_HD bool operator==(const Kmer& o) const {
        bool e = true;
        for (int i = 0; i < W; ++i) e &= (w[i] == o.w[i]);
        return e;
    }
    CDBG_HD bool operator!=(const Kmer& o) const { return !(*this == o); }
    CDBG_HD bool operator<(const Kmer& o) const {
        for (int i = W - 1; i > 0; --i) {
            if (w[i] != o.w[i]) return w[i] < o.w[i];
        }
        return w[0] < o.w[0];
    }

    // keep the low 2*k bits
    CDBG_HD void mask(int k) {
        const int bits = 2 * k;
        for (int i = 0; i < W; ++i) {
            const int lo = 64 * i;
            if (bits >= lo + 64) continue;
            if (bits <= lo) w[i] = 0;
            else w[i] &= (~0ULL) >> (64 - (bits - lo));
        }
    }
    // w[idx], or 0 when idx is outside [0, W): select chain, because a runtime index into w[] sends the
    // array to scratch memory (even for W == 1)
    CDBG_HD uint64_t word_z(int idx) const {
        uint64_t r = 0;
#pragma unroll
        for (int j = 0; j < W; ++j) { uint64_t e = w[j]; if (W > 1) CDBG_PIN(e); r = (idx == j) ? e : r; }
        return r;
    }
    // logical shifts of the whole W-word integer by s bits, 0 <= s < 64*W
    CDBG_HD Kmer shr(int s) const {
        Kmer r;
        const int ws = s >> 6, bs = s & 63;
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const uint64_t lo = word_z(i + ws), hi = word_z(i + ws + 1);
            r.w[i] = bs ? ((lo >> bs) | (hi << (64 - bs))) : lo;
        }
        return r;
    }
    CDBG_HD Kmer shl(int s) const {
        Kmer r;
        const int ws = s >> 6, bs = s & 63;
#pragma unroll
        for (int i = W - 1; i >= 0; --i) {
            const uint64_t hi = word_z(i - ws), lo = word_z(i - ws - 1);
            r.w[i] = bs ? ((hi << bs) | (lo >> (64 - bs))) : hi;
        }
        return r;
    }
    // append base c on the right (drop the leftmost base of a k-mer)
    CDBG_HD void push_right(int k, uint32_t c) {
        for (int i = W - 1; i > 0; --i) w[i] = (w[i] << 2) | (w[i - 1] >> 62);
        w[0] = (w[0] << 2) | (uint64_t)c;
        mask(k);
    }
    // word `idx` without dynamic indexing of the register array (a runtime index sends w[] to scratch)
    CDBG_HD uint64_t word(int idx) const {
        uint64_t r = w[0];
        if (W > 1) CDBG_PIN(r);
#pragma unroll
        for (int j = 1; j < W; ++j) { uint64_t e = w[j]; CDBG_PIN(e); r = (idx == j) ? e : r; }
        return r;
    }
    CDBG_HD void or_word(int idx, uint64_t v) {
#pragma unroll
        for (int j = 0; j < W; ++j) w[j] |= (idx == j) ? v : 0ULL;
    }
    // prepend base c on the left (drop the rightmost base)
    CDBG_HD void push_left(int k, uint32_t c) {
        for (int i = 0; i < W - 1; ++i) w[i] = (w[i] >> 2) | (w[i + 1] << 62);
        w[W - 1] >>= 2;
        const int pos = 2 * (k - 1);
        or_word(pos >> 6, (uint64_t)c << (pos & 63));
    }
    // i-th base counted from the left end of a k-mer
    CDBG_HD uint32_t base(int k, int i) const {
        const int pos = 2 * (k - 1 - i);
        return (uint32_t)(word(pos >> 6) >> (pos & 63)) & 3u;
    }
    // reverse complement of a k-mer
    CDBG_HD Kmer rc(int k) const {
        Kmer r;
        if (W == 1) { r.w[0] = (~rev2(w[0])) >> (64 - 2 * k); return r; }   // one word: no cross-word funnel (k <= 31)
        Kmer t;
        for (int i = 0; i < W; ++i) t.w[i] = ~rev2(w[W - 1 - i]);
        // shift right by s = 64 W - 2 k bits.  Under the span rule (32 (W - 1) <= k < 32 W; the (k-1)-mers of junctions one
        // base shorter) s is 2 .. 66: at most one whole word, decided by a wave-uniform branch on k -- no select chain over
        // the words (the general shr() costs 2 W of them per word: the rc of every member k-mer paid ~60 v_cndmask at W = 4)
        const int s = 64 * W - 2 * k;
        if (s < 64) {
#pragma unroll
            for (int i = 0; i < W; ++i) r.w[i] = (t.w[i] >> s) | (i + 1 < W ? t.w[i + 1 < W ? i + 1 : 0] << (64 - s) : 0ULL);
        } else if (s < 128) {
            const int b = s - 64;
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const uint64_t lo = i + 1 < W ? t.w[i + 1 < W ? i + 1 : 0] : 0ULL, hi = i + 2 < W ? t.w[i + 2 < W ? i + 2 : 0] : 0ULL;
                r.w[i] = b ? ((lo >> b) | (hi << (64 - b))) : lo;
            }
        } else r = t.shr(s);
        return r;                               // high bits are already zero after the shift
    }
    CDBG_HD Kmer canonical(int k) const {
        Kmer r = rc(k);
        return (r < *this) ? r : *this;
    }
    // hash of the LDS tables (k_count's fast path): one 32-bit multiply-add per 32-bit half and one finishing multiply;
    // the GOOD bits are the high ones (slot = hash_lds() >> (32 - log2 slots)).  The 64-bit product of hash() below costs
    // four 32-bit multiplies (bench_micro/micro_r02: 18 cycles per wave against 4.5 for one v_mul_lo_u32).
    CDBG_HD uint32_t hash_lds() const {
        // (measured and discarded in round 3: a fold of the dwords by rotations + two 24-bit multiplies -- full-rate operations
        //  only -- probed as well as this on random k-mers but sent 50 % more partitions of real read sets to the multi-pass
        //  kernel: count 169 -> 175 ms at k = 55, 352 -> 362 ms at k = 127)
        uint32_t t = (uint32_t)(w[0] >> 32);
        t = (uint32_t)w[0] * 0x9E3779B1u + t;
        for (int i = 1; i < W; ++i) { t = t * 0x85EBCA77u + (uint32_t)w[i]; t = t * 0xC2B2AE3Du + (uint32_t)(w[i] >> 32); }
        return t * 0x27D4EB2Fu;
    }
    // multiply-shift hash: the high half of a 64-bit product depends on every input bit
    CDBG_HD uint32_t hash() const {
        uint64_t a = w[0];
        for (int i = 1; i < W; ++i) a = (a ^ (a >> 29)) * 0xBF58476D1CE4E5B9ULL + w[i];
        a ^= a >> 32;
        return (uint32_t)((a * 0x9E3779B97F4A7C15ULL) >> 32);
    }
};

// (k-1)-mer on the right / left side of a k-mer (as a number with the same layout)
template <int W>
CDBG_HD Kmer<W> suffix_km1(const Kmer<W>& x, int k) { Kmer<W> r = x; r.mask(k - 1); return r; }
template <int W>
CDBG_HD Kmer<W> prefix_km1(const Kmer<W>& x, int /*k*/) { return x.shr(2); }

// m-mer (m <= 16) starting at base i of a sequence of `len` bases held in a Kmer-like integer
template <int W>
CDBG_HD uint32_t mmer_at(const Kmer<W>& x, int len, int i, int m) {
    const int pos = 2 * (len - m - i);
    uint64_t v = x.word(pos >> 6) >> (pos & 63);
    if ((pos & 63) + 2 * m > 64 && (pos >> 6) + 1 < W) v |= x.word((pos >> 6) + 1) << (64 - (pos & 63));
    return (uint32_t)(v & ((m == 16) ? 0xFFFFFFFFULL : ((1ULL << (2 * m)) - 1)));
}
// ordering key of an m-mer: bijective hash of its canonical form
CDBG_HD uint32_t mmer_key(uint32_t v, int m) {
    uint32_t r = (~rev2_32(v)) >> (32 - 2 * m);
    return mix32(r < v ? r : v);
}
// minimizer key of the (k-1)-mer `j` (given as an integer of k-1 bases)
template <int W>
CDBG_HD uint32_t junction_min(const Kmer<W>& j, int k, int m) {
    uint32_t g = 0xFFFFFFFFu;
    for (int i = 0; i + m <= k - 1; ++i) {
        uint32_t key = mmer_key(mmer_at<W>(j, k - 1, i, m), m);
        g = key < g ? key : g;
    }
    return g;
}
// minimizer keys of BOTH junctions of k-mer x (left = prefix (k-1)-mer, right = suffix (k-1)-mer) in one
// rolling pass over its k-m+1 m-mers (canonical m-mers: the strand of x does not matter)
template <int W>
CDBG_HD void kmer_junction_mins(const Kmer<W>& x, int k, int m, uint32_t& g_left, uint32_t& g_right, const uint32_t seed = 0u) {
    // (seed != 0: the same minima under ANOTHER order of the m-mers -- the sub-minimizers of k_split.h)
    // every m-mer and its reverse complement are bit fields of x and of rc(x): the reverse complement of the
    // m-mer at base j is the m-mer of rc(x) at base k-m-j (no per-base rolling, one rc() for the whole k-mer)
    uint32_t gl = 0xFFFFFFFFu, gr = 0xFFFFFFFFu;
    const int last = k - m;                                // m-mer starts 0 .. k-m; left junction owns 0 .. k-m-1, right 1 .. k-m
    if (W > 1) {
        // multi-word k-mers: two bit-field extractions per m-mer cost a select chain over the words each (a runtime word
        // index); instead stream the bases from the top word down and roll the m-mer and its reverse complement
        // (k steps of ~22 instructions instead of k-m+1 of ~44: half of k_compact_wave<4>'s time went here at k = 127)
        const uint32_t mmask = m == 16 ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);
        const int rsh = 2 * (m - 1);
        uint32_t fw = 0, rc = 0; int i = 0;
#pragma unroll
        for (int wi = W - 1; wi >= 0; --wi) {
            int nb = (2 * k - 64 * wi) / 2; nb = nb > 32 ? 32 : nb;   // bases of the k-mer held in word wi
            if (nb <= 0) continue;
            uint64_t v = x.w[wi] << (64 - 2 * nb);
            for (int t = 0; t < nb; ++t) {
                const uint32_t b = (uint32_t)(v >> 62); v <<= 2;
                fw = ((fw << 2) | b) & mmask;
                rc = (rc >> 2) | ((3u - b) << rsh);
                ++i;
                if (i >= m) {
                    const int j = i - m;
                    const uint32_t key = mix32((rc < fw ? rc : fw) ^ seed);
                    if (j < last) gl = key < gl ? key : gl;
                    if (j >= 1) gr = key < gr ? key : gr;
                }
            }
        }
        g_left = gl; g_right = gr;
        return;
    }
    const Kmer<W> r = x.rc(k);
    for (int j = 0; j <= last; ++j) {
        const uint32_t fw = mmer_at<W>(x, k, j, m), rc = mmer_at<W>(r, k, last - j, m);
        const uint32_t key = mix32((rc < fw ? rc : fw) ^ seed);
        if (j < last) gl = key < gl ? key : gl;
        if (j >= 1) gr = key < gr ? key : gr;
    }
    g_left = gl; g_right = gr;
}
// partition of a minimizer key; log_np == 0 -> single partition
CDBG_HD uint32_t part_of(uint32_t g, int log_np) {
    return log_np ? (uint32_t)((g * 0x9E3779B1u) >> (32 - log_np)) : 0u;
}
// sub-partition of a minimizer key inside its partition: the four hash bits below the partition's (log_np <= 26).  It rides in bits
// 12-15 of every record's meta word; the multi-pass count kernel (k_count.h) takes the records of one sub-partition per pass.
CDBG_HD uint32_t sub_of(uint32_t g, int log_np) {
    return (uint32_t)((g * 0x9E3779B1u) >> (28 - log_np)) & 15u;
}

}  // namespace cdbg
