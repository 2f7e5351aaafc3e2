// pgz.h -- host side of row f2 (/root/reference/README.md:45-50: "FASTA/FASTQ, gzipped or not"): ONE gzip file inflated by many
// threads.  Pure host code (no HIP); zlib is used only for crc32 / crc32_combine.
//
// A deflate stream is sequential twice over: a block starts at an arbitrary BIT, and a match may copy from the 32 KB before it.  Both are
// worked around the way pugz / rapidgzip do it for text:
//   A  the file is cut into chunks of `chunk_bytes`; in each chunk a thread LOOKS for the start of a deflate block: a bit position where a
//      dynamic-Huffman header parses (complete code-length / literal / distance codes, an end-of-block code), the whole block decodes to text
//      bytes, and what follows is again a plausible block header (or a gzip trailer + the next member / the end of the file).
//   B  every chunk is decoded from its start to the start of the next one with the window UNKNOWN: the output is 16-bit symbols, a byte or
//      "position j of the 32 KB before this chunk" (a copy of a placeholder is a placeholder).  A chunk must END exactly where the next one
//      starts; if it runs past that position the next start was not a block start and the chunk carries on to the one after.
//   C  the chain of chunks is walked from chunk 0 (whose start is known): each resolves its last 32 KB against the window handed to it and
//      hands the result on -- 32 K look-ups per chunk, the only sequential work.
//   D  all chunks of the wave are resolved to bytes in parallel; CRC32 per piece between member ends (parallel), combined (crc32_combine) and
//      checked against every member's trailer (CRC32, ISIZE) before the caller sees the text.
// The file is processed in WAVES of `threads` chunks; the text of a wave is handed to the caller as one contiguous buffer (it answers with
// the number of trailing bytes it wants to see again in front of the next wave: an unfinished record).  The caller works on a wave while
// the next one is being decoded.
// Anything this code cannot do BEFORE the first wave was delivered (no block starts found: stored / fixed blocks only; a tiny file) is
// reported as "not handled" and the caller inflates with zlib; after that an inconsistency is an error (corrupt file).
#pragma once
#include <zlib.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace pgz {

constexpr uint64_t NONE = ~0ull;
constexpr size_t DAMAGED = ~(size_t)0 - 1;     // (internal: a member's CRC32 / ISIZE did not match)
constexpr size_t ABORT = ~(size_t)0;            // on_wave: "I cannot use this text" (only honoured for the first wave)
constexpr uint32_t WIN = 32768;
constexpr size_t MAX_CHUNK_SYMBOLS = (size_t)1 << 31;   // per chunk of compressed bytes: deflate reaches 1032 : 1 on constant input, and a wave holds `threads` chunks as 16-bit symbols

inline uint64_t peek_bits(const uint8_t* d, size_t n, uint64_t bit) {          // >= 56 valid bits (zeros beyond the end)
    const size_t b = (size_t)(bit >> 3);
    uint64_t v = 0;
    if (b + 8 <= n) memcpy(&v, d + b, 8);
    else if (b < n) memcpy(&v, d + b, n - b);
    return v >> (bit & 7);
}

// one-level table: index = the next `bits` bits of the stream, entry = symbol << 4 | code length (0: no such code)
struct Huff {
    std::vector<uint16_t> tab; int bits = 0;
    uint16_t fast[1 << 10];                       // the first 10 bits: the entry when the code is that short (most are), else 0 -> `tab`; stays in L1
    // zlib's rule (inftrees.c): over-subscribed -> invalid; incomplete -> valid only as ONE code of length 1
    bool build(const uint8_t* lens, int n) {
        int count[16] = {0};
        for (int i = 0; i < n; ++i) ++count[lens[i]];
        int maxb = 15; while (maxb > 0 && !count[maxb]) --maxb;
        bits = maxb;
        if (maxb == 0) { tab.assign(1, 0); memset(fast, 0, sizeof fast); return true; }             // no codes at all (a block without matches has no distance code)
        int left = 1;
        for (int l = 1; l <= 15; ++l) { left <<= 1; left -= count[l]; if (left < 0) return false; }
        if (left > 0 && maxb != 1) return false;
        uint32_t next[16]; uint32_t code = 0; count[0] = 0;
        for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code; }
        tab.assign((size_t)1 << maxb, 0);
        for (int s = 0; s < n; ++s) {
            const int l = lens[s]; if (!l) continue;
            uint32_t c = next[l]++, r = 0;
            for (int i = 0; i < l; ++i) { r = (r << 1) | (c & 1u); c >>= 1; }
            const uint16_t e = (uint16_t)((s << 4) | l);
            for (uint32_t j = r; j < (1u << maxb); j += 1u << l) tab[j] = e;
        }
        for (uint32_t j = 0; j < 1024; ++j) { const uint16_t e = tab[j & ((1u << maxb) - 1)]; fast[j] = (e & 15) <= 10 ? e : 0; }
        return true;
    }
};

static const uint16_t LEN_BASE[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const uint8_t LEN_EXTRA[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint16_t DIST_BASE[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const uint8_t DIST_EXTRA[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
static const uint8_t CL_ORDER[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

// position after the gzip member header that starts at byte `pos`; 0: not a gzip header
inline size_t gzip_header(const uint8_t* d, size_t n, size_t pos) {
    if (pos + 10 > n || d[pos] != 0x1f || d[pos + 1] != 0x8b || d[pos + 2] != 8 || (d[pos + 3] & 0xE0)) return 0;
    const uint8_t flg = d[pos + 3]; size_t p = pos + 10;
    if (flg & 4) { if (p + 2 > n) return 0; p += 2 + ((size_t)d[p] | ((size_t)d[p + 1] << 8)); }
    if (flg & 8) { while (p < n && d[p]) ++p; ++p; }
    if (flg & 16) { while (p < n && d[p]) ++p; ++p; }
    if (flg & 2) p += 2;
    return p < n ? p : 0;
}

// symbol buffers go round: a fresh 100 MB buffer per chunk costs its page faults every time
struct BufPool {
    std::mutex mu; std::vector<std::pair<uint16_t*, size_t>> v;
    std::pair<uint16_t*, size_t> get() { std::lock_guard<std::mutex> l(mu); if (v.empty()) return {nullptr, 0}; auto b = v.back(); v.pop_back(); return b; }
    void put(uint16_t* p, size_t cap) { if (!p) return; std::lock_guard<std::mutex> l(mu); v.emplace_back(p, cap); }
    ~BufPool() { for (auto& b : v) free(b.first); }
};

struct MemberEnd { size_t out_off; uint32_t crc, isize; };

// a resumable decoder of deflate blocks into 16-bit symbols
struct Inflater {
    const uint8_t* d; size_t n; uint64_t bit = 0;
    bool placeholders = false;                    // the 32 KB before the first output symbol are unknown (else: there is nothing before it)
    bool text_only = false;                       // (block search) reject literals that are not text
    size_t limit = ~(size_t)0;                    // (block search) reject a block longer than this
    uint16_t* o = nullptr; size_t pos = 0, cap = 0;
    bool last_final = false, eof = false;
    std::vector<MemberEnd> members;
    Huff lit, dist, fixed_lit, fixed_dist; bool have_fixed = false;
    const char* err = nullptr;
    Inflater(const uint8_t* data, size_t len) : d(data), n(len) {}
    ~Inflater() { free(o); }
    Inflater(const Inflater&) = delete; Inflater& operator=(const Inflater&) = delete;
    bool fail(const char* e) { err = e; return false; }
    bool room(size_t more) {
        if (pos + more <= cap) return true;
        if (pos + more > MAX_CHUNK_SYMBOLS) return fail("a chunk inflates to more than 2 GB (not sequence text; BCALM_GZ_SERIAL=1 reads it with zlib)");
        size_t nc = cap ? cap * 2 : (size_t)1 << 20; while (nc < pos + more) nc *= 2;
        void* p = realloc(o, nc * sizeof(uint16_t)); if (!p) return fail("out of memory");
        o = (uint16_t*)p; cap = nc; return true;
    }
    // the header of a dynamic block at `bit` (behind the three block bits): tables built, bit advanced
    bool dynamic_header() {
        uint64_t v = peek_bits(d, n, bit);
        const int hlit = (int)(v & 31) + 257, hdist = (int)((v >> 5) & 31) + 1, hclen = (int)((v >> 10) & 15) + 4;
        if (hlit > 286 || hdist > 30) return fail("bad block header");
        bit += 14;
        uint8_t cl[19] = {0};
        v = peek_bits(d, n, bit);                                     // 19 x 3 = 57 bits: two peeks
        for (int i = 0; i < hclen; ++i) { if (i == 16) v = peek_bits(d, n, bit + 48); cl[CL_ORDER[i]] = (uint8_t)((v >> (3 * (i < 16 ? i : i - 16))) & 7); }
        bit += 3 * (uint64_t)hclen;
        Huff clh; if (!clh.build(cl, 19) || clh.bits == 0) return fail("bad code-length code");
        uint8_t lens[320]; int i = 0; const int tot = hlit + hdist;
        while (i < tot) {
            v = peek_bits(d, n, bit);
            const uint16_t e = clh.tab[v & ((1u << clh.bits) - 1)]; const int l = e & 15, s = e >> 4;
            if (!l) return fail("bad code-length symbol");
            bit += l; v >>= l;
            if (s < 16) lens[i++] = (uint8_t)s;
            else {
                int rep; uint8_t val = 0;
                if (s == 16) { if (!i) return fail("repeat without a length"); val = lens[i - 1]; rep = 3 + (int)(v & 3); bit += 2; }
                else if (s == 17) { rep = 3 + (int)(v & 7); bit += 3; }
                else { rep = 11 + (int)(v & 127); bit += 7; }
                if (i + rep > tot) return fail("too many lengths");
                while (rep--) lens[i++] = val;
            }
        }
        if (!lens[256]) return fail("no end-of-block code");
        if (!lit.build(lens, hlit) || !dist.build(lens + hlit, hdist)) return fail("bad Huffman code");
        if (bit > (uint64_t)n * 8) return fail("truncated");
        return true;
    }
    void fixed_tables() {
        if (have_fixed) return;
        uint8_t l[288]; for (int i = 0; i < 288; ++i) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
        fixed_lit.build(l, 288);
        uint8_t dl[32]; for (int i = 0; i < 32; ++i) dl[i] = 5;
        fixed_dist.build(dl, 32);                                      // (symbols 30 / 31 have codes and no meaning: rejected where a distance is decoded)
        have_fixed = true;
    }
    // the symbols of one block (tables L / D) from `bit` on.  The input is read through a 64-bit buffer that is topped up once per symbol (56+ bits: a
    // length + its extra bits + a distance + its extra bits need 48); beyond the end of the input it is fed zero bytes, and `bit` says so afterwards.
    bool codes(const Huff& L, const Huff& D) {
        const uint32_t lmask = (1u << L.bits) - 1, dmask = D.bits ? (1u << D.bits) - 1 : 0;
        const uint16_t* lt = L.tab.data(); const uint16_t* dt = D.tab.data(); const uint16_t* lf = L.fast;
        const uint8_t* in = d + (size_t)(bit >> 3); const uint8_t* const in_end = d + n;
        uint64_t bb = 0; unsigned bc = 0, zeros = 0;
        const size_t start = pos;
#define PGZ_REFILL() do { \
            if ((size_t)(in_end - in) >= 8) { uint64_t w; memcpy(&w, in, 8); bb |= w << bc; in += (63 - bc) >> 3; bc |= 56; } \
            else { while (bc <= 56) { if (in < in_end) bb |= (uint64_t)*in++ << bc; else ++zeros; bc += 8; } if (zeros > 16) { store_bit(in, zeros, bc); return fail("truncated"); } } } while (0)
#define PGZ_DROP(k) do { bb >>= (k); bc -= (k); } while (0)
        if (in > in_end) return fail("truncated");
        PGZ_REFILL();
        PGZ_DROP((unsigned)(bit & 7));
        for (;;) {
            if (pos + 300 > cap && !room(1 << 16)) return false;
            PGZ_REFILL();
            uint32_t e = lf[bb & 1023]; if (!e) e = lt[bb & lmask];
            uint32_t l = e & 15, s = e >> 4;
            if (!l) { store_bit(in, zeros, bc); return fail("bad literal/length code"); }
            PGZ_DROP(l);
            if (s < 256) {
                if (text_only && !(s >= 32 ? s < 127 : (s == '\n' || s == '\r' || s == '\t'))) return fail("not text");
                o[pos++] = (uint16_t)s;
                // a second and a third literal from the same buffer (most symbols of text are literals; 3 x 15 bits fit)
                e = lf[bb & 1023]; l = e & 15; s = e >> 4;
                if (l && s < 256 && !text_only) {
                    PGZ_DROP(l); o[pos++] = (uint16_t)s;
                    e = lf[bb & 1023]; l = e & 15; s = e >> 4;
                    if (l && s < 256) { PGZ_DROP(l); o[pos++] = (uint16_t)s; }
                }
                continue;
            }
            if (s == 256) break;
            s -= 257; if (s >= 29) return fail("bad length symbol");
            const uint32_t len = LEN_BASE[s] + (uint32_t)(bb & ((1u << LEN_EXTRA[s]) - 1)); PGZ_DROP(LEN_EXTRA[s]);
            if (!D.bits) return fail("match without a distance code");
            e = dt[bb & dmask]; l = e & 15; s = e >> 4;
            if (!l || s >= 30) return fail("bad distance code");
            PGZ_DROP(l);
            const uint32_t dd = DIST_BASE[s] + (uint32_t)(bb & ((1u << DIST_EXTRA[s]) - 1)); PGZ_DROP(DIST_EXTRA[s]);
            if (dd > pos) {
                if (!placeholders || dd - pos > WIN) return fail("distance too far back");
                for (uint32_t i = 0; i < len; ++i) {
                    const int64_t src = (int64_t)pos - (int64_t)dd;
                    o[pos] = src < 0 ? (uint16_t)(256 + (int64_t)WIN + src) : o[src];
                    ++pos;
                }
            } else {
                const uint16_t* src = o + pos - dd; uint16_t* dst = o + pos;
                if (dd >= len) memcpy(dst, src, len * sizeof(uint16_t));
                else for (uint32_t i = 0; i < len; ++i) dst[i] = src[i];
                pos += len;
            }
            if (pos - start > limit) return fail("block too long");
        }
#undef PGZ_REFILL
#undef PGZ_DROP
        store_bit(in, zeros, bc);
        if (bit > (uint64_t)n * 8) return fail("truncated");
        return true;
    }
    void store_bit(const uint8_t* in, unsigned zeros, unsigned bc) { bit = ((uint64_t)(in - d) + zeros) * 8 - bc; }
    // one block at `bit`
    bool block() {
        if (bit + 3 > (uint64_t)n * 8) return fail("truncated");
        const uint64_t v = peek_bits(d, n, bit);
        last_final = v & 1; const int type = (int)((v >> 1) & 3);
        bit += 3;
        if (type == 0) {
            size_t p = (size_t)((bit + 7) >> 3);
            if (p + 4 > n) return fail("truncated");
            const uint32_t len = d[p] | ((uint32_t)d[p + 1] << 8), nlen = d[p + 2] | ((uint32_t)d[p + 3] << 8);
            if ((len ^ nlen) != 0xFFFFu) return fail("bad stored block");
            p += 4; if (p + len > n) return fail("truncated");
            if (!room(len + 300)) return false;
            for (uint32_t i = 0; i < len; ++i) o[pos + i] = d[p + i];
            pos += len; bit = (uint64_t)(p + len) * 8;
            return true;
        }
        if (type == 1) { fixed_tables(); return codes(fixed_lit, fixed_dist); }
        if (type == 2) return dynamic_header() && codes(lit, dist);
        return fail("bad block type");
    }
    // behind a final block: trailer, then the next member's header or the end of the file (trailing garbage is ignored, as gzip does)
    bool member_end() {
        size_t p = (size_t)((bit + 7) >> 3);
        if (p + 8 > n) return fail("truncated gzip trailer");
        MemberEnd m; m.out_off = pos;
        m.crc = d[p] | ((uint32_t)d[p + 1] << 8) | ((uint32_t)d[p + 2] << 16) | ((uint32_t)d[p + 3] << 24);
        m.isize = d[p + 4] | ((uint32_t)d[p + 5] << 8) | ((uint32_t)d[p + 6] << 16) | ((uint32_t)d[p + 7] << 24);
        members.push_back(m);
        p += 8;
        while (p < n && d[p] == 0) ++p;                                // (zero padding between members is legal for gzip -- tape blocks)
        const size_t h = p < n ? gzip_header(d, n, p) : 0;
        if (!h) { eof = true; bit = (uint64_t)n * 8; return true; }
        bit = (uint64_t)h * 8;
        return true;
    }
    // blocks from `bit` until a block (or member) boundary at or behind `target`; true: bit >= target or eof
    bool run_until(uint64_t target) {
        while (!eof && bit < target) {
            if (!block()) return false;
            if (last_final && !member_end()) return false;
        }
        return true;
    }
};

// is `bit` the start of a dynamic block of text, followed by something that looks like a block again?
inline bool probe(const uint8_t* d, size_t n, uint64_t bit, Inflater& t) {
    t.bit = bit; t.pos = 0; t.err = nullptr; t.eof = false; t.members.clear();
    if (!t.block()) return false;
    if (t.pos < 64) return false;                                      // (an empty or tiny block proves nothing)
    if (t.last_final) {                                                 // a trailer and the next member, or the end of the file
        const size_t p = (size_t)((t.bit + 7) >> 3);
        if (p + 8 > n) return false;
        if (p + 8 == n) return true;
        return gzip_header(d, n, p + 8) != 0;
    }
    const uint64_t v = peek_bits(d, n, t.bit); const int type = (int)((v >> 1) & 3);
    if (type == 3) return false;
    if (type == 0) {
        const size_t p = (size_t)((t.bit + 3 + 7) >> 3); if (p + 4 > n) return false;
        return ((d[p] | ((uint32_t)d[p + 1] << 8)) ^ (d[p + 2] | ((uint32_t)d[p + 3] << 8))) == 0xFFFFu;
    }
    if (type == 2) { const uint64_t keep = t.bit; t.bit += 3; const bool ok = t.dynamic_header(); t.bit = keep; return ok; }
    return true;
}
inline uint64_t find_block(const uint8_t* d, size_t n, uint64_t from_bit, uint64_t to_bit) {
    Inflater t(d, n); t.placeholders = true; t.text_only = true; t.limit = (size_t)8 << 20;
    for (uint64_t b = from_bit; b < to_bit; ++b) {
        const uint64_t v = peek_bits(d, n, b);
        if (((v >> 1) & 3) != 2) continue;                             // BTYPE = dynamic (either BFINAL)
        if (((v >> 3) & 31) > 29 || ((v >> 8) & 31) > 29) continue;    // HLIT <= 286, HDIST <= 30
        if (probe(d, n, b, t)) return b;
    }
    return NONE;
}

template <class Fn>
inline void parallel_for(size_t n, int threads, Fn&& fn) {
    std::atomic<size_t> next{0};
    auto w = [&]() { for (;;) { const size_t i = next.fetch_add(1); if (i >= n) break; fn(i); } };
    std::vector<std::thread> th;
    const int nt = (int)std::min<size_t>((size_t)std::max(1, threads), n);
    for (int i = 1; i < nt; ++i) th.emplace_back(w);
    w();
    for (auto& t : th) t.join();
}

struct Stats { size_t chunks = 0, starts_found = 0, chunks_on_chain = 0, waves = 0, members = 0; uint64_t out_bytes = 0; double s_find = 0, s_decode = 0, s_resolve = 0, s_caller = 0, s_wait = 0; };
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// on_wave(char* text, size_t n, bool last) -> bytes at the end of `text` to put in front of the next wave (ABORT: see above).
// returns 0: the whole file was delivered; 1: not handled, nothing was delivered (inflate it with zlib).  Throws std::runtime_error on a
// corrupt file once text has been delivered.
inline int inflate_parallel(const uint8_t* d, size_t n, int threads, size_t chunk_bytes,
                            const std::function<size_t(char*, size_t, bool)>& on_wave, Stats* stats = nullptr) {
    const size_t hdr = gzip_header(d, n, 0);
    if (!hdr || threads < 2 || chunk_bytes < 1024) return 1;
    const size_t nchunks = (n + chunk_bytes - 1) / chunk_bytes;
    if (nchunks < 2) return 1;
    std::vector<uint64_t> starts(nchunks + 1, NONE);
    double t0 = now_s();
    starts[0] = (uint64_t)hdr * 8;
    parallel_for(nchunks - 1, threads, [&](size_t j) {
        const size_t i = j + 1;
        starts[i] = find_block(d, n, (uint64_t)i * chunk_bytes * 8, std::min<uint64_t>((uint64_t)(i + 1) * chunk_bytes, n) * 8);
    });
    size_t found = 0; for (size_t i = 1; i < nchunks; ++i) found += starts[i] != NONE;
    if (found * 2 < nchunks - 1) return 1;
    Stats st; st.chunks = nchunks; st.starts_found = found; st.s_find = now_s() - t0;

    struct Chunk { std::unique_ptr<Inflater> inf; size_t reached = 0; bool ok = false; std::string err; size_t out_at = 0; };
    struct Seg { size_t off, len; bool ends_member; uint32_t crc, isize, got = 0; };
    std::vector<uint8_t> window(WIN, 0);
    BufPool pool;
    struct Text { char* p = nullptr; size_t cap = 0; ~Text() { free(p); } } txs[2];  // the text of a wave (not zero-filled, kept over the waves); two: see `pending`
    int cur_tx = 0;
    std::vector<char> carry;
    uint32_t crc_run = (uint32_t)crc32(0L, Z_NULL, 0); uint64_t len_run = 0;
    bool delivered = false;
    auto bad = [&](const std::string& what) -> int { if (!delivered) return 1; throw std::runtime_error("corrupt gzip input: " + what); };
    // The caller works on wave w (parsing it) WHILE wave w + 1 is decoded; its answer -- the carry -- is needed when the text of wave w + 1 is laid out.
    struct Pending { std::future<size_t> f; char* text = nullptr; size_t total = 0; bool on = false; } pending;
    bool aborted = false;
    auto settle = [&]() {                                              // wait for the caller; false: it refused the first wave
        if (!pending.on) return true;
        const double tw = now_s();
        const size_t keep = pending.f.get(); pending.on = false;
        st.s_wait += now_s() - tw;
        if (keep == DAMAGED) { if (!delivered) { aborted = true; return false; } throw std::runtime_error("corrupt gzip input: CRC32 / length of a member do not match its trailer"); }
        if (keep == ABORT) { if (!delivered) { aborted = true; return false; } throw std::runtime_error("input text changed its format"); }
        delivered = true;
        if (keep > pending.total) throw std::runtime_error("pgz: bad carry");
        carry.assign(pending.text + (pending.total - keep), pending.text + pending.total);
        return true;
    };
    struct Settle { Pending& p; ~Settle() { if (p.on) { try { p.f.get(); } catch (...) {} } } } settle_on_exit{pending};   // (an exception on the way: the caller's threads are joined first)
    size_t cursor = 0;
    while (cursor < nchunks) {
        std::vector<size_t> wave{cursor};
        for (size_t j = cursor + 1; j < nchunks && (int)wave.size() < threads; ++j) if (starts[j] != NONE) wave.push_back(j);
        std::vector<Chunk> ch(wave.size());
        t0 = now_s();
        parallel_for(wave.size(), threads, [&](size_t w) {
            const size_t i = wave[w]; Chunk& c = ch[w];
            c.inf.reset(new Inflater(d, n)); Inflater& f = *c.inf;
            { auto b = pool.get(); f.o = b.first; f.cap = b.second; }
            f.placeholders = i != 0; f.bit = starts[i];
            size_t k = i + 1; int passed = 0;
            for (;;) {
                while (k < nchunks && starts[k] == NONE) ++k;
                const uint64_t target = k < nchunks ? starts[k] : NONE;
                if (!f.run_until(target)) { c.err = f.err ? f.err : "?"; return; }
                if (f.eof) { c.reached = nchunks; c.ok = true; return; }
                if (f.bit == target) { c.reached = k; c.ok = true; return; }
                ++k; if (++passed > 8) { c.err = "block boundaries do not meet"; return; }      // (ran past starts[k]: that was not a block start)
            }
        });
        st.s_decode += now_s() - t0;
        if (!settle()) return 1;
        t0 = now_s();
        // the chain from this wave's first chunk
        std::vector<size_t> chain; size_t c = cursor;
        for (;;) {
            size_t w = 0; while (w < wave.size() && wave[w] != c) ++w;
            if (w == wave.size()) break;
            if (!ch[w].ok) return bad(ch[w].err);
            chain.push_back(w); c = ch[w].reached;
            if (c >= nchunks) break;
        }
        const bool last = c >= nchunks;
        // windows along the chain (sequential), output offsets
        size_t total = carry.size();
        std::vector<std::vector<uint8_t>> win(chain.size());
        for (size_t q = 0; q < chain.size(); ++q) {
            Chunk& k = ch[chain[q]]; Inflater& f = *k.inf;
            win[q] = window; k.out_at = total; total += f.pos;
            std::vector<uint8_t> nw(WIN);
            for (uint32_t j = 0; j < WIN; ++j) {                       // text position p of this chunk's output; p < 0: the old window, whose last byte is position -1
                const int64_t p = (int64_t)f.pos - (int64_t)WIN + (int64_t)j;
                if (p < 0) nw[j] = window[(size_t)((int64_t)WIN + p)];
                else { const uint16_t s = f.o[p]; nw[j] = s < 256 ? (uint8_t)s : window[s - 256]; }
            }
            window.swap(nw);
        }
        Text& tx = txs[cur_tx]; cur_tx ^= 1;
        if (total + 1 > tx.cap) { free(tx.p); tx.cap = (total + 1) * 5 / 4; tx.p = (char*)malloc(tx.cap); if (!tx.p) throw std::runtime_error("pgz: out of memory"); }
        char* const text = tx.p;
        if (!carry.empty()) memcpy(text, carry.data(), carry.size());
        parallel_for(chain.size(), threads, [&](size_t q) {
            Chunk& k = ch[chain[q]]; Inflater& f = *k.inf; const uint8_t* wv = win[q].data();
            uint8_t* out = (uint8_t*)text + k.out_at;
            for (size_t j = 0; j < f.pos; ++j) { const uint16_t s = f.o[j]; out[j] = s < 256 ? (uint8_t)s : wv[s - 256]; }
            pool.put(f.o, f.cap); f.o = nullptr; f.cap = 0;
        });
        // the pieces of this wave's text between member ends, in order: their CRC32s are taken and checked beside the next wave's decoding, before the caller sees the text
        std::vector<Seg> segs;
        for (size_t q = 0; q < chain.size(); ++q) {
            Chunk& k = ch[chain[q]]; Inflater& f = *k.inf; size_t a = 0;
            for (size_t m = 0; m <= f.members.size(); ++m) {
                const size_t b = m < f.members.size() ? f.members[m].out_off : f.pos;
                Seg g; g.off = k.out_at + a; g.len = b - a; g.ends_member = m < f.members.size(); g.crc = g.ends_member ? f.members[m].crc : 0; g.isize = g.ends_member ? f.members[m].isize : 0;
                segs.push_back(g); a = b;
            }
            st.out_bytes += f.pos;
        }
        st.chunks_on_chain += chain.size(); ++st.waves;
        st.s_resolve += now_s() - t0; t0 = now_s();
        for (auto& k : ch) if (k.inf) { pool.put(k.inf->o, k.inf->cap); k.inf->o = nullptr; }        // (chunks that were not on the chain)
        ch.clear();
        pending.text = text; pending.total = total; pending.on = true;
        pending.f = std::async(std::launch::async, [&on_wave, &st, &crc_run, &len_run, threads, text, total, last, segs]() mutable -> size_t {
            parallel_for(segs.size(), threads, [&](size_t i) { segs[i].got = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef*)text + segs[i].off, (uInt)segs[i].len); });
            for (const Seg& g : segs) {
                crc_run = (uint32_t)crc32_combine(crc_run, g.got, (z_off_t)g.len); len_run += g.len;
                if (g.ends_member) {
                    if (crc_run != g.crc || (uint32_t)len_run != g.isize) return DAMAGED;
                    crc_run = (uint32_t)crc32(0L, Z_NULL, 0); len_run = 0; ++st.members;
                }
            }
            if (last && len_run) return DAMAGED;
            const double tc = now_s(); const size_t k = on_wave(text, total, last); st.s_caller += now_s() - tc; return k;
        });
        cursor = c;
    }
    if (!settle()) return 1;
    if (stats) *stats = st;
    return 0;
}

}  // namespace pgz
