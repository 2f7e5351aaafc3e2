// k_scan_fast.h -- stage 1a, instruction-lean variant of k_scan (same records, same rules).
//
// Profiling k_scan on MI355X (profiles/) showed the scan was not atomic- or HBM-bound but
// issue-bound: ~600 instructions per junction.  This variant keeps the tile in LDS but gives
// every lane 16 CONSECUTIVE positions so that per-position work becomes incremental:
//   A  encode 16 bases per lane with SIMD-in-register tricks (4 bases per multiply)
//   B  rolling canonical m-mer + validity run length -> ordering key (one hash per position)
//   C  window minimum over k-m keys from a register window (no LDS ping-pong, no barriers)
//   D  validity / run-break bits from a 64-bit validity window, written as 16-bit words
//   E  run starts are compacted into an LDS list (popcount prefix over the workgroup), then
//      every lane handles ONE run: end search, boundary (home / traveller) rules, records.
// Used when k <= 127: k - m <= SCANF_WNMAX from a register window; longer windows (k = 127, m = 16: 111 keys) through the TWO-LEVEL
// window minimum (WNT = -1: a lane's 16 window minima = suffix minima of its own 16 keys, the minima of the whole 16-key blocks in
// between -- one LDS word per block, written with the keys -- and prefix minima of the one or two blocks the windows end in) and a
// 192-bit validity window.  k_scan remains the generic path (k > 127).  The generic scan spent 7.4 ps per base at k = 127, this one 3 - 4.  Tile-local index q: byte (tile_start - 16 + q); junction jq <-> q = 15+jq.
#pragma once
#include "k_scan.h"

namespace cdbg {

constexpr int SCANF_TILE = 4064;                       // junctions per workgroup (254 x 16)
constexpr int SCANF_GRID = 256 * 5;                       // fallback grid of the persistent launch (cdbg_impl.cpp asks the runtime)
constexpr int SCANF_WNMAX = 48;                        // largest k-m handled by the register window
constexpr int SCANF_NQ = 4400;                         // tile-local positions held in LDS
constexpr int SCANF_PKW = SCANF_NQ / 16 + 5;           // packed words (16 bases each) incl. over-read
CDBG_DEV int scanf_pad(int q) { return q + (q >> 4); } // 17-word stride: conflict-free 16-per-lane access

// 4 ASCII bases (little-endian in x) -> 8 bits, first base on top; and 4 validity bits (bit j = byte j)
CDBG_DEV uint32_t scanf_enc4(uint32_t x, uint32_t& vbits) {
    const uint32_t t = ((x >> 1) ^ (x >> 2)) & 0x03030303u;             // 2-bit code per byte
    // expected upper-case letter of each code: "ACGT"[code]
    const uint32_t e = ((t & 0x01010101u) * 2u) + ((t & 0x02020202u) * 3u) + 0x41414141u
                     + (((t >> 1) & t & 0x01010101u) * 11u);              // 0:A(41) 1:C(43) 2:G(47) 3:T(54)
    const uint32_t d = e ^ (x & 0xDFDFDFDFu);                              // zero byte <=> valid base
    const uint32_t z = ~((((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d)) & 0x80808080u;
    vbits = (((z >> 7) * 0x01020408u) >> 24) & 0xFu;                       // bit j <- byte j
    return (t * 0x40100401u) >> 24;                                        // b0<<6 | b1<<4 | b2<<2 | b3
}

// (the generic-window variant took 222 VGPRs when left alone: two workgroups per CU where the LDS has room for five;
//  promised 3 waves per SIMD it takes 168 and no scratch: scan 118 -> 93 ms at the config-4 share; 4 spills: 113 ms)
#ifndef CDBG_SCAN_WAVES0
#define CDBG_SCAN_WAVES0 3
#endif
// (the 15- / 16-key windows: five workgroups per CU fit the LDS; with the next tile's 16-byte loads held across a tile the kernel takes 102 VGPRs when
//  left alone -- four waves per SIMD -- and 96 when promised five)
#ifndef CDBG_SCAN_WAVES15
#define CDBG_SCAN_WAVES15 5
#endif
template <int W, int MODE, int WNT>
__global__ void __launch_bounds__(SCAN_THREADS, WNT == 0 ? CDBG_SCAN_WAVES0 : (WNT == 15 || WNT == 16) ? CDBG_SCAN_WAVES15 : WNT < 0 ? 2 : 4) k_scan_fast(ScanParams P) {
    constexpr int RW = RecFmt<W>::RW;
    constexpr int CAPB = RecFmt<W>::CAPB;
    // packed bases + validity bits of TWO tiles: the one being scanned and the next one, whose bytes are loaded while this one is scanned (below)
    CDBG_SHARED uint32_t pk2[2][SCANF_PKW];
    CDBG_SHARED uint32_t vm2[2][SCANF_PKW / 2 + 4];
    CDBG_SHARED uint32_t kg[SCANF_NQ + SCANF_NQ / 16 + 32];   // keys, then g (padded layout)
    CDBG_SHARED uint32_t blk[WNT < 0 ? SCANF_NQ / 16 + 8 : 1];   // two-level window: minimum of every 16-key block
    CDBG_SHARED uint32_t brk[SCANF_NQ / 32 + 4];              // bit q: junction q does not continue a run
    CDBG_SHARED uint32_t stt[SCANF_NQ / 32 + 4];              // bit q: junction q starts a run
    CDBG_SHARED uint16_t sl[SCANF_TILE + 16];                 // compacted run starts
    CDBG_SHARED uint32_t s_wsum[SCAN_THREADS / 64], s_nstart, s_members, s_trav;
    CDBG_SHARED uint32_t s_defer[SCAN_DEFER_MAX];            // deferred placement: this workgroup's stream cursors

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = P.k, m = P.m, WN = k - m;
#if defined(CDBG_PROFILE_PHASES) && !defined(CDBG_HOSTSIM)
    uint64_t sph[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t st_prev = wall_clock64();
#endif
    uint32_t n_members = 0, n_trav = 0;
    if (tid == 0) { s_members = 0; s_trav = 0; }
    scan_defer_init<MODE>(P, s_defer);                       // (the first tile's barriers come before any record)
    // ---- A. load + encode, software-pipelined over the tiles (round 6) ----
    // The bytes of tile t + 1 are REQUESTED when tile t starts and encoded into the other LDS buffer just before tile t's records leave (phase E2).  Every
    // barrier inside a tile is LDS-only, so nothing waits for global memory except that encode -- and what it waits for, besides its own loads, are the
    // stores of tile t - 1, a whole tile old.  Before, a tile ended with __syncthreads() (s_waitcnt vmcnt(0): every record store of the tile acknowledged
    // before the next tile's loads were even issued) and began by waiting for its loads: two exposed round trips to memory per tile and workgroup.
    static_assert(SCANF_PKW <= 2 * SCAN_THREADS, "two 16-byte loads per lane cover a tile");
    uint4 pv0, pv1; bool pin0 = false, pin1 = false;
    auto tile_request = [&](uint64_t tl) {
        const int64_t b = (((int64_t)tl * P.tile_stride + P.tile_offset) * SCANF_TILE) - 16;
        const int64_t o0 = b + 16 * (int64_t)tid, o1 = b + 16 * (int64_t)(tid + SCAN_THREADS);
        pin0 = o0 >= 0 && o0 < (int64_t)P.nbytes_padded;
        pin1 = tid + SCAN_THREADS < SCANF_PKW && o1 >= 0 && o1 < (int64_t)P.nbytes_padded;
        if (pin0) pv0 = *reinterpret_cast<const uint4*>(P.reads + o0);
        if (pin1) pv1 = *reinterpret_cast<const uint4*>(P.reads + o1);
    };
    auto tile_encode = [&](int buf) {
        uint32_t* const pkd = pk2[buf]; uint16_t* const vmd = reinterpret_cast<uint16_t*>(vm2[buf]);
        auto enc = [&](const uint4& v, bool in, int w) {
            uint32_t packed = 0, vbits = 0;
            if (in) {
                uint32_t v0, v1, v2, v3;
                packed = (scanf_enc4(v.x, v0) << 24) | (scanf_enc4(v.y, v1) << 16) | (scanf_enc4(v.z, v2) << 8) | scanf_enc4(v.w, v3);
                vbits = v0 | (v1 << 4) | (v2 << 8) | (v3 << 12);
            }
            pkd[w] = packed; vmd[w] = (uint16_t)vbits;
        };
        enc(pv0, pin0, tid);
        if (tid + SCAN_THREADS < SCANF_PKW) enc(pv1, pin1, tid + SCAN_THREADS);
        if (tid < 4) vm2[buf][SCANF_PKW / 2 + tid] = 0;
        if (SCANF_PKW & 1) { if (tid == 4) vmd[SCANF_PKW] = 0; }
    };
    // (not for the long compile-time windows -- k = 55, m = 16: 39 keys -- whose register window leaves no room for eight more registers: the kernel spilled
    //  and the config-4 share's scan went from 49 to 58 ms; promised three waves instead of four it does not spill and takes 57.5 ms, with 24 bytes of
    //  scratch at four waves 59.3: there a tile loads its own bytes when it starts, as before)
    constexpr bool PIPE = !(WNT > 32);
    int cur = 0;
    if (PIPE && (uint64_t)blockIdx.x < P.n_tiles) { tile_request(blockIdx.x); tile_encode(0); }
    if (PIPE) CDBG_LDS_BARRIER();
    // persistent workgroups: a tile lives ~30 us, far too short to pay a workgroup launch for each
    for (uint64_t tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
    const bool has_next = PIPE && tile + gridDim.x < P.n_tiles;
    if (has_next) tile_request(tile + gridDim.x);
    if (!PIPE) {                                            // one 16-byte load at a time, encoded at once (four registers instead of eight)
        const int64_t b = (((int64_t)tile * P.tile_stride + P.tile_offset) * SCANF_TILE) - 16;
        for (int w = tid; w < SCANF_PKW; w += SCAN_THREADS) {
            const int64_t off = b + 16 * (int64_t)w;
            uint32_t packed = 0, vbits = 0;
            if (off >= 0 && off < (int64_t)P.nbytes_padded) {
                const uint4 v = *reinterpret_cast<const uint4*>(P.reads + off);
                uint32_t v0, v1, v2, v3;
                packed = (scanf_enc4(v.x, v0) << 24) | (scanf_enc4(v.y, v1) << 16) | (scanf_enc4(v.z, v2) << 8) | scanf_enc4(v.w, v3);
                vbits = v0 | (v1 << 4) | (v2 << 8) | (v3 << 12);
            }
            pk2[0][w] = packed;
            reinterpret_cast<uint16_t*>(vm2[0])[w] = (uint16_t)vbits;
        }
        if (tid < 4) vm2[0][SCANF_PKW / 2 + tid] = 0;
        if (SCANF_PKW & 1) { if (tid == 4) reinterpret_cast<uint16_t*>(vm2[0])[SCANF_PKW] = 0; }
        CDBG_LDS_BARRIER();
    }
    uint32_t* const pk = pk2[cur]; uint32_t* const vm = vm2[cur];
    CDBG_SPH(0);

    // ---- B. rolling m-mer keys: lane chunk c covers m-mer starts [16c, 16c+16) ----
    const int nq_keys = 15 + SCANF_TILE + 2 + WN;          // keys needed for q < nq_keys
    const uint32_t mmask = m == 16 ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);
    for (int c = tid; 16 * c < nq_keys; c += SCAN_THREADS) {
        // the 31 bases 16c .. 16c+30 as one 64-bit string X (first base on top) and its reverse complement
        // RX: the m-mer starting at base s is a bit field of X, its reverse complement a bit field of RX
        const uint32_t w0 = pk[c], w1 = pk[c + 1];
        const uint32_t vv = (uint32_t)reinterpret_cast<const uint16_t*>(vm)[c] | ((uint32_t)reinterpret_cast<const uint16_t*>(vm)[c + 1] << 16);
        uint32_t xv = vv;                                  // bit s <- bases s .. s+m-1 all valid (AND by doubling)
        { int len = 1; while (2 * len <= m) { xv &= xv >> len; len *= 2; } xv &= xv >> (m - len); }
        const uint64_t X = ((uint64_t)w0 << 32) | w1;
        const uint64_t RX = ~(((uint64_t)rev2_32(w1) << 32) | rev2_32(w0));
        const int fsh = 64 - 2 * m;
        uint32_t* dst = kg + 17 * c;                       // scanf_pad(16c + s) = 17c + s for s < 16
        uint32_t bmin = 0xFFFFFFFFu;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const uint32_t fw = (uint32_t)(X >> (fsh - 2 * s)) & mmask;
            const uint32_t rc = (uint32_t)(RX >> (2 * s)) & mmask;
            const uint32_t key = mix32(rc < fw ? rc : fw) | (((xv >> s) & 1u) - 1u);      // invalid m-mer -> 0xFFFFFFFF
            dst[s] = key;
            if (WNT < 0) bmin = key < bmin ? key : bmin;
        }
        if (WNT < 0) blk[c] = bmin;
    }
    if (WNT < 0) { for (int c = (nq_keys + 15) / 16 + tid; c < SCANF_NQ / 16 + 8; c += SCAN_THREADS) blk[c] = 0xFFFFFFFFu; }   // (blocks past the keys: never the minimum)
    CDBG_LDS_BARRIER();
    CDBG_SPH(1);

    // ---- C. g[q] = min of keys[q .. q+WN-1], 16 junctions per lane from a register window ----
    // ---- D. validity and run-break bits ----
    uint32_t gq[16];
    const int nchunk_g = (15 + SCANF_TILE + 2 + 15) / 16;   // chunks containing junction q <= TILE+16
    for (int c = tid; c < nchunk_g; c += SCAN_THREADS) {    // (one iteration: nchunk_g <= 256)
        if (WNT < 0) {
            // two-level: window of junction 16c + j = keys [16c + j, 16c + j + WN).  With E = (WN - 1) / 16, R = (WN - 1) % 16 it ends in block
            // c + E at offset j + R (j + R < 16) or in block c + E + 1 at offset j + R - 16; blocks c + 1 .. c + E - 1 are covered whole
            // (and block c + E as well in the second case).  WN >= 17 (E >= 1).
            const int E = (WN - 1) >> 4, R = (WN - 1) & 15;
            uint32_t a[16], pe[16], pf[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { a[i] = kg[17 * c + i]; pe[i] = kg[17 * (c + E) + i]; pf[i] = kg[17 * (c + E + 1) + i]; }
            uint32_t mid = 0xFFFFFFFFu;                           // whole blocks c + 1 .. c + E - 1
            for (int b = 1; b < E; ++b) { const uint32_t v = blk[c + b]; mid = v < mid ? v : mid; }
            const uint32_t full_e = blk[c + E];
#pragma unroll
            for (int i = 14; i >= 0; --i) a[i] = a[i] < a[i + 1] ? a[i] : a[i + 1];          // suffix minima of the own block
#pragma unroll
            for (int i = 1; i < 16; ++i) { pe[i] = pe[i] < pe[i - 1] ? pe[i] : pe[i - 1]; pf[i] = pf[i] < pf[i - 1] ? pf[i] : pf[i - 1]; }   // prefix minima
            // tail[j] = (pe ++ pf)[j + R], and the whole block c + E joins where the window runs past it: R is wave-uniform, so one of
            // sixteen fully unrolled cases runs (constant register indices; a select chain over 32 registers per junction otherwise)
#define CDBG_TL_CASE(RR) case RR: { _Pragma("unroll") for (int j = 0; j < 16; ++j) { const int idx = j + RR; uint32_t v = a[j] < mid ? a[j] : mid;   \
                if (idx >= 16) v = v < full_e ? v : full_e;                                                                                    \
                const uint32_t tv = idx < 16 ? pe[idx < 16 ? idx : 0] : pf[idx >= 16 ? idx - 16 : 0]; gq[j] = v < tv ? v : tv; } } break;
            switch (R) { CDBG_TL_CASE(0) CDBG_TL_CASE(1) CDBG_TL_CASE(2) CDBG_TL_CASE(3) CDBG_TL_CASE(4) CDBG_TL_CASE(5) CDBG_TL_CASE(6) CDBG_TL_CASE(7)
                         CDBG_TL_CASE(8) CDBG_TL_CASE(9) CDBG_TL_CASE(10) CDBG_TL_CASE(11) CDBG_TL_CASE(12) CDBG_TL_CASE(13) CDBG_TL_CASE(14) default: CDBG_TL_CASE(15) }
#undef CDBG_TL_CASE
        } else if (WNT == 15) {
            // window of exactly 15 keys (k = 31, m = 16): g[j] = min(a[j..14]) min min(a[15..j+14]), i.e. a
            // suffix minimum of the first 15 keys and a prefix minimum of the next 15: 42 min instead of 224
            uint32_t a[30];
#pragma unroll
            for (int i = 0; i < 30; ++i) a[i] = kg[17 * c + i + (i >> 4)];
#pragma unroll
            for (int i = 13; i >= 0; --i) a[i] = a[i] < a[i + 1] ? a[i] : a[i + 1];
#pragma unroll
            for (int i = 16; i < 30; ++i) a[i] = a[i] < a[i - 1] ? a[i] : a[i - 1];
            gq[0] = a[0]; gq[15] = a[29];
#pragma unroll
            for (int j = 1; j < 15; ++j) gq[j] = a[j] < a[j + 14] ? a[j] : a[j + 14];
        } else if (WNT == 16) {
            // window of exactly 16 keys (k = 31, m = 15): the window of junction j is the tail a[j..15] of the lane's own block and
            // the head a[16..j+15] of the next: suffix minima of one, prefix minima of the other, 44 min
            uint32_t a[31];
#pragma unroll
            for (int i = 0; i < 31; ++i) a[i] = kg[17 * c + i + (i >> 4)];
#pragma unroll
            for (int i = 14; i >= 0; --i) a[i] = a[i] < a[i + 1] ? a[i] : a[i + 1];
#pragma unroll
            for (int i = 17; i < 31; ++i) a[i] = a[i] < a[i - 1] ? a[i] : a[i - 1];
            gq[0] = a[0];
#pragma unroll
            for (int j = 1; j < 16; ++j) gq[j] = a[j] < a[j + 15] ? a[j] : a[j + 15];
        } else if (WNT > 0) {
            // any other compile-time window (k = 55, m = 16: 39 keys): minima over 2, 4, ... 2^p <= WNT keys by doubling in
            // place, then g[j] = min of the two 2^p-blocks that cover [j, j + WNT): ~ (16 + WNT) log2(WNT) min instead of 16 WNT
            constexpr int N = 16 + WNT - 1;
            uint32_t a[N];
#pragma unroll
            for (int i = 0; i < N; ++i) a[i] = kg[scanf_pad(16 * c + i)];
            constexpr int L = WNT >= 32 ? 32 : WNT >= 16 ? 16 : WNT >= 8 ? 8 : WNT >= 4 ? 4 : WNT >= 2 ? 2 : 1;
#pragma unroll
            for (int len = 1; len < L; len *= 2) {
#pragma unroll
                for (int i = 0; i + len < N; ++i) a[i] = a[i] < a[i + len] ? a[i] : a[i + len];   // ascending i: a[i + len] is still the previous level
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) gq[j] = a[j] < a[j + WNT - L] ? a[j] : a[j + WNT - L];
        } else {
            uint32_t a[16 + SCANF_WNMAX - 1];
#pragma unroll
            for (int i = 0; i < 16 + SCANF_WNMAX - 1; ++i) a[i] = (i < 16 + WN - 1) ? kg[scanf_pad(16 * c + i)] : 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < 16; ++j) gq[j] = a[j];
#pragma unroll
            for (int w = 1; w < SCANF_WNMAX; ++w) {
                if (w < WN) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) gq[j] = gq[j] < a[j + w] ? gq[j] : a[j + w];
                }
            }
        }
    }
    CDBG_LDS_BARRIER();                                        // every lane has read its keys
    if (tid < nchunk_g) {
#pragma unroll
        for (int j = 0; j < 16; ++j) kg[scanf_pad(16 * tid + j)] = gq[j];
    }
    CDBG_LDS_BARRIER();
    CDBG_SPH(2);
    {
        // validity window: bit i <-> base 16*tid + i - 16 (i.e. starts one chunk earlier, for q-1 tests)
        const int c = tid;
        uint32_t brk16 = 0, stt16 = 0;
        if (c >= 1 && c <= SCANF_TILE / 16) {               // own junctions: q in [16, 16+TILE)
            const uint16_t* v16 = reinterpret_cast<const uint16_t*>(vm);
            uint32_t vj, vp;
            if (WNT < 0) {
                // k - 1 up to 126 bases behind each of the 17 junctions: a 192-bit window (three words), the same AND-doubling
                auto h4 = [&](int i) -> uint64_t { return (uint64_t)v16[i] | ((uint64_t)v16[i + 1] << 16) | ((uint64_t)v16[i + 2] << 32) | ((uint64_t)v16[i + 3] << 48); };
                const uint64_t w0 = h4(c - 1), w1 = h4(c + 3), w2 = h4(c + 7), w3 = (uint64_t)v16[c + 11];
                uint64_t Y[3] = { (w0 >> 15) | (w1 << 49), (w1 >> 15) | (w2 << 49), (w2 >> 15) | (w3 << 49) };
                auto yw = [&](int i) -> uint64_t { uint64_t r = 0; _Pragma("unroll") for (int j = 0; j < 3; ++j) r = (i == j) ? Y[j] : r; return r; };
                auto shr_and = [&](int sft) {                                   // Y &= Y >> sft (zero fill), sft wave-uniform, 1 .. 126
                    const int ws = sft >> 6, bs = sft & 63;
#pragma unroll
                    for (int i = 0; i < 3; ++i) { const uint64_t a0 = yw(i + ws), a1 = yw(i + ws + 1); Y[i] &= bs ? ((a0 >> bs) | (a1 << (64 - bs))) : a0; }
                };
                { const int K1 = k - 1; int len = 1; while (2 * len <= K1) { shr_and(len); len *= 2; } if (K1 > len) shr_and(K1 - len); }
                vj = (uint32_t)(Y[0] >> 1) & 0xFFFFu; vp = (uint32_t)Y[0] & 0xFFFFu;
            } else {
            const uint64_t lo = (uint64_t)v16[c - 1] | ((uint64_t)v16[c] << 16) | ((uint64_t)v16[c + 1] << 32) | ((uint64_t)v16[c + 2] << 48);
            const uint64_t hi = (uint64_t)v16[c + 3] | ((uint64_t)v16[c + 4] << 16);
            // bit t of Y <- the k-1 bases starting at window bit 15+t are all valid (AND by doubling): bit j+1 is
            // junction q = 16c+j, bit j is junction q-1
            unsigned __int128 Y = (((unsigned __int128)hi << 64) | lo) >> 15;
            { const int K1 = k - 1; int len = 1; while (2 * len <= K1) { Y &= Y >> len; len *= 2; } Y &= Y >> (K1 - len); }
            vj = (uint32_t)(Y >> 1) & 0xFFFFu; vp = (uint32_t)Y & 0xFFFFu;
            }
            const uint32_t gprev = kg[scanf_pad(16 * c - 1)];
            uint32_t eq = gq[0] == gprev ? 1u : 0u;
#pragma unroll
            for (int j = 1; j < 16; ++j) eq |= (gq[j] == gq[j - 1] ? 1u : 0u) << j;
            uint32_t cont = vj & vp & eq;
            if (c == 1) cont &= ~1u;                        // the tile's first junction never continues a run
            brk16 = ~cont & 0xFFFFu;
            stt16 = vj & ~cont;
        } else {
            brk16 = 0xFFFFu;                                // outside the tile's own junctions: always a break
        }
        reinterpret_cast<uint16_t*>(brk)[c] = (uint16_t)brk16;
        reinterpret_cast<uint16_t*>(stt)[c] = (uint16_t)stt16;
        if (tid < 24) reinterpret_cast<uint16_t*>(brk)[SCAN_THREADS + tid] = 0xFFFFu;   // sentinels past the tile

        // ---- E1. compact the run starts into sl[] ----
        const int cnt = __popc(stt16);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int x = __shfl_up(incl, d); if (lane >= d) incl += x; }
        if (lane == 63) s_wsum[wave] = (uint32_t)incl;
        CDBG_LDS_BARRIER();
        int off = incl - cnt;
        for (int w = 0; w < wave; ++w) off += (int)s_wsum[w];
        uint32_t bits = stt16;
        while (bits) { const int j = __ffs((int)bits) - 1; bits &= bits - 1; sl[off++] = (uint16_t)(16 * c + j); }
        if (tid == SCAN_THREADS - 1) s_nstart = (uint32_t)off;
    }
    CDBG_LDS_BARRIER();
    CDBG_SPH(3);

    if (has_next) tile_encode(cur ^ 1);                     // (the only wait for global memory in a tile: its loads had phases B - E1 to arrive)
    // ---- E2. one lane per run ----
    const int NMAX = CAPB - k + 1;
    const uint32_t rank_mask = (1u << P.rank_bits) - 1u;
    const int nstart = (int)s_nstart;
    for (int i = tid; i < nstart; i += SCAN_THREADS) {
        const int s = sl[i];                                // run = junctions [s, e] in q space
        const uint32_t g0 = kg[scanf_pad(s)];
        const uint32_t part = part_of(g0, P.log_np);
        // own partitions only (every rank scans the same text), or -- reads sharded over the ranks -- all partitions, laid
        // out owner-major ([owner][local partition]) so that each owner's block of the record array is contiguous
        if (!P.emit_all && (part & rank_mask) != (uint32_t)P.rank) continue;
        const uint32_t lpart = P.emit_all ? (part & rank_mask) * P.npl + (part >> P.rank_bits) : part >> P.rank_bits;
        int e;
        {
            int nb = s + 1;
            uint32_t wv = brk[nb >> 5] >> (nb & 31);
            if (wv) e = nb + __ffs((int)wv) - 1;
            else { int wi = (nb >> 5) + 1; while (brk[wi] == 0) ++wi; e = wi * 32 + __ffs((int)brk[wi]) - 1; }
            e -= 1;                                         // last junction of the run
            if (e > 15 + SCANF_TILE) e = 15 + SCANF_TILE;
        }
        // validity of the boundary k-mers: k-mer at q-1 (right junction s) and k-mer at e (left junction e)
        bool first_incl = false, first_trav = false, last_incl = false, last_trav = false, first_foreign = false, last_foreign = false;
        if (scan_all_valid(vm, s - 1, k)) {
            const uint32_t g2 = kg[scanf_pad(s - 1)];
            first_foreign = part_of(g2, P.log_np) != part;   // the junction this k-mer shares with the run before belongs to another bucket
            if (g0 < g2) first_incl = true;
            else if (first_foreign) { first_incl = true; first_trav = true; }
        }
        if (scan_all_valid(vm, e, k)) {
            const uint32_t g2 = kg[scanf_pad(e + 1)];
            last_foreign = part_of(g2, P.log_np) != part;
            if (g0 < g2) last_incl = true;
            else if (last_foreign) { last_incl = true; last_trav = true; }
            else if (g0 == g2) last_incl = true;
        }
        int c = s; bool firstchunk = true;
        while (c <= e) {
            const int ms = (firstchunk && first_incl) ? c - 1 : c;
            int ce = ms + NMAX - 1; if (ce > e) ce = e;
            const int me = (ce == e && !last_incl) ? e - 1 : ce;
            const int n = me - ms + 1;
            if (n > 0) {
                uint32_t meta = (uint32_t)n | (sub_of(g0, P.log_np) << 12);   // (bits 12-15: the minimizer's sub-partition, for the multi-pass count)
                const bool ft = firstchunk && first_incl && first_trav;
                const bool lt = (ce == e) && last_incl && last_trav;
                if (ft) meta |= 0x100u;
                if (lt) meta |= 0x200u;
                if (firstchunk && first_incl && first_foreign) meta |= 0x400u;
                if ((ce == e) && last_incl && last_foreign) meta |= 0x800u;
                scan_emit_record<W, MODE>(P, pk, 2 * ms, meta, lpart, s_defer);
                n_members += (uint32_t)n;
                n_trav += (ft ? 1u : 0u) + (lt ? 1u : 0u);
            }
            firstchunk = false;
            c = ce + 1;
        }
    }
    CDBG_LDS_BARRIER();                                        // the next tile reuses the LDS arrays (the records' words were read from them before this barrier; their stores drain behind it)
    CDBG_SPH(4);
    if (PIPE) cur ^= 1;
    }
#if defined(CDBG_PROFILE_PHASES) && !defined(CDBG_HOSTSIM)
    if (threadIdx.x == 0) for (int i = 0; i < 6; ++i) atomic_add_u64(&P.stats[16 + i], sph[i]);
#endif
    scan_defer_publish<MODE>(P, s_defer);                    // (behind the last tile's closing barrier)
    if (MODE != SCAN_EMIT || P.var_limit) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { n_members += __shfl_xor(n_members, d); n_trav += __shfl_xor(n_trav, d); }
        if (lane == 0) { if (n_members) atomic_add_u32(&s_members, n_members); if (n_trav) atomic_add_u32(&s_trav, n_trav); }
        __syncthreads();
        if (tid == 0 && s_members) { atomic_add_u64(&P.stats[0], (uint64_t)s_members); atomic_add_u64(&P.stats[1], (uint64_t)s_trav); }
    }
}

}  // namespace cdbg
