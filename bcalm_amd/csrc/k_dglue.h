// k_dglue.h -- stage 3 across GPUs: the glue of k_glue.h with every array SHARDED by owner.
//
// The reference has nothing here: bglue runs inside the one shared-memory call
// GraphUnitigsTemplate<span>::create (/root/reference/src/bcalm_1.cpp:57).  Design: SURVEY.md section 8(e), with
// north_star's "RCCL all-to-all over xGMI to exchange cross-bucket glue records" taken literally:
//
//   1 junction join   every glue record (junction key, piece end) travels ONCE, to the rank its key hashes to
//                     (all-to-all-v); that rank joins its keys in LDS (k_join_bucket) and sends every joined pair
//                     (end, partner) to the rank that owns the end (all-to-all-v)
//   2 list ranking    traversal states stay with the rank that owns their piece; pointer jumping runs locally while the
//                     successor is local, and a jump over a remote state is a query / reply pair of all-to-all-v's per
//                     round (8-byte states: "y k-mers ahead lies state x", exactly k_rank8_jump's invariant)
//   3 emission        a unitig belongs to the rank that owns its head piece: every piece goes ONCE (lengths, abundance
//                     sums, bases 2 bit packed) to that rank, which places it with k_emit
//
// so a rank's traffic is ~1/N of the graph instead of all of it (the replicated exchange of rounds 1-2 all-gathered every
// piece and the whole junction log to every rank and ranked the whole graph N times: projected 2.7 x on 8 GPUs).
// Closed chains that cross ranks never finish ranking: the host then falls back to the replicated exchange, which can
// cut cycles (rare: isolated circular unitigs).
#pragma once
#include "k_glue.h"

namespace cdbg {

constexpr int DG_MAX_WORLD = 64;
constexpr int DG_THREADS = 256;
constexpr uint8_t DG_NODEST = 0xFF;

// piece id ranges of the ranks: rank r owns global pieces [b[r], b[r + 1]); global end / state id = 2 * piece + side
struct DgOwners {
    uint32_t b[DG_MAX_WORLD + 1]; int world;
    CDBG_HD uint32_t of_piece(uint32_t pid) const { int r = 0; while (r + 1 < world && pid >= b[r + 1]) ++r; return (uint32_t)r; }
};

// ---- routing: items -> per-destination blocks of a send buffer (two passes: count, place) ----
struct RouteParams {
    uint64_t n; const uint8_t* dest;         // destination rank of every item (DG_NODEST: the item does not travel)
    uint64_t* cnt;                           // [world] pass 1: items per destination
    const uint64_t* off; uint64_t* cur;      // [world] pass 2: block offsets (exclusive scan of cnt, host) and running cursors (zeroed)
    uint64_t* pos;                           // pass 2: position of every item in the send buffer
    int world;
};
__global__ void __launch_bounds__(DG_THREADS) k_route_count(RouteParams P) {
    CDBG_SHARED uint32_t h[DG_MAX_WORLD];
    if (threadIdx.x < DG_MAX_WORLD) h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += stride) {
        const uint8_t d = P.dest[i];
        if (d != DG_NODEST) atomic_add_u32(&h[d], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x < P.world && h[threadIdx.x]) atomic_add_u64(&P.cnt[threadIdx.x], (uint64_t)h[threadIdx.x]);
}
constexpr int DG_ITEMS = 4;                              // items per thread and tile: three barriers per 1024 items
__global__ void __launch_bounds__(DG_THREADS) k_route_place(RouteParams P) {
    CDBG_SHARED uint32_t h[DG_MAX_WORLD]; CDBG_SHARED uint64_t base[DG_MAX_WORLD];
    constexpr uint64_t TILE = (uint64_t)DG_THREADS * DG_ITEMS;
    const uint64_t tiles = (P.n + TILE - 1) / TILE;
    for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {          // uniform trip count per workgroup
        if (threadIdx.x < DG_MAX_WORLD) h[threadIdx.x] = 0;
        __syncthreads();
        uint8_t d[DG_ITEMS]; uint32_t r[DG_ITEMS];
#pragma unroll
        for (int j = 0; j < DG_ITEMS; ++j) {
            const uint64_t i = t * TILE + (uint64_t)j * DG_THREADS + threadIdx.x;
            d[j] = DG_NODEST; r[j] = 0;
            if (i < P.n) { d[j] = P.dest[i]; if (d[j] != DG_NODEST) r[j] = atomic_add_u32(&h[d[j]], 1u); }
        }
        __syncthreads();
        if ((int)threadIdx.x < P.world && h[threadIdx.x]) base[threadIdx.x] = atomic_add_u64(&P.cur[threadIdx.x], (uint64_t)h[threadIdx.x]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < DG_ITEMS; ++j) {
            const uint64_t i = t * TILE + (uint64_t)j * DG_THREADS + threadIdx.x;
            if (d[j] != DG_NODEST) P.pos[i] = P.off[d[j]] + base[d[j]] + r[j];
        }
        __syncthreads();
    }
}

// ---- 1a. junction log -> key owners.  Wire record: W key words + one word holding the GLOBAL tag ----
struct LogRouteParams {
    const uint64_t* glog_keys; const uint32_t* glog_tag; uint64_t n; int world; uint32_t end_base;   // end_base = 2 * first global piece of this rank
    uint8_t* dest; const uint64_t* pos; uint64_t* wire;
};
template <int W>
__global__ void k_log_dest(LogRouteParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += stride) {
        uint8_t d = DG_NODEST;
        if (P.glog_tag[i] != GTAG_EMPTY) {
            Kmer<W> jc;
            for (int j = 0; j < W; ++j) jc.w[j] = P.glog_keys[i * W + j];
            d = (uint8_t)(mix32(jc.hash()) & (uint32_t)(P.world - 1));
        }
        P.dest[i] = d;
    }
}
template <int W>
__global__ void k_log_write(LogRouteParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += stride) {
        if (P.dest[i] == DG_NODEST) continue;
        const uint64_t o = P.pos[i] * (W + 1);
        for (int j = 0; j < W; ++j) P.wire[o + j] = P.glog_keys[i * W + j];
        const uint32_t t = P.glog_tag[i];
        P.wire[o + W] = t == GTAG_CONFIRM ? (uint64_t)t : (uint64_t)(((t & ~GTAG_CONFBIT) + P.end_base) | (t & GTAG_CONFBIT));
    }
}
// receiver: wire records -> join buckets (k_glue.h)
struct WireScatterParams { const uint64_t* wire; uint64_t n; int log_jb; uint32_t* jfill; uint64_t* jrecs; uint32_t* error; };
template <int W>
__global__ void k_join_scatter_wire(WireScatterParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += stride) {
        Kmer<W> jc;
        for (int j = 0; j < W; ++j) jc.w[j] = P.wire[i * (W + 1) + j];
        join_bucket_put<W>(P.jfill, P.jrecs, P.log_jb, P.error, jc, (uint32_t)P.wire[i * (W + 1) + W]);
    }
}

// ---- 1b. joined pairs (end, partner) -> end owners ----
struct PairRouteParams { const uint2* pairs; uint64_t n; DgOwners own; uint8_t* dest; const uint64_t* pos; uint2* wire; };
__global__ void k_pair_dest(PairRouteParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += stride) { const uint32_t e = P.pairs[i].x; P.dest[i] = e == NONE32 ? DG_NODEST : (uint8_t)P.own.of_piece(e >> 1); }   // (NONE32: unused tail of a wave's chunk)
}
__global__ void k_pair_write(PairRouteParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += stride) if (P.dest[i] != DG_NODEST) P.wire[P.pos[i]] = P.pairs[i];
}
struct PairApplyParams { const uint2* pairs; uint64_t n; uint32_t end_base; uint32_t* link; };
__global__ void k_pair_apply(PairApplyParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += stride) { const uint2 p = P.pairs[i]; P.link[p.x - P.end_base] = p.y; }
}

// ---- 2. distributed list ranking ----
// state of local traversal state e (global id e + base): {x, y}: x = successor (GLOBAL state id), or RANK_TAIL | tail (global)
struct DRankParams {
    uint32_t n_local, base; DgOwners own; int me;
    const uint32_t* link; const uint32_t* piece_n;       // local arrays (link values are global end ids)
    uint2* st;
    uint8_t* dest; const uint64_t* pos;                  // routing of this round's queries
    uint32_t* q_send; uint32_t* q_src;                   // query = the remote state id; q_src[pos] = the local state that asked
    const uint32_t* q_recv; uint2* r_send; uint64_t n_recv;   // owner side: queries received, replies
    const uint2* r_recv; uint64_t n_sent;                // asker side: replies, in the order of q_send
};
__global__ void k_dr_init(DRankParams P) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.n_local) return;
    const uint32_t nxt = P.link[e ^ 1u];
    uint2 v; v.x = nxt == NONE32 ? (RANK_TAIL | (e + P.base)) : nxt; v.y = P.piece_n[e >> 1];
    P.st[e] = v;
}
// in place and asynchronous like k_rank8_jump: jumps while the successor is local, then names the rank to ask
__global__ void k_dr_jump(DRankParams P) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.n_local) return;
    uint2 v = P.st[e];
    uint8_t d = DG_NODEST;
    if (!(v.x & RANK_TAIL)) {
#pragma unroll 1
        for (int j = 0; j < 2 * RANK8_JUMPS; ++j) {
            const uint32_t x = v.x;
            if (x < P.base || x - P.base >= P.n_local) break;            // remote successor
            const uint2 t = P.st[x - P.base];
            v.y += t.y; v.x = t.x;
            if (v.x & RANK_TAIL) break;
        }
        P.st[e] = v;
        if (!(v.x & RANK_TAIL)) d = (uint8_t)P.own.of_piece(v.x >> 1);   // (a local successor after the jump budget asks this rank itself)
    }
    P.dest[e] = d;
}
__global__ void k_dr_query(DRankParams P) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.n_local || P.dest[e] == DG_NODEST) return;
    const uint64_t o = P.pos[e];
    P.q_send[o] = P.st[e].x; P.q_src[o] = e;
}
__global__ void k_dr_reply(DRankParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n_recv; i += stride) P.r_send[i] = P.st[P.q_recv[i] - P.base];
}
__global__ void k_dr_apply(DRankParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n_sent; i += stride) {
        const uint32_t e = P.q_src[i]; const uint2 t = P.r_recv[i];
        uint2 v = P.st[e]; v.y += t.y; v.x = t.x; P.st[e] = v;
    }
}

// ---- closed chains.  Pointer jumping never finishes a state that lies on a cycle; once a round finishes no state at all, what
// is left are exactly the cycles (isolated circular unitigs: plasmids, organelles).  They are few: every rank lists its
// unfinished states with their original successor, the lists are all-gathered, every rank elects the same cut -- the junction
// at the left end of the smallest piece of each cycle -- clears its own ends of those junctions and the ranking starts again.
struct DrOpenParams { uint32_t n_local, base; const uint2* st; const uint32_t* link; uint2* out; uint64_t* cursor; uint64_t cap; };
__global__ void k_dr_collect_open(DrOpenParams P) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.n_local || (P.st[e].x & RANK_TAIL)) return;
    const uint64_t i = atomic_add_u64(P.cursor, 1ull);
    if (i < P.cap) { uint2 v; v.x = e + P.base; v.y = P.link[e ^ 1u]; P.out[i] = v; }
}
struct DrCutParams { const uint32_t* ends; uint32_t n; uint32_t base, n_local; uint32_t* link; };
__global__ void k_dr_cut(DrCutParams P) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n) return;
    const uint32_t e = P.ends[i];
    if (e >= P.base && e - P.base < P.n_local) P.link[e - P.base] = NONE32;
}

// ---- 3. unitigs: size of what this rank will emit (its head states), then pieces -> head owners ----
struct HeadMeasureParams { uint32_t n_states; int k; const uint32_t* link; const uint2* st; uint64_t* out; };   // out[0] unitigs, out[1] bases
__global__ void k_heads_measure(HeadMeasureParams P) {
    const uint32_t stride = gridDim.x * blockDim.x;
    uint64_t cnt = 0, sum = 0;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < P.n_states; e += stride) {
        if (P.link[e] != NONE32) continue;
        const uint2 v = P.st[e];
        if (v.y != 0 && v.x > P.st[e ^ 1u].x) { ++cnt; sum += (uint64_t)v.y + (uint64_t)P.k - 1u; }
    }
    cnt = wave_sum_u64(cnt); sum = wave_sum_u64(sum);
    if ((threadIdx.x & 63) == 0 && cnt) { atomic_add_u64(&P.out[0], cnt); atomic_add_u64(&P.out[1], sum); }
}
// wire of a piece: three words {head state (global) | k-mers from the piece's entry state to the tail << 32,
//                               k-mers of the piece | reversed << 31, abundance sum}; its bases travel in a packed stream
struct PieceRouteParams {
    uint32_t n_pieces; int k; DgOwners own;
    const uint2* st; const uint32_t* piece_n; const uint64_t* piece_kc; const uint64_t* piece_boff;
    uint8_t* dest; const uint64_t* pos;
    uint64_t* meta; uint32_t* lens; uint64_t* boff; uint32_t* alen;      // in send-buffer order (alen: k-mers, for the abundance stream)
};
CDBG_DEV void dg_piece_head(const uint2* st, uint32_t p, uint32_t& head, uint32_t& dist, bool& rev) {
    const uint4 s01 = reinterpret_cast<const uint4*>(st)[p];
    const uint32_t z0 = s01.x & ~RANK_TAIL, z1 = s01.z & ~RANK_TAIL;
    rev = !(z0 > z1);                                                    // k_emit's rule: the direction with the larger tail
    head = (rev ? z0 : z1) ^ 1u; dist = rev ? s01.w : s01.y;
}
__global__ void k_piece_dest(PieceRouteParams P) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.n_pieces) return;
    uint8_t d = DG_NODEST;
    if (P.piece_n[p]) { uint32_t h, di; bool rv; dg_piece_head(P.st, p, h, di, rv); d = (uint8_t)P.own.of_piece(h >> 1); }
    P.dest[p] = d;
}
__global__ void k_piece_write(PieceRouteParams P) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.n_pieces || P.dest[p] == DG_NODEST) return;
    uint32_t h, di; bool rv; dg_piece_head(P.st, p, h, di, rv);
    const uint64_t o = P.pos[p]; const uint32_t n = P.piece_n[p];
    P.meta[3 * o] = (uint64_t)h | ((uint64_t)di << 32); P.meta[3 * o + 1] = (uint64_t)(n | (rv ? 0x80000000u : 0u)); P.meta[3 * o + 2] = P.piece_kc[p];
    P.lens[o] = n + (uint32_t)P.k - 1u; P.boff[o] = P.piece_boff[p]; P.alen[o] = n;
}
// receiver: metas -> the piece arrays k_emit reads.  The two traversal states of a received piece are synthesised so that
// k_emit's own rule picks the direction and the head that the sender computed: the entry state gets the largest tail.
struct PieceRecvParams { uint64_t n; int k; uint32_t base; const uint64_t* meta; uint32_t* piece_n; uint64_t* piece_kc; uint4* st; uint32_t* lens; };
__global__ void k_piece_recv(PieceRecvParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += stride) {
        const uint64_t m0 = P.meta[3 * i], m1 = P.meta[3 * i + 1];
        const uint32_t h = (uint32_t)m0 - P.base, dist = (uint32_t)(m0 >> 32), n = (uint32_t)m1 & 0x7FFFFFFFu; const bool rev = (m1 >> 31) & 1u;
        P.piece_n[i] = n; P.piece_kc[i] = P.meta[3 * i + 2]; P.lens[i] = n + (uint32_t)P.k - 1u;
        uint4 s; const uint32_t big = RANK_TAIL | 0x7FFFFFFFu, low = RANK_TAIL | (h ^ 1u);
        if (!rev) { s.x = big; s.y = dist; s.z = low; s.w = 0; } else { s.x = low; s.y = 0; s.z = big; s.w = dist; }
        P.st[i] = s;
    }
}
__global__ void k_add_u64(uint64_t* a, uint64_t n, uint64_t v) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += v;
}

}  // namespace cdbg
