// host_glue_sharded.h -- libcdbg.so, host side: the glue stage SHARDED by owner across the ranks of a multi-GPU job
// (k_dglue.h).  Included by cdbg_impl.cpp only.
#pragma once

namespace {

// =======================================================================================
// Sharded glue across ranks (k_dglue.h): join by key owner, ranking by piece owner, emission by head owner.
// Returns DG_FALLBACK (every rank, together) when the distributed ranking does not converge -- closed chains that cross
// ranks -- and the caller then runs the replicated exchange, which can cut cycles.
// =======================================================================================
// The stage's end state: the unitig arena at 2 bits per base next to the ASCII one (one streaming pass:
// 64 bases -> 16 bytes per lane; base i of the arena = bits [2 (i & 3), 2 (i & 3) + 2) of byte i >> 2, A0 C1 G2 T3)
int pack_unitigs(cdbg_ctx* c) {
    const uint64_t chunks = (c->unitig_total + 63) / 64;
    CK(c->unitig_packed.alloc(chunks * 16 + 16, false));
    if (chunks) { StreamPackParams pp{ chunks, c->unitig_bases.p, c->unitig_packed.p, c->unitig_total }; CDBG_LAUNCH(k_pack_stream, (chunks + 255) / 256, 256, c->stream, pp); }
    return CDBG_OK;
}
constexpr int DG_FALLBACK = 1;
struct DgRoute { std::vector<uint64_t> scnt, soff, rcnt, roff, all; uint64_t n_send = 0, n_recv = 0, n_all = 0; };
// positions of the n items whose destinations are in c->dg_dest: send-block counts / offsets, receive counts / offsets
// (extra: n_extra more words of this rank ride in the same small all-gather -- extra_all[r * n_extra + i] = word i of rank r: the
//  piece counts and the status words that used to cost a host-synchronous collective of their own)
// `pending`: a rank-local failure of the caller since the last exchange (an allocation, a consistency check).  It travels with the counts -- as does
// a failure inside the routing itself -- so that every rank leaves together instead of one rank returning while its peers wait in the transport.
int dg_route(cdbg_ctx* c, uint64_t n, DgRoute& R, const uint64_t* extra = nullptr, int n_extra = 0, std::vector<uint64_t>* extra_all = nullptr, int pending = CDBG_OK) {
    const int world = c->prm.world_size, me = c->prm.rank; hipStream_t s = c->stream;
    R.scnt.assign(world, 0); R.soff.assign(world + 1, 0); R.rcnt.assign(world, 0); R.roff.assign(world + 1, 0); R.all.assign((size_t)world * world, 0);
    auto local = [&]() -> int {                          // this rank's part: counts per destination, positions of the items in the send buffer
        CK(pending);
        CK(c->dg_cnt.alloc(3 * DG_MAX_WORLD, true)); CK(c->dg_pos.alloc(n, false));
        RouteParams rp{ n, c->dg_dest.p, c->dg_cnt.p, c->dg_cnt.p + DG_MAX_WORLD, c->dg_cnt.p + 2 * DG_MAX_WORLD, c->dg_pos.p, world };
        if (n) CDBG_LAUNCH(k_route_count, std::min<uint64_t>((n + DG_THREADS - 1) / DG_THREADS, 256 * 8), DG_THREADS, s, rp);
        CK(read_u64(c->dg_cnt.p, R.scnt.data(), world));
        for (int d = 0; d < world; ++d) R.soff[d + 1] = R.soff[d] + R.scnt[d];
        HIPCK(hipMemcpy(c->dg_cnt.p + DG_MAX_WORLD, R.soff.data(), world * sizeof(uint64_t), hipMemcpyHostToDevice));
        if (n) CDBG_LAUNCH(k_route_place, std::min<uint64_t>((n + DG_THREADS * DG_ITEMS - 1) / (DG_THREADS * DG_ITEMS), 256 * 8), DG_THREADS, s, rp);
        return CDBG_OK;
    };
    const int rc_local = local();
    const std::string my_err = rc_local != CDBG_OK ? g_err : std::string();
    if (rc_local != CDBG_OK) { R.scnt.assign(world, 0); R.soff.assign(world + 1, 0); }      // (a failed rank sends nothing)
    R.n_send = R.soff[world];
    {
        const int row = world + n_extra + 1;             // counts, the caller's words, this rank's status
        std::vector<uint64_t> mine(row), got((size_t)row * world);
        for (int d = 0; d < world; ++d) mine[d] = R.scnt[d];
        for (int i = 0; i < n_extra; ++i) mine[world + i] = extra[i];
        mine[row - 1] = (uint64_t)(int64_t)rc_local;
        if (c->tr.all_gather_u64(c->tr.user, mine.data(), got.data(), row) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_u64 failed");
        if (rc_local != CDBG_OK) { g_err = my_err; return rc_local; }
        for (int r = 0; r < world; ++r) {
            if (got[(size_t)r * row + row - 1]) return fail(CDBG_E_INTERNAL, "sharded glue: rank %d reported error %lld; all ranks stop", r, (long long)(int64_t)got[(size_t)r * row + row - 1]);
            for (int d = 0; d < world; ++d) R.all[(size_t)r * world + d] = got[(size_t)r * row + d];
            if (extra_all) for (int i = 0; i < n_extra; ++i) (*extra_all)[(size_t)r * n_extra + i] = got[(size_t)r * row + world + i];
        }
    }
    R.n_all = 0;
    for (int r = 0; r < world; ++r) { R.rcnt[r] = R.all[(size_t)r * world + me]; for (int d = 0; d < world; ++d) R.n_all += R.all[(size_t)r * world + d]; }
    for (int r = 0; r < world; ++r) R.roff[r + 1] = R.roff[r] + R.rcnt[r];
    R.n_recv = R.roff[world];
    return CDBG_OK;
}
// all-to-all-v of fixed-size items laid out by dg_route (reverse = true: the replies travel back along the same blocks)
int dg_a2a(cdbg_ctx* c, const void* send, void* recv, const DgRoute& R, uint64_t item, bool reverse = false) {
    const int world = c->prm.world_size, me = c->prm.rank;
    std::vector<uint64_t> so(world), sc(world), ro(world), rc(world);
    for (int r = 0; r < world; ++r) {
        const uint64_t* sof = reverse ? R.roff.data() : R.soff.data(); const uint64_t* scn = reverse ? R.rcnt.data() : R.scnt.data();
        const uint64_t* rof = reverse ? R.soff.data() : R.roff.data(); const uint64_t* rcn = reverse ? R.scnt.data() : R.rcnt.data();
        so[r] = sof[r] * item; sc[r] = scn[r] * item; ro[r] = rof[r] * item; rc[r] = rcn[r] * item;
        if (r != me) c->comm_bytes += sc[r] + rc[r];
    }
    if (!c->tr_ordered) HIPCK(hipStreamSynchronize(c->stream));   // (a caller-supplied transport reads the buffers from the host side)
    if (c->tr.all_to_all_v(c->tr.user, send, so.data(), sc.data(), recv, ro.data(), rc.data()) != 0) return fail(CDBG_E_INTERNAL, "transport all_to_all_v failed");
    return CDBG_OK;
}
// all-to-all-v of variable-size blocks: send_bytes[d] at send_off[d]; the receive sizes are exchanged first
int dg_a2a_blocks(cdbg_ctx* c, const void* send, const std::vector<uint64_t>& send_off, const std::vector<uint64_t>& send_bytes,
                  std::vector<uint64_t>& recv_off, std::vector<uint64_t>& recv_bytes, DBuf<uint8_t>& recv, uint64_t align) {
    const int world = c->prm.world_size, me = c->prm.rank;
    std::vector<uint64_t> all((size_t)world * world);
    if (c->tr.all_gather_u64(c->tr.user, send_bytes.data(), all.data(), world) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_u64 failed");
    recv_off.assign(world + 1, 0); recv_bytes.assign(world, 0);
    for (int r = 0; r < world; ++r) { recv_bytes[r] = all[(size_t)r * world + me]; recv_off[r + 1] = recv_off[r] + (recv_bytes[r] + align - 1) / align * align; if (r != me) c->comm_bytes += send_bytes[r] + recv_bytes[r]; }
    CK(recv.alloc(recv_off[world] + align, false));
    if (!c->tr_ordered) HIPCK(hipStreamSynchronize(c->stream));   // (a caller-supplied transport reads the buffers from the host side)
    if (c->tr.all_to_all_v(c->tr.user, send, send_off.data(), send_bytes.data(), recv.p, recv_off.data(), recv_bytes.data()) != 0) return fail(CDBG_E_INTERNAL, "transport all_to_all_v failed");
    return CDBG_OK;
}

// Closed chains in the distributed ranking (k_dglue.h): every rank's unfinished states, all-gathered; the same cut on every rank.
// DG_FALLBACK when they are too many to handle this way (the replicated exchange then cuts them).
constexpr uint64_t DG_CYCLE_MAX = 1ull << 22;
int dg_cut_cycles(cdbg_ctx* c, const DRankParams& dp, uint64_t* n_cycles) {
    const int world = c->prm.world_size, me = c->prm.rank; hipStream_t s = c->stream;
    const uint32_t NSl = dp.n_local;
    DBuf<uint2> mine, all; DBuf<uint64_t> cur; DBuf<uint32_t> cut;
    // (a rank-local failure -- an allocation, a state that is not where it should be -- must not leave the other ranks waiting in the
    //  next collective: every local status goes through agree() before the ranks move on)
    uint64_t n_mine = 0;
    auto collect = [&]() -> int {
        CK(cur.alloc(1, true));
        CK(mine.alloc(std::min<uint64_t>(NSl, DG_CYCLE_MAX) + 1, false));
        DrOpenParams op{ NSl, dp.base, dp.st, dp.link, mine.p, cur.p, std::min<uint64_t>(NSl, DG_CYCLE_MAX) };
        if (NSl) CDBG_LAUNCH(k_dr_collect_open, (NSl + 255) / 256, 256, s, op);
        return read_u64(cur.p, &n_mine);
    };
    CK(agree(c, collect(), "sharded glue: closed chains (collect)"));
    std::vector<uint64_t> cnt(world);
    if (c->tr.all_gather_u64(c->tr.user, &n_mine, cnt.data(), 1) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_u64 failed");
    uint64_t total = 0; std::vector<uint64_t> roff(world), rcnt(world);
    for (int r = 0; r < world; ++r) { roff[r] = total * sizeof(uint2); rcnt[r] = cnt[r] * sizeof(uint2); total += cnt[r]; }
    if (total > DG_CYCLE_MAX) return DG_FALLBACK;        // (every rank sees the same total)
    CK(agree(c, all.alloc(total + 1, false), "sharded glue: closed chains (gather buffer)"));
    if (!c->tr_ordered) HIPCK(hipStreamSynchronize(s));
    if (c->tr.all_gather_v(c->tr.user, mine.p, n_mine * sizeof(uint2), all.p, roff.data(), rcnt.data()) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_v failed");
    for (int r = 0; r < world; ++r) if (r != me) c->comm_bytes += n_mine * sizeof(uint2) + rcnt[r];
    std::vector<uint2> h(total);
    std::vector<uint32_t> cuts; uint64_t ncyc = 0;
    auto analyse = [&]() -> int {
    HIPCK(hipMemcpy(h.data(), all.p, total * sizeof(uint2), hipMemcpyDeviceToHost));
    // state -> successor; every state of a closed chain is in the list (both directions of every piece of it)
    std::sort(h.begin(), h.end(), [](const uint2& a, const uint2& b) { return a.x < b.x; });
    auto next_of = [&](uint32_t e, uint32_t& nx) -> bool {
        size_t lo = 0, hi = h.size();
        while (lo < hi) { const size_t m = (lo + hi) / 2; if (h[m].x < e) lo = m + 1; else hi = m; }
        if (lo == h.size() || h[lo].x != e) return false;
        nx = h[lo].y; return true;
    };
    std::vector<uint8_t> seen(h.size(), 0);
    std::unordered_set<uint32_t> cut_pieces;             // (the two directions of a cycle elect the same piece: cut once)
    for (size_t i = 0; i < h.size(); ++i) {
        if (seen[i]) continue;
        uint32_t e = h[i].x, pmin = e >> 1; size_t steps = 0;
        for (;;) {                                       // walk the cycle of states that starts at h[i]
            size_t lo = 0, hi = h.size();
            while (lo < hi) { const size_t m = (lo + hi) / 2; if (h[m].x < e) lo = m + 1; else hi = m; }
            if (lo == h.size() || h[lo].x != e) return fail(CDBG_E_INTERNAL, "sharded glue: state %u of a closed chain is missing from the gathered list", e);
            if (seen[lo]) break;
            seen[lo] = 1; pmin = std::min(pmin, e >> 1);
            e = h[lo].y;
            if (e == NONE32 || ++steps > h.size()) return fail(CDBG_E_INTERNAL, "sharded glue: an unfinished state is not on a closed chain");
        }
        // the two directions of a piece cycle elect the same piece; the one that passes its left end as an EXIT names the junction
        uint32_t partner = 0;
        if (!next_of(2u * pmin + 1u, partner)) return fail(CDBG_E_INTERNAL, "sharded glue: closed chain without its reverse direction");
        if (cut_pieces.insert(pmin).second) { cuts.push_back(2u * pmin); cuts.push_back(partner); ++ncyc; }
    }
    return CDBG_OK;
    };
    CK(agree(c, analyse(), "sharded glue: closed chains (analysis)"));
    if (!cuts.empty()) {
        CK(cut.alloc(cuts.size(), false));
        HIPCK(hipMemcpy(cut.p, cuts.data(), cuts.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        DrCutParams cp{ cut.p, (uint32_t)cuts.size(), dp.base, NSl, const_cast<uint32_t*>(dp.link) };
        CDBG_LAUNCH(k_dr_cut, ((uint32_t)cuts.size() + 255) / 256, 256, s, cp);
        HIPCK(hipStreamSynchronize(s));
    }
    *n_cycles = ncyc;
    return CDBG_OK;
}

template <int W>
int glue_sharded(cdbg_ctx* c) {
    const int world = c->prm.world_size, me = c->prm.rank, k = c->k;
    hipStream_t s = c->stream;
    if (world > DG_MAX_WORLD) return fail(CDBG_E_PARAM, "sharded glue supports up to %d ranks", DG_MAX_WORLD);
    Timer t; CK(t.start(s));
    const uint64_t NP = c->n_pieces;
    auto grid = [](uint64_t n) { return (uint32_t)std::min<uint64_t>(std::max<uint64_t>((n + 255) / 256, 1), 256 * 16); };
    DgOwners own{}; own.world = world;
    const uint32_t NSl = (uint32_t)(2 * NP);
    uint32_t end_base = 0;
    HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
    HIPCK(hipMemsetAsync(c->derr.p, 0, 4 * sizeof(uint32_t), s));
    // ---- 1a. junction records to their key owners (the piece id ranges of the ranks ride in the routing's count exchange) ----
    DgRoute R;
    {
        const uint64_t n = c->n_glog;
        CK(c->dg_dest.alloc(std::max<uint64_t>(std::max<uint64_t>(n, NSl), NP) + 1, false));
        LogRouteParams lp{ c->glog_keys.p, c->glog_tag.p, n, world, 0, c->dg_dest.p, nullptr, nullptr };
        if (n) CDBG_LAUNCH((k_log_dest<W>), grid(n), 256, s, lp);
        std::vector<uint64_t> all(world);
        CK(dg_route(c, n, R, &NP, 1, &all));
        uint64_t tot = 0; for (int r = 0; r < world; ++r) { own.b[r] = (uint32_t)tot; tot += all[r]; }
        if (2 * tot >= 0x7FFFFFF0ULL) return fail(CDBG_E_INTERNAL, "too many pieces for 31-bit end ids (%llu): use more partitions per GPU or fewer reads", (unsigned long long)tot);   // (every rank sees the same total)
        own.b[world] = (uint32_t)tot; c->piece_lo = own.b[me]; c->piece_hi = own.b[me + 1];
        end_base = 2u * own.b[me]; lp.end_base = end_base;
        CK(c->dg_wire_s.alloc(R.n_send * (W + 1) + 1, false)); CK(c->dg_wire_r.alloc(R.n_recv * (W + 1) + 1, false));
        lp.pos = c->dg_pos.p; lp.wire = c->dg_wire_s.p;
        if (n) CDBG_LAUNCH((k_log_write<W>), grid(n), 256, s, lp);
        CK(dg_a2a(c, c->dg_wire_s.p, c->dg_wire_r.p, R, (uint64_t)(W + 1) * 8));
    }
    // ---- join this rank's keys; joined pairs to the end owners ----
    uint64_t n_pairs = 0; uint32_t join_err = 0;
    {
        const uint64_t n = R.n_recv;
        int log_jb = 0; while (((uint64_t)(JB_CAP / 2) << log_jb) < n && log_jb < 26) ++log_jb;
        const uint64_t JB = 1ull << log_jb;
        CK(c->jfill.alloc(JB, false)); CK(c->jrecs.alloc(JB * JB_CAP * (W + 1), false));
        HIPCK(hipMemsetAsync(c->jfill.p, 0, JB * sizeof(uint32_t), s));
        HIPCK(hipMemsetAsync(c->cursors.p + 7, 0, sizeof(uint64_t), s));
        WireScatterParams wp{ c->dg_wire_r.p, n, log_jb, c->jfill.p, c->jrecs.p, c->derr.p };
        if (n) CDBG_LAUNCH((k_join_scatter_wire<W>), grid(n), 256, s, wp);
        // (pair list: <= n entries used, in per-wave chunks whose tails stay unused -- pre-filled with the 'no pair' marker)
        // (+ n / 8: a wave also abandons the rest of its current chunk whenever a bucket's pairs do not fit into it)
        const uint64_t pair_cap = n + n / 8 + 2 + (uint64_t)JB_PAIR_CHUNK * std::min<uint64_t>((JB + 3) / 4, 256 * 16) * (JB_THREADS / 64);
        CK(c->dg_pairs.alloc(pair_cap, false));
        HIPCK(hipMemsetAsync(c->dg_pairs.p, 0xFF, pair_cap * sizeof(uint2), s));
        JoinBucketParams bp{ c->jfill.p, c->jrecs.p, (uint32_t)JB, nullptr, c->dstats.p, c->dg_pairs.p, c->cursors.p + 7, pair_cap, c->derr.p };
        CDBG_LAUNCH((k_join_bucket<W>), std::min<uint64_t>((JB + 3) / 4, 256 * 16), JB_THREADS, s, bp);
        CK(read_u32(c->derr.p, &join_err));
        CK(read_u64(c->cursors.p + 7, &n_pairs));
        if (join_err || n_pairs > pair_cap) n_pairs = 0;     // (nothing of a failed join travels; the ranks agree on what happens below)
        uint64_t gs = 0; CK(read_u64(c->dstats.p, &gs)); c->n_join_local = gs;
    }
    {
        PairRouteParams pp{ c->dg_pairs.p, n_pairs, own, c->dg_dest.p, nullptr, nullptr };
        CK(c->dg_dest.alloc(std::max<uint64_t>(std::max<uint64_t>(n_pairs, NSl), NP) + 1, false)); pp.dest = c->dg_dest.p;
        if (n_pairs) CDBG_LAUNCH(k_pair_dest, grid(n_pairs), 256, s, pp);
        // the status of every rank's join rides in the routing's count exchange: a bucket overflow (8) stops all ranks together, a
        // pair list that did not fit (9: nearly every record joined and the chunk tails ate the headroom) sends all of them to
        // the replicated exchange
        const uint64_t stw = join_err; std::vector<uint64_t> sts(world);
        CK(dg_route(c, n_pairs, R, &stw, 1, &sts));
        bool any9 = false;
        for (int r = 0; r < world; ++r) {
            if (sts[r] == 9) any9 = true;
            else if (sts[r]) return fail(CDBG_E_INTERNAL, "sharded junction join: rank %d reported device error %llu (8 bucket overflow); all ranks stop", r, (unsigned long long)sts[r]);
        }
        if (any9) { HIPCK(hipMemsetAsync(c->derr.p, 0, 4 * sizeof(uint32_t), s)); float ms = 0; CK(t.stop(&ms)); c->st.ms_exchange += ms; return DG_FALLBACK; }
        CK(c->dg_pair_s.alloc(R.n_send + 1, false)); CK(c->dg_pair_r.alloc(R.n_recv + 1, false));
        pp.pos = c->dg_pos.p; pp.wire = c->dg_pair_s.p;
        if (n_pairs) CDBG_LAUNCH(k_pair_write, grid(n_pairs), 256, s, pp);
        CK(dg_a2a(c, c->dg_pair_s.p, c->dg_pair_r.p, R, sizeof(uint2)));
        CK(c->link.alloc(NSl, false));
        HIPCK(hipMemsetAsync(c->link.p, 0xFF, (size_t)std::max<uint32_t>(NSl, 1) * sizeof(uint32_t), s));
        PairApplyParams ap{ c->dg_pair_r.p, R.n_recv, end_base, c->link.p };
        if (R.n_recv) CDBG_LAUNCH(k_pair_apply, grid(R.n_recv), 256, s, ap);
    }
    // ---- 2. distributed list ranking ----
    CK(c->rank_a.alloc(NSl, false)); CK(c->rank_b.alloc(NSl, false));
    uint2* const st = reinterpret_cast<uint2*>(c->rank_b.p);
    DRankParams dp{}; dp.n_local = NSl; dp.base = end_base; dp.own = own; dp.me = me; dp.link = c->link.p; dp.piece_n = c->piece_n.p; dp.st = st; dp.dest = c->dg_dest.p;
    const uint32_t gridS = std::max<uint32_t>((NSl + 255) / 256, 1);
    if (NSl) CDBG_LAUNCH(k_dr_init, gridS, 256, s, dp);
    {
        uint64_t total_states = 2ull * own.b[world];
        int max_rounds = 4; while ((1ull << (max_rounds - 3)) < total_states) ++max_rounds;
        bool done = false;
        uint64_t prev_open = ~0ull; int cuts_made = 0; uint64_t cycles_cut = 0;
        const int rounds_per_ranking = max_rounds;
        for (int round = 0; round < max_rounds; ++round) {
            if (NSl) CDBG_LAUNCH(k_dr_jump, gridS, 256, s, dp);
            CK(dg_route(c, NSl, R));
            if (R.n_all == 0) { done = true; break; }                       // no rank has an unfinished state
            if (R.n_all == prev_open && cuts_made < 2) {
                // a round that finished nothing: only closed chains are left (every round finishes the states within reach of a
                // tail).  Cut them where they are (k_dglue.h) and rank again -- every rank takes this branch together.
                uint64_t ncyc = 0;
                const int rc = dg_cut_cycles(c, dp, &ncyc);
                if (rc == DG_FALLBACK) break;
                CK(rc);
                cycles_cut += ncyc; ++cuts_made;
                if (NSl) CDBG_LAUNCH(k_dr_init, gridS, 256, s, dp);
                prev_open = ~0ull;
                max_rounds = round + 1 + rounds_per_ranking;                // (the ranking starts over: a full budget of rounds again)
                continue;
            }
            prev_open = R.n_all;
            ++c->st.n_glue_rounds;
            CK(c->dg_qs.alloc(R.n_send + 1, false)); CK(c->dg_qsrc.alloc(R.n_send + 1, false)); CK(c->dg_qr.alloc(R.n_recv + 1, false));
            CK(c->dg_rs.alloc(R.n_recv + 1, false)); CK(c->dg_rr.alloc(R.n_send + 1, false));
            dp.pos = c->dg_pos.p; dp.q_send = c->dg_qs.p; dp.q_src = c->dg_qsrc.p; dp.q_recv = c->dg_qr.p; dp.r_send = c->dg_rs.p; dp.n_recv = R.n_recv;
            dp.r_recv = c->dg_rr.p; dp.n_sent = R.n_send;
            if (NSl) CDBG_LAUNCH(k_dr_query, gridS, 256, s, dp);
            CK(dg_a2a(c, c->dg_qs.p, c->dg_qr.p, R, sizeof(uint32_t)));
            if (R.n_recv) CDBG_LAUNCH(k_dr_reply, grid(R.n_recv), 256, s, dp);
            CK(dg_a2a(c, c->dg_rs.p, c->dg_rr.p, R, sizeof(uint2), true));
            if (R.n_send) CDBG_LAUNCH(k_dr_apply, grid(R.n_send), 256, s, dp);
        }
        if (!done) { float ms = 0; CK(t.stop(&ms)); c->st.ms_exchange += ms; c->st.n_glue_rounds = 0; return DG_FALLBACK; }   // closed chains across ranks
        c->st.n_cycles += cycles_cut;                                       // (only now: the fallback counts the chains it cuts itself)
    }
    // ---- 3. heads of this rank, then every piece to the owner of its head ----
    uint64_t hm[2] = {0, 0};
    {
        HIPCK(hipMemsetAsync(c->dstats.p + 8, 0, 2 * sizeof(uint64_t), s));
        HeadMeasureParams mp{ NSl, k, c->link.p, st, c->dstats.p + 8 };
        if (NSl) CDBG_LAUNCH(k_heads_measure, grid(NSl), 256, s, mp);
        CK(read_u64(c->dstats.p + 8, hm, 2));
    }
    const uint64_t ucap = std::max<uint64_t>(hm[0], 1), ocap = std::max<uint64_t>(hm[1], 1);
    CK(c->unitig_off.alloc(ucap, false)); CK(c->unitig_len.alloc(ucap, false)); CK(c->unitig_kc.alloc(ucap, false));
    CK(c->unitig_bases.alloc(ocap + 64, false));
    if (c->prm.all_abundance_counts) CK(c->unitig_ab.alloc(ocap + 64, false));
    HIPCK(hipMemsetAsync(c->cursors.p + 2, 0, 2 * sizeof(uint64_t), s));
    HeadParams hp{};
    hp.n_states = NSl; hp.k = k; hp.link = c->link.p; hp.st = st; hp.hinfo = c->rank_a.p;
    hp.unitig_off = c->unitig_off.p; hp.unitig_len = c->unitig_len.p; hp.unitig_kc = c->unitig_kc.p;
    hp.unitig_cap = ucap; hp.out_cap = ocap; hp.n_unitigs = c->cursors.p + 2; hp.out_cursor = c->cursors.p + 3; hp.error = c->derr.p; hp.own_lo = 0; hp.own_hi = NSl;
    if (NSl) CDBG_LAUNCH(k_unitig_heads, (NSl + HEADS_PER_WG - 1) / HEADS_PER_WG, GLUE_THREADS, s, hp);
    uint64_t NR = 0; int rc_late = CDBG_OK;              // a rank-local failure behind the last exchange: agreed on before the ranks part
    {
        PieceRouteParams pr{}; pr.n_pieces = (uint32_t)NP; pr.k = k; pr.own = own; pr.st = st; pr.piece_n = c->piece_n.p; pr.piece_kc = c->piece_kc.p; pr.piece_boff = c->piece_boff.p; pr.dest = c->dg_dest.p;
        const uint32_t gridP = std::max<uint32_t>((uint32_t)((NP + 255) / 256), 1);
        if (NP) CDBG_LAUNCH(k_piece_dest, gridP, 256, s, pr);
        CK(dg_route(c, NP, R));
        const uint64_t ns = R.n_send; NR = R.n_recv;
        CK(c->dg_meta_s.alloc(3 * ns + 3, false)); CK(c->dg_lens.alloc(ns + 1, false)); CK(c->dg_boff.alloc(ns + 1, false)); CK(c->dg_alen.alloc(ns + 1, false));
        pr.pos = c->dg_pos.p; pr.meta = c->dg_meta_s.p; pr.lens = c->dg_lens.p; pr.boff = c->dg_boff.p; pr.alen = c->dg_alen.p;
        if (NP) CDBG_LAUNCH(k_piece_write, gridP, 256, s, pr);
        CK(c->dg_meta_r.alloc(3 * NR + 3, false));
        CK(dg_a2a(c, c->dg_meta_s.p, c->dg_meta_r.p, R, 24));
        // the bases of every destination's pieces as one gap-free stream, 2 bits per base (whole 64-base chunks)
        CK(c->dg_uoff.alloc(ns + world + 1, false));
        std::vector<uint64_t> dbase(world + 1, 0), dtot(world, 0), sbytes(world), soffb(world);
        for (int d = 0; d < world; ++d) {
            CK(exscan_u32(c, c->dg_lens.p + R.soff[d], c->dg_uoff.p + R.soff[d] + d, R.scnt[d]));
            CK(read_u64(c->dg_uoff.p + R.soff[d] + d + R.scnt[d], &dtot[d]));
            dbase[d + 1] = dbase[d] + (dtot[d] + 63) / 64 * 64;
        }
        CK(c->dg_dense.alloc(dbase[world] + 64, false)); CK(c->dg_packed.alloc(dbase[world] / 4 + 16, false));
        if (dbase[world]) HIPCK(hipMemsetAsync(c->dg_dense.p, 'A', dbase[world], s));
        for (int d = 0; d < world; ++d) {
            if (!R.scnt[d]) { sbytes[d] = 0; soffb[d] = dbase[d] / 4; continue; }
            SqueezeParams sq{ R.scnt[d], c->dg_lens.p + R.soff[d], c->dg_uoff.p + R.soff[d] + d, c->dg_boff.p + R.soff[d], c->piece_bases.p, c->dg_dense.p + dbase[d] };
            CDBG_LAUNCH(k_squeeze_bases, (R.scnt[d] + 255) / 256, 256, s, sq);
            sbytes[d] = (dtot[d] + 63) / 64 * 16; soffb[d] = dbase[d] / 4;
        }
        const uint64_t chunks = dbase[world] / 64;
        if (chunks) { StreamPackParams pp{ chunks, c->dg_dense.p, c->dg_packed.p, dbase[world] }; CDBG_LAUNCH(k_pack_stream, (chunks + 255) / 256, 256, s, pp); }
        std::vector<uint64_t> roffb, rbytes;
        CK(dg_a2a_blocks(c, c->dg_packed.p, soffb, sbytes, roffb, rbytes, c->dg_rpacked, 16));
        // receiver: metas -> piece arrays, packed streams -> ASCII
        CK(c->dg_rn.alloc(NR + 1, false)); CK(c->dg_rkc.alloc(NR + 1, false)); CK(c->dg_rst.alloc(NR + 1, false)); CK(c->dg_rlens.alloc(NR + 1, false)); CK(c->dg_rboff.alloc(NR + 2, false));
        PieceRecvParams rv{ NR, k, end_base, c->dg_meta_r.p, c->dg_rn.p, c->dg_rkc.p, c->dg_rst.p, c->dg_rlens.p };
        if (NR) CDBG_LAUNCH(k_piece_recv, grid(NR), 256, s, rv);
        std::vector<uint64_t> rbase(world + 1, 0), rtot(world, 0);
        for (int r = 0; r < world; ++r) {
            CK(exscan_u32(c, c->dg_rlens.p + R.roff[r], c->dg_rboff.p + R.roff[r], R.rcnt[r]));
            CK(read_u64(c->dg_rboff.p + R.roff[r] + R.rcnt[r], &rtot[r]));
            if ((rtot[r] + 63) / 64 * 16 != rbytes[r] && rc_late == CDBG_OK) rc_late = fail(CDBG_E_INTERNAL, "sharded glue: rank %d sent %llu packed bytes for %llu bases", r, (unsigned long long)rbytes[r], (unsigned long long)rtot[r]);
            rbase[r + 1] = rbase[r] + (rtot[r] + 63) / 64 * 64;
        }
        // (an inconsistent stream: nothing of it is unpacked; the ranks learn of it together -- the abundance exchange's status word, or the stage's last one)
        if (rc_late == CDBG_OK) { const int rc_a = c->dg_rdense.alloc(rbase[world] + 64, false); if (rc_a != CDBG_OK) rc_late = rc_a; }
        for (int r = 0; r < world && rc_late == CDBG_OK; ++r) {
            if (!R.rcnt[r]) continue;
            if (rbase[r]) CDBG_LAUNCH(k_add_u64, (R.rcnt[r] + 255) / 256, 256, s, c->dg_rboff.p + R.roff[r], R.rcnt[r], rbase[r]);
            const uint64_t ch = (rtot[r] + 63) / 64;
            if (ch) { StreamUnpackParams up{ ch, c->dg_rpacked.p + roffb[r], c->dg_rdense.p + rbase[r], rtot[r] }; CDBG_LAUNCH(k_unpack_stream, (ch + 255) / 256, 256, s, up); }
        }
        // -all-abundance-counts: one u32 per k-mer of every piece, same routing
        if (c->prm.all_abundance_counts) {
            CK(agree(c, rc_late, "glue: pieces"));
            CK(c->dg_aoff.alloc(ns + world + 1, false));
            std::vector<uint64_t> abase(world + 1, 0), atot(world, 0), ab_sb(world), ab_so(world);
            for (int d = 0; d < world; ++d) {
                CK(exscan_u32(c, c->dg_alen.p + R.soff[d], c->dg_aoff.p + R.soff[d] + d, R.scnt[d]));
                CK(read_u64(c->dg_aoff.p + R.soff[d] + d + R.scnt[d], &atot[d]));
                abase[d + 1] = abase[d] + (atot[d] + 3) / 4 * 4;
            }
            CK(c->dg_ab_s.alloc(abase[world] + 4, false));
            for (int d = 0; d < world; ++d) {
                ab_sb[d] = atot[d] * 4; ab_so[d] = abase[d] * 4;
                if (!R.scnt[d]) continue;
                AbStreamParams ap{ R.scnt[d], k, 0, c->dg_alen.p + R.soff[d], c->dg_aoff.p + R.soff[d] + d, nullptr, c->dg_boff.p + R.soff[d], 0, c->piece_ab.p, c->dg_ab_s.p + abase[d] };
                CDBG_LAUNCH(k_ab_stream, (R.scnt[d] + 255) / 256, 256, s, ap);
            }
            std::vector<uint64_t> ab_ro, ab_rb; DBuf<uint8_t>& rbuf = c->xsend;
            CK(dg_a2a_blocks(c, c->dg_ab_s.p, ab_so, ab_sb, ab_ro, ab_rb, rbuf, 16));
            CK(c->dg_rab.alloc(rbase[world] + 64, false)); CK(c->dg_raoff.alloc(NR + 2, false));
            for (int r = 0; r < world; ++r) {
                if (!R.rcnt[r]) continue;
                CK(exscan_u32(c, c->dg_rn.p + R.roff[r], c->dg_raoff.p + R.roff[r], R.rcnt[r]));
                uint64_t tot = 0; CK(read_u64(c->dg_raoff.p + R.roff[r] + R.rcnt[r], &tot));
                if (tot * 4 != ab_rb[r]) { rc_late = fail(CDBG_E_INTERNAL, "sharded glue: abundance stream of rank %d does not match its pieces", r); break; }
                AbStreamParams ap{ R.rcnt[r], k, 1, c->dg_rn.p + R.roff[r], c->dg_raoff.p + R.roff[r], nullptr, c->dg_rboff.p + R.roff[r], 0, c->dg_rab.p, reinterpret_cast<uint32_t*>(rbuf.p + ab_ro[r]) };
                CDBG_LAUNCH(k_ab_stream, (R.rcnt[r] + 255) / 256, 256, s, ap);
                HIPCK(hipStreamSynchronize(s));                             // (the next source's prefix sums reuse dg_raoff's boundary word)
            }
        }
    }
    // ---- emit what this rank owns ----
    if (NR && rc_late == CDBG_OK) {
        EmitParams ep{};
        ep.n_pieces = (uint32_t)NR; ep.k = k; ep.st = reinterpret_cast<const uint2*>(c->dg_rst.p); ep.hinfo = c->rank_a.p;
        ep.piece_n = c->dg_rn.p; ep.piece_kc = c->dg_rkc.p; ep.piece_boff = c->dg_rboff.p; ep.piece_bases = c->dg_rdense.p;
        ep.unitig_kc = c->unitig_kc.p; ep.out = c->unitig_bases.p;
        ep.piece_ab = c->prm.all_abundance_counts ? c->dg_rab.p : nullptr; ep.unitig_ab = c->unitig_ab.p;
        CDBG_LAUNCH(k_emit, (uint32_t)((NR + GLUE_THREADS - 1) / GLUE_THREADS), GLUE_THREADS, s, ep);
    }
    { uint64_t cur[2]; CK(read_u64(c->cursors.p + 2, cur, 2)); c->n_unitigs = cur[0]; c->unitig_total = cur[1]; }
    CK(pack_unitigs(c));
    float ms = 0; CK(t.stop(&ms));
    c->st.ms_glue = ms;
    CK(agree(c, rc_late != CDBG_OK ? rc_late : check_device_error(c, "sharded glue"), "glue: emit"));
    c->joined = false; c->xchg_done = true;
    c->st.n_glue_joined = c->n_join_local; c->st.n_unitigs = c->n_unitigs; c->st.unitig_bases = c->unitig_total;
    c->st.ms_total += c->st.ms_glue;
    c->stage = 3;
    return CDBG_OK;
}

}  // namespace
