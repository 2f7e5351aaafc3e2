// host_glue.h -- libcdbg.so, host side of stage 3 and of what follows it: junction join, list
// ranking, emission; links (row a10 / f1); the device-side check of the unitig definition.  Included by cdbg_impl.cpp only.
#pragma once

namespace {

// Glue, first half: hash-join the piece ends on their junction (k-1)-mers -> link[end] = partner end.
// sharded (multi-GPU, after xchg_*): this rank joins only the junctions whose key hash selects it --
// 1/world of the device atomics -- and leaves the other ends at NONE; the caller combines the link arrays of all
// ranks with an element-wise MAX all-reduce (every end is set by exactly one rank) before cdbg_glue.
template <int W>
int glue_join_impl(cdbg_ctx* c, bool sharded) {
    if (c->stage < 2) return fail(CDBG_E_STATE, "cdbg_glue before cdbg_compact");
    hipStream_t s = c->stream;
    const uint64_t NP = c->n_pieces;
    if (2 * NP >= 0x7FFFFFF0ULL) return fail(CDBG_E_INTERNAL, "too many pieces for 31-bit end ids (%llu)", (unsigned long long)NP);
    const uint32_t NS = (uint32_t)(2 * NP);
    Timer t; CK(t.start(s));
    CK(c->link.alloc(NS, false));
    HIPCK(hipMemsetAsync(c->link.p, 0xFF, (size_t)std::max<uint32_t>(NS, 1) * sizeof(uint32_t), s));
    HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
    HIPCK(hipMemsetAsync(c->derr.p, 0, 4 * sizeof(uint32_t), s));
    const uint32_t world = sharded ? (uint32_t)c->prm.world_size : 1u;
    const uint64_t n_mine = c->n_glog / world + (world > 1 ? (c->n_glog >> 6) + 1024 : 0);      // records this rank joins (estimate when sharded)
    bool bucketed = (c->knobs.get("CDBG_GLUE_TABLE") == nullptr || c->direct_join) && c->n_glog > 0;
    if (bucketed && c->direct_join) {                        // the buckets were filled by the compaction kernels
        const uint64_t JB = 1ull << c->join_log_jb;
        JoinBucketParams bp{ c->jfill.p, c->jrecs.p, (uint32_t)JB, c->link.p, c->dstats.p, nullptr, nullptr, 0, nullptr };
        CDBG_LAUNCH((k_join_bucket<W>), std::min<uint64_t>((JB + 3) / 4, 256 * 16), JB_THREADS, s, bp);
    } else if (bucketed) {
        // bucketed join (k_glue.h): scatter the log into buckets of ~JB_CAP / 2 records, one wave joins a bucket in LDS
        int log_jb = 0; while (((uint64_t)(JB_CAP / 2) << log_jb) < n_mine && log_jb < 26) ++log_jb;
        const uint64_t JB = 1ull << log_jb;
        CK(c->jfill.alloc(JB, false)); CK(c->jrecs.alloc(JB * JB_CAP * (W + 1), false));
        HIPCK(hipMemsetAsync(c->jfill.p, 0, JB * sizeof(uint32_t), s));
        JoinScatterParams sp{ c->glog_keys.p, c->glog_tag.p, c->n_glog, log_jb, c->jfill.p, c->jrecs.p, c->derr.p,
                              world - 1, world > 1 ? (uint32_t)c->prm.rank : 0u };
        CDBG_LAUNCH((k_join_scatter<W>), std::min<uint64_t>((c->n_glog + GLUE_THREADS - 1) / GLUE_THREADS, 1u << 16), GLUE_THREADS, s, sp);
        JoinBucketParams bp{ c->jfill.p, c->jrecs.p, (uint32_t)JB, c->link.p, c->dstats.p, nullptr, nullptr, 0, nullptr };
        CDBG_LAUNCH((k_join_bucket<W>), std::min<uint64_t>((JB + 3) / 4, 256 * 16), JB_THREADS, s, bp);
        HIPCK(hipStreamSynchronize(s));
        uint32_t e = 0; CK(read_u32(c->derr.p, &e));
        if (e == 8) {                                        // a bucket overflowed (cannot happen with a sound hash): global table instead
            bucketed = false;
            HIPCK(hipMemsetAsync(c->link.p, 0xFF, (size_t)std::max<uint32_t>(NS, 1) * sizeof(uint32_t), s));
            HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
            HIPCK(hipMemsetAsync(c->derr.p, 0, 4 * sizeof(uint32_t), s));
        }
    }
    if (!bucketed && c->n_glog) {
        // fallback: one junction table in HBM (at most one junction per glue record)
        CK(glue_table_slots(n_mine + n_mine / 4 + 1024, &c->glue_cap));
        CK(c->glue_keys.alloc((uint64_t)c->glue_cap * W, false));
        CK(c->glue_a.alloc(c->glue_cap, false)); CK(c->glue_b.alloc(c->glue_cap, false)); CK(c->glue_conf.alloc(c->glue_cap, false));
        HIPCK(hipMemsetAsync(c->glue_keys.p, 0xFF, (uint64_t)c->glue_cap * W * sizeof(uint64_t), s));
        HIPCK(hipMemsetAsync(c->glue_a.p, 0, (uint64_t)c->glue_cap * sizeof(uint32_t), s));
        HIPCK(hipMemsetAsync(c->glue_b.p, 0, (uint64_t)c->glue_cap * sizeof(uint32_t), s));
        HIPCK(hipMemsetAsync(c->glue_conf.p, 0, (uint64_t)c->glue_cap * sizeof(uint32_t), s));
        GlueBuildParams bp{ c->glog_keys.p, c->glog_tag.p, c->n_glog, c->glue_keys.p, c->glue_a.p, c->glue_b.p, c->glue_conf.p, c->glue_cap - 1,
                            world - 1, world > 1 ? (uint32_t)c->prm.rank : 0u };
        CDBG_LAUNCH((k_glue_build<W>), std::min<uint64_t>((c->n_glog + GLUE_THREADS - 1) / GLUE_THREADS, MAX_GRID), GLUE_THREADS, s, bp);
        GlueResolveParams gp{};
        gp.keys = c->glue_keys.p; gp.a = c->glue_a.p; gp.b = c->glue_b.p; gp.conf = c->glue_conf.p;
        gp.cap = c->glue_cap; gp.W = W; gp.link = c->link.p; gp.stats = c->dstats.p;
        CDBG_LAUNCH(k_glue_resolve, std::min<uint64_t>((c->glue_cap + GLUE_THREADS - 1) / GLUE_THREADS, GLUE_RESOLVE_GRID), GLUE_THREADS, s, gp);
    }
    float ms = 0; CK(t.stop(&ms));
    CK(check_device_error(c, "glue join"));
    uint64_t gs = 0; CK(read_u64(c->dstats.p, &gs));
    c->n_join_local = gs; c->st.ms_glue = ms; c->joined = true;
    return CDBG_OK;
}


// Glue, second half on one GPU: the chains walked from their heads (k_walk.h).  *done = false: some chain was too long or
// closed -- the caller ranks (link[] is untouched).
int glue_walk(cdbg_ctx* c, bool* done) {
    hipStream_t s = c->stream;
    const uint64_t NP = c->n_pieces, NS = 2 * NP;
    *done = false;
    if (!NP) return CDBG_OK;
    const uint64_t ucap = std::max<uint64_t>(NP, 1);
    const uint64_t ocap = std::max<uint64_t>(c->n_piece_bases, 1);
    CK(c->unitig_off.alloc(ucap, false)); CK(c->unitig_len.alloc(ucap, false)); CK(c->unitig_kc.alloc(ucap, false));
    CK(c->unitig_bases.alloc(ocap + 64, false));             // (+ 64: the 2-bit packing pass reads whole 64-base chunks)
    if (c->prm.all_abundance_counts) CK(c->unitig_ab.alloc(ocap, false));
    CK(c->rank_flag.alloc(4, true));
    CK(c->walk_rec.alloc(NS, false)); CK(c->walk_heads.alloc(NS, false)); CK(c->walk_hlen.alloc(NS, false)); CK(c->walk_hoff.alloc(NS, false));
    HIPCK(hipMemsetAsync(c->cursors.p + 2, 0, 2 * sizeof(uint64_t), s));
    HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
    HIPCK(hipMemsetAsync(c->derr.p, 0, 4 * sizeof(uint32_t), s));
    HIPCK(hipMemsetAsync(c->rank_flag.p, 0, sizeof(uint32_t), s));
    uint32_t max_steps = 4096;
    if (const char* e = c->knobs.get("CDBG_WALK_MAX")) max_steps = (uint32_t)strtoul(e, nullptr, 10);
    // dstats: [4] live pieces, [5] pieces the kept walks visited, [6] head states (stays on the device: the launches cover the upper bound)
    uint64_t* const n_heads = c->dstats.p + 6;
    WalkInitParams ip{ (uint32_t)NP, c->piece_n.p, c->piece_kc.p, c->piece_boff.p, c->link.p, c->walk_rec.p, c->walk_heads.p, n_heads, c->dstats.p + 4 };
    CDBG_LAUNCH(k_walk_init, (NP + WALK_PIECES - 1) / WALK_PIECES, WALK_THREADS, s, ip);
    const uint64_t spans = (NS + WALK_SPAN - 1) / WALK_SPAN, wave_grid = (spans + WALK_THREADS / 64 - 1) / (WALK_THREADS / 64);
    WalkMeasureParams mp{ c->walk_rec.p, c->walk_heads.p, n_heads, c->walk_hlen.p, max_steps, c->rank_flag.p };
    CDBG_LAUNCH(k_walk_measure, wave_grid, WALK_THREADS, s, mp);
    WalkPlaceParams pp{ n_heads, c->walk_hlen.p, c->walk_hoff.p, c->k, c->cursors.p + 2, c->cursors.p + 3, ucap, ocap, c->derr.p };
    CDBG_LAUNCH(k_walk_place, spans, WALK_THREADS, s, pp);
    WalkCopyParams wp{};
    wp.k = c->k; wp.rec = c->walk_rec.p; wp.piece_bases = c->piece_bases.p;
    wp.heads = c->walk_heads.p; wp.n_heads = n_heads; wp.hlen = c->walk_hlen.p; wp.hoff = c->walk_hoff.p;
    wp.unitig_off = c->unitig_off.p; wp.unitig_len = c->unitig_len.p; wp.unitig_kc = c->unitig_kc.p; wp.out = c->unitig_bases.p; wp.visited = c->dstats.p + 5;
    wp.piece_ab = c->prm.all_abundance_counts ? c->piece_ab.p : nullptr; wp.unitig_ab = c->unitig_ab.p;
    CDBG_LAUNCH(k_walk_copy, wave_grid, WALK_THREADS, s, wp);
    HIPCK(hipStreamSynchronize(s));
    uint64_t d[2]; CK(read_u64(c->dstats.p + 4, d, 2));      // live, visited
    uint32_t gave_up = 0; CK(read_u32(c->rank_flag.p, &gave_up));
    if (gave_up || d[1] != d[0]) { c->walk_off = true; return CDBG_OK; }    // a chain beyond max_steps, or closed chains (no head, never visited)
    *done = true;
    return CDBG_OK;
}

template <int W>
int glue_impl(cdbg_ctx* c) {
    if (c->stage < 2) return fail(CDBG_E_STATE, "cdbg_glue before cdbg_compact");
    if ((c->prm.world_size > 1 || c->force_multi) && c->have_tr && !c->xchg_done && !c->joined) {
        // multi-GPU.  Every rank emits its own unitigs (emit_replicated = 0): the sharded glue of k_dglue.h -- every record and
        // every piece travels once.  emit_replicated = 1 (the CLI's rank 0 writes one file and needs the whole graph for the
        // links), or closed chains across ranks: the replicated exchange -- pieces + junction log of all ranks to every rank.
        if (!c->prm.emit_replicated && c->knobs.get("CDBG_GLUE_REPLICATED") == nullptr) {
            const int rc = glue_sharded<W>(c);
            if (rc != DG_FALLBACK) return rc;
        }
        CK(glue_exchange(c));
    }
    if (!c->joined) CK(glue_join_impl<W>(c, false));         // (cdbg_glue_join ran it already in the sharded flow)
    hipStream_t s = c->stream;
    const uint64_t NP = c->n_pieces;
    const uint32_t NS = (uint32_t)(2 * NP);
    const float ms_join = c->st.ms_glue;
    HostMarks hm;
    Timer t; CK(t.start(s));
    // one GPU: walk the chains from their heads; chains the walk does not take (too long, closed) -> list ranking below, for the rest of this context's life
    bool walked = false;
    if (c->prm.world_size <= 1 && !c->force_multi && !c->xchg_done && !c->walk_off && c->knobs.get("CDBG_GLUE_RANK") == nullptr) CK(glue_walk(c, &walked));
    if (walked) {
        { uint64_t cur[2]; CK(read_u64(c->cursors.p + 2, cur, 2)); c->n_unitigs = cur[0]; c->unitig_total = cur[1]; }
        CK(pack_unitigs(c));
        float ms_fin = 0; CK(t.stop(&ms_fin));
        hm.mark("glue: walk + copy");
        c->st.ms_glue = ms_join + ms_fin;
        CK(check_device_error(c, "glue"));
        c->joined = false;
        c->st.n_glue_joined = c->n_join_local; c->st.n_unitigs = c->n_unitigs; c->st.unitig_bases = c->unitig_total; c->st.n_walked_unitigs = c->n_unitigs;
        c->st.ms_total += c->st.ms_glue;
        c->stage = 3;
        return CDBG_OK;
    }
    DBuf<uint32_t>& flag = c->rank_flag; DBuf<uint4>& st_a = c->rank_a; DBuf<uint4>& st_b = c->rank_b;
    CK(st_a.alloc(NS, false)); CK(st_b.alloc(NS, false));
    CK(flag.alloc(4, true));
    HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));
    HIPCK(hipMemsetAsync(c->derr.p, 0, 4 * sizeof(uint32_t), s));
    uint32_t* const link_p = c->link.p;

    uint64_t n_cycles_cut = 0;
    const uint32_t gridS = (NS + GLUE_THREADS - 1) / GLUE_THREADS;
    RankParams rp{};
    uint4* fa_st = nullptr;                                  // final state array of the 16-byte ranking
    const uint2* st8 = nullptr; uint4* hinfo = nullptr;      // what heads / emit read: 8-byte states, and the array for the per-head records
    bool ranked = false;
    if (NS) {
        // usual case (no closed chains): doubling on 8-byte states, expanded once at the end.  The 8-byte
        // state array (updated in place) lives in st_b; st_a takes the per-head records of heads / emit.
        int max_rounds8 = 2; while ((1ull << (max_rounds8 - 1)) < NS) ++max_rounds8;
        Rank8Params r8{ NS, link_p, c->piece_n.p, reinterpret_cast<uint2*>(st_b.p), flag.p };
        CDBG_LAUNCH(k_rank8_init, gridS, GLUE_THREADS, s, r8);
        for (int r = 0; r < max_rounds8 && !ranked; ++r) {
            HIPCK(hipMemsetAsync(flag.p, 0, sizeof(uint32_t), s));
            CDBG_LAUNCH(k_rank8_jump, gridS, GLUE_THREADS, s, r8);
            HIPCK(hipStreamSynchronize(s));
            uint32_t ch = 0; CK(read_u32(flag.p, &ch));
            if (!ch) ranked = true;
        }
        if (ranked) { st8 = r8.a; hinfo = st_a.p; }
    }
    if (NS && !ranked) {                                     // closed chains: the 16-byte version elects cut points
        int max_rounds = 2; while ((1ull << (max_rounds - 1)) < NS) ++max_rounds;
        for (int pass = 0; pass < 2; ++pass) {
            rp.n_states = NS; rp.link = link_p; rp.piece_n = c->piece_n.p;
            rp.st_a = st_a.p; rp.st_b = st_b.p; rp.changed = flag.p;
            CDBG_LAUNCH(k_rank_init, gridS, GLUE_THREADS, s, rp);
            bool converged = false;
            for (int r = 0; r < max_rounds; ++r) {
                HIPCK(hipMemsetAsync(flag.p, 0, sizeof(uint32_t), s));
                CDBG_LAUNCH(k_rank_jump, gridS, GLUE_THREADS, s, rp);
                std::swap(rp.st_a, rp.st_b);
                HIPCK(hipStreamSynchronize(s));
                uint32_t ch = 0; CK(read_u32(flag.p, &ch));
                if (!ch) { converged = true; break; }
            }
            fa_st = rp.st_a;
            if (converged) break;
            if (pass == 1) return fail(CDBG_E_INTERNAL, "list ranking did not converge after cutting cycles");
            // closed chains: cut each at its smallest piece, then rank again
            HIPCK(hipMemsetAsync(flag.p, 0, sizeof(uint32_t), s));
            CutParams cu{ NS, rp.st_a, link_p, flag.p };
            CDBG_LAUNCH(k_cut_cycles, gridS, GLUE_THREADS, s, cu);
            HIPCK(hipStreamSynchronize(s));
            uint32_t nc = 0; CK(read_u32(flag.p, &nc)); n_cycles_cut += nc;
        }
    }
    if (NS && !st8) {                                        // (16-byte ranking: narrow its final states into the other buffer)
        uint4* const other = (fa_st == st_a.p) ? st_b.p : st_a.p;
        RankNarrowParams np{ NS, fa_st, reinterpret_cast<uint2*>(other) };
        CDBG_LAUNCH(k_rank_narrow, gridS, GLUE_THREADS, s, np);
        st8 = reinterpret_cast<const uint2*>(other); hinfo = fa_st;
    }
    // unitig heads + emission
    const uint64_t ucap = std::max<uint64_t>(NP, 1);
    const uint64_t ocap = std::max<uint64_t>(c->n_piece_bases, 1);
    CK(c->unitig_off.alloc(ucap, false)); CK(c->unitig_len.alloc(ucap, false)); CK(c->unitig_kc.alloc(ucap, false));
    CK(c->unitig_bases.alloc(ocap + 64, false));             // (+ 64: the 2-bit packing pass reads whole 64-base chunks)
    if (c->prm.all_abundance_counts) CK(c->unitig_ab.alloc(ocap, false));
    HIPCK(hipMemsetAsync(c->cursors.p + 2, 0, 2 * sizeof(uint64_t), s));
    if (NS) {
        HeadParams hp{};
        hp.n_states = NS; hp.k = c->k; hp.link = link_p; hp.st = st8; hp.hinfo = hinfo;
        hp.unitig_off = c->unitig_off.p; hp.unitig_len = c->unitig_len.p; hp.unitig_kc = c->unitig_kc.p;
        hp.unitig_cap = ucap; hp.out_cap = ocap; hp.n_unitigs = c->cursors.p + 2; hp.out_cursor = c->cursors.p + 3; hp.error = c->derr.p;
        hp.own_lo = 0; hp.own_hi = NS;
        if (c->prm.world_size > 1 && !c->prm.emit_replicated && c->xchg_done) { hp.own_lo = (uint32_t)(2 * c->piece_lo); hp.own_hi = (uint32_t)(2 * c->piece_hi); }
        CDBG_LAUNCH(k_unitig_heads, (NS + HEADS_PER_WG - 1) / HEADS_PER_WG, GLUE_THREADS, s, hp);
        EmitParams ep{};
        ep.n_pieces = (uint32_t)NP; ep.k = c->k; ep.st = st8; ep.hinfo = hp.hinfo;
        ep.piece_n = c->piece_n.p; ep.piece_kc = c->piece_kc.p; ep.piece_boff = c->piece_boff.p; ep.piece_bases = c->piece_bases.p;
        ep.unitig_kc = c->unitig_kc.p; ep.out = c->unitig_bases.p;
        ep.piece_ab = c->prm.all_abundance_counts ? c->piece_ab.p : nullptr; ep.unitig_ab = c->unitig_ab.p;
        CDBG_LAUNCH(k_emit, (uint32_t)((NP + GLUE_THREADS - 1) / GLUE_THREADS), GLUE_THREADS, s, ep);
    }
    { uint64_t cur[2]; CK(read_u64(c->cursors.p + 2, cur, 2)); c->n_unitigs = cur[0]; c->unitig_total = cur[1]; }
    CK(pack_unitigs(c));
    float ms_fin = 0; CK(t.stop(&ms_fin));
    hm.mark("glue: rank + heads + emit");
    c->st.ms_glue = ms_join + ms_fin;
    CK(check_device_error(c, "glue"));
    c->joined = false;
    c->st.n_glue_joined = c->n_join_local; c->st.n_unitigs = c->n_unitigs; c->st.unitig_bases = c->unitig_total; c->st.n_cycles += n_cycles_cut;
    c->st.ms_total += c->st.ms_glue;
    c->stage = 3;
    return CDBG_OK;
}

// a unitig set in device memory: the resident result, or one the caller supplied (cdbg_verify_unitigs)
struct UnitigView { uint64_t U; const uint64_t* off; const uint32_t* len; const uint8_t* bases; uint64_t total_bases; };

// the link table of a unitig set (k_links.h): link_off[2U + 1], link_to[n_links]
template <int W>
int links_build(cdbg_ctx* c, const UnitigView& v, DBuf<uint64_t>& link_off, DBuf<uint32_t>& link_to, uint64_t& n_links) {
    hipStream_t s = c->stream;
    const uint64_t U = v.U, NE = 2 * U;
    // (k_links.h packs a slot index with a flag in bit 30: the table may have at most 2^30 slots)
    if (pow2_at_least(4 * U + 64) > (1ull << 30)) return fail(CDBG_E_INTERNAL, "too many unitigs (%llu) for the 30-bit slots of the link table", (unsigned long long)U);
    const uint32_t cap = (uint32_t)pow2_at_least(4 * U + 64);
    DBuf<uint64_t> lk_keys; DBuf<uint32_t> lk_cnt, lk_ends, end_slot, deg;
    CK(lk_keys.alloc((uint64_t)cap * W, false)); CK(lk_cnt.alloc((uint64_t)cap * 2, true));
    CK(lk_ends.alloc((uint64_t)cap * 2 * LINK_PER_FLAG, false)); CK(end_slot.alloc(NE, false)); CK(deg.alloc(NE, false));
    HIPCK(hipMemsetAsync(lk_keys.p, 0xFF, (uint64_t)cap * W * sizeof(uint64_t), s));
    CK(link_off.alloc(NE + 1, true));
    LinkParams lp{};
    lp.n_unitigs = U; lp.k = c->k; lp.unitig_off = v.off; lp.unitig_len = v.len; lp.bases = v.bases;
    lp.lk_keys = lk_keys.p; lp.lk_cnt = lk_cnt.p; lp.lk_ends = lk_ends.p; lp.lk_mask = cap - 1;
    lp.end_slot = end_slot.p; lp.deg = deg.p;
    n_links = 0;
    if (NE) {
        const uint64_t grid = (NE + LINK_THREADS - 1) / LINK_THREADS;
        CDBG_LAUNCH((k_link_insert<W>), grid, LINK_THREADS, s, lp);
        CDBG_LAUNCH(k_link_count, grid, LINK_THREADS, s, lp);
        const uint64_t nb = (NE + EXSCAN_BLOCK - 1) / EXSCAN_BLOCK;
        CK(c->exscan_tmp.alloc(nb + 1, false));
        const uint32_t* degp = deg.p;                        // (plain pointers: launch arguments are captured by value)
        uint64_t* const loff = link_off.p;
        CDBG_LAUNCH(k_exscan_sums, nb, EXSCAN_THREADS, s, degp, c->exscan_tmp.p, NE);
        CDBG_LAUNCH(k_exscan_top, 1, EXSCAN_THREADS, s, c->exscan_tmp.p, nb, loff + NE);
        CDBG_LAUNCH(k_exscan_apply, nb, EXSCAN_THREADS, s, degp, (const uint64_t*)c->exscan_tmp.p, loff, NE);
        CK(read_u64(loff + NE, &n_links));
        CK(link_to.alloc(n_links, false));
        lp.link_off = link_off.p; lp.link_to = link_to.p;
        CDBG_LAUNCH(k_link_fill, grid, LINK_THREADS, s, lp);
        HIPCK(hipStreamSynchronize(s));
    }
    return CDBG_OK;
}
UnitigView resident_unitigs(cdbg_ctx* c) { return UnitigView{ c->n_unitigs, c->unitig_off.p, c->unitig_len.p, c->unitig_bases.p, c->unitig_total }; }
inline bool holds_a_share(const cdbg_ctx* c) { return (c->prm.world_size > 1 || c->force_multi) && !c->prm.emit_replicated; }   // this rank holds a share of the job's unitigs

// The link table of a unitig set that is SHARDED over the ranks (k_links.h): collective.  Unitig ids are numbered rank after rank
// (c->unitig_id_base = the unitigs of the ranks before this one); link_off covers this rank's ends, link_to holds job-wide end ids.
template <int W>
int links_build_sharded(cdbg_ctx* c) {
    hipStream_t s = c->stream;
    const int world = c->prm.world_size, me = c->prm.rank;
    if (!c->have_tr) return fail(CDBG_E_STATE, "cdbg_link on a sharded unitig set needs the transport");
    const uint64_t U = c->n_unitigs, NE = 2 * U;
    std::vector<uint64_t> all(world); const uint64_t mine = U;
    if (c->tr.all_gather_u64(c->tr.user, &mine, all.data(), 1) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_u64 failed");
    uint64_t tot = 0, base = 0; for (int r = 0; r < world; ++r) { if (r == me) base = tot; tot += all[r]; }
    c->unitig_id_base = base; c->unitig_id_total = tot;
    const uint64_t NA = 2 * tot;
    int rc = CDBG_OK;
    DBuf<uint64_t> lk_keys, end_keys, all_keys; DBuf<uint32_t> lk_cnt, lk_ends, end_slot, deg, end_meta, all_meta;
    uint32_t cap = 0;
    auto local = [&]() -> int {                              // (a rank-local failure must not leave the others in the collective: agreed on below)
        if (NA >= (1ull << 31) || pow2_at_least(2 * NA + 64) > (1ull << 30)) return fail(CDBG_E_INTERNAL, "too many unitigs (%llu) for the 30-bit slots of the link table", (unsigned long long)tot);
        cap = (uint32_t)pow2_at_least(2 * NA + 64);
        CK(end_keys.alloc(NE * W + 1, false)); CK(end_meta.alloc(NE + 1, false)); CK(all_keys.alloc(NA * W + 1, false)); CK(all_meta.alloc(NA + 1, false));
        CK(lk_keys.alloc((uint64_t)cap * W, false)); CK(lk_cnt.alloc((uint64_t)cap * 2, true));
        CK(lk_ends.alloc((uint64_t)cap * 2 * LINK_PER_FLAG, false)); CK(end_slot.alloc(NA + 1, false)); CK(deg.alloc(NE + 1, false));
        HIPCK(hipMemsetAsync(lk_keys.p, 0xFF, (uint64_t)cap * W * sizeof(uint64_t), s));
        CK(c->link_off.alloc(NE + 1, true));
        return CDBG_OK;
    };
    rc = local();
    CK(agree(c, rc, "links: buffers"));
    LinkParams lp{};
    lp.n_unitigs = U; lp.k = c->k; lp.unitig_off = c->unitig_off.p; lp.unitig_len = c->unitig_len.p; lp.bases = c->unitig_bases.p;
    lp.lk_keys = lk_keys.p; lp.lk_cnt = lk_cnt.p; lp.lk_ends = lk_ends.p; lp.lk_mask = cap - 1;
    lp.end_slot = end_slot.p; lp.deg = deg.p; lp.end_keys = end_keys.p; lp.end_meta = end_meta.p;
    lp.all_keys = all_keys.p; lp.all_meta = all_meta.p; lp.n_all_ends = NA; lp.e0 = 2 * base;
    const uint64_t grid = (NE + LINK_THREADS - 1) / LINK_THREADS;
    if (NE) CDBG_LAUNCH((k_link_describe<W>), grid, LINK_THREADS, s, lp);
    std::vector<uint64_t> roff(world), rcnt(world);
    { uint64_t o = 0; for (int r = 0; r < world; ++r) { roff[r] = o * 2 * W * 8; rcnt[r] = all[r] * 2 * W * 8; o += all[r]; } }
    if (!c->tr_ordered) HIPCK(hipStreamSynchronize(s));
    if (c->tr.all_gather_v(c->tr.user, end_keys.p, NE * W * 8, all_keys.p, roff.data(), rcnt.data()) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_v failed");
    { uint64_t o = 0; for (int r = 0; r < world; ++r) { roff[r] = o * 2 * 4; rcnt[r] = all[r] * 2 * 4; o += all[r]; if (r != me) c->comm_bytes += (NE + 2 * all[r]) * (uint64_t)(W * 8 + 4); } }   // (sent to / received from every other rank)
    if (c->tr.all_gather_v(c->tr.user, end_meta.p, NE * 4, all_meta.p, roff.data(), rcnt.data()) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_v failed");
    c->n_links = 0;
    auto join = [&]() -> int {
        if (NA) CDBG_LAUNCH((k_link_insert_described<W>), (NA + LINK_THREADS - 1) / LINK_THREADS, LINK_THREADS, s, lp);
        if (NE) {
            CDBG_LAUNCH(k_link_count, grid, LINK_THREADS, s, lp);
            const uint64_t nb = (NE + EXSCAN_BLOCK - 1) / EXSCAN_BLOCK;
            CK(c->exscan_tmp.alloc(nb + 1, false));
            const uint32_t* degp = deg.p; uint64_t* const loff = c->link_off.p;
            CDBG_LAUNCH(k_exscan_sums, nb, EXSCAN_THREADS, s, degp, c->exscan_tmp.p, NE);
            CDBG_LAUNCH(k_exscan_top, 1, EXSCAN_THREADS, s, c->exscan_tmp.p, nb, loff + NE);
            CDBG_LAUNCH(k_exscan_apply, nb, EXSCAN_THREADS, s, degp, (const uint64_t*)c->exscan_tmp.p, loff, NE);
            CK(read_u64(loff + NE, &c->n_links));
            CK(c->link_to.alloc(c->n_links, false));
            lp.link_off = c->link_off.p; lp.link_to = c->link_to.p;
            CDBG_LAUNCH(k_link_fill, grid, LINK_THREADS, s, lp);
        }
        HIPCK(hipStreamSynchronize(s));
        return CDBG_OK;
    };
    rc = join();
    CK(agree(c, rc, "links: join"));
    return CDBG_OK;
}

template <int W>
int link_impl(cdbg_ctx* c) {
    if (c->stage < 3) return fail(CDBG_E_STATE, "cdbg_link before cdbg_glue");
    if (holds_a_share(c)) CK(links_build_sharded<W>(c));
    else { CK(links_build<W>(c, resident_unitigs(c), c->link_off, c->link_to, c->n_links)); c->unitig_id_base = 0; c->unitig_id_total = c->n_unitigs; }
    c->linked = true;
    return CDBG_OK;
}

// k-mer-set equality + maximality of a unitig set against the solid table (k_verify.h); out[8]
template <int W>
int verify_view(cdbg_ctx* c, const UnitigView& v, const uint64_t* link_off, const uint32_t* link_to, bool with_links, uint64_t* out) {
    hipStream_t s = c->stream;
    DBuf<uint64_t> d; CK(d.alloc(8, true));
    VerifyParams vp{ v.U, c->k, v.off, v.len, v.bases,
                     c->seg_off.p, c->seg_n.p, c->solid_keys.p, c->solid_cnt.p, c->n_local_parts, link_off, link_to, d.p };
    if (v.U) CDBG_LAUNCH((k_verify_unitig_kmers<W>), std::min<uint64_t>((v.U + 255) / 256, 1u << 16), 256, s, vp);
    CDBG_LAUNCH((k_verify_solid<W>), std::min<uint64_t>((c->n_local_parts + 255) / 256, 1u << 16), 256, s, vp);
    if (with_links && v.U) CDBG_LAUNCH(k_verify_maximal, (2 * v.U + 255) / 256, 256, s, vp);
    HIPCK(hipStreamSynchronize(s));
    CK(read_u64(d.p, out, 8));
    if (!with_links) out[6] = out[7] = ~0ull;
    return CDBG_OK;
}
// edge conservation (k_verify.h): out[0] = D from the solid k-mers, out[1] = links of the unitig set, out[2] = 2 sum(LN - k), out[3] = distinct junctions
template <int W>
int verify_edges_view(cdbg_ctx* c, const UnitigView& v, uint64_t n_links, uint64_t* out) {
    hipStream_t s = c->stream;
    const uint64_t n_home = c->st.n_solid;
    const uint64_t cap = pow2_at_least(3 * n_home + 1024);   // (at most 2 junctions per k-mer: never more than two thirds full)
    if (cap > (1ull << 32)) return fail(CDBG_E_INTERNAL, "too many solid k-mers (%llu) for the 32-bit slots of the junction table", (unsigned long long)n_home);
    DBuf<uint64_t> jt_keys, d; DBuf<uint32_t> jt_cnt;
    CK(jt_keys.alloc(cap * W, false)); CK(jt_cnt.alloc(cap, false)); CK(d.alloc(4, true));
    HIPCK(hipMemsetAsync(jt_keys.p, 0xFF, cap * W * sizeof(uint64_t), s));
    HIPCK(hipMemsetAsync(jt_cnt.p, 0, cap * sizeof(uint32_t), s));
    VerifyEdgeParams ep{ c->k, c->seg_off.p, c->seg_n.p, c->solid_keys.p, c->solid_cnt.p, c->n_local_parts, jt_keys.p, jt_cnt.p, (uint32_t)(cap - 1), d.p };
    CDBG_LAUNCH((k_verify_edge_insert<W>), std::min<uint64_t>((c->n_local_parts + 255) / 256, 1u << 16), 256, s, ep);
    CDBG_LAUNCH(k_verify_edge_sum, std::min<uint64_t>((cap + 255) / 256, 1u << 16), 256, s, ep);
    HIPCK(hipStreamSynchronize(s));
    uint64_t r[4]; CK(read_u64(d.p, r, 4));
    if (r[2]) return fail(CDBG_E_INTERNAL, "junction table overflow in cdbg_verify_edges (%llu ends lost)", (unsigned long long)r[2]);
    out[0] = r[0]; out[1] = n_links; out[2] = 2 * (v.total_bases - v.U * (uint64_t)c->k); out[3] = r[1];
    return CDBG_OK;
}

// the unitig definition checked on the resident result (k_verify.h)
template <int W>
int verify_impl(cdbg_ctx* c, uint64_t* out) {
    if (c->stage < 3) return fail(CDBG_E_STATE, "cdbg_verify before cdbg_glue");
    const bool sharded_set = holds_a_share(c);
    if (!sharded_set && !c->linked) CK(link_impl<W>(c));
    return verify_view<W>(c, resident_unitigs(c), c->link_off.p, c->link_to.p, !sharded_set, out);
}
template <int W>
int verify_edges_impl(cdbg_ctx* c, uint64_t* out) {
    if (c->stage < 3) return fail(CDBG_E_STATE, "cdbg_verify_edges before cdbg_glue");
    if (c->prm.world_size > 1 || c->force_multi) { out[0] = out[1] = out[2] = out[3] = ~0ull; return CDBG_OK; }   // (a rank counts a share of the k-mers; their junctions belong to all ranks)
    if (!c->linked) CK(link_impl<W>(c));
    return verify_edges_view<W>(c, resident_unitigs(c), c->n_links, out);
}
// the same checks for a unitig set the CALLER supplies (host memory): out[0..7] as cdbg_verify, out[8..11] as cdbg_verify_edges
template <int W>
int verify_unitigs_impl(cdbg_ctx* c, const char* bases, const uint64_t* off, uint64_t n, uint64_t* out) {
    if (c->stage < 1) return fail(CDBG_E_STATE, "cdbg_verify_unitigs before cdbg_count");
    if (c->prm.world_size > 1 || c->force_multi) return fail(CDBG_E_STATE, "cdbg_verify_unitigs needs the whole solid set on this rank");
    const uint64_t total = n ? off[n] : 0;
    std::vector<uint32_t> len(std::max<uint64_t>(n, 1));
    for (uint64_t i = 0; i < n; ++i) {
        if (off[i + 1] < off[i] || off[i + 1] - off[i] < (uint64_t)c->k || off[i + 1] - off[i] > 0xFFFFFFFFull) return fail(CDBG_E_PARAM, "unitig %llu: bad length", (unsigned long long)i);
        len[i] = (uint32_t)(off[i + 1] - off[i]);
    }
    for (uint64_t i = 0; i < total; ++i) if (!base_valid((uint8_t)bases[i])) return fail(CDBG_E_PARAM, "unitig base %llu is not one of ACGT", (unsigned long long)i);
    DBuf<uint64_t> d_off, l_off; DBuf<uint32_t> d_len, l_to; DBuf<uint8_t> d_bases;
    CK(d_off.alloc(n + 1, false)); CK(d_len.alloc(n, false)); CK(d_bases.alloc(total + 64, false));
    HIPCK(hipMemcpy(d_off.p, off, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(d_len.p, len.data(), std::max<uint64_t>(n, 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (total) HIPCK(hipMemcpy(d_bases.p, bases, total, hipMemcpyHostToDevice));
    const UnitigView v{ n, d_off.p, d_len.p, d_bases.p, total };
    uint64_t nl = 0;
    CK(links_build<W>(c, v, l_off, l_to, nl));
    CK(verify_view<W>(c, v, l_off.p, l_to.p, true, out));
    return verify_edges_view<W>(c, v, nl, out + 8);
}

}  // namespace
