// host_exchange.h -- libcdbg.so, host side: the REPLICATED glue exchange of a multi-GPU job (pieces + junction log of every
// rank to every rank; emit_replicated = 1 and the fallback for closed chains that cross ranks).  Included by cdbg_impl.cpp only.
#pragma once

namespace {

template <int W> int glue_join_impl(cdbg_ctx* c, bool sharded);   // (host_glue.h)

// ---- the replicated glue exchange, step by step (internal; the caller-driven variant of this API was removed in round 3:
// a context with world_size > 1 always exchanges through its transport) ----
// ---- multi-GPU exchange: the pieces and glue records of every rank are gathered (RCCL all-gather
// driven by the caller through torch.distributed; this library only copies device-to-device into / out
// of caller-provided device buffers) and merged in rank order, after which cdbg_glue runs on the union ----
int xchg_export(cdbg_ctx* c, int what, void* dst_dev, uint64_t nbytes) {
    if (!c || !dst_dev) return fail(CDBG_E_PARAM, "null argument");
    if (c->stage != 2) return fail(CDBG_E_STATE, "xchg_export before cdbg_compact");
    if (c->direct_join && what >= 4) return fail(CDBG_E_STATE, "this single-rank context keeps no junction log (CDBG_GLUE_LOG=1 keeps it)");
    const void* src = nullptr; uint64_t have = 0;
    switch (what) {
        case 0: src = c->piece_n.p; have = c->n_pieces * sizeof(uint32_t); break;
        case 1: src = c->piece_kc.p; have = c->n_pieces * sizeof(uint64_t); break;
        case 2: src = c->piece_boff.p; have = c->n_pieces * sizeof(uint64_t); break;
        case 3: src = c->piece_bases.p; have = c->n_piece_bases; break;
        case 4: src = c->glog_keys.p; have = c->n_glog * (uint64_t)c->W * sizeof(uint64_t); break;
        case 5: src = c->glog_tag.p; have = c->n_glog * sizeof(uint32_t); break;
        default: return fail(CDBG_E_PARAM, "unknown export kind %d", what);
    }
    if (nbytes < have) return fail(CDBG_E_PARAM, "export buffer too small (%llu < %llu)", (unsigned long long)nbytes, (unsigned long long)have);
    if (have) HIPCK(hipMemcpyAsync(dst_dev, src, have, hipMemcpyDeviceToDevice, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    return CDBG_OK;
}
int xchg_begin(cdbg_ctx* c, uint64_t total_pieces, uint64_t total_bases, uint64_t total_glog) {
    if (!c) return fail(CDBG_E_PARAM, "null context");
    if (c->stage != 2) return fail(CDBG_E_STATE, "xchg_begin before cdbg_compact");
    // the merged arrays are swapped with the context's own in xchg_end: give them at least the same
    // capacity, so that a re-run after cdbg_reset finds arrays that are large enough and never reallocates
    CK(c->mg_n.alloc(total_pieces, false, c->piece_n.cap)); CK(c->mg_kc.alloc(total_pieces, false, c->piece_kc.cap));
    CK(c->mg_boff.alloc(total_pieces, false, c->piece_boff.cap));
    const uint64_t bases_slack = 64ull * 4096;               // the packed exchange starts every rank's bases on a 64-byte boundary
    CK(c->mg_bases.alloc(total_bases + bases_slack, false, c->piece_bases.cap));
    CK(c->mg_gkeys.alloc(total_glog * c->W, false, c->glog_keys.cap)); CK(c->mg_gtag.alloc(total_glog, false, c->glog_tag.cap));
    if (c->prm.all_abundance_counts) CK(c->mg_ab.alloc(total_bases + bases_slack, false, c->piece_ab.cap));
    c->mg_np = c->mg_nb = c->mg_nl = 0; c->mg_cap_p = total_pieces; c->mg_cap_b = total_bases + bases_slack; c->mg_cap_l = total_glog; c->mg_open = true;
    return CDBG_OK;
}
// ---- packed variant of the exchange: bases travel as 2 bits (pieces padded to whole bytes, reservation gaps squeezed
// out) and the per-piece base offsets do not travel at all -- the receiver recomputes them from the piece lengths ----
int xchg_sizes_packed(cdbg_ctx* c, uint64_t out[4]) {
    if (!c || !out) return fail(CDBG_E_PARAM, "null argument");
    if (c->stage != 2) return fail(CDBG_E_STATE, "xchg_* needs a compacted, not yet glued context");
    if (c->direct_join) return fail(CDBG_E_STATE, "this single-rank context joined its junction records in place and keeps no log to exchange (create it with world_size > 1, or set CDBG_GLUE_LOG=1)");
    hipStream_t s = c->stream;
    const uint64_t NP = c->n_pieces;
    CK(c->xp_lens.alloc(NP, false)); CK(c->xp_uoff.alloc(NP + 1, false));
    if (NP) {
        PackLenParams lp{ NP, c->k, c->piece_n.p, c->xp_lens.p };
        CDBG_LAUNCH(k_pack_lens, (NP + 255) / 256, 256, s, lp);
    }
    CK(exscan_u32(c, c->xp_lens.p, c->xp_uoff.p, NP));
    HIPCK(hipStreamSynchronize(s));
    CK(read_u64(c->xp_uoff.p + NP, &c->xp_unpacked));
    const uint64_t chunks = (c->xp_unpacked + 63) / 64;      // 64 bases -> 16 bytes per lane
    c->xp_bytes = chunks * 16;
    CK(c->xp_dense.alloc(chunks * 64 + 64, false)); CK(c->xp_bases.alloc(c->xp_bytes + 16, false));
    if (chunks) HIPCK(hipMemsetAsync(c->xp_dense.p + (chunks - 1) * 64, 'A', 64, s));     // tail padding of the last chunk
    if (NP) {
        SqueezeParams sq{ NP, c->xp_lens.p, c->xp_uoff.p, c->piece_boff.p, c->piece_bases.p, c->xp_dense.p };
        CDBG_LAUNCH(k_squeeze_bases, (NP + 255) / 256, 256, s, sq);
    }
    if (chunks) {
        StreamPackParams pp{ chunks, c->xp_dense.p, c->xp_bases.p, c->xp_unpacked };
        CDBG_LAUNCH(k_pack_stream, (chunks + 255) / 256, 256, s, pp);
    }
    HIPCK(hipStreamSynchronize(s));
    out[0] = NP; out[1] = c->xp_unpacked; out[2] = c->n_glog; out[3] = c->xp_bytes;
    return CDBG_OK;
}
int xchg_export_packed(cdbg_ctx* c, void* dst_dev, uint64_t nbytes) {
    if (!c || !dst_dev) return fail(CDBG_E_PARAM, "null argument");
    if (c->stage != 2 || !c->xp_bases.p) return fail(CDBG_E_STATE, "xchg_export_packed before xchg_sizes_packed");
    if (nbytes < c->xp_bytes) return fail(CDBG_E_PARAM, "export buffer too small (%llu < %llu)", (unsigned long long)nbytes, (unsigned long long)c->xp_bytes);
    if (c->xp_bytes) HIPCK(hipMemcpyAsync(dst_dev, c->xp_bases.p, c->xp_bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    return CDBG_OK;
}
int xchg_add_packed(cdbg_ctx* c, uint64_t n_pieces, uint64_t n_bases, uint64_t n_packed, uint64_t n_glog, const void* piece_n, const void* piece_kc,
                             const void* packed_bases, const void* glog_keys, const void* glog_tag) {
    if (!c) return fail(CDBG_E_PARAM, "null context");
    if (!c->mg_open) return fail(CDBG_E_STATE, "xchg_add_packed without xchg_begin");
    c->mg_nb = (c->mg_nb + 63) / 64 * 64;                    // 16-byte stores of the streaming unpack
    if (c->mg_np + n_pieces > c->mg_cap_p || c->mg_nb + n_bases > c->mg_cap_b || c->mg_nl + n_glog > c->mg_cap_l)
        return fail(CDBG_E_PARAM, "xchg_add_packed exceeds the totals given to xchg_begin");
    if (2 * (c->mg_np + n_pieces) >= 0x7FFFFFF0ULL) return fail(CDBG_E_INTERNAL, "too many pieces for 31-bit end ids");
    if (n_packed != (n_bases + 63) / 64 * 16) return fail(CDBG_E_PARAM, "packed size %llu does not match %llu bases", (unsigned long long)n_packed, (unsigned long long)n_bases);
    hipStream_t s = c->stream;
    // offsets of the source rank's pieces inside its gap-free stream, recomputed here from its piece_n
    // (context members: a step must not allocate or free device memory once the buffers of the first step exist)
    DBuf<uint32_t>& lens = c->xr_lens; DBuf<uint64_t>& uoff = c->xr_uoff;
    CK(lens.alloc(n_pieces, false)); CK(uoff.alloc(n_pieces + 1, false));
    if (n_pieces) {
        PackLenParams lp{ n_pieces, c->k, (const uint32_t*)piece_n, lens.p };
        CDBG_LAUNCH(k_pack_lens, (n_pieces + 255) / 256, 256, s, lp);
    }
    CK(exscan_u32(c, lens.p, uoff.p, n_pieces));
    HIPCK(hipStreamSynchronize(s));
    uint64_t tu = 0; CK(read_u64(uoff.p + n_pieces, &tu));
    if (tu != n_bases) return fail(CDBG_E_PARAM, "piece lengths (%llu bases) do not match the packed stream (%llu bases)", (unsigned long long)tu, (unsigned long long)n_bases);
    const uint64_t chunks = (n_bases + 63) / 64;
    if (chunks) {
        StreamUnpackParams up{ chunks, (const uint8_t*)packed_bases, c->mg_bases.p + c->mg_nb, n_bases };
        CDBG_LAUNCH(k_unpack_stream, (chunks + 255) / 256, 256, s, up);
    }
    MergeParams mp{ n_pieces, n_glog, c->mg_np, c->mg_nb, c->mg_nl, c->W,
                    (const uint32_t*)piece_n, (const uint64_t*)piece_kc, uoff.p, (const uint64_t*)glog_keys, (const uint32_t*)glog_tag,
                    c->mg_n.p, c->mg_kc.p, c->mg_boff.p, c->mg_gkeys.p, c->mg_gtag.p };
    const uint64_t work = std::max(n_pieces, n_glog);
    if (work) CDBG_LAUNCH(k_merge_append, std::min<uint64_t>((work + 255) / 256, MAX_GRID), 256, s, mp);
    HIPCK(hipStreamSynchronize(s));
    c->last_add_np = c->mg_np; c->last_add_nb = c->mg_nb; c->last_add_pieces = n_pieces;
    c->mg_np += n_pieces; c->mg_nb += n_bases; c->mg_nl += n_glog;
    return CDBG_OK;
}
// -all-abundance-counts: the abundances of this rank's pieces as a gap-free stream, one u32 per k-mer, in the piece
// order of xchg_sizes_packed
int xchg_abundance_values(cdbg_ctx* c, uint64_t* n_values) {
    if (!c || !n_values) return fail(CDBG_E_PARAM, "null argument");
    if (!c->prm.all_abundance_counts) return fail(CDBG_E_STATE, "context was created without all_abundance_counts");
    if (c->stage != 2) return fail(CDBG_E_STATE, "xchg_* needs a compacted, not yet glued context");
    CK(c->xp_aoff.alloc(c->n_pieces + 1, false));
    CK(exscan_u32(c, c->piece_n.p, c->xp_aoff.p, c->n_pieces));
    HIPCK(hipStreamSynchronize(c->stream));
    CK(read_u64(c->xp_aoff.p + c->n_pieces, &c->xp_nab));
    *n_values = c->xp_nab; c->xp_ab_ready = true;
    return CDBG_OK;
}
int xchg_export_abundances(cdbg_ctx* c, void* dst_dev, uint64_t nbytes) {
    if (!c || !dst_dev) return fail(CDBG_E_PARAM, "null argument");
    if (c->stage != 2 || !c->xp_ab_ready) return fail(CDBG_E_STATE, "xchg_export_abundances before xchg_abundance_values");
    const uint64_t NP = c->n_pieces;
    if (nbytes < c->xp_nab * sizeof(uint32_t)) return fail(CDBG_E_PARAM, "export buffer too small (%llu < %llu)", (unsigned long long)nbytes, (unsigned long long)(c->xp_nab * sizeof(uint32_t)));
    if (NP) {
        AbStreamParams ap{ NP, c->k, 0, c->piece_n.p, c->xp_aoff.p, nullptr, c->piece_boff.p, 0, c->piece_ab.p, (uint32_t*)dst_dev };
        CDBG_LAUNCH(k_ab_stream, (NP + 255) / 256, 256, c->stream, ap);
    }
    HIPCK(hipStreamSynchronize(c->stream));
    return CDBG_OK;
}
// ... and the stream of the rank whose pieces the latest xchg_add_packed appended
int xchg_add_abundances(cdbg_ctx* c, const void* ab_stream, uint64_t n_values) {
    if (!c || !ab_stream) return fail(CDBG_E_PARAM, "null argument");
    if (!c->prm.all_abundance_counts) return fail(CDBG_E_STATE, "context was created without all_abundance_counts");
    if (!c->mg_open) return fail(CDBG_E_STATE, "xchg_add_abundances without xchg_begin");
    const uint64_t NP = c->last_add_pieces;
    CK(c->xr_aoff.alloc(NP + 1, false));
    CK(exscan_u32(c, c->mg_n.p + c->last_add_np, c->xr_aoff.p, NP));
    HIPCK(hipStreamSynchronize(c->stream));
    uint64_t tot = 0; CK(read_u64(c->xr_aoff.p + NP, &tot));
    if (n_values != tot) return fail(CDBG_E_PARAM, "abundance stream of %llu values does not match the %llu k-mers of the pieces added last", (unsigned long long)n_values, (unsigned long long)tot);
    if (NP) {
        AbStreamParams ap{ NP, c->k, 1, c->mg_n.p + c->last_add_np, c->xr_aoff.p, c->xr_uoff.p, nullptr, c->last_add_nb, c->mg_ab.p, (uint32_t*)ab_stream };
        CDBG_LAUNCH(k_ab_stream, (NP + 255) / 256, 256, c->stream, ap);
    }
    HIPCK(hipStreamSynchronize(c->stream));
    return CDBG_OK;
}
int xchg_end(cdbg_ctx* c) {
    if (!c) return fail(CDBG_E_PARAM, "null context");
    if (!c->mg_open) return fail(CDBG_E_STATE, "xchg_end without xchg_begin");
    c->piece_n.swap(c->mg_n); c->piece_kc.swap(c->mg_kc); c->piece_boff.swap(c->mg_boff);
    c->piece_bases.swap(c->mg_bases); c->glog_keys.swap(c->mg_gkeys); c->glog_tag.swap(c->mg_gtag);
    if (c->prm.all_abundance_counts) c->piece_ab.swap(c->mg_ab);
    c->n_pieces = c->mg_np; c->n_piece_bases = c->mg_nb; c->n_glog = c->mg_nl; c->glog_cap = c->mg_cap_l;
    c->mg_open = false;
    HIPCK(hipStreamSynchronize(c->stream));
    return CDBG_OK;
}

// ---- multi-GPU: sharded junction join.  After xchg_end every rank holds the union of the glue records;
// instead of every rank joining all of them, xchg_glue_join joins this rank's share of the junctions, the caller
// MAX-all-reduces the int32 link arrays (cdbg_glue_links_export / _import) and cdbg_glue then ranks and emits ----
int xchg_glue_join(cdbg_ctx* c, uint64_t* n_ends) {
    if (!c || !n_ends) return fail(CDBG_E_PARAM, "null argument");
    if (c->stage != 2) return fail(CDBG_E_STATE, "xchg_glue_join needs a compacted, not yet glued context");
    int rc;
    switch (c->W) {
        case 1: rc = glue_join_impl<1>(c, true); break; case 2: rc = glue_join_impl<2>(c, true); break; case 3: rc = glue_join_impl<3>(c, true); break; case 4: rc = glue_join_impl<4>(c, true); break;
#if CDBG_MAX_W >= 8
        case 5: rc = glue_join_impl<5>(c, true); break; case 6: rc = glue_join_impl<6>(c, true); break; case 7: rc = glue_join_impl<7>(c, true); break; case 8: rc = glue_join_impl<8>(c, true); break;
#endif
        default: rc = fail(CDBG_E_PARAM, "k-mers of %d words: rebuild with CDBG_MAX_W", c->W);
    }
    if (rc == CDBG_OK) *n_ends = 2 * c->n_pieces;
    return rc;
}

// ---- multi-GPU glue exchange, driven by the library through the context's transport: every rank's pieces (lengths,
// abundance sums, bases packed 4 per byte, no offsets) and junction log are all-gathered and merged in rank order
// (xchg_*); the junction hash-join is sharded by key hash and its result, one partner id per piece end, is
// combined with ONE MAX all-reduce (every end is set by exactly one rank) ----
int glue_exchange(cdbg_ctx* c) {
    const int world = c->prm.world_size, W = c->W;
    hipStream_t s = c->stream;
    Timer t; CK(t.start(s));
    uint64_t mine[4]; CK(xchg_sizes_packed(c, mine));          // pieces, bases once unpacked, glue-log records, packed bytes
    std::vector<uint64_t> all((size_t)world * 4);
    if (c->tr.all_gather_u64(c->tr.user, mine, all.data(), 4) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_u64 failed");
    auto col = [&](int r, int j) { return all[(size_t)r * 4 + j]; };
    // the five arrays: piece_n (u32), piece_kc (u64), packed bases, glue keys (u64 x W), glue tags (u32)
    const uint64_t item[5] = { 4, 8, 1, 8ull * W, 4 }; const int which[5] = { 0, 0, 3, 2, 2 };
    std::vector<std::vector<uint64_t>> roff(5, std::vector<uint64_t>(world)), rcnt(5, std::vector<uint64_t>(world));
    DBuf<uint8_t>& sendbuf = c->xsend;
    for (int a = 0; a < 5; ++a) {
        uint64_t tot = 0;
        for (int r = 0; r < world; ++r) { rcnt[a][r] = col(r, which[a]) * item[a]; roff[a][r] = tot; tot += (rcnt[a][r] + 15) / 16 * 16; }
        CK(c->xg[a].alloc(tot + 16, false));
        const uint64_t nb = rcnt[a][c->prm.rank];
        CK(sendbuf.alloc(nb + 16, false));
        if (a == 2) CK(xchg_export_packed(c, sendbuf.p, nb + 16));
        else CK(xchg_export(c, a == 0 ? 0 : a == 1 ? 1 : a == 3 ? 4 : 5, sendbuf.p, nb + 16));
        if (c->tr.all_gather_v(c->tr.user, sendbuf.p, nb, c->xg[a].p, roff[a].data(), rcnt[a].data()) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_v failed");
        for (int r = 0; r < world; ++r) if (r != c->prm.rank) c->comm_bytes += nb + rcnt[a][r];
    }
    uint64_t tp = 0, tb = 0, tl = 0;
    for (int r = 0; r < world; ++r) { if (r == c->prm.rank) c->piece_lo = tp; tp += col(r, 0); if (r == c->prm.rank) c->piece_hi = tp; tb += col(r, 1); tl += col(r, 2); }
    // -all-abundance-counts: a sixth array, one u32 per k-mer of the rank's pieces
    std::vector<uint64_t> aoff(world), acnt(world);
    if (c->prm.all_abundance_counts) {
        uint64_t nv = 0; CK(xchg_abundance_values(c, &nv));
        std::vector<uint64_t> allv(world);
        if (c->tr.all_gather_u64(c->tr.user, &nv, allv.data(), 1) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_u64 failed");
        uint64_t tot = 0;
        for (int r = 0; r < world; ++r) { acnt[r] = allv[r] * 4; aoff[r] = tot; tot += (acnt[r] + 15) / 16 * 16; }
        CK(c->xp_ab.alloc(tot / 4 + 4, false));
        const uint64_t nb = acnt[c->prm.rank];
        CK(sendbuf.alloc(nb + 16, false));
        CK(xchg_export_abundances(c, sendbuf.p, nb + 16));
        if (c->tr.all_gather_v(c->tr.user, sendbuf.p, nb, c->xp_ab.p, aoff.data(), acnt.data()) != 0) return fail(CDBG_E_INTERNAL, "transport all_gather_v failed");
        for (int r = 0; r < world; ++r) if (r != c->prm.rank) c->comm_bytes += nb + acnt[r];
    }
    CK(xchg_begin(c, tp, tb, tl));
    for (int r = 0; r < world; ++r) {
        CK(xchg_add_packed(c, col(r, 0), col(r, 1), col(r, 3), col(r, 2), c->xg[0].p + roff[0][r], c->xg[1].p + roff[1][r],
                                    c->xg[2].p + roff[2][r], c->xg[3].p + roff[3][r], c->xg[4].p + roff[4][r]));
        if (c->prm.all_abundance_counts) CK(xchg_add_abundances(c, (const uint8_t*)c->xp_ab.p + aoff[r], acnt[r] / 4));
    }
    CK(xchg_end(c));
    // sharded junction join
    uint64_t n_ends = 0; CK(xchg_glue_join(c, &n_ends));
    if (c->tr.all_reduce_max_i32(c->tr.user, c->link.p, n_ends) != 0) return fail(CDBG_E_INTERNAL, "transport all_reduce_max_i32 failed");
    c->comm_bytes += 2 * n_ends * 4 * (uint64_t)(world - 1) / (uint64_t)world;          // (ring all-reduce volume per rank)
    float ms = 0; CK(t.stop(&ms)); c->st.ms_exchange += ms;
    c->xchg_done = true;
    return CDBG_OK;
}

}  // namespace
