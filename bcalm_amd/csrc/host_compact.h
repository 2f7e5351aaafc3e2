// host_compact.h -- libcdbg.so, host side of stage 2: the compaction tiers (one wave per
// bucket, workgroup per bucket, second-level split of what is left, HBM tables as the last resort).  Included by cdbg_impl.cpp only.
#pragma once

namespace {

template <int W>
int compact_impl(cdbg_ctx* c) {
    constexpr int TS = Cfg<W>::TSK;
    if (c->stage < 1) return fail(CDBG_E_STATE, "cdbg_compact before cdbg_count");
    hipStream_t s = c->stream;
    const uint64_t NPL = c->n_local_parts;
    const uint64_t S = c->st.n_solid;
    HostMarks hm;
    Timer t; CK(t.start(s));
    // (the junction join works from the glue LOG; its tables are built in cdbg_glue)
    CK(c->cursors.alloc(8, false));

    // Single-rank contexts: the glue records go straight into the join buckets of the glue stage (k_glue.h) instead of a
    // sequential log that a scatter pass re-reads; contexts that exchange the log with other ranks, and the global-table
    // join (CDBG_GLUE_TABLE), keep the log.  Observed 1.8-2.0 records per solid traveller (bound: 3): buckets sized for
    // a mean fill of at most 96 of JB_CAP = 256 at 2.2, 131 at the bound.
    bool direct = !(c->prm.world_size > 1 || c->force_multi) && c->knobs.get("CDBG_GLUE_TABLE") == nullptr && c->knobs.get("CDBG_GLUE_LOG") == nullptr;
    int log_jb = 0;
    // (the second-level split of overfull buckets, k_split.h, makes travellers of its own -- a home k-mer whose two junctions land in
    //  different sub-buckets leaves a copy, up to 3 more glue records each: what a first attempt learns about them sizes the next)
    uint64_t split_extra = 0, split_buckets_now = 0;
    for (int attempt = 0; attempt < 3; ++attempt) {
        const uint64_t trav_est = c->st.n_solid_travellers + split_extra, sized_extra = split_extra;
        split_buckets_now = 0;
        { log_jb = 0; const uint64_t est = trav_est * 22 / 10 + 1024; while ((96ull << log_jb) < est && log_jb < 26) ++log_jb; }
        if (const char* ev = c->knobs.get("CDBG_JOIN_LOG_JB")) log_jb = std::max(0, std::min(26, atoi(ev)));   // (tests: force the overflow fallback)
        // glue log: <= 2 open ends + 1 confirm per junction, one junction per solid traveller at most; the tail of a
        // chunk that the next bucket does not fit into is abandoned, hence the generous second attempt
        // (every persistent wave of tier 0 may strand one partly used chunk of each output array as well)
        // (... in the wave tiers over the buckets and over the sub-buckets of the second-level split)
        const uint64_t wave_slack = 2 * std::min<uint64_t>(NPL, 256ull * 32) + 2 * std::min<uint64_t>(c->n_solid_entries / 32 + 4, 256ull * 32);
        c->glog_cap = (attempt == 0 ? 3 : 8) * trav_est + (attempt + 1) * (CHUNK_SLACK_WGS * (uint64_t)GLOG_CHUNK + wave_slack * CW_GLOG_CHUNK) + 64;
        if (direct) {
            c->glog_cap = ~0ull >> 2;                        // (the log cursor only counts)
            CK(c->jfill.alloc(1ull << log_jb, false)); CK(c->jrecs.alloc((JB_CAP << log_jb) * (uint64_t)(W + 1), false));
            HIPCK(hipMemsetAsync(c->jfill.p, 0, sizeof(uint32_t) << log_jb, s));
        } else {
            CK(c->glog_keys.alloc(c->glog_cap * W, false)); CK(c->glog_tag.alloc(c->glog_cap, false));
        }
        const uint64_t pslack = CHUNK_SLACK_WGS * (uint64_t)PIECE_CHUNK + wave_slack * CW_PIECE_CHUNK, bslack = CHUNK_SLACK_WGS * (uint64_t)BASES_CHUNK + wave_slack * CW_BASES_CHUNK;
        const uint64_t pcap = (attempt == 0 ? std::min<uint64_t>(S, S / 3 + 4096) + 16 : S + 16) + pslack;
        const uint64_t bcap = (attempt == 0 ? S + (pcap - pslack) * (uint64_t)(c->k - 1) + 64 : S * (uint64_t)c->k + 64) + bslack;
        CK(c->piece_n.alloc(pcap, false)); HIPCK(hipMemsetAsync(c->piece_n.p, 0, pcap * sizeof(uint32_t), s));
        CK(c->piece_kc.alloc(pcap, false)); CK(c->piece_boff.alloc(pcap, false));
        CK(c->piece_bases.alloc(bcap, false));
        if (c->prm.all_abundance_counts) CK(c->piece_ab.alloc(bcap, false));
        HIPCK(hipMemsetAsync(c->cursors.p, 0, 8 * sizeof(uint64_t), s));
        if (!direct) HIPCK(hipMemsetAsync(c->glog_tag.p, 0xFF, c->glog_cap * sizeof(uint32_t), s));
        HIPCK(hipMemsetAsync(c->big_count.p, 0, 4 * sizeof(uint32_t), s));
        HIPCK(hipMemsetAsync(c->derr.p, 0, 4 * sizeof(uint32_t), s));
        HIPCK(hipMemsetAsync(c->dstats.p, 0, 32 * sizeof(uint64_t), s));

        CompactParams kp{};
        kp.solid_keys = c->solid_keys.p; kp.solid_cnt = c->solid_cnt.p; kp.seg_off = c->seg_off.p; kp.seg_n = c->seg_n.p;
        kp.part_list = nullptr; kp.k = c->k; kp.m = c->m; kp.log_np = c->log_np; kp.rank_bits = c->rank_bits; kp.rank = c->prm.rank;
        kp.piece_n = c->piece_n.p; kp.piece_kc = c->piece_kc.p; kp.piece_boff = c->piece_boff.p; kp.piece_bases = c->piece_bases.p;
        kp.piece_ab = c->prm.all_abundance_counts ? c->piece_ab.p : nullptr;
        kp.piece_cap = pcap; kp.bases_cap = bcap; kp.piece_cursor = c->cursors.p; kp.bases_cursor = c->cursors.p + 1;
        kp.glue_keys = nullptr; kp.glue_a = nullptr; kp.glue_b = nullptr; kp.glue_conf = nullptr; kp.glue_mask = 0;
        kp.glog_keys = c->glog_keys.p; kp.glog_tag = c->glog_tag.p; kp.glog_cap = c->glog_cap; kp.glog_cursor = c->cursors.p + 4;
        kp.jfill = direct ? c->jfill.p : nullptr; kp.jrecs = direct ? c->jrecs.p : nullptr; kp.log_jb = log_jb;
        kp.big_list = c->big_list.p; kp.big_count = c->big_count.p; kp.error = c->derr.p; kp.stats = c->dstats.p;
        kp.n_items = (uint32_t)NPL;
        // The LDS tiers over the buckets 0 .. n of `base` (their segments: base.seg_off / seg_n): one wave per bucket
        // (k_compact_wave.h), W >= 2: again one wave each with a table twice the size, then a workgroup per bucket with an LDS
        // table of TS and of 2 TS slots (k_compact.h).  Every tier hands the buckets beyond its table to the next on a list;
        // la / lb: the two lists (>= n entries each).  Returns the survivors (count, and which list holds them).
        auto lds_tiers = [&](const CompactParams& base, uint32_t n, DBuf<uint32_t>& la, DBuf<uint32_t>& lb, uint32_t& nleft, const uint32_t*& left) -> int {
            CK(c->big_count.alloc(4, false)); CK(c->big_count2.alloc(4, false));
            HIPCK(hipMemsetAsync(c->big_count.p, 0, 4 * sizeof(uint32_t), s)); HIPCK(hipMemsetAsync(c->big_count2.p, 0, 4 * sizeof(uint32_t), s));
            uint32_t nbig = 0;
            {   // tier 0
                CompactParams k0 = base; k0.part_list = nullptr; k0.n_items = n; k0.big_list = la.p; k0.big_count = c->big_count.p;
                HIPCK(hipMemsetAsync(c->cursors.p + 5, 0, sizeof(uint64_t), s));       // the bucket queue (re)starts
                CompactWaveParams wp{ k0, n, reinterpret_cast<uint32_t*>(c->cursors.p + 5) };
                const uint64_t wgrid = resident_grid(k_compact_wave<W, Cfg<W>::TSW>, CW_THREADS, 256 * 3);
                CDBG_LAUNCH((k_compact_wave<W, Cfg<W>::TSW>), std::min<uint64_t>(((uint64_t)n + CW_THREADS / 64 - 1) / (CW_THREADS / 64), wgrid), CW_THREADS, s, wp);
                HIPCK(hipStreamSynchronize(s));
                CK(read_u32(c->big_count.p, &nbig));
            }
            DBuf<uint32_t>* cur = &la; DBuf<uint32_t>* oth = &lb; uint32_t* cnt_cur = c->big_count.p; uint32_t* cnt_oth = c->big_count2.p;
            auto next_tier = [&](CompactParams& kt) { kt = base; kt.part_list = cur->p; kt.n_items = nbig; kt.big_list = oth->p; kt.big_count = cnt_oth; };
            auto flip = [&]() -> int { HIPCK(hipStreamSynchronize(s)); CK(read_u32(cnt_oth, &nbig)); std::swap(cur, oth); std::swap(cnt_cur, cnt_oth);
                                       HIPCK(hipMemsetAsync(cnt_oth, 0, 4 * sizeof(uint32_t), s)); return CDBG_OK; };
            // (one-word k-mers, round 5: a junction table of 1024 slots with 10-bit end ids -- buckets of 257 .. 512 entries.  Hostile config-3 line:
            //  compact 54.6 -> 44.8 ms, the workgroup tier is left with 24 K of its 770 K buckets; uniform config 3: 24.3 -> 23.2 ms.  CDBG_CW_TIER2 = 0: off)
            bool tier0b = Cfg<W>::TSW2 > Cfg<W>::TSW && nbig;
            if (const char* e = c->knobs.get("CDBG_CW_TIER2")) tier0b = tier0b && atoi(e) != 0;
            if (tier0b) {
                // tier 0b: the deferred buckets again one wave each, with a table twice the size (fewer waves per CU, but no
                // workgroup barriers: at the config-4 share the workgroup tier below spent 44 ms on the 129..256-entry buckets)
                CompactParams k0; next_tier(k0);
                HIPCK(hipMemsetAsync(c->cursors.p + 5, 0, sizeof(uint64_t), s));
                CompactWaveParams wp{ k0, nbig, reinterpret_cast<uint32_t*>(c->cursors.p + 5) };
                const uint64_t wgrid = resident_grid(k_compact_wave<W, Cfg<W>::TSW2>, CW_THREADS, 256 * 2);
                CDBG_LAUNCH((k_compact_wave<W, Cfg<W>::TSW2>), std::min<uint64_t>(((uint64_t)nbig + CW_THREADS / 64 - 1) / (CW_THREADS / 64), wgrid), CW_THREADS, s, wp);
                CK(flip());
            }
            if constexpr (W >= 2 && W <= 4) if (nbig && c->knobs.get("CDBG_CW_TIER3") == nullptr) {
                // tier 0c (k-mers of two to four words, round 5): once more one wave per bucket, 1024 slots (buckets of 257 .. 512 entries), before the workgroup tiers
                CompactParams k0; next_tier(k0);
                HIPCK(hipMemsetAsync(c->cursors.p + 5, 0, sizeof(uint64_t), s));
                CompactWaveParams wp{ k0, nbig, reinterpret_cast<uint32_t*>(c->cursors.p + 5) };
                const uint64_t wgrid = resident_grid(k_compact_wave<W, 1024>, CW_THREADS, 256 * 2);
                CDBG_LAUNCH((k_compact_wave<W, 1024>), std::min<uint64_t>(((uint64_t)nbig + CW_THREADS / 64 - 1) / (CW_THREADS / 64), wgrid), CW_THREADS, s, wp);
                CK(flip());
            }
            if (nbig) {                                      // tier 1: a workgroup per bucket, LDS table of TS slots
                CompactParams k1; next_tier(k1);
                CDBG_LAUNCH((k_compact<W, TS, false>), std::min<uint64_t>(nbig, PERSISTENT_GRID), COMPACT_THREADS, s, k1);
                CK(flip());
            }
            if (nbig) {                                      // tier 2: the deferred buckets with a table twice the size
                CompactParams k2; next_tier(k2);
                CDBG_LAUNCH((k_compact<W, Cfg<W>::TSK2, false>), std::min<uint64_t>(nbig, PERSISTENT_GRID), COMPACT_THREADS, s, k2);
                CK(flip());
            }
            nleft = nbig; left = cur->p;
            return CDBG_OK;
        };
        uint32_t nbig = 0; const uint32_t* left = nullptr;
        CK(c->big_list.alloc(NPL, false)); CK(c->big_list2.alloc(NPL, false));
        CK(lds_tiers(kp, (uint32_t)NPL, c->big_list, c->big_list2, nbig, left));
        c->st.n_launch_compact = NPL;
        hm.mark("compact: buffers + LDS tiers");
        CompactParams kh = kp; uint64_t n_src = NPL;         // what the HBM tier below reads: the buckets themselves, or their sub-buckets
        if (nbig && c->knobs.get("CDBG_NO_SPLIT") == nullptr) {
            // Second-level split (k_split.h): what no LDS tier could take is re-bucketed by junction into sub-buckets of ~100 entries,
            // which go through the same tiers again (the hostile config-3 line spent 76 ms walking 17 K such buckets through HBM tables)
            CK(c->split_cur.alloc(4, true));
            SplitParams sp{ kp.solid_keys, kp.solid_cnt, kp.seg_off, kp.seg_n, left, nbig, c->k, c->m, nullptr, nullptr, 0, nullptr, nullptr, 0, c->split_cur.p, c->derr.p };
            CDBG_LAUNCH(k_split_measure, (nbig + 255) / 256, 256, s, sp);
            uint64_t need[2] = {0, 0}; CK(read_u64(c->split_cur.p + 2, need, 2));
            if (need[1] >= (1ull << 31)) return fail(CDBG_E_INTERNAL, "bucket split: %llu sub-buckets exceed 31-bit ids", (unsigned long long)need[1]);
            CK(c->split_keys.alloc(2 * need[0] * W + W, false)); CK(c->split_cnt.alloc(2 * need[0] + 1, false));
            CK(c->vseg_off.alloc(need[1] + 1, false)); CK(c->vseg_n.alloc(need[1] + 1, false));
            CK(c->vlist_a.alloc(need[1] + 1, false)); CK(c->vlist_b.alloc(need[1] + 1, false));
            sp.out_keys = c->split_keys.p; sp.out_cnt = c->split_cnt.p; sp.out_cap = 2 * need[0]; sp.vseg_off = c->vseg_off.p; sp.vseg_n = c->vseg_n.p; sp.vcap = (uint32_t)need[1];
            CDBG_LAUNCH((k_split_buckets<W>), std::min<uint64_t>(nbig, PERSISTENT_GRID), SPLIT_THREADS, s, sp);
            kh = kp; kh.solid_keys = c->split_keys.p; kh.solid_cnt = c->split_cnt.p; kh.seg_off = c->vseg_off.p; kh.seg_n = c->vseg_n.p; kh.split = 1u;
            n_src = need[1];
            split_buckets_now = nbig; split_extra = std::max(split_extra, need[0]);
            CK(lds_tiers(kh, (uint32_t)need[1], c->vlist_a, c->vlist_b, nbig, left));
            hm.mark("compact: split + LDS tiers");
        }
        DBuf<uint64_t> g_keys, big_off; DBuf<uint32_t> g_cnt, g_lnk, g_aux, hb_list;
        if (nbig) {                                          // buckets with more entries than fit LDS
            std::vector<uint32_t> bl(nbig); CK(read_u32(left, bl.data(), nbig));
            std::sort(bl.begin(), bl.end());
            std::vector<uint64_t> offs(nbig + 1, 0);
            std::vector<uint32_t> h_segn;
            if (nbig > 64) { h_segn.resize(n_src); CK(read_u32(kh.seg_n, h_segn.data(), n_src)); }   // (one bulk copy, not one per bucket)
            for (uint32_t i = 0; i < nbig; ++i) {
                uint32_t e = 0;
                if (!h_segn.empty()) e = h_segn[bl[i]]; else CK(read_u32(kh.seg_n + bl[i], &e));
                offs[i + 1] = offs[i] + pow2_at_least(2 * (uint64_t)e + 16);
            }
            CK(g_keys.alloc(offs[nbig] * W, false)); CK(g_cnt.alloc(offs[nbig], false));
            CK(g_lnk.alloc(2 * offs[nbig], false)); CK(g_aux.alloc(3 * offs[nbig], false));
            CK(big_off.alloc(nbig + 1, false)); CK(hb_list.alloc(nbig, false));
            HIPCK(hipMemcpy(big_off.p, offs.data(), (nbig + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
            HIPCK(hipMemcpy(hb_list.p, bl.data(), nbig * sizeof(uint32_t), hipMemcpyHostToDevice));
            CompactParams bp = kh;
            bp.part_list = hb_list.p; bp.g_keys = g_keys.p; bp.g_cnt = g_cnt.p;
            bp.g_lnk = g_lnk.p; bp.g_aux = g_aux.p; bp.big_off = big_off.p;
            bp.n_items = nbig;
            CDBG_LAUNCH((k_compact<W, TS, true>), std::min<uint64_t>(nbig, PERSISTENT_GRID), COMPACT_THREADS, s, bp);
            HIPCK(hipStreamSynchronize(s));
        }
        uint32_t e = 0; CK(read_u32(c->derr.p, &e));
        if (e == 8 && direct && split_extra > sized_extra && attempt < 2) continue;   // join buckets sized without the split's records: once more, with them
        if (e == 8 && direct) { direct = false; --attempt; continue; }   // a join bucket overflowed (cannot happen with a sound hash): through the log instead
        if ((e == 3 || e == 5) && attempt < 2) continue;     // piece arrays / glue log too small: retry with the safe bounds (and what the split added)
        if (nbig) c->st.n_big_partitions += nbig;
        break;
    }
    c->st.n_split_buckets += split_buckets_now;              // (of the attempt that is kept)
    CK(t.stop(&c->st.ms_compact));
    hm.mark("compact: workgroup tiers");
    CK(check_device_error(c, "compact"));
    uint64_t cur[5]; CK(read_u64(c->cursors.p, cur, 5));
    c->n_pieces = cur[0]; c->n_piece_bases = cur[1]; c->n_glog = cur[4];
    c->direct_join = direct; c->join_log_jb = log_jb;
    uint64_t ks[4]; CK(read_u64(c->dstats.p, ks, 4));
#ifdef CDBG_PROFILE_PHASES
    { uint64_t ph[9]; CK(read_u64(c->dstats.p + 8, ph, 9)); fprintf(stderr, "k_compact_wave phase cycles (summed over waves): between buckets %llu | load+mins %llu | classify %llu | mutual+terminals %llu | walk1(+cycles) %llu | reserve+confirms %llu | walk2+prefix bases %llu | last bases+glog %llu | reset %llu\n",
        (unsigned long long)ph[0], (unsigned long long)ph[1], (unsigned long long)ph[2], (unsigned long long)ph[3], (unsigned long long)ph[4], (unsigned long long)ph[5], (unsigned long long)ph[6], (unsigned long long)ph[7], (unsigned long long)ph[8]); }
#endif
    c->st.n_pieces = ks[3]; c->st.n_glue_open_ends = ks[0]; c->st.n_cycles = ks[2];
    c->st.ms_total += c->st.ms_compact;
    c->stage = 2; c->joined = false;
    return CDBG_OK;
}

}  // namespace
