"""Multi-process CPU tests (gloo, world 2 / 4 / 8) of the multi-GPU data path of libcdbg: reads SHARDED over the
ranks, super-k-mer records all-to-all-v'd to the partition owners, pieces + junction log all-gathered, junction
join sharded by key hash + MAX all-reduce, owner-sharded emission.  The library is the kernel-logic simulator
(same source as the HIP build); the transport is bcalm_amd.dist.TorchTransport over gloo on host memory.  Also
bench.py's weak-scaling reductions (MAX of the time, SUM of the distinct k-mers)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib

ROOT = oracle_lib.ROOT


def _shard(text, world, rank):
    """reads r with r % world == rank, as text"""
    reads = [x for x in text.decode().split("\n") if x]
    return ("\n".join(reads[rank::world]) + "\n").encode()


def _run_sharded(api, cdist, lib, orc, text, k, amin, world, rank, steps=1, **kw):
    kw = dict(kw)
    scan_mode = kw.pop("scan_mode", None)                # CDBG_SCAN_MODE: the single-pass capped scan + region packing at test sizes
    kw.pop("expect_fallback", None)
    check_links = kw.pop("links", False)                 # cdbg_link on the sharded set, against the brute-force link oracle
    empty_rank = kw.pop("empty_rank", None)              # this rank receives no reads at all (a small input dealt out in chunks)
    part_cap = kw.pop("part_cap", None)                  # CDBG_PART_CAP: regions far too small -- spilled records are packed behind their regions
    if scan_mode:
        os.environ["CDBG_SCAN_MODE"] = scan_mode
    else:
        os.environ.pop("CDBG_SCAN_MODE", None)
    if part_cap:
        os.environ["CDBG_PART_CAP"] = part_cap
    else:
        os.environ.pop("CDBG_PART_CAP", None)
    g = api.Graph(k, amin, lib=lib, world_size=world, rank=rank, **kw)
    cdist.TorchTransport(dist).attach(g)
    if kw.get("reads_replicated"):
        g.push_text(text)                                # X0: every rank holds all the reads
    elif empty_rank is not None:
        if rank != empty_rank:
            others = [r for r in range(world) if r != empty_rank]
            g.push_text(_shard(text, world - 1, others.index(rank)))
    else:
        g.push_text(_shard(text, world, rank))
    out = None
    for i in range(steps):
        if i:
            g.reset()
        g.run()
        mine = g.unitigs(); st = g.stats(); nbytes = g.comm_bytes()
        ab = g.unitig_abundances() if kw.get("all_abundance_counts") else None
        lk = None
        if check_links and not kw.get("emit_replicated"):
            # the link table of a SHARDED unitig set (collective cdbg_link): job-wide ids, every rank the links of its own unitigs
            my_links = g.links(); first, total = g.unitig_id_base()
            lk = (first, total, my_links)
        gathered = [None] * world
        dist.all_gather_object(gathered, (mine, st["n_distinct"], st["n_solid"], st["n_occurrences"], lk))
        out = {"union": sorted((orc.canonical_unitig(s, k), int(kc)) for part in gathered for s, kc in part[0]),
               "mine": mine, "distinct": sum(p[1] for p in gathered), "solid": sum(p[2] for p in gathered),
               "occ": sum(p[3] for p in gathered), "comm_bytes": nbytes, "per_rank": [len(p[0]) for p in gathered], "ab": ab,
               "rounds": st["n_glue_rounds"]}
        if lk is not None:
            # ids are numbered rank after rank; the union of the ranks' link lists is the brute-force link set of the whole graph
            import oracle_py as op
            seqs = [s for part in gathered for s, _ in part[0]]
            firsts = [part[4][0] for part in gathered]
            ok_ids = firsts == [sum(len(q[0]) for q in gathered[:r]) for r in range(world)] and all(part[4][1] == len(seqs) for part in gathered)
            got = set()
            for part in gathered:
                first, _, links = part[4]
                for u, ls in enumerate(links):
                    for fs, v, ts in ls:
                        got.add((first + u, fs, v, ts))
            out["links_ok"] = ok_ids and got == op.links(seqs, k)
    g.close()
    return out


def _worker(rank, world, port, q, cases):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hostsim_lib
    from bcalm_amd import api, dist as cdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = hostsim_lib.load()
    orc = oracle_lib.load()
    ok = []
    for case in cases:
        k, amin, n_reads, read_len, cfg, kw = case
        kw = dict(kw)
        if isinstance(cfg, str) and cfg.startswith("@circular"):
            # isolated circular unitigs (plasmids): n_reads random circles of read_len bases each, every k-mer of a circle once
            import random
            rng = random.Random(n_reads * 1000 + read_len + k)
            circles = ["".join(rng.choice("ACGT") for _ in range(read_len)) for _ in range(n_reads)]
            text = ("\n".join(g + g[:k - 1] for g in circles) + "\n").encode()
        else:
            text = oracle_lib.read_input(cfg).encode() if isinstance(cfg, str) else orc.synth_reads(n_reads, read_len, cfg)
        exp = orc.run(text, k, amin)
        got = _run_sharded(api, cdist, lib, orc, text, k, amin, world, rank, steps=kw.pop("steps", 1), **kw)
        if kw.get("emit_replicated"):
            # every rank holds the complete set
            ok.append(oracle_lib.canonical_set(orc, got["mine"], k) == exp["unitigs"])
        else:
            # the union over the ranks is the graph: every unitig emitted by exactly one rank
            ok.append(got["union"] == exp["unitigs"])
        ok.append(got["distinct"] == exp["stats"]["distinct"] and got["solid"] == exp["stats"]["solid"] and got["occ"] == exp["stats"]["occurrences"])
        ok.append(got["comm_bytes"] > 0)
        # which glue ran: the sharded one (distributed ranking rounds > 0) unless every rank emits everything
        if kw.get("links") and not kw.get("emit_replicated"):
            ok.append(got.get("links_ok") is True)
        if "rounds" in got:
            ok.append((got["rounds"] > 0) == (not kw.get("emit_replicated") and not kw.get("expect_fallback")))
        if kw.get("all_abundance_counts"):
            # -all-abundance-counts across ranks: the vector of every unitig this rank emitted is the oracle's count of its k-mers
            solid = dict(orc.run(text, k, amin, want_solid=True)["solid"]); comp = str.maketrans("ACGT", "TGCA")
            good = len(got["ab"]) == len(got["mine"])
            for (s_, kc), a in zip(got["mine"], got["ab"]):
                want = [solid[min(s_[i:i + k], s_[i:i + k].translate(comp)[::-1])] for i in range(len(s_) - k + 1)]
                good = good and a == want and sum(a) == kc
            ok.append(good)
    # bench.py's reductions for N > 1
    t = torch.tensor([0.5 + rank], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok.append(float(t.item()) == 0.5 + world - 1)
    q.put((rank, ok))
    dist.destroy_process_group()


def _launch(world, cases, port_base, timeout):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = port_base + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, cases)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=timeout) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert len(res) == world
    for rank, ok in res:
        assert all(ok), (rank, ok)


def test_two_rank_gloo():
    """world 2: one-, two- and four-word k-mers, automatic and forced partition counts, repeated steps (reset keeps the
    buffers), emit_replicated (every rank ends with the whole graph: what the multi-GPU CLI's rank 0 writes),
    -all-abundance-counts across the ranks"""
    _launch(2, [(31, 2, 300, 150, 3, {"steps": 2}), (31, 1, 150, 150, 3, {"log2_partitions": 6}),
                (55, 2, 200, 150, 4, {"log2_partitions": 5}), (127, 1, 40, 600, 5, {"log2_partitions": 4}),
                (31, 2, 200, 150, 3, {"emit_replicated": True}),
                (31, 2, 250, 150, 3, {"all_abundance_counts": True}), (55, 1, 100, 150, 4, {"all_abundance_counts": True, "emit_replicated": True}),
                (31, 2, 300, 150, 3, {"reads_replicated": True}), (31, 2, 300, 150, 3, {"scan_mode": "capped", "log2_partitions": 6}),
                (31, 2, 300, 150, 3, {"scan_mode": "capped", "log2_partitions": 6, "part_cap": "3"}), (55, 2, 150, 150, 4, {"scan_mode": "capped", "part_cap": "1"}),
                (55, 2, 200, 150, 4, {"scan_mode": "capped", "reads_replicated": True}), (31, 2, 200, 150, 3, {"empty_rank": 1}),
                # reads replicated through the capped scan with 2^11 partitions per rank: every rank defers half of its partitions' records (k_place, two count slices)
                (31, 2, 300, 150, 3, {"scan_mode": "capped", "reads_replicated": True, "log2_partitions": 12, "part_cap": "2"}),
                (30, 2, 250, 150, 3, {"links": True}), (64, 1, 100, 300, 5, {"log2_partitions": 4, "links": True}),
                (31, 2, 300, 150, 3, {"links": True}), (9, 1, 0, 0, "pufferize_refs", {"log2_partitions": 4, "minimizer_size": 4, "links": True}),
                (8, 1, 0, 0, "even_k8", {"log2_partitions": 3, "minimizer_size": 4, "links": True}),
                # closed chains across ranks (example/circular_unitigs_unittests): a ranking round that finishes nothing -> the unfinished
                # states are gathered, every rank cuts the same junction, the sharded ranking starts again (no replicated fallback)
                (7, 1, 0, 0, "circ_test1", {"log2_partitions": 3, "minimizer_size": 3}), (7, 1, 0, 0, "circ_test1", {"log2_partitions": 5, "minimizer_size": 4}),
                (7, 1, 0, 0, "circ_test2", {"log2_partitions": 3, "minimizer_size": 3}), (9, 1, 0, 0, "pufferize_refs", {"log2_partitions": 4, "minimizer_size": 4})], 29500, 400)


def test_four_rank_gloo():
    _launch(4, [(31, 2, 400, 150, 3, {"links": True}), (55, 1, 160, 150, 4, {"log2_partitions": 7, "links": True}), (127, 2, 80, 500, 5, {"log2_partitions": 5}),
                # three plasmid-like circles of 1500 bp next to ordinary reads' worth of chains: cut in place, ranking restarted
                (31, 1, 3, 1500, "@circular", {"log2_partitions": 6}), (55, 1, 2, 900, "@circular", {"log2_partitions": 5}),
                (31, 2, 400, 150, 3, {"reads_replicated": True}), (31, 2, 300, 150, 3, {"scan_mode": "capped", "empty_rank": 3}),
                (31, 2, 400, 150, 3, {"scan_mode": "capped", "reads_replicated": True, "log2_partitions": 12})], 31500, 300)


def test_eight_rank_gloo():
    """the world size of the full node: partitions split eight ways (3 rank bits), eight-way sharded join"""
    _launch(8, [(31, 2, 400, 150, 3, {"log2_partitions": 7}), (55, 1, 160, 150, 4, {"log2_partitions": 7})], 33500, 400)


def test_multi_rank_needs_a_transport():
    """a context with world_size > 1 and no transport refuses to count (no silent single-rank result)"""
    import hostsim_lib
    from bcalm_amd import api
    lib = hostsim_lib.load()
    g = api.Graph(31, 2, lib=lib, world_size=2, rank=0)
    g.push_text(b"ACGTACGTTGCATGCATGCATTTGACCAGTACCAGGGATTTACCA\n")
    with pytest.raises(api.CdbgError) as e:
        g.count()
    assert "transport" in str(e.value)
    g.close()
