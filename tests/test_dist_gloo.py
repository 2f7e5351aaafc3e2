"""world_size-2 CPU test (gloo) of the multi-process path bench.py uses for N > 1:
each rank runs the full path on its own read set, then MAX-reduces the time and
SUM-reduces the distinct k-mer count; plus the minimizer-sharded counting mode
(world_size/rank in cdbg_params) checked for an exact 2-way split of the k-mer set."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib

ROOT = oracle_lib.ROOT


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hostsim_lib
    from bcalm_amd import api
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = hostsim_lib.load()
    orc = oracle_lib.load()
    # (a) bench.py's N>1 mode: disjoint read sets, no data-path collective
    text = orc.synth_reads(120, 150, 3 + 16 * rank)
    g = api.Graph(31, 2, lib=lib)
    g.push_text(text); g.run()
    st = g.stats(); canon = oracle_lib.canonical_set(orc, g.unitigs(), 31); g.close()
    exp = orc.run(text, 31, 2)
    ok = canon == exp["unitigs"]
    t = torch.tensor([0.5 + rank], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n = torch.tensor([st["n_distinct"]], dtype=torch.int64); dist.all_reduce(n, op=dist.ReduceOp.SUM)
    # (b) minimizer-sharded counting of ONE read set: the two ranks' solid sets partition the oracle's
    shared = orc.synth_reads(150, 150, 3)
    g = api.Graph(31, 2, lib=lib, log2_partitions=5, world_size=world, rank=rank)
    g.push_text(shared); g.count()
    mine = g.solid_kmers(); g.close()
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    full = sorted(x for part in gathered for x in part)
    exp2 = orc.run(shared, 31, 2, want_solid=True)
    # (c) single-graph mode: sharded count + compact, all-gather of glue records, glue on the union
    from bcalm_amd import dist as cdist
    g = api.Graph(31, 2, lib=lib, log2_partitions=6, world_size=world, rank=rank)
    g.push_text(shared); g.count(); g.compact()
    info = cdist.exchange_glue(g, dist, torch.device("cpu"), 1)
    g.glue()
    canon2 = oracle_lib.canonical_set(orc, g.unitigs(), 31); g.close()
    ok_graph = canon2 == exp2["unitigs"] and info["glue_records"] > 0 and info["link_bytes_reduced"] > 0
    # (d) the same with the junction join NOT sharded (every rank joins all glue records inside cdbg_glue), and
    #     a larger graph through the sharded join with repeated steps (reset keeps every buffer)
    g = api.Graph(31, 2, lib=lib, log2_partitions=6, world_size=world, rank=rank)
    g.push_text(shared); g.count(); g.compact()
    cdist.exchange_glue(g, dist, torch.device("cpu"), 1, sharded_join=False)
    g.glue()
    ok_graph = ok_graph and oracle_lib.canonical_set(orc, g.unitigs(), 31) == exp2["unitigs"]
    for _ in range(2):
        g.reset(); g.count(); g.compact()
        cdist.exchange_glue(g, dist, torch.device("cpu"), 1)
        g.glue()
        ok_graph = ok_graph and oracle_lib.canonical_set(orc, g.unitigs(), 31) == exp2["unitigs"]
    g.close()
    big = orc.synth_reads(600, 150, 4)
    exp3 = orc.run(big, 21, 1)
    g = api.Graph(21, 1, lib=lib, log2_partitions=8, world_size=world, rank=rank)
    g.push_text(big); g.count(); g.compact()
    cdist.exchange_glue(g, dist, torch.device("cpu"), 1)
    g.glue()
    ok_graph = ok_graph and oracle_lib.canonical_set(orc, g.unitigs(), 21) == exp3["unitigs"]
    g.close()
    q.put((rank, ok, float(t.item()), int(n.item()), st["n_distinct"], full == exp2["solid"], len(mine), ok_graph))
    dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "per-rank parity failed"
    assert res[0][2] == res[1][2] == 1.5                      # MAX over ranks
    assert res[0][3] == res[1][3] == res[0][4] + res[1][4]    # SUM of distinct k-mers
    assert all(r[5] for r in res), "sharded k-mer sets do not partition the oracle's set"
    assert res[0][6] > 0 and res[1][6] > 0
    assert all(r[7] for r in res), "sharded single-graph mode (all-gather + merge + glue) differs from the oracle"


def _worker_graph(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hostsim_lib
    from bcalm_amd import api, dist as cdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = hostsim_lib.load()
    orc = oracle_lib.load()
    ok = True
    for k, amin, n_reads, cfg in ((31, 2, 200, 3), (55, 1, 120, 4)):
        text = orc.synth_reads(n_reads, 150, cfg)
        exp = orc.run(text, k, amin)
        g = api.Graph(k, amin, lib=lib, log2_partitions=7, world_size=world, rank=rank)
        g.push_text(text); g.count(); g.compact()
        cdist.exchange_glue(g, dist, torch.device("cpu"), 1 if k <= 31 else 2)
        g.glue()
        ok = ok and oracle_lib.canonical_set(orc, g.unitigs(), k) == exp["unitigs"]
        g.close()
    q.put((rank, ok))
    dist.destroy_process_group()


def test_eight_rank_sharded_join_gloo():
    """the world size of the full node: 8 ranks, partitions split eight ways (3 rank bits), eight-way sharded join"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_graph, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert len(res) == 8 and all(r[1] for r in res), res


def test_four_rank_sharded_join_gloo():
    """4 ranks: partitions split four ways, glue records all-gathered, junction join sharded by key hash and
    combined with the MAX all-reduce; one- and two-word k-mers"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_graph, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=400) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
