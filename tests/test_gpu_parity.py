"""GPU parity tests proper (-m gpu): the HIP library through its C ABI against the CPU
oracle, bit-exact on (k-mer, count) sets and on canonical unitig sets (sequence + KC)."""
import json
import os
import random

import pytest

import oracle_lib
from parity import assert_parity, run_graph

pytestmark = pytest.mark.gpu
ROOT = oracle_lib.ROOT
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))


@pytest.fixture(scope="module")
def hip():
    import bcalm_amd
    return bcalm_amd.load()          # fails loudly when the extension is missing


def _case(key):
    name, k, amin = key.split("/")
    return name, int(k), int(amin)


@pytest.mark.parametrize("key", sorted(GOLD))
@pytest.mark.parametrize("log_np,m", [(-1, 0), (0, 0), (6, 5), (10, 4)])
def test_golden_parity(oracle, hip, key, log_np, m):
    name, k, amin = _case(key)
    text = oracle_lib.read_input(name)
    got = assert_parity(oracle, hip, text, k, amin, log2_partitions=log_np, minimizer_size=min(m, k - 1) if m else 0)
    assert oracle_lib.canonical_set(oracle, got["unitigs"], k) == [tuple(u) for u in GOLD[key]["unitigs"]]
    assert oracle_lib.solid_sha256(got["solid"]) == GOLD[key]["solid"]["sha256"]


@pytest.mark.parametrize("seed", range(12))
def test_random_low_complexity(oracle, hip, seed):
    rng = random.Random(9000 + seed)
    k = rng.choice([5, 7, 9, 11, 13, 33, 65])
    g = "".join(rng.choice("AT" if seed % 2 else "ACG") for _ in range(rng.randrange(60, 1500)))
    reads = []
    for _ in range(rng.randrange(3, 60)):
        L = rng.randrange(1, len(g)); s = rng.randrange(0, len(g) - L + 1)
        reads.append(g[s:s + L])
    text = "\n".join(reads) + "\n"
    assert_parity(oracle, hip, text, k, rng.choice([1, 1, 2]), log2_partitions=rng.choice([0, 2, 5, 8]),
                  minimizer_size=rng.choice([2, 3, 4]))


def test_generator_bit_exact(oracle, hip):
    import bcalm_amd
    g = bcalm_amd.Graph(31, 2, lib=hip)
    g.generate_reads(1000, 150, 3, first_read=17, total_reads=5000)
    got = g.read_text(0, 1000 * 151)
    g.close()
    assert got == oracle.synth_reads(1000, 150, 3, first=17, total=5000)


@pytest.mark.parametrize("k,amin,n_reads,read_len,cfg", [(31, 2, 60000, 150, 3), (55, 2, 30000, 150, 4), (127, 2, 4000, 1000, 5), (21, 1, 20000, 100, 2)])
def test_synthetic_parity(oracle, hip, k, amin, n_reads, read_len, cfg):
    """BASELINE configs 3/4/5 shapes at sizes the oracle finishes in seconds"""
    import bcalm_amd
    text = oracle.synth_reads(n_reads, read_len, cfg)
    exp = oracle.run(text, k, amin)
    g = bcalm_amd.Graph(k, amin, lib=hip)
    g.generate_reads(n_reads, read_len, cfg)
    g.run()
    st = g.stats()
    canon = oracle_lib.canonical_set(oracle, g.unitigs(), k)
    g.close()
    assert st["n_distinct"] == exp["stats"]["distinct"]
    assert st["n_solid"] == exp["stats"]["solid"]
    assert st["n_occurrences"] == exp["stats"]["occurrences"]
    assert canon == exp["unitigs"]


def test_determinism(oracle, hip):
    """same input, repeated runs and different partitionings -> identical canonical set"""
    text = oracle.synth_reads(20000, 150, 3)
    ref = None
    for log_np in (-1, 8, 12, -1):
        got = run_graph(hip, text, 31, 2, log2_partitions=log_np)
        canon = oracle_lib.canonical_set(oracle, got["unitigs"], 31)
        if ref is None:
            ref = canon
        assert canon == ref


def test_unitig_properties_large(hip):
    """size-independent properties at a size the oracle would not finish quickly:
    every solid k-mer appears exactly once over all unitigs, KC sums to solid occurrences"""
    import bcalm_amd
    k = 31
    g = bcalm_amd.Graph(k, 2, lib=hip)
    g.generate_reads(400000, 150, 3)
    g.count()
    solid = g.solid_kmers()
    g.compact(); g.glue()
    ut = g.unitigs()
    st = g.stats()
    g.close()
    comp = str.maketrans("ACGT", "TGCA")
    seen = set()
    for s, kc in ut:
        for i in range(len(s) - k + 1):
            x = s[i:i + k]; r = x.translate(comp)[::-1]
            c = x if x <= r else r
            assert c not in seen
            seen.add(c)
    assert len(seen) == len(solid) == st["n_solid"]
    assert seen == {x for x, _ in solid}
    assert sum(kc for _, kc in ut) == sum(c for _, c in solid)
    assert sum(len(s) for s, _ in ut) == st["unitig_bases"]


def test_no_device_fallback_symbols(hip):
    import bcalm_amd
    for sym in bcalm_amd.EXPORTS:
        assert hasattr(hip, sym)


def test_exchange_path_single_rank_nccl(oracle, hip):
    """the multi-GPU glue exchange (RCCL all-gather + merge + glue on the union) with a 1-rank
    'nccl' group: exercises the device-tensor / D2D / merge-kernel path that N>1 uses"""
    import torch
    import torch.distributed as dist
    import bcalm_amd
    from bcalm_amd import dist as cdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        text = oracle.synth_reads(30000, 150, 3)
        exp = oracle.run(text, 31, 2)
        g = bcalm_amd.Graph(31, 2, lib=hip, world_size=1, rank=0)
        g.push_text(text); g.count(); g.compact()
        info = cdist.exchange_glue(g, dist, torch.device("cuda", 0), 1)
        # the sharded-join leg (dist.py runs it for world > 1): join, int32 link array through an RCCL MAX all-reduce
        n = g.glue_join()
        links = torch.empty(n, dtype=torch.int32, device="cuda:0")
        g.glue_links_export(links.data_ptr(), n * 4)
        dist.all_reduce(links, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        assert int((links >= 0).sum().item()) > 0 and int(links.max().item()) < n
        g.glue_links_import(links.data_ptr(), n * 4)
        g.glue()
        canon = oracle_lib.canonical_set(oracle, g.unitigs(), 31)
        g.close()
        assert info["glue_records"] > 0
        assert canon == exp["unitigs"]
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("part_cap", [None, "64"])
def test_capped_single_pass_scan_gpu(oracle, hip, part_cap, monkeypatch):
    """the large-input scan path (single pass, fixed-capacity partition regions, spill repair)"""
    monkeypatch.setenv("CDBG_SCAN_MODE", "capped")
    if part_cap:
        monkeypatch.setenv("CDBG_PART_CAP", part_cap)
    text = oracle.synth_reads(40000, 150, 3)
    assert_parity(oracle, hip, text, 31, 2, log2_partitions=10)


def test_config2_genome_shape(oracle, hip):
    """BASELINE config 2 shape: one 4.64 Mbp sequence (E. coli MG1655 length; the FASTA itself is not
    available offline, so a seeded synthetic genome with planted direct and inverted repeats stands in),
    k=31, abundance-min 1: every k-mer distinct -> exercises the multi-pass LDS counting path;
    bit-exact unitig set vs the oracle."""
    import bcalm_amd
    from parity import config2_genome
    text = config2_genome(oracle)
    exp = oracle.run(text, 31, 1)
    g = bcalm_amd.Graph(31, 1, lib=hip)
    g.push_text(text); g.run()
    st = g.stats()
    canon = oracle_lib.canonical_set(oracle, g.unitigs(), 31)
    g.close()
    assert st["n_distinct"] == exp["stats"]["distinct"] == st["n_solid"]
    assert canon == exp["unitigs"]
    assert exp["stats"]["unitigs"] > 50                  # the repeats really branch the graph


def test_streaming_ingest_roundtrip_gpu(hip):
    """pinned double-buffered H2D ingest on the copy stream (several staging buffers, growing device text)"""
    from test_hostsim_pipeline import _ingest_roundtrip
    _ingest_roundtrip(hip)


def test_config5_shape_long_reads_k127(hip):
    """BASELINE config 5 shape (1 kbp reads, k = 127, four-word k-mers) at a size where every compaction tier
    runs with thousands of persistent workgroups each (regression: the chunk slack of the piece / base / glue-log
    arrays was sized for two launches per stage, the third overflowed the glue log).  Properties instead of the
    oracle: each solid k-mer exactly once over the unitigs, KC conserved."""
    import bcalm_amd
    k = 127
    g = bcalm_amd.Graph(k, 2, lib=hip, log2_partitions=12)   # ~500 entries per bucket: all three compaction tiers get thousands of buckets
    g.generate_reads(60000, 1000, 5)
    g.count()
    solid = g.solid_kmers()
    g.compact(); g.glue()
    ut = g.unitigs()
    st = g.stats()
    g.close()
    comp = str.maketrans("ACGT", "TGCA")
    seen = set()
    for s, kc in ut:
        for i in range(len(s) - k + 1):
            x = s[i:i + k]; r = x.translate(comp)[::-1]
            c = x if x <= r else r
            assert c not in seen
            seen.add(c)
    assert len(seen) == len(solid) == st["n_solid"]
    assert sum(kc for _, kc in ut) == sum(c for _, c in solid)
    # the size that overflowed: ~450 entries per bucket over 16384 buckets, so each of the three compaction
    # launches runs its full complement of persistent workgroups while the traveller-based bounds are tiny
    g = bcalm_amd.Graph(k, 2, lib=hip, log2_partitions=14)
    g.generate_reads(200000, 1000, 5)
    g.run()
    st = g.stats()
    ut = g.unitigs()
    g.close()
    assert st["n_big_partitions"] > 3000
    assert sum(len(s) - k + 1 for s, _ in ut) == st["n_solid"] and len(ut) == st["n_unitigs"]


def test_parity_one_million_reads(oracle, oracle_1m, hip):
    """bit-exact against the oracle at the largest size the oracle finishes in seconds (1 M x 150 bp, 36 M distinct
    k-mers): single-pass capped scan, persistent kernels with thousands of workgroups, chunked output reservations"""
    import bcalm_amd
    text, exp = oracle_1m
    g = bcalm_amd.Graph(31, 2, lib=hip)
    g.push_text(text); g.run()
    got = oracle_lib.canonical_set(oracle, g.unitigs(), 31)
    st = g.stats(); g.close()
    assert st["n_distinct"] == exp["stats"]["distinct"] and st["n_solid"] == exp["stats"]["solid"]
    assert got == exp["unitigs"]


def test_two_rank_flow_on_one_device(oracle, hip):
    """the complete N = 2 data path on the real device, ranks emulated one after the other on GPU 0 (RCCL refuses two
    ranks on one GPU): partitions split two ways, glue records exchanged through cdbg_exchange_* with device buffers,
    junction join sharded by key hash, link arrays combined with an element-wise MAX, rank + emit on both; the two
    ranks must end with the same unitig set as the single-rank run and the oracle"""
    import torch
    import bcalm_amd
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0); torch.zeros(1, device=dev)     # torch's HIP runtime initialises before libcdbg's first call here
    text = oracle.synth_reads(300000, 150, 3)
    exp = oracle.run(text, 31, 2)
    world = 2
    gs = []
    for r in range(world):
        g = bcalm_amd.Graph(31, 2, lib=hip, world_size=world, rank=r)
        g.push_text(text); g.count(); g.compact()
        gs.append(g)
    sizes = [g.exchange_sizes() for g in gs]
    kinds = [(0, 4, 0), (1, 8, 0), (2, 8, 0), (3, 1, 1), (4, 8, 2), (5, 4, 2)]        # (export kind, bytes per item, which size)
    bufs = []                                                                          # bufs[r][kind] = device tensor
    for r, g in enumerate(gs):
        row = []
        for kind, item, which in kinds:
            nb = max(int(sizes[r][which]) * item, 16)
            t = torch.empty(nb, dtype=torch.uint8, device=dev)
            g.exchange_export(kind, t.data_ptr(), nb)
            row.append(t)
        bufs.append(row)
    totals = [sum(int(sizes[r][j]) for r in range(world)) for j in range(3)]
    links = []
    for g in gs:
        g.exchange_begin(*totals)
        for r in range(world):
            g.exchange_add(int(sizes[r][0]), int(sizes[r][1]), int(sizes[r][2]), [t.data_ptr() for t in bufs[r]])
        g.exchange_end()
        n = g.glue_join()
        t = torch.empty(n, dtype=torch.int32, device=dev)
        g.glue_links_export(t.data_ptr(), n * 4)
        links.append(t)
    assert links[0].numel() == links[1].numel() == 2 * totals[0]
    both = (links[0] >= 0) & (links[1] >= 0)
    assert int(both.sum().item()) == 0, "an end was joined by both ranks"
    assert int((links[0] >= 0).sum().item()) > 0 and int((links[1] >= 0).sum().item()) > 0
    merged = torch.maximum(links[0], links[1])
    torch.cuda.synchronize()
    sets = []
    for g in gs:
        g.glue_links_import(merged.data_ptr(), merged.numel() * 4)
        g.glue()
        sets.append(oracle_lib.canonical_set(oracle, g.unitigs(), 31))
        g.close()
    assert sets[0] == sets[1] == exp["unitigs"]
    # the packed exchange (2-bit bases, no offsets on the wire: what bcalm_amd/dist.py ships) through the same emulation
    gs = []
    for r in range(world):
        g = bcalm_amd.Graph(31, 2, lib=hip, world_size=world, rank=r)
        g.push_text(text); g.count(); g.compact()
        gs.append(g)
    psizes = [g.exchange_sizes_packed() for g in gs]
    assert all(ps[3] * 3 < ps[1] for ps in psizes)            # packed bytes ~ bases / 4 (+ padding)
    pbufs = []
    for r, g in enumerate(gs):
        row = []
        for kind, nb in ((0, psizes[r][0] * 4), (1, psizes[r][0] * 8), (None, psizes[r][3]), (4, psizes[r][2] * 8), (5, psizes[r][2] * 4)):
            t = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
            if kind is None:
                g.exchange_export_packed(t.data_ptr(), t.numel())
            else:
                g.exchange_export(kind, t.data_ptr(), t.numel())
            row.append(t)
        pbufs.append(row)
    ptot = [sum(psizes[r][j] for r in range(world)) for j in range(3)]
    g = gs[0]
    g.exchange_begin(*ptot)
    for r in range(world):
        g.exchange_add_packed(psizes[r][0], psizes[r][1], psizes[r][3], psizes[r][2], [t.data_ptr() for t in pbufs[r]])
    g.exchange_end()
    g.glue()                                                  # unsharded join on the union
    assert oracle_lib.canonical_set(oracle, g.unitigs(), 31) == exp["unitigs"]
    for g in gs:
        g.close()


def test_one_giant_partition_statistics(oracle, hip):
    """the whole input in ONE partition (log2_partitions = 0, abundance-min 1): 300 K solid k-mers go through the
    HBM-table fallbacks of count and compact; found by bench_micro/fuzz_gpu.py -- the per-partition statistics were
    reduced in 16-bit fields and n_solid (which sizes the later stages) came out short"""
    rng = random.Random(5)
    g = "".join(rng.choice("ACGT") for _ in range(300000))
    text = g + "\n" + g[1000:5000] + "\n"
    for k in (21, 61):
        exp = oracle.run(text, k, 1)
        got = assert_parity(oracle, hip, text, k, 1, log2_partitions=0)
        assert got["stats"]["n_solid"] == exp["stats"]["solid"] == got["stats"]["n_distinct"]


def test_repartition_when_buckets_overflow(oracle, hip):
    """abundance-min 1 keeps every k-mer: with the partition count chosen for the count table the compaction buckets
    would hold ~900 entries and fall back to HBM tables (measured 716 ms instead of 40 ms for 3 M reads); cdbg_count
    looks at the exact solid count and counts again with more partitions.  Parity + the fallback must stay rare."""
    import bcalm_amd
    text = oracle.synth_reads(300000, 150, 3)
    exp = oracle.run(text, 31, 1)
    g = bcalm_amd.Graph(31, 1, lib=hip)
    g.push_text(text); g.run()
    got = oracle_lib.canonical_set(oracle, g.unitigs(), 31)
    st = g.stats(); g.close()
    assert got == exp["unitigs"] and st["n_solid"] == exp["stats"]["solid"] == st["n_distinct"]
    assert (st["n_solid"] + st["n_solid_travellers"]) / (1 << st["log2_partitions"]) <= 300
    assert st["n_big_partitions"] < 50
