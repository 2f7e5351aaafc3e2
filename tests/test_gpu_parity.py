"""GPU parity tests proper (-m gpu): the HIP library through its C ABI against the CPU
oracle, bit-exact on (k-mer, count) sets and on canonical unitig sets (sequence + KC)."""
import json
import os
import random
import sys

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


@pytest.mark.parametrize("seed", range(22))
def test_random_low_complexity(oracle, hip, seed):
    """two- and three-letter genomes; seeds >= 12: even k (k-mers that are their own reverse complement, .md:30)"""
    rng = random.Random(9000 + seed)
    k = rng.choice([5, 7, 9, 11, 13, 33, 65] if seed < 12 else [4, 6, 8, 10, 12, 32, 34, 64, 66, 96])
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


@pytest.mark.parametrize("k,amin,n_reads,read_len,cfg", [(31, 2, 60000, 150, 3), (55, 2, 30000, 150, 4), (127, 2, 4000, 1000, 5), (21, 1, 20000, 100, 2),
                                                         (30, 2, 40000, 150, 3), (32, 2, 30000, 150, 4), (64, 2, 8000, 500, 5), (77, 2, 4000, 1000, 5),
                                                         (96, 2, 4000, 1000, 5), (126, 1, 2000, 1000, 5)])
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


@pytest.mark.parametrize("k,m", [(31, 16), (31, 15), (32, 16), (31, 12), (55, 16), (63, 8)])
def test_scan_window_specialisations_gpu(oracle, hip, k, m):
    """the compile-time minimizer windows of k_scan_fast (15 keys, 16 keys, the doubling window, 39 keys) and the run-time one,
    with the minimizer length named by the caller"""
    import bcalm_amd
    cfg = 3 if k <= 32 else 4
    text = oracle.synth_reads(30000, 150, cfg)
    exp = oracle.run(text, k, 2)
    g = bcalm_amd.Graph(k, 2, lib=hip, minimizer_size=m)
    g.generate_reads(30000, 150, cfg)
    g.run()
    st = g.stats()
    canon = oracle_lib.canonical_set(oracle, g.unitigs(), k)
    g.close()
    assert st["minimizer_size"] == m
    assert st["n_distinct"] == exp["stats"]["distinct"] and st["n_solid"] == exp["stats"]["solid"]
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


def test_multi_gpu_path_single_rank_rccl(oracle, hip, monkeypatch):
    """the multi-GPU code path of the library with a ONE-rank RCCL communicator (CDBG_FORCE_MULTI): librccl bound at run
    time, ncclCommInitRank, the all-to-all-v's through RCCL (grouped ncclSend / ncclRecv), record exchange + merge kernels, the
    sharded glue (junction records, pairs, ranking rounds, pieces); result against the oracle, repeated steps"""
    import torch.distributed as dist
    import bcalm_amd
    from bcalm_amd import dist as cdist
    monkeypatch.setenv("CDBG_FORCE_MULTI", "1")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)          # (carries the ncclUniqueId only)
    try:
        for k, n, L, cfg in ((31, 30000, 150, 3), (55, 10000, 150, 4)):
            text = oracle.synth_reads(n, L, cfg)
            exp = oracle.run(text, k, 2)
            g = bcalm_amd.Graph(k, 2, lib=hip, world_size=1, rank=0)
            cdist.init_rccl(g, dist)
            g.push_text(text)
            for step in range(2):
                if step:
                    g.reset()
                g.run()
                st = g.stats()
                assert oracle_lib.canonical_set(oracle, g.unitigs(), k) == exp["unitigs"]
                assert st["n_distinct"] == exp["stats"]["distinct"] and st["ms_exchange"] > 0
            g.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,amin,n_reads,read_len,cfg,kw", [
    (2, 31, 2, 300000, 150, 3, {}), (4, 31, 1, 60000, 150, 3, {"log2_partitions": 12}), (2, 31, 2, 20000, 150, 3, {"links": True}), (4, 55, 2, 12000, 150, 4, {"links": True}),
    (2, 55, 2, 100000, 150, 4, {}), (2, 127, 2, 8000, 1000, 5, {}), (2, 31, 2, 100000, 150, 3, {"emit_replicated": True}),
    (2, 31, 2, 60000, 150, 3, {"all_abundance_counts": True}),
    (2, 31, 2, 200000, 150, 3, {"reads_replicated": True}), (4, 32, 2, 60000, 150, 4, {"reads_replicated": True}),
    (2, 31, 2, 300000, 150, 3, {"scan_mode": "capped"}), (4, 77, 2, 6000, 1000, 5, {"scan_mode": "capped"}),
    (2, 31, 2, 300000, 150, 3, {"scan_mode": "capped", "part_cap": "12"}), (4, 55, 2, 60000, 150, 4, {"scan_mode": "capped", "part_cap": "2"}),
    (2, 31, 1, 40, 5000, "circular", {}), (4, 55, 1, 25, 3000, "circular", {}),
    # reads replicated (X0) through the capped scan: every rank defers half of ITS partitions' records (k_place beside the count of the other half)
    (2, 31, 2, 300000, 150, 3, {"reads_replicated": True, "scan_mode": "capped", "log2_partitions": 13, "deferred": 2}),
    (4, 31, 2, 200000, 150, 3, {"reads_replicated": True, "scan_mode": "capped", "log2_partitions": 13, "part_cap": "380", "deferred": 2})])
def test_multi_rank_flow_on_one_device(oracle, hip, world, k, amin, n_reads, read_len, cfg, kw, monkeypatch):
    """the complete N-rank data path on the real device: N contexts on GPU 0 driven by N host threads, reads sharded,
    records / pieces / junction log / partner ids moved by an in-process loop-back transport (tests/loopback.py: RCCL
    refuses two ranks on one GPU).  The union of the ranks' unitigs must be the oracle's set, each unitig exactly once."""
    import threading
    import bcalm_amd
    from loopback import hip_loopback
    kw = dict(kw)
    if kw.pop("scan_mode", None):                # sharded reads through the single-pass capped scan + region packing
        monkeypatch.setenv("CDBG_SCAN_MODE", "capped")
    if kw.get("part_cap"):                       # regions far too small: the spilled records are packed behind their regions (k_pack_spills)
        monkeypatch.setenv("CDBG_PART_CAP", kw.pop("part_cap"))
    want_links = kw.pop("links", False)           # cdbg_link on the sharded set: job-wide ids, the union of the ranks' links == brute force
    want_slices = kw.pop("deferred", None)        # deferred record placement must have run on every rank
    if cfg == "circular":
        # isolated circular unitigs (plasmids) among the ranks: closed chains, cut in place by the sharded glue (k_dglue.h)
        rng = random.Random(n_reads + read_len)
        circles = ["".join(rng.choice("ACGT") for _ in range(read_len)) for _ in range(n_reads)]
        text = ("\n".join(g + g[:k - 1] for g in circles) + "\n").encode()
    else:
        text = oracle.synth_reads(n_reads, read_len, cfg)
    exp = oracle.run(text, k, amin)
    reads = [x for x in text.decode().split("\n") if x]
    hub = hip_loopback(world)
    out = [None] * world
    def rank_main(r):
        try:
            g = bcalm_amd.Graph(k, amin, lib=hip, world_size=world, rank=r, **kw)
            ep = hub.endpoint(r); ep.attach(g)
            g.push_text(text if kw.get("reads_replicated") else ("\n".join(reads[r::world]) + "\n").encode())
            g.run()
            lk = (g.links(), g.unitig_id_base()) if want_links else None       # (collective: every rank's thread calls it)
            out[r] = (g.unitigs(), g.stats(), g.comm_bytes(), ep.error, g.unitig_abundances() if kw.get("all_abundance_counts") else None, lk)
            g.close()
        except Exception as e:                   # noqa: BLE001
            out[r] = e
            hub.barrier.abort()
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    for r in range(world):
        assert not isinstance(out[r], Exception), out[r]
        assert out[r] is not None and out[r][3] is None, out[r]
    if kw.get("emit_replicated"):
        for r in range(world):
            assert oracle_lib.canonical_set(oracle, out[r][0], k) == exp["unitigs"]
    else:
        union = sorted((oracle.canonical_unitig(s, k), int(kc)) for r in range(world) for s, kc in out[r][0])
        assert union == exp["unitigs"]
        assert cfg == "circular" or all(len(out[r][0]) > 0 for r in range(world))   # (a circle is cut at its smallest piece id: rank 0 owns most of those heads)
        assert all(out[r][1]["n_glue_rounds"] > 0 for r in range(world))        # the sharded glue ran (k_dglue.h), not the replicated fallback
    assert sum(out[r][1]["n_distinct"] for r in range(world)) == exp["stats"]["distinct"]
    assert sum(out[r][1]["n_solid"] for r in range(world)) == exp["stats"]["solid"]
    assert all(out[r][2] > 0 for r in range(world))
    if want_slices:
        assert all(out[r][1]["count_slices"] == want_slices and out[r][1]["n_deferred_records"] > 0 for r in range(world)), [out[r][1]["count_slices"] for r in range(world)]
    if want_links:
        sys.path.insert(0, os.path.join(oracle_lib.ROOT, "oracle"))
        import oracle_py as op
        seqs = [s for r in range(world) for s, _ in out[r][0]]
        assert [out[r][5][1][0] for r in range(world)] == [sum(len(out[q][0]) for q in range(r)) for r in range(world)]
        got = {(out[r][5][1][0] + u, fs, v, ts) for r in range(world) for u, ls in enumerate(out[r][5][0]) for fs, v, ts in ls}
        assert got == op.links(seqs, k)
    if kw.get("all_abundance_counts"):           # -all-abundance-counts across ranks: every k-mer of every unitig carries the oracle's count
        solid = dict(oracle.run(text, k, amin, want_solid=True)["solid"]); comp = str.maketrans("ACGT", "TGCA")
        for r in range(world):
            assert len(out[r][4]) == len(out[r][0])
            for (s, kc), a in zip(out[r][0], out[r][4]):
                assert a == [solid[min(s[i:i + k], s[i:i + k].translate(comp)[::-1])] for i in range(len(s) - k + 1)] and sum(a) == kc


@pytest.mark.parametrize("mode", ["log", "table", "overflow", "rank", "walkmax"])
@pytest.mark.parametrize("k,n_reads,read_len,cfg", [(31, 100000, 150, 3), (55, 40000, 150, 4), (127, 4000, 1000, 5)])
def test_glue_record_paths_gpu(oracle, hip, mode, k, n_reads, read_len, cfg, monkeypatch):
    """the default single-rank path puts the glue records straight into the join buckets from the compaction kernels and walks
    the chains from their heads; the sequential log + scatter pass, the global-table join, the bucket-overflow fallback, the
    list-ranking path (rank) and the walk that gives up on a chain of more than one piece and hands over to the ranking
    (walkmax) must give the same graph"""
    monkeypatch.setenv({"log": "CDBG_GLUE_LOG", "table": "CDBG_GLUE_TABLE", "overflow": "CDBG_JOIN_LOG_JB", "rank": "CDBG_GLUE_RANK", "walkmax": "CDBG_WALK_MAX"}[mode],
                       "0" if mode in ("overflow", "walkmax") else "1")
    st = assert_parity(oracle, hip, oracle.synth_reads(n_reads, read_len, cfg), k, 2)["stats"]
    assert st["n_walked_unitigs"] == (0 if mode in ("rank", "walkmax") else st["n_unitigs"])


@pytest.mark.parametrize("k", [64, 127, 160])
@pytest.mark.parametrize("case", ["sifted", "solid_overflow", "fingerprint_overflow", "many_members"])
def test_count_sift_tier_gpu(oracle, hip, k, case):
    """k-mers of three words and more under an abundance filter, ONE partition of mostly once-seen k-mers that overflows the
    one-pass table: the sifting tier (fingerprints first, exact counts for what was seen again: k_count_fast.h) takes it; too
    many k-mers seen again for its small exact table, or too many fingerprints, and the multi-pass kernel does"""
    rng = random.Random(k * 7 + len(case))
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    solid_len = {"sifted": 250, "solid_overflow": 1500, "fingerprint_overflow": 250, "many_members": 300}[case] + k
    # (many_members: 45 noise reads = 4.8 K once-seen k-mers, 59 % of the 8192 fingerprint words.  With 55 -- 73 % -- the k = 160 case sat at the
    #  probe limit of the fingerprint table and the tier that took the partition depended on the order the records had landed in: right
    #  results either way, but this test pins the tier)
    noise = {"sifted": 26, "solid_overflow": 8, "fingerprint_overflow": 90, "many_members": 45}[case]
    g = rnd(solid_len)
    reads = [g, g, g[5:], g[::-1].translate(str.maketrans("ACGT", "TGCA"))] + [rnd(k + 99) for _ in range(noise)]
    if case == "many_members":                           # > 1280 member k-mers per wave (k_count_fast.h SIFT_MS_CAP): the members beyond find their fingerprint by its tag
        reads += [g] * 24
    reads.append(g[:k + 10] + rnd(1) + g[k + 11:2 * k + 30])
    got = assert_parity(oracle, hip, "\n".join(reads) + "\n", k, 2, log2_partitions=0)
    assert got["stats"]["n_multipass_partitions"] == (0 if case in ("sifted", "many_members") else 1), got["stats"]


@pytest.mark.parametrize("k", [55, 127])
def test_identical_multiword_keys_in_one_wave_gpu(oracle, hip, k):
    """the same multi-word k-mers from every lane of a wave (2000 copies of one read, both strands) on the device: a wave has
    no independent thread scheduling, so the W = 2 / 4 find-or-insert must publish inside the claiming iteration"""
    rng = random.Random(k)
    r = "".join(rng.choice("ACGT") for _ in range(2 * k + 40))
    rc = r[::-1].translate(str.maketrans("ACGT", "TGCA"))
    for log_np in (0, 3):
        assert_parity(oracle, hip, "\n".join([r, rc] * 1000) + "\n", k, 2, log2_partitions=log_np)


@pytest.mark.parametrize("k,cfg,n_reads,read_len,log_np", [(31, 3, 60000, 150, 12), (55, 4, 40000, 150, 11), (127, 5, 3000, 1000, 10), (255, 5, 1500, 1000, 10)])
@pytest.mark.parametrize("slices,part_cap,defer_cap", [("2", None, None), ("4", None, None), ("16", None, None), ("4,4,4,2,1,1", None, None), ("8,4,2,1,1", None, None), ("4", "260", None), ("4", None, "16"), ("0", None, None)])
def test_deferred_record_placement_gpu(oracle, hip, k, cfg, n_reads, read_len, log_np, slices, part_cap, defer_cap, monkeypatch):
    """deferred placement on the device (host_count.h, k_scan.h k_place): the scan places the first slice of the partition space and appends the
    records of the others to streams; k_place scatters stream q on a second HIP stream while k_count_fast counts slice q - 1 (events between
    them).  Against the oracle: unitig set and (k-mer, count) set, for one-, two-, four- and eight-word k-mers; 2 / 4 / 16 slices; regions so
    small that k_place spills (repair behind the last stream); streams so small that they fill up (the scan places the rest); 0 = off"""
    monkeypatch.setenv("CDBG_SCAN_MODE", "capped"); monkeypatch.setenv("CDBG_DEFER_SLICES", slices)
    if part_cap:
        monkeypatch.setenv("CDBG_PART_CAP", part_cap)
    if defer_cap:
        monkeypatch.setenv("CDBG_DEFER_CAP", defer_cap)
    text = oracle.synth_reads(n_reads, read_len, cfg)
    st = assert_parity(oracle, hip, text, k, 2, log2_partitions=log_np)["stats"]
    if slices == "0":
        assert st["count_slices"] == 1 and st["n_deferred_records"] == 0
    else:
        assert st["count_slices"] == (len(slices.split(",")) if "," in slices else int(slices)) and st["n_deferred_records"] > 0


@pytest.mark.parametrize("part_cap", [None, "64"])
def test_capped_single_pass_scan_gpu(oracle, hip, part_cap, monkeypatch):
    """the large-input scan path (single pass, fixed-capacity partition regions, spill repair)"""
    monkeypatch.setenv("CDBG_SCAN_MODE", "capped")
    if part_cap:
        monkeypatch.setenv("CDBG_PART_CAP", part_cap)
    text = oracle.synth_reads(40000, 150, 3)
    assert_parity(oracle, hip, text, 31, 2, log2_partitions=10)


@pytest.mark.parametrize("k,cfg,n_reads,read_len,log_np,part_cap,var_scale", [
    (31, 3 | 0x100, 60000, 150, 10, None, None), (31, 3 | 0x100, 60000, 150, 10, "24", None), (31, 3, 40000, 150, 10, "16", "0.5"),
    (55, 4 | 0x100, 30000, 150, 9, "8", None), (127, 5 | 0x100, 2500, 1000, 8, "8", "0.3")])
def test_capped_regions_with_overflow_regions_gpu(oracle, hip, k, cfg, n_reads, read_len, log_np, part_cap, var_scale, monkeypatch):
    """the single-pass record layout of skewed inputs (CDBG_SCAN_MODE=var): uniform capped regions + an overflow region for every partition
    the sampled histogram finds heavy, the uniform region's records moved to the front of it afterwards; a small CDBG_PART_CAP makes every
    busy partition heavy, CDBG_VAR_SCALE makes the overflow regions too small (spill list + repair of partitions that have one)"""
    monkeypatch.setenv("CDBG_SCAN_MODE", "var")
    if part_cap:
        monkeypatch.setenv("CDBG_PART_CAP", part_cap)
    if var_scale:
        monkeypatch.setenv("CDBG_VAR_SCALE", var_scale)
    text = oracle.synth_reads(n_reads, read_len, cfg)
    assert_parity(oracle, hip, text, k, 2, log2_partitions=log_np)


@pytest.mark.parametrize("k,cfg,n_reads,read_len,log_np,part_cap,var_scale,slices", [
    (31, 3 | 0x100, 60000, 150, 10, None, None, "2"), (31, 3 | 0x100, 60000, 150, 10, "24", None, "4"), (31, 3, 40000, 150, 10, "16", "0.5", "2"),
    (55, 4 | 0x100, 30000, 150, 10, "8", "0.4", "4")])
def test_deferred_placement_with_overflow_regions_gpu(oracle, hip, k, cfg, n_reads, read_len, log_np, part_cap, var_scale, slices, monkeypatch):
    """deferred placement on the layout of skewed inputs (CDBG_SCAN_MODE=var): k_place looks a heavy partition's overflow word up like the scan,
    k_ovf_finish runs slice by slice in front of every slice's count, spills out of overflow regions are repaired behind the last stream"""
    monkeypatch.setenv("CDBG_SCAN_MODE", "var"); monkeypatch.setenv("CDBG_DEFER_SLICES", slices)
    if part_cap:
        monkeypatch.setenv("CDBG_PART_CAP", part_cap)
    if var_scale:
        monkeypatch.setenv("CDBG_VAR_SCALE", var_scale)
    text = oracle.synth_reads(n_reads, read_len, cfg)
    st = assert_parity(oracle, hip, text, k, 2, log2_partitions=log_np)["stats"]
    # (overflow regions made too small on purpose may overflow the spill list as well: the step then runs once more without deferral -- parity holds either way)
    assert st["count_slices"] in ((1, int(slices)) if var_scale else (int(slices),))


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
    # (the buckets no LDS tier takes: through the second-level split, k_split.h -- and, with CDBG_NO_SPLIT, through the HBM-table tier)
    from parity import assert_verified
    digests = []
    for no_split in (False, True):
        if no_split:
            os.environ["CDBG_NO_SPLIT"] = "1"
        try:
            g = bcalm_amd.Graph(k, 2, lib=hip, log2_partitions=14)
            g.generate_reads(200000, 1000, 5)
            g.run()
            st = g.stats()
            ut = g.unitigs()
            assert_verified(g)
            digests.append(g.digest()["set_digest"])
            g.close()
        finally:
            os.environ.pop("CDBG_NO_SPLIT", None)
        assert (st["n_big_partitions"] > 3000) if no_split else (st["n_split_buckets"] > 3000 and st["n_big_partitions"] < 50), st
        assert sum(len(s) - k + 1 for s, _ in ut) == st["n_solid"] and len(ut) == st["n_unitigs"]
    assert digests[0] == digests[1]


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


def test_one_giant_partition_many_weak_kmers(oracle, hip):
    """ONE partition in the HBM count table whose fill (distinct k-mers) is far above the solid capacity
    (members / abundance-min): 2 % errors, abundance-min 3, k = 97.  Found by bench_micro/fuzz_gpu.py (seed 21,
    iteration 2124): the single-pass sweep reserved the table's fill and reported a solid overflow."""
    rng = random.Random(97)
    g = "".join(rng.choice("ACGT") for _ in range(200000))
    reads = []
    for _ in range(5000):
        L = rng.randrange(150, 300); s = rng.randrange(0, len(g) - L)
        reads.append("".join((rng.choice("ACGT") if rng.random() < 0.02 else c) for c in g[s:s + L]))
    text = "\n".join(reads) + "\n"
    exp = oracle.run(text, 97, 3)
    got = assert_parity(oracle, hip, text, 97, 3, log2_partitions=0)
    assert got["stats"]["n_distinct"] == exp["stats"]["distinct"] > 4 * exp["stats"]["solid"]


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


@pytest.mark.parametrize("k,n_reads,read_len", [(128, 12000, 1000), (191, 8000, 1000), (255, 8000, 1000)])
def test_wide_kmers_gpu(oracle, hip, k, n_reads, read_len):
    """k = 128 .. 255 on the device (five-, six- and eight-word k-mers: /root/reference/README.md:91-99, spans beyond the default
    list): config-5-like reads against the oracle, automatic partitioning and one forced bucket, plus the device-side definition check"""
    import bcalm_amd
    from parity import assert_verified
    text = oracle.synth_reads(n_reads, read_len, 5)
    for kw in ({}, {"log2_partitions": 0}):
        exp = oracle.run(text, k, 2)
        g = bcalm_amd.Graph(k, 2, lib=hip, **kw)
        g.push_text(text); g.run()
        st = g.stats(); got = oracle_lib.canonical_set(oracle, g.unitigs(), k); assert_verified(g); g.close()
        assert st["kmer_words"] == k // 32 + 1
        assert st["n_distinct"] == exp["stats"]["distinct"] and st["n_solid"] == exp["stats"]["solid"]
        assert got == exp["unitigs"]


@pytest.mark.gpu
@pytest.mark.parametrize("k,amin,n,L,cfg", [(31, 1, 200_000, 150, 3), (31, 2, 400_000, 150, 3 | 0x100), (32, 1, 100_000, 150, 3), (55, 1, 100_000, 150, 4), (127, 1, 20_000, 1000, 5)])
def test_edge_conservation_sees_over_compaction_gpu(oracle, hip, k, amin, n, L, cfg):
    """cdbg_verify_edges on the HIP result (bidirected-graphs-in-bcalm2.md:85, the inner-junction half of the unitig definition):
    the edges of the solid graph are the links plus the inner adjacencies; a unitig merged THROUGH a branching junction -- planted
    in the fetched result and handed back through cdbg_verify_unitigs -- breaks exactly that; a cut unitig shows as mergeable ends"""
    from bcalm_amd import api
    from parity import check_edge_conservation_is_sensitive
    g = api.Graph(k, amin, lib=hip)
    g.generate_reads(n, L, cfg)
    g.run()
    planted = check_edge_conservation_is_sensitive(g, k)
    g.close()
    assert planted >= 1


@pytest.mark.parametrize("k,glen,log_np", [(31, 400000, 10), (21, 200000, 9), (15, 60000, 8), (30, 300000, 10), (55, 400000, 10), (96, 200000, 9), (127, 200000, 9)])
@pytest.mark.parametrize("tier2", ["1", "0"])
def test_second_wave_tier_one_word_gpu(oracle, hip, k, glen, log_np, tier2, monkeypatch):
    """buckets of 257 .. 512 entries of one-word k-mers through the second one-wave compaction tier (1024-slot junction table) on the device"""
    import bcalm_amd
    from parity import assert_verified
    from test_hostsim_pipeline import _mid_bucket_text
    if k <= 31:
        monkeypatch.setenv("CDBG_CW_TIER2", tier2)
    elif tier2 == "0":
        monkeypatch.setenv("CDBG_CW_TIER3", "off")
    text = _mid_bucket_text(k, glen, glen + k)
    assert_parity(oracle, hip, text, k, 1, log2_partitions=log_np)
    gg = bcalm_amd.Graph(k, 1, lib=hip, log2_partitions=log_np)
    gg.push_text(text); gg.run(); assert_verified(gg); gg.close()
