"""CPU tests of the real kernel source through the SIMT simulator (tests/hostsim):
scan -> count -> compact -> glue against the oracle, over every golden input, with the
partition count and minimizer length swept so that every code path (travellers, open
ends, confirms, cross-bucket cycles, big-partition fallback) is exercised."""
import json
import os
import random
import sys

import pytest

import hostsim_lib
import oracle_lib
from parity import assert_parity, run_graph

ROOT = oracle_lib.ROOT
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))


@pytest.fixture(scope="module")
def sim():
    return hostsim_lib.load()


def _case(key):
    name, k, amin = key.split("/")
    return name, int(k), int(amin)


@pytest.mark.parametrize("key", sorted(GOLD))
@pytest.mark.parametrize("log_np,m", [(0, 0), (3, 0), (6, 5), (9, 4)])
def test_golden_parity(oracle, sim, key, log_np, m):
    name, k, amin = _case(key)
    text = oracle_lib.read_input(name)
    got = assert_parity(oracle, sim, text, k, amin, log2_partitions=log_np, minimizer_size=min(m, k - 1) if m else 0)
    exp = GOLD[key]
    assert oracle_lib.canonical_set(oracle, got["unitigs"], k) == [tuple(u) for u in exp["unitigs"]]
    assert oracle_lib.solid_sha256(got["solid"]) == exp["solid"]["sha256"]


@pytest.mark.parametrize("seed", range(16))
def test_random_low_complexity(oracle, sim, seed):
    """two-letter genomes: palindromic junctions, hairpins, self-loops, cycles; seeds >= 8: even k, where the k-mers
    themselves can be palindromes (AT genomes are full of them)"""
    rng = random.Random(4200 + seed)
    k = rng.choice([5, 7, 9, 11, 13] if seed < 8 else [4, 6, 8, 10, 12])
    g = "".join(rng.choice("AT" if seed % 2 else "ACG") for _ in range(rng.randrange(60, 400)))
    reads = []
    for _ in range(rng.randrange(3, 30)):
        L = rng.randrange(1, len(g)); s = rng.randrange(0, len(g) - L + 1)
        reads.append(g[s:s + L])
    text = "\n".join(reads) + "\n"
    assert_parity(oracle, sim, text, k, rng.choice([1, 1, 2]), log2_partitions=rng.choice([0, 2, 5]),
                  minimizer_size=rng.choice([2, 3, 4]))


@pytest.mark.parametrize("cfg,first,total", [(3, 5, 1000), (3 | 0x100, 5, 1000), (3 | 0x100, 70000, 200000), (4 | 0x100, 123456, 400000), (-1, 0, 400)])
def test_synthetic_generator_matches_oracle(oracle, sim, cfg, first, total):
    """plain, hostile (cfg | 0x100: low-complexity blocks, repeat copies, homopolymer runs, coverage skew) and all-'A' modes:
    the device generator and the oracle's write the same bytes"""
    from bcalm_amd import api
    g = api.Graph(31, 2, lib=sim)
    g.generate_reads(400, 150, cfg, first_read=first, total_reads=total)
    got = g.read_text(0, 400 * 151)
    g.close()
    exp = oracle.synth_reads(400, 150, cfg, first=first, total=total)
    assert got == exp
    if cfg >= 0x100 and total >= 200000:
        # the hostile features are really there: exact repeat copies make some 31-mers far more frequent than 30x
        whole = oracle.synth_reads(20000, 150, cfg, first=0, total=total).decode()
        from collections import Counter
        c = Counter(whole[i:i + 16] for i in range(0, len(whole) - 16, 7))
        assert c.most_common(1)[0][1] > 200


def test_packed_unitigs_match_ascii(oracle, sim):
    """the 2-bit arena the glue stage leaves resident (cdbg_fetch_unitigs_packed; SURVEY.md 8d end state) decodes to the
    ASCII unitigs, KC included"""
    from bcalm_amd import api
    for k, text in ((31, oracle.synth_reads(300, 150, 3)), (8, oracle_lib.read_input("even_k8").encode()), (77, oracle_lib.read_input("rand_w3").encode())):
        g = api.Graph(k, 1, lib=sim, log2_partitions=4)
        g.push_text(text); g.run()
        ascii_set = g.unitigs()
        arena, off, ln, kc = g.unitigs_packed()
        g.close()
        dec = [("".join("ACGT"[(arena[j >> 2] >> (2 * (j & 3))) & 3] for j in range(o, o + n)), c) for o, n, c in zip(off, ln, kc)]
        assert dec == ascii_set and len(dec) > 0


def test_hostile_reads_parity(oracle, sim):
    """the hostile generator (cfg | 0x100) through the whole pipeline: low-complexity blocks and skewed coverage overfill
    partitions, so the capped scan spills and the count tiers defer (every tier forced by the small partition count)"""
    text = oracle.synth_reads(1500, 150, 3 | 0x100, first=0, total=1500).decode()
    assert_parity(oracle, sim, text, 31, 2, log2_partitions=4)
    assert_parity(oracle, sim, text, 21, 1, log2_partitions=7, minimizer_size=8)


@pytest.mark.parametrize("k,cfg,n_reads,read_len,log_np", [(55, 4 | 0x100, 600, 150, 3), (32, 3 | 0x100, 600, 150, 3), (64, 4 | 0x100, 500, 150, 2), (127, 5 | 0x100, 80, 1000, 2)])
def test_hostile_reads_parity_multiword(oracle, sim, k, cfg, n_reads, read_len, log_np):
    """the hostile generator with multi-word k-mers (two, three, four words; even k on the word-count boundaries): duplicates of one
    key and top words of all T in the slot-claim protocol of the multi-word tables, through the simulator build -- which also
    re-derives the junction-ownership flag of every solid k-mer from its minimizers (device error 9 on a mismatch)"""
    text = oracle.synth_reads(n_reads, read_len, cfg, first=0, total=n_reads).decode()
    assert_parity(oracle, sim, text, k, 2, log2_partitions=log_np)


def test_synthetic_reads_parity(oracle, sim):
    text = oracle.synth_reads(300, 150, 3).decode()
    assert_parity(oracle, sim, text, 31, 2, log2_partitions=5)


# (the multi-rank path -- reads sharded, records exchanged to the partition owners -- needs a transport between
#  processes: tests/test_dist_gloo.py runs it with world sizes 2, 4 and 8 over gloo)


def test_errors(sim):
    from bcalm_amd import api
    api.Graph(30, 1, lib=sim).close()             # even k is accepted (README.md:99)
    with pytest.raises(api.CdbgError):
        api.Graph(256, 1, lib=sim)
    with pytest.raises(api.CdbgError):
        api.Graph(2, 1, lib=sim)
    g = api.Graph(21, 1, lib=sim)
    with pytest.raises(api.CdbgError):
        g.count()                                 # no reads
    with pytest.raises(api.CdbgError):
        g.glue()                                  # out of order
    g.close()


@pytest.mark.parametrize("key", ["rand_a/15/2", "rand_b/31/2", "circ_test3/7/1", "minitip/21/1"])
@pytest.mark.parametrize("part_cap", [None, "8", "1"])
def test_capped_single_pass_scan(oracle, sim, key, part_cap, monkeypatch):
    """single-pass scan into fixed-capacity partition regions (the large-input path), including
    forced spills + repair (CDBG_PART_CAP) -- same unitigs as the exact two-pass layout"""
    monkeypatch.setenv("CDBG_SCAN_MODE", "capped")
    if part_cap:
        monkeypatch.setenv("CDBG_PART_CAP", part_cap)
    name, k, amin = _case(key)
    assert_parity(oracle, sim, oracle_lib.read_input(name), k, amin, log2_partitions=4)


# (six cases of ~ 10 s each in the simulator -- 1024 partitions, one workgroup of fibers each; the GPU suite runs the full cross product at real sizes)
@pytest.mark.parametrize("key,slices,part_cap,defer_cap", [("rand_b/31/2", "2", None, None), ("rand_b/31/2", "4,4,4,2,1,1", None, None), ("rand_b/31/2", "4", "1", None),
                                                           ("rand_b/31/2", "4", None, "3"), ("rand_w2/55/2", "8", "2", "5"), ("rand_w2/55/2", "2", None, None)])
def test_deferred_record_placement(oracle, sim, key, slices, part_cap, defer_cap, monkeypatch):
    """deferred placement (host_count.h, k_scan.h k_place): the scan places the records of the first slice of the partition space and appends
    the others to streams, which k_place scatters while the one-pass count tier runs slice by slice.  Same unitigs and the same (k-mer, count)
    set as the oracle: with 2 / 4 / 16 slices; with regions so small that the placement kernel spills (repair after the last stream); with streams
    so small that most records find them full and are placed by the scan after all"""
    monkeypatch.setenv("CDBG_SCAN_MODE", "capped"); monkeypatch.setenv("CDBG_DEFER_SLICES", slices)
    if part_cap:
        monkeypatch.setenv("CDBG_PART_CAP", part_cap)
    if defer_cap:
        monkeypatch.setenv("CDBG_DEFER_CAP", defer_cap)
    name, k, amin = _case(key)
    text = oracle_lib.read_input(name)
    text = text + oracle.synth_reads(60, 150, 3).decode() if isinstance(text, str) else text + oracle.synth_reads(60, 150, 3)
    st = assert_parity(oracle, sim, text, k, amin, log2_partitions=10)["stats"]
    assert st["count_slices"] == (len(slices.split(",")) if "," in slices else int(slices)) and st["n_deferred_records"] > 0
    if not defer_cap and "," not in slices and st["n_records"] > 500:
        assert st["n_deferred_records"] > st["n_records"] * (int(slices) - 1) // int(slices) * 3 // 4      # ~ (S - 1) / S of the records went through the streams


def test_deferred_placement_off_for_few_partitions(oracle, sim, monkeypatch):
    """fewer than 64 partitions per slice: every record placed by the scan, one launch of the count tier"""
    monkeypatch.setenv("CDBG_SCAN_MODE", "capped")
    st = assert_parity(oracle, sim, oracle_lib.read_input("rand_b"), 31, 2, log2_partitions=6)["stats"]
    assert st["count_slices"] == 1 and st["n_deferred_records"] == 0


@pytest.mark.parametrize("key", ["rand_a/15/2", "rand_w2/55/2", "rand_w4/127/1", "circ_test3/7/1"])
@pytest.mark.parametrize("mode", ["log", "table", "overflow", "rank", "walkmax"])
def test_glue_record_paths(oracle, sim, key, mode, monkeypatch):
    """single-rank contexts put the glue records straight into the join buckets (the default, every other test);
    here the other paths: the sequential log + scatter pass (what multi-rank contexts exchange), the global-table join,
    and a join bucket that overflows during compaction (forced: ONE bucket of 256 records) -> compaction again through the log"""
    if mode == "log":
        monkeypatch.setenv("CDBG_GLUE_LOG", "1")
    elif mode == "table":
        monkeypatch.setenv("CDBG_GLUE_TABLE", "1")
    elif mode == "rank":                                   # list ranking instead of the walk from the heads (k_walk.h)
        monkeypatch.setenv("CDBG_GLUE_RANK", "1")
    elif mode == "walkmax":                                # the walk gives up on any chain of more than one piece: link[] rebuilt, ranking
        monkeypatch.setenv("CDBG_WALK_MAX", "0")
    else:
        monkeypatch.setenv("CDBG_JOIN_LOG_JB", "0")
    name, k, amin = _case(key)
    text = oracle_lib.read_input(name)
    if mode == "overflow":
        text = text + oracle.synth_reads(400, 150, 3).decode() if isinstance(text, str) else text + oracle.synth_reads(400, 150, 3)
    st = assert_parity(oracle, sim, text, k, amin, log2_partitions=5)["stats"]
    walked = not (mode in ("rank", "walkmax") or st["n_cycles"])        # (closed chains have no head: the walk hands over to the ranking)
    assert st["n_walked_unitigs"] == (st["n_unitigs"] if walked else 0)


def test_spilled_and_deferred_partition_is_reported(sim, monkeypatch):
    """a partition that spilled out of its capped region AND is deferred out of the repair launch (more LDS passes than
    allowed) has no region the HBM pass could read: the library must report it, not drop its k-mers"""
    import random
    from bcalm_amd import api
    monkeypatch.setenv("CDBG_SCAN_MODE", "capped"); monkeypatch.setenv("CDBG_PART_CAP", "1"); monkeypatch.setenv("CDBG_REPAIR_MAX_PASSES", "1")
    rng = random.Random(7)
    text = "".join(rng.choice("ACGT") for _ in range(9000)) + "\n"
    g = api.Graph(15, 1, lib=sim, log2_partitions=0)
    g.push_text(text)
    with pytest.raises(api.CdbgError) as e:
        g.run()
    assert "error 7" in str(e.value)
    g.close()


@pytest.mark.parametrize("k", [55, 127])
def test_identical_multiword_keys_in_one_wave(oracle, sim, k):
    """the same multi-word k-mers from every lane of a wave (200 copies of one read, both strands): the find-or-insert of
    W = 2 / 4 keys must publish inside the iteration that claimed the slot"""
    import random
    rng = random.Random(k)
    r = "".join(rng.choice("ACGT") for _ in range(2 * k + 40))
    rc = r[::-1].translate(str.maketrans("ACGT", "TGCA"))
    assert_parity(oracle, sim, "\n".join([r, rc] * 100) + "\n", k, 2, log2_partitions=2)


def test_partial_unitig_fetch(oracle, sim):
    """cdbg_fetch_unitigs over sub-ranges (gathered on the device, one copy) returns the same records as the full fetch"""
    from bcalm_amd import api
    g = api.Graph(21, 2, lib=sim)
    g.push_text(oracle.synth_reads(300, 150, 3)); g.run()
    full = g.unitigs()
    assert len(full) > 10
    for first, count in [(0, 1), (3, 5), (len(full) - 2, 2), (1, len(full) - 1), (len(full), 0)]:
        assert g.unitigs(first, count) == full[first:first + count]
    g.close()


@pytest.mark.parametrize("k,glen", [(31, 3800), (31, 11000), (55, 1900), (55, 6000), (127, 1900), (127, 6000)])
@pytest.mark.parametrize("mode", ["exact", "capped"])
def test_count_tiers(oracle, sim, k, glen, mode, monkeypatch):
    """ONE partition whose distinct k-mers overflow the one-pass LDS table: the second tier (same kernel, table twice the
    size, over the retry list) takes the smaller case, the multi-pass kernel the larger one"""
    import random
    from bcalm_amd import api
    if mode == "capped":
        monkeypatch.setenv("CDBG_SCAN_MODE", "capped")
    rng = random.Random(glen + k)
    g = "".join(rng.choice("ACGT") for _ in range(glen))
    text = g + "\n" + g[100:100 + 2 * k] + "\n"
    got = assert_parity(oracle, sim, text, k, 1, log2_partitions=0)
    small = glen in (3800, 1900)
    assert got["stats"]["n_multipass_partitions"] == (0 if small else 1)


@pytest.mark.parametrize("k", [64, 96, 127, 160])
@pytest.mark.parametrize("case", ["sifted", "solid_overflow", "fingerprint_overflow", "many_members"])
def test_count_sift_tier(oracle, sim, k, case):
    """k-mers of three words and more under an abundance filter: ONE partition of mostly once-seen k-mers overflows the one-pass table and
    goes to the sifting tier (fingerprints first, exact counts for what was seen again: k_count_fast.h); too many k-mers seen
    again for its small exact table, or too many fingerprints, and the multi-pass kernel takes the partition"""
    import random
    rng = random.Random(k * 7 + len(case))
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    solid_len = {"sifted": 250, "solid_overflow": 1500, "fingerprint_overflow": 250, "many_members": 300}[case] + k
    noise = {"sifted": 26, "solid_overflow": 8, "fingerprint_overflow": 90, "many_members": 55}[case]
    g = rnd(solid_len)
    reads = [g, g, g[5:], g[::-1].translate(str.maketrans("ACGT", "TGCA"))] + [rnd(k + 99) for _ in range(noise)]
    if case == "many_members":                           # > 1280 member k-mers per wave (k_count_fast.h SIFT_MS_CAP): the members beyond find their fingerprint by its tag
        reads += [g] * 16
    # a k-mer seen once inside an otherwise repeated read, and a once-seen k-mer whose reverse complement comes from another read
    reads.append(g[:k + 10] + rnd(1) + g[k + 11:2 * k + 30])
    got = assert_parity(oracle, sim, "\n".join(reads) + "\n", k, 2, log2_partitions=0)
    assert got["stats"]["n_multipass_partitions"] == (0 if case in ("sifted", "many_members") else 1), got["stats"]


@pytest.mark.parametrize("k,amin,log_np,m", [(128, 1, 0, 0), (128, 2, 3, 0), (160, 1, 2, 12), (191, 2, 3, 16), (192, 1, 1, 0), (224, 1, 0, 0), (255, 1, 4, 16), (255, 2, 2, 0)])
def test_wide_kmers_beyond_the_default_span_list(oracle, sim, k, amin, log_np, m):
    """k = 128 .. 255 (five- to eight-word k-mers: the KSIZE_LIST entries beyond 128 that the reference takes as a build option,
    /root/reference/README.md:91-99): reads of both strands with errors and an N, against the oracle; the definition check on top"""
    from bcalm_amd import api
    from parity import assert_verified
    rng = random.Random(k)
    g = "".join(rng.choice("ACGT") for _ in range(5000))
    reads = []
    for i in range(40):
        L = rng.randrange(200, 900); s0 = rng.randrange(0, len(g) - L)
        r = g[s0:s0 + L]
        if rng.random() < 0.5:
            r = r[::-1].translate(str.maketrans("ACGT", "TGCA"))
        r = "".join((rng.choice("ACGT") if rng.random() < 0.004 else c) for c in r)
        if rng.random() < 0.2:
            r = r[:len(r) // 2] + "N" + r[len(r) // 2:]
        reads.append(r)
    text = "\n".join(reads) + "\n"
    got = assert_parity(oracle, sim, text, k, amin, log2_partitions=log_np, minimizer_size=m)
    assert got["stats"]["kmer_words"] == k // 32 + 1
    gg = api.Graph(k, amin, lib=sim, log2_partitions=log_np, minimizer_size=m)
    gg.push_text(text); gg.run(); assert_verified(gg); gg.close()


@pytest.mark.parametrize("k,cfg,n_reads,read_len,log_np,part_min,var_scale", [
    (31, 3 | 0x100, 1500, 150, 6, None, None), (31, 3, 1500, 150, 5, None, None), (55, 4 | 0x100, 700, 150, 4, None, None), (127, 5 | 0x100, 60, 1000, 3, None, None),
    (21, 3 | 0x100, 1500, 150, 6, 1, "0.02"), (64, 4 | 0x100, 700, 150, 4, 1, "0.02"),
    (31, 3 | 0x100, 1500, 150, 6, 8, None), (55, 4, 700, 150, 4, 16, None), (96, 5 | 0x100, 60, 1000, 3, 8, None), (31, 3, 1500, 150, 5, 24, "0.7")])
def test_single_pass_scan_into_estimated_regions(oracle, sim, k, cfg, n_reads, read_len, log_np, part_min, var_scale, monkeypatch):
    """CDBG_SCAN_MODE=var: the record layout of skewed inputs -- ONE scan pass into the uniform capped regions plus an overflow region
    for every partition the sampled histogram finds heavy (round 5; k_ovf_* in k_count.h).  CDBG_PART_CAP = 8 .. 24: a uniform capacity so
    small that every busy partition is 'heavy' and runs over into its overflow region (which holds it: the sample of a small input is the
    full histogram); with CDBG_VAR_SCALE the overflow regions are made too small on purpose, so that partitions spill out of them as
    well and are repaired on the device.  Same solid set and unitigs as the oracle"""
    monkeypatch.setenv("CDBG_SCAN_MODE", "var")
    if part_min:
        monkeypatch.setenv("CDBG_PART_CAP", str(part_min))
    if var_scale:
        monkeypatch.setenv("CDBG_VAR_SCALE", var_scale)
    text = oracle.synth_reads(n_reads, read_len, cfg)
    assert_parity(oracle, sim, text, k, 2, log2_partitions=log_np)
    assert_parity(oracle, sim, text, k, 1, log2_partitions=log_np)


@pytest.mark.parametrize("k,cfg,n_reads,read_len,part_min,var_scale,slices", [
    (31, 3 | 0x100, 800, 150, None, None, "2"), (21, 3 | 0x100, 800, 150, 1, "0.02", "4")])
def test_deferred_placement_with_overflow_regions(oracle, sim, k, cfg, n_reads, read_len, part_min, var_scale, slices, monkeypatch):
    """deferred placement on the layout of skewed inputs (CDBG_SCAN_MODE=var): k_place looks a heavy partition's overflow word up like the scan,
    k_ovf_finish runs slice by slice in front of every slice's count, spills out of overflow regions are repaired behind the last stream"""
    monkeypatch.setenv("CDBG_SCAN_MODE", "var"); monkeypatch.setenv("CDBG_DEFER_SLICES", slices)
    if part_min:
        monkeypatch.setenv("CDBG_PART_CAP", str(part_min))
    if var_scale:
        monkeypatch.setenv("CDBG_VAR_SCALE", var_scale)
    text = oracle.synth_reads(n_reads, read_len, cfg)
    st = assert_parity(oracle, sim, text, k, 2, log2_partitions=10)["stats"]
    assert st["count_slices"] in ((1, int(slices)) if var_scale else (int(slices),))


@pytest.mark.parametrize("k,glen,log_np", [(15, 6000, 0), (31, 9000, 1), (32, 6000, 0), (55, 5000, 0), (96, 4000, 1), (127, 4000, 0)])
@pytest.mark.parametrize("split", [True, False])
def test_second_level_bucket_split(oracle, sim, k, glen, log_np, split, monkeypatch):
    """buckets that no LDS tier of the compaction takes (thousands of solid k-mers: a random stretch, a two-letter
    low-complexity stretch, an inverted and a direct repeat, a homopolymer run, all in one or two buckets) are re-bucketed by
    sub-minimizer and go through the wave tier again (k_split.h); with CDBG_NO_SPLIT the HBM-table tier takes them.
    Same unitigs either way, and the device-side definition check holds"""
    from bcalm_amd import api
    from parity import assert_verified
    if not split:
        monkeypatch.setenv("CDBG_NO_SPLIT", "1")
    rng = random.Random(glen + k)
    comp = str.maketrans("ACGT", "TGCA")
    g = "".join(rng.choice("ACGT") for _ in range(glen))
    low = "".join(rng.choice("AT") for _ in range(glen // 3))
    text = "\n".join([g, low, g[300:300 + 3 * k][::-1].translate(comp) + "A" * (2 * k) + g[700:700 + 2 * k], g[100:100 + 2 * k]]) + "\n"
    got = assert_parity(oracle, sim, text, k, 1, log2_partitions=log_np)
    st = got["stats"]
    # (n_big_partitions counts the HBM-table fallbacks of count AND compact; a sub-bucket may still need one: short minimizers of a two-letter stretch collide)
    assert (st["n_split_buckets"] >= 1) == split and (split or st["n_big_partitions"] >= 1), st
    gg = api.Graph(k, 1, lib=sim, log2_partitions=log_np)
    gg.push_text(text); gg.run(); assert_verified(gg); gg.close()


@pytest.mark.parametrize("k,m", [(31, 16), (31, 15), (32, 16), (21, 6), (25, 10), (31, 10), (17, 16), (55, 16), (63, 16), (33, 15), (45, 12), (63, 8), (71, 16), (127, 16)])
def test_scan_window_variants(oracle, sim, k, m):
    """every (k, m) shape of the register-window scan: the 15-key specialisation (k-m == 15, with and without a
    full 32-bit m-mer mask), the 16-key one (k-m == 16), shorter and longer windows, two-word k-mers; reads with N, lower case, reads
    shorter than k and shorter than m, runs of invalid bytes next to tile and 16-base chunk borders"""
    rng = random.Random(1000 * k + m)
    g = "".join(rng.choice("ACGT") for _ in range(2500))
    g = g[:900] + g[200:420] + g[900:1500] + g[1000:1100][::-1].translate(str.maketrans("ACGT", "TGCA")) + g[1500:]
    reads = []
    for i in range(160):
        L = rng.choice([3, 10, k - 1, k, k + 1, 40, 90, 150, 151, 260])
        s = rng.randrange(0, len(g) - L)
        r = g[s:s + L]
        if rng.random() < 0.5:
            r = r[::-1].translate(str.maketrans("ACGT", "TGCA"))
        if rng.random() < 0.25 and L > 20:
            p = rng.randrange(0, L); r = r[:p] + rng.choice(["N", "NN", "n", "X" * 17]) + r[p + 1:]
        if rng.random() < 0.2:
            r = r.lower()
        reads.append(r)
    text = "\n".join(reads) + "\n"
    assert_parity(oracle, sim, text, k, rng.choice([1, 2]), log2_partitions=rng.choice([0, 4, 7]), minimizer_size=m)


def _ingest_roundtrip(lib):
    """push ~75 MB in uneven pieces (several 32 MB staging buffers, device text growing by doubling) and read it back"""
    from bcalm_amd import api
    rng = random.Random(99)
    block = "".join(rng.choice("ACGT") for _ in range(1 << 16))
    g = api.Graph(31, 1, lib=lib)
    expect = []
    total = 0
    while total < 75 * (1 << 20):
        n = rng.choice([1, 7, 150, 4096, 65536, 3 * 65536 + 11, 40 * 65536])
        piece = (block * (n // len(block) + 1))[:n]
        g.push_text(piece)
        expect.append(piece); total += n + 1
    text = "\n".join(expect) + "\n"
    assert len(text) == total
    for off, n in ((0, 1000), (total - 1000, 1000), ((32 << 20) - 500, 1000), ((64 << 20) - 500, 1000), (12345678, 200000)):
        assert g.read_text(off, n).decode() == text[off:off + n], off
    g.close()


def test_streaming_ingest_roundtrip(sim):
    _ingest_roundtrip(sim)


def test_result_digest_matches_formula(oracle, sim):
    """cdbg_digest (what bench.py asserts at full size) against the same formula evaluated on the ORACLE's unitigs"""
    from bcalm_amd import api
    from parity import set_digest
    for k, amin, n, L, cfg in ((31, 2, 1200, 150, 3), (55, 1, 600, 150, 4)):
        text = oracle.synth_reads(n, L, cfg)
        exp = oracle.run(text, k, amin, want_solid=True)
        g = api.Graph(k, amin, lib=sim)
        g.push_text(text); g.run()
        d = g.digest(); st = g.stats(); g.close()
        assert d["set_digest"] == set_digest(exp["unitigs"])
        assert d["kc_sum"] == d["solid_count_sum"] == sum(c for _, c in exp["solid"])
        assert d["kmers_in_unitigs"] == st["n_solid"] == exp["stats"]["solid"]


@pytest.mark.parametrize("k,amin,n,L,cfg", [(31, 2, 3000, 150, 3), (55, 1, 1500, 150, 4), (30, 1, 1500, 100, 3), (77, 2, 400, 400, 5), (127, 1, 300, 500, 5)])
def test_verify_matches_the_kmer_set(oracle, sim, k, amin, n, L, cfg):
    """cdbg_verify (bench.py's full-size check): the sums it reports for the unitigs and for the solid table are the
    formula evaluated on the ORACLE's solid k-mers; nothing is left mergeable; a set with one k-mer more, less or twice
    has other sums (the check is sensitive)"""
    from bcalm_amd import api
    from parity import kmer_set_sums
    text = oracle.synth_reads(n, L, cfg)
    exp = oracle.run(text, k, amin, want_solid=True)
    g = api.Graph(k, amin, lib=sim, log2_partitions=5)
    g.push_text(text); g.run()
    v = g.verify(); g.close()
    kmers = [s for s, _ in exp["solid"]]
    want = kmer_set_sums(kmers, k)
    assert v["unitig_kmers"] == want and v["solid_kmers"] == want
    assert v["mergeable_ends"] == 0
    assert kmer_set_sums(kmers[1:], k) != want and kmer_set_sums(kmers + kmers[:1], k) != want
    assert kmer_set_sums(kmers[1:] + kmers[1:2], k)[0] == want[0] and kmer_set_sums(kmers[1:] + kmers[1:2], k) != want


@pytest.mark.parametrize("k,amin,n,L,cfg", [(31, 1, 3000, 150, 3), (12, 1, 1500, 100, 3), (55, 1, 1500, 150, 4), (127, 1, 300, 500, 5)])
def test_edge_conservation_sees_over_compaction(sim, oracle, k, amin, n, L, cfg):
    """cdbg_verify_edges (.md:85, the inner-junction half of the definition): holds for the result, fails for a unitig that was
    merged through a branching junction (which the k-mer-set and maximality checks do not see), holds for a cut unitig"""
    from bcalm_amd import api
    from parity import check_edge_conservation_is_sensitive
    text = oracle.synth_reads(n, L, cfg)                 # (1 % errors at abundance-min 1: thousands of branching junctions)
    g = api.Graph(k, amin, lib=sim, log2_partitions=5)
    g.push_text(text); g.run()
    planted = check_edge_conservation_is_sensitive(g, k)
    g.close()
    assert planted >= 1


@pytest.mark.parametrize("key", sorted(GOLD))
def test_verify_on_goldens(sim, key):
    """every golden input (cycles, hairpins, palindromes, even k): unitig k-mers == solid set, no mergeable pair of ends"""
    from bcalm_amd import api
    from parity import assert_verified
    name, k, amin = _case(key)
    g = api.Graph(k, amin, lib=sim, log2_partitions=3)
    g.push_text(oracle_lib.read_input(name)); g.run()
    v = assert_verified(g); g.close()
    assert v["unitig_kmers"][0] == GOLD[key]["solid"]["n"]


@pytest.mark.parametrize("k,amin,n_reads,cfg", [(31, 2, 3000, 3), (21, 1, 2500, 2), (55, 2, 1500, 4)])
@pytest.mark.parametrize("prewarm", [False, True])
def test_streaming_scan_while_ingesting(oracle, sim, monkeypatch, k, amin, n_reads, cfg, prewarm):
    """cdbg_expect_input: the scan runs on the tiles that have landed while the rest is still being pushed (thresholds
    shrunk to simulator sizes); same result as the oracle, and tiles really were scanned early"""
    from bcalm_amd import api
    monkeypatch.setenv("CDBG_STAGE_BYTES", "16384"); monkeypatch.setenv("CDBG_STREAM_MIN_BYTES", "40000"); monkeypatch.setenv("CDBG_STREAM_BATCH_TILES", "4")
    if prewarm:                                          # the background thread that obtains text buffer, record region and solid arrays (cdbg_expect_input on a large input)
        monkeypatch.setenv("CDBG_PREWARM_MIN_BYTES", "1")
    text = oracle.synth_reads(n_reads, 150, cfg)
    exp = oracle.run(text, k, amin)
    reads = text.split(b"\n")
    g = api.Graph(k, amin, lib=sim, log2_partitions=6)
    g.expect_input(len(text))
    for i in range(0, len(reads), 97):
        chunk = b"\n".join(reads[i:i + 97])
        if chunk:
            g.push_text(chunk)
    g.run()
    st = g.stats()
    canon = oracle_lib.canonical_set(oracle, g.unitigs(), k)
    g.close()
    assert st["n_tiles_overlapped"] > 0
    assert st["n_distinct"] == exp["stats"]["distinct"] and canon == exp["unitigs"]


def test_solid_capacity_second_attempt(oracle, sim, monkeypatch):
    """the count stage sizes the solid arrays from a third of the bound first; an input that needs more makes the kernels report the
    overflow (nothing is written out of bounds) and the stage runs once more with the bound: same result (CDBG_SOLID_FIRST_TINY forces it)"""
    monkeypatch.setenv("CDBG_SOLID_FIRST_TINY", "1")
    for k, amin, n, L, cfg in ((31, 1, 1300, 150, 3), (127, 1, 200, 500, 5)):       # (just beyond the forced first capacity of 2^15 entries: the simulator is slow)
        got = assert_parity(oracle, sim, oracle.synth_reads(n, L, cfg), k, amin)
        assert got["stats"]["n_solid"] > (1 << 15)


def _mid_bucket_text(k, glen, seed):
    """a genome whose buckets hold a few hundred solid k-mers each (abundance-min 1), with an inverted repeat, a direct repeat, a two-letter stretch and a
    homopolymer run among them: buckets of 257 .. 512 entries for the second one-wave compaction tier of one-word k-mers"""
    rng = random.Random(seed)
    comp = str.maketrans("ACGT", "TGCA")
    g = "".join(rng.choice("ACGT") for _ in range(glen))
    low = "".join(rng.choice("AC") for _ in range(glen // 10))
    return "\n".join([g, low, g[300:300 + 3 * k][::-1].translate(comp) + "A" * (2 * k) + g[700:700 + 2 * k], g[100:100 + 2 * k], g[glen // 2:glen // 2 + 400]]) + "\n"


@pytest.mark.parametrize("k,glen,log_np", [(31, 12000, 5), (21, 6000, 4), (15, 5000, 4), (8, 3000, 3), (30, 9000, 5), (31, 40000, 7),
                                           # (k-mers of two to four words: the 1024-slot tier is their THIRD one-wave tier, behind 256 / 512 slots)
                                           (55, 12000, 5), (63, 6000, 4), (64, 6000, 4), (96, 12000, 5), (127, 6000, 4)])
@pytest.mark.parametrize("tier2", ["1", "0"])
def test_second_wave_tier_one_word(oracle, sim, k, glen, log_np, tier2, monkeypatch):
    """round 5: buckets of 257 .. 512 entries of one-word k-mers through a second one-wave tier (junction table of 1024 slots, 10-bit end ids in a
    slot) instead of the workgroup tier; CDBG_CW_TIER2 = 1 / 0 forces / forbids it (by default it runs when the first tier defers many buckets)"""
    from bcalm_amd import api
    from parity import assert_verified
    if k <= 31:
        monkeypatch.setenv("CDBG_CW_TIER2", tier2)
    elif tier2 == "0":
        monkeypatch.setenv("CDBG_CW_TIER3", "off")
    text = _mid_bucket_text(k, glen, glen + k)
    assert_parity(oracle, sim, text, k, 1, log2_partitions=log_np)
    gg = api.Graph(k, 1, lib=sim, log2_partitions=log_np)
    gg.push_text(text); gg.run(); assert_verified(gg); gg.close()
