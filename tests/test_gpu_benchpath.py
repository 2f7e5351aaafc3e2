"""GPU tests (-m gpu) of exactly the code path bench.py times, of the reference's own checker on
the HIP output, and of the real `bcalm` CLI binary (HIP build).

  * BENCH_r01 ran `minimizer_size 16, log2_partitions 22` -> k_scan_fast<1, EMIT_CAPPED, 15>, the
    capped single-pass record layout and sparse partitions: forced here at sizes the oracle
    finishes in seconds, bit-exact (VERDICT r1 "What's weak" #2).
  * /root/reference/scripts/unitigEvaluator.cpp:147-217 (prebuilt oracle/_ref/unitigEvaluator)
    judges the HIP unitigs of the config-2-shaped genome: the only reference-produced verdict
    available on this machine.
  * bcalm_amd/_build/bcalm (the replacement of /root/reference/src/main.cpp:26-51 and
    src/bcalm_1.cpp:49-97) on FASTA, FASTQ.gz and file-list inputs.
"""
import gzip
import random
import os
import re
import subprocess
import sys

import pytest

import oracle_lib
from parity import assert_parity, config2_genome

pytestmark = pytest.mark.gpu
ROOT = oracle_lib.ROOT
EVAL = os.path.join(ROOT, "oracle", "_ref", "unitigEvaluator")
BCALM = os.path.join(ROOT, "bcalm_amd", "_build", "bcalm")


@pytest.fixture(scope="module")
def hip():
    import bcalm_amd
    return bcalm_amd.load()


def _run_stats(hip, text, k, amin, **kw):
    import bcalm_amd
    g = bcalm_amd.Graph(k, amin, lib=hip, **kw)
    try:
        g.push_text(text); g.run()
        return g.stats(), oracle_lib.canonical_set(oracle_lib.load(), g.unitigs(), k)
    finally:
        g.close()


@pytest.mark.parametrize("log_np", [20, 22])
def test_config3_instantiation_parity(oracle, oracle_1m, hip, log_np, monkeypatch):
    """k = 31, m = 16 (window of exactly 15 m-mers), 2^20 / 2^22 partitions, capped single-pass scan, 1 M reads:
    the template instance and layout of the bench line, against the oracle"""
    monkeypatch.setenv("CDBG_SCAN_MODE", "capped")
    text, exp = oracle_1m
    st, canon = _run_stats(hip, text, 31, 2, minimizer_size=16, log2_partitions=log_np)
    assert st["minimizer_size"] == 16 and st["log2_partitions"] == log_np
    assert st["n_occurrences"] == exp["stats"]["occurrences"]
    assert st["n_distinct"] == exp["stats"]["distinct"] and st["n_solid"] == exp["stats"]["solid"]
    assert canon == exp["unitigs"]


def test_config3_instantiation_solid_set(oracle, hip, monkeypatch):
    """same instantiation, stage-1 surface: the exact (k-mer, count) set"""
    monkeypatch.setenv("CDBG_SCAN_MODE", "capped")
    text = oracle.synth_reads(200000, 150, 3)
    assert_parity(oracle, hip, text, 31, 2, minimizer_size=16, log2_partitions=20)
    assert_parity(oracle, hip, text, 31, 1, minimizer_size=16, log2_partitions=20)


def test_config4_instantiation_parity(oracle, hip, monkeypatch):
    """k = 55 (two-word k-mers), m = 16, 2^20 partitions, capped scan: the config-4 shape of bench.py --cfg 4"""
    monkeypatch.setenv("CDBG_SCAN_MODE", "capped")
    text = oracle.synth_reads(400000, 150, 4)
    exp = oracle.run(text, 55, 2)
    st, canon = _run_stats(hip, text, 55, 2, minimizer_size=16, log2_partitions=20)
    assert st["n_distinct"] == exp["stats"]["distinct"] and st["n_solid"] == exp["stats"]["solid"]
    assert canon == exp["unitigs"]


def test_config5_instantiation_parity(oracle, hip, monkeypatch):
    """k = 127 (four-word k-mers, generic scan), m = 16, 2^18 partitions, capped scan, 1 kbp reads"""
    monkeypatch.setenv("CDBG_SCAN_MODE", "capped")
    text = oracle.synth_reads(30000, 1000, 5)
    exp = oracle.run(text, 127, 2)
    st, canon = _run_stats(hip, text, 127, 2, minimizer_size=16, log2_partitions=18)
    assert st["n_distinct"] == exp["stats"]["distinct"] and st["n_solid"] == exp["stats"]["solid"]
    assert canon == exp["unitigs"]


def _evaluator_verdict(tmp_path, unitigs, ref_records, k):
    """/root/reference/scripts/unitigEvaluator.cpp:147-217 (string-keyed, any k) on `unitigs` against the k-mers of `ref_records`"""
    ref = tmp_path / "ref.fa"; utg = tmp_path / "utg.fa"
    ref.write_text("".join(f">r{i}\n{r}\n" for i, r in enumerate(ref_records)))
    utg.write_text("".join(f">{i}\n{s}\n" for i, (s, _) in enumerate(unitigs)))
    # (ONE thread: with more, a thread that passes the evaluator's `while (not eof)` test just before another thread's read
    #  sets the flag keeps its previous record -- getline leaves the string alone once the stream is bad -- and judges it a
    #  second time: spurious "REPEATED kmers" (unitigEvaluator.cpp:147-160); a single thread is exact and takes ~15 s at k = 127)
    out = subprocess.run([EVAL, str(utg), str(ref), str(k), "1"], capture_output=True, text=True, timeout=900).stdout
    final = out[out.index("FINAL RESULTS"):]
    nums = re.search(r"FINAL RESULTS:\s*\n(\d+) (\d+)", final)
    assert nums, out
    assert re.search(r"ERRONEOUS kmers:\s*0\b", final), final
    assert re.search(r"MISSING kmers:\s*0\b", final), final
    assert "REPEATED" not in final
    return int(nums.group(1)), int(nums.group(2))


@pytest.mark.skipif(not os.path.exists(EVAL), reason="oracle/_ref/unitigEvaluator not prebuilt (needs /root/reference at build time)")
@pytest.mark.parametrize("k", [31, 32, 55, 96, 127])
def test_reference_checker_accepts_hip_unitigs(oracle, hip, tmp_path, k):
    """the reference's own checker on the GPU output of the config-2 shape (4.64 Mbp genome with planted direct and
    inverted repeats, abundance-min 1) for one-, two-, three- and four-word k-mers and an even k: TP == all,
    FP == FN == 0, no repeated k-mer -- the only reference-produced verdict available on this machine"""
    import bcalm_amd
    from parity import assert_verified
    text = config2_genome(oracle)
    g = bcalm_amd.Graph(k, 1, lib=hip)
    g.push_text(text); g.run()
    ut = g.unitigs(); st = g.stats(); assert_verified(g); g.close()
    in_ref, tp = _evaluator_verdict(tmp_path, ut, [text.decode().strip()], k)
    assert in_ref == tp == st["n_solid"] == st["n_distinct"]


@pytest.mark.skipif(not os.path.exists(EVAL), reason="oracle/_ref/unitigEvaluator not prebuilt (needs /root/reference at build time)")
@pytest.mark.parametrize("k,n_reads,read_len,cfg", [(31, 150000, 150, 3 | 0x100), (55, 100000, 150, 4 | 0x100), (127, 12000, 1000, 5 | 0x100), (64, 12000, 1000, 5 | 0x100)])
def test_reference_checker_on_hostile_reads(oracle, hip, tmp_path, k, n_reads, read_len, cfg):
    """the same verdict on READS of the hostile generator (low-complexity blocks, 1000 copies of a repeat, homopolymer runs,
    coverage skew; 1 % substitutions) at abundance-min 1: the unitigs must spell exactly the k-mers of the reads, each once"""
    import bcalm_amd
    from parity import assert_verified
    text = oracle.synth_reads(n_reads, read_len, cfg)
    g = bcalm_amd.Graph(k, 1, lib=hip)
    g.push_text(text); g.run()
    ut = g.unitigs(); st = g.stats(); assert_verified(g); g.close()
    in_ref, tp = _evaluator_verdict(tmp_path, ut, [r for r in text.decode().split("\n") if r], k)
    assert in_ref == tp == st["n_solid"] == st["n_distinct"]


def test_against_a_real_bcalm_binary(oracle, hip):
    """the only door out of "parity unpinned" (/root/reference/test/simple_test.sh:5-9 diffs unitig sets the same way): when a
    real BCALM 2 executable is reachable ($BCALM_BIN, or `bcalm` on PATH that is not this repo's CLI), run it on a FASTA dump of
    200 K synthetic reads and compare the canonical (sequence, KC) sets with the HIP output.  Skips cleanly when there is none
    (the expected case: gatb-core is absent from /root/reference and nothing can be installed)."""
    import bcalm_amd
    from parity import diff_against_reference, reference_binary
    ref = reference_binary()
    if not ref:
        pytest.skip("no BCALM 2 binary on this box ($BCALM_BIN / `bcalm` on PATH)")
    for k, amin, cfg in ((31, 2, 3), (21, 1, 2)):
        text = oracle.synth_reads(200_000, 150, cfg)
        g = bcalm_amd.Graph(k, amin, lib=hip)
        g.push_text(text); g.run()
        ours = g.unitigs(); g.close()
        d = diff_against_reference(oracle, ref, text, k, amin, ours)
        assert d["equal"], d


def _parse_fa(path, k):
    recs = []
    lines = open(path).read().split("\n")
    for i in range(0, len(lines) - 1, 2):
        m = re.match(r">(\d+) LN:i:(\d+) KC:i:(\d+) km:f:(\d+\.\d)((?: L:[+-]:\d+:[+-])*) $", lines[i])
        assert m, lines[i]
        s = lines[i + 1]
        assert int(m.group(2)) == len(s)
        assert abs(float(m.group(4)) - round(int(m.group(3)) / (len(s) - k + 1), 1)) < 1e-9
        recs.append((s, int(m.group(3))))
    return recs


def test_real_cli_binary_on_gpu(oracle, tmp_path):
    """the HIP build of the CLI, file to file: FASTA, FASTQ.gz, a file listing both; `<prefix>.unitigs.fa` naming
    (bcalm_1.cpp:68-74), header grammar (README.md:62-72), error contract (main.cpp:39-48)"""
    assert os.path.exists(BCALM), "bcalm_amd/_build/bcalm missing: run __graft_entry__.build()"
    k = 31
    text = oracle.synth_reads(60000, 150, 3).decode()
    reads = [r for r in text.split("\n") if r]
    half = len(reads) // 2
    fa = tmp_path / "part1.fa"; fq = tmp_path / "part2.fastq.gz"; lst = tmp_path / "list_reads"
    # FASTA with wrapped lines, lower case and an N; FASTQ gzipped
    with open(fa, "w") as f:
        for i, r in enumerate(reads[:half]):
            r2 = r.lower() if i % 7 == 0 else r
            f.write(f">r{i} some comment\n{r2[:70]}\n{r2[70:]}\n")
        f.write(">with_n\nACGTACGTACGTACGTACGTACGTACGTACGTACGTNACGTTTGACCAGTAGGATACCAGATTTAGGACCATTAGGACCAT\n")
    with gzip.open(fq, "wt") as f:
        for i, r in enumerate(reads[half:]):
            f.write(f"@q{i}\n{r}\n+\n{'I' * len(r)}\n")
    lst.write_text(f"{fa}\n{fq}\n")
    whole = "\n".join(reads) + "\nACGTACGTACGTACGTACGTACGTACGTACGTACGTNACGTTTGACCAGTAGGATACCAGATTTAGGACCATTAGGACCAT\n"
    exp_all = oracle.run(whole, k, 2)
    exp_1 = oracle.run("\n".join(reads[:half]) + "\nACGTACGTACGTACGTACGTACGTACGTACGTACGTNACGTTTGACCAGTAGGATACCAGATTTAGGACCATTAGGACCAT\n", k, 2)
    exp_2 = oracle.run("\n".join(reads[half:]) + "\n", k, 1)

    r = subprocess.run([BCALM, "-in", str(lst), "-kmer-size", str(k), "-abundance-min", "2", "-gfa"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    recs = _parse_fa(tmp_path / "list_reads.unitigs.fa", k)
    assert oracle_lib.canonical_set(oracle, recs, k) == exp_all["unitigs"]
    gfa = (tmp_path / "list_reads.unitigs.gfa").read_text()
    assert gfa.startswith(f"H\tVN:Z:1.0\tks:i:{k}\n")
    fa_links = re.findall(r" L:([+-]):(\d+):([+-])", (tmp_path / "list_reads.unitigs.fa").read_text())
    gfa_links = re.findall(r"^L\t\d+\t([+-])\t(\d+)\t([+-])\t30M$", gfa, flags=re.M)
    assert sorted(fa_links) == sorted(gfa_links) and len(fa_links) > 0

    r = subprocess.run([BCALM, "-in", str(fa), "-kmer-size", str(k), "-abundance-min", "2"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert oracle_lib.canonical_set(oracle, _parse_fa(tmp_path / "part1.unitigs.fa", k), k) == exp_1["unitigs"]

    r = subprocess.run([BCALM, "-in", str(fq), "-kmer-size", str(k), "-abundance-min", "1", "-out", "fromfq"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert oracle_lib.canonical_set(oracle, _parse_fa(tmp_path / "fromfq.unitigs.fa", k), k) == exp_2["unitigs"]

    r = subprocess.run([BCALM, "-kmer-size", "21"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "EXCEPTION: Specifiy -in" in r.stdout
    r = subprocess.run([BCALM, "-in", "/nonexistent.fa"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "EXCEPTION:" in r.stdout


def test_real_cli_binary_gzip_inflated_by_all_threads(oracle, tmp_path):
    """one FASTQ.gz (single member, then 7 members) cut into 64 KB chunks and inflated by 8 threads (bcalm_amd/host/pgz.h), against the oracle and
    against the one-thread zlib path; a damaged file is an error"""
    assert os.path.exists(BCALM)
    k = 31
    reads = [r for r in oracle.synth_reads(40000, 150, 5).decode().split("\n") if r]
    exp = oracle.run("\n".join(reads) + "\n", k, 2)
    rng = random.Random(2)
    fq = "".join("@SRR1.%d\n%s\n+\n%s\n" % (i, r, "".join(chr(33 + rng.randrange(20, 41)) for _ in r)) for i, r in enumerate(reads)).encode()
    (tmp_path / "one.fastq.gz").write_bytes(gzip.compress(fq, 6))
    step = len(fq) // 7 + 1
    (tmp_path / "seven.fastq.gz").write_bytes(b"".join(gzip.compress(fq[i:i + step], 6) for i in range(0, len(fq), step)))
    env = dict(os.environ, BCALM_GZ_CHUNK="65536", BCALM_GZ_VERBOSE="1")
    for name in ("one", "seven"):
        r = subprocess.run([BCALM, "-in", name + ".fastq.gz", "-kmer-size", str(k), "-abundance-min", "2", "-nb-cores", "8", "-out", name], cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "inflated by 8 threads" in r.stderr and "input: 40000 sequences, 6000000 bases" in r.stdout, r.stdout + r.stderr
        assert oracle_lib.canonical_set(oracle, _parse_fa(tmp_path / (name + ".unitigs.fa"), k), k) == exp["unitigs"]
    r = subprocess.run([BCALM, "-in", "one.fastq.gz", "-kmer-size", str(k), "-abundance-min", "2", "-out", "serial"], cwd=tmp_path, capture_output=True, text=True, timeout=600,
                       env=dict(env, BCALM_GZ_SERIAL="1"))
    assert r.returncode == 0 and "inflated by" not in r.stderr
    assert oracle_lib.canonical_set(oracle, _parse_fa(tmp_path / "serial.unitigs.fa", k), k) == exp["unitigs"]
    blob = bytearray((tmp_path / "one.fastq.gz").read_bytes()); blob[len(blob) * 3 // 4] ^= 0x20
    (tmp_path / "bad.fastq.gz").write_bytes(bytes(blob))
    r = subprocess.run([BCALM, "-in", "bad.fastq.gz", "-kmer-size", str(k), "-abundance-min", "2", "-nb-cores", "8"], cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "EXCEPTION" in r.stdout, r.stdout + r.stderr


def test_result_digest_on_gpu(oracle, oracle_1m, hip):
    """the digests bench.py asserts at config-3 size, against the same formula on the oracle's unitigs (1 M reads)"""
    import bcalm_amd
    from parity import set_digest
    text, exp = oracle_1m
    g = bcalm_amd.Graph(31, 2, lib=hip)
    g.push_text(text); g.run()
    d = g.digest(); st = g.stats()
    g.reset(); g.run()
    d2 = g.digest(); g.close()
    assert d["set_digest"] == set_digest(exp["unitigs"]) == d2["set_digest"]
    assert d["kc_sum"] == d["solid_count_sum"] == sum(kc for _, kc in exp["unitigs"])
    assert d["kmers_in_unitigs"] == st["n_solid"] == exp["stats"]["solid"]


def test_four_million_reads_against_the_multithreaded_restatement(oracle, hip):
    """the largest oracle-checked run of the suite: 4 M x 150 bp reads (600 Mbp; the capped single-pass scan, 2^17+ partitions,
    the bench's kernels) against oracle/cpu_mt.cpp -- the multithreaded CPU restatement that bench.py times as its baseline,
    itself pinned to the scalar oracle in tests/test_oracle.py: distinct / solid / unitig counts, KC sum and the
    order-independent set digest must be equal"""
    import bcalm_amd
    text = oracle.synth_reads(4_000_000, 150, 3)
    cpu = oracle_lib.cpu_mt_run(text, 31, 2, os.cpu_count() or 8)
    g = bcalm_amd.Graph(31, 2, lib=hip)
    g.push_text(text); g.run()
    st = g.stats(); d = g.digest(); g.close()
    assert st["n_occurrences"] == cpu["occurrences"] == 4_000_000 * 120
    assert (st["n_distinct"], st["n_solid"], st["n_unitigs"]) == (cpu["distinct"], cpu["solid"], cpu["unitigs"])
    assert d["kc_sum"] == cpu["kc_sum"] and d["set_digest"] == cpu["set_digest"]
    assert st["unitig_bases"] == cpu["unitig_bases"]


def test_hostile_four_million_reads(oracle, hip):
    """what uniform random reads never exercise, at a size where the fallbacks stop being rare: 4 M x 150 bp reads of the HOSTILE
    generator (cfg | 0x100: two-letter low-complexity blocks, 400 exact copies of a 5 kbp repeat, 50 homopolymer runs, 20x
    coverage skew) generated on the device, against oracle/cpu_mt.cpp on the same bytes: counts, KC sum, set digest.  The
    partitions that overflow (spill repair, second count tier, multi-pass, HBM tables) are reported by the statistics."""
    import bcalm_amd
    cfg = 3 | 0x100
    g = bcalm_amd.Graph(31, 2, lib=hip)
    g.generate_reads(4_000_000, 150, cfg)
    text = g.read_text(0, 4_000_000 * 151)
    assert text[:151 * 1000] == oracle.synth_reads(1000, 150, cfg, first=0, total=4_000_000)
    cpu = oracle_lib.cpu_mt_run(text, 31, 2, os.cpu_count() or 8)
    g.run()
    st = g.stats(); d = g.digest(); g.close()
    assert st["n_occurrences"] == cpu["occurrences"] == 4_000_000 * 120
    assert (st["n_distinct"], st["n_solid"], st["n_unitigs"]) == (cpu["distinct"], cpu["solid"], cpu["unitigs"])
    assert d["kc_sum"] == cpu["kc_sum"] and d["set_digest"] == cpu["set_digest"]
    assert st["unitig_bases"] == cpu["unitig_bases"]
    assert st["n_multipass_partitions"] + st["n_big_partitions"] + st["n_split_buckets"] > 0          # the hostile input did reach the fallback tiers


@pytest.mark.parametrize("k,cfg,n_reads", [(55, 4, 4_000_000), (55, 4 | 0x100, 3_000_000), (40, 3 | 0x100, 2_000_000)])
def test_millions_of_reads_two_word_kmers_against_the_multithreaded_restatement(oracle, hip, k, cfg, n_reads):
    """round 5 (oracle/cpu_mt.cpp counts two-word k-mers): the same size class for k = 55 / 40 -- uniform and HOSTILE reads generated on the device against
    the multithreaded CPU restatement on the same bytes: occurrences, distinct / solid / unitig counts, KC sum, unitig bases and the set digest"""
    import bcalm_amd
    g = bcalm_amd.Graph(k, 2, lib=hip)
    g.generate_reads(n_reads, 150, cfg)
    text = g.read_text(0, n_reads * 151)
    cpu = oracle_lib.cpu_mt_run(text, k, 2, os.cpu_count() or 8)
    g.run()
    st = g.stats(); d = g.digest(); g.close()
    assert st["n_occurrences"] == cpu["occurrences"] == n_reads * (150 - k + 1)
    assert (st["n_distinct"], st["n_solid"], st["n_unitigs"]) == (cpu["distinct"], cpu["solid"], cpu["unitigs"])
    assert d["kc_sum"] == cpu["kc_sum"] and d["set_digest"] == cpu["set_digest"]
    assert st["unitig_bases"] == cpu["unitig_bases"]


@pytest.mark.parametrize("k,cfg,n_reads,read_len", [(55, 4 | 0x100, 150_000, 150), (32, 3 | 0x100, 150_000, 150), (64, 4 | 0x100, 100_000, 150),
                                                    (96, 5 | 0x100, 20_000, 1000), (127, 5 | 0x100, 20_000, 1000)])
def test_hostile_multiword_parity(oracle, hip, k, cfg, n_reads, read_len):
    """the hostile generator with two-, three- and four-word k-mers, odd and even k (k = 32, 64, 96 sit on the word-count
    boundaries of the span rule): homopolymer runs and low-complexity blocks put k-mers whose top word is all T -- the value next to
    the table's EMPTY / claimed encodings -- and hundreds of duplicates of one key into the same step of the multi-word slot-claim
    protocol, and the repeat copies overfill partitions (second count tier, multi-pass, workgroup compaction tiers).  Against the
    scalar oracle: occurrences, distinct, solid and the canonical unitig set with KC."""
    text = oracle.synth_reads(n_reads, read_len, cfg)
    exp = oracle.run(text, k, 2)
    st, canon = _run_stats(hip, text, k, 2)
    assert st["n_occurrences"] == exp["stats"]["occurrences"]
    assert st["n_distinct"] == exp["stats"]["distinct"] and st["n_solid"] == exp["stats"]["solid"]
    assert canon == exp["unitigs"]


@pytest.mark.parametrize("log_np", [-1, 0])
def test_abundance_saturates_at_31_bits(hip, log_np):
    """one k-mer seen more than 2^31 times (2.3 M reads of 1000 x 'A', k = 31): the count must clamp at 2^31 - 1 -- not carry
    into bit 31 of the count word, the traveller flag.  Expected by hand: ONE distinct k-mer A^31 whose only
    edge is the self-loop, so one unitig of 31 bases with KC = 2147483647 (README.md:62-72: KC is the sum of abundances)"""
    import bcalm_amd
    n_reads, L, k = 2_300_000, 1000, 31
    assert n_reads * (L - k + 1) > (1 << 31)
    g = bcalm_amd.Graph(k, 2, lib=hip, log2_partitions=log_np)
    g.generate_reads(n_reads, L, -1)
    g.run()
    st = g.stats(); ut = g.unitigs(); solid = g.solid_kmers() if hasattr(g, "solid_kmers") else None
    g.close()
    assert st["n_distinct"] == 1 and st["n_solid"] == 1 and st["n_unitigs"] == 1
    assert len(ut) == 1 and ut[0][0] in ("A" * 31, "T" * 31) and ut[0][1] == (1 << 31) - 1      # (orientation is unspecified, README.md:84-87)


def test_streaming_scan_while_ingesting_gpu(oracle, oracle_1m, hip, monkeypatch):
    """cdbg_expect_input on the device: the single-pass scan runs on the tiles that have landed while later chunks are
    still being pushed through the pinned staging buffers (copy stream -> event -> compute stream); 1 M reads vs the oracle"""
    import bcalm_amd
    monkeypatch.setenv("CDBG_STREAM_MIN_BYTES", str(16 << 20)); monkeypatch.setenv("CDBG_STREAM_BATCH_TILES", "2048")
    text, exp = oracle_1m
    g = bcalm_amd.Graph(31, 2, lib=hip)
    g.expect_input(len(text))
    step = 8 << 20
    pos = 0
    while pos < len(text):
        end = text.find(b"\n", min(len(text) - 1, pos + step))
        end = len(text) if end < 0 else end + 1
        g.push_text(text[pos:end - 1] if text[end - 1:end] == b"\n" else text[pos:end])
        pos = end
    g.run()
    st = g.stats()
    canon = oracle_lib.canonical_set(oracle, g.unitigs(), 31)
    g.close()
    assert st["n_tiles_overlapped"] > 0.5 * st["n_launch_scan"]
    assert st["n_distinct"] == exp["stats"]["distinct"] and st["n_solid"] == exp["stats"]["solid"]
    assert canon == exp["unitigs"]


def _fa_links(path):
    """-> {(u, from_sign, v, to_sign)} of a unitigs.fa"""
    out = set()
    for line in open(path):
        if line.startswith(">"):
            u = int(line[1:].split()[0])
            for fs, v, ts in re.findall(r" L:([+-]):(\d+):([+-])", line):
                out.add((u, fs, int(v), ts))
    return out


def test_cli_sharded_writer_through_one_rank_rccl(oracle, tmp_path):
    """the CLI's multi-GPU output path on a 1-GPU box: CDBG_FORCE_MULTI sends the one rank through the multi-rank code (RCCL
    communicator of one rank, sharded glue, the COLLECTIVE cdbg_link with job-wide unitig ids, the writer that walks the ranks).
    Same unitig set as the oracle, and the L: tokens are exactly the brute-force links of the written sequences"""
    sys.path.insert(0, os.path.join(oracle_lib.ROOT, "oracle"))
    import oracle_py as op
    assert os.path.exists(BCALM)
    k = 31
    text = oracle.synth_reads(20000, 150, 3).decode()
    fa = tmp_path / "reads.fa"
    with open(fa, "w") as f:
        for i, r in enumerate(x for x in text.split("\n") if x):
            f.write(f">r{i}\n{r}\n")
    env = dict(os.environ, CDBG_FORCE_MULTI="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([BCALM, "-in", str(fa), "-kmer-size", str(k), "-abundance-min", "2"], cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    recs = _parse_fa(tmp_path / "reads.unitigs.fa", k)
    assert oracle_lib.canonical_set(oracle, recs, k) == oracle.run(text, k, 2)["unitigs"]
    assert _fa_links(tmp_path / "reads.unitigs.fa") == op.links([s for s, _ in recs], k)


def test_cli_on_every_visible_gpu(oracle, tmp_path):
    """`bcalm -nb-gpus N` on a box with N >= 2 GPUs: every rank writes its share (job-wide ids, links across ranks).  Skips on one GPU"""
    import torch
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    worlds = [w for w in (2, 4, 8) if w <= n]
    if not worlds:
        pytest.skip("needs >= 2 GPUs: %d visible" % n)
    sys.path.insert(0, os.path.join(oracle_lib.ROOT, "oracle"))
    import oracle_py as op
    k = 31
    text = oracle.synth_reads(40000, 150, 3).decode()
    fa = tmp_path / "reads.fa"
    with open(fa, "w") as f:
        for i, r in enumerate(x for x in text.split("\n") if x):
            f.write(f">r{i}\n{r}\n")
    exp = oracle.run(text, k, 2)["unitigs"]
    for w in worlds:
        r = subprocess.run([BCALM, "-in", str(fa), "-kmer-size", str(k), "-abundance-min", "2", "-nb-gpus", str(w), "-out", f"w{w}"], cwd=tmp_path,
                           capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        assert r.returncode == 0, r.stdout + r.stderr
        recs = _parse_fa(tmp_path / f"w{w}.unitigs.fa", k)
        assert oracle_lib.canonical_set(oracle, recs, k) == exp
        assert _fa_links(tmp_path / f"w{w}.unitigs.fa") == op.links([s for s, _ in recs], k)
