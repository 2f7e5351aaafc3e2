"""CPU tests of the oracle itself: C restatement == independent Python restatement
== hand-transcribed SURVEY anchors; and the reference's own checker
(scripts/unitigEvaluator.cpp, built in place into oracle/_ref/) agrees that the
oracle's unitigs spell exactly the input's k-mer set with no repeats."""
import json
import os
import random
import re
import subprocess
import sys

import pytest

import oracle_lib

ROOT = oracle_lib.ROOT
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as op  # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
ANCH = json.load(open(os.path.join(ROOT, "tests", "golden", "anchors.json")))


def _case(key):
    name, k, amin = key.split("/")
    return name, int(k), int(amin)


@pytest.mark.parametrize("key", sorted(GOLD))
def test_c_oracle_matches_golden(oracle, key):
    name, k, amin = _case(key)
    text = oracle_lib.read_input(name)
    got = oracle.run(text, k, amin, want_solid=True)
    exp = GOLD[key]
    assert got["stats"] == exp["stats"]
    assert got["unitigs"] == [tuple(u) for u in exp["unitigs"]]
    assert len(got["solid"]) == exp["solid"]["n"]
    assert oracle_lib.solid_sha256(got["solid"]) == exp["solid"]["sha256"]
    if "list" in exp["solid"]:
        assert got["solid"] == [tuple(x) for x in exp["solid"]["list"]]
    assert got["digest"] == oracle_lib.digest_of(got["unitigs"])


@pytest.mark.parametrize("key", sorted(ANCH))
def test_survey_anchors(oracle, key):
    """SURVEY.md section 4 table (spec-derived by hand, independent of both oracles)"""
    name, k, amin = _case(key)
    got = oracle.run(oracle_lib.read_input(name), k, amin)
    a = ANCH[key]
    assert got["stats"]["distinct"] == a["distinct"]
    assert got["stats"]["solid"] == a["solid"]
    if "unitigs" in a:
        exp = sorted((oracle.canonical_unitig(s, k), kc) for s, ln, kc in a["unitigs"])
        assert got["unitigs"] == exp
        for s, ln, kc in a["unitigs"]:
            assert len(s) == ln
    if "circular" in a:
        assert [(len(s) - k + 1, len(s), kc) for s, kc in got["unitigs"]] == [tuple(x) for x in a["circular"]]
        assert got["circular"] == [1] * len(a["circular"])
    if "unitigs_partial" in a:
        assert got["stats"]["unitigs"] == a["n_unitigs"]
        have = dict(got["unitigs"])
        for s, ln, kc in a["unitigs_partial"]:
            assert have[oracle.canonical_unitig(s, k)] == kc
        rest = sorted((len(s), kc) for s, kc in got["unitigs"] if s not in {oracle.canonical_unitig(x[0], k) for x in a["unitigs_partial"]})
        assert rest == [tuple(x) for x in a["other"]]


@pytest.mark.parametrize("seed", range(12))
def test_c_vs_python_random(oracle, seed):
    rng = random.Random(1000 + seed)
    k = rng.choice([5, 7, 9, 11, 15, 21, 31, 33, 45, 63, 65, 99, 127] if seed < 6 else [4, 6, 8, 12, 16, 20, 30, 32, 48, 62, 64, 66, 80, 94, 96, 126])
    glen = rng.randrange(200, 900)
    alphabet = "ACGT" if seed % 3 else "AC"          # low complexity -> cycles, palindromes, self-loops
    g = "".join(rng.choice(alphabet) for _ in range(glen))
    reads = []
    for _ in range(rng.randrange(5, 60)):
        L = rng.randrange(1, min(glen, 3 * k + 40))
        s = rng.randrange(0, glen - L + 1)
        r = g[s:s + L]
        if rng.random() < 0.5:
            r = op.revcomp(r)
        if rng.random() < 0.2 and L > 2:
            p = rng.randrange(L); r = r[:p] + rng.choice("NnxACGTacgt") + r[p + 1:]
        reads.append(r)
    text = "\n".join(reads) + "\n"
    amin = rng.choice([1, 1, 2, 3])
    pu, pst = op.unitigs(text, k, amin)
    got = oracle.run(text, k, amin, want_solid=True)
    assert got["stats"] == pst
    assert got["unitigs"] == pu
    assert got["solid"] == op.solid_kmers(text, k, amin)


def test_unitig_kmer_partition(oracle):
    """maximal unitigs are a vertex decomposition (.md 'should be a vertex decomposition'):
    every solid k-mer appears in exactly one unitig, KC sums to the solid occurrences"""
    for key in sorted(GOLD):
        name, k, amin = _case(key)
        text = oracle_lib.read_input(name)
        got = oracle.run(text, k, amin, want_solid=True)
        seen = {}
        for s, kc in got["unitigs"]:
            for i in range(len(s) - k + 1):
                c = op.canonical(s[i:i + k])
                assert c not in seen, "k-mer repeated across unitigs"
                seen[c] = 1
        assert sorted(seen) == [x for x, _ in got["solid"]]
        assert sum(kc for _, kc in got["unitigs"]) == sum(c for _, c in got["solid"])


def test_even_k_palindromic_kmer_is_its_own_unitig(oracle):
    """even k (README.md:99): a k-mer equal to its reverse complement is reached by two distinct edges (.md:7,41-46), so it
    never merges: w + rc(w) in the middle of an otherwise linear read splits it into three unitigs"""
    left, w, right = "GGATCTTAGC", "ACCTGA", "CCATTGAGTC"
    pal = w + op.revcomp(w)                                   # 12 bases
    read = left + pal + right
    got = oracle.run(read + "\n", 12, 1)
    seqs = sorted(s for s, _ in got["unitigs"])
    assert oracle.canonical_unitig(pal, 12) in seqs
    assert got["stats"]["unitigs"] == 3 and sorted(len(s) for s in seqs) == [12, 12 + len(left) - 1, 12 + len(right) - 1]
    assert op.unitigs(read + "\n", 12, 1)[0] == got["unitigs"]
    with pytest.raises(ValueError):
        oracle.run("ACGTACGTACGT\n", 256, 1)


EVAL = os.path.join(ROOT, "oracle", "_ref", "unitigEvaluator")


@pytest.mark.skipif(not os.path.exists(EVAL) and not os.path.exists("/root/reference/scripts/unitigEvaluator.cpp"),
                    reason="reference checker neither prebuilt nor buildable here")
@pytest.mark.parametrize("key", ["tiny_read/21/1", "minitip/21/1", "pufferize_refs/9/1", "circ_test2/7/1", "rand_c/21/1"])
def test_reference_checker_accepts_oracle_unitigs(oracle, tmp_path, key):
    """the reference's own k-mer-set checker (scripts/unitigEvaluator.cpp:147-217):
    TP == all, FP == FN == 0, no 'REPEATED kmers' line.  abundance_min == 1 cases only
    (the checker has no abundance notion)."""
    if not os.path.exists(EVAL):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    name, k, amin = _case(key)
    assert amin == 1
    text = oracle_lib.read_input(name)
    got = oracle.run(text, k, amin)
    ref = tmp_path / "ref.fa"
    utg = tmp_path / "utg.fa"
    # the checker wants one-line-per-sequence FASTA and treats only 'N' as a break
    reads = [r for r in re.split(r"[^ACGTacgt]+", text) if len(r) >= k]
    ref.write_text("".join(f">r{i}\n{r.upper()}\n" for i, r in enumerate(reads)))
    utg.write_text("".join(f">{i}\n{s}\n" for i, (s, _) in enumerate(got["unitigs"])))
    out = subprocess.run([EVAL, str(utg), str(ref), str(k), "1"], capture_output=True, text=True, timeout=60).stdout
    final = out[out.index("FINAL RESULTS"):]
    nums = re.search(r"FINAL RESULTS:\s*\n(\d+) (\d+)", final)
    assert nums, out
    assert int(nums.group(1)) == int(nums.group(2)) == got["stats"]["solid"]
    assert re.search(r"ERRONEOUS kmers:\s*0\b", final), final
    assert re.search(r"MISSING kmers:\s*0\b", final), final
    assert "REPEATED" not in final


@pytest.mark.parametrize("k,amin,n_reads,cfg,threads", [(31, 2, 4000, 3, 4), (21, 1, 2000, 2, 3), (27, 3, 3000, 3, 8), (30, 2, 3000, 3, 4), (8, 1, 300, 3, 3),
                                                           # two-word k-mers (round 5: config 4's CPU baseline is the same program as config 3's)
                                                           (32, 2, 3000, 4, 4), (55, 2, 4000, 4, 8), (63, 1, 1500, 4, 3), (40, 2, 2000, 3 | 0x100, 5)])
def test_cpu_mt_baseline_matches_oracle(oracle, k, amin, n_reads, cfg, threads):
    """oracle/cpu_mt.cpp (bench.py's multithreaded CPU baseline) against the oracle: counts, KC sum and the set digest"""
    from parity import set_digest
    text = oracle.synth_reads(n_reads, 150, cfg)
    exp = oracle.run(text, k, amin)
    got = oracle_lib.cpu_mt_run(text, k, amin, threads)
    assert got["occurrences"] == exp["stats"]["occurrences"] and got["distinct"] == exp["stats"]["distinct"]
    assert got["solid"] == exp["stats"]["solid"] and got["unitigs"] == exp["stats"]["unitigs"]
    assert got["kc_sum"] == sum(kc for _, kc in exp["unitigs"])
    assert got["set_digest"] == set_digest(exp["unitigs"])
    assert got["unitig_bases"] == exp["total_bases"]
