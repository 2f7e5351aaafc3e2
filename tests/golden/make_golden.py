#!/usr/bin/env python3
"""Regenerates tests/golden/*.json.

Inputs: the reference's own fixture FASTA files (data, copied verbatim into
tests/golden/inputs/ because /root/reference does not exist on the GPU box):
  example/tiny_read.fa, test/minitip.fa,
  example/circular_unitigs_unittests/test{1,2,3}.fa, example/pufferize/refs.fa
The reference stores NO expected outputs for them, so the
expected unitig sets are produced by oracle/oracle_py.py (spec restatement) and are
cross-checked in tests/test_oracle.py against (a) oracle/cdbg_oracle.c and (b) the
hand-transcribed anchor table of SURVEY.md section 4 (anchors.json, written here
from the literal table, not from any oracle).

Run from the repo root in the build container:  python tests/golden/make_golden.py
"""
import hashlib, json, os, shutil, sys, random
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as op

REF = "/root/reference"
FIXTURES = {
    "tiny_read": "example/tiny_read.fa",
    "minitip": "test/minitip.fa",
    "circ_test1": "example/circular_unitigs_unittests/test1.fa",
    "circ_test2": "example/circular_unitigs_unittests/test2.fa",
    "circ_test3": "example/circular_unitigs_unittests/test3.fa",
    "pufferize_refs": "example/pufferize/refs.fa",
}
# (fixture, k, abundance_min) cases: the ks the reference's own scripts use
# (example/run-tiny.sh:2 k=13; BASELINE config 1 k=21; CMD:4,8 k=7; run.sh:1 k=9)
CASES = [
    ("tiny_read", 13, 1), ("tiny_read", 21, 1),
    ("minitip", 21, 1), ("minitip", 21, 2),
    ("circ_test1", 7, 1), ("circ_test2", 7, 1), ("circ_test3", 7, 1),
    ("pufferize_refs", 9, 1),
    # even k (README.md:99 "any k value"): a k-mer may equal its reverse complement (.md:30,57)
    ("tiny_read", 12, 1), ("minitip", 20, 1), ("minitip", 20, 2), ("circ_test1", 6, 1), ("circ_test2", 8, 1),
    ("circ_test3", 6, 1), ("pufferize_refs", 8, 1), ("pufferize_refs", 10, 1), ("palin4", 4, 1),
    # the spec's own worked example (bidirected-graphs-in-bcalm2.md:64: k = 3, S = {GTATAC}; inputs/spec_gtatac.fa holds that string)
    ("spec_gtatac", 3, 1),
]

# SURVEY.md section 4 anchor table, transcribed by hand (seq, LN, KC); circular
# entries give only (n_kmers, LN, KC) because the rotation is arbitrary.
ANCHORS = {
    "tiny_read/13/1": {"distinct": 13, "solid": 13, "unitigs": [["ACCCACACATGACTCAGTCAGCAGT", 25, 13]]},
    "tiny_read/21/1": {"distinct": 5, "solid": 5, "unitigs": [["ACCCACACATGACTCAGTCAGCAGT", 25, 5]]},
    "minitip/21/1": {"distinct": 21, "solid": 21, "unitigs": [
        ["ACTGATGCAGATGACACTGATGCAGATGAC", 30, 30], ["ATGACACTGATGCAGATGACAGTAGTGGGG", 30, 30],
        ["AGTCATCTGCATCAGTGTCAT", 21, 1]]},
    "minitip/21/2": {"distinct": 21, "solid": 20, "unitigs": [["ACTGATGCAGATGACACTGATGCAGATGACAGTAGTGGGG", 40, 60]]},
    "circ_test1/7/1": {"distinct": 9, "solid": 9, "circular": [[9, 15, 10]]},
    "circ_test3/7/1": {"distinct": 9, "solid": 9, "circular": [[9, 15, 10]]},
    "circ_test2/7/1": {"distinct": 13, "solid": 13, "unitigs": [["ACCATGATTCAGAAAAAA", 18, 12], ["AAAAAAA", 7, 3]]},
    # even k, by hand from .md:7,41-46: CAATTG has the 4-mers CAAT, AATT, ATTG = rc(CAAT): nodes ATTG (seen twice) and AATT, which
    # is its own reverse complement.  ATTG reaches AATT by the two distinct edges (ATTG,AATT,-,+) and (ATTG,AATT,-,-), so it has no
    # unique out-edge and neither node extends: two unitigs.  ACGTAC: ACGT (palindrome), CGTA, GTAC (palindrome): three unitigs
    "palin4/4/1": {"distinct": 5, "solid": 5, "unitigs": [["AATT", 4, 1], ["ATTG", 4, 2], ["ACGT", 4, 1], ["CGTA", 4, 1], ["GTAC", 4, 1]]},
    # the spec's worked example, .md:64-68 + :78: k = 3, S = {GTATAC} -- two nodes, {GTA, TAC} and {ATA, TAT}, each seen twice, and the walk
    # (e2, e3, e1) spells the input.  By hand: GTA overlaps TAT and TAC by its suffix TA (two outgoing edges, one of them the self-mirror
    # GTA -> TAC) and nothing ends in GT; ATA -> TAT, ATA -> TAC, TAT -> ATA: neither node has a unique edge on either side: two unitigs
    "spec_gtatac/3/1": {"distinct": 2, "solid": 2, "unitigs": [["GTA", 3, 2], ["ATA", 3, 2]]},
    "pufferize_refs/9/1": {"distinct": 70, "solid": 70, "unitigs_partial": [["AATTGGTCT", 9, 2], ["ATTGGTCTGGTTGGATTGTACTCATGATG", 29, 21]],
                           "n_unitigs": 3, "other": [[56, 49]]},
}

def random_genome_case(seed, glen, nreads, rlen, err, k, amin):
    rng = random.Random(seed)
    g = "".join(rng.choice("ACGT") for _ in range(glen))
    # plant a repeat and an inverted repeat so the graph branches
    rep = g[100:160]
    g = g[:400] + rep + g[400:700] + op.revcomp(rep) + g[700:]
    reads = []
    for _ in range(nreads):
        s = rng.randrange(0, len(g) - rlen + 1)
        r = list(g[s:s + rlen])
        for i in range(rlen):
            if rng.random() < err:
                r[i] = rng.choice([c for c in "ACGT" if c != r[i]])
        r = "".join(r)
        if rng.random() < 0.5:
            r = op.revcomp(r)
        if rng.random() < 0.02:
            p = rng.randrange(rlen); r = r[:p] + "N" + r[p + 1:]
        reads.append(r)
    return "\n".join(reads) + "\n"

def palindrome_case(seed, glen, nreads, rlen, err, k, amin):
    """even k: random genome with planted k-mers that equal their reverse complement (w + rc(w)), alone, back to back,
    inside a repeat and at read ends; optional substitutions"""
    rng = random.Random(seed)
    g = [rng.choice("ACGT") for _ in range(glen)]
    pals = []
    for i in range(8):
        w = "".join(rng.choice("ACGT") for _ in range(k // 2))
        pals.append(w + op.revcomp(w))
    pos = sorted(rng.sample(range(k, glen - 3 * k), 10))
    for i, p in enumerate(pos):
        q = pals[i % len(pals)] + (pals[(i + 1) % len(pals)] if i % 4 == 3 else "")
        g[p:p + len(q)] = list(q)
    g = "".join(g)
    reads = []
    for i in range(nreads):
        if i < len(pos):                                   # a read that ends exactly on a planted palindrome
            s = max(0, pos[i] + k - rlen)
        else:
            s = rng.randrange(0, len(g) - rlen + 1)
        r = list(g[s:s + rlen])
        for j in range(len(r)):
            if rng.random() < err:
                r[j] = rng.choice([c for c in "ACGT" if c != r[j]])
        r = "".join(r)
        if rng.random() < 0.5:
            r = op.revcomp(r)
        reads.append(r)
    reads.append(pals[0])                                  # a palindromic k-mer as a read of its own
    return "\n".join(reads) + "\n"

def solid_entry(text, k, amin):
    """small sets verbatim; large ones as sha256 of 'KMER COUNT\n' lines (sorted)"""
    sk = op.solid_kmers(text, k, amin)
    blob = "".join(f"{x} {c}\n" for x, c in sk).encode()
    e = {"n": len(sk), "sha256": hashlib.sha256(blob).hexdigest()}
    if len(sk) <= 80:
        e["list"] = sk
    return e

def main():
    os.makedirs(os.path.join(HERE, "inputs"), exist_ok=True)
    for name, rel in FIXTURES.items():
        src = os.path.join(REF, rel)
        if os.path.exists(src):
            shutil.copyfile(src, os.path.join(HERE, "inputs", name + ".fa"))
    golden = {}
    for name, k, amin in CASES:
        text = op.read_fasta_text(os.path.join(HERE, "inputs", name + ".fa"))
        u, st = op.unitigs(text, k, amin)
        golden[f"{name}/{k}/{amin}"] = {"stats": st, "unitigs": u,
                                         "solid": solid_entry(text, k, amin)}
    # seeded random read sets (synthetic; written out as input fixtures too)
    for tag, args in {"rand_a": (11, 1500, 400, 60, 0.01, 15, 2),
                      "rand_b": (12, 2500, 500, 80, 0.005, 31, 2),
                      "rand_c": (13, 1200, 150, 100, 0.0, 21, 1),
                      "rand_w2": (14, 3000, 300, 120, 0.004, 55, 2),
                      "rand_w4": (15, 3000, 120, 300, 0.002, 127, 1)}.items():
        text = random_genome_case(*args[:5], args[5], args[6])
        with open(os.path.join(HERE, "inputs", tag + ".txt"), "w") as f:
            f.write(text)
        k, amin = args[5], args[6]
        u, st = op.unitigs(text, k, amin)
        golden[f"{tag}/{k}/{amin}"] = {"stats": st, "unitigs": u, "solid": solid_entry(text, k, amin)}
    # even k with planted palindromic k-mers (W = 1, 2, 3, 4), and one odd three-word case
    for tag, args in {"even_k4": (21, 300, 60, 30, 0.0, 4, 1),
                      "even_k8": (22, 900, 150, 50, 0.01, 8, 1),
                      "even_k16": (23, 1500, 300, 60, 0.01, 16, 2),
                      "even_k32": (24, 2500, 300, 100, 0.004, 32, 2),
                      "even_k64": (25, 3000, 200, 160, 0.003, 64, 1),
                      "even_k96": (26, 3000, 150, 250, 0.002, 96, 1),
                      "even_k126": (27, 3000, 120, 300, 0.002, 126, 1)}.items():
        text = palindrome_case(*args)
        with open(os.path.join(HERE, "inputs", tag + ".txt"), "w") as f:
            f.write(text)
        k, amin = args[5], args[6]
        u, st = op.unitigs(text, k, amin)
        golden[f"{tag}/{k}/{amin}"] = {"stats": st, "unitigs": u, "solid": solid_entry(text, k, amin)}
    text = random_genome_case(16, 3000, 150, 250, 0.003, 77, 1)
    with open(os.path.join(HERE, "inputs", "rand_w3.txt"), "w") as f:
        f.write(text)
    u, st = op.unitigs(text, 77, 1)
    golden["rand_w3/77/1"] = {"stats": st, "unitigs": u, "solid": solid_entry(text, 77, 1)}
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(golden, f, indent=0, sort_keys=True)
    with open(os.path.join(HERE, "anchors.json"), "w") as f:
        json.dump(ANCHORS, f, indent=1, sort_keys=True)
    print("wrote", len(golden), "golden cases")

if __name__ == "__main__":
    main()
