"""Shared parity helpers: run a libcdbg build (HIP or simulator) and compare with the oracle -- and, whenever a real
BCALM 2 binary is reachable ($BCALM_BIN or `bcalm` on PATH), with the reference itself (the only door out of "parity
unpinned": /root/reference/test/simple_test.sh:5-9 does exactly this diff)."""
import os
import re
import shutil
import subprocess
import tempfile

import oracle_lib
from bcalm_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_binary():
    """a real BCALM 2 executable, or None: $BCALM_BIN, else `bcalm` on PATH -- never this repo's own CLI"""
    cand = os.environ.get("BCALM_BIN") or shutil.which("bcalm")
    own = {os.path.realpath(os.path.join(ROOT, "bcalm_amd", "_build", "bcalm")),
           os.path.realpath(os.path.join(ROOT, "tests", "hostsim", "_build", "bcalm_hostsim"))}
    if cand and os.path.isfile(cand) and os.access(cand, os.X_OK) and os.path.realpath(cand) not in own:
        return cand
    return None


def reference_unitigs(ref_bin, text, k, amin, cores=None, timeout=3600):
    """run the reference on a FASTA dump of `text` (bytes or str, reads separated by any non-ACGT byte) exactly as its README
    says (README.md:11: bcalm -in X -kmer-size K -abundance-min A) and parse <prefix>.unitigs.fa (README.md:62-72)
    -> ([(sequence, KC)], seconds)"""
    import time
    if isinstance(text, bytes):
        text = text.decode()
    with tempfile.TemporaryDirectory() as t:
        fa = os.path.join(t, "reads.fa")
        with open(fa, "w") as f:
            for i, r in enumerate(x for x in re.split(r"[^ACGTacgt]+", text) if x):
                f.write(">r%d\n%s\n" % (i, r))
        t0 = time.time()
        subprocess.run([ref_bin, "-in", fa, "-kmer-size", str(k), "-abundance-min", str(amin), "-nb-cores", str(cores or os.cpu_count() or 1)],
                       cwd=t, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout)
        dt = time.time() - t0
        out = []
        with open(os.path.join(t, "reads.unitigs.fa")) as f:
            kc = None
            for line in f:
                if line.startswith(">"):
                    kc = int(re.search(r"KC:i:(\d+)", line).group(1))
                elif line.strip():
                    out.append((line.strip(), kc))
    return out, dt


def diff_against_reference(oracle, ref_bin, text, k, amin, ours):
    """canonical (sequence, KC) sets: ours (list of (seq, KC)) vs the reference binary's output on the same reads"""
    ref, dt = reference_unitigs(ref_bin, text, k, amin)
    a = oracle_lib.canonical_set(oracle, ours, k)
    b = oracle_lib.canonical_set(oracle, ref, k)
    sa, sb = set(a), set(b)
    return {"equal": a == b, "ours": len(a), "reference": len(b), "only_ours": sorted(sa - sb)[:5], "only_reference": sorted(sb - sa)[:5],
            "reference_seconds": dt, "reference_binary": ref_bin}


def run_graph(lib, text, k, amin, **kw):
    g = api.Graph(k, amin, lib=lib, **kw)
    try:
        g.push_text(text)
        g.count()
        solid = g.solid_kmers()
        g.compact()
        g.glue()
        return {"stats": g.stats(), "solid": solid, "unitigs": g.unitigs()}
    finally:
        g.close()


def assert_parity(oracle, lib, text, k, amin, **kw):
    exp = oracle.run(text, k, amin, want_solid=True)
    got = run_graph(lib, text, k, amin, **kw)
    st = got["stats"]
    assert st["n_occurrences"] == exp["stats"]["occurrences"], (st, exp["stats"])
    assert st["n_distinct"] == exp["stats"]["distinct"], (st, exp["stats"])
    assert st["n_solid"] == exp["stats"]["solid"], (st, exp["stats"])
    assert got["solid"] == exp["solid"], "stage-1 (k-mer, count) set differs"
    canon = oracle_lib.canonical_set(oracle, got["unitigs"], k)
    if canon != exp["unitigs"]:
        only_got = sorted(set(canon) - set(exp["unitigs"]))[:5]
        only_exp = sorted(set(exp["unitigs"]) - set(canon))[:5]
        raise AssertionError(f"unitig sets differ: got {len(canon)} exp {len(exp['unitigs'])}\n only got: {only_got}\n only exp: {only_exp}\n stats={st}")
    assert st["n_unitigs"] == exp["stats"]["unitigs"]
    assert oracle_lib.digest_of(canon) == exp["digest"]
    return got


def config2_genome(oracle):
    """BASELINE config 2: $CDBG_ECOLI_FASTA (E. coli MG1655, one sequence) when set; otherwise -- the FASTA is not
    available offline -- a seeded synthetic genome of the same length with planted direct and inverted repeats -> bytes with '\n'"""
    real = os.environ.get("CDBG_ECOLI_FASTA")
    if real and os.path.isfile(real):                    # the real NC_000913.3 when the box has it
        import gzip
        op = gzip.open if real.endswith(".gz") else open
        with op(real, "rt") as f:
            seq = "".join(line.strip() for line in f if not line.startswith(">"))
        return seq.upper().encode() + b"\n"
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    g0 = bytearray(oracle.synth_reads(1, 4_641_652, 2)[:-1])
    for i in range(7):                                   # 7 x 5 kbp repeats, alternating strands
        seg = bytes(g0[100_000 + 50_000 * i:105_000 + 50_000 * i])
        if i & 1:
            seg = seg.translate(comp)[::-1]
        pos = 2_000_000 + 300_000 * i
        g0[pos:pos + 5000] = seg
    for i in range(20):                                  # 20 x 1.3 kbp repeats
        seg = bytes(g0[3_000_000 + 7_000 * i:3_001_300 + 7_000 * i])
        if i % 3 == 0:
            seg = seg.translate(comp)[::-1]
        pos = 500_000 + 60_000 * i
        g0[pos:pos + 1300] = seg
    return bytes(g0) + b"\n"


M64 = (1 << 64) - 1


def _mix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def set_digest(unitigs):
    """the formula of cdbg_digest (bcalm_amd/csrc/k_links.h k_digest_unitigs) over [(sequence, KC)]"""
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    B = 0x100000001B3
    tot = 0
    for s, kc in unitigs:
        hf = hr = 0
        n = len(s)
        for i in range(n):
            hf = (hf * B + code[s[i]] + 1) & M64
            hr = (hr * B + (3 - code[s[n - 1 - i]]) + 1) & M64
        tot = (tot + _mix64(((hf + hr) & M64) ^ _mix64(((hf * hr) + kc) & M64))) & M64
    return tot


def kmer_set_sums(kmers, k):
    """the formula of cdbg_verify (bcalm_amd/csrc/k_verify.h verify_mix) over canonical k-mers given as strings:
    (count, sum of mix A, sum of mix B) -- pins what the device reports to the k-mer SET, not to the device's own code"""
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    W = k // 32 + 1
    n = sa = sb = 0
    for s in kmers:
        v = 0
        for ch in s:
            v = (v << 2) | code[ch]
        h = 0x243F6A8885A308D3
        for i in range(W):
            h = _mix64(h ^ ((v >> (64 * i)) & M64))
        n += 1; sa = (sa + h) & M64; sb = (sb + _mix64(h ^ 0xA4093822299F31D0)) & M64
    return (n, sa, sb)


def assert_verified(g):
    """the oracle-independent device check of a finished graph: unitig k-mers == solid set, every unitig maximal"""
    v = g.verify()
    assert v["unitig_kmers"] == v["solid_kmers"], v
    assert v["mergeable_ends"] in (0, None), v
    assert api.Graph.edges_conserved(v), v               # every inner junction of every unitig 1-in / 1-out (cdbg_verify_edges)
    return v


_COMP = str.maketrans("ACGT", "TGCA")


def _rc(s):
    return s.translate(_COMP)[::-1]


def check_edge_conservation_is_sensitive(g, k):
    """cdbg_verify_edges / cdbg_verify_unitigs on a finished graph `g` that has at least one branching junction:
    (1) the resident set and the same set handed back by the caller give the same numbers and conserve the edges;
    (2) a unitig merged with a neighbour THROUGH a branching junction (over-compaction: same k-mer set, no mergeable ends
        added) breaks the conservation and nothing else;
    (3) a unitig cut in two (under-compaction) keeps the conservation and shows up as a mergeable pair instead.
    -> number of planted merges that were tried"""
    uni = [s for s, _ in g.unitigs()]
    links = g.links()
    v0 = g.verify()
    assert v0["unitig_kmers"] == v0["solid_kmers"] and v0["mergeable_ends"] == 0 and api.Graph.edges_conserved(v0), v0
    v1 = g.verify_unitigs(uni)
    assert v1 == v0, (v1, v0)
    planted = 0
    for u, lk in enumerate(links):
        for side in "+-":
            out = [(v, ts) for fs, v, ts in lk if fs == side and v != u]
            if len(out) < 2:
                continue                                  # (not a branching junction on this side)
            v, ts = out[0]
            head = uni[u] if side == "+" else _rc(uni[u])
            tail = uni[v] if ts == "+" else _rc(uni[v])
            assert head[-(k - 1):] == tail[:k - 1], "a link is a (k-1)-overlap"
            merged = head + tail[k - 1:]
            bad = [s for i, s in enumerate(uni) if i not in (u, v)] + [merged]
            vb = g.verify_unitigs(bad)
            assert vb["unitig_kmers"] == vb["solid_kmers"], "the planted merge keeps the k-mer set"
            assert not api.Graph.edges_conserved(vb), ("over-compaction not seen", vb)
            assert vb["edges"]["graph"] > vb["edges"]["links"] + vb["edges"]["inner"], vb
            planted += 1
            if planted >= 3:
                break
        if planted >= 3:
            break
    long_ones = [i for i, s in enumerate(uni) if len(s) >= k + 1]
    if long_ones:
        i = long_ones[0]
        cut = [s for j, s in enumerate(uni) if j != i] + [uni[i][:k], uni[i][1:]]
        vc = g.verify_unitigs(cut)
        assert vc["unitig_kmers"] == vc["solid_kmers"] and api.Graph.edges_conserved(vc), vc
        assert vc["mergeable_ends"] == 2, vc
    return planted
