"""Shared parity helpers: run a libcdbg build (HIP or simulator) and compare with the oracle."""
import oracle_lib
from bcalm_amd import api


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
    """BASELINE config 2 shape: one 4.64 Mbp sequence (E. coli MG1655 length; the FASTA itself is not available
    offline, so a seeded synthetic genome with planted direct and inverted repeats stands in) -> bytes with '\n'"""
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
