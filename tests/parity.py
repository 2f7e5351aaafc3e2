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
