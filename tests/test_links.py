"""unitig links (L: tokens / GFA L lines): GPU hash-join (k_links.h) vs brute force over all
pairs of unitig ends (oracle_py.links), on the simulator (CPU) and on the GPU."""
import os
import random
import sys

import pytest

import oracle_lib

sys.path.insert(0, os.path.join(oracle_lib.ROOT, "oracle"))
import oracle_py as op  # noqa: E402
from bcalm_amd import api  # noqa: E402

CASES = [("pufferize_refs", 9, 1), ("minitip", 21, 1), ("circ_test2", 7, 1), ("rand_a", 15, 2), ("rand_b", 31, 2), ("rand_w2", 55, 2),
         # even k: a k-mer that is its own reverse complement is a unitig whose two ends leave through the same junction (.md:30)
         ("palin4", 4, 1), ("even_k8", 8, 1), ("even_k16", 16, 2), ("even_k32", 32, 2), ("even_k64", 64, 1), ("rand_w3", 77, 1)]


def _check(lib, text, k, amin, **kw):
    g = api.Graph(k, amin, lib=lib, **kw)
    try:
        g.push_text(text); g.run()
        ut = g.unitigs()
        got = set()
        for u, ls in enumerate(g.links()):
            for fs, v, ts in ls:
                assert (u, fs, v, ts) not in got, "duplicate link"
                got.add((u, fs, v, ts))
    finally:
        g.close()
    exp = op.links([s for s, _ in ut], k)
    assert got == exp, (sorted(got - exp)[:5], sorted(exp - got)[:5])
    # mirror constraint (.md:18-30): every edge has its mirror (self-mirrors are their own)
    flip = {"+": "-", "-": "+"}
    for (u, fs, v, ts) in got:
        assert (v, flip[ts], u, flip[fs]) in got
    return len(got)


@pytest.mark.parametrize("name,k,amin", CASES)
@pytest.mark.parametrize("log_np", [0, 5])
def test_links_sim(name, k, amin, log_np):
    import hostsim_lib
    n = _check(hostsim_lib.load(), oracle_lib.read_input(name), k, amin, log2_partitions=log_np)
    if name in ("pufferize_refs", "rand_a"):
        assert n > 0


def test_links_low_complexity_sim():
    import hostsim_lib
    lib = hostsim_lib.load()
    for seed in range(8):
        rng = random.Random(77 + seed)
        g = "".join(rng.choice("AT" if seed % 2 else "ACG") for _ in range(300))
        text = "\n".join(g[i:i + 60] for i in range(0, 240, 17)) + "\n"
        _check(lib, text, rng.choice([5, 7, 9] if seed < 4 else [4, 6, 8]), 1, log2_partitions=3, minimizer_size=3)


@pytest.mark.gpu
@pytest.mark.parametrize("name,k,amin", CASES)
def test_links_gpu(name, k, amin):
    import bcalm_amd
    _check(bcalm_amd.load(), oracle_lib.read_input(name), k, amin)


@pytest.mark.gpu
def test_links_gpu_synthetic(oracle):
    import bcalm_amd
    _check(bcalm_amd.load(), oracle.synth_reads(3000, 150, 3).decode(), 31, 2)
