"""-all-abundance-counts (/root/reference/README.md:74-80): the ab:Z: vector of every unitig must be
the oracle's count of each of its k-mers, in the orientation of the emitted sequence."""
import os
import re
import subprocess

import pytest

import oracle_lib
from bcalm_amd import api

CASES = [("minitip", 21, 1), ("minitip", 21, 2), ("rand_a", 15, 2), ("rand_b", 31, 2), ("circ_test1", 7, 1), ("rand_w2", 55, 2)]
COMP = str.maketrans("ACGT", "TGCA")


def _check(lib, oracle, text, k, amin, **kw):
    exp = dict(oracle.run(text, k, amin, want_solid=True)["solid"])
    g = api.Graph(k, amin, lib=lib, all_abundance_counts=True, **kw)
    try:
        g.push_text(text); g.run()
        ut = g.unitigs(); ab = g.unitig_abundances()
    finally:
        g.close()
    assert len(ut) == len(ab)
    for (s, kc), a in zip(ut, ab):
        assert len(a) == len(s) - k + 1
        want = []
        for i in range(len(s) - k + 1):
            x = s[i:i + k]; r = x.translate(COMP)[::-1]
            want.append(exp[min(x, r)])
        assert a == want
        assert sum(a) == kc


@pytest.mark.parametrize("name,k,amin", CASES)
@pytest.mark.parametrize("log_np", [0, 5])
def test_abundances_sim(oracle, name, k, amin, log_np):
    import hostsim_lib
    _check(hostsim_lib.load(), oracle, oracle_lib.read_input(name), k, amin, log2_partitions=log_np)


def test_cli_all_abundance_counts(oracle, tmp_path):
    import hostsim_lib
    hostsim_lib.load()
    exe = os.path.join(oracle_lib.ROOT, "tests", "hostsim", "_build", "bcalm_hostsim")
    inp = os.path.join(oracle_lib.ROOT, "tests", "golden", "inputs", "minitip.fa")
    r = subprocess.run([exe, "-in", inp, "-kmer-size", "21", "-abundance-min", "1", "-all-abundance-counts"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    heads = [l for l in (tmp_path / "minitip.unitigs.fa").read_text().split("\n") if l.startswith(">")]
    assert len(heads) == 3
    for h in heads:
        m = re.match(r">\d+ LN:i:(\d+) ab:Z:((?:\d+ ?)+)", h)
        assert m, h
        vals = [int(x) for x in m.group(2).split()]
        assert len(vals) == int(m.group(1)) - 21 + 1 and set(vals) <= {1, 3}


@pytest.mark.gpu
@pytest.mark.parametrize("name,k,amin", CASES)
def test_abundances_gpu(oracle, name, k, amin):
    import bcalm_amd
    _check(bcalm_amd.load(), oracle, oracle_lib.read_input(name), k, amin)


@pytest.mark.gpu
def test_abundances_gpu_synthetic(oracle):
    import bcalm_amd
    _check(bcalm_amd.load(), oracle, oracle.synth_reads(5000, 150, 3), 31, 2)
