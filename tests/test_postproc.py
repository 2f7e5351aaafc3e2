"""the reference's downstream helpers (scripts/convertToGFA.py,
split_unitigs.py, pufferize.py, abundance_stats.py) restated as the native host tool
bcalm_amd/_build/bcalm_tools.  PINNED parity: tests/golden/postproc/*/expected.json holds what the
reference's own scripts produced on these inputs in the build container
(tests/golden/make_postproc_golden.py executed them); every output file, stdout, the exit status and
the abort message must match byte for byte.  No GPU involved (pure host code)."""
import glob
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden", "postproc")
CASES = sorted(os.path.basename(os.path.dirname(p)) for p in glob.glob(os.path.join(GOLD, "*", "expected.json")))


@pytest.fixture(scope="module")
def tools():
    import __graft_entry__ as ge
    ge.build_host_tools()
    exe = os.path.join(ROOT, "bcalm_amd", "_build", "bcalm_tools")
    assert os.path.exists(exe)
    return exe


def test_golden_cases_present():
    assert len(CASES) >= 14


@pytest.mark.parametrize("case", CASES)
def test_matches_reference_scripts(tools, case, tmp_path):
    d = os.path.join(GOLD, case)
    exp = json.load(open(os.path.join(d, "expected.json")))
    for label, e in exp["commands"].items():
        work = tmp_path / label.replace(" ", "_")
        work.mkdir()
        shutil.copy(os.path.join(d, "refs.fa"), work)
        shutil.copy(os.path.join(d, "unitigs.fa"), work)
        before = set(os.listdir(work))
        p = subprocess.run([tools, label.split()[0]] + e["args"], cwd=work, capture_output=True, text=True, timeout=60)
        assert p.returncode == e["rc"], (case, label, p.stdout, p.stderr)
        if label == "pufferize" and e["rc"] == 0:
            assert p.stdout.startswith(e["stdout"]), (case, label)          # then hints naming this tool, not the scripts
        else:
            assert p.stdout == e["stdout"], (case, label)
        if e["rc"] != 0:
            assert p.stderr == e["stderr"], (case, label)
        made = {f: open(work / f).read() for f in sorted(os.listdir(work)) if f not in before}
        assert made == e["files"], (case, label)


def test_usage_errors(tools):
    for cmd in ("split_unitigs", "pufferize"):
        p = subprocess.run([tools, cmd], capture_output=True, text=True)
        assert p.returncode == 1 and "arguments: references.fa unitigs.fa k" in p.stderr
    p = subprocess.run([tools, "abundance_stats"], capture_output=True, text=True)
    assert p.returncode == 1 and "arguments: unitigs.fa" in p.stderr
    assert subprocess.run([tools, "nonsense"], capture_output=True).returncode == 1


def test_cli_gfa_equals_converter_on_cli_fasta(tools, tmp_path):
    """The host CLI's own -gfa writer (bcalm_main.cpp) against the pinned converter run on the CLI's FASTA:
    with the converter pinned to scripts/convertToGFA.py above, this pins the -gfa file transitively.
    (simulator build of the CLI: same host source, kernels run on the CPU)"""
    import hostsim_lib
    hostsim_lib.load()
    cli = os.path.join(ROOT, "tests", "hostsim", "_build", "bcalm_hostsim")
    for fx, k in (("pufferize_refs", 9), ("circ_test2", 7), ("minitip", 21), ("rand_a", 15)):
        inp = os.path.join(ROOT, "tests", "golden", "inputs", fx + (".fa" if fx[:4] != "rand" else ".txt"))
        src = tmp_path / (fx + ".fa")
        if fx.startswith("rand"):
            src.write_text("".join(">r%d\n%s\n" % (i, l) for i, l in enumerate(open(inp).read().split("\n")) if l))
        else:
            shutil.copy(inp, src)
        r = subprocess.run([cli, "-in", str(src), "-kmer-size", str(k), "-abundance-min", "1", "-gfa"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout
        fa = tmp_path / (fx + ".unitigs.fa")
        r = subprocess.run([tools, "convertToGFA", str(fa), "conv.gfa", str(k)], cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0
        assert (tmp_path / "conv.gfa").read_text() == (tmp_path / (fx + ".unitigs.gfa")).read_text(), fx
        # the CLI's FASTA is valid input for the other helpers too
        r = subprocess.run([tools, "abundance_stats", str(fa)], capture_output=True, text=True)
        assert r.returncode == 0 and len(r.stdout.splitlines()) >= 2
        r = subprocess.run([tools, "split_unitigs", str(src), str(fa), str(k)], cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0 and os.path.exists(str(fa) + ".split.fa")
