"""CPU tests (no GPU): the HIP library loads and exports exactly the C ABI that
include/cdbg.h declares, fails loudly without a device (no CPU fallback), and the
`bcalm` host keeps the reference's CLI contract (exercised through the simulator build)."""
import os
import re
import subprocess
import sys

import pytest

import oracle_lib

ROOT = oracle_lib.ROOT
HDR = open(os.path.join(ROOT, "include", "cdbg.h")).read()
DECLARED = sorted(set(re.findall(r"\b(cdbg_[a-z_]+)\s*\(", HDR)))


def _hip_lib():
    import bcalm_amd
    import __graft_entry__ as ge
    ge.build()                                  # no-op when libcdbg.so is newer than its sources
    return bcalm_amd.load()


def test_header_symbols_exported():
    lib = _hip_lib()
    import bcalm_amd
    assert DECLARED, "no declarations parsed"
    for sym in DECLARED:
        assert hasattr(lib, sym), f"{sym} declared in include/cdbg.h but not exported by libcdbg.so"
    assert sorted(bcalm_amd.EXPORTS) == DECLARED


def test_product_has_no_cpu_fallback():
    """without a HIP device the product refuses to run (on a GPU box it succeeds instead)"""
    import bcalm_amd
    lib = _hip_lib()
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = os.path.exists("/dev/kfd")
    if has_gpu:
        g = bcalm_amd.Graph(21, 1, lib=lib); g.close()
    else:
        with pytest.raises(bcalm_amd.CdbgError) as e:
            bcalm_amd.Graph(21, 1, lib=lib)
        assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def test_package_never_binds_oracle_or_simulator():
    """product sources must not reference the checker or the simulator library"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bcalm_amd")):
        if "_build" in dirpath or "__pycache__" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")) and f != "hostsim.h" and f != "devrt.h":
                txt = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in txt and "oracle_lib" not in txt and "libcdbg_hostsim" not in txt, f


@pytest.fixture(scope="module")
def cli():
    import hostsim_lib
    hostsim_lib.load()
    exe = os.path.join(ROOT, "tests", "hostsim", "_build", "bcalm_hostsim")
    assert os.path.exists(exe)
    return exe


def _parse_fa(path):
    recs = []
    lines = open(path).read().split("\n")
    for i in range(0, len(lines) - 1, 2):
        h = lines[i]
        m = re.match(r">(\d+) LN:i:(\d+) KC:i:(\d+) km:f:(\d+\.\d)((?: L:[+-]:\d+:[+-])*) $", h)
        assert m, h
        recs.append((lines[i + 1], int(m.group(2)), int(m.group(3)), float(m.group(4))))
    return recs


def test_cli_contract(cli, oracle, tmp_path):
    inp = os.path.join(ROOT, "tests", "golden", "inputs", "pufferize_refs.fa")
    r = subprocess.run([cli, "-in", inp, "-kmer-size", "9", "-abundance-min", "1", "-minimizer-size", "5"],
                       cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    fa = tmp_path / "pufferize_refs.unitigs.fa"        # <basename of -in>.unitigs.fa  (bcalm_1.cpp:68-74)
    assert fa.exists()
    recs = _parse_fa(fa)
    exp = oracle.run(oracle_lib.read_input("pufferize_refs"), 9, 1)
    assert oracle_lib.canonical_set(oracle, [(s, kc) for s, _, kc, _ in recs], 9) == exp["unitigs"]
    for s, ln, kc, km in recs:
        assert ln == len(s) and abs(km - round(kc / (ln - 9 + 1), 1)) < 1e-9
    # -out prefix, gz input, FASTQ input
    import gzip
    fq = tmp_path / "r.fastq.gz"
    with gzip.open(fq, "wt") as f:
        f.write("@a\nACTGATGCAGATGACACTGATGCAGATGAC\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n@b\nATGACACTGATGCAGATGACAGTAGTGGGG\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n")
    r = subprocess.run([cli, "-in", str(fq), "-kmer-size", "21", "-abundance-min", "1", "-out", "xyz", "-gfa"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    recs = _parse_fa(tmp_path / "xyz.unitigs.fa")
    exp = oracle.run("ACTGATGCAGATGACACTGATGCAGATGAC\nATGACACTGATGCAGATGACAGTAGTGGGG\n", 21, 1)
    assert oracle_lib.canonical_set(oracle, [(s, kc) for s, _, kc, _ in recs], 21) == exp["unitigs"]
    gfa = (tmp_path / "xyz.unitigs.gfa").read_text()
    assert gfa.startswith("H\tVN:Z:1.0\tks:i:21\n")
    # L tokens of the FASTA header and GFA L lines agree; overlap is (k-1)M (convertToGFA.py:103-112)
    fa_links = re.findall(r" L:([+-]):(\d+):([+-])", (tmp_path / "xyz.unitigs.fa").read_text())
    gfa_links = re.findall(r"^L\t\d+\t([+-])\t(\d+)\t([+-])\t20M$", gfa, flags=re.M)
    assert sorted(fa_links) == sorted(gfa_links)


def test_cli_wrapped_fastq(cli, oracle, tmp_path):
    """FASTQ whose sequence and quality wrap over several lines, with quality lines that start with '@' and '+'
    (README.md:45-50 takes any FASTQ): the result equals that of the unwrapped reads"""
    reads = ["ACTGATGCAGATGACACTGATGCAGATGACAGTAGTGGGG", "ATGACACTGATGCAGATGACAGTAGTGGGGTTTACG", "GATTACAGATTACAGATTACACCCGT"]
    with open(tmp_path / "w.fastq", "w") as f:
        for i, r in enumerate(reads):
            f.write("@read%d some text\n" % i)
            for j in range(0, len(r), 11):
                f.write(r[j:j + 11] + "\n")
            f.write("+read%d\n" % i)
            q = ("@+I" * len(r))[:len(r)]
            for j in range(0, len(r), 7):
                f.write(q[j:j + 7] + "\n")
    r = subprocess.run([cli, "-in", "w.fastq", "-kmer-size", "21", "-abundance-min", "1"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    recs = _parse_fa(tmp_path / "w.unitigs.fa")
    exp = oracle.run("\n".join(reads) + "\n", 21, 1)
    assert oracle_lib.canonical_set(oracle, [(s, kc) for s, _, kc, _ in recs], 21) == exp["unitigs"]
    (tmp_path / "bad.fastq").write_text("@a\nACGT\n+\nIIII\nACGT\n")
    r = subprocess.run([cli, "-in", "bad.fastq", "-kmer-size", "3", "-abundance-min", "1"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "malformed FASTQ" in r.stdout


def test_cli_solid_kmers_out(cli, oracle, tmp_path):
    """-solid-kmers-out (hidden option of the reference, bcalm_1.cpp:37): canonical k-mer + abundance per line"""
    inp = os.path.join(ROOT, "tests", "golden", "inputs", "minitip.fa")
    r = subprocess.run([cli, "-in", inp, "-kmer-size", "21", "-abundance-min", "2", "-solid-kmers-out", "solid.txt"],
                       cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    got = sorted((a, int(b)) for a, b in (l.split() for l in (tmp_path / "solid.txt").read_text().splitlines()))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    exp = oracle_py.solid_kmers(oracle_py.read_fasta_text(inp), 21, 2)
    assert got == exp and len(got) == 20


def test_cli_errors(cli, tmp_path):
    r = subprocess.run([cli, "-kmer-size", "21"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "EXCEPTION: Specifiy -in" in r.stdout          # main.cpp:44-48, bcalm_1.cpp:61
    r = subprocess.run([cli, "-in", "/nonexistent.fa"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "EXCEPTION:" in r.stdout
    r = subprocess.run([cli, "-in", os.path.join(ROOT, "tests", "golden", "inputs", "tiny_read.fa"), "-kmer-size", "256"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "out of range" in r.stdout
    r = subprocess.run([cli, "-in", os.path.join(ROOT, "tests", "golden", "inputs", "tiny_read.fa"), "-kmer-size", "20"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout                                       # even k runs (README.md:99)
    r = subprocess.run([cli, "-v"], capture_output=True, text=True)
    assert r.returncode == 0 and "version" in r.stdout


def _rand_reads(seed, n, lo, hi, glen=3000):
    import random
    rng = random.Random(seed)
    g = "".join(rng.choice("ACGT") for _ in range(glen))
    comp = str.maketrans("ACGT", "TGCA")
    out = []
    for _ in range(n):
        L = rng.randrange(lo, hi); s = rng.randrange(0, glen - L)
        r = g[s:s + L]
        if rng.random() < 0.5:
            r = r.translate(comp)[::-1]
        if rng.random() < 0.1:
            p = rng.randrange(L); r = r[:p] + "N" + r[p + 1:]
        out.append(r)
    return out


def _cli_set(cli, oracle, tmp_path, args, k, env=None):
    e = dict(os.environ); e.update(env or {})
    r = subprocess.run([cli] + args + ["-kmer-size", str(k), "-abundance-min", "1", "-out", "o"], cwd=tmp_path, capture_output=True, text=True, env=e)
    assert r.returncode == 0, r.stdout + r.stderr
    recs = _parse_fa(tmp_path / "o.unitigs.fa")
    return oracle_lib.canonical_set(oracle, [(s, kc) for s, _, kc, _ in recs], k), r.stdout


@pytest.mark.parametrize("stage_bytes", ["100000", "192"])
def test_cli_parallel_fasta_slices(cli, oracle, tmp_path, stage_bytes):
    """a plain FASTA (wrapped lines, comment lines, CRLF) cut into many record-aligned slices for 5 parser threads that write straight into
    the library's staging buffers (cdbg_stage_acquire / _commit); with 192-byte buffers every 300 .. 900 bp sequence is continued over
    several buffers with a k-1 overlap: same unitig set as the oracle on the plain reads"""
    reads = _rand_reads(7, 120, 300, 900)
    with open(tmp_path / "in.fa", "w", newline="") as f:
        for i, r in enumerate(reads):
            f.write(">r%d x>y\r\n" % i if i % 3 == 0 else ">r%d\n" % i)
            if i % 7 == 0:
                f.write(";comment\n")
            for j in range(0, len(r), 61):
                f.write(r[j:j + 61] + ("\r\n" if i % 3 == 0 else "\n"))
    exp = oracle.run("\n".join(reads) + "\n", 21, 1)
    got, out = _cli_set(cli, oracle, tmp_path, ["-in", "in.fa", "-nb-cores", "5"], 21, {"BCALM_SLICE_BYTES": "700", "CDBG_STAGE_BYTES": stage_bytes})
    assert got == exp["unitigs"]
    assert "input: 120 sequences, %d bases" % sum(len(r) for r in reads) in out


def test_cli_parallel_fastq_slices_and_fallback(cli, oracle, tmp_path):
    """strict four-line FASTQ in slices (quality lines that start with '@' and '+' must not be taken for headers); a file whose FIRST
    record is four lines but a later one wraps falls back to the tolerant serial parser; a file list of plain and gzip files"""
    import gzip
    reads = _rand_reads(11, 150, 60, 200)
    with open(tmp_path / "s.fastq", "w") as f:
        for i, r in enumerate(reads):
            q = ("@+I@" * len(r))[:len(r)] if i % 2 else ("+@" * len(r))[:len(r)]
            f.write("@r%d\n%s\n+\n%s\n" % (i, r, q))
    exp = oracle.run("\n".join(reads) + "\n", 21, 1)
    got, out = _cli_set(cli, oracle, tmp_path, ["-in", "s.fastq", "-nb-cores", "4"], 21, {"BCALM_SLICE_BYTES": "500", "CDBG_STAGE_BYTES": "4096"})
    assert got == exp["unitigs"] and "input: 150 sequences" in out
    with open(tmp_path / "mixed.fastq", "w") as f:
        for i, r in enumerate(reads):
            if i == 70:                                  # one wrapped record in the middle
                f.write("@r%d\n%s\n%s\n+\n%s\n%s\n" % (i, r[:30], r[30:], "I" * 30, "@" * (len(r) - 30)))
            else:
                f.write("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)))
    got, out = _cli_set(cli, oracle, tmp_path, ["-in", "mixed.fastq", "-nb-cores", "4"], 21, {"BCALM_SLICE_BYTES": "500"})
    assert got == exp["unitigs"] and "input: 150 sequences" in out
    # file list: two plain FASTA parts and one gzip FASTQ part
    parts = [reads[:50], reads[50:100], reads[100:]]
    with open(tmp_path / "p0.fa", "w") as f:
        f.write("".join(">a\n%s\n" % r for r in parts[0]))
    with open(tmp_path / "p1.fa", "w") as f:
        f.write("".join(">b\n%s\n" % r for r in parts[1]))
    with gzip.open(tmp_path / "p2.fq.gz", "wt") as f:
        f.write("".join("@c\n%s\n+\n%s\n" % (r, "I" * len(r)) for r in parts[2]))
    (tmp_path / "list.txt").write_text("p0.fa\np1.fa\np2.fq.gz\n")
    got, out = _cli_set(cli, oracle, tmp_path, ["-in", "list.txt", "-nb-cores", "3"], 21, {"BCALM_SLICE_BYTES": "900"})
    assert got == exp["unitigs"] and "input: 150 sequences" in out


def test_stage_calls_through_the_abi(oracle):
    """cdbg_stage_acquire / cdbg_stage_commit from Python (simulator build): same graph as cdbg_push_text; mixing both; a buffer handed back unused"""
    import hostsim_lib
    from bcalm_amd import api
    sim = hostsim_lib.load()
    text = oracle.synth_reads(800, 150, 3)
    exp = oracle.run(text, 31, 2)
    g = api.Graph(31, 2, lib=sim)
    half = text.rfind(b"\n", 0, len(text) // 2) + 1
    g.stage_text(text[:half]); g.push_text(text[half:])
    buf, cap = api.C.c_void_p(), api.C.c_uint64()
    assert sim.cdbg_stage_acquire(g._h, api.C.byref(buf), api.C.byref(cap)) == 0 and cap.value >= 64
    assert sim.cdbg_stage_commit(g._h, buf, 0) == 0
    assert sim.cdbg_stage_commit(g._h, buf, 0) != 0          # not held any more
    g.run()
    assert oracle_lib.canonical_set(oracle, g.unitigs(), 31) == exp["unitigs"]
    g.close()
