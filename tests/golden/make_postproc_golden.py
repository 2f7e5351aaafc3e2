#!/usr/bin/env python3
"""Regenerates tests/golden/postproc/ -- golden vectors for SURVEY.md section 8 row f4.

Unlike the unitig construction itself (whose reference implementation, gatb-core, is absent),
the reference's downstream helpers are Python scripts that run in the build container:
    /root/reference/scripts/convertToGFA.py, split_unitigs.py, pufferize.py, abundance_stats.py
This script EXECUTES them (subprocess, untouched, from where they lie) on small inputs and
stores what they produced: every output file's bytes, stdout and the exit status.  Only data is
committed (inputs + expected outputs); no line of the scripts is.  tests/test_postproc.py then
requires bcalm_amd/_build/bcalm_tools to reproduce each vector byte for byte.

Inputs: unitig FASTA files in BCALM's format (>id LN:i: KC:i: km:f: L:...), written here from
oracle/oracle_py.py (unitigs + brute-force links) for the reference's fixture files and for
seeded random genomes.

Run from the repo root in the build container:  python tests/golden/make_postproc_golden.py
"""
import json, os, random, shutil, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as op

SCRIPTS = "/root/reference/scripts"
OUT = os.path.join(HERE, "postproc")


def unitig_fasta(text, k, amin, orient=None):
    """BCALM-format FASTA of the unitigs of `text` (README.md:62-72 of the reference)"""
    us, _ = op.unitigs(text, k, amin)
    seqs = [s for s, _ in us]
    if orient:
        seqs = [orient(s) for s in seqs]
    links = op.links(seqs, k)
    lines = []
    for i, (s, (_, kc)) in enumerate(zip(seqs, us)):
        n = len(s) - k + 1
        toks = ["L:%s:%d:%s" % (fs, v, ts) for (u, fs, v, ts) in sorted(links, key=lambda x: (x[0], x[1] != "+", x[2], x[3])) if u == i]
        lines.append(">%d LN:i:%d KC:i:%d km:f:%.1f %s \n%s\n" % (i, len(s), kc, kc / n, " ".join(toks), s) if toks else
                     ">%d LN:i:%d KC:i:%d km:f:%.1f \n%s\n" % (i, len(s), kc, kc / n, s))
    return "".join(lines)


def refs_fasta(seqs, width=None):
    out = []
    for i, s in enumerate(seqs):
        out.append(">ref%d\n" % i)
        if width:
            out.extend(s[j:j + width] + "\n" for j in range(0, len(s), width))
        else:
            out.append(s + "\n")
    return "".join(out)


def run_script(name, args, cwd):
    p = subprocess.run([sys.executable, os.path.join(SCRIPTS, name)] + args, cwd=cwd, capture_output=True, text=True)
    return p.returncode, p.stdout, p.stderr


def snapshot(cwd, before):
    files = {}
    for f in sorted(os.listdir(cwd)):
        if f not in before:
            files[f] = open(os.path.join(cwd, f)).read()
    return files


def make_case(name, refs_text, unitigs_text, k):
    d = os.path.join(OUT, name)
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "refs.fa"), "w").write(refs_text)
    open(os.path.join(d, "unitigs.fa"), "w").write(unitigs_text)
    exp = {"k": k, "commands": {}}
    runs = [("convertToGFA", "convertToGFA.py", ["unitigs.fa", "out.gfa", str(k)], 0),
            ("convertToGFA -s", "convertToGFA.py", ["-s", "unitigs.fa", "out.gfa", str(k)], 0),
            ("split_unitigs", "split_unitigs.py", ["refs.fa", "unitigs.fa", str(k)], 0),
            ("pufferize", "pufferize.py", ["refs.fa", "unitigs.fa", str(k)], 2),
            ("abundance_stats", "abundance_stats.py", ["unitigs.fa"], 0)]
    for label, script, args, _ in runs:
        with tempfile.TemporaryDirectory() as t:
            shutil.copy(os.path.join(d, "refs.fa"), t); shutil.copy(os.path.join(d, "unitigs.fa"), t)
            before = set(os.listdir(t))
            rc, out, err = run_script(script, args, t)
            if label == "pufferize" and rc == 0:
                # the lines after "done. result is in" name the script's own install path: not part of the vector
                out = out[:out.index("to update unitig links")]
            exp["commands"][label] = {"args": args, "rc": rc, "stdout": out, "files": snapshot(t, before)}
            if rc != 0:
                assert "Traceback" not in err, err               # an exit("message"), not a crash
                exp["commands"][label]["stderr"] = err
    json.dump(exp, open(os.path.join(d, "expected.json"), "w"), indent=1, sort_keys=True)
    return exp


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def main():
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    os.makedirs(OUT)
    summary = {}
    # 1. the reference's own example (example/pufferize/run.sh: k=9 on refs.fa)
    refs = open(os.path.join(HERE, "inputs", "pufferize_refs.fa")).read()
    text = op.read_fasta_text(os.path.join(HERE, "inputs", "pufferize_refs.fa"))
    summary["example_pufferize_k9"] = make_case("example_pufferize_k9", refs, unitig_fasta(text, 9, 1), 9)
    # 2. fixtures with self-loops, palindromes and circular unitigs (links of every kind for the GFA writer)
    for fx, k in (("circ_test2", 7), ("circ_test1", 7), ("minitip", 21), ("tiny_read", 13)):
        path = os.path.join(HERE, "inputs", fx + ".fa")
        text = op.read_fasta_text(path)
        summary[fx] = make_case("%s_k%d" % (fx, k), open(path).read(), unitig_fasta(text, k, 1), k)
    # 3. unitigs stored as the reverse complement of the references: the one orientation in which
    #    pufferize.py's path reconstruction succeeds (it only ever fills its end-k-mer map)
    rng = random.Random(20240931)
    a, b = rand_seq(rng, 60), rand_seq(rng, 45)
    summary["revcomp_refs"] = make_case("revcomp_refs_k11", refs_fasta([a, b]), unitig_fasta(a + "\n" + b + "\n", 11, 1, orient=op.revcomp), 11)
    # 4. random genomes, references = overlapping windows of the genome (so unitigs are cut inside), wrapped lines
    for seed, glen, k in ((1, 300, 11), (2, 500, 15), (3, 400, 9), (4, 800, 21), (5, 250, 7), (6, 600, 13)):
        rng = random.Random(seed)
        g = rand_seq(rng, glen)
        if seed % 2 == 0:                                   # plant a repeat and an inverted repeat
            g = g[:120] + g[30:80] + g[120:200] + op.revcomp(g[40:90]) + g[200:]
        nref = 2 + seed % 3
        windows = []
        for _ in range(nref):
            s = rng.randrange(0, len(g) - 3 * k)
            e = rng.randrange(s + 2 * k, min(len(g), s + 200) + 1)
            w = g[s:e]
            windows.append(op.revcomp(w) if rng.random() < 0.4 else w)
        reads = "\n".join([g] + windows) + "\n"
        summary["random_%d" % seed] = make_case("random_%d_k%d" % (seed, k), refs_fasta(windows, width=60 if seed % 2 else None), unitig_fasta(reads, k, 1), k)
    # 5. reads with coverage (km:f: values vary) for abundance_stats
    rng = random.Random(77)
    g = rand_seq(rng, 400)
    reads = []
    for _ in range(120):
        s = rng.randrange(0, 340)
        r = g[s:s + 60]
        reads.append(op.revcomp(r) if rng.random() < 0.5 else r)
    summary["coverage"] = make_case("coverage_k15", refs_fasta([g[:200], g[150:]]), unitig_fasta("\n".join(reads) + "\n", 15, 2), 15)
    # 6. malformed unitig file (same record twice): pufferize.py aborts, split_unitigs.py warns
    u = unitig_fasta(a + "\n", 11, 1)
    summary["duplicate_record"] = make_case("duplicate_record_k11", refs_fasta([a[5:40]]), u + u.replace(">0", ">1"), 11)
    # 7. ... and the same unitig again reverse-complemented: the other abort of pufferize.py
    seq = u.split("\n")[1]
    summary["duplicate_revcomp"] = make_case("duplicate_revcomp_k11", refs_fasta([a[5:40]]), u + ">1 LN:i:%d KC:i:1 km:f:1.0 \n%s\n" % (len(seq), op.revcomp(seq)), 11)
    for n, e in summary.items():
        print(n, {c: (v["rc"], sorted(v["files"])) for c, v in e["commands"].items()})


if __name__ == "__main__":
    main()
