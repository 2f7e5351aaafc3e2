"""dev tool: the real `bcalm` binary on random gzip inputs cut into small chunks (parallel inflate, bcalm_amd/host/pgz.h) against the CPU oracle: fuzz_cli_gz.py SECONDS SEED"""
import os, random, subprocess, sys, tempfile, time, zlib, gzip
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
orc = oracle_lib.load()
BCALM = os.path.join(ROOT, "bcalm_amd", "_build", "bcalm")
budget = float(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget; n = 0; par = 0
comp = str.maketrans("ACGT", "TGCA")
with tempfile.TemporaryDirectory() as td:
    while time.time() < t_end:
        seed += 1; rng = random.Random(seed)
        k = rng.choice([9, 15, 21, 31, 31, 33, 55, 63, 65, 127]); amin = rng.choice([1, 1, 2, 3])
        g = "".join(rng.choice("ACGT") for _ in range(rng.choice([500, 5000, 50000])))
        reads = []
        for _ in range(rng.choice([50, 500, 5000, 20000])):
            L = max(1, min(len(g), int(rng.choice([k, 2 * k, 100, 150, 250, 1000]) * rng.uniform(0.5, 1.2)))); s = rng.randrange(0, len(g) - L + 1); r = g[s:s + L]
            if rng.random() < 0.5: r = r[::-1].translate(comp)
            if rng.random() < 0.02: p = rng.randrange(len(r)); r = r[:p] + "N" + r[p + 1:]
            reads.append(r)
        fmt = rng.choice(["fq", "fq", "fa", "fa_wrapped", "fq_wrapped"])
        if fmt == "fq": t = "".join("@r%d x\n%s\n+\n%s\n" % (i, r, "".join(chr(33 + rng.randrange(2, 41)) for _ in r)) for i, r in enumerate(reads))
        elif fmt == "fq_wrapped": t = "".join("@r%d\n%s\n%s\n+\n%s\n%s\n" % (i, r[:len(r) // 2], r[len(r) // 2:], "I" * (len(r) // 2), "I" * (len(r) - len(r) // 2)) for i, r in enumerate(reads))
        elif fmt == "fa": t = "".join(">r%d\n%s\n" % (i, r) for i, r in enumerate(reads))
        else: w = rng.choice([50, 60, 70]); t = "".join(">r%d\n%s\n" % (i, "\n".join(r[j:j + w] for j in range(0, len(r), w))) for i, r in enumerate(reads))
        t = t.encode()
        c = zlib.compressobj(rng.choice([1, 6, 9]), zlib.DEFLATED, 31, rng.choice([3, 4, 8]), rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED]))
        blob = c.compress(t) + c.flush()
        if rng.random() < 0.3: h = len(t) // 2; blob = gzip.compress(t[:h], 6) + gzip.compress(t[h:], 6)
        open(os.path.join(td, "in.gz"), "wb").write(blob)
        env = dict(os.environ, BCALM_GZ_CHUNK=str(rng.choice([1024, 3000, 20000, 100000])), BCALM_GZ_VERBOSE="1")
        r = subprocess.run([BCALM, "-in", "in.gz", "-kmer-size", str(k), "-abundance-min", str(amin), "-nb-cores", str(rng.choice([2, 3, 5, 8])), "-out", "o"], cwd=td, capture_output=True, text=True, env=env, timeout=300)
        n += 1; par += "inflated by" in r.stderr
        exp = orc.run("\n".join(reads) + "\n", k, amin)
        ok = r.returncode == 0
        if ok:
            lines = open(os.path.join(td, "o.unitigs.fa")).read().split("\n")
            recs = [(lines[i + 1], int(lines[i].split("KC:i:")[1].split()[0])) for i in range(0, len(lines) - 1, 2)]
            ok = oracle_lib.canonical_set(orc, recs, k) == exp["unitigs"]
        if not ok:
            print("FAIL seed", seed, "k", k, "amin", amin, fmt, env["BCALM_GZ_CHUNK"], r.returncode, (r.stdout + r.stderr)[-400:]); open(os.path.join(ROOT, "gpurun_out", "fuzz_cli_gz_fail_%d.gz" % seed), "wb").write(blob); sys.exit(1)
print("cases", n, "inflated in parallel", par, "ALL OK")
