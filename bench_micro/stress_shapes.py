"""dev tool: run unusual shapes through the whole path on the GPU and check size-independent invariants
(no error, every solid k-mer in exactly one unitig position: sum over unitigs of (len-k+1) == n_solid, sum KC == sum of solid counts)"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bcalm_amd
lib = bcalm_amd.load(os.environ.get("CDBG_LIB"))

def check(name, g, k, want_solid_sum=True):
    g.run(); g.reset()                                   # first run pays the allocations
    t = time.time(); g.run(); dt = time.time() - t
    st = g.stats()
    ut = g.unitigs()
    nk = sum(len(s) - k + 1 for s, _ in ut)
    ok = nk == st["n_solid"] and len(ut) == st["n_unitigs"]
    print(json.dumps({"case": name, "ok": ok, "s": round(dt, 2), "ms": {x[3:]: round(st[x], 1) for x in ("ms_scan_emit", "ms_count", "ms_compact", "ms_glue")}, **{x: st[x] for x in ("n_distinct", "n_solid", "n_big_partitions", "log2_partitions", "minimizer_size")}}), flush=True)
    g.close()
    return ok

def rand_text(rng, glen, n_reads, L, alphabet="ACGT", err=0.01):
    gen = rng.integers(0, len(alphabet), glen).astype(np.uint8)
    lut = np.frombuffer(alphabet.encode(), dtype=np.uint8)
    starts = rng.integers(0, glen - L, n_reads)
    idx = starts[:, None] + np.arange(L)[None, :]
    reads = gen[idx]
    if err:
        m = rng.random(reads.shape) < err
        reads = np.where(m, (reads + 1 + rng.integers(0, len(alphabet) - 1, reads.shape)) % len(alphabet), reads).astype(np.uint8)
    out = np.full((n_reads, L + 1), ord("\n"), dtype=np.uint8)
    out[:, :L] = lut[reads]
    return out.tobytes()

ok = True
rng = np.random.default_rng(7)
for name, k, amin, n, L, cfg in (("k21_L100", 21, 2, 5_000_000, 100, 3), ("k31_amin1", 31, 1, 3_000_000, 150, 3), ("k63_L250", 63, 2, 2_000_000, 250, 4),
                                  ("k65_L300", 65, 2, 1_000_000, 300, 5), ("k31_L35", 31, 2, 4_000_000, 35, 3), ("k127_L1000", 127, 2, 1_000_000, 1000, 5),
                                  ("k15_L150", 15, 2, 2_000_000, 150, 3), ("k33_L150", 33, 3, 3_000_000, 150, 4)):
    g = bcalm_amd.Graph(k, amin, lib=lib); g.generate_reads(n, L, cfg); ok &= check(name, g, k)
# low-complexity: two-letter genome (minimizer collisions, palindromes, huge buckets) and a tandem-repeat genome
g = bcalm_amd.Graph(31, 2, lib=lib); g.push_text(rand_text(rng, 200_000, 400_000, 150, "AT")); ok &= check("AT_genome", g, 31)
unit = "".join("ACGT"[i] for i in rng.integers(0, 4, 5000))
rep = (unit * 40)
g = bcalm_amd.Graph(31, 2, lib=lib)
arr = np.frombuffer(rep.encode(), dtype=np.uint8)
starts = rng.integers(0, len(arr) - 150, 300_000)
out = np.full((300_000, 151), ord("\n"), dtype=np.uint8); out[:, :150] = arr[starts[:, None] + np.arange(150)[None, :]]
g.push_text(out.tobytes()); ok &= check("tandem_repeat", g, 31)
g = bcalm_amd.Graph(31, 1, lib=lib); g.push_text(("A" * 5000 + "\n") * 2000 + ("ACGT" * 2000 + "\n") * 500); ok &= check("homopolymers", g, 31)
print("ALL OK" if ok else "FAILURES")
