"""dev tool: fuzz -all-abundance-counts vectors and unitig links on the GPU (tests' own checkers) for a time budget"""
import sys, os, time, random, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib, bcalm_amd
import test_abundance, test_links
orc = oracle_lib.load(); lib = bcalm_amd.load()
budget = float(sys.argv[1]); seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
comp = str.maketrans("ACGT", "TGCA")
n_ok = 0; fails = []; it = 0
while time.time() < t_end and len(fails) < 5:
    it += 1
    rng = random.Random(seed0 * 9000011 + it)
    k = rng.choice([5, 7, 9, 13, 21, 31, 31, 33, 55, 63, 65, 127])
    amin = rng.choice([1, 1, 2, 3])
    alphabet = rng.choice(["ACGT"] * 4 + ["AT", "ACG"])
    glen = rng.choice([200, 1500, 8000])
    g = "".join(rng.choice(alphabet) for _ in range(glen))
    if rng.random() < 0.5 and glen > 600:
        a = rng.randrange(0, glen - 300); b = rng.randrange(0, glen - 300); L = rng.randrange(k, min(250, 3 * k + 20))
        g = g[:a] + g[b:b + L] + g[a:] + g[b:b + L][::-1].translate(comp)
    reads = []
    for _ in range(rng.choice([2, 30, 400])):
        L = max(1, min(len(g), int(rng.choice([k, 2 * k, 150]) * rng.uniform(0.6, 1.3))))
        s = rng.randrange(0, len(g) - L + 1); r = g[s:s + L]
        if rng.random() < 0.5: r = r[::-1].translate(comp)
        reads.append(r)
    if rng.random() < 0.3: reads.append(g + g[:k - 1])
    text = "\n".join(reads) + "\n"
    kw = dict(log2_partitions=rng.choice([-1, 0, 4, 9]), minimizer_size=rng.choice([0, 0, min(k - 1, rng.randrange(2, 17))]))
    try:
        test_abundance._check(lib, orc, text, k, amin, **kw)
        test_links._check(lib, text, k, amin, **kw)
        n_ok += 1
    except Exception as e:
        fails.append((it, k, amin, kw, len(text), repr(e)[:300]))
print(json.dumps({"iterations": it, "ok": n_ok, "fails": fails}))
