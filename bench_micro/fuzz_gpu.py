"""dev tool: randomized parity fuzzing on the GPU against the CPU oracle for a time budget (seconds)"""
import sys, os, time, random, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib, bcalm_amd
orc = oracle_lib.load()
SIM = os.environ.get('FUZZ_SIM') == '1'            # same fuzzer through the CPU simulator build (small cases)
if SIM:
    import hostsim_lib
    lib = hostsim_lib.load()
else:
    lib = bcalm_amd.load()
budget = float(sys.argv[1]); seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else None
dump = os.environ.get('FUZZ_DUMP')
t_end = time.time() + budget
comp = str.maketrans("ACGT", "TGCA")
n_ok = 0; fails = []
it = 0
while time.time() < t_end:
    it += 1
    if only and it > max(only): break
    rng = random.Random(seed0 * 1000003 + it)
    k = rng.choice([5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31, 31, 31, 33, 35, 41, 47, 55, 55, 61, 63, 65, 71, 95, 97, 127])
    amin = rng.choice([1, 1, 2, 2, 3])
    alphabet = rng.choice(["ACGT"] * 5 + ["AT", "ACG", "AAC"])
    glen = rng.choice([300, 2000, 20000, 200000])
    g = "".join(rng.choice(alphabet) for _ in range(glen))
    if rng.random() < 0.5 and glen > 500:       # repeats, inverted repeats, a circular part
        a = rng.randrange(0, glen - 200); b = rng.randrange(0, glen - 200); L = rng.randrange(k, min(200, 3 * k + 20))
        g = g[:a] + g[b:b + L] + g[a:] + g[b:b + L][::-1].translate(comp)
    n_reads = rng.choice([1, 5, 50, 500, 5000, 20000])
    rl = rng.choice([k - 1, k, k + 1, 2 * k, 100, 150, 250, 1000])
    err = rng.choice([0, 0, 0.005, 0.02])
    reads = []
    for _ in range(n_reads):
        L = max(1, min(len(g), int(rl * rng.uniform(0.5, 1.2))))
        s = rng.randrange(0, len(g) - L + 1)
        r = g[s:s + L]
        if rng.random() < 0.5: r = r[::-1].translate(comp)
        if err:
            r = "".join((rng.choice("ACGT") if rng.random() < err else c) for c in r)
        if rng.random() < 0.05: p = rng.randrange(0, len(r)); r = r[:p] + rng.choice(["N", "n", "NNNN", "-"]) + r[p + 1:]
        if rng.random() < 0.05: r = r.lower()
        reads.append(r)
    if rng.random() < 0.2: reads.append(g + g[:k - 1])          # whole (circularised) genome as one read
    text = "\n".join(reads) + "\n"
    if len(text) > (60_000 if SIM else 6_000_000): continue
    if only and it not in only: continue
    if dump: open(os.path.join(dump, 'fuzz_%d_%d.txt' % (seed0, it)), 'w').write(text)
    lnp = [int(x) for x in os.environ["FUZZ_LOG_NP"].split(",")] if os.environ.get("FUZZ_LOG_NP") else [-1, -1, 0, 3, 8, 12]   # (deferred placement needs >= 1024 partitions)
    params = dict(k=k, amin=amin, log2_partitions=rng.choice(lnp), minimizer_size=rng.choice([0, 0, 0, min(k - 1, rng.randrange(2, 17))]))
    try:
        exp = orc.run(text, k, amin)
        gr = bcalm_amd.Graph(k, amin, lib=lib, log2_partitions=params["log2_partitions"], minimizer_size=params["minimizer_size"])
        gr.push_text(text); gr.run()
        got = oracle_lib.canonical_set(orc, gr.unitigs(), k); st = gr.stats(); gr.close()
        if got != exp["unitigs"] or st["n_distinct"] != exp["stats"]["distinct"] or st["n_solid"] != exp["stats"]["solid"]:
            fails.append((it, params, len(text), "MISMATCH", st["n_distinct"], exp["stats"]["distinct"], st["n_solid"], exp["stats"]["solid"], len(got), len(exp["unitigs"]), st["n_big_partitions"], st["log2_partitions"], st["minimizer_size"]))
        else:
            n_ok += 1
    except Exception as e:
        fails.append((it, params, len(text), repr(e)[:200]))
    if len(fails) >= 5: break
print(json.dumps({"iterations": it, "ok": n_ok, "fails": fails}))
