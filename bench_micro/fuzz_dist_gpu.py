"""dev tool: randomized fuzzing of the MULTI-RANK data path on one GPU (N contexts as N host threads over the in-process
loop-back transport of tests/loopback.py) against the CPU oracle, for a time budget: fuzz_dist_gpu.py SECONDS [SEED]"""
import sys, os, time, random, json, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib, bcalm_amd
from loopback import hip_loopback
orc = oracle_lib.load()
HOSTSIM = bool(os.environ.get("FUZZ_HOSTSIM"))          # the same cases through the SIMT simulator build (CPU): FUZZ_HOSTSIM=1
if HOSTSIM:
    import ctypes, hostsim_lib
    from loopback import Loopback
    lib = hostsim_lib.load()
    def hip_loopback(world):
        mv = lambda dst, src, n: ctypes.memmove(dst, src, n)
        hub = Loopback(world, mv); hub.memcpy_d2h = mv; hub.memcpy_h2d = mv
        return hub
else:
    lib = bcalm_amd.load()
budget = float(sys.argv[1]); seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else None      # fuzz_dist_gpu.py SECONDS SEED IT[,IT..]: just these iterations, errors in full
LONG = bool(only) or bool(os.environ.get("FUZZ_LONGERR"))
t_end = time.time() + budget
comp = str.maketrans("ACGT", "TGCA")
n_ok = 0; fails = []; it = 0
while time.time() < t_end and len(fails) < (1 if os.environ.get("FUZZ_LONGERR") else 5):
    it += 1
    if only and it > max(only): break
    if only and it not in only: continue
    rng = random.Random(seed0 * 7000003 + it)
    k = rng.choice([7, 15, 21, 31, 31, 33, 55, 63, 65, 97, 127])
    amin = rng.choice([1, 2, 2, 3])
    world = rng.choice([2, 2, 4, 8])
    glen = rng.choice([500, 5000, 50000, 300000])
    g = "".join(rng.choice("ACGT") for _ in range(glen))
    if rng.random() < 0.5 and glen > 600:
        a = rng.randrange(0, glen - 300); b = rng.randrange(0, glen - 300); L = rng.randrange(k, min(250, 3 * k + 20))
        g = g[:a] + g[b:b + L] + g[a:] + g[b:b + L][::-1].translate(comp)
    reads = []
    err = rng.choice([0, 0.005, 0.02])
    for _ in range(rng.choice([8, 200, 3000, 20000])):
        L = max(1, min(len(g), int(rng.choice([k, 2 * k, 150, 400]) * rng.uniform(0.6, 1.3))))
        s = rng.randrange(0, len(g) - L + 1); r = g[s:s + L]
        if rng.random() < 0.5: r = r[::-1].translate(comp)
        if err: r = "".join((rng.choice("ACGT") if rng.random() < err else c) for c in r)
        reads.append(r)
    if rng.random() < 0.2: reads.append(g + g[:k - 1])
    text = "\n".join(reads) + "\n"
    if len(text) > 4_000_000: continue
    kw = dict(log2_partitions=rng.choice([-1, -1, 3, 8, 12]), emit_replicated=rng.random() < 0.25, all_abundance_counts=rng.random() < 0.25)
    if kw["log2_partitions"] >= 0 and (1 << kw["log2_partitions"]) < world: kw["log2_partitions"] = 3
    try:
        exp = orc.run(text, k, amin, want_solid=True)
        hub = hip_loopback(world); out = [None] * world
        def rank_main(r):
            try:
                gr = bcalm_amd.Graph(k, amin, lib=lib, world_size=world, rank=r, **kw)
                ep = hub.endpoint(r); ep.attach(gr)
                gr.push_text(("\n".join(reads[r::world]) + "\n").encode())
                gr.run()
                out[r] = (gr.unitigs(), gr.stats(), ep.error, gr.unitig_abundances() if kw["all_abundance_counts"] else None)
                gr.close()
            except Exception as e:               # noqa: BLE001
                out[r] = e; hub.barrier.abort()
        ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for t in ts: t.start()
        for t in ts: t.join(300)
        bad = [repr(o)[:2000 if LONG else 200] for o in out if isinstance(o, Exception) or o is None or o[2] is not None]
        if bad: raise RuntimeError("; ".join(bad))
        if kw["emit_replicated"]:
            ok = all(oracle_lib.canonical_set(orc, out[r][0], k) == exp["unitigs"] for r in range(world))
        else:
            ok = sorted((orc.canonical_unitig(s, k), int(kc)) for r in range(world) for s, kc in out[r][0]) == exp["unitigs"]
        ok = ok and sum(out[r][1]["n_distinct"] for r in range(world)) == exp["stats"]["distinct"] and sum(out[r][1]["n_solid"] for r in range(world)) == exp["stats"]["solid"]
        if kw["all_abundance_counts"]:
            solid = dict(exp["solid"])
            for r in range(world):
                for (s, kc), a in zip(out[r][0], out[r][3]):
                    ok = ok and a == [solid[min(s[i:i + k], s[i:i + k].translate(comp)[::-1])] for i in range(len(s) - k + 1)] and sum(a) == kc
        if ok: n_ok += 1
        else: fails.append((it, k, amin, world, kw, len(text), "MISMATCH"))
    except Exception as e:                       # noqa: BLE001
        fails.append((it, k, amin, world, kw, len(text), repr(e)[:20000 if LONG else 300]))
print(json.dumps({"iterations": it, "ok": n_ok, "fails": fails}))
