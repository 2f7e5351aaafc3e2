"""dev tool: randomized parity fuzzing of the multi-rank data path, ranks emulated one after the other on one GPU
(packed exchange, sharded junction join, MAX-combined link arrays), against the CPU oracle, for a time budget"""
import sys, os, time, random, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_lib, bcalm_amd
orc = oracle_lib.load(); lib = bcalm_amd.load()
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
budget = float(sys.argv[1]); seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
comp = str.maketrans("ACGT", "TGCA")
n_ok = 0; fails = []; it = 0
while time.time() < t_end and len(fails) < 5:
    it += 1
    rng = random.Random(seed0 * 7000003 + it)
    k = rng.choice([7, 11, 15, 21, 25, 31, 31, 33, 47, 55, 63, 65, 97, 127])
    amin = rng.choice([1, 2, 2, 3])
    world = rng.choice([2, 2, 4, 8])
    glen = rng.choice([500, 5000, 50000, 300000])
    g = "".join(rng.choice("ACGT") for _ in range(glen))
    if rng.random() < 0.5 and glen > 600:
        a = rng.randrange(0, glen - 300); b = rng.randrange(0, glen - 300); L = rng.randrange(k, min(250, 3 * k + 20))
        g = g[:a] + g[b:b + L] + g[a:] + g[b:b + L][::-1].translate(comp)
    reads = []
    for _ in range(rng.choice([3, 100, 3000, 20000])):
        L = max(1, min(len(g), int(rng.choice([k, 2 * k, 150, 400]) * rng.uniform(0.6, 1.2))))
        s = rng.randrange(0, len(g) - L + 1); r = g[s:s + L]
        if rng.random() < 0.5: r = r[::-1].translate(comp)
        if rng.random() < 0.3: r = "".join((rng.choice("ACGT") if rng.random() < 0.01 else c) for c in r)
        reads.append(r)
    if rng.random() < 0.3: reads.append(g + g[:k - 1])
    text = "\n".join(reads) + "\n"
    if len(text) > 5_000_000: continue
    lnp = rng.choice([-1, -1, 3, 6, 10])
    if lnp >= 0 and (1 << lnp) < world: lnp = 3
    params = dict(k=k, amin=amin, world=world, log2_partitions=lnp)
    try:
        exp = orc.run(text, k, amin)
        gs = []
        for r in range(world):
            gr = bcalm_amd.Graph(k, amin, lib=lib, log2_partitions=lnp, world_size=world, rank=r)
            gr.push_text(text); gr.count(); gr.compact(); gs.append(gr)
        W = 1 if k <= 31 else 2 if k <= 63 else 4
        ps = [gr.exchange_sizes_packed() for gr in gs]
        bufs = []
        for r, gr in enumerate(gs):
            row = []
            for kind, nb in ((0, ps[r][0] * 4), (1, ps[r][0] * 8), (None, ps[r][3]), (4, ps[r][2] * 8 * W), (5, ps[r][2] * 4)):
                t = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
                (gr.exchange_export_packed(t.data_ptr(), t.numel()) if kind is None else gr.exchange_export(kind, t.data_ptr(), t.numel()))
                row.append(t)
            bufs.append(row)
        tot = [sum(ps[r][j] for r in range(world)) for j in range(3)]
        links = []
        for gr in gs:
            gr.exchange_begin(*tot)
            for r in range(world):
                gr.exchange_add_packed(ps[r][0], ps[r][1], ps[r][3], ps[r][2], [t.data_ptr() for t in bufs[r]])
            gr.exchange_end()
            n = gr.glue_join()
            t = torch.full((max(n, 1),), -1, dtype=torch.int32, device=dev)
            gr.glue_links_export(t.data_ptr(), n * 4)
            links.append(t)
        merged = links[0]
        for t in links[1:]: merged = torch.maximum(merged, t)
        torch.cuda.synchronize()
        ok = True
        for i, gr in enumerate(gs):
            gr.glue_links_import(merged.data_ptr(), merged.numel() * 4 if tot[0] else 0)
            gr.glue()
            if i in (0, world - 1):
                ok = ok and oracle_lib.canonical_set(orc, gr.unitigs(), k) == exp["unitigs"]
            gr.close()
        if ok: n_ok += 1
        else: fails.append((it, params, len(text), "MISMATCH"))
    except Exception as e:
        fails.append((it, params, len(text), repr(e)[:200]))
        for gr in gs:
            try: gr.close()
            except Exception: pass
print(json.dumps({"iterations": it, "ok": n_ok, "fails": fails}))
