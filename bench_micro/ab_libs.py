"""dev tool: A/B of libcdbg variants built into bench_micro/variants/ (hipcc ... -D<knob> -o bench_micro/variants/libcdbg_<name>.so):
ab_libs.py [name[:log2_partitions] ...]  -> one line per variant: best-of-4 stage times at config 3 and the set digest (must agree)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import bcalm_amd
    name, lnp, cfg = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    k, L, n = {3: (31, 150, 100_000_000), 4: (55, 150, 125_000_000), 5: (127, 1000, 6_250_000)}[cfg & 0xFF]
    lib = bcalm_amd.load(os.path.join(ROOT, "bench_micro", "variants", "libcdbg_%s.so" % name))
    g = bcalm_amd.Graph(k, 2, lib=lib, log2_partitions=lnp)
    g.generate_reads(n, L, cfg)
    best = None
    for rep in range(4):
        g.run(); st = g.stats(); dg = g.digest(); g.reset()
        if rep and (best is None or st["ms_total"] < best["ms_total"]): best = st
    g.close()
    print(json.dumps({"lib": name, "log_np": best["log2_partitions"], "multipass": best["n_multipass_partitions"], "digest": "%016x" % dg["set_digest"],
                      **{x: round(best[x], 1) for x in ("ms_scan_emit", "ms_count", "ms_compact", "ms_glue", "ms_total")}}), flush=True)
else:
    cfg = int(os.environ.get("AB_CFG", "3"))
    for spec in sys.argv[1:]:
        name, _, lnp = spec.partition(":")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name, lnp or "-1", str(cfg)], stderr=subprocess.DEVNULL)
