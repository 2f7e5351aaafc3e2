"""dev sweep over minimizer size / partition count at config-3 size"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bcalm_amd
lib = bcalm_amd.load()
n = int(sys.argv[1])
for m, lnp in [(16, 22), (14, 22), (13, 22), (12, 22), (14, 21), (13, 21), (14, 23)]:
    g = bcalm_amd.Graph(31, 2, lib=lib, minimizer_size=m, log2_partitions=lnp)
    g.generate_reads(n, 150, 3)
    best = None
    for rep in range(3):
        g.run(); st = g.stats(); g.reset()
        if best is None or st["ms_total"] < best["ms_total"]: best = st
    g.close()
    print(json.dumps({"m": m, "log_np": lnp, "records": best["n_records"], "big": best["n_big_partitions"], **{k: round(best[k], 1) for k in ("ms_scan_hist", "ms_scan_emit", "ms_count", "ms_compact", "ms_glue", "ms_total")}}), flush=True)
