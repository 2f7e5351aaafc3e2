"""dev sweep over minimizer size / partition count: sweep.py READS [K] [m:lnp,m:lnp,...]  (the set digest must not depend on either)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bcalm_amd
lib = bcalm_amd.load(os.environ.get("CDBG_LIB"))
n = int(sys.argv[1])
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
cfg, L = (3, 150) if k <= 31 else (4, 150) if k <= 63 else (5, 1000)
combos = [(16, 22), (15, 22), (14, 22), (13, 22), (12, 22), (13, 21), (12, 21)]
if len(sys.argv) > 3:
    combos = [tuple(int(x) for x in c.split(":")) for c in sys.argv[3].split(",")]
for m, lnp in combos:
    g = bcalm_amd.Graph(k, 2, lib=lib, minimizer_size=m, log2_partitions=lnp)
    g.generate_reads(n, L, cfg)
    best = None
    for rep in range(3):
        g.run(); st = g.stats(); dg = g.digest(); g.reset()
        if best is None or st["ms_total"] < best["ms_total"]: best = st
    g.close()
    print(json.dumps({"m": m, "log_np": lnp, "records": best["n_records"], "members": best["n_member_kmers"], "big": best["n_big_partitions"], "multipass": best["n_multipass_partitions"],
                      "digest": "%016x" % dg["set_digest"], **{x: round(best[x], 1) for x in ("ms_scan_hist", "ms_scan_emit", "ms_count", "ms_compact", "ms_glue", "ms_total")}}), flush=True)
