"""dev tool: stage times vs minimizer size / partition count at config 3"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bcalm_amd
lib = bcalm_amd.load(os.environ.get("CDBG_LIB"))
n_reads = int(sys.argv[1])
g = None
for m, lnp in [(16, 22), (15, 22), (14, 22), (16, 21), (15, 21)]:
    g = bcalm_amd.Graph(31, 2, lib=lib, minimizer_size=m, log2_partitions=lnp)
    g.generate_reads(n_reads, 150, 3)
    for rep in range(2):
        g.run(); st = g.stats()
        if rep == 0: g.reset()
    print(json.dumps({"m": m, "log_np": lnp, **{x: (round(st[x], 1) if isinstance(st[x], float) else st[x]) for x in ("n_records", "n_big_partitions", "n_pieces", "ms_scan_emit", "ms_count", "ms_compact", "ms_glue", "ms_total")}}), flush=True)
    g.close()
