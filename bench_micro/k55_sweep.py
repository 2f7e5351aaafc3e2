"""dev tool: stage times vs partition count at the config-4 shape (k=55)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bcalm_amd
lib = bcalm_amd.load(os.environ.get("CDBG_LIB"))
n_reads = int(sys.argv[1]); k = int(sys.argv[2])
for lnp in [int(x) for x in sys.argv[3].split(",")]:
    g = bcalm_amd.Graph(k, 2, lib=lib, log2_partitions=lnp)
    g.generate_reads(n_reads, int(sys.argv[4]) if len(sys.argv) > 4 else 150, 4 if k == 55 else 5)
    for rep in range(2):
        g.run(); st = g.stats()
        if rep == 0: g.reset()
    print(json.dumps({"k": k, "log_np": lnp, **{x: (round(st[x], 1) if isinstance(st[x], float) else st[x]) for x in ("n_records", "n_big_partitions", "n_pieces", "ms_scan_emit", "ms_count", "ms_compact", "ms_glue", "ms_total")}}), flush=True)
    g.close()
