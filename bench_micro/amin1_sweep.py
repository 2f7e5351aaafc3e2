"""dev tool: abundance-min 1 (every k-mer solid) vs partition count"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bcalm_amd
lib = bcalm_amd.load()
for lnp in (-1, 18, 19, 20):
    g = bcalm_amd.Graph(31, 1, lib=lib, log2_partitions=lnp); g.generate_reads(3000000, 150, 3)
    for rep in range(2):
        g.run(); st = g.stats()
        if rep == 0: g.reset()
    print(json.dumps({"log_np": st["log2_partitions"], "m": st["minimizer_size"], **{x: (round(st[x], 1) if isinstance(st[x], float) else st[x]) for x in ("n_solid", "n_big_partitions", "ms_scan_emit", "ms_count", "ms_compact", "ms_glue", "ms_total")}}), flush=True)
    g.close()
