"""ad-hoc stage timing on synthetic reads (dev tool, not the bench contract)"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bcalm_amd
lib = bcalm_amd.load(os.environ.get("CDBG_LIB"))
for n_reads, k, L, cfg in [(int(x), int(sys.argv[2]) if len(sys.argv) > 2 else 31, 150, 3) for x in sys.argv[1].split(",")]:
    for rep in range(2):
        g = bcalm_amd.Graph(k, 2, lib=lib)
        t0 = time.time(); g.generate_reads(n_reads, L, cfg); t1 = time.time()
        g.run(); t2 = time.time()
        st = g.stats(); g.close()
        keys = ("n_distinct", "n_big_partitions", "minimizer_size", "log2_partitions", "ms_scan_hist", "ms_scan_emit", "ms_count", "ms_compact", "ms_glue", "ms_total")
        print(json.dumps({"n_reads": n_reads, "k": k, "gen_s": round(t1 - t0, 3), "run_wall_s": round(t2 - t1, 3), "Gkmers_per_s": round(st["n_distinct"] / st["ms_total"] / 1e6, 3), **{x: (round(st[x], 2) if isinstance(st[x], float) else st[x]) for x in keys}}), flush=True)
