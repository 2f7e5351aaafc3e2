"""ad-hoc stage timing on synthetic reads (dev tool, not the bench contract): quick_timing.py READS[,READS..] [K] [REPS]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bcalm_amd
lib = bcalm_amd.load(os.environ.get("CDBG_LIB"))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg, L = (3, 150) if k <= 31 else (4, 150) if k <= 63 else (5, 1000)
for n_reads in [int(x) for x in sys.argv[1].split(",")]:
    g = bcalm_amd.Graph(k, 2, lib=lib, log2_partitions=int(os.environ.get('CDBG_LOG_NP', -1)), minimizer_size=int(os.environ.get('CDBG_M', 0)))
    g.generate_reads(n_reads, L, cfg)
    for rep in range(reps):
        t1 = time.time(); g.run(); t2 = time.time()
        st = g.stats(); g.reset()
        keys = ("n_distinct", "n_big_partitions", "n_multipass_partitions", "minimizer_size", "log2_partitions", "ms_scan_hist", "ms_scan_emit", "ms_count", "ms_place", "count_slices", "ms_compact", "ms_glue")
        print(json.dumps({"n_reads": n_reads, "k": k, "run_wall_ms": round((t2 - t1) * 1e3, 1), "Gkmers_per_s": round(st["n_distinct"] / (t2 - t1) / 1e9, 3), **{x: (round(st[x], 2) if isinstance(st[x], float) else st[x]) for x in keys}}), flush=True)
    g.close()
