"""dev tool: stage timing of the HOSTILE generator (cfg | 0x100) or any cfg: hostile_timing.py READS K REPS [GENCFG]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bcalm_amd
lib = bcalm_amd.load(os.environ.get("CDBG_LIB"))
n_reads = int(sys.argv[1]); k = int(sys.argv[2]); reps = int(sys.argv[3])
cfg, L = (3, 150) if k <= 31 else (4, 150) if k <= 63 else (5, 1000)
gen = int(sys.argv[4], 0) if len(sys.argv) > 4 else (cfg | 0x100)
g = bcalm_amd.Graph(k, 2, lib=lib, log2_partitions=int(os.environ.get("CDBG_LOG_NP", -1)), minimizer_size=int(os.environ.get("CDBG_M", 0)))
g.generate_reads(n_reads, L, gen)
for rep in range(reps):
    t1 = time.time(); g.run(); t2 = time.time()
    st = g.stats(); dg = g.digest(); g.reset()
    keys = ("n_distinct", "n_solid", "n_records", "n_big_partitions", "n_multipass_partitions", "n_split_buckets", "ms_scan_hist", "ms_scan_emit", "ms_count", "ms_compact", "ms_glue", "ms_total")
    print(json.dumps({"gen": hex(gen), "run_wall_ms": round((t2 - t1) * 1e3, 1), "set_digest": "%016x" % dg["set_digest"], **{x: (round(st[x], 2) if isinstance(st[x], float) else st[x]) for x in keys}}), flush=True)
g.close()
