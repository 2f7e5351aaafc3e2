import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bcalm_amd
lib = bcalm_amd.load(os.environ.get("CDBG_LIB"))
n = int(sys.argv[1]); m = int(sys.argv[2])
g = bcalm_amd.Graph(31, 2, lib=lib, minimizer_size=m)
g.generate_reads(n, 150, 3)
for rep in range(2):
    g.run(); st = g.stats(); g.reset()
    print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in st.items() if k in ("n_records", "n_distinct", "n_unitigs", "n_big_partitions", "ms_scan_emit", "ms_count", "ms_total", "log2_partitions")}), flush=True)
g.close()
