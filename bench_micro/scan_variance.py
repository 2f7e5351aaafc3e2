import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bcalm_amd
for i in range(5):
    g = bcalm_amd.Graph(31, 2)
    g.generate_reads(100000000, 150, 3)
    ts = []
    for rep in range(3):
        g.run(); st = g.stats(); g.reset(); ts.append(round(st["ms_scan_emit"], 2))
    print("context", i, "scan", ts, flush=True)
    g.close()
