"""dev tool: MEASURED per-rank scan / count / compact times of an N-rank job on one GPU, for the multi-GPU projection
(no multi-GPU node is reachable).  One context is created as rank 0 of N with reads_replicated = 1 (every
rank scans ALL reads for its own partitions p mod N); its transport pretends that the other ranks reported the same numbers
(all_gather_u64 echoes), which is all cdbg_count / cdbg_compact ask of it in this mode.  The glue stage needs real peers and is
not run here: its per-rank share is the 1-rank figure of forcedist_timing.py divided by N.
   solo_rank_timing.py READS [K]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bcalm_amd
from bcalm_amd import dist as cdist
lib = bcalm_amd.load(os.environ.get("CDBG_LIB"))
n = int(sys.argv[1]); k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
cfg, L = (3, 150) if k <= 31 else (4, 150) if k <= 63 else (5, 1000)
for world in (1, 2, 4, 8):
    def ag(user, send, recv, cnt, world=world):
        for r in range(world):
            for i in range(cnt):
                recv[r * cnt + i] = send[i]
        return 0
    cbs = (cdist.FN_AG64(ag), cdist.FN_A2AV(lambda *a: -1), cdist.FN_AGV(lambda *a: -1), cdist.FN_ARMAX(lambda *a: -1))
    tr = cdist.Transport(None, *cbs)
    g = bcalm_amd.Graph(k, 2, lib=lib, world_size=world, rank=0, reads_replicated=True)
    if world > 1:
        g._ck(lib.cdbg_set_transport(g._h, C.byref(tr)))
    g.generate_reads(n, L, cfg)
    best = None
    for rep in range(3):
        g.count(); g.compact(); st = g.stats(); g.reset()
        if best is None or st["ms_scan_emit"] + st["ms_count"] + st["ms_compact"] < best["ms_scan_emit"] + best["ms_count"] + best["ms_compact"]:
            best = st
    g.close()
    print(json.dumps({"ranks": world, "k": k, "reads_scanned": n, "log2_partitions": best["log2_partitions"], "own_records": best["n_records"], "own_solid": best["n_solid"],
                      **{x: round(best[x], 2) for x in ("ms_scan_hist", "ms_scan_emit", "ms_count", "ms_compact")}}), flush=True)
