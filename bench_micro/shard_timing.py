"""dev tool: per-stage times of ONE rank of a sharded run (world_size W, rank 0), on one GPU.
The scan is replicated and keeps only this rank's partitions; count/compact see 1/W of the partitions."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bcalm_amd
lib = bcalm_amd.load(os.environ.get("CDBG_LIB"))
n_reads = int(sys.argv[1]); k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
for world in (1, 2, 4, 8):
    g = bcalm_amd.Graph(k, 2, lib=lib, world_size=world, rank=0)
    g.generate_reads(n_reads, 150, 3)
    for rep in range(2):
        g.count(); g.compact()
        st = g.stats()
        if rep == 0: g.reset()
    print(json.dumps({"world": world, **{x: round(st[x], 2) for x in ("ms_scan_hist", "ms_scan_emit", "ms_count", "ms_compact")}, "sizes": g.exchange_sizes() if world > 1 else None}), flush=True)
    g.close()
