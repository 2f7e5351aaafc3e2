"""dev tool: the multi-GPU code path (sharded reads or replicated, sharded glue) through a ONE-rank RCCL communicator:
forcedist_timing.py READS [K] [REPS] [replicated]   -- stage times; run under rocprofv3 --kernel-trace --stats for the kernels"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CDBG_FORCE_MULTI"] = "1"
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29733", RANK="0", WORLD_SIZE="1")
import torch.distributed as dist
import bcalm_amd
from bcalm_amd import dist as cdist
n = int(sys.argv[1]); k = int(sys.argv[2]) if len(sys.argv) > 2 else 31; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
repl = len(sys.argv) > 4 and sys.argv[4] == "replicated"
cfg, L = (3, 150) if k <= 31 else (4, 150) if k <= 63 else (5, 1000)
dist.init_process_group("gloo", rank=0, world_size=1)
g = bcalm_amd.Graph(k, 2, lib=bcalm_amd.load(os.environ.get("CDBG_LIB")), world_size=1, rank=0, reads_replicated=repl)
cdist.init_rccl(g, dist)
g.generate_reads(n, L, cfg)
for rep in range(reps):
    t1 = time.time(); g.run(); t2 = time.time()
    st = g.stats(); dg = g.digest(); g.reset()
    print(json.dumps({"n_reads": n, "k": k, "wall_ms": round((t2 - t1) * 1e3, 1), "digest": "%016x" % dg["set_digest"], "rounds": st["n_glue_rounds"],
                      **{x: round(st[x], 1) for x in ("ms_scan_hist", "ms_scan_emit", "ms_count", "ms_compact", "ms_glue", "ms_exchange", "ms_total")}}), flush=True)
g.close(); dist.destroy_process_group()
