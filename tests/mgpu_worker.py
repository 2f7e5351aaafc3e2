"""One rank of a REAL multi-GPU job (tests/test_gpu_multi.py launches N of these through torch.distributed.run, one per
physical GPU): the library's own RCCL transport (grouped ncclSend / ncclRecv between different devices over xGMI), reads
sharded (X1) or replicated (X0); rank 0 gathers every rank's unitigs and compares their union with the oracle's set --
SURVEY.md section 8(e): "canonical unitig set from n GPUs == set from 1 GPU == oracle"."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, required=True)
    ap.add_argument("--amin", type=int, default=2)
    ap.add_argument("--reads", type=int, required=True)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--cfg", type=int, default=3)
    ap.add_argument("--replicated", type=int, default=0)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    import bcalm_amd
    from bcalm_amd import dist as cdist
    import oracle_lib
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    lib = bcalm_amd.load()
    orc = oracle_lib.load()
    text = orc.synth_reads(a.reads, a.read_len, a.cfg)           # (every rank writes the same bytes: counter-based generator)
    g = bcalm_amd.Graph(a.k, a.amin, lib=lib, device_id=local, world_size=world, rank=rank, reads_replicated=bool(a.replicated))
    cdist.init_rccl(g, dist, device=torch.device("cuda", local))
    if a.replicated:
        g.push_text(text)
    else:
        reads = [x for x in text.decode().split("\n") if x]
        g.push_text(("\n".join(reads[rank::world]) + "\n").encode())
    res = None
    for step in range(a.steps):
        if step:
            g.reset()
        g.run()
        st = g.stats()
        v = g.verify()
        mine = (g.unitigs(), st["n_distinct"], st["n_solid"], st["n_occurrences"], g.comm_bytes(), st["n_glue_rounds"], v["unitig_kmers"], v["solid_kmers"])
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        if rank == 0:
            exp = orc.run(text, a.k, a.amin)
            union = sorted((orc.canonical_unitig(s, a.k), int(kc)) for part in gathered for s, kc in part[0])
            M = (1 << 64) - 1
            vu = tuple(sum(p[6][i] for p in gathered) & M for i in range(3)); vs = tuple(sum(p[7][i] for p in gathered) & M for i in range(3))
            res = {"world": world, "step": step, "set_equal": union == exp["unitigs"], "unitigs": len(union), "expected": len(exp["unitigs"]),
                   "distinct_equal": sum(p[1] for p in gathered) == exp["stats"]["distinct"], "solid_equal": sum(p[2] for p in gathered) == exp["stats"]["solid"],
                   "per_rank_unitigs": [len(p[0]) for p in gathered], "comm_bytes": [p[4] for p in gathered], "rounds": gathered[0][5],
                   "verify_equal": vu == vs, "devices": torch.cuda.device_count()}
            if not (res["set_equal"] and res["distinct_equal"] and res["solid_equal"] and res["verify_equal"]):
                break
    g.close()
    if rank == 0:
        with open(a.out, "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
