"""Multi-GPU glue exchange: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box; "gloo" in the CPU tests with the simulator library).

Minimizer partitions are sharded over the ranks (cdbg_params.world_size / rank): every rank scans
the same resident reads and counts / compacts only the partitions it owns.  What has to cross
ranks are the glue records: pieces (length, abundance, bases packed 4 per byte) and the glue log
(junction key, piece-end id / CONFIRM).  They are gathered with all_gather_into_tensor and merged in rank order
by libcdbg (cdbg_exchange_*).  The junction hash-join over the union is sharded by key hash
(cdbg_glue_join) and its result, one partner id per piece end, is combined with a MAX all-reduce;
every rank then ranks the chains and emits, and holds the complete unitig set.  torch only moves
bytes here; all compute stays in the HIP library.
"""
from __future__ import annotations

import torch


def _gather_bytes(dist, device, world, graph, nbytes_per_rank, export):
    """all-gather one variable-length byte array (padded to the longest rank); -> (tensor, stride)"""
    pad = max(max(nbytes_per_rank), 16)
    pad = (pad + 15) // 16 * 16
    send = torch.empty(pad, dtype=torch.uint8, device=device)
    export(send.data_ptr(), pad)
    recv = torch.empty(world * pad, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(recv, send)
    return recv, pad


def exchange_glue(graph, dist, device, W: int, sharded_join: bool = True):
    """all-gather every rank's pieces + glue log and merge them into `graph` (stage: compacted).
    On the wire per rank: piece lengths (u32), piece abundance sums (u64), piece bases packed 4 per byte with the
    reservation gaps squeezed out (no base offsets: the receiver recomputes them), glue-log keys and tags."""
    world = dist.get_world_size()
    device = torch.device(device)
    mine = torch.tensor(graph.exchange_sizes_packed(), dtype=torch.int64, device=device)      # pieces, bases, glog, packed bytes
    sizes = torch.empty(world * 4, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, mine)
    sizes = sizes.view(world, 4).cpu().tolist()
    col = lambda j: [int(sizes[r][j]) for r in range(world)]
    parts = [
        _gather_bytes(dist, device, world, graph, [n * 4 for n in col(0)], lambda p, n: graph.exchange_export(0, p, n)),          # piece_n
        _gather_bytes(dist, device, world, graph, [n * 8 for n in col(0)], lambda p, n: graph.exchange_export(1, p, n)),          # piece_kc
        _gather_bytes(dist, device, world, graph, col(3), lambda p, n: graph.exchange_export_packed(p, n)),                        # packed bases
        _gather_bytes(dist, device, world, graph, [n * 8 * W for n in col(2)], lambda p, n: graph.exchange_export(4, p, n)),      # glog keys
        _gather_bytes(dist, device, world, graph, [n * 4 for n in col(2)], lambda p, n: graph.exchange_export(5, p, n)),          # glog tags
    ]
    if device.type == "cuda":
        torch.cuda.synchronize(device)                   # collectives run on torch's stream, libcdbg on its own
    totals = [sum(col(j)) for j in range(3)]
    graph.exchange_begin(*totals)
    for r in range(world):
        ptrs = [recv.data_ptr() + r * pad for recv, pad in parts]
        graph.exchange_add_packed(int(sizes[r][0]), int(sizes[r][1]), int(sizes[r][3]), int(sizes[r][2]), ptrs)
    gathered = parts
    graph.exchange_end()
    info = {"pieces": totals[0], "piece_bases": totals[1], "glue_records": totals[2],
            "bytes_gathered": sum(recv.numel() for recv, _ in gathered)}
    del gathered
    if sharded_join and world > 1:
        # every rank hash-joins 1/world of the junctions; the link arrays (one int32 per piece end, -1 = not
        # joined by this rank) are combined with ONE all-reduce(MAX): each end is set by exactly one rank
        n = graph.glue_join()
        links = torch.empty(max(n, 1), dtype=torch.int32, device=device)
        graph.glue_links_export(links.data_ptr(), n * 4)
        dist.all_reduce(links, op=dist.ReduceOp.MAX)
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        graph.glue_links_import(links.data_ptr(), n * 4)
        info["link_bytes_reduced"] = n * 4
    return info
