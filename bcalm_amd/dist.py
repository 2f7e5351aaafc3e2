"""Multi-GPU set-up: one process per GPU.

The data path lives in libcdbg.so (include/cdbg.h, "Multi-GPU"): a context with world_size > 1 owns
the minimizer partitions p with p % world == rank.  Reads are either replicated (X0: every rank scans all reads for its own
partitions, nothing travels) or sharded (X1: one all-to-all-v of super-k-mer records to the partition owners).  The glue is
sharded by owner: junction records travel once to the rank their key hashes to, joined pairs once to the owner of the end,
list ranking runs where the pieces live with one query / reply all-to-all-v per round, and every piece travels once to the
owner of its unitig's head -- each rank ends with the unitigs it owns, their union is the graph.  (emit_replicated = 1, or
closed chains that cross ranks: the replicated exchange -- pieces + junction log all-gathered, every rank ends with the
complete set.)  The bytes move through the context's TRANSPORT:

  * `init_rccl(graph, dist)`      the product path: RCCL inside libcdbg.so (ncclSend/ncclRecv all-to-all-v,
                                  ncclAllGather, ncclAllReduce over xGMI).  torch.distributed is used for ONE thing:
                                  broadcasting rank 0's 128-byte ncclUniqueId.
  * `TorchTransport(dist)`        the four transport functions implemented with torch.distributed on HOST memory
                                  (gloo): what the CPU tests plug into the kernel-logic simulator, whose "device"
                                  pointers are host pointers.  Never used on a GPU.
"""
from __future__ import annotations

import ctypes as C

import torch

U64P = C.POINTER(C.c_uint64)
FN_AG64 = C.CFUNCTYPE(C.c_int, C.c_void_p, U64P, U64P, C.c_int)
FN_A2AV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, U64P, U64P, C.c_void_p, U64P, U64P)
FN_AGV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, U64P, U64P)
FN_ARMAX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64)


class Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_gather_u64", FN_AG64), ("all_to_all_v", FN_A2AV),
                ("all_gather_v", FN_AGV), ("all_reduce_max_i32", FN_ARMAX)]


def init_rccl(graph, dist, device=None):
    """RCCL communicator inside libcdbg for `graph` (created with world_size / rank); collective over `dist`'s group."""
    uid = torch.zeros(128, dtype=torch.uint8)
    if dist.get_rank() == 0:
        buf = (C.c_uint8 * 128)()
        graph._ck(graph.lib.cdbg_comm_unique_id(buf))
        uid = torch.tensor(list(buf), dtype=torch.uint8)
    if device is not None:
        uid = uid.to(device)
    dist.broadcast(uid, src=0)
    raw = bytes(uid.cpu().tolist())
    graph._ck(graph.lib.cdbg_comm_init_rccl(graph._h, raw))


def _host_tensor(ptr, nbytes):
    """zero-copy uint8 view of host memory (the simulator's 'device' buffers)"""
    if not nbytes:
        return torch.empty(0, dtype=torch.uint8)
    return torch.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=torch.uint8)


class TorchTransport:
    """cdbg_transport over torch.distributed point-to-point / collectives on host memory (gloo). Keep the object
    alive as long as the graph uses it (it owns the ctypes callbacks)."""

    def __init__(self, dist):
        self.dist = dist
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.error = None
        self._cbs = (FN_AG64(self._ag64), FN_A2AV(self._a2av), FN_AGV(self._agv), FN_ARMAX(self._armax))
        self.struct = Transport(None, *self._cbs)

    def attach(self, graph):
        graph._ck(graph.lib.cdbg_set_transport(graph._h, C.byref(self.struct)))
        graph._transport = self                          # keep-alive

    def _guard(self, fn):
        try:
            fn()
            return 0
        except Exception as e:                           # a Python exception must not unwind through C
            self.error = e
            return -1

    def _ag64(self, user, send, recv, n):
        def run():
            mine = torch.tensor([send[i] for i in range(n)], dtype=torch.int64)
            out = [torch.empty(n, dtype=torch.int64) for _ in range(self.world)]
            self.dist.all_gather(out, mine)
            for r in range(self.world):
                for i in range(n):
                    recv[r * n + i] = int(out[r][i]) & 0xFFFFFFFFFFFFFFFF
        return self._guard(run)

    def _exchange(self, sends, recvs):
        """sends[r] / recvs[r]: uint8 tensors for / from rank r (own slot: local copy)"""
        ops = []
        for r in range(self.world):
            if r == self.rank:
                continue
            if sends[r].numel():
                ops.append(self.dist.P2POp(self.dist.isend, sends[r].clone(), r))
            if recvs[r].numel():
                ops.append(self.dist.P2POp(self.dist.irecv, recvs[r], r))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        if recvs[self.rank].numel():
            recvs[self.rank].copy_(sends[self.rank])

    def _a2av(self, user, send, soff, scnt, recv, roff, rcnt):
        def run():
            sends = [_host_tensor(send + soff[r], scnt[r]) for r in range(self.world)]
            recvs = [_host_tensor(recv + roff[r], rcnt[r]) for r in range(self.world)]
            self._exchange(sends, recvs)
        return self._guard(run)

    def _agv(self, user, send, nbytes, recv, roff, rcnt):
        def run():
            mine = _host_tensor(send, nbytes)
            sends = [mine for _ in range(self.world)]
            recvs = [_host_tensor(recv + roff[r], rcnt[r]) for r in range(self.world)]
            self._exchange(sends, recvs)
        return self._guard(run)

    def _armax(self, user, dev, n):
        def run():
            if n:
                t = _host_tensor(dev, n * 4).view(torch.int32)
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return self._guard(run)
