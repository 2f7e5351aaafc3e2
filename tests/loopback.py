"""In-process loop-back transport (TEST INFRASTRUCTURE): N ranks as N Python threads in ONE process, each with its
own libcdbg context; the four cdbg_transport functions rendezvous on a barrier and copy between the ranks' buffers
with hipMemcpy (device-to-device).  Lets the GPU suite run the complete multi-rank data path on a 1-GPU box, where
RCCL itself refuses two ranks on one device."""
import ctypes as C
import threading

from bcalm_amd import dist as cdist


class Loopback:
    def __init__(self, world, memcpy):
        self.world = world
        self.memcpy = memcpy                     # memcpy(dst_ptr, src_ptr, nbytes)
        self.barrier = threading.Barrier(world)
        self.slot = [None] * world

    def endpoint(self, rank):
        return _Endpoint(self, rank)


class _Endpoint:
    def __init__(self, hub, rank):
        self.hub, self.rank, self.error = hub, rank, None
        self._cbs = (cdist.FN_AG64(self._ag64), cdist.FN_A2AV(self._a2av), cdist.FN_AGV(self._agv), cdist.FN_ARMAX(self._armax))
        self.struct = cdist.Transport(None, *self._cbs)

    def attach(self, graph):
        graph._ck(graph.lib.cdbg_set_transport(graph._h, C.byref(self.struct)))
        graph._transport = self

    def _round(self, post, pull):
        try:
            self.hub.slot[self.rank] = post
            self.hub.barrier.wait(timeout=120)
            pull(self.hub.slot)
            self.hub.barrier.wait(timeout=120)
            return 0
        except Exception as e:
            self.error = e
            self.hub.barrier.abort()
            return -1

    def _ag64(self, user, send, recv, n):
        def pull(slots):
            for r, vals in enumerate(slots):
                for i in range(n):
                    recv[r * n + i] = vals[i]
        return self._round([send[i] for i in range(n)], pull)

    def _a2av(self, user, send, soff, scnt, recv, roff, rcnt):
        w = self.hub.world
        post = (send, [soff[r] for r in range(w)], [scnt[r] for r in range(w)])
        def pull(slots):
            for s, (sp, so, sc) in enumerate(slots):
                assert sc[self.rank] == rcnt[s], (s, sc[self.rank], rcnt[s])
                if rcnt[s]:
                    self.hub.memcpy(recv + roff[s], sp + so[self.rank], rcnt[s])
        return self._round(post, pull)

    def _agv(self, user, send, nbytes, recv, roff, rcnt):
        def pull(slots):
            for s, (sp, nb) in enumerate(slots):
                assert nb == rcnt[s]
                if nb:
                    self.hub.memcpy(recv + roff[s], sp, nb)
        return self._round((send, nbytes), pull)

    def _armax(self, user, dev, n):
        import numpy as np
        # gather every rank's array on the host, reduce, write back (test sizes only)
        def pull(slots):
            acc = None
            for sp in slots:
                h = (C.c_int32 * n)()
                self.hub.memcpy_d2h(C.addressof(h), sp, n * 4)
                a = np.frombuffer(h, dtype=np.int32)
                acc = a.copy() if acc is None else np.maximum(acc, a)
            self.result = acc
        def run():
            self.hub.slot[self.rank] = dev
            self.hub.barrier.wait(timeout=120)
            pull(self.hub.slot)
            self.hub.barrier.wait(timeout=120)          # everyone has read every array
            if n:
                self.hub.memcpy_h2d(dev, self.result.ctypes.data, n * 4)
            self.hub.barrier.wait(timeout=120)
        try:
            run()
            return 0
        except Exception as e:
            self.error = e
            self.hub.barrier.abort()
            return -1


def hip_loopback(world):
    """loop-back over device memory: copies through the process's HIP runtime"""
    import importlib.util
    import os
    spec = importlib.util.find_spec("torch")
    path = None
    for root in (spec.submodule_search_locations or []) if spec else []:
        cand = os.path.join(root, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            path = cand
    hip = C.CDLL(path or "libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    def cp(kind):
        def f(dst, src, n):
            rc = hip.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), n, kind)
            assert rc == 0, rc
        return f
    hub = Loopback(world, cp(3))                 # hipMemcpyDeviceToDevice
    hub.memcpy_d2h = cp(2); hub.memcpy_h2d = cp(1)
    return hub
