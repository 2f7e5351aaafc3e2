"""ctypes binding of libcdbg.so (include/cdbg.h) -- the host-side Python mirror.

Plain pointers and sizes only; torch is not involved.  `load()` opens the HIP build
(bcalm_amd/_build/libcdbg.so); there is no CPU fallback: without the library or
without a HIP device every entry point raises.  (tests/ may pass `path=` to bind the
kernel-logic simulator built by tests/hostsim/build.sh; nothing in this package does.)
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "_build", "libcdbg.so")


class CdbgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libcdbg error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [("k", C.c_int), ("abundance_min", C.c_int), ("minimizer_size", C.c_int),
                ("log2_partitions", C.c_int), ("device_id", C.c_int), ("world_size", C.c_int),
                ("rank", C.c_int), ("all_abundance_counts", C.c_int), ("emit_replicated", C.c_int), ("reads_replicated", C.c_int)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "input_bytes", "n_records", "n_member_kmers", "n_occurrences", "n_distinct", "n_solid",
        "n_solid_travellers", "n_pieces", "n_glue_open_ends", "n_glue_joined", "n_unitigs",
        "unitig_bases", "n_big_partitions", "n_cycles")] + [
        ("minimizer_size", C.c_int), ("log2_partitions", C.c_int), ("kmer_words", C.c_int)] + [
        (n, C.c_float) for n in ("ms_scan_hist", "ms_scan_emit", "ms_count", "ms_compact", "ms_glue", "ms_total", "ms_exchange")] + [
        (n, C.c_uint64) for n in ("n_launch_scan", "n_launch_count", "n_launch_compact", "n_multipass_partitions", "n_tiles_overlapped", "n_split_buckets", "n_glue_rounds", "n_walked_unitigs", "n_deferred_records")] + [
        ("ms_place", C.c_float), ("count_slices", C.c_int)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


ABI_VERSION = 7                                        # include/cdbg.h CDBG_ABI_VERSION this binding was written for
EXPORTS = ["cdbg_abi_version", "cdbg_stats_sizeof", "cdbg_create", "cdbg_destroy", "cdbg_release_cached", "cdbg_last_error", "cdbg_push_reads", "cdbg_push_text",
           "cdbg_generate_reads", "cdbg_expect_input", "cdbg_stage_acquire", "cdbg_stage_commit", "cdbg_read_text", "cdbg_count", "cdbg_compact", "cdbg_glue", "cdbg_run", "cdbg_reset",
           "cdbg_num_solid", "cdbg_fetch_solid", "cdbg_num_unitigs", "cdbg_fetch_unitigs", "cdbg_stats", "cdbg_digest", "cdbg_verify",
           "cdbg_verify_edges", "cdbg_verify_unitigs",
           "cdbg_fetch_unitigs_packed", "cdbg_fetch_unitig_abundances", "cdbg_link", "cdbg_num_links", "cdbg_fetch_links", "cdbg_unitig_id_base",
           "cdbg_set_transport", "cdbg_comm_unique_id", "cdbg_comm_init_rccl", "cdbg_comm_bytes"]


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64 and looks it up by file name, libcdbg.so
    asks for the soname libamdhip64.so.7: whichever of the two is loaded first decides whether the process ends up
    with one runtime (torch first) or two (libcdbg first; the second one to touch the device then finds no GPU).
    Loading torch's copy before libcdbg.so -- without importing torch -- makes every order work."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        for root in (spec.submodule_search_locations or []) if spec else []:
            cand = os.path.join(root, "lib", "libamdhip64.so")
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
                return
    except Exception:
        pass                                             # no torch, or not a ROCm build: libcdbg uses the system runtime


def load(path: str | None = None) -> C.CDLL:
    path = path or DEFAULT_LIB
    if not os.path.exists(path):
        raise CdbgError(-2, f"{path} not found: build the HIP extension first "
                            f"(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    if os.path.abspath(path) == os.path.abspath(DEFAULT_LIB):
        _share_hip_runtime_with_torch()
    lib = C.CDLL(path)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    lib.cdbg_stats_sizeof.restype = u64
    if lib.cdbg_abi_version() != ABI_VERSION or lib.cdbg_stats_sizeof() != C.sizeof(Stats):
        raise CdbgError(-1, f"{path}: ABI version {lib.cdbg_abi_version()} / cdbg_stats_t of {lib.cdbg_stats_sizeof()} bytes, "
                            f"this binding expects version {ABI_VERSION} / {C.sizeof(Stats)} bytes")
    lib.cdbg_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    lib.cdbg_destroy.argtypes = [vp]
    lib.cdbg_destroy.restype = None
    lib.cdbg_last_error.restype = C.c_char_p
    lib.cdbg_push_reads.argtypes = [vp, C.c_char_p, C.POINTER(u64), u64]
    lib.cdbg_push_text.argtypes = [vp, C.c_char_p, u64]
    lib.cdbg_generate_reads.argtypes = [vp, u64, u64, u64, u64, i32]
    lib.cdbg_expect_input.argtypes = [vp, u64]
    lib.cdbg_stage_acquire.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    lib.cdbg_stage_commit.argtypes = [vp, vp, u64]
    lib.cdbg_read_text.argtypes = [vp, u64, u64, C.c_char_p]
    for f in ("cdbg_count", "cdbg_compact", "cdbg_glue", "cdbg_run", "cdbg_reset"):
        getattr(lib, f).argtypes = [vp]
    lib.cdbg_num_solid.argtypes = [vp, C.POINTER(u64)]
    lib.cdbg_fetch_solid.argtypes = [vp, C.c_char_p, C.POINTER(C.c_uint32), u64, C.POINTER(u64)]
    lib.cdbg_num_unitigs.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    lib.cdbg_fetch_unitigs.argtypes = [vp, u64, u64, C.c_char_p, C.POINTER(u64), C.POINTER(u64)]
    lib.cdbg_fetch_unitigs_packed.argtypes = [vp, C.POINTER(C.c_uint8), u64, C.POINTER(u64), C.POINTER(C.c_uint32), C.POINTER(u64)]
    lib.cdbg_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.cdbg_digest.argtypes = [vp, C.POINTER(u64)]
    lib.cdbg_verify.argtypes = [vp, C.POINTER(u64)]
    lib.cdbg_verify_edges.argtypes = [vp, C.POINTER(u64)]
    lib.cdbg_verify_unitigs.argtypes = [vp, C.c_char_p, C.POINTER(u64), u64, C.POINTER(u64)]
    lib.cdbg_release_cached.argtypes = []
    lib.cdbg_fetch_unitig_abundances.argtypes = [vp, u64, u64, C.POINTER(C.c_uint32), C.POINTER(u64)]
    lib.cdbg_link.argtypes = [vp]
    lib.cdbg_num_links.argtypes = [vp, C.POINTER(u64)]
    lib.cdbg_fetch_links.argtypes = [vp, C.POINTER(u64), C.POINTER(C.c_uint32)]
    lib.cdbg_unitig_id_base.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    lib.cdbg_set_transport.argtypes = [vp, vp]
    lib.cdbg_comm_unique_id.argtypes = [vp]
    lib.cdbg_comm_init_rccl.argtypes = [vp, C.c_char_p]
    lib.cdbg_comm_bytes.argtypes = [vp, C.POINTER(u64)]
    return lib


class Graph:
    """One construction job.  Mirrors the reference's single entry point
    GraphUnitigsTemplate<span>::create(props, false) (/root/reference/src/bcalm_1.cpp:57)
    with its three properties -kmer-size / -abundance-min / -in, staged as
    count() -> compact() -> glue() (or run())."""

    def __init__(self, k: int, abundance_min: int = 2, minimizer_size: int = 0, log2_partitions: int = -1,
                 device_id: int = 0, world_size: int = 1, rank: int = 0, lib: C.CDLL | None = None,
                 all_abundance_counts: bool = False, emit_replicated: bool = False, reads_replicated: bool = False):
        self.lib = lib or load()
        self.k = k
        p = Params(k, abundance_min, minimizer_size, log2_partitions, device_id, world_size, rank, 1 if all_abundance_counts else 0,
                   1 if emit_replicated else 0, 1 if reads_replicated else 0)
        self._h = C.c_void_p()
        self._ck(self.lib.cdbg_create(C.byref(p), C.byref(self._h)))

    def _ck(self, rc):
        if rc != 0:
            raise CdbgError(rc, (self.lib.cdbg_last_error() or b"").decode())

    def close(self):
        if self._h:
            self.lib.cdbg_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- input ----
    def push_text(self, text):
        b = text if isinstance(text, (bytes, bytearray)) else text.encode()
        self._ck(self.lib.cdbg_push_text(self._h, bytes(b), len(b)))

    def stage_text(self, text):
        """push_text through the zero-copy staging calls (cdbg_stage_acquire / cdbg_stage_commit): `text` is cut at sequence
        boundaries into pieces that fit a staging buffer"""
        b = text if isinstance(text, (bytes, bytearray)) else text.encode()
        pos = 0
        while pos < len(b):
            buf, cap = C.c_void_p(), C.c_uint64()
            self._ck(self.lib.cdbg_stage_acquire(self._h, C.byref(buf), C.byref(cap)))
            end = min(len(b), pos + cap.value - 1)
            if end < len(b):
                cut = b.rfind(b"\n", pos, end)
                if cut <= pos:
                    self.lib.cdbg_stage_commit(self._h, buf, 0)
                    raise ValueError("a sequence longer than a staging buffer: split it with a k-1 overlap")
                end = cut + 1
            C.memmove(buf, bytes(b[pos:end]), end - pos)
            self._ck(self.lib.cdbg_stage_commit(self._h, buf, end - pos))
            pos = end

    def expect_input(self, nbytes):
        self._ck(self.lib.cdbg_expect_input(self._h, nbytes))

    def push_reads(self, reads):
        seqs = [r if isinstance(r, bytes) else r.encode() for r in reads]
        offs = (C.c_uint64 * (len(seqs) + 1))()
        acc = 0
        for i, s in enumerate(seqs):
            offs[i] = acc
            acc += len(s)
        offs[len(seqs)] = acc
        self._ck(self.lib.cdbg_push_reads(self._h, b"".join(seqs), offs, len(seqs)))

    def generate_reads(self, n_reads, read_len, cfg, first_read=0, total_reads=None):
        total = n_reads if total_reads is None else total_reads
        self._ck(self.lib.cdbg_generate_reads(self._h, first_read, n_reads, total, read_len, cfg))

    def read_text(self, first, n):
        buf = C.create_string_buffer(n)
        self._ck(self.lib.cdbg_read_text(self._h, first, n, buf))
        return buf.raw

    # ---- stages ----
    def count(self):
        self._ck(self.lib.cdbg_count(self._h))

    def compact(self):
        self._ck(self.lib.cdbg_compact(self._h))

    def glue(self):
        self._ck(self.lib.cdbg_glue(self._h))

    def run(self):
        self._ck(self.lib.cdbg_run(self._h))

    def reset(self):
        self._ck(self.lib.cdbg_reset(self._h))

    def unitigs_packed(self):
        """the unitig set at 2 bits per base as the glue stage leaves it in HBM: (arena bytes, base_off[], len[], kc[])"""
        n = C.c_uint64(); tot = C.c_uint64()
        self._ck(self.lib.cdbg_num_unitigs(self._h, C.byref(n), C.byref(tot)))
        nb = (tot.value + 3) // 4
        buf = (C.c_uint8 * max(nb, 1))(); off = (C.c_uint64 * max(n.value, 1))(); ln = (C.c_uint32 * max(n.value, 1))(); kc = (C.c_uint64 * max(n.value, 1))()
        self._ck(self.lib.cdbg_fetch_unitigs_packed(self._h, buf, nb, off, ln, kc))
        return bytes(buf)[:nb], list(off)[:n.value], list(ln)[:n.value], list(kc)[:n.value]

    def unitig_abundances(self):
        """-> per unitig (same order as unitigs()) the list of its k-mers' abundances (ab:Z: vector)"""
        n, tb = C.c_uint64(), C.c_uint64()
        self._ck(self.lib.cdbg_num_unitigs(self._h, C.byref(n), C.byref(tb)))
        n, tb = n.value, tb.value
        if n == 0:
            return []
        ab = (C.c_uint32 * max(tb, 1))()
        off = (C.c_uint64 * (n + 1))()
        self._ck(self.lib.cdbg_fetch_unitig_abundances(self._h, 0, n, ab, off))
        return [list(ab[off[i]:off[i + 1]]) for i in range(n)]

    def unitig_id_base(self):
        """-> (first job-wide unitig id of this rank's share, unitigs of the whole job); after links() / cdbg_link"""
        a, b = C.c_uint64(), C.c_uint64()
        self._ck(self.lib.cdbg_unitig_id_base(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def links(self):
        """-> per unitig (same order as unitigs()) a list of (from_sign, target_unitig, to_sign); a sharded set (several ranks,
        emit_replicated = 0): collective, and target_unitig is a job-wide id (unitig_id_base() + the position in the owner's unitigs())"""
        self._ck(self.lib.cdbg_link(self._h))
        n, tb = C.c_uint64(), C.c_uint64()
        self._ck(self.lib.cdbg_num_unitigs(self._h, C.byref(n), C.byref(tb)))
        U = n.value
        nl = C.c_uint64()
        self._ck(self.lib.cdbg_num_links(self._h, C.byref(nl)))
        off = (C.c_uint64 * (2 * U + 1))()
        to = (C.c_uint32 * max(nl.value, 1))()
        self._ck(self.lib.cdbg_fetch_links(self._h, off, to))
        out = []
        for u in range(U):
            l = []
            for side, fs in ((1, "+"), (0, "-")):
                for i in range(off[2 * u + side], off[2 * u + side + 1]):
                    t = to[i]
                    l.append((fs, t >> 1, "+" if (t & 1) == 0 else "-"))
            out.append(l)
        return out

    def stats(self) -> dict:
        s = Stats()
        self._ck(self.lib.cdbg_stats(self._h, C.byref(s)))
        return s.as_dict()

    def comm_bytes(self):
        """bytes this rank sent + received through the transport since the last reset()"""
        n = C.c_uint64()
        self._ck(self.lib.cdbg_comm_bytes(self._h, C.byref(n)))
        return n.value

    def digest(self):
        """device-side digests of the result: {kc_sum, solid_count_sum, set_digest, kmers_in_unitigs}"""
        out = (C.c_uint64 * 4)()
        self._ck(self.lib.cdbg_digest(self._h, out))
        return {"kc_sum": out[0], "solid_count_sum": out[1], "set_digest": out[2], "kmers_in_unitigs": out[3]}

    def verify(self, edges=True):
        """the unitig definition checked on the device, without the oracle (cdbg_verify, bcalm_amd/csrc/k_verify.h):
        k-mer multiset of the unitigs == solid set (counts + two commutative sums), and no pair of unitig ends that are each
        other's only link (maximality).  mergeable_ends is None on a rank that holds a share of the unitigs.
        edges: also run the edge-conservation pass (verify_edges: a 12-byte x pow2(3 x n_solid) table).  A graph too large for that
        pass (or a card without room for its table) reports edges = None -- with the reason under "edges_error" -- instead of failing
        the k-mer-set / maximality verdict; edges=False skips the pass."""
        out = (C.c_uint64 * 8)()
        self._ck(self.lib.cdbg_verify(self._h, out))
        e, why = None, None
        if edges:
            try:
                e = self.verify_edges()
            except CdbgError as ex:
                why = str(ex)
        d = self._verify_dict(out, e)
        if why is not None:
            d["edges_error"] = why
        return d

    @staticmethod
    def _verify_dict(out, edges):
        none = 0xFFFFFFFFFFFFFFFF
        return {"unitig_kmers": (out[0], out[1], out[2]), "solid_kmers": (out[3], out[4], out[5]),
                "mergeable_ends": None if out[6] == none else out[6], "closed_chains": None if out[7] == none else out[7],
                "edges": edges}

    def verify_edges(self):
        """edge conservation (cdbg_verify_edges): {graph: D, links: L, inner: 2 sum(LN - k), junctions}; a unitig set whose every
        inner junction is 1-in / 1-out has graph == links + inner.  None on a multi-rank job."""
        out = (C.c_uint64 * 4)()
        self._ck(self.lib.cdbg_verify_edges(self._h, out))
        if out[0] == 0xFFFFFFFFFFFFFFFF:
            return None
        return {"graph": out[0], "links": out[1], "inner": out[2], "junctions": out[3]}

    def verify_unitigs(self, seqs):
        """the checks of verify() for a unitig set supplied by the caller (sequences as str / bytes) against the resident solid k-mers"""
        bs = [x if isinstance(x, bytes) else x.encode() for x in seqs]
        off = (C.c_uint64 * (len(bs) + 1))()
        acc = 0
        for i, b in enumerate(bs):
            off[i] = acc
            acc += len(b)
        off[len(bs)] = acc
        out = (C.c_uint64 * 12)()
        self._ck(self.lib.cdbg_verify_unitigs(self._h, b"".join(bs), off, len(bs), out))
        return self._verify_dict(out, {"graph": out[8], "links": out[9], "inner": out[10], "junctions": out[11]})

    @staticmethod
    def edges_conserved(v):
        e = v["edges"]
        return e is None or e["graph"] == e["links"] + e["inner"]

    def release_cached(self):
        """hand the process-wide pool of device buffers back to the driver (cdbg_release_cached): call after close() when torch,
        RCCL or another library of the same process needs the HBM"""
        self.lib.cdbg_release_cached()

    def solid_kmers(self):
        n = C.c_uint64()
        self._ck(self.lib.cdbg_num_solid(self._h, C.byref(n)))
        n = n.value
        if n == 0:
            return []
        kb = C.create_string_buffer(n * (self.k + 1))
        cnt = (C.c_uint32 * n)()
        nw = C.c_uint64()
        self._ck(self.lib.cdbg_fetch_solid(self._h, kb, cnt, n, C.byref(nw)))
        raw = kb.raw
        k1 = self.k + 1
        return sorted((raw[i * k1:i * k1 + self.k].decode(), cnt[i]) for i in range(nw.value))

    def unitigs(self, first=0, count=None):
        """-> [(sequence, KC)] of unitigs [first, first + count) (default: all) in the library's arbitrary order/orientation"""
        n, tb = C.c_uint64(), C.c_uint64()
        self._ck(self.lib.cdbg_num_unitigs(self._h, C.byref(n), C.byref(tb)))
        n, tb = n.value, tb.value
        if count is not None or first:
            n = max(0, min(n - first, n if count is None else count))
        if n == 0:
            return []
        seq = C.create_string_buffer(max(tb, 1))
        off = (C.c_uint64 * (n + 1))()
        kc = (C.c_uint64 * n)()
        self._ck(self.lib.cdbg_fetch_unitigs(self._h, first, n, seq, off, kc))
        raw = seq.raw
        return [(raw[off[i]:off[i + 1]].decode(), kc[i]) for i in range(n)]
