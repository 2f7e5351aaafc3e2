"""ctypes binding of the CPU oracle (oracle/cdbg_oracle.c).  TEST INFRASTRUCTURE:
imported only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ODIR, "_build", "liboracle.so")


def build():
    src_m = max(os.path.getmtime(os.path.join(ODIR, f)) for f in ("cdbg_oracle.c", "oracle_impl.h", "cdbg_oracle.h", "cpu_mt.cpp"))
    mt = os.path.join(ODIR, "_build", "libcpu_mt.so")
    if not os.path.exists(SO) or os.path.getmtime(SO) < src_m or not os.path.exists(mt):
        subprocess.check_call(["make", "-C", ODIR, "-s"])
    return SO


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.orc_build.restype = C.c_void_p
        L.orc_build.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int]
        L.orc_free.argtypes = [C.c_void_p]
        for f in ("orc_n_occurrences", "orc_n_distinct", "orc_n_solid", "orc_n_unitigs", "orc_total_bases", "orc_digest"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_unitig_seq.restype = C.c_char_p
        L.orc_unitig_seq.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_unitig_kc.restype = C.c_uint64
        L.orc_unitig_kc.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_unitig_circular.restype = C.c_int
        L.orc_unitig_circular.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_solid_kmer.restype = C.c_char_p
        L.orc_solid_kmer.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_solid_count.restype = C.c_uint32
        L.orc_solid_count.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_canonical_unitig.restype = C.c_void_p
        L.orc_canonical_unitig.argtypes = [C.c_char_p, C.c_uint64, C.c_int]
        L.orc_synth_reads.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]
        L.orc_synth_genome_len.restype = C.c_uint64
        L.orc_synth_genome_len.argtypes = [C.c_uint64, C.c_uint64]
        self._libc = C.CDLL(None)
        self._libc.free.argtypes = [C.c_void_p]

    def run(self, text, k, amin, want_solid=False):
        """-> dict(stats, unitigs=[(seq, kc)], digest, [solid=[(kmer,count)]])"""
        data = text if isinstance(text, bytes) else text.encode()
        r = self.lib.orc_build(data, len(data), k, amin)
        if not r:
            raise ValueError("orc_build rejected parameters")
        try:
            n = self.lib.orc_n_unitigs(r)
            out = {
                "stats": {"occurrences": self.lib.orc_n_occurrences(r), "distinct": self.lib.orc_n_distinct(r),
                          "solid": self.lib.orc_n_solid(r), "unitigs": n},
                "unitigs": [(self.lib.orc_unitig_seq(r, i).decode(), self.lib.orc_unitig_kc(r, i)) for i in range(n)],
                "circular": [self.lib.orc_unitig_circular(r, i) for i in range(n)],
                "digest": self.lib.orc_digest(r),
                "total_bases": self.lib.orc_total_bases(r),
            }
            if want_solid:
                ns = self.lib.orc_n_solid(r)
                out["solid"] = [(self.lib.orc_solid_kmer(r, i).decode(), self.lib.orc_solid_count(r, i)) for i in range(ns)]
            return out
        finally:
            self.lib.orc_free(r)

    def canonical_unitig(self, s, k):
        b = s.encode() if isinstance(s, str) else s
        p = self.lib.orc_canonical_unitig(b, len(b), k)
        try:
            return C.string_at(p).decode()
        finally:
            self._libc.free(p)

    def synth_reads(self, n_reads, read_len, cfg, first=0, total=None):
        total = n_reads if total is None else total
        buf = C.create_string_buffer(n_reads * (read_len + 1))
        self.lib.orc_synth_reads(buf, first, n_reads, total, read_len, cfg)
        return buf.raw


def load():
    return Oracle(C.CDLL(build()))


def cpu_mt_run(text, k, amin, threads):
    """oracle/cpu_mt.cpp: the multithreaded CPU restatement (k <= 63) -> dict of counts, set digest and seconds"""
    build()
    lib = C.CDLL(os.path.join(ODIR, "_build", "libcpu_mt.so"))
    lib.cpu_mt_run.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    out = (C.c_uint64 * 8)(); secs = (C.c_double * 4)()
    data = text if isinstance(text, bytes) else text.encode()
    rc = lib.cpu_mt_run(data, len(data), k, amin, threads, out, secs)
    if rc != 0:
        raise ValueError("cpu_mt_run: unsupported k")
    return {"occurrences": out[0], "distinct": out[1], "solid": out[2], "unitigs": out[3], "kc_sum": out[4], "set_digest": out[5],
            "unitig_bases": out[6], "s_count": secs[0], "s_solid": secs[1], "s_unitigs": secs[2], "s_total": secs[3]}


def canonical_set(oracle, unitigs, k):
    """[(seq, kc)] in any orientation/order -> sorted canonical list"""
    return sorted((oracle.canonical_unitig(s, k), int(kc)) for s, kc in unitigs)


def digest_of(canon_sorted):
    """same FNV-1a as orc_digest over a sorted canonical [(seq,kc)] list"""
    h = 0xcbf29ce484222325
    M = (1 << 64) - 1
    for s, kc in canon_sorted:
        for ch in s.encode():
            h ^= ch; h = (h * 0x100000001b3) & M
        for b in range(8):
            h ^= (kc >> (8 * b)) & 0xff; h = (h * 0x100000001b3) & M
        h ^= 0x0a; h = (h * 0x100000001b3) & M
    return h


def solid_sha256(solid):
    return hashlib.sha256("".join(f"{x} {c}\n" for x, c in solid).encode()).hexdigest()


def read_input(name):
    """tests/golden/inputs/<name>.fa|.txt -> read text with '\\n' separators"""
    base = os.path.join(ROOT, "tests", "golden", "inputs", name)
    if os.path.exists(base + ".txt"):
        return open(base + ".txt").read()
    recs, cur = [], []
    for line in open(base + ".fa"):
        line = line.strip()
        if line.startswith(">"):
            if cur:
                recs.append("".join(cur))
            cur = []
        elif line:
            cur.append(line)
    if cur:
        recs.append("".join(cur))
    return "\n".join(recs) + "\n"
