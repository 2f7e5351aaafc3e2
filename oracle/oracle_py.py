"""oracle_py.py -- independent pure-Python restatement of the unitig spec.

TEST INFRASTRUCTURE ONLY (same status as oracle/cdbg_oracle.c: the checker, never
the product; only tests/, smoke() and bench.py's cpu_baseline may import it).

PARITY UNPINNED: the reference implementation (gatb-core) is absent from
/root/reference and its tests hold no expected outputs, so this file restates the
specification the tree does contain, written on strings and dicts so that it shares
no code and no data layout with cdbg_oracle.c:

  * nodes = distinct canonical k-mers            bidirected-graphs-in-bcalm2.md:64
  * edges = all (k-1)-overlaps, with signs        .md:39-46
  * unitig = path whose internal vertices have no other incident edges, whose
    first edge is the only out-edge with its sign at v0 and whose last edge is the
    only in-edge with its sign at vn              .md:83-88
  * abundance filter keeps count >= abundance_min README.md:23-25
  * canonical k-mer = lexicographic min of the two strands, k-mers with N skipped
                                                  scripts/unitigEvaluator.cpp:64-66,130-131
"""
from __future__ import annotations

import re
from collections import Counter

_COMP = str.maketrans("ACGT", "TGCA")


def revcomp(s: str) -> str:
    return s.translate(_COMP)[::-1]


def canonical(s: str) -> str:
    r = revcomp(s)
    return s if s <= r else r


def split_reads(text: str):
    """any character outside ACGT (case-insensitive) separates sequences"""
    return [t for t in re.split(r"[^ACGT]+", text.upper()) if t]


def count_kmers(text: str, k: int) -> Counter:
    c: Counter = Counter()
    for read in split_reads(text):
        for i in range(len(read) - k + 1):
            c[canonical(read[i:i + k])] += 1
    return c


def _out_edges(solid, node: str, sign: str):
    """edges leaving (node, sign): list of (to_node, to_sign)"""
    u = node if sign == "+" else revcomp(node)
    out = []
    for c in "ACGT":
        v = u[1:] + c
        cv = canonical(v)
        if cv in solid:
            out.append((cv, "+" if v == cv else "-"))
            if v == revcomp(v):        # even k, palindromic neighbour: label == rc(label), so the rows (s,+) and
                out.append((cv, "-"))  # (s,-) of the overlap table (.md:41-46) are two distinct edges (.md:7)
    return out


def _flip(sign: str) -> str:
    return "-" if sign == "+" else "+"


def _succ(solid, node, sign):
    o = _out_edges(solid, node, sign)
    if len(o) != 1:
        return None
    y, t = o[0]
    if y == node:                      # a path does not repeat vertices (.md:83)
        return None
    # in-edges of (y,t) are mirrors of the out-edges of (y, flip(t))  (.md:18-24)
    if len(_out_edges(solid, y, _flip(t))) != 1:
        return None
    return y, t


def canonical_unitig(s: str, k: int) -> str:
    """orientation- and (for cyclic unitigs) cut-point-normalised form"""
    r = revcomp(s)
    if len(s) >= k and s[:k - 1] == s[len(s) - (k - 1):]:
        n = len(s) - k + 1
        best = None
        for c in (s, r):
            cyc = c[:n]
            for rot in range(n):
                lin = "".join(cyc[(rot + i) % n] for i in range(len(s)))
                if best is None or lin < best:
                    best = lin
        return best
    return min(s, r)


def unitigs(text: str, k: int, abundance_min: int):
    """returns (sorted list of (canonical_seq, KC), stats dict)"""
    counts = count_kmers(text, k)
    solid = {x: c for x, c in counts.items() if c >= abundance_min}
    seen = set()
    out = []
    for x0 in sorted(solid):
        if x0 in seen:
            continue
        # walk backwards from (x0,+)
        x, s = x0, "+"
        while True:
            p = _succ(solid, x, _flip(s))
            if p is None:
                break
            x, s = p[0], _flip(p[1])
            if x == x0:
                break
        start = x
        seq = None
        kc = 0
        y, t = x, s
        while True:
            u = y if t == "+" else revcomp(y)
            seq = u if seq is None else seq + u[-1]
            seen.add(y)
            kc += solid[y]
            nx = _succ(solid, y, t)
            if nx is None or nx[0] == start:
                break
            y, t = nx
        out.append((canonical_unitig(seq, k), kc))
    out.sort()
    stats = {
        "occurrences": sum(counts.values()),
        "distinct": len(counts),
        "solid": len(solid),
        "unitigs": len(out),
    }
    return out, stats


def solid_kmers(text: str, k: int, abundance_min: int):
    return sorted((x, c) for x, c in count_kmers(text, k).items() if c >= abundance_min)


def read_fasta_text(path: str) -> str:
    """sequence lines joined per record, records separated by newline"""
    recs, cur = [], []
    with open(path) as f:
        for line in f:
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


def links(unitig_seqs, k: int):
    """Edges between unitigs, by brute force over all pairs of ends (spec:
    bidirected-graphs-in-bcalm2.md:39-46,100-103; token format README.md:72).
    -> set of (u, from_sign, v, to_sign): leaving u (as given for '+', reverse-complemented for
    '-') through its last k-1 bases equals entering v ('+': as given, '-': reverse-complemented)
    at its first k-1 bases."""
    out = set()
    ori = {"+": lambda s: s, "-": revcomp}
    for u, su in enumerate(unitig_seqs):
        for fs in "+-":
            tail = ori[fs](su)[-(k - 1):]
            for v, sv in enumerate(unitig_seqs):
                for ts in "+-":
                    if ori[ts](sv)[:k - 1] == tail:
                        out.add((u, fs, v, ts))
    return out
