"""dev tool (CPU, numpy): records per junction of candidate partition-key schemes for k = 31 (VERDICT r4 next #3a).
A record is a run of consecutive junctions ((k-1)-mers, 30 bases) that select the same key occurrence; the scan pays per record
(one returning atomic + one 16-byte store), so the density of the selection scheme IS the scan's cost.  Every scheme here is
strand-symmetric (a junction and its reverse complement select the same canonical key), which the bucket rule needs.
  * random minimizer (shipped: canonical hash-ordered m-mer, m = 16, window of 15): density 2 / (w + 1)
  * shorter m: lower density, but the key repeats in the genome (m <= 14 overflowed the LDS tiers in round 3)
  * mod-minimizer (Groot Koerkamp & Pibiri 2024): a short canonical t-mer picks the position x, the key is the canonical m-mer at
    x mod w, t = m mod w (exactly the condition under which both strands pick the same m-mer); ties of the t-mer (frequent for
    t <= 5) broken by the smaller key hash so that the rule stays symmetric
Bound: a run of L junctions shares 30 - L + 1 bases, a key of m bases needs L <= 31 - m: density >= 1 / (31 - m) -- 1 / 15 at m = 16.
python bench_micro/density_schemes.py > profiles/r05_ab_partition_key_density.log"""
import numpy as np, sys
rng = np.random.default_rng(1)
N = 2_000_000
s = rng.integers(0, 4, N, dtype=np.uint64)
J = 30  # junction length (k-1)

def kmers(s, m):
    # forward and rc integer codes of all m-mers
    n = len(s) - m + 1
    f = np.zeros(n, dtype=np.uint64); r = np.zeros(n, dtype=np.uint64)
    for i in range(m):
        f = (f << np.uint64(2)) | s[i:i+n]
        r = r | ((np.uint64(3) - s[i:i+n]) << np.uint64(2*i))
    return f, r

def h64(x):
    x = x.copy()
    x ^= x >> np.uint64(33); x *= np.uint64(0xff51afd7ed558ccd); x ^= x >> np.uint64(33); x *= np.uint64(0xc4ceb9fe1a85ec53); x ^= x >> np.uint64(33)
    return x

def canon_hash(s, m):
    f, r = kmers(s, m)
    c = np.minimum(f, r)
    return h64(c + np.uint64(0x9e3779b97f4a7c15))

def sliding_argmin(h, w):
    # leftmost argmin over windows of w consecutive entries -> for window i: positions i..i+w-1
    n = len(h) - w + 1
    best = h[:n].copy(); arg = np.zeros(n, dtype=np.int64)
    for j in range(1, w):
        v = h[j:j+n]
        lt = v < best
        best = np.where(lt, v, best); arg = np.where(lt, j, arg)
    return best, arg

def runs(abs_pos):
    return 1 + int(np.count_nonzero(abs_pos[1:] != abs_pos[:-1]))

nj = N - J + 1
# (a) plain random minimizer, m=16
for m in (16, 15, 14, 12, 10):
    w = J - m + 1
    h = canon_hash(s, m)
    best, arg = sliding_argmin(h, w)
    ap = arg[:nj] + np.arange(nj)
    print("random minimizer m=%d w=%d: density %.4f (2/(w+1)=%.4f)" % (m, w, runs(ap)/nj, 2/(w+1)))

# (b) mod-minimizer, canonical: t-mer canonical hash picks x (ties -> symmetric rule: smallest sampled-m-mer hash among tied), sampled m-mer at x mod w
def modmini(m, t, tie="sym"):
    w = J - m + 1
    assert (m - t) % w == 0, (m, t, w)
    ht = canon_hash(s, t)
    nt = J - t + 1
    hm = canon_hash(s, m)
    best, arg = sliding_argmin(ht, nt)      # leftmost
    best = best[:nj]; arg = arg[:nj]
    if tie == "left":
        p = arg % w
        ap = p + np.arange(nj)
        return runs(ap)/nj, 0.0
    # symmetric: among all x with ht == best, candidates p = x mod w; choose min hm at p
    bk = np.full(nj, np.iinfo(np.uint64).max, dtype=np.uint64); bp = np.zeros(nj, dtype=np.int64)
    ties = np.zeros(nj, dtype=np.int64)
    idx = np.arange(nj)
    for x in range(nt):
        is_min = ht[x:x+nj] == best
        ties += is_min
        p = x % w
        key = hm[idx + p]
        better = is_min & (key < bk)
        bk = np.where(better, key, bk); bp = np.where(better, p, bp)
    ap = bp + idx
    return runs(ap)/nj, float(np.mean(ties > 1))

for (m, t) in ((18, 5), (17, 3), (19, 7), (20, 9), (16, 1), (17, 17), (22, 4), (21, 1), (23, 7)):
    try:
        d, tf = modmini(m, t)
        dl, _ = modmini(m, t, "left")
        print("mod-minimizer m=%d w=%d t=%d: density %.4f (ties in %.1f%% of windows); leftmost-tie (not strand-symmetric) %.4f" % (m, J-m+1, t, d, 100*tf, dl))
    except AssertionError as e:
        print("skip", m, t)
