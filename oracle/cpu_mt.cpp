/* cpu_mt.cpp -- multithreaded CPU restatement of the reads -> unitigs path (k <= 63: one- and two-word k-mers), the CPU BASELINE of bench.py.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE, never linked into the product (see oracle/cdbg_oracle.c for the checker proper).
 * "CPU restatement, NOT BCALM 2": gatb-core, where the reference's implementation of this path lives, is an absent
 * submodule (/root/reference/.gitmodules:1-3), so its binary cannot be timed here (SURVEY.md section 8 d ii asks for
 * exactly this fallback: std::thread x all cores, one shared table, one input).
 *
 * Same definitions as the oracle (oracle/oracle_impl.h): canonical k-mers counted in ONE shared lock-free table
 * (README.md:23-25: keep count >= abundance-min), then the unitig definition of
 * bidirected-graphs-in-bcalm2.md:83-88 applied to the solid set: out-edges by probing the four extensions, an edge is
 * compactable iff it is the only out-edge of its source, the only in-edge of its target, and not a self-loop / hairpin.
 * Parallel over the solid table: every thread starts unitigs at the ends it finds in its slot range; the end with the
 * smaller (slot, sign) id emits the unitig.  Isolated cycles are swept up sequentially afterwards.
 * The result is reported as counts plus the set digest of bcalm_amd/csrc/k_links.h (k_digest_unitigs), so that tests can
 * pin this program against the oracle's unitigs.
 */
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

typedef unsigned __int128 km2_t;                       /* two-word k-mers (32 <= k <= 63): config 4's baseline is the same kind as config 3's */

inline int code(unsigned char c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; }
    return -1;
}
inline uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31);
}
inline uint64_t rc64(uint64_t x) {                      /* complement, then reverse the 2-bit groups of the word */
    x = ~x;
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    return __builtin_bswap64(x);
}
inline uint64_t rc_of(uint64_t x, int k) { return rc64(x) >> (64 - 2 * k); }
inline km2_t rc_of(km2_t x, int k) { return (((km2_t)rc64((uint64_t)x) << 64) | (km2_t)rc64((uint64_t)(x >> 64))) >> (128 - 2 * k); }
inline uint64_t hash_of(uint64_t x) { return mix64(x); }
inline uint64_t hash_of(km2_t x) { return mix64((uint64_t)x ^ mix64((uint64_t)(x >> 64))); }

template <class KM> struct Table;
/* one-word keys: the key word itself is claimed (all ones = empty: unreachable for k <= 31) */
template <> struct Table<uint64_t> {
    typedef uint64_t km_t;
    std::vector<km_t> keys; std::vector<uint32_t> cnt; uint64_t mask = 0;
    void init(uint64_t cap) { keys.assign(cap, ~0ULL); cnt.assign(cap, 0); mask = cap - 1; }
    void release() { std::vector<km_t>().swap(keys); std::vector<uint32_t>().swap(cnt); }
    bool used(uint64_t s) const { return keys[s] != ~0ULL; }
    km_t key(uint64_t s) const { return keys[s]; }
    uint64_t slot_insert(km_t key) {                       /* find-or-insert, lock-free */
        uint64_t s = mix64(key) & mask;
        for (;;) {
            km_t cur = __atomic_load_n(&keys[s], __ATOMIC_RELAXED);
            if (cur == key) return s;
            if (cur == ~0ULL) {
                km_t exp = ~0ULL;
                if (__atomic_compare_exchange_n(&keys[s], &exp, key, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED) || exp == key) return s;
            }
            s = (s + 1) & mask;
        }
    }
    int64_t find(km_t key) const {
        uint64_t s = mix64(key) & mask;
        for (;;) {
            const km_t cur = keys[s];
            if (cur == key) return (int64_t)s;
            if (cur == ~0ULL) return -1;
            s = (s + 1) & mask;
        }
    }
};
/* two-word keys: a state byte per slot is claimed (0 empty -> 1 being written -> 2 published); no 16-byte atomics */
template <> struct Table<km2_t> {
    typedef km2_t km_t;
    std::vector<uint64_t> lo, hi; std::vector<uint8_t> st; std::vector<uint32_t> cnt; uint64_t mask = 0;
    void init(uint64_t cap) { lo.assign(cap, 0); hi.assign(cap, 0); st.assign(cap, 0); cnt.assign(cap, 0); mask = cap - 1; }
    void release() { std::vector<uint64_t>().swap(lo); std::vector<uint64_t>().swap(hi); std::vector<uint8_t>().swap(st); std::vector<uint32_t>().swap(cnt); }
    bool used(uint64_t s) const { return st[s] == 2; }
    km_t key(uint64_t s) const { return ((km_t)hi[s] << 64) | (km_t)lo[s]; }
    uint64_t slot_insert(km_t key) {
        const uint64_t klo = (uint64_t)key, khi = (uint64_t)(key >> 64);
        uint64_t s = hash_of(key) & mask;
        for (;;) {
            const uint8_t v = __atomic_load_n(&st[s], __ATOMIC_ACQUIRE);
            if (v == 2) { if (lo[s] == klo && hi[s] == khi) return s; s = (s + 1) & mask; continue; }
            if (v == 0) {
                uint8_t exp = 0;
                if (__atomic_compare_exchange_n(&st[s], &exp, (uint8_t)1, false, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED)) {
                    lo[s] = klo; hi[s] = khi;
                    __atomic_store_n(&st[s], (uint8_t)2, __ATOMIC_RELEASE);
                    return s;
                }
            }
            /* being written by another thread: look again */
        }
    }
    int64_t find(km_t key) const {
        const uint64_t klo = (uint64_t)key, khi = (uint64_t)(key >> 64);
        uint64_t s = hash_of(key) & mask;
        for (;;) {
            if (st[s] == 0) return -1;
            if (lo[s] == klo && hi[s] == khi) return (int64_t)s;
            s = (s + 1) & mask;
        }
    }
};

struct Ref { int64_t node; int sign; };

template <class KM>
struct Graph {
    const Table<KM>* S; int k; KM kmask;
    KM oriented(int64_t x, int sign) const { const KM u = S->key((uint64_t)x); return sign ? rc_of(u, k) : u; }
    int out_edges(int64_t x, int sign, Ref out[5]) const {
        const KM u = oriented(x, sign);
        int n = 0;
        for (unsigned c = 0; c < 4; ++c) {
            const KM v = ((u << 2) | (KM)c) & kmask, r = rc_of(v, k);
            const int vs = v <= r ? 0 : 1;
            const int64_t y = S->find(vs ? r : v);
            if (y >= 0) {
                out[n].node = y; out[n].sign = vs; ++n;
                if (v == r) { out[n].node = y; out[n].sign = 1; ++n; }   /* even k, palindromic neighbour: two edges (.md:7,41-46) */
            }
        }
        return n;
    }
    Ref succ(int64_t x, int sign) const {
        Ref none = { -1, 0 }, o[5], b[5];
        if (out_edges(x, sign, o) != 1) return none;
        if (o[0].node == x) return none;
        if (out_edges(o[0].node, !o[0].sign, b) != 1) return none;
        return o[0];
    }
};

/* orientation-independent hash of one unitig (+ KC): the formula of k_digest_unitigs */
inline uint64_t unitig_digest(const std::string& s, uint64_t kc) {
    const uint64_t B = 0x100000001B3ULL;
    uint64_t hf = 0, hr = 0; const size_t n = s.size();
    for (size_t i = 0; i < n; ++i) {
        hf = hf * B + (uint64_t)(code((unsigned char)s[i]) + 1);
        hr = hr * B + (uint64_t)(3 - code((unsigned char)s[n - 1 - i]) + 1);
    }
    return mix64((hf + hr) ^ mix64(hf * hr + kc));
}

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }


template <class km_t>
int run_t(const char* text, uint64_t n, int k, int amin, int n_threads, uint64_t out[8], double secs[4]) {
    if (n_threads < 1) n_threads = 1;
    const km_t kmask = (~(km_t)0) >> (8 * (int)sizeof(km_t) - 2 * k);
    const double t0 = now();
    /* ---- 1. count: threads take byte ranges cut at separators ---- */
    std::vector<uint64_t> cut(n_threads + 1, n);
    cut[0] = 0;
    for (int t = 1; t < n_threads; ++t) { uint64_t p = n / n_threads * t; while (p < n && code((unsigned char)text[p]) >= 0) ++p; cut[t] = p; }
    uint64_t cap = 1024; while (cap < (uint64_t)(0.7 * (double)n) + 1024) cap <<= 1;     /* distinct <= ~0.33 n at 30x with 1 % errors: load <= 0.5 */
    Table<km_t> A; A.init(cap);
    std::vector<uint64_t> occ_t(n_threads, 0);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back([&, t]() {
            km_t fw = 0, rc = 0; int run = 0; uint64_t occ = 0;
            for (uint64_t i = cut[t]; i < cut[t + 1]; ++i) {
                const int c = code((unsigned char)text[i]);
                if (c < 0) { run = 0; continue; }
                fw = ((fw << 2) | (km_t)c) & kmask;
                rc = (rc >> 2) | ((km_t)(3 - c) << (2 * (k - 1)));
                if (++run >= k) { const uint64_t s = A.slot_insert(fw <= rc ? fw : rc); __atomic_fetch_add(&A.cnt[s], 1u, __ATOMIC_RELAXED); ++occ; }
            }
            occ_t[t] = occ;
        });
        for (auto& x : th) x.join();
    }
    const double t1 = now();
    /* ---- 2. solid table ---- */
    uint64_t n_distinct = 0, n_solid = 0, n_occ = 0;
    for (uint64_t v : occ_t) n_occ += v;
    {
        std::vector<uint64_t> d(n_threads, 0), s(n_threads, 0);
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back([&, t]() {
            for (uint64_t i = cap / n_threads * t, e = t + 1 == n_threads ? cap : cap / n_threads * (t + 1); i < e; ++i)
                if (A.used(i)) { ++d[t]; if (A.cnt[i] >= (uint32_t)amin) ++s[t]; }
        });
        for (auto& x : th) x.join();
        for (int t = 0; t < n_threads; ++t) { n_distinct += d[t]; n_solid += s[t]; }
    }
    uint64_t scap = 1024; while (scap * 7 < n_solid * 10 + 16) scap <<= 1;
    Table<km_t> S; S.init(scap);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back([&, t]() {
            for (uint64_t i = cap / n_threads * t, e = t + 1 == n_threads ? cap : cap / n_threads * (t + 1); i < e; ++i)
                if (A.used(i) && A.cnt[i] >= (uint32_t)amin) { const uint64_t s = S.slot_insert(A.key(i)); S.cnt[s] = A.cnt[i]; }
        });
        for (auto& x : th) x.join();
    }
    A.release();                                                                           /* free the big table */
    const double t2 = now();
    /* ---- 3. unitigs ---- */
    Graph<km_t> G{ &S, k, kmask };
    std::vector<uint8_t> seen(scap, 0);
    std::vector<uint64_t> nu_t(n_threads, 0), kc_t(n_threads, 0), dg_t(n_threads, 0), tb_t(n_threads, 0);
    auto spell = [&](int64_t x, int sg, bool cyclic, uint64_t& nu, uint64_t& kcs, uint64_t& dg, uint64_t& tb) {
        std::string s; uint64_t kc = 0; int64_t y = x; int ys = sg;
        for (;;) {
            const km_t u = G.oriented(y, ys);
            if (s.empty()) { for (int b = 0; b < k; ++b) s.push_back("ACGT"[(unsigned)(u >> (2 * (k - 1 - b))) & 3u]); }
            else s.push_back("ACGT"[(unsigned)u & 3u]);
            seen[y] = 1; kc += S.cnt[y];
            const Ref nx = G.succ(y, ys);
            if (nx.node < 0 || (cyclic && nx.node == x)) break;
            y = nx.node; ys = nx.sign;
        }
        ++nu; kcs += kc; dg += unitig_digest(s, kc); tb += s.size();
    };
    {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back([&, t]() {
            for (uint64_t i = scap / n_threads * t, e = t + 1 == n_threads ? scap : scap / n_threads * (t + 1); i < e; ++i) {
                if (!S.used(i)) continue;
                for (int sg = 0; sg < 2; ++sg) {
                    if (G.succ((int64_t)i, !sg).node >= 0) continue;          /* (i, sg) has a predecessor: not a start */
                    /* find the far end; the start with the smaller id emits */
                    int64_t y = (int64_t)i; int ys = sg;
                    for (;;) { const Ref nx = G.succ(y, ys); if (nx.node < 0) break; y = nx.node; ys = nx.sign; }
                    const uint64_t mine = 2 * i + (uint64_t)sg, other = 2 * (uint64_t)y + (uint64_t)(!ys);
                    if (mine <= other) spell((int64_t)i, sg, false, nu_t[t], kc_t[t], dg_t[t], tb_t[t]);
                }
            }
        });
        for (auto& x : th) x.join();
    }
    uint64_t nu = 0, kcs = 0, dg = 0, tb = 0;
    for (int t = 0; t < n_threads; ++t) { nu += nu_t[t]; kcs += kc_t[t]; dg += dg_t[t]; tb += tb_t[t]; }
    for (uint64_t i = 0; i < scap; ++i)                                                    /* isolated cycles: cut at the first k-mer met */
        if (S.used(i) && !seen[i]) spell((int64_t)i, 0, true, nu, kcs, dg, tb);
    const double t3 = now();
    out[0] = n_occ; out[1] = n_distinct; out[2] = n_solid; out[3] = nu; out[4] = kcs; out[5] = dg; out[6] = tb; out[7] = 0;
    secs[0] = t1 - t0; secs[1] = t2 - t1; secs[2] = t3 - t2; secs[3] = t3 - t0;
    return 0;
}

}  // namespace

/* out: [0] occurrences [1] distinct [2] solid [3] unitigs [4] sum KC [5] set digest [6] total unitig bases
 * secs: [0] count [1] solid table [2] unitigs [3] total.  Returns 0, or -1 for unsupported k. */
extern "C" int cpu_mt_run(const char* text, uint64_t n, int k, int amin, int n_threads, uint64_t out[8], double secs[4]) {
    if (k < 3 || k > 63) return -1;
    return k <= 31 ? run_t<uint64_t>(text, n, k, amin, n_threads, out, secs) : run_t<km2_t>(text, n, k, amin, n_threads, out, secs);
}
