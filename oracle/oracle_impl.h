/* oracle_impl.h -- body of the CPU oracle, instantiated once per k-mer width.
 *
 * TEST INFRASTRUCTURE ONLY (see cdbg_oracle.c header).  Included three times by
 * cdbg_oracle.c with ORC_W = 1 .. 8 (k <= 31, 63, 95, 127, ... 255), the analogue of the
 * reference's KSIZE_LIST spans (/root/reference/README.md:91-99,
 * /root/reference/src/bcalm_1.cpp:95 Integer::apply).
 *
 * Deliberately shares NOTHING with the GPU design: no minimizers, no buckets,
 * no glue.  One global hash table of canonical k-mers, then the unitig
 * definition of /root/reference/bidirected-graphs-in-bcalm2/
 * bidirected-graphs-in-bcalm2.md:64 (node-centric bidirected dBG) and :83-88
 * (unitig conditions) applied literally.
 */

#define ORC_CAT_(a, b) a##b
#define ORC_CAT(a, b) ORC_CAT_(a, b)
#define FN(name) ORC_CAT(ORC_CAT(name, _w), ORC_W)

typedef struct { uint64_t w[ORC_W]; } FN(kmer_t);   /* w[0] = least significant 64 bits */
#define KM FN(kmer_t)

/* ---- multi-word 2-bit arithmetic; base codes A0 C1 G2 T3 so that numeric
 *      order == lexicographic order (scripts/unitigEvaluator.cpp:70-82) ---- */
static inline int FN(km_cmp)(const KM* a, const KM* b) {
    for (int i = ORC_W - 1; i >= 0; --i) {
        if (a->w[i] < b->w[i]) return -1;
        if (a->w[i] > b->w[i]) return 1;
    }
    return 0;
}
static inline int FN(km_eq)(const KM* a, const KM* b) { return FN(km_cmp)(a, b) == 0; }

static inline void FN(km_mask)(KM* a, int k) {
    int bits = 2 * k;
    for (int i = 0; i < ORC_W; ++i) {
        int lo = 64 * i;
        if (bits >= lo + 64) continue;
        if (bits <= lo) a->w[i] = 0;
        else a->w[i] &= (~0ULL) >> (64 - (bits - lo));
    }
}
/* append base c at the right (least significant) end, drop the leftmost */
static inline void FN(km_push_right)(KM* a, int k, unsigned c) {
    for (int i = ORC_W - 1; i > 0; --i) a->w[i] = (a->w[i] << 2) | (a->w[i - 1] >> 62);
    a->w[0] = (a->w[0] << 2) | c;
    FN(km_mask)(a, k);
}
/* prepend base c at the left (most significant) end, drop the rightmost */
static inline void FN(km_push_left)(KM* a, int k, unsigned c) {
    for (int i = 0; i < ORC_W - 1; ++i) a->w[i] = (a->w[i] >> 2) | (a->w[i + 1] << 62);
    a->w[ORC_W - 1] >>= 2;
    int pos = 2 * (k - 1);
    a->w[pos / 64] |= (uint64_t)c << (pos % 64);
}
static inline unsigned FN(km_base)(const KM* a, int k, int i) { /* i-th base from the left */
    int pos = 2 * (k - 1 - i);
    return (unsigned)(a->w[pos / 64] >> (pos % 64)) & 3u;
}
static inline KM FN(km_rc)(const KM* a, int k) {
    KM r; memset(&r, 0, sizeof r);
    for (int i = 0; i < k; ++i) {
        unsigned b = 3u - FN(km_base)(a, k, i);      /* complement */
        int pos = 2 * i;                                 /* base i (from left) lands i-th from right */
        r.w[pos / 64] |= (uint64_t)b << (pos % 64);
    }
    return r;
}
static inline uint64_t FN(km_hash)(const KM* a) {
    uint64_t h = 0x9E3779B97F4A7C15ULL;
    for (int i = 0; i < ORC_W; ++i) {
        uint64_t x = a->w[i] + h;
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
        x ^= x >> 27; x *= 0x94D049BB133111EBULL;
        x ^= x >> 31; h = x;
    }
    return h;
}

/* ---- open-addressing table: canonical k-mer -> count ---- */
typedef struct {
    KM* keys; uint32_t* cnt; uint8_t* used;
    uint64_t cap, n;
} FN(tab_t);
#define TAB FN(tab_t)

static void FN(tab_init)(TAB* t, uint64_t cap) {
    t->cap = cap; t->n = 0;
    t->keys = (KM*)malloc(cap * sizeof(KM));
    t->cnt = (uint32_t*)calloc(cap, sizeof(uint32_t));
    t->used = (uint8_t*)calloc(cap, 1);
    if (!t->keys || !t->cnt || !t->used) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
}
static void FN(tab_free)(TAB* t) { free(t->keys); free(t->cnt); free(t->used); memset(t, 0, sizeof *t); }
static int64_t FN(tab_find)(const TAB* t, const KM* key) {
    uint64_t i = FN(km_hash)(key) & (t->cap - 1);
    while (t->used[i]) {
        if (FN(km_eq)(&t->keys[i], key)) return (int64_t)i;
        i = (i + 1) & (t->cap - 1);
    }
    return -1;
}
static uint64_t FN(tab_slot)(TAB* t, const KM* key) {      /* find-or-insert, no growth */
    uint64_t i = FN(km_hash)(key) & (t->cap - 1);
    while (t->used[i]) {
        if (FN(km_eq)(&t->keys[i], key)) return i;
        i = (i + 1) & (t->cap - 1);
    }
    t->used[i] = 1; t->keys[i] = *key; t->cnt[i] = 0; t->n++;
    return i;
}
static void FN(tab_grow)(TAB* t) {
    TAB nt; FN(tab_init)(&nt, t->cap * 2);
    for (uint64_t i = 0; i < t->cap; ++i) if (t->used[i]) {
        uint64_t s = FN(tab_slot)(&nt, &t->keys[i]);
        nt.cnt[s] = t->cnt[i];
    }
    FN(tab_free)(t); *t = nt;
}
static inline void FN(tab_add)(TAB* t, const KM* key, uint32_t by) {
    if ((t->n + 1) * 10 > t->cap * 7) FN(tab_grow)(t);
    uint64_t s = FN(tab_slot)(t, key);
    /* abundances saturate: exact below ORC_COUNT_SAT, 2^31 - 1 from there on (the 31-bit ceiling the product documents in
     * include/cdbg.h; gatb-core's own `-abundance-max` default is 2147483647 [UPSTREAM-RECALL]) */
    uint64_t c = (uint64_t)t->cnt[s] + by;
    t->cnt[s] = c >= ORC_COUNT_SAT ? ORC_COUNT_MAX : (uint32_t)c;
}

/* ---- stage 1: count canonical k-mers.  Any byte that is not ACGT/acgt breaks
 *      the sequence (read separators, 'N': scripts/unitigEvaluator.cpp:130-131
 *      skips k-mers containing N; README.md:23-25 abundance filter) ---- */
static void FN(count_kmers)(TAB* t, const char* seq, uint64_t n, int k) {
    KM fw, rc; memset(&fw, 0, sizeof fw); memset(&rc, 0, sizeof rc);
    int run = 0;
    for (uint64_t i = 0; i < n; ++i) {
        int c = orc_code((unsigned char)seq[i]);
        if (c < 0) { run = 0; continue; }
        FN(km_push_right)(&fw, k, (unsigned)c);
        FN(km_push_left)(&rc, k, 3u - (unsigned)c);
        if (++run >= k) {
            const KM* can = FN(km_cmp)(&fw, &rc) <= 0 ? &fw : &rc;
            FN(tab_add)(t, can, 1);
        }
    }
}

/* ---- stage 2+3 (spec form): maximal unitigs of the solid sub-graph ---- */
typedef struct { int64_t node; int sign; } FN(ref_t);   /* sign: 0 = '+', 1 = '-' ; node<0 = none */
#define REF FN(ref_t)

/* out-neighbours of (x, sign): bidirected-graphs-in-bcalm2.md:39-46 overlap table.
 * Even k: a k-mer can be its own reverse complement.  Such a node y has label == rc(label), so the overlap
 * "suffix of x = prefix of y.label" and the overlap "suffix of x = prefix of rc(y.label)" both hold: the rows
 * (s,+) and (s,-) of the table are two distinct edges (x,y,s,+) and (x,y,s,-) (edges are 5-tuples, .md:7).  Both
 * are listed (at most one of the four extensions can be a palindrome, hence out[5]); a node with a palindromic
 * neighbour therefore never has a unique out-edge towards it and a palindromic k-mer is always a unitig of its own. */
static int FN(out_edges)(const TAB* s, int k, int64_t x, int sign, REF out[5]) {
    KM u = s->keys[x];
    if (sign) u = FN(km_rc)(&u, k);
    int n = 0;
    for (unsigned c = 0; c < 4; ++c) {
        KM v = u; FN(km_push_right)(&v, k, c);
        KM r = FN(km_rc)(&v, k);
        int cmp = FN(km_cmp)(&v, &r);
        int vs = cmp <= 0 ? 0 : 1;                       /* label is the canonical strand */
        int64_t y = FN(tab_find)(s, vs ? &r : &v);
        if (y >= 0) {
            out[n].node = y; out[n].sign = vs; ++n;
            if (cmp == 0) { out[n].node = y; out[n].sign = 1; ++n; }
        }
    }
    return n;
}
/* the unique compactable successor of (x,sign), or none.  Edge (x,s)->(y,t) may be
 * merged iff it is the only out-edge of (x,s), the only in-edge of (y,t)
 * [in-edges of (y,t) are the mirrors of the out-edges of (y,!t): .md:18-24] and
 * x != y (a unitig is a path: ".md:83 does not repeat vertices" -- this rules out
 * self-loops and self-mirror hairpins, .md:30). */
static REF FN(succ)(const TAB* s, int k, int64_t x, int sign) {
    REF none = { -1, 0 }, o[5], b[5];
    if (FN(out_edges)(s, k, x, sign, o) != 1) return none;
    if (o[0].node == x) return none;
    if (FN(out_edges)(s, k, o[0].node, !o[0].sign, b) != 1) return none;
    return o[0];
}

typedef struct { char* seq; uint64_t len, kc, nk; int circular; } FN(utg_t);

static int FN(utg_cmp)(const void* a, const void* b) {
    const FN(utg_t)* x = (const FN(utg_t)*)a; const FN(utg_t)* y = (const FN(utg_t)*)b;
    int c = strcmp(x->seq, y->seq);
    if (c) return c;
    return x->kc < y->kc ? -1 : x->kc > y->kc;
}

static const TAB* FN(g_sort_tab);
static int FN(order_cmp)(const void* a, const void* b) {
    uint64_t i = *(const uint64_t*)a, j = *(const uint64_t*)b;
    return FN(km_cmp)(&FN(g_sort_tab)->keys[i], &FN(g_sort_tab)->keys[j]);
}

static orc_result* FN(build)(const char* seq, uint64_t n, int k, int amin) {
    orc_result* R = (orc_result*)calloc(1, sizeof *R);
    R->k = k; R->W = ORC_W;
    TAB all; FN(tab_init)(&all, 1 << 12);
    FN(count_kmers)(&all, seq, n, k);
    R->n_distinct = all.n;
    for (uint64_t i = 0; i < all.cap; ++i) if (all.used[i]) R->n_occ += all.cnt[i];

    /* solid sub-table */
    uint64_t ns = 0;
    for (uint64_t i = 0; i < all.cap; ++i) if (all.used[i] && all.cnt[i] >= (uint32_t)amin) ++ns;
    uint64_t cap = 16; while (cap * 7 < ns * 10 + 16) cap <<= 1;
    TAB sol; FN(tab_init)(&sol, cap);
    for (uint64_t i = 0; i < all.cap; ++i) if (all.used[i] && all.cnt[i] >= (uint32_t)amin) {
        uint64_t s = FN(tab_slot)(&sol, &all.keys[i]); sol.cnt[s] = all.cnt[i];
    }
    FN(tab_free)(&all);
    R->n_solid = sol.n;

    /* solid k-mer dump (ASCII, sorted) for stage-1 parity */
    R->solid_kmers = (char*)malloc(ns * (uint64_t)(k + 1) + 1);
    R->solid_counts = (uint32_t*)malloc((ns + 1) * sizeof(uint32_t));
    {
        uint64_t* order = (uint64_t*)malloc((ns + 1) * sizeof(uint64_t)); uint64_t j = 0;
        for (uint64_t i = 0; i < sol.cap; ++i) if (sol.used[i]) order[j++] = i;
        /* simple indirect sort by key */
        FN(g_sort_tab) = &sol;
        qsort(order, ns, sizeof(uint64_t), FN(order_cmp));
        for (j = 0; j < ns; ++j) {
            for (int b = 0; b < k; ++b) R->solid_kmers[j * (k + 1) + b] = "ACGT"[FN(km_base)(&sol.keys[order[j]], k, b)];
            R->solid_kmers[j * (k + 1) + k] = 0;
            R->solid_counts[j] = sol.cnt[order[j]];
        }
        free(order);
    }

    /* unitigs */
    uint8_t* seen = (uint8_t*)calloc(sol.cap, 1);
    uint64_t ucap = 1024, nu = 0;
    FN(utg_t)* U = (FN(utg_t)*)malloc(ucap * sizeof *U);
    for (uint64_t i0 = 0; i0 < sol.cap; ++i0) {
        if (!sol.used[i0] || seen[i0]) continue;
        /* walk backwards from (i0,+) to the start of its maximal unitig */
        int64_t x = (int64_t)i0; int sg = 0; int circular = 0;
        for (;;) {
            REF p = FN(succ)(&sol, k, x, !sg);          /* predecessor of (x,sg) = mirror of succ(x,!sg) */
            if (p.node < 0) break;
            x = p.node; sg = !p.sign;
            if (x == (int64_t)i0) { circular = 1; break; } /* isolated cycle: cut at i0 */
        }
        /* walk forward, spelling (.md "spelling rule") */
        uint64_t cap_s = 256, len = 0, kc = 0, nk = 0;
        char* s = (char*)malloc(cap_s);
        int64_t y = x; int ys = sg;
        for (;;) {
            KM u = sol.keys[y]; if (ys) u = FN(km_rc)(&u, k);
            if (len + (uint64_t)k + 2 > cap_s) { cap_s = cap_s * 2 + k; s = (char*)realloc(s, cap_s); }
            if (nk == 0) { for (int b = 0; b < k; ++b) s[len++] = "ACGT"[FN(km_base)(&u, k, b)]; }
            else s[len++] = "ACGT"[FN(km_base)(&u, k, k - 1)];
            seen[y] = 1; kc += sol.cnt[y]; ++nk;
            REF nx = FN(succ)(&sol, k, y, ys);
            if (nx.node < 0 || nx.node == x) break;      /* end of path, or closed the cycle */
            y = nx.node; ys = nx.sign;
        }
        s[len] = 0;
        if (nu == ucap) { ucap *= 2; U = (FN(utg_t)*)realloc(U, ucap * sizeof *U); }
        U[nu].seq = s; U[nu].len = len; U[nu].kc = kc; U[nu].nk = nk; U[nu].circular = circular; ++nu;
    }
    free(seen);
    /* canonical comparison form (README.md:84-87: orientation is not stable) */
    for (uint64_t i = 0; i < nu; ++i) {
        char* c = orc_canonical_unitig(U[i].seq, U[i].len, k);
        free(U[i].seq); U[i].seq = c;
    }
    qsort(U, nu, sizeof *U, FN(utg_cmp));
    R->n_unitigs = nu;
    R->utg_seq = (char**)malloc((nu + 1) * sizeof(char*));
    R->utg_len = (uint64_t*)malloc((nu + 1) * sizeof(uint64_t));
    R->utg_kc = (uint64_t*)malloc((nu + 1) * sizeof(uint64_t));
    R->utg_circ = (int*)malloc((nu + 1) * sizeof(int));
    for (uint64_t i = 0; i < nu; ++i) {
        R->utg_seq[i] = U[i].seq; R->utg_len[i] = U[i].len; R->utg_kc[i] = U[i].kc; R->utg_circ[i] = U[i].circular;
        R->total_bases += U[i].len;
    }
    free(U);
    FN(tab_free)(&sol);
    return R;
}

#undef REF
#undef TAB
#undef KM
#undef FN
#undef ORC_CAT
#undef ORC_CAT_
