/* cdbg_oracle.c -- CPU oracle for the reads -> unitigs hot path.
 *
 * ============================ TEST INFRASTRUCTURE ============================
 * This file is the CHECKER, not the product.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  Nothing under bcalm_amd/ links,
 * imports or executes it; the product path (libcdbg.so) fails loudly when the
 * HIP device/extension is missing instead of falling back to this code.
 *
 * PARITY UNPINNED.  The reference implementation of this path lives in the
 * un-vendored gatb-core submodule (/root/reference/.gitmodules:1-3, empty
 * directory; /root/reference/src/bcalm_1.cpp:6-8 says so) and its version is not
 * pinned by the tree (no gitlink; CI tracks remote HEAD,
 * /root/reference/.circleci/config.yml:20-22).  The reference cannot be built here
 * (CMakeLists.txt:52 include(GatbCore) fails) and its tests store no expected
 * outputs (test/simple_test.sh:4-9 downloads its inputs and comparer).  This
 * oracle therefore restates the PUBLISHED SPECIFICATION the tree does contain:
 *   - node-centric bidirected de Bruijn graph:  bidirected-graphs-in-bcalm2/
 *     bidirected-graphs-in-bcalm2.md:64
 *   - overlap/sign table and mirrors:            .md:18-30, :39-46
 *   - unitig conditions, maximality, compaction: .md:83-92
 *   - abundance filter "seen (strictly) less than X times ... filtered out":
 *     README.md:23-25
 *   - canonical k-mers, orientation not stable:  README.md:84-87
 *   - k-mers containing N are skipped, canonical = lexicographic min with
 *     A<C<G<T:  scripts/unitigEvaluator.cpp:64-66,70-82,130-131
 *   - FASTA header fields LN/KC/km:              README.md:62-72
 * It is pinned by: an independent pure-Python restatement (oracle/oracle_py.py),
 * the spec-derived anchors of SURVEY.md section 4 (tests/golden/), and the
 * reference's own k-mer-set checker scripts/unitigEvaluator.cpp, compiled in place
 * into oracle/_ref/ (oracle/Makefile target `ref`).
 * ============================================================================
 *
 * Build:  make -C oracle          (liboracle.so + bcalm_oracle CLI)
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "cdbg_oracle.h"

static inline int orc_code(unsigned char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}

static char* orc_revcomp(const char* s, uint64_t n) {
    char* r = (char*)malloc(n + 1);
    for (uint64_t i = 0; i < n; ++i) {
        char c = s[n - 1 - i];
        r[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
    }
    r[n] = 0;
    return r;
}

/* compare two linearisations of a cyclic sequence (period n) of length len */
static int orc_cyc_cmp(const char* a, uint64_t ra, const char* b, uint64_t rb, uint64_t n, uint64_t len) {
    for (uint64_t i = 0; i < len; ++i) {
        char x = a[(ra + i) % n], y = b[(rb + i) % n];
        if (x != y) return x < y ? -1 : 1;
    }
    return 0;
}

/* Canonical comparison form of one unitig (orientation / cut-point normalised):
 *   linear  : min(s, revcomp(s))
 *   if the first k-1 bases equal the last k-1 bases the unitig can be read as a
 *   cycle of n = len-k+1 k-mers (isolated circular unitig,
 *   example/circular_unitigs_unittests/README:1); the reference's cut point is
 *   unknowable, so take the minimum over all 2n linearisations. */
char* orc_canonical_unitig(const char* s, uint64_t len, int k) {
    char* r = orc_revcomp(s, len);
    int cyc = len >= (uint64_t)k && memcmp(s, s + (len - (k - 1)), (size_t)(k - 1)) == 0;
    if (!cyc) {
        if (strcmp(s, r) <= 0) { free(r); char* c = (char*)malloc(len + 1); memcpy(c, s, len + 1); return c; }
        return r;
    }
    uint64_t n = len - (uint64_t)k + 1;
    const char* best = s; uint64_t br = 0;
    for (int strand = 0; strand < 2; ++strand) {
        const char* c = strand ? r : s;
        for (uint64_t rot = 0; rot < n; ++rot)
            if (orc_cyc_cmp(c, rot, best, br, n, len) < 0) { best = c; br = rot; }
    }
    char* out = (char*)malloc(len + 1);
    for (uint64_t i = 0; i < len; ++i) out[i] = best[(br + i) % n];
    out[len] = 0;
    free(r);
    return out;
}

struct orc_result {
    int k, W;
    uint64_t n_occ, n_distinct, n_solid, n_unitigs, total_bases;
    char* solid_kmers; uint32_t* solid_counts;       /* sorted, (k+1)-byte stride */
    char** utg_seq; uint64_t* utg_len; uint64_t* utg_kc; int* utg_circ;
};

#define ORC_COUNT_MAX 0x7FFFFFFFu
#define ORC_COUNT_SAT (ORC_COUNT_MAX - 4095u)
#define ORC_W 1
#include "oracle_impl.h"
#undef ORC_W
#define ORC_W 2
#include "oracle_impl.h"
#undef ORC_W
#define ORC_W 3
#include "oracle_impl.h"
#undef ORC_W
#define ORC_W 4
#include "oracle_impl.h"
#undef ORC_W
/* wider spans (the reference's KSIZE_LIST is open-ended, README.md:91-99): k = 128 .. 255 */
#define ORC_W 5
#include "oracle_impl.h"
#undef ORC_W
#define ORC_W 6
#include "oracle_impl.h"
#undef ORC_W
#define ORC_W 7
#include "oracle_impl.h"
#undef ORC_W
#define ORC_W 8
#include "oracle_impl.h"
#undef ORC_W

orc_result* orc_build(const char* seq, uint64_t n, int k, int abundance_min) {
    /* any k in 3..255, even or odd (README.md:99 "any k value up to the largest one"); the word count follows the
     * reference's span rule k < 32 W (README.md:91-99) */
    if (k < 3 || k > 255 || abundance_min < 1) return NULL;
    switch (k / 32 + 1) {
        case 1: return build_w1(seq, n, k, abundance_min);
        case 2: return build_w2(seq, n, k, abundance_min);
        case 3: return build_w3(seq, n, k, abundance_min);
        case 4: return build_w4(seq, n, k, abundance_min);
        case 5: return build_w5(seq, n, k, abundance_min);
        case 6: return build_w6(seq, n, k, abundance_min);
        case 7: return build_w7(seq, n, k, abundance_min);
        default: return build_w8(seq, n, k, abundance_min);
    }
}
void orc_free(orc_result* r) {
    if (!r) return;
    for (uint64_t i = 0; i < r->n_unitigs; ++i) free(r->utg_seq[i]);
    free(r->utg_seq); free(r->utg_len); free(r->utg_kc); free(r->utg_circ);
    free(r->solid_kmers); free(r->solid_counts); free(r);
}
uint64_t orc_n_occurrences(const orc_result* r) { return r->n_occ; }
uint64_t orc_n_distinct(const orc_result* r) { return r->n_distinct; }
uint64_t orc_n_solid(const orc_result* r) { return r->n_solid; }
uint64_t orc_n_unitigs(const orc_result* r) { return r->n_unitigs; }
uint64_t orc_total_bases(const orc_result* r) { return r->total_bases; }
const char* orc_unitig_seq(const orc_result* r, uint64_t i) { return r->utg_seq[i]; }
uint64_t orc_unitig_len(const orc_result* r, uint64_t i) { return r->utg_len[i]; }
uint64_t orc_unitig_kc(const orc_result* r, uint64_t i) { return r->utg_kc[i]; }
int orc_unitig_circular(const orc_result* r, uint64_t i) { return r->utg_circ[i]; }
const char* orc_solid_kmer(const orc_result* r, uint64_t i) { return r->solid_kmers + i * (uint64_t)(r->k + 1); }
uint32_t orc_solid_count(const orc_result* r, uint64_t i) { return r->solid_counts[i]; }

/* FNV-1a over the canonical, sorted (sequence, KC) records: one number that
 * identifies a whole unitig set (used for full-size parity checks). */
uint64_t orc_digest(const orc_result* r) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (uint64_t i = 0; i < r->n_unitigs; ++i) {
        for (const char* p = r->utg_seq[i]; *p; ++p) { h ^= (unsigned char)*p; h *= 0x100000001b3ULL; }
        for (int b = 0; b < 8; ++b) { h ^= (r->utg_kc[i] >> (8 * b)) & 0xff; h *= 0x100000001b3ULL; }
        h ^= 0x0a; h *= 0x100000001b3ULL;
    }
    return h;
}

/* ---- counter-based synthetic reads ---- */
static inline uint64_t orc_mix(uint64_t x) {          /* splitmix64 finaliser */
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
uint64_t orc_synth_genome_len(uint64_t n_reads, uint64_t read_len) {
    uint64_t g = (n_reads * read_len + 29) / 30;
    return g < read_len ? read_len : g;
}
/* cfg 0 .. 255: the plain generator.  cfg | ORC_GEN_HOSTILE: the same reads over a hostile genome (two-letter blocks,
 * up to 1000 copies of one 5 kbp repeat, 50 homopolymer runs of 600 bases, a third of the reads inside 1/40 of the
 * genome).  cfg < 0: every base 'A'.  Bit for bit what bcalm_amd/csrc/k_scan.h:k_gen_reads writes. */
#define ORC_GEN_HOSTILE 0x100
static unsigned orc_genome_base(uint64_t gp, uint64_t G, uint64_t seed_g, int hostile) {
    const unsigned plain = (unsigned)(orc_mix(seed_g + gp) >> 62);
    if (!hostile) return plain;
    const uint64_t hp_stride = G / 50;
    if (hp_stride >= 2400) {
        const uint64_t off = gp % hp_stride;
        if (off >= hp_stride / 2 && off < hp_stride / 2 + 600) return (unsigned)((gp / hp_stride) & 3u);
    }
    uint64_t nc = G / 50000; if (nc > 1000) nc = 1000;
    if (nc) {
        const uint64_t rs = G / nc, off = gp % rs;
        if (off >= rs / 4 && off < rs / 4 + 5000) return (unsigned)(orc_mix((seed_g ^ 0x5EED0000ULL) + (off - rs / 4)) >> 62);
    }
    const uint64_t bh = orc_mix(seed_g + 0x10C0000000ULL + (gp >> 12));
    if (bh % 20 == 0) {
        const unsigned a = (unsigned)(bh >> 8) & 3u, b2 = (a + 1u + (unsigned)((bh >> 16) % 3)) & 3u;
        return (plain & 2u) ? a : b2;
    }
    return plain;
}
static uint64_t orc_read_start(uint64_t u, uint64_t G, uint64_t L, int hostile) {
    if (hostile && (u >> 32) % 3 == 0) {
        uint64_t hot = G / 40; if (hot < L) hot = L;
        return G / 2 - hot / 2 + u % (hot - L + 1);
    }
    return u % (G - L + 1);
}
/* writes n_reads * (read_len + 1) bytes: each read followed by '\n' */
void orc_synth_reads(char* out, uint64_t first_read, uint64_t n_reads, uint64_t total_reads,
                     uint64_t read_len, int cfg) {
    const uint64_t SEED_G = 0xBCA10000ULL + (uint64_t)cfg, SEED_R = 0xBCA11000ULL + (uint64_t)cfg,
                   SEED_E = 0xBCA12000ULL + (uint64_t)cfg;
    const uint64_t G = orc_synth_genome_len(total_reads, read_len);
    const int hostile = cfg >= 0 && (cfg & ORC_GEN_HOSTILE);
    for (uint64_t i = 0; i < n_reads; ++i) {
        uint64_t r = first_read + i;
        char* dst = out + i * (read_len + 1);
        dst[read_len] = '\n';
        if (cfg < 0) { memset(dst, 'A', read_len); continue; }
        uint64_t start = orc_read_start(orc_mix(SEED_R + 2 * r), G, read_len, hostile);
        int strand = (int)(orc_mix(SEED_R + 2 * r + 1) & 1);
        for (uint64_t j = 0; j < read_len; ++j) {
            /* j-th base of the read as sequenced; on the reverse strand it is the
             * complement of genome base start+L-1-j */
            uint64_t gp = strand ? start + read_len - 1 - j : start + j;
            unsigned b = orc_genome_base(gp, G, SEED_G, hostile);
            if (strand) b = 3u - b;
            uint64_t x = orc_mix(SEED_E + r * read_len + j);
            if (x % 10000 < 100) b = (b + 1 + (unsigned)((x >> 32) % 3)) & 3u;
            dst[j] = "ACGT"[b];
        }
    }
}

#ifdef ORC_MAIN
/* bcalm_oracle: CPU tool with the reference CLI surface (README.md:11-25) for
 * fixtures and the cpu_baseline leg.  Reads FASTA/FASTQ-ish text: every line
 * starting with '>' '@' '+' toggles header/quality handling the simple way. */
static char* slurp_sequences(const char* path, uint64_t* n_out) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(1); }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    char* raw = (char*)malloc((size_t)sz + 1);
    if (fread(raw, 1, (size_t)sz, f) != (size_t)sz) { fprintf(stderr, "short read\n"); exit(1); }
    raw[sz] = 0; fclose(f);
    char* out = (char*)malloc((size_t)sz + 2); uint64_t n = 0;
    int fastq = raw[0] == '@';
    long i = 0; int line_in_rec = 0;
    while (i < sz) {
        long e = i; while (e < sz && raw[e] != '\n') ++e;
        int keep;
        if (fastq) { keep = (line_in_rec == 1); line_in_rec = (line_in_rec + 1) & 3; }
        else keep = raw[i] != '>';
        if (keep) { memcpy(out + n, raw + i, (size_t)(e - i)); n += (uint64_t)(e - i); if (fastq) out[n++] = '\n'; }
        else if (!fastq) out[n++] = '\n';          /* FASTA header terminates the previous sequence */
        i = e + 1;
    }
    out[n++] = '\n';
    free(raw); *n_out = n; return out;
}
int main(int argc, char** argv) {
    const char* in = NULL; const char* outp = NULL; int k = 31, amin = 2; int synth = 0;
    uint64_t sr = 0, sl = 150; int cfg = 3;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-in") && i + 1 < argc) in = argv[++i];
        else if (!strcmp(argv[i], "-out") && i + 1 < argc) outp = argv[++i];
        else if (!strcmp(argv[i], "-kmer-size") && i + 1 < argc) k = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-abundance-min") && i + 1 < argc) amin = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-synthetic") && i + 3 < argc) { synth = 1; sr = strtoull(argv[++i], 0, 10); sl = strtoull(argv[++i], 0, 10); cfg = atoi(argv[++i]); }
    }
    char* seq; uint64_t n;
    if (synth) { n = sr * (sl + 1); seq = (char*)malloc(n); orc_synth_reads(seq, 0, sr, sr, sl, cfg); }
    else { if (!in) { fprintf(stderr, "Specifiy -in\n"); return 1; } seq = slurp_sequences(in, &n); }
    struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    orc_result* r = orc_build(seq, n, k, amin);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (!r) { fprintf(stderr, "bad parameters (k: 3..255)\n"); return 1; }
    double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    fprintf(stderr, "oracle: occ=%llu distinct=%llu solid=%llu unitigs=%llu bases=%llu digest=%016llx  %.3fs  %.3f Mkmers/s\n",
            (unsigned long long)r->n_occ, (unsigned long long)r->n_distinct, (unsigned long long)r->n_solid,
            (unsigned long long)r->n_unitigs, (unsigned long long)r->total_bases,
            (unsigned long long)orc_digest(r), sec, 1e-6 * (double)r->n_distinct / sec);
    FILE* o = outp ? fopen(outp, "w") : stdout;
    for (uint64_t i = 0; i < r->n_unitigs; ++i)
        fprintf(o, ">%llu LN:i:%llu KC:i:%llu km:f:%.1f\n%s\n", (unsigned long long)i,
                (unsigned long long)r->utg_len[i], (unsigned long long)r->utg_kc[i],
                (double)r->utg_kc[i] / (double)(r->utg_len[i] - (uint64_t)k + 1), r->utg_seq[i]);
    if (outp) fclose(o);
    orc_free(r); free(seq);
    return 0;
}
#endif
