/* cdbg.h -- C ABI of libcdbg.so: MI355X-native compacted de Bruijn graph construction.
 *
 * DROP-IN BOUNDARY.  The reference has no FFI: its whole hot path is ONE statically
 * linked C++ call,
 *     GraphUnitigsTemplate<span>::create(IProperties*, false)
 *         /root/reference/src/bcalm_1.cpp:57   (span chosen by Integer::apply, :95)
 * fed by the option parser borrowed at /root/reference/src/bcalm_1.cpp:31 and the three
 * properties the wrapper itself reads: STR_URI_INPUT (:55), STR_URI_OUTPUT (:69-72),
 * STR_KMER_SIZE (:92).  This header is what a binding for that call site binds instead
 * (INTEGRATION.md shows the replacement of bcalm_1.cpp's Functor body and the ctypes
 * stub).  Plain C types only, opaque context, caller-owned host buffers, no C++
 * exception crosses the boundary (the reference maps gatb Exceptions to a message +
 * exit 1 at /root/reference/src/main.cpp:39-48; here: negative return code +
 * cdbg_last_error()).
 *
 * Stage entry points mirror what create() runs internally:
 *   cdbg_count    a4-a6  read scan -> minimizer partitions -> solid (k-mer, count)
 *   cdbg_compact  a7-a8  per-bucket compaction in LDS -> pieces + glue records
 *   cdbg_glue     a9     hash-join on junction (k-1)-mers + list ranking -> unitigs
 * All functions return 0 on success and a negative code on error.
 */
#ifndef CDBG_H
#define CDBG_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cdbg_ctx cdbg_ctx;

typedef struct cdbg_params {
    int k;                    /* -kmer-size (README.md:17-19,99): any k in 3..255, even or odd (the reference's KSIZE_LIST "32 64 96 128" and the
                                 larger spans it takes as a build option, README.md:91-99; here: CDBG_MAX_W words of 32 bases, default 8) */
    int abundance_min;        /* -abundance-min (README.md:21-25): keep k-mers seen >= this many times */
    int minimizer_size;       /* -minimizer-size (example/circular_unitigs_unittests/CMD:4); 0 = auto */
    int log2_partitions;      /* minimizer partitions = 1 << this; -1 = auto from the input volume */
    int device_id;            /* HIP device ordinal */
    int world_size;           /* GPUs sharing the minimizer space (power of two); 1 = single GPU */
    int rank;                 /* this context owns partitions p with p % world_size == rank */
    int all_abundance_counts; /* -all-abundance-counts (README.md:74-80): keep the abundance of every k-mer of every unitig */
    int emit_replicated;      /* world_size > 1: 0 = every rank emits the unitigs whose head piece it owns (the union over the
                                 ranks is the graph); 1 = every rank emits the complete set (the CLI's rank 0 writes one file) */
    int reads_replicated;     /* world_size > 1: 0 = every rank holds a SHARD of the reads and the super-k-mer records travel to
                                 the partition owners; 1 = every rank holds ALL the reads and scans them for
                                 its own partitions only -- no record exchange (X0: the better choice on 2-4 GPUs, where one
                                 xGMI link would have to carry a large share of the records; free with cdbg_generate_reads) */
} cdbg_params;

typedef struct cdbg_stats_t {
    uint64_t input_bytes;         /* bytes scanned (bases + separators) */
    uint64_t n_records;           /* super-k-mer records written */
    uint64_t n_member_kmers;      /* k-mer occurrences stored in records (home + traveller copies) */
    uint64_t n_occurrences;       /* k-mer occurrences (home only) */
    uint64_t n_distinct;          /* distinct canonical k-mers (the metric's numerator) */
    uint64_t n_solid;             /* solid k-mers (count >= abundance_min) */
    uint64_t n_solid_travellers;  /* solid traveller copies held next to foreign junctions */
    uint64_t n_pieces;            /* compacted pieces before glue */
    uint64_t n_glue_open_ends;    /* open piece ends posted to the glue table */
    uint64_t n_glue_joined;       /* junctions joined by glue */
    uint64_t n_unitigs;
    uint64_t unitig_bases;
    uint64_t n_big_partitions;    /* partitions that fell back to HBM tables (count + compact) */
    uint64_t n_cycles;            /* circular unitigs cut open (in-bucket + across buckets) */
    int minimizer_size, log2_partitions, kmer_words;
    float ms_scan_hist, ms_scan_emit, ms_count, ms_compact, ms_glue, ms_total;
    float ms_exchange;            /* multi-GPU: time inside the record and glue exchanges (transport + merge kernels) */
    uint64_t n_launch_scan, n_launch_count, n_launch_compact;   /* workgroups launched */
    uint64_t n_multipass_partitions; /* partitions whose distinct k-mers did not fit one LDS pass (multi-pass kernel) */
    uint64_t n_tiles_overlapped;     /* scan tiles processed while the input was still arriving (cdbg_expect_input) */
    uint64_t n_split_buckets;        /* buckets that no LDS tier of the compaction could take and that were split by sub-minimizer (k_split.h) */
    uint64_t n_glue_rounds;          /* multi-GPU sharded glue: query / reply rounds of the distributed list ranking (0: single GPU, or
                                        the replicated exchange -- emit_replicated, or closed chains across ranks) */
    uint64_t n_walked_unitigs;       /* unitigs written by walks from the chain heads (k_walk.h: one GPU, no chain beyond the step limit, no
                                        closed chain); 0: the list-ranking path of k_glue.h glued this run */
    /* ABI 7 -- deferred record placement: the partition space is cut into count_slices ranges; the scan places the records
     * of the first itself and appends the others to streams, which k_place scatters on a second HIP stream WHILE the count stage counts the
     * slice before.  ms_scan_emit is then the scan kernel alone, ms_count the wall from the scan's end to the count stage's end (it contains
     * the waits for the placement), ms_place the busy span of the placement stream (overlapped with ms_count, not a summand of the step). */
    uint64_t n_deferred_records;     /* records that went through the streams (0: the scan placed every record itself) */
    float ms_place;
    int count_slices;                /* 1: no deferral */
} cdbg_stats_t;

/* ABI version: bumped whenever a struct of this header changes.  From version 5 on cdbg_stats_t only ever GROWS AT ITS END;
 * a binding checks cdbg_abi_version() against the header it was written for and sizeof(cdbg_stats_t) against
 * cdbg_stats_sizeof() when it loads the library (bcalm_amd/api.py does), instead of reading fields at stale offsets. */
#define CDBG_ABI_VERSION 7
int cdbg_abi_version(void);
uint64_t cdbg_stats_sizeof(void);

/* error codes */
#define CDBG_OK 0
#define CDBG_E_PARAM (-1)     /* bad parameter (k out of range, ...) */
#define CDBG_E_NODEVICE (-2)  /* no HIP device / HIP call failed: there is NO CPU fallback */
#define CDBG_E_NOMEM (-3)
#define CDBG_E_STATE (-4)     /* stages called out of order */
#define CDBG_E_INTERNAL (-5)

int  cdbg_create(const cdbg_params* params, cdbg_ctx** out);
void cdbg_destroy(cdbg_ctx* ctx);
/* cdbg_destroy hands the context's device buffers of 1 MB and more (the super-k-mer record region: 75 GB at config 3) to a per-device
 * pool of the process, and the next context of the same shape runs on the very same pages: the read scan is sensitive to the physical
 * placement of its region and counters, and every free + re-allocation made it slower.  At most 55 % of the
 * card's memory is held per device (nothing with CDBG_NO_POOL set in the environment); cdbg_release_cached returns everything to the
 * driver -- call it before another library of the same process (torch, RCCL) needs the HBM. */
int cdbg_release_cached(void);
const char* cdbg_last_error(void);

/* Input.  Host ASCII, caller-owned, borrowed for the call; may be called repeatedly.
 * push_reads: read i is bases[offsets[i] .. offsets[i+1]).  push_text: sequences already
 * separated by any byte outside ACGTacgt ('\n', 'N', ...; such bytes break k-mers, like
 * /root/reference/scripts/unitigEvaluator.cpp:130-131). */
int cdbg_push_reads(cdbg_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n_reads);
int cdbg_push_text(cdbg_ctx* ctx, const char* text, uint64_t nbytes);
/* Zero-copy input for multi-threaded parsers (the `bcalm` CLI: N threads on record-aligned slices of a memory-mapped FASTA /
 * FASTQ, one inflate thread per .gz of a file list; README.md:45-50).  cdbg_stage_acquire hands out one of the context's
 * PINNED staging buffers; the caller writes sequence text into it (sequences separated by any byte outside ACGTacgt) and
 * cdbg_stage_commit appends the first `nbytes` of it to the device text with an asynchronous copy (a '\n' is added when the
 * text does not end with a separator: a commit always ends a sequence -- a sequence longer than a buffer is continued in
 * the next one by repeating its last k-1 bases).  nbytes = 0 hands the buffer back unused.  Both calls are thread-safe
 * and may be mixed with the push calls; commits append in the order in which they arrive (the result does not depend on
 * it).  With cdbg_expect_input the scan of the complete tiles is launched from the commits as from the pushes. */
int cdbg_stage_acquire(cdbg_ctx* ctx, char** buf, uint64_t* capacity);
int cdbg_stage_commit(cdbg_ctx* ctx, char* buf, uint64_t nbytes);
/* Optional, before the first push: the (approximate) number of text bytes that will be pushed, e.g. from the file size.
 * The device text is then allocated once, and the read scan starts on the part of the input that has already landed
 * while the caller is still parsing and pushing the rest (single GPU; README.md:45-50 inputs through the CLI). */
int cdbg_expect_input(cdbg_ctx* ctx, uint64_t text_bytes);
/* Synthetic reads generated directly in HBM (BASELINE.md section 2 generator): reads
 * [first_read, first_read + n_reads) of a set of total_reads reads of read_len bases.
 * cfg 0 .. 255: uniform random genome at 30x coverage, 1 % substitutions (the seed).  cfg | 0x100: the same reads over a
 * HOSTILE genome -- two-letter low-complexity blocks (1 in 20), up to 1000 exact copies of one 5 kbp repeat, 50 homopolymer
 * runs of 600 bases, a third of the reads inside 1/40 of the genome (~20x coverage skew).  cfg < 0: every base 'A' (test
 * hook for the abundance ceiling).  oracle/cdbg_oracle.c:orc_synth_reads writes the same bytes. */
int cdbg_generate_reads(cdbg_ctx* ctx, uint64_t first_read, uint64_t n_reads, uint64_t total_reads,
                        uint64_t read_len, int cfg);
/* copy (part of) the resident read text back to the host (tests, FASTA dumps) */
int cdbg_read_text(cdbg_ctx* ctx, uint64_t first_byte, uint64_t nbytes, char* out);

/* The three hot stages, in order.  cdbg_run = count + compact + glue. */
int cdbg_count(cdbg_ctx* ctx);
int cdbg_compact(cdbg_ctx* ctx);
int cdbg_glue(cdbg_ctx* ctx);
int cdbg_run(cdbg_ctx* ctx);
/* forget the results but keep the resident reads and all device buffers, so the same job can
 * be run again (benchmark steps) without allocator traffic */
int cdbg_reset(cdbg_ctx* ctx);

/* ---- Multi-GPU: one context per GPU (one process per GPU, or one host thread per GPU in one process) ----
 * The reference has nothing here (GraphUnitigsTemplate<span>::create is one shared-memory call, src/bcalm_1.cpp:57);
 * the design is SURVEY.md section 8(e).  Rank r owns the minimizer partitions p with p % world_size == r.  cdbg_run /
 * cdbg_count / cdbg_compact / cdbg_glue of a context with world_size > 1 perform, through the context's transport:
 *   count   reads_replicated = 0 (X1): scan the own SHARD of the reads into super-k-mer records of ALL partitions (one capped
 *           pass, regions squeezed into the owner-major layout), all-to-all-v the records to the partition owners, count the
 *           own partitions.  reads_replicated = 1 (X0): scan ALL reads for the own partitions only; nothing travels.
 *   compact the own buckets (no exchange: the owner of a junction holds every solid k-mer adjacent to it)
 *   glue    emit_replicated = 0: sharded by owner (bcalm_amd/csrc/k_dglue.h) -- every junction record travels once to the rank
 *           its key hashes to, every joined pair once to the owner of the end, list ranking runs on the owners of the pieces
 *           with one query / reply exchange per round, every piece travels once to the owner of its unitig's head; each rank
 *           ends with the unitigs it owns (their union is the graph).  emit_replicated = 1, or closed chains that cross ranks:
 *           the replicated exchange -- pieces + junction log all-gathered, join sharded by key hash with one MAX all-reduce,
 *           chains ranked on every rank; every rank ends with the complete set.
 * A rank that received no reads still takes part in every collective; a rank-local failure is agreed on before the next
 * collective stage, so that every rank returns an error together instead of leaving its peers inside the transport.
 * The transport moves bytes between DEVICE buffers of the ranks.  Built in: RCCL (grouped ncclSend/ncclRecv all-to-all-v,
 * ncclAllGather, ncclAllReduce over xGMI), bound at run time from librccl.so.1:
 *   cdbg_comm_unique_id   rank 0 creates the 128-byte id; the caller distributes it (MPI, torch.distributed, a file)
 *   cdbg_comm_init_rccl   every rank, after cdbg_create on its device
 * or caller-supplied functions (cdbg_set_transport; the CPU tests use gloo and an in-process loop-back).
 * All four functions are collective: every rank calls them in the same order; they return 0 on success.
 *   all_gather_u64  host buffers: n words from every rank, rank order
 *   all_to_all_v    device buffers: send_cnt[r] bytes at send_off[r] go to rank r; recv_cnt[s] bytes from rank s land at recv_off[s]
 *   all_gather_v    device buffers: nbytes of every rank (recv_cnt[s], known from an all_gather_u64) at recv_off[s]   (replicated exchange only)
 *   all_reduce_max_i32  in place on a device buffer of n int32                                                        (replicated exchange only) */
typedef struct cdbg_transport {
    void* user;
    int (*all_gather_u64)(void* user, const uint64_t* send, uint64_t* recv, int n);
    int (*all_to_all_v)(void* user, const void* send_dev, const uint64_t* send_off, const uint64_t* send_cnt,
                        void* recv_dev, const uint64_t* recv_off, const uint64_t* recv_cnt);
    int (*all_gather_v)(void* user, const void* send_dev, uint64_t nbytes, void* recv_dev, const uint64_t* recv_off, const uint64_t* recv_cnt);
    int (*all_reduce_max_i32)(void* user, void* dev, uint64_t n);
} cdbg_transport;
int cdbg_set_transport(cdbg_ctx* ctx, const cdbg_transport* t);
int cdbg_comm_unique_id(void* out_128_bytes);
int cdbg_comm_init_rccl(cdbg_ctx* ctx, const void* unique_id_128_bytes);
/* bytes this rank sent + received through the transport since cdbg_reset (bench.py reports them) */
int cdbg_comm_bytes(cdbg_ctx* ctx, uint64_t* out);

/* Results.  Solid k-mers: kmers has (k+1)-byte stride, NUL-terminated ASCII, canonical strand. */
int cdbg_num_solid(cdbg_ctx* ctx, uint64_t* n);
int cdbg_fetch_solid(cdbg_ctx* ctx, char* kmers, uint32_t* counts, uint64_t capacity, uint64_t* n_written);
/* Unitigs [first, first+n): sequences concatenated into seq_buf (ASCII, no terminators),
 * seq_off[n+1] offsets into seq_buf, kc[n] summed abundances (KC; LN = length,
 * km = KC / (LN-k+1): /root/reference/README.md:62-70).  Orientation and order are
 * unspecified, as in the reference (README.md:84-87). */
int cdbg_num_unitigs(cdbg_ctx* ctx, uint64_t* n, uint64_t* total_bases);
int cdbg_fetch_unitigs(cdbg_ctx* ctx, uint64_t first, uint64_t n, char* seq_buf, uint64_t* seq_off, uint64_t* kc);
/* The same set at 2 bits per base, the form the glue stage leaves resident in HBM next to the ASCII arena (SURVEY.md 8d:
 * "canonical unitigs + KC resident in HBM (2-bit packed)").  All n = cdbg_num_unitigs unitigs at once: `packed` receives
 * ceil(total_bases / 4) bytes of ONE arena; unitig i is bases [base_off[i], base_off[i] + len[i]) of it; base j of the
 * arena = bits [2 (j & 3), 2 (j & 3) + 2) of byte j >> 2, codes A0 C1 G2 T3. */
int cdbg_fetch_unitigs_packed(cdbg_ctx* ctx, uint8_t* packed, uint64_t packed_capacity, uint64_t* base_off, uint32_t* len, uint64_t* kc);
/* per-k-mer abundances of unitigs [first, first+n) (contexts created with all_abundance_counts = 1):
 * ab_off[n+1] offsets into ab; unitig i has LN-k+1 values in the orientation of its sequence
 * (the `ab:Z:` vector of /root/reference/README.md:74-80) */
int cdbg_fetch_unitig_abundances(cdbg_ctx* ctx, uint64_t first, uint64_t n, uint32_t* ab, uint64_t* ab_off);
/* Abundances are 31-bit and saturate: exact below 2^31 - 4096, reported as 2147483647 from there on (KC sums the reported
 * values in 64 bits).  README.md:62-72 (KC = sum of abundances); gatb-core's `-abundance-max` default is the same number. */
int cdbg_stats(cdbg_ctx* ctx, cdbg_stats_t* out);
/* Digests of the resident result, computed on the device (bench.py checks them at sizes no oracle follows):
 * out[0] = sum of KC over the unitigs, out[1] = sum of the solid k-mers' counts (must equal out[0]),
 * out[2] = order- and orientation-independent digest of the set {(unitig, KC)} (formula: k_links.h k_digest_unitigs;
 * pinned against the oracle's unitigs in tests/), out[3] = sum of (LN - k + 1) (must equal n_solid). */
int cdbg_digest(cdbg_ctx* ctx, uint64_t out[4]);
/* The unitig definition (/root/reference/bidirected-graphs-in-bcalm2/bidirected-graphs-in-bcalm2.md:64,83-92) checked on
 * the resident result at any size, by code that shares nothing with the compaction or the glue (bcalm_amd/csrc/k_verify.h):
 *   out[0..2]  k-mer positions of all unitigs, and two independent commutative 64-bit sums over their canonical k-mers
 *   out[3..5]  the same three numbers over the solid k-mers as the count stage left them
 *              -- equal triples <=> the unitigs spell every solid k-mer exactly once and nothing else
 *   out[6]     ends e, f of two different unitigs that are each other's ONLY link (counted from both ends): must be 0 --
 *              every unitig is maximal; builds the links (cdbg_link) when they are not there yet
 *   out[7]     unitigs whose two ends are each other's only link (closed chains, cut open once: legitimate)
 * Several ranks: the sums are additive over the ranks; a rank that holds a share of the unitigs (emit_replicated = 0) has
 * no link table and reports out[6] = out[7] = UINT64_MAX. */
int cdbg_verify(cdbg_ctx* ctx, uint64_t out[8]);
/* Edge conservation: the inner-junction half of the unitig definition (bidirected-graphs-in-bcalm2.md:85 -- "for every
 * 0 < i < n the only edges incident on v_i are e_{i-1}, e_i and their mirrors") at any size.  cdbg_verify's maximality pass
 * sees unitig ENDS only; a unitig that runs THROUGH a branching node passes it.  Here the edges of the solid k-mer graph are
 * counted from the count stage's keys alone (one global table of canonical (k-1)-mers; bcalm_amd/csrc/k_verify.h):
 *   out[0]  D = sum over the ends of all solid k-mers of the k-mer ends they see across their junction
 *   out[1]  L = the same sum over the unitig ENDS (= number of links; builds them when they are not there yet)
 *   out[2]  2 * sum over the unitigs of (LN - k): every inner adjacency of a unitig, seen from both sides
 *   out[3]  distinct junctions of the solid graph
 * out[0] == out[1] + out[2]  <=>  every inner junction of every unitig is 1-in / 1-out (the shortfall of a junction that
 * a unitig runs through wrongly is strictly positive: nothing cancels).  Several ranks: every rank counts a share of the k-mers, their junctions belong to all: UINT64_MAX. */
int cdbg_verify_edges(cdbg_ctx* ctx, uint64_t out[4]);
/* The checks of cdbg_verify (out[0..7]) and cdbg_verify_edges (out[8..11]) for a unitig set SUPPLIED BY THE CALLER -- ASCII
 * bases, offsets[n + 1] -- against the solid k-mers resident after cdbg_count: a FASTA written by any program can be judged
 * against this library's counted k-mer set (and the tests plant an over-compacted unitig to see the edge check fail). */
int cdbg_verify_unitigs(cdbg_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n_unitigs, uint64_t out[12]);

/* Edges between unitigs (the `L:<+/->:<id>:<+/->` tokens of /root/reference/README.md:62-72; GFA `L`
 * lines of scripts/convertToGFA.py:103-112).  After cdbg_glue: cdbg_link builds them on the GPU.
 * cdbg_fetch_links: end_off[2U+1] offsets per unitig END (end 2u = leaving u through the reverse
 * complement of its first k-mer, from-sign '-'; end 2u+1 = leaving through its last k-mer, from-sign
 * '+'); link_to[i] = target end 2v+side: side 0 = enters v at its first k-mer (to-sign '+'),
 * side 1 = enters the reverse complement of its last k-mer (to-sign '-'). */
int cdbg_link(cdbg_ctx* ctx);
int cdbg_num_links(cdbg_ctx* ctx, uint64_t* n);
int cdbg_fetch_links(cdbg_ctx* ctx, uint64_t* end_off, uint32_t* link_to);
/* Several ranks, every rank holding a share of the unitigs (emit_replicated = 0): cdbg_link is COLLECTIVE (all ranks call it).  Unitig
 * ids are then job-wide, numbered rank after rank: this rank's unitig i (the order of cdbg_fetch_unitigs) is first_id + i, end_off
 * covers this rank's ends and link_to holds job-wide end ids (2 x id + side) -- every rank can write its own share of the output
 * file, `L:` tokens included (the bcalm CLI with -nb-gpus does).  One rank, or emit_replicated = 1: first_id = 0, total = this
 * context's unitigs.  After cdbg_link. */
int cdbg_unitig_id_base(cdbg_ctx* ctx, uint64_t* first_id, uint64_t* total);

/* Environment variables read by the library -- test hooks that force paths an ordinary input does not reach (tests/), not
 * tuning knobs; results are identical with and without them:
 *   CDBG_SCAN_MODE=capped|exact|var  record layout (default: by input size and skew; var = one pass into per-partition regions sized
 *                                 from a quarter-sample, what skewed inputs get)   CDBG_PART_CAP=<n>    capped region size (forces spills)
 *   CDBG_VAR_SCALE=<f>            with var: capacity per sampled record (tiny values force spills)      CDBG_NO_SPLIT=1   overfull buckets through
 *                                 the HBM-table compaction tier instead of the second-level split by sub-minimizer (k_split.h)
 *   CDBG_REPAIR_MAX_PASSES=<n>    LDS pass limit of the spill-repair launch   CDBG_NO_COUNT_TIER2=1     skip the second one-pass count tier
 *   CDBG_GLUE_LOG=1               junction records through the sequential log CDBG_GLUE_TABLE=1         global-table junction join
 *   CDBG_JOIN_LOG_JB=<n>          log2 of the join buckets (0 forces the overflow fallback)
 *   CDBG_FORCE_MULTI=1            run the multi-GPU data path with one rank   CDBG_STAGE_BYTES, CDBG_STREAM_MIN_BYTES, CDBG_STREAM_BATCH_TILES: ingest staging sizes
 *   CDBG_GLUE_REPLICATED=1        the replicated glue exchange for emit_replicated = 0 as well       CDBG_HOST_MARKS=1  wall-clock marks between host-side phases (stderr) */

#ifdef __cplusplus
}
#endif
#endif /* CDBG_H */
