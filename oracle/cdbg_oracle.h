/* cdbg_oracle.h -- C API of the CPU oracle (TEST INFRASTRUCTURE ONLY; see cdbg_oracle.c). */
#ifndef CDBG_ORACLE_H
#define CDBG_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct orc_result orc_result;
/* seq: ASCII bases; any byte outside ACGTacgt separates reads.  any k in 3..255. */
orc_result* orc_build(const char* seq, uint64_t n, int k, int abundance_min);
void orc_free(orc_result*);
uint64_t orc_n_occurrences(const orc_result*);
uint64_t orc_n_distinct(const orc_result*);
uint64_t orc_n_solid(const orc_result*);
uint64_t orc_n_unitigs(const orc_result*);
uint64_t orc_total_bases(const orc_result*);
const char* orc_unitig_seq(const orc_result*, uint64_t i);   /* canonical form, sorted */
uint64_t orc_unitig_len(const orc_result*, uint64_t i);
uint64_t orc_unitig_kc(const orc_result*, uint64_t i);
int orc_unitig_circular(const orc_result*, uint64_t i);
const char* orc_solid_kmer(const orc_result*, uint64_t i);   /* sorted canonical k-mers */
uint32_t orc_solid_count(const orc_result*, uint64_t i);
uint64_t orc_digest(const orc_result*);
char* orc_canonical_unitig(const char* s, uint64_t len, int k); /* malloc'ed */
uint64_t orc_synth_genome_len(uint64_t n_reads, uint64_t read_len);
void orc_synth_reads(char* out, uint64_t first_read, uint64_t n_reads, uint64_t total_reads,
                     uint64_t read_len, int cfg);
#ifdef __cplusplus
}
#endif
#endif
