/*
 * paf.h -- the per-record C API of the reference (inc/paf.h:52-269) over the MI355X batch engine.
 *
 * Same type names, struct layouts and function signatures as the reference header, so that C code written against
 * `stPaf.a` (tests/paf_unit_test.c:20-47 builds and inspects Paf / Cigar fields directly) compiles against this one and
 * links with lib/libstPaf_hip.so. Every function that parses, transforms or serialises a record calls the GPU through
 * include/paffy_hip.h (host/paf_api.c): a record is handed over as one PAF line, the kernels do the work, the result
 * comes back as arrays (paffy_hip_parse_host) or text. One call = one small batch: this API is for drop-in
 * compatibility, the throughput path is the batch C-ABI the `paffy <cmd>` drivers use.
 *
 * Differences from the reference header:
 *  - no `#include "sonLib.h"` (inc/paf.h:10; sonLib is an external library this build does not carry). The five prototypes
 *    that expose sonLib containers (read_pafs, write_pafs, paf_chain, paf_shatter, get_alignment_count_array) are declared
 *    only when PAFFY_WITH_SONLIB is defined before including this file; sonLib-free equivalents over plain arrays are
 *    always available (read_pafs_array, write_pafs_array, paf_shatter_array, paf_chain_array).
 *    (get_alignment_count_array_in keeps the count arrays in a plain array instead of a stHash.)
 *  - errors end the process the way st_errAbort / assert do in the reference (message on stderr, exit 1 or abort()).
 *  - like the reference's, the transforms validate nothing: paf_check is the call that does (impl/paf.c:427-461).
 */
#ifndef ST_PAF_H_
#define ST_PAF_H_

#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum _cigarOp { /* inc/paf.h:52-58 */
    match = 0,
    query_insert = 1,
    query_delete = 2,
    sequence_match = 3,
    sequence_mismatch = 4
} CigarOp;

typedef struct _cigar_record { /* inc/paf.h:61-64: 8 bytes, length in bits 0-55, op in bits 56-63 */
    int64_t length : 56;
    int64_t op : 8;
} CigarRecord;

typedef struct _cigar Cigar; /* inc/paf.h:67-73 */
struct _cigar {
    CigarRecord *recs;
    int64_t length;
    int64_t start;
    int64_t capacity;
};

static inline int64_t cigar_count(Cigar *c) { return c ? c->length : 0; }
static inline CigarRecord *cigar_get(Cigar *c, int64_t i) { return &c->recs[c->start + i]; }

Cigar *cigar_parse(char *cigar_string); /* inc/paf.h:81 */
void cigar_destruct(Cigar *cigar);      /* inc/paf.h:86 */

typedef struct _paf { /* inc/paf.h:88-109 */
    char *query_name;
    int64_t query_length;
    int64_t query_start;
    int64_t query_end;
    char *target_name;
    int64_t target_length;
    int64_t target_start;
    int64_t target_end;
    Cigar *cigar;
    char *cigar_string;
    int64_t score;
    int64_t mapping_quality;
    int64_t num_matches;
    int64_t num_bases;
    int64_t tile_level;
    int64_t chain_id;
    int64_t chain_score;
    bool same_strand;
    char type;
} Paf;

void paf_destruct(Paf *paf);                                         /* inc/paf.h:114 */
Paf *paf_parse(char *paf_string, bool parse_cigar_string);           /* inc/paf.h:119 */
Paf *paf_read(FILE *fh, bool parse_cigar_string);                    /* inc/paf.h:124 */
Paf *paf_read2(FILE *fh);                                            /* inc/paf.h:129 */
Paf *paf_read_with_buffer(FILE *fh, bool parse_cigar_string, char **paf_buffer, int64_t *paf_length_buffer); /* inc/paf.h:135 */
char *paf_print(Paf *paf);                                           /* inc/paf.h:140 */
void paf_stats_calc(Paf *paf, int64_t *matches, int64_t *mismatches, int64_t *query_inserts, int64_t *query_deletes,
                    int64_t *query_insert_bases, int64_t *query_delete_bases, bool zero_counts); /* inc/paf.h:145 */
void paf_pretty_print(Paf *paf, char *query_seq, char *target_seq, FILE *fh, bool include_alignment); /* inc/paf.h:151 */
void paf_write(Paf *paf, FILE *fh);                                  /* inc/paf.h:156 */
void paf_write_with_buffer(Paf *paf, FILE *fh, char **paf_buffer, int64_t *paf_length_buffer); /* inc/paf.h:161 */
void paf_check(Paf *paf);                                            /* inc/paf.h:167 */
void paf_invert(Paf *paf);                                           /* inc/paf.h:172 */
int64_t paf_get_number_of_aligned_bases(Paf *paf);                   /* inc/paf.h:194 */
void paf_trim_ends(Paf *paf, int64_t end_bases_to_trim);             /* inc/paf.h:199 */
void paf_trim_end_fraction(Paf *paf, float percentage);              /* inc/paf.h:204 */
void paf_encode_mismatches(Paf *paf, char *query_seq, char *target_seq); /* inc/paf.h:257 */
void paf_remove_mismatches(Paf *paf);                                /* inc/paf.h:262 */
void paf_trim_unreliable_tails(Paf *paf, float score_fraction, float max_fraction_to_trim); /* inc/paf.h:268 */

/* sonLib-free forms of read_pafs / write_pafs / paf_shatter (inc/paf.h:177,182,209): malloc'ed arrays of Paf pointers, each
 * freed with paf_destruct, the array with free(). One GPU batch per call whatever the number of records. */
Paf **read_pafs_array(FILE *paf_file, bool parse_cigar_string, int64_t *n_pafs);
void write_pafs_array(FILE *paf_file, Paf **pafs, int64_t n_pafs);
Paf **paf_shatter_array(Paf *paf, int64_t *n_pafs);
/* paf_chain (inc/paf.h:187) over a plain array, with the affine gap cost of its caller (impl/paf_chain.c:36-45) as two numbers: the
 * same Paf objects in a new malloc'ed array, by descending score, chain_id and chain_score set */
Paf **paf_chain_array(Paf **pafs, int64_t n_pafs, int64_t gap_open, int64_t gap_extend, int64_t max_gap_length, float percentage_to_trim, int64_t *n_out);

typedef struct _sequenceCountArray { /* inc/paf.h:214-218: alignment coverage along a sequence */
    char *name;
    int64_t length;
    uint16_t *counts; /* one per base */
} SequenceCountArray;
void sequenceCountArray_destruct(SequenceCountArray *seq_count_array);                    /* inc/paf.h:223 */
void increase_alignment_level_counts(SequenceCountArray *seq_count_array, Paf *paf);      /* inc/paf.h:233 */
/* get_alignment_count_array (inc/paf.h:228) over a plain array of count arrays (*arrays grows by realloc) instead of a stHash */
SequenceCountArray *get_alignment_count_array_in(SequenceCountArray ***arrays, int64_t *n_arrays, Paf *paf);

typedef struct _interval { /* inc/paf.h:235-238 */
    char *name;
    int64_t start, end, length;
} Interval;
void interval_destruct(Interval *interval);       /* inc/paf.h:243 */
Interval *decode_fasta_header(char *fasta_header); /* inc/paf.h:248 */
int cmp_intervals(const void *i, const void *j);  /* inc/paf.h:253 */

#ifdef PAFFY_WITH_SONLIB /* needs sonLib's stList at compile and link time */
stList *read_pafs(FILE *paf_file, bool parse_cigar_string);
void write_pafs(FILE *paf_file, stList *pafs);
stList *paf_shatter(Paf *paf);
stList *paf_chain(stList *pafs, int64_t (*gap_cost)(int64_t, int64_t, void *), void *gap_cost_params, int64_t max_gap_length, float percentage_to_trim);
#endif

#ifdef __cplusplus
}
#endif
#endif /* ST_PAF_H_ */
