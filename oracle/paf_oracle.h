/*
 * paf_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A scalar, record-at-a-time CPU restatement of the reference paffy hot path
 * (parse -> cigar walk -> shatter/invert/trim/add_mismatches/tile -> write),
 * written from the behaviour of /root/reference (file:line cited per function
 * in paf_oracle.c). It is the checker for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (paffy_amd/, host/) never links or calls anything in oracle/.
 *
 * Pinning: the reference cannot be built in this image (its only dependency,
 * the sonLib submodule, is absent), so this restatement is pinned by the
 * reference's own known-answer tests (tests/paf_unit_test.c) and its fixture
 * round-trip (tests/paf_test.c on tests/human_chimp.paf); see
 * tests/test_oracle_kat.py.  Behaviours that live in sonLib and that no
 * reference test pins (reverse-complement of non-ACGT letters, qsort tie
 * order, FASTA header keys, '\r' handling) are "parity unpinned" and follow
 * SURVEY.md Appendix C.
 */
#ifndef PAF_ORACLE_H_
#define PAF_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Stage kinds: one per reference command on the hot path. */
enum {
    PO_INVERT = 1,            /* paffy invert            impl/paf_invert.c:84-89  */
    PO_TRIM_IDENTITY = 2,     /* paffy trim (default)    impl/paf_trim.c:117-119  p0=-r, p1=-t */
    PO_TRIM_FIXED = 3,        /* paffy trim -f           impl/paf_trim.c:120-122  p1=-t */
    PO_SHATTER = 4,           /* paffy shatter           impl/paf_shatter.c:88-95 */
    PO_ADD_MISMATCHES = 5,    /* paffy add_mismatches    impl/paf_add_mismatches.c:113-131 */
    PO_REMOVE_MISMATCHES = 6, /* paffy add_mismatches -a impl/paf_add_mismatches.c:110-112 */
    PO_PASS = 7,              /* parse + write only (paf_read -> paf_write)       */
    PO_FILTER = 8             /* paffy filter            impl/paf_filter.c:120-156 (thresholds: po_set_filter) */
};

/* Error codes; the exit status the reference would give is in po_error_exit_status(). */
enum {
    PO_OK = 0,
    PO_ERR_FEW_FIELDS = 1,     /* <12 fields / blank line: reference dereferences NULL (impl/paf.c:144-172) */
    PO_ERR_STRAND = 2,         /* st_errAbort impl/paf.c:155-157 */
    PO_ERR_TP_ASSERT = 3,      /* assert impl/paf.c:190 */
    PO_ERR_CIGAR_CHAR = 4,     /* st_errAbort impl/paf.c:102 */
    PO_ERR_CHECK_QSTART = 5,   /* impl/paf.c:428 */
    PO_ERR_CHECK_QEND = 6,     /* impl/paf.c:431 */
    PO_ERR_CHECK_TSTART = 7,   /* impl/paf.c:434 */
    PO_ERR_CHECK_TEND = 8,     /* impl/paf.c:437 */
    PO_ERR_CHECK_CIGAR_Q = 9,  /* impl/paf.c:452 */
    PO_ERR_CHECK_CIGAR_T = 10, /* impl/paf.c:456 */
    PO_ERR_SHATTER_ZERO_LEN = 11, /* assert impl/paf.c:635 */
    PO_ERR_SHATTER_BAD_OP = 12,   /* assert impl/paf.c:650 */
    PO_ERR_SHATTER_END = 13,      /* asserts impl/paf.c:654-660 */
    PO_ERR_TRIM_IDENTITY_ASSERT = 14, /* assert impl/paf.c:952 */
    PO_ERR_TRIM_FIXED_ASSERT = 15,    /* assert impl/paf.c:591 */
    PO_ERR_NULL_CIGAR = 16,    /* NULL cigar dereferenced (impl/paf.c:520, paf_tile.c:166) */
    PO_ERR_MISSING_QUERY_SEQ = 17,  /* exit(1) impl/paf_add_mismatches.c:117-120 */
    PO_ERR_MISSING_TARGET_SEQ = 18, /* exit(1) impl/paf_add_mismatches.c:123-127 */
    PO_ERR_TILE_ASSERT = 19,   /* asserts impl/paf.c:685,698,708, impl/paf_tile.c:57,86,171 */
    PO_ERR_STATS_BAD_OP = 20,  /* assert impl/paf.c:255 */
    PO_ERR_SEQ_RANGE = 21,     /* encode_mismatches would read outside a sequence (undefined in the reference) */
    PO_ERR_CHAIN_ASSERT = 22   /* asserts impl/chaining.c:275,278-281 */
};

typedef struct {
    int32_t kind;
    float p0; /* trim: trim_by_identity_fraction (-r), float as in impl/paf_trim.c:16 */
    float p1; /* trim: trim_end_fraction (-t),        float as in impl/paf_trim.c:14 */
} po_stage;

/* `paffy filter` options as its main() holds them (impl/paf_filter.c:27-32; -s -t -w go through atoi, -u -v through atof). */
typedef struct {
    int64_t min_chain_score;       /* -s, default -1 */
    int64_t min_alignment_score;   /* -t, default -1 */
    double min_identity;           /* -u, default -1.0 */
    double min_identity_with_gaps; /* -v, default -1.0 */
    int64_t max_tile_level;        /* -w, default -1 (no limit) */
    int32_t invert;                /* -x */
} po_filter;
/* thresholds used by every PO_FILTER stage of later po_run calls (NULL: defaults) */
void po_set_filter(const po_filter *f);

typedef struct {
    const char *name; /* NUL-terminated key (FASTA header) */
    const char *seq;  /* sequence bytes */
    int64_t len;
} po_seq;

typedef struct {
    int32_t code;   /* PO_ERR_* */
    int32_t stage;  /* index into the stage list (-1 = parse) */
    int64_t record; /* zero-based input record */
    int64_t aux;    /* offending character etc. */
} po_error;

/*
 * Run a chain of stream commands (`paffy a | paffy b | ...`) record by record
 * over a PAF text buffer. Output is malloc'ed (release with po_free). On an
 * error the output holds everything the records before the failing one
 * produced and the return value is the error code.
 */
int po_run(const po_stage *stages, int32_t n_stages, const char *in, int64_t in_len,
           const po_seq *seqs, int64_t n_seqs, char **out, int64_t *out_len, po_error *err);

/* `paffy tile` over a whole buffer (impl/paf_tile.c:156-178). */
int po_tile(const char *in, int64_t in_len, char **out, int64_t *out_len, po_error *err);
/* paffy to_bed [-b -e -f -m min_size -n] (impl/paf_to_bed.c); sequences in order of first appearance */
int po_to_bed(const char *in, int64_t in_len, int binary, int exclude_unaligned, int exclude_aligned, int64_t min_size, int include_inverted,
              char **out, int64_t *out_len, po_error *err);

/*
 * `paffy dedupe [-a]` over a whole buffer (impl/paf_dedupe.c:117-143): records are read without parsing the cigar; a
 * record is written (cigar text verbatim) unless an earlier written record has the same query name, target name, strand
 * and four coordinates; with check_inverse also unless an earlier one equals it with query and target swapped -- in that
 * case paf_check runs on the record (impl/paf_dedupe.c:122-127: only then).
 */
int po_dedupe(const char *in, int64_t in_len, int check_inverse, char **out, int64_t *out_len, po_error *err);

/*
 * `paffy chain` over a whole buffer (impl/paf_chain.c:123-127, impl/chaining.c:136-343) with the affine gap cost of
 * impl/paf_chain.c:36-45 (0 for no gap, else gap_open + gap_extend * (query gap + target gap)). Address comparisons are restated
 * as creation order (see the .c file): parity on exact ties is unpinned. *fresh_hits counts the candidates that were only seen
 * because a search found no active chain <= the key (libavl's avl_t_prev from a fresh traverser starts at the largest element).
 */
int po_chain(const char *in, int64_t in_len, int64_t gap_open, int64_t gap_extend, int64_t max_gap, float pct, char **out, int64_t *out_len,
             int64_t *fresh_hits, po_error *err);

/*
 * `paffy split_file` (impl/paf_split_file.c:131-173): every record (cigar text verbatim) goes to "<prefix><contig>.paf"
 * ('/' in the name becomes '_'), the contig being the target name, or the query name with by_query; contigs shorter than
 * min_length (> 0) share "<prefix>small_<k>.paf" files filled in first-seen order up to min_length bases each. Files are
 * created by this call (test infrastructure: point the prefix into a scratch directory).
 */
int po_split_file(const char *in, int64_t in_len, const char *prefix, int by_query, int64_t min_length, po_error *err);

void po_free(void *p);

/* Exit status the reference process would end with for an error code (1, 134 or 139). */
int po_error_exit_status(int32_t code);

/* ---- small library-level probes used by the known-answer tests ---- */

/* cigar_parse (impl/paf.c:70-111): returns op count, -1 for NULL (empty string), -2 on bad char. */
int64_t po_cigar_parse(const char *cigar, int64_t *lens, int32_t *ops, int64_t cap);

/* paf_stats_calc (impl/paf.c:236-260) on a cigar string; out[6] = matches, mismatches,
 * query_inserts, query_deletes, query_insert_bases, query_delete_bases (accumulated). */
int po_cigar_stats(const char *cigar, int64_t *out6, int zero_counts);

/* paf_get_number_of_aligned_bases (impl/paf.c:507-516). */
int64_t po_cigar_aligned_bases(const char *cigar);

/* paf_trim_ends (impl/paf.c:578-587) on one PAF line with an explicit base count. */
int po_trim_ends_line(const char *line, int64_t line_len, int64_t end_bases, char **out, int64_t *out_len);

/* paf_pretty_print (impl/paf.c:262-316) of one line: the stats line and, with include_alignment, the base-level rows */
int po_pretty_print(const char *line, int64_t line_len, const char *query_seq, const char *target_seq, int include_alignment, char **out,
                    int64_t *out_len);

/* Coverage counters (impl/paf.c:675-709): apply the records of a buffer in input order to
 * the counter array of query `name` (length `len`, caller zeroed); returns records applied. */
int64_t po_coverage_counts(const char *in, int64_t in_len, const char *name, uint16_t *counts, int64_t len);

#ifdef __cplusplus
}
#endif
#endif
