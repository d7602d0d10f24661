/*
 * Known-answer tests of the per-record C API (include/paf.h over the MI355X engine). The records and the expected values
 * are those of the reference's own unit tests (tests/paf_unit_test.c: cigar parsing :52-95, paf parsing :110-190,
 * invert :334-399, aligned bases :403-409, trimming :413-467, shatter :471-560, mismatches :565-700), rebuilt here against
 * this repo's header. Usage: paf_api_kat [fixture.paf out.paf]  -- exit status 0 = all checks passed.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/paf.h"

static int failures = 0, checks = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        checks++;                                                                    \
        if (!(cond)) {                                                               \
            failures++;                                                              \
            fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);          \
        }                                                                            \
    } while (0)

static Paf *make_paf(const char *qname, int64_t qlen, int64_t qs, int64_t qe, bool same_strand, const char *tname, int64_t tlen, int64_t ts,
                     int64_t te, int64_t nm, int64_t nb, int64_t mq, const char *cigar_str) {
    Paf *p = calloc(1, sizeof(Paf));
    p->query_name = strdup(qname);
    p->query_length = qlen; p->query_start = qs; p->query_end = qe;
    p->target_name = strdup(tname);
    p->target_length = tlen; p->target_start = ts; p->target_end = te;
    p->same_strand = same_strand;
    p->num_matches = nm; p->num_bases = nb; p->mapping_quality = mq;
    p->tile_level = p->chain_id = p->chain_score = -1;
    if (cigar_str) {
        char *cs = strdup(cigar_str);
        p->cigar = cigar_parse(cs);
        free(cs);
    }
    return p;
}
static Paf *parse_str(const char *s, bool cigar) {
    char *copy = strdup(s);
    Paf *p = paf_parse(copy, cigar);
    free(copy);
    return p;
}
static int op_is(Paf *p, int64_t i, CigarOp op, int64_t len) { return cigar_get(p->cigar, i)->op == (int64_t)op && cigar_get(p->cigar, i)->length == len; }

static void cigars(void) {
    char e[] = "";
    CHECK(cigar_parse(e) == NULL);
    char s1[] = "10M";
    Cigar *c = cigar_parse(s1);
    CHECK(c && cigar_count(c) == 1 && cigar_get(c, 0)->op == match && cigar_get(c, 0)->length == 10);
    cigar_destruct(c);
    char s2[] = "5M3I2D4=1X";
    c = cigar_parse(s2);
    CHECK(c && cigar_count(c) == 5);
    const CigarOp want_op[5] = {match, query_insert, query_delete, sequence_match, sequence_mismatch};
    const int64_t want_len[5] = {5, 3, 2, 4, 1};
    for (int i = 0; c && i < 5; i++) CHECK(cigar_get(c, i)->op == (int64_t)want_op[i] && cigar_get(c, i)->length == want_len[i]);
    cigar_destruct(c);
    char s3[] = "1000000M";
    c = cigar_parse(s3);
    CHECK(c && cigar_count(c) == 1 && cigar_get(c, 0)->length == 1000000);
    cigar_destruct(c);
    CHECK(cigar_count(NULL) == 0);
    /* ops that do not fit the 4-byte stores of the fast path (length >= 2^29) and more ops than any LDS store holds: the arena class */
    char s4[] = "600000000M3I72057594037927935D";
    c = cigar_parse(s4);
    CHECK(c && cigar_count(c) == 3 && cigar_get(c, 0)->length == 600000000 && cigar_get(c, 1)->op == query_insert && cigar_get(c, 2)->op == query_delete);
    CHECK(c && cigar_get(c, 2)->length == -1); /* 2^56 - 1 in the signed 56-bit field, as the reference's bitfield stores it */
    cigar_destruct(c);
    const int n_ops = 50001;
    char *big = malloc((size_t)n_ops * 2 + 1);
    for (int i = 0; i < n_ops; i++) {
        big[2 * i] = (char)('1' + i % 9);
        big[2 * i + 1] = "MID"[i % 3];
    }
    big[2 * n_ops] = '\0';
    c = cigar_parse(big);
    CHECK(c && cigar_count(c) == n_ops && cigar_get(c, 50000)->length == 1 + 50000 % 9 && cigar_get(c, 50000)->op == (int64_t)(50000 % 3));
    CHECK(c && cigar_get(c, 12345)->length == 1 + 12345 % 9 && cigar_get(c, 12345)->op == (int64_t)(12345 % 3));
    cigar_destruct(c);
    free(big);
}

static void parsing(void) {
    Paf *p = parse_str("query1\t100\t0\t50\t+\ttarget1\t200\t10\t60\t50\t50\t255", true);
    CHECK(strcmp(p->query_name, "query1") == 0 && p->query_length == 100 && p->query_start == 0 && p->query_end == 50);
    CHECK(strcmp(p->target_name, "target1") == 0 && p->target_length == 200 && p->target_start == 10 && p->target_end == 60);
    CHECK(p->num_matches == 50 && p->num_bases == 50 && p->mapping_quality == 255 && p->same_strand);
    CHECK(p->cigar == NULL && p->cigar_string == NULL && p->tile_level == -1 && p->chain_id == -1 && p->chain_score == -1 && p->score == 0 && p->type == '\0');
    paf_destruct(p);
    p = parse_str("q\t100\t0\t10\t-\tt\t200\t5\t15\t10\t10\t60\ttp:A:S\tAS:i:77\ttl:i:2\tcn:i:5\ts1:i:1234\tzz:i:9\tcg:Z:4M2I4M2D", true);
    CHECK(!p->same_strand && p->type == 'S' && p->score == 77 && p->tile_level == 2 && p->chain_id == 5 && p->chain_score == 1234);
    CHECK(cigar_count(p->cigar) == 4 && op_is(p, 0, match, 4) && op_is(p, 1, query_insert, 2) && op_is(p, 2, match, 4) && op_is(p, 3, query_delete, 2));
    char *line = paf_print(p);
    CHECK(strcmp(line, "q\t100\t0\t10\t-\tt\t200\t5\t15\t10\t10\t60\ttp:A:S\tAS:i:77\ttl:i:2\tcn:i:5\ts1:i:1234\tcg:Z:4M2I4M2D") == 0);
    free(line);
    paf_destruct(p);
    p = parse_str("q\t100\t0\t10\t+\tt\t200\t5\t15\t10\t10\t60\tcg:Z:10M", false); /* cigar kept as text */
    CHECK(p->cigar == NULL && p->cigar_string && strcmp(p->cigar_string, "10M") == 0);
    line = paf_print(p);
    CHECK(strcmp(line, "q\t100\t0\t10\t+\tt\t200\t5\t15\t10\t10\t60\tAS:i:0\tcg:Z:10M") == 0);
    free(line);
    paf_destruct(p);
}

static void inverting(void) {
    Paf *p = make_paf("query", 100, 10, 18, true, "target", 200, 20, 27, 8, 10, 60, "5M3I2D");
    paf_invert(p);
    CHECK(strcmp(p->query_name, "target") == 0 && strcmp(p->target_name, "query") == 0);
    CHECK(p->query_start == 20 && p->query_end == 27 && p->query_length == 200 && p->target_start == 10 && p->target_end == 18 && p->target_length == 100);
    CHECK(p->same_strand && cigar_count(p->cigar) == 3 && op_is(p, 0, match, 5) && op_is(p, 1, query_delete, 3) && op_is(p, 2, query_insert, 2));
    paf_destruct(p);
    p = make_paf("query", 100, 10, 18, false, "target", 200, 20, 25, 5, 8, 60, "5M3I");
    paf_invert(p);
    CHECK(!p->same_strand && cigar_count(p->cigar) == 2 && op_is(p, 0, query_delete, 3) && op_is(p, 1, match, 5));
    paf_destruct(p);
    p = make_paf("query", 100, 10, 18, true, "target", 200, 20, 27, 8, 10, 60, "5M3I2D");
    char *orig = paf_print(p);
    paf_invert(p);
    paf_invert(p);
    char *trip = paf_print(p);
    CHECK(strcmp(orig, trip) == 0);
    paf_check(p);
    free(orig);
    free(trip);
    paf_destruct(p);
}

static void trimming(void) {
    Paf *p = make_paf("q", 100, 0, 13, true, "t", 100, 0, 12, 10, 15, 60, "5M3I2D4=1X");
    CHECK(paf_get_number_of_aligned_bases(p) == 10);
    int64_t mat = 0, mis = 0, qi = 0, qd = 0, qib = 0, qdb = 0;
    paf_stats_calc(p, &mat, &mis, &qi, &qd, &qib, &qdb, true);
    CHECK(mat == 9 && mis == 1 && qi == 1 && qib == 3 && qd == 1 && qdb == 2);
    paf_destruct(p);
    p = make_paf("q", 100, 0, 5, true, "t", 100, 0, 5, 5, 5, 60, "5M"); /* zero_counts = false accumulates, true resets first */
    mat = mis = qi = qd = qib = qdb = 0;
    paf_stats_calc(p, &mat, &mis, &qi, &qd, &qib, &qdb, false);
    paf_stats_calc(p, &mat, &mis, &qi, &qd, &qib, &qdb, false);
    CHECK(mat == 10);
    paf_stats_calc(p, &mat, &mis, &qi, &qd, &qib, &qdb, true);
    CHECK(mat == 5 && mis == 0 && qi == 0 && qd == 0);
    paf_destruct(p);
    p = make_paf("q", 100, 5, 15, true, "t", 100, 5, 15, 10, 10, 60, "10M");
    paf_trim_ends(p, 0);
    CHECK(p->query_start == 5 && p->query_end == 15 && p->target_start == 5 && p->target_end == 15 && cigar_count(p->cigar) == 1 && op_is(p, 0, match, 10));
    paf_destruct(p);
    p = make_paf("q", 100, 0, 10, true, "t", 100, 0, 10, 10, 10, 60, "10M");
    paf_trim_ends(p, 2);
    CHECK(p->query_start == 2 && p->query_end == 8 && p->target_start == 2 && p->target_end == 8 && cigar_count(p->cigar) == 1 && op_is(p, 0, match, 6));
    paf_destruct(p);
    p = make_paf("q", 100, 0, 8, true, "t", 100, 0, 7, 7, 8, 60, "2M1I5M");
    paf_trim_ends(p, 3);
    CHECK(p->query_start == 4 && p->target_start == 3 && p->query_end == 5 && p->target_end == 4);
    paf_destruct(p);
    p = make_paf("q", 100, 0, 10, true, "t", 100, 0, 10, 10, 10, 60, "10M");
    paf_trim_end_fraction(p, 0.4f);
    CHECK(p->query_start == 2 && p->query_end == 8 && p->target_start == 2 && p->target_end == 8);
    paf_destruct(p);
    /* identity trim: the noisy head goes; on the + strand the reference's second pass looks at the prefix again,
       so the tail stays -- expected values from the oracle */
    p = make_paf("q", 1000, 0, 124, true, "t", 1000, 0, 124, 100, 124, 60, "2=10X100=10X2=");
    paf_trim_unreliable_tails(p, 0.05f, 1.0f);
    CHECK(cigar_count(p->cigar) == 3 && op_is(p, 0, sequence_match, 100) && op_is(p, 1, sequence_mismatch, 10) && op_is(p, 2, sequence_match, 2));
    CHECK(p->query_start == 12 && p->query_end == 124 && p->target_start == 12 && p->target_end == 124);
    paf_destruct(p);
}

static void shattering(void) {
    Paf *p = make_paf("q", 100, 0, 5, true, "t", 100, 0, 5, 5, 5, 60, "5M");
    int64_t n = 0;
    Paf **parts = paf_shatter_array(p, &n);
    CHECK(n == 1 && strcmp(parts[0]->query_name, "q") == 0 && parts[0]->query_start == 0 && parts[0]->query_end == 5 && parts[0]->target_start == 0 && parts[0]->target_end == 5);
    for (int64_t i = 0; i < n; i++) paf_destruct(parts[i]);
    free(parts);
    paf_destruct(p);
    /* 3M2I4M1D2M on the + strand: blocks q[10,13) t[20,23); q[15,19) t[23,27); q[19,21) t[28,30) */
    p = make_paf("q", 100, 10, 21, true, "t", 100, 20, 30, 9, 12, 60, "3M2I4M1D2M");
    parts = paf_shatter_array(p, &n);
    CHECK(n == 3);
    const int64_t wq[3][2] = {{10, 13}, {15, 19}, {19, 21}}, wt[3][2] = {{20, 23}, {23, 27}, {28, 30}};
    for (int64_t i = 0; i < n && i < 3; i++) {
        CHECK(parts[i]->query_start == wq[i][0] && parts[i]->query_end == wq[i][1] && parts[i]->target_start == wt[i][0] && parts[i]->target_end == wt[i][1]);
        CHECK(cigar_count(parts[i]->cigar) == 1 && cigar_get(parts[i]->cigar, 0)->op == match && parts[i]->num_matches == wq[i][1] - wq[i][0]);
    }
    for (int64_t i = 0; i < n; i++) paf_destruct(parts[i]);
    free(parts);
    paf_destruct(p);
    /* - strand: the query runs backwards from query_end (impl/paf.c:641-646) */
    p = make_paf("q", 100, 10, 19, false, "t", 100, 20, 27, 7, 9, 60, "3M2I4M");
    parts = paf_shatter_array(p, &n);
    CHECK(n == 2 && parts[0]->query_start == 16 && parts[0]->query_end == 19 && parts[0]->target_start == 20 && parts[0]->target_end == 23);
    CHECK(n == 2 && parts[1]->query_start == 10 && parts[1]->query_end == 14 && parts[1]->target_start == 23 && parts[1]->target_end == 27 && !parts[1]->same_strand);
    for (int64_t i = 0; i < n; i++) paf_destruct(parts[i]);
    free(parts);
    paf_destruct(p);
}

static void mismatches(void) {
    /*            0123456789 */
    char q[] = "ACGTACGTAC", t[] = "ACGTTCGTAC";
    Paf *p = make_paf("q", 10, 0, 10, true, "t", 10, 0, 10, 10, 10, 60, "10M");
    paf_encode_mismatches(p, q, t);
    CHECK(cigar_count(p->cigar) == 3 && op_is(p, 0, sequence_match, 4) && op_is(p, 1, sequence_mismatch, 1) && op_is(p, 2, sequence_match, 5));
    CHECK(strcmp(p->query_name, "q") == 0 && strcmp(p->target_name, "t") == 0);
    paf_remove_mismatches(p);
    CHECK(cigar_count(p->cigar) == 1 && op_is(p, 0, match, 10));
    paf_destruct(p);
    /* - strand: column i pairs T[i] with the complement of Q[qe - 1 - i]; lower case and N compare by toupper */
    char q2[] = "GTaCGn", t2[] = "NCGTAC"; /* rc(q2) = nCGtAC */
    p = make_paf("q", 6, 0, 6, false, "t", 6, 0, 6, 6, 6, 60, "6M");
    paf_encode_mismatches(p, q2, t2);
    CHECK(cigar_count(p->cigar) == 1 && op_is(p, 0, sequence_match, 6));
    paf_destruct(p);
    /* a base edited IN PLACE between two calls (same pointers, same lengths) must be seen: the reference reads the caller's strings
       on every call (impl/paf.c:752-757); the pair cached on the GPU is keyed by a hash over every byte, not a sample.
       20 000 bases: a sample of 4 096 positions would step over base 7 777 */
    enum { N = 20000 };
    static char big_q[N + 1], big_t[N + 1];
    for (int i = 0; i < N; i++) big_q[i] = big_t[i] = "ACGT"[(i * 7 + i / 13) & 3];
    big_q[N] = big_t[N] = '\0';
    p = make_paf("q", N, 0, N, true, "t", N, 0, N, N, N, 60, "20000M");
    paf_encode_mismatches(p, big_q, big_t);
    CHECK(cigar_count(p->cigar) == 1 && op_is(p, 0, sequence_match, N));
    paf_destruct(p);
    big_q[7777] = big_q[7777] == 'A' ? 'C' : 'A';
    p = make_paf("q", N, 0, N, true, "t", N, 0, N, N, N, 60, "20000M");
    paf_encode_mismatches(p, big_q, big_t);
    CHECK(cigar_count(p->cigar) == 3 && op_is(p, 0, sequence_match, 7777) && op_is(p, 1, sequence_mismatch, 1) && op_is(p, 2, sequence_match, N - 7778));
    paf_destruct(p);
}

/* paf_pretty_print (impl/paf.c:262-316): the answers are the columns derived by hand in tests/test_oracle_kat.py */
static void pretty(void) {
    char q[] = "ACGTTGGACA", t[] = "NACGTAACTAGG";
    Paf *p = make_paf("q", 10, 0, 10, true, "t", 12, 1, 10, 8, 10, 60, "4=1X2I2=1D1=");
    p->score = 7;
    char *text = NULL;
    size_t len = 0;
    FILE *fh = open_memstream(&text, &len);
    paf_pretty_print(p, q, t, fh, true);
    fclose(fh);
    CHECK(strcmp(text, "Query:q\tQ-start:0\tQ-length:10\tTarget:t\tT-start:1\tT-length:9\tSame-strand:1\tScore:7\tIdentity:0.875000"
                       "\tIdentity-with-gaps0.636364\tAligned-bases:8\tQuery-inserts:1\tQuery-deletes:1\n"
                       "ACGTA--ACTA\nACGTTGGAC-A\n****   ** *\n") == 0);
    free(text);
    fh = open_memstream(&text, &len);
    paf_pretty_print(p, q, t, fh, false);
    fclose(fh);
    CHECK(strchr(text, '\n') == text + len - 1 && strncmp(text, "Query:q\tQ-start:0", 17) == 0);
    free(text);
    paf_destruct(p);
    /* - strand, mixed case, and a second pair of sequences right after the first (the loaded pair must be replaced) */
    char q2[] = "GTaCGn", t2[] = "NCGTAC";
    p = make_paf("q", 6, 0, 6, false, "t", 6, 0, 6, 6, 6, 60, "6M");
    fh = open_memstream(&text, &len);
    paf_pretty_print(p, q2, t2, fh, true);
    fclose(fh);
    const char *rows = strchr(text, '\n');
    CHECK(rows && strcmp(rows + 1, "NCGTAC\nnCGtAC\n******\n") == 0);
    free(text);
    paf_destruct(p);
    /* three windows: 150 + 150 + 10 columns */
    char *qa = malloc(311), *ta = malloc(311);
    memset(qa, 'A', 310);
    memset(ta, 'A', 310);
    qa[310] = ta[310] = '\0';
    ta[200] = 'c';
    p = make_paf("q", 310, 0, 310, true, "t", 310, 0, 310, 310, 310, 60, "310M");
    fh = open_memstream(&text, &len);
    paf_pretty_print(p, qa, ta, fh, true);
    fclose(fh);
    rows = strchr(text, '\n') + 1;
    CHECK(strlen(rows) == 3 * 310 + 9);
    CHECK(rows[150] == '\n' && rows[3 * 151 + 50] == 'c' && rows[3 * 151 + 2 * 151 + 50] == ' ' && rows[3 * 151 + 2 * 151 + 49] == '*');
    CHECK(rows[6 * 151 + 10] == '\n' && rows[6 * 151 + 32] == '\n');
    free(text);
    free(qa);
    free(ta);
    paf_destruct(p);
}

/* the library transforms check nothing and keep the tags they do not compute (impl/paf.c:463-490) */
static void unchecked_and_preserved(void) {
    Paf *p = make_paf("q", 100, 0, 9, true, "t", 100, 0, 10, 10, 10, 60, "10M"); /* cigar and query coordinates disagree */
    p->score = 2147483647; /* INT_MAX: the writer leaves the tag out; the struct must keep the value */
    p->tile_level = 3;
    p->type = '\0';
    p->chain_id = 5;
    p->chain_score = 77;
    paf_invert(p);
    CHECK(strcmp(p->query_name, "t") == 0 && p->query_end == 10 && p->target_end == 9);
    CHECK(p->score == 2147483647 && p->tile_level == 3 && p->type == '\0' && p->chain_id == 5 && p->chain_score == 77);
    paf_invert(p);
    CHECK(strcmp(p->query_name, "q") == 0 && p->query_end == 9 && p->target_end == 10);
    paf_destruct(p);
    p = make_paf("q", 100, 10, 20, true, "t", 100, 30, 40, 10, 10, 60, "10M");
    paf_check(p); /* a sound record passes and is left alone */
    CHECK(p->query_start == 10 && cigar_count(p->cigar) == 1);
    paf_destruct(p);
}

/* coverage counters and fasta-header intervals (impl/paf.c:667-737) */
static void counts_and_intervals(void) {
    SequenceCountArray **arrays = NULL;
    int64_t n = 0;
    Paf *a = make_paf("q", 30, 2, 12, true, "t", 100, 0, 9, 10, 10, 60, "4M1D2M2I2=");  /* query bases 2-5, 6-7, 10-11 */
    Paf *b = make_paf("q", 30, 4, 8, false, "u", 100, 50, 54, 4, 4, 60, "4M");
    Paf *c = make_paf("r", 8, 0, 8, true, "t", 100, 0, 8, 8, 8, 60, "8M");
    SequenceCountArray *ca = get_alignment_count_array_in(&arrays, &n, a);
    CHECK(n == 1 && ca->length == 30 && strcmp(ca->name, "q") == 0 && ca->counts[0] == 0);
    increase_alignment_level_counts(ca, a);
    CHECK(get_alignment_count_array_in(&arrays, &n, b) == ca && n == 1);
    increase_alignment_level_counts(ca, b);
    SequenceCountArray *cr = get_alignment_count_array_in(&arrays, &n, c);
    CHECK(n == 2 && cr != ca && cr->length == 8);
    increase_alignment_level_counts(cr, c);
    const uint16_t want[14] = {0, 0, 1, 1, 2, 2, 2, 2, 0, 0, 1, 1, 0, 0};
    for (int i = 0; i < 14; i++) CHECK(ca->counts[i] == want[i]);
    for (int i = 0; i < 8; i++) CHECK(cr->counts[i] == 1);
    ca->counts[4] = 32766; /* INT16_MAX - 1 stays (impl/paf.c:701) */
    ca->counts[5] = 32765;
    increase_alignment_level_counts(ca, b);
    CHECK(ca->counts[4] == 32766 && ca->counts[5] == 32766 && ca->counts[6] == 3 && ca->counts[8] == 0);
    sequenceCountArray_destruct(ca);
    sequenceCountArray_destruct(cr);
    free(arrays);
    paf_destruct(a);
    paf_destruct(b);
    paf_destruct(c);
    char h1[] = "chr1|1000|50", h2[] = "id=3|chr1|1000|7", h3[] = "chr1|1000|7";
    Interval *i1 = decode_fasta_header(h1), *i2 = decode_fasta_header(h2), *i3 = decode_fasta_header(h3);
    CHECK(strcmp(i1->name, "chr1") == 0 && i1->length == 1000 && i1->start == 50);
    CHECK(strcmp(i2->name, "id=3|chr1") == 0 && i2->length == 1000 && i2->start == 7);
    CHECK(cmp_intervals(i3, i1) < 0 && cmp_intervals(i1, i3) > 0 && cmp_intervals(i1, i1) == 0 && cmp_intervals(i1, i2) < 0);
    interval_destruct(i1);
    interval_destruct(i2);
    interval_destruct(i3);
}

/* paf_chain (impl/chaining.c:266-343): the example worked by hand in tests/test_oracle_kat.py */
static void chaining(void) {
    Paf *in[4] = {make_paf("q", 1000, 0, 100, true, "t", 1000, 0, 100, 10, 20, 60, "5M"), make_paf("q", 1000, 110, 200, true, "t", 1000, 120, 200, 10, 20, 60, "5M"),
                  make_paf("q", 1000, 105, 150, true, "t", 1000, 300, 350, 10, 20, 60, "5M"), make_paf("q", 1000, 210, 300, true, "t", 1000, 210, 300, 10, 20, 60, "5M")};
    const int64_t score[4] = {100, 80, 50, 90};
    for (int i = 0; i < 4; i++) in[i]->score = score[i];
    int64_t n = 0;
    Paf **out = paf_chain_array(in, 4, 10, 1, 1000, 0.0f, &n);
    CHECK(n == 4 && out[0] == in[0] && out[1] == in[3] && out[2] == in[1] && out[3] == in[2]); /* by descending score: the same objects */
    CHECK(in[0]->chain_id == 0 && in[1]->chain_id == 0 && in[3]->chain_id == 0 && in[2]->chain_id == 1);
    CHECK(in[0]->chain_score == 200 && in[3]->chain_score == 200 && in[2]->chain_score == 50);
    CHECK(in[1]->query_start == 110 && in[1]->target_end == 200); /* coordinates untouched */
    for (int i = 0; i < 4; i++) paf_destruct(in[i]);
    free(out);
}

static void files(const char *in_path, const char *out_path) {
    FILE *in = fopen(in_path, "r");
    if (!in) {
        fprintf(stderr, "cannot open %s\n", in_path);
        failures++;
        return;
    }
    int64_t n = 0;
    Paf **all = read_pafs_array(in, true, &n);
    fclose(in);
    CHECK(n > 0);
    FILE *out = fopen(out_path, "w");
    write_pafs_array(out, all, n);
    /* and record by record through the readers / writers */
    in = fopen(in_path, "r");
    Paf *p;
    int64_t k = 0;
    char *buf = NULL;
    int64_t buf_len = 0;
    while ((p = paf_read_with_buffer(in, k % 2 == 0, &buf, &buf_len)) != NULL) {
        if (k < 20) paf_write(p, out); /* every record is a GPU round trip: a few are enough */
        CHECK(strcmp(p->query_name, all[k]->query_name) == 0 && p->target_end == all[k]->target_end);
        paf_destruct(p);
        k++;
        if (k >= 40) break;
    }
    free(buf);
    fclose(in);
    fclose(out);
    for (int64_t i = 0; i < n; i++) paf_destruct(all[i]);
    free(all);
}

/* Records the reference stops at: the process must end the way st_errAbort / assert end it (the caller checks the status). */
static void must_fail(int which) {
    if (which == 1) { /* impl/paf.c:440: query_start beyond query_length */
        Paf *p = make_paf("q", 10, 12, 14, true, "t", 100, 0, 2, 2, 2, 60, "2M");
        paf_check(p);
    } else if (which == 2) { /* impl/paf.c:102: a character that is no cigar op */
        char s[] = "5M3Q";
        cigar_parse(s);
    } else if (which == 3) { /* impl/paf.c:155-157: strand */
        parse_str("q\t100\t0\t50\t*\tt\t200\t10\t60\t50\t50\t255", true);
    } else if (which == 4) { /* impl/paf.c:452: cigar and coordinates disagree */
        Paf *p = make_paf("q", 100, 0, 9, true, "t", 100, 0, 10, 10, 10, 60, "10M");
        paf_invert(p); /* no check here (impl/paf.c:463-490) ... */
        paf_check(p);  /* ... this is the call that ends the process */
    }
}

int main(int argc, char **argv) {
    if (argc == 3 && strcmp(argv[1], "--fail") == 0) {
        must_fail(atoi(argv[2]));
        return 0; /* not reached when the failure is detected */
    }
    cigars();
    parsing();
    inverting();
    trimming();
    shattering();
    mismatches();
    pretty();
    unchecked_and_preserved();
    counts_and_intervals();
    chaining();
    if (argc >= 3) files(argv[1], argv[2]);
    fprintf(stderr, "%d checks, %d failures\n", checks, failures);
    return failures ? 1 : 0;
}
