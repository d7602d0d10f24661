/*
 * paf_api.c -- the per-record C API of the reference (inc/paf.h:75-269, implemented there in impl/paf.c) as a thin host layer
 * over the gfx950 batch engine (include/paffy_hip.h). Builds lib/libstPaf_hip.so.
 *
 * How a call runs: the Paf object is written out as one PAF line in a fixed, fully explicit form (every tag present, so that
 * each struct field travels as it is), the GPU parses / transforms / serialises it exactly as the `paffy <cmd>` drivers'
 * batches, and the result returns either as text (paf_print, paf_write) or as field arrays parsed on the GPU
 * (paffy_hip_parse_host) from which the struct is refilled. No record logic lives here: which tags are printed, how a
 * cigar is parsed, what a transform does to coordinates and ops are all the kernels' (csrc/record_kernel.h).
 *
 * Not thread safe (one process-wide context), like the reference's readers (inc/paf.h:131-135).
 */
#include "../include/paf.h"

#include <inttypes.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>

#include "../include/paffy_hip.h"

static paffy_hip_ctx *g_ctx;

static void die(const char *what, const char *detail) {
    fprintf(stderr, "%s%s%s\n", what, detail ? ": " : "", detail ? detail : "");
    exit(1);
}

static paffy_hip_ctx *ctx(void) {
    if (!g_ctx && paffy_hip_create(&g_ctx, -1) != 0) die("paf.h on MI355X: no HIP device or library", NULL);
    return g_ctx;
}

/* A record the reference would have stopped at: same message class and the same way to end (exit 1 / abort / SIGSEGV). */
static void record_failure(const paffy_plan_info *info) {
    fprintf(stderr, "%s\n", paffy_hip_error_string(info->error.code));
    int st = paffy_hip_error_exit_status(info->error.code);
    if (st == 134) abort();
    if (st == 139) raise(SIGSEGV);
    exit(1);
}

/* ---- text buffer ---- */
typedef struct {
    char *p;
    size_t n, cap;
} Buf;
static void buf_need(Buf *b, size_t extra) {
    if (b->n + extra <= b->cap) return;
    size_t cap = (b->n + extra) * 2 + 256;
    b->p = realloc(b->p, cap);
    if (!b->p) die("out of memory", NULL);
    b->cap = cap;
}
static void buf_str(Buf *b, const char *s, size_t len) {
    buf_need(b, len);
    memcpy(b->p + b->n, s, len);
    b->n += len;
}
static void buf_int(Buf *b, int64_t v) {
    char tmp[24];
    int k = 0;
    uint64_t u = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
    do {
        tmp[k++] = (char)('0' + (int)(u % 10));
        u /= 10;
    } while (u);
    if (v < 0) tmp[k++] = '-';
    buf_need(b, (size_t)k);
    while (k) b->p[b->n++] = tmp[--k];
}
static void buf_field(Buf *b, const char *sep, int64_t v) {
    buf_str(b, sep, strlen(sep));
    buf_int(b, v);
}

/*
 * The hand-over form of a Paf: all twelve columns and every tag the GPU parser knows, so that the parsed record equals the
 * struct field by field (a score of INT_MAX, a tile level of -1 ... are written out too; the GPU writer decides what a
 * PAF line shows, impl/paf.c:343-365). `qname` / `tname` override the names (paf_encode_mismatches).
 */
static void hand_over(Buf *b, const Paf *p, const char *qname, const char *tname) {
    static const char OPC[5] = {'M', 'I', 'D', '=', 'X'};
    const char *qn = qname ? qname : p->query_name, *tn = tname ? tname : p->target_name;
    buf_str(b, qn, strlen(qn));
    buf_field(b, "\t", p->query_length);
    buf_field(b, "\t", p->query_start);
    buf_field(b, "\t", p->query_end);
    buf_str(b, p->same_strand ? "\t+\t" : "\t-\t", 3);
    buf_str(b, tn, strlen(tn));
    buf_field(b, "\t", p->target_length);
    buf_field(b, "\t", p->target_start);
    buf_field(b, "\t", p->target_end);
    buf_field(b, "\t", p->num_matches);
    buf_field(b, "\t", p->num_bases);
    buf_field(b, "\t", p->mapping_quality);
    if (p->type != '\0') {
        buf_str(b, "\ttp:A:", 6);
        buf_str(b, &p->type, 1);
    }
    buf_field(b, "\tAS:i:", p->score);
    buf_field(b, "\ttl:i:", p->tile_level);
    buf_field(b, "\tcn:i:", p->chain_id);
    buf_field(b, "\ts1:i:", p->chain_score);
    if (p->cigar) {
        buf_str(b, "\tcg:Z:", 6);
        for (int64_t i = 0; i < cigar_count(p->cigar); i++) {
            CigarRecord *c = cigar_get(p->cigar, i);
            buf_int(b, c->length);
            buf_str(b, c->op >= 0 && c->op < 5 ? &OPC[c->op] : "N", 1);
        }
    } else if (p->cigar_string) {
        buf_str(b, "\tcg:Z:", 6);
        buf_str(b, p->cigar_string, strlen(p->cigar_string));
    }
    buf_str(b, "\n", 1);
}

/* ---- arrays from the GPU -> structs ---- */
static char *dup_slice(const char *text, uint32_t off, uint32_t len) {
    char *s = malloc((size_t)len + 1);
    if (!s) die("out of memory", NULL);
    memcpy(s, text + off, len);
    s[len] = '\0';
    return s;
}

static Cigar *cigar_from(const paffy_record *r, const uint64_t *ops) {
    if (r->n_ops < 0) return NULL;
    Cigar *c = malloc(sizeof(Cigar));
    if (!c) die("out of memory", NULL);
    c->length = c->capacity = r->n_ops;
    c->start = 0;
    c->recs = malloc(sizeof(CigarRecord) * (size_t)(r->n_ops > 0 ? r->n_ops : 1));
    if (!c->recs) die("out of memory", NULL);
    memcpy(c->recs, ops + r->ops_first, sizeof(CigarRecord) * (size_t)r->n_ops); /* same 8-byte layout */
    return c;
}

static void fill(Paf *p, const char *text, const paffy_record *r, const uint64_t *ops, bool parse_cigar_string) {
    p->query_name = dup_slice(text, r->query_name_off, r->query_name_len);
    p->target_name = dup_slice(text, r->target_name_off, r->target_name_len);
    p->query_length = r->query_length; p->query_start = r->query_start; p->query_end = r->query_end;
    p->target_length = r->target_length; p->target_start = r->target_start; p->target_end = r->target_end;
    p->score = r->score; p->mapping_quality = r->mapping_quality; p->num_matches = r->num_matches; p->num_bases = r->num_bases;
    p->tile_level = r->tile_level; p->chain_id = r->chain_id; p->chain_score = r->chain_score;
    p->same_strand = r->same_strand != 0;
    p->type = (char)r->type;
    p->cigar = NULL;
    p->cigar_string = NULL;
    if (parse_cigar_string) p->cigar = cigar_from(r, ops);                  /* impl/paf.c:197-199 */
    else if (r->cigar_len > 0 || r->n_ops >= 0) p->cigar_string = dup_slice(text, r->cigar_off, r->cigar_len); /* :200-202 */
}

/* Parse `n` lines of text on the GPU into new Paf objects. */
static Paf **parse_lines(const char *text, int64_t len, bool parse_cigar_string, int64_t *n_out) {
    paffy_record *recs = NULL;
    uint64_t *ops = NULL;
    int64_t n_ops = 0;
    paffy_plan_info info;
    int rc = paffy_hip_parse_host(ctx(), text, len, &recs, &ops, &n_ops, &info);
    if (rc) die("paffy_hip_parse_host", paffy_hip_last_error(ctx()));
    if (info.error.code) record_failure(&info);
    Paf **out = malloc(sizeof(Paf *) * (size_t)(info.n_records > 0 ? info.n_records : 1));
    if (!out) die("out of memory", NULL);
    for (int64_t i = 0; i < info.n_records; i++) {
        out[i] = calloc(1, sizeof(Paf));
        if (!out[i]) die("out of memory", NULL);
        fill(out[i], text, &recs[i], ops, parse_cigar_string);
    }
    free(recs);
    free(ops);
    *n_out = info.n_records;
    return out;
}

/* text in -> stage list on the GPU -> text out (malloc'ed) */
static char *run_text(const paffy_stage *st, int n, const char *text, int64_t len, int64_t *out_len) {
    char *out = NULL;
    paffy_plan_info info;
    int rc = paffy_hip_run_host(ctx(), st, n, text, len, &out, out_len, &info);
    if (rc) die("paffy_hip_run_host", paffy_hip_last_error(ctx()));
    if (info.error.code) record_failure(&info);
    return out;
}

static void drop_contents(Paf *p) {
    free(p->query_name);
    free(p->target_name);
    free(p->cigar_string);
    cigar_destruct(p->cigar);
}

/*
 * One record through a stage list, in place. The library transforms of the reference touch coordinates, names (invert) and the
 * cigar only (impl/paf.c:463-598, 739-809, 929-953): score, type, tile level and the chain tags stay as the caller set them,
 * whatever the line writer would show of them (it drops a score of INT_MAX and derives tp from the tile level, impl/paf.c:343-365).
 */
static void transform(Paf *p, const paffy_stage *st, int n, const char *qname, const char *tname) {
    const int64_t score = p->score, tile_level = p->tile_level, chain_id = p->chain_id, chain_score = p->chain_score;
    const char type = p->type;
    Buf b = {0};
    hand_over(&b, p, qname, tname);
    int64_t out_len = 0, n_recs = 0;
    char *text = run_text(st, n, b.p, (int64_t)b.n, &out_len);
    free(b.p);
    Paf **res = parse_lines(text, out_len, p->cigar != NULL || p->cigar_string == NULL, &n_recs);
    if (n_recs != 1) die("paf.h on MI355X: a transform must give one record", NULL);
    if (qname) { /* the stand-in names go, the record keeps its own */
        free(res[0]->query_name);
        free(res[0]->target_name);
        res[0]->query_name = p->query_name;
        res[0]->target_name = p->target_name;
        p->query_name = p->target_name = NULL;
    }
    drop_contents(p);
    *p = *res[0];
    p->score = score;
    p->tile_level = tile_level;
    p->chain_id = chain_id;
    p->chain_score = chain_score;
    p->type = type;
    free(res[0]);
    free(res);
    free(text);
}

/* ---- the API ---- */

Cigar *cigar_parse(char *cigar_string) { /* impl/paf.c:70-111 */
    if (cigar_string[0] == '\0') return NULL;
    Buf b = {0};
    static const char carrier[] = "q\t1\t0\t0\t+\tt\t1\t0\t0\t0\t0\t0\tcg:Z:"; /* a record whose only content is the cigar */
    buf_str(&b, carrier, sizeof(carrier) - 1);
    buf_str(&b, cigar_string, strlen(cigar_string));
    buf_str(&b, "\n", 1);
    int64_t n = 0;
    Paf **res = parse_lines(b.p, (int64_t)b.n, true, &n);
    free(b.p);
    Cigar *c = res[0]->cigar;
    res[0]->cigar = NULL;
    paf_destruct(res[0]);
    free(res);
    return c;
}

void cigar_destruct(Cigar *cigar) { /* impl/paf.c:50-55 */
    if (!cigar) return;
    free(cigar->recs);
    free(cigar);
}

void paf_destruct(Paf *paf) { /* impl/paf.c:57-68 */
    if (!paf) return;
    drop_contents(paf);
    free(paf);
}

Paf *paf_parse(char *paf_string, bool parse_cigar_string) { /* impl/paf.c:137-209 */
    size_t len = strlen(paf_string);
    while (len > 0 && paf_string[len - 1] == '\n') len--; /* one record: the line without its end */
    Buf b = {0};
    buf_str(&b, paf_string, len);
    buf_str(&b, "\n", 1);
    int64_t n = 0;
    Paf **res = parse_lines(b.p, (int64_t)b.n, parse_cigar_string, &n);
    free(b.p);
    Paf *p = n > 0 ? res[0] : NULL;
    for (int64_t i = 1; i < n; i++) paf_destruct(res[i]);
    free(res);
    return p;
}

Paf *paf_read_with_buffer(FILE *fh, bool parse_cigar_string, char **paf_buffer, int64_t *paf_length_buffer) { /* impl/paf.c:211-218 */
    size_t cap = (size_t)*paf_length_buffer;
    ssize_t got = getline(paf_buffer, &cap, fh);
    *paf_length_buffer = (int64_t)cap;
    if (got <= 0) return NULL; /* end of file */
    return paf_parse(*paf_buffer, parse_cigar_string);
}

Paf *paf_read(FILE *fh, bool parse_cigar_string) { /* impl/paf.c:220-226 */
    char *buf = NULL;
    int64_t len = 0;
    Paf *p = paf_read_with_buffer(fh, parse_cigar_string, &buf, &len);
    free(buf);
    return p;
}

Paf *paf_read2(FILE *fh) { return paf_read(fh, true); } /* impl/paf.c:228-230 */

char *paf_print(Paf *paf) { /* impl/paf.c:417-425: the line without its '\n' */
    Buf b = {0};
    hand_over(&b, paf, NULL, NULL);
    const paffy_stage pass = {PAFFY_PASS, 0.0f, 0.0f};
    int64_t out_len = 0;
    char *text = run_text(&pass, 1, b.p, (int64_t)b.n, &out_len);
    free(b.p);
    text = realloc(text, (size_t)out_len + 1);
    text[out_len > 0 ? out_len - 1 : 0] = '\0';
    return text;
}

void paf_write(Paf *paf, FILE *fh) { /* impl/paf.c:406-415 */
    char *s = paf_print(paf);
    fputs(s, fh);
    fputc('\n', fh);
    free(s);
}

void paf_write_with_buffer(Paf *paf, FILE *fh, char **paf_buffer, int64_t *paf_length_buffer) { /* impl/paf.c:396-404 */
    (void)paf_buffer;
    (void)paf_length_buffer; /* the line is built on the GPU; the caller's buffer is left as it is */
    paf_write(paf, fh);
}

void paf_check(Paf *paf) { /* impl/paf.c:427-461: the one call of this API that validates; the record is left as it is */
    const paffy_stage st = {PAFFY_CHECK, 0.0f, 0.0f};
    Buf b = {0};
    hand_over(&b, paf, NULL, NULL);
    int64_t out_len = 0;
    free(run_text(&st, 1, b.p, (int64_t)b.n, &out_len));
    free(b.p);
}

void paf_invert(Paf *paf) { /* impl/paf.c:463-490 */
    const paffy_stage st = {PAFFY_INVERT | PAFFY_NO_CHECK, 0.0f, 0.0f};
    transform(paf, &st, 1, NULL, NULL);
}

void paf_trim_ends(Paf *paf, int64_t end_bases_to_trim) { /* impl/paf.c:575-598 */
    paffy_stage st = paffy_stage_trim_ends(end_bases_to_trim);
    st.kind |= PAFFY_NO_CHECK;
    transform(paf, &st, 1, NULL, NULL);
}

void paf_trim_end_fraction(Paf *paf, float percentage) { /* impl/paf.c:586-598 */
    const paffy_stage st = {PAFFY_TRIM_FIXED | PAFFY_NO_CHECK, 0.0f, percentage};
    transform(paf, &st, 1, NULL, NULL);
}

void paf_trim_unreliable_tails(Paf *paf, float score_fraction, float max_fraction_to_trim) { /* impl/paf.c:929-953 */
    const paffy_stage st = {PAFFY_TRIM_IDENTITY | PAFFY_NO_CHECK, score_fraction, max_fraction_to_trim};
    transform(paf, &st, 1, NULL, NULL);
}

void paf_remove_mismatches(Paf *paf) { /* impl/paf.c:786-809 */
    const paffy_stage st = {PAFFY_REMOVE_MISMATCHES | PAFFY_NO_CHECK, 0.0f, 0.0f};
    transform(paf, &st, 1, NULL, NULL);
}

/*
 * The pair of sequences a call works on goes to the GPU under the stand-in names "Q" and "T" (the two strings are the record's own
 * sequences whatever it calls them). Callers walk a file of alignments over the same few sequences (impl/paf_view.c:150-172), so
 * the last pair stays loaded: it is sent again only when a pointer, a length or a hash over EVERY byte differs (the reference reads the
 * caller's strings on every call, impl/paf.c:752-757: a base edited in place between two calls must be seen; hashing costs about
 * as much as the strlen beside it and far less than the copy to the GPU it saves).
 */
static uint64_t hash_of(const char *s, int64_t len) {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)len;
    int64_t i = 0;
    for (; i + 8 <= len; i += 8) { /* eight bytes per step: multiply, fold the high half back in */
        uint64_t w;
        memcpy(&w, s + i, 8);
        h = (h ^ w) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
    }
    for (; i < len; i++) h = (h ^ (unsigned char)s[i]) * 1099511628211ull;
    return h ^ (h >> 32);
}
static void load_pair(char *query_seq, char *target_seq) {
    static const char *last[2];
    static int64_t last_len[2] = {-1, -1};
    static uint64_t last_sample[2];
    const char *names[2] = {"Q", "T"};
    const char *seqs[2] = {query_seq, target_seq};
    const int64_t lens[2] = {(int64_t)strlen(query_seq), (int64_t)strlen(target_seq)};
    const uint64_t smp[2] = {hash_of(query_seq, lens[0]), hash_of(target_seq, lens[1])};
    if (last[0] == seqs[0] && last[1] == seqs[1] && last_len[0] == lens[0] && last_len[1] == lens[1] && last_sample[0] == smp[0] &&
        last_sample[1] == smp[1])
        return;
    paffy_hip_keep_raw_sequences(ctx(), 1); /* paf_pretty_print shows the bases in their own case */
    if (paffy_hip_set_sequences(ctx(), 2, names, seqs, lens)) die("paffy_hip_set_sequences", paffy_hip_last_error(ctx()));
    for (int k = 0; k < 2; k++) {
        last[k] = seqs[k];
        last_len[k] = lens[k];
        last_sample[k] = smp[k];
    }
}

void paf_encode_mismatches(Paf *paf, char *query_seq, char *target_seq) { /* impl/paf.c:739-784 */
    load_pair(query_seq, target_seq);
    const paffy_stage st = {PAFFY_ADD_MISMATCHES | PAFFY_NO_CHECK, 0.0f, 0.0f};
    transform(paf, &st, 1, "Q", "T");
}

/* paf_stats_calc (impl/paf.c:236-260) of one record: a PAFFY_STATS plan, the six sums read back */
static void stats_of(Paf *paf, int64_t sums[6]) {
    for (int k = 0; k < 6; k++) sums[k] = 0;
    if (!paf->cigar) return; /* cigar_count(NULL) == 0 */
    Buf b = {0};
    hand_over(&b, paf, NULL, NULL);
    const paffy_stage st = {PAFFY_STATS, 0.0f, 0.0f};
    void *d_in = NULL;
    paffy_plan_info info;
    if (paffy_hip_malloc(&d_in, (int64_t)b.n + 64) || paffy_hip_memcpy_h2d(d_in, b.p, (int64_t)b.n) || paffy_hip_plan(ctx(), &st, 1, d_in, (int64_t)b.n, &info) ||
        paffy_hip_plan_stats(ctx(), sums))
        die("paf_stats_calc on MI355X", paffy_hip_last_error(ctx()));
    if (info.error.code) record_failure(&info);
    paffy_hip_free(d_in);
    free(b.p);
}

int64_t paf_get_number_of_aligned_bases(Paf *paf) { /* impl/paf.c:507-517: the bases of M, = and X ops */
    int64_t t[6];
    stats_of(paf, t);
    return t[0] + t[1];
}

void paf_stats_calc(Paf *paf, int64_t *matches, int64_t *mismatches, int64_t *query_inserts, int64_t *query_deletes,
                    int64_t *query_insert_bases, int64_t *query_delete_bases, bool zero_counts) {
    int64_t t[6];
    stats_of(paf, t);
    if (zero_counts) *matches = *mismatches = *query_inserts = *query_deletes = *query_insert_bases = *query_delete_bases = 0;
    *matches += t[0];
    *mismatches += t[1];
    *query_inserts += t[2];
    *query_deletes += t[3];
    *query_insert_bases += t[4];
    *query_delete_bases += t[5];
}

void paf_pretty_print(Paf *paf, char *query_seq, char *target_seq, FILE *fh, bool include_alignment) { /* impl/paf.c:269-316 */
    int64_t t[6] = {0, 0, 0, 0, 0, 0};
    void *d_in = NULL;
    Buf b = {0};
    if (paf->cigar) { /* cigar_count(NULL) == 0: all sums zero, no rows */
        if (include_alignment) load_pair(query_seq, target_seq);
        hand_over(&b, paf, include_alignment ? "Q" : NULL, include_alignment ? "T" : NULL);
        const paffy_stage st = {PAFFY_STATS, 0.0f, 0.0f};
        paffy_plan_info info;
        if (paffy_hip_malloc(&d_in, (int64_t)b.n + 64) || paffy_hip_memcpy_h2d(d_in, b.p, (int64_t)b.n) || paffy_hip_plan(ctx(), &st, 1, d_in, (int64_t)b.n, &info) ||
            paffy_hip_plan_stats(ctx(), t))
            die("paf_pretty_print on MI355X", paffy_hip_last_error(ctx()));
        if (info.error.code) record_failure(&info);
    }
    fprintf(fh, "Query:%s\tQ-start:%" PRIi64 "\tQ-length:%" PRIi64 "\tTarget:%s\tT-start:%" PRIi64 "\tT-length:%" PRIi64
                "\tSame-strand:%i\tScore:%" PRIi64 "\tIdentity:%f\tIdentity-with-gaps%f\tAligned-bases:%" PRIi64 "\tQuery-inserts:%" PRIi64
                "\tQuery-deletes:%" PRIi64 "\n",
            paf->query_name, paf->query_start, paf->query_end - paf->query_start, paf->target_name, paf->target_start,
            paf->target_end - paf->target_start, (int)paf->same_strand, paf->score, (float)t[0] / (t[0] + t[1]),
            (float)t[0] / (t[0] + t[1] + t[4] + t[5]), t[0] + t[1], t[2], t[3]);
    if (include_alignment && paf->cigar) { /* the three rows per 150 columns are written by the GPU (csrc/pretty_kernel.h) */
        int64_t off[2] = {0, 0};
        if (paffy_hip_plan_alignment_sizes(ctx(), 0, 1, &off[1])) die("paf_pretty_print on MI355X", paffy_hip_last_error(ctx()));
        char *rows = malloc((size_t)off[1] + 1);
        paffy_error e;
        if (!rows || paffy_hip_plan_alignment_rows(ctx(), 0, 1, off, rows, &e)) die("paf_pretty_print on MI355X", paffy_hip_last_error(ctx()));
        if (e.code) {
            paffy_plan_info bad;
            memset(&bad, 0, sizeof(bad));
            bad.error = e;
            record_failure(&bad);
        }
        fwrite(rows, 1, (size_t)off[1], fh);
        free(rows);
    }
    if (d_in) paffy_hip_free(d_in);
    free(b.p);
}

/* ---- coverage counters per sequence: what paf_tile and paf_to_bed keep (impl/paf.c:667-712) ---- */

void sequenceCountArray_destruct(SequenceCountArray *seq_count_array) { /* impl/paf.c:669-673 */
    free(seq_count_array->name);
    free(seq_count_array->counts);
    free(seq_count_array);
}

SequenceCountArray *get_alignment_count_array_in(SequenceCountArray ***arrays, int64_t *n_arrays, Paf *paf) { /* impl/paf.c:675-689 over a plain array */
    for (int64_t i = 0; i < *n_arrays; i++)
        if (strcmp((*arrays)[i]->name, paf->query_name) == 0) {
            if ((*arrays)[i]->length != paf->query_length) { /* the reference asserts that a name means one length */
                fprintf(stderr, "get_alignment_count_array: sequence %s seen with two lengths\n", paf->query_name);
                abort();
            }
            return (*arrays)[i];
        }
    SequenceCountArray *a = calloc(1, sizeof(SequenceCountArray));
    if (!a) die("out of memory", NULL);
    a->name = dup_slice(paf->query_name, 0, (uint32_t)strlen(paf->query_name));
    a->length = paf->query_length;
    a->counts = calloc(paf->query_length > 0 ? (size_t)paf->query_length : 1, sizeof(uint16_t));
    *arrays = realloc(*arrays, sizeof(SequenceCountArray *) * (size_t)(*n_arrays + 1));
    if (!a->counts || !*arrays) die("out of memory", NULL);
    (*arrays)[(*n_arrays)++] = a;
    return a;
}

/*
 * impl/paf.c:691-712. The record goes through the coverage engine of `paffy to_bed` (csrc/coverage_kernel.h: cigar walk, one
 * counter per query base, the same range checks) as a batch of one, and the engine's counters of [query_start, query_end) are added
 * to the caller's on the GPU, stopping at INT16_MAX - 1 like the reference's increments.
 */
void increase_alignment_level_counts(SequenceCountArray *seq_count_array, Paf *paf) {
    if (paf->query_length > seq_count_array->length) { /* assert(i + j < seq_count_array->length) */
        fprintf(stderr, "increase_alignment_level_counts: the record's query is longer than the count array\n");
        abort();
    }
    Buf b = {0};
    hand_over(&b, paf, NULL, NULL);
    const paffy_bed_opts opts = {0, 0, 0, 0, 1};
    paffy_plan_info info;
    void *d_in = NULL;
    if (paffy_hip_malloc(&d_in, (int64_t)b.n + 64) || paffy_hip_memcpy_h2d(d_in, b.p, (int64_t)b.n) || paffy_hip_bed_plan(ctx(), d_in, (int64_t)b.n, &opts, &info))
        die("increase_alignment_level_counts on MI355X", paffy_hip_last_error(ctx()));
    if (info.error.code) record_failure(&info);
    if (paf->query_end > paf->query_start &&
        paffy_hip_bed_counts(ctx(), 0, paf->query_start, paf->query_end, seq_count_array->counts + paf->query_start, 1))
        die("increase_alignment_level_counts on MI355X", paffy_hip_last_error(ctx()));
    paffy_hip_free(d_in);
    free(b.p);
}

/* ---- intervals of `paffy to_bed`-style fasta headers (impl/paf.c:714-737) ---- */

void interval_destruct(Interval *interval) {
    free(interval->name);
    free(interval);
}

/*
 * "name parts | length | start" -> Interval. fastaDecodeHeader / fastaEncodeHeader are sonLib's (absent here): the header is cut at
 * every '|', the last token is the start, the one before it the length, what is left is joined with '|' again.
 */
Interval *decode_fasta_header(char *fasta_header) {
    Interval *iv = calloc(1, sizeof(Interval));
    if (!iv) die("out of memory", NULL);
    size_t len = strlen(fasta_header);
    const char *cut[2] = {NULL, NULL}; /* the last two '|' */
    for (size_t i = len; i > 0 && !cut[1]; i--)
        if (fasta_header[i - 1] == '|') cut[cut[0] ? 1 : 0] = fasta_header + i - 1;
    if (!cut[1] || sscanf(cut[0] + 1, "%" SCNi64, &iv->start) != 1 || sscanf(cut[1] + 1, "%" SCNi64, &iv->length) != 1) {
        fprintf(stderr, "decode_fasta_header: %s does not end in |length|start\n", fasta_header);
        abort(); /* the reference's asserts */
    }
    iv->name = dup_slice(fasta_header, 0, (uint32_t)(cut[1] - fasta_header));
    return iv;
}

int cmp_intervals(const void *i, const void *j) {
    const Interval *x = (const Interval *)i, *y = (const Interval *)j;
    int k = strcmp(x->name, y->name);
    return k == 0 ? (x->start < y->start ? -1 : (x->start > y->start ? 1 : 0)) : k;
}

/* ---- whole files and lists: one batch per call ---- */

Paf **read_pafs_array(FILE *paf_file, bool parse_cigar_string, int64_t *n_pafs) { /* impl/paf.c:492-499 */
    Buf b = {0};
    char chunk[1 << 16];
    size_t got;
    while ((got = fread(chunk, 1, sizeof(chunk), paf_file)) > 0) buf_str(&b, chunk, got);
    Paf **res = parse_lines(b.p ? b.p : "", (int64_t)b.n, parse_cigar_string, n_pafs);
    free(b.p);
    return res;
}

void write_pafs_array(FILE *paf_file, Paf **pafs, int64_t n_pafs) { /* impl/paf.c:501-505 */
    if (n_pafs <= 0) return;
    Buf b = {0};
    for (int64_t i = 0; i < n_pafs; i++) hand_over(&b, pafs[i], NULL, NULL);
    const paffy_stage pass = {PAFFY_PASS, 0.0f, 0.0f};
    int64_t out_len = 0;
    char *text = run_text(&pass, 1, b.p, (int64_t)b.n, &out_len);
    fwrite(text, 1, (size_t)out_len, paf_file);
    free(text);
    free(b.p);
}

Paf **paf_shatter_array(Paf *paf, int64_t *n_pafs) { /* impl/paf.c:629-663 */
    const paffy_stage st = {PAFFY_SHATTER, 0.0f, 0.0f};
    Buf b = {0};
    hand_over(&b, paf, NULL, NULL);
    int64_t out_len = 0;
    char *text = run_text(&st, 1, b.p, (int64_t)b.n, &out_len);
    free(b.p);
    Paf **res = parse_lines(text ? text : "", out_len, true, n_pafs);
    free(text);
    return res;
}

/*
 * paf_chain (impl/chaining.c:266-343) over a plain array and with the gap cost of its one caller spelled out (impl/paf_chain.c:36-45:
 * no gap costs nothing, otherwise gap_open + gap_extend * (query gap + target gap)) -- a cost function cannot be handed to the GPU
 * as a C callback. The records go to the GPU as one batch; like the reference the same Paf objects come back, in a new array
 * ordered by descending score, with chain_id and chain_score set.
 */
Paf **paf_chain_array(Paf **pafs, int64_t n_pafs, int64_t gap_open, int64_t gap_extend, int64_t max_gap_length, float percentage_to_trim, int64_t *n_out) {
    *n_out = 0;
    Paf **out = malloc(sizeof(Paf *) * (size_t)(n_pafs > 0 ? n_pafs : 1));
    if (!out) die("out of memory", NULL);
    if (n_pafs <= 0) return out;
    Buf b = {0};
    for (int64_t i = 0; i < n_pafs; i++) hand_over(&b, pafs[i], NULL, NULL);
    const paffy_chain_opts opts = {gap_open, gap_extend, max_gap_length, percentage_to_trim};
    paffy_plan_info info;
    void *d_in = NULL;
    if (paffy_hip_malloc(&d_in, (int64_t)b.n + 64) || paffy_hip_memcpy_h2d(d_in, b.p, (int64_t)b.n) || paffy_hip_chain_begin(ctx()) ||
        paffy_hip_chain_add(ctx(), d_in, (int64_t)b.n) || paffy_hip_chain_run(ctx(), &opts, &info))
        die("paf_chain on MI355X", paffy_hip_last_error(ctx()));
    if (info.error.code) record_failure(&info);
    uint32_t *rows = malloc(sizeof(uint32_t) * (size_t)(n_pafs + 1));
    int64_t *offs = malloc(sizeof(int64_t) * (size_t)(n_pafs + 1)), *ids = malloc(sizeof(int64_t) * (size_t)n_pafs), *scores = malloc(sizeof(int64_t) * (size_t)n_pafs);
    if (!rows || !offs || !ids || !scores) die("out of memory", NULL);
    if (info.n_records != n_pafs || paffy_hip_plan_rows(ctx(), n_pafs + 1, rows, offs) != n_pafs || paffy_hip_chain_tags(ctx(), n_pafs, ids, scores) != n_pafs)
        die("paf_chain on MI355X", paffy_hip_last_error(ctx()));
    for (int64_t k = 0; k < n_pafs; k++) {
        out[k] = pafs[rows[k]];
        out[k]->chain_id = ids[k];
        out[k]->chain_score = scores[k];
    }
    *n_out = n_pafs;
    free(rows); free(offs); free(ids); free(scores);
    paffy_hip_free(d_in);
    free(b.p);
    return out;
}

#ifdef PAFFY_WITH_SONLIB
/* the stList form: the callback is probed and must be the affine cost above (it is, for the reference's only caller) */
stList *paf_chain(stList *pafs, int64_t (*gap_cost)(int64_t, int64_t, void *), void *gap_cost_params, int64_t max_gap_length, float percentage_to_trim) {
    const int64_t zero = gap_cost(0, 0, gap_cost_params), c1 = gap_cost(1, 0, gap_cost_params), c2 = gap_cost(2, 0, gap_cost_params);
    const int64_t extend = c2 - c1, open = c1 - extend;
    if (zero != 0 || gap_cost(0, 1, gap_cost_params) != c1 || gap_cost(1000, 234, gap_cost_params) != open + extend * 1234)
        die("paf_chain on MI355X", "the gap cost must be 0 for no gap and open + extend * (query gap + target gap) otherwise");
    int64_t n = stList_length(pafs), n_out = 0;
    Paf **a = malloc(sizeof(Paf *) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) a[i] = stList_get(pafs, i);
    Paf **res = paf_chain_array(a, n, open, extend, max_gap_length, percentage_to_trim, &n_out);
    stList *l = stList_construct3(0, (void (*)(void *))paf_destruct);
    for (int64_t i = 0; i < n_out; i++) stList_append(l, res[i]);
    free(a);
    free(res);
    return l;
}
stList *read_pafs(FILE *paf_file, bool parse_cigar_string) {
    int64_t n = 0;
    Paf **a = read_pafs_array(paf_file, parse_cigar_string, &n);
    stList *l = stList_construct3(0, (void (*)(void *))paf_destruct);
    for (int64_t i = 0; i < n; i++) stList_append(l, a[i]);
    free(a);
    return l;
}
void write_pafs(FILE *paf_file, stList *pafs) {
    int64_t n = stList_length(pafs);
    Paf **a = malloc(sizeof(Paf *) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) a[i] = stList_get(pafs, i);
    write_pafs_array(paf_file, a, n);
    free(a);
}
stList *paf_shatter(Paf *paf) {
    int64_t n = 0;
    Paf **a = paf_shatter_array(paf, &n);
    stList *l = stList_construct3(0, (void (*)(void *))paf_destruct);
    for (int64_t i = 0; i < n; i++) stList_append(l, a[i]);
    free(a);
    return l;
}
#endif
