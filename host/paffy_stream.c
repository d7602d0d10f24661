/*
 * paffy_stream.c -- chunked streaming of PAF text through the C-ABI (host side of the
 * shatter / invert / trim drivers; replaces the read-transform-write loop of
 * impl/paf_invert.c:84-89 and friends).
 */
#define _GNU_SOURCE
#include <inttypes.h>
#include <signal.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <fcntl.h>
#include <unistd.h>

#include "paffy_host.h"

/* ---- what the N-GPU launcher (host/paffy_launch.c) tells a worker through its environment ---- */
int host_device(void) {
    const char *e = getenv("PAFFY_DEVICE");
    return e && *e ? atoi(e) : -1;
}
typedef struct {
    int fd;
    int64_t pos, end;
} range_cookie;
static ssize_t range_read(void *c, char *buf, size_t n) {
    range_cookie *r = (range_cookie *)c;
    const int64_t left = r->end - r->pos;
    if (left <= 0) return 0;
    if ((int64_t)n > left) n = (size_t)left;
    const ssize_t got = pread(r->fd, buf, n, (off_t)r->pos);
    if (got > 0) r->pos += got;
    return got;
}
static int range_close(void *c) {
    range_cookie *r = (range_cookie *)c;
    close(r->fd);
    free(r);
    return 0;
}
FILE *host_open_input(const char *path) {
    if (!path) return stdin;
    const char *rg = getenv("PAFFY_RANGE");
    long long a = 0, b = 0;
    if (!rg || sscanf(rg, "%lld:%lld", &a, &b) != 2) return fopen(path, "r");
    range_cookie *r = (range_cookie *)malloc(sizeof(range_cookie));
    if (!r) return NULL;
    r->fd = open(path, O_RDONLY);
    r->pos = a;
    r->end = b;
    if (r->fd < 0) {
        free(r);
        return NULL;
    }
    cookie_io_functions_t io = {range_read, NULL, NULL, range_close};
    FILE *fh = fopencookie(r, "r", io);
    if (fh) setvbuf(fh, NULL, _IOFBF, 1 << 22);
    return fh;
}

static int g_log_level = 0;
static const char *const *g_seq_names, *const *g_seq_data;
static const int64_t *g_seq_lens;
static int64_t g_seq_n = 0;

void host_set_sequences(const char *const *names, const char *const *seqs, const int64_t *lens, int64_t n) {
    g_seq_names = names;
    g_seq_data = seqs;
    g_seq_lens = lens;
    g_seq_n = n;
}

static int g_keep_raw = 0;
void host_keep_raw_sequences(int on) { g_keep_raw = on; }

void host_set_log_level(const char *s) {
    g_log_level = 0;
    if (!s) return;
    if (!strcasecmp(s, "INFO")) g_log_level = 1;
    else if (!strcasecmp(s, "DEBUG")) g_log_level = 2;
}

void host_log_info(const char *fmt, ...) {
    if (g_log_level < 1) return;
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
}

static size_t chunk_bytes(void) {
    const char *e = getenv("PAFFY_CHUNK_MB");
    long mb = e ? atol(e) : 256;
    if (mb < 1) mb = 1;
    if (mb > 1900) mb = 1900; /* one batch stays below 2 GiB */
    return (size_t)mb << 20;
}

/* Ends the process the way the reference would for this record error. */
static void die_like_reference(const paffy_error *e, int64_t record_base) {
    int status = paffy_hip_error_exit_status(e->code);
    fflush(stdout);
    if (e->code == PAFFY_ERR_STRAND)
        fprintf(stderr, "Got an unexpected strand character (%c) in a paf string\n", (int)e->aux);
    else if (e->code == PAFFY_ERR_CIGAR_CHAR)
        fprintf(stderr, "Got an unexpected character paf cigar string: %c\n", (int)e->aux);
    else if (e->code == PAFFY_ERR_MISSING_QUERY_SEQ)
        fprintf(stderr, "No query sequence found for record %lld\n", (long long)(record_base + e->record));
    else if (e->code == PAFFY_ERR_MISSING_TARGET_SEQ)
        fprintf(stderr, "No target sequence found for record %lld\n", (long long)(record_base + e->record));
    else
        fprintf(stderr, "%s (record %lld)\n", paffy_hip_error_string(e->code), (long long)(record_base + e->record));
    if (status == 134) raise(SIGABRT);
    if (status == 139) raise(SIGSEGV);
    exit(status ? status : 1);
}

static paffy_filter g_filter = {-1, -1, -1.0, -1.0, -1, 0};
void host_set_filter(const paffy_filter *f) { g_filter = *f; }

/* paffy dedupe: the chunks of the stream go through paffy_hip_dedupe_plan of one context, which remembers what it wrote */
static int g_dedupe_mode = 0; /* 0: stage list, 1: dedupe, 2: dedupe -a */
void host_set_dedupe(int check_inverse) { g_dedupe_mode = check_inverse ? 2 : 1; }
static int dedupe_chunk(paffy_hip_ctx *ctx, const char *h_in, int64_t in_len, char **h_out, int64_t *out_len, paffy_plan_info *info) {
    void *d_in = NULL, *d_out = NULL;
    int rc = -1;
    *h_out = NULL;
    *out_len = 0;
    if (paffy_hip_malloc(&d_in, in_len + 64) == 0 && paffy_hip_memcpy_h2d(d_in, h_in, in_len) == 0 &&
        paffy_hip_dedupe_plan(ctx, d_in, in_len, g_dedupe_mode == 2, info) == 0) {
        rc = 0;
        if (info->out_bytes > 0) {
            *h_out = (char *)malloc((size_t)info->out_bytes);
            if (paffy_hip_malloc(&d_out, info->out_bytes + 64) == 0 && paffy_hip_emit(ctx, d_out, info->out_bytes + 64) == 0 &&
                paffy_hip_sync(ctx) == 0 && paffy_hip_memcpy_d2h(*h_out, d_out, info->out_bytes) == 0)
                *out_len = info->out_bytes;
            else
                rc = -1;
        }
    }
    if (d_in) paffy_hip_free(d_in);
    if (d_out) paffy_hip_free(d_out);
    return rc;
}

/* paffy view -s -t: nothing is written per record; the PAFFY_STATS sums of every chunk are added up */
static int g_stats_mode = 0;
static int64_t g_stats[6], g_stats_records;
void host_set_stats(int on) {
    g_stats_mode = on;
    memset(g_stats, 0, sizeof(g_stats));
    g_stats_records = 0;
}
void host_get_stats(int64_t sums[6], int64_t *n_records) {
    memcpy(sums, g_stats, sizeof(g_stats));
    *n_records = g_stats_records;
}
static FILE *g_stats_lines = NULL; /* paffy view without -t: one paf_pretty_print stats line per record goes here */
void host_set_stats_lines(FILE *fh) { g_stats_lines = fh; }

static int g_alignment_rows = 0; /* paffy view -a: the base-level rows under every stats line */
void host_set_alignment_rows(int on) { g_alignment_rows = on; }

/*
 * paf_pretty_print per record (impl/paf.c:269-315): the stats line -- the six sums come from the GPU's PAFFY_STATS stage, the
 * fields from the GPU's parse -- and under -a the base-level rows, written by the GPU (paffy_hip_plan_alignment_rows) and fetched
 * in pieces of at most 128 MiB.
 */
static int stats_lines(paffy_hip_ctx *ctx, const char *buf, const paffy_record *recs, int64_t n_records, int64_t record_base) {
    int64_t *t = (int64_t *)malloc(sizeof(int64_t) * 6 * (size_t)(n_records > 0 ? n_records : 1));
    int64_t *off = NULL;
    char *rows = NULL;
    int64_t rows_first = 0, rows_end = 0; /* records [rows_first, rows_end) are in `rows` */
    int rc = paffy_hip_plan_record_stats(ctx, n_records, t) == n_records ? 0 : PAFFY_E_STATE;
    if (!rc && g_alignment_rows) {
        off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_records + 1));
        rc = paffy_hip_plan_alignment_sizes(ctx, 0, n_records, off);
        if (!rc) { /* sizes -> running sums */
            int64_t at = 0;
            for (int64_t r = 0; r < n_records; r++) {
                int64_t b = off[r];
                off[r] = at;
                at += b;
            }
            off[n_records] = at;
        }
    }
    for (int64_t r = 0; r < n_records && !rc; r++) {
        const paffy_record *p = &recs[r];
        const int64_t *s = t + 6 * r; /* matches, mismatches, inserts, deletes, insert bases, delete bases */
        fprintf(g_stats_lines, "Query:%.*s\tQ-start:%" PRIi64 "\tQ-length:%" PRIi64 "\tTarget:%.*s\tT-start:%" PRIi64 "\tT-length:%" PRIi64
                "\tSame-strand:%i\tScore:%" PRIi64 "\tIdentity:%f\tIdentity-with-gaps%f\tAligned-bases:%" PRIi64 "\tQuery-inserts:%" PRIi64
                "\tQuery-deletes:%" PRIi64 "\n",
                (int)p->query_name_len, buf + p->query_name_off, p->query_start, p->query_end - p->query_start, (int)p->target_name_len,
                buf + p->target_name_off, p->target_start, p->target_end - p->target_start, (int)p->same_strand, p->score,
                (float)s[0] / (s[0] + s[1]), (float)s[0] / (s[0] + s[1] + s[4] + s[5]), s[0] + s[1], s[2], s[3]);
        if (!g_alignment_rows) continue;
        if (r >= rows_end) { /* the next piece: as many records as fit 128 MiB (at least one) */
            rows_first = r;
            rows_end = r + 1;
            while (rows_end < n_records && off[rows_end + 1] - off[rows_first] <= ((int64_t)128 << 20)) rows_end++;
            free(rows);
            rows = (char *)malloc((size_t)(off[rows_end] - off[rows_first]) + 1);
            paffy_error e;
            rc = paffy_hip_plan_alignment_rows(ctx, rows_first, rows_end - rows_first, off + rows_first, rows, &e);
            if (!rc && e.code) {
                fflush(g_stats_lines);
                e.record += rows_first;
                die_like_reference(&e, record_base);
            }
            if (rc) break;
        }
        fwrite(rows + (off[r] - off[rows_first]), 1, (size_t)(off[r + 1] - off[r]), g_stats_lines);
    }
    free(rows);
    free(off);
    free(t);
    return rc;
}

static int stats_chunk(paffy_hip_ctx *ctx, const paffy_stage *stages, int n_stages, const char *buf, int64_t len, int64_t record_base,
                       paffy_plan_info *info) {
    /* the fields of the stats lines first: parsing is a plan of its own and the stage plan must be the current one afterwards */
    paffy_record *recs = NULL;
    uint64_t *ops = NULL;
    int64_t n_ops = 0;
    if (g_stats_lines) {
        int rc0 = paffy_hip_parse_host(ctx, buf, len, &recs, &ops, &n_ops, info);
        free(ops);
        if (rc0 || info->error.code) {
            free(recs);
            return rc0;
        }
    }
    void *d_in = NULL;
    if (paffy_hip_malloc(&d_in, len + 64) != 0) {
        free(recs);
        return PAFFY_E_HIP;
    }
    int rc = paffy_hip_memcpy_h2d(d_in, buf, len);
    if (!rc) rc = paffy_hip_plan(ctx, stages, n_stages, d_in, len, info);
    int64_t sums[6];
    if (!rc) rc = paffy_hip_plan_stats(ctx, sums);
    if (!rc && info->error.code == 0) {
        for (int k = 0; k < 6; k++) g_stats[k] += sums[k];
        g_stats_records += info->n_records;
        if (g_stats_lines) rc = stats_lines(ctx, buf, recs, info->n_records, record_base);
    }
    paffy_hip_free(d_in);
    free(recs);
    return rc;
}

/* every piece of the oldest submitted chunk to `out` */
static int drain_chunk(paffy_hip_stream *st, FILE *out) {
    for (;;) {
        const char *piece = NULL;
        int64_t len = 0;
        if (paffy_hip_stream_read(st, &piece, &len) != 0) return 1;
        if (len == 0) return 0;
        if (fwrite(piece, 1, (size_t)len, out) != (size_t)len) return 1;
    }
}

/*
 * The stream commands (impl/paf_invert.c:84-89 and friends): chunks of whole lines go through pinned buffers; while the GPU works on
 * chunk k + 1 (copy in, sizing, line writer) the host drains the output of chunk k piece by piece into `out`.
 */
static int pipelined_stream(paffy_hip_ctx *ctx, const paffy_stage *stages, int n_stages, FILE *in, FILE *out) {
    const int64_t cap0 = (int64_t)chunk_bytes();
    paffy_hip_stream *st = NULL;
    if (paffy_hip_stream_open(ctx, stages, n_stages, cap0, (int64_t)64 << 20, &st) != 0) {
        fprintf(stderr, "paffy: could not set up the streaming buffers: %s\n", paffy_hip_last_error(ctx));
        return 1;
    }
    const char *carry = NULL; /* the partial last line of the chunk before, still in that chunk's buffer */
    int64_t carry_len = 0, records_done = 0;
    int eof = 0, rc = 0, pending = 0;
    paffy_plan_info info;
    memset(&info, 0, sizeof(info));
    while (!rc && (!eof || carry_len > 0)) {
        int64_t cap = 0;
        char *buf = paffy_hip_stream_input(st, cap0, 0, &cap);
        if (!buf) {
            rc = 1;
            break;
        }
        int64_t have = carry_len;
        if (carry_len > cap) buf = paffy_hip_stream_input(st, carry_len * 2, 0, &cap);
        if (!buf) {
            rc = 1;
            break;
        }
        if (carry_len) memcpy(buf, carry, (size_t)carry_len);
        int64_t use = 0;
        for (;;) {
            if (!eof && have < cap) {
                size_t got = fread(buf + have, 1, (size_t)(cap - have), in);
                have += (int64_t)got;
                if ((int64_t)got < cap - have + (int64_t)got) eof = feof(in) || ferror(in);
            }
            use = have;
            if (!eof) /* keep the partial last line for the next chunk */
                while (use > 0 && buf[use - 1] != '\n') use--;
            if (use > 0 || eof) break;
            buf = paffy_hip_stream_input(st, cap * 2, have, &cap); /* a single line longer than the buffer: grow */
            if (!buf) {
                fprintf(stderr, "paffy: a line too long for one batch (2 GiB)\n");
                rc = 1;
                break;
            }
        }
        if (rc || use == 0) break;
        if (paffy_hip_stream_submit(st, use, &info) != 0) {
            fprintf(stderr, "paffy: GPU call failed: %s\n", paffy_hip_last_error(ctx));
            rc = 1;
            break;
        }
        carry = buf + use; /* stays where it is until this slot is filled again, two chunks from now */
        carry_len = have - use;
        if (pending && drain_chunk(st, out)) rc = 1; /* the chunk before, while the GPU works on this one */
        pending = 1;
        if (info.error.code) { /* everything before the failing record is written, then the process ends like the reference */
            if (!rc) rc = drain_chunk(st, out);
            fflush(out);
            die_like_reference(&info.error, records_done);
        }
        records_done += info.n_records;
    }
    if (!rc && pending && drain_chunk(st, out)) rc = 1;
    if (rc && ctx) fprintf(stderr, "paffy: streaming failed: %s\n", paffy_hip_last_error(ctx));
    paffy_hip_stream_close(st);
    return rc;
}

int host_stream(const paffy_stage *stages, int n_stages, FILE *in, FILE *out) {
    paffy_hip_ctx *ctx = NULL;
    if (paffy_hip_create(&ctx, host_device()) != 0) {
        fprintf(stderr, "paffy: no usable GPU (this build has no CPU path)\n");
        return 1;
    }
    paffy_hip_set_filter(ctx, &g_filter);
    if (g_keep_raw) paffy_hip_keep_raw_sequences(ctx, 1);
    if (g_seq_n > 0 && paffy_hip_set_sequences(ctx, g_seq_n, g_seq_names, g_seq_data, g_seq_lens) != 0) {
        fprintf(stderr, "paffy: could not load the sequences onto the GPU: %s\n", paffy_hip_last_error(ctx));
        return 1;
    }
    if (!g_dedupe_mode && !g_stats_mode) {
        int rc = pipelined_stream(ctx, stages, n_stages, in, out);
        paffy_hip_destroy(ctx);
        fflush(out);
        return rc;
    }
    const size_t cap = chunk_bytes();
    size_t buf_cap = cap + (1 << 20), have = 0;
    char *buf = (char *)malloc(buf_cap);
    int64_t records_done = 0;
    int eof = 0, rc = 0;
    while (!eof || have > 0) {
        if (!eof) {
            if (have == buf_cap) { /* a single line longer than the chunk: grow */
                buf_cap *= 2;
                buf = (char *)realloc(buf, buf_cap);
            }
            size_t want = (have < cap ? cap : buf_cap) - have;
            size_t got = fread(buf + have, 1, want, in);
            have += got;
            if (got < want) eof = 1;
        }
        size_t use = have;
        if (!eof) { /* keep the partial last line for the next chunk */
            while (use > 0 && buf[use - 1] != '\n') use--;
            if (use == 0) continue; /* no complete line yet: read more */
        }
        if (use == 0) break;
        char *h_out = NULL;
        int64_t out_len = 0;
        paffy_plan_info info;
        int r = g_dedupe_mode ? dedupe_chunk(ctx, buf, (int64_t)use, &h_out, &out_len, &info) : stats_chunk(ctx, stages, n_stages, buf, (int64_t)use, records_done, &info);
        if (r != 0) {
            fprintf(stderr, "paffy: GPU call failed (%d): %s\n", r, paffy_hip_last_error(ctx));
            rc = 1;
            free(h_out);
            break;
        }
        if (out_len > 0) fwrite(h_out, 1, (size_t)out_len, out);
        free(h_out);
        if (info.error.code) die_like_reference(&info.error, records_done);
        records_done += info.n_records;
        memmove(buf, buf + use, have - use);
        have -= use;
    }
    free(buf);
    paffy_hip_destroy(ctx);
    fflush(out);
    return rc;
}

/* ---- paffy split_file (impl/paf_split_file.c:131-173): the GPU normalises the lines, the host routes them ---- */
typedef struct { char *name; size_t len; FILE *fh; } split_slot;
typedef struct { split_slot *v; size_t n, cap; } split_map;
static FILE *split_find(const split_map *m, const char *name, size_t len) {
    for (size_t i = 0; i < m->n; i++)
        if (m->v[i].len == len && memcmp(m->v[i].name, name, len) == 0) return m->v[i].fh;
    return NULL;
}
static void split_add(split_map *m, const char *name, size_t len, FILE *fh) {
    if (m->n == m->cap) {
        m->cap = m->cap ? m->cap * 2 : 64;
        m->v = (split_slot *)realloc(m->v, sizeof(split_slot) * m->cap);
    }
    m->v[m->n].name = (char *)malloc(len + 1);
    memcpy(m->v[m->n].name, name, len);
    m->v[m->n].len = len;
    m->v[m->n].fh = fh;
    m->n++;
}
/* field `want` (0-based) of a PAF line split the way strtok_r(.., "\t") splits it (runs of tabs are one separator) */
static int line_field(const char *p, const char *e, int want, const char **fs, size_t *fl) {
    int f = 0;
    while (p < e) {
        while (p < e && *p == '\t') p++;
        if (p >= e) break;
        const char *q = p;
        while (q < e && *q != '\t') q++;
        if (f == want) {
            *fs = p;
            *fl = (size_t)(q - p);
            return 1;
        }
        f++;
        p = q;
    }
    return 0;
}
static FILE *open_or_die(const char *path) {
    FILE *fh = fopen(path, "w");
    if (!fh) {
        fprintf(stderr, "Could not open output file: %s\n", path);
        exit(1);
    }
    host_log_info("Opened output file: %s\n", path);
    return fh;
}

int host_split_file(FILE *in, const char *prefix, int by_query, int64_t min_length) {
    paffy_hip_ctx *ctx = NULL;
    if (paffy_hip_create(&ctx, host_device()) != 0) {
        fprintf(stderr, "paffy: no usable GPU (this build has no CPU path)\n");
        return 1;
    }
    split_map big = {0, 0, 0}, small = {0, 0, 0};
    FILE **small_files = NULL;
    size_t n_small_files = 0;
    FILE *current = NULL;
    int64_t current_len = 0, records_done = 0;
    const size_t cap = chunk_bytes();
    size_t buf_cap = cap + (1 << 20), have = 0;
    char *buf = (char *)malloc(buf_cap);
    int eof = 0, rc = 0;
    while (!eof || have > 0) {
        if (!eof) {
            if (have == buf_cap) {
                buf_cap *= 2;
                buf = (char *)realloc(buf, buf_cap);
            }
            size_t want = (have < cap ? cap : buf_cap) - have;
            size_t got = fread(buf + have, 1, want, in);
            have += got;
            if (got < want) eof = 1;
        }
        size_t use = have;
        if (!eof) {
            while (use > 0 && buf[use - 1] != '\n') use--;
            if (use == 0) continue;
        }
        if (use == 0) break;
        void *d_in = NULL, *d_out = NULL;
        paffy_plan_info info;
        char *h_out = NULL;
        uint32_t *rows = NULL;
        int64_t *offs = NULL;
        int64_t n_rows = 0;
        int ok = paffy_hip_malloc(&d_in, (int64_t)use + 64) == 0 && paffy_hip_memcpy_h2d(d_in, buf, (int64_t)use) == 0 &&
                 paffy_hip_dedupe_plan(ctx, d_in, (int64_t)use, PAFFY_DEDUPE_KEEP_ALL, &info) == 0;
        if (ok && info.out_bytes > 0) {
            h_out = (char *)malloc((size_t)info.out_bytes);
            rows = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(info.n_rows + 1));
            offs = (int64_t *)malloc(sizeof(int64_t) * (size_t)(info.n_rows + 1));
            ok = paffy_hip_malloc(&d_out, info.out_bytes + 64) == 0 && paffy_hip_emit(ctx, d_out, info.out_bytes + 64) == 0 &&
                 paffy_hip_sync(ctx) == 0 && paffy_hip_memcpy_d2h(h_out, d_out, info.out_bytes) == 0 &&
                 (n_rows = paffy_hip_plan_rows(ctx, info.n_rows + 1, rows, offs)) >= 0;
        }
        if (!ok) {
            fprintf(stderr, "paffy split_file: GPU call failed: %s\n", paffy_hip_last_error(ctx));
            rc = 1;
        } else {
            /* line starts of the chunk, then every written line to the file of its contig */
            const char *p = buf, *e = buf + use;
            int64_t line = 0, k = 0;
            while (p < e && k < n_rows) {
                const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
                const char *le = nl ? nl : e;
                if ((int64_t)rows[k] == line) {
                    const char *name = NULL, *lens = NULL;
                    size_t name_len = 0, lens_len = 0;
                    line_field(p, le, by_query ? 0 : 5, &name, &name_len);
                    line_field(p, le, by_query ? 1 : 6, &lens, &lens_len);
                    int64_t contig_len = 0; /* str_to_int64, impl/paf.c:37-48 */
                    {
                        size_t i = 0;
                        int neg = 0;
                        if (i < lens_len && lens[i] == '-') { neg = 1; i++; }
                        uint64_t v = 0;
                        for (; i < lens_len && lens[i] >= '0' && lens[i] <= '9'; i++) v = v * 10 + (uint64_t)(lens[i] - '0');
                        contig_len = neg ? -(int64_t)v : (int64_t)v;
                    }
                    FILE *fh;
                    if (min_length > 0 && contig_len < min_length) {
                        fh = split_find(&small, name, name_len);
                        if (!fh) {
                            if (!current || current_len + contig_len > min_length) {
                                char path[4096];
                                snprintf(path, sizeof(path), "%ssmall_%lld.paf", prefix, (long long)n_small_files);
                                current = open_or_die(path);
                                small_files = (FILE **)realloc(small_files, sizeof(FILE *) * (n_small_files + 1));
                                small_files[n_small_files++] = current;
                                current_len = 0;
                            }
                            current_len += contig_len;
                            split_add(&small, name, name_len, current);
                            fh = current;
                        }
                    } else {
                        fh = split_find(&big, name, name_len);
                        if (!fh) {
                            char path[4096];
                            int w = snprintf(path, sizeof(path), "%s", prefix);
                            for (size_t i = 0; i < name_len && w < (int)sizeof(path) - 8; i++) path[w++] = name[i] == '/' ? '_' : name[i];
                            snprintf(path + w, sizeof(path) - (size_t)w, ".paf");
                            fh = open_or_die(path);
                            split_add(&big, name, name_len, fh);
                        }
                    }
                    fwrite(h_out + offs[k], 1, (size_t)(offs[k + 1] - offs[k]), fh);
                    k++;
                }
                line++;
                p = nl ? nl + 1 : e;
            }
        }
        if (d_in) paffy_hip_free(d_in);
        if (d_out) paffy_hip_free(d_out);
        free(h_out);
        free(rows);
        free(offs);
        if (rc) break;
        if (info.error.code) {
            for (size_t i = 0; i < big.n; i++) fclose(big.v[i].fh);
            for (size_t i = 0; i < n_small_files; i++) fclose(small_files[i]);
            die_like_reference(&info.error, records_done);
        }
        records_done += info.n_records;
        memmove(buf, buf + use, have - use);
        have -= use;
    }
    for (size_t i = 0; i < big.n; i++) { fclose(big.v[i].fh); free(big.v[i].name); }
    for (size_t i = 0; i < small.n; i++) free(small.v[i].name);
    for (size_t i = 0; i < n_small_files; i++) fclose(small_files[i]);
    free(big.v); free(small.v); free(small_files);
    free(buf);
    paffy_hip_destroy(ctx);
    host_log_info("Split %lld records\n", (long long)records_done);
    return rc;
}

static int whole_file(FILE *in, FILE *out, const paffy_bed_opts *bed, const paffy_chain_opts *chain);
int host_tile(FILE *in, FILE *out) { return whole_file(in, out, NULL, NULL); }
/* paffy to_bed: every record counts before anything is written, like tile */
int host_to_bed(FILE *in, FILE *out, const paffy_bed_opts *opts) { return whole_file(in, out, opts, NULL); }
/* paffy chain: read_pafs, paf_chain, write_pafs (impl/paf_chain.c:123-127) */
int host_chain(FILE *in, FILE *out, const paffy_chain_opts *opts) { return whole_file(in, out, NULL, opts); }

/*
 * tile / to_bed / chain read the whole input before they write (read_pafs, impl/paf.c:492-499; the loop of impl/paf_to_bed.c:166-190).
 * The text goes to the GPU in batches of whole lines (at most PAFFY_CHUNK_MB each, below the 2 GiB a batch may hold) and stays
 * there; the output comes back through a bounded staging buffer. Inputs are limited by the GPU's memory, not by a batch.
 */
static int whole_file(FILE *in, FILE *out, const paffy_bed_opts *bed, const paffy_chain_opts *chain) {
    paffy_hip_ctx *ctx = NULL;
    if (paffy_hip_create(&ctx, host_device()) != 0) {
        fprintf(stderr, "paffy: no usable GPU (this build has no CPU path)\n");
        return 1;
    }
    const char *what = bed ? "to_bed" : (chain ? "chain" : "tile");
    const size_t cap = chunk_bytes();
    size_t buf_cap = cap + (1 << 20), have = 0;
    char *buf = (char *)malloc(buf_cap);
    void **d_batches = NULL;
    size_t n_batches = 0;
    int eof = 0, rc = 0;
    if (!buf) {
        fprintf(stderr, "paffy %s: out of memory\n", what);
        return 1;
    }
    if ((bed ? paffy_hip_bed_begin(ctx, bed) : (chain ? paffy_hip_chain_begin(ctx) : paffy_hip_tile_begin(ctx))) != 0) rc = 1;
    while (!rc && (!eof || have > 0)) {
        if (!eof) {
            if (have == buf_cap) { /* a single line longer than the chunk: grow */
                char *nb = (char *)realloc(buf, buf_cap * 2);
                if (!nb) {
                    fprintf(stderr, "paffy %s: out of memory\n", what);
                    rc = 1;
                    break;
                }
                buf = nb;
                buf_cap *= 2;
            }
            size_t want = (have < cap ? cap : buf_cap) - have;
            size_t got = fread(buf + have, 1, want, in);
            have += got;
            if (got < want) eof = 1;
        }
        size_t use = have;
        if (!eof) {
            while (use > 0 && buf[use - 1] != '\n') use--;
            if (use == 0) continue;
        }
        if (use == 0) break;
        if (use >= ((size_t)1 << 31) - 64) {
            fprintf(stderr, "paffy %s: a single line of 2 GiB or more\n", what);
            rc = 1;
            break;
        }
        void *d = NULL;
        if (paffy_hip_malloc(&d, (int64_t)use + 64) != 0 || paffy_hip_memcpy_h2d(d, buf, (int64_t)use) != 0 ||
            (bed ? paffy_hip_bed_add(ctx, d, (int64_t)use) : (chain ? paffy_hip_chain_add(ctx, d, (int64_t)use) : paffy_hip_tile_add(ctx, d, (int64_t)use))) != 0) {
            fprintf(stderr, "paffy %s: GPU call failed: %s (the input must fit the GPU's memory)\n", what, paffy_hip_last_error(ctx));
            if (d) paffy_hip_free(d);
            rc = 1;
            break;
        }
        d_batches = (void **)realloc(d_batches, sizeof(void *) * (n_batches + 1));
        d_batches[n_batches++] = d;
        memmove(buf, buf + use, have - use);
        have -= use;
    }
    free(buf);
    paffy_plan_info info;
    memset(&info, 0, sizeof(info));
    if (!rc && (bed ? paffy_hip_bed_run(ctx, bed, &info) : (chain ? paffy_hip_chain_run(ctx, chain, &info) : paffy_hip_tile_run(ctx, &info))) != 0) {
        fprintf(stderr, "paffy %s: GPU call failed: %s\n", what, paffy_hip_last_error(ctx));
        rc = 1;
    }
    if (!rc && info.error.code) die_like_reference(&info.error, 0);
    if (!rc && info.out_bytes <= 0 && !bed && !chain && getenv("PAFFY_ROWS_FILE")) { /* under the N-GPU launcher: no output line, an empty list */
        FILE *rf = fopen(getenv("PAFFY_ROWS_FILE"), "w");
        if (!rf || fclose(rf) != 0) {
            fprintf(stderr, "paffy %s: cannot write %s\n", what, getenv("PAFFY_ROWS_FILE"));
            rc = 1;
        }
    }
    if (!rc && info.out_bytes > 0) {
        if (bed) { /* the runs: one buffer */
            void *d_out = NULL;
            char *h = (char *)malloc((size_t)info.out_bytes);
            if (h && paffy_hip_malloc(&d_out, info.out_bytes + 64) == 0 && paffy_hip_emit(ctx, d_out, info.out_bytes + 64) == 0 && paffy_hip_sync(ctx) == 0 &&
                paffy_hip_memcpy_d2h(h, d_out, info.out_bytes) == 0)
                fwrite(h, 1, (size_t)info.out_bytes, out);
            else
                rc = 1;
            free(h);
            if (d_out) paffy_hip_free(d_out);
        } else { /* the lines, as many at a time as fit the staging buffer */
            const int64_t n = info.n_rows;
            uint32_t *rows = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n + 1));
            int64_t *offs = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
            int64_t stage = (int64_t)256 << 20;
            if (!rows || !offs || paffy_hip_plan_rows(ctx, n + 1, rows, offs) != n) rc = 1;
            if (!rc && getenv("PAFFY_ROWS_FILE")) { /* under the N-GPU launcher: the input record of every output line, for its merge */
                FILE *rf = fopen(getenv("PAFFY_ROWS_FILE"), "w");
                if (!rf || fwrite(rows, sizeof(uint32_t), (size_t)n, rf) != (size_t)n || fclose(rf) != 0) {
                    fprintf(stderr, "paffy %s: cannot write %s\n", what, getenv("PAFFY_ROWS_FILE"));
                    rc = 1;
                }
            }
            for (int64_t k = 0; !rc && k < n; k++)
                if (offs[k + 1] - offs[k] > stage) stage = offs[k + 1] - offs[k];
            void *d_stage = NULL;
            char *h = rc ? NULL : (char *)malloc((size_t)stage);
            if (!rc && (!h || paffy_hip_malloc(&d_stage, stage + 64) != 0)) rc = 1;
            int64_t first = 0;
            while (!rc && first < n) {
                int64_t last = first + 1;
                while (last < n && offs[last + 1] - offs[first] <= stage) last++;
                int64_t bytes = 0;
                if (paffy_hip_emit_lines(ctx, first, last - first, d_stage, stage + 64, &bytes) != 0 || paffy_hip_sync(ctx) != 0 ||
                    paffy_hip_memcpy_d2h(h, d_stage, bytes) != 0)
                    rc = 1;
                else
                    fwrite(h, 1, (size_t)bytes, out);
                first = last;
            }
            free(rows);
            free(offs);
            free(h);
            if (d_stage) paffy_hip_free(d_stage);
        }
        if (rc) fprintf(stderr, "paffy %s: GPU call failed: %s\n", what, paffy_hip_last_error(ctx));
    }
    for (size_t i = 0; i < n_batches; i++) paffy_hip_free(d_batches[i]);
    free(d_batches);
    paffy_hip_destroy(ctx);
    fflush(out);
    return rc;
}
