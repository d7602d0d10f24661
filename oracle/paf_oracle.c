/*
 * paf_oracle.c -- TEST INFRASTRUCTURE ONLY (see paf_oracle.h).
 *
 * Scalar CPU restatement of the reference paffy hot path. Each function names
 * the reference lines whose behaviour it restates (paths relative to
 * /root/reference). Nothing here is shipped or called by the product path.
 *
 * Data model: one record = fixed numeric fields + two name slices that point
 * into the input buffer + a window [lo, lo+n) over an array of (length, op)
 * pairs. Lengths are kept in the reference's 56-bit signed range (wrap56).
 */
#include "paf_oracle.h"

#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* small utilities                                                      */
/* ------------------------------------------------------------------ */

enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_EQ = 3, OP_X = 4 }; /* inc/paf.h:52-58 */

typedef struct {
    int64_t len;
    int32_t op;
} oop;

/* CigarRecord.length is a 56-bit signed bitfield (inc/paf.h:61-64). */
static inline int64_t wrap56(int64_t v) { return (int64_t)((uint64_t)v << 8) >> 8; }

typedef struct {
    char *p;
    int64_t n, cap;
} obuf;

static void ob_need(obuf *b, int64_t extra) {
    if (b->n + extra <= b->cap) return;
    int64_t c = b->cap ? b->cap : 4096;
    while (c < b->n + extra) c *= 2;
    b->p = (char *)realloc(b->p, (size_t)c);
    b->cap = c;
}
static void ob_bytes(obuf *b, const char *s, int64_t n) {
    ob_need(b, n);
    memcpy(b->p + b->n, s, (size_t)n);
    b->n += n;
}
static void ob_char(obuf *b, char c) {
    ob_need(b, 1);
    b->p[b->n++] = c;
}
/* int64_to_str, impl/paf.c:10-34: '0' for zero, '-' prefix, plain decimal. */
static void ob_int(obuf *b, int64_t v) {
    char tmp[24];
    int k = 0;
    uint64_t u = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
    if (u == 0) tmp[k++] = '0';
    while (u) {
        tmp[k++] = (char)('0' + u % 10);
        u /= 10;
    }
    ob_need(b, k + 1);
    if (v < 0) b->p[b->n++] = '-';
    while (k) b->p[b->n++] = tmp[--k];
}

/* str_to_int64, impl/paf.c:37-48: optional '-', digits, no validation, wraps. */
static int64_t parse_i64(const char *s, const char *e) {
    uint64_t v = 0;
    int neg = 0;
    if (s < e && *s == '-') {
        neg = 1;
        s++;
    }
    while (s < e && *s >= '0' && *s <= '9') v = v * 10 + (uint64_t)(*s++ - '0');
    return (int64_t)(neg ? (uint64_t)0 - v : v);
}

/* ------------------------------------------------------------------ */
/* record                                                               */
/* ------------------------------------------------------------------ */

typedef struct {
    const char *qname, *tname; /* slices (not NUL terminated) */
    int64_t qname_len, tname_len;
    int64_t qlen, qs, qe, tlen, ts, te;
    int64_t nmatch, nbases, mapq;
    int64_t score, tile_level, chain_id, chain_score;
    int same_strand;
    char type;
    int has_cigar; /* parsed cigar present (non-NULL in the reference) */
    oop *ops;      /* owned */
    int64_t lo, n, cap;
    const char *cg_str; /* unparsed cigar text (tile mode), NULL if no cg tag */
    int64_t cg_str_len;
} rec;

static void rec_free(rec *r) {
    free(r->ops);
    r->ops = NULL;
}

static int op_code(char c) {
    switch (c) { /* impl/paf.c:96-103 */
        case 'M': return OP_M;
        case '=': return OP_EQ;
        case 'X': return OP_X;
        case 'I': return OP_I;
        case 'D': return OP_D;
        default: return -1;
    }
}
static char op_char(int op) { /* impl/paf.c:372-379 */
    switch (op) {
        case OP_M: return 'M';
        case OP_I: return 'I';
        case OP_D: return 'D';
        case OP_EQ: return '=';
        case OP_X: return 'X';
        default: return 'N';
    }
}

/*
 * cigar_parse, impl/paf.c:70-111. Empty text -> no cigar (returns 0 and
 * *present=0). Every character that is not a digit ends an op; anything
 * outside MID=X aborts, as does a trailing digit run (the reference's switch
 * then sees the terminating NUL).
 */
static int cigar_from_text(const char *s, const char *e, oop **ops_out, int64_t *n_out, int *present, int64_t *bad) {
    *ops_out = NULL;
    *n_out = 0;
    *present = 0;
    if (s == e) return PO_OK;
    int64_t cnt = 0;
    for (const char *c = s; c < e; c++)
        if ((unsigned)(*c - '0') > 9u) cnt++;
    oop *ops = (oop *)malloc(sizeof(oop) * (size_t)(cnt + 1));
    int64_t k = 0;
    const char *c = s;
    while (c < e) {
        uint64_t len = 0;
        while (c < e && *c >= '0' && *c <= '9') len = len * 10 + (uint64_t)(*c++ - '0');
        int op = c < e ? op_code(*c) : -1;
        if (op < 0) {
            *bad = c < e ? (unsigned char)*c : 0;
            free(ops);
            return PO_ERR_CIGAR_CHAR;
        }
        ops[k].len = wrap56((int64_t)len);
        ops[k].op = op;
        k++;
        c++;
    }
    *ops_out = ops;
    *n_out = k;
    *present = 1;
    return PO_OK;
}

/*
 * paf_parse, impl/paf.c:137-209, on the byte range [p, e) of one line.
 * strtok_r(.., "\t") semantics: tokens are maximal runs of non-tab bytes.
 * Deviation (documented): a tag token shorter than 5 bytes is never
 * recognised; the reference would read past the token terminator there.
 */
static int parse_line(const char *p, const char *e, int parse_cigar, rec *r, int64_t *aux) {
    memset(r, 0, sizeof(*r));
    r->tile_level = -1;
    r->chain_id = -1;
    r->chain_score = -1;
    int field = 0;
    const char *c = p;
    for (;;) {
        while (c < e && *c == '\t') c++;
        if (c >= e) break;
        const char *t = c;
        while (c < e && *c != '\t') c++;
        const char *te = c;
        switch (field) {
            case 0: r->qname = t; r->qname_len = te - t; break;
            case 1: r->qlen = parse_i64(t, te); break;
            case 2: r->qs = parse_i64(t, te); break;
            case 3: r->qe = parse_i64(t, te); break;
            case 4:
                if (*t != '+' && *t != '-') {
                    *aux = (unsigned char)*t;
                    return PO_ERR_STRAND;
                }
                r->same_strand = *t == '+';
                break;
            case 5: r->tname = t; r->tname_len = te - t; break;
            case 6: r->tlen = parse_i64(t, te); break;
            case 7: r->ts = parse_i64(t, te); break;
            case 8: r->te = parse_i64(t, te); break;
            case 9: r->nmatch = parse_i64(t, te); break;
            case 10: r->nbases = parse_i64(t, te); break;
            case 11: r->mapq = parse_i64(t, te); break;
            default: {
                if (te - t < 5 || t[2] != ':' || t[4] != ':') break;
                const char *v = t + 5;
                if (t[0] == 't' && t[1] == 'p') {
                    r->type = v < te ? *v : '\0';
                    if (r->type != 'P' && r->type != 'S' && r->type != 'I') {
                        *aux = (unsigned char)r->type;
                        return PO_ERR_TP_ASSERT;
                    }
                } else if (t[0] == 'A' && t[1] == 'S') {
                    r->score = parse_i64(v, te);
                } else if (t[0] == 'c' && t[1] == 'g') {
                    if (parse_cigar) {
                        free(r->ops); /* duplicate tag: last one wins (impl/paf.c:193-198) */
                        r->ops = NULL;
                        int rc = cigar_from_text(v, te, &r->ops, &r->n, &r->has_cigar, aux);
                        r->lo = 0;
                        r->cap = r->n;
                        if (rc) return rc;
                    } else {
                        r->cg_str = v;
                        r->cg_str_len = te - v;
                    }
                } else if (t[0] == 't' && t[1] == 'l') {
                    r->tile_level = parse_i64(v, te);
                } else if (t[0] == 'c' && t[1] == 'n') {
                    r->chain_id = parse_i64(v, te);
                } else if (t[0] == 's' && t[1] == '1') {
                    r->chain_score = parse_i64(v, te);
                }
            }
        }
        field++;
    }
    if (field < 12) return PO_ERR_FEW_FIELDS;
    return PO_OK;
}

/* paf_write_to_buffer, impl/paf.c:317-389 (grammar: SURVEY Appendix B). */
static void write_rec(const rec *r, obuf *b) {
    ob_bytes(b, r->qname, r->qname_len);
    ob_char(b, '\t');
    ob_int(b, r->qlen); ob_char(b, '\t');
    ob_int(b, r->qs); ob_char(b, '\t');
    ob_int(b, r->qe); ob_char(b, '\t');
    ob_char(b, r->same_strand ? '+' : '-'); ob_char(b, '\t');
    ob_bytes(b, r->tname, r->tname_len);
    ob_char(b, '\t');
    ob_int(b, r->tlen); ob_char(b, '\t');
    ob_int(b, r->ts); ob_char(b, '\t');
    ob_int(b, r->te); ob_char(b, '\t');
    ob_int(b, r->nmatch); ob_char(b, '\t');
    ob_int(b, r->nbases); ob_char(b, '\t');
    ob_int(b, r->mapq);
    if (r->type != '\0' || r->tile_level != -1) {
        char t = r->type;
        if (t == '\0') t = r->tile_level > 1 ? 'S' : 'P';
        ob_bytes(b, "\ttp:A:", 6);
        ob_char(b, t);
    }
    /* score != INT_MAX is the only guard (impl/paf.c:349): AS is always written */
    if (r->score != 2147483647LL) {
        ob_bytes(b, "\tAS:i:", 6);
        ob_int(b, r->score);
    }
    if (r->tile_level != -1) { ob_bytes(b, "\ttl:i:", 6); ob_int(b, r->tile_level); }
    if (r->chain_id != -1) { ob_bytes(b, "\tcn:i:", 6); ob_int(b, r->chain_id); }
    if (r->chain_score != -1) { ob_bytes(b, "\ts1:i:", 6); ob_int(b, r->chain_score); }
    if (r->has_cigar) {
        ob_bytes(b, "\tcg:Z:", 6);
        for (int64_t i = 0; i < r->n; i++) {
            ob_int(b, r->ops[r->lo + i].len);
            ob_char(b, op_char(r->ops[r->lo + i].op));
        }
    } else if (r->cg_str) {
        ob_bytes(b, "\tcg:Z:", 6);
        ob_bytes(b, r->cg_str, r->cg_str_len);
    }
    ob_char(b, '\n');
}

/* paf_check, impl/paf.c:427-461. */
static int check_rec(const rec *r) {
    if (r->qs < 0 || r->qs >= r->qlen) return PO_ERR_CHECK_QSTART;
    if (r->qs > r->qe || r->qe > r->qlen) return PO_ERR_CHECK_QEND;
    if (r->ts < 0 || r->ts >= r->tlen) return PO_ERR_CHECK_TSTART;
    if (r->ts > r->te || r->te > r->tlen) return PO_ERR_CHECK_TEND;
    if (r->has_cigar) {
        int64_t i = 0, j = 0;
        for (int64_t k = 0; k < r->n; k++) {
            const oop *o = &r->ops[r->lo + k];
            if (o->op != OP_D) i += o->len;
            if (o->op != OP_I) j += o->len;
        }
        if (i != r->qe - r->qs) return PO_ERR_CHECK_CIGAR_Q;
        if (j != r->te - r->ts) return PO_ERR_CHECK_CIGAR_T;
    }
    return PO_OK;
}

/* paf_invert + cigar_reverse, impl/paf.c:124-135,469-490. */
static void invert_rec(rec *r) {
    int64_t t;
    const char *s;
    t = r->qs; r->qs = r->ts; r->ts = t;
    t = r->qe; r->qe = r->te; r->te = t;
    t = r->qlen; r->qlen = r->tlen; r->tlen = t;
    s = r->qname; r->qname = r->tname; r->tname = s;
    t = r->qname_len; r->qname_len = r->tname_len; r->tname_len = t;
    if (!r->has_cigar) return;
    for (int64_t k = 0; k < r->n; k++) {
        oop *o = &r->ops[r->lo + k];
        if (o->op == OP_I) o->op = OP_D;
        else if (o->op == OP_D) o->op = OP_I;
    }
    if (!r->same_strand) {
        int64_t a = r->lo, b = r->lo + r->n - 1;
        while (a < b) {
            oop tmp = r->ops[a];
            r->ops[a] = r->ops[b];
            r->ops[b] = tmp;
            a++;
            b--;
        }
    }
}

/* ---- identity trim: impl/paf.c:811-953, arithmetic per SURVEY Appendix A 13-18 ---- */

/* paf_trim_unreliable_ends2, impl/paf.c:811-840 (less_than is always 1 at its call sites). */
static int64_t trim_scan(const rec *r, int64_t *m_out, int64_t *x_out, double thr, int64_t max_trim) {
    int64_t m = 0, x = 0, idx_found = -1;
    for (int64_t k = 0; k < r->n; k++) {
        const oop *o = &r->ops[r->lo + k];
        if (o->op == OP_EQ || o->op == OP_M) m += o->len;
        else x += o->len; /* X, I and D all count as mismatches */
        if (max_trim >= 0 && m + x > max_trim) break;
        volatile float fm = (float)m, fd = (float)(m + x);
        volatile float q = fm / fd; /* float32 divide, widened afterwards (impl/paf.c:832) */
        double pid = (double)q;
        if (pid < thr) idx_found = k;
    }
    *m_out = m;
    *x_out = x;
    return idx_found;
}

/* paf_trim_upto, impl/paf.c:842-861. */
static void trim_upto(rec *r, int64_t count) {
    for (int64_t k = 0; k < count; k++) {
        const oop *o = &r->ops[r->lo + k];
        if (o->op != OP_I) r->ts += o->len;
        if (o->op != OP_D) {
            if (r->same_strand) r->qs += o->len;
            else r->qe -= o->len;
        }
    }
    r->lo += count;
    r->n -= count;
}

/* paf_trim_unreliable_prefix, impl/paf.c:863-904: both thresholds arrive as float32. */
static void trim_prefix(rec *r, float thr_f, float id_f, int64_t max_trim) {
    int64_t m, x;
    int64_t trim_idx = trim_scan(r, &m, &x, (double)thr_f, max_trim);
    if (trim_idx < 0) return;
    int64_t sm = 0, sx = 0, best = -1;
    for (int64_t k = trim_idx; k >= 0; k--) {
        const oop *o = &r->ops[r->lo + k];
        if (o->op == OP_EQ || o->op == OP_M) sm += o->len;
        else sx += o->len;
        volatile float fm = (float)sm, fd = (float)(sm + sx);
        volatile float q = fm / fd;
        double sid = (double)q;
        if (sid >= (double)id_f) best = k;
    }
    int64_t count = best >= 0 ? best : trim_idx + 1;
    if (count > 0) trim_upto(r, count);
}

/* paf_trim_unreliable_tails, impl/paf.c:906-953. */
static int trim_identity(rec *r, float score_fraction, float max_fraction) {
    int64_t m, x;
    trim_scan(r, &m, &x, 0.0, -1);
    volatile float fm = (float)m, fd = (float)(m + x);
    volatile float q = fm / fd;
    double identity = (double)q;
    volatile double prod = identity * (double)score_fraction;
    double thr = identity - prod;
    volatile float ft = (float)(m + x);
    volatile float fmt = ft * max_fraction;
    int64_t max_trim = (int64_t)fmt;
    trim_prefix(r, (float)thr, (float)identity, max_trim);
    invert_rec(r);
    trim_prefix(r, (float)thr, (float)identity, max_trim);
    invert_rec(r);
    int64_t m2, x2;
    trim_scan(r, &m2, &x2, 0.0, -1);
    volatile float fm2 = (float)m2, fd2 = (float)(m2 + x2);
    volatile float q2 = fm2 / fd2;
    double final_identity = (double)q2;
    if (!(final_identity >= identity)) return PO_ERR_TRIM_IDENTITY_ASSERT;
    return PO_OK;
}

/* ---- fixed trim: impl/paf.c:507-598 ---- */

static int is_aligned_op(int op) { return op == OP_M || op == OP_EQ || op == OP_X; }

static int64_t aligned_bases(const rec *r) { /* impl/paf.c:507-516 */
    int64_t a = 0;
    for (int64_t k = 0; k < r->n; k++)
        if (is_aligned_op(r->ops[r->lo + k].op)) a += r->ops[r->lo + k].len;
    return a;
}

/* cigar_trim, impl/paf.c:518-545 (front == 1) and cigar_trim_back, :547-576 (front == 0). */
static void trim_one_end(rec *r, int64_t *qc, int64_t *tc, int64_t end, int qsign, int tsign, int front) {
    int64_t done = 0;
    while (r->n > 0) {
        oop *o = &r->ops[front ? r->lo : r->lo + r->n - 1];
        int al = is_aligned_op(o->op);
        if (al && !(done < end)) break;
        if (al) {
            if (done + o->len > end) {
                int64_t i = end - done;
                o->len = wrap56(o->len - i);
                *qc += qsign * i;
                *tc += tsign * i;
                break;
            }
            done += o->len;
            *qc += qsign * o->len;
            *tc += tsign * o->len;
        } else if (o->op == OP_I) {
            *qc += qsign * o->len;
        } else {
            *tc += tsign * o->len;
        }
        if (front) r->lo++;
        r->n--;
    }
}

/* paf_trim_ends, impl/paf.c:578-587. */
static int trim_ends(rec *r, int64_t end) {
    if (!r->has_cigar) return PO_ERR_NULL_CIGAR; /* reference dereferences the NULL cigar */
    if (r->same_strand) {
        trim_one_end(r, &r->qs, &r->ts, end, 1, 1, 1);
        trim_one_end(r, &r->qe, &r->te, end, -1, -1, 0);
    } else {
        trim_one_end(r, &r->qe, &r->ts, end, -1, 1, 1);
        trim_one_end(r, &r->qs, &r->te, end, 1, -1, 0);
    }
    return PO_OK;
}

/* paf_trim_end_fraction, impl/paf.c:589-598. */
static int trim_fixed(rec *r, float pct) {
    if (!(pct >= 0 && pct <= 1.0)) return PO_ERR_TRIM_FIXED_ASSERT;
    int64_t a = aligned_bases(r);
    volatile float prod = (float)a * pct; /* int64 * float -> float32 multiply */
    double half = (double)prod / 2.0;
    int64_t end = (int64_t)half;
    return trim_ends(r, end);
}

/* ---- mismatch encoding: impl/paf.c:739-809 ---- */

/* stString_reverseComplementChar [sonLib, absent]: A<->T, C<->G in both cases; every other
 * byte maps to itself (parity unpinned for non-ACGT letters). */
static char rc_char(char c) {
    switch (c) {
        case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C';
        case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c';
        default: return c;
    }
}
static int up(char c) { return (c >= 'a' && c <= 'z') ? c - 32 : (unsigned char)c; } /* toupper, C locale */

/* paf_encode_mismatches, impl/paf.c:739-784. */
static int encode_mismatches(rec *r, const po_seq *q, const po_seq *t) {
    if (!r->has_cigar) return PO_OK;
    int64_t cap = r->n * 2 + 16, out = 0;
    oop *nw = (oop *)malloc(sizeof(oop) * (size_t)cap);
    int64_t qi = 0, tj = r->ts;
    for (int64_t k = 0; k < r->n; k++) {
        const oop *o = &r->ops[r->lo + k];
        if (o->op == OP_M) {
            int64_t toff = tj;
            int64_t qoff = r->same_strand ? r->qs + qi : r->qe - (qi + 1);
            int prev = 0, first = 1;
            for (int64_t i = 0; i < o->len; i++) {
                int64_t tp = toff + i, qp = r->same_strand ? qoff + i : qoff - i;
                if (tp < 0 || tp >= t->len || qp < 0 || qp >= q->len) {
                    free(nw);
                    return PO_ERR_SEQ_RANGE;
                }
                char qc = r->same_strand ? q->seq[qp] : rc_char(q->seq[qp]);
                int is_match = up(t->seq[tp]) == up(qc);
                if (first || is_match != prev) {
                    if (out >= cap) {
                        cap *= 2;
                        nw = (oop *)realloc(nw, sizeof(oop) * (size_t)cap);
                    }
                    nw[out].op = is_match ? OP_EQ : OP_X;
                    nw[out].len = 1;
                    out++;
                    first = 0;
                } else {
                    nw[out - 1].len = wrap56(nw[out - 1].len + 1);
                }
                prev = is_match;
            }
            qi += o->len;
            tj += o->len;
        } else {
            if (out >= cap) {
                cap *= 2;
                nw = (oop *)realloc(nw, sizeof(oop) * (size_t)cap);
            }
            nw[out++] = *o;
            if (o->op == OP_I) qi += o->len;
            else if (o->op == OP_D) tj += o->len;
            else { qi += o->len; tj += o->len; }
        }
    }
    free(r->ops);
    r->ops = nw;
    r->lo = 0;
    r->n = out;
    r->cap = cap;
    return PO_OK;
}

/* paf_remove_mismatches, impl/paf.c:786-809. */
static void remove_mismatches(rec *r) {
    if (!r->has_cigar) return;
    int64_t w = 0;
    for (int64_t k = 0; k < r->n; k++) {
        oop o = r->ops[r->lo + k];
        if (o.op == OP_EQ || o.op == OP_X || o.op == OP_M) {
            if (w > 0 && r->ops[r->lo + w - 1].op == OP_M) {
                r->ops[r->lo + w - 1].len = wrap56(r->ops[r->lo + w - 1].len + o.len);
            } else {
                r->ops[r->lo + w].len = o.len;
                r->ops[r->lo + w].op = OP_M;
                w++;
            }
        } else {
            r->ops[r->lo + w] = o;
            w++;
        }
    }
    r->n = w;
}

/* ------------------------------------------------------------------ */
/* stream driver                                                        */
/* ------------------------------------------------------------------ */

typedef struct {
    const po_stage *stages;
    int32_t n_stages;
    const po_seq *seqs;
    int64_t n_seqs;
    obuf out;
    po_error *err;
    int64_t record;
} run_ctx;

static int fail(run_ctx *c, int code, int stage, int64_t aux) {
    if (c->err) {
        c->err->code = code;
        c->err->stage = stage;
        c->err->record = c->record;
        c->err->aux = aux;
    }
    return code;
}

static const po_seq *find_seq(const run_ctx *c, const char *name, int64_t len) {
    for (int64_t i = 0; i < c->n_seqs; i++)
        if ((int64_t)strlen(c->seqs[i].name) == len && memcmp(c->seqs[i].name, name, (size_t)len) == 0) return &c->seqs[i];
    return NULL;
}

/*
 * What a later process sees after `paf_write | paf_parse` of this record:
 * an empty cigar text parses to "no cigar" (impl/paf.c:71-73) and a
 * synthesised tp letter (impl/paf.c:343-348) becomes a stored one.
 */
static void reparse_normalise(rec *r) {
    if (r->has_cigar && r->n == 0) r->has_cigar = 0;
    if (r->type == '\0' && r->tile_level != -1) r->type = r->tile_level > 1 ? 'S' : 'P';
}

static int run_from(run_ctx *c, rec *r, int32_t s);

/* paffy filter, impl/paf_filter.c:120-156: 1 = the record is written, 0 = it is dropped. */
static po_filter g_filter = {-1, -1, -1.0, -1.0, -1, 0};
void po_set_filter(const po_filter *f) {
    po_filter d = {-1, -1, -1.0, -1.0, -1, 0};
    g_filter = f ? *f : d;
}
static int filter_keeps(const rec *r) {
    /* paf_stats_calc, impl/paf.c:236-260 (a NULL cigar has no ops, inc/paf.h:75) */
    int64_t matches = 0, mismatches = 0, qi_bases = 0, qd_bases = 0;
    for (int64_t k = 0; r->has_cigar && k < r->n; k++) {
        const oop *o = &r->ops[r->lo + k];
        if (o->op == OP_EQ || o->op == OP_M) matches += o->len;
        else if (o->op == OP_X) mismatches += o->len;
        else if (o->op == OP_I) qi_bases += o->len;
        else qd_bases += o->len;
    }
    /* float32 quotients widened to double, impl/paf_filter.c:129-130 (0/0 is NaN: every >= fails) */
    double identity = (float)matches / (matches + mismatches);
    double identity_with_gaps = (float)matches / (matches + mismatches + qi_bases + qd_bases);
    int pass = r->score >= g_filter.min_alignment_score && r->chain_score >= g_filter.min_chain_score &&
               (g_filter.max_tile_level == -1 || r->tile_level <= g_filter.max_tile_level) && identity >= g_filter.min_identity &&
               identity_with_gaps >= g_filter.min_identity_with_gaps;
    return g_filter.invert ? !pass : pass;
}

/* paf_shatter + paf_shatter2, impl/paf.c:600-663, then the driver loop impl/paf_shatter.c:88-95. */
static int shatter_stage(run_ctx *c, rec *r, int32_t s) {
    int64_t qc = r->same_strand ? r->qs : r->qe;
    int64_t tc = r->ts;
    int64_t nkids = 0, kcap = r->has_cigar ? r->n : 0;
    rec *kids = (rec *)malloc(sizeof(rec) * (size_t)(kcap + 1));
    int rc = PO_OK;
    for (int64_t k = 0; r->has_cigar && k < r->n; k++) {
        const oop *o = &r->ops[r->lo + k];
        if (!(o->len >= 1)) { rc = PO_ERR_SHATTER_ZERO_LEN; break; }
        if (o->op == OP_M) {
            int64_t qstart;
            if (r->same_strand) {
                qstart = qc;
                qc += o->len;
            } else {
                qc -= o->len;
                qstart = qc;
            }
            rec *kid = &kids[nkids];
            memset(kid, 0, sizeof(*kid)); /* calloc: chain_score stays 0, impl/paf.c:601 */
            kid->qname = r->qname; kid->qname_len = r->qname_len;
            kid->qlen = r->qlen; kid->qs = qstart; kid->qe = qstart + o->len;
            kid->tname = r->tname; kid->tname_len = r->tname_len;
            kid->tlen = r->tlen; kid->ts = tc; kid->te = tc + o->len;
            kid->same_strand = r->same_strand;
            kid->has_cigar = 1;
            kid->ops = (oop *)malloc(sizeof(oop));
            kid->ops[0].len = o->len;
            kid->ops[0].op = OP_M;
            kid->lo = 0; kid->n = 1; kid->cap = 1;
            kid->score = r->score;
            kid->mapq = r->mapq;
            kid->nmatch = o->len;
            kid->nbases = o->len;
            kid->tile_level = r->tile_level;
            kid->type = r->type;
            kid->chain_id = r->chain_id;
            nkids++;
            rc = check_rec(kid);
            if (rc) break;
            tc += o->len;
        } else if (o->op == OP_I) {
            qc += r->same_strand ? o->len : -o->len;
        } else {
            if (o->op != OP_D) { rc = PO_ERR_SHATTER_BAD_OP; break; }
            tc += o->len;
        }
    }
    if (!rc) {
        if (tc != r->te) rc = PO_ERR_SHATTER_END;
        else if (r->same_strand ? qc != r->qe : qc != r->qs) rc = PO_ERR_SHATTER_END;
    }
    if (rc) rc = fail(c, rc, s, 0);
    for (int64_t k = 0; k < nkids; k++) {
        if (!rc) rc = run_from(c, &kids[k], s + 1);
        rec_free(&kids[k]);
    }
    free(kids);
    return rc;
}

static int run_from(run_ctx *c, rec *r, int32_t s) {
    for (; s < c->n_stages; s++) {
        const po_stage *st = &c->stages[s];
        int rc = PO_OK;
        if (s > 0) reparse_normalise(r);
        switch (st->kind) {
            case PO_INVERT:
                invert_rec(r);
                rc = check_rec(r);
                break;
            case PO_TRIM_IDENTITY:
                rc = trim_identity(r, st->p0, st->p1);
                if (!rc) rc = check_rec(r);
                break;
            case PO_TRIM_FIXED:
                rc = trim_fixed(r, st->p1);
                if (!rc) rc = check_rec(r);
                break;
            case PO_SHATTER:
                return shatter_stage(c, r, s);
            case PO_ADD_MISMATCHES: {
                const po_seq *q = find_seq(c, r->qname, r->qname_len);
                if (!q) { rc = PO_ERR_MISSING_QUERY_SEQ; break; }
                const po_seq *t = find_seq(c, r->tname, r->tname_len);
                if (!t) { rc = PO_ERR_MISSING_TARGET_SEQ; break; }
                rc = encode_mismatches(r, q, t);
                if (!rc) rc = check_rec(r);
                break;
            }
            case PO_REMOVE_MISMATCHES:
                remove_mismatches(r);
                rc = check_rec(r);
                break;
            case PO_PASS:
                break;
            case PO_FILTER:
                if (!filter_keeps(r)) return PO_OK; /* nothing is written, nothing reaches the later stages */
                break;
            default:
                rc = -1;
        }
        if (rc) return fail(c, rc, s, 0);
    }
    write_rec(r, &c->out);
    return PO_OK;
}

int po_run(const po_stage *stages, int32_t n_stages, const char *in, int64_t in_len, const po_seq *seqs,
           int64_t n_seqs, char **out, int64_t *out_len, po_error *err) {
    run_ctx c;
    memset(&c, 0, sizeof(c));
    c.stages = stages;
    c.n_stages = n_stages;
    c.seqs = seqs;
    c.n_seqs = n_seqs;
    c.err = err;
    if (err) memset(err, 0, sizeof(*err));
    int rc = PO_OK;
    const char *p = in, *end = in + in_len;
    /* paf_read_with_buffer, impl/paf.c:211-218: a final line without '\n' is still a record */
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        rec r;
        int64_t aux = 0;
        rc = parse_line(p, le, 1, &r, &aux);
        if (rc) {
            fail(&c, rc, -1, aux);
            rec_free(&r);
            break;
        }
        rc = run_from(&c, &r, 0);
        rec_free(&r);
        if (rc) break;
        c.record++;
        p = nl ? nl + 1 : end;
    }
    *out = c.out.p;
    *out_len = c.out.n;
    return rc;
}

/* ------------------------------------------------------------------ */
/* tile: impl/paf_tile.c:28-93,156-178 and impl/paf.c:675-709          */
/* ------------------------------------------------------------------ */

typedef struct {
    const char *name;
    int64_t name_len, length;
    uint16_t *counts;
} count_array;

static const rec *g_sort_recs;
/* paf_cmp_by_descending_score, impl/paf_tile.c:28-34; input index as the final key restates the
 * stable glibc-2.35 qsort the survey ran against (parity unpinned). */
static int cmp_rank(const void *a, const void *b) {
    int64_t i = *(const int64_t *)a, j = *(const int64_t *)b;
    const rec *x = &g_sort_recs[i], *y = &g_sort_recs[j];
    if (x->chain_score != y->chain_score) return x->chain_score > y->chain_score ? -1 : 1;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    return i < j ? -1 : (i > j ? 1 : 0);
}

/* paffy dedupe, impl/paf_dedupe.c:27-46,117-143. A linear probe table of the written records (exact key compare). */
static int dedupe_same(const rec *a, const rec *b, int b_swapped) {
    const char *bq = b_swapped ? b->tname : b->qname, *bt = b_swapped ? b->qname : b->tname;
    int64_t bql = b_swapped ? b->tname_len : b->qname_len, btl = b_swapped ? b->qname_len : b->tname_len;
    int64_t bqs = b_swapped ? b->ts : b->qs, bqe = b_swapped ? b->te : b->qe, bts = b_swapped ? b->qs : b->ts, bte = b_swapped ? b->qe : b->te;
    return a->qname_len == bql && a->tname_len == btl && memcmp(a->qname, bq, (size_t)bql) == 0 && memcmp(a->tname, bt, (size_t)btl) == 0 &&
           a->same_strand == b->same_strand && a->ts == bts && a->te == bte && a->qs == bqs && a->qe == bqe;
}
int po_dedupe(const char *in, int64_t in_len, int check_inverse, char **out, int64_t *out_len, po_error *err) {
    run_ctx c;
    memset(&c, 0, sizeof(c));
    c.err = err;
    if (err) memset(err, 0, sizeof(*err));
    int rc = PO_OK;
    int64_t nkept = 0, kcap = 1024;
    rec *kept = (rec *)malloc(sizeof(rec) * (size_t)kcap);
    int64_t tcap = 4096; /* open addressing on the coordinate sum, as paf_hash_key (impl/paf_dedupe.c:27-35) keys on it */
    int64_t *table = (int64_t *)malloc(sizeof(int64_t) * (size_t)tcap);
    for (int64_t i = 0; i < tcap; i++) table[i] = -1;
    const char *p = in, *end = in + in_len;
    while (p < end && !rc) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        rec r;
        int64_t aux = 0;
        rc = parse_line(p, le, 0, &r, &aux); /* paf_read_with_buffer(input, 0, ...): the cigar stays text */
        if (rc) {
            fail(&c, rc, -1, aux);
            break;
        }
        uint64_t key = (uint64_t)(r.qs + r.qe + r.ts + r.te);
        int found = 0;
        for (uint64_t h = key % (uint64_t)tcap;; h = (h + 1) % (uint64_t)tcap) {
            if (table[h] < 0) break;
            if (dedupe_same(&r, &kept[table[h]], 0)) { found = 1; break; }
        }
        if (!found && check_inverse) { /* the inverse has the same coordinate sum */
            for (uint64_t h = key % (uint64_t)tcap;; h = (h + 1) % (uint64_t)tcap) {
                if (table[h] < 0) break;
                if (dedupe_same(&r, &kept[table[h]], 1)) { found = 1; break; }
            }
            rc = check_rec(&r); /* paf_check after inverting back, impl/paf_dedupe.c:126 (the cigar is not parsed: coordinates only) */
            if (rc) {
                fail(&c, rc, 0, 0);
                rec_free(&r);
                break;
            }
        }
        if (!found) {
            if (nkept == kcap) {
                kcap *= 2;
                kept = (rec *)realloc(kept, sizeof(rec) * (size_t)kcap);
            }
            if ((nkept + 1) * 2 > tcap) { /* grow and rehash */
                tcap *= 4;
                table = (int64_t *)realloc(table, sizeof(int64_t) * (size_t)tcap);
                for (int64_t i = 0; i < tcap; i++) table[i] = -1;
                for (int64_t k = 0; k < nkept; k++) {
                    uint64_t h = (uint64_t)(kept[k].qs + kept[k].qe + kept[k].ts + kept[k].te) % (uint64_t)tcap;
                    while (table[h] >= 0) h = (h + 1) % (uint64_t)tcap;
                    table[h] = k;
                }
            }
            kept[nkept] = r;
            uint64_t h = key % (uint64_t)tcap;
            while (table[h] >= 0) h = (h + 1) % (uint64_t)tcap;
            table[h] = nkept++;
            write_rec(&r, &c.out);
        } else {
            rec_free(&r);
        }
        c.record++;
        p = nl ? nl + 1 : end;
    }
    for (int64_t k = 0; k < nkept; k++) rec_free(&kept[k]);
    free(kept);
    free(table);
    *out = c.out.p;
    *out_len = c.out.n;
    return rc;
}

/* paffy split_file, impl/paf_split_file.c:27-56,131-173 */
typedef struct { char *name; int64_t name_len; FILE *fh; } split_slot;
static FILE *split_find(split_slot *v, int64_t n, const char *name, int64_t len) {
    for (int64_t i = 0; i < n; i++)
        if (v[i].name_len == len && memcmp(v[i].name, name, (size_t)len) == 0) return v[i].fh;
    return NULL;
}
static void split_add(split_slot **v, int64_t *n, int64_t *cap, const char *name, int64_t len, FILE *fh) {
    if (*n == *cap) {
        *cap = *cap ? *cap * 2 : 64;
        *v = (split_slot *)realloc(*v, sizeof(split_slot) * (size_t)*cap);
    }
    (*v)[*n].name = (char *)malloc((size_t)len + 1);
    memcpy((*v)[*n].name, name, (size_t)len);
    (*v)[*n].name_len = len;
    (*v)[*n].fh = fh;
    (*n)++;
}
/* ---- paffy chain: impl/chaining.c:1-343, impl/paf_chain.c:36-45,123-130 ---- */

/*
 * The reference keeps the chains it can still extend in a sonLib stSortedSet (libavl underneath; sonLib is an un-vendored
 * submodule, absent here) ordered by chain_cmp_by_location, and every comparator ends in a comparison of object addresses
 * (impl/chaining.c:18,47,62). Restated with a sorted array and, for the addresses, the order the objects were made in: Paf objects
 * in input order (read_pafs), Chain objects in processing order. Parity on exact ties is therefore unpinned (heap layout).
 * When no active chain sorts <= the search key the reference takes a fresh iterator and calls stSortedSet_getPrevious
 * (impl/chaining.c:74-76): with libavl's avl_t_prev that starts at the largest element of the set; restated so, and every
 * candidate evaluated from such a start is counted in *fresh_hits (it can only be an exactly abutting alignment whose address
 * is higher -- the same unpinned tie).
 */
typedef struct {
    int64_t rec;   /* index into recs (input order = address order) */
    int64_t score; /* best chain score ending here */
    int64_t prev;  /* chain index or -1 */
} ochain;

static int name_cmp(const char *a, int64_t al, const char *b, int64_t bl) { /* strcmp of the NUL-terminated names */
    int64_t m = al < bl ? al : bl;
    int d = memcmp(a, b, (size_t)m);
    if (d) return d < 0 ? -1 : 1;
    return al < bl ? -1 : (al > bl ? 1 : 0);
}
static int icmp(int64_t i, int64_t j) { return i > j ? 1 : (i < j ? -1 : 0); }

/* chain_cmp_by_location, impl/chaining.c:37-54, with explicit end coordinates for the searched key */
static int loc_cmp(const rec *a, int64_t a_te, int64_t a_qe, int64_t a_ptr, const rec *b, int64_t b_ptr) {
    int i = name_cmp(a->qname, a->qname_len, b->qname, b->qname_len);
    if (i == 0) i = name_cmp(a->tname, a->tname_len, b->tname, b->tname_len);
    if (i == 0) i = icmp(a_te, b->te);
    if (i == 0) i = icmp(a_qe, b->qe);
    if (i == 0) i = icmp(a_ptr, b_ptr);
    return i;
}

/* Tests only: 0 restates the search WITHOUT the walk from a fresh iterator (no candidate at all when no active chain sorts <= the key):
   what the GPU does on those inputs, see DESIGN 5. The default (1) is the reference as read. */
static int g_chain_fresh_walk = 1;
void po_set_chain_fresh_walk(int on) { g_chain_fresh_walk = on; }

static int64_t chain_gap_cost(int64_t dq, int64_t dt, int64_t gap_open, int64_t gap_extend) { /* impl/paf_chain.c:36-45 */
    return dq + dt == 0 ? 0 : gap_open + gap_extend * (dq + dt);
}

static const rec *g_chain_recs;
static int cmp_query_location(const void *a, const void *b) { /* paf_cmp_by_query_location, impl/chaining.c:14-21 */
    int64_t i = *(const int64_t *)a, j = *(const int64_t *)b;
    int c = icmp(g_chain_recs[i].qs, g_chain_recs[j].qs);
    return c ? c : icmp(i, j);
}
static const ochain *g_chain_set;
static int cmp_chain_score(const void *a, const void *b) { /* chain_cmp_by_score, impl/chaining.c:59-66 */
    int64_t i = *(const int64_t *)a, j = *(const int64_t *)b;
    int c = icmp(g_chain_set[i].score, g_chain_set[j].score);
    return c ? c : icmp(i, j);
}
static int cmp_paf_score(const void *a, const void *b) { /* paf_cmp_by_score, impl/chaining.c:261-264; ties: the stable merge sort of glibc */
    int64_t i = ((const int64_t *)a)[0], j = ((const int64_t *)b)[0];
    int c = icmp(g_chain_recs[j].score, g_chain_recs[i].score);
    return c ? c : icmp(((const int64_t *)a)[1], ((const int64_t *)b)[1]);
}

/* paf_chain_ignore_strand, impl/chaining.c:136-250: `list` (n records of one strand) -> appended to out_order; sets chain_id / chain_score */
static void chain_one_strand(rec *recs, int64_t *list, int64_t n, int64_t gap_open, int64_t gap_extend, int64_t max_gap, int64_t *chain_id,
                             int64_t *out_order, int64_t *n_out, int64_t *fresh_hits) {
    if (n == 0) return;
    g_chain_recs = recs;
    qsort(list, (size_t)n, sizeof(int64_t), cmp_query_location);
    ochain *ch = (ochain *)malloc(sizeof(ochain) * (size_t)n);
    int64_t *act = (int64_t *)malloc(sizeof(int64_t) * (size_t)n), n_act = 0; /* chain indices sorted by location */
    int64_t *to_remove = (int64_t *)malloc(sizeof(int64_t) * (size_t)n), n_rm = 0;
    for (int64_t k = 0; k < n; k++) {
        const rec *paf = &recs[list[k]];
        ch[k].rec = list[k];
        ch[k].score = paf->score;
        ch[k].prev = -1;
        /* get_predecessor_chains: the largest active chain <= (names, target_start, query_start, this paf's address) */
        int64_t lo = 0, hi = n_act; /* first position whose element is > key */
        while (lo < hi) {
            int64_t mid = (lo + hi) / 2;
            const ochain *e = &ch[act[mid]];
            /* cmp(key, e) >= 0  <=>  e <= key */
            if (loc_cmp(paf, paf->ts, paf->qs, list[k], &recs[e->rec], e->rec) >= 0) lo = mid + 1;
            else hi = mid;
        }
        int fresh = lo == 0;
        int64_t it = fresh ? (g_chain_fresh_walk ? n_act - 1 : -1) : lo - 1; /* fresh iterator: getPrevious gives the last element (libavl avl_t_prev) */
        for (; it >= 0; it--) {
            ochain *pc = &ch[act[it]];
            const rec *pp = &recs[pc->rec];
            if (name_cmp(paf->qname, paf->qname_len, pp->qname, pp->qname_len) != 0 || name_cmp(paf->tname, paf->tname_len, pp->tname, pp->tname_len) != 0 ||
                paf->same_strand != pp->same_strand)
                break;
            if (paf->qs < pp->qe) continue;
            if (paf->qs - pp->qe > max_gap) {
                to_remove[n_rm++] = act[it];
                continue;
            }
            if (paf->ts < pp->te) continue;
            if (paf->ts - pp->te > max_gap) break;
            if (fresh) (*fresh_hits)++;
            int64_t g = chain_gap_cost(paf->qs - pp->qe, paf->ts - pp->te, gap_open, gap_extend);
            int64_t cs = paf->score + pc->score - g;
            if (g < paf->score && cs > ch[k].score) {
                ch[k].score = cs;
                ch[k].prev = act[it];
            }
        }
        /* insert into the active set */
        lo = 0; hi = n_act;
        while (lo < hi) {
            int64_t mid = (lo + hi) / 2;
            const ochain *e = &ch[act[mid]];
            if (loc_cmp(paf, paf->te, paf->qe, list[k], &recs[e->rec], e->rec) > 0) lo = mid + 1;
            else hi = mid;
        }
        memmove(act + lo + 1, act + lo, sizeof(int64_t) * (size_t)(n_act - lo));
        act[lo] = k;
        n_act++;
        while (n_rm > 0) {
            int64_t victim = to_remove[--n_rm];
            for (int64_t a = 0; a < n_act; a++)
                if (act[a] == victim) {
                    memmove(act + a, act + a + 1, sizeof(int64_t) * (size_t)(n_act - a - 1));
                    n_act--;
                    break;
                }
        }
    }
    /* chains from the highest score down, impl/chaining.c:213-230 */
    int64_t *by_score = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    char *in_set = (char *)malloc((size_t)n);
    for (int64_t k = 0; k < n; k++) { by_score[k] = k; in_set[k] = 1; }
    g_chain_set = ch;
    qsort(by_score, (size_t)n, sizeof(int64_t), cmp_chain_score);
    for (int64_t top = n - 1; top >= 0; top--) {
        int64_t k = by_score[top];
        if (!in_set[k]) continue;
        in_set[k] = 0;
        int64_t tail = k;
        while (ch[k].prev != -1) {
            if (!in_set[ch[k].prev]) { ch[k].prev = -1; break; }
            k = ch[k].prev;
            in_set[k] = 0;
        }
        /* chain_to_pafs, impl/chaining.c:118-135 + get_chain_score :93-116 */
        int64_t total = recs[ch[tail].rec].score;
        for (int64_t c = tail; ch[c].prev != -1; c = ch[c].prev) {
            const rec *q = &recs[ch[c].rec], *p = &recs[ch[ch[c].prev].rec];
            total += p->score - chain_gap_cost(q->qs - p->qe, q->ts - p->te, gap_open, gap_extend);
        }
        for (int64_t c = tail; c != -1; c = ch[c].prev) {
            recs[ch[c].rec].chain_id = *chain_id;
            recs[ch[c].rec].chain_score = total;
            out_order[(*n_out)++] = ch[c].rec;
        }
        (*chain_id)++;
    }
    free(by_score); free(in_set); free(to_remove); free(act); free(ch);
}

int po_chain(const char *in, int64_t in_len, int64_t gap_open, int64_t gap_extend, int64_t max_gap, float pct, char **out, int64_t *out_len,
             int64_t *fresh_hits, po_error *err) {
    run_ctx c;
    memset(&c, 0, sizeof(c));
    c.err = err;
    if (err) memset(err, 0, sizeof(*err));
    int rc = PO_OK;
    int64_t nrec = 0, rcap = 1024, fresh = 0;
    rec *recs = (rec *)malloc(sizeof(rec) * (size_t)rcap);
    const char *p = in, *end = in + in_len;
    while (p < end) { /* read_pafs(input, 0), impl/paf_chain.c:123 */
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        if (nrec == rcap) {
            rcap *= 2;
            recs = (rec *)realloc(recs, sizeof(rec) * (size_t)rcap);
        }
        int64_t aux = 0;
        c.record = nrec;
        rc = parse_line(p, le, 0, &recs[nrec], &aux);
        if (rc) {
            fail(&c, rc, -1, aux);
            break;
        }
        nrec++;
        p = nl ? nl + 1 : end;
    }
    int64_t *trim = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nrec + 1));
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nrec + 1)), *neg = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nrec + 1));
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)(nrec + 1));
    int64_t npos = 0, nneg = 0, n_out = 0;
    for (int64_t i = 0; i < nrec && !rc; i++) { /* paf_chain, impl/chaining.c:266-300 */
        rec *r = &recs[i];
        c.record = i;
        if (!(pct >= 0 && pct <= 1.0)) { rc = fail(&c, PO_ERR_CHAIN_ASSERT, 0, 0); break; }
        volatile float fq = (float)(r->qe - r->qs) * pct, ft = (float)(r->te - r->ts) * pct; /* int64 * float -> float */
        int64_t mq = (int64_t)fq, mt = (int64_t)ft;
        if (mq < 0 || mt < 0) { rc = fail(&c, PO_ERR_CHAIN_ASSERT, 0, 1); break; }
        int64_t t = (mq < mt ? mq : mt) / 2;
        trim[i] = t;
        r->qs += t; r->qe -= t; r->ts += t; r->te -= t;
        if (r->same_strand) pos[npos++] = i;
        else { /* invert_query_strand, impl/chaining.c:255-259 */
            int64_t k = r->qs; r->qs = -r->qe; r->qe = -k;
            neg[nneg++] = i;
        }
    }
    if (!rc) {
        int64_t chain_id = 0;
        chain_one_strand(recs, pos, npos, gap_open, gap_extend, max_gap, &chain_id, order, &n_out, &fresh);
        int64_t first_neg = n_out;
        chain_one_strand(recs, neg, nneg, gap_open, gap_extend, max_gap, &chain_id, order, &n_out, &fresh);
        for (int64_t k = first_neg; k < n_out; k++) {
            rec *r = &recs[order[k]];
            int64_t q = r->qs; r->qs = -r->qe; r->qe = -q;
        }
        for (int64_t k = 0; k < n_out && !rc; k++) { /* remove the trim, paf_check, impl/chaining.c:323-335 */
            rec *r = &recs[order[k]];
            int64_t t = trim[order[k]];
            r->qs -= t; r->qe += t; r->ts -= t; r->te += t;
            int chk = check_rec(r);
            if (chk) { c.record = order[k]; rc = fail(&c, chk, 0, 0); }
        }
        if (!rc) {
            int64_t *keyed = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)(n_out + 1));
            for (int64_t k = 0; k < n_out; k++) { keyed[2 * k] = order[k]; keyed[2 * k + 1] = k; }
            g_chain_recs = recs;
            qsort(keyed, (size_t)n_out, 2 * sizeof(int64_t), cmp_paf_score);
            for (int64_t k = 0; k < n_out; k++) write_rec(&recs[keyed[2 * k]], &c.out); /* write_pafs, impl/paf_chain.c:127 */
            free(keyed);
        }
    }
    for (int64_t i = 0; i < nrec; i++) rec_free(&recs[i]);
    free(recs); free(trim); free(pos); free(neg); free(order);
    if (fresh_hits) *fresh_hits = fresh;
    if (rc) { free(c.out.p); c.out.p = NULL; c.out.n = 0; }
    *out = c.out.p;
    *out_len = c.out.n;
    return rc;
}

int po_split_file(const char *in, int64_t in_len, const char *prefix, int by_query, int64_t min_length, po_error *err) {
    run_ctx c;
    memset(&c, 0, sizeof(c));
    c.err = err;
    if (err) memset(err, 0, sizeof(*err));
    int rc = PO_OK;
    split_slot *big = NULL, *small = NULL;
    int64_t nbig = 0, cbig = 0, nsmall = 0, csmall = 0;
    FILE **small_files = NULL;
    int64_t n_small_files = 0, current_len = 0;
    FILE *current = NULL;
    const char *p = in, *end = in + in_len;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        rec r;
        int64_t aux = 0;
        rc = parse_line(p, le, 0, &r, &aux);
        if (rc) {
            fail(&c, rc, -1, aux);
            break;
        }
        const char *name = by_query ? r.qname : r.tname;
        int64_t name_len = by_query ? r.qname_len : r.tname_len, contig_len = by_query ? r.qlen : r.tlen;
        FILE *fh;
        if (min_length > 0 && contig_len < min_length) {
            fh = split_find(small, nsmall, name, name_len);
            if (!fh) {
                if (!current || current_len + contig_len > min_length) {
                    char path[4096];
                    snprintf(path, sizeof(path), "%ssmall_%lld.paf", prefix, (long long)n_small_files);
                    current = fopen(path, "w");
                    small_files = (FILE **)realloc(small_files, sizeof(FILE *) * (size_t)(n_small_files + 1));
                    small_files[n_small_files++] = current;
                    current_len = 0;
                }
                current_len += contig_len;
                split_add(&small, &nsmall, &csmall, name, name_len, current);
                fh = current;
            }
        } else {
            fh = split_find(big, nbig, name, name_len);
            if (!fh) {
                char path[4096];
                int k = snprintf(path, sizeof(path), "%s", prefix);
                for (int64_t i = 0; i < name_len && k < (int)sizeof(path) - 8; i++) path[k++] = name[i] == '/' ? '_' : name[i];
                snprintf(path + k, sizeof(path) - (size_t)k, ".paf");
                fh = fopen(path, "w");
                split_add(&big, &nbig, &cbig, name, name_len, fh);
            }
        }
        obuf b = {0, 0, 0};
        write_rec(&r, &b);
        if (fh) fwrite(b.p, 1, (size_t)b.n, fh);
        free(b.p);
        rec_free(&r);
        c.record++;
        p = nl ? nl + 1 : end;
    }
    for (int64_t i = 0; i < nbig; i++) { if (big[i].fh) fclose(big[i].fh); free(big[i].name); }
    for (int64_t i = 0; i < nsmall; i++) free(small[i].name);
    for (int64_t i = 0; i < n_small_files; i++) if (small_files[i]) fclose(small_files[i]);
    free(big); free(small); free(small_files);
    return rc;
}

/*
 * The coverage counters of `paffy tile` and `paffy to_bed` -- ONE restatement of each reference function, used by po_tile,
 * po_to_bed and the known-answer probe po_coverage_counts alike, so that the reference's own test of them
 * (tests/paf_unit_test.c:576-603) pins what tile and to_bed run.
 */
/* get_alignment_count_array, impl/paf.c:675-688: the array of the record's query name, zeroed on first sight; NULL (and *mismatch
 * set) when the name is known with another length (the assert of :685) */
static count_array *counts_for(count_array **arrs, int64_t *narr, int64_t *acap, const char *name, int64_t name_len, int64_t qlen, int *mismatch) {
    *mismatch = 0;
    for (int64_t a = 0; a < *narr; a++)
        if ((*arrs)[a].name_len == name_len && memcmp((*arrs)[a].name, name, (size_t)name_len) == 0) {
            if ((*arrs)[a].length != qlen) {
                *mismatch = 1;
                return NULL;
            }
            return &(*arrs)[a];
        }
    if (*narr == *acap) {
        *acap = *acap ? *acap * 2 : 64;
        *arrs = (count_array *)realloc(*arrs, sizeof(count_array) * (size_t)*acap);
    }
    count_array *ca = &(*arrs)[(*narr)++];
    ca->name = name;
    ca->name_len = name_len;
    ca->length = qlen;
    ca->counts = (uint16_t *)calloc((size_t)(qlen > 0 ? qlen : 1), sizeof(uint16_t));
    return ca;
}
/* increase_alignment_level_counts, impl/paf.c:690-709: walks upward from query_start in cigar order whatever the strand; M, = and X
 * bases bump their counter unless it has reached INT16_MAX - 1; I skips its bases, D none. Returns 0, or 1 when one of the
 * reference's asserts (:698 position, :708 end) would fire. */
static int bump_counts(uint16_t *counts, const oop *ops, int64_t n, int64_t qs, int64_t qe, int64_t qlen) {
    int64_t i = qs;
    for (int64_t o = 0; o < n; o++) {
        if (ops[o].op == OP_D) continue;
        if (ops[o].op != OP_I)
            for (int64_t j = 0; j < ops[o].len; j++) {
                const int64_t pos = i + j;
                if (!(pos < qe && pos >= 0 && pos < qlen)) return 1;
                if (counts[pos] < 32767 - 1) counts[pos]++;
            }
        i += ops[o].len;
    }
    return i != qe;
}

int po_tile(const char *in, int64_t in_len, char **out, int64_t *out_len, po_error *err) {
    run_ctx c;
    memset(&c, 0, sizeof(c));
    c.err = err;
    if (err) memset(err, 0, sizeof(*err));
    int rc = PO_OK;
    int64_t nrec = 0, rcap = 1024;
    rec *recs = (rec *)malloc(sizeof(rec) * (size_t)rcap);
    const char *p = in, *end = in + in_len;
    while (p < end) { /* read_pafs(input, 0), impl/paf.c:492-499 */
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        if (nrec == rcap) {
            rcap *= 2;
            recs = (rec *)realloc(recs, sizeof(rec) * (size_t)rcap);
        }
        int64_t aux = 0;
        c.record = nrec;
        rc = parse_line(p, le, 0, &recs[nrec], &aux);
        if (rc) {
            fail(&c, rc, -1, aux);
            break;
        }
        nrec++;
        p = nl ? nl + 1 : end;
    }
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nrec + 1));
    count_array *arrs = NULL;
    int64_t narr = 0, acap = 0;
    int64_t *hist = (int64_t *)calloc(65536 + 2, sizeof(int64_t));
    if (!rc) {
        for (int64_t i = 0; i < nrec; i++) order[i] = i;
        g_sort_recs = recs;
        qsort(order, (size_t)nrec, sizeof(int64_t), cmp_rank);
        for (int64_t k = 0; k < nrec && !rc; k++) {
            rec *r = &recs[order[k]];
            c.record = order[k];
            if (!r->cg_str) { rc = fail(&c, PO_ERR_NULL_CIGAR, 0, 0); break; }
            oop *ops; int64_t n; int present; int64_t aux = 0;
            rc = cigar_from_text(r->cg_str, r->cg_str + r->cg_str_len, &ops, &n, &present, &aux);
            if (rc) { fail(&c, rc, 0, aux); break; }
            int mismatch = 0;
            count_array *ca = counts_for(&arrs, &narr, &acap, r->qname, r->qname_len, r->qlen, &mismatch);
            if (mismatch) { free(ops); rc = fail(&c, PO_ERR_TILE_ASSERT, 0, 1); break; }
            if (bump_counts(ca->counts, ops, n, r->qs, r->qe, r->qlen)) { free(ops); rc = fail(&c, PO_ERR_TILE_ASSERT, 0, 2); break; }
            int64_t i;
            /* get_median_alignment_level, impl/paf_tile.c:36-93 */
            int64_t maxlev = 0, matches = 0;
            for (int64_t q = r->qs; q < r->qe; q++) if (ca->counts[q] > maxlev) maxlev = ca->counts[q];
            memset(hist, 0, sizeof(int64_t) * (size_t)(maxlev + 2));
            i = r->qs;
            for (int64_t o = 0; o < n; o++) {
                if (ops[o].op == OP_D) continue;
                if (ops[o].op != OP_I)
                    for (int64_t j = 0; j < ops[o].len; j++) { hist[ca->counts[i + j]]++; matches++; }
                i += ops[o].len;
            }
            int64_t level = 32767; /* matches == 0 -> INT16_MAX */
            if (matches != 0) {
                int64_t acc = 0, lv;
                for (lv = 0; lv <= maxlev; lv++) {
                    acc += hist[lv];
                    if ((double)acc >= (double)matches / 2.0) break;
                }
                level = lv;
                if (lv > maxlev || !(lv > 0)) { free(ops); rc = fail(&c, PO_ERR_TILE_ASSERT, 0, 3); break; }
            }
            r->tile_level = level;
            free(ops);
        }
        if (!rc)
            for (int64_t k = 0; k < nrec; k++) write_rec(&recs[order[k]], &c.out); /* write_pafs, impl/paf.c:501-505 */
    }
    for (int64_t a = 0; a < narr; a++) free(arrs[a].counts);
    free(arrs);
    free(hist);
    free(order);
    free(recs);
    *out = c.out.p;
    *out_len = c.out.n;
    return rc;
}

/*
 * paffy to_bed, impl/paf_to_bed.c:166-190 (the loop) and :33-55 (write_bed): every record, parsed with its cigar, bumps the
 * counters of its query range (get_alignment_count_array + increase_alignment_level_counts, impl/paf.c:675-709) and, with
 * include_inverted, once more after paf_invert; then the runs of every sequence. The reference iterates a sonLib hash, whose
 * order is not defined: sequences are written in order of first appearance here (tests compare the lines as a set).
 */
static int bed_count(run_ctx *c, rec *r, count_array **arrs, int64_t *narr, int64_t *acap) {
    int mismatch = 0;
    count_array *ca = counts_for(arrs, narr, acap, r->qname, r->qname_len, r->qlen, &mismatch);
    if (mismatch) return fail(c, PO_ERR_TILE_ASSERT, 0, 1);
    /* a record without cigar walks nothing (cigar_count(NULL) == 0, inc/paf.h:75): only the end assert can fire */
    if (bump_counts(ca->counts, r->has_cigar ? r->ops + r->lo : NULL, r->has_cigar ? r->n : 0, r->qs, r->qe, r->qlen)) return fail(c, PO_ERR_TILE_ASSERT, 0, 2);
    return PO_OK;
}

int po_to_bed(const char *in, int64_t in_len, int binary, int exclude_unaligned, int exclude_aligned, int64_t min_size, int include_inverted,
              char **out, int64_t *out_len, po_error *err) {
    run_ctx c;
    memset(&c, 0, sizeof(c));
    c.err = err;
    if (err) memset(err, 0, sizeof(*err));
    int rc = PO_OK;
    count_array *arrs = NULL;
    int64_t narr = 0, acap = 0, nrec = 0;
    /* names point into the input text, which outlives the records */
    const char *p = in, *end = in + in_len;
    while (p < end && !rc) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        rec r;
        int64_t aux = 0;
        c.record = nrec;
        rc = parse_line(p, le, 1, &r, &aux);
        if (rc) {
            fail(&c, rc, -1, aux);
            break;
        }
        rc = bed_count(&c, &r, &arrs, &narr, &acap);
        if (!rc && include_inverted) {
            invert_rec(&r);
            rc = bed_count(&c, &r, &arrs, &narr, &acap);
        }
        rec_free(&r);
        nrec++;
        p = nl ? nl + 1 : end;
    }
    if (!rc)
        for (int64_t a = 0; a < narr; a++) { /* write_bed */
            const count_array *ca = &arrs[a];
            for (int64_t i = 0; i < ca->length;) {
                int64_t j = i + 1;
                while (j < ca->length && !(binary ? (ca->counts[i] > 0) != (ca->counts[j] > 0) : ca->counts[i] != ca->counts[j])) j++;
                if (j - i >= min_size && (ca->counts[i] == 0 ? !exclude_unaligned : !exclude_aligned)) {
                    char tmp[96];
                    ob_bytes(&c.out, ca->name, ca->name_len);
                    int k = snprintf(tmp, sizeof(tmp), " %lld %lld %i\n", (long long)i, (long long)j, binary ? ca->counts[i] > 0 : ca->counts[i]);
                    ob_bytes(&c.out, tmp, k);
                }
                i = j;
            }
        }
    for (int64_t a = 0; a < narr; a++) free(arrs[a].counts);
    free(arrs);
    *out = c.out.p;
    *out_len = c.out.n;
    return rc;
}

void po_free(void *p) { free(p); }

int po_error_exit_status(int32_t code) {
    switch (code) {
        case PO_OK: return 0;
        case PO_ERR_STRAND: case PO_ERR_CIGAR_CHAR:
        case PO_ERR_CHECK_QSTART: case PO_ERR_CHECK_QEND: case PO_ERR_CHECK_TSTART: case PO_ERR_CHECK_TEND:
        case PO_ERR_CHECK_CIGAR_Q: case PO_ERR_CHECK_CIGAR_T:
        case PO_ERR_MISSING_QUERY_SEQ: case PO_ERR_MISSING_TARGET_SEQ:
            return 1; /* st_errAbort / exit(1) */
        case PO_ERR_FEW_FIELDS: case PO_ERR_NULL_CIGAR: case PO_ERR_SEQ_RANGE:
            return 139; /* NULL / wild dereference in the reference */
        default:
            return 134; /* assert -> abort() */
    }
}

/* ------------------------------------------------------------------ */
/* library-level probes for the known-answer tests                      */
/* ------------------------------------------------------------------ */

int64_t po_cigar_parse(const char *cigar, int64_t *lens, int32_t *ops, int64_t cap) {
    oop *o; int64_t n; int present; int64_t bad = 0;
    int rc = cigar_from_text(cigar, cigar + strlen(cigar), &o, &n, &present, &bad);
    if (rc) return -2;
    if (!present) return -1;
    for (int64_t i = 0; i < n && i < cap; i++) { lens[i] = o[i].len; ops[i] = o[i].op; }
    free(o);
    return n;
}

int po_cigar_stats(const char *cigar, int64_t *s, int zero_counts) {
    oop *o; int64_t n; int present; int64_t bad = 0;
    int rc = cigar_from_text(cigar, cigar + strlen(cigar), &o, &n, &present, &bad);
    if (rc) return rc;
    if (zero_counts) memset(s, 0, sizeof(int64_t) * 6);
    for (int64_t i = 0; i < n; i++) { /* paf_stats_calc, impl/paf.c:236-260 */
        if (o[i].op == OP_EQ || o[i].op == OP_M) s[0] += o[i].len;
        else if (o[i].op == OP_X) s[1] += o[i].len;
        else if (o[i].op == OP_I) { s[2]++; s[4] += o[i].len; }
        else { s[3]++; s[5] += o[i].len; }
    }
    free(o);
    return PO_OK;
}

int64_t po_cigar_aligned_bases(const char *cigar) {
    rec r;
    memset(&r, 0, sizeof(r));
    int64_t bad = 0;
    if (cigar_from_text(cigar, cigar + strlen(cigar), &r.ops, &r.n, &r.has_cigar, &bad)) return -1;
    int64_t a = aligned_bases(&r);
    rec_free(&r);
    return a;
}

int po_trim_ends_line(const char *line, int64_t line_len, int64_t end_bases, char **out, int64_t *out_len) {
    rec r; int64_t aux = 0;
    const char *e = line + line_len;
    if (line_len && e[-1] == '\n') e--;
    int rc = parse_line(line, e, 1, &r, &aux);
    obuf b = {0, 0, 0};
    if (!rc) rc = trim_ends(&r, end_bases);
    if (!rc) write_rec(&r, &b);
    rec_free(&r);
    *out = b.p;
    *out_len = b.n;
    return rc;
}

/*
 * paf_pretty_print, impl/paf.c:262-316, of one PAF line (already carrying the cigar to show: `paffy view` encodes the mismatches
 * first, impl/paf_view.c:167). The stats line is formatted with the reference's own float expression; with include_alignment the
 * three rows per 150 columns follow. Sequences are NUL-terminated strings as the reference holds them.
 */
int po_pretty_print(const char *line, int64_t line_len, const char *query_seq, const char *target_seq, int include_alignment, char **out,
                    int64_t *out_len) {
    rec r; int64_t aux = 0;
    const char *e = line + line_len;
    if (line_len && e[-1] == '\n') e--;
    int rc = parse_line(line, e, 1, &r, &aux);
    obuf b = {0, 0, 0};
    if (!rc) {
        int64_t s[6] = {0, 0, 0, 0, 0, 0};
        for (int64_t i = 0; r.has_cigar && i < r.n; i++) { /* paf_stats_calc, impl/paf.c:236-260 */
            const oop *o = &r.ops[r.lo + i];
            if (o->op == OP_EQ || o->op == OP_M) s[0] += o->len;
            else if (o->op == OP_X) s[1] += o->len;
            else if (o->op == OP_I) { s[2]++; s[4] += o->len; }
            else { s[3]++; s[5] += o->len; }
        }
        int need = snprintf(NULL, 0, "Query:%.*s\tQ-start:%" PRIi64 "\tQ-length:%" PRIi64 "\tTarget:%.*s\tT-start:%" PRIi64 "\tT-length:%" PRIi64
                            "\tSame-strand:%i\tScore:%" PRIi64 "\tIdentity:%f\tIdentity-with-gaps%f\tAligned-bases:%" PRIi64 "\tQuery-inserts:%" PRIi64
                            "\tQuery-deletes:%" PRIi64 "\n",
                            (int)r.qname_len, r.qname, r.qs, r.qe - r.qs, (int)r.tname_len, r.tname, r.ts, r.te - r.ts, r.same_strand, r.score,
                            (float)s[0] / (s[0] + s[1]), (float)s[0] / (s[0] + s[1] + s[4] + s[5]), s[0] + s[1], s[2], s[3]);
        ob_need(&b, need + 1);
        snprintf(b.p + b.n, (size_t)need + 1, "Query:%.*s\tQ-start:%" PRIi64 "\tQ-length:%" PRIi64 "\tTarget:%.*s\tT-start:%" PRIi64 "\tT-length:%" PRIi64
                 "\tSame-strand:%i\tScore:%" PRIi64 "\tIdentity:%f\tIdentity-with-gaps%f\tAligned-bases:%" PRIi64 "\tQuery-inserts:%" PRIi64
                 "\tQuery-deletes:%" PRIi64 "\n",
                 (int)r.qname_len, r.qname, r.qs, r.qe - r.qs, (int)r.tname_len, r.tname, r.ts, r.te - r.ts, r.same_strand, r.score,
                 (float)s[0] / (s[0] + s[1]), (float)s[0] / (s[0] + s[1] + s[4] + s[5]), s[0] + s[1], s[2], s[3]);
        b.n += need;
        if (include_alignment) { /* impl/paf.c:283-315 */
            int64_t max_len = r.qe - r.qs + r.te - r.ts;
            char *qa = (char *)malloc((size_t)max_len + 1), *ta = (char *)malloc((size_t)max_len + 1), *sa = (char *)malloc((size_t)max_len + 1);
            int64_t i = 0, j = r.ts, k = 0;
            for (int64_t ci = 0; r.has_cigar && ci < r.n; ci++) {
                const oop *o = &r.ops[r.lo + ci];
                for (int64_t l = 0; l < o->len; l++) {
                    char m = '-', n = '-';
                    if (o->op != OP_I) m = target_seq[j++];
                    if (o->op != OP_D) {
                        if (r.same_strand) n = query_seq[r.qs + i++];
                        else n = rc_char(query_seq[r.qe - (++i)]);
                    }
                    ta[k] = m;
                    qa[k] = n;
                    sa[k++] = up(m) == up(n) ? '*' : ' ';
                }
            }
            const int64_t window = 150;
            for (int64_t l = 0; l < k; l += window) {
                int64_t hi = l + window < k ? l + window : k;
                ob_bytes(&b, ta + l, hi - l); ob_char(&b, '\n');
                ob_bytes(&b, qa + l, hi - l); ob_char(&b, '\n');
                ob_bytes(&b, sa + l, hi - l); ob_char(&b, '\n');
            }
            free(qa); free(ta); free(sa);
        }
    }
    rec_free(&r);
    *out = b.p;
    *out_len = b.n;
    return rc;
}

int64_t po_coverage_counts(const char *in, int64_t in_len, const char *name, uint16_t *counts, int64_t len) {
    /* the records of `in` through counts_for + bump_counts, the functions po_tile and po_to_bed use; the counters of sequence `name`
     * are copied out. Returns the number of records applied to it, or -1 - k when record k would trip an assert. */
    const char *p = in, *end = in + in_len;
    int64_t applied = 0, nl_name = (int64_t)strlen(name), rec_no = 0, rc = 0;
    count_array *arrs = NULL;
    int64_t narr = 0, acap = 0;
    rec *kept = NULL; /* the names point into the parsed records: keep them until the end */
    int64_t nkept = 0, kcap = 0;
    while (p < end && rc == 0) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        rec r; int64_t aux = 0;
        if (parse_line(p, le, 1, &r, &aux) == PO_OK) {
            int mismatch = 0;
            count_array *ca = counts_for(&arrs, &narr, &acap, r.qname, r.qname_len, r.qlen, &mismatch);
            if (mismatch || bump_counts(ca->counts, r.has_cigar ? r.ops + r.lo : NULL, r.has_cigar ? r.n : 0, r.qs, r.qe, r.qlen)) rc = -1 - rec_no;
            else if (r.qname_len == nl_name && memcmp(r.qname, name, (size_t)nl_name) == 0) applied++;
            if (nkept == kcap) { kcap = kcap ? kcap * 2 : 16; kept = (rec *)realloc(kept, sizeof(rec) * (size_t)kcap); }
            kept[nkept++] = r;
        }
        rec_no++;
        p = nl ? nl + 1 : end;
    }
    for (int64_t a = 0; a < narr; a++) {
        if (rc == 0 && arrs[a].name_len == nl_name && memcmp(arrs[a].name, name, (size_t)nl_name) == 0)
            for (int64_t i = 0; i < len && i < arrs[a].length; i++) counts[i] = arrs[a].counts[i];
        free(arrs[a].counts);
    }
    free(arrs);
    for (int64_t k = 0; k < nkept; k++) rec_free(&kept[k]);
    free(kept);
    return rc ? rc : applied;
}
