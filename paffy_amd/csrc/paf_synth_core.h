/*
 * paf_synth_core.h -- deterministic synthetic PAF workload.
 *
 * Record r of a workload is a pure function of (seed, r): a counter-based RNG (splitmix64
 * finaliser) and integer-only distributions, so the host (C) and the device (HIP) builds of
 * this header produce the same bytes. Shapes follow the reference's fixture
 * (/root/reference/tests/human_chimp.paf): 24-field minimap2-style lines, ops alternating
 * M and I|D, starting and ending with M, ~24 % '-' strand, ignored tags present so that the
 * parser's tag skipping (impl/paf.c:181-206) is exercised.
 *
 * Distributions (integer only):
 *   E(x)      ~ -log2(U) in 16.16 fixed point from the leading-zero count of a 64-bit draw and
 *               a linear mantissa; scaled(m, x) = floor(m * E(x) * ln2) is ~exponential, mean m.
 *   #M ops k  = exponential body + a heavy tail, capped at 2^19 (psynth_num_match_ops);  ops = 2k-1
 *   M length  = 1 + scaled(39)   indel length = 1 + scaled(2)   I vs D: one bit
 */
#ifndef PAF_SYNTH_CORE_H_
#define PAF_SYNTH_CORE_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define PSYNTH_HD __host__ __device__
#else
#define PSYNTH_HD
#endif

typedef struct {
    uint64_t seed;
    uint32_t mean_ops;  /* mean cigar ops per record (512 for cfg2, 2048 for cfg3) */
    uint32_t n_contigs; /* contigs per genome (24) */
} psynth_cfg;

PSYNTH_HD static inline uint64_t psynth_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
PSYNTH_HD static inline uint64_t psynth_rnd(uint64_t rkey, uint64_t k) { return psynth_mix(rkey + k * 0xD1B54A32D192ED03ull); }
PSYNTH_HD static inline uint64_t psynth_rkey(uint64_t seed, uint64_t r) { return psynth_mix(seed ^ (r * 0x9E3779B97F4A7C15ull)); }

PSYNTH_HD static inline int psynth_clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return __builtin_clzll(x);
#endif
}

/* floor(m * -ln(U)) for a 64-bit draw x, integer arithmetic only. */
PSYNTH_HD static inline uint64_t psynth_scaled(uint64_t m, uint64_t x) {
    x |= 1;
    int lz = psynth_clz64(x);
    uint64_t mant = lz == 63 ? 0 : (x << (lz + 1)) >> 48; /* 16 bits below the leading one */
    uint64_t e2 = ((uint64_t)(lz + 1) << 16) - mant;       /* -log2(U), 16.16 */
    return (m * e2 * 45426ull) >> 32;                      /* * ln2 (45426/65536) */
}

PSYNTH_HD static inline int64_t psynth_contig_len(uint64_t seed, int genome, uint32_t c) {
    return 50000000ll + (int64_t)(psynth_mix(seed ^ (0xC0117100ull + (uint64_t)genome * 4096 + c)) % 200000001ull);
}

/*
 * Number of M ops of a record (ops = 2k - 1). Round 4: the op count has its heavy tail again ("heavy tail capped at 2^20").
 * The reference's fixture (tests/human_chimp.paf: mean 1 785 ops) has 5 % of its records above six times the mean and its longest at
 * 11.6 times; an exponential alone has 0.25 % and, in 207 records, nothing above 7. So: fifteen records in sixteen come from an
 * exponential body of mean 81/128 mean_k, one in sixteen is a tail record of k = 4 mean_k + Exp(5/2 mean_k) -- 2.8 % of the records
 * above 6 x the mean, 0.8 % above 9 x, the longest of 207 at about 10 x, the longest of a 10 M-record stream at about 37 x (76 k ops at
 * mean 2 048) -- and the overall mean stays mean_k (15/16 * 81/128 + 1/16 * 6.5 = 0.9995). Cap: 2^19 M ops = 2^20 - 1 ops.
 * (Rounds 1-3: one exponential of mean mean_k capped at 8 mean_k -- no cfg3 record above 16 383 ops.)
 */
PSYNTH_HD static inline uint32_t psynth_num_match_ops(const psynth_cfg *c, uint64_t rkey) {
    const uint64_t mean_k = ((uint64_t)c->mean_ops + 1) / 2;
    const uint64_t x = psynth_rnd(rkey, 0);
    uint64_t k;
    if ((psynth_rnd(rkey, 14) & 15u) == 0) k = 4 * mean_k + psynth_scaled((5 * mean_k) >> 1, x);
    else k = 1 + psynth_scaled(mean_k ? ((mean_k - 1) * 81) >> 7 : 0, x);
    const uint64_t cap = 1ull << 19;
    if (k < 1) k = 1;
    return (uint32_t)(k > cap ? cap : k);
}

/* op j of a record: even j = M, odd j = I or D; returns length, sets *op (0 M, 1 I, 2 D). */
PSYNTH_HD static inline int64_t psynth_op(uint64_t rkey, uint32_t j, int *op) {
    uint64_t x = psynth_rnd(rkey, 16 + (uint64_t)j);
    if ((j & 1) == 0) {
        *op = 0;
        return 1 + (int64_t)psynth_scaled(39, x);
    }
    *op = (x & 1) ? 1 : 2;
    return 1 + (int64_t)psynth_scaled(2, x);
}

/* ---- byte emission: with out == NULL only the length is counted ---- */

PSYNTH_HD static inline int64_t psynth_put_int(char *out, int64_t pos, int64_t v) {
    char tmp[24];
    int k = 0;
    uint64_t u = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
    if (u == 0) tmp[k++] = '0';
    while (u) {
        tmp[k++] = (char)('0' + (int)(u % 10));
        u /= 10;
    }
    if (v < 0) {
        if (out) out[pos] = '-';
        pos++;
    }
    while (k) {
        --k;
        if (out) out[pos] = tmp[k];
        pos++;
    }
    return pos;
}
PSYNTH_HD static inline int64_t psynth_put_str(char *out, int64_t pos, const char *s) {
    while (*s) {
        if (out) out[pos] = *s;
        pos++;
        s++;
    }
    return pos;
}
PSYNTH_HD static inline int64_t psynth_put_tag(char *out, int64_t pos, const char *tag, int64_t v) {
    pos = psynth_put_str(out, pos, tag);
    return psynth_put_int(out, pos, v);
}

/* What varies between the workloads: where a record's ops come from and where it lies on its two contigs. */
typedef struct {
    uint64_t rkey;  /* the record's own draws (tags) */
    uint64_t opkey; /* op j of the record = psynth_op(opkey, op0 + j) */
    uint64_t op0;
    uint32_t n_ops, qc, tc;
    int minus;
    int64_t qlen, tlen, qs, ts;
} psynth_shape;

/* One '\n'-terminated PAF line for `sh` at out[0..); returns its length (out == NULL: the length only). */
PSYNTH_HD static inline int64_t psynth_emit_text(const psynth_shape *sh, char *out) {
    const uint64_t rkey = sh->rkey;
    int64_t sum_m = 0, sum_i = 0, sum_d = 0;
    for (uint32_t j = 0; j < sh->n_ops; j++) {
        int op;
        int64_t len = psynth_op(sh->opkey, sh->op0 + j, &op);
        if (op == 0) sum_m += len;
        else if (op == 1) sum_i += len;
        else sum_d += len;
    }
    const int64_t qspan = sum_m + sum_i, tspan = sum_m + sum_d;
    const int64_t qs = sh->qs, ts = sh->ts;
    uint64_t tpr = psynth_rnd(rkey, 6) % 100;

    int64_t p = 0;
    p = psynth_put_str(out, p, "hs.chr");
    p = psynth_put_int(out, p, sh->qc + 1);
    p = psynth_put_tag(out, p, "\t", sh->qlen);
    p = psynth_put_tag(out, p, "\t", qs);
    p = psynth_put_tag(out, p, "\t", qs + qspan);
    p = psynth_put_str(out, p, sh->minus ? "\t-\tpt.chr" : "\t+\tpt.chr");
    p = psynth_put_int(out, p, sh->tc + 1);
    p = psynth_put_tag(out, p, "\t", sh->tlen);
    p = psynth_put_tag(out, p, "\t", ts);
    p = psynth_put_tag(out, p, "\t", ts + tspan);
    p = psynth_put_tag(out, p, "\t", sum_m);
    p = psynth_put_tag(out, p, "\t", sum_m + sum_i + sum_d);
    p = psynth_put_str(out, p, "\t60");
    p = psynth_put_tag(out, p, "\tNM:i:", sum_i + sum_d);
    p = psynth_put_tag(out, p, "\tms:i:", (int64_t)(psynth_rnd(rkey, 7) % 10000000ull));
    p = psynth_put_tag(out, p, "\tAS:i:", 1000 + (int64_t)(psynth_rnd(rkey, 8) % 9999001ull));
    p = psynth_put_str(out, p, "\tnn:i:0\ttp:A:");
    p = psynth_put_str(out, p, tpr < 72 ? "P" : (tpr < 95 ? "S" : "I"));
    p = psynth_put_tag(out, p, "\tcm:i:", (int64_t)(psynth_rnd(rkey, 9) % 1000000ull));
    p = psynth_put_tag(out, p, "\ts1:i:", 1000 + (int64_t)(psynth_rnd(rkey, 10) % 9999001ull));
    p = psynth_put_tag(out, p, "\ts2:i:", (int64_t)(psynth_rnd(rkey, 11) % 1000000ull));
    p = psynth_put_str(out, p, "\tde:f:0.");
    {
        int64_t de = (int64_t)(psynth_rnd(rkey, 12) % 10000ull);
        if (de < 1000) p = psynth_put_str(out, p, "0");
        if (de < 100) p = psynth_put_str(out, p, "0");
        if (de < 10) p = psynth_put_str(out, p, "0");
        p = psynth_put_int(out, p, de);
    }
    p = psynth_put_tag(out, p, "\trl:i:", (int64_t)(psynth_rnd(rkey, 13) % 10000000ull));
    p = psynth_put_str(out, p, "\tcg:Z:");
    for (uint32_t j = 0; j < sh->n_ops; j++) {
        int op;
        int64_t len = psynth_op(sh->opkey, sh->op0 + j, &op);
        p = psynth_put_int(out, p, len);
        if (out) out[p] = op == 0 ? 'M' : (op == 1 ? 'I' : 'D');
        p++;
    }
    if (out) out[p] = '\n';
    p++;
    return p;
}

/*
 * Writes record r (one '\n'-terminated PAF line) at out[0..) and returns its length; with
 * out == NULL returns the length only. Coordinates are consistent with the cigar so that
 * paf_check (impl/paf.c:427-461) passes, and all op lengths are >= 1 (impl/paf.c:635).
 */
PSYNTH_HD static inline int64_t psynth_emit_record(const psynth_cfg *c, uint64_t r, char *out) {
    psynth_shape sh;
    sh.rkey = sh.opkey = psynth_rkey(c->seed, r);
    sh.op0 = 0;
    const uint64_t rkey = sh.rkey;
    uint32_t k = psynth_num_match_ops(c, rkey);
    sh.n_ops = 2 * k - 1;
    int64_t sum_m = 0, sum_i = 0, sum_d = 0;
    for (uint32_t j = 0; j < sh.n_ops; j++) {
        int op;
        int64_t len = psynth_op(rkey, j, &op);
        if (op == 0) sum_m += len;
        else if (op == 1) sum_i += len;
        else sum_d += len;
    }
    sh.qc = (uint32_t)(psynth_rnd(rkey, 1) % c->n_contigs);
    sh.tc = (uint32_t)(psynth_rnd(rkey, 2) % c->n_contigs);
    sh.minus = (psynth_rnd(rkey, 3) % 100) < 24;
    sh.qlen = psynth_contig_len(c->seed, 0, sh.qc);
    sh.tlen = psynth_contig_len(c->seed, 1, sh.tc);
    sh.qs = (int64_t)(psynth_rnd(rkey, 4) % (uint64_t)(sh.qlen - (sum_m + sum_i) + 1));
    sh.ts = (int64_t)(psynth_rnd(rkey, 5) % (uint64_t)(sh.tlen - (sum_m + sum_d) + 1));
    return psynth_emit_text(&sh, out);
}

/* ------------------------------------------------------------------------------------------------
 * cfg4 workload (records + two genomes for add_mismatches). Each contig pair c
 * (hs.chr<c+1>, pt.chr<c+1>) has ONE master alignment: ops psynth_op(mkey(c), j), j = 0 .. n_ops-1,
 * covering the target contig from base 0. A record is a window of consecutive master ops that starts
 * and ends on an M op, so that its columns really pair homologous bases:
 *   target base (c, p)      = i.i.d. ACGT from a hash of (seed, c, p / 32), 0.5 % lower case
 *   query base at x (in alignment orientation) = the paired target base, substituted with p = 2 %,
 *                             inside an M op; an i.i.d. base inside an I op; 0.5 % lower case
 *   contigs with c % 4 == 3 hold the query reverse-complemented: all their records are '-' strand.
 * Every PSYNTH4_G ops the running (query, target) offsets are kept as checkpoints.
 * ---------------------------------------------------------------------------------------------- */
#define PSYNTH4_G 256

typedef struct {
    uint64_t seed;
    uint32_t mean_ops;
    uint32_t n_contigs;
    int64_t tlen_min;  /* target contig length = tlen_min + hash % (tlen_span + 1); >= 2048 */
    int64_t tlen_span;
} psynth4_cfg;

typedef struct {
    int64_t tlen, qlen;
    uint64_t n_ops;     /* odd: the master alignment starts and ends with M */
    uint64_t ckpt_base; /* first checkpoint of this contig in the checkpoint arrays */
    uint64_t ckpt_cap;
} psynth4_contig;

typedef struct {
    const psynth4_contig *contigs;
    const int64_t *ckpt_q, *ckpt_t; /* offsets in front of op i * PSYNTH4_G */
} psynth4_tab;

PSYNTH_HD static inline uint64_t psynth4_mkey(uint64_t seed, uint32_t c) { return psynth_mix(seed ^ (0x4D41535445520000ull + c)); }
PSYNTH_HD static inline int64_t psynth4_tlen(const psynth4_cfg *c, uint32_t k) {
    return c->tlen_min + (int64_t)(psynth_mix(c->seed ^ (0xC0117100ull + 4096 + k)) % (uint64_t)(c->tlen_span + 1));
}
/* checkpoints kept per contig; the master alignment stops at ckpt_cap * PSYNTH4_G ops at the latest */
PSYNTH_HD static inline uint64_t psynth4_ckpt_cap(int64_t tlen) { return (uint64_t)tlen / (8 * PSYNTH4_G) + 8; }
PSYNTH_HD static inline int psynth4_minus(uint32_t c) { return (c & 3u) == 3u; }

/* 0..3 = ACGT index of target base p of contig c; *lower = printed in lower case */
PSYNTH_HD static inline uint32_t psynth4_tbase(uint64_t seed, uint32_t c, int64_t p, int *lower) {
    const uint64_t h = psynth_mix(seed ^ 0x7A46E70000000000ull ^ ((uint64_t)c << 40) ^ (uint64_t)(p >> 5));
    const uint64_t hl = psynth_mix(h ^ 0x10CA5Eull);
    *lower = (hl % 6 == 0) && (((hl >> 8) & 31u) == (uint64_t)(p & 31));
    return (uint32_t)(h >> (2 * (p & 31))) & 3u;
}
/* query base at alignment-orientation offset x of contig c: paired with target base tp (tp < 0: inside an I op) */
PSYNTH_HD static inline uint32_t psynth4_qbase(uint64_t seed, uint32_t c, int64_t x, int64_t tp, int *lower) {
    const uint64_t h = psynth_mix(seed ^ 0x9E47000000000000ull ^ ((uint64_t)c << 40) ^ (uint64_t)x);
    *lower = ((h >> 16) % 200) == 0;
    if (tp < 0) return (uint32_t)(h >> 8) & 3u;
    int tl;
    const uint32_t tb = psynth4_tbase(seed, c, tp, &tl);
    return (h % 50 == 0) ? ((tb + 1 + (uint32_t)((h >> 8) % 3)) & 3u) : tb;
}
PSYNTH_HD static inline char psynth4_letter(uint32_t b, int lower, int complement) {
    const char *up = "ACGT", *lo = "acgt";
    if (complement) b = 3u - b;
    return lower ? lo[b] : up[b];
}

/* (query, target) offsets in front of master op j of contig c */
PSYNTH_HD static inline void psynth4_prefix(const psynth4_cfg *cfg, const psynth4_tab *tab, uint32_t c, uint64_t j, int64_t *q, int64_t *t) {
    const psynth4_contig *ct = &tab->contigs[c];
    const uint64_t b = j / PSYNTH4_G;
    int64_t qq = tab->ckpt_q[ct->ckpt_base + b], tt = tab->ckpt_t[ct->ckpt_base + b];
    const uint64_t mkey = psynth4_mkey(cfg->seed, c);
    for (uint64_t i = b * PSYNTH4_G; i < j; i++) {
        int op;
        int64_t len = psynth_op(mkey, i, &op);
        if (op != 2) qq += len;
        if (op != 1) tt += len;
    }
    *q = qq;
    *t = tt;
}

/* cfg4 record r: a window of the master alignment of one contig pair. */
PSYNTH_HD static inline int64_t psynth4_emit_record(const psynth4_cfg *cfg, const psynth4_tab *tab, uint64_t r, char *out) {
    psynth_shape sh;
    sh.rkey = psynth_rkey(cfg->seed, r);
    const uint32_t c = (uint32_t)(psynth_rnd(sh.rkey, 1) % cfg->n_contigs);
    const psynth4_contig *ct = &tab->contigs[c];
    psynth_cfg base;
    base.seed = cfg->seed;
    base.mean_ops = cfg->mean_ops;
    base.n_contigs = cfg->n_contigs;
    uint64_t k = psynth_num_match_ops(&base, sh.rkey);
    if (2 * k - 1 > ct->n_ops) k = (ct->n_ops + 1) / 2;
    sh.n_ops = (uint32_t)(2 * k - 1);
    sh.opkey = psynth4_mkey(cfg->seed, c);
    sh.op0 = 2 * (psynth_rnd(sh.rkey, 4) % ((ct->n_ops - sh.n_ops) / 2 + 1));
    sh.qc = sh.tc = c;
    sh.minus = psynth4_minus(c);
    sh.qlen = ct->qlen;
    sh.tlen = ct->tlen;
    int64_t q0, t0;
    psynth4_prefix(cfg, tab, c, sh.op0, &q0, &t0);
    sh.ts = t0;
    if (!sh.minus) {
        sh.qs = q0;
    } else { /* the stored query is the reverse complement: [q0, q0 + span) lies at [qlen - q0 - span, qlen - q0) */
        int64_t span = 0;
        for (uint32_t j = 0; j < sh.n_ops; j++) {
            int op;
            int64_t len = psynth_op(sh.opkey, sh.op0 + j, &op);
            if (op != 2) span += len;
        }
        sh.qs = ct->qlen - q0 - span;
    }
    return psynth_emit_text(&sh, out);
}

#endif
