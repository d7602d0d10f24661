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
 *   #M ops k  = 1 + scaled((mean_ops+1)/2 - 1), capped at min(8*mean_k, 25000);  ops = 2k-1
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

PSYNTH_HD static inline uint32_t psynth_num_match_ops(const psynth_cfg *c, uint64_t rkey) {
    uint64_t mean_k = ((uint64_t)c->mean_ops + 1) / 2;
    uint64_t k = 1 + psynth_scaled(mean_k ? mean_k - 1 : 0, psynth_rnd(rkey, 0));
    uint64_t cap = 8 * mean_k;
    if (cap > 25000) cap = 25000;
    if (cap < 1) cap = 1;
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

/*
 * Writes record r (one '\n'-terminated PAF line) at out[0..) and returns its length; with
 * out == NULL returns the length only. Coordinates are consistent with the cigar so that
 * paf_check (impl/paf.c:427-461) passes, and all op lengths are >= 1 (impl/paf.c:635).
 */
PSYNTH_HD static inline int64_t psynth_emit_record(const psynth_cfg *c, uint64_t r, char *out) {
    uint64_t rkey = psynth_rkey(c->seed, r);
    uint32_t k = psynth_num_match_ops(c, rkey);
    uint32_t n_ops = 2 * k - 1;
    int64_t sum_m = 0, sum_i = 0, sum_d = 0;
    for (uint32_t j = 0; j < n_ops; j++) {
        int op;
        int64_t len = psynth_op(rkey, j, &op);
        if (op == 0) sum_m += len;
        else if (op == 1) sum_i += len;
        else sum_d += len;
    }
    uint32_t qc = (uint32_t)(psynth_rnd(rkey, 1) % c->n_contigs), tc = (uint32_t)(psynth_rnd(rkey, 2) % c->n_contigs);
    int minus = (psynth_rnd(rkey, 3) % 100) < 24;
    int64_t qlen = psynth_contig_len(c->seed, 0, qc), tlen = psynth_contig_len(c->seed, 1, tc);
    int64_t qspan = sum_m + sum_i, tspan = sum_m + sum_d;
    int64_t qs = (int64_t)(psynth_rnd(rkey, 4) % (uint64_t)(qlen - qspan + 1));
    int64_t ts = (int64_t)(psynth_rnd(rkey, 5) % (uint64_t)(tlen - tspan + 1));
    uint64_t tpr = psynth_rnd(rkey, 6) % 100;

    int64_t p = 0;
    p = psynth_put_str(out, p, "hs.chr");
    p = psynth_put_int(out, p, qc + 1);
    p = psynth_put_tag(out, p, "\t", qlen);
    p = psynth_put_tag(out, p, "\t", qs);
    p = psynth_put_tag(out, p, "\t", qs + qspan);
    p = psynth_put_str(out, p, minus ? "\t-\tpt.chr" : "\t+\tpt.chr");
    p = psynth_put_int(out, p, tc + 1);
    p = psynth_put_tag(out, p, "\t", tlen);
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
    for (uint32_t j = 0; j < n_ops; j++) {
        int op;
        int64_t len = psynth_op(rkey, j, &op);
        p = psynth_put_int(out, p, len);
        if (out) out[p] = op == 0 ? 'M' : (op == 1 ? 'I' : 'D');
        p++;
    }
    if (out) out[p] = '\n';
    p++;
    return p;
}

#endif
