/*
 * pretty_kernel.h -- the base-level rows of paf_pretty_print (impl/paf.c:283-315) for records of a planned batch.
 *
 * The reference fills three arrays column by column (target base or '-', query base or '-', '*' where the two agree
 * without case) and prints them in windows of 150 columns, three lines per window. Here one workgroup takes a record:
 * its ops (as the plan's stages left them, read through the same view as the emit pass) go through LDS 256 at a time,
 * three exclusive sums give each op its first column, target base and query base, and the threads then fill the
 * columns of the chunk, each finding its op by bisection of the 256 column starts. A column's three bytes land at
 *     row_off[rec] + (col / 150) * 453 + col % 150 + {0, w + 1, 2 w + 2},   w = the window's width (150, the last one less)
 * so the block is the reference's text byte for byte. The bases are the sequences as loaded (seq_raw, case kept); on the
 * - strand the query is read backwards through stString_reverseComplementChar (A<->T, C<->G in both cases, others kept).
 */
#pragma once

#define PRETTY_WINDOW 150
#define PRETTY_NT 256

struct PrettyParams {
    const RecMeta *meta;
    const RecPlan *plan;
    const uint32_t *status;
    const uint64_t *arena;
    const uint64_t *arena_off;
    const uint32_t *ops_mirror;
    const uint8_t *seq_raw;
    const SeqEntry *seqs;
    const int32_t *rec_qseq, *rec_tseq;
    uint32_t first;             /* the batch's record of row_off[0] */
    const int64_t *row_off;     /* count + 1 offsets into out */
    uint8_t *out;
    unsigned long long *err;    /* min over failing records of rec << 8 | code */
};

__device__ __forceinline__ uint8_t pretty_rc(uint8_t c) {
    switch (c) {
        case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C';
        case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c';
        default: return c;
    }
}
__device__ __forceinline__ uint8_t pretty_up(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; } /* toupper, C locale */

/* exclusive sum over the workgroup of three values at once; total[] = the sums */
__device__ __forceinline__ void pretty_scan3(int64_t v[3], int64_t (*tmp)[3], int64_t total[3]) {
    const uint32_t t = threadIdx.x;
    for (int k = 0; k < 3; k++) tmp[t][k] = v[k];
    __syncthreads();
    for (uint32_t d = 1; d < PRETTY_NT; d <<= 1) {
        int64_t a[3] = {0, 0, 0};
        if (t >= d)
            for (int k = 0; k < 3; k++) a[k] = tmp[t - d][k];
        __syncthreads();
        for (int k = 0; k < 3; k++) tmp[t][k] += a[k];
        __syncthreads();
    }
    for (int k = 0; k < 3; k++) {
        total[k] = tmp[PRETTY_NT - 1][k];
        v[k] = tmp[t][k] - v[k];
    }
    __syncthreads();
}

/* the record's columns: the sum of its op lengths */
__device__ __forceinline__ int64_t pretty_columns(const PrettyParams &P, const RecMeta &m, const RecPlan &pl, uint32_t rec, int64_t (*tmp)[3]) {
    const bool rev = (pl.flags & 1u) != 0, wide = (P.status[rec] >> 16) == KLASS_ARENA;
    const uint64_t *ops8 = P.arena + P.arena_off[rec];
    const uint32_t *ops4 = (pl.flags & 0x20000u) ? reinterpret_cast<const uint32_t *>(P.arena + P.arena_off[rec]) : P.ops_mirror + (m.cg_off >> 1);
    const bool half = (pl.flags & 0x60000u) == 0x40000u; /* the mirror holds 2-byte words (record_kernel.h: emit_ops_half) */
    int64_t part = 0;
    for (uint32_t i = threadIdx.x; i < pl.n; i += PRETTY_NT) {
        const uint32_t raw = rev ? pl.lo + pl.n - 1 - i : pl.lo + i;
        const uint32_t w4 = half ? (uint32_t) reinterpret_cast<const uint16_t *>(ops4)[raw] : ops4[raw];
        int64_t len = wide ? (int64_t)ops8[raw] >> 8 : (int64_t)(w4 >> 3);
        if (raw == pl.lo) len -= pl.sub_lo;
        if (raw == pl.lo + pl.n - 1) len -= pl.sub_hi;
        part += len;
    }
    int64_t v[3] = {part, 0, 0}, tot[3];
    pretty_scan3(v, tmp, tot);
    return tot[0];
}

/* bytes of a record's block: three lines per window of 150 columns */
__global__ __launch_bounds__(PRETTY_NT) void k_pretty_size(PrettyParams P, int64_t *bytes) {
    __shared__ int64_t tmp[PRETTY_NT][3];
    const uint32_t rec = P.first + blockIdx.x;
    const RecMeta m = P.meta[rec];
    const RecPlan pl = P.plan[rec];
    int64_t cols = 0;
    if ((pl.flags & 8u) && pl.n != 0) cols = pretty_columns(P, m, pl, rec, tmp);
    if (cols < 0) cols = 0;
    if (threadIdx.x == 0) bytes[blockIdx.x] = 3 * cols + 3 * ((cols + PRETTY_WINDOW - 1) / PRETTY_WINDOW);
}

__global__ __launch_bounds__(PRETTY_NT) void k_pretty_rows(PrettyParams P) {
    __shared__ int64_t tmp[PRETTY_NT][3];
    __shared__ int64_t col0[PRETTY_NT + 1], tb0[PRETTY_NT], qb0[PRETTY_NT];
    __shared__ uint8_t opk[PRETTY_NT];
    const uint32_t rec = P.first + blockIdx.x, t = threadIdx.x;
    const RecMeta m = P.meta[rec];
    const RecPlan pl = P.plan[rec];
    if (!(pl.flags & 8u) || pl.n == 0) return; /* cigar_count(NULL) == 0: no rows */
    const bool swapped = (pl.flags & 4u) != 0, rev = (pl.flags & 1u) != 0, swp = (pl.flags & 2u) != 0;
    const int32_t qi = swapped ? P.rec_tseq[rec] : P.rec_qseq[rec], ti = swapped ? P.rec_qseq[rec] : P.rec_tseq[rec];
    if (qi < 0 || ti < 0) {
        if (t == 0) atomicMin(P.err, ((unsigned long long)rec << 8) | (unsigned)(qi < 0 ? PAFFY_ERR_MISSING_QUERY_SEQ : PAFFY_ERR_MISSING_TARGET_SEQ));
        return;
    }
    const SeqEntry qs = P.seqs[qi], ts = P.seqs[ti];
    if (pl.qs < 0 || pl.ts < 0 || pl.qe > qs.len || pl.te > ts.len) { /* the reference would read outside the strings */
        if (t == 0) atomicMin(P.err, ((unsigned long long)rec << 8) | (unsigned)PAFFY_ERR_SEQ_RANGE);
        return;
    }
    const uint8_t *Q = P.seq_raw + qs.off, *T = P.seq_raw + ts.off;
    const bool same = m.same_strand != 0;
    const bool wide = (P.status[rec] >> 16) == KLASS_ARENA; /* 8-byte ops in the arena */
    const uint64_t *ops8 = P.arena + P.arena_off[rec];
    const uint32_t *ops4 = (pl.flags & 0x20000u) ? reinterpret_cast<const uint32_t *>(P.arena + P.arena_off[rec]) : P.ops_mirror + (m.cg_off >> 1);
    const bool half = (pl.flags & 0x60000u) == 0x40000u; /* the mirror holds 2-byte words (record_kernel.h: emit_ops_half) */
    /* the width of the last window needs the number of columns: one pass over the ops first */
    const int64_t cols_total = pretty_columns(P, m, pl, rec, tmp);
    uint8_t *out = P.out + P.row_off[blockIdx.x];
    int64_t col_base = 0, t_base = pl.ts, q_base = 0; /* q_base: query bases consumed so far (i of impl/paf.c:290) */
    for (uint32_t c0 = 0; c0 < pl.n; c0 += PRETTY_NT) {
        const uint32_t i = c0 + t;
        int64_t len = 0;
        int op = OP_M;
        if (i < pl.n) {
            const uint32_t raw = rev ? pl.lo + pl.n - 1 - i : pl.lo + i;
            if (wide) {
                len = (int64_t)ops8[raw] >> 8;
                op = (int)(ops8[raw] & 0xffu);
            } else {
                const uint32_t w4 = half ? (uint32_t) reinterpret_cast<const uint16_t *>(ops4)[raw] : ops4[raw];
                len = (int64_t)(w4 >> 3);
                op = (int)(w4 & 7u);
            }
            if (swp) op ^= (int)((0x6u >> op) & 1u) * 3;
            if (raw == pl.lo) len -= pl.sub_lo;
            if (raw == pl.lo + pl.n - 1) len -= pl.sub_hi;
        }
        int64_t v[3] = {len, op != OP_I ? len : 0, op != OP_D ? len : 0}, tot[3];
        pretty_scan3(v, tmp, tot);
        col0[t] = v[0];
        tb0[t] = v[1];
        qb0[t] = v[2];
        opk[t] = (uint8_t)op;
        if (t == 0) col0[PRETTY_NT] = tot[0];
        __syncthreads();
        for (int64_t c = t; c < tot[0]; c += PRETTY_NT) {
            uint32_t lo = 0, hi = PRETTY_NT; /* the last op whose first column is <= c (ops of no length share a start: take the last) */
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (col0[mid] <= c) lo = mid;
                else hi = mid;
            }
            const int64_t within = c - col0[lo];
            const int o = opk[lo];
            uint8_t mch = '-', nch = '-';
            if (o != OP_I) mch = T[t_base + tb0[lo] + within];
            if (o != OP_D) {
                const int64_t iq = q_base + qb0[lo] + within;
                nch = same ? Q[pl.qs + iq] : pretty_rc(Q[pl.qe - (iq + 1)]);
            }
            const int64_t col = col_base + c, win = col / PRETTY_WINDOW, k = col % PRETTY_WINDOW;
            const int64_t left = cols_total - win * PRETTY_WINDOW, w = left < PRETTY_WINDOW ? left : PRETTY_WINDOW;
            uint8_t *p = out + win * (3 * (PRETTY_WINDOW + 1)) + k;
            p[0] = mch;
            p[w + 1] = nch;
            p[2 * w + 2] = pretty_up(mch) == pretty_up(nch) ? '*' : ' ';
            if (k == w - 1) p[1] = p[w + 2] = p[2 * w + 3] = '\n';
        }
        col_base += tot[0];
        t_base += tot[1];
        q_base += tot[2];
        __syncthreads();
    }
}
