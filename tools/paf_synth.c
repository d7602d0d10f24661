/*
 * paf_synth.c -- host build of the synthetic PAF workload (paffy_amd/csrc/paf_synth_core.h).
 * Workload tooling: used by the tests and by bench.py to make inputs; not part of the hot path.
 */
#include "../paffy_amd/csrc/paf_synth_core.h"

#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/*
 * Generates records [r0, r0+n) back to back into out. Returns the total byte count; with
 * out == NULL (or cap too small) nothing is written and the required size is returned.
 * rec_off (optional, n+1 entries) receives each record's byte offset.
 */
int64_t psynth_generate(const psynth_cfg *cfg, uint64_t r0, uint64_t n, char *out, int64_t cap, int64_t *rec_off,
                        int threads) {
    int64_t *off = rec_off ? rec_off : (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)n; i++) off[i + 1] = psynth_emit_record(cfg, r0 + (uint64_t)i, NULL);
    off[0] = 0;
    for (uint64_t i = 0; i < n; i++) off[i + 1] += off[i];
    int64_t total = off[n];
    if (out && total <= cap) {
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads)
        for (int64_t i = 0; i < (int64_t)n; i++) psynth_emit_record(cfg, r0 + (uint64_t)i, out + off[i]);
    }
    if (!rec_off) free(off);
    return total;
}

/* ---------------- cfg4: master alignments, genomes, records (host build) ---------------- */

typedef struct {
    psynth4_cfg cfg;
    psynth4_contig *contigs;
    int64_t *ckpt_q, *ckpt_t;
    psynth4_tab tab;
} psynth4_host;

void psynth4_destroy(psynth4_host *h) {
    if (!h) return;
    free(h->contigs);
    free(h->ckpt_q);
    free(h->ckpt_t);
    free(h);
}

/* Walks every master alignment once: contig lengths, op counts, checkpoints. NULL on bad arguments. */
psynth4_host *psynth4_create(const psynth4_cfg *cfg) {
    if (!cfg || cfg->tlen_min < 2048 || cfg->tlen_span < 0 || cfg->n_contigs < 1) return NULL;
    psynth4_host *h = (psynth4_host *)calloc(1, sizeof(*h));
    h->cfg = *cfg;
    h->contigs = (psynth4_contig *)calloc(cfg->n_contigs, sizeof(psynth4_contig));
    uint64_t total = 0;
    for (uint32_t c = 0; c < cfg->n_contigs; c++) {
        h->contigs[c].tlen = psynth4_tlen(cfg, c);
        h->contigs[c].ckpt_cap = psynth4_ckpt_cap(h->contigs[c].tlen);
        h->contigs[c].ckpt_base = total;
        total += h->contigs[c].ckpt_cap;
    }
    h->ckpt_q = (int64_t *)calloc(total, sizeof(int64_t));
    h->ckpt_t = (int64_t *)calloc(total, sizeof(int64_t));
    for (uint32_t c = 0; c < cfg->n_contigs; c++) {
        psynth4_contig *ct = &h->contigs[c];
        const uint64_t mkey = psynth4_mkey(cfg->seed, c);
        int64_t q = 0, t = 0;
        for (uint64_t j = 0; j < ct->ckpt_cap * PSYNTH4_G; j++) {
            if (j % PSYNTH4_G == 0) {
                h->ckpt_q[ct->ckpt_base + j / PSYNTH4_G] = q;
                h->ckpt_t[ct->ckpt_base + j / PSYNTH4_G] = t;
            }
            int op;
            int64_t len = psynth_op(mkey, j, &op);
            if (op != 2) q += len;
            if (op != 1) t += len;
            if (t > ct->tlen) break;
            if ((j & 1) == 0) {
                ct->n_ops = j + 1;
                ct->qlen = q;
            }
        }
    }
    h->tab.contigs = h->contigs;
    h->tab.ckpt_q = h->ckpt_q;
    h->tab.ckpt_t = h->ckpt_t;
    return h;
}

/* genome 0 = query (hs.chr<c+1>), 1 = target (pt.chr<c+1>) */
int64_t psynth4_contig_len(const psynth4_host *h, int genome, uint32_t c) { return genome ? h->contigs[c].tlen : h->contigs[c].qlen; }

void psynth4_genome(const psynth4_host *h, int genome, uint32_t c, char *out) {
    const psynth4_contig *ct = &h->contigs[c];
    const uint64_t seed = h->cfg.seed;
    if (genome) {
        for (int64_t p = 0; p < ct->tlen; p++) {
            int lower;
            uint32_t b = psynth4_tbase(seed, c, p, &lower);
            out[p] = psynth4_letter(b, lower, 0);
        }
        return;
    }
    const uint64_t mkey = psynth4_mkey(seed, c);
    const int minus = psynth4_minus(c);
    int64_t q = 0, t = 0;
    for (uint64_t j = 0; j < ct->n_ops; j++) {
        int op;
        int64_t len = psynth_op(mkey, j, &op);
        if (op != 2) {
            for (int64_t i = 0; i < len; i++) {
                int lower;
                uint32_t b = psynth4_qbase(seed, c, q + i, op == 0 ? t + i : -1, &lower);
                out[minus ? ct->qlen - 1 - (q + i) : q + i] = psynth4_letter(b, lower, minus);
            }
            q += len;
        }
        if (op != 1) t += len;
    }
}

int64_t psynth4_generate(const psynth4_host *h, uint64_t r0, uint64_t n, char *out, int64_t cap, int threads) {
    int64_t *off = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)n; i++) off[i + 1] = psynth4_emit_record(&h->cfg, &h->tab, r0 + (uint64_t)i, NULL);
    off[0] = 0;
    for (uint64_t i = 0; i < n; i++) off[i + 1] += off[i];
    int64_t total = off[n];
    if (out && total <= cap) {
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads)
        for (int64_t i = 0; i < (int64_t)n; i++) psynth4_emit_record(&h->cfg, &h->tab, r0 + (uint64_t)i, out + off[i]);
    }
    free(off);
    return total;
}
