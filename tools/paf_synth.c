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
