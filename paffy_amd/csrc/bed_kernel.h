/*
 * bed_kernel.h -- the output side of `paffy to_bed` (impl/paf_to_bed.c:33-55): the coverage counters of all sequences (laid back to
 * back by coverage_host.h, in order of first appearance) as maximal runs "name start end value". The counters themselves come
 * from the slice walk of coverage_kernel.h.
 */
#ifndef PAFFY_BED_KERNEL_H_
#define PAFFY_BED_KERNEL_H_

#include "device_util.h"
#include "record_types.h"

struct BedParams {
    const uint16_t *counts;
    uint64_t n_counts;           /* all sequences back to back, each followed by a little padding */
    const uint64_t *contig_base; /* [n_contigs + 1] */
    const int64_t *contig_len;   /* [n_contigs] */
    const uint32_t *name_off, *name_len; /* [n_contigs] slices of the input text */
    uint32_t n_contigs;
    const uint8_t *in;
    int32_t binary, exclude_unaligned, exclude_aligned;
    int64_t min_size;
};

#define BED_PER 16u /* counters per lane per step */
/* which sequence a global counter position belongs to */
__device__ __forceinline__ uint32_t bed_contig_of(const BedParams &B, uint64_t g) {
    uint32_t lo = 0, hi = B.n_contigs - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (B.contig_base[mid] <= g) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}
/* a run starts at g: the first counter of a sequence, the first position behind one, or a value that differs from the one before */
__device__ __forceinline__ bool bed_starts_run(const BedParams &B, uint64_t g, uint32_t v, uint32_t before, uint32_t c) {
    const uint64_t rel = g - B.contig_base[c];
    if (rel == 0 || rel == (uint64_t)B.contig_len[c]) return true;
    if (rel > (uint64_t)B.contig_len[c]) return false; /* padding */
    return B.binary ? (v > 0) != (before > 0) : v != before;
}
/* pass 1 (starts == nullptr): run starts per workgroup tile; pass 2: their positions, at tile_off[tile] + rank inside the tile */
__global__ __launch_bounds__(PAFFY_NT) void k_bed_runs(BedParams B, const int64_t *tile_off, int64_t *tile_cnt, uint64_t *starts) {
    __shared__ int64_t scratch_mem[2 * PAFFY_NWAVE * 4];
    BlockComm bc{scratch_mem, 0};
    const uint64_t g0 = ((uint64_t)blockIdx.x * PAFFY_NT + threadIdx.x) * BED_PER;
    uint32_t flags = 0;
    if (g0 < B.n_counts) {
        uint32_t c = bed_contig_of(B, g0);
        uint32_t before = g0 ? B.counts[g0 - 1] : 0;
        for (uint32_t j = 0; j < BED_PER && g0 + j < B.n_counts; j++) {
            const uint64_t g = g0 + j;
            while (c + 1 < B.n_contigs && B.contig_base[c + 1] <= g) c++;
            const uint32_t v = B.counts[g];
            if (bed_starts_run(B, g, v, before, c)) flags |= 1u << j;
            before = v;
        }
    }
    int64_t n[1] = {(int64_t)__popc(flags)}, tot[1];
    block_excl_scan<1>(n, tot, bc);
    if (!starts) {
        if (threadIdx.x == 0) tile_cnt[blockIdx.x] = tot[0];
        return;
    }
    uint64_t o = (uint64_t)tile_off[blockIdx.x] + (uint64_t)n[0];
    while (flags) {
        const uint32_t j = (uint32_t)__ffs((int)flags) - 1u;
        flags &= flags - 1u;
        starts[o++] = g0 + j;
    }
}
__device__ __forceinline__ uint32_t bed_digits(uint64_t v) {
    uint32_t d = 1;
    while (v >= 10) {
        v /= 10;
        d++;
    }
    return d;
}
__device__ __forceinline__ uint8_t *bed_put(uint8_t *p, uint64_t v) {
    const uint32_t d = bed_digits(v);
    for (uint32_t i = 0; i < d; i++) {
        p[d - 1 - i] = (uint8_t)('0' + (uint32_t)(v % 10));
        v /= 10;
    }
    return p + d;
}
/* one lane per run: its line "name start end value\n" (impl/paf_to_bed.c:44-47) -- out == nullptr: the length only */
__global__ __launch_bounds__(PAFFY_NT) void k_bed_lines(BedParams B, const uint64_t *starts, uint64_t n_runs, int64_t *len, const int64_t *off, uint8_t *out) {
    const uint64_t k = (uint64_t)blockIdx.x * PAFFY_NT + threadIdx.x;
    if (k >= n_runs) return;
    const uint64_t g = starts[k];
    const uint32_t c = bed_contig_of(B, g);
    const uint64_t i = g - B.contig_base[c], L = (uint64_t)(B.contig_len[c] > 0 ? B.contig_len[c] : 0);
    int64_t bytes = 0;
    if (i < L) { /* not the padding behind a sequence */
        uint64_t j = k + 1 < n_runs ? starts[k + 1] - B.contig_base[c] : L;
        if (j > L) j = L;
        const uint32_t v = B.counts[g];
        const bool keep = (int64_t)(j - i) >= B.min_size && (v == 0 ? !B.exclude_unaligned : !B.exclude_aligned);
        if (keep) {
            const uint64_t shown = B.binary ? (v > 0 ? 1u : 0u) : v;
            bytes = (int64_t)B.name_len[c] + 1 + bed_digits(i) + 1 + bed_digits(j) + 1 + bed_digits(shown) + 1;
            if (out) {
                uint8_t *p = out + off[k];
                for (uint32_t t = 0; t < B.name_len[c]; t++) p[t] = B.in[B.name_off[c] + t];
                p += B.name_len[c];
                *p++ = ' ';
                p = bed_put(p, i);
                *p++ = ' ';
                p = bed_put(p, j);
                *p++ = ' ';
                p = bed_put(p, shown);
                *p++ = '\n';
            }
        }
    }
    if (!out) len[k] = bytes;
}
/* exclusive scan of n int64 values in two levels: sums per 4096-value tile, a one-workgroup scan of those, then the tiles */
__global__ __launch_bounds__(PAFFY_NT) void k_scan64_tiles(const int64_t *in, uint64_t n, int64_t *tile_sum) {
    __shared__ int64_t scratch_mem[2 * PAFFY_NWAVE * 4];
    BlockComm bc{scratch_mem, 0};
    const uint64_t base = (uint64_t)blockIdx.x * (PAFFY_NT * 16);
    int64_t v[1] = {0};
    for (uint32_t j = 0; j < 16; j++) {
        const uint64_t i = base + (uint64_t)threadIdx.x * 16 + j;
        if (i < n) v[0] += in[i];
    }
    block_sum<1>(v, bc);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = v[0];
}
__global__ __launch_bounds__(PAFFY_NT) void k_scan64_fix(const int64_t *in, uint64_t n, const int64_t *tile_off, int64_t *out) {
    __shared__ int64_t scratch_mem[2 * PAFFY_NWAVE * 4];
    BlockComm bc{scratch_mem, 0};
    const uint64_t base = (uint64_t)blockIdx.x * (PAFFY_NT * 16);
    int64_t mine[16], v[1] = {0}, tot[1];
    for (uint32_t j = 0; j < 16; j++) {
        const uint64_t i = base + (uint64_t)threadIdx.x * 16 + j;
        mine[j] = i < n ? in[i] : 0;
        v[0] += mine[j];
    }
    block_excl_scan<1>(v, tot, bc);
    int64_t run = tile_off[blockIdx.x] + v[0];
    for (uint32_t j = 0; j < 16; j++) {
        const uint64_t i = base + (uint64_t)threadIdx.x * 16 + j;
        if (i < n) out[i] = run;
        run += mine[j];
    }
}

#endif
