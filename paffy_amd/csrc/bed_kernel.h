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
/* pass 1 (starts == nullptr): run starts per workgroup tile; pass 2: their positions, at tile_off[tile] + rank inside the tile.
   A lane's sixteen counters are two 16-byte loads; when they lie strictly inside one sequence (all but a handful of lanes) a run starts
   wherever a counter differs from the one in front of it -- sixteen compares in registers, the counter in front of the first being
   the neighbour lane's last (round 3; the first version loaded and tested the counters one by one: 11.5 ms per pass over 14 GB). */
__global__ __launch_bounds__(PAFFY_NT) void k_bed_runs(BedParams B, const int64_t *tile_off, int64_t *tile_cnt, uint64_t *starts) {
    __shared__ int64_t scratch_mem[2 * PAFFY_NWAVE * 4];
    BlockComm bc{scratch_mem, 0};
    const uint64_t g0 = ((uint64_t)blockIdx.x * PAFFY_NT + threadIdx.x) * BED_PER;
    uint32_t flags = 0;
    const bool whole = g0 + BED_PER <= B.n_counts; /* n_counts is padded per sequence, the allocation beyond it is not read */
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (whole) {
        const uint4 a = *reinterpret_cast<const uint4 *>(B.counts + g0), b = *reinterpret_cast<const uint4 *>(B.counts + g0 + 8);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    }
    /* the counter in front of this lane's first: the last of the lane before (lane 0 of a wave and partial lanes read it) */
    uint32_t before = (uint32_t)__shfl_up((int)(w[7] >> 16), 1);
    const bool neighbour_whole = __shfl_up((int)whole, 1) != 0;
    if ((threadIdx.x & 63u) == 0 || !neighbour_whole) before = (g0 && g0 - 1 < B.n_counts) ? B.counts[g0 - 1] : 0u;
    if (g0 < B.n_counts) {
        uint32_t c = bed_contig_of(B, g0);
        const uint64_t rel0 = g0 - B.contig_base[c];
        if (whole && rel0 > 0 && rel0 + BED_PER <= (uint64_t)B.contig_len[c]) { /* strictly inside the sequence: no boundary among the sixteen */
            uint32_t prev = before;
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t lo = w[k] & 0xffffu, hi = w[k] >> 16;
                const bool s0 = B.binary ? (lo > 0) != (prev > 0) : lo != prev;
                const bool s1 = B.binary ? (hi > 0) != (lo > 0) : hi != lo;
                flags |= (s0 ? 1u : 0u) << (2 * k) | (s1 ? 1u : 0u) << (2 * k + 1);
                prev = hi;
            }
        } else {
            for (uint32_t j = 0; j < BED_PER && g0 + j < B.n_counts; j++) {
                const uint64_t g = g0 + j;
                while (c + 1 < B.n_contigs && B.contig_base[c + 1] <= g) c++;
                const uint32_t v = B.counts[g];
                if (bed_starts_run(B, g, v, before, c)) flags |= 1u << j;
                before = v;
            }
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
    if ((v >> 32) == 0) { /* every coordinate of a sequence below 4.29 Gb: ten compares instead of a division per digit */
        const uint32_t x = (uint32_t)v;
        return 1u + (x >= 10u) + (x >= 100u) + (x >= 1000u) + (x >= 10000u) + (x >= 100000u) + (x >= 1000000u) + (x >= 10000000u) + (x >= 100000000u) +
               (x >= 1000000000u);
    }
    uint32_t d = 1;
    while (v >= 10) {
        v /= 10;
        d++;
    }
    return d;
}
/* the d digits of v (d = bed_digits(v)) at p[0 .. d), most significant first; p: any byte-addressable memory */
__device__ __forceinline__ uint8_t *bed_put(uint8_t *p, uint64_t v, uint32_t d) {
    if ((v >> 32) == 0) {
        uint32_t x = (uint32_t)v;
        for (uint32_t i = 0; i < d; i++) {
            const uint32_t q = x / 10u;
            p[d - 1 - i] = (uint8_t)('0' + (x - q * 10u));
            x = q;
        }
        return p + d;
    }
    for (uint32_t i = 0; i < d; i++) {
        p[d - 1 - i] = (uint8_t)('0' + (uint32_t)(v % 10));
        v /= 10;
    }
    return p + d;
}
#define BED_STAGE 6144u /* bytes of a wave's 64 lines that are staged in LDS (96 per line on average); longer spans go out byte by byte */
/*
 * One lane per run: its line "name start end value\n" (impl/paf_to_bed.c:44-47) -- out == nullptr: the length only.
 * The runs are in counter order, so a wave's 64 lines are one contiguous piece of the output: the lanes put their bytes into the wave's
 * LDS stage at the piece's own alignment (ds_write_b8 runs at full rate at any address) and the wave writes the piece as whole 16-byte
 * chunks, the ragged first and last chunk byte by byte (round 3; the first version stored every byte of a line to HBM from its lane,
 * searched the sequence table per lane and counted digits with a 64-bit division per digit: 15.2 ms per pass over 390 M runs).
 */
__global__ __launch_bounds__(PAFFY_NT) void k_bed_lines(BedParams B, const uint64_t *starts, uint64_t n_runs, int64_t *len, const int64_t *off, uint8_t *out) {
    __shared__ __attribute__((aligned(16))) uint8_t stage_mem[PAFFY_NT / 64][BED_STAGE + 32];
    const uint64_t k = (uint64_t)blockIdx.x * PAFFY_NT + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const bool valid = k < n_runs;
    const uint64_t g = valid ? starts[k] : 0;
    /* the sequence: the same for the whole wave when its first and its last run agree (the runs are sorted) */
    const uint64_t wave_k0 = k - lane, wave_kl = wave_k0 + 63 < n_runs ? wave_k0 + 63 : n_runs - 1;
    uint32_t c = 0;
    if (wave_k0 < n_runs) {
        const uint32_t c_first = bed_contig_of(B, starts[wave_k0]), c_last = bed_contig_of(B, starts[wave_kl]); /* wave-uniform */
        c = c_first;
        if (c_first != c_last && valid) c = bed_contig_of(B, g);
    }
    int64_t bytes = 0;
    uint64_t i = 0, j = 0, shown = 0;
    uint32_t di = 0, dj = 0, ds = 0, nlen = 0, noff = 0;
    if (valid) {
        const uint64_t L = (uint64_t)(B.contig_len[c] > 0 ? B.contig_len[c] : 0);
        i = g - B.contig_base[c];
        if (i < L) { /* not the padding behind a sequence */
            j = k + 1 < n_runs ? starts[k + 1] - B.contig_base[c] : L;
            if (j > L) j = L;
            const uint32_t v = B.counts[g];
            const bool keep = (int64_t)(j - i) >= B.min_size && (v == 0 ? !B.exclude_unaligned : !B.exclude_aligned);
            if (keep) {
                shown = B.binary ? (v > 0 ? 1u : 0u) : v;
                nlen = B.name_len[c];
                noff = B.name_off[c];
                di = bed_digits(i); dj = bed_digits(j); ds = bed_digits(shown);
                bytes = (int64_t)nlen + 1 + di + 1 + dj + 1 + ds + 1;
            }
        }
    }
    if (!out) {
        if (valid) len[k] = bytes;
        return;
    }
    /* the wave's piece of the output: [o0, o0 + span) */
    const uint64_t my_off = valid ? (uint64_t)off[k] : 0;
    const uint64_t o0 = (uint64_t)__shfl((long long)my_off, 0);
    uint64_t end = valid ? my_off + (uint64_t)bytes : 0;
    for (int d = 32; d; d >>= 1) { /* the furthest end of the wave's lines (the lanes are in output order: the last valid lane's) */
        const uint64_t other = (uint64_t)__shfl_xor((long long)end, d);
        end = other > end ? other : end;
    }
    if (end <= o0) return; /* nothing kept in this wave */
    const uint64_t span = end - o0;
    const uint32_t shift = (uint32_t)(reinterpret_cast<uintptr_t>(out + o0) & 15u);
    const bool staged = span + shift <= BED_STAGE; /* wave-uniform */
    uint8_t *st = stage_mem[threadIdx.x >> 6];
    uint8_t *p = staged ? st + shift + (uint32_t)(my_off - o0) : out + my_off;
    if (bytes) {
        for (uint32_t t = 0; t < nlen; t++) p[t] = B.in[noff + t];
        p += nlen;
        *p++ = ' ';
        p = bed_put(p, i, di);
        *p++ = ' ';
        p = bed_put(p, j, dj);
        *p++ = ' ';
        p = bed_put(p, shown, ds);
        *p++ = '\n';
    }
    if (!staged) return;
    __builtin_amdgcn_wave_barrier(); /* a wave's LDS operations execute in order */
    /* stage bytes [shift, shift + span) -> out + o0; chunk q of the stage is the 16-byte aligned chunk at out + o0 - shift + 16 q */
    uint8_t *base = out + o0 - shift;
    const uint32_t total = shift + (uint32_t)span, n_chunks = (total + 15u) >> 4;
    for (uint32_t q = lane; q < n_chunks; q += 64) {
        const uint32_t lo = q * 16u, hi = lo + 16u;
        if (lo >= shift && hi <= total) {
            *reinterpret_cast<uint4 *>(base + lo) = *reinterpret_cast<const uint4 *>(st + lo);
        } else { /* the first or the last chunk: only the bytes of this wave's piece */
            for (uint32_t b = lo < shift ? shift : lo; b < (hi < total ? hi : total); b++) base[b] = st[b];
        }
    }
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
