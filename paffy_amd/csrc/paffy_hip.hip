/*
 * paffy_hip.hip -- gfx950 implementation of include/paffy_hip.h.
 *
 * Kernels, in launch order for one batch of PAF text resident in HBM:
 *   k_sep_count / k_sep_write  one coalesced pass each over the bytes (16 B per lane): '\t' and
 *                              '\n' positions are compacted into sep_pos[], newline ranks into
 *                              nl_idx[] (replaces stFile_getLine + strtok_r, impl/paf.c:144-212)
 *   k_header                   one lane per record: fixed fields and tags (paf_parse, impl/paf.c:137-209)
 *   k_size_lds                 one workgroup per record: cigar -> LDS ops (mirrored to HBM) -> transforms -> exact size + plan
 *   k_arena_size               same for records whose ops do not fit LDS (ops in an HBM arena)
 *   k_scan_part / k_scan_fix   exclusive prefix sum of the sizes up to the first failing record (two levels)
 *   k_emit_lds<rows|line> / k_arena_emit   the lines: each wave formats its share of the record through its own
 *                              LDS ring and flushes 16-byte coalesced stores
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/paffy_hip.h"
#include "paf_synth_core.h"
#include "record_groups.h"
#include "flat_kernel.h"
#include "flat_add_kernel.h"
#include "coverage_kernel.h"
#include "bed_kernel.h"

#define SEP_TILE 65536u /* bytes per workgroup in the separator passes */
#define WAVE_OPS_CAP 2048u   /* op store of the one-wave sizing kernel: 8 KiB, sixteen workgroups per CU */
#define LVL0_LONG_BYTES 20000u /* cigars up to this length start at the first store level once a batch has shown that none of them overflows it */
#define WAVE_MAX_BYTES 6000u /* cigars up to this length go there (about 1950 ops at 3.08 bytes per op; the few denser ones are redone by the four-wave build) */

/* ------------------------------------------------------------------ */
/* separators                                                           */
/* ------------------------------------------------------------------ */

template <bool ND = false>
__device__ __forceinline__ void sep_masks(const uint8_t *in, uint32_t in_len, uint32_t g, uint32_t &tabs_nl, uint32_t &nl, uint32_t *nondigits = nullptr) {
    /* 16 bytes at g (multiple of 16): bit j of tabs_nl set for '\t' or '\n', of nl for '\n'; ND: *nondigits = bytes that are not ASCII digits
       (the flat sizing pass, flat_kernel.h, places a chunk's ops inside its record with the counts per 1 KiB tile) */
    tabs_nl = nl = 0;
    if (ND) *nondigits = 0;
    if (g >= in_len) return;
    uint4 v = *reinterpret_cast<const uint4 *>(in + g);
    if (ND) *nondigits = (uint32_t)__popc(nondigit16(v));
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++) { /* four bytes at a time: a byte equals the pattern where the xor has a zero byte */
        const uint32_t t = equal4(w[k], 0x09090909u), n = equal4(w[k], 0x0a0a0a0au);
        tabs_nl |= (t | n) << (4 * k);
        nl |= n << (4 * k);
    }
    if (in_len - g < 16u) { /* the text ends inside these 16 bytes */
        const uint32_t keep = (1u << (in_len - g)) - 1u;
        tabs_nl &= keep;
        nl &= keep;
    }
}

/*
 * The same masks and the non-digit count with fewer instructions (the one-pass index spends most of its time on them: 2 700 VALU
 * instructions per wave, two thirds of them here). Per 4-byte word: bytes 9 and 10 are the bytes b < 0x80 with ((b | 0x80) - 9) & 0xfe
 * == 0x80 -- one subtraction without borrows between the bytes, one zero-byte test -- and the newlines among them have bit 0 of the
 * difference set; the four flag bits of a word (bit 7 of each byte) are gathered by one multiplication; non-digit bytes are counted on
 * the flags themselves; the newline mask is only put together in the (few) rounds in which a lane of the wave has one.
 */
__device__ __forceinline__ void sep_masks_nd(const uint4 &v, uint32_t in_len, uint32_t g, uint32_t &tabs_nl, uint32_t &nl, uint32_t &nondigits) {
    /* v: the 16 bytes at g (a multiple of 16), loaded by the caller -- all the loads of a workgroup's tile are issued before the first is looked at */
    tabs_nl = nl = nondigits = 0;
    if (g >= in_len) return;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t nlf[4], any_nl = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        nondigits += (uint32_t)__popc(nondigit4(w[k]));
        const uint32_t u = (w[k] | 0x80808080u) - 0x09090909u;                       /* per byte: 0x80 for a tab, 0x81 for a newline; no borrows */
        const uint32_t z = ((u ^ 0x80808080u) & 0xfefefefeu) | (w[k] & 0x80808080u); /* zero byte <=> tab or newline */
        const uint32_t f = ~(((z & 0x7f7f7f7fu) + 0x7f7f7f7fu) | z) & 0x80808080u;   /* bit 7 of every zero byte */
        nlf[k] = f & (u << 7);
        any_nl |= nlf[k];
        tabs_nl |= ((f * 0x00204081u) >> 28) << (4 * k); /* bits 7, 15, 23, 31 -> 28 .. 31: the other partial products stay below bit 24 */
    }
    if (any_nl) {
#pragma unroll
        for (int k = 0; k < 4; k++) nl |= ((nlf[k] * 0x00204081u) >> 28) << (4 * k);
    }
    if (in_len - g < 16u) { /* the text ends inside these 16 bytes */
        const uint32_t keep = (1u << (in_len - g)) - 1u;
        tabs_nl &= keep;
        nl &= keep; /* (the non-digit count takes the 16 bytes as loaded, like sep_masks<true>: no cigar reaches beyond the text) */
    }
}

__global__ __launch_bounds__(PAFFY_NT) void k_sep_count(const uint8_t *in, uint32_t in_len, uint2 *tile_counts, uint16_t *nd) {
    __shared__ int64_t scratch_mem[2 * PAFFY_NWAVE * 4];
    BlockComm scratch{scratch_mem, 0};
    const uint32_t tile0 = blockIdx.x * SEP_TILE;
    int64_t acc[2] = {0, 0};
    for (uint32_t off = threadIdx.x * 16; off < SEP_TILE; off += PAFFY_NT * 16) {
        uint32_t a, b, c;
        uint4 txt = make_uint4(0, 0, 0, 0);
        if (tile0 + off < in_len) txt = *reinterpret_cast<const uint4 *>(in + tile0 + off);
        sep_masks_nd(txt, in_len, tile0 + off, a, b, c);
        acc[0] += __popc(a);
        acc[1] += __popc(b);
        c = wave_sum_u32(c); /* a wave's 64 x 16 bytes are one 1 KiB tile */
        if ((threadIdx.x & 63u) == 0 && nd && tile0 + off < in_len) nd[(tile0 + off) >> FLAT_TILE_SHIFT] = (uint16_t)c;
    }
    block_sum<2>(acc, scratch);
    if (threadIdx.x == 0) {
        /* a final line without '\n' still is a record (impl/paf.c:213): virtual newline at in_len */
        if (blockIdx.x == gridDim.x - 1 && in_len > 0 && in[in_len - 1] != '\n') {
            acc[0] += 1;
            acc[1] += 1;
        }
        tile_counts[blockIdx.x] = make_uint2((uint32_t)acc[0], (uint32_t)acc[1]);
    }
}

/* single workgroup: exclusive scan of the per-tile counts */
__global__ __launch_bounds__(PAFFY_NT) void k_scan_tiles(uint2 *tile_counts, uint32_t n_tiles, DevInfo *info) {
    __shared__ int64_t scratch_mem[2 * PAFFY_NWAVE * 4];
    BlockComm scratch{scratch_mem, 0};
    int64_t carry0 = 0, carry1 = 0;
    for (uint32_t base = 0; base < n_tiles; base += PAFFY_NT) {
        uint32_t i = base + threadIdx.x;
        uint2 c = i < n_tiles ? tile_counts[i] : make_uint2(0, 0);
        int64_t v[2] = {c.x, c.y}, tot[2];
        block_excl_scan<2>(v, tot, scratch);
        if (i < n_tiles) tile_counts[i] = make_uint2((uint32_t)(carry0 + v[0]), (uint32_t)(carry1 + v[1]));
        carry0 += tot[0];
        carry1 += tot[1];
    }
    if (threadIdx.x == 0) {
        info->n_seps = (uint32_t)carry0;
        info->n_lines = (uint32_t)carry1;
    }
}

__global__ __launch_bounds__(PAFFY_NT) void k_sep_write(const uint8_t *in, uint32_t in_len, const uint2 *tile_off, uint32_t *sep_pos,
                                                         uint32_t *nl_idx) {
    __shared__ int64_t scratch_mem[2 * PAFFY_NWAVE * 4];
    BlockComm scratch{scratch_mem, 0};
    /* tiles in reverse launch order: k_sep_count read the end of the text last, so it is the part most likely still in
       the Infinity Cache */
    const uint32_t tile = gridDim.x - 1 - blockIdx.x;
    const uint32_t tile0 = tile * SEP_TILE;
    uint2 base = tile_off[tile];
    uint32_t sbase = base.x, lbase = base.y;
    for (uint32_t off = threadIdx.x * 16; off < SEP_TILE; off += PAFFY_NT * 16) {
        uint32_t a, b;
        const uint32_t g = tile0 + off;
        sep_masks(in, in_len, g, a, b);
        uint32_t v[2] = {(uint32_t)__popc(a), (uint32_t)__popc(b)}, tot[2];
        block_excl_scan_u32<2>(v, tot, scratch);
        uint32_t si = sbase + v[0], li = lbase + v[1];
        while (a) {
            int j = __ffs((int)a) - 1;
            a &= a - 1;
            sep_pos[si] = g + j;
            if (b & (1u << j)) nl_idx[li++] = si;
            si++;
        }
        sbase += tot[0];
        lbase += tot[1];
    }
    if (threadIdx.x == 0 && tile == gridDim.x - 1 && in_len > 0 && in[in_len - 1] != '\n') {
        sep_pos[sbase] = in_len;
        nl_idx[lbase] = sbase;
    }
}

/*
 * The separator index in ONE pass over the text (round 3; k_sep_count / k_scan_tiles / k_sep_write above read it twice and are kept for
 * the first batch of a context, whose separator density is not known yet). A workgroup takes the next 64 KiB tile from a ticket
 * counter, keeps the tab / newline masks of its sixteen rounds in registers (two 16-bit masks per round and thread), turns the counts
 * into positions inside the tile with ONE workgroup scan (the per-round counts are laid out in LDS in text order, every thread sums
 * sixteen consecutive entries; the two-pass version ran a workgroup scan per round), learns the number of separators and lines in
 * front of its tile by decoupled look-back over the tiles' published counts (a tile publishes its own counts at once and the running
 * totals as soon as it knows them; tiles are handed out in dispatch order, so a tile only ever waits for tiles that are running), and
 * writes the positions. Capacities are the caller's guess: nothing is written past them, the totals tell the host whether to repeat.
 */
#define SEP_AGG (1ull << 62)
#define SEP_INC (2ull << 62)
__device__ __forceinline__ unsigned long long sep_pack(uint32_t seps, uint32_t lines) { return (unsigned long long)seps | ((unsigned long long)lines << 31); }
__global__ __launch_bounds__(PAFFY_NT) void k_sep_index(const uint8_t *in, uint32_t in_len, uint32_t n_tiles, unsigned long long *state, uint32_t *sep_pos,
                                                         uint32_t cap_seps, uint32_t *nl_idx, uint32_t cap_lines, DevInfo *info, uint16_t *nd) {
    __shared__ int64_t scratch_mem[2 * PAFFY_NWAVE * 4];
    /* per round and thread, in text order (entry e = round * 256 + thread): separators | newlines << 16 -- first the counts, then the
       exclusive prefix inside the tile. One spare word per sixteen entries: a thread's sixteen consecutive entries are then 17 words from
       its neighbour's, not 16 (sixteen lanes on four banks) */
    __shared__ uint32_t pref[16 * PAFFY_NT + PAFFY_NT];
#define SEP_PREF(e) pref[(e) + ((e) >> 4)]
    __shared__ uint32_t s_tile, s_base[2];
    BlockComm scratch{scratch_mem, 0};
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(reinterpret_cast<unsigned int *>(state), 1u);
    __syncthreads();
    const uint32_t tile = s_tile, tile0 = tile * SEP_TILE;
    unsigned long long *status = state + 1;
    uint32_t m[16];
    uint4 txt[16]; /* the tile's sixteen loads in flight together: one at a time, each waited for, was most of the kernel's time */
#pragma unroll
    for (uint32_t it = 0; it < 16; it++) {
        const uint32_t g = tile0 + it * (PAFFY_NT * 16u) + tid * 16u;
        txt[it] = make_uint4(0, 0, 0, 0);
        if (g < in_len) txt[it] = *reinterpret_cast<const uint4 *>(in + g);
    }
#pragma unroll
    for (uint32_t it = 0; it < 16; it++) {
        uint32_t a, b, c;
        sep_masks_nd(txt[it], in_len, tile0 + it * (PAFFY_NT * 16u) + tid * 16u, a, b, c);
        c = wave_sum_u32(c); /* a wave's 64 x 16 bytes are one 1 KiB tile */
        if (lane == 0 && nd && tile0 + it * (PAFFY_NT * 16u) + tid * 16u < in_len) nd[(tile0 + it * (PAFFY_NT * 16u) + tid * 16u) >> FLAT_TILE_SHIFT] = (uint16_t)c;
        m[it] = a | (b << 16);
        SEP_PREF(it * PAFFY_NT + tid) = (uint32_t)__popc(a) | ((uint32_t)__popc(b) << 16);
    }
    __syncthreads();
    /* entries in text order: e = round * 256 + thread; this thread sums entries [16 tid, 16 tid + 16) (at most 256 per field: no carry) */
    uint32_t loc[16], run = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
        loc[k] = run;
        run += pref[17u * tid + k]; /* = SEP_PREF(16 tid + k) */
    }
    uint32_t v[2] = {run & 0xffffu, run >> 16}, tot[2];
    block_excl_scan_u32<2>(v, tot, scratch);
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) pref[17u * tid + k] = (v[0] + (loc[k] & 0xffffu)) | ((v[1] + (loc[k] >> 16)) << 16); /* below 65 536 each */
    /* a final line without '\n' still is a record (impl/paf.c:213): virtual newline at in_len */
    const bool virt = tile == n_tiles - 1 && in_len > 0 && in[in_len - 1] != '\n';
    const uint32_t t_seps = tot[0] + (virt ? 1u : 0u), t_lines = tot[1] + (virt ? 1u : 0u);
    if (wave == 0) {
        uint32_t e_seps = 0, e_lines = 0;
        if (tile > 0) {
            if (lane == 0) __hip_atomic_store(&status[tile], SEP_AGG | sep_pack(t_seps, t_lines), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t look = (int64_t)tile - 1;
            for (;;) {
                const int64_t idx = look - (int64_t)lane;
                unsigned long long sv = SEP_INC; /* in front of tile 0: nothing */
                if (idx >= 0) sv = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__any((sv >> 62) == 0)) continue; /* a tile in the window has not published yet: it is running, ask again */
                const unsigned long long inc = __ballot((sv >> 62) == 2);
                const uint32_t first = inc ? (uint32_t)__ffsll((long long)inc) - 1u : 63u; /* nearest tile that knows its running totals */
                const bool use = lane <= first;
                uint32_t a = use ? (uint32_t)(sv & 0x7fffffffull) : 0u, b = use ? (uint32_t)((sv >> 31) & 0x7fffffffull) : 0u;
                a = wave_last_u32(wave_incl_scan_u32(a));
                b = wave_last_u32(wave_incl_scan_u32(b));
                e_seps += a;
                e_lines += b;
                if (inc) break;
                look -= 64;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&status[tile], SEP_INC | sep_pack(e_seps + t_seps, e_lines + t_lines), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_base[0] = e_seps;
            s_base[1] = e_lines;
            if (tile == n_tiles - 1) {
                info->n_seps = e_seps + t_seps;
                info->n_lines = e_lines + t_lines;
            }
        }
    }
    __syncthreads();
    const uint32_t e_seps = s_base[0], e_lines = s_base[1];
#pragma unroll
    for (uint32_t it = 0; it < 16; it++) {
        uint32_t a = m[it] & 0xffffu;
        const uint32_t b = m[it] >> 16, g = tile0 + it * (PAFFY_NT * 16u) + tid * 16u, pp = SEP_PREF(it * PAFFY_NT + tid);
        uint32_t si = e_seps + (pp & 0xffffu), li = e_lines + (pp >> 16);
        while (a) {
            const int j = __ffs((int)a) - 1;
            a &= a - 1;
            if (si < cap_seps) sep_pos[si] = g + j;
            if (b & (1u << j)) {
                if (li < cap_lines) nl_idx[li] = si;
                li++;
            }
            si++;
        }
    }
    if (virt && tid == 0) {
        const uint32_t si = e_seps + tot[0], li = e_lines + tot[1];
        if (si < cap_seps) sep_pos[si] = in_len;
        if (li < cap_lines) nl_idx[li] = si;
    }
#undef SEP_PREF
}

/* ------------------------------------------------------------------ */
/* header fields                                                        */
/* ------------------------------------------------------------------ */

/* str_to_int64, impl/paf.c:37-48 */
__device__ int64_t parse_i64_dev(const uint8_t *in, uint32_t p, uint32_t e) {
    uint64_t v = 0;
    bool neg = false;
    if (p < e && in[p] == '-') {
        neg = true;
        p++;
    }
    while (p < e) {
        uint32_t d = (uint32_t)in[p] - '0';
        if (d > 9u) break;
        v = v * 10 + d;
        p++;
    }
    return (int64_t)(neg ? (uint64_t)0 - v : v);
}

/* the same for a token of at most 20 bytes with its text in registers: three aligned 8-byte words that hold bytes of the token are loaded at
   once (a word is only read when it starts in front of the token's end: nothing beyond the text is touched), where parse_i64_dev waits
   for one byte load per digit -- a line's twelve columns were most of k_header's time */
__device__ __forceinline__ int64_t parse_i64_words(const uint8_t *in, uint32_t p, uint32_t e) {
    if (e - p > 20u) return parse_i64_dev(in, p, e);
    const uintptr_t a = reinterpret_cast<uintptr_t>(in + p);
    const uint64_t *w = reinterpret_cast<const uint64_t *>(a & ~(uintptr_t)7);
    const uint32_t sh = (uint32_t)(a & 7u) * 8u, n = e - p;
    const uint32_t span = (uint32_t)(a & 7u) + n; /* bytes from the first word's start to the token's end: <= 27 */
    const uint64_t w0 = w[0], w1 = span > 8u ? w[1] : 0ull, w2 = span > 16u ? w[2] : 0ull, w3 = span > 24u ? w[3] : 0ull;
    uint64_t t[3];
    if (sh) {
        t[0] = (w0 >> sh) | (w1 << (64u - sh));
        t[1] = (w1 >> sh) | (w2 << (64u - sh));
        t[2] = (w2 >> sh) | (w3 << (64u - sh));
    } else {
        t[0] = w0; t[1] = w1; t[2] = w2;
    }
    uint64_t v = 0;
    uint32_t i = 0;
    const bool neg = (uint32_t)(t[0] & 0xffu) == (uint32_t)'-'; /* n >= 1: the token is not empty */
    if (neg) i = 1;
    for (; i < n; i++) {
        const uint64_t word = i < 8u ? t[0] : (i < 16u ? t[1] : t[2]);
        const uint32_t d = ((uint32_t)(word >> (8u * (i & 7u))) & 0xffu) - (uint32_t)'0';
        if (d > 9u) break;
        v = v * 10 + d;
    }
    return (int64_t)(neg ? (uint64_t)0 - v : v);
}

/*
 * paf_parse, impl/paf.c:137-209: tokens are the non-empty gaps between consecutive separators of
 * the line (strtok_r collapses runs of tabs). Tag tokens shorter than 5 bytes are never
 * recognised (documented deviation: the reference reads past the token there).
 *
 * 32 lanes per record, one separator (= one token) per lane and round: a token's field number is the count of non-empty tokens in
 * front of it (ballot + popcount), every lane parses its own token and drops the result into the record's RecMeta image in LDS, the
 * image leaves as 36 coalesced words. The reference stops at the first token it aborts on (strand, tp): tokens behind the first
 * failing one are ignored; of several tags of one kind the last one wins. (The one-lane-per-record version of rounds 1-2 needed 256
 * registers -- one wave per SIMD -- and 0.16 ms per 131 072 records.)
 */
#define HDR_GROUP 32u
__global__ __launch_bounds__(PAFFY_NT) void k_header(const uint8_t *in, const uint32_t *sep_pos, const uint32_t *nl_idx, uint32_t n_lines,
                                                      RecMeta *meta, uint32_t *big_list, DevInfo *info, uint32_t lvl0_max, uint32_t *chunk_rec) {
    constexpr uint32_t kWords = sizeof(RecMeta) / 4; /* 36 */
    static_assert(sizeof(RecMeta) % 4 == 0 && kWords > 32 && kWords <= 64, "the image is written as two words per lane at most");
    __shared__ __attribute__((aligned(16))) RecMeta image[PAFFY_NT / HDR_GROUP];
    const uint32_t g = threadIdx.x / HDR_GROUP, gl = threadIdx.x % HDR_GROUP;
    const uint32_t r = blockIdx.x * (PAFFY_NT / HDR_GROUP) + g;
    const bool live = r < n_lines;
    const uint32_t shift = (threadIdx.x & 32u); /* this group's half of the wave's 64-bit ballots */
    RecMeta *m = &image[g];
    uint32_t *mw = reinterpret_cast<uint32_t *>(m);
    mw[gl] = 0;
    if (gl + 32 < kWords) mw[32 + gl] = 0;
    __builtin_amdgcn_wave_barrier();
    if (gl == 0) {
        m->tile_level = -1;
        m->chain_id = -1;
        m->chain_score = -1;
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t s_end = 0, s_begin = 1;
    if (live) {
        s_end = nl_idx[r];
        s_begin = r == 0 ? 0 : nl_idx[r - 1] + 1;
    }
    uint32_t n_fields = 0; /* non-empty tokens taken so far (group-uniform) */
    int32_t err = 0;       /* group-uniform */
    for (uint32_t c = s_begin; c <= s_end && err == 0; c += HDR_GROUP) { /* s_end - s_begin < 2^31: no wrap */
        const uint32_t si = c + gl;
        const bool valid = live && si <= s_end && si >= c;
        uint32_t te = 0, t = 0;
        if (valid) {
            te = sep_pos[si];
            t = si == 0 ? 0u : sep_pos[si - 1] + 1u; /* the separator in front of a line's first token is the newline of the line before */
        }
        bool on = valid && te > t;
        const uint32_t ne = (uint32_t)(__ballot(on) >> shift);
        const uint32_t field = n_fields + (uint32_t)__popc(ne & ((1u << gl) - 1u));
        /* what this token is */
        int kind = -1; /* 0..5: tp AS cg tl cn s1 */
        int32_t my_err = 0, my_aux = 0;
        int64_t val = 0;
        uint32_t v_off = 0;
        uint8_t c0 = 0;
        if (on) {
            if (field < 12) {
                if (field == 4) {
                    c0 = in[t];
                    if (c0 != '+' && c0 != '-') {
                        my_err = PAFFY_ERR_STRAND;
                        my_aux = c0;
                    }
                } else if (field != 0 && field != 5) {
                    val = parse_i64_words(in, t, te);
                }
            } else if (te - t >= 5 && in[t + 2] == ':' && in[t + 4] == ':') {
                const uint8_t t0 = in[t], t1 = in[t + 1];
                v_off = t + 5;
                if (t0 == 't' && t1 == 'p') {
                    kind = 0;
                    c0 = v_off < te ? in[v_off] : 0;
                    if (c0 != 'P' && c0 != 'S' && c0 != 'I') {
                        my_err = PAFFY_ERR_TP_ASSERT;
                        my_aux = c0;
                    }
                } else if (t0 == 'A' && t1 == 'S') {
                    kind = 1;
                } else if (t0 == 'c' && t1 == 'g') {
                    kind = 2;
                } else if (t0 == 't' && t1 == 'l') {
                    kind = 3;
                } else if (t0 == 'c' && t1 == 'n') {
                    kind = 4;
                } else if (t0 == 's' && t1 == '1') {
                    kind = 5;
                }
                if ((kind == 1 || kind >= 3) && v_off < te) val = parse_i64_words(in, v_off, te);
            }
        }
        /* the reference never sees a token behind the first one it aborts on */
        const uint32_t em = (uint32_t)(__ballot(my_err != 0) >> shift);
        if (em) {
            const uint32_t first = (uint32_t)__ffs((int)em) - 1u;
            on = on && gl <= first;
            err = __shfl(my_err, (int)(first + shift), 64);
            const int32_t aux = __shfl(my_aux, (int)(first + shift), 64);
            if (gl == first) {
                m->err = err;
                m->err_aux = aux;
            }
        }
        const uint32_t taken = (uint32_t)(__ballot(on) >> shift);
        n_fields += (uint32_t)__popc(taken);
        if (on && field < 12) {
            switch (field) {
                case 0: m->qname_off = t; m->qname_len = te - t; break;
                case 4: m->same_strand = c0 == '+'; break;
                case 5: m->tname_off = t; m->tname_len = te - t; break;
                default: (&m->qlen)[field <= 3 ? field - 1 : field - 3] = val; break; /* qlen qs qe | tlen ts te nmatch nbases mapq */
            }
        }
        /* of several tags of one kind the last one wins: the highest lane writes */
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const uint32_t km = (uint32_t)(__ballot(on && kind == k) >> shift);
            if (km && gl == 31u - (uint32_t)__clz((int)km)) {
                if (k == 0) m->type = c0;
                else if (k == 1) m->score = val;
                else if (k == 2) { m->has_cg = 1; m->cg_off = v_off; m->cg_len = te - v_off; } /* impl/paf.c:193-198 */
                else if (k == 3) m->tile_level = val;
                else if (k == 4) m->chain_id = val;
                else m->chain_score = val;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (live && err == 0 && n_fields < 12) {
        err = PAFFY_ERR_FEW_FIELDS;
        if (gl == 0) m->err = err;
    }
    __builtin_amdgcn_wave_barrier();
    if (live) {
        uint32_t *dst = reinterpret_cast<uint32_t *>(meta + r);
        dst[gl] = mw[gl];
        if (gl + 32 < kWords) dst[32 + gl] = mw[32 + gl];
        /* long cigars go straight to the sizing launch with the bigger LDS store (runs beside the main one) */
        if (gl == 0 && err == 0 && (m->cg_len >> 1) > lvl0_max) big_list[atomicAdd(&info->b_count[0], 1u)] = r;
    }
    if (chunk_rec && live && err == 0 && m->has_cg && m->cg_len > 0) {
        /* the flat sizing pass (flat_kernel.h): the record's cigar is pieces (its text cut at the 1 KiB boundaries of the batch) and chunks of
           four pieces; chunk c of record r stands at slot (cg_off >> 12) + r + c of the chunk list (records are in text order: no two
           chunks share a slot, no scan or atomic hands the slots out) */
        const uint32_t tf = m->cg_off >> FLAT_TILE_SHIFT;
        const uint32_t np = ((m->cg_off + m->cg_len - 1u) >> FLAT_TILE_SHIFT) - tf + 1u;
        const uint32_t nc = (np + FLAT_CHUNK_PIECES - 1u) / FLAT_CHUNK_PIECES;
        for (uint32_t j = gl; j < nc; j += HDR_GROUP) chunk_rec[(tf >> 2) + r + j] = r;
    }
}

/* ------------------------------------------------------------------ */
/* sequence lookup (add_mismatches)                                     */
/* ------------------------------------------------------------------ */

/* names sorted bytewise (memcmp, shorter first on a tie): binary search per record name; replaces the
 * stHash_search by name of impl/paf_add_mismatches.c:116,123 */
__device__ int32_t find_seq(const uint8_t *in, uint32_t off, uint32_t len, const uint8_t *names, const uint32_t *name_off, int32_t n) {
    int32_t lo = 0, hi = n - 1;
    while (lo <= hi) {
        int32_t mid = (lo + hi) >> 1;
        const uint8_t *nm = names + name_off[mid];
        uint32_t nl = name_off[mid + 1] - name_off[mid];
        uint32_t k = 0, m = len < nl ? len : nl;
        int cmp = 0;
        for (; k < m; k++) {
            int d = (int)in[off + k] - (int)nm[k];
            if (d) {
                cmp = d;
                break;
            }
        }
        if (cmp == 0) cmp = len < nl ? -1 : (len > nl ? 1 : 0);
        if (cmp == 0) return mid;
        if (cmp < 0) hi = mid - 1;
        else lo = mid + 1;
    }
    return -1;
}
__global__ __launch_bounds__(PAFFY_NT) void k_seq_lookup(const uint8_t *in, const RecMeta *meta, uint32_t n_rec, const uint8_t *names,
                                                          const uint32_t *name_off, int32_t n_seqs, int32_t *rec_qseq, int32_t *rec_tseq) {
    uint32_t r = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (r >= n_rec) return;
    const RecMeta &m = meta[r];
    rec_qseq[r] = m.err ? -1 : find_seq(in, m.qname_off, m.qname_len, names, name_off, n_seqs);
    rec_tseq[r] = m.err ? -1 : find_seq(in, m.tname_off, m.tname_len, names, name_off, n_seqs);
}

/* ------------------------------------------------------------------ */
/* record sizes -> offsets                                              */
/* ------------------------------------------------------------------ */

/*
 * Exclusive prefix sum of the record sizes up to the first failing record (records at or after it contribute
 * nothing), two levels: k_scan_part sums and scans blocks of SCAN_BLOCK records, k_scan_fix adds the block bases
 * (at most a few hundred of them, re-added by every workgroup) and leaves the totals in DevInfo.
 */
#define SCAN_PER 8
#define SCAN_BLOCK (PAFFY_NT * SCAN_PER)
__global__ __launch_bounds__(PAFFY_NT) void k_scan_part(const int64_t *out_len, const int64_t *out_rows, uint32_t n, int64_t *out_off, int64_t *part,
                                                         const DevInfo *info) {
    __shared__ int64_t scratch_mem[2 * PAFFY_NWAVE * 4];
    BlockComm scratch{scratch_mem, 0};
    const uint32_t first_err = (uint32_t)(info->first_err_key >> 16);
    const uint32_t live_n = n < first_err ? n : first_err;
    const uint32_t i0 = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER;
    int64_t len[SCAN_PER];
    int64_t v[2] = {0, 0}, tot[2];
#pragma unroll
    for (int j = 0; j < SCAN_PER; j++) {
        const uint32_t i = i0 + j;
        len[j] = i < live_n ? out_len[i] : 0;
        v[0] += len[j];
        v[1] += i < live_n ? out_rows[i] : 0;
    }
    block_excl_scan<2>(v, tot, scratch);
    int64_t run = v[0];
#pragma unroll
    for (int j = 0; j < SCAN_PER; j++) {
        const uint32_t i = i0 + j;
        if (i < n) out_off[i] = run; /* block-local; k_scan_fix adds the base */
        run += len[j];
    }
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = tot[0];
        part[2 * blockIdx.x + 1] = tot[1];
    }
}
__global__ __launch_bounds__(PAFFY_NT) void k_scan_fix(uint32_t n, uint32_t n_blocks, int64_t *out_off, const int64_t *part, DevInfo *info) {
    __shared__ int64_t scratch_mem[2 * PAFFY_NWAVE * 4];
    BlockComm scratch{scratch_mem, 0};
    /* base of this block = sum of the totals of the blocks before it; the last workgroup also sees the grand totals */
    int64_t v[2] = {0, 0};
    for (uint32_t b = threadIdx.x; b < n_blocks; b += PAFFY_NT) {
        if (b < blockIdx.x) v[0] += part[2 * b];
    }
    int64_t t[2] = {0, 0};
    if (blockIdx.x == n_blocks - 1)
        for (uint32_t b = threadIdx.x; b < n_blocks; b += PAFFY_NT) {
            t[0] += part[2 * b];
            t[1] += part[2 * b + 1];
        }
    int64_t s[3] = {v[0], t[0], t[1]};
    block_sum<3>(s, scratch);
    const uint32_t i0 = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER;
#pragma unroll
    for (int j = 0; j < SCAN_PER; j++)
        if (i0 + j < n) out_off[i0 + j] += s[0];
    if (blockIdx.x == n_blocks - 1 && threadIdx.x == 0) {
        info->out_bytes = (unsigned long long)s[1];
        info->out_rows = (unsigned long long)s[2];
    }
}

/*
 * Launch order of the record kernels: records in 64 classes of size (the sizing pass: 512 bytes of cigar text per class,
 * the writers: 16 KiB of output), largest class first, so that the long records do not start last and stretch the kernel's tail (longest-processing-time-first, coarse).
 * k_order_count histograms the classes, k_order_scatter gives every record its slot.
 */
#define ORDER_CLASSES 64u
__device__ __forceinline__ uint32_t order_class(int64_t out_len, uint32_t shift) {
    const uint64_t c = (uint64_t)(out_len < 0 ? 0 : out_len) >> shift;
    return c < ORDER_CLASSES ? (uint32_t)c : ORDER_CLASSES - 1;
}
__global__ __launch_bounds__(PAFFY_NT) void k_cigar_bytes(const RecMeta *meta, uint32_t n, int64_t *out) {
    const uint32_t r = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (r < n) out[r] = meta[r].has_cg ? (int64_t)meta[r].cg_len : 0;
}
__global__ __launch_bounds__(PAFFY_NT) void k_order_count(const int64_t *out_len, uint32_t n, uint32_t shift, uint32_t *counts) {
    __shared__ uint32_t h[ORDER_CLASSES];
    if (threadIdx.x < ORDER_CLASSES) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t r = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (r < n) atomicAdd(&h[order_class(out_len[r], shift)], 1u);
    __syncthreads();
    if (threadIdx.x < ORDER_CLASSES && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], h[threadIdx.x]);
}
__global__ __launch_bounds__(PAFFY_NT) void k_order_scatter(const int64_t *out_len, uint32_t n, uint32_t shift, const uint32_t *counts, uint32_t *cursor,
                                                             uint32_t *order) {
    __shared__ uint32_t base[ORDER_CLASSES], mine[ORDER_CLASSES], got[ORDER_CLASSES];
    if (threadIdx.x < ORDER_CLASSES) mine[threadIdx.x] = 0;
    if (threadIdx.x == 0) { /* classes in descending order */
        uint32_t run = 0;
        for (int c = (int)ORDER_CLASSES - 1; c >= 0; c--) {
            base[c] = run;
            run += counts[c];
        }
    }
    __syncthreads();
    const uint32_t r = blockIdx.x * PAFFY_NT + threadIdx.x;
    uint32_t c = 0, local = 0;
    if (r < n) {
        c = order_class(out_len[r], shift);
        local = atomicAdd(&mine[c], 1u); /* slot inside this workgroup's share of the class */
    }
    __syncthreads();
    if (threadIdx.x < ORDER_CLASSES && mine[threadIdx.x]) got[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], mine[threadIdx.x]); /* one global atomic per class */
    __syncthreads();
    if (r < n) order[base[c] + got[c] + local] = r;
}

/* ------------------------------------------------------------------ */
/* tile: keys for the host ordering, sizes and the verbatim writer       */
/* ------------------------------------------------------------------ */

/* dedupe: 128-bit key of (query name, target name, strand, four coordinates), the same for the swapped record, the
 * coordinate part of paf_check (impl/paf.c:427-438; the cigar is not parsed here) and the record's own tile level */
struct DedupeKey {
    uint64_t a, b;   /* key */
    uint64_t ia, ib; /* key of the record with query and target swapped */
    int32_t err;     /* parse error (PAFFY_ERR_*) */
    int32_t check;   /* paf_check code on the coordinates, 0 = fine */
};
__device__ __forceinline__ void dedupe_mix(uint64_t &h1, uint64_t &h2, uint64_t x) {
    h1 = (h1 ^ x) * 0x100000001b3ull;
    h1 ^= h1 >> 29;
    h2 = (h2 + x) * 0x9e3779b97f4a7c15ull;
    h2 ^= h2 >> 32;
}
__device__ __forceinline__ void dedupe_name(const uint8_t *in, uint32_t off, uint32_t len, uint64_t &h1, uint64_t &h2) {
    h1 = 0xcbf29ce484222325ull;
    h2 = 0x6a09e667f3bcc909ull;
    for (uint32_t i = 0; i < len; i++) dedupe_mix(h1, h2, in[off + i]);
    dedupe_mix(h1, h2, 0x100u + len);
}
__global__ __launch_bounds__(PAFFY_NT) void k_dedupe_keys(const uint8_t *in, const RecMeta *meta, uint32_t n, DedupeKey *keys, int64_t *level) {
    uint32_t r = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (r >= n) return;
    const RecMeta &m = meta[r];
    DedupeKey k;
    k.err = m.err;
    uint64_t q1, q2, t1, t2;
    dedupe_name(in, m.qname_off, m.qname_len, q1, q2);
    dedupe_name(in, m.tname_off, m.tname_len, t1, t2);
    uint64_t a = q1, b = q2, ia = t1, ib = t2;
    dedupe_mix(a, b, t1); dedupe_mix(a, b, t2);
    dedupe_mix(ia, ib, q1); dedupe_mix(ia, ib, q2);
    const uint64_t strand = m.same_strand ? 1 : 2;
    const uint64_t f[4] = {(uint64_t)m.qs, (uint64_t)m.qe, (uint64_t)m.ts, (uint64_t)m.te};
    dedupe_mix(a, b, strand); dedupe_mix(ia, ib, strand);
    for (int i = 0; i < 4; i++) {
        dedupe_mix(a, b, f[i]);
        dedupe_mix(ia, ib, f[i ^ 2]); /* the swapped record's query coordinates are this one's target coordinates */
    }
    k.a = a; k.b = b; k.ia = ia; k.ib = ib;
    int chk = 0;
    if (m.qs < 0 || m.qs >= m.qlen) chk = PAFFY_ERR_CHECK_QSTART;
    else if (m.qs > m.qe || m.qe > m.qlen) chk = PAFFY_ERR_CHECK_QEND;
    else if (m.ts < 0 || m.ts >= m.tlen) chk = PAFFY_ERR_CHECK_TSTART;
    else if (m.ts > m.te || m.te > m.tlen) chk = PAFFY_ERR_CHECK_TEND;
    k.check = chk;
    keys[r] = k;
    level[r] = m.tile_level;
}

/* PAFFY_STATS: the batch's six sums from the records' (rec_stats[6 n], zero for records that stopped in front of the stage) */
__global__ __launch_bounds__(PAFFY_NT) void k_stats_reduce(const int64_t *rec_stats, uint32_t n, unsigned long long *sums) {
    __shared__ unsigned long long part[PAFFY_NT / 64][6];
    unsigned long long a[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x; i < n; i += gridDim.x * PAFFY_NT) {
#pragma unroll
        for (int k = 0; k < 6; k++) a[k] += (unsigned long long)rec_stats[6ull * i + k];
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
        for (int d = 32; d; d >>= 1) a[k] += __shfl_down(a[k], d, 64);
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int k = 0; k < 6; k++) part[wave][k] = a[k];
    __syncthreads();
    if (threadIdx.x < 6) {
        unsigned long long t = 0;
        for (uint32_t w = 0; w < PAFFY_NT / 64; w++) t += part[w][threadIdx.x];
        if (t) atomicAdd(&sums[threadIdx.x], t);
    }
}

__device__ __forceinline__ void tile_state(const RecMeta &m, int64_t level, RecState &s) {
    load_state(m, s);
    s.has_cigar = false; /* the cigar is written verbatim from the text, impl/paf.c:381-385 */
    s.tile_level = level;
}

/* ------------------------------------------------------------------ */
/* synthetic workload, one lane per record                   */
/* ------------------------------------------------------------------ */

__global__ __launch_bounds__(PAFFY_NT) void k_synth_size(psynth_cfg cfg, uint64_t r0, uint32_t n, int64_t *sizes) {
    uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i < n) sizes[i] = psynth_emit_record(&cfg, r0 + i, nullptr);
}
__global__ __launch_bounds__(PAFFY_NT) void k_synth_fill(psynth_cfg cfg, uint64_t r0, uint32_t n, const int64_t *off, uint8_t *out) {
    uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i < n) psynth_emit_record(&cfg, r0 + i, reinterpret_cast<char *>(out) + off[i]);
}
__global__ __launch_bounds__(PAFFY_NT) void k_scan_i64(const int64_t *in, uint32_t n, int64_t *out_excl, int64_t *total) {
    __shared__ int64_t scratch_mem[2 * PAFFY_NWAVE * 4];
    BlockComm scratch{scratch_mem, 0};
    int64_t carry = 0;
    for (uint32_t base = 0; base < n; base += PAFFY_NT) {
        uint32_t i = base + threadIdx.x;
        int64_t v[1] = {i < n ? in[i] : 0}, tot[1];
        block_excl_scan<1>(v, tot, scratch);
        if (i < n) out_excl[i] = carry + v[0];
        carry += tot[0];
    }
    if (threadIdx.x == 0) *total = carry;
}

/* Sequences are kept in the form they are compared in (impl/paf.c:752-757: toupper of both bases, the query base complemented
 * on the - strand): seq upper-cased in place, comp = its complement. 16 bytes per lane. */
__global__ __launch_bounds__(PAFFY_NT) void k_seq_canon(uint8_t *seq, uint8_t *comp, uint64_t n16) {
    const uint64_t i = (uint64_t)blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i >= n16) return;
    uint4 w = reinterpret_cast<uint4 *>(seq)[i];
    w.x = upper4(w.x); w.y = upper4(w.y); w.z = upper4(w.z); w.w = upper4(w.w);
    reinterpret_cast<uint4 *>(seq)[i] = w;
    w.x = comp4(w.x); w.y = comp4(w.y); w.z = comp4(w.z); w.w = comp4(w.w);
    reinterpret_cast<uint4 *>(comp)[i] = w;
}

/* ---- cfg4 workload: master alignments, genomes, records (paf_synth_core.h; host twin in tools/paf_synth.c) ---- */

/* one wave per contig pair walks the master ops 64 at a time: op count, query length, checkpoints */
__global__ __launch_bounds__(64) void k_synth4_master(psynth4_cfg cfg, psynth4_contig *contigs, int64_t *ckpt_q, int64_t *ckpt_t) {
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    psynth4_contig ct = contigs[c];
    const uint64_t mkey = psynth4_mkey(cfg.seed, c);
    int64_t q = 0, t = 0, best_q = 0;
    uint64_t best_n = 0;
    for (uint64_t base = 0; base < ct.ckpt_cap * PSYNTH4_G; base += 64) {
        if (base % PSYNTH4_G == 0 && lane == 0) {
            ckpt_q[ct.ckpt_base + base / PSYNTH4_G] = q;
            ckpt_t[ct.ckpt_base + base / PSYNTH4_G] = t;
        }
        int op;
        const int64_t len = psynth_op(mkey, base + lane, &op);
        const int64_t qi = q + wave_incl_scan(op != 2 ? len : 0), ti = t + wave_incl_scan(op != 1 ? len : 0);
        const uint64_t ok = __ballot(ti <= ct.tlen); /* a prefix of the lanes: the offsets only grow */
        const uint64_t even_ok = ok & 0x5555555555555555ull;
        if (even_ok) {
            const int top = 63 - __clzll((long long)even_ok);
            best_n = base + (uint64_t)top + 1;
            best_q = __shfl(qi, top);
        }
        if (ok != ~0ull) break;
        q = wave_last(qi);
        t = wave_last(ti);
    }
    if (lane == 0) {
        contigs[c].n_ops = best_n;
        contigs[c].qlen = best_q;
    }
}

/* target genome: one thread per 32 bases */
__global__ __launch_bounds__(PAFFY_NT) void k_synth4_target(psynth4_cfg cfg, uint32_t c, int64_t tlen, uint8_t *out) {
    const int64_t p0 = ((int64_t)blockIdx.x * PAFFY_NT + threadIdx.x) * 32;
    for (int64_t p = p0; p < p0 + 32 && p < tlen; p++) {
        int lower;
        const uint32_t b = psynth4_tbase(cfg.seed, c, p, &lower);
        out[p] = (uint8_t)psynth4_letter(b, lower, 0);
    }
}

/* query genome: one wave per checkpoint block of master ops, one lane per op */
__global__ __launch_bounds__(64) void k_synth4_query(psynth4_cfg cfg, uint32_t c, psynth4_tab tab, uint8_t *out) {
    const psynth4_contig ct = tab.contigs[c];
    const uint64_t mkey = psynth4_mkey(cfg.seed, c);
    const int minus = psynth4_minus(c);
    int64_t q = tab.ckpt_q[ct.ckpt_base + blockIdx.x], t = tab.ckpt_t[ct.ckpt_base + blockIdx.x];
    for (uint64_t base = (uint64_t)blockIdx.x * PSYNTH4_G; base < ((uint64_t)blockIdx.x + 1) * PSYNTH4_G && base < ct.n_ops; base += 64) {
        const uint64_t j = base + threadIdx.x;
        int op;
        int64_t len = psynth_op(mkey, j, &op);
        const int64_t dq = op != 2 ? len : 0, dt = op != 1 ? len : 0;
        const int64_t qi = wave_incl_scan(dq), ti = wave_incl_scan(dt);
        if (j < ct.n_ops && op != 2) {
            const int64_t q0 = q + qi - dq, t0 = t + ti - dt;
            for (int64_t i = 0; i < len; i++) {
                int lower;
                const uint32_t b = psynth4_qbase(cfg.seed, c, q0 + i, op == 0 ? t0 + i : -1, &lower);
                out[minus ? ct.qlen - 1 - (q0 + i) : q0 + i] = (uint8_t)psynth4_letter(b, lower, minus);
            }
        }
        q += wave_last(qi);
        t += wave_last(ti);
    }
}

__global__ __launch_bounds__(PAFFY_NT) void k_synth4_size(psynth4_cfg cfg, psynth4_tab tab, uint64_t r0, uint32_t n, int64_t *sizes) {
    uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i < n) sizes[i] = psynth4_emit_record(&cfg, &tab, r0 + i, nullptr);
}
__global__ __launch_bounds__(PAFFY_NT) void k_synth4_fill(psynth4_cfg cfg, psynth4_tab tab, uint64_t r0, uint32_t n, const int64_t *off, uint8_t *out) {
    uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i < n) psynth4_emit_record(&cfg, &tab, r0 + i, reinterpret_cast<char *>(out) + off[i]);
}

/* ------------------------------------------------------------------ */
/* host side                                                            */
/* ------------------------------------------------------------------ */

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct ProfEntry {
    std::string name;
    double ms = 0;
    int64_t launches = 0;
};

struct paffy_hip_ctx {
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr; /* sizing launches of the long-cigar records run here, beside the main launch */
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::string last_error;
    double sep_guess_per_mib = 0, line_guess_per_mib = 0; /* separators / lines per MiB of the last batch indexed: sizes the one-pass index of the next */
    DevBuf tile_counts, sep_pos, nl_idx, meta, out_len, out_rows, status, err_aux, n_ops, arena_off, out_off, w_list, b_list, b_list1, arena, info, synth_sizes, rec_plan, ops_mirror, seq_blob, seq_table, seq_names, seq_name_off, rec_qseq, rec_tseq, seq_comp;
    int32_t n_seqs = 0;
    DevBuf synth4_contigs, synth4_q, synth4_t; /* cfg4 workload tables (paffy_hip_synth4_setup) */
    psynth4_cfg synth4_cfg = {0, 0, 0, 0, 0};
    paffy_filter filter = {-1, -1, -1.0, -1.0, -1, 0};
    DevBuf dedupe_keys;
    struct DedupeState *dedupe = nullptr; /* dedupe_host.h: keys of the records written so far (sorted, on the device) and scratch */
    DevBuf scan_part, emit_order, order_cnt, tile_keys, tile_order, tile_rank, tile_coff, tile_cbase, tile_cov, tile_level, tile_len, tile_items, tile_slots, tile_parts;
    bool plan_is_tile = false;
    bool plan_is_bed = false; /* paffy to_bed: emit writes the run lines */
    bool keep_raw = false;    /* paffy_hip_keep_raw_sequences: seq_raw holds the bases as loaded (paf_pretty_print shows their case) */
    bool plan_seq_lookup = false; /* rec_qseq / rec_tseq belong to the current plan */
    DevBuf seq_raw, pretty_off, pretty_out, pretty_err, host_in, host_out;
    DevBuf rec_stats; /* six sums per record of the PAFFY_STATS stage */
    DevBuf flat_nd, flat_rec, flat_chunks, flat_sums, flat_done, flat_items, flat_pieces; /* the flat sizing pass (flat_kernel.h) */
    DevBuf add_pieces, add_scr_cnt, add_scr_off, add_new_cnt, add_new_off, add_text, add_bad, add_part, add_scratch, add_new_ops; /* flat_add_kernel.h */
    uint32_t flat_piece_slots = 0;
    DevBuf bed_keys, bed_tab, bed_starts, bed_len, bed_off, bed_tiles;
    struct BedParams *bed_params = nullptr; /* host copy */
    uint64_t bed_runs = 0;
    uint32_t tile_n = 0;
    const uint8_t *tile_in = nullptr;
    struct ChainState *chain = nullptr; /* `paffy chain` (chain_host.h) */
    struct CovState *cov = nullptr; /* `paffy tile` / `paffy to_bed` over any number of batches (coverage_host.h) */
    /* the batch paffy_hip_query_names indexed last: paffy_hip_split_by_owner on the same batch reuses the index (one use) */
    const void *indexed_in = nullptr;
    int64_t indexed_len = 0;
    uint32_t indexed_n = 0;
    /* the index of batches whose names were asked for (paffy_hip_query_names), kept until the batch is split: a sharded tile asks
       for the names of all its batches before it splits the first one */
    struct KeptIndex {
        const void *in;
        int64_t len;
        uint32_t n, n_seps;
        DevBuf meta, sep_pos, nl_idx;
    };
    std::vector<KeptIndex> kept_index;
    std::vector<KeptIndex> index_pool; /* buffers of dropped entries, reused by the next kept index (no hipMalloc / hipFree per batch) */
    DevBuf one_batch;               /* table with the single text pointer of a one-batch plan (dedupe / split_file lines) */
    /* what emit writes for a line plan (tile, dedupe): record order, levels, offsets */
    const uint8_t *const *line_batches = nullptr;
    const RecMeta *line_meta = nullptr;
    const uint32_t *line_order = nullptr;
    const int64_t *line_level = nullptr;
    const uint64_t *line_off = nullptr;
    uint64_t line_n = 0;
    DevInfo *h_info = nullptr; /* pinned */
    /* plan state */
    bool planned = false;
    KParams kp;
    paffy_plan_info plan;
    /* profiling */
    bool profile = false;
    int64_t flat_left = -1, flat_reasons[16] = {0}; /* paffy_hip_flat_stats */
    uint32_t flat_chunk_slots = 0;
    bool lvl0_long_ok = false, lvl0_long_off = false; /* the longer first store level: shown safe by the batch before / overflowed once */
    /* device buffers of the two slots of a closed stream (paffy_hip_stream_close), taken again by the next paffy_hip_stream_open: a
       hipMalloc of the tens of GB a slot's output needs takes 16 ms most of the time and 0.5-2.6 s right behind the hipFree of the stream
       before (tools/probes/d2h_pieces.py) */
    void *kept_slot_buf[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; /* [slot][0 = input, 1 = output] */
    size_t kept_slot_cap[2][2] = {{0, 0}, {0, 0}};
    std::vector<struct paffy_hip_stream *> open_streams; /* streams of this context that are open: paffy_hip_destroy detaches them (a stream closed later frees
                                                           its buffers itself instead of leaving them with a context that is gone) */
    std::string profile_only; /* when not empty: only launches of this kernel are bracketed (paffy_hip_profile_only) */
    std::vector<hipEvent_t> event_pool;
    std::vector<ProfEntry> prof;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending;
    std::vector<std::string> name_store;
};

#define HIPCHK(ctx, call)                                                                   \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e_);          \
            return PAFFY_E_HIP;                                                             \
        }                                                                                   \
    } while (0)

static int ensure(paffy_hip_ctx *c, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) HIPCHK(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 2 + 256; /* room to grow: a batch a little larger than the last must not cost a free and an allocation */
    if (hipMalloc(&b.p, want) != hipSuccess) { /* no room for the slack: an input that fits HBM at its exact size must not fail here */
        (void)hipGetLastError();
        b.p = nullptr;
        want = bytes;
        HIPCHK(c, hipMalloc(&b.p, want));
    }
    b.cap = want;
    return 0;
}

static int prof_slot(paffy_hip_ctx *c, const char *name) {
    for (size_t i = 0; i < c->prof.size(); i++)
        if (c->prof[i].name == name) return (int)i;
    ProfEntry e;
    e.name = name;
    c->prof.push_back(e);
    return (int)c->prof.size() - 1;
}
static void prof_collect(paffy_hip_ctx *c) {
    for (auto &p : c->pending) {
        float ms = 0;
        (void)hipEventSynchronize(p.second.second);
        (void)hipEventElapsedTime(&ms, p.second.first, p.second.second);
        c->prof[p.first].ms += ms;
        c->prof[p.first].launches += 1;
        c->event_pool.push_back(p.second.first);
        c->event_pool.push_back(p.second.second);
    }
    c->pending.clear();
}
static hipEvent_t prof_event(paffy_hip_ctx *c) {
    hipEvent_t e = nullptr;
    if (!c->event_pool.empty()) {
        e = c->event_pool.back();
        c->event_pool.pop_back();
    } else {
        (void)hipEventCreate(&e);
    }
    return e;
}
static bool prof_wants(const paffy_hip_ctx *c, const char *name) { return c->profile && (c->profile_only.empty() || c->profile_only == name); }

/* launch wrapper: optional HIP events on the context's stream around each kernel */
#define LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                                  \
    do {                                                                                    \
        hipEvent_t e0_ = nullptr, e1_ = nullptr;                                            \
        const bool prof_ = prof_wants(ctx, name);                                           \
        if (prof_) {                                                                        \
            e0_ = prof_event(ctx);                                                          \
            e1_ = prof_event(ctx);                                                          \
            (void)hipEventRecord(e0_, (ctx)->stream);                                             \
        }                                                                                   \
        hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__);         \
        if (prof_) {                                                                        \
            (void)hipEventRecord(e1_, (ctx)->stream);                                             \
            (ctx)->pending.push_back({prof_slot(ctx, name), {e0_, e1_}});                   \
        }                                                                                   \
        HIPCHK(ctx, hipGetLastError());                                                     \
    } while (0)

extern "C" {

/* Experiments only: extra dynamic LDS per workgroup of the row writer (0), the one-wave sizing launch (1) and the four-wave sizing launch (2)
   of the lean pipes, from PAFFY_DBG_LDS_PAD="emit,size64,size256" (bytes) -- lowers a kernel's occupancy so that two contexts' kernels can be
   resident on a CU at the same time (bench.py --pipeline 2). */
static size_t dbg_lds_pad(int which) {
    static long pad[3] = {-1, 0, 0};
    if (pad[0] < 0) {
        pad[0] = 0;
        const char *e = getenv("PAFFY_DBG_LDS_PAD");
        if (e) sscanf(e, "%ld,%ld,%ld", &pad[0], &pad[1], &pad[2]);
        for (int k = 0; k < 3; k++) pad[k] = pad[k] < 0 ? 0 : pad[k] & ~15l;
    }
    return (size_t)pad[which];
}
int paffy_hip_create(paffy_hip_ctx **out, int device) {
    if (!out) return PAFFY_E_ARG;
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return PAFFY_E_HIP;
    paffy_hip_ctx *c = new paffy_hip_ctx();
    if (hipHostMalloc(reinterpret_cast<void **>(&c->h_info), sizeof(DevInfo)) != hipSuccess) {
        delete c;
        return PAFFY_E_HIP;
    }
    if (ensure(c, c->info, sizeof(DevInfo))) {
        delete c;
        return PAFFY_E_HIP;
    }
    (void)hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    (void)hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    /* the record kernels use more than the default 64 KiB of LDS */
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cov_walk<true>), hipFuncAttributeMaxDynamicSharedMemorySize, sizeof(CovWalkLds));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cov_walk<false>), hipFuncAttributeMaxDynamicSharedMemorySize, sizeof(CovWalkLds));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_size_lds<PAFFY_MASK_ALL>), hipFuncAttributeMaxDynamicSharedMemorySize, PAFFY_SIZE_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_size_lds<PAFFY_MASK_LEAN>), hipFuncAttributeMaxDynamicSharedMemorySize, PAFFY_SIZE_LDS_BYTES + dbg_lds_pad(2));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_size_lds<PAFFY_MASK_ADD>), hipFuncAttributeMaxDynamicSharedMemorySize, PAFFY_SIZE_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_size_lds<PAFFY_MASK_PLAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, PAFFY_SIZE_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_size_lds<PAFFY_MASK_SEL>), hipFuncAttributeMaxDynamicSharedMemorySize, PAFFY_SIZE_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_size_lds_long<PAFFY_MASK_PLAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, PAFFY_SIZE_LDS_BYTES_FOR(PAFFY_OPS_CAP_BIG));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_size_lds_long<PAFFY_MASK_SEL>), hipFuncAttributeMaxDynamicSharedMemorySize, PAFFY_SIZE_LDS_BYTES_FOR(PAFFY_OPS_CAP_BIG));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_size_lds_long<PAFFY_MASK_ADD>), hipFuncAttributeMaxDynamicSharedMemorySize, PAFFY_SIZE_LDS_BYTES_FOR(PAFFY_OPS_CAP_BIG));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_size_lds_long<PAFFY_MASK_ALL>), hipFuncAttributeMaxDynamicSharedMemorySize, PAFFY_SIZE_LDS_BYTES_FOR(PAFFY_OPS_CAP_BIG));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_size_lds_long<PAFFY_MASK_LEAN>), hipFuncAttributeMaxDynamicSharedMemorySize, PAFFY_SIZE_LDS_BYTES_FOR(PAFFY_OPS_CAP_BIG));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_emit_lds<true>), hipFuncAttributeMaxDynamicSharedMemorySize, PAFFY_EMIT_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_emit_lds<false>), hipFuncAttributeMaxDynamicSharedMemorySize, PAFFY_EMIT_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_arena_size), hipFuncAttributeMaxDynamicSharedMemorySize, PAFFY_SIZE_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_arena_emit<true>), hipFuncAttributeMaxDynamicSharedMemorySize, PAFFY_EMIT_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_arena_emit<false>), hipFuncAttributeMaxDynamicSharedMemorySize, PAFFY_EMIT_LDS_BYTES);
    *out = c;
    return 0;
}

static void stream_detach(struct paffy_hip_stream *s);
static void cov_free(paffy_hip_ctx *c); /* coverage_host.h state */
static void dedupe_free(paffy_hip_ctx *c); /* dedupe_host.h state */
static void index_drop(paffy_hip_ctx *c, const void *d_in);
static void index_drop_all(paffy_hip_ctx *c);
static void chain_free(paffy_hip_ctx *c);

void paffy_hip_destroy(paffy_hip_ctx *c) {
    if (!c) return;
    for (paffy_hip_stream *st : c->open_streams) stream_detach(st); /* streams still open: they close later without this context */
    c->open_streams.clear();
    prof_collect(c);
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    c->event_pool.clear();
    cov_free(c);
    chain_free(c);
    while (!c->kept_index.empty()) index_drop(c, c->kept_index.back().in);
    for (paffy_hip_ctx::KeptIndex &k : c->index_pool) {
        if (k.meta.p) (void)hipFree(k.meta.p);
        if (k.sep_pos.p) (void)hipFree(k.sep_pos.p);
        if (k.nl_idx.p) (void)hipFree(k.nl_idx.p);
    }
    c->index_pool.clear();
    if (c->one_batch.p) (void)hipFree(c->one_batch.p);
    DevBuf *bufs[] = {&c->tile_counts, &c->sep_pos, &c->nl_idx, &c->meta, &c->out_len, &c->out_rows, &c->status, &c->err_aux,
                      &c->n_ops, &c->arena_off, &c->out_off, &c->w_list, &c->b_list, &c->b_list1, &c->arena, &c->info, &c->synth_sizes, &c->rec_plan, &c->ops_mirror, &c->seq_blob, &c->seq_table, &c->seq_names, &c->seq_name_off, &c->synth4_contigs, &c->synth4_q, &c->synth4_t, &c->seq_comp, &c->seq_raw, &c->pretty_off, &c->pretty_out, &c->pretty_err, &c->host_in, &c->host_out, &c->rec_stats, &c->flat_nd, &c->flat_rec, &c->flat_chunks, &c->flat_sums, &c->flat_done, &c->flat_items, &c->flat_pieces, &c->add_pieces, &c->add_scr_cnt, &c->add_scr_off, &c->add_new_cnt, &c->add_new_off, &c->add_text, &c->add_bad, &c->add_part, &c->add_scratch, &c->add_new_ops, &c->bed_keys, &c->bed_tab, &c->bed_starts, &c->bed_len, &c->bed_off, &c->bed_tiles,
                      &c->rec_qseq, &c->rec_tseq, &c->tile_keys, &c->tile_order, &c->tile_rank, &c->tile_coff, &c->tile_cbase,
                      &c->tile_cov, &c->tile_level, &c->tile_len, &c->tile_items, &c->tile_slots, &c->tile_parts, &c->scan_part, &c->dedupe_keys, &c->emit_order, &c->order_cnt};
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    dedupe_free(c);
    (void)paffy_hip_stream_trim(c);
    delete c->bed_params;
    if (c->h_info) (void)hipHostFree(c->h_info);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    delete c;
}

int paffy_hip_set_stream(paffy_hip_ctx *c, void *s) {
    if (!c) return PAFFY_E_ARG;
    c->stream = reinterpret_cast<hipStream_t>(s);
    return 0;
}

const char *paffy_hip_last_error(paffy_hip_ctx *c) { return c ? c->last_error.c_str() : "null context"; }

static int fetch_info(paffy_hip_ctx *c) {
    HIPCHK(c, hipMemcpyAsync(c->h_info, c->info.p, sizeof(DevInfo), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

/* Separator index + header parse shared by plan and tile_plan. */
static void index_drop(paffy_hip_ctx *c, const void *d_in);
static int index_and_parse(paffy_hip_ctx *c, const uint8_t *in, uint32_t len, uint32_t *n_lines_out, uint32_t lvl0_max = PAFFY_OPS_CAP, bool flat = false) {
    index_drop(c, in); /* a kept index of this buffer describes what it held before */
    const uint32_t n_tiles = (len + SEP_TILE - 1) / SEP_TILE;
    c->indexed_in = nullptr; /* the index buffers are about to describe another batch */

    DevInfo zero;
    memset(&zero, 0, sizeof(zero));
    zero.first_err_key = ~0ull;
    HIPCHK(c, hipMemcpyAsync(c->info.p, &zero, sizeof(zero), hipMemcpyHostToDevice, c->stream));
    /* One pass (k_sep_index) when the separator density of this context's batches is known from the batch before: the index buffers are
       sized by that guess, the kernel writes nothing past them and reports the true counts; a batch that needs more is indexed again
       with exact sizes. The first batch of a context (no guess) takes the two-pass form, which sizes the buffers exactly. */
    uint32_t n_seps = 0, n_lines = 0;
    bool indexed = false;
    /* the flat sizing pass wants the non-digit bytes of every 1 KiB tile of the text: the index kernels count them on their way */
    uint16_t *nd = nullptr;
    if (flat) {
        if (ensure(c, c->flat_nd, sizeof(uint16_t) * (((size_t)len >> FLAT_TILE_SHIFT) + 2))) return PAFFY_E_HIP;
        nd = static_cast<uint16_t *>(c->flat_nd.p);
    }
    if (c->sep_guess_per_mib > 0 && !getenv("PAFFY_TWO_PASS_INDEX")) {
        const double mib = (double)len / (1 << 20) + 1.0;
        const size_t want_seps = (size_t)(c->sep_guess_per_mib * mib * 1.25) + 4096, want_lines = (size_t)(c->line_guess_per_mib * mib * 1.25) + 4096;
        if (ensure(c, c->sep_pos, sizeof(uint32_t) * (want_seps + 1)) || ensure(c, c->nl_idx, sizeof(uint32_t) * (want_lines + 1)) ||
            ensure(c, c->tile_counts, sizeof(unsigned long long) * ((size_t)n_tiles + 1)))
            return PAFFY_E_HIP;
        const uint32_t cap_seps = (uint32_t)std::min<size_t>(c->sep_pos.cap / sizeof(uint32_t), 0x7fffffffu), cap_lines = (uint32_t)std::min<size_t>(c->nl_idx.cap / sizeof(uint32_t), 0x7fffffffu);
        HIPCHK(c, hipMemsetAsync(c->tile_counts.p, 0, sizeof(unsigned long long) * ((size_t)n_tiles + 1), c->stream));
        LAUNCH(c, "k_sep_index", k_sep_index, dim3(n_tiles), dim3(PAFFY_NT), 0, in, len, n_tiles, static_cast<unsigned long long *>(c->tile_counts.p),
               static_cast<uint32_t *>(c->sep_pos.p), cap_seps, static_cast<uint32_t *>(c->nl_idx.p), cap_lines, static_cast<DevInfo *>(c->info.p), nd);
        if (fetch_info(c)) return PAFFY_E_HIP;
        n_seps = c->h_info->n_seps;
        n_lines = c->h_info->n_lines;
        indexed = n_seps <= cap_seps && n_lines <= cap_lines; /* otherwise: denser than the guess, once more below with exact sizes */
    }
    if (!indexed) {
        if (ensure(c, c->tile_counts, sizeof(uint2) * n_tiles)) return PAFFY_E_HIP;
        LAUNCH(c, "k_sep_count", k_sep_count, dim3(n_tiles), dim3(PAFFY_NT), 0, in, len, static_cast<uint2 *>(c->tile_counts.p), nd);
        LAUNCH(c, "k_scan_tiles", k_scan_tiles, dim3(1), dim3(PAFFY_NT), 0, static_cast<uint2 *>(c->tile_counts.p), n_tiles,
               static_cast<DevInfo *>(c->info.p));
        if (fetch_info(c)) return PAFFY_E_HIP;
        n_seps = c->h_info->n_seps;
        n_lines = c->h_info->n_lines;
        if (ensure(c, c->sep_pos, sizeof(uint32_t) * (size_t)(n_seps + 1))) return PAFFY_E_HIP;
        if (ensure(c, c->nl_idx, sizeof(uint32_t) * (size_t)(n_lines + 1))) return PAFFY_E_HIP;
    }
    *n_lines_out = n_lines;
    {
        const double mib = (double)len / (1 << 20) + 1.0;
        c->sep_guess_per_mib = (double)n_seps / mib + 1.0;
        c->line_guess_per_mib = (double)n_lines / mib + 1.0;
    }

    if (ensure(c, c->meta, sizeof(RecMeta) * (size_t)(n_lines + 1))) return PAFFY_E_HIP;
    if (ensure(c, c->b_list, sizeof(uint32_t) * (size_t)(n_lines + 1))) return PAFFY_E_HIP;
    if (ensure(c, c->out_len, sizeof(int64_t) * (size_t)(n_lines + 1))) return PAFFY_E_HIP;
    if (ensure(c, c->out_rows, sizeof(int64_t) * (size_t)(n_lines + 1))) return PAFFY_E_HIP;
    if (ensure(c, c->out_off, sizeof(int64_t) * (size_t)(n_lines + 1))) return PAFFY_E_HIP;
    if (ensure(c, c->status, sizeof(uint32_t) * (size_t)(n_lines + 1))) return PAFFY_E_HIP;
    if (ensure(c, c->err_aux, sizeof(int32_t) * (size_t)(n_lines + 1))) return PAFFY_E_HIP;
    if (ensure(c, c->n_ops, sizeof(uint32_t) * (size_t)(n_lines + 1))) return PAFFY_E_HIP;
    if (ensure(c, c->arena_off, sizeof(uint64_t) * (size_t)(n_lines + 1))) return PAFFY_E_HIP;
    if (ensure(c, c->w_list, sizeof(uint32_t) * (size_t)(n_lines + 1))) return PAFFY_E_HIP;
    if (ensure(c, c->b_list, sizeof(uint32_t) * (size_t)(n_lines + 1))) return PAFFY_E_HIP;
    if (ensure(c, c->b_list1, sizeof(uint32_t) * (size_t)(n_lines + 1))) return PAFFY_E_HIP;
    if (ensure(c, c->rec_plan, sizeof(RecPlan) * (size_t)(n_lines + 1))) return PAFFY_E_HIP;
    if (ensure(c, c->ops_mirror, sizeof(uint32_t) * ((size_t)len / 2 + 64))) return PAFFY_E_HIP;
    if (c->arena.cap == 0 && ensure(c, c->arena, (size_t)8 << 20)) return PAFFY_E_HIP;

    if (!indexed)
        LAUNCH(c, "k_sep_write", k_sep_write, dim3(n_tiles), dim3(PAFFY_NT), 0, in, len, static_cast<const uint2 *>(c->tile_counts.p),
               static_cast<uint32_t *>(c->sep_pos.p), static_cast<uint32_t *>(c->nl_idx.p));
    if (flat && n_lines > 0) {
        /* summary slot of piece p of record r: (cg_off >> 10) + r + p; chunk slot of its chunk c: (cg_off >> 12) + r + c (flat_kernel.h) */
        const size_t piece_slots = ((size_t)len >> FLAT_TILE_SHIFT) + (size_t)n_lines + 8, chunk_slots = ((size_t)len >> (FLAT_TILE_SHIFT + 2)) + (size_t)n_lines + 8;
        if (ensure(c, c->flat_chunks, sizeof(uint32_t) * chunk_slots) || ensure(c, c->flat_sums, sizeof(PieceSum) * piece_slots) ||
            ensure(c, c->flat_done, (size_t)n_lines + 16))
            return PAFFY_E_HIP;
        HIPCHK(c, hipMemsetAsync(c->flat_chunks.p, 0xff, sizeof(uint32_t) * chunk_slots, c->stream)); /* FLAT_NO_CHUNK */
        c->flat_chunk_slots = (uint32_t)chunk_slots;
        c->flat_piece_slots = (uint32_t)piece_slots;
    }
    if (n_lines > 0)
        LAUNCH(c, "k_header", k_header, dim3((n_lines + PAFFY_NT / HDR_GROUP - 1) / (PAFFY_NT / HDR_GROUP)), dim3(PAFFY_NT), 0, in,
               static_cast<const uint32_t *>(c->sep_pos.p), static_cast<const uint32_t *>(c->nl_idx.p), n_lines,
               static_cast<RecMeta *>(c->meta.p), static_cast<uint32_t *>(c->b_list.p), static_cast<DevInfo *>(c->info.p), lvl0_max,
               flat ? static_cast<uint32_t *>(c->flat_chunks.p) : static_cast<uint32_t *>(nullptr));

    return 0;
}

int paffy_hip_plan(paffy_hip_ctx *c, const paffy_stage *stages, int32_t n_stages, const void *d_in, int64_t in_len,
                   paffy_plan_info *info) {
    if (!c || !info || (n_stages > 0 && !stages) || n_stages < 0 || n_stages > PAFFY_MAX_STAGES) return PAFFY_E_ARG;
    if (in_len < 0 || in_len >= (1ll << 31) - 64 || (in_len > 0 && !d_in) || (reinterpret_cast<uintptr_t>(d_in) & 15)) return PAFFY_E_ARG;
    bool need_seqs = false;
    paffy_stage norm[PAFFY_MAX_STAGES]; /* kinds without the PAFFY_NO_CHECK flag */
    uint32_t nocheck_mask = 0;
    for (int32_t i = 0; i < n_stages; i++) {
        norm[i] = stages[i];
        if (stages[i].kind & PAFFY_NO_CHECK) nocheck_mask |= 1u << i;
        norm[i].kind = stages[i].kind & ~PAFFY_NO_CHECK;
    }
    stages = norm;
    for (int32_t i = 0; i < n_stages; i++) {
        int k = stages[i].kind;
        bool ok = k == PAFFY_CHECK || k == PAFFY_INVERT || k == PAFFY_TRIM_IDENTITY || k == PAFFY_TRIM_FIXED || k == PAFFY_TRIM_ENDS || k == PAFFY_PASS || k == PAFFY_FILTER || k == PAFFY_STATS ||
                  k == PAFFY_REMOVE_MISMATCHES || (k == PAFFY_ADD_MISMATCHES && c->n_seqs > 0) ||
                  (k == PAFFY_SHATTER && i == n_stages - 1);
        if (k == PAFFY_ADD_MISMATCHES) need_seqs = true;
        if (!ok) {
            c->last_error = "stage list not fusable in this build";
            return PAFFY_E_UNSUPPORTED;
        }
    }
    bool lean = true; /* only stage kinds the lean sizing kernel knows */
    for (int32_t i = 0; i < n_stages; i++) lean = lean && ((PAFFY_MASK_LEAN >> stages[i].kind) & 1u);
    bool lean_add = !lean; /* the lean kinds and add_mismatches: its own instantiation (the encoder wants the registers) */
    for (int32_t i = 0; i < n_stages; i++) lean_add = lean_add && ((PAFFY_MASK_ADD >> stages[i].kind) & 1u);
    bool sel = !lean && !lean_add; /* the lean kinds with filter / trim -f / stats / check */
    for (int32_t i = 0; i < n_stages; i++) sel = sel && ((PAFFY_MASK_SEL >> stages[i].kind) & 1u);
    bool plain = !lean; /* no stage of the kinds that came with the encoder: the instantiation without them */
    for (int32_t i = 0; i < n_stages; i++) plain = plain && ((PAFFY_MASK_PLAIN >> stages[i].kind) & 1u);
    c->planned = false;
    c->plan_is_tile = false;
    c->plan_is_bed = false;
    c->flat_left = -1;
    memset(info, 0, sizeof(*info));
    info->in_bytes = in_len;
    memset(&c->plan, 0, sizeof(c->plan));
    c->plan.in_bytes = in_len;
    KParams &kp = c->kp;
    memset(&kp, 0, sizeof(kp));
    if (in_len == 0) {
        c->planned = true;
        kp.n_rec = 0;
        return 0;
    }
    const uint8_t *in = static_cast<const uint8_t *>(d_in);
    const uint32_t len = (uint32_t)in_len;
    uint32_t n_lines = 0;
    /* add_mismatches makes more ops of a cigar (about 1.8 x at 2 % substitutions, 1.45 per two cigar bytes): when a stage follows it, its
       records start one store level up at 5/8 of the usual length, so that the rebuilt array usually fits the level the record was parsed at */
    const bool add_not_last = need_seqs && [&] {
        for (int32_t i = 0; i + 1 < n_stages; i++)
            if (stages[i].kind == PAFFY_ADD_MISMATCHES) return true;
        return false;
    }();
    /* Cigars of more than 2 x lvl0_max bytes start at the second store level (k_header queues them), whose 64 KB of ops leave a CU two
       workgroups: on cfg3 / cfg4 the 7 % of the records there cost as much kernel time as all the others. Two bytes per op is the bound
       that can never overflow the first level's 8 192-op store, but a cigar of the usual density (2.5-3 bytes per op) fits it up to about
       LVL0_LONG_BYTES. So a context starts with the safe bound and lets the second level count the records that would have overflowed the
       longer one (DevInfo::lvl0_probe_dense); a batch without any switches the following batches to the longer first level (cfg4 12.23 ->
       11.72 ms per step, cfg3 -1.2 to -1.6 %), and the first record that does overflow there (it goes to the arena class: slow, correct)
       switches the context back for good. PAFFY_LVL0_BYTES overrides the length (0: never). */
    static const uint32_t lvl0_long_bytes = getenv("PAFFY_LVL0_BYTES") ? (uint32_t)atol(getenv("PAFFY_LVL0_BYTES")) : LVL0_LONG_BYTES;
    const bool lvl0_long = !add_not_last && c->lvl0_long_ok && lvl0_long_bytes > 2u * PAFFY_OPS_CAP;
    const uint32_t lvl0_max = add_not_last ? PAFFY_OPS_CAP * 5 / 8 : (lvl0_long ? lvl0_long_bytes / 2u : PAFFY_OPS_CAP);
    /* records with at most this many cigar bytes (about WAVE_OPS_CAP ops at three bytes per op) are sized one wave per record; denser
       cigars of that length overflow the wave's store and are redone by the four-wave build. add_mismatches rebuilds the op array: when
       a stage follows it the new array must fit the store, so only pipes that end with it take the one-wave build. */
    static const uint32_t wave_env = getenv("PAFFY_WAVE_BYTES") ? (uint32_t)atoi(getenv("PAFFY_WAVE_BYTES")) : WAVE_MAX_BYTES;
    static const uint32_t wave_cap_env = getenv("PAFFY_WAVE_OPS") ? (uint32_t)atoi(getenv("PAFFY_WAVE_OPS")) : WAVE_OPS_CAP;
    const uint32_t wave_bytes = (need_seqs && add_not_last) ? 0u : wave_env;
    /* the lean pipes are sized by the flat pass (flat_kernel.h): the text parsed in chunks whatever record they belong to, one wave per
       record on the chunks' summaries; what it leaves (FLAT_F_IRREG and friends) goes through the record kernels below as before */
    static const bool flat_off = getenv("PAFFY_NO_FLAT") != nullptr;
    /* `paffy add_mismatches` alone (BASELINE cfg4): the same parse, the encoder on the pieces (flat_add_kernel.h) */
    const bool flat_add = n_stages == 1 && stages[0].kind == PAFFY_ADD_MISMATCHES && nocheck_mask == 0 && !flat_off && c->n_seqs > 0;
    bool lean_or_filter = n_stages > 0; /* the flat pass also knows `paffy filter` (a predicate on the sums it keeps anyway) */
    /* ... and the stats stage of `paffy view -s` (paf_stats_calc, impl/paf.c:236-260: sums the pieces' summaries hold, with the I ops counted
       where a shatter pipe counts the digits of its rows -- so not both in one pipe) */
    bool has_stats = false, has_shatter_stage = false;
    for (int32_t i = 0; i < n_stages; i++) {
        has_stats = has_stats || stages[i].kind == PAFFY_STATS;
        has_shatter_stage = has_shatter_stage || stages[i].kind == PAFFY_SHATTER;
    }
    /* ... and a fixed trim (`paffy trim -f`) as the pipe's last stage: the wave kernel finds the two ops it stops at (flat_find_aligned) */
    const bool fixed_last = n_stages > 0 && stages[n_stages - 1].kind == PAFFY_TRIM_FIXED;
    const uint32_t flat_kinds = PAFFY_MASK_LEAN | (1u << PAFFY_FILTER) | (has_shatter_stage ? 0u : 1u << PAFFY_STATS);
    for (int32_t i = 0; i < n_stages; i++)
        lean_or_filter = lean_or_filter && (((flat_kinds >> stages[i].kind) & 1u) || (fixed_last && i == n_stages - 1));
    const bool flat = (lean_or_filter && nocheck_mask == 0 && !flat_off) || flat_add;
    {
        int rc = index_and_parse(c, in, len, &n_lines, lvl0_max, flat);
        if (rc) return rc;
    }
    kp.lvl0_max = lvl0_max;
    kp.lvl0_long_bytes = (!add_not_last && !lvl0_long && !c->lvl0_long_off && lvl0_long_bytes > 2u * PAFFY_OPS_CAP) ? lvl0_long_bytes : 0u;

    kp.in = in;
    kp.in_len = len;
    kp.n_rec = n_lines;
    kp.meta = static_cast<const RecMeta *>(c->meta.p);
    for (int32_t i = 0; i < n_stages; i++) kp.stages[i] = stages[i];
    kp.n_stages = n_stages;
    kp.nocheck_mask = nocheck_mask;
    kp.out_len = static_cast<int64_t *>(c->out_len.p);
    kp.out_rows = static_cast<int64_t *>(c->out_rows.p);
    kp.status = static_cast<uint32_t *>(c->status.p);
    kp.err_aux = static_cast<int32_t *>(c->err_aux.p);
    kp.n_ops = static_cast<uint32_t *>(c->n_ops.p);
    kp.arena_off = static_cast<uint64_t *>(c->arena_off.p);
    kp.rec_plan = c->rec_plan.p;
    kp.ops_mirror = static_cast<uint32_t *>(c->ops_mirror.p);
    c->plan_seq_lookup = need_seqs && n_lines > 0;
    if (need_seqs && n_lines > 0) {
        if (ensure(c, c->rec_qseq, sizeof(int32_t) * (size_t)(n_lines + 1))) return PAFFY_E_HIP;
        if (ensure(c, c->rec_tseq, sizeof(int32_t) * (size_t)(n_lines + 1))) return PAFFY_E_HIP;
        LAUNCH(c, "k_seq_lookup", k_seq_lookup, dim3((n_lines + PAFFY_NT - 1) / PAFFY_NT), dim3(PAFFY_NT), 0, in,
               static_cast<const RecMeta *>(c->meta.p), n_lines, static_cast<const uint8_t *>(c->seq_names.p),
               static_cast<const uint32_t *>(c->seq_name_off.p), c->n_seqs, static_cast<int32_t *>(c->rec_qseq.p),
               static_cast<int32_t *>(c->rec_tseq.p));
        kp.seq_base = static_cast<const uint8_t *>(c->seq_blob.p);
        kp.seq_comp = static_cast<const uint8_t *>(c->seq_comp.p);
        kp.seqs = static_cast<const SeqEntry *>(c->seq_table.p);
        kp.rec_qseq = static_cast<const int32_t *>(c->rec_qseq.p);
        kp.rec_tseq = static_cast<const int32_t *>(c->rec_tseq.p);
    }
    kp.out_off = static_cast<const int64_t *>(c->out_off.p);
    kp.w_list = static_cast<uint32_t *>(c->w_list.p);
    kp.b_list[0] = static_cast<uint32_t *>(c->b_list.p);
    kp.b_list[1] = static_cast<uint32_t *>(c->b_list1.p);
    kp.info = static_cast<DevInfo *>(c->info.p);
    kp.filter = c->filter;
    for (int32_t i = 0; i < n_stages; i++)
        if (stages[i].kind == PAFFY_STATS && n_lines > 0) {
            if (ensure(c, c->rec_stats, sizeof(int64_t) * 6 * (size_t)n_lines)) return PAFFY_E_HIP;
            kp.rec_stats = static_cast<int64_t *>(c->rec_stats.p);
            HIPCHK(c, hipMemsetAsync(c->rec_stats.p, 0, sizeof(int64_t) * 6 * (size_t)n_lines, c->stream)); /* records that stop before the stage */
        }

    auto post_scans = [&]() -> int {
        /* the scan rides along: one host synchronisation per plan in the usual case (the arena was big enough) */
        {
            const uint32_t n_blocks = (n_lines + SCAN_BLOCK - 1) / SCAN_BLOCK;
            if (ensure(c, c->scan_part, sizeof(int64_t) * 2 * (size_t)n_blocks)) return PAFFY_E_HIP;
            LAUNCH(c, "k_scan_part", k_scan_part, dim3(n_blocks), dim3(PAFFY_NT), 0, kp.out_len, kp.out_rows, n_lines,
                   static_cast<int64_t *>(c->out_off.p), static_cast<int64_t *>(c->scan_part.p), kp.info);
            LAUNCH(c, "k_scan_fix", k_scan_fix, dim3(n_blocks), dim3(PAFFY_NT), 0, n_lines, n_blocks, static_cast<int64_t *>(c->out_off.p),
                   static_cast<const int64_t *>(c->scan_part.p), kp.info);
        }
        {
            if (ensure(c, c->emit_order, sizeof(uint32_t) * (size_t)n_lines)) return PAFFY_E_HIP;
            if (ensure(c, c->order_cnt, sizeof(uint32_t) * 2 * ORDER_CLASSES)) return PAFFY_E_HIP;
            uint32_t *cnt = static_cast<uint32_t *>(c->order_cnt.p);
            HIPCHK(c, hipMemsetAsync(cnt, 0, sizeof(uint32_t) * 2 * ORDER_CLASSES, c->stream));
            const uint32_t g = (n_lines + PAFFY_NT - 1) / PAFFY_NT;
            LAUNCH(c, "k_order_count", k_order_count, dim3(g), dim3(PAFFY_NT), 0, kp.out_len, n_lines, 14u, cnt);
            LAUNCH(c, "k_order_scatter", k_order_scatter, dim3(g), dim3(PAFFY_NT), 0, kp.out_len, n_lines, 14u, cnt, cnt + ORDER_CLASSES,
                   static_cast<uint32_t *>(c->emit_order.p));
            kp.emit_order = static_cast<const uint32_t *>(c->emit_order.p);
        }
        return 0;
    };
    uint32_t flat_g_count = 0;
    bool need_legacy = true;
    if (flat_add && n_lines > 0) {
        FlatParams fp;
        fp.in = in;
        fp.in_len = len;
        fp.meta = kp.meta;
        fp.chunk_rec = static_cast<const uint32_t *>(c->flat_chunks.p);
        fp.n_chunk_slots = c->flat_chunk_slots;
        fp.nd = static_cast<const uint16_t *>(c->flat_nd.p);
        fp.sums = static_cast<PieceSum *>(c->flat_sums.p);
        fp.ops_mirror = kp.ops_mirror;
        fp.info = kp.info;
        fp.items_mode = 1;
        LAUNCH(c, "k_flat_parse", k_flat_parse<1u>, dim3(2048), dim3(64 * FLAT_PARSE_WAVES), 0, fp);
        const uint32_t n_slots = c->flat_piece_slots, n_sblocks = (n_slots + SCAN32_BLOCK - 1) / SCAN32_BLOCK;
        if (ensure(c, c->add_pieces, sizeof(AddPiece) * (size_t)n_slots) || ensure(c, c->add_scr_cnt, sizeof(uint32_t) * (size_t)n_slots) ||
            ensure(c, c->add_scr_off, sizeof(uint64_t) * (size_t)n_slots) || ensure(c, c->add_new_cnt, sizeof(uint32_t) * (size_t)n_slots) ||
            ensure(c, c->add_new_off, sizeof(uint64_t) * (size_t)n_slots) || ensure(c, c->add_text, sizeof(uint32_t) * (size_t)n_slots) ||
            ensure(c, c->add_bad, sizeof(uint32_t) * (size_t)(n_lines + 1)) || ensure(c, c->add_part, sizeof(uint64_t) * (size_t)(n_sblocks + 1)))
            return PAFFY_E_HIP;
        /* item words + passed-through ops, and the new 4-byte ops: about 2.4 and 2.9 bytes per byte of text for 2 %-divergent sequences; a
           batch that needs more is encoded again with what it asked for (the totals come back with the plan's one synchronisation) */
        size_t scr_words = std::max<size_t>(c->add_scratch.cap / 4, (size_t)len + ((size_t)1 << 20)), new_words = std::max<size_t>(c->add_new_ops.cap / 4, (size_t)len + ((size_t)1 << 20));
        for (int attempt = 0;; attempt++) {
            if (ensure(c, c->add_scratch, 4 * scr_words) || ensure(c, c->add_new_ops, 4 * new_words)) return PAFFY_E_HIP;
            AddParams ap;
            ap.P = kp;
            ap.sums = fp.sums;
            ap.pieces = static_cast<AddPiece *>(c->add_pieces.p);
            ap.n_piece_slots = n_slots;
            ap.scr_cnt = static_cast<uint32_t *>(c->add_scr_cnt.p);
            ap.scr_off = static_cast<const uint64_t *>(c->add_scr_off.p);
            ap.scratch = static_cast<uint32_t *>(c->add_scratch.p);
            ap.scr_cap = c->add_scratch.cap / 4;
            ap.new_cnt = static_cast<uint32_t *>(c->add_new_cnt.p);
            ap.new_off = static_cast<const uint64_t *>(c->add_new_off.p);
            ap.new_ops = static_cast<uint32_t *>(c->add_new_ops.p);
            ap.new_cap = c->add_new_ops.cap / 4;
            ap.text_cnt = static_cast<uint32_t *>(c->add_text.p);
            ap.rec_bad = static_cast<uint32_t *>(c->add_bad.p);
            ap.flat_done = static_cast<uint8_t *>(c->flat_done.p);
            { /* segments of the lines that become more than PAFFY_ROWS_MAX_OPS ops */
                const size_t items_cap = new_words / (PAFFY_ROWS_MAX_OPS / 4u) + ((size_t)len >> 14) + 64;
                if (ensure(c, c->flat_items, sizeof(EmitItem) * items_cap)) return PAFFY_E_HIP;
                kp.items = static_cast<EmitItem *>(c->flat_items.p);
                kp.items_cap = (uint32_t)items_cap;
                ap.P = kp;
            }
            HIPCHK(c, hipMemsetAsync(c->add_pieces.p, 0xff, sizeof(AddPiece) * (size_t)n_slots, c->stream)); /* rec = FLAT_NO_CHUNK */
            HIPCHK(c, hipMemsetAsync(c->add_scr_cnt.p, 0, sizeof(uint32_t) * (size_t)n_slots, c->stream));
            HIPCHK(c, hipMemsetAsync(c->add_new_cnt.p, 0, sizeof(uint32_t) * (size_t)n_slots, c->stream));
            LAUNCH(c, "k_add_prep", k_add_prep, dim3((n_lines + 255) / 256), dim3(256), 0, ap);
            LAUNCH(c, "k_scan32_part", k_scan32_part, dim3(n_sblocks), dim3(256), 0, ap.scr_cnt, n_slots, static_cast<uint64_t *>(c->add_scr_off.p), static_cast<uint64_t *>(c->add_part.p));
            LAUNCH(c, "k_scan32_fix", k_scan32_fix, dim3(n_sblocks), dim3(256), 0, n_slots, n_sblocks, static_cast<uint64_t *>(c->add_scr_off.p), static_cast<const uint64_t *>(c->add_part.p),
                   reinterpret_cast<uint64_t *>(&static_cast<DevInfo *>(c->info.p)->add_scr_total));
            LAUNCH(c, "k_add_count", k_add_count, dim3(2048), dim3(64 * ADD_WAVES), 0, ap);
            LAUNCH(c, "k_scan32_part", k_scan32_part, dim3(n_sblocks), dim3(256), 0, ap.new_cnt, n_slots, static_cast<uint64_t *>(c->add_new_off.p), static_cast<uint64_t *>(c->add_part.p));
            LAUNCH(c, "k_scan32_fix", k_scan32_fix, dim3(n_sblocks), dim3(256), 0, n_slots, n_sblocks, static_cast<uint64_t *>(c->add_new_off.p), static_cast<const uint64_t *>(c->add_part.p),
                   reinterpret_cast<uint64_t *>(&static_cast<DevInfo *>(c->info.p)->add_new_total));
            LAUNCH(c, "k_add_fill", k_add_fill, dim3(2048), dim3(64 * ADD_WAVES), 0, ap);
            LAUNCH(c, "k_add_final", k_add_final, dim3((n_lines + 255) / 256), dim3(256), 0, ap);
            kp.new_ops = ap.new_ops;
            if (post_scans()) return PAFFY_E_HIP;
            if (fetch_info(c)) return PAFFY_E_HIP;
            kp.n_items = c->h_info->n_items;
            if (c->h_info->add_scr_total <= ap.scr_cap && c->h_info->add_new_total <= ap.new_cap) break;
            if (attempt == 2) {
                c->last_error = "add_mismatches: the scratch demand kept growing";
                return PAFFY_E_HIP;
            }
            scr_words = std::max<size_t>(scr_words, (size_t)c->h_info->add_scr_total + ((size_t)c->h_info->add_scr_total >> 3) + 1024);
            new_words = std::max<size_t>(new_words, (size_t)c->h_info->add_new_total + ((size_t)c->h_info->add_new_total >> 3) + 1024);
            DevInfo z = *c->h_info; /* once more: nothing of the first try counts */
            z.flat_legacy = 0;
            z.n_items = 0;
            z.out_bytes = z.out_rows = 0;
            HIPCHK(c, hipMemcpyAsync(c->info.p, &z, sizeof(z), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        need_legacy = c->h_info->flat_legacy > 0;
        c->flat_left = c->h_info->flat_legacy;
        for (int k = 0; k < 16; k++) c->flat_reasons[k] = 0;
        kp.flat_done = static_cast<const uint8_t *>(c->flat_done.p);
    } else if (flat && n_lines > 0) {
        FlatParams fp;
        fp.in = in;
        fp.in_len = len;
        fp.meta = kp.meta;
        fp.chunk_rec = static_cast<const uint32_t *>(c->flat_chunks.p);
        fp.n_chunk_slots = c->flat_chunk_slots;
        fp.nd = static_cast<const uint16_t *>(c->flat_nd.p);
        fp.sums = static_cast<PieceSum *>(c->flat_sums.p);
        fp.ops_mirror = kp.ops_mirror;
        fp.info = kp.info;
        fp.items_mode = has_stats ? 2u : 0u;
        /* persistent waves over the chunks: eight workgroups of four waves per CU */
        if (has_stats) LAUNCH(c, "k_flat_parse", k_flat_parse<2u>, dim3(2048), dim3(64 * FLAT_PARSE_WAVES), 0, fp);
        else LAUNCH(c, "k_flat_parse", k_flat_parse<0u>, dim3(2048), dim3(64 * FLAT_PARSE_WAVES), 0, fp);
        /* segments of the shatter records too long for one wave of the row writer: a record of more than PAFFY_ROWS_MAX_OPS ops has
           2 x that many cigar bytes at least, a segment holds half that many ops */
        const size_t items_cap = ((size_t)len >> 15) + ((size_t)len >> 16) + 16;
        if (ensure(c, c->flat_items, sizeof(EmitItem) * items_cap)) return PAFFY_E_HIP;
        kp.items = static_cast<EmitItem *>(c->flat_items.p);
        kp.items_cap = (uint32_t)items_cap;
        /* the constant pieces of a shatter record's rows, 144 bytes per record (lane_row_pieces, flat_kernel.h) */
        static const bool no_pieces = getenv("PAFFY_NO_ROW_PIECES") != nullptr;
        const bool has_shatter = [&] {
            for (int32_t i = 0; i < n_stages; i++)
                if (stages[i].kind == PAFFY_SHATTER) return true;
            return false;
        }();
        if (has_shatter && !no_pieces) {
            if (ensure(c, c->flat_pieces, FLAT_ROW_PIECES_BYTES * (size_t)(n_lines + 1))) return PAFFY_E_HIP;
            kp.row_pieces = static_cast<uint8_t *>(c->flat_pieces.p);
        }
        FlatSizeParams fs;
        fs.P = kp;
        fs.sums = fp.sums;
        fs.flat_done = static_cast<uint8_t *>(c->flat_done.p);
        if (ensure(c, c->flat_rec, sizeof(uint32_t) * (size_t)(n_lines + 1))) return PAFFY_E_HIP;
        fs.defer = static_cast<uint32_t *>(c->flat_rec.p);
        /* one lane per record for the records that need little, one wave per record for the rest (a list whose length only the device knows) */
        if (fixed_last) { /* one wave per record for all of them */
            LAUNCH(c, "k_flat_size", k_flat_size<true>, dim3(std::min<uint32_t>(2048u, (n_lines + FLAT_SIZE_WAVES - 1) / FLAT_SIZE_WAVES)), dim3(64 * FLAT_SIZE_WAVES), 0, fs);
        } else {
            LAUNCH(c, "k_flat_lane", k_flat_lane, dim3((n_lines + 255) / 256), dim3(256), 0, fs);
            LAUNCH(c, "k_flat_size", k_flat_size<false>, dim3(std::min<uint32_t>(2048u, (n_lines + FLAT_SIZE_WAVES - 1) / FLAT_SIZE_WAVES)), dim3(64 * FLAT_SIZE_WAVES), 0, fs);
        }
        if (post_scans()) return PAFFY_E_HIP;
        if (fetch_info(c)) return PAFFY_E_HIP;
        flat_g_count = c->h_info->g_count;
        kp.n_items = c->h_info->n_items;
        need_legacy = c->h_info->flat_legacy > 0;
        c->flat_left = c->h_info->flat_legacy;
        for (int k = 0; k < 16; k++) c->flat_reasons[k] = c->h_info->flat_reason[k];
        kp.flat_done = static_cast<const uint8_t *>(c->flat_done.p);
        if (kp.rec_stats && !need_legacy) { /* the batch's sums (the record kernels' launch chain does this when it runs) */
            LAUNCH(c, "k_stats_reduce", k_stats_reduce, dim3(std::min<uint32_t>(512u, (n_lines + PAFFY_NT - 1) / PAFFY_NT)), dim3(PAFFY_NT), 0, kp.rec_stats, n_lines,
                   static_cast<DevInfo *>(c->info.p)->stats);
            if (fetch_info(c)) return PAFFY_E_HIP;
        }
    }
    if (n_lines > 0 && need_legacy) {
        if (!flat) { /* launch order of the sizing workgroups: long cigars first (out_len is scratch until the sizing pass fills it) */
            if (ensure(c, c->emit_order, sizeof(uint32_t) * (size_t)n_lines)) return PAFFY_E_HIP;
            if (ensure(c, c->order_cnt, sizeof(uint32_t) * 2 * ORDER_CLASSES)) return PAFFY_E_HIP;
            uint32_t *cnt = static_cast<uint32_t *>(c->order_cnt.p);
            HIPCHK(c, hipMemsetAsync(cnt, 0, sizeof(uint32_t) * 2 * ORDER_CLASSES, c->stream));
            const uint32_t g = (n_lines + PAFFY_NT - 1) / PAFFY_NT;
            LAUNCH(c, "k_cigar_bytes", k_cigar_bytes, dim3(g), dim3(PAFFY_NT), 0, kp.meta, n_lines, kp.out_len);
            LAUNCH(c, "k_order_count", k_order_count, dim3(g), dim3(PAFFY_NT), 0, kp.out_len, n_lines, 9u, cnt);
            LAUNCH(c, "k_order_scatter", k_order_scatter, dim3(g), dim3(PAFFY_NT), 0, kp.out_len, n_lines, 9u, cnt, cnt + ORDER_CLASSES,
                   static_cast<uint32_t *>(c->emit_order.p));
            kp.size_order = static_cast<const uint32_t *>(c->emit_order.p); /* any permutation serves a repeated sizing pass too */
        }
        /* add_mismatches keeps one word per 16 columns and the new 4-byte ops of every record in the arena: about eight times
           the text for 2 %-divergent sequences; start there instead of finding out through repeated passes */
        if (need_seqs && ensure(c, c->arena, (size_t)len * 8 + ((size_t)8 << 20))) return PAFFY_E_HIP;
        const int max_attempts = 6;
        for (int attempt = 0; attempt < max_attempts; attempt++) {
            kp.arena = static_cast<uint64_t *>(c->arena.p);
            kp.arena_cap = c->arena.cap / 8;
            /* fork: levels 1 and 2 (long cigars, queued by k_header) on the side stream, level 0 on the main one */
            HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
            {
                KParams k1 = kp;
                k1.ops_cap = PAFFY_OPS_CAP_MID;
                k1.next_cap = PAFFY_OPS_CAP_BIG;
                k1.level = 1;
                if (lean) hipLaunchKernelGGL(k_size_lds_long<PAFFY_MASK_LEAN>, dim3(768), dim3(PAFFY_NT), PAFFY_SIZE_LDS_BYTES_FOR(PAFFY_OPS_CAP_MID), c->side, k1);
                else if (lean_add) hipLaunchKernelGGL(k_size_lds_long<PAFFY_MASK_ADD>, dim3(768), dim3(PAFFY_NT), PAFFY_SIZE_LDS_BYTES_FOR(PAFFY_OPS_CAP_MID), c->side, k1);
                else if (sel) hipLaunchKernelGGL(k_size_lds_long<PAFFY_MASK_SEL>, dim3(768), dim3(PAFFY_NT), PAFFY_SIZE_LDS_BYTES_FOR(PAFFY_OPS_CAP_MID), c->side, k1);
                else if (plain) hipLaunchKernelGGL(k_size_lds_long<PAFFY_MASK_PLAIN>, dim3(768), dim3(PAFFY_NT), PAFFY_SIZE_LDS_BYTES_FOR(PAFFY_OPS_CAP_MID), c->side, k1);
                else hipLaunchKernelGGL(k_size_lds_long<PAFFY_MASK_ALL>, dim3(768), dim3(PAFFY_NT), PAFFY_SIZE_LDS_BYTES_FOR(PAFFY_OPS_CAP_MID), c->side, k1);
                k1.ops_cap = PAFFY_OPS_CAP_BIG;
                k1.next_cap = 0;
                k1.level = 2;
                if (lean) hipLaunchKernelGGL(k_size_lds_long<PAFFY_MASK_LEAN>, dim3(256), dim3(PAFFY_NT), PAFFY_SIZE_LDS_BYTES_FOR(PAFFY_OPS_CAP_BIG), c->side, k1);
                else if (lean_add) hipLaunchKernelGGL(k_size_lds_long<PAFFY_MASK_ADD>, dim3(256), dim3(PAFFY_NT), PAFFY_SIZE_LDS_BYTES_FOR(PAFFY_OPS_CAP_BIG), c->side, k1);
                else if (sel) hipLaunchKernelGGL(k_size_lds_long<PAFFY_MASK_SEL>, dim3(256), dim3(PAFFY_NT), PAFFY_SIZE_LDS_BYTES_FOR(PAFFY_OPS_CAP_BIG), c->side, k1);
                else if (plain) hipLaunchKernelGGL(k_size_lds_long<PAFFY_MASK_PLAIN>, dim3(256), dim3(PAFFY_NT), PAFFY_SIZE_LDS_BYTES_FOR(PAFFY_OPS_CAP_BIG), c->side, k1);
                else hipLaunchKernelGGL(k_size_lds_long<PAFFY_MASK_ALL>, dim3(256), dim3(PAFFY_NT), PAFFY_SIZE_LDS_BYTES_FOR(PAFFY_OPS_CAP_BIG), c->side, k1);
                HIPCHK(c, hipGetLastError());
                HIPCHK(c, hipEventRecord(c->ev_join, c->side));
            }
            kp.ops_cap = PAFFY_OPS_CAP;
            kp.next_cap = 0;
            kp.level = 0;
            kp.wave_max_bytes = wave_bytes;
            if (wave_bytes) { /* short cigars: one wave per record, sixteen records in flight per CU */
                KParams kw = kp;
                kw.ops_cap = wave_cap_env;
                const size_t wlds = (size_t)wave_cap_env * 4 + PAFFY_HALO + 64 * 16 + 64 * 8 + 64;
                if (lean) LAUNCH(c, "k_size_wave", g64::k_size_lds<PAFFY_MASK_LEAN>, dim3(n_lines), dim3(64), wlds + dbg_lds_pad(1), kw);
                else if (lean_add) LAUNCH(c, "k_size_wave", g64::k_size_lds<PAFFY_MASK_ADD>, dim3(n_lines), dim3(64), wlds, kw);
                else if (sel) LAUNCH(c, "k_size_wave", g64::k_size_lds<PAFFY_MASK_SEL>, dim3(n_lines), dim3(64), wlds, kw);
                else if (plain) LAUNCH(c, "k_size_wave", g64::k_size_lds<PAFFY_MASK_PLAIN>, dim3(n_lines), dim3(64), wlds, kw);
                else LAUNCH(c, "k_size_wave", g64::k_size_lds<PAFFY_MASK_ALL>, dim3(n_lines), dim3(64), wlds, kw);
            }
            if (lean) LAUNCH(c, "k_size_lds", k_size_lds<PAFFY_MASK_LEAN>, dim3(n_lines), dim3(PAFFY_NT), PAFFY_SIZE_LDS_BYTES + dbg_lds_pad(2), kp);
            else if (lean_add) LAUNCH(c, "k_size_lds", k_size_lds<PAFFY_MASK_ADD>, dim3(n_lines), dim3(PAFFY_NT), PAFFY_SIZE_LDS_BYTES, kp);
            else if (sel) LAUNCH(c, "k_size_lds", k_size_lds<PAFFY_MASK_SEL>, dim3(n_lines), dim3(PAFFY_NT), PAFFY_SIZE_LDS_BYTES, kp);
            else if (plain) LAUNCH(c, "k_size_lds", k_size_lds<PAFFY_MASK_PLAIN>, dim3(n_lines), dim3(PAFFY_NT), PAFFY_SIZE_LDS_BYTES, kp);
            else LAUNCH(c, "k_size_lds", k_size_lds<PAFFY_MASK_ALL>, dim3(n_lines), dim3(PAFFY_NT), PAFFY_SIZE_LDS_BYTES, kp);
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0)); /* join */
            LAUNCH(c, "k_arena_size", k_arena_size, dim3(2048), dim3(PAFFY_NT), PAFFY_SIZE_LDS_BYTES, kp);
            if (kp.rec_stats)
                LAUNCH(c, "k_stats_reduce", k_stats_reduce, dim3(std::min<uint32_t>(512u, (n_lines + PAFFY_NT - 1) / PAFFY_NT)), dim3(PAFFY_NT), 0, kp.rec_stats, n_lines,
                       static_cast<DevInfo *>(c->info.p)->stats);
            if (post_scans()) return PAFFY_E_HIP;
            if (fetch_info(c)) return PAFFY_E_HIP;
            if (c->h_info->arena_used <= kp.arena_cap) {
                if (lvl0_long && c->h_info->lvl0_over > 0) { /* denser cigars than the longer first level takes: back to the safe bound, for good */
                    c->lvl0_long_ok = false;
                    c->lvl0_long_off = true;
                } else if (kp.lvl0_long_bytes) { /* the safe bound was in force and the second level kept count */
                    c->lvl0_long_ok = c->h_info->lvl0_probe_dense == 0;
                }
                break;
            }
            /* arena too small: grow to the demand seen so far and redo the sizing pass */
            /* a record that found no room stopped asking, so the demand seen is a lower bound: at least double what there was */
            size_t need = (size_t)c->h_info->arena_used * 8 * 2;
            if (need < c->arena.cap * 2) need = c->arena.cap * 2;
            if (ensure(c, c->arena, need)) return PAFFY_E_HIP;
            DevInfo z = *c->h_info;
            z.arena_used = 0;
            z.w_count = 0;
            z.g_count = flat_g_count; /* the records the flat pass left to the four-wave writers stay counted */
            z.b_count[1] = 0; /* b_count[0] was filled by k_header and stays */
            for (int k = 0; k < 6; k++) z.stats[k] = 0;
            z.first_err_key = ~0ull;
            z.out_bytes = z.out_rows = 0;
            HIPCHK(c, hipMemcpyAsync(c->info.p, &z, sizeof(z), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (attempt == max_attempts - 1) {
                c->last_error = "arena demand kept growing";
                return PAFFY_E_HIP;
            }
        }
    } else if (n_lines == 0 && fetch_info(c)) {
        return PAFFY_E_HIP;
    }
    if (c->profile) prof_collect(c);
    if (c->h_info->internal) {
        char buf[128];
        snprintf(buf, sizeof(buf), "internal limit hit (flags 0x%x): record pieces longer than the LDS staging area", c->h_info->internal);
        c->last_error = buf;
        return PAFFY_E_UNSUPPORTED;
    }
    paffy_plan_info &pl = c->plan;
    pl.n_records = n_lines;
    pl.n_rows = (int64_t)c->h_info->out_rows;
    pl.out_bytes = (int64_t)c->h_info->out_bytes;
    if (c->h_info->first_err_key != ~0ull) {
        unsigned long long k = c->h_info->first_err_key;
        pl.error.code = (int32_t)(k & 0xff);
        pl.error.stage = (int32_t)((k >> 8) & 0xff) - 1;
        pl.error.record = (int64_t)(k >> 16);
        int32_t aux = 0;
        HIPCHK(c, hipMemcpy(&aux, static_cast<int32_t *>(c->err_aux.p) + pl.error.record, sizeof(aux), hipMemcpyDeviceToHost));
        pl.error.aux = aux;
    }
    *info = pl;
    c->planned = true;
    return 0;
}


} /* extern "C" */

#include "coverage_host.h"
#include "dedupe_host.h"
#include "chain_host.h"
#include "pretty_kernel.h"

static CovState &cov_state(paffy_hip_ctx *c) {
    if (!c->cov) c->cov = new CovState();
    return *c->cov;
}
static ChainState &chain_state(paffy_hip_ctx *c) {
    if (!c->chain) c->chain = new ChainState();
    return *c->chain;
}
static void chain_free(paffy_hip_ctx *c) {
    if (!c->chain) return;
    ChainState &H = *c->chain;
    DevBuf *bufs[] = {&H.qkey, &H.ghash, &H.ord1, &H.ord2, &H.rank, &H.start, &H.gid, &H.idx, &H.prank, &H.pred, &H.neg, &H.taken, &H.is_tail, &H.tail_of, &H.link, &H.total,
                      &H.chain_of_tail, &H.chain_id, &H.score_key, &H.o1, &H.o2, &H.o3, &H.cls, &H.tag_chain, &H.tag_score, &H.check_key, &H.iota, &H.big_list, &H.n_big, &H.rank_of, &H.claim};
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (DevBuf &b : H.i64)
        if (b.p) (void)hipFree(b.p);
    delete c->chain;
    c->chain = nullptr;
}
static void cov_free(paffy_hip_ctx *c) {
    if (!c->cov) return;
    CovState &S = *c->cov;
    DevBuf *bufs[] = {&S.meta, &S.batch_ptrs, &S.key_score, &S.key_chain, &S.idx, &S.info, &S.tmp, &S.k64a, &S.k64b, &S.v32a, &S.v32b, &S.order, &S.entries, &S.name_hash,
                      &S.seq_len, &S.flags, &S.scan32, &S.first_entry, &S.contig_len, &S.contig_cov, &S.contig_slice0, &S.bm_words, &S.n_pairs, &S.bm_off, &S.pair_off,
                      &S.aligned, &S.bitmap, &S.pairs, &S.pairs2, &S.item_start, &S.item_key, &S.item_idx, &S.item_key2, &S.item_order, &S.slots, &S.arena, &S.arena_used,
                      &S.cov, &S.backup, &S.level, &S.err_aux, &S.out_len, &S.out_off, &S.names, &S.name_tab};
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    delete c->cov;
    c->cov = nullptr;
}

extern "C" {

/* the lines of S.order (records) with the levels of S.level: sizes, offsets, and the line table emit reads */
static int lines_plan(paffy_hip_ctx *c, CovState &S, uint64_t n) {
    if (ensure(c, S.out_len, sizeof(uint64_t) * (size_t)(n + 1))) return PAFFY_E_HIP;
    if (ensure(c, S.out_off, sizeof(uint64_t) * (size_t)(n + 1))) return PAFFY_E_HIP;
    LAUNCH(c, "k_line_size", k_line_size, dim3((unsigned)((n + 1 + PAFFY_NT - 1) / PAFFY_NT)), dim3(PAFFY_NT), 0, static_cast<const RecMeta *>(S.meta.p),
           static_cast<const uint32_t *>(S.order.p), static_cast<const int64_t *>(S.level.p), n, static_cast<uint64_t *>(S.out_len.p));
    if (cov_excl_scan64(c, S, static_cast<uint64_t *>(S.out_len.p), static_cast<uint64_t *>(S.out_off.p), (size_t)n)) return PAFFY_E_HIP;
    uint64_t total = 0;
    if (cov_fetch(c, &total, static_cast<uint64_t *>(S.out_off.p) + n, sizeof(total))) return PAFFY_E_HIP;
    c->plan.out_bytes = (int64_t)total;
    c->plan.n_rows = (int64_t)n;
    c->line_batches = static_cast<const uint8_t *const *>(S.batch_ptrs.p);
    c->line_meta = static_cast<const RecMeta *>(S.meta.p);
    c->line_order = static_cast<const uint32_t *>(S.order.p);
    c->line_level = static_cast<const int64_t *>(S.level.p);
    c->line_off = static_cast<const uint64_t *>(S.out_off.p);
    c->line_n = n;
    return 0;
}

/*
 * `paffy tile` (impl/paf_tile.c:156-178) over any number of text batches: begin, add every batch (the text stays where it is until
 * the output has been emitted), run. The visiting order (chain_score desc, score desc, input order) and the grouping by query name
 * are radix sorts on the device; the per-base walk keeps the counters of a 32 Ki-base slice in LDS (coverage_kernel.h).
 * Any failing record means nothing is written (the reference writes only after the last record).
 */
int paffy_hip_tile_begin(paffy_hip_ctx *c) {
    if (c) (void)paffy_hip_stream_trim(c); /* a closed stream's slot buffers (tens of GB) are this command's to use */
    if (!c) return PAFFY_E_ARG;
    c->planned = false;
    index_drop_all(c); /* indexes kept by paffy_hip_query_names for batches that were never split */
    return cov_begin(c, 1);
}
int paffy_hip_tile_add(paffy_hip_ctx *c, const void *d_in, int64_t in_len) {
    if (!c) return PAFFY_E_ARG;
    return cov_add(c, d_in, in_len, true);
}
int paffy_hip_tile_run(paffy_hip_ctx *c, paffy_plan_info *info) {
    if (!c || !info) return PAFFY_E_ARG;
    CovState &S = cov_state(c);
    c->planned = false;
    c->plan_is_tile = true;
    c->plan_is_bed = false;
    memset(info, 0, sizeof(*info));
    memset(&c->plan, 0, sizeof(c->plan));
    memset(&c->kp, 0, sizeof(c->kp));
    c->line_n = 0;
    for (const CovBatch &b : S.batches) c->plan.in_bytes += b.len;
    c->plan.n_records = (int64_t)S.n_rec;
    const uint64_t n = S.n_rec;
    if (n > 0) {
        int rc = cov_run(c, 0, &c->plan.error);
        if (rc) return rc;
        if (c->plan.error.code == 0) {
            int rl = lines_plan(c, S, n);
            if (rl) return rl;
        }
        if (c->profile) prof_collect(c);
    }
    *info = c->plan;
    c->planned = true;
    return 0;
}
/* the one-batch form */
int paffy_hip_tile_plan(paffy_hip_ctx *c, const void *d_in, int64_t in_len, paffy_plan_info *info) {
    if (!c || !info) return PAFFY_E_ARG;
    if (in_len < 0 || in_len >= (1ll << 31) - 64 || (in_len > 0 && !d_in) || (reinterpret_cast<uintptr_t>(d_in) & 15)) return PAFFY_E_ARG;
    int rc = paffy_hip_tile_begin(c);
    if (!rc) rc = paffy_hip_tile_add(c, d_in, in_len);
    if (!rc) rc = paffy_hip_tile_run(c, info);
    return rc;
}

/*
 * `paffy chain` (impl/paf_chain.c:123-127, impl/chaining.c:266-343) over any number of text batches, like tile: begin, add, run;
 * the output lines (cn / s1 tags set, cigar text verbatim) are written by paffy_hip_emit or paffy_hip_emit_lines.
 */
int paffy_hip_chain_begin(paffy_hip_ctx *c) {
    if (c) (void)paffy_hip_stream_trim(c); /* a closed stream's slot buffers (tens of GB) are this command's to use */
    if (!c) return PAFFY_E_ARG;
    c->planned = false;
    index_drop_all(c); /* indexes kept by paffy_hip_query_names for batches that were never split */
    return cov_begin(c, 1);
}
int paffy_hip_chain_add(paffy_hip_ctx *c, const void *d_in, int64_t in_len) {
    if (!c) return PAFFY_E_ARG;
    return cov_add(c, d_in, in_len, false);
}
int paffy_hip_chain_run(paffy_hip_ctx *c, const paffy_chain_opts *opts, paffy_plan_info *info) {
    if (!c || !info || !opts) return PAFFY_E_ARG;
    CovState &S = cov_state(c);
    c->planned = false;
    c->plan_is_tile = true; /* the output is a line table, as for tile */
    c->plan_is_bed = false;
    memset(info, 0, sizeof(*info));
    memset(&c->plan, 0, sizeof(c->plan));
    memset(&c->kp, 0, sizeof(c->kp));
    c->line_n = 0;
    for (const CovBatch &b : S.batches) c->plan.in_bytes += b.len;
    c->plan.n_records = (int64_t)S.n_rec;
    const uint64_t n = S.n_rec;
    if (n > 0) {
        const ChainOpts o{opts->gap_open, opts->gap_extend, opts->max_gap_length, opts->trim_fraction};
        int rc = chain_run(c, o, &c->plan.error);
        if (rc) return rc;
        if (c->plan.error.code == 0) {
            int rl = lines_plan(c, S, n);
            if (rl) return rl;
        }
        if (c->profile) prof_collect(c);
    }
    *info = c->plan;
    c->planned = true;
    return 0;
}
int64_t paffy_hip_chain_tags(paffy_hip_ctx *c, int64_t cap, int64_t *chain_id, int64_t *chain_score) {
    if (!c || cap < 0 || !chain_id || !chain_score) return PAFFY_E_ARG;
    if (!c->planned || !c->chain || c->plan.error.code) return PAFFY_E_STATE;
    const int64_t n = (int64_t)c->chain->n_out;
    if (cap < n) return PAFFY_E_CAPACITY;
    if (n > 0) {
        HIPCHK(c, hipMemcpy(chain_id, c->chain->tag_chain.p, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(chain_score, c->chain->tag_score.p, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost));
    }
    return n;
}

/*
 * Sharding `paffy tile` by query sequence (the reference pipeline's `split_file -q`, tests/paf_pipeline_test.sh:42):
 *   paffy_hip_query_names    the distinct query names of a batch as 64-bit hashes with the bytes of their lines (the weights a
 *                            partitioner balances)
 *   paffy_hip_split_by_owner the lines of a batch regrouped by the owner of their query name (owner table: hash -> part), input order
 *                            kept inside a part: what an all-to-all over the ranks sends
 *   paffy_hip_scatter_lines  lines to given offsets of an output (the ordered write once every line's place is known)
 */
__global__ __launch_bounds__(PAFFY_NT) void k_query_hash(const uint8_t *in, const RecMeta *meta, const uint32_t *sep_pos, const uint32_t *nl_idx, uint32_t n, uint32_t in_len,
                                                          uint64_t *hash, uint64_t *line_len, uint32_t *idx) {
    const uint32_t r = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (r >= n) return;
    const RecMeta &m = meta[r];
    hash[r] = cov_name_hash(in, m.qname_off, m.qname_len);
    const uint32_t start = r == 0 ? 0u : sep_pos[nl_idx[r - 1]] + 1u, end = sep_pos[nl_idx[r]];
    (void)in_len;
    line_len[r] = (uint64_t)(end - start) + 1u; /* with its newline (a last line without one gets one) */
    idx[r] = r;
}
/* sorted by hash: line bytes in sorted order (for a scan), and at the head of every run of equal hashes its hash */
__global__ __launch_bounds__(PAFFY_NT) void k_run_lens(const uint32_t *sorted_idx, const uint64_t *line_len, uint32_t n, uint64_t *out) {
    const uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i < n) out[i] = line_len[sorted_idx[i]];
    if (i == 0) out[n] = 0;
}
/* weight of run c = (bytes in front of the next run) - (bytes in front of this one): no atomics on a handful of addresses */
__global__ __launch_bounds__(PAFFY_NT) void k_run_heads_out(const uint64_t *sorted_hash, const uint32_t *flag, const uint32_t *scan, const uint64_t *byte_off, uint32_t n,
                                                             uint64_t *out_hash, uint64_t *out_start, uint64_t *out_first) {
    const uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) {
        out_hash[scan[i] - 1u] = sorted_hash[i];
        out_start[scan[i] - 1u] = byte_off[i];
        out_first[scan[i] - 1u] = i; /* lines in front of the run: run lengths are differences */
    }
    if (i == n - 1) {
        out_start[scan[i]] = byte_off[n];
        out_first[scan[i]] = n;
    }
}
__global__ __launch_bounds__(PAFFY_NT) void k_owner_keys(const uint64_t *hash, uint32_t n, const uint64_t *tab_hash, const uint32_t *tab_owner, uint32_t n_tab, uint32_t n_parts,
                                                          uint64_t *key) {
    const uint32_t r = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (r >= n) return;
    uint32_t lo = 0, hi = n_tab; /* first entry >= hash */
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tab_hash[mid] < hash[r]) lo = mid + 1;
        else hi = mid;
    }
    uint32_t owner = (lo < n_tab && tab_hash[lo] == hash[r]) ? tab_owner[lo] : (uint32_t)(hash[r] % n_parts); /* a name the table does not know */
    if (owner >= n_parts) owner = n_parts - 1;
    key[r] = ((uint64_t)owner << 32) | r;
}
__global__ __launch_bounds__(PAFFY_NT) void k_gather_len(const uint64_t *sorted_key, const uint64_t *line_len, uint32_t n, uint64_t *out) {
    const uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i < n) out[i] = line_len[(uint32_t)sorted_key[i]];
    if (i == 0) out[n] = 0;
}
/* where every part starts in the sorted order: first[p] = index of its first line, start[p] = its first byte (parts without lines keep -1) */
__global__ __launch_bounds__(PAFFY_NT) void k_part_bounds(const uint64_t *sorted_key, const uint64_t *off, uint32_t n, int64_t *first, int64_t *start) {
    const uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = (uint32_t)(sorted_key[i] >> 32);
    if (i == 0 || (uint32_t)(sorted_key[i - 1] >> 32) != p) {
        first[p] = i;
        start[p] = (int64_t)off[i];
    }
}
/* one workgroup per line: bytes [src_off, src_off + len) of src to dst + dst_off; a source line that ends without newline gets one */
__device__ __forceinline__ void copy_line(const uint8_t *src, uint64_t len, uint8_t *dst, bool add_nl) {
    const uint64_t body = add_nl ? len - 1 : len;
    const uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
    if (head >= body) {
        for (uint64_t x = threadIdx.x; x < body; x += PAFFY_NT) dst[x] = src[x];
    } else {
        for (uint32_t x = threadIdx.x; x < head; x += PAFFY_NT) dst[x] = src[x];
        const uint64_t n_ch = (body - head) >> 4;
        for (uint64_t ch = threadIdx.x; ch < n_ch; ch += PAFFY_NT)
            *reinterpret_cast<u32x4 *>(dst + head + 16 * ch) = *reinterpret_cast<const u32x4_unaligned *>(src + head + 16 * ch);
        for (uint64_t x = head + 16 * n_ch + threadIdx.x; x < body; x += PAFFY_NT) dst[x] = src[x];
    }
    if (add_nl && threadIdx.x == 0) dst[body] = '\n';
}
/* line blockIdx.x of the (part, input index) order. Without part_dst the parts follow each other in `out` and in rec_index; with it
   part p's lines start at out + part_dst[p] and its record indices at rec_index + part_dst[n_parts + p] (the caller's send buffer, in
   which the parts of several batches stand behind each other per destination). rec_base is added to every index. */
__global__ __launch_bounds__(PAFFY_NT) void k_split_copy(const uint8_t *in, uint32_t in_len, const uint32_t *sep_pos, const uint32_t *nl_idx, const uint64_t *sorted_key,
                                                          const uint64_t *off, uint8_t *out, const int64_t *part_first, const int64_t *part_start, const int64_t *part_dst,
                                                          uint32_t n_parts, int64_t *rec_index, int64_t rec_base) {
    const uint64_t key = sorted_key[blockIdx.x];
    const uint32_t r = (uint32_t)key, p = (uint32_t)(key >> 32);
    const uint32_t start = r == 0 ? 0u : sep_pos[nl_idx[r - 1]] + 1u, end = sep_pos[nl_idx[r]];
    uint64_t at = off[blockIdx.x];
    int64_t slot = (int64_t)blockIdx.x;
    if (part_dst) {
        at = (uint64_t)part_dst[p] + (at - (uint64_t)part_start[p]);
        slot = part_dst[n_parts + p] + ((int64_t)blockIdx.x - part_first[p]);
    }
    if (rec_index && threadIdx.x == 0) rec_index[slot] = (int64_t)r + rec_base;
    copy_line(in + start, (uint64_t)(end - start) + 1u, out + at, end >= in_len);
}
__global__ __launch_bounds__(PAFFY_NT) void k_scatter_lines(const uint8_t *src, const int64_t *src_off, const int64_t *dst_off, uint8_t *dst) {
    const uint64_t k = blockIdx.x;
    copy_line(src + src_off[k], (uint64_t)(src_off[k + 1] - src_off[k]), dst + dst_off[k], false);
}

/* the kept indexes (see paffy_hip_ctx::kept_index): copy out after query_names, copy back for split_by_owner */
static void index_drop(paffy_hip_ctx *c, const void *d_in) {
    for (size_t i = 0; i < c->kept_index.size(); i++)
        if (c->kept_index[i].in == d_in) {
            c->index_pool.push_back(c->kept_index[i]); /* the buffers stay allocated for the next batch */
            c->kept_index.erase(c->kept_index.begin() + (long)i);
            return;
        }
}
static int index_keep(paffy_hip_ctx *c, const void *d_in, int64_t in_len, uint32_t n) {
    index_drop(c, d_in);
    if (c->kept_index.size() >= 64) return 0; /* a caller that never splits: stop keeping */
    paffy_hip_ctx::KeptIndex k;
    if (!c->index_pool.empty()) {
        k = c->index_pool.back();
        c->index_pool.pop_back();
    }
    k.in = d_in;
    k.len = in_len;
    k.n = n;
    k.n_seps = c->h_info->n_seps;
    const size_t mb = sizeof(RecMeta) * (size_t)n, sb = sizeof(uint32_t) * ((size_t)k.n_seps + 1), nb = sizeof(uint32_t) * ((size_t)n + 1);
    if (ensure(c, k.meta, mb) || ensure(c, k.sep_pos, sb) || ensure(c, k.nl_idx, nb)) return PAFFY_E_HIP;
    HIPCHK(c, hipMemcpyAsync(k.meta.p, c->meta.p, mb, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(k.sep_pos.p, c->sep_pos.p, sb, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(k.nl_idx.p, c->nl_idx.p, nb, hipMemcpyDeviceToDevice, c->stream));
    c->kept_index.push_back(k);
    return 0;
}
/* 0: the index buffers describe d_in again; 1: nothing kept for it */
static int index_restore(paffy_hip_ctx *c, const void *d_in, int64_t in_len, uint32_t *n) {
    for (size_t i = 0; i < c->kept_index.size(); i++) {
        paffy_hip_ctx::KeptIndex &k = c->kept_index[i];
        if (k.in != d_in || k.len != in_len) continue;
        const size_t mb = sizeof(RecMeta) * (size_t)k.n, sb = sizeof(uint32_t) * ((size_t)k.n_seps + 1), nb = sizeof(uint32_t) * ((size_t)k.n + 1);
        if (ensure(c, c->meta, mb) || ensure(c, c->sep_pos, sb) || ensure(c, c->nl_idx, nb)) return 1;
        if (hipMemcpyAsync(c->meta.p, k.meta.p, mb, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(c->sep_pos.p, k.sep_pos.p, sb, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(c->nl_idx.p, k.nl_idx.p, nb, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) /* the kept copy is freed below */
            return 1;
        *n = k.n;
        index_drop(c, d_in);
        return 0;
    }
    return 1;
}

static void index_drop_all(paffy_hip_ctx *c) {
    while (!c->kept_index.empty()) index_drop(c, c->kept_index.back().in);
    c->indexed_in = nullptr;
}
static int64_t query_names_impl(paffy_hip_ctx *c, const void *d_in, int64_t in_len, int64_t cap, uint64_t *hashes, int64_t *weights, int64_t *records) {
    if (!c || !hashes || !weights || cap < 0) return PAFFY_E_ARG;
    if (in_len < 0 || in_len >= (1ll << 31) - 64 || (in_len > 0 && !d_in) || (reinterpret_cast<uintptr_t>(d_in) & 15)) return PAFFY_E_ARG;
    if (in_len == 0) return 0;
    CovState &S = cov_state(c);
    uint32_t n = 0;
    const uint8_t *in = static_cast<const uint8_t *>(d_in);
    c->indexed_in = nullptr;
    int rc = index_and_parse(c, in, (uint32_t)in_len, &n);
    if (rc) return rc;
    if (n == 0) return 0;
    c->indexed_in = d_in;
    c->indexed_len = in_len;
    c->indexed_n = n;
    if (index_keep(c, d_in, in_len, n)) return PAFFY_E_HIP;
    const uint32_t g = (n + PAFFY_NT - 1) / PAFFY_NT;
    if (ensure(c, S.k64a, sizeof(uint64_t) * ((size_t)n + 1)) || ensure(c, S.k64b, sizeof(uint64_t) * ((size_t)n + 1)) || ensure(c, S.name_hash, sizeof(uint64_t) * ((size_t)n + 1)) ||
        ensure(c, S.v32a, sizeof(uint32_t) * ((size_t)n + 1)) || ensure(c, S.v32b, sizeof(uint32_t) * ((size_t)n + 1)) || ensure(c, S.flags, sizeof(uint32_t) * ((size_t)n + 1)) ||
        ensure(c, S.scan32, sizeof(uint32_t) * ((size_t)n + 1)))
        return PAFFY_E_HIP;
    uint64_t *hash = static_cast<uint64_t *>(S.name_hash.p), *len = static_cast<uint64_t *>(S.k64b.p), *sorted = static_cast<uint64_t *>(S.k64a.p);
    uint32_t *idx = static_cast<uint32_t *>(S.v32a.p), *sidx = static_cast<uint32_t *>(S.v32b.p), *flags = static_cast<uint32_t *>(S.flags.p), *scan = static_cast<uint32_t *>(S.scan32.p);
    LAUNCH(c, "k_query_hash", k_query_hash, dim3(g), dim3(PAFFY_NT), 0, in, static_cast<const RecMeta *>(c->meta.p), static_cast<const uint32_t *>(c->sep_pos.p),
           static_cast<const uint32_t *>(c->nl_idx.p), n, (uint32_t)in_len, hash, len, idx);
    if (cov_sort_pairs(c, S, hash, sorted, idx, sidx, n)) return PAFFY_E_HIP;
    LAUNCH(c, "k_cov_run_heads", k_cov_run_heads, dim3(g), dim3(PAFFY_NT), 0, static_cast<const uint64_t *>(sorted), n, flags);
    if (cov_incl_scan32(c, S, flags, scan, n)) return PAFFY_E_HIP;
    uint32_t n_names = 0;
    if (cov_fetch(c, &n_names, scan + (n - 1), sizeof(uint32_t))) return PAFFY_E_HIP;
    if ((int64_t)n_names > cap) return PAFFY_E_CAPACITY;
    if (ensure(c, S.pairs, sizeof(uint64_t) * (size_t)n_names) || ensure(c, S.pairs2, sizeof(uint64_t) * 2 * ((size_t)n_names + 1)) || ensure(c, S.bm_words, sizeof(uint64_t) * ((size_t)n + 1)) ||
        ensure(c, S.bm_off, sizeof(uint64_t) * ((size_t)n + 2)))
        return PAFFY_E_HIP;
    uint64_t *slen = static_cast<uint64_t *>(S.bm_words.p), *soff = static_cast<uint64_t *>(S.bm_off.p);
    LAUNCH(c, "k_run_lens", k_run_lens, dim3(g), dim3(PAFFY_NT), 0, static_cast<const uint32_t *>(sidx), static_cast<const uint64_t *>(len), n, slen);
    if (cov_excl_scan64(c, S, slen, soff, n)) return PAFFY_E_HIP;
    LAUNCH(c, "k_run_heads_out", k_run_heads_out, dim3(g), dim3(PAFFY_NT), 0, static_cast<const uint64_t *>(sorted), static_cast<const uint32_t *>(flags), static_cast<const uint32_t *>(scan),
           static_cast<const uint64_t *>(soff), n, static_cast<uint64_t *>(S.pairs.p), static_cast<uint64_t *>(S.pairs2.p), static_cast<uint64_t *>(S.pairs2.p) + n_names + 1);
    std::vector<uint64_t> starts(2 * ((size_t)n_names + 1));
    HIPCHK(c, hipMemcpyAsync(hashes, S.pairs.p, sizeof(uint64_t) * (size_t)n_names, hipMemcpyDeviceToHost, c->stream));
    if (cov_fetch(c, starts.data(), S.pairs2.p, sizeof(uint64_t) * 2 * ((size_t)n_names + 1))) return PAFFY_E_HIP;
    for (uint32_t k = 0; k < n_names; k++) weights[k] = (int64_t)(starts[k + 1] - starts[k]);
    if (records)
        for (uint32_t k = 0; k < n_names; k++) records[k] = (int64_t)(starts[(size_t)n_names + 1 + k + 1] - starts[(size_t)n_names + 1 + k]);
    return (int64_t)n_names;
}

static int split_impl0(paffy_hip_ctx *c, const void *d_in, int64_t in_len, int32_t n_parts, const uint64_t *table_hash, const uint32_t *table_owner, int64_t n_table,
                       void *d_out, int64_t out_cap, const int64_t *part_dst, const int64_t *rec_dst, int64_t rec_base, int64_t *part_bytes, int64_t *part_records,
                       void *d_rec_index, int64_t rec_index_cap, int64_t *n_records) {
    if (!c || n_parts < 1 || n_table < 0 || (n_table > 0 && (!table_hash || !table_owner)) || !part_bytes || !part_records || !n_records) return PAFFY_E_ARG;
    if (in_len < 0 || in_len >= (1ll << 31) - 64 || (in_len > 0 && !d_in) || (reinterpret_cast<uintptr_t>(d_in) & 15)) return PAFFY_E_ARG;
    if ((part_dst != nullptr) != (rec_dst != nullptr)) return PAFFY_E_ARG;
    for (int32_t p = 0; p < n_parts; p++) part_bytes[p] = part_records[p] = 0;
    *n_records = 0;
    if (in_len == 0) return 0;
    if (!d_out || (!part_dst && out_cap < in_len + 1)) return PAFFY_E_CAPACITY;
    CovState &S = cov_state(c);
    uint32_t n = 0;
    const uint8_t *in = static_cast<const uint8_t *>(d_in);
    if (c->indexed_in == d_in && c->indexed_len == in_len) { /* indexed by paffy_hip_query_names just before */
        n = c->indexed_n;
        index_drop(c, d_in);
    } else if (index_restore(c, d_in, in_len, &n) != 0) { /* ... or some batches ago: its index comes back from the kept copy */
        int rc = index_and_parse(c, in, (uint32_t)in_len, &n);
        if (rc) return rc;
    }
    c->indexed_in = nullptr;
    *n_records = n;
    if (n == 0) return 0;
    if (d_rec_index && !part_dst && rec_index_cap < (int64_t)n) return PAFFY_E_CAPACITY;
    const uint32_t g = (n + PAFFY_NT - 1) / PAFFY_NT;
    if (ensure(c, S.k64a, sizeof(uint64_t) * ((size_t)n + 1)) || ensure(c, S.k64b, sizeof(uint64_t) * ((size_t)n + 1)) || ensure(c, S.name_hash, sizeof(uint64_t) * ((size_t)n + 1)) ||
        ensure(c, S.v32a, sizeof(uint32_t) * ((size_t)n + 1)) || ensure(c, S.bm_words, sizeof(uint64_t) * ((size_t)n + 1)) || ensure(c, S.bm_off, sizeof(uint64_t) * ((size_t)n + 2)) ||
        ensure(c, S.pairs, (sizeof(uint64_t) + sizeof(uint32_t)) * ((size_t)n_table + 1)) || ensure(c, S.pairs2, sizeof(int64_t) * 4 * (size_t)n_parts))
        return PAFFY_E_HIP;
    uint64_t *hash = static_cast<uint64_t *>(S.name_hash.p), *len = static_cast<uint64_t *>(S.k64b.p), *key = static_cast<uint64_t *>(S.k64a.p), *skey = static_cast<uint64_t *>(S.bm_words.p);
    uint64_t *off = static_cast<uint64_t *>(S.bm_off.p);
    uint64_t *d_th = static_cast<uint64_t *>(S.pairs.p);
    uint32_t *d_to = reinterpret_cast<uint32_t *>(d_th + n_table + 1);
    if (n_table) {
        HIPCHK(c, hipMemcpyAsync(d_th, table_hash, sizeof(uint64_t) * (size_t)n_table, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_to, table_owner, sizeof(uint32_t) * (size_t)n_table, hipMemcpyHostToDevice, c->stream));
    }
    LAUNCH(c, "k_query_hash", k_query_hash, dim3(g), dim3(PAFFY_NT), 0, in, static_cast<const RecMeta *>(c->meta.p), static_cast<const uint32_t *>(c->sep_pos.p),
           static_cast<const uint32_t *>(c->nl_idx.p), n, (uint32_t)in_len, hash, len, static_cast<uint32_t *>(S.v32a.p));
    LAUNCH(c, "k_owner_keys", k_owner_keys, dim3(g), dim3(PAFFY_NT), 0, static_cast<const uint64_t *>(hash), n, static_cast<const uint64_t *>(d_th), static_cast<const uint32_t *>(d_to),
           (uint32_t)n_table, (uint32_t)n_parts, key);
    if (cov_sort_keys(c, S, key, skey, n)) return PAFFY_E_HIP; /* (owner, input index): input order inside a part */
    LAUNCH(c, "k_gather_len", k_gather_len, dim3(g), dim3(PAFFY_NT), 0, static_cast<const uint64_t *>(skey), static_cast<const uint64_t *>(len), n, key);
    if (cov_excl_scan64(c, S, key, off, n)) return PAFFY_E_HIP;
    int64_t *d_tot = static_cast<int64_t *>(S.pairs2.p);
    HIPCHK(c, hipMemsetAsync(d_tot, 0xff, sizeof(int64_t) * 2 * (size_t)n_parts, c->stream));
    LAUNCH(c, "k_part_bounds", k_part_bounds, dim3(g), dim3(PAFFY_NT), 0, static_cast<const uint64_t *>(skey), static_cast<const uint64_t *>(off), n, d_tot, d_tot + n_parts);
    /* the sizes of the parts first: a caller's destinations are checked against them before a byte is written */
    std::vector<int64_t> tot(2 * (size_t)n_parts);
    uint64_t all_bytes = 0;
    HIPCHK(c, hipMemcpyAsync(&all_bytes, off + n, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    if (cov_fetch(c, tot.data(), d_tot, sizeof(int64_t) * 2 * (size_t)n_parts)) return PAFFY_E_HIP;
    int64_t next_first = (int64_t)n, next_start = (int64_t)all_bytes; /* a part ends where the next non-empty one starts */
    for (int32_t p = n_parts - 1; p >= 0; p--) {
        const int64_t first = tot[(size_t)p], start = tot[(size_t)n_parts + p];
        if (first < 0) continue;
        part_records[p] = next_first - first;
        part_bytes[p] = next_start - start;
        next_first = first;
        next_start = start;
    }
    const int64_t *d_dst = nullptr;
    if (part_dst) {
        std::vector<int64_t> dst(2 * (size_t)n_parts);
        for (int32_t p = 0; p < n_parts; p++) {
            if (part_dst[p] < 0 || rec_dst[p] < 0 || part_dst[p] + part_bytes[p] > out_cap || (d_rec_index && rec_dst[p] + part_records[p] > rec_index_cap)) {
                for (int32_t q = 0; q < n_parts; q++) part_bytes[q] = part_records[q] = 0;
                return PAFFY_E_CAPACITY;
            }
            dst[(size_t)p] = part_dst[p];
            dst[(size_t)n_parts + p] = rec_dst[p];
        }
        HIPCHK(c, hipMemcpyAsync(d_tot + 2 * n_parts, dst.data(), sizeof(int64_t) * 2 * (size_t)n_parts, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream)); /* `dst` is a local */
        d_dst = d_tot + 2 * n_parts;
    }
    LAUNCH(c, "k_split_copy", k_split_copy, dim3(n), dim3(PAFFY_NT), 0, in, (uint32_t)in_len, static_cast<const uint32_t *>(c->sep_pos.p), static_cast<const uint32_t *>(c->nl_idx.p),
           static_cast<const uint64_t *>(skey), static_cast<const uint64_t *>(off), static_cast<uint8_t *>(d_out), static_cast<const int64_t *>(d_tot),
           static_cast<const int64_t *>(d_tot + n_parts), d_dst, (uint32_t)n_parts, static_cast<int64_t *>(d_rec_index), rec_base);
    return 0;
}

/* A failing call leaves no kept index behind: whatever the caller does next starts from the text. */
int64_t paffy_hip_query_names(paffy_hip_ctx *c, const void *d_in, int64_t in_len, int64_t cap, uint64_t *hashes, int64_t *weights) {
    const int64_t n = query_names_impl(c, d_in, in_len, cap, hashes, weights, nullptr);
    if (n < 0 && c) index_drop_all(c);
    return n;
}
int64_t paffy_hip_query_names_counts(paffy_hip_ctx *c, const void *d_in, int64_t in_len, int64_t cap, uint64_t *hashes, int64_t *weights, int64_t *records) {
    if (!records) return PAFFY_E_ARG;
    const int64_t n = query_names_impl(c, d_in, in_len, cap, hashes, weights, records);
    if (n < 0 && c) index_drop_all(c);
    return n;
}
static int split_impl(paffy_hip_ctx *c, const void *d_in, int64_t in_len, int32_t n_parts, const uint64_t *table_hash, const uint32_t *table_owner, int64_t n_table,
                      void *d_out, int64_t out_cap, const int64_t *part_dst, const int64_t *rec_dst, int64_t rec_base, int64_t *part_bytes, int64_t *part_records,
                      void *d_rec_index, int64_t rec_index_cap, int64_t *n_records) {
    const int rc = split_impl0(c, d_in, in_len, n_parts, table_hash, table_owner, n_table, d_out, out_cap, part_dst, rec_dst, rec_base, part_bytes, part_records, d_rec_index,
                               rec_index_cap, n_records);
    if (rc && c) index_drop_all(c);
    return rc;
}
int paffy_hip_split_by_owner(paffy_hip_ctx *c, const void *d_in, int64_t in_len, int32_t n_parts, const uint64_t *table_hash, const uint32_t *table_owner, int64_t n_table,
                             void *d_out, int64_t out_cap, int64_t *part_bytes, int64_t *part_records, void *d_rec_index, int64_t rec_index_cap, int64_t *n_records) {
    return split_impl(c, d_in, in_len, n_parts, table_hash, table_owner, n_table, d_out, out_cap, nullptr, nullptr, 0, part_bytes, part_records, d_rec_index, rec_index_cap,
                      n_records);
}
int paffy_hip_split_to(paffy_hip_ctx *c, const void *d_in, int64_t in_len, int32_t n_parts, const uint64_t *table_hash, const uint32_t *table_owner, int64_t n_table,
                       void *d_out, int64_t out_cap, const int64_t *part_dst, const int64_t *rec_dst, int64_t rec_base, int64_t *part_bytes, int64_t *part_records,
                       void *d_rec_index, int64_t rec_index_cap, int64_t *n_records) {
    if (!part_dst || !rec_dst) return PAFFY_E_ARG;
    return split_impl(c, d_in, in_len, n_parts, table_hash, table_owner, n_table, d_out, out_cap, part_dst, rec_dst, rec_base, part_bytes, part_records, d_rec_index,
                      rec_index_cap, n_records);
}
int paffy_hip_drop_index(paffy_hip_ctx *c, const void *d_in) {
    if (!c) return PAFFY_E_ARG;
    if (d_in) {
        index_drop(c, d_in);
        if (c->indexed_in == d_in) c->indexed_in = nullptr;
    } else {
        index_drop_all(c);
    }
    return 0;
}

int paffy_hip_scatter_lines(paffy_hip_ctx *c, const void *d_src, const void *d_src_off, const void *d_dst_off, int64_t n_lines, void *d_dst) {
    if (!c || n_lines < 0 || n_lines >= (1ll << 31)) return PAFFY_E_ARG;
    if (n_lines == 0) return 0;
    if (!d_src || !d_src_off || !d_dst_off || !d_dst) return PAFFY_E_ARG;
    LAUNCH(c, "k_scatter_lines", k_scatter_lines, dim3((unsigned)n_lines), dim3(PAFFY_NT), 0, static_cast<const uint8_t *>(d_src), static_cast<const int64_t *>(d_src_off),
           static_cast<const int64_t *>(d_dst_off), static_cast<uint8_t *>(d_dst));
    return 0;
}

/* After a tile run: (chain_score, score, input record, line bytes, tile level) of the lines emit writes, in output order -- what ranks
 * exchange to merge their shares of a `paffy tile` sharded by query sequence. Returns the line count. */
__global__ __launch_bounds__(PAFFY_NT) void k_tile_keys_out(const RecMeta *meta, const uint32_t *order, const int64_t *level, const uint64_t *out_len, uint64_t n,
                                                             int64_t *keys) {
    const uint64_t k = (uint64_t)blockIdx.x * PAFFY_NT + threadIdx.x;
    if (k >= n) return;
    const RecMeta &m = meta[order[k]];
    keys[5 * k + 0] = m.chain_score;
    keys[5 * k + 1] = m.score;
    keys[5 * k + 2] = order[k];
    keys[5 * k + 3] = (int64_t)out_len[k];
    keys[5 * k + 4] = level[order[k]];
}
int64_t paffy_hip_tile_keys(paffy_hip_ctx *c, int64_t cap_lines, void *d_keys) {
    if (!c || !d_keys) return PAFFY_E_ARG;
    if (!c->planned || !c->plan_is_tile || !c->cov) return PAFFY_E_STATE;
    CovState &S = cov_state(c);
    const uint64_t n = c->line_n;
    if (c->line_meta != static_cast<const RecMeta *>(S.meta.p)) return PAFFY_E_STATE; /* a dedupe plan */
    if ((uint64_t)cap_lines < n) return PAFFY_E_CAPACITY;
    if (n) LAUNCH(c, "k_tile_keys_out", k_tile_keys_out, dim3((unsigned)((n + PAFFY_NT - 1) / PAFFY_NT)), dim3(PAFFY_NT), 0, c->line_meta, c->line_order, c->line_level,
                  static_cast<const uint64_t *>(S.out_len.p), n, static_cast<int64_t *>(d_keys));
    return (int64_t)n;
}

int paffy_hip_dedupe_reset(paffy_hip_ctx *c) {
    if (!c) return PAFFY_E_ARG;
    if (c->dedupe) c->dedupe->seen_n = 0;
    return 0;
}

int paffy_hip_dedupe_plan(paffy_hip_ctx *c, const void *d_in, int64_t in_len, int check_inverse, paffy_plan_info *info) {
    if (!c || !info) return PAFFY_E_ARG;
    if (in_len < 0 || in_len >= (1ll << 31) - 64 || (in_len > 0 && !d_in) || (reinterpret_cast<uintptr_t>(d_in) & 15)) return PAFFY_E_ARG;
    c->planned = false;
    c->plan_is_tile = true; /* same writer as tile: header + the cigar text as it was read */
    memset(info, 0, sizeof(*info));
    memset(&c->plan, 0, sizeof(c->plan));
    info->in_bytes = c->plan.in_bytes = in_len;
    memset(&c->kp, 0, sizeof(c->kp));
    c->tile_n = 0;
    if (in_len == 0) {
        c->planned = true;
        return 0;
    }
    const uint8_t *in = static_cast<const uint8_t *>(d_in);
    uint32_t n = 0;
    {
        int rc = index_and_parse(c, in, (uint32_t)in_len, &n);
        if (rc) return rc;
    }
    c->plan.n_records = n;
    if (n == 0) {
        *info = c->plan;
        c->planned = true;
        return 0;
    }
    const uint32_t grid = (n + PAFFY_NT - 1) / PAFFY_NT;
    if (ensure(c, c->dedupe_keys, sizeof(DedupeKey) * (size_t)n)) return PAFFY_E_HIP;
    if (ensure(c, c->tile_level, sizeof(int64_t) * (size_t)n)) return PAFFY_E_HIP;
    if (ensure(c, c->tile_order, sizeof(uint32_t) * (size_t)n)) return PAFFY_E_HIP;
    if (ensure(c, c->tile_len, sizeof(int64_t) * (size_t)(n + 2))) return PAFFY_E_HIP;
    LAUNCH(c, "k_dedupe_keys", k_dedupe_keys, dim3(grid), dim3(PAFFY_NT), 0, in, static_cast<const RecMeta *>(c->meta.p), n,
           static_cast<DedupeKey *>(c->dedupe_keys.p), static_cast<int64_t *>(c->tile_level.p));
    /* the reference's loop, record by record (impl/paf_dedupe.c:117-143): first seen wins -- on the device, dedupe_host.h */
    if (!c->dedupe) c->dedupe = new DedupeState();
    uint32_t nk = 0, first_bad = n;
    {
        int rc = dedupe_select(c, *c->dedupe, static_cast<const DedupeKey *>(c->dedupe_keys.p), n, check_inverse, static_cast<uint32_t *>(c->tile_order.p), &nk, &first_bad);
        if (rc) return rc;
    }
    if (first_bad < n) { /* the record that ends the run: a parse error, or paf_check (impl/paf_dedupe.c:126) */
        DedupeKey k;
        HIPCHK(c, hipMemcpy(&k, static_cast<const DedupeKey *>(c->dedupe_keys.p) + first_bad, sizeof(k), hipMemcpyDeviceToHost));
        c->plan.error.record = first_bad;
        if (k.err) {
            c->plan.error.code = k.err;
            c->plan.error.stage = -1;
            RecMeta m;
            HIPCHK(c, hipMemcpy(&m, static_cast<RecMeta *>(c->meta.p) + first_bad, sizeof(m), hipMemcpyDeviceToHost));
            c->plan.error.aux = m.err_aux;
        } else {
            c->plan.error.code = k.check;
            c->plan.error.stage = 0;
        }
    }
    int64_t total = 0;
    if (nk > 0) {
        int64_t *lens = static_cast<int64_t *>(c->tile_len.p);
        LAUNCH(c, "k_line_size", k_line_size, dim3((nk + 1 + PAFFY_NT - 1) / PAFFY_NT), dim3(PAFFY_NT), 0, static_cast<const RecMeta *>(c->meta.p),
               static_cast<const uint32_t *>(c->tile_order.p), static_cast<const int64_t *>(c->tile_level.p), (uint64_t)nk, reinterpret_cast<uint64_t *>(lens));
        { /* offsets of the lines: exclusive scan of nk + 1 lengths (the last one zero, k_line_size), so that entry nk is the total */
            if (ensure(c, c->out_off, sizeof(int64_t) * ((size_t)nk + 1))) return PAFFY_E_HIP;
            DedupeState &D = *c->dedupe;
            size_t bytes = 0;
            RPCHK(c, rocprim::exclusive_scan(nullptr, bytes, lens, static_cast<int64_t *>(c->out_off.p), (int64_t)0, (size_t)nk + 1, rocprim::plus<int64_t>(), c->stream));
            if (ensure(c, D.tmp, bytes + 16)) return PAFFY_E_HIP;
            RPCHK(c, rocprim::exclusive_scan(D.tmp.p, bytes, lens, static_cast<int64_t *>(c->out_off.p), (int64_t)0, (size_t)nk + 1, rocprim::plus<int64_t>(), c->stream));
        }
        HIPCHK(c, hipMemcpyAsync(&total, static_cast<int64_t *>(c->out_off.p) + nk, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        if (ensure(c, c->one_batch, sizeof(void *))) return PAFFY_E_HIP;
        HIPCHK(c, hipMemcpyAsync(c->one_batch.p, &in, sizeof(void *), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (c->profile) prof_collect(c);
    c->plan.out_bytes = total;
    c->plan.n_rows = nk;
    c->tile_n = total ? nk : 0;
    c->line_batches = static_cast<const uint8_t *const *>(c->one_batch.p);
    c->line_meta = static_cast<const RecMeta *>(c->meta.p);
    c->line_order = static_cast<const uint32_t *>(c->tile_order.p);
    c->line_level = static_cast<const int64_t *>(c->tile_level.p);
    c->line_off = static_cast<const uint64_t *>(c->out_off.p);
    c->line_n = c->tile_n;
    *info = c->plan;
    c->planned = true;
    return 0;
}

int64_t paffy_hip_plan_rows(paffy_hip_ctx *c, int64_t cap, uint32_t *record, int64_t *out_off) {
    if (!c || !record || !out_off) return PAFFY_E_ARG;
    if (!c->planned || !c->plan_is_tile) return PAFFY_E_STATE;
    const int64_t n = (int64_t)c->line_n;
    if (cap < n + 1) return PAFFY_E_CAPACITY;
    if (n > 0) {
        HIPCHK(c, hipMemcpyAsync(record, c->line_order, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(out_off, c->line_off, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    out_off[n] = c->plan.out_bytes;
    return n;
}

int paffy_hip_emit(paffy_hip_ctx *c, void *d_out, int64_t out_cap) {
    if (!c) return PAFFY_E_ARG;
    if (!c->planned) return PAFFY_E_STATE;
    if (c->plan.out_bytes == 0 || (!c->plan_is_tile && !c->plan_is_bed && c->kp.n_rec == 0)) return 0;
    if (!d_out || (reinterpret_cast<uintptr_t>(d_out) & 15)) return PAFFY_E_ARG;
    if (out_cap < c->plan.out_bytes) return PAFFY_E_CAPACITY;
    if (c->plan_is_bed) {
        const uint64_t n_runs = c->bed_runs;
        LAUNCH(c, "k_bed_lines", k_bed_lines, dim3((unsigned)((n_runs + PAFFY_NT - 1) / PAFFY_NT)), dim3(PAFFY_NT), 0, *c->bed_params,
               static_cast<const uint64_t *>(c->bed_starts.p), n_runs, static_cast<int64_t *>(c->bed_len.p), static_cast<const int64_t *>(c->bed_off.p),
               static_cast<uint8_t *>(d_out));
        return 0;
    }
    if (c->plan_is_tile) {
        if (c->line_n) LAUNCH(c, "k_tile_emit", k_line_emit, dim3((unsigned)((c->line_n + PAFFY_NWAVE - 1) / PAFFY_NWAVE)), dim3(PAFFY_NT), 0, c->line_batches, c->line_meta,
                              c->line_order, c->line_level, c->line_off, (uint64_t)0, c->line_n, (uint64_t)0, static_cast<uint8_t *>(d_out));
        return 0;
    }
    KParams kp = c->kp;
    kp.out = static_cast<uint8_t *>(d_out);
    const bool shatter = kp.n_stages > 0 && kp.stages[kp.n_stages - 1].kind == PAFFY_SHATTER;
    if (shatter) {
        LAUNCH(c, "k_emit_rows", k_emit_rows, dim3(kp.n_rec + kp.n_items), dim3(64), PAFFY_ROWS_LDS_BYTES + dbg_lds_pad(0), kp);
        if (c->h_info->g_count > 0) LAUNCH(c, "k_emit_lds", k_emit_lds<true>, dim3(kp.n_rec), dim3(PAFFY_NT), PAFFY_EMIT_LDS_BYTES, kp);
        if (c->h_info->w_count > 0) LAUNCH(c, "k_arena_emit", k_arena_emit<true>, dim3(2048), dim3(PAFFY_NT), PAFFY_EMIT_LDS_BYTES, kp);
    } else {
        LAUNCH(c, "k_emit_line", k_emit_line, dim3(kp.n_rec + kp.n_items), dim3(64), PAFFY_LINE_LDS_BYTES, kp);
        /* the lines of the flat pass whose cigar is a stretch of the input's (a workgroup without such a record ends at once) */
        if (kp.flat_done && !kp.new_ops) LAUNCH(c, "k_emit_copy", k_emit_copy, dim3(kp.n_rec), dim3(64), PAFFY_COPY_LDS_BYTES, kp);
        if (c->h_info->g_count > 0) LAUNCH(c, "k_emit_lds<line>", k_emit_lds<false>, dim3(kp.n_rec), dim3(PAFFY_NT), PAFFY_EMIT_LDS_BYTES, kp);
        if (c->h_info->w_count > 0) LAUNCH(c, "k_arena_emit<line>", k_arena_emit<false>, dim3(2048), dim3(PAFFY_NT), PAFFY_EMIT_LDS_BYTES, kp);
    }
    return 0;
}

/* lines [first, first + n) of a line plan (tile, dedupe) into d_out, the first of them at d_out[0]: for hosts that drain a
 * large output through a bounded staging buffer. *bytes = what was written. */
int paffy_hip_emit_lines(paffy_hip_ctx *c, int64_t first, int64_t n, void *d_out, int64_t out_cap, int64_t *bytes) {
    if (!c || !bytes || first < 0 || n < 0) return PAFFY_E_ARG;
    if (!c->planned || !c->plan_is_tile) return PAFFY_E_STATE;
    *bytes = 0;
    if ((uint64_t)(first + n) > c->line_n) return PAFFY_E_ARG;
    if (n == 0) return 0;
    if (!d_out || (reinterpret_cast<uintptr_t>(d_out) & 15)) return PAFFY_E_ARG;
    uint64_t lo = 0, hi = (uint64_t)c->plan.out_bytes;
    HIPCHK(c, hipMemcpyAsync(&lo, c->line_off + first, sizeof(lo), hipMemcpyDeviceToHost, c->stream));
    if ((uint64_t)(first + n) < c->line_n) HIPCHK(c, hipMemcpyAsync(&hi, c->line_off + first + n, sizeof(hi), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if ((int64_t)(hi - lo) > out_cap) return PAFFY_E_CAPACITY;
    LAUNCH(c, "k_tile_emit", k_line_emit, dim3((unsigned)((n + PAFFY_NWAVE - 1) / PAFFY_NWAVE)), dim3(PAFFY_NT), 0, c->line_batches, c->line_meta, c->line_order, c->line_level,
           c->line_off, (uint64_t)first, (uint64_t)n, lo, static_cast<uint8_t *>(d_out));
    *bytes = (int64_t)(hi - lo);
    return 0;
}

int paffy_hip_sync(paffy_hip_ctx *c) {
    if (!c) return PAFFY_E_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->profile) prof_collect(c);
    return 0;
}

int paffy_hip_run_host(paffy_hip_ctx *c, const paffy_stage *stages, int32_t n_stages, const char *h_in, int64_t in_len, char **h_out,
                       int64_t *out_len, paffy_plan_info *info) {
    if (!c || !h_out || !out_len || !info) return PAFFY_E_ARG;
    *h_out = nullptr;
    *out_len = 0;
    /* the device buffers stay with the context: the record API (host/paf_api.c) makes one such call per record */
    void *d_in = nullptr, *d_out = nullptr;
    int rc = 0;
    if (in_len > 0) {
        if (ensure(c, c->host_in, (size_t)in_len + 64)) return PAFFY_E_HIP;
        d_in = c->host_in.p;
        if (hipMemcpyAsync(d_in, h_in, (size_t)in_len, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = PAFFY_E_HIP;
    }
    if (!rc) rc = paffy_hip_plan(c, stages, n_stages, d_in, in_len, info);
    if (!rc && info->out_bytes > 0) {
        if (ensure(c, c->host_out, (size_t)info->out_bytes + 64)) return PAFFY_E_HIP;
        d_out = c->host_out.p;
        rc = paffy_hip_emit(c, d_out, info->out_bytes + 64);
        if (!rc) {
            *h_out = static_cast<char *>(malloc((size_t)info->out_bytes));
            if (!*h_out || hipMemcpyAsync(*h_out, d_out, (size_t)info->out_bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = PAFFY_E_HIP;
            *out_len = info->out_bytes;
        }
        if (!rc) rc = paffy_hip_sync(c);
    }
    c->planned = false; /* the input buffer will be overwritten by the next call */
    return rc;
}

/*
 * Streaming through host buffers (the loop of impl/paf_invert.c:84-89 and friends with the GPU in the middle): two slots, each a
 * pinned input buffer and device in / out buffers; the output returns through two pinned pieces. Chunk k + 1 is copied in, planned
 * and written on the GPU while the host still drains chunk k: H2D, kernels and D2H run on three streams.
 */
struct StreamSlot {
    char *h_in = nullptr;
    size_t h_cap = 0;
    void *d_in = nullptr;
    size_t d_in_cap = 0;
    void *d_out = nullptr;
    size_t d_out_cap = 0;
    hipEvent_t ev_in = nullptr, ev_emit = nullptr;
    int64_t out_len = 0, read_at = 0; /* bytes of output, bytes already handed to the host */
    bool busy = false;
};
struct paffy_hip_stream {
    paffy_hip_ctx *c = nullptr;
    paffy_stage stages[PAFFY_MAX_STAGES];
    int32_t n_stages = 0;
    StreamSlot slot[2];
    int fill = 0, drain = 0; /* slot the next chunk goes to, slot being read */
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    /* three output pieces: the one handed out, the one handed out before it (still the caller's until the next read) and the one in flight */
    char *piece[3] = {nullptr, nullptr, nullptr};
    size_t piece_cap = 0;
    hipEvent_t ev_piece[3] = {nullptr, nullptr, nullptr};
    int64_t piece_len[3] = {0, 0, 0};
    int64_t issued_at = 0; /* bytes of the draining chunk whose copy has been issued */
    int n_issued = 0, n_returned = 0;
};

static void stream_detach(paffy_hip_stream *s) { s->c = nullptr; }

int paffy_hip_stream_trim(paffy_hip_ctx *c) {
    if (!c) return PAFFY_E_ARG;
    for (int k = 0; k < 2; k++)
        for (int b = 0; b < 2; b++) {
            if (c->kept_slot_buf[k][b]) (void)hipFree(c->kept_slot_buf[k][b]);
            c->kept_slot_buf[k][b] = nullptr;
            c->kept_slot_cap[k][b] = 0;
        }
    return 0;
}

int paffy_hip_stream_open(paffy_hip_ctx *c, const paffy_stage *stages, int32_t n_stages, int64_t chunk_bytes, int64_t piece_bytes, paffy_hip_stream **out) {
    if (!c || !out || n_stages < 0 || n_stages > PAFFY_MAX_STAGES || (n_stages > 0 && !stages) || chunk_bytes < 4096 || chunk_bytes >= (1ll << 31) - 64 || piece_bytes < 4096)
        return PAFFY_E_ARG;
    paffy_hip_stream *s = new paffy_hip_stream();
    s->c = c;
    c->open_streams.push_back(s);
    s->n_stages = n_stages;
    for (int32_t i = 0; i < n_stages; i++) s->stages[i] = stages[i];
    bool ok = hipStreamCreateWithFlags(&s->s_h2d, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&s->s_d2h, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; k < 2; k++) { /* the device buffers the stream before left with the context */
        StreamSlot &sl = s->slot[k];
        sl.d_in = c->kept_slot_buf[k][0]; sl.d_in_cap = c->kept_slot_cap[k][0];
        sl.d_out = c->kept_slot_buf[k][1]; sl.d_out_cap = c->kept_slot_cap[k][1];
        c->kept_slot_buf[k][0] = c->kept_slot_buf[k][1] = nullptr;
        c->kept_slot_cap[k][0] = c->kept_slot_cap[k][1] = 0;
    }
    for (int k = 0; k < 2 && ok; k++) {
        StreamSlot &sl = s->slot[k];
        sl.h_cap = (size_t)chunk_bytes;
        ok = hipHostMalloc(reinterpret_cast<void **>(&sl.h_in), sl.h_cap + 64) == hipSuccess && hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&sl.ev_emit, hipEventDisableTiming) == hipSuccess;
    }
    for (int k = 0; k < 3 && ok; k++)
        ok = hipHostMalloc(reinterpret_cast<void **>(&s->piece[k]), (size_t)piece_bytes) == hipSuccess && hipEventCreateWithFlags(&s->ev_piece[k], hipEventDisableTiming) == hipSuccess;
    s->piece_cap = (size_t)piece_bytes;
    if (!ok) {
        c->last_error = "paffy_hip_stream_open: pinned buffers / streams";
        paffy_hip_stream_close(s);
        return PAFFY_E_HIP;
    }
    *out = s;
    return 0;
}

void paffy_hip_stream_close(paffy_hip_stream *s) {
    if (s && s->c) { /* no longer an open stream of its context */
        auto &v = s->c->open_streams;
        for (size_t k = 0; k < v.size(); k++)
            if (v[k] == s) {
                v.erase(v.begin() + (long)k);
                break;
            }
    }
    if (!s) return;
    (void)hipDeviceSynchronize();
    for (int k = 0; k < 2; k++) {
        StreamSlot &sl = s->slot[k];
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        /* the device buffers stay with the context for its next stream (freed by paffy_hip_destroy or paffy_hip_stream_trim) */
        void *bufs[2] = {sl.d_in, sl.d_out};
        const size_t caps[2] = {sl.d_in_cap, sl.d_out_cap};
        for (int b = 0; b < 2; b++) {
            if (!bufs[b]) continue;
            if (s->c && !s->c->kept_slot_buf[k][b]) {
                s->c->kept_slot_buf[k][b] = bufs[b];
                s->c->kept_slot_cap[k][b] = caps[b];
            } else {
                (void)hipFree(bufs[b]);
            }
        }
        if (sl.ev_in) (void)hipEventDestroy(sl.ev_in);
        if (sl.ev_emit) (void)hipEventDestroy(sl.ev_emit);
    }
    for (int k = 0; k < 3; k++) {
        if (s->piece[k]) (void)hipHostFree(s->piece[k]);
        if (s->ev_piece[k]) (void)hipEventDestroy(s->ev_piece[k]);
    }
    if (s->s_h2d) (void)hipStreamDestroy(s->s_h2d);
    if (s->s_d2h) (void)hipStreamDestroy(s->s_d2h);
    delete s;
}

/* the pinned buffer the next chunk is to be written into (NULL while both slots hold unread output); with want > its capacity it is
   enlarged, keeping the first `keep` bytes (a line longer than the chunk) */
char *paffy_hip_stream_input(paffy_hip_stream *s, int64_t want, int64_t keep, int64_t *cap) {
    if (!s || !cap) return nullptr;
    StreamSlot &sl = s->slot[s->fill];
    if (sl.busy) return nullptr;
    if (want > (int64_t)sl.h_cap) {
        if (want >= (1ll << 31) - 64) return nullptr;
        char *nb = nullptr;
        if (hipHostMalloc(reinterpret_cast<void **>(&nb), (size_t)want + 64) != hipSuccess) return nullptr;
        if (keep > 0) memcpy(nb, sl.h_in, (size_t)keep);
        (void)hipHostFree(sl.h_in);
        sl.h_in = nb;
        sl.h_cap = (size_t)want;
    }
    *cap = (int64_t)sl.h_cap;
    return sl.h_in;
}

/* the first in_len bytes of the current input buffer (whole lines) go through the stage list; *info is complete on return (the
   plan is synchronous), the output is drained with paffy_hip_stream_read */
int paffy_hip_stream_submit(paffy_hip_stream *s, int64_t in_len, paffy_plan_info *info) {
    if (!s || !info || in_len < 0) return PAFFY_E_ARG;
    paffy_hip_ctx *c = s->c;
    StreamSlot &sl = s->slot[s->fill];
    if (sl.busy || in_len > (int64_t)sl.h_cap) return PAFFY_E_STATE;
    if ((size_t)in_len + 64 > sl.d_in_cap) {
        if (sl.d_in) HIPCHK(c, hipFree(sl.d_in));
        sl.d_in = nullptr;
        sl.d_in_cap = (size_t)in_len + (size_t)in_len / 4 + 4096;
        HIPCHK(c, hipMalloc(&sl.d_in, sl.d_in_cap));
    }
    if (in_len > 0) HIPCHK(c, hipMemcpyAsync(sl.d_in, sl.h_in, (size_t)in_len, hipMemcpyHostToDevice, s->s_h2d));
    HIPCHK(c, hipEventRecord(sl.ev_in, s->s_h2d));
    HIPCHK(c, hipStreamWaitEvent(c->stream, sl.ev_in, 0));
    int rc = paffy_hip_plan(c, s->stages, s->n_stages, sl.d_in, in_len, info);
    if (rc) return rc;
    sl.out_len = info->out_bytes;
    sl.read_at = 0;
    if (sl.out_len > 0) {
        if ((size_t)sl.out_len + 64 > sl.d_out_cap) {
            if (sl.d_out) HIPCHK(c, hipFree(sl.d_out));
            sl.d_out = nullptr;
            sl.d_out_cap = (size_t)sl.out_len + (size_t)sl.out_len / 4 + 4096;
            HIPCHK(c, hipMalloc(&sl.d_out, sl.d_out_cap));
        }
        rc = paffy_hip_emit(c, sl.d_out, (int64_t)sl.d_out_cap);
        if (rc) return rc;
    }
    HIPCHK(c, hipEventRecord(sl.ev_emit, c->stream));
    sl.busy = true;
    s->fill ^= 1;
    return 0;
}

/* the next piece of the oldest submitted chunk's output: *len bytes at *piece, valid until the call after next (three pinned pieces
   rotate: this call starts the copy that follows the piece it hands out, into the buffer handed out two calls ago); *len = 0: that
   chunk has been read completely (or nothing was submitted) */
int paffy_hip_stream_read(paffy_hip_stream *s, const char **piece, int64_t *len) {
    if (!s || !piece || !len) return PAFFY_E_ARG;
    paffy_hip_ctx *c = s->c;
    *piece = nullptr;
    *len = 0;
    StreamSlot &sl = s->slot[s->drain];
    if (!sl.busy) return 0;
    if (s->n_issued == 0) HIPCHK(c, hipStreamWaitEvent(s->s_d2h, sl.ev_emit, 0)); /* first piece of this chunk */
    /* keep two copies in flight: the piece handed out now and the one after it (PAFFY_D2H_INFLIGHT: experiments) */
    static const int in_flight = getenv("PAFFY_D2H_INFLIGHT") ? atoi(getenv("PAFFY_D2H_INFLIGHT")) : 2;
    while (s->n_issued < s->n_returned + (in_flight < 1 ? 1 : in_flight > 2 ? 2 : in_flight) && s->issued_at < sl.out_len) {
        const int k = s->n_issued % 3;
        const int64_t n = sl.out_len - s->issued_at < (int64_t)s->piece_cap ? sl.out_len - s->issued_at : (int64_t)s->piece_cap;
        HIPCHK(c, hipMemcpyAsync(s->piece[k], static_cast<char *>(sl.d_out) + s->issued_at, (size_t)n, hipMemcpyDeviceToHost, s->s_d2h));
        HIPCHK(c, hipEventRecord(s->ev_piece[k], s->s_d2h));
        s->piece_len[k] = n;
        s->issued_at += n;
        s->n_issued++;
    }
    if (s->n_returned == s->n_issued) { /* everything of this chunk has been handed out */
        sl.busy = false;
        s->drain ^= 1;
        s->issued_at = 0;
        s->n_issued = s->n_returned = 0;
        return 0;
    }
    const int k = s->n_returned % 3;
    HIPCHK(c, hipEventSynchronize(s->ev_piece[k]));
    *piece = s->piece[k];
    *len = s->piece_len[k];
    s->n_returned++;
    return 0;
}

/* exclusive scan of n int64 values on the device (two levels); *total on the host */
static int scan64(paffy_hip_ctx *c, const int64_t *in, uint64_t n, int64_t *out, int64_t *total) {
    const uint64_t tiles = (n + PAFFY_NT * 16 - 1) / (PAFFY_NT * 16);
    if (ensure(c, c->bed_tiles, sizeof(int64_t) * (size_t)(2 * tiles + 2))) return PAFFY_E_HIP;
    int64_t *sums = static_cast<int64_t *>(c->bed_tiles.p), *offs = sums + tiles, *tot = offs + tiles;
    LAUNCH(c, "k_scan64_tiles", k_scan64_tiles, dim3((unsigned)tiles), dim3(PAFFY_NT), 0, in, n, sums);
    LAUNCH(c, "k_scan_i64", k_scan_i64, dim3(1), dim3(PAFFY_NT), 0, sums, (uint32_t)tiles, offs, tot);
    LAUNCH(c, "k_scan64_fix", k_scan64_fix, dim3((unsigned)tiles), dim3(PAFFY_NT), 0, in, n, static_cast<const int64_t *>(offs), out);
    HIPCHK(c, hipMemcpyAsync(total, tot, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

/*
 * `paffy to_bed` (impl/paf_to_bed.c:166-190) over any number of text batches: coverage counters per sequence as in `paffy tile`
 * (every record bumps the counters of its query range; with include_inverted also those of its target range, as the inverted
 * record would), then each sequence's counters as maximal runs. Sequences are written in order of first appearance (the
 * reference iterates a sonLib hash: its order is not defined). Any failing record means no output.
 */
int paffy_hip_bed_begin(paffy_hip_ctx *c, const paffy_bed_opts *opts) {
    if (c) (void)paffy_hip_stream_trim(c); /* a closed stream's slot buffers (tens of GB) are this command's to use */
    if (!c || !opts) return PAFFY_E_ARG;
    c->planned = false;
    index_drop_all(c);
    return cov_begin(c, opts->include_inverted ? 2 : 1);
}
int paffy_hip_bed_add(paffy_hip_ctx *c, const void *d_in, int64_t in_len) {
    if (!c) return PAFFY_E_ARG;
    return cov_add(c, d_in, in_len, false);
}
int paffy_hip_bed_run(paffy_hip_ctx *c, const paffy_bed_opts *opts, paffy_plan_info *info) {
    if (!c || !info || !opts) return PAFFY_E_ARG;
    CovState &S = cov_state(c);
    if (S.sides != (opts->include_inverted ? 2u : 1u)) return PAFFY_E_STATE;
    c->planned = false;
    c->plan_is_tile = false;
    c->plan_is_bed = true;
    memset(info, 0, sizeof(*info));
    memset(&c->plan, 0, sizeof(c->plan));
    memset(&c->kp, 0, sizeof(c->kp));
    c->bed_runs = 0;
    for (const CovBatch &b : S.batches) c->plan.in_bytes += b.len;
    c->plan.n_records = (int64_t)S.n_rec;
    if (S.n_rec == 0) {
        *info = c->plan;
        c->planned = true;
        return 0;
    }
    int rc = cov_run(c, 1, &c->plan.error);
    if (rc) return rc;
    if (c->plan.error.code) {
        if (c->plan.error.code == PAFFY_ERR_CIGAR_CHAR) c->plan.error.stage = -1; /* the read loop parses the cigar (paf_read(.., 1)) */
        *info = c->plan;
        c->planned = true;
        return 0;
    }
    /* the runs: sequences in order of first appearance = ascending counter bases */
    const size_t ns = S.n_contigs;
    std::vector<uint64_t> cbase(ns + 1);
    std::vector<int64_t> clen(ns);
    std::vector<uint32_t> name_off(ns), name_len(ns);
    {
        std::vector<CovName> cname(ns);
        if (cov_fetch(c, cname.data(), S.name_tab.p, sizeof(CovName) * ns)) return PAFFY_E_HIP;
        std::vector<uint32_t> boff(ns + 1);
        uint32_t at = 0;
        for (size_t ci = 0; ci < ns; ci++) {
            boff[ci] = at;
            at += cname[ci].len;
        }
        for (size_t k = 0; k < ns; k++) {
            const uint32_t ci = S.appearance[k];
            cbase[k] = S.h_contig_cov[ci];
            clen[k] = S.h_contig_len[ci];
            name_off[k] = boff[ci];
            name_len[k] = cname[ci].len;
        }
        cbase[ns] = S.cov_total;
    }
    if (ensure(c, c->bed_tab, 8 * (2 * ns + 1) + 4 * 2 * ns + 64)) return PAFFY_E_HIP;
    uint64_t *d_cbase = static_cast<uint64_t *>(c->bed_tab.p);
    int64_t *d_clen = reinterpret_cast<int64_t *>(d_cbase + ns + 1);
    uint32_t *d_noff = reinterpret_cast<uint32_t *>(d_clen + ns), *d_nlen = d_noff + ns;
    HIPCHK(c, hipMemcpyAsync(d_cbase, cbase.data(), 8 * (ns + 1), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_clen, clen.data(), 8 * ns, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_noff, name_off.data(), 4 * ns, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_nlen, name_len.data(), 4 * ns, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const uint64_t cov_total = S.cov_total;
    if (!c->bed_params) c->bed_params = new BedParams;
    BedParams &B = *c->bed_params;
    memset(&B, 0, sizeof(B));
    B.counts = static_cast<const uint16_t *>(S.cov.p);
    B.n_counts = cov_total;
    B.contig_base = d_cbase;
    B.contig_len = d_clen;
    B.name_off = d_noff;
    B.name_len = d_nlen;
    B.n_contigs = (uint32_t)ns;
    B.in = static_cast<const uint8_t *>(S.names.p);
    B.binary = opts->binary;
    B.exclude_unaligned = opts->exclude_unaligned;
    B.exclude_aligned = opts->exclude_aligned;
    B.min_size = opts->min_size;
    if (ns == 0 || cov_total == 0) {
        *info = c->plan;
        c->planned = true;
        return 0;
    }
    const uint64_t rtiles = (cov_total + (uint64_t)PAFFY_NT * BED_PER - 1) / ((uint64_t)PAFFY_NT * BED_PER);
    if (ensure(c, c->bed_len, sizeof(int64_t) * (size_t)(2 * rtiles + 2))) return PAFFY_E_HIP;
    int64_t *tile_cnt = static_cast<int64_t *>(c->bed_len.p), *tile_off = tile_cnt + rtiles;
    LAUNCH(c, "k_bed_runs", k_bed_runs, dim3((unsigned)rtiles), dim3(PAFFY_NT), 0, B, static_cast<const int64_t *>(nullptr), tile_cnt, static_cast<uint64_t *>(nullptr));
    int64_t n_runs = 0;
    if (scan64(c, tile_cnt, rtiles, tile_off, &n_runs)) return PAFFY_E_HIP;
    if (ensure(c, c->bed_starts, sizeof(uint64_t) * (size_t)(n_runs + 1))) return PAFFY_E_HIP;
    LAUNCH(c, "k_bed_runs", k_bed_runs, dim3((unsigned)rtiles), dim3(PAFFY_NT), 0, B, static_cast<const int64_t *>(tile_off), tile_cnt, static_cast<uint64_t *>(c->bed_starts.p));
    HIPCHK(c, hipStreamSynchronize(c->stream)); /* bed_len is reused below */
    if (ensure(c, c->bed_len, sizeof(int64_t) * (size_t)(n_runs + 1))) return PAFFY_E_HIP;
    if (ensure(c, c->bed_off, sizeof(int64_t) * (size_t)(n_runs + 1))) return PAFFY_E_HIP;
    const unsigned lgrid = (unsigned)(((uint64_t)n_runs + PAFFY_NT - 1) / PAFFY_NT);
    LAUNCH(c, "k_bed_lines", k_bed_lines, dim3(lgrid), dim3(PAFFY_NT), 0, B, static_cast<const uint64_t *>(c->bed_starts.p), (uint64_t)n_runs,
           static_cast<int64_t *>(c->bed_len.p), static_cast<const int64_t *>(nullptr), static_cast<uint8_t *>(nullptr));
    int64_t total = 0;
    if (scan64(c, static_cast<const int64_t *>(c->bed_len.p), (uint64_t)n_runs, static_cast<int64_t *>(c->bed_off.p), &total)) return PAFFY_E_HIP;
    if (c->profile) prof_collect(c);
    c->bed_runs = (uint64_t)n_runs;
    c->plan.out_bytes = total;
    c->plan.n_rows = 0; /* lines: counted by the caller if wanted */
    *info = c->plan;
    c->planned = true;
    return 0;
}
int paffy_hip_bed_plan(paffy_hip_ctx *c, const void *d_in, int64_t in_len, const paffy_bed_opts *opts, paffy_plan_info *info) {
    if (!c || !info || !opts) return PAFFY_E_ARG;
    if (in_len < 0 || in_len >= (1ll << 31) - 64 || (in_len > 0 && !d_in) || (reinterpret_cast<uintptr_t>(d_in) & 15)) return PAFFY_E_ARG;
    int rc = paffy_hip_bed_begin(c, opts);
    if (!rc) rc = paffy_hip_bed_add(c, d_in, in_len);
    if (!rc) rc = paffy_hip_bed_run(c, opts, info);
    return rc;
}

int paffy_hip_flat_stats(paffy_hip_ctx *c, int64_t *left, int64_t reasons[16]) {
    if (!c || !left) return PAFFY_E_ARG;
    *left = c->flat_left;
    if (reasons)
        for (int k = 0; k < 16; k++) reasons[k] = c->flat_left >= 0 ? c->flat_reasons[k] : 0;
    return 0;
}

int paffy_hip_plan_stats(paffy_hip_ctx *c, int64_t sums[6]) {
    if (!c || !sums) return PAFFY_E_ARG;
    if (!c->planned || c->plan_is_tile) return PAFFY_E_STATE;
    for (int k = 0; k < 6; k++) sums[k] = c->plan.in_bytes > 0 ? (int64_t)c->h_info->stats[k] : 0;
    return 0;
}

int64_t paffy_hip_plan_record_stats(paffy_hip_ctx *c, int64_t cap_records, int64_t *sums) {
    if (!c || !sums || cap_records < 0) return PAFFY_E_ARG;
    if (!c->planned || c->plan_is_tile || c->plan_is_bed) return PAFFY_E_STATE;
    const int64_t n = c->plan.n_records;
    if (n == 0) return 0;
    if (!c->kp.rec_stats) return PAFFY_E_STATE; /* no PAFFY_STATS stage in the plan */
    if (cap_records < n) return PAFFY_E_CAPACITY;
    HIPCHK(c, hipMemcpy(sums, c->kp.rec_stats, sizeof(int64_t) * 6 * (size_t)n, hipMemcpyDeviceToHost));
    return n;
}

/* ---- the counters of a to_bed run, for hosts that keep SequenceCountArray objects (inc/paf.h:214-233) ---- */
__global__ __launch_bounds__(PAFFY_NT) void k_counts_add_sat(uint16_t *acc, const uint16_t *add, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = (uint32_t)acc[i] + add[i]; /* one record at a time stops at INT16_MAX - 1 (impl/paf.c:701); so does the sum */
    acc[i] = acc[i] >= 32766 ? acc[i] : (uint16_t)(s < 32766u ? s : 32766u);
}
int64_t paffy_hip_bed_sequences(paffy_hip_ctx *c) {
    if (!c) return PAFFY_E_ARG;
    if (!c->planned || !c->plan_is_bed || !c->cov) return PAFFY_E_STATE;
    return (int64_t)c->cov->appearance.size();
}
int paffy_hip_bed_counts(paffy_hip_ctx *c, int64_t sequence, int64_t start, int64_t end, uint16_t *h_counts, int accumulate) {
    if (!c || !h_counts || sequence < 0 || start < 0 || end < start) return PAFFY_E_ARG;
    if (!c->planned || !c->plan_is_bed || !c->cov) return PAFFY_E_STATE;
    CovState &S = *c->cov;
    if ((size_t)sequence >= S.appearance.size()) return PAFFY_E_ARG;
    const uint32_t ci = S.appearance[(size_t)sequence];
    if (end > S.h_contig_len[ci]) return PAFFY_E_ARG;
    const uint64_t n = (uint64_t)(end - start);
    if (n == 0) return 0;
    const uint16_t *src = static_cast<const uint16_t *>(S.cov.p) + S.h_contig_cov[ci] + start;
    if (!accumulate) {
        HIPCHK(c, hipMemcpyAsync(h_counts, src, sizeof(uint16_t) * n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    if (ensure(c, c->pretty_out, sizeof(uint16_t) * n + 64)) return PAFFY_E_HIP;
    uint16_t *acc = static_cast<uint16_t *>(c->pretty_out.p);
    HIPCHK(c, hipMemcpyAsync(acc, h_counts, sizeof(uint16_t) * n, hipMemcpyHostToDevice, c->stream));
    LAUNCH(c, "k_counts_add_sat", k_counts_add_sat, dim3((unsigned)((n + PAFFY_NT - 1) / PAFFY_NT)), dim3(PAFFY_NT), 0, acc, src, n);
    HIPCHK(c, hipMemcpyAsync(h_counts, acc, sizeof(uint16_t) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

/* ---- paf_pretty_print's base-level rows (pretty_kernel.h) ---- */
int paffy_hip_keep_raw_sequences(paffy_hip_ctx *c, int on) {
    if (!c) return PAFFY_E_ARG;
    c->keep_raw = on != 0;
    return 0;
}

static int pretty_params(paffy_hip_ctx *c, int64_t first, int64_t count, PrettyParams *pp) {
    if (!c->planned || c->plan_is_tile || c->plan_is_bed) return PAFFY_E_STATE;
    if (first < 0 || count < 0 || first + count > c->plan.n_records) return PAFFY_E_ARG;
    if (c->n_seqs <= 0 || !c->seq_raw.p) {
        c->last_error = "alignment rows: no sequences loaded with paffy_hip_keep_raw_sequences on";
        return PAFFY_E_STATE;
    }
    const uint32_t n = c->kp.n_rec;
    if (!c->plan_seq_lookup && n > 0) { /* a plan without the mismatch encoder did not look the names up */
        if (ensure(c, c->rec_qseq, sizeof(int32_t) * (size_t)(n + 1))) return PAFFY_E_HIP;
        if (ensure(c, c->rec_tseq, sizeof(int32_t) * (size_t)(n + 1))) return PAFFY_E_HIP;
        LAUNCH(c, "k_seq_lookup", k_seq_lookup, dim3((n + PAFFY_NT - 1) / PAFFY_NT), dim3(PAFFY_NT), 0, c->kp.in, c->kp.meta, n,
               static_cast<const uint8_t *>(c->seq_names.p), static_cast<const uint32_t *>(c->seq_name_off.p), c->n_seqs,
               static_cast<int32_t *>(c->rec_qseq.p), static_cast<int32_t *>(c->rec_tseq.p));
        c->plan_seq_lookup = true;
    }
    memset(pp, 0, sizeof(*pp));
    pp->meta = c->kp.meta;
    pp->plan = static_cast<const RecPlan *>(c->rec_plan.p);
    pp->status = c->kp.status;
    pp->arena = c->kp.arena;
    pp->arena_off = c->kp.arena_off;
    pp->ops_mirror = c->kp.ops_mirror;
    pp->seq_raw = static_cast<const uint8_t *>(c->seq_raw.p);
    pp->seqs = static_cast<const SeqEntry *>(c->seq_table.p);
    pp->rec_qseq = static_cast<const int32_t *>(c->rec_qseq.p);
    pp->rec_tseq = static_cast<const int32_t *>(c->rec_tseq.p);
    pp->first = (uint32_t)first;
    return 0;
}

int paffy_hip_plan_alignment_sizes(paffy_hip_ctx *c, int64_t first, int64_t count, int64_t *h_bytes) {
    if (!c || (count > 0 && !h_bytes)) return PAFFY_E_ARG;
    PrettyParams pp;
    int rc = pretty_params(c, first, count, &pp);
    if (rc || count == 0) return rc;
    if (ensure(c, c->pretty_off, sizeof(int64_t) * (size_t)(count + 1))) return PAFFY_E_HIP;
    LAUNCH(c, "k_pretty_size", k_pretty_size, dim3((unsigned)count), dim3(PRETTY_NT), 0, pp, static_cast<int64_t *>(c->pretty_off.p));
    HIPCHK(c, hipMemcpyAsync(h_bytes, c->pretty_off.p, sizeof(int64_t) * (size_t)count, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int paffy_hip_plan_alignment_rows(paffy_hip_ctx *c, int64_t first, int64_t count, const int64_t *h_off, char *h_out, paffy_error *err) {
    if (!c || !err || (count > 0 && (!h_off || !h_out))) return PAFFY_E_ARG;
    memset(err, 0, sizeof(*err));
    PrettyParams pp;
    int rc = pretty_params(c, first, count, &pp);
    if (rc || count == 0) return rc;
    const int64_t total = h_off[count] - h_off[0];
    if (total < 0) return PAFFY_E_ARG;
    if (ensure(c, c->pretty_off, sizeof(int64_t) * (size_t)(count + 1))) return PAFFY_E_HIP;
    if (ensure(c, c->pretty_out, (size_t)total + 64)) return PAFFY_E_HIP;
    if (ensure(c, c->pretty_err, sizeof(unsigned long long))) return PAFFY_E_HIP;
    std::vector<int64_t> off((size_t)count + 1);
    for (int64_t i = 0; i <= count; i++) off[(size_t)i] = h_off[i] - h_off[0];
    HIPCHK(c, hipMemcpyAsync(c->pretty_off.p, off.data(), sizeof(int64_t) * (size_t)(count + 1), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->pretty_err.p, 0xff, sizeof(unsigned long long), c->stream));
    pp.row_off = static_cast<const int64_t *>(c->pretty_off.p);
    pp.out = static_cast<uint8_t *>(c->pretty_out.p);
    pp.err = static_cast<unsigned long long *>(c->pretty_err.p);
    LAUNCH(c, "k_pretty_rows", k_pretty_rows, dim3((unsigned)count), dim3(PRETTY_NT), 0, pp);
    unsigned long long e = 0;
    HIPCHK(c, hipMemcpyAsync(&e, c->pretty_err.p, sizeof(e), hipMemcpyDeviceToHost, c->stream));
    if (total > 0) HIPCHK(c, hipMemcpyAsync(h_out, c->pretty_out.p, (size_t)total, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (e != ~0ull) {
        err->code = (int32_t)(e & 0xff);
        err->record = (int64_t)(e >> 8);
        err->stage = -1;
    }
    return 0;
}

int paffy_hip_parse_host(paffy_hip_ctx *c, const char *h_in, int64_t in_len, paffy_record **recs, uint64_t **ops, int64_t *n_ops_total,
                         paffy_plan_info *info) {
    if (!c || !recs || !ops || !n_ops_total || !info || in_len < 0 || (in_len > 0 && !h_in)) return PAFFY_E_ARG;
    *recs = nullptr;
    *ops = nullptr;
    *n_ops_total = 0;
    void *d_in = nullptr;
    int rc = 0;
    if (in_len > 0) { /* the context's own input buffer, like paffy_hip_run_host */
        if (ensure(c, c->host_in, (size_t)in_len + 64)) return PAFFY_E_HIP;
        d_in = c->host_in.p;
        if (hipMemcpy(d_in, h_in, (size_t)in_len, hipMemcpyHostToDevice) != hipSuccess) rc = PAFFY_E_HIP;
    }
    const paffy_stage pass = {PAFFY_PASS, 0.0f, 0.0f};
    if (!rc) rc = paffy_hip_plan(c, &pass, 1, d_in, in_len, info); /* index, header parse, cigar parse: the ops are in the mirror / arena */
    const size_t n = (size_t)info->n_records;
    if (!rc && info->error.code == 0 && n > 0) {
        std::vector<RecMeta> meta(n);
        std::vector<RecPlan> plan(n);
        std::vector<uint32_t> status(n), mirror((size_t)in_len / 2 + 64);
        std::vector<uint64_t> arena_off(n), arena((size_t)c->h_info->arena_used);
        if (hipMemcpy(meta.data(), c->meta.p, sizeof(RecMeta) * n, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(plan.data(), c->rec_plan.p, sizeof(RecPlan) * n, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(status.data(), c->status.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(arena_off.data(), c->arena_off.p, sizeof(uint64_t) * n, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(mirror.data(), c->ops_mirror.p, sizeof(uint32_t) * mirror.size(), hipMemcpyDeviceToHost) != hipSuccess ||
            (!arena.empty() && hipMemcpy(arena.data(), c->arena.p, sizeof(uint64_t) * arena.size(), hipMemcpyDeviceToHost) != hipSuccess)) {
            c->last_error = "paffy_hip_parse_host: copy back failed";
            rc = PAFFY_E_HIP;
        } else {
            int64_t total = 0;
            for (size_t r = 0; r < n; r++)
                if (plan[r].flags & 8u) total += plan[r].n;
            paffy_record *out = static_cast<paffy_record *>(calloc(n, sizeof(paffy_record)));
            uint64_t *oo = static_cast<uint64_t *>(malloc(sizeof(uint64_t) * (size_t)(total > 0 ? total : 1)));
            int64_t at = 0;
            for (size_t r = 0; r < n; r++) {
                const RecMeta &m = meta[r];
                paffy_record &o = out[r];
                o.query_length = m.qlen; o.query_start = m.qs; o.query_end = m.qe;
                o.target_length = m.tlen; o.target_start = m.ts; o.target_end = m.te;
                o.score = m.score; o.mapping_quality = m.mapq; o.num_matches = m.nmatch; o.num_bases = m.nbases;
                o.tile_level = m.tile_level; o.chain_id = m.chain_id; o.chain_score = m.chain_score;
                o.query_name_off = m.qname_off; o.query_name_len = m.qname_len; o.target_name_off = m.tname_off; o.target_name_len = m.tname_len;
                o.cigar_off = m.has_cg ? m.cg_off : 0; o.cigar_len = m.has_cg ? m.cg_len : 0;
                o.same_strand = m.same_strand; o.type = m.type;
                o.ops_first = at;
                o.n_ops = -1;
                if (plan[r].flags & 8u) { /* has a cigar: the ops as parsed (a pass stage leaves the view on the whole array) */
                    o.n_ops = plan[r].n;
                    if ((status[r] >> 16) == KLASS_ARENA) {
                        const uint64_t *src = arena.data() + arena_off[r];
                        for (uint32_t i = 0; i < plan[r].n; i++) oo[at++] = (((uint64_t)((int64_t)src[i] >> 8)) & 0x00ffffffffffffffull) | ((src[i] & 0xffull) << 56);
                    } else {
                        const uint32_t *src = mirror.data() + (m.cg_off >> 1);
                        const uint16_t *src16 = reinterpret_cast<const uint16_t *>(src);
                        const bool half = (plan[r].flags & 0x60000u) == 0x40000u; /* 2-byte words (every length below 8192): emit_ops_half() */
                        for (uint32_t i = 0; i < plan[r].n; i++) {
                            const uint32_t w = half ? (uint32_t)src16[i] : src[i];
                            oo[at++] = (uint64_t)(w >> 3) | ((uint64_t)(w & 7u) << 56);
                        }
                    }
                }
            }
            *recs = out;
            *ops = oo;
            *n_ops_total = at;
        }
    }
    c->planned = false; /* the input buffer will be overwritten by the next call */
    return rc;
}

/* Lays the sequence store out for n named sequences (sorted by name for the lookup kernel) and uploads the name tables;
 * blob_off[i] = where sequence i starts in seq_blob. The bases are written by the caller. */
static int seq_store_layout(paffy_hip_ctx *c, int64_t n, const char *const *names, const int64_t *lens, std::vector<uint64_t> &blob_off) {
    c->n_seqs = 0;
    std::vector<int64_t> order((size_t)n);
    for (int64_t i = 0; i < n; i++) order[(size_t)i] = i;
    std::vector<size_t> nlen((size_t)n);
    for (int64_t i = 0; i < n; i++) nlen[(size_t)i] = strlen(names[i]);
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        size_t m = nlen[(size_t)a] < nlen[(size_t)b] ? nlen[(size_t)a] : nlen[(size_t)b];
        int d = memcmp(names[a], names[b], m);
        if (d) return d < 0;
        return nlen[(size_t)a] < nlen[(size_t)b];
    });
    std::vector<uint32_t> name_off((size_t)n + 1);
    std::string blob;
    std::vector<SeqEntry> table((size_t)n);
    blob_off.assign((size_t)n, 0);
    uint64_t total = 0;
    for (int64_t k = 0; k < n; k++) {
        int64_t i = order[(size_t)k];
        name_off[(size_t)k] = (uint32_t)blob.size();
        blob.append(names[i], nlen[(size_t)i]);
        table[(size_t)k].off = total;
        table[(size_t)k].len = lens[i];
        blob_off[(size_t)i] = total;
        total += (uint64_t)lens[i];
    }
    name_off[(size_t)n] = (uint32_t)blob.size();
    if (ensure(c, c->seq_blob, total + 64)) return PAFFY_E_HIP;
    if (ensure(c, c->seq_table, sizeof(SeqEntry) * (size_t)n)) return PAFFY_E_HIP;
    if (ensure(c, c->seq_names, blob.size() + 16)) return PAFFY_E_HIP;
    if (ensure(c, c->seq_name_off, sizeof(uint32_t) * ((size_t)n + 1))) return PAFFY_E_HIP;
    HIPCHK(c, hipMemcpy(c->seq_table.p, table.data(), sizeof(SeqEntry) * (size_t)n, hipMemcpyHostToDevice));
    if (!blob.empty()) HIPCHK(c, hipMemcpy(c->seq_names.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->seq_name_off.p, name_off.data(), sizeof(uint32_t) * ((size_t)n + 1), hipMemcpyHostToDevice));
    return 0;
}

/* After the bases are in seq_blob: upper-case them and build the complemented copy. */
static int seq_store_canon(paffy_hip_ctx *c) {
    const size_t bytes = c->seq_blob.cap & ~(size_t)15; /* the whole allocation, whole 16-byte words */
    if (c->keep_raw) {
        if (ensure(c, c->seq_raw, c->seq_blob.cap)) return PAFFY_E_HIP;
        HIPCHK(c, hipMemcpyAsync(c->seq_raw.p, c->seq_blob.p, c->seq_blob.cap, hipMemcpyDeviceToDevice, c->stream));
    }
    if (ensure(c, c->seq_comp, c->seq_blob.cap)) return PAFFY_E_HIP;
    const uint64_t n16 = bytes / 16;
    if (n16) {
        LAUNCH(c, "k_seq_canon", k_seq_canon, dim3((unsigned)((n16 + PAFFY_NT - 1) / PAFFY_NT)), dim3(PAFFY_NT), 0, static_cast<uint8_t *>(c->seq_blob.p),
               static_cast<uint8_t *>(c->seq_comp.p), n16);
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}

int paffy_hip_set_sequences(paffy_hip_ctx *c, int64_t n, const char *const *names, const char *const *seqs, const int64_t *lens) {
    if (!c || n < 0 || (n > 0 && (!names || !seqs || !lens))) return PAFFY_E_ARG;
    c->n_seqs = 0;
    if (n == 0) return 0;
    std::vector<uint64_t> blob_off;
    int rc = seq_store_layout(c, n, names, lens, blob_off);
    if (rc) return rc;
    for (int64_t i = 0; i < n; i++)
        if (lens[i] > 0)
            HIPCHK(c, hipMemcpy(static_cast<uint8_t *>(c->seq_blob.p) + blob_off[(size_t)i], seqs[i], (size_t)lens[i], hipMemcpyHostToDevice));
    if (seq_store_canon(c)) return PAFFY_E_HIP;
    c->n_seqs = (int32_t)n;
    return 0;
}

int paffy_hip_set_filter(paffy_hip_ctx *c, const paffy_filter *f) {
    if (!c) return PAFFY_E_ARG;
    const paffy_filter d = {-1, -1, -1.0, -1.0, -1, 0};
    c->filter = f ? *f : d;
    return 0;
}

int paffy_hip_error_exit_status(int32_t code) {
    switch (code) {
        case PAFFY_OK: return 0;
        case PAFFY_ERR_STRAND: case PAFFY_ERR_CIGAR_CHAR:
        case PAFFY_ERR_CHECK_QSTART: case PAFFY_ERR_CHECK_QEND: case PAFFY_ERR_CHECK_TSTART: case PAFFY_ERR_CHECK_TEND:
        case PAFFY_ERR_CHECK_CIGAR_Q: case PAFFY_ERR_CHECK_CIGAR_T:
        case PAFFY_ERR_MISSING_QUERY_SEQ: case PAFFY_ERR_MISSING_TARGET_SEQ:
            return 1; /* st_errAbort / exit(1) */
        case PAFFY_ERR_FEW_FIELDS: case PAFFY_ERR_NULL_CIGAR: case PAFFY_ERR_SEQ_RANGE:
            return 139; /* the reference dereferences NULL / reads out of bounds */
        default:
            return 134; /* assert -> abort() */
    }
}

const char *paffy_hip_error_string(int32_t code) {
    switch (code) {
        case PAFFY_OK: return "ok";
        case PAFFY_ERR_FEW_FIELDS: return "Paf line has fewer than 12 fields";
        case PAFFY_ERR_STRAND: return "Got an unexpected strand character in a paf string";
        case PAFFY_ERR_TP_ASSERT: return "tp tag is not P, S or I";
        case PAFFY_ERR_CIGAR_CHAR: return "Got an unexpected character paf cigar string";
        case PAFFY_ERR_CHECK_QSTART: return "Paf query start coordinates are invalid";
        case PAFFY_ERR_CHECK_QEND: return "Paf query end coordinates are invalid";
        case PAFFY_ERR_CHECK_TSTART: return "Paf target start coordinates are invalid";
        case PAFFY_ERR_CHECK_TEND: return "Paf target end coordinates are invalid";
        case PAFFY_ERR_CHECK_CIGAR_Q: return "Paf cigar alignment does not match query length";
        case PAFFY_ERR_CHECK_CIGAR_T: return "Paf cigar alignment does not match target length";
        case PAFFY_ERR_SHATTER_ZERO_LEN: return "shatter: cigar op of length < 1";
        case PAFFY_ERR_SHATTER_BAD_OP: return "shatter: cigar op other than M, I or D";
        case PAFFY_ERR_SHATTER_END: return "shatter: cigar does not end on the record's end coordinates";
        case PAFFY_ERR_TRIM_IDENTITY_ASSERT: return "trim: final identity below the starting identity";
        case PAFFY_ERR_TRIM_FIXED_ASSERT: return "trim: fraction outside [0, 1]";
        case PAFFY_ERR_NULL_CIGAR: return "record has no cigar";
        case PAFFY_ERR_MISSING_QUERY_SEQ: return "No query sequence found";
        case PAFFY_ERR_MISSING_TARGET_SEQ: return "No target sequence found";
        case PAFFY_ERR_TILE_ASSERT: return "tile: coverage assertion failed";
        case PAFFY_ERR_SEQ_RANGE: return "alignment reaches outside a sequence";
        case PAFFY_ERR_CHAIN_ASSERT: return "chain: trim fraction outside [0, 1] or a negative alignment length";
        default: return "unknown error";
    }
}

int paffy_hip_profile_enable(paffy_hip_ctx *c, int on) {
    if (!c) return PAFFY_E_ARG;
    c->profile = on != 0;
    return 0;
}
int paffy_hip_profile_only(paffy_hip_ctx *c, const char *kernel) {
    if (!c) return PAFFY_E_ARG;
    c->profile_only = kernel ? kernel : "";
    return 0;
}
int paffy_hip_profile_reset(paffy_hip_ctx *c) {
    if (!c) return PAFFY_E_ARG;
    prof_collect(c);
    for (auto &e : c->prof) {
        e.ms = 0;
        e.launches = 0;
    }
    return 0;
}
int paffy_hip_profile_read(paffy_hip_ctx *c, const char **names, double *total_ms, int64_t *launches, int cap) {
    if (!c) return PAFFY_E_ARG;
    prof_collect(c);
    int n = (int)c->prof.size();
    for (int i = 0; i < n && i < cap; i++) {
        if (names) names[i] = c->prof[i].name.c_str();
        if (total_ms) total_ms[i] = c->prof[i].ms;
        if (launches) launches[i] = c->prof[i].launches;
    }
    return n;
}

int paffy_hip_malloc(void **p, int64_t bytes) { return hipMalloc(p, (size_t)bytes + 64) == hipSuccess ? 0 : PAFFY_E_HIP; }
int paffy_hip_free(void *p) { return hipFree(p) == hipSuccess ? 0 : PAFFY_E_HIP; }
int paffy_hip_memcpy_h2d(void *d, const void *h, int64_t n) { return hipMemcpy(d, h, (size_t)n, hipMemcpyHostToDevice) == hipSuccess ? 0 : PAFFY_E_HIP; }
int paffy_hip_memcpy_d2h(void *h, const void *d, int64_t n) { return hipMemcpy(h, d, (size_t)n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : PAFFY_E_HIP; }
int paffy_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int paffy_hip_synth_contigs(paffy_hip_ctx *c, uint64_t seed, uint32_t mean_ops, uint32_t n_contigs, uint64_t r0, uint64_t n, void *d_out, int64_t out_cap,
                            int64_t *bytes);
int paffy_hip_synth(paffy_hip_ctx *c, uint64_t seed, uint32_t mean_ops, uint64_t r0, uint64_t n, void *d_out, int64_t out_cap,
                    int64_t *bytes) {
    return paffy_hip_synth_contigs(c, seed, mean_ops, 24, r0, n, d_out, out_cap, bytes);
}
int paffy_hip_synth_contigs(paffy_hip_ctx *c, uint64_t seed, uint32_t mean_ops, uint32_t n_contigs, uint64_t r0, uint64_t n, void *d_out, int64_t out_cap,
                            int64_t *bytes) {
    if (!c || !bytes || n >= (1ull << 31) || n_contigs < 1) return PAFFY_E_ARG;
    *bytes = 0;
    if (n == 0) return 0;
    psynth_cfg cfg;
    cfg.seed = seed;
    cfg.mean_ops = mean_ops;
    cfg.n_contigs = n_contigs;
    if (ensure(c, c->synth_sizes, sizeof(int64_t) * (size_t)(2 * n + 2))) return PAFFY_E_HIP;
    int64_t *sizes = static_cast<int64_t *>(c->synth_sizes.p), *offs = sizes + n, *total = offs + n;
    const uint32_t nn = (uint32_t)n, grid = (nn + PAFFY_NT - 1) / PAFFY_NT;
    LAUNCH(c, "k_synth_size", k_synth_size, dim3(grid), dim3(PAFFY_NT), 0, cfg, r0, nn, sizes);
    LAUNCH(c, "k_scan_i64", k_scan_i64, dim3(1), dim3(PAFFY_NT), 0, sizes, nn, offs, total);
    int64_t h_total = 0;
    HIPCHK(c, hipMemcpyAsync(&h_total, total, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *bytes = h_total;
    if (!d_out) return 0;
    if (out_cap < h_total) return PAFFY_E_CAPACITY;
    LAUNCH(c, "k_synth_fill", k_synth_fill, dim3(grid), dim3(PAFFY_NT), 0, cfg, r0, nn, offs, static_cast<uint8_t *>(d_out));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

static psynth4_tab synth4_tab(paffy_hip_ctx *c) {
    psynth4_tab t;
    t.contigs = static_cast<const psynth4_contig *>(c->synth4_contigs.p);
    t.ckpt_q = static_cast<const int64_t *>(c->synth4_q.p);
    t.ckpt_t = static_cast<const int64_t *>(c->synth4_t.p);
    return t;
}

int paffy_hip_synth4_setup(paffy_hip_ctx *c, uint64_t seed, uint32_t mean_ops, uint32_t n_contigs, int64_t tlen_min, int64_t tlen_span,
                           int with_genomes) {
    if (!c || n_contigs < 1 || n_contigs > 4096 || tlen_min < 2048 || tlen_span < 0) return PAFFY_E_ARG;
    psynth4_cfg cfg;
    cfg.seed = seed;
    cfg.mean_ops = mean_ops;
    cfg.n_contigs = n_contigs;
    cfg.tlen_min = tlen_min;
    cfg.tlen_span = tlen_span;
    c->synth4_cfg.n_contigs = 0;
    std::vector<psynth4_contig> ct(n_contigs);
    uint64_t total = 0;
    for (uint32_t k = 0; k < n_contigs; k++) {
        memset(&ct[k], 0, sizeof(psynth4_contig));
        ct[k].tlen = psynth4_tlen(&cfg, k);
        ct[k].ckpt_cap = psynth4_ckpt_cap(ct[k].tlen);
        ct[k].ckpt_base = total;
        total += ct[k].ckpt_cap;
    }
    if (ensure(c, c->synth4_contigs, sizeof(psynth4_contig) * n_contigs)) return PAFFY_E_HIP;
    if (ensure(c, c->synth4_q, sizeof(int64_t) * total)) return PAFFY_E_HIP;
    if (ensure(c, c->synth4_t, sizeof(int64_t) * total)) return PAFFY_E_HIP;
    HIPCHK(c, hipMemcpyAsync(c->synth4_contigs.p, ct.data(), sizeof(psynth4_contig) * n_contigs, hipMemcpyHostToDevice, c->stream));
    LAUNCH(c, "k_synth4_master", k_synth4_master, dim3(n_contigs), dim3(64), 0, cfg, static_cast<psynth4_contig *>(c->synth4_contigs.p),
           static_cast<int64_t *>(c->synth4_q.p), static_cast<int64_t *>(c->synth4_t.p));
    HIPCHK(c, hipMemcpyAsync(ct.data(), c->synth4_contigs.p, sizeof(psynth4_contig) * n_contigs, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->synth4_cfg = cfg;
    if (!with_genomes) return 0;
    std::vector<std::string> names;
    std::vector<int64_t> lens;
    for (int g = 0; g < 2; g++)
        for (uint32_t k = 0; k < n_contigs; k++) {
            names.push_back(std::string(g ? "pt.chr" : "hs.chr") + std::to_string(k + 1));
            lens.push_back(g ? ct[k].tlen : ct[k].qlen);
        }
    std::vector<const char *> name_ptr;
    for (const std::string &s : names) name_ptr.push_back(s.c_str());
    std::vector<uint64_t> blob_off;
    int rc = seq_store_layout(c, (int64_t)names.size(), name_ptr.data(), lens.data(), blob_off);
    if (rc) return rc;
    uint8_t *blob = static_cast<uint8_t *>(c->seq_blob.p);
    const psynth4_tab tab = synth4_tab(c);
    for (uint32_t k = 0; k < n_contigs; k++) {
        const uint32_t q_blocks = (uint32_t)((ct[k].n_ops + PSYNTH4_G - 1) / PSYNTH4_G);
        LAUNCH(c, "k_synth4_query", k_synth4_query, dim3(q_blocks), dim3(64), 0, cfg, k, tab, blob + blob_off[k]);
        const uint32_t t_blocks = (uint32_t)((ct[k].tlen + 32ll * PAFFY_NT - 1) / (32ll * PAFFY_NT));
        LAUNCH(c, "k_synth4_target", k_synth4_target, dim3(t_blocks), dim3(PAFFY_NT), 0, cfg, k, ct[k].tlen, blob + blob_off[n_contigs + k]);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (seq_store_canon(c)) return PAFFY_E_HIP;
    c->n_seqs = (int32_t)names.size();
    return 0;
}

int paffy_hip_synth4(paffy_hip_ctx *c, uint64_t r0, uint64_t n, void *d_out, int64_t out_cap, int64_t *bytes) {
    if (!c || !bytes || n >= (1ull << 31)) return PAFFY_E_ARG;
    if (c->synth4_cfg.n_contigs == 0) return PAFFY_E_STATE;
    *bytes = 0;
    if (n == 0) return 0;
    const psynth4_cfg cfg = c->synth4_cfg;
    const psynth4_tab tab = synth4_tab(c);
    if (ensure(c, c->synth_sizes, sizeof(int64_t) * (size_t)(2 * n + 2))) return PAFFY_E_HIP;
    int64_t *sizes = static_cast<int64_t *>(c->synth_sizes.p), *offs = sizes + n, *total = offs + n;
    const uint32_t nn = (uint32_t)n, grid = (nn + PAFFY_NT - 1) / PAFFY_NT;
    LAUNCH(c, "k_synth4_size", k_synth4_size, dim3(grid), dim3(PAFFY_NT), 0, cfg, tab, r0, nn, sizes);
    LAUNCH(c, "k_scan_i64", k_scan_i64, dim3(1), dim3(PAFFY_NT), 0, sizes, nn, offs, total);
    int64_t h_total = 0;
    HIPCHK(c, hipMemcpyAsync(&h_total, total, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *bytes = h_total;
    if (!d_out) return 0;
    if (out_cap < h_total) return PAFFY_E_CAPACITY;
    LAUNCH(c, "k_synth4_fill", k_synth4_fill, dim3(grid), dim3(PAFFY_NT), 0, cfg, tab, r0, nn, offs, static_cast<uint8_t *>(d_out));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

} /* extern "C" */
