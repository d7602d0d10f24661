/*
 * flat_kernel.h -- the sizing pass of the lean pipes (invert / identity trim / shatter / pass) without an op store per record (round 4).
 *
 * The record kernels (record_kernel.h) give every record a workgroup that holds the record's ops in LDS: a record costs a cold prologue
 * whatever its length (half of a typical cfg3 record's time), a long record needs a store of its own size (three LDS levels, then an
 * arena in HBM), and the 6 % of the records above 8 192 ops -- 41 % of the ops of the heavy-tailed stream of SURVEY 8d -- run at one or
 * two workgroups per CU. Here the work is cut differently:
 *
 *   k_flat_parse   cigar_parse (impl/paf.c:70-111) over CHUNKS of cigar text, whatever record they belong to: a wave takes four 1 KiB
 *                  tiles of one record's cigar ("pieces": a record's cigar cut at the 1 KiB boundaries of the batch text), finds the op
 *                  letters, converts the numbers in front of them, writes the ops as 2-byte words into the record's mirror (the
 *                  layout the writers already read) and leaves one 32-byte summary per piece: ops, bases by kind, rows and digits of
 *                  the M ops, text offset. Where a chunk's ops start inside its record's mirror comes from the non-digit counts per
 *                  1 KiB tile that the separator index leaves behind (k_sep_index reads those bytes anyway). Any length of record
 *                  is just more chunks; every wave has the same work.
 *   k_flat_size    one wave per record, on the summaries: totals, paf_check (impl/paf.c:427-461), the identity trim
 *                  (impl/paf.c:811-953) with the pieces as the chunks of its pruned searches -- only the pieces that can hold a hit
 *                  are walked, op by op, from the mirror --, the rows and bytes of paf_shatter (impl/paf.c:600-663) or the length of
 *                  the written line (impl/paf.c:317-389) from the summaries of the window that is left.
 *
 * What the flat pass does not take -- numbers of five digits or more or with leading zeros, lengths of 8 192 or more, zero lengths,
 * characters outside MID=X, a failing check, rows whose digit counts change inside the record, pieces longer than the writers' LDS
 * staging, sums beyond 31 bits, any error at all -- it leaves to the record kernels (flat_done[rec] = 0), which run afterwards over
 * exactly those records; errors are reported there, by the code that has always reported them.
 */
#ifndef PAFFY_FLAT_KERNEL_H_
#define PAFFY_FLAT_KERNEL_H_

#define FLAT_TILE 1024u
#define FLAT_TILE_SHIFT 10u
#define FLAT_CHUNK_PIECES 4u /* pieces (1 KiB of cigar text each) per parse work item */
#define FLAT_P_CAP 512u      /* op letters a 1 KiB piece of a regular cigar can hold (two bytes per op at least) */
#define FLAT_NO_CHUNK 0xffffffffu
#define FLAT_F_IRREG 1u      /* the piece holds something the flat pass leaves to the record kernels */
#define FLAT_F_NONPLAIN 2u   /* the piece holds = or X ops (paf_shatter asserts on them, impl/paf.c:649-651) */

/* Summary of one piece as k_flat_parse leaves it; k_flat_size turns the seven sums into inclusive prefix sums over the record's pieces,
   in place, for the records of more than 64 pieces. */
struct PieceSum {
    uint32_t cnt;      /* ops whose letter lies in the piece; as parsed: | flags << 16 */
    uint32_t m, x;     /* bases of M and = ops; of X, I and D ops (the matches / mismatches of impl/paf.c:823-828) */
    uint32_t ins, del; /* bases of I ops, of D ops */
    uint32_t rows;     /* M ops */
    uint32_t extra;    /* digits of the M ops' lengths beyond the first; or the 16-column chunks of the M ops; or the I ops (k_flat_parse's MODE) */
    uint32_t text_end; /* offset from the cigar's first byte just behind the last op letter at or before the end of the piece */
};
static_assert(sizeof(PieceSum) == 32, "two 16-byte stores");

struct FlatParams {
    const uint8_t *in;
    uint32_t in_len;
    const RecMeta *meta;
    /* Places without a scan or an atomic: piece p of record r has summary slot (cg_off >> 10) + r + p, its chunk c the list slot
       (cg_off >> 12) + r + c -- records lie in text order, so a record's first tile is at or behind the last tile of the record before,
       and the "+ r" keeps the slots of two records that share a tile apart. Slots no chunk maps to hold FLAT_NO_CHUNK. */
    const uint32_t *chunk_rec; /* per chunk slot: its record */
    uint32_t n_chunk_slots;
    const uint16_t *nd;        /* per 1 KiB tile of the text: bytes that are not digits */
    PieceSum *sums;
    uint32_t *ops_mirror;
    DevInfo *info;
    uint32_t items_mode; /* the MODE of k_flat_parse: what PieceSum::extra counts (the host picks the instantiation by it) */
};

__device__ __forceinline__ uint32_t nondigit16(const uint4 &v) { /* bit j: byte j of the 16 is not an ASCII digit */
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t nd = nondigit4(w[j]) >> 7;
        nd = (nd | (nd >> 7) | (nd >> 14) | (nd >> 21)) & 0xfu;
        m |= nd << (4 * j);
    }
    return m;
}
/* sum over the wave: six DPP adds leave it in lane 63 (an inclusive scan takes nine instructions) */
__device__ __forceinline__ uint32_t wave_fold_u32(uint32_t x) {
    x += dpp_mov_u32<DPP_ROW_SHR(1), 0xf, 0xf>(x);
    x += dpp_mov_u32<DPP_ROW_SHR(2), 0xf, 0xf>(x);
    x += dpp_mov_u32<DPP_ROW_SHR(4), 0xf, 0xf>(x);
    x += dpp_mov_u32<DPP_ROW_SHR(8), 0xf, 0xf>(x);
    x += dpp_mov_u32<DPP_BCAST15, 0xa, 0xf>(x);
    x += dpp_mov_u32<DPP_BCAST31, 0xc, 0xf>(x);
    return x;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return wave_last_u32(wave_fold_u32(v)); }
/* six sums at once, step by step: the chains are independent, so no DPP instruction waits for the one before it */
__device__ __forceinline__ void wave_sum6_u32(uint32_t (&x)[6]) {
#define FLAT_FOLD_STEP(CTRL, RM)                                                  \
    _Pragma("unroll") for (int k = 0; k < 6; k++) x[k] += dpp_mov_u32<CTRL, RM, 0xf>(x[k]);
    FLAT_FOLD_STEP(DPP_ROW_SHR(1), 0xf)
    FLAT_FOLD_STEP(DPP_ROW_SHR(2), 0xf)
    FLAT_FOLD_STEP(DPP_ROW_SHR(4), 0xf)
    FLAT_FOLD_STEP(DPP_ROW_SHR(8), 0xf)
    FLAT_FOLD_STEP(DPP_BCAST15, 0xa)
    FLAT_FOLD_STEP(DPP_BCAST31, 0xc)
#undef FLAT_FOLD_STEP
#pragma unroll
    for (int k = 0; k < 6; k++) x[k] = wave_last_u32(x[k]);
}
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint32_t lane_val(uint32_t x, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)l); }

/* bytes of LDS per wave: 16 bytes in front of the tile (the text before it), the tile, the positions of its op letters */
#define FLAT_PARSE_LDS (16u + FLAT_TILE + 2u * FLAT_P_CAP)
#define FLAT_PARSE_WAVES 4u
/* MODE = what PieceSum::extra counts: 0 the digits of the M ops' lengths beyond the first (shatter's row bytes), 1 the 16-column chunks of the M
   ops (add_mismatches, flat_add_kernel.h), 2 the I ops (the counts of paf_stats_calc, impl/paf.c:236-260) */
template <uint32_t MODE>
__global__ __launch_bounds__(64 * FLAT_PARSE_WAVES) void k_flat_parse(FlatParams F) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[FLAT_PARSE_WAVES][FLAT_PARSE_LDS];
    /* the wave's number, in a scalar register: everything derived from it (the chunk, its record, the tile addresses) is wave-uniform,
       and the compiler only knows that when it is told */
    const uint32_t lane = threadIdx.x & 63u, wave = uni(threadIdx.x >> 6);
    uint8_t *T = smem[wave];
    uint16_t *Pl = reinterpret_cast<uint16_t *>(T + 16u + FLAT_TILE);
    const uint32_t n_waves = gridDim.x * FLAT_PARSE_WAVES;
    for (uint32_t c = blockIdx.x * FLAT_PARSE_WAVES + wave; c < F.n_chunk_slots; c += n_waves) {
        const uint32_t rec = F.chunk_rec[c];
        if (rec == FLAT_NO_CHUNK) continue;
        const uint32_t cg_off = F.meta[rec].cg_off, cg_end = cg_off + F.meta[rec].cg_len;
        const uint32_t tile_first = cg_off >> FLAT_TILE_SHIFT;
        const uint32_t np = ((cg_end - 1u) >> FLAT_TILE_SHIFT) - tile_first + 1u;
        const uint32_t p0 = (c - ((tile_first >> 2) + rec)) * FLAT_CHUNK_PIECES, p1 = p0 + FLAT_CHUNK_PIECES < np ? p0 + FLAT_CHUNK_PIECES : np;
        PieceSum *const rec_sums = F.sums + (tile_first + rec);
        uint32_t tile = tile_first + p0;
        /* the 16 bytes in front of the chunk's first tile */
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            uint4 h = make_uint4(0, 0, 0, 0);
            if (tile > 0) h = *reinterpret_cast<const uint4 *>(F.in + ((size_t)tile << FLAT_TILE_SHIFT) - 16u);
            *reinterpret_cast<uint4 *>(T) = h;
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t n = 0;        /* ops of the record in front of the current tile */
        int32_t prev;          /* the last op letter in front of the tile, relative to the tile's first byte (negative) */
        uint32_t text_end = 0; /* offset from cg_off just behind that letter */
        if (p0 == 0) {
            prev = (int32_t)cg_off - 1 - (int32_t)(tile << FLAT_TILE_SHIFT);
        } else {
            /* ops in front of the chunk: the letters of the record's first, partial tile and the non-digit bytes of the whole tiles
               between it and the chunk (inside a cigar the flat pass keeps, every byte that is not a digit is an op letter) */
            uint32_t cnt = 0;
            {
                const uint32_t g = (tile_first << FLAT_TILE_SHIFT) + lane * 16u;
                uint4 v = make_uint4(0x30303030u, 0x30303030u, 0x30303030u, 0x30303030u);
                if (g < F.in_len) v = *reinterpret_cast<const uint4 *>(F.in + g);
                const uint32_t lo_b = cg_off > g ? (cg_off - g < 16u ? cg_off - g : 16u) : 0u;
                cnt = (uint32_t)__popc(nondigit16(v) & 0xffffu & ~((1u << lo_b) - 1u));
            }
            for (uint32_t t = tile_first + 1u + lane; t < tile; t += 64u) cnt += F.nd[t];
            n = wave_sum_u32(cnt);
            const uint4 h = *reinterpret_cast<const uint4 *>(T);
            const uint32_t hm = nondigit16(h);
            prev = hm ? (int32_t)(31 - __clz((int)hm)) - 16 : -17; /* no letter within 16 bytes: the first number of the chunk counts as too long */
            text_end = (uint32_t)((int32_t)(tile << FLAT_TILE_SHIFT) + prev + 1 - (int32_t)cg_off);
        }
        uint16_t *dst = reinterpret_cast<uint16_t *>(F.ops_mirror + (cg_off >> 1));
        for (uint32_t p = p0; p < p1; p++, tile++) {
            const uint32_t t0 = tile << FLAT_TILE_SHIFT, g = t0 + lane * 16u;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (g < F.in_len) v = *reinterpret_cast<const uint4 *>(F.in + g);
            *reinterpret_cast<uint4 *>(T + 16u + lane * 16u) = v;
            const uint32_t ndm = nondigit16(v);
            const uint32_t lo_b = cg_off > g ? (cg_off - g < 16u ? cg_off - g : 16u) : 0u;
            const uint32_t hi_b = cg_end < g + 16u ? (cg_end > g ? cg_end - g : 0u) : 16u;
            uint32_t opmask = ndm & (hi_b >= 16u ? 0xffffu : ((1u << hi_b) - 1u)) & ~((1u << lo_b) - 1u);
            /* a cigar that ends in digits: the reference's switch sees the NUL (impl/paf.c:96-103) */
            uint32_t bad = (hi_b > lo_b && g + hi_b == cg_end && !((ndm >> (hi_b - 1u)) & 1u)) ? 1u : 0u;
            const uint32_t cnt = (uint32_t)__popc(opmask), inc = wave_incl_scan_u32(cnt), total = wave_last_u32(inc);
            uint32_t rank = inc - cnt;
            while (opmask) {
                const uint32_t j = (uint32_t)__ffs((int)opmask) - 1u;
                opmask &= opmask - 1u;
                if (rank < FLAT_P_CAP) Pl[rank] = (uint16_t)(lane * 16u + j);
                rank++;
            }
            if (total > FLAT_P_CAP) bad = 1u; /* op letters side by side: lengths without digits */
            __builtin_amdgcn_wave_barrier();
            const uint32_t top = total < FLAT_P_CAP ? total : FLAT_P_CAP;
            /* per lane at most eight ops of at most 8 191 bases: two 16-bit sums per register */
            uint32_t acc_mx = 0, acc_id = 0, acc_re = 0, nonplain = 0;
            uint32_t before_first = (uint32_t)prev; /* the letter in front of the iteration's first op (lane 0 takes it) */
            for (uint32_t i = lane; i < top; i += 64u) {
                const uint32_t pos = Pl[i];
                /* the letter in front: the lane below's (active whenever this one is) */
                const uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp((int)before_first, (int)pos, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                before_first = lane_val(pos, 63);
                const uint32_t k = pos - before - 1u; /* digits in front of the letter */
                /* the four bytes in front of the letter and the letter itself: two aligned words of the staged text */
                const uint32_t a = 12u + pos, sh = a & 3u;
                const uint32_t *wp = reinterpret_cast<const uint32_t *>(T + (a - sh));
                const uint32_t d0 = wp[0], d1 = wp[1];
                const uint32_t hi = __builtin_amdgcn_alignbyte(d1, d0, sh);
                const uint32_t ch = (d1 >> (8u * sh)) & 0xffu;
                const uint32_t kk = k < 4u ? k : 4u;
                const uint32_t x = kk ? ((hi ^ 0x30303030u) & (0xffffffffu << (32u - 8u * kk))) : 0u;
                const uint32_t pr = ((x << 3) + (x << 1) + (x >> 8)) & 0x00ff00ffu; /* swar4 without the 32-bit multiply: two 2-digit numbers */
                const uint32_t len = __umul24(pr & 0xffu, 100u) + (pr >> 16);
                /* letter -> code (impl/paf.c:96-103): '=' 0x3d, 'D' 0x44, 'I' 0x49, 'M' 0x4d, 'X' 0x58 lie 0, 7, 12, 16, 27 above '=' */
                const uint32_t dl = ch - 0x3du;
                const bool known = dl < 28u && ((0x08011081u >> dl) & 1u);
                const uint32_t code = (0x04001023u >> (dl & 0x1cu)) & 7u;
                const bool lead0 = kk >= 2u && ((hi >> (8u * (4u - kk))) & 0xffu) == (uint32_t)'0';
                /* what the flat pass keeps: one to four digits without a leading zero, 1 <= length < 8192, a letter of MID=X */
                bad |= ((k - 1u > 3u) | (len - 1u >= 8191u) | lead0 | !known) ? 1u : 0u;
                dst[n + i] = (uint16_t)((len << 3) | code);
                const uint32_t l16 = len & 0xffffu;
                acc_mx += l16 << (((0x16u >> code) & 1u) << 4);                                     /* M = | X I D */
                acc_id += (code == (uint32_t)OP_I ? l16 : 0u) + (code == (uint32_t)OP_D ? l16 << 16 : 0u); /* I | D */
                if (MODE == 2u) acc_re += (code == (uint32_t)OP_M ? 1u : 0u) + (code == (uint32_t)OP_I ? 1u << 16 : 0u); /* rows | I ops */
                else acc_re += code == (uint32_t)OP_M ? 1u + ((MODE == 1u ? (len + 15u) >> 4 : kk - 1u) << 16) : 0u; /* rows | digits beyond the first (or 16-column chunks) */
                nonplain |= code > (uint32_t)OP_D ? 1u : 0u;
            }
            uint32_t sums6[6] = {acc_mx & 0xffffu, acc_mx >> 16, acc_id & 0xffffu, acc_id >> 16, acc_re & 0xffffu, acc_re >> 16};
            wave_sum6_u32(sums6);
            const uint32_t s_m = sums6[0], s_x = sums6[1], s_i = sums6[2], s_d = sums6[3], s_r = sums6[4], s_e = sums6[5];
            const uint32_t flags = (__any(bad != 0) ? FLAT_F_IRREG : 0u) | (__any(nonplain != 0) ? FLAT_F_NONPLAIN : 0u);
            if (total) {
                const int32_t last = (int32_t)Pl[top - 1u];
                text_end = t0 + (uint32_t)last + 1u - cg_off;
                prev = last - (int32_t)FLAT_TILE;
            } else {
                prev -= (int32_t)FLAT_TILE;
                if (prev < -64) prev = -64;
            }
            if (lane == 0) {
                uint4 *o = reinterpret_cast<uint4 *>(rec_sums + p);
                o[0] = make_uint4(total | (flags << 16), s_m, s_x, s_i);
                o[1] = make_uint4(s_d, s_r, s_e, text_end);
            }
            n += total;
            /* the tile's last 16 bytes are the next tile's front */
            __builtin_amdgcn_wave_barrier();
            if (lane == 63u) *reinterpret_cast<uint4 *>(T) = v;
            __builtin_amdgcn_wave_barrier();
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------------------------ */

struct FlatSizeParams {
    KParams P;
    PieceSum *sums;
    uint8_t *flat_done;
    uint32_t *defer; /* records k_flat_lane hands on to k_flat_size (DevInfo::flat_defer of them) */
};

/* prefix sums at a boundary between raw ops of a record (wave-uniform) */
struct FlatPre {
    uint32_t cnt, m, x, ins, del, rows, extra, text;
};
__device__ __forceinline__ FlatPre flat_sub(const FlatPre &a, const FlatPre &b) {
    FlatPre r;
    r.cnt = a.cnt - b.cnt; r.m = a.m - b.m; r.x = a.x - b.x; r.ins = a.ins - b.ins; r.del = a.del - b.del;
    r.rows = a.rows - b.rows; r.extra = a.extra - b.extra; r.text = a.text - b.text;
    return r;
}
/* a word another lane of this wave has just stored: read past the L1 */
__device__ __forceinline__ uint32_t flat_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

/* One record's pieces: their inclusive prefix sums, lane p = piece p for records of at most 64 pieces, in HBM (scanned in place) for longer ones. */
struct FlatRec {
    const PieceSum *ps;
    uint32_t np, n_ops;
    bool in_regs;
    FlatPre inc; /* in_regs: lane p holds the sums of pieces [0, p] */
    const uint16_t *ops;
    bool want_text;
    bool extra_icount; /* PieceSum::extra counts I ops (a pipe with a stats stage), not digits */

    __device__ __forceinline__ FlatPre piece_prefix(uint32_t j) const { /* pieces [0, j) */
        FlatPre r;
        r.cnt = r.m = r.x = r.ins = r.del = r.rows = r.extra = r.text = 0;
        if (j == 0) return r;
        if (in_regs) {
            const uint32_t l = j - 1u;
            r.cnt = lane_val(inc.cnt, l); r.m = lane_val(inc.m, l); r.x = lane_val(inc.x, l); r.ins = lane_val(inc.ins, l);
            r.del = lane_val(inc.del, l); r.rows = lane_val(inc.rows, l); r.extra = lane_val(inc.extra, l); r.text = lane_val(inc.text, l);
        } else {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(ps + (j - 1u));
            r.cnt = uni(flat_ld(q)); r.m = uni(flat_ld(q + 1)); r.x = uni(flat_ld(q + 2)); r.ins = uni(flat_ld(q + 3));
            r.del = uni(flat_ld(q + 4)); r.rows = uni(flat_ld(q + 5)); r.extra = uni(flat_ld(q + 6)); r.text = uni(flat_ld(q + 7));
        }
        return r;
    }
    /* the piece that holds raw op r (r < n_ops): the number of pieces that end at or before it */
    __device__ __forceinline__ uint32_t piece_of(uint32_t r) const {
        if (in_regs) {
            const uint32_t lane = threadIdx.x & 63u;
            return (uint32_t)__popcll(__ballot(lane < np && inc.cnt <= r));
        }
        uint32_t lo = 0, hi = np; /* first piece whose inclusive count is above r */
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (uni(flat_ld(&ps[mid].cnt)) <= r) lo = mid + 1u;
            else hi = mid;
        }
        return lo;
    }
    /* sums of raw ops [0, r) */
    __device__ __forceinline__ FlatPre raw_prefix(uint32_t r) const {
        if (r >= n_ops) return piece_prefix(np);
        const uint32_t pc = piece_of(r);
        FlatPre a = piece_prefix(pc);
        if (a.cnt == r) return a;
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t sm = 0, sx = 0, si = 0, sd = 0, sr = 0, se = 0, st = 0;
        for (uint32_t i = a.cnt + lane; i < r; i += 64u) {
            const uint32_t w = ops[i], len = w >> 3, code = w & 7u;
            const uint32_t is_m = 0u - ((0x9u >> code) & 1u);
            const uint32_t dg = (len >= 10u) + (len >= 100u) + (len >= 1000u);
            sm += len & is_m;
            sx += len & ~is_m;
            si += code == (uint32_t)OP_I ? len : 0u;
            sd += code == (uint32_t)OP_D ? len : 0u;
            sr += code == (uint32_t)OP_M ? 1u : 0u;
            se += extra_icount ? (code == (uint32_t)OP_I ? 1u : 0u) : (code == (uint32_t)OP_M ? dg : 0u);
            st += dg + 2u;
        }
        uint32_t sums6[6] = {sm, sx, si, sd, sr, se};
        wave_sum6_u32(sums6);
        a.m += sums6[0]; a.x += sums6[1]; a.ins += sums6[2]; a.del += sums6[3];
        a.rows += sums6[4]; a.extra += sums6[5];
        if (want_text) a.text += wave_sum_u32(st);
        a.cnt = r;
        return a;
    }
};

/* The window of raw ops a record's view is, with the prefix sums at its two ends. */
struct FlatView {
    uint32_t lo, n;
    bool rev, swp;
    FlatPre wlo, whi;
    uint32_t cut_m = 0, cut_x = 0; /* bases a fixed trim has cut from the window's end ops (M / = ops, X ops): the ops stay, shortened */
    __device__ __forceinline__ uint32_t tm() const { return whi.m - wlo.m - cut_m; }
    __device__ __forceinline__ uint32_t tx() const { return whi.x - wlo.x - cut_x; }
    __device__ __forceinline__ uint32_t ins_v() const { return swp ? whi.del - wlo.del : whi.ins - wlo.ins; } /* I <-> D under an invert */
    __device__ __forceinline__ uint32_t del_v() const { return swp ? whi.ins - wlo.ins : whi.del - wlo.del; }
    __device__ __forceinline__ int64_t tq() const { return (int64_t)tm() + (int64_t)tx() - (int64_t)del_v(); }
    __device__ __forceinline__ int64_t tt() const { return (int64_t)tm() + (int64_t)tx() - (int64_t)ins_v(); }
};

/* what the transforms touch of a record: the rest of its fields is read again when the output is sized */
struct FlatState {
    int64_t qlen, qs, qe, tlen, ts, te;
    bool same;
};
__device__ __forceinline__ void flat_invert(FlatState &s) { /* paf_invert, impl/paf.c:469-474 */
    int64_t t;
    t = s.qs; s.qs = s.ts; s.ts = t;
    t = s.qe; s.qe = s.te; s.te = t;
    t = s.qlen; s.qlen = s.tlen; s.tlen = t;
}
__device__ __forceinline__ int flat_check(const FlatState &s, const FlatView &v) { /* paf_check, impl/paf.c:427-461 */
    if (s.qs < 0 || s.qs >= s.qlen) return PAFFY_ERR_CHECK_QSTART;
    if (s.qs > s.qe || s.qe > s.qlen) return PAFFY_ERR_CHECK_QEND;
    if (s.ts < 0 || s.ts >= s.tlen) return PAFFY_ERR_CHECK_TSTART;
    if (s.ts > s.te || s.te > s.tlen) return PAFFY_ERR_CHECK_TEND;
    if (v.tq() != s.qe - s.qs) return PAFFY_ERR_CHECK_CIGAR_Q;
    if (v.tt() != s.te - s.ts) return PAFFY_ERR_CHECK_CIGAR_T;
    return 0;
}

/* `paffy filter` on a record's totals (impl/paf_filter.c:120-156, paf_stats_calc impl/paf.c:236-260): true = the record passes the thresholds */
__device__ __forceinline__ bool flat_filter_pass(const KParams &P, const RecMeta &m, const FlatView &v) {
    const int64_t mm = v.tm(), mx = v.tx();
    const int64_t all = mm + mx;
    const int64_t ins = all - v.tt(), del = all - v.tq();
    const double identity = ratio_f32(mm, mm + (mx - ins - del));
    const double identity_with_gaps = ratio_f32(mm, all);
    const int64_t f_as = P.filter.min_alignment_score, f_cs = P.filter.min_chain_score, f_tl = P.filter.max_tile_level;
    const double f_id = P.filter.min_identity, f_idg = P.filter.min_identity_with_gaps;
    return m.score >= f_as && m.chain_score >= f_cs && (f_tl == -1 || m.tile_level <= f_tl) && identity >= f_id && identity_with_gaps >= f_idg;
}
/* a record a filter stage drops: no output, later stages never see it */
__device__ __forceinline__ void flat_dropped(const KParams &P, uint8_t *flat_done, uint32_t rec) {
    RecPlan *dp = static_cast<RecPlan *>(P.rec_plan) + rec;
    flat_done[rec] = 1;
    P.status[rec] = (uint32_t)KLASS_LDS << 16;
    P.err_aux[rec] = 0;
    P.n_ops[rec] = 0;
    P.out_len[rec] = 0;
    P.out_rows[rec] = 0;
    dp->flags = 128u;
    dp->n = 0;
}

/*
 * (double)((float)num / (float)den) < thr and >= idd, thr = (double)thr_f, idd = (double)id_f -- the comparisons of impl/paf.c:832-833 and
 * 886-887 -- decided without the division unless the quotient is within 4e-6 of the threshold: conversions, product and quotient are
 * each off by at most 2^-24 relative, so a numerator below 0.999996 x threshold x denominator (above 1.000004 x) has its rounded quotient
 * strictly below (above) the threshold. 0 / 0 takes the exact path (NaN compares false, as it does in the reference).
 */
/* paf_stats_calc (impl/paf.c:236-260) of the view from its sums: matches (M and =), mismatches (X), inserts, deletes, insert bases, delete
   bases -- the order of the record kernels' rec_stats. The I ops are what PieceSum::extra counts in a pipe with a stats stage; the other
   ops that are not M are the D ops as long as the cigar is plain (the caller leaves = and X to the record kernels); an inverted view
   has them swapped. */
__device__ __forceinline__ void flat_stats(const KParams &P, uint32_t rec, const FlatView &v) {
    const uint32_t n_i = v.whi.extra - v.wlo.extra, n_other = (v.whi.cnt - v.wlo.cnt) - (v.whi.rows - v.wlo.rows);
    const uint32_t n_d = n_other - n_i;
    int64_t *o = P.rec_stats + 6ull * rec;
    o[0] = (int64_t)v.tm();
    o[1] = (int64_t)v.tx() - (int64_t)(v.whi.ins - v.wlo.ins) - (int64_t)(v.whi.del - v.wlo.del);
    o[2] = (int64_t)(v.swp ? n_d : n_i);
    o[3] = (int64_t)(v.swp ? n_i : n_d);
    o[4] = (int64_t)v.ins_v();
    o[5] = (int64_t)v.del_v();
}
__device__ __forceinline__ bool flat_ratio_lt(uint32_t num, uint32_t den, float thr_f, double thr) {
    const float nf = __uint2float_rn(num), df = __uint2float_rn(den);
    const float p = __fmul_rn(thr_f, df);
    if (nf < __fmul_rn(p, 0.999996f)) return true;
    if (nf > __fmul_rn(p, 1.000004f)) return false;
    return (double)__fdiv_rn(nf, df) < thr;
}
__device__ __forceinline__ bool flat_ratio_ge(uint32_t num, uint32_t den, float id_f, double idd) {
    const float nf = __uint2float_rn(num), df = __uint2float_rn(den);
    const float p = __fmul_rn(id_f, df);
    if (nf > __fmul_rn(p, 1.000004f)) return true;
    if (nf < __fmul_rn(p, 0.999996f)) return false;
    return (double)__fdiv_rn(nf, df) >= idd;
}

/*
 * paf_trim_unreliable_prefix + paf_trim_upto (impl/paf.c:842-904) on the front of the view; the same searches as trim_prefix32 of
 * record_kernel.h with the record's pieces as the chunks: a piece whose best-case prefix identity clears the threshold by 1e-5 cannot
 * hold a hit (float32 conversions and the divide are off by less than 2e-7 relative) and is not looked at, the others are walked by
 * the wave, one op per lane, from the mirror.
 */
__device__ __forceinline__ void flat_trim_prefix(const FlatRec &R, FlatState &s, FlatView &v, float thr_f, float id_f, int64_t max_trim) {
    const double thr = (double)thr_f, idd = (double)id_f;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w_end = v.lo + v.n;
    const uint32_t nb = (R.np + 63u) >> 6;
    /* last view index with cumulative <= max_trim and prefix identity < threshold (impl/paf.c:820-838): pieces in descending view order */
    int32_t trim_idx = -1;
    uint32_t hit_m = 0, hit_x = 0;
    for (uint32_t bi = 0; bi < nb && trim_idx < 0; bi++) {
        const uint32_t b = v.rev ? bi : nb - 1u - bi;
        const uint32_t p = b * 64u + lane;
        /* prefix sums at the piece's two ends, clipped to the window */
        uint32_t a_cnt = 0, a_m = 0, a_x = 0, b_cnt = 0, b_m = 0, b_x = 0;
        if (p < R.np) {
            if (R.in_regs) {
                b_cnt = R.inc.cnt; b_m = R.inc.m; b_x = R.inc.x;
            } else {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(R.ps + p);
                b_cnt = flat_ld(q); b_m = flat_ld(q + 1); b_x = flat_ld(q + 2);
            }
        }
        if (R.in_regs) {
            a_cnt = dpp_mov_u32<0x138, 0xf, 0xf>(b_cnt); a_m = dpp_mov_u32<0x138, 0xf, 0xf>(b_m); a_x = dpp_mov_u32<0x138, 0xf, 0xf>(b_x); /* wave_shr:1, lane 0 receives 0 */
        } else if (p > 0 && p < R.np) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(R.ps + (p - 1u));
            a_cnt = flat_ld(q); a_m = flat_ld(q + 1); a_x = flat_ld(q + 2);
        }
        if (a_cnt < v.lo) { a_cnt = v.lo; a_m = v.wlo.m; a_x = v.wlo.x; }
        if (a_cnt > w_end) { a_cnt = w_end; a_m = v.whi.m; a_x = v.whi.x; }
        if (b_cnt < v.lo) { b_cnt = v.lo; b_m = v.wlo.m; b_x = v.wlo.x; }
        if (b_cnt > w_end) { b_cnt = w_end; b_m = v.whi.m; b_x = v.whi.x; }
        const bool live = p < R.np && b_cnt > a_cnt;
        /* cumulative sums in front of the piece, in view order */
        const uint32_t c_m = v.rev ? v.whi.m - b_m : a_m - v.wlo.m, c_x = v.rev ? v.whi.x - b_x : a_x - v.wlo.x;
        const uint32_t chunk_x = b_x - a_x;
        const bool may_hit = live && !(c_m > 0 && (double)c_m >= thr * 1.00001 * (double)(c_m + c_x + chunk_x)) &&
                             !(max_trim >= 0 && (int64_t)c_m + (int64_t)c_x > max_trim);
        unsigned long long flagged = __ballot(may_hit);
        while (flagged) {
            /* descending view order: the highest piece of a forward view, the lowest of a reversed one */
            const uint32_t t = v.rev ? (uint32_t)__ffsll((long long)flagged) - 1u : 63u - (uint32_t)__clzll((long long)flagged);
            flagged &= ~(1ull << t);
            const uint32_t ra = lane_val(a_cnt, t), rb = lane_val(b_cnt, t);
            uint32_t pm = lane_val(c_m, t), px = lane_val(c_x, t);
            const uint32_t x_end = px + lane_val(chunk_x, t); /* mismatch bases in front of the piece's far end */
            const uint32_t vb = v.rev ? w_end - rb : ra - v.lo, ve = v.rev ? w_end - ra : rb - v.lo; /* view indices of the piece's ops */
            int32_t hit = -1;
            for (uint32_t i0 = vb; i0 < ve; i0 += 64u) {
                /* the piece's bound again for what is left of it: once the prefix identity clears the threshold even with every
                   mismatch still to come, nothing further on in this piece can hit (the first 64 or 128 ops decide most records) */
                if (i0 != vb && pm > 0 && (double)pm >= thr * 1.00001 * (double)(pm + x_end)) break;
                const uint32_t i = i0 + lane;
                uint32_t len = 0, code = 0;
                if (i < ve) {
                    const uint32_t w = R.ops[v.rev ? w_end - 1u - i : v.lo + i];
                    len = w >> 3;
                    code = w & 7u;
                }
                const uint32_t is_m = 0u - ((0x9u >> code) & 1u);
                const uint32_t im = wave_incl_scan_u32(len & is_m), ix = wave_incl_scan_u32(len & ~is_m);
                const uint32_t cm = pm + im, cx = px + ix;
                const bool ok = i < ve && !(max_trim >= 0 && (int64_t)(cm + cx) > max_trim) && flat_ratio_lt(cm, cm + cx, thr_f, thr);
                const unsigned long long hb = __ballot(ok);
                if (hb) {
                    const uint32_t hl = 63u - (uint32_t)__clzll((long long)hb);
                    hit = (int32_t)(i0 + hl);
                    hit_m = lane_val(cm, hl);
                    hit_x = lane_val(cx, hl);
                }
                pm += wave_last_u32(im);
                px += wave_last_u32(ix);
            }
            if (hit >= 0) {
                trim_idx = hit;
                break;
            }
        }
    }
    if (trim_idx < 0) return;
    /* smallest view index <= trim_idx whose suffix [i, trim_idx] has identity >= identity (impl/paf.c:879-890): ops in view order from the front */
    uint32_t best = 0xffffffffu;
    {
        uint32_t pm = 0, px = 0;
        for (uint32_t i0 = 0; i0 <= (uint32_t)trim_idx && best == 0xffffffffu; i0 += 64u) {
            const uint32_t i = i0 + lane;
            uint32_t len = 0, code = 0;
            if (i <= (uint32_t)trim_idx) {
                const uint32_t w = R.ops[v.rev ? w_end - 1u - i : v.lo + i];
                len = w >> 3;
                code = w & 7u;
            }
            const uint32_t is_m = 0u - ((0x9u >> code) & 1u);
            const uint32_t vm = len & is_m, vx = len & ~is_m;
            const uint32_t im = wave_incl_scan_u32(vm), ix = wave_incl_scan_u32(vx);
            const uint32_t sm = hit_m - (pm + im - vm), sx = hit_x - (px + ix - vx); /* sums of [i, trim_idx] */
            const bool ok = i <= (uint32_t)trim_idx && flat_ratio_ge(sm, sm + sx, id_f, idd);
            const unsigned long long hb = __ballot(ok);
            if (hb) best = i0 + (uint32_t)__ffsll((long long)hb) - 1u;
            pm += wave_last_u32(im);
            px += wave_last_u32(ix);
        }
    }
    const uint32_t count = best != 0xffffffffu ? best : (uint32_t)trim_idx + 1u;
    if (count == 0) return;
    /* paf_trim_upto: the coordinates move over the dropped ops (impl/paf.c:842-861) */
    const int64_t old_tt = v.tt(), old_tq = v.tq();
    {
        const FlatPre cut = R.raw_prefix(v.rev ? v.lo + v.n - count : v.lo + count); /* the window's new end */
        if (v.rev) {
            v.whi = cut;
        } else {
            v.wlo = cut;
            v.lo += count;
        }
    }
    v.n -= count;
    const int64_t d_t = old_tt - v.tt(), d_q = old_tq - v.tq();
    s.ts += d_t;
    if (s.same) s.qs += d_q;
    else s.qe -= d_q;
}

/* ------------------------------------------------------------------------------------------------------------------------------ */
/*
 * k_flat_lane: the same sizing with ONE LANE per record, for the records that need little (at most FLAT_LANE_MAX_PIECES pieces, trims
 * that look at a few hundred ops, rows whose coordinates keep their digit counts). What a wave does for a record -- scans, ballots,
 * wave-uniform arithmetic repeated in 64 lanes -- a lane does here for its own record as plain sequential code over the pieces'
 * summaries and a handful of ops, sixty-four records per wave: about a twentieth of the instructions (PMC on the wave kernel: 1 586
 * VALU + 1 346 scalar instructions per record). Anything else -- long records, deep trims, a power of ten inside a coordinate range,
 * whatever the flat pass leaves to the record kernels -- is put on a list for k_flat_size, which sizes it the way described above.
 */
/*
 * The three constant pieces of a shatter record's rows (A = qname \t qlen \t | B = \t strand \t tname \t tlen \t | C = \t mapq tags \tcg:Z:,
 * paf_shatter2 + paf_write, impl/paf.c:600-627, 317-368), at most 48 bytes each, left in the record's 144 bytes of P.row_pieces (3 x 48,
 * zero filled) by the sizing pass. In the row writer this was every wave's prologue -- 64 lanes for thirteen items, 0.16 ms of a cfg3
 * step and 0.09 ms of a cfg2 step --; in the lane kernel ONE lane's few hundred instructions are shared by the 64 records of its wave:
 * the names come with three 16-byte loads each, the text is put together byte by byte in the lane's column of the kernel's LDS (36
 * dwords; the summaries that lived there are no longer needed) and leaves as nine 16-byte stores. RecPlan flag bit 21 tells the row writer that the
 * block is there; for the records of the wave kernel (one in eight on cfg3) it builds the pieces itself as before (row_pieces_lanes).
 */
#define FLAT_F_ROW_PIECES 0x200000u
#define FLAT_F_COPY 0x400000u       /* RecPlan flag bit 22: the line's cigar is a stretch of the input text, written by k_emit_copy (wq[0] = first byte from cg_off, wt[0] = bytes) */
#define FLAT_COPY_MAX (1u << 18)    /* bytes of cigar text one wave copies; longer lines take the writers that split a record */
#define FLAT_ROW_PIECES_BYTES 144u
struct LanePieceSink {
    uint32_t (*S)[256];
    uint32_t tid;
    __device__ __forceinline__ void put(uint32_t piece, uint32_t n, uint32_t c) const { reinterpret_cast<uint8_t *>(&S[12u * piece + (n >> 2)][tid])[n & 3u] = (uint8_t)c; }
    __device__ __forceinline__ uint32_t dec(uint32_t piece, uint32_t n, int64_t val) const {
        uint64_t u = val < 0 ? 0ull - (uint64_t)val : (uint64_t)val;
        if (val < 0) put(piece, n++, '-');
        const uint32_t d = (uint32_t)dec_len((int64_t)(u > 0x7fffffffffffffffull ? 0x7fffffffffffffffull : u)); /* |INT64_MIN|: 19 digits like INT64_MAX */
        if ((u >> 32) == 0) {
            uint32_t w = (uint32_t)u;
            for (uint32_t i = d; i-- > 0;) { put(piece, n + i, '0' + w % 10u); w /= 10u; }
        } else {
            for (uint32_t i = d; i-- > 0;) { put(piece, n + i, '0' + (uint32_t)(u % 10ull)); u /= 10ull; }
        }
        return n + d;
    }
    __device__ __forceinline__ uint32_t tag(uint32_t piece, uint32_t n, char a, char b, char t) const {
        put(piece, n, '\t'); put(piece, n + 1, (uint8_t)a); put(piece, n + 2, (uint8_t)b); put(piece, n + 3, ':'); put(piece, n + 4, (uint8_t)t); put(piece, n + 5, ':');
        return n + 6;
    }
};
__device__ __forceinline__ void lane_row_pieces(uint32_t (*S)[256], uint32_t tid, uint8_t *dst, const RecState &s, const uint8_t *in) { /* lenA, lenB, lenC <= 48 */
    u32x4 qn[3], tn[3];
#pragma unroll
    for (uint32_t j = 0; j < 3; j++) { /* at most 15 bytes beyond a name: the line's later columns */
        qn[j] = u32x4{0, 0, 0, 0};
        tn[j] = u32x4{0, 0, 0, 0};
        if (16u * j < s.qn_len) qn[j] = *reinterpret_cast<const u32x4_unaligned *>(in + s.qn_off + 16u * j);
        if (16u * j < s.tn_len) tn[j] = *reinterpret_cast<const u32x4_unaligned *>(in + s.tn_off + 16u * j);
    }
#pragma unroll
    for (uint32_t j = 0; j < 36; j++) S[j][tid] = 0;
    const uint32_t qd[12] = {qn[0].x, qn[0].y, qn[0].z, qn[0].w, qn[1].x, qn[1].y, qn[1].z, qn[1].w, qn[2].x, qn[2].y, qn[2].z, qn[2].w};
    const uint32_t td[12] = {tn[0].x, tn[0].y, tn[0].z, tn[0].w, tn[1].x, tn[1].y, tn[1].z, tn[1].w, tn[2].x, tn[2].y, tn[2].z, tn[2].w};
    /* whole dwords of the names; what a name's last dword holds beyond its end (at most three bytes) is covered by the tab, digit and tab that follow */
#pragma unroll
    for (uint32_t j = 0; j < 12; j++)
        if (4u * j < s.qn_len) S[j][tid] = qd[j];
    uint32_t prev = ((uint32_t)'\t' | ((uint32_t)(s.same ? '+' : '-') << 8) | ((uint32_t)'\t' << 16)) << 8; /* B = \t strand \t tname ...: the name starts at byte 3 */
#pragma unroll
    for (uint32_t j = 0; j < 12; j++) {
        if (4u * j < s.tn_len + 3u) S[12 + j][tid] = __builtin_amdgcn_alignbyte(td[j], prev, 1);
        prev = td[j];
    }
    const LanePieceSink k{S, tid};
    uint32_t n = s.qn_len;
    k.put(0, n++, '\t');
    n = k.dec(0, n, s.qlen);
    k.put(0, n++, '\t');
    n = 3u + s.tn_len;
    k.put(1, n++, '\t');
    n = k.dec(1, n, s.tlen);
    k.put(1, n++, '\t');
    n = 0;
    k.put(2, n++, '\t');
    n = k.dec(2, n, s.mapq);
    if (s.type != 0 || s.tile_level != -1) { /* impl/paf.c:343-348 */
        n = k.tag(2, n, 't', 'p', 'A');
        k.put(2, n++, s.type ? s.type : (uint8_t)(s.tile_level > 1 ? 'S' : 'P'));
    }
    if (s.score != 2147483647ll) n = k.dec(2, k.tag(2, n, 'A', 'S', 'i'), s.score); /* INT_MAX guard, impl/paf.c:349 */
    if (s.tile_level != -1) n = k.dec(2, k.tag(2, n, 't', 'l', 'i'), s.tile_level);
    if (s.chain_id != -1) n = k.dec(2, k.tag(2, n, 'c', 'n', 'i'), s.chain_id);
    n = k.tag(2, n, 's', '1', 'i'); /* the children carry s1:i:0 (calloc, impl/paf.c:601) */
    k.put(2, n++, '0');
    k.tag(2, n, 'c', 'g', 'Z');
#pragma unroll
    for (uint32_t q = 0; q < 9; q++)
        reinterpret_cast<uint4 *>(dst)[q] = make_uint4(S[4 * q][tid], S[4 * q + 1][tid], S[4 * q + 2][tid], S[4 * q + 3][tid]);
}

#ifndef FLAT_LANE_MAX_PIECES
#define FLAT_LANE_MAX_PIECES 12u
#endif
#ifndef FLAT_LANE_WALK_BUDGET
#define FLAT_LANE_WALK_BUDGET 1024
#endif
/* ops a lane looks at one by one before it hands its record to a wave */

struct LaneRec {
    const PieceSum *ps;
    const uint16_t *ops;
    uint32_t np, n_ops;
    int32_t budget;
#if defined(PAFFY_ABL) && PAFFY_ABL == 52
    uint32_t *abl; /* [4]: blocks skipped / looked at op by op, ops of the second search, ops summed for a cut */
#endif
    bool want_text;
    bool extra_icount;
    /* ops, matches and mismatches of the record's pieces: in LDS, [piece][thread] (read again and again by the trim's searches) */
    uint32_t (*cnt)[256], (*m)[256], (*x)[256];
};
__device__ __forceinline__ FlatPre lane_piece(const PieceSum *ps, uint32_t p) { /* as parsed: cnt carries the flags */
    const uint4 a = reinterpret_cast<const uint4 *>(ps + p)[0], b = reinterpret_cast<const uint4 *>(ps + p)[1];
    FlatPre q;
    q.cnt = a.x; q.m = a.y; q.x = a.z; q.ins = a.w; q.del = b.x; q.rows = b.y; q.extra = b.z; q.text = b.w;
    return q;
}
__device__ __forceinline__ void lane_add_op(FlatPre &a, uint32_t w, bool want_text, bool extra_icount) {
    const uint32_t len = w >> 3, code = w & 7u;
    const uint32_t is_m = 0u - ((0x9u >> code) & 1u);
    const uint32_t dg = (len >= 10u) + (len >= 100u) + (len >= 1000u);
    a.m += len & is_m;
    a.x += len & ~is_m;
    a.ins += code == (uint32_t)OP_I ? len : 0u;
    a.del += code == (uint32_t)OP_D ? len : 0u;
    a.rows += code == (uint32_t)OP_M ? 1u : 0u;
    a.extra += extra_icount ? (code == (uint32_t)OP_I ? 1u : 0u) : (code == (uint32_t)OP_M ? dg : 0u);
    if (want_text) a.text += dg + 2u;
}
/* Eight ops of the view at a time, one 16-byte load (a lane that asked for its ops one by one would wait for a load per op): view
   indices [i, i + 8) -- the raw ops in front of the window's far end for a reversed view, read from the top. Ops past the window's end
   belong to the neighbours in the mirror: loaded, never used. */
struct LaneOps {
    u32x4 w;
    __device__ __forceinline__ void load(const uint16_t *ops, const FlatView &v, uint32_t i) {
        const uint32_t w_end = v.lo + v.n;
        const uint16_t *p = v.rev ? ops + (w_end - i) - 8u : ops + v.lo + i;
        w = *reinterpret_cast<const u32x4_unaligned *>(p);
    }
    __device__ __forceinline__ void sums(uint32_t &m, uint32_t &x) const { /* bases of the eight ops: in M and = ops, in the others */
        /* two ops per word, both halves at once: lengths below 8 192, so four of them add up inside a half. An op is a match (M 0, = 3)
           when bit 2 of its code is clear and bits 1 and 0 agree */
        const uint32_t d[4] = {w.x, w.y, w.z, w.w};
        uint32_t all2 = 0, x2 = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t len2 = (d[k] >> 3) & 0x1fff1fffu, c = d[k] & 0x00070007u;
            const uint32_t nm = ((c ^ (c >> 1)) | (c >> 2)) & 0x00010001u; /* 1 in a half whose op is not a match */
            all2 += len2;
            x2 += len2 & (nm * 0x1fffu); /* the product stays below 2^30: the 24-bit multiply */
        }
        const uint32_t all = (all2 & 0xffffu) + (all2 >> 16);
        x = (x2 & 0xffffu) + (x2 >> 16);
        m = all - x;
    }
    __device__ __forceinline__ uint32_t get(const FlatView &v, uint32_t k) const { /* op i + k, k = 0..7 */
        const uint32_t j = v.rev ? 7u - k : k;
        const uint32_t d = j < 4 ? (j < 2 ? w.x : w.y) : (j < 6 ? w.z : w.w);
        return (j & 1u) ? d >> 16 : d & 0xffffu;
    }
};
/* sums of raw ops [0, r): whole pieces from their summaries, the piece that holds the boundary walked from its nearer end */
__device__ __forceinline__ FlatPre lane_raw_prefix(LaneRec &R, uint32_t r) {
    FlatPre a;
    a.cnt = a.m = a.x = a.ins = a.del = a.rows = a.extra = a.text = 0;
    for (uint32_t p = 0; p < R.np; p++) {
        const FlatPre q = lane_piece(R.ps, p);
        const uint32_t cnt = q.cnt & 0xffffu;
        if (a.cnt + cnt <= r) { /* the whole piece lies in front of the boundary */
            a.cnt += cnt; a.m += q.m; a.x += q.x; a.ins += q.ins; a.del += q.del; a.rows += q.rows; a.extra += q.extra; a.text = q.text;
            continue;
        }
        if (r - a.cnt <= a.cnt + cnt - r) { /* forwards from the piece's first op */
            R.budget -= (int32_t)(r - a.cnt);
#if defined(PAFFY_ABL) && PAFFY_ABL == 52
            R.abl[3] += r - a.cnt;
#endif
            for (uint32_t i = a.cnt; i < r; i++) lane_add_op(a, R.ops[i], R.want_text, R.extra_icount);
        } else { /* backwards from its last op */
            FlatPre b;
            b.cnt = b.m = b.x = b.ins = b.del = b.rows = b.extra = b.text = 0;
            R.budget -= (int32_t)(a.cnt + cnt - r);
#if defined(PAFFY_ABL) && PAFFY_ABL == 52
            R.abl[3] += a.cnt + cnt - r;
#endif
            for (uint32_t i = r; i < a.cnt + cnt; i++) lane_add_op(b, R.ops[i], R.want_text, R.extra_icount);
            a.m += q.m - b.m; a.x += q.x - b.x; a.ins += q.ins - b.ins; a.del += q.del - b.del; a.rows += q.rows - b.rows; a.extra += q.extra - b.extra;
            a.text = q.text - b.text;
        }
        a.cnt = r;
        return a;
    }
    return a; /* r == n_ops */
}

/* flat_trim_prefix for one lane: the pieces in view order, the ones that can hold a hit walked op by op */
__device__ __forceinline__ void lane_trim_prefix(LaneRec &R, FlatState &s, FlatView &v, float thr_f, float id_f, int64_t max_trim, uint32_t tot_m, uint32_t tot_x) {
    const double thr = (double)thr_f, idd = (double)id_f;
    const uint32_t w_end = v.lo + v.n, tid = threadIdx.x;
    int32_t trim_idx = -1;
    uint32_t hit_m = 0, hit_x = 0;
    /* prefix sums (count, matches, mismatches) at the near end of the current piece, in raw order */
    uint32_t e_cnt = v.rev ? R.n_ops : 0u, e_m = v.rev ? tot_m : 0u, e_x = v.rev ? tot_x : 0u;
    for (uint32_t k = 0; k < R.np && R.budget >= 0; k++) {
        const uint32_t p = v.rev ? R.np - 1u - k : k;
        const uint4 q = make_uint4(R.cnt[p][tid], R.m[p][tid], R.x[p][tid], 0);
        const uint32_t cnt = q.x;
        /* the piece's raw range and the prefix sums at its two ends */
        uint32_t a_cnt, a_m, a_x, b_cnt, b_m, b_x;
        if (v.rev) {
            b_cnt = e_cnt; b_m = e_m; b_x = e_x;
            e_cnt -= cnt; e_m -= q.y; e_x -= q.z;
            a_cnt = e_cnt; a_m = e_m; a_x = e_x;
        } else {
            a_cnt = e_cnt; a_m = e_m; a_x = e_x;
            e_cnt += cnt; e_m += q.y; e_x += q.z;
            b_cnt = e_cnt; b_m = e_m; b_x = e_x;
        }
        if (a_cnt < v.lo) { a_cnt = v.lo; a_m = v.wlo.m; a_x = v.wlo.x; }
        if (a_cnt > w_end) { a_cnt = w_end; a_m = v.whi.m; a_x = v.whi.x; }
        if (b_cnt < v.lo) { b_cnt = v.lo; b_m = v.wlo.m; b_x = v.wlo.x; }
        if (b_cnt > w_end) { b_cnt = w_end; b_m = v.whi.m; b_x = v.whi.x; }
        if (b_cnt <= a_cnt) continue;
        uint32_t pm = v.rev ? v.whi.m - b_m : a_m - v.wlo.m, px = v.rev ? v.whi.x - b_x : a_x - v.wlo.x; /* in front of the piece, in view order */
        if (max_trim >= 0 && (int64_t)pm + (int64_t)px > max_trim) break; /* the sums only grow */
        const uint32_t x_end = px + (b_x - a_x);
        if (pm > 0 && (double)pm >= thr * 1.00001 * (double)(pm + x_end)) continue; /* cannot hold a hit (flat_trim_prefix) */
        const uint32_t vb = v.rev ? w_end - b_cnt : a_cnt - v.lo, ve = v.rev ? w_end - a_cnt : b_cnt - v.lo;
        bool stop = false;
        /* a block's load is requested while the block before it is looked at (a lane's sixteen bytes are a cache line of their own).
           One block ahead, taken before the next request: the compiler waits for every outstanding load at the first use of any
           (s_waitcnt vmcnt(0) behind a loop's back edge), so a deeper queue in plain C++ only waits for its newest request */
        LaneOps nxt;
        nxt.load(R.ops, v, vb);
        for (uint32_t i0 = vb; i0 < ve && !stop; i0 += 8u) {
            if (i0 != vb && pm > 0 && (double)pm >= thr * 1.00001 * (double)(pm + x_end)) break;
            const LaneOps blk = nxt;
            if (i0 + 8u < ve) nxt.load(R.ops, v, i0 + 8u);
            R.budget -= 8;
            if (i0 + 8u <= ve && pm > 0) {
                /* a whole block of eight ops: a prefix that ends inside it has at least the matches in front of the block and at most the
                   mismatches in front of it plus the block's, so when even that ratio is not below the threshold (the margin of the
                   piece bound above) none of the eight is a hit and their sums are all the walk needs of them -- an eighth of the
                   instructions of the op-by-op look. Past the first few blocks of an end that is the usual case. */
                uint32_t bm, bx;
                blk.sums(bm, bx);
                if ((double)pm >= thr * 1.00001 * (double)(pm + px + bx) && !(max_trim >= 0 && (int64_t)pm + (int64_t)px + (int64_t)bm + (int64_t)bx > max_trim)) {
                    pm += bm;
                    px += bx;
#if defined(PAFFY_ABL) && PAFFY_ABL == 52
                    R.abl[0]++;
#endif
                    continue;
                }
            }
#if defined(PAFFY_ABL) && PAFFY_ABL == 52
            R.abl[1]++;
#endif
#pragma unroll
            for (uint32_t j = 0; j < 8u; j++) {
                if (i0 + j < ve && !stop) {
                    const uint32_t w = blk.get(v, j), len = w >> 3, code = w & 7u;
                    if ((0x9u >> code) & 1u) pm += len;
                    else px += len;
                    if (max_trim >= 0 && (int64_t)(pm + px) > max_trim) {
                        stop = true;
                    } else if (flat_ratio_lt(pm, pm + px, thr_f, thr)) {
                        trim_idx = (int32_t)(i0 + j);
                        hit_m = pm;
                        hit_x = px;
                    }
                }
            }
        }
        if (stop) break;
    }
    if (trim_idx < 0 || R.budget < 0) return;
    R.budget -= trim_idx + 1;
#if defined(PAFFY_ABL) && PAFFY_ABL == 52
    R.abl[2] += (uint32_t)trim_idx + 1u;
#endif
    if (R.budget < 0) return;
    uint32_t best = 0xffffffffu;
    {
        uint32_t pm = 0, px = 0;
        for (uint32_t i0 = 0; i0 <= (uint32_t)trim_idx && best == 0xffffffffu; i0 += 8u) {
            LaneOps blk;
            blk.load(R.ops, v, i0);
#pragma unroll
            for (uint32_t j = 0; j < 8u; j++) {
                if (i0 + j <= (uint32_t)trim_idx && best == 0xffffffffu) {
                    const uint32_t w = blk.get(v, j), len = w >> 3, code = w & 7u;
                    const uint32_t sm = hit_m - pm, sx = hit_x - px; /* sums of [i, trim_idx] */
                    if (flat_ratio_ge(sm, sm + sx, id_f, idd)) best = i0 + j;
                    if ((0x9u >> code) & 1u) pm += len;
                    else px += len;
                }
            }
        }
    }
    const uint32_t count = best != 0xffffffffu ? best : (uint32_t)trim_idx + 1u;
    if (count == 0) return;
    const int64_t old_tt = v.tt(), old_tq = v.tq();
    {
        const FlatPre cut = lane_raw_prefix(R, v.rev ? v.lo + v.n - count : v.lo + count);
        if (v.rev) {
            v.whi = cut;
        } else {
            v.wlo = cut;
            v.lo += count;
        }
    }
    v.n -= count;
    const int64_t d_t = old_tt - v.tt(), d_q = old_tq - v.tq();
    s.ts += d_t;
    if (s.same) s.qs += d_q;
    else s.qe -= d_q;
}

__global__ __launch_bounds__(256) void k_flat_lane(FlatSizeParams F) {
    __shared__ uint32_t s_all[3 * FLAT_LANE_MAX_PIECES][256]; /* per lane: count / M / X sums of its record's pieces, at the end the row pieces' text */
    static_assert(3 * FLAT_LANE_MAX_PIECES >= 36, "lane_row_pieces() needs 36 dwords per lane");
    uint32_t (*s_cnt)[256] = s_all, (*s_m)[256] = s_all + FLAT_LANE_MAX_PIECES, (*s_x)[256] = s_all + 2 * FLAT_LANE_MAX_PIECES;
    const KParams &P = F.P;
    const uint32_t rec = blockIdx.x * 256u + threadIdx.x, tid = threadIdx.x;
#if defined(PAFFY_ABL) && PAFFY_ABL == 51
    const unsigned long long abl_w0 = wall_clock64();
#endif
    if (rec >= P.n_rec) return;
#if defined(PAFFY_ABL) && PAFFY_ABL == 52 /* what the trim walks of a wave consist of: sums over its lanes in DevInfo::flat_reason[0..3], the longest lane's in [4..7] */
    uint32_t abl_cnt[4] = {0, 0, 0, 0};
#endif
    bool done = false;
    do { /* one pass; `break` = the record goes to k_flat_size */
        const RecMeta &m = P.meta[rec];
        if (m.err || !m.has_cg || m.cg_len == 0 || P.nocheck_mask) break;
        const uint32_t cg_off = m.cg_off, cg_end = cg_off + m.cg_len;
        LaneRec R;
        R.np = ((cg_end - 1u) >> FLAT_TILE_SHIFT) - (cg_off >> FLAT_TILE_SHIFT) + 1u;
        if (R.np > FLAT_LANE_MAX_PIECES) break;
        R.ps = F.sums + ((cg_off >> FLAT_TILE_SHIFT) + rec);
        R.ops = reinterpret_cast<const uint16_t *>(P.ops_mirror + (cg_off >> 1));
        R.budget = FLAT_LANE_WALK_BUDGET;
#if defined(PAFFY_ABL) && PAFFY_ABL == 52
        R.abl = abl_cnt;
#endif
        R.cnt = s_cnt; R.m = s_m; R.x = s_x;
        const bool shatter_last = P.n_stages > 0 && P.stages[P.n_stages - 1].kind == PAFFY_SHATTER;
        R.want_text = !shatter_last;
        R.extra_icount = P.rec_stats != nullptr;
        FlatPre tot;
        tot.cnt = tot.m = tot.x = tot.ins = tot.del = tot.rows = tot.extra = tot.text = 0;
        uint32_t flags = 0;
        for (uint32_t p0 = 0; p0 < R.np; p0 += 4u) { /* four pieces' loads in flight together */
            FlatPre q[4];
#pragma unroll
            for (uint32_t j = 0; j < 4u; j++)
                if (p0 + j < R.np) q[j] = lane_piece(R.ps, p0 + j);
#pragma unroll
            for (uint32_t j = 0; j < 4u; j++)
                if (p0 + j < R.np) {
                    flags |= q[j].cnt >> 16;
                    s_cnt[p0 + j][tid] = q[j].cnt & 0xffffu;
                    s_m[p0 + j][tid] = q[j].m;
                    s_x[p0 + j][tid] = q[j].x;
                    tot.cnt += q[j].cnt & 0xffffu; tot.m += q[j].m; tot.x += q[j].x; tot.ins += q[j].ins; tot.del += q[j].del; tot.rows += q[j].rows;
                    tot.extra += q[j].extra;
                    tot.text = q[j].text;
                }
        }
        R.n_ops = tot.cnt;
        /* at most twelve pieces of 512 ops of 8 191 bases: the sums stay below 2^31 */
        if ((flags & FLAT_F_IRREG) || R.n_ops == 0) break;
        FlatState s;
        s.qlen = m.qlen; s.qs = m.qs; s.qe = m.qe; s.tlen = m.tlen; s.ts = m.ts; s.te = m.te;
        s.same = m.same_strand != 0;
        FlatView v;
        v.lo = 0; v.n = R.n_ops; v.rev = false; v.swp = false;
        v.wlo.cnt = v.wlo.m = v.wlo.x = v.wlo.ins = v.wlo.del = v.wlo.rows = v.wlo.extra = v.wlo.text = 0;
        v.whi = tot;
        bool swapped = false, shatter = false, checked = false, rewritten = false, give_up = false;
        for (int32_t si = 0; si < P.n_stages && !give_up; si++) {
            const paffy_stage st = P.stages[si];
            if (si > 0) {
                if (v.n == 0) { give_up = true; break; }
                rewritten = true;
            }
            int rc = 0;
            if (st.kind == PAFFY_INVERT) {
                flat_invert(s);
                v.swp = !v.swp;
                if (!s.same) v.rev = !v.rev;
                swapped = !swapped;
                rc = flat_check(s, v);
            } else if (st.kind == PAFFY_TRIM_IDENTITY) {
                const uint32_t mm = v.tm(), mx = v.tx();
                const double identity = ratio_f32((int64_t)mm, (int64_t)mm + (int64_t)mx);
                const double thr = __dsub_rn(identity, __dmul_rn(identity, (double)st.p0));
                const int64_t max_trim = __float2ll_rz(__fmul_rn(__ll2float_rn((int64_t)mm + (int64_t)mx), st.p1));
                const float thr_f = __double2float_rn(thr), id_f = __double2float_rn(identity);
                const uint32_t n_before = v.n;
#pragma unroll 1
                for (int pass = 0; pass < 2; pass++) {
                    if (pass == 1) {
                        if (s.same && v.n == n_before) break;
                        flat_invert(s);
                        v.swp = !v.swp;
                        if (!s.same) v.rev = !v.rev;
                    }
                    lane_trim_prefix(R, s, v, thr_f, id_f, max_trim, tot.m, tot.x);
                    if (pass == 1) {
                        flat_invert(s);
                        v.swp = !v.swp;
                        if (!s.same) v.rev = !v.rev;
                    }
                }
                if (R.budget < 0) { give_up = true; break; }
                const uint32_t m2 = v.tm(), x2 = v.tx();
                const double final_identity = ratio_f32((int64_t)m2, (int64_t)m2 + (int64_t)x2);
                if (!(final_identity >= identity)) { give_up = true; break; }
                rc = flat_check(s, v);
            } else if (st.kind == PAFFY_SHATTER) {
                shatter = true;
                break;
            } else if (st.kind == PAFFY_FILTER) {
                if (flat_filter_pass(P, m, v) == (P.filter.invert != 0)) {
                    flat_dropped(P, F.flat_done, rec);
                    return;
                }
            } else if (st.kind == PAFFY_STATS) {
                if (flags & FLAT_F_NONPLAIN) { give_up = true; break; }
                flat_stats(P, rec, v);
            } else if (st.kind != PAFFY_PASS) {
                give_up = true;
                break;
            }
            if (rc) { give_up = true; break; }
            if (st.kind != PAFFY_STATS) checked = st.kind != PAFFY_PASS && st.kind != PAFFY_FILTER;
        }
        if (give_up || v.n == 0) break;
        if (shatter && ((flags & FLAT_F_NONPLAIN) || (!checked && flat_check(s, v)))) break;
        RecState rs;
        load_state(m, rs);
        if (swapped) invert_state(rs);
        rs.qs = s.qs; rs.qe = s.qe; rs.ts = s.ts; rs.te = s.te;
        if (rewritten && rs.type == 0 && rs.tile_level != -1) rs.type = rs.tile_level > 1 ? 'S' : 'P';
        const FlatPre win = flat_sub(v.whi, v.wlo);
        int64_t bytes, rows;
        bool rows_kernel = false, line_kernel = false, copy = false;
        if (shatter) {
            ShatterConst k;
            shatter_consts(rs, k);
            const uint32_t dq0 = (uint32_t)dec_len(rs.qs), dt0 = (uint32_t)dec_len(rs.ts);
            if (!shatter_fits(k) || !shatter_fast_ok(rs, k) || k.lenA > 48 || k.lenB > 48 || k.lenC > 48 || dq0 != (uint32_t)dec_len(rs.qe) ||
                dt0 != (uint32_t)dec_len(rs.te) || rs.qe - rs.qs >= 0x7fffffffll || rs.te - rs.ts >= 0x7fffffffll || v.n > PAFFY_ROWS_MAX_OPS)
                break;
            rows = win.rows;
            bytes = (int64_t)win.rows * (int64_t)(k.row_const + 2u * dq0 + 2u * dt0 + 3u) + 3ll * (int64_t)win.extra;
            rows_kernel = true;
        } else {
            const uint32_t lenH = header_len(rs, false);
            if (lenH + 8 > PAFFY_TMPL_MAX) break;
            /* a window that is not reversed is a stretch of the record's own cigar text (what the flat pass keeps is text as paf_write
               would print it: no leading zeros, lengths below 8 192, letters MID=X): the copy writer takes the line */
            copy = !v.rev && win.text <= FLAT_COPY_MAX;
            if (!copy && v.n > PAFFY_ROWS_MAX_OPS) break;
            line_kernel = !copy;
            bytes = (int64_t)lenH + (int64_t)win.text + 1;
            rows = 1;
        }
        RecPlan *plan = static_cast<RecPlan *>(P.rec_plan) + rec;
        F.flat_done[rec] = 1;
        P.status[rec] = (uint32_t)KLASS_LDS << 16;
        P.err_aux[rec] = 0;
        P.n_ops[rec] = 0;
        P.out_len[rec] = bytes;
        P.out_rows[rec] = rows;
        plan->qs = s.qs; plan->qe = s.qe; plan->ts = s.ts; plan->te = s.te; plan->sub_lo = 0; plan->sub_hi = 0;
        plan->lo = v.lo; plan->n = v.n;
        const bool pieces = rows_kernel && P.row_pieces != nullptr;
        plan->flags = (v.rev ? 1u : 0u) | (v.swp ? 2u : 0u) | (swapped ? 4u : 0u) | 8u | ((uint32_t)rs.type << 8) | (shatter ? 16u : 0u) |
                      (rows_kernel ? 64u : 0u) | (line_kernel ? 0x10000u : 0u) | 0x40000u | (pieces ? FLAT_F_ROW_PIECES : 0u) | (copy ? FLAT_F_COPY : 0u);
        plan->chunk = ((v.n + 63u) / 64u) | 1u;
        for (int w = 0; w < 4; w++) plan->wq[w] = plan->wt[w] = plan->wo[w] = 0;
        if (copy) { /* the stretch of text: from this byte of the cigar, this many */
            plan->wq[0] = (int64_t)v.wlo.text;
            plan->wt[0] = (int64_t)win.text;
        }
        if (pieces) lane_row_pieces(s_all, tid, P.row_pieces + (uint64_t)FLAT_ROW_PIECES_BYTES * rec, rs, P.in);
        done = true;
    } while (false);
    if (!done) F.defer[atomicAdd(&P.info->flat_defer, 1u)] = rec;
#if defined(PAFFY_ABL) && PAFFY_ABL == 52
    for (int k = 0; k < 4; k++) {
        atomicAdd(&P.info->flat_reason[k], abl_cnt[k]);
        uint32_t mx = abl_cnt[k];
        for (int d = 32; d; d >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)mx, d);
            mx = o > mx ? o : mx;
        }
        if ((threadIdx.x & 63u) == 0) atomicAdd(&P.info->flat_reason[4 + k], mx);
    }
#endif
#if defined(PAFFY_ABL) && PAFFY_ABL == 51 /* how long the waves of the lane kernel live: histogram of log2(ticks of the 100 MHz wall clock) in DevInfo::flat_reason */
    {
        const unsigned long long dt = wall_clock64() - abl_w0;
        const uint32_t b = dt ? 63u - (uint32_t)__clzll((long long)dt) : 0u;
        if ((threadIdx.x & 63u) == 0) atomicAdd(&P.info->flat_reason[b < 15u ? b : 15u], 1u);
    }
#endif
}


/*
 * Rows whose coordinates gain a digit inside the record. A row's bytes are a constant plus the digits of its two query and two target
 * coordinates plus three times those of its length (paf_write_to_buffer, impl/paf.c:317-389, for the children of impl/paf.c:600-627); along
 * the view the target coordinates only grow and the query coordinates only grow ('+') or only shrink ('-'), so the rows whose coordinate
 * is at or above a power of ten B are a suffix or a prefix of the view's rows, and where it starts follows from the first op whose
 * cumulative bases reach B's distance from the record's start. flat_find() finds that op: the piece by the summaries, the op by a walk
 * of the piece. *rows_e = M ops of the view in front of the first view index E whose EXCLUSIVE cumulative bases are >= X (X > 0; all
 * the view's M ops when there is none), *m_last = 1 when op E - 1 (the op whose inclusive sum reached X) is an M op.
 */
__device__ __forceinline__ void flat_find(const FlatRec &R, const FlatView &v, bool query, uint32_t X, uint32_t &rows_e, uint32_t &m_last) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w_end = v.lo + v.n;
    const uint32_t nb = (R.np + 63u) >> 6;
    /* the raw op kind that does not advance the coordinate: the view's D for the query, its I for the target (I <-> D under an invert) */
    const bool skip_ins = query ? v.swp : !v.swp;
    const uint32_t skip_code = skip_ins ? (uint32_t)OP_I : (uint32_t)OP_D;
    const uint32_t lo_base = v.wlo.m + v.wlo.x - (skip_ins ? v.wlo.ins : v.wlo.del), hi_base = v.whi.m + v.whi.x - (skip_ins ? v.whi.ins : v.whi.del);
    rows_e = v.whi.rows - v.wlo.rows;
    m_last = 0;
    for (uint32_t bi = 0; bi < nb; bi++) {
        const uint32_t b = v.rev ? nb - 1u - bi : bi; /* pieces in view order */
        const uint32_t p = b * 64u + lane;
        uint32_t a_cnt = 0, a_base = 0, a_rows = 0, b_cnt = 0, b_base = 0, b_rows = 0;
        if (p < R.np) {
            if (R.in_regs) {
                b_cnt = R.inc.cnt; b_base = R.inc.m + R.inc.x - (skip_ins ? R.inc.ins : R.inc.del); b_rows = R.inc.rows;
            } else {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(R.ps + p);
                b_cnt = flat_ld(q); b_base = flat_ld(q + 1) + flat_ld(q + 2) - flat_ld(q + (skip_ins ? 3 : 4)); b_rows = flat_ld(q + 5);
            }
        }
        if (R.in_regs) {
            a_cnt = dpp_mov_u32<0x138, 0xf, 0xf>(b_cnt); a_base = dpp_mov_u32<0x138, 0xf, 0xf>(b_base); a_rows = dpp_mov_u32<0x138, 0xf, 0xf>(b_rows);
        } else if (p > 0 && p < R.np) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(R.ps + (p - 1u));
            a_cnt = flat_ld(q); a_base = flat_ld(q + 1) + flat_ld(q + 2) - flat_ld(q + (skip_ins ? 3 : 4)); a_rows = flat_ld(q + 5);
        }
        if (a_cnt < v.lo) { a_cnt = v.lo; a_base = lo_base; a_rows = v.wlo.rows; }
        if (a_cnt > w_end) { a_cnt = w_end; a_base = hi_base; a_rows = v.whi.rows; }
        if (b_cnt < v.lo) { b_cnt = v.lo; b_base = lo_base; b_rows = v.wlo.rows; }
        if (b_cnt > w_end) { b_cnt = w_end; b_base = hi_base; b_rows = v.whi.rows; }
        const bool live = p < R.np && b_cnt > a_cnt;
        /* cumulative bases at the piece's far end, in view order */
        const uint32_t c_end = v.rev ? hi_base - a_base : b_base - lo_base;
        const unsigned long long crossing = __ballot(live && c_end >= X);
        if (!crossing) continue;
        const uint32_t t = v.rev ? 63u - (uint32_t)__clzll((long long)crossing) : (uint32_t)__ffsll((long long)crossing) - 1u; /* the first in view order */
        const uint32_t ra = lane_val(a_cnt, t), rb = lane_val(b_cnt, t);
        uint32_t cum = v.rev ? hi_base - lane_val(b_base, t) : lane_val(a_base, t) - lo_base;        /* in front of the piece */
        uint32_t rows = v.rev ? v.whi.rows - lane_val(b_rows, t) : lane_val(a_rows, t) - v.wlo.rows;
        const uint32_t vb = v.rev ? w_end - rb : ra - v.lo, ve = v.rev ? w_end - ra : rb - v.lo;
        for (uint32_t i0 = vb; i0 < ve; i0 += 64u) {
            const uint32_t i = i0 + lane;
            uint32_t len = 0, code = skip_code;
            if (i < ve) {
                const uint32_t w = R.ops[v.rev ? w_end - 1u - i : v.lo + i];
                len = w >> 3;
                code = w & 7u;
            }
            const uint32_t inc = wave_incl_scan_u32(code != skip_code ? len : 0u);
            const unsigned long long is_m = __ballot(i < ve && code == (uint32_t)OP_M);
            const unsigned long long hb = __ballot(i < ve && cum + inc >= X);
            if (hb) {
                const uint32_t hl = (uint32_t)__ffsll((long long)hb) - 1u;
                rows_e = rows + (uint32_t)__popcll(is_m & ((2ull << hl) - 1ull));
                m_last = (uint32_t)((is_m >> hl) & 1ull);
                return;
            }
            cum += wave_last_u32(inc);
            rows += (uint32_t)__popcll(is_m);
        }
        return; /* not reached: the piece's far end is at or above X */
    }
}

/*
 * The op a fixed trim stops at (cigar_trim / cigar_trim_back, impl/paf.c:518-576, on the front of the view: the back is the front of the
 * reversed view): ops are popped while the front op is an indel or fewer than `end` aligned bases (M, =, X) are gone, i.e. up to the first
 * aligned op whose inclusive sum of aligned bases exceeds `end` -- that op stays, shortened by what is missing. Found like flat_find():
 * the piece by the summaries, the op by a walk of the piece. The view's LAST op counts `far_cut` bases less (the cut the other end has
 * made already; the summaries do not know it, so the last piece may be walked in vain). Returns false when no op stops the trim (it
 * takes everything); else *idx = the op's view index, *tb = the aligned bases in front of it, *len its length (less far_cut), *code.
 */
__device__ __forceinline__ bool flat_find_aligned(const FlatRec &R, const FlatView &v, uint32_t end, uint32_t far_cut, uint32_t &idx, uint32_t &tb, uint32_t &len_out,
                                                   uint32_t &code_out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w_end = v.lo + v.n;
    const uint32_t nb = (R.np + 63u) >> 6;
    const uint32_t lo_base = v.wlo.m + v.wlo.x - v.wlo.ins - v.wlo.del, hi_base = v.whi.m + v.whi.x - v.whi.ins - v.whi.del;
    for (uint32_t bi = 0; bi < nb; bi++) {
        const uint32_t b = v.rev ? nb - 1u - bi : bi; /* pieces in view order */
        const uint32_t p = b * 64u + lane;
        uint32_t a_cnt = 0, a_base = 0, b_cnt = 0, b_base = 0;
        if (p < R.np) {
            if (R.in_regs) {
                b_cnt = R.inc.cnt; b_base = R.inc.m + R.inc.x - R.inc.ins - R.inc.del;
            } else {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(R.ps + p);
                b_cnt = flat_ld(q); b_base = flat_ld(q + 1) + flat_ld(q + 2) - flat_ld(q + 3) - flat_ld(q + 4);
            }
        }
        if (R.in_regs) {
            a_cnt = dpp_mov_u32<0x138, 0xf, 0xf>(b_cnt); a_base = dpp_mov_u32<0x138, 0xf, 0xf>(b_base);
        } else if (p > 0 && p < R.np) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(R.ps + (p - 1u));
            a_cnt = flat_ld(q); a_base = flat_ld(q + 1) + flat_ld(q + 2) - flat_ld(q + 3) - flat_ld(q + 4);
        }
        if (a_cnt < v.lo) { a_cnt = v.lo; a_base = lo_base; }
        if (a_cnt > w_end) { a_cnt = w_end; a_base = hi_base; }
        if (b_cnt < v.lo) { b_cnt = v.lo; b_base = lo_base; }
        if (b_cnt > w_end) { b_cnt = w_end; b_base = hi_base; }
        const bool live = p < R.np && b_cnt > a_cnt;
        const uint32_t c_end = v.rev ? hi_base - a_base : b_base - lo_base; /* aligned bases through the piece's far end, in view order */
        const unsigned long long crossing = __ballot(live && c_end > end);
        if (!crossing) continue;
        const uint32_t t = v.rev ? 63u - (uint32_t)__clzll((long long)crossing) : (uint32_t)__ffsll((long long)crossing) - 1u; /* the first in view order */
        const uint32_t ra = lane_val(a_cnt, t), rb = lane_val(b_cnt, t);
        uint32_t cum = v.rev ? hi_base - lane_val(b_base, t) : lane_val(a_base, t) - lo_base; /* in front of the piece */
        const uint32_t vb = v.rev ? w_end - rb : ra - v.lo, ve = v.rev ? w_end - ra : rb - v.lo;
        for (uint32_t i0 = vb; i0 < ve; i0 += 64u) {
            const uint32_t i = i0 + lane;
            uint32_t len = 0, code = (uint32_t)OP_I;
            if (i < ve) {
                const uint32_t w = R.ops[v.rev ? w_end - 1u - i : v.lo + i];
                len = (w >> 3) - (i == v.n - 1u ? far_cut : 0u);
                code = w & 7u;
            }
            const bool al = code != (uint32_t)OP_I && code != (uint32_t)OP_D;
            const uint32_t inc = wave_incl_scan_u32(al ? len : 0u);
            const unsigned long long hb = __ballot(i < ve && al && cum + inc > end);
            if (hb) {
                const uint32_t hl = (uint32_t)__ffsll((long long)hb) - 1u;
                idx = i0 + hl;
                len_out = lane_val(len, hl);
                code_out = lane_val(code, hl);
                tb = cum + lane_val(inc, hl) - len_out;
                return true;
            }
            cum += wave_last_u32(inc);
        }
        return false; /* the last piece, its sums too large by far_cut */
    }
    return false;
}

/* what the record kernels take instead: nothing is written for the record but its mark */
enum { FLAT_WHY_HEADER = 0, FLAT_WHY_IRREG = 1, FLAT_WHY_SUMS = 2, FLAT_WHY_EMPTY = 3, FLAT_WHY_CHECK = 4, FLAT_WHY_TRIM_ASSERT = 5, FLAT_WHY_STAGE = 6,
       FLAT_WHY_NONPLAIN = 7, FLAT_WHY_ROW_SHAPE = 8, FLAT_WHY_DIGITS = 9, FLAT_WHY_HEADER_LEN = 10 };
__device__ __forceinline__ void flat_leave(const FlatSizeParams &F, uint32_t rec, int why) {
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(&F.P.info->flat_reason[why], 1u);
        F.flat_done[rec] = 0;
        F.P.out_len[rec] = 0;
        F.P.out_rows[rec] = 0;
        F.P.status[rec] = 0;
        atomicAdd(&F.P.info->flat_legacy, 1u);
    }
}

#define FLAT_SIZE_WAVES 4u
#define FLAT_SEG_OPS (PAFFY_ROWS_MAX_OPS / 2u) /* ops of a segment of a long shatter record: one workgroup of k_emit_rows (about a megabyte of rows) */
#define FLAT_MAX_CROSS 6u /* powers of ten inside a record's query and target ranges together; more (coordinates of a few digits) go to the record kernels */
/* digits beyond those of the record's start coordinates in the coordinates of the first ra rows of the view (flat_find) */
__device__ __forceinline__ uint64_t flat_cross_digits(const uint32_t (*cross)[3], uint32_t n_cross, uint32_t ra) {
    uint64_t extra = 0;
    for (uint32_t c = 0; c < n_cross; c++) {
        const uint32_t re = cross[c][0], ml = cross[c][1];
        if (cross[c][2] == 0) { /* growing: rows from the crossing on; the row's end coordinate crosses one op earlier when that op is the row */
            extra += (ra > re ? ra - re : 0u) + (ra > re - ml ? ra - (re - ml) : 0u);
        } else { /* shrinking: rows in front of the crossing */
            extra += (ra < re ? ra : re) + (ra < re - ml ? ra : re - ml);
        }
    }
    return extra;
}
/* digits the end ops of a fixed trim's window lost with the bases cut from them (lengths below 8 192): len / amt [0] the view's first op, [1]
   its last; `one`: they are the same op; `first_only`: the first op's share (the text in front of a later op) */
__device__ __forceinline__ uint32_t flat_cut_digits(const uint32_t *len, const uint32_t *amt, bool one, bool first_only) {
    auto dl = [](uint32_t x) { return 1u + (x >= 10u) + (x >= 100u) + (x >= 1000u); };
    if (one) return dl(len[0]) - dl(len[0] - amt[0] - amt[1]);
    const uint32_t a = dl(len[0]) - dl(len[0] - amt[0]);
    return first_only ? a : a + dl(len[1]) - dl(len[1] - amt[1]);
}
/* FIXED: the instantiation for pipes that end with a fixed trim (`paffy trim -f`); the others do not carry its code */
template <bool FIXED>
__device__ __forceinline__ void flat_size_one(const FlatSizeParams &F, uint32_t rec, uint32_t (*cross)[3]) {
    const KParams &P = F.P;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t n_cross = 0;
    const RecMeta &m = P.meta[rec];
    if (m.err || !m.has_cg || m.cg_len == 0) return flat_leave(F, rec, FLAT_WHY_HEADER);
    const uint32_t cg_off = m.cg_off, cg_end = cg_off + m.cg_len;
    FlatRec R;
    R.np = ((cg_end - 1u) >> FLAT_TILE_SHIFT) - (cg_off >> FLAT_TILE_SHIFT) + 1u;
    PieceSum *const rec_sums = F.sums + ((cg_off >> FLAT_TILE_SHIFT) + rec);
    R.ps = rec_sums;
    R.ops = reinterpret_cast<const uint16_t *>(P.ops_mirror + (cg_off >> 1));
    R.in_regs = R.np <= 64u;
    const bool shatter_last = P.n_stages > 0 && P.stages[P.n_stages - 1].kind == PAFFY_SHATTER;
    R.want_text = !shatter_last;
    R.extra_icount = P.rec_stats != nullptr;
    /* the pieces' sums become inclusive prefix sums: in registers for a record of at most 64 pieces, in place in HBM for a longer one */
    uint32_t flags = 0;
    unsigned long long tot_m = 0, tot_x = 0;
    {
        FlatPre carry;
        carry.cnt = carry.m = carry.x = carry.ins = carry.del = carry.rows = carry.extra = carry.text = 0;
        const uint32_t nb = (R.np + 63u) >> 6;
        for (uint32_t b = 0; b < nb; b++) {
            const uint32_t p = b * 64u + lane;
            uint4 qa = make_uint4(0, 0, 0, 0), qb = make_uint4(0, 0, 0, 0);
            if (p < R.np) {
                qa = reinterpret_cast<const uint4 *>(R.ps + p)[0];
                qb = reinterpret_cast<const uint4 *>(R.ps + p)[1];
            }
            flags |= qa.x >> 16;
            FlatPre inc;
            inc.cnt = carry.cnt + wave_incl_scan_u32(qa.x & 0xffffu);
            inc.m = carry.m + wave_incl_scan_u32(qa.y);
            inc.x = carry.x + wave_incl_scan_u32(qa.z);
            inc.ins = carry.ins + wave_incl_scan_u32(qa.w);
            inc.del = carry.del + wave_incl_scan_u32(qb.x);
            inc.rows = carry.rows + wave_incl_scan_u32(qb.y);
            inc.extra = carry.extra + wave_incl_scan_u32(qb.z);
            inc.text = qb.w; /* cumulative as parsed */
            tot_m += wave_sum_u32(qa.y); /* a block's own sums stay below 2^32 (64 pieces of 512 ops of 8 191 bases) */
            tot_x += wave_sum_u32(qa.z);
            if (R.in_regs) {
                R.inc = inc;
            } else if (p < R.np) {
                uint4 *o = reinterpret_cast<uint4 *>(rec_sums + p);
                o[0] = make_uint4(inc.cnt, inc.m, inc.x, inc.ins);
                o[1] = make_uint4(inc.del, inc.rows, inc.extra, inc.text);
            }
            carry.cnt = lane_val(inc.cnt, 63); carry.m = lane_val(inc.m, 63); carry.x = lane_val(inc.x, 63); carry.ins = lane_val(inc.ins, 63);
            carry.del = lane_val(inc.del, 63); carry.rows = lane_val(inc.rows, 63); carry.extra = lane_val(inc.extra, 63);
        }
        R.n_ops = carry.cnt;
    }
    flags = (__any((flags & FLAT_F_IRREG) != 0) ? FLAT_F_IRREG : 0u) | (__any((flags & FLAT_F_NONPLAIN) != 0) ? FLAT_F_NONPLAIN : 0u);
    if ((flags & FLAT_F_IRREG) || R.n_ops == 0) return flat_leave(F, rec, FLAT_WHY_IRREG);
    if (tot_m + tot_x >= 0x7fffffffull || P.nocheck_mask) return flat_leave(F, rec, FLAT_WHY_SUMS);
    if (!R.in_regs) __threadfence(); /* the scanned sums are read back below by this wave (other lanes' stores): past the L1, flat_ld() */
    FlatState s;
    s.qlen = m.qlen; s.qs = m.qs; s.qe = m.qe; s.tlen = m.tlen; s.ts = m.ts; s.te = m.te;
    s.same = m.same_strand != 0;
    FlatView v;
    v.lo = 0; v.n = R.n_ops; v.rev = false; v.swp = false;
    v.wlo = R.piece_prefix(0);
    v.whi = R.piece_prefix(R.np);
    bool swapped = false, shatter = false, checked = false, rewritten = false;
    uint32_t fx_amt[2] = {0, 0}, fx_len[2] = {1, 1}; /* fixed trim: bases cut from the op it stops at and that op's length, front / back of the view */
    uint32_t sub_lo = 0, sub_hi = 0;                 /* ... as the writers want them: cut from the window's raw first / last op */
    for (int32_t si = 0; si < P.n_stages; si++) {
        const paffy_stage st = P.stages[si];
        if (si > 0) { /* what `paf_write | paf_parse` between two processes does to the record */
            if (v.n == 0) return flat_leave(F, rec, FLAT_WHY_EMPTY);
            rewritten = true; /* type: below, where the other fields are read */
        }
        int rc = 0;
        if (st.kind == PAFFY_INVERT) {
            flat_invert(s);
            v.swp = !v.swp;
            if (!s.same) v.rev = !v.rev;
            swapped = !swapped;
            rc = flat_check(s, v);
        } else if (FIXED && st.kind == PAFFY_TRIM_FIXED) {
            /* paf_trim_end_fraction + paf_trim_ends (impl/paf.c:578-598), the pipe's last stage (the host sees to that): half of the fraction of
               the aligned bases goes at either end. The ops in front of the op a cut stops at leave the window as in the identity trim; the op
               itself stays, shorter (FlatView::cut_m / cut_x for the sums, sub_lo / sub_hi for the writers) */
            const float pct = st.p1;
            if (!(pct >= 0.0f && pct <= 1.0f)) return flat_leave(F, rec, FLAT_WHY_STAGE); /* the record kernels report the assert */
            const int64_t aligned = (int64_t)v.tm() + (int64_t)v.tx() - (int64_t)(v.whi.ins - v.wlo.ins) - (int64_t)(v.whi.del - v.wlo.del);
            const uint32_t end = (uint32_t)__double2ll_rz((double)__fmul_rn(__ll2float_rn(aligned), pct) / 2.0);
#pragma unroll 1
            for (int pass = 0; pass < 2; pass++) {
                if (pass == 1) v.rev = !v.rev; /* the back is the front of the reversed view */
                uint32_t idx = 0, tb = 0, len = 1, code = 0;
                if (!flat_find_aligned(R, v, end, pass == 1 ? fx_amt[0] : 0u, idx, tb, len, code)) return flat_leave(F, rec, FLAT_WHY_EMPTY); /* nothing is left */
                const uint32_t amt = tb < end ? end - tb : 0u;
                const int64_t old_tt = v.tt(), old_tq = v.tq();
                if (idx) {
                    const FlatPre cut = R.raw_prefix(v.rev ? v.lo + v.n - idx : v.lo + idx);
                    if (v.rev) {
                        v.whi = cut;
                    } else {
                        v.wlo = cut;
                        v.lo += idx;
                    }
                    v.n -= idx;
                }
                if (code == (uint32_t)OP_X) v.cut_x += amt;
                else v.cut_m += amt;
                if (v.rev) sub_hi += amt;
                else sub_lo += amt;
                const int64_t d_t = old_tt - v.tt(), d_q = old_tq - v.tq();
                if (pass == 0) {
                    s.ts += d_t;
                    if (s.same) s.qs += d_q;
                    else s.qe -= d_q;
                } else {
                    s.te -= d_t;
                    if (s.same) s.qe -= d_q;
                    else s.qs += d_q;
                }
                fx_amt[pass] = amt;
                fx_len[pass] = len;
            }
            v.rev = !v.rev;
            rc = flat_check(s, v);
        } else if (st.kind == PAFFY_TRIM_IDENTITY) { /* paf_trim_unreliable_tails, impl/paf.c:906-953 */
            const uint32_t mm = v.tm(), mx = v.tx();
            const double identity = ratio_f32((int64_t)mm, (int64_t)mm + (int64_t)mx);
            const double thr = __dsub_rn(identity, __dmul_rn(identity, (double)st.p0));
            const int64_t max_trim = __float2ll_rz(__fmul_rn(__ll2float_rn((int64_t)mm + (int64_t)mx), st.p1));
            const float thr_f = __double2float_rn(thr), id_f = __double2float_rn(identity);
            const uint32_t n_before = v.n;
#pragma unroll 1
            for (int pass = 0; pass < 2; pass++) {
                if (pass == 1) {
                    if (s.same && v.n == n_before) break;
                    flat_invert(s);
                    v.swp = !v.swp;
                    if (!s.same) v.rev = !v.rev;
                }
                flat_trim_prefix(R, s, v, thr_f, id_f, max_trim);
                if (pass == 1) {
                    flat_invert(s);
                    v.swp = !v.swp;
                    if (!s.same) v.rev = !v.rev;
                }
            }
            const uint32_t m2 = v.tm(), x2 = v.tx();
            const double final_identity = ratio_f32((int64_t)m2, (int64_t)m2 + (int64_t)x2);
            if (!(final_identity >= identity)) return flat_leave(F, rec, FLAT_WHY_TRIM_ASSERT);
            rc = flat_check(s, v);
        } else if (st.kind == PAFFY_SHATTER) {
            shatter = true;
            break;
        } else if (st.kind == PAFFY_FILTER) {
            if (flat_filter_pass(P, m, v) == (P.filter.invert != 0)) {
                if (lane == 0) flat_dropped(P, F.flat_done, rec);
                return;
            }
        } else if (st.kind == PAFFY_STATS) {
            if (flags & FLAT_F_NONPLAIN) return flat_leave(F, rec, FLAT_WHY_NONPLAIN);
            if (lane == 0) flat_stats(P, rec, v);
        } else if (st.kind != PAFFY_PASS) {
            return flat_leave(F, rec, FLAT_WHY_STAGE);
        }
        if (rc) return flat_leave(F, rec, FLAT_WHY_CHECK);
        if (st.kind != PAFFY_STATS) checked = st.kind != PAFFY_PASS && st.kind != PAFFY_FILTER;
    }
    RecPlan *plan = static_cast<RecPlan *>(P.rec_plan) + rec;
    int64_t bytes, rows;
    bool rows_kernel = false, line_kernel = false, copy = false;
    uint32_t row_bytes1 = 0; /* shatter: bytes of a row whose length has one digit */
    const FlatPre win = flat_sub(v.whi, v.wlo);
    if (shatter && !checked && flat_check(s, v)) return flat_leave(F, rec, FLAT_WHY_CHECK); /* every row of a record that passes paf_check passes its own (impl/paf.c:624) */
    /* the record as the writers will see it */
    RecState rs;
    load_state(m, rs);
    if (swapped) invert_state(rs);
    rs.qs = s.qs; rs.qe = s.qe; rs.ts = s.ts; rs.te = s.te;
    if (rewritten && rs.type == 0 && rs.tile_level != -1) rs.type = rs.tile_level > 1 ? 'S' : 'P';
    if (shatter) {
        if (v.n == 0 || (flags & FLAT_F_NONPLAIN)) return flat_leave(F, rec, FLAT_WHY_NONPLAIN);
        ShatterConst k;
        shatter_consts(rs, k);
        const uint32_t dq0 = (uint32_t)dec_len(rs.qs), dt0 = (uint32_t)dec_len(rs.ts);
        if (!shatter_fits(k) || !shatter_fast_ok(rs, k) || k.lenA > 48 || k.lenB > 48 || k.lenC > 48) return flat_leave(F, rec, FLAT_WHY_ROW_SHAPE);
        if (rs.qe - rs.qs >= 0x7fffffffll || rs.te - rs.ts >= 0x7fffffffll) return flat_leave(F, rec, FLAT_WHY_DIGITS);
        /* a row whose coordinates have the digits of the record's start coordinates and whose length has one */
        row_bytes1 = k.row_const + 2u * dq0 + 2u * dt0 + 3u;
        rows = win.rows;
        bytes = (int64_t)win.rows * (int64_t)row_bytes1 + 3ll * (int64_t)win.extra;
        /* the powers of ten inside the record's two ranges (almost always none): the rows at or above one have a digit more */
        if (dq0 != (uint32_t)dec_len(rs.qe) || dt0 != (uint32_t)dec_len(rs.te)) {
#pragma unroll 1
            for (uint32_t f = 0; f < 2; f++) { /* target, query */
                const int64_t c_lo = f ? rs.qs : rs.ts, c_hi = f ? rs.qe : rs.te;
                const bool falling = f == 1 && !rs.same; /* the query coordinates of a '-' record shrink along the view */
#pragma unroll 1
                for (int64_t B = 10; B <= c_hi; B *= 10) {
                    if (B <= c_lo) continue;
                    if (n_cross == FLAT_MAX_CROSS) return flat_leave(F, rec, FLAT_WHY_DIGITS);
                    uint32_t re, ml;
                    flat_find(R, v, f == 1, (uint32_t)(falling ? c_hi - B + 1 : B - c_lo), re, ml);
                    if (lane == 0) {
                        cross[n_cross][0] = re;
                        cross[n_cross][1] = ml;
                        cross[n_cross][2] = falling ? 1u : 0u;
                    }
                    n_cross++;
                }
            }
            __builtin_amdgcn_wave_barrier();
            bytes += (int64_t)flat_cross_digits(cross, n_cross, win.rows);
        }
        rows_kernel = v.n <= PAFFY_ROWS_MAX_OPS;
    } else {
        if (v.n == 0) return flat_leave(F, rec, FLAT_WHY_EMPTY);
        const uint32_t lenH = header_len(rs, false);
        if (lenH > 3 * PAFFY_TMPL_MAX) return flat_leave(F, rec, FLAT_WHY_HEADER_LEN);
        /* a window that is not reversed, with whole end ops, is a stretch of the record's own cigar text: the copy writer (see k_flat_lane) */
        copy = lenH + 8 <= PAFFY_TMPL_MAX && !v.rev && win.text <= FLAT_COPY_MAX; /* (a fixed trim's shortened end ops are written anew, the stretch between them copied) */
        line_kernel = !copy && lenH + 8 <= PAFFY_TMPL_MAX && v.n <= PAFFY_ROWS_MAX_OPS;
        bytes = (int64_t)lenH + (int64_t)win.text + 1;
        rows = 1;
        if (FIXED) bytes -= (int64_t)flat_cut_digits(fx_len, fx_amt, v.n == 1u, false); /* the shortened end ops may have lost digits */
    }
    /* a shatter record too long for one wave of the row writer: segments of FLAT_SEG_OPS ops, one workgroup of k_emit_rows each (EmitItem) */
    const bool itemised = shatter && !rows_kernel;
    const uint32_t n_seg = itemised ? (v.n + FLAT_SEG_OPS - 1u) / FLAT_SEG_OPS : 0u;
    uint32_t item0 = 0;
    if (itemised) {
        if (lane == 0) item0 = atomicAdd(&P.info->n_items, n_seg);
        item0 = uni(item0);
        if (item0 + n_seg > P.items_cap) return flat_leave(F, rec, FLAT_WHY_ROW_SHAPE); /* cannot happen: the host sizes the list for every op of the text */
    }
    const bool four_waves = !shatter && !line_kernel && !copy;
    if (lane == 0) {
        if (four_waves) atomicAdd(&P.info->g_count, 1u);
        F.flat_done[rec] = 1;
        P.status[rec] = (uint32_t)KLASS_LDS << 16;
        P.err_aux[rec] = 0;
        P.n_ops[rec] = 0;
        P.out_len[rec] = bytes;
        P.out_rows[rec] = rows;
        plan->qs = s.qs; plan->qe = s.qe; plan->ts = s.ts; plan->te = s.te; plan->sub_lo = (int64_t)sub_lo; plan->sub_hi = (int64_t)sub_hi;
        plan->lo = v.lo; plan->n = v.n;
        plan->flags = (v.rev ? 1u : 0u) | (v.swp ? 2u : 0u) | (swapped ? 4u : 0u) | 8u | ((uint32_t)rs.type << 8) | (shatter ? 16u : 0u) |
                      (rows_kernel ? 64u : 0u) | (line_kernel ? 0x10000u : 0u) | 0x40000u | (itemised ? 0x80000u : 0u) | (copy ? FLAT_F_COPY : 0u);
        plan->chunk = four_waves ? (((v.n + 255u) / 256u) | 1u) : (((v.n + 63u) / 64u) | 1u);
        for (int w = 0; w < 4; w++) plan->wq[w] = plan->wt[w] = plan->wo[w] = 0;
        if (copy) {
            plan->wq[0] = (int64_t)v.wlo.text;
            plan->wt[0] = (int64_t)win.text;
            if (FIXED && (sub_lo | sub_hi)) {
                /* the window's first / last op as the cuts left them: new length (0: untouched) and the bytes of its text in the input */
                auto dl = [](uint32_t x) { return 1u + (x >= 10u) + (x >= 100u) + (x >= 1000u); };
                const bool one = v.n == 1u;
                if (fx_amt[0] || (one && fx_amt[1])) {
                    plan->wq[1] = (int64_t)(fx_len[0] - fx_amt[0] - (one ? fx_amt[1] : 0u));
                    plan->wq[2] = (int64_t)(dl(fx_len[0]) + 1u);
                }
                if (!one && fx_amt[1]) {
                    plan->wt[1] = (int64_t)(fx_len[1] - fx_amt[1]);
                    plan->wt[2] = (int64_t)(dl(fx_len[1]) + 1u);
                }
            }
        }
    }
    if (itemised) {
#pragma unroll 1
        for (uint32_t g = 0; g < n_seg; g++) {
            const uint32_t at = g * FLAT_SEG_OPS; /* < v.n */
            FlatPre e = flat_sub(v.whi, v.whi); /* zeros */
            if (at) {
                const FlatPre cut = R.raw_prefix(v.rev ? v.lo + v.n - at : v.lo + at);
                e = v.rev ? flat_sub(v.whi, cut) : flat_sub(cut, v.wlo);
            }
            if (lane == 0) {
                EmitItem it;
                it.rec = rec;
                it.wb = at;
                it.we = at + FLAT_SEG_OPS < v.n ? at + FLAT_SEG_OPS : v.n;
                it.pad = 0;
                it.cq0 = (int64_t)e.m + e.x - (v.swp ? e.ins : e.del);
                it.ct0 = (int64_t)e.m + e.x - (v.swp ? e.del : e.ins);
                it.wo = (int64_t)e.rows * (int64_t)row_bytes1 + 3ll * (int64_t)e.extra + (int64_t)flat_cross_digits(cross, n_cross, e.rows);
                P.items[item0 + g] = it;
            }
        }
    }
    if (four_waves) {
        /* the four-wave line writer: text bytes in front of each wave's share of the view (wave w owns the view's ops [64 w chunk,
           64 (w + 1) chunk), as sweep_bounds() of record_kernel.h cuts them) */
        const uint32_t chunk = ((v.n + 255u) / 256u) | 1u;
#pragma unroll 1
        for (uint32_t w = 1; w < 4; w++) {
            const uint64_t at64 = 64ull * w * chunk;
            const uint32_t at = at64 < v.n ? (uint32_t)at64 : v.n;
            const FlatPre cut = R.raw_prefix(v.rev ? v.lo + v.n - at : v.lo + at);
            const FlatPre e = v.rev ? flat_sub(v.whi, cut) : flat_sub(cut, v.wlo);
            /* (a fixed trim's shortened first op: the digits it lost are missing in front of every later wave's share) */
            if (lane == 0) plan->wo[w] = (int64_t)e.text - (FIXED && at ? (int64_t)flat_cut_digits(fx_len, fx_amt, v.n == 1u, true) : 0ll);
        }
    }
}

template <bool FIXED>
__global__ __launch_bounds__(64 * FLAT_SIZE_WAVES) void k_flat_size(FlatSizeParams F) {
    __shared__ uint32_t s_cross[FLAT_SIZE_WAVES][FLAT_MAX_CROSS][3]; /* per wave: the powers of ten inside a record's coordinate ranges (flat_find) */
    /* the records k_flat_lane handed on, one wave each (what derives from the wave's number is wave-uniform: told to the compiler, it
       lives in scalar registers) */
    const uint32_t wave_in_group = uni(threadIdx.x >> 6);
    /* FIXED: every record (the lane kernel does not know the fixed trim and is not launched) */
    const uint32_t n_defer = FIXED ? F.P.n_rec : F.P.info->flat_defer;
    for (uint32_t li = uni(blockIdx.x * FLAT_SIZE_WAVES + (threadIdx.x >> 6)); li < n_defer; li += gridDim.x * FLAT_SIZE_WAVES) {
        flat_size_one<FIXED>(F, FIXED ? li : uni(F.defer[li]), s_cross[wave_in_group]);
        __builtin_amdgcn_wave_barrier();
    }
}

#endif
