/*
 * tile_kernel.h -- `paffy tile` (impl/paf_tile.c:156-178) on gfx950.
 *
 * Semantics restated: records are visited in (chain_score desc, score
 * desc, input order) order; state is one uint16 coverage counter per base of each QUERY sequence
 * (impl/paf.c:675-688); visiting a record walks its cigar upward from query_start whatever the
 * strand, bumps the counter of every aligned base (saturating at 32766, impl/paf.c:700) and sets
 * tile_level = the smallest level L with #(aligned bases whose new count <= L) >= aligned / 2.0
 * (impl/paf_tile.c:36-93); no aligned base -> 32767. Output = every record in visiting order,
 * cigar text verbatim (impl/paf.c:381-385), tl and a synthesised tp written (impl/paf.c:343-356).
 *
 * Mapping: records of different query sequences are independent, records of one sequence are
 * strictly ordered. One persistent workgroup per query sequence walks that sequence's records in
 * order. Per record the cigar text is streamed in 2 KiB tiles: every lane parses the ops that end
 * in its 8 bytes, a workgroup scan places them on the query, the aligned ops of the tile go to an
 * LDS list, and the waves split the tile's aligned bases evenly: 64 consecutive bases per step,
 * coalesced 2-byte read-modify-writes of the counters in HBM, level histogram in LDS.
 */
#ifndef PAFFY_TILE_KERNEL_H_
#define PAFFY_TILE_KERNEL_H_

#include "device_util.h"
#include "record_types.h"

#define TILE_TEXT (PAFFY_NT * 8u) /* cigar bytes per round */
#define TILE_HIST 4096u           /* histogram window (levels) held in LDS */

struct TileParams {
    const uint8_t *in;
    const RecMeta *meta;
    const uint32_t *order;       /* records in visiting order, grouped by query sequence */
    const uint32_t *contig_off;  /* [n_contigs + 1] ranges of `order` */
    const uint64_t *contig_base; /* [n_contigs] first counter of the sequence in `counts` */
    const uint32_t *rank_of;     /* visiting rank of every record (for the first-error rule) */
    uint16_t *counts;
    uint32_t n_contigs;
    int64_t *tile_level; /* per record */
    DevInfo *info;
    int32_t *err_aux;
};

struct TileOp { /* one aligned op of the current text tile */
    uint64_t qpos;  /* first query position (absolute) */
    uint32_t len;   /* clipped to 2^32-1: longer ops cannot pass the position asserts */
    uint32_t apos;  /* aligned bases of the tile before this op */
};

__device__ __forceinline__ void tile_fail(const TileParams &P, uint32_t rec, int code, int aux) {
    if (threadIdx.x == 0) {
        P.err_aux[rec] = aux;
        atomicMin(&P.info->first_err_key, ((unsigned long long)P.rank_of[rec] << 16) | (1ull << 8) | (unsigned long long)code);
    }
}

/*
 * One pass over the cigar of `rec`. bump = true: increment the counters (first pass); otherwise
 * only read them. Levels in [win, win + TILE_HIST) are histogrammed into hist[]; levels below win
 * are counted in *below. Returns 0, or an error code (bad cigar character / position assert).
 */
__device__ int tile_walk(const TileParams &P, uint32_t rec, const RecMeta &m, uint16_t *counts, bool bump, uint32_t win, uint32_t *hist,
                         TileOp *list, uint8_t *txt, BlockComm &bc, Shared *sh, int64_t *aligned_out, int64_t *below_out) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t cg_off = m.cg_off, end = m.cg_off + m.cg_len;
    const uint32_t a0 = cg_off & ~7u;
    for (uint32_t i = tid; i < TILE_HIST; i += PAFFY_NT) hist[i] = 0;
    if (tid == 0) {
        sh->err_pos = 0xffffffffu;
        sh->flags = 0;
    }
    __syncthreads();
    int64_t qcur = m.qs;      /* query position of the next op (wave-uniform carry) */
    int64_t aligned = 0, below = 0;
    for (uint32_t tb = a0; tb < end; tb += TILE_TEXT) {
        /* stage the tile; 32 bytes of halo carry digit runs across tiles */
        uint4 h = make_uint4(0, 0, 0, 0);
        if (tb != a0 && tid < 2) h = reinterpret_cast<uint4 *>(txt + PAFFY_HALO + TILE_TEXT - 32)[tid];
        __syncthreads();
        if (tb != a0 && tid < 2) reinterpret_cast<uint4 *>(txt)[tid] = h;
        const uint32_t g = tb + tid * 8;
        uint2 v = make_uint2(0, 0);
        if (g < end) v = *reinterpret_cast<const uint2 *>(P.in + g);
        reinterpret_cast<uint2 *>(txt + PAFFY_HALO)[tid] = v;
        __syncthreads();
        /* ops ending in my 8 bytes: lengths, and my totals of query-consuming / aligned bases */
        const uint32_t w[2] = {v.x, v.y};
        uint32_t opmask = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t c = (w[j >> 2] >> ((j & 3) * 8)) & 0xffu;
            uint32_t pos = g + j;
            bool inr = pos >= cg_off && pos < end;
            bool dig = (c - '0') < 10u;
            if (inr && !dig) opmask |= 1u << j;
            if (inr && dig && pos == end - 1) atomicMin(&sh->err_pos, end);
        }
        int64_t lens[8];
        int codes[8];
        int64_t sums[3] = {0, 0, 0}, tots[3]; /* query-consuming bases, aligned bases, aligned ops */
#pragma unroll
        for (int j = 0; j < 8; j++) {
            lens[j] = 0;
            codes[j] = -1;
            if (opmask & (1u << j)) {
                uint32_t c = (w[j >> 2] >> ((j & 3) * 8)) & 0xffu;
                int code = c == 'M' ? OP_M : c == 'I' ? OP_I : c == 'D' ? OP_D : c == '=' ? OP_EQ : c == 'X' ? OP_X : -1;
                if (code < 0) {
                    atomicMin(&sh->err_pos, g + j);
                    code = OP_D;
                }
                int p = (int)(PAFFY_HALO + tid * 8 + j) - 1;
                uint32_t avail = g + j - cg_off, reach = (uint32_t)(p + 1);
                if (tb == a0 && reach > tid * 8 + j) reach = tid * 8 + j;
                uint32_t lim = avail < reach ? avail : reach;
                uint64_t len = 0, pw = 1;
                uint32_t k = 0;
                for (; k < lim; k++) {
                    uint32_t d = (uint32_t)txt[p - (int)k] - '0';
                    if (d > 9u) break;
                    len += d * pw;
                    pw *= 10;
                }
                if (k == lim && lim < avail) atomicOr(&sh->flags, 1u); /* digit run longer than the halo */
                int64_t l56 = (int64_t)(len << 8) >> 8;
                lens[j] = l56;
                codes[j] = code;
                if (code != OP_D) sums[0] += l56;
                if (code == OP_M || code == OP_EQ || code == OP_X) {
                    sums[1] += l56;
                    sums[2] += 1;
                }
            }
        }
        block_excl_scan<3>(sums, tots, bc);
        /* my aligned ops go to the LDS list */
        {
            int64_t q = qcur + sums[0], a = sums[1];
            uint32_t li = (uint32_t)sums[2];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (codes[j] < 0) continue;
                if (codes[j] == OP_M || codes[j] == OP_EQ || codes[j] == OP_X) {
                    /* assert(i + j < query_end && i + j >= 0 && i + j < query_length), impl/paf.c:698 */
                    if (lens[j] > 0 && (q < 0 || q + lens[j] > m.qe || q + lens[j] > m.qlen)) atomicOr(&sh->flags, 2u);
                    TileOp o;
                    o.qpos = (uint64_t)q;
                    o.len = lens[j] < 0 ? 0u : (lens[j] > 0xffffffffll ? 0xffffffffu : (uint32_t)lens[j]);
                    o.apos = (uint32_t)a;
                    list[li++] = o;
                    a += lens[j];
                }
                if (codes[j] != OP_D) q += lens[j];
            }
        }
        __syncthreads();
        const uint32_t fl = sh->flags;
        const uint32_t n_al = (uint32_t)tots[2];
        const int64_t a_tile = tots[1];
        if (fl == 0 && sh->err_pos == 0xffffffffu && n_al > 0 && a_tile > 0) {
            /* waves split the tile's aligned bases; 64 consecutive bases per step */
            const uint64_t per = ((uint64_t)a_tile + PAFFY_NWAVE - 1) / PAFFY_NWAVE;
            uint64_t s0 = per * wave, s1 = s0 + per;
            if (s0 > (uint64_t)a_tile) s0 = (uint64_t)a_tile;
            if (s1 > (uint64_t)a_tile) s1 = (uint64_t)a_tile;
            if (s0 < s1) {
                uint32_t lo = 0, hi = n_al - 1; /* last op with apos <= s0 */
                while (lo < hi) {
                    uint32_t mid = (lo + hi + 1) >> 1;
                    if ((uint64_t)list[mid].apos <= s0) lo = mid;
                    else hi = mid - 1;
                }
                for (uint32_t oi = lo; oi < n_al; oi++) {
                    const TileOp o = list[oi];
                    if ((uint64_t)o.apos >= s1) break;
                    uint64_t b0 = s0 > o.apos ? s0 - o.apos : 0, b1 = s1 - o.apos;
                    if (b1 > o.len) b1 = o.len;
                    for (uint64_t b = b0 + lane; b < b1; b += 64) {
                        uint16_t *cp = counts + o.qpos + b;
                        uint32_t cnt = *cp;
                        if (bump && cnt < 32766u) { /* INT16_MAX - 1, impl/paf.c:700 */
                            cnt++;
                            *cp = (uint16_t)cnt;
                        }
                        if (cnt >= win && cnt < win + TILE_HIST) atomicAdd(&hist[cnt - win], 1u);
                        else if (cnt < win) below++;
                    }
                }
            }
        }
        aligned += a_tile;
        qcur += tots[0];
        __syncthreads();
    }
    int64_t bs[1] = {below};
    block_sum<1>(bs, bc);
    *below_out = bs[0];
    *aligned_out = aligned;
    const uint32_t ep = sh->err_pos, fl = sh->flags;
    __syncthreads();
    if (ep != 0xffffffffu) {
        if (bump) tile_fail(P, rec, PAFFY_ERR_CIGAR_CHAR, ep < end ? P.in[ep] : 0);
        return PAFFY_ERR_CIGAR_CHAR;
    }
    if (fl || qcur != m.qe) { /* position asserts / assert(i == query_end), impl/paf.c:708 */
        if (bump) tile_fail(P, rec, PAFFY_ERR_TILE_ASSERT, (int)fl);
        return PAFFY_ERR_TILE_ASSERT;
    }
    return 0;
}

#define TILE_LDS_BYTES (TILE_HIST * 4 + (PAFFY_NT * 8) * 16 + (PAFFY_HALO + TILE_TEXT) + 64 * 8 + 64)

__global__ __launch_bounds__(PAFFY_NT) void k_tile(TileParams P) {
    extern __shared__ uint4 smem4[];
    uint8_t *smem = reinterpret_cast<uint8_t *>(smem4);
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem);
    TileOp *list = reinterpret_cast<TileOp *>(smem + TILE_HIST * 4);
    uint8_t *txt = smem + TILE_HIST * 4 + (PAFFY_NT * 8) * 16;
    BlockComm bc;
    bc.scratch = reinterpret_cast<int64_t *>(txt + PAFFY_HALO + TILE_TEXT);
    bc.flip = 0;
    Shared *sh = reinterpret_cast<Shared *>(reinterpret_cast<uint8_t *>(bc.scratch) + 64 * 8);
    for (uint32_t c = blockIdx.x; c < P.n_contigs; c += gridDim.x) {
        uint16_t *counts = P.counts + P.contig_base[c];
        for (uint32_t k = P.contig_off[c]; k < P.contig_off[c + 1]; k++) {
            const uint32_t rec = P.order[k];
            if (threadIdx.x == 0) sh->bcast[0] = (int64_t)(P.info->first_err_key >> 16);
            __syncthreads();
            const bool stop = (uint64_t)sh->bcast[0] < P.rank_of[rec]; /* an earlier record already failed: nothing is written */
            __syncthreads();
            if (stop) break;
            const RecMeta m = P.meta[rec];
            if (!m.has_cg) { /* cigar_parse(NULL): the reference dereferences NULL, impl/paf_tile.c:166 */
                tile_fail(P, rec, PAFFY_ERR_NULL_CIGAR, 0);
                break;
            }
            int64_t aligned = 0, below = 0;
            int rc = tile_walk(P, rec, m, counts, true, 0, hist, list, txt, bc, sh, &aligned, &below);
            if (rc) break;
            int64_t level = 32767; /* no aligned base: INT16_MAX, impl/paf_tile.c:62-65 */
            if (aligned > 0) {
                uint32_t win = 0;
                int64_t acc = 0;
                bool found = false;
                for (;;) {
                    /* prefix over the window, one lane per 16 levels, then the crossing level */
                    uint32_t local = 0;
                    const uint32_t base = threadIdx.x * (TILE_HIST / PAFFY_NT);
                    for (uint32_t i = 0; i < TILE_HIST / PAFFY_NT; i++) local += hist[base + i];
                    int64_t pre[1] = {local}, tot[1];
                    block_excl_scan<1>(pre, tot, bc);
                    int64_t run = acc + pre[0];
                    int64_t mine = INT64_MAX;
                    for (uint32_t i = 0; i < TILE_HIST / PAFFY_NT; i++) {
                        run += hist[base + i];
                        if (mine == INT64_MAX && 2 * run >= aligned) mine = win + base + i; /* j >= matches / 2.0 */
                    }
                    int64_t lv = block_min_i64(mine, bc);
                    if (lv != INT64_MAX) {
                        level = lv;
                        found = true;
                        break;
                    }
                    acc += tot[0];
                    win += TILE_HIST;
                    if (win >= 32768u) break;
                    rc = tile_walk(P, rec, m, counts, false, win, hist, list, txt, bc, sh, &aligned, &below);
                    if (rc) break;
                }
                if (!found || level <= 0) { /* assert(i > 0) / assert(0), impl/paf_tile.c:86-90 */
                    tile_fail(P, rec, PAFFY_ERR_TILE_ASSERT, 3);
                    break;
                }
            }
            if (threadIdx.x == 0) P.tile_level[rec] = level;
            __syncthreads();
        }
        __syncthreads();
    }
}

#endif
