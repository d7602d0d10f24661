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
#define TILE_UNROLL 8             /* counter loads in flight per lane */

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
    /* sliced mode: work item i = one TILE_SLICE-base range of one query sequence with >= 1 record */
    const uint32_t *item_off;    /* [n_items + 1] ranges of `order` (records overlapping the slice, visiting order) */
    const uint32_t *item_contig; /* [n_items] */
    const uint32_t *item_slice;  /* [n_items] slice index inside the sequence */
    const uint32_t *slot_base;   /* per record: first partial-histogram slot */
    struct TilePartial *partials;
    uint32_t n_items;
};

#define TILE_SLICE_SHIFT 20 /* 1 Mi bases per slice */
#define TILE_PAIRS 128      /* distinct levels a (record, slice) partial histogram can hold: at coverage c the new counts of one
                               record spread over about 8 sqrt(c) levels, so 128 serves coverage in the hundreds */

struct TilePartial { /* pairs in increasing level order */
    uint32_t n, overflow;
    uint16_t level[TILE_PAIRS]; /* < TILE_HIST in sliced mode */
    uint32_t count[TILE_PAIRS];
};

/* first / last slice a record's query range touches (host and device use the same arithmetic) */
__host__ __device__ inline uint32_t tile_first_slice(int64_t qs, int64_t qe, int64_t qlen) {
    int64_t a = qs < 0 ? 0 : qs;
    if (qlen > 0 && a >= qlen) a = qlen - 1;
    (void)qe;
    return (uint32_t)(a >> TILE_SLICE_SHIFT);
}
__host__ __device__ inline uint32_t tile_last_slice(int64_t qs, int64_t qe, int64_t qlen) {
    int64_t b = qe > qlen ? qlen : qe;
    b -= 1;
    int64_t a = qs < 0 ? 0 : qs;
    if (qlen > 0 && a >= qlen) a = qlen - 1;
    if (b < a) b = a;
    return (uint32_t)(b >> TILE_SLICE_SHIFT);
}

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
/*
 * SIDE 0: the query sequence, as `paffy tile` and `paffy to_bed` walk it. SIDE 1 / 2: the TARGET sequence of the record, what
 * `paffy to_bed -n` walks after paf_invert (impl/paf_to_bed.c:176-180): the inverted record's query is the target, its I ops are
 * the D ops, and on the - strand its cigar is reversed (impl/paf.c:463-490) -- walking the reversed ops upward from target_start
 * covers the mirror image of what the forward ops cover, so SIDE 2 walks forward and mirrors every position in [ts, te).
 */
template <int SIDE = 0>
__device__ int tile_walk(const TileParams &P, uint32_t rec, const RecMeta &m, uint16_t *counts, bool bump, uint32_t win, uint32_t *hist,
                         TileOp *list, uint8_t *txt, BlockComm &bc, Shared *sh, int64_t *aligned_out, int64_t *below_out,
                         uint64_t clip_lo = 0, uint64_t clip_hi = ~0ull) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t cg_off = m.cg_off, end = m.cg_off + m.cg_len;
    const uint32_t a0 = cg_off & ~7u;
    for (uint32_t i = tid; i < TILE_HIST; i += PAFFY_NT) hist[i] = 0;
    if (tid == 0) {
        sh->err_pos = 0xffffffffu;
        sh->flags = 0;
    }
    __syncthreads();
    const int64_t w_start = SIDE ? m.ts : m.qs, w_end = SIDE ? m.te : m.qe, w_len = SIDE ? m.tlen : m.qlen;
    const int skip_op = SIDE ? OP_I : OP_D; /* the op that does not advance along the walked sequence */
    int64_t qcur = w_start;   /* position of the next op on the walked sequence (wave-uniform carry) */
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
                if (code != skip_op) sums[0] += l56;
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
                    if (lens[j] > 0 && (q < 0 || q + lens[j] > w_end || q + lens[j] > w_len)) atomicOr(&sh->flags, 2u);
                    TileOp o;
                    o.qpos = (uint64_t)q;
                    o.len = lens[j] < 0 ? 0u : (lens[j] > 0xffffffffll ? 0xffffffffu : (uint32_t)lens[j]);
                    o.apos = (uint32_t)a;
                    list[li++] = o;
                    a += lens[j];
                }
                if (codes[j] != skip_op) q += lens[j];
            }
        }
        __syncthreads();
        const uint32_t fl = sh->flags & 0xffffu;
        const uint32_t n_al = (uint32_t)tots[2];
        const int64_t a_tile = tots[1];
        if (fl == 0 && sh->err_pos == 0xffffffffu && n_al > 0 && a_tile > 0) {
            /* waves split the tile's aligned bases; 64 consecutive bases per step */
            const uint64_t per = ((uint64_t)a_tile + PAFFY_NWAVE - 1) / PAFFY_NWAVE;
            uint64_t s0 = per * wave, s1 = s0 + per;
            if (s0 > (uint64_t)a_tile) s0 = (uint64_t)a_tile;
            if (s1 > (uint64_t)a_tile) s1 = (uint64_t)a_tile;
            if (s0 < s1) {
                /*
                 * Lane l handles aligned bases s0 + l, s0 + l + 64, ...: 64 consecutive bases of an op are 64
                 * consecutive counters (coalesced). A per-lane cursor follows the op list; TILE_UNROLL loads are
                 * issued before the first increment so that the HBM latency is paid once per 8 steps, not per op.
                 */
                uint32_t lo = 0, hi = n_al - 1; /* last op with apos <= first base of this lane */
                const uint64_t first = s0 + lane;
                while (lo < hi) {
                    uint32_t mid = (lo + hi + 1) >> 1;
                    if ((uint64_t)list[mid].apos <= first) lo = mid;
                    else hi = mid - 1;
                }
                uint32_t oi = lo;
                for (uint64_t a = first; a < s1; a += 64ull * TILE_UNROLL) {
                    uint16_t *cp[TILE_UNROLL];
                    uint32_t cv[TILE_UNROLL];
#pragma unroll
                    for (int u = 0; u < TILE_UNROLL; u++) {
                        const uint64_t au = a + 64ull * u;
                        cp[u] = nullptr;
                        cv[u] = 0;
                        if (au < s1) {
                            while (oi + 1 < n_al && (uint64_t)list[oi + 1].apos <= au) oi++;
                            const TileOp o = list[oi];
                            const uint64_t rel = au - o.apos;
                            const uint64_t at = SIDE == 2 ? (uint64_t)(w_start + w_end - 1) - (o.qpos + rel) : o.qpos + rel;
                            if (rel < o.len && at >= clip_lo && at < clip_hi) { /* inside the op and inside this slice */
                                cp[u] = counts + at;
                                cv[u] = *cp[u];
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < TILE_UNROLL; u++) {
                        uint32_t cnt = cv[u];
                        const bool valid = cp[u] != nullptr;
                        if (valid && bump && cnt < 32766u) { /* INT16_MAX - 1, impl/paf.c:700 */
                            cnt++;
                            *cp[u] = (uint16_t)cnt;
                        }
                        /* level histogram: most bases share a level, so count equal levels with ballots and
                           issue one LDS atomic per distinct level of the wave instead of 64 colliding ones */
                        unsigned long long todo = __ballot(valid);
                        while (todo) {
                            const int leader = __ffsll((long long)todo) - 1;
                            const uint32_t lv = (uint32_t)__builtin_amdgcn_readlane((int)cnt, leader);
                            const unsigned long long same = __ballot(valid && cnt == lv);
                            if ((int)lane == leader) {
                                const uint32_t k = (uint32_t)__popcll(same);
                                if (lv >= win && lv < win + TILE_HIST) atomicAdd(&hist[lv - win], k);
                                else if (lv < win) below += k;
                                else atomicOr(&sh->flags, 0x10000u); /* level beyond the window (sliced mode: fall back) */
                            }
                            todo &= ~same;
                        }
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
    if ((fl & 0xffffu) || qcur != w_end) { /* position asserts / assert(i == query_end), impl/paf.c:708 */
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
            const bool stop = (uint64_t)sh->bcast[0] <= P.rank_of[rec]; /* this or an earlier record already failed: nothing is written */
            __syncthreads();
            if (stop) break;
            const RecMeta m = P.meta[rec];
            if (!m.has_cg) { /* cigar_parse(NULL): the reference dereferences NULL, impl/paf_tile.c:166 */
                tile_fail(P, rec, PAFFY_ERR_NULL_CIGAR, 0);
                break;
            }
            /* pull the counter lines of the following records' query ranges towards L2 while this one is walked:
               a single workgroup keeps too few bytes in flight to hide a cold HBM miss per step */
            for (uint32_t ahead = 1; ahead <= 2 && k + ahead < P.contig_off[c + 1]; ahead++) {
                const RecMeta &nx = P.meta[P.order[k + ahead]];
                const int64_t q0 = nx.qs < 0 ? 0 : nx.qs, q1 = nx.qe > nx.qlen ? nx.qlen : nx.qe;
                if ((ahead == 2 || k == P.contig_off[c]) && nx.qlen == m.qlen)
                    for (int64_t pos = q0 + (int64_t)threadIdx.x * 64; pos < q1; pos += (int64_t)PAFFY_NT * 64) {
                        uint32_t touch = counts[pos];
                        asm volatile("" ::"v"(touch));
                    }
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


/*
 * Sliced tiling: one workgroup per (query sequence, 1 Mi-base slice) that has records. It walks, in
 * visiting order, every record overlapping the slice, touches only the counters inside the slice, and
 * leaves for each (record, slice) the histogram of the new counts as at most TILE_PAIRS (level, count)
 * pairs. Records of one sequence still meet every base in visiting order, so the counters evolve
 * exactly as in the sequential reference. k_tile_merge then sums a record's partial histograms and
 * takes the median level. Anything that does not fit (a level >= TILE_HIST, too many distinct levels)
 * raises DevInfo.internal and the host repeats the batch with k_tile.
 */
__global__ __launch_bounds__(PAFFY_NT) void k_tile_slices(TileParams P) {
    extern __shared__ uint4 smem4[];
    uint8_t *smem = reinterpret_cast<uint8_t *>(smem4);
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem);
    TileOp *list = reinterpret_cast<TileOp *>(smem + TILE_HIST * 4);
    uint8_t *txt = smem + TILE_HIST * 4 + (PAFFY_NT * 8) * 16;
    BlockComm bc;
    bc.scratch = reinterpret_cast<int64_t *>(txt + PAFFY_HALO + TILE_TEXT);
    bc.flip = 0;
    Shared *sh = reinterpret_cast<Shared *>(reinterpret_cast<uint8_t *>(bc.scratch) + 64 * 8);
    const uint32_t item = blockIdx.x;
    const uint32_t c = P.item_contig[item], slice = P.item_slice[item];
    uint16_t *counts = P.counts + P.contig_base[c];
    const uint64_t lo = (uint64_t)slice << TILE_SLICE_SHIFT, hi = lo + (1ull << TILE_SLICE_SHIFT);
    for (uint32_t k = P.item_off[item]; k < P.item_off[item + 1]; k++) {
        const uint32_t rec = P.order[k];
        if (threadIdx.x == 0) sh->bcast[0] = (int64_t)(P.info->first_err_key >> 16);
        __syncthreads();
        const bool stop = (uint64_t)sh->bcast[0] <= P.rank_of[rec];
        __syncthreads();
        if (stop) break;
        const RecMeta m = P.meta[rec];
        if (!m.has_cg) {
            tile_fail(P, rec, PAFFY_ERR_NULL_CIGAR, 0);
            break;
        }
        int64_t aligned = 0, below = 0;
        int rc = tile_walk(P, rec, m, counts, true, 0, hist, list, txt, bc, sh, &aligned, &below, lo, hi);
        if (rc) break;
        if (sh->flags & 0x10000u) {
            if (threadIdx.x == 0) atomicOr(&P.info->internal, 0x100u);
        }
        /* compact the window into (level, count) pairs */
        TilePartial *part = P.partials + P.slot_base[rec] + (slice - tile_first_slice(m.qs, m.qe, m.qlen));
        uint32_t mine = 0;
        const uint32_t base = threadIdx.x * (TILE_HIST / PAFFY_NT);
        for (uint32_t i = 0; i < TILE_HIST / PAFFY_NT; i++) mine += hist[base + i] != 0;
        int64_t pre[1] = {mine}, tot[1];
        block_excl_scan<1>(pre, tot, bc);
        uint32_t o = (uint32_t)pre[0];
        for (uint32_t i = 0; i < TILE_HIST / PAFFY_NT; i++) {
            const uint32_t h = hist[base + i];
            if (h) {
                if (o < TILE_PAIRS) {
                    part->level[o] = (uint16_t)(base + i);
                    part->count[o] = h;
                }
                o++;
            }
        }
        if (threadIdx.x == 0) {
            part->n = tot[0] < TILE_PAIRS ? (uint32_t)tot[0] : TILE_PAIRS;
            part->overflow = tot[0] > TILE_PAIRS;
            if (tot[0] > TILE_PAIRS) atomicOr(&P.info->internal, 0x100u);
        }
        __syncthreads();
    }
}

/* one lane per record: sum its partial histograms, median level (impl/paf_tile.c:81-88) */
__global__ __launch_bounds__(PAFFY_NT) void k_tile_merge(TileParams P, uint32_t n_rec) {
    const uint32_t rec = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (rec >= n_rec) return;
    if ((P.info->first_err_key >> 16) <= P.rank_of[rec]) return; /* nothing is written anyway */
    const RecMeta &m = P.meta[rec];
    const uint32_t s0 = tile_first_slice(m.qs, m.qe, m.qlen), s1 = tile_last_slice(m.qs, m.qe, m.qlen);
    const TilePartial *part = P.partials + P.slot_base[rec];
    uint64_t aligned = 0;
    for (uint32_t s = 0; s <= s1 - s0; s++)
        for (uint32_t i = 0; i < part[s].n; i++) aligned += part[s].count[i];
    int64_t level = 32767; /* no aligned base */
    if (aligned > 0) {
        /* the partial histograms are sorted by level: merge them, smallest level first, until half the bases are covered */
        const uint32_t ns = s1 - s0 + 1;
        uint64_t acc = 0;
        level = -1;
        if (ns <= 8) {
            uint32_t cur[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (;;) {
                uint32_t best = 0xffffffffu;
                for (uint32_t s = 0; s < ns; s++)
                    if (cur[s] < part[s].n && part[s].level[cur[s]] < best) best = part[s].level[cur[s]];
                if (best == 0xffffffffu) break;
                for (uint32_t s = 0; s < ns; s++)
                    if (cur[s] < part[s].n && part[s].level[cur[s]] == best) acc += part[s].count[cur[s]++];
                if (2 * acc >= aligned) {
                    level = best;
                    break;
                }
            }
        } else { /* a record over more than eight slices: no cursors, repeatedly take the smallest level above the last one */
            int64_t last = -1;
            for (;;) {
                uint32_t best = 0xffffffffu;
                for (uint32_t s = 0; s < ns; s++)
                    for (uint32_t i = 0; i < part[s].n; i++)
                        if ((int64_t)part[s].level[i] > last && part[s].level[i] < best) best = part[s].level[i];
                if (best == 0xffffffffu) break;
                for (uint32_t s = 0; s < ns; s++)
                    for (uint32_t i = 0; i < part[s].n; i++)
                        if (part[s].level[i] == best) acc += part[s].count[i];
                if (2 * acc >= aligned) {
                    level = best;
                    break;
                }
                last = best;
            }
        }
        if (level <= 0) { /* assert(i > 0) / assert(0) */
            P.err_aux[rec] = 3;
            atomicMin(&P.info->first_err_key, ((unsigned long long)P.rank_of[rec] << 16) | (1ull << 8) | (unsigned long long)PAFFY_ERR_TILE_ASSERT);
            return;
        }
    }
    P.tile_level[rec] = level;
}

/* ------------------------------------------------------------------------------------------------
 * `paffy to_bed` (impl/paf_to_bed.c:33-55, 166-190): the coverage counters of `paffy tile` without the levels, then every
 * sequence's counters as maximal runs "name start end value". Work items are (sequence, slice) like the sliced tile mode; an
 * entry of the item's list is a record index, with bits 30-31 = the SIDE of tile_walk (-n adds the target side of every record).
 * ---------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(PAFFY_NT) void k_bed_cover(TileParams P) {
    extern __shared__ uint4 smem4[];
    uint8_t *smem = reinterpret_cast<uint8_t *>(smem4);
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem);
    TileOp *list = reinterpret_cast<TileOp *>(smem + TILE_HIST * 4);
    uint8_t *txt = smem + TILE_HIST * 4 + (PAFFY_NT * 8) * 16;
    BlockComm bc;
    bc.scratch = reinterpret_cast<int64_t *>(txt + PAFFY_HALO + TILE_TEXT);
    bc.flip = 0;
    Shared *sh = reinterpret_cast<Shared *>(reinterpret_cast<uint8_t *>(bc.scratch) + 64 * 8);
    const uint32_t item = blockIdx.x;
    const uint32_t c = P.item_contig[item], slice = P.item_slice[item];
    uint16_t *counts = P.counts + P.contig_base[c];
    const uint64_t lo = (uint64_t)slice << TILE_SLICE_SHIFT, hi = lo + (1ull << TILE_SLICE_SHIFT);
    for (uint32_t k = P.item_off[item]; k < P.item_off[item + 1]; k++) {
        const uint32_t rec = P.order[k] & 0x3fffffffu, side = P.order[k] >> 30;
        const RecMeta m = P.meta[rec];
        int64_t aligned = 0, below = 0;
        if (!m.has_cg) continue; /* no cigar: nothing to count (cigar_count(NULL) == 0, inc/paf.h:75) -- only the end assert can fail */
        /* the level window is of no interest here: counts beyond it only raise a flag that nobody reads */
        if (side == 0) tile_walk<0>(P, rec, m, counts, true, 0, hist, list, txt, bc, sh, &aligned, &below, lo, hi);
        else if (side == 1) tile_walk<1>(P, rec, m, counts, true, 0, hist, list, txt, bc, sh, &aligned, &below, lo, hi);
        else tile_walk<2>(P, rec, m, counts, true, 0, hist, list, txt, bc, sh, &aligned, &below, lo, hi);
        __syncthreads();
    }
}

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
