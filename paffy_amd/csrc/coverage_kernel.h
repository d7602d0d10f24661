/*
 * coverage_kernel.h -- the per-base coverage counters of `paffy tile` (impl/paf_tile.c:36-93,156-178) and `paffy to_bed`
 * (impl/paf_to_bed.c:33-55,166-190) with the counters of a 32 Ki-base slice resident in LDS.
 *
 * Reference semantics restated: state is one uint16 counter per base of every sequence
 * (get_alignment_count_array, impl/paf.c:675-688); a record walks its cigar upward from query_start whatever the strand and
 * bumps the counter of every aligned base unless it has reached INT16_MAX - 1 (increase_alignment_level_counts, impl/paf.c:690-709);
 * `tile` visits the records in (chain_score desc, score desc, input order) order and gives each one the smallest level L with
 * #(aligned bases whose new count <= L) >= aligned / 2.0 (get_median_alignment_level, impl/paf_tile.c:36-93).
 *
 * MI355X mapping. A walk (an *entry*: a record and the side of it that is walked) is first turned into a bitmap, one bit per
 * base of its range, by k_cov_bitmap: the cigar text is parsed once (digits before every op letter converted eight bytes at a time)
 * and the aligned ops set their bits. The counters are then processed slice by slice: one workgroup owns a slice of 32 Ki bases,
 * keeps its 64 KB of counters in LDS, and applies, in visiting order, the bitmaps of all entries that touch the slice -- a lane
 * always owns the same 32 bases of the slice, so no synchronisation is needed between entries for the counters; adding a bitmap word
 * is four packed, saturating 16-bit adds per eight bases. For `tile` the new counts feed a level histogram in LDS (indexed by
 * level mod 512 in four copies of packed 16-bit counters; one workgroup barrier per entry, the histogram of entry k is compacted by one wave while the others walk entry
 * k + 1), stored per (entry, slice) as a dense run of counts; k_cov_merge sums an entry's runs and takes the median level.
 * Counters live in HBM between launches (2 bytes per base, as in the reference), so any number of entries can be processed in chunks.
 */
#ifndef PAFFY_COVERAGE_KERNEL_H_
#define PAFFY_COVERAGE_KERNEL_H_

#include "device_util.h"
#include "record_types.h"

#define COV_SLICE_SHIFT 15
#define COV_SLICE (1u << COV_SLICE_SHIFT) /* bases per slice: 64 KB of counters */
#define COV_WORDS (COV_SLICE / 32u)       /* bitmap words per slice */
#define COV_NT 512                        /* threads of a slice workgroup: words tid and tid + 512 of the slice are this thread's */
#define COV_NWAVE (COV_NT / 64)
#ifndef COV_HIST_W
#define COV_HIST_W 256u
#endif                   /* levels the LDS histogram tells apart (index = level mod 256) */
#define COV_BIAS 32769u                   /* LDS counters hold count + 32769: the clamped 16-bit add then stops at count 32766 (impl/paf.c:700) */

#define COV_TEXT 4096u   /* cigar bytes per round of the bitmap kernel (256 threads x 16) */
#define COV_WIN 2048u    /* bitmap words staged in LDS per round (65 536 bases); longer rounds set their bits in HBM directly */

struct CovEntry {      /* one walk, in visiting order (index = rank) */
    int64_t lo, hi;    /* range walked on the sequence, clamped to it: [lo, hi); lo == hi: nothing */
    uint64_t bm_off;   /* first word of the bitmap; word 0 holds bases [32 * (lo >> 5), +32) */
    uint32_t rec;      /* record (index into the concatenated RecMeta of all batches) */
    uint32_t side;     /* 0: query sequence. 1 / 2: target sequence of a + / - record, as the inverted record walks it (to_bed -n) */
    uint32_t contig;   /* sequence id */
    uint32_t pair_base; /* first (entry, slice) pair of this entry = its first partial-histogram slot */
    uint32_t first_slice, n_slices;
};

struct CovSlot { /* partial level histogram of one (entry, slice): counts of levels [mn, mn + n), dense, in the arena */
    uint64_t off;
    uint32_t mn, n;
};

struct CovParams {
    const uint8_t *const *batch_in; /* text of every batch */
    const RecMeta *meta;            /* all batches back to back; pad1 = batch */
    CovEntry *entries;
    uint32_t n_entries;
    uint32_t e0, e1;                /* entries of this chunk */
    uint32_t *bitmap;               /* this chunk's bitmaps; entry e starts at bm_off - bm_base */
    uint64_t bm_base;
    int64_t *aligned;               /* per entry: aligned bases (tile: `matches` of impl/paf_tile.c:45-58) */
    /* sequences */
    const int64_t *contig_len;
    const uint64_t *contig_cov;     /* first counter of the sequence in `cov` */
    const uint32_t *contig_slice0;  /* first global slice id of the sequence */
    uint16_t *cov;                  /* counters in HBM */
    /* slice work items of the chunk */
    const uint64_t *pairs;          /* sorted (global slice id << 32 | entry) */
    const uint32_t *item_start;     /* [n_items + 1] ranges of `pairs` */
    const uint32_t *item_order;     /* items by descending size, or NULL */
    uint32_t n_items;
    /* partial histograms (tile) */
    CovSlot *slots;
    uint16_t *arena;
    uint64_t arena_cap;
    unsigned long long *arena_used;
    /* results / errors */
    int64_t *level;                 /* per record */
    DevInfo *info;
    int32_t *err_aux;               /* per record */
};

/* first failing entry in visiting order wins (nothing is written then): key = rank << 16 | 1 << 8 | code, as the record kernels */
__device__ __forceinline__ void cov_fail(const CovParams &P, uint32_t e, uint32_t rec, int code, int aux) {
    P.err_aux[rec] = aux;
    atomicMin(&P.info->first_err_key, ((unsigned long long)e << 16) | (1ull << 8) | (unsigned long long)code);
}

/* ---------------------------------------------------------------------------------------------------------------------- */
/* cigar text -> ops, eight bytes of digits at a time                                                                      */
/* ---------------------------------------------------------------------------------------------------------------------- */

__device__ __forceinline__ int cov_op_code(uint32_t c) { /* impl/paf.c:96-103 */
    return c == 'M' ? OP_M : c == 'I' ? OP_I : c == 'D' ? OP_D : c == '=' ? OP_EQ : c == 'X' ? OP_X : -1;
}

/* set bits [a, b) (absolute bases, a < b, inside the entry's range) of a bitmap whose word 0 starts at base 32 * w0: LDS window */
__device__ __forceinline__ void bits_set_lds(uint32_t *win, int64_t w0, int64_t a, int64_t b) {
    const uint32_t wa = (uint32_t)((a >> 5) - w0), wb = (uint32_t)(((b - 1) >> 5) - w0);
    const uint32_t ma = 0xffffffffu << (a & 31), mb = 0xffffffffu >> (31 - ((b - 1) & 31));
    if (wa == wb) {
        atomicOr(&win[wa], ma & mb);
    } else {
        atomicOr(&win[wa], ma);
        for (uint32_t w = wa + 1; w < wb; w++) win[w] = 0xffffffffu; /* words inside one op belong to it alone */
        atomicOr(&win[wb], mb);
    }
}
__device__ __forceinline__ void bits_set_global(uint32_t *bm, int64_t w0, int64_t a, int64_t b) {
    const uint64_t wa = (uint64_t)((a >> 5) - w0), wb = (uint64_t)(((b - 1) >> 5) - w0);
    const uint32_t ma = 0xffffffffu << (a & 31), mb = 0xffffffffu >> (31 - ((b - 1) & 31));
    if (wa == wb) {
        atomicOr(&bm[wa], ma & mb);
    } else {
        atomicOr(&bm[wa], ma);
        for (uint64_t w = wa + 1; w < wb; w++) atomicOr(&bm[w], 0xffffffffu);
        atomicOr(&bm[wb], mb);
    }
}

struct CovWalkSide { /* what the walk of one entry looks at */
    int64_t start, end, len; /* query_start / query_end / query_length of the record as walked (impl/paf.c:690-709) */
    int skip_op;             /* the op that does not advance along the walked sequence */
    bool mirror;             /* - strand target side: the inverted record's cigar is reversed (impl/paf.c:463-490) */
};
__device__ __forceinline__ CovWalkSide cov_side(const RecMeta &m, uint32_t side) {
    CovWalkSide s;
    s.start = side ? m.ts : m.qs;
    s.end = side ? m.te : m.qe;
    s.len = side ? m.tlen : m.qlen;
    s.skip_op = side ? OP_I : OP_D;
    s.mirror = side == 2;
    return s;
}

/*
 * One workgroup (256 threads) per entry: cigar -> bitmap of the aligned bases of the walked range, the aligned-base count, and
 * the record's first failing check in the reference's order: cigar_parse (impl/paf.c:102), then the position asserts and the end
 * assert of increase_alignment_level_counts (impl/paf.c:698,708).
 */
/* The kernel in two workgroup shapes, like the sizing pass of the record kernels (round 3, last session): one wave per entry for the short
   cigars -- no barrier costs anything, twenty entries in flight per CU -- and four waves for the rest; an entry belongs to the shape
   whose byte range [lo_bytes, hi_bytes] holds its cigar (entries without a cigar and lines that did not parse: the four-wave shape). */
struct CovGroup64 {
    static constexpr uint32_t NT = 64, TEXT = 64 * 16, WIN = 512; /* 16 384 bases of bitmap staged per round of at most 512 ops */
    typedef g64::BlockComm Comm;
    typedef g64::Shared Sh;
    template <int K>
    static __device__ __forceinline__ void scan_u32(uint32_t (&v)[K], uint32_t (&t)[K], Comm &bc) { g64::block_excl_scan_u32<K>(v, t, bc); }
    template <int K>
    static __device__ __forceinline__ void scan_i64(int64_t (&v)[K], int64_t (&t)[K], Comm &bc) { g64::block_excl_scan<K>(v, t, bc); }
};
struct CovGroup256 {
    static constexpr uint32_t NT = 256, TEXT = COV_TEXT, WIN = COV_WIN;
    typedef g256::BlockComm Comm;
    typedef g256::Shared Sh;
    template <int K>
    static __device__ __forceinline__ void scan_u32(uint32_t (&v)[K], uint32_t (&t)[K], Comm &bc) { g256::block_excl_scan_u32<K>(v, t, bc); }
    template <int K>
    static __device__ __forceinline__ void scan_i64(int64_t (&v)[K], int64_t (&t)[K], Comm &bc) { g256::block_excl_scan<K>(v, t, bc); }
};
#define COV_BM_LDS_BYTES_OF(G) (PAFFY_HALO + G::TEXT + G::TEXT / 2 * 8 + G::WIN * 4 + 64 * 8 + 64)
template <class G>
__global__ __launch_bounds__(G::NT) void k_cov_bitmap(CovParams P, int null_cigar_is_error, uint32_t lo_bytes, uint32_t hi_bytes) {
    extern __shared__ uint4 smem4[];
    uint8_t *smem = reinterpret_cast<uint8_t *>(smem4);
    uint8_t *txt = smem;                                                        /* halo + tile */
    uint64_t *ops = reinterpret_cast<uint64_t *>(smem + PAFFY_HALO + G::TEXT); /* len << 8 | op, at most TEXT / 2 per round */
    uint32_t *win = reinterpret_cast<uint32_t *>(smem + PAFFY_HALO + G::TEXT + G::TEXT / 2 * 8);
    typename G::Comm bc;
    bc.scratch = reinterpret_cast<int64_t *>(smem + PAFFY_HALO + G::TEXT + G::TEXT / 2 * 8 + G::WIN * 4);
    bc.flip = 0;
    typename G::Sh *sh = reinterpret_cast<typename G::Sh *>(reinterpret_cast<uint8_t *>(bc.scratch) + 64 * 8);
    const uint32_t tid = threadIdx.x;
    const uint32_t e = P.e0 + blockIdx.x;
    const CovEntry E = P.entries[e];
    const RecMeta &m = P.meta[E.rec];
    {
        const uint32_t route = (m.err || !m.has_cg) ? 0xffffffffu : m.cg_len; /* no cigar to read: the four-wave shape's few lines */
        if (route < lo_bytes || route > hi_bytes) return;                     /* the other shape's entry (workgroup-uniform) */
    }
    const uint8_t *in = P.batch_in[m.pad1];
    const CovWalkSide S = cov_side(m, E.side);
    uint32_t *bm = P.bitmap + (E.bm_off - P.bm_base);
    const int64_t w0 = E.lo >> 5; /* base word of the bitmap */
    if (m.err) { /* the line did not parse: reported by k_cov_collect */
        if (tid == 0) P.aligned[e] = 0;
        return;
    }
    if (!m.has_cg) {
        /* tile: cigar_parse(NULL) -- the reference dereferences NULL (impl/paf_tile.c:166); to_bed: no cigar, nothing walked, only
           the end assert can fire (cigar_count(NULL) == 0, inc/paf.h:75) */
        if (tid == 0) {
            P.aligned[e] = 0;
            if (null_cigar_is_error) cov_fail(P, e, E.rec, PAFFY_ERR_NULL_CIGAR, 0);
            else if (S.start != S.end) cov_fail(P, e, E.rec, PAFFY_ERR_TILE_ASSERT, 2);
        }
        return;
    }
    const uint32_t cg_off = m.cg_off, end = m.cg_off + m.cg_len;
    const uint32_t a0 = cg_off & ~15u;
    if (tid < PAFFY_HALO / 4) reinterpret_cast<uint32_t *>(txt)[tid] = 0; /* nothing in front of the first tile */
    if (tid == 0) {
        sh->err_pos = 0xffffffffu;
        sh->flags = 0;
    }
    __syncthreads();
    int64_t cur = S.start; /* position of the next op on the walked sequence */
    int64_t aligned = 0;
    bool serial = false;
    for (uint32_t tb = a0; tb < end && !serial; tb += G::TEXT) {
        /* stage the tile; the last 32 bytes of the tile before it become the halo */
        uint4 h = make_uint4(0, 0, 0, 0);
        if (tb != a0 && tid < 2) h = reinterpret_cast<uint4 *>(txt + G::TEXT)[tid];
        __syncthreads();
        if (tb != a0 && tid < 2) reinterpret_cast<uint4 *>(txt)[tid] = h;
        const uint32_t g = tb + tid * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g < end) v = *reinterpret_cast<const uint4 *>(in + g);
        reinterpret_cast<uint4 *>(txt + PAFFY_HALO)[tid] = v;
        /* op letters in my 16 bytes: bytes of the cigar that are not digits */
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t opmask = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t nd = nondigit4(w[j]) >> 7;                          /* bits 0, 8, 16, 24 */
            nd = (nd | (nd >> 7) | (nd >> 14) | (nd >> 21)) & 0xfu;      /* -> bits 0..3 */
            opmask |= nd << (4 * j);
        }
        { /* bytes outside [cg_off, end) are not cigar text */
            const uint32_t first = g < cg_off ? (cg_off - g < 16u ? cg_off - g : 16u) : 0u;
            const uint32_t last = g + 16u > end ? (end > g ? end - g : 0u) : 16u;
            const uint32_t valid = (last >= 16u ? 0xffffu : ((1u << last) - 1u)) & ~((1u << first) - 1u);
            /* a cigar that ends in digits: the reference's switch sees the NUL (impl/paf.c:96-103) */
            if (last > 0 && last <= 16u && g + last == end && !((opmask >> (last - 1)) & 1u) && last > first) atomicMin(&sh->err_pos, end);
            opmask &= valid;
        }
        uint32_t cnt[1] = {(uint32_t)__popc(opmask)}, ctot[1];
        G::template scan_u32<1>(cnt, ctot, bc); /* also orders the staging stores before the reads below */
        if (ctot[0] > G::TEXT / 2) { /* more ops than the LDS list holds (ops without digits): the serial path */
            serial = true;
            break;
        }
        /* pass 1, in two steps (round 3, as the sizing parser: a thread used to convert the numbers of its own 16 bytes in a loop over
           its letters -- as many iterations as the densest 16 bytes of the wave hold letters): first every thread leaves the LDS
           positions of its letters in the op list, at the ops' indices; then the round's ops are dealt out evenly, a run of consecutive
           ops per thread -- the same run it walks again in pass 2 -- and converted where they stand */
        {
            uint32_t idx = cnt[0], mk = opmask;
            while (mk) {
                const uint32_t j = (uint32_t)__ffs((int)mk) - 1u;
                mk &= mk - 1u;
                ops[idx++] = PAFFY_HALO + tid * 16u + j;
            }
        }
        __syncthreads();
        const uint32_t n_round = ctot[0], per = (n_round + G::NT - 1u) / G::NT;
        const uint32_t i0 = tid * per < n_round ? tid * per : n_round, i1 = i0 + per < n_round ? i0 + per : n_round;
        int64_t sums[2] = {0, 0}, tots[2]; /* bases along the walked sequence, aligned bases */
        {
            uint32_t bad_at = 0xffffffffu, long_num = 0;
            for (uint32_t i = i0; i < i1; i++) {
                const uint32_t pos = (uint32_t)ops[i];
                const uint32_t c = txt[pos];
                int code = cov_op_code(c);
                if (code < 0) {
                    const uint32_t at = tb + (pos - PAFFY_HALO);
                    bad_at = at < bad_at ? at : bad_at;
                    code = OP_D;
                }
                uint32_t k;
                const uint32_t len = number_before(txt, pos, &k);
                long_num |= k >= 8u ? 1u : 0u; /* eight digits or more: the serial path decides */
                ops[i] = ((uint64_t)len << 8) | (uint64_t)code;
                if (code != S.skip_op) sums[0] += len;
                if (code == OP_M || code == OP_EQ || code == OP_X) sums[1] += len;
            }
            if (bad_at != 0xffffffffu) atomicMin(&sh->err_pos, bad_at);
            if (long_num) atomicOr(&sh->flags, 1u);
        }
        G::template scan_i64<2>(sums, tots, bc);
        if (sh->flags & 1u) { /* uniform: written before the scan's barrier */
            serial = true;
            break;
        }
        /* the bitmap words this round can touch */
        const int64_t r_lo = S.mirror ? S.start + S.end - (cur + tots[0]) : cur, r_hi = S.mirror ? S.start + S.end - cur : cur + tots[0];
        const int64_t c_lo = r_lo < E.lo ? E.lo : r_lo, c_hi = r_hi > E.hi ? E.hi : r_hi; /* inside the entry's range */
        const bool any = c_hi > c_lo && tots[1] > 0;
        const int64_t ww0 = c_lo >> 5;
        const uint64_t n_w = any ? (uint64_t)(((c_hi - 1) >> 5) - ww0 + 1) : 0;
        const bool in_lds = n_w <= G::WIN;
        if (any && in_lds)
            for (uint32_t i = tid; i < (uint32_t)n_w; i += G::NT) win[i] = 0;
        __syncthreads();
        /* pass 2: my ops again, from LDS, with their positions */
        if (any) {
            int64_t p = cur + sums[0];
            for (uint32_t i = i0; i < i1; i++) {
                const uint64_t o = ops[i];
                const int code = (int)(o & 0xffu);
                const int64_t len = (int64_t)(o >> 8);
                if (code == OP_M || code == OP_EQ || code == OP_X) {
                    if (len > 0) {
                        /* assert(i + j < query_end && i + j >= 0 && i + j < query_length), impl/paf.c:698 */
                        if (p < 0 || p + len > S.end || p + len > S.len) atomicOr(&sh->flags, 2u);
                        int64_t a = S.mirror ? S.start + S.end - (p + len) : p, b = a + len;
                        if (a < E.lo) a = E.lo;
                        if (b > E.hi) b = E.hi;
                        if (a < b) {
                            if (in_lds) bits_set_lds(win, ww0, a, b);
                            else bits_set_global(bm, w0, a, b);
                        }
                    }
                }
                if (code != S.skip_op) p += len;
            }
        }
        __syncthreads();
        if (any && in_lds) { /* window -> HBM: its first and last word may be shared with the rounds before and after */
            const uint64_t gw = (uint64_t)(ww0 - w0);
            for (uint32_t i = tid; i < (uint32_t)n_w; i += G::NT) {
                const uint32_t x = win[i];
                if (i == 0 || i + 1 == (uint32_t)n_w) {
                    if (x) atomicOr(&bm[gw + i], x);
                } else {
                    bm[gw + i] = x;
                }
            }
        }
        cur += tots[0];
        aligned += tots[1];
    }
    __syncthreads();
    if (serial) {
        /* a number of eight or more digits somewhere (leading zeros, lengths beyond 10^7): one thread walks the text the way
           cigar_parse does (impl/paf.c:86-107: value mod 2^64, then the 56-bit field) */
        if (tid == 0) {
            int64_t p = S.start, al = 0;
            uint32_t ep = 0xffffffffu, fl = 0;
            uint32_t q = cg_off;
            while (q < end) {
                uint64_t len = 0;
                while (q < end && (uint32_t)(in[q] - '0') < 10u) len = len * 10 + (uint32_t)(in[q++] - '0');
                int code = q < end ? cov_op_code(in[q]) : -1;
                if (code < 0) {
                    ep = q;
                    break;
                }
                const int64_t l56 = (int64_t)(len << 8) >> 8;
                if (code == OP_M || code == OP_EQ || code == OP_X) {
                    if (l56 > 0) {
                        if (p < 0 || p + l56 > S.end || p + l56 > S.len) fl |= 2u;
                        int64_t a = S.mirror ? S.start + S.end - (p + l56) : p, b = a + l56;
                        if (a < E.lo) a = E.lo;
                        if (b > E.hi) b = E.hi;
                        if (a < b && !(fl & 2u)) bits_set_global(bm, w0, a, b);
                        al += l56;
                    }
                }
                if (code != S.skip_op) p += l56;
                q++;
            }
            sh->err_pos = ep;
            sh->flags = fl;
            sh->bcast[0] = p;
            sh->bcast[1] = al;
        }
        __syncthreads();
        cur = sh->bcast[0];
        aligned = sh->bcast[1];
    }
    if (tid == 0) {
        const uint32_t ep = sh->err_pos, fl = sh->flags;
        P.aligned[e] = aligned;
        if (ep != 0xffffffffu) cov_fail(P, e, E.rec, PAFFY_ERR_CIGAR_CHAR, ep < end ? in[ep] : 0);
        else if ((fl & 2u) || cur != S.end) cov_fail(P, e, E.rec, PAFFY_ERR_TILE_ASSERT, 2); /* impl/paf.c:698, :708 */
    }
}

/* ---------------------------------------------------------------------------------------------------------------------- */
/* the slice walk                                                                                                          */
/* ---------------------------------------------------------------------------------------------------------------------- */

typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add_sat(uint32_t a, uint32_t b) { /* v_pk_add_u16 ... clamp */
    const u16x2 r = __builtin_elementwise_add_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ unsigned long long uniform_u64_cov(unsigned long long x) { /* lane 0's value in every lane */
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t wave_min_all_u32(uint32_t v) { return wave_min_u32(v); }
__device__ __forceinline__ uint32_t wave_max_all_u32(uint32_t v) { return ~wave_min_u32(~v); }

/*
 * Slice words and threads. Word w (bases [32 w, 32 w + 32) of the slice) belongs to wave w % 8, lane (w / 8) % 64, turn w / 512: any
 * run of consecutive words -- what an entry touches -- spreads evenly over the eight waves, so they reach the entry's barrier
 * together. In LDS the words of a wave are kept side by side (word w at index (w % 8) * 128 + w / 8 of the counter array): a wave's
 * 128-bit accesses then walk consecutive 64-byte rows, which the per-lane chunk rotation of cov_word makes conflict-free.
 */
__device__ __forceinline__ uint32_t cov_phys_word(uint32_t w) { return (w & 7u) * (COV_WORDS / 8u) + (w >> 3); }

#define COV_COPIES 8u /* level histogram copies (lane & 7): an eighth of the same-address collisions inside one ds_add */
#define COV_HIST_ROW (COV_HIST_W / 2 + 32u / COV_COPIES) /* words per copy: the spare ones put one level on a different bank in every copy */
/* a copy: COV_HIST_W 16-bit counters, levels L and L + COV_HIST_W / 2 in one word (a count is at most 32 768) */

struct CovWalkLds {
    uint16_t cnt[COV_SLICE];                    /* count + COV_BIAS, words in cov_phys_word order */
    uint32_t hist[2][COV_COPIES][COV_HIST_ROW]; /* level histograms of the entry being walked and the one before (being compacted) */
    uint32_t mn[4], mx[4];                      /* range of the new counts of entries j, j + 1, ... (index j & 3) */
    unsigned long long bcast;
};

/* histogram slot of a (biased) count: byte offset of its word inside a copy, and the increment that adds `n` to its half */
__device__ __forceinline__ void cov_hist_add(uint32_t *copy, uint32_t biased, uint32_t n) {
    const uint32_t lvl = biased - COV_BIAS;
    atomicAdd(&copy[lvl & (COV_HIST_W / 2 - 1u)], n << (((lvl / (COV_HIST_W / 2)) & 1u) * 16u));
}

/* one bitmap word of the slice (32 bases = four 16-byte chunks of counters) for one entry; `pw` = cov_phys_word of the word */
template <bool HIST>
__device__ __forceinline__ void cov_word(CovWalkLds &L, uint32_t pw, uint32_t bits, uint32_t rot, uint32_t *copy, uint32_t &tmin, uint32_t &tmax) {
    /* lane i takes its four chunks in the order (g + rot) mod 4, rot chosen by the caller so that neither the lane groups of the
       128-bit reads nor those of the 128-bit writes meet on a bank */
    uint4 *base = reinterpret_cast<uint4 *>(&L.cnt[pw * 32u]);
#pragma unroll
    for (uint32_t g = 0; g < 4; g++) {
        const uint32_t c = (g + rot) & 3u;
        const uint32_t b8 = (bits >> (8u * c)) & 0xffu;
        uint4 v = base[c];
        uint32_t d[4] = {v.x, v.y, v.z, v.w};
        if (HIST) {
            /* all eight old counts equal (the usual case at low coverage): one histogram update for the chunk */
            const uint32_t o0 = d[0];
            const uint32_t same = (o0 ^ d[1]) | (o0 ^ d[2]) | (o0 ^ d[3]) | (o0 ^ __builtin_amdgcn_alignbit(o0, o0, 16));
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t t = (b8 >> (2 * j)) & 3u;
                d[j] = pk_add_sat(d[j], (t | (t << 15)) & 0x00010001u);
                tmin = pk_min(tmin, d[j]);
                tmax = pk_max(tmax, d[j]);
            }
            if (same == 0) {
                cov_hist_add(copy, pk_add_sat(o0, 0x00010001u) & 0xffffu, (uint32_t)__popc(b8)); /* new count of the covered bases: old + 1 (clamped) */
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    cov_hist_add(copy, d[j] & 0xffffu, (b8 >> (2 * j)) & 1u); /* an uncovered base adds 0 */
                    cov_hist_add(copy, d[j] >> 16, (b8 >> (2 * j + 1)) & 1u);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t t = (b8 >> (2 * j)) & 3u;
                d[j] = pk_add_sat(d[j], (t | (t << 15)) & 0x00010001u);
            }
        }
        base[c] = make_uint4(d[0], d[1], d[2], d[3]);
    }
}

/* count of level `lvl` in histogram buffer h: the sum of the copies */
__device__ __forceinline__ uint32_t cov_hist_get(const uint32_t (*h)[COV_HIST_ROW], uint32_t lvl) {
    const uint32_t w = lvl & (COV_HIST_W / 2 - 1u), sh = ((lvl / (COV_HIST_W / 2)) & 1u) * 16u;
    uint32_t c = 0;
#pragma unroll
    for (uint32_t k = 0; k < COV_COPIES; k++) c += (h[k][w] >> sh) & 0xffffu;
    return c;
}

/* histogram of one entry, window [base, base + COV_HIST_W) only, into copy 0 as plain 32-bit counters (two words per... no: one per
   level, COV_HIST_W / 2 levels per pass): the rare entry whose new counts spread over more levels than the LDS histogram tells
   apart. The counters are final for this entry, so they are only read. */
__device__ __forceinline__ void cov_rehist(CovWalkLds &L, const uint32_t *bmw, uint32_t w_lo, uint32_t w_hi, uint32_t *flat, uint32_t base, uint32_t width) {
    for (uint32_t w = w_lo + threadIdx.x; w < w_hi; w += COV_NT) {
        const uint32_t bits = bmw[w], pw = cov_phys_word(w);
        for (uint32_t i = 0; i < 32; i++) {
            if (!((bits >> i) & 1u)) continue;
            const uint32_t lvl = (uint32_t)L.cnt[pw * 32u + i] - COV_BIAS;
            if (lvl >= base && lvl - base < width) atomicAdd(&flat[lvl - base], 1u);
        }
    }
}

/* backup: 32 Ki counters per item -- what the item's slice held before this launch (from_backup = 0: saved here; 1: loaded from
   there instead of from `cov`: the launch is a repeat, `cov` already holds the result of the first try) */
template <bool HIST>
__global__ __launch_bounds__(COV_NT, 4) void k_cov_walk(CovParams P, uint16_t *backup, int from_backup) { /* two workgroups per CU (LDS): four waves per SIMD */
    extern __shared__ uint4 smem4[];
    CovWalkLds &L = *reinterpret_cast<CovWalkLds *>(smem4);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t item = P.item_order ? P.item_order[blockIdx.x] : blockIdx.x;
    const uint32_t p0 = P.item_start[item], p1 = P.item_start[item + 1];
    const uint32_t gid = (uint32_t)(P.pairs[p0] >> 32);
    const uint32_t contig = P.entries[(uint32_t)P.pairs[p0]].contig;
    const uint32_t slice = gid - P.contig_slice0[contig];
    const int64_t s_lo = (int64_t)slice << COV_SLICE_SHIFT;
    const int64_t clen = P.contig_len[contig];
    const uint32_t live = clen - s_lo >= (int64_t)COV_SLICE ? COV_SLICE : (clen > s_lo ? (uint32_t)(clen - s_lo) : 0u); /* counters that exist */
    uint16_t *cov = P.cov + P.contig_cov[contig] + (uint64_t)s_lo;
    uint16_t *bak = backup ? backup + (uint64_t)item * COV_SLICE : nullptr;
    /* counters in: 16 bytes per lane (sequences start at multiples of 8 counters), + bias; base i of the slice lands in word
       cov_phys_word(i / 32) */
    for (uint32_t i = tid * 8u; i < COV_SLICE; i += COV_NT * 8u) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (from_backup) {
            v = *reinterpret_cast<const uint4 *>(bak + i);
        } else {
            if (i + 8u <= live) {
                v = *reinterpret_cast<const uint4 *>(cov + i);
            } else if (i < live) {
                uint32_t t[4] = {0, 0, 0, 0};
                for (uint32_t k = 0; i + k < live; k++) t[k >> 1] |= (uint32_t)cov[i + k] << (16u * (k & 1u));
                v = make_uint4(t[0], t[1], t[2], t[3]);
            }
            if (bak) *reinterpret_cast<uint4 *>(bak + i) = v;
        }
        const uint32_t bias2 = COV_BIAS | (COV_BIAS << 16);
        *reinterpret_cast<uint4 *>(&L.cnt[cov_phys_word(i >> 5) * 32u + (i & 31u)]) =
            make_uint4(v.x + bias2, v.y + bias2, v.z + bias2, v.w + bias2); /* counts <= 32766: no carry between halves */
    }
    if (HIST) {
        for (uint32_t i = tid; i < 2 * COV_COPIES * COV_HIST_ROW; i += COV_NT) (&L.hist[0][0][0])[i] = 0;
        if (tid < 4) {
            L.mn[tid] = 0xffffffffu;
            L.mx[tid] = 0;
        }
    }
    __syncthreads();
    /* chunk rotation of a lane (cov_word): the four 16-lane groups of ds_read_b128 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32;
       banks from the address mod 256) and the eight 8-lane groups of ds_write_b128 (contiguous lanes; banks from the address mod 128)
       must both see every 16-byte column once: lanes 64 bytes apart, bits b1 b2 b3 of the lane -> 2 (b2 ^ b3) + (b1 ^ b3) */
    const uint32_t rot = ((((lane >> 2) ^ (lane >> 3)) & 1u) << 1) | (((lane >> 1) ^ (lane >> 3)) & 1u);
    const uint32_t wa = 8u * lane + wave, wb = wa + COV_NT;          /* my two words of the slice ... */
    const uint32_t pa = wave * (COV_WORDS / 8u) + lane, pb = pa + 64u; /* ... and where their counters are */
    for (uint32_t p = p0; p < p1; p++) {
        const uint32_t j = p - p0;
        const uint32_t e = (uint32_t)P.pairs[p];
        const CovEntry &E = P.entries[e];
        const int64_t lo = E.lo > s_lo ? E.lo : s_lo, hi = E.hi < s_lo + (int64_t)COV_SLICE ? E.hi : s_lo + (int64_t)COV_SLICE;
        const uint32_t w_lo = (uint32_t)((lo - s_lo) >> 5), w_hi = (uint32_t)((hi - 1 - s_lo) >> 5) + 1u; /* slice words the entry touches */
        /* slice word w is word (s_lo >> 5) + w - (E.lo >> 5) of the entry's bitmap */
        const uint32_t *bmw = P.bitmap + ((int64_t)(E.bm_off - P.bm_base) + ((s_lo >> 5) - (E.lo >> 5)));
        uint32_t *copy = L.hist[j & 1u][lane & (COV_COPIES - 1u)];
        uint32_t tmin = 0xffffffffu, tmax = 0;
        const bool ha = wa >= w_lo && wa < w_hi, hb = wb >= w_lo && wb < w_hi;
        uint32_t ba = 0, bb = 0;
        if (ha) ba = bmw[wa]; /* words in front of the entry's bitmap are never read: wa >= w_lo */
        if (hb) bb = bmw[wb];
        if (ha) cov_word<HIST>(L, pa, ba, rot, copy, tmin, tmax);
        if (hb) cov_word<HIST>(L, pb, bb, rot, copy, tmin, tmax);
        if (!HIST) continue; /* to_bed: nothing is shared between entries */
        {
            uint32_t lo16 = tmin & 0xffffu, hi16 = tmin >> 16;
            const uint32_t mn = wave_min_all_u32(lo16 < hi16 ? lo16 : hi16);
            lo16 = tmax & 0xffffu;
            hi16 = tmax >> 16;
            const uint32_t mx = wave_max_all_u32(lo16 > hi16 ? lo16 : hi16);
            if (lane == 0 && mx >= mn) { /* a wave without a word of this entry has mn = 0xffff > mx = 0 */
                atomicMin(&L.mn[j & 3u], mn - COV_BIAS);
                atomicMax(&L.mx[j & 3u], mx - COV_BIAS);
            }
        }
        __syncthreads();
        const uint32_t mn = L.mn[j & 3u], mx = L.mx[j & 3u]; /* read by every wave before anything resets them (see below) */
        const uint32_t n = mx - mn + 1u;
        CovSlot *slot = P.slots + E.pair_base + (slice - E.first_slice);
        const uint32_t (*hbuf)[COV_HIST_ROW] = L.hist[j & 1u];
        if (n <= COV_HIST_W) {
            if (wave == (j & (COV_NWAVE - 1u))) { /* this wave compacts entry j while the others walk entry j + 1 */
                unsigned long long off = 0;
                if (lane == 0) off = atomicAdd(P.arena_used, (unsigned long long)n);
                off = uniform_u64_cov(off);
                const bool room = off + n <= P.arena_cap;
                for (uint32_t i = lane; i < n; i += 64) {
                    const uint32_t c = cov_hist_get(hbuf, mn + i);
                    if (room) P.arena[off + i] = (uint16_t)c;
                }
                /* the words used (levels mn .. mx, two per word) back to zero: after every read of this wave (in order) */
                const uint32_t nw = n < COV_HIST_W / 2 ? n : COV_HIST_W / 2;
                for (uint32_t i = lane; i < nw; i += 64)
#pragma unroll
                    for (uint32_t k = 0; k < COV_COPIES; k++) L.hist[j & 1u][k][(mn + i) & (COV_HIST_W / 2 - 1u)] = 0;
                if (lane == 0) {
                    slot->off = off;
                    slot->mn = mn;
                    slot->n = n;
                    /* the range words of entry j + 2 (last used by entry j - 2, read by every wave two barriers ago) */
                    L.mn[(j + 2u) & 3u] = 0xffffffffu;
                    L.mx[(j + 2u) & 3u] = 0;
                }
            }
        } else {
            /* spread beyond the histogram: the whole workgroup redoes this entry's histogram window by window (exact, rare) */
            __syncthreads(); /* every wave has read mn / mx */
            if (tid == 0) {
                const unsigned long long o = atomicAdd(P.arena_used, (unsigned long long)n);
                slot->off = o;
                slot->mn = mn;
                slot->n = n;
                L.mn[(j + 2u) & 3u] = 0xffffffffu;
                L.mx[(j + 2u) & 3u] = 0;
                L.bcast = o;
            }
            uint32_t *flat = &L.hist[j & 1u][0][0]; /* the buffer as COV_COPIES * COV_HIST_W / 2 plain counters */
            const uint32_t width = COV_COPIES * (COV_HIST_W / 2);
            /* the WHOLE buffer goes back to zero, spare words included: a copy starts every COV_HIST_ROW words, so the walk above left this
               entry's (aliased) levels in words up to COV_COPIES * COV_HIST_ROW. Rounds 2-3 cleared `width` words only: the last 28 words of
               the last copy kept the counts and entry j + 2 read them as levels 100..127 (mod 128) of its own -- found by the soak at
               445-fold coverage (tests/golden/fuzz/tile_r3_fail.paf), where an entry's levels first spread over more than COV_HIST_W */
            const uint32_t whole = COV_COPIES * COV_HIST_ROW;
            for (uint32_t i = tid; i < whole; i += COV_NT) flat[i] = 0;
            __syncthreads();
            const unsigned long long off = L.bcast;
            const bool room = off + n <= P.arena_cap;
            for (uint32_t base = mn; base <= mx; base += width) {
                cov_rehist(L, bmw, w_lo, w_hi, flat, base, width);
                __syncthreads();
                for (uint32_t i = tid; i < width && base + i <= mx; i += COV_NT) {
                    if (room) P.arena[off + (base - mn) + i] = (uint16_t)flat[i];
                }
                __syncthreads();
                for (uint32_t i = tid; i < whole; i += COV_NT) flat[i] = 0;
                __syncthreads();
            }
        }
    }
    __syncthreads();
    /* counters out */
    for (uint32_t i = tid * 8u; i < COV_SLICE; i += COV_NT * 8u) {
        if (i >= live) break;
        const uint4 v = *reinterpret_cast<const uint4 *>(&L.cnt[cov_phys_word(i >> 5) * 32u + (i & 31u)]);
        const uint32_t bias2 = COV_BIAS | (COV_BIAS << 16);
        const uint32_t t[4] = {v.x - bias2, v.y - bias2, v.z - bias2, v.w - bias2};
        if (i + 8u <= live) {
            *reinterpret_cast<uint4 *>(cov + i) = make_uint4(t[0], t[1], t[2], t[3]);
        } else {
            for (uint32_t k = 0; i + k < live; k++) cov[i + k] = (uint16_t)(t[k >> 1] >> (16u * (k & 1u)));
        }
    }
}

/* one wave per entry: the entry's partial histograms, smallest level first, until half of its aligned bases are covered
   (impl/paf_tile.c:81-88: the first level with cumulative >= matches / 2.0; no aligned base: INT16_MAX) */
__global__ __launch_bounds__(PAFFY_NT) void k_cov_merge(CovParams P) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t e = blockIdx.x * PAFFY_NWAVE + (threadIdx.x >> 6);
    if (e >= P.n_entries) return;
    if ((P.info->first_err_key >> 16) <= e) return; /* nothing is written anyway */
    const CovEntry &E = P.entries[e];
    const int64_t aligned = P.aligned[e];
    int64_t level = 32767;
    if (aligned > 0) {
        const CovSlot *sl = P.slots + E.pair_base;
        uint32_t gmn = 0xffffffffu, gmx = 0;
        for (uint32_t s = lane; s < E.n_slices; s += 64)
            if (sl[s].n) {
                gmn = sl[s].mn < gmn ? sl[s].mn : gmn;
                gmx = sl[s].mn + sl[s].n - 1u > gmx ? sl[s].mn + sl[s].n - 1u : gmx;
            }
        gmn = wave_min_all_u32(gmn);
        gmx = wave_max_all_u32(gmx);
        int64_t acc = 0;
        level = -1;
        for (uint32_t base = gmn; base <= gmx && level < 0; base += 64) {
            const uint32_t lv = base + lane;
            int64_t c = 0;
            for (uint32_t s = 0; s < E.n_slices; s++) {
                const uint32_t mn = sl[s].mn, n = sl[s].n; /* wave-uniform */
                if (lv >= mn && lv - mn < n) c += P.arena[sl[s].off + (lv - mn)];
            }
            const int64_t inc = wave_incl_scan(c);
            const unsigned long long hit = __ballot(2 * (acc + inc) >= aligned);
            if (hit) level = (int64_t)base + (__ffsll((long long)hit) - 1);
            acc += wave_last(inc);
            if (base + 64 < base) break;
        }
        if (level <= 0) { /* assert(i > 0) / assert(0), impl/paf_tile.c:86-90 */
            if (lane == 0) cov_fail(P, e, E.rec, PAFFY_ERR_TILE_ASSERT, 3);
            return;
        }
    }
    if (lane == 0) P.level[E.rec] = level;
}

/* ---------------------------------------------------------------------------------------------------------------------- */
/* small kernels around the walk                                                                                           */
/* ---------------------------------------------------------------------------------------------------------------------- */

/* Names are grouped by a 64-bit hash and then CHECKED byte for byte against the head of their group (k_cov_verify_names; the reference
   keys its tables by string equality, impl/paf_tile.c:160-161 stHash_stringEqualKey, impl/paf.c:675-688): when two different names
   share a hash the grouping is redone with the next salt. Salt 0 is the hash that travels between ranks (shard.name_hash). */
__device__ __forceinline__ uint64_t cov_name_hash(const uint8_t *in, uint32_t off, uint32_t len, uint32_t salt = 0) {
    uint64_t h = 0xcbf29ce484222325ull ^ ((uint64_t)salt * 0x9E3779B97F4A7C15ull); /* FNV-1a over the name, then its length */
    for (uint32_t i = 0; i < len; i++) h = (h ^ in[off + i]) * 0x100000001b3ull;
    h = (h ^ (0x100u + len)) * 0x100000001b3ull;
    return h ^ (h >> 29);
}
static inline __host__ __device__ uint64_t cov_desc_key(int64_t x) { return ~((uint64_t)x ^ 0x8000000000000000ull); } /* ascending in this = descending in x */

/* batch metadata -> the concatenated array (pad1 = batch) + the sort keys of `paffy tile` (paf_cmp_by_descending_score, impl/paf_tile.c:28-34) */
__global__ __launch_bounds__(PAFFY_NT) void k_cov_collect(const RecMeta *src, uint32_t n, uint32_t batch, RecMeta *dst, uint64_t *key_score, uint64_t *key_chain,
                                                           uint32_t *idx, uint32_t first, uint32_t sides, DevInfo *info) {
    const uint32_t r = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (r >= n) return;
    RecMeta m = src[r];
    m.pad1 = batch;
    dst[r] = m;
    if (key_score) {
        key_score[r] = cov_desc_key(m.score);
        key_chain[r] = cov_desc_key(m.chain_score);
    }
    idx[r] = first + r;
    /* read_pafs / the read loop parses every line before anything else happens: the first bad line in input order wins */
    /* key = rank of the record's first entry << 16 | code (stage bits 0: found while parsing) */
    if (m.err) atomicMin(&info->first_err_key, ((unsigned long long)(first + r) * sides << 16) | (unsigned long long)m.err);
}
__global__ __launch_bounds__(PAFFY_NT) void k_gather_u64(const uint64_t *src, const uint32_t *idx, uint32_t n, uint64_t *dst) {
    const uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

/* entry e (visiting order) -> record, side, clamped range, sequence name hash, sequence length */
__global__ __launch_bounds__(PAFFY_NT) void k_cov_entry_init(CovParams P, const uint32_t *order, uint32_t sides, uint64_t *name_hash, int64_t *seq_len, uint32_t salt) {
    const uint32_t e = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (e >= P.n_entries) return;
    const uint32_t rec = order ? order[e / sides] : e / sides;
    const RecMeta &m = P.meta[rec];
    const uint32_t side = (e % sides) == 0 ? 0u : (m.same_strand ? 1u : 2u);
    const CovWalkSide S = cov_side(m, side);
    CovEntry E;
    E.rec = rec;
    E.side = side;
    int64_t lo = S.start < 0 ? 0 : S.start;
    const int64_t top = S.len > 0 ? S.len : 0;
    if (lo > top) lo = top;
    int64_t hi = S.end > top ? top : S.end;
    if (hi < lo) hi = lo;
    if (!m.has_cg || m.cg_len == 0 || m.err) hi = lo; /* nothing can be aligned */
    E.lo = lo;
    E.hi = hi;
    E.bm_off = 0;
    E.contig = 0;
    E.pair_base = 0;
    E.first_slice = (uint32_t)(lo >> COV_SLICE_SHIFT);
    E.n_slices = hi > lo ? (uint32_t)(((hi - 1) >> COV_SLICE_SHIFT) - (lo >> COV_SLICE_SHIFT)) + 1u : 0u;
    P.entries[e] = E;
    const uint8_t *in = P.batch_in[m.pad1];
    name_hash[e] = side ? cov_name_hash(in, m.tname_off, m.tname_len, salt) : cov_name_hash(in, m.qname_off, m.qname_len, salt);
    seq_len[e] = S.len;
}
/* the name behind a hash: every entry against the first entry of its sequence, byte for byte; *collide != 0: two names share a hash */
__device__ __forceinline__ bool cov_same_bytes(const uint8_t *a, uint32_t la, const uint8_t *b, uint32_t lb) {
    if (la != lb) return false;
    for (uint32_t i = 0; i < la; i++)
        if (a[i] != b[i]) return false;
    return true;
}
__global__ __launch_bounds__(PAFFY_NT) void k_cov_verify_names(CovParams P, const uint32_t *first_entry, uint32_t *collide) {
    const uint32_t e = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (e >= P.n_entries) return;
    const CovEntry &E = P.entries[e];
    const uint32_t h = first_entry[E.contig];
    if (h == e) return;
    const CovEntry &H = P.entries[h];
    const RecMeta &m = P.meta[E.rec], &mh = P.meta[H.rec];
    const uint8_t *a = P.batch_in[m.pad1] + (E.side ? m.tname_off : m.qname_off), *b = P.batch_in[mh.pad1] + (H.side ? mh.tname_off : mh.qname_off);
    if (!cov_same_bytes(a, E.side ? m.tname_len : m.qname_len, b, H.side ? mh.tname_len : mh.qname_len)) atomicOr(collide, 1u);
}
/* sorted name hashes -> sequence ids: flag[i] = first of its run */
__global__ __launch_bounds__(PAFFY_NT) void k_cov_run_heads(const uint64_t *sorted, uint32_t n, uint32_t *flag) {
    const uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || sorted[i] != sorted[i - 1]) ? 1u : 0u;
}
/* after an inclusive scan of the flags: entry ids of sorted position i gets sequence id scan[i] - 1; the first entry (smallest
   index = first in visiting order) of every sequence is remembered: its length is the sequence's (impl/paf.c:680) */
__global__ __launch_bounds__(PAFFY_NT) void k_cov_assign_contig(CovParams P, const uint32_t *sorted_entry, const uint32_t *flag, const uint32_t *scan, uint32_t n,
                                                                 uint32_t *first_entry) {
    const uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = scan[i] - 1u, e = sorted_entry[i];
    P.entries[e].contig = c;
    if (flag[i]) first_entry[c] = e; /* the sort is stable and the entries went in by ascending index: a run's head is its smallest */
}
__global__ __launch_bounds__(PAFFY_NT) void k_cov_contig_len(const uint32_t *first_entry, const int64_t *seq_len, uint32_t n_contigs, int64_t *contig_len) {
    const uint32_t c = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (c < n_contigs) contig_len[c] = seq_len[first_entry[c]];
}
/* assert(seq_count_array->length == paf->query_length), impl/paf.c:685; bitmap words and pairs of every entry */
__global__ __launch_bounds__(PAFFY_NT) void k_cov_entry_sizes(CovParams P, const int64_t *seq_len, uint64_t *bm_words, uint32_t *n_pairs) {
    const uint32_t e = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (e >= P.n_entries) return;
    CovEntry &E = P.entries[e];
    if (seq_len[e] != P.contig_len[E.contig]) {
        cov_fail(P, e, E.rec, PAFFY_ERR_TILE_ASSERT, 1);
        E.hi = E.lo; /* a sequence of another length: nothing of it is walked here */
        E.n_slices = 0;
    }
    bm_words[e] = E.hi > E.lo ? (uint64_t)(((E.hi - 1) >> 5) - (E.lo >> 5)) + 1ull : 0ull;
    n_pairs[e] = E.n_slices;
}
__global__ __launch_bounds__(PAFFY_NT) void k_cov_entry_offsets(CovParams P, const uint64_t *bm_off, const uint64_t *pair_off) {
    const uint32_t e = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (e >= P.n_entries) return;
    P.entries[e].bm_off = bm_off[e];
    P.entries[e].pair_base = (uint32_t)pair_off[e];
}
/* the (slice, entry) pairs of the chunk's entries */
__global__ __launch_bounds__(PAFFY_NT) void k_cov_pairs(CovParams P, uint32_t pair0, uint64_t *keys) {
    const uint32_t e = P.e0 + blockIdx.x * PAFFY_NT + threadIdx.x;
    if (e >= P.e1) return;
    const CovEntry &E = P.entries[e];
    const uint32_t g0 = P.contig_slice0[E.contig] + E.first_slice;
    for (uint32_t s = 0; s < E.n_slices; s++) keys[E.pair_base - pair0 + s] = ((uint64_t)(g0 + s) << 32) | e;
}
__global__ __launch_bounds__(PAFFY_NT) void k_cov_item_heads(const uint64_t *pairs, uint32_t n, uint32_t *flag) {
    const uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || (pairs[i] >> 32) != (pairs[i - 1] >> 32)) ? 1u : 0u;
}
__global__ __launch_bounds__(PAFFY_NT) void k_cov_item_starts(const uint32_t *flag, const uint32_t *scan, uint32_t n, uint32_t *item_start) {
    const uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i < n && flag[i]) item_start[scan[i] - 1u] = i;
    if (i == 0) item_start[scan[n - 1]] = n;
}
/* items by descending size: the heavy slices start first */
__global__ __launch_bounds__(PAFFY_NT) void k_cov_item_sizes(const uint32_t *item_start, uint32_t n_items, uint32_t *key, uint32_t *idx) {
    const uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i >= n_items) return;
    key[i] = ~(item_start[i + 1] - item_start[i]);
    idx[i] = i;
}

#endif
