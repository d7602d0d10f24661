/*
 * device_util.h -- wave64 / LDS building blocks for the gfx950 PAF kernels.
 *
 *  - wave64 scans / reductions on DPP row shifts and broadcasts (no LDS traffic), combined across
 *    the 4 waves of a 256-thread workgroup with one LDS hop and a single barrier
 *  - decimal digit counting and packed ASCII conversion (impl/paf.c:10-34 int64_to_str semantics)
 *  - RingWriter: a per-lane byte stream funnelled into aligned 8-byte stores of an LDS ring whose
 *    address is (global output offset mod RING), so that a workgroup's output leaves LDS as full
 *    16-byte coalesced global stores whatever the byte alignment of the rows inside it.
 */
#ifndef PAFFY_DEVICE_UTIL_H_
#define PAFFY_DEVICE_UTIL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

/* PAFFY_NT (threads per record workgroup) and PAFFY_NWAVE are set by record_groups.h, which compiles the record program twice:
   for workgroups of four waves (namespace g256, the default everywhere else) and of one wave (g64) */
#define PAFFY_RING 32768u            /* LDS output ring bytes (power of two) */

/* ---------------- wave64 DPP primitives ---------------- */

/*
 * Cross-lane data movement with DPP modifiers (no LDS round trip, unlike ds_bpermute-based
 * __shfl): row_shr:n within rows of 16 lanes, then row_bcast:15 / row_bcast:31 to finish a
 * 64-lane inclusive scan in 7 steps.
 */
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_BCAST15 0x142
#define DPP_BCAST31 0x143

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int64_t dpp_mov_i64(int64_t x) { /* lanes without a source receive 0 */
    int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, CTRL, ROW_MASK, BANK_MASK, false);
    int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)((uint64_t)x >> 32), CTRL, ROW_MASK, BANK_MASK, false);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

/* inclusive prefix sum over the 64 lanes of a wave */
__device__ __forceinline__ int64_t wave_incl_scan(int64_t v) {
    int64_t x = v;
    x += dpp_mov_i64<DPP_ROW_SHR(1), 0xf, 0xf>(v);
    x += dpp_mov_i64<DPP_ROW_SHR(2), 0xf, 0xf>(v);
    x += dpp_mov_i64<DPP_ROW_SHR(3), 0xf, 0xf>(v);
    x += dpp_mov_i64<DPP_ROW_SHR(4), 0xf, 0xe>(x);
    x += dpp_mov_i64<DPP_ROW_SHR(8), 0xf, 0xc>(x);
    x += dpp_mov_i64<DPP_BCAST15, 0xa, 0xf>(x);
    x += dpp_mov_i64<DPP_BCAST31, 0xc, 0xf>(x);
    return x;
}
__device__ __forceinline__ int64_t wave_last(int64_t x) { /* value held by lane 63, wave-uniform */
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, 63);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)x >> 32), 63);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
/* 32-bit variants: one v_add_u32 with a DPP operand per step */
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ uint32_t dpp_mov_u32(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    uint32_t x = v;
    x += dpp_mov_u32<DPP_ROW_SHR(1), 0xf, 0xf>(v);
    x += dpp_mov_u32<DPP_ROW_SHR(2), 0xf, 0xf>(v);
    x += dpp_mov_u32<DPP_ROW_SHR(3), 0xf, 0xf>(v);
    x += dpp_mov_u32<DPP_ROW_SHR(4), 0xf, 0xe>(x);
    x += dpp_mov_u32<DPP_ROW_SHR(8), 0xf, 0xc>(x);
    x += dpp_mov_u32<DPP_BCAST15, 0xa, 0xf>(x);
    x += dpp_mov_u32<DPP_BCAST31, 0xc, 0xf>(x);
    return x;
}
__device__ __forceinline__ uint32_t wave_last_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)x, 63); }
/* min over the wave (same 7-step pattern; lanes without a source keep their own value) */
__device__ __forceinline__ int64_t wave_min(int64_t v) {
    int64_t x = v;
#define PAFFY_MIN_STEP(CTRL, RM, BM, SRC)                                                            \
    {                                                                                                \
        int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)x, (int)(uint32_t)(SRC), CTRL, RM, BM, false); \
        int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)((uint64_t)x >> 32), (int)(uint32_t)((uint64_t)(SRC) >> 32), CTRL, RM, BM, false); \
        int64_t t = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);                        \
        x = t < x ? t : x;                                                                           \
    }
    PAFFY_MIN_STEP(DPP_ROW_SHR(1), 0xf, 0xf, v)
    PAFFY_MIN_STEP(DPP_ROW_SHR(2), 0xf, 0xf, v)
    PAFFY_MIN_STEP(DPP_ROW_SHR(3), 0xf, 0xf, v)
    PAFFY_MIN_STEP(DPP_ROW_SHR(4), 0xf, 0xe, x)
    PAFFY_MIN_STEP(DPP_ROW_SHR(8), 0xf, 0xc, x)
    PAFFY_MIN_STEP(DPP_BCAST15, 0xa, 0xf, x)
    PAFFY_MIN_STEP(DPP_BCAST31, 0xc, 0xf, x)
#undef PAFFY_MIN_STEP
    return wave_last(x);
}

/* exclusive scan inside one wave; tot[] = wave totals (wave-uniform). No LDS, no barrier. */
template <int K>
__device__ __forceinline__ void wave_excl_scan(int64_t (&v)[K], int64_t (&tot)[K]) {
#pragma unroll
    for (int k = 0; k < K; k++) {
        int64_t inc = wave_incl_scan(v[k]);
        tot[k] = wave_last(inc);
        v[k] = inc - v[k];
    }
}

/* ---------------- decimal helpers ---------------- */

__device__ __constant__ uint64_t PAFFY_P10[20] = {1ull,
                                                  10ull,
                                                  100ull,
                                                  1000ull,
                                                  10000ull,
                                                  100000ull,
                                                  1000000ull,
                                                  10000000ull,
                                                  100000000ull,
                                                  1000000000ull,
                                                  10000000000ull,
                                                  100000000000ull,
                                                  1000000000000ull,
                                                  10000000000000ull,
                                                  100000000000000ull,
                                                  1000000000000000ull,
                                                  10000000000000000ull,
                                                  100000000000000000ull,
                                                  1000000000000000000ull,
                                                  10000000000000000000ull};

/* Characters int64_to_str (impl/paf.c:10-34) writes for v: digits plus a leading '-'. */
__device__ __forceinline__ int dec_len(int64_t v) {
    uint64_t u = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
    int d;
    if ((u >> 32) == 0) { /* common case: immediate compares, no memory access */
        uint32_t x = (uint32_t)u;
        d = 1 + (x >= 10u) + (x >= 100u) + (x >= 1000u) + (x >= 10000u) + (x >= 100000u) + (x >= 1000000u) + (x >= 10000000u) +
            (x >= 100000000u) + (x >= 1000000000u);
    } else {
        int bits = 64 - __clzll((long long)u);
        int t = (bits * 1233) >> 12;
        d = t + (u >= PAFFY_P10[t] ? 1 : 0);
    }
    return d + (v < 0 ? 1 : 0);
}

/* 4 decimal digits of y < 10000 as bytes, most significant digit in byte 0. All products fit
 * 24 x 24 -> 32 bits, so they use the full-rate v_mul_u32_u24 (32-bit mul_hi/mul_lo are slower). */
__device__ __forceinline__ uint32_t bcd4(uint32_t y) {
    uint32_t a = __umul24(y, 5243u) >> 19; /* y / 100 for y < 43699 */
    uint32_t b = y - __umul24(a, 100u);
    uint32_t d0 = __umul24(a, 103u) >> 10; /* a / 10 for a < 100 */
    uint32_t d1 = a - __umul24(d0, 10u);
    uint32_t d2 = __umul24(b, 103u) >> 10;
    uint32_t d3 = b - __umul24(d2, 10u);
    return d0 | (d1 << 8) | (d2 << 16) | (d3 << 24);
}

/* ---------------- cigar numbers, eight bytes at a time ---------------- */

/* bytes of w that are not ASCII digits: bit 7 of each such byte */
__device__ __forceinline__ uint32_t nondigit4(uint32_t w) {
    const uint32_t t = w ^ 0x30303030u;
    return (((t & 0x7f7f7f7fu) + 0x76767676u) | t) & 0x80808080u;
}
/* value of four digit bytes (values 0..9, most significant in byte 0) */
__device__ __forceinline__ uint32_t swar4(uint32_t x) {
    const uint32_t p = (x * 10u + (x >> 8)) & 0x00ff00ffu; /* two 2-digit numbers */
    return __umul24(p & 0xffu, 100u) + (p >> 16);
}
/*
 * The number that ends right before position q of a text staged in LDS (txt[q - 1] is its last digit, txt[q] the op letter): at most
 * 8 digits are converted. Returns the digit count (0..8) in *k; 8 means "eight or more": callers take their general path for such
 * a record. txt must be readable from q - 8 (a halo in front of the text) and 4-byte aligned at index 0.
 */
__device__ __forceinline__ uint32_t number_before(const uint8_t *txt, uint32_t q, uint32_t *k, uint32_t *letter = nullptr) {
    const uint32_t a = q - 8u, sh = a & 3u;
    const uint32_t *wp = reinterpret_cast<const uint32_t *>(txt + (a - sh));
    const uint32_t d0 = wp[0], d1 = wp[1], d2 = wp[2];
    if (letter) *letter = (d2 >> (8u * sh)) & 0xffu; /* txt[q]: the op letter itself */
    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh), hi = __builtin_amdgcn_alignbyte(d2, d1, sh); /* bytes q-8..q-5, q-4..q-1 */
    const uint32_t nh = nondigit4(hi);
    uint32_t kk = nh ? (uint32_t)__clz((int)nh) >> 3 : 4u; /* digits at the top of hi */
    uint32_t v;
    if (__all(nh != 0)) { /* the usual case, whole wave: numbers of at most three digits */
        /* xor, not minus: a byte below '0' in front of the digits would borrow from them */
        const uint32_t x = kk ? ((hi ^ 0x30303030u) & (0xffffffffu << (32u - 8u * kk))) : 0u;
        v = swar4(x);
    } else {
        uint32_t xl = 0;
        if (!nh) {
            const uint32_t nl = nondigit4(lo);
            const uint32_t kl = nl ? (uint32_t)__clz((int)nl) >> 3 : 4u;
            kk = 4u + kl;
            xl = kl ? ((lo ^ 0x30303030u) & (0xffffffffu << (32u - 8u * kl))) : 0u;
        }
        const uint32_t xh = kk >= 4u ? hi ^ 0x30303030u : (kk ? ((hi ^ 0x30303030u) & (0xffffffffu << (32u - 8u * kk))) : 0u);
        v = swar4(xl) * 10000u + swar4(xh);
    }
    *k = kk;
    return v;
}

/* ---------------- byte sinks ---------------- */

/*
 * RingWriter: one lane's contiguous byte stream [s, e) of the workgroup's output window, funnelled
 * through a 64-bit accumulator into aligned 8-byte LDS stores of a ring indexed by
 * (output offset mod ring size). Two phases per window:
 *   phase 1 (put): full 8-byte words only. A lane's first word also covers the (< 8) bytes before
 *     s in that word; they belong to the tails of earlier lanes and are rewritten in phase 2.
 *   barrier
 *   phase 2 (tail): the (< 8) bytes after the last full word, as 4/2/1-byte stores.
 * A lane whose first word reaches back before the window start preloads the previous window's
 * pending bytes found there.
 * put(w, n): append the low n (0..8) bytes of w; bytes of w above n must be zero.
 */
struct RingWriter {
    uint8_t *ring;
    uint64_t acc;
    uint32_t nacc, head;
    uint32_t rpos, rsize; /* ring cursor (multiple of 8, < rsize); the ring size need not be a power of two */
    /*
     * The window starts at ring index p0r (< ring_bytes, congruent to the output offset mod 16);
     * this lane's stream starts `off` (< ring_bytes) bytes into the window.
     */
    __device__ __forceinline__ void init(uint8_t *r, uint32_t ring_bytes, uint32_t p0r, uint32_t off) {
        ring = r;
        rsize = ring_bytes;
        uint32_t s = p0r + off;
        if (s >= ring_bytes) s -= ring_bytes;
        head = s & 7u;
        rpos = s - head;
        nacc = head;
        acc = 0;
        if (head > off) /* the first word reaches back before the window: keep the previous window's pending bytes */
            acc = *reinterpret_cast<const uint64_t *>(ring + rpos) & ((1ull << (8 * (head - off))) - 1ull);
    }
    __device__ __forceinline__ void store(uint64_t t) {
        *reinterpret_cast<uint64_t *>(ring + rpos) = t;
        rpos += 8;
        if (rpos == rsize) rpos = 0;
        head = 0;
    }
    __device__ __forceinline__ void put(uint64_t w, uint32_t n) {
        const uint32_t sh = nacc * 8;
        const uint64_t t = acc | (w << sh);
        nacc += n;
        if (nacc >= 8) {
            store(t);
            acc = (w >> 1) >> (63 - sh); /* = w >> (64 - sh), and 0 for sh == 0 */
            nacc -= 8;
        } else {
            acc = t;
        }
    }
    __device__ __forceinline__ void put8(uint64_t w) { /* exactly 8 bytes: always one full word out */
        const uint32_t sh = nacc * 8;
        store(acc | (w << sh));
        acc = (w >> 1) >> (63 - sh);
    }
    __device__ __forceinline__ void tail() { /* phase 2 */
        uint8_t *p = ring + rpos;
        if (head == 0) {
            uint32_t b = 0;
            if (nacc & 4u) {
                *reinterpret_cast<uint32_t *>(p) = (uint32_t)acc;
                b = 4;
            }
            if (nacc & 2u) {
                *reinterpret_cast<uint16_t *>(p + b) = (uint16_t)(acc >> (8 * b));
                b += 2;
            }
            if (nacc & 1u) p[b] = (uint8_t)(acc >> (8 * b));
        } else { /* no full word was stored: only bytes [head, nacc) are ours */
            for (uint32_t b = head; b < nacc; b++) p[b] = (uint8_t)(acc >> (8 * b));
        }
        nacc = 0;
    }
};

/* Plain byte-at-a-time sink: LDS header builder and the sequential fallbacks. */
struct ByteWriter {
    uint8_t *p;
    uint32_t n;
    __device__ __forceinline__ void put(uint64_t w, uint32_t k) {
        for (uint32_t b = 0; b < k; b++) p[n + b] = (uint8_t)(w >> (8 * b));
        n += k;
    }
    __device__ __forceinline__ void put8(uint64_t w) { put(w, 8); }
};

/* 8 decimal digits of x < 10^8 as ASCII, most significant digit in byte 0 */
__device__ __forceinline__ uint64_t ascii8(uint32_t x) {
    uint32_t hi4 = x / 10000u, lo4 = x - __umul24(hi4, 10000u);
    return ((((uint64_t)bcd4(lo4)) << 32) | bcd4(hi4)) + 0x3030303030303030ull;
}
/* decimal digits of x < 10^8 without leading zeros, left-aligned; *n = digit count (1..8) */
__device__ __forceinline__ uint64_t ascii_upto8(uint32_t x, uint32_t *n) {
    uint32_t w0 = 0, w1;
    if (x < 10000u) { /* one group of four: no 32-bit division at all */
        w1 = bcd4(x);
    } else {
        uint32_t hi4 = x / 10000u, lo4 = x - __umul24(hi4, 10000u);
        w0 = bcd4(hi4);
        w1 = bcd4(lo4);
    }
    uint32_t z = w0 ? ((uint32_t)__ffs((int)w0) - 1) >> 3 : (w1 ? 4 + (((uint32_t)__ffs((int)w1) - 1) >> 3) : 7);
    *n = 8 - z;
    return ((((uint64_t)w1 << 32) | w0) + 0x3030303030303030ull) >> (8 * z);
}

/* A number pre-rendered as up to three 8-byte groups (int64_to_str, impl/paf.c:10-34). */
struct DecText {
    uint64_t top;  /* sign and leading group, left-aligned */
    uint32_t ntop; /* bytes in top (1..8); a '-' that does not fit goes out on its own */
    uint32_t g1, g0; /* following full 8-digit groups */
    uint32_t groups; /* 0..2 */
    bool neg_separate;
};
__device__ __forceinline__ void dec_text(int64_t v, DecText &d) {
    uint64_t u = (uint64_t)v;
    const bool neg = v < 0;
    if (neg) u = (uint64_t)0 - u;
    uint32_t top;
    d.g1 = d.g0 = 0;
    d.groups = 0;
    if ((u >> 32) == 0) {
        uint32_t x = (uint32_t)u;
        if (x < 100000000u) {
            top = x;
        } else {
            top = x / 100000000u;
            d.g0 = x - top * 100000000u;
            d.groups = 1;
        }
    } else if (u < 10000000000000000ull) {
        uint64_t q = u / 100000000ull;
        top = (uint32_t)q;
        d.g0 = (uint32_t)(u - q * 100000000ull);
        d.groups = 1;
    } else {
        uint64_t q = u / 10000000000000000ull, r = u - q * 10000000000000000ull;
        uint64_t r1 = r / 100000000ull;
        top = (uint32_t)q;
        d.g1 = (uint32_t)r1;
        d.g0 = (uint32_t)(r - r1 * 100000000ull);
        d.groups = 2;
    }
    d.top = ascii_upto8(top, &d.ntop);
    d.neg_separate = false;
    if (neg) {
        if (d.ntop < 8) {
            d.top = (d.top << 8) | (uint64_t)'-';
            d.ntop += 1;
        } else {
            d.neg_separate = true;
        }
    }
}
__device__ __forceinline__ uint32_t text_len(const DecText &d) { return d.ntop + 8 * d.groups + (d.neg_separate ? 1u : 0u); }
/* lead: one extra character in front (0 = none), folded into the first group when it fits */
template <class SINK>
__device__ __forceinline__ void put_text(SINK &s, const DecText &d, uint32_t lead) {
    uint64_t t = d.top;
    uint32_t n = d.ntop;
    if (d.neg_separate) {
        s.put(lead ? ((uint64_t)'-' << 8) | lead : (uint64_t)'-', lead ? 2 : 1);
    } else if (lead) {
        if (n < 8) {
            t = (t << 8) | lead;
            n += 1;
        } else {
            s.put(lead, 1);
        }
    }
    s.put(t, n);
#pragma unroll 1
    for (uint32_t g = d.groups; g > 0; g--) s.put8(ascii8(g == 2 ? d.g1 : d.g0));
}
template <class SINK>
__device__ __forceinline__ void put_dec(SINK &s, int64_t v) {
    DecText d;
    dec_text(v, d);
    put_text(s, d, 0);
}
/* Append len bytes of an 8-byte-aligned LDS string. */
template <class SINK>
__device__ __forceinline__ void put_lds(SINK &s, const uint64_t *src, uint32_t len) {
    const uint32_t full = len >> 3;
#pragma unroll 1
    for (uint32_t k = 0; k < full; k++) s.put8(src[k]);
    const uint32_t rem = len & 7u;
    if (rem) s.put(src[full] & ((1ull << (8 * rem)) - 1ull), rem);
}

#endif
