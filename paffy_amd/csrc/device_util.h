/*
 * device_util.h -- wave64 / LDS building blocks for the gfx950 PAF kernels.
 *
 *  - block-wide exclusive scans and min/max reductions over int64 tuples (wave shuffles,
 *    one LDS hop between the 4 waves of a 256-thread workgroup)
 *  - decimal digit counting and packed ASCII conversion (impl/paf.c:10-34 int64_to_str semantics)
 *  - RingWriter: a per-lane byte stream funnelled into aligned dword stores of an LDS ring whose
 *    address is (global output offset mod RING), so that a workgroup's output leaves LDS as full
 *    16-byte coalesced global stores whatever the byte alignment of the rows inside it.
 */
#ifndef PAFFY_DEVICE_UTIL_H_
#define PAFFY_DEVICE_UTIL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PAFFY_NT 256                 /* threads per record workgroup (4 waves of 64) */
#define PAFFY_NWAVE (PAFFY_NT / 64)
#define PAFFY_RING 32768u            /* LDS output ring bytes (power of two) */

/* ---------------- block-wide scans / reductions ---------------- */

/* Exclusive scan of K int64 values per thread; tot[] receives the block totals. scratch: NWAVE*K. */
template <int K>
__device__ __forceinline__ void block_excl_scan(int64_t (&v)[K], int64_t (&tot)[K], int64_t *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t inc[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        int64_t x = v[k];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int64_t t = __shfl_up(x, d, 64);
            if (lane >= d) x += t;
        }
        inc[k] = x;
    }
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < K; k++) scratch[wave * K + k] = inc[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        int64_t base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < PAFFY_NWAVE; w++) {
            int64_t s = scratch[w * K + k];
            if (w < wave) base += s;
            total += s;
        }
        tot[k] = total;
        v[k] = base + inc[k] - v[k];
    }
    __syncthreads();
}

/* Block totals only (every thread receives them). */
template <int K>
__device__ __forceinline__ void block_sum(int64_t (&v)[K], int64_t *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; k++) {
        int64_t x = v[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
        v[k] = x;
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) scratch[wave * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        int64_t total = 0;
#pragma unroll
        for (int w = 0; w < PAFFY_NWAVE; w++) total += scratch[w * K + k];
        v[k] = total;
    }
    __syncthreads();
}

__device__ __forceinline__ int64_t block_min_i64(int64_t x, int64_t *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        int64_t t = __shfl_xor(x, d, 64);
        x = t < x ? t : x;
    }
    if (lane == 0) scratch[wave] = x;
    __syncthreads();
    int64_t r = scratch[0];
#pragma unroll
    for (int w = 1; w < PAFFY_NWAVE; w++) r = scratch[w] < r ? scratch[w] : r;
    __syncthreads();
    return r;
}
__device__ __forceinline__ int64_t block_max_i64(int64_t x, int64_t *scratch) { return -block_min_i64(-x, scratch); }

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
    int bits = 64 - __clzll((long long)(u | 1));
    int t = (bits * 1233) >> 12;
    int d = t + (u >= PAFFY_P10[t] ? 1 : 0);
    if (d < 1) d = 1;
    return d + (v < 0 ? 1 : 0);
}

/* 4 decimal digits of y < 10000 as bytes, most significant digit in byte 0. */
__device__ __forceinline__ uint32_t bcd4(uint32_t y) {
    uint32_t a = y / 100u, b = y - a * 100u;
    uint32_t d0 = a / 10u, d1 = a - d0 * 10u, d2 = b / 10u, d3 = b - d2 * 10u;
    return d0 | (d1 << 8) | (d2 << 16) | (d3 << 24);
}

/* ---------------- byte sinks ---------------- */

/*
 * RingWriter: one lane's contiguous byte stream [s, e) of the workgroup's output window.
 * Bytes are funnelled through a 64-bit accumulator and leave as aligned dword LDS stores;
 * the (at most 3) bytes that share a dword with a neighbouring lane's stream go out as byte
 * stores. put(w, n): append the low n (0..4) bytes of w; bytes of w above n must be zero.
 */
struct RingWriter {
    uint8_t *ring;
    uint64_t acc;
    uint32_t nacc, wpos, head;
    bool first;
    __device__ __forceinline__ void init(uint8_t *r, uint32_t s) {
        ring = r;
        head = s & 3u;
        wpos = s - head;
        nacc = head;
        acc = 0;
        first = head != 0;
    }
    __device__ __forceinline__ void word(uint32_t d) {
        uint32_t a = wpos & (PAFFY_RING - 1);
        if (first) {
            for (uint32_t b = head; b < 4; b++) ring[a + b] = (uint8_t)(d >> (8 * b));
            first = false;
        } else {
            *reinterpret_cast<uint32_t *>(ring + a) = d;
        }
        wpos += 4;
    }
    __device__ __forceinline__ void put(uint32_t w, uint32_t n) {
        acc |= (uint64_t)w << (nacc * 8);
        nacc += n;
        if (nacc >= 4) {
            word((uint32_t)acc);
            acc >>= 32;
            nacc -= 4;
        }
    }
    __device__ __forceinline__ void finish() {
        uint32_t a = wpos & (PAFFY_RING - 1);
        uint32_t lo = first ? head : 0;
        for (uint32_t b = lo; b < nacc; b++) ring[a + b] = (uint8_t)(acc >> (8 * b));
        nacc = 0;
    }
};

/* Plain byte-at-a-time sink: LDS header builder and the sequential fallbacks. */
struct ByteWriter {
    uint8_t *p;
    uint32_t n;
    __device__ __forceinline__ void put(uint32_t w, uint32_t k) {
        for (uint32_t b = 0; b < k; b++) p[n + b] = (uint8_t)(w >> (8 * b));
        n += k;
    }
};

template <class SINK>
__device__ __forceinline__ void put_upto8(SINK &s, uint32_t x) { /* x < 10^8, no leading zeros */
    uint32_t hi4 = x / 10000u, lo4 = x - hi4 * 10000u;
    uint32_t w0 = bcd4(hi4), w1 = bcd4(lo4);
    uint32_t z = w0 ? ((uint32_t)__ffs((int)w0) - 1) >> 3 : (w1 ? 4 + (((uint32_t)__ffs((int)w1) - 1) >> 3) : 7);
    uint64_t c = (((uint64_t)w1 << 32) | w0) + 0x3030303030303030ull;
    c >>= 8 * z;
    uint32_t n = 8 - z;
    s.put((uint32_t)c, n < 4 ? n : 4);
    s.put((uint32_t)(c >> 32), n > 4 ? n - 4 : 0);
}
template <class SINK>
__device__ __forceinline__ void put_exact8(SINK &s, uint32_t x) { /* 8 digits with leading zeros */
    uint32_t hi4 = x / 10000u, lo4 = x - hi4 * 10000u;
    s.put(bcd4(hi4) + 0x30303030u, 4);
    s.put(bcd4(lo4) + 0x30303030u, 4);
}
/* int64_to_str, impl/paf.c:10-34. */
template <class SINK>
__device__ __forceinline__ void put_dec(SINK &s, int64_t v) {
    uint64_t u = (uint64_t)v;
    if (v < 0) {
        s.put('-', 1);
        u = (uint64_t)0 - u;
    }
    if ((u >> 32) == 0) {
        uint32_t x = (uint32_t)u;
        if (x < 100000000u) {
            put_upto8(s, x);
        } else {
            uint32_t q = x / 100000000u;
            put_upto8(s, q);
            put_exact8(s, x - q * 100000000u);
        }
    } else if (u < 10000000000000000ull) {
        uint64_t q = u / 100000000ull;
        put_upto8(s, (uint32_t)q);
        put_exact8(s, (uint32_t)(u - q * 100000000ull));
    } else {
        uint64_t q = u / 10000000000000000ull, r = u - q * 10000000000000000ull;
        uint64_t r1 = r / 100000000ull;
        put_upto8(s, (uint32_t)q);
        put_exact8(s, (uint32_t)r1);
        put_exact8(s, (uint32_t)(r - r1 * 100000000ull));
    }
}
/* Append len bytes of a dword-aligned LDS string. */
template <class SINK>
__device__ __forceinline__ void put_lds(SINK &s, const uint32_t *src, uint32_t len) {
    uint32_t full = len >> 2;
    for (uint32_t k = 0; k < full; k++) s.put(src[k], 4);
    uint32_t rem = len & 3u;
    if (rem) s.put(src[full] & ((1u << (8 * rem)) - 1u), rem);
}

#endif
