/*
 * block_util.h -- collectives of a record workgroup of PAFFY_NT threads (PAFFY_NWAVE waves): DPP wave scans combined with one LDS
 * hop and one barrier. Included once per group size by record_groups.h, inside namespace PAFFY_NS; no include guard on purpose.
 */
namespace PAFFY_NS {

/* ---------------- block-wide scans / reductions ---------------- */

/*
 * Scratch protocol: every collective writes one slot set of `scratch` and ends with a single
 * barrier; consecutive collectives alternate between two slot sets (toggle kept per thread, all
 * threads call the same sequence), so the slots of collective i are not rewritten before the
 * barrier of collective i+1 has been passed by every reader of i. scratch: 2 * NWAVE * 4 words.
 */
struct BlockComm {
    int64_t *scratch;
    uint32_t flip;
    __device__ __forceinline__ int64_t *slots() {
        int64_t *p = scratch + flip * (PAFFY_NWAVE * 4);
        flip ^= 1u;
        return p;
    }
};

/* Exclusive scan of K (<= 4) int64 values per thread; tot[] receives the block totals. */
template <int K>
__device__ __forceinline__ void block_excl_scan(int64_t (&v)[K], int64_t (&tot)[K], BlockComm &bc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t inc[K];
    int64_t *sl = bc.slots();
#pragma unroll
    for (int k = 0; k < K; k++) {
        inc[k] = wave_incl_scan(v[k]);
        if (lane == 63) sl[wave * K + k] = inc[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        int64_t base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < PAFFY_NWAVE; w++) {
            int64_t s = sl[w * K + k];
            if (w < wave) base += s;
            total += s;
        }
        tot[k] = total;
        v[k] = base + inc[k] - v[k];
    }
}

/* Block totals only (every thread receives them). */
template <int K>
__device__ __forceinline__ void block_sum(int64_t (&v)[K], BlockComm &bc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t *sl = bc.slots();
#pragma unroll
    for (int k = 0; k < K; k++) {
        int64_t t = wave_last(wave_incl_scan(v[k]));
        if (lane == 0) sl[wave * K + k] = t;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        int64_t total = 0;
#pragma unroll
        for (int w = 0; w < PAFFY_NWAVE; w++) total += sl[w * K + k];
        v[k] = total;
    }
}

__device__ __forceinline__ int64_t block_min_i64(int64_t x, BlockComm &bc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t *sl = bc.slots();
    int64_t m = wave_min(x);
    if (lane == 0) sl[wave] = m;
    __syncthreads();
    int64_t r = sl[0];
#pragma unroll
    for (int w = 1; w < PAFFY_NWAVE; w++) r = sl[w] < r ? sl[w] : r;
    return r;
}
__device__ __forceinline__ int64_t block_max_i64(int64_t x, BlockComm &bc) { return -block_min_i64(-x, bc); }

/* 32-bit collectives for records whose sums fit 31 bits: a quarter of the DPP instructions of the 64-bit ones */
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    uint32_t x = v;
#define PAFFY_MIN32_STEP(CTRL, RM, BM, SRC)                                                              \
    {                                                                                                    \
        uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)(SRC), CTRL, RM, BM, false);      \
        x = t < x ? t : x;                                                                               \
    }
    PAFFY_MIN32_STEP(DPP_ROW_SHR(1), 0xf, 0xf, v)
    PAFFY_MIN32_STEP(DPP_ROW_SHR(2), 0xf, 0xf, v)
    PAFFY_MIN32_STEP(DPP_ROW_SHR(3), 0xf, 0xf, v)
    PAFFY_MIN32_STEP(DPP_ROW_SHR(4), 0xf, 0xe, x)
    PAFFY_MIN32_STEP(DPP_ROW_SHR(8), 0xf, 0xc, x)
    PAFFY_MIN32_STEP(DPP_BCAST15, 0xa, 0xf, x)
    PAFFY_MIN32_STEP(DPP_BCAST31, 0xc, 0xf, x)
#undef PAFFY_MIN32_STEP
    return wave_last_u32(x);
}
__device__ __forceinline__ uint32_t block_min_u32(uint32_t x, BlockComm &bc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *sl = reinterpret_cast<uint32_t *>(bc.slots());
    const uint32_t m = wave_min_u32(x);
    if (lane == 0) sl[wave] = m;
    __syncthreads();
    uint32_t r = sl[0];
#pragma unroll
    for (int w = 1; w < PAFFY_NWAVE; w++) r = sl[w] < r ? sl[w] : r;
    return r;
}
/* max of values >= -1 (indices, -1 = none) */
__device__ __forceinline__ int32_t block_max_idx(int32_t x, BlockComm &bc) { return (int32_t)(0xfffffffeu - block_min_u32(0xfffffffeu - (uint32_t)(x + 1), bc)) - 1; }
template <int K>
__device__ __forceinline__ void block_excl_scan_u32(uint32_t (&v)[K], uint32_t (&tot)[K], BlockComm &bc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc[K];
    uint32_t *sl = reinterpret_cast<uint32_t *>(bc.slots());
#pragma unroll
    for (int k = 0; k < K; k++) {
        inc[k] = wave_incl_scan_u32(v[k]);
        if (lane == 63) sl[wave * K + k] = inc[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        uint32_t base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < PAFFY_NWAVE; w++) {
            const uint32_t s = sl[w * K + k];
            if (w < wave) base += s;
            total += s;
        }
        tot[k] = total;
        v[k] = base + inc[k] - v[k];
    }
}
/* exclusive scan of four sums and the minimum of a fifth value with a single barrier */
__device__ __forceinline__ void block_excl_scan4_min_u32(uint32_t (&v)[4], uint32_t (&tot)[4], uint32_t &mn, BlockComm &bc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc[4];
    uint32_t *sl = reinterpret_cast<uint32_t *>(bc.slots());
#pragma unroll
    for (int k = 0; k < 4; k++) {
        inc[k] = wave_incl_scan_u32(v[k]);
        if (lane == 63) sl[wave * 5 + k] = inc[k];
    }
    const uint32_t wm = wave_min_u32(mn);
    if (lane == 0) sl[wave * 5 + 4] = wm;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < PAFFY_NWAVE; w++) {
            const uint32_t s = sl[w * 5 + k];
            if (w < wave) base += s;
            total += s;
        }
        tot[k] = total;
        v[k] = base + inc[k] - v[k];
    }
    uint32_t r = sl[4];
#pragma unroll
    for (int w = 1; w < PAFFY_NWAVE; w++) r = sl[w * 5 + 4] < r ? sl[w * 5 + 4] : r;
    mn = r;
}
template <int K>
__device__ __forceinline__ void block_sum_u32(uint32_t (&v)[K], BlockComm &bc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *sl = reinterpret_cast<uint32_t *>(bc.slots());
#pragma unroll
    for (int k = 0; k < K; k++) {
        const uint32_t t = wave_last_u32(wave_incl_scan_u32(v[k]));
        if (lane == 0) sl[wave * K + k] = t;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < PAFFY_NWAVE; w++) total += sl[w * K + k];
        v[k] = total;
    }
}


} /* namespace */
