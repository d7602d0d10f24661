// Probe: where should the row writer place its byte-aligned 16-byte pieces?  One wave per record of 1 062 rows of 124..130 bytes,
// nine pieces per row, 131 072 records (16.7 GB) -- the skeleton of k_emit_rows without its arithmetic.
//   V0  pieces into the wave's linear LDS window with byte-aligned ds_write_b128, window flushed as coalesced 16-byte global stores
//   V1  pieces straight to HBM: byte-aligned global_store_dwordx4, one row per lane
//   V2  straight to HBM, eight lanes per row (lane = row * 8 + piece): neighbouring lanes write neighbouring bytes
//   V3  pieces into the LDS window with ds_write_b8 (sixteen per piece), flushed like V0
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/rowstore_global rowstore_global.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define ROWS 1062u
#define WIN_LDS 9728u

__device__ __forceinline__ uint32_t row_len(uint32_t rec, uint32_t row) {
    uint32_t h = (rec * 2654435761u) ^ (row * 40503u);
    h ^= h >> 13;
    h *= 0x5bd1e995u;
    h ^= h >> 15;
    return 124u + h % 7u;
}

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t lane, uint32_t &total) {
    uint32_t s = v;
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(s, d, 64);
        if ((int)lane >= d) s += t;
    }
    total = __shfl(s, 63, 64);
    return s - v;
}

template <int V, int FA, int NT>
__global__ __launch_bounds__(256, 4) void k(uint8_t *out, const uint64_t *rec_off, uint32_t n_rec) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4][WIN_LDS];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t rec = blockIdx.x * 4 + wave;
    if (rec >= n_rec) return;
    uint8_t *dst = out + rec_off[rec];
    uint8_t *win = lds[wave];
    const uint32_t win_a = (uint32_t)(uintptr_t)win;
    uint32_t carry = (uint32_t)((uintptr_t)dst & (uintptr_t)(FA - 1));  // bytes in front of the first row inside its 16-byte chunk
    uint8_t *flush_to = dst - carry;
    uint64_t done = 0;
    for (uint32_t r0 = 0; r0 < ROWS; r0 += 64) {
        const uint32_t row = r0 + lane;
        const uint32_t len = row < ROWS ? (V == 7 ? 127u : row_len(rec, row)) : 0u;
        uint32_t total;
        const uint32_t off = V == 7 ? lane * 127u : wave_excl_scan(len, lane, total);
        if (V == 7) total = (ROWS - r0 < 64u ? ROWS - r0 : 64u) * 127u;
        u32x4 v = {row * 0x01010101u, len * 0x01010101u, rec, lane * 0x03030303u};
        if (V == 0 || V == 3 || V == 4 || V == 5 || V == 6 || V == 7) {
            if (len && V != 4 && V != 6 && V != 7) {
                const uint32_t a = win_a + carry + off;
#pragma unroll
                for (int j = 8; j >= 0; j--) {
                    const uint32_t at = j == 8 ? a + len - 16u : a + j * 14u;
                    v.x += j;
                    if (V == 0 || V == 5) {
                        asm volatile("ds_write_b128 %0, %1" ::"v"(at), "v"(v) : "memory");
                    } else {
#pragma unroll
                        for (int w = 0; w < 4; w++) {
                            const uint32_t x = w == 0 ? v.x : w == 1 ? v.y : w == 2 ? v.z : v.w;
                            const uint32_t y = x >> 8;
                            asm volatile("ds_write_b8 %0, %1 offset:%2\n\tds_write_b8_d16_hi %0, %1 offset:%3\n\t"
                                         "ds_write_b8 %0, %4 offset:%5\n\tds_write_b8_d16_hi %0, %4 offset:%6" ::"v"(at),
                                         "v"(x), "n"(w * 4), "n"(w * 4 + 2), "v"(y), "n"(w * 4 + 1), "n"(w * 4 + 3)
                                         : "memory");
                        }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint32_t have = carry + total;
            const uint32_t chunks = (have / FA) * (FA / 16);
            if (V != 5) {
                for (uint32_t c = lane; c < chunks; c += 64) {
                    u32x4 t = v;
                    if (V != 6 && V != 7) t = *reinterpret_cast<const u32x4 *>(win + c * 16);
                    uint8_t *to = flush_to + c * 16;
                    if (NT == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(to), "v"(t) : "memory");
                    else if (NT == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(to), "v"(t) : "memory");
                    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(to), "v"(t) : "memory");
                }
            }
            const uint32_t rest = have - chunks * 16;
            u32x4 tail = {0, 0, 0, 0};
            if (V != 6 && V != 7) {
                if (lane * 16 < rest) tail = *reinterpret_cast<const u32x4 *>(win + chunks * 16 + lane * 16);
                __builtin_amdgcn_wave_barrier();
                if (lane * 16 < rest) *reinterpret_cast<u32x4 *>(win + lane * 16) = tail;
            }
            flush_to += chunks * 16;
            carry = rest;
        } else if (V == 1) {
            if (len) {
                uint8_t *a = dst + done + off;
#pragma unroll
                for (int j = 8; j >= 0; j--) {
                    uint8_t *at = j == 8 ? a + len - 16u : a + j * 14u;
                    v.x += j;
                    asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(at), "v"(v) : "memory");
                }
            }
        } else {
            // eight lanes per row: pass p handles rows p*8 .. p*8+7 of the window; the ninth piece (end-anchored) by lane 7 again
            const uint32_t sub = lane & 7u;
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const uint32_t src = p * 8 + (lane >> 3);
                const uint32_t o = __shfl(off, src, 64), l = __shfl(len, src, 64);
                if (l) {
                    uint8_t *a = dst + done + o;
                    v.x += p;
                    uint8_t *at = a + sub * 16u;
                    if (sub * 16u + 16u > l) at = a + l - 16u;
                    asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(at), "v"(v) : "memory");
                    if (l > 128u && sub == 7u) {
                        at = a + l - 16u;
                        asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(at), "v"(v) : "memory");
                    }
                }
            }
        }
        done += total;
    }
    if (V == 0 || V == 3 || V == 4) {
        if (lane == 0 && carry) {
            for (uint32_t i = 0; i < carry; i++) flush_to[i] = win[i];
        }
    }
}

__global__ void k_sizes(uint64_t *rec_len, uint32_t n_rec) {
    const uint32_t rec = blockIdx.x * blockDim.x + threadIdx.x;
    if (rec >= n_rec) return;
    uint64_t s = 0;
    for (uint32_t r = 0; r < ROWS; r++) s += row_len(rec, r);
    rec_len[rec] = s;
}

template <int V, int FA = 16, int NT = 0>
static void run(uint8_t *d, const uint64_t *off, uint32_t n, double bytes, const char *name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        k<V, FA, NT><<<(n + 3) / 4, 256>>>(d, off, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipError_t e = hipGetLastError();
    printf("%-66s %.3f ms  %.0f GB/s  (%s)\n", name, best, bytes / best / 1e6, hipGetErrorString(e));
}

// Several waves per record: wave w of a group of WPR takes the windows w, w + WPR, ... of the record (window = 64 rows), each at the
// byte offset a sizing pass would have left per window.  PIECES: the nine byte-aligned LDS stores per row are made or not.
// Each window is written on its own: whole 16-byte chunks from LDS, the ragged edges (< 16 bytes at either end) with byte stores.
#define NWIN ((ROWS + 63u) / 64u)
template <int WPR, int PIECES>
__global__ __launch_bounds__(256, 4) void k2(uint8_t *out, const uint64_t *rec_off, const uint32_t *win_off, uint32_t n_rec) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4][WIN_LDS];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t gw = blockIdx.x * 4 + wave;   // global wave
    const uint32_t rec = gw / WPR, wv = gw % WPR;
    if (rec >= n_rec) return;
    uint8_t *win = lds[wave];
    const uint32_t win_a = (uint32_t)(uintptr_t)win;
    for (uint32_t wi = wv; wi < NWIN; wi += WPR) {
        uint8_t *dst = out + rec_off[rec] + win_off[rec * (NWIN + 1) + wi];
        const uint32_t phase = (uint32_t)((uintptr_t)dst & 15u);
        const uint32_t row = wi * 64 + lane;
        const uint32_t len = row < ROWS ? row_len(rec, row) : 0u;
        uint32_t total;
        const uint32_t off = wave_excl_scan(len, lane, total);
        u32x4 v = {row * 0x01010101u, len * 0x01010101u, rec, lane * 0x03030303u};
        if (PIECES && len) {
            const uint32_t a = win_a + phase + off;
#pragma unroll
            for (int j = 8; j >= 0; j--) {
                const uint32_t at = j == 8 ? a + len - 16u : a + j * 14u;
                v.x += j;
                asm volatile("ds_write_b128 %0, %1" ::"v"(at), "v"(v) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        uint8_t *base = dst - phase;   // 16-byte aligned
        const uint32_t have = phase + total;
        const uint32_t c_lo = phase ? 1u : 0u, c_hi = have >> 4;
        if (phase && lane >= phase && lane < 16) base[lane] = win[lane];
        for (uint32_t c = c_lo + lane; c < c_hi; c += 64) {
            u32x4 t = *reinterpret_cast<const u32x4 *>(win + c * 16);
            uint8_t *to = base + c * 16;
            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(to), "v"(t) : "memory");
        }
        if (lane < (have & 15u)) base[c_hi * 16 + lane] = win[c_hi * 16 + lane];
        __builtin_amdgcn_wave_barrier();
    }
}
__global__ void k_win(uint32_t *win_off, uint32_t n_rec) {
    const uint32_t rec = blockIdx.x * blockDim.x + threadIdx.x;
    if (rec >= n_rec) return;
    uint32_t s = 0;
    for (uint32_t r = 0; r < ROWS; r++) {
        if ((r & 63u) == 0) win_off[rec * (NWIN + 1) + (r >> 6)] = s;
        s += row_len(rec, r);
    }
    win_off[rec * (NWIN + 1) + NWIN] = s;
}
template <int WPR, int PIECES>
static void run2(uint8_t *d, const uint64_t *off, const uint32_t *woff, uint32_t n, double bytes, const char *name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        k2<WPR, PIECES><<<(uint32_t)(((uint64_t)n * WPR + 3) / 4), 256>>>(d, off, woff, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipError_t e = hipGetLastError();
    printf("%-66s %.3f ms  %.0f GB/s  (%s)\n", name, best, bytes / best / 1e6, hipGetErrorString(e));
}

int main() {
    const uint32_t n = 131072;
    uint64_t *len, *off;
    hipMalloc(&len, 8 * n);
    hipMalloc(&off, 8 * n);
    k_sizes<<<n / 256, 256>>>(len, n);
    uint64_t *h = (uint64_t *)malloc(8 * n);
    hipMemcpy(h, len, 8 * n, hipMemcpyDeviceToHost);
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; i++) { uint64_t l = h[i]; h[i] = total; total += l; }
    hipMemcpy(off, h, 8 * n, hipMemcpyHostToDevice);
    uint8_t *d;
    if (hipMalloc(&d, total + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, total + 4096);
    printf("%u records, %.2f GB of rows\n", n, total / 1e9);
    uint32_t *woff;
    hipMalloc(&woff, 4ull * n * (NWIN + 1));
    k_win<<<n / 256, 256>>>(woff, n);
    run<0>(d, off, n, (double)total, "V0 LDS window, byte-aligned ds_write_b128, flush in 16 B granules");
    run<4, 128>(d, off, n, (double)total, "V4 flush only, 128 B granules");
    run<6, 128>(d, off, n, (double)total, "V6 flush only from registers (no LDS read), 128 B granules");
    run<6, 16>(d, off, n, (double)total, "V6 flush only from registers, 16 B granules");
    run<7, 128>(d, off, n, (double)total, "V7 like V6, rows of 127 bytes (no hash, no scan), 128 B granules");
    run<5>(d, off, n, (double)total, "V5 piece stores only (no global stores)");
    return 0;
}
