// Probe: what does it cost to place a 16-byte piece at a byte address of the LDS on gfx950?
//   mode 0  one byte-aligned ds_write_b128 (what the row writer does today)
//   mode 1  ds_or_b32 on the first dword + ds_write_b128 on the four dwords behind it (dword-aligned, not 16-byte-aligned)
//   mode 2  ds_write_b128 at a dword-aligned address alone
//   mode 3  ds_write_b64 x2 at dword-aligned addresses
//   mode 4  ds_write_b32 x4 at dword-aligned addresses
//   mode 5  ds_write_b128 at a 16-byte-aligned address (floor)
//   mode 6  ds_write_b96 + ds_write_b32, dword-aligned
//   mode 7  ds_write_b64 at an 8-byte aligned address x2
// Rows of 125..130 bytes back to back, lane = row, like the shatter writer: nine pieces per row.
// build: hipcc -O3 --offload-arch=gfx950 -o lds_rowstore lds_rowstore.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v3u __attribute__((ext_vector_type(3)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k_time(uint32_t *out, int iters, uint32_t jitter) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[4 * 9216];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t *mine = buf + wave * 9216;
    // row offsets: 125 + (hash % jitter) bytes per row, prefix sum over the lanes
    uint32_t len = 125u + ((lane * 2654435761u) >> 28) % (jitter ? jitter : 1u);
    uint32_t o = len;
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(o, d);
        if (lane >= (uint32_t)d) o += t;
    }
    o -= len;
    uint32_t acc = 0;
    const uint32_t base = (uint32_t)(uintptr_t)mine;
    for (int it = 0; it < iters; it++) {
        const v4u q = {(unsigned)it, lane, (unsigned)it + 1, lane + 1};
#pragma unroll
        for (int p = 0; p < 9; p++) {
            uint32_t a = base + o + 13u * p + (it & 3);
            if (MODE == 0) {
                asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(q) : "memory");
            } else if (MODE == 1) {
                const uint32_t d = a & ~3u;
                asm volatile("ds_or_b32 %0, %1" ::"v"(d), "v"(q.x) : "memory");
                asm volatile("ds_write_b128 %0, %1 offset:4" ::"v"(d), "v"(q) : "memory");
            } else if (MODE == 2) {
                const uint32_t d = a & ~3u;
                asm volatile("ds_write_b128 %0, %1" ::"v"(d), "v"(q) : "memory");
            } else if (MODE == 3) {
                const uint32_t d = a & ~3u;
                const v2u h0 = {q.x, q.y}, h1 = {q.z, q.w};
                asm volatile("ds_write_b64 %0, %1" ::"v"(d), "v"(h0) : "memory");
                asm volatile("ds_write_b64 %0, %1 offset:8" ::"v"(d), "v"(h1) : "memory");
            } else if (MODE == 4) {
                const uint32_t d = a & ~3u;
                asm volatile("ds_write_b32 %0, %1" ::"v"(d), "v"(q.x) : "memory");
                asm volatile("ds_write_b32 %0, %1 offset:4" ::"v"(d), "v"(q.y) : "memory");
                asm volatile("ds_write_b32 %0, %1 offset:8" ::"v"(d), "v"(q.z) : "memory");
                asm volatile("ds_write_b32 %0, %1 offset:12" ::"v"(d), "v"(q.w) : "memory");
            } else if (MODE == 5) {
                const uint32_t d = a & ~15u;
                asm volatile("ds_write_b128 %0, %1" ::"v"(d), "v"(q) : "memory");
            } else if (MODE == 6) {
                const uint32_t d = a & ~3u;
                const v3u t = {q.x, q.y, q.z};
                asm volatile("ds_write_b96 %0, %1" ::"v"(d), "v"(t) : "memory");
                asm volatile("ds_write_b32 %0, %1 offset:12" ::"v"(d), "v"(q.w) : "memory");
            } else {
                const uint32_t d = a & ~7u;
                const v2u h0 = {q.x, q.y}, h1 = {q.z, q.w};
                asm volatile("ds_write_b64 %0, %1" ::"v"(d), "v"(h0) : "memory");
                asm volatile("ds_write_b64 %0, %1 offset:8" ::"v"(d), "v"(h1) : "memory");
            }
        }
        if ((it & 15) == 15) acc += *reinterpret_cast<uint32_t *>(mine + lane * 4);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    uint32_t *o;
    hipMalloc(&o, 4 * 256 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2048;
    const char *names[8] = {"b128 byte-aligned", "or_b32 + b128 dword-aligned", "b128 dword-aligned", "2 x b64 dword-aligned", "4 x b32", "b128 16-aligned",
                            "b96 + b32 dword-aligned", "2 x b64 8-aligned"};
    for (uint32_t jitter : {1u, 6u}) {
        for (int mode = 0; mode < 8; mode++) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                switch (mode) {
                case 0: k_time<0><<<1024, 256>>>(o, iters, jitter); break;
                case 1: k_time<1><<<1024, 256>>>(o, iters, jitter); break;
                case 2: k_time<2><<<1024, 256>>>(o, iters, jitter); break;
                case 3: k_time<3><<<1024, 256>>>(o, iters, jitter); break;
                case 4: k_time<4><<<1024, 256>>>(o, iters, jitter); break;
                case 5: k_time<5><<<1024, 256>>>(o, iters, jitter); break;
                case 6: k_time<6><<<1024, 256>>>(o, iters, jitter); break;
                default: k_time<7><<<1024, 256>>>(o, iters, jitter); break;
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            // pieces placed per second per CU: 1024 blocks x 4 waves x iters x 9 pieces over 256 CUs
            const double pieces = 1024.0 * 4 * iters * 9 / 256.0;
            printf("jitter %u  %-30s %.3f ms  %.1f ns per piece and CU (%.1f cycles at 2.4 GHz)\n", jitter, names[mode], ms, ms * 1e6 / pieces, ms * 1e-3 * 2.4e9 / pieces);
        }
    }
    return 0;
}
