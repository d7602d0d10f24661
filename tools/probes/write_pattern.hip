// Probe: HBM write rate of the emit kernel's store pattern (many sequential streams written in 8 KB bursts)
// against a plain fill, with the same launch shape (65536 workgroups of 256 threads, 128 KB each).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// MODE 0: each wave owns a contiguous quarter (32 KB) and writes it in 8 KB bursts (8 x 1 KB stores)
// MODE 1: the 4 waves of the workgroup interleave 1 KB stores over the 128 KB (workgroup-contiguous)
// MODE 2: like 0 with ~400 VALU instructions between bursts
template <int MODE>
__global__ __launch_bounds__(256, 4) void k(uint8_t *out, uint32_t bytes_per_wg) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t *base = out + (size_t)blockIdx.x * bytes_per_wg;
    uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
    uint32_t a = lane, b = wave + 3, c = 5;
    if (MODE == 1) {
        for (uint32_t off = wave * 1024; off < bytes_per_wg; off += 4096) *reinterpret_cast<uint4 *>(base + off + lane * 16) = v;
    } else {
        const uint32_t share = bytes_per_wg / 4;
        uint8_t *p = base + wave * share;
        for (uint32_t off = 0; off < share; off += 8192) {
#pragma unroll
            for (int j = 0; j < 8; j++) *reinterpret_cast<uint4 *>(p + off + j * 1024 + lane * 16) = v;
            if (MODE == 2) {
                for (int r = 0; r < 3; r++) { REP64(asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %2, %2, %0" : "+v"(a), "+v"(b), "+v"(c));) }
                v.w = a ^ c;
            }
        }
    }
}

template <int MODE>
static void run(uint8_t *d, const char *name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const uint32_t wgs = 65536, per = 131072;
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k<MODE><<<wgs, 256>>>(d, per);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-58s %.3f ms  %.0f GB/s\n", name, ms, (double)wgs * per / ms / 1e6);
}

int main() {
    uint8_t *d;
    hipMalloc(&d, (size_t)65536 * 131072);
    hipMemset(d, 0, (size_t)65536 * 131072);
    run<1>(d, "workgroup-contiguous, waves interleave 1 KB stores");
    run<0>(d, "wave-contiguous quarters, 8 KB bursts");
    run<2>(d, "wave-contiguous quarters, 8 KB bursts + 400 VALU");
    return 0;
}
