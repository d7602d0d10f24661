// Probe: do byte-aligned (serialised) LDS stores of one wave overlap with VALU work of the other waves of the CU?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int NST, int NVALU64>
__global__ __launch_bounds__(256, 4) void k(uint32_t *out, int iters, uint32_t stride, int aligned) {
    extern __shared__ uint8_t buf[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t a = threadIdx.x, b = a * 3 + 1, c = b ^ 0x55;
    const uint32_t base = (uint32_t)(uintptr_t)(buf + wave * 9216) + (aligned ? lane * 16 : lane * stride + 3);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int s = 0; s < NST; s++) {
            const v4u q = {a, b, c, (unsigned)i};
            asm volatile("ds_write_b128 %0, %1" ::"v"(base + (aligned ? s * 1024 : s * 16)), "v"(q) : "memory");
        }
#pragma unroll
        for (int r = 0; r < NVALU64; r++) { REP64(asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %2, %2, %0" : "+v"(a), "+v"(b), "+v"(c));) }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c;
}

template <int NST, int NV>
static float run(uint32_t *o, int aligned) {
    hipFuncSetAttribute((const void *)k<NST, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 39000);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        k<NST, NV><<<256 * 4, 256, 39000>>>(o, 1000, 120, aligned);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

int main() {
    uint32_t *o;
    hipMalloc(&o, 4 * 256 * 1024 * 4);
    // per iteration and wave: NST stores, NV*128 VALU instructions; 16 waves per CU, 4 per SIMD
    printf("stores only (10 unaligned)      : %.3f ms\n", run<10, 0>(o, 0));
    printf("stores only (10 aligned)        : %.3f ms\n", run<10, 0>(o, 1));
    printf("VALU only (640 instr)           : %.3f ms\n", run<0, 5>(o, 0));
    printf("10 unaligned stores + 640 VALU  : %.3f ms\n", run<10, 5>(o, 0));
    printf("10 aligned stores + 640 VALU    : %.3f ms\n", run<10, 5>(o, 1));
    printf("VALU only (1280 instr)          : %.3f ms\n", run<0, 10>(o, 0));
    printf("10 unaligned stores + 1280 VALU : %.3f ms\n", run<10, 10>(o, 0));
    printf("5 unaligned stores + 640 VALU   : %.3f ms\n", run<5, 5>(o, 0));
    return 0;
}
