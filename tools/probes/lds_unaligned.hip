// Probe: are byte-aligned ds_write_b64 / ds_write_b128 correct on gfx950 and what do they cost?
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/lds_unaligned lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
typedef uint64_t __attribute__((aligned(1))) u64u;
typedef uint4 __attribute__((aligned(1))) u128u;

__global__ void k_check(uint8_t *out, uint32_t stride, uint32_t phase) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[16384];
    for (uint32_t i = threadIdx.x; i < 16384 / 4; i += blockDim.x) reinterpret_cast<uint32_t *>(buf)[i] = 0;
    __syncthreads();
    const uint32_t o = phase + threadIdx.x * stride;
    uint64_t v = 0;
    for (int b = 0; b < 8; b++) v |= (uint64_t)((threadIdx.x * 8 + b) & 0xff) << (8 * b);
    *reinterpret_cast<u64u *>(buf + o) = v;
    uint4 w = make_uint4(threadIdx.x * 4 + 0x1000000u, threadIdx.x * 4 + 0x2000001u, threadIdx.x * 4 + 0x3000002u, threadIdx.x * 4 + 0x4000003u);
    *reinterpret_cast<u128u *>(buf + 8192 + o) = w;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 16384; i += blockDim.x) out[i] = buf[i];
}

template <int MODE>
__global__ void k_time(uint32_t *out, uint32_t stride, uint32_t phase, int iters) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[65536 / 2];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t *mine = buf + wave * 8192;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const uint32_t o = phase + lane * stride + (it & 3) * 16;
        const uint32_t a = (uint32_t)(uintptr_t)(mine + o); /* LDS byte address */
        const uint64_t v = (uint64_t)it * 77 + lane;
        if (MODE == 0) {
            asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory");
        } else if (MODE == 1) {
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            const v4u q = {(unsigned)it, lane, (unsigned)it + 1, lane + 1};
            asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(q) : "memory");
        } else {
            asm volatile("ds_write_b32 %0, %1" ::"v"(a & ~3u), "v"((uint32_t)v) : "memory");
        }
        if ((it & 63) == 63) acc += *reinterpret_cast<uint32_t *>(mine + lane * 4);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    uint8_t *d;
    hipMalloc(&d, 16384);
    std::vector<uint8_t> h(16384);
    int bad = 0;
    for (uint32_t stride : {8u, 16u, 27u, 125u}) {
        for (uint32_t phase = 0; phase < 16; phase++) {
            if (phase + 63 * stride + 16 > 8192) continue;
            k_check<<<1, 64>>>(d, stride, phase);
            hipMemcpy(h.data(), d, 16384, hipMemcpyDeviceToHost);
            std::vector<uint8_t> want(16384, 0);
            if (stride >= 16) {
                for (uint32_t t = 0; t < 64; t++) {
                    for (int b = 0; b < 8; b++) want[phase + t * stride + b] = (t * 8 + b) & 0xff;
                    uint32_t w[4] = {t * 4 + 0x1000000u, t * 4 + 0x2000001u, t * 4 + 0x3000002u, t * 4 + 0x4000003u};
                    memcpy(&want[8192 + phase + t * stride], w, 16);
                }
                if (memcmp(want.data(), h.data(), 16384)) { bad++; printf("MISMATCH stride %u phase %u\n", stride, phase); }
            }
        }
    }
    printf("unaligned ds_write check: %s\n", bad ? "FAILED" : "ok");
    uint32_t *o;
    hipMalloc(&o, 4 * 256 * 2048);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4096;
    for (uint32_t stride : {16u, 20u, 124u, 125u, 132u}) {
        for (uint32_t phase : {0u, 4u, 8u, 2u}) {
            float ms[3];
            for (int mode = 0; mode < 3; mode++) {
                for (int rep = 0; rep < 2; rep++) {
                    hipEventRecord(e0);
                    if (mode == 0) k_time<0><<<2048, 256>>>(o, stride, phase, iters);
                    if (mode == 1) k_time<1><<<2048, 256>>>(o, stride, phase, iters);
                    if (mode == 2) k_time<2><<<2048, 256>>>(o, stride, phase, iters);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms[mode], e0, e1);
                }
            }
            // wave-instructions per second per CU
            double n = 2048.0 * 4 * iters;
            printf("stride %3u phase %u: b64 %.3f ms (%.1f cyc/instr/CU)  b128 %.3f ms (%.1f)  b32 aligned %.3f ms (%.1f)\n", stride, phase, ms[0],
                   ms[0] * 1e-3 * 2.4e9 * 256 / n, ms[1], ms[1] * 1e-3 * 2.4e9 * 256 / n, ms[2], ms[2] * 1e-3 * 2.4e9 * 256 / n);
        }
    }
    return bad;
}
