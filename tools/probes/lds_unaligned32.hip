// Probe: byte-aligned ds_write_b32 / ds_read_b32 / ds_write_b16 on gfx950: are they correct, and what do they cost next to the
// 64-cycle replay of a misaligned ds_write_b64 / ds_write_b128 (lds_unaligned.hip)?
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/lds_unaligned32 lds_unaligned32.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

__global__ void k_check(uint8_t *out, uint32_t stride, uint32_t phase) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[16384];
    for (uint32_t i = threadIdx.x; i < 16384 / 4; i += blockDim.x) reinterpret_cast<uint32_t *>(buf)[i] = 0;
    __syncthreads();
    const uint32_t a = (uint32_t)(uintptr_t)(buf + phase + threadIdx.x * stride);
    const uint32_t v = 0x04030201u + threadIdx.x * 0x01010101u;
    asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
    asm volatile("ds_write_b16 %0, %1" ::"v"(a + 8192u), "v"(v) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
    for (uint32_t i = threadIdx.x; i < 16384 - 256; i += blockDim.x) out[i] = buf[i];
    reinterpret_cast<uint32_t *>(out + 16384 - 256)[threadIdx.x] = r;
}

template <int MODE>
__global__ void k_time(uint32_t *out, uint32_t stride, uint32_t phase, int iters) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[65536 / 2];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t *mine = buf + wave * 8192;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const uint32_t o = phase + lane * stride + (it & 3) * 16;
        const uint32_t a = (uint32_t)(uintptr_t)(mine + o);
        const uint32_t v = (uint32_t)it * 77u + lane;
        if (MODE == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
        else if (MODE == 1) asm volatile("ds_write_b16 %0, %1" ::"v"(a), "v"(v) : "memory");
        else if (MODE == 2) asm volatile("ds_write_b8 %0, %1" ::"v"(a), "v"(v) : "memory");
        else {
            uint32_t r;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
            acc += r;
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
    for (uint32_t stride : {16u, 27u, 125u}) {
        for (uint32_t phase = 0; phase < 8; phase++) {
            k_check<<<1, 64>>>(d, stride, phase);
            hipMemcpy(h.data(), d, 16384, hipMemcpyDeviceToHost);
            std::vector<uint8_t> want(16384, 0);
            for (uint32_t t = 0; t < 64; t++) {
                const uint32_t v = 0x04030201u + t * 0x01010101u;
                memcpy(&want[phase + t * stride], &v, 4);
                memcpy(&want[8192 + phase + t * stride], &v, 2);
            }
            if (memcmp(want.data(), h.data(), 16384 - 256)) { bad++; printf("WRITE MISMATCH stride %u phase %u\n", stride, phase); }
            for (uint32_t t = 0; t < 64; t++) {
                uint32_t r, v = 0x04030201u + t * 0x01010101u;
                memcpy(&r, &h[16384 - 256 + 4 * t], 4);
                if (r != v) { bad++; printf("READ MISMATCH stride %u phase %u lane %u: %08x != %08x\n", stride, phase, t, r, v); break; }
            }
        }
    }
    printf("byte-aligned ds_write_b32 / ds_write_b16 / ds_read_b32 check: %s\n", bad ? "FAILED" : "ok");
    uint32_t *o;
    hipMalloc(&o, 4 * 256 * 2048);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4096;
    for (uint32_t stride : {16u, 124u, 125u}) {
        for (uint32_t phase : {0u, 1u, 2u, 3u}) {
            float ms[4];
            for (int mode = 0; mode < 4; mode++) {
                hipEventRecord(e0);
                if (mode == 0) k_time<0><<<2048, 256>>>(o, stride, phase, iters);
                if (mode == 1) k_time<1><<<2048, 256>>>(o, stride, phase, iters);
                if (mode == 2) k_time<2><<<2048, 256>>>(o, stride, phase, iters);
                if (mode == 3) k_time<3><<<2048, 256>>>(o, stride, phase, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[mode], e0, e1);
            }
            /* 2048 workgroups x 4 waves x iters instructions over 256 CUs: cycles per wave-instruction and CU at 2.4 GHz */
            const double k = 2.4e6 / (2048.0 * 4 * iters / 256.0);
            printf("stride %3u phase %u: write_b32 %.1f  write_b16 %.1f  write_b8 %.1f  read_b32 %.1f cycles per instruction and CU\n", stride, phase, ms[0] * k, ms[1] * k,
                   ms[2] * k, ms[3] * k);
        }
    }
    return 0;
}
