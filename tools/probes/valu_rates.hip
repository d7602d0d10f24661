// Probe: issue cost of the integer VALU instructions the text kernels lean on, relative to v_add_u32.
// Every workgroup runs 256 threads of an unrolled block of 64 independent-ish instructions, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = b ^ 0x55, d = c + 7;
    uint64_t p = ((uint64_t)a << 32) | b, q = ((uint64_t)c << 32) | d;
    uint32_t sh = (threadIdx.x & 7) + 1;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { REP64(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1" : "+v"(a), "+v"(b), "+v"(c));) }
        if (MODE == 1) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %1" : "+v"(a), "+v"(b), "+v"(c));) }
        if (MODE == 2) { REP64(asm volatile("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %2, %2, %1" : "+v"(a), "+v"(b), "+v"(c));) }
        if (MODE == 3) { REP64(asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %2, %2, %1" : "+v"(a), "+v"(b), "+v"(c));) }
        if (MODE == 4) { REP64(asm volatile("v_lshlrev_b64 %0, %2, %0\n v_lshlrev_b64 %1, %2, %1" : "+v"(p), "+v"(q) : "v"(sh));) }
        if (MODE == 5) { REP64(asm volatile("v_lshrrev_b64 %0, %2, %0\n v_lshrrev_b64 %1, %2, %1" : "+v"(p), "+v"(q) : "v"(sh));) }
        if (MODE == 6) { REP64(asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %3, %3, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == 7) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %1, vcc" : "+v"(a), "+v"(b), "+v"(c)::"vcc");) }
        if (MODE == 8) { REP64(asm volatile("v_lshl_or_b32 %0, %0, %1, %2\n v_lshl_or_b32 %3, %3, %1, %2" : "+v"(a), "+v"(sh), "+v"(c), "+v"(d));) }
        if (MODE == 9) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(p), "+v"(q) : "v"(a), "v"(b) : "vcc");) }
        if (MODE == 10) { REP64(asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %1, vcc" : "+v"(a), "+v"(b), "+v"(c)::"vcc");) }
        if (MODE == 11) { REP64(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %1 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c));) }
        if (MODE == 12) { REP64(asm volatile("v_readlane_b32 s20, %0, 5\n v_writelane_b32 %1, s20, 7" : "+v"(a), "+v"(b)::"s20");) }
        if (MODE == 13) { REP64(asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %1, vcc" : "+v"(a), "+v"(b), "+v"(c)::"vcc");) }
        if (MODE == 14) { REP64(asm volatile("s_add_u32 s20, s20, 1\n s_lshl_b32 s21, s20, 2" ::: "s20", "s21", "scc");) }
        if (MODE == 15) { REP64(asm volatile("v_or_b32 %0, %0, %1\n v_and_b32 %2, %2, %1" : "+v"(a), "+v"(b), "+v"(c));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ (uint32_t)p ^ (uint32_t)(p >> 32) ^ (uint32_t)q ^ (uint32_t)(q >> 32);
}

template <int MODE>
static double run(uint32_t *o, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        k<MODE><<<256 * 4, 256>>>(o, iters, rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    // per SIMD: 4 workgroups per CU -> 4 waves per SIMD, each iters * 128 instructions
    double instr_per_simd = 4.0 * iters * 128.0;
    return ms * 1e-3 * 2.4e9 / instr_per_simd; /* cycles per wave instruction at 2.4 GHz */
}

int main() {
    uint32_t *o;
    hipMalloc(&o, 4 * 256 * 1024 * 4);
    const int iters = 2000;
    const char *names[] = {"v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_lshlrev_b64", "v_lshrrev_b64", "v_perm_b32", "v_cndmask_b32",
                           "v_lshl_or_b32", "v_mad_u64_u32", "v_add_co+addc", "v_mov_dpp", "readlane+writelane", "v_cmp+cndmask", "s_add+s_lshl (SALU)", "v_or/v_and"};
    double r[16];
    r[0] = run<0>(o, iters); r[1] = run<1>(o, iters); r[2] = run<2>(o, iters); r[3] = run<3>(o, iters);
    r[4] = run<4>(o, iters); r[5] = run<5>(o, iters); r[6] = run<6>(o, iters); r[7] = run<7>(o, iters);
    r[8] = run<8>(o, iters); r[9] = run<9>(o, iters); r[10] = run<10>(o, iters); r[11] = run<11>(o, iters);
    r[12] = run<12>(o, iters); r[13] = run<13>(o, iters); r[14] = run<14>(o, iters); r[15] = run<15>(o, iters);
    for (int i = 0; i < 16; i++) printf("%-22s %.2f cycles per wave instruction (if 2.4 GHz)\n", names[i], r[i]);
    return 0;
}
