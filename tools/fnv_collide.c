/*
 * fnv_collide.c -- finds two different 13-character names with the same 64-bit FNV-1a hash, i.e. the same cov_name_hash
 * (paffy_amd/csrc/coverage_kernel.h) and the same shard.name_hash: test data for the name check behind the hashes
 * (tests/golden/fnv_collision.txt was made by this program; tests/test_gpu_tile.py, tests/test_gpu_chain.py use it).
 *
 * Method: parallel collision search with distinguished points (van Oorschot & Wiener). f(x) = FNV-1a(encode(x)), encode = 13 base-32
 * characters of the 64-bit value (injective), so f(a) == f(b) with a != b is a pair of different names with equal hashes. About
 * sqrt(pi/2 * 2^64) = 5.4e9 evaluations: a minute on eight cores.
 *
 *   gcc -O2 -fopenmp -o /tmp/fnv_collide tools/fnv_collide.c && /tmp/fnv_collide
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

static const char ALPHA[33] = "abcdefghijklmnopqrstuvwxyz012345";
#define NAME_LEN 13
#define DP_BITS 22

static inline void encode(uint64_t x, char *s) {
    for (int i = 0; i < NAME_LEN; i++) {
        s[i] = ALPHA[x & 31u];
        x >>= 5;
    }
}
static inline uint64_t fnv_raw(uint64_t x) { /* the FNV-1a state after the 13 bytes: the rest of cov_name_hash is a bijection of it */
    uint64_t h = 0xcbf29ce484222325ull;
    for (int i = 0; i < NAME_LEN; i++) {
        h = (h ^ (uint64_t)(unsigned char)ALPHA[x & 31u]) * 0x100000001b3ull;
        x >>= 5;
    }
    return h;
}
static uint64_t name_hash(const char *s, int len) { /* cov_name_hash / shard.name_hash */
    uint64_t h = 0xcbf29ce484222325ull;
    for (int i = 0; i < len; i++) h = (h ^ (uint64_t)(unsigned char)s[i]) * 0x100000001b3ull;
    h = (h ^ (0x100u + (uint64_t)len)) * 0x100000001b3ull;
    return h ^ (h >> 29);
}

typedef struct {
    uint64_t dp, start, steps;
} Trail;
#define TAB_BITS 16
static Trail table[1u << TAB_BITS];

int main(void) {
    volatile int done = 0;
    uint64_t a_out = 0, b_out = 0;
    memset(table, 0, sizeof(table));
#pragma omp parallel
    {
        uint64_t seed = 0x9E3779B97F4A7C15ull * (uint64_t)(omp_get_thread_num() + 1);
        while (!done) {
            seed = seed * 6364136223846793005ull + 1442695040888963407ull;
            uint64_t start = seed, x = start, steps = 0;
            while (!done && steps < (1ull << (DP_BITS + 5))) {
                x = fnv_raw(x);
                steps++;
                if ((x & ((1ull << DP_BITS) - 1)) == 0) break;
            }
            if (done || (x & ((1ull << DP_BITS) - 1)) != 0) continue;
            Trail other = {0, 0, 0};
            int hit = 0;
#pragma omp critical
            {
                uint32_t slot = (uint32_t)((x >> DP_BITS) & ((1u << TAB_BITS) - 1));
                for (;;) {
                    if (table[slot].steps == 0) {
                        table[slot].dp = x;
                        table[slot].start = start;
                        table[slot].steps = steps;
                        break;
                    }
                    if (table[slot].dp == x) {
                        other = table[slot];
                        hit = other.start != start;
                        break;
                    }
                    slot = (slot + 1) & ((1u << TAB_BITS) - 1);
                }
            }
            if (!hit) continue;
            /* two trails end in the same point: bring them to the same distance from it, then walk together */
            uint64_t a = start, na = steps, b = other.start, nb = other.steps;
            while (na > nb) { a = fnv_raw(a); na--; }
            while (nb > na) { b = fnv_raw(b); nb--; }
            if (a == b) continue; /* one trail is a tail of the other: no collision here */
            while (na > 0) {
                const uint64_t fa = fnv_raw(a), fb = fnv_raw(b);
                if (fa == fb) break;
                a = fa;
                b = fb;
                na--;
            }
            if (na > 0 && a != b) {
#pragma omp critical
                {
                    if (!done) {
                        a_out = a;
                        b_out = b;
                        done = 1;
                    }
                }
            }
        }
    }
    char s1[NAME_LEN + 1] = {0}, s2[NAME_LEN + 1] = {0};
    encode(a_out, s1);
    encode(b_out, s2);
    if (strcmp(s1, s2) == 0 || name_hash(s1, NAME_LEN) != name_hash(s2, NAME_LEN)) {
        fprintf(stderr, "no collision found\n");
        return 1;
    }
    printf("%s %s %016llx\n", s1, s2, (unsigned long long)name_hash(s1, NAME_LEN));
    return 0;
}
