/* Exhaustive check over all 2^32 binary32 inputs x that
 *     q = x*inv;  r = fma(-q, c, x);  q' = fma(r, inv, q)
 * equals the correctly rounded x / c, for c = sqrtf(2) and inv = 1.0f / c (the constants of
 * the tetrahedral fold, kifs.wgsl:6-14 with normals (1,1,0) etc.).  Prints every mismatch class.
 * Build: gcc -O2 -mfma -ffp-contract=off -fopenmp div_const_exhaustive.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
int main(void) {
    const float c = sqrtf(2.0f), inv = 1.0f / c;
    printf("c = %a, inv = %a\n", c, inv);
    unsigned long long bad = 0, bad_normal_range = 0;
    uint32_t min_bad_abs = 0xffffffffu, max_bad_abs = 0;
#pragma omp parallel for reduction(+ : bad, bad_normal_range) reduction(min : min_bad_abs) reduction(max : max_bad_abs) schedule(static)
    for (long long i = 0; i < (1LL << 32); ++i) {
        float x = u2f((uint32_t)i);
        float want = x / c;
        float q = x * inv;
        float r = __builtin_fmaf(-q, c, x);
        float got = __builtin_fmaf(r, inv, q);
        int same = (f2u(want) == f2u(got)) || (want != want && got != got);
        if (!same) {
            bad++;
            uint32_t a = (uint32_t)i & 0x7fffffffu;
            if (a < min_bad_abs) min_bad_abs = a;
            if (a > max_bad_abs) max_bad_abs = a;
            if (a >= 0x01800000u && a < 0x7f000000u) bad_normal_range++;
        }
    }
    printf("mismatches: %llu (|x| bits from 0x%08x to 0x%08x); in [2^-124, 2^127): %llu\n", bad,
           min_bad_abs, max_bad_abs, bad_normal_range);
    return 0;
}
