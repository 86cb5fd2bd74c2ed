/* Exhaustive check of the two-correction constant division
 *   q0 = x*I; e1 = fma(-c,q0,x); q1 = fma(e1,I,q0); e2 = fma(-c,q1,x); q2 = fma(e2,I,q1)
 * against x / c for c = sqrtf(2), I = 1/c (optionally refined once). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
int main(void) {
    const float c = sqrtf(2.0f);
    float I = 1.0f / c;
    float e = __builtin_fmaf(-c, I, 1.0f);
    float I2 = __builtin_fmaf(e, I, I);
    printf("c=%a I=%a refined=%a\n", c, I, I2);
    for (int variant = 0; variant < 2; ++variant) {
        const float R = variant ? I2 : I;
        unsigned long long bad = 0;
        uint32_t lo = 0xffffffffu, hi = 0, lo_mid = 0xffffffffu, hi_mid = 0;
#pragma omp parallel for reduction(+ : bad) reduction(min : lo, lo_mid) reduction(max : hi, hi_mid) schedule(static)
        for (long long i = 0; i < (1LL << 32); ++i) {
            float x = u2f((uint32_t)i);
            float want = x / c;
            float q0 = x * R;
            float e1 = __builtin_fmaf(-c, q0, x);
            float q1 = __builtin_fmaf(e1, R, q0);
            float e2 = __builtin_fmaf(-c, q1, x);
            float got = __builtin_fmaf(e2, R, q1);
            int same = (f2u(want) == f2u(got)) || (want != want && got != got);
            if (!same) {
                bad++;
                uint32_t a = (uint32_t)i & 0x7fffffffu;
                if (a < lo) lo = a;
                if (a > hi) hi = a;
                if (a >= 0x0c800000u && a < 0x7f000000u) { if (a < lo_mid) lo_mid = a; if (a > hi_mid) hi_mid = a; }
            }
        }
        printf("variant %d: mismatches %llu, |x| bits range [0x%08x, 0x%08x]; within [2^-102,2^127): [0x%08x, 0x%08x]\n",
               variant, bad, lo, hi, lo_mid, hi_mid);
    }
    return 0;
}
