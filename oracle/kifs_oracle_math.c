/*
 * kifs_oracle_math.c -- f32-only elementary functions of the ORACLE (test
 * infrastructure; see kifs_oracle.h).
 *
 * The reference calls WGSL builtins (log, sin, cos, acos, pow, log2) whose
 * accuracy is only ULP-bounded by the WGSL spec and decided by naga + the GPU
 * driver (Cargo.lock:1275-1276 naga 25.0.1).  For a reproducible oracle each
 * one is pinned here to a fixed sequence of binary32 operations (+, *, fma,
 * exact integer bit moves).  Polynomial coefficients are the single-precision
 * Cephes ones (S. Moshier, cephes/single: logf.c, sinf.c, asinf.c, exp2f.c,
 * log2f.c), which keep every function within ~1-2 ULP of the true value on
 * the ranges the shader reaches -- inside the 3-ULP / 2^-21-abs envelopes the
 * WGSL spec grants the builtins.  tests/test_oracle_math.py measures the error
 * against float64.
 */
#include "kifs_oracle.h"

#include <math.h>
#include <string.h>

static inline float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

#define KOR_NAN (u2f(0x7fc00000u))
#define KOR_INF (u2f(0x7f800000u))

/* frexp for positive finite normal-or-denormal x: x = m * 2^e, m in [0.5, 1). */
static inline float frexp_pos(float x, int* e_out) {
    int e = 0;
    uint32_t ix = f2u(x);
    if (ix < 0x00800000u) { /* denormal: scale by 2^23 (exact) */
        x = x * 8388608.0f;
        ix = f2u(x);
        e = -23;
    }
    e += (int)(ix >> 23) - 126;
    *e_out = e;
    return u2f((ix & 0x007fffffu) | 0x3f000000u);
}

/* Shared core of log/log2: reduces x to m in [sqrt(1/2), sqrt(2)) - 1 and
 * returns the polynomial y with log(1+m) = m - m^2/2 + y. */
static inline float log_poly(float m) {
    float p = 7.0376836292E-2f;
    p = fma_(p, m, -1.1514610310E-1f);
    p = fma_(p, m, 1.1676998740E-1f);
    p = fma_(p, m, -1.2420140846E-1f);
    p = fma_(p, m, 1.4249322787E-1f);
    p = fma_(p, m, -1.6668057665E-1f);
    p = fma_(p, m, 2.0000714765E-1f);
    p = fma_(p, m, -2.4999993993E-1f);
    p = fma_(p, m, 3.3333331174E-1f);
    return p;
}

float kor_logf(float x) {
    if (x != x) return x + x;
    if (x < 0.0f) return KOR_NAN;
    if (x == 0.0f) return -KOR_INF;
    if (x == KOR_INF) return x;
    int e;
    float m = frexp_pos(x, &e);
    if (m < 0.70710678118654752440f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float y = (log_poly(m) * m) * z;
    float fe = (float)e;
    y = fma_(fe, -2.12194440e-4f, y);
    y = fma_(-0.5f, z, y);
    float r = m + y;
    r = fma_(fe, 0.693359375f, r);
    return r;
}

float kor_log2f(float x) {
    if (x != x) return x + x;
    if (x < 0.0f) return KOR_NAN;
    if (x == 0.0f) return -KOR_INF;
    if (x == KOR_INF) return x;
    int e;
    float m = frexp_pos(x, &e);
    if (m < 0.70710678118654752440f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float y = (log_poly(m) * m) * z;
    y = fma_(-0.5f, z, y);
    /* log2(1+m) = (m + y) * log2(e), with log2(e) = 1 + 0.44269504088896340735992 */
    const float L = 0.44269504088896340735992f;
    float r = y * L;
    r = fma_(m, L, r);
    r = r + y;
    r = r + m;
    r = r + (float)e;
    return r;
}

float kor_exp2f(float x) {
    if (x != x) return x + x;
    if (x > 127.99999f) return KOR_INF;  /* >= 128 overflows; 127.99999f is the largest f32 < 128 */
    if (x < -150.0f) return 0.0f;
    float n = rintf(x);
    float r = x - n; /* exact, |r| <= 0.5 */
    float p = 1.535336188319500E-004f;
    p = fma_(p, r, 1.339887440266574E-003f);
    p = fma_(p, r, 9.618437357674640E-003f);
    p = fma_(p, r, 5.550332471162809E-002f);
    p = fma_(p, r, 2.402264791363012E-001f);
    p = fma_(p, r, 6.931472028550421E-001f);
    p = fma_(p, r, 1.0f);
    /* scale by 2^n in two exact steps so denormal results round once */
    int ni = (int)n; /* |n| <= 150 */
    int n1 = ni / 2, n2 = ni - n1;
    float s1 = u2f((uint32_t)(n1 + 127) << 23);
    float s2 = u2f((uint32_t)(n2 + 127) << 23);
    return (p * s1) * s2;
}

/* pow(x, y) = exp2(y * log2(x)): exactly the formula the WGSL spec defines
 * the builtin's accuracy by. */
float kor_powf(float x, float y) { return kor_exp2f(y * kor_log2f(x)); }

/* ---- sin / cos ------------------------------------------------------------ */
/* Cody-Waite reduction by pi/2 in three f32 pieces (fma keeps each step exact
 * for |n| < 2^15 or so), then the Cephes minimax polynomials on [-pi/4, pi/4]. */
static inline int reduce_pio2(float x, float* r_out) {
    const float TWO_OVER_PI = 0.63661977236758134308f;
    const float P1 = 1.5703125f;                 /* pi/2 high bits (exact in 8 bits) */
    const float P2 = 4.837512969970703125e-4f;   /* next bits */
    const float P3 = 7.54978995489188216e-8f;    /* remainder */
    float n = rintf(x * TWO_OVER_PI);
    float r = fma_(-n, P1, x);
    r = fma_(-n, P2, r);
    r = fma_(-n, P3, r);
    *r_out = r;
    /* quadrant = n mod 4 computed in float (exact for |n| < 2^24) */
    float q = n - 4.0f * rintf(n * 0.25f); /* in [-2, 2] */
    int qi = (int)q;
    return qi & 3;
}

static inline float sin_kernel(float r) {
    float z = r * r;
    float p = -1.9515295891E-4f;
    p = fma_(p, z, 8.3321608736E-3f);
    p = fma_(p, z, -1.6666654611E-1f);
    return fma_(p * z, r, r);
}

static inline float cos_kernel(float r) {
    float z = r * r;
    float p = 2.443315711809948E-005f;
    p = fma_(p, z, -1.388731625493765E-003f);
    p = fma_(p, z, 4.166664568298827E-002f);
    float y = (p * z) * z;
    y = fma_(-0.5f, z, y);
    return y + 1.0f;
}

float kor_sinf(float x) {
    if (!(fabsf(x) <= 1048576.0f)) return x - x; /* beyond 2^20 the 3-term reduction is meaningless: finite -> 0, inf/NaN -> NaN */
    float r;
    int q = reduce_pio2(x, &r);
    float s = (q & 1) ? cos_kernel(r) : sin_kernel(r);
    return (q & 2) ? -s : s;
}

float kor_cosf(float x) {
    if (!(fabsf(x) <= 1048576.0f)) return x - x;
    float r;
    int q = reduce_pio2(x, &r);
    float c = (q & 1) ? sin_kernel(r) : cos_kernel(r);
    return ((q + 1) & 2) ? -c : c;
}

/* ---- acos ----------------------------------------------------------------- */
static inline float asin_poly(float x, float z) { /* asin(x), z = x*x, |x| <= 0.5 */
    float p = 4.2163199048E-2f;
    p = fma_(p, z, 2.4181311049E-2f);
    p = fma_(p, z, 4.5470025998E-2f);
    p = fma_(p, z, 7.4953002686E-2f);
    p = fma_(p, z, 1.6666752422E-1f);
    return fma_(p * z, x, x);
}

float kor_acosf(float x) {
    const float PI_F = 3.14159265358979323846f;
    const float PIO2_F = 1.57079632679489661923f;
    if (x != x) return x + x;
    if (x > 1.0f || x < -1.0f) return KOR_NAN;
    if (x > 0.5f) {
        float z = 0.5f * (1.0f - x);
        float s = sqrtf(z);
        return 2.0f * asin_poly(s, z);
    }
    if (x < -0.5f) {
        float z = 0.5f * (1.0f + x);
        float s = sqrtf(z);
        return PI_F - 2.0f * asin_poly(s, z);
    }
    return PIO2_F - asin_poly(x, x * x);
}
