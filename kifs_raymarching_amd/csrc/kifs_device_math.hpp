// kifs_device_math.hpp -- f32 building blocks of the gfx950 raymarching kernels.
//
// Every function here is a fixed sequence of IEEE binary32 operations: +, -, *,
// correctly rounded / and sqrt, explicit fma, and integer moves of the bit
// pattern.  The translation unit is compiled with -ffp-contract=off, so the
// compiler neither fuses nor splits anything: what is written is what the VALU
// executes (v_fma_f32 only where fmaf_() appears).  That makes a frame a pure
// function of its 156 uniform bytes, reproducible on any IEEE machine -- the
// property the parity tests rely on.
//
// The WGSL builtins the reference shader calls (log, sin, cos, acos, pow, log2:
// julia.wgsl:26, gen_julia.wgsl:16,26,51-53, quaternions.wgsl:57-63,
// kifs.wgsl:90-133) are only ULP-bounded by the WGSL spec; here they are
// polynomial kernels (single-precision Cephes coefficients) that stay inside
// those bounds.  The hardware transcendentals (v_log_f32, v_sin_f32, ...) are
// deliberately not used: they are not reproducible off-chip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kifs_params.hpp"

namespace kifs {

#define KIFS_DEV __device__ __forceinline__

KIFS_DEV float fmaf_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
KIFS_DEV float sqrt_(float a) { return __builtin_sqrtf(a); }
KIFS_DEV float abs_(float a) { return __builtin_fabsf(a); }
KIFS_DEV float rint_(float a) { return __builtin_rintf(a); }
KIFS_DEV uint32_t bits(float f) { return __float_as_uint(f); }
KIFS_DEV float from_bits(uint32_t u) { return __uint_as_float(u); }
// WGSL leaves min/max with NaN to the implementation; these forms are the contract.
KIFS_DEV float min_(float a, float b) { return (b < a) ? b : a; }
KIFS_DEV float max_(float a, float b) { return (a < b) ? b : a; }
KIFS_DEV float clamp_(float e, float lo, float hi) { return min_(max_(e, lo), hi); }

KIFS_DEV float qnan() { return from_bits(0x7fc00000u); }
KIFS_DEV float pinf() { return from_bits(0x7f800000u); }

// Packed f32: one v_pk_* instruction does two lanes' worth of work per thread and costs a
// lone wave the same ~5 cycles as a scalar-f32 VALU op (tools/microbench/issue_cost.hip),
// so on the latency-bound tail of a frame it halves the critical path of whatever packs.
typedef float F2 __attribute__((ext_vector_type(2)));
KIFS_DEV F2 pk_fma(F2 a, F2 b, F2 c) { return __builtin_elementwise_fma(a, b, c); }

// dot products as fma chains, first component first
KIFS_DEV float dot(V2 a, V2 b) { return fmaf_(a.y, b.y, a.x * b.x); }
KIFS_DEV float dot(V3 a, V3 b) { return fmaf_(a.z, b.z, fmaf_(a.y, b.y, a.x * b.x)); }
KIFS_DEV float dot(V4 a, V4 b) {
    return fmaf_(a.w, b.w, fmaf_(a.z, b.z, fmaf_(a.y, b.y, a.x * b.x)));
}
KIFS_DEV float length(V2 a) { return sqrt_(dot(a, a)); }
KIFS_DEV float length(V3 a) { return sqrt_(dot(a, a)); }
KIFS_DEV float length(V4 a) { return sqrt_(dot(a, a)); }
KIFS_DEV V3 normalize(V3 a) {
    float l = length(a);
    return V3{a.x / l, a.y / l, a.z / l};
}

// ---- logarithms ---------------------------------------------------------------
// positive finite x -> mantissa in [0.5,1) and exponent
KIFS_DEV float split_pos(float x, int& e) {
    int ex = 0;
    uint32_t ix = bits(x);
    if (ix < 0x00800000u) {  // denormal
        x = x * 8388608.0f;
        ix = bits(x);
        ex = -23;
    }
    e = ex + int(ix >> 23) - 126;
    return from_bits((ix & 0x007fffffu) | 0x3f000000u);
}

KIFS_DEV float log_poly(float m) {
    float p = 7.0376836292E-2f;
    p = fmaf_(p, m, -1.1514610310E-1f);
    p = fmaf_(p, m, 1.1676998740E-1f);
    p = fmaf_(p, m, -1.2420140846E-1f);
    p = fmaf_(p, m, 1.4249322787E-1f);
    p = fmaf_(p, m, -1.6668057665E-1f);
    p = fmaf_(p, m, 2.0000714765E-1f);
    p = fmaf_(p, m, -2.4999993993E-1f);
    p = fmaf_(p, m, 3.3333331174E-1f);
    return p;
}

// reduced argument m in [sqrt(1/2), sqrt(2)) - 1 and its exponent
KIFS_DEV float log_reduce(float x, int& e) {
    float m = split_pos(x, e);
    if (m < 0.70710678118654752440f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    return m;
}

// log of a positive, finite, normal x: the whole function for almost every call.
KIFS_DEV float log_normal(float x) {
    uint32_t ix = bits(x);
    int e = int(ix >> 23) - 126;
    float m = from_bits((ix & 0x007fffffu) | 0x3f000000u);
    if (m < 0.70710678118654752440f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float y = (log_poly(m) * m) * z;
    float fe = float(e);
    y = fmaf_(fe, -2.12194440e-4f, y);
    y = fmaf_(-0.5f, z, y);
    float r = m + y;
    return fmaf_(fe, 0.693359375f, r);
}

KIFS_DEV float log_general(float x) {
    if (x != x) return x + x;
    if (x < 0.0f) return qnan();
    if (x == 0.0f) return -pinf();
    if (x == pinf()) return x;
    int e;
    float m = log_reduce(x, e);
    float z = m * m;
    float y = (log_poly(m) * m) * z;
    float fe = float(e);
    y = fmaf_(fe, -2.12194440e-4f, y);
    y = fmaf_(-0.5f, z, y);
    float r = m + y;
    return fmaf_(fe, 0.693359375f, r);
}

// Zero, denormal, negative, infinite and NaN arguments are rare: the wave takes the general
// routine (identical results on normal inputs) only if some lane holds one.
KIFS_DEV float log_(float x) {
    const bool ordinary = (x >= 1.17549435e-38f) && (x <= 3.40282347e38f);
    if (__builtin_amdgcn_ballot_w64(!ordinary) != 0ull) return log_general(x);
    return log_normal(x);
}

KIFS_DEV float log2_(float x) {
    if (x != x) return x + x;
    if (x < 0.0f) return qnan();
    if (x == 0.0f) return -pinf();
    if (x == pinf()) return x;
    int e;
    float m = log_reduce(x, e);
    float z = m * m;
    float y = (log_poly(m) * m) * z;
    y = fmaf_(-0.5f, z, y);
    const float L = 0.44269504088896340735992f;  // log2(e) - 1
    float r = y * L;
    r = fmaf_(m, L, r);
    r = r + y;
    r = r + m;
    return r + float(e);
}

KIFS_DEV float exp2_(float x) {
    if (x != x) return x + x;
    if (x > 127.99999f) return pinf();
    if (x < -150.0f) return 0.0f;
    float n = rint_(x);
    float r = x - n;
    float p = 1.535336188319500E-004f;
    p = fmaf_(p, r, 1.339887440266574E-003f);
    p = fmaf_(p, r, 9.618437357674640E-003f);
    p = fmaf_(p, r, 5.550332471162809E-002f);
    p = fmaf_(p, r, 2.402264791363012E-001f);
    p = fmaf_(p, r, 6.931472028550421E-001f);
    p = fmaf_(p, r, 1.0f);
    int ni = int(n);
    int n1 = ni / 2, n2 = ni - n1;
    float s1 = from_bits(uint32_t(n1 + 127) << 23);
    float s2 = from_bits(uint32_t(n2 + 127) << 23);
    return (p * s1) * s2;
}

// WGSL defines pow's accuracy by exp2(y * log2(x)); that is the implementation.
KIFS_DEV float pow_(float x, float y) { return exp2_(y * log2_(x)); }

// ---- sin / cos ----------------------------------------------------------------
KIFS_DEV int reduce_pio2(float x, float& r) {
    const float TWO_OVER_PI = 0.63661977236758134308f;
    const float P1 = 1.5703125f;
    const float P2 = 4.837512969970703125e-4f;
    const float P3 = 7.54978995489188216e-8f;
    float n = rint_(x * TWO_OVER_PI);
    float t = fmaf_(-n, P1, x);
    t = fmaf_(-n, P2, t);
    t = fmaf_(-n, P3, t);
    r = t;
    // The quadrant is n mod 4.  The arithmetic contract writes it int(n - 4 rint(n / 4)) & 3; every caller bounds |x| by 2^20, so n
    // is an integer below 2^20 in magnitude: n / 4, its rounding, the product and the difference are exact, the
    // difference is congruent to n modulo 4, and the low two bits of a two's-complement integer ARE its residue
    // modulo 4 -- int(n) & 3 is the same number for every such n, four instructions shorter.
    return int(n) & 3;
}

KIFS_DEV float sin_kernel(float r) {
    float z = r * r;
    float p = -1.9515295891E-4f;
    p = fmaf_(p, z, 8.3321608736E-3f);
    p = fmaf_(p, z, -1.6666654611E-1f);
    return fmaf_(p * z, r, r);
}

KIFS_DEV float cos_kernel(float r) {
    float z = r * r;
    float p = 2.443315711809948E-005f;
    p = fmaf_(p, z, -1.388731625493765E-003f);
    p = fmaf_(p, z, 4.166664568298827E-002f);
    float y = (p * z) * z;
    y = fmaf_(-0.5f, z, y);
    return y + 1.0f;
}

KIFS_DEV float sin_(float x) {
    if (!(abs_(x) <= 1048576.0f)) return x - x;
    float r;
    int q = reduce_pio2(x, r);
    float s = (q & 1) ? cos_kernel(r) : sin_kernel(r);
    return (q & 2) ? -s : s;
}

// sin_ without a branch: both kernels evaluated, everything else selects.  The same values (the kernels are
// pure functions of r; lanes beyond the reduction's range reduce 0 instead and get x - x at the end), but
// straight-line code: four of these interleave in the bunny network, where sin_'s three exec-mask regions each
// would have fenced the scheduler and, in a diverged wave, run both kernels anyway.
KIFS_DEV float sin_flat(float x) {
    const bool ok = abs_(x) <= 1048576.0f;
    float r;
    const int q = reduce_pio2(ok ? x : 0.0f, r);
    const float sk = sin_kernel(r), ck = cos_kernel(r);
    const float s = (q & 1) ? ck : sk;
    const float v = (q & 2) ? -s : s;
    return ok ? v : x - x;
}

KIFS_DEV float cos_(float x) {
    if (!(abs_(x) <= 1048576.0f)) return x - x;
    float r;
    int q = reduce_pio2(x, r);
    float c = (q & 1) ? sin_kernel(r) : cos_kernel(r);
    return ((q + 1) & 2) ? -c : c;
}

// sin_(x) and cos_(x) at once: one range reduction, each kernel evaluated once.  Bit-identical to
// the two separate calls (same r, same quadrant, same kernels, same selects).
KIFS_DEV void sincos_(float x, float& s, float& c) {
    if (!(abs_(x) <= 1048576.0f)) {
        s = x - x;
        c = x - x;
        return;
    }
    float r;
    int q = reduce_pio2(x, r);
    const float sk = sin_kernel(r), ck = cos_kernel(r);
    const float ss = (q & 1) ? ck : sk;
    const float cc = (q & 1) ? sk : ck;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}

// ---- acos ---------------------------------------------------------------------
KIFS_DEV float asin_poly(float x, float z) {
    float p = 4.2163199048E-2f;
    p = fmaf_(p, z, 2.4181311049E-2f);
    p = fmaf_(p, z, 4.5470025998E-2f);
    p = fmaf_(p, z, 7.4953002686E-2f);
    p = fmaf_(p, z, 1.6666752422E-1f);
    return fmaf_(p * z, x, x);
}

KIFS_DEV float acos_(float x) {
    const float PI_F = 3.14159265358979323846f;
    const float PIO2_F = 1.57079632679489661923f;
    if (x != x) return x + x;
    if (x > 1.0f || x < -1.0f) return qnan();
    if (x > 0.5f) {
        float z = 0.5f * (1.0f - x);
        return 2.0f * asin_poly(sqrt_(z), z);
    }
    if (x < -0.5f) {
        float z = 0.5f * (1.0f + x);
        return PI_F - 2.0f * asin_poly(sqrt_(z), z);
    }
    return PIO2_F - asin_poly(x, x * x);
}

// ---- straight-line cores ---------------------------------------------------------
// log2_, exp2_, acos_ and sincos_ start with special-case tests and (acos_) a three-way range split;
// compiled as written each is a chain of divergent branches.  The *_core forms are the same
// operations on the same operands for ORDINARY arguments -- the only ones an orbit meets in
// practice -- with the range split done by selects, and *_ordinary says whether an argument is one.
// A caller evaluates a whole step with the cores, then asks once per wave whether every lane's
// arguments were ordinary, and only otherwise repeats the step with the general forms (see
// quat_pow_shared in kifs_scene.hpp).  On any other argument a core returns garbage, never a trap.
// Correctly rounded 1/b and sqrt(x) for MID-RANGE operands, 2^-60 <= b, x < 2^60: exactly the sequences
// the compiler expands `/` and sqrt into (v_div_scale / v_rcp / Newton / v_div_fmas / v_div_fixup; v_sqrt
// + one-ulp fix-up with a 2^32 pre-scaling and a zero/infinity patch) minus the parts that are no-ops
// in that range -- the scaling (exponents too close to 0 for v_div_scale to act, quotient far from the
// denormals), the fix-up's special cases, the pre-scaling (x >= 2^-96).  Same bits, six and seven
// instructions and two and three compares shorter.  sqrt_mid(0) = 0 as well (the candidates fail both
// tests), which the acos tail relies on.
KIFS_DEV bool mid_range(float x) { return (bits(x) - 0x21800000u) < (0x5d800000u - 0x21800000u); }  // [2^-60, 2^60)
KIFS_DEV float rcp_mid(float b) {
    float r = __builtin_amdgcn_rcpf(b);
    float e = fmaf_(-b, r, 1.0f);
    r = fmaf_(e, r, r);
    float q = r;  // 1 * r
    float rem = fmaf_(-b, q, 1.0f);
    q = fmaf_(rem, r, q);
    rem = fmaf_(-b, q, 1.0f);
    return fmaf_(rem, r, q);
}
KIFS_DEV float sqrt_mid(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = from_bits(bits(s) - 1u), sp = from_bits(bits(s) + 1u);  // candidates one ulp either side
    const float rm = fmaf_(-sm, s, x), rp = fmaf_(-sp, s, x);
    float r = (0.0f >= rm) ? sm : s;
    return (0.0f < rp) ? sp : r;
}

KIFS_DEV bool log2_ordinary(float x) { return (x >= 1.17549435e-38f) && (x <= 3.40282347e38f); }  // positive normal
KIFS_DEV float log2_core(float x) {
    // log_reduce() for a NORMAL x: without split_pos' 2^23 pre-scaling of denormals -- a core's argument is ordinary
    // by definition (quat_pow_step checks |q|^2 >= 2^-60 once per step), so the scaling selects nothing
    const uint32_t ix = bits(x);
    int e = int(ix >> 23) - 126;
    float m = from_bits((ix & 0x007fffffu) | 0x3f000000u);
    if (m < 0.70710678118654752440f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float y = (log_poly(m) * m) * z;
    y = fmaf_(-0.5f, z, y);
    const float L = 0.44269504088896340735992f;  // log2(e) - 1
    float r = y * L;
    r = fmaf_(m, L, r);
    r = r + y;
    r = r + m;
    return r + float(e);
}

KIFS_DEV bool exp2_ordinary(float x) { return (x <= 127.99999f) && (x >= -150.0f); }  // false for NaN
KIFS_DEV float exp2_core(float x) {
    float n = rint_(x);
    float r = x - n;
    float p = 1.535336188319500E-004f;
    p = fmaf_(p, r, 1.339887440266574E-003f);
    p = fmaf_(p, r, 9.618437357674640E-003f);
    p = fmaf_(p, r, 5.550332471162809E-002f);
    p = fmaf_(p, r, 2.402264791363012E-001f);
    p = fmaf_(p, r, 6.931472028550421E-001f);
    p = fmaf_(p, r, 1.0f);
    // exp2_ scales in two exact-or-once-rounded steps, (p 2^n1) 2^n2 with n1 + n2 = n, because 2^n itself may not be a
    // float (n = 128, n < -126).  For an ordinary argument n is in [-128, 128] and p in (0.7, 1.5): the first product
    // is exact (a normal number), the second rounds at most once -- when the result is denormal -- or overflows to
    // infinity: that is scalbn(p, n) by definition, which v_ldexp_f32 computes (f32 denormals are on), one instruction
    // for eight.  tests/test_gpu_parity_points.py::test_straight_line_cores_... sweeps it against the contract's exp2.
    return __builtin_ldexpf(p, int(n));
}

KIFS_DEV bool acos_ordinary(float x) { return (x >= -1.0f) && (x <= 1.0f); }  // false for NaN
KIFS_DEV float acos_core(float x) {
    const float PI_F = 3.14159265358979323846f;
    const float PIO2_F = 1.57079632679489661923f;
    const bool hi = x > 0.5f, lo = x < -0.5f;
    // |x| > 0.5: z = (1 - |x|) / 2 written as acos_ writes it for each sign
    const float zt = 0.5f * (hi ? (1.0f - x) : (1.0f + x));
    const bool tail = hi || lo;
    const float z = tail ? zt : x * x;
    const float a = tail ? sqrt_mid(zt) : x;  // zt is 0 or in [2^-25, 0.75] for |x| <= 1
    const float r = asin_poly(a, z);
    const float two_r = 2.0f * r;
    return hi ? two_r : (lo ? PI_F - two_r : PIO2_F - r);
}

KIFS_DEV bool sincos_ordinary(float x) { return abs_(x) <= 1048576.0f; }  // false for NaN
KIFS_DEV void sincos_core(float x, float& s, float& c) {
    float r;
    const int q = reduce_pio2(x, r);
    const float sk = sin_kernel(r), ck = cos_kernel(r);
    const float ss = (q & 1) ? ck : sk;
    const float cc = (q & 1) ? sk : ck;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}

// ---- colour target ---------------------------------------------------------------
// UNORM8: clamp, scale, +0.5, truncate; NaN -> 0.
KIFS_DEV uint32_t unorm8(float x) {
    float c = (x >= 0.0f) ? x : 0.0f;
    c = (c > 1.0f) ? 1.0f : c;
    return uint32_t(int(c * 255.0f + 0.5f));
}

// sRGB UNORM8: code = number of thresholds t[k] (k = 1..255) with x >= t[k], found
// by an 8-step branch-free search over the table staged in LDS.  t[k] is the
// smallest f32 whose ideal encoding rounds to >= k, so this is the exactly
// rounded sRGB conversion; NaN compares false everywhere -> 0.
KIFS_DEV uint32_t srgb8(float x, const float* __restrict__ t) {
    uint32_t k = 0;
#pragma unroll
    for (uint32_t step = 128; step >= 1; step >>= 1) k += (x >= t[k + step]) ? step : 0u;
    return k;
}

}  // namespace kifs
