// kifs_scene.hpp -- device-side scene: the signed-distance functions, normals and
// the sphere-tracing loop of the reference fragment shader, written for wave64.
//
// What each function computes is fixed by the reference (paths under
// src/shaders/ of LesbianLemon/kifs-raymarching):
//   julia_sdf / julia_normal        julia.wgsl:5-27 / :29-56
//   genjulia_sdf / genjulia_normal  gen_julia.wgsl:5-27 / :30-55, quaternions.wgsl:57-63
//   primitive SDFs, fold, Sierpinski, bunny, central-difference normal   kifs.wgsl:16-167
//   ray set-up and march            dependencies/entry.wgsl:49-59 / :6-29
// How it is computed is this library's own: the variant (fractal group x
// primitive) is a template parameter so the uniform `switch` of kifs.wgsl:139-155
// and the pipeline choice of graphics.rs:310-321 vanish at compile time; all
// frame constants sit in SGPRs (kernarg); per-pixel state is registers only.
#pragma once

#include "kifs_bunny_weights.h"
#include "kifs_device_math.hpp"

namespace kifs {

// ---- quaternion helpers (quaternions.wgsl) ----------------------------------------------
// Evaluation order of the quaternion step (a legal reading of quaternions.wgsl:22-28,42-50,
// chosen because it is exactly what the packed-f32 loop below computes):
//     s = y*y + z*z;  d = fma(w,w,s) = |ijk|^2;  |q|^2 = fma(x,x,d)
//     q^2 + c: real = fma(x,x,-d) + c.x;  ijk = fma(2x, ijk, c.ijk)
KIFS_DEV float quat_ijk2(V4 q) { return fmaf_(q.w, q.w, q.y * q.y + q.z * q.z); }
KIFS_DEV float quat_norm2(V4 q) { return fmaf_(q.x, q.x, quat_ijk2(q)); }
KIFS_DEV V4 quat_sq_add(V4 q, V4 c) {
    float tr = 2.0f * q.x;
    return V4{fmaf_(q.x, q.x, -quat_ijk2(q)) + c.x, fmaf_(tr, q.y, c.y), fmaf_(tr, q.z, c.z),
              fmaf_(tr, q.w, c.w)};
}

// quat_pow (quaternions.wgsl:57-63) with the work it shares with its callers made explicit (the
// arithmetic contract): d = |ijk|^2 and qs = |q|^2 come from the caller;
// length(q) = sqrt(qs); the two divisions are products with correctly rounded reciprocals;
// L = log2(qs) is taken once: pow(norm, x) = exp2(x L / 2), and `dq_factor` receives
// gen_julia.wgsl:16's pow(qs, x - 1) = exp2((x - 1) L).
// This is the inner loop of the slowest pipeline and a lone wave pays ~35 cycles for every taken
// branch, so the step runs on the straight-line cores of kifs_device_math.hpp and checks ONCE, at
// its end, that every lane's arguments were ordinary; a wave with a NaN, an infinity, a zero or a
// denormal anywhere repeats the step with the general functions (same operations, same bits for
// ordinary lanes).
template <bool GENERAL>
KIFS_DEV V4 quat_pow_step(V4 q, float d, float qs, float x, float& dq_factor, bool& ordinary) {
    const float L = GENERAL ? log2_(qs) : log2_core(qs);
    const float e1 = (x - 1.0f) * L;
    dq_factor = GENERAL ? exp2_(e1) : exp2_core(e1);
    const float inv = GENERAL ? 1.0f / sqrt_(qs) : rcp_mid(sqrt_mid(qs));
    const float ca = q.x * inv;
    const float phi = GENERAL ? acos_(ca) : acos_core(ca);
    const float ninv = GENERAL ? 1.0f / sqrt_(d) : rcp_mid(sqrt_mid(d));
    const V3 n{q.y * ninv, q.z * ninv, q.w * ninv};
    const float e2 = x * (0.5f * L);
    const float pw = GENERAL ? exp2_(e2) : exp2_core(e2);
    const float a = x * phi;
    float cs, sn;
    if constexpr (GENERAL) sincos_(a, sn, cs);
    else sincos_core(a, sn, cs);
    // every argument a core saw was in its range: |q|^2 and |ijk|^2 mid-range (then their square roots
    // are too, and L is finite), both exponents inside (-128, 128) (a NaN exponent means a NaN L: caught
    // by the first test; v_max skips NaNs), |cos phi| <= 1 (false for NaN); the sin/cos argument x phi is
    // then at most 10 pi for the powers the caller lets through
    ordinary = (max(bits(qs) - 0x21800000u, bits(d) - 0x21800000u) < (0x5d800000u - 0x21800000u)) &&
               (__builtin_fmaxf(abs_(e1), abs_(e2)) < 127.99999f) && (abs_(ca) <= 1.0f);
    return V4{pw * cs, pw * (n.x * sn), pw * (n.y * sn), pw * (n.z * sn)};
}

// `lanes`: the lanes whose result is used (the others may hold anything and do not force the general path).
#ifdef KIFS_EVAL_COUNT
static __device__ unsigned long long g_pow_counts[8];  // [0] wave-level steps, [1] of which repeated with the general functions, [2] lanes at [0]
#endif
KIFS_DEV V4 quat_pow_shared(V4 q, float d, float qs, float x, float& dq_factor, bool lanes = true) {
    bool ordinary = false;
    V4 t{};
    const bool power_ok = abs_(x) <= 1.0e5f;  // uniform: |x phi| <= pi |x| stays inside the sin/cos cores' range
    if (__builtin_expect(power_ok, 1)) t = quat_pow_step<false>(q, d, qs, x, dq_factor, ordinary);
#ifdef KIFS_EVAL_COUNT
    {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(true);
        if (__lane_id() == uint32_t(__builtin_ctzll(m))) {
            atomicAdd(&g_pow_counts[0], 1ull);
            atomicAdd(&g_pow_counts[2], (unsigned long long)__builtin_popcountll(m));
            if (__builtin_amdgcn_ballot_w64(lanes && !ordinary) != 0ull) atomicAdd(&g_pow_counts[1], 1ull);
        }
    }
#endif
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(lanes && !ordinary) != 0ull, 0))
        t = quat_pow_step<true>(q, d, qs, x, dq_factor, ordinary);
    return t;
}

KIFS_DEV V4 quat_pow(V4 q, float x) {
    const float d = quat_ijk2(q);
    const float qs = fmaf_(q.x, q.x, d);
    float unused;
    return quat_pow_shared(q, d, qs, x, unused);
}

// ---- Julia ---------------------------------------------------------------------
// The orbit loop of julia.wgsl:15-23,
//     dqs *= 4*qs;  q = q^2 + c;  qs = |q|^2;  if (qs > max_distance) break;
// hand-written for gfx950.  A frame's run time is the latency of its longest rays, and a
// lone wave pays ~5 cycles per instruction of any kind, so the loop is written to the
// minimum instruction count: 8 packed-f32 ops + 1 compare + 1 exec update per trip.
//
// Register pairs (fixed, so that 32-bit halves can be named):
//   v[40:41] YZ = [y, z]          v[42:43] WD = [w, dq]        v[44:45] Q = [|q|^2, x_next]
//   v[46:47] T  = [2x, 4|q_prev|^2]                            v[48:51] temporaries
// Per trip (HEAD then TAIL):
//   YZ = fma(T.lo, YZ, [cy,cz])              y,z of q_{k+1}
//   WD = fma(T, WD, [cw, 0])                 w of q_{k+1};  dq *= 4|q_{k-1}|^2  (one trip late)
//   T  = [Q.hi, Q.lo] * [2, 4]               2 x_{k+1}, 4|q_k|^2
//   A  = YZ*YZ; B = A.lo + A.hi (both lanes); A = fma(WD.lo, WD.lo, B) = d (both lanes)
//   Q  = fma(Q.hi, Q.hi, [d, -d]) + [0, cx]  = [|q_{k+1}|^2, x_{k+2}]
//   vcc = |q_{k+1}|^2 > max_distance;  exec &= ~vcc     (an escaped lane's registers freeze)
// x_{k+2} is computed one trip early from the squares that |q_{k+1}|^2 needs anyway.  The
// factor 4|q_k|^2 still missing from dq when a lane stops sits frozen in T.hi and is applied
// after the loop, so dq goes through exactly the products of the textbook loop, in order.
// The trip count is wave-uniform (SGPR); the loop is unrolled twice with one peeled trip
// for odd counts and leaves as soon as every lane has escaped.
#define KIFS_ORBIT_HEAD                                                                    \
    "v_pk_fma_f32 v[40:41], v[46:47], v[40:41], %[cyz] op_sel_hi:[0,1,1]\n"                  \
    "v_pk_fma_f32 v[42:43], v[46:47], v[42:43], %[cw0]\n"                                    \
    "v_pk_mul_f32 v[46:47], v[44:45], %[k24] op_sel:[1,0] op_sel_hi:[0,1]\n"
#define KIFS_ORBIT_TAIL                                                                    \
    "v_pk_mul_f32 v[48:49], v[40:41], v[40:41]\n"                                            \
    "v_pk_add_f32 v[50:51], v[48:49], v[48:49] op_sel:[0,1] op_sel_hi:[0,1]\n"               \
    "v_pk_fma_f32 v[48:49], v[42:43], v[42:43], v[50:51] op_sel_hi:[0,0,1]\n"                \
    "v_pk_fma_f32 v[44:45], v[44:45], v[44:45], v[48:49] op_sel:[1,1,0] op_sel_hi:[1,1,1] neg_hi:[0,0,1]\n" \
    "v_pk_add_f32 v[44:45], v[44:45], %[c0x]\n"
#define KIFS_ORBIT_TRIP                                                                    \
    KIFS_ORBIT_HEAD KIFS_ORBIT_TAIL                                                          \
    "v_cmp_lt_f32 vcc, %[maxd], v44\n"                                                       \
    "s_andn2_b64 exec, exec, vcc\n"

// Orbit + distance estimate for points inside the bounding sphere.  `lanes` is the exec mask
// of the lanes whose result is wanted (wave-uniform SGPR pair); the others keep garbage.
// The trip count is wave-uniform: the loop is unrolled six times (a taken branch costs a lone
// wave ~35 cycles, as much as half a trip) with a one-trip remainder loop in front, and leaves
// as soon as every lane has escaped.
KIFS_DEV float julia_interior(const FrameParams& P, V3 p, unsigned long long lanes) {
    F2 yz{p.y, p.z};
    F2 wd{0.1f, 1.0f};          // w = 0.1 (julia.wgsl:1), dq = 1 (:14)
    F2 q{1.0f, p.x};            // .hi = real part of q_0 (.lo is overwritten by the first TAIL)
    F2 t{p.x + p.x, 1.0f};      // 2 x_0, neutral first dq factor
    F2 ta, tb;
    const F2 cyz{P.c.y, P.c.z}, cw0{P.c.w, 0.0f}, c0x{0.0f, P.c.x}, k24{2.0f, 4.0f};
    int n = P.sdf_iters, rem;
    unsigned long long saved_exec;
    asm volatile(
        "s_mov_b64 %[save], exec\n"
        "s_and_b64 exec, exec, %[lanes]\n"
        KIFS_ORBIT_TAIL  // squares of q_0: Q = [|q_0|^2, x_1]
        // rem = n % 6 single trips, then n / 6 blocks of six
        "s_mul_hi_u32 %[rem], %[n], 0x2aaaaaab\n"   // n / 6 for 0 <= n < 2^31
        "s_mul_i32 %[rem], %[rem], 6\n"
        "s_sub_u32 %[rem], %[n], %[rem]\n"
        "s_sub_u32 %[n], %[n], %[rem]\n"
        "s_cmp_eq_u32 %[rem], 0\n"
        "s_cbranch_scc1 1f\n"
        "0:\n"
        KIFS_ORBIT_TRIP
        "s_sub_u32 %[rem], %[rem], 1\n"
        "s_cmp_lg_u32 %[rem], 0\n"
        "s_cbranch_scc1 0b\n"
        "1:\n"
        "s_cmp_eq_u32 %[n], 0\n"
        "s_cbranch_scc1 3f\n"
        "2:\n"
        KIFS_ORBIT_TRIP KIFS_ORBIT_TRIP KIFS_ORBIT_TRIP
        "s_cbranch_execz 3f\n"
        KIFS_ORBIT_TRIP KIFS_ORBIT_TRIP KIFS_ORBIT_TRIP
        "s_cbranch_execz 3f\n"
        "s_sub_u32 %[n], %[n], 6\n"
        "s_cmp_lg_u32 %[n], 0\n"
        "s_cbranch_scc1 2b\n"
        "3:\n"
        "s_mov_b64 exec, %[save]\n"
        : "+{v[40:41]}"(yz), "+{v[42:43]}"(wd), "+{v[44:45]}"(q), "+{v[46:47]}"(t),
          "=&{v[48:49]}"(ta), "=&{v[50:51]}"(tb), [save] "=&s"(saved_exec), [n] "+s"(n),
          [rem] "=&s"(rem)
        : [cyz] "s"(cyz), [cw0] "s"(cw0), [c0x] "s"(c0x), [k24] "s"(k24),
          [maxd] "s"(P.max_distance), [lanes] "s"(lanes)
        : "vcc", "scc");
    const float qs = q.x;
    const float dqs = wd.y * t.y;  // the factor the loop had not applied yet
    // zero / denormal / inf / NaN |q|^2 is rare: only then does the wave run the general log
    const bool ordinary = (qs >= 1.17549435e-38f) && (qs <= 3.40282347e38f);
    float lg;
    if (__builtin_expect((__builtin_amdgcn_ballot_w64(!ordinary) & lanes) == 0ull, 1))
        lg = log_normal(qs);
    else
        lg = log_general(qs);
    return (0.25f * lg) * sqrt_(qs / dqs);
}

// scene_SDF of julia.wgsl:5-27 for one point (point evaluation, normals of other variants).
KIFS_DEV float julia_sdf(const FrameParams& P, V3 p) {
    const float n2 = dot(p, p);
    const bool outside = n2 > P.bound_n2;  // == length(p) > 2 + epsilon (:7-10)
    const unsigned long long in_lanes = __builtin_amdgcn_ballot_w64(!outside);
    float d = sqrt_(n2) - 2.0f;
    if (in_lanes != 0ull) {
        float di = julia_interior(P, p, in_lanes);
        d = outside ? d : di;
    }
    return d;
}

KIFS_DEV V3 julia_normal(const FrameParams& P, V3 p) {
    V4 q{p.x, p.y, p.z, 0.1f};
    // Jacobian columns; A(q) = [x -y -z -w; y x 0 0; z 0 x 0; w 0 0 x]^T-as-columns
    // (julia.wgsl:40-45 uses the column-major constructor), J <- A*J column by column.
    V4 j0{1, 0, 0, 0}, j1{0, 1, 0, 0}, j2{0, 0, 1, 0}, j3{0, 0, 0, 1};
    auto apply = [&](V4 v) {
        V4 r;
        r.x = fmaf_(q.w, v.w, fmaf_(q.z, v.z, fmaf_(q.y, v.y, q.x * v.x)));
        r.y = fmaf_(q.x, v.y, (-q.y) * v.x);
        r.z = fmaf_(q.x, v.z, (-q.z) * v.x);
        r.w = fmaf_(q.x, v.w, (-q.w) * v.x);
        return r;
    };
    for (int i = 0; i < P.normal_iters; ++i) {
        j0 = apply(j0); j1 = apply(j1); j2 = apply(j2); j3 = apply(j3);
        q = quat_sq_add(q, P.c);
        if (quat_norm2(q) > P.max_distance) break;
    }
    V3 g;
    g.x = fmaf_(j3.x, q.w, fmaf_(j2.x, q.z, fmaf_(j1.x, q.y, j0.x * q.x)));
    g.y = fmaf_(j3.y, q.w, fmaf_(j2.y, q.z, fmaf_(j1.y, q.y, j0.y * q.x)));
    g.z = fmaf_(j3.z, q.w, fmaf_(j2.z, q.z, fmaf_(j1.z, q.y, j0.z * q.x)));
    return normalize(g);
}

// ---- generalised Julia -----------------------------------------------------------
// `lanes`: the lanes whose estimate the caller will use.  The others leave at once (their value is unspecified): a ray
// that has hit sits near the set, where orbits are longest, and stays in its wave until the round ends -- unmasked it
// made the wave run full-length orbits for a result nobody reads (the Julia loop and the Sierpinski folds mask theirs
// the same way).
KIFS_DEV float genjulia_sdf(const FrameParams& P, V3 p, unsigned long long lanes = ~0ull) {
    const float n2 = dot(p, p);
    if (n2 > P.bound_n2) return sqrt_(n2) - 2.0f;  // == length(p) > 2 + epsilon
    if ((lanes & (1ull << __lane_id())) == 0ull) return 0.0f;
    V4 q{p.x, p.y, p.z, 0.1f};
    float d = quat_ijk2(q);
    float qs = fmaf_(q.x, q.x, d);  // = quat_norm2(q)
    float dqs = 1.0f;
    const float pp = P.power * P.power;
    for (int i = 0; i < P.sdf_iters; ++i) {
        float pw1;  // pow(qs, power - 1)
        V4 t = quat_pow_shared(q, d, qs, P.power, pw1);
        dqs = dqs * (pp * pw1);
        q = V4{t.x + P.c.x, t.y + P.c.y, t.z + P.c.z, t.w + P.c.w};
        d = quat_ijk2(q);
        qs = fmaf_(q.x, q.x, d);
        if (qs > P.max_distance) break;
    }
    return (0.25f * log_(qs)) * sqrt_(qs / dqs);
}

KIFS_DEV V3 genjulia_normal(const FrameParams& P, V3 p) {
    const float h = P.epsilon;
    // six offset orbits (gen_julia.wgsl:35-40); `position +- h_axis` adds +-0 elsewhere
    V4 o[6] = {
        {p.x + h, p.y + 0.0f, p.z + 0.0f, 0.1f}, {p.x - h, p.y - 0.0f, p.z - 0.0f, 0.1f},
        {p.x + 0.0f, p.y + h, p.z + 0.0f, 0.1f}, {p.x - 0.0f, p.y - h, p.z - 0.0f, 0.1f},
        {p.x + 0.0f, p.y + 0.0f, p.z + h, 0.1f}, {p.x - 0.0f, p.y - 0.0f, p.z - h, 0.1f},
    };
    for (int i = 0; i < P.normal_iters; ++i) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            V4 t = quat_pow(o[k], P.power);
            o[k] = V4{t.x + P.c.x, t.y + P.c.y, t.z + P.c.z, t.w + P.c.w};
        }
    }
    float l[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) l[k] = log2_(length(o[k]));
    return normalize(V3{l[0] - l[1], l[2] - l[3], l[4] - l[5]});
}

// ---- KIFS ------------------------------------------------------------------------
// Mirror in a plane through the origin with normal e_a + e_b (kifs.wgsl:6-14 with the
// normals of :58-62): signed distance (pa+pb)/|n|, reflect only from the negative side.  The
// division by the constant |n| = sqrt(2) is a multiplication by fl(1/sqrt(2)): the arithmetic
// contract's choice, one ulp from the quotient, inside WGSL's bound for `/`.
KIFS_DEV void mirror2(float& pa, float& pb) {
    const float nn = 1.0f / sqrt_(2.0f);
    const float sd = (pa + pb) * nn;
    const float k = 2.0f * min_(sd, 0.0f);
    pa = fmaf_(-k, nn, pa);
    pb = fmaf_(-k, nn, pb);
}

KIFS_DEV V3 tetrahedral_fold(V3 p) {  // kifs.wgsl:56-66
    mirror2(p.x, p.y);
    mirror2(p.y, p.z);
    mirror2(p.x, p.z);
    return p;
}

// One mirror of the fold in the loop below: sd = (a + b) * fl(1/sqrt(2)); m = min(sd, 0) -- the
// hardware minimum, see below -- left in v46 for the caller's two updates  p -= m * (2/sqrt(2)).
#define KIFS_MIRROR_SD(a, b)                                                               \
    "v_add_f32_e32 v46, " a ", " b "\n"                                                    \
    "v_mul_f32_e32 v46, 0x3f3504f3, v46\n"                                                 \
    "v_min_f32_e32 v46, 0, v46\n"

// One fold + scale step of kifs.wgsl:72-78 for the lanes in EXEC, then EXEC &= n2 < stop.
//   v[40:41] = [x, y]   v42 = z   v43 = n2   v44 = scale   v[46:47] = [m, -]   s[76:77] = [-2/sqrt(2), -]
#define KIFS_FOLD_STEP                                                                     \
    KIFS_MIRROR_SD("v40", "v41")                                                           \
    "v_pk_fma_f32 v[40:41], v[46:47], s[76:77], v[40:41] op_sel_hi:[0,0,1]\n"              \
    KIFS_MIRROR_SD("v41", "v42")                                                           \
    "v_fmac_f32_e32 v41, 0xbfb504f3, v46\n"                                                \
    "v_fmac_f32_e32 v42, 0xbfb504f3, v46\n"                                                \
    KIFS_MIRROR_SD("v40", "v42")                                                           \
    "v_fmac_f32_e32 v40, 0xbfb504f3, v46\n"                                                \
    "v_fmac_f32_e32 v42, 0xbfb504f3, v46\n"                                                \
    "v_pk_fma_f32 v[40:41], v[40:41], 2.0, -1.0 op_sel_hi:[1,0,0]\n"   /* p = 2 p - 1 */    \
    "v_fma_f32 v42, v42, 2.0, -1.0\n"                                                      \
    "v_add_f32_e32 v44, v44, v44\n"                                    /* scale *= 2 */     \
    "v_mul_f32_e32 v43, v40, v40\n"                                    /* n2 = dot(p, p) */ \
    "v_fmac_f32_e32 v43, v41, v41\n"                                                       \
    "v_fmac_f32_e32 v43, v42, v42\n"                                                       \
    "v_cmpx_gt_f32_e32 vcc, %[stop], v43\n"

// The fold loop of the Sierpinski estimate, hand-scheduled: 20 instructions per fold (the
// compiler's version of the same arithmetic: 27, and a taken branch every fold; here every second).
//  * min_(sd, 0) is the hardware minimum: they differ only for sd = -0 (the mirror then adds a
//    zero of the other sign to both coordinates, and p = 2 p - 1, fma(2, +-0, -1) = -1, forgets the
//    sign of a zero before anything but another signed-zero sum can see it) and for NaN (a NaN
//    sum means a coordinate is NaN already and stays NaN on both paths: the estimate is NaN);
//  * 2 * m is exact, so fma(-(2 m), 1/sqrt(2), p) = fma(m, -(2/sqrt(2)), p) bit for bit.
// `lanes`: lanes that take part (the others keep their values).
KIFS_DEV void sierpinski_folds(const FrameParams& P, V3& p, float& n2, float& scale,
                               unsigned long long lanes) {
    F2 xy{p.x, p.y}, q;
    float z = p.z;
    const F2 knn{-2.0f * (1.0f / sqrt_(2.0f)), 0.0f};
    int n = P.fold_iters;
    unsigned long long saved_exec;
    asm volatile(
        "s_mov_b64 %[save], exec\n"
        "s_cmp_lt_i32 %[n], 1\n"
        "s_cbranch_scc1 1f\n"
        "s_and_b64 exec, exec, %[lanes]\n"   // lanes whose result nobody reads do not fold
        "v_cmpx_gt_f32_e32 vcc, %[stop], v43\n"
        "s_cbranch_execz 1f\n"
        ".p2align 6\n"   // the back edge's target on a 64-byte line (a lone wave refetches after a taken branch)
        "0:\n"
        KIFS_FOLD_STEP
        "s_sub_u32 %[n], %[n], 1\n"
        "s_cmp_eq_u32 %[n], 0\n"
        "s_cbranch_scc1 1f\n"
        "s_cbranch_execz 1f\n"
        KIFS_FOLD_STEP
        "s_sub_u32 %[n], %[n], 1\n"
        "s_cmp_eq_u32 %[n], 0\n"
        "s_cbranch_scc1 1f\n"
        "s_cbranch_execnz 0b\n"
        "1:\n"
        "s_mov_b64 exec, %[save]\n"
        : "+{v[40:41]}"(xy), "+{v42}"(z), "+{v43}"(n2), "+{v44}"(scale), "=&{v[46:47]}"(q),
          [save] "=&s"(saved_exec), [n] "+s"(n)
        : [stop] "s"(P.fold_n2_stop), "{s[76:77]}"(knn), [lanes] "s"(lanes)
        : "vcc", "scc");
    p = V3{xy.x, xy.y, z};
}

// `lanes`: the lanes whose estimate the caller will use (the others get an unspecified value).
KIFS_DEV float sierpinski_sdf(const FrameParams& P, V3 p, unsigned long long lanes) {  // kifs.wgsl:68-81
    // The loop condition `r < max_distance` is evaluated on the squared norm (exact: see
    // FrameParams::fold_n2_stop), so the square root is taken once, after the loop.
    float scale = 1.0f;
    float n2 = dot(p, p);
    sierpinski_folds(P, p, n2, scale, lanes);
    // scale is 2^k exactly (k folds made), so dividing by it is multiplying by 2^-k, itself exact: the
    // same single rounding (if any: only in the denormal range).  2^-k by an exponent flip while it is a
    // normal number (k <= 126: decided per launch); the division otherwise.
    if (__builtin_expect(P.fold_iters <= 126, 1)) return (sqrt_(n2) - 2.0f) * from_bits(0x7f000000u - bits(scale));
    return (sqrt_(n2) - 2.0f) / scale;
}

KIFS_DEV V4 mat4_vec(const float* m, V4 v) {  // column-major 4x4 times vector
    V4 r;
    r.x = fmaf_(m[12], v.w, fmaf_(m[8], v.z, fmaf_(m[4], v.y, m[0] * v.x)));
    r.y = fmaf_(m[13], v.w, fmaf_(m[9], v.z, fmaf_(m[5], v.y, m[1] * v.x)));
    r.z = fmaf_(m[14], v.w, fmaf_(m[10], v.z, fmaf_(m[6], v.y, m[2] * v.x)));
    r.w = fmaf_(m[15], v.w, fmaf_(m[11], v.z, fmaf_(m[7], v.y, m[3] * v.x)));
    return r;
}
KIFS_DEV V4 add4(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
KIFS_DEV V4 sin4(V4 a) { return V4{sin_(a.x), sin_(a.y), sin_(a.z), sin_(a.w)}; }
// the same values without branches (kifs_device_math.hpp), for the form whose dependent chain sets the frame time
KIFS_DEV V4 sin4_flat(V4 a) { return V4{sin_flat(a.x), sin_flat(a.y), sin_flat(a.z), sin_flat(a.w)}; }
KIFS_DEV V4 ld4(const float* v) { return V4{v[0], v[1], v[2], v[3]}; }

KIFS_DEV float bunny_sdf(V3 p) {  // kifs.wgsl:84-137; weights in constant memory
    if (dot(p, p) > 1.0f) return length(p) - 0.8f;
    V4 q{p.x * -1.0f, p.z * 1.0f, p.y * -1.0f, 1.0f};
    V4 f0[4], f1[4], f2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) f0[k] = sin4(mat4_vec(KIFS_BUNNY_L0[k], q));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        V4 a = mat4_vec(KIFS_BUNNY_L1[k][0], f0[0]);
        a = add4(a, mat4_vec(KIFS_BUNNY_L1[k][1], f0[1]));
        a = add4(a, mat4_vec(KIFS_BUNNY_L1[k][2], f0[2]));
        a = add4(a, mat4_vec(KIFS_BUNNY_L1[k][3], f0[3]));
        a = add4(a, ld4(KIFS_BUNNY_B1[k]));
        f1[k] = add4(sin4(a), f0[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        V4 a = mat4_vec(KIFS_BUNNY_L2[k][0], f1[0]);
        a = add4(a, mat4_vec(KIFS_BUNNY_L2[k][1], f1[1]));
        a = add4(a, mat4_vec(KIFS_BUNNY_L2[k][2], f1[2]));
        a = add4(a, mat4_vec(KIFS_BUNNY_L2[k][3], f1[3]));
        a = add4(a, ld4(KIFS_BUNNY_B2[k]));
        V4 sn = sin4(a);
        f2[k] = V4{sn.x / 1.4f + f1[k].x, sn.y / 1.4f + f1[k].y, sn.z / 1.4f + f1[k].z,
                   sn.w / 1.4f + f1[k].w};
    }
    float r = dot(f2[0], ld4(KIFS_BUNNY_OUT[0]));
#pragma unroll
    for (int k = 1; k < 4; ++k) r = r + dot(f2[k], ld4(KIFS_BUNNY_OUT[k]));
    return r - 0.16f;
}

template <int PRIM>
KIFS_DEV float kifs_sdf(const FrameParams& P, V3 p, unsigned long long lanes = ~0ull) {  // kifs.wgsl:139-155
    if constexpr (PRIM == PRIM_SPHERE) {
        return length(p) - 1.0f;
    } else if constexpr (PRIM == PRIM_CYLINDER) {
        V2 d{abs_(length(V2{p.x, p.y})) - 1.0f, abs_(p.z) - 2.0f};
        V2 dm{max_(d.x, 0.0f), max_(d.y, 0.0f)};
        return min_(max_(d.x, d.y), 0.0f) + length(dm);
    } else if constexpr (PRIM == PRIM_BOX) {
        V3 q{abs_(p.x) - 1.0f, abs_(p.y) - 1.0f, abs_(p.z) - 1.0f};
        V3 qm{max_(q.x, 0.0f), max_(q.y, 0.0f), max_(q.z, 0.0f)};
        return length(qm) + min_(max_(q.x, max_(q.y, q.z)), 0.0f);
    } else if constexpr (PRIM == PRIM_TORUS) {
        V2 q{length(V2{p.x, p.y}) - 1.0f, p.z};
        return length(q) - 0.3f;
    } else if constexpr (PRIM == PRIM_SIERPINSKI) {
        return sierpinski_sdf(P, p, lanes);
    } else if constexpr (PRIM == PRIM_BUNNY) {
        return bunny_sdf(p);
    } else {
        return 1.0f;
    }
}

// Central differences with h = epsilon, no division by 2h (kifs.wgsl:157-167), for any estimate.
template <class Sdf>
KIFS_DEV V3 normal_fd(float h, V3 p, Sdf sdf) {
    float dx = sdf(V3{p.x + h, p.y + 0.0f, p.z + 0.0f}) - sdf(V3{p.x - h, p.y - 0.0f, p.z - 0.0f});
    float dy = sdf(V3{p.x + 0.0f, p.y + h, p.z + 0.0f}) - sdf(V3{p.x - 0.0f, p.y - h, p.z - 0.0f});
    float dz = sdf(V3{p.x + 0.0f, p.y + 0.0f, p.z + h}) - sdf(V3{p.x - 0.0f, p.y - 0.0f, p.z - h});
    return normalize(V3{dx, dy, dz});
}

template <int PRIM>
KIFS_DEV V3 kifs_normal(const FrameParams& P, V3 p) {
    return normal_fd(P.epsilon, p, [&](V3 q) { return kifs_sdf<PRIM>(P, q); });
}

// `lanes`: mask of the lanes whose estimate is used; a scene may skip the others' work.
template <int GROUP, int PRIM>
KIFS_DEV float scene_sdf(const FrameParams& P, V3 p, unsigned long long lanes = ~0ull) {
    if constexpr (GROUP == GROUP_JULIA) return julia_sdf(P, p);
    else if constexpr (GROUP == GROUP_GENJULIA) return genjulia_sdf(P, p, lanes);
    else return kifs_sdf<PRIM>(P, p, lanes);
}

template <int GROUP, int PRIM>
KIFS_DEV V3 scene_normal(const FrameParams& P, V3 p) {
    if constexpr (GROUP == GROUP_JULIA) return julia_normal(P, p);
    else if constexpr (GROUP == GROUP_GENJULIA) return genjulia_normal(P, p);
    else return kifs_normal<PRIM>(P, p);
}

// ---- fs_main: pixel -> ray (entry.wgsl:49-59) ---------------------------------------
KIFS_DEV V3 ray_direction(const FrameParams& P, int x, int y) {
    float px = float(x) + 0.5f, py = float(y) + 0.5f;  // fragment centre, y = 0 at the top
    float uvx = (2.0f * px) / P.height - P.aspect;
    float uvy = (2.0f * py) / P.height - 1.0f;
    V3 d{(uvx * P.m1.x - uvy * P.m2.x) - P.m0.x, (uvx * P.m1.y - uvy * P.m2.y) - P.m0.y,
         (uvx * P.m1.z - uvy * P.m2.z) - P.m0.z};
    return normalize(d);
}

// ---- extension: soft shadows (not in the reference; see KifsExtensions) ------------------
// Secondary march from the hit point towards the light, operation for operation as
// specified in include/kifs_hip.h (KifsExtensions).  `lanes_hit` selects the lanes that take part; the loop leaves
// when none of them is still marching.
template <class Sdf>
KIFS_DEV float soft_shadow(const FrameParams& P, V3 p, V3 n, bool lanes_hit, Sdf sdf) {
    const V3 L = normalize(V3{1.0f, 1.0f, 1.0f});
    const float off = 2.0f * P.epsilon;
    const V3 start{fmaf_(off, n.x, p.x), fmaf_(off, n.y, p.y), fmaf_(off, n.z, p.z)};
    float res = 1.0f;
    float t = P.shadow_t0;
    bool marching = lanes_hit && (0 < P.shadow_steps);
    for (int j = 0; __builtin_amdgcn_ballot_w64(marching) != 0ull; ++j) {
        if (marching) {
            const V3 q{fmaf_(t, L.x, start.x), fmaf_(t, L.y, start.y), fmaf_(t, L.z, start.z)};
            const float h = sdf(q, ~0ull);
            if (h < P.epsilon) {
                res = 0.0f;
                marching = false;
            } else {
                res = min_(res, (P.shadow_k * h) / t);
                t = t + h;
                marching = !(t > P.shadow_max_t) && (j + 1 < P.shadow_steps);
            }
        }
    }
    return res;
}

// ---- the long-ray loop: every live lane inside the bounding sphere -------------------------
// The rays that decide a frame's run time skim the fractal for hundreds of steps, deep inside
// the bounding sphere, alone on their SIMD.  For that state -- no live lane outside the sphere,
// |q|^2 an ordinary positive number, not heatmap mode -- this routine runs whole march steps
// (entry.wgsl:12-25 with julia.wgsl:5-27 inlined) as one hand-scheduled instruction stream
// with two taken branches per step; anything else returns to the general loop, which redoes
// the step it was handed.  Arithmetic is operation-for-operation the C++ above: dot(p,p) as
// an fma chain, the packed orbit, log_normal, the correctly rounded divide and square root
// (the same v_div_scale/v_rcp/v_div_fmas/v_div_fixup and v_sqrt + one-ulp-fixup sequences
// hipcc emits, with their wait states), `d < epsilon`, t += d, p = fma(t, dir, origin).
//
// Registers: v[30:31] p.yz  v[32:33] [p.x, 1]  v[34:35] [t, dir.x]  v[36:37] dir.yz
//            v[38:39] [0.1, 1]  v[40:51] orbit (see julia_interior)  v52-v63 scratch, v59 = c1
//            s[84:85] caller exec  s[86:87] live lanes of the step  s[88:91] [2,4],[2,1]
//            s92 mantissa mask  s93/s94 class masks  s96/s97 trip counters  s[80:83] compare masks
// One orbit trip, two forms of the same operations (results bit for bit the same):
//  * PACKED: 8 v_pk_*_f32 + v_cmpx -- the fewest instructions, for a wave alone on its SIMD (a lone wave
//    issues one instruction per ~5 cycles whatever it is): the latency path.
//  * SCALAR: 13 plain f32 operations + v_cmpx -- the fewest VALU CYCLES, for SIMDs with several busy
//    waves, where the vector pipe is the limit: a wave64 v_fma_f32 occupies it for 2.25 cycles, a packed
//    one for 4.3 (tools/microbench/valu_rate.hip), so the packed trip costs 38 cycles and this one 33.
//    Same registers, the pairs of the packed form taken apart: v40 y  v41 z  v42 w  v43 dq  v44 |q|^2
//    v45 x_next  v46 2x  v47 4|q_prev|^2  v48-v50 scratch.
#define KIFS_FAST_TRIP_PACKED                                                               \
    "v_pk_fma_f32 v[40:41], v[46:47], v[40:41], %[cyz] op_sel_hi:[0,1,1]\n"                   \
    "v_pk_fma_f32 v[42:43], v[46:47], v[42:43], %[cw0]\n"                                     \
    "v_pk_mul_f32 v[46:47], v[44:45], s[88:89] op_sel:[1,0] op_sel_hi:[0,1]\n"                \
    "v_pk_mul_f32 v[48:49], v[40:41], v[40:41]\n"                                             \
    "v_pk_add_f32 v[50:51], v[48:49], v[48:49] op_sel:[0,1] op_sel_hi:[0,1]\n"                \
    "v_pk_fma_f32 v[48:49], v[42:43], v[42:43], v[50:51] op_sel_hi:[0,0,1]\n"                 \
    "v_pk_fma_f32 v[44:45], v[44:45], v[44:45], v[48:49] op_sel:[1,1,0] op_sel_hi:[1,1,1] neg_hi:[0,0,1]\n" \
    "v_pk_add_f32 v[44:45], v[44:45], %[c0x]\n"                                               \
    "v_cmpx_nlt_f32 vcc, %[maxd], v44\n" /* exec &= !(|q|^2 > max_distance): escaped lanes freeze */
#define KIFS_FAST_TRIP_SCALAR                                                               \
    "v_fma_f32 v40, v46, v40, %[cy]\n"        /* y' = fma(2x, y, c.y) */                      \
    "v_fma_f32 v41, v46, v41, %[cz]\n"                                                        \
    "v_fma_f32 v42, v46, v42, %[cw]\n"                                                        \
    "v_fma_f32 v43, v47, v43, 0\n"            /* dq = fma(4|q_prev|^2, dq, 0) (one trip late) */ \
    "v_mul_f32_e32 v46, 2.0, v45\n"           /* 2 x_next */                                  \
    "v_mul_f32_e32 v47, 4.0, v44\n"           /* 4 |q|^2 */                                   \
    "v_mul_f32_e32 v48, v40, v40\n"                                                           \
    "v_mul_f32_e32 v49, v41, v41\n"                                                           \
    "v_add_f32_e32 v50, v48, v49\n"           /* s = y*y + z*z */                             \
    "v_fma_f32 v48, v42, v42, v50\n"          /* d = fma(w, w, s) */                          \
    "v_fma_f32 v44, v45, v45, v48\n"          /* |q|^2 = fma(x, x, d) */                      \
    "v_fma_f32 v45, v45, v45, -v48\n"         /* fma(x, x, -d) */                             \
    "v_add_f32_e32 v45, %[cx], v45\n"         /* + c.x: the x after next */                   \
    "v_cmpx_nlt_f32 vcc, %[maxd], v44\n"
// the squares of q_0 in front of the first trip, same two forms
#define KIFS_JULIA_PROLOGUE_PACKED                                                          \
    "v_pk_mul_f32 v[46:47], v[32:33], s[90:91]\n"         /* T = [2 x_0, 1] */               \
    "v_mov_b64 v[40:41], v[30:31]\n"                      /* YZ = [y_0, z_0] */              \
    "v_mov_b64 v[42:43], v[38:39]\n"                      /* WD = [0.1, 1] */                \
    "v_pk_mul_f32 v[48:49], v[40:41], v[40:41]\n"         /* squares of q_0 -> Q = [|q_0|^2, x_1] */ \
    "v_pk_add_f32 v[50:51], v[48:49], v[48:49] op_sel:[0,1] op_sel_hi:[0,1]\n"               \
    "v_pk_fma_f32 v[48:49], v[42:43], v[42:43], v[50:51] op_sel_hi:[0,0,1]\n"                \
    "v_pk_fma_f32 v[44:45], v[32:33], v[32:33], v[48:49] op_sel_hi:[0,0,1] neg_hi:[0,0,1]\n" \
    "v_pk_add_f32 v[44:45], v[44:45], %[c0x]\n"
#define KIFS_JULIA_PROLOGUE_SCALAR                                                          \
    "v_mul_f32_e32 v46, 2.0, v32\n"                                                          \
    "v_mov_b32_e32 v47, 1.0\n"                                                               \
    "v_mov_b64 v[40:41], v[30:31]\n"                                                         \
    "v_mov_b64 v[42:43], v[38:39]\n"                                                         \
    "v_mul_f32_e32 v48, v40, v40\n"                                                          \
    "v_mul_f32_e32 v49, v41, v41\n"                                                          \
    "v_add_f32_e32 v50, v48, v49\n"                                                          \
    "v_fma_f32 v48, v42, v42, v50\n"                                                         \
    "v_fma_f32 v44, v32, v32, v48\n"                                                         \
    "v_fma_f32 v45, v32, v32, -v48\n"                                                        \
    "v_add_f32_e32 v45, %[cx], v45\n"

// The loop over blocks of six trips (s96 = blocks left; label 12 in, label 14 out), two forms:
//  * SINGLE: one block per pass -- the throughput kernel's (its SIMDs hold several busy waves: a taken branch is
//    hidden, the code it fetches is not: two blocks per pass cost the headline 0.4 %).
//  * PAIRED: two blocks per pass, an odd block left over out of line (labels 16 / 17) -- the latency kernels': a lone
//    wave pays for every taken branch, and with one block per pass the reference's 100 iterations took the back edge
//    sixteen times per march step (lone 1080p frame at the reference's constants 0.218 -> 0.210 ms, the headline
//    workload's 12 iterations 0.1557 -> 0.1546; profiles/r04/ab_paired_blocks.txt).
#define KIFS_ORBIT_BLOCK_(exit)                                                             \
    KIFS_FAST_TRIP exit KIFS_FAST_TRIP exit KIFS_FAST_TRIP
#define KIFS_ORBIT_LOOP_SINGLE                                                              \
    "12:\n"                                                                                 \
    "s_cmp_eq_u32 s96, 0\n"                                                                 \
    "s_cbranch_scc1 14f\n"                                                                  \
    "13:\n"                                                                                 \
    KIFS_ORBIT_BLOCK_(KIFS_TRIP_EXIT) "s_cbranch_execz 14f\n"                               \
    KIFS_ORBIT_BLOCK_(KIFS_TRIP_EXIT) "s_cbranch_execz 14f\n"                               \
    "s_sub_u32 s96, s96, 1\n"                                                               \
    "s_cmp_lg_u32 s96, 0\n"                                                                 \
    "s_cbranch_scc1 13b\n"
#define KIFS_ORBIT_LOOP_PAIRED                                                              \
    "12:\n"                                                                                 \
    "s_cmp_lt_u32 s96, 2\n"                                                                 \
    "s_cbranch_scc1 16f\n"                                                                  \
    "13:\n"                                                                                 \
    KIFS_ORBIT_BLOCK_(KIFS_TRIP_EXIT) "s_cbranch_execz 14f\n"                               \
    KIFS_ORBIT_BLOCK_(KIFS_TRIP_EXIT) "s_cbranch_execz 14f\n"                               \
    KIFS_ORBIT_BLOCK_(KIFS_TRIP_EXIT) "s_cbranch_execz 14f\n"                               \
    KIFS_ORBIT_BLOCK_(KIFS_TRIP_EXIT) "s_cbranch_execz 14f\n"                               \
    "s_sub_u32 s96, s96, 2\n"                                                               \
    "s_cmp_gt_u32 s96, 1\n"                                                                 \
    "s_cbranch_scc1 13b\n"                                                                  \
    "s_cmp_lg_u32 s96, 0\n"                                                                 \
    "s_cbranch_scc1 17f\n" /* one block left (out of line) */
#define KIFS_ORBIT_LOOP_PAIRED_OUT_OF_LINE                                                  \
    "16:\n"                                                                                 \
    "s_cmp_eq_u32 s96, 0\n"                                                                 \
    "s_cbranch_scc1 14b\n"                                                                  \
    "17:\n"                                                                                 \
    KIFS_ORBIT_BLOCK_(KIFS_TRIP_EXIT_BACK) "s_cbranch_execz 14b\n"                          \
    KIFS_ORBIT_BLOCK_(KIFS_TRIP_EXIT_BACK) "s_branch 14b\n"

// quot = |q|^2 / dqs and root = sqrt(quot), both correctly rounded, any operands: the sequences
// hipcc emits (v_div_scale / v_rcp / Newton / v_div_fmas / v_div_fixup; v_sqrt + one-ulp fixup
// with the 2^32 pre-scaling for tiny arguments), with their wait states.  v44 = |q|^2, v53 = dqs,
// v55 = lg in; v54 = 0.25 lg and v60 = root out.
#define KIFS_DIVSQRT_FULL \
    /* ---- quot = |q|^2 / dqs, correctly rounded */ \
    "v_div_scale_f32 v60, s[82:83], v53, v53, v44\n" \
    "v_div_scale_f32 v61, vcc, v44, v53, v44\n" \
    "v_rcp_f32_e32 v62, v60\n" \
    "v_mul_f32_e32 v54, 0x3e800000, v55\n"  /* 0.25 * lg */ \
    "v_fma_f32 v63, -v60, v62, 1.0\n" \
    "v_fmac_f32_e32 v62, v63, v62\n" \
    "v_mul_f32_e32 v63, v61, v62\n" \
    "v_fma_f32 v55, -v60, v63, v61\n" \
    "v_fmac_f32_e32 v63, v55, v62\n" \
    "v_fma_f32 v60, -v60, v63, v61\n" \
    "v_div_fmas_f32 v60, v60, v62, v63\n" \
    "v_div_fixup_f32 v60, v60, v53, v44\n" \
    /* ---- root = sqrt(quot), correctly rounded */ \
    "v_mul_f32_e32 v61, 0x4f800000, v60\n" \
    "v_cmp_gt_f32_e32 vcc, 0x0f800000, v60\n"  /* below 2^-96: work on quot * 2^32 */ \
    "s_nop 1\n" \
    "v_cndmask_b32_e32 v60, v60, v61, vcc\n" \
    "v_sqrt_f32_e32 v61, v60\n" \
    "s_nop 0\n" \
    "v_add_u32_e32 v62, -1, v61\n"  /* candidates one ulp either side */ \
    "v_add_u32_e32 v63, 1, v61\n" \
    "v_fma_f32 v52, -v62, v61, v60\n" \
    "v_fma_f32 v56, -v63, v61, v60\n" \
    "v_cmp_ge_f32_e64 s[80:81], 0, v52\n" \
    "v_cmp_lt_f32_e64 s[82:83], 0, v56\n" \
    "s_nop 0\n" \
    "v_cndmask_b32_e64 v62, v61, v62, s[80:81]\n" \
    "v_cndmask_b32_e64 v61, v62, v63, s[82:83]\n" \
    "v_mul_f32_e32 v62, 0x37800000, v61\n" \
    "v_cndmask_b32_e32 v61, v61, v62, vcc\n"  /* undo the 2^32 scaling */ \
    "v_cmp_class_f32_e64 vcc, v60, s94\n"  /* sqrt(+-0) = +-0, sqrt(inf) = inf */ \
    "s_nop 1\n" \
    "v_cndmask_b32_e32 v60, v61, v60, vcc\n"

// The same two results when both operands are ordinary -- |q|^2 and dqs within [2^-10, 2^84): the
// exponents differ by at most 94, so v_div_scale would scale nothing (it does from 96), v_div_fmas
// is a plain fma, v_div_fixup passes the quotient through, and the quotient (> 2^-95) needs
// neither the 2^32 pre-scaling (below 2^-96) nor the zero / infinity patch of the square root: the same Newton and fixup arithmetic, nine instructions and four wait
// states shorter.  A wave with a lane outside the range runs KIFS_DIVSQRT_FULL out of line.
// Used when sdf_iters is small: after ~25 iterations dqs of a bounded orbit leaves the range.
#define KIFS_DIVSQRT_ORDINARY \
    "v_subrev_u32_e32 v60, 0x3a800000, v44\n"   /* bits - bits(2^-10) */ \
    "v_subrev_u32_e32 v61, 0x3a800000, v53\n" \
    "v_max_u32_e32 v60, v60, v61\n" \
    "v_cmp_gt_u32_e32 vcc, 0x2f000000, v60\n"   /* both below 2^84 (and not negative / NaN) */ \
    "s_xor_b64 vcc, vcc, exec\n" \
    "s_cbranch_scc1 46f\n" \
    "v_rcp_f32_e32 v62, v53\n" \
    "v_mul_f32_e32 v54, 0x3e800000, v55\n"      /* 0.25 * lg */ \
    "v_fma_f32 v63, -v53, v62, 1.0\n" \
    "v_fmac_f32_e32 v62, v63, v62\n" \
    "v_mul_f32_e32 v63, v44, v62\n" \
    "v_fma_f32 v55, -v53, v63, v44\n" \
    "v_fmac_f32_e32 v63, v55, v62\n" \
    "v_fma_f32 v60, -v53, v63, v44\n" \
    "v_fma_f32 v60, v60, v62, v63\n" \
    "v_sqrt_f32_e32 v61, v60\n" \
    "s_nop 0\n" \
    "v_add_u32_e32 v62, -1, v61\n"               /* candidates one ulp either side */ \
    "v_add_u32_e32 v63, 1, v61\n" \
    "v_fma_f32 v52, -v62, v61, v60\n" \
    "v_fma_f32 v56, -v63, v61, v60\n" \
    "v_cmp_ge_f32_e64 s[80:81], 0, v52\n" \
    "v_cmp_lt_f32_e64 s[82:83], 0, v56\n" \
    "s_nop 0\n" \
    "v_cndmask_b32_e64 v62, v61, v62, s[80:81]\n" \
    "v_cndmask_b32_e64 v60, v62, v63, s[82:83]\n" \
    "47:\n"

// SHORT_DIVSQRT: the launcher's choice for frames with few SDF iterations (KIFS_DIVSQRT_ORDINARY);
// THROUGHPUT: the scalar form of the orbit trip (launches whose SIMDs hold several busy waves).
template <bool SHORT_DIVSQRT, bool THROUGHPUT>
KIFS_DEV void julia_fast_march(const FrameParams& P, V3 dir, float& t, V3& p, bool& hit,
                               bool& marching, int& trips, int& outside_steps, int limit) {
    F2 pyz{p.y, p.z}, px1{p.x, 1.0f}, tdx{t, dir.x};
    const F2 dyz{dir.y, dir.z}, w0{0.1f, 1.0f};
    const F2 oyz{P.origin.y, P.origin.z};
    const float c1 = -1.1514610310E-1f;  // second log coefficient, needed in a VGPR
    const unsigned long long lanes = __builtin_amdgcn_ballot_w64(marching);
    unsigned long long hit_mask = 0, live_out;
    if constexpr (THROUGHPUT) {
#define KIFS_FAST_TRIP KIFS_FAST_TRIP_SCALAR
#define KIFS_TRIP_EXIT "s_cbranch_execz 14f\n"  /* scalar slots are free where the vector pipe is the limit: test every trip */
#define KIFS_TRIP_EXIT_BACK "s_cbranch_execz 14b\n"  /* the same from the remainder trips, which sit BEHIND label 14 */
#define KIFS_JULIA_PROLOGUE KIFS_JULIA_PROLOGUE_SCALAR
#define KIFS_ORBIT_LOOP KIFS_ORBIT_LOOP_SINGLE
#define KIFS_ORBIT_LOOP_OUT_OF_LINE
#define KIFS_JULIA_C_OPERANDS [cy] "s"(P.c.y), [cz] "s"(P.c.z), [cw] "s"(P.c.w), [cx] "s"(P.c.x)
        if constexpr (SHORT_DIVSQRT) {
#define KIFS_JULIA_DIVSQRT KIFS_DIVSQRT_ORDINARY
#define KIFS_JULIA_DIVSQRT_OUT_OF_LINE "46:\n" KIFS_DIVSQRT_FULL "s_branch 47b\n"
#include "kifs_julia_march_asm.hpp"
#undef KIFS_JULIA_DIVSQRT
#undef KIFS_JULIA_DIVSQRT_OUT_OF_LINE
        } else {
#define KIFS_JULIA_DIVSQRT KIFS_DIVSQRT_FULL
#define KIFS_JULIA_DIVSQRT_OUT_OF_LINE
#include "kifs_julia_march_asm.hpp"
#undef KIFS_JULIA_DIVSQRT
#undef KIFS_JULIA_DIVSQRT_OUT_OF_LINE
        }
#undef KIFS_FAST_TRIP
#undef KIFS_TRIP_EXIT
#undef KIFS_TRIP_EXIT_BACK
#undef KIFS_JULIA_PROLOGUE
#undef KIFS_ORBIT_LOOP
#undef KIFS_ORBIT_LOOP_OUT_OF_LINE
#undef KIFS_JULIA_C_OPERANDS
    } else {
        const F2 cyz{P.c.y, P.c.z}, cw0{P.c.w, 0.0f}, c0x{0.0f, P.c.x};
#define KIFS_FAST_TRIP KIFS_FAST_TRIP_PACKED
#define KIFS_TRIP_EXIT  /* a lone wave pays for every instruction: test every third trip only */
#define KIFS_TRIP_EXIT_BACK
#define KIFS_JULIA_PROLOGUE KIFS_JULIA_PROLOGUE_PACKED
#define KIFS_ORBIT_LOOP KIFS_ORBIT_LOOP_PAIRED
#define KIFS_ORBIT_LOOP_OUT_OF_LINE KIFS_ORBIT_LOOP_PAIRED_OUT_OF_LINE
#define KIFS_JULIA_C_OPERANDS [cyz] "s"(cyz), [cw0] "s"(cw0), [c0x] "s"(c0x)
        if constexpr (SHORT_DIVSQRT) {
#define KIFS_JULIA_DIVSQRT KIFS_DIVSQRT_ORDINARY
#define KIFS_JULIA_DIVSQRT_OUT_OF_LINE "46:\n" KIFS_DIVSQRT_FULL "s_branch 47b\n"
#include "kifs_julia_march_asm.hpp"
#undef KIFS_JULIA_DIVSQRT
#undef KIFS_JULIA_DIVSQRT_OUT_OF_LINE
        } else {
#define KIFS_JULIA_DIVSQRT KIFS_DIVSQRT_FULL
#define KIFS_JULIA_DIVSQRT_OUT_OF_LINE
#include "kifs_julia_march_asm.hpp"
#undef KIFS_JULIA_DIVSQRT
#undef KIFS_JULIA_DIVSQRT_OUT_OF_LINE
        }
#undef KIFS_FAST_TRIP
#undef KIFS_TRIP_EXIT
#undef KIFS_TRIP_EXIT_BACK
#undef KIFS_JULIA_PROLOGUE
#undef KIFS_ORBIT_LOOP
#undef KIFS_ORBIT_LOOP_OUT_OF_LINE
#undef KIFS_JULIA_C_OPERANDS
    }
    const unsigned lane = __lane_id();
    p = V3{px1.x, pyz.x, pyz.y};
    t = tdx.x;
    hit = hit || (((hit_mask >> lane) & 1ull) != 0ull);
    marching = ((live_out >> lane) & 1ull) != 0ull;
}

// ---- raymarch for the Julia pipeline: control flow kept wave-uniform ----------------------
// Same results as the generic loop below.  Each trip classifies the wave with two ballots:
// every marching lane inside the bounding sphere (the state of all long rays: one straight
// run through julia_interior), every marching lane outside (background: a sqrt and a
// subtract), or mixed.  That keeps taken branches -- the expensive thing for a lone wave --
// to the loop's back edge.
// A ray that can never come within the cull radius of the origin (see fill_params): decided at
// ray set-up from the closest approach of the ray's line.
KIFS_DEV bool ray_never_inside(const FrameParams& P, V3 dir) {
    const float oo = dot(P.origin, P.origin);
    const float b = -dot(P.origin, dir);
    const float c2 = fmaf_(-b, b, oo);
    return (b <= 0.0f) ? (oo > P.cull_n2) : (c2 > P.cull_n2);
}

// Issue priority of a wave by the age of its rays (wave-uniform `trips`): see julia_loop.
KIFS_DEV void march_priority(int trips) {
    if (trips >= 96) __builtin_amdgcn_s_setprio(3);
    else if (trips >= 32) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(1);
}

struct JuliaDiag {  // diagnostics of one wave's march (SGPRs)
    int fast_steps = 0, fast_entries = 0, general_steps = 0;
    unsigned long long fast_ticks = 0;
};

// The march loop proper: steps the wave's marching lanes until none is left or `trips` reaches
// `limit` (max_iterations for a whole ray, the end of the current round when the workgroup
// re-queues its rays).  State in, state out; i_final is the heatmap's loop counter.
template <bool SHORT_DIVSQRT, bool THROUGHPUT>
KIFS_DEV void julia_loop(const FrameParams& P, V3 dir, float& t, V3& p, bool& hit, bool& marching,
                         int& trips, int& i_final, int limit, JuliaDiag& diag) {
    const bool fast_ok = (P.is_heatmap == 0u) && (P.sdf_iters >= 1);  // wave-uniform
    for (;;) {
        const unsigned long long live = __builtin_amdgcn_ballot_w64(marching);
        if (live == 0ull || trips >= limit) break;
        if (__builtin_expect(fast_ok, 1)) {
            // The hand-written loop takes whole steps for inside and outside lanes alike; it only
            // comes back when it meets a |q|^2 that needs the general log (or it is finished).
            const int before = trips;
            // (unconditional: two s_memtime per entry cost nothing, and the conditional form trips the
            // compiler's SGPR handling around the asm block)
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            int outside_steps = 0;  // (a fresh SGPR for the asm block: struct members confuse the allocator)
            // issue priority by the ray's age: after the culls only rays that reach the fractal get here, and the
            // longer a ray has marched the more likely it is the one its launch ends with
            // (levels 1 / 2 / 3 from 0 / 32 / 96 steps; same-box A/B against "3 for all", three runs each, r03: 48 frames per
            // launch 105.5 -> 106.4 Gpixel/s, 4096^2 x16 64.6 -> 65.3, nothing either way at 8 / 1 per launch; of the other
            // marks tried 64 / 16 and 48 / 16 were 0.7 % behind, and level 0 for the first round cost 5 % at 8 per launch)
            march_priority(trips);
            julia_fast_march<SHORT_DIVSQRT, THROUGHPUT>(P, dir, t, p, hit, marching, trips, outside_steps,
                                            limit < P.max_iterations ? limit : P.max_iterations);
            diag.general_steps += outside_steps;
            diag.fast_ticks += __builtin_amdgcn_s_memtime() - t0;
            ++diag.fast_entries;
            diag.fast_steps += trips - before;
            if (__builtin_amdgcn_ballot_w64(marching) == 0ull || trips >= limit) break;
        }
        const bool more = (trips + 1) < P.max_iterations;
        const float n2 = dot(p, p);
        const bool outside = n2 > P.bound_n2;  // == length(p) > 2 + epsilon
        const unsigned long long out_lanes = __builtin_amdgcn_ballot_w64(outside) & live;
        float d;
        if (__builtin_expect(out_lanes == 0ull, 1)) {
            d = julia_interior(P, p, live);
        } else {
            d = sqrt_(n2) - 2.0f;
            const unsigned long long in_lanes = live & ~out_lanes;
            if (in_lanes != 0ull) {
                float di = julia_interior(P, p, in_lanes);
                d = outside ? d : di;
            }
        }
        const bool h = marching && (d < P.epsilon);
        const bool go = marching && !h;
        hit = hit || h;
        if (__builtin_expect(P.is_heatmap != 0u, 0))  // i is only observable in heatmap mode
            i_final = h ? trips : (go ? trips + 1 : i_final);
        const float tn = t + d;
        t = go ? tn : t;
        p = V3{go ? fmaf_(tn, dir.x, P.origin.x) : p.x, go ? fmaf_(tn, dir.y, P.origin.y) : p.y,
               go ? fmaf_(tn, dir.z, P.origin.z) : p.z};
        marching = go && more && (tn < P.max_distance);
        ++trips;
        ++diag.general_steps;
    }
}

// Colour of a Julia hit at p (entry.wgsl:14-19 with julia.wgsl:29-56), soft shadows if enabled.
KIFS_DEV V3 julia_shade(const FrameParams& P, V3 p) {
    V3 n = julia_normal(P, p);
    float ndl = (n.x + n.y) + n.z;
    float lit = clamp_(ndl, 0.0f, 1.0f);
    if (__builtin_expect(P.soft_shadow != 0u, 0))
        lit = lit * soft_shadow(P, p, n, lit > 0.0f, [&](V3 q, unsigned long long) { return julia_sdf(P, q); });
    float diffuse = fmaf_(0.9f, lit, 0.1f);
    return V3{diffuse * P.fractal_color.x, diffuse * P.fractal_color.y, diffuse * P.fractal_color.z};
}

// A whole ray per lane, start to finish (heatmap mode, diagnostics, and the building block the
// re-queuing kernel path is checked against).
template <bool SHORT_DIVSQRT>
KIFS_DEV V3 raymarch_julia(const FrameParams& P, V3 dir, bool valid, int& steps) {
    float t = 0.0f;
    V3 p = P.origin;
    bool hit = false;
    int trips = 0;     // == the loop counter i of entry.wgsl:11 for every marching lane
    int i_final = 0;
    bool marching = valid && (0 < P.max_iterations) && (t < P.max_distance);
    // Bounding-sphere cull.  Outside the sphere of radius R = 2 + epsilon the estimate is
    // length(p) - 2 > epsilon (julia.wgsl:8-9), so a ray that never enters the sphere can never
    // satisfy `d < epsilon`: its pixel is background whatever else the loop does (only the
    // heatmap's step count would notice).  The test keeps a 10 % margin on R^2, orders of
    // magnitude above the rounding of p = fma(t, dir, origin) for any t < max_distance.
    if (P.is_heatmap == 0u && P.cull_n2 > 0.0f) marching = marching && !ray_never_inside(P, dir);
    JuliaDiag diag;
    const bool stamp = P.counters != nullptr;
    const unsigned long long wave_t0 = stamp ? __builtin_amdgcn_s_memtime() : 0ull;
    julia_loop<SHORT_DIVSQRT, false>(P, dir, t, p, hit, marching, trips, i_final, P.max_iterations, diag);
    __builtin_amdgcn_s_setprio(0);
    steps = trips;
    if (__builtin_expect(stamp, 0)) {
        // per-wave record (no atomics: they would serialise the waves being measured):
        // total ticks, ticks in the long-ray loop, long-ray steps | entries << 32, general steps
        const unsigned long long wave_t1 = __builtin_amdgcn_s_memtime();
        if (__lane_id() == 0) {
            unsigned long long* rec = P.counters + 8 + 4ull * (blockIdx.x * 4u + (threadIdx.x >> 6));
            rec[0] = wave_t1 - wave_t0;
            rec[1] = diag.fast_ticks;
            rec[2] = (unsigned long long)diag.fast_steps | ((unsigned long long)diag.fast_entries << 32);
            rec[3] = (unsigned long long)diag.general_steps;
        }
    }
    V3 colour = P.background_color;
    if (hit) colour = julia_shade(P, p);
    if (P.is_heatmap) {
        float f = float(i_final) / float(P.max_iterations);
        colour = V3{f * P.fractal_color.x, f * P.fractal_color.y, f * P.fractal_color.z};
    }
    return colour;
}

// `sdf(p, lanes)`: the scene's estimate (lanes = whose value is used); `normal(p)`: its normal.
// The march loop proper, as julia_loop: until no lane marches or `trips` reaches `limit`.
// AGE_PRIORITY: issue priority by the rays' age at entry and at the marks (march_priority), for the one-wave-per-tile
// throughput kernel, where a round starts here with its rays' step count (same-box A/B, r03: 1080p Sierpinski x48 80.2 ->
// 81.7 Gpixel/s, the lone 8K frame 1.181 -> 1.142 ms); elsewhere one step up after 32 steps, as before (the graded form
// cost the lone 1080p Sierpinski frame 3 % and 8 per launch 1 %).
template <bool AGE_PRIORITY = false, class Sdf>
KIFS_DEV void generic_loop(const FrameParams& P, V3 dir, float& t, V3& p, bool& hit, bool& marching,
                           int& trips, int& i_final, int limit, Sdf sdf) {
    // Bounding-sphere culls (see raymarch_julia and fill_params): every scene's estimate obeys
    // d(p) >= |p| - B, so a lane outside R = B + epsilon and moving away can never satisfy
    // `d < epsilon`.  Not in heatmap mode.
    const bool cull = (P.is_heatmap == 0u) && (P.cull_n2 > 0.0f);  // wave-uniform
    if constexpr (AGE_PRIORITY) march_priority(trips);
    while (__builtin_amdgcn_ballot_w64(marching) != 0ull && trips < limit) {
        if constexpr (AGE_PRIORITY) {
            if (trips == 32 || trips == 96) march_priority(trips);
        } else {
            // a ray still marching after 32 steps is on the frame's critical path: issue it first
            if (trips == 32) __builtin_amdgcn_s_setprio(3);
        }
        const bool more = (trips + 1) < P.max_iterations;  // scalar
        if (cull) {  // early ray termination for lanes that are leaving for good
            const bool leaving = (dot(p, p) > P.cull_n2) && (dot(p, dir) > 0.0f);
            marching = marching && !leaving;
            if (__builtin_amdgcn_ballot_w64(marching) == 0ull) break;
        }
        // The SDF is evaluated for every lane (stopped lanes hold a valid old position, their
        // result is discarded): no divergent region around the expensive part, the state update
        // is a handful of selects.
        const float d = sdf(p, __builtin_amdgcn_ballot_w64(marching));
        const bool h = marching && (d < P.epsilon);
        const bool go = marching && !h;
        hit = hit || h;  // break leaves i un-incremented (:20)
        if (__builtin_expect(P.is_heatmap != 0u, 0))
            i_final = h ? trips : (go ? trips + 1 : i_final);
        const float tn = t + d;
        t = go ? tn : t;
        p = V3{go ? fmaf_(tn, dir.x, P.origin.x) : p.x, go ? fmaf_(tn, dir.y, P.origin.y) : p.y,
               go ? fmaf_(tn, dir.z, P.origin.z) : p.z};
        marching = go && more && (tn < P.max_distance);
        ++trips;
    }
}

// Colour of a hit at p (entry.wgsl:14-19), soft shadows if enabled.
template <class Sdf, class Normal>
KIFS_DEV V3 generic_shade(const FrameParams& P, V3 p, Sdf sdf, Normal normal) {
    V3 n = normal(p);
    float ndl = (n.x + n.y) + n.z;  // dot(n, (1,1,1)): the light is not normalised (:17)
    float lit = clamp_(ndl, 0.0f, 1.0f);
    // (a lane whose direct term is not positive marches no secondary ray: its factor is 1 and 0 * 1 = 0)
    if (__builtin_expect(P.soft_shadow != 0u, 0)) lit = lit * soft_shadow(P, p, n, lit > 0.0f, sdf);
    float diffuse = fmaf_(0.9f, lit, 0.1f);
    return V3{diffuse * P.fractal_color.x, diffuse * P.fractal_color.y, diffuse * P.fractal_color.z};
}

// A whole ray per lane, start to finish.
template <class Sdf, class Normal>
KIFS_DEV V3 raymarch_with(const FrameParams& P, V3 dir, bool valid, int& steps, Sdf sdf, Normal normal) {
    float t = 0.0f;
    V3 p = P.origin;
    bool hit = false;
    // Every marching lane has made the same number of steps, so the loop counter `i` of
    // entry.wgsl:11 is the wave-uniform trip count (an SGPR); a lane records it when it stops.
    int trips = 0;
    int i_final = 0;
    bool marching = valid && (0 < P.max_iterations) && (t < P.max_distance);
    if ((P.is_heatmap == 0u) && (P.cull_n2 > 0.0f)) marching = marching && !ray_never_inside(P, dir);
    generic_loop(P, dir, t, p, hit, marching, trips, i_final, P.max_iterations, sdf);
    __builtin_amdgcn_s_setprio(0);
    steps = trips;
    V3 colour = P.background_color;
    if (hit) colour = generic_shade(P, p, sdf, normal);
    if (P.is_heatmap) {
        float f = float(i_final) / float(P.max_iterations);
        colour = V3{f * P.fractal_color.x, f * P.fractal_color.y, f * P.fractal_color.z};
    }
    return colour;
}

template <int GROUP, int PRIM>
KIFS_DEV V3 raymarch(const FrameParams& P, V3 dir, bool valid, int& steps) {
    // for the Julia pipeline the PRIM slot carries the long-ray loop's variant (launch_render)
    if constexpr (GROUP == GROUP_JULIA) return raymarch_julia<PRIM == 1>(P, dir, valid, steps);
    return raymarch_with(
        P, dir, valid, steps,
        [&](V3 q, unsigned long long lanes) { return scene_sdf<GROUP, PRIM>(P, q, lanes); },
        [&](V3 q) { return scene_normal<GROUP, PRIM>(P, q); });
}

// One round of the workgroup's ray queue for pipeline <GROUP, PRIM>: step the wave's lanes until
// `trips` reaches `limit`; and the colour of a hit.  (For the Julia pipeline the PRIM slot is the
// variant of the long-ray loop, see raymarch.)
template <int GROUP, int PRIM, bool THROUGHPUT = false>
KIFS_DEV void march_round(const FrameParams& P, V3 dir, float& t, V3& p, bool& hit, bool& marching,
                          int& trips, int limit) {
    int i_final = 0;  // heatmap frames do not take this path
    if constexpr (GROUP == GROUP_JULIA) {
        JuliaDiag diag;
        julia_loop<PRIM == 1, THROUGHPUT>(P, dir, t, p, hit, marching, trips, i_final, limit, diag);
    } else {
        generic_loop<THROUGHPUT>(P, dir, t, p, hit, marching, trips, i_final, limit,
                                 [&](V3 q, unsigned long long lanes) { return scene_sdf<GROUP, PRIM>(P, q, lanes); });
    }
}

template <int GROUP, int PRIM>
KIFS_DEV V3 shade_hit(const FrameParams& P, V3 p) {
    if constexpr (GROUP == GROUP_JULIA) return julia_shade(P, p);
    else
        return generic_shade(
            P, p, [&](V3 q, unsigned long long lanes) { return scene_sdf<GROUP, PRIM>(P, q, lanes); },
            [&](V3 q) { return scene_normal<GROUP, PRIM>(P, q); });
}

}  // namespace kifs
