/*
 * kifs_oracle.c -- CPU ORACLE: plain-C restatement of the reference fragment
 * shader.  TEST INFRASTRUCTURE ONLY (see kifs_oracle.h for the contract and
 * the "parity unpinned" statement).  Build with -ffp-contract=off.
 *
 * Follows, function by function (paths under /root/reference/src/shaders):
 *   dependencies/entry.wgsl:6-29   raymarch
 *   dependencies/entry.wgsl:49-59  fs_main (pixel -> ray)
 *   dependencies/quaternions.wgsl  quat_* helpers
 *   julia.wgsl:5-56                Julia scene_SDF / get_normal
 *   gen_julia.wgsl:5-55            generalised Julia scene_SDF / get_normal
 *   kifs.wgsl:1-167                primitives, tetrahedral fold, Sierpinski, bunny, normal
 * and the implicit colour target (render.rs:72-80, render/graphics.rs:84-93):
 * clamp, linear->sRGB, UNORM8.
 *
 * Op-order conventions chosen where WGSL leaves them to the implementation:
 *   dot(a,b)      = fma(a.w,b.w, fma(a.z,b.z, fma(a.y,b.y, a.x*b.x)))  (prefix for vec3/vec2)
 *   length(v)     = sqrt(dot(v,v));   normalize(v) = v / length(v)  (IEEE divides)
 *   min(a,b)      = b < a ? b : a;    max(a,b) = a < b ? b : a;   clamp = min(max(e,lo),hi)
 *   a*b + c written in one WGSL expression inside a hot loop is fused where fma_() appears
 *   quat_sq_norm2(q) = fma(x,x, fma(w,w, y*y + z*z))   (see quat_norm2)
 *   M * v         = fma chain over the columns, column 0 first
 *   structurally-zero products of sparse constant matrices/vectors are dropped
 */
#include "kifs_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef struct { float x, y; } v2;
typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4; /* quaternion: (real, i, j, k) */

static inline float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
static inline float min_(float a, float b) { return (b < a) ? b : a; }
static inline float max_(float a, float b) { return (a < b) ? b : a; }
static inline float clamp_(float e, float lo, float hi) { return min_(max_(e, lo), hi); }
static inline float abs_(float a) { return fabsf(a); }

static inline float dot2(v2 a, v2 b) { return fma_(a.y, b.y, a.x * b.x); }
static inline float dot3(v3 a, v3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
static inline float dot4(v4 a, v4 b) {
    return fma_(a.w, b.w, fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)));
}
static inline float len2(v2 a) { return sqrtf(dot2(a, a)); }
static inline float len3(v3 a) { return sqrtf(dot3(a, a)); }
static inline float len4(v4 a) { return sqrtf(dot4(a, a)); }
static inline v3 normalize3(v3 a) {
    float l = len3(a);
    return (v3){a.x / l, a.y / l, a.z / l};
}

/* Per-frame scene: the three uniforms unpacked + iteration parameters. */
typedef struct {
    float width, height, aspect;
    v3 origin, m0, m1, m2; /* camera.matrix columns */
    int32_t max_iterations;
    float max_distance, epsilon;
    v3 fractal_color, background_color;
    uint32_t is_heatmap, group, primitive;
    float power;
    v4 c;
    int32_t sdf_iters, normal_iters, fold_iters;
    KorExt ext; /* extension block, all zero = reference behaviour */
    /* counters (instrumented run only) */
    uint64_t n_sdf, n_inner, n_inside; /* n_inside: SDF calls that got past the bounding-sphere early-out */
    uint64_t shade_inner, shade_inside; /* the share of get_normal (+ the extension's secondary ray) in the two above */
} Scene;

static void scene_init(Scene* s, const KorScreen* sc, const KorCamera* cam, const KorOptions* o,
                       const KorIters* it) {
    memset(s, 0, sizeof *s);
    if (sc) { s->width = sc->width; s->height = sc->height; s->aspect = sc->aspect_ratio; }
    if (cam) {
        s->origin = (v3){cam->origin[0], cam->origin[1], cam->origin[2]};
        s->m0 = (v3){cam->matrix[0][0], cam->matrix[0][1], cam->matrix[0][2]};
        s->m1 = (v3){cam->matrix[1][0], cam->matrix[1][1], cam->matrix[1][2]};
        s->m2 = (v3){cam->matrix[2][0], cam->matrix[2][1], cam->matrix[2][2]};
    }
    s->max_iterations = o->max_iterations;
    s->max_distance = o->max_distance;
    s->epsilon = o->epsilon;
    s->fractal_color = (v3){o->fractal_color[0], o->fractal_color[1], o->fractal_color[2]};
    s->background_color =
        (v3){o->background_color[0], o->background_color[1], o->background_color[2]};
    s->is_heatmap = o->is_heatmap;
    s->group = o->fractal_group_id;
    s->primitive = o->primitive_id;
    s->power = o->power;
    s->c = (v4){o->constant[0], o->constant[1], o->constant[2], o->constant[3]};
    s->sdf_iters = it ? it->sdf_iters : 100;
    s->normal_iters = it ? it->normal_iters : 10;
    s->fold_iters = it ? it->fold_iters : 10;
}

/* ======================= quaternions (quaternions.wgsl) ===================== */

/* The quaternion step is where a long ray spends its time, so its evaluation order is
 * fixed to what maps 1:1 onto packed-f32 (v_pk_*) instructions on gfx950; all of it is a
 * legal evaluation of quaternions.wgsl:22-28,42-50:
 *     s   = y*y + z*z              (two rounded products, one rounded sum)
 *     d   = fma(w, w, s)           = dot(ijk, ijk)
 *     |q|^2 = fma(x, x, d)         quat_sq_norm2
 *     q^2 + c: real = fma(x, x, -d) + c.x ;  ijk = fma(2x, ijk, c.ijk)               */
static inline float quat_ijk2(v4 q) { return fma_(q.w, q.w, q.y * q.y + q.z * q.z); }

static inline float quat_norm2(v4 q) { return fma_(q.x, q.x, quat_ijk2(q)); }

/* quat_sq, quaternions.wgsl:42-50 */
static inline v4 quat_sq(v4 q) {
    float tr = 2.0f * q.x;
    return (v4){fma_(q.x, q.x, -quat_ijk2(q)), tr * q.y, tr * q.z, tr * q.w};
}

/* quat_add(quat_sq(q), c) as one step, julia.wgsl:17 / :47 */
static inline v4 quat_sq_add(v4 q, v4 c) {
    float tr = 2.0f * q.x;
    return (v4){fma_(q.x, q.x, -quat_ijk2(q)) + c.x, fma_(tr, q.y, c.y), fma_(tr, q.z, c.z),
                fma_(tr, q.w, c.w)};
}

/* quat_mul, quaternions.wgsl:30-40 (unused by the shaders; kept for the identity test). */
static inline v4 quat_mul(v4 a, v4 b) {
    v3 ai = {a.y, a.z, a.w}, bi = {b.y, b.z, b.w};
    v3 cr = {fma_(ai.y, bi.z, -(ai.z * bi.y)), fma_(ai.z, bi.x, -(ai.x * bi.z)),
             fma_(ai.x, bi.y, -(ai.y * bi.x))};
    return (v4){fma_(a.x, b.x, -dot3(ai, bi)), fma_(a.x, bi.x, b.x * ai.x) + cr.x,
                fma_(a.x, bi.y, b.x * ai.y) + cr.y, fma_(a.x, bi.z, b.x * ai.z) + cr.z};
}

/* quat_pow, quaternions.wgsl:57-63:
 *     norm = length(q); phi = acos(q.real / norm); n = normalize(q.ijk);
 *     norm^x * (cos(x phi), n sin(x phi))
 * with the work it shares with its callers made explicit (contract change of round 2; every step is
 * a legal evaluation of the WGSL: builtins are ULP-bounded, not exact):
 *   d  = |ijk|^2 and qs = |q|^2 as in quat_norm2 -- the caller's loop needs them anyway;
 *   length(q) = sqrt(qs), length(ijk) = sqrt(d): the same sums of squares, one evaluation order;
 *   L  = log2(qs), taken once: pow(norm, x) = exp2(x log2(norm)) with log2(sqrt(qs)) = L / 2, and
 *        gen_julia.wgsl:16's pow(qs, power - 1) = exp2((power - 1) L) in the caller;
 *   a / norm and ijk / length(ijk) as products with the correctly rounded reciprocals (what a shader
 *        compiler makes of a division by a value used several times; within 1.5 ulp of the quotient).
 * One log2, two divisions and two dot products less per orbit step than the literal reading. */
static inline v4 quat_pow_shared(v4 q, float d, float qs, float L, float x) {
    float inv = 1.0f / sqrtf(qs);
    float phi = kor_acosf(q.x * inv);
    float ninv = 1.0f / sqrtf(d);
    v3 n = {q.y * ninv, q.z * ninv, q.w * ninv};
    float pw = kor_exp2f(x * (0.5f * L));
    float a = x * phi;
    float cs = kor_cosf(a), sn = kor_sinf(a);
    return (v4){pw * cs, pw * (n.x * sn), pw * (n.y * sn), pw * (n.z * sn)};
}

static inline v4 quat_pow(v4 q, float x) {
    float d = quat_ijk2(q);
    float qs = fma_(q.x, q.x, d);
    return quat_pow_shared(q, d, qs, kor_log2f(qs), x);
}

/* ============================ Julia (julia.wgsl) =========================== */

static float julia_sdf(Scene* s, v3 p) {
    s->n_sdf++;
    float norm = len3(p);                       /* julia.wgsl:7 */
    if (norm > 2.0f + s->epsilon) return norm - 2.0f; /* :8-10 */
    s->n_inside++;

    v4 q = {p.x, p.y, p.z, 0.1f};               /* :12, w = 0.1 (:1) */
    float qs = quat_norm2(q);                   /* :13 */
    float dqs = 1.0f;                           /* :14 */
    for (int i = 0; i < s->sdf_iters; i++) {    /* :15 */
        s->n_inner++;
        dqs = dqs * (4.0f * qs);                /* :16 */
        q = quat_sq_add(q, s->c);               /* :17 */
        qs = quat_norm2(q);                     /* :19 */
        if (qs > s->max_distance) break;        /* :20-22 */
    }
    return (0.25f * kor_logf(qs)) * sqrtf(qs / dqs); /* :26 */
}

static v3 julia_normal(Scene* s, v3 p) {
    v4 q = {p.x, p.y, p.z, 0.1f};               /* julia.wgsl:30-31 */
    /* J as 4 columns; identity (:33-38) */
    v4 J[4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int i = 0; i < s->normal_iters; i++) { /* :39 */
        /* A (column-major constructor, :40-45): col0=(x,-y,-z,-w) col1=(y,x,0,0)
         * col2=(z,0,x,0) col3=(w,0,0,x).  (A*J)[j] = A * J[j]. */
        for (int j = 0; j < 4; j++) {
            v4 v = J[j];
            v4 r;
            r.x = fma_(q.w, v.w, fma_(q.z, v.z, fma_(q.y, v.y, q.x * v.x)));
            r.y = fma_(q.x, v.y, (-q.y) * v.x);
            r.z = fma_(q.x, v.z, (-q.z) * v.x);
            r.w = fma_(q.x, v.w, (-q.w) * v.x);
            J[j] = r;
        }
        q = quat_sq_add(q, s->c);               /* :47-48 */
        if (quat_norm2(q) > s->max_distance) break; /* :50-52 */
    }
    /* (J * q_vec).xyz, :55 */
    v3 g;
    g.x = fma_(J[3].x, q.w, fma_(J[2].x, q.z, fma_(J[1].x, q.y, J[0].x * q.x)));
    g.y = fma_(J[3].y, q.w, fma_(J[2].y, q.z, fma_(J[1].y, q.y, J[0].y * q.x)));
    g.z = fma_(J[3].z, q.w, fma_(J[2].z, q.z, fma_(J[1].z, q.y, J[0].z * q.x)));
    return normalize3(g);
}

/* ===================== generalised Julia (gen_julia.wgsl) ================== */

static float genjulia_sdf(Scene* s, v3 p) {
    s->n_sdf++;
    float norm = len3(p);
    if (norm > 2.0f + s->epsilon) return norm - 2.0f; /* gen_julia.wgsl:7-10 */
    s->n_inside++;
    v4 q = {p.x, p.y, p.z, 0.1f};
    float d = quat_ijk2(q);
    float qs = fma_(q.x, q.x, d);               /* = quat_norm2(q) */
    float dqs = 1.0f;
    float pp = s->power * s->power;
    float pm1 = s->power - 1.0f;
    for (int i = 0; i < s->sdf_iters; i++) {
        s->n_inner++;
        float L = kor_log2f(qs);
        dqs = dqs * (pp * kor_exp2f(pm1 * L));  /* :16, pow(qs, power - 1) */
        v4 t = quat_pow_shared(q, d, qs, L, s->power); /* :17 */
        q = (v4){t.x + s->c.x, t.y + s->c.y, t.z + s->c.z, t.w + s->c.w};
        d = quat_ijk2(q);
        qs = fma_(q.x, q.x, d);
        if (qs > s->max_distance) break;
    }
    return (0.25f * kor_logf(qs)) * sqrtf(qs / dqs); /* :26 */
}

static v3 genjulia_normal(Scene* s, v3 p) {
    float h = s->epsilon;                        /* gen_julia.wgsl:31-33 */
    v4 qv[6] = {
        {p.x + h, p.y + 0.0f, p.z + 0.0f, 0.1f}, {p.x - h, p.y - 0.0f, p.z - 0.0f, 0.1f},
        {p.x + 0.0f, p.y + h, p.z + 0.0f, 0.1f}, {p.x - 0.0f, p.y - h, p.z - 0.0f, 0.1f},
        {p.x + 0.0f, p.y + 0.0f, p.z + h, 0.1f}, {p.x - 0.0f, p.y - 0.0f, p.z - h, 0.1f},
    };
    for (int i = 0; i < s->normal_iters; i++) {  /* :41-48 */
        for (int k = 0; k < 6; k++) {
            v4 t = quat_pow(qv[k], s->power);
            qv[k] = (v4){t.x + s->c.x, t.y + s->c.y, t.z + s->c.z, t.w + s->c.w};
        }
    }
    float l[6];
    for (int k = 0; k < 6; k++) l[k] = kor_log2f(len4(qv[k])); /* :51-53 */
    return normalize3((v3){l[0] - l[1], l[2] - l[3], l[4] - l[5]});
}

/* ============================== KIFS (kifs.wgsl) =========================== */

/* plane_mirror for a plane through the origin whose normal has two unit
 * components a,b and one zero (kifs.wgsl:6-14 with the normals of :58-62).
 *   plane_SDF    = dot(n, p - 0) / length(n) = (pa + pb) / sqrt(2)
 *   normalize(n) = n / length(n) -> 1/sqrt(2) on the two live axes, 0 on the third
 *   p - 2*min(sdf,0)*normalize(n): live axes fused, dead axis unchanged.
 * The division by the constant length(n) is evaluated as a multiplication by its correctly
 * rounded reciprocal nn = fl(1/sqrt(2)) -- what a shader compiler makes of `x / const`, within
 * one ulp of the quotient and so inside WGSL's 2.5 ULP bound for `/`.  (Contract change of
 * round 2: the correctly rounded quotient cost the kernels four extra instructions and a range
 * check per mirror, three mirrors per fold; the NumPy restatement keeps the true division and
 * tools/parity_envelope.py measures what the choice is worth in pixels.) */
static inline void mirror2(float* pa, float* pb) {
    const float nn = 1.0f / sqrtf(2.0f);
    float sd = (*pa + *pb) * nn;
    float k = 2.0f * min_(sd, 0.0f);
    *pa = fma_(-k, nn, *pa);
    *pb = fma_(-k, nn, *pb);
}

/* tetrahedral_fold, kifs.wgsl:56-66: normals (1,1,0) -> .zxy (0,1,1) -> (1,0,1) */
static inline v3 tetrahedral_fold(v3 p) {
    mirror2(&p.x, &p.y);
    mirror2(&p.y, &p.z);
    mirror2(&p.x, &p.z);
    return p;
}

static float sierpinski_sdf(Scene* s, v3 p) { /* kifs.wgsl:68-81 */
    float scale = 1.0f;
    float r = len3(p);
    for (int i = 0; i < s->fold_iters && r < s->max_distance; i++) {
        s->n_inner++;
        p = tetrahedral_fold(p);
        scale = scale * 2.0f;
        p = (v3){fma_(2.0f, p.x, -1.0f), fma_(2.0f, p.y, -1.0f), fma_(2.0f, p.z, -1.0f)};
        r = len3(p);
    }
    return (r - 2.0f) / scale;
}

/* Bunny network weights: numeric data of kifs.wgsl:90-136 (munrocket gist cited at :83),
 * stored as one flat table.  Layout: 4 first-layer 4x4 matrices, then for each of
 * the 8 hidden units 4 matrices + 1 bias, then 4 output vectors.  Column-major. */
#include "kifs_oracle_bunny.inc"

static inline v4 mat4_vec(const float* m, v4 v) { /* column-major 4x4 times v */
    v4 r;
    r.x = fma_(m[12], v.w, fma_(m[8], v.z, fma_(m[4], v.y, m[0] * v.x)));
    r.y = fma_(m[13], v.w, fma_(m[9], v.z, fma_(m[5], v.y, m[1] * v.x)));
    r.z = fma_(m[14], v.w, fma_(m[10], v.z, fma_(m[6], v.y, m[2] * v.x)));
    r.w = fma_(m[15], v.w, fma_(m[11], v.z, fma_(m[7], v.y, m[3] * v.x)));
    return r;
}
static inline v4 add4(v4 a, v4 b) { return (v4){a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
static inline v4 sin4(v4 a) {
    return (v4){kor_sinf(a.x), kor_sinf(a.y), kor_sinf(a.z), kor_sinf(a.w)};
}

static float bunny_sdf(v3 p) { /* kifs.wgsl:84-137 */
    if (dot3(p, p) > 1.0f) return len3(p) - 0.8f;         /* :85-87 */
    v4 q = {p.x * -1.0f, p.z * 1.0f, p.y * -1.0f, 1.0f};  /* :89 position.xzy * (-1,1,-1) */
    v4 f0[4], f1[4], f2[4];
    for (int k = 0; k < 4; k++) f0[k] = sin4(mat4_vec(KOR_BUNNY_L0[k], q)); /* :90-93 */
    for (int k = 0; k < 4; k++) {                         /* :94-113 */
        v4 a = mat4_vec(KOR_BUNNY_L1[k][0], f0[0]);
        a = add4(a, mat4_vec(KOR_BUNNY_L1[k][1], f0[1]));
        a = add4(a, mat4_vec(KOR_BUNNY_L1[k][2], f0[2]));
        a = add4(a, mat4_vec(KOR_BUNNY_L1[k][3], f0[3]));
        a = add4(a, (v4){KOR_BUNNY_B1[k][0], KOR_BUNNY_B1[k][1], KOR_BUNNY_B1[k][2],
                         KOR_BUNNY_B1[k][3]});
        f1[k] = add4(sin4(a), f0[k]);
    }
    for (int k = 0; k < 4; k++) {                         /* :114-133 */
        v4 a = mat4_vec(KOR_BUNNY_L2[k][0], f1[0]);
        a = add4(a, mat4_vec(KOR_BUNNY_L2[k][1], f1[1]));
        a = add4(a, mat4_vec(KOR_BUNNY_L2[k][2], f1[2]));
        a = add4(a, mat4_vec(KOR_BUNNY_L2[k][3], f1[3]));
        a = add4(a, (v4){KOR_BUNNY_B2[k][0], KOR_BUNNY_B2[k][1], KOR_BUNNY_B2[k][2],
                         KOR_BUNNY_B2[k][3]});
        v4 sn = sin4(a);
        /* `sin(...) / 1.4 + f1k` */
        f2[k] = (v4){sn.x / 1.4f + f1[k].x, sn.y / 1.4f + f1[k].y, sn.z / 1.4f + f1[k].z,
                     sn.w / 1.4f + f1[k].w};
    }
    /* :135-136 */
    float r = dot4(f2[0], (v4){KOR_BUNNY_OUT[0][0], KOR_BUNNY_OUT[0][1], KOR_BUNNY_OUT[0][2],
                               KOR_BUNNY_OUT[0][3]});
    for (int k = 1; k < 4; k++)
        r = r + dot4(f2[k], (v4){KOR_BUNNY_OUT[k][0], KOR_BUNNY_OUT[k][1], KOR_BUNNY_OUT[k][2],
                                 KOR_BUNNY_OUT[k][3]});
    return r - 0.16f;
}

static float kifs_sdf(Scene* s, v3 p) { /* kifs.wgsl:139-155 */
    s->n_sdf++;
    s->n_inside++;
    switch (s->primitive) {
    case 0: /* sphere r=1, :20-22 */
        return len3(p) - 1.0f;
    case 1: { /* cylinder(r=1,h=2), :29-32 */
        v2 d = {abs_(len2((v2){p.x, p.y})) - 1.0f, abs_(p.z) - 2.0f};
        v2 dm = {max_(d.x, 0.0f), max_(d.y, 0.0f)};
        return min_(max_(d.x, d.y), 0.0f) + len2(dm);
    }
    case 2: { /* box(1,1,1), :40-43 */
        v3 q = {abs_(p.x) - 1.0f, abs_(p.y) - 1.0f, abs_(p.z) - 1.0f};
        v3 qm = {max_(q.x, 0.0f), max_(q.y, 0.0f), max_(q.z, 0.0f)};
        return len3(qm) + min_(max_(q.x, max_(q.y, q.z)), 0.0f);
    }
    case 3: { /* torus(R=1, r=0.3), :50-53 */
        v2 q = {len2((v2){p.x, p.y}) - 1.0f, p.z};
        return len2(q) - 0.3f;
    }
    case 4:
        return sierpinski_sdf(s, p);
    case 5:
        return bunny_sdf(p);
    default:
        return 1.0f; /* :154 */
    }
}

static float scene_sdf(Scene* s, v3 p);

/* central differences, kifs.wgsl:157-167 (`position + h_x` adds 0 on the other axes) */
static v3 kifs_normal(Scene* s, v3 p) {
    float h = s->epsilon;
    float dx = kifs_sdf(s, (v3){p.x + h, p.y + 0.0f, p.z + 0.0f}) -
               kifs_sdf(s, (v3){p.x - h, p.y - 0.0f, p.z - 0.0f});
    float dy = kifs_sdf(s, (v3){p.x + 0.0f, p.y + h, p.z + 0.0f}) -
               kifs_sdf(s, (v3){p.x - 0.0f, p.y - h, p.z - 0.0f});
    float dz = kifs_sdf(s, (v3){p.x + 0.0f, p.y + 0.0f, p.z + h}) -
               kifs_sdf(s, (v3){p.x - 0.0f, p.y - 0.0f, p.z - h});
    return normalize3((v3){dx, dy, dz});
}

/* pipeline selection, graphics.rs:310-321 */
static float scene_sdf(Scene* s, v3 p) {
    switch (s->group) {
    case 1: return julia_sdf(s, p);
    case 2: return genjulia_sdf(s, p);
    default: return kifs_sdf(s, p);
    }
}
static v3 scene_normal(Scene* s, v3 p) {
    switch (s->group) {
    case 1: return julia_normal(s, p);
    case 2: return genjulia_normal(s, p);
    default: return kifs_normal(s, p);
    }
}

/* ======================== entry.wgsl: fs_main + raymarch =================== */

static v3 ray_direction(const Scene* s, int x, int y) {
    /* @builtin(position).xy = pixel centre, y = 0 at the top (entry.wgsl:51,54) */
    float px = (float)x + 0.5f, py = (float)y + 0.5f;
    float uvx = (2.0f * px) / s->height - s->aspect; /* :51 */
    float uvy = (2.0f * py) / s->height - 1.0f;
    v3 d = {(uvx * s->m1.x - uvy * s->m2.x) - s->m0.x, /* :55 */
            (uvx * s->m1.y - uvy * s->m2.y) - s->m0.y,
            (uvx * s->m1.z - uvy * s->m2.z) - s->m0.z};
    return normalize3(d);
}

/* EXTENSION (absent from the reference): soft shadow factor in [0, 1] for a hit at p with
 * normal n.  The secondary ray starts 2*epsilon off the surface along the normal and marches
 * towards the light direction normalize((1,1,1)) -- the direction of the reference's
 * un-normalised light vector (entry.wgsl:17):
 *     res = 1;  t = t0
 *     repeat up to shadow_steps times:
 *         h = scene_SDF(start + t * L);  if h < epsilon: return 0      (occluded)
 *         res = min(res, (k * h) / t);   t = t + h;  if t > max_t: stop
 *     return res
 * Only the direct term is attenuated: diffuse = 0.1 + 0.9 * clamp(n.(1,1,1)) * res -- and only where there is one:
 * a hit whose direct term is not positive (it faces away from the light) marches no secondary ray. */
static float soft_shadow(Scene* s, v3 p, v3 n) {
    const v3 L = normalize3((v3){1.0f, 1.0f, 1.0f});
    const float off = 2.0f * s->epsilon;
    const v3 start = {fma_(off, n.x, p.x), fma_(off, n.y, p.y), fma_(off, n.z, p.z)};
    float res = 1.0f;
    float t = s->ext.shadow_t0;
    for (int j = 0; j < s->ext.shadow_steps; j++) {
        v3 q = {fma_(t, L.x, start.x), fma_(t, L.y, start.y), fma_(t, L.z, start.z)};
        float h = scene_sdf(s, q);
        if (h < s->epsilon) return 0.0f;
        res = min_(res, (s->ext.shadow_k * h) / t);
        t = t + h;
        if (t > s->ext.shadow_max_t) break;
    }
    return res;
}

/* Returns loop counter i; rgba = linear colour (entry.wgsl:6-29). */
static int raymarch(Scene* s, v3 dir, float rgba[4], int* hit_out) {
    float out_r = s->background_color.x, out_g = s->background_color.y,
          out_b = s->background_color.z;
    float t = 0.0f;
    v3 p = s->origin;
    int i;
    int hit = 0;
    for (i = 0; i < s->max_iterations && t < s->max_distance; i++) { /* :12 */
        float d = scene_sdf(s, p);                                    /* :13 */
        if (d < s->epsilon) {                                         /* :15 */
            const uint64_t inner0 = s->n_inner, inside0 = s->n_inside; /* (instrumentation only) */
            v3 n = scene_normal(s, p);                                /* :16 */
            float ndl = (n.x + n.y) + n.z; /* dot(n, (1,1,1)), :17 */
            float lit = clamp_(ndl, 0.0f, 1.0f);
            if (s->ext.soft_shadow && lit > 0.0f) lit = lit * soft_shadow(s, p, n); /* extension */
            s->shade_inner += s->n_inner - inner0;
            s->shade_inside += s->n_inside - inside0;
            float diffuse = fma_(0.9f, lit, 0.1f);
            out_r = diffuse * s->fractal_color.x;                     /* :19 */
            out_g = diffuse * s->fractal_color.y;
            out_b = diffuse * s->fractal_color.z;
            hit = 1;
            break;                                                    /* :20 */
        }
        t = t + d;                                                    /* :23 */
        p = (v3){fma_(t, dir.x, s->origin.x), fma_(t, dir.y, s->origin.y),
                 fma_(t, dir.z, s->origin.z)};                        /* :24 */
    }
    if (s->is_heatmap) {                                              /* :27-28 */
        float f = (float)i / (float)s->max_iterations;
        out_r = f * s->fractal_color.x;
        out_g = f * s->fractal_color.y;
        out_b = f * s->fractal_color.z;
    }
    rgba[0] = out_r; rgba[1] = out_g; rgba[2] = out_b; rgba[3] = 1.0f;
    if (hit_out) *hit_out = hit;
    return i;
}

/* ====================== colour target: sRGB / UNORM8 ======================= */

static float g_srgb_t[256];
static pthread_once_t g_srgb_once = PTHREAD_ONCE_INIT;

static double srgb_oetf(double l) {
    return (l <= 0.0031308) ? 12.92 * l : 1.055 * pow(l, 1.0 / 2.4) - 0.055;
}
static double srgb_eotf(double v) {
    return (v <= 0.04045) ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4);
}

/* t[k] = smallest f32 x with 255*OETF(x) >= k - 0.5, i.e. the value at which
 * an ideal round-to-nearest sRGB UNORM8 conversion steps from k-1 to k. */
static void srgb_init(void) {
    g_srgb_t[0] = 0.0f;
    for (int k = 1; k < 256; k++) {
        double target = (double)k - 0.5;
        float f = (float)srgb_eotf(target / 255.0);
        while (srgb_oetf((double)f) * 255.0 < target) f = nextafterf(f, 2.0f);
        for (;;) {
            float g = nextafterf(f, -1.0f);
            if (srgb_oetf((double)g) * 255.0 >= target) f = g; else break;
        }
        g_srgb_t[k] = f;
    }
}

void kor_srgb_thresholds(float t[256]) {
    pthread_once(&g_srgb_once, srgb_init);
    memcpy(t, g_srgb_t, sizeof g_srgb_t);
}

uint8_t kor_encode_channel(float x, int encode) {
    if (encode == KOR_ENCODE_SRGB) {
        pthread_once(&g_srgb_once, srgb_init);
        /* code = number of thresholds <= x (NaN compares false -> 0); 8-step search */
        int k = 0;
        for (int step = 128; step >= 1; step >>= 1)
            if (x >= g_srgb_t[k + step]) k += step;
        return (uint8_t)k;
    }
    /* UNORM8: clamp, scale, +0.5, truncate (NaN -> 0) */
    float c = (x >= 0.0f) ? x : 0.0f; /* also maps NaN to 0 */
    c = (c > 1.0f) ? 1.0f : c;
    return (uint8_t)(int)(c * 255.0f + 0.5f);
}

static inline void encode_px(const float rgba[4], int encode, uint8_t* o) {
    o[0] = kor_encode_channel(rgba[0], encode);
    o[1] = kor_encode_channel(rgba[1], encode);
    o[2] = kor_encode_channel(rgba[2], encode);
    o[3] = kor_encode_channel(rgba[3], KOR_ENCODE_UNORM); /* alpha is never gamma-encoded */
}

/* ================================ frame ==================================== */

static int check_args(const KorScreen* sc, const KorCamera* cam, const KorOptions* o, int y0,
                      int y1, const uint8_t* out, size_t pitch) {
    if (!sc || !cam || !o || !out) return -1;
    int w = (int)sc->width, h = (int)sc->height;
    if (w <= 0 || h <= 0 || y0 < 0 || y1 > h || y0 > y1) return -1;
    if (pitch < (size_t)w * 4) return -1;
    return 0;
}

typedef struct {
    Scene scene;
    int encode, y0, y1, w;
    uint8_t* out;
    size_t pitch;
    volatile int* next_row;
} Job;

static void* worker(void* arg) {
    Job* j = (Job*)arg;
    Scene s = j->scene;
    for (;;) {
        int y = __sync_fetch_and_add(j->next_row, 1);
        if (y >= j->y1) break;
        uint8_t* row = j->out + (size_t)(y - j->y0) * j->pitch;
        for (int x = 0; x < j->w; x++) {
            float rgba[4];
            raymarch(&s, ray_direction(&s, x, y), rgba, NULL);
            encode_px(rgba, j->encode, row + 4 * (size_t)x);
        }
    }
    return NULL;
}

int kor_render(const KorScreen* screen, const KorCamera* camera, const KorOptions* options,
               const KorIters* iters, int encode, int y0, int y1, uint8_t* out, size_t pitch,
               int nthreads) {
    return kor_render_ext(screen, camera, options, iters, NULL, encode, y0, y1, out, pitch, nthreads);
}

int kor_render_ext(const KorScreen* screen, const KorCamera* camera, const KorOptions* options,
                   const KorIters* iters, const KorExt* ext, int encode, int y0, int y1,
                   uint8_t* out, size_t pitch, int nthreads) {
    if (check_args(screen, camera, options, y0, y1, out, pitch)) return -1;
    pthread_once(&g_srgb_once, srgb_init);
    if (nthreads <= 0) nthreads = (int)sysconf(_SC_NPROCESSORS_ONLN);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    Job job;
    scene_init(&job.scene, screen, camera, options, iters);
    if (ext) job.scene.ext = *ext;
    volatile int next = y0;
    job.encode = encode; job.y0 = y0; job.y1 = y1; job.w = (int)screen->width;
    job.out = out; job.pitch = pitch; job.next_row = &next;
    if (nthreads == 1) { worker(&job); return 0; }
    pthread_t th[256];
    int started = 0;
    for (int i = 0; i < nthreads; i++)
        if (pthread_create(&th[started], NULL, worker, &job) == 0) started++;
    if (started == 0) worker(&job);
    for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
    return 0;
}

int kor_render_stats(const KorScreen* screen, const KorCamera* camera, const KorOptions* options,
                     const KorIters* iters, int encode, int y0, int y1, uint8_t* out,
                     size_t pitch, KorStats* stats, uint16_t* steps_out) {
    if (check_args(screen, camera, options, y0, y1, out, pitch)) return -1;
    pthread_once(&g_srgb_once, srgb_init);
    Scene s;
    scene_init(&s, screen, camera, options, iters);
    KorStats st;
    memset(&st, 0, sizeof st);
    int w = (int)screen->width;
    for (int y = y0; y < y1; y++) {
        uint8_t* row = out + (size_t)(y - y0) * pitch;
        for (int x = 0; x < w; x++) {
            float rgba[4];
            int hit;
            uint64_t before = s.n_sdf;
            int i = raymarch(&s, ray_direction(&s, x, y), rgba, &hit);
            (void)before;
            encode_px(rgba, encode, row + 4 * (size_t)x);
            /* raymarch makes i SDF calls, +1 if it ended on a hit (break leaves i unincremented) */
            uint32_t steps = (uint32_t)i + (hit ? 1u : 0u);
            st.march_steps += steps;
            if (steps > st.max_steps) st.max_steps = steps;
            st.hits += (uint64_t)hit;
            st.pixels++;
            if (steps_out) steps_out[(size_t)(y - y0) * w + x] = (uint16_t)(i > 65535 ? 65535 : i);
        }
    }
    st.sdf_calls = s.n_sdf;
    st.inner_iters = s.n_inner;
    if (stats) *stats = st;
    return 0;
}

/* Per-ray work of the MARCH (instrumented; test and measurement tooling only): for every pixel of rows [y0, y1) the
 * number of scene_SDF calls its march made (`calls`: the loop counter, +1 when it ended on a hit), how many of them got
 * past the bounding-sphere early-out (`inside`: julia.wgsl:7-10; every call for the KIFS pipeline) and the inner
 * iterations they ran (`inner`: Julia iterations / Sierpinski folds).  What get_normal (and the extension's secondary
 * ray) adds after the march is not counted.  Arrays of (y1 - y0) * width entries; any of them may be NULL. */
int kor_render_ray_costs(const KorScreen* screen, const KorCamera* camera, const KorOptions* options,
                         const KorIters* iters, int y0, int y1, uint16_t* calls, uint16_t* inside, uint32_t* inner) {
    uint8_t px[4];
    if (check_args(screen, camera, options, y0, y1, px, (size_t)-1)) return -1; /* (no frame is written) */
    pthread_once(&g_srgb_once, srgb_init);
    Scene s;
    scene_init(&s, screen, camera, options, iters);
    int w = (int)screen->width;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < w; x++) {
            float rgba[4];
            int hit;
            const uint64_t in0 = s.n_inside - s.shade_inside, it0 = s.n_inner - s.shade_inner;
            int i = raymarch(&s, ray_direction(&s, x, y), rgba, &hit);
            const uint32_t steps = (uint32_t)i + (hit ? 1u : 0u);
            const uint64_t din = (s.n_inside - s.shade_inside) - in0, dit = (s.n_inner - s.shade_inner) - it0;
            const size_t k = (size_t)(y - y0) * (size_t)w + (size_t)x;
            if (calls) calls[k] = (uint16_t)(steps > 65535u ? 65535u : steps);
            if (inside) inside[k] = (uint16_t)(din > 65535u ? 65535u : din);
            if (inner) inner[k] = (uint32_t)dit;
        }
    return 0;
}

/* The march of ONE pixel, step by step (tooling for lane-occupancy studies): trips[i] = inner iterations of the i-th
 * scene_SDF call of the march (0: the call took the bounding-sphere early-out), for the first `max_steps` calls.
 * Returns the number of calls the march made (the loop counter, +1 on a hit). */
int kor_march_trace(const KorScreen* screen, const KorCamera* camera, const KorOptions* options, const KorIters* iters,
                    int x, int y, uint8_t* trips, int max_steps) {
    Scene s;
    scene_init(&s, screen, camera, options, iters);
    v3 dir = ray_direction(&s, x, y);
    float t = 0.0f;
    v3 p = s.origin;
    int i, n = 0;
    for (i = 0; i < s.max_iterations && t < s.max_distance; i++) { /* entry.wgsl:12-25, shading left out */
        const uint64_t before = s.n_inner;
        float d = scene_sdf(&s, p);
        const uint64_t k = s.n_inner - before;
        if (n < max_steps) trips[n] = (uint8_t)(k > 255 ? 255 : k);
        n++;
        if (d < s.epsilon) break;
        t = t + d;
        p = (v3){fma_(t, dir.x, s.origin.x), fma_(t, dir.y, s.origin.y), fma_(t, dir.z, s.origin.z)};
    }
    return n;
}

int kor_shade_pixel(const KorScreen* screen, const KorCamera* camera, const KorOptions* options,
                    const KorIters* iters, int x, int y, float rgba[4]) {
    Scene s;
    scene_init(&s, screen, camera, options, iters);
    return raymarch(&s, ray_direction(&s, x, y), rgba, NULL);
}

/* ============================ pieces for KATs ============================== */

float kor_scene_sdf(const KorOptions* options, const KorIters* iters, const float p[3]) {
    Scene s;
    scene_init(&s, NULL, NULL, options, iters);
    return scene_sdf(&s, (v3){p[0], p[1], p[2]});
}

void kor_get_normal(const KorOptions* options, const KorIters* iters, const float p[3],
                    float n[3]) {
    Scene s;
    scene_init(&s, NULL, NULL, options, iters);
    v3 r = scene_normal(&s, (v3){p[0], p[1], p[2]});
    n[0] = r.x; n[1] = r.y; n[2] = r.z;
}

void kor_ray_direction(const KorScreen* screen, const KorCamera* camera, int x, int y,
                       float dir[3]) {
    KorOptions o;
    memset(&o, 0, sizeof o);
    Scene s;
    scene_init(&s, screen, camera, &o, NULL);
    v3 d = ray_direction(&s, x, y);
    dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}

void kor_quat_sq(const float q[4], float out[4]) {
    v4 r = quat_sq((v4){q[0], q[1], q[2], q[3]});
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

void kor_quat_mul(const float a[4], const float b[4], float out[4]) {
    v4 r = quat_mul((v4){a[0], a[1], a[2], a[3]}, (v4){b[0], b[1], b[2], b[3]});
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

void kor_tetrahedral_fold(const float p[3], float out[3]) {
    v3 r = tetrahedral_fold((v3){p[0], p[1], p[2]});
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
