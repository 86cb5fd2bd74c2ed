/*
 * kifs_oracle.h -- CPU ORACLE for the kifs-raymarching hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's WGSL fragment shader
 * (fs_main -> raymarch -> scene_SDF / get_normal) plus the host-side uniform
 * packing, used only as the checker in tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  Nothing under kifs_raymarching_amd/ may
 * include, link or call it.
 *
 * PARITY STATUS: "parity unpinned" for shader pixels.  The reference
 * (Rust + WGSL behind wgpu/naga) cannot be built or run in this environment
 * and its own tests pin no shader output.  What IS
 * pinned: the host half (camera matrix, rotation matrices, mat*vec, packing)
 * against the reference's own unit tests, restated as known-answer tests in
 * tests/test_oracle_host_kats.py, plus analytic known answers for the SDFs.
 *
 * Arithmetic contract ("KIFS-f32"): every operation is an individually
 * rounded IEEE-754 binary32 operation in the order written in the .c file;
 * fused multiply-adds appear only where fma_() is written; sqrt and divide
 * are correctly rounded; log is the f32-only polynomial kor_logf().  With
 * -ffp-contract=off this makes the result a pure function of the inputs, so a
 * GPU kernel that performs the same operation sequence matches bit for bit.
 *
 * All citations are file:line under /root/reference/src.
 */
#ifndef KIFS_ORACLE_H
#define KIFS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Uniform images, byte-identical to data.rs:17-49 / shaders/dependencies/bindings.wgsl:1-35. */
typedef struct {
    float width, height, aspect_ratio;
} KorScreen; /* 12 B */

typedef struct {
    float origin[3];
    uint32_t _padding;
    float matrix[3][4]; /* 3 columns, each padded to 16 B (packed.rs:78-92) */
} KorCamera; /* 64 B */

typedef struct {
    int32_t max_iterations;
    float max_distance;
    float epsilon;
    uint32_t _padding1;
    float fractal_color[3];
    uint32_t _padding2;
    float background_color[3];
    uint32_t is_heatmap;
    uint32_t fractal_group_id;
    uint32_t primitive_id;
    float power;
    uint32_t _padding3;
    float constant[4]; /* (real, i, j, k) */
} KorOptions; /* 80 B */

/* Iteration counts the reference hard-codes (julia.wgsl:2-3, kifs.wgsl:72). */
typedef struct {
    int32_t sdf_iters;    /* JULIA_ITERATIONS = 100 */
    int32_t normal_iters; /* JULIA_NORMAL_ITERATIONS = 10 */
    int32_t fold_iters;   /* Sierpinski loop bound = 10 */
} KorIters;

enum { KOR_ENCODE_UNORM = 0, KOR_ENCODE_SRGB = 1 };

/* Extension (NOT in the reference; BASELINE config 5 names it): soft shadows by a secondary
 * march from the hit point towards the light (1,1,1).  Semantics are this project's own,
 * stated in kifs_oracle.c:soft_shadow and mirrored by the HIP kernels. */
typedef struct {
    uint32_t soft_shadow; /* 0 = off (the reference's shading) */
    int32_t shadow_steps; /* cap on secondary march steps */
    float shadow_k;       /* penumbra sharpness: res = min(res, k * h / t) */
    float shadow_t0;      /* first sample distance along the light ray */
    float shadow_max_t;   /* stop once the secondary ray has travelled this far */
} KorExt;

/* Per-frame work counters (instrumented run; used for the flops/pixel figure). */
typedef struct {
    uint64_t pixels;
    uint64_t march_steps;   /* scene_SDF calls made by raymarch */
    uint64_t inner_iters;   /* Julia iterations / Sierpinski folds executed, all SDF calls */
    uint64_t hits;          /* pixels that ran get_normal */
    uint64_t sdf_calls;     /* all scene_SDF calls incl. those from get_normal */
    uint32_t max_steps;     /* longest march */
} KorStats;

/* ---- frame ---------------------------------------------------------------- */
/* Renders rows [y0, y1) of the W x H frame described by `screen` into `out`
 * (row y at out + (y - y0) * pitch, 4 bytes per pixel RGBA8).  nthreads <= 0
 * means one thread per online core.  Returns 0, or -1 on bad arguments. */
int kor_render(const KorScreen* screen, const KorCamera* camera, const KorOptions* options,
               const KorIters* iters, int encode, int y0, int y1, uint8_t* out, size_t pitch,
               int nthreads);

/* kor_render with the extension block (ext == NULL: identical to kor_render). */
int kor_render_ext(const KorScreen* screen, const KorCamera* camera, const KorOptions* options,
                   const KorIters* iters, const KorExt* ext, int encode, int y0, int y1,
                   uint8_t* out, size_t pitch, int nthreads);

/* Same, single-threaded, also filling `stats`; `steps_out` (optional, W*(y1-y0)
 * uint16) receives the loop counter `i` of every pixel (entry.wgsl:11-27). */
int kor_render_stats(const KorScreen* screen, const KorCamera* camera, const KorOptions* options,
                     const KorIters* iters, int encode, int y0, int y1, uint8_t* out,
                     size_t pitch, KorStats* stats, uint16_t* steps_out);

/* Per-ray work of the march (instrumented): scene_SDF calls, those past the bounding-sphere early-out, inner iterations. */
int kor_render_ray_costs(const KorScreen* screen, const KorCamera* camera, const KorOptions* options,
                         const KorIters* iters, int y0, int y1, uint16_t* calls, uint16_t* inside, uint32_t* inner);

/* The march of one pixel step by step: inner iterations of every scene_SDF call (0 = bounding-sphere early-out). */
int kor_march_trace(const KorScreen* screen, const KorCamera* camera, const KorOptions* options, const KorIters* iters,
                    int x, int y, uint8_t* trips, int max_steps);

/* Linear (pre-encode) colour of one pixel; rgba[4] f32.  Returns loop counter i. */
int kor_shade_pixel(const KorScreen* screen, const KorCamera* camera, const KorOptions* options,
                    const KorIters* iters, int x, int y, float rgba[4]);

/* ---- pieces, for known-answer tests --------------------------------------- */
float kor_scene_sdf(const KorOptions* options, const KorIters* iters, const float p[3]);
void kor_get_normal(const KorOptions* options, const KorIters* iters, const float p[3],
                    float n[3]);
void kor_ray_direction(const KorScreen* screen, const KorCamera* camera, int x, int y,
                       float dir[3]);
void kor_quat_sq(const float q[4], float out[4]);
void kor_quat_mul(const float a[4], const float b[4], float out[4]);
void kor_tetrahedral_fold(const float p[3], float out[3]);

float kor_logf(float x);
float kor_sinf(float x);
float kor_cosf(float x);
float kor_acosf(float x);
float kor_exp2f(float x);
float kor_log2f(float x);
float kor_powf(float x, float y);

void kor_srgb_thresholds(float t[256]); /* t[0] = 0; code = #{k>=1 : x >= t[k]} */
uint8_t kor_encode_channel(float x, int encode);

/* ---- host half of the boundary (data.rs, packed.rs, math.rs, graphics.rs) -- */
void kor_screen_uniform(uint32_t width, uint32_t height, KorScreen* out);
void kor_camera_matrix(float phi, float theta, float m[9] /* column-major */);
void kor_camera_uniform(float origin_distance, float phi, float theta, KorCamera* out);
float kor_linear_from_srgb_u8(uint8_t v);
void kor_options_from_gui(uint32_t max_iterations, float max_distance, float epsilon,
                          const uint8_t fractal_srgb[3], const uint8_t background_srgb[3],
                          int is_heatmap, uint32_t fractal_group, uint32_t primitive_shape,
                          float power, const float constant[4], KorOptions* out);
void kor_rotation_matrix(int axis /*0=x,1=y,2=z*/, float angle, float m[9]);
void kor_mat3_mul(const float a[9], const float b[9], float out[9]);
void kor_mat3_vec(const float a[9], const float v[3], float out[3]);
float kor_radians_from_degrees(float deg);
float kor_radians_standardize(float rad);
void kor_rotate_camera(float* phi, float* theta, float dphi, float dtheta);
float kor_zoom_camera(float origin_distance, float min_distance, float delta);

#ifdef __cplusplus
}
#endif
#endif
