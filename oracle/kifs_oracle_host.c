/*
 * kifs_oracle_host.c -- ORACLE restatement of the host half of the boundary
 * (test infrastructure; see kifs_oracle.h).  Follows, under /root/reference/src:
 *   data.rs:66-81     ScreenData::into_buffer_data
 *   data.rs:91-129    CameraData::camera_matrix / into_buffer_data
 *   data.rs:176-220   OptionsData (from GuiData) -> OptionsUniformData
 *   data/packed.rs:78-92, 116-139  matrix packing, LinearRgb::from_srgb (the /256 quirk)
 *   util/math.rs:326-367  Matrix3x3 * Matrix3x3, * Vector3 (row . column dot products)
 *   util/math.rs:386-416  rotation matrices, column-major
 *   util/math.rs:429-453  Radians::from_degrees / clamp / standardize
 *   render/graphics.rs:268-302  zoom_camera / rotate_camera
 * Rust's f32::cos/sin/powf are the platform libm ones; so are these.
 * These ARE pinned by the reference's own unit tests (tests/test_oracle_host_kats.py).
 */
#include "kifs_oracle.h"

#include <math.h>
#include <string.h>

#define PI_F 3.14159274101257324219f /* std::f32::consts::PI (math.rs:37) */
#define TWO_PI_F (2.0f * PI_F)       /* math.rs:38 */

void kor_screen_uniform(uint32_t width, uint32_t height, KorScreen* out) {
    float w = (float)width, h = (float)height; /* data.rs:71-73 */
    out->width = w;
    out->height = h;
    out->aspect_ratio = w / h; /* :78 */
}

/* m is column-major: m[3*c + r] */
void kor_rotation_matrix(int axis, float angle, float m[9]) {
    float c = cosf(angle), s = sinf(angle); /* Radians::cos_sin, math.rs:465-467 */
    if (axis == 0) {        /* math.rs:386-395 */
        float t[9] = {1, 0, 0, 0, c, s, 0, -s, c};
        memcpy(m, t, sizeof t);
    } else if (axis == 1) { /* math.rs:397-406 */
        float t[9] = {c, 0, -s, 0, 1, 0, s, 0, c};
        memcpy(m, t, sizeof t);
    } else {                /* math.rs:408-416 */
        float t[9] = {c, s, 0, -s, c, 0, 0, 0, 1};
        memcpy(m, t, sizeof t);
    }
}

/* Vector3 * Vector3 (dot) as the generic impl does it: x*x' + y*y' + z*z', left to right */
static float rdot(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

void kor_mat3_mul(const float a[9], const float b[9], float out[9]) { /* math.rs:326-353 */
    float r[9];
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++) {
            float row[3] = {a[rr], a[3 + rr], a[6 + rr]};
            r[3 * c + rr] = rdot(row, &b[3 * c]);
        }
    memcpy(out, r, sizeof r);
}

void kor_mat3_vec(const float a[9], const float v[3], float out[3]) { /* math.rs:355-367 */
    float r[3];
    for (int rr = 0; rr < 3; rr++) {
        float row[3] = {a[rr], a[3 + rr], a[6 + rr]};
        r[rr] = rdot(row, v);
    }
    memcpy(out, r, sizeof r);
}

void kor_camera_matrix(float phi, float theta, float m[9]) { /* data.rs:91-98 */
    float rz[9], ry[9];
    kor_rotation_matrix(2, phi, rz);
    kor_rotation_matrix(1, -theta, ry);
    kor_mat3_mul(rz, ry, m);
}

void kor_camera_uniform(float origin_distance, float phi, float theta, KorCamera* out) {
    float m[9], sm[9], o[3];
    const float ex[3] = {1.0f, 0.0f, 0.0f};
    kor_camera_matrix(phi, theta, m);
    /* `self.origin_distance * camera_matrix * Vector3(1,0,0)`: scalar*matrix first (data.rs:120) */
    for (int i = 0; i < 9; i++) sm[i] = origin_distance * m[i];
    kor_mat3_vec(sm, ex, o);
    memset(out, 0, sizeof *out);
    memcpy(out->origin, o, sizeof o);
    for (int c = 0; c < 3; c++) { /* packed.rs:78-92: each column extended with 0 */
        out->matrix[c][0] = m[3 * c + 0];
        out->matrix[c][1] = m[3 * c + 1];
        out->matrix[c][2] = m[3 * c + 2];
        out->matrix[c][3] = 0.0f;
    }
}

float kor_linear_from_srgb_u8(uint8_t v) { /* packed.rs:119-138 */
    float g = (float)v / 256.0f;           /* the reference divides by 256, not 255 */
    if (g <= 0.04045f) return g / 12.92f;
    return powf((g + 0.055f) / 1.055f, 2.4f);
}

void kor_options_from_gui(uint32_t max_iterations, float max_distance, float epsilon,
                          const uint8_t fractal_srgb[3], const uint8_t background_srgb[3],
                          int is_heatmap, uint32_t fractal_group, uint32_t primitive_shape,
                          float power, const float constant[4], KorOptions* out) {
    memset(out, 0, sizeof *out);
    out->max_iterations = (int32_t)max_iterations; /* data.rs:182 */
    out->max_distance = max_distance;
    out->epsilon = epsilon;
    for (int i = 0; i < 3; i++) {
        out->fractal_color[i] = kor_linear_from_srgb_u8(fractal_srgb[i]);
        out->background_color[i] = kor_linear_from_srgb_u8(background_srgb[i]);
    }
    out->is_heatmap = is_heatmap ? 1u : 0u;
    out->fractal_group_id = fractal_group;
    out->primitive_id = primitive_shape;
    out->power = power;
    memcpy(out->constant, constant, 4 * sizeof(float));
}

float kor_radians_from_degrees(float deg) { return (deg / 180.0f) * PI_F; } /* math.rs:429-431 */

float kor_radians_standardize(float rad) { /* math.rs:451-453; Rust % is fmod */
    return fmodf(fmodf(rad, TWO_PI_F) + TWO_PI_F, TWO_PI_F);
}

void kor_rotate_camera(float* phi, float* theta, float dphi, float dtheta) { /* graphics.rs:280-302 */
    float np = *phi + dphi, nt = *theta + dtheta;
    float lo = -PI_F / 2.0f, hi = PI_F / 2.0f;
    nt = (nt < lo) ? lo : (nt > hi ? hi : nt); /* f32::clamp */
    *phi = kor_radians_standardize(np);
    *theta = nt;
}

float kor_zoom_camera(float origin_distance, float min_distance, float delta) { /* graphics.rs:268-278 */
    float d = origin_distance - delta;
    return (min_distance > d) ? min_distance : d; /* f32::max */
}
