// kifs_host.cpp -- the caller side of the boundary: scene description -> the 156 bytes
// of uniforms, restated in C++ because the reference's Rust host cannot be built here.
// Mirrors (paths under src/ of the reference):
//   ScreenData / CameraData / GuiData / OptionsData and into_buffer_data   data.rs:51-220
//   LinearRgb::from_srgb (divides by 256, not 255)                          data/packed.rs:116-139
//   Matrix3x3 column-major algebra, rotation matrices, Radians               util/math.rs:277-469
//   GraphicState::zoom_camera / rotate_camera                                render/graphics.rs:268-302
//   mouse -> camera deltas                                                    render.rs:239-270
// Rust's f32::sin/cos/powf are the platform libm functions; <cmath> float overloads
// are the same calls.
#include <cmath>
#include <cstring>

#include "../../include/kifs_hip.h"

namespace {

constexpr float PI = 3.14159274101257324219f;  // std::f32::consts::PI, math.rs:37
constexpr float TWO_PI = 2.0f * PI;            // math.rs:38

struct Vec3 {
    float x, y, z;
};

// `Vector3 * Vector3` is the dot product, accumulated left to right (math.rs macro
// impl_vector_dot_product)
inline float operator*(const Vec3& a, const Vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

class Radians {  // math.rs:419-469
public:
    explicit Radians(float r) : r_(r) {}
    static Radians from_degrees(float d) { return Radians((d / 180.0f) * PI); }
    float radians() const { return r_; }
    Radians clamp(float lo, float hi) const {  // f32::clamp
        float v = r_;
        if (v < lo) v = lo;
        if (v > hi) v = hi;
        return Radians(v);
    }
    Radians standardize() const {  // Rust `%` on f32 is fmod
        return Radians(std::fmod(std::fmod(r_, TWO_PI) + TWO_PI, TWO_PI));
    }
    float cos() const { return std::cos(r_); }
    float sin() const { return std::sin(r_); }

private:
    float r_;
};

struct Mat3 {  // three columns, math.rs:277-283
    Vec3 c0, c1, c2;

    static Mat3 from_rows(Vec3 r0, Vec3 r1, Vec3 r2) {
        return Mat3{{r0.x, r1.x, r2.x}, {r0.y, r1.y, r2.y}, {r0.z, r1.z, r2.z}};
    }
    Vec3 row(int i) const {
        switch (i) {
        case 0: return {c0.x, c1.x, c2.x};
        case 1: return {c0.y, c1.y, c2.y};
        default: return {c0.z, c1.z, c2.z};
        }
    }
    Mat3 operator*(const Mat3& rhs) const {  // rows of self . columns of rhs, math.rs:326-353
        Vec3 r0 = row(0), r1 = row(1), r2 = row(2);
        return from_rows({r0 * rhs.c0, r0 * rhs.c1, r0 * rhs.c2},
                         {r1 * rhs.c0, r1 * rhs.c1, r1 * rhs.c2},
                         {r2 * rhs.c0, r2 * rhs.c1, r2 * rhs.c2});
    }
    Vec3 operator*(const Vec3& v) const {  // math.rs:355-367
        return {row(0) * v, row(1) * v, row(2) * v};
    }
    Mat3 scaled(float s) const {
        return Mat3{{s * c0.x, s * c0.y, s * c0.z},
                    {s * c1.x, s * c1.y, s * c1.z},
                    {s * c2.x, s * c2.y, s * c2.z}};
    }
    static Mat3 rotation_x(Radians a) {  // math.rs:386-395
        float c = a.cos(), s = a.sin();
        return Mat3{{1, 0, 0}, {0, c, s}, {0, -s, c}};
    }
    static Mat3 rotation_y(Radians a) {  // math.rs:397-406
        float c = a.cos(), s = a.sin();
        return Mat3{{c, 0, -s}, {0, 1, 0}, {s, 0, c}};
    }
    static Mat3 rotation_z(Radians a) {  // math.rs:408-416
        float c = a.cos(), s = a.sin();
        return Mat3{{c, s, 0}, {-s, c, 0}, {0, 0, 1}};
    }
    void store(float m[9]) const {
        const Vec3* cols[3] = {&c0, &c1, &c2};
        for (int c = 0; c < 3; ++c) {
            m[3 * c + 0] = cols[c]->x;
            m[3 * c + 1] = cols[c]->y;
            m[3 * c + 2] = cols[c]->z;
        }
    }
    static Mat3 load(const float m[9]) {
        return Mat3{{m[0], m[1], m[2]}, {m[3], m[4], m[5]}, {m[6], m[7], m[8]}};
    }
};

Mat3 camera_matrix(const KifsCameraData& cam) {  // data.rs:91-98: Rz(phi) * Ry(-theta)
    return Mat3::rotation_z(Radians(cam.phi)) * Mat3::rotation_y(Radians(-cam.theta));
}

float linear_from_gamma(float g) {  // packed.rs:119-125
    return g <= 0.04045f ? g / 12.92f : std::pow((g + 0.055f) / 1.055f, 2.4f);
}

}  // namespace

extern "C" {

void kifs_host_gui_default(KifsGuiData* g) {  // data.rs:145-160
    if (!g) return;
    std::memset(g, 0, sizeof *g);
    g->max_iterations = 256;
    g->max_distance = 1000.0f;
    g->epsilon = 0.0001f;
    g->fractal_color[0] = g->fractal_color[1] = g->fractal_color[2] = 200;
    g->background_color[0] = g->background_color[1] = g->background_color[2] = 0;
    g->is_heatmap = 0;
    g->fractal_group = KIFS_GROUP_KIFS;     // FractalGroup::default, scene.rs:6-7
    g->primitive_shape = KIFS_PRIM_SPHERE;  // PrimitiveShape::default, scene.rs:37-38
    g->power = 2.0f;
    g->constant[0] = -0.1f;
    g->constant[1] = 0.6f;
    g->constant[2] = 0.9f;
    g->constant[3] = -0.3f;
}

void kifs_host_camera_default(KifsCameraData* c) {  // data.rs:105-113
    if (!c) return;
    c->origin_distance = 5.0f;
    c->min_distance = 2.0f;
    c->phi = 0.0f;
    c->theta = 0.0f;
}

int kifs_host_screen(uint32_t width, uint32_t height, KifsScreenUniform* out) {  // data.rs:66-81
    if (!out) return KIFS_ERR_BAD_ARG;
    if (width == 0 || height == 0) return KIFS_ERR_BAD_SIZE;  // render.rs:211
    float w = static_cast<float>(width), h = static_cast<float>(height);
    out->width = w;
    out->height = h;
    out->aspect_ratio = w / h;
    return KIFS_OK;
}

int kifs_host_camera(const KifsCameraData* cam, KifsCameraUniform* out) {  // data.rs:115-129
    if (!cam || !out) return KIFS_ERR_BAD_ARG;
    Mat3 m = camera_matrix(*cam);
    // `origin_distance * camera_matrix * Vector3(1,0,0)` groups as (d * M) * e_x
    Vec3 origin = m.scaled(cam->origin_distance) * Vec3{1.0f, 0.0f, 0.0f};
    std::memset(out, 0, sizeof *out);
    out->origin[0] = origin.x;
    out->origin[1] = origin.y;
    out->origin[2] = origin.z;
    const Vec3* cols[3] = {&m.c0, &m.c1, &m.c2};
    for (int c = 0; c < 3; ++c) {  // each column extended with 0 (packed.rs:78-92)
        out->matrix[c][0] = cols[c]->x;
        out->matrix[c][1] = cols[c]->y;
        out->matrix[c][2] = cols[c]->z;
        out->matrix[c][3] = 0.0f;
    }
    return KIFS_OK;
}

int kifs_host_options(const KifsGuiData* g, KifsOptionsUniform* out) {  // data.rs:176-220
    if (!g || !out) return KIFS_ERR_BAD_ARG;
    std::memset(out, 0, sizeof *out);
    out->max_iterations = static_cast<int32_t>(g->max_iterations);
    out->max_distance = g->max_distance;
    out->epsilon = g->epsilon;
    for (int i = 0; i < 3; ++i) {  // LinearRgb::from_srgb: byte / 256 (packed.rs:131-137)
        out->fractal_color[i] = linear_from_gamma(static_cast<float>(g->fractal_color[i]) / 256.0f);
        out->background_color[i] =
            linear_from_gamma(static_cast<float>(g->background_color[i]) / 256.0f);
    }
    out->is_heatmap = g->is_heatmap ? 1u : 0u;
    out->fractal_group_id = g->fractal_group;
    out->primitive_id = g->primitive_shape;
    out->power = g->power;
    std::memcpy(out->constant, g->constant, sizeof out->constant);
    return KIFS_OK;
}

int kifs_host_rotate(KifsCameraData* cam, float dphi, float dtheta) {  // graphics.rs:280-302
    if (!cam) return KIFS_ERR_BAD_ARG;
    Radians phi(cam->phi + dphi), theta(cam->theta + dtheta);
    theta = theta.clamp(-PI / 2.0f, PI / 2.0f);
    cam->phi = phi.standardize().radians();
    cam->theta = theta.radians();
    return KIFS_OK;
}

int kifs_host_zoom(KifsCameraData* cam, float distance) {  // graphics.rs:268-278
    if (!cam) return KIFS_ERR_BAD_ARG;
    cam->origin_distance = std::fmax(cam->min_distance, cam->origin_distance - distance);
    return KIFS_OK;
}

int kifs_host_mouse_motion(KifsCameraData* cam, double dx, double dy) {  // render.rs:255-270
    if (!cam) return KIFS_ERR_BAD_ARG;
    Radians dphi = Radians::from_degrees(static_cast<float>(-(dx / 10.0)));
    Radians dtheta = Radians::from_degrees(static_cast<float>(dy / 10.0));
    return kifs_host_rotate(cam, dphi.radians(), dtheta.radians());
}

float kifs_host_radians_from_degrees(float degrees) { return Radians::from_degrees(degrees).radians(); }

void kifs_host_camera_matrix(const KifsCameraData* cam, float m[9]) {
    if (cam && m) camera_matrix(*cam).store(m);
}

void kifs_host_rotation_matrix(int axis, float radians, float m[9]) {
    if (!m) return;
    Radians a(radians);
    (axis == 0 ? Mat3::rotation_x(a) : axis == 1 ? Mat3::rotation_y(a) : Mat3::rotation_z(a)).store(m);
}

void kifs_host_mat3_mul(const float a[9], const float b[9], float out[9]) {
    (Mat3::load(a) * Mat3::load(b)).store(out);
}

void kifs_host_mat3_vec(const float a[9], const float v[3], float out[3]) {
    Vec3 r = Mat3::load(a) * Vec3{v[0], v[1], v[2]};
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
}

}  // extern "C"
