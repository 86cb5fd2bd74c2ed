// kifs_api.cpp -- the C ABI of include/kifs_hip.h: context lifetime, uniform upload,
// frame render.  Host code only; kernels live in kifs_kernels.hip.
//
// A kifs_ctx plays the part of the reference's GraphicState (render/graphics.rs:25-37):
// it owns the "device objects" (stream, events, the sRGB table in HBM, a scratch frame
// for host-destination renders) and a copy of the three uniform images.  There is no
// CPU path: every entry point that produces pixels launches the HIP kernels or fails.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <vector>

#include "../../include/kifs_hip.h"
#include "kifs_internal.hpp"

static_assert(sizeof(KifsScreenUniform) == 12, "ScreenUniformData is 12 bytes (data.rs:17-23)");
static_assert(sizeof(KifsCameraUniform) == 64, "CameraUniformData is 64 bytes (data.rs:25-31)");
static_assert(sizeof(KifsOptionsUniform) == 80, "OptionsUniformData is 80 bytes (data.rs:33-49)");
static_assert(offsetof(KifsCameraUniform, matrix) == 16, "camera.matrix at byte 16");
static_assert(offsetof(KifsOptionsUniform, fractal_color) == 16, "fractal_color at byte 16");
static_assert(offsetof(KifsOptionsUniform, background_color) == 32, "background_color at 32");
static_assert(offsetof(KifsOptionsUniform, is_heatmap) == 44, "is_heatmap at 44");
static_assert(offsetof(KifsOptionsUniform, power) == 56, "power at 56");
static_assert(offsetof(KifsOptionsUniform, constant) == 64, "constant at 64");
static_assert(sizeof(kifs::BatchParams) <= 4096, "the kernel argument segment is limited to 4 KB");
static_assert(KIFS_STRIPE_ROWS == kifs::TILE_H, "a stripe is one row of the kernels' tiles");

// A row shard is a list of 8-row stripes of the frame (kifs_shard_stripes); its device image --
// first frame row of every stripe -- is cached per context (a root unpacks the shards of every peer).
struct RowTable {
    std::vector<int> stripes;  // stripe indices, ascending
    uint32_t* d_rows = nullptr;
};

// Tile order tables are keyed by the geometry they were built for and kept on the
// device; a context alternates between very few geometries (full frame, its band or shard).
struct TileTable {
    int width = 0, height = 0, y0 = 0, y1 = 0;
    const RowTable* rows = nullptr;   // non-null: the table of a row shard (then y0 = 0, y1 = height)
    uint32_t* d_order = nullptr;      // order used by the next launch
    uint32_t* d_order_alt = nullptr;  // the other half of the double buffer (the sort's target)
    uint32_t* d_cost[2] = {nullptr, nullptr};  // per-tile cost, written by launch k into [k & 1]
    hipEvent_t rendered[2] = {nullptr, nullptr};  // [0]: after the cost-recording launch; [1]: stream changes
    hipEvent_t sorted = nullptr;      // recorded after the sort that fills d_order_alt
    uint64_t launches = 0;            // consecutive feedback launches made with this table
    hipStream_t last_stream = nullptr;  // stream of the latest of them
    bool sort_pending = false;        // d_order_alt holds (or will hold) a fresh order
    bool feedback = true;             // reorder from costs (off once the caller pins an order)
    uint32_t count = 0;
    uint32_t cost_shift = 0;          // scale of the costs the latest recording launch wrote (see record_costs)
    uint64_t last_use = 0;
};
constexpr int MAX_TILE_TABLES = 8;

struct kifs_ctx {
    int device = 0;
    TileTable tables[MAX_TILE_TABLES];
    std::vector<RowTable*> row_tables;  // never evicted while the context lives (a few hundred bytes each)
    uint64_t use_clock = 0;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;  // tile-order sorts run here, beside the renders
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    float* d_srgb = nullptr;       // 256 thresholds
    uint8_t* d_scratch = nullptr;  // frame staging for host destinations
    size_t scratch_bytes = 0;
    KifsScreenUniform screen{};
    KifsCameraUniform camera{};
    KifsOptionsUniform options{};
    bool have_screen = false, have_camera = false, have_options = false;
    int sdf_iters = 100, normal_iters = 10, fold_iters = 10;  // julia.wgsl:2-3, kifs.wgsl:72
    KifsExtensions ext{};  // all zero: the reference's behaviour
    int frames_in_flight = 1;  // kifs_set_frames_in_flight
    int last_round_steps = 0;  // kifs_debug_last_round_steps
    int last_group_tiles = -1; // kifs_debug_last_group_tiles
    float h_srgb[256] = {};    // host copy of the sRGB threshold table (d_srgb)
    // per-launch profiling ring (kifs_set_profiling)
    bool profiling = false;
    int prof_every = 1;      // time every n-th launch
    uint64_t prof_seen = 0;  // launches seen while profiling
    std::vector<hipEvent_t> prof_a, prof_b;
    size_t prof_count = 0;
    double last_ms = -1.0;
    bool timing_pending = false;
    unsigned long long* d_counters = nullptr;  // diagnostics buffer, see FrameParams
    size_t counter_words = 0;
    // View tables of batches beyond MAX_BATCH_INLINE: a ring of device tables, each with its pinned host
    // image and an event recorded after the launch that read it (allocated on first use).
    static constexpr int VIEW_RING = 4;
    kifs::BatchView* d_views[VIEW_RING] = {};
    kifs::BatchView* h_views[VIEW_RING] = {};
    hipEvent_t views_used[VIEW_RING] = {};
    bool views_busy[VIEW_RING] = {};
    int view_slot = 0;
};

namespace {

// Tuning overrides (KIFS_ROUND_STEPS, KIFS_GROUP_TILES, KIFS_TILE_FEEDBACK, KIFS_FEEDBACK_PERIOD,
// KIFS_BATCH_PERIOD; KIFS_LDS_PAD in kifs_kernels.hip) are honoured only when KIFS_TUNING=1 is set as
// well: they exist for tools/sweep_kernels.sh and friends, not for production hosts.  -1 = not set.
int tuning_knob(const char* name) {
    static const bool enabled = [] {
        const char* e = std::getenv("KIFS_TUNING");
        return e && e[0] == '1';
    }();
    if (!enabled) return -1;
    const char* e = std::getenv(name);
    return e ? int(std::strtol(e, nullptr, 10)) : -1;
}

// KIFS_DEBUG=1 prints the failing HIP call to stderr (status codes stay the contract).
bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    static const bool verbose = std::getenv("KIFS_DEBUG") != nullptr;
    if (verbose) std::fprintf(stderr, "kifs: %s failed: %s\n", what, hipGetErrorString(e));
    return false;
}

struct DeviceGuard {  // make ctx's device current for the duration of a call
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

int frame_dims(const kifs_ctx* c, int* w, int* h) {
    // width/height arrive as f32 (data.rs:71-73 casts u32 -> f32); demand exact integers
    float fw = c->screen.width, fh = c->screen.height;
    if (!(fw >= 1.0f) || !(fh >= 1.0f) || fw > 65536.0f || fh > 65536.0f) return KIFS_ERR_BAD_SIZE;
    if (fw != std::floor(fw) || fh != std::floor(fh)) return KIFS_ERR_BAD_SIZE;
    *w = int(fw);
    *h = int(fh);
    return KIFS_OK;
}

// Exact squared form of `norm > T` for norm = sqrtf(n2) (correctly rounded, monotone):
// returns the largest binary32 v with sqrtf(v) <= T, so that norm > T  <=>  n2 > v.
float squared_threshold(float T) {
    if (T != T) return INFINITY;          // norm > NaN is never true
    if (T < 0.0f) return -1.0f;           // every non-NaN norm (>= 0) exceeds a negative T
    if (T == INFINITY) return INFINITY;
    double sq = double(T) * double(T);
    float v = sq >= double(FLT_MAX) ? FLT_MAX : float(sq);
    while (v > 0.0f && std::sqrt(v) > T) v = std::nextafterf(v, -INFINITY);
    for (;;) {
        float up = std::nextafterf(v, INFINITY);
        if (up != INFINITY && std::sqrt(up) <= T) v = up; else break;
    }
    return v;
}

// Exact squared form of `norm < T`: returns the smallest binary32 v with sqrtf(v) >= T, so
// that norm < T  <=>  n2 < v  (n2 >= +0 or NaN).
float squared_lower_threshold(float T) {
    if (T != T || T <= 0.0f) return 0.0f;  // norm < T is never true
    if (T == INFINITY) return INFINITY;    // true for every finite norm
    double sq = double(T) * double(T);
    float v = sq >= double(FLT_MAX) ? FLT_MAX : float(sq);
    while (std::sqrt(v) < T) {
        if (v == FLT_MAX) return INFINITY;
        v = std::nextafterf(v, INFINITY);
    }
    for (;;) {
        float down = std::nextafterf(v, -INFINITY);
        if (down >= 0.0f && std::sqrt(down) >= T) v = down; else break;
    }
    return v;
}

int fill_params(const kifs_ctx* c, kifs::FrameParams* P) {
    int w, h;
    int st = frame_dims(c, &w, &h);
    if (st != KIFS_OK) return st;
    const KifsCameraUniform& cam = c->camera;
    const KifsOptionsUniform& o = c->options;
    if (o.fractal_group_id > 2u) return KIFS_ERR_BAD_ARG;  // FractalGroup::from_id -> None
    P->height = c->screen.height;
    P->aspect = c->screen.aspect_ratio;
    P->origin = {cam.origin[0], cam.origin[1], cam.origin[2]};
    P->m0 = {cam.matrix[0][0], cam.matrix[0][1], cam.matrix[0][2]};
    P->m1 = {cam.matrix[1][0], cam.matrix[1][1], cam.matrix[1][2]};
    P->m2 = {cam.matrix[2][0], cam.matrix[2][1], cam.matrix[2][2]};
    P->max_iterations = o.max_iterations;
    P->max_distance = o.max_distance;
    P->epsilon = o.epsilon;
    P->fractal_color = {o.fractal_color[0], o.fractal_color[1], o.fractal_color[2]};
    P->background_color = {o.background_color[0], o.background_color[1], o.background_color[2]};
    P->is_heatmap = o.is_heatmap;
    P->power = o.power;
    P->c = {o.constant[0], o.constant[1], o.constant[2], o.constant[3]};
    P->sdf_iters = c->sdf_iters;
    P->normal_iters = c->normal_iters;
    P->fold_iters = c->fold_iters;
    P->soft_shadow = c->ext.soft_shadow;
    P->shadow_steps = c->ext.shadow_steps;
    P->shadow_k = c->ext.shadow_k;
    P->shadow_t0 = c->ext.shadow_t0;
    P->shadow_max_t = c->ext.shadow_max_t;
    P->bound_n2 = squared_threshold(2.0f + o.epsilon);
    {   // Bounding-sphere culls: every scene's estimate obeys d(p) >= |p| - B, so outside radius
        // R = B + epsilon (plus margin) `d < epsilon` cannot happen.  B per scene:
        //   Julia / gen-Julia: 2 (the patch of julia.wgsl:8-9)      sphere r=1: 1
        //   cylinder (r=1, half-height 2): sqrt(5)                   box (1,1,1): sqrt(3)
        //   torus (1, 0.3): 1.3        bunny: 1 (patch |p| - 0.8 outside the unit ball)
        //   Sierpinski: 2 -- folds are isometries and pos <- 2 pos - 1 gives r_k >= 2^k r_0 -
        //   sqrt(3)(2^k - 1), hence (r_k - 2)/2^k >= r_0 - 2 for every number of folds.
        float B = 2.0f;
        if (o.fractal_group_id == uint32_t(kifs::GROUP_KIFS)) {
            switch (o.primitive_id) {
            case kifs::PRIM_SPHERE: B = 1.0f; break;
            case kifs::PRIM_CYLINDER: B = 2.2360680f; break;
            case kifs::PRIM_BOX: B = 1.7320508f; break;
            case kifs::PRIM_TORUS: B = 1.3f; break;
            case kifs::PRIM_SIERPINSKI: B = 2.0f; break;
            case kifs::PRIM_BUNNY: B = 1.0f; break;
            default: B = -1.0f; break;  // unknown id: the SDF is the constant 1, no bound
            }
        }
        const float R = B + o.epsilon;
        const bool sane = B > 0.0f && R > 0.5f && R < 1.0e6f && o.epsilon >= 0.0f;
        P->cull_n2 = sane ? 1.1f * R * R : 0.0f;
        // the wave-level quick exit uses a sphere 9 % larger again; like the culls, not in heatmap mode
        P->quick_cull_n2 = (sane && !o.is_heatmap && o.max_iterations > 0) ? 1.2f * R * R : 0.0f;
        P->inv_height = 1.0f / c->screen.height;
        // Tile-level form (render_wave_kernel): for an orthonormal camera matrix |d| >= 1 and two pixel
        // centres of a 32 x 8 tile are at most (31, 7) pixels = (31, 7) * 2 / height apart in uv, so a
        // ray of the tile and the ray through the tile's centre differ by at most
        // asin(|(31, 7)| / height) <= 1.05 * 31.8 / height radians (the ratio is below 0.5 from 64 rows);
        // 34 / height leaves 2 % for the matrix check's tolerance.  enqueue_batch() switches it off when
        // a view's matrix is not orthonormal.
        P->tile_cull_sqrtk = std::sqrt(P->quick_cull_n2);
        P->tile_cull_beta = (P->quick_cull_n2 > 0.0f && c->screen.height >= 64.0f) ? 34.0f / c->screen.height : 0.0f;
    }
    {   // Ray re-queuing (render_group_kernel): rounds of this many march steps -- 16 for the Julia
        // pipelines (generalised Julia: 1080p lone 0.882 -> 0.869 ms, x8 +2.7 %, x48 +1 % over rounds of 8),
        // 8 for the others (measured; KIFS_ROUND_STEPS overrides, 0 switches it off).
        // Not for heatmap frames (their per-ray step count is kept by the one-wave-per-block
        // march), not with a non-positive epsilon (the queue rebuilds p from t and relies on
        // t > 0 after a step), not for marches too short to repay the rounds' barriers.
        static const int forced = tuning_knob("KIFS_ROUND_STEPS");
        int rounds = forced >= 0 ? forced : (o.fractal_group_id != uint32_t(kifs::GROUP_KIFS) ? 16 : 8);
        if (forced < 0 && rounds == 16 && o.max_iterations < 32 && o.fractal_group_id == uint32_t(kifs::GROUP_GENJULIA))
            rounds = 8;  // (a short march of heavy steps still repays shorter rounds)
        if (o.is_heatmap || !(o.epsilon > 0.0f) || o.max_iterations < 2 * rounds) rounds = 0;
        P->round_steps = rounds;
    }
    P->orbit_blocks = c->sdf_iters / 6;
    P->orbit_rem = c->sdf_iters % 6;
    P->fold_n2_stop = squared_lower_threshold(o.max_distance);
    P->width = w;
    P->y0 = 0;
    P->y1 = h;
    P->stripe_rows = nullptr;
    P->out_frame_rows = 0;
    P->encode = KIFS_ENCODE_SRGB;
    P->pitch_words = uint32_t(w);
    P->out = nullptr;
    P->srgb_table = c->d_srgb;
    P->tile_order = nullptr;
    P->tile_count = 0;
    P->tile_cost = nullptr;
    P->counters = c->d_counters;
    P->workgroups_per_cu = 0;
    P->group_tiles = 1;
    return KIFS_OK;
}

bool is_device_pointer(const void* p) {
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // unregistered host memory reports an error; clear it
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

static void free_table(TileTable& t) {
    if (t.d_order) (void)hipFree(t.d_order);
    if (t.d_order_alt) (void)hipFree(t.d_order_alt);
    for (int i = 0; i < 2; ++i) {
        if (t.d_cost[i]) (void)hipFree(t.d_cost[i]);
        if (t.rendered[i]) (void)hipEventDestroy(t.rendered[i]);
    }
    if (t.sorted) (void)hipEventDestroy(t.sorted);
    t = TileTable();
}

// KIFS_TILE_FEEDBACK (tuning): 0 = never, 1 (default) = per pipeline thresholds, 2 = every frame of 2048+ tiles.
static int tile_feedback_mode() {
    static const int mode = [] {
        const int v = tuning_knob("KIFS_TILE_FEEDBACK");
        return v >= 0 ? v : 1;
    }();
    return mode;
}

// Order in which workgroups take tiles: nearest to the frame centre first (squared
// distance of the tile centre, ties by row then column), so the long rays start first.
// Tiles are TILE_W x TILE_H pixels; rows are counted from the top of the band.
// Device image of a stripe list, cached by content.  Stripes must be ascending and inside the frame.
const RowTable* row_table(kifs_ctx* c, const int* stripes, int n, int height) {
    for (const RowTable* r : c->row_tables)
        if (int(r->stripes.size()) == n && std::equal(stripes, stripes + n, r->stripes.begin())) return r;
    std::vector<uint32_t> rows(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) {
        if (stripes[i] < 0 || int64_t(stripes[i]) * kifs::TILE_H >= height || (i > 0 && stripes[i] <= stripes[i - 1]))
            return nullptr;
        rows[size_t(i)] = uint32_t(stripes[i]) * uint32_t(kifs::TILE_H);
    }
    if (c->row_tables.size() >= 4096) return nullptr;  // a caller inventing a new partition every frame
    RowTable* r = new (std::nothrow) RowTable();
    if (!r) return nullptr;
    r->stripes.assign(stripes, stripes + n);
    if (!hip_ok(hipMalloc(reinterpret_cast<void**>(&r->d_rows), std::max<size_t>(rows.size(), 1) * sizeof(uint32_t)),
                "hipMalloc(stripe rows)") ||
        (n > 0 && !hip_ok(hipMemcpy(r->d_rows, rows.data(), rows.size() * sizeof(uint32_t), hipMemcpyHostToDevice),
                          "hipMemcpy(stripe rows)"))) {
        if (r->d_rows) (void)hipFree(r->d_rows);
        delete r;
        return nullptr;
    }
    c->row_tables.push_back(r);
    return r;
}

// `rows` non-null: the table of a row shard (tile row j = stripe rows->stripes[j]; y0 = 0, y1 = height).
TileTable* tile_table(kifs_ctx* c, int width, int height, int y0, int y1, const RowTable* rows = nullptr) {
    TileTable* slot = nullptr;
    for (auto& t : c->tables) {
        if (t.d_order && t.width == width && t.height == height && t.y0 == y0 && t.y1 == y1 && t.rows == rows) {
            t.last_use = ++c->use_clock;
            return &t;
        }
        if (!slot || t.last_use < slot->last_use) slot = &t;
    }
    const int tx = (width + kifs::TILE_W - 1) / kifs::TILE_W;
    const int ty = rows ? int(rows->stripes.size()) : (y1 - y0 + kifs::TILE_H - 1) / kifs::TILE_H;
    if (tx > 0xffff || ty > 0xffff) return nullptr;
    struct Key { int64_t d2; uint32_t id; };
    std::vector<Key> keys;
    keys.reserve(size_t(tx) * ty);
    for (int j = 0; j < ty; ++j)
        for (int i = 0; i < tx; ++i) {
            // doubled coordinates keep everything in integers
            int64_t cx = int64_t(2 * i + 1) * kifs::TILE_W - width;
            const int64_t first = rows ? int64_t(rows->stripes[size_t(j)]) * kifs::TILE_H
                                       : int64_t(y0) + int64_t(j) * kifs::TILE_H;  // the tile's first frame row
            int64_t cy = 2 * first + kifs::TILE_H - height;
            keys.push_back({cx * cx + cy * cy, (uint32_t(j) << 16) | uint32_t(i)});
        }
    std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) {
        return a.d2 != b.d2 ? a.d2 < b.d2 : a.id < b.id;
    });
    std::vector<uint32_t> order(keys.size());
    for (size_t k = 0; k < keys.size(); ++k) order[k] = keys[k].id;
    // the slot being replaced may still be read by an enqueued launch: drain first
    if (slot->d_order) {
        hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize(before tile table eviction)");
        free_table(*slot);
    }
    const size_t bytes = order.size() * sizeof(uint32_t);
    if (!hip_ok(hipMalloc(reinterpret_cast<void**>(&slot->d_order), bytes), "hipMalloc(tile order)") ||
        !hip_ok(hipMalloc(reinterpret_cast<void**>(&slot->d_order_alt), bytes), "hipMalloc(tile order 2)") ||
        !hip_ok(hipMalloc(reinterpret_cast<void**>(&slot->d_cost[0]), bytes), "hipMalloc(tile cost)") ||
        !hip_ok(hipMalloc(reinterpret_cast<void**>(&slot->d_cost[1]), bytes), "hipMalloc(tile cost 2)") ||
        !hip_ok(hipEventCreateWithFlags(&slot->rendered[0], hipEventDisableTiming), "hipEventCreate") ||
        !hip_ok(hipEventCreateWithFlags(&slot->rendered[1], hipEventDisableTiming), "hipEventCreate") ||
        !hip_ok(hipEventCreateWithFlags(&slot->sorted, hipEventDisableTiming), "hipEventCreate") ||
        !hip_ok(hipMemcpy(slot->d_order, order.data(), bytes, hipMemcpyHostToDevice), "hipMemcpy(tile order)") ||
        !hip_ok(hipMemset(slot->d_cost[0], 0, bytes), "hipMemset(tile cost)") ||
        !hip_ok(hipMemset(slot->d_cost[1], 0, bytes), "hipMemset(tile cost)")) {
        free_table(*slot);
        return nullptr;
    }
    slot->width = width; slot->height = height; slot->y0 = y0; slot->y1 = y1;
    slot->rows = rows;
    slot->count = uint32_t(order.size());
    slot->last_use = ++c->use_clock;
    return slot;
}

// How many of a launch's tiles (per view) can contain rays with real work: those the projected
// bounding sphere of the scene covers (fill_params: every estimate obeys d(p) >= |p| - B, so a ray that
// passes the origin at more than R = B + epsilon never hits).  pi r_px^2 / 256 with
// r_px = H/2 * R / sqrt(d^2 - R^2) (focal length 1, uv.y in [-1, 1]), scaled by the launch's share of the
// frame's rows; every tile when the camera is inside the sphere or the culls are off.  This is the
// quantity the launch-shape rules below are written in: it follows the camera distance and the frame
// size together, where tile counts and pixel counts do not.
double disc_tiles(const kifs::FrameParams& P, int frame_height, uint32_t tile_count) {
    if (P.cull_n2 <= 0.0f || P.is_heatmap) return double(tile_count);
    const double R2 = double(P.cull_n2) / 1.1;  // (B + epsilon)^2
    const double d2 = double(P.origin.x) * P.origin.x + double(P.origin.y) * P.origin.y +
                      double(P.origin.z) * P.origin.z;
    const double frame_px = double(P.width) * double(frame_height);
    double disk_px = frame_px;  // camera inside the sphere: everything is a candidate
    if (d2 > R2 * 1.0001) {
        const double r_uv = std::sqrt(R2 / (d2 - R2));          // tangent of the sphere's angular radius
        const double r_px = r_uv * 0.5 * double(frame_height);
        disk_px = std::min(frame_px, 3.14159265358979 * r_px * r_px);
    }
    // a band or shard of a frame gets its share of the disk
    const double share = frame_px > 0 ? double(tile_count) * (kifs::TILE_W * kifs::TILE_H) / frame_px : 1.0;
    return std::min(double(tile_count), disk_px * std::min(1.0, share) / (kifs::TILE_W * kifs::TILE_H));
}

// Residency rule for the Julia pipelines.  The long rays of a frame slow each other down as soon
// as they share a SIMD (~1490 cycles per march step alone, ~1570 with one neighbour, ~1900 with
// seven), and after the bounding-sphere culls nothing else needs the slots: the only tiles
// with real work are those the projected bounding sphere covers.  If those are few enough to be
// spread over the 256 CUs in a couple of rounds, capping residency lets every long wave run
// near its lone-wave speed (1080p, camera at distance 5: 207 -> 172 us at one workgroup per
// CU); if they are many (4096^2, or a camera close to the fractal) the frame needs every slot for
// its long-marching waves and full residency wins (4096^2: 0.70 ms vs 1.95 ms capped).
int residency_for(const kifs::FrameParams& P, uint32_t group, double heavy_tiles) {
    if (group != kifs::GROUP_JULIA || P.cull_n2 <= 0.0f || P.is_heatmap) return 0;
    if (heavy_tiles <= 1024.0) return 1;
    if (heavy_tiles <= 2048.0) return 2;
    return 0;
}

// The background pixel, encoded exactly as the kernels would (unorm8 / srgb8 of kifs_device_math.hpp).
uint32_t background_pixel(const kifs_ctx* c, kifs::V3 colour, int encode) {
    uint32_t ch[3];
    const float bg[3] = {colour.x, colour.y, colour.z};
    for (int i = 0; i < 3; ++i) {
        const float x = bg[i];
        if (encode == KIFS_ENCODE_SRGB) {
            uint32_t k = 0;
            for (uint32_t step = 128; step >= 1; step >>= 1) k += (x >= c->h_srgb[k + step]) ? step : 0u;
            ch[i] = k;
        } else {
            float v = (x >= 0.0f) ? x : 0.0f;
            v = (v > 1.0f) ? 1.0f : v;
            ch[i] = uint32_t(int(v * 255.0f + 0.5f));
        }
    }
    return ch[0] | (ch[1] << 8) | (ch[2] << 16) | 0xff000000u;
}

// One launch: `count` frames (count == 1: the context's camera; count > 1: cameras[i] -> outs[i])
// sharing everything else.
// `stripes` non-null: the launch renders that row shard (y0 = 0, y1 = height) instead of a band, into
// packed rows (in_place == 0) or at the rows' frame positions (in_place != 0).
int enqueue_batch(kifs_ctx* c, hipStream_t stream, int count, const KifsCameraUniform* cameras,
                  uint8_t* const* outs, size_t pitch, int y0, int y1, int encode,
                  const int* stripes = nullptr, int n_stripes = 0, int in_place = 0) {
    hip_ok(hipGetLastError(), "stale error before enqueue");
    if (!c->have_screen || !c->have_options || (!c->have_camera && !cameras)) return KIFS_ERR_UNCONFIGURED;
    if (count < 1 || count > kifs::MAX_BATCH || !outs) return KIFS_ERR_BAD_ARG;
    for (int i = 0; i < count; ++i)
        if (!outs[i] || (reinterpret_cast<uintptr_t>(outs[i]) & 3u) != 0) return KIFS_ERR_BAD_ARG;
    uint8_t* const dev_out = outs[0];
    if (encode != KIFS_ENCODE_UNORM && encode != KIFS_ENCODE_SRGB) return KIFS_ERR_BAD_ARG;
    kifs::BatchParams B;
    kifs::FrameParams& P = B.frame;
    int st = fill_params(c, &P);
    if (st != KIFS_OK) return st;
    B.count = count;
    B.table = nullptr;
    // a batch beyond the kernel argument's room: the views go through a device table (ring slot `vs`)
    const bool big = count > kifs::MAX_BATCH_INLINE;
    int vs = -1;
    if (big) {
        vs = c->view_slot;
        c->view_slot = (vs + 1) % kifs_ctx::VIEW_RING;
        if (!c->d_views[vs]) {
            const size_t bytes = sizeof(kifs::BatchView) * size_t(kifs::MAX_BATCH);
            if (!hip_ok(hipMalloc(reinterpret_cast<void**>(&c->d_views[vs]), bytes), "hipMalloc(view table)") ||
                !hip_ok(hipHostMalloc(reinterpret_cast<void**>(&c->h_views[vs]), bytes, hipHostMallocDefault), "hipHostMalloc(view table)") ||
                !hip_ok(hipEventCreateWithFlags(&c->views_used[vs], hipEventDisableTiming), "hipEventCreate(view table)"))
                return KIFS_ERR_RUNTIME;
        }
        // the launch that last read this slot (four big launches ago) must be over before its images change
        if (c->views_busy[vs] && !hip_ok(hipEventSynchronize(c->views_used[vs]), "wait(view table)")) return KIFS_ERR_RUNTIME;
        c->views_busy[vs] = false;
    }
    for (int i = 0; i < count; ++i) {
        const KifsCameraUniform& cam = cameras ? cameras[i] : c->camera;
        kifs::BatchView& v = big ? c->h_views[vs][i] : B.view[i];
        v.origin = {cam.origin[0], cam.origin[1], cam.origin[2]};
        v.m0 = {cam.matrix[0][0], cam.matrix[0][1], cam.matrix[0][2]};
        v.m1 = {cam.matrix[1][0], cam.matrix[1][1], cam.matrix[1][2]};
        v.m2 = {cam.matrix[2][0], cam.matrix[2][1], cam.matrix[2][2]};
        v.out = reinterpret_cast<uint32_t*>(outs[i]);
        if (P.tile_cull_beta > 0.0f) {  // the tile-level cull's angle bound assumes an orthonormal matrix
            const kifs::V3* m[3] = {&v.m0, &v.m1, &v.m2};
            for (int a = 0; a < 3; ++a)
                for (int b = a; b < 3; ++b) {
                    const double dot = double(m[a]->x) * m[b]->x + double(m[a]->y) * m[b]->y + double(m[a]->z) * m[b]->z;
                    if (!(std::fabs(dot - (a == b ? 1.0 : 0.0)) <= 1.0e-3)) P.tile_cull_beta = 0.0f;
                }
        }
    }
    const kifs::BatchView& view0 = big ? c->h_views[vs][0] : B.view[0];
    P.origin = view0.origin;
    P.m0 = view0.m0;
    P.m1 = view0.m1;
    P.m2 = view0.m2;
    const int h = P.y1;
    if (y0 < 0 || y1 > h || y0 > y1) return KIFS_ERR_BAD_ARG;
    if (pitch < size_t(P.width) * 4 || (pitch & 3u) != 0 || (pitch >> 2) > 0xffffffffull)
        return KIFS_ERR_BAD_SIZE;
    P.y0 = y0;
    P.y1 = y1;
    P.encode = encode;
    P.background_rgba = background_pixel(c, P.background_color, encode);
    P.pitch_words = uint32_t(pitch >> 2);
    P.out = reinterpret_cast<uint32_t*>(dev_out);
    if (y1 == y0) return KIFS_OK;
    const RowTable* rows = nullptr;
    if (stripes) {
        if (n_stripes == 0) return KIFS_OK;
        rows = row_table(c, stripes, n_stripes, h);
        if (!rows) return KIFS_ERR_BAD_ARG;
        P.stripe_rows = rows->d_rows;
        P.out_frame_rows = in_place ? 1 : 0;
    }
    TileTable* tt = tile_table(c, P.width, h, y0, y1, rows);
    if (!tt) return KIFS_ERR_RUNTIME;
    // Temporal feedback on the tile order.  A launch can leave a cost per tile (the run time of
    // the tile's slowest wave); a one-workgroup counting sort on the context's side stream turns
    // those costs into a new order while the following launch is running, so the sort is off
    // the critical path.  The longest rays sit at the fractal's silhouette, which no static
    // order knows; with them first the frame ends when they do.  Tables:
    //   d_order      read by the launches      d_order_alt   written by the sort, then swapped in
    //   d_cost[0]    written by the first launch of a period, read by the sort
    // Events order everything whichever streams the caller uses.  Off for small frames, where it
    // does not pay for itself.
    // KIFS frames gain from it only when they are large (8K: 2.58 -> 2.23 ms; 1080p: nothing).
    const bool is_kifs = c->options.fractal_group_id == uint32_t(kifs::GROUP_KIFS);
    // (the bunny's quad kernel records no costs)
    const bool records_costs = !(is_kifs && c->options.primitive_id == uint32_t(kifs::PRIM_BUNNY));
    const bool use_feedback = tt->feedback && tile_feedback_mode() != 0 && records_costs &&
                              tt->count >= ((is_kifs && tile_feedback_mode() < 2) ? 16384u : 2048u);
    // The order is refreshed every FEEDBACK_PERIOD launches (views change slowly; the events the
    // refresh needs cost a few microseconds each).  Within a period of launches k = 0..P-1:
    //   k == 0: record costs, event;   k == 1: sort the costs of launch 0 on the side stream;
    //   k == 2: adopt the new order (wait for the sort);   otherwise: a plain launch.
    static const uint64_t FEEDBACK_PERIOD = [] {
        const int v = tuning_knob("KIFS_FEEDBACK_PERIOD");
        return uint64_t(v < 0 ? 4 : v < 3 ? 3 : v);
    }();
    if (use_feedback && tt->last_stream && tt->last_stream != stream) {
        // The caller moved to another stream: order this stream after the launches of the old
        // one, so that the buffer rotation below keeps its "nobody still reads it" guarantee.
        if (!hip_ok(hipEventRecord(tt->rendered[1], tt->last_stream), "record(stream change)") ||
            !hip_ok(hipStreamWaitEvent(stream, tt->rendered[1], 0), "wait(stream change)"))
            return KIFS_ERR_RUNTIME;
    }
    if (use_feedback) tt->last_stream = stream;
    // a batch is launched with the sort in its own stream and refreshes every third launch (its
    // launches are long and its views move: an orbit; measured best for fixed and moving cameras;
    // KIFS_BATCH_PERIOD overrides)
    static const uint64_t BATCH_PERIOD = [] {
        const int v = tuning_knob("KIFS_BATCH_PERIOD");
        return uint64_t(v < 0 ? 3 : v < 2 ? 2 : v);
    }();
    const uint64_t period = count > 1 ? BATCH_PERIOD : FEEDBACK_PERIOD;
    const uint64_t k = use_feedback ? tt->launches % period : 0;
    // With several frames in flight (several contexts and streams on one device) the sort runs
    // in the launch stream itself: streams share a handful of hardware queues, and an event wait
    // parked in a queue also holds up whatever other context's launches sit behind it (measured:
    // two contexts fell back to running one after the other).  The 10 us then hide behind the
    // other frames' kernels.  A lone frame keeps the side stream: there nothing else can.
    const bool inline_sort = c->frames_in_flight > 1 || count > 1;
    if (use_feedback && tt->sort_pending && (inline_sort || k != 2)) {
        // A side-stream sort from earlier launches still owns d_cost[0] and d_order_alt -- lone launches
        // before a batch, or a period cut short when feedback was switched off in between (options
        // changed to a pipeline without it and back).  Take its result before anything here records
        // costs or sorts again: the sort reads cost[] twice and must not see it change.
        if (!hip_ok(hipStreamWaitEvent(stream, tt->sorted, 0), "wait(sorted)")) return KIFS_ERR_RUNTIME;
        std::swap(tt->d_order, tt->d_order_alt);
        tt->sort_pending = false;
    }
    if (use_feedback && inline_sort && k == 1) {
        const uint32_t tiles_x = uint32_t((P.width + kifs::TILE_W - 1) / kifs::TILE_W);
        if (!hip_ok(kifs::launch_tile_order(tt->d_cost[0], tt->d_order_alt, tt->count, tiles_x, tt->cost_shift, stream),
                    "tile_order_kernel launch"))
            return KIFS_ERR_RUNTIME;
        std::swap(tt->d_order, tt->d_order_alt);  // stream order: the sort precedes this launch
    }
    if (use_feedback && k == 2 && tt->sort_pending) {  // adopt the order the side stream prepared
        if (!hip_ok(hipStreamWaitEvent(stream, tt->sorted, 0), "wait(sorted)")) return KIFS_ERR_RUNTIME;
        std::swap(tt->d_order, tt->d_order_alt);
        tt->sort_pending = false;
    }
    const bool record_costs = use_feedback && k == 0;
    P.tile_order = tt->d_order;
    P.tile_count = tt->count;
    P.tile_cost = record_costs ? tt->d_cost[0] : nullptr;
    if (P.counters) P.round_steps = 0;  // the per-wave diagnostics belong to the one-wave-per-block march
    // ---- launch shape.  Everything below is decided from `load`: the launch's tiles that can hold rays
    // with real work (the projected bounding sphere's tiles, all views), tools/cliff_sweep.py's x axis.
    const uint32_t group_id = c->options.fractal_group_id;
    const bool lone = count == 1 && c->frames_in_flight <= 1;
    const bool bunny_scene = group_id == uint32_t(kifs::GROUP_KIFS) && c->options.primitive_id == uint32_t(kifs::PRIM_BUNNY);
    const double heavy_tiles = disc_tiles(P, h, tt->count);
    const double load = heavy_tiles * double(count);
    // the residency cap serves a lone frame's latency; concurrent frames want every slot
    P.workgroups_per_cu = lone ? residency_for(P, group_id, heavy_tiles) : 0;
    // a residency-capped launch is a lone frame bound by its longest rays: re-queuing helps
    // throughput, not that (1080p Julia: 0.143 ms without, 0.146 ms with)
    if (P.workgroups_per_cu >= 1) P.round_steps = 0;
    // (an uncapped lone Julia frame -- 4096^2 -- prefers longer rounds: 0.430 ms at 32 steps, 0.445 at 16)
    if (P.round_steps == 16 && count == 1 && group_id == uint32_t(kifs::GROUP_JULIA) && P.max_iterations >= 64 &&
        tuning_knob("KIFS_ROUND_STEPS") < 0)
        P.round_steps = 32;
    // (nor does the lone bunny frame: 0.461 ms with the quad kernel, 0.670 ms in rounds)
    if (bunny_scene && count == 1) P.round_steps = 0;
    // nor does a launch too small to fill the device twice over (256x256 x 8 views = 2048
    // workgroups: 0.038 ms without, 0.062 ms with)
    if (uint64_t(tt->count) * uint64_t(count) < 4096u) P.round_steps = 0;
    {   // Shape of the re-queuing path (profiles/r02/sweep_shapes.jsonl: 5 frame sizes x 4 camera
        // distances x 2 scenes x batches of 1 / 8 / 32, every shape forced in turn):
        //   one WAVE per tile (render_wave_kernel) once the launch has several times more heavy tiles
        //     than the device has workgroup slots -- then slots, not critical paths, set its duration, and
        //     single-wave workgroups give four times as many (1080p Julia x32: 1.13 -> 0.88 ms; 4096^2 x8
        //     +27 %; 8K Sierpinski x4 +16 %) -- from a load of 16 000 tiles for the Julia pipeline, 32 000
        //     for the others, 30 000 for a lone frame (all its heavy tiles are one view's);
        //   otherwise 256-thread workgroups (render_group_kernel), whose four waves take a tile's first,
        //     crowded rounds side by side (a lone wave needs +30 % for the same tile): TWO tiles of the cost
        //     order per workgroup when the launch is a batch with enough heavy tiles to pair (one tile's
        //     queue is short for most of its life, neighbours of the cost order fill each other's waves:
        //     batched 1080p Julia 0.319 -> 0.281 ms; below 3 500 heavy tiles pairing only halves the
        //     workgroups that can run side by side: 720p x8 at distance 5, 0.222 -> 0.188 ms with one) or a
        //     big lone KIFS frame (1440p Sierpinski at distance 2: -11 %), else ONE.
        // Not the bunny (four lanes per ray, 216 VGPRs: pairs just run longer); the generalised Julia pairs
        // tiles only from 12 000 heavy tiles (1080p x32: 0.140 -> 0.125 ms per frame; x8: nothing, and its
        // few, very long workgroups lost 7 % when paired on smaller launches) and keeps 256-thread
        // workgroups throughout (one wave per tile: x32 0.150 ms, x8 0.31 against 0.22).
        static const int forced = tuning_knob("KIFS_GROUP_TILES");
        const bool julia = group_id == uint32_t(kifs::GROUP_JULIA);
        const bool genjulia = group_id == uint32_t(kifs::GROUP_GENJULIA);
        const bool kifs_scene = group_id == uint32_t(kifs::GROUP_KIFS);
        const double wave_from = lone ? 30000.0 : (julia ? 16000.0 : 32000.0);
        int shape = 1;
        if (load >= wave_from && !genjulia) shape = 0;
        else if (!lone && load >= (genjulia ? 12000.0 : 3500.0)) shape = 2;
        else if (lone && kifs_scene && load >= 12000.0) shape = 2;
        if (forced >= 0) shape = forced;
        if (bunny_scene) shape = 1;
        P.group_tiles = shape;
    }
    const bool timed = c->profiling && !c->prof_a.empty() && (c->prof_seen++ % uint64_t(c->prof_every)) == 0;
    const size_t pslot = c->prof_count % (c->prof_a.empty() ? 1 : c->prof_a.size());
    if (timed && !hip_ok(hipEventRecord(c->prof_a[pslot], stream), "record(profile start)")) return KIFS_ERR_RUNTIME;
    if (record_costs) {
        // render_kernel / render_group_kernel record run times in units of 1024 cycles; the stream kernel
        // sums the march steps of a tile's long rays over the batch's views: scale to the sort's 1024 bins
        tt->cost_shift = 0;
    }
    c->last_round_steps = P.round_steps;
    c->last_group_tiles = P.round_steps > 0 ? P.group_tiles : -1;
    if (big) {
        if (!hip_ok(hipMemcpyAsync(c->d_views[vs], c->h_views[vs], sizeof(kifs::BatchView) * size_t(count),
                                   hipMemcpyHostToDevice, stream), "copy(view table)"))
            return KIFS_ERR_RUNTIME;
        B.table = c->d_views[vs];
    }
    hipError_t e = kifs::launch_render(B, c->options.fractal_group_id, c->options.primitive_id,
                                       stream);
    if (!hip_ok(e, "render_kernel launch")) return KIFS_ERR_RUNTIME;
    if (big) {
        if (!hip_ok(hipEventRecord(c->views_used[vs], stream), "record(view table)")) return KIFS_ERR_RUNTIME;
        c->views_busy[vs] = true;
    }
    if (timed) {
        if (!hip_ok(hipEventRecord(c->prof_b[pslot], stream), "record(profile stop)")) return KIFS_ERR_RUNTIME;
        ++c->prof_count;
    }
    if (!use_feedback) {  // no bookkeeping, no events: nothing depends on this launch
        tt->launches = 0;  // (a pending side-stream sort stays pending: the next feedback launch waits for it)
        return KIFS_OK;
    }
    tt->launches += 1;
    if (inline_sort) return KIFS_OK;
    if (record_costs) {
        // Launch k = 0 of the period wrote d_cost[0].  The previous sort (period before) read it
        // and finished before that period's launch 2 started, i.e. long ago on this timeline.
        if (!hip_ok(hipEventRecord(tt->rendered[0], stream), "record(render)")) return KIFS_ERR_RUNTIME;
    } else if (k == 1) {
        // sort those costs into d_order_alt: the buffer last read by launches of the period
        // before the previous adoption, all of which precede launch 0 of this period
        if (!c->side_stream &&
            !hip_ok(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking), "side stream"))
            return KIFS_ERR_RUNTIME;
        const uint32_t tiles_x = uint32_t((P.width + kifs::TILE_W - 1) / kifs::TILE_W);
        if (!hip_ok(hipStreamWaitEvent(c->side_stream, tt->rendered[0], 0), "wait(render 0)") ||
            !hip_ok(kifs::launch_tile_order(tt->d_cost[0], tt->d_order_alt, tt->count, tiles_x, tt->cost_shift,
                                            c->side_stream), "tile_order_kernel launch") ||
            !hip_ok(hipEventRecord(tt->sorted, c->side_stream), "record(sorted)"))
            return KIFS_ERR_RUNTIME;
        tt->sort_pending = true;
    }
    return KIFS_OK;
}

int enqueue(kifs_ctx* c, hipStream_t stream, uint8_t* dev_out, size_t pitch, int y0, int y1,
            int encode) {
    if (!c->have_camera) return KIFS_ERR_UNCONFIGURED;
    if (!dev_out) return KIFS_ERR_BAD_ARG;
    return enqueue_batch(c, stream, 1, nullptr, &dev_out, pitch, y0, y1, encode);
}

}  // namespace

extern "C" {

int kifs_abi_version(void) { return KIFS_ABI_VERSION; }

const char* kifs_strerror(int status) {
    switch (status) {
    case KIFS_OK: return "ok";
    case KIFS_ERR_NO_DEVICE: return "no such HIP device";
    case KIFS_ERR_DEVICE_INIT: return "HIP device initialisation failed";
    case KIFS_ERR_BAD_SIZE: return "bad frame size or pitch";
    case KIFS_ERR_UNCONFIGURED: return "render before screen, camera and options were set";
    case KIFS_ERR_RUNTIME: return "HIP runtime error";
    case KIFS_ERR_COMM: return "inter-GPU transfer failed (peer copy over xGMI)";
    case KIFS_ERR_BAD_ARG: return "bad argument";
    default: return "unknown status";
    }
}

kifs_ctx* kifs_create(int device_ordinal, int* status) {
    auto fail = [&](int st) -> kifs_ctx* {
        if (status) *status = st;
        return nullptr;
    };
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(KIFS_ERR_NO_DEVICE);
    }
    if (device_ordinal < 0 || device_ordinal >= ndev) return fail(KIFS_ERR_NO_DEVICE);
    DeviceGuard g(device_ordinal);
    if (!g.ok) return fail(KIFS_ERR_DEVICE_INIT);
    kifs_ctx* c = new (std::nothrow) kifs_ctx();
    if (!c) return fail(KIFS_ERR_DEVICE_INIT);
    c->device = device_ordinal;
    float* table = c->h_srgb;
    kifs::build_srgb_thresholds(table);
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&c->ev_start) == hipSuccess &&
              hipEventCreate(&c->ev_stop) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&c->d_srgb), sizeof c->h_srgb) == hipSuccess &&
              hipMemcpy(c->d_srgb, table, sizeof c->h_srgb, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        kifs_destroy(c);
        return fail(KIFS_ERR_DEVICE_INIT);
    }
    if (status) *status = KIFS_OK;
    return c;
}

void kifs_destroy(kifs_ctx* c) {
    if (!c) return;
    DeviceGuard g(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->side_stream) {
        (void)hipStreamSynchronize(c->side_stream);
        (void)hipStreamDestroy(c->side_stream);
    }
    for (auto& t : c->tables)
        if (t.d_order) free_table(t);
    for (RowTable* r : c->row_tables) {
        if (r->d_rows) (void)hipFree(r->d_rows);
        delete r;
    }
    for (hipEvent_t ev : c->prof_a) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : c->prof_b) if (ev) (void)hipEventDestroy(ev);
    if (c->d_counters) (void)hipFree(c->d_counters);
    for (int i = 0; i < kifs_ctx::VIEW_RING; ++i) {
        if (c->views_used[i]) {
            (void)hipEventSynchronize(c->views_used[i]);
            (void)hipEventDestroy(c->views_used[i]);
        }
        if (c->d_views[i]) (void)hipFree(c->d_views[i]);
        if (c->h_views[i]) (void)hipHostFree(c->h_views[i]);
    }
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->d_srgb) (void)hipFree(c->d_srgb);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int kifs_set_screen(kifs_ctx* c, const KifsScreenUniform* s) {
    if (!c || !s) return KIFS_ERR_BAD_ARG;
    kifs_ctx probe;
    probe.screen = *s;
    int w, h;
    int st = frame_dims(&probe, &w, &h);
    if (st != KIFS_OK) return st;  // cf. render.rs:211: zero-sized frames are ignored
    c->screen = *s;
    c->have_screen = true;
    return KIFS_OK;
}

int kifs_set_camera(kifs_ctx* c, const KifsCameraUniform* cam) {
    if (!c || !cam) return KIFS_ERR_BAD_ARG;
    c->camera = *cam;
    c->have_camera = true;
    return KIFS_OK;
}

int kifs_set_options(kifs_ctx* c, const KifsOptionsUniform* o) {
    if (!c || !o) return KIFS_ERR_BAD_ARG;
    if (o->fractal_group_id > 2u) return KIFS_ERR_BAD_ARG;
    c->options = *o;
    c->have_options = true;
    return KIFS_OK;
}

int kifs_set_iters(kifs_ctx* c, int sdf_iters, int normal_iters, int fold_iters) {
    if (!c || sdf_iters < 0 || normal_iters < 0 || fold_iters < 0) return KIFS_ERR_BAD_ARG;
    c->sdf_iters = sdf_iters;
    c->normal_iters = normal_iters;
    c->fold_iters = fold_iters;
    return KIFS_OK;
}

int kifs_set_extensions(kifs_ctx* c, const KifsExtensions* ext) {
    if (!c || !ext) return KIFS_ERR_BAD_ARG;
    if (ext->soft_shadow && ext->shadow_steps < 0) return KIFS_ERR_BAD_ARG;
    c->ext = *ext;
    return KIFS_OK;
}

int kifs_band_range(int height, int rank, int world, int* y0, int* y1) {
    if (height < 0 || world <= 0 || rank < 0 || rank >= world || !y0 || !y1)
        return KIFS_ERR_BAD_ARG;
    const long long h = height;
    *y0 = int(h * rank / world);
    *y1 = int(h * (rank + 1) / world);
    return KIFS_OK;
}

int kifs_shard_stripes(int height, int world, const int* weights, int rank, int* stripes, int max_stripes,
                       int* n_stripes, int* rows) {
    if (height < 0 || world <= 0 || world > 1024 || rank < 0 || rank >= world || !n_stripes) return KIFS_ERR_BAD_ARG;
    long long total = 0;
    for (int r = 0; r < world; ++r) {
        const int w = weights ? weights[r] : 1;
        if (w < 0 || w > (1 << 20)) return KIFS_ERR_BAD_ARG;
        total += w;
    }
    if (total <= 0) return KIFS_ERR_BAD_ARG;
    // Smooth weighted round robin: every stripe goes to the rank with the largest running credit;
    // equal weights deal 0, 1, .., world-1, 0, 1, ..; a rank of weight w gets w stripes in every
    // `total`, spread evenly through the frame (the expensive rows sit in its middle).
    std::vector<long long> credit(static_cast<size_t>(world), 0);
    const int all = (height + KIFS_STRIPE_ROWS - 1) / KIFS_STRIPE_ROWS;
    int n = 0, nrows = 0;
    for (int s = 0; s < all; ++s) {
        int best = 0;
        for (int r = 0; r < world; ++r) {
            credit[size_t(r)] += weights ? weights[r] : 1;
            if (credit[size_t(r)] > credit[size_t(best)]) best = r;
        }
        credit[size_t(best)] -= total;
        if (best != rank) continue;
        if (stripes) {
            if (n >= max_stripes) return KIFS_ERR_BAD_ARG;
            stripes[n] = s;
        }
        ++n;
        nrows += std::min(KIFS_STRIPE_ROWS, height - s * KIFS_STRIPE_ROWS);
    }
    *n_stripes = n;
    if (rows) *rows = nrows;
    return KIFS_OK;
}

int kifs_render_shard_async(kifs_ctx* c, void* hip_stream, int count, const KifsCameraUniform* cameras,
                            uint8_t* const* dev_outs, size_t pitch, const int* stripes, int n_stripes,
                            int in_place, int encode) {
    if (!c || !dev_outs || !stripes || n_stripes < 0 || (!cameras && count != 1)) return KIFS_ERR_BAD_ARG;
    if (!c->have_screen) return KIFS_ERR_UNCONFIGURED;
    DeviceGuard g(c->device);
    if (!g.ok) return KIFS_ERR_RUNTIME;
    int w, h;
    int st = frame_dims(c, &w, &h);
    if (st != KIFS_OK) return st;
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    return enqueue_batch(c, s, count, cameras, dev_outs, pitch, 0, h, encode, stripes, n_stripes, in_place);
}

int kifs_unpack_shard_async(kifs_ctx* c, void* hip_stream, int count, uint8_t* dev_frames, size_t frame_pitch,
                            size_t frame_stride, const uint8_t* dev_shards, size_t shard_pitch,
                            size_t shard_stride, const int* stripes, int n_stripes) {
    if (!c || !dev_frames || !dev_shards || !stripes || n_stripes < 0 || count < 0) return KIFS_ERR_BAD_ARG;
    if (!c->have_screen) return KIFS_ERR_UNCONFIGURED;
    DeviceGuard g(c->device);
    if (!g.ok) return KIFS_ERR_RUNTIME;
    int w, h;
    int st = frame_dims(c, &w, &h);
    if (st != KIFS_OK) return st;
    const size_t row_bytes = size_t(w) * 4;
    if (frame_pitch < row_bytes || shard_pitch < row_bytes || ((frame_pitch | shard_pitch | frame_stride | shard_stride) & 3u) ||
        ((reinterpret_cast<uintptr_t>(dev_frames) | reinterpret_cast<uintptr_t>(dev_shards)) & 3u))
        return KIFS_ERR_BAD_SIZE;
    if (n_stripes == 0 || count == 0) return KIFS_OK;
    const RowTable* rows = row_table(c, stripes, n_stripes, h);
    if (!rows) return KIFS_ERR_BAD_ARG;
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    return hip_ok(kifs::launch_unpack_stripes(dev_frames, frame_pitch, frame_stride, dev_shards, shard_pitch,
                                              shard_stride, rows->d_rows, n_stripes, count, w, h, s),
                  "unpack_stripes_kernel launch") ? KIFS_OK : KIFS_ERR_RUNTIME;
}

// ---- sparse shards ---------------------------------------------------------------------------
namespace {
// what the three entry points share: the frame's size, the stripes' row table, the background pixel
int sparse_setup(kifs_ctx* c, const int* stripes, int n_stripes, int encode, int* w, int* h, const RowTable** rows,
                 uint32_t* background) {
    if (!c->have_screen || (background && !c->have_options)) return KIFS_ERR_UNCONFIGURED;
    int st = frame_dims(c, w, h);
    if (st != KIFS_OK) return st;
    if (background) {
        if (encode != KIFS_ENCODE_UNORM && encode != KIFS_ENCODE_SRGB) return KIFS_ERR_BAD_ARG;
        const float* bc = c->options.background_color;
        *background = background_pixel(c, kifs::V3{bc[0], bc[1], bc[2]}, encode);
    }
    *rows = row_table(c, stripes, n_stripes, *h);
    return *rows ? KIFS_OK : KIFS_ERR_BAD_ARG;
}
}  // namespace

int kifs_pack_sparse_async(kifs_ctx* c, void* hip_stream, int count, const uint8_t* dev_shards, size_t shard_pitch,
                           size_t shard_stride, const int* stripes, int n_stripes, int encode, uint8_t* dev_records,
                           size_t capacity_records, uint32_t* dev_n_records, uint32_t* host_n_records) {
    if (!c || !dev_shards || !stripes || !dev_records || !dev_n_records || n_stripes < 0 || count < 0) return KIFS_ERR_BAD_ARG;
    DeviceGuard g(c->device);
    if (!g.ok) return KIFS_ERR_RUNTIME;
    int w = 0, h = 0;
    const RowTable* rows = nullptr;
    uint32_t background = 0;
    if (n_stripes == 0 || count == 0) {
        if (!c->have_screen || !c->have_options) return KIFS_ERR_UNCONFIGURED;
    } else {
        int st = sparse_setup(c, stripes, n_stripes, encode, &w, &h, &rows, &background);
        if (st != KIFS_OK) return st;
        const size_t tiles = size_t(count) * size_t(n_stripes) * size_t((w + kifs::TILE_W - 1) / kifs::TILE_W);
        if (shard_pitch < size_t(w) * 4 || ((shard_pitch | shard_stride) & 3u) || capacity_records < tiles ||
            tiles > 0xffffffffull || (reinterpret_cast<uintptr_t>(dev_shards) & 3u) ||
            (reinterpret_cast<uintptr_t>(dev_records) & 15u) || (reinterpret_cast<uintptr_t>(dev_n_records) & 3u))
            return KIFS_ERR_BAD_SIZE;
    }
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    if (!hip_ok(hipMemsetAsync(dev_n_records, 0, sizeof(uint32_t), s), "memset(n_records)")) return KIFS_ERR_RUNTIME;
    if (rows && !hip_ok(kifs::launch_pack_sparse(dev_shards, shard_pitch, shard_stride, rows->d_rows, n_stripes, count, w, h,
                                                 background, reinterpret_cast<uint32_t*>(dev_records), dev_n_records, s),
                        "pack_sparse_kernel launch"))
        return KIFS_ERR_RUNTIME;
    if (host_n_records &&
        !hip_ok(hipMemcpyAsync(host_n_records, dev_n_records, sizeof(uint32_t), hipMemcpyDeviceToHost, s), "copy(n_records)"))
        return KIFS_ERR_RUNTIME;
    return KIFS_OK;
}

namespace {
int unpack_or_erase(kifs_ctx* c, void* hip_stream, int count, uint8_t* dev_frames, size_t frame_pitch, size_t frame_stride,
                    const uint8_t* dev_records, size_t n_records, const int* stripes, int n_stripes, bool erase, int encode) {
    if (!c || !dev_frames || !stripes || n_stripes < 0 || count < 0 || (n_records && !dev_records)) return KIFS_ERR_BAD_ARG;
    DeviceGuard g(c->device);
    if (!g.ok) return KIFS_ERR_RUNTIME;
    if (!c->have_screen || (erase && !c->have_options)) return KIFS_ERR_UNCONFIGURED;
    if (n_stripes == 0 || count == 0 || n_records == 0) return KIFS_OK;
    int w = 0, h = 0;
    const RowTable* rows = nullptr;
    uint32_t background = 0;
    int st = sparse_setup(c, stripes, n_stripes, encode, &w, &h, &rows, erase ? &background : nullptr);
    if (st != KIFS_OK) return st;
    const size_t tiles = size_t(count) * size_t(n_stripes) * size_t((w + kifs::TILE_W - 1) / kifs::TILE_W);
    if (frame_pitch < size_t(w) * 4 || ((frame_pitch | frame_stride) & 3u) || n_records > tiles ||
        (reinterpret_cast<uintptr_t>(dev_frames) & 3u) || (reinterpret_cast<uintptr_t>(dev_records) & 15u))
        return KIFS_ERR_BAD_SIZE;
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    return hip_ok(kifs::launch_unpack_sparse(dev_frames, frame_pitch, frame_stride, reinterpret_cast<const uint32_t*>(dev_records),
                                             uint32_t(n_records), rows->d_rows, n_stripes, count, w, h, erase ? 1 : 0,
                                             background, s),
                  "unpack_sparse_kernel launch") ? KIFS_OK : KIFS_ERR_RUNTIME;
}
}  // namespace

int kifs_unpack_sparse_async(kifs_ctx* c, void* hip_stream, int count, uint8_t* dev_frames, size_t frame_pitch,
                             size_t frame_stride, const uint8_t* dev_records, size_t n_records, const int* stripes,
                             int n_stripes) {
    return unpack_or_erase(c, hip_stream, count, dev_frames, frame_pitch, frame_stride, dev_records, n_records, stripes,
                           n_stripes, false, 0);
}

int kifs_erase_sparse_async(kifs_ctx* c, void* hip_stream, int count, uint8_t* dev_frames, size_t frame_pitch,
                            size_t frame_stride, const uint8_t* dev_records, size_t n_records, const int* stripes,
                            int n_stripes, int encode) {
    return unpack_or_erase(c, hip_stream, count, dev_frames, frame_pitch, frame_stride, dev_records, n_records, stripes,
                           n_stripes, true, encode);
}

int kifs_fill_shard_async(kifs_ctx* c, void* hip_stream, int count, uint8_t* dev_frames, size_t frame_pitch,
                          size_t frame_stride, const int* stripes, int n_stripes, int encode) {
    if (!c || !dev_frames || !stripes || n_stripes < 0 || count < 0) return KIFS_ERR_BAD_ARG;
    DeviceGuard g(c->device);
    if (!g.ok) return KIFS_ERR_RUNTIME;
    if (!c->have_screen || !c->have_options) return KIFS_ERR_UNCONFIGURED;
    if (n_stripes == 0 || count == 0) return KIFS_OK;
    int w = 0, h = 0;
    const RowTable* rows = nullptr;
    uint32_t background = 0;
    int st = sparse_setup(c, stripes, n_stripes, encode, &w, &h, &rows, &background);
    if (st != KIFS_OK) return st;
    if (frame_pitch < size_t(w) * 4 || ((frame_pitch | frame_stride) & 3u) || (reinterpret_cast<uintptr_t>(dev_frames) & 3u))
        return KIFS_ERR_BAD_SIZE;
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    return hip_ok(kifs::launch_fill_stripes(dev_frames, frame_pitch, frame_stride, rows->d_rows, n_stripes, count, w, h,
                                            background, s),
                  "fill_stripes_kernel launch") ? KIFS_OK : KIFS_ERR_RUNTIME;
}

int kifs_render_async(kifs_ctx* c, void* hip_stream, uint8_t* dev_out, size_t pitch, int y0,
                      int y1, int encode) {
    if (!c) return KIFS_ERR_BAD_ARG;
    DeviceGuard g(c->device);
    if (!g.ok) return KIFS_ERR_RUNTIME;
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    return enqueue(c, s, dev_out, pitch, y0, y1, encode);
}

int kifs_render_batch_async(kifs_ctx* c, void* hip_stream, int count, const KifsCameraUniform* cameras,
                            uint8_t* const* dev_outs, size_t pitch, int y0, int y1, int encode) {
    if (!c || !cameras || !dev_outs) return KIFS_ERR_BAD_ARG;
    static_assert(KIFS_MAX_BATCH == kifs::MAX_BATCH, "header and kernels agree on the batch limit");
    DeviceGuard g(c->device);
    if (!g.ok) return KIFS_ERR_RUNTIME;
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    return enqueue_batch(c, s, count, cameras, dev_outs, pitch, y0, y1, encode);
}

int kifs_render(kifs_ctx* c, uint8_t* out, size_t pitch, int y0, int y1, int encode) {
    if (!c || !out) return KIFS_ERR_BAD_ARG;
    DeviceGuard g(c->device);
    if (!g.ok) return KIFS_ERR_RUNTIME;
    const bool on_device = is_device_pointer(out);
    uint8_t* target = out;
    size_t tpitch = pitch;
    if (!on_device) {
        if (!c->have_screen) return KIFS_ERR_UNCONFIGURED;
        int w, h;
        int st = frame_dims(c, &w, &h);
        if (st != KIFS_OK) return st;
        if (y0 < 0 || y1 > h || y0 > y1) return KIFS_ERR_BAD_ARG;
        if (pitch < size_t(w) * 4) return KIFS_ERR_BAD_SIZE;
        tpitch = size_t(w) * 4;
        size_t need = tpitch * size_t(y1 - y0);
        if (need > c->scratch_bytes) {
            if (c->d_scratch) (void)hipFree(c->d_scratch);
            c->d_scratch = nullptr;
            c->scratch_bytes = 0;
            if (!hip_ok(hipMalloc(reinterpret_cast<void**>(&c->d_scratch), need), "hipMalloc(scratch)"))
                return KIFS_ERR_RUNTIME;
            c->scratch_bytes = need;
        }
        target = c->d_scratch;
        if (need == 0) return KIFS_OK;
    }
    if (hipEventRecord(c->ev_start, c->stream) != hipSuccess) return KIFS_ERR_RUNTIME;
    int st = enqueue(c, c->stream, target, tpitch, y0, y1, encode);
    if (st != KIFS_OK) return st;
    if (hipEventRecord(c->ev_stop, c->stream) != hipSuccess) return KIFS_ERR_RUNTIME;
    if (!on_device) {
        if (hipMemcpy2DAsync(out, pitch, target, tpitch, tpitch, size_t(y1 - y0),
                             hipMemcpyDeviceToHost, c->stream) != hipSuccess)
            return KIFS_ERR_RUNTIME;
    }
    if (!hip_ok(hipStreamSynchronize(c->stream), "hipStreamSynchronize")) return KIFS_ERR_RUNTIME;
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->ev_start, c->ev_stop) == hipSuccess) c->last_ms = ms;
    return KIFS_OK;
}

double kifs_last_kernel_ms(kifs_ctx* c) { return c ? c->last_ms : -1.0; }

int kifs_debug_last_round_steps(kifs_ctx* c) { return c ? c->last_round_steps : -1; }

int kifs_debug_last_group_tiles(kifs_ctx* c) { return c ? c->last_group_tiles : -2; }

int kifs_set_frames_in_flight(kifs_ctx* c, int n) {
    if (!c || n < 1) return KIFS_ERR_BAD_ARG;
    c->frames_in_flight = n;
    return KIFS_OK;
}

int kifs_set_profiling(kifs_ctx* c, int enable) {
    if (!c) return KIFS_ERR_BAD_ARG;
    DeviceGuard g(c->device);
    if (enable && c->prof_a.empty()) {
        constexpr size_t RING = 4096;
        c->prof_a.assign(RING, nullptr);
        c->prof_b.assign(RING, nullptr);
        for (size_t i = 0; i < RING; ++i)
            if (hipEventCreate(&c->prof_a[i]) != hipSuccess || hipEventCreate(&c->prof_b[i]) != hipSuccess)
                return KIFS_ERR_RUNTIME;
    }
    c->profiling = enable != 0;
    c->prof_every = enable > 1 ? enable : 1;
    c->prof_seen = 0;
    c->prof_count = 0;
    return KIFS_OK;
}

int kifs_profile_read(kifs_ctx* c, int* launches, double* mean_ms, double* min_ms, double* max_ms) {
    if (!c || !launches) return KIFS_ERR_BAD_ARG;
    DeviceGuard g(c->device);
    if (hipDeviceSynchronize() != hipSuccess) return KIFS_ERR_RUNTIME;
    const size_t ring = c->prof_a.size();
    const size_t n = ring ? std::min(c->prof_count, ring) : 0;
    double sum = 0.0, lo = 1e300, hi = 0.0;
    size_t ok = 0;
    for (size_t i = 0; i < n; ++i) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->prof_a[i], c->prof_b[i]) == hipSuccess) {
            sum += ms; lo = std::min(lo, double(ms)); hi = std::max(hi, double(ms)); ++ok;
        }
    }
    *launches = int(ok);
    if (mean_ms) *mean_ms = ok ? sum / double(ok) : -1.0;
    if (min_ms) *min_ms = ok ? lo : -1.0;
    if (max_ms) *max_ms = ok ? hi : -1.0;
    c->prof_count = 0;
    return KIFS_OK;
}

int kifs_synchronize(kifs_ctx* c) {
    if (!c) return KIFS_ERR_BAD_ARG;
    DeviceGuard g(c->device);
    return hipStreamSynchronize(c->stream) == hipSuccess ? KIFS_OK : KIFS_ERR_RUNTIME;
}

// Diagnostics buffer: 8 aggregate counters followed by one 4-word record per wave of the
// largest frame the context can be asked for with the current screen (4 waves per tile).
static size_t debug_words(const kifs_ctx* c) {
    int w = 0, h = 0;
    if (!c->have_screen || frame_dims(c, &w, &h) != KIFS_OK) return 8;
    size_t tiles = size_t((w + kifs::TILE_W - 1) / kifs::TILE_W) * size_t((h + kifs::TILE_H - 1) / kifs::TILE_H);
    return 8 + 16 * tiles;
}

int kifs_debug_counters(kifs_ctx* c, int enable, unsigned long long out[8]) {
    if (!c) return KIFS_ERR_BAD_ARG;
    DeviceGuard g(c->device);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return KIFS_ERR_RUNTIME;
    if (c->d_counters && out &&
        hipMemcpy(out, c->d_counters, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
        return KIFS_ERR_RUNTIME;
    const size_t words = debug_words(c);
    if (enable && (!c->d_counters || c->counter_words != words)) {
        if (c->d_counters) (void)hipFree(c->d_counters);
        c->d_counters = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&c->d_counters), words * sizeof(unsigned long long)) != hipSuccess)
            return KIFS_ERR_RUNTIME;
        c->counter_words = words;
    }
    if (c->d_counters &&
        hipMemset(c->d_counters, 0, c->counter_words * sizeof(unsigned long long)) != hipSuccess)
        return KIFS_ERR_RUNTIME;
    if (!enable && c->d_counters) {
        (void)hipFree(c->d_counters);
        c->d_counters = nullptr;
        c->counter_words = 0;
    }
    return KIFS_OK;
}

int kifs_debug_set_tile_order(kifs_ctx* c, const uint32_t* order, size_t count) {
    if (!c || !order) return KIFS_ERR_BAD_ARG;
    int w, h;
    if (!c->have_screen || frame_dims(c, &w, &h) != KIFS_OK) return KIFS_ERR_UNCONFIGURED;
    DeviceGuard g(c->device);
    TileTable* tt = tile_table(c, w, h, 0, h);
    if (!tt) return KIFS_ERR_RUNTIME;
    if (count != tt->count) return KIFS_ERR_BAD_ARG;
    tt->feedback = false;  // a pinned order stays until the table is rebuilt
    std::vector<char> seen(count, 0);  // must be a permutation of the frame's tiles
    const uint32_t tx = uint32_t((w + kifs::TILE_W - 1) / kifs::TILE_W);
    const uint32_t ty = uint32_t((h + kifs::TILE_H - 1) / kifs::TILE_H);
    for (size_t i = 0; i < count; ++i) {
        uint32_t x = order[i] & 0xffffu, y = order[i] >> 16;
        if (x >= tx || y >= ty || seen[size_t(y) * tx + x]) return KIFS_ERR_BAD_ARG;
        seen[size_t(y) * tx + x] = 1;
    }
    if (hipDeviceSynchronize() != hipSuccess) return KIFS_ERR_RUNTIME;
    return hipMemcpy(tt->d_order, order, count * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess
               ? KIFS_OK : KIFS_ERR_RUNTIME;
}

int kifs_debug_get_tile_order(kifs_ctx* c, uint32_t* order, size_t max_count, size_t* count) {
    if (!c || !order || !count) return KIFS_ERR_BAD_ARG;
    int w, h;
    if (!c->have_screen || frame_dims(c, &w, &h) != KIFS_OK) return KIFS_ERR_UNCONFIGURED;
    DeviceGuard g(c->device);
    const TileTable* tt = tile_table(c, w, h, 0, h);
    if (!tt) return KIFS_ERR_RUNTIME;
    if (tt->count > max_count) return KIFS_ERR_BAD_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return KIFS_ERR_RUNTIME;
    *count = tt->count;
    return hipMemcpy(order, tt->d_order, tt->count * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess
               ? KIFS_OK : KIFS_ERR_RUNTIME;
}

int kifs_debug_wave_records(kifs_ctx* c, unsigned long long* out, size_t max_waves, size_t* n_waves) {
    if (!c || !out || !n_waves) return KIFS_ERR_BAD_ARG;
    if (!c->d_counters) return KIFS_ERR_UNCONFIGURED;
    DeviceGuard g(c->device);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return KIFS_ERR_RUNTIME;
    size_t waves = (c->counter_words - 8) / 4;
    if (waves > max_waves) waves = max_waves;
    if (hipMemcpy(out, c->d_counters + 8, waves * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
        return KIFS_ERR_RUNTIME;
    *n_waves = waves;
    return KIFS_OK;
}

int kifs_eval_points(kifs_ctx* c, const float* pts, int n, float* sdf_out, float* nrm_out) {
    if (!c || !pts || n < 0) return KIFS_ERR_BAD_ARG;
    if (!c->have_options) return KIFS_ERR_UNCONFIGURED;
    if (n == 0) return KIFS_OK;
    DeviceGuard g(c->device);
    kifs::FrameParams P;
    const KifsScreenUniform saved_screen = c->screen;  // options-only evaluation: any valid screen does
    c->screen = {1.0f, 1.0f, 1.0f};
    int st = fill_params(c, &P);
    c->screen = saved_screen;
    if (st != KIFS_OK) return st;
    float *d_pts = nullptr, *d_sdf = nullptr, *d_nrm = nullptr;
    int rc = KIFS_ERR_RUNTIME;
    size_t nb = size_t(n) * sizeof(float);
    do {
        if (hipMalloc(reinterpret_cast<void**>(&d_pts), 3 * nb) != hipSuccess) break;
        if (sdf_out && hipMalloc(reinterpret_cast<void**>(&d_sdf), nb) != hipSuccess) break;
        if (nrm_out && hipMalloc(reinterpret_cast<void**>(&d_nrm), 3 * nb) != hipSuccess) break;
        if (hipMemcpyAsync(d_pts, pts, 3 * nb, hipMemcpyHostToDevice, c->stream) != hipSuccess) break;
        if (kifs::launch_eval_points(P, c->options.fractal_group_id, c->options.primitive_id,
                                     d_pts, n, d_sdf, d_nrm, c->stream) != hipSuccess) break;
        if (sdf_out && hipMemcpyAsync(sdf_out, d_sdf, nb, hipMemcpyDeviceToHost, c->stream) != hipSuccess) break;
        if (nrm_out && hipMemcpyAsync(nrm_out, d_nrm, 3 * nb, hipMemcpyDeviceToHost, c->stream) != hipSuccess) break;
        if (hipStreamSynchronize(c->stream) != hipSuccess) break;
        rc = KIFS_OK;
    } while (0);
    if (d_pts) (void)hipFree(d_pts);
    if (d_sdf) (void)hipFree(d_sdf);
    if (d_nrm) (void)hipFree(d_nrm);
    return rc;
}

int kifs_eval_math(kifs_ctx* c, int fn, const float* in, float param, float* out, int n) {
    if (!c || !in || !out || n < 0 || fn < 0 || fn > 10) return KIFS_ERR_BAD_ARG;
    if (n == 0) return KIFS_OK;
    DeviceGuard g(c->device);
    float *d_in = nullptr, *d_out = nullptr;
    int rc = KIFS_ERR_RUNTIME;
    size_t nb = size_t(n) * sizeof(float);
    do {
        if (hipMalloc(reinterpret_cast<void**>(&d_in), nb) != hipSuccess) break;
        if (hipMalloc(reinterpret_cast<void**>(&d_out), nb) != hipSuccess) break;
        if (hipMemcpyAsync(d_in, in, nb, hipMemcpyHostToDevice, c->stream) != hipSuccess) break;
        if (kifs::launch_eval_math(fn, d_in, param, c->d_srgb, d_out, n, c->stream) != hipSuccess) break;
        if (hipMemcpyAsync(out, d_out, nb, hipMemcpyDeviceToHost, c->stream) != hipSuccess) break;
        if (hipStreamSynchronize(c->stream) != hipSuccess) break;
        rc = KIFS_OK;
    } while (0);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

}  // extern "C"

// ---- single-process multi-GPU ------------------------------------------------------------
struct kifs_multi {
    std::vector<kifs_ctx*> ctx;
    std::vector<int> dev;
    std::vector<int> weight;               // share of each device (kifs_shard_stripes weights)
    std::vector<std::vector<int>> stripes; // the shard of each device for the current frame height
    std::vector<int> rows;
    int stripes_height = -1;
    std::vector<uint8_t*> shard;           // per-device packed shard buffer (on that device; non-root)
    std::vector<size_t> shard_bytes;
    std::vector<uint8_t*> recv;            // the same shards after the peer copy (on the root device)
    std::vector<size_t> recv_bytes;
    std::vector<hipEvent_t> ev0, ev1;      // kernel start/stop on each device's stream
    std::vector<double> shard_ms;
    uint8_t* root_frame = nullptr;         // staging frame on the root when the destination is host memory
    size_t root_frame_bytes = 0;
};

namespace {

bool grow(uint8_t*& buf, size_t& have, size_t need, const char* what) {
    if (need <= have) return true;
    if (buf) (void)hipFree(buf);
    buf = nullptr;
    have = 0;
    if (!hip_ok(hipMalloc(reinterpret_cast<void**>(&buf), need), what)) return false;
    have = need;
    return true;
}

// (Re)deal the frame's stripes to the devices.
int multi_partition(kifs_multi* m, int h) {
    if (m->stripes_height == h) return KIFS_OK;
    const int n = int(m->ctx.size());
    const int all = (h + KIFS_STRIPE_ROWS - 1) / KIFS_STRIPE_ROWS;
    for (int i = 0; i < n; ++i) {
        m->stripes[size_t(i)].assign(size_t(all), 0);
        int count = 0, rows = 0;
        int st = kifs_shard_stripes(h, n, m->weight.data(), i, m->stripes[size_t(i)].data(), all, &count, &rows);
        if (st != KIFS_OK) return st;
        m->stripes[size_t(i)].resize(size_t(count));
        m->rows[size_t(i)] = rows;
    }
    m->stripes_height = h;
    return KIFS_OK;
}

}  // namespace

extern "C" {

void kifs_multi_destroy(kifs_multi* m) {
    if (!m) return;
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        if (!m->ctx[i]) continue;
        DeviceGuard g(m->dev[i]);
        (void)hipStreamSynchronize(m->ctx[i]->stream);
        if (i < m->shard.size() && m->shard[i]) (void)hipFree(m->shard[i]);
        if (i < m->ev0.size() && m->ev0[i]) (void)hipEventDestroy(m->ev0[i]);
        if (i < m->ev1.size() && m->ev1[i]) (void)hipEventDestroy(m->ev1[i]);
    }
    if (!m->dev.empty()) {
        DeviceGuard g(m->dev[0]);
        for (uint8_t* r : m->recv)
            if (r) (void)hipFree(r);
        if (m->root_frame) (void)hipFree(m->root_frame);
    }
    for (kifs_ctx* c : m->ctx) kifs_destroy(c);
    delete m;
}

kifs_multi* kifs_multi_create(const int* devices, int n, int* status) {
    auto fail = [&](int st, kifs_multi* m) -> kifs_multi* {
        if (status) *status = st;
        kifs_multi_destroy(m);
        return nullptr;
    };
    if (!devices || n <= 0 || n > 64) return fail(KIFS_ERR_BAD_ARG, nullptr);
    kifs_multi* m = new (std::nothrow) kifs_multi();
    if (!m) return fail(KIFS_ERR_DEVICE_INIT, nullptr);
    const size_t N = size_t(n);
    m->weight.assign(N, 1);
    m->stripes.assign(N, {});
    m->rows.assign(N, 0);
    m->shard.assign(N, nullptr);
    m->shard_bytes.assign(N, 0);
    m->recv.assign(N, nullptr);
    m->recv_bytes.assign(N, 0);
    m->ev0.assign(N, nullptr);
    m->ev1.assign(N, nullptr);
    m->shard_ms.assign(N, -1.0);
    for (int i = 0; i < n; ++i) {
        int st = KIFS_OK;
        kifs_ctx* c = kifs_create(devices[i], &st);
        if (!c) return fail(st, m);
        m->ctx.push_back(c);
        m->dev.push_back(devices[i]);
        DeviceGuard g(devices[i]);
        if (hipEventCreate(&m->ev0[size_t(i)]) != hipSuccess || hipEventCreate(&m->ev1[size_t(i)]) != hipSuccess)
            return fail(KIFS_ERR_DEVICE_INIT, m);
        if (devices[i] != devices[0]) {  // direct xGMI access both ways; failure only means staged copies
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[i], devices[0]) == hipSuccess && can)
                (void)hipDeviceEnablePeerAccess(devices[0], 0);
            (void)hipGetLastError();
        }
    }
    if (status) *status = KIFS_OK;
    return m;
}

#define KIFS_MULTI_FORWARD(call)                 \
    if (!m) return KIFS_ERR_BAD_ARG;             \
    for (kifs_ctx* c : m->ctx) {                 \
        int st = (call);                         \
        if (st != KIFS_OK) return st;            \
    }                                            \
    return KIFS_OK;

int kifs_multi_set_screen(kifs_multi* m, const KifsScreenUniform* s) { KIFS_MULTI_FORWARD(kifs_set_screen(c, s)) }
int kifs_multi_set_camera(kifs_multi* m, const KifsCameraUniform* cam) { KIFS_MULTI_FORWARD(kifs_set_camera(c, cam)) }
int kifs_multi_set_options(kifs_multi* m, const KifsOptionsUniform* o) { KIFS_MULTI_FORWARD(kifs_set_options(c, o)) }
int kifs_multi_set_iters(kifs_multi* m, int a, int b, int f) { KIFS_MULTI_FORWARD(kifs_set_iters(c, a, b, f)) }
int kifs_multi_set_extensions(kifs_multi* m, const KifsExtensions* e) { KIFS_MULTI_FORWARD(kifs_set_extensions(c, e)) }

int kifs_multi_set_weights(kifs_multi* m, const int* weights) {
    if (!m) return KIFS_ERR_BAD_ARG;
    long long total = 0;
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        const int w = weights ? weights[i] : 1;
        if (w < 0 || w > (1 << 20)) return KIFS_ERR_BAD_ARG;
        total += w;
    }
    if (total <= 0) return KIFS_ERR_BAD_ARG;
    for (size_t i = 0; i < m->ctx.size(); ++i) m->weight[i] = weights ? weights[i] : 1;
    m->stripes_height = -1;
    return KIFS_OK;
}

int kifs_multi_shard(kifs_multi* m, int i, int* device, int* n_stripes, int* rows) {
    if (!m || i < 0 || size_t(i) >= m->ctx.size()) return KIFS_ERR_BAD_ARG;
    int w, h;
    if (!m->ctx[0]->have_screen) return KIFS_ERR_UNCONFIGURED;
    int st = frame_dims(m->ctx[0], &w, &h);
    if (st != KIFS_OK) return st;
    st = multi_partition(m, h);
    if (st != KIFS_OK) return st;
    if (device) *device = m->dev[size_t(i)];
    if (n_stripes) *n_stripes = int(m->stripes[size_t(i)].size());
    if (rows) *rows = m->rows[size_t(i)];
    return KIFS_OK;
}

double kifs_multi_shard_ms(kifs_multi* m, int i) {
    return (m && i >= 0 && size_t(i) < m->shard_ms.size()) ? m->shard_ms[size_t(i)] : -1.0;
}

int kifs_multi_render(kifs_multi* m, uint8_t* out, size_t pitch, int encode) {
    if (!m || !out) return KIFS_ERR_BAD_ARG;
    kifs_ctx* root = m->ctx[0];
    if (!root->have_screen || !root->have_camera || !root->have_options) return KIFS_ERR_UNCONFIGURED;
    int w, h;
    int st = frame_dims(root, &w, &h);
    if (st != KIFS_OK) return st;
    const size_t row_bytes = size_t(w) * 4;
    if (pitch < row_bytes || (pitch & 3u)) return KIFS_ERR_BAD_SIZE;
    st = multi_partition(m, h);
    if (st != KIFS_OK) return st;
    const int n = int(m->ctx.size());
    // the frame the shards are collected into: the caller's buffer if it is root-device memory
    uint8_t* frame = out;
    size_t fpitch = pitch;
    bool host_dst;
    {
        DeviceGuard g(m->dev[0]);
        host_dst = !is_device_pointer(out);
        if (host_dst) {
            if (!grow(m->root_frame, m->root_frame_bytes, row_bytes * size_t(h), "hipMalloc(multi frame)")) return KIFS_ERR_RUNTIME;
            frame = m->root_frame;
            fpitch = row_bytes;
        }
    }
    // 1. every device renders its shard: the root straight into the frame, the others into a packed buffer
    for (int i = 0; i < n; ++i) {
        kifs_ctx* c = m->ctx[size_t(i)];
        DeviceGuard g(m->dev[size_t(i)]);
        const std::vector<int>& stripes = m->stripes[size_t(i)];
        uint8_t* dst = frame;
        size_t dpitch = fpitch;
        if (i != 0) {
            if (!grow(m->shard[size_t(i)], m->shard_bytes[size_t(i)], row_bytes * size_t(m->rows[size_t(i)]), "hipMalloc(shard)"))
                return KIFS_ERR_RUNTIME;
            dst = m->shard[size_t(i)];
            dpitch = row_bytes;
        }
        if (hipEventRecord(m->ev0[size_t(i)], c->stream) != hipSuccess) return KIFS_ERR_RUNTIME;
        if (!stripes.empty()) {
            st = enqueue_batch(c, c->stream, 1, nullptr, &dst, dpitch, 0, h, encode, stripes.data(), int(stripes.size()),
                               i == 0 ? 1 : 0);
            if (st != KIFS_OK) return st;
        }
        if (hipEventRecord(m->ev1[size_t(i)], c->stream) != hipSuccess) return KIFS_ERR_RUNTIME;
    }
    // 2. the root pulls each finished shard over xGMI and moves its stripes to their frame rows
    {
        DeviceGuard g(m->dev[0]);
        for (int i = 1; i < n; ++i) {
            const std::vector<int>& stripes = m->stripes[size_t(i)];
            if (stripes.empty()) continue;
            const size_t bytes = row_bytes * size_t(m->rows[size_t(i)]);
            if (!grow(m->recv[size_t(i)], m->recv_bytes[size_t(i)], bytes, "hipMalloc(received shard)")) return KIFS_ERR_RUNTIME;
            if (hipStreamWaitEvent(root->stream, m->ev1[size_t(i)], 0) != hipSuccess) return KIFS_ERR_RUNTIME;
            if (!hip_ok(hipMemcpyPeerAsync(m->recv[size_t(i)], m->dev[0], m->shard[size_t(i)], m->dev[size_t(i)], bytes,
                                           root->stream), "peer copy of a shard"))
                return KIFS_ERR_COMM;
            const RowTable* rows = row_table(root, stripes.data(), int(stripes.size()), h);
            if (!rows) return KIFS_ERR_RUNTIME;
            if (!hip_ok(kifs::launch_unpack_stripes(frame, fpitch, 0, m->recv[size_t(i)], row_bytes, 0, rows->d_rows,
                                                    int(stripes.size()), 1, w, h, root->stream), "unpack_stripes_kernel launch"))
                return KIFS_ERR_RUNTIME;
        }
        if (host_dst &&
            !hip_ok(hipMemcpy2DAsync(out, pitch, frame, fpitch, row_bytes, size_t(h), hipMemcpyDeviceToHost,
                                     root->stream), "frame to host"))
            return KIFS_ERR_RUNTIME;
        if (!hip_ok(hipStreamSynchronize(root->stream), "multi sync")) return KIFS_ERR_RUNTIME;
    }
    for (int i = 0; i < n; ++i) {
        DeviceGuard g(m->dev[size_t(i)]);
        if (hipStreamSynchronize(m->ctx[size_t(i)]->stream) != hipSuccess) return KIFS_ERR_RUNTIME;
        float ms = 0.0f;
        m->shard_ms[size_t(i)] =
            hipEventElapsedTime(&ms, m->ev0[size_t(i)], m->ev1[size_t(i)]) == hipSuccess ? double(ms) : -1.0;
    }
    return KIFS_OK;
}

}  // extern "C"

// ---- sRGB threshold table ---------------------------------------------------------------
namespace kifs {

static double oetf(double l) { return l <= 0.0031308 ? 12.92 * l : 1.055 * std::pow(l, 1.0 / 2.4) - 0.055; }
static double eotf(double v) { return v <= 0.04045 ? v / 12.92 : std::pow((v + 0.055) / 1.055, 2.4); }

// t[k] (k >= 1): the smallest binary32 x for which 255 * OETF(x) >= k - 0.5, located by
// inverting in double precision and then walking single ulps until the predicate flips.
void build_srgb_thresholds(float t[256]) {
    t[0] = 0.0f;
    for (int k = 1; k < 256; ++k) {
        const double want = double(k) - 0.5;
        float x = float(eotf(want / 255.0));
        while (oetf(double(x)) * 255.0 < want) x = std::nextafterf(x, 2.0f);
        while (true) {
            float below = std::nextafterf(x, -1.0f);
            if (oetf(double(below)) * 255.0 >= want) x = below; else break;
        }
        t[k] = x;
    }
}

}  // namespace kifs
