// kifs_schedule.cpp -- one launch of the render path: kernel parameters from the uniform images, tile
// tables and their temporal feedback, and the launch shape (which kernel form, how many tiles per
// workgroup, residency).  Host code only; kernels live in the three *_kernels.hip files.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <vector>

#include "kifs_context.hpp"

namespace kifs {
namespace host {

// ---- launch-shape rules: every tuned constant of this file, in one place ---------------------------------------
// Each was read off a sweep on MI355X (gfx950, 256 CUs, 1024 SIMDs); the record it came from is named beside it.
// To re-derive them on other silicon: tools/sweep_kernels.sh forces every shape in turn over 5 frame sizes x 4
// camera distances x 2 scenes x batches of 1 / 8 / 32 (-> sweep_shapes.jsonl), tools/cliff_sweep.py then checks
// the rule in place (-> sweep.jsonl: ns per disc pixel must fall smoothly with the load).  `load` = the launch's
// tiles that can hold rays with real work: disc_tiles() x views.
namespace rules {
// residency_for(): lone Julia frames whose heavy tiles fit the device once or twice cap their own residency
// (1080p at distance 5: 207 -> 172 us at one workgroup per CU; 4096^2 needs every slot: 0.70 vs 1.95 ms capped)
constexpr double RESIDENCY_ONE_PER_CU = 1024.0;   // heavy tiles <= this: one workgroup per CU   (profiles/r01/README.md: 182.9 -> 172.8 us)
constexpr double RESIDENCY_TWO_PER_CU = 2048.0;   // <= this: two; above: uncapped               (same table)
// tile-order feedback (cost per tile -> counting sort -> next order) pays from this many tiles per launch geometry
constexpr uint32_t FEEDBACK_MIN_TILES = 2048u;        // Julia pipelines: 720p and up (round 1; no sweep record kept --
                                                      //   re-derive: KIFS_TILE_FEEDBACK=0 / 2 under tools/cliff_sweep.py)
constexpr uint32_t FEEDBACK_MIN_TILES_KIFS = 16384u;  // KIFS: 8K 2.58 -> 2.23 ms, 1080p nothing (round 1, same way)
constexpr int FEEDBACK_PERIOD_LONE = 4;               // refresh every n-th lone launch           (tools/sweep_period.sh, r02)
constexpr int FEEDBACK_PERIOD_BATCH = 3;              // every n-th batched launch                (tools/sweep_period.sh, r02)
// ray re-queuing: march steps per round
constexpr int ROUND_STEPS_JULIA = 16;     // Julia and generalised Julia (4/8/12/16/24/32 -> 62.8/72.2/75.0/76.8/76.8/77.6
                                          // Gpixel/s at 32 frames per launch)                    (tools/sweep_rounds.sh)
constexpr int ROUND_STEPS_OTHER = 8;      // KIFS scenes; gen-Julia marches under 32 steps        (tools/sweep_rounds.sh)
constexpr int ROUND_STEPS_KIFS_WAVE = 4;  // KIFS scenes, one wave per tile (2/3/4/5/6/8/12/16 steps: 1080p Sierpinski x48 82.2/83.2/82.7/82.4/81.9/81.8/80.0/78.9
                                          // Gpixel/s, 8K x8 32.9/33.0/33.2/33.0/32.9/32.8/32.2/31.9; the 256-thread kernel keeps 8: x8 43.2 against 42.8)   (r03)
constexpr int ROUND_STEPS_LONE_JULIA = 32;  // an uncapped lone Julia frame (4096^2: 0.430 vs 0.445 ms at 16)
constexpr uint64_t REQUEUE_MIN_WORKGROUPS = 4096u;  // below: no rounds at all (256^2 x 8: 0.038 vs 0.062 ms)
// shape of the re-queuing path, from profiles/r02/sweep_shapes.jsonl (every shape forced in turn):
constexpr double WAVE_FROM_LONE = 30000.0;    // one wave per tile (render_wave_kernel) from this load: lone frames,
constexpr double WAVE_FROM_JULIA = 12500.0;   //   batched Julia (1080p x32: 1.13 -> 0.88 ms; 4096^2 x8 +27 %; re-swept r03 after the wave
                                              //   kernel's +15 %: 1080p x16 63.9 against 65.8 with pairs, x20 70.5 against 66.6, x24 76.2 / 67.8),
constexpr double WAVE_FROM_OTHER = 32000.0;   //   everything else (8K Sierpinski x4 +16 %)
constexpr double PAIR_FROM_BATCH = 3500.0;    // two tiles per 256-thread workgroup: batches (1080p Julia 0.319 -> 0.281 ms;
                                              //   below it pairing halves the workgroups side by side: 720p x8 0.222 -> 0.188 with one)
constexpr double PAIR_FROM_GENJULIA = 5000.0;   // generalised Julia (r04, chunks drawn by ticket: x4 8.9 -> 8.0 Gpixel/s, x8 13.4 / 13.5, x12 15.8 -> 16.4,
                                                //   x16 17.7 -> 19.4, x24 19.4 -> 22.4; 12 000 until then)          (tools/sweep_group_shapes.sh)
constexpr double PAIR_FROM_BATCH_KIFS = 7000.0; // KIFS scenes (r04: 1080p Sierpinski x8 43.4 with one tile against 42.2 with two, x16 54.6 / 61.7)
constexpr double PAIR_FROM_LONE_KIFS = 12000.0; // a big lone KIFS frame (1440p Sierpinski at distance 2: -11 %)
// the bunny (tools/sweep_bunny_coop.sh -> profiles/r03/sweep_bunny_coop.txt; 1080p at distance 5 is 149 heavy tiles a frame):
constexpr double BUNNY_ROUNDS_FROM = 450.0;   // below: whole rays, four lanes per pixel (x2: 0.421 against 0.528 ms; x4: 0.592 against 0.536)
constexpr double BUNNY_PAIR_FROM = 1600.0;    // two tiles per workgroup (r04, chunks drawn by ticket: x8 28.3 -> 25.4 Gpixel/s; x12 30.4 -> 33.0; x16 30.0 -> 39.1)
constexpr int ROUND_STEPS_BUNNY_COOP = 4;     // its rounds (x48: 1/2/3/4/6/8 steps -> 52.2/53.6/53.8/53.5/53.3/52.8 Gpixel/s; the lanes-per-ray form: 4-8 alike)
constexpr double BUNNY_W2LDS_FROM = 2600.0;   // pairs with layer 2 of the network in LDS, three waves per SIMD (r04, profiles/r04/sweep_bunny_w2lds.txt:
                                              //   x16 39.1 -> 37.1 Gpixel/s, x20 40.1 -> 44.4, x24 40.6 -> 49.0, x28 40.4 -> 49.0, x32 39.9 -> 49.1)
constexpr double BUNNY_COOP_FROM = 5000.0;    // four waves per 64 rays (same file: x28 43.9 against 49.0 for the form above, x32 48.4 / 49.1, x40 55.0 / 50.0, x48 58.0 / 50.0)
}  // namespace rules

int tuning_knob(const char* name) {
    static const bool enabled = [] {
        const char* e = std::getenv("KIFS_TUNING");
        return e && e[0] == '1';
    }();
    if (!enabled) return -1;
    const char* e = std::getenv(name);
    return e ? int(std::strtol(e, nullptr, 10)) : -1;
}

bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    static const bool verbose = std::getenv("KIFS_DEBUG") != nullptr;
    if (verbose) std::fprintf(stderr, "kifs: %s failed: %s\n", what, hipGetErrorString(e));
    return false;
}

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
        // (max_distance below 1e15 -- the GUI's range ends at 1e4 -- and, per view, an origin within 1e15 of the
        // scene: enqueue_batch; the long-ray loop's square root of |p|^2 relies on |p|^2 being finite then)
        const bool sane = B > 0.0f && R > 0.5f && R < 1.0e6f && o.epsilon >= 0.0f && o.max_distance < 1.0e15f;
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
        int rounds = forced >= 0 ? forced : (o.fractal_group_id != uint32_t(kifs::GROUP_KIFS) ? rules::ROUND_STEPS_JULIA : rules::ROUND_STEPS_OTHER);
        if (forced < 0 && rounds == rules::ROUND_STEPS_JULIA && o.max_iterations < 32 && o.fractal_group_id == uint32_t(kifs::GROUP_GENJULIA))
            rounds = rules::ROUND_STEPS_OTHER;  // (a short march of heavy steps still repays shorter rounds)
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
    P->bunny_coop = 0;
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

void free_table(TileTable& t) {
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

// Device image of a stripe list, cached by content.  Stripes must be ascending and inside the frame: the
// list is validated against THIS frame height before the cache is consulted (the cache is keyed by the
// list alone, and a list cached for a taller frame must not be accepted after kifs_set_screen shrank it).
const RowTable* row_table(kifs_ctx* c, const int* stripes, int n, int height) {
    for (int i = 0; i < n; ++i)
        if (stripes[i] < 0 || int64_t(stripes[i]) * kifs::TILE_H >= height || (i > 0 && stripes[i] <= stripes[i - 1]))
            return nullptr;
    for (const RowTable* r : c->row_tables)
        if (int(r->stripes.size()) == n && std::equal(stripes, stripes + n, r->stripes.begin())) return r;
    std::vector<uint32_t> rows(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) rows[size_t(i)] = uint32_t(stripes[i]) * uint32_t(kifs::TILE_H);
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

// Order in which workgroups take tiles: nearest to the frame centre first (squared
// distance of the tile centre, ties by row then column), so the long rays start first.
// Tiles are TILE_W x TILE_H pixels; rows are counted from the top of the band.
TileTable* tile_table(kifs_ctx* c, int width, int height, int y0, int y1, const RowTable* rows) {
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
    if (heavy_tiles <= rules::RESIDENCY_ONE_PER_CU) return 1;
    if (heavy_tiles <= rules::RESIDENCY_TWO_PER_CU) return 2;
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

int enqueue_batch(kifs_ctx* c, hipStream_t stream, int count, const KifsCameraUniform* cameras,
                  uint8_t* const* outs, size_t pitch, int y0, int y1, int encode,
                  const int* stripes, int n_stripes, int in_place) {
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
    bool far_origin = false;  // a view whose origin is not within 1e15 of the scene: no culls for this launch
    for (int i = 0; i < count; ++i) {
        const KifsCameraUniform& cam = cameras ? cameras[i] : c->camera;
        kifs::BatchView& v = big ? c->h_views[vs][i] : B.view[i];
        v.origin = {cam.origin[0], cam.origin[1], cam.origin[2]};
        v.m0 = {cam.matrix[0][0], cam.matrix[0][1], cam.matrix[0][2]};
        v.m1 = {cam.matrix[1][0], cam.matrix[1][1], cam.matrix[1][2]};
        v.m2 = {cam.matrix[2][0], cam.matrix[2][1], cam.matrix[2][2]};
        v.out = reinterpret_cast<uint32_t*>(outs[i]);
        if (!(double(v.origin.x) * v.origin.x + double(v.origin.y) * v.origin.y + double(v.origin.z) * v.origin.z < 1.0e30))
            far_origin = true;  // (also NaN)
        if (P.tile_cull_beta > 0.0f) {  // the tile-level cull's angle bound assumes an orthonormal matrix
            const kifs::V3* m[3] = {&v.m0, &v.m1, &v.m2};
            for (int a = 0; a < 3; ++a)
                for (int b = a; b < 3; ++b) {
                    const double dot = double(m[a]->x) * m[b]->x + double(m[a]->y) * m[b]->y + double(m[a]->z) * m[b]->z;
                    if (!(std::fabs(dot - (a == b ? 1.0 : 0.0)) <= 1.0e-3)) P.tile_cull_beta = 0.0f;
                }
        }
    }
    if (far_origin) {
        P.cull_n2 = 0.0f;
        P.quick_cull_n2 = 0.0f;
        P.tile_cull_beta = 0.0f;
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
                              tt->count >= ((is_kifs && tile_feedback_mode() < 2) ? rules::FEEDBACK_MIN_TILES_KIFS : rules::FEEDBACK_MIN_TILES);
    // The order is refreshed every FEEDBACK_PERIOD launches (views change slowly; the events the
    // refresh needs cost a few microseconds each).  Within a period of launches k = 0..P-1:
    //   k == 0: record costs, event;   k == 1: sort the costs of launch 0 on the side stream;
    //   k == 2: adopt the new order (wait for the sort);   otherwise: a plain launch.
    static const uint64_t FEEDBACK_PERIOD = [] {
        const int v = tuning_knob("KIFS_FEEDBACK_PERIOD");
        return uint64_t(v < 0 ? rules::FEEDBACK_PERIOD_LONE : v < 3 ? 3 : v);
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
        return uint64_t(v < 0 ? rules::FEEDBACK_PERIOD_BATCH : v < 2 ? 2 : v);
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
    if (P.round_steps == rules::ROUND_STEPS_JULIA && count == 1 && group_id == uint32_t(kifs::GROUP_JULIA) && P.max_iterations >= 64 &&
        tuning_knob("KIFS_ROUND_STEPS") < 0)
        P.round_steps = rules::ROUND_STEPS_LONE_JULIA;
    // (nor does the lone bunny frame: 0.461 ms with the quad kernel, 0.670 ms in rounds; nor two of them)
    if (bunny_scene && (count == 1 || (load < rules::BUNNY_ROUNDS_FROM && tuning_knob("KIFS_ROUND_STEPS") < 0))) P.round_steps = 0;
    // nor does a launch too small to fill the device twice over (256x256 x 8 views = 2048
    // workgroups: 0.038 ms without, 0.062 ms with)
    if (uint64_t(tt->count) * uint64_t(count) < rules::REQUEUE_MIN_WORKGROUPS) P.round_steps = 0;
    {   // Shape of the re-queuing path (profiles/r02/sweep_shapes.jsonl: 5 frame sizes x 4 camera
        // distances x 2 scenes x batches of 1 / 8 / 32, every shape forced in turn):
        //   one WAVE per tile (render_wave_kernel) once the launch has several times more heavy tiles
        //     than the device has workgroup slots -- then slots, not critical paths, set its duration, and
        //     single-wave workgroups give four times as many (1080p Julia x32: 1.13 -> 0.88 ms; 4096^2 x8
        //     +27 %; 8K Sierpinski x4 +16 %) -- from a load of 12 500 tiles for the Julia pipeline, 32 000
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
        const double wave_from = lone ? rules::WAVE_FROM_LONE : (julia ? rules::WAVE_FROM_JULIA : rules::WAVE_FROM_OTHER);
        int shape = 1;
        // (one wave per tile needs views to interleave: TWO frames of 4096^2 -- 19 600 heavy tiles, past WAVE_FROM_JULIA -- run
        // 31.5 Gpixel/s that way against 42.8 with pairs, four 52.0 against 44.0: r04, profiles/r04/sweep_group_shapes.txt)
        if (load >= wave_from && !genjulia && (count >= 3 || load >= rules::WAVE_FROM_LONE)) shape = 0;
        else if (!lone && load >= (genjulia ? rules::PAIR_FROM_GENJULIA : kifs_scene ? rules::PAIR_FROM_BATCH_KIFS : rules::PAIR_FROM_BATCH))
            shape = 2;
        else if (lone && kifs_scene && load >= rules::PAIR_FROM_LONE_KIFS) shape = 2;
        if (forced >= 0) shape = forced;
        P.bunny_coop = 0;
        if (bunny_scene) {  // its own rules: four lanes per ray (216 VGPRs) or four waves per 64 rays
            static const int coop = tuning_knob("KIFS_BUNNY_COOP");
            // (KIFS_BUNNY_COOP under KIFS_TUNING forces a form: 0 / 1 / 2, see FrameParams::bunny_coop)
            P.bunny_coop = coop >= 0 ? std::min(coop, 2) : (load >= rules::BUNNY_COOP_FROM ? 1 : load >= rules::BUNNY_W2LDS_FROM ? 2 : 0);
            shape = P.bunny_coop ? 2 : forced >= 1 ? forced : (load >= rules::BUNNY_PAIR_FROM ? 2 : 1);
            if (P.bunny_coop == 1 && P.round_steps == rules::ROUND_STEPS_OTHER && tuning_knob("KIFS_ROUND_STEPS") < 0)
                P.round_steps = rules::ROUND_STEPS_BUNNY_COOP;
        }
        P.group_tiles = shape;
        if (shape == 0 && kifs_scene && !bunny_scene && P.round_steps == rules::ROUND_STEPS_OTHER && tuning_knob("KIFS_ROUND_STEPS") < 0)
            P.round_steps = rules::ROUND_STEPS_KIFS_WAVE;
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
    c->last_bunny_form = bunny_scene && P.round_steps > 0 ? P.bunny_coop : -1;
    c->last_kernel = P.round_steps > 0 ? (bunny_scene ? (P.bunny_coop == 1 ? KIFS_KERNEL_BUNNY_COOP : KIFS_KERNEL_GROUP)
                                                      : (P.group_tiles == 0 ? KIFS_KERNEL_WAVE : KIFS_KERNEL_GROUP))
                                       : (bunny_scene ? KIFS_KERNEL_BUNNY_QUAD : KIFS_KERNEL_BLOCK);
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


bool grow(uint8_t*& buf, size_t& have, size_t need, const char* what) {
    if (need <= have) return true;
    if (buf) (void)hipFree(buf);
    buf = nullptr;
    have = 0;
    if (!hip_ok(hipMalloc(reinterpret_cast<void**>(&buf), need), what)) return false;
    have = need;
    return true;
}

}  // namespace host
}  // namespace kifs
