// kifs_api.cpp -- the C ABI of include/kifs_hip.h: context lifetime, uniform upload,
// frame render.  Host code only; kernels live in the three *_kernels.hip files.
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

#include "kifs_context.hpp"

using namespace kifs::host;

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
              hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming) == hipSuccess &&
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
    if (c->ev_order) (void)hipEventDestroy(c->ev_order);
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

int kifs_order_after(kifs_ctx* c, void* hip_stream, void* producer_stream) {
    if (!c) return KIFS_ERR_BAD_ARG;
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    hipStream_t producer = static_cast<hipStream_t>(producer_stream);  // NULL: the legacy default stream
    if (s == producer) return KIFS_OK;  // one stream: already in order
    DeviceGuard g(c->device);
    if (!g.ok) return KIFS_ERR_RUNTIME;
    // (one event serves every call: a wait captures the record that precedes it, a later record does not move it)
    return hip_ok(hipEventRecord(c->ev_order, producer), "record(order_after)") &&
                   hip_ok(hipStreamWaitEvent(s, c->ev_order, 0), "wait(order_after)")
               ? KIFS_OK : KIFS_ERR_RUNTIME;
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
int kifs_debug_last_kernel(kifs_ctx* c) { return c ? c->last_kernel : -2; }
int kifs_debug_last_bunny_form(kifs_ctx* c) { return c ? c->last_bunny_form : -2; }

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
    if (!c || !in || !out || n < 0 || fn < 0 || fn > 16) return KIFS_ERR_BAD_ARG;
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
