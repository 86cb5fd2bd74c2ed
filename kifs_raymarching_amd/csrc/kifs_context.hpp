// kifs_context.hpp -- the host side's private types and helpers, shared by the translation units behind
// include/kifs_hip.h:
//   kifs_api.cpp       context lifetime, uniforms, the render entry points, profiling, diagnostics
//   kifs_schedule.cpp  one launch: parameters, tile tables, tile-order feedback, launch shape (enqueue_batch)
//   kifs_shards.cpp    row shards and sparse shards (the multi-GPU partition's per-device entry points)
//   kifs_multi.cpp     one process driving several devices (kifs_multi_*)
// A kifs_ctx plays the part of the reference's GraphicState (render/graphics.rs:25-37): it owns the
// "device objects" (stream, events, the sRGB table in HBM, a scratch frame for host-destination renders)
// and a copy of the three uniform images.  There is no CPU path: every entry point that produces pixels
// launches the HIP kernels or fails.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "../../include/kifs_hip.h"
#include "kifs_internal.hpp"

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
    hipEvent_t ev_order = nullptr;  // kifs_order_after: recorded on the producer's stream, waited for on the launch stream
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
    int last_kernel = -1;      // kifs_debug_last_kernel
    int last_bunny_form = -1;  // kifs_debug_last_bunny_form
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

namespace kifs {
namespace host {

// Tuning overrides (KIFS_ROUND_STEPS, KIFS_GROUP_TILES, KIFS_TILE_FEEDBACK, KIFS_FEEDBACK_PERIOD,
// KIFS_BATCH_PERIOD; KIFS_LDS_PAD in kifs_kernels.hip) are honoured only when KIFS_TUNING=1 is set as
// well: they exist for tools/sweep_kernels.sh and friends, not for production hosts.  -1 = not set.
int tuning_knob(const char* name);

// KIFS_DEBUG=1 prints the failing HIP call to stderr (status codes stay the contract).
bool hip_ok(hipError_t e, const char* what);

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
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

int frame_dims(const kifs_ctx* c, int* w, int* h);
int fill_params(const kifs_ctx* c, kifs::FrameParams* P);
bool is_device_pointer(const void* p);
void free_table(TileTable& t);
// Device image of a stripe list, cached by content.  Stripes must be ascending and inside the frame.
const RowTable* row_table(kifs_ctx* c, const int* stripes, int n, int height);
// `rows` non-null: the table of a row shard (tile row j = stripe rows->stripes[j]; y0 = 0, y1 = height).
TileTable* tile_table(kifs_ctx* c, int width, int height, int y0, int y1, const RowTable* rows = nullptr);
// The background pixel, encoded exactly as the kernels would (unorm8 / srgb8 of kifs_device_math.hpp).
uint32_t background_pixel(const kifs_ctx* c, kifs::V3 colour, int encode);
// One launch: `count` frames (count == 1 and cameras NULL: the context's camera; else cameras[i] -> outs[i])
// sharing everything else.  `stripes` non-null: the launch renders that row shard (y0 = 0, y1 = height)
// instead of a band, into packed rows (in_place == 0) or at the rows' frame positions (in_place != 0).
int enqueue_batch(kifs_ctx* c, hipStream_t stream, int count, const KifsCameraUniform* cameras,
                  uint8_t* const* outs, size_t pitch, int y0, int y1, int encode,
                  const int* stripes = nullptr, int n_stripes = 0, int in_place = 0);
int enqueue(kifs_ctx* c, hipStream_t stream, uint8_t* dev_out, size_t pitch, int y0, int y1, int encode);
// hipMalloc-backed buffer that only ever grows
bool grow(uint8_t*& buf, size_t& have, size_t need, const char* what);

}  // namespace host
}  // namespace kifs
