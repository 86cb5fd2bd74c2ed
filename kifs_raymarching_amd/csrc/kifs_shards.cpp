// kifs_shards.cpp -- row shards and sparse shards: the per-device entry points of the multi-GPU partition
// (include/kifs_hip.h "row shards" and "sparse shards").  Host code only.
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

static_assert(KIFS_STRIPE_ROWS == kifs::TILE_H, "a stripe is one row of the kernels' tiles");

extern "C" {

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

}  // extern "C"

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

extern "C" {

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

}  // extern "C"

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

extern "C" {

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

}  // extern "C"
