// kifs_internal.hpp -- launcher prototypes shared by the host code and the kernel files (kifs_kernels.hip, kifs_support_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "kifs_params.hpp"

namespace kifs {

hipError_t launch_render(const BatchParams& B, uint32_t group, uint32_t primitive,
                         hipStream_t stream);
// Device-side counting sort: order[] = tile ids (x | y << 16) by descending cost[] >> shift (1024 bins);
// clears cost[].
hipError_t launch_tile_order(uint32_t* cost, uint32_t* order, uint32_t tile_count,
                             uint32_t tiles_x, uint32_t shift, hipStream_t stream);
// Packed row shards -> frame rows: stripe s of shard f (8 rows at src + f * src_shard_stride + 8 s *
// src_pitch) goes to frame rows stripe_rows[s].. of dst + f * dst_frame_stride.
hipError_t launch_unpack_stripes(uint8_t* dst, size_t dst_pitch, size_t dst_frame_stride, const uint8_t* src,
                                 size_t src_pitch, size_t src_shard_stride, const uint32_t* stripe_rows,
                                 int n_stripes, int count, int width, int height, hipStream_t stream);
// Sparse shards (see pack_sparse_kernel): records of SPARSE_RECORD_WORDS words for the tiles of packed shards
// that hold a pixel other than `background`; *n_records must be zero beforehand.  And back: records -> frame
// rows; the background over the rows of the listed stripes.
constexpr int SPARSE_RECORD_WORDS_HOST = 260;
hipError_t launch_pack_sparse(const uint8_t* src, size_t src_pitch, size_t src_shard_stride, const uint32_t* stripe_rows,
                              int n_stripes, int count, int width, int height, uint32_t background, uint32_t* records,
                              uint32_t* n_records, hipStream_t stream);
hipError_t launch_unpack_sparse(uint8_t* dst, size_t dst_pitch, size_t dst_frame_stride, const uint32_t* records,
                                uint32_t n_records, const uint32_t* stripe_rows, int n_stripes, int count, int width,
                                int height, int erase, uint32_t background, hipStream_t stream);
hipError_t launch_fill_stripes(uint8_t* dst, size_t dst_pitch, size_t dst_frame_stride, const uint32_t* stripe_rows,
                               int n_stripes, int count, int width, int height, uint32_t background, hipStream_t stream);
hipError_t launch_eval_points(const FrameParams& P, uint32_t group, uint32_t primitive,
                              const float* pts, int n, float* sdf, float* nrm,
                              hipStream_t stream);
hipError_t launch_eval_math(int fn, const float* in, float param, const float* srgb_table,
                            float* out, int n, hipStream_t stream);

// Workgroup tile geometry of render_kernel (shared with the host-side tile ordering).
constexpr int TILE_W = 32;
constexpr int TILE_H = 8;

// 256 sRGB thresholds: t[k] = smallest f32 whose ideal sRGB UNORM8 encoding is >= k.
void build_srgb_thresholds(float t[256]);

}  // namespace kifs
