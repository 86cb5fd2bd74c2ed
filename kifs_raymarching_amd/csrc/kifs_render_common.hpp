// kifs_render_common.hpp -- what the render kernels of kifs_kernels.hip and kifs_bunny_kernels.hip share: a view's
// parameters out of the batch, the tile <-> frame row mapping of bands and row shards, and the wave- and tile-level
// forms of the bounding-sphere cull.
#pragma once

#include "kifs_internal.hpp"
#include "kifs_scene.hpp"
#include "kifs_bunny.hpp"

namespace kifs {

constexpr int BLOCK = TILE_W * TILE_H;  // 256 threads = 4 waves

// Frame `view` of the batch: the common parameters with that view's camera and destination.
// Workgroup b of a launch works on view b % count and takes entry b / count of the tile order,
// so the expensive tiles of every frame of the batch start at t = 0.
__device__ __forceinline__ FrameParams batch_frame(const BatchParams& B, uint32_t view) {
    FrameParams P = B.frame;
    if (B.count > 1) {  // uniform; a batch of one carries its view in B.frame already
        if (B.table) {
            // a table in device memory, read through the constant address space: the index is uniform, so
            // these are scalar loads like the kernel argument's own (a generic pointer would cost VGPRs)
            typedef const BatchView __attribute__((address_space(4))) * ConstView;
            const ConstView v = (ConstView)(B.table + view);
            P.origin = V3{v->origin.x, v->origin.y, v->origin.z};
            P.m0 = V3{v->m0.x, v->m0.y, v->m0.z};
            P.m1 = V3{v->m1.x, v->m1.y, v->m1.z};
            P.m2 = V3{v->m2.x, v->m2.y, v->m2.z};
            P.out = v->out;
        } else {
            const BatchView& v = B.view[view];
            P.origin = v.origin;
            P.m0 = v.m0;
            P.m1 = v.m1;
            P.m2 = v.m2;
            P.out = v.out;
        }
    }
    return P;
}

// Frame row at which local tile row `tile_row` of the launch starts: a contiguous band counts on
// from y0, a row shard looks its stripe up (scalar load: tile_row is uniform per workgroup).
__device__ __forceinline__ int tile_frame_row(const FrameParams& P, uint32_t tile_row) {
    return P.stripe_rows ? int(P.stripe_rows[tile_row]) : P.y0 + int(tile_row) * TILE_H;
}
// Row of the destination for frame row `y` = row `local` of the launch's rows.
__device__ __forceinline__ size_t out_row(const FrameParams& P, int y, int local) {
    return size_t(P.out_frame_rows ? y : local);
}

// True when no pixel of this wave can ever be hit: every valid lane's ray passes the origin at more
// than sqrt(1.2) (B + epsilon), B the scene's bounding radius (fill_params).  Same geometry as
// ray_never_inside, but on the unnormalised direction and an approximate uv (28 instructions, no
// divide, no square root): closest approach c^2 = |o|^2 - (o.d)^2 / |d|^2 > K  <=>
// (|o|^2 - K) |d|^2 > (o.d)^2.  K is 9 % above the radius the exact cull uses, five orders of
// magnitude more than the rounding of this arithmetic, so a wave that leaves here would have had
// all its lanes culled at ray set-up anyway and its pixels are the background colour either way.
// In a 1080p frame nine waves in ten leave here without setting up a single ray.
__device__ __forceinline__ bool wave_is_culled(const FrameParams& P, int x, int y, bool valid) {
    if (!(P.quick_cull_n2 > 0.0f)) return false;  // uniform
    const float px = float(x) + 0.5f, py = float(y) + 0.5f;
    const float ux = (2.0f * px) * P.inv_height - P.aspect;
    const float uy = (2.0f * py) * P.inv_height - 1.0f;
    const V3 d{(ux * P.m1.x - uy * P.m2.x) - P.m0.x, (ux * P.m1.y - uy * P.m2.y) - P.m0.y,
               (ux * P.m1.z - uy * P.m2.z) - P.m0.z};
    const float s = -dot(P.origin, d);  // > 0: the ray approaches the origin
    const float dd = dot(d, d);
    const float room = dot(P.origin, P.origin) - P.quick_cull_n2;
    const bool never = (s <= 0.0f) ? (room > 0.0f) : (room * dd > s * s);
    return __builtin_amdgcn_ballot_w64(valid && !never) == 0ull;
}

// The same exit for a whole 32 x 8 tile, before anything else is computed: the quick test at the
// tile's centre against a sphere grown by what the tile subtends.  With theta the angle between a ray
// and the direction to the origin, the ray's line passes the origin at |o| sin(theta) (theta < 90
// degrees; beyond that the ray moves away and never enters as long as the camera is outside).  Every ray
// of the tile is within beta = P.tile_cull_beta of the ray through the tile's centre (fill_params), and
// sin is 1-Lipschitz and increasing up to 90 degrees, so all of them pass at more than sqrt(K) if the
// centre's ray passes at more than T = sqrt(K) + |o| beta -- or points away while |o| > T, which also
// covers the rays of such a tile that still approach: theirs is |o| cos(beta) >= |o| (1 - beta) > sqrt(K).
// K = quick_cull_n2 as in wave_is_culled, with the same 9 % of room over the exact cull for the
// rounding of this arithmetic and of hardware sqrt.  In a 1080p frame 89 tiles in 100 leave here; the
// ring of tiles within half a tile's diagonal of the projected sphere goes on to the per-block tests.
__device__ __forceinline__ bool tile_is_culled(const FrameParams& P, int tile_x, int frame_y) {
    if (!(P.tile_cull_beta > 0.0f)) return false;  // uniform
    const float oo = dot(P.origin, P.origin);
    const float T = fmaf_(1.01f * P.tile_cull_beta, __builtin_amdgcn_sqrtf(oo), P.tile_cull_sqrtk);
    const float room = oo - T * T;
    // pixel centres x + 0.5 .. x + 31.5 and y + 0.5 .. y + 7.5: the tile's centre is (x + 16, y + 4)
    const float ux = (2.0f * (float(tile_x) + 16.0f)) * P.inv_height - P.aspect;
    const float uy = (2.0f * (float(frame_y) + 4.0f)) * P.inv_height - 1.0f;
    const V3 d{(ux * P.m1.x - uy * P.m2.x) - P.m0.x, (ux * P.m1.y - uy * P.m2.y) - P.m0.y,
               (ux * P.m1.z - uy * P.m2.z) - P.m0.z};
    const float s = -dot(P.origin, d);
    const float dd = dot(d, d);
    const bool never = (room > 0.0f) && ((s <= 0.0f) || (room * dd > s * s));
    return __builtin_amdgcn_readfirstlane(int(never)) != 0;  // every lane holds the same value
}
__device__ __forceinline__ bool tile_is_whole(const FrameParams& P, int tile_x, int frame_y) {
    return tile_x + TILE_W <= P.width && frame_y + TILE_H <= P.y1;
}
// The background over the row pairs [k0, k1) of a whole tile (pair k = rows 2k, 2k + 1: one wave's
// 64 lanes), straight from registers: one address, one store per pair.  (Only render_wave_kernel uses
// the tile-level exit: a 256-thread workgroup's empty tile costs its four wave launches, whatever they
// execute -- 8 frames per launch: 50.3 Gpixel/s with, 51.1 without.)
__device__ __forceinline__ void store_background(const FrameParams& P, int tile_x, int tile_y, int frame_y,
                                                 uint32_t lane, int k0, int k1) {
    uint32_t* row = P.out + (out_row(P, frame_y, tile_y) + size_t(2 * k0)) * P.pitch_words + uint32_t(tile_x);
    const uint32_t at = (lane >> 5) * P.pitch_words + (lane & 31u);
    for (int k = k0; k < k1; ++k) {
        row[at] = P.background_rgba;
        row += 2u * P.pitch_words;
    }
}

// A linear colour as the frame's RGBA8 pixel: the colour target of the reference (render.rs:72-80, graphics.rs:87-91:
// sRGB or UNORM format, alpha 1.0), `table` = the sRGB thresholds (LDS or memory).
__device__ __forceinline__ uint32_t encode_rgba(V3 colour, bool srgb, const float* __restrict__ table) {
    uint32_t r, g, b;
    if (srgb) {
        r = srgb8(colour.x, table);
        g = srgb8(colour.y, table);
        b = srgb8(colour.z, table);
    } else {
        r = unorm8(colour.x);
        g = unorm8(colour.y);
        b = unorm8(colour.z);
    }
    return r | (g << 8) | (b << 16) | 0xff000000u;  // alpha = 1.0 -> 255
}

// the bunny's kernels (kifs_bunny_kernels.hip), launched from launch_render's dispatch in kifs_kernels.hip
hipError_t launch_bunny_coop(const BatchParams& B, hipStream_t stream);        // four waves per 64 rays, re-queued
hipError_t launch_bunny_whole_rays(const BatchParams& B, hipStream_t stream);  // four lanes per pixel, start to finish

}  // namespace kifs
