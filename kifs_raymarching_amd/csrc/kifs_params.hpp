// kifs_params.hpp -- plain data shared by the host side of the library and the kernels.
#pragma once

#include <stdint.h>

namespace kifs {

struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };  // quaternion (real, i, j, k) -- quaternions.wgsl:2-3

enum : int { GROUP_KIFS = 0, GROUP_JULIA = 1, GROUP_GENJULIA = 2 };  // data/scene.rs:4-11
enum : int {                                                           // data/scene.rs:35-45
    PRIM_SPHERE = 0, PRIM_CYLINDER, PRIM_BOX, PRIM_TORUS, PRIM_SIERPINSKI, PRIM_BUNNY, PRIM_OTHER
};

// Everything a frame needs, passed by value as the kernel argument (scalar loads ->
// SGPRs).  Unpacked from the three uniform images of bindings.wgsl:1-35.
struct FrameParams {
    float height, aspect;               // ScreenUniform; the width is `width` below
    V3 origin, m0, m1, m2;              // CameraUniform: origin + matrix columns
    int max_iterations;                 // OptionsUniform
    float max_distance, epsilon;
    V3 fractal_color, background_color;
    uint32_t is_heatmap;
    float power;
    V4 c;
    int sdf_iters, normal_iters, fold_iters;  // julia.wgsl:2-3, kifs.wgsl:72 made parameters
    // Largest f32 v with sqrt(v) <= 2 + epsilon: `length(p) > 2 + epsilon` (julia.wgsl:8) is
    // exactly `dot(p,p) > bound_n2` because correctly rounded sqrt is monotone.
    float bound_n2;
    int orbit_blocks, orbit_rem;        // sdf_iters = 6 * orbit_blocks + orbit_rem
    // (2 + epsilon)^2 * 1.1 for the bounding-sphere culls (0 disables them: NaN/odd epsilon)
    float cull_n2;
    // Smallest f32 v with sqrt(v) >= max_distance: `length(pos) < max_distance` (kifs.wgsl:72)
    // is exactly `dot(pos,pos) < fold_n2_stop`.
    float fold_n2_stop;
    // extension (include/kifs_hip.h KifsExtensions); soft_shadow == 0: the reference's shading
    uint32_t soft_shadow;
    int shadow_steps;
    float shadow_k, shadow_t0, shadow_max_t;
    int width, y0, y1;                  // frame width, row band [y0, y1)
    // Row shards (multi-GPU): when non-null, local tile row j of the launch is the 8-row stripe that
    // starts at frame row stripe_rows[j] (device table; y0 = 0, y1 = frame height); when null, tile
    // row j starts at frame row y0 + 8 j (a contiguous band).
    const uint32_t* stripe_rows;
    // 0: row k of the launch's rows goes to out + k * pitch (a packed band or shard);
    // 1: every row goes to its FRAME position, out + y * pitch (a shard rendered in place).
    int out_frame_rows;
    int encode;                         // KifsEncode
    uint32_t pitch_words;               // output row pitch in 32-bit words
    uint32_t* out;                      // first row of the band
    const float* srgb_table;            // 256 thresholds, device memory
    const uint32_t* tile_order;         // workgroup b renders tile (order[b] & 0xffff, order[b] >> 16)
    uint32_t tile_count;
    uint32_t* tile_cost;                // per tile (row-major tile index): march steps of its slowest wave
    // Optional device counters (nullptr in normal operation): [0] wave-steps taken in the
    // hand-written long-ray loop, [1] wave-steps taken on the general path, [2] entries into
    // the long-ray loop, [3] waves.  Enabled by kifs_debug_counters().
    unsigned long long* counters;
    // Wave-level early exit (see wave_is_culled in kifs_render_common.hpp): a cheaper, more conservative
    // form of the bounding-sphere cull, evaluated before any ray is set up.  0 disables it.
    float quick_cull_n2;                // 1.2 (B + epsilon)^2: well outside cull_n2
    float inv_height;                   // ~1 / height (the quick test needs no exact uv)
    // Tile-level form of the same exit (tile_is_culled in kifs_render_common.hpp): the quick test at the tile's
    // centre against a sphere grown by what the tile subtends.  tile_cull_beta bounds the angle (radians)
    // between the centre's ray and any ray of a 32 x 8 tile; 0 disables the test (camera matrix not
    // orthonormal, frame under 64 rows, quick cull off).  tile_cull_sqrtk = sqrt(quick_cull_n2).
    float tile_cull_beta, tile_cull_sqrtk;
    uint32_t background_rgba;           // the encoded background pixel (same encoder, run on the host)
    // Ray re-queuing (render_kernel): march steps per round between two re-packings of the
    // workgroup's surviving rays into full waves; 0 = every wave marches its own 64 pixels to the end.
    int round_steps;
    // tiles (consecutive entries of the order) per 256-thread workgroup of that path: 1 or 2;
    // 0: one single-wave workgroup per tile (render_wave_kernel)
    int group_tiles;
    // the bunny's throughput path, decided on the host from the launch's load (rules::BUNNY_* in kifs_schedule.cpp):
    // 0 = four lanes per ray, weights in VGPRs (render_group_kernel<KIFS, BUNNY, T>); 1 = four waves per 64 rays
    // (render_bunny_coop_kernel); 2 = four lanes per ray with layer 2 in LDS (render_group_kernel<KIFS, BUNNY, 2, true>)
    int bunny_coop;
    // Host-side launch hint, not read by the kernels: how many workgroups may share a CU
    // (0 = no cap).  See residency_for() in kifs_api.cpp.
    int workgroups_per_cu;
};

// A launch renders a batch of up to MAX_BATCH frames that share screen, options and tile
// order and differ in camera and destination (the frames of an orbit).  One frame's run time
// is the critical path of a few long rays with most SIMDs idle; a batch fills them.  Up to
// MAX_BATCH_INLINE views travel in the kernel argument (56 B each: 3.9 KB with the frame constants,
// limit 4 KB); a bigger batch -- the row shards of a multi-GPU step: a rank's eighth of 384 frames is
// the work of 48 whole ones -- reads them from a device table the host uploads before the launch.
constexpr int MAX_BATCH = 512;
constexpr int MAX_BATCH_INLINE = 64;
struct BatchView {
    V3 origin, m0, m1, m2;  // CameraUniform of this frame
    uint32_t* out;          // first row of its band
};
struct BatchParams {
    FrameParams frame;      // everything common; its camera fields and `out` are those of view 0
    int count;              // 1..MAX_BATCH
    const BatchView* table; // non-null: the views live there (count > MAX_BATCH_INLINE)
    BatchView view[MAX_BATCH_INLINE];
};

}  // namespace kifs
