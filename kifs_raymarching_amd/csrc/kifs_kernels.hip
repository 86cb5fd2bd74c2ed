// kifs_kernels.hip -- gfx950 kernels of the raymarching library and their launchers.
//
// Four render kernels over 32 x 8 pixel tiles taken from a tile ORDER table, all fed by the kernel
// argument (BatchParams: frame constants + up to 64 views; scalar loads -> SGPRs), all storing encoded
// pixels through an LDS tile so that every wave-level store instruction writes two full 128-byte row
// segments:
//   render_kernel<GROUP, PRIM>          the latency path, 256 threads: each wave owns an 8 x 8 block of
//                                       its tile and marches its 64 rays from start to finish (lone
//                                       frames, heatmap frames, diagnostics, small launches)
//   render_group_kernel<GROUP, PRIM, T> mid-size launches, 256 threads: the rays of T tiles share an LDS
//                                       queue and are re-packed into full waves every few march steps
//   render_wave_kernel<GROUP, PRIM>     big launches, 64 threads: one wave per tile with a private queue,
//                                       no barrier, the orbit in its scalar form (VALU bound)
//   render_bunny_quad_kernel            the bunny primitive, four lanes per pixel
// (which one a launch gets: enqueue_batch in kifs_api.cpp, from the projected-disc tile count)
// Tile order: a frame's run time is set by its longest rays (a lone wave pays ~5 cycles per
// instruction whatever else the chip does), so workgroups start with the expensive tiles --
// nearest-to-the-image-centre first on a geometry's first launches (the camera always looks at
// the world origin, data.rs:115-129, where every scene of the reference sits), then by the
// per-tile run times the launches themselves record (tile_order_kernel).  Neighbouring entries of
// the table are dealt round-robin over the 8 XCDs by the dispatcher, which spreads the expensive
// tiles evenly; there is no inter-workgroup data, so placement only affects speed.
//
// Replaces: vs_main + rasteriser + fs_main + ROP of the reference
// (src/shaders/dependencies/entry.wgsl:35-59, src/render/graphics.rs:310-325,
// src/render.rs:72-80).
#include <cstdlib>

#include "kifs_internal.hpp"
#include "kifs_scene.hpp"

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
template <int GROUP, int PRIM>
__global__ __launch_bounds__(BLOCK) void render_kernel(const BatchParams B) {
    __shared__ float s_srgb[256];
    __shared__ uint32_t s_tile[TILE_H][TILE_W];
    __shared__ int s_steps[BLOCK / 64];

    const uint32_t batch = uint32_t(B.count);
    const uint32_t view = batch > 1 ? blockIdx.x % batch : 0u;
    const uint32_t slot = batch > 1 ? blockIdx.x / batch : blockIdx.x;
    const FrameParams P = batch_frame(B, view);
    const int tid = threadIdx.x;
    const bool srgb = (P.encode == 1);
    if (srgb) s_srgb[tid] = P.srgb_table[tid];

    // compute mapping: wave w -> 8x8 sub-tile w, lane -> (lane & 7, lane >> 3)
    const int wave = tid >> 6, lane = tid & 63;
    const int lx = (wave << 3) | (lane & 7);
    const int ly = lane >> 3;
    const uint32_t tile = P.tile_order[slot];  // scalar load: uniform per workgroup
    const int tile_x = int(tile & 0xffffu) * TILE_W;
    const int tile_y = int(tile >> 16) * TILE_H;     // row offset within the launch's rows
    const int frame_y = tile_frame_row(P, tile >> 16);  // the tile's first frame row
    const int x = tile_x + lx;
    const int y = frame_y + ly;
    const bool valid = (x < P.width) && (y < P.y1);

    const bool feedback = P.tile_cost != nullptr;  // wave-uniform
    const unsigned long long wave_start = feedback ? __builtin_amdgcn_s_memtime() : 0ull;
    const bool culled = wave_is_culled(P, x, y, valid);  // wave-uniform
    const uint32_t tiles_x = uint32_t(P.width + TILE_W - 1) / TILE_W;
    uint32_t* const cost_slot = feedback ? &P.tile_cost[(tile >> 16) * tiles_x + (tile & 0xffffu)] : nullptr;

    // one wave per 8x8 block, every ray from start to finish
    V3 colour{0.0f, 0.0f, 0.0f};
    int steps = 0;  // wave-uniform: march steps this wave needed
    if (!culled && __ballot(valid) != 0ull) {
        V3 dir = ray_direction(P, x, y);
        colour = raymarch<GROUP, PRIM>(P, dir, valid, steps);
    }
    // cost of this wave for the next frame's tile order: its run time in units of 1024 cycles,
    // minus a floor that maps culled / instant waves to 0 (march steps alone are too coarse:
    // hundreds of tiles tie at max_iterations)
    if (feedback) {
        const unsigned long long wave_cycles = __builtin_amdgcn_s_memtime() - wave_start;
        if (lane == 0)
            s_steps[wave] = int(min(wave_cycles > 4096ull ? (wave_cycles - 4096ull) >> 10 : 0ull, 1ull << 20));
    }
    (void)steps;
    __syncthreads();  // s_srgb and s_steps visible
    if (tid == 0 && feedback) {  // the tile's slowest wave
        const int m = max(max(s_steps[0], s_steps[1]), max(s_steps[2], s_steps[3]));
        if (batch > 1) atomicMax(cost_slot, uint32_t(m));  // the batch's views share the table (the sort clears it)
        else *cost_slot = uint32_t(m);
    }
    uint32_t rgba = P.background_rgba;
    if (!culled) {
        uint32_t r, g, b;
        if (srgb) {
            r = srgb8(colour.x, s_srgb);
            g = srgb8(colour.y, s_srgb);
            b = srgb8(colour.z, s_srgb);
        } else {
            r = unorm8(colour.x);
            g = unorm8(colour.y);
            b = unorm8(colour.z);
        }
        rgba = r | (g << 8) | (b << 16) | 0xff000000u;  // alpha = 1.0 -> 255
    }
    s_tile[ly][lx] = rgba;
    __syncthreads();

    // store mapping: thread -> (tid & 31, tid >> 5): linear rows of 128 bytes
    const int sx = tid & (TILE_W - 1), sy = tid >> 5;
    const int ox = tile_x + sx;
    if (ox < P.width && (frame_y + sy) < P.y1)
        P.out[out_row(P, frame_y + sy, tile_y + sy) * P.pitch_words + ox] = s_tile[sy][sx];
}

// render_group_kernel<GROUP, PRIM, T>: the throughput path.  A workgroup renders T consecutive
// entries of its view's tile order and RE-QUEUES its rays.
//   Rays of one 8x8 block leave the march at very different steps (one wave per block: 54-66 % of
//   the lanes of a VALU instruction were live).  So the workgroup keeps its live rays in an LDS
//   queue and marches in rounds of `round_steps` steps: a round takes the queue's rays 64 at a
//   time -- full waves -- and every ray ends the round in one of three places: the queue of the
//   next round, the hit list, or nowhere (a miss keeps the pre-filled background).  Hits are
//   shaded at the end, again 64 at a time.  T tiles share one queue because a single tile's queue
//   is short for most of its life (1080p Julia: mean 100 rays, four rounds in ten with <= 16):
//   neighbours in the cost order have tails of similar length and fill each other's waves.
//   A ray's own arithmetic is unchanged (t, and p = fma(t, dir, origin) rebuilt from it exactly as
//   the march itself does), all rays of the workgroup make their k-th step in the same round (the
//   step counter stays uniform), and pixels do not interact: same pixels as render_kernel.
template <int GROUP, int PRIM, int T>
__global__ __launch_bounds__(BLOCK) void render_group_kernel(const BatchParams B) {
    constexpr uint32_t CAP = uint32_t(BLOCK) * T;  // every pixel of the group could be a live ray
    // lanes per ray in the march: 4 for the bunny (bunny_sdf_quad: one network column group per lane,
    // a wave then carries 16 rays), 1 otherwise.  Set-up is one lane per pixel either way.
    constexpr bool BUNNY = (GROUP == GROUP_KIFS && PRIM == PRIM_BUNNY);
    constexpr uint32_t LPR = BUNNY ? 4u : 1u;
    constexpr uint32_t RAYS = 64u / LPR;  // rays per wave and chunk
    __shared__ float s_srgb[256];
    __shared__ uint32_t s_tile[T][TILE_H][TILE_W];
    __shared__ uint32_t s_tiles[T];  // the group's tiles (x | y << 16), 0xffffffff past the table's end
    __shared__ int s_rows[T];        // first frame row of each of them
    // queue entry: pixel (tile-in-group << 8 | ly << 5 | lx) and t: 8 bytes per ray; the ray's direction is
    // computed once at set-up and looked up by pixel (fs_main's two divisions by the height, a square root and
    // normalize()'s three divisions are ~90 instructions a ray would otherwise pay every round)
    __shared__ uint32_t q_pix[2][CAP];
    __shared__ float q_t[2][CAP];
    __shared__ float s_dir[3][CAP];
    __shared__ uint32_t h_pix[CAP];  // hit list: pixel and t (the direction is recomputed)
    __shared__ float h_t[CAP];
    // three counters for two buffers: round r reads count[r % 3], appends under count[(r + 1) % 3] and
    // clears count[(r + 2) % 3], which nobody else touches in that round -- one barrier per round
    __shared__ uint32_t q_count[3], h_count;

    const uint32_t batch = uint32_t(B.count);
    const uint32_t view = batch > 1 ? blockIdx.x % batch : 0u;
    const uint32_t group = batch > 1 ? blockIdx.x / batch : blockIdx.x;
    const FrameParams P = batch_frame(B, view);
    const int tid = threadIdx.x;
    const bool srgb = (P.encode == 1);
    if (srgb) s_srgb[tid] = P.srgb_table[tid];
    const int wave = tid >> 6, lane = tid & 63;
    const int lx = (wave << 3) | (lane & 7);  // wave w -> 8x8 block w of a tile
    const int ly = lane >> 3;
    const bool feedback = P.tile_cost != nullptr;
    const unsigned long long t_start = feedback ? __builtin_amdgcn_s_memtime() : 0ull;
    if (tid == 0) {
        q_count[0] = 0;
        q_count[1] = 0;
        q_count[2] = 0;
        h_count = 0;
    }
    if (tid < T) {
        const uint32_t ti = group * uint32_t(T) + uint32_t(tid);
        const uint32_t tile = ti < P.tile_count ? P.tile_order[ti] : 0xffffffffu;
        s_tiles[tid] = tile;
        s_rows[tid] = tile != 0xffffffffu ? tile_frame_row(P, tile >> 16) : 0;
    }
#pragma unroll
    for (int j = 0; j < T; ++j) s_tile[j][ly][lx] = P.background_rgba;
    __syncthreads();

    // ---- round 0's queue: the rays that survive the culls
    for (int j = 0; j < T; ++j) {
        const uint32_t tile = s_tiles[j];  // uniform
        if (tile == 0xffffffffu) break;
        const int x = int(tile & 0xffffu) * TILE_W + lx;
        const int y = s_rows[j] + ly;
        const bool valid = (x < P.width) && (y < P.y1);
        if (wave_is_culled(P, x, y, valid) || __ballot(valid) == 0ull) continue;  // wave-uniform
        const V3 dir = ray_direction(P, x, y);
        bool alive = valid && (0 < P.max_iterations) && (0.0f < P.max_distance);
        if (P.cull_n2 > 0.0f) alive = alive && !ray_never_inside(P, dir);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(alive);
        if (m == 0ull) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&q_count[0], uint32_t(__builtin_popcountll(m)));
        base = __builtin_amdgcn_readfirstlane(base);
        if (alive) {
            const uint32_t i = base + uint32_t(__builtin_popcountll(m & ((1ull << lane) - 1ull)));
            const uint32_t pix = (uint32_t(j) << 8) | (uint32_t(ly) << 5) | uint32_t(lx);
            q_pix[0][i] = pix;
            q_t[0][i] = 0.0f;
            s_dir[0][pix] = dir.x;
            s_dir[1][pix] = dir.y;
            s_dir[2][pix] = dir.z;
        }
    }
    __syncthreads();

    // ---- rounds
    BunnyQuad W;  // (bunny only) this lane's column group of the network, loaded with the wave's first rays
    bool weights_loaded = false;
    int trips = 0;
    for (uint32_t cur = 0, cnt = 0;; cur ^= 1u, cnt = (cnt + 1u) % 3u) {
        const uint32_t n = q_count[cnt];  // uniform
        if (n == 0u) break;
        const uint32_t cnt_next = (cnt + 1u) % 3u;
        if (tid == 0) q_count[(cnt + 2u) % 3u] = 0;  // the counter of the round after next
        // (one chunk left: nothing more to merge, it is marched to the end -- see render_wave_kernel; not the
        // generalised Julia set, whose compiled march loses 5-10 % that way: 48 frames per launch 20.6 -> 19.6 Gpixel/s)
        const int limit = (GROUP != GROUP_GENJULIA && n <= RAYS) ? P.max_iterations : min(trips + P.round_steps, P.max_iterations);
        for (uint32_t chunk = uint32_t(wave); chunk * RAYS < n; chunk += uint32_t(BLOCK / 64)) {
            const uint32_t idx = chunk * RAYS + uint32_t(lane) / LPR;  // the LPR lanes of a ray hold the same state
            const bool have = idx < n;
            const bool leader = (uint32_t(lane) % LPR) == 0u;      // the lane that files the ray afterwards
            uint32_t pix = 0;
            float t = 0.0f;
            V3 dir{0.0f, 0.0f, 1.0f};
            if (have) {
                pix = q_pix[cur][idx];
                t = q_t[cur][idx];
                dir = V3{s_dir[0][pix], s_dir[1][pix], s_dir[2][pix]};
            }
            // a ray that has advanced has t > 0 (epsilon > 0 on this path), and then its position
            // is what the march last computed: fma(t, dir, origin)
            V3 p = (trips == 0) ? P.origin
                                : V3{fmaf_(t, dir.x, P.origin.x), fmaf_(t, dir.y, P.origin.y),
                                     fmaf_(t, dir.z, P.origin.z)};
            bool hit = false, marching = have;
            int wave_trips = trips;
            if constexpr (BUNNY) {
                if (!weights_loaded) {  // wave-uniform: first chunk of this wave
                    bunny_quad_load(W, lane & 3);
                    weights_loaded = true;
                }
                int i_final = 0;
                generic_loop(P, dir, t, p, hit, marching, wave_trips, i_final, limit,
                             [&](V3 q, unsigned long long) { return bunny_sdf_quad(W, q); });
            } else {
                march_round<GROUP, PRIM>(P, dir, t, p, hit, marching, wave_trips, limit);
            }
            __builtin_amdgcn_s_setprio(0);
            const unsigned long long mh = __builtin_amdgcn_ballot_w64(hit && leader);
            const unsigned long long mq = __builtin_amdgcn_ballot_w64(marching && leader);
            uint32_t bh = 0, bq = 0;
            if (lane == 0) {
                if (mh) bh = atomicAdd(&h_count, uint32_t(__builtin_popcountll(mh)));
                if (mq) bq = atomicAdd(&q_count[cnt_next], uint32_t(__builtin_popcountll(mq)));
            }
            bh = __builtin_amdgcn_readfirstlane(bh);
            bq = __builtin_amdgcn_readfirstlane(bq);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (hit && leader) {
                const uint32_t i = bh + uint32_t(__builtin_popcountll(mh & below));
                h_pix[i] = pix;
                h_t[i] = t;
            } else if (marching && leader) {
                const uint32_t i = bq + uint32_t(__builtin_popcountll(mq & below));
                q_pix[cur ^ 1u][i] = pix;
                q_t[cur ^ 1u][i] = t;
            }
        }
        trips = limit;
        __syncthreads();  // the next queue and the hit list are complete
    }

    // ---- shade the hits, 64 to a wave
    const uint32_t hits = h_count;  // uniform
    if constexpr (BUNNY) {
        if (hits != 0u && !weights_loaded) bunny_quad_load(W, lane & 3);  // a wave that marched nothing shades too
    }
    for (uint32_t i0 = 0; i0 < hits; i0 += uint32_t(BLOCK) / LPR) {
        const uint32_t i = i0 + uint32_t(tid) / LPR;
        if (i < hits) {
            const uint32_t pix = h_pix[i];
            const float t = h_t[i];
            const int hx = int(pix & 31u), hy = int((pix >> 5) & 7u);
            const V3 dir = V3{s_dir[0][pix], s_dir[1][pix], s_dir[2][pix]};
            const V3 p = (t == 0.0f) ? P.origin
                                     : V3{fmaf_(t, dir.x, P.origin.x), fmaf_(t, dir.y, P.origin.y),
                                          fmaf_(t, dir.z, P.origin.z)};
            V3 colour;
            if constexpr (BUNNY) {
                colour = generic_shade(
                    P, p, [&](V3 q, unsigned long long) { return bunny_sdf_quad(W, q); },
                    [&](V3 q) { return normal_fd(P.epsilon, q, [&](V3 u) { return bunny_sdf_quad(W, u); }); });
            } else {
                colour = shade_hit<GROUP, PRIM>(P, p);
            }
            uint32_t r, g, b;
            if (srgb) {
                r = srgb8(colour.x, s_srgb);
                g = srgb8(colour.y, s_srgb);
                b = srgb8(colour.z, s_srgb);
            } else {
                r = unorm8(colour.x);
                g = unorm8(colour.y);
                b = unorm8(colour.z);
            }
            if ((uint32_t(tid) % LPR) == 0u) s_tile[pix >> 8][hy][hx] = r | (g << 8) | (b << 16) | 0xff000000u;
        }
    }
    __syncthreads();

    // ---- store: linear rows of 128 bytes; cost of the group's tiles: the workgroup's run time
    const uint32_t tiles_x = uint32_t(P.width + TILE_W - 1) / TILE_W;
    uint32_t cost = 0;
    if (feedback) {
        const unsigned long long cycles = __builtin_amdgcn_s_memtime() - t_start;
        cost = uint32_t(min(cycles > 4096ull ? (cycles - 4096ull) >> 10 : 0ull, 1ull << 20));
    }
    const int sx = tid & (TILE_W - 1), sy = tid >> 5;
    for (int j = 0; j < T; ++j) {
        const uint32_t tile = s_tiles[j];
        if (tile == 0xffffffffu) break;
        const int ox = int(tile & 0xffffu) * TILE_W + sx;
        const int oy = int(tile >> 16) * TILE_H + sy;  // row within the launch's rows
        const int fy = s_rows[j] + sy;                 // row of the frame
        if (ox < P.width && fy < P.y1) P.out[out_row(P, fy, oy) * P.pitch_words + ox] = s_tile[j][sy][sx];
        if (tid == 0 && feedback) {
            uint32_t* slot = &P.tile_cost[(tile >> 16) * tiles_x + (tile & 0xffffu)];
            if (batch > 1) atomicMax(slot, cost);  // the batch's views share the table (the sort clears it)
            else *slot = cost;
        }
    }
}

// render_wave_kernel<GROUP, PRIM>: the throughput path with ONE WAVE per workgroup and tile.
//   In a batched launch the workgroups that matter spend most of their life as one wave marching a
//   thin tail of long rays while their other three waves are parked at the round barrier, and it is
//   workgroup SLOTS -- six 256-thread workgroups per CU with these kernels' SGPR budget -- that the
//   device runs out of: average residency 1.1 waves per SIMD.  A single-wave workgroup holds a quarter
//   of the slot: four times as many tails march side by side, the hardware's own dispatcher hands the
//   next tile of the cost order to whichever CU has room (no software queue could do that cheaper),
//   and there is nothing to wait for -- no barrier, no atomic.
//   The wave keeps its tile's live rays in an LDS queue (pixel and t; the direction is recomputed from
//   the pixel every round) and marches them in rounds of `round_steps` steps, 64 rays at a time: four
//   chunks at first, one soon after; survivors go to the next round's queue, hits to a list that is
//   shaded at the end, 64 at a time.  Same arithmetic per ray as everywhere else, same pixels.
template <int GROUP, int PRIM>
__global__ __launch_bounds__(64) void render_wave_kernel(const BatchParams B) {
    constexpr uint32_t CAP = TILE_W * TILE_H;  // every pixel of the tile could be a live ray
    __shared__ uint8_t q_pix[2][CAP];  // ly << 5 | lx: a byte
    __shared__ float q_t[2][CAP];
    // The ray direction of every live pixel, computed once at set-up: fs_main's two divisions by the height, a
    // square root and normalize()'s three divisions are ~90 of the ~160 instructions a chunk pays per round outside
    // the march itself.  3 KB -- paid for by the pixel ids shrinking to bytes, the staging tile moving into queue
    // buffer 0 once the march is over (the hit list is in buffer 1), and the sRGB thresholds read from memory (only
    // hit pixels are encoded: 2 % of a frame): 5.5 KB per wave, and LDS must stay under 6.6 KB -- the 24 waves per CU
    // the kernel's SGPRs allow are worth 7 % over 22 (KIFS_LDS_PAD sweep: 7 / 8 / 9 KB: 100.2 / 93.6 / 88.5 Gpixel/s).
    __shared__ float s_dir[3][CAP];
    uint32_t (*const s_tile)[TILE_W] = reinterpret_cast<uint32_t (*)[TILE_W]>(&q_t[0][0]);
    // The hit list grows down from the top of buffer 1 (hit i at index CAP - 1 - i).  Every pixel is a live
    // ray, a hit or finished, so (rays at the start of a round) + (hits before it) <= CAP.  When buffer 1
    // receives the next round's queue (growing up from 0) the two cannot meet.  When buffer 1 holds THIS
    // round's queue [0, n), its chunks are taken from the top down: after the chunks above index c the
    // list has grown by at most n - c entries, so it ends at or above CAP - (hits before) - (n - c) >= c,
    // clear of the part [0, c) still to be read.

    const uint32_t batch = uint32_t(B.count);
    const uint32_t view = batch > 1 ? blockIdx.x % batch : 0u;
    const uint32_t slot = batch > 1 ? blockIdx.x / batch : blockIdx.x;
    const FrameParams P = batch_frame(B, view);
    const uint32_t lane = threadIdx.x;
    const bool srgb = (P.encode == 1);
    const bool feedback = P.tile_cost != nullptr;
    const unsigned long long t_start = feedback ? __builtin_amdgcn_s_memtime() : 0ull;
    const uint32_t tile = P.tile_order[slot];  // scalar load
    const int tile_x = int(tile & 0xffffu) * TILE_W;
    const int tile_y = int(tile >> 16) * TILE_H;         // row offset within the launch's rows
    const int frame_y = tile_frame_row(P, tile >> 16);   // the tile's first frame row
    const unsigned long long below = (1ull << lane) - 1ull;

    // ---- nine tiles in ten of a 1080p frame hold no ray that can hit: one test for the tile, the
    // background straight from registers (whole tiles: one address, four stores), and the wave is gone
    if (tile_is_whole(P, tile_x, frame_y) && tile_is_culled(P, tile_x, frame_y)) {
        store_background(P, tile_x, tile_y, frame_y, lane, 0, TILE_H / 2);
        if (feedback && lane == 0 && batch == 1) {  // (in a batch the sort has cleared the table: atomicMax with 0 is a no-op)
            const uint32_t tiles_x = uint32_t(P.width + TILE_W - 1) / TILE_W;
            P.tile_cost[(tile >> 16) * tiles_x + (tile & 0xffffu)] = 0u;
        }
        return;
    }

    // ---- round 0's queue: the rays that survive the culls, block by block
    uint32_t n = 0;  // wave-uniform throughout (sums of ballot counts)
    for (int b = 0; b < TILE_W / 8; ++b) {
        const int lx = (b << 3) | int(lane & 7u), ly = int(lane >> 3);
        const int x = tile_x + lx, y = frame_y + ly;
        const bool valid = (x < P.width) && (y < P.y1);
        if (wave_is_culled(P, x, y, valid) || __ballot(valid) == 0ull) continue;  // wave-uniform
        const V3 dir = ray_direction(P, x, y);
        bool alive = valid && (0 < P.max_iterations) && (0.0f < P.max_distance);
        if (P.cull_n2 > 0.0f) alive = alive && !ray_never_inside(P, dir);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(alive);
        if (alive) {
            const uint32_t i = n + uint32_t(__builtin_popcountll(m & below));
            const uint32_t pix = (uint32_t(ly) << 5) | uint32_t(lx);
            q_pix[0][i] = uint8_t(pix);
            q_t[0][i] = 0.0f;
            s_dir[0][pix] = dir.x;
            s_dir[1][pix] = dir.y;
            s_dir[2][pix] = dir.z;
        }
        n += uint32_t(__builtin_popcountll(m));
    }
    if (n == 0u) {
        // No ray survives the culls (a ragged tile, or one in the ring the tile-level test leaves): the
        // background straight from registers -- no LDS, no table.
#pragma unroll
        for (int r = 0; r < TILE_H; r += 2) {
            const int sx = int(lane & 31u), sy = r + int(lane >> 5);
            const int ox = tile_x + sx;
            if (ox < P.width && (frame_y + sy) < P.y1)
                P.out[out_row(P, frame_y + sy, tile_y + sy) * P.pitch_words + uint32_t(ox)] = P.background_rgba;
        }
        if (feedback && lane == 0 && batch == 1) {  // (in a batch the sort has cleared the table: atomicMax with 0 is a no-op)
            const uint32_t tiles_x = uint32_t(P.width + TILE_W - 1) / TILE_W;
            P.tile_cost[(tile >> 16) * tiles_x + (tile & 0xffffu)] = 0u;
        }
        return;
    }
    __syncthreads();  // one wave: orders the LDS traffic, costs nothing

    // ---- rounds
    uint32_t hits = 0;
    int trips = 0;
    for (uint32_t cur = 0; n != 0u; cur ^= 1u) {
        // Re-queuing pays when it turns several partly empty chunks into fewer full ones.  Once the tile's rays fit ONE
        // chunk there is nothing left to merge: the wave marches them to the end in one go (dead lanes cost the
        // hand-written loop nothing extra, it leaves when the last lane does) and saves every later round's queue
        // traffic, direction set-up and loop entry.
        const int limit = n <= 64u ? P.max_iterations : min(trips + P.round_steps, P.max_iterations);
        uint32_t n_next = 0;
        for (uint32_t c0 = ((n - 1u) >> 6) << 6;; c0 -= 64u) {  // top chunk first (see the hit list above)
            const bool have = c0 + lane < n;
            uint32_t pix = 0;
            float t = 0.0f;
            V3 dir{0.0f, 0.0f, 1.0f};
            if (have) {
                pix = q_pix[cur][c0 + lane];
                t = q_t[cur][c0 + lane];
                dir = V3{s_dir[0][pix], s_dir[1][pix], s_dir[2][pix]};
            }
            // a ray that has advanced has t > 0 (epsilon > 0 on this path) and sits at fma(t, dir, origin)
            V3 p = (trips == 0) ? P.origin
                                : V3{fmaf_(t, dir.x, P.origin.x), fmaf_(t, dir.y, P.origin.y), fmaf_(t, dir.z, P.origin.z)};
            bool hit = false, marching = have;
            int wave_trips = trips;
            march_round<GROUP, PRIM, true>(P, dir, t, p, hit, marching, wave_trips, limit);  // the orbit in its scalar form
            __builtin_amdgcn_s_setprio(0);
            const unsigned long long mh = __builtin_amdgcn_ballot_w64(hit);
            const unsigned long long mq = __builtin_amdgcn_ballot_w64(marching);
            if (hit) {
                const uint32_t i = CAP - 1u - (hits + uint32_t(__builtin_popcountll(mh & below)));
                q_pix[1][i] = uint8_t(pix);
                q_t[1][i] = t;
            } else if (marching) {
                const uint32_t i = n_next + uint32_t(__builtin_popcountll(mq & below));
                q_pix[cur ^ 1u][i] = uint8_t(pix);
                q_t[cur ^ 1u][i] = t;
            }
            hits += uint32_t(__builtin_popcountll(mh));
            n_next += uint32_t(__builtin_popcountll(mq));
            if (c0 == 0u) break;
        }
        trips = limit;
        n = n_next;
        __syncthreads();
    }

    // ---- the march is over: queue buffer 0 becomes the staging tile (background)
    const float* const s_srgb = P.srgb_table;  // (read from memory: see the LDS budget above)
#pragma unroll
    for (int r = 0; r < TILE_H; r += 2) s_tile[r + int(lane >> 5)][lane & 31u] = P.background_rgba;
    __syncthreads();

    // ---- shade the hits, 64 at a time
    for (uint32_t i0 = 0; i0 < hits; i0 += 64u) {
        if (i0 + lane < hits) {
            const uint32_t pix = q_pix[1][CAP - 1u - (i0 + lane)];
            const float t = q_t[1][CAP - 1u - (i0 + lane)];
            const int hx = int(pix & 31u), hy = int(pix >> 5);
            const V3 dir = V3{s_dir[0][pix], s_dir[1][pix], s_dir[2][pix]};
            const V3 p = (t == 0.0f) ? P.origin
                                     : V3{fmaf_(t, dir.x, P.origin.x), fmaf_(t, dir.y, P.origin.y),
                                          fmaf_(t, dir.z, P.origin.z)};
            const V3 colour = shade_hit<GROUP, PRIM>(P, p);
            uint32_t r, g, b;
            if (srgb) {
                r = srgb8(colour.x, s_srgb);
                g = srgb8(colour.y, s_srgb);
                b = srgb8(colour.z, s_srgb);
            } else {
                r = unorm8(colour.x);
                g = unorm8(colour.y);
                b = unorm8(colour.z);
            }
            s_tile[hy][hx] = r | (g << 8) | (b << 16) | 0xff000000u;
        }
    }
    __syncthreads();

    // ---- store: two full 128-byte rows per instruction; cost of the tile: the wave's run time
#pragma unroll
    for (int r = 0; r < TILE_H; r += 2) {
        const int sx = int(lane & 31u), sy = r + int(lane >> 5);
        const int ox = tile_x + sx;
        if (ox < P.width && (frame_y + sy) < P.y1)
            P.out[out_row(P, frame_y + sy, tile_y + sy) * P.pitch_words + uint32_t(ox)] = s_tile[sy][sx];
    }
    if (feedback) {
        const unsigned long long cycles = __builtin_amdgcn_s_memtime() - t_start;
        const uint32_t cost = uint32_t(min(cycles > 4096ull ? (cycles - 4096ull) >> 10 : 0ull, 1ull << 20));
        if (lane == 0) {
            const uint32_t tiles_x = uint32_t(P.width + TILE_W - 1) / TILE_W;
            uint32_t* cs = &P.tile_cost[(tile >> 16) * tiles_x + (tile & 0xffffu)];
            if (batch > 1) atomicMax(cs, cost);  // the batch's views share the table (the sort clears it)
            else *cs = cost;
        }
    }
}

// The bunny primitive with four lanes per pixel (see bunny_sdf_quad in kifs_scene.hpp): a
// workgroup renders a quarter of a 32 x 8 tile, rows [2 sub, 2 sub + 2); wave w owns the 8 x 2
// pixels at columns [8w, 8w + 8), lane -> pixel lane >> 2, column group lane & 3.  Same tile
// order table, same LDS-staged store (two full 128-byte rows per workgroup).
__global__ __launch_bounds__(BLOCK) void render_bunny_quad_kernel(const BatchParams B) {
    __shared__ float s_srgb[256];
    __shared__ uint32_t s_tile[2][TILE_W];

    const uint32_t batch = uint32_t(B.count);
    const uint32_t view = batch > 1 ? blockIdx.x % batch : 0u;
    const uint32_t block = batch > 1 ? blockIdx.x / batch : blockIdx.x;
    const FrameParams P = batch_frame(B, view);
    const int tid = threadIdx.x;
    const bool srgb = (P.encode == 1);
    if (srgb) s_srgb[tid] = P.srgb_table[tid];

    const int wave = tid >> 6, lane = tid & 63;
    const int pixel = lane >> 2, group = lane & 3;
    const int lx = (wave << 3) | (pixel & 7);
    const int ly = pixel >> 3;
    const uint32_t tile = P.tile_order[block >> 2];
    const int sub = int(block & 3u);
    const int tile_x = int(tile & 0xffffu) * TILE_W;
    const int tile_y = int(tile >> 16) * TILE_H + 2 * sub;  // row offset within the launch's rows
    const int frame_y = tile_frame_row(P, tile >> 16) + 2 * sub;
    const int x = tile_x + lx;
    const int y = frame_y + ly;
    const bool valid = (x < P.width) && (y < P.y1);

    V3 colour{0.0f, 0.0f, 0.0f};
    int steps = 0;
    const bool culled = wave_is_culled(P, x, y, valid);  // wave-uniform
    if (!culled && __ballot(valid) != 0ull) {
        V3 dir = ray_direction(P, x, y);
        colour = raymarch_bunny_quad(P, dir, valid, group, steps);
    }
    (void)steps;
    __syncthreads();  // s_srgb visible
    uint32_t rgba = P.background_rgba;
    if (!culled) {
        uint32_t r, g, b;
        if (srgb) {
            r = srgb8(colour.x, s_srgb);
            g = srgb8(colour.y, s_srgb);
            b = srgb8(colour.z, s_srgb);
        } else {
            r = unorm8(colour.x);
            g = unorm8(colour.y);
            b = unorm8(colour.z);
        }
        rgba = r | (g << 8) | (b << 16) | 0xff000000u;
    }
    if (group == 0) s_tile[ly][lx] = rgba;
    __syncthreads();
    if (tid < 2 * TILE_W) {
        const int sx = tid & (TILE_W - 1), sy = tid >> 5;
        const int ox = tile_x + sx;
        if (ox < P.width && (frame_y + sy) < P.y1)
            P.out[out_row(P, frame_y + sy, tile_y + sy) * P.pitch_words + ox] = s_tile[sy][sx];
    }
}

// render_bunny_coop_kernel<T>: the bunny's throughput path.  render_group_kernel's ray queue, but the four waves
// of the workgroup march the SAME 64 rays of a chunk together, wave j evaluating column group j of the network
// (bunny_sdf_coop in kifs_scene.hpp: weights as scalar operands, activations exchanged through LDS).  All four
// waves hold the same ray state and take the same branches; wave 0 files the rays afterwards.  Chunks are taken
// one after the other by the whole workgroup.
template <int T>
__global__ __launch_bounds__(BLOCK) void render_bunny_coop_kernel(const BatchParams B) {
    constexpr uint32_t CAP = uint32_t(BLOCK) * T;
    // LDS: 24 KB -- six workgroups per CU (at 30 KB, five: 53.8 against 55.8 Gpixel/s at 48 frames per launch).  Pixel ids as 16-bit words, the sRGB
    // thresholds read from memory (only hit pixels are encoded), the staging tiles in queue buffer 0 once the march
    // is over (as in render_wave_kernel).
    __shared__ uint32_t s_tiles[T];
    __shared__ int s_rows[T];
    __shared__ uint16_t q_pix[2][CAP];
    __shared__ float q_t[2][CAP];
    __shared__ float s_dir[3][CAP];
    __shared__ uint16_t h_pix[CAP];
    __shared__ float h_t[CAP];
    static_assert(sizeof(uint32_t) * T * TILE_H * TILE_W == sizeof(float) * CAP, "the staging tiles fill queue buffer 0 exactly");
    uint32_t (*const s_tile)[TILE_H][TILE_W] = reinterpret_cast<uint32_t (*)[TILE_H][TILE_W]>(&q_t[0][0]);
    __shared__ uint32_t q_count[3], h_count;
    __shared__ float x_a[4][4][64], x_b[4][4][64], x_c[4][64];  // the network's exchanges

    const uint32_t batch = uint32_t(B.count);
    const uint32_t view = batch > 1 ? blockIdx.x % batch : 0u;
    const uint32_t group = batch > 1 ? blockIdx.x / batch : blockIdx.x;
    const FrameParams P = batch_frame(B, view);
    const int tid = threadIdx.x;
    const bool srgb = (P.encode == 1);
    const float* const s_srgb = P.srgb_table;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lx = (wave << 3) | (lane & 7);  // set-up: wave w -> 8x8 block w of a tile
    const int ly = lane >> 3;
    const bool feedback = P.tile_cost != nullptr;
    const unsigned long long t_start = feedback ? __builtin_amdgcn_s_memtime() : 0ull;
    if (tid == 0) {
        q_count[0] = 0;
        q_count[1] = 0;
        q_count[2] = 0;
        h_count = 0;
    }
    if (tid < T) {
        const uint32_t ti = group * uint32_t(T) + uint32_t(tid);
        const uint32_t tile = ti < P.tile_count ? P.tile_order[ti] : 0xffffffffu;
        s_tiles[tid] = tile;
        s_rows[tid] = tile != 0xffffffffu ? tile_frame_row(P, tile >> 16) : 0;
    }
    __syncthreads();

    // ---- round 0's queue: the rays that survive the culls (one lane per pixel, the four waves side by side)
    for (int j = 0; j < T; ++j) {
        const uint32_t tile = s_tiles[j];  // uniform
        if (tile == 0xffffffffu) break;
        const int x = int(tile & 0xffffu) * TILE_W + lx;
        const int y = s_rows[j] + ly;
        const bool valid = (x < P.width) && (y < P.y1);
        if (wave_is_culled(P, x, y, valid) || __ballot(valid) == 0ull) continue;  // wave-uniform
        const V3 dir = ray_direction(P, x, y);
        bool alive = valid && (0 < P.max_iterations) && (0.0f < P.max_distance);
        if (P.cull_n2 > 0.0f) alive = alive && !ray_never_inside(P, dir);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(alive);
        if (m == 0ull) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&q_count[0], uint32_t(__builtin_popcountll(m)));
        base = __builtin_amdgcn_readfirstlane(base);
        if (alive) {
            const uint32_t i = base + uint32_t(__builtin_popcountll(m & ((1ull << lane) - 1ull)));
            const uint32_t pix = (uint32_t(j) << 8) | (uint32_t(ly) << 5) | uint32_t(lx);
            q_pix[0][i] = uint16_t(pix);
            q_t[0][i] = 0.0f;
            s_dir[0][pix] = dir.x;
            s_dir[1][pix] = dir.y;
            s_dir[2][pix] = dir.z;
        }
    }
    __syncthreads();

    const BunnyCoop X{x_a, x_b, x_c, wave};
    auto sdf = [&](V3 q, unsigned long long) { return bunny_sdf_coop(X, q); };
    const unsigned long long below = (1ull << lane) - 1ull;
    // ---- rounds: every chunk of 64 rays by all four waves
    // (tried, r03: a queue with two ends -- rays inside the unit ball, whose next estimate runs the network, filed
    // from the front and the others from the back, so that a chunk pays for the network only if it is made of such
    // rays: 52.8 against 53.7 Gpixel/s at 48 frames per launch, nothing at 8 / 16 / 24)
    int trips = 0;
    for (uint32_t cur = 0, cnt = 0;; cur ^= 1u, cnt = (cnt + 1u) % 3u) {
        const uint32_t n = q_count[cnt];  // uniform
        if (n == 0u) break;
        const uint32_t cnt_next = (cnt + 1u) % 3u;
        if (tid == 0) q_count[(cnt + 2u) % 3u] = 0;  // the counter of the round after next
        // (one chunk left: nothing more to merge, it is marched to the end)
        const int limit = n <= 64u ? P.max_iterations : min(trips + P.round_steps, P.max_iterations);
        for (uint32_t c0 = 0; c0 < n; c0 += 64u) {
            const uint32_t idx = c0 + uint32_t(lane);
            const bool have = idx < n;
            uint32_t pix = 0;
            float t = 0.0f;
            V3 dir{0.0f, 0.0f, 1.0f};
            if (have) {
                pix = q_pix[cur][idx];
                t = q_t[cur][idx];
                dir = V3{s_dir[0][pix], s_dir[1][pix], s_dir[2][pix]};
            }
            V3 p = (trips == 0) ? P.origin
                                : V3{fmaf_(t, dir.x, P.origin.x), fmaf_(t, dir.y, P.origin.y),
                                     fmaf_(t, dir.z, P.origin.z)};
            bool hit = false, marching = have;
            int wave_trips = trips, i_final = 0;
            generic_loop(P, dir, t, p, hit, marching, wave_trips, i_final, limit, sdf);
            __builtin_amdgcn_s_setprio(0);
            if (wave == 0) {  // one copy of the chunk's rays goes on
                const unsigned long long mh = __builtin_amdgcn_ballot_w64(hit);
                const unsigned long long mq = __builtin_amdgcn_ballot_w64(marching);
                // (no atomics: wave 0 is the only writer of these counters after the set-up)
                const uint32_t bh = h_count, bq = q_count[cnt_next];
                if (hit) {
                    const uint32_t i = bh + uint32_t(__builtin_popcountll(mh & below));
                    h_pix[i] = uint16_t(pix);
                    h_t[i] = t;
                } else if (marching) {
                    const uint32_t i = bq + uint32_t(__builtin_popcountll(mq & below));
                    q_pix[cur ^ 1u][i] = uint16_t(pix);
                    q_t[cur ^ 1u][i] = t;
                }
                if (lane == 0) {
                    h_count = bh + uint32_t(__builtin_popcountll(mh));
                    q_count[cnt_next] = bq + uint32_t(__builtin_popcountll(mq));
                }
            }
        }
        trips = limit;
        __syncthreads();  // the next queue and the hit list are complete
    }

    // ---- the queues are dead: buffer 0 becomes the staging tiles, background first
#pragma unroll
    for (int j = 0; j < T; ++j) s_tile[j][ly][lx] = P.background_rgba;
    __syncthreads();
    // ---- shade the hits, 64 at a time by all four waves
    const uint32_t hits = h_count;  // uniform
    for (uint32_t i0 = 0; i0 < hits; i0 += 64u) {
        const uint32_t i = min(i0 + uint32_t(lane), hits - 1u);  // (idle lanes repeat the last hit: no divergence around the barriers)
        const uint32_t pix = h_pix[i];
        const float t = h_t[i];
        const int hx = int(pix & 31u), hy = int((pix >> 5) & 7u);
        const V3 dir = V3{s_dir[0][pix], s_dir[1][pix], s_dir[2][pix]};
        const V3 p = (t == 0.0f) ? P.origin
                                 : V3{fmaf_(t, dir.x, P.origin.x), fmaf_(t, dir.y, P.origin.y),
                                      fmaf_(t, dir.z, P.origin.z)};
        const V3 colour = generic_shade(P, p, sdf, [&](V3 q) {
            return normal_fd(P.epsilon, q, [&](V3 u) { return bunny_sdf_coop(X, u); });
        });
        uint32_t r, g, b;
        if (srgb) {
            r = srgb8(colour.x, s_srgb);
            g = srgb8(colour.y, s_srgb);
            b = srgb8(colour.z, s_srgb);
        } else {
            r = unorm8(colour.x);
            g = unorm8(colour.y);
            b = unorm8(colour.z);
        }
        if (wave == 0 && i0 + uint32_t(lane) < hits) s_tile[pix >> 8][hy][hx] = r | (g << 8) | (b << 16) | 0xff000000u;
    }
    __syncthreads();

    // ---- store: linear rows of 128 bytes; cost of the group's tiles: the workgroup's run time
    const uint32_t tiles_x = uint32_t(P.width + TILE_W - 1) / TILE_W;
    uint32_t cost = 0;
    if (feedback) {
        const unsigned long long cycles = __builtin_amdgcn_s_memtime() - t_start;
        cost = uint32_t(min(cycles > 4096ull ? (cycles - 4096ull) >> 10 : 0ull, 1ull << 20));
    }
    const int sx = tid & (TILE_W - 1), sy = tid >> 5;
    for (int j = 0; j < T; ++j) {
        const uint32_t tile = s_tiles[j];
        if (tile == 0xffffffffu) break;
        const int ox = int(tile & 0xffffu) * TILE_W + sx;
        const int oy = int(tile >> 16) * TILE_H + sy;
        const int fy = s_rows[j] + sy;
        if (ox < P.width && fy < P.y1) P.out[out_row(P, fy, oy) * P.pitch_words + ox] = s_tile[j][sy][sx];
        if (tid == 0 && feedback) {
            uint32_t* slot = &P.tile_cost[(tile >> 16) * tiles_x + (tile & 0xffffu)];
            if (batch > 1) atomicMax(slot, cost);
            else *slot = cost;
        }
    }
}

// Dynamic LDS requested only to cap how many workgroups share a CU (the kernel never touches
// it); the cap itself is decided on the host (residency_for() in kifs_api.cpp).
// KIFS_LDS_PAD=<bytes> overrides it (tuning; honoured only with KIFS_TUNING=1).
static unsigned residency_pad_bytes(int workgroups_per_cu) {
    static const long forced = [] {
        const char* on = std::getenv("KIFS_TUNING");
        const char* e = (on && on[0] == '1') ? std::getenv("KIFS_LDS_PAD") : nullptr;
        return e ? std::strtol(e, nullptr, 10) : -1L;
    }();
    if (forced >= 0) return unsigned(forced);
    switch (workgroups_per_cu) {  // static LDS of render_kernel is 2 KiB; a CU has 160 KiB
    case 1: return 100 * 1024;
    case 2: return 72 * 1024;
    case 3: return 50 * 1024;
    default: return 0;
    }
}

// Pipeline selection: the reference keeps three render pipelines and picks one per
// frame by fractal_group (graphics.rs:310-321); the KIFS shader then switches on
// primitive_id per SDF call (kifs.wgsl:139-155).  Here both are template parameters.
template <int GROUP, int PRIM>
static hipError_t launch_variant(const BatchParams& B, hipStream_t stream) {
    const FrameParams& P = B.frame;
    const unsigned pad = residency_pad_bytes(P.workgroups_per_cu);
    if (pad > 48 * 1024) {
        // beyond the default dynamic-LDS limit: opt in once per kernel AND per device (the
        // attribute belongs to the device's copy of the code object)
        static bool opted_in[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
        if (dev < 0 || dev >= 64 || !opted_in[dev]) {
            hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<GROUP, PRIM>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (attr != hipSuccess) return attr;
            if (dev >= 0 && dev < 64) opted_in[dev] = true;
        }
    }
    if (P.round_steps > 0 && P.group_tiles == 0) {  // the throughput path, one wave per tile
        hipLaunchKernelGGL((render_wave_kernel<GROUP, PRIM>), dim3(P.tile_count * uint32_t(B.count)), dim3(64),
                           residency_pad_bytes(0), stream, B);  // (the pad: KIFS_LDS_PAD under KIFS_TUNING, else 0)
        return hipGetLastError();
    }
    if (P.round_steps > 0) {  // the throughput path: rays re-queued, one or two tiles per workgroup
        if (P.group_tiles >= 2) {
            hipLaunchKernelGGL((render_group_kernel<GROUP, PRIM, 2>), dim3(((P.tile_count + 1u) / 2u) * uint32_t(B.count)),
                               dim3(BLOCK), 0, stream, B);
        } else {
            hipLaunchKernelGGL((render_group_kernel<GROUP, PRIM, 1>), dim3(P.tile_count * uint32_t(B.count)), dim3(BLOCK), 0,
                               stream, B);
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL((render_kernel<GROUP, PRIM>), dim3(P.tile_count * uint32_t(B.count)), dim3(BLOCK), pad, stream, B);
    return hipGetLastError();
}

static hipError_t launch_bunny_quad(const BatchParams& B, hipStream_t stream) {
    const FrameParams& P = B.frame;
    if (P.round_steps > 0 && P.bunny_coop) {  // the throughput path: re-queued rays, four waves per 64 rays
        // (always two tiles per workgroup: one was slower at every batch size where this form wins at all --
        // 48 frames per launch 46.5 against 51.8 Gpixel/s, profiles/r03/sweep_bunny_coop.txt)
        hipLaunchKernelGGL((render_bunny_coop_kernel<2>), dim3(((P.tile_count + 1u) / 2u) * uint32_t(B.count)),
                           dim3(BLOCK), 0, stream, B);
        return hipGetLastError();
    }
    if (P.round_steps > 0) {  // the throughput path: re-queued rays, four lanes per ray
        if (P.group_tiles >= 2) {
            hipLaunchKernelGGL((render_group_kernel<GROUP_KIFS, PRIM_BUNNY, 2>),
                               dim3(((P.tile_count + 1u) / 2u) * uint32_t(B.count)), dim3(BLOCK), 0, stream, B);
        } else {
            hipLaunchKernelGGL((render_group_kernel<GROUP_KIFS, PRIM_BUNNY, 1>), dim3(P.tile_count * uint32_t(B.count)),
                               dim3(BLOCK), 0, stream, B);
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(render_bunny_quad_kernel, dim3(B.frame.tile_count * 4u * uint32_t(B.count)), dim3(BLOCK), 0,
                       stream, B);
    return hipGetLastError();
}

hipError_t launch_render(const BatchParams& B, uint32_t group, uint32_t primitive,
                         hipStream_t stream) {
    const FrameParams& P = B.frame;
    if (P.y1 <= P.y0 || P.width <= 0 || P.tile_count == 0) return hipSuccess;
    if (B.count < 1 || B.count > MAX_BATCH) return hipErrorInvalidValue;
    switch (group) {
    case GROUP_JULIA:  // two builds of the long-ray loop, see KIFS_DIVSQRT_ORDINARY in kifs_scene.hpp
        return P.sdf_iters <= 24 ? launch_variant<GROUP_JULIA, 1>(B, stream)
                                 : launch_variant<GROUP_JULIA, 0>(B, stream);
    case GROUP_GENJULIA: return launch_variant<GROUP_GENJULIA, 0>(B, stream);
    case GROUP_KIFS:
        switch (primitive) {
        case PRIM_SPHERE: return launch_variant<GROUP_KIFS, PRIM_SPHERE>(B, stream);
        case PRIM_CYLINDER: return launch_variant<GROUP_KIFS, PRIM_CYLINDER>(B, stream);
        case PRIM_BOX: return launch_variant<GROUP_KIFS, PRIM_BOX>(B, stream);
        case PRIM_TORUS: return launch_variant<GROUP_KIFS, PRIM_TORUS>(B, stream);
        case PRIM_SIERPINSKI: return launch_variant<GROUP_KIFS, PRIM_SIERPINSKI>(B, stream);
        case PRIM_BUNNY: return launch_bunny_quad(B, stream);
        default: return launch_variant<GROUP_KIFS, PRIM_OTHER>(B, stream);  // kifs.wgsl:154
        }
    default: return hipErrorInvalidValue;
    }
}

// ---- tile order from the previous frame's costs -----------------------------------------
// One 1024-thread workgroup: histogram of clamped costs (bin 0 = heaviest), exclusive scan,
// scatter.  Whatever the cost values are, the result is a permutation of the tile ids, so a
// stale or garbage cost table can only cost speed, never pixels.
__global__ __launch_bounds__(1024) void tile_order_kernel(uint32_t* __restrict__ cost,
                                                          uint32_t* __restrict__ order, uint32_t n,
                                                          uint32_t tiles_x, uint32_t shift) {
    constexpr uint32_t BINS = 1024, LAST = BINS - 1;  // bin 0 = heaviest, LAST = cost 0
    __shared__ uint32_t bins[BINS];
    __shared__ uint32_t wave_total[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    bins[tid] = 0;
    __syncthreads();
    const uint32_t rounds = (n + 1023u) / 1024u;  // wave-uniform trip count (ballots inside)
    // After the culls almost every tile has cost 0: that class is counted with one atomic per
    // wave (ballot + popcount) instead of 64 atomics on the same LDS word.
    for (uint32_t k = 0; k < rounds; ++k) {
        const uint32_t i = k * 1024u + tid;
        const bool live = i < n;
        const uint32_t bin = live ? LAST - min(cost[i] >> shift, LAST) : 0u;
        const bool zero = live && bin == LAST;
        const unsigned long long zmask = __builtin_amdgcn_ballot_w64(zero);
        if (lane == 0 && zmask) atomicAdd(&bins[LAST], uint32_t(__builtin_popcountll(zmask)));
        if (live && !zero) atomicAdd(&bins[bin], 1u);
    }
    __syncthreads();
    const uint32_t mine = bins[tid];
    uint32_t incl = mine;  // inclusive scan inside the wave
#pragma unroll
    for (uint32_t off = 1; off < 64; off <<= 1) {
        uint32_t up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    if (lane == 63u) wave_total[w] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t k = 0; k < w; ++k) base += wave_total[k];
    bins[tid] = base + incl - mine;  // exclusive prefix = first slot of this bin
    __syncthreads();
    for (uint32_t k = 0; k < rounds; ++k) {
        const uint32_t i = k * 1024u + tid;
        const bool live = i < n;
        const uint32_t bin = live ? LAST - min(cost[i] >> shift, LAST) : 0u;
        const bool zero = live && bin == LAST;
        const unsigned long long zmask = __builtin_amdgcn_ballot_w64(zero);
        uint32_t zbase = 0;
        if (lane == 0 && zmask) zbase = atomicAdd(&bins[LAST], uint32_t(__builtin_popcountll(zmask)));
        zbase = __shfl(zbase, 0);
        uint32_t pos;
        if (zero) pos = zbase + uint32_t(__builtin_popcountll(zmask & ((1ull << lane) - 1ull)));
        else if (live) pos = atomicAdd(&bins[bin], 1u);
        if (live) {
            order[pos] = (i % tiles_x) | ((i / tiles_x) << 16);
            cost[i] = 0;  // ready for the next recording launch (batches accumulate with atomicMax)
        }
    }
}

hipError_t launch_tile_order(uint32_t* cost, uint32_t* order, uint32_t tile_count,
                             uint32_t tiles_x, uint32_t shift, hipStream_t stream) {
    if (tile_count == 0) return hipSuccess;
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, stream, cost, order, tile_count, tiles_x, shift);
    return hipGetLastError();
}

// ---- row shards: packed stripes -> frame rows (the root's side of the multi-GPU gather) ----
// Workgroup (s, f): stripe s of shard f.  Row k of stripe s sits at packed row 8 s + k and goes to
// frame row stripe_rows[s] + k.  Rows are copied 16 bytes per lane when everything is aligned.
__global__ __launch_bounds__(256) void unpack_stripes_kernel(
    uint8_t* __restrict__ dst, size_t dst_pitch, size_t dst_frame_stride, const uint8_t* __restrict__ src,
    size_t src_pitch, size_t src_shard_stride, const uint32_t* __restrict__ stripe_rows, int row_bytes,
    int height, int vec16) {
    const uint32_t s = blockIdx.x, f = blockIdx.y;
    const int y0 = int(stripe_rows[s]);
    const int rows = min(TILE_H, height - y0);
    const uint8_t* from = src + size_t(f) * src_shard_stride + size_t(s) * TILE_H * src_pitch;
    uint8_t* to = dst + size_t(f) * dst_frame_stride + size_t(y0) * dst_pitch;
    if (vec16) {
        const int per_row = row_bytes >> 4;
        for (int i = threadIdx.x; i < rows * per_row; i += 256) {
            const int r = i / per_row, c = i - r * per_row;
            reinterpret_cast<uint4*>(to + size_t(r) * dst_pitch)[c] =
                reinterpret_cast<const uint4*>(from + size_t(r) * src_pitch)[c];
        }
    } else {
        const int per_row = row_bytes >> 2;
        for (int i = threadIdx.x; i < rows * per_row; i += 256) {
            const int r = i / per_row, c = i - r * per_row;
            reinterpret_cast<uint32_t*>(to + size_t(r) * dst_pitch)[c] =
                reinterpret_cast<const uint32_t*>(from + size_t(r) * src_pitch)[c];
        }
    }
}

hipError_t launch_unpack_stripes(uint8_t* dst, size_t dst_pitch, size_t dst_frame_stride, const uint8_t* src,
                                 size_t src_pitch, size_t src_shard_stride, const uint32_t* stripe_rows,
                                 int n_stripes, int count, int width, int height, hipStream_t stream) {
    if (n_stripes <= 0 || count <= 0) return hipSuccess;
    const int row_bytes = width * 4;
    const uintptr_t all = reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | dst_pitch |
                          dst_frame_stride | src_pitch | src_shard_stride | uintptr_t(row_bytes);
    hipLaunchKernelGGL(unpack_stripes_kernel, dim3(uint32_t(n_stripes), uint32_t(count)), dim3(256), 0, stream, dst,
                       dst_pitch, dst_frame_stride, src, src_pitch, src_shard_stride, stripe_rows, row_bytes, height,
                       (all & 15u) == 0 ? 1 : 0);
    return hipGetLastError();
}

// ---- sparse shards: a peer's packed shards without their background tiles ----------------------
// A 1080p frame of these scenes is nine tenths background, and the root of a gather takes every
// peer's rows over ONE xGMI link each: the link, not the rendering, would set the rate.  So a peer
// sends only the 32 x 8 tiles that hold a pixel other than the background, as records of
// SPARSE_RECORD_WORDS words -- [tile id, 0, 0, 0, 256 pixels row by row] -- and the root fills the
// rest with the background itself.  Tile id = (shard * n_stripes + stripe slot) * tiles_x + column.
// Lossless whatever the frame holds: a frame without background costs 1.6 % more than the dense form.
//
// pack: a workgroup takes 16 consecutive tiles, four per wave (lane -> row lane >> 3, four pixels from
// column 4 (lane & 7)); pixels outside the frame count, and are written, as background.  One atomic per
// workgroup reserves its records: their order in the payload is arbitrary, the ids say what they are.
constexpr int SPARSE_RECORD_WORDS = SPARSE_RECORD_WORDS_HOST;
__global__ __launch_bounds__(256) void pack_sparse_kernel(
    const uint8_t* __restrict__ src, size_t src_pitch, size_t src_shard_stride, const uint32_t* __restrict__ stripe_rows,
    int n_stripes, int count, int width, int height, uint32_t background, uint32_t* __restrict__ records,
    uint32_t* __restrict__ n_records, int vec16) {
    __shared__ uint32_t s_wave_count[4], s_base;
    const uint32_t tiles_x = uint32_t(width + TILE_W - 1) / TILE_W;
    const uint32_t total = uint32_t(count) * uint32_t(n_stripes) * tiles_x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t first = (blockIdx.x * 4u + wave) * 4u;
    const int row = int(lane >> 3), col = int(lane & 7u) * 4;
    uint32_t px[4][4];
    uint32_t mask = 0;  // wave-uniform: bit j = tile first + j holds something
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t id = first + uint32_t(j);
#pragma unroll
        for (int q = 0; q < 4; ++q) px[j][q] = background;
        if (id < total) {
            const uint32_t tx = id % tiles_x, sk = (id / tiles_x) % uint32_t(n_stripes), shard = id / (tiles_x * uint32_t(n_stripes));
            const int rows = min(TILE_H, height - int(stripe_rows[sk]));
            const int x = int(tx) * TILE_W + col;
            if (row < rows) {
                const uint32_t* from = reinterpret_cast<const uint32_t*>(src + size_t(shard) * src_shard_stride +
                                                                         (size_t(sk) * TILE_H + size_t(row)) * src_pitch) + x;
                if (vec16 && x + 3 < width) {  // rows and pitches 16-byte aligned: one load per lane
                    const uint4 v = *reinterpret_cast<const uint4*>(from);
                    px[j][0] = v.x; px[j][1] = v.y; px[j][2] = v.z; px[j][3] = v.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (x + q < width) px[j][q] = from[q];
                }
            }
        }
        const bool differs = (px[j][0] != background) || (px[j][1] != background) || (px[j][2] != background) ||
                             (px[j][3] != background);
        if (__builtin_amdgcn_ballot_w64(differs) != 0ull) mask |= 1u << j;
    }
    if (lane == 0) s_wave_count[wave] = uint32_t(__builtin_popcount(mask));
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t n = s_wave_count[0] + s_wave_count[1] + s_wave_count[2] + s_wave_count[3];
        s_base = n ? atomicAdd(n_records, n) : 0u;
    }
    __syncthreads();
    uint32_t at = s_base;
    for (uint32_t w = 0; w < wave; ++w) at += s_wave_count[w];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!(mask & (1u << j))) continue;
        uint32_t* rec = records + size_t(at) * SPARSE_RECORD_WORDS;
        if (lane < 4u) rec[lane] = lane == 0u ? first + uint32_t(j) : 0u;
        *reinterpret_cast<uint4*>(rec + 4 + row * TILE_W + col) = make_uint4(px[j][0], px[j][1], px[j][2], px[j][3]);
        ++at;
    }
}

// unpack: one wave per record; the tile goes to its frame rows, clipped to the frame.  Ids that do not
// belong to the shard are skipped (the payload crossed a network).
//   erase != 0: the record's tile is overwritten with `background` instead (a frame buffer that is reused
//   needs the background back only where the previous frame's records went, not everywhere).
__global__ __launch_bounds__(256) void unpack_sparse_kernel(
    uint8_t* __restrict__ dst, size_t dst_pitch, size_t dst_frame_stride, const uint32_t* __restrict__ records,
    uint32_t n_records, const uint32_t* __restrict__ stripe_rows, int n_stripes, int count, int width, int height,
    int erase, uint32_t background) {
    const uint32_t r = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (r >= n_records) return;
    const uint32_t* rec = records + size_t(r) * SPARSE_RECORD_WORDS;
    const uint32_t tiles_x = uint32_t(width + TILE_W - 1) / TILE_W;
    const uint32_t id = rec[0];
    if (id >= uint32_t(count) * uint32_t(n_stripes) * tiles_x) return;
    const uint32_t tx = id % tiles_x, sk = (id / tiles_x) % uint32_t(n_stripes), shard = id / (tiles_x * uint32_t(n_stripes));
    const int row = int(lane >> 3), col = int(lane & 7u) * 4;
    const int y = int(stripe_rows[sk]) + row, x = int(tx) * TILE_W + col;
    if (y >= height || row >= TILE_H) return;
    const uint4 v = erase ? make_uint4(background, background, background, background)
                          : *reinterpret_cast<const uint4*>(rec + 4 + row * TILE_W + col);
    uint32_t* to = reinterpret_cast<uint32_t*>(dst + size_t(shard) * dst_frame_stride + size_t(y) * dst_pitch) + x;
    const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (x + k < width) to[k] = q[k];
}

// fill: the background over the rows of the listed stripes (what the records leave out).
__global__ __launch_bounds__(256) void fill_stripes_kernel(uint8_t* __restrict__ dst, size_t dst_pitch, size_t dst_frame_stride,
                                                            const uint32_t* __restrict__ stripe_rows, int width, int height,
                                                            uint32_t background, int vec16) {
    const uint32_t s = blockIdx.x, f = blockIdx.y;
    const int y0 = int(stripe_rows[s]);
    const int rows = min(TILE_H, height - y0);
    uint8_t* to = dst + size_t(f) * dst_frame_stride + size_t(y0) * dst_pitch;
    if (vec16) {
        const int per_row = width >> 2;
        const uint4 v = make_uint4(background, background, background, background);
        for (int i = threadIdx.x; i < rows * per_row; i += 256) {
            const int r = i / per_row, c = i - r * per_row;
            reinterpret_cast<uint4*>(to + size_t(r) * dst_pitch)[c] = v;
        }
    } else {
        for (int i = threadIdx.x; i < rows * width; i += 256) {
            const int r = i / width, c = i - r * width;
            reinterpret_cast<uint32_t*>(to + size_t(r) * dst_pitch)[c] = background;
        }
    }
}

hipError_t launch_pack_sparse(const uint8_t* src, size_t src_pitch, size_t src_shard_stride, const uint32_t* stripe_rows,
                              int n_stripes, int count, int width, int height, uint32_t background, uint32_t* records,
                              uint32_t* n_records, hipStream_t stream) {
    if (n_stripes <= 0 || count <= 0) return hipSuccess;
    const uint64_t tiles = uint64_t(count) * uint64_t(n_stripes) * uint64_t((width + TILE_W - 1) / TILE_W);
    const uintptr_t all = reinterpret_cast<uintptr_t>(src) | src_pitch | src_shard_stride;
    hipLaunchKernelGGL(pack_sparse_kernel, dim3(uint32_t((tiles + 15u) / 16u)), dim3(256), 0, stream, src, src_pitch,
                       src_shard_stride, stripe_rows, n_stripes, count, width, height, background, records, n_records,
                       (all & 15u) == 0 ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_unpack_sparse(uint8_t* dst, size_t dst_pitch, size_t dst_frame_stride, const uint32_t* records,
                                uint32_t n_records, const uint32_t* stripe_rows, int n_stripes, int count, int width,
                                int height, int erase, uint32_t background, hipStream_t stream) {
    if (n_records == 0 || n_stripes <= 0 || count <= 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_sparse_kernel, dim3((n_records + 3u) / 4u), dim3(256), 0, stream, dst, dst_pitch,
                       dst_frame_stride, records, n_records, stripe_rows, n_stripes, count, width, height, erase, background);
    return hipGetLastError();
}

hipError_t launch_fill_stripes(uint8_t* dst, size_t dst_pitch, size_t dst_frame_stride, const uint32_t* stripe_rows,
                               int n_stripes, int count, int width, int height, uint32_t background, hipStream_t stream) {
    if (n_stripes <= 0 || count <= 0) return hipSuccess;
    const uintptr_t all = reinterpret_cast<uintptr_t>(dst) | dst_pitch | dst_frame_stride | uintptr_t(width * 4);
    hipLaunchKernelGGL(fill_stripes_kernel, dim3(uint32_t(n_stripes), uint32_t(count)), dim3(256), 0, stream, dst, dst_pitch,
                       dst_frame_stride, stripe_rows, width, height, background, (all & 15u) == 0 ? 1 : 0);
    return hipGetLastError();
}

// ---- point evaluation (parity tests) --------------------------------------------------
template <int GROUP, int PRIM>
__global__ void eval_points_kernel(const FrameParams P, const float* __restrict__ pts, int n,
                                   float* __restrict__ sdf, float* __restrict__ nrm) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 p{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    if (sdf) sdf[i] = scene_sdf<GROUP, PRIM>(P, p);
    if (nrm) {
        V3 v = scene_normal<GROUP, PRIM>(P, p);
        nrm[3 * i] = v.x; nrm[3 * i + 1] = v.y; nrm[3 * i + 2] = v.z;
    }
}

template <int GROUP, int PRIM>
static hipError_t launch_eval_variant(const FrameParams& P, const float* pts, int n, float* sdf,
                                      float* nrm, hipStream_t stream) {
    hipLaunchKernelGGL((eval_points_kernel<GROUP, PRIM>), dim3((n + 255) / 256), dim3(256), 0,
                       stream, P, pts, n, sdf, nrm);
    return hipGetLastError();
}

hipError_t launch_eval_points(const FrameParams& P, uint32_t group, uint32_t primitive,
                              const float* pts, int n, float* sdf, float* nrm,
                              hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    switch (group) {
    case GROUP_JULIA: return launch_eval_variant<GROUP_JULIA, 0>(P, pts, n, sdf, nrm, stream);
    case GROUP_GENJULIA: return launch_eval_variant<GROUP_GENJULIA, 0>(P, pts, n, sdf, nrm, stream);
    case GROUP_KIFS:
        switch (primitive) {
        case PRIM_SPHERE: return launch_eval_variant<GROUP_KIFS, PRIM_SPHERE>(P, pts, n, sdf, nrm, stream);
        case PRIM_CYLINDER: return launch_eval_variant<GROUP_KIFS, PRIM_CYLINDER>(P, pts, n, sdf, nrm, stream);
        case PRIM_BOX: return launch_eval_variant<GROUP_KIFS, PRIM_BOX>(P, pts, n, sdf, nrm, stream);
        case PRIM_TORUS: return launch_eval_variant<GROUP_KIFS, PRIM_TORUS>(P, pts, n, sdf, nrm, stream);
        case PRIM_SIERPINSKI: return launch_eval_variant<GROUP_KIFS, PRIM_SIERPINSKI>(P, pts, n, sdf, nrm, stream);
        case PRIM_BUNNY: return launch_eval_variant<GROUP_KIFS, PRIM_BUNNY>(P, pts, n, sdf, nrm, stream);
        default: return launch_eval_variant<GROUP_KIFS, PRIM_OTHER>(P, pts, n, sdf, nrm, stream);
        }
    default: return hipErrorInvalidValue;
    }
}

__global__ void eval_math_kernel(int fn, const float* __restrict__ in, float param,
                                 const float* __restrict__ srgb_table, float* __restrict__ out,
                                 int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = in[i], r;
    switch (fn) {
    case 0: r = log_(x); break;
    case 1: r = log2_(x); break;
    case 2: r = exp2_(x); break;
    case 3: r = sin_(x); break;
    case 4: r = cos_(x); break;
    case 5: r = acos_(x); break;
    case 6: r = pow_(x, param); break;
    case 7: r = float(srgb8(x, srgb_table)); break;
    case 8: r = float(unorm8(x)); break;
    case 9: r = rcp_mid(x); break;
    case 10: r = sqrt_mid(x); break;
    case 11: r = sin_flat(x); break;
    default: r = x; break;
    }
    out[i] = r;
}

hipError_t launch_eval_math(int fn, const float* in, float param, const float* srgb_table,
                            float* out, int n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(eval_math_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, fn, in,
                       param, srgb_table, out, n);
    return hipGetLastError();
}

}  // namespace kifs

#ifdef KIFS_EVAL_COUNT
extern "C" int kifs_debug_eval_counts(unsigned long long* out8, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(kifs::g_eval_counts), 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(kifs::g_eval_counts), z, 64) != hipSuccess) return -1;
    }
    return 0;
}
#endif
