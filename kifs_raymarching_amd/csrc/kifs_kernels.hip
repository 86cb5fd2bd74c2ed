// kifs_kernels.hip -- the render kernels and their dispatch.  (The bunny's two own kernels: kifs_bunny_kernels.hip;
// tile order, shard packing and the parity tooling's kernels: kifs_support_kernels.hip; shared helpers:
// kifs_render_common.hpp.)
//
// Render kernels over 32 x 8 pixel tiles taken from a tile ORDER table, all fed by the kernel
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
//   render_bunny_quad_kernel            the bunny primitive, four lanes per pixel       } kifs_bunny_kernels.hip
//   render_bunny_coop_kernel            the bunny's batches, four waves per 64 rays     }
// (which one a launch gets: enqueue_batch in kifs_schedule.cpp, from the projected-disc tile count)
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
#include <atomic>
#include <cstdlib>

#include "kifs_render_common.hpp"

namespace kifs {

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
        rgba = encode_rgba(colour, srgb, s_srgb);
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
template <int GROUP, int PRIM, int T, bool W2LDS = false>
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
    // The chunks of a round are DRAWN, not dealt: a wave takes the next chunk of the queue from a ticket counter when it
    // has finished its last one.  A chunk of rays deep inside the bounding sphere costs several times a chunk of rays on
    // their way in (gen-Julia: 190 instructions per orbit trip against 45 for a step outside), and dealt in turn
    // (wave w: chunks w, w + 4, ..) the waves with the cheap ones waited at the round's barrier -- SQ_WAIT_ANY was 54 %
    // of the gen-Julia kernel's wave-cycles, 48 % of the bunny's (profiles/r04/stalls.json).  Same rotation as q_count.
    __shared__ uint32_t q_ticket[3];
    static_assert(!W2LDS || (GROUP == GROUP_KIFS && PRIM == PRIM_BUNNY), "W2LDS is a form of the bunny's network");
    __shared__ float s_w2[W2LDS ? 256 : 1];  // (bunny, W2LDS: layer 2 of the network for the four column groups, see kifs_bunny.hpp)

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
        q_ticket[0] = 0;
        q_ticket[1] = 0;
        q_ticket[2] = 0;
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
    if constexpr (W2LDS) bunny_w2_stage(s_w2, tid);
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
    BunnyQuadT<W2LDS> W;  // (bunny only) this lane's column group of the network, loaded with the wave's first rays
    bool weights_loaded = false;
    int trips = 0;
    for (uint32_t cur = 0, cnt = 0;; cur ^= 1u, cnt = (cnt + 1u) % 3u) {
        const uint32_t n = q_count[cnt];  // uniform
        if (n == 0u) break;
        const uint32_t cnt_next = (cnt + 1u) % 3u;
        if (tid == 0) {  // the counters of the round after next
            q_count[(cnt + 2u) % 3u] = 0;
            q_ticket[(cnt + 2u) % 3u] = 0;
        }
        // (one chunk left: nothing more to merge, it is marched to the end -- see render_wave_kernel; not the
        // generalised Julia set, whose compiled march loses 5-10 % that way: 48 frames per launch 20.6 -> 19.6 Gpixel/s)
        const int limit = (GROUP != GROUP_GENJULIA && n <= RAYS) ? P.max_iterations : min(trips + P.round_steps, P.max_iterations);
        for (;;) {
            uint32_t chunk = 0;
            if (lane == 0) chunk = atomicAdd(&q_ticket[cnt], 1u);
            chunk = __builtin_amdgcn_readfirstlane(chunk);
            if (chunk * RAYS >= n) break;
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
                    bunny_quad_load(W, lane & 3, s_w2);
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
        if (hits != 0u && !weights_loaded) bunny_quad_load(W, lane & 3, s_w2);  // a wave that marched nothing shades too
    }
    bool shaded = false;
    if constexpr (!BUNNY) {
        // Soft shadows: the secondary rays from a pool, as in render_wave_kernel (shade_hits_with_pooled_shadows), the pool
        // being the workgroup's: a barrier between reading a chunk of 256 hits and filing its secondary rays over the hit
        // list's consumed slots, an LDS cursor the four waves draw from.  Same operations per ray: same pixels.
        if (P.soft_shadow != 0u && P.shadow_steps > 0) {
            shaded = true;
            __shared__ uint32_t sh_count, sh_next;
            constexpr int NPRIM = GROUP == GROUP_JULIA ? 0 : PRIM;  // (the Julia pipeline's PRIM slot is its loop's variant)
            const unsigned long long below = (1ull << lane) - 1ull;
            const V3 L = normalize(V3{1.0f, 1.0f, 1.0f});
            const float off = 2.0f * P.epsilon;
            if (tid == 0) {
                sh_count = 0;
                sh_next = 0;
            }
            // pass 1: normals and direct terms; the hits that need no secondary ray get their colour
            for (uint32_t i0 = 0; i0 < hits; i0 += uint32_t(BLOCK)) {
                const uint32_t i = i0 + uint32_t(tid);
                const bool have = i < hits;
                const uint32_t pix = have ? h_pix[i] : 0u;
                const float t = have ? h_t[i] : 0.0f;
                __syncthreads();  // the chunk's entries are read (and the counters cleared) before any slot is filed over
                float lit = 0.0f;
                V3 start{0.0f, 0.0f, 0.0f};
                if (have) {
                    const V3 dir = V3{s_dir[0][pix], s_dir[1][pix], s_dir[2][pix]};
                    const V3 p = (t == 0.0f) ? P.origin
                                             : V3{fmaf_(t, dir.x, P.origin.x), fmaf_(t, dir.y, P.origin.y),
                                                  fmaf_(t, dir.z, P.origin.z)};
                    const V3 nrm = scene_normal<GROUP, NPRIM>(P, p);
                    const float ndl = (nrm.x + nrm.y) + nrm.z;
                    lit = clamp_(ndl, 0.0f, 1.0f);
                    start = V3{fmaf_(off, nrm.x, p.x), fmaf_(off, nrm.y, p.y), fmaf_(off, nrm.z, p.z)};
                }
                const bool secondary = have && (lit > 0.0f);
                const unsigned long long ms = __builtin_amdgcn_ballot_w64(secondary);
                if (ms != 0ull) {
                    const int first = __builtin_ctzll(ms);
                    uint32_t base = 0;
                    if (lane == first) base = atomicAdd(&sh_count, uint32_t(__builtin_popcountll(ms)));
                    base = uint32_t(__builtin_amdgcn_readlane(int(base), first));
                    if (secondary) {  // (slot k <= the hits read so far: every one of them is in a register)
                        const uint32_t k = base + uint32_t(__builtin_popcountll(ms & below));
                        h_pix[k] = pix;
                        h_t[k] = lit;
                        s_dir[0][pix] = start.x;
                        s_dir[1][pix] = start.y;
                        s_dir[2][pix] = start.z;
                    }
                }
                if (have && !secondary) {
                    const float diffuse = fmaf_(0.9f, lit, 0.1f);
                    const V3 colour{diffuse * P.fractal_color.x, diffuse * P.fractal_color.y, diffuse * P.fractal_color.z};
                    s_tile[pix >> 8][(pix >> 5) & 7u][pix & 31u] = encode_rgba(colour, srgb, s_srgb);
                }
            }
            __syncthreads();
            // pass 2: every lane marches a secondary ray of the pool and takes the next one when it ends
            {
                const uint32_t n_sh = sh_count;  // uniform
                bool busy = false, exhausted = false;
                uint32_t k = 0;
                int j = 0;
                float lit = 0.0f, t = 0.0f, res = 1.0f;
                V3 start{0.0f, 0.0f, 0.0f};
                for (;;) {
                    const unsigned long long idle = __builtin_amdgcn_ballot_w64(!busy);
                    if (idle != 0ull && !exhausted) {
                        const int first = __builtin_ctzll(idle);
                        uint32_t base = 0;
                        if (lane == first) base = atomicAdd(&sh_next, uint32_t(__builtin_popcountll(idle)));
                        base = uint32_t(__builtin_amdgcn_readlane(int(base), first));
                        const uint32_t mine = base + uint32_t(__builtin_popcountll(idle & below));
                        if (!busy && mine < n_sh) {
                            k = mine;
                            const uint32_t pix = h_pix[k];
                            lit = h_t[k];
                            start = V3{s_dir[0][pix], s_dir[1][pix], s_dir[2][pix]};
                            t = P.shadow_t0;
                            res = 1.0f;
                            j = 0;
                            busy = true;
                        }
                        exhausted = base + uint32_t(__builtin_popcountll(idle)) >= n_sh;
                    }
                    const unsigned long long lanes = __builtin_amdgcn_ballot_w64(busy);
                    if (lanes == 0ull) break;
                    const V3 q{fmaf_(t, L.x, start.x), fmaf_(t, L.y, start.y), fmaf_(t, L.z, start.z)};
                    const float h = scene_sdf<GROUP, NPRIM>(P, q, lanes);
                    if (busy) {
                        bool done;
                        if (h < P.epsilon) {
                            res = 0.0f;
                            done = true;
                        } else {
                            res = min_(res, (P.shadow_k * h) / t);
                            t = t + h;
                            done = (t > P.shadow_max_t) || !(j + 1 < P.shadow_steps);
                            ++j;
                        }
                        if (done) {
                            h_t[k] = lit * res;  // the attenuated direct term
                            busy = false;
                        }
                    }
                }
            }
            __syncthreads();
            // pass 3: their colours
            const uint32_t n_sh = sh_count;
            for (uint32_t k0 = 0; k0 < n_sh; k0 += uint32_t(BLOCK)) {
                const uint32_t k = k0 + uint32_t(tid);
                if (k < n_sh) {
                    const uint32_t pix = h_pix[k];
                    const float diffuse = fmaf_(0.9f, h_t[k], 0.1f);
                    const V3 colour{diffuse * P.fractal_color.x, diffuse * P.fractal_color.y, diffuse * P.fractal_color.z};
                    s_tile[pix >> 8][(pix >> 5) & 7u][pix & 31u] = encode_rgba(colour, srgb, s_srgb);
                }
            }
        }
    }
    for (uint32_t i0 = 0; !shaded && i0 < hits; i0 += uint32_t(BLOCK) / LPR) {
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
            const uint32_t rgba8 = encode_rgba(colour, srgb, s_srgb);
            if ((uint32_t(tid) % LPR) == 0u) s_tile[pix >> 8][hy][hx] = rgba8;
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

// Shading of a tile's hits with the soft-shadow extension (include/kifs_hip.h), for render_wave_kernel (every pipeline).
// A hit's secondary march is 1 to shadow_steps estimates long, so 64 of them side by side finish one by one and the
// wave waits for the longest.  Instead the lanes take the secondary rays from a pool: pass 1 computes every hit's
// normal and direct term, colours the hits that need no secondary ray and files the others (start point over the
// pixel's cached direction, which nothing reads any more; pixel id and direct term over the hit list's slots already
// consumed); pass 2 marches them with per-lane step counters, a lane that finishes its ray taking the next one of the
// pool; pass 3 encodes.  Every ray's own sequence of operations is soft_shadow()'s: same pixels.
// hit_pix / hit_val: the hit list, entry i at [CAP - 1 - i] (pixel id, t); dir: the per-pixel direction cache.
template <int GROUP, int PRIM, uint32_t CAP>
__device__ __forceinline__ void shade_hits_with_pooled_shadows(const FrameParams& P, uint32_t hits, uint32_t lane, uint8_t* hit_pix,
                                                               float* hit_val, float (*dir_cache)[CAP], uint32_t (*s_tile)[TILE_W],
                                                               bool srgb, const float* s_srgb) {
    const unsigned long long below = (1ull << lane) - 1ull;
    const V3 L = normalize(V3{1.0f, 1.0f, 1.0f});
    const float off = 2.0f * P.epsilon;
    uint32_t n_sh = 0;  // uniform
    for (uint32_t i0 = 0; i0 < hits; i0 += 64u) {
        const bool have = i0 + lane < hits;
        uint32_t pix = 0;
        float lit = 0.0f;
        V3 start{0.0f, 0.0f, 0.0f};
        if (have) {
            pix = hit_pix[CAP - 1u - (i0 + lane)];
            const float t = hit_val[CAP - 1u - (i0 + lane)];
            const V3 dir = V3{dir_cache[0][pix], dir_cache[1][pix], dir_cache[2][pix]};
            const V3 p = (t == 0.0f) ? P.origin
                                     : V3{fmaf_(t, dir.x, P.origin.x), fmaf_(t, dir.y, P.origin.y),
                                          fmaf_(t, dir.z, P.origin.z)};
            const V3 nrm = scene_normal<GROUP, PRIM>(P, p);
            const float ndl = (nrm.x + nrm.y) + nrm.z;
            lit = clamp_(ndl, 0.0f, 1.0f);
            start = V3{fmaf_(off, nrm.x, p.x), fmaf_(off, nrm.y, p.y), fmaf_(off, nrm.z, p.z)};
        }
        const bool secondary = have && (lit > 0.0f);
        const unsigned long long ms = __builtin_amdgcn_ballot_w64(secondary);
        if (secondary) {  // (slot k <= this hit's own: read above, by every lane, before anything is written)
            const uint32_t k = n_sh + uint32_t(__builtin_popcountll(ms & below));
            hit_pix[CAP - 1u - k] = uint8_t(pix);
            hit_val[CAP - 1u - k] = lit;
            dir_cache[0][pix] = start.x;
            dir_cache[1][pix] = start.y;
            dir_cache[2][pix] = start.z;
        } else if (have) {  // no direct light: the colour is final
            const float diffuse = fmaf_(0.9f, lit, 0.1f);
            const V3 colour{diffuse * P.fractal_color.x, diffuse * P.fractal_color.y, diffuse * P.fractal_color.z};
            s_tile[pix >> 5][pix & 31u] = encode_rgba(colour, srgb, s_srgb);
        }
        n_sh += uint32_t(__builtin_popcountll(ms));
    }
    // pass 2: the pool of secondary rays
    {
        uint32_t next = 0;  // uniform: rays handed out so far
        bool busy = false;
        uint32_t k = 0;
        int j = 0;
        float lit = 0.0f, t = 0.0f, res = 1.0f;
        V3 start{0.0f, 0.0f, 0.0f};
        for (;;) {
            const unsigned long long idle = __builtin_amdgcn_ballot_w64(!busy);
            const uint32_t mine = next + uint32_t(__builtin_popcountll(idle & below));
            if (!busy && mine < n_sh) {
                k = mine;
                const uint32_t pix = hit_pix[CAP - 1u - k];
                lit = hit_val[CAP - 1u - k];
                start = V3{dir_cache[0][pix], dir_cache[1][pix], dir_cache[2][pix]};
                t = P.shadow_t0;
                res = 1.0f;
                j = 0;
                busy = true;
            }
            next = min(n_sh, next + uint32_t(__builtin_popcountll(idle)));
            const unsigned long long lanes = __builtin_amdgcn_ballot_w64(busy);
            if (lanes == 0ull) break;
            const V3 q{fmaf_(t, L.x, start.x), fmaf_(t, L.y, start.y), fmaf_(t, L.z, start.z)};
            const float h = scene_sdf<GROUP, PRIM>(P, q, lanes);
            if (busy) {
                bool done;
                if (h < P.epsilon) {
                    res = 0.0f;
                    done = true;
                } else {
                    res = min_(res, (P.shadow_k * h) / t);
                    t = t + h;
                    done = (t > P.shadow_max_t) || !(j + 1 < P.shadow_steps);
                    ++j;
                }
                if (done) {
                    hit_val[CAP - 1u - k] = lit * res;  // the attenuated direct term
                    busy = false;
                }
            }
        }
    }
    // pass 3: their colours
    for (uint32_t k0 = 0; k0 < n_sh; k0 += 64u) {
        if (k0 + lane < n_sh) {
            const uint32_t pix = hit_pix[CAP - 1u - (k0 + lane)];
            const float lit = hit_val[CAP - 1u - (k0 + lane)];
            const float diffuse = fmaf_(0.9f, lit, 0.1f);
            const V3 colour{diffuse * P.fractal_color.x, diffuse * P.fractal_color.y, diffuse * P.fractal_color.z};
            s_tile[pix >> 5][pix & 31u] = encode_rgba(colour, srgb, s_srgb);
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
    bool shaded = false;
    if (P.soft_shadow != 0u && P.shadow_steps > 0) {  // secondary rays from a pool: see the function
        shaded = true;
        // (for the Julia pipeline the PRIM slot is the long-ray loop's variant: its SDF and normal do not depend on it)
        shade_hits_with_pooled_shadows<GROUP, GROUP == GROUP_JULIA ? 0 : PRIM, CAP>(P, hits, lane, &q_pix[1][0], &q_t[1][0], s_dir, s_tile,
                                                                                   srgb, s_srgb);
    }
    for (uint32_t i0 = 0; !shaded && i0 < hits; i0 += 64u) {
        if (i0 + lane < hits) {
            const uint32_t pix = q_pix[1][CAP - 1u - (i0 + lane)];
            const float t = q_t[1][CAP - 1u - (i0 + lane)];
            const int hx = int(pix & 31u), hy = int(pix >> 5);
            const V3 dir = V3{s_dir[0][pix], s_dir[1][pix], s_dir[2][pix]};
            const V3 p = (t == 0.0f) ? P.origin
                                     : V3{fmaf_(t, dir.x, P.origin.x), fmaf_(t, dir.y, P.origin.y),
                                          fmaf_(t, dir.z, P.origin.z)};
            const V3 colour = shade_hit<GROUP, PRIM>(P, p);
            s_tile[hy][hx] = encode_rgba(colour, srgb, s_srgb);
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
        // (two contexts on two threads may come through here at once -- the header allows one caller thread per
        // context -- hence atomics; setting the attribute twice is harmless, a torn flag would not be)
        static std::atomic<bool> opted_in[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
        if (dev < 0 || dev >= 64 || !opted_in[dev].load(std::memory_order_acquire)) {
            hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<GROUP, PRIM>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (attr != hipSuccess) return attr;
            if (dev >= 0 && dev < 64) opted_in[dev].store(true, std::memory_order_release);
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
    if (P.round_steps > 0 && P.bunny_coop == 1) return launch_bunny_coop(B, stream);  // re-queued rays, four waves per 64 rays
    if (P.round_steps > 0 && P.bunny_coop == 2) {  // four lanes per ray, layer 2 in LDS: three waves per SIMD (pairs of tiles only)
        hipLaunchKernelGGL((render_group_kernel<GROUP_KIFS, PRIM_BUNNY, 2, true>),
                           dim3(((P.tile_count + 1u) / 2u) * uint32_t(B.count)), dim3(BLOCK), 0, stream, B);
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
    return launch_bunny_whole_rays(B, stream);
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


}  // namespace kifs

#ifdef KIFS_EVAL_COUNT
extern "C" int kifs_debug_pow_counts(unsigned long long* out8, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(kifs::g_pow_counts), 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(kifs::g_pow_counts), z, 64) != hipSuccess) return -1;
    }
    return 0;
}
#endif
