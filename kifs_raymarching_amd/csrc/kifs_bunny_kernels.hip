// kifs_bunny_kernels.hip -- the bunny primitive's own render kernels (kifs.wgsl:84-137; the network itself:
// bunny_sdf_quad / bunny_sdf_coop in kifs_bunny.hpp).  Its mid-size launches use render_group_kernel<KIFS, BUNNY, T>
// of kifs_kernels.hip (four lanes per ray); which form a launch gets: rules::BUNNY_* in kifs_schedule.cpp.
#include "kifs_render_common.hpp"

namespace kifs {

// The bunny primitive with four lanes per pixel (see bunny_sdf_quad in kifs_bunny.hpp): a
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
        rgba = encode_rgba(colour, srgb, s_srgb);
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
// (bunny_sdf_coop in kifs_bunny.hpp: weights as scalar operands, activations exchanged through LDS).  All four
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
        const uint32_t rgba8 = encode_rgba(colour, srgb, s_srgb);
        if (wave == 0 && i0 + uint32_t(lane) < hits) s_tile[pix >> 8][hy][hx] = rgba8;
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

hipError_t launch_bunny_coop(const BatchParams& B, hipStream_t stream) {
    const FrameParams& P = B.frame;
    // (always two tiles per workgroup: one was slower at every batch size where this form wins at all --
    // 48 frames per launch 46.5 against 51.8 Gpixel/s, profiles/r03/sweep_bunny_coop.txt)
    hipLaunchKernelGGL((render_bunny_coop_kernel<2>), dim3(((P.tile_count + 1u) / 2u) * uint32_t(B.count)), dim3(BLOCK), 0,
                       stream, B);
    return hipGetLastError();
}

hipError_t launch_bunny_whole_rays(const BatchParams& B, hipStream_t stream) {
    hipLaunchKernelGGL(render_bunny_quad_kernel, dim3(B.frame.tile_count * 4u * uint32_t(B.count)), dim3(BLOCK), 0, stream, B);
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

