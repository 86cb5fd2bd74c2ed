// kifs_support_kernels.hip -- the kernels around the render kernels: the tile order from recorded costs, the row
// shards' packing / unpacking / filling (the multi-GPU gather's device side), and the point / elementary-function
// evaluation behind kifs_eval_points / kifs_eval_math (parity tooling).
#include "kifs_render_common.hpp"

namespace kifs {

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
    case 12: r = exp2_core(x); break;
    case 13: r = log2_core(x); break;
    case 14: { float sn, cs; sincos_core(x, sn, cs); r = sn; break; }
    case 15: { float sn, cs; sincos_core(x, sn, cs); r = cs; break; }
    case 16: r = acos_core(x); break;
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
