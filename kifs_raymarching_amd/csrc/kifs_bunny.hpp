// kifs_bunny.hpp -- the bunny network (kifs.wgsl:84-137) mapped onto the machine two ways: four LANES per pixel
// (bunny_sdf_quad: weights in VGPRs, activations through DPP) and four WAVES per 64 rays (bunny_sdf_coop: weights as
// scalar operands, activations through LDS).  The literal one-lane form, bunny_sdf, is in kifs_scene.hpp with the
// other primitives; all three produce every value by the same operation sequence.
#pragma once

#include "kifs_scene.hpp"

namespace kifs {

// ---- the bunny, four lanes per pixel -------------------------------------------------------
// The bunny network (kifs.wgsl:84-137) is 3000 instructions per estimate when one lane does it
// all, and a frame's run time is the longest ray's estimates back to back (a lone wave issues
// one instruction per ~5 cycles whatever it is).  But the network is four independent column
// groups per layer: lane j of a quad computes group j of its pixel -- f0[j], f1[j], f2[j] and
// the j-th partial dot product -- reading the other groups' activations through DPP quad_perm
// operands (no extra instructions, no LDS).  A wave then holds 16 pixels and an estimate is
// ~700 instructions per lane; every value is produced by the same operation sequence as in
// bunny_sdf, so the result is bit-identical.  The four lanes of a quad carry identical ray
// state, which keeps all control flow quad-uniform (DPP never reads an inactive lane).
// W2LDS: layer 2's 64 weights per column group stay in the workgroup's LDS (1 KB for the four groups, staged once by
// bunny_w2_stage) instead of VGPRs: 216 -> 168 registers, THREE waves per SIMD instead of two, for sixteen 128-bit LDS
// reads per estimate, four at a time.  A throughput form: the reads lengthen a lone wave's chain (1 to 12 frames of 1080p
// per launch: -11 to -14 %), the third wave pays from ~18 frames (x20 40.1 -> 44.4 Gpixel/s, x24 40.6 -> 49.0, x32 39.9 ->
// 49.1: profiles/r04/sweep_bunny_w2lds.txt) until the four-waves form takes over (x40 50.0 against 55.0).  Round 3 had tried
// ALL 156 weights in LDS: 39 reads per estimate, LDS-port-bound, slower everywhere.  Same operations on the same values.
template <bool W2LDS>
struct BunnyQuadT {  // the weights of column group j, resident in VGPRs
    float w0[16], w1[4][16], b1[4];
    float w2[W2LDS ? 1 : 4][16];  // (W2LDS: unused; layer 2 is read through `w2_lds`)
    const float* w2_lds;          // W2LDS: this lane's column group of layer 2 in LDS, [4][16]
    float b2[4], wo[4];
};
using BunnyQuad = BunnyQuadT<false>;

// Layer 2 of all four column groups into the workgroup's LDS (256 floats, one per thread of a 256-thread workgroup);
// the caller's next barrier publishes it.
KIFS_DEV void bunny_w2_stage(float* s_w2, int tid) {
    if (tid < 256) s_w2[tid] = (&KIFS_BUNNY_L2[0][0][0])[tid];
}

template <bool W2LDS>
KIFS_DEV void bunny_quad_load(BunnyQuadT<W2LDS>& W, int j, const float* s_w2 = nullptr) {
#pragma unroll
    for (int e = 0; e < 16; ++e) W.w0[e] = KIFS_BUNNY_L0[j][e];
    W.w2_lds = W2LDS ? s_w2 + 64 * j : nullptr;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            W.w1[m][e] = KIFS_BUNNY_L1[j][m][e];
            if constexpr (!W2LDS) W.w2[m][e] = KIFS_BUNNY_L2[j][m][e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        W.b1[e] = KIFS_BUNNY_B1[j][e];
        W.b2[e] = KIFS_BUNNY_B2[j][e];
        W.wo[e] = KIFS_BUNNY_OUT[j][e];
    }
}

template <int M>
KIFS_DEV float quad_lane(float v) {  // v of lane M of the caller's quad
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), M * 0x55, 0xf, 0xf, true));
}
template <int M>
KIFS_DEV V4 quad_lane4(V4 v) {
    return V4{quad_lane<M>(v.x), quad_lane<M>(v.y), quad_lane<M>(v.z), quad_lane<M>(v.w)};
}

#ifdef KIFS_EVAL_COUNT
static __device__ unsigned long long g_eval_counts[8];  // (per translation unit; read in kifs_bunny_kernels.hip)
KIFS_DEV void eval_count(int slot, unsigned long long v) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(true);
    if (__lane_id() == uint32_t(__builtin_ctzll(m))) atomicAdd(&g_eval_counts[slot], v);
}
#endif
// (Tried, r03: a `lanes` mask as in genjulia_sdf, so that a ray that has stopped inside the unit ball does not keep its wave
// evaluating the network -- lone frame 0.376 -> 0.420 ms, 8 per launch 25.7 -> 25.3, 48 per launch 55.8 -> 55.2 Gpixel/s: the
// lane test costs more than the evaluations it saves.)
template <bool W2LDS>
KIFS_DEV float bunny_sdf_quad(const BunnyQuadT<W2LDS>& W, V3 p) {
#ifdef KIFS_EVAL_COUNT
    eval_count(2, 1);
    eval_count(3, __builtin_popcountll(__builtin_amdgcn_ballot_w64(true)) / 4);
    if (!(dot(p, p) > 1.0f)) {
        eval_count(0, 1);
        eval_count(1, __builtin_popcountll(__builtin_amdgcn_ballot_w64(true)) / 4);
    }
#endif
    if (dot(p, p) > 1.0f) return length(p) - 0.8f;
    V4 q{p.x * -1.0f, p.z * 1.0f, p.y * -1.0f, 1.0f};
    const V4 f0 = sin4_flat(mat4_vec(W.w0, q));
    V4 a = mat4_vec(W.w1[0], quad_lane4<0>(f0));
    a = add4(a, mat4_vec(W.w1[1], quad_lane4<1>(f0)));
    a = add4(a, mat4_vec(W.w1[2], quad_lane4<2>(f0)));
    a = add4(a, mat4_vec(W.w1[3], quad_lane4<3>(f0)));
    a = add4(a, ld4(W.b1));
    const V4 f1 = add4(sin4_flat(a), f0);
    if constexpr (W2LDS) {
        // opaque per estimate and per matrix: the sixteen reads stay reads, four at a time (hoisted out of the march they
        // are 64 VGPRs again; issued all at once, 64 temporaries) -- each matrix's address is tied to the sum before it
        const float* w2 = W.w2_lds;
        asm volatile("" : "+v"(w2));
        a = mat4_vec(w2, quad_lane4<0>(f1));
        asm volatile("" : "+v"(w2), "+v"(a.x));
        a = add4(a, mat4_vec(w2 + 16, quad_lane4<1>(f1)));
        asm volatile("" : "+v"(w2), "+v"(a.x));
        a = add4(a, mat4_vec(w2 + 32, quad_lane4<2>(f1)));
        asm volatile("" : "+v"(w2), "+v"(a.x));
        a = add4(a, mat4_vec(w2 + 48, quad_lane4<3>(f1)));
    } else {
        a = mat4_vec(W.w2[0], quad_lane4<0>(f1));
        a = add4(a, mat4_vec(W.w2[1], quad_lane4<1>(f1)));
        a = add4(a, mat4_vec(W.w2[2], quad_lane4<2>(f1)));
        a = add4(a, mat4_vec(W.w2[3], quad_lane4<3>(f1)));
    }
    a = add4(a, ld4(W.b2));
    const V4 sn = sin4_flat(a);
    const V4 f2{sn.x / 1.4f + f1.x, sn.y / 1.4f + f1.y, sn.z / 1.4f + f1.z, sn.w / 1.4f + f1.w};
    const float d = dot(f2, ld4(W.wo));
    float r = quad_lane<0>(d);
    r = r + quad_lane<1>(d);
    r = r + quad_lane<2>(d);
    r = r + quad_lane<3>(d);
    return r - 0.16f;
}

// ---- the bunny, four WAVES per 64 pixels ---------------------------------------------------------
// The quad form above keeps a column group's 156 weights in VGPRs: 216 registers, two waves per SIMD, and a wave
// is one dependent chain -- the vector pipes are 28 % busy (profiles/r03).  Throughput launches turn the mapping
// by ninety degrees: the four waves of a workgroup evaluate the network for the SAME 64 points, wave j computing
// column group j.  The group index is then wave-uniform, so the weights are scalar loads from constant memory and
// scalar operands of the fmas (no VGPRs at all), and the other groups' activations come through LDS: per estimate
// three exchanges of 4 KB / 4 KB / 1 KB with a workgroup barrier each, instead of 156 registers per lane.  The four
// waves carry identical ray state, so their control flow is identical and every barrier is reached by all of them.
// Same operations on the same values in the same order as bunny_sdf: bit-identical.
// (Tried and dropped, r03: the next layer's first two matrices loaded in front of the barrier -- 22 -> 56 SGPR spills,
// 51.8 -> 36.8 Gpixel/s with this sine, 50.1 -> 49.0 with sin_flat; a latency kernel of this form, one workgroup per
// 64 pixels and no queue -- lone 1080p frame 0.50 ms against the quad form's 0.38: gpurun_out sweep kept as
// profiles/r03/sweep_bunny_variants.txt.)
struct BunnyCoop {
    float (*a)[4][64];  // [group][component][lane]: layer-0 activations
    float (*b)[4][64];  // layer-1 activations
    float (*c)[64];     // [group][lane]: the groups' partial dot products
    int j;              // this wave's column group (wave-uniform)
};
KIFS_DEV V4 coop_load(float (*x)[4][64], int g, uint32_t lane) {
    return V4{x[g][0][lane], x[g][1][lane], x[g][2][lane], x[g][3][lane]};
}
KIFS_DEV void coop_store(float (*x)[4][64], int g, uint32_t lane, V4 v) {
    x[g][0][lane] = v.x;
    x[g][1][lane] = v.y;
    x[g][2][lane] = v.z;
    x[g][3][lane] = v.w;
}
// (the sine with its branches here, sin_flat in the quad form -- measured both ways: 48 frames per launch 51.8 against
// 50.1 Gpixel/s here, where the vector pipe is the limit and a wave whose lanes all take one kernel skips the other;
// the quad form's lone frame 0.461 -> 0.377 ms with sin_flat, where the dependent chain is)
KIFS_DEV float bunny_sdf_coop(const BunnyCoop& X, V3 p) {
    const bool far = dot(p, p) > 1.0f;
    const float outside = length(p) - 0.8f;
#ifdef KIFS_EVAL_COUNT
    eval_count(2, 1);
    if (X.j == 0) eval_count(3, __builtin_popcountll(__builtin_amdgcn_ballot_w64(true)));
    if (__builtin_amdgcn_ballot_w64(!far) != 0ull) {
        eval_count(0, 1);
        if (X.j == 0) eval_count(1, __builtin_popcountll(__builtin_amdgcn_ballot_w64(!far)));
    }
#endif
    // (uniform over the workgroup: every wave holds the same points)
    if (__builtin_amdgcn_ballot_w64(!far) == 0ull) return outside;
    const uint32_t lane = __lane_id();
    const int j = X.j;
    V4 q{p.x * -1.0f, p.z * 1.0f, p.y * -1.0f, 1.0f};
    const V4 f0 = sin4(mat4_vec(KIFS_BUNNY_L0[j], q));
    coop_store(X.a, j, lane, f0);
    __syncthreads();
    V4 a = mat4_vec(KIFS_BUNNY_L1[j][0], coop_load(X.a, 0, lane));
    a = add4(a, mat4_vec(KIFS_BUNNY_L1[j][1], coop_load(X.a, 1, lane)));
    a = add4(a, mat4_vec(KIFS_BUNNY_L1[j][2], coop_load(X.a, 2, lane)));
    a = add4(a, mat4_vec(KIFS_BUNNY_L1[j][3], coop_load(X.a, 3, lane)));
    a = add4(a, ld4(KIFS_BUNNY_B1[j]));
    const V4 f1 = add4(sin4(a), f0);
    coop_store(X.b, j, lane, f1);
    __syncthreads();
    a = mat4_vec(KIFS_BUNNY_L2[j][0], coop_load(X.b, 0, lane));
    a = add4(a, mat4_vec(KIFS_BUNNY_L2[j][1], coop_load(X.b, 1, lane)));
    a = add4(a, mat4_vec(KIFS_BUNNY_L2[j][2], coop_load(X.b, 2, lane)));
    a = add4(a, mat4_vec(KIFS_BUNNY_L2[j][3], coop_load(X.b, 3, lane)));
    a = add4(a, ld4(KIFS_BUNNY_B2[j]));
    const V4 sn = sin4(a);
    const V4 f2{sn.x / 1.4f + f1.x, sn.y / 1.4f + f1.y, sn.z / 1.4f + f1.z, sn.w / 1.4f + f1.w};
    X.c[j][lane] = dot(f2, ld4(KIFS_BUNNY_OUT[j]));
    __syncthreads();
    float r = X.c[0][lane];
    r = r + X.c[1][lane];
    r = r + X.c[2][lane];
    r = r + X.c[3][lane];
    return far ? outside : r - 0.16f;
}

// Ray of one pixel, marched by the four lanes of a quad together.
KIFS_DEV V3 raymarch_bunny_quad(const FrameParams& P, V3 dir, bool valid, int quad_lane_id, int& steps) {
    const bool cull = (P.is_heatmap == 0u) && (P.cull_n2 > 0.0f);
    const bool worth = valid && !(cull && ray_never_inside(P, dir));
    if (__builtin_amdgcn_ballot_w64(worth) == 0ull) {  // nothing to march: skip the weight loads
        steps = 0;
        return P.background_color;
    }
    BunnyQuad W;
    bunny_quad_load(W, quad_lane_id);
    return raymarch_with(
        P, dir, valid, steps, [&](V3 q, unsigned long long) { return bunny_sdf_quad(W, q); },
        [&](V3 q) { return normal_fd(P.epsilon, q, [&](V3 u) { return bunny_sdf_quad(W, u); }); });
}


}  // namespace kifs
