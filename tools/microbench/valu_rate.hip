// valu_rate.hip -- sustained VALU issue rate of a full chip (gfx950): wave-instructions per cycle per SIMD
// for the instruction kinds the march loops are made of, at 1 / 2 / 4 / 8 waves per SIMD.  Kernel time from
// HIP events, clock from s_memtime / s_memrealtime inside the kernel (100 MHz reference).
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define LOOPS 2048
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* sink, unsigned long long* clk, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0000001f, c = 1e-7f;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < LOOPS; ++i) {
        if (MODE == 0) {  // 16 independent-ish v_fma_f32 (8 accumulators twice)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (MODE == 1) {  // 16 v_pk_fma_f32 (4 accumulator pairs, four times)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
        } else if (MODE == 2) {  // 16 v_pk_mul_f32
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
        } else if (MODE == 3) {  // 16 v_mul_f32
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (MODE == 5 || MODE == 6) {  // 16 v_fma_f32 with only the low 32 (5) or low 8 (6) lanes enabled
            asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, %10\n"
                         "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         "s_mov_b64 exec, s[20:21]\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(b), "v"(c), "s"(MODE == 5 ? 0xffffffffull : 0xffull) : "s20", "s21");
        } else if (MODE == 7) {  // 16 v_fma_f32 in ONE dependent chain (the Sierpinski fold, a Horner polynomial)
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(b), "v"(c));
        } else if (MODE == 8) {  // 16 instructions in TWO interleaved dependent chains
            asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                         "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                         "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                         "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                         : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
        } else if (MODE == 9) {  // the fold's own mix in one dependent chain: add, mul, min, mul, fma, fma (x2) + 4 fma
            asm volatile("v_add_f32 %1, %0, %0\n v_mul_f32 %1, %1, %2\n v_min_f32 %1, 0, %1\n v_fma_f32 %0, %1, %2, %0\n"
                         "v_add_f32 %1, %0, %0\n v_mul_f32 %1, %1, %2\n v_min_f32 %1, 0, %1\n v_fma_f32 %0, %1, %2, %0\n"
                         "v_add_f32 %1, %0, %0\n v_mul_f32 %1, %1, %2\n v_min_f32 %1, 0, %1\n v_fma_f32 %0, %1, %2, %0\n"
                         "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n"
                         : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
        } else if (MODE == 4) {  // 16 v_cmp + v_cndmask pairs (8 pairs)
            asm volatile("v_cmp_gt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc\n v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %1, %1, %2, vcc\n"
                         "v_cmp_gt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc\n v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %1, %1, %2, vcc\n"
                         "v_cmp_gt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc\n v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %1, %1, %2, vcc\n"
                         "v_cmp_gt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc\n v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %1, %1, %2, vcc\n"
                         : "+v"(a0), "+v"(a1) : "v"(b) : "vcc");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
void run(const char* name) {
    float* sink; unsigned long long* clk;
    hipMalloc(&sink, 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 3, 4, 5, 6, 8}) {  // waves per SIMD: 256 CUs x 4 SIMDs x wps waves, as blocks of 256 threads
        int blocks = 256 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, sink, clk, 1.0f);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, sink, clk, 1.0f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double ghz = double(h[0]) / double(h[1]) * 0.1;           // memtime ticks per 10 ns
        double instr_per_simd = double(LOOPS) * 16.0 * wps;       // wave-instructions issued on one SIMD
        double cycles = ms * 1e-3 * ghz * 1e9;
        printf("%-14s %d waves/SIMD: %.3f ms, in-kernel clock %.2f GHz, %.3f wave-instr/cycle/SIMD (%.2f cycles each), one wave: %.2f ticks per instr\n",
               name, wps, ms, ghz, instr_per_simd / cycles, cycles / instr_per_simd, double(h[0]) / (LOOPS * 16.0));
    }
}

int main() {
    run<0>("v_fma_f32"); run<1>("v_pk_fma_f32"); run<2>("v_pk_mul_f32"); run<3>("v_mul_f32"); run<4>("v_cmp+cndmask");
    run<5>("v_fma lanes<32"); run<6>("v_fma lanes<8");
    run<7>("fma 1 chain"); run<8>("fma 2 chains"); run<9>("fold mix chain");
    return 0;
}
