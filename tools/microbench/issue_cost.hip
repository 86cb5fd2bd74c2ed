// issue_cost.hip -- lone-wave and full-chip issue cost of the VALU instructions the march
// loop is made of (gfx950).  Prints cycles per instruction from s_memtime around unrolled
// asm blocks.  Build: hipcc --offload-arch=gfx950 -O2 -o issue_cost issue_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
#define LOOPS 256

template <int MODE>
__global__ void k(unsigned long long* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
          a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0000001f, c = 1e-7f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int i = 0; i < LOOPS; ++i) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (MODE == 0) {  // 8 independent v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                             "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                             "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(b), "v"(c));
            } else if (MODE == 1) {  // 8 dependent v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(a0) : "v"(b), "v"(c));
            } else if (MODE == 2) {  // 4 independent v_pk_fma_f32 (= 8 fma)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n"
                             "v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
            } else if (MODE == 3) {  // 4 dependent v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                             "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(p0) : "v"(pb), "v"(pc));
            } else if (MODE == 4) {  // 2 dependent chains interleaved (8 fma)
                asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n"
                             "v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                             "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                             : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
            } else if (MODE == 5) {  // 8 independent v_mul_f32 (e32)
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(b));
            } else if (MODE == 6) {  // 8 dependent v_sqrt_f32
                asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %0, %0\n v_sqrt_f32 %0, %0\n v_sqrt_f32 %0, %0\n"
                             "v_sqrt_f32 %0, %0\n v_sqrt_f32 %0, %0\n v_sqrt_f32 %0, %0\n v_sqrt_f32 %0, %0\n"
                             : "+v"(a0));
            } else if (MODE == 7) {  // 8 independent v_rcp_f32
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (MODE == 8) {  // 8 scalar adds (dependent)
                int s = i;
                asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                             "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                             : "+s"(s));
                a0 += (float)s * 0.0f;
            } else if (MODE == 9) {  // v_cmp + v_cndmask pairs (4 pairs = 8 instr), dependent through vcc
                asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n"
                             "v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n"
                             "v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n"
                             "v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n"
                             : "+v"(a0) : "v"(b) : "vcc");
            } else if (MODE == 10) {  // 4 chains x2 interleaved mul+fma mix typical of the Julia step
                asm volatile("v_mul_f32 %0, %0, %4\n v_fma_f32 %1, %1, %4, %5\n v_mul_f32 %2, %2, %4\n v_fma_f32 %3, %3, %4, %5\n"
                             "v_fma_f32 %0, %0, %4, %5\n v_mul_f32 %1, %1, %4\n v_fma_f32 %2, %2, %4, %5\n v_mul_f32 %3, %3, %4\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
            }
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
    if (s == 123.456f) out[blockIdx.x * 2 + 1] = 1;
}

template <int MODE>
void run(const char* name, int per_block_instr) {
    unsigned long long* d;
    hipMalloc(&d, 8192 * 2 * sizeof(unsigned long long));
    for (int cfg = 0; cfg < 4; ++cfg) {
        // cfg0: 1 wave on the chip; cfg1: 1 wave per SIMD (256 CU x 4); cfg2: 2 waves/SIMD; cfg3: 8 waves/SIMD
        int blocks = cfg == 0 ? 1 : 256, threads = cfg == 0 ? 64 : cfg == 1 ? 256 : cfg == 2 ? 512 : 1024;
        if (cfg == 3) blocks = 512;
        hipMemset(d, 0, 8192 * 2 * sizeof(unsigned long long));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 1.0f);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 2);
        hipMemcpy(h.data(), d, blocks * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sum = 0;
        for (int b = 0; b < blocks; ++b) sum += h[2 * b];
        double cyc = sum / blocks / (double(LOOPS) * (REP / 8) * per_block_instr);
        printf("%-28s cfg%d waves/SIMD=%s : %.2f memtime-ticks per instr\n", name, cfg,
               cfg == 0 ? "1 (lone)" : cfg == 1 ? "1" : cfg == 2 ? "2" : "8", cyc);
    }
    hipFree(d);
}

int main() {
    run<0>("v_fma_f32 x8 independent", 8);
    run<1>("v_fma_f32 x8 dependent", 8);
    run<2>("v_pk_fma_f32 x4 independent", 4);
    run<3>("v_pk_fma_f32 x4 dependent", 4);
    run<4>("v_fma_f32 2 chains interl.", 8);
    run<5>("v_mul_f32 x8 independent", 8);
    run<6>("v_sqrt_f32 x8 dependent", 8);
    run<7>("v_rcp_f32 x8 independent", 8);
    run<8>("s_add_u32 x8 dependent", 8);
    run<9>("v_cmp+v_cndmask x4 pairs", 8);
    run<10>("mul/fma 4 chains mixed", 8);
    return 0;
}
