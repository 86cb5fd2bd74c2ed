// orbit_cost.hip -- cycles per trip of the Julia orbit loop body variants, lone wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float F2 __attribute__((ext_vector_type(2)));

#define HEAD \
    "v_pk_fma_f32 v[40:41], v[46:47], v[40:41], %[cyz] op_sel_hi:[0,1,1]\n" \
    "v_pk_fma_f32 v[42:43], v[46:47], v[42:43], %[cw0]\n" \
    "v_pk_mul_f32 v[46:47], v[44:45], %[k24] op_sel:[1,0] op_sel_hi:[0,1]\n"
#define TAIL \
    "v_pk_mul_f32 v[48:49], v[40:41], v[40:41]\n" \
    "v_pk_add_f32 v[50:51], v[48:49], v[48:49] op_sel:[0,1] op_sel_hi:[0,1]\n" \
    "v_pk_fma_f32 v[48:49], v[42:43], v[42:43], v[50:51] op_sel_hi:[0,0,1]\n" \
    "v_pk_fma_f32 v[44:45], v[44:45], v[44:45], v[48:49] op_sel:[1,1,0] op_sel_hi:[1,1,1] neg_hi:[0,0,1]\n" \
    "v_pk_add_f32 v[44:45], v[44:45], %[c0x]\n"

template <int MODE>
__global__ void k(unsigned long long* out, float seed, int n, float maxd) {
    F2 yz{seed * 0.3f, seed * 0.2f}, wd{0.1f, 1.0f}, q{0.0f, seed * 0.1f}, t{seed * 0.2f, 1.0f}, ta, tb;
    const F2 cyz{0.6f, 0.2f}, cw0{0.2f, 0.0f}, c0x{0.0f, -0.2f}, k24{2.0f, 4.0f};
    unsigned long long save;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == 0) {  // shipped loop: cmp + exec update every trip, unrolled x2
        asm volatile(TAIL "s_mov_b64 %[save], exec\n"
                     "1:\n" HEAD TAIL "v_cmp_lt_f32 vcc, %[maxd], v44\n s_andn2_b64 exec, exec, vcc\n"
                     HEAD TAIL "v_cmp_lt_f32 vcc, %[maxd], v44\n s_andn2_b64 exec, exec, vcc\n"
                     "s_cbranch_execz 2f\n s_sub_u32 %[n], %[n], 1\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n"
                     "2:\n s_mov_b64 exec, %[save]\n"
                     : "+{v[40:41]}"(yz), "+{v[42:43]}"(wd), "+{v[44:45]}"(q), "+{v[46:47]}"(t),
                       "=&{v[48:49]}"(ta), "=&{v[50:51]}"(tb), [save] "=&s"(save), [n] "+s"(n)
                     : [cyz] "s"(cyz), [cw0] "s"(cw0), [c0x] "s"(c0x), [k24] "s"(k24), [maxd] "s"(maxd)
                     : "vcc", "scc");
    } else if (MODE == 1) {  // no compare, no exec update (pure arithmetic)
        asm volatile(TAIL "1:\n" HEAD TAIL HEAD TAIL
                     "s_sub_u32 %[n], %[n], 1\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n"
                     : "+{v[40:41]}"(yz), "+{v[42:43]}"(wd), "+{v[44:45]}"(q), "+{v[46:47]}"(t),
                       "=&{v[48:49]}"(ta), "=&{v[50:51]}"(tb), [n] "+s"(n)
                     : [cyz] "s"(cyz), [cw0] "s"(cw0), [c0x] "s"(c0x), [k24] "s"(k24), [maxd] "s"(maxd)
                     : "vcc", "scc");
    } else if (MODE == 2) {  // compare every trip, exec never touched (mask accumulated in SGPRs)
        unsigned long long acc = 0;
        asm volatile(TAIL "1:\n" HEAD TAIL "v_cmp_lt_f32 vcc, %[maxd], v44\n s_or_b64 %[acc], %[acc], vcc\n"
                     HEAD TAIL "v_cmp_lt_f32 vcc, %[maxd], v44\n s_or_b64 %[acc], %[acc], vcc\n"
                     "s_sub_u32 %[n], %[n], 1\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n"
                     : "+{v[40:41]}"(yz), "+{v[42:43]}"(wd), "+{v[44:45]}"(q), "+{v[46:47]}"(t),
                       "=&{v[48:49]}"(ta), "=&{v[50:51]}"(tb), [n] "+s"(n), [acc] "+s"(acc)
                     : [cyz] "s"(cyz), [cw0] "s"(cw0), [c0x] "s"(c0x), [k24] "s"(k24), [maxd] "s"(maxd)
                     : "vcc", "scc");
        if (acc == 12345) out[1] = 1;
    } else if (MODE == 3) {  // exec update delayed by one trip (uses the previous trip's compare)
        asm volatile(TAIL "s_mov_b64 %[save], exec\n s_mov_b64 vcc, 0\n"
                     "1:\n" HEAD "s_andn2_b64 exec, exec, vcc\n" TAIL "v_cmp_lt_f32 vcc, %[maxd], v44\n"
                     HEAD "s_andn2_b64 exec, exec, vcc\n" TAIL "v_cmp_lt_f32 vcc, %[maxd], v44\n"
                     "s_sub_u32 %[n], %[n], 1\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n"
                     "s_mov_b64 exec, %[save]\n"
                     : "+{v[40:41]}"(yz), "+{v[42:43]}"(wd), "+{v[44:45]}"(q), "+{v[46:47]}"(t),
                       "=&{v[48:49]}"(ta), "=&{v[50:51]}"(tb), [save] "=&s"(save), [n] "+s"(n)
                     : [cyz] "s"(cyz), [cw0] "s"(cw0), [c0x] "s"(c0x), [k24] "s"(k24), [maxd] "s"(maxd)
                     : "vcc", "scc");
    } else if (MODE == 4) {  // fully unrolled 12 trips (no loop branch), cmp + exec each trip
#define TRIP HEAD TAIL "v_cmp_lt_f32 vcc, %[maxd], v44\n s_andn2_b64 exec, exec, vcc\n"
        asm volatile(TAIL "s_mov_b64 %[save], exec\n"
                     "1:\n" TRIP TRIP TRIP TRIP TRIP TRIP TRIP TRIP TRIP TRIP TRIP TRIP
                     "s_sub_u32 %[n], %[n], 6\n s_cmp_gt_i32 %[n], 0\n s_cbranch_scc1 1b\n"
                     "s_mov_b64 exec, %[save]\n"
                     : "+{v[40:41]}"(yz), "+{v[42:43]}"(wd), "+{v[44:45]}"(q), "+{v[46:47]}"(t),
                       "=&{v[48:49]}"(ta), "=&{v[50:51]}"(tb), [save] "=&s"(save), [n] "+s"(n)
                     : [cyz] "s"(cyz), [cw0] "s"(cw0), [c0x] "s"(c0x), [k24] "s"(k24), [maxd] "s"(maxd)
                     : "vcc", "scc");
    }
    asm volatile("s_nop 0" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (yz.x + wd.x + q.x + t.x == 123.456f) out[1] = 2;
}

template <int MODE> void run(const char* name) {
    unsigned long long* d; hipMalloc(&d, 64);
    const int pairs = 6000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, d, 1.0f, pairs, __builtin_inff());
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-44s %.1f ticks per trip\n", name, double(h[0]) / (2.0 * pairs));
    hipFree(d);
}
int main() {
    run<0>("shipped: cmp + exec update per trip (x2)");
    run<1>("arithmetic only (8 pk ops)");
    run<2>("cmp + s_or into SGPR mask, exec untouched");
    run<3>("exec update one trip late");
    run<4>("12 trips straight-line, cmp + exec per trip");
    return 0;
}
