// step_cost.hip -- lone-wave cost of the non-orbit blocks of the Julia fast march step.
#include <hip/hip_runtime.h>
#include <cstdio>
#define LOGB \
 "v_and_or_b32 v56, v44, s92, 0.5\n v_lshrrev_b32_e32 v58, 23, v44\n v_cmp_gt_f32_e32 vcc, 0x3f3504f3, v56\n" \
 "v_add_u32_e32 v58, 0xffffff82, v58\n v_mul_f32_e32 v53, v43, v47\n v_cndmask_b32_e32 v55, 0, v56, vcc\n" \
 "v_subbrev_co_u32_e32 v58, vcc, 0, v58, vcc\n v_add_f32_e32 v56, v55, v56\n v_add_f32_e32 v56, -1.0, v56\n" \
 "v_cvt_f32_i32_e32 v58, v58\n v_fmamk_f32 v57, v56, 0x3d9021bb, v59\n v_fmaak_f32 v57, v57, v56, 0x3def251a\n" \
 "v_fmaak_f32 v57, v57, v56, 0xbdfe5d4f\n v_fmaak_f32 v57, v57, v56, 0x3e11e9bf\n v_fmaak_f32 v57, v57, v56, 0xbe2aae50\n" \
 "v_fmaak_f32 v57, v57, v56, 0x3e4cceac\n v_fmaak_f32 v57, v57, v56, 0xbe7ffffc\n v_fmaak_f32 v57, v57, v56, 0x3eaaaaaa\n" \
 "v_pk_mul_f32 v[54:55], v[56:57], v[56:57] op_sel_hi:[0,1]\n v_mul_f32_e32 v55, v55, v54\n v_fmac_f32_e32 v55, 0xb95e8083, v58\n" \
 "v_fmac_f32_e32 v55, -0.5, v54\n v_add_f32_e32 v55, v56, v55\n v_fmac_f32_e32 v55, 0x3f318000, v58\n"
#define DIVB \
 "v_div_scale_f32 v60, s[82:83], v53, v53, v44\n v_div_scale_f32 v61, vcc, v44, v53, v44\n v_rcp_f32_e32 v62, v60\n" \
 "v_mul_f32_e32 v54, 0x3e800000, v55\n v_fma_f32 v63, -v60, v62, 1.0\n v_fmac_f32_e32 v62, v63, v62\n v_mul_f32_e32 v63, v61, v62\n" \
 "v_fma_f32 v55, -v60, v63, v61\n v_fmac_f32_e32 v63, v55, v62\n v_fma_f32 v60, -v60, v63, v61\n v_div_fmas_f32 v60, v60, v62, v63\n" \
 "v_div_fixup_f32 v60, v60, v53, v44\n"
#define SQRTB \
 "v_mul_f32_e32 v61, 0x4f800000, v60\n v_cmp_gt_f32_e32 vcc, 0x0f800000, v60\n s_nop 1\n v_cndmask_b32_e32 v60, v60, v61, vcc\n" \
 "v_sqrt_f32_e32 v61, v60\n s_nop 0\n v_add_u32_e32 v62, -1, v61\n v_add_u32_e32 v63, 1, v61\n v_fma_f32 v52, -v62, v61, v60\n" \
 "v_fma_f32 v56, -v63, v61, v60\n v_cmp_ge_f32_e64 s[80:81], 0, v52\n v_cmp_lt_f32_e64 s[82:83], 0, v56\n s_nop 0\n" \
 "v_cndmask_b32_e64 v62, v61, v62, s[80:81]\n v_cndmask_b32_e64 v61, v62, v63, s[82:83]\n v_mul_f32_e32 v62, 0x37800000, v61\n" \
 "v_cndmask_b32_e32 v61, v61, v62, vcc\n v_cmp_class_f32_e64 vcc, v60, s94\n s_nop 1\n v_cndmask_b32_e32 v60, v61, v60, vcc\n"
#define BOUNDB \
 "v_mul_f32_e32 v52, v32, v32\n v_fma_f32 v52, v30, v30, v52\n v_fma_f32 v52, v31, v31, v52\n v_cmp_lt_f32 vcc, s95, v52\n s_cbranch_vccnz 9f\n"
#define CLASSB \
 "v_cmp_class_f32_e64 vcc, v44, s93\n s_xor_b64 vcc, vcc, exec\n s_cbranch_scc1 9f\n"
#define ADVB \
 "v_mul_f32_e32 v54, v54, v60\n v_cmp_gt_f32_e32 vcc, s90, v54\n s_or_b64 s[86:87], s[86:87], vcc\n s_andn2_b64 exec, exec, vcc\n" \
 "v_add_f32_e32 v34, v34, v54\n v_fma_f32 v32, v34, v35, s91\n v_pk_fma_f32 v[30:31], v[34:35], v[36:37], s[88:89] op_sel_hi:[0,1,1]\n" \
 "v_cmp_gt_f32_e32 vcc, s96, v34\n s_and_b64 exec, exec, vcc\n s_add_u32 s97, s97, 1\n s_cbranch_execz 9f\n"
#define SETUP \
 "v_pk_mul_f32 v[46:47], v[32:33], s[88:89]\n v_mov_b64 v[40:41], v[30:31]\n v_mov_b64 v[42:43], v[38:39]\n" \
 "v_pk_mul_f32 v[48:49], v[40:41], v[40:41]\n v_pk_add_f32 v[50:51], v[48:49], v[48:49] op_sel:[0,1] op_sel_hi:[0,1]\n" \
 "v_pk_fma_f32 v[48:49], v[42:43], v[42:43], v[50:51] op_sel_hi:[0,0,1]\n" \
 "v_pk_fma_f32 v[44:45], v[32:33], v[32:33], v[48:49] op_sel_hi:[0,0,1] neg_hi:[0,0,1]\n v_pk_add_f32 v[44:45], v[44:45], s[88:89]\n" \
 "s_mov_b64 s[84:85], exec\n s_mov_b32 s98, s99\n s_cmp_lg_u32 s99, 0\n s_cbranch_scc1 9f\n s_cmp_eq_u32 s98, 7\n s_cbranch_scc1 9f\n"

#define KERNEL(NAME, BODY, REPS) \
__global__ void NAME(unsigned long long* out, float seed, int n) { \
    unsigned long long t0, t1; \
    asm volatile( \
      "v_mov_b32 v44, 0x40490fdb\n v_mov_b32 v43, 0x3f800000\n v_mov_b32 v47, 0x42c80000\n v_mov_b32 v59, 0xbdebd1b8\n" \
      "v_mov_b32 v30, 0x3e99999a\n v_mov_b32 v31, 0x3e4ccccd\n v_mov_b32 v32, 0x3dcccccd\n v_mov_b32 v33, 1.0\n v_mov_b32 v34, 1.0\n v_mov_b32 v35, 0.5\n" \
      "v_mov_b32 v36, 0.5\n v_mov_b32 v37, 0.5\n v_mov_b32 v38, 0x3dcccccd\n v_mov_b32 v39, 1.0\n v_mov_b32 v53, 0x42c80000\n v_mov_b32 v55, 1.0\n v_mov_b32 v54, 1.0\n v_mov_b32 v60, 2.0\n" \
      "s_mov_b32 s92, 0x7fffff\n s_movk_i32 s93, 0x100\n s_movk_i32 s94, 0x260\n s_mov_b32 s95, 0x501502f9\n s_mov_b32 s96, 0x501502f9\n s_mov_b32 s97, 0\n" \
      "s_mov_b32 s90, 0x38d1b717\n s_mov_b32 s88, 2.0\n s_mov_b32 s89, 1.0\n s_mov_b32 s91, 0\n s_mov_b64 s[86:87], 0\n s_mov_b32 s99, 0\n" \
      "s_memtime %0\n s_waitcnt lgkmcnt(0)\n" \
      "1:\n" BODY \
      "s_sub_u32 %2, %2, 1\n s_cmp_lg_u32 %2, 0\n s_cbranch_scc1 1b\n 9:\n" \
      "s_memtime %1\n s_waitcnt lgkmcnt(0)\n" \
      : "=&s"(t0), "=&s"(t1), "+s"(n) : \
      : "vcc","scc","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
        "s90","s80","s81","s82","s83","s84","s85","s86","s87","s88","s89","s91","s92","s93","s94","s95","s96","s97","s98","s99"); \
    if (threadIdx.x == 0) { out[0] = t1 - t0; } \
}
KERNEL(k_empty, "", 1)
KERNEL(k_log, LOGB, 1)
KERNEL(k_div, DIVB, 1)
KERNEL(k_sqrt, SQRTB, 1)
KERNEL(k_bound, BOUNDB, 1)
KERNEL(k_class, CLASSB, 1)
KERNEL(k_adv, ADVB, 1)
KERNEL(k_setup, SETUP, 1)
KERNEL(k_all, BOUNDB SETUP CLASSB LOGB DIVB SQRTB ADVB, 1)

template <typename K> void run(const char* name, K kern, int instrs) {
    unsigned long long* d; hipMalloc(&d, 64);
    const int n = 4000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d, 1.0f, n);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-10s %7.1f ticks per rep (incl. ~loop overhead), %d instrs\n", name, double(h) / n, instrs);
    hipFree(d);
}
int main() {
    run("empty", k_empty, 0); run("log", k_log, 24); run("div", k_div, 12); run("sqrt", k_sqrt, 20);
    run("bound", k_bound, 5); run("class", k_class, 3); run("advance", k_adv, 11); run("setup", k_setup, 14);
    run("all", k_all, 89);
    return 0;
}
