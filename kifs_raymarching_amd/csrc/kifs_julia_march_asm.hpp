// The asm statement of julia_fast_march (kifs_scene.hpp), included once per variant with
// KIFS_JULIA_DIVSQRT / KIFS_JULIA_DIVSQRT_OUT_OF_LINE, KIFS_FAST_TRIP / KIFS_TRIP_EXIT / KIFS_TRIP_EXIT_BACK / KIFS_JULIA_PROLOGUE /
// KIFS_JULIA_C_OPERANDS / KIFS_ORBIT_LOOP / KIFS_ORBIT_LOOP_OUT_OF_LINE defined by the includer.  See the
// register map and the description there.
    asm volatile(
        "s_mov_b64 s[84:85], exec\n"
        "s_and_b64 exec, exec, %[lanes]\n"
        "s_mov_b32 s88, 2.0\n"
        "s_mov_b32 s89, 4.0\n"
        "s_mov_b32 s90, 2.0\n"
        "s_mov_b32 s91, 1.0\n"
        "s_mov_b32 s92, 0x7fffff\n"
        "s_movk_i32 s93, 0x100\n"                              // class mask: +normal
        "s_movk_i32 s94, 0x260\n"                              // class mask: -0, +0, +inf
        "v_cmp_lt_f32_e64 s[74:75], 0, %[cullv]\n"            // all ones when the culls are enabled
        // ------------------------------------------------------------------ one march step
        // the step loop's head on a 64-byte line: a lone wave refetches after every taken branch
        ".p2align 6\n"
        "10:\n"
        "v_mul_f32_e32 v52, v32, v32\n"                       // dot(p,p): x*x, +y*y, +z*z
        "v_fma_f32 v52, v30, v30, v52\n"
        "v_fma_f32 v52, v31, v31, v52\n"
        "v_cmp_lt_f32 vcc, %[bound], v52\n"               // length(p) > 2 + epsilon ?
        "s_mov_b64 s[78:79], vcc\n"                           // lanes outside the sphere (usually none)
        "s_andn2_b64 exec, exec, vcc\n"                       // lanes inside run the orbit
        "s_cbranch_execz 40f\n"
        KIFS_JULIA_PROLOGUE
        "s_mov_b64 s[86:87], exec\n"
        "s_mov_b32 s96, %[blocks]\n"
        "s_cmp_lg_u32 %[rem], 0\n"
        "s_cbranch_scc1 30f\n"                                // trip count not a multiple of 6 (out of line)
        KIFS_ORBIT_LOOP                                       // blocks of six trips from label 12, falls out at 14
        "14:\n"
        "s_mov_b64 exec, s[86:87]\n"
        // |q|^2 must be a positive normal number for the short log; otherwise hand the step back
        "v_cmp_class_f32_e64 vcc, v44, s93\n"
        "s_xor_b64 vcc, vcc, exec\n"
        "s_cbranch_scc1 19f\n"
        // ---- lg = log(|q|^2)  (log_normal)
        "v_and_or_b32 v56, v44, s92, 0.5\n"                   // mantissa in [0.5, 1)
        "v_lshrrev_b32_e32 v58, 23, v44\n"                    // biased exponent
        "v_cmp_gt_f32_e32 vcc, 0x3f3504f3, v56\n"             // m < sqrt(1/2)
        "v_add_u32_e32 v58, 0xffffff82, v58\n"                // e = biased - 126
        "v_mul_f32_e32 v53, v43, v47\n"                       // dqs = dq * (4|q_last|^2)
        "v_cndmask_b32_e32 v55, 0, v56, vcc\n"                // m or +0.0
        "v_subbrev_co_u32_e32 v58, vcc, 0, v58, vcc\n"        // e -= 1 where m < sqrt(1/2)
        "v_add_f32_e32 v56, v55, v56\n"                       // m + m  or  m
        "v_add_f32_e32 v56, -1.0, v56\n"                      // reduced argument
        "v_cvt_f32_i32_e32 v58, v58\n"                        // fe
        "v_fmamk_f32 v57, v56, 0x3d9021bb, v59\n"             // Horner, 9 coefficients
        "v_fmaak_f32 v57, v57, v56, 0x3def251a\n"
        "v_fmaak_f32 v57, v57, v56, 0xbdfe5d4f\n"
        "v_fmaak_f32 v57, v57, v56, 0x3e11e9bf\n"
        "v_fmaak_f32 v57, v57, v56, 0xbe2aae50\n"
        "v_fmaak_f32 v57, v57, v56, 0x3e4cceac\n"
        "v_fmaak_f32 v57, v57, v56, 0xbe7ffffc\n"
        "v_fmaak_f32 v57, v57, v56, 0x3eaaaaaa\n"
        "v_pk_mul_f32 v[54:55], v[56:57], v[56:57] op_sel_hi:[0,1]\n"  // [z = m*m, p*m]
        "v_mul_f32_e32 v55, v55, v54\n"                       // y = (p*m)*z
        "v_fmac_f32_e32 v55, 0xb95e8083, v58\n"               // y = fma(fe, -2.12194440e-4, y)
        "v_fmac_f32_e32 v55, -0.5, v54\n"                     // y = fma(-0.5, z, y)
        "v_add_f32_e32 v55, v56, v55\n"                       // r = m + y
        "v_fmac_f32_e32 v55, 0x3f318000, v58\n"               // lg = fma(fe, 0.693359375, r)
        KIFS_JULIA_DIVSQRT
        "v_mul_f32_e32 v54, v54, v60\n"                       // d = (0.25 lg) * root
        // ---- lanes outside the bounding sphere: d = length(p) - 2 (julia.wgsl:8-9)
        "40:\n"
        "s_or_b64 exec, exec, s[78:79]\n"                     // all live lanes again
        "s_cmp_lg_u64 s[78:79], 0\n"
        "s_cbranch_scc1 45f\n"                                // out of line; comes back to 41
        "41:\n"
        // ---- hit test, advance
        "v_cmp_gt_f32_e32 vcc, %[eps], v54\n"                 // d < epsilon
        "s_or_b64 %[hit], %[hit], vcc\n"
        "s_andn2_b64 exec, exec, vcc\n"                       // hit lanes freeze at their hit point
        "v_add_f32_e32 v34, v34, v54\n"                       // t += d
        "v_fma_f32 v32, v34, v35, %[ox]\n"                    // p = origin + t * dir
        "v_pk_fma_f32 v[30:31], v[34:35], v[36:37], %[oyz] op_sel_hi:[0,1,1]\n"
        "v_cmpx_gt_f32 vcc, %[maxd], v34\n"                   // t < max_distance; the others stop as misses
        "s_add_u32 %[trips], %[trips], 1\n"
        "s_cbranch_execz 18f\n"
        "s_cmp_lt_i32 %[trips], %[limit]\n"                  // limit = min(max_iterations, end of the round)
        "s_cbranch_scc1 10b\n"
        "s_cmp_lt_i32 %[trips], %[maxit]\n"
        "s_cbranch_scc0 18f\n"
        "s_mov_b64 %[live], exec\n"                          // end of a round: these lanes march on next round
        "s_branch 20f\n"
        "18:\n"                                               // nobody left, or out of iterations
        "s_mov_b64 %[live], 0\n"
        "s_branch 20f\n"
        KIFS_ORBIT_LOOP_OUT_OF_LINE
        // ---- remainder trips (sdf_iters % 6), out of the hot line
        "30:\n"
        "s_mov_b32 s97, %[rem]\n"
        "31:\n"
        KIFS_FAST_TRIP KIFS_TRIP_EXIT_BACK
        "s_sub_u32 s97, s97, 1\n"
        "s_cmp_lg_u32 s97, 0\n"
        "s_cbranch_scc1 31b\n"
        "s_branch 12b\n"
        // ---- sqrt(n2) - 2 for the outside lanes (n2 = v52 is still intact: the orbit lanes
        //      only overwrite it inside their own sqrt, under their own exec)
        "45:\n"
        "s_add_u32 %[nout], %[nout], 1\n"                    // diagnostics: steps with outside lanes
        "s_mov_b64 s[76:77], exec\n"
        "s_mov_b64 exec, s[78:79]\n"
        // With the culls enabled (s[74:75] != 0: a sane scene, fill_params / enqueue_batch) n2 is a finite normal
        // number here -- above bound_n2 >= 4, below (|origin| + max_distance)^2 < 4e30 -- so the correctly rounded
        // square root needs neither the 2^32 pre-scaling (below 2^-96) nor the zero / infinity patch: the same
        // v_sqrt and one-ulp fix-up on the same operand, eight instructions and two wait states shorter.  Every
        // ray spends its first steps and, if it misses, its last ones out here.
        "s_cmp_lg_u64 s[74:75], 0\n"
        "s_cbranch_scc0 48f\n"                                // culls off: the general form, out of line
        "v_sqrt_f32_e32 v61, v52\n"
        "s_nop 0\n"
        "v_add_u32_e32 v62, -1, v61\n"                       // candidates one ulp either side
        "v_add_u32_e32 v63, 1, v61\n"
        "v_fma_f32 v55, -v62, v61, v52\n"
        "v_fma_f32 v56, -v63, v61, v52\n"
        "v_cmp_ge_f32_e64 s[80:81], 0, v55\n"
        "v_cmp_lt_f32_e64 s[82:83], 0, v56\n"
        "s_nop 0\n"
        "v_cndmask_b32_e64 v62, v61, v62, s[80:81]\n"
        "v_cndmask_b32_e64 v60, v62, v63, s[82:83]\n"
        "49:\n"
        "v_add_f32_e32 v54, -2.0, v60\n"                      // d = norm - 2
        // early ray termination: outside the sphere with margin and heading away from it, the
        // ray cannot come back inside, so it can never hit: retire the lane as a miss now
        "v_mul_f32_e32 v55, v32, v35\n"                       // dot(p, dir)
        "v_fma_f32 v55, v30, v36, v55\n"
        "v_fma_f32 v55, v31, v37, v55\n"
        "v_cmp_lt_f32_e32 vcc, %[cull], v52\n"                // n2 > 1.1 R^2 (never when cull = 0 -> see below)
        "v_cmp_lt_f32_e64 s[80:81], 0, v55\n"                 // moving outwards
        "s_and_b64 vcc, vcc, s[80:81]\n"
        "s_and_b64 vcc, vcc, s[74:75]\n"                      // culling enabled?
        "s_andn2_b64 s[76:77], s[76:77], vcc\n"               // drop them from the live lanes
        "s_mov_b64 exec, s[76:77]\n"
        "s_branch 41b\n"
        // ---- the general square root of n2 (any operand), for scenes whose culls are off
        "48:\n"
        "v_mul_f32_e32 v61, 0x4f800000, v52\n"
        "v_cmp_gt_f32_e32 vcc, 0x0f800000, v52\n"
        "s_nop 1\n"
        "v_cndmask_b32_e32 v60, v52, v61, vcc\n"
        "v_sqrt_f32_e32 v61, v60\n"
        "s_nop 0\n"
        "v_add_u32_e32 v62, -1, v61\n"
        "v_add_u32_e32 v63, 1, v61\n"
        "v_fma_f32 v55, -v62, v61, v60\n"
        "v_fma_f32 v56, -v63, v61, v60\n"
        "v_cmp_ge_f32_e64 s[80:81], 0, v55\n"
        "v_cmp_lt_f32_e64 s[82:83], 0, v56\n"
        "s_nop 0\n"
        "v_cndmask_b32_e64 v62, v61, v62, s[80:81]\n"
        "v_cndmask_b32_e64 v61, v62, v63, s[82:83]\n"
        "v_mul_f32_e32 v62, 0x37800000, v61\n"
        "v_cndmask_b32_e32 v61, v61, v62, vcc\n"
        "v_cmp_class_f32_e64 vcc, v60, s94\n"
        "s_nop 1\n"
        "v_cndmask_b32_e32 v60, v61, v60, vcc\n"
        "s_branch 49b\n"
        KIFS_JULIA_DIVSQRT_OUT_OF_LINE
        "19:\n"                                               // hand the current step to the general loop
        "s_or_b64 %[live], exec, s[78:79]\n"
        "20:\n"
        "s_mov_b64 exec, s[84:85]\n"
        : "+{v[30:31]}"(pyz), "+{v[32:33]}"(px1), "+{v[34:35]}"(tdx), [trips] "+s"(trips),
          [hit] "+s"(hit_mask), [live] "=&s"(live_out), [nout] "+s"(outside_steps)
        : "{v[36:37]}"(dyz), "{v[38:39]}"(w0), "{v59}"(c1), [oyz] "s"(oyz), [ox] "s"(P.origin.x),
          [eps] "s"(P.epsilon), [maxd] "s"(P.max_distance), [bound] "s"(P.bound_n2), KIFS_JULIA_C_OPERANDS,
          [maxit] "s"(P.max_iterations), [limit] "s"(limit), [blocks] "s"(P.orbit_blocks),
          [rem] "s"(P.orbit_rem), [lanes] "s"(lanes), [cull] "s"(P.cull_n2), [cullv] "v"(P.cull_n2)
        : "vcc", "scc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50",
          "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v60", "v61", "v62", "v63", "s84",
          "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s96", "s97", "s74", "s75", "s76", "s77", "s78",
          "s79", "s80", "s81", "s82", "s83");
