// Microbenchmark: how many scalar (SALU) and vector (VALU) instructions a CU of gfx950 issues per cycle, alone and mixed.
// Build: hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o issue_rate     (scripts/ubench/Makefile)
// Every workgroup is 256 threads (one wavefront per SIMD); `wgs_per_cu` of them per CU give that many wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 32
// MODE 0: 8 independent s_add_u32          1: 8 independent v_add_f32        2: 8 v_add + 8 s_add interleaved
// MODE 3: 8 v_add + 4 s_add                4: 8 x (v_cmp_lt + s_and_b64)     5: 8 v_add + 2 s_add
// MODE 6: 8 s_and_b64 (64-bit scalar ops)  7: 8 v_add + 8 s_nop 0           8: 4 x (v_cmp + v_cndmask)
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *clk, int iters, float a) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4, s4 = 5, s5 = 6, s6 = 7, s7 = 8;
    unsigned long long m0 = 1, m1 = 2, m2 = 3, m3 = 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#define VADD8 "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
            if (MODE == 0) {
                asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                             "s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : : "scc");
            } else if (MODE == 1) {
                asm volatile(VADD8 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
            } else if (MODE == 2) {
                asm volatile("v_add_f32 %0, %0, %16\n s_add_u32 %8, %8, 1\n v_add_f32 %1, %1, %16\n s_add_u32 %9, %9, 1\n"
                             "v_add_f32 %2, %2, %16\n s_add_u32 %10, %10, 1\n v_add_f32 %3, %3, %16\n s_add_u32 %11, %11, 1\n"
                             "v_add_f32 %4, %4, %16\n s_add_u32 %12, %12, 1\n v_add_f32 %5, %5, %16\n s_add_u32 %13, %13, 1\n"
                             "v_add_f32 %6, %6, %16\n s_add_u32 %14, %14, 1\n v_add_f32 %7, %7, %16\n s_add_u32 %15, %15, 1\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7),
                               "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "v"(a) : "scc");
            } else if (MODE == 3) {
                asm volatile("v_add_f32 %0, %0, %12\n v_add_f32 %1, %1, %12\n s_add_u32 %8, %8, 1\n v_add_f32 %2, %2, %12\n v_add_f32 %3, %3, %12\n s_add_u32 %9, %9, 1\n"
                             "v_add_f32 %4, %4, %12\n v_add_f32 %5, %5, %12\n s_add_u32 %10, %10, 1\n v_add_f32 %6, %6, %12\n v_add_f32 %7, %7, %12\n s_add_u32 %11, %11, 1\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7),
                               "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a) : "scc");
            } else if (MODE == 4) {
                asm volatile("v_cmp_lt_f32 vcc, %4, %5\n s_and_b64 %0, %0, vcc\n v_cmp_lt_f32 vcc, %5, %6\n s_and_b64 %1, %1, vcc\n"
                             "v_cmp_lt_f32 vcc, %6, %7\n s_and_b64 %2, %2, vcc\n v_cmp_lt_f32 vcc, %7, %4\n s_and_b64 %3, %3, vcc\n"
                             "v_cmp_lt_f32 vcc, %4, %6\n s_and_b64 %0, %0, vcc\n v_cmp_lt_f32 vcc, %5, %7\n s_and_b64 %1, %1, vcc\n"
                             "v_cmp_lt_f32 vcc, %6, %4\n s_and_b64 %2, %2, vcc\n v_cmp_lt_f32 vcc, %7, %5\n s_and_b64 %3, %3, vcc\n"
                             : "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "vcc", "scc");
            } else if (MODE == 5) {
                asm volatile("v_add_f32 %0, %0, %10\n v_add_f32 %1, %1, %10\n v_add_f32 %2, %2, %10\n v_add_f32 %3, %3, %10\n s_add_u32 %8, %8, 1\n"
                             "v_add_f32 %4, %4, %10\n v_add_f32 %5, %5, %10\n v_add_f32 %6, %6, %10\n v_add_f32 %7, %7, %10\n s_add_u32 %9, %9, 1\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+s"(s0), "+s"(s1) : "v"(a) : "scc");
            } else if (MODE == 6) {
                asm volatile("s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %2\n s_and_b64 %2, %2, %3\n s_or_b64 %3, %3, %0\n"
                             "s_andn2_b64 %0, %0, %2\n s_or_b64 %1, %1, %3\n s_and_b64 %2, %2, %0\n s_or_b64 %3, %3, %1\n"
                             : "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : : "scc");
            } else if (MODE == 7) {
                asm volatile("v_add_f32 %0, %0, %8\n s_nop 0\n v_add_f32 %1, %1, %8\n s_nop 0\n v_add_f32 %2, %2, %8\n s_nop 0\n v_add_f32 %3, %3, %8\n s_nop 0\n"
                             "v_add_f32 %4, %4, %8\n s_nop 0\n v_add_f32 %5, %5, %8\n s_nop 0\n v_add_f32 %6, %6, %8\n s_nop 0\n v_add_f32 %7, %7, %8\n s_nop 0\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
            } else if (MODE == 9) {      // 8 v_add_f32 with a 32-bit LITERAL operand
                asm volatile("v_add_f32 %0, 0x3089705f, %0\n v_add_f32 %1, 0x3089705f, %1\n v_add_f32 %2, 0x3089705f, %2\n v_add_f32 %3, 0x3089705f, %3\n"
                             "v_add_f32 %4, 0x3089705f, %4\n v_add_f32 %5, 0x3089705f, %5\n v_add_f32 %6, 0x3089705f, %6\n v_add_f32 %7, 0x3089705f, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (MODE == 10) {     // 8 v_add_f32 with an SGPR operand
                asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                             "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a));
            } else if (MODE == 11) {     // 8 v_add_f32 with an inline constant
                asm volatile("v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n v_add_f32 %2, 1.0, %2\n v_add_f32 %3, 1.0, %3\n"
                             "v_add_f32 %4, 1.0, %4\n v_add_f32 %5, 1.0, %5\n v_add_f32 %6, 1.0, %6\n v_add_f32 %7, 1.0, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (MODE == 12) {     // 8 v_min_f32 / v_max_f32 (VOP2) on VGPRs
                asm volatile("v_min_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_max_f32 %3, %3, %4\n"
                             "v_min_f32 %4, %4, %5\n v_max_f32 %5, %5, %6\n v_min_f32 %6, %6, %7\n v_max_f32 %7, %7, %0\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (MODE == 13) {     // 8 VOP3: v_fma_f32 / v_min3_f32 on VGPRs
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_min3_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %4\n v_min3_f32 %3, %3, %4, %5\n"
                             "v_fma_f32 %4, %4, %5, %6\n v_min3_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %0\n v_min3_f32 %7, %7, %0, %1\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (MODE == 14) {     // 8 v_add_f32 with a DPP modifier (quad_perm)
                asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (MODE == 15) {     // 8 v_cndmask_b32 with the condition in an SGPR pair (not vcc)
                asm volatile("v_cndmask_b32 %0, %0, %1, %8\n v_cndmask_b32 %1, %1, %2, %8\n v_cndmask_b32 %2, %2, %3, %8\n v_cndmask_b32 %3, %3, %4, %8\n"
                             "v_cndmask_b32 %4, %4, %5, %8\n v_cndmask_b32 %5, %5, %6, %8\n v_cndmask_b32 %6, %6, %7, %8\n v_cndmask_b32 %7, %7, %0, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(m0));
            } else if (MODE == 16) {     // 8 v_cmp_lt_f32 into DIFFERENT scalar pairs (no consumer)
                asm volatile("v_cmp_lt_f32 %0, %4, %5\n v_cmp_lt_f32 %1, %5, %6\n v_cmp_lt_f32 %2, %6, %7\n v_cmp_lt_f32 %3, %7, %4\n"
                             "v_cmp_lt_f32 %0, %4, %6\n v_cmp_lt_f32 %1, %5, %7\n v_cmp_lt_f32 %2, %6, %4\n v_cmp_lt_f32 %3, %7, %5\n"
                             : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
            } else if (MODE == 17) {     // 8 v_min3_f32 only
                asm volatile("v_min3_f32 %0, %0, %1, %1\n v_min3_f32 %1, %1, %2, %2\n v_min3_f32 %2, %2, %3, %3\n v_min3_f32 %3, %3, %4, %4\n"
                             "v_min3_f32 %4, %4, %5, %5\n v_min3_f32 %5, %5, %6, %6\n v_min3_f32 %6, %6, %7, %7\n v_min3_f32 %7, %7, %0, %0\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (MODE == 18) {     // 8 v_mul_f32 (VOP2)
                asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %4\n"
                             "v_mul_f32 %4, %4, %5\n v_mul_f32 %5, %5, %6\n v_mul_f32 %6, %6, %7\n v_mul_f32 %7, %7, %0\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (MODE == 19) {     // 8 v_min_f32 with INDEPENDENT destinations (no chain at all)
                asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
                             "v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
            } else if (MODE == 20) {     // 8 v_add_f32 chained like MODE 12 (x0 = x0 + x1, x1 = x1 + x2, ...)
                asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %4\n"
                             "v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n v_add_f32 %6, %6, %7\n v_add_f32 %7, %7, %0\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (MODE == 21) {     // 4 x (v_cmp_lt_f32 vcc + v_cndmask vcc), independent pairs on 8 registers
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n"
                             "v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");
            } else if (MODE == 22) {     // 8 v_max3_f32
                asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n"
                             "v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (MODE == 23) {     // 8 integer logic ops (v_or_b32 / v_and_b32)
                asm volatile("v_or_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_or_b32 %2, %2, %3\n v_and_b32 %3, %3, %4\n"
                             "v_or_b32 %4, %4, %5\n v_and_b32 %5, %5, %6\n v_or_b32 %6, %6, %7\n v_and_b32 %7, %7, %0\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (MODE == 24) {     // 8 integer adds / shifts
                asm volatile("v_add_u32 %0, %0, %1\n v_lshlrev_b32 %1, 1, %2\n v_add_u32 %2, %2, %3\n v_lshlrev_b32 %3, 1, %4\n"
                             "v_add_u32 %4, %4, %5\n v_lshlrev_b32 %5, 1, %6\n v_add_u32 %6, %6, %7\n v_lshlrev_b32 %7, 1, %0\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (MODE == 25) {     // 8 three-operand integer ops (v_or3_b32 / v_and_or_b32 / v_add3_u32)
                asm volatile("v_or3_b32 %0, %0, %1, %2\n v_and_or_b32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %4\n v_or3_b32 %3, %3, %4, %5\n"
                             "v_and_or_b32 %4, %4, %5, %6\n v_add3_u32 %5, %5, %6, %7\n v_or3_b32 %6, %6, %7, %0\n v_and_or_b32 %7, %7, %0, %1\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (MODE == 26) {     // 8 v_max_i32 / v_min_u32 (integer min/max)
                asm volatile("v_max_i32 %0, %0, %1\n v_min_u32 %1, %1, %2\n v_max_i32 %2, %2, %3\n v_min_u32 %3, %3, %4\n"
                             "v_max_i32 %4, %4, %5\n v_min_u32 %5, %5, %6\n v_max_i32 %6, %6, %7\n v_min_u32 %7, %7, %0\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (MODE == 27) {     // 4 v_sqrt_f32 + 4 v_rcp_f32 (transcendental)
                asm volatile("v_sqrt_f32 %0, %0\n v_rcp_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_sqrt_f32 %4, %4\n v_rcp_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (MODE == 28) {     // 8 v_sub_f32 with |abs| / -neg modifiers (VOP3 encoded)
                asm volatile("v_sub_f32 %0, |%0|, %1\n v_sub_f32 %1, -%1, %2\n v_sub_f32 %2, |%2|, %3\n v_sub_f32 %3, -%3, %4\n"
                             "v_sub_f32 %4, |%4|, %5\n v_sub_f32 %5, -%5, %6\n v_sub_f32 %6, |%6|, %7\n v_sub_f32 %7, -%7, %0\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (MODE == 8) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc\n"
                             "v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %5, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %5, vcc\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(x4), "v"(x5) : "vcc");
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7) + (float)(m0 + m1 + m2 + m3);
}

template <int MODE>
void run(const char *name, int wgs_per_cu, int valu_per_rep, int salu_per_rep) {
    float *d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    unsigned long long *clk; hipMalloc(&clk, 16);
    const int iters = 4000, blocks = 256 * wgs_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(d, clk, 10, 1.0001f);
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(d, clk, iters, 1.0001f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);        // s_memtime ticks per ns (s_memrealtime: 100 MHz)
    const double cycles = (double)h[0];                              // of one wavefront, start to end
    const double per_cu = (double)iters * REP * 4.0 * wgs_per_cu;    // wavefront-instruction groups per CU
    (void)cycles;
    const double kernel_cycles = ms * 1e-3 * 2.4e9;                  // at the nominal clock, from the elapsed time
    printf("%-30s %d waves/SIMD: %7.3f ms, clock %.2f GHz: per CU and cycle (2.4 GHz) %.2f VALU + %.2f SALU\n", name, wgs_per_cu, ms, ghz,
           per_cu * valu_per_rep / kernel_cycles, per_cu * salu_per_rep / kernel_cycles);
    hipFree(d); hipFree(clk);
}

int main() {
    for (int w : {1, 8}) {
        run<0>("s_add_u32 x8", w, 0, 8);
        run<6>("s_and/or_b64 x8", w, 0, 8);
        run<1>("v_add_f32 x8", w, 8, 0);
        run<2>("v_add x8 + s_add x8", w, 8, 8);
        run<3>("v_add x8 + s_add x4", w, 8, 4);
        run<5>("v_add x8 + s_add x2", w, 8, 2);
        run<4>("(v_cmp + s_and_b64) x8", w, 8, 8);
        run<7>("v_add x8 + s_nop x8", w, 8, 8);
        run<8>("(v_cmp + v_cndmask) x4", w, 8, 0);
        run<11>("v_add_f32 inline constant x8", w, 8, 0);
        run<9>("v_add_f32 32-bit literal x8", w, 8, 0);
        run<10>("v_add_f32 SGPR operand x8", w, 8, 0);
        run<12>("v_min/max_f32 x8", w, 8, 0);
        run<13>("v_fma / v_min3 (VOP3) x8", w, 8, 0);
        run<14>("v_add_f32_dpp x8", w, 8, 0);
        run<15>("v_cndmask (SGPR pair) x8", w, 8, 0);
        run<16>("v_cmp -> SGPR pairs x8", w, 8, 0);
        run<19>("v_min_f32 independent x8", w, 8, 0);
        run<17>("v_min3_f32 x8", w, 8, 0);
        run<22>("v_max3_f32 x8", w, 8, 0);
        run<18>("v_mul_f32 chained x8", w, 8, 0);
        run<20>("v_add_f32 chained x8", w, 8, 0);
        run<21>("(v_cmp vcc + v_cndmask vcc) x4", w, 8, 0);
        run<23>("v_or_b32 / v_and_b32 x8", w, 8, 0);
        run<24>("v_add_u32 / v_lshlrev_b32 x8", w, 8, 0);
        run<25>("v_or3 / v_and_or / v_add3 x8", w, 8, 0);
        run<26>("v_max_i32 / v_min_u32 x8", w, 8, 0);
        run<27>("v_sqrt_f32 / v_rcp_f32 x8", w, 8, 0);
        run<28>("v_sub_f32 with |x| / -x x8", w, 8, 0);
    }
    return 0;
}
