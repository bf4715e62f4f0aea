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
    printf("%-30s %d waves/SIMD: %7.3f ms, clock %.2f GHz: per CU and cycle %.2f VALU + %.2f SALU\n", name, wgs_per_cu, ms, ghz,
           per_cu * valu_per_rep / cycles, per_cu * salu_per_rep / cycles);
    hipFree(d); hipFree(clk);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("s_add_u32 x8", w, 0, 8);
        run<6>("s_and/or_b64 x8", w, 0, 8);
        run<1>("v_add_f32 x8", w, 8, 0);
        run<2>("v_add x8 + s_add x8", w, 8, 8);
        run<3>("v_add x8 + s_add x4", w, 8, 4);
        run<5>("v_add x8 + s_add x2", w, 8, 2);
        run<4>("(v_cmp + s_and_b64) x8", w, 8, 8);
        run<7>("v_add x8 + s_nop x8", w, 8, 8);
        run<8>("(v_cmp + v_cndmask) x4", w, 8, 0);
    }
    return 0;
}
