// Microbenchmark: issue rate of scalar vs packed fp32 VALU ops on gfx950, per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));

#define REP 64
template <int MODE>
__global__ void k(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2_ p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    float2_ pa = {a, b};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (MODE == 0) {  // 8 independent scalar v_mul_f32
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
            } else if (MODE == 1) {  // 4 independent v_pk_mul_f32 (same number of element ops)
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));
            } else if (MODE == 2) {  // 8 scalar v_add
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
            } else if (MODE == 3) {  // 4 pk_add
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));
            } else if (MODE == 4) {  // 8 scalar v_mul with SGPR operand
                asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                             "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a));
            } else if (MODE == 5) {  // v_readlane (4) + 4 muls
                float s0, s1, s2, s3;
                asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %5, 3\n v_readlane_b32 %2, %6, 3\n v_readlane_b32 %3, %7, 3\n"
                             : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(x4), "v"(x5), "v"(x6), "v"(x7));
                asm volatile("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %5, %1\n v_mul_f32 %2, %6, %2\n v_mul_f32 %3, %7, %3\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(s0), "s"(s1), "s"(s2), "s"(s3));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
void run(const char *name, int waves_per_simd, int elem_ops_per_rep) {
    float *d; hipMalloc(&d, 256 * 4 * 1024 * 64 * sizeof(float));
    int iters = 2000;
    int blocks = 256 * 4;  // 4 blocks per CU of 256*waves... use block = 64*waves_per_simd, grid = 256 CUs * 4 SIMDs
    int block = 64 * waves_per_simd;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, block>>>(d, 10, 1.0001f, 0.9999f);
    hipEventRecord(a);
    k<MODE><<<blocks, block>>>(d, iters, 1.0001f, 0.9999f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double elem_ops = (double)iters * REP * elem_ops_per_rep * 64.0 * waves_per_simd * blocks;
    // cycles per wave-level element-op-group per SIMD at 2.4 GHz
    double cyc = ms * 1e-3 * 2.4e9;
    double per_simd_wave_ops = (double)iters * REP * elem_ops_per_rep * waves_per_simd;  // wave64 element-op instr-equivalents per SIMD
    printf("%-28s waves/SIMD %d: %8.3f ms  %7.2f Tops/s  %.2f cycles per wave64 elem-op (at 2.4 GHz)\n", name, waves_per_simd, ms,
           elem_ops / ms / 1e9, cyc / per_simd_wave_ops);
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_mul_f32 x8", w, 8);
        run<1>("v_pk_mul_f32 x4", w, 8);
        run<2>("v_add_f32 x8", w, 8);
        run<3>("v_pk_add_f32 x4", w, 8);
        run<4>("v_mul_f32 sgpr x8", w, 8);
        run<5>("4 readlane + 4 mul", w, 8);
    }
    return 0;
}
