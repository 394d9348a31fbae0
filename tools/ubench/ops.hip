// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the fp64 / conversion instructions the
// 2-opt kernels are made of.  Each wave runs a long chain of 8 independent streams of one instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang fp contract(off)

#define ITER 2048
template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double seed) {
    double a[8];
    float f[8];
    for (int q = 0; q < 8; ++q) { a[q] = seed + threadIdx.x * 1e-3 + q; f[q] = (float)a[q]; }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (OP == 0) a[q] = __builtin_amdgcn_sqrt(a[q]) + 1e6;            // v_sqrt_f64 (+ add)
            if (OP == 1) a[q] = a[q] + 1e6;                                    // v_add_f64 alone
            if (OP == 2) a[q] = fma(a[q], 1.0000001, 0.5);                     // v_fma_f64
            if (OP == 3) a[q] = __builtin_amdgcn_rsq(a[q]) + 1e6;              // v_rsq_f64 (+ add)
            if (OP == 4) f[q] = __builtin_amdgcn_sqrtf(f[q]) + 1e6f;           // v_sqrt_f32 (+ add f32)
            if (OP == 5) f[q] = f[q] + 1e6f;                                   // v_add_f32
            if (OP == 6) { f[q] = (float)a[q]; a[q] = a[q] + (double)f[q]; }   // cvt f32<-f64, cvt f64<-f32, add
            if (OP == 7) a[q] = floor(a[q]) + 1.5;                             // v_floor_f64 (+ add)
            if (OP == 8) a[q] = rint(a[q]) + 1.5;                              // v_rndne_f64 (+ add)
            if (OP == 9) a[q] = (a[q] > 3.0) ? a[q] * 0.999 : a[q] + 1.0;      // cmp + mul + add + cndmask x2
            if (OP == 10) f[q] = __builtin_amdgcn_rsqf(f[q]) + 1e6f;           // v_rsq_f32 (+ add f32)
            if (OP == 11) a[q] = a[q] * 1.0000001;                             // v_mul_f64
            if (OP == 12) a[q] = trunc(a[q]) + 1.5;                            // v_trunc_f64 (+ add)
        }
    }
    double s = 0;
    for (int q = 0; q < 8; ++q) s += a[q] + f[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
double run(const char *name, int extra_adds) {
    const int blocks = 256 * 4;  // 4 blocks of 4 waves per CU: 4 waves per SIMD
    double *d; hipMalloc(&d, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x ITER x 8 ops
    const double ops_per_simd = 4.0 * ITER * 8;
    const double cycles = ms * 1e-3 * 2.4e9;   // nominal clock
    printf("%-34s %8.3f ms  %6.1f cycles per wave-op (at 2.4 GHz nominal)\n", name, ms, cycles / ops_per_simd);
    hipFree(d);
    return cycles / ops_per_simd;
}

int main() {
    run<1>("v_add_f64", 0);
    run<11>("v_mul_f64", 0);
    run<2>("v_fma_f64", 0);
    run<0>("v_sqrt_f64 + add", 1);
    run<3>("v_rsq_f64 + add", 1);
    run<5>("v_add_f32", 0);
    run<4>("v_sqrt_f32 + add_f32", 1);
    run<10>("v_rsq_f32 + add_f32", 1);
    run<6>("cvt_f32_f64 + cvt_f64_f32 + add_f64", 0);
    run<7>("v_floor_f64 + add", 1);
    run<8>("v_rndne_f64 + add", 1);
    run<12>("v_trunc_f64 + add", 1);
    run<9>("cmp+mul+add+2cndmask (f64)", 0);
    return 0;
}
