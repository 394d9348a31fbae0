// Micro-benchmark: what one exact integer-root distance of the exhaustive sweep (two_opt_exh.hpp: exh_dist) costs a SIMD, in
// cycles per wave, for several ways of rounding the raw root, plus the conversion instructions on their own.
// 4 waves per SIMD, 8 independent columns per lane, no memory traffic in the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)

#define ITER 4096
template <int V>
__device__ __forceinline__ int dist(double cx, double cy, double rx, double ry) {
    const double dx = cx - rx, dy = cy - ry;
    const double s = __builtin_fma(dx, dx, dy * dy);
    const double g = __builtin_amdgcn_sqrt(s);
    if (V == 0) {          // floor + cvt (the first form)
        const double k = floor(g + 0.25);
        const double e = __builtin_fma(-k, k, s);
        return __double2int_rz(k) + (e > k ? 1 : 0);
    } else if (V == 1) {   // magic constant: the integer falls out of the low word, no floor, no cvt
        const double t = (g - 0.25) + 6755399441055744.0;   // 1.5 * 2^52: round to nearest integer
        const double k = t - 6755399441055744.0;
        const double e = __builtin_fma(-k, k, s);
        return __double2loint(t) + (e > k ? 1 : 0);
    } else if (V == 2) {   // rndne + cvt
        const double k = rint(g - 0.25);
        const double e = __builtin_fma(-k, k, s);
        return __double2int_rz(k) + (e > k ? 1 : 0);
    } else if (V == 3) {   // no fix-up at all (not exact: the floor of the issue cost)
        return __double2loint(g + 6755399441055744.0);
    } else {               // f32 seed: s rounded to f32, v_sqrt_f32, two-sided fix-up in f64
        const float gf = __builtin_amdgcn_sqrtf((float)s);
        const double t = (double)gf + 6755399441055744.0;
        const double k = t - 6755399441055744.0;
        const double e = __builtin_fma(-k, k, s);           // s - k^2
        return __double2loint(t) + (e > k ? 1 : 0) - (e < -k ? 1 : 0);
    }
}

template <int V>
__global__ __launch_bounds__(256) void k(int *out, double seed) {
    double cx[8], cy[8];
    int acc[8];
    for (int q = 0; q < 8; ++q) { cx[q] = seed * 1000 + threadIdx.x * 37 + q * 1001; cy[q] = seed * 777 + threadIdx.x * 11 + q * 313; acc[q] = 0; }
    double rx = seed, ry = seed * 2;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += dist<V>(cx[q], cy[q], rx, ry);
        rx += 3.0; ry += 5.0;
    }
    int s = 0;
    for (int q = 0; q < 8; ++q) s += acc[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
__global__ __launch_bounds__(256) void kop(double *out, double seed) {
    double a[8];
    int i[8];
    for (int q = 0; q < 8; ++q) { a[q] = seed + threadIdx.x * 1e-3 + q; i[q] = threadIdx.x + q; }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (OP == 0) a[q] = floor(a[q]) + 1.5;                                   // v_floor_f64 + add
            if (OP == 1) { i[q] += __double2int_rz(a[q]); a[q] += 1.5; }            // v_cvt_i32_f64 + add_u32 + add_f64
            if (OP == 2) a[q] = a[q] + 1.5;                                          // add alone
            if (OP == 3) { a[q] = (double)i[q] + a[q]; i[q] += 3; }                  // v_cvt_f64_i32 + add_f64 + add_u32
            if (OP == 4) { i[q] += (a[q] > 3.0) ? 1 : 0; a[q] += 1.5; }              // v_cmp_gt_f64 + addc + add_f64
            if (OP == 5) { i[q] = min(min(i[q] + 3, i[(q + 1) & 7]), i[(q + 2) & 7]); }   // add + min3
            if (OP == 6) a[q] = rint(a[q]) + 1.5;                                    // v_rndne_f64 + add
            if (OP == 7) { a[q] = (double)(float)a[q] + 1.5; }                       // cvt_f32_f64 + cvt_f64_f32 + add
        }
    }
    double s = 0;
    for (int q = 0; q < 8; ++q) s += a[q] + i[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
double timeit(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int blocks = 256 * 4;   // 4 waves per SIMD
    void *d; hipMalloc(&d, 8 * blocks * 256);
    const double per_simd = 4.0 * ITER * 8;   // wave-level calls per SIMD
    const char *names[] = {"floor + cvt_i32 (first form)", "magic constant (no floor, no cvt)", "rndne + cvt_i32", "no fix-up (not exact)", "f32 seed, two-sided fix-up"};
    double ms[5];
    ms[0] = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, (int *)d, 3.0); });
    ms[1] = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, (int *)d, 3.0); });
    ms[2] = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, (int *)d, 3.0); });
    ms[3] = timeit([&] { hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, (int *)d, 3.0); });
    ms[4] = timeit([&] { hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, (int *)d, 3.0); });
    for (int v = 0; v < 5; ++v) printf("exact distance, %-36s %8.3f ms  %6.1f cycles per wave-distance (2.4 GHz nominal)\n", names[v], ms[v], ms[v] * 1e-3 * 2.4e9 / per_simd);
    const char *on[] = {"v_floor_f64 + add_f64", "v_cvt_i32_f64 + add_u32 + add_f64", "v_add_f64", "v_cvt_f64_i32 + add_f64 + add_u32", "v_cmp_gt_f64 + addc + add_f64", "add_u32 + min3_i32", "v_rndne_f64 + add_f64", "cvt_f32_f64 + cvt_f64_f32 + add_f64"};
    double mo[8];
    mo[0] = timeit([&] { hipLaunchKernelGGL(kop<0>, dim3(blocks), dim3(256), 0, 0, (double *)d, 3.0); });
    mo[1] = timeit([&] { hipLaunchKernelGGL(kop<1>, dim3(blocks), dim3(256), 0, 0, (double *)d, 3.0); });
    mo[2] = timeit([&] { hipLaunchKernelGGL(kop<2>, dim3(blocks), dim3(256), 0, 0, (double *)d, 3.0); });
    mo[3] = timeit([&] { hipLaunchKernelGGL(kop<3>, dim3(blocks), dim3(256), 0, 0, (double *)d, 3.0); });
    mo[4] = timeit([&] { hipLaunchKernelGGL(kop<4>, dim3(blocks), dim3(256), 0, 0, (double *)d, 3.0); });
    mo[5] = timeit([&] { hipLaunchKernelGGL(kop<5>, dim3(blocks), dim3(256), 0, 0, (double *)d, 3.0); });
    mo[6] = timeit([&] { hipLaunchKernelGGL(kop<6>, dim3(blocks), dim3(256), 0, 0, (double *)d, 3.0); });
    mo[7] = timeit([&] { hipLaunchKernelGGL(kop<7>, dim3(blocks), dim3(256), 0, 0, (double *)d, 3.0); });
    for (int v = 0; v < 8; ++v) printf("%-40s %8.3f ms  %6.1f cycles per wave-iteration\n", on[v], mo[v], mo[v] * 1e-3 * 2.4e9 / per_simd);
    return 0;
}
