// Micro-benchmark: cycles per exact integer-root distance (two_opt_exh.hpp: exh_dist) against the independent distances a
// wave has in flight (NCH) and the waves per SIMD -- what limits k_exh's row loop.  Row operands uniform (SGPR), like the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)
#define ITER 2048
__device__ __forceinline__ int dist(double cx, double cy, double rx, double ry) {
    const double dx = cx - rx, dy = cy - ry;
    const double s = __builtin_fma(dx, dx, dy * dy);
    const double k = floor(__builtin_amdgcn_sqrt(s) + 0.25);
    const double e = __builtin_fma(-k, k, s);
    return __double2int_rz(k) + (e > k ? 1 : 0);
}
template <int NCH>
__global__ __launch_bounds__(256) void k(int *out, const double2 *__restrict__ rows, int nrows) {
    double cx[NCH], cy[NCH];
    int acc[NCH];
    for (int q = 0; q < NCH; ++q) { cx[q] = 1000.0 + threadIdx.x * 37 + q * 1001; cy[q] = 777.0 + threadIdx.x * 11 + q * 313; acc[q] = 0; }
    int p = __builtin_amdgcn_readfirstlane((int)blockIdx.x) % nrows;
    for (int it = 0; it < ITER; ++it) {
        const double2 r = rows[p];          // wave-uniform: scalar load
        p = p + 1 == nrows ? 0 : p + 1;
#pragma unroll
        for (int q = 0; q < NCH; ++q) acc[q] += dist(cx[q], cy[q], r.x, r.y);
    }
    int s = 0;
    for (int q = 0; q < NCH; ++q) s += acc[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NCH>
void run(int waves_per_simd, int *d, double2 *rows) {
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, d, rows, 4096);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, d, rows, 4096);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)waves_per_simd * ITER * NCH;
    printf("chains per wave %d, waves per SIMD %d: %7.3f ms, %6.1f ns per wave-distance per SIMD (= %5.1f cycles at 2.15 GHz)\n", NCH, waves_per_simd, ms,
           ms * 1e6 / per_simd, ms * 1e-3 * 2.15e9 / per_simd);
}
int main() {
    int *d; (void)hipMalloc(&d, 4 * 256 * 8 * 256);
    double2 *rows; (void)hipMalloc(&rows, 4096 * sizeof(double2));
    (void)hipMemset(rows, 0, 4096 * sizeof(double2));
    for (int w : {1, 2, 4, 8}) { run<1>(w, d, rows); run<2>(w, d, rows); run<4>(w, d, rows); run<8>(w, d, rows); }
    return 0;
}
