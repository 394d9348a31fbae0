// Micro-benchmark: k_exh's row step piece by piece -- the four exact distances of a lane (tools/ubench/dist2.hip) plus, switched
// on one at a time, what follows them in the kernel: the DPP-fed sums, the min3 / compare / branch on a scalar bound, the update
// of the kept sums.  4 chains per wave; waves per SIMD on the command line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#pragma clang fp contract(off)
#define ITER 2048
__device__ __forceinline__ int dist(double cx, double cy, double rx, double ry) {
    const double dx = cx - rx, dy = cy - ry;
    const double s = __builtin_fma(dx, dx, dy * dy);
    const double k = floor(__builtin_amdgcn_sqrt(s) + 0.25);
    const double e = __builtin_fma(-k, k, s);
    return __double2int_rz(k) + (e > k ? 1 : 0);
}
template <int TAIL>
__global__ __launch_bounds__(256) void k(int *out, const double2 *__restrict__ rows, const int *__restrict__ er, int nrows, int wbd) {
    constexpr int NCH = 4;
    double cx[NCH], cy[NCH];
    int S[NCH], ce[NCH], acc = 0, bd = wbd;
    for (int q = 0; q < NCH; ++q) { cx[q] = 1000.0 + threadIdx.x * 37 + q * 1001; cy[q] = 777.0 + threadIdx.x * 11 + q * 313; S[q] = q; ce[q] = 3 + q; }
    int p = __builtin_amdgcn_readfirstlane((int)blockIdx.x) % nrows;
    for (int it = 0; it < ITER; ++it) {
        const double2 r = rows[p];
        const int erow = er[p];
        p = p + 1 == nrows ? 0 : p + 1;
        int D[NCH], sum[NCH];
#pragma unroll
        for (int q = 0; q < NCH; ++q) D[q] = dist(cx[q], cy[q], r.x, r.y);
        if (TAIL & 1) {
            sum[0] = __builtin_amdgcn_update_dpp(0, S[NCH - 1], 0x138, 0xf, 0xf, true) + D[0];
#pragma unroll
            for (int q = 1; q < NCH; ++q) sum[q] = S[q - 1] + D[q];
        } else {
#pragma unroll
            for (int q = 0; q < NCH; ++q) sum[q] = D[q];
        }
        if (TAIL & 2) {
            const int thr = bd + erow;
            bool hit = false;
#pragma unroll
            for (int q = 0; q < NCH; ++q) hit = hit || sum[q] <= thr;
            if (__builtin_expect(__any(hit), 0)) { acc += sum[0]; bd = min(bd, __builtin_amdgcn_readfirstlane(sum[1])); }
        } else {
#pragma unroll
            for (int q = 0; q < NCH; ++q) acc += sum[q];
        }
        if (TAIL & 4) {
#pragma unroll
            for (int q = 0; q < NCH; ++q) S[q] = D[q] - ce[q];
        }
    }
    int s = acc;
    for (int q = 0; q < NCH; ++q) s += S[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int TAIL>
void run(int waves_per_simd, int *d, double2 *rows, int *er) {
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<TAIL>, dim3(blocks), dim3(256), 0, 0, d, rows, er, 4096, -100000000);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<TAIL>, dim3(blocks), dim3(256), 0, 0, d, rows, er, 4096, -100000000);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)waves_per_simd * ITER;
    printf("tail %d, waves per SIMD %d: %7.3f ms, %6.1f ns per wave row step per SIMD (= %5.1f cycles at 2.15 GHz; 4 distances alone: 241)\n", TAIL, waves_per_simd, ms,
           ms * 1e6 / per_simd, ms * 1e-3 * 2.15e9 / per_simd);
}
int main(int argc, char **argv) {
    int *d; (void)hipMalloc(&d, 4 * 256 * 8 * 256);
    double2 *rows; (void)hipMalloc(&rows, 4096 * sizeof(double2));
    int *er; (void)hipMalloc(&er, 4096 * sizeof(int));
    (void)hipMemset(rows, 0, 4096 * sizeof(double2)); (void)hipMemset(er, 0, 4096 * sizeof(int));
    for (int w : {1, 4}) { run<0>(w, d, rows, er); run<1>(w, d, rows, er); run<2>(w, d, rows, er); run<3>(w, d, rows, er); run<4>(w, d, rows, er); run<7>(w, d, rows, er); }
    return 0;
}
