// (1) Is s_memtime a shader-cycle counter on gfx950 under float64 load (DVFS)?  (2) How many wavefronts per SIMD does it take to
// fill the float64 issue port?  Workgroups of 256 threads (one wavefront per SIMD of a CU), W of them per CU, run an issue-bound loop
// of independent v_fma_f64 (or a mix with 32-bit DPP moves, as the pair kernels' hand-overs) and bracket it with s_memtime /
// s_memrealtime.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/clockcheck tools/clockcheck.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int MIX>
__global__ void __launch_bounds__(256) k(double* out, unsigned long long* t, double a, double b, int iters) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    int i0 = threadIdx.x, i1 = i0 + 1;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
        x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
        if (MIX == 1) {      // two 32-bit DPP moves per eight FMAs (the pair kernels' ratio is 10 per 73)
            i0 = __builtin_amdgcn_update_dpp(0, i0, 0x111, 0xf, 0xf, true);
            i1 = __builtin_amdgcn_update_dpp(0, i1, 0x111, 0xf, 0xf, true);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + i0 + i1;
    if ((threadIdx.x & 63) == 0) { t[2 * (blockIdx.x * 4 + threadIdx.x / 64)] = c1 - c0; t[2 * (blockIdx.x * 4 + threadIdx.x / 64) + 1] = r1 - r0; }
}

template <int MIX>
void run(int iters, int wps) {
    const int blocks = 256 * wps, waves = blocks * 4;
    double* out; unsigned long long* t;
    CK(hipMalloc(&out, sizeof(double) * 256 * blocks)); CK(hipMalloc(&t, 16 * waves));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MIX>, dim3(blocks), dim3(256), 0, 0, out, t, 0.999, 1e-3, 1000);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<MIX>, dim3(blocks), dim3(256), 0, 0, out, t, 0.999, 1e-3, iters); CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * waves);
    CK(hipMemcpy(h.data(), t, 16 * waves, hipMemcpyDeviceToHost));
    std::vector<double> dc, dr;
    for (int i = 0; i < waves; ++i) { dc.push_back(double(h[2 * i])); dr.push_back(double(h[2 * i + 1])); }
    std::sort(dc.begin(), dc.end()); std::sort(dr.begin(), dr.end());
    const double mc = dc[waves / 2], mr = dr[waves / 2];
    const double insts = (8.0 + (MIX ? 2.0 : 0.0)) * iters * wps;       // vector instructions one SIMD issued during a wave's life
    printf("%s  %d wave(s)/SIMD: event %.3f ms | median wave: %.4g cycles (min %.4g max %.4g), %.3f ms | memtime/realtime %.4f GHz | cycles per vector instruction of the SIMD %.3f | "
           "fp64 rate %.1f TFLOP/s\n", MIX ? "8 v_fma_f64 + 2 v_mov_dpp" : "8 v_fma_f64              ", wps, ms, mc, dc.front(), dc.back(), mr * 1e-5, mc / mr * 0.1, mc / insts,
           16.0 * iters * 64.0 * waves / (ms * 1e-3) / 1e12);
    CK(hipFree(out)); CK(hipFree(t));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    for (int wps : {1, 2, 3, 4, 6, 8}) run<0>(iters, wps);
    for (int wps : {1, 2, 3, 4}) run<1>(iters, wps);
    return 0;
}
