// Is s_memtime a shader-cycle counter on gfx950 under float64 load (DVFS)?  One wavefront per SIMD runs an issue-bound loop of
// independent v_fma_f64 (4 cycles per wave instruction: 32 * ITERS shader cycles, whatever the clock) and brackets it with
// s_memtime / s_memrealtime.  If s_memtime ticks per shader cycle, d(memtime) = 32 * ITERS; if it ticks at a fixed rate, it grows
// with 1 / clock.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/clockcheck tools/clockcheck.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(64) k(double* out, unsigned long long* t, double a, double b, int iters) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
        x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = c1 - c0; t[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400000;
    for (int waves_per_simd : {1, 2, 3}) {
        for (int blocks : {1, 1024 * waves_per_simd}) {
            double* out; unsigned long long* t;
            CK(hipMalloc(&out, sizeof(double) * 64 * blocks)); CK(hipMalloc(&t, 16 * blocks));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, t, 0.999, 1e-3, 1000);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, t, 0.999, 1e-3, iters); CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h(2 * blocks);
            CK(hipMemcpy(h.data(), t, 16 * blocks, hipMemcpyDeviceToHost));
            std::vector<double> dc, dr;
            for (int i = 0; i < blocks; ++i) { dc.push_back(double(h[2 * i])); dr.push_back(double(h[2 * i + 1])); }
            std::sort(dc.begin(), dc.end()); std::sort(dr.begin(), dr.end());
            const double mc = dc[blocks / 2], mr = dr[blocks / 2];
            const double issue_cycles = 32.0 * iters * (blocks == 1 ? 1 : waves_per_simd);
            printf("blocks %5d (%d wave/SIMD) iters %d: event %.3f ms | median d(memtime) %.4g = %.4f x issue cycles (%g) | d(realtime) %.4g (%.3f ms) | memtime/realtime %.4f GHz | "
                   "issue cycles / realtime = %.4f GHz\n", blocks, blocks == 1 ? 1 : waves_per_simd, iters, ms, mc, mc / issue_cycles, issue_cycles, mr, mr * 1e-5,
                   mc / mr * 0.1, issue_cycles / mr * 0.1);
            CK(hipFree(out)); CK(hipFree(t));
        }
    }
    return 0;
}
