// Actual shader clock and per-instruction issue cycles, measured with s_memtime inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1;} } while (0)
constexpr int ITERS = 8192;
template <int KIND>
__global__ void k(double* out, long long* cyc, double a, double b) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    long long t0 = __builtin_readcyclecounter();
    long long w0 = wall_clock64();
    for (int i = 0; i < ITERS; ++i) {
        if (KIND == 0) { x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b); x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b); }
        if (KIND == 1) { x0 = x0 + a; x1 = x1 + b; x2 = x2 + a; x3 = x3 + b; x4 = x4 + a; x5 = x5 + b; x6 = x6 + a; x7 = x7 + b; }
        if (KIND == 2) {
#define DPP64(v) { int lo = __double2loint(v), hi = __double2hiint(v); lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true); v = __hiloint2double(hi, lo); }
            DPP64(x0) DPP64(x1) DPP64(x2) DPP64(x3) DPP64(x4) DPP64(x5) DPP64(x6) DPP64(x7)
        }
        if (KIND == 3) {   // 4 fma + 2 dpp64 interleaved (the kernel's mix)
            x0 = fma(x0, a, b); x1 = fma(x1, a, b); DPP64(x4) x2 = fma(x2, a, b); x3 = fma(x3, a, b); DPP64(x5)
            x6 = fma(x6, a, b); x7 = fma(x7, a, b);
        }
        if (KIND == 4) {   // 32-bit integer VALU
            int i0 = __double2loint(x0), i1 = __double2loint(x1), i2 = __double2loint(x2), i3 = __double2loint(x3);
            i0 = i0 * 3 + i; i1 = i1 * 5 + i; i2 = i2 * 7 + i; i3 = i3 * 9 + i;
            x0 = __hiloint2double(__double2hiint(x0), i0); x1 = __hiloint2double(__double2hiint(x1), i1);
            x2 = __hiloint2double(__double2hiint(x2), i2); x3 = __hiloint2double(__double2hiint(x3), i3);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    long long w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = w1 - w0; }
}
template <int KIND> int run(const char* name, int wps, double per_iter_insts) {
    int blocks = 256 * wps;
    double* d; long long* c;
    CK(hipMalloc(&d, sizeof(double) * blocks * 256)); CK(hipMalloc(&c, sizeof(long long) * 2 * blocks));
    k<KIND><<<blocks, 256>>>(d, c, 1.0000001, 1e-9); CK(hipDeviceSynchronize());
    k<KIND><<<blocks, 256>>>(d, c, 1.0000001, 1e-9); CK(hipDeviceSynchronize());
    std::vector<long long> h(2 * blocks); CK(hipMemcpy(h.data(), c, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost));
    double cs = 0, ws = 0; for (int i = 0; i < blocks; ++i) { cs += h[2 * i]; ws += h[2 * i + 1]; }
    cs /= blocks; ws /= blocks;
    // wall_clock64 ticks at 100 MHz
    printf("%-28s wps=%d  shader cycles/iter %.1f  -> %.2f cyc per wave-inst (x%d waves/SIMD => %.2f issue cycles)  clock %.3f GHz\n", name, wps,
           cs / ITERS, cs / ITERS / per_iter_insts, wps, cs / ITERS / per_iter_insts / wps, cs / (ws / 100e6) / 1e9);
    hipFree(d); hipFree(c); return 0;
}
int main() {
    for (int wps = 1; wps <= 4; wps *= 2) {
        run<0>("8 x v_fma_f64", wps, 8);
        run<1>("8 x v_add_f64", wps, 8);
        run<2>("16 x v_mov_b32_dpp", wps, 16);
        run<3>("6 fma64 + 4 dpp32", wps, 10);
        run<4>("int ops", wps, 8);
    }
    return 0;
}
