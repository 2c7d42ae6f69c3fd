// Device micro-benchmarks that size the seq-gram kernel design on gfx950:
// fp64 VALU issue rate, 32-bit DPP moves, fp64 exp, fp64 MFMA, LDS b128 broadcast reads.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

constexpr int ITERS = 4096;

__global__ void k_fma64(double* out, double a, double b) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; ++i) {
        x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
        x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_add64(double* out, double a, double b) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; ++i) {
        x0 = x0 + a; x1 = x1 + b; x2 = x2 + a; x3 = x3 + b; x4 = x4 + a; x5 = x5 + b; x6 = x6 + a; x7 = x7 + b;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_fma32(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; ++i) {
        x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
        x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
// 8 x (2 DPP moves + 1 fp64 add): the carry hand-off pattern
template <int CTRL>
__global__ void k_dpp(double* out, double a) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; ++i) {
        x0 = dpp_f64<CTRL>(x0) + a; x1 = dpp_f64<CTRL>(x1) + a; x2 = dpp_f64<CTRL>(x2) + a; x3 = dpp_f64<CTRL>(x3) + a;
        x4 = dpp_f64<CTRL>(x4) + a; x5 = dpp_f64<CTRL>(x5) + a; x6 = dpp_f64<CTRL>(x6) + a; x7 = dpp_f64<CTRL>(x7) + a;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_exp64(double* out, double a) {
    double x0 = threadIdx.x * 1e-3, x1 = x0 + .1, x2 = x0 + .2, x3 = x0 + .3;
    for (int i = 0; i < ITERS / 4; ++i) {
        x0 = exp(-x0 * a); x1 = exp(-x1 * a); x2 = exp(-x2 * a); x3 = exp(-x3 * a);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k_mfma64(double* out, double a, double b) {
    double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double av = a + threadIdx.x, bv = b - threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
// fp64 MFMA and fp64 VALU issued from the same wave: do the pipes overlap?
__global__ void k_mfma64_plus_fma(double* out, double a, double b) {
    double4_t c0 = {0, 0, 0, 0}, c1 = c0;
    double av = a + threadIdx.x, bv = b - threadIdx.x;
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c0, 0, 0, 0);
        x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c1, 0, 0, 0);
        x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
// per-lane 64-byte row reads (4 x ds_read_b128) + 8 fp64 FMAs per read batch
__global__ void k_lds_row(double* out, int stride) {
    __shared__ double4_t buf[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) buf[i] = double4_t{1.0 * i, 2.0, 3.0, 4.0};
    __syncthreads();
    double acc = 0;
    int row = (threadIdx.x & 15);
    for (int i = 0; i < ITERS; ++i) {
        const double4_t* p = &buf[((row + i) & 63) * 2 * stride];
        double4_t u = p[0], v = p[1];
        acc = fma(u[0], 1.5, acc); acc = fma(u[1], 1.5, acc); acc = fma(u[2], 1.5, acc); acc = fma(u[3], 1.5, acc);
        acc = fma(v[0], 1.5, acc); acc = fma(v[1], 1.5, acc); acc = fma(v[2], 1.5, acc); acc = fma(v[3], 1.5, acc);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// does wave_shr:1 (DPP ctrl 0x138) behave as a full-wave shift on gfx950?
__global__ void k_wave_shr_check(int* out) {
    int v = threadIdx.x + 100;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);     // row_shr:1 bound_ctrl
}

template <class F>
static double time_kernel(F launch, int reps = 5) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best * 1e-3;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs %d  clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    const int CUS = prop.multiProcessorCount;
    double* d; CK(hipMalloc(&d, sizeof(double) * 256 * CUS * 64));
    for (int wps = 1; wps <= 4; wps *= 2) {           // waves per SIMD
        int blocks = CUS * wps, thr = 256;            // 4 waves per block -> one per SIMD
        double waves = double(blocks) * thr / 64;
        auto report = [&](const char* name, double sec, double inst_per_wave, double flop_per_inst_lane) {
            double cyc_per_inst = sec * 2.4e9 / (inst_per_wave * wps);   // per SIMD at nominal 2.4 GHz
            printf("  %-22s wps=%d  %8.3f ms  %6.2f cyc/wave-inst@2.4GHz  %8.2f TFLOP/s\n", name, wps, sec * 1e3, cyc_per_inst,
                   waves * inst_per_wave * 64 * flop_per_inst_lane / sec * 1e-12);
        };
        report("v_fma_f64", time_kernel([&] { k_fma64<<<blocks, thr>>>(d, 1.0000001, 1e-9); }), ITERS * 8.0, 2);
        report("v_add_f64", time_kernel([&] { k_add64<<<blocks, thr>>>(d, 1.0000001, 1e-9); }), ITERS * 8.0, 1);
        report("v_fma_f32", time_kernel([&] { k_fma32<<<blocks, thr>>>((float*)d, 1.0000001f, 1e-9f); }), ITERS * 8.0, 2);
        report("2dpp+add64 row_shr1", time_kernel([&] { k_dpp<0x111><<<blocks, thr>>>(d, 1e-9); }), ITERS * 8.0, 1);
        report("2dpp+add64 wave_shr1", time_kernel([&] { k_dpp<0x138><<<blocks, thr>>>(d, 1e-9); }), ITERS * 8.0, 1);
        report("exp(f64)", time_kernel([&] { k_exp64<<<blocks, thr>>>(d, 1.0000001); }), ITERS * 1.0, 1);
        report("mfma_f64_16x16x4", time_kernel([&] { k_mfma64<<<blocks, thr>>>(d, 1.0000001, 1e-9); }), ITERS * 4.0, 2.0 * 16 * 16 * 4 / 64);
        report("2mfma64+8fma64", time_kernel([&] { k_mfma64_plus_fma<<<blocks, thr>>>(d, 1.0000001, 1e-9); }), ITERS * 1.0, 2 * (2.0 * 16 * 16 * 4 / 64) + 16);
        report("lds 64B row + 8 fma", time_kernel([&] { k_lds_row<<<blocks, thr>>>(d, 4); }), ITERS * 8.0, 2);
    }
    int* di; CK(hipMalloc(&di, 192 * sizeof(int)));
    k_wave_shr_check<<<1, 64>>>(di); CK(hipDeviceSynchronize());
    std::vector<int> h(192); CK(hipMemcpy(h.data(), di, 192 * sizeof(int), hipMemcpyDeviceToHost));
    printf("wave_shr:1   lanes 0,1,15,16,17,32,63 -> %d %d %d %d %d %d %d (expect -1 100 114 115 116 131 162)\n", h[0], h[1], h[15], h[16], h[17], h[32], h[63]);
    printf("row_shr:1    lanes 0,1,15,16,17,32,63 -> %d %d %d %d %d %d %d (expect -1 100 114 -1 116 -1 162)\n", h[64], h[65], h[79], h[80], h[81], h[96], h[127]);
    printf("row_shr:1 bc lanes 0,1,15,16,17,32,63 -> %d %d %d %d %d %d %d (expect 0 100 114 0 116 0 162)\n", h[128], h[129], h[143], h[144], h[145], h[160], h[191]);
    return 0;
}
