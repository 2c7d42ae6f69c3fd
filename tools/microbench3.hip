// Do the matrix pipe and the vector ALU of a gfx950 SIMD overlap?  (round 4)
// profiles/r01_microbench_gfx950.txt showed float64 MFMA + float64 FMA serialising (2 MFMA + 8 FMA = 166 cycles = 128 + 32 + loop).
// This asks the same of the other combinations a pair kernel could use: float32 MFMA beside packed float32 FMAs (BASELINE configs[4]:
// the inner products of a step are half of its instructions), float64 MFMA beside non-float64 vector work (DPP moves, integer ops,
// float32), and bf16 MFMA beside float32 as the known-good case.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench3 tools/microbench3.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

constexpr int ITERS = 2048;

// NM MFMAs and NV vector instructions per iteration, interleaved.  KIND of the MFMA: 0 f32 32x32x2 (64 cycles), 1 f64 16x16x4 (64 cycles),
// 2 bf16 32x32x16 (gfx950: 8 passes = 32 cycles?), 3 f32 16x16x4 (32 cycles).  VK of the vector work: 0 v_pk_fma_f32, 1 v_fma_f64, 2 v_add_u32 chain,
// 3 v_mov_dpp (row_shr:1), 4 v_fma_f32 (unpacked, kept unpacked by dependent scalar ops)
template <int KIND, int NM, int VK, int NV>
__global__ __launch_bounds__(256) void k_mix(float* out, float a, float b) {
    f16v c32[2] = {f16v(0.f), f16v(0.f)};
    d4 c64[2] = {d4(0.0), d4(0.0)};
    f4 c16[2] = {f4(0.f), f4(0.f)};
    const float av = a + threadIdx.x, bv = b - threadIdx.x;
    const double ad = av, bd = bv;
    bf8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(av + i); bb[i] = (__bf16)(bv - i); }
    f2 x[8];
    double y[8];
    int z[8];
    float w[8];
    for (int i = 0; i < 8; ++i) { x[i] = f2{float(threadIdx.x + i), float(i)}; y[i] = threadIdx.x + i; z[i] = threadIdx.x * 3 + i; w[i] = threadIdx.x + i; }
    const f2 a2 = {a, a}, b2 = {b, b};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int m = 0; m < (NM > 0 ? NM : 1); ++m) {
            if (NM > 0) {
                if (KIND == 0) c32[m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c32[m & 1], 0, 0, 0);
                if (KIND == 1) c64[m & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad, bd, c64[m & 1], 0, 0, 0);
                if (KIND == 2) c32[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c32[m & 1], 0, 0, 0);
                if (KIND == 3) c16[m & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c16[m & 1], 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < NV / (NM > 0 ? NM : 1); ++v) {
                const int k = v & 7;
                if (VK == 0) x[k] = __builtin_elementwise_fma(x[k], a2, b2);
                if (VK == 1) y[k] = fma(y[k], (double)a, (double)b);
                if (VK == 2) z[k] = z[k] * 3 + it;
                if (VK == 3) z[k] = __builtin_amdgcn_update_dpp(0, z[k], 0x111, 0xf, 0xf, true) + 1;
                if (VK == 4) w[k] = fmaf(w[k], a, b);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y + float(y[i]) + float(z[i]) + w[i];
    s += c32[0][0] + c32[1][1] + float(c64[0][0] + c64[1][1]) + c16[0][0] + c16[1][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
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

template <int KIND, int NM, int VK, int NV>
static void run(const char* name, float* d, int CUS) {
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = CUS * wps;
        const double sec = time_kernel([&] { k_mix<KIND, NM, VK, NV><<<blocks, 256>>>(d, 1.0000001f, 1e-9f); });
        // cycles per iteration and SIMD (nominal 2.4 GHz; the chip clocks lower under load, so compare lines, not absolutes)
        printf("  %-44s wps=%d  %8.3f ms  %8.1f cyc/iter/wave@2.4GHz\n", name, wps, sec * 1e3, sec * 2.4e9 / (double(ITERS) * wps));
    }
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs %d\n", prop.gcnArchName, prop.multiProcessorCount);
    const int CUS = prop.multiProcessorCount;
    float* d; CK(hipMalloc(&d, sizeof(float) * 256 * CUS * 8));
    printf("-- alone\n");
    run<0, 4, 0, 0>("4 mfma_f32_32x32x2", d, CUS);
    run<3, 4, 0, 0>("4 mfma_f32_16x16x4", d, CUS);
    run<1, 4, 0, 0>("4 mfma_f64_16x16x4", d, CUS);
    run<2, 4, 0, 0>("4 mfma_f32_32x32x16_bf16", d, CUS);
    run<0, 0, 0, 32>("32 v_pk_fma_f32", d, CUS);
    run<0, 0, 4, 32>("32 v_fma_f32", d, CUS);
    run<0, 0, 1, 32>("32 v_fma_f64", d, CUS);
    run<0, 0, 2, 32>("32 v_mad_u32 (int)", d, CUS);
    run<0, 0, 3, 32>("32 (v_mov_dpp + v_add_u32)", d, CUS);
    printf("-- float32 MFMA beside float32 vector work\n");
    run<0, 4, 0, 32>("4 mfma_f32_32x32x2 + 32 v_pk_fma_f32", d, CUS);
    run<0, 4, 0, 64>("4 mfma_f32_32x32x2 + 64 v_pk_fma_f32", d, CUS);
    run<0, 4, 4, 64>("4 mfma_f32_32x32x2 + 64 v_fma_f32", d, CUS);
    run<3, 4, 0, 32>("4 mfma_f32_16x16x4 + 32 v_pk_fma_f32", d, CUS);
    printf("-- float64 MFMA beside other vector work\n");
    run<1, 4, 1, 32>("4 mfma_f64 + 32 v_fma_f64", d, CUS);
    run<1, 4, 0, 32>("4 mfma_f64 + 32 v_pk_fma_f32", d, CUS);
    run<1, 4, 2, 32>("4 mfma_f64 + 32 int", d, CUS);
    run<1, 4, 3, 32>("4 mfma_f64 + 32 (dpp + add)", d, CUS);
    printf("-- bf16 MFMA beside float32 vector work (the flash-attention case)\n");
    run<2, 4, 0, 32>("4 mfma_bf16 + 32 v_pk_fma_f32", d, CUS);
    run<2, 4, 0, 64>("4 mfma_bf16 + 64 v_pk_fma_f32", d, CUS);
    return 0;
}
