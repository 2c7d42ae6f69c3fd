// Issue cost of single vector instructions on gfx950 at W wavefronts per SIMD (round 5: what an "instruction diet" of the float64 pair
// kernels can trade against what).  Each kernel runs a loop of 16 independent instances of one instruction (inline asm, distinct registers)
// bracketed by s_memtime; blocks of 256 threads, W blocks per CU, so every SIMD holds W wavefronts.  Printed: shader cycles per instruction
// and SIMD (wave cycles / (instructions x W)).  Mixes interleave two kinds 1:1 to show whether they share an issue slot.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench4 tools/microbench4.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

enum { FMA64, ADD64, MUL64, LDEXP64, RNDNE64, CVTI32F64, CNDMASK32, MOVDPP32, AND32, LSHLADD32, BFE32, MOV64, FMA64_SGPR, FMA32, DSREAD64,
       MIX_FMA64_AND32, MIX_FMA64_DPP, MIX_FMA64_CNDMASK, MIX_FMA64_DSREAD, MIX_FMA64_FMA32, NOPS };
static const char* names[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_ldexp_f64", "v_rndne_f64", "v_cvt_i32_f64", "v_cndmask_b32", "v_mov_b32 dpp row_shr:1",
                              "v_and_b32", "v_lshl_add_u32", "v_bfe_u32", "v_mov_b64", "v_fma_f64 (sgpr operand)", "v_fma_f32", "ds_read_b64 (same address)",
                              "1:1 v_fma_f64 + v_and_b32", "1:1 v_fma_f64 + v_mov_b32 dpp", "1:1 v_fma_f64 + v_cndmask_b32", "1:1 v_fma_f64 + ds_read_b64",
                              "1:1 v_fma_f64 + v_fma_f32"};

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ void __launch_bounds__(256) k(double* out, unsigned long long* t, double a, double b, int iters) {
    __shared__ double lds[64];
    if (threadIdx.x < 64) lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    double x[16];
    float f[16];
    int n[16];
    for (int i = 0; i < 16; ++i) { x[i] = threadIdx.x + i; f[i] = float(threadIdx.x) + i; n[i] = threadIdx.x * 7 + i; }
    const unsigned laddr = (threadIdx.x & 7) * 8;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#define F64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
#define A64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(b));
#define M64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define L64(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x[i]) : "v"(n[i] & 1));
#define R64(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(x[i]));
#define C64(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[i]) : "v"(x[i]));
#define CM(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(n[(i + 1) & 15]));
#define DPP(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(n[i]));
#define AND(i) asm volatile("v_and_b32 %0, 0x3ff, %0" : "+v"(n[i]));
#define LSA(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 15]));
#define BFE(i) asm volatile("v_bfe_u32 %0, %0, 0, 10" : "+v"(n[i]));
#define MV64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(x[i]) : "v"(x[(i + 1) & 15]));
#define F64S(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(b));
#define F32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(float(a)), "v"(float(b)));
#define DSR(i) asm volatile("ds_read_b64 %0, %1" : "=v"(x[i]) : "v"(laddr));
        if constexpr (OP == FMA64) { REP16(F64) }
        else if constexpr (OP == ADD64) { REP16(A64) }
        else if constexpr (OP == MUL64) { REP16(M64) }
        else if constexpr (OP == LDEXP64) { REP16(L64) }
        else if constexpr (OP == RNDNE64) { REP16(R64) }
        else if constexpr (OP == CVTI32F64) { REP16(C64) }
        else if constexpr (OP == CNDMASK32) { REP16(CM) }
        else if constexpr (OP == MOVDPP32) { REP16(DPP) }
        else if constexpr (OP == AND32) { REP16(AND) }
        else if constexpr (OP == LSHLADD32) { REP16(LSA) }
        else if constexpr (OP == BFE32) { REP16(BFE) }
        else if constexpr (OP == MOV64) { REP16(MV64) }
        else if constexpr (OP == FMA64_SGPR) { REP16(F64S) }
        else if constexpr (OP == FMA32) { REP16(F32) }
        else if constexpr (OP == DSREAD64) { REP16(DSR) asm volatile("s_waitcnt lgkmcnt(0)"); }
#define MIXA(i) F64(i) AND(i)
#define MIXD(i) F64(i) DPP(i)
#define MIXC(i) F64(i) CM(i)
#define MIXL(i) F64(i) asm volatile("ds_read_b64 %0, %1" : "=v"(x[(i + 8) & 15]) : "v"(laddr));
#define MIXF(i) F64(i) F32(i)
        else if constexpr (OP == MIX_FMA64_AND32) { REP16(MIXA) }
        else if constexpr (OP == MIX_FMA64_DPP) { REP16(MIXD) }
        else if constexpr (OP == MIX_FMA64_CNDMASK) { REP16(MIXC) }
        else if constexpr (OP == MIX_FMA64_DSREAD) { MIXL(0) MIXL(1) MIXL(2) MIXL(3) MIXL(4) MIXL(5) MIXL(6) MIXL(7) asm volatile("s_waitcnt lgkmcnt(0)"); MIXL(0) MIXL(1) MIXL(2) MIXL(3) MIXL(4) MIXL(5) MIXL(6) MIXL(7) asm volatile("s_waitcnt lgkmcnt(0)"); }
        else if constexpr (OP == MIX_FMA64_FMA32) { REP16(MIXF) }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += x[i] + f[i] + n[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { t[2 * (blockIdx.x * 4 + threadIdx.x / 64)] = c1 - c0; t[2 * (blockIdx.x * 4 + threadIdx.x / 64) + 1] = r1 - r0; }
}

template <int OP>
void run(int iters, int wps) {
    const int blocks = 256 * wps, waves = blocks * 4;
    double* out; unsigned long long* t;
    CK(hipMalloc(&out, sizeof(double) * 256 * blocks)); CK(hipMalloc(&t, 16 * waves));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, t, 0.999, 1e-3, 100);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, t, 0.999, 1e-3, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> hh(2 * waves), h(waves), hr(waves);
    CK(hipMemcpy(hh.data(), t, 16 * waves, hipMemcpyDeviceToHost));
    for (int i = 0; i < waves; ++i) { h[i] = hh[2 * i]; hr[i] = hh[2 * i + 1]; }
    std::sort(h.begin(), h.end());
    std::sort(hr.begin(), hr.end());
    const double per_iter = OP >= MIX_FMA64_AND32 ? 32.0 : 16.0;
    // wall clock: instructions one SIMD issued / the launch's event time -> ns per instruction and SIMD, and the clock that equates the two
    const double ns = double(ms) * 1e6 / (per_iter * iters * wps), cyc = double(h[waves / 2]) / (per_iter * iters * wps);
    // in-kernel wall clock: s_memrealtime counts at 100 MHz
    const double wave_ns = double(hr[waves / 2]) * 10.0, ns_in = wave_ns / (per_iter * iters * wps);
    printf("%-34s %d wave(s)/SIMD: %.2f s_memtime ticks per instruction and SIMD; in-kernel %.3f ns per instruction and SIMD (median wave %.3f ms; ticks/ns %.2f); event %.3f ms = %.3f ns\n",
           names[OP], wps, cyc, ns_in, wave_ns * 1e-6, cyc / ns_in, ms, ns);
    CK(hipFree(out)); CK(hipFree(t));
}

template <int OP>
void all(int iters) {
    for (int wps : {1, 2, 3, 4}) run<OP>(iters, wps);
    if constexpr (OP + 1 < NOPS) all<OP + 1>(iters);
}

int main(int argc, char** argv) {
    all<0>(argc > 1 ? atoi(argv[1]) : 20000);
    return 0;
}
