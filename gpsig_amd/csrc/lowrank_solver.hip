// lowrank_solver.hip -- Nystrom whitening on the device: gpsig_lr_whitening.
//
// Reference: gpsig/low_rank_calculations.py:50-57 (Nystrom_map): W = kappa(S, S) + diag(random jitter), self-adjoint
// eigendecomposition, eigenvalues + jitter, U / sqrt(eigenvalues).  The Gram of the landmarks is formed by the library's own
// base-kernel kernel; the eigendecomposition is rocSOLVER's dsyevd.  rocSOLVER / rocBLAS are opened at first use (dlopen by
// SONAME: a process that has PyTorch-ROCm loaded gets the copies PyTorch loaded, everything else the ROCm installation's),
// so that the evaluation path does not pull two BLAS stacks into every process that never uses low-rank mode.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <mutex>

#include "../../include/gpsig_hip.h"
#include "ctx.hpp"
#include "lowrank_kernels.hpp"

using namespace gpsig;

namespace {

struct SolverApi {
    void *blas = nullptr, *solver = nullptr;
    rocblas_status (*create_handle)(rocblas_handle*) = nullptr;
    rocblas_status (*destroy_handle)(rocblas_handle) = nullptr;
    rocblas_status (*set_stream)(rocblas_handle, hipStream_t) = nullptr;
    rocblas_status (*dsyevd)(rocblas_handle, const rocblas_evect, const rocblas_fill, const rocblas_int, double*, const rocblas_int,
                             double*, double*, rocblas_int*) = nullptr;
    std::string err;
};

SolverApi g_api;
std::once_flag g_api_once;

void* open_first(const char* const* names, std::string* err) {
    for (int i = 0; names[i]; ++i) {
        void* h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (h) return h;
        *err = dlerror();
    }
    return nullptr;
}

void load_api() {
    static const char* const blas_names[] = {"librocblas.so.5", "/opt/rocm/lib/librocblas.so.5", "librocblas.so", "/opt/rocm/lib/librocblas.so", nullptr};
    static const char* const solver_names[] = {"librocsolver.so.0", "/opt/rocm/lib/librocsolver.so.0", "librocsolver.so", "/opt/rocm/lib/librocsolver.so", nullptr};
    g_api.blas = open_first(blas_names, &g_api.err);
    if (!g_api.blas) return;
    g_api.solver = open_first(solver_names, &g_api.err);
    if (!g_api.solver) return;
    g_api.create_handle = reinterpret_cast<decltype(g_api.create_handle)>(dlsym(g_api.blas, "rocblas_create_handle"));
    g_api.destroy_handle = reinterpret_cast<decltype(g_api.destroy_handle)>(dlsym(g_api.blas, "rocblas_destroy_handle"));
    g_api.set_stream = reinterpret_cast<decltype(g_api.set_stream)>(dlsym(g_api.blas, "rocblas_set_stream"));
    g_api.dsyevd = reinterpret_cast<decltype(g_api.dsyevd)>(dlsym(g_api.solver, "rocsolver_dsyevd"));
    if (!g_api.create_handle || !g_api.destroy_handle || !g_api.set_stream || !g_api.dsyevd) g_api.err = "rocBLAS / rocSOLVER symbols not found";
}

int fail_ctx(gpsig_ctx* c, int code, const std::string& msg) {
    c->err = msg;
    return code;
}

// W (c x c, symmetric) += diag(jd)
__global__ void add_diag_kernel(double* __restrict__ W, const double* __restrict__ jd, int c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) W[int64_t(i) * c + i] += jd[i];
}

// Wh[i][j] = U[i][j] / sqrt(ev[j] + jitter), U column-major as dsyevd leaves it (low_rank_calculations.py:56-57, :60)
__global__ void whiten_kernel(const double* __restrict__ Ucm, const double* __restrict__ ev, int c, double jitter, double* __restrict__ Wh) {
    const int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (idx >= int64_t(c) * c) return;
    const int i = int(idx / c), j = int(idx - int64_t(i) * c);
    Wh[idx] = Ucm[int64_t(j) * c + i] / sqrt(ev[j] + jitter);
}

}  // namespace

void gpsig_solver_release(gpsig_ctx* c) {
    if (c->blas_handle && g_api.destroy_handle) (void)g_api.destroy_handle(static_cast<rocblas_handle>(c->blas_handle));
    c->blas_handle = nullptr;
}

extern "C" int gpsig_lr_whitening(gpsig_ctx* c, const gpsig_params* p, const double* landmarks_host, int32_t nc, int32_t d,
                                  const double* jitter_diag_host, double* whitening_host, double* eigenvalues_host) {
    if (!c) return GPSIG_ERR_INVALID;
    if (!p || !landmarks_host || !jitter_diag_host || !whitening_host || nc < 1 || d < 1)
        return fail_ctx(c, GPSIG_ERR_INVALID, "bad whitening request");
    if (hipSetDevice(c->device) != hipSuccess) return fail_ctx(c, GPSIG_ERR_HIP, "hipSetDevice failed");
    std::call_once(g_api_once, load_api);
    if (!g_api.dsyevd) return fail_ctx(c, GPSIG_ERR_HIP, "rocSOLVER is not available: " + g_api.err);
    if (!c->blas_handle) {
        rocblas_handle h = nullptr;
        if (g_api.create_handle(&h) != rocblas_status_success) return fail_ctx(c, GPSIG_ERR_HIP, "rocblas_create_handle failed");
        c->blas_handle = h;
    }
    rocblas_handle h = static_cast<rocblas_handle>(c->blas_handle);
    if (g_api.set_stream(h, c->stream) != rocblas_status_success) return fail_ctx(c, GPSIG_ERR_HIP, "rocblas_set_stream failed");

    auto need = [&](int id, size_t bytes, void** out) { return ctx_ensure(c, id, bytes, out); };
    void *dS, *dW, *dJ, *dEv, *dE, *dInfo, *dWh;
    int rc;
    if ((rc = need(B_LR2, sizeof(double) * size_t(nc) * d + 8, &dS))) return rc;
    if ((rc = need(B_LR4, sizeof(double) * size_t(nc) * nc + 8, &dW))) return rc;
    if ((rc = need(B_LR3, sizeof(double) * size_t(nc) * 3 + 64, &dJ))) return rc;
    if ((rc = need(B_LR5, sizeof(double) * size_t(nc) * nc + 8, &dWh))) return rc;
    dEv = static_cast<double*>(dJ) + nc;
    dE = static_cast<double*>(dJ) + 2 * size_t(nc);
    dInfo = static_cast<double*>(dJ) + 3 * size_t(nc);
#define HC(call)                                                                                       \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return fail_ctx(c, GPSIG_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
    HC(hipMemcpyAsync(dS, landmarks_host, sizeof(double) * size_t(nc) * d, hipMemcpyHostToDevice, c->stream));
    HC(hipMemcpyAsync(dJ, jitter_diag_host, sizeof(double) * size_t(nc), hipMemcpyHostToDevice, c->stream));
    double p0, p1;
    ctx_base_params(p, &p0, &p1);
    const double* spec = nullptr;
    if ((rc = ctx_spectral_table(c, p, &spec))) return rc;
    const int64_t total = int64_t(nc) * nc;
    hipLaunchKernelGGL(base_kernel_matrix_kernel<double>, dim3(unsigned((total + 255) / 256)), dim3(256), 0, c->stream,
                       static_cast<const double*>(dS), static_cast<const double*>(dS), int64_t(nc), int64_t(nc), int(d), int(p->base_kernel), p0, p1,
                       spec, static_cast<double*>(dW));                                                     // low_rank_calculations.py:51
    HC(hipGetLastError());
    hipLaunchKernelGGL(add_diag_kernel, dim3((nc + 255) / 256), dim3(256), 0, c->stream, static_cast<double*>(dW),
                       static_cast<const double*>(dJ), int(nc));                                            // :52
    HC(hipGetLastError());
    if (g_api.dsyevd(h, rocblas_evect_original, rocblas_fill_lower, nc, static_cast<double*>(dW), nc, static_cast<double*>(dEv),
                     static_cast<double*>(dE), static_cast<rocblas_int*>(dInfo)) != rocblas_status_success)    // :55
        return fail_ctx(c, GPSIG_ERR_HIP, "rocsolver_dsyevd failed");
    hipLaunchKernelGGL(whiten_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, c->stream, static_cast<const double*>(dW),
                       static_cast<const double*>(dEv), int(nc), p->jitter, static_cast<double*>(dWh));     // :56-57, :60
    HC(hipGetLastError());
    rocblas_int info = 0;
    HC(hipMemcpyAsync(whitening_host, dWh, sizeof(double) * size_t(total), hipMemcpyDeviceToHost, c->stream));
    if (eigenvalues_host) HC(hipMemcpyAsync(eigenvalues_host, dEv, sizeof(double) * size_t(nc), hipMemcpyDeviceToHost, c->stream));
    HC(hipMemcpyAsync(&info, dInfo, sizeof(info), hipMemcpyDeviceToHost, c->stream));
    HC(hipStreamSynchronize(c->stream));
#undef HC
    if (info != 0) return fail_ctx(c, GPSIG_ERR_HIP, "rocsolver_dsyevd did not converge (info = " + std::to_string(info) + ")");
    return GPSIG_OK;
}
