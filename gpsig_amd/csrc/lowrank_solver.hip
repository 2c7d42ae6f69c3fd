// lowrank_solver.hip -- the eigendecomposition behind gpsig_lr_whitening (api.hip), on rocSOLVER.
//
// Reference: gpsig/low_rank_calculations.py:55 (tf.self_adjoint_eig of the landmark Gram inside Nystrom_map).  rocSOLVER / rocBLAS are opened at first use (dlopen by
// SONAME: a process that has PyTorch-ROCm loaded gets the copies PyTorch loaded, everything else the ROCm installation's),
// so that the evaluation path does not pull two BLAS stacks into every process that never uses low-rank mode.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <mutex>

#include <string>

namespace {

struct SolverApi {
    void *blas = nullptr, *solver = nullptr;
    rocblas_status (*create_handle)(rocblas_handle*) = nullptr;
    rocblas_status (*destroy_handle)(rocblas_handle) = nullptr;
    rocblas_status (*set_stream)(rocblas_handle, hipStream_t) = nullptr;
    rocblas_status (*dsyevd)(rocblas_handle, const rocblas_evect, const rocblas_fill, const rocblas_int, double*, const rocblas_int,
                             double*, double*, rocblas_int*) = nullptr;
    rocblas_status (*dgemm)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int, rocblas_int, const double*, const double*,
                            rocblas_int, const double*, rocblas_int, const double*, double*, rocblas_int) = nullptr;
    rocblas_status (*dgemm_sb)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int, rocblas_int, const double*, const double*,
                               rocblas_int, rocblas_stride, const double*, rocblas_int, rocblas_stride, const double*, double*, rocblas_int, rocblas_stride,
                               rocblas_int) = nullptr;
    rocblas_status (*set_pointer_mode)(rocblas_handle, rocblas_pointer_mode) = nullptr;
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
    // (rocBLAS alone serves solver_dgemm; rocSOLVER is needed by solver_dsyevd only)
    g_api.dgemm = reinterpret_cast<decltype(g_api.dgemm)>(dlsym(g_api.blas, "rocblas_dgemm"));
    g_api.dgemm_sb = reinterpret_cast<decltype(g_api.dgemm_sb)>(dlsym(g_api.blas, "rocblas_dgemm_strided_batched"));
    g_api.set_pointer_mode = reinterpret_cast<decltype(g_api.set_pointer_mode)>(dlsym(g_api.blas, "rocblas_set_pointer_mode"));
    g_api.create_handle = reinterpret_cast<decltype(g_api.create_handle)>(dlsym(g_api.blas, "rocblas_create_handle"));
    g_api.destroy_handle = reinterpret_cast<decltype(g_api.destroy_handle)>(dlsym(g_api.blas, "rocblas_destroy_handle"));
    g_api.set_stream = reinterpret_cast<decltype(g_api.set_stream)>(dlsym(g_api.blas, "rocblas_set_stream"));
    g_api.solver = open_first(solver_names, &g_api.err);
    if (!g_api.solver) return;
    g_api.create_handle = reinterpret_cast<decltype(g_api.create_handle)>(dlsym(g_api.blas, "rocblas_create_handle"));
    g_api.destroy_handle = reinterpret_cast<decltype(g_api.destroy_handle)>(dlsym(g_api.blas, "rocblas_destroy_handle"));
    g_api.set_stream = reinterpret_cast<decltype(g_api.set_stream)>(dlsym(g_api.blas, "rocblas_set_stream"));
    g_api.dsyevd = reinterpret_cast<decltype(g_api.dsyevd)>(dlsym(g_api.solver, "rocsolver_dsyevd"));
    if (!g_api.create_handle || !g_api.destroy_handle || !g_api.set_stream || !g_api.dsyevd) g_api.err = "rocBLAS / rocSOLVER symbols not found";
}

}  // namespace

namespace gpsig {

// Eigendecomposition of the symmetric n x n matrix at A (device, overwritten by the eigenvectors, column-major) on `stream`;
// ev (n), work (n) and info (one int) are device scratch.  *handle_slot caches the rocBLAS handle of the calling context.
// Returns false with *err set when rocSOLVER cannot be opened or refuses the call; convergence is reported through *info.
bool solver_dsyevd(void** handle_slot, hipStream_t stream, int n, double* A, double* ev, double* work, int* info, std::string* err) {
    std::call_once(g_api_once, load_api);
    if (!g_api.dsyevd) { *err = "rocSOLVER is not available: " + g_api.err; return false; }
    if (!*handle_slot) {
        rocblas_handle h = nullptr;
        if (g_api.create_handle(&h) != rocblas_status_success) { *err = "rocblas_create_handle failed"; return false; }
        *handle_slot = h;
    }
    rocblas_handle h = static_cast<rocblas_handle>(*handle_slot);
    if (g_api.set_stream(h, stream) != rocblas_status_success) { *err = "rocblas_set_stream failed"; return false; }
    if (g_api.dsyevd(h, rocblas_evect_original, rocblas_fill_lower, n, A, n, ev, work, info) != rocblas_status_success) {
        *err = "rocsolver_dsyevd failed";
        return false;
    }
    return true;
}

// C (m x n, ldc) = alpha op(A) op(B) + beta C in rocBLAS's column-major convention, on `stream`; alpha / beta by value.  The gradient of
// SignatureLinear's levels with respect to the level features is such a product per level (grad_api.hip, sig_features_grad): a plain
// library GEMM, which is what north_star assigns to the BLAS.
bool solver_dgemm(void** handle_slot, hipStream_t stream, bool transA, bool transB, int m, int n, int k, double alpha, const double* A, int lda,
                  const double* B, int ldb, double beta, double* C, int ldc, std::string* err) {
    std::call_once(g_api_once, load_api);
    if (!g_api.dgemm || !g_api.create_handle || !g_api.set_stream) { *err = "rocBLAS is not available: " + g_api.err; return false; }
    if (!*handle_slot) {
        rocblas_handle h = nullptr;
        if (g_api.create_handle(&h) != rocblas_status_success) { *err = "rocblas_create_handle failed"; return false; }
        *handle_slot = h;
    }
    rocblas_handle h = static_cast<rocblas_handle>(*handle_slot);
    if (g_api.set_stream(h, stream) != rocblas_status_success) { *err = "rocblas_set_stream failed"; return false; }
    if (g_api.set_pointer_mode) (void)g_api.set_pointer_mode(h, rocblas_pointer_mode_host);
    const rocblas_status st = g_api.dgemm(h, transA ? rocblas_operation_transpose : rocblas_operation_none,
                                          transB ? rocblas_operation_transpose : rocblas_operation_none, m, n, k, &alpha, A, lda, B, ldb, &beta, C, ldc);
    if (st != rocblas_status_success) { *err = "rocblas_dgemm failed (status " + std::to_string(int(st)) + ")"; return false; }
    return true;
}

// The same for `batch` problems at constant strides (rocblas_dgemm_strided_batched): the per-sequence argument lattices of the wide route (wide_api.hip).
bool solver_dgemm_batched(void** handle_slot, hipStream_t stream, bool transA, bool transB, int m, int n, int k, double alpha, const double* A, int lda,
                          int64_t sa, const double* B, int ldb, int64_t sb, double beta, double* C, int ldc, int64_t sc, int batch, std::string* err) {
    std::call_once(g_api_once, load_api);
    if (!g_api.dgemm_sb || !g_api.create_handle || !g_api.set_stream) { *err = "rocBLAS is not available: " + g_api.err; return false; }
    if (!*handle_slot) {
        rocblas_handle h = nullptr;
        if (g_api.create_handle(&h) != rocblas_status_success) { *err = "rocblas_create_handle failed"; return false; }
        *handle_slot = h;
    }
    rocblas_handle h = static_cast<rocblas_handle>(*handle_slot);
    if (g_api.set_stream(h, stream) != rocblas_status_success) { *err = "rocblas_set_stream failed"; return false; }
    if (g_api.set_pointer_mode) (void)g_api.set_pointer_mode(h, rocblas_pointer_mode_host);
    const rocblas_status st = g_api.dgemm_sb(h, transA ? rocblas_operation_transpose : rocblas_operation_none,
                                             transB ? rocblas_operation_transpose : rocblas_operation_none, m, n, k, &alpha, A, lda, sa, B, ldb, sb, &beta, C, ldc, sc,
                                             batch);
    if (st != rocblas_status_success) { *err = "rocblas_dgemm_strided_batched failed (status " + std::to_string(int(st)) + ")"; return false; }
    return true;
}

void solver_release(void* handle) {
    if (handle && g_api.destroy_handle) (void)g_api.destroy_handle(static_cast<rocblas_handle>(handle));
}

}  // namespace gpsig
