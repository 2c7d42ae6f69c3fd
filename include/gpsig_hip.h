/* gpsig_hip.h -- C ABI of libgpsig_hip.so: the MI355X (gfx950) signature-kernel evaluation path.
 *
 * The reference (tgcsaba/GPSig) is pure Python on TensorFlow 1.15 / GPflow 1.5.1 and has no native
 * boundary of its own; the entry points below are what a GPflow-side binding for this path would
 * call instead of building the TF graph.  Each one names the reference method it replaces
 * (paths relative to the reference checkout).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - All arrays are row-major and contiguous.  Sequences are passed exactly as the reference's
 *     numpy-facing wrappers take them: X is (N, L*d) viewed as (N, L, d) (gpsig/kernels.py:417-418),
 *     Z is (lt, T, d) or (lt, T, 2, d) with lt = M(M+1)/2 (gpsig/inducing_variables.py:28-46).
 *   - "levels" outputs have a leading axis of num_levels+1 (level 0 == 1, gpsig/signature_algs.py:20).
 *   - Data pointers are host or device pointers according to gpsig_set_pointer_mode(); the
 *     hyper-parameter arrays inside gpsig_params are always HOST pointers.
 *   - Every function returns GPSIG_OK (0) or a negative error code; gpsig_last_error() gives text.
 *   - Caller owns all inputs and outputs.  Scratch memory is owned by the ctx.  One ctx per
 *     (device, stream); calls on different ctxs may run concurrently; a ctx is not re-entrant.
 *   - In device-pointer mode calls are asynchronous on the ctx stream; in host-pointer mode they
 *     return after the result has been copied back.
 *   - There is no CPU fallback anywhere behind this ABI.
 */
#ifndef GPSIG_HIP_H
#define GPSIG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPSIG_ABI_VERSION 1

enum gpsig_status {
    GPSIG_OK = 0,
    GPSIG_ERR_INVALID = -1,      /* bad argument (the Python shim raises ValueError) */
    GPSIG_ERR_UNSUPPORTED = -2,  /* valid in the reference, not built here yet (NotImplementedError) */
    GPSIG_ERR_HIP = -3,          /* HIP runtime failure (RuntimeError) */
    GPSIG_ERR_NOMEM = -4
};

/* static (state-space) kernels, gpsig/kernels.py:786-993 */
enum gpsig_base_kernel {
    GPSIG_BASE_LINEAR = 0,    /* SignatureLinear    _lin       :799-806 */
    GPSIG_BASE_RBF = 1,       /* SignatureRBF/Gauss _rbf       :862-864 */
    GPSIG_BASE_COSINE = 2,    /* SignatureCosine    _cos       :820-828 */
    GPSIG_BASE_POLY = 3,      /* SignaturePoly      _poly      :844-848  base_params = {gamma, degree} */
    GPSIG_BASE_MIX = 4,       /* SignatureMix       _mix       :881-892  base_params = {mixing} */
    GPSIG_BASE_MATERN12 = 5,  /* SignatureMatern12/Laplace/Exponential :955-958 */
    GPSIG_BASE_MATERN32 = 6,  /* SignatureMatern32  :974-977 */
    GPSIG_BASE_MATERN52 = 7,  /* SignatureMatern52  :991-993 */
    GPSIG_BASE_SPECTRAL = 8   /* SignatureSpectral  _spectral  :921-942  base_params = {Q, family (0 rbf, 1 exp, 2 mixed)};
                                 base_table = alpha[Q], omega[Q][d], gamma[Q][d]; needs lengthscales == NULL, num_lags == 0
                                 (as the reference: :907, and its gamma clashes with the lag weights of :82) */
};

enum gpsig_dtype { GPSIG_F64 = 0, GPSIG_F32 = 1 };
enum gpsig_pointer_mode { GPSIG_PTR_HOST = 0, GPSIG_PTR_DEVICE = 1 };

typedef struct gpsig_ctx gpsig_ctx;

/* Constructor state of gpsig.kernels.SignatureKernel (gpsig/kernels.py:18-88), constrained values. */
typedef struct gpsig_params {
    int32_t base_kernel;     /* enum gpsig_base_kernel */
    int32_t dtype;           /* enum gpsig_dtype: element type of every data pointer */
    int32_t num_features;    /* d: state-space dimension of ONE lag copy (kernels.py:54) */
    int32_t num_levels;      /* M (kernels.py:55) */
    int32_t order;           /* 1..M, already clamped as kernels.py:57 does */
    int32_t difference;      /* kernels.py:63 */
    int32_t normalization;   /* kernels.py:62 */
    int32_t num_lags;        /* kernels.py:70-82 */
    double sigma;            /* kernels.py:66 */
    double jitter;           /* gpflow.settings.jitter, 1e-6 (kernels.py:431,463,578) */
    double base_params[4];
    const double* variances;     /* host, M+1 entries (kernels.py:65) */
    const double* lengthscales;  /* host, d entries, or NULL = no scaling (kernels.py:84-88) */
    const double* lags;          /* host, num_lags entries (kernels.py:79), NULL if num_lags == 0 */
    const double* gamma;         /* host, num_lags+1 entries (kernels.py:80-82), NULL if num_lags == 0 */
    const double* base_table;    /* host, extra parameters of the base kernel (GPSIG_BASE_SPECTRAL), else NULL */
    int64_t base_table_len;      /* number of doubles in base_table */
} gpsig_params;

/* ---- context ----------------------------------------------------------------------------- */
int gpsig_abi_version(void);
/* stream: a hipStream_t cast to void*, or NULL for the default stream */
int gpsig_ctx_create(int device, void* stream, gpsig_ctx** out);
void gpsig_ctx_destroy(gpsig_ctx* ctx);
const char* gpsig_last_error(gpsig_ctx* ctx); /* ctx may be NULL: error of the last failed create */
int gpsig_set_pointer_mode(gpsig_ctx* ctx, int mode);
int gpsig_sync(gpsig_ctx* ctx);
/* Restrict the following K / Kzx calls to shard `index` of `count` (independent pair blocks, no
 * exchange).  Entries outside the shard are left untouched in the output.  (0, 1) = everything. */
int gpsig_set_shard(gpsig_ctx* ctx, int index, int count);
/* Tuning / debugging knobs; unknown names return GPSIG_ERR_INVALID.
 *   "glds"        1: stage x-side records with LDS-DMA (global_load_lds), 0: load + ds_write
 *   "exact"       1: allow the kernels specialised on num_levels, 0: generic kernels only
 *   "pk2"         float32 kernels with two y sequences per pair group on the packed v_pk_* instructions (seq_pk2_kernel.hpp):
 *                 1 (default) wherever they are the faster ones, 2 always where built, 0 never
 *   "f32_waves"   their wavefronts per workgroup on one LDS ring of x records: 4, 1, or 0 = by launch size
 *   "f32_pack"    their y sequences per pair group: 2 (packed) or 1 (the same code on scalar instructions; for A/B runs)
 *   "keep_reset"  1 (default): the first-order pair kernels clear a lane's accumulators at a pair boundary through the
 *                 recursion's own FMAs (acc = acc * keep + inc with keep = 0 for one step), 0: by an explicit reset in the
 *                 boundary block (round 1; for A/B runs)
 *   "max_run"     >0: x-side run length per task, 0: automatic
 *   "tensor_lanes" tensor-vs-sequence kernel: 1 one lane per tensor, 0 one lane per sequence, -1 automatic
 *   "grad_scratch_mb" lattice scratch of one gradient / fallback launch in MiB (default 4096)
 *   "grad_impl"   gradient kernels: 0 planner's choice, 1 one pair per thread with the lattice in HBM scratch, 2 one pair per
 *                 thread scratch-free (tensor vs sequence), 3 wavefront kernel with the lattice in HBM scratch, 4 scratch-free wavefront
 *                 kernels wherever they are built (rounds 3-4's choice: for the point kernels both sweeps in one wavefront, dL/d(increments)
 *                 through HBM to a contraction kernel per side).  0 additionally picks the fused reverse kernel of round 5 (an evaluator and a
 *                 sweeper wavefront per four sequence pairs, both sides contracted on chip) where it is built: RBF and the Matern families on
 *                 points with differences, order 1, at most 256 points on one side, at most 8 columns of state space, 2 to 6 levels
 *   "matern_fast" float64 sequence Grams of SignatureMatern12 / 32 / 52: 1 (default) compile-time instances on prescaled records where the exact
 *                 shapes are built (distances from coordinate differences, table exp), 0 the run-time-kind instances with the library sqrt / exp
 *   "grad_stash_mb" gpsig_seq_gram_levels_stash keeps at most this many MiB for the backward call (default 4096; 0: never)
 *   "tvs_grad_tile" reverse pass of the tensor-vs-sequence chains: 1 (default) the tile kernel (all levels in one reverse sweep per
 *                 sequence, d/dx summed in LDS, no atomics) where it is built (order 1, at most 8 columns, at most 6 levels), 0 the round-1 kernels
 *   "pinned_staging" host-pointer mode: 1 (default) transfers of 2 MiB and more run in 16 MiB chunks through two pinned buffers of the
 *                 context, the DMA of one chunk overlapping a threaded host copy of the previous one; 0 plain hipMemcpy on the caller's
 *                 (pageable) memory
 *   "sig_features" SignatureLinear and SignatureCosine, any order: the Gram as ONE contraction of explicit level features on the float64 matrix cores
 *                 (sig_feat_kernel.hpp: K_m(x, y) = <Phi_m(x), Phi_m(y)>, d^m numbers per level): -1 (default) where that costs fewer
 *                 flops than the lattice sweep and the feature matrices fit, 0 never, 1 wherever it is built (d <= 32 columns after lags, d^M <= 32768); float32 calls are
 *                 computed in float64 on this route (inputs widened, result rounded)
 *   "sig_features_keep" 1: the feature matrix built by the next such evaluation is kept and reused by the evaluations that follow with the
 *                 same sequence pointer, shape and parameters -- the row-block calls of one decomposed Gram (gpsig_kernel_K_symm_rows*);
 *                 the caller must not write the sequences until it sets the option back to 0 (which also drops the matrix's validity)
 *   "sig_gemm_dma" that contraction's operand slabs by LDS-DMA into an XOR-swizzled image, fragments prefetched across the barrier
 *                 (1, default) or staged through registers (0); bit-identical results
 *   "sig_features_grad" gpsig_seq_gram_levels_grad / gpsig_seq_diag_levels_grad of SignatureLinear / SignatureCosine (every order) through the same feature space
 *                 (features, one rocBLAS dgemm per level and side, a reverse sweep per sequence: csrc/sig_feat_grad_api.hip): -1 (default)
 *                 where a time model says it is cheaper than the pair kernels' reverse pass, 0 never, 1 wherever it is built
 *   "sig_graded"  that contraction's depth pieces: 1 (default) the last of the equal pieces is cut into finer ones of halving size where a
 *                 launch runs for several rounds of workgroups (they fill the slots that fall free at the end: 12 -> 11.5 rounds' time
 *                 at BASELINE configs[1]), 0 equal pieces throughout (round 3).  The split depends on the Gram's total size and depth
 *                 alone, so row blocks still reassemble the one-call Gram bit for bit
 *   "lr_jacobi"   gpsig_lr_draw: 1 (default) the landmark Gram's eigendecomposition by the one-workgroup Jacobi kernel (c <= 64), 0 rocSOLVER
 *   "tvs_zreg"    tensor-lane gradient: components in registers (1) or LDS (0), -1 automatic
 *   "tvs_features" K_tens_vs_seq / the Kzx of K_tens_n_seq_covs (level sum) of SignatureLinear / SignatureCosine as ONE product of the tensors'
 *                 rank-one level features and the sequences' level features (rocBLAS dgemm; round 4): -1 (default) where a time model
 *                 prefers it to the tile kernel, 0 never, 1 wherever built (float64, 2 <= num_levels <= 8, not inside a graph capture)
 *   "tvs_tile"    tensor-vs-sequence tile kernel (levels split over the waves of a workgroup, coalesced result tiles):
 *                 -1 wherever it is built and there are at least 32 tensors, 0 never, 1 also for fewer tensors
 *   "tvs_tile_nw" its waves per workgroup (1 or 2), 0 automatic
 *   "tens_tile"   tensor-vs-tensor kernel (Kzz): 1 (default) 16 x 16 tiles with the tensors staged in LDS, 0 one thread per entry
 *                 gathering its components from HBM (round 1)
 *   "spectral_wave" SignatureSpectral's sequence kernels: 1 (default) the wavefront kernels with the family at compile time where they
 *                 are built (float64, first order, differences, d <= 16), 0 one pair per thread (round 1)
 *   "diag_own"    diagonal pass of the pair kernel (level diagonals for normalisation, Kdiag): 1 (default) every pair group of a
 *                 wavefront sweeps its own sequence, 0 all groups sweep the same 64/G sequences and emit one pair each (round 1)
 *   "lr_gemm"     low-rank Gram products on the fp64 matrix cores: 1 (default) 128 x 128 tiles with k-slabs staged through LDS,
 *                 0 operand fragments read straight from L2 (round 1)
 *   "lr_fused"    low-rank sequence features (gpsig_lr_seq_features): 1 (default) one fused kernel, a workgroup per sequence with
 *                 the (width, length) intermediates in LDS, wherever they fit -- two arrays where a wavefront can hold its output
 *                 columns in registers (at most 64 time steps, components and rank bound), three otherwise; 2 always the three-array
 *                 form; 0 one elementwise kernel per reference op
 *   "wide"        (round 6) the wide-state-space route (csrc/wide_api.hip): kernel arguments by rocBLAS dgemm on augmented rows, fused
 *                 map / difference / recursion kernels, behind the levels, weighted-sum, fused-evaluation entry points and their gradients.
 *                 -1 (default) where the exact-shape kernels are not built or lose: Kzx beyond 8 columns, the sequence lattices beyond
 *                 32 columns / their register sides (reverse pass: beyond 8 columns where no fused kernel is built), Kzz beyond 12 --
 *                 the reference's own run settings, benchmarks/run_gpsig_benchmarks.py:32; 0 never; 1 wherever built (float64,
 *                 RBF and the Matern families, at most 8 levels, lattices of at most 512 columns; order 1, and at order > 1 the
 *                 tensor-vs-sequence chains (orders <= 4) and the sequence lattices' reverse pass (<= 5 levels)).  Not taken inside
 *                 a graph capture.
 *   "wide_chunk_mb"  its argument chunk in HBM (0: "grad_scratch_mb", 4 GB by default: a latency-bound launch split in two takes twice as long)
 *   "wide_contract"  the two contractions of a reverse pass with the adjoint array (rows of at most 32 augmented columns) as one hand-written MFMA pass:
 *                 1 (default) part tiles of 16 rows in LDS with the second operands in registers (rows of at most 16 columns; wider: the first form),
 *                 3 part tiles of 32 rows, 2 the first form (whole 64 x 64 tiles and both operand blocks in LDS), 0 two rocBLAS dgemms (for A/B runs)
 *   "wide_lat_waves"  wavefronts per sequence lattice on the wide route: -1 (default) eight for a launch of at most 128 lattices of more than 256
 *                 columns, one otherwise; 0 always one; 1 two / four / eight wherever a lane would hold that many columns
 *   "wide_sym_fold"  1 (default): the reverse pass of a symmetric Gram on the wide route runs over the pairs i <= j with the upstream gradient
 *                 folded onto them (G + G^T above the diagonal); 0: over all ordered pairs (for A/B runs)
 *   "wide_few_cols"  widest state space (default 8) at which the tensor-vs-sequence FORWARD pass of at most 256 sequences against tensors with increments
 *                 takes the wide chains (SignatureRBF; the reverse tile kernel continues from their totals); "ho_g32": -1 (default) the higher-order
 *                 sweeps of 33 .. 64 lattice columns with two columns per lane at order >= 3, four otherwise; 0 always four; 1 always two
 *   "wide_o1_sweeps"  first-order reverse pass of sequence lattices of at most 64 columns on the wide route: 1 (default) four lattices per wavefront from a dM
 *                 lattice (seq_grad_wave_o1_kernel) for launches of 1,024 lattices or more, 2 wherever the shape fits, 0 the lattice kernels (one per wavefront)
 *   "tvs_grad_matern" 1 (default): the Matern families in the tensor-vs-sequence reverse tile kernel as a compile-time kind; 0: the run-time family (A/B runs)
 *   order > 1 and "grad_impl": the sequence recursion's reverse pass runs as two sweeps of a wavefront per pair (csrc/grad_wave_ho_kernel.hpp;
 *                 <= 5 levels, min(order, levels) <= 4, lattices of <= 512 columns): 0 scratch-free where the row totals fit LDS, 3 with
 *                 the prefixes in an HBM slot per pair group, any other value the lattice operations of rounds 2-5 (tests' A/B reference) */
int gpsig_set_option(gpsig_ctx* ctx, const char* name, int value);
/* HIP-event timing of the dominant kernel (the pair recursion) launched by the calls since the last
 * reset, measured on the ctx stream: total milliseconds and number of launches (the first 4096 timed
 * launches after a reset; none inside a graph capture). */
int gpsig_timing_reset(gpsig_ctx* ctx);
int gpsig_timing_get(gpsig_ctx* ctx, double* kernel_ms, int64_t* launches, int64_t* pairs);
/* What the timed launches were: *kernel = "sig_gram_dma_kernel" (or "sig_gram_kernel", its register-staged form) when they were the matrix-core contraction of the explicit signature
 * features (option "sig_features"), NULL for the pair recursion / chain kernels; *flops = the floating-point operations those
 * launches executed on the matrix cores (whole tiles, padded depth), 0 otherwise. */
int gpsig_timing_info(gpsig_ctx* ctx, const char** kernel, double* flops);
/* Effective shader clock while the calls that follow run (diagnostics for benchmarks; no reference analogue): a single sleeping
 * wavefront per XCD (eight one-wave workgroups) on a stream of its own takes up to `samples` readings of s_memtime (shader cycles) against s_memrealtime (100 MHz),
 * duration_ms / (samples - 1) apart; _read tells it to take a last reading and leave, waits for it, and returns the mean /
 * smallest / largest clock between consecutive readings in GHz and the time the readings span (it leaves by itself after 1.25 x
 * duration_ms).  Read it before any device-wide synchronisation, which would wait for the probe.  float64-heavy kernels run
 * this chip at 2.0-2.2 GHz instead of 2.4 (DVFS), differently from box to box: a benchmark line should carry the value. */
int gpsig_clock_probe_start(gpsig_ctx* ctx, double duration_ms, int32_t samples);
int gpsig_clock_probe_read(gpsig_ctx* ctx, double* ghz_mean, double* ghz_min, double* ghz_max, double* covered_ms);
/* The last read's mean clock per sampled wavefront and the XCD each sat on (HW_REG_XCC_ID); ghz_mean above is their average. */
int gpsig_clock_probe_xcds(gpsig_ctx* ctx, double* ghz, int32_t* xcc, int32_t cap, int32_t* n);

/* ---- HIP graphs for launch-bound evaluations (no reference analogue) -------------------------------------
 * An end-to-end evaluation is 5-15 short kernels.  The calls made between gpsig_graph_begin and gpsig_graph_end are
 * recorded from the ctx stream instead of executed; gpsig_graph_launch replays them with one launch, reading and writing
 * the same device buffers (refresh the inputs in place).  What it saves is host time per evaluation; the device time of a
 * small evaluation is the serial sweep of one pair's lattice and is the same either way (profiles/r01_graph_latency.txt).
 * Conditions, all checked: device-pointer mode; a context created on a stream of its own (not the default stream);
 * every recorded call was made once before with the same shapes and
 * hyper-parameters on this ctx (so that scratch buffers, task lists and level weights are in place -- a call that would
 * allocate, upload or wait fails with GPSIG_ERR_INVALID and the capture ends); a graph is replayable until a later call
 * moves a scratch buffer (gpsig_graph_launch then returns GPSIG_ERR_INVALID: capture again). */
typedef struct gpsig_graph gpsig_graph;
int gpsig_graph_begin(gpsig_ctx* ctx);
int gpsig_graph_end(gpsig_ctx* ctx, gpsig_graph** out);
int gpsig_graph_launch(gpsig_ctx* ctx, gpsig_graph* graph);
void gpsig_graph_destroy(gpsig_graph* graph);

/* ---- unnormalised level tensors (the signature_algs.py layer) ----------------------------- */
/* SignatureKernel._K_seq (kernels.py:208-237) -> signature_kern_first_order / _higher_order
 * (signature_algs.py:8-74) on the base-kernel tensor of (X, X2).  X2 == NULL: symmetric.
 * Inputs are used as given (no lengthscale division).  out: (M+1, N1, N2). */
int gpsig_seq_gram_levels(gpsig_ctx* ctx, const gpsig_params* p, const void* X, const void* X2,
                          int64_t N1, int64_t N2, int32_t L1, int32_t L2, void* out);
/* SignatureKernel._K_seq_diag (kernels.py:188-205).  out: (M+1, N). */
int gpsig_seq_diag_levels(gpsig_ctx* ctx, const gpsig_params* p, const void* X, int64_t N, int32_t L, void* out);
/* SignatureKernel._K_tens (kernels.py:263-283) -> tensor_kern (signature_algs.py:76-99).  out: (M+1, T, T). */
int gpsig_tens_gram_levels(gpsig_ctx* ctx, const gpsig_params* p, const void* Z, int64_t T, int32_t increments, void* out);
/* SignatureKernel._K_tens_vs_seq (kernels.py:313-340) -> signature_kern_tens_vs_seq_first_order /
 * _higher_order (signature_algs.py:101-160).  out: (M+1, T, N). */
int gpsig_tens_vs_seq_levels(gpsig_ctx* ctx, const gpsig_params* p, const void* Z, const void* X,
                             int64_t T, int64_t N, int32_t L, int32_t increments, void* out);

/* Weighted level sum of the tensor-vs-sequence levels,  out[t][n] = sum_m fac[n][m] * _K_tens_vs_seq(Z, X)[m][t][n]:  what
 * K_tens_vs_seq (kernels.py:572-588) and the Kzx of K_tens_n_seq_covs (:638 / :660, :667) are once the per-sequence factors
 * fac[n][m] = sigma variances[m] / sqrt(diag_m(x_n) + jitter) are known -- in one pass, the level sum taken in the kernel's
 * epilogue, without the (M+1, T, N) level array ever reaching memory.  It is the form a training step uses (the factors are
 * differentiable inputs; gpsig_tens_vs_seq_weighted_grad below).  Inputs are used as given, like the level primitives.
 * fac: (N, M+1), out: (T, N), element type p->dtype.
 * aux (optional; float64, device-pointer mode): gpsig_tens_vs_seq_aux_elems() doubles that receive the totals of every chain of
 * every (tensor, sequence) pair when the tile kernel evaluates the call (*aux_written = 1); handed to _weighted_grad they save its
 * forward sweep.  NULL, or *aux_written == 0 afterwards: the reverse pass rebuilds them itself. */
int gpsig_tens_vs_seq_weighted(gpsig_ctx* ctx, const gpsig_params* p, const void* Z, const void* X, int64_t T, int64_t N,
                               int32_t L, int32_t increments, const void* fac, void* out, void* aux, int32_t* aux_written);
int64_t gpsig_tens_vs_seq_aux_elems(const gpsig_params* p, int64_t T, int64_t N);

/* ---- end-to-end kernel evaluations (scaling, lags, normalisation, sigma*variances, level sum) */
/* SignatureKernel.K (kernels.py:401-476).  out: (N1, N2), or (M+1, N1, N2) if return_levels. */
int gpsig_kernel_K(gpsig_ctx* ctx, const gpsig_params* p, const void* X, const void* X2,
                   int64_t N1, int64_t N2, int32_t L1, int32_t L2, int32_t return_levels, void* out);
/* Row-block form of the symmetric K(X) for multi-GPU runs (no reference analogue: the reference is single
 * device).  Every unordered pair {i, j} of the N sequences is owned by exactly one row: row j owns the
 * columns i with (j - i) mod N in [0, N/2] (ties at N/2, N even, go to the smaller index as column).  A rank
 * computes the owned entries of rows [row_begin, row_end) -- row_begin a multiple of 4 -- into `out_rows`,
 * a (row_end - row_begin, N) block; entries it does not own are left untouched.  The blocks of all ranks,
 * stacked, are turned into the full symmetric matrix by gpsig_symmetrize_owned_rows. */
int gpsig_kernel_K_symm_rows(gpsig_ctx* ctx, const gpsig_params* p, const void* X, int64_t N, int32_t L,
                             int64_t row_begin, int64_t row_end, void* out_rows);
/* out[r][c] = half[r][c] if row r owns column c, else half[c][r].  half, out: (N, N), distinct buffers. */
int gpsig_symmetrize_owned_rows(gpsig_ctx* ctx, int32_t dtype, const void* half, int64_t N, void* out);
/* The same two steps with COMPACT row blocks: row j's N/2+1 owned columns j-N/2 .. j (mod N) are stored side by side,
 * out_rows[(j - row_begin) * (N/2+1) + N/2 - ((j-i) mod N)] -- half the bytes to move between GPUs; the one slot per row of an
 * even N that belongs to the other row of its pair (the tie at distance N/2) is left untouched.  half: the stacked blocks,
 * (N, N/2+1); out: (N, N). */
int gpsig_kernel_K_symm_rows_compact(gpsig_ctx* ctx, const gpsig_params* p, const void* X, int64_t N, int32_t L,
                                     int64_t row_begin, int64_t row_end, void* out_rows);
int gpsig_symmetrize_compact_rows(gpsig_ctx* ctx, int32_t dtype, const void* half, int64_t N, void* out);
/* SignatureKernel.Kdiag (kernels.py:479-510).  out: (N,) or (M+1, N). */
int gpsig_kernel_Kdiag(gpsig_ctx* ctx, const gpsig_params* p, const void* X, int64_t N, int32_t L,
                       int32_t return_levels, void* out);
/* SignatureKernel.K_tens (kernels.py:513-536).  out: (T, T) or (M+1, T, T). */
int gpsig_kernel_K_tens(gpsig_ctx* ctx, const gpsig_params* p, const void* Z, int64_t T, int32_t increments,
                        int32_t return_levels, void* out);
/* SignatureKernel.K_tens_vs_seq (kernels.py:539-588).  out: (T, N) or (M+1, T, N). */
int gpsig_kernel_K_tens_vs_seq(gpsig_ctx* ctx, const gpsig_params* p, const void* Z, const void* X,
                               int64_t T, int64_t N, int32_t L, int32_t increments, int32_t return_levels, void* out);
/* SignatureKernel.K_tens_n_seq_covs (kernels.py:591-671).  Kxx is (N,) / (M+1, N) unless full_X_cov. */
int gpsig_kernel_K_tens_n_seq_covs(gpsig_ctx* ctx, const gpsig_params* p, const void* Z, const void* X,
                                   int64_t T, int64_t N, int32_t L, int32_t increments, int32_t full_X_cov,
                                   int32_t return_levels, void* Kzz, void* Kzx, void* Kxx);
/* SignatureKernel.K_seq_n_seq_covs (kernels.py:674-761): X = inducing sequences, X2 = data. */
int gpsig_kernel_K_seq_n_seq_covs(gpsig_ctx* ctx, const gpsig_params* p, const void* X, const void* X2,
                                  int64_t N1, int64_t N2, int32_t L1, int32_t L2, int32_t full_X2_cov,
                                  int32_t return_levels, void* Kxx, void* Kxx2, void* Kx2x2);

/* ---- low-rank mode (low_rank=True): gpsig/low_rank_calculations.py, gpsig/signature_algs.py:162-222,
 * gpsig/kernels.py:239-311, :424-426, :442-458, :499-501, :525-527, :560-574.  float64 only.
 *
 * The reference draws landmarks and random projections with TensorFlow's RNG inside the graph; here they are
 * drawn by the caller and passed in (all HOST pointers), which makes the device code deterministic:
 *   landmarks  (c, d') scaled points, whitening (c, c) = U / sqrt(S + jitter) of the landmark Gram
 *              (low_rank_calculations.py:50-60);
 *   sketches   one per level 2..M: the projection of low_rank_calculations.py:104-193 stored by output column,
 *              out[j] = sum_{e in [colptr[j], colptr[j+1])} val[e] * A[i1[e]] * B[i2[e]].
 * Factor matrices Phi are (rows, F), F = 1 + c + (M-1) r, level blocks [1 | c | r | ... | r].
 * Host-side objects are uploaded once and recognised on later calls by a 64-bit FNV-1a hash of their dimensions and contents (an
 * evaluation hashes ~100 KB instead of uploading it); the parameter tables of the spectral kernel, wide lengthscales and the level
 * offsets are compared in full.  A recorded graph (gpsig_graph_*) reads these device copies: a later eager call on the same ctx
 * with other objects rewrites them in place, and replays of the earlier recording are refused. */
typedef struct gpsig_sketch {
    int32_t k1, k2, r, nnz;
    const int32_t* colptr;   /* r + 1 */
    const int32_t* i1;       /* nnz, < k1 */
    const int32_t* i2;       /* nnz, < k2 */
    const double* val;       /* nnz */
} gpsig_sketch;
typedef struct gpsig_lr_state gpsig_lr_state;   /* the same objects drawn and kept on the device: gpsig_lr_draw below */
typedef struct gpsig_lowrank {
    int32_t num_components;  /* c */
    int32_t rank_bound;      /* r */
    int32_t num_sketches;    /* M - 1 */
    const double* landmarks;
    const double* whitening;
    const gpsig_sketch* sketches;
    const gpsig_lr_state* device_state;   /* non-NULL: the three arrays above are ignored */
} gpsig_lowrank;
/* The random objects of one low-rank evaluation drawn ON THE DEVICE, as the reference draws them inside its graph at every
 * evaluation (low_rank_calculations.py:12-20, :47-57, :92-101, :104-127, :139-193): c landmark rows uniformly without replacement
 * among the scaled components of Z (or NULL), the scaled observations of X and of X2 (or NULL) -- in this order --, the jitter
 * diagonal, the whitening of the landmark Gram (a one-workgroup Jacobi eigendecomposition for c <= 64, rocSOLVER beyond), and one
 * projection per level 2..M; sparsity 0 'sqrt', 1 'log', 2 'lin'.  A counter-based generator (Philox-4x32-10) keyed by `seed`: the same
 * seed gives the same objects.  Device pointers; everything is queued on the ctx stream, nothing waits for the host.  *state: NULL to
 * create, or a state of this context to draw into again.  gpsig_lr_state_sizes / _export copy what was drawn to the host (for the
 * CPU restatement of the same evaluation); they wait for the stream. */
int gpsig_lr_draw(gpsig_ctx* ctx, const gpsig_params* p, int32_t num_components, int32_t rank_bound, int32_t sparsity, uint64_t seed,
                  const void* X, int64_t N, int32_t L, const void* X2, int64_t N2, int32_t L2, const void* Z, int64_t T,
                  int32_t increments, gpsig_lr_state** state);
void gpsig_lr_state_destroy(gpsig_lr_state* state);
int gpsig_lr_state_sizes(gpsig_ctx* ctx, const gpsig_lr_state* state, int32_t* sizes /* 5: c, d', r, projections, Jacobi sweeps */, int32_t* nnz);
int gpsig_lr_state_export(gpsig_ctx* ctx, const gpsig_lr_state* state, double* landmarks, double* jitter_diag, double* whitening,
                          double* eigenvalues, const gpsig_sketch* sketches);
/* scaled / lagged observations number idx[0..R) (flat index n*L + t) of X -> out (R, d') on the HOST (landmark candidates) */
int gpsig_lr_gather_points(gpsig_ctx* ctx, const gpsig_params* p, const void* X, int64_t N, int32_t L, const int64_t* idx,
                           int64_t R, double* out_host);
/* kappa(A, B) of already scaled points, all on the HOST: A (na, d'), B (nb, d') -> out (na, nb) */
int gpsig_base_kernel_matrix(gpsig_ctx* ctx, const gpsig_params* p, const double* A_host, const double* B_host, int64_t na,
                             int64_t nb, int32_t d, double* out_host);
/* Nystrom whitening (low_rank_calculations.py:50-57, :60) of c landmarks S (c, d') -- already scaled points, HOST --:
 * W = kappa(S, S) + diag(jitter_diag) on the device, rocSOLVER dsyevd, whitening (c, c) = U / sqrt(eigenvalues + p->jitter) back
 * on the HOST (the layout gpsig_lowrank.whitening takes).  eigenvalues_host: (c) ascending, or NULL.  The reference draws
 * jitter_diag as settings.jitter * uniform(0, 1) per landmark (:52); it is an argument here so that the caller owns the RNG. */
int gpsig_lr_whitening(gpsig_ctx* ctx, const gpsig_params* p, const double* landmarks_host, int32_t c, int32_t d,
                       const double* jitter_diag_host, double* whitening_host, double* eigenvalues_host);
/* The same feature map for the TRAINING path (round 4): when the reference trains in low-rank mode the landmarks are gathered from the
 * scaled inputs and their Gram is decomposed inside the differentiated graph (low_rank_calculations.py:47-60), so landmarks S (c, d) and
 * whitening Wh (c, c) are functions of the trainable parameters and live on the device -- DEVICE pointers, like X (N, L, d: columns as
 * they come, no scaling inside, p->num_lags = 0) and Phi (N, F = 1 + c + (M-1) r); device-pointer mode only.  The projections of
 * levels 2..M are value-independent random objects: HOST arrays, kept on the device by content across calls.
 * _grad: dPhi (N, F) -> gX (N, L, d), gS (c, d), gWh (c, c), g_base[0] (the base kernel's own parameter, or NULL) -- what tf.gradients
 * returns for _K_seq_lr_feat (kernels.py:239-261) given the landmarks and the whitening; the caller chains gS and gWh through the
 * gather and the eigendecomposition (gpsig_amd/autodiff.py).  num_components <= 64. */
int gpsig_lr_seq_features_dev(gpsig_ctx* ctx, const gpsig_params* p, int32_t num_components, int32_t rank_bound, int32_t num_sketches,
                              const gpsig_sketch* sketches, const void* X, int64_t N, int32_t L, const double* S, const double* Wh, void* Phi);
int gpsig_lr_seq_features_grad(gpsig_ctx* ctx, const gpsig_params* p, int32_t num_components, int32_t rank_bound, int32_t num_sketches,
                               const gpsig_sketch* sketches, const void* X, int64_t N, int32_t L, const double* S, const double* Wh,
                               const void* dPhi, void* gX, double* gS, double* gWh, double* g_base);
/* SignatureKernel._K_seq_lr_feat (kernels.py:239-261): Nystrom_map + signature_kern_first_order_lr_feature.  Phi: (N, F). */
int gpsig_lr_seq_features(gpsig_ctx* ctx, const gpsig_params* p, const gpsig_lowrank* lr, const void* X, int64_t N, int32_t L, void* Phi);
/* SignatureKernel._K_tens_lr_feat (kernels.py:285-311): Nystrom_map + tensor_kern_lr_feature.  Phi: (T, F). */
int gpsig_lr_tens_features(gpsig_ctx* ctx, const gpsig_params* p, const gpsig_lowrank* lr, const void* Z, int64_t T,
                           int32_t increments, void* Phi);
/* Kernel matrix from factors: level Grams PhiA_m PhiB_m^T (fp64 MFMA), optional per-side level normalisation
 * 1/sqrt(|Phi_m|^2 + jitter) (kernels.py:457-469, :574-581), sigma*variances, level sum.  PhiB == NULL: symmetric, with
 * the jitter of kernels.py:431 on the diagonal when normalising.  out: (N1, N2) or (M+1, N1, N2). */
int gpsig_lr_kernel(gpsig_ctx* ctx, const gpsig_params* p, const gpsig_lowrank* lr, const void* PhiA, const void* PhiB, int64_t N1,
                    int64_t N2, int32_t normalize_a, int32_t normalize_b, int32_t return_levels, void* out);
/* diagonal sum_j Phi_m[n][j]^2 * sigma * variances[m] (kernels.py:499-510).  out: (N,) or (M+1, N). */
int gpsig_lr_kernel_diag(gpsig_ctx* ctx, const gpsig_params* p, const gpsig_lowrank* lr, const void* Phi, int64_t N,
                         int32_t return_levels, void* out);

/* ---- gradients (reverse mode) of the level primitives ------------------------------------------------------------
 * The reference has no gradient code: it trains through TensorFlow's autodiff of the graph these primitives stand
 * for (gpsig/training.py:149-164, gpsig/models.py:40-59).  Each function below takes the upstream gradient G of the
 * (M+1, ...) level array its forward counterpart returns and produces the gradients with respect to the inputs of
 * that primitive, i.e. what tf.gradients would return for _K_seq / _K_seq_diag / _K_tens / _K_tens_vs_seq
 * (gpsig/kernels.py:188-340) with order = 1.  As for the forward primitives, inputs are taken as they come (already
 * scaled, d = num_features * (num_lags + 1) columns); float64 only.  Outputs are overwritten.  g_base: 2 doubles in the
 * context's pointer mode or NULL; [0] receives the gradient with respect to base_params[0] (gamma of SignaturePoly,
 * kernels.py:844-848; mixing of SignatureMix, :881-892).  Option "grad_scratch_mb" bounds the lattice scratch. */
/* X2 == NULL: symmetric Gram, gX receives both roles of every sequence. */
int gpsig_seq_gram_levels_grad(gpsig_ctx* ctx, const gpsig_params* p, const void* X, const void* X2, int64_t N1, int64_t N2,
                               int32_t L1, int32_t L2, const void* G /* (M+1, N1, N2) */, void* gX, void* gX2, double* g_base);
/* The pair of them for a forward pass that will be differentiated (round 5; what the TensorFlow graph of the reference does implicitly -- the
 * forward op's intermediates are kept for its gradient op).  _levels_stash evaluates like gpsig_seq_gram_levels and, where the fused reverse
 * kernel can continue from it (SignatureRBF / SignatureMatern12 / 32 / 52 with differences, order 1, float64, at most 64 observations and 8
 * columns, num_levels 4 / 5, within
 * option "grad_stash_mb", default 4096; not inside a graph capture), keeps the forward recursion's row totals and final states in the context:
 * desc (8 integers) describes what was kept, desc[0] == 0 nothing.  _levels_grad_stash continues from it with the backward sweep only (*taken = 1),
 * or does nothing (*taken = 0: never kept, or another evaluation has overwritten it since) -- then call gpsig_seq_gram_levels_grad.  Device
 * pointers.  K(X) forward + backward of 1,024 sequences of 64 x 8, five levels: 17.0 -> 13 ms. */
int gpsig_seq_gram_levels_stash(gpsig_ctx* ctx, const gpsig_params* p, const void* X, const void* X2, int64_t N1, int64_t N2,
                                int32_t L1, int32_t L2, void* out, int64_t* desc /* 8 */);
int gpsig_seq_gram_levels_grad_stash(gpsig_ctx* ctx, const gpsig_params* p, const void* X, const void* X2, int64_t N1, int64_t N2,
                                     int32_t L1, int32_t L2, const void* G, void* gX, void* gX2, const int64_t* desc, int32_t* taken);
/* The explicit level features of the linear (cosine: of the unit vectors) kernel as an op of their own (round 4; no function of this name
 * in the reference: signature_algs.py:8-35 unrolled -- level m of SignatureLinear is <Phi_m(x), Phi_m(y)> -- and, for order > 1, the
 * truncated-exponential steps of :37-74; order = num_levels with difference on is the signature of the piecewise-linear path, what the
 * reference's notebook checks against esig).  out: (N, ld) doubles, ld = gpsig_seq_features_ld(p, L): columns [0, F) hold levels 1..M one
 * after the other, F = sum_m d^m, each in the natural order of its multi-indices (last index fastest, first index = earliest time), column F
 * is level 0 (= 1), the rest zeros.  With them the tensor-vs-sequence kernel of a rank-one tensor is <z_1 (x) .. (x) z_m, Phi_m(x)>
 * (signature_algs.py:101-127).  Inputs as they come (p->lengthscales NULL, no lags), float64; _ld returns 0 for shapes the feature kernels
 * are not built for (then _seq_features fails with GPSIG_ERR_UNSUPPORTED).  _grad: dPhi (N, ld) back through the feature sweep to gX
 * (N, L, d); Phi is the forward's output; device pointers.  Plain kernel launches on the context's stream: usable inside a graph capture. */
int64_t gpsig_seq_features_ld(const gpsig_params* p, int32_t L);
int gpsig_seq_features(gpsig_ctx* ctx, const gpsig_params* p, const void* X, int64_t N, int32_t L, void* out);
int gpsig_seq_features_grad(gpsig_ctx* ctx, const gpsig_params* p, const void* X, int64_t N, int32_t L, const void* Phi, const void* dPhi, void* gX);
/* The gradient of gpsig_kernel_K's level SUM (return_levels = 0) for SignatureLinear / SignatureCosine of every order through the
 * feature space (round 4): with K[i][j] = sum_m w_m <u_m(x_i), u_m(y_j)>, u_m = Phi_m / sqrt(|Phi_m|^2 + jitter) when p->normalization
 * (else Phi_m) and w_m = sigma variances[m], every level shares the upstream g (N1, N2), so ONE product g U(Y) over the whole feature
 * width (rocBLAS dgemm), the normalisation's reverse step per row and one reverse sweep per sequence give gX [, gX2] -- the
 * (M+1, N1, N2) level arrays of the level primitives and their upstream gradients never exist (what tf.gradients does through
 * kernels.py:401-476 for these kernels).  Inputs as they come, like the level primitives (p->lengthscales NULL, no lags); float64,
 * device-pointer mode.  X2 == NULL: symmetric Gram (a normalised one has a constant diagonal: no gradient from it).
 * g_weights: (M+1) doubles on the device or NULL; entries 1..M receive d/dw_m from the feature products; entry 0 (level 0 is constant)
 * and a normalised symmetric Gram's diagonal terms (trace of g) are plain sums of g left to the caller.
 * *taken = 0: the route does not apply (another base kernel, a shape the feature kernels are not built for, inside a graph capture,
 * option "sig_features_grad" 0 ...): nothing was written, use the level primitives.  g == NULL: only answer that question. */
int gpsig_kernel_K_grad(gpsig_ctx* ctx, const gpsig_params* p, const void* X, const void* X2, int64_t N1, int64_t N2, int32_t L1, int32_t L2,
                        const void* g /* (N1, N2) */, void* gX, void* gX2, void* g_weights, int32_t* taken);
int gpsig_seq_diag_levels_grad(gpsig_ctx* ctx, const gpsig_params* p, const void* X, int64_t N, int32_t L,
                               const void* G /* (M+1, N) */, void* gX, double* g_base);
int gpsig_tens_gram_levels_grad(gpsig_ctx* ctx, const gpsig_params* p, const void* Z, int64_t T, int32_t increments,
                                const void* G /* (M+1, T, T) */, void* gZ, double* g_base);
int gpsig_tens_vs_seq_levels_grad(gpsig_ctx* ctx, const gpsig_params* p, const void* Z, const void* X, int64_t T, int64_t N,
                                  int32_t L, int32_t increments, const void* G /* (M+1, T, N) */, void* gZ, void* gX,
                                  double* g_base);

/* Reverse pass of gpsig_tens_vs_seq_weighted: G (T, N) is the upstream gradient of the weighted sum; gZ, gX as above, gfac (N, M+1)
 * the gradient with respect to the factors (through which the level diagonals of the sequences, sigma and the variances are
 * reached: gpsig/kernels.py:572-581, :471).  float64.  aux: device pointer in either pointer mode. */
int gpsig_tens_vs_seq_weighted_grad(gpsig_ctx* ctx, const gpsig_params* p, const void* Z, const void* X, int64_t T, int64_t N,
                                    int32_t L, int32_t increments, const void* fac /* (N, M+1) */, const void* G /* (T, N) */,
                                    const void* aux /* what the forward call wrote, or NULL */, void* gZ, void* gX,
                                    void* gfac /* (N, M+1) */, double* g_base);

/* ---- the recursions on GIVEN increment lattices ("matrix route") ---------------------------------------------------------------
 * What gpsig/signature_algs.py does after it has differenced the base-kernel tensor (:25-26, :55-56 / :114): for callers that
 * build that tensor themselves -- state spaces wider than the gradient kernels' 64 columns, where it is a d-deep contraction and
 * belongs on a library GEMM, and base kernels whose parameters are differentiated outside this library (SignatureSpectral's alpha,
 * omega, gamma: gpsig/kernels.py:912-914).  The lattices of a block of pairs are held in scratch memory (option "grad_scratch_mb"):
 * sized for training minibatches, not for BASELINE-sized Grams.  float64.
 *   dM: (P, R1, R2) increment lattices of P pairs -> out (M+1, P): signature_kern_first_order / _higher_order (:28-35, :58-74), p->order;
 *   m:  (lt, R, P) component increments, pair index fastest -> out (M+1, P): signature_kern_tens_vs_seq_first_order (:116-127), order 1.
 * _grad: G (M+1, P) upstream -> the gradient with respect to dM / m, same layouts. */
int gpsig_lattice_levels(gpsig_ctx* ctx, const gpsig_params* p, const void* dM, int64_t P, int32_t R1, int32_t R2, void* out);
int gpsig_lattice_levels_grad(gpsig_ctx* ctx, const gpsig_params* p, const void* dM, int64_t P, int32_t R1, int32_t R2, const void* G, void* gdM);
int gpsig_chain_levels(gpsig_ctx* ctx, const gpsig_params* p, const void* m, int64_t P, int32_t R, void* out);
int gpsig_chain_levels_grad(gpsig_ctx* ctx, const gpsig_params* p, const void* m, int64_t P, int32_t R, const void* G, void* gm);

#ifdef __cplusplus
}
#endif
#endif /* GPSIG_HIP_H */
