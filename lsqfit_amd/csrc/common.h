// Shared declarations for the gfx950 backend (device launchers + handle).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <cstdio>
#include <exception>
#include <mutex>
#include <new>

#include "../../include/lsqfit_amd.h"

namespace lsqamd {

struct lsqamd_abi_no_handle { const char *err; };   // (abi_exception(nullptr): nothing to record the text in)

// ---- exception barrier of the C ABI ------------------------------------------------
// "never throw across the ABI" (the reference's callbacks are `noexcept`, src/lsqfit/_gsl.pyx:726-760, and a
// C++ exception unwinding into ctypes / cgo / JNI frames aborts the host process).  Every `extern "C"` export with a body of
// more than one statement is a function-try-block closed by LSQAMD_ABI_CATCH: std::bad_alloc -> LSQAMD_ENOMEM, anything
// else -> LSQAMD_EINTERNAL, the text in the handle's last_error where there is a (non-const) handle.
// (tests/test_abi.py checks the source for the guard on every export and drives it through lsqamd_debug_throw.)
template <class H> int abi_exception(H *h) noexcept {
  int code = LSQAMD_EINTERNAL;
  char buf[320];
  std::snprintf(buf, sizeof buf, "internal error: unknown C++ exception caught at the C ABI");
  try {
    throw;   // the exception in flight (only ever called from a catch handler)
  } catch (const std::bad_alloc &) {
    code = LSQAMD_ENOMEM;
    std::snprintf(buf, sizeof buf, "out of host memory (std::bad_alloc caught at the C ABI)");
  } catch (const std::exception &e) {
    std::snprintf(buf, sizeof buf, "internal error: %s (C++ exception caught at the C ABI)", e.what());
  } catch (...) {
  }
  if (h) {
    try { h->err = buf; } catch (...) {}
  }
  return code;
}
inline int abi_exception(std::nullptr_t) noexcept { return abi_exception(static_cast<lsqamd_abi_no_handle *>(nullptr)); }
#define LSQAMD_ABI_CATCH(...) catch (...) { __VA_ARGS__ }

// ---- a stream whose graph capture was invalidated ---------------------------------
// On ROCm 7 a legacy-stream call from ANY thread (hipMemcpy, a NULL-stream launch: the library makes none, other code in the
// process may) fails with hipErrorStreamCaptureImplicit while a capture is open and invalidates that capture; and
// hipStreamEndCapture then returns hipErrorStreamCaptureInvalidated but LEAVES the stream in the invalidated state: every
// later launch and synchronisation on it fails (measured: tools/dbg_capture_reset.hip).  What brings the stream back is a
// fresh (empty) begin / end capture pair.  Called on every failed capture before the work is queued again eagerly.
inline void capture_reset(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  hipGraph_t g = nullptr;
  if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) {
    (void)hipStreamEndCapture(st, &g);
    if (g) (void)hipGraphDestroy(g);
    g = nullptr;
  }
  (void)hipGetLastError();
  if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      (void)hipStreamEndCapture(st, &g);
      if (g) (void)hipGraphDestroy(g);
    }
  }
  (void)hipGetLastError();
}

// ---- once per DEVICE, from any thread ---------------------------------------------
// hipFuncSetAttribute(..., MaxDynamicSharedMemorySize, ...) applies to the CURRENT device only, and the ABI lets one
// process hold handles on several GPUs from several host threads (lsqamd_create records the device, lsqamd_query_devices
// enumerates them): every "set the attribute the first time this launcher runs" is therefore bookkeeping per device id,
// read lock-free on the launch path and written under a mutex.  Devices beyond the table run the setter every time.
struct PerDeviceOnce {
  static constexpr int kMaxDev = 64;
  std::atomic<uint64_t> done{0};   // bit d: device d holds the attributes
  std::mutex mu;
  template <class F> hipError_t run_for(int dev, F &&setter) {
    if (dev < 0 || dev >= kMaxDev) return setter();
    const uint64_t bit = 1ull << dev;
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    std::lock_guard<std::mutex> lk(mu);
    if (done.load(std::memory_order_relaxed) & bit) return hipSuccess;
    const hipError_t e = setter();
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
  }
  template <class F> hipError_t run(F &&setter) {
    int dev = 0;
    const hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    return run_for(dev, static_cast<F &&>(setter));
  }
};
// the same for an attribute that grows with the problem (a kernel whose dynamic LDS depends on the tape): the largest value
// set so far, per device; the setter runs under the mutex only when a launch needs more
struct PerDeviceMax {
  std::atomic<size_t> have[PerDeviceOnce::kMaxDev] = {};
  std::mutex mu;
  template <class F> hipError_t ensure_for(int dev, size_t want, F &&setter) {
    if (dev < 0 || dev >= PerDeviceOnce::kMaxDev) return setter(want);
    if (have[dev].load(std::memory_order_acquire) >= want) return hipSuccess;
    std::lock_guard<std::mutex> lk(mu);
    if (have[dev].load(std::memory_order_relaxed) >= want) return hipSuccess;
    const hipError_t e = setter(want);
    if (e == hipSuccess) have[dev].store(want, std::memory_order_release);
    return e;
  }
  template <class F> hipError_t ensure(size_t want, F &&setter) {
    int dev = 0;
    const hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    return ensure_for(dev, want, static_cast<F &&>(setter));
  }
};

// ---- fp64 MFMA GEMM, "TN" form (gemm_tn_f64.hip) ---------------------------------
// C[M x N] = alpha * sum_k X[k][m] * Y[k][n] + beta * C      (all row-major)
// Every dense contraction on the LM path has this shape with k-major operands:
//   J^T J (X = Y = J), whitening (X = W^T), Cholesky row-panel solve
//   (X = inv(U_kk)), trailing update (X = Y = panel), covariance (X = Y = U^-T).
struct GemmTN {
  const double *X = nullptr, *Y = nullptr;
  double *C = nullptr;
  int64_t M = 0, N = 0, K = 0;
  int64_t ldx = 0, ldy = 0, ldc = 0;
  double alpha = 1.0, beta = 0.0;
  int32_t batch = 1;
  int64_t sx = 0, sy = 0, sc = 0;  // batch strides (elements)
  int32_t upper_only = 0;          // skip tiles strictly below the diagonal (C square-aligned)
  int32_t x_upper_tri = 0;         // X[k][m] == 0 for k > m  -> k loop stops at the tile's last row
  int32_t xy_lower_tri = 0;        // X[k][m] == 0 for m > k (and same for Y) -> k loop starts at the tile
  int32_t splits = 1;              // split-K: split s writes C + s*split_stride (beta ignored)
  int64_t split_stride = 0;
  const int32_t *work_map = nullptr;  // device int4 list (tm, tn, split, 0), see syrk_work_fill
  int32_t n_work = 0;
  int32_t force_generic = 0;       // A/B switch: never take the direct-to-LDS interior kernel
  const int32_t *batch_active = nullptr;  // device flags[batch]: 0 = skip that batch entry
  // triangular X, few long tile rows: the caller offers a scratch slab at C + split_stride (same layout as C) and adds
  // it to C afterwards when gemm_tn_wants_tri_halves() says the launch will use it (see gemm_tn_f64.hip)
  int32_t tri_halves = 0;
  // fused column sums (triangular-X interior path only; ignored elsewhere -- check
  // gemm_tn_fuses_colsum()): colsum_out[(b * tiles_m + tm) * colsum_ld + n] = sum over the 128
  // rows of tile row tm of C[row][n] * C_b[row][colsum_rcol]   (the weights are a column of the
  // output matrix itself, written by an earlier launch: J^T f from the whitened Jacobian)
  // With a work list (the split-K J^T J launch, X == Y): colsum_out[(b * splits + split) * colsum_ld + m] =
  // sum over the split's rows k of X[k][m] * X[k][colsum_rcol] for m < M, and at m = colsum_rcol the sum of
  // X[k][colsum_rcol]^2 -- formed by the diagonal tiles' workgroups from the rows they stage anyway.
  double *colsum_out = nullptr;
  int64_t colsum_ld = 0, colsum_rcol = 0;
  // work-list launches: the list's 4th word is the entry's exchange group (1-based, syrk_work_fill_grouped); the workgroup
  // that completes an entry adds one to done_ctr[group - 1] (release) -- see launch_wait_counter
  int32_t *done_ctr = nullptr;
};
// true when launch_gemm_tn would take the kernel that honours colsum_out for this call
bool gemm_tn_fuses_colsum(const GemmTN &g);
bool gemm_tn_wants_tri_halves(const GemmTN &g);
hipError_t launch_gemm_tn(hipStream_t st, const GemmTN &g);
// Jacobian rows of a sum model synthesised INSIDE the whitening product (gemm_tn_f64.hip):
//   J_b = W_b x [d f / d p] for nb uniform triangular blocks of B rows, raw rows never written.
struct WhitenSynth {
  int32_t model = 0;            // LSQAMD_MODEL_COSMIX | _MULTIEXP
  const double *Wt = nullptr;   // [nb][B][B] transposed weights (upper triangular)
  const double *x = nullptr;    // [nb * B] predictor of every row (n_x = 1)
  const double *p = nullptr;    // [2 K] device parameters
  double *J = nullptr;          // [nb * B][ld] output rows; column P = 2 K holds the whitened residual already
  int64_t ld = 0, B = 0, K = 0;
  int32_t nb = 0;
  double *colsum_out = nullptr; // [(b * B / 128 + tm)][2 K]: per-tile-row pieces of J^T f
  const int32_t *trig_far = nullptr;   // device flag (launch_trig_range); null: the safe kernel only
};
bool whiten_synth_eligible(int32_t model, int64_t B, int64_t P);
hipError_t launch_whiten_synth(hipStream_t st, const WhitenSynth &a);
int64_t syrk_work_count(int64_t P, int32_t splits);
void syrk_work_fill(int64_t P, int32_t splits, int32_t *out);
int64_t syrk_work_fill_rows(int64_t P, int32_t splits, int row0, int row1, int32_t *out);   // tile rows [row0, row1) only
// one list whose entries complete group by group (rows[g] .. rows[g + 1] = tile rows of group g; 4th word = group + 1)
void syrk_work_fill_grouped(int64_t P, int32_t splits, int G, const int32_t *rows, int32_t *out, int32_t *count);
// a one-wave kernel on `st` that returns when *ctr >= expect (acquire; bounded: *timed_out = 1 after ~4 s)
hipError_t launch_wait_counter(hipStream_t st, const int32_t *ctr, int32_t expect, int32_t *timed_out);

// ---- Cholesky family (chol.hip) ---------------------------------------------------
constexpr int CHOL_NB = 128;
// workspace: inverses of the diagonal blocks, ceil(n/NB) * NB*NB doubles
size_t potrf_work_bytes(int64_t n);
// In-place upper Cholesky A = U^T U of the leading n x n block of A[n x n_cols]
// (row-major, lda); columns n..n_cols-1 are carried along (they become U^-T * A[:, n:]).
// *dev_info (device int32) is set to (first failing pivot + 1) if not positive definite.
// info_zeroed: *dev_info is 0 already (the caller's previous kernel did it): no 4-byte memset
hipError_t potrf_upper(hipStream_t st, double *A, int64_t n, int64_t lda, int64_t n_cols,
                       double *work, int32_t *dev_info, bool info_zeroed = false);
// batched: matrix b at A + b*strideA, block inverses at work + b*strideW, info[b];
// trailing update C -= P^T P (upper tiles; P = 128 x rest panel, C = mrest x rest, both multiples of
// 128, 16-byte aligned rows) fused with the diagonal block of the next step (= tile (0, 0) of C)
bool trail_potf2_available();
hipError_t launch_trail_potf2(hipStream_t st, const double *P, double *C, int64_t lda, int64_t mrest,
                              int64_t rest, int nb_next, double *uinv_next, int32_t *info, int32_t k0_next);
// diagonal block (nb <= 128) of the blocked Cholesky: A_kk -> U_kk in place, inv(U_kk) -> uinv
// (128 x 128, row-major); one workgroup per batch entry (potf2_mfma.hip)
hipError_t launch_potf2_mfma(hipStream_t st, double *A, int64_t lda, int nb, double *uinv, int32_t *info,
                             int32_t k0, int32_t batch, int64_t strideA, int64_t strideW,
                             const int32_t *active);

// entries with active[b] == 0 (if given) are skipped
hipError_t potrf_upper_batched(hipStream_t st, double *A, int64_t n, int64_t lda, int64_t n_cols,
                               double *work, int32_t *dev_info, int32_t batch, int64_t strideA,
                               int64_t strideW, const int32_t *active, bool info_zeroed = false);
hipError_t backsolve_upper_batched(hipStream_t st, const double *A, int64_t n, int64_t lda,
                                   const double *work, double *y_inout, int32_t batch, int64_t strideA,
                                   int64_t strideW, int64_t strideY, const int32_t *active,
                                   void *scratch = nullptr, int32_t *info_dev = nullptr, bool scratch_zeroed = false);
hipError_t trtri_upper_to_lower_T_batched(hipStream_t st, const double *A, int64_t n, int64_t lda,
                                          const double *work, double *Wl, int64_t ldw, int32_t batch,
                                          int64_t strideA, int64_t strideW, int64_t strideWl);
// v = U^-1 y (y = column `ycol` of A rows 0..n-1), using the diagonal-block inverses in work
// scratch (backsolve_scratch_bytes(n), with info_dev): the one-launch chain of workgroups (n a multiple of 128)
size_t backsolve_scratch_bytes(int64_t n);
// scratch_zeroed: the scratch is all zero already (launch_copy_column_zero): no memset before the chain
hipError_t backsolve_upper(hipStream_t st, const double *A, int64_t n, int64_t lda,
                           const double *work, double *y_inout, void *scratch = nullptr, int32_t *info_dev = nullptr,
                           bool scratch_zeroed = false);
// Wl (n x n, ld) = U^-T (lower triangular), given factored A and the block inverses.
hipError_t trtri_upper_to_lower_T(hipStream_t st, const double *A, int64_t n, int64_t lda,
                                  const double *work, double *Wl, int64_t ldw);
// sum_j log(A[j][j]) -> *dev_out (device double)
hipError_t logdiag_sum(hipStream_t st, const double *A, int64_t n, int64_t lda, double *dev_out);
extern long long *g_potf2_dbg;  // cycle stamps of the diagonal kernel (LSQAMD_POTF2_TIMING builds only)

// ---- model kernels (model.hip) ----------------------------------------------------
// one formula of a tape model with several (lsqamd_set_tape_programs): rows [row0, row0 + n_rows) run
// instructions [tape_off, tape_off + n_tape) of the handle's tape
struct TapeProgram {
  int64_t row0 = 0, n_rows = 0;
  int32_t tape_off = 0, n_tape = 0;
  const void *jit = nullptr;   // this formula compiled (jit.hip), or null: the forward-mode interpreter kernel
};
struct ModelArgs {
  int32_t model = 0;
  int64_t n_data = 0, n_param = 0;
  int32_t n_x = 0;
  const double *x = nullptr;          // [n_data][n_x]
  const double *ymean = nullptr;      // [n_data]
  const double *wdiag = nullptr;      // [n_data] 1/sdev of 1x1 rows
  const uint8_t *in_block = nullptr;  // [n_data] 1 if the row belongs to a correlated block (or null)
  const double *p = nullptr;          // [n_param] device
  const int32_t *tape = nullptr;
  int32_t n_tape = 0;
  const double *consts = nullptr;
  // batched fits sharing (x, ymean, wdiag): fit b uses p + b*p_stride and writes at + b*out_stride
  int32_t n_batch = 1;
  int64_t p_stride = 0, out_stride = 0;
  int64_t ymean_stride = 0;  // 0: the fits share ymean; n_data: fit b reads ymean + b*n_data
  const int32_t *batch_active = nullptr;
  // cosine model: device flag "some |w x| may reach the far range of the trig reduction" (launch_trig_range);
  // set: the kernel without far-range code runs unless the flag says otherwise.  null: the safe kernel only
  const int32_t *trig_far = nullptr;
  // tape model, single fit: buffers of the reverse-mode Jacobian (null: forward-mode kernel)
  const int32_t *tape_poff = nullptr;   // first partial-derivative slot of every instruction
  double *tape_part = nullptr;          // [tape_wgs * 4 waves][tape_slots][64]
  double *tape_jt = nullptr;            // [(n_param + 1)][tape_ldn]
  int64_t tape_ldn = 0, tape_wgs = 0;
  int32_t tape_slots = 0;
  const int32_t *tape_seg = nullptr;    // root-sum segments [n][3] (first, last instruction, sign), or null
  int32_t tape_n_seg = 0, tape_seg_depth = 0, tape_seg_slots = 0, tape_slot_cap = 0, tape_single = 0;
  const void *jit = nullptr;            // the tape compiled (jit.hip, lsqamd_jit::Kernel); null: the interpreter kernels
  const TapeProgram *progs = nullptr;   // host array: one formula per row range (n_prog > 0), instead of one tape for all rows
  int32_t n_prog = 0;
};
int tape_slots_of_op(int op);
// flag[0] = 1 unless max_k |q_k| * xmax is (finite and) below the fast range of the model kernels' trig reduction
hipError_t launch_trig_range(hipStream_t st, const double *q, int64_t n, double xmax, int32_t *flag);
// r_w[i] = w_i (f(x_i;p) - y_i) for 1x1 rows; r_raw[i] = f - y for rows inside blocks
hipError_t launch_residual_ex(hipStream_t st, const ModelArgs &m, double *r_w, double *r_raw);
// J[i][0..P) = w_i d f_i / d p ; J[i][P] = w_i delta_i (ld >= P+1); block rows -> J_raw unweighted
hipError_t launch_jacobian_ex(hipStream_t st, const ModelArgs &m, double *J_w, double *J_raw,
                              int64_t ld);

// rows with row_param[i] = j >= 0 are parameter rows f_i = p_j: overwrite what the model kernel left
hipError_t launch_param_rows(hipStream_t st, const int32_t *row_param, int64_t N, int64_t P, int64_t ld,
                             const double *p, const double *ymean, const double *wdiag, const uint8_t *in_block,
                             double *out_w, double *out_raw, int jac, int32_t n_batch = 1, int64_t p_stride = 0,
                             int64_t out_stride = 0);

// ---- vector / reduction kernels (vecops.hip) --------------------------------------
// out[j] = sum_i J[i][j] * J[i][rcol] for j in [0, ncols)   (ncols = P+1 gives grad and chi2)
hipError_t launch_colsum_dot(hipStream_t st, const double *J, int64_t nrows, int64_t ld,
                             int64_t ncols, int64_t rcol, double *partial, int64_t npartial,
                             double *out, const double *rvec = nullptr,   // rvec: weights instead of J[:, rcol]
                             int64_t rvec_stride = 1, int tri = 0);      // tri: J[i][j] == 0 for i > j, those rows are skipped
// out[j] = sum over nchunks rows of partial[chunk][j] (partial row stride = ncols)
hipError_t launch_colsum_reduce(hipStream_t st, const double *partial, int64_t nchunks, int64_t ncols,
                                double *out);
// y = A x (row-major, one wave per row)
hipError_t launch_gemv_rows(hipStream_t st, const double *A, int64_t ld, int64_t rows, int64_t cols,
                            const double *x, double *y);
// out[0] = sum_i r[i]^2   (partial: >= 1024 doubles)
hipError_t launch_sumsq(hipStream_t st, const double *r, int64_t n, double *partial, double *out);
// whitened block residual: r_out[row0_b + m] = sum_j Wt_b[j][m] * delta[row0_b + j]
hipError_t launch_block_whiten_vec(hipStream_t st, const double *wt, const int64_t *row0,
                                   const int64_t *bsize, const int64_t *woff, int32_t n_blocks,
                                   int64_t max_block, const double *delta, double *r_out,
                                   int32_t batch = 1, int64_t stride = 0,
                                   const int32_t *batch_active = nullptr,
                                   int64_t skip_from = INT64_MAX);  // blocks of >= skip_from rows are left out
// packed upper tiles <- sum over split-K slabs (matrix layout, P x ld each)
hipError_t launch_finalize_pack(hipStream_t st, const double *slabs, int32_t splits,
                                int64_t split_stride, int64_t P, int64_t ld, double *apk,
                                const double *prior = nullptr, int32_t prior_dense = 0,
                                int64_t tile0 = 0, int64_t n_tiles = -1);   // packed tiles [tile0, tile0 + n_tiles) only (-1: all)
// prior precision into the packed tiles (with_matrix) and into gvec = [J^T f ; chi2]
hipError_t launch_add_prior(hipStream_t st, double *apk, int64_t P, const double *prec, int32_t dense,
                            const double *pmean, const double *p, double *tvec, double *gvec,
                            int32_t with_matrix, int32_t tvec_ready = 0);
hipError_t launch_prior_chi2(hipStream_t st, int64_t P, const double *prec, int32_t dense,
                             const double *pmean, const double *p, double *tvec, double *scalar);
// M (P x ld, upper tiles) = A + mu diag(d^2); M[:, P] = g
hipError_t launch_build_damped(hipStream_t st, const double *apk, int64_t P, int64_t ld, double mu,
                               const double *diag, const double *g, double *Mout,
                               const double *frozen = nullptr, const double *mu_dev = nullptr, int32_t *info_zero = nullptr);
hipError_t launch_copy_column_zero(hipStream_t st, const double *src, int64_t ld, double *dst, int64_t n,
                                   void *zbuf, size_t zbytes);
hipError_t launch_packed_diag(hipStream_t st, const double *apk, int64_t P, double *out);
hipError_t launch_unpack_sym(hipStream_t st, const double *apk, int64_t P, double *out, int64_t ld);
hipError_t launch_symmetrize_from_upper(hipStream_t st, double *A, int64_t P, int64_t ld);
hipError_t launch_set_identity(hipStream_t st, double *A, int64_t P, int64_t ld);
// dst[i][j] += src[i][j]  (rows x cols, both with leading dimension ld; cols even, 16-byte aligned rows)
hipError_t launch_add_rows(hipStream_t st, double *dst, const double *src, int64_t rows, int64_t cols, int64_t ld);
hipError_t launch_copy_strided(hipStream_t st, const double *src, int64_t lds_, double *dst,
                               int64_t ldd, int64_t rows, int64_t cols);

// dst[c][r] = src[r][c] * scale[r] (rows with skip[r] != 0 unscaled); batched over blockIdx.z
hipError_t launch_transpose_scale(hipStream_t st, const double *src, int64_t lds_, double *dst,
                                  int64_t ldd, int64_t rows, int64_t cols, const double *scale,
                                  const uint8_t *skip, int64_t batch, int64_t s_src, int64_t s_dst);
// dst[r][:] = scale[r] * src[r][:] for rows with skip[r] == 0 (scale / skip may be null)
hipError_t launch_rows_scale_copy(hipStream_t st, const double *src, int64_t lds_, double *dst,
                                  int64_t ldd, int64_t rows, int64_t cols, const double *scale,
                                  const uint8_t *skip);

// out[m] (+)= |r_m|^2 for m points, r_m = r + m*stride
hipError_t launch_rows_sumsq(hipStream_t st, const double *r, int64_t n, int64_t stride, int64_t m,
                             double *out, int accumulate);
// out[m] += (p_m - pmean)^T Lambda (p_m - pmean); dense: Dt, T are P x ldt scratch (ldt >= m, even)
hipError_t launch_prior_chi2_points(hipStream_t st, int64_t P, const double *prec, int32_t dense,
                                    const double *pmean, const double *p, int64_t m, double *Dt,
                                    double *T, int64_t ldt, double *out);

// scaling matrix D on the device (scaling.c): init / update from the squared column norms
hipError_t launch_scale_update(hipStream_t st, int64_t P, int scaler, int init, const double *coln2,
                               double *dscale);
// xt = x - v
hipError_t launch_trial_point(hipStream_t st, int64_t P, const double *x, const double *v, double *xt);
// robust losses of the scipy plugin (robust.hip): out[0] = f_scale^2 sum rho((r_i / f_scale)^2) over r[i * stride]; rows of
// [J | f] rescaled in place (scipy's scale_for_robust_loss_function)
hipError_t launch_robust_cost(hipStream_t st, const double *r, int64_t n, int64_t stride, int loss, double f_scale, double *partial,
                              double *out);
hipError_t launch_robust_scale_rows(hipStream_t st, double *J, int64_t n, int64_t P, int64_t ld, int loss, double f_scale);
// tape functions beyond LSQAMD_OP_POWI (the interpreter kernels of model.hip / batch.hip; jit.hip emits the same formulas):
// value and derivative
__device__ __forceinline__ void tape_unary_ext(int op, double x, double &v, double &d) {
  switch (op) {
    case LSQAMD_OP_TAN: v = tan(x); d = 1.0 + v * v; break;
    case LSQAMD_OP_SINH: v = sinh(x); d = cosh(x); break;
    case LSQAMD_OP_COSH: v = cosh(x); d = sinh(x); break;
    case LSQAMD_OP_TANH: { v = tanh(x); const double e = exp(-2.0 * fabs(x)); d = 4.0 * e / ((1.0 + e) * (1.0 + e)); break; }   // sech^2 without 1 - v^2's cancellation
    case LSQAMD_OP_ASIN: v = asin(x); d = 1.0 / sqrt(1.0 - x * x); break;
    case LSQAMD_OP_ACOS: v = acos(x); d = -1.0 / sqrt(1.0 - x * x); break;
    case LSQAMD_OP_ABS: v = fabs(x); d = x >= 0.0 ? 1.0 : -1.0; break;
    default: v = x; d = 1.0; break;
  }
}
// LM state record on the device (plain lm; vecops.hip)
enum { LMS_CHI2 = 0, LMS_MU, LMS_NU, LMS_DELTA, LMS_VG, LMS_DV2, LMS_VFINITE, LMS_RHO, LMS_CHI2_TRIAL, LMS_ACCEPT,
       LMS_SOLVED, LMS_INFO, LMS_PIVMIN,
       LMS_SEQ /* counts the half steps published: the host may poll the mirror for it instead of sleeping in a stream synchronisation */,
       LMS_HOSTPTR /* bits of a device-visible HOST address the record is mirrored to by the kernels that finish a half step (0: none) */,
       LMS_CHECK /* lm_record_checksum of the other fifteen words, as bits: what makes the host mirror self-verifying */,
       LMS_COUNT = 16 };
// position-weighted sum over the record's words (all but LMS_CHECK): the device writes it with the record, the host recomputes it
// over a snapshot of the mirror -- stores to host memory arrive in no particular order, a sum that fits means they all have
__host__ __device__ inline unsigned long long lm_record_checksum(const unsigned long long *w) {
  unsigned long long acc = 0;
  for (int i = 0; i < LMS_COUNT; ++i)
    if (i != LMS_CHECK) acc += (w[i] ^ 0x9E3779B97F4A7C15ull) * (2ull * (unsigned long long)i + 1ull);
  return acc;
}
// U / ldu / a_diag given: state[LMS_PIVMIN] = min_i U_ii^2 / (a_diag_i + mu d_i^2), the smallest share of a column of the
// damped matrix that its Cholesky pivot retained (1 / it ~ the condition number the solve has just gone through)
hipError_t launch_lm_trial(hipStream_t st, int64_t P, const double *x, const double *v, const double *g,
                           const double *d, double *xt, double *state, const double *U = nullptr, int64_t ldu = 0,
                           const double *a_diag = nullptr);
hipError_t launch_lm_trial_tail(hipStream_t st, const double *r, int64_t n, double *partial, int64_t P,
                                const double *prec, int32_t dense, const double *pmean, const double *p, double *tvec,
                                bool with_prior, double *chi2_out, const int32_t *chol_info, double factor_up,
                                double factor_down, double *lmd);
// tvec / pmean given: g += tvec, chi2 += (x - pmean) . tvec first (the prior's share, prior_apply_kernel folded in)
hipError_t launch_lm_accept_tail(hipStream_t st, const double *apk, int64_t P, int scaler, double *coln2,
                                 double *dscale, const double *x, const double *v, double *gvec, double xtol,
                                 double gtol, double *lmd, const double *tvec = nullptr, const double *pmean = nullptr,
                                 // nrm_part: the <= 64 per-workgroup sums of the fused normal-equation kernel are totalled and
                                 // unpacked (into apk / gvec, prior precision added) by this launch first
                                 const double *nrm_part = nullptr, int nrm_blocks = 0, const double *nrm_prior = nullptr,
                                 int nrm_prior_dense = 0);
// small systems (n <= 256): back substitution + trial point + the record's dot products in one single-workgroup launch
hipError_t launch_lm_solve_tail_small(hipStream_t st, const double *M, int64_t ld, int64_t n, const double *uinv,
                                      const double *x, const double *g, const double *d, double *xt, double *v_out,
                                      double *lmd, const double *a_diag, const int32_t *chol_info);
// P <= 12: build (A + mu D^2 | g) from the packed tile, factor, solve, trial point, record -- one wave, one launch
hipError_t launch_lm_tiny12_solve(hipStream_t st, const double *apk, int64_t P, const double *g, const double *d,
                                  const double *x, double *xt, double *v_out, double *lmd, int32_t *chol_info, int watch);
// 13 <= P <= 32: the same, T = 16 / 24 / 32 elimination steps with scalar-register broadcasts
hipError_t launch_lm_small_solve(hipStream_t st, const double *apk, int64_t P, const double *g, const double *d,
                                 const double *x, double *xt, double *v_out, double *lmd, int32_t *chol_info, int watch);
// q = [J^T J upper | J^T f | chi2] of the fused normal-equation kernel -> packed tile (+ prior precision), gvec
hipError_t launch_nrm_unpack(hipStream_t st, const double *q, int64_t P, double *apk, double *gvec, const double *prior,
                             int32_t prior_dense);
// small fits (n <= 65536 residuals, diagonal or no prior): |f_trial|^2 + the prior's share + the decision in one launch
hipError_t launch_lm_trial_tail_small(hipStream_t st, const double *r, int64_t n, int64_t P, const double *prec,
                                      const double *pmean, const double *p, double *tvec, double *chi2_out,
                                      const int32_t *chol_info, double factor_up, double factor_down, double *lmd);
hipError_t launch_lm_decide(hipStream_t st, const double *chi2_trial, const int32_t *chol_info, double factor_up,
                            double factor_down, double *state);
hipError_t launch_lm_converge(hipStream_t st, int64_t P, const double *x, const double *v, const double *gvec,
                              double xtol, double gtol, double *state);

inline int64_t packed_doubles(int64_t P) {
  const int64_t T = (P + 127) / 128;
  return T * (T + 1) / 2 * 128 * 128;
}

}  // namespace lsqamd
