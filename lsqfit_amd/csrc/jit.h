// Compiled expression tapes (jit.hip): internal to the library, not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

namespace lsqamd_jit {

struct Kernel;   // a loaded code object with the residual and the Jacobian kernel of ONE tape; owned by a process-wide cache

struct LaunchArgs {
  const double *x = nullptr, *p = nullptr, *ymean = nullptr, *wdiag = nullptr;
  const unsigned char *in_block = nullptr;
  double *out_w = nullptr, *out_raw = nullptr;
  int64_t ld = 1, n_data = 0;
  // batch (blockIdx.y): fit b reads p + b * p_stride, ymean + b * ymean_stride, writes at out + b * out_stride
  int32_t n_batch = 1;
  int64_t p_stride = 0, out_stride = 0, ymean_stride = 0;
  const int *batch_active = nullptr;
};

// nullptr (and why) when hiprtc is not available or the formula is outside what the generator handles.  The kernel comes back
// RETAINED: whoever keeps the pointer hands it back with release() when done (a handle: when its tape is replaced or it is
// destroyed).  The cache keeps at most LSQAMD_JIT_CACHE_CAP (default 1024) loaded kernels: beyond that, kernels nobody holds are
// unloaded, least recently used first (a sweep whose literal constants change per data set no longer grows without bound).
const Kernel *compile_tape(const int32_t *code, int n_code, const double *consts, int n_consts, int P, int n_x, std::string &why);
void release(const Kernel *k);
// {loaded kernels, kernels currently held by someone, kernels unloaded so far}
void cache_stats(long long out[3]);
// residual: out[row] (ld ignored); Jacobian: out[row * ld + 0..P] with the residual in column P
hipError_t launch(const Kernel *k, hipStream_t st, bool jac, const LaunchArgs &a);
bool available(std::string *why);

// Few parameters (<= NRM_MAX_P), one lane per data row, uncorrelated rows: a third kernel accumulates J^T J (upper, row-major),
// J^T f and |f|^2 -- normal_nq() numbers per workgroup into partial[blocks][nq] -- without ever writing the Jacobian.
constexpr int NRM_MAX_P = 12;
int normal_nq(const Kernel *k);     // 0: this formula has no such kernel
hipError_t launch_normal(const Kernel *k, hipStream_t st, const LaunchArgs &a, double *partial, int blocks);

// A whole small fit in one launch (same formulas as the normal-equation kernel: <= NRM_MAX_P parameters, uncorrelated rows):
// plain Levenberg-Marquardt from p0 to convergence by ONE workgroup, see kLmDriver in jit.hip.  The struct is the kernel's
// argument block, member for member.
constexpr int FIT_MAX_ROWS = 8192, FIT_MAX_BLOCK_ROWS = 256;
// ... 13 to FIT_MAX_P parameters: the rows of the Jacobian go through LDS, fit_wide_rows(P) of them at a time (a correlated fit
// must fit one such chunk)
constexpr int FIT_MAX_P = 32;
constexpr int fit_wide_rows(int P) { return P <= 20 ? 256 : 128; }
// record block of a fit with P parameters: [24, 24 + 5 (P + 1)) x g D coln2 v, then 8 (five cycle counters: developer
// diagnostics), then the covariance (P x P)
constexpr int fit_host_diag(int P) { return 24 + 5 * (P + 1); }
constexpr int fit_host_cov(int P) { return fit_host_diag(P) + 8; }
constexpr int FIT_HOST_DOUBLES = fit_host_cov(FIT_MAX_P) + FIT_MAX_P * FIT_MAX_P;
struct FitArgs {
  const double *x, *ymean, *wdiag; long long n_data;
  // correlated rows (n_blocks > 0: at most FIT_MAX_BLOCK_ROWS rows in all): the handle's block tables and W^T factors
  const unsigned char *in_block; const double *wt; const long long *blk_row0, *blk_size, *blk_woff; long long n_blocks;
  const double *p0;                      // start (may be device-visible host memory)
  double *p, *p_trial, *dscale, *apk, *gvec, *v_out, *coln2, *st;
  const double *prior_prec, *prior_mean; // prior_prec null: no prior
  int prior_dense, scaler, maxit, watch;
  double xtol, gtol, factor_up, factor_down, hostptr_bits;
  double *cov; long long ldc; int want_cov, pad_;   // want_cov: (J^T J + prior)^-1 at the end point -> cov[i * ldc + j] and host[96 + i * P + j]
  double *host;                          // record block, DEVICE memory, FIT_HOST_DOUBLES: [0,16) record, [16] reason (1 done, 2 hand
                                         // the fit to the general path), [17..20] nit nfev njev ntrial, [21] 1 when the covariance
                                         // was formed, [22] its log det(J^T J + prior), [23] -, [24..) x g D coln2 v, counters, covariance
  // pub != null: device-visible pinned HOST block the finished record block is published to for a host that polls instead of
  // waiting for the stream.  Stores to host memory land in no particular order, so the block is self-verifying: words
  // [0, fit_host_cov(P) + P * P) are the record block, except [23] = fit_checksum over the others seeded with seq, and
  // [16] = fit_flag(reason, info, nit, seq), the word the host waits for (api.hip run_one_launch)
  unsigned long long *pub, seq;
};
// the self-verification of a published record block (host side; the kernel's twin is at the end of lm_fit in jit.hip)
inline unsigned long long fit_checksum(const unsigned long long *w, int n_words, unsigned long long seq) {
  unsigned long long acc = seq * 0xD6E8FEB86659FD93ull;
  for (int i = 0; i < n_words; ++i)
    if (i != 16 && i != 23) acc += (w[i] ^ 0x9E3779B97F4A7C15ull) * (2ull * (unsigned long long)i + 1ull);
  return acc;
}
inline unsigned long long fit_flag(int reason, int info, int nit, unsigned long long seq) {
  return (unsigned long long)(reason & 0xff) | ((unsigned long long)(info & 0xff) << 8) | ((unsigned long long)(nit & 0xffffff) << 16) | (seq << 40);
}
bool has_fit_kernel(const Kernel *k);
int64_t fit_row_limit(const Kernel *k, bool correlated);   // most rows the kernel takes (0: no kernel)
hipError_t launch_fit(const Kernel *k, hipStream_t st, const FitArgs &a);
// ... and many same-shape fits, one workgroup each (the kernel's second argument block, member for member; the pointers
// of FitArgs are those of fit 0, `host` is DEVICE scratch of scratch_stride >= fit_host_cov(P) + P * P + 32 + P doubles per fit)
struct FitBatch {
  long long ymean_stride, prec_stride, tile_stride, cov_stride, scratch_stride;
  double *logdet, *mu, *chi2;
  int *nit, *info, *status, *nfev, *njev, *active, *reason;   // reason: 1 done + covariance, 3 done, 2 irregular
};
bool has_batch_fit_kernel(const Kernel *k);
hipError_t launch_fit_batch(const Kernel *k, hipStream_t st, const FitArgs &a, const FitBatch &b, int n_fits);

}  // namespace lsqamd_jit
