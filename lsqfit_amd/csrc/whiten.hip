// Whitening set-up on the device: the O(B^3) part of what nonlinear_fit obtains from
// gvar.PDF(...) at src/lsqfit/__init__.py:1892-1900 (per covariance block: factor the correlation
// matrix, build the weights W with W^T W = inv(C), log det C; svdcut semantics
// doc/source/overview.rst:1546-1606).  gvar eigen-decomposes every block on the host; when the
// svdcut floor touches no mode (the usual case) any W with W^T W = inv(C) gives the same chi2,
// J^T J, J^T f, p, cov and logGBF, so this path uses the Cholesky factor:
//
//   C = D corr D,  corr = U^T U  (potrf_upper, batched over same-size blocks)
//   Wl = U^-T (trtri)            W = Wl D^-1,   stored transposed: Wt = D^-1 U^-1 (upper triangular)
//   log det C = 2 sum log U_jj + 2 sum log sd_j
//   prior blocks also need inv(C) = W^T W (TN SYRK of W, as the post-fit covariance does)
//
// and decides ON THE DEVICE whether the floor could touch a mode: rigorous bounds first
// (lambda_max <= max_i sum_j |corr_ij|, lambda_min >= 1 / |Wl|_F^2 since |inv(corr)|_2 <= tr inv(corr)),
// then -- only for blocks the bounds leave undecided -- inverse iteration with the factor
// (u <- Wl^T Wl u, 40 sweeps).  Blocks whose floor may bind (status 1) or that are not positive
// definite (status 2) go to the host's eigen route (lsqfit_amd/whiten.py), which is what gvar does
// for every block.
#include <cmath>
#include <vector>

#include "common.h"

namespace lsqamd {
namespace {

constexpr int64_t rup(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

__device__ __forceinline__ double wsum_w(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// sd[b][i] = sqrt(C_b[i][i]); bad[b] = 1 when a variance is not positive and finite
__global__ __launch_bounds__(256) void sdev_kernel(const double *A, int64_t B, int64_t lda, int64_t strideA,
                                                   double *sd, double *logsd_sum, int32_t *bad) {
  __shared__ double sh[4];
  const double *a = A + (int64_t)blockIdx.x * strideA;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < B; i += 256) {
    const double v = a[i * lda + i];
    const bool ok = v > 0.0 && v < 1.0e300;
    if (!ok) bad[blockIdx.x] = 1;
    const double s = ok ? sqrt(v) : 1.0;
    sd[(int64_t)blockIdx.x * B + i] = s;
    acc += log(s);
  }
  acc = wsum_w(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) logsd_sum[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// A_b <- corr_b = C_b / (sd_i sd_j) in place; rowsum[b][i] = sum_j |corr_ij|  (one wave per row)
__global__ __launch_bounds__(256) void to_corr_kernel(double *A, int64_t B, int64_t lda, int64_t strideA,
                                                      const double *sd, double *rowsum) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B) return;
  double *a = A + (int64_t)blockIdx.y * strideA + i * lda;
  const double *s = sd + (int64_t)blockIdx.y * B;
  const double si = 1.0 / s[i];
  double acc = 0.0;
  for (int64_t j = lane; j < B; j += 64) {
    const double v = (j == i) ? 1.0 : a[j] * si / s[j];
    a[j] = v;
    acc += fabs(v);
  }
  acc = wsum_w(acc);
  if (lane == 0) rowsum[(int64_t)blockIdx.y * B + i] = acc;
}

// out[b] = sum_j log U_b[j][j]
__global__ __launch_bounds__(256) void logdiag_batched_kernel(const double *A, int64_t B, int64_t lda,
                                                              int64_t strideA, double *out) {
  __shared__ double sh[4];
  const double *a = A + (int64_t)blockIdx.x * strideA;
  double acc = 0.0;
  for (int64_t j = threadIdx.x; j < B; j += 256) acc += log(a[j * lda + j]);
  acc = wsum_w(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// fro2[b][i] = sum_j Wl_b[i][j]^2 (row i of the lower-triangular inverse factor: j <= i);
// optionally W_b[i][j] = Wl_b[i][j] / sd_j (the untransposed weights, for inv(C) = W^T W)
__global__ __launch_bounds__(256) void inv_rows_kernel(const double *Wl, int64_t B, int64_t ldw, int64_t strideW,
                                                       const double *sd, double *W, double *fro2) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B) return;
  const double *w = Wl + (int64_t)blockIdx.y * strideW + i * ldw;
  double *o = W ? W + (int64_t)blockIdx.y * strideW + i * ldw : nullptr;
  const double *s = sd + (int64_t)blockIdx.y * B;
  double acc = 0.0;
  for (int64_t j = lane; j < B; j += 64) {
    const double v = j <= i ? w[j] : 0.0;
    acc += v * v;
    if (o) o[j] = v / s[j];
  }
  acc = wsum_w(acc);
  if (lane == 0) fro2[(int64_t)blockIdx.y * B + i] = acc;
}

// Wt_b row i: scaled by 1 / sd_i, entries left of the diagonal cleared (Wt = D^-1 U^-1 is upper triangular)
__global__ __launch_bounds__(256) void wt_rows_kernel(double *wt, const double *sd, int64_t B) {
  const int64_t i = blockIdx.x, b = blockIdx.y;
  const double s = 1.0 / sd[b * B + i];
  double *row = wt + (b * B + i) * B;
  for (int64_t j = threadIdx.x; j < B; j += 256) row[j] = j >= i ? row[j] * s : 0.0;
}

// u <- w / |w|, |w|^2 given in *nrm2
__global__ __launch_bounds__(256) void normalise_kernel(const double *w, const double *nrm2, double *u, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) u[i] = w[i] / sqrt(nrm2[0]);
}

struct Plan {
  int64_t ldw, strideA;
  double *A, *Wl, *uinv, *sd, *rowsum, *fro2, *logsd, *logu, *vec, *partial, *scal;
  int32_t *info, *bad;
  size_t bytes, readback;
};

Plan make_plan(int64_t B, int64_t nb, void *base) {
  Plan p;
  p.ldw = rup(B, 16);
  p.strideA = B * p.ldw;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    void *q = base ? (char *)base + off : nullptr;
    off += (bytes + 255) / 256 * 256;
    return q;
  };
  p.A = (double *)take(sizeof(double) * nb * p.strideA);
  p.Wl = (double *)take(sizeof(double) * nb * p.strideA);
  p.uinv = (double *)take(potrf_work_bytes(B) * nb);
  p.sd = (double *)take(sizeof(double) * nb * B);
  // (what the host reads back, in one piece: rowsum .. bad are fetched by ONE copy, see lsqamd_whiten_blocks)
  p.rowsum = (double *)take(sizeof(double) * nb * B);
  p.fro2 = (double *)take(sizeof(double) * nb * B);
  p.logsd = (double *)take(sizeof(double) * nb);
  p.logu = (double *)take(sizeof(double) * nb);
  p.info = (int32_t *)take(sizeof(int32_t) * nb);
  p.bad = (int32_t *)take(sizeof(int32_t) * nb);
  p.readback = (size_t)((char *)p.bad - (char *)p.rowsum) + sizeof(int32_t) * (size_t)nb;
  p.vec = (double *)take(sizeof(double) * 3 * B);
  p.partial = (double *)take(sizeof(double) * (256 * B > 2048 ? 256 * B : 2048));
  p.scal = (double *)take(sizeof(double) * 8);
  p.bytes = off;
  return p;
}

#define WCHK(expr)                          \
  do {                                      \
    if ((expr) != hipSuccess) return LSQAMD_EHIP; \
  } while (0)

}  // namespace
}  // namespace lsqamd

using namespace lsqamd;

extern "C" {

size_t lsqamd_whiten_work_bytes(int64_t block_size, int32_t n_blocks) try {
  if (block_size < 1 || n_blocks < 1) return 0;
  return make_plan(block_size, n_blocks, nullptr).bytes + 256;
} LSQAMD_ABI_CATCH((void)lsqamd::abi_exception(nullptr); return 0;)

int lsqamd_whiten_blocks(void *stream, int64_t B, int32_t nb, const double *cov, double svdcut,
                         double *wt_out, double *prec_out, void *dev_work, size_t work_bytes,
                         double *logdet_out, double *lam_min_out, double *lam_max_out, int32_t *status_out) try {
  if (B < 1 || nb < 1 || !cov || !wt_out || !dev_work || !logdet_out || !status_out) return LSQAMD_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char *base = (char *)dev_work;
  const size_t pad = (size_t)((-(intptr_t)base) & 255);
  if (work_bytes < pad) return LSQAMD_ENOMEM;
  Plan p = make_plan(B, nb, base + pad);
  if (work_bytes < p.bytes + pad) return LSQAMD_ENOMEM;
  const int64_t ldw = p.ldw, sA = p.strideA;
  // covariance blocks (host or device memory, contiguous [nb][B][B]) -> A (ldw)
  WCHK(hipMemcpy2DAsync(p.A, sizeof(double) * ldw, cov, sizeof(double) * B, sizeof(double) * B, (size_t)(nb * B),
                        hipMemcpyDefault, st));
  WCHK(hipMemsetAsync(p.bad, 0, sizeof(int32_t) * nb, st));
  hipLaunchKernelGGL(sdev_kernel, dim3((unsigned)nb), dim3(256), 0, st, p.A, B, ldw, sA, p.sd, p.logsd, p.bad);
  hipLaunchKernelGGL(to_corr_kernel, dim3((unsigned)((B + 3) / 4), (unsigned)nb), dim3(256), 0, st, p.A, B, ldw, sA,
                     p.sd, p.rowsum);
  const int64_t sW = (int64_t)(potrf_work_bytes(B) / sizeof(double));
  WCHK(potrf_upper_batched(st, p.A, B, ldw, B, p.uinv, p.info, nb, sA, sW, nullptr));
  hipLaunchKernelGGL(logdiag_batched_kernel, dim3((unsigned)nb), dim3(256), 0, st, p.A, B, ldw, sA, p.logu);
  WCHK(trtri_upper_to_lower_T_batched(st, p.A, B, ldw, p.uinv, p.Wl, ldw, nb, sA, sW, sA));
  // A is free from here on: it takes W = Wl D^-1 when inv(C) is wanted
  hipLaunchKernelGGL(inv_rows_kernel, dim3((unsigned)((B + 3) / 4), (unsigned)nb), dim3(256), 0, st, p.Wl, B, ldw, sA,
                     p.sd, prec_out ? p.A : nullptr, p.fro2);
  // Wt_b[i][j] = Wl_b[j][i] / sd_i : transposed weights, contiguous B x B as lsqamd_set_data takes them
  WCHK(launch_transpose_scale(st, p.Wl, ldw, wt_out, B, B, B, nullptr, nullptr, nb, sA, B * B));
  hipLaunchKernelGGL(wt_rows_kernel, dim3((unsigned)B, (unsigned)nb), dim3(256), 0, st, wt_out, p.sd, B);
  if (prec_out) {
    GemmTN g;   // inv(C_b) = W_b^T W_b (upper tiles), then mirrored
    g.X = p.A; g.Y = p.A; g.ldx = g.ldy = ldw; g.sx = g.sy = sA;
    g.C = prec_out; g.ldc = B; g.sc = B * B;
    g.M = B; g.N = B; g.K = B;
    g.upper_only = 1;
    g.xy_lower_tri = 1;
    g.batch = nb;
    WCHK(launch_gemm_tn(st, g));
    for (int32_t b = 0; b < nb; ++b) WCHK(launch_symmetrize_from_upper(st, prec_out + (int64_t)b * B * B, B, B));
  }
  // one copy for everything the host decides on (six small pageable copies were a third of a 48-row block's 0.18 ms)
  std::vector<char> back(p.readback);
  WCHK(hipMemcpyAsync(back.data(), p.rowsum, p.readback, hipMemcpyDeviceToHost, st));
  WCHK(hipStreamSynchronize(st));
  auto at = [&](const void *dev) { return back.data() + ((const char *)dev - (const char *)p.rowsum); };
  const double *rowsum = (const double *)at(p.rowsum), *fro2 = (const double *)at(p.fro2);
  const double *logsd = (const double *)at(p.logsd), *logu = (const double *)at(p.logu);
  const int32_t *info = (const int32_t *)at(p.info), *bad = (const int32_t *)at(p.bad);
  const double cut = std::fabs(svdcut);
  for (int32_t b = 0; b < nb; ++b) {
    double hi = 0.0, tr = 0.0;
    for (int64_t i = 0; i < B; ++i) {
      hi = std::fmax(hi, rowsum[(size_t)(b * B + i)]);
      tr += fro2[(size_t)(b * B + i)];
    }
    if (hi > (double)B) hi = (double)B;            // trace of a correlation matrix
    double lo = 1.0 / tr;                          // rigorous: lambda_min >= 1 / tr inv(corr)
    logdet_out[b] = 2.0 * (logu[b] + logsd[b]);
    int status = 0;
    if (bad[b] || info[b] != 0 || !std::isfinite(tr) || !std::isfinite(logdet_out[b])) {
      status = 2;
      lo = 0.0;
    } else if (cut > 0.0 && !(lo > cut * hi)) {
      // undecided by the bounds: inverse iteration with the factor, u <- Wl^T (Wl u); the estimate
      // approaches lambda_min from above, hence the margin the host applied before (x4)
      const double *Wl = p.Wl + (int64_t)b * sA;
      double *u = p.vec, *t = p.vec + B, *w = p.vec + 2 * B;
      std::vector<double> u0((size_t)B);
      uint64_t s = 0x9E3779B97F4A7C15ull;
      for (int64_t i = 0; i < B; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        u0[(size_t)i] = (double)(s >> 11) / 9007199254740992.0 - 0.5;
      }
      WCHK(hipMemcpyAsync(w, u0.data(), sizeof(double) * B, hipMemcpyHostToDevice, st));
      double nrm2 = 1.0;
      for (int it = 0; it < 40; ++it) {
        WCHK(launch_sumsq(st, w, B, p.partial, p.scal));
        hipLaunchKernelGGL(normalise_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, w, p.scal, u, B);
        WCHK(launch_gemv_rows(st, Wl, ldw, B, B, u, t));
        WCHK(launch_colsum_dot(st, Wl, B, ldw, B, 0, p.partial, 256, w, t, 1));
      }
      WCHK(launch_sumsq(st, w, B, p.partial, p.scal));
      WCHK(hipMemcpyAsync(&nrm2, p.scal, sizeof(double), hipMemcpyDeviceToHost, st));
      WCHK(hipStreamSynchronize(st));
      const double est = 1.0 / std::sqrt(nrm2);    // |inv(corr) u| -> 1 / lambda_min
      if (std::isfinite(est) && est > 4.0 * cut * hi) lo = est / 4.0;
      else status = 1;
    }
    status_out[b] = status;
    if (lam_min_out) lam_min_out[b] = lo;
    if (lam_max_out) lam_max_out[b] = hi;
  }
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)

}  // extern "C"
