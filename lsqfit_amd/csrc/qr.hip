// solver = 'qr' (the reference's default, src/lsqfit/_gsl.pyx:571,646-647; 'svd' :650-651): the
// post-fit covariance with the accuracy of an orthogonal factorisation of J.
//
// GSL's qr solver factors the Jacobian itself, so gsl_multifit_nlinear_covar (_gsl.pyx:704-706)
// delivers (J^T J)^-1 with an error ~ cond(J) eps; the normal equations (solver = 'cholesky')
// square the condition number.  examples/y-noerr.out at nexp = 5 has cond(J) = 7e9: 4 % error
// in the covariance from the normal equations, 1e-10 from this route.
//
// Device form: column-equilibrated, shifted CholeskyQR with re-orthogonalisation (Yamamoto et
// al. / Fukaya et al.) -- every pass is the fp64 MFMA work the LM step is already made of:
//     J_s = J D_c                         (D_c = diag(J^T J)^-1/2)
//     pass 1   R_1 = chol(J_s^T J_s [+ s I])   from the Gram matrix the fit already holds;
//              Q_1 = J_s R_1^-1                (TN GEMM on a transposed copy of J)
//     pass k   G_k = Q_(k-1)^T Q_(k-1)  (split-K SYRK, all-reduced when rows are sharded) ;
//              R_k = chol(G_k) ;  stop once max |G_k - I| < 1e-6 (one more factor makes it eps)
//     R = R_k ... R_1 ,  cov = D_c R^-1 R^-T D_c ,  log det J^T J = 2 sum_k log det R_k - 2 log det D_c
// The shift s is only taken when the first factorisation meets a non-positive pivot (cond(J_s)^2
// beyond 1/eps); the prior enters every Gram matrix through its precision, Lambda_k =
// (R^-T) D_c Lambda D_c (R^-1), never through J.  LM steps themselves stay on the damped normal
// equations: the stationary point is defined by g = J^T f = 0, which is formed from J directly, so
// the end point does not depend on how accurately each damped step is solved.
#include <cmath>
#include <vector>

#include "fit_state.h"

using namespace lsqamd;

namespace {

constexpr int TB = 128;
constexpr int64_t rup(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

__device__ __forceinline__ double wmax(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}

// dc[j] = 1 / sqrt(A_jj) (1 when A_jj is not positive) from the packed diagonal
__global__ __launch_bounds__(256) void col_equil_kernel(const double *diag, int64_t P, double *dc, double *logdc) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= P) return;
  const double d = diag[j];
  const double s = (d > 0.0 && d < 1.0e300) ? 1.0 / sqrt(d) : 1.0;
  dc[j] = s;
  logdc[j] = log(s);
}

// M (upper tiles) = cs_i cs_j A_ij + shift [i == j]   from packed tiles; cs may be null
__global__ __launch_bounds__(256) void build_scaled_kernel(const double *apk, int64_t P, int64_t T, int64_t ld,
                                                           const double *cs, double shift, double *M,
                                                           const double *extra = nullptr) {
  int64_t t = blockIdx.x, tm = 0;
  while (t >= T - tm) { t -= T - tm; ++tm; }
  const int64_t tn = tm + t;
  const double *src = apk + (int64_t)blockIdx.x * TB * TB;
  const int c = threadIdx.x & 127;
  for (int rr = (threadIdx.x >> 7); rr < 8; rr += 2) {
    const int r = blockIdx.y * 8 + rr;
    const int64_t i = tm * TB + r, j = tn * TB + c;
    if (i < P && j < P) {
      double v = src[r * TB + c];
      if (extra && i == j) v += extra[i];      // (the damping term mu D^2 of a trial solve)
      if (cs) v *= cs[i] * cs[j];
      if (i == j) v += shift;
      M[i * ld + j] = v;
    }
  }
}

// part[block] = max over the block's elements (i <= j < P) of |A_ij - delta_ij|
__global__ __launch_bounds__(256) void packed_maxdev_kernel(const double *apk, int64_t P, int64_t T, double *part) {
  __shared__ double sh[4];
  int64_t t = blockIdx.x, tm = 0;
  while (t >= T - tm) { t -= T - tm; ++tm; }
  const int64_t tn = tm + t;
  const double *src = apk + (int64_t)blockIdx.x * TB * TB;
  const int c = threadIdx.x & 127;
  double m = 0.0;
  for (int rr = (threadIdx.x >> 7); rr < 8; rr += 2) {
    const int r = blockIdx.y * 8 + rr;
    const int64_t i = tm * TB + r, j = tn * TB + c;
    if (i < P && j < P && i <= j) m = fmax(m, fabs(src[r * TB + c] - (i == j ? 1.0 : 0.0)));
  }
  m = wmax(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[(int64_t)blockIdx.x * gridDim.y + blockIdx.y] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

__global__ __launch_bounds__(256) void max_stage2_kernel(const double *part, int64_t n, double *out) {
  __shared__ double sh[4];
  double m = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const double v = part[i];
    m = (v > m || v != v) ? v : m;      // a NaN wins: the caller must see it
  }
  m = wmax(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

// apk (upper tiles) += D (P x P, ldd)
__global__ __launch_bounds__(256) void packed_add_dense_kernel(double *apk, int64_t P, int64_t T, const double *D,
                                                               int64_t ldd) {
  int64_t t = blockIdx.x, tm = 0;
  while (t >= T - tm) { t -= T - tm; ++tm; }
  const int64_t tn = tm + t;
  double *dst = apk + (int64_t)blockIdx.x * TB * TB;
  const int c = threadIdx.x & 127;
  for (int rr = (threadIdx.x >> 7); rr < 8; rr += 2) {
    const int r = blockIdx.y * 8 + rr;
    const int64_t i = tm * TB + r, j = tn * TB + c;
    if (i < P && j < P) dst[r * TB + c] += D[i * ldd + j];
  }
}

// dst[c][r] = src[r][c] * cs[c]   (64 x 64 tiles through LDS; src rows x cols, dst cols x rows)
__global__ __launch_bounds__(256) void transpose_cscale_kernel(const double *src, int64_t lds_, double *dst, int64_t ldd,
                                                               int64_t rows, int64_t cols, const double *cs) {
  __shared__ double tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int64_t r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[r * lds_ + c] * (cs ? cs[c] : 1.0) : 0.0;
  }
  __syncthreads();
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int64_t c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[c * ldd + r] = tile[tx][i];
  }
}

// A[i][j] *= s[i] * s[j]
__global__ __launch_bounds__(256) void sym_scale_kernel(double *A, int64_t P, int64_t ld, const double *s) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = blockIdx.y;
  if (j < P) A[i * ld + j] *= s[i] * s[j];
}

// out[0] = sum_j log U[j][j] + in0 ; out[1] = sum v[j]
__global__ __launch_bounds__(256) void sum_kernel(const double *v, int64_t n, double *out) {
  __shared__ double sh[4];
  double a = 0.0;
  for (int64_t j = threadIdx.x; j < n; j += 256) a += v[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = sh[0] + sh[1] + sh[2] + sh[3];
}

// e[j] = mu d[j]^2 ; diag[j] += e[j]
__global__ __launch_bounds__(256) void damp_vec_kernel(int64_t P, double mu, const double *d, double *e, double *diag) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= P) return;
  const double v = mu * d[j] * d[j];
  e[j] = v;
  diag[j] += v;
}

// T[i][:] += e[i] * B[i][:]
__global__ __launch_bounds__(256) void rows_axpy_kernel(double *T, const double *B, int64_t ld, int64_t P, const double *e) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = blockIdx.y;
  if (j < P) T[i * ld + j] += e[i] * B[i * ld + j];
}

// out[j] = a[j] * b[j]
__global__ __launch_bounds__(256) void vec_mul_kernel(int64_t P, const double *a, const double *b, double *out) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < P) out[j] = a[j] * b[j];
}

struct QrPlan {
  int64_t ldn;
  double *T, *Q, *Ri, *Wtot, *Ritot, *T1, *Bm, *Gpk, *dc, *logdc, *part, *scal, *damp, *dwork, *vwork;
  size_t bytes;
};

QrPlan qr_plan(const lsqamd_fit *f, void *base) {
  QrPlan p;
  const int64_t N = f->N, P = f->P;
  p.ldn = rup(N > 0 ? N : 1, 16);
  size_t off = 0;
  auto take = [&](int64_t count) {
    double *q = base ? reinterpret_cast<double *>((char *)base + off) : nullptr;
    off += (size_t)rup((count < 1 ? 1 : count) * (int64_t)sizeof(double), 256);
    return q;
  };
  p.T = take(P * p.ldn);
  p.Q = take((N > 0 ? N : 1) * f->ld);
  p.Ri = take(P * f->ldm);
  p.Wtot = take(P * f->ldm);
  p.Ritot = take(P * f->ldm);
  p.T1 = take(P * f->ldm);
  p.Bm = take(P * f->ldm);
  p.Gpk = take(f->npk + P + 1);
  p.dc = take(P);
  p.logdc = take(P);
  const int64_t T = (P + TB - 1) / TB;
  p.part = take(T * (T + 1) / 2 * 16);
  p.scal = take(8);
  p.damp = take(P);
  p.dwork = take(P);
  p.vwork = take(2 * P);
  p.bytes = off;
  return p;
}

hipError_t launch_transpose_sq(hipStream_t st, const double *src, int64_t lds_, double *dst, int64_t ldd, int64_t P) {
  return launch_transpose_scale(st, src, lds_, dst, ldd, P, P, nullptr, nullptr, 1, 0, 0);
}

}  // namespace

namespace lsqamd_host {

size_t qr_work_bytes(const lsqamd_fit *f) { return qr_plan(f, nullptr).bytes + 256; }

// [J ; W_prior ; sqrt(mu) D] D_c = Q R by shifted, re-orthogonalised CholeskyQR (mu = 0: no damping rows).
// Leaves q.Wtot = R^-T (lower triangular), q.dc = D_c, *logdet_r = log det R; f->qr_passes / qr_delta say how it went.
static int qr_factor(lsqamd_fit *f, QrPlan &q, double mu, double *logdet_out) {
  const int64_t P = f->P, N = f->N, ldm = f->ldm;
  hipStream_t st = f->st;
  const int64_t T = (P + TB - 1) / TB;
  const dim3 tgrid((unsigned)(T * (T + 1) / 2), 16);
  const unsigned pb = (unsigned)((P + 255) / 256);
  const bool damped = mu > 0.0;
  const bool mine = f->adds_prior;
  if (const int rcj = ensure_J(f)) return rcj;   // (few-parameter fits form their normal equations without writing J)            // the replicated terms (prior, damping) enter the sums on one rank
  // D_c from the Gram matrix of the current point (all-reduced, prior included; + mu D^2 for a damped solve)
  HIPCHK(f, launch_packed_diag(st, f->redbuf, P, q.dwork));
  if (damped) hipLaunchKernelGGL(damp_vec_kernel, dim3(pb), dim3(256), 0, st, P, mu, f->dscale, q.damp, q.dwork);
  hipLaunchKernelGGL(col_equil_kernel, dim3(pb), dim3(256), 0, st, q.dwork, P, q.dc, q.logdc);
  double logdet_r = 0.0, shift = 0.0, delta = INFINITY;
  const double *Gsrc = f->redbuf;
  const double *cs = q.dc;
  int pass = 0;
  for (pass = 1; pass <= 6; ++pass) {
    // ---- R_k = chol(G_k) (pass 1: the equilibrated Gram matrix of the fit, shifted if it must be)
    int32_t info = 0;
    for (int attempt = 0; attempt < 6; ++attempt) {
      hipLaunchKernelGGL(build_scaled_kernel, tgrid, dim3(256), 0, st, Gsrc, P, T, ldm, cs, shift, f->M,
                         (damped && pass == 1) ? q.damp : nullptr);
      HIPCHK(f, potrf_upper(st, f->M, P, ldm, P, f->chol_work, f->info_dev));
      HIPCHK(f, hipMemcpyAsync(&info, f->info_dev, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      HIPCHK(f, hipStreamSynchronize(st));
      if (info == 0) break;
      if (pass > 1) break;                       // Q^T Q of a preconditioned Q cannot fail: give up
      shift = shift == 0.0 ? 1e-13 * (double)(P + N + 1) : shift * 100.0;
    }
    if (info != 0) {
      char b[200];
      snprintf(b, sizeof(b), "J^T J is not positive definite at the solution (qr route, pass %d, pivot %d); covariance undefined", pass, info);
      f->err = b;
      return LSQAMD_ENOTPD;
    }
    double ld = 0.0;
    HIPCHK(f, logdiag_sum(st, f->M, P, ldm, f->scal));
    HIPCHK(f, hipMemcpyAsync(&ld, f->scal, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(f, trtri_upper_to_lower_T(st, f->M, P, ldm, f->chol_work, f->Wl, ldm));          // Wl = R_k^-T
    HIPCHK(f, launch_transpose_sq(st, f->Wl, ldm, q.Ri, ldm, P));                             // Ri = R_k^-1
    if (pass == 1) {
      HIPCHK(f, hipMemcpyAsync(q.Wtot, f->Wl, sizeof(double) * P * ldm, hipMemcpyDeviceToDevice, st));
    } else {
      GemmTN g;   // Wtot <- Wl Wtot  :  C[m][n] = sum_k Ri[k][m] Wtot[k][n]
      g.X = q.Ri; g.ldx = ldm; g.Y = q.Wtot; g.ldy = ldm; g.C = q.T1; g.ldc = ldm;
      g.M = P; g.N = P; g.K = P;
      HIPCHK(f, launch_gemm_tn(st, g));
      HIPCHK(f, hipMemcpyAsync(q.Wtot, q.T1, sizeof(double) * P * ldm, hipMemcpyDeviceToDevice, st));
    }
    HIPCHK(f, hipStreamSynchronize(st));
    logdet_r += ld;
    if (delta < 1e-6) break;                     // G_k was already the identity to 1e-6: R is converged
    if (pass == 6) break;
    // ---- Q_k = Q_(k-1) R_k^-1  (pass 1: Q_0 = J D_c), through a transposed copy (TN GEMM: X is k-major)
    if (N > 0) {
      const double *src = pass == 1 ? f->J : q.Q;
      dim3 grid((unsigned)((P + 63) / 64), (unsigned)((N + 63) / 64));
      hipLaunchKernelGGL(transpose_cscale_kernel, grid, dim3(256), 0, st, src, f->ld, q.T, q.ldn, N, P,
                         pass == 1 ? q.dc : nullptr);
      GemmTN g;
      g.X = q.T; g.ldx = q.ldn; g.Y = q.Ri; g.ldy = ldm; g.C = q.Q; g.ldc = f->ld;
      g.M = N; g.N = P; g.K = P;
      HIPCHK(f, launch_gemm_tn(st, g));
      GemmTN s;   // G_(k+1) = Q^T Q, the J^T J launch on Q
      s.X = q.Q; s.Y = q.Q; s.ldx = s.ldy = f->ld;
      s.C = f->slabs; s.ldc = ldm;
      s.M = P; s.N = P; s.K = N;
      s.upper_only = 1;
      s.splits = f->splits;
      s.split_stride = P * ldm;
      s.work_map = f->syrk_map;
      s.n_work = f->syrk_nwork;
      HIPCHK(f, launch_gemm_tn(st, s));
    } else {
      HIPCHK(f, hipMemsetAsync(f->slabs, 0, sizeof(double) * f->splits * P * ldm, st));
    }
    HIPCHK(f, launch_finalize_pack(st, f->slabs, f->splits, P * ldm, P, ldm, q.Gpk));
    const bool with_prior = f->cfg.has_prior && mine;
    if (with_prior || (damped && mine)) {
      // prior rows W_p D_c R^-1 and damping rows sqrt(mu) D D_c R^-1: their Gram matrix B^T (Lambda + mu D^2) B
      // with B = D_c R^-1 (R^-1 = Wtot^T)
      HIPCHK(f, launch_transpose_sq(st, q.Wtot, ldm, q.Ritot, ldm, P));
      HIPCHK(f, launch_rows_scale_copy(st, q.Ritot, ldm, q.Bm, ldm, P, P, q.dc, nullptr));
      if (with_prior && f->cfg.prior_dense) {
        GemmTN a;   // T1 = Lambda B
        a.X = f->prior_prec; a.ldx = P; a.Y = q.Bm; a.ldy = ldm; a.C = q.T1; a.ldc = ldm;
        a.M = P; a.N = P; a.K = P;
        HIPCHK(f, launch_gemm_tn(st, a));
      } else if (with_prior) {
        HIPCHK(f, launch_rows_scale_copy(st, q.Bm, ldm, q.T1, ldm, P, P, f->prior_prec, nullptr));
      } else {
        HIPCHK(f, launch_rows_scale_copy(st, q.Bm, ldm, q.T1, ldm, P, P, q.damp, nullptr));
      }
      if (with_prior && damped)
        hipLaunchKernelGGL(rows_axpy_kernel, dim3(pb, (unsigned)P), dim3(256), 0, st, q.T1, q.Bm, ldm, P, q.damp);
      GemmTN b;     // Ritot (reused as output) = B^T T1
      b.X = q.Bm; b.ldx = ldm; b.Y = q.T1; b.ldy = ldm; b.C = q.Ritot; b.ldc = ldm;
      b.M = P; b.N = P; b.K = P;
      HIPCHK(f, launch_gemm_tn(st, b));
      hipLaunchKernelGGL(packed_add_dense_kernel, tgrid, dim3(256), 0, st, q.Gpk, P, T, q.Ritot, ldm);
    }
    int rc = do_reduce(f, q.Gpk, f->npk);
    if (rc) return rc;
    hipLaunchKernelGGL(packed_maxdev_kernel, tgrid, dim3(256), 0, st, q.Gpk, P, T, q.part);
    hipLaunchKernelGGL(max_stage2_kernel, dim3(1), dim3(256), 0, st, q.part, (int64_t)tgrid.x * 16, q.scal);
    HIPCHK(f, hipMemcpyAsync(&delta, q.scal, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(f, hipStreamSynchronize(st));
    if (!std::isfinite(delta)) FAIL(f, LSQAMD_ENONFINITE, "solver = qr: the orthogonalised Jacobian is not finite (pass %d)", pass);
    Gsrc = q.Gpk;
    cs = nullptr;
    shift = 0.0;
  }
  f->qr_passes = pass;
  f->qr_delta = delta;
  *logdet_out = logdet_r;
  return 0;
}

int do_covariance_qr(lsqamd_fit *f) {
  const int64_t P = f->P, ldm = f->ldm;
  if (!f->qr_work) FAIL(f, LSQAMD_EINVAL, "solver = qr: call lsqamd_set_qr_work first (lsqamd_qr_work_bytes gives the size)");
  char *base = (char *)f->qr_work;
  const size_t pad = (size_t)((-(intptr_t)base) & 255);
  QrPlan q = qr_plan(f, base + pad);
  if (f->qr_work_bytes < q.bytes + pad) FAIL(f, LSQAMD_ENOMEM, "solver = qr: the work buffer needs %zu bytes", q.bytes + 256);
  Scope sc(f, LSQAMD_T_COVAR);
  hipStream_t st = f->st;
  const unsigned pb = (unsigned)((P + 255) / 256);
  f->have_dense_A = false;   // Wl is reused below
  f->have_cov = true;
  f->logdet = NAN;
  double logdet_r = 0.0;
  {
    const int rc = qr_factor(f, q, 0.0, &logdet_r);
    if (rc) return rc;
  }
  // six passes without reaching |Q^T Q - I| < 1e-6 (shift escalations on a nearly rank-deficient J):
  // the factor is not the R of an orthogonal Q -- say so instead of passing for a cond(J) eps result
  f->cov_inaccurate = !(f->qr_delta < 1e-6);
  // cov = D_c (Wtot^T Wtot) D_c
  GemmTN g;
  g.X = q.Wtot; g.Y = q.Wtot; g.ldx = g.ldy = ldm;
  g.C = f->cov; g.ldc = ldm;
  g.M = P; g.N = P; g.K = P;
  g.upper_only = 1;
  g.xy_lower_tri = 1;
  HIPCHK(f, launch_gemm_tn(st, g));
  HIPCHK(f, launch_symmetrize_from_upper(st, f->cov, P, ldm));
  hipLaunchKernelGGL(sym_scale_kernel, dim3(pb, (unsigned)P), dim3(256), 0, st, f->cov, P, ldm, q.dc);
  double sumlogdc = 0.0;
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, st, q.logdc, P, q.scal);
  HIPCHK(f, hipMemcpyAsync(&sumlogdc, q.scal, sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(f, hipStreamSynchronize(st));
  f->logdet = 2.0 * logdet_r - 2.0 * sumlogdc;
  return 0;
}

// The damped step of an LM trial from an orthogonal factorisation, as gsl's qr solver computes it
// (src/lsqfit/_gsl.pyx:646-647; the reference's default): min |J v - f|^2 + mu |D v|^2 through the factor R of
// [J ; W_prior ; sqrt(mu) D], v = R^-1 R^-T g with g = J^T f formed from J itself.  Called when the Cholesky
// factorisation of the damped normal equations has met a non-positive pivot (cond(J D^-1)^2 beyond 1/eps at a
// small mu): GSL proceeds there, and so does this -- instead of rejecting the trial and inflating mu.
// v -> f->yv[P .. 2P) (where the trial-point kernel reads it), *f->info_dev <- 0.
int solve_damped_qr(lsqamd_fit *f, double mu) {
  const int64_t P = f->P, ldm = f->ldm;
  if (!f->qr_work) return LSQAMD_EUNSUPPORTED;
  char *base = (char *)f->qr_work;
  const size_t pad = (size_t)((-(intptr_t)base) & 255);
  QrPlan q = qr_plan(f, base + pad);
  if (f->qr_work_bytes < q.bytes + pad) return LSQAMD_EUNSUPPORTED;
  Scope sc(f, LSQAMD_T_CHOLESKY);
  hipStream_t st = f->st;
  const unsigned pb = (unsigned)((P + 255) / 256);
  f->have_dense_A = false;
  f->have_cov = false;
  double logdet_r = 0.0;
  const std::string keep = f->err;
  const int rc = qr_factor(f, q, mu, &logdet_r);
  if (rc == LSQAMD_ENOTPD) { f->err = keep; return LSQAMD_ENOTPD; }
  if (rc) return rc;
  if (!(f->qr_delta < 1e-6)) return LSQAMD_ENOTPD;          // not an orthogonal factor: let the caller reject the trial
  const double *g = f->redbuf + f->npk;
  double *y = q.vwork, *t = q.vwork + P;
  hipLaunchKernelGGL(vec_mul_kernel, dim3(pb), dim3(256), 0, st, P, q.dc, g, y);              // y = D_c g
  HIPCHK(f, launch_gemv_rows(st, q.Wtot, ldm, P, P, y, t));                                    // t = R^-T y
  HIPCHK(f, launch_transpose_sq(st, q.Wtot, ldm, q.Ritot, ldm, P));                            // R^-1
  HIPCHK(f, launch_gemv_rows(st, q.Ritot, ldm, P, P, t, y));                                   // y = R^-1 t
  hipLaunchKernelGGL(vec_mul_kernel, dim3(pb), dim3(256), 0, st, P, q.dc, y, f->yv + P);       // v = D_c y
  HIPCHK(f, hipMemsetAsync(f->info_dev, 0, sizeof(int32_t), st));
  return 0;
}

}  // namespace lsqamd_host
