// Batched LM engine: B independent fits of one shape advance in lockstep, all LM state
// on the device, one round of kernels captured in a hipGraph and replayed.
//
// This is the device counterpart of what lsqfit.empbayes_fit does with a Python loop of
// whole fits (src/lsqfit/_extras.py:153-174, BASELINE.json config 5): the fits share the
// model, x and the data whitening (1/sdev for the 1x1 rows, W_b per correlated block), and
// differ in the prior (mean / sdev per fit), the starting point and -- for simulated /
// bootstrap refits (src/lsqfit/__init__.py:1391-1469,1548-1642) -- the data means.  Per fit the GSL trust/lm/nielsen/scaling/
// convergence logic of api.hip is restated as small per-fit device kernels (b_decide,
// b_post); the heavy kernels are the same ones the single-fit path uses, launched with a
// batch dimension: model kernels (grid.y), TN GEMM (grid.z) for J^T J and the Cholesky
// panels/updates, the diagonal-block kernel (one workgroup per fit), back substitution.
//
// One ROUND = one trial step of every still-active fit:
//   build (A_b + mu_b D_b^2 | g_b) -> Cholesky -> back substitution -> x_trial, v.g, |Dv|^2
//   -> residual at x_trial -> chi2_trial -> decide (rho, accept / reject, mu, nu)
//   -> Jacobian, J^T J, J^T f, chi2 at the fit's current x (unchanged if it rejected)
//   -> post (scaling update, niter, convergence tests, retire finished fits).
// Fits that have finished are masked out of every kernel through `active[b]`.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "common.h"
#include "jit.h"

namespace lsqamd_host {   // process-wide recycling of streams and events (api.hip; see fit_state.h)
hipEvent_t event_take();
void event_give(hipEvent_t e, int dev = -1);
hipStream_t stream_take();
void stream_give(hipStream_t s, int dev = -1);
}  // namespace lsqamd_host

namespace lsqamd {

constexpr int TBK = 128;  // packed tile edge (vecops.hip)

__device__ __forceinline__ double bsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

__device__ __forceinline__ double block_sum(double v, double *sh) {
  v = bsum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__device__ __forceinline__ double block_max(double v, double *sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

__device__ __forceinline__ int64_t pk_diag(int64_t j, int64_t T) {  // packed offset of A[j][j]
  const int64_t tm = j / TBK, r = j % TBK;
  return (tm * T - tm * (tm - 1) / 2) * TBK * TBK + r * TBK + r;
}

struct BState {  // per-fit device scalars
  double *mu, *chi2, *chi2t, *vg, *dv2;
  int32_t *nu, *bad, *iter, *nit, *info, *status, *active, *accepted, *enoprog, *cholinfo, *nfev, *njev;
  int32_t *n_active;
  int32_t *moved;   // active and accepted this round: the only fits whose normal equations change
};

// ---- slabs -> packed tiles ---------------------------------------------------------------
__global__ __launch_bounds__(256) void b_finalize_pack_kernel(const double *slabs, int32_t splits,
                                                              int64_t split_stride, int64_t fit_stride,
                                                              int64_t P, int64_t ld, int64_t T, double *apk,
                                                              int64_t apk_stride, const int32_t *active) {
  const int b = blockIdx.z;
  if (!active[b]) return;
  int64_t t = blockIdx.x, tm = 0;
  while (t >= T - tm) { t -= T - tm; ++tm; }
  const int64_t tn = tm + t;
  double *dst = apk + b * apk_stride + (int64_t)blockIdx.x * TBK * TBK;
  const double *src = slabs + b * fit_stride;
  const int c = threadIdx.x & 127;
  for (int rr = (threadIdx.x >> 7); rr < 8; rr += 2) {
    const int r = blockIdx.y * 8 + rr;
    const int64_t i = tm * TBK + r, j = tn * TBK + c;
    double a = 0.0;
    if (i < P && j < P)
      for (int s = 0; s < splits; ++s) a += src[s * split_stride + i * ld + j];
    dst[r * TBK + c] = a;
  }
}

// ---- J^T f, |f|^2 ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void b_colsum_stage1(const double *J, int64_t nrows, int64_t ld,
                                                       int64_t ncols, int64_t rcol, int64_t rpc,
                                                       int64_t j_stride, double *partial,
                                                       int64_t part_stride, const int32_t *active) {
  const int b = blockIdx.z;
  if (!active[b]) return;
  J += b * j_stride;
  partial += b * part_stride;
  const int64_t j = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  const int64_t r0 = (int64_t)blockIdx.y * rpc;
  int64_t r1 = r0 + rpc;
  if (r1 > nrows) r1 = nrows;
  if (j >= ncols) return;
  const bool two = j + 1 < ncols;
  double a0 = 0.0, a1 = 0.0;
  for (int64_t i = r0; i < r1; ++i) {
    const double ri = J[i * ld + rcol];
    a0 += J[i * ld + j] * ri;
    if (two) a1 += J[i * ld + j + 1] * ri;
  }
  partial[(int64_t)blockIdx.y * ncols + j] = a0;
  if (two) partial[(int64_t)blockIdx.y * ncols + j + 1] = a1;
}

__global__ __launch_bounds__(256) void b_colsum_stage2(const double *partial, int64_t nchunks,
                                                       int64_t ncols, int64_t part_stride, double *gvec,
                                                       int64_t g_stride, const int32_t *active) {
  const int b = blockIdx.y;
  if (!active[b]) return;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= ncols) return;
  const double *p = partial + b * part_stride;
  double a = 0.0;
  for (int64_t c = 0; c < nchunks; ++c) a += p[c * ncols + j];
  gvec[b * g_stride + j] = a;
}

// ---- diagonal prior into (packed A, g, chi2); one workgroup per fit ----------------------------
__global__ __launch_bounds__(256) void b_prior_kernel(int64_t P, int64_t T, const double *prec,
                                                      const double *pmean, const double *x, double *apk,
                                                      int64_t apk_stride, double *gvec, int64_t g_stride,
                                                      const int32_t *active) {
  __shared__ double sh[4];
  const int b = blockIdx.x;
  if (!active[b]) return;
  double *A = apk + b * apk_stride, *g = gvec + b * g_stride;
  double c2 = 0.0;
  for (int64_t j = threadIdx.x; j < P; j += 256) {
    const double w = prec[b * P + j], d = x[b * P + j] - pmean[b * P + j];
    A[pk_diag(j, T)] += w;
    g[j] += w * d;
    c2 += w * d * d;
  }
  c2 = block_sum(c2, sh);
  if (threadIdx.x == 0) g[P] += c2;
}

// ---- dense prior precision shared by the fits (per-fit means): A_b += Lambda, g_b += Lambda d_b,
// chi2_b += d_b^T Lambda d_b with d_b = x_b - pbar_b --------------------------------------------------
__global__ __launch_bounds__(256) void b_prior_matrix_dense_kernel(double *apk, int64_t apk_stride, int64_t P,
                                                                   int64_t T, const double *prec,
                                                                   const int32_t *active) {
  const int b = blockIdx.z;
  if (!active[b]) return;
  int64_t t = blockIdx.x, tm = 0;
  while (t >= T - tm) { t -= T - tm; ++tm; }
  const int64_t tn = tm + t;
  double *dst = apk + b * apk_stride + (int64_t)blockIdx.x * TBK * TBK;
  const int c = threadIdx.x & 127;
  for (int rr = (threadIdx.x >> 7); rr < 8; rr += 2) {
    const int r = blockIdx.y * 8 + rr;
    const int64_t i = tm * TBK + r, j = tn * TBK + c;
    if (i < P && j < P) dst[r * TBK + c] += prec[i * P + j];
  }
}

// tvec[b][j] = sum_k Lambda[j][k] (x_b[k] - pbar_b[k]); one wave per row j
__global__ __launch_bounds__(256) void b_prior_vec_dense_kernel(int64_t P, const double *prec,
                                                                const double *pmean, const double *x,
                                                                double *tvec, const int32_t *active) {
  const int b = blockIdx.y;
  if (!active[b]) return;
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= P) return;
  double a = 0.0;
  for (int64_t k = lane; k < P; k += 64) a += prec[j * P + k] * (x[b * P + k] - pmean[b * P + k]);
  a = bsum(a);
  if (lane == 0) tvec[b * P + j] = a;
}

// g_b += t_b ; g_b[P] += d_b . t_b ; one workgroup per fit
__global__ __launch_bounds__(256) void b_prior_apply_kernel(int64_t P, const double *pmean, const double *x,
                                                            const double *tvec, double *gvec, int64_t g_stride,
                                                            const int32_t *active) {
  __shared__ double sh[4];
  const int b = blockIdx.x;
  if (!active[b]) return;
  double *g = gvec + b * g_stride;
  double c2 = 0.0;
  for (int64_t j = threadIdx.x; j < P; j += 256) {
    const double t = tvec[b * P + j];
    g[j] += t;
    c2 += (x[b * P + j] - pmean[b * P + j]) * t;
  }
  c2 = block_sum(c2, sh);
  if (threadIdx.x == 0) g[P] += c2;
}

// ---- trust_init: D, mu, nu, counters -----------------------------------------------------------
__global__ __launch_bounds__(256) void b_init_kernel(int64_t P, int64_t T, const double *apk,
                                                     int64_t apk_stride, const double *gvec,
                                                     int64_t g_stride, double *diag, int scaler,
                                                     BState s) {
  __shared__ double sh[4];
  const int b = blockIdx.x;
  const double *A = apk + b * apk_stride;
  double mx = -1.0;
  for (int64_t j = threadIdx.x; j < P; j += 256) {
    const double a = A[pk_diag(j, T)];
    const double cn = sqrt(a > 0.0 ? a : 0.0);
    const double d = (scaler == LSQAMD_SCALE_LEVENBERG) ? 1.0 : (cn == 0.0 ? 1.0 : cn);
    diag[b * P + j] = d;
    mx = fmax(mx, cn / d);
  }
  mx = block_max(mx, sh);
  if (threadIdx.x == 0) {
    s.mu[b] = 1e-3 * mx * mx;
    s.nu[b] = 2;
    s.chi2[b] = gvec[b * g_stride + P];
    s.bad[b] = 0; s.iter[b] = 0; s.nit[b] = 0; s.info[b] = 0; s.status[b] = -2;
    s.active[b] = 1; s.accepted[b] = 0; s.enoprog[b] = 0; s.nfev[b] = 1; s.njev[b] = 1;
  }
}

// ---- M_b = A_b + mu_b D_b^2, column P = g_b ------------------------------------------------------
__global__ __launch_bounds__(256) void b_build_damped_kernel(const double *apk, int64_t apk_stride,
                                                             int64_t P, int64_t T, int64_t ld,
                                                             const double *mu, const double *diag,
                                                             const double *gvec, int64_t g_stride,
                                                             double *M, int64_t m_stride,
                                                             const int32_t *active, int with_damping) {
  const int b = blockIdx.z;
  if (active && !active[b]) return;
  int64_t t = blockIdx.x, tm = 0;
  while (t >= T - tm) { t -= T - tm; ++tm; }
  const int64_t tn = tm + t;
  const double *src = apk + b * apk_stride + (int64_t)blockIdx.x * TBK * TBK;
  double *Mb = M + b * m_stride;
  const double m = with_damping ? mu[b] : 0.0;
  const int c = threadIdx.x & 127;
  for (int rr = (threadIdx.x >> 7); rr < 8; rr += 2) {
    const int r = blockIdx.y * 8 + rr;
    const int64_t i = tm * TBK + r, j = tn * TBK + c;
    if (i < P && j < P) {
      double v = src[r * TBK + c];
      if (i == j) {
        const double d = diag[b * P + i];
        v += m * d * d;
      }
      Mb[i * ld + j] = v;
    }
    if (tn == T - 1 && c == 0 && i < P && gvec) Mb[i * ld + P] = gvec[b * g_stride + i];
  }
}

// y_b = column P of the factored M_b -> yv_b[0..P)
__global__ __launch_bounds__(256) void b_extract_y_kernel(const double *M, int64_t m_stride, int64_t ld,
                                                          int64_t P, double *yv, int64_t y_stride,
                                                          const int32_t *active) {
  const int b = blockIdx.y;
  if (!active[b]) return;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < P) yv[b * y_stride + j] = M[b * m_stride + j * ld + P];
}

// dx = -v, x_trial = x + dx, v.g, |D v|^2 ; non-finite v marks the factorisation as failed
__global__ __launch_bounds__(256) void b_trial_kernel(int64_t P, const double *yv, int64_t y_stride,
                                                      const double *x, const double *gvec,
                                                      int64_t g_stride, const double *diag, double *dx,
                                                      double *xt, BState s) {
  __shared__ double sh[4];
  const int b = blockIdx.x;
  if (!s.active[b]) return;
  const double *v = yv + b * y_stride + P;
  double vg = 0.0, dv2 = 0.0, bad = 0.0;
  for (int64_t j = threadIdx.x; j < P; j += 256) {
    const double vj = v[j];
    if (!(fabs(vj) < 1.0e300)) bad = 1.0;
    dx[b * P + j] = -vj;
    xt[b * P + j] = x[b * P + j] - vj;
    vg += vj * gvec[b * g_stride + j];
    const double t = diag[b * P + j] * vj;
    dv2 += t * t;
  }
  vg = block_sum(vg, sh);
  dv2 = block_sum(dv2, sh);
  bad = block_sum(bad, sh);
  if (threadIdx.x == 0) {
    s.vg[b] = vg;
    s.dv2[b] = dv2;
    if (bad > 0.0) s.cholinfo[b] = -1;
  }
}

// chi2_trial_b = |r_b|^2 + prior term at x_trial ; two stages
__global__ __launch_bounds__(256) void b_sumsq_stage1(const double *r, int64_t n, int64_t r_stride,
                                                      double *partial, int nparts,
                                                      const int32_t *active) {
  __shared__ double sh[4];
  const int b = blockIdx.y;
  if (!active[b]) return;
  const double *rb = r + b * r_stride;
  double a = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double v = rb[i];
    a += v * v;
  }
  a = block_sum(a, sh);
  if (threadIdx.x == 0) partial[b * nparts + blockIdx.x] = a;
}

__global__ __launch_bounds__(256) void b_sumsq_stage2(const double *partial, int nparts, int64_t P,
                                                      const double *prec, const double *pmean,
                                                      const double *xt, int has_prior, const double *tvec,
                                                      BState s) {
  __shared__ double sh[4];
  const int b = blockIdx.x;
  if (!s.active[b]) return;
  double a = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) a += partial[b * nparts + i];
  if (has_prior)
    for (int64_t j = threadIdx.x; j < P; j += 256) {
      const double d = xt[b * P + j] - pmean[b * P + j];
      a += tvec ? d * tvec[b * P + j] : prec[b * P + j] * d * d;   // tvec = Lambda d (dense prior)
    }
  a = block_sum(a, sh);
  if (threadIdx.x == 0) {
    s.chi2t[b] = a;
    s.nfev[b] += 1;
  }
}

// trust_eval_step + nielsen accept/reject, one thread per fit
__global__ void b_decide_kernel(int B, double factor_up, double factor_down, BState s) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  s.moved[b] = 0;
  if (!s.active[b]) return;
  double rho = -1.0;
  if (s.cholinfo[b] == 0) {
    const double normf = sqrt(s.chi2[b]), normf_t = sqrt(s.chi2t[b]);
    if (normf_t < normf) {
      const double u = normf_t / normf;
      const double actual = 1.0 - u * u;
      const double pred = (s.vg[b] + s.mu[b] * s.dv2[b]) / s.chi2[b];
      rho = pred > 0.0 ? actual / pred : -1.0;
    }
  }
  s.accepted[b] = 0;
  s.enoprog[b] = 0;
  if (rho > 0.0) {
    s.accepted[b] = 1;
    s.moved[b] = 1;
    const double bb = 2.0 * rho - 1.0;
    s.mu[b] *= fmax(0.333333333333333, 1.0 - bb * bb * bb);
    s.nu[b] = 2;
    s.bad[b] = 0;
  } else {
    s.mu[b] *= (double)s.nu[b];
    s.nu[b] <<= 1;
    s.bad[b] += 1;
    if (s.bad[b] > 15) {
      s.enoprog[b] = 1;
      s.bad[b] = 0;
    }
  }
}

// accepted fits: x <- x_trial
__global__ __launch_bounds__(256) void b_commit_kernel(int64_t P, double *x, const double *xt, BState s) {
  const int b = blockIdx.y;
  if (!s.active[b] || !s.accepted[b]) return;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < P) x[b * P + j] = xt[b * P + j];
}

// scaling update, niter, gsl_multifit_nlinear_test, retirement; one workgroup per fit
__global__ __launch_bounds__(256) void b_post_kernel(int64_t P, int64_t T, const double *apk,
                                                     int64_t apk_stride, const double *gvec,
                                                     int64_t g_stride, double *diag, const double *x,
                                                     const double *dx, int scaler, double xtol,
                                                     double gtol, int maxit, BState s) {
  __shared__ double sh[4];
  const int b = blockIdx.x;
  if (!s.active[b]) return;
  const bool acc = s.accepted[b] != 0, enp = s.enoprog[b] != 0;
  if (!acc && !enp) return;  // rejected trial: stay in the inner loop
  if (acc) {
    const double *A = apk + b * apk_stride;
    for (int64_t j = threadIdx.x; j < P; j += 256) {
      const double a = A[pk_diag(j, T)];
      const double cn = sqrt(a > 0.0 ? a : 0.0);
      if (scaler == LSQAMD_SCALE_MORE) diag[b * P + j] = fmax(diag[b * P + j], cn);
      else if (scaler == LSQAMD_SCALE_MARQUARDT) diag[b * P + j] = cn == 0.0 ? 1.0 : cn;
    }
  }
  // convergence tests on (dx, x, g, chi2)
  double notx = 0.0, gn = 0.0;
  for (int64_t j = threadIdx.x; j < P; j += 256) {
    const double xj = x[b * P + j];
    if (!(fabs(dx[b * P + j]) < xtol * xtol + xtol * fabs(xj))) notx = 1.0;
    gn = fmax(gn, fabs(fmax(xj, 1.0) * gvec[b * g_stride + j]));
  }
  notx = block_sum(notx, sh);
  gn = block_max(gn, sh);
  if (threadIdx.x == 0) {
    const double c2 = gvec[b * g_stride + P];
    if (acc) {
      s.chi2[b] = c2;
      s.njev[b] += 1;
    }
    s.nit[b] += 1;
    if (enp && s.iter[b] == 0) {  // no progress on the very first iteration
      s.info[b] = LSQAMD_ENOPROG;
      s.status[b] = LSQAMD_EMAXITER;
      s.active[b] = 0;
      return;
    }
    s.iter[b] += 1;
    int info = 0;
    if (notx == 0.0) info = 1;
    else if (gn <= gtol * fmax(0.5 * s.chi2[b], 1.0)) info = 2;
    if (info) {
      s.info[b] = info;
      s.status[b] = 0;
      s.active[b] = 0;
    } else if (s.iter[b] >= maxit) {
      s.info[b] = 0;
      s.status[b] = LSQAMD_EMAXITER;
      s.active[b] = 0;
    }
  }
}

__global__ void b_count_kernel(int B, BState s) {
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int c = 0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) c += s.active[b] ? 1 : 0;
  atomicAdd(&cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) *s.n_active = cnt;
}

__global__ void b_logdiag_kernel(const double *M, int64_t m_stride, int64_t n, int64_t ld, double *out) {
  __shared__ double sh[4];
  const int b = blockIdx.x;
  double a = 0.0;
  for (int64_t j = threadIdx.x; j < n; j += 256) a += log(M[b * m_stride + j * ld + j]);
  a = block_sum(a, sh);
  if (threadIdx.x == 0) out[b] = 2.0 * a;
}

__global__ __launch_bounds__(256) void b_symmetrize_kernel(double *A, int64_t a_stride, int64_t P,
                                                           int64_t ld) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = blockIdx.y;
  double *Ab = A + (int64_t)blockIdx.z * a_stride;
  if (j >= P || j >= i) return;
  Ab[i * ld + j] = Ab[j * ld + i];
}

}  // namespace lsqamd

// ===============================================================================================
using namespace lsqamd;

namespace {
constexpr int64_t ALIGNB = 256;
inline int64_t rupb(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct CarverB {
  char *base;
  size_t off = 0;
  bool dry;
  CarverB(void *b, bool d) : base((char *)b), dry(d) {}
  template <typename T>
  T *take(int64_t count) {
    const size_t bytes = (size_t)rupb((count < 1 ? 1 : count) * (int64_t)sizeof(T), ALIGNB);
    T *p = dry ? nullptr : reinterpret_cast<T *>(base + off);
    off += bytes;
    return p;
  }
};
}  // namespace

struct lsqamdb_fits {
  lsqamd_config cfg;
  lsqamd_options opt;
  int32_t B = 0;
  hipStream_t user_st = nullptr, st = nullptr;
  int dev = -1;                 // the device that was current at lsqamdb_create (recycled streams / events are filed under it)
  std::string err;
  int64_t N = 0, P = 0, ld = 0, ldm = 0, ncols_aug = 0, npk = 0, T = 0, nblk = 0;
  int32_t splits = 1, nparts = 256, nrparts = 64;
  // shared inputs
  double *x = nullptr, *ymean = nullptr, *wdiag = nullptr;
  int64_t ymean_stride = 0;  // 0 shared, N per-fit
  // correlated data blocks (shared by the fits)
  uint8_t *in_block = nullptr;
  int64_t *blk_row0 = nullptr, *blk_size = nullptr, *blk_woff = nullptr;
  double *wt = nullptr, *Jraw = nullptr, *r_raw = nullptr;
  std::vector<int64_t> h_row0, h_size, h_modes, h_woff;
  std::vector<int32_t> h_tri;
  bool have_blocks = false;
  int32_t *tape = nullptr;
  double *consts = nullptr;
  int32_t n_tape = 0;
  const void *jit = nullptr;   // the tape compiled (jit.hip); null: the forward-mode interpreter kernel
  // per fit
  double *ptvec = nullptr;  // dense prior: Lambda (x - pbar) per fit
  double *pmean = nullptr, *pprec = nullptr, *px = nullptr, *pxt = nullptr, *dx = nullptr, *diag = nullptr;
  double *r = nullptr, *J = nullptr, *slabs = nullptr, *red = nullptr, *M = nullptr, *chol_work = nullptr;
  double *yv = nullptr, *partial = nullptr, *spart = nullptr, *Wl = nullptr, *cov = nullptr, *logdet = nullptr;
  int32_t *syrk_map = nullptr;
  int32_t syrk_nwork = 0;
  BState s;
  bool have_x = false, have_data = false, have_prior = false, have_tape = false, ran = false, have_cov = false;
  hipGraph_t graph = nullptr;
  hipGraphExec_t gexec = nullptr;
  int32_t graph_used = 0, rounds = 0;
  bool one_launch = false, have_cov_from_run = false;   // the last run was one launch (lsqamd_jit_lmb) / it formed the covariances too
  // phase timers (lsqamdb_timing_enable): HIP events around the J^T J launch and the batched factorisation of every round;
  // rounds then run eagerly (events are not recorded into a captured graph)
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> tm_syrk, tm_chol;
  ~lsqamdb_fits() {
    for (auto *v : {&tm_syrk, &tm_chol})
      for (auto &pr : *v) {
        lsqamd_host::event_give(pr.first, dev);
        lsqamd_host::event_give(pr.second, dev);
      }
  }
};

namespace {

#define BFAIL(f, code, ...)                \
  do {                                     \
    char _b[512];                          \
    snprintf(_b, sizeof(_b), __VA_ARGS__); \
    (f)->err = _b;                         \
    return (code);                         \
  } while (0)
#define BHIP(f, expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) BFAIL(f, LSQAMD_EHIP, "%s: %s", #expr, hipGetErrorString(_e));  \
  } while (0)

// Synchronous copies go through the engine's OWN (non-blocking) stream, never the legacy default stream: a legacy-stream
// operation synchronises with every blocking stream of the process and -- on this runtime -- fails with
// hipErrorStreamCaptureImplicit while ANY thread of the process is capturing a graph, invalidating that thread's capture
// too (found by tests/test_gpu_threads.py: a batch's setters against another handle's step-graph capture).
hipError_t copy_sync(lsqamdb_fits *f, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  if (bytes == 0) return hipSuccess;
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, f->st);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(f->st);
}

hipError_t copy2d_sync(lsqamdb_fits *f, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height,
                       hipMemcpyKind kind) {
  hipError_t e = hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, f->st);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(f->st);
}

size_t carve_b(lsqamdb_fits *f, void *ws, bool dry) {
  const lsqamd_config &c = f->cfg;
  const int64_t N = c.n_data, P = c.n_param, B = f->B;
  f->N = N; f->P = P;
  f->ld = rupb(P + 1, 16);
  f->ldm = (P % 128 == 0) ? P + 128 : rupb(P + 1, 16);
  f->ncols_aug = (P % 128 == 0) ? P + 128 : P + 1;
  f->npk = packed_doubles(P);
  f->T = (P + 127) / 128;
  f->nblk = f->T;
  {  // split-K: enough workgroups over the whole batch
    const int64_t tiles = f->T * (f->T + 1) / 2 * B;
    int64_t s = (4096 + tiles - 1) / tiles;
    const int64_t maxs = N / 1024 > 1 ? N / 1024 : 1;
    if (s > maxs) s = maxs;
    if (s > 8) s = 8;
    if (s < 1) s = 1;
    if (const char *e = getenv("LSQAMD_BATCH_SPLITS")) {  // developer knob
      const int v = atoi(e);
      if (v >= 1 && v <= 16) s = v;
    }
    f->splits = (int32_t)s;
  }
  f->nparts = 64;
  f->nrparts = 32;
  CarverB cv(ws, dry);
  f->x = cv.take<double>(N * (c.n_x > 0 ? c.n_x : 1));
  f->ymean = cv.take<double>(B * N);
  f->wdiag = cv.take<double>(N);
  f->in_block = cv.take<uint8_t>(N);
  f->blk_row0 = cv.take<int64_t>(c.n_blocks);
  f->blk_size = cv.take<int64_t>(c.n_blocks);
  f->blk_woff = cv.take<int64_t>(c.n_blocks);
  f->wt = cv.take<double>(c.sum_block_sq);
  f->Jraw = cv.take<double>(c.n_blocks > 0 ? B * N * f->ld : 1);
  f->r_raw = cv.take<double>(c.n_blocks > 0 ? B * N : 1);
  f->tape = cv.take<int32_t>(1024);
  f->consts = cv.take<double>(256);
  f->pmean = cv.take<double>(B * P);
  f->pprec = cv.take<double>(c.prior_dense ? P * P : B * P);
  f->ptvec = cv.take<double>(c.prior_dense ? B * P : 1);
  f->px = cv.take<double>(B * P);
  f->pxt = cv.take<double>(B * P);
  f->dx = cv.take<double>(B * P);
  f->diag = cv.take<double>(B * P);
  f->r = cv.take<double>(B * N);
  f->J = cv.take<double>(B * N * f->ld);
  f->slabs = cv.take<double>(B * f->splits * P * f->ldm);
  f->red = cv.take<double>(B * (f->npk + P + 1));
  f->M = cv.take<double>(B * P * f->ldm);
  f->chol_work = cv.take<double>(B * f->nblk * 128 * 128);
  f->yv = cv.take<double>(B * 2 * P);
  f->partial = cv.take<double>(B * f->nparts * (P + 1));
  f->spart = cv.take<double>(B * f->nrparts);
  f->Wl = cv.take<double>(B * P * f->ldm);
  f->cov = cv.take<double>(B * P * f->ldm);
  f->logdet = cv.take<double>(B);
  f->s.mu = cv.take<double>(B); f->s.chi2 = cv.take<double>(B); f->s.chi2t = cv.take<double>(B);
  f->s.vg = cv.take<double>(B); f->s.dv2 = cv.take<double>(B);
  f->s.nu = cv.take<int32_t>(B); f->s.bad = cv.take<int32_t>(B); f->s.iter = cv.take<int32_t>(B);
  f->s.nit = cv.take<int32_t>(B); f->s.info = cv.take<int32_t>(B); f->s.status = cv.take<int32_t>(B);
  f->s.active = cv.take<int32_t>(B); f->s.accepted = cv.take<int32_t>(B); f->s.enoprog = cv.take<int32_t>(B);
  f->s.cholinfo = cv.take<int32_t>(B); f->s.nfev = cv.take<int32_t>(B); f->s.njev = cv.take<int32_t>(B);
  f->s.n_active = cv.take<int32_t>(4);
  f->s.moved = cv.take<int32_t>(B);
  f->syrk_nwork = (int32_t)syrk_work_count(P, f->splits);
  f->syrk_map = cv.take<int32_t>(4 * (int64_t)f->syrk_nwork);
  return cv.off;
}

int check_cfg_b(const lsqamd_config *c, int32_t B) {
  if (!c || c->abi_version != LSQAMD_ABI_VERSION || B < 1) return LSQAMD_EINVAL;
  if (c->n_data < 1 || c->n_param < 1) return LSQAMD_EINVAL;
  if (c->n_blocks < 0) return LSQAMD_EUNSUPPORTED;
  if (c->model < LSQAMD_MODEL_COSMIX || c->model > LSQAMD_MODEL_IDENTITY) return LSQAMD_EINVAL;
  if (c->model == LSQAMD_MODEL_TAPE && c->n_param > LSQAMD_TAPE_MAX_PARAM) return LSQAMD_EINVAL;
  if ((c->model == LSQAMD_MODEL_COSMIX || c->model == LSQAMD_MODEL_MULTIEXP) && (c->n_param & 1))
    return LSQAMD_EINVAL;
  return 0;
}

ModelArgs model_args_b(const lsqamdb_fits *f, const double *p) {
  ModelArgs m;
  m.model = f->cfg.model;
  m.n_data = f->N; m.n_param = f->P;
  m.n_x = f->cfg.n_x > 0 ? f->cfg.n_x : 1;
  m.x = f->x; m.ymean = f->ymean; m.wdiag = f->wdiag;
  m.ymean_stride = f->ymean_stride;
  m.in_block = f->cfg.n_blocks > 0 ? f->in_block : nullptr;
  m.p = p; m.tape = f->tape; m.n_tape = f->n_tape; m.consts = f->consts;
  m.n_batch = f->B; m.p_stride = f->P; m.batch_active = f->s.active;
  m.jit = f->jit;
  return m;
}

// Jacobian, J^T J (packed), J^T f, chi2 for every fit flagged in `mask` at its current x (inside a
// round: the fits whose trial was accepted -- a rejected trial leaves x, J and the sums as they are)
int normal_all(lsqamdb_fits *f, const int32_t *mask) {
  const int64_t P = f->P, N = f->N, B = f->B;
  ModelArgs m = model_args_b(f, f->px);
  m.batch_active = mask;
  m.out_stride = N * f->ld;
  BHIP(f, launch_jacobian_ex(f->st, m, f->J, f->cfg.n_blocks > 0 ? f->Jraw : nullptr, f->ld));
  // block rows: J_b <- W_b . Jraw_b for every fit (W shared: X stride 0, batch = fits)
  for (size_t k = 0; k < f->h_size.size(); ++k) {
    const int64_t Bk = f->h_size[k];
    GemmTN w;
    w.X = f->wt + f->h_woff[k]; w.ldx = Bk; w.sx = 0;
    w.Y = f->Jraw + f->h_row0[k] * f->ld; w.ldy = f->ld; w.sy = N * f->ld;
    w.C = f->J + f->h_row0[k] * f->ld; w.ldc = f->ld; w.sc = N * f->ld;
    w.M = Bk; w.N = P + 1; w.K = Bk;
    w.x_upper_tri = f->h_tri[k];
    w.batch = (int32_t)B; w.batch_active = mask;
    BHIP(f, launch_gemm_tn(f->st, w));
  }
  GemmTN g;
  g.X = f->J; g.Y = f->J; g.ldx = g.ldy = f->ld;
  g.sx = g.sy = N * f->ld;
  g.C = f->slabs; g.ldc = f->ldm; g.sc = (int64_t)f->splits * P * f->ldm;
  g.M = P; g.N = P; g.K = N;
  g.upper_only = 1;
  g.splits = f->splits;
  g.split_stride = P * f->ldm;
  g.work_map = f->syrk_map; g.n_work = f->syrk_nwork;
  g.batch = (int32_t)B; g.batch_active = mask;
  // J^T f and chi2 from the diagonal tiles of the same launch when it is the kernel that can (P a multiple of 128):
  // partial[(b * splits + split)][P + 1]   (splits <= 16 <= nparts)
  g.colsum_out = f->partial; g.colsum_ld = P + 1; g.colsum_rcol = P;
  const bool syrk_colsum = gemm_tn_fuses_colsum(g);
  if (!syrk_colsum) g.colsum_out = nullptr;
  if (f->timing) {
    hipEvent_t a = lsqamd_host::event_take(), b = lsqamd_host::event_take();
    (void)hipEventRecord(a, f->st);
    BHIP(f, launch_gemm_tn(f->st, g));
    (void)hipEventRecord(b, f->st);
    f->tm_syrk.emplace_back(a, b);
  } else {
    BHIP(f, launch_gemm_tn(f->st, g));
  }
  const int64_t red_stride = f->npk + P + 1;
  hipLaunchKernelGGL(b_finalize_pack_kernel, dim3((unsigned)(f->T * (f->T + 1) / 2), 16, (unsigned)B),
                     dim3(256), 0, f->st, f->slabs, f->splits, P * f->ldm, (int64_t)f->splits * P * f->ldm, P,
                     f->ldm, f->T, f->red, red_stride, mask);
  if (syrk_colsum) {
    hipLaunchKernelGGL(b_colsum_stage2, dim3((unsigned)((P + 1 + 255) / 256), (unsigned)B), dim3(256), 0, f->st,
                       f->partial, (int64_t)f->splits, P + 1, (int64_t)f->splits * (P + 1), f->red + f->npk, red_stride,
                       mask);
  } else {
    int64_t nchunks = f->nparts;
    if (nchunks > N) nchunks = N;
    const int64_t rpc = (N + nchunks - 1) / nchunks;
    nchunks = (N + rpc - 1) / rpc;
    hipLaunchKernelGGL(b_colsum_stage1, dim3((unsigned)((P + 1 + 511) / 512), (unsigned)nchunks, (unsigned)B),
                       dim3(256), 0, f->st, f->J, N, f->ld, P + 1, P, rpc, N * f->ld, f->partial,
                       (int64_t)f->nparts * (P + 1), mask);
    hipLaunchKernelGGL(b_colsum_stage2, dim3((unsigned)((P + 1 + 255) / 256), (unsigned)B), dim3(256), 0, f->st,
                       f->partial, nchunks, P + 1, (int64_t)f->nparts * (P + 1), f->red + f->npk, red_stride,
                       mask);
  }
  if (f->cfg.has_prior && f->cfg.prior_dense) {
    hipLaunchKernelGGL(b_prior_matrix_dense_kernel, dim3((unsigned)(f->T * (f->T + 1) / 2), 16, (unsigned)B),
                       dim3(256), 0, f->st, f->red, red_stride, P, f->T, f->pprec, mask);
    hipLaunchKernelGGL(b_prior_vec_dense_kernel, dim3((unsigned)((P + 3) / 4), (unsigned)B), dim3(256), 0, f->st,
                       P, f->pprec, f->pmean, f->px, f->ptvec, mask);
    hipLaunchKernelGGL(b_prior_apply_kernel, dim3((unsigned)B), dim3(256), 0, f->st, P, f->pmean, f->px, f->ptvec,
                       f->red + f->npk, red_stride, mask);
  } else if (f->cfg.has_prior)
    hipLaunchKernelGGL(b_prior_kernel, dim3((unsigned)B), dim3(256), 0, f->st, P, f->T, f->pprec, f->pmean,
                       f->px, f->red, red_stride, f->red + f->npk, red_stride, mask);
  BHIP(f, hipGetLastError());
  return 0;
}

int round_all(lsqamdb_fits *f) {
  const int64_t P = f->P, N = f->N, B = f->B;
  const int64_t red_stride = f->npk + P + 1, m_stride = P * f->ldm, w_stride = f->nblk * 128 * 128;
  const unsigned ntile = (unsigned)(f->T * (f->T + 1) / 2);
  hipLaunchKernelGGL(b_build_damped_kernel, dim3(ntile, 16, (unsigned)B), dim3(256), 0, f->st, f->red,
                     red_stride, P, f->T, f->ldm, f->s.mu, f->diag, f->red + f->npk, red_stride, f->M, m_stride,
                     f->s.active, 1);
  if (f->timing) {
    hipEvent_t a = lsqamd_host::event_take(), b = lsqamd_host::event_take();
    (void)hipEventRecord(a, f->st);
    BHIP(f, potrf_upper_batched(f->st, f->M, P, f->ldm, f->ncols_aug, f->chol_work, f->s.cholinfo, (int32_t)B,
                                m_stride, w_stride, f->s.active));
    (void)hipEventRecord(b, f->st);
    f->tm_chol.emplace_back(a, b);
  } else {
    BHIP(f, potrf_upper_batched(f->st, f->M, P, f->ldm, f->ncols_aug, f->chol_work, f->s.cholinfo, (int32_t)B,
                                m_stride, w_stride, f->s.active));
  }
  hipLaunchKernelGGL(b_extract_y_kernel, dim3((unsigned)((P + 255) / 256), (unsigned)B), dim3(256), 0, f->st,
                     f->M, m_stride, f->ldm, P, f->yv, 2 * P, f->s.active);
  BHIP(f, backsolve_upper_batched(f->st, f->M, P, f->ldm, f->chol_work, f->yv, (int32_t)B, m_stride, w_stride,
                                  2 * P, f->s.active));
  hipLaunchKernelGGL(b_trial_kernel, dim3((unsigned)B), dim3(256), 0, f->st, P, f->yv, 2 * P, f->px,
                     f->red + f->npk, red_stride, f->diag, f->dx, f->pxt, f->s);
  ModelArgs m = model_args_b(f, f->pxt);
  m.out_stride = N;
  BHIP(f, launch_residual_ex(f->st, m, f->r, f->cfg.n_blocks > 0 ? f->r_raw : nullptr));
  if (f->cfg.n_blocks > 0)
    BHIP(f, launch_block_whiten_vec(f->st, f->wt, f->blk_row0, f->blk_size, f->blk_woff, f->cfg.n_blocks,
                                    f->cfg.max_block, f->r_raw, f->r, (int32_t)B, N, f->s.active));
  hipLaunchKernelGGL(b_sumsq_stage1, dim3((unsigned)f->nrparts, (unsigned)B), dim3(256), 0, f->st, f->r, N, N,
                     f->spart, f->nrparts, f->s.active);
  const bool dense_prior = f->cfg.has_prior && f->cfg.prior_dense;
  if (dense_prior)
    hipLaunchKernelGGL(b_prior_vec_dense_kernel, dim3((unsigned)((P + 3) / 4), (unsigned)B), dim3(256), 0, f->st,
                       P, f->pprec, f->pmean, f->pxt, f->ptvec, f->s.active);
  hipLaunchKernelGGL(b_sumsq_stage2, dim3((unsigned)B), dim3(256), 0, f->st, f->spart, f->nrparts, P, f->pprec,
                     f->pmean, f->pxt, f->cfg.has_prior, dense_prior ? f->ptvec : nullptr, f->s);
  hipLaunchKernelGGL(b_decide_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, f->st, (int)B,
                     f->opt.factor_up, f->opt.factor_down, f->s);
  hipLaunchKernelGGL(b_commit_kernel, dim3((unsigned)((P + 255) / 256), (unsigned)B), dim3(256), 0, f->st, P,
                     f->px, f->pxt, f->s);
  int rc = normal_all(f, f->s.moved);
  if (rc) return rc;
  hipLaunchKernelGGL(b_post_kernel, dim3((unsigned)B), dim3(256), 0, f->st, P, f->T, f->red, red_stride,
                     f->red + f->npk, red_stride, f->diag, f->px, f->dx, f->opt.scaler, f->opt.xtol, f->opt.gtol,
                     f->opt.maxit, f->s);
  hipLaunchKernelGGL(b_count_kernel, dim3(1), dim3(256), 0, f->st, (int)B, f->s);
  BHIP(f, hipGetLastError());
  return 0;
}

}  // namespace

extern "C" {

size_t lsqamdb_workspace_bytes(const lsqamd_config *cfg, int32_t n_fits) try {
  if (check_cfg_b(cfg, n_fits) != 0) return 0;
  lsqamdb_fits tmp;
  tmp.cfg = *cfg;
  tmp.B = n_fits;
  return carve_b(&tmp, nullptr, true);
} LSQAMD_ABI_CATCH((void)lsqamd::abi_exception(nullptr); return 0;)

int lsqamdb_create(const lsqamd_config *cfg, int32_t n_fits, void *dev_workspace, size_t workspace_bytes,
                   void *stream, lsqamdb_fits **out) try {
  if (!out) return LSQAMD_EINVAL;
  *out = nullptr;
  const int rc = check_cfg_b(cfg, n_fits);
  if (rc) return rc;
  if (!dev_workspace || (reinterpret_cast<uintptr_t>(dev_workspace) & 255)) return LSQAMD_EINVAL;
  lsqamdb_fits *f = new (std::nothrow) lsqamdb_fits;
  if (!f) return LSQAMD_ENOMEM;
  f->cfg = *cfg;
  f->B = n_fits;
  f->user_st = reinterpret_cast<hipStream_t>(stream);
  if (carve_b(f, dev_workspace, true) > workspace_bytes) { delete f; return LSQAMD_ENOMEM; }
  carve_b(f, dev_workspace, false);
  // own (capturable) stream: the caller's may be the legacy default stream
  (void)hipGetDevice(&f->dev);
  f->st = lsqamd_host::stream_take();
  if (!f->st) { delete f; return LSQAMD_EHIP; }
  f->opt.xtol = 1e-8; f->opt.gtol = 1e-10; f->opt.ftol = 1e-10;
  f->opt.maxit = 1000; f->opt.scaler = LSQAMD_SCALE_MORE; f->opt.solver = LSQAMD_SOLVER_CHOLESKY;
  f->opt.factor_up = 3.0; f->opt.factor_down = 2.0; f->opt.trs = LSQAMD_TRS_LM; f->opt.avmax = 0.75;
  std::vector<int32_t> wm(4 * (size_t)f->syrk_nwork);
  syrk_work_fill(f->P, f->splits, wm.data());
  if (hipMemcpyAsync(f->syrk_map, wm.data(), wm.size() * sizeof(int32_t), hipMemcpyHostToDevice, f->st) != hipSuccess ||
      hipMemsetAsync(f->M, 0, sizeof(double) * (size_t)(f->B * f->P * f->ldm), f->st) != hipSuccess ||
      hipStreamSynchronize(f->st) != hipSuccess) {
    (void)hipStreamSynchronize(f->st);
    lsqamd_host::stream_give(f->st, f->dev);
    delete f;
    return LSQAMD_EHIP;
  }
  *out = f;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)

int lsqamdb_destroy(lsqamdb_fits *f) try {
  if (!f) return 0;
  (void)hipStreamSynchronize(f->st);
  if (f->gexec) (void)hipGraphExecDestroy(f->gexec);
  if (f->graph) (void)hipGraphDestroy(f->graph);
  lsqamd_host::stream_give(f->st, f->dev);       // (synchronised above)
  if (f->jit) lsqamd_jit::release(static_cast<const lsqamd_jit::Kernel *>(f->jit));
  delete f;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)

const char *lsqamdb_last_error(const lsqamdb_fits *f) { return f ? f->err.c_str() : "null handle"; }

int lsqamdb_set_x(lsqamdb_fits *f, const double *x, int64_t n_rows, int32_t n_x) try {
  if (!f) return LSQAMD_EINVAL;
  if (!x || n_rows != f->N || n_x != (f->cfg.n_x > 0 ? f->cfg.n_x : 1)) BFAIL(f, LSQAMD_EINVAL, "set_x: shape");
  BHIP(f, copy_sync(f, f->x, x, sizeof(double) * n_rows * n_x, hipMemcpyHostToDevice));
  f->have_x = true;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamdb_set_tape(lsqamdb_fits *f, const int32_t *code, int32_t n_code, const double *consts, int32_t n_consts) try {
  if (!f) return LSQAMD_EINVAL;
  if (!code || n_code < 1 || n_code > LSQAMD_TAPE_MAX_CODE || n_consts < 0 || n_consts > 1024 || (n_consts > 0 && !consts))
    BFAIL(f, LSQAMD_EINVAL, "set_tape: sizes");
  {   // stack discipline and operand ranges (the compiled route indexes by them)
    int sp = 0;
    for (int t = 0; t < n_code; ++t) {
      const int op = code[t] & 0xff, arg = code[t] >> 8;
      bool ok = true;
      if (op == LSQAMD_OP_CONST) { ok = arg >= 0 && arg < n_consts; ++sp; }
      else if (op == LSQAMD_OP_X) { ok = arg >= 0 && arg < (f->cfg.n_x > 0 ? f->cfg.n_x : 1); ++sp; }
      else if (op == LSQAMD_OP_P) { ok = arg >= 0 && arg < f->P; ++sp; }
      else if (op >= LSQAMD_OP_ADD && op <= LSQAMD_OP_POW) { ok = sp >= 2; --sp; }
      else if (op >= LSQAMD_OP_NEG && op <= LSQAMD_OP_LAST) ok = sp >= 1;
      else ok = false;
      if (!ok || sp > LSQAMD_TAPE_MAX_STACK) BFAIL(f, LSQAMD_EINVAL, "set_tape: malformed tape at instruction %d", t);
    }
    if (sp != 1) BFAIL(f, LSQAMD_EINVAL, "set_tape: the tape must leave exactly one value");
  }
  // the formula compiled (jit.hip): one launch per evaluation with the fits as blockIdx.y, whatever P is -- the
  // interpreter route (forward mode, ceil(P / 16) passes per row) remains for tapes of <= 1024 instructions when
  // hiprtc is absent or the generator declines
  std::string why;
  if (f->jit) lsqamd_jit::release(static_cast<const lsqamd_jit::Kernel *>(f->jit));
  f->jit = lsqamd_jit::compile_tape(code, n_code, consts, n_consts, (int)f->P, f->cfg.n_x > 0 ? f->cfg.n_x : 1, why);
  if (!f->jit && (n_code > 1024 || n_consts > 256))
    BFAIL(f, LSQAMD_EUNSUPPORTED, "set_tape: %d instructions need the compiled route, which is unavailable (%s)", n_code, why.c_str());
  if (n_code <= 1024 && n_consts <= 256) {
    BHIP(f, copy_sync(f, f->tape, code, sizeof(int32_t) * n_code, hipMemcpyHostToDevice));
    if (n_consts > 0) BHIP(f, copy_sync(f, f->consts, consts, sizeof(double) * n_consts, hipMemcpyHostToDevice));
  }
  f->n_tape = n_code;
  f->have_tape = true;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

/* shared data: ymean[N], wdiag[N] = 1/sdev of the 1x1 rows */
int lsqamdb_set_data(lsqamdb_fits *f, const double *ymean, const double *wdiag) try {
  if (!f || !ymean || !wdiag) return LSQAMD_EINVAL;
  BHIP(f, copy_sync(f, f->ymean, ymean, sizeof(double) * f->N, hipMemcpyHostToDevice));
  BHIP(f, copy_sync(f, f->wdiag, wdiag, sizeof(double) * f->N, hipMemcpyHostToDevice));
  f->ymean_stride = 0;
  f->have_data = true;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

/* per-fit data means ymean[B*N] (simulated / bootstrap copies of one data set: same
 * covariance, new means); the whitening set by lsqamdb_set_data / _set_blocks is kept */
int lsqamdb_set_data_means(lsqamdb_fits *f, const double *ymean) try {
  if (!f || !ymean) return LSQAMD_EINVAL;
  if (!f->have_data) BFAIL(f, LSQAMD_EINVAL, "set_data_means: call lsqamdb_set_data first");
  BHIP(f, copy_sync(f, f->ymean, ymean, sizeof(double) * f->B * f->N, hipMemcpyHostToDevice));
  f->ymean_stride = f->N;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

/* correlated data blocks shared by all fits; same layout as lsqamd_set_data */
int lsqamdb_set_blocks(lsqamdb_fits *f, int32_t n_blocks, const int64_t *row0, const int64_t *size,
                       const int64_t *modes, const int32_t *tri, const double *wt) try {
  if (!f) return LSQAMD_EINVAL;
  if (n_blocks != f->cfg.n_blocks) BFAIL(f, LSQAMD_EINVAL, "set_blocks: n_blocks differs from the config");
  if (n_blocks == 0) { f->have_blocks = true; return 0; }
  if (!row0 || !size || !modes || !tri || !wt) return LSQAMD_EINVAL;
  std::vector<uint8_t> inb((size_t)f->N, 0);
  std::vector<int64_t> woff((size_t)n_blocks);
  f->h_row0.assign(row0, row0 + n_blocks);
  f->h_size.assign(size, size + n_blocks);
  f->h_modes.assign(modes, modes + n_blocks);
  f->h_tri.assign(tri, tri + n_blocks);
  int64_t off = 0;
  for (int32_t k = 0; k < n_blocks; ++k) {
    if (size[k] < 1 || size[k] > f->cfg.max_block || row0[k] < 0 || row0[k] + size[k] > f->N ||
        modes[k] < 0 || modes[k] > size[k])
      BFAIL(f, LSQAMD_EINVAL, "set_blocks: block %d out of range", k);
    for (int64_t i = 0; i < size[k]; ++i) {
      if (inb[(size_t)(row0[k] + i)]) BFAIL(f, LSQAMD_EINVAL, "set_blocks: blocks overlap");
      inb[(size_t)(row0[k] + i)] = 1;
    }
    woff[(size_t)k] = off;
    off += size[k] * size[k];
  }
  if (off != f->cfg.sum_block_sq) BFAIL(f, LSQAMD_EINVAL, "set_blocks: sum of squares differs from the config");
  f->h_woff = woff;
  BHIP(f, copy_sync(f, f->in_block, inb.data(), inb.size(), hipMemcpyHostToDevice));
  BHIP(f, copy_sync(f, f->blk_row0, row0, sizeof(int64_t) * n_blocks, hipMemcpyHostToDevice));
  BHIP(f, copy_sync(f, f->blk_size, size, sizeof(int64_t) * n_blocks, hipMemcpyHostToDevice));
  BHIP(f, copy_sync(f, f->blk_woff, woff.data(), sizeof(int64_t) * n_blocks, hipMemcpyHostToDevice));
  BHIP(f, copy_sync(f, f->wt, wt, sizeof(double) * off, hipMemcpyDefault));   // host or device source
  f->have_blocks = true;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

/* per-fit priors: mean[B*P]; prec[B*P] = 1/sdev^2 per fit (diagonal), or -- cfg.prior_dense -- ONE dense
 * P x P precision shared by all fits (copies of a fit differ in their prior means, not its covariance) */
int lsqamdb_set_priors(lsqamdb_fits *f, const double *mean, const double *prec) try {
  if (!f || !mean || !prec) return LSQAMD_EINVAL;
  if (!f->cfg.has_prior) BFAIL(f, LSQAMD_EINVAL, "set_priors: config has no prior");
  BHIP(f, copy_sync(f, f->pmean, mean, sizeof(double) * f->B * f->P, hipMemcpyHostToDevice));
  BHIP(f, copy_sync(f, f->pprec, prec, sizeof(double) * (f->cfg.prior_dense ? f->P * f->P : f->B * f->P),
                    hipMemcpyDefault));   // host or device source
  f->have_prior = true;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamdb_set_options(lsqamdb_fits *f, const lsqamd_options *opt) try {
  if (!f || !opt) return LSQAMD_EINVAL;
  if (opt->xtol < 0 || opt->gtol < 0 || opt->maxit < 0 || opt->scaler < 0 || opt->scaler > LSQAMD_SCALE_MARQUARDT)
    BFAIL(f, LSQAMD_EINVAL, "set_options: bad value");
  if (opt->trs != LSQAMD_TRS_LM) BFAIL(f, LSQAMD_EUNSUPPORTED, "set_options: the batched engine runs alg='lm' only");
  f->opt = *opt;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

/* run all fits from p0[B*P]; summaries[B] (may be NULL).  use_graph != 0: capture one round in a
 * hipGraph after the first eager round and replay it. */
int lsqamdb_run(lsqamdb_fits *f, const double *p0, lsqamd_summary *summaries, int32_t use_graph) try {
  if (!f || !p0) return LSQAMD_EINVAL;
  if (!f->have_data || (f->cfg.has_prior && !f->have_prior) || (f->cfg.n_blocks > 0 && !f->have_blocks) ||
      (f->cfg.model != LSQAMD_MODEL_IDENTITY && !f->have_x) || (f->cfg.model == LSQAMD_MODEL_TAPE && !f->have_tape))
    BFAIL(f, LSQAMD_EINVAL, "run: inputs missing");
  const int64_t P = f->P, B = f->B;
  if (f->timing) use_graph = 0;      // (events around single launches: eager rounds)
  // inputs the caller made on ITS stream (device-resident weights) are complete before the engine's stream reads them.  A null
  // stream is not waited for: synchronising the legacy stream from here would collide with any graph capture another
  // thread has open (see copy_sync); a caller working on the legacy stream orders its own work (include/lsqfit_amd.h)
  if (f->user_st) (void)hipStreamSynchronize(f->user_st);
  hipEvent_t e0 = lsqamd_host::event_take(), e1 = lsqamd_host::event_take();
  (void)hipEventRecord(e0, f->st);
  BHIP(f, hipMemcpyAsync(f->px, p0, sizeof(double) * B * P, hipMemcpyHostToDevice, f->st));
  {  // everything active for the initial evaluation
    std::vector<int32_t> ones((size_t)B, 1);
    BHIP(f, hipMemcpyAsync(f->s.active, ones.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice, f->st));
    BHIP(f, hipStreamSynchronize(f->st));
  }
  // Small fits (a compiled formula, <= 12 parameters, <= 4096 uncorrelated rows or <= 256 with covariance blocks): every
  // fit of the batch is ONE workgroup of ONE launch (jit.hip lsqamd_jit_lmb -- the loop of the single-fit kernel, api.hip
  // run_one_launch) instead of ~20 lockstep launches per round.  Any irregular fit sends the whole batch through the
  // lockstep engine below.  LSQAMD_ONE_LAUNCH_FIT=0 disables.
  f->one_launch = false;
  {
    const char *e = getenv("LSQAMD_ONE_LAUNCH_FIT");
    const lsqamd_jit::Kernel *k = static_cast<const lsqamd_jit::Kernel *>(f->jit);
    const bool rows_ok = f->N <= lsqamd_jit::fit_row_limit(k, f->cfg.n_blocks != 0) && f->cfg.n_blocks <= 64;
    if (!(e && e[0] == '0') && lsqamd_jit::has_batch_fit_kernel(k) && P <= lsqamd_jit::FIT_MAX_P && f->N >= 1 && rows_ok &&
        f->opt.maxit >= 1 && f->opt.maxit <= 20000 && f->opt.trs == LSQAMD_TRS_LM) {
      const int64_t red_stride = f->npk + P + 1, w_stride = f->nblk * 128 * 128;
      lsqamd_jit::FitArgs a;
      a.x = f->x; a.ymean = f->ymean; a.wdiag = f->wdiag; a.n_data = f->N;
      a.in_block = f->in_block; a.wt = f->wt;
      a.blk_row0 = reinterpret_cast<const long long *>(f->blk_row0); a.blk_size = reinterpret_cast<const long long *>(f->blk_size);
      a.blk_woff = reinterpret_cast<const long long *>(f->blk_woff); a.n_blocks = f->cfg.n_blocks;
      a.p0 = f->px; a.p = f->px; a.p_trial = f->pxt; a.dscale = f->diag; a.apk = f->red; a.gvec = f->red + f->npk;
      a.v_out = f->dx; a.coln2 = nullptr; a.st = nullptr;
      a.prior_prec = f->cfg.has_prior ? f->pprec : nullptr;
      a.prior_mean = f->cfg.has_prior ? f->pmean : nullptr;
      a.prior_dense = f->cfg.prior_dense; a.scaler = f->opt.scaler; a.maxit = f->opt.maxit; a.watch = 0;
      a.xtol = f->opt.xtol; a.gtol = f->opt.gtol; a.factor_up = f->opt.factor_up; a.factor_down = f->opt.factor_down;
      a.hostptr_bits = 0.0;
      a.cov = f->cov; a.ldc = f->ldm; a.want_cov = 1; a.pad_ = 0;
      a.host = f->chol_work;                       // (free until a covariance is asked for: 16384 doubles per fit)
      a.pub = nullptr; a.seq = 0;                  // (the batch's results are copied after a stream synchronisation)
      lsqamd_jit::FitBatch bt;
      bt.ymean_stride = f->ymean_stride; bt.prec_stride = f->cfg.prior_dense ? 0 : P; bt.tile_stride = red_stride;
      bt.cov_stride = P * f->ldm; bt.scratch_stride = w_stride;
      bt.logdet = f->logdet; bt.mu = f->s.mu; bt.chi2 = f->s.chi2;
      bt.nit = f->s.nit; bt.info = f->s.info; bt.status = f->s.status; bt.nfev = f->s.nfev; bt.njev = f->s.njev;
      bt.active = f->s.active; bt.reason = f->s.bad;
      BHIP(f, lsqamd_jit::launch_fit_batch(k, f->st, a, bt, (int)B));
      std::vector<int32_t> reason((size_t)B, 0);
      BHIP(f, hipMemcpyAsync(reason.data(), f->s.bad, sizeof(int32_t) * B, hipMemcpyDeviceToHost, f->st));
      BHIP(f, hipStreamSynchronize(f->st));
      bool all_done = true, all_cov = true;
      for (int64_t b = 0; b < B; ++b) {
        all_done = all_done && (reason[(size_t)b] == 1 || reason[(size_t)b] == 3);
        all_cov = all_cov && reason[(size_t)b] == 1;
      }
      if (all_done) {
        f->one_launch = true;
        f->rounds = 1;
        f->graph_used = 0;
        f->have_cov_from_run = all_cov;
      } else {   // the lockstep engine from the start: p0 again, every fit active
        BHIP(f, hipMemcpyAsync(f->px, p0, sizeof(double) * B * P, hipMemcpyHostToDevice, f->st));
        std::vector<int32_t> ones((size_t)B, 1);
        BHIP(f, hipMemcpyAsync(f->s.active, ones.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice, f->st));
        BHIP(f, hipStreamSynchronize(f->st));
      }
    }
  }
  int rc = 0;
  if (!f->one_launch) {
  rc = normal_all(f, f->s.active);
  if (rc) return rc;
  hipLaunchKernelGGL(b_init_kernel, dim3((unsigned)B), dim3(256), 0, f->st, P, f->T, f->red, f->npk + P + 1,
                     f->red + f->npk, f->npk + P + 1, f->diag, f->opt.scaler, f->s);
  f->rounds = 0;
  f->graph_used = 0;
  // kernel arguments (tolerances, maxit, factors) are baked into graph nodes: re-capture per run
  if (f->gexec) { (void)hipGraphExecDestroy(f->gexec); f->gexec = nullptr; }
  if (f->graph) { (void)hipGraphDestroy(f->graph); f->graph = nullptr; }
  int32_t n_active = (int32_t)B;
  const int64_t max_rounds = (int64_t)(f->opt.maxit > 0 ? f->opt.maxit : 0) * 17 + 1;
  while (n_active > 0 && f->rounds < max_rounds && f->opt.maxit > 0) {
    if (use_graph && f->rounds >= 1) {
      if (!f->gexec) {
        hipError_t e = hipStreamBeginCapture(f->st, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
          rc = round_all(f);
          hipGraph_t gr = nullptr;
          e = hipStreamEndCapture(f->st, &gr);
          // a capture another thread's activity invalidated makes the captured launches fail too: that is a failed CAPTURE
          // (nothing ran, nothing changed on the device), not a failed round -- the round below runs eagerly
          if (rc && e == hipSuccess) return rc;
          rc = 0;
          if (e == hipSuccess && gr && hipGraphInstantiate(&f->gexec, gr, nullptr, nullptr, 0) == hipSuccess) {
            f->graph = gr;
          } else {
            if (gr) (void)hipGraphDestroy(gr);
            f->gexec = nullptr;
            use_graph = 0;
            capture_reset(f->st);      // (an invalidated capture leaves the stream unusable until it is reset: common.h)
          }
        } else {
          use_graph = 0;
        }
        (void)hipGetLastError();
      }
      if (f->gexec) {
        BHIP(f, hipGraphLaunch(f->gexec, f->st));
        f->graph_used += 1;
      } else {
        rc = round_all(f);
        if (rc) return rc;
      }
    } else {
      rc = round_all(f);
      if (rc) return rc;
    }
    f->rounds += 1;
    BHIP(f, hipMemcpyAsync(&n_active, f->s.n_active, sizeof(int32_t), hipMemcpyDeviceToHost, f->st));
    BHIP(f, hipStreamSynchronize(f->st));
  }
  }   // (!one_launch)
  (void)hipEventRecord(e1, f->st);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  lsqamd_host::event_give(e0);
  lsqamd_host::event_give(e1);
  f->ran = true;
  f->have_cov = f->one_launch && f->have_cov_from_run;     // (the one-launch kernel leaves covariances and log dets behind)
  if (summaries) {
    // the per-fit scalars lie one after the other in the workspace (carve_b: s.mu .. s.njev): ONE copy instead of seven
    const char *lo = reinterpret_cast<const char *>(f->s.mu), *hi = reinterpret_cast<const char *>(f->s.njev + B);
    std::vector<char> span((size_t)(hi - lo));
    BHIP(f, copy_sync(f, span.data(), lo, span.size(), hipMemcpyDeviceToHost));
    auto at = [&](const void *dev) { return span.data() + (reinterpret_cast<const char *>(dev) - lo); };
    const double *mu = reinterpret_cast<const double *>(at(f->s.mu)), *chi2 = reinterpret_cast<const double *>(at(f->s.chi2));
    const int32_t *nit = reinterpret_cast<const int32_t *>(at(f->s.nit)), *info = reinterpret_cast<const int32_t *>(at(f->s.info));
    const int32_t *status = reinterpret_cast<const int32_t *>(at(f->s.status)), *nfev = reinterpret_cast<const int32_t *>(at(f->s.nfev));
    const int32_t *njev = reinterpret_cast<const int32_t *>(at(f->s.njev));
    for (int64_t b = 0; b < B; ++b) {
      lsqamd_summary &s = summaries[b];
      std::memset(&s, 0, sizeof(s));
      s.status = status[b] == -2 ? LSQAMD_EMAXITER : status[b];
      s.info = info[b];
      s.stopping_criterion = (info[b] >= 0 && info[b] <= 3) ? info[b] : (info[b] == 27 ? 4 : 0);
      s.nit = nit[b]; s.nfev = nfev[b]; s.njev = njev[b]; s.ntrial = nfev[b] - 1;
      s.chi2 = chi2[b]; s.mu = mu[b];
      s.logdet_jtj = NAN;
      s.t_run_ms = ms;
      s.t_setup_ms = (double)f->graph_used;  // rounds replayed from the captured graph
    }
  }
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamdb_get_x(lsqamdb_fits *f, double *out, size_t cap) try {
  if (!f || !out) return LSQAMD_EINVAL;
  if (cap < (size_t)(f->B * f->P)) BFAIL(f, LSQAMD_ECAPACITY, "get_x: need %lld", (long long)(f->B * f->P));
  BHIP(f, copy_sync(f, out, f->px, sizeof(double) * f->B * f->P, hipMemcpyDeviceToHost));
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

/* (J^T J)^-1 of every fit at its final point + log det(J^T J); results stay on the device */
int lsqamdb_covariance(lsqamdb_fits *f, double *logdet_out, size_t cap) try {
  if (!f) return LSQAMD_EINVAL;
  if (!f->ran) BFAIL(f, LSQAMD_EINVAL, "covariance: run first");
  if (f->have_cov && f->one_launch) {     // formed by the fit kernel itself
    if (logdet_out) {
      if (cap < (size_t)f->B) BFAIL(f, LSQAMD_ECAPACITY, "covariance: need %lld", (long long)f->B);
      BHIP(f, copy_sync(f, logdet_out, f->logdet, sizeof(double) * f->B, hipMemcpyDeviceToHost));
    }
    return 0;
  }
  const int64_t P = f->P, B = f->B, m_stride = P * f->ldm, w_stride = f->nblk * 128 * 128;
  const int64_t red_stride = f->npk + P + 1;
  const unsigned ntile = (unsigned)(f->T * (f->T + 1) / 2);
  hipLaunchKernelGGL(b_build_damped_kernel, dim3(ntile, 16, (unsigned)B), dim3(256), 0, f->st, f->red,
                     red_stride, P, f->T, f->ldm, f->s.mu, f->diag, (const double *)nullptr, red_stride, f->M,
                     m_stride, (const int32_t *)nullptr, 0);
  BHIP(f, potrf_upper_batched(f->st, f->M, P, f->ldm, P, f->chol_work, f->s.cholinfo, (int32_t)B, m_stride,
                              w_stride, nullptr));
  hipLaunchKernelGGL(b_logdiag_kernel, dim3((unsigned)B), dim3(256), 0, f->st, f->M, m_stride, P, f->ldm,
                     f->logdet);
  BHIP(f, trtri_upper_to_lower_T_batched(f->st, f->M, P, f->ldm, f->chol_work, f->Wl, f->ldm, (int32_t)B,
                                         m_stride, w_stride, m_stride));
  GemmTN g;
  g.X = f->Wl; g.Y = f->Wl; g.ldx = g.ldy = f->ldm; g.sx = g.sy = m_stride;
  g.C = f->cov; g.ldc = f->ldm; g.sc = m_stride;
  g.M = P; g.N = P; g.K = P;
  g.upper_only = 1; g.xy_lower_tri = 1;
  g.batch = (int32_t)B;
  BHIP(f, launch_gemm_tn(f->st, g));
  hipLaunchKernelGGL(b_symmetrize_kernel, dim3((unsigned)((P + 255) / 256), (unsigned)P, (unsigned)B), dim3(256), 0,
                     f->st, f->cov, m_stride, P, f->ldm);
  BHIP(f, hipStreamSynchronize(f->st));
  f->have_cov = true;
  if (logdet_out) {
    if (cap < (size_t)B) BFAIL(f, LSQAMD_ECAPACITY, "covariance: need %lld", (long long)B);
    BHIP(f, copy_sync(f, logdet_out, f->logdet, sizeof(double) * B, hipMemcpyDeviceToHost));
  }
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamdb_get_cov(lsqamdb_fits *f, int32_t fit, double *out, size_t cap) try {
  if (!f || !out || fit < 0 || fit >= f->B) return LSQAMD_EINVAL;
  const int64_t P = f->P;
  if (cap < (size_t)(P * P)) BFAIL(f, LSQAMD_ECAPACITY, "get_cov: need %lld", (long long)(P * P));
  if (!f->have_cov) {
    const int rc = lsqamdb_covariance(f, nullptr, 0);
    if (rc) return rc;
  }
  BHIP(f, copy2d_sync(f, out, sizeof(double) * P, f->cov + (int64_t)fit * P * f->ldm, sizeof(double) * f->ldm,
                      sizeof(double) * P, (size_t)P, hipMemcpyDeviceToHost));
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamdb_get_cov_all(lsqamdb_fits *f, double *out, size_t cap) try {
  if (!f || !out) return LSQAMD_EINVAL;
  const int64_t P = f->P, B = f->B;
  if (cap < (size_t)(B * P * P)) BFAIL(f, LSQAMD_ECAPACITY, "get_cov_all: need %lld", (long long)(B * P * P));
  if (!f->have_cov) {
    const int rc = lsqamdb_covariance(f, nullptr, 0);
    if (rc) return rc;
  }
  // fit b's rows follow fit b - 1's in f->cov (P rows of ldm doubles each): one strided copy of B * P rows
  BHIP(f, copy2d_sync(f, out, sizeof(double) * P, f->cov, sizeof(double) * f->ldm, sizeof(double) * P, (size_t)(B * P),
                      hipMemcpyDeviceToHost));
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int32_t lsqamdb_rounds(const lsqamdb_fits *f) { return f ? f->rounds : -1; }

int lsqamdb_timing_enable(lsqamdb_fits *f, int32_t on) try {
  if (!f) return LSQAMD_EINVAL;
  (void)hipStreamSynchronize(f->st);
  f->timing = on != 0;
  for (auto *v : {&f->tm_syrk, &f->tm_chol}) {
    for (auto &pr : *v) {
      lsqamd_host::event_give(pr.first);
      lsqamd_host::event_give(pr.second);
    }
    v->clear();
  }
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamdb_timing_get(lsqamdb_fits *f, int32_t which, double *total_ms, int64_t *count) try {
  if (!f || (which != LSQAMD_T_SYRK && which != LSQAMD_T_CHOLESKY)) return LSQAMD_EINVAL;
  (void)hipStreamSynchronize(f->st);
  double tot = 0.0;
  int64_t n = 0;
  for (auto &pr : which == LSQAMD_T_SYRK ? f->tm_syrk : f->tm_chol) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { tot += ms; ++n; }
  }
  if (total_ms) *total_ms = tot;
  if (count) *count = n;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

}  // extern "C"
