// The covariance of a fit whose final Jacobian is rank deficient -- what the reference's two plugins
// return there, instead of "undefined".
//
//   gsl_multifit (src/lsqfit/_gsl.pyx:704-706): gsl_multifit_nlinear_covar(J, epsrel = 0.0, covar) --
//     QR of J with column pivoting, columns whose pivot |R_kk| <= epsrel |R_11| = 0 are dropped, the
//     leading block inverted:  cov = Pi [ (R_k^T R_k)^-1  0 ; 0  0 ] Pi^T   [GSL covar.c, third party].
//     With R^T R = Pi^T (J^T J) Pi that R is the Cholesky factor of J^T J under DIAGONAL pivoting (the
//     largest remaining column norm is the largest remaining diagonal entry), which is how it is formed
//     here -- from the Gram matrix the fit already holds;
//   scipy_least_squares (src/lsqfit/_scipy.py:170-175): cov = V diag(1/s^2) V^T over the singular
//     values s > eps max(shape) s_0 of the Jacobian: the eigen-decomposition of J^T J (cyclic Jacobi).
// Both work from G = J^T J (+ prior precision), whose entries carry rounding noise of P eps max|G|:
// a direction counts as null when its pivot / eigenvalue is below that floor (the reference's own
// thresholds -- 0 and (eps max(shape))^2 in these units -- lie below it; for a column that is exactly
// zero, the reference's test case, the results are identical).  This is the rare path: it runs on the
// host (O(P^3) on one core), only after the device factorisation has met a non-positive pivot.
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

#include "fit_state.h"

using namespace lsqamd;

namespace {

// G (n x n, row-major, symmetric) -> cov per gsl's pivoted-QR recipe; returns the number of dropped columns
int pivoted_inverse(const std::vector<double> &G, int64_t n, std::vector<double> &cov) {
  std::vector<double> A(G);
  std::vector<int64_t> piv((size_t)n);
  for (int64_t i = 0; i < n; ++i) piv[(size_t)i] = i;
  double gmax = 0.0;
  for (int64_t i = 0; i < n; ++i) gmax = std::max(gmax, std::fabs(A[(size_t)(i * n + i)]));
  const double floor_ = (double)n * std::numeric_limits<double>::epsilon() * gmax;
  int64_t k = 0;
  // right-looking Cholesky with diagonal pivoting on the lower triangle: A = L L^T on the permuted matrix
  for (; k < n; ++k) {
    int64_t best = k;
    for (int64_t i = k + 1; i < n; ++i)
      if (A[(size_t)(i * n + i)] > A[(size_t)(best * n + best)]) best = i;
    if (!(A[(size_t)(best * n + best)] > floor_)) break;
    if (best != k) {   // symmetric swap of rows / columns k and best
      for (int64_t j = 0; j < n; ++j) std::swap(A[(size_t)(k * n + j)], A[(size_t)(best * n + j)]);
      for (int64_t i = 0; i < n; ++i) std::swap(A[(size_t)(i * n + k)], A[(size_t)(i * n + best)]);
      std::swap(piv[(size_t)k], piv[(size_t)best]);
    }
    const double d = std::sqrt(A[(size_t)(k * n + k)]);
    A[(size_t)(k * n + k)] = d;
    for (int64_t i = k + 1; i < n; ++i) A[(size_t)(i * n + k)] /= d;
    for (int64_t i = k + 1; i < n; ++i) {
      const double lik = A[(size_t)(i * n + k)];
      if (lik == 0.0) continue;
      for (int64_t j = k + 1; j <= i; ++j) A[(size_t)(i * n + j)] -= lik * A[(size_t)(j * n + k)];
    }
    for (int64_t i = k + 1; i < n; ++i)   // keep the trailing block symmetric for the next swap
      for (int64_t j = k + 1; j < i; ++j) A[(size_t)(j * n + i)] = A[(size_t)(i * n + j)];
  }
  // inverse of the leading k x k block: W = L_k^-1 (lower), block = W^T W
  std::vector<double> W((size_t)(k * k), 0.0);
  for (int64_t c = 0; c < k; ++c) {
    W[(size_t)(c * k + c)] = 1.0 / A[(size_t)(c * n + c)];
    for (int64_t i = c + 1; i < k; ++i) {
      double s = 0.0;
      for (int64_t j = c; j < i; ++j) s += A[(size_t)(i * n + j)] * W[(size_t)(j * k + c)];
      W[(size_t)(i * k + c)] = -s / A[(size_t)(i * n + i)];
    }
  }
  cov.assign((size_t)(n * n), 0.0);
  for (int64_t a = 0; a < k; ++a)
    for (int64_t b = 0; b <= a; ++b) {
      double s = 0.0;
      for (int64_t i = a; i < k; ++i) s += W[(size_t)(i * k + a)] * W[(size_t)(i * k + b)];
      cov[(size_t)(piv[(size_t)a] * n + piv[(size_t)b])] = s;
      cov[(size_t)(piv[(size_t)b] * n + piv[(size_t)a])] = s;
    }
  return (int)(n - k);
}

// G = V diag(lam) V^T by cyclic Jacobi; cov = sum over lam_i above the threshold of v_i v_i^T / lam_i
int eigen_pseudo_inverse(const std::vector<double> &G, int64_t n, int64_t n_rows, std::vector<double> &cov) {
  std::vector<double> A(G), V((size_t)(n * n), 0.0);
  for (int64_t i = 0; i < n; ++i) V[(size_t)(i * n + i)] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int64_t i = 0; i < n; ++i) {
      diag += A[(size_t)(i * n + i)] * A[(size_t)(i * n + i)];
      for (int64_t j = 0; j < i; ++j) off += A[(size_t)(i * n + j)] * A[(size_t)(i * n + j)];
    }
    if (!(off > 1e-60 * diag) || off == 0.0) break;
    for (int64_t p = 0; p < n - 1; ++p)
      for (int64_t q = p + 1; q < n; ++q) {
        const double apq = A[(size_t)(p * n + q)];
        if (apq == 0.0) continue;
        const double app = A[(size_t)(p * n + p)], aqq = A[(size_t)(q * n + q)];
        if (std::fabs(apq) < 1e-300 + 1e-34 * std::sqrt(std::fabs(app * aqq))) { A[(size_t)(p * n + q)] = A[(size_t)(q * n + p)] = 0.0; continue; }
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int64_t k = 0; k < n; ++k) {   // columns p, q
          const double akp = A[(size_t)(k * n + p)], akq = A[(size_t)(k * n + q)];
          A[(size_t)(k * n + p)] = c * akp - s * akq;
          A[(size_t)(k * n + q)] = s * akp + c * akq;
        }
        for (int64_t k = 0; k < n; ++k) {   // rows p, q
          const double apk = A[(size_t)(p * n + k)], aqk = A[(size_t)(q * n + k)];
          A[(size_t)(p * n + k)] = c * apk - s * aqk;
          A[(size_t)(q * n + k)] = s * apk + c * aqk;
        }
        A[(size_t)(p * n + q)] = A[(size_t)(q * n + p)] = 0.0;
        for (int64_t k = 0; k < n; ++k) {
          const double vkp = V[(size_t)(k * n + p)], vkq = V[(size_t)(k * n + q)];
          V[(size_t)(k * n + p)] = c * vkp - s * vkq;
          V[(size_t)(k * n + q)] = s * vkp + c * vkq;
        }
      }
  }
  double lmax = 0.0;
  for (int64_t i = 0; i < n; ++i) lmax = std::max(lmax, A[(size_t)(i * n + i)]);
  const double eps = std::numeric_limits<double>::epsilon();
  const double m = (double)std::max(n, n_rows);
  // _scipy.py:171-174 in units of s^2, floored by the rounding noise of the Gram matrix
  const double thr = std::max(eps * m * eps * m, (double)n * eps) * lmax;
  cov.assign((size_t)(n * n), 0.0);
  int dropped = 0;
  for (int64_t i = 0; i < n; ++i) {
    const double lam = A[(size_t)(i * n + i)];
    if (!(lam > thr)) { ++dropped; continue; }
    for (int64_t a = 0; a < n; ++a) {
      const double va = V[(size_t)(a * n + i)] / lam;
      if (va == 0.0) continue;
      for (int64_t b = 0; b < n; ++b) cov[(size_t)(a * n + b)] += va * V[(size_t)(b * n + i)];
    }
  }
  return dropped;
}

}  // namespace

namespace lsqamd_host {

// after do_covariance met a non-positive pivot: the reference's truncated inverse into f->cov.
// Returns the number of dropped directions (> 0), or a negative code (then the caller reports ENOTPD as before).
int covariance_rank_deficient(lsqamd_fit *f) {
  const int64_t P = f->P;
  const bool scipy = f->opt.trs >= LSQAMD_TRS_TRF;
  static const int64_t cap = [] { const char *e = getenv("LSQAMD_RANKDEF_MAXP"); return e ? atoll(e) : (int64_t)2048; }();
  if (P > (scipy ? std::min<int64_t>(cap, 1024) : cap)) return LSQAMD_EUNSUPPORTED;   // O(P^3) on one host core
  if (f->comm || f->reduce) {
    // every rank holds the same reduced G and runs the same host arithmetic: identical results, no exchange
  }
  // G = J^T J (+ prior), dense, from the packed tiles the fit holds
  HIPCHK(f, launch_unpack_sym(f->st, f->redbuf, P, f->Wl, f->ldm));
  f->have_dense_A = false;
  std::vector<double> G((size_t)(P * P)), cov;
  HIPCHK(f, hipMemcpy2DAsync(G.data(), sizeof(double) * P, f->Wl, sizeof(double) * f->ldm, sizeof(double) * P, (size_t)P,
                             hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipStreamSynchronize(f->st));
  for (double v : G)
    if (!std::isfinite(v)) return LSQAMD_ENONFINITE;
  const int64_t nf = f->N + (f->cfg.has_prior ? P : 0);
  const int dropped = scipy ? eigen_pseudo_inverse(G, P, nf, cov) : pivoted_inverse(G, P, cov);
  if (dropped <= 0) return LSQAMD_ENOTPD;   // nothing to drop and still no factor: leave it to the caller's report
  HIPCHK(f, hipMemcpy2DAsync(f->cov, sizeof(double) * f->ldm, cov.data(), sizeof(double) * P, sizeof(double) * P, (size_t)P,
                             hipMemcpyHostToDevice, f->st));
  HIPCHK(f, hipStreamSynchronize(f->st));
  f->have_cov = true;
  f->logdet = -INFINITY;   // log det of a singular J^T J (numpy's slogdet, src/lsqfit/__init__.py:711-719, says the same)
  return dropped;
}

}  // namespace lsqamd_host

// host-only entry point for the parity tests of the two recipes (no GPU needed)
extern "C" int lsqamd_op_truncated_inverse(const double *G, int64_t n, int64_t n_rows, int32_t scipy_form, double *cov_out,
                                           int32_t *dropped) try {
  if (!G || !cov_out || n < 1) return LSQAMD_EINVAL;
  std::vector<double> g(G, G + n * n), cov;
  const int k = scipy_form ? eigen_pseudo_inverse(g, n, n_rows, cov) : pivoted_inverse(g, n, cov);
  std::copy(cov.begin(), cov.end(), cov_out);
  if (dropped) *dropped = k;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)
