/* TEST / BASELINE INFRASTRUCTURE -- not part of the product (nothing under lsqfit_amd/ links or loads this).
 *
 * Unblocked column-pivoted Householder QR of a row-major m x n matrix, the algorithm class of the
 * reference's default solver: gsl_multifit_nlinear's `qr` solver (src/lsqfit/_gsl.pyx:646-647) factors the
 * Jacobian with gsl_linalg_QRPT_decomp once per LM iteration.  GSL is a third-party dependency that is
 * not under /root/reference and not in this image; this restates its published algorithm
 * [3P-recalled: gsl/linalg/qrpt.c, householder.c -- level-2 operations, row-major gsl_matrix, one
 * thread]: per column k
 *     pivot   = the remaining column of largest updated norm (swap, norms downdated as in LINPACK dqrdc);
 *     v, tau  = householder_transform(column k below the diagonal);
 *     A[k:, k+1:] -= tau v (v^T A[k:, k+1:])      -- householder_hm: a dgemv and a dger on the row-major block.
 * Used (a) by tests/test_oracle_qr_c.py against numpy's QR, (b) by bench.py's cpu_baseline.faithful_qr_1thread,
 * which times it on a bounded sample on the GPU box's host and scales by the flop count
 * 2 m n^2 - 2/3 n^3 (LAPACK's blocked dgeqp3, used there before, flatters an unblocked code by 5-10x).
 * Built by oracle/build_c.py (gcc -O2, no BLAS). */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>

/* A: m x n row-major (lda = n), overwritten by R (upper triangle) and the Householder vectors below it;
 * tau[n], perm[n] (column k of the factored matrix is column perm[k] of the input); work[2 n].
 * Returns 0. */
int oracle_qrpt_unblocked(double *A, long m, long n, double *tau, long *perm, double *work) {
  const long kmax = m < n ? m : n;
  for (long j = 0; j < n; ++j) {
    perm[j] = j;
    double s = 0.0;
    for (long i = 0; i < m; ++i) s += A[i * n + j] * A[i * n + j];
    work[j] = sqrt(s);
  }
  for (long k = 0; k < kmax; ++k) {
    /* pivot: largest remaining column norm */
    long kbest = k;
    double best = work[k];
    for (long j = k + 1; j < n; ++j)
      if (work[j] > best) { best = work[j]; kbest = j; }
    if (kbest != k) {
      for (long i = 0; i < m; ++i) { const double t = A[i * n + k]; A[i * n + k] = A[i * n + kbest]; A[i * n + kbest] = t; }
      { const long t = perm[k]; perm[k] = perm[kbest]; perm[kbest] = t; }
      { const double t = work[k]; work[k] = work[kbest]; work[kbest] = t; }
    }
    /* Householder transform of column k, rows k..m-1: v = (1, v_1..), tau; A[k][k] <- beta */
    double xnorm2 = 0.0;
    for (long i = k + 1; i < m; ++i) xnorm2 += A[i * n + k] * A[i * n + k];
    double tk = 0.0;
    if (xnorm2 > 0.0) {
      const double alpha = A[k * n + k];
      const double beta = -(alpha >= 0.0 ? 1.0 : -1.0) * hypot(alpha, sqrt(xnorm2));
      tk = (beta - alpha) / beta;
      const double s = 1.0 / (alpha - beta);
      for (long i = k + 1; i < m; ++i) A[i * n + k] *= s;
      A[k * n + k] = beta;
    }
    tau[k] = tk;
    /* apply to the remaining columns (householder_hm = dgemv + dger on the row-major block, row by row as
     * gslcblas walks it): w = A[k:, k+1:]^T v ; A[k:, k+1:] -= tau v w^T */
    if (tk != 0.0 && k + 1 < n) {
      double *w = work + n;
      for (long j = k + 1; j < n; ++j) w[j] = A[k * n + j];
      for (long i = k + 1; i < m; ++i) {
        const double vi = A[i * n + k];
        const double *row = A + i * n;
        for (long j = k + 1; j < n; ++j) w[j] += row[j] * vi;
      }
      for (long j = k + 1; j < n; ++j) A[k * n + j] -= tk * w[j];
      for (long i = k + 1; i < m; ++i) {
        const double tv = tk * A[i * n + k];
        double *row = A + i * n;
        for (long j = k + 1; j < n; ++j) row[j] -= tv * w[j];
      }
    }
    /* downdate the column norms (dqrdc's formula, recomputed when cancellation bites) */
    for (long j = k + 1; j < n; ++j) {
      if (work[j] == 0.0) continue;
      double t = fabs(A[k * n + j]) / work[j];
      t = 1.0 - t * t;
      if (t < 0.0) t = 0.0;
      if (t < 1e-6) {   /* the downdated estimate has lost its digits: recompute from what is left of the column */
        double s = 0.0;
        for (long i = k + 1; i < m; ++i) s += A[i * n + j] * A[i * n + j];
        work[j] = sqrt(s);
      } else {
        work[j] *= sqrt(t);
      }
    }
  }
  return 0;
}
