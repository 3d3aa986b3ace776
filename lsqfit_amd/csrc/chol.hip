// LDS-blocked Cholesky family for the damped normal equations (gfx950).
//
// Replaces, for solver='cholesky', what GSL does behind
// gsl_multifit_nlinear_driver (src/lsqfit/_gsl.pyx:677): factor J^T J + mu D^T D
// and solve for the step; and gsl_multifit_nlinear_covar (_gsl.pyx:704-706):
// (J^T J)^-1 at the final point.
//
// Layout choice: the matrix is row-major, so the factorisation is the UPPER,
// row-panel-oriented right-looking one (A = U^T U).  Then every bulk update is a
// k-major "TN" contraction for gemm_tn_f64 (no transposes, coalesced rows):
//   diagonal block   potf2 + in-place triangular inverse, one workgroup, in LDS
//   row panel        U[k, k+nb:] = inv(U_kk)^T A[k, k+nb:]          (TN GEMM)
//   trailing update  A[k+nb:, k+nb:] -= U[k, k+nb:]^T U[k, k+nb:]    (TN GEMM, upper tiles)
// Extra columns to the right of the n x n block ride along in the row panels, so
// appending the gradient as column n yields the forward substitution U^-T g for free.
#include "common.h"

namespace lsqamd {

constexpr int NB = CHOL_NB;
constexpr int PLD = NB + 1;  // LDS leading dimension (odd -> column walks are conflict-free)

size_t potrf_work_bytes(int64_t n) {
  const int64_t nblk = (n + NB - 1) / NB;
  return (size_t)nblk * NB * NB * sizeof(double);
}

// One workgroup: Cholesky of the nb x nb diagonal block (upper), U written back,
// then U^-1 (upper) written to `uinv` (ld = NB, rows >= nb untouched).
__global__ __launch_bounds__(256) void potf2_inv_kernel(double *A, int64_t lda, int nb, double *uinv,
                                                        int32_t *info, int32_t k0) {
  extern __shared__ __attribute__((aligned(16))) double s[];
  double *xt = s + NB * PLD;
  const int tid = threadIdx.x;
  const int ti = tid >> 4, tj = tid & 15;
  for (int idx = tid; idx < nb * nb; idx += 256) {
    const int i = idx / nb, j = idx - i * nb;
    s[i * PLD + j] = (j >= i) ? A[(int64_t)i * lda + j] : 0.0;
  }
  __syncthreads();
  for (int j = 0; j < nb; ++j) {
    double d = s[j * PLD + j];
    if (!(d > 0.0) || !(d < 1.0e300)) {  // not positive definite / not finite
      if (tid == 0) atomicCAS(info, 0, k0 + j + 1);
      d = 1.0;
    }
    const double sq = sqrt(d), inv = 1.0 / sq;
    __syncthreads();
    for (int c = j + tid; c < nb; c += 256) s[j * PLD + c] = (c == j) ? sq : s[j * PLD + c] * inv;
    __syncthreads();
    for (int i = j + 1 + ti; i < nb; i += 16) {
      const double uji = s[j * PLD + i];
      for (int c = j + 1 + tj; c < nb; c += 16)
        if (c >= i) s[i * PLD + c] -= uji * s[j * PLD + c];
    }
    __syncthreads();
  }
  // U back to global (upper triangle of the block)
  for (int idx = tid; idx < nb * nb; idx += 256) {
    const int i = idx / nb, j = idx - i * nb;
    if (j >= i) A[(int64_t)i * lda + j] = s[i * PLD + j];
  }
  __syncthreads();
  // in-place inverse of the upper triangle (column by column, as LAPACK dtrti2)
  for (int j = 0; j < nb; ++j) {
    if (tid < j) xt[tid] = s[tid * PLD + j];
    __syncthreads();
    const double ajj = 1.0 / s[j * PLD + j];
    if (tid < j) {
      double y = 0.0;
      for (int k = tid; k < j; ++k) y += s[tid * PLD + k] * xt[k];
      s[tid * PLD + j] = -ajj * y;
    }
    __syncthreads();
    if (tid == 0) s[j * PLD + j] = ajj;
    __syncthreads();
  }
  for (int idx = tid; idx < nb * nb; idx += 256) {
    const int i = idx / nb, j = idx - i * nb;
    uinv[i * NB + j] = (j >= i) ? s[i * PLD + j] : 0.0;
  }
}

static bool g_potf2_attr = false;
static constexpr size_t POTF2_LDS = (size_t)(NB * PLD + NB) * sizeof(double);

static hipError_t launch_potf2(hipStream_t st, double *A, int64_t lda, int nb, double *uinv,
                               int32_t *info, int32_t k0) {
  if (!g_potf2_attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(potf2_inv_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)POTF2_LDS);
    if (e != hipSuccess) return e;
    g_potf2_attr = true;
  }
  hipLaunchKernelGGL(potf2_inv_kernel, dim3(1), dim3(256), POTF2_LDS, st, A, lda, nb, uinv, info, k0);
  return hipGetLastError();
}

hipError_t potrf_upper(hipStream_t st, double *A, int64_t n, int64_t lda, int64_t n_cols,
                       double *work, int32_t *dev_info) {
  hipError_t e = hipMemsetAsync(dev_info, 0, sizeof(int32_t), st);
  if (e != hipSuccess) return e;
  for (int64_t k0 = 0; k0 < n; k0 += NB) {
    const int nb = (int)((n - k0) < NB ? (n - k0) : NB);
    double *uinv = work + (k0 / NB) * NB * NB;
    e = launch_potf2(st, A + k0 * lda + k0, lda, nb, uinv, dev_info, (int32_t)k0);
    if (e != hipSuccess) return e;
    const int64_t rest = n_cols - (k0 + nb);
    if (rest <= 0) continue;
    GemmTN p;  // row panel: U[k, k+nb:] = inv(U_kk)^T * A[k, k+nb:]   (in place)
    p.X = uinv; p.ldx = NB;
    p.Y = A + k0 * lda + k0 + nb; p.ldy = lda;
    p.C = A + k0 * lda + k0 + nb; p.ldc = lda;
    p.M = nb; p.N = rest; p.K = nb;
    p.x_upper_tri = 1;
    e = launch_gemm_tn(st, p);
    if (e != hipSuccess) return e;
    const int64_t mrest = n - (k0 + nb);
    if (mrest <= 0) continue;
    GemmTN t;  // trailing: A[k+nb:, k+nb:] -= panel^T panel  (upper tiles only)
    t.X = p.C; t.ldx = lda;
    t.Y = p.C; t.ldy = lda;
    t.C = A + (k0 + nb) * lda + (k0 + nb); t.ldc = lda;
    t.M = mrest; t.N = rest; t.K = nb;
    t.alpha = -1.0; t.beta = 1.0;
    t.upper_only = 1;
    e = launch_gemm_tn(st, t);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// ---- back substitution U v = y ------------------------------------------------------
constexpr int BS_ROWS = 32;  // rows of y updated per workgroup

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Every workgroup recomputes v_k = inv(U_kk) y_k (128 x 128 GEMV, L2-resident) into
// LDS; workgroup 0 publishes it; workgroups 1.. subtract U[r, k-block] . v_k from
// their rows r < k0 of y.
__global__ __launch_bounds__(256) void backsolve_step_kernel(const double *A, int64_t lda, int64_t k0,
                                                             int nb, const double *uinv, double *y,
                                                             double *v) {
  __shared__ double vk[NB];
  __shared__ double yk[NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < nb) yk[tid] = y[k0 + tid];
  __syncthreads();
  for (int r = wave; r < nb; r += 4) {
    double acc = 0.0;
    for (int c = r + lane; c < nb; c += 64) acc += uinv[r * NB + c] * yk[c];
    acc = wave_sum(acc);
    if (lane == 0) vk[r] = acc;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    if (tid < nb) v[k0 + tid] = vk[tid];
    return;
  }
  const int64_t r0 = (int64_t)(blockIdx.x - 1) * BS_ROWS;
  for (int rr = wave; rr < BS_ROWS; rr += 4) {
    const int64_t r = r0 + rr;
    if (r >= k0) break;
    const double *row = A + r * lda + k0;
    double acc = 0.0;
    for (int c = lane; c < nb; c += 64) acc += row[c] * vk[c];
    acc = wave_sum(acc);
    if (lane == 0) y[r] -= acc;
  }
}

hipError_t backsolve_upper(hipStream_t st, const double *A, int64_t n, int64_t lda,
                           const double *work, double *y_inout) {
  // y_inout: [0,n) = y (destroyed), [n, 2n) = v on return
  double *y = y_inout, *v = y_inout + n;
  const int64_t nblk = (n + NB - 1) / NB;
  for (int64_t kb = nblk - 1; kb >= 0; --kb) {
    const int64_t k0 = kb * NB;
    const int nb = (int)((n - k0) < NB ? (n - k0) : NB);
    const unsigned grid = 1 + (unsigned)((k0 + BS_ROWS - 1) / BS_ROWS);
    hipLaunchKernelGGL(backsolve_step_kernel, dim3(grid), dim3(256), 0, st, A, lda, k0, nb,
                       work + kb * NB * NB, y, v);
  }
  return hipGetLastError();
}

// ---- W = U^-T (lower triangular, row-major) -------------------------------------------
__global__ void set_identity_kernel(double *A, int64_t n, int64_t ld) {
  const int64_t i = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) A[i * ld + j] = (i == j) ? 1.0 : 0.0;
}

hipError_t launch_set_identity(hipStream_t st, double *A, int64_t P, int64_t ld) {
  dim3 grid((unsigned)((P + 255) / 256), (unsigned)P);
  hipLaunchKernelGGL(set_identity_kernel, grid, dim3(256), 0, st, A, P, ld);
  return hipGetLastError();
}

hipError_t trtri_upper_to_lower_T(hipStream_t st, const double *A, int64_t n, int64_t lda,
                                  const double *work, double *W, int64_t ldw) {
  hipError_t e = launch_set_identity(st, W, n, ldw);
  if (e != hipSuccess) return e;
  for (int64_t k0 = 0; k0 < n; k0 += NB) {
    const int nb = (int)((n - k0) < NB ? (n - k0) : NB);
    const double *uinv = work + (k0 / NB) * NB * NB;
    GemmTN p;  // W[k, 0:k0+nb] = inv(U_kk)^T R[k, 0:k0+nb]  (in place)
    p.X = uinv; p.ldx = NB;
    p.Y = W + k0 * ldw; p.ldy = ldw;
    p.C = W + k0 * ldw; p.ldc = ldw;
    p.M = nb; p.N = k0 + nb; p.K = nb;
    p.x_upper_tri = 1;
    e = launch_gemm_tn(st, p);
    if (e != hipSuccess) return e;
    const int64_t mrest = n - (k0 + nb);
    if (mrest <= 0) continue;
    GemmTN t;  // R[k+nb:, 0:k0+nb] -= U[k, k+nb:]^T W[k, 0:k0+nb]
    t.X = A + k0 * lda + k0 + nb; t.ldx = lda;
    t.Y = W + k0 * ldw; t.ldy = ldw;
    t.C = W + (k0 + nb) * ldw; t.ldc = ldw;
    t.M = mrest; t.N = k0 + nb; t.K = nb;
    t.alpha = -1.0; t.beta = 1.0;
    e = launch_gemm_tn(st, t);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

__global__ __launch_bounds__(256) void logdiag_kernel(const double *A, int64_t n, int64_t lda,
                                                      double *out) {
  __shared__ double part[4];
  double acc = 0.0;
  for (int64_t j = threadIdx.x; j < n; j += 256) acc += log(A[j * lda + j]);
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = part[0] + part[1] + part[2] + part[3];
}

hipError_t logdiag_sum(hipStream_t st, const double *A, int64_t n, int64_t lda, double *dev_out) {
  hipLaunchKernelGGL(logdiag_kernel, dim3(1), dim3(256), 0, st, A, n, lda, dev_out);
  return hipGetLastError();
}

}  // namespace lsqamd
