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
constexpr int PLD = NB;  // LDS leading dimension: every access pattern below walks rows
constexpr int SB = 32;   // register-resident sub-block edge

size_t potrf_work_bytes(int64_t n) {
  const int64_t nblk = (n + NB - 1) / NB;
  return (size_t)nblk * NB * NB * sizeof(double);
}

__device__ __forceinline__ double readlane_d(double v, int lane) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}

// ---- diagonal-block kernel ---------------------------------------------------------------
// One workgroup (4 waves) factors a 128 x 128 block A = U^T U held in LDS and then inverts
// U in place, in 32-wide sub-blocks so that the sequential pivot chain runs out of
// registers inside a single wave (no workgroup barriers on the critical path):
//   1a  wave 0: 32 x 32 Cholesky, one column per lane, pivots broadcast with v_readlane
//   1b  forward substitution of the 32-row panel, one column per thread
//   1c  rank-32 update of the trailing block, 16 x 16 threads x (NA x NA) register tiles
//   3a  the four 32 x 32 triangular inverses, one per wave, one column per lane
//   3b/3c  off-diagonal blocks of the inverse by the 2 x 2 block formula
//          inv([[a, b], [0, c]]) = [[a^-1, -a^-1 b c^-1], [0, c^-1]]  (two small matmuls,
//          the dead strictly-lower triangle is the scratch for the intermediate product).
// Blocks smaller than 128 are padded with the identity.

__device__ __forceinline__ void chol32_wave(double *s, int j0, double *dinv, int32_t *info, int k0,
                                            int lane) {
  const int c = lane & 31;
  double col[SB];
#pragma unroll
  for (int i = 0; i < SB; ++i) col[i] = s[(j0 + i) * PLD + j0 + c];
#pragma unroll
  for (int j = 0; j < SB; ++j) {
    double d = readlane_d(col[j], j);
    if (!(d > 0.0) || !(d < 1.0e300)) {  // wave-uniform: not positive definite / not finite
      if (lane == 0) atomicCAS(info, 0, k0 + j0 + j + 1);
      d = 1.0;
    }
    const double inv = 1.0 / sqrt(d);
    const double u = (c >= j) ? col[j] * inv : 0.0;
    col[j] = u;
    if (lane == 0) dinv[j0 + j] = inv;
#pragma unroll
    for (int i = j + 1; i < SB; ++i) col[i] -= readlane_d(u, i) * u;
  }
  if (lane < SB) {
#pragma unroll
    for (int i = 0; i < SB; ++i) s[(j0 + i) * PLD + j0 + c] = (i <= c) ? col[i] : 0.0;
  }
}

// columns cc >= j0 + 32 of rows j0..j0+31: x <- U11^-T x
// (`off` is threaded through an empty asm after every pivot so the LDS reads of row j+1
// cannot be scheduled above the arithmetic of row j -- hoisting all 496 of them spills)
__device__ __forceinline__ void panel32_solve(double *s, int j0, const double *dinv, int cc) {
  double x[SB];
#pragma unroll
  for (int i = 0; i < SB; ++i) x[i] = s[(j0 + i) * PLD + cc];
  int off = j0 * PLD + j0;
#pragma unroll
  for (int j = 0; j < SB; ++j) {
    x[j] *= dinv[j0 + j];
#pragma unroll
    for (int i = j + 1; i < SB; ++i) x[i] -= s[off + j * PLD + i] * x[j];
    if (j + 1 < SB) asm volatile("" : "+v"(off) : "v"(x[j + 1]));
  }
#pragma unroll
  for (int i = 0; i < SB; ++i) s[(j0 + i) * PLD + cc] = x[i];
}

// A22 -= U12^T U12 for the (16 NA) x (16 NA) trailing block starting at r0 = j0 + 32
template <int NA>
__device__ __forceinline__ void trailing32(double *s, int j0, int tid) {
  const int r0 = j0 + SB, ty = tid >> 4, tx = tid & 15;
  double acc[NA][NA];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NA; ++b) acc[a][b] = 0.0;
#pragma unroll 4
  for (int k = 0; k < SB; ++k) {
    const double *row = s + (j0 + k) * PLD + r0;
    double ui[NA], uc[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      ui[a] = row[ty + 16 * a];
      uc[a] = row[tx + 16 * a];
    }
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int b = 0; b < NA; ++b) acc[a][b] += ui[a] * uc[b];
  }
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NA; ++b) {
      const int i = r0 + ty + 16 * a, c = r0 + tx + 16 * b;
      if (c >= i) s[i * PLD + c] -= acc[a][b];
    }
}

// in-place inverse of the upper-triangular 32 x 32 block at (j0, j0); one column per lane
__device__ __forceinline__ void trinv32_wave(double *s, int j0, const double *dinv, int lane) {
  const int c = lane & 31;
  double v[SB];
  int off = j0 * PLD + j0;
#pragma unroll
  for (int i = SB - 1; i >= 0; --i) {
    double acc = 0.0;
#pragma unroll
    for (int k = i + 1; k < SB; ++k) acc += s[off + i * PLD + k] * v[k];
    const double di = dinv[j0 + i];
    v[i] = (i < c) ? -di * acc : ((i == c) ? di : 0.0);
    asm volatile("" : "+v"(off) : "v"(v[i]));  // serialise the rows' LDS reads (see panel32_solve)
  }
  // all lanes of the wave have finished reading the block before anyone overwrites it
  __builtin_amdgcn_wave_barrier();
  if (lane < SB) {
#pragma unroll
    for (int i = 0; i < SB; ++i) s[(j0 + i) * PLD + j0 + c] = v[i];
  }
}

// C[i][j] (+)= sign * sum_k L[i][k] R[k][j] for an n x n block triple inside s.
// thread -> column j = tid % n, RPT consecutive rows; L rows broadcast, R rows contiguous.
template <int N_, int RPT>
__device__ __forceinline__ void small_matmul(const double *L, const double *R, double *Cout,
                                             double sign, int tid) {
  constexpr int GROUPS = N_ / RPT;
  const int j = tid % N_, g = tid / N_;
  if (g >= GROUPS) return;
  double acc[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) acc[r] = 0.0;
#pragma unroll 4
  for (int k = 0; k < N_; ++k) {
    const double rk = R[k * PLD + j];
#pragma unroll
    for (int r = 0; r < RPT; ++r) acc[r] += L[(g * RPT + r) * PLD + k] * rk;
  }
#pragma unroll
  for (int r = 0; r < RPT; ++r) Cout[(g * RPT + r) * PLD + j] = sign * acc[r];
}

__global__ __launch_bounds__(256) void potf2_inv_kernel(double *A, int64_t lda, int nb, double *uinv,
                                                        int32_t *info, int32_t k0) {
  extern __shared__ __attribute__((aligned(16))) double s[];
  double *dinv = s + NB * PLD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nsb = (nb + SB - 1) / SB;  // active 32-wide sub-blocks
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int i = idx >> 7, j = idx & 127;
    double v = (i == j) ? 1.0 : 0.0;
    if (i < nb && j < nb) v = (j >= i) ? A[(int64_t)i * lda + j] : 0.0;
    s[i * PLD + j] = v;
  }
  __syncthreads();
  // ---- phase 1: Cholesky
  for (int jb = 0; jb < nsb; ++jb) {
    const int j0 = jb * SB;
    if (wave == 0) chol32_wave(s, j0, dinv, info, k0, lane);
    __syncthreads();
    const int rest = nsb * SB - (j0 + SB);
    if (rest > 0) {
      if (tid < rest) panel32_solve(s, j0, dinv, j0 + SB + tid);
      __syncthreads();
      if (rest > 64) trailing32<6>(s, j0, tid);
      else if (rest > 32) trailing32<4>(s, j0, tid);
      else trailing32<2>(s, j0, tid);
      __syncthreads();
    }
  }
  // U back to global (upper triangle of the block)
  for (int idx = tid; idx < nb * NB; idx += 256) {
    const int i = idx >> 7, j = idx & 127;
    if (j >= i && j < nb) A[(int64_t)i * lda + j] = s[i * PLD + j];
  }
  __syncthreads();
  // ---- phase 3: inverse of U in place
  if (wave < nsb) trinv32_wave(s, wave * SB, dinv, lane);
  __syncthreads();
  if (nsb > 1) {
    // level 1: the two 64-blocks; T = b c^-1 into the dead lower sub-block, then b' = -a^-1 T
    const int half = tid >> 7, t = tid & 127;  // 128 threads per 64-block
    const bool act = (half == 0) || (nsb > 3);  // the second 64-block has an off-diagonal only with 4 sub-blocks
    const int q = half * 64;
    if (act)
      small_matmul<32, 8>(s + q * PLD + q + 32, s + (q + 32) * PLD + q + 32, s + (q + 32) * PLD + q, 1.0, t);
    __syncthreads();
    if (act)
      small_matmul<32, 8>(s + q * PLD + q, s + (q + 32) * PLD + q, s + q * PLD + q + 32, -1.0, t);
    __syncthreads();
    // the level-1 scratch sits inside the triangles level 2 multiplies with: clear it
    if (act) {
      for (int e = t; e < 32 * 32; e += 128) s[(q + 32 + (e >> 5)) * PLD + q + (e & 31)] = 0.0;
    }
    __syncthreads();
    if (nsb > 2) {
      // level 2: B = rows 0..63, cols 64..127; scratch = rows 64..127, cols 0..63
      small_matmul<64, 16>(s + 64, s + 64 * PLD + 64, s + 64 * PLD, 1.0, tid);
      __syncthreads();
      small_matmul<64, 16>(s, s + 64 * PLD, s + 64, -1.0, tid);
      __syncthreads();
    }
  }
  for (int idx = tid; idx < nb * NB; idx += 256) {
    const int i = idx >> 7, j = idx & 127;
    uinv[i * NB + j] = (j >= i && j < nb) ? s[i * PLD + j] : 0.0;
  }
}

static bool g_potf2_attr = false;
static constexpr size_t POTF2_LDS = (size_t)(NB * PLD + NB) * sizeof(double);  // 129 KiB

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
