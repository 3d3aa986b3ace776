// Robust loss functions of the scipy plugin: scipy.optimize.least_squares' loss / f_scale, which
// lsqfit.scipy_least_squares documents (src/lsqfit/_scipy.py:76-79) and forwards verbatim (:147-153).  scipy applies a
// loss rho(z), z = (f_i / f_scale)^2, to every element of the residual vector the plugin hands it -- here: every row of
// the WHITENED residual -- by rescaling rows before the trust-region step sees them (optimize/_lsq/common.py
// scale_for_robust_loss_function; restated and pinned on scipy itself in oracle/trf.py):
//     J_i <- J_i * s_i,  f_i <- f_i * rho'(z_i) / s_i,  s_i = sqrt(max(eps, rho'(z_i) + 2 rho''(z_i) z_i)),
//     cost = 0.5 f_scale^2 sum_i rho(z_i).
// Both kernels are one pass over rows that are already in HBM (HBM-bound, N (P + 1) doubles read and written once).
#include "common.h"

namespace lsqamd {

namespace {

__device__ __forceinline__ double wsum_r(double v) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// rho, rho', rho'' of z (scipy's least_squares.py: huber, soft_l1, cauchy, arctan)
__device__ __forceinline__ void rho_of(int loss, double z, double &r0, double &r1, double &r2) {
  switch (loss) {
    case LSQAMD_LOSS_HUBER:
      if (z <= 1.0) { r0 = z; r1 = 1.0; r2 = 0.0; }
      else { const double s = sqrt(z); r0 = 2.0 * s - 1.0; r1 = 1.0 / s; r2 = -0.5 / (z * s); }
      break;
    case LSQAMD_LOSS_SOFT_L1: {
      const double t = 1.0 + z, s = sqrt(t);
      r0 = 2.0 * (s - 1.0); r1 = 1.0 / s; r2 = -0.5 / (t * s);
      break;
    }
    case LSQAMD_LOSS_CAUCHY: {
      const double t = 1.0 + z;
      r0 = log1p(z); r1 = 1.0 / t; r2 = -1.0 / (t * t);
      break;
    }
    case LSQAMD_LOSS_ARCTAN: {
      const double t = 1.0 + z * z;
      r0 = atan(z); r1 = 1.0 / t; r2 = -2.0 * z / (t * t);
      break;
    }
    default: r0 = z; r1 = 1.0; r2 = 0.0; break;
  }
}

// stage 1 of  f_scale^2 sum rho((r_i / f_scale)^2)  over a residual vector (stride 1) or a column of the Jacobian (stride ld)
__global__ __launch_bounds__(256) void robust_cost_stage1(const double *r, int64_t n, int64_t stride, int loss, double f_scale,
                                                          double *partial) {
  __shared__ double part[4];
  double a = 0.0;
  const double inv = 1.0 / f_scale;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double u = r[i * stride] * inv;
    double r0, r1, r2;
    rho_of(loss, u * u, r0, r1, r2);
    a += r0;
  }
  a = wsum_r(a);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = f_scale * f_scale * (part[0] + part[1] + part[2] + part[3]);
}

__global__ __launch_bounds__(256) void robust_sum_stage2(const double *partial, int n, double *out) {
  __shared__ double part[4];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += partial[i];
  a = wsum_r(a);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = part[0] + part[1] + part[2] + part[3];
}

// rows of [J | f] (ld doubles apart, f in column P) rescaled in place: one wave per row, lanes across the columns
__global__ __launch_bounds__(256) void robust_scale_rows_kernel(double *J, int64_t n, int64_t P, int64_t ld, int loss, double f_scale) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  const double inv = 1.0 / f_scale;
  for (int64_t i = wave; i < n; i += nwaves) {
    double *row = J + i * ld;
    const double fi = row[P], u = fi * inv;
    double r0, r1, r2;
    rho_of(loss, u * u, r0, r1, r2);
    double s = r1 + 2.0 * r2 * (u * u);           // (rho'' / f_scale^2) f^2 = rho'' z
    s = sqrt(s < 2.220446049250313e-16 ? 2.220446049250313e-16 : s);
    for (int64_t j = lane; j < P; j += 64) row[j] *= s;
    if (lane == 0) row[P] = fi * r1 / s;
  }
}

}  // namespace

hipError_t launch_robust_cost(hipStream_t st, const double *r, int64_t n, int64_t stride, int loss, double f_scale, double *partial,
                              double *out) {
  int blocks = (int)((n + 1023) / 1024);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(robust_cost_stage1, dim3(blocks), dim3(256), 0, st, r, n, stride, loss, f_scale, partial);
  hipLaunchKernelGGL(robust_sum_stage2, dim3(1), dim3(256), 0, st, partial, blocks, out);
  return hipGetLastError();
}

hipError_t launch_robust_scale_rows(hipStream_t st, double *J, int64_t n, int64_t P, int64_t ld, int loss, double f_scale) {
  if (n <= 0) return hipSuccess;
  int64_t blocks = (n + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(robust_scale_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, J, n, P, ld, loss, f_scale);
  return hipGetLastError();
}

}  // namespace lsqamd
