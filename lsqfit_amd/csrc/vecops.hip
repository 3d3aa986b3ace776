// Streaming (HBM-bound) kernels of the LM step: J^T f, |f|^2, block whitening of
// the residual vector, split-K slab reduction into the packed upper-triangular
// tile format, prior terms, damped-matrix assembly.  All reductions are
// two-stage and order-fixed (no floating-point atomics), so every rank of a
// row-sharded fit computes bit-identical replicated quantities.
//
// Packed format "Apk": the 128x128 tiles (tm, tn), tn >= tm, of J^T J stored
// back to back, tile index tm*T - tm(tm-1)/2 + (tn - tm), T = ceil(P/128),
// followed (by the caller) by the vector [J^T f ; |f|^2].  It is the buffer
// that is all-reduced between GPUs: 8*(T(T+1)/2*16384 + P + 1) bytes.
#include "common.h"

namespace lsqamd {

constexpr int TB = 128;  // packed tile edge

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

__host__ __device__ inline int64_t packed_tile_index(int64_t tm, int64_t tn, int64_t T) {
  return tm * T - tm * (tm - 1) / 2 + (tn - tm);
}

// ---- J^T f and |f|^2 in one pass over J ------------------------------------------------
// stage 1: partial[rc][j] = sum_{i in row chunk rc} J[i][j] * J[i][rcol]
// tri: J[i][j] == 0 for i > j (a whitening factor W_b^T): rows below a column pair are not read
__global__ __launch_bounds__(256) void colsum_dot_stage1(const double *J, int64_t nrows, int64_t ld,
                                                         int64_t ncols, const double *rv,
                                                         int64_t rs, int64_t rows_per_chunk,
                                                         double *partial, int tri) {
  const int64_t j = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
  int64_t r1 = r0 + rows_per_chunk;
  if (r1 > nrows) r1 = nrows;
  if (j >= ncols) return;
  if (tri && r1 > j + 2) r1 = j + 2;     // (an empty range leaves zeros)
  const bool two = j + 1 < ncols;
  double a0 = 0.0, a1 = 0.0;
  if ((ld & 1) == 0 && two) {
#pragma unroll 8   // eight rows requested before the first is used (the sums stay in row order): 2048 rows per
    for (int64_t i = r0; i < r1; ++i) {   // thread at N = 524288 were 2048 round trips to memory, one at a time
      const double ri = rv[i * rs];
      const double2 v = *reinterpret_cast<const double2 *>(J + i * ld + j);
      a0 += v.x * ri;
      a1 += v.y * ri;
    }
  } else {
#pragma unroll 4
    for (int64_t i = r0; i < r1; ++i) {
      const double ri = rv[i * rs];
      a0 += J[i * ld + j] * ri;
      if (two) a1 += J[i * ld + j + 1] * ri;
    }
  }
  partial[(int64_t)blockIdx.y * ncols + j] = a0;
  if (two) partial[(int64_t)blockIdx.y * ncols + j + 1] = a1;
}

// The same for a NARROW matrix (ncols <= 256: few parameters, many rows): with one thread per column pair a
// workgroup would have 33 of its 256 threads at work on 65 columns, each walking its rows alone (123 us for
// 65536 x 65).  Here CW column pairs x 256 / CW rows per sweep; the row lanes meet in LDS in a fixed order.
template <int CW>
__global__ __launch_bounds__(256) void colsum_dot_narrow(const double *J, int64_t nrows, int64_t ld,
                                                         int64_t ncols, const double *rv, int64_t rs,
                                                         int64_t rows_per_chunk, double *partial) {
  constexpr int CG = 256 / CW;
  __shared__ double sh[CG][2 * CW + 2];
  const int tx = threadIdx.x % CW, ty = threadIdx.x / CW;
  const int64_t j = 2 * tx;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk;
  int64_t r1 = r0 + rows_per_chunk;
  if (r1 > nrows) r1 = nrows;
  const bool one = j < ncols, two = j + 1 < ncols;
  double a0 = 0.0, a1 = 0.0;
  if (one) {
    if ((ld & 1) == 0 && two) {
#pragma unroll 4
      for (int64_t i = r0 + ty; i < r1; i += CG) {
        const double ri = rv[i * rs];
        const double2 v = *reinterpret_cast<const double2 *>(J + i * ld + j);
        a0 += v.x * ri;
        a1 += v.y * ri;
      }
    } else {
      for (int64_t i = r0 + ty; i < r1; i += CG) {
        const double ri = rv[i * rs];
        a0 += J[i * ld + j] * ri;
        if (two) a1 += J[i * ld + j + 1] * ri;
      }
    }
  }
  sh[ty][2 * tx] = a0;
  sh[ty][2 * tx + 1] = a1;
  __syncthreads();
  if (ty == 0 && one) {
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int q = 0; q < CG; ++q) {   // fixed order: deterministic
      t0 += sh[q][2 * tx];
      t1 += sh[q][2 * tx + 1];
    }
    partial[(int64_t)blockIdx.x * ncols + j] = t0;
    if (two) partial[(int64_t)blockIdx.x * ncols + j + 1] = t1;
  }
}

hipError_t launch_colsum_reduce(hipStream_t st, const double *partial, int64_t nchunks, int64_t ncols,
                                double *out);

hipError_t launch_colsum_dot(hipStream_t st, const double *J, int64_t nrows, int64_t ld,
                             int64_t ncols, int64_t rcol, double *partial, int64_t npartial,
                             double *out, const double *rvec, int64_t rvec_stride, int tri) {
  const double *rv = rvec ? rvec : J + rcol;
  const int64_t rs = rvec ? rvec_stride : ld;
  int64_t nchunks = npartial;
  if (nchunks > nrows) nchunks = nrows > 0 ? nrows : 1;
  const int64_t rpc = nrows > 0 ? (nrows + nchunks - 1) / nchunks : 1;
  nchunks = nrows > 0 ? (nrows + rpc - 1) / rpc : 0;
  if (nchunks > 0 && ncols <= 256 && nrows >= 4096 && !tri) {
    const int64_t cp = (ncols + 1) / 2;
    const dim3 grid((unsigned)nchunks);
    if (cp <= 16) hipLaunchKernelGGL(colsum_dot_narrow<16>, grid, dim3(256), 0, st, J, nrows, ld, ncols, rv, rs, rpc, partial);
    else if (cp <= 32) hipLaunchKernelGGL(colsum_dot_narrow<32>, grid, dim3(256), 0, st, J, nrows, ld, ncols, rv, rs, rpc, partial);
    else if (cp <= 64) hipLaunchKernelGGL(colsum_dot_narrow<64>, grid, dim3(256), 0, st, J, nrows, ld, ncols, rv, rs, rpc, partial);
    else hipLaunchKernelGGL(colsum_dot_narrow<128>, grid, dim3(256), 0, st, J, nrows, ld, ncols, rv, rs, rpc, partial);
  } else if (nchunks > 0) {
    dim3 grid((unsigned)((ncols + 511) / 512), (unsigned)nchunks);
    hipLaunchKernelGGL(colsum_dot_stage1, grid, dim3(256), 0, st, J, nrows, ld, ncols, rv, rs, rpc,
                       partial, tri);
  }
  return launch_colsum_reduce(st, partial, nchunks, ncols, out);
}

// out[j] = sum_c partial[c][j]  (second stage on its own: the first may have been fused elsewhere).
// Many chunks (one per 128-row tile row): 16 columns x 16 chunk-groups per workgroup, so a thread
// adds nchunks/16 values and there are ncols/16 workgroups (the one-thread-per-column stage 2
// above took 0.21 ms for 512 x 4096, this one is bandwidth-trivial).
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const double *partial, int64_t nchunks,
                                                            int64_t ncols, double *out) {
  __shared__ double sh[16][17];
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int64_t j = (int64_t)blockIdx.x * 16 + c;
  double a = 0.0;
  if (j < ncols) {
#pragma unroll 8   // (4096 chunks of a 524288-row fit: 256 values per thread, requested eight at a time, added in order)
    for (int64_t k = g; k < nchunks; k += 16) a += partial[k * ncols + j];
  }
  sh[g][c] = a;
  __syncthreads();
  if (g == 0 && j < ncols) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sh[q][c];   // fixed order: deterministic
    out[j] = t;
  }
}

hipError_t launch_colsum_reduce(hipStream_t st, const double *partial, int64_t nchunks, int64_t ncols,
                                double *out) {
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((ncols + 15) / 16)), dim3(256), 0, st, partial,
                     nchunks, ncols, out);
  return hipGetLastError();
}

// ---- |r|^2 ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_stage1(const double *r, int64_t n, double *partial) {
  __shared__ double part[4];
  double a = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double v = r[i];
    a += v * v;
  }
  a = wsum(a);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void sum_stage2(const double *partial, int n, double *out) {
  __shared__ double part[4];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += partial[i];
  a = wsum(a);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = part[0] + part[1] + part[2] + part[3];
}

hipError_t launch_sumsq(hipStream_t st, const double *r, int64_t n, double *partial, double *out) {
  int blocks = (int)((n + 1023) / 1024);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sumsq_stage1, dim3(blocks), dim3(256), 0, st, r, n, partial);
  hipLaunchKernelGGL(sum_stage2, dim3(1), dim3(256), 0, st, partial, blocks, out);
  return hipGetLastError();
}

// ---- whitening of the residual inside correlated blocks -----------------------------------
// 64 output rows per workgroup, the sum over the block's rows split four ways across the
// workgroup (coalesced 512-byte reads of W^T rows, four of them in flight per output row)
__global__ __launch_bounds__(256) void block_whiten_vec_kernel(const double *wt, const int64_t *row0,
                                                               const int64_t *bsize,
                                                               const int64_t *woff,
                                                               const double *delta, double *r_out,
                                                               int64_t stride, const int32_t *active,
                                                               int64_t skip_from) {
  __shared__ double sh[4][64];
  const int b = blockIdx.y;
  if (active && !active[blockIdx.z]) return;
  if (bsize[b] >= skip_from) return;  // large blocks go through the two-stage column-sum kernel
  const int64_t B = bsize[b], r0 = row0[b];
  if ((int64_t)blockIdx.x * 64 >= B) return;
  delta += (int64_t)blockIdx.z * stride;
  r_out += (int64_t)blockIdx.z * stride;
  const double *W = wt + woff[b];
  const int ml = threadIdx.x & 63, jg = threadIdx.x >> 6;
  const int64_t m = (int64_t)blockIdx.x * 64 + ml;
  double a = 0.0;
  if (m < B)
    for (int64_t j = jg; j < B; j += 4) a += W[j * B + m] * delta[r0 + j];
  sh[jg][ml] = a;
  __syncthreads();
  if (jg == 0 && m < B) r_out[r0 + m] = (sh[0][ml] + sh[1][ml]) + (sh[2][ml] + sh[3][ml]);
}

hipError_t launch_block_whiten_vec(hipStream_t st, const double *wt, const int64_t *row0,
                                   const int64_t *bsize, const int64_t *woff, int32_t n_blocks,
                                   int64_t max_block, const double *delta, double *r_out,
                                   int32_t batch, int64_t stride, const int32_t *batch_active,
                                   int64_t skip_from) {
  if (n_blocks <= 0) return hipSuccess;
  dim3 grid((unsigned)((max_block + 63) / 64), (unsigned)n_blocks, (unsigned)(batch < 1 ? 1 : batch));
  hipLaunchKernelGGL(block_whiten_vec_kernel, grid, dim3(256), 0, st, wt, row0, bsize, woff, delta,
                     r_out, stride, batch_active, skip_from);
  return hipGetLastError();
}

// ---- split-K slabs -> packed upper tiles ---------------------------------------------------
// prior (nullable): the prior precision is added on the way (dense: P x P, else its diagonal) -- one pass
// over the packed tiles instead of two
__global__ __launch_bounds__(256) void finalize_pack_kernel(const double *slabs, int32_t splits,
                                                            int64_t split_stride, int64_t P,
                                                            int64_t ld, int64_t T, double *apk,
                                                            const double *prior, int prior_dense, int64_t tile0) {
  // tile0 + blockIdx.x = packed tile index, blockIdx.y = 8-row strip inside the tile; thread -> two columns, two rows
  const int64_t tile = tile0 + blockIdx.x;
  int64_t t = tile, tm = 0;
  while (t >= T - tm) { t -= T - tm; ++tm; }
  const int64_t tn = tm + t;
  double *dst = apk + tile * TB * TB;
  const int c = 2 * (threadIdx.x & 63);
  const int64_t j = tn * TB + c;
  const bool vec = !(ld & 1) && !(split_stride & 1) && !(reinterpret_cast<uintptr_t>(slabs) & 15) && j + 1 < P;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = blockIdx.y * 8 + (threadIdx.x >> 6) + 4 * h;
    const int64_t i = tm * TB + r;
    double a0 = 0.0, a1 = 0.0;
    if (i < P && vec) {
      const double *src = slabs + i * ld + j;
#pragma unroll 8
      for (int s = 0; s < splits; ++s) {
        const double2 v = *reinterpret_cast<const double2 *>(src + s * split_stride);
        a0 += v.x;
        a1 += v.y;
      }
    } else if (i < P) {
      for (int s = 0; s < splits; ++s) {
        if (j < P) a0 += slabs[s * split_stride + i * ld + j];
        if (j + 1 < P) a1 += slabs[s * split_stride + i * ld + j + 1];
      }
    }
    if (prior && i < P) {
      if (prior_dense) {
        if (j < P) a0 += prior[i * P + j];
        if (j + 1 < P) a1 += prior[i * P + j + 1];
      } else {
        if (i == j) a0 += prior[i];
        if (i == j + 1) a1 += prior[i];
      }
    }
    dst[r * TB + c] = a0;
    dst[r * TB + c + 1] = a1;
  }
}

hipError_t launch_finalize_pack(hipStream_t st, const double *slabs, int32_t splits,
                                int64_t split_stride, int64_t P, int64_t ld, double *apk,
                                const double *prior, int32_t prior_dense, int64_t tile0, int64_t n_tiles) {
  const int64_t T = (P + TB - 1) / TB;
  if (n_tiles < 0) { tile0 = 0; n_tiles = T * (T + 1) / 2; }
  if (n_tiles == 0) return hipSuccess;
  dim3 grid((unsigned)n_tiles, 16);
  hipLaunchKernelGGL(finalize_pack_kernel, grid, dim3(256), 0, st, slabs, splits, split_stride, P, ld,
                     T, apk, prior, (int)prior_dense, tile0);
  return hipGetLastError();
}

// ---- prior: A += Lambda, g += Lambda (p - pbar), chi2 += (p-pbar)^T Lambda (p-pbar) ------------
__global__ __launch_bounds__(256) void prior_matrix_kernel(double *apk, int64_t P, int64_t T,
                                                           const double *prec, int dense) {
  int64_t t = blockIdx.x, tm = 0;
  while (t >= T - tm) { t -= T - tm; ++tm; }
  const int64_t tn = tm + t;
  if (!dense && tn != tm) return;
  double *dst = apk + (int64_t)blockIdx.x * TB * TB;
  const int c = threadIdx.x & 127;
  for (int rr = (threadIdx.x >> 7); rr < 8; rr += 2) {
    const int r = blockIdx.y * 8 + rr;
    const int64_t i = tm * TB + r, j = tn * TB + c;
    if (i < P && j < P) {
      if (dense) dst[r * TB + c] += prec[i * P + j];
      else if (i == j) dst[r * TB + c] += prec[i];
    }
  }
}

// t[j] = (Lambda d)_j, d = p - pbar; one wave per row j
__global__ __launch_bounds__(256) void prior_vec_kernel(int64_t P, const double *prec, int dense,
                                                        const double *pmean, const double *p,
                                                        double *tvec) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= P) return;
  if (!dense) {
    if (lane == 0) tvec[j] = prec[j] * (p[j] - pmean[j]);
    return;
  }
  double a = 0.0;
  for (int64_t k = lane; k < P; k += 64) a += prec[j * P + k] * (p[k] - pmean[k]);
  a = wsum(a);
  if (lane == 0) tvec[j] = a;
}

// g[j] += t[j]; g[P] += sum_j d_j t_j  (single workgroup, fixed order)
__global__ __launch_bounds__(256) void prior_apply_kernel(int64_t P, const double *pmean,
                                                          const double *p, const double *tvec,
                                                          double *gvec) {
  __shared__ double part[4];
  double a = 0.0;
  for (int64_t j = threadIdx.x; j < P; j += 256) {
    const double t = tvec[j];
    gvec[j] += t;
    a += (p[j] - pmean[j]) * t;
  }
  a = wsum(a);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) gvec[P] += part[0] + part[1] + part[2] + part[3];
}

hipError_t launch_add_prior(hipStream_t st, double *apk, int64_t P, const double *prec, int32_t dense,
                            const double *pmean, const double *p, double *tvec, double *gvec,
                            int32_t with_matrix, int32_t tvec_ready) {
  const int64_t T = (P + TB - 1) / TB;
  if (with_matrix) {
    dim3 grid((unsigned)(T * (T + 1) / 2), 16);
    hipLaunchKernelGGL(prior_matrix_kernel, grid, dim3(256), 0, st, apk, P, T, prec, dense);
  }
  if (!tvec_ready)    // (the caller's trial evaluation at this very point left Lambda (p - pbar) in tvec)
    hipLaunchKernelGGL(prior_vec_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, st, P, prec, dense,
                       pmean, p, tvec);
  hipLaunchKernelGGL(prior_apply_kernel, dim3(1), dim3(256), 0, st, P, pmean, p, tvec, gvec);
  return hipGetLastError();
}

// scalar[0] += sum_j (p-pbar)_j t_j   (single workgroup, fixed order)
__global__ __launch_bounds__(256) void prior_chi2_kernel(int64_t P, const double *pmean,
                                                         const double *p, const double *tvec,
                                                         double *scalar) {
  __shared__ double part[4];
  double a = 0.0;
  for (int64_t j = threadIdx.x; j < P; j += 256) a += (p[j] - pmean[j]) * tvec[j];
  a = wsum(a);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) scalar[0] += part[0] + part[1] + part[2] + part[3];
}

hipError_t launch_prior_chi2(hipStream_t st, int64_t P, const double *prec, int32_t dense,
                             const double *pmean, const double *p, double *tvec, double *scalar) {
  hipLaunchKernelGGL(prior_vec_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, st, P, prec, dense,
                     pmean, p, tvec);
  hipLaunchKernelGGL(prior_chi2_kernel, dim3(1), dim3(256), 0, st, P, pmean, p, tvec, scalar);
  return hipGetLastError();
}

// ---- parameter rows (data-prior cross-correlations) ---------------------------------------------
// Rows flagged row_param[i] = j >= 0 are not model rows: f_i = p_j (the prior entries of the
// reference's concat(y, prior) vector, src/lsqfit/_utilities.pyx:74-77).  Written after the model
// kernel, same conventions: rows inside a covariance block go unweighted to the raw buffers, the
// others weighted by wdiag; column P of a Jacobian row holds the residual.
__global__ __launch_bounds__(256) void param_rows_kernel(const int32_t *row_param, int64_t N, int64_t P, int64_t ld,
                                                         const double *p, const double *ymean, const double *wdiag,
                                                         const uint8_t *in_block, double *out_w, double *out_raw,
                                                         int jac, int64_t p_stride, int64_t out_stride) {
  const int64_t i = blockIdx.x;
  const int32_t j = row_param[i];
  if (j < 0) return;
  p += (int64_t)blockIdx.y * p_stride;          // blockIdx.y: one of many parameter points (lsqamd_chi2_points)
  out_w += (int64_t)blockIdx.y * out_stride;
  if (out_raw) out_raw += (int64_t)blockIdx.y * out_stride;
  const bool blk = in_block && in_block[i];
  const double w = blk ? 1.0 : wdiag[i];
  double *dst = blk ? out_raw : out_w;
  const double delta = w * (p[j] - ymean[i]);
  if (!jac) {
    if (threadIdx.x == 0) dst[i] = delta;
    return;
  }
  for (int64_t c = threadIdx.x; c <= P; c += 256) dst[i * ld + c] = c == P ? delta : (c == j ? w : 0.0);
}

hipError_t launch_param_rows(hipStream_t st, const int32_t *row_param, int64_t N, int64_t P, int64_t ld,
                             const double *p, const double *ymean, const double *wdiag, const uint8_t *in_block,
                             double *out_w, double *out_raw, int jac, int32_t n_batch, int64_t p_stride, int64_t out_stride) {
  if (N <= 0) return hipSuccess;
  hipLaunchKernelGGL(param_rows_kernel, dim3((unsigned)N, (unsigned)(n_batch < 1 ? 1 : n_batch)), dim3(jac ? 256 : 64), 0, st,
                     row_param, N, P, ld, p, ymean, wdiag, in_block, out_w, out_raw, jac, p_stride, out_stride);
  return hipGetLastError();
}

// ---- packed tiles -> dense working copies ------------------------------------------------------
// M[i][j] = A[i][j] + mu d_i^2 [i==j] for the upper tiles; M[i][P] = g[i].  frozen (nullable):
// parameters with frozen[i] != 0 are taken out of the system (unit row/column, zero right-hand side)
__global__ __launch_bounds__(256) void build_damped_kernel(const double *apk, int64_t P, int64_t T,
                                                           int64_t ld, double mu_arg, const double *diag,
                                                           const double *g, double *M, const double *frozen,
                                                           const double *mu_dev, int32_t *info_zero) {
  const double mu = mu_dev ? *mu_dev : mu_arg;   // the device-resident LM state (a captured step must not bake mu in)
  if (info_zero && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *info_zero = 0;   // (saves the factorisation its 4-byte memset)
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
      if (i == j && mu != 0.0) {  // (mu = 0 callers may pass a D that is not set up yet)
        const double d = diag[i];
        v += mu * d * d;
      }
      if (frozen && (frozen[i] != 0.0 || frozen[j] != 0.0)) v = i == j ? 1.0 : 0.0;
      M[i * ld + j] = v;
    }
    if (tn == T - 1 && c == 0 && i < P && g) M[i * ld + P] = (frozen && frozen[i] != 0.0) ? 0.0 : g[i];
  }
}

hipError_t launch_build_damped(hipStream_t st, const double *apk, int64_t P, int64_t ld, double mu,
                               const double *diag, const double *g, double *Mout, const double *frozen,
                               const double *mu_dev, int32_t *info_zero) {
  const int64_t T = (P + TB - 1) / TB;
  dim3 grid((unsigned)(T * (T + 1) / 2), 16);
  hipLaunchKernelGGL(build_damped_kernel, grid, dim3(256), 0, st, apk, P, T, ld, mu, diag, g, Mout, frozen, mu_dev, info_zero);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void packed_diag_kernel(const double *apk, int64_t P, int64_t T,
                                                          double *out) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= P) return;
  const int64_t tm = j / TB, r = j % TB;
  out[j] = apk[packed_tile_index(tm, tm, T) * TB * TB + r * TB + r];
}

hipError_t launch_packed_diag(hipStream_t st, const double *apk, int64_t P, double *out) {
  const int64_t T = (P + TB - 1) / TB;
  hipLaunchKernelGGL(packed_diag_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, apk, P,
                     T, out);
  return hipGetLastError();
}

// full symmetric P x P (ld) from the packed upper tiles
__global__ __launch_bounds__(256) void unpack_sym_kernel(const double *apk, int64_t P, int64_t T,
                                                         double *out, int64_t ld) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = blockIdx.y;
  if (j >= P) return;
  const int64_t a = i < j ? i : j, b = i < j ? j : i;  // a <= b : upper element (a, b)
  const int64_t tm = a / TB, tn = b / TB;
  out[i * ld + j] = apk[packed_tile_index(tm, tn, T) * TB * TB + (a % TB) * TB + (b % TB)];
}

hipError_t launch_unpack_sym(hipStream_t st, const double *apk, int64_t P, double *out, int64_t ld) {
  const int64_t T = (P + TB - 1) / TB;
  dim3 grid((unsigned)((P + 255) / 256), (unsigned)P);
  hipLaunchKernelGGL(unpack_sym_kernel, grid, dim3(256), 0, st, apk, P, T, out, ld);
  return hipGetLastError();
}

// A[j][i] = A[i][j] for j > i (dense, in place)
__global__ __launch_bounds__(256) void symmetrize_kernel(double *A, int64_t P, int64_t ld) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = blockIdx.y;
  if (j >= P || j >= i) return;
  A[i * ld + j] = A[j * ld + i];
}

hipError_t launch_symmetrize_from_upper(hipStream_t st, double *A, int64_t P, int64_t ld) {
  dim3 grid((unsigned)((P + 255) / 256), (unsigned)P);
  hipLaunchKernelGGL(symmetrize_kernel, grid, dim3(256), 0, st, A, P, ld);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void copy_strided_kernel(const double *src, int64_t lds_,
                                                           double *dst, int64_t ldd, int64_t rows,
                                                           int64_t cols) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= cols) return;
  for (int64_t i = blockIdx.y; i < rows; i += gridDim.y) dst[i * ldd + j] = src[i * lds_ + j];
}

__global__ __launch_bounds__(256) void add_rows_kernel(double *dst, const double *src, int64_t rows, int64_t cols2, int64_t ld) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= cols2) return;
  for (int64_t i = blockIdx.y; i < rows; i += gridDim.y) {
    v2d *d = reinterpret_cast<v2d *>(dst + i * ld) + j;
    const v2d a = *d, b = reinterpret_cast<const v2d *>(src + i * ld)[j];
    *d = a + b;
  }
}

hipError_t launch_add_rows(hipStream_t st, double *dst, const double *src, int64_t rows, int64_t cols, int64_t ld) {
  if (rows <= 0 || cols <= 0) return hipSuccess;
  if ((cols & 1) || (ld & 1)) return hipErrorInvalidValue;
  dim3 grid((unsigned)((cols / 2 + 255) / 256), (unsigned)(rows < 16384 ? rows : 16384));
  hipLaunchKernelGGL(add_rows_kernel, grid, dim3(256), 0, st, dst, src, rows, cols / 2, ld);
  return hipGetLastError();
}

// dst[i] = src[i * ld] (a column) and zbuf[0 .. zwords) = 0 in one launch: the right-hand side of the back
// substitution and the zeroed hand-off granules its chain needs, instead of a copy and a memset
__global__ __launch_bounds__(256) void copy_column_zero_kernel(const double *src, int64_t ld, double *dst, int64_t n,
                                                               unsigned long long *zbuf, int64_t zwords) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i * ld];
  for (int64_t k = i; k < zwords; k += (int64_t)gridDim.x * 256) zbuf[k] = 0ull;
}

hipError_t launch_copy_column_zero(hipStream_t st, const double *src, int64_t ld, double *dst, int64_t n,
                                   void *zbuf, size_t zbytes) {
  hipLaunchKernelGGL(copy_column_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, ld, dst, n,
                     static_cast<unsigned long long *>(zbuf), (int64_t)(zbytes / 8));
  return hipGetLastError();
}

hipError_t launch_copy_strided(hipStream_t st, const double *src, int64_t lds_, double *dst,
                               int64_t ldd, int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return hipSuccess;
  dim3 grid((unsigned)((cols + 255) / 256), (unsigned)(rows < 32768 ? rows : 32768));
  hipLaunchKernelGGL(copy_strided_kernel, grid, dim3(256), 0, st, src, lds_, dst, ldd, rows, cols);
  return hipGetLastError();
}

// dst[c][r] = src[r][c] * (skip && skip[r] ? 1 : scale[r])  -- 64 x 64 tiles through LDS so
// both the read and the write are full 512-byte row segments.  batch: blockIdx.z.
__global__ __launch_bounds__(256) void transpose_scale_kernel(const double *src, int64_t lds_,
                                                              double *dst, int64_t ldd, int64_t rows,
                                                              int64_t cols, const double *scale,
                                                              const uint8_t *skip, int64_t s_src,
                                                              int64_t s_dst) {
  __shared__ double tile[64][65];
  src += (int64_t)blockIdx.z * s_src;
  dst += (int64_t)blockIdx.z * s_dst;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int64_t r = r0 + i, c = c0 + tx;
    double v = 0.0;
    if (r < rows && c < cols) {
      v = src[r * lds_ + c];
      if (scale && !(skip && skip[r])) v *= scale[r];
    }
    tile[i][tx] = v;
  }
  __syncthreads();
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int64_t c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[c * ldd + r] = tile[tx][i];
  }
}

hipError_t launch_transpose_scale(hipStream_t st, const double *src, int64_t lds_, double *dst,
                                  int64_t ldd, int64_t rows, int64_t cols, const double *scale,
                                  const uint8_t *skip, int64_t batch, int64_t s_src, int64_t s_dst) {
  if (rows <= 0 || cols <= 0 || batch <= 0) return hipSuccess;
  dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64), (unsigned)batch);
  hipLaunchKernelGGL(transpose_scale_kernel, grid, dim3(256), 0, st, src, lds_, dst, ldd, rows, cols,
                     scale, skip, s_src, s_dst);
  return hipGetLastError();
}

// dst[r][c] = src[r][c] * (scale ? scale[r] : 1) for the rows with skip[r] == 0
__global__ __launch_bounds__(256) void rows_scale_copy_kernel(const double *src, int64_t lds_,
                                                              double *dst, int64_t ldd, int64_t rows,
                                                              int64_t cols, const double *scale,
                                                              const uint8_t *skip) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= cols) return;
  for (int64_t i = blockIdx.y; i < rows; i += gridDim.y) {
    if (skip && skip[i]) continue;
    const double s = scale ? scale[i] : 1.0;
    dst[i * ldd + j] = s * src[i * lds_ + j];
  }
}

hipError_t launch_rows_scale_copy(hipStream_t st, const double *src, int64_t lds_, double *dst,
                                  int64_t ldd, int64_t rows, int64_t cols, const double *scale,
                                  const uint8_t *skip) {
  if (rows <= 0 || cols <= 0) return hipSuccess;
  dim3 grid((unsigned)((cols + 255) / 256), (unsigned)(rows < 32768 ? rows : 32768));
  hipLaunchKernelGGL(rows_scale_copy_kernel, grid, dim3(256), 0, st, src, lds_, dst, ldd, rows, cols,
                     scale, skip);
  return hipGetLastError();
}

// ---- chi2 of many parameter points -------------------------------------------
__device__ __forceinline__ double wsum64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// out[m] (+)= sum_i r[m*stride + i]^2 ; one workgroup per point
__global__ __launch_bounds__(256) void rows_sumsq_kernel(const double *r, int64_t n, int64_t stride,
                                                         double *out, int accumulate) {
  __shared__ double sh[4];
  const double *rm = r + (int64_t)blockIdx.x * stride;
  double a = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const double v = rm[i];
    a += v * v;
  }
  a = wsum64(a);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = sh[0] + sh[1] + sh[2] + sh[3];
    out[blockIdx.x] = accumulate ? out[blockIdx.x] + t : t;
  }
}

hipError_t launch_rows_sumsq(hipStream_t st, const double *r, int64_t n, int64_t stride, int64_t m,
                             double *out, int accumulate) {
  if (m <= 0) return hipSuccess;
  hipLaunchKernelGGL(rows_sumsq_kernel, dim3((unsigned)m), dim3(256), 0, st, r, n, stride, out, accumulate);
  return hipGetLastError();
}

// out[m] += sum_j prec[j] (p[m][j] - pmean[j])^2
__global__ __launch_bounds__(256) void prior_chi2_points_diag_kernel(int64_t P, const double *prec,
                                                                     const double *pmean,
                                                                     const double *p, double *out) {
  __shared__ double sh[4];
  const double *pm = p + (int64_t)blockIdx.x * P;
  double a = 0.0;
  for (int64_t j = threadIdx.x; j < P; j += 256) {
    const double d = pm[j] - pmean[j];
    a += prec[j] * d * d;
  }
  a = wsum64(a);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] += sh[0] + sh[1] + sh[2] + sh[3];
}

// Dt[k][m] = p[m][k] - pmean[k]
__global__ __launch_bounds__(256) void delta_transpose_kernel(const double *p, const double *pmean,
                                                              double *Dt, int64_t ldt, int64_t M,
                                                              int64_t P) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t k = blockIdx.y;
  if (m < M) Dt[k * ldt + m] = p[m * P + k] - pmean[k];
}

// out[m] += sum_k A[k][m] * B[k][m]
__global__ __launch_bounds__(256) void coldot_accum_kernel(const double *A, const double *B, int64_t ld,
                                                           int64_t K, int64_t M, double *out) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  double a = 0.0;
  for (int64_t k = 0; k < K; ++k) a += A[k * ld + m] * B[k * ld + m];
  out[m] += a;
}

hipError_t launch_prior_chi2_points(hipStream_t st, int64_t P, const double *prec, int32_t dense,
                                    const double *pmean, const double *p, int64_t m, double *Dt,
                                    double *T, int64_t ldt, double *out) {
  if (m <= 0) return hipSuccess;
  if (!dense) {
    hipLaunchKernelGGL(prior_chi2_points_diag_kernel, dim3((unsigned)m), dim3(256), 0, st, P, prec, pmean,
                       p, out);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(delta_transpose_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)P), dim3(256), 0,
                     st, p, pmean, Dt, ldt, m, P);
  GemmTN g;  // T = Lambda . Dt  (Lambda symmetric: its rows are the k-major operand)
  g.X = prec; g.ldx = P; g.Y = Dt; g.ldy = ldt; g.C = T; g.ldc = ldt;
  g.M = P; g.N = m; g.K = P;
  hipError_t e = launch_gemm_tn(st, g);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(coldot_accum_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, Dt, T, ldt,
                     P, m, out);
  return hipGetLastError();
}

// y[i] = sum_j A[i][j] x[j] : one wave per row
__global__ __launch_bounds__(256) void gemv_rows_kernel(const double *A, int64_t ld, int64_t rows,
                                                        int64_t cols, const double *x, double *y) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= rows) return;
  const double *a = A + i * ld;
  double acc = 0.0;
  for (int64_t j = lane; j < cols; j += 64) acc += a[j] * x[j];
  acc = wsum64(acc);
  if (lane == 0) y[i] = acc;
}

hipError_t launch_gemv_rows(hipStream_t st, const double *A, int64_t ld, int64_t rows, int64_t cols,
                            const double *x, double *y) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(gemv_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, A, ld, rows,
                     cols, x, y);
  return hipGetLastError();
}

// ---- device-resident pieces of the LM step (no host-to-device traffic inside a step) ------------
// scaling.c on the device: D from the column norms sqrt(diag(J^T J)) (coln2 = their squares)
__global__ __launch_bounds__(256) void scale_update_kernel(int64_t P, int scaler, int init,
                                                           const double *coln2, double *dscale) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= P) return;
  const double c2 = coln2[j];
  const double cn = sqrt(c2 > 0.0 ? c2 : 0.0);
  double d;
  if (scaler == LSQAMD_SCALE_LEVENBERG) d = init ? 1.0 : dscale[j];
  else if (scaler == LSQAMD_SCALE_MORE) d = init ? (cn == 0.0 ? 1.0 : cn) : fmax(dscale[j], cn);
  else d = cn == 0.0 ? 1.0 : cn;
  dscale[j] = d;
}

hipError_t launch_scale_update(hipStream_t st, int64_t P, int scaler, int init, const double *coln2,
                               double *dscale) {
  hipLaunchKernelGGL(scale_update_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, P, scaler,
                     init, coln2, dscale);
  return hipGetLastError();
}

// x_trial = x - v
__global__ __launch_bounds__(256) void trial_point_kernel(int64_t P, const double *x, const double *v,
                                                          double *xt) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < P) xt[j] = x[j] - v[j];
}

hipError_t launch_trial_point(hipStream_t st, int64_t P, const double *x, const double *v, double *xt) {
  hipLaunchKernelGGL(trial_point_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, P, x, v, xt);
  return hipGetLastError();
}

// ---- LM state on the device (plain lm): what trust.c / nielsen.c / convergence.c do with O(P)
// vectors and a handful of scalars, as three one-workgroup kernels.  The host reads ONE small
// status record per trial and one per accepted step; x, g, D, v never leave the device.
//   st[LMS_*]: the record (doubles; also copied to the host)
__device__ __forceinline__ double block_sum(double a, double *sh) {
  a = wsum64(a);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// the record, mirrored into pinned host memory by the kernel that completes a half step: the host polls the mirror instead
// of sleeping in a stream synchronisation -- no copy kernel (4 us + a dependent-launch gap each, 2.6 per step)
__device__ __forceinline__ void lm_publish(double *st) {
  // called by ONE thread after it has written the record.  Stores to host memory become visible to the CPU in no
  // particular order -- a system-scope fence between two of them does not help (measured on the one-launch fit's larger
  // block, jit.hip lm_fit) -- so the mirror is self-verifying: word LMS_CHECK carries lm_record_checksum of the other
  // fifteen, the new sequence number among them; the host believes a snapshot only when the sum fits (api.hip wait_record).
  const long long hp = __double_as_longlong(st[LMS_HOSTPTR]);
  const double seq = st[LMS_SEQ] + 1.0;
  st[LMS_SEQ] = seq;
  if (hp == 0) return;
  unsigned long long w[LMS_COUNT];
#pragma unroll
  for (int i = 0; i < LMS_COUNT; ++i) w[i] = (unsigned long long)__double_as_longlong(st[i]);
  w[LMS_CHECK] = lm_record_checksum(w);
  st[LMS_CHECK] = __longlong_as_double((long long)w[LMS_CHECK]);
  typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
  u2 *h = reinterpret_cast<u2 *>(hp);           // (128-byte aligned: lsqamd_create)
#pragma unroll
  for (int i = 0; i < LMS_COUNT; i += 2)        // 16-byte stores: eight posted writes instead of sixteen
    __builtin_nontemporal_store((u2){w[i], w[i + 1]}, h + i / 2);
  __threadfence_system();
}

// trial point + the two dot products of the gain ratio:  xt = x - v,  st[VG] = v.g,  st[DV2] = |D v|^2,
// st[VFINITE] = 1 when every component of v is finite
__global__ __launch_bounds__(256) void lm_trial_kernel(int64_t P, const double *x, const double *v, const double *g,
                                                       const double *d, double *xt, double *st, const double *U, int64_t ldu,
                                                       const double *a_diag) {
  __shared__ double sh[4];
  double vg = 0.0, dv2 = 0.0, bad = 0.0, pmin = INFINITY;
  const double mu = st[LMS_MU];
  for (int64_t j = threadIdx.x; j < P; j += 256) {
    const double vj = v[j];
    xt[j] = x[j] - vj;
    vg += vj * g[j];
    const double t = d[j] * vj;
    dv2 += t * t;
    bad += (vj - vj == 0.0) ? 0.0 : 1.0;   // NaN / inf
    if (U) {
      const double u = U[j * ldu + j], m = a_diag[j] + mu * d[j] * d[j];
      const double r = m > 0.0 ? u * u / m : 1.0;
      pmin = r < pmin ? r : pmin;          // (a NaN pivot never wins: the factorisation's own status reports it)
    }
  }
  vg = block_sum(vg, sh);
  dv2 = block_sum(dv2, sh);
  bad = block_sum(bad, sh);
  if (U) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pmin = fmin(pmin, __shfl_xor(pmin, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = pmin;
    __syncthreads();
    pmin = fmin(fmin(sh[0], sh[1]), fmin(sh[2], sh[3]));
  }
  if (threadIdx.x == 0) {
    st[LMS_VG] = vg;
    st[LMS_DV2] = dv2;
    st[LMS_VFINITE] = bad == 0.0 ? 1.0 : 0.0;
    st[LMS_PIVMIN] = U ? pmin : 1.0;
  }
}

hipError_t launch_lm_trial(hipStream_t stream, int64_t P, const double *x, const double *v, const double *g,
                           const double *d, double *xt, double *st, const double *U, int64_t ldu, const double *a_diag) {
  hipLaunchKernelGGL(lm_trial_kernel, dim3(1), dim3(256), 0, stream, P, x, v, g, d, xt, st, U, ldu, a_diag);
  return hipGetLastError();
}

// trust_iterate's decision (trust.c) and nielsen.c's update of mu:
//   rho = (1 - |f_t|^2 / |f|^2) / ((v.g + mu |D v|^2) / |f|^2)   [|J v|^2 = v.g - mu |D v|^2]
__global__ void lm_decide_kernel(const double *chi2_trial, const int32_t *chol_info, double factor_up,
                                 double factor_down, double *st) {
  const double chi2 = st[LMS_CHI2], mu = st[LMS_MU];
  const double ct = chi2_trial[0];
  const bool solved = chol_info[0] == 0 && st[LMS_VFINITE] != 0.0;
  double rho = -1.0;
  if (solved) {
    const double normf = sqrt(chi2), normf_t = sqrt(ct);
    if (normf_t < normf) {   // NaN-safe: anything else rejects
      const double u = normf_t / normf;
      const double pred = (st[LMS_VG] + mu * st[LMS_DV2]) / chi2;
      rho = pred > 0.0 ? (1.0 - u * u) / pred : -1.0;
    }
  }
  if (rho > 0.75) st[LMS_DELTA] *= factor_up;
  else if (rho < 0.25) st[LMS_DELTA] /= factor_down;
  if (rho > 0.0) {
    const double b = 2.0 * rho - 1.0;
    st[LMS_MU] = mu * fmax(0.333333333333333, 1.0 - b * b * b);
    st[LMS_NU] = 2.0;
  } else {
    st[LMS_MU] = mu * st[LMS_NU];
    st[LMS_NU] *= 2.0;
  }
  st[LMS_RHO] = rho;
  st[LMS_CHI2_TRIAL] = ct;
  st[LMS_ACCEPT] = rho > 0.0 ? 1.0 : 0.0;
  st[LMS_SOLVED] = solved ? 1.0 : 0.0;
  lm_publish(st);
}

// The tail of a trial in ONE launch (single rank: no exchange in between): the second stage of |f_trial|^2,
// the prior's share of chi2 and the decision -- the same sums in the same order as sum_stage2,
// prior_chi2_kernel and lm_decide_kernel, which the multi-rank path keeps (it all-reduces the scalar first).
__global__ __launch_bounds__(256) void lm_trial_tail_kernel(const double *partial, int n_partial, int64_t P,
                                                            const double *pmean, const double *p, const double *tvec,
                                                            double *chi2_out, const int32_t *chol_info,
                                                            double factor_up, double factor_down, double *st) {
  __shared__ double part[4];
  double a = 0.0;
  for (int i = threadIdx.x; i < n_partial; i += 256) a += partial[i];
  a = wsum(a);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
  __syncthreads();
  double ct = part[0] + part[1] + part[2] + part[3];
  __syncthreads();
  if (tvec) {
    double b = 0.0;
    for (int64_t j = threadIdx.x; j < P; j += 256) b += (p[j] - pmean[j]) * tvec[j];
    b = wsum(b);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = b;
    __syncthreads();
    ct += part[0] + part[1] + part[2] + part[3];
  }
  if (threadIdx.x != 0) return;
  chi2_out[0] = ct;
  const double chi2 = st[LMS_CHI2], mu = st[LMS_MU];
  const bool solved = chol_info[0] == 0 && st[LMS_VFINITE] != 0.0;
  double rho = -1.0;
  if (solved) {
    const double normf = sqrt(chi2), normf_t = sqrt(ct);
    if (normf_t < normf) {   // NaN-safe: anything else rejects
      const double u = normf_t / normf;
      const double pred = (st[LMS_VG] + mu * st[LMS_DV2]) / chi2;
      rho = pred > 0.0 ? (1.0 - u * u) / pred : -1.0;
    }
  }
  if (rho > 0.75) st[LMS_DELTA] *= factor_up;
  else if (rho < 0.25) st[LMS_DELTA] /= factor_down;
  if (rho > 0.0) {
    const double b = 2.0 * rho - 1.0;
    st[LMS_MU] = mu * fmax(0.333333333333333, 1.0 - b * b * b);
    st[LMS_NU] = 2.0;
  } else {
    st[LMS_MU] = mu * st[LMS_NU];
    st[LMS_NU] *= 2.0;
  }
  st[LMS_RHO] = rho;
  st[LMS_CHI2_TRIAL] = ct;
  st[LMS_ACCEPT] = rho > 0.0 ? 1.0 : 0.0;
  st[LMS_SOLVED] = solved ? 1.0 : 0.0;
  lm_publish(st);
}

// |r|^2 first stage + (prior: t = Lambda (p - pbar)) + the tail above
hipError_t launch_lm_trial_tail(hipStream_t st, const double *r, int64_t n, double *partial, int64_t P,
                                const double *prec, int32_t dense, const double *pmean, const double *p, double *tvec,
                                bool with_prior, double *chi2_out, const int32_t *chol_info, double factor_up,
                                double factor_down, double *lmd) {
  int blocks = (int)((n + 1023) / 1024);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sumsq_stage1, dim3(blocks), dim3(256), 0, st, r, n, partial);
  if (with_prior)
    hipLaunchKernelGGL(prior_vec_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, st, P, prec, dense, pmean, p, tvec);
  hipLaunchKernelGGL(lm_trial_tail_kernel, dim3(1), dim3(256), 0, st, partial, blocks, P, pmean, p,
                     with_prior ? tvec : nullptr, chi2_out, chol_info, factor_up, factor_down, lmd);
  return hipGetLastError();
}



// ---- few parameters: the sums of the fused normal-equation kernel (jit.hip lsqamd_jit_nrm) into their places -----
// q = [J^T J upper row-major | J^T f | |f|^2] -> the packed 128 x 128 tile (both triangles, prior precision added on the
// way like finalize_pack does), gvec = [J^T f ; chi2]
__global__ __launch_bounds__(256) void nrm_unpack_kernel(const double *q, int P, double *apk, double *gvec, const double *prior,
                                                         int prior_dense) {
  const int NA = P * (P + 1) / 2;
  for (int e = threadIdx.x; e < P * P; e += 256) {
    const int i = e / P, j = e % P;
    const int a = i < j ? i : j, b = i < j ? j : i;
    double v = q[a * P - a * (a - 1) / 2 + (b - a)];
    if (prior) v += prior_dense ? prior[i * P + j] : (i == j ? prior[i] : 0.0);
    apk[i * TB + j] = v;
  }
  for (int j = threadIdx.x; j <= P; j += 256) gvec[j] = q[NA + j];
}

hipError_t launch_nrm_unpack(hipStream_t st, const double *q, int64_t P, double *apk, double *gvec, const double *prior,
                             int32_t prior_dense) {
  hipLaunchKernelGGL(nrm_unpack_kernel, dim3(1), dim3(256), 0, st, q, (int)P, apk, gvec, prior, (int)prior_dense);
  return hipGetLastError();
}


// ---- a dozen parameters at most: a whole trial solve by ONE WAVE, the matrix in registers ------------------------
// (A + mu D^2) v = g for P <= 12: lane j holds column j of the upper triangle (12 registers), lane 12 the right-hand
// side; the factorisation is 12 unrolled steps of shuffles and FMAs (the forward substitution rides along in lane
// 12), the back substitution 12 wave sums -- no LDS, no barrier, ~200 shuffles in all -- followed by the trial point
// and the record's dot products.  Replaces build_damped + diagonal-block kernel + row panel + column copy + back
// substitution + lm_trial (six dependent launches, 50 us) for the fits lsqfit is used for every day.
constexpr int T12 = 12;
__global__ __launch_bounds__(64) void lm_tiny12_solve_kernel(const double *apk, int P, const double *g, const double *d,
                                                             const double *x, double *xt, double *v_out, double *st,
                                                             int32_t *chol_info, int watch) {
  const int lane = threadIdx.x;
  const double mu = st[LMS_MU];
  double m[T12];
  double diag0 = 1.0;                     // the damped diagonal entry of this lane's column, before the factorisation
#pragma unroll
  for (int i = 0; i < T12; ++i) {
    double v = 0.0;
    if (lane < T12) {
      if (lane < P && i <= lane && i < P) {
        v = apk[i * TB + lane];           // (P <= 128: the packed form is ONE 128 x 128 tile)
        if (i == lane) { v += mu * d[i] * d[i]; diag0 = v; }
      } else if (i == lane) v = 1.0;      // padding: unit diagonal
    } else if (lane == T12 && i < P) v = g[i];
    m[i] = v;
  }
  int fail = 0;
  double pmin = INFINITY;
#pragma unroll
  for (int k = 0; k < T12; ++k) {
    const double pk = __shfl(m[k], k, 64);
    if (!(pk > 0.0) && fail == 0) fail = k + 1;               // uniform
    const double uk = sqrt(pk > 0.0 ? pk : 1.0), inv = 1.0 / uk;
    if (lane == k) { pmin = pk / diag0 < pmin ? pk / diag0 : pmin; m[k] = uk; }
    else if (lane > k) m[k] *= inv;
#pragma unroll
    for (int i = k + 1; i < T12; ++i) {
      const double ui = __shfl(m[k], i, 64);
      if (lane >= i) m[i] -= ui * m[k];
    }
  }
  // back substitution: y lives in lane 12, v_k ends up in lane k
  double vl = 0.0;
#pragma unroll
  for (int k = T12 - 1; k >= 0; --k) {
    double t = (lane > k && lane < T12) ? m[k] * vl : 0.0;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);   // lanes 0..15 hold everything that matters
    const double yk = __shfl(m[k], T12, 64), ukk = __shfl(m[k], k, 64);
    const double vk = (yk - __shfl(t, 0, 64)) / ukk;
    if (lane == k) vl = vk;
  }
  const bool bad = fail != 0;
  double vg = 0.0, dv2 = 0.0, nf = 0.0;
  if (lane < P) {
    const double vj = bad ? NAN : vl;
    v_out[lane] = vj;
    xt[lane] = x[lane] - vj;
    vg = vj * g[lane];
    const double t = d[lane] * vj;
    dv2 = t * t;
    nf = (vj - vj == 0.0) ? 0.0 : 1.0;
  } else {
    pmin = INFINITY;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    vg += __shfl_xor(vg, o, 64);
    dv2 += __shfl_xor(dv2, o, 64);
    nf += __shfl_xor(nf, o, 64);
    pmin = fmin(pmin, __shfl_xor(pmin, o, 64));
  }
  if (lane == 0) {
    chol_info[0] = fail;
    st[LMS_VG] = vg;
    st[LMS_DV2] = dv2;
    st[LMS_VFINITE] = nf == 0.0 ? 1.0 : 0.0;
    st[LMS_PIVMIN] = (watch && !bad) ? pmin : 1.0;
  }
}

hipError_t launch_lm_tiny12_solve(hipStream_t stream, const double *apk, int64_t P, const double *g, const double *d,
                                  const double *x, double *xt, double *v_out, double *st, int32_t *chol_info, int watch) {
  if (P < 1 || P > T12) return hipErrorInvalidValue;
  hipLaunchKernelGGL(lm_tiny12_solve_kernel, dim3(1), dim3(64), 0, stream, apk, (int)P, g, d, x, xt, v_out, st, chol_info, watch);
  return hipGetLastError();
}

// ---- 13 .. 32 parameters: the same one-launch trial solve, one wave, T = 16 / 24 / 32 elimination steps ----------------
// The formulation of the whole-fit kernel's solve (jit.hip lm_solve): lane j holds column j of the upper triangle, lane T
// the right-hand side; every cross-lane read names its lane at compile time (v_readlane: the value arrives in scalar
// registers), pivots are inverted by v_rsq_f64 + two Newton steps, the back substitution runs on a wave-uniform copy of
// the right-hand side.  P < T: unit diagonal in the padding.  Outputs as lm_tiny12_solve_kernel.
__device__ __forceinline__ double rl_d(double v, int lane) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}

__device__ __forceinline__ double rsqrt_newton2(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double h = -0.5 * d;
  double e = __builtin_fma(h * y, y, 0.5);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(h * y, y, 0.5);
  return __builtin_fma(y, e, y);
}

template <int T>
__global__ __launch_bounds__(64) void lm_small_solve_kernel(const double *apk, int P, const double *g, const double *d,
                                                            const double *x, double *xt, double *v_out, double *st,
                                                            int32_t *chol_info, int watch) {
  const int lane = threadIdx.x;
  const double mu = st[LMS_MU];
  const int col = lane < P ? lane : P - 1;
  const double dl = d[col], gl = g[col];
  // the damped matrix and the right-hand side (column T) go through LDS: filled element-parallel (padding included), read back
  // as one unconditional load per register -- loaded straight from the tile, every register came with its own address
  // clamp and three lane masks, all set up front: 1200 scalar-register spills
  __shared__ double sM[T * (T + 1)];
  for (int e = lane; e < T * (T + 1); e += 64) {
    const int i = e / (T + 1), j = e % (T + 1);
    double v = 0.0;
    if (i < P) {
      if (j == T) v = g[i];
      else if (j < P && i <= j) {
        v = apk[i * TB + j];
        if (i == j) { const double di = d[i]; v += mu * di * di; }
      }
    } else if (i == j) v = 1.0;                  // padding: unit diagonal
    sM[e] = v;
  }
  __syncthreads();
  double m[T], uinv[T];
  const int jc = lane < T ? lane : T;            // (lanes beyond T: a copy of the right-hand side, never read)
#pragma unroll
  for (int i = 0; i < T; ++i) m[i] = sM[i * (T + 1) + jc];
  const double diag0 = lane < P ? sM[lane * (T + 1) + lane] : 1.0;
  int fail = 0;
  double pmin = INFINITY;
#pragma unroll
  for (int k = 0; k < T; ++k) {
    const double pk = rl_d(m[k], k);
    if (!(pk > 0.0) && fail == 0) fail = k + 1;
    const double inv = rsqrt_newton2(pk > 0.0 ? pk : 1.0);
    uinv[k] = inv;
    if (lane == k) pmin = pk;
    // (unconditional: what this writes below the diagonal -- lanes < i of m[i] -- is never read; a compare and two
    //  selects per update saved, half the instructions of the sweep)
    m[k] *= inv;
#pragma unroll
    for (int i = k + 1; i < T; ++i) m[i] = __builtin_fma(-rl_d(m[k], i), m[k], m[i]);
  }
  // (the compiler sees that the back substitution broadcasts the values the sweep broadcast already and keeps all 500 of them
  //  alive in spilled scalar registers, two v_writelane and two v_readlane each: cheaper to broadcast them again)
#pragma unroll
  for (int i = 0; i < T; ++i) asm volatile("" : "+v"(m[i]));
  double y[T], v[T];
#pragma unroll
  for (int i = 0; i < T; ++i) y[i] = rl_d(m[i], T);
#pragma unroll
  for (int k = T - 1; k >= 0; --k) {
    v[k] = y[k] * uinv[k];
#pragma unroll
    for (int i = 0; i < k; ++i) y[i] -= rl_d(m[i], k) * v[k];
  }
  const bool bad = fail != 0;
  double vl = 0.0, vg = 0.0, dv2 = 0.0, nf = 0.0;
#pragma unroll
  for (int k = 0; k < T; ++k) {
    vl = lane == k ? v[k] : vl;
    const double gk = rl_d(gl, k < 63 ? k : 63), dk = rl_d(dl, k < 63 ? k : 63);
    if (k < P) {                                             // (uniform)
      vg = __builtin_fma(v[k], gk, vg);
      const double t = dk * v[k];
      dv2 = __builtin_fma(t, t, dv2);
      nf += (v[k] - v[k] == 0.0) ? 0.0 : 1.0;
    }
  }
  if (bad) { vl = NAN; nf = 1.0; }
  if (lane < P) {
    v_out[lane] = vl;
    xt[lane] = x[lane] - vl;
    pmin = pmin / diag0;
  } else {
    pmin = INFINITY;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pmin = fmin(pmin, __shfl_xor(pmin, o, 64));
  if (lane == 0) {
    chol_info[0] = fail;
    st[LMS_VG] = vg;
    st[LMS_DV2] = dv2;
    st[LMS_VFINITE] = nf == 0.0 ? 1.0 : 0.0;
    st[LMS_PIVMIN] = (watch && !bad) ? pmin : 1.0;
  }
}

hipError_t launch_lm_small_solve(hipStream_t stream, const double *apk, int64_t P, const double *g, const double *d,
                                 const double *x, double *xt, double *v_out, double *st, int32_t *chol_info, int watch) {
  if (P < 1 || P > 32) return hipErrorInvalidValue;
  if (P <= 16) hipLaunchKernelGGL(lm_small_solve_kernel<16>, dim3(1), dim3(64), 0, stream, apk, (int)P, g, d, x, xt, v_out, st, chol_info, watch);
  else if (P <= 24) hipLaunchKernelGGL(lm_small_solve_kernel<24>, dim3(1), dim3(64), 0, stream, apk, (int)P, g, d, x, xt, v_out, st, chol_info, watch);
  else hipLaunchKernelGGL(lm_small_solve_kernel<32>, dim3(1), dim3(64), 0, stream, apk, (int)P, g, d, x, xt, v_out, st, chol_info, watch);
  return hipGetLastError();
}

// ---- small fits: the tail of a trial in ONE single-workgroup launch ------------------------------------------
// |f_trial|^2 over n <= 65536 residuals, the prior's share for a diagonal (or absent) prior -- t = Lambda (p - pbar)
// is left in tvec for the accepted branch, as prior_vec_kernel does -- and the decision of lm_trial_tail_kernel:
// instead of sumsq_stage1 + prior_vec + lm_trial_tail (three dependent launches).
__global__ __launch_bounds__(256) void lm_trial_tail_small_kernel(const double *r, int64_t n, int64_t P, const double *prec,
                                                                  const double *pmean, const double *p, double *tvec,
                                                                  double *chi2_out, const int32_t *chol_info,
                                                                  double factor_up, double factor_down, double *st) {
  __shared__ double part[4];
  __shared__ double blk[64];
  // the SAME sums in the same order as sumsq_stage1 (one workgroup per 1024 residuals, grid-strided) followed by the
  // second stage of lm_trial_tail_kernel: a fit takes bit-identical decisions with or without this fused tail
  const int blocks = (int)((n + 1023) / 1024) < 1 ? 1 : (int)((n + 1023) / 1024);
  for (int b = 0; b < blocks; ++b) {
    double a = 0.0;
    for (int64_t i = (int64_t)b * 256 + threadIdx.x; i < n; i += (int64_t)blocks * 256) {
      const double v = r[i];
      a += v * v;
    }
    a = wsum(a);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) blk[b] = part[0] + part[1] + part[2] + part[3];
  }
  __syncthreads();
  double a = 0.0;
  for (int i = threadIdx.x; i < blocks; i += 256) a += blk[i];
  a = wsum(a);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
  __syncthreads();
  double ct = part[0] + part[1] + part[2] + part[3];
  __syncthreads();
  if (prec) {
    double b = 0.0;
    for (int64_t j = threadIdx.x; j < P; j += 256) {
      const double t = prec[j] * (p[j] - pmean[j]);     // (prior_vec_kernel, then the tail's (p - pbar) . t)
      tvec[j] = t;
      b += (p[j] - pmean[j]) * t;
    }
    b = wsum(b);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = b;
    __syncthreads();
    ct += part[0] + part[1] + part[2] + part[3];
  }
  if (threadIdx.x != 0) return;
  chi2_out[0] = ct;
  const double chi2 = st[LMS_CHI2], mu = st[LMS_MU];
  const bool solved = chol_info[0] == 0 && st[LMS_VFINITE] != 0.0;
  double rho = -1.0;
  if (solved) {
    const double normf = sqrt(chi2), normf_t = sqrt(ct);
    if (normf_t < normf) {   // NaN-safe: anything else rejects
      const double u = normf_t / normf;
      const double pred = (st[LMS_VG] + mu * st[LMS_DV2]) / chi2;
      rho = pred > 0.0 ? (1.0 - u * u) / pred : -1.0;
    }
  }
  if (rho > 0.75) st[LMS_DELTA] *= factor_up;
  else if (rho < 0.25) st[LMS_DELTA] /= factor_down;
  if (rho > 0.0) {
    const double b = 2.0 * rho - 1.0;
    st[LMS_MU] = mu * fmax(0.333333333333333, 1.0 - b * b * b);
    st[LMS_NU] = 2.0;
  } else {
    st[LMS_MU] = mu * st[LMS_NU];
    st[LMS_NU] *= 2.0;
  }
  st[LMS_RHO] = rho;
  st[LMS_CHI2_TRIAL] = ct;
  st[LMS_ACCEPT] = rho > 0.0 ? 1.0 : 0.0;
  st[LMS_SOLVED] = solved ? 1.0 : 0.0;
  lm_publish(st);
}

hipError_t launch_lm_trial_tail_small(hipStream_t stream, const double *r, int64_t n, int64_t P, const double *prec,
                                      const double *pmean, const double *p, double *tvec, double *chi2_out,
                                      const int32_t *chol_info, double factor_up, double factor_down, double *lmd) {
  hipLaunchKernelGGL(lm_trial_tail_small_kernel, dim3(1), dim3(256), 0, stream, r, n, P, prec, pmean, p, tvec, chi2_out,
                     chol_info, factor_up, factor_down, lmd);
  return hipGetLastError();
}

// The tail of an accepted step in ONE launch: diag(J^T J) out of the packed tiles, the update of the scaling D
// and the convergence test (packed_diag_kernel + scale_update_kernel + lm_converge_kernel, element for element).
__global__ __launch_bounds__(256) void lm_accept_tail_kernel(const double *apk, int64_t P, int64_t T, int scaler,
                                                             double *coln2, double *dscale, const double *x,
                                                             const double *v, double *gvec, double xtol,
                                                             double gtol, double *st, const double *tvec,
                                                             const double *pmean, const double *nrm_part, int nrm_blocks,
                                                             const double *nrm_prior, int nrm_prior_dense, double *apk_w) {
  __shared__ double sh[4];
  __shared__ double nq_s[96];
  if (nrm_part) {
    // few parameters, few rows (jit.hip lsqamd_jit_nrm, <= 64 workgroups): the per-workgroup sums are added up here, in
    // workgroup order, and unpacked into the packed tile / gvec -- colsum_reduce + nrm_unpack folded into this launch
    const int Pn = (int)P, NA = Pn * (Pn + 1) / 2, NQ = NA + Pn + 1;
    if ((int)threadIdx.x < NQ) {
      double a = 0.0;
      for (int b = 0; b < nrm_blocks; ++b) a += nrm_part[b * NQ + threadIdx.x];
      nq_s[threadIdx.x] = a;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < Pn * Pn; e += 256) {
      const int i = e / Pn, j = e % Pn;
      const int a = i < j ? i : j, b = i < j ? j : i;
      double val = nq_s[a * Pn - a * (a - 1) / 2 + (b - a)];
      if (nrm_prior) val += nrm_prior_dense ? nrm_prior[i * Pn + j] : (i == j ? nrm_prior[i] : 0.0);
      apk_w[i * TB + j] = val;
    }
    for (int j = threadIdx.x; j <= Pn; j += 256) gvec[j] = nq_s[NA + j];
    __syncthreads();
  }
  double notx = 0.0, gn = 0.0;
  if (tvec) {   // the prior's share of g and chi2 (prior_apply_kernel, same sums in the same order) rides along
    double a = 0.0;
    for (int64_t j = threadIdx.x; j < P; j += 256) {
      const double t = tvec[j];
      gvec[j] += t;
      a += (x[j] - pmean[j]) * t;
    }
    a = wsum(a);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) gvec[P] += sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
  }
  for (int64_t j = threadIdx.x; j < P; j += 256) {
    const int64_t tm = j / TB, r = j % TB;
    const double c2 = apk[packed_tile_index(tm, tm, T) * TB * TB + r * TB + r];
    coln2[j] = c2;
    const double cn = sqrt(c2 > 0.0 ? c2 : 0.0);
    double d;
    if (scaler == LSQAMD_SCALE_LEVENBERG) d = dscale[j];
    else if (scaler == LSQAMD_SCALE_MORE) d = fmax(dscale[j], cn);
    else d = cn == 0.0 ? 1.0 : cn;
    dscale[j] = d;
    const double xj = x[j];
    notx += (fabs(v[j]) < xtol * xtol + xtol * fabs(xj)) ? 0.0 : 1.0;
    gn = fmax(gn, fabs(fmax(xj, 1.0) * gvec[j]));
  }
  notx = block_sum(notx, sh);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gn = fmax(gn, __shfl_down(gn, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = gn;
  __syncthreads();
  if (threadIdx.x == 0) {
    gn = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
    const double chi2 = gvec[P];
    st[LMS_CHI2] = chi2;
    st[LMS_INFO] = notx == 0.0 ? 1.0 : (gn <= gtol * fmax(0.5 * chi2, 1.0) ? 2.0 : 0.0);
    lm_publish(st);
  }
}

hipError_t launch_lm_accept_tail(hipStream_t stream, const double *apk, int64_t P, int scaler, double *coln2,
                                 double *dscale, const double *x, const double *v, double *gvec, double xtol,
                                 double gtol, double *st, const double *tvec, const double *pmean, const double *nrm_part,
                                 int nrm_blocks, const double *nrm_prior, int nrm_prior_dense) {
  const int64_t T = (P + TB - 1) / TB;
  hipLaunchKernelGGL(lm_accept_tail_kernel, dim3(1), dim3(256), 0, stream, apk, P, T, scaler, coln2, dscale, x, v, gvec,
                     xtol, gtol, st, tvec, pmean, nrm_part, nrm_blocks, nrm_prior, nrm_prior_dense, const_cast<double *>(apk));
  return hipGetLastError();
}

hipError_t launch_lm_decide(hipStream_t stream, const double *chi2_trial, const int32_t *chol_info, double factor_up,
                            double factor_down, double *st) {
  hipLaunchKernelGGL(lm_decide_kernel, dim3(1), dim3(1), 0, stream, chi2_trial, chol_info, factor_up, factor_down, st);
  return hipGetLastError();
}

// after the accepted point's normal equations: chi2 into the record and gsl_multifit_nlinear_test
// (convergence.c): info 1 when every |dx_i| < xtol^2 + xtol |x_i|, else 2 when
// max_i |g_i max(x_i, 1)| <= gtol max(chi2 / 2, 1), else 0.  dx = -v.
__global__ __launch_bounds__(256) void lm_converge_kernel(int64_t P, const double *x, const double *v, const double *gvec,
                                                          double xtol, double gtol, double *st) {
  __shared__ double sh[4];
  double notx = 0.0, gn = 0.0;
  for (int64_t j = threadIdx.x; j < P; j += 256) {
    const double xj = x[j];
    notx += (fabs(v[j]) < xtol * xtol + xtol * fabs(xj)) ? 0.0 : 1.0;
    gn = fmax(gn, fabs(fmax(xj, 1.0) * gvec[j]));
  }
  notx = block_sum(notx, sh);
  // max over the block
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gn = fmax(gn, __shfl_down(gn, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = gn;
  __syncthreads();
  if (threadIdx.x == 0) {
    gn = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
    const double chi2 = gvec[P];
    st[LMS_CHI2] = chi2;
    st[LMS_INFO] = notx == 0.0 ? 1.0 : (gn <= gtol * fmax(0.5 * chi2, 1.0) ? 2.0 : 0.0);
    lm_publish(st);
  }
}

hipError_t launch_lm_converge(hipStream_t stream, int64_t P, const double *x, const double *v, const double *gvec,
                              double xtol, double gtol, double *st) {
  hipLaunchKernelGGL(lm_converge_kernel, dim3(1), dim3(256), 0, stream, P, x, v, gvec, xtol, gtol, st);
  return hipGetLastError();
}

}  // namespace lsqamd
