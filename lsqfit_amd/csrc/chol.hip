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
//   diagonal block   factor + triangular inverse by one workgroup: potf2_mfma.hip (tiles
//                    resident in the MFMA accumulator layout; default) or the LDS-resident
//                    kernel below (LSQAMD_POTF2=lds)
//   row panel        U[k, k+nb:] = inv(U_kk)^T A[k, k+nb:]          (in-place TN GEMM, 128 x 64 tiles)
//   trailing update  A[k+nb:, k+nb:] -= U[k, k+nb:]^T U[k, k+nb:]    (TN GEMM, upper tiles; while it
//                    is large, one launch with the next diagonal block: trail_potf2_kernel)
// Extra columns to the right of the n x n block ride along in the row panels, so
// appending the gradient as column n yields the forward substitution U^-T g for free.
#include <cstdlib>

#include "common.h"

namespace lsqamd {

constexpr int NB = CHOL_NB;
constexpr int PLD = NB + 2;  // LDS leading dimension: row walks AND 16-lane column walks
                             // (MFMA A fragments) are both bank-conflict free at 260 dwords
constexpr int SB = 32;       // register-resident sub-block edge

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

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
//   1a  wave 0: 32 x 32 Cholesky, one column per lane, pivots broadcast with v_readlane,
//       1/sqrt by v_rsq_f64 + two Newton steps, branch-free (one basic block: the
//       scheduler overlaps pivot j+1's rsqrt chain with pivot j's rank-1 update)
//   1b  forward substitution of the 32-row panel, one column per thread
//   1c  rank-32 update of the trailing block on the matrix cores (fp64 MFMA from LDS)
//   3a  the four 32 x 32 triangular inverses, one per wave, one column per lane
//   3b/3c  off-diagonal blocks of the inverse by the 2 x 2 block formula
//          inv([[a, b], [0, c]]) = [[a^-1, -a^-1 b c^-1], [0, c^-1]]  (MFMA products,
//          the dead strictly-lower triangle is the scratch for the intermediate one).
// Blocks smaller than 128 are padded with the identity.

constexpr int WLD = SB + 2;  // leading dimension of the 32 x 32 scratch copy of inv(L11)

__device__ __forceinline__ double rsqrt_refined(double d) {
  double y = __builtin_amdgcn_rsq(d);  // v_rsq_f64, then two Newton steps to full precision
  const double h = -0.5 * d;
  double e = __builtin_fma(h * y, y, 0.5);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(h * y, y, 0.5);
  return __builtin_fma(y, e, y);
}

// 32 x 32 Cholesky D = U^T U by ONE wave, one column per lane, entirely in registers.
//  * lanes 0..31 hold the columns of D, lanes 32..63 the columns of the identity: the same
//    row operations turn the identity into W = U^-T for free (the otherwise idle half of
//    the wave does it), so the panel solve below is a matrix-core product and no separate
//    triangular inversion of the diagonal sub-blocks is needed;
//  * row j of U is broadcast with v_readlane in chunks of 8 fenced by sched_barrier: left
//    alone, hipcc hoists all 31 broadcasts of a pivot to the top and spills hundreds of
//    SGPRs through v_writelane;
//  * the next pivot's 1/sqrt chain is started from lane j+1's own updated diagonal before
//    the rank-1 update of pivot j, so it overlaps with that update;
//  * branch-free (not-positive-definite is recorded, the pivot replaced by 1).
// Results: U (upper) and the strictly-lower part of W in place, 1/U_jj in dinv, all of W in
// wbuf (row-major, WLD).
__device__ __noinline__ void chol32_wave(double *s, int j0, double *dinv, double *wbuf,
                                         int32_t *info, int k0, int lane) {
  const int c = lane & 31;
  const bool isW = lane >= SB;
  double col[SB];
#pragma unroll
  for (int i = 0; i < SB; ++i) col[i] = s[(j0 + i) * PLD + j0 + c];  // all 64 lanes load
#pragma unroll
  for (int i = 0; i < SB; ++i) col[i] = isW ? ((i == c) ? 1.0 : 0.0) : col[i];
  int first_bad = -1;
  double d = readlane_d(col[0], 0);
  bool ok = (d > 0.0) && (d < 1.0e300);
  first_bad = ok ? first_bad : 0;
  double y = rsqrt_refined(ok ? d : 1.0);
  double ykeep = 0.0;
#pragma unroll
  for (int j = 0; j < SB; ++j) {
    double u = col[j] * y;
    u = (!isW && c < j) ? 0.0 : u;
    col[j] = u;
    ykeep = (c == j) ? y : ykeep;
    double ynext = 0.0;
    if (j + 1 < SB) {
      const double dn = readlane_d(col[j + 1] - u * u, j + 1);
      ok = (dn > 0.0) && (dn < 1.0e300);
      first_bad = (!ok && first_bad < 0) ? j + 1 : first_bad;
      ynext = rsqrt_refined(ok ? dn : 1.0);
    }
    // rank-1 update in groups of 8 rows: the 8 broadcasts first (16 distinct SGPRs), then the
    // 8 FMAs, so no FMA waits on the readlane right in front of it
#pragma unroll
    for (int i0 = j + 1; i0 < SB; i0 += 8) {
      double ub[8];
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (i0 + t < SB) ub[t] = readlane_d(u, i0 + t);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (i0 + t < SB) col[i0 + t] -= ub[t] * u;
      __builtin_amdgcn_sched_barrier(0);
    }
    y = ynext;
  }
  if (first_bad >= 0 && lane == 0) atomicCAS(info, 0, k0 + j0 + first_bad + 1);
  if (!isW) dinv[j0 + c] = ykeep;
#pragma unroll
  for (int i = 0; i < SB; ++i) {
    if (isW) {
      wbuf[i * WLD + c] = (i >= c) ? col[i] : 0.0;
      if (i > c) s[(j0 + i) * PLD + j0 + c] = col[i];
    } else if (i <= c) {
      s[(j0 + i) * PLD + j0 + c] = col[i];
    }
  }
}

// Row panel U12 = W * A12 (W = inv(L11), 32 x 32 lower triangular, in wbuf) on the matrix
// cores, in place: each wave owns whole 16-column strips, so both 16-row halves of a strip
// are accumulated before either is overwritten.
__device__ __noinline__ void panel32_mma(double *s, int j0, const double *wbuf, int rest, int wave,
                                            int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  for (int tj = wave; tj < rest / 16; tj += 4) {
    double *A12 = s + j0 * PLD + j0 + SB + tj * 16;
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < SB; k += 4) {
      const double b = A12[(k + fq) * PLD + fr];
      if (k < 16) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wbuf[fr * WLD + k + fq], b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wbuf[(16 + fr) * WLD + k + fq], b, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      A12[(fq + 4 * r) * PLD + fr] = acc0[r];
      A12[(16 + fq + 4 * r) * PLD + fr] = acc1[r];
    }
  }
}

// Small dense products on the matrix cores, operands and result in LDS (row-major, PLD):
//   C[i][j] = (ACC ? C[i][j] : 0) + sign * sum_k A(i,k) * R[k][j]
// with A(i,k) = L[k][i] when TA (k-major left operand) else L[i][k].  16 x 16 output tiles
// are dealt round-robin to the 4 waves; `upper` skips tiles strictly below the diagonal;
// `r_upper` / `l_upper` restrict k to the non-zero part of an upper-triangular R / L.
template <bool TA, bool ACC, int KS>
__device__ __forceinline__ void lds_mma(const double *L, const double *R, double *Cm, int mt, int nt,
                                        double sign, bool upper, bool r_upper, bool l_upper, int wave,
                                        int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  int t = 0;
  for (int ti = 0; ti < mt; ++ti)
    for (int tj = (upper ? ti : 0); tj < nt; ++tj, ++t) {
      if ((t & 3) != wave) continue;
      const int i0 = ti * 16, j0 = tj * 16;
      int klo = 0, khi = KS * 4;
      if (r_upper && khi > j0 + 16) khi = j0 + 16;
      if (l_upper && klo < i0) klo = i0;
      // rolled k loop with the next fragments in flight behind the current MFMA (a fully
      // unrolled, guarded chain made hipcc emit exec-masked MFMA variants that came out wrong)
      v4d acc = {0.0, 0.0, 0.0, 0.0};
      if (klo < khi) {
        double a0 = TA ? L[(klo + fq) * PLD + i0 + fr] : L[(i0 + fr) * PLD + klo + fq];
        double b0 = R[(klo + fq) * PLD + j0 + fr];
        for (int k = klo; k < khi; k += 4) {
          const int kn = (k + 4 < khi) ? k + 4 : k;
          const double a1 = TA ? L[(kn + fq) * PLD + i0 + fr] : L[(i0 + fr) * PLD + kn + fq];
          const double b1 = R[(kn + fq) * PLD + j0 + fr];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
          a0 = a1;
          b0 = b1;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double *cp = Cm + (i0 + fq + 4 * r) * PLD + j0 + fr;
        *cp = ACC ? *cp + sign * acc[r] : sign * acc[r];
      }
    }
}

#ifdef LSQAMD_POTF2_TIMING
#define STAMP(i) do { if (tid == 0 && dbg) dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

template <bool VEC>
__global__ __launch_bounds__(256) void potf2_inv_kernel(double *A, int64_t lda, int nb, double *uinv,
                                                        int32_t *info, int32_t k0, long long *dbg,
                                                        int64_t strideA, int64_t strideW,
                                                        const int32_t *active) {
  extern __shared__ __attribute__((aligned(16))) double s[];
  // batched use: workgroup b factors matrix b
  if (active && !active[blockIdx.x]) return;
  A += (int64_t)blockIdx.x * strideA;
  uinv += (int64_t)blockIdx.x * strideW;
  info += blockIdx.x;
  double *dinv = s + NB * PLD;
  double *wbuf = dinv + NB;          // 32 x WLD copy of inv(L11) of the current sub-block
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nsb = (nb + SB - 1) / SB;  // active 32-wide sub-blocks
  // ---- load: thread -> 2 columns, 32 rows; clamped (always valid) addresses, masked after
  {
    const int c2 = (tid & 63) * 2, rg = tid >> 6;
    const int cc0 = c2 < nb ? c2 : nb - 1, cc1 = c2 + 1 < nb ? c2 + 1 : nb - 1;
    const int cv = c2 + 1 < nb ? c2 : (nb >= 2 ? (nb - 2) & ~1 : 0);
#pragma unroll 8
    for (int it = 0; it < 32; ++it) {
      const int r = it * 4 + rg;
      const int rc = r < nb ? r : nb - 1;
      double x0, x1;
      if (VEC && c2 + 1 < nb) {
        const v2d v = *reinterpret_cast<const v2d *>(A + (int64_t)rc * lda + cv);
        x0 = v.x; x1 = v.y;
      } else {
        x0 = A[(int64_t)rc * lda + cc0];
        x1 = A[(int64_t)rc * lda + cc1];
      }
      const bool in0 = r < nb && c2 < nb, in1 = r < nb && c2 + 1 < nb;
      x0 = in0 ? (c2 >= r ? x0 : 0.0) : (r == c2 ? 1.0 : 0.0);
      x1 = in1 ? (c2 + 1 >= r ? x1 : 0.0) : (r == c2 + 1 ? 1.0 : 0.0);
      *reinterpret_cast<v2d *>(s + r * PLD + c2) = (v2d){x0, x1};
    }
  }
  STAMP(0);
  __syncthreads();
  STAMP(1);
  // ---- phase 1: Cholesky
  for (int jb = 0; jb < nsb; ++jb) {
    const int j0 = jb * SB;
    if (wave == 0) chol32_wave(s, j0, dinv, wbuf, info, k0, lane);
    __syncthreads();
    STAMP(2 + 3 * jb);
    const int rest = nsb * SB - (j0 + SB);
    if (rest > 0) {
      panel32_mma(s, j0, wbuf, rest, wave, lane);
      __syncthreads();
      STAMP(3 + 3 * jb);
      // A22 -= U12^T U12 (upper tiles), U12 = rows j0..j0+31, columns from j0+32
      const double *u12 = s + j0 * PLD + j0 + SB;
      lds_mma<true, true, 8>(u12, u12, s + (j0 + SB) * PLD + j0 + SB, rest / 16, rest / 16, -1.0, true,
                             false, false, wave, lane);
      __syncthreads();
      STAMP(4 + 3 * jb);
    }
  }
  STAMP(14);
  // U back to global (upper triangle of the block)
  {
    const int c2 = (tid & 63) * 2, rg = tid >> 6;
    for (int it = 0; it < 32; ++it) {
      const int r = it * 4 + rg;
      if (r >= nb || c2 + 1 < r || c2 >= nb) continue;
      const v2d v = *reinterpret_cast<const v2d *>(s + r * PLD + c2);
      if (VEC && c2 >= r && c2 + 1 < nb) {
        *reinterpret_cast<v2d *>(A + (int64_t)r * lda + c2) = v;
      } else {
        if (c2 >= r) A[(int64_t)r * lda + c2] = v.x;
        if (c2 + 1 < nb) A[(int64_t)r * lda + c2 + 1] = v.y;
      }
    }
  }
  __syncthreads();
  STAMP(15);
  // ---- phase 3: inverse of U in place.  Diagonal sub-blocks: inv(U11) = W^T, W's strictly
  // lower part sits below U11's diagonal, its diagonal in dinv.
  {
    double t[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int idx = tid + 256 * e;              // 4 sub-blocks x 32 x 32
      const int sb = idx >> 10, i = (idx >> 5) & 31, k = idx & 31, q = sb * SB;
      t[e] = (k > i) ? s[(q + k) * PLD + q + i] : ((k == i) ? dinv[q + i] : 0.0);
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int idx = tid + 256 * e;
      const int sb = idx >> 10, i = (idx >> 5) & 31, k = idx & 31, q = sb * SB;
      if (sb < nsb) s[(q + i) * PLD + q + k] = t[e];
    }
  }
  __syncthreads();
  STAMP(16);
  if (nsb > 1) {
    // level 1, both 64-blocks q = 0, 64: T = b c^-1 into the dead lower sub-block,
    // then b' = -a^-1 T.  (The second 64-block has an off-diagonal only with 4 sub-blocks.)
    const int nq = nsb > 3 ? 2 : 1;
    for (int qi = 0; qi < nq; ++qi) {
      const int q = qi * 64;
      lds_mma<false, false, 8>(s + q * PLD + q + 32, s + (q + 32) * PLD + q + 32, s + (q + 32) * PLD + q,
                               2, 2, 1.0, false, true, false, (wave + 2 * qi) & 3, lane);
    }
    __syncthreads();
    for (int qi = 0; qi < nq; ++qi) {
      const int q = qi * 64;
      lds_mma<false, false, 8>(s + q * PLD + q, s + (q + 32) * PLD + q, s + q * PLD + q + 32, 2, 2, -1.0,
                               false, false, true, (wave + 2 * qi) & 3, lane);
    }
    __syncthreads();
    // the level-1 scratch sits inside the triangles level 2 multiplies with: clear it
    for (int e = tid; e < nq * 32 * 32; e += 256) {
      const int q = (e >> 10) * 64, ee = e & 1023;
      s[(q + 32 + (ee >> 5)) * PLD + q + (ee & 31)] = 0.0;
    }
    __syncthreads();
    if (nsb > 2) {
      // level 2: B = rows 0..63, cols 64..127; scratch = rows 64..127, cols 0..63
      lds_mma<false, false, 16>(s + 64, s + 64 * PLD + 64, s + 64 * PLD, 4, 4, 1.0, false, true, false,
                                wave, lane);
      __syncthreads();
      lds_mma<false, false, 16>(s, s + 64 * PLD, s + 64, 4, 4, -1.0, false, false, true, wave, lane);
      __syncthreads();
    }
  }
  STAMP(17);
  {
    const int c2 = (tid & 63) * 2, rg = tid >> 6;
    for (int it = 0; it < 32; ++it) {
      const int r = it * 4 + rg;
      if (r >= nb) continue;
      v2d v = *reinterpret_cast<const v2d *>(s + r * PLD + c2);
      v.x = (c2 >= r && c2 < nb) ? v.x : 0.0;
      v.y = (c2 + 1 >= r && c2 + 1 < nb) ? v.y : 0.0;
      *reinterpret_cast<v2d *>(uinv + r * NB + c2) = v;
    }
  }
  STAMP(18);
}

static PerDeviceOnce g_potf2_attr;   // per device, thread-safe (common.h)
long long *g_potf2_dbg = nullptr;  // device buffer of 32 cycle stamps (LSQAMD_POTF2_TIMING builds)
static constexpr size_t POTF2_LDS = (size_t)(NB * PLD + NB + SB * WLD) * sizeof(double);  // 140 KiB

static hipError_t launch_potf2(hipStream_t st, double *A, int64_t lda, int nb, double *uinv,
                               int32_t *info, int32_t k0, int32_t batch = 1, int64_t strideA = 0,
                               int64_t strideW = 0, const int32_t *active = nullptr) {
  {
    const hipError_t ea = g_potf2_attr.run([] {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(potf2_inv_kernel<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)POTF2_LDS);
      if (e != hipSuccess) return e;
      return hipFuncSetAttribute(reinterpret_cast<const void *>(potf2_inv_kernel<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)POTF2_LDS);
    });
    if (ea != hipSuccess) return ea;
  }
  static const bool use_mfma = [] {
    const char *e = getenv("LSQAMD_POTF2");
    return !(e && e[0] == 'l');   // LSQAMD_POTF2=lds selects the LDS-resident kernel (developer knob)
  }();
  if (use_mfma) {
    return launch_potf2_mfma(st, A, lda, nb, uinv, info, k0, batch, strideA, strideW, active);
  }
  const bool vec = !(lda & 1) && !(reinterpret_cast<uintptr_t>(A) & 15) && !(strideA & 1);
  if (vec)
    hipLaunchKernelGGL(potf2_inv_kernel<true>, dim3((unsigned)batch), dim3(256), POTF2_LDS, st, A, lda, nb,
                       uinv, info, k0, g_potf2_dbg, strideA, strideW, active);
  else
    hipLaunchKernelGGL(potf2_inv_kernel<false>, dim3((unsigned)batch), dim3(256), POTF2_LDS, st, A, lda, nb,
                       uinv, info, k0, g_potf2_dbg, strideA, strideW, active);
  return hipGetLastError();
}

hipError_t potrf_upper_batched(hipStream_t st, double *A, int64_t n, int64_t lda, int64_t n_cols,
                               double *work, int32_t *dev_info, int32_t batch, int64_t strideA,
                               int64_t strideW, const int32_t *active, bool info_zeroed) {
  hipError_t e = info_zeroed ? hipSuccess : hipMemsetAsync(dev_info, 0, sizeof(int32_t) * (size_t)batch, st);
  if (e != hipSuccess) return e;
  // single matrix, everything tile-aligned: the trailing update of step k carries the diagonal
  // block of step k + 1 along (launch_trail_potf2) while there are enough tiles to hide it behind
  // (sweep at P = 4096 with the pivot-wave diagonal kernel: >= 1 tile 1.553 ms, >= 6 1.541, >= 15 1.543, >= 28 1.546,
  // >= 45 1.550, >= 66 1.565; round 1: never 2.73 ms, >= 3 tiles 2.60, >= 30 2.55, >= 136 2.53)
  static const int64_t fuse_min_tiles = [] { const char *v = getenv("LSQAMD_FUSE_MIN_TILES"); return v ? atoll(v) : (int64_t)10; }();
  const bool can_fuse = batch == 1 && !active && fuse_min_tiles >= 0 && trail_potf2_available() && (n % NB) == 0 &&
                        (n_cols % NB) == 0 &&
                        !(lda & 1) && !(reinterpret_cast<uintptr_t>(A) & 15);
  bool diag_done = false;  // the diagonal block of this step was factored by the previous launch
  for (int64_t k0 = 0; k0 < n; k0 += NB) {
    const int nb = (int)((n - k0) < NB ? (n - k0) : NB);
    double *uinv = work + (k0 / NB) * NB * NB;
    if (!diag_done) {
      e = launch_potf2(st, A + k0 * lda + k0, lda, nb, uinv, dev_info, (int32_t)k0, batch, strideA, strideW,
                       active);
      if (e != hipSuccess) return e;
    }
    diag_done = false;
    const int64_t rest = n_cols - (k0 + nb);
    if (rest <= 0) continue;
    GemmTN p;  // row panel: U[k, k+nb:] = inv(U_kk)^T * A[k, k+nb:]   (in place)
    p.X = uinv; p.ldx = NB; p.sx = strideW;
    p.Y = A + k0 * lda + k0 + nb; p.ldy = lda; p.sy = strideA;
    p.C = A + k0 * lda + k0 + nb; p.ldc = lda; p.sc = strideA;
    p.M = nb; p.N = rest; p.K = nb;
    p.x_upper_tri = 1;
    p.batch = batch; p.batch_active = active;
    e = launch_gemm_tn(st, p);
    if (e != hipSuccess) return e;
    const int64_t mrest = n - (k0 + nb);
    if (mrest <= 0) continue;
    const int64_t tiles = (mrest / NB) * (mrest / NB + 1) / 2;
    if (can_fuse && tiles >= fuse_min_tiles) {
      e = launch_trail_potf2(st, p.C, A + (k0 + nb) * lda + (k0 + nb), lda, mrest, rest, NB,
                             work + ((k0 + nb) / NB) * NB * NB, dev_info, (int32_t)(k0 + nb));
      if (e != hipSuccess) return e;
      diag_done = true;
      continue;
    }
    GemmTN t;  // trailing: A[k+nb:, k+nb:] -= panel^T panel  (upper tiles only)
    t.X = p.C; t.ldx = lda; t.sx = strideA;
    t.Y = p.C; t.ldy = lda; t.sy = strideA;
    t.C = A + (k0 + nb) * lda + (k0 + nb); t.ldc = lda; t.sc = strideA;
    t.M = mrest; t.N = rest; t.K = nb;
    t.alpha = -1.0; t.beta = 1.0;
    t.upper_only = 1;
    t.batch = batch; t.batch_active = active;
    e = launch_gemm_tn(st, t);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t potrf_upper(hipStream_t st, double *A, int64_t n, int64_t lda, int64_t n_cols,
                       double *work, int32_t *dev_info, bool info_zeroed) {
  return potrf_upper_batched(st, A, n, lda, n_cols, work, dev_info, 1, 0, 0, nullptr, info_zeroed);
}

// ---- back substitution U v = y ------------------------------------------------------
constexpr int BS_ROWS = 16;  // rows of y updated per workgroup (32: 0.370 ms, 16: 0.353, 64: 0.504 at P = 4096)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Every workgroup recomputes v_k = inv(U_kk) y_k (128 x 128 GEMV, L2-resident) into
// LDS; workgroup 0 publishes it; workgroups 1.. subtract U[r, k-block] . v_k from
// their rows r < k0 of y.  All global loads of a phase are issued before the first
// reduction, so a step costs about two memory round trips instead of ~40.
__global__ __launch_bounds__(256) void backsolve_step_kernel(const double *A, int64_t lda, int64_t k0,
                                                             int nb, const double *uinv, double *y,
                                                             double *v, int64_t strideA, int64_t strideW,
                                                             int64_t strideY, const int32_t *active) {
  if (active && !active[blockIdx.y]) return;
  A += (int64_t)blockIdx.y * strideA;
  uinv += (int64_t)blockIdx.y * strideW;
  y += (int64_t)blockIdx.y * strideY;
  v += (int64_t)blockIdx.y * strideY;
  __shared__ double vk[NB];
  __shared__ double yk[NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < NB) yk[tid] = tid < nb ? y[k0 + tid] : 0.0;
  // rows wave*32 .. wave*32+31 of inv(U_kk): two coalesced 512-byte loads per row
  double a0[32], a1[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int r = wave * 32 + i;
    const bool ok = r < nb;
    a0[i] = (ok && lane < nb) ? uinv[r * NB + lane] : 0.0;
    a1[i] = (ok && lane + 64 < nb) ? uinv[r * NB + lane + 64] : 0.0;
  }
  __syncthreads();
  const double y0 = yk[lane], y1 = yk[lane + 64];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const double acc = wave_sum(a0[i] * y0 + a1[i] * y1);
    if (lane == 0) vk[wave * 32 + i] = acc;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    if (tid < nb) v[k0 + tid] = vk[tid];
    return;
  }
  const int64_t r0 = (int64_t)(blockIdx.x - 1) * BS_ROWS + wave * (BS_ROWS / 4);
  double u0[BS_ROWS / 4], u1[BS_ROWS / 4];
#pragma unroll
  for (int i = 0; i < BS_ROWS / 4; ++i) {
    const int64_t r = r0 + i;
    const bool ok = r < k0;
    const double *row = A + (ok ? r : 0) * lda + k0;
    u0[i] = (ok && lane < nb) ? row[lane] : 0.0;
    u1[i] = (ok && lane + 64 < nb) ? row[lane + 64] : 0.0;
  }
  const double v0 = vk[lane], v1 = vk[lane + 64];
#pragma unroll
  for (int i = 0; i < BS_ROWS / 4; ++i) {
    const double acc = wave_sum(u0[i] * v0 + u1[i] * v1);
    if (lane == 0 && r0 + i < k0) y[r0 + i] -= acc;
  }
}

// ---- back substitution, four diagonal blocks per launch ---------------------------------------
// The step kernel above is launch-bound (32 launches of ~14 us at n = 4096).  Here every workgroup
// recomputes the solution of a GROUP of Q consecutive diagonal blocks on its own (Q inverse GEMVs
// plus Q (Q - 1) / 2 tile GEMVs, all L2-resident operands) and then updates its rows above the
// group with all Q pieces: a quarter of the launches, each a little longer.
// 32-row GEMV pieces: lane -> two columns, the 32 per-row partial products are reduced across the
// wave by a halving butterfly (32 shuffles instead of 32 x 6); row i's total lands in lanes 2i, 2i+1.
__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }

template <int CNT>
__device__ __forceinline__ void halve(double (&p)[32], int lane, int mask) {
  const bool hi = (lane & mask) != 0;
#pragma unroll
  for (int i = 0; i < CNT / 2; ++i) {
    const double send = hi ? p[i] : p[i + CNT / 2];
    const double keep = hi ? p[i + CNT / 2] : p[i];
    p[i] = keep + shfl_xor_d(send, mask);
  }
}

// ys[32 w + i] (-)= sum_c M[(32 w + i)][c] xs[c]  for this wave's 32 rows; M row-major (ld), 128 columns
__device__ __forceinline__ void gemv32(const double *M, int64_t ld, const double *xs, double *ys, bool subtract,
                                       int wave, int lane) {
  const double x0 = xs[lane], x1 = xs[lane + 64];
  double p[32];
  const double *row = M + (int64_t)(wave * 32) * ld;
#pragma unroll
  for (int i = 0; i < 32; ++i) p[i] = row[i * ld + lane] * x0 + row[i * ld + lane + 64] * x1;
  halve<32>(p, lane, 32);
  halve<16>(p, lane, 16);
  halve<8>(p, lane, 8);
  halve<4>(p, lane, 4);
  halve<2>(p, lane, 2);
  const double tot = p[0] + shfl_xor_d(p[0], 1);
  if (!(lane & 1)) {
    const int r = wave * 32 + (lane >> 1);
    ys[r] = subtract ? ys[r] - tot : tot;
  }
}


// ---- small systems (n = 128 or 256): the rest of a trial solve in ONE single-workgroup launch -------------------
// After potrf_upper the factor U sits in M, column n of M holds y = U^-T g (the forward substitution rode along with
// the row panels) and uinv the inverses of the diagonal blocks.  This kernel is the back substitution v = U^-1 y (at
// most three 128 x 128 GEMVs, every wave's 32 rows requested at once: gemv32), the trial point x - v and the dot
// products / finiteness / pivot watch of lm_trial_kernel (vecops.hip) -- instead of copy_column_zero +
// backsolve_chain + lm_trial, three dependent launches of 4-7 us each.
__global__ __launch_bounds__(256) void lm_solve_tail_small_kernel(const double *M, int64_t ld, int n, const double *uinv,
                                                                  const double *x, const double *g, const double *d,
                                                                  double *xt, double *v_out, double *st,
                                                                  const double *a_diag, const int32_t *chol_info) {
  __shared__ double ys[2 * NB], vs[2 * NB];
  __shared__ double sh[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nblk = n / NB;
  if (tid < n) ys[tid] = M[(int64_t)tid * ld + n];
  __syncthreads();
  for (int k = nblk - 1; k >= 0; --k) {
    gemv32(uinv + (int64_t)k * NB * NB, NB, ys + k * NB, vs + k * NB, false, wave, lane);      // v_k = inv(U_kk) y_k
    __syncthreads();
    for (int b = 0; b < k; ++b)                                                                 // y_b -= U[b, k] v_k
      gemv32(M + (int64_t)b * NB * ld + (int64_t)k * NB, ld, vs + k * NB, ys + b * NB, true, wave, lane);
    __syncthreads();
  }
  double vg = 0.0, dv2 = 0.0, bad = 0.0, pmin = INFINITY;
  const double mu = st[LMS_MU];
  if (tid < n) {
    const double vj = vs[tid];
    v_out[tid] = vj;
    xt[tid] = x[tid] - vj;
    vg = vj * g[tid];
    const double t = d[tid] * vj;
    dv2 = t * t;
    bad = (vj - vj == 0.0) ? 0.0 : 1.0;
    if (a_diag) {
      const double u = M[(int64_t)tid * ld + tid], m = a_diag[tid] + mu * d[tid] * d[tid];
      pmin = m > 0.0 ? u * u / m : 1.0;
      pmin = (pmin == pmin) ? pmin : INFINITY;   // (a NaN pivot: the factorisation's own status reports it)
    }
  }
  auto bsum = [&](double a) {
    a = wave_sum(a);
    __syncthreads();
    if (lane == 0) sh[wave] = a;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
  };
  vg = bsum(vg);
  dv2 = bsum(dv2);
  bad = bsum(bad);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pmin = fmin(pmin, __shfl_xor(pmin, o, 64));
  __syncthreads();
  if (lane == 0) sh[wave] = pmin;
  __syncthreads();
  if (tid == 0) {
    st[LMS_VG] = vg;
    st[LMS_DV2] = dv2;
    st[LMS_VFINITE] = bad == 0.0 ? 1.0 : 0.0;
    st[LMS_PIVMIN] = a_diag ? fmin(fmin(sh[0], sh[1]), fmin(sh[2], sh[3])) : 1.0;
  }
  (void)chol_info;
}

hipError_t launch_lm_solve_tail_small(hipStream_t stream, const double *M, int64_t ld, int64_t n, const double *uinv,
                                      const double *x, const double *g, const double *d, double *xt, double *v_out,
                                      double *st, const double *a_diag, const int32_t *chol_info) {
  if (n != NB && n != 2 * NB) return hipErrorInvalidValue;
  hipLaunchKernelGGL(lm_solve_tail_small_kernel, dim3(1), dim3(256), 0, stream, M, ld, (int)n, uinv, x, g, d, xt, v_out, st,
                     a_diag, chol_info);
  return hipGetLastError();
}

constexpr int BSQ = 4;  // measured at n = 4096: 2 -> 0.52 ms, 4 -> 0.37 ms, 8 -> 0.56 ms (step kernel: 0.48 ms)

__global__ __launch_bounds__(256) void backsolve_group_kernel(const double *A, int64_t lda, int kb,
                                                              const double *uinv, double *y, double *v,
                                                              int64_t strideA, int64_t strideW,
                                                              int64_t strideY, const int32_t *active) {
  if (active && !active[blockIdx.y]) return;
  A += (int64_t)blockIdx.y * strideA;
  uinv += (int64_t)blockIdx.y * strideW;
  y += (int64_t)blockIdx.y * strideY;
  v += (int64_t)blockIdx.y * strideY;
  __shared__ double yg[BSQ][NB];
  __shared__ double vg[BSQ][NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nq = kb + 1 < BSQ ? kb + 1 : BSQ;   // blocks kb, kb-1, .., kb-nq+1
  for (int e = tid; e < nq * NB; e += 256) yg[e / NB][e % NB] = y[(int64_t)(kb - e / NB) * NB + e % NB];
  __syncthreads();
  for (int q = 0; q < nq; ++q) {
    const int b = kb - q;
    for (int p = 0; p < q; ++p)   // y_b -= U[b, kb - p] v_(kb-p): each wave its own 32 rows of yg[q]
      gemv32(A + (int64_t)b * NB * lda + (int64_t)(kb - p) * NB, lda, vg[p], yg[q], true, wave, lane);
    __syncthreads();
    gemv32(uinv + (int64_t)b * NB * NB, NB, yg[q], vg[q], false, wave, lane);
    __syncthreads();
  }
  if (blockIdx.x == 0) {
    for (int e = tid; e < nq * NB; e += 256) v[(int64_t)(kb - e / NB) * NB + e % NB] = vg[e / NB][e % NB];
    return;
  }
  // rows above the group: BS_ROWS per workgroup, BS_ROWS / 4 per wave, lane -> two columns of each of the nq blocks
  const int64_t top = (int64_t)(kb - nq + 1) * NB;
  const int64_t r0 = (int64_t)(blockIdx.x - 1) * BS_ROWS + wave * (BS_ROWS / 4);
  double acc[BS_ROWS / 4];
#pragma unroll
  for (int i = 0; i < BS_ROWS / 4; ++i) acc[i] = 0.0;
  for (int q = 0; q < nq; ++q) {
    const double v0 = vg[q][lane], v1 = vg[q][lane + 64];
    const int64_t c0 = (int64_t)(kb - q) * NB;
#pragma unroll
    for (int i = 0; i < BS_ROWS / 4; ++i) {
      const int64_t r = r0 + i;
      const double *row = A + (r < top ? r : 0) * lda + c0;
      acc[i] += row[lane] * v0 + row[lane + 64] * v1;
    }
  }
#pragma unroll
  for (int i = 0; i < BS_ROWS / 4; ++i) {
    const double t = wave_sum(acc[i]);
    if (lane == 0 && r0 + i < top) y[r0 + i] -= t;
  }
}

// ---- back substitution as ONE launch: a chain of workgroups --------------------------------------
// Workgroup i owns block row i (128 rows of U, n / 128 workgroups, all resident).  It folds
// U[i, k] v_k into its 128 partial sums as the v_k appear (k = T-1 .. i+1, each tile prefetched into
// registers while it waits), then v_i = inv(U_ii) (y_i - sums) and publishes v_i for the rows above.
// What is serial is T hand-offs of 1 KiB plus two 128 x 128 GEMVs out of registers each -- no kernel
// boundary, no re-computation of the solved blocks in every workgroup (backsolve_group_kernel).
// Hand-off (cdna_hip_programming.md guideline 16, form R2): the data is the flag.  Every double
// travels as two 8-byte granules {tag, 32 bits}, each written by ONE agent-scope (write-through)
// store; one wave of every consumer re-reads its granules until all tags carry this call's epoch.
// The granule words are zeroed by a memset before every launch.  Spins are bounded: a workgroup that
// gives up raises the abort word, everybody leaves, *info reports LSQAMD_CHAIN_TIMEOUT.
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
constexpr unsigned CHAIN_EPOCH = 0x5a17u;
constexpr int32_t CHAIN_TIMEOUT_INFO = -77;

__device__ __forceinline__ void granule_store(gu64 *g, double x) {
  union { double d; unsigned u[2]; } c;
  c.d = x;
  __hip_atomic_store(g, ((unsigned long long)CHAIN_EPOCH << 32) | c.u[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(g + 1, ((unsigned long long)CHAIN_EPOCH << 32) | c.u[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// GEMV piece layout of the chain: wave w owns rows 32 w .. 32 w + 31 of a 128 x 128 tile; its four
// 16-lane groups own 8 rows each and lane c of a group the 8 columns 8 c .. 8 c + 7 -- the 8 per-row
// partial sums of a lane are reduced over 16 lanes only (halving butterfly: 4 + 2 + 1 + 1 exchanges
// instead of 32 over the whole wave); row 8 g + (c >> 1)'s total lands in lanes c and c ^ 1.
__device__ __forceinline__ double reduce8(double (&p)[8], int lane) {
#pragma unroll
  for (int cnt = 8, mask = 8; cnt >= 2; cnt >>= 1, mask >>= 1) {
    const bool hi = (lane & mask) != 0;
#pragma unroll
    for (int i = 0; i < cnt / 2; ++i) {
      const double send = hi ? p[i] : p[i + cnt / 2];
      const double keep = hi ? p[i + cnt / 2] : p[i];
      p[i] = keep + shfl_xor_d(send, mask);
    }
  }
  return p[0] + shfl_xor_d(p[0], 1);
}

struct ChainTile {
  v2d t[8][4];
  __device__ __forceinline__ void fetch(const double *M, int64_t ld, int wave, int lane) {
    const double *row = M + (int64_t)(wave * 32 + (lane >> 4) * 8) * ld + (lane & 15) * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) t[r][q] = *reinterpret_cast<const v2d *>(row + r * ld + 2 * q);
  }
  __device__ __forceinline__ double times(const double *xs, int lane) const {
    const v2d *xp = reinterpret_cast<const v2d *>(xs + (lane & 15) * 8);
    const v2d x0 = xp[0], x1 = xp[1], x2 = xp[2], x3 = xp[3];
    double p[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      double a = t[r][0][0] * x0[0];
      a = fma(t[r][0][1], x0[1], a);
      double b = t[r][1][0] * x1[0];
      b = fma(t[r][1][1], x1[1], b);
      a = fma(t[r][2][0], x2[0], a);
      a = fma(t[r][2][1], x2[1], a);
      b = fma(t[r][3][0], x3[0], b);
      b = fma(t[r][3][1], x3[1], b);
      p[r] = a + b;
    }
    return reduce8(p, lane);
  }
};

__global__ __launch_bounds__(256) void backsolve_chain_kernel(const double *A, int64_t lda, int T, const double *uinv,
                                                              const double *y, double *v, unsigned long long *gran_,
                                                              unsigned int *abort_, int32_t *info) {
  __shared__ __attribute__((aligned(16))) double xs[NB];
  __shared__ int give_up;
  gu64 *gran = (gu64 *)gran_;
  gu32 *abortw = (gu32 *)abort_;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = T - 1 - (int)blockIdx.x;              // the first workgroups dispatched start the chain
  if (tid == 0) give_up = 0;
  const int myrow = wave * 32 + (lane >> 4) * 8 + ((lane & 15) >> 1);   // the row whose totals this lane keeps
  const bool keeper = !(lane & 1);
  double sum = 0.0;
  ChainTile cur, inv;
  const double *Arow = A + (int64_t)i * NB * lda;
  if (i < T - 1) cur.fetch(Arow + (int64_t)(T - 1) * NB, lda, wave, lane);
  // inv(U_ii) stays in registers from the start: its GEMV is the last link before the hand-off
  inv.fetch(uinv + (int64_t)i * NB * NB, NB, wave, lane);
  const double yi = y[(int64_t)i * NB + myrow];
  __syncthreads();
  for (int k = T - 1; k > i; --k) {
    if (wave == 0) {                                    // one wave polls: doubles lane and lane + 64 of v_k
      gu64 *g = gran + ((int64_t)k * NB + lane) * 2;
      unsigned long long a0, a1, b0, b1;
      unsigned spins = 0;
      bool bad = false;
      for (;;) {
        a0 = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a1 = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b0 = __hip_atomic_load(g + 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b1 = __hip_atomic_load(g + 129, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ok = (a0 >> 32) == CHAIN_EPOCH && (a1 >> 32) == CHAIN_EPOCH && (b0 >> 32) == CHAIN_EPOCH &&
                        (b1 >> 32) == CHAIN_EPOCH;
        if (__all(ok)) break;
        if ((++spins & 63) == 0) {
          const unsigned ab = __hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (ab != 0 || spins > (1u << 22)) { bad = true; break; }
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (bad) {
        if (lane == 0) {
          give_up = 1;
          __hip_atomic_store(abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicCAS(info, 0, CHAIN_TIMEOUT_INFO);
        }
      } else {
        union { double d; unsigned u[2]; } c;
        c.u[0] = (unsigned)a0; c.u[1] = (unsigned)a1;
        xs[lane] = c.d;
        c.u[0] = (unsigned)b0; c.u[1] = (unsigned)b1;
        xs[lane + 64] = c.d;
      }
    }
    __syncthreads();
    if (give_up) return;
    const double part = cur.times(xs, lane);
    // the next operand is requested right away: its latency hides behind the wait for the next piece
    if (k - 1 > i) cur.fetch(Arow + (int64_t)(k - 1) * NB, lda, wave, lane);
    sum += part;
    __syncthreads();
  }
  if (keeper) xs[myrow] = yi - sum;
  __syncthreads();
  const double vi = inv.times(xs, lane);
  if (keeper) {
    if (i > 0) granule_store(gran + ((int64_t)i * NB + myrow) * 2, vi);
    v[(int64_t)i * NB + myrow] = vi;
  }
}

size_t backsolve_scratch_bytes(int64_t n) { return (size_t)n * 16 + 64; }

hipError_t backsolve_upper_batched(hipStream_t st, const double *A, int64_t n, int64_t lda,
                                   const double *work, double *y_inout, int32_t batch, int64_t strideA,
                                   int64_t strideW, int64_t strideY, const int32_t *active, void *scratch,
                                   int32_t *info_dev, bool scratch_zeroed) {
  // y_inout (per batch entry): [0,n) = y (destroyed), [n, 2n) = v on return
  double *y = y_inout, *v = y_inout + n;
  const int64_t nblk = (n + NB - 1) / NB;
  static const bool grouped = [] { const char *e = getenv("LSQAMD_BACKSOLVE"); return !(e && e[0] == 's'); }();
  static const bool chained = [] { const char *e = getenv("LSQAMD_BACKSOLVE"); return !e || e[0] == 'c'; }();
  if (chained && scratch && batch == 1 && !active && n % NB == 0 && nblk >= 2 && nblk <= 192 && info_dev &&
      !(lda & 1) && !(reinterpret_cast<uintptr_t>(A) & 15) && !(reinterpret_cast<uintptr_t>(work) & 15)) {
    // granules for n doubles, then the abort word
    hipError_t e = scratch_zeroed ? hipSuccess : hipMemsetAsync(scratch, 0, backsolve_scratch_bytes(n), st);
    if (e != hipSuccess) return e;
    unsigned long long *gran = static_cast<unsigned long long *>(scratch);
    unsigned int *abortw = reinterpret_cast<unsigned int *>(gran + 2 * n);
    hipLaunchKernelGGL(backsolve_chain_kernel, dim3((unsigned)nblk), dim3(256), 0, st, A, lda, (int)nblk, work, y, v,
                       gran, abortw, info_dev);
    return hipGetLastError();
  }
  if (grouped && n % NB == 0 && nblk >= 2 * BSQ) {
    for (int64_t kb = nblk - 1; kb >= 0; kb -= BSQ) {
      const int64_t top = (kb - BSQ + 1 > 0 ? kb - BSQ + 1 : 0) * NB;
      const unsigned grid = 1 + (unsigned)((top + BS_ROWS - 1) / BS_ROWS);
      hipLaunchKernelGGL(backsolve_group_kernel, dim3(grid, (unsigned)batch), dim3(256), 0, st, A, lda, (int)kb,
                         work, y, v, strideA, strideW, strideY, active);
    }
    return hipGetLastError();
  }
  for (int64_t kb = nblk - 1; kb >= 0; --kb) {
    const int64_t k0 = kb * NB;
    const int nb = (int)((n - k0) < NB ? (n - k0) : NB);
    const unsigned grid = 1 + (unsigned)((k0 + BS_ROWS - 1) / BS_ROWS);
    hipLaunchKernelGGL(backsolve_step_kernel, dim3(grid, (unsigned)batch), dim3(256), 0, st, A, lda, k0, nb,
                       work + kb * NB * NB, y, v, strideA, strideW, strideY, active);
  }
  return hipGetLastError();
}

hipError_t backsolve_upper(hipStream_t st, const double *A, int64_t n, int64_t lda,
                           const double *work, double *y_inout, void *scratch, int32_t *info_dev, bool scratch_zeroed) {
  return backsolve_upper_batched(st, A, n, lda, work, y_inout, 1, 0, 0, 0, nullptr, scratch, info_dev, scratch_zeroed);
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

hipError_t trtri_upper_to_lower_T_batched(hipStream_t st, const double *A, int64_t n, int64_t lda,
                                          const double *work, double *W, int64_t ldw, int32_t batch,
                                          int64_t strideA, int64_t strideW, int64_t strideWl) {
  hipError_t e = hipSuccess;
  for (int32_t b = 0; b < batch; ++b) {
    e = launch_set_identity(st, W + b * strideWl, n, ldw);
    if (e != hipSuccess) return e;
  }
  for (int64_t k0 = 0; k0 < n; k0 += NB) {
    const int nb = (int)((n - k0) < NB ? (n - k0) : NB);
    const double *uinv = work + (k0 / NB) * NB * NB;
    GemmTN p;  // W[k, 0:k0+nb] = inv(U_kk)^T R[k, 0:k0+nb]  (in place)
    p.X = uinv; p.ldx = NB; p.sx = strideW;
    p.Y = W + k0 * ldw; p.ldy = ldw; p.sy = strideWl;
    p.C = W + k0 * ldw; p.ldc = ldw; p.sc = strideWl;
    p.M = nb; p.N = k0 + nb; p.K = nb;
    p.x_upper_tri = 1;
    p.batch = batch;
    e = launch_gemm_tn(st, p);
    if (e != hipSuccess) return e;
    const int64_t mrest = n - (k0 + nb);
    if (mrest <= 0) continue;
    GemmTN t;  // R[k+nb:, 0:k0+nb] -= U[k, k+nb:]^T W[k, 0:k0+nb]
    t.X = A + k0 * lda + k0 + nb; t.ldx = lda; t.sx = strideA;
    t.Y = W + k0 * ldw; t.ldy = ldw; t.sy = strideWl;
    t.C = W + (k0 + nb) * ldw; t.ldc = ldw; t.sc = strideWl;
    t.M = mrest; t.N = k0 + nb; t.K = nb;
    t.alpha = -1.0; t.beta = 1.0;
    t.batch = batch;
    e = launch_gemm_tn(st, t);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t trtri_upper_to_lower_T(hipStream_t st, const double *A, int64_t n, int64_t lda,
                                  const double *work, double *W, int64_t ldw) {
  return trtri_upper_to_lower_T_batched(st, A, n, lda, work, W, ldw, 1, 0, 0, 0);
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
