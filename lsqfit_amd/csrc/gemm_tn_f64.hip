// fp64 MFMA GEMM in "TN" form for gfx950 (MI355X):
//
//     C[M x N] = alpha * sum_k X[k][m] * Y[k][n] + beta * C        (row-major)
//
// This one kernel carries every dense contraction of the LM step:
//   J^T J            X = Y = J                       (replaces gsl_blas_dsyrk in GSL's
//                                                     cholesky solver init, reached from
//                                                     src/lsqfit/_gsl.pyx:677)
//   whitening        X = W_b^T, Y = raw Jacobian     (replaces `dot`, _utilities.pyx:20-36,90-93)
//   Cholesky panel   X = inv(U_kk), Y = row panel
//   trailing update  X = Y = row panel, alpha = -1, beta = 1
//   covariance       X = Y = U^-T                     (replaces gsl_multifit_nlinear_covar,
//                                                     _gsl.pyx:704-706)
//
// Design (MI355X_MICROARCH / cdna_hip_programming guides):
//   * v_mfma_f64_16x16x4_f64: A operand lane l = A[l&15][l>>4], B operand lane l =
//     B[l>>4][l&15], C/D lane l reg r = C[(l>>4) + 4r][l&15].  With k-major
//     operands both fragments are plain row reads of the staged tile -- no
//     transpose anywhere, and global loads are full 1 KiB rows (coalesced).
//   * 128x128 block tile, BK = 16, 256 threads = 4 waves in 2x2, each wave a
//     64x64 sub-tile = 4x4 MFMA tiles (128 accumulator VGPRs): 16 MFMAs (1024
//     matrix-pipe cycles) per 8 ds_read_b64.
//   * LDS rows padded 128 -> 144 doubles so the four k-rows a wave reads in one
//     ds_read_b64 fall in different bank halves (stride 288 dwords = 32 mod 64).
//   * two LDS stages (73.7 KB) -> 2 blocks / CU; global loads for stage t+1 are
//     issued before the MFMAs of stage t and written to LDS after them.
//   * split-K over rows (the long dimension, N_data) into per-split slabs summed
//     by a second pass: deterministic, and gives >> 256 workgroups.
#include <cstdlib>

#include "common.h"
#include <type_traits>
#include <vector>
#include "devmath.h"

namespace lsqamd {

#ifdef LSQAMD_SYRK_STAMPS
// developer build (tools/build_variant.sh syrkstamps gemm_tn_f64.hip -DLSQAMD_SYRK_STAMPS; tools/syrk_schedule.py): every
// workgroup of the work-list launch leaves (start, end) in 100 MHz ticks, XCC_ID, HW_ID and its list index
__device__ unsigned long long *g_syrk_stamps = nullptr;
#endif

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDT = BM + 16;                       // padded LDS row (doubles)
constexpr int STAGE = 2 * BK * LDT;                // X tile + Y tile per stage (doubles)
constexpr size_t GEMM_LDS_BYTES = 2 * STAGE * sizeof(double);

struct GemmDev {
  const double *X, *Y;
  double *C;
  int64_t M, N, K, ldx, ldy, ldc;
  double alpha, beta;
  int64_t sx, sy, sc;
  int32_t tiles_n, upper_only, x_upper_tri, xy_lower_tri, splits;
  int32_t tiles_m, pair_rows;
  int32_t tri_halves;    // XTRI interior launch with two K-halves per tile row (grid.y = 2), see GemmTN::tri_halves
  int32_t xcd_groups;    // XTRI interior launch as a 1-D grid: the tile columns of one (tile row [pair], K-half) on ONE XCD; = number of tile rows [pairs]
  int32_t syrk_diag;     // WORKMAP launch with X == Y: diagonal tiles take the triangular schedule (see kernel)
  int32_t vec_x, vec_y;  // operand rows are 16-byte aligned -> dwordx4 loads
  int64_t kchunk, split_stride;
  const int32_t *work_map;  // optional: work item -> (tm, tn, split, -) with XCD-aware order
  int32_t n_work;
  const int32_t *batch_active;  // optional: skip batch entries whose flag is 0
  double *colsum_out;           // optional (XTRI interior kernel): fused column sums, see GemmTN
  int64_t colsum_ld, colsum_rcol;
  int32_t *done_ctr;            // optional (WORKMAP + SIGNAL): done_ctr[group - 1] += 1 when an entry whose 4th word is `group` is complete
};

// Branch-free staging loads.  Out-of-range rows/columns are CLAMPED to a valid
// address and zeroed when the registers are written to LDS, i.e. after the wait
// the compiler places in front of the ds_write -- so the global loads of stage
// t+1 stay in flight across the MFMAs of stage t (a mask applied at load time
// forces an s_waitcnt vmcnt(0) right behind every load).
template <bool VEC>
__device__ __forceinline__ v2d load2(const double *rowptr, int64_t i0, int64_t i1) {
  if (VEC) return *reinterpret_cast<const v2d *>(rowptr + i0);  // i0 even, i0 + 1 < ld
  v2d v;
  v.x = rowptr[i0];
  v.y = rowptr[i1];
  return v;
}

template <bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_tn_f64_kernel(GemmDev g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tm, tn, split;
  if (g.work_map) {
    // Workgroup b runs on XCD b % 8 (observed dispatch order; speed only, never correctness).
    // Give every XCD one contiguous run of the work list, whose neighbours share operand
    // panels, so that the 64 workgroups resident on an XCD hit its private L2.
    const int nw = g.n_work, bid = blockIdx.x;
    const int xcd = bid & 7, q = nw >> 3, r = nw & 7;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int4 e = reinterpret_cast<const int4 *>(g.work_map)[w];
    tm = e.x; tn = e.y; split = e.z;
  } else {
    tm = blockIdx.x / g.tiles_n;
    tn = blockIdx.x % g.tiles_n;
    split = blockIdx.y;
  }
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  if (g.upper_only && n0 + BN <= m0) return;  // tile strictly below the diagonal

  const int64_t b = blockIdx.z;
  if (g.batch_active && !g.batch_active[b]) return;
  const double *X = g.X + b * g.sx;
  const double *Y = g.Y + b * g.sy;
  double *C = g.C + b * g.sc + (int64_t)split * g.split_stride;

  int64_t kb = (int64_t)split * g.kchunk;
  int64_t ke = kb + g.kchunk < g.K ? kb + g.kchunk : g.K;
  if (g.x_upper_tri) {  // X[k][m] = 0 for k > m
    const int64_t lim = m0 + BM;
    if (ke > lim) ke = lim;
  }
  if (g.xy_lower_tri) {  // X[k][m] = 0 for m > k, same for Y
    int64_t lo = m0 > n0 ? m0 : n0;
    lo -= lo % BK;
    if (kb < lo) kb = lo;
  }

  v4d acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

  // staging map: thread -> (row group, 2 columns); column validity is loop-invariant
  const int c2 = (tid & 63) * 2;
  const int rg = tid >> 6;
  const int64_t xc = m0 + c2, yc = n0 + c2;
  const bool x0ok = xc < g.M, x1ok = xc + 1 < g.M;
  const bool y0ok = yc < g.N, y1ok = yc + 1 < g.N;
  // always-readable indices: VEC -> aligned pair inside the (even) leading dimension,
  // scalar -> each element clamped to the last valid column
  const int64_t xi0 = VEC ? (xc < g.ldx - 2 ? xc : g.ldx - 2) : (xc < g.M ? xc : g.M - 1);
  const int64_t xi1 = xc + 1 < g.M ? xc + 1 : g.M - 1;
  const int64_t yi0 = VEC ? (yc < g.ldy - 2 ? yc : g.ldy - 2) : (yc < g.N ? yc : g.N - 1);
  const int64_t yi1 = yc + 1 < g.N ? yc + 1 : g.N - 1;
  v2d xr[4], yr[4];

  auto gload = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int64_t row = k0 + rg + 4 * i;
      row = row < ke ? row : ke - 1;
      xr[i] = load2<VEC>(X + row * g.ldx, xi0, xi1);
      yr[i] = load2<VEC>(Y + row * g.ldy, yi0, yi1);
    }
  };
  auto sstore = [&](int buf, int64_t k0) {
    double *Xs = smem + buf * STAGE;
    double *Ys = Xs + BK * LDT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = rg + 4 * i;
      const bool rok = k0 + row < ke;
      v2d xv, yv;
      xv.x = (rok && x0ok) ? xr[i].x : 0.0;
      xv.y = (rok && x1ok) ? xr[i].y : 0.0;
      yv.x = (rok && y0ok) ? yr[i].x : 0.0;
      yv.y = (rok && y1ok) ? yr[i].y : 0.0;
      *reinterpret_cast<v2d *>(Xs + row * LDT + c2) = xv;
      *reinterpret_cast<v2d *>(Ys + row * LDT + c2) = yv;
    }
  };

  const int fr = lane & 15, fq = lane >> 4;
  if (kb < ke) {
    gload(kb);
    sstore(0, kb);
  }
  __syncthreads();
  int cur = 0;
  for (int64_t k0 = kb; k0 < ke; k0 += BK) {
    const bool more = k0 + BK < ke;
    if (more) gload(k0 + BK);
    const double *Xs = smem + cur * STAGE;
    const double *Ys = Xs + BK * LDT;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int kr = kk * 4 + fq;
      double a[4], bb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = Xs[kr * LDT + wm * 64 + i * 16 + fr];
#pragma unroll
      for (int j = 0; j < 4; ++j) bb[j] = Ys[kr * LDT + wn * 64 + j * 16 + fr];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[j], acc[i][j], 0, 0, 0);
    }
    if (more) sstore(cur ^ 1, k0 + BK);
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: lane holds rows fq + 4r, column fr of each 16x16 sub-tile.  With beta != 0
  // the 16 C values of a sub-tile row are loaded together (one branch, loads in flight
  // at once) -- a per-element "if (beta) load" costs one memory round trip per element.
  const double alpha = g.alpha;
  const double beta = g.splits > 1 ? 0.0 : g.beta;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double cv[4][4];
    if (beta != 0.0) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t col = n0 + wn * 64 + j * 16 + fr;
          const int64_t row = m0 + wm * 64 + i * 16 + fq + 4 * r;
          const bool ok = col < g.N && row < g.M;
          cv[j][r] = C[(ok ? row : 0) * g.ldc + (ok ? col : 0)];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t col = n0 + wn * 64 + j * 16 + fr;
      if (col >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wm * 64 + i * 16 + fq + 4 * r;
        if (row >= g.M) continue;
        double v = alpha * acc[i][j][r];
        if (beta != 0.0) v += beta * cv[j][r];
        C[row * g.ldc + col] = v;
      }
    }
  }
}

// Interior fast path: every tile full (M, N multiples of 128), every K-range a multiple of
// BK, 16-byte aligned rows.  Operand rows go HBM -> LDS directly (global_load_lds_dwordx4:
// one wave instruction lands one 1 KiB tile row, lane l -> bytes [16 l, 16 l + 16) of the
// row, so the padded row stride costs nothing), no staging VGPRs, no ds_write pass, no
// masks; row pointers advance by a constant instead of being recomputed.
// XTRI (X upper triangular, the whitening product with X = W_b^T): the 16-row MFMA strips of
// the two wave rows are interleaved (strip 2i + wm instead of 4 wm + i) and a strip skips the
// k-steps that lie entirely below its rows -- X is zero there.  Interleaving matters: with
// contiguous 64-row halves the lower half would still do 81 % of the k-steps of a diagonal tile
// while the upper half idles; interleaved, the busier wave row does 62 %.
// WORKMAP: the split-K J^T J launch (XCD-aware work list).  A separate instantiation so that it
// shows up under its own name in kernel traces (the Cholesky's trailing updates use <false, false>).
// Diagonal tiles of that launch (X == Y, tm == tn) are symmetric: the lower-left 64 x 64 quadrant is
// the transpose of the upper-right one and the 16 x 16 sub-tiles below the diagonal of the two
// diagonal quadrants repeat the ones above.  Schedule: waves 0 and 3 compute the 10 sub-tiles
// i <= j of their quadrants, waves 1 and 2 each take two 16-row strips of the upper-right quadrant
// (8 sub-tiles), i.e. at most 10 MFMAs per k-step and wave instead of 16, the Y tile is not staged
// (it is the X tile), and the epilogue writes every value to (row, col) and (col, row): the tile in
// the slab is bit-identical to the one the full schedule writes (same products, same k order).
// CS (with WORKMAP, diagonal tiles on the triangular schedule): the workgroup of diagonal tile (tm, tm, split)
// also forms sum_k X[k][m] * X[k][rcol] over its K-range for its 128 columns m -- J^T f out of the tile rows
// that are in LDS anyway, so the gradient costs no second pass over J.  The residual column's 16 values of a
// stage land in the (unused) Y half of the stage buffer; column rcol itself gets sum_k X[k][rcol]^2 from tile 0.
// colsum_out[(b * splits + split) * colsum_ld + m]: per-split partials, summed in split order by the caller.
// SIGNAL (with WORKMAP): the work list's 4th word names the entry's exchange group (1-based); when the entry's tile is in its
// slab the workgroup adds one to done_ctr[group - 1] with release semantics -- a one-wave kernel on another stream of the
// handle waits for the count of a group's entries (api.hip: the grouped exchange without splitting this launch).
template <bool XTRI, bool WORKMAP, bool CS = false, bool SIGNAL = false>
__global__ __launch_bounds__(256, 2) void gemm_tn_f64_interior_kernel(GemmDev g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tm, tn, split;
  int sig_group = 0;
#ifdef LSQAMD_SYRK_STAMPS
  const unsigned long long stamp0 = __builtin_amdgcn_s_memrealtime();
  int stamp_w = -1;
#endif
  if (WORKMAP) {
    // (as in gemm_tn_f64_kernel: workgroup b runs on XCD b % 8; every XCD walks one contiguous run of the list)
    const int nw = g.n_work, bid = blockIdx.x;
    const int xcd = bid & 7, q = nw >> 3, r = nw & 7;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
#ifdef LSQAMD_SYRK_STAMPS
    stamp_w = w;
#endif
    const int4 e = reinterpret_cast<const int4 *>(g.work_map)[w];
    tm = e.x; tn = e.y; split = e.z;
    if (SIGNAL) sig_group = e.w;
  } else if (XTRI && g.xcd_groups) {
    // Workgroup b runs on XCD b % 8.  With the tile column as the fastest index (and 8 of them) every XCD would own ONE
    // tile column and read ALL of X for it -- config 3's triangular whitening matrix went past the L2 eight times (L2 hit
    // 12 %, 3.9 GB per launch, r05_c3_pmc.json).  Here the tile columns of one (tile row [pair], K-half) are neighbours on
    // ONE XCD: they walk the same rows of X at the same pace and share them in its L2.
    const int L = blockIdx.x, xcd = L & 7, sl = L >> 3;
    const int gidx = xcd + 8 * (sl / g.tiles_n);
    tn = sl % g.tiles_n;
    tm = gidx % g.xcd_groups;
    split = gidx / g.xcd_groups;
  } else {
    tm = blockIdx.x / g.tiles_n;
    tn = blockIdx.x % g.tiles_n;
    split = blockIdx.y;
  }
  // pair_rows (triangular X with many tile rows): this workgroup does tile rows tm and
  // T-1-tm one after the other, so every workgroup sees the same total K (the k-range of a
  // tile row grows with tm: unpaired, the last rows take 64x as long as the first)
  const int tm_other = (XTRI && g.pair_rows) ? g.tiles_m - 1 - tm : tm;
  const int npass = tm_other != tm ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
  if (pass >= 1) {
    tm = tm_other;
    __syncthreads();
  }
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  if (g.upper_only && n0 + BN <= m0) return;
  const int64_t b = blockIdx.z;
  if (g.batch_active && !g.batch_active[b]) return;
  double *C = g.C + b * g.sc + (int64_t)split * g.split_stride;
  int64_t kb = (int64_t)split * g.kchunk;
  int64_t ke = kb + g.kchunk < g.K ? kb + g.kchunk : g.K;
  if (g.x_upper_tri) {
    const int64_t lim = m0 + BM;
    if (ke > lim) ke = lim;
  }
  if (g.xy_lower_tri) {
    int64_t lo = m0 > n0 ? m0 : n0;
    lo -= lo % BK;
    if (kb < lo) kb = lo;
  }
  if (XTRI && g.tri_halves) {   // the tile row's own K-range [0, m0 + 128) cut in two: split 0 the lower half, split 1 the upper
    const int64_t lim = m0 + BM, h = (lim >> 1) & ~(int64_t)(BK - 1);
    kb = split ? h : 0;
    ke = split ? lim : h;
  }

  v4d acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

  const bool diag = WORKMAP && !XTRI && g.syrk_diag && tm == tn;
  // tile rows / columns of this wave's 16-wide strips: strip i -> rows arow0 + 16 i, strip j -> columns bcol0 + 16 j
  const int arow0 = diag ? (wave == 2 ? 32 : (wave == 3 ? 64 : 0)) : wm * 64;
  const int bcol0 = diag ? (wave == 0 ? 0 : 64) : wn * 64;

  // wave w stages rows w, w+4, w+8, w+12 of both tiles; lane -> 2 columns
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const double *xp = g.X + b * g.sx + (kb + wave) * g.ldx + m0 + 2 * lane;
  const double *yp = g.Y + b * g.sy + (kb + wave) * g.ldy + n0 + 2 * lane;
  const int64_t xstep = 4 * g.ldx, ystep = 4 * g.ldy;
  // CS: lanes 0..31 of wave 1 fetch the 16 residuals of a stage as 32 dwords (lane -> row lane / 2, half lane & 1)
  const char *fp = reinterpret_cast<const char *>(g.X + b * g.sx + (kb + (lane >> 1)) * g.ldx + g.colsum_rcol) + 4 * (lane & 1);
  const int64_t fstep = (int64_t)BK * g.ldx * (int64_t)sizeof(double);
  double cs = 0.0, cs2 = 0.0;

  auto stage = [&](int buf) {
    double *Xs = smem + buf * STAGE + wave * LDT;
    double *Ys = Xs + BK * LDT;
    if (diag) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((glb_void *)(xp + i * xstep), (lds_void *)(Xs + 4 * i * LDT), 16, 0, 0);
      if (CS && wave == 1) {
        if (lane < 32)
          __builtin_amdgcn_global_load_lds((glb_void *)fp, (lds_void *)(smem + buf * STAGE + BK * LDT), 4, 0, 0);
        fp += fstep;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_global_load_lds((glb_void *)(xp + i * xstep), (lds_void *)(Xs + 4 * i * LDT), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void *)(yp + i * ystep), (lds_void *)(Ys + 4 * i * LDT), 16, 0, 0);
      }
    }
    xp += 4 * xstep;
    yp += 4 * ystep;
  };

  // weights of the fused column sums: fetched now, used in the epilogue (latency hidden by the k loop)
  const bool fuse = XTRI && g.colsum_out != nullptr;
  double wreg = 0.0;
  if (fuse && tid < BM) wreg = (g.C + b * g.sc)[(m0 + tid) * g.ldc + g.colsum_rcol];
  const int fr = lane & 15, fq = lane >> 4;
  if (kb < ke) stage(0);
  __syncthreads();
  // ROLE 0: all 16 sub-tiles; 1: sub-tiles i <= j (diagonal quadrant of a diagonal tile); 2: strips i < 2
  auto kloop = [&](auto role_tag) {
    constexpr int ROLE = decltype(role_tag)::value;
    int cur = 0;
    for (int64_t k0 = kb; k0 < ke; k0 += BK) {
      if (k0 + BK < ke) stage(cur ^ 1);
      const double *Xs = smem + cur * STAGE;
      const double *Ys = ROLE == 0 ? Xs + BK * LDT : Xs;
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        const int kr = kk * 4 + fq;
        double a[4], bb[4];
#pragma unroll
        for (int i = 0; i < (ROLE == 2 ? 2 : 4); ++i)
          a[i] = Xs[kr * LDT + (XTRI ? (2 * i + wm) * 16 : arow0 + i * 16) + fr];
#pragma unroll
        for (int j = 0; j < 4; ++j) bb[j] = Ys[kr * LDT + bcol0 + j * 16 + fr];
#pragma unroll
        for (int i = 0; i < (ROLE == 2 ? 2 : 4); ++i) {
          // X[k][m] = 0 for k > m: strip rows m0 + 16 s .. + 15 see nothing from k-steps beyond them
          if (XTRI && k0 + kk * 4 > m0 + (2 * i + wm) * 16 + 15) continue;
#pragma unroll
          for (int j = (ROLE == 1 ? i : 0); j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
      }
      if (CS && ROLE != 0) {   // thread -> column tid & 127, rows 8 (tid >> 7) .. + 7 of the stage
        const double *Fs = Xs + BK * LDT + 8 * (tid >> 7);
        const double *Xc = Xs + 8 * (tid >> 7) * LDT + (tid & 127);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const double fv = Fs[r];
          cs = fma(Xc[r * LDT], fv, cs);
          cs2 = fma(fv, fv, cs2);
        }
      }
      __syncthreads();
      cur ^= 1;
    }
  };
  if (WORKMAP && !XTRI && diag) {
    if (wave == 0 || wave == 3) kloop(std::integral_constant<int, 1>{});
    else kloop(std::integral_constant<int, 2>{});
    // epilogue of a diagonal tile: each value to (row, col) and to its mirror image
    const double alpha = g.alpha;
    const bool tri = wave == 0 || wave == 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!tri && i >= 2) break;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (tri && j < i) continue;
        const int64_t col = n0 + bcol0 + j * 16 + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = m0 + arow0 + i * 16 + fq + 4 * r;
          const double v = alpha * acc[i][j][r];
          C[row * g.ldc + col] = v;
          if (!(tri && i == j)) C[col * g.ldc + row] = v;
        }
      }
    }
    if (CS) {   // (the k loop ended on a barrier: the stage buffers are free)
      smem[tid] = cs;
      if ((tid & 127) == 0) smem[256 + (tid >> 7)] = cs2;
      __syncthreads();
      double *out = g.colsum_out + (b * g.splits + split) * g.colsum_ld;
      if (tid < BM) out[m0 + tid] = smem[tid] + smem[tid + 128];
      if (tm == 0 && tid == 0) out[g.colsum_rcol] = smem[256] + smem[257];
    }
    continue;
  }
  kloop(std::integral_constant<int, 0>{});

  const double alpha = g.alpha;
  const double beta = g.splits > 1 ? 0.0 : g.beta;
  double *wcol = smem;             // [128] weights of this tile's rows (LDS is free after the k loop)
  double *red = smem + 128;        // [4 waves][64] partial column sums
  if (fuse) {
    if (tid < BM) wcol[tid] = wreg;
    __syncthreads();
  }
  double sj[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rloc = (XTRI ? (2 * i + wm) * 16 : wm * 64 + i * 16) + fq;
    const int64_t rbase = m0 + rloc;
    double cv[4][4];
    if (beta != 0.0) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          cv[j][r] = C[(rbase + 4 * r) * g.ldc + n0 + wn * 64 + j * 16 + fr];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double w = fuse ? wcol[rloc + 4 * r] : 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double v = alpha * acc[i][j][r];
        if (beta != 0.0) v += beta * cv[j][r];
        C[(rbase + 4 * r) * g.ldc + n0 + wn * 64 + j * 16 + fr] = v;
        if (XTRI) sj[j] += v * w;
      }
    }
  }
  if (fuse) {
    // fused J^T f: the tile times the weight column of its rows, summed over the rows -- in-thread
    // over (i, r) above, across the four 16-lane groups by two butterfly steps, across the two
    // wave rows through LDS
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sj[j] += __shfl_xor(sj[j], 16, 64);
      sj[j] += __shfl_xor(sj[j], 32, 64);
    }
    if (fq == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wave * 64 + j * 16 + fr] = sj[j];
    }
    __syncthreads();
    if (wm == 0 && fq == 0) {
      double *out = g.colsum_out + (b * g.tiles_m + tm) * g.colsum_ld + n0 + wn * 64;
#pragma unroll
      for (int j = 0; j < 4; ++j) out[j * 16 + fr] = red[wave * 64 + j * 16 + fr] + red[(wave + 2) * 64 + j * 16 + fr];
    }
  }
  }  // pass
  if (SIGNAL && WORKMAP && sig_group > 0 && g.done_ctr) {
    // the guide's counter hand-off (cdna_hip_programming.md 6 G16 / 5.x "in-launch split-K reduction"): every wave drains its
    // own stores, barrier, ONE lane issues the agent-scope release (buffer_wbl2: this XCD's L2 is not coherent with the one
    // the consumer kernel will read through), waits for it, then counts with a RELAXED add.  NOT __threadfence() in every
    // thread: that is a write-back AND an invalidate per thread -- it cost the product 6 % (the panels' L2 lines went with it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(g.done_ctr + (sig_group - 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#ifdef LSQAMD_SYRK_STAMPS
  if (WORKMAP && g_syrk_stamps && tid == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long *o = g_syrk_stamps + 5 * (size_t)blockIdx.x;
    o[0] = stamp0;
    o[1] = __builtin_amdgcn_s_memrealtime();
    o[2] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    o[3] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    o[4] = (unsigned long long)stamp_w;
  }
#endif
}

// 64 x 64-tile variant for the latency-bound contractions of the Cholesky family (K = 128,
// a handful of 128-tiles): four times the workgroups, a quarter of the MFMA chain each.
// Same contract as gemm_tn_f64_kernel (edges, triangular flags, beta, batch).
constexpr int TS = 64;
constexpr int LDS_S = TS + 16;
constexpr int STAGE_S = 2 * BK * LDS_S;
constexpr size_t GEMM_LDS_BYTES_S = 2 * STAGE_S * sizeof(double);

template <bool VEC>
__global__ __launch_bounds__(256, 4) void gemm_tn_f64_small_kernel(GemmDev g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tm = blockIdx.x / g.tiles_n, tn = blockIdx.x % g.tiles_n;
  const int split = blockIdx.y;
  const int64_t m0 = (int64_t)tm * TS, n0 = (int64_t)tn * TS;
  if (g.upper_only && n0 + TS <= m0) return;
  const int64_t b = blockIdx.z;
  if (g.batch_active && !g.batch_active[b]) return;
  const double *X = g.X + b * g.sx;
  const double *Y = g.Y + b * g.sy;
  double *C = g.C + b * g.sc + (int64_t)split * g.split_stride;
  int64_t kb = (int64_t)split * g.kchunk;
  int64_t ke = kb + g.kchunk < g.K ? kb + g.kchunk : g.K;
  if (g.x_upper_tri) {
    const int64_t lim = m0 + TS;
    if (ke > lim) ke = lim;
  }
  if (g.xy_lower_tri) {
    int64_t lo = m0 > n0 ? m0 : n0;
    lo -= lo % BK;
    if (kb < lo) kb = lo;
  }
  v4d acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

  const int c2 = (tid & 31) * 2;
  const int rg = tid >> 5;  // 0..7 ; rows rg, rg + 8
  const int64_t xc = m0 + c2, yc = n0 + c2;
  const bool x0ok = xc < g.M, x1ok = xc + 1 < g.M;
  const bool y0ok = yc < g.N, y1ok = yc + 1 < g.N;
  const int64_t xi0 = VEC ? (xc < g.ldx - 2 ? xc : g.ldx - 2) : (xc < g.M ? xc : g.M - 1);
  const int64_t xi1 = xc + 1 < g.M ? xc + 1 : g.M - 1;
  const int64_t yi0 = VEC ? (yc < g.ldy - 2 ? yc : g.ldy - 2) : (yc < g.N ? yc : g.N - 1);
  const int64_t yi1 = yc + 1 < g.N ? yc + 1 : g.N - 1;
  v2d xr[2], yr[2];
  auto gload = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int64_t row = k0 + rg + 8 * i;
      row = row < ke ? row : ke - 1;
      xr[i] = load2<VEC>(X + row * g.ldx, xi0, xi1);
      yr[i] = load2<VEC>(Y + row * g.ldy, yi0, yi1);
    }
  };
  auto sstore = [&](int buf, int64_t k0) {
    double *Xs = smem + buf * STAGE_S;
    double *Ys = Xs + BK * LDS_S;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = rg + 8 * i;
      const bool rok = k0 + row < ke;
      v2d xv, yv;
      xv.x = (rok && x0ok) ? xr[i].x : 0.0;
      xv.y = (rok && x1ok) ? xr[i].y : 0.0;
      yv.x = (rok && y0ok) ? yr[i].x : 0.0;
      yv.y = (rok && y1ok) ? yr[i].y : 0.0;
      *reinterpret_cast<v2d *>(Xs + row * LDS_S + c2) = xv;
      *reinterpret_cast<v2d *>(Ys + row * LDS_S + c2) = yv;
    }
  };
  const int fr = lane & 15, fq = lane >> 4;
  if (kb < ke) {
    gload(kb);
    sstore(0, kb);
  }
  __syncthreads();
  int cur = 0;
  for (int64_t k0 = kb; k0 < ke; k0 += BK) {
    const bool more = k0 + BK < ke;
    if (more) gload(k0 + BK);
    const double *Xs = smem + cur * STAGE_S;
    const double *Ys = Xs + BK * LDS_S;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int kr = kk * 4 + fq;
      double a[2], bb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = Xs[kr * LDS_S + wm * 32 + i * 16 + fr];
#pragma unroll
      for (int j = 0; j < 2; ++j) bb[j] = Ys[kr * LDS_S + wn * 32 + j * 16 + fr];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[j], acc[i][j], 0, 0, 0);
    }
    if (more) sstore(cur ^ 1, k0 + BK);
    __syncthreads();
    cur ^= 1;
  }
  const double alpha = g.alpha;
  const double beta = g.splits > 1 ? 0.0 : g.beta;
  double cv[2][2][4];
  if (beta != 0.0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t col = n0 + wn * 32 + j * 16 + fr;
          const int64_t row = m0 + wm * 32 + i * 16 + fq + 4 * r;
          const bool ok = col < g.N && row < g.M;
          cv[i][j][r] = C[(ok ? row : 0) * g.ldc + (ok ? col : 0)];
        }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = n0 + wn * 32 + j * 16 + fr;
      if (col >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wm * 32 + i * 16 + fq + 4 * r;
        if (row >= g.M) continue;
        double v = alpha * acc[i][j][r];
        if (beta != 0.0) v += beta * cv[i][j][r];
        C[row * g.ldc + col] = v;
      }
    }
}

// 128 x 64-tile variant for IN-PLACE row panels (C aliases Y, M <= 128: the Cholesky row panel
// U[k, k+1:] = inv(U_kk)^T A[k, k+1:]): one workgroup still owns whole columns of the operand --
// which is what makes the in-place update safe -- but there are twice as many of them as with
// 128 x 128 tiles and each carries half the MFMA chain.  Single tile row only (blockIdx.x = tile
// column).  Same edge / triangular / beta / batch contract as the other kernels.
// PNW = 64, 32 or 16 columns per workgroup: late in a factorisation the panel has a handful of
// 64-column tiles for 256 CUs, and what it costs is the length of one workgroup's MFMA chain --
// narrower tiles spread the same columns over more CUs and shorten that chain.
constexpr int PN = 64;
constexpr int LDS_PX = BM + 16;   // 144
constexpr int LDS_PY = PN + 16;   // 80 (widest variant)
constexpr int STAGE_P = BK * (LDS_PX + LDS_PY);
constexpr size_t GEMM_LDS_BYTES_P = 2 * STAGE_P * sizeof(double);

// ---- one-shot row panel -----------------------------------------------------------------------
// The Cholesky row panel U[k, k+1:] = inv(U_kk)^T A[k, k+1:] at full tile size (M = K = 128, X upper
// triangular, in place) is latency, not work: 36 MFMAs per wave behind EIGHT dependent K stages of
// the kernel above (9.4 us at P = 4096, 35 launches per factorisation).  Here the whole inverse
// block (128 KiB, the same for every workgroup: L2 hits) and the workgroup's 16 operand columns go
// to LDS with ALL loads in flight at once -- one memory round trip -- then the MFMAs, balanced over
// the waves (16-row blocks i and 7 - i need 4 (i + 1) and 4 (8 - i) k-chunks of the triangle).
constexpr int OS_LDX = 136, OS_PN = 16;
constexpr size_t GEMM_LDS_BYTES_OS = (size_t)(128 * OS_LDX + 128 * OS_PN) * sizeof(double);

__global__ __launch_bounds__(256) void gemm_tn_f64_panel_oneshot_kernel(GemmDev g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.z;
  if (g.batch_active && !g.batch_active[b]) return;
  const double *X = g.X + b * g.sx;
  const int64_t n0 = (int64_t)blockIdx.x * OS_PN;
  double *Yg = g.C + b * g.sc + n0;              // C aliases Y
  double *Xs = smem, *Ys = smem + 128 * OS_LDX;
  // X: row k = 1 KiB = one wave-instruction (lane -> 2 doubles); 32 rows per wave.  (Leaving the lanes
  // left of the diagonal block out of the request -- half the bytes -- was measured: no gain.)
#pragma unroll 8
  for (int i = 0; i < 32; ++i) {
    const int k = wave * 32 + i;
    __builtin_amdgcn_global_load_lds((glb_void *)(X + (int64_t)k * g.ldx + 2 * lane), (lds_void *)(Xs + k * OS_LDX), 16, 0, 0);
  }
  // Y: 8 rows of 16 doubles per wave-instruction (lane -> row lane / 8, doubles 2 (lane % 8) ..)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k8 = (wave * 4 + i) * 8;
    __builtin_amdgcn_global_load_lds((glb_void *)(Yg + (int64_t)(k8 + (lane >> 3)) * g.ldc + 2 * (lane & 7)),
                                     (lds_void *)(Ys + k8 * OS_PN), 16, 0, 0);
  }
  __syncthreads();   // (s_waitcnt vmcnt(0) is part of it)
  const int col = lane & 15, q = lane >> 4;
  const int i0 = wave, i1 = 7 - wave;            // this wave's two 16-row blocks of the output
  v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  const int n0c = 4 * (i0 + 1), n1c = 4 * (i1 + 1);
#pragma unroll 4
  for (int kc = 0; kc < n1c; ++kc) {
    const double bb = Ys[(4 * kc + q) * OS_PN + col];
    const double a1 = Xs[(4 * kc + q) * OS_LDX + 16 * i1 + col];
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bb, acc1, 0, 0, 0);
    if (kc < n0c) {
      const double a0 = Xs[(4 * kc + q) * OS_LDX + 16 * i0 + col];
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bb, acc0, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    Yg[(int64_t)(16 * i0 + 4 * r + q) * g.ldc + col] = acc0[r];
    Yg[(int64_t)(16 * i1 + 4 * r + q) * g.ldc + col] = acc1[r];
  }
}

template <bool VEC, int PNW>
__global__ __launch_bounds__(256, 2) void gemm_tn_f64_panel_kernel(GemmDev g) {
  constexpr int WN = PNW >= 32 ? 2 : 1, WM = 4 / WN;       // waves along n / m
  constexpr int RW = BM / WM, NI = RW / 16;                  // rows per wave, 16-row blocks
  constexpr int CW = PNW / WN, NJ = CW / 16;                 // columns per wave, 16-column blocks
  constexpr int YT = PNW / 2, YROWS = 256 / YT;              // Y stage: column pairs, rows per pass
  constexpr int YP = (BK + YROWS - 1) / YROWS;               // passes over the 16 stage rows
  constexpr int LDY = PNW + 16;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
  const int64_t m0 = 0, n0 = (int64_t)blockIdx.x * PNW;
  const int64_t b = blockIdx.z;
  if (g.batch_active && !g.batch_active[b]) return;
  const double *X = g.X + b * g.sx;
  const double *Y = g.Y + b * g.sy;
  double *C = g.C + b * g.sc;
  int64_t kb = 0, ke = g.K;
  if (g.x_upper_tri && ke > BM) ke = BM;
  v4d acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
  // X stage: 16 x 128 -> thread: column pair (tid & 63) * 2, rows (tid >> 6) + 4 i
  // Y stage: 16 x 64  -> thread: column pair (tid & 31) * 2, rows (tid >> 5) + 8 i
  const int xc2 = (tid & 63) * 2, xrg = tid >> 6;
  const int yc2 = (tid % YT) * 2, yrg = tid / YT;
  const int64_t xc = m0 + xc2, yc = n0 + yc2;
  const bool x0ok = xc < g.M, x1ok = xc + 1 < g.M;
  const bool y0ok = yc < g.N, y1ok = yc + 1 < g.N;
  const int64_t xi0 = VEC ? (xc < g.ldx - 2 ? xc : g.ldx - 2) : (xc < g.M ? xc : g.M - 1);
  const int64_t xi1 = xc + 1 < g.M ? xc + 1 : g.M - 1;
  const int64_t yi0 = VEC ? (yc < g.ldy - 2 ? yc : g.ldy - 2) : (yc < g.N ? yc : g.N - 1);
  const int64_t yi1 = yc + 1 < g.N ? yc + 1 : g.N - 1;
  v2d xr[4], yr[YP];
  auto gload = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int64_t row = k0 + xrg + 4 * i;
      row = row < ke ? row : ke - 1;
      xr[i] = load2<VEC>(X + row * g.ldx, xi0, xi1);
    }
#pragma unroll
    for (int i = 0; i < YP; ++i) {
      int64_t row = k0 + yrg + YROWS * i;
      row = row < ke ? row : ke - 1;
      yr[i] = load2<VEC>(Y + row * g.ldy, yi0, yi1);
    }
  };
  auto sstore = [&](int buf, int64_t k0) {
    double *Xs = smem + buf * STAGE_P;
    double *Ys = Xs + BK * LDS_PX;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = xrg + 4 * i;
      const bool rok = k0 + row < ke;
      v2d xv;
      xv.x = (rok && x0ok) ? xr[i].x : 0.0;
      xv.y = (rok && x1ok) ? xr[i].y : 0.0;
      *reinterpret_cast<v2d *>(Xs + row * LDS_PX + xc2) = xv;
    }
#pragma unroll
    for (int i = 0; i < YP; ++i) {
      const int row = yrg + YROWS * i;
      if (row >= BK) continue;            // narrow variants: the upper thread rows have nothing to stage
      const bool rok = k0 + row < ke;
      v2d yv;
      yv.x = (rok && y0ok) ? yr[i].x : 0.0;
      yv.y = (rok && y1ok) ? yr[i].y : 0.0;
      *reinterpret_cast<v2d *>(Ys + row * LDY + yc2) = yv;
    }
  };
  const int fr = lane & 15, fq = lane >> 4;
  if (kb < ke) {
    gload(kb);
    sstore(0, kb);
  }
  __syncthreads();
  int cur = 0;
  for (int64_t k0 = kb; k0 < ke; k0 += BK) {
    const bool more = k0 + BK < ke;
    if (more) gload(k0 + BK);
    const double *Xs = smem + cur * STAGE_P;
    const double *Ys = Xs + BK * LDS_PX;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int kr = kk * 4 + fq;
      double a[NI], bb[NJ];
#pragma unroll
      for (int i = 0; i < NI; ++i) a[i] = Xs[kr * LDS_PX + wm * RW + i * 16 + fr];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bb[j] = Ys[kr * LDY + wn * CW + j * 16 + fr];
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[j], acc[i][j], 0, 0, 0);
    }
    if (more) sstore(cur ^ 1, k0 + BK);
    __syncthreads();
    cur ^= 1;
  }
  // every read of the operand columns this workgroup owns has happened (the loop's last barrier);
  // only now may C, which may alias Y, be written
  const double alpha = g.alpha, beta = g.beta;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int64_t col = n0 + wn * CW + j * 16 + fr;
      if (col >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wm * RW + i * 16 + fq + 4 * r;
        if (row >= g.M) continue;
        double v = alpha * acc[i][j][r];
        if (beta != 0.0) v += beta * C[row * g.ldc + col];
        C[row * g.ldc + col] = v;
      }
    }
}

static PerDeviceOnce g_attr_once;   // dynamic-LDS attributes of the product kernels, per device

static int64_t small_gemm_max() {
  static const int64_t v = [] {
    const char *e = getenv("LSQAMD_SMALL_GEMM_MAX");  // developer knob
    return e ? (int64_t)atoll(e) : (int64_t)600;
  }();
  return v;
}

static bool interior_eligible(const GemmTN &a) {
  const bool vx = !(a.ldx & 1) && !(reinterpret_cast<uintptr_t>(a.X) & 15) && !(a.sx & 1);
  const bool vy = !(a.ldy & 1) && !(reinterpret_cast<uintptr_t>(a.Y) & 15) && !(a.sy & 1);
  return vx && vy && (a.M % BM == 0) && (a.N % BN == 0) && (a.K % BK == 0) && !a.force_generic;
}

// few long tile rows (one large dense block: config 3's 8192 x 8192 W against 1024 columns is 32 row pairs x 8 tile
// columns = 256 workgroups, one per CU, four waves each with nothing to cover their barrier and LDS latencies):
// every tile row's K-range in two halves, 2x the workgroups, the second half into a scratch slab the caller adds.
bool gemm_tn_wants_tri_halves(const GemmTN &a) {
  const char *e = getenv("LSQAMD_TRI_HALVES");   // developer knob, read per call (A/B tests): 0 = one workgroup per tile-row pair
  const bool off = e && atoi(e) == 0;
  const int64_t tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  const int64_t nwg = ((tiles_m + 1) / 2) * tiles_n * (a.batch < 1 ? 1 : a.batch);
  return !off && a.tri_halves && interior_eligible(a) && a.x_upper_tri && a.splits <= 1 && !a.work_map && tiles_m >= 8 &&
         nwg < 400 && a.split_stride != 0 && a.C != a.X && a.C != a.Y && !a.colsum_out;
}

static bool syrk_diag_off() {
  static const bool v = [] { const char *e = getenv("LSQAMD_SYRK_DIAG"); return e && atoi(e) == 0; }();  // developer knob
  return v;
}

bool gemm_tn_fuses_colsum(const GemmTN &a) {
  if (a.work_map) {
    // the split-K J^T J launch: diagonal tiles on the triangular schedule carry the column sums (<false, true, true>)
    const char *e = getenv("LSQAMD_SYRK_COLSUM");   // developer knob, read per call (A/B tests): 0 = separate J^T f pass
    const bool off = e && atoi(e) == 0;
    const int splits = a.splits < 1 ? 1 : a.splits;
    return interior_eligible(a) && !a.x_upper_tri && a.upper_only && a.X == a.Y && a.ldx == a.ldy && a.sx == a.sy &&
           a.M == a.N && (splits > 1 || a.beta == 0.0) && !syrk_diag_off() && !off && a.C != a.X;
  }
  // the conditions under which launch_gemm_tn reaches gemm_tn_f64_interior_kernel<true>
  const int64_t tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  const int64_t nblk128 = tiles_m * tiles_n * (a.splits < 1 ? 1 : a.splits) * (a.batch < 1 ? 1 : a.batch);
  const bool small = !a.work_map && a.K <= 512 && nblk128 <= small_gemm_max();
  return interior_eligible(a) && a.x_upper_tri && a.splits <= 1 && !a.work_map && !small &&
         a.C != a.Y && a.C != a.X && a.alpha != 0.0;
}

hipError_t launch_gemm_tn(hipStream_t st, const GemmTN &a) {
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  {
    const hipError_t ea = g_attr_once.run([] {
      const void *fns[] = {reinterpret_cast<const void *>(gemm_tn_f64_kernel<true>),
                           reinterpret_cast<const void *>(gemm_tn_f64_kernel<false>),
                           reinterpret_cast<const void *>(gemm_tn_f64_interior_kernel<false, false>),
                           reinterpret_cast<const void *>(gemm_tn_f64_interior_kernel<false, true>),
                           reinterpret_cast<const void *>(gemm_tn_f64_interior_kernel<true, false>),
                           reinterpret_cast<const void *>(gemm_tn_f64_interior_kernel<false, true, true>),
                           reinterpret_cast<const void *>(gemm_tn_f64_interior_kernel<false, true, false, true>),
                           reinterpret_cast<const void *>(gemm_tn_f64_interior_kernel<false, true, true, true>)};
      for (const void *fn : fns) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS_BYTES);
        if (e != hipSuccess) return e;
      }
      return hipSuccess;
    });
    if (ea != hipSuccess) return ea;
  }
  GemmDev g;
  g.X = a.X; g.Y = a.Y; g.C = a.C;
  g.M = a.M; g.N = a.N; g.K = a.K;
  g.ldx = a.ldx; g.ldy = a.ldy; g.ldc = a.ldc;
  g.alpha = a.alpha; g.beta = a.beta;
  g.sx = a.sx; g.sy = a.sy; g.sc = a.sc;
  g.vec_x = !((a.ldx & 1) || (reinterpret_cast<uintptr_t>(a.X) & 15) || (a.sx & 1));
  g.vec_y = !((a.ldy & 1) || (reinterpret_cast<uintptr_t>(a.Y) & 15) || (a.sy & 1));
  const int64_t tiles_m = (a.M + BM - 1) / BM;
  const int64_t tiles_n = (a.N + BN - 1) / BN;
  g.tiles_n = (int32_t)tiles_n;
  g.tiles_m = (int32_t)tiles_m;
  g.pair_rows = 0;
  g.tri_halves = 0;
  g.xcd_groups = 0;
  g.upper_only = a.upper_only;
  g.x_upper_tri = a.x_upper_tri;
  g.xy_lower_tri = a.xy_lower_tri;
  g.splits = a.splits < 1 ? 1 : a.splits;
  int64_t kchunk = (a.K + g.splits - 1) / g.splits;
  kchunk = (kchunk + BK - 1) / BK * BK;
  if (kchunk < BK) kchunk = BK;
  g.kchunk = kchunk;
  g.split_stride = g.splits > 1 ? a.split_stride : 0;
  g.work_map = a.work_map;
  g.n_work = a.n_work;
  const bool diag_off = syrk_diag_off();
  g.syrk_diag = a.work_map && a.X == a.Y && a.ldx == a.ldy && a.sx == a.sy && a.M == a.N &&
                (g.splits > 1 || a.beta == 0.0) && !diag_off;
  g.batch_active = a.batch_active;
  g.colsum_out = a.colsum_out;
  g.colsum_ld = a.colsum_ld;
  g.colsum_rcol = a.colsum_rcol;
  g.done_ctr = a.work_map ? a.done_ctr : nullptr;
  dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)g.splits, (unsigned)(a.batch < 1 ? 1 : a.batch));
  if (a.work_map) grid = dim3((unsigned)a.n_work, 1, (unsigned)(a.batch < 1 ? 1 : a.batch));
  const bool interior = g.vec_x && g.vec_y && (a.M % BM == 0) && (a.N % BN == 0) && (a.K % BK == 0) &&
                        !a.force_generic && !(a.x_upper_tri && a.work_map);  // no <XTRI, WORKMAP> instantiation
  // few tiles and a short K: latency-bound -> 64 x 64 tiles (4x the workgroups); threshold from a
  // sweep of potrf_upper at P = 4096 (160: 2.67 ms, 600: 2.56 ms, 1000: 2.56 ms)
  const int64_t nblk128 = tiles_m * tiles_n * g.splits * (a.batch < 1 ? 1 : a.batch);
  const int64_t small_max = small_gemm_max();
  // in-place products (C aliases an operand: the Cholesky row panel, the triangular inverse) are
  // only safe when ONE workgroup owns a whole column range of the operand, i.e. a single tile row
  const bool inplace = (a.C == a.Y) || (a.C == a.X);
  if (inplace && tiles_m > 1) return hipErrorInvalidValue;
  const bool small_ok = !inplace || a.M <= TS;
  if (inplace && a.C == a.Y && a.M > TS && a.K <= BM && g.splits == 1 && !a.force_generic) {
    // row panel: 128 x PNW tiles, one workgroup per PNW operand columns; PNW shrinks until there
    // are a few hundred workgroups (LSQAMD_PANEL_PN forces 64 / 32 / 16: developer knob)
    static const int force_pn = [] { const char *e = getenv("LSQAMD_PANEL_PN"); return e ? atoi(e) : 0; }();
    const int64_t nb_ = a.batch < 1 ? 1 : a.batch;
    if (force_pn == 0 && a.M == BM && a.K == BM && a.N % OS_PN == 0 && a.x_upper_tri && g.vec_x && g.vec_y &&
        a.alpha == 1.0 && a.beta == 0.0 && a.ldx >= BM) {
      static PerDeviceOnce attr_os;
      const hipError_t eo = attr_os.run([] {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_tn_f64_panel_oneshot_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS_BYTES_OS);
      });
      if (eo != hipSuccess) return eo;
      dim3 grido((unsigned)(a.N / OS_PN), 1, (unsigned)nb_);
      hipLaunchKernelGGL(gemm_tn_f64_panel_oneshot_kernel, grido, dim3(256), GEMM_LDS_BYTES_OS, st, g);
      return hipGetLastError();
    }
    int pn = (a.N * nb_ >= 200 * 64) ? 64 : ((a.N * nb_ >= 200 * 32) ? 32 : 16);
    if (force_pn == 64 || force_pn == 32 || force_pn == 16) pn = force_pn;
    g.tiles_n = (int32_t)((a.N + pn - 1) / pn);
    dim3 gridp((unsigned)g.tiles_n, 1, (unsigned)nb_);
    const bool vec = g.vec_x && g.vec_y;
#define LSQAMD_PANEL(V, W) hipLaunchKernelGGL((gemm_tn_f64_panel_kernel<V, W>), gridp, dim3(256), GEMM_LDS_BYTES_P, st, g)
    if (pn == 64) { if (vec) LSQAMD_PANEL(true, 64); else LSQAMD_PANEL(false, 64); }
    else if (pn == 32) { if (vec) LSQAMD_PANEL(true, 32); else LSQAMD_PANEL(false, 32); }
    else { if (vec) LSQAMD_PANEL(true, 16); else LSQAMD_PANEL(false, 16); }
#undef LSQAMD_PANEL
    return hipGetLastError();
  }
  // (... or ONE 64 x 64 tile whatever K is: J^T J for up to 64 parameters, where a 128 x 128 tile is three
  // quarters padding -- 0.33 -> 0.1 ms at N = 524288)
  const bool one_small_tile = a.M <= TS && a.N <= TS;
  if (!a.work_map && (a.K <= 512 || one_small_tile) && nblk128 <= small_max && !a.force_generic && small_ok) {
    const int64_t tm64 = (a.M + TS - 1) / TS, tn64 = (a.N + TS - 1) / TS;
    g.tiles_n = (int32_t)tn64;
    dim3 grid64((unsigned)(tm64 * tn64), (unsigned)g.splits, (unsigned)(a.batch < 1 ? 1 : a.batch));
    if (g.vec_x && g.vec_y)
      hipLaunchKernelGGL(gemm_tn_f64_small_kernel<true>, grid64, dim3(256), GEMM_LDS_BYTES_S, st, g);
    else
      hipLaunchKernelGGL(gemm_tn_f64_small_kernel<false>, grid64, dim3(256), GEMM_LDS_BYTES_S, st, g);
    return hipGetLastError();
  }
  if (interior && a.x_upper_tri) {
    if (tiles_m >= 4 && g.splits == 1 && !a.work_map) {  // balance the growing k-ranges (see kernel)
      g.pair_rows = 1;
      grid.x = (unsigned)(((tiles_m + 1) / 2) * tiles_n);
      if (gemm_tn_wants_tri_halves(a)) {
        g.tri_halves = 1;
        g.splits = 2;                       // (epilogue: beta ignored, split s -> C + s * split_stride)
        g.split_stride = a.split_stride;
        grid.y = 2;
      }
    }
    {
      static const bool xg_off = [] { const char *e = getenv("LSQAMD_XTRI_XCD_GROUPS"); return e && e[0] == '0'; }();   // developer knob
      const unsigned rows = grid.x / (unsigned)tiles_n, groups = rows * grid.y;
      if (!xg_off && !a.work_map && tiles_n >= 2 && groups % 8 == 0 && grid.x == rows * (unsigned)tiles_n) {
        g.xcd_groups = (int32_t)rows;
        grid.x = groups * (unsigned)tiles_n;
        grid.y = 1;
      }
    }
    hipLaunchKernelGGL((gemm_tn_f64_interior_kernel<true, false>), grid, dim3(256), GEMM_LDS_BYTES, st, g);
  }
  else if (interior && a.work_map && a.colsum_out) {
    if (!g.syrk_diag || !a.upper_only) return hipErrorInvalidValue;   // (callers ask gemm_tn_fuses_colsum first)
    if (g.done_ctr) hipLaunchKernelGGL((gemm_tn_f64_interior_kernel<false, true, true, true>), grid, dim3(256), GEMM_LDS_BYTES, st, g);
    else hipLaunchKernelGGL((gemm_tn_f64_interior_kernel<false, true, true>), grid, dim3(256), GEMM_LDS_BYTES, st, g);
  }
  else if (interior && a.work_map && g.done_ctr)
    hipLaunchKernelGGL((gemm_tn_f64_interior_kernel<false, true, false, true>), grid, dim3(256), GEMM_LDS_BYTES, st, g);
  else if (interior && a.work_map)
    hipLaunchKernelGGL((gemm_tn_f64_interior_kernel<false, true>), grid, dim3(256), GEMM_LDS_BYTES, st, g);
  else if (interior)
    hipLaunchKernelGGL((gemm_tn_f64_interior_kernel<false, false>), grid, dim3(256), GEMM_LDS_BYTES, st, g);
  else if (g.vec_x && g.vec_y)
    hipLaunchKernelGGL(gemm_tn_f64_kernel<true>, grid, dim3(256), GEMM_LDS_BYTES, st, g);
  else
    hipLaunchKernelGGL(gemm_tn_f64_kernel<false>, grid, dim3(256), GEMM_LDS_BYTES, st, g);
  return hipGetLastError();
}

// Work list for the upper-triangular tiles of a P x P SYRK with `splits` K-chunks:
// split-major, then 8 x 8 patches of tiles, then tiles of the patch.  Consecutive
// entries share row/column panels AND the K-chunk.  Within the run of entries each XCD walks (the
// kernel's blockIdx -> entry map: eight contiguous runs) the diagonal tiles go LAST: they take 10/16
// of the time of the others (triangular schedule), so the round of workgroups that drains each XCD
// is the short one.
int64_t syrk_work_count(int64_t P, int32_t splits) {
  const int64_t T = (P + BM - 1) / BM;
  return T * (T + 1) / 2 * (splits < 1 ? 1 : splits);
}

void syrk_work_fill(int64_t P, int32_t splits, int32_t *out) { (void)syrk_work_fill_rows(P, splits, 0, (int)((P + BM - 1) / BM), out); }

// the same list for the tile rows [row0, row1) only (the exchange groups of a sharded fit, api.hip eval_normal_dev: one launch
// per group of tile rows, so that a group's packed tiles can be on their way to the other ranks while the next group is being
// computed); returns the number of entries.  Same ordering rules as the full list: 8 x 8 patches of tiles for the L2, one
// eighth of the list per XCD, a chunk's diagonal tiles last.
int64_t syrk_work_fill_rows(int64_t P, int32_t splits, int row0, int row1, int32_t *out) {
  const int T = (int)((P + BM - 1) / BM), PS = 8;
  const int NP = (T + PS - 1) / PS;
  if (row0 < 0) row0 = 0;
  if (row1 > T) row1 = T;
  int64_t o = 0;
  for (int s = 0; s < (splits < 1 ? 1 : splits); ++s)
    for (int pi = 0; pi < NP; ++pi)
      for (int pj = pi; pj < NP; ++pj)
        for (int tm = pi * PS; tm < (pi + 1) * PS && tm < T; ++tm) {
          if (tm < row0 || tm >= row1) continue;
          for (int tn = pj * PS; tn < (pj + 1) * PS && tn < T; ++tn) {
            if (tn < tm) continue;
            out[o++] = tm; out[o++] = tn; out[o++] = s; out[o++] = 0;
          }
        }
  const int64_t nw = o / 4, q = nw / 8, r = nw % 8;
  std::vector<int32_t> tmp;
  for (int xcd = 0; xcd < 8; ++xcd) {
    const int64_t b = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, n = q + (xcd < r ? 1 : 0);
    tmp.assign(out + 4 * b, out + 4 * (b + n));
    int64_t w = b;
    for (int keep = 0; keep < 2; ++keep)          // stable: off-diagonal entries, then diagonal ones
      for (int64_t e = 0; e < n; ++e)
        if ((tmp[4 * e] == tmp[4 * e + 1]) == (keep == 1)) {
          for (int c = 0; c < 4; ++c) out[4 * w + c] = tmp[4 * e + c];
          ++w;
        }
  }
  return nw;
}

// ONE list for ONE launch whose entries finish group by group: every XCD's run of the list (the kernel's blockIdx -> entry
// map: eight contiguous runs) holds its share of group 0 first, then of group 1, ... -- so all of group 0's tiles are in their
// slabs when about count[0] / total of the launch has run, and their exchange can start while the rest is computed.
// rows[g] .. rows[g + 1]: the tile rows of group g.  The 4th word of an entry is its group + 1; count[g] = entries of group g.
void syrk_work_fill_grouped(int64_t P, int32_t splits, int G, const int32_t *rows, int32_t *out, int32_t *count) {
  const int64_t nw = syrk_work_count(P, splits), q = nw / 8, r = nw % 8;
  std::vector<std::vector<int32_t>> lists((size_t)G);
  for (int g = 0; g < G; ++g) {
    lists[(size_t)g].resize(4 * (size_t)nw);
    // (rows-only fill WITHOUT its per-XCD diagonal-last pass: that pass is applied per (XCD run, group) segment below)
    const int T = (int)((P + BM - 1) / BM), PS = 8, NP = (T + PS - 1) / PS;
    int64_t o = 0;
    int32_t *L = lists[(size_t)g].data();
    for (int s = 0; s < (splits < 1 ? 1 : splits); ++s)
      for (int pi = 0; pi < NP; ++pi)
        for (int pj = pi; pj < NP; ++pj)
          for (int tm = pi * PS; tm < (pi + 1) * PS && tm < T; ++tm) {
            if (tm < rows[g] || tm >= rows[g + 1]) continue;
            for (int tn = pj * PS; tn < (pj + 1) * PS && tn < T; ++tn) {
              if (tn < tm) continue;
              L[o++] = tm; L[o++] = tn; L[o++] = s; L[o++] = g + 1;
            }
          }
    lists[(size_t)g].resize((size_t)o);
    count[g] = (int32_t)(o / 4);
  }
  // shares: group g gives XCD x  c[g][x] entries, sum_x = count[g], sum_g = the run length the kernel computes for x
  std::vector<int64_t> cap(8), pos((size_t)G, 0);
  for (int x = 0; x < 8; ++x) cap[(size_t)x] = q + (x < r ? 1 : 0);
  std::vector<std::vector<int64_t>> c((size_t)G, std::vector<int64_t>(8, 0));
  for (int g = 0; g < G; ++g) {
    if (g == G - 1) {
      for (int x = 0; x < 8; ++x) c[(size_t)g][(size_t)x] = cap[(size_t)x];
      break;
    }
    int64_t left = count[g];
    for (int x = 0; x < 8; ++x) {
      int64_t v = count[g] / 8;
      if (v > cap[(size_t)x]) v = cap[(size_t)x];
      c[(size_t)g][(size_t)x] = v;
      cap[(size_t)x] -= v;
      left -= v;
    }
    while (left > 0) {            // the remainder to the runs with the most room
      int best = 0;
      for (int x = 1; x < 8; ++x)
        if (cap[(size_t)x] > cap[(size_t)best]) best = x;
      c[(size_t)g][(size_t)best]++;
      cap[(size_t)best]--;
      --left;
    }
  }
  int64_t w = 0;
  for (int x = 0; x < 8; ++x)
    for (int g = 0; g < G; ++g) {
      const int32_t *L = lists[(size_t)g].data() + 4 * pos[(size_t)g];
      const int64_t n = c[(size_t)g][(size_t)x];
      for (int keep = 0; keep < 2; ++keep)          // stable: off-diagonal entries, then diagonal ones
        for (int64_t e = 0; e < n; ++e)
          if ((L[4 * e] == L[4 * e + 1]) == (keep == 1)) {
            for (int k = 0; k < 4; ++k) out[4 * w + k] = L[4 * e + k];
            ++w;
          }
      pos[(size_t)g] += n;
    }
}

// one wave waits until *ctr >= expect (acquire); gives up after ~4 s of wall clock and says so in *timed_out
__global__ __launch_bounds__(64) void wait_counter_kernel(const int32_t *ctr, int32_t expect, int32_t *timed_out) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz
  // poll RELAXED (an acquire per poll would invalidate this CU's L1 -- and cost the product beside it -- every time); the
  // kernels queued behind this one on the stream start with clean caches: the kernel boundary is their acquire
  while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect) {
    __builtin_amdgcn_s_sleep(64);
    if (__builtin_amdgcn_s_memrealtime() - t0 > 400000000ull) {
      __hip_atomic_store(timed_out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
}

hipError_t launch_wait_counter(hipStream_t st, const int32_t *ctr, int32_t expect, int32_t *timed_out) {
  hipLaunchKernelGGL(wait_counter_kernel, dim3(1), dim3(64), 0, st, ctr, expect, timed_out);
  return hipGetLastError();
}

// ---- whitening product with the raw Jacobian rows synthesised in LDS ---------------------------------
// J_b = W_b x Jraw_b for the sum models, the Y operand (16 raw rows x 128 columns per stage) computed
// by the workgroup instead of being read: the raw Jacobian (2.15 GB at the named shape) is neither
// written nor re-read.  Same tile, staging order and MFMA sequence as
// gemm_tn_f64_interior_kernel<true, false> -- the whitened rows come out bit-identical to the
// two-kernel route -- with the 128 tile columns = 64 terms k and their 64 partners K + k
// (value columns and frequency / exponent columns of the same terms share the sincos / exp):
//   thread t: term c = t & 63, stage rows t >> 6, + 4, + 8, + 12  ->  4 transcendental pairs per stage.
// fp64 VALU and fp64 MFMA share the datapath, so this work adds to the MFMA time (about a quarter)
// instead of hiding behind it; what is saved is the Jacobian kernel and 4.3 GB of traffic.
// NT = terms per tile: 64 (128 output columns) or 32 (64 output columns: twice the workgroups for the shapes
// that would otherwise leave one workgroup per CU -- config 3's single 8192-row block -- where nothing but
// a second workgroup covers the barrier and LDS latencies of the first); same products in the same order.
template <int MODEL, int NT, bool FAR>
__global__ __launch_bounds__(256, 2) void whiten_synth_kernel(WhitenSynth a, int tiles_m, int tiles_n, int pair_rows) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (a.trig_far && (a.trig_far[0] != 0) != FAR) return;    // (FAR = false: no far-range trig code, see model.hip)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: the triangular skips become scalar branches)
  const int wm = wave >> 1, wn = wave & 1;
  const int tm_first = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  // pair_rows (blocks of many tile rows: the k-range of a tile row grows with it): this workgroup does tile
  // rows tm and tiles_m - 1 - tm one after the other, so that every workgroup sees the same total K
  const int npass = (pair_rows && tiles_m - 1 - tm_first != tm_first) ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
  if (pass == 1) __syncthreads();
  const int tm = pass == 0 ? tm_first : tiles_m - 1 - tm_first;
  const int64_t b = blockIdx.z;
  const int64_t m0 = (int64_t)tm * BM;
  const int64_t ke = m0 + BM;                     // X[k][m] = 0 for k > m
  const double *Xb = a.Wt + b * a.B * a.B;
  const double *xrow = a.x + b * a.B;
  double *C = a.J + b * a.B * a.ld;
  constexpr int NJ = NT / 16;                      // 16-column MFMA tiles per wave
  constexpr int SR = 256 / NT, NS = BK / SR;       // stage rows synthesised per pass of the workgroup, passes
  const int sc = tid % NT, srow = tid / NT;        // synthesis: term within the tile, first stage row
  const double amp = a.p[(int64_t)tn * NT + sc], frq = a.p[a.K + (int64_t)tn * NT + sc];

  v4d acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const double *xp = Xb + (int64_t)wave * a.B + m0 + 2 * lane;
  const int64_t xstep = 4 * a.B;
  // abscissae of the stage rows: a ring of three 32-double slots in LDS, slot s % 3 = x[16 s .. 16 s + 31],
  // filled two stages ahead by one 256-byte DMA of wave 0 -- no register ever waits on a global load in
  // the loop (vmcnt retires in order: a wait for x would be a wait for the operand rows behind it)
  double *xring = smem + 2 * STAGE;
  auto xdma = [&](int64_t sidx) {
    if (wave == 0) {
      int64_t e = sidx * BK + (lane >> 1);
      if (e > a.B - 1) e = a.B - 1;
      const char *src = reinterpret_cast<const char *>(xrow + e) + 4 * (lane & 1);
      __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(xring + (sidx % 3) * 32), 4, 0, 0);
    }
  };
  int64_t ssyn = 0;     // index of the stage `stage_synth` builds next
  auto stage_dma = [&](int buf) {
    double *Xs = smem + buf * STAGE + wave * LDT;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((glb_void *)(xp + i * xstep), (lds_void *)(Xs + 4 * i * LDT), 16, 0, 0);
    xp += 4 * xstep;
  };
  auto stage_synth = [&](int buf) {
    double xs4[NS];
    const double *xq = xring + (ssyn % 3) * 32 + srow;
#pragma unroll
    for (int i = 0; i < NS; ++i) xs4[i] = xq[SR * i];
    double *Ys = smem + buf * STAGE + BK * LDT;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int row = srow + SR * i;
      const double xv = xs4[i];
      double t, dq;
      if (MODEL == LSQAMD_MODEL_COSMIX) {
        double sn, cs;
        sincos_moderate<FAR>(frq * xv, &sn, &cs);
        t = cs;
        dq = -amp * xv * sn;
      } else {
        const double e = exp(-frq * xv);
        t = e;
        dq = -amp * xv * e;
      }
      Ys[row * LDT + sc] = t;
      Ys[row * LDT + NT + sc] = dq;
    }
    ++ssyn;
  };
  const double wreg = (tid < BM) ? C[(m0 + tid) * a.ld + 2 * a.K] : 0.0;   // whitened residual of this tile's rows
  const int fr = lane & 15, fq = lane >> 4;
  xdma(0);
  xdma(1);
  stage_dma(0);          // the first operand rows travel together with the abscissae: one exposed latency, not two
  __syncthreads();
  stage_synth(0);
  __syncthreads();
  int cur = 0;
  for (int64_t k0 = 0; k0 < ke; k0 += BK) {
    if (k0 + BK < ke) {
      if (k0 + 2 * BK < ke) xdma(k0 / BK + 2);
      stage_dma(cur ^ 1);
      stage_synth(cur ^ 1);
    }
    const double *Xs = smem + cur * STAGE;
    const double *Ys = Xs + BK * LDT;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int kr = kk * 4 + fq;
      double av[4], bb[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = Xs[kr * LDT + (2 * i + wm) * 16 + fr];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bb[j] = Ys[kr * LDT + wn * NT + j * 16 + fr];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (k0 + kk * 4 > m0 + (2 * i + wm) * 16 + 15) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bb[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  double *wcol = smem, *red = smem + 128;
  if (tid < BM) wcol[tid] = wreg;
  __syncthreads();
  // tile column (wn, j, fr)  ->  Jacobian column: the value half (wn = 0) or the partner half at K + ..
  const int64_t cbase = (wn ? a.K : 0) + (int64_t)tn * NT;
  double sj[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) sj[j] = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rloc = (2 * i + wm) * 16 + fq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double w = wcol[rloc + 4 * r];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const double v = acc[i][j][r];
        C[(m0 + rloc + 4 * r) * a.ld + cbase + j * 16 + fr] = v;
        sj[j] += v * w;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    sj[j] += __shfl_xor(sj[j], 16, 64);
    sj[j] += __shfl_xor(sj[j], 32, 64);
  }
  if (fq == 0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) red[wave * 64 + j * 16 + fr] = sj[j];
  }
  __syncthreads();
  if (wm == 0 && fq == 0) {
    double *out = a.colsum_out + (b * tiles_m + tm) * (2 * a.K) + cbase;
#pragma unroll
    for (int j = 0; j < NJ; ++j) out[j * 16 + fr] = red[wave * 64 + j * 16 + fr] + red[(wave + 2) * 64 + j * 16 + fr];
  }
  }  // pass
}

// (Round 4 measured a variant in which ONE eight-wave workgroup owns both tile rows of a 256-row block and synthesises every
// raw row once -- a third less transcendental work: bit-identical rows, config 4's whitening 1.908 ms against 1.679.  The
// 512-thread barrier per stage with nothing else resident on the CU costs more than the sincos saved.  Removed in round 5;
// the record is in DESIGN_HISTORY.md.)

bool whiten_synth_eligible(int32_t model, int64_t B, int64_t P) {
  const char *e = getenv("LSQAMD_FUSED_JACOBIAN");   // developer knob, read per call: 0 = never, 2 = also for large blocks
  const bool off = e && e[0] == '0', always = e && e[0] == '2';
  // Large blocks: every tile row of the block synthesises the raw rows below it again -- B / 256 times each on average.
  // From eight tile rows on, writing the raw Jacobian once (it stays in the last-level cache) and reading it back is the
  // faster route (config 3, one 8192-row block: 1.52 ms synthesised, 1.45 ms read, 1.1 ms read with split tile rows).
  return !off && (model == LSQAMD_MODEL_COSMIX || model == LSQAMD_MODEL_MULTIEXP) && B >= BM && B % BM == 0 &&
         (B < 8 * BM || always) && P % 128 == 0 && P >= 128;
}

constexpr size_t SYNTH_LDS_BYTES = GEMM_LDS_BYTES + 3 * 32 * sizeof(double);

template <int MODEL, int NT, bool FAR>
static hipError_t launch_whiten_synth_one(hipStream_t st, const WhitenSynth &a, dim3 grid, int tiles_m, int tiles_n, int pair) {
  static PerDeviceOnce attr;   // one per instantiation
  const hipError_t ea = attr.run([] {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(whiten_synth_kernel<MODEL, NT, FAR>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)SYNTH_LDS_BYTES);
  });
  if (ea != hipSuccess) return ea;
  hipLaunchKernelGGL((whiten_synth_kernel<MODEL, NT, FAR>), grid, dim3(256), SYNTH_LDS_BYTES, st, a, tiles_m, tiles_n, pair);
  return hipGetLastError();
}

template <int MODEL, int NT>
static hipError_t launch_whiten_synth_nt(hipStream_t st, const WhitenSynth &a, dim3 grid, int tiles_m, int tiles_n, int pair) {
  // cosine model with a range flag: the kernel without far-range trig code first, then the safe one -- the
  // device flag leaves exactly one of them with work
  if (MODEL == LSQAMD_MODEL_COSMIX && a.trig_far) {
    hipError_t e = launch_whiten_synth_one<MODEL, NT, false>(st, a, grid, tiles_m, tiles_n, pair);
    if (e != hipSuccess) return e;
  }
  return launch_whiten_synth_one<MODEL, NT, true>(st, a, grid, tiles_m, tiles_n, pair);
}

hipError_t launch_whiten_synth(hipStream_t st, const WhitenSynth &a0) {
  WhitenSynth a = a0;
  if (a.model != LSQAMD_MODEL_COSMIX) a.trig_far = nullptr;
  const int tiles_m = (int)(a.B / BM);
  // few large blocks: unpaired, the workgroups of the last tile rows run 2 tiles_m / (tiles_m + 1) times the average
  const int pair = tiles_m >= 4 ? 1 : 0;
  const int64_t rows = pair ? (tiles_m + 1) / 2 : tiles_m;
  // 64 terms per tile unless that leaves the chip with (about) one workgroup per CU
  static const int64_t narrow_below = [] { const char *e = getenv("LSQAMD_SYNTH_NARROW"); return e ? atoll(e) : (int64_t)400; }();
  const bool narrow = rows * (a.K / 64) * a.nb < narrow_below && a.K % 32 == 0;
  const int tiles_n = (int)(a.K / (narrow ? 32 : 64));
  dim3 grid((unsigned)(rows * tiles_n), 1, (unsigned)a.nb);
  if (a.model == LSQAMD_MODEL_COSMIX)
    return narrow ? launch_whiten_synth_nt<LSQAMD_MODEL_COSMIX, 32>(st, a, grid, tiles_m, tiles_n, pair)
                  : launch_whiten_synth_nt<LSQAMD_MODEL_COSMIX, 64>(st, a, grid, tiles_m, tiles_n, pair);
  return narrow ? launch_whiten_synth_nt<LSQAMD_MODEL_MULTIEXP, 32>(st, a, grid, tiles_m, tiles_n, pair)
                : launch_whiten_synth_nt<LSQAMD_MODEL_MULTIEXP, 64>(st, a, grid, tiles_m, tiles_n, pair);
}

}  // namespace lsqamd

#ifdef LSQAMD_SYRK_STAMPS
extern "C" int lsqamd_debug_set_syrk_stamps(void *dev_ptr) try {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(lsqamd::g_syrk_stamps), &dev_ptr, sizeof(void *));
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)
#endif
