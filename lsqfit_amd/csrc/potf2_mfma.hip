// Diagonal-block Cholesky + triangular inverse, matrix-core formulation (gfx950).
//
// Replaces the LDS-resident diagonal kernel of chol.hip on the critical path of
// potrf_upper (the damped normal-equation solve that stands in for GSL's
// solver->init/presolve/solve, src/lsqfit/_gsl.pyx:646-653,:677).  Compiled with
// -mllvm -amdgpu-mfma-vgpr-form (see build.py): with the whole tile file in
// architectural VGPRs the sweep is straight-line MFMA code; hipcc's default
// heuristic parks the tiles in AccVGPRs and spends 10k v_accvgpr_read/_mov on them.
#include <cstdlib>

#include "common.h"

#ifndef LSQAMD_LEAF_FMA_LAYOUT
#define LSQAMD_LEAF_FMA_LAYOUT 1
#endif

namespace lsqamd {

constexpr int NB = CHOL_NB;
typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_d(double v, int lane) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}

__device__ __forceinline__ double rsqrt_refined(double d) {
  double y = __builtin_amdgcn_rsq(d);  // v_rsq_f64, then two Newton steps to full precision
  const double h = -0.5 * d;
  double e = __builtin_fma(h * y, y, 0.5);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(h * y, y, 0.5);
  return __builtin_fma(y, e, y);
}

// The same to full precision with a shorter dependent chain (5 levels after v_rsq_f64 instead of
// 6): with e = 1 - d y0^2,  1/sqrt(d) = y0 (1 - e)^-1/2 = y0 (1 + e/2 + 3 e^2/8 + 5 e^3/16 + 35 e^4/128 + ..),
// the polynomial in Estrin form.  v_rsq_f64 is good to ~2^-16 only (a cubic truncation was measured
// at 1e-14 relative -- visible in log det J^T J at cond 1e8), so the first neglected term is ~2^-77.
__device__ __forceinline__ double rsqrt_cubic(double d) {
  const double y0 = __builtin_amdgcn_rsq(d);
  const double e = __builtin_fma(-(d * y0), y0, 1.0);
  const double e2 = e * e;
  const double pa = __builtin_fma(0.375, e, 0.5);
  const double pb = __builtin_fma(0.2734375, e, 0.3125);
  const double p = __builtin_fma(pb, e2, pa);
  return __builtin_fma(y0 * e, p, y0);
}

// ---- diagonal-block kernel, matrix-core formulation -------------------------------------------
// The fp64 MFMA accumulator layout (16 x 16 tile: lane l, register r <-> row 4r + (l >> 4),
// column l & 15) makes register r of a tile a 4-row slab whose lane mapping IS the k-major
// operand layout of v_mfma_f64_16x16x4 (k = l >> 4, m or n = l & 15).  So with the augmented
// block [A | I] (8 x 16 tiles of 16 x 16) resident in accumulator layout, blocked elimination
// with block size 4 needs no data movement at all:
//     slab  <- D_u^-T slab            one MFMA per tile of the slab row (A operand = D_u^-T)
//     tile(i, j) -= slab_i^T slab_j   one MFMA per trailing tile, operands = slab registers
// [A | I] -> [U | U^-T]: the inverse the row-panel GEMMs need falls out of the same sweep.
// Tile columns are dealt to the 4 waves (wave w: columns w and w + 4 of both halves = 18 live
// tiles, 72 fp64 registers per lane); per slab the owner of the diagonal tile factors the 4 x 4
// pivot block (scalar-uniform arithmetic on 10 broadcast values), everyone scales its part of
// the slab, the negated slab goes through LDS (4 KiB) as A operands, everyone updates its tiles.
// Two workgroup barriers per slab, 32 slabs; the matrix itself never touches LDS.
struct TileRegs {
  v4d X[2][8];  // slot c <-> tile column tj = wave + 4c; X[c][ti]: A-part tile (ti, tj) if ti < tj,
                // E-part tile (ti, tj) if ti > tj (unused for ti == tj)
  v4d DA[2];    // A-part diagonal tile (tj, tj)
  v4d DE[2];    // E-part diagonal tile (tj, tj)
};

__device__ __forceinline__ v4d mfma4(double a, double b, v4d c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// 4 x 4 pivot block D = Du^T Du and V = Du^-1 -> mop: scalar-uniform arithmetic on ten broadcast values.
// (Slotting the wave's pending update MFMAs between the stages of this chain was measured and
// is slower: fp64 MFMA and fp64 VALU do not overlap usefully within one wave.)
template <int R>
__device__ __forceinline__ void pivot4(double reg, double *mop, int32_t *info, int32_t row0, int lane) {
  const double d00 = readlane_d(reg, 0 * 16 + 4 * R + 0), d01 = readlane_d(reg, 0 * 16 + 4 * R + 1);
  const double d02 = readlane_d(reg, 0 * 16 + 4 * R + 2), d03 = readlane_d(reg, 0 * 16 + 4 * R + 3);
  const double d11 = readlane_d(reg, 1 * 16 + 4 * R + 1), d12 = readlane_d(reg, 1 * 16 + 4 * R + 2);
  const double d13 = readlane_d(reg, 1 * 16 + 4 * R + 3), d22 = readlane_d(reg, 2 * 16 + 4 * R + 2);
  const double d23 = readlane_d(reg, 2 * 16 + 4 * R + 3), d33 = readlane_d(reg, 3 * 16 + 4 * R + 3);
  int bad = -1;
  const bool ok0 = (d00 > 0.0) && (d00 < 1.0e300);
  bad = ok0 ? bad : 0;
  const double y0 = rsqrt_refined(ok0 ? d00 : 1.0);
  const double u01 = d01 * y0, u02 = d02 * y0, u03 = d03 * y0;
  const double t11 = __builtin_fma(-u01, u01, d11);
  const bool ok1 = (t11 > 0.0) && (t11 < 1.0e300);
  bad = (!ok1 && bad < 0) ? 1 : bad;
  const double y1 = rsqrt_refined(ok1 ? t11 : 1.0);
  const double u12 = __builtin_fma(-u01, u02, d12) * y1, u13 = __builtin_fma(-u01, u03, d13) * y1;
  const double t22 = __builtin_fma(-u12, u12, __builtin_fma(-u02, u02, d22));
  const bool ok2 = (t22 > 0.0) && (t22 < 1.0e300);
  bad = (!ok2 && bad < 0) ? 2 : bad;
  const double y2 = rsqrt_refined(ok2 ? t22 : 1.0);
  const double u23 = __builtin_fma(-u12, u13, __builtin_fma(-u02, u03, d23)) * y2;
  const double t33 = __builtin_fma(-u23, u23, __builtin_fma(-u13, u13, __builtin_fma(-u03, u03, d33)));
  const bool ok3 = (t33 > 0.0) && (t33 < 1.0e300);
  bad = (!ok3 && bad < 0) ? 3 : bad;
  const double y3 = rsqrt_refined(ok3 ? t33 : 1.0);
  if (bad >= 0 && lane == 0) atomicCAS(info, 0, row0 + bad + 1);
  // V = Du^-1 (upper), v_ii = y_i
  const double v01 = -y0 * (u01 * y1);
  const double v12 = -y1 * (u12 * y2);
  const double v23 = -y2 * (u23 * y3);
  const double v02 = -y0 * __builtin_fma(u01, v12, u02 * y2);
  const double v13 = -y1 * __builtin_fma(u12, v23, u13 * y3);
  const double v03 = -y0 * __builtin_fma(u01, v13, __builtin_fma(u02, v23, u03 * y3));
  // A operand of the scaling product: M[m][k] = V[k][m] (m = lane & 15 < 4, k = lane >> 4 <= m);
  // two-level select (by k, then by m) instead of a ten-deep chain
  const int col = lane & 15, q = lane >> 4;
  const double r0 = col == 0 ? y0 : (col == 1 ? v01 : (col == 2 ? v02 : v03));
  const double r1 = col == 1 ? y1 : (col == 2 ? v12 : v13);
  const double r2 = col == 2 ? y2 : v23;
  double m = q == 0 ? r0 : (q == 1 ? r1 : (q == 2 ? r2 : y3));
  m = (col < 4 && q <= col) ? m : 0.0;
  mop[lane] = m;
}

// ---- the update list of one (slab S, wave W): position p <-> tile row TI + p / 2, slot p % 2
constexpr bool upd_valid(int S, int W, int p) {
  const int TI = S / 4, tr = TI + p / 2, tj = W + 4 * (p % 2);
  if (tr > 7) return false;
  if (tr < tj) return true;                 // A-part tile
  if (tr > tj) return tj <= TI;             // E-part tile: its slab piece is zero for tj > TI
  return true;                              // diagonal slot
}
constexpr int upd_count(int S, int W, int skip) {
  int n = 0;
  for (int p = 0; p < 16; ++p) n += (p != skip && upd_valid(S, W, p)) ? 1 : 0;
  return n;
}
constexpr int upd_nth(int S, int W, int skip, int k) {  // position of the k-th entry, -1 past the end
  for (int p = 0; p < 16; ++p)
    if (p != skip && upd_valid(S, W, p)) {
      if (k == 0) return p;
      --k;
    }
  return -1;
}

template <int S, int W, int P>
__device__ __forceinline__ void upd_at(TileRegs &T, const double *slabA, int lane) {
  if constexpr (P >= 0) {
    constexpr int TI = S / 4, R = S % 4, tr = TI + P / 2, c = P % 2, tj = W + 4 * c;
    const double a = slabA[tr * 64 + lane];
    if constexpr (tr < tj) {          // A-part tile (tr, tj); slab piece: A-part (TI, tj), TI <= tr < tj
      T.X[c][tr] = mfma4(a, T.X[c][TI][R], T.X[c][tr]);
    } else if constexpr (tr > tj) {   // E-part tile (tr, tj); slab piece: E-part (TI, tj), tj <= TI
      if constexpr (tj < TI) T.X[c][tr] = mfma4(a, T.X[c][TI][R], T.X[c][tr]);
      else T.X[c][tr] = mfma4(a, T.DE[c][R], T.X[c][tr]);
    } else {                          // diagonal slot tr == tj
      if constexpr (TI < tj) T.DA[c] = mfma4(a, T.X[c][TI][R], T.DA[c]);
      else { T.DA[c] = mfma4(a, T.DA[c][R], T.DA[c]); T.DE[c] = mfma4(a, T.DE[c][R], T.DE[c]); }
    }
  }
}

template <int S, int W, int SKIP, int K0, int K1>
__device__ __forceinline__ void upd_range(TileRegs &T, const double *slabA, int lane) {
  if constexpr (K0 < K1) {
    upd_at<S, W, upd_nth(S, W, SKIP, K0)>(T, slabA, lane);
    upd_range<S, W, SKIP, K0 + 1, K1>(T, slabA, lane);
  }
}

// One slab: pivot block (owner of the diagonal tile), scale, publish, update.
template <int S, int W>
__device__ __forceinline__ void slab_step(TileRegs &T, double *mop, double *slabA, int32_t *info,
                                          int32_t k0, int lane) {
  constexpr int TI = S / 4, R = S % 4;
  const int col = lane & 15;
  if constexpr (TI % 4 == W) pivot4<R>(T.DA[TI / 4][R], mop, info, k0 + 4 * S, lane);
  __syncthreads();
  const double mo = mop[lane];
  const v4d zero = {0.0, 0.0, 0.0, 0.0};
  // ---- scale this wave's pieces of the slab row; publish the negated A-part as A operands
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int tj = W + 4 * c;
    if (tj != TI) {
      const v4d t = mfma4(mo, T.X[c][TI][R], zero);
      T.X[c][TI][R] = t[0];
      if (tj > TI) slabA[tj * 64 + lane] = -t[0];
    } else {
      const v4d ta = mfma4(mo, T.DA[c][R], zero);
      const v4d te = mfma4(mo, T.DE[c][R], zero);
      T.DA[c][R] = ta[0];
      T.DE[c][R] = te[0];
      // rows at and above the slab must not be touched by the update: drop the pivot block
      // and everything left of it from the operand
      slabA[tj * 64 + lane] = (col > 4 * R + 3) ? -ta[0] : 0.0;
    }
  }
  __syncthreads();
  // ---- rank-4 update of the tiles in tile rows TI..7 (the next pivot block's tile is early in
  // the list: its owner goes straight from these MFMAs into the next pivot chain)
  upd_range<S, W, -1, 0, upd_count(S, W, -1)>(T, slabA, lane);
}

template <int S, int W>
struct SlabLoop {
  static __device__ __forceinline__ void run(TileRegs &T, double *mop, double *slabA, int32_t *info,
                                             int32_t k0, int nb, int lane) {
    if (4 * S >= nb) return;  // identity padding: nothing left to eliminate (uniform)
    slab_step<S, W>(T, mop, slabA, info, k0, lane);
    SlabLoop<S + 1, W>::run(T, mop, slabA, info, k0, nb, lane);
  }
};
template <int W>
struct SlabLoop<32, W> {
  static __device__ __forceinline__ void run(TileRegs &, double *, double *, int32_t *, int32_t, int, int) {}
};

// One wave's whole job with its tile columns known at compile time: every ownership test
// folds away, the sweep is straight-line code (the four variants execute the same sequence
// of workgroup barriers).
// ---- [A | I] in accumulator layout: load / store of one wave's tile columns ----------------------
template <int W>
__device__ __forceinline__ void load_tiles(TileRegs &T, const double *A, int64_t lda, int nb, int lane) {
  constexpr int wave = W;
  const int col = lane & 15, q = lane >> 4;
  // outside the nb x nb block: identity
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int tj = wave + 4 * c;
    const int gc = 16 * tj + col;
    const int gcc = gc < nb ? gc : nb - 1;
#pragma unroll
    for (int ti = 0; ti < 8; ++ti) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gr = 16 * ti + 4 * r + q;
        const int grc = gr < nb ? gr : nb - 1;
        double a = 0.0;
        if (ti <= tj) a = A[(int64_t)grc * lda + gcc];
        a = (gr < nb && gc < nb) ? a : (gr == gc ? 1.0 : 0.0);
        if (ti < tj) T.X[c][ti][r] = a;
        else if (ti > tj) T.X[c][ti][r] = 0.0;
        else {
          T.DA[c][r] = a;
          T.DE[c][r] = (4 * r + q == col) ? 1.0 : 0.0;
        }
      }
    }
  }
}

// U (upper triangle of the block) and inv(U) = (U^-T)^T (whole 128 x 128 tile)
template <int W>
__device__ __forceinline__ void store_tiles(const TileRegs &T, double *A, int64_t lda, int nb, double *uinv, int lane) {
  constexpr int wave = W;
  const int col = lane & 15, q = lane >> 4;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int tj = wave + 4 * c;
    const int gc = 16 * tj + col;
#pragma unroll
    for (int ti = 0; ti < 8; ++ti) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gr = 16 * ti + 4 * r + q;
        if (ti <= tj) {
          const double u = (ti < tj) ? T.X[c][ti][r] : T.DA[c][r];
          if (gr < nb && gc < nb && gc >= gr) A[(int64_t)gr * lda + gc] = u;
        }
        // E(gr, gc) = W[gr][gc], lower triangular  ->  uinv[gc][gr]; the rest of row gc is zero
        double w = 0.0;
        if (ti > tj) w = T.X[c][ti][r];
        else if (ti == tj) w = T.DE[c][r];
        w = (gr >= gc && gr < nb && gc < nb) ? w : 0.0;
        if (gc < nb) uinv[gc * NB + gr] = w;
      }
    }
  }
}

// One wave's whole job with its tile columns known at compile time: every ownership test
// folds away, the sweep is straight-line code (the four variants execute the same sequence
// of workgroup barriers).
template <int W>
__device__ __forceinline__ void potf2_wave(double *A, int64_t lda, int nb, double *uinv, int32_t *info,
                                           int32_t k0, double *mop, double *slabA, int lane) {
  TileRegs T;
  load_tiles<W>(T, A, lda, nb, lane);
  SlabLoop<0, W>::run(T, mop, slabA, info, k0, nb, lane);
  store_tiles<W>(T, A, lda, nb, uinv, lane);
}

// ================================================================================================
// Second formulation (default): 16-row slabs.
//
// The 4-row sweep above costs two workgroup barriers and one dependent scale -> update MFMA pair per
// FOUR rows in every tile.  Here the owner of the diagonal tile factors the whole 16 x 16 leaf
// [D | I] -> [U | W = U^-T] by itself (four rounds of: pivot chain in registers, 2 scaling MFMAs, 2
// update MFMAs -- no LDS, no barrier), publishes U and W, and the tile waves work in 16-row slabs:
//   scale    tile <- W x tile: 4 MFMAs with K = 16 (register j of a tile in accumulator layout IS
//            the k-chunk j operand), independent of each other across tiles;
//   update   tile(tr, tj) -= S(TI, tr)^T S(TI, tj): 4 MFMAs per tile, A operands = the negated
//            scaled slab tiles from LDS;
// the same MFMA count as 32 rank-4 steps, a quarter of the barriers, and the final store of inv(U)
// goes out in whole 128-byte segments.  Measured (s_memtime, 128 x 128 block): 79 k -> 62 k cycles,
// of which 33 k are the eight leaves -- a chain of 128 dependent pivots at ~260 cycles each (fp64
// MFMA and fp64 VALU share the DP datapath: a fifth wave that does nothing but pivots, with
// look-ahead, was built and measured and bought nothing -- whatever shares its SIMD slows down by
// what the leaf gains).
constexpr int LW = 17;  // LDS leading dimension of the 16 x 16 leaf buffers

struct LeafShared {
  double Ul[16 * LW];         // leaf factor U (upper), row-major
  double Wl[16 * LW];         // leaf W = U^-T (lower), row-major
  double slabA[8 * 4 * 64];   // negated scaled slab tiles (TI, tr), register j, lane: A operands of the update
  double stage[4][16 * LW];   // per wave: transposition buffer of the final store
#ifdef LSQAMD_POTF2_TIMING
  long long stamps[64];
#endif
};

#ifdef LSQAMD_POTF2_TIMING
// stamps go to LDS and are dumped at the end: a global store in front of a barrier would make the
// barrier's s_waitcnt vmcnt(0) wait a microsecond for it
#define V3STAMP(i) do { if (dbg && (lane == 0)) sh.stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define V3STAMP(i) do { } while (0)
#endif

// 4 x 4 pivot block as in pivot4 above, tuned for the leaf: the operand stays in a register, the
// positivity test is off the dependent chain (the pivot is clamped with one v_max_f64; a
// non-positive one is recorded and makes the results meaningless, as the caller knows from info).
template <int R>
__device__ __forceinline__ double pivot4_reg(double reg, int &bad, int lane) {
  const double d00 = readlane_d(reg, 0 * 16 + 4 * R + 0), d01 = readlane_d(reg, 0 * 16 + 4 * R + 1);
  const double d02 = readlane_d(reg, 0 * 16 + 4 * R + 2), d03 = readlane_d(reg, 0 * 16 + 4 * R + 3);
  const double d11 = readlane_d(reg, 1 * 16 + 4 * R + 1), d12 = readlane_d(reg, 1 * 16 + 4 * R + 2);
  const double d13 = readlane_d(reg, 1 * 16 + 4 * R + 3), d22 = readlane_d(reg, 2 * 16 + 4 * R + 2);
  const double d23 = readlane_d(reg, 2 * 16 + 4 * R + 3), d33 = readlane_d(reg, 3 * 16 + 4 * R + 3);
  constexpr double TINY = 1.0e-300;
  const double y0 = rsqrt_cubic(__builtin_fmax(d00, TINY));
  const double u01 = d01 * y0, u02 = d02 * y0, u03 = d03 * y0;
  const double t11 = __builtin_fma(-u01, u01, d11);
  const double y1 = rsqrt_cubic(__builtin_fmax(t11, TINY));
  const double u12 = __builtin_fma(-u01, u02, d12) * y1, u13 = __builtin_fma(-u01, u03, d13) * y1;
  const double t22 = __builtin_fma(-u12, u12, __builtin_fma(-u02, u02, d22));
  const double y2 = rsqrt_cubic(__builtin_fmax(t22, TINY));
  const double u23 = __builtin_fma(-u12, u13, __builtin_fma(-u02, u03, d23)) * y2;
  const double t33 = __builtin_fma(-u23, u23, __builtin_fma(-u13, u13, __builtin_fma(-u03, u03, d33)));
  const double y3 = rsqrt_cubic(__builtin_fmax(t33, TINY));
  const bool ok0 = (d00 > 0.0) && (d00 < 1.0e300), ok1 = (t11 > 0.0) && (t11 < 1.0e300);
  const bool ok2 = (t22 > 0.0) && (t22 < 1.0e300), ok3 = (t33 > 0.0) && (t33 < 1.0e300);
  const int b = !ok0 ? 0 : (!ok1 ? 1 : (!ok2 ? 2 : (!ok3 ? 3 : -1)));
  bad = (bad < 0 && b >= 0) ? 4 * R + b : bad;
  const double v01 = -y0 * (u01 * y1);
  const double v12 = -y1 * (u12 * y2);
  const double v23 = -y2 * (u23 * y3);
  const double v02 = -y0 * __builtin_fma(u01, v12, u02 * y2);
  const double v13 = -y1 * __builtin_fma(u12, v23, u13 * y3);
  const double v03 = -y0 * __builtin_fma(u01, v13, __builtin_fma(u02, v23, u03 * y3));
  const int col = lane & 15, q = lane >> 4;
#if LSQAMD_LEAF_FMA_LAYOUT
  // The operand's ten entries laid out with ten FMAs against 0 / 1 lane masks (loop invariants: the compiler keeps them in
  // registers) instead of two levels of selects -- 10 instructions for 28 (each select is two v_cndmask_b32 + a compare),
  // and the last value off the chain, v03, is one FMA from the MFMA instead of three selects.
  auto at = [&](int qq, int cc) { return (q == qq && col == cc) ? 1.0 : 0.0; };
  double m = y0 * at(0, 0);
  m = __builtin_fma(y1, at(1, 1), m);
  m = __builtin_fma(v01, at(0, 1), m);
  m = __builtin_fma(y2, at(2, 2), m);
  m = __builtin_fma(v12, at(1, 2), m);
  m = __builtin_fma(v02, at(0, 2), m);
  m = __builtin_fma(y3, at(3, 3), m);
  m = __builtin_fma(v23, at(2, 3), m);
  m = __builtin_fma(v13, at(1, 3), m);
  return __builtin_fma(v03, at(0, 3), m);
#else
  const double r0 = col == 0 ? y0 : (col == 1 ? v01 : (col == 2 ? v02 : v03));
  const double r1 = col == 1 ? y1 : (col == 2 ? v12 : v13);
  const double r2 = col == 2 ? y2 : v23;
  const double m = q == 0 ? r0 : (q == 1 ? r1 : (q == 2 ? r2 : y3));
  return (col < 4 && q <= col) ? m : 0.0;
#endif
}

template <int R>
__device__ __forceinline__ void leaf_slab(v4d &DA, v4d &DE, int &bad, int lane) {
  const int col = lane & 15;
  const v4d zero = {0.0, 0.0, 0.0, 0.0};
  const double mo = pivot4_reg<R>(DA[R], bad, lane);
  // (round 4 measured the chain's two MFMAs issued before the identity half's: inside the noise -- the compiler hoists them)
  const v4d ta = mfma4(mo, DA[R], zero);
  const v4d te = mfma4(mo, DE[R], zero);
  DA[R] = ta[0];
  DE[R] = te[0];
  if constexpr (R < 3) {
    const double a = (col > 4 * R + 3) ? -ta[0] : 0.0;   // rows at and above the slab stay untouched
    DA = mfma4(a, DA[R], DA);
    DE = mfma4(a, DE[R], DE);
  }
}

// 16 x 16 leaf [D | I] -> [U | U^-T] in place in the owner's registers (accumulator layout), then
// published row-major for the other waves
template <class SH>
__device__ __forceinline__ void leaf16(v4d &DA, v4d &DE, SH &sh, int32_t *info, int32_t row0, int lane) {
  const int col = lane & 15, q = lane >> 4;
  int bad = -1;
  leaf_slab<0>(DA, DE, bad, lane);
  leaf_slab<1>(DA, DE, bad, lane);
  leaf_slab<2>(DA, DE, bad, lane);
  leaf_slab<3>(DA, DE, bad, lane);
  if (bad >= 0 && lane == 0) atomicCAS(info, 0, row0 + bad + 1);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    DA[r] = (col >= 4 * r + q) ? DA[r] : 0.0;
    DE[r] = (col <= 4 * r + q) ? DE[r] : 0.0;
    sh.Ul[(4 * r + q) * LW + col] = DA[r];
    sh.Wl[(4 * r + q) * LW + col] = DE[r];
  }
}

__device__ __forceinline__ v4d mfma4x4(const double (&a)[4], v4d b, v4d c) {
  c = mfma4(a[0], b[0], c);
  c = mfma4(a[1], b[1], c);
  c = mfma4(a[2], b[2], c);
  return mfma4(a[3], b[3], c);
}

// rank-16 update of ONE tile (TR, tj = W + 4 C) with slab TI; no-op when the tile is not live
template <int TI, int W, int TR, int C>
__device__ __forceinline__ void upd16_tile(TileRegs &T, const LeafShared &sh, int lane) {
  constexpr int tj = W + 4 * C;
  constexpr bool live = (TR < tj) || (TR == tj) || (tj <= TI);   // A part, diagonal slot, E part
  if constexpr (live) {
    double a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = sh.slabA[(TR * 4 + j) * 64 + lane];
    if constexpr (TR < tj) T.X[C][TR] = mfma4x4(a, T.X[C][TI], T.X[C][TR]);
    else if constexpr (TR == tj) T.DA[C] = mfma4x4(a, T.X[C][TI], T.DA[C]);
    else if constexpr (tj < TI) T.X[C][TR] = mfma4x4(a, T.X[C][TI], T.X[C][TR]);
    else T.X[C][TR] = mfma4x4(a, T.DE[C], T.X[C][TR]);           // tj == TI: the slab piece is W itself
  }
}

template <int TI, int W, int TR, int C, int SKIP_TR, int SKIP_C>
__device__ __forceinline__ void upd16_rest(TileRegs &T, const LeafShared &sh, int lane) {
  if constexpr (TR <= 7) {
    if constexpr (!(TR == SKIP_TR && C == SKIP_C)) upd16_tile<TI, W, TR, C>(T, sh, lane);
    if constexpr (C == 0) upd16_rest<TI, W, TR, 1, SKIP_TR, SKIP_C>(T, sh, lane);
    else upd16_rest<TI, W, TR + 1, 0, SKIP_TR, SKIP_C>(T, sh, lane);
  }
}

template <int TI, int W>
__device__ __forceinline__ void slab16_step(TileRegs &T, LeafShared &sh, int32_t *info, int32_t k0, int lane,
                                            long long *dbg) {
  const int col = lane & 15, q = lane >> 4;
  if constexpr (TI % 4 == W) {
    V3STAMP(3 * TI);
    leaf16(T.DA[TI / 4], T.DE[TI / 4], sh, info, k0 + 16 * TI, lane);
    V3STAMP(3 * TI + 1);
  }
  __syncthreads();   // B1: U, W of leaf TI are in LDS; every wave is done reading slab TI - 1
  double wop[4];     // A operand of the scaling: W[m][k], m = lane & 15, k = 4 j + (lane >> 4)
#pragma unroll
  for (int j = 0; j < 4; ++j) wop[j] = sh.Wl[col * LW + 4 * j + q];
  const v4d zero = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int tj = W + 4 * c;
    if (tj != TI) {
      const v4d t = mfma4x4(wop, T.X[c][TI], zero);
      T.X[c][TI] = t;
      if (tj > TI) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sh.slabA[(tj * 4 + j) * 64 + lane] = -t[j];
      }
    }
  }
  if constexpr (W == 0) V3STAMP(32 + 3 * TI);
  __syncthreads();   // B2: the scaled slab row is published
  if constexpr (W == 0) V3STAMP(33 + 3 * TI);
  if constexpr (TI < 7) {
    // the next diagonal tile first: its owner goes from these MFMAs straight into the next leaf
    constexpr int NT = TI + 1, NW = NT % 4, NC = NT / 4;
    if constexpr (NW == W) {
      upd16_tile<TI, W, NT, NC>(T, sh, lane);
      upd16_rest<TI, W, TI + 1, 0, NT, NC>(T, sh, lane);
    } else {
      upd16_rest<TI, W, TI + 1, 0, -1, -1>(T, sh, lane);
    }
  }
  if constexpr (W == 0) V3STAMP(34 + 3 * TI);
}

template <int TI, int W>
struct Slab16Loop {
  static __device__ __forceinline__ void run(TileRegs &T, LeafShared &sh, int32_t *info, int32_t k0, int nb, int lane,
                                             long long *dbg) {
    if (16 * TI >= nb) return;  // identity padding: nothing left to eliminate (uniform)
    slab16_step<TI, W>(T, sh, info, k0, lane, dbg);
    Slab16Loop<TI + 1, W>::run(T, sh, info, k0, nb, lane, dbg);
  }
};
template <int W>
struct Slab16Loop<8, W> {
  static __device__ __forceinline__ void run(TileRegs &, LeafShared &, int32_t *, int32_t, int, int, long long *) {}
};

// Final store of the 16-slab kernel.  U as in store_tiles; inv(U)[gc][gr] = E(gr, gc) goes through a
// per-wave LDS tile so that every store instruction writes whole 128-byte row segments (lane ->
// row gc = 16 tj + lane / 4, four consecutive gr) instead of 64 scattered doubles a kilobyte apart:
// the scattered form cost 16 k cycles, a fifth of the kernel.
template <int W>
__device__ __forceinline__ void store_tiles_v3(const TileRegs &T, double *A, int64_t lda, int nb, double *uinv,
                                               double *stage, int lane) {
  const int col = lane & 15, q = lane >> 4;
  const int jj = lane >> 2, i4 = 4 * (lane & 3);
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int tj = W + 4 * c;
    const int gc = 16 * tj + col;
#pragma unroll
    for (int ti = 0; ti < 8; ++ti) {
      if (ti <= tj) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gr = 16 * ti + 4 * r + q;
          const double u = (ti < tj) ? T.X[c][ti][r] : T.DA[c][r];
          if (gr < nb && gc < nb && gc >= gr) A[(int64_t)gr * lda + gc] = u;
        }
      }
      v4d w4 = {0.0, 0.0, 0.0, 0.0};
      if (ti >= tj) {
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[(4 * r + q) * LW + col] = (ti > tj) ? T.X[c][ti][r] : T.DE[c][r];
#pragma unroll
        for (int t = 0; t < 4; ++t) w4[t] = stage[(i4 + t) * LW + jj];
      }
      const int gcw = 16 * tj + jj;                    // uinv row this lane writes
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int gr = 16 * ti + i4 + t;
        w4[t] = (gr >= gcw && gr < nb && gcw < nb) ? w4[t] : 0.0;
      }
      // every lane stores (rows >= nb of the 128 x 128 inverse block receive zeros nobody reads): a
      // store under `if (gcw < nb)` left lanes 16..63 of the NEXT tile's LDS staging writes masked off
      // for partial blocks (nb = 16 k + 4 in the second tile column of waves 0..2)
      *reinterpret_cast<v4d *>(uinv + gcw * NB + 16 * ti + i4) = w4;
    }
  }
}

template <int W>
__device__ __forceinline__ void potf2v3_wave(double *A, int64_t lda, int nb, double *uinv, int32_t *info, int32_t k0,
                                             LeafShared &sh, int lane, long long *dbg) {
  TileRegs T;
  if constexpr (W == 0) V3STAMP(60);
  load_tiles<W>(T, A, lda, nb, lane);
  if constexpr (W == 0) V3STAMP(61);
  Slab16Loop<0, W>::run(T, sh, info, k0, nb, lane, dbg);
  if constexpr (W == 0) V3STAMP(62);
  store_tiles_v3<W>(T, A, lda, nb, uinv, sh.stage[W], lane);
  if constexpr (W == 0) V3STAMP(63);
}

// the whole job of a 4-wave workgroup (all 256 threads call it)
__device__ __forceinline__ void potf2v3_run(double *A, int64_t lda, int nb, double *uinv, int32_t *info, int32_t k0,
                                            LeafShared &sh, int tid, long long *dbg = nullptr) {
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave == 0) potf2v3_wave<0>(A, lda, nb, uinv, info, k0, sh, lane, dbg);
  else if (wave == 1) potf2v3_wave<1>(A, lda, nb, uinv, info, k0, sh, lane, dbg);
  else if (wave == 2) potf2v3_wave<2>(A, lda, nb, uinv, info, k0, sh, lane, dbg);
  else potf2v3_wave<3>(A, lda, nb, uinv, info, k0, sh, lane, dbg);
}

__global__ __launch_bounds__(256) void potf2_v3_kernel(double *A, int64_t lda, int nb, double *uinv, int32_t *info,
                                                       int32_t k0, int64_t strideA, int64_t strideW,
                                                       const int32_t *active, long long *dbg) {
  __shared__ LeafShared sh;
  if (active && !active[blockIdx.x]) return;
  A += (int64_t)blockIdx.x * strideA;
  uinv += (int64_t)blockIdx.x * strideW;
  info += blockIdx.x;
  potf2v3_run(A, lda, nb, uinv, info, k0, sh, threadIdx.x, dbg);
#ifdef LSQAMD_POTF2_TIMING
  __syncthreads();
  if (dbg && threadIdx.x < 64) dbg[threadIdx.x] = sh.stamps[threadIdx.x];
#endif
}

__global__ __launch_bounds__(256) void potf2_mfma_kernel(double *A, int64_t lda, int nb, double *uinv,
                                                         int32_t *info, int32_t k0, int64_t strideA,
                                                         int64_t strideW, const int32_t *active) {
  __shared__ double mop[64];
  __shared__ double slabA[8 * 64];
  if (active && !active[blockIdx.x]) return;
  A += (int64_t)blockIdx.x * strideA;
  uinv += (int64_t)blockIdx.x * strideW;
  info += blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  // wave-uniform by construction (readfirstlane tells the compiler): scalar dispatch
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave == 0) potf2_wave<0>(A, lda, nb, uinv, info, k0, mop, slabA, lane);
  else if (wave == 1) potf2_wave<1>(A, lda, nb, uinv, info, k0, mop, slabA, lane);
  else if (wave == 2) potf2_wave<2>(A, lda, nb, uinv, info, k0, mop, slabA, lane);
  else potf2_wave<3>(A, lda, nb, uinv, info, k0, mop, slabA, lane);
}

// ================================================================================================
// Third formulation: ONE pivot wave + THREE tile waves (16-row slabs as above).
//
// In the formulation above the wave that factors leaf TI + 1 also carries a quarter of the scaling and
// of the rank-16 update of slab TI: that share sits between two leaves of the pivot chain (stamps:
// leaf 5.1 k cycles, everything between two leaves 2.6 k, per slab).  Here wave 0 holds no tiles and
// does nothing but leaves; the eight tile columns are dealt to waves 1..3 (V4_COLS below).  The owner of column TI + 1 scales its tile (TI, TI + 1) first, applies it to the
// next diagonal tile at once (its own registers are the A operand) and hands that tile to the pivot
// wave through LDS with a flag -- 8 MFMAs and one LDS hop between two leaves; the other scaling and
// the whole update run on the other three SIMDs while the next leaf is being factored.  One
// workgroup barrier pair per slab as before (the pivot wave takes part: B1 = "U, W of leaf TI are
// published", B2 = "the scaled slab row is published").
// column tj takes part in tj (tj + 1) / 2 + (7 - tj)(8 - tj) / 2 tile updates over the sweep: 28 22 18 16 16 18 22 28;
// dealt so that the three waves carry 56 / 60 / 52 of the 168
constexpr int V4_COLS[3][3] = {{0, 7, -1}, {1, 3, 6}, {2, 4, 5}};
constexpr int v4_col(int U, int c) { return V4_COLS[U][c]; }
constexpr bool v4_has(int U, int c) { return V4_COLS[U][c] >= 0; }
constexpr int v4_owner(int tj) {
  for (int u = 0; u < 3; ++u)
    for (int c = 0; c < 3; ++c)
      if (V4_COLS[u][c] == tj) return u;
  return -1;
}
constexpr int v4_slot(int tj) {
  for (int u = 0; u < 3; ++u)
    for (int c = 0; c < 3; ++c)
      if (V4_COLS[u][c] == tj) return c;
  return -1;
}

struct TileRegs4 {
  v4d X[3][8];  // slot c <-> tile column V4_COLS[U][c] (as TileRegs)
  v4d DA[3];
  v4d DE[3];
};

struct Leaf4Shared {
  double Ul[2][16 * LW];      // by leaf parity: the pivot wave may be a leaf ahead of the slowest tile wave
  double Wl[2][16 * LW];
  double slabA[2][8 * 4 * 64];
  double stage[4][16 * LW];   // per wave: transposition buffer of the inverse's stores (0..2 tile waves, 3 pivot wave)
  double Dbuf[4 * 64];        // the next diagonal tile in accumulator layout (register r, lane)
  volatile int flagD[8];      // != 0: diagonal tile t is in Dbuf            (tile wave -> pivot wave)
  volatile int flagW[8];      // != 0: U, W of leaf t are in Ul / Wl[t & 1]  (pivot wave -> tile waves)
  int cnt[8];                 // tile waves that have published their part of scaled slab row t
#ifdef LSQAMD_POTF2_TIMING
  long long stamps[64];
#endif
};
struct LeafOut { double *Ul, *Wl; };   // what leaf16 writes to

#ifdef LSQAMD_POTF2_TIMING
#define V4STAMP(i) do { if (lane == 0) sh.stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define V4STAMP(i) do { } while (0)
#endif

template <int U>
__device__ __forceinline__ void load_tiles4(TileRegs4 &T, const double *A, int64_t lda, int nb, int lane) {
  const int col = lane & 15, q = lane >> 4;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (v4_has(U, c)) {
      const int tj = v4_col(U, c);
      const int gc = 16 * tj + col;
      const int gcc = gc < nb ? gc : nb - 1;
#pragma unroll
      for (int ti = 0; ti < 8; ++ti) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gr = 16 * ti + 4 * r + q;
          const int grc = gr < nb ? gr : nb - 1;
          double a = 0.0;
          if (ti <= tj) a = A[(int64_t)grc * lda + gcc];
          a = (gr < nb && gc < nb) ? a : (gr == gc ? 1.0 : 0.0);
          if (ti < tj) T.X[c][ti][r] = a;
          else if (ti == tj) T.DA[c][r] = a;
          // E-part tiles (ti > tj) come to life at slab tj (upd4_tile); DE with the leaf: with all 27
          // tiles of a wave live here the compiler serialises the loads behind register spills
        }
      }
    }
  }
}

// The inverse: E(gr, gc) = W[gr][gc], lower triangular -> uinv[gc][gr].  A 16 x 16 tile (ti, tj) of E is
// final the moment row slab ti is scaled and leaves right then, through the wave's LDS tile so that
// every store writes whole 128-byte row segments (as store_tiles_v3); the diagonal tiles are the pivot
// wave's.  The tiles above the diagonal of the inverse (ti < tj) are zeros, written once up front.
// U itself: every row slab is stored as it is scaled, the diagonal tiles by the pivot wave.  At most
// ~16 tiles of a wave are alive at any time, and nothing is left to store when the last leaf is done.
__device__ __forceinline__ void store_inv_tile(const v4d &t, int ti, int tj, int nb, double *uinv, double *stage,
                                               int lane) {
  const int col = lane & 15, q = lane >> 4;
  const int jj = lane >> 2, i4 = 4 * (lane & 3);
#pragma unroll
  for (int r = 0; r < 4; ++r) stage[(4 * r + q) * LW + col] = t[r];
  v4d w4;
#pragma unroll
  for (int k = 0; k < 4; ++k) w4[k] = stage[(i4 + k) * LW + jj];
  const int gcw = 16 * tj + jj;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int gr = 16 * ti + i4 + k;
    w4[k] = (gr >= gcw && gr < nb && gcw < nb) ? w4[k] : 0.0;
  }
  *reinterpret_cast<v4d *>(uinv + gcw * NB + 16 * ti + i4) = w4;   // unconditional: see store_tiles_v3
}

template <int U>
__device__ __forceinline__ void zero_upper_inv(double *uinv, int lane) {
  const int jj = lane >> 2, i4 = 4 * (lane & 3);
  const v4d zero = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (v4_has(U, c)) {
      const int tj = v4_col(U, c);
#pragma unroll
      for (int ti = 0; ti < 8; ++ti)
        if (ti < tj) *reinterpret_cast<v4d *>(uinv + (16 * tj + jj) * NB + 16 * ti + i4) = zero;
    }
  }
}

// rank-16 update of ONE tile (TR, tj = U + 3 C) with slab TI (cases as upd16_tile)
template <int TI, int U, int TR, int C>
__device__ __forceinline__ void upd4_tile(TileRegs4 &T, const Leaf4Shared &sh, int lane) {
  if constexpr (v4_has(U, C)) {
    constexpr int tj = v4_col(U, C);
    constexpr bool live = (TR < tj) || (TR == tj) || (tj <= TI);
    if constexpr (live) {
      double a[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = sh.slabA[TI & 1][(TR * 4 + j) * 64 + lane];
      if constexpr (TR < tj) T.X[C][TR] = mfma4x4(a, T.X[C][TI], T.X[C][TR]);
      else if constexpr (TR == tj) T.DA[C] = mfma4x4(a, T.X[C][TI], T.DA[C]);
      else if constexpr (tj < TI) T.X[C][TR] = mfma4x4(a, T.X[C][TI], T.X[C][TR]);
      else {                                                       // tj == TI: the E tile's first contribution
        const v4d zero = {0.0, 0.0, 0.0, 0.0};
        T.X[C][TR] = mfma4x4(a, T.DE[C], zero);
      }
    }
  }
}

template <int TI, int U, int TR, int C, int SKIP_TR, int SKIP_C>
__device__ __forceinline__ void upd4_rest(TileRegs4 &T, const Leaf4Shared &sh, int lane) {
  if constexpr (TR <= 7) {
    if constexpr (!(TR == SKIP_TR && C == SKIP_C)) upd4_tile<TI, U, TR, C>(T, sh, lane);
    if constexpr (C < 2) upd4_rest<TI, U, TR, C + 1, SKIP_TR, SKIP_C>(T, sh, lane);
    else upd4_rest<TI, U, TR + 1, 0, SKIP_TR, SKIP_C>(T, sh, lane);
  }
}

// hand a diagonal tile to the pivot wave
__device__ __forceinline__ void publish_diag(const v4d &D, Leaf4Shared &sh, int t, int lane) {
#pragma unroll
  for (int r = 0; r < 4; ++r) sh.Dbuf[r * 64 + lane] = D[r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) sh.flagD[t] = 1;
}

template <int TI, int U>
__device__ __forceinline__ void slab4_step(TileRegs4 &T, Leaf4Shared &sh, double *A, int64_t lda, int nb, double *uinv,
                                           int lane) {
  const int col = lane & 15, q = lane >> 4;
  auto store_row_tile = [&](const v4d &t, int tj) {   // rows 16 TI.. of U, tile column tj > TI: final
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gr = 16 * TI + 4 * r + q, gc = 16 * tj + col;
      if (gr < nb && gc < nb) A[(int64_t)gr * lda + gc] = t[r];
    }
  };
  while (sh.flagW[TI] == 0) __builtin_amdgcn_s_sleep(0);     // U, W of leaf TI are in LDS
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const double *Ul = sh.Ul[TI & 1], *Wl = sh.Wl[TI & 1];
  double *slab = sh.slabA[TI & 1];
  (void)Ul;
  if constexpr (v4_owner(TI) == U) {   // my column TI: the leaf's W is the diagonal tile of its E part
    constexpr int c = v4_slot(TI);
#pragma unroll
    for (int r = 0; r < 4; ++r) T.DE[c][r] = Wl[(4 * r + q) * LW + col];
  }
  double wop[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wop[j] = Wl[col * LW + 4 * j + q];
  const v4d zero = {0.0, 0.0, 0.0, 0.0};
  constexpr int NT = TI + 1;
  constexpr bool next_owner = NT < 8 && v4_owner(NT < 8 ? NT : 0) == U;
  constexpr int NC = v4_slot(NT < 8 ? NT : 0);
  if constexpr (next_owner) {
    // tile (TI, NT) first, then the next diagonal tile with it: the pivot wave is waiting for that one
    const v4d t = mfma4x4(wop, T.X[NC][TI], zero);
    T.X[NC][TI] = t;
    double a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = -t[j];
      slab[(NT * 4 + j) * 64 + lane] = a[j];
    }
    T.DA[NC] = mfma4x4(a, t, T.DA[NC]);
    publish_diag(T.DA[NC], sh, NT, lane);
    store_row_tile(t, NT);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (v4_has(U, c)) {
      const int tj = v4_col(U, c);
      if (tj != TI && !(next_owner && c == NC)) {
        const v4d t = mfma4x4(wop, T.X[c][TI], zero);
        T.X[c][TI] = t;
        if (tj > TI) {
#pragma unroll
          for (int j = 0; j < 4; ++j) slab[(tj * 4 + j) * 64 + lane] = -t[j];
          store_row_tile(t, tj);
        } else {
          store_inv_tile(t, TI, tj, nb, uinv, sh.stage[U], lane);   // E(TI, tj): final
        }
      }
    }
  }
  // the scaled slab row is complete when all three tile waves have published their part (the pivot
  // wave is not in this: it is already factoring the next leaf)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_fetch_add(&sh.cnt[TI], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(&sh.cnt[TI], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 3) __builtin_amdgcn_s_sleep(0);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if constexpr (TI < 7) {
    if constexpr (next_owner) upd4_rest<TI, U, TI + 1, 0, NT, NC>(T, sh, lane);
    else upd4_rest<TI, U, TI + 1, 0, -1, -1>(T, sh, lane);
  }
  if constexpr (U == 1) V4STAMP(32 + TI);     // tile wave 2: done with slab TI
}

template <int TI, int U>
struct Slab4Loop {
  static __device__ __forceinline__ void run(TileRegs4 &T, Leaf4Shared &sh, double *A, int64_t lda, int nb, double *uinv,
                                             int lane) {
    if (16 * TI >= nb) return;  // identity padding: nothing left to eliminate (uniform; the pivot wave stops too)
    slab4_step<TI, U>(T, sh, A, lda, nb, uinv, lane);
    Slab4Loop<TI + 1, U>::run(T, sh, A, lda, nb, uinv, lane);
  }
};
template <int U>
struct Slab4Loop<8, U> {
  static __device__ __forceinline__ void run(TileRegs4 &, Leaf4Shared &, double *, int64_t, int, double *, int) {}
};

// the pivot wave: leaves only
template <bool FIRST_FROM_MEMORY>
__device__ __forceinline__ void pivot_wave4(Leaf4Shared &sh, double *A, int64_t lda, double *uinv, int32_t *info,
                                            int32_t k0, int nb, int lane) {
  const int col = lane & 15, q = lane >> 4;
  for (int ti = 0; ti < 8; ++ti) {
    if (16 * ti >= nb) return;
    V4STAMP(4 * ti);
    v4d DA, DE;
    if (FIRST_FROM_MEMORY && ti == 0) {   // the first diagonal tile comes straight from memory (identity outside nb x nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gr = 4 * r + q;
        const double a = A[(int64_t)(gr < nb ? gr : nb - 1) * lda + (col < nb ? col : nb - 1)];
        DA[r] = (gr < nb && col < nb) ? a : (gr == col ? 1.0 : 0.0);
      }
    } else {
      while (sh.flagD[ti] == 0) __builtin_amdgcn_s_sleep(0);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
      for (int r = 0; r < 4; ++r) DA[r] = sh.Dbuf[r * 64 + lane];
    }
    V4STAMP(4 * ti + 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) DE[r] = (4 * r + q == col) ? 1.0 : 0.0;
    LeafOut out = {sh.Ul[ti & 1], sh.Wl[ti & 1]};
    leaf16(DA, DE, out, info, k0 + 16 * ti, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) sh.flagW[ti] = 1;
    V4STAMP(4 * ti + 2);
#pragma unroll
    for (int r = 0; r < 4; ++r) {            // the diagonal tile of U (leaf16 left it masked to its upper triangle)
      const int gr = 16 * ti + 4 * r + q, gc = 16 * ti + col;
      if (gr < nb && gc < nb && gc >= gr) A[(int64_t)gr * lda + gc] = DA[r];
    }
    store_inv_tile(DE, ti, ti, nb, uinv, sh.stage[3], lane);   // ... and of the inverse
  }
}

template <int U>
__device__ __forceinline__ void tile_wave4(double *A, int64_t lda, int nb, double *uinv, Leaf4Shared &sh, int lane) {
  TileRegs4 T;
  load_tiles4<U>(T, A, lda, nb, lane);
  zero_upper_inv<U>(uinv, lane);
  Slab4Loop<0, U>::run(T, sh, A, lda, nb, uinv, lane);
}

__device__ __forceinline__ void potf2v4_run(double *A, int64_t lda, int nb, double *uinv, int32_t *info, int32_t k0,
                                            Leaf4Shared &sh, int tid) {
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < 8) {
    sh.flagD[tid] = 0;
    sh.flagW[tid] = 0;
    sh.cnt[tid] = 0;
  }
  __syncthreads();
  if (wave == 0) pivot_wave4<true>(sh, A, lda, uinv, info, k0, nb, lane);
  else if (wave == 1) tile_wave4<0>(A, lda, nb, uinv, sh, lane);
  else if (wave == 2) tile_wave4<1>(A, lda, nb, uinv, sh, lane);
  else tile_wave4<2>(A, lda, nb, uinv, sh, lane);
}

__global__ __launch_bounds__(256) void potf2_v4_kernel(double *A, int64_t lda, int nb, double *uinv, int32_t *info,
                                                       int32_t k0, int64_t strideA, int64_t strideW,
                                                       const int32_t *active, long long *dbg) {
  __shared__ Leaf4Shared sh;
  if (active && !active[blockIdx.x]) return;
  A += (int64_t)blockIdx.x * strideA;
  uinv += (int64_t)blockIdx.x * strideW;
  info += blockIdx.x;
#ifdef LSQAMD_POTF2_TIMING
  if (threadIdx.x == 0) sh.stamps[60] = (long long)__builtin_readcyclecounter();
#endif
  potf2v4_run(A, lda, nb, uinv, info, k0, sh, threadIdx.x);
#ifdef LSQAMD_POTF2_TIMING
  __syncthreads();
  if (threadIdx.x == 0) sh.stamps[63] = (long long)__builtin_readcyclecounter();
  __syncthreads();
  if (dbg && threadIdx.x < 64) dbg[threadIdx.x] = sh.stamps[threadIdx.x];
#endif
}

// ---- trailing update of step k with the diagonal block of step k + 1 riding along -------------
// The diagonal kernel needs ONE tile of the trailing update (the next diagonal tile) and occupies
// one CU; the workgroup that computes that tile goes straight on to factor it while the other 255
// CUs finish the update: the pivot chain leaves the critical path and a kernel boundary per step
// goes with it.  (The same overlap through a second stream and events was measured and is
// slower than no overlap at all: cross-stream waits cost more than the diagonal kernel.)
// Tile workgroups: the direct-to-LDS 128 x 128 x 16 pipeline of gemm_tn_f64_interior_kernel,
// specialised to C -= P[:, m]^T P[:, n] with P the 128-row panel (K = 128), upper tiles only.
struct TrailArgs {
  const double *P;   // panel: row block k, starting at column k0 + nb   (128 x rest, lda)
  double *C;         // A[(k0 + nb).., (k0 + nb)..]
  int64_t lda;
  int32_t tiles_m, tiles_n;   // mrest / 128, rest / 128
  int32_t halves;             // tiles are done as two 64-column halves by two workgroups
};

constexpr int TBK = 16, TLD = 128 + 16, TSTAGE = 2 * TBK * TLD;
constexpr size_t TRAIL_LDS_BYTES = 2 * TSTAGE * sizeof(double);

// NJ = 4: the whole 128 x 128 tile; NJ = 2: its 64-column half `half` (same staging -- the panel rows come
// from L2 -- half the products: used while the update needs more than one round of workgroups, where
// the finer grain cuts the quantisation of the last round).
template <int NJ>
__device__ __forceinline__ void trail_tile(const TrailArgs &t, int tm, int tn, int half, double *smem, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)tm * 128, n0 = (int64_t)tn * 128;
  const int cw = (NJ == 4 ? 0 : 64 * half) + wn * 16 * NJ;      // this wave's first column inside the tile
  v4d acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const double *xp = t.P + (int64_t)wave * t.lda + m0 + 2 * lane;
  const double *yp = t.P + (int64_t)wave * t.lda + n0 + 2 * lane;
  const int64_t step4 = 4 * t.lda;
  auto stage = [&](int buf) {
    double *Xs = smem + buf * TSTAGE + wave * TLD;
    double *Ys = Xs + TBK * TLD;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((glb_void *)(xp + i * step4), (lds_void *)(Xs + 4 * i * TLD), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void *)(yp + i * step4), (lds_void *)(Ys + 4 * i * TLD), 16, 0, 0);
    }
    xp += 4 * step4;
    yp += 4 * step4;
  };
  const int fr = lane & 15, fq = lane >> 4;
  stage(0);
  __syncthreads();
  int cur = 0;
  for (int k0 = 0; k0 < 128; k0 += TBK) {
    if (k0 + TBK < 128) stage(cur ^ 1);
    const double *Xs = smem + cur * TSTAGE;
    const double *Ys = Xs + TBK * TLD;
#pragma unroll
    for (int kk = 0; kk < TBK / 4; ++kk) {
      const int kr = kk * 4 + fq;
      double a[4], bb[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = Xs[kr * TLD + wm * 64 + i * 16 + fr];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bb[j] = Ys[kr * TLD + cw + j * 16 + fr];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mfma4(a[i], bb[j], acc[i][j]);
    }
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double cv[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cv[j][r] = t.C[(m0 + wm * 64 + i * 16 + fq + 4 * r) * t.lda + n0 + cw + j * 16 + fr];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        t.C[(m0 + wm * 64 + i * 16 + fq + 4 * r) * t.lda + n0 + cw + j * 16 + fr] = cv[j][r] - acc[i][j][r];
  }
}

// The next diagonal block, updated where it is factored: workgroup 0 loads the block in the
// accumulator layout of the 16-slab sweep (load_tiles), DMAs the 128 panel rows over its 128 columns
// into LDS in one shot (all loads in flight at once), applies C -= P^T P to the tiles it holds --
// upper 16 x 16 tiles only, 32 k-chunks each, operands straight from LDS -- and goes on to the
// factorisation without the block ever returning to memory.  (Before: a full 128 x 128 x 128 tile
// product through eight dependent DMA stages, a store and a reload: 15.6 us of MFMA alone.)
// ---- the next diagonal block of the fused launch: pivot-wave formulation --------------------------
// Upper tiles per tile wave with V4_COLS: 9 / 13 / 14.  The pivot wave (idle until the first leaf) computes
// the contributions to four of wave 2's tiles ((0..3, 6)) and five of wave 3's ((0..4, 5)): 9 MFMAs per
// k-chunk on every wave; 18 KiB of LDS behind the panel tile (136-double rows here: 139 + 18 KiB).
constexpr int DLD4 = 128 + 8;
constexpr int DIAG4_HELP_OFF = 128 * DLD4;
constexpr size_t DIAG4_LDS_BYTES = (size_t)(128 * DLD4 + 9 * 4 * 64) * sizeof(double);
constexpr int v4_help_skip(int U) { return U == 1 ? 4 : (U == 2 ? 5 : 0); }   // leading tiles of slot 2 done by the pivot wave

template <int U>
__device__ __forceinline__ void diag_update_from_panel4(TileRegs4 &T, const double *Ps, int lane) {
  const int col = lane & 15, q = lane >> 4;
#pragma unroll 4
  for (int kc = 0; kc < 32; ++kc) {
    const double *row = Ps + (4 * kc + q) * DLD4 + col;
    double a[8];
#pragma unroll
    for (int ti = 0; ti < 8; ++ti) a[ti] = -row[16 * ti];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (v4_has(U, c)) {
        const int tj = v4_col(U, c);
        const double b = row[16 * tj];
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) {
          if (ti < tj && !(c == 2 && ti < v4_help_skip(U))) T.X[c][ti] = mfma4(a[ti], b, T.X[c][ti]);
          if (ti == tj) T.DA[c] = mfma4(a[ti], b, T.DA[c]);
        }
      }
    }
  }
}

__device__ __forceinline__ void diag_update_help4(double *Ps, int lane) {   // the pivot wave's share
  const int col = lane & 15, q = lane >> 4;
  const v4d zero = {0.0, 0.0, 0.0, 0.0};
  v4d h[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) h[i] = zero;
#pragma unroll 4
  for (int kc = 0; kc < 32; ++kc) {
    const double *row = Ps + (4 * kc + q) * DLD4 + col;
    double a[5];
#pragma unroll
    for (int ti = 0; ti < 5; ++ti) a[ti] = -row[16 * ti];
    const double b6 = row[16 * 6], b5 = row[16 * 5];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) h[ti] = mfma4(a[ti], b6, h[ti]);
#pragma unroll
    for (int ti = 0; ti < 5; ++ti) h[4 + ti] = mfma4(a[ti], b5, h[4 + ti]);
  }
  double *help = Ps + DIAG4_HELP_OFF;
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) help[(i * 4 + r) * 64 + lane] = h[i][r];
}

template <int U>
__device__ __forceinline__ void diag_update_collect4(TileRegs4 &T, const double *Ps, int lane) {
  const double *help = Ps + DIAG4_HELP_OFF;
  static_assert(V4_COLS[1][2] == 6 && V4_COLS[2][2] == 5, "the pivot wave's helper tiles are columns 6 and 5");
  if constexpr (U == 1) {
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) T.X[2][ti][r] += help[(ti * 4 + r) * 64 + lane];
  }
  if constexpr (U == 2) {
#pragma unroll
    for (int ti = 0; ti < 5; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) T.X[2][ti][r] += help[((4 + ti) * 4 + r) * 64 + lane];
  }
}

template <int WV>   // 0: pivot wave, 1..3: tile waves U = WV - 1
__device__ __forceinline__ void fused_diag_wave4(const TrailArgs &t, double *Adiag, int nb, double *uinv, int32_t *info,
                                                 int32_t k0n, double *smem, int tid) {
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int lane = tid & 63;
#pragma unroll 8
  for (int i = 0; i < 32; ++i) {
    const int k = WV * 32 + i;
    __builtin_amdgcn_global_load_lds((glb_void *)(t.P + (int64_t)k * t.lda + 2 * lane), (lds_void *)(smem + k * DLD4), 16, 0, 0);
  }
  TileRegs4 T;
  if constexpr (WV > 0) load_tiles4<WV - 1>(T, Adiag, t.lda, nb, lane);
  __syncthreads();
  if constexpr (WV > 0) diag_update_from_panel4<WV - 1>(T, smem, lane);
  else diag_update_help4(smem, lane);
  __syncthreads();   // the panel tile is dead: its LDS becomes the sweep's scratch
  Leaf4Shared &sh = *reinterpret_cast<Leaf4Shared *>(smem);
  if (tid < 8) {
    sh.flagD[tid] = 0;
    sh.flagW[tid] = 0;
    sh.cnt[tid] = 0;
  }
  if constexpr (WV > 0) diag_update_collect4<WV - 1>(T, smem, lane);
  __syncthreads();
  if constexpr (WV == 0) {
    pivot_wave4<false>(sh, Adiag, t.lda, uinv, info, k0n, nb, lane);
  } else {
    if constexpr (WV == 1) publish_diag(T.DA[0], sh, 0, lane);
    zero_upper_inv<WV - 1>(uinv, lane);
    Slab4Loop<0, WV - 1>::run(T, sh, Adiag, t.lda, nb, uinv, lane);
  }
}

__global__ __launch_bounds__(256) void trail_potf2_kernel(TrailArgs t, double *Adiag, int nb, double *uinv,
                                                          int32_t *info, int32_t k0n, int pivot_wave) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x;
  [[maybe_unused]] const int lane = tid & 63;
  // workgroup 0: the next diagonal block (update + factorisation); dispatched first, it works while
  // the other CUs do the remaining tiles of the update
  if (blockIdx.x > 0) {
    // upper tiles only, numbered row by row (tile (0, 0) belongs to workgroup 0).  A tiles_m x tiles_n
    // grid whose lower half exits at once looks the same but is not: workgroup b runs on XCD b % 8, so
    // with tiles_n a multiple of 8 a tile COLUMN stays on one XCD and the XCD of the longest columns
    // got 80 of the 496 tiles of the first step -- three rounds on its 32 CUs instead of two
    int rem = (int)blockIdx.x, half = 0, tm = 0;
    if (t.halves) {                     // two workgroups per tile: (tile, column half); workgroup 0 keeps tile (0, 0)
      half = (rem + 1) & 1;
      rem = (rem + 1) >> 1;
    }
    while (rem >= t.tiles_n - tm) {
      rem -= t.tiles_n - tm;
      ++tm;
    }
    if (t.halves) trail_tile<2>(t, tm, tm + rem, half, smem, tid);
    else trail_tile<4>(t, tm, tm + rem, 0, smem, tid);
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (pivot_wave) {
    if (wave == 0) fused_diag_wave4<0>(t, Adiag, nb, uinv, info, k0n, smem, tid);
    else if (wave == 1) fused_diag_wave4<1>(t, Adiag, nb, uinv, info, k0n, smem, tid);
    else if (wave == 2) fused_diag_wave4<2>(t, Adiag, nb, uinv, info, k0n, smem, tid);
    else fused_diag_wave4<3>(t, Adiag, nb, uinv, info, k0n, smem, tid);
    return;
  }
}

static_assert(sizeof(Leaf4Shared) <= (size_t)128 * DLD4 * sizeof(double), "the sweep's scratch lives in the dead panel tile");
constexpr size_t TRAIL_KERNEL_LDS = TRAIL_LDS_BYTES > DIAG4_LDS_BYTES ? TRAIL_LDS_BYTES : DIAG4_LDS_BYTES;
static PerDeviceOnce g_trail_attr;   // per device, thread-safe (common.h)

// trailing update of the step whose panel is P (128 x rest), fused with the diagonal block of the
// next step.  Requires mrest and rest multiples of 128 and 16-byte aligned rows.
// the fused launch exists for the pivot-wave formulation only (LSQAMD_POTF2=v3 / v2 / lds take the unfused path)
bool trail_potf2_available() {
  static const bool v4 = [] { const char *e = getenv("LSQAMD_POTF2"); return !e || (e[0] == 'v' && e[1] == '4'); }();
  return v4;
}

hipError_t launch_trail_potf2(hipStream_t st, const double *P, double *C, int64_t lda, int64_t mrest,
                              int64_t rest, int nb_next, double *uinv_next, int32_t *info, int32_t k0_next) {
  {
    const hipError_t ea = g_trail_attr.run([] {
      return hipFuncSetAttribute(reinterpret_cast<const void *>(trail_potf2_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)TRAIL_KERNEL_LDS);
    });
    if (ea != hipSuccess) return ea;
  }
  TrailArgs t;
  t.P = P; t.C = C; t.lda = lda;
  t.tiles_m = (int32_t)(mrest / 128);
  t.tiles_n = (int32_t)(rest / 128);
  static const int v4 = [] { const char *e = getenv("LSQAMD_POTF2"); return (!e || (e[0] == 'v' && e[1] == '4')) ? 1 : 0; }();
  const int64_t n_upper = (int64_t)t.tiles_m * t.tiles_n - (int64_t)t.tiles_m * (t.tiles_m - 1) / 2;
  // one workgroup per CU (workgroup 0's LDS): more tiles than CUs means a second round of workgroups,
  // mostly empty -- 64-column halves (twice the workgroups, half the work each) fill the rounds better
  static const int64_t half_from = [] { const char *e = getenv("LSQAMD_TRAIL_HALVES"); return e ? atoll(e) : (int64_t)256; }();
  t.halves = (n_upper > half_from) ? 1 : 0;
  const int64_t n_wg = t.halves ? 2 * (n_upper - 1) + 1 : n_upper;
  hipLaunchKernelGGL(trail_potf2_kernel, dim3((unsigned)n_wg), dim3(256), TRAIL_KERNEL_LDS,
                     st, t, C, nb_next, uinv_next, info, k0_next, v4);
  return hipGetLastError();
}

hipError_t launch_potf2_mfma(hipStream_t st, double *A, int64_t lda, int nb, double *uinv, int32_t *info,
                             int32_t k0, int32_t batch, int64_t strideA, int64_t strideW,
                             const int32_t *active) {
  static const bool v2 = [] { const char *e = getenv("LSQAMD_POTF2"); return e && e[0] == 'v' && e[1] == '2'; }();
  static const bool v4 = [] { const char *e = getenv("LSQAMD_POTF2"); return !e || (e[0] == 'v' && e[1] == '4'); }();   // the default
  if (v4)
    hipLaunchKernelGGL(potf2_v4_kernel, dim3((unsigned)batch), dim3(256), 0, st, A, lda, nb, uinv, info, k0, strideA,
                       strideW, active, g_potf2_dbg);
  else if (v2)
    hipLaunchKernelGGL(potf2_mfma_kernel, dim3((unsigned)batch), dim3(256), 0, st, A, lda, nb, uinv, info,
                       k0, strideA, strideW, active);
  else
    hipLaunchKernelGGL(potf2_v3_kernel, dim3((unsigned)batch), dim3(256), 0, st, A, lda, nb, uinv, info,
                       k0, strideA, strideW, active, g_potf2_dbg);
  return hipGetLastError();
}

}  // namespace lsqamd
