// Residual and Jacobian assembly kernels (gfx950).
//
// Replace the two callbacks GSL re-enters Python through on every evaluation
// (src/lsqfit/_gsl.pyx:727-740 `_c_f`, :742-760 `_c_df`) together with the body
// of `chiv.__call__` (src/lsqfit/_utilities.pyx:65-94): delta = f(x;p) - mean,
// times the 1x1 whitening weights.  The reference gets d f / d p by pushing
// gvar.valder derivative vectors through the user's Python function; here each
// row model carries its own forward-mode (dual-number) evaluation.
//
// Sum models (cosmix, multiexp): one wave64 per data row, lanes stride the K
// terms, so the two Jacobian half-rows are written as contiguous 512-byte
// segments per wave instruction; the parameter vector is staged once per
// workgroup in LDS and reused for all of the workgroup's rows.  Rows that belong
// to a correlated block are written unweighted into the raw buffer and whitened
// afterwards by the TN GEMM (X = W_b^T).
//
// Tape model -- the stand-in for the user's own fit function: an RPN program over (x_i, p).
//   single fits: REVERSE-mode AD, one lane per data row: one forward sweep stores the local partial
//     derivatives of every instruction (coalesced: [slot][lane]), one reverse sweep propagates the
//     adjoint through an LDS stack and adds d f / d p_j into a TRANSPOSED Jacobian (row j, lanes =
//     consecutive data rows: 512-byte segments); a tiled transpose then writes the weighted rows.
//     Cost O(tape length) per row whatever P is -- the forward-mode kernel below needs ceil(P / 16)
//     passes over the tape per row with a 16 x 16 dual stack in scratch;
//   batched fits and residual-only evaluations: the forward kernel (16 parameters per pass).
#include <cstdlib>

#include "common.h"
#include "devmath.h"
#include "jit.h"

namespace lsqamd {

__device__ __forceinline__ double wave_sum_all(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct ModelDev {
  int32_t model, n_x, n_tape;
  int64_t n_data, n_param;
  const double *x, *ymean, *wdiag, *p, *consts;
  const uint8_t *in_block;
  const int32_t *tape;
  double *out_w;    // whitened destination (residual vector, or J rows)
  double *out_raw;  // raw destination for rows inside correlated blocks
  int64_t ld;
  int32_t p_in_lds;
  int64_t p_stride, out_stride, ymean_stride;
  const int32_t *batch_active;
  const int32_t *trig_far;
};

// FAR (cosine model): false = no far-range trig code; both variants are launched and the device flag decides
// which one works (an empty launch costs ~2 us, the far-range code 10 % of the fused whitening kernel)
template <int MODEL, bool JAC, bool FAR = true>
__global__ __launch_bounds__(256) void sum_model_kernel(ModelDev m) {
  extern __shared__ __attribute__((aligned(16))) double sp[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t P = m.n_param, K = P / 2;
  if (m.trig_far && (m.trig_far[0] != 0) != FAR) return;
  if (m.batch_active && !m.batch_active[blockIdx.y]) return;
  m.p += (int64_t)blockIdx.y * m.p_stride;
  m.ymean += (int64_t)blockIdx.y * m.ymean_stride;
  m.out_w += (int64_t)blockIdx.y * m.out_stride;
  if (m.out_raw) m.out_raw += (int64_t)blockIdx.y * m.out_stride;
  const double *pp = m.p;
  if (m.p_in_lds) {
    for (int64_t i = tid; i < P; i += 256) sp[i] = m.p[i];
    __syncthreads();
    pp = sp;
  }
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < m.n_data; row += (int64_t)gridDim.x * 4) {
    const double x = m.x[row * m.n_x];
    const bool blk = m.in_block && m.in_block[row];
    const double w = blk ? 1.0 : m.wdiag[row];
    double *dst = blk ? m.out_raw : m.out_w;
    double f = 0.0;
    if (JAC && !(K & 1)) {
      // two terms per lane: the two half-rows of J go out as 16-byte stores (1 KiB per wave instruction)
      typedef double v2d __attribute__((ext_vector_type(2)));
      for (int64_t k = 2 * lane; k < K; k += 128) {
        double t[2], dq[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const double a = pp[k + u], q = pp[K + k + u];
          if (MODEL == LSQAMD_MODEL_COSMIX) {
            double s, c;
            sincos_moderate<FAR>(q * x, &s, &c);
            t[u] = c;
            dq[u] = -a * x * s;
          } else {
            const double e = exp(-q * x);
            t[u] = e;
            dq[u] = -a * x * e;
          }
          f += a * t[u];
        }
        *reinterpret_cast<v2d *>(dst + row * m.ld + k) = (v2d){w * t[0], w * t[1]};
        *reinterpret_cast<v2d *>(dst + row * m.ld + K + k) = (v2d){w * dq[0], w * dq[1]};
      }
    } else {
    for (int64_t k = lane; k < K; k += 64) {
      const double a = pp[k], q = pp[K + k];
      double term, dq;
      if (MODEL == LSQAMD_MODEL_COSMIX) {
        if (JAC) {
          double s, c;
          sincos_moderate<FAR>(q * x, &s, &c);
          term = c;
          dq = -a * x * s;
        } else {
          term = cos_moderate<FAR>(q * x);
          dq = 0.0;
        }
      } else {
        const double e = exp(-q * x);
        term = e;
        dq = -a * x * e;
      }
      f += a * term;
      if (JAC) {
        dst[row * m.ld + k] = w * term;
        dst[row * m.ld + K + k] = w * dq;
      }
    }
    }
    f = wave_sum_all(f);
    if (lane == 0) {
      const double delta = f - m.ymean[row];
      if (JAC)
        dst[row * m.ld + P] = w * delta;
      else
        dst[row] = w * delta;
    }
  }
}

template <bool JAC>
__global__ __launch_bounds__(256) void identity_model_kernel(ModelDev m) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= m.n_data) return;
  if (m.batch_active && !m.batch_active[blockIdx.y]) return;
  m.p += (int64_t)blockIdx.y * m.p_stride;
  m.ymean += (int64_t)blockIdx.y * m.ymean_stride;
  m.out_w += (int64_t)blockIdx.y * m.out_stride;
  if (m.out_raw) m.out_raw += (int64_t)blockIdx.y * m.out_stride;
  const bool blk = m.in_block && m.in_block[row];
  const double w = blk ? 1.0 : m.wdiag[row];
  double *dst = blk ? m.out_raw : m.out_w;
  const double delta = m.p[row] - m.ymean[row];
  if (JAC) {
    for (int64_t j = 0; j < m.n_param; ++j) dst[row * m.ld + j] = (j == row) ? w : 0.0;
    dst[row * m.ld + m.n_param] = w * delta;
  } else {
    dst[row] = w * delta;
  }
}

// RPN tape with forward-mode duals; one lane per row.  Derivatives are carried for
// LSQAMD_TAPE_CHUNK parameters at a time: a model with P parameters costs ceil(P / 16) passes over
// the tape per row (values are recomputed in every pass; the residual needs a single one).
template <bool JAC>
__global__ __launch_bounds__(64) void tape_model_kernel(ModelDev m) {
  constexpr int MP = LSQAMD_TAPE_CHUNK, MS = LSQAMD_TAPE_MAX_STACK;
  const int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (row >= m.n_data) return;
  if (m.batch_active && !m.batch_active[blockIdx.y]) return;
  m.p += (int64_t)blockIdx.y * m.p_stride;
  m.ymean += (int64_t)blockIdx.y * m.ymean_stride;
  m.out_w += (int64_t)blockIdx.y * m.out_stride;
  if (m.out_raw) m.out_raw += (int64_t)blockIdx.y * m.out_stride;
  const int P = (int)m.n_param;
  // wave-uniform addresses, never written while the kernel runs: scalar loads (see tape_segment_kernel)
  typedef const __attribute__((address_space(4))) int32_t *ConstI32;
  typedef const __attribute__((address_space(4))) double *ConstF64;
  const ConstI32 tape = (ConstI32)m.tape;
  const ConstF64 consts = (ConstF64)m.consts, par = (ConstF64)m.p;
  const bool blk = m.in_block && m.in_block[row];
  const double w = blk ? 1.0 : m.wdiag[row];
  double *dst = blk ? m.out_raw : m.out_w;
  double sv[MS];
  double sd[JAC ? MS : 1][JAC ? MP : 1];
  const int npass = JAC ? (P + MP - 1) / MP : 1;
  for (int pass = 0; pass < npass; ++pass) {
    const int c0 = pass * MP;                       // this pass differentiates p[c0 .. c0 + MP)
    const int nc = JAC ? (P - c0 < MP ? P - c0 : MP) : 0;
    int sp = 0;
    for (int t = 0; t < m.n_tape; ++t) {
      const int32_t ins = tape[t];
      const int op = ins & 0xff, arg = ins >> 8;
      switch (op) {
        case LSQAMD_OP_CONST:
          sv[sp] = consts[arg];
          if (JAC) for (int j = 0; j < MP; ++j) sd[sp][j] = 0.0;
          ++sp;
          break;
        case LSQAMD_OP_X:
          sv[sp] = m.x[row * m.n_x + arg];
          if (JAC) for (int j = 0; j < MP; ++j) sd[sp][j] = 0.0;
          ++sp;
          break;
        case LSQAMD_OP_P:
          sv[sp] = par[arg];
          if (JAC) for (int j = 0; j < MP; ++j) sd[sp][j] = (c0 + j == arg) ? 1.0 : 0.0;
          ++sp;
          break;
        case LSQAMD_OP_ADD:
        case LSQAMD_OP_SUB:
        case LSQAMD_OP_MUL:
        case LSQAMD_OP_DIV:
        case LSQAMD_OP_POW: {
          const double b = sv[sp - 1], a = sv[sp - 2];
          double v, da, db;  // d/da, d/db
          if (op == LSQAMD_OP_ADD) { v = a + b; da = 1.0; db = 1.0; }
          else if (op == LSQAMD_OP_SUB) { v = a - b; da = 1.0; db = -1.0; }
          else if (op == LSQAMD_OP_MUL) { v = a * b; da = b; db = a; }
          else if (op == LSQAMD_OP_DIV) { v = a / b; da = 1.0 / b; db = -v / b; }
          else {
            v = pow(a, b);
            da = b * pow(a, b - 1.0);
            db = (a > 0.0) ? v * log(a) : 0.0;
          }
          if (JAC) for (int j = 0; j < MP; ++j) sd[sp - 2][j] = da * sd[sp - 2][j] + db * sd[sp - 1][j];
          sv[sp - 2] = v;
          --sp;
          break;
        }
        default: {
          const double a = sv[sp - 1];
          double v, da;
          switch (op) {
            case LSQAMD_OP_NEG: v = -a; da = -1.0; break;
            case LSQAMD_OP_EXP: v = exp(a); da = v; break;
            case LSQAMD_OP_LOG: v = log(a); da = 1.0 / a; break;
            case LSQAMD_OP_SIN: v = sin(a); da = cos(a); break;
            case LSQAMD_OP_COS: v = cos(a); da = -sin(a); break;
            case LSQAMD_OP_ATAN: v = atan(a); da = 1.0 / (1.0 + a * a); break;
            case LSQAMD_OP_SQRT: v = sqrt(a); da = 0.5 / v; break;
            case LSQAMD_OP_POWI: {
              const int n = arg;  // sign-extended by the arithmetic shift
              v = pow(a, (double)n);
              da = (n == 0) ? 0.0 : n * pow(a, (double)(n - 1));
              break;
            }
            default: tape_unary_ext(op, a, v, da); break;
          }
          if (JAC) for (int j = 0; j < MP; ++j) sd[sp - 1][j] *= da;
          sv[sp - 1] = v;
          break;
        }
      }
    }
    if (JAC) {
      for (int j = 0; j < nc; ++j) dst[row * m.ld + c0 + j] = w * sd[0][j];
    }
  }
  const double delta = sv[0] - m.ymean[row];
  if (JAC) dst[row * m.ld + P] = w * delta;
  else dst[row] = w * delta;
}


// ---- reverse-mode tape Jacobian ------------------------------------------------------------------
// slots of local partial derivatives an instruction stores in the forward sweep
__host__ __device__ inline int tape_slots_of(int op) {
  if (op == LSQAMD_OP_MUL || op == LSQAMD_OP_DIV || op == LSQAMD_OP_POW) return 2;
  if (op >= LSQAMD_OP_EXP && op <= LSQAMD_OP_LAST) return 1;
  return 0;   // pushes, ADD, SUB, NEG: constants
}

struct TapeRev {
  ModelDev m;
  const int32_t *poff;   // first slot of instruction t
  double *part;          // [wave slot][n_slots][64]
  double *jt;            // [(P + 1)][ldn]: d f / d p_j per data row; row P = f - ymean
  int64_t ldn;
  int32_t n_slots;
  int64_t n_groups;      // ceil(n_data / 64)
  int32_t single;        // no parameter occurs twice on the tape: its column is stored, not accumulated
};

// JAC = false: the forward sweep alone (trial residuals): no partials stored, the weighted residual written
// straight to the vector the block whitening / sum of squares read.
template <bool JAC>
__global__ __launch_bounds__(256) void tape_reverse_kernel(TapeRev a) {
  __shared__ double stk[4][LSQAMD_TAPE_MAX_STACK][64];   // value stack, then adjoint stack
  const ModelDev &m = a.m;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: the interpreter must branch on scalars)
  // tape, slot offsets, constants and parameters: wave-uniform addresses, never written while this kernel
  // runs -- the constant address space makes them scalar loads (see tape_segment_kernel below)
  typedef const __attribute__((address_space(4))) int32_t *ConstI32;
  typedef const __attribute__((address_space(4))) double *ConstF64;
  const ConstI32 tcode = (ConstI32)m.tape, toff = (ConstI32)a.poff;
  const ConstF64 consts = (ConstF64)m.consts, par = (ConstF64)m.p;
  const int64_t slot = (int64_t)blockIdx.x * 4 + wave, nslots = (int64_t)gridDim.x * 4;
  double *part = a.part + slot * a.n_slots * 64 + lane;
  double (*S)[64] = stk[wave];
  // The top of the stack lives in a register (tos); the entry k below it (k < sp - 1) at S[k + 1]: a push is
  // one LDS store, a binary operation one LDS load, a function none.
  for (int64_t g = slot; g < a.n_groups; g += nslots) {
    const int64_t row = g * 64 + lane;
    const bool valid = row < m.n_data;
    const int64_t rr = valid ? row : m.n_data - 1;
    // ---- forward: values through the stack, local partials to the store
    int sp = 0;
    double tos = 0.0;
    int32_t ins_n = tcode[0], off_n = toff[0];
    for (int t = 0; t < m.n_tape; ++t) {
      const int32_t ins = ins_n;
      double *pd = part + (int64_t)off_n * 64;
      if (t + 1 < m.n_tape) { ins_n = tcode[t + 1]; off_n = toff[t + 1]; }   // decode one ahead
      const int op = ins & 0xff, arg = ins >> 8;
      if (op <= LSQAMD_OP_P) {
        S[sp++][lane] = tos;
        if (op == LSQAMD_OP_P) tos = par[arg];
        else if (op == LSQAMD_OP_X) tos = m.x[rr * m.n_x + arg];
        else tos = consts[arg];
      } else if (op <= LSQAMD_OP_POW) {
        const double b = tos, x = S[--sp][lane];
        if (op == LSQAMD_OP_MUL) { tos = x * b; if (JAC) { pd[0] = b; pd[64] = x; } }
        else if (op == LSQAMD_OP_ADD) tos = x + b;
        else if (op == LSQAMD_OP_SUB) tos = x - b;
        else if (op == LSQAMD_OP_DIV) { tos = x / b; if (JAC) { pd[0] = 1.0 / b; pd[64] = -tos / b; } }
        else { tos = pow(x, b); if (JAC) { pd[0] = b * pow(x, b - 1.0); pd[64] = (x > 0.0) ? tos * log(x) : 0.0; } }
      } else {
        const double x = tos;
        double v, d = 1.0;
        if constexpr (JAC) {
          switch (op) {
            case LSQAMD_OP_NEG: v = -x; break;
            case LSQAMD_OP_EXP: v = exp(x); d = v; break;
            case LSQAMD_OP_LOG: v = log(x); d = 1.0 / x; break;
            case LSQAMD_OP_SIN: { double sn, cs; sincos(x, &sn, &cs); v = sn; d = cs; break; }
            case LSQAMD_OP_COS: { double sn, cs; sincos(x, &sn, &cs); v = cs; d = -sn; break; }
            case LSQAMD_OP_ATAN: v = atan(x); d = 1.0 / (1.0 + x * x); break;
            case LSQAMD_OP_SQRT: v = sqrt(x); d = 0.5 / v; break;
            case LSQAMD_OP_POWI: v = pow(x, (double)arg); d = (arg == 0) ? 0.0 : arg * pow(x, (double)(arg - 1)); break;
            default: tape_unary_ext(op, x, v, d); break;
          }
          if (op != LSQAMD_OP_NEG) pd[0] = d;
        } else {   // the values the Jacobian sweep computes, from the same library calls (sincos(x).s == sin(x) bit for bit is not promised: keep sincos)
          switch (op) {
            case LSQAMD_OP_NEG: v = -x; break;
            case LSQAMD_OP_EXP: v = exp(x); break;
            case LSQAMD_OP_LOG: v = log(x); break;
            case LSQAMD_OP_SIN: { double sn, cs; sincos(x, &sn, &cs); v = sn; break; }
            case LSQAMD_OP_COS: { double sn, cs; sincos(x, &sn, &cs); v = cs; break; }
            case LSQAMD_OP_ATAN: v = atan(x); break;
            case LSQAMD_OP_SQRT: v = sqrt(x); break;
            case LSQAMD_OP_POWI: v = pow(x, (double)arg); break;
            default: tape_unary_ext(op, x, v, d); break;
          }
          (void)d;
        }
        tos = v;
      }
    }
    if constexpr (!JAC) {
      if (valid) {
        const bool blk = m.in_block && m.in_block[row];
        (blk ? m.out_raw : m.out_w)[row] = (blk ? 1.0 : m.wdiag[row]) * (tos - m.ymean[row]);
      }
      continue;
    }
    if (valid) a.jt[(int64_t)m.n_param * a.ldn + row] = tos - m.ymean[row];
    // ---- reverse: adjoints through the same stack; d f / d p_j accumulates in the transposed Jacobian
    tos = 1.0;
    sp = 1;
    ins_n = tcode[m.n_tape - 1];
    off_n = toff[m.n_tape - 1];
    for (int t = m.n_tape - 1; t >= 0; --t) {
      const int32_t ins = ins_n;
      const double *pd = part + (int64_t)off_n * 64;
      if (t > 0) { ins_n = tcode[t - 1]; off_n = toff[t - 1]; }
      const int op = ins & 0xff, arg = ins >> 8;
      if (op <= LSQAMD_OP_P) {
        const double gbar = tos;
        tos = S[--sp][lane];
        if (op == LSQAMD_OP_P && valid) {
          double *dst = a.jt + (int64_t)arg * a.ldn + row;
          if (a.single) *dst = gbar;   // the only read of this parameter
          else   // fire and forget: the parameter occurs more than once on the tape
            (void)__hip_atomic_fetch_add(dst, gbar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else if (op <= LSQAMD_OP_POW) {
        const double gbar = tos;
        double da = 1.0, db = 1.0;
        if (op == LSQAMD_OP_SUB) db = -1.0;
        else if (op != LSQAMD_OP_ADD) { da = pd[0]; db = pd[64]; }
        S[sp++][lane] = gbar * da;
        tos = gbar * db;
      } else {
        tos *= (op == LSQAMD_OP_NEG) ? -1.0 : pd[0];
      }
    }
  }
}

// ---- the same for tapes whose root is a SUM of independent sub-expressions ----------------------------
// f = +-S_1 +- S_2 ... +- S_m (lsqamd_set_tape finds the root-level ADDs / SUBs): the adjoint of every S_k is
// +-1, so each segment is differentiated right after it is evaluated -- forward through the LDS stack with the
// local partials in LDS as well, reverse at once -- and the segments are independent of each other: the
// launch spreads (row group, chunk of segments) pairs over the chip.  What the whole-tape kernel above
// suffers from at N = 65536 is occupancy, not arithmetic: 1024 row groups are 1024 waves, ONE per SIMD, each
// walking 2 x 3583 dependent LDS / memory round trips alone (2.3 ms).  With eight chunks there are eight
// waves per SIMD to cover each other; the per-wave LDS footprint is sized by the deepest / widest segment
// (a cosine term: 3 stack levels + 5 partials = 4 KB), the parameters sit in LDS, the tape is read through
// scalar loads one instruction ahead, x of the row through registers.  Per-chunk values of f go to a
// scratch array and are summed in chunk order (tape_total_kernel): reproducible bit for bit.
[[maybe_unused]] constexpr int TAPE_SEG_SLOTS = 32;     // partial-derivative slots a segment may use at most

struct TapeSeg {
  TapeRev r;
  const int32_t *seg;    // [n_seg][3]: first and last instruction of every segment and its sign (the joining ADD / SUB is implied)
  int32_t n_seg, n_chunks, seg_per_chunk;
  int32_t depth, slots;  // LDS rows per wave: value / adjoint stack, local partials
  double *ftot;          // [n_chunks][ldn] partial values of f
};

// R = data rows per lane (1 or 2): the interpreter's own work per instruction -- thirty scalar instructions
// of decode, dispatch and address arithmetic -- is paid once per wave whatever the lanes carry, so with two
// rows per lane (rows g*128 + lane and + 64) it is spread over twice the arithmetic.
template <bool JAC, int R>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void tape_segment_kernel(TapeSeg a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];   // [n_param] parameters, then per wave [depth + slots][R][64]
  constexpr int RS = R * 64;
  const ModelDev &m = a.r.m;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: the interpreter must branch on scalars)
  double *plds = lds;
  for (int64_t i = threadIdx.x; i < m.n_param; i += 256) plds[i] = m.p[i];
  __syncthreads();
  double *mine = lds + ((m.n_param + 1) & ~(int64_t)1) + (int64_t)wave * (a.depth + a.slots) * RS + lane;
  auto S = [&](int i, int r) -> double & { return mine[i * RS + r * 64]; };
  double *PL = mine + a.depth * RS;
  // The tape, its slot offsets, the segment table and the constants are read at wave-uniform addresses and are
  // never written on the device: through the constant address space they become scalar loads (s_load, scalar
  // cache) that the decode-ahead below can really hide -- as ordinary global pointers the compiler must
  // assume the stores to jt may alias them and issues a vector load per instruction (540 cycles each, measured).
  typedef const __attribute__((address_space(4))) int32_t *ConstI32;
  typedef const __attribute__((address_space(4))) double *ConstF64;
  const ConstI32 tape = (ConstI32)m.tape, poff = (ConstI32)a.r.poff, segs = (ConstI32)a.seg;
  const ConstF64 consts = (ConstF64)m.consts;
  const int64_t unit = (int64_t)blockIdx.x * 4 + wave;          // (row group, chunk), chunk fastest
  const int64_t g = unit / a.n_chunks;
  const int chunk = (int)(unit % a.n_chunks);
  if (g >= a.r.n_groups) return;
  int64_t row[R], rr[R];
  bool valid[R];
  double x0[R], x1[R], total[R], tos[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    row[r] = (g * R + r) * 64 + lane;
    valid[r] = row[r] < m.n_data;
    rr[r] = valid[r] ? row[r] : m.n_data - 1;
    x0[r] = m.x[rr[r] * m.n_x];
    x1[r] = m.n_x > 1 ? m.x[rr[r] * m.n_x + 1] : 0.0;
    total[r] = 0.0;
  }
  const int s_lo = chunk * a.seg_per_chunk, s_hi = s_lo + a.seg_per_chunk < a.n_seg ? s_lo + a.seg_per_chunk : a.n_seg;
  for (int sgi = s_lo; sgi < s_hi; ++sgi) {
    const int lo = segs[3 * sgi], hi = segs[3 * sgi + 1];
    const double sign = (double)segs[3 * sgi + 2];
    const int base = poff[lo];
    // ---- forward over the segment.  The top of the stack lives in a register (tos); the entry k below it
    // (k < sp - 1) at S(k + 1), so a push is one LDS store, a binary operation one LDS load, a function none.
    int sp = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) tos[r] = 0.0;
    int32_t ins_n = tape[lo], off_n = base;
    for (int t = lo; t <= hi; ++t) {
      const int32_t ins = ins_n;
      double *pd = PL + (off_n - base) * RS;
      if (t < hi) { ins_n = tape[t + 1]; off_n = poff[t + 1]; }      // decode one ahead (scalar loads)
      const int op = ins & 0xff, arg = ins >> 8;
      if (op <= LSQAMD_OP_P) {
#pragma unroll
        for (int r = 0; r < R; ++r) S(sp, r) = tos[r];
        ++sp;
        if (op == LSQAMD_OP_P) {
          const double v = plds[arg];
#pragma unroll
          for (int r = 0; r < R; ++r) tos[r] = v;
        } else if (op == LSQAMD_OP_X) {
#pragma unroll
          for (int r = 0; r < R; ++r) tos[r] = arg == 0 ? x0[r] : (arg == 1 ? x1[r] : m.x[rr[r] * m.n_x + arg]);
        } else {
          const double v = consts[arg];
#pragma unroll
          for (int r = 0; r < R; ++r) tos[r] = v;
        }
      } else if (op <= LSQAMD_OP_POW) {
        --sp;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const double b = tos[r], x = S(sp, r);
          double *q = pd + r * 64;
          if (op == LSQAMD_OP_MUL) { tos[r] = x * b; if (JAC) { q[0] = b; q[RS] = x; } }
          else if (op == LSQAMD_OP_ADD) tos[r] = x + b;
          else if (op == LSQAMD_OP_SUB) tos[r] = x - b;
          else if (op == LSQAMD_OP_DIV) { tos[r] = x / b; if (JAC) { q[0] = 1.0 / b; q[RS] = -tos[r] / b; } }
          else { tos[r] = pow(x, b); if (JAC) { q[0] = b * pow(x, b - 1.0); q[RS] = (x > 0.0) ? tos[r] * log(x) : 0.0; } }
        }
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const double x = tos[r];
          double v, d = 1.0;
          if constexpr (JAC) {
            switch (op) {
              case LSQAMD_OP_NEG: v = -x; break;
              case LSQAMD_OP_EXP: v = exp(x); d = v; break;
              case LSQAMD_OP_LOG: v = log(x); d = 1.0 / x; break;
              case LSQAMD_OP_SIN: { double sn, cs; sincos(x, &sn, &cs); v = sn; d = cs; break; }
              case LSQAMD_OP_COS: { double sn, cs; sincos(x, &sn, &cs); v = cs; d = -sn; break; }
              case LSQAMD_OP_ATAN: v = atan(x); d = 1.0 / (1.0 + x * x); break;
              case LSQAMD_OP_SQRT: v = sqrt(x); d = 0.5 / v; break;
              case LSQAMD_OP_POWI: v = pow(x, (double)arg); d = (arg == 0) ? 0.0 : arg * pow(x, (double)(arg - 1)); break;
              default: tape_unary_ext(op, x, v, d); break;
            }
            if (op != LSQAMD_OP_NEG) pd[r * 64] = d;
          } else {   // the values the Jacobian sweep computes, from the same library calls (sincos(x).s == sin(x) bit for bit is not promised: keep sincos)
            switch (op) {
              case LSQAMD_OP_NEG: v = -x; break;
              case LSQAMD_OP_EXP: v = exp(x); break;
              case LSQAMD_OP_LOG: v = log(x); break;
              case LSQAMD_OP_SIN: { double sn, cs; sincos(x, &sn, &cs); v = sn; break; }
              case LSQAMD_OP_COS: { double sn, cs; sincos(x, &sn, &cs); v = cs; break; }
              case LSQAMD_OP_ATAN: v = atan(x); break;
              case LSQAMD_OP_SQRT: v = sqrt(x); break;
              case LSQAMD_OP_POWI: v = pow(x, (double)arg); break;
              default: tape_unary_ext(op, x, v, d); break;
            }
            (void)d;
          }
          tos[r] = v;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) total[r] += sign * tos[r];
    if constexpr (!JAC) continue;
    // ---- reverse over the segment: its root carries the adjoint of the sum, +-1 (same stack convention)
#pragma unroll
    for (int r = 0; r < R; ++r) tos[r] = sign;
    sp = 1;
    ins_n = tape[hi];
    off_n = poff[hi];
    for (int t = hi; t >= lo; --t) {
      const int32_t ins = ins_n;
      const double *pd = PL + (off_n - base) * RS;
      if (t > lo) { ins_n = tape[t - 1]; off_n = poff[t - 1]; }
      const int op = ins & 0xff, arg = ins >> 8;
      if (op <= LSQAMD_OP_P) {
        --sp;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const double gbar = tos[r];
          tos[r] = S(sp, r);
          if (op == LSQAMD_OP_P && valid[r]) {
            double *dst = a.r.jt + (int64_t)arg * a.r.ldn + row[r];
            if (a.r.single) __builtin_nontemporal_store(gbar, dst);
            else   // fire and forget: the parameter occurs more than once on the tape
              (void)__hip_atomic_fetch_add(dst, gbar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      } else if (op <= LSQAMD_OP_POW) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const double gbar = tos[r];
          double da = 1.0, db = 1.0;
          if (op == LSQAMD_OP_SUB) db = -1.0;
          else if (op != LSQAMD_OP_ADD) { da = pd[r * 64]; db = pd[RS + r * 64]; }
          S(sp, r) = gbar * da;
          tos[r] = gbar * db;
        }
        ++sp;
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) tos[r] *= (op == LSQAMD_OP_NEG) ? -1.0 : pd[r * 64];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (valid[r]) a.ftot[(int64_t)chunk * a.r.ldn + row[r]] = total[r];
}

// row P of the transposed Jacobian: f - ymean, the chunks' pieces of f summed in chunk order; for a residual
// evaluation (out_w given) the weighted residual itself, rows inside covariance blocks unweighted to out_raw
__global__ __launch_bounds__(256) void tape_total_kernel(const double *ftot, int n_chunks, int64_t ldn, int64_t N,
                                                         const double *ymean, double *out, const double *wdiag,
                                                         const uint8_t *in_block, double *out_w, double *out_raw) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= N) return;
  double f = 0.0;
  for (int c = 0; c < n_chunks; ++c) f += ftot[(int64_t)c * ldn + row];
  if (out_w) {
    const bool blk = in_block && in_block[row];
    (blk ? out_raw : out_w)[row] = (blk ? 1.0 : wdiag[row]) * (f - ymean[row]);
  } else {
    out[row] = f - ymean[row];
  }
}

// dst[row][c] = w_row * jt[c][row] for c <= P: 64 x 64 tiles through LDS, rows inside covariance
// blocks go unweighted to the raw buffer
__global__ __launch_bounds__(256) void tape_finish_kernel(const double *jt, int64_t ldn, int64_t N, int64_t ncols,
                                                          const double *wdiag, const uint8_t *in_block,
                                                          double *out_w, double *out_raw, int64_t ld) {
  __shared__ double tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int64_t c = c0 + i, r = r0 + tx;
    tile[i][tx] = (c < ncols && r < N) ? jt[c * ldn + r] : 0.0;
  }
  __syncthreads();
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int64_t r = r0 + i, c = c0 + tx;
    if (r < N && c < ncols) {
      const bool blk = in_block && in_block[r];
      (blk ? out_raw : out_w)[r * ld + c] = (blk ? 1.0 : wdiag[r]) * tile[tx][i];
    }
  }
}

__global__ __launch_bounds__(256) void trig_range_kernel(const double *q, int64_t n, double xmax, int32_t *flag) {
  __shared__ int any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  bool far = false;
  for (int64_t k = threadIdx.x; k < n; k += 256) far |= !(fabs(q[k]) * xmax < 0.5 * TRIG_FAST_LIMIT);
  if (far) any = 1;
  __syncthreads();
  if (threadIdx.x == 0) flag[0] = any;
}

hipError_t launch_trig_range(hipStream_t st, const double *q, int64_t n, double xmax, int32_t *flag) {
  hipLaunchKernelGGL(trig_range_kernel, dim3(1), dim3(256), 0, st, q, n, xmax, flag);
  return hipGetLastError();
}

template <bool JAC>
static hipError_t launch_model(hipStream_t st, const ModelArgs &a, double *out_w, double *out_raw,
                               int64_t ld) {
  if (a.n_data <= 0) return hipSuccess;
  ModelDev m;
  m.model = a.model; m.n_x = a.n_x; m.n_tape = a.n_tape;
  m.n_data = a.n_data; m.n_param = a.n_param;
  m.x = a.x; m.ymean = a.ymean; m.wdiag = a.wdiag; m.p = a.p; m.consts = a.consts;
  m.in_block = a.in_block; m.tape = a.tape;
  m.out_w = out_w; m.out_raw = out_raw; m.ld = ld;
  m.p_in_lds = 0;
  m.p_stride = a.p_stride; m.out_stride = a.out_stride; m.batch_active = a.batch_active;
  m.ymean_stride = a.ymean_stride;
  m.trig_far = a.model == LSQAMD_MODEL_COSMIX ? a.trig_far : nullptr;
  const unsigned nb = (unsigned)(a.n_batch < 1 ? 1 : a.n_batch);
  switch (a.model) {
    case LSQAMD_MODEL_COSMIX:
    case LSQAMD_MODEL_MULTIEXP: {
      size_t lds = 0;
      if ((size_t)a.n_param * sizeof(double) <= 48 * 1024) {
        lds = (size_t)a.n_param * sizeof(double);
        m.p_in_lds = 1;
      }
      int64_t blocks = (a.n_data + 3) / 4;
      if (blocks > 4096) blocks = 4096;
      if (a.model == LSQAMD_MODEL_COSMIX) {
        if (m.trig_far)
          hipLaunchKernelGGL((sum_model_kernel<LSQAMD_MODEL_COSMIX, JAC, false>), dim3((unsigned)blocks, nb),
                             dim3(256), lds, st, m);
        hipLaunchKernelGGL((sum_model_kernel<LSQAMD_MODEL_COSMIX, JAC, true>), dim3((unsigned)blocks, nb),
                           dim3(256), lds, st, m);
      } else
        hipLaunchKernelGGL((sum_model_kernel<LSQAMD_MODEL_MULTIEXP, JAC>), dim3((unsigned)blocks, nb),
                           dim3(256), lds, st, m);
      break;
    }
    case LSQAMD_MODEL_IDENTITY:
      hipLaunchKernelGGL((identity_model_kernel<JAC>), dim3((unsigned)((a.n_data + 255) / 256), nb),
                         dim3(256), 0, st, m);
      break;
    case LSQAMD_MODEL_TAPE:
      if (a.n_prog > 0) {
        // one formula per row range (the flattened dict-valued fit function of the reference,
        // src/lsqfit/__init__.py:1997-2042): each range is a launch of its own over offset views of the rows
        for (int i = 0; i < a.n_prog; ++i) {
          const TapeProgram &pg = a.progs[i];
          if (pg.n_rows <= 0) continue;
          if (pg.jit) {
            lsqamd_jit::LaunchArgs la;
            la.n_batch = (int32_t)nb; la.p_stride = a.p_stride; la.out_stride = a.out_stride; la.ymean_stride = a.ymean_stride;
            la.batch_active = a.batch_active;
            la.x = a.x + pg.row0 * a.n_x; la.p = a.p; la.ymean = a.ymean + pg.row0; la.wdiag = a.wdiag + pg.row0;
            la.in_block = a.in_block ? a.in_block + pg.row0 : nullptr;
            la.out_w = out_w + pg.row0 * ld; la.out_raw = out_raw ? out_raw + pg.row0 * ld : nullptr;
            la.ld = ld; la.n_data = pg.n_rows;
            hipError_t e = lsqamd_jit::launch(static_cast<const lsqamd_jit::Kernel *>(pg.jit), st, JAC, la);
            if (e != hipSuccess) return e;
            continue;
          }
          ModelDev q = m;
          q.x += pg.row0 * a.n_x; q.ymean += pg.row0; q.wdiag += pg.row0;
          if (q.in_block) q.in_block += pg.row0;
          q.out_w += pg.row0 * ld;
          if (q.out_raw) q.out_raw += pg.row0 * ld;
          q.n_data = pg.n_rows;
          q.tape += pg.tape_off;
          q.n_tape = pg.n_tape;
          hipLaunchKernelGGL((tape_model_kernel<JAC>), dim3((unsigned)((pg.n_rows + 63) / 64), nb), dim3(64), 0, st, q);
        }
        break;
      }
      if (a.jit) {   // the formula compiled at lsqamd_set_tape time (jit.hip); batched fits / many points: blockIdx.y
        lsqamd_jit::LaunchArgs la;
        la.n_batch = (int32_t)nb; la.p_stride = a.p_stride; la.out_stride = a.out_stride; la.ymean_stride = a.ymean_stride;
        la.batch_active = a.batch_active;
        la.x = a.x; la.p = a.p; la.ymean = a.ymean; la.wdiag = a.wdiag; la.in_block = a.in_block;
        la.out_w = out_w; la.out_raw = out_raw; la.ld = ld; la.n_data = a.n_data;
        return lsqamd_jit::launch(static_cast<const lsqamd_jit::Kernel *>(a.jit), st, JAC, la);
      }
      if (nb == 1 && a.tape_part && a.tape_jt && a.tape_poff) {
        TapeRev r;
        r.m = m; r.poff = a.tape_poff; r.part = a.tape_part; r.jt = a.tape_jt; r.ldn = a.tape_ldn;
        r.n_slots = a.tape_slots > 0 ? a.tape_slots : 1;
        r.n_groups = (a.n_data + 63) / 64;
        r.single = a.tape_single;
        int64_t wgs = (r.n_groups + 3) / 4;
        if (wgs > a.tape_wgs) wgs = a.tape_wgs;
        static const bool fwd = [] { const char *e = getenv("LSQAMD_TAPE"); return e && e[0] == 'f'; }();
        if (fwd) {   // developer knob: the forward-mode kernel, for comparison
          hipLaunchKernelGGL((tape_model_kernel<JAC>), dim3((unsigned)((a.n_data + 63) / 64), nb), dim3(64), 0, st, m);
          break;
        }
        static const bool whole = [] { const char *e = getenv("LSQAMD_TAPE"); return e && e[0] == 'w'; }();   // developer knob
        const bool by_segment = a.tape_seg && a.tape_n_seg > 0 && !whole;
        if (JAC && !(by_segment && a.tape_single == 2)) {   // (every column is stored in full otherwise)
          hipError_t e = hipMemsetAsync(a.tape_jt, 0, sizeof(double) * (size_t)((a.n_param + 1) * a.tape_ldn), st);
          if (e != hipSuccess) return e;
        }
        if (by_segment) {   // the root is a sum: segment by segment, partials never leave LDS
          TapeSeg sg;
          sg.r = r; sg.seg = a.tape_seg; sg.n_seg = a.tape_n_seg;
          sg.depth = a.tape_seg_depth; sg.slots = a.tape_seg_slots > 0 ? a.tape_seg_slots : 1;
          // two rows per lane once there are row groups enough to fill the chip either way (developer knob: LSQAMD_TAPE_ROWS);
          static const int rows_knob = [] { const char *e = getenv("LSQAMD_TAPE_ROWS"); return e ? atoi(e) : 0; }();
          // two only if the halved row groups times the chunks the tape admits still give every SIMD its three waves
          // (four rows: 0.76-0.80 against 0.74 ms, fewer waves and spills)
          const int64_t units2 = ((a.n_data + 127) / 128) * ((sg.n_seg + 7) / 8);
          const int R = rows_knob == 1 || rows_knob == 2 ? rows_knob : (a.n_data >= 16384 && units2 >= 12 * 256 ? 2 : 1);
          const int64_t groups = (a.n_data + 64 * R - 1) / (64 * R);
          sg.r.n_groups = groups;
          // chunks of segments: enough (row group, chunk) units for ~24 waves per CU (two rounds of the three per
          // SIMD the registers allow), at least 8 segments per unit
          static const int64_t per_cu = [] { const char *e = getenv("LSQAMD_TAPE_WAVES_PER_CU"); return e && atoi(e) > 0 ? (int64_t)atoi(e) : (int64_t)24; }();
          int64_t chunks = (per_cu * 256 + groups - 1) / groups;
          if (chunks > (sg.n_seg + 7) / 8) chunks = (sg.n_seg + 7) / 8;
          if (chunks < 1) chunks = 1;
          const int64_t cap = (a.tape_wgs * 4 * (int64_t)(a.tape_slot_cap > 0 ? a.tape_slot_cap : 1) * 64) / a.tape_ldn;   // room in the scratch
          if (chunks > cap) chunks = cap;
          if (chunks < 1) chunks = 1;
          sg.n_chunks = (int32_t)chunks;
          sg.seg_per_chunk = (int32_t)((sg.n_seg + chunks - 1) / chunks);
          sg.ftot = a.tape_part;
          const size_t lds = sizeof(double) * (size_t)(((a.n_param + 1) & ~(int64_t)1) + 4 * (sg.depth + sg.slots) * 64 * R);
          static PerDeviceMax lds_set[3];   // per device, thread-safe (common.h)
          const void *kfn = R == 2 ? reinterpret_cast<const void *>(tape_segment_kernel<JAC, 2>)
                                   : reinterpret_cast<const void *>(tape_segment_kernel<JAC, 1>);
          {
            const hipError_t e2 = lds_set[R].ensure(lds, [kfn](size_t want) {
              return hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
            });
            if (e2 != hipSuccess) return e2;
          }
          const int64_t units = groups * chunks;
          if (R == 2) hipLaunchKernelGGL((tape_segment_kernel<JAC, 2>), dim3((unsigned)((units + 3) / 4)), dim3(256), lds, st, sg);
          else hipLaunchKernelGGL((tape_segment_kernel<JAC, 1>), dim3((unsigned)((units + 3) / 4)), dim3(256), lds, st, sg);
          hipLaunchKernelGGL(tape_total_kernel, dim3((unsigned)((a.n_data + 255) / 256)), dim3(256), 0, st, sg.ftot,
                             sg.n_chunks, a.tape_ldn, a.n_data, a.ymean, a.tape_jt + (int64_t)a.n_param * a.tape_ldn,
                             a.wdiag, a.in_block, JAC ? nullptr : out_w, JAC ? nullptr : out_raw);
        } else {
          hipLaunchKernelGGL(tape_reverse_kernel<JAC>, dim3((unsigned)wgs), dim3(256), 0, st, r);
        }
        if (JAC) {
          dim3 grid((unsigned)((a.n_data + 63) / 64), (unsigned)((a.n_param + 1 + 63) / 64));
          hipLaunchKernelGGL(tape_finish_kernel, grid, dim3(256), 0, st, a.tape_jt, a.tape_ldn, a.n_data, a.n_param + 1,
                             a.wdiag, a.in_block, out_w, out_raw, ld);
        }
        break;
      }
      hipLaunchKernelGGL((tape_model_kernel<JAC>), dim3((unsigned)((a.n_data + 63) / 64), nb), dim3(64),
                         0, st, m);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

int tape_slots_of_op(int op) { return tape_slots_of(op); }

hipError_t launch_residual_ex(hipStream_t st, const ModelArgs &m, double *r_w, double *r_raw) {
  return launch_model<false>(st, m, r_w, r_raw, 1);
}
hipError_t launch_jacobian_ex(hipStream_t st, const ModelArgs &m, double *J_w, double *J_raw,
                              int64_t ld) {
  return launch_model<true>(st, m, J_w, J_raw, ld);
}

}  // namespace lsqamd
