// Device math shared by the model kernels (model.hip) and the kernel that synthesises Jacobian
// rows inside the whitening product (gemm_tn_f64.hip): both must produce the same bits.
#pragma once
#include <hip/hip_runtime.h>

namespace lsqamd {

// sin and cos of a moderate argument (|t| < 1e5; here t = w_k x_i <= a few 10^4): two-constant
// Cody-Waite reduction by pi/2 with FMAs (the product n * PIO2_HI is formed exactly inside the
// FMA), then the fdlibm minimax kernels on [-pi/4, pi/4].  Absolute error ~1e-16; about a third of
// the instructions of the general-range library sincos, which remains the fallback for huge t.
// Beyond the fast range (and for NaN / Inf): the library routine, kept out of line -- inlined, its
// Payne-Hanek branch costs the fused whitening kernel 34 VGPRs and a scratch frame (1.83 vs 1.66 ms).
struct SinCos { double s, c; };
__device__ __attribute__((noinline)) inline SinCos sincos_far(double t) {
  SinCos r;
  sincos(t, &r.s, &r.c);
  return r;
}
__device__ __attribute__((noinline)) inline double cos_far(double t) { return cos(t); }

// Two-term Cody-Waite reduction carried by FMAs (the products n * HI, n * LO are exact inside the FMA; HI + LO
// is pi/2 to 2e-32): 1.1e-16 absolute up to |t| = 1e13 (checked against libm), where a double's own spacing is
// already 2e-3.
constexpr double TRIG_FAST_LIMIT = 1.0e13;
constexpr double ROUND_MAGIC = 6755399441055744.0;      // 1.5 * 2^52
// v with its sign flipped where bit 1 of q is set: the bit moved onto the sign bit, one shift + and + xor
__device__ __forceinline__ double flip_sign(double v, int q) {
  return __hiloint2double(__double2hiint(v) ^ (int)(((unsigned)q << 30) & 0x80000000u), __double2loint(v));
}
// FAR = false: the caller guarantees |t| < TRIG_FAST_LIMIT (a device flag computed from the parameters, see
// trig_range_kernel) -- no far-range code in the kernel at all
template <bool FAR = true>
__device__ __forceinline__ void sincos_moderate(double t, double *sn, double *cs) {
  if (FAR && !(fabs(t) < TRIG_FAST_LIMIT)) {
    const SinCos f = sincos_far(t);
    *sn = f.s;
    *cs = f.c;
    return;
  }
  // n = t * 2/pi rounded to an integer by adding 1.5 * 2^52 (|n| < 2^51: the sum's last place is 1, the rounding
  // mode does the rest); its two low bits -- the quadrant -- are the two low bits of the sum's mantissa.  No
  // double -> integer conversion (four instructions, none of them full rate).
  const double m = __builtin_fma(t, 6.36619772367581382433e-01, ROUND_MAGIC);      // 2/pi
  const double n = m - ROUND_MAGIC;
  double r = __builtin_fma(-n, 1.57079632679489655800e+00, t);   // pi/2 = HI + MID + ...
  r = __builtin_fma(-n, 6.12323399573676603587e-17, r);
  const double z = r * r;
  const double ps = -1.66666666666666324348e-01 +
                    z * (8.33333333332248946124e-03 +
                         z * (-1.98412698298579493134e-04 +
                              z * (2.75573137070700676789e-06 +
                                   z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
  const double pc = 4.16666666666666019037e-02 +
                    z * (-1.38888888888741095749e-03 +
                         z * (2.48015872894767294178e-05 +
                              z * (-2.75573143513906633035e-07 +
                                   z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
  const double s = __builtin_fma(r * z, ps, r);
  const double c = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
  const int q = __double2loint(m);
  const double s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
  *sn = flip_sign(s1, q);              // quadrants 2, 3
  *cs = flip_sign(c1, q + 1);          // quadrants 1, 2
}

// cos alone (the residual kernels): ONE polynomial with its coefficients chosen per lane by the quadrant
// instead of both -- the same operations in the same order as the cosine of sincos_moderate, so the same bits.
template <bool FAR = true>
__device__ __forceinline__ double cos_moderate(double t) {
  if (FAR && !(fabs(t) < TRIG_FAST_LIMIT)) return cos_far(t);
  const double m = __builtin_fma(t, 6.36619772367581382433e-01, ROUND_MAGIC);
  const double n = m - ROUND_MAGIC;
  double r = __builtin_fma(-n, 1.57079632679489655800e+00, t);
  r = __builtin_fma(-n, 6.12323399573676603587e-17, r);
  const double z = r * r;
  const int q = __double2loint(m);
  const bool odd = q & 1;             // cos(t) = -+ sin(r) in the odd quadrants, +- cos(r) in the even ones
  const double c5 = odd ? 1.58969099521155010221e-10 : -1.13596475577881948265e-11;
  const double c4 = odd ? -2.50507602534068634195e-08 : 2.08757232129817482790e-09;
  const double c3 = odd ? 2.75573137070700676789e-06 : -2.75573143513906633035e-07;
  const double c2 = odd ? -1.98412698298579493134e-04 : 2.48015872894767294178e-05;
  const double c1 = odd ? 8.33333333332248946124e-03 : -1.38888888888741095749e-03;
  const double c0 = odd ? -1.66666666666666324348e-01 : 4.16666666666666019037e-02;
  const double p = c0 + z * (c1 + z * (c2 + z * (c3 + z * (c4 + z * c5))));
  const double a = odd ? r * z : z * z;
  const double b = odd ? r : __builtin_fma(-0.5, z, 1.0);
  const double v = __builtin_fma(a, p, b);
  return flip_sign(v, q + 1);
}

}  // namespace lsqamd
