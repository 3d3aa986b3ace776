// Developer microbenchmark (round 6, review item 6): do v_mfma_f64_16x16x4_f64 and v_fma_f64 share a datapath on gfx950?
//
// DESIGN.md prices whiten_synth_kernel (MFMA product + a sincos per synthesised Jacobian element on the fp64 VALU) against a
// "shared fp64 datapath".  That premise was inferred from two failed overlap experiments, never measured.  Here: ONE
// workgroup of 8 waves on one CU (wave i lands on SIMD i mod 4 -- checked with HW_ID and printed); wave A issues only fp64
// MFMAs (8 independent accumulator sets: no dependency stall), wave B only fp64 FMAs (16 independent chains), each for a fixed
// number of shader clocks, counting what it got done:
//   (1) A alone   (2) B alone   (3) A and B on the SAME SIMD   (4) A and B on DIFFERENT SIMDs of the CU
//   (5) two A on the same SIMD, (6) two B on the same SIMD  (controls: what plain time-sharing of ONE pipe looks like)
// If the rates of (3) add up (each keeps ~its solo rate) the pipes are separate and producer / consumer wave specialisation
// can pay; if each falls to about half, the datapath (or the issue slot) is shared and it cannot.
//   hipcc --offload-arch=gfx950 -O3 tools/dp_datapath.hip -o tools/dp_datapath && tools/dp_datapath
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

struct Out {
  unsigned long long count, cycles, t0, t1;
  unsigned hw_id, role;
};

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }   // s_memtime: shader clock

// role of each of the 8 waves: 0 idle, 1 MFMA, 2 FMA, 3 = MFMA and FMA interleaved in ONE wave (what whiten_synth does),
// 4 = fp32 FMAs only (v_pk_fma_f32: does NON-fp64 VALU work hide behind fp64 MFMAs?), 5 = 32-bit integer multiply-adds only
// (v_mad_u64_u32), 6 = ONE wave with 8 fp32 FMAs (4 v_pk_fma_f32) after every MFMA, 7 = ONE wave with 4 v_mad_u64_u32 after every MFMA
__global__ __launch_bounds__(512, 1) void probe(const int *roles, unsigned long long budget, Out *out, double seed) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int role = roles[wave];
  unsigned hw = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  __syncthreads();
  if (role == 0) return;
  unsigned long long n = 0;
  const unsigned long long t0 = now();
  unsigned long long t1 = t0;
  if (role == 1) {
    v4d acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4d){seed, 0.0, 0.0, 0.0};
    const double a = seed + lane * 1e-9, b = 1.0 - seed;
    do {
#pragma unroll
      for (int rep = 0; rep < 8; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      n += 64;
      t1 = now();
    } while (t1 - t0 < budget);
    double s = 0.0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 12345.678) out[8].count = 1;   // keep the accumulators alive
  } else if (role == 2) {
    double x[16];
    for (int i = 0; i < 16; ++i) x[i] = seed + i + lane * 1e-9;
    const double a = 1.0 - 1e-12 * seed, b = 1e-13;
    do {
#pragma unroll
      for (int rep = 0; rep < 16; ++rep)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], a, b);
      n += 256;
      t1 = now();
    } while (t1 - t0 < budget);
    double s = 0.0;
    for (int i = 0; i < 16; ++i) s += x[i];
    if (s == 12345.678) out[8].count = 1;
  } else if (role == 4 || role == 5) {
    float xf[16];
    int xi[16];
    for (int i = 0; i < 16; ++i) { xf[i] = (float)seed + i + lane * 1e-3f; xi[i] = i + lane; }
    const float a = 1.0f - 1e-6f * (float)seed, b = 1e-7f;
    const int ia = 3 + (int)seed, ib = 7;
    do {
      if (role == 4) {
#pragma unroll
        for (int rep = 0; rep < 16; ++rep)
#pragma unroll
          for (int i = 0; i < 16; ++i) xf[i] = __builtin_fmaf(xf[i], a, b);
      } else {
#pragma unroll
        for (int rep = 0; rep < 16; ++rep)
#pragma unroll
          for (int i = 0; i < 16; ++i) xi[i] = xi[i] * ia + ib;
      }
      n += 256;
      t1 = now();
    } while (t1 - t0 < budget);
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += xf[i] + (float)xi[i];
    if (s == 12345.678f) out[8].count = 1;
  } else if (role == 6 || role == 7) {
    v4d acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4d){seed, 0.0, 0.0, 0.0};
    float xf[16];
    int xi[16];
    for (int i = 0; i < 16; ++i) { xf[i] = (float)seed + i + lane * 1e-3f; xi[i] = i + lane; }
    const double a = seed + lane * 1e-9, b = 1.0 - seed;
    const float fa = 1.0f - 1e-6f * (float)seed, fb = 1e-7f;
    const int ia = 3 + (int)seed, ib = 7;
    do {
#pragma unroll
      for (int rep = 0; rep < 8; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
          if (role == 6) {
#pragma unroll
            for (int q = 0; q < 8; ++q) xf[(2 * i + q) & 15] = __builtin_fmaf(xf[(2 * i + q) & 15], fa, fb);
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) xi[(2 * i + q) & 15] = xi[(2 * i + q) & 15] * ia + ib;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      n += 64;
      t1 = now();
    } while (t1 - t0 < budget);
    double s = 0.0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 16; ++i) s += xf[i] + xi[i];
    if (s == 12345.678) out[8].count = 1;
  } else {
    // one wave, both kinds interleaved: 4 FMAs after every MFMA (an MFMA occupies the pipe for 16 passes x 4 clocks; four
    // wave64 FMAs are 4 x 4 = 16 issue clocks: would fit in its shadow if the pipes were separate)
    v4d acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4d){seed, 0.0, 0.0, 0.0};
    double x[16];
    for (int i = 0; i < 16; ++i) x[i] = seed + i + lane * 1e-9;
    const double a = seed + lane * 1e-9, b = 1.0 - seed, fa = 1.0 - 1e-12 * seed, fb = 1e-13;
    do {
#pragma unroll
      for (int rep = 0; rep < 8; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
          x[(2 * i) & 15] = __builtin_fma(x[(2 * i) & 15], fa, fb);
          x[(2 * i + 1) & 15] = __builtin_fma(x[(2 * i + 1) & 15], fa, fb);
          x[(2 * i + 8) & 15] = __builtin_fma(x[(2 * i + 8) & 15], fa, fb);
          x[(2 * i + 9) & 15] = __builtin_fma(x[(2 * i + 9) & 15], fa, fb);
          __builtin_amdgcn_sched_barrier(0);     // keep the 1 : 4 interleave (the scheduler would sort the kinds apart)
        }
      n += 64;      // MFMAs; FMAs = 4 n
      t1 = now();
    } while (t1 - t0 < budget);
    double s = 0.0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 16; ++i) s += x[i];
    if (s == 12345.678) out[8].count = 1;
  }
  if (lane == 0) {
    out[wave].count = n; out[wave].cycles = t1 - t0; out[wave].t0 = t0; out[wave].t1 = t1;
    out[wave].hw_id = hw; out[wave].role = (unsigned)role;
  }
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  int *droles; Out *dout;
  CHK(hipMalloc(&droles, 8 * sizeof(int)));
  CHK(hipMalloc(&dout, 9 * sizeof(Out)));
  const unsigned long long budget = 20000000ull;   // shader-clock ticks (s_memtime runs at 100 MHz on gfx9: 0.2 s) -- see the note printed
  struct Cfg { const char *name; int roles[8]; } cfgs[] = {
      {"(1) MFMA wave alone", {1, 0, 0, 0, 0, 0, 0, 0}},
      {"(2) FMA wave alone", {2, 0, 0, 0, 0, 0, 0, 0}},
      {"(3) MFMA + FMA, same SIMD (waves 0, 4)", {1, 0, 0, 0, 2, 0, 0, 0}},
      {"(4) MFMA + FMA, different SIMDs (waves 0, 1)", {1, 2, 0, 0, 0, 0, 0, 0}},
      {"(5) MFMA + MFMA, same SIMD", {1, 0, 0, 0, 1, 0, 0, 0}},
      {"(6) FMA + FMA, same SIMD", {2, 0, 0, 0, 2, 0, 0, 0}},
      {"(7) one wave, 1 MFMA : 4 FMA interleaved", {3, 0, 0, 0, 0, 0, 0, 0}},
      {"(8) MFMA on all 4 SIMDs + FMA on all 4 SIMDs", {1, 1, 1, 1, 2, 2, 2, 2}},
      {"(9) MFMA on all 4 SIMDs", {1, 1, 1, 1, 0, 0, 0, 0}},
      {"(10) FMA on all 4 SIMDs", {2, 2, 2, 2, 0, 0, 0, 0}},
      {"(11) fp32 FMA wave alone", {4, 0, 0, 0, 0, 0, 0, 0}},
      {"(12) MFMA + fp32 FMA, same SIMD", {1, 0, 0, 0, 4, 0, 0, 0}},
      {"(13) int32 mad wave alone", {5, 0, 0, 0, 0, 0, 0, 0}},
      {"(14) MFMA + int32 mad, same SIMD", {1, 0, 0, 0, 5, 0, 0, 0}},
      {"(15) fp64 FMA + fp32 FMA, same SIMD", {2, 0, 0, 0, 4, 0, 0, 0}},
      {"(16) one wave, 1 MFMA : 8 fp32 FMA (4 v_pk_fma_f32) interleaved", {6, 0, 0, 0, 0, 0, 0, 0}},
      {"(17) one wave, 1 MFMA : 4 v_mad_u64_u32 interleaved", {7, 0, 0, 0, 0, 0, 0, 0}},
  };
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  printf("fp64 MFMA (v_mfma_f64_16x16x4_f64, 2048 flop) vs fp64 VALU (v_fma_f64 wave64, 128 flop) on ONE CU of gfx950\n");
  printf("rates per wave in instructions per microsecond of wall time (HIP events around the launch); SIMD = HW_ID bits [5:4]\n\n");
  for (auto &c : cfgs) {
    CHK(hipMemcpy(droles, c.roles, sizeof(c.roles), hipMemcpyHostToDevice));
    CHK(hipMemset(dout, 0, 9 * sizeof(Out)));
    for (int warm = 0; warm < 2; ++warm) {
      CHK(hipEventRecord(e0));
      hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, droles, warm == 0 ? budget / 20 : budget, dout, 0.5);
      CHK(hipEventRecord(e1));
      CHK(hipDeviceSynchronize());
    }
    float ms = 0.f;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    Out h[9];
    CHK(hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost));
    printf("%s   [launch %.3f ms]\n", c.name, ms);
    for (int w = 0; w < 8; ++w) {
      if (!h[w].role) continue;
      const double us = 1e3 * ms * (double)h[w].cycles / (double)(h[w].cycles ? h[w].cycles : 1);   // every wave runs (almost) the whole launch
      const double ticks = (double)h[w].cycles;
      const char *kind = h[w].role == 1 ? "MFMA" : h[w].role == 2 ? "FMA " : h[w].role == 4 ? "FMA32" : h[w].role == 5 ? "IMAD32" :
                         h[w].role == 6 ? "MFMA(+8 FMA32 each)" : h[w].role == 7 ? "MFMA(+4 IMAD each)" : "MFMA(+4 FMA each)";
      const double flop = h[w].role == 1 ? 2048.0 : (h[w].role == 2 || h[w].role == 4 || h[w].role == 5) ? 128.0 :
                          h[w].role == 6 ? 2048.0 + 8 * 128.0 : h[w].role == 7 ? 2048.0 + 4 * 128.0 : 2048.0 + 4 * 128.0;
      printf("    wave %d  SIMD %u  CU %u  %-18s %12llu instr in %10.0f ticks  = %9.2f instr/us = %7.2f GFLOP/s  (ticks/us %.1f)\n", w,
             (h[w].hw_id >> 4) & 3, (h[w].hw_id >> 8) & 15, kind, h[w].count, ticks, (double)h[w].count / us,
             (double)h[w].count * flop / us * 1e-3, ticks / us);
    }
  }
  printf("\nreference: 78.6 TFLOP/s / (256 CU x 4 SIMD) = 76.8 GFLOP/s per SIMD at 2.4 GHz for EITHER instruction kind\n");
  return 0;
}
