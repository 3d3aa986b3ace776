// The factorisation streamed behind the J^T J product (sf_chol.hip): host-side interface.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lsqamd {

constexpr int32_t SF_TIMEOUT_INFO = -88;     // *info when a bounded wait of the streamed factorisation gave up

struct SfLaunch {
  hipStream_t st_main = nullptr;    // the handle's stream: everything before is waited for, everything after waits for both
  hipStream_t st_work = nullptr;    // CU-masked: the chip minus the reserved CUs
  hipStream_t st_chain = nullptr;   // CU-masked: the reserved CUs
  hipEvent_t ev_fork = nullptr, ev_work = nullptr, ev_chain = nullptr;
  int32_t n_workers = 0;            // resident workgroups of the worker launch (2 per CU of st_work)
  const double *J = nullptr;        // whitened Jacobian [n_rows][ldj]
  int64_t ldj = 0, n_rows = 0;
  int32_t splits = 1;
  double *slabs = nullptr;          // [splits][P][ld_slab]
  int64_t ld_slab = 0, split_stride = 0;
  double *M = nullptr;              // [P][ldm], ldm >= P + 128
  int64_t ldm = 0;
  double *uinv = nullptr;           // potrf work: inverses of the diagonal blocks
  double *apk = nullptr;            // packed upper tiles of A (written here)
  const double *prior = nullptr;    // prior precision added into A (nullable)
  int32_t prior_dense = 0;
  const double *gvec = nullptr;     // [P] right-hand side (complete before the call)
  int64_t P = 0;
  const int32_t *q1 = nullptr;      // device lists (sf_q1_fill / sf_q2_fill)
  int32_t q1_run0[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int32_t *q2 = nullptr;
  int32_t *sync = nullptr;          // sf_sync_bytes(P)
  int32_t *info = nullptr;          // pivot failure word (as potrf_upper's)
  int32_t scaler = 0;               // LSQAMD_SCALE_*: the scaling D is updated block by block on the way
  double *dscale = nullptr;
  const double *mu_dev = nullptr;   // damping parameter (device)
  long long *dbg = nullptr;         // optional device buffer [8 + 4 T]: wall-clock stamps of the chain (developer tool)
  int32_t idle_max = 16;            // longest idle sleep of a worker, x ~0.2 us
  int32_t chain_tiles = 1;          // panel tiles right of the diagonal the chain makes itself (the lists must match)
};

size_t sf_sync_bytes(int64_t P);
int64_t sf_q1_count(int64_t P, int32_t splits);
int64_t sf_q2_count(int64_t P, int chain_tiles);
void sf_q1_fill(int64_t P, int32_t splits, int group_rows, int32_t *out, int32_t run0[9]);
void sf_q2_fill(int64_t P, int chain_tiles, int32_t *out);
// Enqueues: J^T J (slabs + packed tiles + prior), M = U with A + mu D^2 = U^T U, column P of M = U^-T g, block inverses,
// D updated.  Requires P a multiple of 128, P >= 256, n_rows a multiple of 16, 16-byte aligned rows.
hipError_t sf_launch(const SfLaunch &a);
// 0 on success; the two CU-masked streams and the number of CUs the worker stream may use
int sf_streams_create(int reserve_per_xcd, int mode, hipStream_t *work, hipStream_t *chain, int *n_work_cus);

}  // namespace lsqamd

// ---- the experiment's own C entry points (were in include/lsqfit_amd.h until round 6: a lost experiment has no place in the
// product ABI; tools/experiments/sf/build.sh links them into a VARIANT library, lsqfit_amd/build/libsf.so) ----
extern "C" {
/* developer probe: where do the workgroups of a launch on `stream` run?  dev_out[2 b] = XCC_ID, dev_out[2 b + 1] = HW_ID of
 * workgroup b (n_wg workgroups of one wave that ask for lds_bytes of LDS each; tools/exp_cumask.py) */
int lsqamd_debug_where(void *stream, int32_t n_wg, uint32_t *dev_out, int32_t lds_bytes);
/* developer / test entry points of the factorisation streamed behind the J^T J product (csrc/sf_chol.hip; the first trial solve
 * after an accepted LM step: replaces gsl's solver init + solve behind src/lsqfit/_gsl.pyx:646-653,:677).  Device pointers.
 *   J [n_rows][ldj] (P columns used), prior (nullable: dense P x P or diagonal), g [P], mu, d [P] (updated like the LM scaling)
 *   -> apk (packed upper 128 x 128 tiles of A = J^T J + prior), M [P][P + 128] (U with A + mu D^2 = U^T U; column P = U^-T g).
 * reserve_per_xcd CUs of every XCD run the latency chain (hipExtStreamCreateWithCUMask; mask_mode 'c' / 'i': bit layout);
 * 0 = no masks (plain streams: results only). */
size_t lsqamd_op_sf_work_bytes(int64_t n_rows, int64_t P, int32_t splits);
int lsqamd_op_sf_factor(void *stream, const double *J, int64_t ldj, int64_t n_rows, int64_t P, int32_t splits, int32_t group_rows,
                        int32_t reserve_per_xcd, int32_t mask_mode, const double *prior, int32_t prior_dense, const double *g,
                        double mu, int32_t scaler, double *d, double *apk, double *M, void *work, size_t work_bytes,
                        int32_t *info_host, long long *dev_stamps /* nullable: [24 + 4 P / 128] wall-clock stamps and per-item-type time sums */, int32_t idle_max);
}
