// Internal to the host side of the library (api.hip, scipy_methods.hip): the fit handle and the
// device-side building blocks the trust-region drivers share.  Not part of the C ABI.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "common.h"
#include "jit.h"

namespace lsqamd_host {

void comm_release(lsqamd_fit *f);                                  // comm.hip
// process-wide recycling of what a handle needs from the runtime besides its workspace (api.hip): creating and releasing a
// pinned block or an event costs tens to hundreds of microseconds (hipHostFree waits for the device) -- more than a small fit
void *pinned_take(size_t bytes, size_t *granted);
void pinned_give(void *p, size_t granted, int dev = -1);   // dev: the device it was taken on (-1: the current one)
hipEvent_t event_take();
void event_give(hipEvent_t e, int dev = -1);
hipStream_t stream_take();          // a non-blocking stream (recycled ones are idle: their last owner synchronised them)
void stream_give(hipStream_t s, int dev = -1);   // dev: the device it was taken on (-1: the current one)
int comm_all_reduce(lsqamd_fit *f, double *buf, int64_t count);   // sums enqueued on f->st
int comm_all_reduce_on(lsqamd_fit *f, hipStream_t st, double *buf, int64_t count);   // ... on another stream of the handle

struct TimerSlot {
  double total_ms = 0.0;
  int64_t count = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  // event pairs OWNED by another timer's `pending` list that count for this timer too (resolved first, never recycled here)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_shared;
};

}  // namespace lsqamd_host

struct lsqamd_fit {
  lsqamd_config cfg;
  lsqamd_options opt;
  hipStream_t st = nullptr;
  int dev = -1;                // the device that was current at lsqamd_create: recycled blocks and events are filed under it
  bool st_used = false;        // the handle has been given a stream (lsqamd_create): the destructor waits for it
  std::string err;

  int64_t N = 0, P = 0, ld = 0, ldm = 0, npk = 0, ncols_aug = 0;
  int32_t splits = 1;
  int64_t npartial = 256;

  // device buffers
  double *x = nullptr, *ymean = nullptr, *wdiag = nullptr;
  uint8_t *in_block = nullptr;
  int32_t *row_param = nullptr;  // >= 0: parameter row f_i = p_j (lsqamd_set_param_rows)
  bool have_param_rows = false;
  int64_t *blk_row0 = nullptr, *blk_size = nullptr, *blk_woff = nullptr;
  double *wt = nullptr;
  double *prior_mean = nullptr, *prior_prec = nullptr;
  double *p_dev = nullptr, *p_trial = nullptr;
  double *r = nullptr, *r_raw = nullptr;
  double *J = nullptr, *Jraw = nullptr;
  double *slabs = nullptr;
  double *redbuf = nullptr;  // [packed J^T J | J^T f | chi2]
  double *red_scalar = nullptr;
  double *M = nullptr, *chol_work = nullptr, *yv = nullptr, *diag_dev = nullptr, *tvec = nullptr;
  double *dscale = nullptr;  // the scaling matrix D, device-resident mirror of hdiag
  double *lmd = nullptr;     // LM state record on the device (LMS_*: chi2, mu, nu, delta, rho, flags)
  double *partial = nullptr, *Wl = nullptr, *cov = nullptr, *scal = nullptr;
  int32_t *info_dev = nullptr;
  int32_t *tape = nullptr;
  double *consts = nullptr;
  int32_t n_tape = 0;
  // reverse-mode tape Jacobian (model.hip): slot offsets, partial store, transposed Jacobian
  int32_t *tape_poff = nullptr;
  int32_t *tape_seg = nullptr;     // root-sum segments of the tape (set_tape), tape_n_seg of them (0: not a sum)
  int32_t tape_single = 0;   // 1: no parameter is read twice by the tape; 2: and every one is read
  int32_t tape_n_seg = 0, tape_seg_depth = 0, tape_seg_slots = 0;   // deepest stack / most partials of a segment
  double *tape_part = nullptr, *tape_jt = nullptr;
  int64_t tape_ldn = 0, tape_wgs = 0;
  int32_t tape_cap = 0, tape_slots = 0, tape_slot_cap = 0;
  std::vector<lsqamd::TapeProgram> progs;   // one formula per row range (lsqamd_set_tape_programs); empty: one tape for all rows
  int progs_compiled = 0;
  const void *jit = nullptr;   // the tape compiled (jit.hip); null: interpreted (jit_why says why)
  double *nrm_part = nullptr;  // few parameters: per-workgroup sums of the fused normal-equation kernel, then their total
  int nrm_in_tail = 0;         // > 0: that many per-workgroup sums wait for the accept-tail kernel to total and unpack them
  bool J_stale = false;        // the last normal equations were formed WITHOUT writing J (jit.hip lsqamd_jit_nrm): ensure_J() first
  std::string jit_why;
  int32_t *syrk_map = nullptr;
  int32_t syrk_nwork = 0;

  // host mirrors of the block structure
  std::vector<int64_t> h_row0, h_size, h_modes, h_woff;
  std::vector<int32_t> h_tri;
  bool uniform_blocks = false;
  int32_t uniform_tri = 0;
  bool have_x = false, have_data = false, have_prior = false, have_tape = false;

  // reduce hook
  lsqamd_reduce_fn reduce = nullptr;
  void *reduce_user = nullptr;
  bool adds_prior = true;
  // in-library RCCL communicator (comm.hip); takes precedence over the hook
  void *comm = nullptr;
  int32_t comm_rank = 0, comm_nranks = 1;
  std::string comm_key;            // its entry in the process-wide registry of communicators (comm.hip)
  // grouped exchange (api.hip eval_normal_dev; LSQAMD_EXCHANGE_GROUPS = G > 1, read at lsqamd_create): the J^T J work list cut
  // into G groups of tile rows, group g's packed tiles summed over the ranks on `xst` while group g + 1 is computed on `st`
  int32_t xg = 1;                  // groups (1: one launch, one exchange on the step's stream)
  int32_t xg_row[9] = {0};         // tile rows [xg_row[g], xg_row[g + 1]) of group g
  int32_t xg_work[9] = {0};        // its entries in syrk_map_g: [xg_work[g], xg_work[g + 1])
  int64_t xg_tile[9] = {0};        // its packed tiles: [xg_tile[g], xg_tile[g + 1])
  int32_t *syrk_map_g = nullptr;   // the groups' work lists, one after the other (device); xg_signal: ONE list in group-major order
  // xg_signal (LSQAMD_EXCHANGE_MODE=signal, the default for G > 1): ONE product launch whose workgroups count a group's
  // finished entries in xg_ctr[g] (release); the exchange stream waits for the count with a one-wave kernel, then packs and
  // exchanges the group -- no split launches.  LSQAMD_EXCHANGE_MODE=split: one product launch per group (round 6's first form)
  bool xg_signal = false;
  int32_t *xg_ctr = nullptr;       // device: [0..8) the counters, [8] "a wait timed out" (sticky)
  int32_t xg_count[8] = {0};       // entries per group
  hipStream_t xst = nullptr;       // exchange stream (taken at the first grouped exchange, given back with the handle)
  hipEvent_t xg_ready[8] = {nullptr}, xg_done[8] = {nullptr};

  // solver = qr: caller-provided device scratch (lsqamd_set_qr_work) and what the last run did
  void *qr_work = nullptr;
  size_t qr_work_bytes = 0;
  int32_t qr_passes = 0;
  double qr_delta = NAN;
  int32_t cov_dropped = 0;       // > 0: the last covariance is the reference's truncated inverse with that many directions dropped
  bool cov_inaccurate = false;   // the last covariance was delivered from a factorisation that did not meet its accuracy test

  // box bounds of the reflective trust-region method (empty: none)
  std::vector<double> lb, ub;
  // scipy plugin: robust loss over the rows of the whitened residual (robust.hip), characteristic parameter scales
  int32_t loss = 0;
  double f_scale = 1.0;
  std::vector<double> x_scale;
  bool robust() const { return loss != 0 && (opt.trs == LSQAMD_TRS_TRF || opt.trs == LSQAMD_TRS_DOGBOX); }
  // parameters the residual is linear in (empty: none): variable projection (api.hip iterate_varpro)
  std::vector<char> linear;

  // LM state (host)
  std::vector<double> hx, hg, hdiag, hdx, hv, hcoln, htmp;
  double chi2 = 0.0, mu = 0.0, delta = 0.0;
  long nu = 2;
  // plain lm keeps its vectors on the device (api.hip iterate_device): the host mirrors hx / hg /
  // hdiag / hcoln / hv / hdx are refreshed on demand (refresh_mirrors)
  bool dev_lm = false, mirrors_stale = false;
  bool prior_deferred = false;  // eval_normal_dev left the prior's share of g / chi2 to the accept-tail kernel
  double lm_seq_expect = 0.0;   // half steps published so far (the record's LMS_SEQ the host waits for)
  bool lm_zero_copy = false;   // the device kernels mirror the LM record into pin_lm themselves (no copy per read)
  // f->r holds the whitened residual AT r_ptr's current contents (set by iterate_device right before
  // the accepted point's normal equations, consumed there: the fused Jacobian path needs it)
  bool r_fresh = false, used_synth = false, used_nrm = false;
  const double *r_ptr = nullptr;
  int32_t conv_info_dev = 0;
  bool initialised = false, have_cov = false, have_dense_A = false;
  bool cov_unavailable = false;     // robust fit whose loss-scaled Jacobian had no covariance: get_cov must not recompute it
                                    // from the unscaled matrix the last evaluation left (cleared by the next evaluation)
  int32_t nit = 0, nfev = 0, njev = 0, ntrial = 0, chol_fail = 0;
  bool qr_steps_on = false;    // solver = qr: this fit's trial steps come from the orthogonal factorisation from now on
  int32_t qr_trials = 0;       // trial steps solved from the orthogonal factorisation after a failed pivot (solver = qr)
  double logdet = NAN;

  // pinned host staging for the per-step transfers (pageable copies go through a blit
  // kernel and cost ~40 us each): [g | chi2](P+1), coln(P), v(P), diag(P), x_trial(P), scalars(8)
  double *pin = nullptr;
  double *pin_g = nullptr, *pin_c = nullptr, *pin_v = nullptr, *pin_d = nullptr, *pin_x = nullptr, *pin_s = nullptr;
  double *pin_lm = nullptr;   // the device's LM state record, as last read
  double *pin_fit = nullptr;  // where the one-launch fit kernel publishes its record block (jit.h FitArgs::pub)
  double *fit_block = nullptr;          // that block in DEVICE memory (FitArgs::host): what the kernel writes first, the fallback
  std::vector<double> fit_rec;          // the block as the host verified it (run_one_launch): everything later reads THIS
  bool cov_host_valid = false;    // pin_fit[96 ..] holds the covariance f->cov holds (get_cov serves it without a copy)
  bool used_one_launch = false;   // the last lsqamd_run was ONE launch (api.hip run_one_launch); lsqamd_debug_flags bit 5
  std::vector<hipEvent_t> event_pool;
  size_t pin_bytes = 0;     // size class the pinned block came with (pinned_take)
  // pinned staging arena for small host inputs (api.hip Upload): a setter copies in, queues the upload and returns
  // without waiting for the stream; regions are not reused before a point where the stream is known to have drained
  char *stage = nullptr;
  size_t stage_bytes = 0, stage_off = 0;

  // timing
  bool timing = false;
  lsqamd_host::TimerSlot timers[LSQAMD_T_COUNT];

  // cosine model: max |x| (set_x) and the device flag "some |w x| may leave the fast range of the trig reduction"
  double xmax = 0.0;
  int32_t *trig_far = nullptr;
  // captured LM step (iterate_device): [p-buffer parity][0 = trial, 1 = accepted branch]
  hipGraphExec_t step_exec[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  int step_seen[2][2] = {{0, 0}, {0, 0}};   // eager executions so far (the first one is the warm-up)
  bool step_graph_off = false;               // capture failed once on this handle: stay eager
  double *p_buf0 = nullptr;                  // the buffer p_dev started as (parity 0)
  int64_t graph_launches = 0;
  void drop_step_graphs() {
    for (auto &row : step_exec)
      for (auto &e : row)
        if (e) { (void)hipGraphExecDestroy(e); e = nullptr; }
    for (auto &row : step_seen) row[0] = row[1] = 0;
  }

  // compiled formulas are held (retained by lsqamd_jit::compile_tape): handed back when the tape is replaced or the handle goes
  void drop_jit() {
    if (jit) lsqamd_jit::release(static_cast<const lsqamd_jit::Kernel *>(jit));
    jit = nullptr;
    for (auto &pg : progs)
      if (pg.jit) { lsqamd_jit::release(static_cast<const lsqamd_jit::Kernel *>(pg.jit)); pg.jit = nullptr; }
  }

  ~lsqamd_fit() {  // every exit path (including the failure returns of lsqamd_create) ends here
    // staged uploads read the pinned arena, kernels write the pinned block: neither goes back to the process-wide recycler
    // (where the next handle may take it at once) before the stream has drained -- and a compiled kernel is released only
    // then: once unheld it may be unloaded by another thread's compile_tape while queued work still runs it
    if (st_used) (void)hipStreamSynchronize(st);
    drop_jit();
    drop_step_graphs();
    for (auto &t : timers)
      for (auto &pr : t.pending) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
      }
    for (hipEvent_t e : event_pool) lsqamd_host::event_give(e, dev);
    if (pin) lsqamd_host::pinned_give(pin, pin_bytes, dev);
    if (stage) lsqamd_host::pinned_give(stage, stage_bytes, dev);
    lsqamd_host::comm_release(this);
    if (xst) {
      (void)hipStreamSynchronize(xst);
      lsqamd_host::stream_give(xst, dev);
    }
    for (int g = 0; g < 8; ++g) {
      if (xg_ready[g]) lsqamd_host::event_give(xg_ready[g], dev);
      if (xg_done[g]) lsqamd_host::event_give(xg_done[g], dev);
    }
  }
};

#define FAIL(fit, code, ...)                         \
  do {                                               \
    char _b[512];                                    \
    snprintf(_b, sizeof(_b), __VA_ARGS__);           \
    (fit)->err = _b;                                 \
    return (code);                                   \
  } while (0)

#define HIPCHK(fit, expr)                                                              \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) FAIL(fit, LSQAMD_EHIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

namespace lsqamd_host {

hipEvent_t take_event(lsqamd_fit *f);  // events are recycled: creating one costs ~10 us

struct Scope {  // HIP-event bracket for one phase
  lsqamd_fit *f;
  int which;
  hipEvent_t a = nullptr, b = nullptr;
  hipStream_t on;
  int also;     // a second timer the same interval counts for (-1: none)
  Scope(lsqamd_fit *fit, int w, hipStream_t stream = nullptr, int also_ = -1) : f(fit), which(w), on(stream ? stream : fit->st), also(also_) {
    if (f->timing) {
      a = take_event(f);
      b = take_event(f);
      (void)hipEventRecord(a, on);
    }
  }
  ~Scope() {
    if (f->timing) {
      (void)hipEventRecord(b, on);
      f->timers[which].pending.emplace_back(a, b);
      if (also >= 0) f->timers[also].pending_shared.emplace_back(a, b);
    }
  }
};

// ---- device-side building blocks (api.hip) ----------------------------------------------------
// sum count doubles over the ranks of a sharded fit, in place (RCCL communicator or the hook)
int do_reduce(lsqamd_fit *f, double *buf, int64_t count);
// (J^T J)^-1 and log det J^T J at the current point: normal equations (api.hip) / CholeskyQR (qr.hip)
int do_covariance(lsqamd_fit *f);
int do_covariance_qr(lsqamd_fit *f);
int covariance_rank_deficient(lsqamd_fit *f);   // rankdef.hip
// (A + mu D^2) v = g through the CholeskyQR factor of [J ; W_prior ; sqrt(mu) D] (qr.hip): v -> f->yv[P..2P)
int solve_damped_qr(lsqamd_fit *f, double mu);
size_t qr_work_bytes(const lsqamd_fit *f);
// whitened residual at device parameters p -> f->r ; chi2 (all ranks) on return; counts one nfev
int eval_residual_dev(lsqamd_fit *f, const double *p, double *chi2_out);
// J, A = J^T J (+ prior), g, chi2, column norms at device parameters p (all-reduced); counts one njev
int eval_normal_dev(lsqamd_fit *f, const double *p, bool mirror = true);
// host copies of x, g, D, column norms, v after steps that kept them on the device
int refresh_mirrors(lsqamd_fit *f);
// the whitened Jacobian at the current point in f->J (a no-op unless the fused normal-equation kernel skipped it)
int ensure_J(lsqamd_fit *f);
// (A + mu D^2) v = g -> f->hv ; LSQAMD_ENOTPD when a pivot fails.  diag_host nullptr: the
// device-resident D.  frozen_host (with mu = 0): flags of parameters taken out of the system.
int solve_damped_dev(lsqamd_fit *f, double mu, const double *diag_host, const double *frozen_host = nullptr);
// second right-hand side with the factor of the last solve_damped_dev still in place
int solve_with_factor(lsqamd_fit *f, const double *rhs, double *out);
// y = A x (one device GEMV on a dense copy of A)
int symv_host(lsqamd_fit *f, const double *x, double *y);
// f, J, A, g, D, mu, delta at p0 (gsl_multifit_nlinear_init); nfev = njev = 1
int do_init(lsqamd_fit *f, const double *p0);
void scale_update(lsqamd_fit *f);
double dot_h(const std::vector<double> &a, const std::vector<double> &b);
double scaled_norm(const std::vector<double> &d, const std::vector<double> &v);

// ---- scipy_least_squares' methods (scipy_methods.hip); *status_out in scipy's numbering --------
int run_trf(lsqamd_fit *f, const double *p0, int *status_out);
int run_dogbox(lsqamd_fit *f, const double *p0, int *status_out);
int run_minpack(lsqamd_fit *f, const double *p0, int *status_out);

}  // namespace lsqamd_host
