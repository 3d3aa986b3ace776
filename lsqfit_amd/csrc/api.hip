// C-ABI entry points + the host side of the trust-region LM driver (gfx950).
//
// The driver restates GSL's multifit_nlinear trust/lm/nielsen/scaling/convergence
// logic that src/lsqfit/_gsl.pyx:676-677 runs (init + driver), on the normal
// equations (solver='cholesky').  All O(N P), O(N P^2) and O(P^3) work is on the
// device; the host keeps only O(P) vectors (x, g, D, dx) and the scalars
// (mu, nu, rho), so that every rank of a row-sharded fit takes identical
// decisions from the identical all-reduced (J^T J, J^T f, chi2).  Nothing is uploaded
// inside a step (D and the trial point are mirrored / formed on the device); one stream
// synchronisation per trial step and one per Jacobian.  Also here: the other trust-region
// sub-problem solvers (lmaccel, dogleg, ddogleg, subspace2D), fit.p sensitivities
// (lsqamd_dpdy), chi2 at many points (lsqamd_chi2_points).  The fit handle and the building
// blocks shared with the scipy-plugin methods (scipy_methods.hip) are declared in fit_state.h.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include <sched.h>

#include "fit_state.h"
#include "jit.h"

// spin-wait hint of the host polling loops
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  __asm__ __volatile__("yield");
#else
  sched_yield();
#endif
}

using namespace lsqamd;

namespace {

constexpr int64_t ALIGN = 256;
constexpr int NRM_BLOCKS = 1024;   // workgroups of the fused normal-equation kernel (few parameters, jit.hip)
inline int64_t rup(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct Carver {
  char *base;
  size_t off = 0, cap;
  bool dry;
  Carver(void *b, size_t c, bool d) : base((char *)b), cap(c), dry(d) {}
  template <typename T>
  T *take(int64_t count) {
    const size_t bytes = (size_t)rup((int64_t)(count < 1 ? 1 : count) * (int64_t)sizeof(T), ALIGN);
    T *p = dry ? nullptr : reinterpret_cast<T *>(base + off);
    off += bytes;
    return p;
  }
};
}  // namespace

namespace lsqamd_host {

// ---- process-wide recycling of pinned blocks and events (see fit_state.h) -------------------------------------------
namespace {
struct Recycler {
  std::mutex mu;
  std::multimap<std::pair<int, size_t>, void *> pinned;   // (device, size class) -> block
  size_t pinned_bytes = 0;
  std::multimap<int, hipEvent_t> events;
  std::multimap<int, hipStream_t> streams;
};
Recycler &recycler() {
  static Recycler *r = new Recycler;    // never destroyed: no runtime calls from a static destructor at exit
  return *r;
}
int current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = -1; }
  return d;
}
constexpr size_t PIN_CLASS = 4096, PIN_KEEP_MAX = 64 << 10, PIN_KEEP_TOTAL = 2 << 20;
}  // namespace

// Hand-offs through polled pinned memory (wait_record, run_one_launch), over the process: snapshots that did not verify (yet),
// hand-offs served from device memory after the stream had drained, words the test knob LSQAMD_VERIFY_HANDOFF=1 found different
// from the device's own copy (must stay 0).  lsqamd_handoff_stats reads them.
std::atomic<int64_t> g_handoff[3];

// LSQAMD_POISON_PINNED=1 (test knob): every pinned block handed out, and every region a kernel is about to publish into, is
// filled with a recognisable NaN first -- a host that acts on words the device has not written yet then acts on NaNs, loudly
bool poison_pinned() {
  static const bool on = [] { const char *e = getenv("LSQAMD_POISON_PINNED"); return e && e[0] == '1'; }();
  return on;
}
void poison_fill(void *p, size_t bytes) {
  const uint64_t nan_bits = 0x7ff8dead0000beefULL;
  uint64_t *w = static_cast<uint64_t *>(p);
  for (size_t i = 0; i < bytes / 8; ++i) w[i] = nan_bits;
}

void *pinned_take(size_t bytes, size_t *granted) {
  const size_t cls = (bytes + PIN_CLASS - 1) / PIN_CLASS * PIN_CLASS;
  *granted = cls;
  const int dev = current_device();
  {
    Recycler &r = recycler();
    std::lock_guard<std::mutex> lk(r.mu);
    auto it = r.pinned.find({dev, cls});
    if (it != r.pinned.end()) {
      void *p = it->second;
      r.pinned.erase(it);
      r.pinned_bytes -= cls;
      if (poison_pinned()) poison_fill(p, cls);
      return p;
    }
  }
  void *p = nullptr;
  if (hipHostMalloc(&p, cls, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (poison_pinned()) poison_fill(p, cls);
  return p;
}

void pinned_give(void *p, size_t granted, int dev_taken) {
  if (!p) return;
  const int dev = dev_taken >= 0 ? dev_taken : current_device();
  if (granted > 0 && granted <= PIN_KEEP_MAX && dev >= 0) {
    Recycler &r = recycler();
    std::lock_guard<std::mutex> lk(r.mu);
    if (r.pinned_bytes + granted <= PIN_KEEP_TOTAL) {
      r.pinned.insert({{dev, granted}, p});
      r.pinned_bytes += granted;
      return;
    }
  }
  (void)hipHostFree(p);
}

hipEvent_t event_take() {
  const int dev = current_device();
  {
    Recycler &r = recycler();
    std::lock_guard<std::mutex> lk(r.mu);
    auto it = r.events.find(dev);
    if (it != r.events.end()) {
      hipEvent_t e = it->second;
      r.events.erase(it);
      return e;
    }
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

void event_give(hipEvent_t e, int dev_taken) {
  if (!e) return;
  const int dev = dev_taken >= 0 ? dev_taken : current_device();
  if (dev >= 0) {
    Recycler &r = recycler();
    std::lock_guard<std::mutex> lk(r.mu);
    if (r.events.size() < 256) {
      r.events.insert({dev, e});
      return;
    }
  }
  (void)hipEventDestroy(e);
}

hipStream_t stream_take() {
  const int dev = current_device();
  {
    Recycler &r = recycler();
    std::lock_guard<std::mutex> lk(r.mu);
    auto it = r.streams.find(dev);
    if (it != r.streams.end()) {
      hipStream_t s = it->second;
      r.streams.erase(it);
      return s;
    }
  }
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return s;
}

void stream_give(hipStream_t s, int dev_taken) {
  if (!s) return;
  const int dev = dev_taken >= 0 ? dev_taken : current_device();      // (filed under the device it was created on, whichever device the calling thread has current)
  if (dev >= 0) {
    Recycler &r = recycler();
    std::lock_guard<std::mutex> lk(r.mu);
    if (r.streams.size() < 16) {
      r.streams.insert({dev, s});
      return;
    }
  }
  (void)hipStreamDestroy(s);
}

// ---- host inputs of small problems: staged through pinned memory, uploaded without a wait -------------------------------
// A pageable source makes hipMemcpyAsync + the synchronisation the caller's buffer needs cost 14-20 us per setter -- four of
// them were a fifth of a whole small fit.  Small inputs (<= 16 KB each, 64 KB between two drains of the stream) are copied
// into a pinned arena the handle owns and uploaded from there: the setter returns at once, the stream orders the rest.
struct Upload {
  lsqamd_fit *f;
  bool waited_for = false;      // something went the direct way: the caller must synchronise before it returns
  explicit Upload(lsqamd_fit *fit) : f(fit) {}
  static bool is_host(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return true; }   // unregistered host memory
    return a.type != hipMemoryTypeDevice && a.type != hipMemoryTypeManaged;
  }
  hipError_t operator()(void *dst, const void *src, size_t bytes, bool maybe_device = false) {
    constexpr size_t ARENA = 64 << 10, ONE = 16 << 10;
    if (bytes == 0) return hipSuccess;
    if (bytes <= ONE && (!maybe_device || is_host(src))) {
      if (!f->stage) {
        f->stage = static_cast<char *>(pinned_take(ARENA, &f->stage_bytes));
        f->stage_off = 0;
      }
      if (f->stage) {
        const size_t need = (bytes + 63) & ~(size_t)63;
        if (f->stage_off + need > f->stage_bytes) {       // arena full: drain the stream, start over
          hipError_t e = hipStreamSynchronize(f->st);
          if (e != hipSuccess) return e;
          f->stage_off = 0;
        }
        std::memcpy(f->stage + f->stage_off, src, bytes);
        hipError_t e = hipMemcpyAsync(dst, f->stage + f->stage_off, bytes, hipMemcpyHostToDevice, f->st);
        f->stage_off += need;
        return e;
      }
    }
    waited_for = true;
    return hipMemcpyAsync(dst, src, bytes, maybe_device ? hipMemcpyDefault : hipMemcpyHostToDevice, f->st);
  }
  hipError_t finish() { return waited_for ? hipStreamSynchronize(f->st) : hipSuccess; }
};

hipEvent_t take_event(lsqamd_fit *f) {  // events are recycled: creating one costs ~10 us
  hipEvent_t e = nullptr;
  if (!f->event_pool.empty()) {
    e = f->event_pool.back();
    f->event_pool.pop_back();
  } else {
    e = event_take();
  }
  return e;
}


// Between the steps of a fit the pairs just pile up: reading them back costs ~100 us of host time per step (18 event
// queries), all of it between the arrival of a step's record and the first launch of the next -- measured as GPU idle
// time with rocprofv3 at the 8-GPU shard shape.  They are read when somebody asks (lsqamd_get_timers / _reset /
// _finish) or when a few hundred have piled up.
void resolve_timers(lsqamd_fit *f, bool only_if_many = false) {
  if (only_if_many) {
    size_t n = 0;
    for (auto &t : f->timers) n += t.pending.size();
    if (n < 512) return;
  }
  for (auto &t : f->timers) {      // intervals another timer owns the events of: before those events are recycled below
    for (auto &pr : t.pending_shared) {
      (void)hipEventSynchronize(pr.second);
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
        t.total_ms += ms;
        t.count += 1;
      }
    }
    t.pending_shared.clear();
  }
  for (auto &t : f->timers) {
    for (auto &pr : t.pending) {
      (void)hipEventSynchronize(pr.second);
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
        t.total_ms += ms;
        t.count += 1;
      }
      f->event_pool.push_back(pr.first);
      f->event_pool.push_back(pr.second);
    }
    t.pending.clear();
  }
}

int32_t choose_splits(int64_t N, int64_t P) {
  if (const char *e = getenv("LSQAMD_SYRK_SPLITS")) {  // tuning override (developer knob)
    const int v = atoi(e);
    if (v >= 1 && v <= 64) return v;
  }
  const int64_t T = (P + 127) / 128;
  const int64_t tiles = T * (T + 1) / 2;
  // enough workgroups to fill the chip a few times over (>= 2048: four rounds of two per CU), K-chunks of
  // ~4096 rows when there are that many rows (tail balance: 16 chunks beat 8 at N = 65536), never below 256
  // rows, never more than 16 slabs.  Their read-back in finalize_pack grows with the count: at the 8-GPU
  // shard shape N = 8192, P = 4096 (528 tiles) the product itself takes 3.07 / 2.53 / 2.24 / 2.10 / 2.10 ms
  // with 1 / 2 / 3 / 4 / 8 chunks and the read-back 0.06 / 0.07 / 0.09 / 0.10 / 0.16 ms: four.
  int64_t s = (2048 + tiles - 1) / tiles;
  if (s < N / 4096) s = N / 4096;
  // (few tiles: chunks down to 64 rows -- at (4096, 256) three tiles times 16 chunks are 48 workgroups on 256 CUs and the
  // product takes 40 us; 64 chunks: 21 us, the longer read-back of the slabs included in the balance below)
  const int64_t minrows = tiles <= 10 ? 64 : 256;
  const int64_t maxs = N / minrows > 1 ? N / minrows : 1;
  if (s > maxs) s = maxs;
  // ... except for FEW parameters and many rows (the usual shape of a fit): one or three tiles times 16 chunks
  // leaves 240 of the 256 CUs idle (N = 65536, P = 64: 0.62 ms for a product that reads 34 MB).  Slabs are
  // small there, so the count may grow to 256 as long as they stay within 64 MB together.
  const int64_t ldm = (P % 128 == 0) ? P + 128 : (P + 1 + 15) / 16 * 16;
  int64_t cap = (int64_t)(64 << 20) / (8 * P * ldm);
  if (cap > 256) cap = 256;
  if (cap < 16) cap = 16;
  if (s > cap) s = cap;
  // ... and for MANY rows at a few dozen tiles 16 chunks are one round of workgroups and a bit ((131072, 1024):
  // 576 workgroups, 2.81 ms; 32 chunks 2.26 ms): chunks of 4096 rows up to 32 of them, while the slabs stay
  // below a quarter of the Jacobian's own size
  int64_t more = N / 4096 < 32 ? N / 4096 : 32;
  const int64_t room = (N * P / 4) / (P * ldm);
  if (more > room) more = room;
  if (tiles >= 36 && s < more) s = more;
  if (s < 1) s = 1;
  return (int32_t)s;
}

size_t carve(lsqamd_fit *f, void *ws, size_t cap, bool dry) {
  const lsqamd_config &c = f->cfg;
  const int64_t N = c.n_data, P = c.n_param;
  f->N = N;
  f->P = P;
  f->ld = rup(P + 1, 16);
  // damped matrix [A | g | zero pad]: with P a multiple of 128 the pad makes every Cholesky
  // panel / trailing GEMM full-tile, so they run the direct-to-LDS interior kernel
  f->ldm = (P % 128 == 0) ? P + 128 : rup(P + 1, 16);
  f->ncols_aug = (P % 128 == 0) ? P + 128 : P + 1;
  f->npk = packed_doubles(P);
  f->splits = choose_splits(N, P);
  // row chunks of the two-stage J^T f: 256, more for many rows and few parameters (a chunk of 2048 rows is a
  // serial walk per thread: (524288, 64) took 0.25 ms for 268 MB) while the partial sums stay within 8 MB
  f->npartial = 256;
  while (f->npartial < 4096 && 2 * f->npartial <= N / 64 && 2 * f->npartial * (P + 1) * 8 <= (int64_t)(8 << 20))
    f->npartial *= 2;
  Carver cv(ws, cap, dry);
  f->x = cv.take<double>(N * (c.n_x > 0 ? c.n_x : 1));
  f->ymean = cv.take<double>(N);
  f->wdiag = cv.take<double>(N);
  f->in_block = cv.take<uint8_t>(N);
  f->row_param = cv.take<int32_t>(N);
  f->blk_row0 = cv.take<int64_t>(c.n_blocks);
  f->blk_size = cv.take<int64_t>(c.n_blocks);
  f->blk_woff = cv.take<int64_t>(c.n_blocks);
  f->wt = cv.take<double>(c.sum_block_sq);
  f->prior_mean = cv.take<double>(P);
  f->prior_prec = cv.take<double>(c.prior_dense ? P * P : P);
  f->p_dev = cv.take<double>(P);
  f->p_trial = cv.take<double>(P);
  f->p_buf0 = f->p_dev;
  f->r = cv.take<double>(N);
  f->r_raw = cv.take<double>(c.n_blocks > 0 ? N : 1);
  f->J = cv.take<double>(N * f->ld);
  f->Jraw = cv.take<double>(c.n_blocks > 0 ? N * f->ld : 1);
  f->slabs = cv.take<double>((int64_t)f->splits * P * f->ldm);
  f->redbuf = cv.take<double>(f->npk + P + 1);
  f->red_scalar = cv.take<double>(8);
  f->M = cv.take<double>(P * f->ldm);
  f->chol_work = cv.take<double>((int64_t)(potrf_work_bytes(P) / sizeof(double)));
  f->yv = cv.take<double>(4 * P + 8);   // y, v, then the hand-off granules of the chained back substitution
  f->diag_dev = cv.take<double>(P);
  f->dscale = cv.take<double>(P);
  f->tvec = cv.take<double>(P + 1);
  const int64_t part = f->npartial * (P + 1);
  f->partial = cv.take<double>(part > 2048 ? part : 2048);
  if (c.model == LSQAMD_MODEL_TAPE && P <= lsqamd_jit::NRM_MAX_P) f->nrm_part = cv.take<double>(NRM_BLOCKS * 96 + 128);
  f->Wl = cv.take<double>(P * f->ldm);
  f->cov = cv.take<double>(P * f->ldm);
  f->scal = cv.take<double>(16);
  f->lmd = cv.take<double>(LMS_COUNT);
  if (c.model == LSQAMD_MODEL_TAPE && P <= lsqamd_jit::FIT_MAX_P) f->fit_block = cv.take<double>(lsqamd_jit::FIT_HOST_DOUBLES);
  f->info_dev = cv.take<int32_t>(16);
  f->trig_far = cv.take<int32_t>(16);
  f->tape_cap = c.tape_len > 1024 ? c.tape_len : 1024;
  f->tape = cv.take<int32_t>(f->tape_cap);
  f->tape_poff = cv.take<int32_t>(f->tape_cap + 1);
  f->tape_seg = cv.take<int32_t>(3 * (int64_t)f->tape_cap + 3);
  f->consts = cv.take<double>(1024);
  if (c.model == LSQAMD_MODEL_TAPE) {
    // one forward + one reverse sweep per row: per resident wave 2 slots per instruction (upper
    // bound) x 64 lanes of local partial derivatives; at most 1024 waves are resident
    const int64_t groups = (N + 63) / 64;
    f->tape_wgs = (groups + 3) / 4 < 256 ? (groups + 3) / 4 : 256;
    if (f->tape_wgs < 1) f->tape_wgs = 1;
    f->tape_slot_cap = 2 * f->tape_cap;
    f->tape_part = cv.take<double>(f->tape_wgs * 4 * (int64_t)f->tape_slot_cap * 64);
    f->tape_ldn = rup(N > 0 ? N : 1, 64);
    f->tape_jt = cv.take<double>((P + 1) * f->tape_ldn);
  }
  f->syrk_nwork = (int32_t)syrk_work_count(P, f->splits);
  f->syrk_map = cv.take<int32_t>(4 * (int64_t)f->syrk_nwork);
  f->syrk_map_g = cv.take<int32_t>(4 * (int64_t)f->syrk_nwork);   // the same entries grouped by tile rows (grouped exchange)
  f->xg_ctr = cv.take<int32_t>(16);
  return cv.off;
}

int check_cfg(const lsqamd_config *c) {
  if (!c || c->abi_version != LSQAMD_ABI_VERSION) return LSQAMD_EINVAL;
  if (c->n_data < 0 || c->n_param < 1 || c->n_blocks < 0) return LSQAMD_EINVAL;
  if (c->model < LSQAMD_MODEL_COSMIX || c->model > LSQAMD_MODEL_IDENTITY) return LSQAMD_EINVAL;
  if (c->model == LSQAMD_MODEL_TAPE && c->n_param > LSQAMD_TAPE_MAX_PARAM) return LSQAMD_EINVAL;
  if (c->tape_len < 0 || c->tape_len > LSQAMD_TAPE_MAX_CODE) return LSQAMD_EINVAL;
  if ((c->model == LSQAMD_MODEL_COSMIX || c->model == LSQAMD_MODEL_MULTIEXP) && (c->n_param & 1))
    return LSQAMD_EINVAL;
  if (c->n_batch > 1) return LSQAMD_EUNSUPPORTED;
  return 0;
}

ModelArgs model_args(const lsqamd_fit *f, const double *p) {
  ModelArgs m;
  m.model = f->cfg.model;
  m.n_data = f->N;
  m.n_param = f->P;
  m.n_x = f->cfg.n_x > 0 ? f->cfg.n_x : 1;
  m.x = f->x;
  m.ymean = f->ymean;
  m.wdiag = f->wdiag;
  m.in_block = f->cfg.n_blocks > 0 ? f->in_block : nullptr;
  m.p = p;
  m.tape = f->tape;
  m.n_tape = f->n_tape;
  m.consts = f->consts;
  if (f->tape_part) {
    m.tape_poff = f->tape_poff;
    m.tape_part = f->tape_part;
    m.tape_jt = f->tape_jt;
    m.tape_ldn = f->tape_ldn;
    m.tape_wgs = f->tape_wgs;
    m.tape_slots = f->tape_slots;
    m.tape_seg = f->tape_seg;
    m.tape_n_seg = f->tape_n_seg;
    m.tape_seg_depth = f->tape_seg_depth;
    m.tape_seg_slots = f->tape_seg_slots;
    m.tape_single = f->tape_single;
    m.tape_slot_cap = f->tape_slot_cap;
  }
  m.jit = f->jit;
  if (!f->progs.empty()) {
    m.progs = f->progs.data();
    m.n_prog = (int32_t)f->progs.size();
  }
  return m;
}

// the library does not switch devices: the device that was current at lsqamd_create must be current in the calling thread
// (a launch on a stream of another device fails -- or, worse, a recycled block of another device is touched)
int device_ok(lsqamd_fit *f) {
  const int dev = current_device();
  if (f->dev >= 0 && dev != f->dev)
    FAIL(f, LSQAMD_EINVAL, "this handle was created on device %d, the calling thread's current device is %d (hipSetDevice first)", f->dev, dev);
  return 0;
}

int ready(lsqamd_fit *f) {
  if (const int rc = device_ok(f)) return rc;
  if (!f->have_data) FAIL(f, LSQAMD_EINVAL, "lsqamd_set_data has not been called");
  if (f->cfg.has_prior && !f->have_prior) FAIL(f, LSQAMD_EINVAL, "lsqamd_set_prior has not been called");
  if (f->cfg.model != LSQAMD_MODEL_IDENTITY && !f->have_x) FAIL(f, LSQAMD_EINVAL, "lsqamd_set_x has not been called");
  if (f->cfg.model == LSQAMD_MODEL_TAPE && !f->have_tape) FAIL(f, LSQAMD_EINVAL, "lsqamd_set_tape has not been called");
  return 0;
}

int do_reduce(lsqamd_fit *f, double *buf, int64_t count) {
  if (f->comm) {   // RCCL on the handle's stream: nothing to wait for on the host
    Scope sc(f, LSQAMD_T_REDUCE);
    Scope sc2(f, LSQAMD_T_EXCH_COLL, nullptr, LSQAMD_T_EXCH_WAIT);   // on the step's own stream: ONE interval, all of it waited for
    return comm_all_reduce(f, buf, count);
  }
  if (!f->reduce) return 0;
  Scope sc(f, LSQAMD_T_REDUCE);
  HIPCHK(f, hipStreamSynchronize(f->st));
  const int rc = f->reduce(f->reduce_user, buf, count);
  if (rc != 0) FAIL(f, LSQAMD_EREDUCE, "all-reduce hook returned %d", rc);
  return 0;
}

// two extra launches per evaluation (the range kernel and the variant that finds nothing to do, ~5 us together)
// buy a trig kernel without far-range code: worth it from a few million cosines per evaluation on
static bool trig_split_pays(const lsqamd_fit *f) { return f->N * (f->P / 2) >= ((int64_t)1 << 22); }

// whitened residual VECTOR at device parameters p -> f->r (model kernel + block whitening); launches only
int residual_vector_launch(lsqamd_fit *f, const double *p) {
  ModelArgs m = model_args(f, p);
  if (f->cfg.model == LSQAMD_MODEL_COSMIX && f->have_x && trig_split_pays(f)) {
    // cosine model: the range flag of THIS point (the frequencies are parameters P/2 .. P - 1); the fused
    // whitening kernel, which only ever runs at the point whose residual is current, reads the same flag
    HIPCHK(f, launch_trig_range(f->st, p + f->P / 2, f->P / 2, f->xmax, f->trig_far));
    m.trig_far = f->trig_far;
  }
  HIPCHK(f, launch_residual_ex(f->st, m, f->r, f->r_raw));
  if (f->have_param_rows)
    HIPCHK(f, launch_param_rows(f->st, f->row_param, f->N, f->P, 1, p, f->ymean, f->wdiag,
                                f->cfg.n_blocks > 0 ? f->in_block : nullptr, f->r, f->r_raw, 0));
  if (f->cfg.n_blocks > 0) {
    // small blocks: one thread per output row; large ones (a single 8192-row block would keep
    // 32 workgroups busy for milliseconds): r_b = Wt_b^T delta_b as a two-stage column sum at
    // HBM speed, staged through the raw-Jacobian buffer (idle during a residual evaluation)
    constexpr int64_t BIG = 1024;
    HIPCHK(f, launch_block_whiten_vec(f->st, f->wt, f->blk_row0, f->blk_size, f->blk_woff,
                                      f->cfg.n_blocks, f->cfg.max_block, f->r_raw, f->r, 1, 0, nullptr, BIG));
    for (size_t b = 0; b < f->h_size.size(); ++b) {
      const int64_t B = f->h_size[b];
      if (B < BIG) continue;
      int64_t nch = (f->N * f->ld) / B;
      if (nch > 256) nch = 256;
      HIPCHK(f, launch_colsum_dot(f->st, f->wt + f->h_woff[b], B, B, B, 0, f->Jraw, nch,
                                  f->r + f->h_row0[b], f->r_raw + f->h_row0[b], 1, f->h_tri[b] ? 1 : 0));
    }
  }
  return 0;
}

// whitened residual at device parameters p -> f->r; chi2 (summed over ranks) -> f->red_scalar[0].
// Launches only: nothing is waited for.
// decide: a single rank's trial -- the sum of squares' second stage, the prior's share and the LM decision in
// one launch (lm_trial_tail_kernel); with ranks to sum over the scalar is exchanged first and the caller decides
static bool small_fuse(const lsqamd_fit *f);
int eval_residual_launch(lsqamd_fit *f, const double *p, bool decide = false) {
  f->r_fresh = false;
  if (decide) {
    Scope sc(f, LSQAMD_T_RESIDUAL);
    const int rc = residual_vector_launch(f, p);
    if (rc) return rc;
    const bool wp = f->cfg.has_prior && f->adds_prior;
    if (small_fuse(f) && f->N <= 65536 && !(wp && f->cfg.prior_dense)) {
      HIPCHK(f, launch_lm_trial_tail_small(f->st, f->r, f->N, f->P, wp ? f->prior_prec : nullptr, f->prior_mean, p, f->tvec,
                                           f->red_scalar, f->info_dev, f->opt.factor_up, f->opt.factor_down, f->lmd));
      return 0;
    }
    HIPCHK(f, launch_lm_trial_tail(f->st, f->r, f->N, f->partial, f->P, f->prior_prec, f->cfg.prior_dense,
                                   f->prior_mean, p, f->tvec, f->cfg.has_prior && f->adds_prior, f->red_scalar,
                                   f->info_dev, f->opt.factor_up, f->opt.factor_down, f->lmd));
    return 0;
  }
  {
    Scope sc(f, LSQAMD_T_RESIDUAL);
    const int rc = residual_vector_launch(f, p);
    if (rc) return rc;
    if (f->robust())    // scipy's loss_function(f, cost_only=True), as 2 x cost
      HIPCHK(f, launch_robust_cost(f->st, f->r, f->N, 1, f->loss, f->f_scale, f->partial, f->red_scalar));
    else
      HIPCHK(f, launch_sumsq(f->st, f->r, f->N, f->partial, f->red_scalar));
    if (f->cfg.has_prior && f->adds_prior)
      HIPCHK(f, launch_prior_chi2(f->st, f->P, f->prior_prec, f->cfg.prior_dense, f->prior_mean, p,
                                  f->tvec, f->red_scalar));
  }
  return do_reduce(f, f->red_scalar, 1);
}

// whitened residual at device parameters p -> f->r ; chi2 (all ranks) on return
int eval_residual_dev(lsqamd_fit *f, const double *p, double *chi2_out) {
  int rc = eval_residual_launch(f, p);
  if (rc) return rc;
  HIPCHK(f, hipMemcpyAsync(f->pin_s, f->red_scalar, sizeof(double), hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipStreamSynchronize(f->st));
  *chi2_out = f->pin_s[0];
  f->nfev++;
  return 0;
}

// Whitening of the block rows of the Jacobian (and of its residual column).  *fused_chunks > 0 on
// return: the bulk GEMM also left per-tile-row partial sums of J^T f in f->slabs
// (fused_chunks x P, to be reduced before the SYRK reuses the slabs).
int whiten_jacobian(lsqamd_fit *f, int64_t *fused_chunks, bool r_here, bool plain = false) {   // plain: no J^T f out of the product's epilogue (the rows are rescaled afterwards)
  *fused_chunks = 0;
  const int nb = f->cfg.n_blocks;
  if (nb <= 0) return 0;
  Scope sc(f, LSQAMD_T_WHITEN);
  const int64_t ncols = f->P + 1;
  if (f->uniform_blocks) {
    const int64_t B = f->h_size[0];
    GemmTN g;
    g.X = f->wt; g.ldx = B; g.sx = B * B;
    g.Y = f->Jraw + f->h_row0[0] * f->ld; g.ldy = f->ld; g.sy = B * f->ld;
    g.C = f->J + f->h_row0[0] * f->ld; g.ldc = f->ld; g.sc = B * f->ld;
    g.M = B; g.N = ncols; g.K = B;
    g.batch = nb;
    g.x_upper_tri = f->uniform_tri;
    if (f->P % 128 == 0 && B % 128 == 0) {
      // full-tile bulk (interior kernel) + the residual column on its own, the column FIRST: the
      // bulk's epilogue can then form its share of J^T f from the tile it still holds
      const int64_t slab_doubles = (int64_t)f->splits * f->P * f->ldm;
      bool col_done = false;
      if (B >= 1024) {
        // the residual column of a large block is a GEMV: two-stage column sum at HBM speed
        // (as a one-column GEMM it would keep B/128 workgroups busy for over a millisecond)
        int64_t nch = slab_doubles / B;
        if (nch > 256) nch = 256;
        if (nch >= 1) {
          for (int b = 0; b < nb; ++b) {
            const int64_t r0 = f->h_row0[b];
            if (!r_here)   // (else f->r already holds the whitened residual of this point: the trial evaluation left it)
              HIPCHK(f, launch_colsum_dot(f->st, f->wt + f->h_woff[b], B, B, B, 0, f->slabs, nch, f->r + r0,
                                          f->Jraw + r0 * f->ld + f->P, f->ld, f->uniform_tri ? 1 : 0));
            HIPCHK(f, launch_copy_strided(f->st, f->r + r0, 1, f->J + r0 * f->ld + f->P, f->ld, B, 1));
          }
          col_done = true;
        }
      }
      if (!col_done) {
        GemmTN c = g;
        c.Y += f->P; c.C += f->P; c.N = 1;
        HIPCHK(f, launch_gemm_tn(f->st, c));
      }
      g.N = f->P;
      const int64_t chunks = (int64_t)nb * (B / 128);
      if (!plain && f->h_row0[0] == 0 && (int64_t)nb * B == f->N && chunks * f->P <= slab_doubles) {
        g.colsum_out = f->slabs;
        g.colsum_ld = f->P;
        g.colsum_rcol = f->P;
        if (gemm_tn_fuses_colsum(g)) *fused_chunks = chunks;
        else g.colsum_out = nullptr;
      }
      // one (or a few) large blocks: too few workgroups for the chip -- both halves of every tile row's K-range at once, the
      // upper halves into the slabs (free until the J^T J launch), added below; J^T f then comes from that launch's
      // diagonal tiles instead of this one's epilogue
      bool halves = false;
      if ((int64_t)nb * B * f->ld <= slab_doubles && f->h_row0[0] == 0 && (int64_t)nb * B == f->N) {
        GemmTN h = g;
        h.colsum_out = nullptr;
        h.tri_halves = 1;
        h.split_stride = f->slabs - g.C;
        if (gemm_tn_wants_tri_halves(h)) {
          g = h;
          *fused_chunks = 0;
          halves = true;
        }
      }
      HIPCHK(f, launch_gemm_tn(f->st, g));
      if (halves) HIPCHK(f, launch_add_rows(f->st, f->J, f->slabs, f->N, f->P, f->ld));
      if (*fused_chunks > 0) {
        // J^T f = sum of the per-tile-row pieces; |f|^2 from the residual column
        double *gvec = f->redbuf + f->npk;
        HIPCHK(f, launch_colsum_reduce(f->st, f->slabs, *fused_chunks, f->P, gvec));
        HIPCHK(f, launch_copy_strided(f->st, f->J + f->P, f->ld, f->r, 1, f->N, 1));
        HIPCHK(f, launch_sumsq(f->st, f->r, f->N, f->partial, gvec + f->P));
      }
      return 0;
    }
    HIPCHK(f, launch_gemm_tn(f->st, g));
    return 0;
  }
  for (int b = 0; b < nb; ++b) {
    const int64_t B = f->h_size[b];
    GemmTN g;
    g.X = f->wt + f->h_woff[b]; g.ldx = B;
    g.Y = f->Jraw + f->h_row0[b] * f->ld; g.ldy = f->ld;
    g.C = f->J + f->h_row0[b] * f->ld; g.ldc = f->ld;
    g.M = B; g.N = ncols; g.K = B;
    g.x_upper_tri = f->h_tri[b];
    HIPCHK(f, launch_gemm_tn(f->st, g));
  }
  return 0;
}

// J, J^T J (packed), J^T f, chi2 at device parameters p; host g/chi2/colnorm refreshed
int eval_normal_dev(lsqamd_fit *f, const double *p, bool mirror) {
  const int64_t P = f->P;
  int64_t fused_chunks = 0;
  int rc = 0;
  const bool r_here = f->r_fresh && f->r_ptr == p;
  f->r_fresh = false;
  const int nbk = f->cfg.n_blocks;
  const int64_t B0 = nbk > 0 ? f->h_size[0] : 0;
  const int64_t slab_doubles = (int64_t)f->splits * P * f->ldm;
  // few parameters, uncorrelated rows, a compiled formula: J^T J, J^T f and chi2 straight from the model kernel's registers
  // (jit.hip lsqamd_jit_nrm) -- the Jacobian is never written, the evaluation reads x, y, w once.  Device-resident LM only
  // (the host-side drivers and the getters call ensure_J() when they need the rows).  LSQAMD_FUSED_NORMAL=0 disables.
  static const bool nrm_off = [] { const char *e = getenv("LSQAMD_FUSED_NORMAL"); return e && e[0] == '0'; }();
  const bool robust = f->robust();     // rows rescaled by the loss before the products below see them: the unfused route
  const int nq = (!mirror && !robust && !nrm_off && f->nrm_part && f->progs.empty() && nbk == 0 && !f->have_param_rows && f->N > 0)
                     ? lsqamd_jit::normal_nq(static_cast<const lsqamd_jit::Kernel *>(f->jit)) : 0;
  if (nq > 0) {
    double *gv = f->redbuf + f->npk;
    const bool with_prior = f->cfg.has_prior && f->adds_prior;
    {
      Scope sc(f, LSQAMD_T_JACOBIAN);
      int64_t blocks = (f->N + 255) / 256;
      if (blocks > NRM_BLOCKS) blocks = NRM_BLOCKS;
      lsqamd_jit::LaunchArgs la;
      la.x = f->x; la.p = p; la.ymean = f->ymean; la.wdiag = f->wdiag; la.n_data = f->N;
      HIPCHK(f, lsqamd_jit::launch_normal(static_cast<const lsqamd_jit::Kernel *>(f->jit), f->st, la, f->nrm_part, (int)blocks));
      // few rows as well, one rank: totalling the sums and unpacking them is left to the accept-tail kernel (one launch
      // instead of three); the prior's share of g and chi2 then has to be deferred to it too
      f->nrm_in_tail = blocks <= 64 && small_fuse(f) && (!with_prior || r_here) ? (int)blocks : 0;
      if (!f->nrm_in_tail) {
        double *tot = f->nrm_part + (int64_t)NRM_BLOCKS * 96;
        HIPCHK(f, launch_colsum_reduce(f->st, f->nrm_part, blocks, nq, tot));
        HIPCHK(f, launch_nrm_unpack(f->st, tot, P, f->redbuf, gv, with_prior ? f->prior_prec : nullptr, f->cfg.prior_dense));
      }
    }
    f->J_stale = true;
    f->used_nrm = true;
    {
      Scope sc(f, LSQAMD_T_GRAD);
      f->prior_deferred = with_prior && r_here && small_fuse(f);
      if (with_prior && !f->prior_deferred)
        HIPCHK(f, launch_add_prior(f->st, f->redbuf, P, f->prior_prec, f->cfg.prior_dense, f->prior_mean, p, f->tvec, gv, 0,
                                   r_here ? 1 : 0));
    }
    rc = do_reduce(f, f->redbuf, f->npk + P + 1);
    if (rc) return rc;
    f->have_cov = false;
    f->have_dense_A = false;
    f->njev++;
    f->mirrors_stale = true;
    return 0;
  }
  f->J_stale = false;
  f->nrm_in_tail = 0;
  f->cov_unavailable = false;
  if (!robust && nbk > 0 && f->uniform_blocks && f->uniform_tri && !f->have_param_rows && f->cfg.n_x <= 1 &&
      f->h_row0[0] == 0 && (int64_t)nbk * B0 == f->N && whiten_synth_eligible(f->cfg.model, B0, P) &&
      (int64_t)nbk * (B0 / 128) * P <= slab_doubles) {
    // the raw Jacobian rows are synthesised inside the whitening product (never written, never re-read);
    // the whitened residual -- column P, and the weights of the fused J^T f -- is the one the trial
    // evaluation just left in f->r when this IS the trial point, else it is evaluated here
    {
      Scope sc(f, LSQAMD_T_JACOBIAN);
      if (!r_here) {
        rc = residual_vector_launch(f, p);
        if (rc) return rc;
      }
      HIPCHK(f, launch_copy_strided(f->st, f->r, 1, f->J + P, f->ld, f->N, 1));
    }
    Scope sc(f, LSQAMD_T_WHITEN);
    WhitenSynth w;
    w.model = f->cfg.model; w.Wt = f->wt; w.x = f->x; w.p = p; w.J = f->J; w.ld = f->ld;
    w.B = B0; w.K = P / 2; w.nb = nbk; w.colsum_out = f->slabs;
    w.trig_far = (f->cfg.model == LSQAMD_MODEL_COSMIX && f->have_x && trig_split_pays(f)) ? f->trig_far : nullptr;   // current: set with the residual at p
    HIPCHK(f, launch_whiten_synth(f->st, w));
    f->used_synth = true;
    fused_chunks = (int64_t)nbk * (B0 / 128);
    double *gv = f->redbuf + f->npk;
    HIPCHK(f, launch_colsum_reduce(f->st, f->slabs, fused_chunks, P, gv));
    HIPCHK(f, launch_sumsq(f->st, f->r, f->N, f->partial, gv + P));
  } else {
    {
      Scope sc(f, LSQAMD_T_JACOBIAN);
      ModelArgs m = model_args(f, p);
      HIPCHK(f, launch_jacobian_ex(f->st, m, f->J, f->Jraw, f->ld));
      if (f->have_param_rows)
        HIPCHK(f, launch_param_rows(f->st, f->row_param, f->N, f->P, f->ld, p, f->ymean, f->wdiag,
                                    f->cfg.n_blocks > 0 ? f->in_block : nullptr, f->J, f->Jraw, 1));
    }
    rc = whiten_jacobian(f, &fused_chunks, r_here, robust);
    if (rc) return rc;
    if (robust) {
      // scipy's scale_for_robust_loss_function on the whitened rows [J | f] (robust.hip); the cost of this point
      // (red_scalar[2], put in chi2's place once J^T f has been formed) from the residual column before it is rescaled
      HIPCHK(f, launch_robust_cost(f->st, f->J + P, f->N, f->ld, f->loss, f->f_scale, f->partial, f->red_scalar + 2));
      HIPCHK(f, launch_robust_scale_rows(f->st, f->J, f->N, P, f->ld, f->loss, f->f_scale));
    }
  }
  bool syrk_colsum = false;
  // the exchange in groups (sharded fits with the library's communicator): launch g forms the tile rows of
  // group g, its slab sum packs them, and their sum over the ranks runs on the exchange stream while launch g + 1 computes
  const bool grouped = f->comm && f->xg > 1 && P > 64 && f->N > 0;
  GemmTN g;
  g.X = f->J; g.Y = f->J; g.ldx = g.ldy = f->ld;
  g.C = f->slabs; g.ldc = f->ldm;
  g.M = P; g.N = P; g.K = f->N;
  g.upper_only = 1;
  g.splits = f->splits;
  g.split_stride = P * f->ldm;
  if (P > 64) {   // (one tile: nothing to order, and the 64 x 64 kernel takes no work list)
    g.work_map = f->syrk_map;
    g.n_work = f->syrk_nwork;
  }
  if (fused_chunks == 0 && f->N > 0) {   // J^T f and chi2 out of the diagonal tiles of this launch, when it is that kernel
    g.colsum_out = f->partial;            // [splits][P + 1] (splits <= 256 <= npartial)
    g.colsum_ld = P + 1;
    g.colsum_rcol = P;
    if (gemm_tn_fuses_colsum(g)) syrk_colsum = true;
    else g.colsum_out = nullptr;
  }
  double *gvec = f->redbuf + f->npk;
  const bool with_prior = f->cfg.has_prior && f->adds_prior;
  auto finish_gvec = [&]() -> int {       // J^T f, chi2 (+ the prior's share) into gvec: after the LAST product launch
    if (syrk_colsum)
      HIPCHK(f, launch_colsum_reduce(f->st, f->partial, f->splits, P + 1, gvec));
    else if (fused_chunks == 0)
      HIPCHK(f, launch_colsum_dot(f->st, f->J, f->N, f->ld, P + 1, P, f->partial, f->npartial, gvec));
    if (robust) HIPCHK(f, launch_copy_strided(f->st, f->red_scalar + 2, 1, gvec + P, 1, 1, 1));   // 2 x cost, not |f_scaled|^2
    f->prior_deferred = with_prior && r_here && !mirror && small_fuse(f);
    if (with_prior && !f->prior_deferred)   // (r_here: the trial evaluation at this point left Lambda (p - pbar) in tvec)
      HIPCHK(f, launch_add_prior(f->st, f->redbuf, P, f->prior_prec, f->cfg.prior_dense,
                                 f->prior_mean, p, f->tvec, gvec, 0, r_here ? 1 : 0));
    return 0;
  };
  if (grouped) {
    if (!f->xst) f->xst = stream_take();
    for (int q = 0; q < f->xg; ++q) {
      if (!f->xg_ready[q]) f->xg_ready[q] = event_take();
      if (!f->xg_done[q]) f->xg_done[q] = event_take();
    }
    if (!f->xst) FAIL(f, LSQAMD_EHIP, "no stream for the grouped exchange");
    if (f->xg_signal) {
      // ONE product launch; the exchange stream follows it group by group through the counters
      HIPCHK(f, hipMemsetAsync(f->xg_ctr, 0, 8 * sizeof(int32_t), f->st));
      HIPCHK(f, hipEventRecord(f->xg_ready[0], f->st));          // "the counters are zero": nothing on xst reads them earlier
      HIPCHK(f, hipStreamWaitEvent(f->xst, f->xg_ready[0], 0));
      {
        Scope sc(f, LSQAMD_T_SYRK);
        GemmTN gq = g;
        gq.work_map = f->syrk_map_g;
        gq.n_work = f->syrk_nwork;
        static const bool dbg_nosig = [] { const char *e = getenv("LSQAMD_EXCHANGE_DEBUG"); return e && e[0] == 'n'; }();   // developer knob: the grouped ORDER with the plain kernel
        gq.done_ctr = dbg_nosig ? nullptr : f->xg_ctr;
        HIPCHK(f, launch_gemm_tn(f->st, gq));
      }
      {
        Scope sc(f, LSQAMD_T_GRAD);
        rc = finish_gvec();
        if (rc) return rc;
      }
      HIPCHK(f, hipEventRecord(f->xg_ready[1], f->st));          // "[J^T f | chi2] is final" (goes out with the last group)
      static const int dbg_seq = [] { const char *e = getenv("LSQAMD_EXCHANGE_DEBUG"); return e ? (e[0] == 's' ? 1 : (e[0] == 'n' ? 2 : 0)) : 0; }();   // developer knob: nothing on the exchange stream before the product is done
      if (dbg_seq) HIPCHK(f, hipStreamWaitEvent(f->xst, f->xg_ready[1], 0));
      for (int q = 0; q < f->xg; ++q) {
        const bool last = q == f->xg - 1;
        if (dbg_seq != 2) HIPCHK(f, launch_wait_counter(f->xst, f->xg_ctr + q, f->xg_count[q], f->xg_ctr + 8));
        {
          Scope sc(f, LSQAMD_T_GRAD, f->xst);
          HIPCHK(f, launch_finalize_pack(f->xst, f->slabs, f->splits, P * f->ldm, P, f->ldm, f->redbuf,
                                         with_prior ? f->prior_prec : nullptr, f->cfg.prior_dense, f->xg_tile[q],
                                         f->xg_tile[q + 1] - f->xg_tile[q]));
        }
        if (last) HIPCHK(f, hipStreamWaitEvent(f->xst, f->xg_ready[1], 0));
        {
          Scope sc(f, LSQAMD_T_EXCH_COLL, f->xst);
          const int64_t off = f->xg_tile[q] * 128 * 128;
          const int64_t cnt = (last ? f->npk + P + 1 : f->xg_tile[q + 1] * 128 * 128) - off;
          rc = comm_all_reduce_on(f, f->xst, f->redbuf + off, cnt);
          if (rc) return rc;
        }
        HIPCHK(f, hipEventRecord(f->xg_done[q], f->xst));
      }
      {
        Scope sc(f, LSQAMD_T_REDUCE);
        Scope sc2(f, LSQAMD_T_EXCH_WAIT);
        for (int q = 0; q < f->xg; ++q) HIPCHK(f, hipStreamWaitEvent(f->st, f->xg_done[q], 0));
      }
    } else {
    for (int q = 0; q < f->xg; ++q) {
      const bool last = q == f->xg - 1;
      {
        Scope sc(f, LSQAMD_T_SYRK);
        GemmTN gq = g;
        gq.work_map = f->syrk_map_g + 4 * (int64_t)f->xg_work[q];
        gq.n_work = f->xg_work[q + 1] - f->xg_work[q];
        HIPCHK(f, launch_gemm_tn(f->st, gq));
      }
      {
        Scope sc(f, LSQAMD_T_GRAD);
        HIPCHK(f, launch_finalize_pack(f->st, f->slabs, f->splits, P * f->ldm, P, f->ldm, f->redbuf,
                                       with_prior ? f->prior_prec : nullptr, f->cfg.prior_dense, f->xg_tile[q],
                                       f->xg_tile[q + 1] - f->xg_tile[q]));
        if (last) {
          rc = finish_gvec();
          if (rc) return rc;
        }
      }
      HIPCHK(f, hipEventRecord(f->xg_ready[q], f->st));
      HIPCHK(f, hipStreamWaitEvent(f->xst, f->xg_ready[q], 0));
      {
        Scope sc(f, LSQAMD_T_EXCH_COLL, f->xst);
        const int64_t off = f->xg_tile[q] * 128 * 128;
        const int64_t cnt = (last ? f->npk + P + 1 : f->xg_tile[q + 1] * 128 * 128) - off;   // the last group carries [J^T f | chi2]
        rc = comm_all_reduce_on(f, f->xst, f->redbuf + off, cnt);
        if (rc) return rc;
      }
      HIPCHK(f, hipEventRecord(f->xg_done[q], f->xst));
    }
    {
      Scope sc(f, LSQAMD_T_REDUCE);
      Scope sc2(f, LSQAMD_T_EXCH_WAIT);
      for (int q = 0; q < f->xg; ++q) HIPCHK(f, hipStreamWaitEvent(f->st, f->xg_done[q], 0));
    }
    }
  } else {
    {
      Scope sc(f, LSQAMD_T_SYRK);
      if (f->N > 0) {
        HIPCHK(f, launch_gemm_tn(f->st, g));
      } else {
        HIPCHK(f, hipMemsetAsync(f->slabs, 0, sizeof(double) * f->splits * P * f->ldm, f->st));
      }
    }
    {
      Scope sc(f, LSQAMD_T_GRAD);
      // when splits == 1 the kernel ignored split_stride and wrote slab 0 directly
      // the slab sum and the prior precision in ONE pass over the packed tiles
      HIPCHK(f, launch_finalize_pack(f->st, f->slabs, f->splits, P * f->ldm, P, f->ldm, f->redbuf,
                                     with_prior ? f->prior_prec : nullptr, f->cfg.prior_dense));
      rc = finish_gvec();
      if (rc) return rc;
    }
    rc = do_reduce(f, f->redbuf, f->npk + P + 1);
    if (rc) return rc;
  }
  if (mirror) HIPCHK(f, launch_packed_diag(f->st, f->redbuf, P, f->diag_dev));   // (else: lm_accept_tail_kernel, the caller's next launch)
  f->have_cov = false;
  f->have_dense_A = false;
  if (!mirror) {   // the caller keeps g, the column norms and chi2 on the device (iterate_device)
    f->njev++;
    f->mirrors_stale = true;
    return 0;
  }
  HIPCHK(f, hipMemcpyAsync(f->pin_g, gvec, sizeof(double) * (P + 1), hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipMemcpyAsync(f->pin_c, f->diag_dev, sizeof(double) * P, hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipStreamSynchronize(f->st));
  for (int64_t j = 0; j < P; ++j) {
    f->hg[j] = f->pin_g[j];
    f->hcoln[j] = std::sqrt(f->pin_c[j] > 0.0 ? f->pin_c[j] : 0.0);
  }
  f->chi2 = f->pin_g[P];
  f->njev++;
  f->have_cov = false;
  f->have_dense_A = false;
  if (!std::isfinite(f->chi2)) FAIL(f, LSQAMD_ENONFINITE, "chi2 is not finite at this point");
  return 0;
}

// (A + mu D^2) v = g  -> f->hv ; returns LSQAMD_ENOTPD when a pivot fails
// launches only: factor (A + mu D^2 | g) and back-substitute; v lands in f->yv[P..2P).
// diag_host == nullptr: D is the device-resident mirror (no host-to-device copy: on this platform
// an SDMA upload followed by a dependent kernel costs ~100 us of cross-engine synchronisation)
// frozen_host (with mu = 0, dogbox): flags of the parameters taken out of the system
int solve_damped_launch(lsqamd_fit *f, double mu, const double *diag_host, const double *frozen_host = nullptr,
                        bool fetch = true, const double *mu_dev = nullptr, bool factor_only = false) {
  const int64_t P = f->P;
  double *gvec = f->redbuf + f->npk;
  {
    Scope sc(f, LSQAMD_T_CHOLESKY);
    const double *dd = f->dscale, *frozen = nullptr;
    if (diag_host || frozen_host) {
      std::memcpy(f->pin_d, diag_host ? diag_host : frozen_host, sizeof(double) * P);
      HIPCHK(f, hipMemcpyAsync(f->diag_dev, f->pin_d, sizeof(double) * P, hipMemcpyHostToDevice, f->st));
      if (diag_host) dd = f->diag_dev;
      else frozen = f->diag_dev;
    }
    // (the pivot-failure word is cleared by the kernel that builds the matrix, the hand-off granules of the
    // back substitution by the one that copies its right-hand side: two memsets fewer per trial)
    HIPCHK(f, launch_build_damped(f->st, f->redbuf, P, f->ldm, mu, dd, gvec, f->M, frozen, mu_dev, f->info_dev));
    HIPCHK(f, potrf_upper(f->st, f->M, P, f->ldm, f->ncols_aug, f->chol_work, f->info_dev, true));
  }
  if (factor_only) return 0;    // (small systems: the caller's single-workgroup kernel does the rest)
  {
    Scope sc(f, LSQAMD_T_SOLVE);
    HIPCHK(f, launch_copy_column_zero(f->st, f->M + P, f->ldm, f->yv, P, f->yv + 2 * P, backsolve_scratch_bytes(P)));
    HIPCHK(f, backsolve_upper(f->st, f->M, P, f->ldm, f->chol_work, f->yv, f->yv + 2 * P, f->info_dev, true));
    if (fetch) {
      HIPCHK(f, hipMemcpyAsync(f->pin_v, f->yv + P, sizeof(double) * P, hipMemcpyDeviceToHost, f->st));
      HIPCHK(f, hipMemcpyAsync(f->pin_s + 4, f->info_dev, sizeof(int32_t), hipMemcpyDeviceToHost, f->st));
    }
  }
  return 0;
}

// after the stream has been synchronised: v -> f->hv; LSQAMD_ENOTPD when a pivot failed
int solve_damped_collect(lsqamd_fit *f) {
  const int64_t P = f->P;
  int32_t info = 0;
  std::memcpy(f->hv.data(), f->pin_v, sizeof(double) * P);
  std::memcpy(&info, f->pin_s + 4, sizeof(int32_t));
  f->ntrial++;
  if (info == -77) FAIL(f, LSQAMD_EHIP, "back substitution: a workgroup of the chain gave up waiting");
  if (info != 0) {
    f->chol_fail++;
    return LSQAMD_ENOTPD;
  }
  for (int64_t j = 0; j < P; ++j)
    if (!std::isfinite(f->hv[j])) {
      f->chol_fail++;
      return LSQAMD_ENOTPD;
    }
  return 0;
}

// (A + mu D^2) v = g  -> f->hv ; returns LSQAMD_ENOTPD when a pivot fails
int solve_damped_dev(lsqamd_fit *f, double mu, const double *diag_host, const double *frozen_host) {
  const int rc = solve_damped_launch(f, mu, diag_host, frozen_host);
  if (rc) return rc;
  HIPCHK(f, hipStreamSynchronize(f->st));
  return solve_damped_collect(f);
}

void scale_init(lsqamd_fit *f) {
  (void)launch_scale_update(f->st, f->P, f->opt.scaler, 1, f->diag_dev, f->dscale);  // diag_dev = coln^2
  for (int64_t j = 0; j < f->P; ++j) {
    if (f->opt.scaler == LSQAMD_SCALE_LEVENBERG)    // (scipy methods: D = 1 / x_scale, lsqamd_set_x_scale)
      f->hdiag[j] = (f->opt.trs >= LSQAMD_TRS_TRF && !f->x_scale.empty()) ? 1.0 / f->x_scale[j] : 1.0;
    else f->hdiag[j] = f->hcoln[j] == 0.0 ? 1.0 : f->hcoln[j];
  }
}

void scale_update(lsqamd_fit *f) {
  (void)launch_scale_update(f->st, f->P, f->opt.scaler, 0, f->diag_dev, f->dscale);
  for (int64_t j = 0; j < f->P; ++j) {
    if (f->opt.scaler == LSQAMD_SCALE_MORE) f->hdiag[j] = std::fmax(f->hdiag[j], f->hcoln[j]);
    else if (f->opt.scaler == LSQAMD_SCALE_MARQUARDT) f->hdiag[j] = f->hcoln[j] == 0.0 ? 1.0 : f->hcoln[j];
  }
}

// ---- pieces shared by the trust-region sub-problem solvers -------------------
// y = A x with A = the reduced J^T J (+ prior): dense symmetric copy kept in f->Wl
int symv_host(lsqamd_fit *f, const double *x, double *y) {
  const int64_t P = f->P;
  if (!f->have_dense_A) {
    HIPCHK(f, launch_unpack_sym(f->st, f->redbuf, P, f->Wl, f->ldm));
    f->have_dense_A = true;
  }
  HIPCHK(f, hipMemcpyAsync(f->tvec, x, sizeof(double) * P, hipMemcpyHostToDevice, f->st));
  HIPCHK(f, launch_gemv_rows(f->st, f->Wl, f->ldm, P, P, f->tvec, f->yv));
  HIPCHK(f, hipMemcpyAsync(y, f->yv, sizeof(double) * P, hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipStreamSynchronize(f->st));
  return 0;
}

double dot_h(const std::vector<double> &a, const std::vector<double> &b) {
  double s = 0.0;
  for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
  return s;
}

// (A + mu D^2) out = rhs with the factor U already in f->M (lmaccel's second solve, lm.c lm_step):
// forward substitution block by block through the TN GEMM (X = inv(U_kk), then X = U[k, k+1:]),
// then the usual back substitution.
int solve_with_factor(lsqamd_fit *f, const double *rhs, double *out) {
  const int64_t P = f->P;
  HIPCHK(f, hipMemcpyAsync(f->yv, rhs, sizeof(double) * P, hipMemcpyHostToDevice, f->st));
  for (int64_t k0 = 0; k0 < P; k0 += 128) {
    const int64_t nb = P - k0 < 128 ? P - k0 : 128;
    GemmTN a;  // y_k <- inv(U_kk)^T y_k
    a.X = f->chol_work + (k0 / 128) * 128 * 128; a.ldx = 128;
    a.Y = f->yv + k0; a.ldy = 1; a.C = f->yv + k0; a.ldc = 1;
    a.M = nb; a.N = 1; a.K = nb;
    a.x_upper_tri = 1;
    HIPCHK(f, launch_gemm_tn(f->st, a));
    const int64_t rest = P - (k0 + nb);
    if (rest <= 0) continue;
    GemmTN b;  // y_rest -= U[k, rest]^T y_k
    b.X = f->M + k0 * f->ldm + k0 + nb; b.ldx = f->ldm;
    b.Y = f->yv + k0; b.ldy = 1; b.C = f->yv + k0 + nb; b.ldc = 1;
    b.M = rest; b.N = 1; b.K = nb;
    b.alpha = -1.0; b.beta = 1.0;
    HIPCHK(f, launch_gemm_tn(f->st, b));
  }
  HIPCHK(f, backsolve_upper(f->st, f->M, P, f->ldm, f->chol_work, f->yv, f->yv + 2 * P, f->info_dev));
  HIPCHK(f, hipMemcpyAsync(out, f->yv + P, sizeof(double) * P, hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipStreamSynchronize(f->st));
  return 0;
}

// G_h = J(x)^T r(x_h) + Lambda (x_h - pbar): the gradient-like vector the finite-difference
// second directional derivative needs (fdfvv.c), all-reduced like the gradient
int grad_like_at(lsqamd_fit *f, const double *xh, double *out) {
  const int64_t P = f->P;
  HIPCHK(f, hipMemcpyAsync(f->p_trial, xh, sizeof(double) * P, hipMemcpyHostToDevice, f->st));
  double c2 = 0.0;
  int rc = eval_residual_dev(f, f->p_trial, &c2);  // f->r = whitened residual at x_h
  if (rc) return rc;
  if ((rc = ensure_J(f))) return rc;
  HIPCHK(f, launch_colsum_dot(f->st, f->J, f->N, f->ld, P, P, f->partial, f->npartial, f->yv, f->r));
  if (f->cfg.has_prior && f->adds_prior)
    HIPCHK(f, launch_add_prior(f->st, nullptr, P, f->prior_prec, f->cfg.prior_dense, f->prior_mean,
                               f->p_trial, f->tvec, f->yv, 0));
  rc = do_reduce(f, f->yv, P);
  if (rc) return rc;
  HIPCHK(f, hipMemcpyAsync(out, f->yv, sizeof(double) * P, hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipStreamSynchronize(f->st));
  return 0;
}

struct Legs {  // dogleg.c / subspace2D.c state for one trust_iterate
  std::vector<double> dx_sd, dx_gn, q0, q1;
  double norm_Dsd = 0.0, norm_Dgn = -1.0, norm_Dinvg = 0.0, norm_JDinv2g = 0.0;
  bool gn_failed = false;
  int sub = -1;  // subspace2D basis: -1 not built, 0 legs parallel, 1 ready
  double subg[2] = {0, 0}, subB[2][2] = {{0, 0}, {0, 0}};
};

double scaled_norm(const std::vector<double> &d, const std::vector<double> &v) {
  double s = 0.0;
  for (size_t i = 0; i < v.size(); ++i) s += d[i] * v[i] * d[i] * v[i];
  return std::sqrt(s);
}

// dogleg_preloop: steepest-descent leg dx_sd = -alpha D^-2 g, alpha = |D^-1 g|^2 / |J D^-2 g|^2
int legs_preloop(lsqamd_fit *f, Legs &L) {
  const int64_t P = f->P;
  std::vector<double> w2(P), Aw(P);
  double n1 = 0.0;
  for (int64_t j = 0; j < P; ++j) {
    const double w1 = f->hg[j] / f->hdiag[j];
    n1 += w1 * w1;
    w2[j] = w1 / f->hdiag[j];
  }
  L.norm_Dinvg = std::sqrt(n1);
  int rc = symv_host(f, w2.data(), Aw.data());
  if (rc) return rc;
  const double q = dot_h(w2, Aw);
  L.norm_JDinv2g = std::sqrt(q > 0.0 ? q : 0.0);
  const double u = L.norm_Dinvg / L.norm_JDinv2g;
  L.dx_sd.resize(P);
  for (int64_t j = 0; j < P; ++j) L.dx_sd[j] = -(u * u) * w2[j];
  L.norm_Dsd = scaled_norm(f->hdiag, L.dx_sd);
  L.norm_Dgn = -1.0;
  L.gn_failed = false;
  L.sub = -1;
  return 0;
}

// Gauss-Newton leg: the damped solve at mu = 0 (dogleg_calc_gn)
int legs_gn(lsqamd_fit *f, Legs &L) {
  if (L.norm_Dgn >= 0.0 || L.gn_failed) return 0;
  const int rc = solve_damped_dev(f, 0.0, f->hdiag.data());
  if (rc < 0 && rc != LSQAMD_ENOTPD) return rc;
  if (rc == LSQAMD_ENOTPD) {  // singular J^T J: no Gauss-Newton point, stay on the gradient leg
    L.gn_failed = true;
    return 0;
  }
  L.dx_gn.resize(f->P);
  for (int64_t j = 0; j < f->P; ++j) L.dx_gn[j] = -f->hv[j];
  L.norm_Dgn = scaled_norm(f->hdiag, L.dx_gn);
  return 0;
}

double dogleg_beta(const lsqamd_fit *f, const Legs &L, double t, double delta) {
  double a = 0.0, b = 0.0;
  for (int64_t j = 0; j < f->P; ++j) {
    const double w = t * L.dx_gn[j] - L.dx_sd[j];
    const double d2 = f->hdiag[j] * f->hdiag[j];
    a += d2 * w * w;
    b += L.dx_sd[j] * d2 * w;
  }
  b *= 2.0;
  const double c = (L.norm_Dsd + delta) * (L.norm_Dsd - delta);
  const double disc = std::sqrt(b * b - 4.0 * a * c);
  return b > 0.0 ? (-2.0 * c) / (b + disc) : (-b + disc) / (2.0 * a);
}

// argmin g.q + q.B.q/2 on |q| = delta for a 2 x 2 positive semi-definite B (secular equation)
void solve_tr_2d(const double B[2][2], const double g[2], double delta, double q[2]) {
  const double tr = B[0][0] + B[1][1], df = B[0][0] - B[1][1];
  const double rad = std::sqrt(0.25 * df * df + B[0][1] * B[0][1]);
  const double w0 = 0.5 * tr - rad, w1 = 0.5 * tr + rad;
  double v0[2], v1[2];  // eigenvectors
  if (std::fabs(B[0][1]) > 0.0) {
    v1[0] = w1 - B[1][1]; v1[1] = B[0][1];
    const double n = std::hypot(v1[0], v1[1]);
    v1[0] /= n; v1[1] /= n;
  } else if (B[0][0] >= B[1][1]) { v1[0] = 1.0; v1[1] = 0.0; }
  else { v1[0] = 0.0; v1[1] = 1.0; }
  v0[0] = -v1[1]; v0[1] = v1[0];
  const double g0 = v0[0] * g[0] + v0[1] * g[1], g1 = v1[0] * g[0] + v1[1] * g[1];
  auto qnorm = [&](double lam) { return std::hypot(g0 / (w0 + lam), g1 / (w1 + lam)); };
  double lo = 0.0, hi = std::fmax(1.0, std::hypot(g[0], g[1]) / delta);
  while (qnorm(hi) > delta) hi *= 2.0;
  for (int it = 0; it < 200; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (qnorm(mid) > delta) lo = mid; else hi = mid;
    if (hi - lo <= 1e-16 * hi) break;
  }
  const double lam = 0.5 * (lo + hi);
  const double c0 = -g0 / (w0 + lam), c1 = -g1 / (w1 + lam);
  q[0] = v0[0] * c0 + v1[0] * c1;
  q[1] = v0[1] * c0 + v1[1] * c1;
}

// orthonormal basis of span(D dx_sd, D dx_gn) and the 2 x 2 model in it (subspace2D_preloop)
int legs_subspace(lsqamd_fit *f, Legs &L) {
  if (L.sub >= 0) return 0;
  const int64_t P = f->P;
  L.q0.resize(P); L.q1.resize(P);
  double c = 0.0;
  for (int64_t j = 0; j < P; ++j) {
    L.q0[j] = f->hdiag[j] * L.dx_sd[j] / L.norm_Dsd;
    L.q1[j] = f->hdiag[j] * L.dx_gn[j] / L.norm_Dgn;
    c += L.q0[j] * L.q1[j];
  }
  double r = 0.0;
  for (int64_t j = 0; j < P; ++j) { L.q1[j] -= c * L.q0[j]; r += L.q1[j] * L.q1[j]; }
  r = std::sqrt(r);
  if (!(r > 20.0 * (double)(P + 2) * 2.220446049250313e-16)) {  // gsl_linalg_QRPT_rank default tolerance
    L.sub = 0;
    return 0;
  }
  for (int64_t j = 0; j < P; ++j) L.q1[j] /= r;
  std::vector<double> d0(P), d1(P), A0(P), A1(P);
  for (int64_t j = 0; j < P; ++j) { d0[j] = L.q0[j] / f->hdiag[j]; d1[j] = L.q1[j] / f->hdiag[j]; }
  int rc = symv_host(f, d0.data(), A0.data());
  if (rc) return rc;
  rc = symv_host(f, d1.data(), A1.data());
  if (rc) return rc;
  L.subB[0][0] = dot_h(d0, A0); L.subB[0][1] = L.subB[1][0] = dot_h(d0, A1); L.subB[1][1] = dot_h(d1, A1);
  L.subg[0] = dot_h(d0, f->hg); L.subg[1] = dot_h(d1, f->hg);
  L.sub = 1;
  return 0;
}

// trs->step for the dogleg family: dx for the current trust radius
int legs_step(lsqamd_fit *f, Legs &L, std::vector<double> &dx) {
  const int64_t P = f->P;
  const double delta = f->delta;
  const int trs = f->opt.trs;
  auto scaled = [&](const std::vector<double> &v, double s) { for (int64_t j = 0; j < P; ++j) dx[j] = s * v[j]; };
  if (trs == LSQAMD_TRS_SUBSPACE2D) {
    int rc = legs_gn(f, L);
    if (rc) return rc;
    if (L.gn_failed) { scaled(L.dx_sd, delta / L.norm_Dsd); return 0; }
    if (L.norm_Dgn <= delta) { dx = L.dx_gn; return 0; }
    rc = legs_subspace(f, L);
    if (rc) return rc;
    if (L.sub == 0) { scaled(L.dx_sd, delta / L.norm_Dsd); return 0; }
    double q[2];
    solve_tr_2d(L.subB, L.subg, delta, q);
    for (int64_t j = 0; j < P; ++j) dx[j] = (L.q0[j] * q[0] + L.q1[j] * q[1]) / f->hdiag[j];
    return 0;
  }
  if (L.norm_Dsd >= delta) { scaled(L.dx_sd, delta / L.norm_Dsd); return 0; }
  int rc = legs_gn(f, L);
  if (rc) return rc;
  if (L.gn_failed) { dx = L.dx_sd; return 0; }
  if (L.norm_Dgn <= delta) { dx = L.dx_gn; return 0; }
  double t = 1.0;
  if (trs == LSQAMD_TRS_DDOGLEG) {
    const double u = L.norm_Dinvg / L.norm_JDinv2g;
    const double gd = dot_h(f->hg, L.dx_gn);
    const double c = u * u * (L.norm_Dinvg / std::fabs(gd)) * L.norm_Dinvg;
    t = 1.0 - 0.8 * (1.0 - c);
    if (t * L.norm_Dgn <= delta) { scaled(L.dx_gn, delta / L.norm_Dgn); return 0; }
  }
  const double beta = dogleg_beta(f, L, t, delta);
  for (int64_t j = 0; j < P; ++j) dx[j] = L.dx_sd[j] + beta * (t * L.dx_gn[j] - L.dx_sd[j]);
  return 0;
}

// ---- variable projection (nonlinear_fit's linear=, src/lsqfit/__init__.py:738-787) -------------
// The residual is linear in the masked parameters a, and the reference hands the plugin
// phi(theta) = min_a chi2(a, theta): its wrapped fit function solves for a at EVERY evaluation.
// Here: every trial point gets a full evaluation (A_t, g_t, chi2_t) followed by the exact linear
// solve A_aa da = -g_a (the masked build of the damped matrix, theta frozen) and
// chi2(a + da, theta_t) = chi2_t + g_a.da; the step in theta is the LM step of the projected
// functional, (S + mu D_t^2) dtheta = -g_t with the Schur complement S = A_tt - A_ta A_aa^-1 A_at,
// obtained by solving the full system with the a-block of the damping matrix set to zero
// (Kaufman's form of the Golub-Pereyra Jacobian; the reference differentiates through its lstsq
// and keeps the second term too: same minimum, slightly different iterates).  An accepted point
// is evaluated once more at the projected a; a rejected trial restores (A, g) from a device copy.
int iterate_varpro(lsqamd_fit *f) {
  const int64_t P = f->P;
  const int64_t nstash = f->npk + P + 1;
  const bool dev_stash = P * f->ldm >= nstash;       // tiny P: the tile padding exceeds P x ldm
  std::vector<double> host_stash, dt(P), frozen(P), xt(P), v(P), keep_g, keep_c;
  for (int64_t j = 0; j < P; ++j) frozen[j] = f->linear[j] ? 0.0 : 1.0;
  int bad_steps = 0;
  while (true) {
    for (int64_t j = 0; j < P; ++j) dt[j] = f->linear[j] ? 0.0 : f->hdiag[j];
    double rho = -1.0;
    int rc = solve_damped_dev(f, f->mu, dt.data());
    if (rc < 0 && rc != LSQAMD_ENOTPD) return rc;
    bool stashed = false;
    const double chi2_cur = f->chi2;
    if (rc == 0) {
      double vg = 0.0, dv2 = 0.0;
      for (int64_t j = 0; j < P; ++j) {
        v[j] = f->hv[j];
        vg += v[j] * f->hg[j];
        dv2 += dt[j] * v[j] * dt[j] * v[j];
        xt[j] = f->hx[j] - v[j];
      }
      const double pred_num = vg + f->mu * dv2;      // |J v|^2 + 2 mu |D v|^2 by the system v solves
      keep_g = f->hg; keep_c = f->hcoln;
      if (dev_stash) {
        HIPCHK(f, hipMemcpyAsync(f->cov, f->redbuf, sizeof(double) * nstash, hipMemcpyDeviceToDevice, f->st));
      } else {
        host_stash.resize(nstash);
        HIPCHK(f, hipMemcpyAsync(host_stash.data(), f->redbuf, sizeof(double) * nstash, hipMemcpyDeviceToHost, f->st));
        HIPCHK(f, hipStreamSynchronize(f->st));
      }
      stashed = true;
      std::memcpy(f->pin_x, xt.data(), sizeof(double) * P);
      HIPCHK(f, hipMemcpyAsync(f->p_trial, f->pin_x, sizeof(double) * P, hipMemcpyHostToDevice, f->st));
      rc = eval_normal_dev(f, f->p_trial);           // full evaluation at the trial point ...
      f->nfev++;
      if (rc < 0 && rc != LSQAMD_ENONFINITE) return rc;
      if (rc == 0) {
        rc = solve_damped_dev(f, 0.0, nullptr, frozen.data());   // ... and the exact linear solve there
        if (rc < 0 && rc != LSQAMD_ENOTPD) return rc;
      }
      if (rc == 0) {
        double chi2_t = f->chi2;
        for (int64_t j = 0; j < P; ++j)
          if (f->linear[j]) { chi2_t -= f->hg[j] * f->hv[j]; xt[j] -= f->hv[j]; }
        for (int64_t j = 0; j < P; ++j) f->hdx[j] = xt[j] - f->hx[j];   // trial steps count for the xtol test
        const double normf = std::sqrt(chi2_cur), normf_t = std::sqrt(chi2_t > 0.0 ? chi2_t : 0.0);
        if (normf_t < normf) {
          const double u = normf_t / normf;
          const double pred = pred_num / chi2_cur;
          rho = pred > 0.0 ? (1.0 - u * u) / pred : -1.0;
        }
      }
    }
    if (rho > 0.0) {
      std::memcpy(f->pin_x, xt.data(), sizeof(double) * P);
      HIPCHK(f, hipMemcpyAsync(f->p_trial, f->pin_x, sizeof(double) * P, hipMemcpyHostToDevice, f->st));
      rc = eval_normal_dev(f, f->p_trial);           // J, A, g at the projected point
      f->nfev++;
      if (rc) return rc;
      f->hx = xt;
      std::swap(f->p_dev, f->p_trial);
      scale_update(f);
      const double b = 2.0 * rho - 1.0;
      f->mu *= std::fmax(0.333333333333333, 1.0 - b * b * b);
      f->nu = 2;
      return 0;
    }
    if (stashed) {                                   // back to the current point's A, g
      if (dev_stash) HIPCHK(f, hipMemcpyAsync(f->redbuf, f->cov, sizeof(double) * nstash, hipMemcpyDeviceToDevice, f->st));
      else {   // (on the handle's stream: the library never touches the legacy default stream -- see copy_sync in batch.hip)
        HIPCHK(f, hipMemcpyAsync(f->redbuf, host_stash.data(), sizeof(double) * nstash, hipMemcpyHostToDevice, f->st));
        HIPCHK(f, hipStreamSynchronize(f->st));
      }
      f->hg = keep_g; f->hcoln = keep_c;
      f->chi2 = chi2_cur;
      f->have_dense_A = false;
    }
    f->mu *= (double)f->nu;
    f->nu <<= 1;
    if (++bad_steps > 15) {
      const int rc2 = eval_normal_dev(f, f->p_dev);  // leave J, f of the CURRENT point behind
      if (rc2) return rc2;
      return LSQAMD_ENOPROG;
    }
  }
}

// host copies of x, g, D, the column norms and the last step after iterations that kept them on
// the device
int ensure_J(lsqamd_fit *f) {
  if (!f->J_stale) return 0;
  ModelArgs m = model_args(f, f->p_dev);
  HIPCHK(f, launch_jacobian_ex(f->st, m, f->J, f->Jraw, f->ld));
  if (f->cfg.n_blocks > 0) {   // (the one-launch fit kernel takes correlated rows too: their whitening product is redone here)
    int64_t fused = 0;
    const int rc = whiten_jacobian(f, &fused, false);
    if (rc) return rc;
  }
  f->J_stale = false;
  return 0;
}

int refresh_mirrors(lsqamd_fit *f) {
  if (!f->mirrors_stale) return 0;
  const int64_t P = f->P;
  HIPCHK(f, hipMemcpyAsync(f->pin_x, f->p_dev, sizeof(double) * P, hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipMemcpyAsync(f->pin_g, f->redbuf + f->npk, sizeof(double) * (P + 1), hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipMemcpyAsync(f->pin_d, f->dscale, sizeof(double) * P, hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipMemcpyAsync(f->pin_c, f->diag_dev, sizeof(double) * P, hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipMemcpyAsync(f->pin_v, f->yv + P, sizeof(double) * P, hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipStreamSynchronize(f->st));
  for (int64_t j = 0; j < P; ++j) {
    f->hx[j] = f->pin_x[j];
    f->hg[j] = f->pin_g[j];
    f->hdiag[j] = f->pin_d[j];
    f->hcoln[j] = std::sqrt(f->pin_c[j] > 0.0 ? f->pin_c[j] : 0.0);
    f->hv[j] = f->pin_v[j];
    f->hdx[j] = -f->pin_v[j];
  }
  f->mirrors_stale = false;
  return 0;
}

// One trust_iterate of plain lm with the state on the device: per trial the host queues
//   solve -> (trial point, v.g, |D v|^2) -> residual (+ all-reduce) -> decision (rho, mu, nu, delta)
// and reads ONE 128-byte record; an accepted trial queues the Jacobian, the normal equations, the
// update of D and the convergence test and reads the record once more.  No P-length vector
// crosses PCIe, no P-length loop runs on the host.
// The two halves of a device-resident LM step as launch sequences (no host reads in between), so that
// each can be captured once per p-buffer parity and replayed: small problems spend more host time
// launching their ~35 kernels than the GPU spends running them.
// small single-rank fits: fused single-workgroup tails (LSQAMD_SMALL_FUSE=0: the general kernels)
static bool small_fuse(const lsqamd_fit *f) {
  const char *e = getenv("LSQAMD_SMALL_FUSE");     // read per call (tests flip it between fits; a getenv is ~50 ns)
  return !(e && e[0] == '0') && !f->comm && !f->reduce;
}

static int enqueue_trial(lsqamd_fit *f) {
  const int64_t P = f->P;
  double *gvec = f->redbuf + f->npk;
  const bool watch = f->opt.solver == LSQAMD_SOLVER_QR;    // solver = qr: how much of each column its pivot retained
  if (small_fuse(f) && P <= 32) {
    // up to 32 parameters: the whole trial solve (build, factor, substitute, trial point, record) by one wave
    // out of the packed tile -- one launch instead of six
    {
      Scope sc(f, LSQAMD_T_CHOLESKY);
      if (P <= 12)
        HIPCHK(f, launch_lm_tiny12_solve(f->st, f->redbuf, P, gvec, f->dscale, f->p_dev, f->p_trial, f->yv + P, f->lmd, f->info_dev,
                                         watch ? 1 : 0));
      else
        HIPCHK(f, launch_lm_small_solve(f->st, f->redbuf, P, gvec, f->dscale, f->p_dev, f->p_trial, f->yv + P, f->lmd, f->info_dev,
                                        watch ? 1 : 0));
    }
    const int rct = eval_residual_launch(f, f->p_trial, true);
    if (rct) return rct;
    if (!f->lm_zero_copy) HIPCHK(f, hipMemcpyAsync(f->pin_lm, f->lmd, sizeof(double) * LMS_COUNT, hipMemcpyDeviceToHost, f->st));
    return 0;
  }
  const bool small = small_fuse(f) && (P == 128 || P == 256);   // (whole 128-blocks: the kernel's GEMVs read full tiles)
  int rc = solve_damped_launch(f, f->mu, nullptr, nullptr, false, f->lmd + LMS_MU, small);
  if (rc) return rc;
  if (small) {
    Scope sc(f, LSQAMD_T_SOLVE);
    HIPCHK(f, launch_lm_solve_tail_small(f->st, f->M, f->ldm, P, f->chol_work, f->p_dev, gvec, f->dscale, f->p_trial, f->yv + P,
                                         f->lmd, watch ? f->diag_dev : nullptr, f->info_dev));
  } else {
    HIPCHK(f, launch_lm_trial(f->st, P, f->p_dev, f->yv + P, gvec, f->dscale, f->p_trial, f->lmd, watch ? f->M : nullptr, f->ldm,
                              watch ? f->diag_dev : nullptr));
  }
  const bool alone = !f->comm && !f->reduce;
  rc = eval_residual_launch(f, f->p_trial, alone);
  if (rc) return rc;
  if (!alone)
    HIPCHK(f, launch_lm_decide(f->st, f->red_scalar, f->info_dev, f->opt.factor_up, f->opt.factor_down, f->lmd));
  if (!f->lm_zero_copy) HIPCHK(f, hipMemcpyAsync(f->pin_lm, f->lmd, sizeof(double) * LMS_COUNT, hipMemcpyDeviceToHost, f->st));
  return 0;
}

static int enqueue_accept(lsqamd_fit *f) {   // p_trial becomes the point; the caller swaps the buffers afterwards
  const int64_t P = f->P;
  double *gvec = f->redbuf + f->npk;
  f->r_fresh = true;      // f->r is the whitened residual AT p_trial: the trial evaluation left it there
  f->r_ptr = f->p_trial;
  int rc = eval_normal_dev(f, f->p_trial, false);
  if (rc) return rc;
  // (prior_deferred: g += Lambda (p - pbar), chi2 += ... inside the tail kernel -- one launch fewer)
  const bool wp = f->cfg.has_prior && f->adds_prior;
  HIPCHK(f, launch_lm_accept_tail(f->st, f->redbuf, P, f->opt.scaler, f->diag_dev, f->dscale, f->p_trial, f->yv + P, gvec,
                                  f->opt.xtol, f->opt.gtol, f->lmd, f->prior_deferred ? f->tvec : nullptr, f->prior_mean,
                                  f->nrm_in_tail ? f->nrm_part : nullptr, f->nrm_in_tail, (f->nrm_in_tail && wp) ? f->prior_prec : nullptr,
                                  f->cfg.prior_dense));
  if (!f->lm_zero_copy) HIPCHK(f, hipMemcpyAsync(f->pin_lm, f->lmd, sizeof(double) * LMS_COUNT, hipMemcpyDeviceToHost, f->st));
  return 0;
}

// run one half eagerly, or capture it (second time this parity sees it) and replay it
static int run_half(lsqamd_fit *f, int which, int (*enqueue)(lsqamd_fit *)) {
  static const bool env_off = [] { const char *e = getenv("LSQAMD_STEP_GRAPH"); return e && e[0] == '0'; }();
  static const int64_t maxp = [] { const char *e = getenv("LSQAMD_STEP_GRAPH_MAXP"); return e ? atoll(e) : (int64_t)1024; }();
  const bool eligible = !env_off && !f->step_graph_off && !f->timing && !f->comm && !f->reduce && f->P <= maxp;
  const int par = f->p_dev == f->p_buf0 ? 0 : 1;
  if (!eligible) return enqueue(f);
  hipGraphExec_t &exec = f->step_exec[par][which];
  if (!exec) {
    if (f->step_seen[par][which]++ == 0) return enqueue(f);      // warm-up: one-time attribute calls happen here
    const int64_t njev0 = f->njev;
    static const bool diag_capture = getenv("LSQAMD_FIT_DIAG") != nullptr;       // developer knob: how long do captures take?
    const auto tc0 = std::chrono::steady_clock::now();
    hipError_t e = hipStreamBeginCapture(f->st, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      f->step_graph_off = true;
      return enqueue(f);
    }
    const int rc = enqueue(f);
    hipGraph_t gr = nullptr;
    e = hipStreamEndCapture(f->st, &gr);
    f->njev = njev0;                                             // counted when the graph runs, below
    if (rc || e != hipSuccess || !gr || hipGraphInstantiate(&exec, gr, nullptr, nullptr, 0) != hipSuccess) {
      if (gr) (void)hipGraphDestroy(gr);
      (void)hipGetLastError();
      exec = nullptr;
      f->step_graph_off = true;
      // enqueue failed AND the capture ended in error: the capture was invalidated (another thread's legacy-stream call, for
      // one) and took the captured launches down with it -- nothing ran; the half step is queued again, eagerly.  A failure
      // with a healthy capture is the enqueue's own
      if (rc && e == hipSuccess) return rc;
      capture_reset(f->st);      // (an invalidated capture leaves the stream unusable until it is reset: common.h)
      return enqueue(f);
    }
    (void)hipGraphDestroy(gr);
    if (diag_capture)
      fprintf(stderr, "lsqamd: captured + instantiated half-step graph [parity %d][%s] of handle %p in %.2f ms\n", par, which ? "accept" : "trial",
              (void *)f, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count());
  }
  HIPCHK(f, hipGraphLaunch(exec, f->st));
  f->graph_launches++;
  if (which == 1) {            // what enqueue_accept's host side does besides launching
    f->r_fresh = false;
    f->have_cov = false;
    f->have_dense_A = false;
    f->njev++;
    f->mirrors_stale = true;
    f->J_stale = f->used_nrm;   // (the captured branch formed its normal equations without writing J: see eval_normal_dev;
                                //  a handle takes that route for all of its accepted steps or for none)
  } else {
    f->r_fresh = false;
  }
  return 0;
}

// The record of the half step just queued has arrived in pin_lm.  With the kernels mirroring it themselves (zero copy)
// the host POLLS the mirror's sequence number for a while before it falls back to a stream synchronisation: waking
// from the latter costs ~15 us of the ~27 us between two half steps of a small fit (LSQAMD_POLL=0: always synchronise).
static int wait_record(lsqamd_fit *f) {
  static const bool poll = [] { const char *e = getenv("LSQAMD_POLL"); return !(e && e[0] == '0'); }();
  f->lm_seq_expect += 1.0;
  if (!f->lm_zero_copy) {     // (the caller queued a copy of the record behind the kernels)
    HIPCHK(f, hipStreamSynchronize(f->st));
    return 0;
  }
  // The kernels' stores to the mirror arrive in no particular order (vecops.hip lm_publish): a snapshot counts only when its
  // sequence number is the awaited one AND its checksum fits.  Once one does, every word of this half step has landed and
  // nothing writes the mirror again before the host queues the next half step: the callers read pin_lm itself afterwards.
  volatile unsigned long long *h = reinterpret_cast<volatile unsigned long long *>(f->pin_lm);
  unsigned long long w[LMS_COUNT];
  auto arrived = [&]() {
    double seq;
    w[LMS_SEQ] = h[LMS_SEQ];
    std::memcpy(&seq, &w[LMS_SEQ], sizeof(double));
    if (!(seq >= f->lm_seq_expect)) return false;
    std::atomic_thread_fence(std::memory_order_acquire);
    for (int i = 0; i < LMS_COUNT; ++i) w[i] = h[i];
    std::memcpy(&seq, &w[LMS_SEQ], sizeof(double));
    if (seq >= f->lm_seq_expect && w[LMS_CHECK] == lm_record_checksum(w)) return true;
    g_handoff[0]++;
    return false;
  };
  auto audited = [&]() -> int {    // test knob: the record the host is about to act on vs the device's own, once the stream has drained
    if (!getenv("LSQAMD_VERIFY_HANDOFF")) return 0;
    unsigned long long dev[LMS_COUNT];
    HIPCHK(f, hipStreamSynchronize(f->st));
    HIPCHK(f, hipMemcpyAsync(dev, f->lmd, sizeof(dev), hipMemcpyDeviceToHost, f->st));
    HIPCHK(f, hipStreamSynchronize(f->st));
    for (int i = 0; i < LMS_COUNT; ++i)
      if (dev[i] != w[i]) {
        fprintf(stderr, "lsqamd HANDOFF MISMATCH record word %d: host %016llx device %016llx\n", i, w[i], dev[i]);
        g_handoff[2]++;
      }
    return 0;
  };
  if (poll) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1;; ++spins) {
      if (arrived()) return audited();
      cpu_relax();
      if ((spins & 63) == 0) {
        const auto dt = std::chrono::steady_clock::now() - t0;
        if (dt > std::chrono::microseconds(3000)) break;
        if (dt > std::chrono::microseconds(200)) sched_yield();   // (a long half step: stop hogging the core)
      }
    }
  }
  HIPCHK(f, hipStreamSynchronize(f->st));
  if (arrived()) return audited();
  // the stream has drained and the mirror still does not verify: take the record from the device
  g_handoff[1]++;
  HIPCHK(f, hipMemcpyAsync(f->pin_lm, f->lmd, sizeof(double) * LMS_COUNT, hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipStreamSynchronize(f->st));
  return 0;
}

int iterate_device(lsqamd_fit *f) {
  const int64_t P = f->P;
  double *gvec = f->redbuf + f->npk;
  int bad_steps = 0;
  while (true) {
    const double mu0 = f->mu, delta0 = f->delta;
    const long nu0 = f->nu;
    const double *st = f->pin_lm;
    static const bool qr_steps_off = [] { const char *e = getenv("LSQAMD_QR_STEPS"); return e && e[0] == '0'; }();   // developer knob
    const bool can_qr = f->opt.solver == LSQAMD_SOLVER_QR && f->qr_work && !f->comm && !f->reduce && !qr_steps_off;
    int rc = 0;
    bool need_qr = can_qr && f->qr_steps_on;
    if (!need_qr) {
      rc = run_half(f, 0, enqueue_trial);
      if (rc) return rc;
      rc = wait_record(f);
      if (rc) return rc;
      // solver = qr (the reference's default, src/lsqfit/_gsl.pyx:571,646-647): gsl factors [J ; sqrt(mu) D] itself, error
      // ~ cond eps; the damped normal equations have just gone through cond^2 ~ 1 / PIVMIN.  Once a trial's factor
      // retains less than 1e-8 of some column (or has no positive pivot at all), this fit's steps come from the
      // orthogonal factorisation (solve_damped_qr) -- the trial just taken included: the device's decision on it is
      // taken back (mu, nu, delta as before) and the trial is repeated with the accurate step
      if (can_qr && (st[LMS_SOLVED] == 0.0 || st[LMS_PIVMIN] < 1e-8)) {
        if (st[LMS_SOLVED] == 0.0) f->chol_fail++;
        f->qr_steps_on = true;
        need_qr = true;
      }
    }
    f->ntrial++;
    if (need_qr) {
      double *rec = f->pin_lm;
      rec[LMS_MU] = mu0; rec[LMS_NU] = (double)nu0; rec[LMS_DELTA] = delta0;
      rec[LMS_ACCEPT] = 0.0; rec[LMS_SOLVED] = 0.0;
      HIPCHK(f, hipMemcpyAsync(f->lmd, rec, sizeof(double) * LMS_COUNT, hipMemcpyHostToDevice, f->st));
      rc = solve_damped_qr(f, mu0);
      if (rc < 0 && rc != LSQAMD_ENOTPD && rc != LSQAMD_EUNSUPPORTED) return rc;
      if (rc == 0) {
        HIPCHK(f, launch_lm_trial(f->st, P, f->p_dev, f->yv + P, gvec, f->dscale, f->p_trial, f->lmd));
        rc = eval_residual_launch(f, f->p_trial, true);
        if (rc) return rc;
        if (!f->lm_zero_copy) HIPCHK(f, hipMemcpyAsync(f->pin_lm, f->lmd, sizeof(double) * LMS_COUNT, hipMemcpyDeviceToHost, f->st));
        rc = wait_record(f);
        if (rc) return rc;
        if (st[LMS_SOLVED] != 0.0) f->qr_trials++;
      } else {   // no orthogonal factor either: the trial is rejected (mu grows)
        f->chol_fail++;
        rec[LMS_MU] = mu0 * (double)nu0; rec[LMS_NU] = 2.0 * (double)nu0;
        HIPCHK(f, hipMemcpyAsync(f->lmd, rec, sizeof(double) * LMS_COUNT, hipMemcpyHostToDevice, f->st));
        HIPCHK(f, hipStreamSynchronize(f->st));
      }
    } else if (st[LMS_SOLVED] == 0.0) {
      f->chol_fail++;
    }
    if (st[LMS_SOLVED] != 0.0) f->nfev++;   // (a residual evaluated at garbage is not a trial)
    f->mu = st[LMS_MU];
    f->nu = (long)st[LMS_NU];
    f->delta = st[LMS_DELTA];
    if (st[LMS_ACCEPT] != 0.0) {
      rc = run_half(f, 1, enqueue_accept);
      if (rc) return rc;
      std::swap(f->p_dev, f->p_trial);
      rc = wait_record(f);
      if (rc) return rc;
      f->chi2 = f->pin_lm[LMS_CHI2];
      f->conv_info_dev = (int32_t)f->pin_lm[LMS_INFO];
      if (!std::isfinite(f->chi2)) FAIL(f, LSQAMD_ENONFINITE, "chi2 is not finite at this point");
      return 0;
    }
    if (++bad_steps > 15) {
      // gsl_multifit_nlinear_driver tests convergence after an iteration that made no progress as
      // well, with the last (rejected) trial step as dx: at a minimum resolved to rounding that
      // step is below xtol and the fit ends on criterion 1
      HIPCHK(f, launch_lm_converge(f->st, P, f->p_dev, f->yv + P, gvec, f->opt.xtol, f->opt.gtol, f->lmd));
      if (!f->lm_zero_copy) HIPCHK(f, hipMemcpyAsync(f->pin_lm, f->lmd, sizeof(double) * LMS_COUNT, hipMemcpyDeviceToHost, f->st));
      rc = wait_record(f);
      if (rc) return rc;
      f->conv_info_dev = (int32_t)f->pin_lm[LMS_INFO];
      return LSQAMD_ENOPROG;
    }
  }
}

// one trust_iterate: GSL_SUCCESS (0) or LSQAMD_ENOPROG; negative on backend failure
int iterate(lsqamd_fit *f) {
  if (f->dev_lm && f->opt.trs == LSQAMD_TRS_LM && f->linear.empty()) return iterate_device(f);
  if (f->mirrors_stale) {
    const int rcm = refresh_mirrors(f);
    if (rcm) return rcm;
  }
  f->dev_lm = false;
  if (!f->linear.empty()) return iterate_varpro(f);
  const int64_t P = f->P;
  const int trs = f->opt.trs;
  const bool lm_family = trs == LSQAMD_TRS_LM || trs == LSQAMD_TRS_LMACCEL;
  int bad_steps = 0;
  std::vector<double> xt(P), dx(P), tmp(P);
  Legs L;
  if (!lm_family) {
    const int rc = legs_preloop(f, L);
    if (rc) return rc;
  }
  while (true) {
    double rho = -1.0, avratio = 0.0;
    bool have_step = false;
    double pred_num = 0.0;  // predicted reduction * chi2
    bool residual_done = false;
    double chi2_fast = 0.0;
    if (lm_family) {
      int rc;
      if (trs == LSQAMD_TRS_LM) {
        // plain lm: everything up to the trial chi2 is queued without a host round trip --
        // solve, x_trial = x - v on the device, residual there; one synchronisation collects
        // v, the factorisation status and chi2(x_trial)
        rc = solve_damped_launch(f, f->mu, nullptr);
        if (rc) return rc;
        HIPCHK(f, launch_trial_point(f->st, P, f->p_dev, f->yv + P, f->p_trial));
        rc = eval_residual_dev(f, f->p_trial, &chi2_fast);
        if (rc) return rc;
        rc = solve_damped_collect(f);
        if (rc == LSQAMD_ENOTPD) f->nfev--;  // that residual was evaluated at garbage: not a trial
        residual_done = rc == 0;
      } else {
        rc = solve_damped_dev(f, f->mu, f->hdiag.data());
      }
      if (rc < 0 && rc != LSQAMD_ENOTPD) return rc;
      if (rc == 0) {
        have_step = true;
        double vg = 0.0, dv2 = 0.0;
        for (int64_t j = 0; j < P; ++j) {
          dx[j] = -f->hv[j];
          vg += f->hv[j] * f->hg[j];
          const double t = f->hdiag[j] * f->hv[j];
          dv2 += t * t;
        }
        // |J v|^2 = v^T A v = v^T g - mu |D v|^2  since (A + mu D^2) v = g
        pred_num = vg + f->mu * dv2;
        if (trs == LSQAMD_TRS_LMACCEL) {
          // geodesic acceleration (lm.c lm_step, fdfvv.c): fvv by finite differences, h = 0.02
          const double h = 0.02;
          std::vector<double> vel(dx), xh(P), Gh(P), Av(P), rhs(P), acc(P);
          for (int64_t j = 0; j < P; ++j) xh[j] = f->hx[j] + h * vel[j];
          rc = grad_like_at(f, xh.data(), Gh.data());
          if (rc) return rc;
          rc = symv_host(f, vel.data(), Av.data());
          if (rc) return rc;
          // J^T fvv = (2/h) ((G_h - g)/h - A v); the acceleration solves (A + mu D^2) a = -J^T fvv
          for (int64_t j = 0; j < P; ++j) rhs[j] = (2.0 / h) * ((Gh[j] - f->hg[j]) / h - Av[j]);
          rc = solve_with_factor(f, rhs.data(), acc.data());
          if (rc) return rc;
          double an = 0.0, vn = 0.0;
          for (int64_t j = 0; j < P; ++j) {
            const double a = -acc[j];
            an += a * a;
            vn += vel[j] * vel[j];
            dx[j] = vel[j] + 0.5 * a;
          }
          avratio = std::sqrt(an) / std::sqrt(vn);
          // lm_preduction on dx: |J dx|^2 + 2 mu |D dx|^2
          rc = symv_host(f, dx.data(), tmp.data());
          if (rc) return rc;
          const double dn = scaled_norm(f->hdiag, dx);
          pred_num = dot_h(dx, tmp) + 2.0 * f->mu * dn * dn;
        }
      }
    } else {
      const int rc = legs_step(f, L, dx);
      if (rc) return rc;
      have_step = true;
      // quadratic_preduction: -(|J dx|^2 + 2 g.dx)
      const int rc2 = symv_host(f, dx.data(), tmp.data());
      if (rc2) return rc2;
      pred_num = -(dot_h(dx, tmp) + 2.0 * dot_h(f->hg, dx));
    }
    if (have_step) {
      for (int64_t j = 0; j < P; ++j) {
        f->hdx[j] = dx[j];
        xt[j] = f->hx[j] + dx[j];
      }
      double chi2_t = chi2_fast;
      if (!residual_done) {
        std::memcpy(f->pin_x, xt.data(), sizeof(double) * P);
        HIPCHK(f, hipMemcpyAsync(f->p_trial, f->pin_x, sizeof(double) * P, hipMemcpyHostToDevice, f->st));
        const int rc = eval_residual_dev(f, f->p_trial, &chi2_t);
        if (rc) return rc;
      }
      const double normf = std::sqrt(f->chi2), normf_t = std::sqrt(chi2_t);
      if (normf_t < normf) {  // NaN-safe: anything else rejects
        const double u = normf_t / normf;
        const double actual = 1.0 - u * u;
        const double pred = pred_num / f->chi2;
        rho = pred > 0.0 ? actual / pred : -1.0;
      }
    }
    if (rho > 0.75) f->delta *= f->opt.factor_up;
    else if (rho < 0.25) f->delta /= f->opt.factor_down;
    // trust_eval_step: with geodesic acceleration the step must also satisfy |a|/|v| <= avmax
    if (rho > 0.0 && !(trs == LSQAMD_TRS_LMACCEL && avratio > f->opt.avmax)) {
      const int rc = eval_normal_dev(f, f->p_trial);
      if (rc) return rc;
      f->hx = xt;
      std::swap(f->p_dev, f->p_trial);
      scale_update(f);
      const double b = 2.0 * rho - 1.0;
      f->mu *= std::fmax(0.333333333333333, 1.0 - b * b * b);
      f->nu = 2;
      return 0;
    }
    f->mu *= (double)f->nu;
    f->nu <<= 1;
    if (++bad_steps > 15) return LSQAMD_ENOPROG;
  }
}

int convergence_test(lsqamd_fit *f) {
  if (f->dev_lm) return f->conv_info_dev;
  const int64_t P = f->P;
  const double xtol = f->opt.xtol, gtol = f->opt.gtol;
  bool ok = true;
  for (int64_t j = 0; j < P; ++j)
    if (!(std::fabs(f->hdx[j]) < xtol * xtol + xtol * std::fabs(f->hx[j]))) { ok = false; break; }
  if (ok) return 1;
  double gnorm = 0.0;
  for (int64_t j = 0; j < P; ++j) {
    const double t = std::fabs(std::fmax(f->hx[j], 1.0) * f->hg[j]);
    if (t > gnorm) gnorm = t;
  }
  if (gnorm <= gtol * std::fmax(0.5 * f->chi2, 1.0)) return 2;
  return 0;
}

int do_init(lsqamd_fit *f, const double *p0) {
  int rc = ready(f);
  if (rc) return rc;
  const int64_t P = f->P;
  if (!f->linear.empty() && f->opt.trs != LSQAMD_TRS_LM)
    FAIL(f, LSQAMD_EINVAL, "linear parameters (lsqamd_set_linear) need the plain lm method");
  f->hx.assign(p0, p0 + P);
  f->hg.assign(P, 0.0);
  f->hdiag.assign(P, 1.0);
  f->hdx.assign(P, 0.0);
  f->hv.assign(P, 0.0);
  f->hcoln.assign(P, 0.0);
  f->nit = f->nfev = f->njev = f->ntrial = f->chol_fail = f->qr_trials = 0;
  f->qr_steps_on = false;
  f->logdet = NAN;
  HIPCHK(f, hipMemcpyAsync(f->p_dev, p0, sizeof(double) * P, hipMemcpyHostToDevice, f->st));
  rc = eval_normal_dev(f, f->p_dev);
  if (rc) return rc;
  f->nfev++;
  if (!f->linear.empty()) {
    // nonlinear_fit's linear= (lsqamd_set_linear): variable projection (iterate_varpro) starts from
    // the projected point -- A_aa da = -g_a with the others frozen, one more evaluation
    std::vector<double> frozen(P);
    for (int64_t j = 0; j < P; ++j) frozen[j] = f->linear[j] ? 0.0 : 1.0;
    rc = solve_damped_dev(f, 0.0, nullptr, frozen.data());
    if (rc == LSQAMD_ENOTPD) FAIL(f, LSQAMD_ENOTPD, "linear parameters: their block of J^T J is not positive definite");
    if (rc) return rc;
    for (int64_t j = 0; j < P; ++j)
      if (f->linear[j]) f->hx[j] -= f->hv[j];
    std::memcpy(f->pin_x, f->hx.data(), sizeof(double) * P);
    HIPCHK(f, hipMemcpyAsync(f->p_dev, f->pin_x, sizeof(double) * P, hipMemcpyHostToDevice, f->st));
    rc = eval_normal_dev(f, f->p_dev);
    if (rc) return rc;
    f->nfev++;
  }
  scale_init(f);
  double mx = 0.0;   // over the damped parameters (all of them unless lsqamd_set_linear was used)
  for (int64_t j = 0; j < P; ++j)
    if (f->linear.empty() || !f->linear[j]) mx = std::fmax(mx, f->hcoln[j] / f->hdiag[j]);
  f->mu = 1e-3 * mx * mx;
  f->nu = 2;
  double dxn = 0.0;
  for (int64_t j = 0; j < P; ++j) dxn += f->hdiag[j] * f->hx[j] * f->hdiag[j] * f->hx[j];
  f->delta = 0.3 * std::fmax(1.0, std::sqrt(dxn));
  f->mirrors_stale = false;
  f->conv_info_dev = 0;
  f->dev_lm = f->opt.trs == LSQAMD_TRS_LM && f->linear.empty() && getenv("LSQAMD_HOST_LM") == nullptr;
  if (f->dev_lm) {   // the record the device-side decisions start from (once per fit)
    for (int i = 0; i < LMS_COUNT; ++i) f->pin_lm[i] = 0.0;
    f->pin_lm[LMS_CHI2] = f->chi2; f->pin_lm[LMS_MU] = f->mu; f->pin_lm[LMS_NU] = (double)f->nu;
    f->pin_lm[LMS_DELTA] = f->delta;
    {   // the kernels that finish a half step mirror the record into this pinned block themselves (LSQAMD_ZERO_COPY=0: a copy per read)
      const char *zc = getenv("LSQAMD_ZERO_COPY");      // (read per call: a tested mode)
      const bool off = zc && zc[0] == '0';
      void *dp = nullptr;
      f->lm_zero_copy = !off && hipHostGetDevicePointer(&dp, f->pin_lm, 0) == hipSuccess && dp != nullptr;
      if (!f->lm_zero_copy) (void)hipGetLastError();
      long long bits = f->lm_zero_copy ? (long long)(intptr_t)dp : 0;
      std::memcpy(&f->pin_lm[LMS_HOSTPTR], &bits, sizeof(double));
      f->lm_seq_expect = 0.0;
    }
    HIPCHK(f, hipMemcpyAsync(f->lmd, f->pin_lm, sizeof(double) * LMS_COUNT, hipMemcpyHostToDevice, f->st));
    HIPCHK(f, hipStreamSynchronize(f->st));
  }
  f->initialised = true;
  return 0;
}

void fill_summary(lsqamd_fit *f, lsqamd_summary *s, int status, int info) {
  if (!s) return;
  std::memset(s, 0, sizeof(*s));
  s->status = status;
  s->info = info;
  if (info >= 0 && info <= 3) s->stopping_criterion = info;
  else if (info == 30) s->stopping_criterion = 1;
  else if (info == 31) s->stopping_criterion = 2;
  else if (info == 29) s->stopping_criterion = 3;
  else if (info == 27) s->stopping_criterion = 4;
  else if (info >= LSQAMD_INFO_TRF && info <= LSQAMD_INFO_TRF + 4) {
    static const int map[5] = {0, 2, 3, 1, 1};   // _scipy.py:176-181
    s->stopping_criterion = map[info - LSQAMD_INFO_TRF];
  } else s->stopping_criterion = 0;
  s->nit = f->nit; s->nfev = f->nfev; s->njev = f->njev; s->ntrial = f->ntrial;
  s->chol_fail = f->chol_fail;
  s->qr_trials = f->qr_trials;
  s->chi2 = f->chi2;
  s->mu = f->mu;
  s->logdet_jtj = f->logdet;
}

// f->logdet = log det of the normal matrix in f->redbuf (NaN when it has no Cholesky factor); f->cov is left alone
int logdet_normal(lsqamd_fit *f) {
  const int64_t P = f->P;
  int32_t info = 0;
  double ld = 0.0;
  HIPCHK(f, launch_build_damped(f->st, f->redbuf, P, f->ldm, 0.0, f->diag_dev, nullptr, f->M));
  HIPCHK(f, potrf_upper(f->st, f->M, P, f->ldm, P, f->chol_work, f->info_dev));
  HIPCHK(f, logdiag_sum(f->st, f->M, P, f->ldm, f->scal));
  HIPCHK(f, hipMemcpyAsync(&info, f->info_dev, sizeof(int32_t), hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipMemcpyAsync(&ld, f->scal, sizeof(double), hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipStreamSynchronize(f->st));
  f->logdet = info != 0 ? NAN : 2.0 * ld;
  return info != 0 ? LSQAMD_ENOTPD : 0;
}

// covariance + logdet at the current point: factor A (mu = 0), invert
static int do_covariance_chol(lsqamd_fit *f);
int do_covariance(lsqamd_fit *f) {
  f->cov_host_valid = false;
  f->cov_inaccurate = false;
  f->cov_dropped = 0;
  const int rc = f->opt.solver == LSQAMD_SOLVER_QR ? do_covariance_qr(f) : do_covariance_chol(f);
  if (rc == LSQAMD_ENOTPD) {
    // a rank-deficient final Jacobian: the reference's plugins return a truncated inverse there (rankdef.hip)
    const std::string keep = f->err;
    const int k = covariance_rank_deficient(f);
    if (k > 0) {
      f->cov_dropped = k;
      return 0;
    }
    f->err = keep;
  }
  return rc;
}

static int do_covariance_chol(lsqamd_fit *f) {
  const int64_t P = f->P;
  Scope sc(f, LSQAMD_T_COVAR);
  int32_t info = 0;
  double ld = 0.0;
  HIPCHK(f, launch_build_damped(f->st, f->redbuf, P, f->ldm, 0.0, f->diag_dev, nullptr, f->M));
  HIPCHK(f, potrf_upper(f->st, f->M, P, f->ldm, P, f->chol_work, f->info_dev));
  HIPCHK(f, logdiag_sum(f->st, f->M, P, f->ldm, f->scal));
  f->have_dense_A = false;  // Wl is reused below
  HIPCHK(f, trtri_upper_to_lower_T(f->st, f->M, P, f->ldm, f->chol_work, f->Wl, f->ldm));
  GemmTN g;
  g.X = f->Wl; g.Y = f->Wl; g.ldx = g.ldy = f->ldm;
  g.C = f->cov; g.ldc = f->ldm;
  g.M = P; g.N = P; g.K = P;
  g.upper_only = 1;
  g.xy_lower_tri = 1;
  HIPCHK(f, launch_gemm_tn(f->st, g));
  HIPCHK(f, launch_symmetrize_from_upper(f->st, f->cov, P, f->ldm));
  HIPCHK(f, hipMemcpyAsync(&info, f->info_dev, sizeof(int32_t), hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipMemcpyAsync(&ld, f->scal, sizeof(double), hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipStreamSynchronize(f->st));
  f->have_cov = true;
  if (info != 0) {
    f->logdet = NAN;
    char b[160];
    snprintf(b, sizeof(b), "J^T J is not positive definite at the solution (pivot %d); covariance undefined", info);
    f->err = b;
    return LSQAMD_ENOTPD;
  }
  f->logdet = 2.0 * ld;
  return 0;
}

}  // namespace lsqamd_host

using namespace lsqamd_host;

// =========================================================================================
extern "C" {

int lsqamd_abi_version(void) { return LSQAMD_ABI_VERSION; }

int lsqamd_query_devices(int32_t *count, int32_t index, char *arch, size_t cap, int64_t *hbm_bytes) try {
  if (!count) return LSQAMD_EINVAL;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;   // no driver / no GPU: zero devices
  *count = n;
  if (arch && cap) arch[0] = 0;
  if (hbm_bytes) *hbm_bytes = 0;
  if (index < 0 || index >= n) return 0;
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, index) != hipSuccess) return LSQAMD_EHIP;
  if (arch && cap) snprintf(arch, cap, "%s", pr.gcnArchName);
  if (hbm_bytes) *hbm_bytes = (int64_t)pr.totalGlobalMem;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)

size_t lsqamd_workspace_bytes(const lsqamd_config *cfg) try {
  if (check_cfg(cfg) != 0) return 0;
  lsqamd_fit tmp;
  tmp.cfg = *cfg;
  return carve(&tmp, nullptr, 0, true);
} LSQAMD_ABI_CATCH((void)lsqamd::abi_exception(nullptr); return 0;)

int lsqamd_create(const lsqamd_config *cfg, void *dev_workspace, size_t workspace_bytes, void *stream,
                  lsqamd_fit **out) try {
  if (!out) return LSQAMD_EINVAL;
  *out = nullptr;
  const int rc = check_cfg(cfg);
  if (rc) return rc;
  if (!dev_workspace || (reinterpret_cast<uintptr_t>(dev_workspace) & 255)) return LSQAMD_EINVAL;
  lsqamd_fit *f = new (std::nothrow) lsqamd_fit;
  if (!f) return LSQAMD_ENOMEM;
  f->cfg = *cfg;
  f->st = reinterpret_cast<hipStream_t>(stream);
  f->st_used = true;
  f->dev = current_device();
  const size_t need = carve(f, dev_workspace, workspace_bytes, true);
  if (need > workspace_bytes) {
    delete f;
    return LSQAMD_ENOMEM;
  }
  carve(f, dev_workspace, workspace_bytes, false);
  f->opt.xtol = 1e-8; f->opt.gtol = 1e-10; f->opt.ftol = 1e-10;   // __init__.py:102, _gsl.pyx:595-596
  f->opt.maxit = 1000;
  f->opt.scaler = LSQAMD_SCALE_MORE;
  f->opt.solver = LSQAMD_SOLVER_CHOLESKY;
  f->opt.factor_up = 3.0;
  f->opt.factor_down = 2.0;
  f->opt.trs = LSQAMD_TRS_LM;
  f->opt.avmax = 0.75;
  {
    const size_t P1 = (size_t)f->P + 1;
    const size_t head = (5 * P1 + 8 + 15) / 16 * 16;     // (the two regions the device writes start on 128-byte lines)
    f->pin = static_cast<double *>(pinned_take(sizeof(double) * (head + LMS_COUNT + 1280), &f->pin_bytes));
    if (!f->pin) {
      delete f;
      return LSQAMD_ENOMEM;
    }
    f->pin_g = f->pin; f->pin_c = f->pin_g + P1; f->pin_v = f->pin_c + P1; f->pin_d = f->pin_v + P1;
    f->pin_x = f->pin_d + P1; f->pin_s = f->pin_x + P1; f->pin_lm = f->pin + head;
    f->pin_fit = f->pin_lm + LMS_COUNT;       // 1280 doubles: what the one-launch fit kernel publishes (jit.h FitArgs::pub)
    static_assert(lsqamd_jit::FIT_HOST_DOUBLES <= 1280, "pin_fit");
    for (int i = 0; i < LMS_COUNT; ++i) f->pin_lm[i] = 0.0;
    f->pin_fit[16] = 0.0;
  }
  if (hipMemsetAsync(f->in_block, 0, (size_t)(f->N > 0 ? f->N : 1), f->st) != hipSuccess) {
    delete f;
    return LSQAMD_EHIP;
  }
  if (hipMemsetAsync(f->M, 0, sizeof(double) * (size_t)(f->P * f->ldm), f->st) != hipSuccess) {
    delete f;
    return LSQAMD_EHIP;
  }
  {
    std::vector<int32_t> wm(4 * (size_t)f->syrk_nwork);
    syrk_work_fill(f->P, f->splits, wm.data());
    Upload up(f);     // (a local: staged through pinned memory when small, else waited for)
    if (up(f->syrk_map, wm.data(), wm.size() * sizeof(int32_t)) != hipSuccess || up.finish() != hipSuccess) {
      delete f;
      return LSQAMD_EHIP;
    }
    // LSQAMD_EXCHANGE_GROUPS = G (default 1): groups of tile rows with (about) equal tile counts -- the last one holds
    // LSQAMD_EXCHANGE_TAIL_PCT per cent of the tiles when given (its exchange is the one nothing can hide)
    const int T = (int)((f->P + 127) / 128);
    int G = 1;
    if (const char *e = getenv("LSQAMD_EXCHANGE_GROUPS")) G = atoi(e);
    if (G > 8) G = 8;
    if (G > T) G = T;
    if (G < 1 || f->P <= 64) G = 1;
    f->xg = G;
    if (G > 1) {
      const int64_t total = (int64_t)T * (T + 1) / 2;
      double tail = 1.0 / G;
      if (const char *e = getenv("LSQAMD_EXCHANGE_TAIL_PCT")) {
        const double v = atof(e) / 100.0;
        if (v > 0.0 && v < 1.0) tail = v;
      }
      int row = 0;
      int64_t tiles = 0;
      f->xg_row[0] = 0;
      for (int g = 1; g < G; ++g) {
        const double want = (1.0 - tail) * g / (G - 1) * (double)total;
        while (row < T - (G - g) && (double)(tiles + (T - row) / 2) < want) { tiles += T - row; ++row; }
        if (row <= f->xg_row[g - 1]) { tiles += T - row; ++row; }      // every group holds at least one tile row
        // the work list walks 8 x 8 PATCHES of tiles (two row panels serve eight tiles each way in the L2): a group
        // boundary inside a patch row leaves 1 x 8 slivers on both sides -- measured + 5 % on the product at the shard
        // shape.  Snap to the nearest multiple of 8 tile rows where that keeps the groups distinct
        if (T >= 16) {
          int snapped = (row + 4) / 8 * 8;
          if (snapped <= f->xg_row[g - 1]) snapped = f->xg_row[g - 1] + 8;
          if (snapped > f->xg_row[g - 1] && snapped <= T - (G - g)) {
            while (row < snapped) { tiles += T - row; ++row; }
            while (row > snapped) { --row; tiles -= T - row; }
          }
        }
        f->xg_row[g] = row;
      }
      f->xg_row[G] = T;
      const char *mode = getenv("LSQAMD_EXCHANGE_MODE");
      f->xg_signal = !(mode && mode[0] == 's' && mode[1] == 'p');          // "split": one product launch per group
      int64_t o = 0;
      for (int g = 0; g < G; ++g) {
        const int r0 = f->xg_row[g];
        f->xg_tile[g] = (int64_t)r0 * T - (int64_t)r0 * (r0 - 1) / 2;
      }
      f->xg_tile[G] = total;
      if (f->xg_signal) {
        syrk_work_fill_grouped(f->P, f->splits, G, f->xg_row, wm.data(), f->xg_count);
        for (int g = 0; g < G; ++g) o += f->xg_count[g];
        if (hipMemsetAsync(f->xg_ctr, 0, 16 * sizeof(int32_t), f->st) != hipSuccess) { delete f; return LSQAMD_EHIP; }
      } else {
        for (int g = 0; g < G; ++g) {
          f->xg_work[g] = (int32_t)o;
          o += syrk_work_fill_rows(f->P, f->splits, f->xg_row[g], f->xg_row[g + 1], wm.data() + 4 * o);
        }
        f->xg_work[G] = (int32_t)o;
      }
      if (o != f->syrk_nwork || up(f->syrk_map_g, wm.data(), wm.size() * sizeof(int32_t)) != hipSuccess || up.finish() != hipSuccess) {
        delete f;
        return LSQAMD_EHIP;
      }
    }
  }
  *out = f;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)

int lsqamd_destroy(lsqamd_fit *fit) try {
  if (!fit) return 0;
  (void)hipStreamSynchronize(fit->st);
  resolve_timers(fit);
  delete fit;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)

const char *lsqamd_last_error(const lsqamd_fit *fit) { return fit ? fit->err.c_str() : "null handle"; }

int lsqamd_set_x(lsqamd_fit *f, const double *x, int64_t n_rows, int32_t n_x) try {
  if (!f) return LSQAMD_EINVAL;
  if (!x || n_rows != f->N || n_x != (f->cfg.n_x > 0 ? f->cfg.n_x : 1))
    FAIL(f, LSQAMD_EINVAL, "set_x: expected %lld x %d", (long long)f->N, f->cfg.n_x);
  {
    Upload up(f);
    HIPCHK(f, up(f->x, x, sizeof(double) * n_rows * n_x));
    HIPCHK(f, up.finish());
  }
  f->drop_step_graphs();   // captured steps bake xmax (and the path choices behind have_x) in by value
  f->xmax = 0.0;
  for (int64_t i = 0; i < n_rows * n_x; ++i) {
    const double ax = std::fabs(x[i]);
    if (!(ax <= f->xmax)) f->xmax = ax;      // (NaN sticks: the flag then always says "far")
  }
  f->have_x = true;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

// stack discipline and operand ranges of one RPN program (host side)
static int validate_tape(lsqamd_fit *f, const int32_t *code, int32_t n_code, int32_t n_consts) {
  int sp = 0;
  for (int t = 0; t < n_code; ++t) {
    const int op = code[t] & 0xff, arg = code[t] >> 8;
    if (op == LSQAMD_OP_CONST) { if (arg < 0 || arg >= n_consts) FAIL(f, LSQAMD_EINVAL, "tape: bad const index"); ++sp; }
    else if (op == LSQAMD_OP_X) { if (arg < 0 || arg >= (f->cfg.n_x > 0 ? f->cfg.n_x : 1)) FAIL(f, LSQAMD_EINVAL, "tape: bad x index"); ++sp; }
    else if (op == LSQAMD_OP_P) { if (arg < 0 || arg >= f->P) FAIL(f, LSQAMD_EINVAL, "tape: bad parameter index"); ++sp; }
    else if (op >= LSQAMD_OP_ADD && op <= LSQAMD_OP_POW) { if (sp < 2) FAIL(f, LSQAMD_EINVAL, "tape: stack underflow"); --sp; }
    else if (op >= LSQAMD_OP_NEG && op <= LSQAMD_OP_LAST) { if (sp < 1) FAIL(f, LSQAMD_EINVAL, "tape: stack underflow"); }
    else FAIL(f, LSQAMD_EINVAL, "tape: unknown opcode %d", op);
    if (sp > LSQAMD_TAPE_MAX_STACK) FAIL(f, LSQAMD_EINVAL, "tape: stack deeper than %d", LSQAMD_TAPE_MAX_STACK);
  }
  if (sp != 1) FAIL(f, LSQAMD_EINVAL, "tape: must leave exactly one value");
  return 0;
}

int lsqamd_set_tape(lsqamd_fit *f, const int32_t *code, int32_t n_code, const double *consts,
                    int32_t n_consts) try {
  if (!f) return LSQAMD_EINVAL;
  f->drop_step_graphs();
  if (f->st_used) (void)hipStreamSynchronize(f->st);      // (work queued on this handle may still run the kernel about to be released)
  f->drop_jit();
  f->progs.clear();
  f->used_nrm = false;
  f->J_stale = false;
  if (!code || n_code < 1 || n_code > f->tape_cap || n_consts < 0 || n_consts > 1024)
    FAIL(f, LSQAMD_EINVAL, "set_tape: 1..%d instructions (lsqamd_config.tape_len), <= 1024 constants", f->tape_cap);
  {
    const int rcv = validate_tape(f, code, n_code, n_consts);
    if (rcv) return rcv;
  }
  {
    std::vector<int32_t> poff((size_t)n_code + 1);
    int32_t slots = 0;
    for (int t = 0; t < n_code; ++t) {
      poff[(size_t)t] = slots;
      slots += tape_slots_of_op(code[t] & 0xff);
    }
    poff[(size_t)n_code] = slots;
    f->tape_slots = slots;
    // Is the root a sum S_1 + S_2 + ... ?  The joining ADDs are the ones that take the stack from depth 2 to 1;
    // S_1 ends where the depth is 1 for the last time before the first of them.  Every segment must fit the
    // kernel's LDS store of local partials; otherwise (or if anything follows the last join) the whole-tape
    // kernel runs.
    std::vector<int32_t> seg;      // (first, last, sign) per segment
    {
      std::vector<int> depth_after((size_t)n_code);
      int d = 0;
      std::vector<int> joins;
      for (int t = 0; t < n_code; ++t) {
        const int op = code[t] & 0xff;
        if (op <= LSQAMD_OP_P) ++d;
        else if (op <= LSQAMD_OP_POW) {
          if ((op == LSQAMD_OP_ADD || op == LSQAMD_OP_SUB) && d == 2) joins.push_back(t);
          --d;
        }
        depth_after[(size_t)t] = d;
      }
      bool ok = !joins.empty() && joins.back() == n_code - 1;
      if (ok) {
        int split = -1;
        for (int t = joins[0] - 1; t >= 0; --t)
          if (depth_after[(size_t)t] == 1) { split = t; break; }
        ok = split >= 0 && split + 1 <= joins[0] - 1;
        if (ok) {
          seg.insert(seg.end(), {0, split, 1});
          for (size_t k = 0; k < joins.size(); ++k) {
            const int lo = k == 0 ? split + 1 : joins[k - 1] + 1, hi = joins[k] - 1;
            if (lo > hi) { ok = false; break; }
            seg.insert(seg.end(), {lo, hi, (code[joins[k]] & 0xff) == LSQAMD_OP_SUB ? -1 : 1});
          }
        }
      }
      int max_depth = 1, max_slots = 1;
      for (size_t k = 0; ok && k + 2 < seg.size(); k += 3) {
        const int nsl = poff[(size_t)seg[k + 1] + 1] - poff[(size_t)seg[k]];
        if (nsl > 32) ok = false;     // TAPE_SEG_SLOTS
        if (nsl > max_slots) max_slots = nsl;
        int sd = 0;                              // every segment is a complete expression of its own
        for (int t = seg[k]; ok && t <= seg[k + 1]; ++t) {
          const int op = code[t] & 0xff;
          if (op <= LSQAMD_OP_P) ++sd;
          else if (op <= LSQAMD_OP_POW) { if (sd < 2) ok = false; --sd; }
          else if (sd < 1) ok = false;
          if (sd > max_depth) max_depth = sd;
        }
        if (sd != 1) ok = false;
      }
      if (!ok) seg.clear();
      f->tape_seg_depth = max_depth + 1;         // (the reverse sweep pushes one adjoint before it pops)
      f->tape_seg_slots = max_slots;
      // a parameter read once has one contribution to its column: a plain store, no atomic; if that is true of
      // all of them and none is missing, the transposed Jacobian needs no zero fill either
      std::vector<int> uses((size_t)f->P, 0);
      for (int t = 0; t < n_code; ++t)
        if ((code[t] & 0xff) == LSQAMD_OP_P) ++uses[(size_t)(code[t] >> 8)];
      int most = 0, least = 1 << 30;
      for (int u : uses) { if (u > most) most = u; if (u < least) least = u; }
      f->tape_single = most <= 1 ? (least == 1 ? 2 : 1) : 0;
    }
    f->tape_n_seg = (int32_t)(seg.size() / 3);
    Upload up(f);
    HIPCHK(f, up(f->tape_poff, poff.data(), sizeof(int32_t) * (n_code + 1)));
    if (!seg.empty()) HIPCHK(f, up(f->tape_seg, seg.data(), sizeof(int32_t) * seg.size()));
    HIPCHK(f, up.finish());   // (poff, seg are locals: staged copies or a wait)
  }
  {
    Upload up(f);
    HIPCHK(f, up(f->tape, code, sizeof(int32_t) * n_code));
    if (n_consts > 0) HIPCHK(f, up(f->consts, consts, sizeof(double) * n_consts));
    HIPCHK(f, up.finish());
  }
  f->n_tape = n_code;
  f->have_tape = true;
  // the formula as straight-line code for gfx950 (hiprtc, cached by content); without it the interpreter
  // kernels above run -- lsqamd_debug_flags bit 3 says which
  f->jit_why.clear();
  f->jit = lsqamd_jit::compile_tape(code, n_code, consts, n_consts, (int)f->P, f->cfg.n_x > 0 ? f->cfg.n_x : 1, f->jit_why);
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_set_tape_programs(lsqamd_fit *f, int32_t n_prog, const int64_t *row0, const int32_t *code,
                             const int32_t *code_off, const double *consts, int32_t n_consts) try {
  if (!f) return LSQAMD_EINVAL;
  f->drop_step_graphs();
  f->used_nrm = false;
  f->J_stale = false;
  if (f->cfg.model != LSQAMD_MODEL_TAPE) FAIL(f, LSQAMD_EINVAL, "set_tape_programs: the handle's model is not LSQAMD_MODEL_TAPE");
  if (n_prog < 1 || !row0 || !code || !code_off || n_consts < 0 || n_consts > 1024 || (n_consts > 0 && !consts))
    FAIL(f, LSQAMD_EINVAL, "set_tape_programs: n_prog >= 1, row0[n_prog + 1], code, code_off[n_prog + 1], <= 1024 constants");
  if (row0[0] != 0 || row0[n_prog] != f->N || code_off[0] != 0 || code_off[n_prog] < 1 || code_off[n_prog] > f->tape_cap)
    FAIL(f, LSQAMD_EINVAL, "set_tape_programs: the row ranges must tile 0..%lld and the programs hold 1..%d instructions in all "
                           "(lsqamd_config.tape_len)", (long long)f->N, f->tape_cap);
  std::vector<lsqamd::TapeProgram> progs((size_t)n_prog);
  for (int i = 0; i < n_prog; ++i) {
    if (row0[i + 1] < row0[i] || code_off[i + 1] <= code_off[i])
      FAIL(f, LSQAMD_EINVAL, "set_tape_programs: row0 must not decrease and every program needs at least one instruction");
    const int rcv = validate_tape(f, code + code_off[i], code_off[i + 1] - code_off[i], n_consts);
    if (rcv) return rcv;
    progs[(size_t)i].row0 = row0[i];
    progs[(size_t)i].n_rows = row0[i + 1] - row0[i];
    progs[(size_t)i].tape_off = code_off[i];
    progs[(size_t)i].n_tape = code_off[i + 1] - code_off[i];
  }
  HIPCHK(f, hipMemcpyAsync(f->tape, code, sizeof(int32_t) * code_off[n_prog], hipMemcpyHostToDevice, f->st));
  if (n_consts > 0)
    HIPCHK(f, hipMemcpyAsync(f->consts, consts, sizeof(double) * n_consts, hipMemcpyHostToDevice, f->st));
  HIPCHK(f, hipStreamSynchronize(f->st));
  // every formula compiled on its own (hiprtc, cached by content); one that cannot be runs through the
  // forward-mode interpreter kernel over its rows
  f->drop_jit();
  f->progs.clear();
  f->jit_why.clear();
  int compiled = 0;
  for (int i = 0; i < n_prog; ++i) {
    std::string why;
    progs[(size_t)i].jit = lsqamd_jit::compile_tape(code + code_off[i], progs[(size_t)i].n_tape, consts, n_consts, (int)f->P,
                                                    f->cfg.n_x > 0 ? f->cfg.n_x : 1, why);
    if (progs[(size_t)i].jit) ++compiled;
    else f->jit_why = why;
  }
  f->progs = std::move(progs);
  f->progs_compiled = compiled;
  f->n_tape = code_off[n_prog];
  f->tape_n_seg = 0;
  f->have_tape = true;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_set_data(lsqamd_fit *f, const double *ymean, const double *wdiag, int32_t n_blocks,
                    const int64_t *block_row0, const int64_t *block_size, const int64_t *block_modes,
                    const int32_t *block_tri, const double *wt) try {
  if (!f) return LSQAMD_EINVAL;
  f->drop_step_graphs();
  if (n_blocks != f->cfg.n_blocks) FAIL(f, LSQAMD_EINVAL, "set_data: n_blocks differs from the config");
  if (f->N > 0 && (!ymean || !wdiag)) FAIL(f, LSQAMD_EINVAL, "set_data: null ymean/wdiag");
  if (n_blocks > 0 && (!block_row0 || !block_size || !block_modes || !wt))
    FAIL(f, LSQAMD_EINVAL, "set_data: null block arrays with n_blocks = %d", n_blocks);
  const int64_t N = f->N;
  std::vector<uint8_t> inb((size_t)(N > 0 ? N : 1), 0);
  f->h_row0.assign(block_row0, block_row0 + n_blocks);
  f->h_size.assign(block_size, block_size + n_blocks);
  f->h_modes.assign(block_modes, block_modes + n_blocks);
  f->h_tri.assign(n_blocks, 0);
  f->h_woff.assign(n_blocks, 0);
  int64_t off = 0, prev_end = 0, maxb = 0;
  bool all_tri = true;
  f->uniform_blocks = n_blocks > 0;
  for (int b = 0; b < n_blocks; ++b) {
    const int64_t r0 = f->h_row0[b], B = f->h_size[b];
    if (B < 1 || r0 < prev_end || r0 + B > N || f->h_modes[b] < 0 || f->h_modes[b] > B)
      FAIL(f, LSQAMD_EINVAL, "set_data: block %d is not a valid ascending contiguous row range", b);
    prev_end = r0 + B;
    if (B > maxb) maxb = B;
    f->h_tri[b] = block_tri ? block_tri[b] : 0;
    f->h_woff[b] = off;
    off += B * B;
    for (int64_t i = 0; i < B; ++i) inb[(size_t)(r0 + i)] = 1;
    if (B != f->h_size[0] || r0 != f->h_row0[0] + b * f->h_size[0]) f->uniform_blocks = false;
    if (!f->h_tri[b]) all_tri = false;
  }
  f->uniform_tri = all_tri ? 1 : 0;  // mixed blocks run batched without the triangular shortcut
  if (off > f->cfg.sum_block_sq || maxb > f->cfg.max_block)
    FAIL(f, LSQAMD_EINVAL, "set_data: blocks exceed the sizes promised in the config");
  Upload up(f);
  if (N > 0) {
    HIPCHK(f, up(f->ymean, ymean, sizeof(double) * N));
    HIPCHK(f, up(f->wdiag, wdiag, sizeof(double) * N));
    HIPCHK(f, up(f->in_block, inb.data(), (size_t)N));
  }
  if (n_blocks > 0) {
    if (!wt) FAIL(f, LSQAMD_EINVAL, "set_data: null block weights");
    HIPCHK(f, up(f->blk_row0, f->h_row0.data(), sizeof(int64_t) * n_blocks));
    HIPCHK(f, up(f->blk_size, f->h_size.data(), sizeof(int64_t) * n_blocks));
    HIPCHK(f, up(f->blk_woff, f->h_woff.data(), sizeof(int64_t) * n_blocks));
    HIPCHK(f, up(f->wt, wt, sizeof(double) * off, true));   // host or device source
  }
  HIPCHK(f, up.finish());
  f->have_data = true;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_set_ymean(lsqamd_fit *f, const double *ymean) try {
  if (!f || !ymean) return LSQAMD_EINVAL;
  if (!f->have_data) FAIL(f, LSQAMD_EINVAL, "set_ymean: call lsqamd_set_data first");
  {
    Upload up(f);
    if (f->N > 0) HIPCHK(f, up(f->ymean, ymean, sizeof(double) * f->N));
    HIPCHK(f, up.finish());
  }
  f->have_cov = false;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_set_prior(lsqamd_fit *f, const double *mean, const double *prec) try {
  if (!f) return LSQAMD_EINVAL;
  if (!f->cfg.has_prior) FAIL(f, LSQAMD_EINVAL, "set_prior: the config says has_prior = 0");
  if (!mean || !prec) FAIL(f, LSQAMD_EINVAL, "set_prior: null argument");
  const int64_t P = f->P;
  {
    Upload up(f);
    HIPCHK(f, up(f->prior_mean, mean, sizeof(double) * P));
    HIPCHK(f, up(f->prior_prec, prec, sizeof(double) * (f->cfg.prior_dense ? P * P : P), true));   // host or device source
    HIPCHK(f, up.finish());
  }
  f->have_prior = true;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_set_options(lsqamd_fit *f, const lsqamd_options *opt) try {
  if (!f || !opt) return LSQAMD_EINVAL;
  if (opt->xtol < 0 || opt->gtol < 0 || opt->maxit < 0) FAIL(f, LSQAMD_EINVAL, "set_options: negative tolerance/maxit");
  if (opt->scaler < 0 || opt->scaler > LSQAMD_SCALE_MARQUARDT) FAIL(f, LSQAMD_EINVAL, "set_options: unknown scaler");
  if (opt->solver != LSQAMD_SOLVER_CHOLESKY && opt->solver != LSQAMD_SOLVER_QR) FAIL(f, LSQAMD_EINVAL, "set_options: unknown solver");
  if (!(opt->factor_up > 1.0) || !(opt->factor_down > 1.0)) FAIL(f, LSQAMD_EINVAL, "set_options: factors must exceed 1");
  if (opt->trs < LSQAMD_TRS_LM || opt->trs > LSQAMD_TRS_MINPACK_LM) FAIL(f, LSQAMD_EINVAL, "set_options: unknown trust-region method");
  f->opt = *opt;
  f->drop_step_graphs();   // tolerances, factors and the scaler are baked into the captured nodes
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

size_t lsqamd_qr_work_bytes(const lsqamd_fit *f) { return f ? qr_work_bytes(f) : 0; }

int lsqamd_set_qr_work(lsqamd_fit *f, void *dev_work, size_t work_bytes) try {
  if (!f) return LSQAMD_EINVAL;
  if (dev_work && work_bytes < qr_work_bytes(f)) FAIL(f, LSQAMD_ENOMEM, "set_qr_work: need %zu bytes", qr_work_bytes(f));
  f->qr_work = dev_work;
  f->qr_work_bytes = dev_work ? work_bytes : 0;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_qr_info(const lsqamd_fit *f, int32_t *passes, double *delta) try {
  if (!f) return LSQAMD_EINVAL;
  if (passes) *passes = f->qr_passes;
  if (delta) *delta = f->qr_delta;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)

int lsqamd_set_bounds(lsqamd_fit *f, const double *lower, const double *upper) try {
  if (!f) return LSQAMD_EINVAL;
  if (!lower && !upper) {
    f->lb.clear();
    f->ub.clear();
    return 0;
  }
  const int64_t P = f->P;
  std::vector<double> lb(P, -INFINITY), ub(P, INFINITY);
  for (int64_t j = 0; j < P; ++j) {
    if (lower) lb[j] = lower[j];
    if (upper) ub[j] = upper[j];
    if (!(lb[j] < ub[j]))
      FAIL(f, LSQAMD_EINVAL, "set_bounds: each lower bound must be strictly less than each upper bound (parameter %lld)",
           (long long)j);
  }
  f->lb.swap(lb);
  f->ub.swap(ub);
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_set_loss(lsqamd_fit *f, int32_t loss, double f_scale) try {
  if (!f) return LSQAMD_EINVAL;
  if (loss < LSQAMD_LOSS_LINEAR || loss > LSQAMD_LOSS_ARCTAN) FAIL(f, LSQAMD_EINVAL, "set_loss: `loss` must be linear, soft_l1, huber, cauchy or arctan");
  if (!(f_scale > 0.0) || !std::isfinite(f_scale)) FAIL(f, LSQAMD_EINVAL, "set_loss: f_scale must be positive");
  f->loss = loss;
  f->f_scale = f_scale;
  f->drop_step_graphs();
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_set_x_scale(lsqamd_fit *f, const double *x_scale) try {
  if (!f) return LSQAMD_EINVAL;
  std::vector<double> v;
  if (x_scale) {
    v.assign(x_scale, x_scale + f->P);
    for (int64_t j = 0; j < f->P; ++j)
      if (!(v[j] > 0.0) || !std::isfinite(v[j])) FAIL(f, LSQAMD_EINVAL, "set_x_scale: `x_scale` must be 'jac' or array_like with positive numbers");
  }
  f->x_scale.swap(v);
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_set_param_rows(lsqamd_fit *f, const int32_t *row_param) try {
  if (!f) return LSQAMD_EINVAL;
  f->drop_step_graphs();
  if (!row_param) {
    f->have_param_rows = false;
    return 0;
  }
  if (f->cfg.has_prior) FAIL(f, LSQAMD_EINVAL, "set_param_rows: the prior is given as rows OR through lsqamd_set_prior, not both");
  bool any = false;
  for (int64_t i = 0; i < f->N; ++i) {
    if (row_param[i] >= f->P) FAIL(f, LSQAMD_EINVAL, "set_param_rows: row %lld names parameter %d", (long long)i, row_param[i]);
    any |= row_param[i] >= 0;
  }
  if (f->N > 0) {
    HIPCHK(f, hipMemcpyAsync(f->row_param, row_param, sizeof(int32_t) * f->N, hipMemcpyHostToDevice, f->st));
    HIPCHK(f, hipStreamSynchronize(f->st));
  }
  f->have_param_rows = any;
  f->initialised = false;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_set_linear(lsqamd_fit *f, const int32_t *index, int32_t n) try {
  if (!f || n < 0 || (n > 0 && !index)) return LSQAMD_EINVAL;
  std::vector<char> mask;
  if (n > 0) {
    mask.assign(f->P, 0);
    for (int32_t k = 0; k < n; ++k) {
      if (index[k] < 0 || index[k] >= f->P) FAIL(f, LSQAMD_EINVAL, "set_linear: index %d out of range", index[k]);
      mask[index[k]] = 1;
    }
  }
  f->linear.swap(mask);
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_set_reduce(lsqamd_fit *f, lsqamd_reduce_fn fn, void *user) try {
  if (!f) return LSQAMD_EINVAL;
  f->reduce = fn;
  f->reduce_user = user;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

// rank that contributes the (replicated) prior terms before the all-reduce; default: this one
int lsqamd_set_adds_prior(lsqamd_fit *f, int32_t on) try {
  if (!f) return LSQAMD_EINVAL;
  f->adds_prior = on != 0;
  f->drop_step_graphs();
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_init(lsqamd_fit *f, const double *p0) try {
  if (!f || !p0) return LSQAMD_EINVAL;
  return do_init(f, p0);
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_step(lsqamd_fit *f, int32_t *info) try {
  if (!f) return LSQAMD_EINVAL;
  if (const int rcd = device_ok(f)) return rcd;
  if (!f->initialised) FAIL(f, LSQAMD_EINVAL, "lsqamd_step before lsqamd_init");
  if (f->opt.trs >= LSQAMD_TRS_TRF) FAIL(f, LSQAMD_EUNSUPPORTED, "lsqamd_step: the scipy-plugin methods (trf, dogbox, minpack lm) run through lsqamd_run only");
  const int rc = iterate(f);
  if (rc < 0) return rc;
  f->nit++;
  if (f->timing) resolve_timers(f, true);
  if (rc == LSQAMD_ENOPROG) {
    if (info) *info = LSQAMD_ENOPROG;
    return LSQAMD_ENOPROG;
  }
  if (info) *info = convergence_test(f);
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_finish(lsqamd_fit *f, lsqamd_summary *out) try {
  if (!f) return LSQAMD_EINVAL;
  if (!f->initialised) FAIL(f, LSQAMD_EINVAL, "lsqamd_finish before lsqamd_init");
  const int rc = do_covariance(f);
  if (f->timing) resolve_timers(f);
  if (f->comm && f->xg > 1 && f->xg_signal) {     // did a wait of the grouped exchange give up? (bounded spin: never a hang)
    int32_t to = 0;
    HIPCHK(f, hipMemcpyAsync(&to, f->xg_ctr + 8, sizeof to, hipMemcpyDeviceToHost, f->st));
    HIPCHK(f, hipStreamSynchronize(f->st));
    if (to) FAIL(f, LSQAMD_EREDUCE, "grouped exchange: the wait for a group of J^T J tiles timed out (the sums of this fit are not to be trusted)");
  }
  fill_summary(f, out, 0, 0);
  if (out) out->cov_status = rc == LSQAMD_ENOTPD ? rc : (rc == 0 && f->cov_inaccurate ? LSQAMD_EINACCURATE : (rc == 0 ? f->cov_dropped : 0));
  return rc == LSQAMD_ENOTPD ? 0 : rc;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

// A whole small fit in ONE launch (jit.hip lsqamd_jit_lm): a compiled formula with at most a dozen parameters, uncorrelated rows,
// a few thousand of them at most, plain lm on one rank -- the fits of examples/nist.py and tests/test_lsqfit.py, where the
// half dozen launches and two host round trips of an iteration of iterate_device cost more than its arithmetic.  One workgroup
// runs gsl_multifit_nlinear_init + _driver (src/lsqfit/_gsl.pyx:676-677) from p0 to the stopping criterion -- and, for the
// normal-equation route, gsl_multifit_nlinear_covar (:706) -- and hands back the state where do_init / iterate_device
// would have left it (device buffers and host mirrors).  -> 1: done (iter, info set); 0: not this fit's route, or the kernel met
// something irregular and the general path runs the fit from the start; < 0: error.  LSQAMD_ONE_LAUNCH_FIT=0 disables (read per call).
static int run_one_launch(lsqamd_fit *f, const double *p0, int *iter, int *info) {
  const char *e = getenv("LSQAMD_ONE_LAUNCH_FIT");
  if (e && e[0] == '0') return 0;
  const int64_t P = f->P;
  const lsqamd_jit::Kernel *k = static_cast<const lsqamd_jit::Kernel *>(f->jit);
  if (!lsqamd_jit::has_fit_kernel(k) || P > lsqamd_jit::FIT_MAX_P || f->N < 1 ||
      f->N > lsqamd_jit::fit_row_limit(k, f->cfg.n_blocks != 0)) return 0;
  if (f->opt.trs != LSQAMD_TRS_LM || !f->linear.empty() || getenv("LSQAMD_HOST_LM") || f->opt.maxit < 1) return 0;
  if (f->opt.maxit > 20000) return 0;      // (a kernel cannot be interrupted: runs of that length stay on the host-driven path)
  if (f->comm || f->reduce || f->timing || !small_fuse(f) || !f->progs.empty() || f->have_param_rows) return 0;
  // correlated rows: the workgroup whitens them itself (one row per thread, the raw rows in LDS) -- up to 256 rows in all
  if (f->cfg.n_blocks > 64) return 0;
  if (f->cfg.has_prior && !f->adds_prior) return 0;
  int rc = ready(f);
  if (rc) return rc;
  void *dfit = nullptr, *dx = nullptr, *dlm = nullptr;
  if (!f->fit_block || hipHostGetDevicePointer(&dfit, f->pin_fit, 0) != hipSuccess || hipHostGetDevicePointer(&dx, f->pin_x, 0) != hipSuccess ||
      hipHostGetDevicePointer(&dlm, f->pin_lm, 0) != hipSuccess || !dfit || !dx || !dlm) {
    (void)hipGetLastError();
    return 0;
  }
  // LSQAMD_ZERO_COPY=0 (read per call: a tested mode): nothing is published to host memory; the host waits for the stream and
  // copies the kernel's device block -- also what happens when a published block does not verify in time
  const char *z = getenv("LSQAMD_ZERO_COPY");
  const bool zc_off = z && z[0] == '0';
  const int nw = lsqamd_jit::fit_host_cov((int)P) + (int)(P * P);       // words of the record block of a P-parameter fit
  static std::atomic<unsigned> g_seq{0};
  unsigned long long seq = 0;
  while (seq == 0) seq = (g_seq.fetch_add(1, std::memory_order_relaxed) + 1u) & 0xffffffu;   // 24 bits, never 0
  std::memcpy(f->pin_x, p0, sizeof(double) * P);
  volatile unsigned long long *hw = reinterpret_cast<volatile unsigned long long *>(f->pin_fit);
  if (poison_pinned()) poison_fill(f->pin_fit, sizeof(double) * lsqamd_jit::FIT_HOST_DOUBLES);
  hw[16] = 0;
  std::atomic_thread_fence(std::memory_order_release);
  lsqamd_jit::FitArgs a;
  a.x = f->x; a.ymean = f->ymean; a.wdiag = f->wdiag; a.n_data = f->N;
  a.in_block = f->in_block; a.wt = f->wt;
  a.blk_row0 = reinterpret_cast<const long long *>(f->blk_row0); a.blk_size = reinterpret_cast<const long long *>(f->blk_size);
  a.blk_woff = reinterpret_cast<const long long *>(f->blk_woff); a.n_blocks = f->cfg.n_blocks;
  a.p0 = static_cast<const double *>(dx);
  a.p = f->p_dev; a.p_trial = f->p_trial; a.dscale = f->dscale; a.apk = f->redbuf; a.gvec = f->redbuf + f->npk;
  a.v_out = f->yv + P; a.coln2 = f->diag_dev; a.st = f->lmd;
  a.prior_prec = f->cfg.has_prior ? f->prior_prec : nullptr;
  a.prior_mean = f->cfg.has_prior ? f->prior_mean : nullptr;
  a.prior_dense = f->cfg.prior_dense; a.scaler = f->opt.scaler; a.maxit = f->opt.maxit;
  a.watch = f->opt.solver == LSQAMD_SOLVER_QR ? 1 : 0;
  a.xtol = f->opt.xtol; a.gtol = f->opt.gtol; a.factor_up = f->opt.factor_up; a.factor_down = f->opt.factor_down;
  const long long bits = zc_off ? 0 : (long long)(intptr_t)dlm;
  std::memcpy(&a.hostptr_bits, &bits, sizeof(double));
  a.host = f->fit_block;
  a.pub = zc_off ? nullptr : static_cast<unsigned long long *>(dfit);
  a.seq = seq;
  // the covariance of the normal-equation route rides along (solver = qr factors the Jacobian itself: do_covariance_qr)
  a.cov = f->cov; a.ldc = f->ldm; a.want_cov = f->opt.solver == LSQAMD_SOLVER_QR ? 0 : 1; a.pad_ = 0;
  HIPCHK(f, lsqamd_jit::launch_fit(k, f->st, a));
  f->fit_rec.resize((size_t)nw);
  unsigned long long *rec = reinterpret_cast<unsigned long long *>(f->fit_rec.data());
  bool have = false;
  if (a.pub) {
    // The flag word is this launch's only when it carries this launch's sequence number; the block behind it counts only when
    // its checksum (seeded with the same number) fits the snapshot taken -- the kernel's stores arrive in no particular
    // order (jit.hip, end of lm_fit).  A short spin (most small fits end within 100 us), then yielding the core, then the
    // stream; a block that still does not verify once the stream has drained is replaced by the device copy.
    auto arrived = [&]() {
      const unsigned long long flag = hw[16];
      if ((flag >> 40) != seq) return false;
      std::atomic_thread_fence(std::memory_order_acquire);
      for (int i = 0; i < nw; ++i) rec[i] = hw[i];
      if (rec[16] != flag || rec[23] != lsqamd_jit::fit_checksum(rec, nw, seq)) { g_handoff[0]++; return false; }
      // the flag IS the decision: reason | info | nit; the record behind it must say the same
      const int reason = (int)(flag & 0xff), info = (int)(int8_t)((flag >> 8) & 0xff), nit = (int)((flag >> 16) & 0xffffff);
      if ((int)f->fit_rec[LMS_INFO] != info || (int)f->fit_rec[17] != nit) { g_handoff[0]++; return false; }
      f->fit_rec[16] = (double)reason;
      f->fit_rec[23] = 0.0;
      return true;
    };
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1; !(have = arrived()); ++spins) {
      cpu_relax();
      if ((spins & 63) == 0) {
        const auto dt = std::chrono::steady_clock::now() - t0;
        if (dt > std::chrono::microseconds(5000)) break;
        if (dt > std::chrono::microseconds(200)) sched_yield();
      }
    }
    if (!have) {
      HIPCHK(f, hipStreamSynchronize(f->st));
      have = arrived();
      if (!have) g_handoff[1]++;
    }
  }
  if (!have) {
    HIPCHK(f, hipStreamSynchronize(f->st));
    HIPCHK(f, hipMemcpyAsync(f->fit_rec.data(), f->fit_block, sizeof(double) * (size_t)nw, hipMemcpyDeviceToHost, f->st));
    HIPCHK(f, hipStreamSynchronize(f->st));
  }
  if (getenv("LSQAMD_VERIFY_HANDOFF")) {   // test knob: the block the host acted on vs the device's own copy once the stream has drained
    std::vector<double> dev((size_t)nw);
    HIPCHK(f, hipStreamSynchronize(f->st));
    HIPCHK(f, hipMemcpyAsync(dev.data(), f->fit_block, sizeof(double) * (size_t)nw, hipMemcpyDeviceToHost, f->st));
    HIPCHK(f, hipStreamSynchronize(f->st));
    int bad = 0;
    for (int i = 0; i < nw; ++i)
      if (i != 23 && std::memcmp(&dev[(size_t)i], &f->fit_rec[(size_t)i], sizeof(double)) != 0) {
        fprintf(stderr, "lsqamd HANDOFF MISMATCH word %d: host %a device %a\n", i, f->fit_rec[(size_t)i], dev[(size_t)i]);
        ++bad;
      }
    if (bad) { g_handoff[2] += bad; fprintf(stderr, "lsqamd HANDOFF: %d mismatches (P %lld N %lld)\n", bad, (long long)P, (long long)f->N); }
  }
  const double *R = f->fit_rec.data();
  if (R[16] != 1.0) {                 // irregular: the general path from the start
    HIPCHK(f, hipStreamSynchronize(f->st));
    return 0;
  }
  const double *m = R + 24;
  f->hx.assign(m, m + P);
  f->hg.assign(m + (P + 1), m + (P + 1) + P);
  f->hdiag.assign(m + 2 * (P + 1), m + 2 * (P + 1) + P);
  f->hcoln.resize(P); f->hv.resize(P); f->hdx.resize(P);
  for (int64_t j = 0; j < P; ++j) {
    const double c2 = m[3 * (P + 1) + j], v = m[4 * (P + 1) + j];
    f->hcoln[j] = std::sqrt(c2 > 0.0 ? c2 : 0.0);
    f->hv[j] = v;
    f->hdx[j] = -v;
  }
  for (int i = 0; i < LMS_COUNT; ++i) f->pin_lm[i] = R[i];
  f->nit = (int)R[17]; f->nfev = (int)R[18]; f->njev = (int)R[19]; f->ntrial = (int)R[20];
  f->chol_fail = f->qr_trials = 0;
  f->qr_steps_on = false;
  f->logdet = NAN;
  f->chi2 = R[LMS_CHI2]; f->mu = R[LMS_MU]; f->nu = (long)R[LMS_NU]; f->delta = R[LMS_DELTA];
  f->conv_info_dev = (int32_t)R[LMS_INFO];
  f->dev_lm = true;
  f->lm_zero_copy = bits != 0;
  f->lm_seq_expect = 0.0;
  f->mirrors_stale = false;
  f->J_stale = true;               // (no Jacobian was written: ensure_J() for whoever reads it)
  f->used_nrm = true;
  f->used_one_launch = true;
  if (getenv("LSQAMD_FIT_DIAG"))     // developer knob: where the kernel's cycles went
    fprintf(stderr, "lsqamd_jit_lm: %.0f shader cycles (normal equations %.0f, solves %.0f, trial residuals %.0f), %.1f us; nit %d trials %d\n",
            R[lsqamd_jit::fit_host_diag((int)P)], R[lsqamd_jit::fit_host_diag((int)P) + 1], R[lsqamd_jit::fit_host_diag((int)P) + 2],
            R[lsqamd_jit::fit_host_diag((int)P) + 3], R[lsqamd_jit::fit_host_diag((int)P) + 4] / 100.0, (int)f->nit, (int)f->ntrial);
  f->nrm_in_tail = 0;
  f->prior_deferred = false;
  f->r_fresh = false;
  f->have_cov = false;
  f->cov_host_valid = false;
  if (a.want_cov && R[21] == 1.0) {     // (else: do_covariance, the caller's next step)
    f->have_cov = true;
    f->cov_host_valid = true;
    f->cov_inaccurate = false;
    f->cov_dropped = 0;
    f->logdet = R[22];
  }
  f->have_dense_A = false;
  f->initialised = true;
  *iter = (int)f->nit;
  *info = f->conv_info_dev;
  return 1;
}

int lsqamd_run(lsqamd_fit *f, const double *p0, lsqamd_summary *out) try {
  if (!f || !p0) return LSQAMD_EINVAL;
  struct Pair {   // recycled events: every return path hands them back
    lsqamd_fit *f;
    hipEvent_t a, b;
    explicit Pair(lsqamd_fit *fit) : f(fit), a(take_event(fit)), b(take_event(fit)) {}
    ~Pair() { f->event_pool.push_back(a); f->event_pool.push_back(b); }
  } ev(f);
  (void)hipEventRecord(ev.a, f->st);
  f->used_one_launch = false;
  f->cov_host_valid = false;
  int rc = 0;
  int iter = 0, info = 0, status = -2;
  bool early = false;
  const int maxit = f->opt.maxit;
  if (f->opt.trs >= LSQAMD_TRS_TRF) {
    int st = 0;
    if (f->loss != LSQAMD_LOSS_LINEAR && f->opt.trs == LSQAMD_TRS_MINPACK_LM)
      FAIL(f, LSQAMD_EINVAL, "method='lm' supports only 'linear' loss function.");      // scipy's wording
    if (f->robust() && f->cfg.has_prior)
      FAIL(f, LSQAMD_EUNSUPPORTED, "a robust loss acts on residual rows: pass the prior as rows (lsqamd_set_param_rows), not through lsqamd_set_prior");
    if (f->robust() && (f->comm || f->reduce))
      FAIL(f, LSQAMD_EUNSUPPORTED, "robust losses are single-rank");
    rc = f->opt.trs == LSQAMD_TRS_TRF ? run_trf(f, p0, &st)
         : f->opt.trs == LSQAMD_TRS_DOGBOX ? run_dogbox(f, p0, &st) : run_minpack(f, p0, &st);
    if (rc) return rc;
    f->nit = f->nfev;                      // _scipy.py:161: nit = number of function evaluations
    info = LSQAMD_INFO_TRF + st;
    status = 0;
  } else if ((rc = run_one_launch(f, p0, &iter, &info)) != 0) {
    if (rc < 0) return rc;
    status = info ? 0 : (iter >= maxit ? LSQAMD_EMAXITER : -2);
  } else if ((rc = do_init(f, p0)) != 0) {
    return rc;
  } else if (maxit > 0) {  // gsl_multifit_nlinear_driver
    do {
      rc = iterate(f);
      if (rc < 0) return rc;
      f->nit++;
      if (rc == LSQAMD_ENOPROG && iter == 0) {
        info = LSQAMD_ENOPROG;
        status = LSQAMD_EMAXITER;
        early = true;
        break;
      }
      ++iter;
      info = convergence_test(f);
      status = info ? 0 : -2;
    } while (status == -2 && iter < maxit);
    if (!early && iter >= maxit && status != 0) status = LSQAMD_EMAXITER;
  } else {
    status = 0;
  }
  rc = (f->used_one_launch && f->have_cov) ? 0 : do_covariance(f);
  if (rc < 0 && rc != LSQAMD_ENOTPD) return rc;
  if (f->robust()) {
    // the covariance above is that of the loss-scaled Jacobian scipy returns (src/lsqfit/_scipy.py:165-169); chi2 and
    // log det(J^T J) -- and what the getters hand out from here on -- are those of the TRUE residuals and Jacobian
    // at the fit point (:160-161, src/lsqfit/__init__.py:667,:719): one more evaluation, without the loss
    const int32_t loss = f->loss;
    f->loss = LSQAMD_LOSS_LINEAR;
    int r2 = eval_normal_dev(f, f->p_dev, true);
    if (r2 == 0) r2 = logdet_normal(f);
    f->loss = loss;
    f->njev--;                      // (bookkeeping of the library, not an evaluation of the method)
    if (r2 < 0 && r2 != LSQAMD_ENOTPD) return r2;
    // (the evaluation above overwrote the loss-scaled normal matrix: the covariance in f->cov stays valid only if it was
    // made; a later lsqamd_get_cov must not recompute it from the UNSCALED matrix that is there now)
    f->have_cov = rc == 0;
    f->cov_unavailable = rc != 0;
  }
  float ms = 0.f;
  if (f->used_one_launch && f->have_cov && rc == 0) {
    // the whole run was the one kernel whose last word the host has already seen: nothing is left on the stream to wait for
    // (an event synchronisation is ~15 us of sleeping and waking); the kernel timed itself (100 MHz ticks)
    ms = (float)(f->fit_rec[(size_t)lsqamd_jit::fit_host_diag((int)f->P) + 4] * 1e-5);
    f->stage_off = 0;
  } else {
    (void)hipEventRecord(ev.b, f->st);
    if (hipEventSynchronize(ev.b) == hipSuccess) f->stage_off = 0;   // (the stream has drained: the staging arena is free again)
    (void)hipEventElapsedTime(&ms, ev.a, ev.b);
  }
  if (f->timing) resolve_timers(f);
  if (f->comm && f->xg > 1 && f->xg_signal) {     // did a wait of the grouped exchange give up? (bounded spin: never a hang)
    int32_t to = 0;
    HIPCHK(f, hipMemcpyAsync(&to, f->xg_ctr + 8, sizeof to, hipMemcpyDeviceToHost, f->st));
    HIPCHK(f, hipStreamSynchronize(f->st));
    if (to) FAIL(f, LSQAMD_EREDUCE, "grouped exchange: the wait for a group of J^T J tiles timed out (the sums of this fit are not to be trusted)");
  }
  fill_summary(f, out, status, info);
  if (out) {
    out->t_run_ms = ms;
    // LSQAMD_ENOTPD: J^T J singular at the end point, cov / logdet undefined; LSQAMD_EINACCURATE: delivered, degraded
    out->cov_status = rc == 0 && f->cov_inaccurate ? LSQAMD_EINACCURATE : (rc == 0 ? f->cov_dropped : rc);
  }
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_eval_residual(lsqamd_fit *f, const double *p, double *chi2) try {
  if (!f || !p || !chi2) return LSQAMD_EINVAL;
  int rc = ready(f);
  if (rc) return rc;
  HIPCHK(f, hipMemcpyAsync(f->p_trial, p, sizeof(double) * f->P, hipMemcpyHostToDevice, f->st));
  rc = eval_residual_dev(f, f->p_trial, chi2);
  if (f->timing) resolve_timers(f);
  return rc;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_eval_fcn(lsqamd_fit *f, const double *p, double *out, size_t cap) try {
  if (!f || !p || !out) return LSQAMD_EINVAL;
  int rc = ready(f);
  if (rc) return rc;
  const int64_t N = f->N;
  if (cap < (size_t)N) FAIL(f, LSQAMD_ECAPACITY, "eval_fcn: need %lld", (long long)N);
  if (N == 0) return 0;
  // residual kernel: w (f - y) for the 1x1 rows, f - y for the rows inside blocks
  HIPCHK(f, hipMemcpyAsync(f->p_trial, p, sizeof(double) * f->P, hipMemcpyHostToDevice, f->st));
  ModelArgs m = model_args(f, f->p_trial);
  HIPCHK(f, launch_residual_ex(f->st, m, f->r, f->r_raw));
  if (f->have_param_rows)     // rows that ARE parameters (lsqamd_set_param_rows): their "function value" is p_j
    HIPCHK(f, launch_param_rows(f->st, f->row_param, f->N, f->P, 1, f->p_trial, f->ymean, f->wdiag,
                                f->cfg.n_blocks > 0 ? f->in_block : nullptr, f->r, f->r_raw, 0));
  std::vector<double> r((size_t)N), rr((size_t)N, 0.0), y((size_t)N), w((size_t)N);
  std::vector<uint8_t> inb((size_t)N, 0);
  HIPCHK(f, hipMemcpyAsync(r.data(), f->r, sizeof(double) * N, hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipMemcpyAsync(y.data(), f->ymean, sizeof(double) * N, hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipMemcpyAsync(w.data(), f->wdiag, sizeof(double) * N, hipMemcpyDeviceToHost, f->st));
  if (f->cfg.n_blocks > 0) {
    HIPCHK(f, hipMemcpyAsync(rr.data(), f->r_raw, sizeof(double) * N, hipMemcpyDeviceToHost, f->st));
    HIPCHK(f, hipMemcpyAsync(inb.data(), f->in_block, (size_t)N, hipMemcpyDeviceToHost, f->st));
  }
  HIPCHK(f, hipStreamSynchronize(f->st));
  for (int64_t i = 0; i < N; ++i)
    out[i] = y[(size_t)i] + (inb[(size_t)i] ? rr[(size_t)i] : r[(size_t)i] / w[(size_t)i]);
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_eval_normal(lsqamd_fit *f, const double *p, double *chi2) try {
  if (!f || !p) return LSQAMD_EINVAL;
  int rc = ready(f);
  if (rc) return rc;
  const int64_t P = f->P;
  f->hx.assign(p, p + P);
  f->hg.assign(P, 0.0);
  f->hcoln.assign(P, 0.0);
  f->hv.assign(P, 0.0);
  f->hdx.assign(P, 0.0);
  if ((int64_t)f->hdiag.size() != P) f->hdiag.assign(P, 1.0);
  f->dev_lm = false;
  f->mirrors_stale = false;
  HIPCHK(f, hipMemcpyAsync(f->p_dev, p, sizeof(double) * P, hipMemcpyHostToDevice, f->st));
  rc = eval_normal_dev(f, f->p_dev);
  if (f->timing) resolve_timers(f);
  if (rc) return rc;
  // the Jacobian evaluation leaves the whitened residual in column P of J; mirror it into r
  HIPCHK(f, launch_copy_strided(f->st, f->J + P, f->ld, f->r, 1, f->N, 1));
  f->initialised = true;
  if (chi2) *chi2 = f->chi2;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_solve_damped(lsqamd_fit *f, double mu, const double *diag, double *v) try {
  if (!f || !diag || !v) return LSQAMD_EINVAL;
  if (!f->initialised) FAIL(f, LSQAMD_EINVAL, "solve_damped needs lsqamd_eval_normal / lsqamd_init first");
  const int rc = solve_damped_dev(f, mu, diag);
  if (f->timing) resolve_timers(f);
  if (rc) {
    if (rc == LSQAMD_ENOTPD) f->err = "damped normal matrix is not positive definite";
    return rc;
  }
  std::memcpy(v, f->hv.data(), sizeof(double) * f->P);
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_op_gemm_tn(void *stream, int64_t M, int64_t N, int64_t K, double alpha, const double *X,
                      int64_t ldx, const double *Y, int64_t ldy, double beta, double *C, int64_t ldc,
                      int32_t upper_only, int32_t x_upper_tri) try {
  GemmTN g;
  g.X = X; g.Y = Y; g.C = C;
  g.M = M; g.N = N; g.K = K;
  g.ldx = ldx; g.ldy = ldy; g.ldc = ldc;
  g.alpha = alpha; g.beta = beta;
  g.upper_only = upper_only;
  g.x_upper_tri = x_upper_tri;
  return launch_gemm_tn(reinterpret_cast<hipStream_t>(stream), g) == hipSuccess ? 0 : LSQAMD_EHIP;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)

size_t lsqamd_op_potrf_work_bytes(int64_t n) { return potrf_work_bytes(n); }

int lsqamd_op_potrf_upper(void *stream, double *A, int64_t n, int64_t lda, int64_t n_cols, double *work,
                          size_t work_bytes, int32_t *dev_info) try {
  if (work_bytes < potrf_work_bytes(n)) return LSQAMD_ENOMEM;
  return potrf_upper(reinterpret_cast<hipStream_t>(stream), A, n, lda, n_cols, work, dev_info) == hipSuccess
             ? 0 : LSQAMD_EHIP;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)

int64_t lsqamd_nf(const lsqamd_fit *f) try {
  if (!f) return 0;
  int64_t nf = f->N;
  for (size_t b = 0; b < f->h_size.size(); ++b) nf -= f->h_size[b] - f->h_modes[b];
  if (f->cfg.has_prior) nf += f->P;
  return nf;
} LSQAMD_ABI_CATCH((void)lsqamd::abi_exception(nullptr); return 0;)

int lsqamd_get_x(lsqamd_fit *f, double *out, size_t cap) try {
  if (!f || !out) return LSQAMD_EINVAL;
  if (cap < (size_t)f->P) FAIL(f, LSQAMD_ECAPACITY, "get_x: need %lld", (long long)f->P);
  if ((int64_t)f->hx.size() != f->P) FAIL(f, LSQAMD_EINVAL, "get_x: no fit has run");
  if (const int rcm = refresh_mirrors(f)) return rcm;
  std::memcpy(out, f->hx.data(), sizeof(double) * f->P);
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_get_grad(lsqamd_fit *f, double *out, size_t cap) try {
  if (!f || !out) return LSQAMD_EINVAL;
  if (cap < (size_t)f->P) FAIL(f, LSQAMD_ECAPACITY, "get_grad: need %lld", (long long)f->P);
  if ((int64_t)f->hg.size() != f->P) FAIL(f, LSQAMD_EINVAL, "get_grad: no fit has run");
  if (const int rcm = refresh_mirrors(f)) return rcm;
  std::memcpy(out, f->hg.data(), sizeof(double) * f->P);
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

// Data part of f / J in the reference's order (_utilities.pyx:85-93): all 1x1 rows first,
// then each block's kept modes.  Prior rows are appended by the host layer, which owns the
// prior's whitening (any W with W^T W = precision is equivalent: SURVEY.md App. B).
int lsqamd_get_f(lsqamd_fit *f, double *out, size_t cap) try {
  if (!f || !out) return LSQAMD_EINVAL;
  if (!f->initialised) FAIL(f, LSQAMD_EINVAL, "get_f: no fit has run");
  int64_t nfd = f->N;
  for (size_t b = 0; b < f->h_size.size(); ++b) nfd -= f->h_size[b] - f->h_modes[b];
  if (cap < (size_t)nfd) FAIL(f, LSQAMD_ECAPACITY, "get_f: need %lld", (long long)nfd);
  std::vector<double> r((size_t)(f->N > 0 ? f->N : 1));
  if (const int rcj = ensure_J(f)) return rcj;
  // the residual at the CURRENT point is column P of J
  HIPCHK(f, launch_copy_strided(f->st, f->J + f->P, f->ld, f->r, 1, f->N, 1));
  HIPCHK(f, hipMemcpyAsync(r.data(), f->r, sizeof(double) * f->N, hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipStreamSynchronize(f->st));
  std::vector<uint8_t> inb((size_t)(f->N > 0 ? f->N : 1), 0);
  for (size_t b = 0; b < f->h_size.size(); ++b)
    for (int64_t i = 0; i < f->h_size[b]; ++i) inb[(size_t)(f->h_row0[b] + i)] = 1;
  size_t o = 0;
  for (int64_t i = 0; i < f->N; ++i)
    if (!inb[(size_t)i]) out[o++] = r[(size_t)i];
  for (size_t b = 0; b < f->h_size.size(); ++b)
    for (int64_t m = 0; m < f->h_modes[b]; ++m) out[o++] = r[(size_t)(f->h_row0[b] + m)];
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_get_J(lsqamd_fit *f, double *out, size_t cap) try {
  if (!f || !out) return LSQAMD_EINVAL;
  if (!f->initialised) FAIL(f, LSQAMD_EINVAL, "get_J: no fit has run");
  const int64_t P = f->P;
  if (const int rcj = ensure_J(f)) return rcj;
  int64_t nfd = f->N;
  for (size_t b = 0; b < f->h_size.size(); ++b) nfd -= f->h_size[b] - f->h_modes[b];
  if (cap < (size_t)(nfd * P)) FAIL(f, LSQAMD_ECAPACITY, "get_J: need %lld", (long long)(nfd * P));
  std::vector<uint8_t> inb((size_t)(f->N > 0 ? f->N : 1), 0);
  for (size_t b = 0; b < f->h_size.size(); ++b)
    for (int64_t i = 0; i < f->h_size[b]; ++i) inb[(size_t)(f->h_row0[b] + i)] = 1;
  // copy row by row ranges with 2D copies (ld -> P)
  auto copy_rows = [&](int64_t src_row, int64_t nrows, size_t dst_row) -> hipError_t {
    return hipMemcpy2DAsync(out + dst_row * P, sizeof(double) * P, f->J + src_row * f->ld,
                            sizeof(double) * f->ld, sizeof(double) * P, (size_t)nrows,
                            hipMemcpyDeviceToHost, f->st);
  };
  size_t o = 0;
  int64_t i = 0;
  while (i < f->N) {
    if (inb[(size_t)i]) { ++i; continue; }
    int64_t j = i;
    while (j < f->N && !inb[(size_t)j]) ++j;
    HIPCHK(f, copy_rows(i, j - i, o));
    o += (size_t)(j - i);
    i = j;
  }
  for (size_t b = 0; b < f->h_size.size(); ++b) {
    if (f->h_modes[b] > 0) HIPCHK(f, copy_rows(f->h_row0[b], f->h_modes[b], o));
    o += (size_t)f->h_modes[b];
  }
  HIPCHK(f, hipStreamSynchronize(f->st));
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_get_jtj(lsqamd_fit *f, double *out, size_t cap) try {
  if (!f || !out) return LSQAMD_EINVAL;
  if (!f->initialised) FAIL(f, LSQAMD_EINVAL, "get_jtj: no fit has run");
  const int64_t P = f->P;
  if (cap < (size_t)(P * P)) FAIL(f, LSQAMD_ECAPACITY, "get_jtj: need %lld", (long long)(P * P));
  HIPCHK(f, launch_unpack_sym(f->st, f->redbuf, P, f->Wl, f->ldm));
  HIPCHK(f, hipMemcpy2DAsync(out, sizeof(double) * P, f->Wl, sizeof(double) * f->ldm, sizeof(double) * P,
                             (size_t)P, hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipStreamSynchronize(f->st));
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_get_cov(lsqamd_fit *f, double *out, size_t cap) try {
  if (!f || !out) return LSQAMD_EINVAL;
  if (!f->initialised) FAIL(f, LSQAMD_EINVAL, "get_cov: no fit has run");
  const int64_t P = f->P;
  if (cap < (size_t)(P * P)) FAIL(f, LSQAMD_ECAPACITY, "get_cov: need %lld", (long long)(P * P));
  if (!f->have_cov) {
    if (f->cov_unavailable) FAIL(f, LSQAMD_ENOTPD, "get_cov: the loss-scaled Jacobian of this robust fit had no covariance (summary.cov_status)");
    const int rc = do_covariance(f);
    if (rc < 0 && rc != LSQAMD_ENOTPD) return rc;
  }
  if (f->cov_host_valid) {       // the one-launch fit kernel mirrored it into pinned memory: no copy, no synchronisation
    std::memcpy(out, f->fit_rec.data() + lsqamd_jit::fit_host_cov((int)P), sizeof(double) * (size_t)(P * P));
    return 0;
  }
  HIPCHK(f, hipMemcpy2DAsync(out, sizeof(double) * P, f->cov, sizeof(double) * f->ldm, sizeof(double) * P,
                             (size_t)P, hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipStreamSynchronize(f->st));
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

namespace {
struct DpdyPlan {
  int64_t ldv, ldn;
  double *gt, *V, *Jt, *T, *out, *wn;
  size_t bytes;
};
DpdyPlan dpdy_plan(const lsqamd_fit *f, int64_t m, void *base) {
  DpdyPlan d;
  d.ldv = rup(m, 16);
  d.ldn = rup(f->N > 0 ? f->N : 1, 16);
  Carver cv(base, 0, base == nullptr);
  d.gt = cv.take<double>(f->P * d.ldv);
  d.V = cv.take<double>(f->P * d.ldv);
  d.Jt = cv.take<double>(f->P * d.ldn);
  d.T = cv.take<double>(f->N * d.ldv);
  d.out = cv.take<double>((f->N + f->P) * d.ldv);
  d.wn = cv.take<double>(f->cfg.sum_block_sq);
  d.bytes = cv.off;
  return d;
}
}  // namespace

size_t lsqamd_dpdy_work_bytes(const lsqamd_fit *f, int64_t m) try {
  if (!f || m < 1) return 0;
  return dpdy_plan(f, m, nullptr).bytes + 256;
} LSQAMD_ABI_CATCH((void)lsqamd::abi_exception(nullptr); return 0;)

int lsqamd_dpdy(lsqamd_fit *f, const double *gt, int64_t m, void *dev_scratch, size_t scratch_bytes,
                double *out_t, size_t cap) try {
  if (!f || !out_t || !dev_scratch) return LSQAMD_EINVAL;
  if (!f->initialised) FAIL(f, LSQAMD_EINVAL, "dpdy: no fit has run");
  const int64_t P = f->P, N = f->N;
  if (m < 1 || (!gt && m != P)) FAIL(f, LSQAMD_EINVAL, "dpdy: gt == NULL needs m == P");
  if (const int rcj = ensure_J(f)) return rcj;
  const int64_t nrows = N + (f->cfg.has_prior ? P : 0);
  if (cap < (size_t)(nrows * m)) FAIL(f, LSQAMD_ECAPACITY, "dpdy: need %lld", (long long)(nrows * m));
  char *base = (char *)dev_scratch;
  const size_t pad = (size_t)((-(intptr_t)base) & 255);
  DpdyPlan d = dpdy_plan(f, m, base + pad);
  if (scratch_bytes < d.bytes + pad) FAIL(f, LSQAMD_ENOMEM, "dpdy: scratch needs %zu bytes", d.bytes + 256);
  if (!f->have_cov) {
    const int rc = do_covariance(f);
    if (rc) return rc;
  }
  Scope sc(f, LSQAMD_T_COVAR);
  // V = cov . gt  (P x m); identity: V is cov itself
  const double *V = f->cov;
  int64_t ldv = f->ldm;
  if (gt) {
    HIPCHK(f, hipMemcpy2DAsync(d.gt, sizeof(double) * d.ldv, gt, sizeof(double) * m, sizeof(double) * m,
                               (size_t)P, hipMemcpyHostToDevice, f->st));
    GemmTN g;
    g.X = f->cov; g.ldx = f->ldm; g.Y = d.gt; g.ldy = d.ldv; g.C = d.V; g.ldc = d.ldv;
    g.M = P; g.N = m; g.K = P;
    HIPCHK(f, launch_gemm_tn(f->st, g));
    V = d.V;
    ldv = d.ldv;
  }
  const bool blocks = f->cfg.n_blocks > 0;
  if (N > 0) {
    // Jt[b][n] = w_n J[n][b]: 1x1 rows get their second factor 1/sigma here, block rows wait
    // for W^T below
    HIPCHK(f, launch_transpose_scale(f->st, f->J, f->ld, d.Jt, d.ldn, N, P, f->wdiag,
                                     blocks ? f->in_block : nullptr, 1, 0, 0));
    GemmTN g;  // T[n][o] = sum_b Jt[b][n] V[b][o]
    g.X = d.Jt; g.ldx = d.ldn; g.Y = V; g.ldy = ldv;
    g.C = blocks ? d.T : d.out; g.ldc = d.ldv;
    g.M = N; g.N = m; g.K = P;
    HIPCHK(f, launch_gemm_tn(f->st, g));
    if (blocks) {
      HIPCHK(f, launch_rows_scale_copy(f->st, d.T, d.ldv, d.out, d.ldv, N, m, nullptr, f->in_block));
      // out_b[i][o] = sum_j W_b[j][i] T_b[j][o]: the X operand is W_b itself = (stored Wt_b)^T
      for (size_t b = 0; b < f->h_size.size(); ++b) {
        const int64_t B = f->h_size[b];
        if (f->uniform_blocks && b > 0) break;
        const int64_t nb = f->uniform_blocks ? (int64_t)f->h_size.size() : 1;
        HIPCHK(f, launch_transpose_scale(f->st, f->wt + f->h_woff[b], B, d.wn + f->h_woff[b], B, B, B,
                                         nullptr, nullptr, nb, B * B, B * B));
        GemmTN w;
        w.X = d.wn + f->h_woff[b]; w.ldx = B; w.sx = B * B;
        w.Y = d.T + f->h_row0[b] * d.ldv; w.ldy = d.ldv; w.sy = B * d.ldv;
        w.C = d.out + f->h_row0[b] * d.ldv; w.ldc = d.ldv; w.sc = B * d.ldv;
        w.M = B; w.N = m; w.K = B;
        w.batch = (int32_t)nb;
        HIPCHK(f, launch_gemm_tn(f->st, w));
      }
    }
  }
  if (f->cfg.has_prior) {  // prior entries: Lambda . V
    double *op = d.out + N * d.ldv;
    if (f->cfg.prior_dense) {
      GemmTN g;
      g.X = f->prior_prec; g.ldx = P; g.Y = V; g.ldy = ldv; g.C = op; g.ldc = d.ldv;
      g.M = P; g.N = m; g.K = P;
      HIPCHK(f, launch_gemm_tn(f->st, g));
    } else {
      HIPCHK(f, launch_rows_scale_copy(f->st, V, ldv, op, d.ldv, P, m, f->prior_prec, nullptr));
    }
  }
  HIPCHK(f, hipMemcpy2DAsync(out_t, sizeof(double) * m, d.out, sizeof(double) * d.ldv, sizeof(double) * m,
                             (size_t)nrows, hipMemcpyDeviceToHost, f->st));
  HIPCHK(f, hipStreamSynchronize(f->st));
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

namespace {
struct Chi2Plan {
  int64_t ldt;
  double *p, *r, *r_raw, *Dt, *T, *out;
  size_t bytes;
};
Chi2Plan chi2_plan(const lsqamd_fit *f, int64_t mc, void *base) {
  Chi2Plan d;
  d.ldt = rup(mc, 16);
  Carver cv(base, 0, base == nullptr);
  d.p = cv.take<double>(mc * f->P);
  d.r = cv.take<double>(mc * f->N);
  d.r_raw = cv.take<double>(f->cfg.n_blocks > 0 ? mc * f->N : 1);
  const bool dense = f->cfg.has_prior && f->cfg.prior_dense;
  d.Dt = cv.take<double>(dense ? f->P * d.ldt : 1);
  d.T = cv.take<double>(dense ? f->P * d.ldt : 1);
  d.out = cv.take<double>(mc);
  d.bytes = cv.off;
  return d;
}
}  // namespace

size_t lsqamd_chi2_points_work_bytes(const lsqamd_fit *f, int64_t m) try {
  if (!f || m < 1) return 0;
  return chi2_plan(f, m, nullptr).bytes + 256;
} LSQAMD_ABI_CATCH((void)lsqamd::abi_exception(nullptr); return 0;)

int lsqamd_chi2_points(lsqamd_fit *f, const double *p, int64_t m, void *dev_scratch, size_t scratch_bytes,
                       double *chi2_out) try {
  if (!f || !p || !chi2_out || !dev_scratch || m < 0) return LSQAMD_EINVAL;
  int rc = ready(f);
  if (rc) return rc;
  if (m == 0) return 0;
  char *base = (char *)dev_scratch;
  const size_t pad = (size_t)((-(intptr_t)base) & 255);
  if (scratch_bytes <= pad) FAIL(f, LSQAMD_ENOMEM, "chi2_points: scratch too small");
  // largest chunk that fits the scratch
  int64_t mc = m;
  while (mc > 1 && chi2_plan(f, mc, nullptr).bytes + pad > scratch_bytes) mc = (mc + 1) / 2;
  if (chi2_plan(f, mc, nullptr).bytes + pad > scratch_bytes)
    FAIL(f, LSQAMD_ENOMEM, "chi2_points: scratch needs at least %zu bytes", chi2_plan(f, 1, nullptr).bytes + 256);
  if (mc > 65535) mc = 65535;  // grid.y of the model kernels
  const int64_t N = f->N, P = f->P;
  Scope sc(f, LSQAMD_T_RESIDUAL);
  for (int64_t m0 = 0; m0 < m; m0 += mc) {
    const int64_t mm = m - m0 < mc ? m - m0 : mc;
    Chi2Plan d = chi2_plan(f, mc, base + pad);
    HIPCHK(f, hipMemcpyAsync(d.p, p + m0 * P, sizeof(double) * mm * P, hipMemcpyHostToDevice, f->st));
    if (N > 0) {
      ModelArgs ma = model_args(f, d.p);
      ma.n_batch = (int32_t)mm; ma.p_stride = P; ma.out_stride = N;
      HIPCHK(f, launch_residual_ex(f->st, ma, d.r, d.r_raw));
      if (f->have_param_rows)   // prior entries whitened together with the data: rows whose "model" is a parameter
        HIPCHK(f, launch_param_rows(f->st, f->row_param, N, P, 1, d.p, f->ymean, f->wdiag,
                                    f->cfg.n_blocks > 0 ? f->in_block : nullptr, d.r, d.r_raw, 0, (int32_t)mm, P, N));
      if (f->cfg.n_blocks > 0)
        HIPCHK(f, launch_block_whiten_vec(f->st, f->wt, f->blk_row0, f->blk_size, f->blk_woff,
                                          f->cfg.n_blocks, f->cfg.max_block, d.r_raw, d.r, (int32_t)mm, N,
                                          nullptr));
      HIPCHK(f, launch_rows_sumsq(f->st, d.r, N, N, mm, d.out, 0));
    } else {
      HIPCHK(f, hipMemsetAsync(d.out, 0, sizeof(double) * mm, f->st));
    }
    if (f->cfg.has_prior && f->adds_prior)
      HIPCHK(f, launch_prior_chi2_points(f->st, P, f->prior_prec, f->cfg.prior_dense, f->prior_mean, d.p, mm,
                                         d.Dt, d.T, d.ldt, d.out));
    rc = do_reduce(f, d.out, mm);
    if (rc) return rc;
    HIPCHK(f, hipMemcpyAsync(chi2_out + m0, d.out, sizeof(double) * mm, hipMemcpyDeviceToHost, f->st));
    HIPCHK(f, hipStreamSynchronize(f->st));
    f->nfev += (int32_t)mm;
  }
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_handoff_stats(int64_t *out3) try {
  if (!out3) return LSQAMD_EINVAL;
  for (int i = 0; i < 3; ++i) out3[i] = g_handoff[i].load();
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)

void lsqamd_debug_set_potf2_stamps(void *dev_ptr) { lsqamd::g_potf2_dbg = (long long *)dev_ptr; }

// self-tests of the boundary's own machinery, runnable without a GPU (tests/test_abi.py)
int lsqamd_debug_throw(lsqamd_fit *f, int32_t kind) try {
  if (kind == 1) throw std::bad_alloc();
  if (kind == 2) throw std::runtime_error("lsqamd_debug_throw");
  if (kind == 3) throw 42;
  if (kind == 4) { std::vector<double> v; v.reserve(v.max_size()); }   // a real allocation failure (std::length_error / bad_alloc)
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

// The runtime behaviour capture_reset (common.h) exists for, reproduced deterministically: a ThreadLocal capture on `stream`,
// ONE legacy-stream hipMemcpy from another thread while it is open, then the recovery.  report[0] = what the intruder's
// hipMemcpy returned, [1] = what hipStreamEndCapture returned, [2] = the stream's capture status after EndCapture
// (2 = still "invalidated": the runtime defect), [3] = the status after capture_reset (0 = usable), [4] = result of an
// eager copy + synchronise on the stream afterwards (0 = success), [5] = the value that copy delivered (42).
int lsqamd_debug_capture_selftest(void *stream, int32_t *report) try {
  if (!report) return LSQAMD_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!st) return LSQAMD_EINVAL;
  for (int i = 0; i < 6; ++i) report[i] = -1;
  int32_t *dev = nullptr, *dev2 = nullptr;
  if (hipMalloc(&dev, 64) != hipSuccess || hipMalloc(&dev2, 64) != hipSuccess) return LSQAMD_EHIP;
  const int32_t v42 = 42;
  (void)hipMemcpyAsync(dev, &v42, sizeof v42, hipMemcpyHostToDevice, st);
  (void)hipStreamSynchronize(st);
  if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return LSQAMD_EHIP; }
  (void)hipMemcpyAsync(dev2, dev, 4, hipMemcpyDeviceToDevice, st);
  {
    std::atomic<int> rc{-1};
    std::thread other([&] {
      int32_t h = 0;
      rc.store((int)hipMemcpy(&h, dev, 4, hipMemcpyDeviceToHost));   // legacy default stream, from ANOTHER thread
      (void)hipGetLastError();
    });
    other.join();
    report[0] = rc.load();
  }
  (void)hipMemcpyAsync(dev2, dev, 4, hipMemcpyDeviceToDevice, st);
  (void)hipGetLastError();
  hipGraph_t g = nullptr;
  report[1] = (int32_t)hipStreamEndCapture(st, &g);
  (void)hipGetLastError();
  if (g) (void)hipGraphDestroy(g);
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cs);
  report[2] = (int32_t)cs;
  lsqamd::capture_reset(st);
  (void)hipStreamIsCapturing(st, &cs);
  report[3] = (int32_t)cs;
  int32_t back = 0;
  hipError_t e = hipMemcpyAsync(&back, dev, 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  report[4] = (int32_t)e;
  report[5] = back;
  (void)hipGetLastError();
  (void)hipFree(dev);
  (void)hipFree(dev2);
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)

// the work lists of the J^T J launch (host arithmetic only): G == 0: the plain list; G >= 1: the list of the grouped exchange
// in group-major order inside every XCD run (rows[G + 1] = tile-row bounds, count[G] = entries per group).  -> entries
int64_t lsqamd_debug_syrk_work(int64_t P, int32_t splits, int32_t G, const int32_t *rows, int32_t *out4, int32_t *count) try {
  if (P < 1 || splits < 1 || !out4) return -1;
  if (G <= 0) {
    syrk_work_fill(P, splits, out4);
    return syrk_work_count(P, splits);
  }
  if (!rows || !count || G > 8) return -1;
  syrk_work_fill_grouped(P, splits, G, rows, out4, count);
  return syrk_work_count(P, splits);
} LSQAMD_ABI_CATCH((void)lsqamd::abi_exception(nullptr); return 0;)

int lsqamd_debug_per_device_once(int32_t dev, int32_t reset) try {
  // how often has the "set the kernel attributes" action of a PerDeviceOnce run for device `dev`?  (the bookkeeping that
  // guards every kernel-attribute call of the library, on an instance of its own: no HIP call)
  static lsqamd::PerDeviceOnce once;
  static std::atomic<int> runs[lsqamd::PerDeviceOnce::kMaxDev + 1];
  if (reset) {
    once.done.store(0);
    for (auto &r : runs) r.store(0);
    return 0;
  }
  const int slot = (dev >= 0 && dev < lsqamd::PerDeviceOnce::kMaxDev) ? dev : lsqamd::PerDeviceOnce::kMaxDev;
  const hipError_t e = once.run_for(dev, [slot] { runs[slot].fetch_add(1); return hipSuccess; });
  if (e != hipSuccess) return LSQAMD_EHIP;
  return runs[slot].load();
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)

// introspection for tests: bit0 uniform-block batched whitening, bits 8.. split-K factor
int64_t lsqamd_debug_flags(const lsqamd_fit *f) try {
  if (!f) return -1;
  return (int64_t)(f->uniform_blocks ? 1 : 0) | (int64_t)(f->used_synth ? 2 : 0) | (int64_t)(f->graph_launches > 0 ? 4 : 0) |
         (int64_t)(f->used_nrm ? 16 : 0) | (int64_t)(f->used_one_launch ? 32 : 0) |
         (int64_t)(f->jit || (!f->progs.empty() && f->progs_compiled == (int)f->progs.size()) ? 8 : 0) |
         ((int64_t)f->splits << 8) |
         ((int64_t)f->h_size.size() << 32);
} LSQAMD_ABI_CATCH((void)lsqamd::abi_exception(nullptr); return 0;)

int lsqamd_timing_enable(lsqamd_fit *f, int32_t on) try {
  if (!f) return LSQAMD_EINVAL;
  f->timing = on != 0;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_timing_get(lsqamd_fit *f, int32_t which, double *total_ms, int64_t *count) try {
  if (!f || which < 0 || which >= LSQAMD_T_COUNT) return LSQAMD_EINVAL;
  resolve_timers(f);
  if (total_ms) *total_ms = f->timers[which].total_ms;
  if (count) *count = f->timers[which].count;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_timing_reset(lsqamd_fit *f) try {
  if (!f) return LSQAMD_EINVAL;
  resolve_timers(f);
  for (auto &t : f->timers) { t.total_ms = 0.0; t.count = 0; }
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

}  // extern "C"
