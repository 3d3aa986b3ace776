// RCCL collective of the row-sharded fit, inside the library.
//
// chi2 is a sum over covariance blocks, so a fit whose data rows are spread over the GPUs of one
// node needs ONE exchange per Jacobian evaluation -- the sum of the packed [J^T J | J^T f | chi2]
// buffer (69 MB at P = 4096) -- and one scalar per trial step.  The reference has no distributed
// path; there is nothing to mirror.  Design:
//   * ONE persistent communicator per (process, device, id), shared by the handles that name the same id
//     (lsqamd_comm_init: ncclCommInitRank -- timed, lsqamd_comm_stats -- the first time an id made by rank 0's
//     lsqamd_comm_unique_id is seen; later handles with that id just take a reference: a fit per problem no
//     longer pays a communicator set-up per problem.  The id travels to the other ranks by ANY host-side
//     channel: torch.distributed in lsqfit_amd/dist.py, MPI / a file / a socket for another host.  Handles that
//     share a communicator must not run collectives at the same time: they are consecutive problems of one job);
//   * the sum is enqueued on the handle's stream as reduce-scatter + all-gather over 256-byte
//     aligned slices (every xGMI link carries 1/n of the buffer in each phase; every rank ends
//     up with the SAME bytes, which is what lets all ranks take identical LM decisions), the few
//     leftover elements and short vectors as one all-reduce -- grouped with the reduce-scatter
//     (ncclGroupStart / End: independent regions, one launch; the all-gather depends on the reduce-scatter
//     and stays a launch of its own);
//   * no stream synchronisation, no host callback: the kernels queued behind the collective on
//     the same stream see the sums.
// RCCL is bound at run time (dlopen): inside a PyTorch process the librccl.so torch already
// loaded is reused (one RCCL per process), elsewhere LSQAMD_RCCL_PATH or the loader's search
// path decide; a build without RCCL on the box still loads and every other entry point works.
#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#include "fit_state.h"

namespace {

// the slice of the NCCL/RCCL C API this file uses (stable since NCCL 2.0)
struct UniqueId { char internal[LSQAMD_COMM_ID_BYTES]; };
typedef void *Comm;
enum { RESULT_SUCCESS = 0 };
enum { DT_FLOAT64 = 8 };   // ncclDouble / ncclFloat64
enum { OP_SUM = 0 };       // ncclSum

struct Api {
  void *lib = nullptr;
  int (*GetUniqueId)(UniqueId *) = nullptr;
  int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  int (*GroupStart)() = nullptr;      // optional
  int (*GroupEnd)() = nullptr;
  std::string why;
  bool ok = false;
};

Api &api() {
  static Api a = [] {
    Api t;
    const char *names[] = {"librccl.so", "librccl.so.1"};
    // LSQAMD_RCCL_PATH is an explicit choice and beats the copy already in the process (a host with
    // its own RCCL build; the multi-rank tests' stand-in, tests/fake_rccl.cpp)
    const char *path = getenv("LSQAMD_RCCL_PATH");
    std::string tried;
    if (path && *path) {
      t.lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
      if (!t.lib) {   // a named library that does not load is an error, not a reason to pick another one
        const char *e = dlerror();
        t.why = std::string("LSQAMD_RCCL_PATH=") + path + " could not be loaded: " + (e ? e : "unknown error");
        return t;
      }
    }
    for (const char *n : names)   // the copy already in the process (PyTorch's), if any
      if (!t.lib) t.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char *n : names)
      if (!t.lib) {
        t.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!t.lib) {
          const char *e = dlerror();   // ONE call: dlerror() clears the message it returns
          tried += std::string(tried.empty() ? "" : "; ") + (e ? e : "not found");
        }
      }
    if (!t.lib) {
      t.why = "librccl.so could not be loaded: " + tried;
      return t;
    }
    struct { const char *name; void **slot; } syms[] = {
        {"ncclGetUniqueId", (void **)&t.GetUniqueId},   {"ncclCommInitRank", (void **)&t.CommInitRank},
        {"ncclCommDestroy", (void **)&t.CommDestroy},   {"ncclAllReduce", (void **)&t.AllReduce},
        {"ncclReduceScatter", (void **)&t.ReduceScatter}, {"ncclAllGather", (void **)&t.AllGather},
        {"ncclGetErrorString", (void **)&t.GetErrorString}};
    for (auto &s : syms) {
      *s.slot = dlsym(t.lib, s.name);
      if (!*s.slot) {
        t.why = std::string("librccl.so lacks ") + s.name;
        return t;
      }
    }
    t.GroupStart = (int (*)())dlsym(t.lib, "ncclGroupStart");
    t.GroupEnd = (int (*)())dlsym(t.lib, "ncclGroupEnd");
    if (!t.GroupStart || !t.GroupEnd) t.GroupStart = t.GroupEnd = nullptr;
    t.ok = true;
    return t;
  }();
  return a;
}

// communicators of this process: key = device + id bytes
struct Shared {
  Comm comm = nullptr;
  int refs = 0, rank = 0, nranks = 1;
  double init_ms = 0.0;
  bool initialising = false;   // a thread is inside ncclCommInitRank for this key (g_reg_mu NOT held meanwhile)
};
std::mutex g_reg_mu;
std::condition_variable g_reg_cv;   // signalled when an entry leaves the `initialising` state
std::map<std::string, Shared> &registry() {
  static std::map<std::string, Shared> r;
  return r;
}

// LSQAMD_COMM_ALGO=allreduce: one ncclAllReduce for everything (A/B switch, developer knob)
bool use_rsag() {   // read per call (a getenv next to a 69 MB collective is free): tests flip it
  const char *e = getenv("LSQAMD_COMM_ALGO");
  return !(e && e[0] == 'a');
}

}  // namespace

#define NCCLCHK(fit, expr)                                                                       \
  do {                                                                                           \
    const int _r = (expr);                                                                       \
    if (_r != RESULT_SUCCESS) FAIL(fit, LSQAMD_EREDUCE, "%s: %s", #expr, api().GetErrorString(_r)); \
  } while (0)

namespace lsqamd_host {

// sums enqueued on `st` (the handle's stream, or its exchange stream for the grouped exchange of api.hip eval_normal_dev)
int comm_all_reduce_on(lsqamd_fit *f, hipStream_t st, double *buf, int64_t count) {
  Api &a = api();
  Comm c = (Comm)f->comm;
  const int64_t n = f->comm_nranks;
  // slices of whole 256-byte lines; below 64 KiB per rank the latency-optimised all-reduce wins
  const int64_t slice = (count / n) / 32 * 32;
  if (!use_rsag() || slice < 8192) {
    NCCLCHK(f, a.AllReduce(buf, buf, (size_t)count, DT_FLOAT64, OP_SUM, c, st));
    return 0;
  }
  double *mine = buf + (int64_t)f->comm_rank * slice;
  const int64_t done = slice * n;
  const bool group = a.GroupStart && done < count;
  if (group) NCCLCHK(f, a.GroupStart());
  // an error between GroupStart and GroupEnd must still close the group (an open group swallows every later call of the
  // thread): remember the first failure, end the group, then report
  int r1 = a.ReduceScatter(buf, mine, (size_t)slice, DT_FLOAT64, OP_SUM, c, st);
  const char *what = "ncclReduceScatter";
  if (r1 == RESULT_SUCCESS && done < count) {
    r1 = a.AllReduce(buf + done, buf + done, (size_t)(count - done), DT_FLOAT64, OP_SUM, c, st);
    what = "ncclAllReduce (tail)";
  }
  if (group) {
    const int r2 = a.GroupEnd();
    if (r1 == RESULT_SUCCESS && r2 != RESULT_SUCCESS) { r1 = r2; what = "ncclGroupEnd"; }
  }
  if (r1 != RESULT_SUCCESS) FAIL(f, LSQAMD_EREDUCE, "%s: %s", what, a.GetErrorString(r1));
  NCCLCHK(f, a.AllGather(mine, buf, (size_t)slice, DT_FLOAT64, c, st));
  return 0;
}

int comm_all_reduce(lsqamd_fit *f, double *buf, int64_t count) { return comm_all_reduce_on(f, f->st, buf, count); }

void comm_release(lsqamd_fit *f) {
  if (f->comm) {     // the communicator itself stays with the process (lsqamd_comm_shutdown ends it): the next handle reuses it
    std::lock_guard<std::mutex> lk(g_reg_mu);
    auto it = registry().find(f->comm_key);
    if (it != registry().end() && it->second.refs > 0) it->second.refs--;
    f->comm = nullptr;
  }
  f->comm_key.clear();
  f->comm_rank = 0;
  f->comm_nranks = 1;
}

}  // namespace lsqamd_host

extern "C" {

int lsqamd_comm_unique_id(void *id_out, size_t cap) try {
  if (!id_out || cap < LSQAMD_COMM_ID_BYTES) return LSQAMD_EINVAL;
  Api &a = api();
  if (!a.ok) return LSQAMD_EUNSUPPORTED;
  UniqueId id;
  if (a.GetUniqueId(&id) != RESULT_SUCCESS) return LSQAMD_EREDUCE;
  std::memcpy(id_out, id.internal, LSQAMD_COMM_ID_BYTES);
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)

int lsqamd_comm_init(lsqamd_fit *f, const void *id, size_t id_bytes, int32_t rank, int32_t nranks) try {
  if (!f) return LSQAMD_EINVAL;
  if (!id || id_bytes != LSQAMD_COMM_ID_BYTES || nranks < 1 || rank < 0 || rank >= nranks)
    FAIL(f, LSQAMD_EINVAL, "comm_init: need the %d-byte id of lsqamd_comm_unique_id and 0 <= rank < nranks", LSQAMD_COMM_ID_BYTES);
  Api &a = api();
  if (!a.ok) FAIL(f, LSQAMD_EUNSUPPORTED, "comm_init: %s", a.why.c_str());
  lsqamd_host::comm_release(f);
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::string key = std::to_string(dev) + ":";
  key.append(static_cast<const char *>(id), LSQAMD_COMM_ID_BYTES);
  // ncclCommInitRank is a COLLECTIVE (it returns when every rank has joined): it must not run under the registry's lock.
  // One process may drive several GPUs with a thread per rank (the key holds the device): thread 0 inside CommInitRank
  // waiting for rank 1 while thread 1 waits for the lock would never finish.  So: claim the key under the lock
  // (`initialising`), release the lock around the call, publish or withdraw the entry afterwards; a second handle naming
  // a key that is being initialised waits on the condition variable for the outcome.
  std::unique_lock<std::mutex> lk(g_reg_mu);
  for (;;) {
    auto it = registry().find(key);
    if (it == registry().end()) break;
    if (it->second.initialising) {
      g_reg_cv.wait(lk);
      continue;                      // (the entry may be gone: the initialisation failed)
    }
    if (it->second.rank != rank || it->second.nranks != nranks)
      FAIL(f, LSQAMD_EINVAL, "comm_init: this id already names a communicator with rank %d of %d", it->second.rank, it->second.nranks);
    it->second.refs++;
    f->comm = it->second.comm;
    f->comm_key = key;
    f->comm_rank = rank;
    f->comm_nranks = nranks;
    return 0;
  }
  {
    Shared claim;
    claim.initialising = true; claim.rank = rank; claim.nranks = nranks;
    registry()[key] = claim;
  }
  lk.unlock();
  UniqueId u;
  std::memcpy(u.internal, id, LSQAMD_COMM_ID_BYTES);
  Comm c = nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  const int ir = a.CommInitRank(&c, nranks, u, rank);
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  lk.lock();
  if (ir != RESULT_SUCCESS) {
    registry().erase(key);
    g_reg_cv.notify_all();
    FAIL(f, LSQAMD_EREDUCE, "ncclCommInitRank: %s", a.GetErrorString(ir));
  }
  {
    Shared &sh = registry()[key];
    sh.comm = c; sh.refs = 1; sh.rank = rank; sh.nranks = nranks; sh.init_ms = ms; sh.initialising = false;
  }
  g_reg_cv.notify_all();
  f->comm = c;
  f->comm_key = key;
  f->comm_rank = rank;
  f->comm_nranks = nranks;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_comm_stats(const lsqamd_fit *f, double *init_ms, int32_t *handles) try {
  if (!f) return LSQAMD_EINVAL;
  std::lock_guard<std::mutex> lk(g_reg_mu);
  auto it = registry().find(f->comm_key);
  const bool have = f->comm && it != registry().end() && !it->second.initialising;
  if (init_ms) *init_ms = have ? it->second.init_ms : 0.0;
  if (handles) *handles = have ? it->second.refs : 0;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)

int lsqamd_comm_shutdown(void) try {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  int busy = 0;
  for (auto it = registry().begin(); it != registry().end();) {
    if (it->second.refs > 0 || it->second.initialising) {
      ++busy;
      ++it;
      continue;
    }
    (void)api().CommDestroy(it->second.comm);
    it = registry().erase(it);
  }
  return busy;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)

int lsqamd_comm_destroy(lsqamd_fit *f) try {
  if (!f) return LSQAMD_EINVAL;
  (void)hipStreamSynchronize(f->st);
  lsqamd_host::comm_release(f);
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(f);)

int lsqamd_comm_info(const lsqamd_fit *f, int32_t *rank, int32_t *nranks) try {
  if (!f) return LSQAMD_EINVAL;
  if (rank) *rank = f->comm ? f->comm_rank : -1;
  if (nranks) *nranks = f->comm ? f->comm_nranks : 0;
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)

}  // extern "C"
