// TEST-ONLY stand-in for the seven RCCL symbols lsqfit_amd/csrc/comm.hip binds (ncclGetUniqueId,
// ncclCommInitRank, ncclCommDestroy, ncclAllReduce, ncclReduceScatter, ncclAllGather,
// ncclGetErrorString).  RCCL refuses two ranks on one device, and the GPU boxes of the pool have one
// GPU; this library lets 2 and 3 ranks that SHARE the GPU drive comm.hip's real slice / offset code
// (reduce-scatter + all-gather over 256-byte slices, tail all-reduce, the count-1 trial scalar).
// Built by tests/test_gpu_comm_multi.py, selected with LSQAMD_RCCL_PATH.  Never shipped, never loaded
// by the product unless that variable names it.
//
// Semantics kept from the real thing: buffers are DEVICE pointers, the call is ordered on the given
// stream (here: the stream is drained, data staged through POSIX shared memory, summed in RANK ORDER
// on the host by every rank -- so all ranks receive identical bytes, as RCCL's ring/tree guarantees
// for a given topology -- and copied back before the call returns), in-place forms included
// (reduce-scatter with recv = send + rank*count, all-gather with send = recv + rank*count).
// Every wait is bounded (LSQAMD_FAKE_RCCL_TIMEOUT_S, default 60 s): a missing rank is an error code,
// not a hang.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

enum { OK = 0, UNHANDLED_HIP = 1, SYSTEM_ERROR = 2, INTERNAL_ERROR = 3, INVALID_ARGUMENT = 4 };
constexpr int ID_BYTES = 128;
constexpr size_t SLOT_BYTES = size_t(160) << 20;   // per rank staging (sparse until touched); 69 MB at P = 4096 fits
constexpr int MAX_RANKS = 8;

struct Header {
  std::atomic<int32_t> joined;
  std::atomic<int32_t> arrive;      // barrier: arrivals of the current generation
  std::atomic<int32_t> generation;
  std::atomic<int32_t> failed;      // a rank gave up: everyone else returns an error too
  std::atomic<int64_t> calls[MAX_RANKS];   // collectives issued per rank (tests read it through the env hook below)
  char pad[4096 - 4 * 4 - 8 * MAX_RANKS];
};

struct FakeComm {
  int rank = 0, nranks = 1;
  Header *h = nullptr;
  char *base = nullptr;
  size_t map_bytes = 0;
  char name[96];
  double *slot(int r) const { return reinterpret_cast<double *>(base + sizeof(Header) + size_t(r) * SLOT_BYTES); }
};

double now_s() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}

double timeout_s() {
  const char *e = getenv("LSQAMD_FAKE_RCCL_TIMEOUT_S");
  return e ? atof(e) : 60.0;
}

// sense-reversing barrier over the shared header, bounded
int barrier(FakeComm *c) {
  Header *h = c->h;
  if (c->nranks == 1) return OK;
  const int32_t gen = h->generation.load(std::memory_order_acquire);
  if (h->arrive.fetch_add(1, std::memory_order_acq_rel) + 1 == c->nranks) {
    h->arrive.store(0, std::memory_order_relaxed);
    h->generation.fetch_add(1, std::memory_order_release);
    return OK;
  }
  const double t0 = now_s(), lim = timeout_s();
  int spins = 0;
  while (h->generation.load(std::memory_order_acquire) == gen) {
    if (h->failed.load(std::memory_order_relaxed)) return SYSTEM_ERROR;
    if (++spins > 2000) {
      sched_yield();
      if ((spins & 1023) == 0 && now_s() - t0 > lim) {
        h->failed.store(1);
        return SYSTEM_ERROR;
      }
    }
  }
  return OK;
}

int stage_out(FakeComm *c, const void *dev, size_t count, hipStream_t st) {
  if (count * 8 > SLOT_BYTES) return INVALID_ARGUMENT;
  if (hipStreamSynchronize(st) != hipSuccess) return UNHANDLED_HIP;
  if (hipMemcpy(c->slot(c->rank), dev, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return UNHANDLED_HIP;
  return OK;
}

int put_back(void *dev, const double *host, size_t count, hipStream_t st) {
  // ordered behind everything the caller queued before and in front of what it queues next
  if (hipMemcpyAsync(dev, host, count * 8, hipMemcpyHostToDevice, st) != hipSuccess) return UNHANDLED_HIP;
  if (hipStreamSynchronize(st) != hipSuccess) return UNHANDLED_HIP;
  return OK;
}

}  // namespace

extern "C" {

int ncclGetUniqueId(void *id) {
  char *p = static_cast<char *>(id);
  std::memset(p, 0, ID_BYTES);
  timespec t;
  clock_gettime(CLOCK_REALTIME, &t);
  snprintf(p, ID_BYTES, "/lsqamd_fake_rccl_%d_%lld_%ld", (int)getpid(), (long long)t.tv_sec, t.tv_nsec);
  return OK;
}

struct IdByValue { char internal[ID_BYTES]; };

int ncclCommInitRank(void **comm, int nranks, IdByValue id, int rank) {
  if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return INVALID_ARGUMENT;
  id.internal[ID_BYTES - 1] = 0;
  if (id.internal[0] != '/') return INVALID_ARGUMENT;
  FakeComm *c = new FakeComm;
  c->rank = rank;
  c->nranks = nranks;
  snprintf(c->name, sizeof(c->name), "%s", id.internal);
  c->map_bytes = sizeof(Header) + size_t(nranks) * SLOT_BYTES;
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
    if (fd >= 0) close(fd);
    delete c;
    return SYSTEM_ERROR;
  }
  void *m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_NORESERVE, fd, 0);
  close(fd);
  if (m == MAP_FAILED) {
    delete c;
    return SYSTEM_ERROR;
  }
  c->base = static_cast<char *>(m);
  c->h = reinterpret_cast<Header *>(m);      // a fresh shm object is zero-filled: the header starts valid
  c->h->joined.fetch_add(1);
  const double t0 = now_s(), lim = timeout_s();
  while (c->h->joined.load() < nranks) {     // collective: returns when all ranks have joined
    sched_yield();
    if (now_s() - t0 > lim) {
      c->h->failed.store(1);
      munmap(m, c->map_bytes);
      shm_unlink(c->name);
      delete c;
      return SYSTEM_ERROR;
    }
  }
  *comm = c;
  return OK;
}

int ncclCommDestroy(void *comm) {
  FakeComm *c = static_cast<FakeComm *>(comm);
  if (!c) return INVALID_ARGUMENT;
  shm_unlink(c->name);                       // the first rank to leave removes the name; mappings stay valid
  munmap(c->base, c->map_bytes);
  delete c;
  return OK;
}

int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t st) {
  FakeComm *c = static_cast<FakeComm *>(comm);
  if (!c || dtype != 8 || op != 0) return INVALID_ARGUMENT;
  c->h->calls[c->rank].fetch_add(1);
  int rc = stage_out(c, send, count, st);
  if (rc) { c->h->failed.store(1); return rc; }
  if ((rc = barrier(c))) return rc;
  double *acc = static_cast<double *>(malloc(count * 8 + 8));
  std::memcpy(acc, c->slot(0), count * 8);
  for (int r = 1; r < c->nranks; ++r) {
    const double *s = c->slot(r);
    for (size_t i = 0; i < count; ++i) acc[i] += s[i];
  }
  rc = put_back(recv, acc, count, st);
  free(acc);
  if (rc) { c->h->failed.store(1); return rc; }
  return barrier(c);                         // nobody overwrites a slot another rank is still reading
}

int ncclReduceScatter(const void *send, void *recv, size_t recvcount, int dtype, int op, void *comm, hipStream_t st) {
  FakeComm *c = static_cast<FakeComm *>(comm);
  if (!c || dtype != 8 || op != 0) return INVALID_ARGUMENT;
  c->h->calls[c->rank].fetch_add(1);
  int rc = stage_out(c, send, recvcount * c->nranks, st);
  if (rc) { c->h->failed.store(1); return rc; }
  if ((rc = barrier(c))) return rc;
  double *acc = static_cast<double *>(malloc(recvcount * 8 + 8));
  const size_t off = size_t(c->rank) * recvcount;
  std::memcpy(acc, c->slot(0) + off, recvcount * 8);
  for (int r = 1; r < c->nranks; ++r) {
    const double *s = c->slot(r) + off;
    for (size_t i = 0; i < recvcount; ++i) acc[i] += s[i];
  }
  rc = put_back(recv, acc, recvcount, st);
  free(acc);
  if (rc) { c->h->failed.store(1); return rc; }
  return barrier(c);
}

int ncclAllGather(const void *send, void *recv, size_t sendcount, int dtype, void *comm, hipStream_t st) {
  FakeComm *c = static_cast<FakeComm *>(comm);
  if (!c || dtype != 8) return INVALID_ARGUMENT;
  c->h->calls[c->rank].fetch_add(1);
  int rc = stage_out(c, send, sendcount, st);
  if (rc) { c->h->failed.store(1); return rc; }
  if ((rc = barrier(c))) return rc;
  for (int r = 0; r < c->nranks && !rc; ++r)
    if (hipMemcpyAsync(static_cast<double *>(recv) + size_t(r) * sendcount, c->slot(r), sendcount * 8,
                       hipMemcpyHostToDevice, st) != hipSuccess)
      rc = UNHANDLED_HIP;
  if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = UNHANDLED_HIP;
  if (rc) { c->h->failed.store(1); return rc; }
  return barrier(c);
}

const char *ncclGetErrorString(int r) {
  switch (r) {
    case OK: return "no error";
    case UNHANDLED_HIP: return "fake rccl: HIP error";
    case SYSTEM_ERROR: return "fake rccl: a rank is missing or failed (bounded wait expired)";
    case INVALID_ARGUMENT: return "fake rccl: invalid argument";
    default: return "fake rccl: internal error";
  }
}

// (the stand-in's collectives complete inside the call: a group is nothing to do)
int ncclGroupStart() { return OK; }
int ncclGroupEnd() { return OK; }

// test hook (not an RCCL symbol): collectives this rank has issued on the communicator
int64_t lsqamd_fake_rccl_calls(void *comm) {
  FakeComm *c = static_cast<FakeComm *>(comm);
  return c ? c->h->calls[c->rank].load() : -1;
}

}  // extern "C"
