// gph_comm.cpp -- see gph_comm.h.  Host code (no kernels): RCCL through dlopen + a shared-memory exchange.
#include "gph_comm.h"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#ifndef GPH_HOSTEMU
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#endif

#define GPH_COMM_MAXN 448      /* >= the columns a stage exchanges: counters + 2 * 39 populations + 2 * 100 bands */
#define GPH_COMM_XROW 2320     /* doubles of a reduced row of the widest build (GPH_RED_ROW with 384 columns, gph_types.h) */

struct ShmSeg {                       // one cache line per rank's arrival counter
  std::atomic<uint32_t> magic;
  std::atomic<uint32_t> failed;
  char pad0[56];
  // named segments only: rank r writes its pid into hello[r], the LIVE rank 0 of this run answers with the same value
  // in ack[r].  A leftover segment of a crashed run (same name, right size, right magic) has nobody to answer: the
  // rank unmaps it and opens the name again until it finds the segment rank 0 has just made.
  std::atomic<uint32_t> hello[64], ack[64];
  struct alignas(64) Slot { std::atomic<uint64_t> seq; char pad[56]; } slot[64];
  double data[2][64][GPH_COMM_MAXN];
};

struct gph_comm_group;
struct gph_comm {
  int kind = 0;                       // 1 RCCL, 2 shm, 3 local (thread ranks of one process on one device)
  int rank = 0, world = 1;
  gph_comm_group *group = nullptr;
  // shm
  ShmSeg *seg = nullptr;
  bool owns_mapping = false;
  std::string shm_name;
  uint64_t seq = 0;
#ifndef GPH_HOSTEMU
  // RCCL
  ncclComm_t nccl = nullptr;
  int device = 0;
  hipStream_t hstream = nullptr;      // host-path collectives (the engine's own stream carries the resident ones)
  double *d_in = nullptr, *d_out = nullptr, *h_buf = nullptr;
#endif
};

#ifndef GPH_HOSTEMU
namespace {
struct Rccl {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl *rccl_load();
Rccl *rccl()
{
  static std::once_flag once;
  static Rccl *R = nullptr;
  std::call_once(once, [] { R = rccl_load(); });
  return R;
}
Rccl *rccl_load()
{
  static Rccl R;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) if ((R.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;   /* a copy this process already holds */
  if (!R.h) for (const char *n : names) if ((R.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!R.h) { fprintf(stderr, "gphocs_hip: cannot load librccl (%s) -- several GPUs need RCCL\n", dlerror()); return nullptr; }
  R.GetUniqueId = (decltype(R.GetUniqueId))dlsym(R.h, "ncclGetUniqueId");
  R.CommInitRank = (decltype(R.CommInitRank))dlsym(R.h, "ncclCommInitRank");
  R.AllGather = (decltype(R.AllGather))dlsym(R.h, "ncclAllGather");
  R.CommDestroy = (decltype(R.CommDestroy))dlsym(R.h, "ncclCommDestroy");
  R.GetErrorString = (decltype(R.GetErrorString))dlsym(R.h, "ncclGetErrorString");
  if (!R.GetUniqueId || !R.CommInitRank || !R.AllGather || !R.CommDestroy || !R.GetErrorString) {
    fprintf(stderr, "gphocs_hip: librccl lacks an entry point\n");
    R.h = nullptr;
    return nullptr;
  }
  return &R;
}
}   // namespace
#endif

extern "C" {

int gph_device_count(void)
{
#ifdef GPH_HOSTEMU
  return 1;
#else
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
#endif
}

int gph_comm_unique_id(void *id128)
{
#ifdef GPH_HOSTEMU
  (void)id128;
  return 1;
#else
  static_assert(sizeof(ncclUniqueId) == GPH_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  Rccl *R = rccl();
  if (!R || !id128) return 1;
  ncclUniqueId id;
  ncclResult_t rc = R->GetUniqueId(&id);
  if (rc != ncclSuccess) { fprintf(stderr, "gphocs_hip: ncclGetUniqueId: %s\n", R->GetErrorString(rc)); return 1; }
  memcpy(id128, &id, sizeof id);
  return 0;
#endif
}

gph_comm *gph_comm_create_rccl(const void *id128, int32_t rank, int32_t world, int32_t device)
{
#ifdef GPH_HOSTEMU
  (void)id128; (void)rank; (void)world; (void)device;
  return nullptr;
#else
  Rccl *R = rccl();
  if (!R || !id128 || world < 1 || world > 64 || rank < 0 || rank >= world) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  gph_comm *c = new gph_comm();
  c->kind = 1; c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  ncclResult_t rc = R->CommInitRank(&c->nccl, world, id, rank);
  if (rc != ncclSuccess) {
    fprintf(stderr, "gphocs_hip: ncclCommInitRank(rank %d of %d, device %d): %s (RCCL wants one GPU per rank)\n", rank, world, device, R->GetErrorString(rc));
    delete c;
    return nullptr;
  }
  if (hipStreamCreate(&c->hstream) != hipSuccess || hipMalloc((void **)&c->d_in, sizeof(double) * GPH_COMM_MAXN) != hipSuccess ||
      hipMalloc((void **)&c->d_out, sizeof(double) * GPH_COMM_MAXN * world) != hipSuccess ||
      hipHostMalloc((void **)&c->h_buf, sizeof(double) * GPH_COMM_MAXN * (world + 1), hipHostMallocDefault) != hipSuccess) {
    gph_comm_destroy(c);
    return nullptr;
  }
  return c;
#endif
}

size_t gph_comm_shm_bytes(int32_t world) { (void)world; return sizeof(ShmSeg); }

gph_comm *gph_comm_attach_shm(void *mapping, int32_t rank, int32_t world)
{
  if (!mapping || world < 1 || world > 64 || rank < 0 || rank >= world) return nullptr;
  gph_comm *c = new gph_comm();
  c->kind = 2; c->rank = rank; c->world = world;
  c->seg = (ShmSeg *)mapping;
  return c;
}

gph_comm *gph_comm_create_shm(const char *name, int32_t rank, int32_t world)
{
  if (!name || world < 1 || world > 64 || rank < 0 || rank >= world) return nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  const double limit = 120.0;
  if (rank == 0) {
    /* a fresh, zero-filled segment under the name (a leftover of a crashed run is unlinked first) */
    shm_unlink(name);
    int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd >= 0 && ftruncate(fd, (off_t)sizeof(ShmSeg)) != 0) { close(fd); fd = -1; }
    if (fd < 0) { fprintf(stderr, "gphocs_hip: cannot create the shared-memory segment %s\n", name); return nullptr; }
    void *m = mmap(nullptr, sizeof(ShmSeg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { shm_unlink(name); return nullptr; }
    gph_comm *c = gph_comm_attach_shm(m, rank, world);
    if (!c) { munmap(m, sizeof(ShmSeg)); shm_unlink(name); return nullptr; }
    c->owns_mapping = true;
    c->shm_name = name;
    c->seg->magic.store(0x47504843u, std::memory_order_release);
    /* answer every rank's hello: only a rank that sits on THIS segment gets an answer */
    for (int seen = 1; seen < world;) {
      seen = 1;
      for (int r = 1; r < world; r++) {
        const uint32_t h = c->seg->hello[r].load(std::memory_order_acquire);
        if (h) { c->seg->ack[r].store(h, std::memory_order_release); seen++; }
      }
      if (seen < world) {
        if (elapsed() > limit) { fprintf(stderr, "gphocs_hip: rank 0 waited %.0f s for the other ranks at %s\n", limit, name); gph_comm_destroy(c); return nullptr; }
        usleep(200);
      }
    }
    return c;
  }
  const uint32_t me = (uint32_t)getpid() | 0x80000000u;   /* never 0 */
  while (elapsed() <= limit) {
    int fd = shm_open(name, O_RDWR, 0600);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(ShmSeg)) { if (fd >= 0) close(fd); usleep(1000); continue; }
    void *m = mmap(nullptr, sizeof(ShmSeg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return nullptr;
    ShmSeg *seg = (ShmSeg *)m;
    /* wait (briefly) for the live rank 0 of this run to answer; silence = a leftover segment: open the name again */
    bool ok = false;
    const auto t1 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() < 0.25) {
      if (seg->magic.load(std::memory_order_acquire) == 0x47504843u) {
        seg->hello[rank].store(me, std::memory_order_release);
        if (seg->ack[rank].load(std::memory_order_acquire) == me) { ok = true; break; }
      }
      usleep(200);
    }
    if (!ok) { munmap(m, sizeof(ShmSeg)); continue; }
    gph_comm *c = gph_comm_attach_shm(m, rank, world);
    if (!c) { munmap(m, sizeof(ShmSeg)); return nullptr; }
    c->owns_mapping = true;
    c->shm_name = name;
    return c;
  }
  fprintf(stderr, "gphocs_hip: rank %d found no live shared-memory segment %s within %.0f s\n", rank, name, limit);
  return nullptr;
}

// ---------------------------------------------------------------- thread ranks on one device
// One group per job: a generation barrier for the host threads, per rank two HIP events and the address of the row it
// contributes.  An all-gather = every rank records `ready` behind its reduction kernel, the threads meet, every rank
// queues (wait for ready[r], copy rank r's row) for every r on ITS stream and records `done`, the threads meet again
// and every stream waits for every `done` (nobody overwrites its row while another rank still reads it).  No host
// synchronisation with the device: the stream semantics are the ones of the RCCL all-gather.
struct gph_comm_group {
  int world = 1, device = 0;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0, refs = 0;
  uint64_t gen = 0;
  bool failed = false;
  const double *src[64] = {nullptr};
  double hbuf[64][GPH_COMM_MAXN];
#ifndef GPH_HOSTEMU
  hipEvent_t ready[64], done[64];
  // in-kernel exchange (gph_comm_peer_exchange): per rank two row slots (generation parity) + a generation word, in one
  // device allocation every rank's kernels can address -- the layout a peer-mapped allocation per GPU would have
  double *xrows = nullptr;
  unsigned long long *xflags = nullptr;
#endif
  unsigned long long xgen[64] = {0};    // per rank: generations of the in-kernel exchange published so far (all ranks issue the same reductions)
};
static int group_barrier(gph_comm_group *g)
{
  std::unique_lock<std::mutex> lk(g->m);
  if (g->failed) return 1;
  const uint64_t my = g->gen;
  if (++g->arrived == g->world) { g->arrived = 0; g->gen++; g->cv.notify_all(); return 0; }
  if (!g->cv.wait_for(lk, std::chrono::seconds(300), [&] { return g->gen != my || g->failed; })) {
    fprintf(stderr, "gphocs_hip: a thread rank waited 300 s for the others\n");
    g->failed = true;
    g->cv.notify_all();
  }
  /* the barrier this rank waited at COMPLETED when the generation moved on: a peer that left right after it (gph_comm_destroy
   * marks the group failed to release stragglers) must not turn a finished collective into an error */
  return g->gen != my ? 0 : 1;
}
// a rank that cannot go on (a HIP call failed between the two barriers) releases the others instead of letting them wait out
// the time limit
static int group_fail(gph_comm_group *g)
{
  std::lock_guard<std::mutex> lk(g->m);
  g->failed = true;
  g->cv.notify_all();
  return 1;
}

gph_comm_group *gph_comm_local_group(int32_t world, int32_t device)
{
#ifdef GPH_HOSTEMU
  (void)world; (void)device;
  return nullptr;
#else
  if (world < 1 || world > 64 || hipSetDevice(device) != hipSuccess) return nullptr;
  gph_comm_group *g = new gph_comm_group();
  g->world = world; g->device = device;
  for (int r = 0; r < world; r++)
    if (hipEventCreateWithFlags(&g->ready[r], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->done[r], hipEventDisableTiming) != hipSuccess) { delete g; return nullptr; }
  const size_t rb = sizeof(double) * 2 * GPH_COMM_XROW * (size_t)world, fb = sizeof(unsigned long long) * 64;
  if (hipMalloc((void **)&g->xrows, rb) != hipSuccess || hipMalloc((void **)&g->xflags, fb) != hipSuccess ||
      hipMemset(g->xrows, 0, rb) != hipSuccess || hipMemset(g->xflags, 0, fb) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    if (g->xrows) (void)hipFree(g->xrows);
    if (g->xflags) (void)hipFree(g->xflags);
    g->xrows = nullptr; g->xflags = nullptr;       /* the event-ordered gather still works */
  }
  return g;
#endif
}

gph_comm *gph_comm_create_local(gph_comm_group *g, int32_t rank)
{
  if (!g || rank < 0 || rank >= g->world) return nullptr;
  gph_comm *c = new gph_comm();
  c->kind = 3; c->rank = rank; c->world = g->world; c->group = g;
#ifndef GPH_HOSTEMU
  c->device = g->device;
#endif
  std::lock_guard<std::mutex> lk(g->m);
  g->refs++;
  return c;
}

void gph_comm_destroy(gph_comm *c)
{
  if (!c) return;
#ifndef GPH_HOSTEMU
  if (c->kind == 1) {
    (void)hipSetDevice(c->device);
    if (c->nccl) { Rccl *R = rccl(); if (R) R->CommDestroy(c->nccl); }
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->h_buf) (void)hipHostFree(c->h_buf);
    if (c->hstream) (void)hipStreamDestroy(c->hstream);
  }
#endif
  if (c->kind == 2 && c->owns_mapping) {
    munmap(c->seg, sizeof(ShmSeg));
    if (c->rank == 0 && !c->shm_name.empty()) shm_unlink(c->shm_name.c_str());
  }
  if (c->kind == 3 && c->group) {
    gph_comm_group *g = c->group;
    bool last;
    { std::lock_guard<std::mutex> lk(g->m); last = --g->refs == 0; if (!last) { g->failed = true; g->cv.notify_all(); } }   /* a rank that leaves early releases the others */
    if (last) {
#ifndef GPH_HOSTEMU
      (void)hipSetDevice(g->device);
      for (int r = 0; r < g->world; r++) { (void)hipEventDestroy(g->ready[r]); (void)hipEventDestroy(g->done[r]); }
      if (g->xrows) (void)hipFree(g->xrows);
      if (g->xflags) (void)hipFree(g->xflags);
#endif
      delete g;
    }
  }
  delete c;
}

// In-kernel exchange of the reduced rows (gph_engine.hip: k_reduce_stage): the base of the row slots
// [world][2][GPH_COMM_XROW] and of the per-rank generation words, addressable by every rank's kernels.  Thread ranks: one
// device allocation of the group.  (Ranks on different GPUs would each own their slot in peer-mapped fine-grained memory;
// not built: it could not be run.)  Returns 1 when the communicator offers it.
int gph_comm_peer_exchange(const gph_comm *c, double **rows, unsigned long long **flags, int32_t *row_stride)
{
#ifdef GPH_HOSTEMU
  (void)c; (void)rows; (void)flags; (void)row_stride;
  return 0;
#else
  if (!c || c->kind != 3 || !c->group || !c->group->xrows || !c->group->xflags) return 0;
  /* OPT-IN (GPH_PEER_EXCHANGE=1): a kernel that waits for another stream's kernel needs the two streams on different
   * hardware queues -- HIP hands streams to a few queues round-robin (GPU_MAX_HW_QUEUES, default 4), and in a process that
   * has created and destroyed many streams two thread ranks can end up behind each other: the bounded wait then fails the
   * run (measured: world 3 at the end of the full test suite).  Ranks on GPUs of their own -- what the exchange is for --
   * have queues of their own; thread ranks default to the event-ordered gather. */
  { const char *ov = getenv("GPH_PEER_EXCHANGE"); if (!ov || atoi(ov) == 0) return 0; }
  if (rows) *rows = c->group->xrows;
  if (flags) *flags = c->group->xflags;
  if (row_stride) *row_stride = GPH_COMM_XROW;
  return 1;
#endif
}
// the next generation of the in-kernel exchange for this rank: the counter lives with the row slots and flags it numbers (the
// group), not with an engine -- engines come and go on one communicator
unsigned long long gph_comm_peer_next_gen(gph_comm *c)
{
  if (!c || c->kind != 3 || !c->group) return 0;
  return ++c->group->xgen[c->rank];
}
int gph_comm_world(const gph_comm *c) { return c ? c->world : 1; }
int gph_comm_rank(const gph_comm *c) { return c ? c->rank : 0; }
int gph_comm_on_stream(const gph_comm *c) { return c && (c->kind == 1 || c->kind == 3); }
const char *gph_comm_kind(const gph_comm *c) { return !c ? "none" : c->kind == 1 ? "rccl" : c->kind == 2 ? "shm" : "local"; }

int gph_comm_allgather_stream(gph_comm *c, const double *d_in, double *d_out, int32_t count, void *stream)
{
#ifdef GPH_HOSTEMU
  (void)c; (void)d_in; (void)d_out; (void)count; (void)stream;
  return 1;
#else
  if (c && c->kind == 3) {
    gph_comm_group *g = c->group;
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = sizeof(double) * (size_t)count;
    g->src[c->rank] = d_in;
    if (hipEventRecord(g->ready[c->rank], st) != hipSuccess) return group_fail(g);
    if (group_barrier(g)) return 1;
    for (int r = 0; r < c->world; r++) {
      if (r != c->rank && hipStreamWaitEvent(st, g->ready[r], 0) != hipSuccess) return group_fail(g);
      if (d_out + (size_t)r * count != g->src[r] &&
          hipMemcpyAsync(d_out + (size_t)r * count, g->src[r], bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return group_fail(g);
    }
    if (hipEventRecord(g->done[c->rank], st) != hipSuccess) return group_fail(g);
    if (group_barrier(g)) return 1;
    for (int r = 0; r < c->world; r++)
      if (r != c->rank && hipStreamWaitEvent(st, g->done[r], 0) != hipSuccess) return group_fail(g);
    return 0;
  }
  if (!c || c->kind != 1) return 1;
  Rccl *R = rccl();
  ncclResult_t rc = R->AllGather(d_in, d_out, (size_t)count, ncclDouble, c->nccl, (hipStream_t)stream);
  if (rc != ncclSuccess) { fprintf(stderr, "gphocs_hip: ncclAllGather: %s\n", R->GetErrorString(rc)); return 1; }
  return 0;
#endif
}

int gph_comm_allreduce_host(gph_comm *c, double *sums, int32_t nsum, double *mins, int32_t nmin)
{
  if (!c) return 0;
  const int n = nsum + nmin;
  if (n > GPH_COMM_MAXN || nsum < 0 || nmin < 0) return 1;
  if (n == 0) return 0;
  if (c->kind == 2) {
    ShmSeg *s = c->seg;
    const uint64_t k = ++c->seq;
    double *mine = s->data[k & 1][c->rank];
    if (nsum) memcpy(mine, sums, sizeof(double) * nsum);
    if (nmin) memcpy(mine + nsum, mins, sizeof(double) * nmin);
    s->slot[c->rank].seq.store(k, std::memory_order_release);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < c->world; r++) {
      unsigned spins = 0;
      while (s->slot[r].seq.load(std::memory_order_acquire) < k) {
        if (s->failed.load(std::memory_order_relaxed)) return 1;
        if ((++spins & 1023) == 0) {
          sched_yield();
          if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 900.0) {
            fprintf(stderr, "gphocs_hip: rank %d waited 900 s for rank %d in the shared-memory exchange\n", c->rank, r);
            s->failed.store(1);
            return 1;
          }
        }
      }
    }
    /* rank order: the same additions on every rank */
    for (int i = 0; i < nsum; i++) {
      double a = s->data[k & 1][0][i];
      for (int r = 1; r < c->world; r++) a += s->data[k & 1][r][i];
      sums[i] = a;
    }
    for (int i = 0; i < nmin; i++) {
      double a = s->data[k & 1][0][nsum + i];
      for (int r = 1; r < c->world; r++) { const double v = s->data[k & 1][r][nsum + i]; a = v < a ? v : a; }
      mins[i] = a;
    }
    return 0;
  }
  if (c->kind == 3) {
    /* host path of the thread ranks (the stepwise entry points): a buffer per rank in the group, two meetings */
    gph_comm_group *g = c->group;
    double *mine = g->hbuf[c->rank];
    if (nsum) memcpy(mine, sums, sizeof(double) * nsum);
    if (nmin) memcpy(mine + nsum, mins, sizeof(double) * nmin);
    if (group_barrier(g)) return 1;
    for (int i = 0; i < nsum; i++) {
      double a = g->hbuf[0][i];
      for (int r = 1; r < c->world; r++) a += g->hbuf[r][i];
      sums[i] = a;
    }
    for (int i = 0; i < nmin; i++) {
      double a = g->hbuf[0][nsum + i];
      for (int r = 1; r < c->world; r++) { const double v = g->hbuf[r][nsum + i]; a = v < a ? v : a; }
      mins[i] = a;
    }
    return group_barrier(g);
  }
#ifndef GPH_HOSTEMU
  if (c->kind == 1) {
    Rccl *R = rccl();
    if (hipSetDevice(c->device) != hipSuccess) return 1;
    double *hin = c->h_buf, *hout = c->h_buf + GPH_COMM_MAXN;
    if (nsum) memcpy(hin, sums, sizeof(double) * nsum);
    if (nmin) memcpy(hin + nsum, mins, sizeof(double) * nmin);
    if (hipMemcpyAsync(c->d_in, hin, sizeof(double) * n, hipMemcpyHostToDevice, c->hstream) != hipSuccess) return 1;
    if (R->AllGather(c->d_in, c->d_out, (size_t)n, ncclDouble, c->nccl, c->hstream) != ncclSuccess) return 1;
    if (hipMemcpyAsync(hout, c->d_out, sizeof(double) * n * c->world, hipMemcpyDeviceToHost, c->hstream) != hipSuccess) return 1;
    if (hipStreamSynchronize(c->hstream) != hipSuccess) return 1;
    for (int i = 0; i < nsum; i++) {
      double a = hout[i];
      for (int r = 1; r < c->world; r++) a += hout[(size_t)r * n + i];
      sums[i] = a;
    }
    for (int i = 0; i < nmin; i++) {
      double a = hout[nsum + i];
      for (int r = 1; r < c->world; r++) { const double v = hout[(size_t)r * n + nsum + i]; a = v < a ? v : a; }
      mins[i] = a;
    }
    return 0;
  }
#endif
  return 1;
}

}   // extern "C"
