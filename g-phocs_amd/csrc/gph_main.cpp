// gph_main.cpp -- G-PhoCS-hip: the reference's command line (GPhoCS.c:84-238)
//   G-PhoCS-hip [-v] [-d device] [-g gpus] <control-file> [secondary-control-file]
// over libgphocs_hip.  The library comes in capacity variants (tighter LDS image = more
// wavefronts per CU); the control file is read once with the default build to learn the model
// dimensions, then the tightest variant that fits runs the chain.
//
// -g N: ONE chain over N GPUs (the analogue of the reference's `-n threads`, GPhoCS.c:95, 116-145: a static split
// of the loci, MultiCoreUtils.h:8).  The launcher forks N children BEFORE anything touches a GPU; child r loads the
// library, takes device r, holds the r-th contiguous block of loci and runs the same chain; the reduced vectors of
// the per-locus loops travel by RCCL all-gather on each child's stream (the id of the communicator goes from child
// 0 to the others through a shared page).  With fewer devices than ranks (tests on a 1-GPU box) the ranks share
// devices and exchange through that shared page instead -- RCCL refuses two ranks on one GPU.
#include "gphocs_hip.h"
#include <dlfcn.h>
#include <libgen.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <cerrno>
#include <unistd.h>
#include <atomic>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct Variant { const char *file; int leaves, pops, bands; };
static const Variant VARIANTS[] = {{"libgphocs_hip_s.so", 16, 9, 4}, {"libgphocs_hip_l.so", 20, 13, 4}, {"libgphocs_hip.so", 24, 16, 8},
                                   {"libgphocs_hip_x.so", 32, 32, 16}, {"libgphocs_hip_g.so", 48, 16, 8}, {"libgphocs_hip_h.so", 64, 40, 16},
                                   {"libgphocs_hip_b.so", 64, 40, 100}, {"libgphocs_hip_n.so", 200, 40, 100}};
/* the library whose gph_control_read reads the control file BEFORE the capacity variant is chosen: the control parser has
 * no compile-time capacities (it keeps the reference's own caps, patch.h:17-22, in every build), so any variant can read
 * any control file; the dimensions it returns then select the tightest variant above */
static const int DEFAULT_VARIANT = 5;

template <class F> static F sym(void *h, const char *name)
{
  void *p = dlsym(h, name);
  if (!p) { fprintf(stderr, "G-PhoCS-hip: %s is missing from the engine library\n", name); exit(2); }
  return (F)p;
}

struct Mailbox {                       // first bytes of the page the ranks share
  std::atomic<int> id_ready, ndev;
  char id[GPH_COMM_ID_BYTES];
};

static void *load_engine(const std::string &dir, const char *ctl, const char *ctl2)
{
  const char *forced = getenv("GPHOCS_HIP_LIB");
  std::string path = forced ? forced : dir + "/" + VARIANTS[DEFAULT_VARIANT].file;
  void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h && !forced) { path = dir + "/" + VARIANTS[2].file; h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL); }
  if (!h) { fprintf(stderr, "G-PhoCS-hip: cannot load %s: %s\n", path.c_str(), dlerror()); exit(2); }
  if (forced) return h;
  gph_control *c = nullptr;
  gph_config cfg;
  if (sym<decltype(&gph_control_read)>(h, "gph_control_read")(ctl, ctl2, &c)) exit(1);
  sym<decltype(&gph_control_get)>(h, "gph_control_get")(c, &cfg, nullptr, nullptr);
  const int n = cfg.n, K = cfg.K, B = cfg.B;
  sym<decltype(&gph_control_free)>(h, "gph_control_free")(c);
  for (const Variant &v : VARIANTS)
    if (n <= v.leaves && K <= v.pops && B <= v.bands) {
      std::string p2 = dir + "/" + v.file;
      if (p2 != path) { void *h2 = dlopen(p2.c_str(), RTLD_NOW | RTLD_LOCAL); if (h2) h = h2; }
      break;
    }
  return h;
}

// the launcher's children, for the signal handler: a launcher that is told to stop takes its ranks with it
static pid_t g_kids[64];
static volatile sig_atomic_t g_nkids = 0;
static void forward_signal(int sig)
{
  for (int r = 0; r < g_nkids; r++) if (g_kids[r] > 0) kill(g_kids[r], sig);
  _exit(128 + sig);
}

static int usage(const char *a0)
{
  fprintf(stderr, "usage: %s [-v] [-d device] [-g gpus] <control-file> [secondary-control-file]\n", a0);
  return 1;
}

int main(int argc, char **argv)
{
  int verbose = 0, device = 0, gpus = 1, i = 1;
  for (; i < argc && argv[i][0] == '-'; i++) {
    if (!strcmp(argv[i], "-v") || !strcmp(argv[i], "--verbose")) verbose = 1;
    else if (!strcmp(argv[i], "-d") && i + 1 < argc) device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-g") && i + 1 < argc) gpus = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-n") && i + 1 < argc) ++i;   /* thread count of the OpenMP build: accepted, ignored */
    else return usage(argv[0]);
  }
  if (i >= argc || gpus < 1 || gpus > 64) return usage(argv[0]);
  const char *ctl = argv[i], *ctl2 = i + 1 < argc ? argv[i + 1] : nullptr;
  char self[PATH_MAX];
  ssize_t k = readlink("/proc/self/exe", self, sizeof self - 1);
  if (k <= 0) { perror("readlink"); return 2; }
  self[k] = 0;
  const std::string dir = dirname(self);
  if (gpus == 1) {
    void *h = load_engine(dir, ctl, ctl2);
    return sym<decltype(&gph_run_control_file)>(h, "gph_run_control_file")(ctl, ctl2, device, verbose) ? 1 : 0;
  }

  // ---- one chain over `gpus` ranks.  Nothing below this line touches a GPU in the parent.
  const size_t page = 1 << 20;
  char *shared = (char *)mmap(nullptr, page, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (shared == MAP_FAILED) { perror("mmap"); return 2; }
  memset(shared, 0, page);
  std::vector<pid_t> kids(gpus, -1);
  for (int r = 0; r < gpus; r++) {
    pid_t p = fork();
    if (p < 0) { perror("fork"); for (int q = 0; q < r; q++) kill(kids[q], SIGTERM); return 2; }
    if (p == 0) {
      prctl(PR_SET_PDEATHSIG, SIGTERM);          /* the launcher died (killed -9, say): do not linger in an exchange */
      if (getppid() == 1) _exit(2);
      void *h = load_engine(dir, ctl, ctl2);
      Mailbox *mb = (Mailbox *)shared;
      // how many devices are there?  Child 0 asks (the first GPU call of this process tree) and tells the others
      // through the engine library: gph_comm_create_* are the only entry points that need to know
      auto create_rccl = sym<decltype(&gph_comm_create_rccl)>(h, "gph_comm_create_rccl");
      auto attach_shm = sym<decltype(&gph_comm_attach_shm)>(h, "gph_comm_attach_shm");
      auto shm_bytes = sym<decltype(&gph_comm_shm_bytes)>(h, "gph_comm_shm_bytes");
      auto unique_id = sym<decltype(&gph_comm_unique_id)>(h, "gph_comm_unique_id");
      auto destroy = sym<decltype(&gph_comm_destroy)>(h, "gph_comm_destroy");
      auto ndevices = sym<int (*)()>(h, "gph_device_count");
      const int ndev = ndevices();
      if (ndev < 1) { fprintf(stderr, "G-PhoCS-hip: no HIP device\n"); _exit(2); }
      const int mydev = device + r < ndev ? device + r : (device + r) % ndev;
      const bool share = gpus > ndev - device || getenv("GPHOCS_HIP_SHM");   /* ranks would share a device */
      gph_comm *comm = nullptr;
      if (share) {
        if (shm_bytes(gpus) + 4096 > page) { fprintf(stderr, "G-PhoCS-hip: shared page too small\n"); _exit(2); }
        comm = attach_shm(shared + 4096, r, gpus);
        if (r == 0 && verbose) printf("%d ranks on %d device(s): host shared-memory exchange (RCCL wants one GPU per rank)\n", gpus, ndev);
      } else {
        if (r == 0) {
          if (unique_id(mb->id)) _exit(2);
          mb->id_ready.store(1, std::memory_order_release);
        } else {
          for (int spin = 0; !mb->id_ready.load(std::memory_order_acquire); spin++) { if (spin > 600000) _exit(2); usleep(100); }
        }
        comm = create_rccl(mb->id, r, gpus, mydev);
      }
      if (!comm) { fprintf(stderr, "G-PhoCS-hip: rank %d could not join the communicator\n", r); _exit(2); }
      int rc = sym<decltype(&gph_run_control_file_comm)>(h, "gph_run_control_file_comm")(ctl, ctl2, mydev, verbose, comm);
      fflush(stdout);
      if (rc == 0) destroy(comm);
      _exit(rc ? 1 : 0);
    }
    kids[r] = p;
    g_kids[r] = p;
    g_nkids = r + 1;
  }
  signal(SIGTERM, forward_signal);
  signal(SIGINT, forward_signal);
  // a rank that fails takes the job down: the others would wait for it in the next exchange
  int status = 0, left = gpus, bad = 0;
  while (left > 0) {
    pid_t p = wait(&status);
    if (p < 0) {
      if (errno == EINTR) continue;
      perror("G-PhoCS-hip: wait");             /* no children left to wait for although some are unaccounted: a failure */
      for (int r = 0; r < gpus; r++) kill(kids[r], SIGTERM);
      return 2;
    }
    left--;
    const bool failed = !WIFEXITED(status) || WEXITSTATUS(status) != 0;
    if (failed && !bad) {
      bad = 1;
      for (int r = 0; r < gpus; r++) if (kids[r] != p) kill(kids[r], SIGTERM);
    }
  }
  return bad;
}
