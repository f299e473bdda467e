// gph_main.cpp -- G-PhoCS-hip: the reference's command line (GPhoCS.c:84-238)
//   G-PhoCS-hip [-v] [-d device] <control-file> [secondary-control-file]
// over libgphocs_hip.  The library comes in capacity variants (tighter LDS image = more
// wavefronts per CU); the control file is read once with the default build to learn the model
// dimensions, then the tightest variant that fits runs the chain.
#include "gphocs_hip.h"
#include <dlfcn.h>
#include <libgen.h>
#include <unistd.h>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

struct Variant { const char *file; int leaves, pops, bands; };
static const Variant VARIANTS[] = {{"libgphocs_hip_s.so", 16, 9, 4}, {"libgphocs_hip_l.so", 20, 13, 4}, {"libgphocs_hip.so", 24, 16, 8}};
static const int DEFAULT_VARIANT = 2;   /* the largest capacities: always able to read the control file */

template <class F> static F sym(void *h, const char *name)
{
  void *p = dlsym(h, name);
  if (!p) { fprintf(stderr, "G-PhoCS-hip: %s is missing from the engine library\n", name); exit(2); }
  return (F)p;
}

int main(int argc, char **argv)
{
  int verbose = 0, device = 0, i = 1;
  for (; i < argc && argv[i][0] == '-'; i++) {
    if (!strcmp(argv[i], "-v") || !strcmp(argv[i], "--verbose")) verbose = 1;
    else if (!strcmp(argv[i], "-d") && i + 1 < argc) device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-n") && i + 1 < argc) ++i;   /* thread count of the OpenMP build: accepted, ignored */
    else { fprintf(stderr, "usage: %s [-v] [-d device] <control-file> [secondary-control-file]\n", argv[0]); return 1; }
  }
  if (i >= argc) { fprintf(stderr, "usage: %s [-v] [-d device] <control-file> [secondary-control-file]\n", argv[0]); return 1; }
  const char *ctl = argv[i], *ctl2 = i + 1 < argc ? argv[i + 1] : nullptr;
  char self[PATH_MAX];
  ssize_t k = readlink("/proc/self/exe", self, sizeof self - 1);
  if (k <= 0) { perror("readlink"); return 2; }
  self[k] = 0;
  const std::string dir = dirname(self);
  const char *forced = getenv("GPHOCS_HIP_LIB");
  std::string path = forced ? forced : dir + "/" + VARIANTS[DEFAULT_VARIANT].file;
  void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "G-PhoCS-hip: cannot load %s: %s\n", path.c_str(), dlerror()); return 2; }
  if (!forced) {
    gph_control *c = nullptr;
    gph_config cfg;
    if (sym<decltype(&gph_control_read)>(h, "gph_control_read")(ctl, ctl2, &c)) return 1;
    sym<decltype(&gph_control_get)>(h, "gph_control_get")(c, &cfg, nullptr, nullptr);
    const int n = cfg.n, K = cfg.K, B = cfg.B;
    sym<decltype(&gph_control_free)>(h, "gph_control_free")(c);
    for (const Variant &v : VARIANTS)
      if (n <= v.leaves && K <= v.pops && B <= v.bands) {
        if (strcmp(v.file, VARIANTS[DEFAULT_VARIANT].file)) {
          std::string p2 = dir + "/" + v.file;
          void *h2 = dlopen(p2.c_str(), RTLD_NOW | RTLD_LOCAL);
          if (h2) h = h2;
        }
        break;
      }
  }
  return sym<decltype(&gph_run_control_file)>(h, "gph_run_control_file")(ctl, ctl2, device, verbose) ? 1 : 0;
}
