// gph_program.cpp -- the reference's main() + the trace-file side of performMCMC
// (GPhoCS.c:84-238, 1232-1330, 1763-1769) over the engine: same control file, same sequence
// file, same trace file.  Everything per-locus runs on the MI355X engine; there is no CPU path.
#include "../../include/gphocs_hip.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

extern "C" int gph_run_control_file(const char *ctl, const char *ctl2, int32_t device, int32_t verbose)
{
  gph_control *C = nullptr;
  gph_loci *LC = nullptr;
  gph_engine *E = nullptr;
  gph_mcmc *M = nullptr;
  gph_config cfg;
  gph_mcmc_config mc;
  gph_control_info info;
  char err[512] = "";
  int rc;
  FILE *trace = nullptr;
  auto fail = [&](int code, const char *what) {
    fprintf(stderr, "gphocs_hip: %s failed (status %d)%s%s\n", what, code, err[0] ? ": " : "", err);
    if (trace) fclose(trace);
    if (M) gph_mcmc_destroy(M);
    if (E) gph_engine_destroy(E);
    if (LC) gph_loci_free(LC);
    if (C) gph_control_free(C);
    return code;
  };
  printf("Reading control settings from file %s...\n", ctl);
  if ((rc = gph_control_read(ctl, ctl2, &C))) return fail(rc, "reading the control file");
  gph_control_get(C, &cfg, &mc, &info);
  printf("Done.\n");
  if (info.findFinetunes) return fail(GPH_EARG, "find-finetunes TRUE (not supported by the engine driver; give explicit finetunes)");
  if (info.mutRateMode == 1) return fail(GPH_EARG, "locus-mut-rate VAR (UpdateLocusRate is serial over loci upstream and not offloaded)");
  if (mc.seed < 0) mc.seed = abs(2 * (int)time(NULL) + 1);   /* GPhoCS.c:188-191 */
  if (verbose) printf("\nRandom seed set to %d\n", mc.seed);

  auto t0 = std::chrono::steady_clock::now();
  if ((rc = gph_loci_read(C, nullptr, 0, &LC, err, sizeof err))) return fail(rc, "reading the sequence file");
  int64_t L = 0;
  int32_t n = 0;
  const int64_t *offs; const uint8_t *leaf, *ph; const int32_t *cnt, *unph; const double *rates;
  gph_loci_arrays(LC, &L, &n, &offs, &leaf, &ph, &cnt, &rates, &unph);
  {
    int64_t up = 0;
    for (int64_t g = 0; g < L; g++) up += unph[g];
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("Read %lld loci over %d samples: %lld patterns (%.2f per locus) -> %lld phased patterns (%.2f per locus) in %.2f s.\n",
           (long long)L, n, (long long)up, (double)up / L, (long long)offs[L], (double)offs[L] / L, sec);
  }
  cfg.L_total = L;
  cfg.locus_begin = 0;
  cfg.device = device;
  if ((rc = gph_engine_create(&cfg, &E))) return fail(rc, "gph_engine_create");
  if ((rc = gph_engine_load_loci(E, L, offs, leaf, ph, cnt, info.mutRateMode == 2 ? rates : nullptr))) return fail(rc, "gph_engine_load_loci");
  if ((rc = gph_mcmc_create(E, &cfg, &mc, &M))) return fail(rc, "gph_mcmc_create");

  // trace file header, GPhoCS.c:1273-1311
  trace = fopen(info.traceFile, "w");
  if (!trace) { snprintf(err, sizeof err, "Could not open trace file %s", info.traceFile); return fail(GPH_EARG, "opening the trace file"); }
  fprintf(trace, "Sample");
  for (int p = 0; p < cfg.K; p++) fprintf(trace, "\ttheta_%s", gph_control_pop_name(C, p));
  for (int p = cfg.Kc; p < cfg.K; p++) fprintf(trace, "\ttau_%s", gph_control_pop_name(C, p));
  for (int b = 0; b < cfg.B; b++)
    fprintf(trace, "\tm_%s->%s", gph_control_pop_name(C, cfg.bandSrc[b]), gph_control_pop_name(C, cfg.bandTgt[b]));
  for (int p = 0; p < cfg.Kc; p++)
    if (mc.updateSampleAge[p] || mc.sampleAge[p] > 0.0) fprintf(trace, "\ttau_%s", gph_control_pop_name(C, p));
  fprintf(trace, "\tData-ld-ln\tFull-ld-ln\n");

  printf("Starting MCMC: %d burnin, %d running, sampled every %d iteration(s).\n", info.burnin, info.numSamples, info.sampleSkip);
  int64_t totalCoals = 0;
  if ((rc = gph_mcmc_initialize(M, &totalCoals))) return fail(rc, "gph_mcmc_initialize");
  std::vector<double> vals(mc.numParameters + 4, 0.0);
  double logL = 0, dataL = 0;
  auto t1 = std::chrono::steady_clock::now();
  for (int it = -info.burnin; it < info.numSamples; it++) {
    if ((rc = gph_mcmc_iteration(M, it))) return fail(rc, "gph_mcmc_iteration");
    if (it >= 0 && it % (info.sampleSkip + 1) == 0) {   /* GPhoCS.c:1763-1769 */
      gph_mcmc_param_vals(M, vals.data(), mc.numParameters);
      gph_mcmc_get_state(M, &logL, &dataL, nullptr, nullptr, nullptr);
      fprintf(trace, "%d\t", it);
      for (int i = 0; i < mc.numParameters; i++) fprintf(trace, "%8.5f\t", vals[i] * mc.printFactors[i]);
      fprintf(trace, "\t%.6f\t%.6f\n", logL, dataL);
      fflush(trace);
    }
    if ((it + 1) % mc.samplesPerLog == 0) {
      int64_t a[9];
      gph_mcmc_accept_counts(M, a);
      gph_mcmc_get_state(M, &logL, &dataL, nullptr, nullptr, nullptr);
      double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
      printf("%7d   coal %lld  mig %lld  spr %lld  theta %lld  migrate %lld  tau %lld  mix %lld | %.6f | %.1f s\n", it + 1,
             (long long)a[0], (long long)a[1], (long long)a[2], (long long)a[3], (long long)a[4], (long long)a[5],
             (long long)a[6], dataL, sec);
      fflush(stdout);
    }
  }
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
  printf("MCMC finished. Time used: %.2f s (%.3f iterations/s).\n", sec, (info.burnin + info.numSamples) / (sec > 0 ? sec : 1));
  fclose(trace);
  gph_mcmc_destroy(M);
  gph_engine_destroy(E);
  gph_loci_free(LC);
  gph_control_free(C);
  return GPH_OK;
}
