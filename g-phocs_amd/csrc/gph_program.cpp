// gph_program.cpp -- the reference's main() + the trace-file side of performMCMC
// (GPhoCS.c:84-238, 1232-1330, 1763-1769) over the engine: same control file, same sequence
// file, same trace file.  Everything per-locus runs on the MI355X engine; there is no CPU path.
#include "../../include/gphocs_hip.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

namespace {
// one step of the find-finetunes search for one step size (GPhoCS.c:1896-2180, the same block per proposal
// type): bisection on [min, max] towards 35 % +- 5 % acceptance
struct Finetune {
  double v, lo = 0.0, hi = 10.0;   /* finetuneMins / finetuneMaxes, MAX_FINETUNE = 10 (GPhoCS.h:21-24) */
  void adjust(double pct)
  {
    const double TARGET = 35, RANGE = 5, RES = 0.0000001, MAXF = 10;
    if (pct > TARGET + RANGE) {
      lo = v;
      if (hi - lo < RES) {
        if (hi >= MAXF) hi = lo = MAXF;
        else hi *= 2.0;
      }
    } else if (pct < TARGET - RANGE) {
      hi = v;
      if (hi - lo < RES) lo /= 2.0;
    }
    v = 0.5 * (hi + lo);
  }
};
}   // namespace

static int run_control_file(const char *ctl, const char *ctl2, int32_t device, int32_t verbose,
                            int32_t rank, int32_t world, gph_allreduce_fn allreduce, void *user, gph_comm *comm)
{
  if (world < 1 || rank < 0 || rank >= world || (world > 1 && !allreduce && !comm)) return GPH_EARG;
  const bool lead = rank == 0;   /* rank 0 talks and writes the trace file; every rank runs the same chain */
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
  if (lead) printf("Reading control settings from file %s...\n", ctl);
  if (lead && ctl2) printf("Reading control settings from secondary file %s...\n", ctl2);   /* GPhoCS.c:157-158 */
  if ((rc = gph_control_read(ctl, ctl2, &C))) return fail(rc, "reading the control file");
  gph_control_get(C, &cfg, &mc, &info);
  if (lead) printf("Done.\n");
  if (mc.seed < 0) {
    if (world > 1) return fail(GPH_EARG, "random-seed must be given in the control file when several ranks run one chain");
    mc.seed = abs(2 * (int)time(NULL) + 1);   /* GPhoCS.c:188-191 */
  }
  if (verbose && lead) printf("\nRandom seed set to %d\n", mc.seed);

  auto t0 = std::chrono::steady_clock::now();
  if ((rc = gph_loci_read(C, nullptr, 0, &LC, err, sizeof err))) {
    /* a rejected rate file (locus-mut-rate FIXED): upstream's two lines (readRateFile, GPhoCS.c:491-579, and its caller :1149-1154) */
    if (info.mutRateMode == 2 && (strstr(err, "rate") || strstr(err, "Rate")) && lead) {
      fprintf(stderr, "Error: %s\n", err);
      if (!strstr(err, "Could not find")) fprintf(stderr, "Error: Unable to reading rate file '%s'. Aborting !!\n", info.rateFile);
    }
    return fail(rc, "reading the sequence file");
  }
  int64_t L = 0;
  int32_t n = 0;
  const int64_t *offs; const uint8_t *leaf; const uint16_t *ph; const int32_t *cnt, *unph; const double *rates;
  gph_loci_arrays(LC, &L, &n, &offs, &leaf, &ph, &cnt, &rates, &unph);
  {
    int64_t up = 0;
    for (int64_t g = 0; g < L; g++) up += unph[g];
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (lead) {   /* the progress text of readSeqFile (AlignmentProcessor.c:547, 562, 678-687) and main's blank line (GPhoCS.c:233) */
      printf("Reading sequence data...  %lld loci, as specified in sequence file %s.\n", (long long)L, info.seqFile);
      printf("Reading loci (.=100 loci): ");
      for (int64_t g = 1; g <= L; g++)
        if (g % 100 == 0) { printf("."); if (g % 1000 == 0) { printf(" "); if (g % 10000 == 0) printf("\n"); } }
      printf("\n");
      if (verbose) printf("Read %lld loci over %d samples: %lld patterns (%.2f per locus) -> %lld phased patterns (%.2f per locus) in %.2f s.\n",
                          (long long)L, n, (long long)up, (double)up / L, (long long)offs[L], (double)offs[L] / L, sec);
      printf("\n");
    }
  }
  // loci shard in contiguous blocks of ceil(L / world) (OpenMP static scheduling of the reference, MultiCoreUtils.h:8)
  const int64_t per = (L + world - 1) / world, lb = std::min<int64_t>(rank * per, L), le = std::min<int64_t>((rank + 1) * per, L);
  if ((int64_t)(world - 1) * per >= L) {
    /* the same test on every rank: the whole job stops here, nobody is left waiting in an exchange */
    if (lead) fprintf(stderr, "gphocs_hip: %d ranks over %lld loci in blocks of %lld leave the last rank(s) without loci -- use at most %lld ranks\n",
                      world, (long long)L, (long long)per, (long long)((L + per - 1) / per));
    return fail(GPH_EARG, "sharding the loci over the ranks");
  }
  (void)le;
  cfg.L_total = L;
  cfg.locus_begin = lb;
  cfg.device = device;
  if ((rc = gph_engine_create(&cfg, &E))) return fail(rc, "gph_engine_create");
  if (world > 1 && allreduce && (rc = gph_engine_set_allreduce(E, allreduce, user))) return fail(rc, "gph_engine_set_allreduce");
  if (comm && (rc = gph_engine_set_comm(E, comm))) return fail(rc, "gph_engine_set_comm");
  /* pattern offsets are absolute indices into the pattern arrays: the shard is a window of the offset array */
  if ((rc = gph_engine_load_loci(E, le - lb, offs + lb, leaf, ph, cnt, info.mutRateMode == 2 ? rates + lb : nullptr))) return fail(rc, "gph_engine_load_loci");
  if ((rc = gph_mcmc_create(E, &cfg, &mc, &M))) return fail(rc, "gph_mcmc_create");

  // trace file header, GPhoCS.c:1273-1311
  trace = fopen(lead ? info.traceFile : "/dev/null", "w");
  if (!trace) { snprintf(err, sizeof err, "Could not open trace file %s", info.traceFile); return fail(GPH_EARG, "opening the trace file"); }
  fprintf(trace, "Sample");
  for (int p = 0; p < cfg.K; p++) fprintf(trace, "\ttheta_%s", gph_control_pop_name(C, p));
  for (int p = cfg.Kc; p < cfg.K; p++) fprintf(trace, "\ttau_%s", gph_control_pop_name(C, p));
  for (int b = 0; b < cfg.B; b++)
    fprintf(trace, "\tm_%s->%s", gph_control_pop_name(C, cfg.bandSrc[b]), gph_control_pop_name(C, cfg.bandTgt[b]));
  for (int p = 0; p < cfg.Kc; p++)
    if (mc.updateSampleAge[p] || mc.sampleAge[p] > 0.0) fprintf(trace, "\ttau_%s", gph_control_pop_name(C, p));
  if (info.mutRateMode == 1) fprintf(trace, "\tVariance-Mut");   /* GPhoCS.c:1311-1312 */
  fprintf(trace, "\tData-ld-ln\tFull-ld-ln\n");

  if (lead) printf("Starting MCMC: %d burnin, %d running, sampled every %d iteration(s).\n", info.burnin, info.numSamples, info.sampleSkip);
  if (lead && info.mutRateMode == 2) printf("Reading locus rates from file %s... ", info.rateFile);   /* readRateFile's progress text (GPhoCS.c:514), printed from initializeMCMC upstream */
  int64_t totalCoals = 0;
  if ((rc = gph_mcmc_initialize(M, &totalCoals))) return fail(rc, "gph_mcmc_initialize");
  std::vector<double> vals(mc.numParameters + 4, 0.0);
  double logL = 0, dataL = 0;
  auto t1 = std::chrono::steady_clock::now();
  const time_t wall0 = time(NULL);
  auto printtime = [&](char *buf, size_t len) {   /* printtime(), utils.c:314-326 */
    const long t = (long)(time(NULL) - wall0);
    const long h = t / 3600, mm = (t % 3600) / 60, ss = t - (t / 60) * 60;
    if (h) snprintf(buf, len, "%ld:%02ld:%02ld", h, mm, ss);
    else snprintf(buf, len, "%2ld:%02ld", mm, ss);
    return buf;
  };
  char tbuf[64];
  // the log on stdout: title, GPhoCS.c:1329, 1356-1374
  if (lead) {
    printf("There are %d parameters in the model.\n", mc.numParameters);
    printf("Samples   CoalTimes MigTimes  SPRs      Thetas    MigRates ");
    for (int p = 0; p < cfg.K; p++) if (p >= cfg.Kc || mc.updateSampleAge[p]) printf("TAU_%2d    ", p);
    printf("RbberBnd  MutRates  Mixing    | DATA-ln-ld |  TIME\n");
    printf("-------------------------------------------------------------"
           "-------------------------------------------------------------"
           "---------------------------\n");
    fflush(stdout);
  }
  int logsPerLine = info.logsPerLine;
  // log periods and the find-finetunes search, GPhoCS.c:1401-1447, 1808-2249
  int samplesPerLog = mc.samplesPerLog, findingFinetunes = 0;
  Finetune fCoal{mc.ftCoalTime}, fMig{mc.ftMigTime}, fTheta{mc.ftTheta}, fRate{mc.ftMigRate}, fMix{mc.ftMixing}, fLocus{mc.ftLocusRate}, fAdmix{-1.0};
  /* fAdmix: upstream also bisects the (unused) admixture step and prints it (GPhoCS.c:2040-2066, 2190) */
  std::vector<Finetune> fTau(cfg.K);
  for (int p = 0; p < cfg.K; p++) fTau[p].v = mc.ftTaus[p];
  auto push_finetunes = [&]() {
    std::vector<double> taus(cfg.K);
    for (int p = 0; p < cfg.K; p++) taus[p] = fTau[p].v;
    gph_mcmc_set_locus_rate_finetune(M, fLocus.v);
    return gph_mcmc_set_finetunes(M, fCoal.v, fMig.v, fTheta.v, fRate.v, fMix.v, taus.data());
  };
  if (info.findFinetunes) {
    findingFinetunes = 1;
    samplesPerLog = info.findFinetunesSamplesPerStep;
    logsPerLine = 1;
    if (lead) {
      printf("   ---  Dynamically finding finetune settings for the first %d samples, updating finetunes every %d samples  ---- \n",
             samplesPerLog * info.findFinetunesNumSteps, samplesPerLog);
      printf("------------------------------------------------------------"
             "------------------------------------------------------------"
             "-----------------------------\n");
    }
    for (Finetune *f : {&fCoal, &fMig, &fTheta, &fRate, &fMix, &fLocus, &fAdmix}) if (f->v < 0) f->v = 1.0;
    for (int p = 0; p < cfg.K; p++) if (fTau[p].v < 0) fTau[p].v = 1.0;
    if ((rc = push_finetunes())) return fail(rc, "gph_mcmc_set_finetunes");
    gph_mcmc_set_log_period(M, samplesPerLog);
  }
  int64_t logCount = 1, a0[9] = {0}, a[9], aLocus0 = 0, aLocus = 0;
  std::vector<int64_t> t0v(cfg.K, 0), tv(cfg.K, 0);
  /* TAU columns of the log exactly as upstream accumulates them (GPhoCS.c:1620-1650): the whole accept array is
   * added after UpdateTau AND after UpdateSampleAge, so an ancestral population's count goes in twice and an
   * estimated sample age's goes in once stale (the previous iteration's) and once fresh */
  std::vector<int64_t> tauPrevTotal(cfg.K, 0), tauLastIter(cfg.K, 0), tauShown(cfg.K, 0);
  for (int it = -info.burnin; it < info.numSamples; it++) {
    if ((rc = gph_mcmc_iteration(M, it))) return fail(rc, "gph_mcmc_iteration");
    gph_mcmc_tau_accept_counts(M, tv.data());
    for (int p = 0; p < cfg.K; p++) {
      const int64_t now = tv[p] - tauPrevTotal[p];
      tauShown[p] += p >= cfg.Kc ? 2 * now : tauLastIter[p] + now;
      tauLastIter[p] = now;
      tauPrevTotal[p] = tv[p];
    }
    if (it >= 0 && it % (info.sampleSkip + 1) == 0) {   /* GPhoCS.c:1763-1769 */
      gph_mcmc_param_vals(M, vals.data(), mc.numParameters);
      gph_mcmc_get_state(M, &logL, &dataL, nullptr, nullptr, nullptr);
      fprintf(trace, "%d\t", it);
      for (int i = 0; i < mc.numParameters; i++) fprintf(trace, "%8.5f\t", vals[i] * mc.printFactors[i]);
      fprintf(trace, "\t%.6f\t%.6f\n", logL, dataL);
      fflush(trace);
    }
    logCount++;
    if ((it + 1) % samplesPerLog == 0) {
      gph_mcmc_accept_counts(M, a);
      gph_mcmc_tau_accept_counts(M, tv.data());
      gph_mcmc_get_state(M, &logL, &dataL, nullptr, nullptr, nullptr);
      // acceptance percentages of the period exactly as upstream computes them (GPhoCS.c:1821-1853): logCount
      // starts at 1, and the tau counts of a period are added twice (after UpdateTau and again after
      // UpdateSampleAge, GPhoCS.c:1620-1623, 1646-1649)
      const double lc = (double)logCount;
      const double pCoal = (a[0] - a0[0]) * 100.0 / (lc * (double)totalCoals);
      const double pMig = (a[1] - a0[1]) * 100.0 / ((double)(a[7] - a0[7]) + 0.000001);
      const double pSpr = (a[2] - a0[2]) * 100.0 / (lc * 2 * (double)totalCoals);
      const double pTheta = (a[3] - a0[3]) * 100.0 / (lc * cfg.K);
      const double pRate = (a[4] - a0[4]) * 100.0 / (lc * cfg.B + 0.000001);
      const double pMix = (a[6] - a0[6]) * 100.0 / lc;
      gph_mcmc_locus_rate_state(M, &aLocus, nullptr);
      const double pLocus = (aLocus - aLocus0) * 100.0 / (lc * (double)(info.numLoci - 1));   /* GPhoCS.c:1842-1845 */
      if (lead) {   /* the log line, GPhoCS.c:1857-1895 */
        printf("\r%7d   %5.1f%%    %5.1f%%    %5.1f%%    %5.1f%%    %5.1f%%    ", it + 1, pCoal, pMig, pSpr, pTheta, pRate);
        for (int p = 0; p < cfg.K; p++)
          if (p >= cfg.Kc || mc.updateSampleAge[p]) printf("%5.1f%%    ", tauShown[p] * 100.0 / lc);
        printf("%6.1f%%    %5.1f%%    %5.1f%%    ", (a[8] - a0[8]) * 100.0 / (lc * (cfg.K - cfg.Kc)), pLocus, pMix);
        printf("|%12.6f|", logL);
        printf(" %s", printtime(tbuf, sizeof tbuf));
        if ((it + 1) % (samplesPerLog * logsPerLine) == 0) printf("\n");
        fflush(stdout);
      }
      if (findingFinetunes) {
        fCoal.adjust(pCoal); fMig.adjust(pMig); fTheta.adjust(pTheta); fRate.adjust(pRate); fMix.adjust(pMix);
        fLocus.adjust(pLocus);
        fAdmix.adjust(0.0);
        for (int p = cfg.Kc; p < cfg.K; p++) fTau[p].adjust(2 * (tv[p] - t0v[p]) * 100.0 / lc);
        if ((rc = push_finetunes())) return fail(rc, "gph_mcmc_set_finetunes");
        if (lead) printf("          %-9.7lf %-9.7lf           %-9.7lf %-9.7lf %-9.7lf ", fCoal.v, fMig.v, fTheta.v, fRate.v, fAdmix.v);
        for (int p = cfg.Kc; p < cfg.K; p++) if (lead) printf("%-9.7lf ", fTau[p].v);
        if (lead) printf("          %-9.7lf %-9.7lf \n", fLocus.v, fMix.v);
      }
      logCount = 1;
      memcpy(a0, a, sizeof a0);
      aLocus0 = aLocus;
      t0v = tv;
      std::fill(tauShown.begin(), tauShown.end(), 0);
      if (findingFinetunes && it + 1 >= info.findFinetunesSamplesPerStep * info.findFinetunesNumSteps) {
        findingFinetunes = 0;
        samplesPerLog = mc.samplesPerLog;
        gph_mcmc_set_log_period(M, samplesPerLog);
        logsPerLine = info.logsPerLine;
        if (lead) {   /* GPhoCS.c:2232-2251 */
          printf("\n");
          printf("-------------------------------------  finetunes  ------------------------------------\n");
          printf("          %8lf  %8lf            %8lf  %8lf  ", fCoal.v, fMig.v, fTheta.v, fRate.v);
          for (int p = 0; p < cfg.K; p++) printf("%8lf  ", fTau[p].v);
          printf("          %8lf  %8lf  \n", fLocus.v, fMix.v);
          printf("--------------------------------------------------------------------------------------\n");
        }
      }
    }
  }
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
  if (lead) {
    printf("\nMCMC finished. Time used: %s\n", printtime(tbuf, sizeof tbuf));   /* GPhoCS.c:2262 */
    if (verbose) printf("(%.2f s, %.3f iterations/s)\n", sec, (info.burnin + info.numSamples) / (sec > 0 ? sec : 1));
  }
  fclose(trace);
  /* a CHECKED build of the library (-DGPH_BOUNDS, tests only) says here whether an index left its array during the run */
  int32_t oob_where = 0, oob_checked = 0;
  (void)gph_engine_debug_oob(E, &oob_where, &oob_checked);
  gph_mcmc_destroy(M);
  gph_engine_destroy(E);
  gph_loci_free(LC);
  gph_control_free(C);
  if (oob_checked && oob_where != 0) {
    fprintf(stderr, "gphocs_hip: checked build: an index left its array at %d (source line + 100000 x file: 1 gph_locus.h, 2 gph_kernels.h; "
                    "8000xx / 9000xx: typed accessors of the image / the dynamic LDS)\n", (int)oob_where);
    return GPH_EKERNEL;
  }
  return GPH_OK;
}

extern "C" int gph_run_control_file_ranked(const char *ctl, const char *ctl2, int32_t device, int32_t verbose,
                                           int32_t rank, int32_t world, gph_allreduce_fn allreduce, void *user)
{
  return run_control_file(ctl, ctl2, device, verbose, rank, world, allreduce, user, nullptr);
}

extern "C" int gph_run_control_file_comm(const char *ctl, const char *ctl2, int32_t device, int32_t verbose, gph_comm *comm)
{
  return run_control_file(ctl, ctl2, device, verbose, gph_comm_rank(comm), gph_comm_world(comm), nullptr, nullptr, comm);
}

extern "C" int gph_run_control_file(const char *ctl, const char *ctl2, int32_t device, int32_t verbose)
{
  return run_control_file(ctl, ctl2, device, verbose, 0, 1, nullptr, nullptr, nullptr);
}
