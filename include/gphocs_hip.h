/*
 * gphocs_hip.h -- C ABI of libgphocs_hip.so, the MI355X-native per-locus likelihood
 * engine for G-PhoCS-style MCMC (drop-in for the reference's per-locus hot path).
 *
 * The reference has no plugin/FFI layer: the seam is the set of plain C entry points
 * performMCMC() calls (upstream src/GPhoCS.h:84-100) operating on process-wide
 * globals.  This ABI exports the same granularity -- batched over loci, an opaque
 * handle instead of globals, an int status instead of exit(-1):
 *
 *   reference (file:line)                              this library
 *   ------------------------------------------------   --------------------------------
 *   processAlignments/createLocusData/initializeLocusData
 *     GPhoCS.c:260-438, LocusDataLikelihood.c:142,239   gph_engine_load_loci
 *   GetMem  patch.c:28                                  gph_engine_create
 *   initRandomGenerator  utils.c:411                    gph_engine_seed
 *   initializeMCMC per-locus loop  GPhoCS.c:1197-1214   gph_engine_init_genealogies
 *   UpdateGB_InternalNode  GPhoCS.c:2287   \
 *   UpdateGB_MigrationNode GPhoCS.c:2439    }           gph_engine_genealogy_sweep
 *   UpdateGB_MigSPR        GPhoCS.c:2598   /
 *   UpdateTau loop 1       GPhoCS.c:3491-3833           gph_engine_tau_evaluate
 *   UpdateTau loop 2       GPhoCS.c:3885-3936           gph_engine_tau_commit
 *   UpdateTau loops 3/4    GPhoCS.c:3965-3989           gph_engine_tau_revert
 *   mixing loops           GPhoCS.c:4793-4801,4818-4848 gph_engine_mixing_evaluate/_commit
 *   UpdateTheta/UpdateMigRates per-locus touch-ups
 *     GPhoCS.c:3084-3093, 3192-3200                     gph_engine_apply_theta/_migrate
 *   computeTotalStats  patch.c:2134                     gph_engine_get_totals
 *   synchronizeEvents  patch.c:3548                     gph_engine_synchronize
 *   checkAll           patch.c:2745                     gph_engine_check_all
 *   performMCMC iteration body GPhoCS.c:1476-1821       gph_mcmc_iteration (host driver)
 *
 * All pointers are plain host pointers unless named *_dev.  Every function returns 0
 * on success or a negative GPH_E* code; the HIP path is the only path -- there is no
 * CPU fallback, and creation fails loudly when no gfx950 device is usable.
 * Not re-entrant per handle (as the reference: called serially from one thread).
 */
#ifndef GPHOCS_HIP_H
#define GPHOCS_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GPH_OK 0
#define GPH_EARG (-1)      /* bad argument / unsupported size */
#define GPH_EHIP (-2)      /* HIP runtime failure (no device, OOM, launch error) */
#define GPH_EKERNEL (-3)   /* a locus reported a fatal consistency error (reference: "Fatal Error NNNN") */
#define GPH_ESTATE (-4)    /* call out of order */

typedef struct gph_engine gph_engine;
typedef struct gph_mcmc gph_mcmc;

typedef struct {
  int32_t n;              /* haploid leaves per locus (dataSetup.numSamples) */
  int32_t Kc, K, B;       /* current pops, all pops, migration bands */
  int32_t rootPop;
  const int32_t *samplesPerPop;                 /* [Kc] haploid leaves per current pop */
  const int32_t *popFather, *popSon0, *popSon1; /* [K], -1 = none */
  const int32_t *bandSrc, *bandTgt;             /* [B] */
  int32_t device;         /* HIP device ordinal */
  int64_t L_total;        /* loci over all ranks (dataSetup.numLoci) */
  int64_t locus_begin;    /* global index of this rank's first locus */
} gph_config;

/* cross-rank reduction hook (one process per GPU): sums[0..nsum) are summed, mins[0..nmin)
 * minimised, in place, over all ranks.  NULL = single rank (or a native communicator, below).
 * nsum + nmin <= 16 + 2K + 2B: one call per reduction point, 6 + #ancestral populations per MCMC iteration.
 * A hook forces a host synchronisation at every reduction point; the native RCCL communicator does not. */
typedef int (*gph_allreduce_fn)(void *user, double *sums, int32_t nsum, double *mins, int32_t nmin);

/* ------------------------------------------------------------------------------------
 * native cross-rank exchange (csrc/gph_comm.cpp), one process per GPU, loci sharded over the ranks.  Replaces the
 * `omp atomic` accumulations of the reference's per-locus loops across GPUs:
 *   RCCL: ncclAllGather of the reduced row over xGMI, queued on the engine's stream -- no host synchronisation;
 *         librccl is dlopen()ed on first use.  RCCL refuses two ranks on one GPU.
 *   shm:  host shared-memory exchange for ranks that share a GPU (tests on a 1-GPU box).
 *   local: the ranks are THREADS of one process that share a device (tests on a 1-GPU box): the all-gather runs on
 *         the engines' streams (HIP events order the device-to-device copies), so the device-resident world > 1
 *         path -- reduction, gather, rank-order combine in k_global -- runs exactly as it does under RCCL.
 * rank 0 makes the id and hands it to the other ranks by any means (pipe, file, torch.distributed). */
typedef struct gph_comm gph_comm;
#define GPH_COMM_ID_BYTES 128
int gph_device_count(void);   /* HIP devices this process sees (the first GPU call of a freshly forked rank) */
int gph_comm_unique_id(void *id128);
gph_comm *gph_comm_create_rccl(const void *id128, int32_t rank, int32_t world, int32_t device);
gph_comm *gph_comm_create_shm(const char *name, int32_t rank, int32_t world);   /* name "/unique-per-run", same on every rank */
gph_comm *gph_comm_attach_shm(void *zeroed_shared_mapping, int32_t rank, int32_t world);
/* thread ranks of one process on one device: one group per job, one communicator per rank (thread); the group is
 * released with its last communicator */
typedef struct gph_comm_group gph_comm_group;
gph_comm_group *gph_comm_local_group(int32_t world, int32_t device);
gph_comm *gph_comm_create_local(gph_comm_group *group, int32_t rank);
size_t gph_comm_shm_bytes(int32_t world);
void gph_comm_destroy(gph_comm *c);
int gph_comm_world(const gph_comm *c);
int gph_comm_rank(const gph_comm *c);
int gph_comm_on_stream(const gph_comm *c);
/* 1 when the communicator's ranks can exchange their reduced rows INSIDE the reduction kernel (row slots + generation words
 * every rank's kernels can address: thread ranks of one process today): a reduction point is then ONE launch per rank, as
 * with a single rank, instead of reduction + all-gather + decision stage.  Opt-in: GPH_PEER_EXCHANGE=1 (see gph_comm.cpp). */
int gph_comm_peer_exchange(const gph_comm *c, double **rows, unsigned long long **flags, int32_t *row_stride);
/* the generation word this rank publishes next (1, 2, ...): counted per communicator, so that a second engine on the same
 * communicator continues the sequence the flags already hold */
unsigned long long gph_comm_peer_next_gen(gph_comm *c);
const char *gph_comm_kind(const gph_comm *c);
int gph_comm_allgather_stream(gph_comm *c, const double *d_in, double *d_out, int32_t count, void *hip_stream);
int gph_comm_allreduce_host(gph_comm *c, double *sums, int32_t nsum, double *mins, int32_t nmin);

typedef struct {
  int64_t accepted_internal, accepted_mignode, accepted_spr;
  double dData_internal, dLog_internal, dLog_mignode, dData_spr, dLog_spr;
  int64_t total_mig_nodes;    /* sum of num_migs after the migration-node sweep (GPhoCS.c:1517-1520) */
} gph_sweep_result;

/* host part of UpdateTau (GPhoCS.c:3224-3461) or, with mode = 1, of UpdateSampleAge
 * (GPhoCS.c:4006-4128; ap = the current population whose sample age moves, son0 = son1 = -1,
 * taub0 = 0, taub1 = father age, tauold/taunew = old/new sample age) */
typedef struct {
  int32_t ap, son0, son1, isRoot, num_aff, mode;
  double tauold, taunew, taub0, taub1, taufactor0, taufactor1;
  int32_t aff_bands[200];     /* 2 * MAX_MIG_BANDS (patch.h:17): a band can enter with its start and its end */
  int32_t start_or_end[200];
  double new_band_ages[200];
} gph_tau_args;

typedef struct {
  int64_t ntj0, ntj1;
  int64_t first_conflict_locus;   /* global locus index, -1 = no migration conflict */
  double genDelta, dataDelta;
} gph_tau_result;

typedef struct {           /* summed over ALL ranks */
  int64_t evals;          /* computeLocusDataLikelihood(useOld=1)-equivalents since last reset */
  int64_t eval_nodes;     /* recomputed internal nodes (R) */
  double eval_bytes;      /* algorithmic bytes 96*R*P + 20*N + 8*U + 8 per evaluation */
  int64_t not_enough_migs;
} gph_counters;

int gph_engine_create(const gph_config *cfg, gph_engine **out);
void gph_engine_destroy(gph_engine *e);
int gph_engine_set_allreduce(gph_engine *e, gph_allreduce_fn fn, void *user);
/* native communicator (not owned by the engine; destroy it after the engine) */
int gph_engine_set_comm(gph_engine *e, gph_comm *c);
/* leafcodes: [Ptot][n] with 0..3 = T,C,A,G and 4 = N; numPhases non-zero on the first phase
 * of each unphased pattern (the reference keeps an int, 2^hets; here a 16-bit word: a count below 2^15 as it is, a count of 2^15 or
 * more -- always a power of two -- as 0x8000 | exponent, GPH_NUMPHASES below); counts on the same rows; pattern_offsets[L+1] */
#define GPH_NUMPHASES(w) ((int)(w) < 0x8000 ? (int)(w) : (1 << ((int)(w) & 31)))
int gph_engine_load_loci(gph_engine *e, int64_t L, const int64_t *pattern_offsets, const uint8_t *leafcodes,
                         const uint16_t *numPhases, const int32_t *counts, const double *mutRates);
int gph_engine_set_model(gph_engine *e, const double *theta, const double *popAge, const double *sampleAge,
                         const double *migRate, const double *bandStart, const double *bandEnd);
int gph_engine_seed(gph_engine *e, uint32_t seed);
int gph_engine_init_genealogies(gph_engine *e, double *sumGenLnL, double *sumDataLnL);
/* flags: 1 = internal-node ages, 2 = migration-node ages, 4 = SPR; fused in one launch */
int gph_engine_genealogy_sweep(gph_engine *e, int32_t flags, double finetuneCoalTime, double finetuneMigTime,
                               gph_sweep_result *out);
int gph_engine_tau_evaluate(gph_engine *e, const gph_tau_args *a, gph_tau_result *out);
int gph_engine_tau_commit(gph_engine *e);
int gph_engine_tau_revert(gph_engine *e, int64_t first_conflict_locus);
int gph_engine_mixing_evaluate(gph_engine *e, double c, double *dataDelta);
int gph_engine_mixing_commit(gph_engine *e, double c, double lnc);
int gph_engine_mixing_revert(gph_engine *e);
/* UpdateLocusRate (GPhoCS.c:4598-4680; `locus-mut-rate VAR alpha`).  The reference loop is serial over loci (each
 * proposal also moves the rate of the reference locus, genRateRef = 0): one wavefront scans the loci in input
 * order with a stateless evaluator, the accepted loci are then rewritten in parallel.  in/out: the three
 * accumulators the reference updates per accepted proposal (dataState.dataLogLikelihood, .logLikelihood,
 * .rateVar); returns the accept count.  finetune <= 0 returns 0 accepted at once (:4606).  With loci sharded over
 * ranks the scan is chained through the ranks in locus order over the all-reduce hook (same additions in the same
 * order: bit-identical to one rank); every rank gets the same result. */
typedef struct gph_locus_rate_result {
  int64_t accepted;
  double dataLogLikelihood, logLikelihood, rateVar;
} gph_locus_rate_result;
int gph_engine_locus_rate_update(gph_engine *e, double finetune, double varRatesAlpha, gph_locus_rate_result *io);
/* starting rates of the local loci in input order, before gph_engine_init_genealogies (initializeMCMC's VAR
 * branch, GPhoCS.c:1157-1178: the host draws and normalises, `draws` = rndu() draws each locus's stream spent
 * on it -- 1 there); variable != 0 makes the rates part of the dumped state */
int gph_engine_set_locus_rates(gph_engine *e, const double *rates, int32_t draws, int32_t variable);
int gph_engine_apply_theta(gph_engine *e, int32_t pop, double lnc, double thetaold, double thetanew);
int gph_engine_apply_migrate(gph_engine *e, int32_t band, double lnc, double old_rate, double new_rate);
/* sums over ALL ranks; num_* returned as doubles holding exact integers */
int gph_engine_get_totals(gph_engine *e, double *coal_stats, double *num_coals, double *mig_stats,
                          double *num_migs);
int gph_engine_synchronize(gph_engine *e, int32_t refresh_genealogy_lnl, double *sumOldGenLnL,
                           double *sumNewGenLnL);
int gph_engine_check_all(gph_engine *e, int32_t *ok, double *sumDataLnL, double *sumGenLnL);
int gph_engine_get_counters(gph_engine *e, gph_counters *out, int32_t reset);
/* After a call returned GPH_EKERNEL: the reference's "Fatal Error NNNN" code and the global index of the FIRST locus that
 * reported it (-1 when the failure is not one locus's: checkAll's accumulator test, synchronizeEvents' flag).  The engine
 * has then already printed, to stderr, the code, the locus and -- what printGenealogyAndExit(gen, ...) prints upstream,
 * GPhoCS.c:660-676 -- that locus's genealogy and event chains in the canonical dump format. */
int gph_engine_last_error(gph_engine *e, int64_t *locus, int32_t *code);
/* tests only: break the event chain of population `pop` of one locus, so that the next kernel raises a fatal code for it
 * (with several ranks every rank calls it: it completes a deferred synchronizeEvents pass first, which is a collective;
 * GPH_EARG on the ranks that do not hold the locus) */
int gph_engine_debug_break_chain(gph_engine *e, int64_t global_locus, int32_t pop);
/* tests only: libgphocs_hip_chk.so is the same library compiled with -DGPH_BOUNDS -- every index the per-locus device code puts
 * into an array of a locus's LDS image, into its dynamic LDS or into its conditional arrays is compared with the array's extent.
 * *checked = 1 in such a build (0 in every product library); *where = 0, or the first violation's source line + 100000 x file
 * (read and cleared).  gph_engine_unit op 8 is the check's self-test: it reads node record `arg`. */
int gph_engine_debug_oob(gph_engine *e, int32_t *where, int32_t *checked);
/* debug / parity: canonical text dump of every local locus (same format as the oracle's) */
int gph_engine_dump_loci(gph_engine *e, const char *path, int32_t withConditionals, int32_t append);
/* debug / parity, kernel level: single calls of the per-locus functions with deterministic arguments on the current
 * chain state, every call undone -- what oracle/ref_harness.c `unit` does with the reference's own functions
 * (tests/golden/ *.unit).  out[local loci][stride] in input order.  op 0: per internal node i (stride >= 3 (n-1)):
 * tnew, lnLd, dprior of adjustGenNodeAge + computeLocusDataLikelihood(1) + considerEventMove (GPhoCS.c:2316-2381);
 * op 1: computeLocusDataLikelihood(useOld=0); op 2: rubberBand(pre) x3 of ancestral population arg + evaluation
 * (GPhoCS.c:3705-3831): delta, n0, n1, lik.  Second set (`unit2`, tests/golden/ *.unit2): op 3: executeGenSPR of node arg
 * (LocusDataLikelihood.c:931-1012; return codes 0 / 1 / 2) onto its father's, its sibling's, the root's and every fifth other
 * legal branch + computeLocusDataLikelihood(1) + revertToSaved (stride >= 1 + 5 (2n-1): calls, then target, age, code, value,
 * root); op 4: scaleAllNodeAges(1 + arg / 1000) (LocusDataLikelihood.c:895-917) + revertToSaved + full recompute: delta,
 * value; op 5: rubberBandRipple(do) + (undo) (patch.c:815-869) over every migration event's source-side event moved 0.01 %
 * up: events, two deltas; op 6: traceLineage(arg, 0) + traceLineage(arg, 1) (patch.c:886-1331) + evaluation (stride >= 13:
 * 1, res, target, father's new population, migration events removed / created, both prior deltas, father's new age, data
 * delta, the locus's generator state); op 7: rubberBandRipple(do) + (undo) over the MIG_BAND_START / MIG_BAND_END events of
 * every band, moved 30 % into the neighbouring gap (the entries UpdateTau adds with start_or_end 1 / 0, GPhoCS.c:3708-3745):
 * events, two deltas */
int gph_engine_unit(gph_engine *e, int32_t op, int32_t arg, double *out, int32_t stride);
/* timing of the last launch of a named kernel class, measured with HIP events on the
 * engine's own stream: which = 0 sweep, 1 tau_eval, 2 mix_eval, 3 init, 4 check,
 * 5 tau_finish (commit or revert, by the decision flag), 7 mix_finish, 8 sync, 9 locus-rate scan, 10 locus-rate apply,
 * 11 locus-rate prepare */
int gph_engine_last_kernel_ms(gph_engine *e, int32_t which, double *ms);
/* classes whose launches are bracketed by HIP events (bit k = class k); default all */
int gph_engine_set_timing(gph_engine *e, uint32_t class_mask);
/* host synchronisations, cross-rank exchanges and kernel launches since the engine was created; resident_mode = 1 when
 * the decisions above the loci are taken on the device (one host synchronisation per iteration) */
int gph_engine_host_stats(gph_engine *e, int64_t *syncs, int64_t *collectives, int64_t *launches, int32_t *resident_mode);
/* accumulated per class: out5 = {launches, summed ms, evaluations, algorithmic bytes, recomputed nodes} */
int gph_engine_class_stats(gph_engine *e, int32_t which, double *out5, int32_t reset);
int64_t gph_engine_num_loci(gph_engine *e);
/* parity probe: out[5n] = exp(x), log(x), sqrt(|x|), x/y, floor(x) evaluated on the device */
int gph_debug_math(const double *x, const double *y, int32_t n, double *out5n, int32_t device);
int gph_engine_hbm_bytes(gph_engine *e, double *bytes);
/* identity of this build of the library: hash of its sources and compiler flags (measurement files under profiles/
 * carry it, and bench.py drops a committed counter measurement that was taken with another build) */
const char *gph_build_id(void);
/* the compiler that built this library (its __clang_version__) and the HIP runtime / driver it is running on
 * ("runtime <hipRuntimeGetVersion>, driver <hipDriverGetVersion>"): both go into the bench line -- a library built by one
 * ROCm release may well run on another */
/* TEST BUILDS ONLY (libgphocs_hip_plain.so and the host build of tests/hostemu; every other build returns GPH_EARG):
 * decision-level transcript of the three genealogy sweeps for the selected loci (global locus indices), the engine's
 * counterpart of upstream's -DLOG_STEPS debug file (GPhoCS.c:2363-2401, 2540-2577, 2654-2718; patch.c:1451-1454).
 * Records are 8 doubles: kind, then  1 node-age proposal: node, t, tnew | 2 considerEventMove: event, source pop, target
 * pop, old age, new age, new event | 3 decision: accepted, lnacceptance | 4 migration-node proposal: node, t, tnew |
 * 5 SPR: node, father, father's population.  tests/test_logsteps.py formats them as upstream prints them. */
int gph_engine_steplog_enable(gph_engine *e, const int64_t *loci, int32_t n, int32_t cap);
int gph_engine_steplog_fetch(gph_engine *e, int32_t idx, double *out, int32_t max_records, int32_t *nrec, int32_t reset);
const char *gph_build_compiler(void);
const char *gph_runtime_version(void);

/* ------------------------------------------------------------------------------------
 * host MCMC driver: the iteration body of performMCMC (GPhoCS.c:1476-1821) above the
 * engine: general-slot RNG, priors, accept decisions of the global proposals
 * (UpdateTheta GPhoCS.c:3037, UpdateMigRates :3115, UpdateTau :3224 host part,
 * UpdateSampleAge :4006 host part, mixing :4688 host part), accumulators dataState.{logLikelihood,dataLogLikelihood}. */
typedef struct {
  const double *thetaAlpha, *thetaBeta, *thetaStart;   /* [K] */
  const double *ageAlpha, *ageBeta, *ageStart;         /* [K] (ancestral pops) */
  const double *sampleAge;                             /* [K] */
  const int32_t *updateSampleAge;                      /* [K] 1 = estimated ("age x e"), NULL = none */
  const double *mrAlpha, *mrBeta;                      /* [B] */
  double ftCoalTime, ftMigTime, ftTheta, ftMigRate, ftMixing;
  const double *ftTaus;                                /* [K] */
  int32_t seed, startMig, doMixing, samplesPerLog;
  int32_t numParameters;
  const double *printFactors;                          /* [numParameters] */
  int32_t mutRateMode;                                 /* 0 CONST, 1 VAR (UpdateLocusRate runs, GPhoCS.c:1554), 2 FIXED */
  double varRatesAlpha, ftLocusRate;                   /* locus-mut-rate VAR <alpha>, finetune-locus-rate */
} gph_mcmc_config;

int gph_mcmc_create(gph_engine *e, const gph_config *cfg, const gph_mcmc_config *mc, gph_mcmc **out);
void gph_mcmc_destroy(gph_mcmc *m);
int gph_mcmc_initialize(gph_mcmc *m, int64_t *totalCoals);
/* optional record file: one "IT <iter> <proposal> <accepted> <dataLnL %a> <logL %a>" line per
 * proposal call and one TRACE line per iteration (the unchanged trace-file row format) */
int gph_mcmc_set_record_file(gph_mcmc *m, const char *path);
int gph_mcmc_iteration(gph_mcmc *m, int32_t iteration);
int gph_mcmc_get_state(gph_mcmc *m, double *logLikelihood, double *dataLogLikelihood, double *theta,
                       double *popAge, double *migRate);
int gph_mcmc_dump_state(gph_mcmc *m, const char *path, int32_t withConditionals);
int gph_mcmc_accept_counts(gph_mcmc *m, int64_t *counts9);
/* cumulative accepted UpdateTau (ancestral) / UpdateSampleAge (current, estimated age) proposals per population [K] */
int gph_mcmc_tau_accept_counts(gph_mcmc *m, int64_t *perPop);
/* new step sizes (the find-finetunes search of performMCMC, GPhoCS.c:1896-2180, changes them between log periods);
 * taus [K] or NULL = unchanged */
int gph_mcmc_set_finetunes(gph_mcmc *m, double coalTime, double migTime, double theta, double migRate, double mixing,
                           const double *taus);
/* iterations between two checkAll() resynchronisations (= the log period, GPhoCS.c:1811-1821; the find-finetunes
 * phase uses find-finetunes-samples-per-step instead of iterations-per-log) */
int gph_mcmc_set_log_period(gph_mcmc *m, int32_t iterations);
/* finetune-locus-rate (changed by the find-finetunes search), and the running state of UpdateLocusRate:
 * cumulative accept count and dataState.rateVar */
int gph_mcmc_set_locus_rate_finetune(gph_mcmc *m, double locusRate);
int gph_mcmc_locus_rate_state(gph_mcmc *m, int64_t *accepted, double *rateVar);
/* recordParamVals (GPhoCS.c:802-849) of the last iteration: thetas, taus, migration rates, sample ages */
int gph_mcmc_param_vals(gph_mcmc *m, double *vals, int32_t n);

/* ------------------------------------------------------------------------------------
 * input front end (host only, no GPU): the reference's file formats, unchanged.
 *   gph_control_read   readControlFile + readSecondaryControlFile + checkSettings +
 *                      finalizeNumParameters              (MCMCcontrol.c:118-463, 575-1478)
 *   gph_loci_read      readSeqFile + processLocusAlignment + processHetPatterns
 *                      (AlignmentProcessor.c:468-1158, 1595-1894, 2242-2339) and readRateFile
 *                      (GPhoCS.c:491-579); the result is what gph_engine_load_loci takes
 *   gph_run_control_file   main() + the trace-file side of performMCMC (GPhoCS.c:84-238, 1232-1330,
 *                      1763-1769): same control file, same sequence file, same trace file */
typedef struct gph_control gph_control;
typedef struct gph_loci gph_loci;
typedef struct {
  const char *seqFile, *traceFile, *rateFile;
  int32_t numLoci;          /* num-loci of the control file, -1 = all loci of the sequence file */
  int32_t burnin, numSamples, sampleSkip, logsPerLine;
  int32_t mutRateMode;      /* 0 CONST, 1 VAR, 2 FIXED */
  int32_t findFinetunes, findFinetunesNumSteps, findFinetunesSamplesPerStep;
  int32_t numSampleSlots;   /* haploid leaves per locus (a diploid sample takes two) */
  double varRatesAlpha, ftLocusRate;
} gph_control_info;
int gph_control_read(const char *ctl_path, const char *secondary_ctl_path_or_null, gph_control **out);
void gph_control_free(gph_control *c);
/* fills any non-NULL output; pointers inside stay owned by (and valid as long as) the control object */
int gph_control_get(const gph_control *c, gph_config *cfg, gph_mcmc_config *mc, gph_control_info *info);
const char *gph_control_pop_name(const gph_control *c, int32_t pop);
const char *gph_control_sample_name(const gph_control *c, int32_t slot);   /* "" = second haploid of a diploid */
/* seq_path NULL = the control file's seq-file; threads <= 0 = all host threads; err receives the
 * reference's error text when the file is rejected */
int gph_loci_read(const gph_control *c, const char *seq_path, int32_t threads, gph_loci **out, char *err, int32_t errlen);
void gph_loci_free(gph_loci *l);
int gph_loci_arrays(const gph_loci *l, int64_t *L, int32_t *n, const int64_t **pattern_offsets, const uint8_t **leafcodes,
                    const uint16_t **numPhases, const int32_t **counts, const double **mutRates, const int32_t **unphased);
int gph_run_control_file(const char *ctl_path, const char *secondary_ctl_path_or_null, int32_t device, int32_t verbose);
/* the same chain over `world` processes, one per GPU: every rank reads the files, holds the contiguous block
 * rank*ceil(L/world) .. of the loci, runs the same host code on the same general RNG stream and combines the
 * reduced vectors through `allreduce` (see gph_engine_set_allreduce); rank 0 writes the trace file */
int gph_run_control_file_ranked(const char *ctl_path, const char *secondary_ctl_path_or_null, int32_t device,
                                int32_t verbose, int32_t rank, int32_t world, gph_allreduce_fn allreduce, void *user);
/* ------------------------------------------------------------------------------------
 * The reference's proposal functions, ONE CALL EACH -- the drop-in boundary at the granularity performMCMC calls them
 * (upstream src/GPhoCS.h:84-100): a host program that keeps the reference's own performMCMC (trace writer, finetune
 * search, log lines) replaces the BODIES of those functions by these calls and mirrors the handful of process-wide
 * globals the reference keeps on its main thread through gph_chain_state.  oracle/integration_binding.c is that
 * replacement written out, linked with the reference's own objects and run (tests/test_boundary_run.py).
 *   reference function                                  call
 *   UpdateGB_InternalNode + _MigrationNode + _MigSPR    gph_mcmc_update_gb        (GPhoCS.c:2287, 2439, 2598; one fused launch)
 *   UpdateLocusRate        GPhoCS.c:4598                gph_mcmc_update_locus_rate
 *   UpdateTheta            GPhoCS.c:3037                gph_mcmc_update_theta
 *   UpdateMigRates         GPhoCS.c:3115                gph_mcmc_update_mig_rates
 *   UpdateTau              GPhoCS.c:3224                gph_mcmc_update_tau        (accepted[ancestral pop], as upstream)
 *   UpdateSampleAge        GPhoCS.c:4006                gph_mcmc_update_sample_age (accepted[current pop])
 *   mixing                 GPhoCS.c:4688                gph_mcmc_mixing
 *   synchronizeEvents loop GPhoCS.c:1705-1714           gph_mcmc_synchronize_events(refresh = 0)
 *   sampleMigRates' genLogLikelihood loop :1749-1757    gph_mcmc_synchronize_events(refresh = 1), after set_chain
 *   checkAll               patch.c:2745                 gph_mcmc_check_all
 * Conventions as upstream: a step size <= 0 makes the call return 0 accepted without drawing anything; errors are a
 * non-zero status (upstream prints "Fatal Error NNNN" and exits). */
#define GPH_ABI_MAXK 40     /* 2 * NSPECIES - 1 = 39 populations upstream (patch.h:19) */
#define GPH_ABI_MAXB 100    /* MAX_MIG_BANDS (patch.h:17) */
typedef struct {
  double theta[GPH_ABI_MAXK], popAge[GPH_ABI_MAXK], sampleAge[GPH_ABI_MAXK];   /* Population.{theta,age,sampleAge}, PopulationTree.h:60-101 */
  double migRate[GPH_ABI_MAXB], bandStart[GPH_ABI_MAXB], bandEnd[GPH_ABI_MAXB]; /* MigrationBand.{migRate,startTime,endTime} */
  uint32_t rng[3];                                       /* the general slot of RndCtx (utils.h:34): rndu_x, rndu_y, rndu_z */
  double logLikelihood, dataLogLikelihood, rateVar;      /* dataState, GPhoCS.h:35-50 */
  double coal_stats[GPH_ABI_MAXK], num_coals[GPH_ABI_MAXK], mig_stats[GPH_ABI_MAXB], num_migs[GPH_ABI_MAXB];   /* genetree_stats_total, patch.h:121 */
  int64_t rubberband_mig_conflicts;                      /* misc_stats, patch.h */
} gph_chain_state;
int gph_mcmc_get_chain(gph_mcmc *m, gph_chain_state *out);
int gph_mcmc_set_chain(gph_mcmc *m, const gph_chain_state *in);
int gph_mcmc_update_gb(gph_mcmc *m, int32_t iteration, double ftCoalTime, double ftMigTime, int64_t accepted[3], int64_t *total_mig_nodes);
int gph_mcmc_update_locus_rate(gph_mcmc *m, int32_t iteration, double finetune, int64_t *accepted);
int gph_mcmc_update_theta(gph_mcmc *m, int32_t iteration, double finetune, int64_t *accepted);
int gph_mcmc_update_mig_rates(gph_mcmc *m, int32_t iteration, double finetune, int64_t *accepted);
int gph_mcmc_update_tau(gph_mcmc *m, int32_t iteration, const double *finetunes, int32_t *accepted);
int gph_mcmc_update_sample_age(gph_mcmc *m, int32_t iteration, const double *finetunes, int32_t *accepted);
int gph_mcmc_mixing(gph_mcmc *m, int32_t iteration, double finetune, int64_t *accepted);
int gph_mcmc_synchronize_events(gph_mcmc *m, int32_t iteration, int32_t refresh);
int gph_mcmc_check_all(gph_mcmc *m, int32_t iteration, int32_t *ok);
/* initializeMCMC's per-locus loop (GPhoCS.c:1197-1214) for a caller that has sampled the population parameters itself
 * (samplePopParameters) and handed them over with gph_mcmc_set_chain: genealogies, event chains, statistics,
 * likelihoods; the chain state then holds dataLogLikelihood / logLikelihood as :1216-1224 leave them */
int gph_mcmc_initialize_genealogies(gph_mcmc *m);

/* the same with a native communicator (RCCL or shared memory): what `G-PhoCS-hip -g N <control-file>` runs in each of
 * its N child processes */
int gph_run_control_file_comm(const char *ctl_path, const char *secondary_ctl_path_or_null, int32_t device,
                              int32_t verbose, gph_comm *comm);

/* ------------------------------------------------------------------------------------
 * post-run summary of a trace file (host only): block means per column, the output of the reference's
 * stand-alone `readTrace` tool (src/readTrace.c:41-291: `-b` block size, default the whole file = -1;
 * `-d` samples to discard).  The text that tool prints goes to out (NUL-terminated, *out_len = its
 * length; out may be NULL to ask for the length), its error text to err.  Returns 0, the tool's exit
 * code on error, or 2 when out_cap was too small. */
int gph_read_trace(const char *trace_file, int block_size, int discard, char *out, size_t out_cap, size_t *out_len,
                   char *err, size_t err_cap);

#ifdef __cplusplus
}
#endif
#endif
