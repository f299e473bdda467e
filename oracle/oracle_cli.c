/*
 * oracle_cli.c -- TEST INFRASTRUCTURE ONLY.  Command-line front end of the
 * CPU restatement, producing the same text records as oracle/ref_harness.c so
 * that outputs can be compared with `diff`:
 *   gphocs_oracle run  <pack> <iters> <out.trace> [statefile] [state_iter] [withCond]
 *   gphocs_oracle time <pack> <iters> [warm]
 *   gphocs_oracle rng  <seed> <count>
 *   gphocs_oracle reflect
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include "gphocs_oracle.h"

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv)
{
  if (argc >= 5 && !strcmp(argv[1], "run")) {
    go_state *s = go_load_pack(argv[2]);
    int iters = atoi(argv[3]), it, tc;
    FILE *tf = fopen(argv[4], "w");
    const char *statefile = argc > 5 ? argv[5] : NULL;
    int stateIter = argc > 6 ? atoi(argv[6]) : iters - 1;
    int withCond = argc > 7 ? atoi(argv[7]) : 0;
    if (!s || !tf) return 2;
    tc = go_initialize_mcmc(s);
    fprintf(tf, "IT %d %s %d %a %a\n", -1, "INIT", tc, s->dataLogLikelihood, s->logLikelihood);
    if (statefile && stateIter < 0) { FILE *sf = fopen(statefile, "w"); go_dump_state(s, sf, withCond); fclose(sf); }
    for (it = 0; it < iters; it++) {
      if (go_iteration(s, it, tf) != 0) return 3;
      go_trace_line(s, it, tf);
      if (statefile && stateIter == it) { FILE *sf = fopen(statefile, "w"); go_dump_state(s, sf, withCond); fclose(sf); }
    }
    fclose(tf);
    go_free(s);
    return 0;
  }
  if (argc >= 4 && !strcmp(argv[1], "time")) {
    go_state *s = go_load_pack(argv[2]);
    int iters = atoi(argv[3]), warm = argc > 4 ? atoi(argv[4]) : 2, it;
    double t0, t1;
    long e0;
    if (!s) return 2;
    go_initialize_mcmc(s);
    for (it = 0; it < warm; it++) go_iteration(s, it, NULL);
    e0 = s->evals;
    t0 = now_s();
    for (; it < warm + iters; it++) go_iteration(s, it, NULL);
    t1 = now_s();
    printf("{\"loci\": %d, \"iters\": %d, \"seconds\": %.6f, \"iters_per_s\": %.6f, \"evals\": %ld, "
           "\"evals_per_s\": %.3f, \"dataLnL\": %.6f}\n", s->L, iters, t1 - t0, iters / (t1 - t0),
           s->evals - e0, (s->evals - e0) / (t1 - t0), s->dataLogLikelihood);
    return 0;
  }
  if (argc >= 4 && !strcmp(argv[1], "rng")) {
    unsigned int seed = (unsigned int)strtoul(argv[2], NULL, 10);
    int count = atoi(argv[3]), i;
    unsigned int v = 170u * (seed % 178u) + 137u;
    unsigned int x0 = 11, y0 = 23, z0 = v, x1 = 11, y1 = 23, z1 = v;
    for (i = 0; i < count; i++) printf("U %a\n", go_rndu(&x0, &y0, &z0));
    for (i = 0; i < count; i++) {
      printf("N8 %a\n", go_rnd2normal8(&x1, &y1, &z1));
      printf("E %a\n", -(0.37) * log(go_rndu(&x1, &y1, &z1)));
      printf("NN %a\n", go_rndnormal(&x1, &y1, &z1));
    }
    printf("X %u %u %u %u %u %u\n", x0, y0, z0, x1, y1, z1);
    return 0;
  }
  if (argc >= 2 && !strcmp(argv[1], "reflect")) {
    static const double as[] = {0.0, 1e-5, 0.25, -3.0};
    static const double ws[] = {1e-10, 2.5e-9, 1e-6, 0.01, 1.0, 7.5};
    int ia, iw, k;
    for (ia = 0; ia < 4; ia++)
      for (iw = 0; iw < 6; iw++)
        for (k = -40; k <= 40; k++) {
        if (k == 0) continue; /* x == a with a 0.5e-9-wide window ping-pongs forever in the reference */
          double a = as[ia], b = a + ws[iw];
          double x = a + ws[iw] * (0.37 * k + 0.011 * k * k * (k % 3 - 1));
          printf("R %a %a %a %a\n", x, a, b, go_reflect(x, a, b));
        }
    return 0;
  }
  fprintf(stderr, "usage: gphocs_oracle run|time|rng|reflect ...\n");
  return 1;
}
