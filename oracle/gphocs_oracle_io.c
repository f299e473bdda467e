/*
 * gphocs_oracle_io.c -- TEST INFRASTRUCTURE ONLY (see gphocs_oracle.h).
 * Pack loader (format written by oracle/ref_harness.c `pack`) and the
 * canonical state dump (same text format as ref_harness.c dump_state).
 */
#include <stdlib.h>
#include <string.h>
#include "gphocs_oracle.h"
#include "gphocs_oracle_int.h"

static void alloc_locus(go_model *m, go_locus *q, int P)
{
  int N = 2 * m->n - 1, i;
  memset(q, 0, sizeof *q);
  q->P = P;
  q->leafcode = (unsigned char *)calloc((size_t)P * m->n + 1, 1);
  q->numPhases = (int *)calloc(P + 1, sizeof(int));
  q->count = (int *)calloc(P + 1, sizeof(int));
  q->father = (int *)calloc(N, sizeof(int));
  q->left = (int *)calloc(N, sizeof(int));
  q->right = (int *)calloc(N, sizeof(int));
  q->age = (double *)calloc(N, sizeof(double));
  q->nodePop = (int *)calloc(N, sizeof(int));
  q->nodeEvent = (int *)calloc(N, sizeof(int));
  q->cond[0] = (double *)calloc((size_t)N * P * 4 + 4, sizeof(double));
  q->cond[1] = (double *)calloc((size_t)N * P * 4 + 4, sizeof(double));
  q->condbit = (unsigned char *)calloc(N, 1);
  q->dirty = (unsigned char *)calloc(N, 1);
  q->changedNodes = (int *)calloc(2 * N, sizeof(int));
  q->changedCond = (int *)calloc(2 * N, sizeof(int));
  q->sv_father = (int *)calloc(N, sizeof(int));
  q->sv_left = (int *)calloc(N, sizeof(int));
  q->sv_right = (int *)calloc(N, sizeof(int));
  q->sv_age = (double *)calloc(N, sizeof(double));
  q->sv_root = -1;
  /* pool size, patch.c:92 */
  q->E = 2 * m->n + 4 * GO_MAX_MIGS + 3 * m->B + m->K + 10;
  q->ev_type = (int *)calloc(q->E, sizeof(int));
  q->ev_node = (int *)calloc(q->E, sizeof(int));
  q->ev_next = (int *)calloc(q->E, sizeof(int));
  q->ev_prev = (int *)calloc(q->E, sizeof(int));
  q->ev_nlin = (int *)calloc(q->E, sizeof(int));
  q->ev_time = (double *)calloc(q->E, sizeof(double));
  q->delta[0].changed_events = (int *)calloc(q->E, sizeof(int));
  q->delta[1].changed_events = (int *)calloc(q->E, sizeof(int));
  for (i = 0; i < 2; i++) { q->delta[i].original_event = -1; q->delta[i].updated_event = -1; }
  for (i = 0; i < GO_MAX_MIGS; i++) {
    q->mig[i].band = -1; q->mig[i].branch = -1; q->mig[i].target_pop = -1; q->mig[i].source_pop = -1;
    q->mig[i].target_event = -1; q->mig[i].source_event = -1; q->mig[i].age = 0;
  }
  q->mutRate = 1.0;
}

static void free_locus(go_locus *q)
{
  free(q->leafcode); free(q->numPhases); free(q->count); free(q->father); free(q->left);
  free(q->right); free(q->age); free(q->nodePop); free(q->nodeEvent); free(q->cond[0]);
  free(q->cond[1]); free(q->condbit); free(q->dirty); free(q->changedNodes); free(q->changedCond);
  free(q->sv_father); free(q->sv_left); free(q->sv_right); free(q->sv_age); free(q->ev_type);
  free(q->ev_node); free(q->ev_next); free(q->ev_prev); free(q->ev_nlin); free(q->ev_time);
  free(q->delta[0].changed_events); free(q->delta[1].changed_events);
}

/* leaf conditionals, computeLeafConditionals LocusDataLikelihood.c:1321-1386 */
static void set_leaf_conditionals(go_model *m, go_locus *q)
{
  int p, leaf, a, bit;
  for (bit = 0; bit < 2; bit++)
    for (leaf = 0; leaf < m->n; leaf++)
      for (p = 0; p < q->P; p++) {
        double *c = q->cond[bit] + ((size_t)leaf * q->P + p) * 4;
        int code = q->leafcode[(size_t)p * m->n + leaf];
        for (a = 0; a < 4; a++) c[a] = (code == 4 || code == a) ? 1.0 : 0.0;
      }
}

static int code_of(int ch)
{
  switch (ch) { case 'T': return 0; case 'C': return 1; case 'A': return 2; case 'G': return 3; case 'N': return 4; }
  return -1;
}

go_state *go_load_pack(const char *path)
{
  FILE *f = fopen(path, "r");
  char key[128], buf[8192];
  go_state *s;
  go_model *m;
  int i, ver, pop, b, g, P, a, d;
  if (!f) { perror(path); return NULL; }
  s = (go_state *)calloc(1, sizeof *s);
  m = &s->m;
  if (fscanf(f, "%127s %d", key, &ver) != 2 || strcmp(key, "GPHOCS-PACK") || ver != 1) goto bad;
  if (fscanf(f, " numLoci %d numSamples %d numCurPops %d numPops %d numMigBands %d rootPop %d",
             &s->L, &m->n, &m->Kc, &m->K, &m->B, &m->rootPop) != 6) goto bad;
  if (m->K > GO_MAXK || m->B > GO_MAXB || m->n > 200) goto bad;
  if (fscanf(f, " %127s", key) != 1 || strcmp(key, "samplesPerPop")) goto bad;
  for (pop = 0; pop < m->Kc; pop++) if (fscanf(f, "%d", &m->samplesPerPop[pop]) != 1) goto bad;
  for (pop = 0; pop < m->K; pop++) {
    int id;
    if (fscanf(f, " pop %d %63s %d %d %d", &id, m->popName[pop], &m->popFather[pop], &m->popSon0[pop],
               &m->popSon1[pop]) != 5 || id != pop) goto bad;
    if (fscanf(f, "%s", buf) != 1) goto bad; m->sampleAge[pop] = strtod(buf, NULL);
    if (fscanf(f, "%d", &m->updateSampleAge[pop]) != 1) goto bad;
    if (fscanf(f, "%s", buf) != 1) goto bad; m->thetaAlpha[pop] = strtod(buf, NULL);
    if (fscanf(f, "%s", buf) != 1) goto bad; m->thetaBeta[pop] = strtod(buf, NULL);
    if (fscanf(f, "%s", buf) != 1) goto bad; m->thetaStart[pop] = strtod(buf, NULL);
    if (fscanf(f, "%s", buf) != 1) goto bad; m->ageAlpha[pop] = strtod(buf, NULL);
    if (fscanf(f, "%s", buf) != 1) goto bad; m->ageBeta[pop] = strtod(buf, NULL);
    if (fscanf(f, "%s", buf) != 1) goto bad; m->ageStart[pop] = strtod(buf, NULL);
    m->popAge[pop] = 0.0;
    m->theta[pop] = 0.0;
  }
  for (b = 0; b < m->B; b++) {
    int id;
    if (fscanf(f, " band %d %d %d", &id, &m->bandSrc[b], &m->bandTgt[b]) != 3 || id != b) goto bad;
    if (fscanf(f, "%s", buf) != 1) goto bad; m->mrAlpha[b] = strtod(buf, NULL);
    if (fscanf(f, "%s", buf) != 1) goto bad; m->mrBeta[b] = strtod(buf, NULL);
  }
  if (fscanf(f, " mcmc %d %d %d %d %d %d %d %d", &m->seed, &m->burnin, &m->numSamples, &m->sampleSkip,
             &m->startMig, &m->doMixing, &m->samplesPerLog, &m->mutRateMode) != 8) goto bad;
  if (fscanf(f, " %127s", key) != 1 || strcmp(key, "finetunes")) goto bad;
  {
    double *ft[5] = {&m->ftCoalTime, &m->ftMigTime, &m->ftTheta, &m->ftMigRate, &m->ftMixing};
    for (i = 0; i < 5; i++) { if (fscanf(f, "%s", buf) != 1) goto bad; *ft[i] = strtod(buf, NULL); }
    for (pop = 0; pop < m->K; pop++) { if (fscanf(f, "%s", buf) != 1) goto bad; m->ftTaus[pop] = strtod(buf, NULL); }
  }
  m->varRatesAlpha = 1.0;
  m->ftLocusRate = -1.0;
  if (m->mutRateMode == 1) {   /* only packs of VAR-rate control files carry this line */
    if (fscanf(f, " %127s", key) != 1 || strcmp(key, "locusrate")) goto bad;
    if (fscanf(f, "%s", buf) != 1) goto bad; m->varRatesAlpha = strtod(buf, NULL);
    if (fscanf(f, "%s", buf) != 1) goto bad; m->ftLocusRate = strtod(buf, NULL);
  }
  if (fscanf(f, " printFactors %d", &m->numParameters) != 1) goto bad;
  for (i = 0; i < m->numParameters; i++) { if (fscanf(f, "%s", buf) != 1) goto bad; m->printFactors[i] = strtod(buf, NULL); }
  /* isAncestralTo (self-inclusive), MCMCcontrol.c:851, 977, 1015-1024 */
  for (a = 0; a < m->K; a++)
    for (d = 0; d < m->K; d++) {
      int x = d;
      m->isAnc[a][d] = 0;
      while (x >= 0) { if (x == a) { m->isAnc[a][d] = 1; break; } x = m->popFather[x]; }
    }
  s->loc = (go_locus *)calloc(s->L, sizeof(go_locus));
  for (g = 0; g < s->L; g++) {
    int id, p;
    go_locus *q = &s->loc[g];
    if (fscanf(f, " locus %d %d %s", &id, &P, buf) != 3 || id != g) goto bad;
    alloc_locus(m, q, P);
    q->mutRate = strtod(buf, NULL);
    for (p = 0; p < P; p++) {
      if (fscanf(f, "%s %d %d", buf, &q->numPhases[p], &q->count[p]) != 3) goto bad;
      if ((int)strlen(buf) != m->n) goto bad;
      for (i = 0; i < m->n; i++) {
        int c = code_of(buf[i]);
        if (c < 0) goto bad;
        q->leafcode[(size_t)p * m->n + i] = (unsigned char)c;
      }
    }
    set_leaf_conditionals(m, q);
  }
  fclose(f);
  go_seed(s, (unsigned int)m->seed);
  return s;
bad:
  fprintf(stderr, "oracle: malformed pack %s\n", path);
  fclose(f);
  return NULL;
}

void go_free(go_state *s)
{
  int g;
  if (!s) return;
  for (g = 0; g < s->L; g++) free_locus(&s->loc[g]);
  free(s->loc);
  free(s->paramVals);
  free(s);
}

void go_dump_state(go_state *s, FILE *f, int withCond)
{
  go_model *m = &s->m;
  int g, i, pop, b, ev, N = 2 * m->n - 1;
  fprintf(f, "STATE %d\n", s->L);
  fprintf(f, "MODEL");
  for (pop = 0; pop < m->K; pop++) fprintf(f, " %a %a %a", m->theta[pop], m->popAge[pop], m->sampleAge[pop]);
  for (b = 0; b < m->B; b++) fprintf(f, " %a %a %a", m->migRate[b], m->bandStart[b], m->bandEnd[b]);
  fprintf(f, "\n");
  fprintf(f, "GLOBAL %a %a %u %u %u\n", s->logLikelihood, s->dataLogLikelihood, s->gx, s->gy, s->gz);
  if (m->mutRateMode == 1) fprintf(f, "RATEVAR %a\n", s->rateVar);
  fprintf(f, "TOTALS");
  for (pop = 0; pop < m->K; pop++) fprintf(f, " %a %d", s->tot_coal_stats[pop], s->tot_num_coals[pop]);
  for (b = 0; b < m->B; b++) fprintf(f, " %a %d", s->tot_mig_stats[b], s->tot_num_migs[b]);
  fprintf(f, "\n");
  /* GPH_DUMP_STRIDE=k: every k-th locus only (full-size parity runs: 100 000 loci would be a 300-MB dump) */
  const int dstride = getenv("GPH_DUMP_STRIDE") && atoi(getenv("GPH_DUMP_STRIDE")) > 0 ? atoi(getenv("GPH_DUMP_STRIDE")) : 1;
  for (g = 0; g < s->L; g += dstride) {
    go_locus *q = &s->loc[g];
    fprintf(f, "LOCUS %d root %d dataLnL %a genLnL %a rng %u %u %u\n", g, q->root, q->dataLnL, q->genLnL,
            q->rx, q->ry, q->rz);
    if (m->mutRateMode == 1) fprintf(f, "R %a\n", q->mutRate);
    for (i = 0; i < N; i++)
      fprintf(f, "N %d %d %d %d %a %d %d\n", i, q->father[i], q->left[i], q->right[i], q->age[i],
              q->nodePop[i], i < m->n ? -1 : q->nodeEvent[i]);
    for (pop = 0; pop < m->K; pop++) {
      fprintf(f, "C %d", pop);
      for (ev = q->first_event[pop]; ev >= 0; ev = q->ev_next[ev])
        fprintf(f, " %d:%d:%d:%d:%a", ev, q->ev_type[ev], q->ev_node[ev], q->ev_nlin[ev], q->ev_time[ev]);
      fprintf(f, "\n");
    }
    fprintf(f, "S");
    for (pop = 0; pop < m->K; pop++) fprintf(f, " %a %d", q->coal_stats[pop], q->num_coals[pop]);
    for (b = 0; b < m->B; b++) fprintf(f, " %a %d", q->mig_stats[b], q->num_migs_band[b]);
    fprintf(f, "\n");
    fprintf(f, "M %d", q->num_migs);
    for (i = 0; i < q->num_migs; i++) {
      int mg = q->living[i];
      go_mignode *mn = &q->mig[mg];
      fprintf(f, " %d:%d:%d:%d:%d:%d:%d:%a", mg, mn->branch, mn->band, mn->source_pop, mn->target_pop,
              mn->source_event, mn->target_event, mn->age);
    }
    fprintf(f, "\n");
    if (withCond) {
      int p, a;
      for (i = m->n; i < N; i++) {
        double *c = q->cond[q->condbit[i]] + (size_t)i * q->P * 4;
        fprintf(f, "K %d", i);
        for (p = 0; p < q->P; p++) for (a = 0; a < 4; a++) fprintf(f, " %a", c[4 * p + a]);
        fprintf(f, "\n");
      }
    }
  }
  fprintf(f, "ENDSTATE\n");
}
