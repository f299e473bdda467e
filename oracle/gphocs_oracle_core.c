/*
 * gphocs_oracle_core.c -- TEST INFRASTRUCTURE ONLY (see gphocs_oracle.h).
 *
 * Per-locus engines of the CPU restatement: RNG, reflect, JC69 pruning with
 * dirty-path recomputation, event chains, sufficient statistics, rubber band,
 * lineage tracing.  Citations are file:line under /root/reference/src.
 *
 * Arithmetic contract: every floating-point expression keeps the reference's
 * operand order and association; compile with -ffp-contract=off.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "gphocs_oracle.h"
#include "gphocs_oracle_int.h"

void go_fatal(go_state *s, int code, const char *what)
{
  fprintf(stderr, "oracle: fatal %d (%s)\n", code, what);
  s->error = code;
  fflush(NULL);        /* the records written so far are evidence too: the reference's exit() flushes its files */
  abort();
}

/* ------------------------------------------------------------------ */
/* RNG: utils.c:498-513 (unsigned 32-bit Wichmann-Hill without the sign fix-up) */
double go_rndu(unsigned int *x, unsigned int *y, unsigned int *z)
{
  double r;
  *x = 171u * (*x % 177u) - 2u * (*x / 177u);
  *y = 172u * (*y % 176u) - 35u * (*y / 176u);
  *z = 170u * (*z % 178u) - 63u * (*z / 178u);
  r = *x / 30269.0 + *y / 30307.0 + *z / 30323.0;
  r = (r - (int)r);
  return r;
}

/* utils.c:459-472 */
double go_rndnormal(unsigned int *x, unsigned int *y, unsigned int *z)
{
  double u, v, s;
  for (;;) {
    u = 2 * go_rndu(x, y, z) - 1;
    v = 2 * go_rndu(x, y, z) - 1;
    s = u * u + v * v;
    if (s > 0 && s < 1) break;
  }
  s = sqrt(-2. * log(s) / s);
  return u * s;
}

/* utils.c:482-488 with the kernel constants of utils.c:427-431 */
double go_rnd2normal8(unsigned int *x, unsigned int *y, unsigned int *z)
{
  const double m2s2 = 8.;
  double m2N = sqrt(m2s2 / (m2s2 + 1.));
  double s2N = sqrt(1. / (m2s2 + 1.));
  double zz = m2N + go_rndnormal(x, y, z) * s2N;
  zz = go_rndu(x, y, z) < 0.5 ? zz : -zz;
  return zz;
}

/* utils.c:421-426: every slot starts from the same triple */
void go_seed(go_state *s, unsigned int seed)
{
  int g;
  unsigned int v = 170u * (seed % 178u) + 137u;
  for (g = 0; g < s->L; g++) { s->loc[g].rx = 11; s->loc[g].ry = 23; s->loc[g].rz = v; }
  s->gx = 11; s->gy = 23; s->gz = v;
}

/* utils.c:333-398 */
double go_reflect(double x, double a, double b)
{
  const double slack = 0.000000001;
  double xnew, double_interval;
  a += slack;
  b -= slack;
  if (b <= a) return (a + b) / 2.;
  if (x < b && x > a) return x;
  xnew = x;
  if (xnew <= a) xnew = 2. * a - xnew;
  double_interval = 2. * (b - a);
  xnew = xnew - double_interval * floor((xnew - a) / double_interval);
  if (xnew >= b) xnew = 2. * b - xnew;
  while (xnew <= a || xnew >= b) {
    if (xnew >= b) xnew = 2. * b - xnew;
    else xnew = 2 * a - xnew;
  }
  return xnew;
}

/* ------------------------------------------------------------------ */
/* population tree: PopulationTree.c:439-491 */
int go_update_band_times(go_model *m, int b)
{
  int res = 0, src = m->bandSrc[b], tgt = m->bandTgt[b];
  double t = m->popAge[src] > m->popAge[tgt] ? m->popAge[src] : m->popAge[tgt];
  double fs, ft;
  if (t != m->bandStart[b]) { m->bandStart[b] = t; res = 1; }
  fs = m->popAge[m->popFather[src]];
  ft = m->popAge[m->popFather[tgt]];
  t = fs < ft ? fs : ft;
  if (t != m->bandEnd[b]) { m->bandEnd[b] = t; res = 1; }
  return res;
}

void go_compute_band_times(go_model *m)
{
  int b;
  for (b = 0; b < m->B; b++) {
    go_update_band_times(m, b);
    if (m->bandStart[b] >= m->bandEnd[b])
      m->bandStart[b] = m->bandEnd[b] = m->popAge[m->bandTgt[b]];
  }
}

/* ------------------------------------------------------------------ */
/* data likelihood: save / revert with value semantics                 */

/* copyNodeConditionals, LocusDataLikelihood.c:1889-1906 */
int go_lik_mark_cond(go_locus *q, int node)
{
  if (q->P <= 0 || q->dirty[node]) return 1;
  q->changedCond[q->numChangedCond++] = node;
  q->dirty[node] = 1;
  q->condbit[node] ^= 1;
  return 0;
}

/* copyNodeToSaved, LocusDataLikelihood.c:1864-1876 */
void go_lik_save_node(go_locus *q, int node, int recalc)
{
  if (recalc) go_lik_mark_cond(q, node);
  q->changedNodes[q->numChangedNodes++] = node;
  q->sv_age[node] = q->age[node];
  q->sv_father[node] = q->father[node];
  q->sv_left[node] = q->left[node];
  q->sv_right[node] = q->right[node];
}

/* adjustGenNodeAge, LocusDataLikelihood.c:875-882 */
void go_lik_adjust_age(go_locus *q, int node, double age)
{
  go_lik_save_node(q, node, 1);
  q->age[node] = age;
}

/* resetSaved, LocusDataLikelihood.c:852-864 */
void go_lik_reset_saved(go_state *s, go_locus *q)
{
  int N = 2 * s->m.n - 1;
  q->copyAll = 0;
  q->numChangedNodes = 0;
  q->numChangedCond = 0;
  q->sv_root = -1;
  q->sv_dataLnL = q->dataLnL;
  memset(q->dirty, 0, N);
}

/* revertToSaved, LocusDataLikelihood.c:768-841 */
void go_lik_revert(go_state *s, go_locus *q)
{
  int i, node, N = 2 * s->m.n - 1;
  q->dataLnL = q->sv_dataLnL;
  if (q->sv_root >= 0) { q->root = q->sv_root; q->sv_root = -1; }
  if (q->copyAll) {
    /* every node record was saved and every conditional array switched */
    for (node = 0; node < N; node++) {
      q->age[node] = q->sv_age[node];
      q->father[node] = q->sv_father[node];
      q->left[node] = q->sv_left[node];
      q->right[node] = q->sv_right[node];
      if (q->dirty[node]) q->condbit[node] ^= 1;
    }
    go_lik_reset_saved(s, q);
    return;
  }
  if (q->numChangedCond == 0 && q->numChangedNodes == 0) return;
  for (i = 0; i < q->numChangedNodes; i++) {
    node = q->changedNodes[i];
    q->age[node] = q->sv_age[node];
    q->father[node] = q->sv_father[node];
    q->left[node] = q->sv_left[node];
    q->right[node] = q->sv_right[node];
    if (q->dirty[node]) { q->condbit[node] ^= 1; q->dirty[node] = 0; }
  }
  for (i = 0; i < q->numChangedCond; i++) {
    node = q->changedCond[i];
    if (q->dirty[node]) { q->condbit[node] ^= 1; q->dirty[node] = 0; }
  }
  q->numChangedNodes = 0;
  q->numChangedCond = 0;
}

/* computeEdgeConditionalJC, LocusDataLikelihood.c:1831-1848 */
static double edge_prob(double len)
{
  if (len < 1e-100) return 0.0;
  return ((1 - exp(-4 * len / 3.0)) / 4.0);
}

/* computeSubtreeConditionals_new, LocusDataLikelihood.c:1650-1673 */
static void apply_child(const double *son, double *par, double p, double qq)
{
  int a;
  double S = 0.0, Sp;
  for (a = 0; a < 4; a++) S += son[a];
  if (S >= 4) return;
  Sp = S * p;
  for (a = 0; a < 4; a++) par[a] *= (Sp + son[a] * qq);
}

static double *cur_cond(go_locus *q, int node)
{
  return q->cond[q->condbit[node]] + (size_t)node * q->P * 4;
}

/* computeConditionalJC_new, LocusDataLikelihood.c:1559-1636 */
static int prune_rec(go_state *s, go_locus *q, int node, int override)
{
  int res, p, a, l, r;
  double pl, ql, pr, qr, *pc, *lc, *rc;
  if (node < s->m.n) return q->dirty[node] ? 100 : 0;
  l = q->left[node];
  r = q->right[node];
  res = prune_rec(s, q, l, override);
  res = prune_rec(s, q, r, override) + res;
  if (!override && !res && !q->dirty[node]) return 0;
  if (!override) go_lik_mark_cond(q, node);
  pl = edge_prob(q->mutRate * (q->age[node] - q->age[l]));
  ql = 1 - 4.0 * pl;
  pr = edge_prob(q->mutRate * (q->age[node] - q->age[r]));
  qr = 1 - 4.0 * pr;
  pc = cur_cond(q, node);
  lc = cur_cond(q, l);
  rc = cur_cond(q, r);
  for (p = 0; p < q->P; p++) {
    for (a = 0; a < 4; a++) pc[4 * p + a] = 1.0;
    apply_child(lc + 4 * p, pc + 4 * p, pl, ql);
    apply_child(rc + 4 * p, pc + 4 * p, pr, qr);
  }
  s->evalNodes++;
  return 1;
}

/* computeLocusDataLikelihood, LocusDataLikelihood.c:426-483 */
double go_lik_compute(go_state *s, go_locus *q, int useOld)
{
  int node, res, patt, pattId = 0, c, nc, U = 0;
  long nodes0 = s->evalNodes;
  double prob, *rc;
  if (q->P == 0) return 0.0;
  if (!useOld)
    for (node = s->m.n; node < 2 * s->m.n - 1; node++) go_lik_mark_cond(q, node);
  q->sv_dataLnL = q->dataLnL;
  res = prune_rec(s, q, q->root, !useOld);
  if (useOld) s->evals++;
  if (!res) return q->dataLnL;
  q->dataLnL = 0.0;
  rc = cur_cond(q, q->root);
  for (patt = 0; patt < q->P; patt += q->numPhases[pattId]) {
    pattId = patt;
    prob = 0.0;
    nc = 4 * q->numPhases[pattId];
    for (c = 0; c < nc; c++) prob += rc[pattId * 4 + c];
    q->dataLnL += log(prob / nc) * q->count[pattId];
    U++;
  }
  if (useOld)
    s->evalBytes += 96L * (s->evalNodes - nodes0) * q->P + 20L * (2 * s->m.n - 1) + 8L * U + 8;
  return q->dataLnL;
}

/* scaleAllNodeAges, LocusDataLikelihood.c:895-917 */
double go_lik_scale_ages(go_state *s, go_locus *q, double factor)
{
  int node, N = 2 * s->m.n - 1;
  double old = q->dataLnL;
  q->copyAll = 1;
  for (node = 0; node < N; node++) go_lik_adjust_age(q, node, factor * q->age[node]);
  go_lik_compute(s, q, 1);
  return q->dataLnL - old;
}

/* executeGenSPR, LocusDataLikelihood.c:931-1012 */
int go_lik_spr(go_locus *q, int subtreeRoot, int target, double age)
{
  int targetFather = q->father[target];
  int father = q->father[subtreeRoot];
  int grandpa = q->father[father];
  int sibling = q->left[father] + q->right[father] - subtreeRoot;
  go_lik_adjust_age(q, father, age);
  if (target == sibling || target == father) return 0;
  go_lik_save_node(q, sibling, 0);
  q->father[sibling] = grandpa;
  if (grandpa >= 0) {
    go_lik_save_node(q, grandpa, 1);
    if (q->left[grandpa] == father) q->left[grandpa] = sibling;
    else q->right[grandpa] = sibling;
  }
  q->father[father] = targetFather;
  q->left[father] = subtreeRoot;
  q->right[father] = target;
  if (target != grandpa) go_lik_save_node(q, target, 0);
  q->father[target] = father;
  if (targetFather < 0) {
    q->sv_root = target;
    q->root = father;
    return 1;
  }
  if (targetFather == sibling) go_lik_mark_cond(q, targetFather);
  else if (targetFather != grandpa) go_lik_save_node(q, targetFather, 1);
  if (q->left[targetFather] == target) q->left[targetFather] = father;
  else q->right[targetFather] = father;
  if (grandpa < 0) {
    q->sv_root = father;
    q->root = sibling;
    return 2;
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* migration-node lookups: patch.c:374-414 */
int go_find_last_mig(go_locus *q, int node, double age)
{
  int i, mig, last = -1;
  for (i = 0; i < q->num_migs; i++) {
    mig = q->living[i];
    if (q->mig[mig].branch != node) continue;
    if ((age < 0 || q->mig[mig].age < age) && (last < 0 || q->mig[mig].age > q->mig[last].age))
      last = mig;
  }
  return last;
}

int go_find_first_mig(go_locus *q, int node, double age)
{
  int i, mig, first = -1;
  for (i = 0; i < q->num_migs; i++) {
    mig = q->living[i];
    if (q->mig[mig].branch != node) continue;
    if (q->mig[mig].age > age && (first < 0 || q->mig[mig].age < q->mig[first].age)) first = mig;
  }
  return first;
}

/* getEdgesForTimePop, patch.c:526-571 */
int go_edges_for_time_pop(go_state *s, go_locus *q, double time, int pop, int exc, int *out)
{
  go_model *m = &s->m;
  int node, mig, pop1, num = 0, f, N = 2 * m->n - 1;
  if (m->popAge[pop] > time + 0.0000001) return 0;
  for (node = 0; node < N; node++) {
    f = q->father[node];
    if (node == exc || q->age[node] > time || (f >= 0 && q->age[f] <= time)) continue;
    if (pop == m->rootPop) { out[num++] = node; continue; }
    mig = go_find_last_mig(q, node, time);
    pop1 = (mig >= 0) ? q->mig[mig].source_pop : q->nodePop[node];
    if (m->isAnc[pop][pop1]) out[num++] = node;
  }
  return num;
}

/* ------------------------------------------------------------------ */
/* event-chain primitives                                              */

/* removeEvent, patch.c:1666-1700 */
int go_remove_event(go_locus *q, int ev)
{
  int nx = q->ev_next[ev], pv = q->ev_prev[ev];
  q->ev_time[nx] += q->ev_time[ev];
  q->ev_prev[nx] = pv;
  if (pv < 0) {
    for (pv = nx; q->ev_type[pv] != GO_END_CHAIN; pv = q->ev_next[pv]) { ; }
    q->first_event[q->ev_node[pv]] = nx;
  } else {
    q->ev_next[pv] = nx;
  }
  nx = q->free_events;
  q->ev_next[ev] = nx;
  q->ev_prev[nx] = ev;
  q->free_events = ev;
  q->ev_time[ev] = 0;
  q->ev_nlin[ev] = 0;
  q->ev_node[ev] = -1;
  return 0;
}

/* createEventBefore, patch.c:1707-1742 */
int go_create_event_before(go_state *s, go_locus *q, int pop, int ev, double elapsed)
{
  int pv = q->ev_prev[ev], nw = q->free_events;
  q->free_events = q->ev_next[nw];
  if (q->free_events < 0) go_fatal(s, 15, "empty event pool");
  q->ev_next[nw] = ev;
  q->ev_prev[nw] = pv;
  q->ev_nlin[nw] = q->ev_nlin[ev];
  q->ev_time[nw] = elapsed;
  q->ev_type[nw] = GO_DUMMY;
  q->ev_prev[ev] = nw;
  q->ev_time[ev] -= elapsed;
  if (pv < 0) q->first_event[pop] = nw;
  else q->ev_next[pv] = nw;
  return nw;
}

/* createEvent, patch.c:1753-1802 */
int go_create_event(go_state *s, go_locus *q, int pop, double age)
{
  go_model *m = &s->m;
  int ev;
  double dt = age - m->popAge[pop];
  if (dt < 0) return -1;
  if (pop != m->rootPop && age > m->popAge[m->popFather[pop]] + 0.000001) return -1;
  for (ev = q->first_event[pop]; q->ev_type[ev] != GO_END_CHAIN && q->ev_time[ev] < dt;
       ev = q->ev_next[ev])
    dt -= q->ev_time[ev];
  if (q->ev_time[ev] < dt) {
    if (q->ev_time[ev] < dt - 0.000001) go_fatal(s, 18, "createEvent above END_CHAIN");
    dt = q->ev_time[ev];
  }
  return go_create_event_before(s, q, pop, ev, dt);
}

/* populationPostOrder, patch.c:1936-1951 */
int go_pop_post_order(go_model *m, int pop, int *out)
{
  int size;
  if (pop < m->Kc) { out[0] = pop; return 1; }
  size = go_pop_post_order(m, m->popSon0[pop], out);
  size += go_pop_post_order(m, m->popSon1[pop], out + size);
  out[size] = pop;
  return size + 1;
}

/* recalcStats, patch.c:2387-2513 */
double go_recalc_stats(go_state *s, go_locus *q, int pop)
{
  go_model *m = &s->m;
  int n, id, b, ev, live[GO_MAXB], nlive = 0;
  double t, delta = 0.0;
  q->chk_coal_stats[pop] = 0.0;
  q->chk_num_coals[pop] = 0;
  ev = q->first_event[pop];
  n = q->ev_nlin[ev];
  for (; ev >= 0; ev = q->ev_next[ev]) {
    q->ev_nlin[ev] = n;
    id = q->ev_node[ev];
    t = q->ev_time[ev];
    q->chk_coal_stats[pop] += n * (n - 1) * t;
    for (b = 0; b < nlive; b++) q->chk_mig_stats[live[b]] += n * t;
    switch (q->ev_type[ev]) {
    case GO_SAMPLES_START: n += m->samplesPerPop[pop]; break;
    case GO_COAL: q->chk_num_coals[pop]++; n--; break;
    case GO_IN_MIG: q->chk_num_migs[q->mig[id].band]++; n--; break;
    case GO_OUT_MIG: n++; break;
    case GO_MIG_BAND_START:
      live[nlive++] = id;
      q->chk_num_migs[id] = 0;
      q->chk_mig_stats[id] = 0.0;
      break;
    case GO_MIG_BAND_END:
      delta -= (q->chk_mig_stats[id] - q->mig_stats[id]) * m->migRate[id];
      s->tot_mig_stats[id] += q->chk_mig_stats[id] - q->mig_stats[id];
      s->tot_num_migs[id] += q->chk_num_migs[id] - q->num_migs_band[id];
      q->mig_stats[id] = q->chk_mig_stats[id];
      q->num_migs_band[id] = q->chk_num_migs[id];
      for (b = 0; b < nlive; b++) if (live[b] == id) break;
      if (b == nlive) go_fatal(s, 25, "recalcStats band not alive");
      live[b] = live[--nlive];
      break;
    case GO_DUMMY:
    case GO_END_CHAIN: break;
    default: go_fatal(s, 26, "recalcStats bad event type");
    }
  }
  if (nlive != 0) go_fatal(s, 27, "recalcStats live bands at end");
  delta -= (q->chk_coal_stats[pop] - q->coal_stats[pop]) / (m->theta[pop]);
  s->tot_coal_stats[pop] += (q->chk_coal_stats[pop] - q->coal_stats[pop]);
  s->tot_num_coals[pop] += q->chk_num_coals[pop] - q->num_coals[pop];
  q->coal_stats[pop] = q->chk_coal_stats[pop];
  q->num_coals[pop] = q->chk_num_coals[pop];
  return delta;
}

/* computeGenetreeStats, patch.c:2330-2354 */
void go_compute_genetree_stats(go_state *s, go_locus *q)
{
  go_model *m = &s->m;
  int i, pop, queue[GO_MAXK];
  go_pop_post_order(m, m->rootPop, queue);
  for (i = 0; i < m->K; i++) {
    pop = queue[i];
    if (pop >= m->Kc)
      q->ev_nlin[q->first_event[pop]] = q->ev_nlin[m->popSon0[pop]] + q->ev_nlin[m->popSon1[pop]];
    else
      q->ev_nlin[q->first_event[pop]] = 0;
    go_recalc_stats(s, q, pop);
  }
}

/* gtreeLnLikelihood, patch.c:2702-2738 (no admixture) */
double go_gtree_lnl(go_state *s, go_locus *q)
{
  go_model *m = &s->m;
  int pop, b;
  double lnLd = 0, theta, rate;
  for (pop = 0; pop < m->K; pop++) {
    theta = m->theta[pop];
    lnLd += q->num_coals[pop] * log(2 / theta) - q->coal_stats[pop] / (theta);
  }
  for (b = 0; b < m->B; b++) {
    rate = m->migRate[b];
    if (rate > 0.0) lnLd += q->num_migs_band[b] * log(rate) - q->mig_stats[b] * rate;
  }
  return lnLd;
}

/* constructEventChain, patch.c:1961-2125 */
void go_construct_event_chain(go_state *s, go_locus *q)
{
  go_model *m = &s->m;
  int i, pop, mig, node, ev, b;
  double age;
  for (pop = 0; pop < m->K; pop++) {
    q->ev_type[pop] = GO_END_CHAIN;
    q->ev_next[pop] = -1;
    q->ev_prev[pop] = -1;
    q->ev_node[pop] = pop;
    q->ev_nlin[pop] = 0;
    if (pop == m->rootPop) q->ev_time[pop] = GO_OLDAGE - m->popAge[m->rootPop];
    else q->ev_time[pop] = m->popAge[m->popFather[pop]] - m->popAge[pop];
    q->first_event[pop] = pop;
  }
  q->free_events = m->K;
  q->ev_prev[m->K] = -1;
  q->ev_next[q->E - 1] = -1;
  for (ev = m->K; ev < q->E - 1; ev++) { q->ev_next[ev] = ev + 1; q->ev_prev[ev + 1] = ev; }
  for (b = 0; b < m->B; b++) {
    pop = m->bandTgt[b];
    ev = go_create_event(s, q, pop, m->bandStart[b]);
    if (ev < 0) go_fatal(s, 20, "band start event");
    q->ev_type[ev] = GO_MIG_BAND_START;
    q->ev_node[ev] = b;
    ev = go_create_event(s, q, pop, m->bandEnd[b]);
    if (ev < 0) go_fatal(s, 21, "band end event");
    q->ev_type[ev] = GO_MIG_BAND_END;
    q->ev_node[ev] = b;
  }
  for (pop = 0; pop < m->Kc; pop++) {
    ev = go_create_event(s, q, pop, m->sampleAge[pop]);
    q->ev_type[ev] = GO_SAMPLES_START;
  }
  for (i = 0; i < q->num_migs; i++) {
    mig = q->living[i];
    age = q->mig[mig].age;
    ev = go_create_event(s, q, q->mig[mig].target_pop, age);
    if (ev < 0) go_fatal(s, 22, "in-mig event");
    q->ev_type[ev] = GO_IN_MIG;
    q->ev_node[ev] = mig;
    q->mig[mig].target_event = ev;
    ev = go_create_event(s, q, q->mig[mig].source_pop, age);
    if (ev < 0) go_fatal(s, 23, "out-mig event");
    q->ev_type[ev] = GO_OUT_MIG;
    q->ev_node[ev] = mig;
    q->mig[mig].source_event = ev;
  }
  for (node = m->n; node < 2 * m->n - 1; node++) {
    ev = go_create_event(s, q, q->nodePop[node], q->age[node]);
    if (ev < 0) go_fatal(s, 24, "coal event");
    q->ev_type[ev] = GO_COAL;
    q->ev_node[ev] = node;
    q->nodeEvent[node] = ev;
  }
}

/* ------------------------------------------------------------------ */
/* considerEventMove and friends                                       */

/* computeMigStatsDelta, patch.c:1838-1864 */
static void mig_stats_delta(go_state *s, go_locus *q, int inst, double bottom_age, int bottom_pop,
                            double top_age, int dlin)
{
  go_model *m = &s->m;
  go_delta *d = &q->delta[inst];
  int b;
  double dt, lo, hi;
  d->num_bands_changed = 0;
  for (b = 0; b < m->B; b++) {
    if (!m->isAnc[m->bandTgt[b]][bottom_pop]) continue;
    hi = m->bandEnd[b] < top_age ? m->bandEnd[b] : top_age;
    lo = m->bandStart[b] > bottom_age ? m->bandStart[b] : bottom_age;
    dt = hi - lo;
    if (dt <= 0) continue;
    d->bands_changed[d->num_bands_changed] = b;
    d->mig_delta[d->num_bands_changed] = dlin * dt;
    d->num_bands_changed++;
  }
}

/* computeCoalStatsDelta, patch.c:1878-1927 */
static void coal_stats_delta(go_state *s, go_locus *q, int inst, int bottom_event, int bottom_pop,
                             int top_event, int dlin)
{
  go_model *m = &s->m;
  go_delta *d = &q->delta[inst];
  int pop = bottom_pop, ev = bottom_event;
  d->num_pops_changed = 1;
  d->pops_changed[0] = pop;
  d->coal_delta[0] = 0;
  d->num_changed_events = 0;
  while (ev >= 0) {
    d->coal_delta[d->num_pops_changed - 1] += dlin * (dlin - 1 + 2 * q->ev_nlin[ev]) * q->ev_time[ev];
    d->changed_events[d->num_changed_events] = ev;
    d->num_changed_events++;
    if (ev == top_event) break;
    ev = q->ev_next[ev];
    if (ev < 0) {
      if (m->popFather[pop] < 0) go_fatal(s, 19, "coal_stats_delta: top event not found");
      pop = m->popFather[pop];
      ev = q->first_event[pop];
      d->pops_changed[d->num_pops_changed] = pop;
      d->coal_delta[d->num_pops_changed] = 0;
      d->num_pops_changed++;
    }
  }
}

/* computeDeltaLnLd, patch.c:1516-1532 */
static double delta_lnld(go_state *s, go_locus *q, int inst)
{
  go_model *m = &s->m;
  go_delta *d = &q->delta[inst];
  int i;
  double r = 0;
  for (i = 0; i < d->num_pops_changed; i++) r -= d->coal_delta[i] / m->theta[d->pops_changed[i]];
  for (i = 0; i < d->num_bands_changed; i++) r -= d->mig_delta[i] * m->migRate[d->bands_changed[i]];
  return r;
}

/* considerEventMove, patch.c:1434-1507 */
double go_consider_event_move(go_state *s, go_locus *q, int inst, int event_id, int source_pop,
                              double original_age, int target_pop, double new_age)
{
  go_model *m = &s->m;
  go_delta *d = &q->delta[inst];
  int new_event, bottom_event, top_event, bottom_pop;
  double top_age, bottom_age, r;
  new_event = go_create_event(s, q, target_pop, new_age);
  if (new_event < 0) go_fatal(s, 13, "considerEventMove: createEvent");
  d->original_event = event_id;
  d->updated_event = new_event;
  if (new_age > original_age) {
    d->num_lin_delta = (q->ev_type[event_id] == GO_OUT_MIG) ? (-1) : (1);
    bottom_event = q->ev_next[event_id];
    top_event = new_event;
    bottom_pop = source_pop;
    top_age = new_age;
    bottom_age = original_age;
  } else {
    d->num_lin_delta = (q->ev_type[event_id] == GO_OUT_MIG) ? (1) : (-1);
    bottom_event = q->ev_next[new_event];
    top_event = event_id;
    bottom_pop = target_pop;
    top_age = original_age;
    bottom_age = new_age;
  }
  coal_stats_delta(s, q, inst, bottom_event, bottom_pop, top_event, d->num_lin_delta);
  mig_stats_delta(s, q, inst, bottom_age, bottom_pop, top_age, d->num_lin_delta);
  r = delta_lnld(s, q, inst);
  if (q->ev_type[event_id] == GO_COAL && source_pop != target_pop)
    r += log(m->theta[source_pop] / m->theta[target_pop]);
  return r;
}

static void delta_clear(go_delta *d)
{
  d->num_pops_changed = 0;
  d->num_bands_changed = 0;
  d->num_changed_events = 0;
  d->num_lin_delta = 0;
  d->original_event = -1;
  d->updated_event = -1;
}

/* acceptEventChainChanges, patch.c:1540-1633 */
void go_accept_event_chain_changes(go_state *s, go_locus *q, int inst)
{
  go_delta *d = &q->delta[inst];
  int i, pop, b, ue;
  for (i = 0; i < d->num_pops_changed; i++) {
    pop = d->pops_changed[i];
    q->coal_stats[pop] += d->coal_delta[i];
    s->tot_coal_stats[pop] += d->coal_delta[i];
  }
  for (i = 0; i < d->num_bands_changed; i++) {
    b = d->bands_changed[i];
    q->mig_stats[b] += d->mig_delta[i];
    s->tot_mig_stats[b] += d->mig_delta[i];
  }
  i = d->num_changed_events - 1;
  if (d->changed_events[i] == d->original_event) i--;
  for (; i >= 0; i--) q->ev_nlin[d->changed_events[i]] += d->num_lin_delta;
  if (d->updated_event >= 0) {
    ue = d->updated_event;
    q->ev_node[ue] = q->ev_node[d->original_event];
    q->ev_type[ue] = q->ev_type[d->original_event];
    switch (q->ev_type[ue]) {
    case GO_COAL: q->nodeEvent[q->ev_node[ue]] = ue; break;
    case GO_OUT_MIG: q->mig[q->ev_node[ue]].source_event = ue; break;
    case GO_IN_MIG: q->mig[q->ev_node[ue]].target_event = ue; break;
    default: go_fatal(s, 14, "acceptEventChainChanges: bad type");
    }
    go_remove_event(q, d->original_event);
  }
  delta_clear(d);
}

/* rejectEventChainChanges, patch.c:1639-1661 */
void go_reject_event_chain_changes(go_state *s, go_locus *q, int inst)
{
  go_delta *d = &q->delta[inst];
  (void)s;
  if (d->updated_event >= 0) go_remove_event(q, d->updated_event);
  delta_clear(d);
}

/* ------------------------------------------------------------------ */
/* rubberBand, patch.c:596-801                                         */
double go_rubber_band(go_state *s, go_locus *q, int pop, double static_point, double moving_point,
                      double factor, int post, int *out_num_events)
{
  go_model *m = &s->m;
  int i, ev, b, node_id, live[GO_MAXB], nlive = 0, num_lins, count_events = 0, flag;
  double age, dt, mig_rate = 0.0, mig_delta, coal_delta = 0.0, lnLd = 0.0, age1;
  double fm1 = factor - 1.0;
  double start_time = static_point < moving_point ? static_point : moving_point;
  double end_time = static_point > moving_point ? static_point : moving_point;
  if (pop == m->rootPop) { start_time = moving_point; end_time = GO_OLDAGE; }
  ev = q->first_event[pop];
  age = m->popAge[pop];
  flag = (age >= start_time);
  while (age < end_time) {
    if (ev == -1) go_fatal(s, 11, "rubberBand: bad event id");
    dt = q->ev_time[ev] < end_time - age ? q->ev_time[ev] : end_time - age;
    age += dt;
    if (!flag && age > start_time) { flag = 1; dt = age - start_time; }
    if (flag) {
      dt *= fm1;
      num_lins = q->ev_nlin[ev];
      mig_delta = dt * num_lins;
      coal_delta += mig_delta * (num_lins - 1);
      lnLd -= mig_delta * mig_rate;
      if (post) {
        q->ev_time[ev] += dt;
        for (b = 0; b < nlive; b++) {
          q->mig_stats[live[b]] += mig_delta;
          s->tot_mig_stats[live[b]] += mig_delta;
        }
      }
    }
    if (age >= end_time && q->ev_type[ev] != GO_SAMPLES_START) break;
    node_id = q->ev_node[ev];
    switch (q->ev_type[ev]) {
    case GO_COAL:
      if (flag) {
        count_events++;
        if (!post) {
          age1 = q->age[node_id];
          age1 += (age1 - static_point) * fm1;
          go_lik_adjust_age(q, node_id, age1);
        }
      }
      break;
    case GO_SAMPLES_START:
      if (flag && m->sampleAge[pop] > 0) {
        if (static_point < moving_point && !post) {
          age1 = m->sampleAge[pop];
          age1 += (age1 - static_point) * fm1;
          for (i = 0; i < m->n; i++)
            if (q->nodePop[i] == pop) go_lik_adjust_age(q, i, age1);
        }
      }
      break;
    case GO_IN_MIG:
      if (flag && post) q->mig[node_id].age += (q->mig[node_id].age - static_point) * fm1;
      break;
    case GO_MIG_BAND_START:
      mig_rate += m->migRate[node_id];
      live[nlive++] = node_id;
      break;
    case GO_MIG_BAND_END:
      mig_rate -= m->migRate[node_id];
      for (i = 0; i < nlive; i++) if (node_id == live[i]) break;
      if (i == nlive) go_fatal(s, 4, "rubberBand: band ended without starting");
      live[i] = live[--nlive];
      break;
    case GO_END_CHAIN: age = end_time; break;
    default: break;
    }
    ev = q->ev_next[ev];
  }
  if (post) {
    q->coal_stats[pop] += coal_delta;
    s->tot_coal_stats[pop] += coal_delta;
  }
  lnLd -= coal_delta / (m->theta[pop]);
  *out_num_events += count_events;
  return lnLd;
}

/* rubberBandRipple, patch.c:815-869 */
double go_rubber_band_ripple(go_state *s, go_locus *q, int do_or_redo)
{
  go_model *m = &s->m;
  int i, pop, nw, orig, affected[GO_MAXK];
  double delta = 0.0;
  if (q->rb_num_moved == 0) return 0.0;
  for (pop = 0; pop < m->K; pop++) affected[pop] = 0;
  for (i = 0; i < q->rb_num_moved; i++) {
    pop = q->rb_pops[i];
    orig = q->rb_orig[i];
    affected[pop] = 1;
    if (do_or_redo) {
      nw = q->rb_new[i] = go_create_event(s, q, pop, q->rb_new_ages[i]);
      if (nw < 0) go_fatal(s, 5, "ripple: createEvent");
      q->ev_type[nw] = q->ev_type[orig];
      q->ev_node[nw] = q->ev_node[orig];
      q->ev_type[orig] = GO_DUMMY;
    } else {
      nw = q->rb_new[i];
      q->ev_type[orig] = q->ev_type[nw];
      if (q->first_event[pop] == nw) q->ev_nlin[q->ev_next[nw]] = q->ev_nlin[nw];
      go_remove_event(q, nw);
    }
  }
  for (pop = 0; pop < m->K; pop++)
    if (affected[pop]) delta += go_recalc_stats(s, q, pop);
  if (!do_or_redo) q->rb_num_moved = 0;
  return delta;
}

/* ------------------------------------------------------------------ */
/* traceLineage, patch.c:886-1331.  reconnect == 0: walk the existing edge
 * above `node`, removing one lineage; reconnect == 1: re-sample its path */
int go_trace_lineage(go_state *s, go_locus *q, int node, int reconnect)
{
  go_model *m = &s->m;
  go_delta *d = &q->delta[reconnect];
  int i, pop, ev, node_id, b = -1, mig_source, proceed, nlive, live[GO_MAXB];
  int target, num_targets, targets[2 * 200];
  double age, t = 0, event_sample, rate, mig_rate, theta;

  pop = q->nodePop[node];
  if (node < m->n) {
    ev = q->first_event[pop];
    while (q->ev_type[ev] != GO_SAMPLES_START && q->ev_type[ev] != GO_END_CHAIN) ev = q->ev_next[ev];
    ev = q->ev_next[ev];
  } else {
    ev = q->ev_next[q->nodeEvent[node]];
  }
  theta = m->theta[pop];
  age = q->age[node];
  q->spr_delta_lnLd[reconnect] = 0.0;
  if (!reconnect) {
    q->spr_num_old_migs = 0;
    if (node != q->root) q->spr_father_event_old = q->nodeEvent[q->father[node]];
  } else {
    q->spr_num_new_migs = 0;
  }
  d->num_changed_events = 0;
  d->num_pops_changed = m->K;
  for (i = 0; i < m->K; i++) { d->pops_changed[i] = i; d->coal_delta[i] = 0.0; }
  d->num_bands_changed = m->B;
  for (i = 0; i < m->B; i++) { d->bands_changed[i] = i; d->mig_delta[i] = 0.0; }
  mig_rate = 0.0;
  nlive = 0;
  for (b = 0; b < m->B; b++) {
    if (m->bandTgt[b] == pop && m->bandStart[b] < age && m->bandEnd[b] > age) {
      mig_rate += m->migRate[b];
      live[nlive++] = b;
    }
  }
  mig_source = -1;
  proceed = 1;
  while (proceed) {
    if (ev < 0) {
      if (m->popFather[pop] < 0) {
        if (reconnect == 1) return -1;
        go_fatal(s, 6, "traceLineage: reached top event");
      }
      pop = m->popFather[pop];
      theta = m->theta[pop];
      ev = q->first_event[pop];
      mig_rate = 0.0;
      if (fabs(age / m->popAge[pop] - 1) > 0.01) go_fatal(s, 8, "traceLineage: age mismatch at pop start");
      age = m->popAge[pop];
    }
    node_id = q->ev_node[ev];
    if (!reconnect) {
      q->ev_nlin[ev]--;
      t = q->ev_time[ev];
      age += t;
      proceed = (ev != q->spr_father_event_old);
      if (q->ev_type[ev] == GO_IN_MIG) {
        if (q->mig[node_id].branch == node) {
          b = q->mig[node_id].band;
          mig_source = q->mig[node_id].source_event;
          q->spr_old_migs[q->spr_num_old_migs++] = node_id;
        }
      }
    } else {
      rate = mig_rate + 2 * q->ev_nlin[ev] / theta;
      if (rate <= 0) t = q->ev_time[ev];
      else t = -(1 / rate) * log(go_rndu(&q->rx, &q->ry, &q->rz));
      if (t >= q->ev_time[ev]) {
        t = q->ev_time[ev];
        age += t;
      } else {
        age += t;
        event_sample = rate * go_rndu(&q->rx, &q->ry, &q->rz);
        if (event_sample < mig_rate) {
          if (GO_MAX_MIGS <= q->num_migs + q->spr_num_new_migs - q->spr_num_old_migs) {
            s->not_enough_migs++;
            return -1;
          }
          for (i = 0; event_sample >= 0 && i < nlive; i++) event_sample -= m->migRate[live[i]];
          if (event_sample >= 0.0 && nlive <= 0) go_fatal(s, 9, "traceLineage: no live bands");
          if (i <= 0) go_fatal(s, 9, "traceLineage: i <= 0");
          q->spr_new_bands[q->spr_num_new_migs] = b = live[i - 1];
          if (m->bandTgt[b] != pop) go_fatal(s, 9, "traceLineage: band target mismatch");
          q->spr_new_ages[q->spr_num_new_migs] = age;
          ev = q->spr_new_in[q->spr_num_new_migs] = go_create_event_before(s, q, pop, ev, t);
          mig_source = q->spr_new_out[q->spr_num_new_migs] = go_create_event(s, q, m->bandSrc[b], age);
          if (mig_source < 0) go_fatal(s, 10, "traceLineage: out-mig event");
          q->spr_num_new_migs++;
        } else {
          num_targets = go_edges_for_time_pop(s, q, (age - t) + q->ev_time[ev] / 2, pop, node, targets);
          if (num_targets != q->ev_nlin[ev]) go_fatal(s, 11, "traceLineage: targets != lineages");
          i = (int)((event_sample - mig_rate) * theta / 2);
          target = targets[i];
          go_lik_spr(q, node, target, age);
          q->spr_father_pop_new = pop;
          q->spr_target = target;
          q->spr_father_event_new = ev = go_create_event_before(s, q, pop, ev, t);
          proceed = 0;
        }
      }
    }
    d->coal_delta[pop] += 2 * q->ev_nlin[ev] * t;
    for (i = 0; i < nlive; i++) d->mig_delta[live[i]] += t;
    d->changed_events[d->num_changed_events++] = ev;
    q->spr_delta_lnLd[reconnect] -= (mig_rate + 2 * q->ev_nlin[ev] / theta) * t;
    if (mig_source >= 0) {
      q->spr_delta_lnLd[reconnect] += log(m->migRate[b]);
      ev = mig_source;
      pop = m->bandSrc[b];
      theta = m->theta[pop];
      mig_source = -1;
      mig_rate = 0.0;
      nlive = 0;
      for (b = 0; b < m->B; b++) {
        if (m->bandTgt[b] == pop && m->bandStart[b] <= age && m->bandEnd[b] > age) {
          mig_rate += m->migRate[b];
          live[nlive++] = b;
        }
      }
    } else if (q->ev_type[ev] == GO_MIG_BAND_START) {
      mig_rate += m->migRate[node_id];
      live[nlive] = node_id;
      nlive++;
    } else if (q->ev_type[ev] == GO_MIG_BAND_END) {
      mig_rate -= m->migRate[node_id];
      if (nlive == 1) mig_rate = 0.0;
      for (i = 0; i < nlive; i++) {
        if (live[i] == node_id) { live[i] = live[--nlive]; break; }
      }
    }
    ev = q->ev_next[ev];
  }
  q->spr_delta_lnLd[reconnect] += log(2 / theta);
  return 0;
}

/* replaceMigNodes, patch.c:1343-1420 */
void go_replace_mig_nodes(go_state *s, go_locus *q, int node)
{
  go_model *m = &s->m;
  int i, j, mig, b;
  int mx = q->spr_num_old_migs > q->spr_num_new_migs ? q->spr_num_old_migs : q->spr_num_new_migs;
  for (i = 0; i < mx; i++) {
    if (i < q->spr_num_old_migs) {
      mig = q->spr_old_migs[i];
      go_remove_event(q, q->mig[mig].source_event);
      go_remove_event(q, q->mig[mig].target_event);
      b = q->mig[mig].band;
      q->num_migs_band[b]--;
      s->tot_num_migs[b]--;
    } else {
      for (mig = 0; mig < GO_MAX_MIGS; mig++) if (q->mig[mig].band < 0) break;
      if (mig == GO_MAX_MIGS) go_fatal(s, 12, "replaceMigNodes: no free mignode");
      q->living[q->num_migs++] = mig;
      q->mig[mig].branch = node;
    }
    if (i < q->spr_num_new_migs) {
      q->mig[mig].source_event = q->spr_new_out[i];
      q->mig[mig].target_event = q->spr_new_in[i];
      q->mig[mig].age = q->spr_new_ages[i];
      b = q->spr_new_bands[i];
      q->mig[mig].band = b;
      q->mig[mig].source_pop = m->bandSrc[b];
      q->mig[mig].target_pop = m->bandTgt[b];
      q->ev_type[q->spr_new_out[i]] = GO_OUT_MIG;
      q->ev_node[q->spr_new_out[i]] = mig;
      q->ev_type[q->spr_new_in[i]] = GO_IN_MIG;
      q->ev_node[q->spr_new_in[i]] = mig;
      q->num_migs_band[b]++;
      s->tot_num_migs[b]++;
    } else {
      q->mig[mig].band = -1;
      for (j = 0; j < q->num_migs; j++) {
        if (q->living[j] == mig) { q->living[j] = q->living[--q->num_migs]; break; }
      }
    }
  }
}

/* synchronizeEvents, patch.c:3548-3633 */
int go_synchronize_events(go_state *s, go_locus *q)
{
  go_model *m = &s->m;
  int i, pop, ev, id, queue[GO_MAXK], res = 1;
  double realAge = 0.0, age, PREC = 0.0000001;
  go_pop_post_order(m, m->rootPop, queue);
  for (i = 0; i < m->K; i++) {
    pop = queue[i];
    ev = q->first_event[pop];
    age = m->popAge[pop];
    for (; ev >= 0; ev = q->ev_next[ev]) {
      id = q->ev_node[ev];
      age += q->ev_time[ev];
      switch (q->ev_type[ev]) {
      case GO_SAMPLES_START: realAge = m->sampleAge[pop]; break;
      case GO_COAL: realAge = q->age[id]; break;
      case GO_IN_MIG:
      case GO_OUT_MIG: realAge = q->mig[id].age; break;
      case GO_MIG_BAND_START: realAge = m->bandStart[id]; break;
      case GO_MIG_BAND_END: realAge = m->bandEnd[id]; break;
      case GO_END_CHAIN:
        if (pop != m->rootPop) realAge = m->popAge[m->popFather[pop]];
        else realAge = age;
        break;
      default: realAge = age; break;
      }
      if (fabs(realAge - age) > PREC) res = 0;
      q->ev_time[ev] += realAge - age;
      if (q->ev_time[ev] < -PREC) res = 0;
      else if (q->ev_time[ev] < 0.0) q->ev_time[ev] = 0.0;
      age = realAge;
    }
  }
  return res;
}
