/* gphocs_oracle_int.h -- TEST INFRASTRUCTURE ONLY: internal prototypes of the restatement */
#ifndef GPHOCS_ORACLE_INT_H
#define GPHOCS_ORACLE_INT_H
#include "gphocs_oracle.h"

void go_fatal(go_state *s, int code, const char *what);
int go_lik_mark_cond(go_locus *q, int node);
void go_lik_save_node(go_locus *q, int node, int recalc);
void go_lik_adjust_age(go_locus *q, int node, double age);
double go_lik_scale_ages(go_state *s, go_locus *q, double factor);
int go_lik_spr(go_locus *q, int subtreeRoot, int target, double age);
int go_find_last_mig(go_locus *q, int node, double age);
int go_find_first_mig(go_locus *q, int node, double age);
int go_edges_for_time_pop(go_state *s, go_locus *q, double time, int pop, int exc, int *out);
int go_remove_event(go_locus *q, int ev);
int go_create_event_before(go_state *s, go_locus *q, int pop, int ev, double elapsed);
int go_create_event(go_state *s, go_locus *q, int pop, double age);
int go_pop_post_order(go_model *m, int pop, int *out);
double go_recalc_stats(go_state *s, go_locus *q, int pop);
void go_compute_genetree_stats(go_state *s, go_locus *q);
void go_construct_event_chain(go_state *s, go_locus *q);
double go_consider_event_move(go_state *s, go_locus *q, int inst, int event_id, int source_pop,
                              double original_age, int target_pop, double new_age);
void go_accept_event_chain_changes(go_state *s, go_locus *q, int inst);
void go_reject_event_chain_changes(go_state *s, go_locus *q, int inst);
double go_rubber_band(go_state *s, go_locus *q, int pop, double static_point, double moving_point,
                      double factor, int post, int *out_num_events);
double go_rubber_band_ripple(go_state *s, go_locus *q, int do_or_redo);
int go_trace_lineage(go_state *s, go_locus *q, int node, int reconnect);
void go_replace_mig_nodes(go_state *s, go_locus *q, int node);

#endif
