/* oracle_internal.h -- shared between the oracle's translation units
 * (TEST INFRASTRUCTURE, see oracle.h). */
#ifndef GRL_ORACLE_INTERNAL_H_
#define GRL_ORACLE_INTERNAL_H_

#include "oracle.h"

#define ORC_MAX_TILINGS 32
#define ORC_MAX_TRACE   64
#define ORC_MAX_ACTIONS 16
#define ORC_INVALID     UINT64_MAX      /* IndexProjection::invalid_index(), projection.h:73-76 */

typedef struct {
  uint64_t idx[ORC_MAX_TILINGS];
  int      n;
} orc_proj;

typedef struct {
  orc_proj p[ORC_MAX_TRACE];            /* oldest first, like the reference's deque */
  double   decay[ORC_MAX_TRACE];
  int      len;
  double   total_decay;
} orc_trace;

struct orc_exp {
  orc_spec   spec;
  orc_rand48 G;                         /* global srand48/lrand48/drand48 state       */
  orc_rand48 TL;                        /* thread-local RandGen instance              */
  orc_rand48 S1, S2;                    /* learning / test sampler private Rand       */
  double    *w[2];                      /* weight tables: 0 = Q or critic, 1 = actor  */
  double    *wt;                        /* target network of table 0 (NULL without one) */
  int32_t   *claim;                     /* TileCodingProjector::indices_ (safe >= 1), -1 = free */
  int64_t    sync_count;                /* ParameterizedRepresentation::count_          */
  int64_t    syncs;                     /* synchronisations so far (diagnostics)        */
  int        A;                         /* number of discrete actions                 */
  double     actions[ORC_MAX_ACTIONS];
  double     state[ORC_MAX_STATE];
  /* TDAgent state (td.h) */
  double     prev_obs[ORC_MAX_DIMS];
  double     prev_action;
  int        prev_action_index;
  double     time;
  double     test_time;                 /* FixedAgent::time_ */
  double     eps_decay;                 /* EpsilonGreedySampler::decay_ */
  double     ac_decay, ac_noise;        /* ActionPolicy::decay_, n_ */
  orc_trace  trace;                     /* predictor (critic) trace */
  int64_t    tt, ss;                    /* trial counter, learning steps */
  uint64_t   steps_budget;              /* experiment/online_learning:steps (online_learning.cpp:154); 0 = none */
  orc_stats  stats;
  /* what the last orc_exp_agent_* call did (taps of orc_run; inspection by the per-step tests) */
  int        last_index;                /* discrete action index chosen (Q agents) */
  double     last_value;                /* action value returned */
  double     last_q[ORC_MAX_ACTIONS];   /* Q(s', .) as the policy saw it (actor-critic: the actor's output in [0]) */
  double     last_delta;                /* TD error of the update */
  orc_proj   last_p, last_ap;           /* projections written by the update: project(prev_obs, prev_action); actor's (AC) / V's (QV) */
};

double orc_m_sin(const orc_spec *s, double x);
double orc_m_cos(const orc_spec *s, double x);
double orc_m_log(const orc_spec *s, double x);
double orc_m_sqr(const orc_spec *s, double x);
double orc_m_powtau(const orc_spec *s, double base, double tau);

void orc_env_start(const orc_spec *s, orc_exp *e, int test, double *x);
int  orc_env_observe(const orc_spec *s, const double *x, double *obs);

#endif
