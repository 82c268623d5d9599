/* experiment.c -- OnlineLearningExperiment loop with TD agents over hashed
 * tile-coded linear representations, restated (TEST INFRASTRUCTURE, see oracle.h).
 *
 * Each function cites the reference lines it follows.  Scalar, single thread,
 * IEEE double, no contraction.
 */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

/* ---------------------------------------------------------------- spec -- */
void orc_spec_pendulum_sarsa(orc_spec *s)
{ /* values of the reference's tests/pendulum-sarsa-tc.yaml */
  memset(s, 0, sizeof(*s));
  s->test_interval = 10;
  s->env = ORC_ENV_PENDULUM;
  s->control_step = 0.03;
  s->integration_steps = 5;
  s->timeout = 2.99;
  s->randomization = 0;
  s->action_min = -3;            /* task/pendulum/swingup action_min/max, pendulum.cpp:89-90 */
  s->action_max = 3;
  s->action_steps = 3;
  s->agent = ORC_AGENT_SARSA;
  s->projector.tilings = 16;
  s->projector.memory = 8388608;
  s->projector.dims = 3;
  s->projector.resolution[0] = 0.31415;
  s->projector.resolution[1] = 3.1415;
  s->projector.resolution[2] = 3;
  s->projector.wrapping[0] = 6.283;
  s->representation.init_min = 0;
  s->representation.init_max = 1;
  s->representation.output_min = -DBL_MAX;   /* linear.cpp:86-96: empty => +-DBL_MAX */
  s->representation.output_max = DBL_MAX;
  s->representation.limit = 1;               /* linear.h:46-49 */
  s->epsilon = 0.05;
  s->decay_rate = 1;                         /* greedy.h:75 */
  s->decay_min = 0;
  s->alpha = 0.2;
  s->gamma = 0.97;
  s->lambda = 0.65;
  s->trace = ORC_TRACE_REPLACING;
  s->ac_step_limit = -1;
  s->math = ORC_MATH_LIBM;
}

void orc_spec_cart_pole_balancing_pid(orc_spec *s)
{ /* values of the reference's tests/cart_pole_balancing-pid.yaml: dynamics/cart_pole (end_stop = 1, the class
   * default, cart_pole.h:52) under model/dynamical {control_step 0.05, integration_steps 5},
   * task/cart_pole/balancing {timeout 9.99}, agent/fixed with policy/parameterized/pid
   * {setpoint [0,0,0,0], p [-10,-50,-6,-10]} as both the agent and the test agent; test_interval 0. */
  memset(s, 0, sizeof(*s));
  s->test_interval = 0;
  s->env = ORC_ENV_CART_POLE_BALANCING;
  s->control_step = 0.05;
  s->integration_steps = 5;
  s->timeout = 9.99;
  s->action_min = -15;           /* task/cart_pole/balancing action_min/max, cart_pole.cpp:255-256 */
  s->action_max = 15;
  s->action_steps = 1;
  s->agent = ORC_AGENT_PID;
  s->pid_p[0] = -10; s->pid_p[1] = -50; s->pid_p[2] = -6; s->pid_p[3] = -10;
  s->ac_step_limit = -1;
  s->math = ORC_MATH_LIBM;
}

/* ------------------------------------------------------- representation -- */
static double lin_read(orc_exp *e, int table, const orc_linear_spec *ls, const orc_proj *p)
{ /* linear.cpp:136-184, IndexProjection with empty weights */
  const double *w = e->w[table];
  double r = 0;
  for (int i = 0; i < p->n; ++i)
    r += w[p->idx[i]];
  r /= p->n;
  r = fmin(fmax(r, ls->output_min), ls->output_max);
  e->stats.weight_reads += (uint64_t)p->n;
  return r;
}

static void target_synchronize(orc_exp *e)
{ /* ParameterizedRepresentation::synchronize (representation.h:284-296): over the WHOLE parameter vector */
  const size_t n = (size_t)e->spec.projector.memory;
  const double tau = e->spec.target_tau;
  if (tau)
    for (size_t i = 0; i < n; ++i) e->wt[i] = tau * e->w[0][i] + (1 - tau) * e->wt[i];
  else
    memcpy(e->wt, e->w[0], n * sizeof(double));
  e->sync_count = 0;
  e->syncs++;
}

static void lin_update(orc_exp *e, int table, const orc_linear_spec *ls, const orc_proj *p, double delta)
{ /* linear.cpp:198-216: every valid index, duplicates applied twice */
  double *w = e->w[table];
  for (int i = 0; i < p->n; ++i)
    if (p->idx[i] != ORC_INVALID)
    {
      if (ls->limit)
        w[p->idx[i]] = fmin(fmax(w[p->idx[i]] + delta, ls->output_min), ls->output_max);
      else
        w[p->idx[i]] = w[p->idx[i]] + delta;
      e->stats.weight_rmws++;
    }
  if (table == 0 && e->wt)
  { /* checkSynchronize (linear.cpp:267, representation.h:298-305): once per update() call */
    e->sync_count++;
    if (e->sync_count >= e->spec.target_interval) target_synchronize(e);
  }
}

static double lin_read_target(orc_exp *e, const orc_linear_spec *ls, const orc_proj *p)
{ /* representation_->target()->read(...): the target network when there is one (representation.h:266-272) */
  if (!e->wt) return lin_read(e, 0, ls, p);
  double *keep = e->w[0];
  e->w[0] = e->wt;
  const double r = lin_read(e, 0, ls, p);
  e->w[0] = keep;
  return r;
}

static void lin_write(orc_exp *e, int table, const orc_linear_spec *ls, const orc_proj *p, double target, double alpha)
{ /* linear.cpp:186-196 */
  double value = lin_read(e, table, ls, p);
  double delta = alpha * (target - value);
  lin_update(e, table, ls, p, delta);
}

/* ---------------------------------------------------------------- trace -- */
static void trace_clear(orc_trace *t)
{ /* trace.h:200-204 */
  t->len = 0;
  t->total_decay = 1.;
}

static void proj_ssub(orc_proj *a, const orc_proj *b)
{ /* projection.h:94-104 */
  for (int i = 0; i < a->n; ++i)
    for (int j = 0; j < b->n; ++j)
      if (a->idx[i] == b->idx[j])
        a->idx[i] = ORC_INVALID;
}

static void trace_pop_front(orc_trace *t)
{
  memmove(&t->p[0], &t->p[1], sizeof(orc_proj) * (size_t)(t->len - 1));
  memmove(&t->decay[0], &t->decay[1], sizeof(double) * (size_t)(t->len - 1));
  t->len--;
}

static void trace_add(orc_trace *t, int kind, const orc_proj *p, double decay)
{
  double cut = (kind == ORC_TRACE_REPLACING) ? 0.01 : 0.0001;   /* trace.h:218, :249 */
  if (decay < cut)
    trace_clear(t);
  if (kind == ORC_TRACE_REPLACING)                              /* trace.h:221-222 */
    for (int i = 0; i < t->len; ++i)
      proj_ssub(&t->p[i], p);
  if (t->len >= ORC_MAX_TRACE)
  {
    fprintf(stderr, "oracle: trace overflow\n");
    abort();
  }
  t->p[t->len] = *p;                                            /* :224-225 / :254-255 */
  t->decay[t->len] = decay;
  t->len++;
  t->total_decay *= decay;
  while (t->total_decay < cut && t->len > 1)                    /* :227-231 / :257-261 */
  {
    t->total_decay /= t->decay[0];
    trace_pop_front(t);
  }
}

static void lin_update_trace(orc_exp *e, int table, const orc_linear_spec *ls, const orc_trace *t, double delta, double ee)
{ /* representation.h:79-83 with the iterator of trace.h:150-178: newest first,
   * weight 1, multiplied by the decay of the entry being left */
  double weight = 1.;
  for (int k = 0; k < t->len && weight > 0.001; ++k)
  {
    int at = t->len - k - 1;
    lin_update(e, table, ls, &t->p[at], weight * delta * ee);
    weight *= t->decay[at];
  }
}

/* ------------------------------------------------------------ projector -- */
static void project_sa_claim(orc_exp *e, const double *obs, double action, int claim, orc_proj *p)
{ /* tile_coding.h:62-73: project(in) = _project(in, true); project(base, variants) = _project(., safe_ > 1) */
  const orc_tile_spec *ts = &e->spec.projector;
  double in[ORC_MAX_DIMS];
  uint32_t out[ORC_MAX_TILINGS];
  int D = ts->dims - 1;
  for (int i = 0; i < D; ++i) in[i] = obs[i];
  in[D] = action;
  if ((e->claim ? orc_tile_project_hash(ts, in, out) : orc_tile_project(ts, in, out)) != 0)
  {
    fprintf(stderr, "oracle: invalid tile coding spec\n");
    abort();
  }
  p->n = ts->tilings;
  for (int j = 0; j < p->n; ++j)
  {
    if (!e->claim) { p->idx[j] = out[j]; continue; }
    /* getFeatureLocation (tile_coding.h:116-151): linear probing from h % memory over slots claimed by OTHER hash sums */
    const uint32_t h = out[j], mem = (uint32_t)ts->memory;
    uint32_t ii = h % mem;
    while (e->claim[ii] != (int32_t)h && e->claim[ii] != -1)
      if (++ii >= mem) ii = 0;
    if (claim) e->claim[ii] = (int32_t)h;
    p->idx[j] = ii;
  }
}
static void project_sa(orc_exp *e, const double *obs, double action, orc_proj *p)
{ /* a SINGLE projection: claims under safe >= 1 (tile_coding.h:62-66) */
  project_sa_claim(e, obs, action, 1, p);
}
static void project_sa_batch(orc_exp *e, const double *obs, double action, orc_proj *p)
{ /* one variant of project(base, variants): claims only under safe = 2 (tile_coding.h:67-73) */
  project_sa_claim(e, obs, action, e->spec.safe > 1, p);
}

static void project_obs(const orc_tile_spec *ts, const double *obs, orc_proj *p)
{ /* TileCodingProjector::project(in) on the bare observation (actor / V-critic) */
  uint32_t out[ORC_MAX_TILINGS];
  if (orc_tile_project(ts, obs, out) != 0)
  {
    fprintf(stderr, "oracle: invalid tile coding spec\n");
    abort();
  }
  p->n = ts->tilings;
  for (int j = 0; j < p->n; ++j) p->idx[j] = out[j];
}

/* -------------------------------------------------------------- sampler -- */
static void findmax(const double *v, int n, int *mai, int *man)
{ /* greedy.cpp:47-61 */
  *mai = 0;
  *man = 1;
  for (int i = 1; i < n; ++i)
  {
    if (v[i] > v[*mai]) { *mai = i; *man = 1; }
    else if (v[i] == v[*mai]) (*man)++;
  }
}

static int tie_break(orc_exp *e, const double *v, int mai, int man)
{ /* greedy.cpp:77-85 / :206-214; getInteger uses the GLOBAL lrand48 (utils.h:127-130) */
  int ii = mai;
  e->stats.ties++;
  for (int jj = (int)(orc_lrand48(&e->G) % (uint32_t)man); jj >= 0; ++ii)
    if (v[ii] == v[mai])
      --jj;
  return ii - 1;
}

static int sample_greedy(orc_exp *e, const double *v, int n)
{ /* greedy.cpp:63-86 */
  int mai, man;
  findmax(v, n, &mai, &man);
  if (man > 1)
    return tie_break(e, v, mai, man);
  return mai;
}

static int sample_eps_greedy(orc_exp *e, double time, const double *v, int n)
{ /* greedy.cpp:144-218, scalar epsilon */
  int mai, man;
  if (time == 0.)
    e->eps_decay = fmax(e->eps_decay * e->spec.decay_rate, e->spec.decay_min);
  findmax(v, n, &mai, &man);
  double r = orc_drand48(&e->S1);
  if (r < e->eps_decay * e->spec.epsilon)
  {
    e->stats.explorations++;
    return (int)(orc_lrand48(&e->G) % (uint32_t)n);
  }
  if (man > 1)
    return tie_break(e, v, mai, man);
  return mai;
}

/* --------------------------------------------------------------- policy -- */
static void q_values(orc_exp *e, const double *obs, double *q)
{ /* q.cpp:94-107 */
  orc_proj p;
  for (int a = 0; a < e->A; ++a)
  {
    project_sa_batch(e, obs, e->actions[a], &p);                  /* projector_->project(in, variants, &actions) */
    q[a] = lin_read(e, 0, &e->spec.representation, &p);
  }
}

/* ------------------------------------------------------------ predictor -- */
static double sarsa_update(orc_exp *e, const double *prev_obs, double prev_action, double tau,
                           double reward, const double *obs, int has_action, double action, orc_proj *pout)
{ /* sarsa.cpp:98-124 (criticize with an empty `action` argument, predictor.h:75-78) */
  const orc_spec *s = &e->spec;
  orc_proj p, pn;
  project_sa(e, prev_obs, prev_action, &p);
  double target = reward;
  if (has_action)
  {
    project_sa(e, obs, action, &pn);
    target += orc_m_powtau(s, s->gamma, tau) * lin_read_target(e, &s->representation, &pn);     /* sarsa.cpp:107 */
  }
  double delta = target - lin_read(e, 0, &s->representation, &p);
  lin_write(e, 0, &s->representation, &p, target, s->alpha);
  if (s->trace != ORC_TRACE_NONE)
  {
    double ee = orc_m_powtau(s, s->gamma * s->lambda, tau);
    lin_update_trace(e, 0, &s->representation, &e->trace, s->alpha * delta, ee);
    trace_add(&e->trace, s->trace, &p, ee);
  }
  if (has_action && e->claim)
  { /* sarsa.cpp:120-121: the critique reads project(prev_obs, action) -- a single projection, which CLAIMS its slots */
    orc_proj pc;
    project_sa(e, prev_obs, action, &pc);
  }
  *pout = p;
  return delta;
}

static double expected_value(orc_exp *e, const double *obs)
{ /* QPolicy::value (q.cpp:60-73) with EpsilonGreedySampler::distribution (greedy.cpp:220-238) over
   * GreedySampler::distribution (:88-99), scalar epsilon */
  double q[ORC_MAX_ACTIONS], dist[ORC_MAX_ACTIONS], v = 0;
  int mai, man;
  q_values(e, obs, q);
  findmax(q, e->A, &mai, &man);
  for (int i = 0; i < e->A; ++i)
    dist[i] = (q[i] == q[mai]) ? 1. / man : 0.;
  for (int i = 0; i < e->A; ++i)
  {
    if (dist[i] == 1)
      dist[i] = 1 - e->eps_decay * e->spec.epsilon;
    dist[i] += e->eps_decay * e->spec.epsilon / e->A;
  }
  for (int i = 0; i < e->A; ++i)
    v += q[i] * dist[i];
  return v;
}

static double expected_sarsa_update(orc_exp *e, const double *prev_obs, double prev_action, double tau,
                                    double reward, const double *obs, int has_action, orc_proj *pout)
{ /* ExpectedSARSAPredictor::criticize, sarsa.cpp:167-194.  parity unpinned by reference tests. */
  const orc_spec *s = &e->spec;
  orc_proj p;
  project_sa(e, prev_obs, prev_action, &p);
  double target = reward;
  if (has_action)
    target += orc_m_powtau(s, s->gamma, tau) * expected_value(e, obs);
  double delta = target - lin_read(e, 0, &s->representation, &p);
  lin_write(e, 0, &s->representation, &p, target, s->alpha);
  if (s->trace != ORC_TRACE_NONE)
  {
    double ee = orc_m_powtau(s, s->gamma * s->lambda, tau);
    lin_update_trace(e, 0, &s->representation, &e->trace, s->alpha * delta, ee);
    trace_add(&e->trace, s->trace, &p, ee);
  }
  *pout = p;
  return delta;
}

static double q_update(orc_exp *e, const double *prev_obs, double prev_action, double tau,
                       double reward, const double *obs, int has_action, double action, orc_proj *pout)
{ /* advantage.cpp:71-110 (QPredictor::criticize).  parity unpinned by reference tests. */
  const orc_spec *s = &e->spec;
  orc_proj p, pa;
  project_sa(e, prev_obs, prev_action, &p);
  double target = reward;
  if (has_action)
  {
    double v = -INFINITY;
    for (int kk = 0; kk < e->A; ++kk)
    {
      project_sa_batch(e, obs, e->actions[kk], &pa);                                            /* advantage.cpp:85-86 */
      v = fmax(v, lin_read_target(e, &s->representation, &pa));                                /* advantage.cpp:88 */
    }
    target += orc_m_powtau(s, s->gamma, tau) * v;
  }
  double delta = target - lin_read(e, 0, &s->representation, &p);
  if (has_action && e->claim)
  { /* advantage.cpp:95-97: the critique reads project(prev_obs, action) -- a single projection, which CLAIMS its slots */
    orc_proj pc;
    project_sa(e, prev_obs, action, &pc);
  }
  lin_write(e, 0, &s->representation, &p, target, s->alpha);
  if (s->trace != ORC_TRACE_NONE)
  {
    double ee = orc_m_powtau(s, s->gamma * s->lambda, tau);
    lin_update_trace(e, 0, &s->representation, &e->trace, s->alpha * delta, ee);
    trace_add(&e->trace, s->trace, &p, ee);
  }
  *pout = p;
  return delta;
}

static double advantage_update(orc_exp *e, const double *prev_obs, double prev_action, double tau,
                               double reward, const double *obs, int has_action, orc_proj *pout)
{ /* advantage.cpp:222-268 (AdvantagePredictor::criticize): advantage learning with scaling kappa.
   * parity unpinned by reference tests. */
  const orc_spec *s = &e->spec;
  orc_proj p, pa;
  project_sa(e, prev_obs, prev_action, &p);
  const double a = lin_read(e, 0, &s->representation, &p);            /* A(x_t, u_t) */
  double v = -INFINITY;                                                /* max_u A(x_t, u) */
  for (int kk = 0; kk < e->A; ++kk)
  {
    project_sa(e, prev_obs, e->actions[kk], &pa);
    v = fmax(v, lin_read(e, 0, &s->representation, &pa));
  }
  double target = v + (reward - v) / s->kappa;
  if (has_action)
  {
    v = -INFINITY;                                                     /* max_u A(x_{t+1}, u) */
    for (int kk = 0; kk < e->A; ++kk)
    {
      project_sa(e, obs, e->actions[kk], &pa);
      v = fmax(v, lin_read(e, 0, &s->representation, &pa));
    }
    target += orc_m_powtau(s, s->gamma, tau) * v / s->kappa;
  }
  double delta = target - a;
  lin_write(e, 0, &s->representation, &p, target, s->alpha);
  if (s->trace != ORC_TRACE_NONE)
  {
    double ee = orc_m_powtau(s, s->gamma * s->lambda, tau);
    lin_update_trace(e, 0, &s->representation, &e->trace, s->alpha * delta, ee);
    trace_add(&e->trace, s->trace, &p, ee);
  }
  *pout = p;
  return delta;
}

/* -------------------------------------------------------- actor-critic ---
 * parity unpinned by reference tests (its cart-pole AC test yaml is stale). */
static double rand_normal(orc_exp *e, double mu, double sigma)
{ /* Rand::getNormal, utils.h:120-125 (Box-Muller, two thread-local draws) */
  double U1 = orc_drand48(&e->TL), U2 = orc_drand48(&e->TL);
  return sqrt(-2 * orc_m_log(&e->spec, U1)) * orc_m_cos(&e->spec, 2 * M_PI * U2) * sigma + mu;
}

static double ac_policy_act(orc_exp *e, int test, double time, const double *obs, double *u_out)
{ /* ActionPolicy::act(time, in, out), action.cpp:127-158.  The test policy has sigma [] => 0. */
  const orc_spec *s = &e->spec;
  orc_proj p;
  project_obs(&s->actor_projector, obs, &p);
  double out = lin_read(e, 1, &s->actor_representation, &p);          /* ActionPolicy::read, :97-110 */
  *u_out = out;
  if (!test)
  {
    if (time == 0) e->ac_noise = 0;
    if (time == 0.) e->ac_decay = fmax(e->ac_decay * s->ac_decay_rate, s->ac_decay_min);
    if (s->sigma)
    {
      e->ac_noise = (1 - s->theta) * e->ac_noise + rand_normal(e, 0., e->ac_decay * s->sigma);
      out += e->ac_noise;
    }
  }
  return fmin(fmax(out, s->action_min), s->action_max);
}

static double td_critic(orc_exp *e, const double *prev_obs, double tau, double reward, const double *obs, int has_next, orc_proj *pout)
{ /* TDPredictor::criticize, predictors/td.cpp:68-91 */
  const orc_spec *s = &e->spec;
  orc_proj p, pn;
  project_obs(&s->projector, prev_obs, &p);
  double target = reward;
  if (has_next)
  {
    project_obs(&s->projector, obs, &pn);
    target += orc_m_powtau(s, s->gamma, tau) * lin_read(e, 0, &s->representation, &pn);
  }
  double delta = target - lin_read(e, 0, &s->representation, &p);
  lin_write(e, 0, &s->representation, &p, target, s->alpha);
  if (s->trace != ORC_TRACE_NONE)
  {
    double ee = orc_m_powtau(s, s->gamma * s->lambda, tau);
    lin_update_trace(e, 0, &s->representation, &e->trace, s->alpha * delta, ee);
    trace_add(&e->trace, s->trace, &p, ee);
  }
  *pout = p;
  return delta;
}

static double ac_update(orc_exp *e, const double *prev_obs, double prev_action, double tau, double reward,
                        const double *obs, int has_next, orc_proj *pout, orc_proj *apout)
{ /* ActionACPredictor::update, ac.cpp:72-110 */
  const orc_spec *s = &e->spec;
  orc_proj ap;
  project_obs(&s->actor_projector, prev_obs, &ap);
  double u = lin_read(e, 1, &s->actor_representation, &ap);
  double critique = td_critic(e, prev_obs, tau, reward, obs, has_next, pout);
  if (s->ac_update_method == ORC_AC_PROPORTIONAL || critique > 0)
  {
    double delta = prev_action - u;
    if (s->ac_update_method == ORC_AC_PROPORTIONAL)
      delta = critique * delta;
    if (s->ac_step_limit >= 0)
      delta = fmin(fmax(delta, -s->ac_step_limit), s->ac_step_limit);
    double target_u = u + delta;
    lin_write(e, 1, &s->actor_representation, &ap, target_u, s->actor_alpha);
  }
  *apout = ap;
  return critique;
}

static double qv_update(orc_exp *e, const double *prev_obs, double prev_action, double tau, double reward,
                        const double *obs, int has_action, orc_proj *qpout, orc_proj *vpout)
{ /* QVPredictor::criticize, qv.cpp:74-108: Q(s,a) and V(s) both move towards r + gamma^tau V(s');
   * only V has a trace.  Table 0 = Q (the policy's), table 1 = V.  parity unpinned by reference tests. */
  const orc_spec *s = &e->spec;
  orc_proj qp, vp, vn;
  project_sa(e, prev_obs, prev_action, &qp);
  project_obs(&s->actor_projector, prev_obs, &vp);
  project_obs(&s->actor_projector, obs, &vn);
  const double vnext = lin_read(e, 1, &s->actor_representation, &vn);
  double target = reward;
  if (has_action)
    target += orc_m_powtau(s, s->gamma, tau) * vnext;
  double delta = target - lin_read(e, 1, &s->actor_representation, &vp);
  lin_write(e, 0, &s->representation, &qp, target, s->alpha);            /* Q update */
  lin_write(e, 1, &s->actor_representation, &vp, target, s->beta);       /* V update */
  if (s->trace != ORC_TRACE_NONE)
  {
    double ee = orc_m_powtau(s, s->gamma * s->lambda, tau);
    lin_update_trace(e, 1, &s->actor_representation, &e->trace, s->beta * delta, ee);
    trace_add(&e->trace, s->trace, &vp, ee);
  }
  *qpout = qp;
  *vpout = vp;
  return delta;
}

/* ---------------------------------------------------------------- agent -- */
typedef struct { double value; int index; double q[ORC_MAX_ACTIONS]; } act_t;

static void policy_act(orc_exp *e, int test, double time, const double *obs, act_t *out)
{ /* q.cpp:143-155 (QPolicy::act with time) */
  if (e->spec.agent == ORC_AGENT_PID)
  { /* PIDPolicy::act(time, in, out), pid.cpp:136-179, with `p` gains only (i, d, il empty), one output:
     * u = sum_ii p[P(ii, 0)] * (setpoint[ii] - in[ii]) accumulated in index order from 0, then clamped to
     * the task's action range (pid.cpp:175); P(i, o) = o*N + i (pid.cpp:30) */
    const int D = orc_env_obs_dims(e->spec.env);
    double u = 0;
    (void)time;
    for (int ii = 0; ii < D; ++ii)
    {
      double err = e->spec.pid_setpoint[ii] - obs[ii];
      u += e->spec.pid_p[ii] * err;
    }
    out->index = 0;
    out->value = fmin(e->spec.action_max, fmax(u, e->spec.action_min));
    out->q[0] = out->value;
    return;
  }
  if (e->spec.agent == ORC_AGENT_AC)
  {
    out->index = 0;
    out->value = ac_policy_act(e, test, time, obs, &out->q[0]);
    return;
  }
  q_values(e, obs, out->q);
  if (test)
    out->index = sample_greedy(e, out->q, e->A);
  else
    out->index = sample_eps_greedy(e, time, out->q, e->A);
  out->value = e->actions[out->index];        /* discretizer_->at(index), uniform.cpp:140-151 */
}

/* --------------------------------------------------------------- create -- */
static double *table_alloc_init(orc_exp *e, const orc_tile_spec *ts, const orc_linear_spec *ls)
{ /* linear.cpp:110-121: memory*outputs uniforms from the thread-local RandGen */
  size_t n = (size_t)ts->memory;
  double *w = (double *)malloc(n * sizeof(double));
  if (!w) return NULL;
  for (size_t i = 0; i < n; ++i)
    w[i] = ls->init_min + orc_drand48(&e->TL) * (ls->init_max - ls->init_min);   /* utils.h:110-113 */
  return w;
}

orc_exp *orc_create(const orc_spec *spec, long seed)
{
  if (orc_env_state_dims(spec->env) < 0) return NULL;             /* environments not restated yet */
  if (spec->agent != ORC_AGENT_SARSA && spec->agent != ORC_AGENT_Q && spec->agent != ORC_AGENT_AC && spec->agent != ORC_AGENT_EXPECTED_SARSA &&
      spec->agent != ORC_AGENT_ADVANTAGE && spec->agent != ORC_AGENT_QV && spec->agent != ORC_AGENT_PID) return NULL;
  if (spec->agent == ORC_AGENT_PID)
  { /* tests/cart_pole_balancing-pid.yaml: no representation, no sampler.  Nothing consumes a random number at
     * instantiation; the first RandGen::get() (the task's start, cart_pole.cpp:269) creates the thread-local
     * Rand, seeded by the first global lrand48() (utils.h:90-93, 160-171). */
    orc_exp *e = (orc_exp *)calloc(1, sizeof(*e));
    if (!e) return NULL;
    e->spec = *spec;
    orc_srand48(&e->G, seed);
    orc_srand48(&e->TL, (long)orc_lrand48(&e->G));
    e->A = 1;
    e->ac_decay = 1;
    e->eps_decay = 1;
    trace_clear(&e->trace);
    return e;
  }
  if (spec->agent == ORC_AGENT_QV && (spec->actor_projector.dims != orc_env_obs_dims(spec->env) || spec->actor_projector.tilings > ORC_MAX_TILINGS)) return NULL;
  if (spec->agent == ORC_AGENT_ADVANTAGE && !(spec->kappa > 0)) return NULL;
  if (spec->agent == ORC_AGENT_AC)
  {
    if (spec->projector.dims != orc_env_obs_dims(spec->env) || spec->actor_projector.dims != orc_env_obs_dims(spec->env)) return NULL;
    if (spec->actor_projector.tilings > ORC_MAX_TILINGS) return NULL;
  }
  else
  {
    if (spec->projector.dims != orc_env_obs_dims(spec->env) + 1) return NULL;
    if (spec->action_steps < 1 || spec->action_steps > ORC_MAX_ACTIONS) return NULL;
  }
  if (spec->projector.tilings > ORC_MAX_TILINGS) return NULL;

  orc_exp *e = (orc_exp *)calloc(1, sizeof(*e));
  if (!e) return NULL;
  e->spec = *spec;

  /* deployer.cpp:70-74 */
  orc_srand48(&e->G, seed);

  /* uniform.cpp:60-95: values = min + delta*k, delta = (max-min)/(steps-1), NaN -> 0 */
  e->A = spec->action_steps;
  {
    double range = spec->action_max - spec->action_min;
    double delta = range / ((double)spec->action_steps - 1);
    if (isnan(delta)) delta = 0.;
    for (int k = 0; k < e->A; ++k)
      e->actions[k] = spec->action_min + delta * k;
  }

  /* Instantiate order of the yaml (configurable.cpp:627-654): the representation's
   * configure()->reset() is the first RandGen::instance() call => `new Rand()`
   * seeds the thread-local stream from the global lrand48() (utils.h:90-93,160-171),
   * then draws memory*outputs uniforms. */
  orc_srand48(&e->TL, (long)orc_lrand48(&e->G));
  if (spec->agent == ORC_AGENT_AC)
  { /* cfg/cart_pole/ac_tc.yaml order: the policy's (actor) representation first, then the
     * critic's; both draw from the same thread-local stream; there are no samplers */
    e->w[1] = table_alloc_init(e, &spec->actor_projector, &spec->actor_representation);
    if (!e->w[1]) { free(e); return NULL; }
  }
  if (spec->safe != 0)
  { /* TileCodingProjector::configure (tile_coding.cpp:50-55): indices_ = -1 everywhere */
    if ((spec->safe != 1 && spec->safe != 2) || (spec->agent != ORC_AGENT_SARSA && spec->agent != ORC_AGENT_Q)) { free(e); return NULL; }
    e->claim = (int32_t *)malloc((size_t)spec->projector.memory * sizeof(int32_t));
    if (!e->claim) { free(e); return NULL; }
    memset(e->claim, 0xFF, (size_t)spec->projector.memory * sizeof(int32_t));
  }
  if (spec->target_interval > 0)
  { /* ParameterizedRepresentation::configure (representation.h:186-190) reinstantiates the representation as its
     * target BEFORE the object's own configure() continues: the target's reset() draws its memory*outputs uniforms
     * first, the main table's reset() draws the next ones and then synchronises (linear.cpp:117-122) */
    if (spec->agent != ORC_AGENT_SARSA && spec->agent != ORC_AGENT_Q) { free(e); return NULL; }
    e->wt = table_alloc_init(e, &spec->projector, &spec->representation);
    if (!e->wt) { free(e); return NULL; }
  }
  e->w[0] = table_alloc_init(e, &spec->projector, &spec->representation);
  if (!e->w[0]) { free(e->w[1]); free(e->wt); free(e); return NULL; }
  if (e->wt) { target_synchronize(e); e->syncs = 0; }
  if (spec->agent == ORC_AGENT_QV)
  { /* cfg/pendulum/qv_tc.yaml order: the policy's Q representation first, then the predictor's
     * v_representation; both draw from the same thread-local stream */
    e->w[1] = table_alloc_init(e, &spec->actor_projector, &spec->actor_representation);
    if (!e->w[1]) { free(e->w[0]); free(e); return NULL; }
  }
  e->ac_decay = 1;
  e->eps_decay = 1;
  trace_clear(&e->trace);
  if (spec->agent == ORC_AGENT_AC) return e;
  /* learning policy's sampler/epsilon_greedy, then the test policy's
   * sampler/greedy: each `new Rand()` (greedy.cpp:38-41) */
  orc_srand48(&e->S1, (long)orc_lrand48(&e->G));
  orc_srand48(&e->S2, (long)orc_lrand48(&e->G));

  e->eps_decay = 1;
  trace_clear(&e->trace);
  return e;
}

void orc_set_steps_budget(orc_exp *e, uint64_t steps) { e->steps_budget = steps; }
int64_t orc_trials(const orc_exp *e) { return e->tt; }

int orc_reset_run(orc_exp *e)
{ /* Experiment reset between runs (online_learning.cpp:307-308 -> Configurable::reset: {action: reset} walks the experiment's
   * subtree).  What the objects of this path do with it:
   *   representation/parameterized/linear (linear.cpp:104-125): every parameter is drawn again from the thread-local RandGen --
   *     the CONTINUING stream, nothing is reseeded -- in the order the representations stand in the yaml (the same order as at
   *     instantiation), followed by synchronize();
   *   predictor/critic/{sarsa,q,...} (sarsa.cpp:60-66, advantage.cpp:67, qv.cpp:71, td.cpp:64): finalize() = the trace is cleared;
   *   sampler/epsilon_greedy (greedy.cpp:140-141) and mapping/policy/action (action.cpp:93-97): decay_ = 1;
   *   sampler/greedy, the environment, the tile coding without a claim table: nothing (greedy.cpp:43-45).
   *   a target network (round 4): the walk reaches the target too -- ObjectConfigurator::instantiate adds every PROVIDED parameter that is
   *     an object as a child configurator (configurable.cpp:690-712: `target`), and ObjectConfigurator::reconfigure visits the children
   *     BEFORE the object itself (configurable.cpp:754-757) -- so the target draws its parameters again first, the representation next, and
   *     its synchronize() then blends the two fresh vectors: the sequence of construction (representation.h:186-190, linear.cpp:104-125),
   *     from the continuing stream; count_ = 0;
   *   projector/tile_coding with a claim table (tile_coding.cpp:82-89): every claim is dropped.
   * The run's own counters (ss, tt: loop variables of online_learning.cpp:154) start again. */
  const orc_spec *s = &e->spec;
  if (s->agent == ORC_AGENT_PID) return -1;
  if (e->claim) memset(e->claim, 0xFF, (size_t)s->projector.memory * sizeof(int32_t));
  if (e->wt)
  {
    const size_t n = (size_t)s->projector.memory;
    for (size_t i = 0; i < n; ++i)
      e->wt[i] = s->representation.init_min + orc_drand48(&e->TL) * (s->representation.init_max - s->representation.init_min);
  }
  const int first = (s->agent == ORC_AGENT_AC) ? 1 : 0, second = (s->agent == ORC_AGENT_AC) ? 0 : 1;
  const orc_tile_spec *ts[2] = {&s->projector, &s->actor_projector};
  const orc_linear_spec *ls[2] = {&s->representation, &s->actor_representation};
  const int order[2] = {first, second};
  for (int k = 0; k < 2; ++k)
  {
    const int t = order[k];
    if (!e->w[t]) continue;
    const size_t n = (size_t)ts[t]->memory;
    for (size_t i = 0; i < n; ++i)
      e->w[t][i] = ls[t]->init_min + orc_drand48(&e->TL) * (ls[t]->init_max - ls[t]->init_min);
  }
  if (e->wt) { target_synchronize(e); e->syncs = 0; }
  trace_clear(&e->trace);
  e->eps_decay = 1;
  e->ac_decay = 1;
  e->tt = 0;
  e->ss = 0;
  return 0;
}

void orc_destroy(orc_exp *e)
{
  if (!e) return;
  free(e->w[0]);
  free(e->w[1]);
  free(e->wt);
  free(e->claim);
  free(e);
}

/* ------------------------------------------------- per-step interfaces -- */
/* The experiment's environment and agents behind the reference's own plug-in interfaces, one call per call of
 * OnlineLearningExperiment::run (online_learning.cpp:172-213): orc_run below is written on top of them, so the golden
 * files pin these entry points too.  They are what the product's per-step C ABI (grlx_env_start / grlx_env_advance /
 * grlx_agent_start / _step / _end) is checked against, with one half of the loop on the host and the other on the GPU. */
void orc_exp_env_start(orc_exp *e, int test, double *obs)
{ /* Environment::start (environment.h:48) -> ModeledEnvironment::start, modeled.cpp:132-158 */
  orc_env_start(&e->spec, e, test, e->state);
  orc_env_observe(&e->spec, e->state, obs);
}

double orc_exp_env_step(orc_exp *e, double action, double *obs, double *reward, int *terminal)
{ /* Environment::step (environment.h:49-51) -> ModeledEnvironment::step, modeled.cpp:160-213 */
  return orc_env_step(&e->spec, e->state, action, obs, reward, terminal);
}

static void remember_action(orc_exp *e, const act_t *act)
{
  e->last_index = act->index;
  e->last_value = act->value;
  memcpy(e->last_q, act->q, sizeof(act->q));
}

double orc_exp_agent_start(orc_exp *e, int test, const double *obs)
{ /* Agent::start (agent.h:44-47) */
  const orc_spec *s = &e->spec;
  const int D = orc_env_obs_dims(s->env);
  act_t act;
  memset(&act, 0, sizeof(act));
  if (test)
  { /* fixed.cpp:47-51 */
    e->test_time = 0.;
    policy_act(e, 1, e->test_time, obs, &act);
  }
  else if (s->agent == ORC_AGENT_PID)
  { /* the learning agent is an agent/fixed too (`test_agent: ../agent`): fixed.cpp:47-51 */
    e->time = 0;
    policy_act(e, 0, e->time, obs, &act);
  }
  else
  { /* td.cpp:50-61 */
    if (s->agent != ORC_AGENT_AC)                            /* predictor_->finalize(), sarsa.cpp:126-132;          */
      trace_clear(&e->trace);                                /* ActionACPredictor::finalize (ac.cpp:170-173) does   */
                                                             /* NOT reach the critic: its trace survives episodes   */
    e->time = 0;
    policy_act(e, 0, e->time, obs, &act);
    memcpy(e->prev_obs, obs, sizeof(double) * (size_t)D);
    e->prev_action = act.value;
    e->prev_action_index = act.index;
  }
  remember_action(e, &act);
  return act.value;
}

static double predictor_update(orc_exp *e, double tau, double reward, const double *obs, int has_action, double action)
{ /* predictor_->update(Transition(prev_obs, prev_action, tau, reward, obs, action)), td.cpp:70 / :79 */
  const orc_spec *s = &e->spec;
  orc_proj *p = &e->last_p, *ap = &e->last_ap;
  if (s->agent == ORC_AGENT_SARSA)
    return sarsa_update(e, e->prev_obs, e->prev_action, tau, reward, obs, has_action, action, p);
  if (s->agent == ORC_AGENT_Q)
    return q_update(e, e->prev_obs, e->prev_action, tau, reward, obs, has_action, action, p);
  if (s->agent == ORC_AGENT_EXPECTED_SARSA)
    return expected_sarsa_update(e, e->prev_obs, e->prev_action, tau, reward, obs, has_action, p);
  if (s->agent == ORC_AGENT_ADVANTAGE)
    return advantage_update(e, e->prev_obs, e->prev_action, tau, reward, obs, has_action, p);
  if (s->agent == ORC_AGENT_QV)
    return qv_update(e, e->prev_obs, e->prev_action, tau, reward, obs, has_action, p, ap);
  return ac_update(e, e->prev_obs, e->prev_action, tau, reward, obs, has_action, p, ap);
}

double orc_exp_agent_step(orc_exp *e, int test, double tau, const double *obs, double reward)
{ /* Agent::step (agent.h:49-52) */
  const orc_spec *s = &e->spec;
  const int D = orc_env_obs_dims(s->env);
  act_t act;
  memset(&act, 0, sizeof(act));
  e->last_delta = 0;
  e->last_p.n = 0;
  e->last_ap.n = 0;
  if (test)
  { /* fixed.cpp:53-61 */
    e->test_time += tau;
    policy_act(e, 1, e->test_time, obs, &act);
  }
  else if (s->agent == ORC_AGENT_PID)
  { /* fixed.cpp:53-61 as the learning agent */
    e->time += tau;
    policy_act(e, 0, e->time, obs, &act);
  }
  else
  { /* td.cpp:63-74: act first, then update */
    e->time += tau;
    policy_act(e, 0, e->time, obs, &act);
    e->stats.trace_entries_sum += (uint64_t)e->trace.len;
    e->last_delta = predictor_update(e, tau, reward, obs, 1, act.value);
    memcpy(e->prev_obs, obs, sizeof(double) * (size_t)D);
    e->prev_action = act.value;
    e->prev_action_index = act.index;
  }
  remember_action(e, &act);
  return act.value;
}

void orc_exp_agent_end(orc_exp *e, int test, double tau, const double *obs, double reward)
{ /* Agent::end (agent.h:54-56): the transition into an absorbing state.  agent/fixed does nothing (fixed.cpp:63-65);
   * agent/td updates with an empty next action (td.cpp:76-81) */
  e->last_delta = 0;
  e->last_p.n = 0;
  e->last_ap.n = 0;
  if (test || e->spec.agent == ORC_AGENT_PID) return;
  e->last_delta = predictor_update(e, tau, reward, obs, 0, 0);
}

/* ------------------------------------------------------------------ run -- */
int orc_run(orc_exp *e, int n_trials, orc_row *rows, int max_rows,
            orc_tap *tap, int tap_cap, int *tap_n)
{ /* online_learning.cpp:154-280 */
  const orc_spec *s = &e->spec;
  int nrows = 0, ntap = 0;
  int D = orc_env_obs_dims(s->env);

  for (int t = 0; t < n_trials && !(e->steps_budget && (uint64_t)e->ss >= e->steps_budget); ++t, ++e->tt)
  { /* :154 `(!trials_ || tt < trials_) && (!steps_ || ss < steps_)` */
    int ti = s->test_interval;
    int test = (ti >= 0 && e->tt % (ti + 1) == ti);           /* :160 */
    double obs[ORC_MAX_DIMS], reward, total_reward = 0, total_time = 0;
    int terminal;
    const int subtrials = (test && s->test_trials > 1) ? s->test_trials : 1;     /* :161 */

    for (int st = 0; st < subtrials; ++st)
    { /* :170-222: every sub-trial starts the environment and the agent; reward and time keep adding up */
    orc_exp_env_start(e, test, obs);                            /* :172 */
    double action = orc_exp_agent_start(e, test, obs);          /* :178 */

    if (s->tap_starts && tap && ntap < tap_cap)
    { /* the row the transition log gets at the start of a trial (online_learning.cpp:183-184) */
      orc_tap *tp = &tap[ntap++];
      memset(tp, 0, sizeof(*tp));
      tp->test = test;
      tp->action_index = e->last_index;
      memcpy(tp->obs, obs, sizeof(double) * (size_t)D);
      tp->action = action;
      tp->terminal = -1;
      tp->trace_len = e->trace.len;
      memcpy(tp->state, e->state, sizeof(double) * (size_t)orc_env_state_dims(s->env));
      for (int a = 0; a < (s->agent == ORC_AGENT_AC ? 1 : e->A) && a < 8; ++a) tp->q[a] = e->last_q[a];
    }

    do
    {
      double tau = orc_exp_env_step(e, action, obs, &reward, &terminal);           /* :196 */
      total_reward += reward;                                                      /* :202 */
      total_time += tau;                                                           /* :203 */

      if (terminal == 2)
        orc_exp_agent_end(e, test, tau, obs, reward);                              /* :210-211 */
      else
        action = orc_exp_agent_step(e, test, tau, obs, reward);                    /* :212-213 */
      if (test)
        e->stats.test_steps++;
      else
      {
        e->ss++;                                                                   /* :218 */
        e->stats.learn_steps++;
      }

      if (tap && ntap < tap_cap)
      {
        orc_tap *tp = &tap[ntap++];
        memset(tp, 0, sizeof(*tp));
        tp->test = test;
        tp->action_index = e->last_index;
        memcpy(tp->obs, obs, sizeof(double) * (size_t)D);
        tp->action = e->last_value;
        tp->reward = reward;
        tp->terminal = terminal;
        tp->trace_len = e->trace.len;
        memcpy(tp->state, e->state, sizeof(double) * (size_t)orc_env_state_dims(s->env));
        for (int a = 0; a < (s->agent == ORC_AGENT_AC ? 1 : e->A) && a < 8; ++a) tp->q[a] = e->last_q[a];
        tp->delta = e->last_delta;
        for (int j = 0; j < e->last_p.n && j < 16; ++j) tp->p_idx[j] = (uint32_t)e->last_p.idx[j];
        for (int j = 0; j < e->last_ap.n && j < 16; ++j) tp->p_idx[16 + j] = (uint32_t)e->last_ap.idx[j];   /* actor projection (AC) */
      }
    } while (!terminal);
    } /* sub-trials */
    total_reward /= subtrials;                                  /* :224-225 */
    total_time /= subtrials;

    if (ti >= 0 ? test : 1)
    { /* :238-262 */
      if (rows && nrows < max_rows)
      {
        rows[nrows].trial = (ti >= 0) ? (e->tt + 1 - (e->tt + 1) / (ti + 1)) : e->tt;
        rows[nrows].steps = e->ss;
        rows[nrows].reward = total_reward;
        rows[nrows].time = total_time;
        nrows++;
      }
    }
  }
  if (tap_n) *tap_n = ntap;
  return nrows;
}

/* ------------------------------------------------------------ accessors -- */
void orc_get_stats(const orc_exp *e, orc_stats *out) { *out = e->stats; }
const double *orc_weights(const orc_exp *e, int table) { return (table == 0 || table == 1) ? e->w[table] : (table == 2 ? e->wt : NULL); }
int64_t orc_target_syncs(const orc_exp *e) { return e->syncs; }

/* ParameterizedRepresentation {action: load} (representation.h:231-263): setParams() overwrites
 * every weight of the table; nothing else of the experiment changes. */
int orc_set_weights(orc_exp *e, int table, const double *w, size_t n)
{
  const size_t mem = (size_t)(table == 1 ? e->spec.actor_projector.memory : e->spec.projector.memory);
  if ((table != 0 && table != 1) || !e->w[table] || !w || n != mem) return -1;
  memcpy(e->w[table], w, n * sizeof(double));
  if (table == 0 && e->wt) target_synchronize(e);      /* {action: load}: setParams(p); synchronize() (representation.h:256-257) */
  return 0;
}
void orc_set_state(orc_exp *e, const double *state) { memcpy(e->state, state, sizeof(double) * (size_t)orc_env_state_dims(e->spec.env)); }
void orc_get_state(const orc_exp *e, double *state) { memcpy(state, e->state, sizeof(double) * (size_t)orc_env_state_dims(e->spec.env)); }
void orc_rng_states(const orc_exp *e, uint64_t out[4]) { out[0] = e->G.x; out[1] = e->TL.x; out[2] = e->S1.x; out[3] = e->S2.x; }

int orc_format_row(const orc_row *r, char *buf, size_t cap)
{ /* the golden file's layout: three setw(15) columns at the default ostream
   * precision (6 significant digits, %g), tests/template/pendulum-sarsa-tc-0.txt */
  return snprintf(buf, cap, "%15lld%15lld%15g\n", (long long)r->trial, (long long)r->steps, r->reward);
}
