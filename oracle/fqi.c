/* fqi.c -- the batch path: BatchLearningExperiment + FQIPredictor + IterativeRepresentation + ANNRepresentation,
 * restated (TEST INFRASTRUCTURE, see oracle.h).  BASELINE.json configs[4], SURVEY.md 8(f-3).
 *
 *   BatchLearningExperiment::run          base/src/experiments/batch_learning.cpp:88-205
 *   FQIPredictor::{update,finalize,rebuild} base/src/predictors/fqi.cpp:187-285
 *   IterativeRepresentation::{write,finalize} base/src/representations/iterative.cpp:63-107
 *   ANNRepresentation::{read,write,finalize,backprop} base/src/representations/ann.cpp:133-263, ann.h:108-117
 *   NormalizingProjector::project         base/src/projectors/normalizing.cpp:78-87 (projector/identity downstream)
 *   QPolicy::act + GreedySampler          base/src/policies/q.cpp:143-155, base/src/samplers/greedy.cpp:47-86
 *   PendulumSwingupTask::invert           base/src/environments/pendulum.cpp:147-155
 *
 * PARITY UNPINNED.  The reference's only fixture for this path, tests/template/pendulum-fqi-ann-0.txt, holds two rows
 * with the return -3508.07 = the return of a CONSTANT action +-3 over a 100-step episode: it pins the test-trial
 * plumbing (tests/test_oracle_fqi.py reproduces that number), not the learning.  Nothing can pin the learning:
 *  D1  initial weights are Vector::Random(sz)*0.01 (ann.cpp:97) = Eigen's wrapper over libc rand(); the Eigen version
 *      and its draw order are not part of /root/reference.  HERE: w[i] = 0.01 * (2*u_i - 1), u_i the i-th draw of a
 *      48-bit LCG stream seeded srand48_r(seed) (its own stream: no other consumer).
 *  D2  the hidden-layer delta is written (W*delta).topRows(layers_[ii+1].size) (ann.cpp:249): for a 3-20-1 network
 *      that keeps ONE row of a 21-row product and multiplies it element-wise with a 20-row vector -- mismatched
 *      Eigen dimensions, unchecked in the reference's RelWithDebInfo build (NDEBUG), i.e. undefined behaviour (and
 *      the reason its template shows an untrained policy).  HERE: the evident intent, topRows(layers_[ii].size):
 *      delta_h = W2[h] * delta_out * a_h (1 - a_h).
 *  D3  summation orders that Eigen leaves to its kernels are fixed: a layer's net input is
 *      ((w_0 a_0 + w_1 a_1) + ...) + bias; the gradient of an epoch is summed over samples either in sample order
 *      (sum_order 0: what ann.cpp:252-253 does) or by the fixed two-level tree the GPU uses (sum_order 1: chunks of
 *      64 samples in sample order, chunk sums strided over 64 lanes and reduced by a wavefront shuffle; see
 *      ann_epoch).  RPROP (eta = 0, ann.cpp:207-213) looks at SIGNS only, so the two orders give the same weights
 *      unless a component of the summed gradient is within rounding of zero; gradient descent (eta > 0, :202-206) and
 *      RMSprop (eta < 0, :214-219) use the sum's value, so there the orders differ by rounding and tests compare the GPU
 *      with sum_order 1 only.
 *  D4  tests/pendulum-fqi-ann.yaml gives input_min as "observation_min+action_min": with the current parser `+`
 *      is an element-wise sum (parser.cpp:49-134), which would make the projector reject its 3-dimensional input
 *      (normalizing.cpp:80-84).  HERE: the role default of normalizing.cpp:50-51, the concatenation (`++`).
 *  D5  (portable arithmetic only) the logistic's argument is clamped to [-690, 690] before exp: 1 / (1 + exp(-net)) is within 1e-299 of 0 or
 *      of 1 beyond it, and inside it exp neither overflows nor leaves the normal range, which lets the GPU evaluate it without range
 *      handling (grlx_math_batch.h: plogistic_batch).  With libm arithmetic (the reference's own formula) nothing is clamped; the
 *      cross-mode test (tests/test_oracle_fqi.py) holds both to 1e-5 relative.
 *  pow(gamma, tau) with tau = control_step (DynamicalModel::step returns tau_, modeled.cpp:275; the batch path has no
 *  discrete_time) is evaluated ONCE with libm by the caller and passed in as gamma_tau in both math modes.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

#define FQI_MAX_HIDDEN 64
#define FQI_MAX_IN 8

struct orc_fqi {
  orc_fqi_spec spec;
  orc_exp     *core;              /* spec + RNG streams (G, TL) + env state, as the online oracle keeps them */
  orc_rand48   R;                 /* D1: weight-initialisation stream */
  int          n_in, H, n_params;
  double      *params, *eta, *Delta, *prev_Delta;   /* layer 1: (n_in+1) x H column-major, then layer 2: (H+1) x 1 */
  /* transition store (FQIPredictor::transitions_) */
  size_t       n, cap;
  double      *in;                /* [n][n_in]  normalised (prev_obs, prev_action)   */
  double      *next_obs;          /* [n][D]                                          */
  double      *reward;            /* [n]                                             */
  int         *absorbing;         /* [n]  terminal == 2                              */
  double      *targets;           /* [n]  of the last iteration                      */
  double       in_min[FQI_MAX_IN], in_scale[FQI_MAX_IN];
  int64_t      batches_done;
  double       last_maxdelta;
  int          last_iterations;
  double       last_error;        /* error_/samples_ of the last epoch (ann.cpp:201) */
};

/* exp(x) for the logistic's clamped argument, |x| <= 690 (D5): the reduction and scaling of orc_pexp (portable_math.c), the kernel
 * 1 + r + r^2 q(r) with q of degree 9 instead of orc_pexp's Taylor polynomial of degree 11 -- a Chebyshev fit of (e^r - 1 - r) / r^2 on
 * |r| <= ln2/2 (mpmath, 200 bits), 0.07 ulp from exp before rounding; the result is normal, so the two-step scaling is exact.  GPU:
 * plogistic_batch, grlx_math_batch.h (same operations, same order). */
double orc_logistic_exp(double x)
{
  static const double c[10] = {0x1.af389ecfc4b9cp-26, 0x1.28917c89a43a7p-22, 0x1.71de0db2f6b19p-19, 0x1.a019b9149a41cp-16, 0x1.a01a01a7c2efep-13,
                               0x1.6c16c17889ef1p-10, 0x1.11111111109b5p-7, 0x1.5555555553d68p-5, 0x1.5555555555556p-3, 0x1.0000000000001p-1};
  const double kd = rint(x * 0x1.71547652b82fep+0);
  double r = fma(-kd, 0x1.62e4200000000p-1, x);
  r = fma(-kd, 0x1.fdf473de6af28p-22, r);
  double q = fma(r, c[0], c[1]);
  for (int s = 2; s < 10; ++s) q = fma(r, q, c[s]);
  const double e = 1.0 + fma(r * r, q, r);
  return ldexp(e, (int)kd);                                         /* exact: the result is normal */
}
static double fq_exp(const orc_fqi *f, double x) { return f->spec.base.math == ORC_MATH_PORTABLE ? orc_logistic_exp(x) : exp(x); }

/* ANNRepresentation::read (ann.cpp:133-160) for one normalised input; hidden activations to a[] when given */
static double ann_forward(const orc_fqi *f, const double *in, double *a_out)
{
  const int n_in = f->n_in, H = f->H;
  const double *W1 = f->params, *W2 = f->params + (size_t)(n_in + 1) * H;
  double out = 0;
  for (int h = 0; h < H; ++h)
  {
    double net = 0;
    for (int i = 0; i < n_in; ++i) net += W1[(size_t)h * (n_in + 1) + i] * in[i];
    net += W1[(size_t)h * (n_in + 1) + n_in];                     /* bias row */
    double x = -net;
    if (f->spec.base.math == ORC_MATH_PORTABLE) x = fmin(fmax(x, -690.), 690.);                      /* D5 (a NaN becomes -690) */
    const double a = 1. / (1. + fq_exp(f, x));                    /* ann.h:108-111 */
    if (a_out) a_out[h] = a;
    out += W2[h] * a;
  }
  out += W2[H];
  return out;                                                     /* linear output neuron */
}

static void normalise(const orc_fqi *f, const double *obs, double action, double *in)
{ /* NormalizingProjector::project (normalizing.cpp:78-87), signed = 0, over (obs ++ action) */
  const int D = f->n_in - 1;
  for (int i = 0; i < D; ++i) in[i] = (obs[i] - f->in_min[i]) * f->in_scale[i] - 0;
  in[D] = (action - f->in_min[D]) * f->in_scale[D] - 0;
}

void orc_fqi_spec_pendulum(orc_fqi_spec *s)
{ /* the reference's tests/pendulum-fqi-ann.yaml */
  memset(s, 0, sizeof(*s));
  orc_spec_pendulum_sarsa(&s->base);
  s->base.gamma = 0.97;
  s->batch_size = 1000;
  s->iterations = 10;
  s->epochs = 500;
  s->hidden = 20;
  s->sum_order = 0;
  s->gamma_tau = pow(0.97, 0.03);
}

orc_fqi *orc_fqi_create(const orc_fqi_spec *spec, long seed)
{
  if (spec->base.env != ORC_ENV_PENDULUM) return NULL;              /* the task must support invert() */
  if (spec->hidden < 1 || spec->hidden > FQI_MAX_HIDDEN || spec->batch_size < 1 || spec->base.action_steps < 1) return NULL;
  orc_fqi *f = (orc_fqi *)calloc(1, sizeof(*f));
  if (!f) return NULL;
  f->spec = *spec;
  f->core = (orc_exp *)calloc(1, sizeof(orc_exp));
  if (!f->core) { free(f); return NULL; }
  f->core->spec = spec->base;
  const int D = orc_env_obs_dims(spec->base.env);
  f->n_in = D + 1;
  f->H = spec->hidden;
  f->n_params = (f->n_in + 1) * f->H + (f->H + 1);
  f->params = (double *)calloc((size_t)f->n_params * 4, sizeof(double));
  if (!f->params) { free(f->core); free(f); return NULL; }
  f->eta = f->params + f->n_params;
  f->Delta = f->eta + f->n_params;
  f->prev_Delta = f->Delta + f->n_params;
  /* observation / action ranges of task/pendulum/swingup (pendulum.cpp:84-88) */
  const double omin[2] = {0., -12 * M_PI}, omax[2] = {2 * M_PI, 12 * M_PI};
  for (int i = 0; i < D; ++i) { f->in_min[i] = omin[i]; f->in_scale[i] = 1. / (omax[i] - omin[i]) * (1 + 0); }
  f->in_min[D] = spec->base.action_min;
  f->in_scale[D] = 1. / (spec->base.action_max - spec->base.action_min) * (1 + 0);
  /* uniform.cpp:60-95 */
  f->core->A = spec->base.action_steps;
  {
    double range = spec->base.action_max - spec->base.action_min;
    double delta = range / ((double)spec->base.action_steps - 1);
    if (isnan(delta)) delta = 0.;
    for (int k = 0; k < f->core->A; ++k) f->core->actions[k] = spec->base.action_min + delta * k;
  }
  /* Instantiate order of the yaml: ... predictor { representation/parameterized/ann: reset() } ...
   * test_agent { policy { sampler/greedy: new Rand() = global lrand48 #1 (greedy.cpp:38-41) } }; the thread-local
   * RandGen is created by the first RandGen::getVector of run() (batch_learning.cpp:109): global lrand48 #2. */
  orc_srand48(&f->core->G, seed);
  orc_srand48(&f->core->S2, (long)orc_lrand48(&f->core->G));
  orc_srand48(&f->core->TL, (long)orc_lrand48(&f->core->G));
  /* ANNRepresentation reset (ann.cpp:92-120) with D1's stream; eta = Ones, x 0.1 for RPROP (ann.cpp:106-109) */
  orc_srand48(&f->R, seed);
  for (int i = 0; i < f->n_params; ++i)
  {
    f->params[i] = (2 * orc_drand48(&f->R) - 1) * 0.01;
    f->eta[i] = (spec->eta == 0) ? 1. * 0.1 : 1.;
    f->Delta[i] = 0;
    f->prev_Delta[i] = 0;
  }
  return f;
}

void orc_fqi_destroy(orc_fqi *f)
{
  if (!f) return;
  free(f->in); free(f->next_obs); free(f->reward); free(f->absorbing); free(f->targets);
  free(f->params);
  free(f->core);
  free(f);
}

/* per-sample gradient contribution g[n_params] (ANNRepresentation::backprop, ann.cpp:224-263, D2) */
static double ann_sample_gradient(const orc_fqi *f, const double *in, double target, double *g)
{
  const int n_in = f->n_in, H = f->H;
  const double *W2 = f->params + (size_t)(n_in + 1) * H;
  double a[FQI_MAX_HIDDEN];
  const double out = ann_forward(f, in, a);
  const double d2 = out - target;                                 /* linear output: delta = activation - out */
  double *g1 = g, *g2 = g + (size_t)(n_in + 1) * H;
  for (int h = 0; h < H; ++h)
  {
    const double d1 = (W2[h] * d2) * (a[h] * (1. - a[h]));       /* ann.cpp:249 with D2; dactivate ann.h:114-117 */
    for (int i = 0; i < n_in; ++i) g1[(size_t)h * (n_in + 1) + i] = in[i] * d1;
    g1[(size_t)h * (n_in + 1) + n_in] = d1;
    g2[h] = a[h] * d2;
  }
  g2[H] = d2;
  return d2 * d2;                                                 /* error_ (ann.cpp:240) */
}

/* one epoch: IterativeRepresentation::finalize's inner loop (iterative.cpp:85-92) + ANNRepresentation::finalize (RPROP) */
static void ann_epoch(orc_fqi *f)
{
  const int P = f->n_params;
  double err = 0;
  double g[(FQI_MAX_IN + 1) * FQI_MAX_HIDDEN + FQI_MAX_HIDDEN + 2];
  if (f->spec.sum_order == 0)
  { /* the reference's order: Delta += per sample, in sample order (ann.cpp:252-253) */
    for (size_t s = 0; s < f->n; ++s)
    {
      err += ann_sample_gradient(f, f->in + s * (size_t)f->n_in, f->targets[s], g);
      for (int k = 0; k < P; ++k) f->Delta[k] += g[k];
    }
  }
  else
  { /* D3: the GPU's fixed tree.  Level 1: a chunk = 64 consecutive samples, summed in sample order from +0.0 (one lane
     * of a wavefront per parameter); level 2: the chunk sums c_0 .. c_{C-1}: lane l of 64 adds c_l, c_{l+64}, ... in
     * order from +0.0, then the 64 lane sums are reduced by v[i] += v[i+off], off = 32 .. 1 (a wavefront's shuffle
     * reduction).  Chunks beyond the store contribute +0.0. */
    static double lanes[64][(FQI_MAX_IN + 1) * FQI_MAX_HIDDEN + FQI_MAX_HIDDEN + 2];
    double chunk[(FQI_MAX_IN + 1) * FQI_MAX_HIDDEN + FQI_MAX_HIDDEN + 2];
    for (int l = 0; l < 64; ++l)
      for (int k = 0; k <= P; ++k) lanes[l][k] = 0.;
    size_t ci = 0;
    for (size_t c0 = 0; c0 < f->n; c0 += 64, ++ci)
    {
      for (int k = 0; k <= P; ++k) chunk[k] = 0.;
      for (size_t sidx = c0; sidx < c0 + 64 && sidx < f->n; ++sidx)
      {
        g[P] = ann_sample_gradient(f, f->in + sidx * (size_t)f->n_in, f->targets[sidx], g);
        for (int k = 0; k <= P; ++k) chunk[k] += g[k];
      }
      for (int k = 0; k <= P; ++k) lanes[ci % 64][k] += chunk[k];
    }
    for (int off = 32; off > 0; off >>= 1)
      for (int l = 0; l < off; ++l)
        for (int k = 0; k <= P; ++k) lanes[l][k] += lanes[l + off][k];
    for (int k = 0; k < P; ++k) f->Delta[k] += lanes[0][k];
    err += lanes[0][P];
  }
  f->last_error = err / (double)f->n;
  /* ANNRepresentation::finalize (ann.cpp:198-221), element-wise; samples_ = the backprop calls of the epoch = n; then Delta = 0 */
  const double eta_ = f->spec.eta, samples = (double)f->n;
  for (int k = 0; k < P; ++k)
  {
    if (eta_ > 0)
    { /* stochastic gradient descent (:202-206): W -= eta_*Delta/samples_, evaluated left to right */
      f->params[k] -= (eta_ * f->Delta[k]) / samples;
    }
    else if (eta_ == 0)
    { /* RPROP (:207-213) */
      f->eta[k] = (f->Delta[k] * f->prev_Delta[k] > 0) ? f->eta[k] * 1.2 : f->eta[k] * 0.5;
      f->params[k] -= (f->Delta[k] > 0) ? f->eta[k] : -f->eta[k];
      f->prev_Delta[k] = f->Delta[k];
    }
    else
    { /* RMSprop (:214-219): eta = 0.9 eta + 0.1 (Delta/samples_)^2;  W += eta_ * Delta / sqrt(eta)  (IEEE sqrt and division) */
      const double g = f->Delta[k] / samples;
      f->eta[k] = 0.9 * f->eta[k] + 0.1 * (g * g);
      f->params[k] += eta_ * (f->Delta[k] / sqrt(f->eta[k]));
    }
    f->Delta[k] = 0;
  }
}

static int reserve(orc_fqi *f, size_t n)
{
  if (n <= f->cap) return 0;
  size_t cap = f->cap ? f->cap : 1024;
  while (cap < n) cap *= 2;
  const int D = f->n_in - 1;
  double *a = (double *)realloc(f->in, cap * (size_t)f->n_in * sizeof(double));
  if (!a) return -1;
  f->in = a;
  a = (double *)realloc(f->next_obs, cap * (size_t)D * sizeof(double));
  if (!a) return -1;
  f->next_obs = a;
  a = (double *)realloc(f->reward, cap * sizeof(double));
  if (!a) return -1;
  f->reward = a;
  a = (double *)realloc(f->targets, cap * sizeof(double));
  if (!a) return -1;
  f->targets = a;
  int *b = (int *)realloc(f->absorbing, cap * sizeof(int));
  if (!b) return -1;
  f->absorbing = b;
  f->cap = cap;
  return 0;
}

/* FQIPredictor::rebuild (fqi.cpp:205-285), reset_strategy never, projector lifetime: permanent */
static void fqi_rebuild(orc_fqi *f)
{
  const orc_fqi_spec *sp = &f->spec;
  const int D = f->n_in - 1;
  double maxdelta = INFINITY;
  int ii;
  for (size_t s = 0; s < f->n; ++s) f->targets[s] = 0.;            /* std::vector<double> targets(n, 0.) */
  for (ii = 0; ii < sp->iterations && maxdelta > 0.001; ++ii)
  {
    maxdelta = 0;
    for (size_t s = 0; s < f->n; ++s)
    {
      double target = f->reward[s];
      if (!f->absorbing[s])
      {
        double v = -INFINITY, in[FQI_MAX_IN];
        for (int k = 0; k < f->core->A; ++k)
        {
          normalise(f, f->next_obs + s * (size_t)D, f->core->actions[k], in);
          v = fmax(v, ann_forward(f, in, NULL));
        }
        target += sp->gamma_tau * v;                                /* pow(gamma_, tau) */
      }
      maxdelta = fmax(maxdelta, fabs(f->targets[s] - target));
      f->targets[s] = target;
    }
    for (int e = 0; e < sp->epochs; ++e) ann_epoch(f);              /* write all samples, then finalize(): iterative.cpp:76-98 */
  }
  f->last_maxdelta = maxdelta;
  f->last_iterations = ii;
}

int orc_fqi_run_batch(orc_fqi *f, orc_row *row)
{ /* one pass of the `bb` loop of BatchLearningExperiment::run (batch_learning.cpp:105-188) */
  const orc_fqi_spec *sp = &f->spec;
  const orc_spec *es = &f->core->spec;
  const int D = f->n_in - 1;
  if (reserve(f, f->n + (size_t)sp->batch_size) != 0) return -1;
  for (int ss = 0; ss < sp->batch_size; ++ss)
  {
    double obs[ORC_MAX_DIMS], state[ORC_MAX_STATE], nobs[ORC_MAX_DIMS], reward;
    int terminal;
    const double omin[2] = {0., -12 * M_PI}, omax[2] = {2 * M_PI, 12 * M_PI};
    for (int i = 0; i < D; ++i) obs[i] = 0 + orc_drand48(&f->core->TL) * (1 - 0);          /* RandGen::getVector */
    double action = 0 + orc_drand48(&f->core->TL) * (1 - 0);
    double next_action = 0 + orc_drand48(&f->core->TL) * (1 - 0);
    (void)next_action;                                              /* drawn, scaled, never read by FQI */
    for (int i = 0; i < D; ++i) obs[i] = omin[i] + obs[i] * (omax[i] - omin[i]);
    action = es->action_min + action * (es->action_max - es->action_min);
    /* task_->invert (pendulum.cpp:147-155), time = 0 */
    state[0] = obs[0] - M_PI;
    state[1] = obs[1];
    state[2] = 0.;
    /* model_->step + observe + evaluate: ModeledEnvironment::step's arithmetic without the task's actuate() clamp --
     * the action is inside [action_min, action_max], where PendulumSwingupTask::actuate is the identity */
    orc_env_step(es, state, action, nobs, &reward, &terminal);
    const size_t s = f->n++;
    normalise(f, obs, action, f->in + s * (size_t)f->n_in);
    memcpy(f->next_obs + s * (size_t)D, nobs, sizeof(double) * (size_t)D);
    f->reward[s] = reward;
    f->absorbing[s] = terminal == 2;
  }
  fqi_rebuild(f);                                                   /* predictor_->finalize(), macro_batch_size 1 */

  /* test trial (batch_learning.cpp:141-176): agent/fixed + policy/discrete/q + sampler/greedy */
  double obs[ORC_MAX_DIMS], reward, total_reward = 0;
  int terminal = 0, steps = 0;
  orc_env_start(es, f->core, 1, f->core->state);
  orc_env_observe(es, f->core->state, obs);
  for (;;)
  {
    double q[ORC_MAX_ACTIONS], in[FQI_MAX_IN];
    for (int k = 0; k < f->core->A; ++k)
    {
      normalise(f, obs, f->core->actions[k], in);
      q[k] = ann_forward(f, in, NULL);
    }
    /* GreedySampler::sample (greedy.cpp:47-86): first maximum, ties broken by lrand48() % ties on the GLOBAL stream */
    int mai = 0, man = 1;
    for (int k = 1; k < f->core->A; ++k)
    {
      if (q[k] > q[mai]) { mai = k; man = 1; }
      else if (q[k] == q[mai]) man++;
    }
    if (man > 1)
    {
      int jj = (int)(orc_lrand48(&f->core->G) % (uint32_t)man);
      for (int k = 0; k < f->core->A; ++k)
        if (q[k] == q[mai])
        {
          if (jj == 0) { mai = k; break; }
          --jj;
        }
    }
    if (terminal) break;                                            /* the action chosen after the last step is never applied */
    orc_env_step(es, f->core->state, f->core->actions[mai], obs, &reward, &terminal);
    total_reward += reward;
    steps++;
    if (terminal == 2) break;                                       /* agent->end: no further action */
  }
  if (row)
  { /* batch_learning.cpp:179: setw(15) bb, bb*batch_size_, total_reward */
    row->trial = f->batches_done;
    row->steps = f->batches_done * (int64_t)sp->batch_size;
    row->reward = total_reward;
    row->time = steps;
  }
  f->batches_done++;
  return 0;
}

const double *orc_fqi_params(const orc_fqi *f, int *n) { if (n) *n = f->n_params; return f->params; }
size_t orc_fqi_transitions(const orc_fqi *f, const double **in, const double **next_obs, const double **reward, const double **targets)
{
  if (in) *in = f->in;
  if (next_obs) *next_obs = f->next_obs;
  if (reward) *reward = f->reward;
  if (targets) *targets = f->targets;
  return f->n;
}
void orc_fqi_info(const orc_fqi *f, double *maxdelta, int *iterations, double *error)
{
  if (maxdelta) *maxdelta = f->last_maxdelta;
  if (iterations) *iterations = f->last_iterations;
  if (error) *error = f->last_error;
}
double orc_fqi_q(const orc_fqi *f, const double *obs, double action)
{
  double in[FQI_MAX_IN];
  normalise(f, obs, action, in);
  return ann_forward(f, in, NULL);
}
void orc_fqi_rng(const orc_fqi *f, uint64_t out[3]) { out[0] = f->core->G.x; out[1] = f->core->TL.x; out[2] = f->R.x; }
