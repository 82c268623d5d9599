/* envs.c -- dynamical-system environments, restated (TEST INFRASTRUCTURE).
 *
 * ModeledEnvironment + DynamicalModel (RK4): base/src/environments/modeled.cpp:132-276
 * Pendulum dynamics + swing-up task:          base/src/environments/pendulum.cpp:40-145
 * (cart-pole, acrobot, compass walker are added by later sections of this file.)
 *
 * All arithmetic is IEEE double in the reference's expression order with no
 * fused multiply-add (compile with -ffp-contract=off).
 */
#include <math.h>
#include <string.h>
#include "oracle_internal.h"

/* ------------------------------------------------------------ math mode -- */
double orc_m_sin(const orc_spec *s, double x) { return s->math == ORC_MATH_PORTABLE ? orc_psin(x) : sin(x); }
double orc_m_cos(const orc_spec *s, double x) { return s->math == ORC_MATH_PORTABLE ? orc_pcos(x) : cos(x); }
double orc_m_log(const orc_spec *s, double x) { return s->math == ORC_MATH_PORTABLE ? orc_plog(x) : log(x); }
/* pow(x, 2) as the reference writes it; the portable mode defines it as x*x */
double orc_m_sqr(const orc_spec *s, double x) { return s->math == ORC_MATH_PORTABLE ? x * x : pow(x, 2); }
/* pow(base, tau): the portable mode supports the discrete-time case tau == 1 */
double orc_m_powtau(const orc_spec *s, double base, double tau)
{
  if (s->math == ORC_MATH_PORTABLE && tau == 1.0)
    return base;
  return pow(base, tau);
}

int orc_env_state_dims(int env)
{
  switch (env)
  {
    case ORC_ENV_PENDULUM: return 3;
    default: return -1;
  }
}

int orc_env_obs_dims(int env)
{
  switch (env)
  {
    case ORC_ENV_PENDULUM: return 2;
    default: return -1;
  }
}

/* ------------------------------------------------------------- pendulum -- */
static void pendulum_eom(const orc_spec *s, const double *x, double u, double *xd)
{ /* pendulum.cpp:40-49 constants, :55-68 equation */
  const double J = 0.000191, m = 0.055, g = 9.81, l = 0.042, b = 0.000003, K = 0.0536, R = 9.5;
  double a = x[0], ad = x[1];
  double add = (1 / J) * (m * g * l * orc_m_sin(s, a) - b * ad - (K * K / R) * ad + (K / R) * u);
  xd[0] = ad;
  xd[1] = add;
  xd[2] = 1;
}

static void pendulum_start(const orc_spec *s, orc_exp *e, int test, double *x)
{ /* pendulum.cpp:97-103: RandGen::get() is evaluated every episode */
  double r = orc_drand48(&e->TL);
  x[0] = M_PI + s->randomization * (test == 0) * r * 2 * M_PI;
  x[1] = 0;
  x[2] = 0;
}

static int pendulum_observe(const orc_spec *s, const double *x, double *obs)
{ /* pendulum.cpp:111-129 */
  double a = fmod(x[0] + M_PI, 2 * M_PI);
  if (a < 0) a += 2 * M_PI;
  obs[0] = a;
  obs[1] = x[1];
  return x[2] > s->timeout ? 1 : 0;
}

static double pendulum_evaluate(const orc_spec *s, const double *x, double action, const double *next)
{ /* pendulum.cpp:131-145 */
  double a = fmod(fabs(next[0]), 2 * M_PI);
  if (a > M_PI) a -= 2 * M_PI;
  double reward = -5 * orc_m_sqr(s, a) - 0.1 * orc_m_sqr(s, next[1]) - 1 * orc_m_sqr(s, action);
  if ((next[2] - x[2]) != 1)
    reward *= (next[2] - x[2]) / 0.03;
  return reward;
}

/* ------------------------------------------------------ generic dispatch -- */
static void env_eom(const orc_spec *s, const double *x, double u, double *xd)
{
  switch (s->env)
  {
    case ORC_ENV_PENDULUM: pendulum_eom(s, x, u, xd); break;
  }
}

static double env_actuate(const orc_spec *s, double action)
{
  switch (s->env)
  {
    case ORC_ENV_PENDULUM: return fmin(fmax(action, -3), 3);   /* pendulum.cpp:105-109 */
  }
  return action;
}

void orc_env_start(const orc_spec *s, orc_exp *e, int test, double *x)
{
  switch (s->env)
  {
    case ORC_ENV_PENDULUM: pendulum_start(s, e, test, x); break;
  }
}

int orc_env_observe(const orc_spec *s, const double *x, double *obs)
{
  switch (s->env)
  {
    case ORC_ENV_PENDULUM: return pendulum_observe(s, x, obs);
  }
  return 0;
}

/* DynamicalModel::step, modeled.cpp:254-276: `steps` classical RK4 sub-steps of
 * h = tau/steps; next += (k1 + 2*k2 + 2*k3 + k4)/6 with a true division. */
static void rk4_step(const orc_spec *s, const double *x, double u, double *next)
{
  int S = orc_env_state_dims(s->env);
  double h = s->control_step / s->integration_steps;
  double xd[ORC_MAX_STATE], k1[ORC_MAX_STATE], k2[ORC_MAX_STATE], k3[ORC_MAX_STATE], k4[ORC_MAX_STATE], t[ORC_MAX_STATE];

  memcpy(next, x, sizeof(double) * S);
  for (int ii = 0; ii < s->integration_steps; ++ii)
  {
    env_eom(s, next, u, xd);
    for (int i = 0; i < S; ++i) { k1[i] = h * xd[i]; t[i] = next[i] + k1[i] / 2; }
    env_eom(s, t, u, xd);
    for (int i = 0; i < S; ++i) { k2[i] = h * xd[i]; t[i] = next[i] + k2[i] / 2; }
    env_eom(s, t, u, xd);
    for (int i = 0; i < S; ++i) { k3[i] = h * xd[i]; t[i] = next[i] + k3[i]; }
    env_eom(s, t, u, xd);
    for (int i = 0; i < S; ++i)
    {
      k4[i] = h * xd[i];
      next[i] = next[i] + (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) / 6;
    }
  }
}

/* ModeledEnvironment::step, modeled.cpp:160-213 (window 1, no delta, no
 * exporter, discrete_time 1 => returns 1). */
double orc_env_step(const orc_spec *s, double *state, double action,
                    double *obs, double *reward, int *terminal)
{
  double next[ORC_MAX_STATE];
  int S = orc_env_state_dims(s->env);
  double actuation = env_actuate(s, action);

  rk4_step(s, state, actuation, next);
  *terminal = orc_env_observe(s, next, obs);
  switch (s->env)
  {
    case ORC_ENV_PENDULUM: *reward = pendulum_evaluate(s, state, action, next); break;
    default: *reward = 0;
  }
  memcpy(state, next, sizeof(double) * S);
  return 1;
}
