/* envs.c -- dynamical-system environments, restated (TEST INFRASTRUCTURE).
 *
 * ModeledEnvironment + DynamicalModel (RK4): base/src/environments/modeled.cpp:132-276
 * Pendulum dynamics + swing-up task:          base/src/environments/pendulum.cpp:40-145
 * Acrobot dynamics + balancing task:         base/src/environments/acrobot.cpp:48-151
 *   (no reference test pins the acrobot: parity unpinned by reference tests)
 * Cart-pole dynamics + swing-up task:        base/src/environments/cart_pole.cpp:58-237
 *   (the swing-up TASK is unpinned; the DYNAMICS -- end_stop = 1, incl. the :65 quirk -- and DynamicalModel::step
 *   are pinned by tests/template/cart_pole_balancing-pid-0.txt through the balancing task below)
 * Cart-pole balancing task:                  base/src/environments/cart_pole.cpp:239-320
 * Compass walker model + walk task:          base/src/environments/compass_walker/SWModel.cpp:15-258,
 *   base/include/grl/environments/compass_walker/SWModel.h:40-59, compass_walker.cpp:41-95, 198-344
 *   (parity unpinned by reference tests)
 *
 * All arithmetic is IEEE double in the reference's expression order with no
 * fused multiply-add (compile with -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
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
    case ORC_ENV_ACROBOT: return 5;
    case ORC_ENV_CART_POLE: return 5;
    case ORC_ENV_CART_POLE_BALANCING: return 5;
    case ORC_ENV_COMPASS_WALKER: return 11;
    default: return -1;
  }
}

int orc_env_obs_dims(int env)
{
  switch (env)
  {
    case ORC_ENV_PENDULUM: return 2;
    case ORC_ENV_ACROBOT: return 4;
    case ORC_ENV_CART_POLE: return 4;
    case ORC_ENV_CART_POLE_BALANCING: return 4;
    case ORC_ENV_COMPASS_WALKER: return 5;
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

/* -------------------------------------------------------------- acrobot -- */
static void acrobot_eom(const orc_spec *s, const double *x, double u, double *xd)
{ /* acrobot.cpp:48-79; state = [theta1, theta2, thetad1, thetad2, time] */
  double l1 = 1, m1 = 1, m2 = 1, lc1 = 0.5, lc2 = 0.5, I1 = 1, I2 = 1, g = 9.8;
  double theta1 = x[0], theta2 = x[1], thetad1 = x[2], thetad2 = x[3];
  double tau = u;
  double sin2 = orc_m_sin(s, theta2), cos2 = orc_m_cos(s, theta2);

  double phi2 = m2*lc2*g*orc_m_cos(s, theta1+theta2-M_PI/2);
  double phi1 = -m2*l1*lc2*thetad2*thetad2*sin2-2*m2*l1*lc2*thetad2*thetad1*sin2 +
                (m1*lc1+m2*l1)*g*orc_m_cos(s, theta1-M_PI/2)+phi2;
  double d2 = m2*(lc2*lc2+l1*lc2*cos2)+I2;
  double d1 = m1*lc1*lc1 + m2*(l1*l1+lc2*lc2+2*l1*lc2*cos2)+I1+I2;
  double thetadd2 = (tau+d2*phi1/d1-m2*l1*lc2*thetad2*thetad2*sin2-phi2)/
                    (m2*lc2*lc2+I2-d2*d2/d1);
  double thetadd1 = -(d2*thetadd2+phi1)/d1;

  /* limit velocity */
  if (thetad1 >  4*M_PI) thetadd1 = fmin(thetadd1, 0);
  if (thetad1 < -4*M_PI) thetadd1 = fmax(thetadd1, 0);
  if (thetad2 >  9*M_PI) thetadd2 = fmin(thetadd2, 0);
  if (thetad2 < -9*M_PI) thetadd2 = fmax(thetadd2, 0);

  xd[0] = thetad1;
  xd[1] = thetad2;
  xd[2] = thetadd1;
  xd[3] = thetadd2;
  xd[4] = 1;
}

static int acrobot_failed(const double *x)
{ /* acrobot.cpp:147-151 */
  return fabs(x[0]-M_PI) > 12*M_PI/180 || fabs(x[1]) > 12*M_PI/180;
}

static void acrobot_start(orc_exp *e, double *x)
{ /* acrobot.cpp:102-107: two RandGen draws, angle 1 first */
  for (int i = 0; i < 5; ++i) x[i] = 0.;
  x[0] = M_PI+orc_drand48(&e->TL)*0.01-0.005;
  x[1] = orc_drand48(&e->TL)*0.01-0.005;
}

static int acrobot_observe(const double *x, double *obs)
{ /* acrobot.cpp:109-125 */
  for (int i = 0; i < 4; ++i) obs[i] = x[i];
  if (acrobot_failed(x)) return 2;
  return x[4] > 20;
}

/* ------------------------------------------------------------ cart-pole -- */
static void cart_pole_eom(const orc_spec *s, const double *x, double u, double *xd)
{ /* cart_pole.cpp:41-49 constants, :58-108 with end_stop = 1; state = [x, theta, xd, thetad, time].
   * QUIRK reproduced on purpose: :65 reads `dtheta = state[3-2*end_stop_]`, which for
   * end_stop = 1 is state[1] -- the ANGLE, not its rate -- so the centrifugal term uses theta^2. */
  const double g = 9.8, mass_cart = 1.0, mass_pole = 0.1, length = 0.5;
  const double total_mass = mass_cart + mass_pole, pole_mass_length = mass_pole * length;
  double acc, thetaacc, costheta, sintheta, temp;
  double theta = x[1], dtheta = x[3 - 2 * 1];

  costheta = orc_m_cos(s, theta);
  sintheta = orc_m_sin(s, theta);
  temp = (u + pole_mass_length * dtheta * dtheta * sintheta) / total_mass;
  thetaacc = (g * sintheta - costheta * temp) /
             (length * ((4. / 3.) - mass_pole * costheta * costheta / total_mass));
  acc = temp - pole_mass_length * thetaacc * costheta / total_mass;

  xd[0] = x[2];
  xd[1] = x[3];
  xd[2] = acc;
  xd[3] = thetaacc;
  xd[4] = 1;
  /* end stops, :93-105 */
  if (x[0] > 2.4 && x[2] > 0)
  {
    xd[0] = 0;
    if (acc > 0) xd[2] = 0;
  }
  else if (x[0] < -2.4 && x[2] < 0)
  {
    xd[0] = 0;
    if (acc < 0) xd[2] = 0;
  }
}

static int cart_pole_failed(const double *x) { return fabs(x[0]) > 2.4; }          /* :212-215 */

static double cart_pole_potential(const orc_spec *s, const double *x)
{ /* :232-238 */
  double a = fmod(fabs(x[1]), 2 * M_PI);
  if (a > M_PI) a -= 2 * M_PI;
  return -2 * orc_m_sqr(s, x[0]) - 0.1 * orc_m_sqr(s, x[2]) - orc_m_sqr(s, a) - 0.1 * orc_m_sqr(s, x[3]);
}

static void cart_pole_start(const orc_spec *s, orc_exp *e, double *x)
{ /* :155-164: one RandGen draw every episode */
  x[0] = 0;
  x[1] = M_PI + s->randomization * ((orc_drand48(&e->TL) * 0.1) - 0.05);
  x[2] = 0;
  x[3] = 0;
  x[4] = 0;
}

static int cart_pole_observe(const orc_spec *s, const double *x, double *obs)
{ /* :166-190 */
  double a = fmod(x[1] + M_PI, 2 * M_PI);
  if (a < 0) a += 2 * M_PI;
  obs[0] = x[0];
  obs[1] = a;
  obs[2] = x[2];
  obs[3] = x[3];
  if (s->end_stop_penalty && cart_pole_failed(x)) return 2;
  return x[4] > s->timeout ? 1 : 0;
}

static double cart_pole_evaluate(const orc_spec *s, const double *x, double action, const double *next)
{ /* :192-201, shaping = 0 */
  (void)x;
  return cart_pole_potential(s, next) - s->action_penalty * orc_m_sqr(s, action / 15) * 2 - s->end_stop_penalty * cart_pole_failed(next) * 10000;
}

/* ------------------------------------------------- cart-pole balancing -- */
static int balancing_failed(const double *x)
{ /* cart_pole.cpp:317-320 */
  return fabs(x[0]) > 2.4 || fabs(x[1]) > 12*M_PI/180;
}

static void balancing_start(orc_exp *e, double *x)
{ /* cart_pole.cpp:264-273: one RandGen draw every episode */
  x[0] = 0;
  x[1] = (orc_drand48(&e->TL) * 0.1) - 0.05;
  x[2] = 0;
  x[3] = 0;
  x[4] = 0;
}

static int balancing_observe(const orc_spec *s, const double *x, double *obs)
{ /* cart_pole.cpp:275-296 */
  obs[0] = x[0];
  obs[1] = x[1];
  obs[2] = x[2];
  obs[3] = x[3];
  if (balancing_failed(x)) return 2;
  return x[4] > s->timeout ? 1 : 0;
}

static double balancing_evaluate(const double *x, const double *next)
{ /* cart_pole.cpp:298-307: the reward is taken at the state BEFORE the step */
  if (balancing_failed(next)) return 0;
  return 1 - (fabs(x[0]) + fabs(x[1])) / (2.4 + 12*M_PI/180);
}

/* -------------------------------------------------------- compass walker -- */
/* state vector (compass_walker.h:40-42): */
enum { W_SLA = 0, W_HA, W_SLAR, W_HAR, W_CHANGED, W_SFX, W_LASTHIPX, W_HIPVEL, W_STEPDIST, W_TIME, W_TIMEOUT };

typedef struct { double sla, slar, ha, har, sfx; int changed; } sw_state;    /* CSWModelState */
typedef struct { double sl, hip; } sw_accels;

static double sw_hip_x(const orc_spec *s, const sw_state *m) { return m->sfx - orc_m_sin(s, m->sla); }          /* SWModel.h:40 */
static double sw_hip_y(const orc_spec *s, const sw_state *m) { return orc_m_cos(s, m->sla); }                   /* :41 */
static double sw_swing_x(const orc_spec *s, const sw_state *m) { return sw_hip_x(s, m) + orc_m_sin(s, m->sla - m->ha); }   /* :42 */
static double sw_swing_y(const orc_spec *s, const sw_state *m) { return sw_hip_y(s, m) - orc_m_cos(s, m->sla - m->ha); }   /* :43 */

static void sw_wrap(sw_state *m)
{ /* SWModel.h:48-59 */
  if (m->sla >= M_PI) m->sla -= 2*M_PI;
  if (m->sla < -M_PI) m->sla += 2*M_PI;
  if (m->ha >= M_PI) m->ha -= 2*M_PI;
  if (m->ha < -M_PI) m->ha += 2*M_PI;
}

static void sw_accel(const orc_spec *s, const sw_state *m, double torque, sw_accels *a)
{ /* SWModel.cpp:212-218 */
  a->sl = orc_m_sin(s, m->sla - s->slope_angle);
  a->hip = orc_m_sin(s, m->ha) * (m->slar*m->slar - orc_m_cos(s, m->sla - s->slope_angle)) + a->sl;
  a->hip += torque;
}

static void sw_rk4(const orc_spec *s, sw_state *state, double torque, double dt)
{ /* SWModel.cpp:220-258 */
  sw_state s1, s2, s3, s4;
  sw_accels k1, k2, k3, k4;
  s1 = *state;
  sw_accel(s, &s1, torque, &k1);
  s2 = s1; s3 = s1; s4 = s1;
  s2.slar = s1.slar + (dt/2)*k1.sl;
  s2.har  = s1.har  + (dt/2)*k1.hip;
  s2.sla  = s1.sla  + (dt/2)*s1.slar;
  s2.ha   = s1.ha   + (dt/2)*s1.har;
  sw_accel(s, &s2, torque, &k2);
  s3.slar = s1.slar + (dt/2)*k2.sl;
  s3.har  = s1.har  + (dt/2)*k2.hip;
  s3.sla  = s1.sla  + (dt/2)*s2.slar;
  s3.ha   = s1.ha   + (dt/2)*s2.har;
  sw_accel(s, &s3, torque, &k3);
  s4.slar = s1.slar + (dt)*k3.sl;
  s4.har  = s1.har  + (dt)*k3.hip;
  s4.sla  = s1.sla  + (dt)*s3.slar;
  s4.ha   = s1.ha   + (dt)*s3.har;
  sw_accel(s, &s4, torque, &k4);
  state->slar = s1.slar + (dt/6)*(k1.sl + 2*k2.sl + 2*k3.sl + k4.sl);
  state->har  = s1.har  + (dt/6)*(k1.hip + 2*k2.hip + 2*k3.hip + k4.hip);
  state->sla  = s1.sla  + (dt/6)*(s1.slar + 2*s2.slar + 2*s3.slar + s4.slar);
  state->ha   = s1.ha   + (dt/6)*(s1.har + 2*s2.har + 2*s3.har + s4.har);
}

static double sw_heelstrike_moment(const orc_spec *s, const sw_state *t0, const sw_state *t1, sw_state *hs, double torque, double precision, double dt)
{ /* SWModel.cpp:53-104: secant search for the moment the swing foot reaches the floor */
  double timeLeft = 0;
  sw_state s0 = *t0, s1 = *t1;
  double s0time = 0, s1time = dt;
  const int maxIterations = 10;
  int iIter;
  for (iIter = 0; iIter < maxIterations; iIter++)
  {
    *hs = s0;
    double newDt = (s1time - s0time) * sw_swing_y(s, &s0) / (sw_swing_y(s, &s0) - sw_swing_y(s, &s1));
    sw_rk4(s, hs, torque, newDt);
    if (sw_swing_y(s, hs) > 0)
    {
      s0 = *hs;
      s0time = s0time + newDt;
    }
    else
    {
      s1 = *hs;
      s1time = s0time + newDt;
    }
    if (sw_swing_y(s, &s0) < precision)
    {
      *hs = s0;
      timeLeft = dt - s0time;
      break;
    }
    else if (-sw_swing_y(s, &s1) < precision)
    {
      *hs = s1;
      timeLeft = dt - s1time;
      break;
    }
  }
  if (iIter >= maxIterations)
  {
    if (sw_swing_y(s, hs) > 0)
      timeLeft = dt - s0time;
    else
      timeLeft = dt - s1time;
  }
  return timeLeft;
}

static double sw_detect_events(const orc_spec *s, const sw_state *t0, sw_state *hs, sw_state *t1, double torque, double dt)
{ /* SWModel.cpp:30-45 and processStanceLegChange :106-124 */
  if ((sw_swing_y(s, t0) >= 0) && (sw_swing_y(s, t1) < 0))
    if (((t0->ha < 0) && (t1->ha < 0)) || ((t0->ha > 0) && (t1->ha > 0)))
      if ((t1->slar < 0) && (t1->ha < 0))
      {
        double timeleft = sw_heelstrike_moment(s, t0, t1, hs, torque, 1.0E-11, dt);
        t1->har  = hs->slar*(orc_m_cos(s, 2.0*hs->sla)*(1.0 - orc_m_cos(s, 2.0*hs->sla)));
        t1->slar = hs->slar*(orc_m_cos(s, 2.0*hs->sla));
        t1->sfx  = sw_swing_x(s, hs);
        t1->sla  = -hs->sla;
        t1->ha   = -2.0*hs->sla;
        t1->changed = 1;
        return timeleft;
      }
  t1->changed = 0;
  return 0;
}

static void walker_model_step(const orc_spec *s, const double *x, double torque, double *next)
{ /* CompassWalkerModel::step (compass_walker.cpp:63-94) around CSWModel::singleStep (SWModel.cpp:142-210) */
  sw_state st, prev, hs;
  st.sfx = x[W_SFX]; st.sla = x[W_SLA]; st.slar = x[W_SLAR]; st.ha = x[W_HA]; st.har = x[W_HAR]; st.changed = 0;
  prev = st;
  hs = st;
  int changed = 0;
  /* setTiming (SWModel.cpp:126-130): step time in whole microseconds */
  double steptime_us = (double)(uint64_t)floor((s->control_step + 0.5E-6)*1E6);
  double partial = 1.0E-6*steptime_us/s->integration_steps;
  for (int i = 0; i < s->integration_steps; i++)
  {
    sw_rk4(s, &st, torque, partial);
    sw_wrap(&st);
    double timeleft = sw_detect_events(s, &prev, &hs, &st, torque, partial);
    changed |= (timeleft > 0);
    if (timeleft > 0)
    {
      sw_rk4(s, &st, torque, timeleft);
      sw_wrap(&st);
    }
    prev = st;
  }
  st.changed = changed;
  for (int i = 0; i < 11; ++i) next[i] = x[i];
  next[W_SLA] = st.sla;
  next[W_HA] = st.ha;
  next[W_SLAR] = st.slar;
  next[W_HAR] = st.har;
  next[W_SFX] = st.sfx;
  next[W_CHANGED] = st.changed;
  if (st.changed)
    next[W_LASTHIPX] = sw_hip_x(s, &st);
  else
    next[W_LASTHIPX] = x[W_LASTHIPX];
  next[W_HIPVEL] = - st.slar * orc_m_cos(s, st.sla);
  next[W_TIME] = x[W_TIME] + s->control_step;
  next[W_TIMEOUT] = x[W_TIMEOUT];
  /* siStepDistance is never written by CompassWalkerModel::step (uninitialised in the reference); it is
   * not observable with the walk task's default mask and is kept at its start value 0 here */
}

static void walker_start(const orc_spec *s, orc_exp *e, int test, double *x)
{ /* CompassWalkerWalkTask::start (compass_walker.cpp:251-290): rejection sampling on the GLOBAL drand48 */
  sw_state init, sw;
  init.sfx = 0; init.sla = 0.1534; init.slar = -0.1561; init.ha = 2.0*0.1534; init.har = -0.0073; init.changed = 0;
  sw = init;
  sw.sfx = 0;
  double variation = (!test) ? s->initial_state_variation : 0;
  do
  {
    sw.sla  = init.sla  * (1.0 - variation + 2.0*variation*orc_drand48(&e->G));
    sw.ha   = init.ha   * (1.0 - variation + 2.0*variation*orc_drand48(&e->G));
    sw.slar = init.slar * (1.0 - variation + 2.0*variation*orc_drand48(&e->G));
    sw.har  = init.har  * (1.0 - variation + 2.0*variation*orc_drand48(&e->G));
  }
  while (sw.slar*sw.slar/2.0 + sw_hip_y(s, &sw)*orc_m_cos(s, s->slope_angle) < orc_m_cos(s, s->slope_angle));
  for (int i = 0; i < 11; ++i) x[i] = 0;
  x[W_SLA] = sw.sla;
  x[W_HA] = sw.ha;
  x[W_SLAR] = sw.slar;
  x[W_HAR] = sw.har;
  x[W_SFX] = sw.sfx;
  x[W_CHANGED] = 0;
  x[W_LASTHIPX] = sw_hip_x(s, &sw);
  x[W_HIPVEL] = -sw.slar * orc_m_cos(s, sw.sla);
  x[W_STEPDIST] = 0;
  x[W_TIME] = 0;
  x[W_TIMEOUT] = test ? 2*s->timeout : s->timeout;
}

static int walker_fallen(const double *x)
{
  return fabs(x[W_SLA]) > M_PI/8 || fabs(x[W_HA] - 2 * x[W_SLA]) > M_PI/4;
}

static int walker_observe(const double *x, double *obs)
{ /* compass_walker.cpp:292-329 with observe = [1,1,1,1,1,0,0], steps = 0 */
  obs[0] = x[W_SLA];
  obs[1] = x[W_HA] - 2 * x[W_SLA];
  obs[2] = x[W_SLAR];
  obs[3] = x[W_HAR] - 2 * x[W_SLAR];
  obs[4] = x[W_CHANGED] > 0.5;
  if (walker_fallen(x)) return 2;
  if (x[W_TIME] > x[W_TIMEOUT]) return 1;
  return 0;
}

static double walker_evaluate(const orc_spec *s, const double *next)
{ /* compass_walker.cpp:331-344 */
  double reward = -1;
  if (next[W_CHANGED] > 0.5)
    reward = fmin(50 * 4 * orc_m_sin(s, next[W_SLA]), 30);
  if (walker_fallen(next))
    if (s->negative_reward)
      reward = s->negative_reward;
  return reward;
}

/* ------------------------------------------------------ generic dispatch -- */
static void env_eom(const orc_spec *s, const double *x, double u, double *xd)
{
  switch (s->env)
  {
    case ORC_ENV_PENDULUM: pendulum_eom(s, x, u, xd); break;
    case ORC_ENV_ACROBOT: acrobot_eom(s, x, u, xd); break;
    case ORC_ENV_CART_POLE: cart_pole_eom(s, x, u, xd); break;
    case ORC_ENV_CART_POLE_BALANCING: cart_pole_eom(s, x, u, xd); break;     /* the same dynamics/cart_pole, end_stop = 1 */
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
    case ORC_ENV_ACROBOT: acrobot_start(e, x); break;
    case ORC_ENV_CART_POLE: cart_pole_start(s, e, x); break;
    case ORC_ENV_CART_POLE_BALANCING: balancing_start(e, x); break;
    case ORC_ENV_COMPASS_WALKER: walker_start(s, e, test, x); break;
  }
}

int orc_env_observe(const orc_spec *s, const double *x, double *obs)
{
  switch (s->env)
  {
    case ORC_ENV_PENDULUM: return pendulum_observe(s, x, obs);
    case ORC_ENV_ACROBOT: return acrobot_observe(x, obs);
    case ORC_ENV_CART_POLE: return cart_pole_observe(s, x, obs);
    case ORC_ENV_CART_POLE_BALANCING: return balancing_observe(s, x, obs);
    case ORC_ENV_COMPASS_WALKER: return walker_observe(x, obs);
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

  if (s->env == ORC_ENV_COMPASS_WALKER)
    walker_model_step(s, state, actuation, next);        /* model/compass_walker has its own integrator */
  else
    rk4_step(s, state, actuation, next);
  *terminal = orc_env_observe(s, next, obs);
  switch (s->env)
  {
    case ORC_ENV_PENDULUM: *reward = pendulum_evaluate(s, state, action, next); break;
    case ORC_ENV_ACROBOT: *reward = !acrobot_failed(next); break;        /* acrobot.cpp:127-133 */
    case ORC_ENV_CART_POLE: *reward = cart_pole_evaluate(s, state, action, next); break;
    case ORC_ENV_CART_POLE_BALANCING: *reward = balancing_evaluate(state, next); break;
    case ORC_ENV_COMPASS_WALKER: *reward = walker_evaluate(s, next); break;
    default: *reward = 0;
  }
  memcpy(state, next, sizeof(double) * S);
  return 1;
}
