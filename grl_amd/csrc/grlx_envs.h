// grlx_envs.h -- environments: dynamics + tasks (pendulum, acrobot, cart-pole, compass walker), RK4 and the environment step
// (modeled.cpp:132-276, pendulum.cpp, acrobot.cpp, cart_pole.cpp, compass_walker.cpp, SWModel.cpp).
// Part of the single translation unit grlx_kernels.hip (included there, in order; not self-contained).
#pragma once

namespace grlx {

// ----------------------------------------------------------- environments --
template <int ENV> struct Env;

// Lanes that integrate the same replica hold identical values and would each evaluate every sin/cos of an equation of
// motion.  With a LaneShare they split the INDEPENDENT evaluations instead: the lane whose role is r evaluates the r-th
// angle (the same psincos code in every lane, another argument per role) and the results are exchanged by ds_bpermute.
// Every value is produced by the operations the unsplit code performs on the same argument, so the bits are the same.
struct NoShare { static constexpr bool kSplit = false; };
struct LaneShare {
  static constexpr bool kSplit = true;
  int src[3];          // lane of my replica that evaluates role 0, 1, 2
  int role3, role2;    // my role when three / two evaluations are shared out
};
// Two lanes per (replica, action) pair share an equation of motion (the environment server of the wide kernels, grlx_env_server_wide.h:
// 8 replicas x 3 actions x 2 roles = 48 lanes): the walker's two angles one each, the acrobot's three as 1 + 2.
struct PairShare {
  static constexpr bool kSplit = true;
  int src[2];          // lane that evaluates role 0, 1
  int role2;           // my role
};
__device__ __forceinline__ double lane_fetch(double v, int src) { return __shfl(v, src, 64); }

// dynamics/pendulum + task/pendulum/swingup (pendulum.cpp:40-145)
template <> struct Env<GRLX_ENV_PENDULUM> {
  static constexpr int S = 3, D = 2;
  static constexpr bool kAbsorbing = false;      // can observe() report an absorbing state (terminal = 2)?
  // pendulum.cpp:40-49, 55-68; the constants are held in registers by the caller (rk4_step)
  struct Consts { SinConsts k; double invJ, mgl, b, kkr, kr; };
  template <bool PIN> __device__ static __forceinline__ Consts consts()
  {
    const double J = 0.000191, m = 0.055, g = 9.81, l = 0.042, b = 0.000003, K = 0.0536, R = 9.5;
    Consts c;
    c.k = sin_consts<PIN>();
    c.invJ = math_const<PIN>(1 / J);
    c.mgl = math_const<PIN>(m * g * l);
    c.b = math_const<PIN>(b);
    c.kkr = math_const<PIN>(K * K / R);
    c.kr = math_const<PIN>(K / R);
    return c;
  }
  __device__ static __forceinline__ void eom(const Consts &c, const double *x, double u, double *xd)
  {
    double a = x[0], ad = x[1];
    double add = c.invJ * (c.mgl * psin(a, c.k) - c.b * ad - c.kkr * ad + c.kr * u);
    xd[0] = ad;
    xd[1] = add;
    xd[2] = 1;
  }
  __device__ static __forceinline__ void start(const DevParams &P, int test, uint64_t &TL, uint64_t &, double *x)
  { // pendulum.cpp:97-103 (the RandGen draw happens every episode)
    TL = lcg_next(TL);
    double r = lcg_double(TL);
    x[0] = GRLX_PI + P.randomization * (test == 0) * r * 2 * GRLX_PI;
    x[1] = 0;
    x[2] = 0;
  }
  __device__ static __forceinline__ double actuate(double a) { return fmin(fmax(a, -3.0), 3.0); }   // :105-109
  __device__ static __forceinline__ bool in_domain(const double *x) { return __builtin_fabs(x[0]) < 0x1p19; }
  __device__ static __forceinline__ int observe(const DevParams &P, const double *x, double *obs)
  { // :111-129
    double a = pfmod(x[0] + GRLX_PI, GRLX_2PI);
    if (a < 0) a += GRLX_2PI;
    obs[0] = a;
    obs[1] = x[1];
    return x[2] > P.timeout ? 1 : 0;
  }
  __device__ static __forceinline__ double evaluate(const DevParams &, const double *x, double action, const double *next)
  { // :131-145; pow(v, 2) is v*v in the portable specification
    double a = pfmod(__builtin_fabs(next[0]), GRLX_2PI);
    if (a > GRLX_PI) a -= GRLX_2PI;
    double reward = -5 * (a * a) - 0.1 * (next[1] * next[1]) - 1 * (action * action);
    if ((next[2] - x[2]) != 1)
      reward *= (next[2] - x[2]) / 0.03;
    return reward;
  }
};

// dynamics/acrobot + task/acrobot/balancing (acrobot.cpp:48-151); state = [theta1, theta2,
// thetad1, thetad2, time].  No reference test pins it: parity is against the oracle only.
template <> struct Env<GRLX_ENV_ACROBOT> {
  static constexpr int S = 5, D = 4;
  static constexpr bool kAbsorbing = true;      // can observe() report an absorbing state (terminal = 2)?
  using Consts = SinConsts;                     // held in registers across the integration loop
  template <bool PIN> __device__ static __forceinline__ Consts consts() { return sin_consts<PIN>(); }
  // Equations of motion of dynamics/acrobot (acrobot.cpp:48-79).  Unit link masses, lengths and inertias fold the
  // reference's parameter products into five constants; each is formed by the same left-to-right products the
  // reference evaluates at run time, and every remaining operation keeps the reference's order (bit parity).
  __device__ static __forceinline__ void eom(const Consts &k, const double *x, double u, double *xd)
  {
    constexpr double kElbowGrav = 1.0 * 0.5 * 9.8;                // m2*lc2*g
    constexpr double kCoriolis = -1.0 * 1.0 * 0.5;                // -m2*l1*lc2
    constexpr double kCoupling = 2 * 1.0 * 1.0 * 0.5;             // 2*m2*l1*lc2  (= 1: exact, the product below keeps it)
    constexpr double kShoulderGrav = (1.0 * 0.5 + 1.0 * 1.0) * 9.8;    // (m1*lc1+m2*l1)*g
    constexpr double kElbowInertia = 1.0 * 0.5 * 0.5 + 1.0;       // m2*lc2*lc2+I2, the denominator's constant part
    const double q1 = x[0], q2 = x[1], w1 = x[2], w2 = x[3];
    double s2, c2;
    psincos(q2, k, s2, c2);

    const double g_elbow = kElbowGrav * pcos(q1 + q2 - GRLX_PI / 2, k);
    const double bias = kCoriolis * w2 * w2 * s2 - kCoupling * w2 * w1 * s2 + kShoulderGrav * pcos(q1 - GRLX_PI / 2, k) + g_elbow;
    const double m_cross = 1.0 * (0.5 * 0.5 + 1.0 * 0.5 * c2) + 1.0;                                    // d2
    const double m_shoulder = 1.0 * 0.5 * 0.5 + 1.0 * (1.0 * 1.0 + 0.5 * 0.5 + 2 * 1.0 * 0.5 * c2) + 1.0 + 1.0;   // d1
    double a_elbow = (u + m_cross * bias / m_shoulder - 1.0 * 1.0 * 0.5 * w2 * w2 * s2 - g_elbow) /
                     (kElbowInertia - m_cross * m_cross / m_shoulder);
    double a_shoulder = -(m_cross * a_elbow + bias) / m_shoulder;

    // joint-rate limits: beyond 4 pi (shoulder) / 9 pi (elbow) a joint may only decelerate
    if (w1 > 4 * GRLX_PI) a_shoulder = fmin(a_shoulder, 0.);
    if (w1 < -4 * GRLX_PI) a_shoulder = fmax(a_shoulder, 0.);
    if (w2 > 9 * GRLX_PI) a_elbow = fmin(a_elbow, 0.);
    if (w2 < -9 * GRLX_PI) a_elbow = fmax(a_elbow, 0.);

    xd[0] = w1;
    xd[1] = w2;
    xd[2] = a_shoulder;
    xd[3] = a_elbow;
    xd[4] = 1;
  }
  // the same with the three sin/cos evaluations shared out over the lanes of the replica
  __device__ static __forceinline__ void eom(const Consts &k, const double *x, double u, double *xd, const LaneShare &ls)
  {
    constexpr double kElbowGrav = 1.0 * 0.5 * 9.8;
    constexpr double kCoriolis = -1.0 * 1.0 * 0.5;
    constexpr double kCoupling = 2 * 1.0 * 1.0 * 0.5;
    constexpr double kShoulderGrav = (1.0 * 0.5 + 1.0 * 1.0) * 9.8;
    constexpr double kElbowInertia = 1.0 * 0.5 * 0.5 + 1.0;
    const double q1 = x[0], q2 = x[1], w1 = x[2], w2 = x[3];
    const double angle = (ls.role3 == 0) ? q2 : (ls.role3 == 1) ? q1 + q2 - GRLX_PI / 2 : q1 - GRLX_PI / 2;
    double sn, cs;
    psincos(angle, k, sn, cs);                                   // psincos' cosine == pcos (same quadrant logic, same kernels)
    const double s2 = lane_fetch(sn, ls.src[0]), c2 = lane_fetch(cs, ls.src[0]);
    const double cos_elbow = lane_fetch(cs, ls.src[1]), cos_shoulder = lane_fetch(cs, ls.src[2]);

    const double g_elbow = kElbowGrav * cos_elbow;
    const double bias = kCoriolis * w2 * w2 * s2 - kCoupling * w2 * w1 * s2 + kShoulderGrav * cos_shoulder + g_elbow;
    const double m_cross = 1.0 * (0.5 * 0.5 + 1.0 * 0.5 * c2) + 1.0;
    const double m_shoulder = 1.0 * 0.5 * 0.5 + 1.0 * (1.0 * 1.0 + 0.5 * 0.5 + 2 * 1.0 * 0.5 * c2) + 1.0 + 1.0;
    double a_elbow = (u + m_cross * bias / m_shoulder - 1.0 * 1.0 * 0.5 * w2 * w2 * s2 - g_elbow) /
                     (kElbowInertia - m_cross * m_cross / m_shoulder);
    double a_shoulder = -(m_cross * a_elbow + bias) / m_shoulder;
    if (w1 > 4 * GRLX_PI) a_shoulder = fmin(a_shoulder, 0.);
    if (w1 < -4 * GRLX_PI) a_shoulder = fmax(a_shoulder, 0.);
    if (w2 > 9 * GRLX_PI) a_elbow = fmin(a_elbow, 0.);
    if (w2 < -9 * GRLX_PI) a_elbow = fmax(a_elbow, 0.);
    xd[0] = w1;
    xd[1] = w2;
    xd[2] = a_shoulder;
    xd[3] = a_elbow;
    xd[4] = 1;
  }
  // ... and shared out over TWO lanes: role 0 evaluates the elbow angle (sine and cosine), role 1 the two gravity cosines
  __device__ static __forceinline__ void eom(const Consts &k, const double *x, double u, double *xd, const PairShare &ls)
  {
    constexpr double kElbowGrav = 1.0 * 0.5 * 9.8;
    constexpr double kCoriolis = -1.0 * 1.0 * 0.5;
    constexpr double kCoupling = 2 * 1.0 * 1.0 * 0.5;
    constexpr double kShoulderGrav = (1.0 * 0.5 + 1.0 * 1.0) * 9.8;
    constexpr double kElbowInertia = 1.0 * 0.5 * 0.5 + 1.0;
    const double q1 = x[0], q2 = x[1], w1 = x[2], w2 = x[3];
    double sn, cs, sn_b, cs_b;
    psincos(ls.role2 == 0 ? q2 : q1 + q2 - GRLX_PI / 2, k, sn, cs);       // psincos' cosine == pcos (same quadrant logic, same kernels)
    psincos(q1 - GRLX_PI / 2, k, sn_b, cs_b);                             // (role 0 evaluates it too and does not use it: one instruction stream)
    const double s2 = lane_fetch(sn, ls.src[0]), c2 = lane_fetch(cs, ls.src[0]);
    const double cos_elbow = lane_fetch(cs, ls.src[1]), cos_shoulder = lane_fetch(cs_b, ls.src[1]);

    const double g_elbow = kElbowGrav * cos_elbow;
    const double bias = kCoriolis * w2 * w2 * s2 - kCoupling * w2 * w1 * s2 + kShoulderGrav * cos_shoulder + g_elbow;
    const double m_cross = 1.0 * (0.5 * 0.5 + 1.0 * 0.5 * c2) + 1.0;
    const double m_shoulder = 1.0 * 0.5 * 0.5 + 1.0 * (1.0 * 1.0 + 0.5 * 0.5 + 2 * 1.0 * 0.5 * c2) + 1.0 + 1.0;
    double a_elbow = (u + m_cross * bias / m_shoulder - 1.0 * 1.0 * 0.5 * w2 * w2 * s2 - g_elbow) /
                     (kElbowInertia - m_cross * m_cross / m_shoulder);
    double a_shoulder = -(m_cross * a_elbow + bias) / m_shoulder;
    if (w1 > 4 * GRLX_PI) a_shoulder = fmin(a_shoulder, 0.);
    if (w1 < -4 * GRLX_PI) a_shoulder = fmax(a_shoulder, 0.);
    if (w2 > 9 * GRLX_PI) a_elbow = fmin(a_elbow, 0.);
    if (w2 < -9 * GRLX_PI) a_elbow = fmax(a_elbow, 0.);
    xd[0] = w1;
    xd[1] = w2;
    xd[2] = a_shoulder;
    xd[3] = a_elbow;
    xd[4] = 1;
  }
  __device__ static __forceinline__ bool failed(const double *x)
  { // :147-151
    return __builtin_fabs(x[0]-GRLX_PI) > 12*GRLX_PI/180 || __builtin_fabs(x[1]) > 12*GRLX_PI/180;
  }
  __device__ static __forceinline__ void start(const DevParams &, int, uint64_t &TL, uint64_t &, double *x)
  { // :102-107
    TL = lcg_next(TL);
    const double r1 = lcg_double(TL);
    TL = lcg_next(TL);
    const double r2 = lcg_double(TL);
    x[0] = GRLX_PI+r1*0.01-0.005;
    x[1] = r2*0.01-0.005;
    x[2] = 0; x[3] = 0; x[4] = 0;
  }
  __device__ static __forceinline__ double actuate(double a) { return a; }                  // Task::actuate default (environment.h:94)
  __device__ static __forceinline__ bool in_domain(const double *x) { return __builtin_fabs(x[0]) < 0x1p18 && __builtin_fabs(x[1]) < 0x1p18; }
  __device__ static __forceinline__ int observe(const DevParams &, const double *x, double *obs)
  { // :109-125
#pragma unroll
    for (int i = 0; i < 4; ++i) obs[i] = x[i];
    if (failed(x)) return 2;
    return x[4] > 20 ? 1 : 0;
  }
  __device__ static __forceinline__ double evaluate(const DevParams &, const double *, double, const double *next)
  { // :127-133
    return failed(next) ? 0. : 1.;
  }
};

// dynamics/cart_pole (end_stop = 1) + task/cart_pole/swingup (cart_pole.cpp:41-237);
// state = [x, theta, xd, thetad, time].  parity unpinned by reference tests.
template <> struct Env<GRLX_ENV_CART_POLE> {
  static constexpr int S = 5, D = 4;
  static constexpr bool kAbsorbing = true;      // can observe() report an absorbing state (terminal = 2)?
  using Consts = SinConsts;                     // held in registers across the integration loop
  template <bool PIN> __device__ static __forceinline__ Consts consts() { return sin_consts<PIN>(); }
  __device__ static __forceinline__ void eom(const Consts &k, const double *x, double u, double *xd)
  { // cart_pole.cpp:58-108.  QUIRK reproduced on purpose: :65 reads dtheta = state[3-2*end_stop_],
    // which for end_stop = 1 is state[1] -- the ANGLE, not its rate.
    const double g = 9.8, mass_cart = 1.0, mass_pole = 0.1, length = 0.5;
    const double total_mass = mass_cart + mass_pole, pole_mass_length = mass_pole * length;
    const double theta = x[1], dtheta = x[3 - 2 * 1];
    double costheta, sintheta;
    psincos(theta, k, sintheta, costheta);
    const double temp = (u + pole_mass_length * dtheta * dtheta * sintheta) / total_mass;
    const double thetaacc = (g * sintheta - costheta * temp) /
                            (length * ((4. / 3.) - mass_pole * costheta * costheta / total_mass));
    const double acc = temp - pole_mass_length * thetaacc * costheta / total_mass;
    xd[0] = x[2];
    xd[1] = x[3];
    xd[2] = acc;
    xd[3] = thetaacc;
    xd[4] = 1;
    if (x[0] > 2.4 && x[2] > 0)
    { // end stops, :93-105
      xd[0] = 0;
      if (acc > 0) xd[2] = 0;
    }
    else if (x[0] < -2.4 && x[2] < 0)
    {
      xd[0] = 0;
      if (acc < 0) xd[2] = 0;
    }
  }
  __device__ static __forceinline__ bool failed(const double *x) { return __builtin_fabs(x[0]) > 2.4; }   // :212-215
  __device__ static __forceinline__ double potential(const double *x)
  { // :232-238
    double a = pfmod(__builtin_fabs(x[1]), GRLX_2PI);
    if (a > GRLX_PI) a -= GRLX_2PI;
    return -2 * (x[0] * x[0]) - 0.1 * (x[2] * x[2]) - (a * a) - 0.1 * (x[3] * x[3]);
  }
  __device__ static __forceinline__ void start(const DevParams &P, int, uint64_t &TL, uint64_t &, double *x)
  { // :155-164
    TL = lcg_next(TL);
    const double r = lcg_double(TL);
    x[0] = 0;
    x[1] = GRLX_PI + P.randomization * ((r * 0.1) - 0.05);
    x[2] = 0; x[3] = 0; x[4] = 0;
  }
  __device__ static __forceinline__ double actuate(double a) { return a; }                  // Task::actuate default (environment.h:94)
  __device__ static __forceinline__ bool in_domain(const double *x) { return __builtin_fabs(x[1]) < 0x1p19; }
  __device__ static __forceinline__ int observe(const DevParams &P, const double *x, double *obs)
  { // :166-190
    double a = pfmod(x[1] + GRLX_PI, GRLX_2PI);
    if (a < 0) a += GRLX_2PI;
    obs[0] = x[0];
    obs[1] = a;
    obs[2] = x[2];
    obs[3] = x[3];
    if (P.end_stop_penalty && failed(x)) return 2;
    return x[4] > P.timeout ? 1 : 0;
  }
  __device__ static __forceinline__ double evaluate(const DevParams &P, const double *, double action, const double *next)
  { // :192-201, shaping = 0
    const double a15 = action / 15;
    return potential(next) - P.action_penalty * (a15 * a15) * 2 - P.end_stop_penalty * (failed(next) ? 1 : 0) * 10000;
  }
};

// dynamics/cart_pole (end_stop = 1) + task/cart_pole/balancing (cart_pole.cpp:239-320): the same dynamics under the
// task of the reference's tests/cart_pole_balancing-pid.yaml, whose golden file pins them (fine-grained
// grlx_env_step only: the reference drives it with a PID policy, not with a TD agent).
template <> struct Env<GRLX_ENV_CART_POLE_BALANCING> : Env<GRLX_ENV_CART_POLE> {
  __device__ static __forceinline__ bool failed(const double *x)
  { // :317-320
    return __builtin_fabs(x[0]) > 2.4 || __builtin_fabs(x[1]) > 12*GRLX_PI/180;
  }
  __device__ static __forceinline__ void start(const DevParams &, int, uint64_t &TL, uint64_t &, double *x)
  { // :264-273
    TL = lcg_next(TL);
    const double r = lcg_double(TL);
    x[0] = 0;
    x[1] = (r * 0.1) - 0.05;
    x[2] = 0; x[3] = 0; x[4] = 0;
  }
  __device__ static __forceinline__ int observe(const DevParams &P, const double *x, double *obs)
  { // :275-296
#pragma unroll
    for (int i = 0; i < 4; ++i) obs[i] = x[i];
    if (failed(x)) return 2;
    return x[4] > P.timeout ? 1 : 0;
  }
  __device__ static __forceinline__ double evaluate(const DevParams &, const double *x, double, const double *next)
  { // :298-307: the reward looks at the state BEFORE the step
    if (failed(next)) return 0.;
    return 1 - (__builtin_fabs(x[0]) + __builtin_fabs(x[1])) / (2.4 + 12*GRLX_PI/180);
  }
};

// model/compass_walker + task/compass_walker/walk: the simplest walking model with its own
// RK4 (velocities and angles staged separately), angle wrapping and heel-strike events located
// by a secant search (SWModel.cpp:15-258, SWModel.h:40-59, compass_walker.cpp:63-94, 251-344).
// state vector (compass_walker.h:40-42).  parity unpinned by reference tests.
template <> struct Env<GRLX_ENV_COMPASS_WALKER> {
  static constexpr int S = 11, D = 5;
  static constexpr bool kAbsorbing = true;      // can observe() report an absorbing state (terminal = 2)?
  static constexpr bool kCustomModel = true;
  enum { SLA = 0, HA, SLAR, HAR, CHANGED, SFX, LASTHIPX, HIPVEL, STEPDIST, TIME, TIMEOUT };
  struct St { double sla, slar, ha, har, sfx; };

  __device__ static __forceinline__ double hip_x(const St &m) { return m.sfx - psin(m.sla); }
  __device__ static __forceinline__ double swing_y(const St &m) { return pcos(m.sla) - pcos(m.sla - m.ha); }
  // the same with the sine constants held in registers by model_step (20 sub-steps x 10 evaluations)
  __device__ static __forceinline__ double hip_x(const SinConsts &k, const St &m) { return m.sfx - psin_s(m.sla, k); }
  __device__ static __forceinline__ double swing_y(const SinConsts &k, const St &m) { return pcos_s(m.sla, k) - pcos_s(m.sla - m.ha, k); }
  // ... its two cosines one each where the lanes of a replica share evaluations (role 0: the stance leg, role 1: the swing leg)
  template <typename SH>
  __device__ static __forceinline__ double swing_y(const SinConsts &k, const St &m, const SH &sh)
  {
    if constexpr (SH::kSplit)
    {
      const double c = pcos_s(sh.role2 == 0 ? m.sla : m.sla - m.ha, k);
      return lane_fetch(c, sh.src[0]) - lane_fetch(c, sh.src[1]);
    }
    else
      return swing_y(k, m);
  }
  __device__ static __forceinline__ void wrap(St &m)
  { // SWModel.h:48-59
    if (m.sla >= GRLX_PI) m.sla -= 2*GRLX_PI;
    if (m.sla < -GRLX_PI) m.sla += 2*GRLX_PI;
    if (m.ha >= GRLX_PI) m.ha -= 2*GRLX_PI;
    if (m.ha < -GRLX_PI) m.ha += 2*GRLX_PI;
  }
  template <typename SH>
  __device__ static __forceinline__ void accel(const DevParams &P, const SinConsts &k, const St &m, double torque, double &asl, double &ahip, const SH &sh)
  { // SWModel.cpp:212-218
    double sn, cs, sin_hip;
    if constexpr (SH::kSplit)
    { // two independent angles: one psincos per lane (role 0: stance leg against the slope, role 1: hip), results exchanged
      double a_sn, a_cs;
      psincos_s(sh.role2 == 0 ? m.sla - P.slope_angle : m.ha, k, a_sn, a_cs);     // psincos_s' sine == psin_s (same kernels)
      sn = lane_fetch(a_sn, sh.src[0]);
      cs = lane_fetch(a_cs, sh.src[0]);
      sin_hip = lane_fetch(a_sn, sh.src[1]);
    }
    else
    {
      psincos_s(m.sla - P.slope_angle, k, sn, cs);
      sin_hip = psin_s(m.ha, k);
    }
    asl = sn;
    ahip = sin_hip * (m.slar*m.slar - cs) + asl;
    ahip += torque;
  }
  // One Runge-Kutta step of the walker's own integrator (SWModel.cpp:220-258): angles and rates are staged separately
  // -- stage j's angles advance with stage j-1's RATES, its rates with stage j-1's ACCELERATIONS -- and the stance
  // foot does not move.  Stage states live in two small arrays (angle pair, rate pair); the update coefficients and
  // their order of evaluation are the reference's.
  template <typename SH>
  __device__ static __forceinline__ void rk4(const DevParams &P, const SinConsts &k, St &state, double torque, double dt, const SH &sh)
  {
    const St base = state;
    double rate_sl[4], rate_hip[4], acc_sl[4], acc_hip[4];       // per stage: d(angle)/dt and d(rate)/dt
    St stage = base;
    const double half = dt / 2;
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
      if (j > 0)
      { // stage j starts from the base state, advanced by the previous stage's derivatives over half a step (full for j = 3)
        const double span = (j == 3) ? dt : half;
        stage.slar = base.slar + span * acc_sl[j - 1];
        stage.har = base.har + span * acc_hip[j - 1];
        stage.sla = base.sla + span * rate_sl[j - 1];
        stage.ha = base.ha + span * rate_hip[j - 1];
      }
      rate_sl[j] = stage.slar;
      rate_hip[j] = stage.har;
      accel(P, k, stage, torque, acc_sl[j], acc_hip[j], sh);
    }
    const double sixth = dt / 6;
    state.slar = base.slar + sixth * (acc_sl[0] + 2 * acc_sl[1] + 2 * acc_sl[2] + acc_sl[3]);
    state.har = base.har + sixth * (acc_hip[0] + 2 * acc_hip[1] + 2 * acc_hip[2] + acc_hip[3]);
    state.sla = base.sla + sixth * (rate_sl[0] + 2 * rate_sl[1] + 2 * rate_sl[2] + rate_sl[3]);
    state.ha = base.ha + sixth * (rate_hip[0] + 2 * rate_hip[1] + 2 * rate_hip[2] + rate_hip[3]);
  }
  // Time of the heel strike inside a sub-step (SWModel.cpp:53-104): the swing foot is above ground at `above` (time 0)
  // and below at `below` (time dt).  A bracketing secant search on the swing-foot height: integrate from the upper
  // bracket by the linearly interpolated time to the zero crossing, replace the bracket on the same side, stop when
  // either bracket is within `tol` of the ground or after ten rounds.  Returns the time LEFT in the sub-step after the
  // strike; `hit` is the state at the strike.  The height is a pure function of the state: the text asks for the
  // brackets' heights again in every round (five evaluations per round); here each state's height is evaluated once,
  // when the state is reached, and carried with it (y_above, y_below: the caller's detectEvents has them).
  template <typename SH>
  __device__ static __forceinline__ double heelstrike_moment(const DevParams &P, const SinConsts &k, const St &above, double y_above, const St &below,
                                                             double y_below, St &hit, double torque, double tol, double dt, const SH &sh)
  {
    St up = above, down = below;                  // brackets: swing foot above / below the ground
    double y_up = y_above, y_down = y_below, y_hit = y_above;
    double t_up = 0, t_down = dt;
    for (int round = 0; round < 10; ++round)
    {
      hit = up;
      const double t_cross = (t_down - t_up) * y_up / (y_up - y_down);
      rk4(P, k, hit, torque, t_cross, sh);
      y_hit = swing_y(k, hit, sh);
      if (y_hit > 0) { up = hit; y_up = y_hit; t_up = t_up + t_cross; }
      else { down = hit; y_down = y_hit; t_down = t_up + t_cross; }
      if (y_up < tol) { hit = up; return dt - t_up; }
      if (-y_down < tol) { hit = down; return dt - t_down; }
    }
    // not converged: the side the last probe fell on
    return (y_hit > 0) ? dt - t_up : dt - t_down;
  }
  template <typename SH, bool PIN = true>
  __device__ static __forceinline__ void model_step(const DevParams &P, const double *x, double torque, double *next, const SH &sh)
  { // CompassWalkerModel::step (compass_walker.cpp:63-94) around CSWModel::singleStep (SWModel.cpp:142-210)
    St st, prev, hs;
    st.sfx = x[SFX]; st.sla = x[SLA]; st.slar = x[SLAR]; st.ha = x[HA]; st.har = x[HAR];
    prev = st;
    hs = st;
    bool changed = false;
    const double partial = P.walker_dt;
    const SinConsts k = sin_consts<PIN>();       // PIN: the sine's constants held in vector registers across the 20 sub-steps
    double y_prev = swing_y(k, prev, sh);            // swing_y(prev) of the next sub-step is this sub-step's swing_y(st)
    for (int i = 0; i < P.integration_steps; i++)
    {
      rk4(P, k, st, torque, partial, sh);
      wrap(st);
      // detectEvents (SWModel.cpp:30-45)
      double timeleft = 0;
      bool struck = false;
      const double y_now = swing_y(k, st, sh);
      if ((y_prev >= 0) && (y_now < 0))
        if (((prev.ha < 0) && (st.ha < 0)) || ((prev.ha > 0) && (st.ha > 0)))
          if ((st.slar < 0) && (st.ha < 0))
          { // processStanceLegChange (:106-124)
            struck = true;
            timeleft = heelstrike_moment(P, k, prev, y_prev, st, y_now, hs, torque, 1.0E-11, partial, sh);
            const double c2 = pcos(2.0*hs.sla, k);
            st.har  = hs.slar*(c2*(1.0 - c2));
            st.slar = hs.slar*(c2);
            st.sfx  = hip_x(k, hs) + psin(hs.sla - hs.ha, k);
            st.sla  = -hs.sla;
            st.ha   = -2.0*hs.sla;
          }
      changed = changed || (timeleft > 0);
      if (timeleft > 0)
      {
        rk4(P, k, st, torque, timeleft, sh);
        wrap(st);
      }
      // a pure function of the state: recomputed only where a heel strike replaced the state
      y_prev = struck ? swing_y(k, st, sh) : y_now;
      prev = st;
    }
#pragma unroll
    for (int i = 0; i < S; ++i) next[i] = x[i];
    next[SLA] = st.sla;
    next[HA] = st.ha;
    next[SLAR] = st.slar;
    next[HAR] = st.har;
    next[SFX] = st.sfx;
    next[CHANGED] = changed ? 1. : 0.;
    next[LASTHIPX] = changed ? hip_x(k, st) : x[LASTHIPX];
    next[HIPVEL] = - st.slar * pcos(st.sla, k);
    next[TIME] = x[TIME] + P.control_step;
    next[TIMEOUT] = x[TIMEOUT];
  }
  __device__ static __forceinline__ void eom(const double *, double, double *) {}
  __device__ static __forceinline__ void start(const DevParams &P, int test, uint64_t &, uint64_t &G, double *x)
  { // compass_walker.cpp:251-290: rejection sampling on the GLOBAL drand48 stream
    const double i_sla = 0.1534, i_slar = -0.1561, i_ha = 2.0*0.1534, i_har = -0.0073;
    const double variation = (!test) ? P.initial_state_variation : 0;
    const double cslope = pcos(P.slope_angle);
    St sw;
    sw.sfx = 0;
    for (int guard = 0; guard < 100000; ++guard)
    {
      G = lcg_next(G); sw.sla  = i_sla  * (1.0 - variation + 2.0*variation*lcg_double(G));
      G = lcg_next(G); sw.ha   = i_ha   * (1.0 - variation + 2.0*variation*lcg_double(G));
      G = lcg_next(G); sw.slar = i_slar * (1.0 - variation + 2.0*variation*lcg_double(G));
      G = lcg_next(G); sw.har  = i_har  * (1.0 - variation + 2.0*variation*lcg_double(G));
      if (!(sw.slar*sw.slar/2.0 + pcos(sw.sla)*cslope < cslope)) break;
    }
#pragma unroll
    for (int i = 0; i < S; ++i) x[i] = 0;
    x[SLA] = sw.sla;
    x[HA] = sw.ha;
    x[SLAR] = sw.slar;
    x[HAR] = sw.har;
    x[SFX] = sw.sfx;
    x[LASTHIPX] = hip_x(sw);
    x[HIPVEL] = -sw.slar * pcos(sw.sla);
    x[TIMEOUT] = test ? 2*P.timeout : P.timeout;
  }
  __device__ static __forceinline__ double actuate(double a) { return a; }
  __device__ static __forceinline__ bool in_domain(const double *x) { return __builtin_fabs(x[SLA]) < 8. && __builtin_fabs(x[HA]) < 8. && __builtin_fabs(x[SLAR]) < 1e6; }
  __device__ static __forceinline__ bool fallen(const double *x)
  {
    return __builtin_fabs(x[SLA]) > GRLX_PI/8 || __builtin_fabs(x[HA] - 2 * x[SLA]) > GRLX_PI/4;
  }
  __device__ static __forceinline__ int observe(const DevParams &, const double *x, double *obs)
  { // :292-329, observe = [1,1,1,1,1,0,0], steps = 0
    obs[0] = x[SLA];
    obs[1] = x[HA] - 2 * x[SLA];
    obs[2] = x[SLAR];
    obs[3] = x[HAR] - 2 * x[SLAR];
    obs[4] = x[CHANGED] > 0.5 ? 1. : 0.;
    if (fallen(x)) return 2;
    if (x[TIME] > x[TIMEOUT]) return 1;
    return 0;
  }
  __device__ static __forceinline__ double evaluate(const DevParams &P, const double *, double, const double *next)
  { // :331-344
    double reward = -1;
    if (next[CHANGED] > 0.5) reward = fmin(50 * 4 * psin(next[SLA]), 30.);
    if (fallen(next))
      if (P.negative_reward != 0) reward = P.negative_reward;
    return reward;
  }
};

template <int ENV> struct HasCustomModel { static constexpr bool value = false; };
template <> struct HasCustomModel<GRLX_ENV_COMPASS_WALKER> { static constexpr bool value = true; };

// DynamicalModel::step (modeled.cpp:254-276): classical RK4 sub-steps.
// The last state component is time (xd = 1 in every supported dynamics, and no eom reads
// it), so its stage values are the constant h and its update the constant
// (h + 2h + 2h + h)/6 -- the same operations the reference performs, hoisted.
// eom with or without shared-out sin/cos evaluations (only the acrobot has a split form)
template <int ENV, typename SH>
__device__ __forceinline__ void env_eom(const typename Env<ENV>::Consts &ec, const double *x, double u, double *xd, const SH &sh)
{
  if constexpr (SH::kSplit && ENV == GRLX_ENV_ACROBOT)
    Env<ENV>::eom(ec, x, u, xd, sh);
  else
    Env<ENV>::eom(ec, x, u, xd);
}

template <int ENV, bool PIN, typename SH = NoShare>
__device__ __forceinline__ void rk4_step(const DevParams &P, const double *x, double u, double *next, const SH &sh = SH())
{
  constexpr int S = Env<ENV>::S, SD = S - 1;
  const double h = P.h;
  const double tinc = (((h + 2 * h) + 2 * h) + h) / 6;
  double xd[S], k1[SD], k2[SD], k3[SD], k4[SD], t[S];
#pragma unroll
  for (int i = 0; i < S; ++i) { next[i] = x[i]; t[i] = x[i]; }
  const typename Env<ENV>::Consts ec = Env<ENV>::template consts<PIN>();   // PIN: constants held in vector registers
  for (int ii = 0; ii < P.integration_steps; ++ii)
  {
    env_eom<ENV>(ec, next, u, xd, sh);
#pragma unroll
    for (int i = 0; i < SD; ++i) { k1[i] = h * xd[i]; t[i] = next[i] + k1[i] / 2; }
    env_eom<ENV>(ec, t, u, xd, sh);
#pragma unroll
    for (int i = 0; i < SD; ++i) { k2[i] = h * xd[i]; t[i] = next[i] + k2[i] / 2; }
    env_eom<ENV>(ec, t, u, xd, sh);
#pragma unroll
    for (int i = 0; i < SD; ++i) { k3[i] = h * xd[i]; t[i] = next[i] + k3[i]; }
    env_eom<ENV>(ec, t, u, xd, sh);
#pragma unroll
    for (int i = 0; i < SD; ++i)
    {
      k4[i] = h * xd[i];
      next[i] = next[i] + div6(k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
    }
    next[SD] = next[SD] + tinc;
  }
}

// ModeledEnvironment::step (modeled.cpp:160-213), window 1, no delta, discrete_time 1
// PIN: hold the dynamics' constants in vector registers across the integration loop (pays at one
// wave per SIMD, costs registers)
// SH: NoShare, or a LaneShare when several lanes integrate the same replica (rollout kernels)
template <int ENV, bool PIN = true, typename SH = NoShare>
__device__ __forceinline__ void env_step(const DevParams &P, double *x, double action, double *obs, double &reward, int &terminal, uint32_t &status,
                                         const SH &sh = SH())
{
  constexpr int S = Env<ENV>::S;
  double next[S];
  if constexpr (HasCustomModel<ENV>::value)
    Env<ENV>::template model_step<SH, PIN>(P, x, Env<ENV>::actuate(action), next, sh);  // model/compass_walker integrates itself
  else
    rk4_step<ENV, PIN, SH>(P, x, Env<ENV>::actuate(action), next, sh);
  terminal = Env<ENV>::observe(P, next, obs);
  reward = Env<ENV>::evaluate(P, x, action, next);
  // the branch-free sin/cos need |angle| < 2^20; 2^19 at step ends leaves room for the stages
  if (!Env<ENV>::in_domain(next)) status |= ST_DOMAIN;
#pragma unroll
  for (int i = 0; i < S; ++i) x[i] = next[i];
}


} // namespace grlx
