// grlx_api.cpp -- the C ABI declared in include/grlx.h.
//
// Host-side logic only: validation (grl's bad_param conditions become
// GRLX_ERR_INVALID with a message), the reference's instantiate-order RNG
// seeding, device memory management and kernel launches.  No compute happens on
// the host: without a HIP device every compute entry point fails.
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "grlx_internal.h"

using namespace grlx;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

} // namespace
namespace grlx {
void set_last_error(const char *msg) { g_err = msg; }     // other translation units of the library (grlx_fqi.hip)
}
namespace {
#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e__ = (expr);                                                            \
    if (e__ != hipSuccess)                                                              \
      return fail(GRLX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

bool have_device()
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return false; }
  return n > 0;
}

// host copy of the drand48-family LCG (utils.h:84-137), used only to seed the replicas
constexpr uint64_t kA = 0x5DEECE66DULL, kC = 0xBULL, kM = (1ULL << 48) - 1;
inline uint64_t h_seed(long s) { return ((((uint64_t)s) & 0xFFFFFFFFULL) << 16) | 0x330EULL; }   // srand48
inline uint64_t h_next(uint64_t x) { return (kA * x + kC) & kM; }
inline uint64_t h_jump(uint64_t x, uint64_t n)
{
  uint64_t a = kA, c = kC;
  while (n)
  {
    if (n & 1) x = (a * x + c) & kM;
    c = ((a + 1) * c) & kM;
    a = (a * a) & kM;
    n >>= 1;
  }
  return x;
}

int env_dims(int env, int *S, int *D)
{
  switch (env)
  {
    case GRLX_ENV_PENDULUM: *S = 3; *D = 2; return GRLX_OK;
    case GRLX_ENV_ACROBOT: *S = 5; *D = 4; return GRLX_OK;
    case GRLX_ENV_CART_POLE: *S = 5; *D = 4; return GRLX_OK;
    case GRLX_ENV_COMPASS_WALKER: *S = 11; *D = 5; return GRLX_OK;
    case GRLX_ENV_CART_POLE_BALANCING: *S = 5; *D = 4; return GRLX_OK;
    default: return GRLX_ERR_INVALID;
  }
}

// TileCodingProjector::configure (tile_coding.cpp:45-80)
int make_tile_params(const grlx_tile_spec &ts, TileParams *tp)
{
  if (ts.dims < 1 || ts.dims > GRLX_MAX_DIMS) return fail(GRLX_ERR_INVALID, "projector/tile_coding:resolution (dims %d)", ts.dims);
  if (ts.tilings < 1 || ts.tilings > 32) return fail(GRLX_ERR_INVALID, "projector/tile_coding:tilings (%d)", ts.tilings);
  if (ts.memory < 1 || ts.memory >= (1 << 26)) return fail(GRLX_ERR_INVALID, "projector/tile_coding:memory (1 .. 2^26-1 supported)");
  if (ts.safe < 0 || ts.safe > 2) return fail(GRLX_ERR_INVALID, "projector/tile_coding:safe must be 0, 1 or 2");
  memset(tp, 0, sizeof(*tp));
  tp->T = ts.tilings;
  tp->D = ts.dims;
  tp->memory = ts.memory;
  for (int i = 0; i < ts.dims; ++i)
  {
    if (!(ts.resolution[i] > 0)) return fail(GRLX_ERR_INVALID, "projector/tile_coding:resolution[%d]", i);
    tp->scaling[i] = ts.tilings / ts.resolution[i];
    double w = ts.wrapping[i] * tp->scaling[i];
    if (fabs(w - round(w)) > 0.001)
      return fail(GRLX_ERR_INVALID, "projector/tile_coding:wrapping (scaled wrapping for dimension %d (%.5f) is not an integer)", i, w);
    tp->wrap[i] = (int)round(w);
  }
  return GRLX_OK;
}

int make_linear_params(const grlx_linear_spec &ls, uint64_t draws_before, LinearParams *lp)
{
  lp->init_min = ls.init_min;
  lp->init_range = ls.init_max - ls.init_min;     // utils.h:110-113: a + get()*(b-a)
  lp->out_min = ls.output_min;
  lp->out_max = ls.output_max;
  lp->limit = ls.limit;
  lp->draws_before = draws_before;
  return GRLX_OK;
}

int make_params(const grlx_config &c, DevParams *P)
{
  memset(P, 0, sizeof(*P));
  if (c.struct_size != sizeof(grlx_config)) return fail(GRLX_ERR_INVALID, "grlx_config.struct_size %u != %zu (ABI mismatch)", c.struct_size, sizeof(grlx_config));
  int S, D;
  const bool external = c.env == GRLX_ENV_EXTERNAL;        // the environment is the caller's: per-step agent entry points only
  if (external)
  {
    S = 0;
    D = (c.agent == GRLX_AGENT_AC) ? c.projector.dims : c.projector.dims - 1;
    if (D < 1 || D >= GRLX_MAX_DIMS) return fail(GRLX_ERR_INVALID, "projector/tile_coding:resolution (an external environment's observation has %d dimensions)", D);
    if (c.agent != GRLX_AGENT_SARSA && c.agent != GRLX_AGENT_Q && c.agent != GRLX_AGENT_EXPECTED_SARSA && c.agent != GRLX_AGENT_AC)
      return fail(GRLX_ERR_INVALID, "an external environment is served by the per-step agent entry points: SARSA / Q / Expected SARSA or actor-critic");
  }
  else if (env_dims(c.env, &S, &D) != GRLX_OK) return fail(GRLX_ERR_INVALID, "environment %d is not supported by the fused path", c.env);
  if (c.agent != GRLX_AGENT_SARSA && c.agent != GRLX_AGENT_Q && c.agent != GRLX_AGENT_AC && c.agent != GRLX_AGENT_EXPECTED_SARSA &&
      c.agent != GRLX_AGENT_ADVANTAGE && c.agent != GRLX_AGENT_QV) return fail(GRLX_ERR_INVALID, "agent %d is not supported by the fused path", c.agent);
  if (c.agent == GRLX_AGENT_ADVANTAGE)
  {
    if (!(c.kappa > 0)) return fail(GRLX_ERR_INVALID, "predictor/critic/advantage:kappa");
    if ((c.env != GRLX_ENV_PENDULUM && c.env != GRLX_ENV_ACROBOT) || c.action_steps != 3)
      return fail(GRLX_ERR_INVALID, "advantage learning is built for the pendulum and the acrobot with 3 actions");
  }
  const bool ac = c.agent == GRLX_AGENT_AC;
  const bool qv = c.agent == GRLX_AGENT_QV;
  if (qv && ((c.env != GRLX_ENV_PENDULUM && c.env != GRLX_ENV_ACROBOT) || c.action_steps != 3))
    return fail(GRLX_ERR_INVALID, "predictor/critic/qv is built for the pendulum and the acrobot with 3 actions");
  if (qv && !(c.beta > 0)) return fail(GRLX_ERR_INVALID, "predictor/critic/qv:beta");
  if (ac && !external && c.env != GRLX_ENV_CART_POLE && c.env != GRLX_ENV_PENDULUM) return fail(GRLX_ERR_INVALID, "actor-critic is built for cart-pole and pendulum");
  if (!external)
  {
  if (c.discrete_time != 1) return fail(GRLX_ERR_INVALID, "environment/modeled:discrete_time must be 1");
  if (!(c.control_step >= 0.00001)) return fail(GRLX_ERR_INVALID, "model/dynamical:control_step");
  if (c.integration_steps < 1 || c.integration_steps > 1000) return fail(GRLX_ERR_INVALID, "model/dynamical:integration_steps (1..1000 supported)");
  // The rollout kernels leave the step loop only when the task reports a terminal state, i.e. (at the
  // latest) on time > timeout: a non-finite or huge timeout would be a kernel that never ends -- on a
  // GPU a hang, not a killable loop.  Episodes are capped at GRLX_MAX_EPISODE_STEPS control steps.
  if (!std::isfinite(c.timeout) || c.timeout < 0) return fail(GRLX_ERR_INVALID, "task:timeout must be finite and >= 0");
  if (!std::isfinite(c.control_step) || c.control_step > 1e6) return fail(GRLX_ERR_INVALID, "model/dynamical:control_step");
  {
    const double horizon = (c.env == GRLX_ENV_COMPASS_WALKER) ? 2 * c.timeout : c.timeout;   // test episodes of the walker: 2 x timeout
    if (std::ceil(horizon / c.control_step) + 1 > (double)GRLX_MAX_EPISODE_STEPS)
      return fail(GRLX_ERR_INVALID, "task:timeout / control_step = %.0f steps per episode exceeds GRLX_MAX_EPISODE_STEPS (%d)",
                  std::ceil(horizon / c.control_step), GRLX_MAX_EPISODE_STEPS);
  }
  }
  if (!ac && (c.action_steps < 1 || c.action_steps > GRLX_MAX_ACTIONS)) return fail(GRLX_ERR_INVALID, "discretizer/uniform:steps (1..%d supported)", GRLX_MAX_ACTIONS);
  if (!ac && !qv && c.agent != GRLX_AGENT_ADVANTAGE && c.trace != GRLX_TRACE_ACCUMULATING && c.env != GRLX_ENV_CART_POLE_BALANCING)
  { // the instantiations of rollout_kernel (launch_rollout): anything else has no kernel to run
    const bool built = ((c.env == GRLX_ENV_PENDULUM || external) && (c.action_steps == 3 || c.action_steps == 5)) ||
                       ((c.env == GRLX_ENV_ACROBOT || c.env == GRLX_ENV_CART_POLE || c.env == GRLX_ENV_COMPASS_WALKER) && c.action_steps == 3);
    if (!built)
      return fail(GRLX_ERR_INVALID, "discretizer/uniform:steps = %d is not built for this environment (pendulum: 3 or 5 actions; "
                                    "acrobot, cart-pole, compass walker: 3 actions)", c.action_steps);
  }
  if (ac && c.trace != GRLX_TRACE_REPLACING && c.trace != GRLX_TRACE_NONE) return fail(GRLX_ERR_INVALID, "trace type %d is not supported by the fused path", c.trace);
  if (c.trace == GRLX_TRACE_ACCUMULATING)
  { // its own (uncached, read-modify-write) kernel: SARSA / Q / Expected SARSA on the pendulum and the acrobot
    if ((c.agent != GRLX_AGENT_SARSA && c.agent != GRLX_AGENT_Q && c.agent != GRLX_AGENT_EXPECTED_SARSA) ||
        (c.env != GRLX_ENV_PENDULUM && c.env != GRLX_ENV_ACROBOT) || c.action_steps != 3)
      return fail(GRLX_ERR_INVALID, "trace/enumerated/accumulating is built for SARSA / Q / Expected SARSA on the pendulum and the acrobot with 3 actions");
  }
  else if (c.trace != GRLX_TRACE_NONE && c.trace != GRLX_TRACE_REPLACING) return fail(GRLX_ERR_INVALID, "trace type %d is not supported by the fused path", c.trace);

  P->test_interval = c.test_interval;
  P->env = c.env;
  P->agent = c.agent;
  P->trace_kind = c.trace;
  P->integration_steps = c.integration_steps;
  P->h = external ? 0. : c.control_step / (double)(size_t)c.integration_steps;      // modeled.cpp:257
  P->timeout = c.timeout;
  P->randomization = c.randomization;

  P->control_step = c.control_step;
  P->slope_angle = c.slope_angle;
  P->initial_state_variation = c.initial_state_variation;
  P->negative_reward = c.negative_reward;
  // CSWModel::setTiming (SWModel.cpp:126-130): whole microseconds, then the sub-step of singleStep (:151)
  P->walker_dt = external ? 0. : 1.0E-6 * (double)(uint64_t)floor((c.control_step + 0.5E-6) * 1E6) / c.integration_steps;
  P->end_stop_penalty = c.end_stop_penalty;
  P->action_penalty = c.action_penalty;
  P->action_min = c.action_min;
  P->action_max = c.action_max;
  // UniformDiscretizer::configure (uniform.cpp:60-95)
  P->A = ac ? 0 : c.action_steps;
  if (!ac)
  {
    double range = c.action_max - c.action_min;
    double delta = range / ((double)c.action_steps - 1);
    if (std::isnan(delta)) delta = 0.;
    for (int k = 0; k < c.action_steps; ++k)
      P->actions[k] = c.action_min + delta * k;
  }

  int rc = make_tile_params(c.projector, &P->tile);
  if (rc != GRLX_OK) return rc;
  if (P->tile.T != kLanesPerReplica) return fail(GRLX_ERR_INVALID, "projector/tile_coding:tilings must be %d on the fused path", kLanesPerReplica);
  if (!ac && P->tile.D != D + 1) return fail(GRLX_ERR_INVALID, "projector/tile_coding:resolution must have %d entries (observation + action)", D + 1);
  make_linear_params(c.representation, 0, &P->lin);
  if (ac)
  { // policy/action + predictor/ac/action: actor table first in the yaml, then the critic's (ac_tc.yaml:36-65)
    if (P->tile.D != D) return fail(GRLX_ERR_INVALID, "critic projector/tile_coding:resolution must have %d entries (observation)", D);
    rc = make_tile_params(c.actor_projector, &P->tile_actor);
    if (rc != GRLX_OK) return rc;
    if (P->tile_actor.T != kLanesPerReplica) return fail(GRLX_ERR_INVALID, "actor projector/tile_coding:tilings must be %d on the fused path", kLanesPerReplica);
    if (P->tile_actor.D != D) return fail(GRLX_ERR_INVALID, "actor projector/tile_coding:resolution must have %d entries (observation)", D);
    make_linear_params(c.actor_representation, 0, &P->lin_actor);
    P->lin.draws_before = (uint64_t)c.actor_projector.memory;         // both tables draw from one thread-local stream
    P->actor_alpha = c.actor_alpha;
    P->sigma = c.sigma;
    P->theta = c.theta;
    P->ac_decay_rate = c.ac_decay_rate;
    P->ac_decay_min = c.ac_decay_min;
    P->ac_step_limit = c.ac_step_limit;
    P->ac_update_method = c.ac_update_method;
    if (c.ac_update_method != 0 && c.ac_update_method != 1) return fail(GRLX_ERR_INVALID, "predictor/ac/action:update_method");
    { // equal tile codings (cfg/cart_pole/ac_tc.yaml: the critic's projector copies the actor's resolution and memory): twin tables
      const TileParams &a = P->tile_actor, &b = P->tile;
      bool same = a.T == b.T && a.D == b.D && a.memory == b.memory && c.actor_projector.safe == c.projector.safe;
      for (int i = 0; i < a.D && same; ++i) same = a.scaling[i] == b.scaling[i] && a.wrap[i] == b.wrap[i];
      P->twin_tables = same ? 1 : 0;
    }
    if (!(c.action_min < c.action_max)) return fail(GRLX_ERR_INVALID, "policy/action:{output_min,output_max}");
  }

  if (qv)
  { // predictor/critic/qv: table 0 = the policy's Q representation (first in the yaml), table 1 = v_representation
    rc = make_tile_params(c.actor_projector, &P->tile_actor);
    if (rc != GRLX_OK) return rc;
    if (P->tile_actor.T != kLanesPerReplica) return fail(GRLX_ERR_INVALID, "v_projector/tile_coding:tilings must be %d on the fused path", kLanesPerReplica);
    if (P->tile_actor.D != D) return fail(GRLX_ERR_INVALID, "v_projector/tile_coding:resolution must have %d entries (observation)", D);
    make_linear_params(c.actor_representation, (uint64_t)c.projector.memory, &P->lin_actor);   // drawn after the Q table
    P->beta = c.beta;
  }

  if (c.projector.safe >= 1)
  { // claim table of the tile coding: its own (plain) kernel
    if ((c.agent != GRLX_AGENT_SARSA && c.agent != GRLX_AGENT_Q) || c.env == GRLX_ENV_CART_POLE_BALANCING ||
        c.action_steps != 3 || c.trace == GRLX_TRACE_ACCUMULATING)
      return fail(GRLX_ERR_INVALID, "projector/tile_coding:safe >= 1 is built for predictor/critic/sarsa and predictor/critic/q with 3 actions, replacing or no trace");
    if (c.env != GRLX_ENV_PENDULUM && c.target_interval > 0) return fail(GRLX_ERR_INVALID, "safe >= 1 together with a target network is built for the pendulum");
    P->tile_safe = c.projector.safe;         // 2: the policy's batch projections claim too
  }
  if ((ac || qv) && c.actor_projector.safe != 0) return fail(GRLX_ERR_INVALID, "projector/tile_coding:safe on the second table");
  if (c.target_interval < 0 || !std::isfinite(c.target_tau)) return fail(GRLX_ERR_INVALID, "representation/parameterized/linear:{interval,tau}");
  if (c.target_interval > 0)
  { // target network of the Q table: its own (plain) kernel
    if (c.agent != GRLX_AGENT_SARSA && c.agent != GRLX_AGENT_Q)
      return fail(GRLX_ERR_INVALID, "representation/parameterized/linear:interval (a target network is built for predictor/critic/sarsa and predictor/critic/q)");
    if (c.env == GRLX_ENV_CART_POLE_BALANCING || c.action_steps != 3 || c.trace == GRLX_TRACE_ACCUMULATING)
      return fail(GRLX_ERR_INVALID, "representation/parameterized/linear:interval (built for 3 actions, replacing or no trace)");
    if (c.target_tau < 0 || c.target_tau > 1) return fail(GRLX_ERR_INVALID, "representation/parameterized/linear:tau");
    P->target_interval = c.target_interval;
    P->target_tau = c.target_tau;
    P->lin.draws_before = (uint64_t)c.projector.memory;     // the target is instantiated -- and draws -- first (representation.h:186-190)
  }
  if (c.test_trials < 0) return fail(GRLX_ERR_INVALID, "experiment/online_learning:test_trials");
  P->test_trials = c.test_trials > 1 ? c.test_trials : 1;
  if (P->test_trials > 1 && c.tap_capacity > 0) return fail(GRLX_ERR_INVALID, "test_trials > 1 is not available with taps");
  P->epsilon = c.epsilon;
  P->decay_rate = c.decay_rate;
  P->decay_min = c.decay_min;
  P->alpha = c.alpha;
  P->kappa = c.kappa;
  P->gamma = c.gamma;                     // pow(gamma, tau) with tau = 1
  P->gl = c.gamma * c.lambda;             // pow(gamma*lambda, tau) with tau = 1
  if (c.trace == GRLX_TRACE_ACCUMULATING)
  { // trace.h:249-261: pops while the total decay < 1e-4; the position trace holds 20 entries
    if (!(P->gl > 0 && P->gl < 1)) return fail(GRLX_ERR_INVALID, "predictor: gamma*lambda must be in (0,1) with a trace");
    double tot = 1;
    int n = 0;
    while (tot >= 0.0001 && n <= 20) { tot *= P->gl; n++; }
    if (n > 20) return fail(GRLX_ERR_INVALID, "predictor: gamma*lambda = %g needs an accumulating trace longer than 20 entries", P->gl);
  }
  if (c.trace == GRLX_TRACE_REPLACING)
  { // the register trace holds kMaxTrace entries: the reference pops while the total decay < 0.01 (trace.h:227-231)
    if (!(P->gl > 0 && P->gl < 1)) return fail(GRLX_ERR_INVALID, "predictor: gamma*lambda must be in (0,1) with a trace");
    double tot = 1;
    int n = 0;
    while (tot >= 0.01 && n <= kMaxTrace) { tot *= P->gl; n++; }
    if (n > kMaxTrace) return fail(GRLX_ERR_INVALID, "predictor: gamma*lambda = %g needs a trace longer than %d entries", P->gl, kMaxTrace);
  }
  return GRLX_OK;
}

} // namespace

struct grlx_ctx {
  grlx_config cfg;
  DevParams   P;
  int         S, D;
  Entry        *tables = nullptr;
  ReplicaState *states = nullptr;
  double       *row_reward = nullptr, *row_time = nullptr;
  int64_t      *row_steps = nullptr, *row_trial = nullptr;
  grlx_tap     *taps = nullptr;
  uint32_t     *tap_count = nullptr;
  uint64_t     *scratch = nullptr;        // 8 x u64
  uint32_t     *queue = nullptr;          // work queue of the wide actor-critic kernel: next unstarted replica
  uint32_t     *max_load = nullptr;       // fullest table of the context after the last grlx_run (max_load_kernel)
  uint32_t     logC_max = 26;             // growth bound of the sparse tables
  unsigned long long *diag = nullptr;
  uint32_t     *trace_state = nullptr;
  double       *tvals = nullptr;          // target network values per table position (target_interval > 0)
  // loaded policy images (grlx_load_weights): image k serves the replicas r with image_of[table][r] == k;
  // an image no replica refers to any more is freed at the next load, the rest with the context
  std::vector<double *> images;
  std::vector<int>      image_of[2];
  std::vector<double *> target_images;    // per replica: the target network's dense vector after a load with tau != 0 (ReplicaState::target_base)
  hipStream_t  run_stream = nullptr;      // stream of the last grlx_run / grlx_curve_stats
  bool         run_pending = false;       // ... with work possibly still in flight on it
  int          n_tables = 1;
  int64_t      trials_run = 0;
  int          last_kernel = GRLX_KERNEL_NONE;
  bool         poison = false;            // GRLX_POISON_REGISTERS=<pattern>: diagnostic, see launch_poison_registers
  uint32_t     poison_pattern = 0;
  // the environment server of the pendulum rollout kernels (grlx_env_server.h): mailboxes, its stream, and the fork / join events
  int          env_server = 1;            // GRLX_ENV_SERVER=0 turns it off
  void         *park = nullptr;           // rotating actor-critic kernel: parked lane state of the third sub-batch, per wave
  EnvMail      *env_mail = nullptr;
  size_t       env_mail_bytes = 0;        // per replica: kEnvMailBytes (pendulum kernels) or kWideMailBytes (wide kernels)
  hipStream_t  srv_stream = nullptr;
  hipEvent_t   srv_go = nullptr, srv_done = nullptr;
  // per-step entry points (grlx_env_start / _advance, grlx_agent_start / _step / _end): agent state between calls and the staging
  // buffers of their host arguments, created at the first such call
  AgentRep     *agent_rep = nullptr;
  uint32_t     *agent_lane = nullptr;
  double       *stage_f64 = nullptr;      // [N][GRLX_MAX_DIMS] obs | [N] reward | [N] action
  int32_t      *stage_i32 = nullptr;      // [N] active | [N] terminal
  uint64_t     step_calls = 0;            // agent calls since the last look at the tables' load
};

// small RAII helper for the copy-in / copy-out entry points
namespace {
struct DevBuf {
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
  template <typename T> T *as() { return (T *)p; }
};
}

// Read-back and fine-grained entry points run on the NULL stream or copy synchronously: wait for the rollouts
// of the last grlx_run first (its stream may be non-blocking with respect to the NULL stream).
static int drain(grlx_ctx *ctx)
{
  if (ctx->run_pending)
  {
    HIP_TRY(hipStreamSynchronize(ctx->run_stream));
    ctx->run_pending = false;
  }
  return GRLX_OK;
}
#define DRAIN(ctx)                          \
  do {                                      \
    int rc__ = drain(ctx);                  \
    if (rc__ != GRLX_OK) return rc__;       \
  } while (0)

extern "C" {

const char *grlx_last_error(void) { return g_err.c_str(); }
int grlx_abi_version(void) { return GRLX_ABI_VERSION; }
#ifndef GRLX_BUILD_PIPELINE
#define GRLX_BUILD_PIPELINE ""
#endif
const char *grlx_build_pipeline(void) { return GRLX_BUILD_PIPELINE; }

int grlx_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

void grlx_config_pendulum_sarsa(grlx_config *c)
{ // the reference's tests/pendulum-sarsa-tc.yaml
  memset(c, 0, sizeof(*c));
  c->struct_size = sizeof(grlx_config);
  c->n_replicas = 1;
  c->test_interval = 10;
  c->env = GRLX_ENV_PENDULUM;
  c->control_step = 0.03;
  c->integration_steps = 5;
  c->discrete_time = 1;
  c->timeout = 2.99;
  c->randomization = 0;
  c->action_min = -3;
  c->action_max = 3;
  c->action_steps = 3;
  c->agent = GRLX_AGENT_SARSA;
  c->projector.tilings = 16;
  c->projector.memory = 8388608;
  c->projector.dims = 3;
  c->projector.resolution[0] = 0.31415;
  c->projector.resolution[1] = 3.1415;
  c->projector.resolution[2] = 3;
  c->projector.wrapping[0] = 6.283;
  c->representation.init_min = 0;
  c->representation.init_max = 1;
  c->representation.output_min = -DBL_MAX;
  c->representation.output_max = DBL_MAX;
  c->representation.limit = 1;
  c->epsilon = 0.05;
  c->decay_rate = 1;
  c->decay_min = 0;
  c->alpha = 0.2;
  c->gamma = 0.97;
  c->lambda = 0.65;
  c->trace = GRLX_TRACE_REPLACING;
  c->ac_step_limit = -1;
  c->table_log2_capacity = 0;
  c->max_rows = 256;
  c->tap_replica = -1;
  c->tap_capacity = 0;
  c->end_stop_penalty = 1;
  c->action_penalty = 0;
  c->slope_angle = 0.004;
  c->initial_state_variation = 0.2;
  c->negative_reward = -100;
}

void grlx_config_cart_pole_ac(grlx_config *c)
{ // the reference's cfg/cart_pole/ac_tc.yaml
  grlx_config_pendulum_sarsa(c);
  c->env = GRLX_ENV_CART_POLE;
  c->control_step = 0.05;
  c->integration_steps = 5;
  c->timeout = 9.99;
  c->randomization = 0;
  c->end_stop_penalty = 0;
  c->action_penalty = 0;
  c->action_min = -15;
  c->action_max = 15;
  c->action_steps = 0;
  c->agent = GRLX_AGENT_AC;
  const double res[4] = {2.5, 0.157075, 2.5, 1.57075}, wrap[4] = {0, 6.283, 0, 0};
  grlx_tile_spec *ts[2] = {&c->projector, &c->actor_projector};
  for (int t = 0; t < 2; ++t)
  {
    memset(ts[t], 0, sizeof(grlx_tile_spec));
    ts[t]->tilings = 16;
    ts[t]->memory = 8388608;
    ts[t]->dims = 4;
    for (int i = 0; i < 4; ++i) { ts[t]->resolution[i] = res[i]; ts[t]->wrapping[i] = wrap[i]; }
  }
  c->actor_representation.init_min = 0;
  c->actor_representation.init_max = 1;
  c->actor_representation.output_min = -15;       // output_min/max: task action_min/max
  c->actor_representation.output_max = 15;
  c->actor_representation.limit = 1;
  c->alpha = 0.2; c->gamma = 0.97; c->lambda = 0.65;   // predictor/critic/td
  c->trace = GRLX_TRACE_REPLACING;
  c->actor_alpha = 0.01;                          // predictor/ac/action:alpha
  c->sigma = 5;
  c->theta = 1;
  c->ac_decay_rate = 1;
  c->ac_decay_min = 0;
  c->ac_update_method = 0;                        // proportional
  c->ac_step_limit = -1;
  // initial size; the tables grow between launches (grlx.h).  2^16 entries hold what ONE launch of 32 cart-pole trials creates from
  // empty tables (about 25 000 slots per table) at a load the 4-way buckets take; 16384 replicas then start at 2 x 16 GiB instead of the
  // 2 x 64 GiB that 2^18 reserved up front.  Callers that must not re-hash between timed launches (bench.py) pass their size.
  c->table_log2_capacity = 16;
}

int grlx_env_dims(int env, int *state_dims, int *obs_dims)
{
  int S, D;
  if (env_dims(env, &S, &D) != GRLX_OK) return fail(GRLX_ERR_INVALID, "unknown environment %d", env);
  if (state_dims) *state_dims = S;
  if (obs_dims) *obs_dims = D;
  return GRLX_OK;
}

int grlx_create(const grlx_config *cfg, const int64_t *seeds, grlx_ctx **out)
{
  if (!cfg || !seeds || !out) return fail(GRLX_ERR_INVALID, "null argument");
  *out = nullptr;
  if (cfg->n_replicas < 1) return fail(GRLX_ERR_INVALID, "n_replicas must be >= 1");
  if (cfg->max_rows < 1) return fail(GRLX_ERR_INVALID, "max_rows must be >= 1");
  // 16 for the TD agents: the four-sub-batch instantiations of rollout_wide_kernel (the acrobot and the compass walker, three actions)
  const bool td16 = cfg->replicas_per_wave == 16 && (cfg->env == GRLX_ENV_ACROBOT || cfg->env == GRLX_ENV_COMPASS_WALKER) && cfg->action_steps == 3 &&
                    (cfg->agent == GRLX_AGENT_SARSA || cfg->agent == GRLX_AGENT_Q || cfg->agent == GRLX_AGENT_EXPECTED_SARSA);
  // 32: the eight-sub-batch instantiation (the compass walker: two lanes per replica suffice for its environment phase)
  const bool td32 = cfg->replicas_per_wave == 32 && cfg->env == GRLX_ENV_COMPASS_WALKER && cfg->action_steps == 3 &&
                    (cfg->agent == GRLX_AGENT_SARSA || cfg->agent == GRLX_AGENT_Q || cfg->agent == GRLX_AGENT_EXPECTED_SARSA);
  if (cfg->replicas_per_wave != 0 && cfg->replicas_per_wave != 4 && cfg->replicas_per_wave != 8 && !td16 && !td32 &&
      !((cfg->replicas_per_wave == 12 || cfg->replicas_per_wave == 16) && cfg->agent == GRLX_AGENT_AC))
    return fail(GRLX_ERR_INVALID, "replicas_per_wave must be 0 (automatic), 4 or 8 (12: actor-critic only; 16: actor-critic, and the TD agents on the acrobot / the compass walker with 3 actions; 32: the TD agents on the compass walker)");
  if (cfg->wave_limit < 0) return fail(GRLX_ERR_INVALID, "wave_limit must be 0 (automatic) or positive");
  if (cfg->tap_deferred && cfg->tap_replica >= 0 && cfg->tap_capacity > 0)
  {
    const bool td = (cfg->agent == GRLX_AGENT_SARSA || cfg->agent == GRLX_AGENT_Q || cfg->agent == GRLX_AGENT_EXPECTED_SARSA);
    const bool built = (cfg->env == GRLX_ENV_PENDULUM && (cfg->action_steps == 3 || cfg->action_steps == 5)) || (cfg->env == GRLX_ENV_ACROBOT && cfg->action_steps == 3);
    if (!td || cfg->trace == GRLX_TRACE_ACCUMULATING || !built)
      return fail(GRLX_ERR_INVALID, "tap_deferred is built for SARSA / Q / Expected SARSA on the pendulum (3 or 5 actions) and the acrobot (3 actions)");
  }
  DevParams P;
  int rc = make_params(*cfg, &P);
  if (rc != GRLX_OK) return rc;
  if (cfg->env == GRLX_ENV_CART_POLE_BALANCING)
    return fail(GRLX_ERR_INVALID, "task/cart_pole/balancing is served by grlx_env_step only (no fused TD rollout is built for it)");
  if (!have_device()) return fail(GRLX_ERR_NO_DEVICE, "no HIP device: grlx has no CPU fallback");

  grlx_ctx *ctx = new grlx_ctx();
  ctx->cfg = *cfg;
  if (const char *es = getenv("GRLX_ENV_SERVER")) ctx->env_server = atoi(es) != 0;
  if (const char *pz = getenv("GRLX_POISON_REGISTERS"))
  { // diagnostic (tests): every rollout launch is preceded by a kernel that fills the register files with this pattern
    ctx->poison = pz[0] != 0;
    ctx->poison_pattern = (uint32_t)strtoul(pz, nullptr, 0);
  }
  if (cfg->env == GRLX_ENV_EXTERNAL) { ctx->S = 0; ctx->D = (cfg->agent == GRLX_AGENT_AC) ? cfg->projector.dims : cfg->projector.dims - 1; }
  else env_dims(cfg->env, &ctx->S, &ctx->D);
  const int N = cfg->n_replicas;
  uint32_t logC = cfg->table_log2_capacity ? (uint32_t)cfg->table_log2_capacity : 17u;
  if (logC < 8 || logC > 26) { delete ctx; return fail(GRLX_ERR_INVALID, "table_log2_capacity must be in 8..26"); }
  if (cfg->table_log2_max != 0 && (cfg->table_log2_max < (int)logC || cfg->table_log2_max > 26))
  { delete ctx; return fail(GRLX_ERR_INVALID, "table_log2_max must be 0 or in table_log2_capacity..26"); }
  ctx->logC_max = cfg->table_log2_max ? (uint32_t)cfg->table_log2_max : 26u;
  P.n_replicas = N;
  P.logC = logC;
  P.max_rows = cfg->max_rows;
  P.no_specialisation = cfg->force_generic;
  P.tap_replica = cfg->tap_replica;
  P.tap_capacity = cfg->tap_replica >= 0 ? cfg->tap_capacity : 0;
  P.tap_starts = cfg->tap_starts != 0 ? 1 : 0;
  P.tap_deferred = (cfg->tap_deferred != 0 && P.tap_capacity > 0) ? 1 : 0;
  { // replicas per wave: wide waves once the batch outnumbers the SIMDs four to one (taps and stamps: always 4)
    const bool has_wide = cfg->target_interval == 0 && cfg->projector.safe == 0 && (cfg->agent == GRLX_AGENT_SARSA || cfg->agent == GRLX_AGENT_Q || cfg->agent == GRLX_AGENT_EXPECTED_SARSA ||
                           cfg->agent == GRLX_AGENT_AC) && cfg->trace != GRLX_TRACE_ACCUMULATING;
    int rpw = cfg->replicas_per_wave;
    hipDeviceProp_t prop;
    int dev = 0;
    int simds = 1024;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) simds = 4 * prop.multiProcessorCount;
    // 8 once the 4-replica waves outnumber the SIMDs.  (Round 2's 16-slot actor-critic kernel parked four sub-batches in 66 KB of LDS,
    // two waves per CU: 213 M vs 329 M env-steps/s at 16384 cart-pole replicas.  Since round 3 the sub-batches beyond the second park in
    // device memory)
    if (rpw == 0) rpw = ((N + kReplicasPerWave - 1) / kReplicasPerWave > simds) ? 8 : 4;
    // actor-critic, more than 8 replicas per SIMD: more slots per wave share one environment phase (grlx_rollout_ac_wide.h; 16384 cart-pole
    // replicas: 373 M env-steps/s with 8 slots, 405 M with 12, 425 M with 16).  16 slots for 15 or more replicas per SIMD; 12, rotated
    // trial by trial over the wave's own replicas, for 9 to 14 (13 replicas keep 12 slots busier than 16).
    if (cfg->replicas_per_wave == 0 && cfg->agent == GRLX_AGENT_AC && rpw == 8 && cfg->wave_limit == 0)
    {
      const int per_simd = (N + simds - 1) / simds;
      if (per_simd >= 15) rpw = 16;
      else if (per_simd >= 9) rpw = 12;
    }
    // TD agents on the acrobot / the compass walker (the environment phase is half of a pass), 15 or more replicas per SIMD: four sub-batches
    // per wave share one environment phase (rollout_wide_kernel<., 3, 4, .>)
    if (cfg->replicas_per_wave == 0 && cfg->agent != GRLX_AGENT_AC && rpw == 8 && cfg->action_steps == 3 &&
        (cfg->env == GRLX_ENV_ACROBOT || cfg->env == GRLX_ENV_COMPASS_WALKER) && (N + simds - 1) / simds >= 15)
      rpw = (cfg->env == GRLX_ENV_COMPASS_WALKER && (N + simds - 1) / simds >= 30) ? 32 : 16;
    if (!has_wide || P.tap_capacity > 0) rpw = 4;
    P.replicas_per_wave = rpw;
    P.wave_limit = cfg->wave_limit > 0 ? cfg->wave_limit : simds;      // these kernels hold a SIMD's whole register file: one wave per SIMD
  }

  const size_t n_tables = (cfg->agent == GRLX_AGENT_AC || cfg->agent == GRLX_AGENT_QV) ? 2 : 1;
  ctx->n_tables = (int)n_tables;
  const size_t table_bytes = ((size_t)N * n_tables * sizeof(Entry)) << logC;
#define CTX_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      int code__ = (e__ == hipErrorOutOfMemory) ? GRLX_ERR_OOM : GRLX_ERR_HIP;               \
      fail(code__, "%s failed: %s", #expr, hipGetErrorString(e__));                          \
      grlx_destroy(ctx);                                                                     \
      return code__;                                                                         \
    }                                                                                        \
  } while (0)
  CTX_TRY(hipMalloc((void **)&ctx->tables, table_bytes));
  CTX_TRY(hipMemset(ctx->tables, 0, table_bytes));
  CTX_TRY(hipMalloc((void **)&ctx->states, sizeof(ReplicaState) * (size_t)N));
  CTX_TRY(hipMalloc((void **)&ctx->row_reward, sizeof(double) * (size_t)N * (size_t)cfg->max_rows));
  CTX_TRY(hipMalloc((void **)&ctx->row_steps, sizeof(int64_t) * (size_t)N * (size_t)cfg->max_rows));
  CTX_TRY(hipMalloc((void **)&ctx->row_trial, sizeof(int64_t) * (size_t)N * (size_t)cfg->max_rows));
  CTX_TRY(hipMemset(ctx->row_reward, 0, sizeof(double) * (size_t)N * (size_t)cfg->max_rows));
  CTX_TRY(hipMemset(ctx->row_steps, 0, sizeof(int64_t) * (size_t)N * (size_t)cfg->max_rows));
  CTX_TRY(hipMemset(ctx->row_trial, 0, sizeof(int64_t) * (size_t)N * (size_t)cfg->max_rows));
  CTX_TRY(hipMalloc((void **)&ctx->row_time, sizeof(double) * (size_t)N * (size_t)cfg->max_rows));
  CTX_TRY(hipMemset(ctx->row_time, 0, sizeof(double) * (size_t)N * (size_t)cfg->max_rows));
  CTX_TRY(hipMalloc((void **)&ctx->scratch, sizeof(uint64_t) * 8));
  CTX_TRY(hipMalloc((void **)&ctx->queue, sizeof(uint32_t)));
  // (the sub-batches beyond the second of a 12- / 16-replica wave park their lane state in registers since round 4: no buffer)
  if (P.replicas_per_wave == 32)
  { // ... five of the eight sub-batches of a 32-replica wave here (grlx_rollout_wide.h: MEMP)
    CTX_TRY(hipMalloc(&ctx->park, kAcParkBytes * 5 * (((size_t)N + 31) / 32)));
    P.park = ctx->park;
  }
  CTX_TRY(hipMalloc((void **)&ctx->max_load, sizeof(uint32_t)));
  CTX_TRY(hipMemset(ctx->max_load, 0, sizeof(uint32_t)));
  CTX_TRY(hipMalloc((void **)&ctx->tap_count, sizeof(uint32_t)));
  CTX_TRY(hipMemset(ctx->tap_count, 0, sizeof(uint32_t)));
  if (cfg->agent == GRLX_AGENT_AC)
  { // the critic's trace survives launches: positions live here, all entries invalid at first
    const size_t words = (size_t)N * 16 * kMaxTrace * 2;
    CTX_TRY(hipMalloc((void **)&ctx->trace_state, words * sizeof(uint32_t)));
    std::vector<uint32_t> init(words, 0u);
    for (size_t i = 0; i < words; i += 2) init[i] = kInvalidPos;
    CTX_TRY(hipMemcpy(ctx->trace_state, init.data(), words * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  if (P.target_interval > 0)
  {
    const size_t bytes = ((size_t)N * sizeof(double)) << logC;
    CTX_TRY(hipMalloc((void **)&ctx->tvals, bytes));
    CTX_TRY(hipMemset(ctx->tvals, 0xFF, bytes));                     // all ones = not materialised
  }
  if (P.tap_capacity > 0)
  {
    CTX_TRY(hipMalloc((void **)&ctx->taps, sizeof(grlx_tap) * (size_t)P.tap_capacity));
    CTX_TRY(hipMemset(ctx->taps, 0, sizeof(grlx_tap) * (size_t)P.tap_capacity));
  }

  // Seed every replica exactly as `grld -s seed` instantiates the yaml
  // (deployer.cpp:70-74; configurable.cpp:627-654 instantiate order):
  //   srand48(seed) -> representation reset: thread-local Rand seeded by global lrand48 #1,
  //   memory*outputs uniforms drawn -> learning sampler Rand (global #2) -> test sampler Rand (global #3).
  std::vector<ReplicaState> hs((size_t)N);
  uint64_t table_draws = (uint64_t)cfg->projector.memory;              // outputs = 1
  if (cfg->agent == GRLX_AGENT_AC || cfg->agent == GRLX_AGENT_QV) table_draws += (uint64_t)cfg->actor_projector.memory;
  if (cfg->target_interval > 0) table_draws += (uint64_t)cfg->projector.memory;      // the target network's own initial draws
  for (int r = 0; r < N; ++r)
  {
    ReplicaState &s = hs[(size_t)r];
    memset(&s, 0, sizeof(s));
    uint64_t G = h_seed((long)seeds[r]);
    G = h_next(G);
    s.TL0 = h_seed((long)(G >> 17));
    s.TL = h_jump(s.TL0, table_draws);
    G = h_next(G);
    s.S1 = h_seed((long)(G >> 17));
    G = h_next(G);
    s.S2 = h_seed((long)(G >> 17));
    s.G = G;
    s.eps_decay = 1;
    s.ac_decay = 1;
    s.tr_len = 0;
    s.tr_total = 1;
    if (cfg->agent == GRLX_AGENT_AC) s.G = h_next(h_seed((long)seeds[r]));   // no samplers: only the thread-local seed is drawn
  }
  CTX_TRY(hipMemcpy(ctx->states, hs.data(), sizeof(ReplicaState) * (size_t)N, hipMemcpyHostToDevice));
#undef CTX_TRY

  P.tables = ctx->tables;
  P.states = ctx->states;
  P.row_reward = ctx->row_reward;
  P.row_time = ctx->row_time;
  P.row_steps = ctx->row_steps;
  P.row_trial = ctx->row_trial;
  P.taps = ctx->taps;
  P.trace_state = ctx->trace_state;
  P.tvals = ctx->tvals;
  P.queue = ctx->queue;
  P.tap_count = ctx->tap_count;
  ctx->P = P;
  *out = ctx;
  return GRLX_OK;
}

int grlx_destroy(grlx_ctx *ctx)
{
  if (!ctx) return GRLX_OK;
  (void)hipFree(ctx->tables);
  (void)hipFree(ctx->states);
  (void)hipFree(ctx->row_reward);
  (void)hipFree(ctx->row_time);
  (void)hipFree(ctx->row_steps);
  (void)hipFree(ctx->row_trial);
  (void)hipFree(ctx->taps);
  (void)hipFree(ctx->tap_count);
  (void)hipFree(ctx->scratch);
  (void)hipFree(ctx->queue);
  if (ctx->env_mail) (void)hipFree(ctx->env_mail);
  if (ctx->park) (void)hipFree(ctx->park);
  if (ctx->agent_rep) (void)hipFree(ctx->agent_rep);
  if (ctx->agent_lane) (void)hipFree(ctx->agent_lane);
  if (ctx->stage_f64) (void)hipFree(ctx->stage_f64);
  if (ctx->stage_i32) (void)hipFree(ctx->stage_i32);
  if (ctx->srv_go) (void)hipEventDestroy(ctx->srv_go);
  if (ctx->srv_done) (void)hipEventDestroy(ctx->srv_done);
  if (ctx->srv_stream) (void)hipStreamDestroy(ctx->srv_stream);
  (void)hipFree(ctx->max_load);
  (void)hipFree(ctx->diag);
  (void)hipFree(ctx->trace_state);
  (void)hipFree(ctx->tvals);
  for (double *timg : ctx->target_images)
    if (timg) (void)hipFree(timg);
  for (double *img : ctx->images)
    if (img) (void)hipFree(img);
  delete ctx;
  return GRLX_OK;
}

int grlx_set_diag(grlx_ctx *ctx, int enable)
{
  if (!ctx) return fail(GRLX_ERR_INVALID, "null ctx");
  const size_t waves = ((size_t)ctx->P.n_replicas + kReplicasPerWave - 1) / kReplicasPerWave;
  if (enable && !ctx->diag)
  {
    HIP_TRY(hipMalloc((void **)&ctx->diag, waves * 8 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(ctx->diag, 0, waves * 8 * sizeof(unsigned long long)));
  }
  ctx->P.diag_out = enable ? ctx->diag : nullptr;
  ctx->P.diag_deferred = enable == 2 ? 1 : 0;     // 1: in-place instantiation (taps possible), 2: the production ordering
  return GRLX_OK;
}

int grlx_read_diag(grlx_ctx *ctx, uint64_t *out, int cap_waves, int *n_waves)
{
  if (!ctx || !out || !n_waves) return fail(GRLX_ERR_INVALID, "bad argument");
  if (!ctx->diag) return fail(GRLX_ERR_INVALID, "diagnostics are not enabled");
  int waves = (ctx->P.n_replicas + kReplicasPerWave - 1) / kReplicasPerWave;
  if (waves > cap_waves) waves = cap_waves;
  DRAIN(ctx);
  HIP_TRY(hipMemcpy(out, ctx->diag, (size_t)waves * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  *n_waves = waves;
  return GRLX_OK;
}

// Re-hash every table of the context into tables of 2^new_logC entries per replica.  Positions are internal to the tables,
// to the persisted actor-critic trace and to the target values (both translated); nothing else refers to them.
static int grow_tables(grlx_ctx *ctx, uint32_t new_logC)
{
  if (new_logC <= ctx->P.logC) return GRLX_OK;
  if (new_logC > 26) return fail(GRLX_ERR_INVALID, "tables cannot grow beyond 2^26 entries");
  HIP_TRY(hipDeviceSynchronize());
  const size_t N = (size_t)ctx->P.n_replicas, n_tables = (size_t)ctx->n_tables;
  const size_t new_bytes = (N * n_tables * sizeof(Entry)) << new_logC;
  DevBuf nt, remap, ntv;                     // freed on every early return; ownership moves to the context at the end
  if (nt.alloc(new_bytes) != hipSuccess ||
      ((ctx->trace_state || ctx->tvals) && remap.alloc((N * sizeof(uint32_t)) << ctx->P.logC) != hipSuccess) ||
      (ctx->tvals && ntv.alloc((N * sizeof(double)) << new_logC) != hipSuccess))
  {
    (void)hipGetLastError();
    return fail(GRLX_ERR_OOM, "no device memory to grow the sparse tables to 2^%u entries per replica (%.1f GiB)", new_logC,
                (double)new_bytes / 1073741824.);
  }
  HIP_TRY(hipMemset(nt.p, 0, new_bytes));
  if (ntv.p) HIP_TRY(hipMemset(ntv.p, 0xFF, (N * sizeof(double)) << new_logC));
  HIP_TRY(launch_rehash(ctx->P, ctx->n_tables, (Entry *)nt.p, new_logC, (uint32_t *)remap.p, nullptr));
  if (remap.p) HIP_TRY(launch_remap_positions(ctx->P, (const uint32_t *)remap.p, new_logC, (double *)ntv.p, nullptr));
  HIP_TRY(hipDeviceSynchronize());
  (void)hipFree(ctx->tables);
  if (ctx->tvals) (void)hipFree(ctx->tvals);
  ctx->tables = (Entry *)nt.p;
  ctx->tvals = (double *)ntv.p;
  nt.p = nullptr;
  ntv.p = nullptr;
  ctx->P.tables = ctx->tables;
  ctx->P.tvals = ctx->tvals;
  ctx->P.logC = new_logC;
  return GRLX_OK;
}

int grlx_table_capacity(grlx_ctx *ctx, uint32_t *log2_entries)
{
  if (!ctx || !log2_entries) return fail(GRLX_ERR_INVALID, "bad argument");
  *log2_entries = ctx->P.logC;
  return GRLX_OK;
}

int grlx_grow_tables(grlx_ctx *ctx, uint32_t new_log2)
{
  if (!ctx) return fail(GRLX_ERR_INVALID, "null ctx");
  DRAIN(ctx);
  return grow_tables(ctx, new_log2);
}

// Experiment::reset() between two runs of one process (online_learning.cpp:307-308 -> {action: reset} over the experiment's subtree):
// every parameter is drawn again from the CONTINUING thread-local stream (linear.cpp:104-125; with lazy initialisation: the stream
// position becomes the new base of the draws and moves on by the tables' sizes, the sparse tables empty), the traces are cleared
// (sarsa.cpp:60-66, td.cpp:64), the exploration decay of sampler/epsilon_greedy and mapping/policy/action goes back to 1
// (greedy.cpp:140-141, action.cpp:93-97), the run's counters (ss, tt) and rows start again.  Nothing is reseeded: the global stream,
// the samplers' private streams and the environment state continue -- which is why run 1 of `runs: 2` differs from a fresh process.
int grlx_reset_run(grlx_ctx *ctx)
{
  if (!ctx) return fail(GRLX_ERR_INVALID, "null ctx");
  DRAIN(ctx);
  HIP_TRY(hipDeviceSynchronize());
  const size_t N = (size_t)ctx->P.n_replicas;
  std::vector<ReplicaState> hs(N);
  HIP_TRY(hipMemcpy(hs.data(), ctx->states, sizeof(ReplicaState) * N, hipMemcpyDeviceToHost));
  uint64_t table_draws = (uint64_t)ctx->cfg.projector.memory;
  if (ctx->cfg.agent == GRLX_AGENT_AC || ctx->cfg.agent == GRLX_AGENT_QV) table_draws += (uint64_t)ctx->cfg.actor_projector.memory;
  // a target network is reached by the walk too (a provided object is a child configurator, visited before the representation itself:
  // configurable.cpp:690-712, 754-757): it draws again first, the representation next, synchronize() blends the two fresh vectors -- the
  // sequence of construction, from the continuing stream; ParameterizedRepresentation::count_ = 0
  if (ctx->cfg.target_interval > 0) table_draws += (uint64_t)ctx->cfg.projector.memory;
  for (ReplicaState &s : hs)
  {
    s.sync_count = 0;
    s.syncs = 0;
    s.TL0 = s.TL;                         // the re-draw starts where the stream stands
    s.TL = h_jump(s.TL, table_draws);
    s.eps_decay = 1;
    s.ac_decay = 1;
    s.tr_len = 0;
    s.tr_total = 1;
    s.tt = 0;
    s.ss = 0;
    s.test_steps = 0;                     // (grlx_step_counts counts per run)
    s.rows = 0;
    s.n_slots[0] = s.n_slots[1] = 0;
    s.lazy_base[0] = s.lazy_base[1] = nullptr;      // a loaded policy is overwritten by the re-draw like every other parameter
    s.target_base = nullptr;
    s.syncs_base = 0;
  }
  HIP_TRY(hipMemcpy(ctx->states, hs.data(), sizeof(ReplicaState) * N, hipMemcpyHostToDevice));
  // (emptying the tables drops the claims of projector/tile_coding:safe with them: tile_coding.cpp:82-89)
  HIP_TRY(hipMemset(ctx->tables, 0, (N * (size_t)ctx->n_tables * sizeof(Entry)) << ctx->P.logC));
  if (ctx->tvals) HIP_TRY(hipMemset(ctx->tvals, 0xFF, (N * sizeof(double)) << ctx->P.logC));      // kTvalUnset: not materialised
  HIP_TRY(hipMemset(ctx->max_load, 0, sizeof(uint32_t)));
  if (ctx->trace_state)
  {
    const size_t words = N * 16 * kMaxTrace * 2;
    std::vector<uint32_t> init(words, 0u);
    for (size_t i = 0; i < words; i += 2) init[i] = kInvalidPos;
    HIP_TRY(hipMemcpy(ctx->trace_state, init.data(), words * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  if (ctx->tap_count) HIP_TRY(hipMemset(ctx->tap_count, 0, sizeof(uint32_t)));
  if (ctx->agent_rep) HIP_TRY(hipMemset(ctx->agent_rep, 0, sizeof(AgentRep) * N));
  if (ctx->agent_lane) HIP_TRY(hipMemset(ctx->agent_lane, 0, sizeof(uint32_t) * N * 16 * 2));
  for (double *&timg : ctx->target_images)
    if (timg) { (void)hipFree(timg); timg = nullptr; }
  ctx->trials_run = 0;
  return GRLX_OK;
}

// Mailboxes, stream and events of the environment server, created at the first launch that wants them.  The server is an optimisation: when
// its 1 KB per replica (or a stream, or an event) cannot be had, the context goes on without it.
static bool env_server_ready(grlx_ctx *ctx)
{
  if (ctx->env_mail) return true;
  ctx->env_mail_bytes = env_server_mail_bytes(ctx->P);
  bool ok = ctx->env_mail_bytes != 0 && hipMalloc((void **)&ctx->env_mail, (size_t)ctx->P.n_replicas * ctx->env_mail_bytes) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&ctx->srv_stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&ctx->srv_go, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&ctx->srv_done, hipEventDisableTiming) == hipSuccess;
  if (ok) return true;
  (void)hipGetLastError();
  if (ctx->env_mail) (void)hipFree(ctx->env_mail);
  if (ctx->srv_go) (void)hipEventDestroy(ctx->srv_go);
  if (ctx->srv_done) (void)hipEventDestroy(ctx->srv_done);
  if (ctx->srv_stream) (void)hipStreamDestroy(ctx->srv_stream);
  ctx->env_mail = nullptr; ctx->srv_go = ctx->srv_done = nullptr; ctx->srv_stream = nullptr;
  ctx->env_server = 0;
  return false;
}

static int run_trials(grlx_ctx *ctx, int n_trials, uint64_t steps_budget, void *stream)
{
  if (!ctx || n_trials < 0) return fail(GRLX_ERR_INVALID, "bad argument");
  if (ctx->cfg.env == GRLX_ENV_EXTERNAL) return fail(GRLX_ERR_INVALID, "this context has no environment (GRLX_ENV_EXTERNAL): drive it through grlx_agent_start / _step / _end");
  if (n_trials == 0) return GRLX_OK;
  if (steps_budget != 0 && (ctx->P.tap_capacity > 0 || ctx->P.diag_out))
    return fail(GRLX_ERR_INVALID, "a steps budget is not available with taps or diagnostics");
  if (!ctx->run_pending && ctx->P.logC < ctx->logC_max)
  { // nothing in flight: how full did the last run leave the fullest table?  Beyond a quarter, grow to an eighth.
    uint32_t used = 0;
    HIP_TRY(hipMemcpy(&used, ctx->max_load, sizeof(used), hipMemcpyDeviceToHost));
    if ((uint64_t)used * 4u > (1ull << ctx->P.logC))
    {
      uint32_t want = ctx->P.logC;
      while ((uint64_t)used * 8u > (1ull << want) && want < ctx->logC_max) ++want;
      const int rc = grow_tables(ctx, want);
      if (rc == GRLX_ERR_OOM)
      { // growing needs the old and the new tables at once: when that does not fit, the run goes on in the tables it has
        // (a quarter full is not an overflow; a real one is still reported as GRLX_ERR_TABLE_FULL) and does not try again
        g_err.clear();
        ctx->logC_max = ctx->P.logC;
      }
      else if (rc != GRLX_OK) return rc;
    }
  }
  // a run still pending on ANOTHER stream is waited for first: only one stream is remembered, and the launches below
  // continue from the replica state that run leaves behind
  if (ctx->run_pending && ctx->run_stream != (hipStream_t)stream) DRAIN(ctx);
  // One launch per <= kTrialsPerLaunch trials: replica state (and the actor-critic trace) persists in
  // HBM between launches, so results do not depend on the chunking (tested), and no single kernel
  // runs for minutes (compass walker: up to 1000 steps per episode).
  // A launch also never holds more than ~kStepsPerLaunch control steps of one replica (long episodes: fewer trials).
  const double kStepsPerLaunch = 65536;
  const double horizon = (ctx->cfg.env == GRLX_ENV_COMPASS_WALKER ? 2 : 1) * ctx->cfg.timeout / ctx->cfg.control_step + 1;
  int kTrialsPerLaunch = 32;
  if (horizon * kTrialsPerLaunch > kStepsPerLaunch) kTrialsPerLaunch = (int)fmax(1., floor(kStepsPerLaunch / horizon));
  // with a steps budget the budget bounds what one replica does in a launch (its remaining steps plus one episode): all trials go into
  // ONE launch, so that a replica which has reached its budget does not wait at 32-trial boundaries for the ones that have not
  DevParams Pb = ctx->P;
  Pb.steps_budget = steps_budget;
  if (steps_budget != 0) kTrialsPerLaunch = n_trials;
  ctx->run_stream = (hipStream_t)stream;
  ctx->run_pending = true;
  for (int done = 0; done < n_trials; done += kTrialsPerLaunch)
  {
    const int n = (n_trials - done < kTrialsPerLaunch) ? n_trials - done : kTrialsPerLaunch;
    if (done > 0 && ctx->P.logC < ctx->logC_max)
    { // a run of several launches: the tables may grow BETWEEN its launches too (a launch cannot grow them).  One 4-byte read-back per
      // further launch, which waits for the launch before it: a call of more than kTrialsPerLaunch trials is no longer fully asynchronous
      // while the tables may still grow (a deployer runs a whole run as ONE grlx_run; cart-pole actor-critic tables start at 2^16 entries
      // and 32 trials create about 25 000 slots per table)
      uint32_t used = 0;
      HIP_TRY(launch_max_load(ctx->P, ctx->n_tables, ctx->max_load, (hipStream_t)stream));
      HIP_TRY(hipMemcpyAsync(&used, ctx->max_load, sizeof(used), hipMemcpyDeviceToHost, (hipStream_t)stream));
      HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
      if ((uint64_t)used * 4u > (1ull << ctx->P.logC))
      {
        uint32_t want = ctx->P.logC;
        while ((uint64_t)used * 8u > (1ull << want) && want < ctx->logC_max) ++want;
        const int rc = grow_tables(ctx, want);
        if (rc == GRLX_ERR_OOM) { g_err.clear(); ctx->logC_max = ctx->P.logC; }
        else if (rc != GRLX_OK) return rc;
        Pb.tables = ctx->P.tables;                 // the launches below work on the new tables
        Pb.tvals = ctx->P.tvals;
        Pb.logC = ctx->P.logC;
      }
    }
    if (ctx->poison) HIP_TRY(launch_poison_registers(ctx->poison_pattern, (hipStream_t)stream));
    if (ctx->cfg.agent == GRLX_AGENT_AC)
      HIP_TRY(launch_rollout_ac(Pb, n, (hipStream_t)stream, &ctx->last_kernel));
    else if (ctx->cfg.agent == GRLX_AGENT_QV)
      HIP_TRY(launch_rollout_qv(Pb, n, (hipStream_t)stream, &ctx->last_kernel));
    else if (ctx->cfg.target_interval > 0 || ctx->cfg.projector.safe != 0)
      HIP_TRY(launch_rollout_tgt(Pb, n, (hipStream_t)stream, &ctx->last_kernel));
    else if (ctx->cfg.trace == GRLX_TRACE_ACCUMULATING)
      HIP_TRY(launch_rollout_acc(Pb, n, (hipStream_t)stream, &ctx->last_kernel));
    else if (ctx->env_server && env_server_serves(Pb) && (size_t)ctx->P.n_replicas * env_server_mail_bytes(Pb) < (1ull << 31) && env_server_ready(ctx))
    { // the server's launch forks off the caller's stream and joins it again: for the caller, still one stream-ordered operation
      Pb.env_mail = ctx->env_mail;
      Pb.env_tune = 3u;        // the rollout wave is the critical path: it issues first, the server fills its gaps (+0.7 %)
      if (const char *tune = getenv("GRLX_ENV_SERVER_TUNE")) Pb.env_tune = (uint32_t)strtoul(tune, nullptr, 0);
      HIP_TRY(hipMemsetAsync(ctx->env_mail, 0, (size_t)ctx->P.n_replicas * ctx->env_mail_bytes, (hipStream_t)stream));
      HIP_TRY(hipEventRecord(ctx->srv_go, (hipStream_t)stream));
      HIP_TRY(hipStreamWaitEvent(ctx->srv_stream, ctx->srv_go, 0));
      HIP_TRY(launch_env_server(Pb, ctx->srv_stream));
      HIP_TRY(launch_rollout(Pb, n, (hipStream_t)stream, &ctx->last_kernel));
      HIP_TRY(hipEventRecord(ctx->srv_done, ctx->srv_stream));
      HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, ctx->srv_done, 0));
    }
    else
      HIP_TRY(launch_rollout(Pb, n, (hipStream_t)stream, &ctx->last_kernel));
  }
  HIP_TRY(launch_max_load(ctx->P, ctx->n_tables, ctx->max_load, (hipStream_t)stream));
  ctx->trials_run += n_trials;
  return GRLX_OK;
}

// diagnostic (not part of include/grlx.h): the mailboxes of the environment server after the last launch (GRLX_ENV_SERVER_STATS builds
// leave cycle counts in EnvMail::stats)
int grlx_env_server_debug(grlx_ctx *ctx, void *out, size_t bytes)
{
  if (!ctx || !out) return fail(GRLX_ERR_INVALID, "bad argument");
  if (!ctx->env_mail) return fail(GRLX_ERR_INVALID, "the environment server has not run in this context");
  DRAIN(ctx);
  const size_t have = (size_t)ctx->P.n_replicas * ctx->env_mail_bytes;
  HIP_TRY(hipMemcpy(out, ctx->env_mail, bytes < have ? bytes : have, hipMemcpyDeviceToHost));
  return GRLX_OK;
}

// diagnostic (not part of include/grlx.h): in the last launch that had the environment server, how many replicas took every step from it
// and how many gave up waiting and integrated themselves; both 0 when no launch of this context had it
int grlx_env_server_counts(grlx_ctx *ctx, int *served, int *fell_back)
{
  if (!ctx || !served || !fell_back) return fail(GRLX_ERR_INVALID, "bad argument");
  *served = *fell_back = 0;
  if (!ctx->env_mail) return GRLX_OK;
  DRAIN(ctx);
  std::vector<unsigned long long> flag((size_t)ctx->P.n_replicas);
  HIP_TRY(hipMemcpy2D(flag.data(), sizeof(unsigned long long), (const char *)ctx->env_mail + kEnvMailFlagOffset, ctx->env_mail_bytes,
                      sizeof(unsigned long long), (size_t)ctx->P.n_replicas, hipMemcpyDeviceToHost));
  for (unsigned long long f : flag)
  {
    if (f == 1u) ++*served;
    if (f == 2u) ++*fell_back;
  }
  return GRLX_OK;
}

int grlx_run(grlx_ctx *ctx, int n_trials, void *stream) { return run_trials(ctx, n_trials, 0, stream); }

// The trial loop of OnlineLearningExperiment::run with both of its bounds (online_learning.cpp:154):
//   for (ss = 0, tt = 0; (!trials_ || tt < trials_) && (!steps_ || ss < steps_); ++tt)
// every replica runs at most `max_trials` further trials and starts none once its learning steps of the run (since grlx_create or
// grlx_reset_run) have reached `steps`.  Replicas stop at trials of their own: rows are ragged (grlx_replica_rows, grlx_curve_stats
// counts per row).  One launch: `steps` bounds the work of a replica.
int grlx_run_steps(grlx_ctx *ctx, int max_trials, uint64_t steps, void *stream)
{
  if (steps == 0) return fail(GRLX_ERR_INVALID, "grlx_run_steps: steps must be > 0 (grlx_run has no budget)");
  if (steps > (1ull << 40)) return fail(GRLX_ERR_INVALID, "grlx_run_steps: steps out of range");
  return run_trials(ctx, max_trials, steps, stream);
}

static int status_to_error(uint64_t st)
{
  if (st & ST_TABLE_FULL) return fail(GRLX_ERR_TABLE_FULL, "a replica's sparse weight table overflowed: raise table_log2_capacity");
  if (st & ST_TRACE_OVERFLOW) return fail(GRLX_ERR_INVALID, "register trace overflow");
  if (st & ST_BAD_POS) return fail(GRLX_ERR_INVALID, "internal error: a TD update was queued without a table position");
  if (st & ST_DOMAIN) return fail(GRLX_ERR_DOMAIN, "sin/cos argument outside |x| < 2^20");
  if (st & ST_ROWS_FULL) return fail(GRLX_ERR_ROWS_FULL, "more test rows than max_rows");
  return GRLX_OK;
}

int grlx_sync(grlx_ctx *ctx, void *stream)
{
  if (!ctx) return fail(GRLX_ERR_INVALID, "null ctx");
  uint64_t h[3];
  if ((hipStream_t)stream != ctx->run_stream) DRAIN(ctx);
  HIP_TRY(launch_step_counts(ctx->P, ctx->scratch, (hipStream_t)stream));
  HIP_TRY(hipMemcpyAsync(h, ctx->scratch, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  if ((hipStream_t)stream == ctx->run_stream) ctx->run_pending = false;
  return status_to_error(h[2]);
}

int grlx_step_counts(grlx_ctx *ctx, uint64_t *learn_steps, uint64_t *test_steps)
{
  if (!ctx) return fail(GRLX_ERR_INVALID, "null ctx");
  uint64_t h[3];
  DRAIN(ctx);
  HIP_TRY(launch_step_counts(ctx->P, ctx->scratch, nullptr));
  HIP_TRY(hipMemcpy(h, ctx->scratch, sizeof(h), hipMemcpyDeviceToHost));
  if (learn_steps) *learn_steps = h[0];
  if (test_steps) *test_steps = h[1];
  return GRLX_OK;
}

int grlx_rows(grlx_ctx *ctx)
{
  if (!ctx) return fail(GRLX_ERR_INVALID, "null ctx");
  ReplicaState s;
  DRAIN(ctx);
  if (hipMemcpy(&s, ctx->states, sizeof(s), hipMemcpyDeviceToHost) != hipSuccess) return fail(GRLX_ERR_HIP, "hipMemcpy failed");
  return (int)s.rows;
}

int grlx_replica_rows(grlx_ctx *ctx, int replica)
{
  if (!ctx || replica < 0 || replica >= ctx->P.n_replicas) return fail(GRLX_ERR_INVALID, "bad argument");
  ReplicaState s;
  DRAIN(ctx);
  if (hipMemcpy(&s, ctx->states + replica, sizeof(s), hipMemcpyDeviceToHost) != hipSuccess) return fail(GRLX_ERR_HIP, "hipMemcpy failed");
  return (int)s.rows;
}

int grlx_read_rows(grlx_ctx *ctx, int replica, int first, int count, int64_t *trial, int64_t *steps, double *reward)
{
  if (!ctx || replica < 0 || replica >= ctx->P.n_replicas || first < 0 || count < 0 || first + count > ctx->P.max_rows)
    return fail(GRLX_ERR_INVALID, "bad argument");
  const size_t N = (size_t)ctx->P.n_replicas;
  if (count == 0) return GRLX_OK;
  DRAIN(ctx);
  // rows are stored [row][replica]: one strided copy per column
  const size_t at = (size_t)first * N + (size_t)replica;
  if (reward) HIP_TRY(hipMemcpy2D(reward, sizeof(double), ctx->row_reward + at, N * sizeof(double), sizeof(double), (size_t)count, hipMemcpyDeviceToHost));
  if (steps) HIP_TRY(hipMemcpy2D(steps, sizeof(int64_t), ctx->row_steps + at, N * sizeof(int64_t), sizeof(int64_t), (size_t)count, hipMemcpyDeviceToHost));
  if (trial) HIP_TRY(hipMemcpy2D(trial, sizeof(int64_t), ctx->row_trial + at, N * sizeof(int64_t), sizeof(int64_t), (size_t)count, hipMemcpyDeviceToHost));
  return GRLX_OK;
}

int grlx_last_kernel(grlx_ctx *ctx) { return ctx ? ctx->last_kernel : GRLX_KERNEL_NONE; }
int grlx_replicas_per_wave(grlx_ctx *ctx) { return ctx ? ctx->P.replicas_per_wave : 0; }

int grlx_read_row_times(grlx_ctx *ctx, int replica, int first, int count, double *episode_time)
{
  if (!ctx || !episode_time || replica < 0 || replica >= ctx->P.n_replicas || first < 0 || count < 0 || first + count > ctx->P.max_rows)
    return fail(GRLX_ERR_INVALID, "bad argument");
  const size_t N = (size_t)ctx->P.n_replicas;
  if (count == 0) return GRLX_OK;
  DRAIN(ctx);
  HIP_TRY(hipMemcpy2D(episode_time, sizeof(double), ctx->row_time + (size_t)first * N + (size_t)replica, N * sizeof(double), sizeof(double),
                      (size_t)count, hipMemcpyDeviceToHost));
  return GRLX_OK;
}

int grlx_curve_stats(grlx_ctx *ctx, int first, int count, double *out_dev, void *stream)
{
  if (!ctx || !out_dev || first < 0 || count < 0 || first + count > ctx->P.max_rows) return fail(GRLX_ERR_INVALID, "bad argument");
  if ((hipStream_t)stream != ctx->run_stream) DRAIN(ctx);
  HIP_TRY(launch_curve_stats(ctx->P, first, count, out_dev, (hipStream_t)stream));
  return GRLX_OK;
}

int grlx_get_env_state(grlx_ctx *ctx, int replica, double *state)
{
  if (!ctx || !state || replica < 0 || replica >= ctx->P.n_replicas) return fail(GRLX_ERR_INVALID, "bad argument");
  ReplicaState s;
  DRAIN(ctx);
  HIP_TRY(hipMemcpy(&s, ctx->states + replica, sizeof(s), hipMemcpyDeviceToHost));
  memcpy(state, s.x, sizeof(double) * GRLX_MAX_STATE);
  return GRLX_OK;
}

int grlx_get_rng(grlx_ctx *ctx, int replica, uint64_t out[4])
{
  if (!ctx || !out || replica < 0 || replica >= ctx->P.n_replicas) return fail(GRLX_ERR_INVALID, "bad argument");
  ReplicaState s;
  DRAIN(ctx);
  HIP_TRY(hipMemcpy(&s, ctx->states + replica, sizeof(s), hipMemcpyDeviceToHost));
  out[0] = s.G; out[1] = s.TL; out[2] = s.S1; out[3] = s.S2;
  return GRLX_OK;
}

int grlx_table_load(grlx_ctx *ctx, int table, int replica, uint32_t *n_slots_used)
{
  if (!ctx || !n_slots_used || table < 0 || table >= ctx->n_tables || replica < 0 || replica >= ctx->P.n_replicas) return fail(GRLX_ERR_INVALID, "bad argument");
  ReplicaState s;
  DRAIN(ctx);
  HIP_TRY(hipMemcpy(&s, ctx->states + replica, sizeof(s), hipMemcpyDeviceToHost));
  *n_slots_used = s.n_slots[table];
  return GRLX_OK;
}

int grlx_get_target_weights(grlx_ctx *ctx, int replica, const uint32_t *slots, int n, double *out, uint32_t *n_syncs)
{
  if (!ctx || !slots || !out || n < 0 || replica < 0 || replica >= ctx->P.n_replicas) return fail(GRLX_ERR_INVALID, "bad argument");
  if (ctx->cfg.target_interval <= 0) return fail(GRLX_ERR_INVALID, "this context has no target network (target_interval = 0)");
  DRAIN(ctx);
  if (n_syncs)
  {
    ReplicaState s;
    HIP_TRY(hipMemcpy(&s, ctx->states + replica, sizeof(s), hipMemcpyDeviceToHost));
    *n_syncs = s.syncs;
  }
  if (n == 0) return GRLX_OK;
  DevBuf ds, dout;
  HIP_TRY(ds.alloc(sizeof(uint32_t) * (size_t)n));
  HIP_TRY(dout.alloc(sizeof(double) * (size_t)n));
  HIP_TRY(hipMemcpy(ds.p, slots, sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice));
  HIP_TRY(launch_get_target_weights(ctx->P, replica, ds.as<uint32_t>(), n, dout.as<double>(), nullptr));
  HIP_TRY(hipMemcpy(out, dout.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  return GRLX_OK;
}

int grlx_read_taps(grlx_ctx *ctx, grlx_tap *out, int cap, int *n)
{
  if (!ctx || !out || !n) return fail(GRLX_ERR_INVALID, "bad argument");
  uint32_t cnt = 0;
  DRAIN(ctx);
  HIP_TRY(hipMemcpy(&cnt, ctx->tap_count, sizeof(cnt), hipMemcpyDeviceToHost));
  int m = (int)cnt < cap ? (int)cnt : cap;
  if (m > ctx->P.tap_capacity) m = ctx->P.tap_capacity;
  if (m > 0) HIP_TRY(hipMemcpy(out, ctx->taps, sizeof(grlx_tap) * (size_t)m, hipMemcpyDeviceToHost));
  *n = m;
  return GRLX_OK;
}

int grlx_get_weights(grlx_ctx *ctx, int table, int replica, const uint32_t *slots, int n, double *out)
{
  if (!ctx || !slots || !out || n < 0 || table < 0 || table >= ctx->n_tables || replica < 0 || replica >= ctx->P.n_replicas) return fail(GRLX_ERR_INVALID, "bad argument");
  if (n == 0) return GRLX_OK;
  DRAIN(ctx);
  DevBuf ds, dout;
  HIP_TRY(ds.alloc(sizeof(uint32_t) * (size_t)n));
  HIP_TRY(dout.alloc(sizeof(double) * (size_t)n));
  HIP_TRY(hipMemcpy(ds.p, slots, sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice));
  HIP_TRY(launch_get_weights(ctx->P, table, replica, ds.as<uint32_t>(), n, dout.as<double>(), nullptr));
  HIP_TRY(hipMemcpy(out, dout.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  return GRLX_OK;
}

int grlx_export_weights(grlx_ctx *ctx, int table, int replica, double *out)
{
  if (!ctx || !out || table < 0 || table >= ctx->n_tables || replica < 0 || replica >= ctx->P.n_replicas) return fail(GRLX_ERR_INVALID, "bad argument");
  const size_t memory = (size_t)(table == 1 ? ctx->P.tile_actor.memory : ctx->P.tile.memory);
  DRAIN(ctx);
  DevBuf dout;
  HIP_TRY(dout.alloc(sizeof(double) * memory));
  HIP_TRY(launch_export_weights(ctx->P, table, replica, dout.as<double>(), nullptr));
  HIP_TRY(hipMemcpy(out, dout.p, sizeof(double) * memory, hipMemcpyDeviceToHost));
  return GRLX_OK;
}

int grlx_load_weights(grlx_ctx *ctx, int table, int first_replica, int n_replicas, const double *dense, uint64_t count)
{
  if (!ctx || !dense || table < 0 || table >= ctx->n_tables) return fail(GRLX_ERR_INVALID, "bad argument");
  const int N = ctx->P.n_replicas;
  if (first_replica < 0 || n_replicas < 0 || first_replica > N || n_replicas > N - first_replica)
    return fail(GRLX_ERR_INVALID, "replica range [%d, %d) outside [0, %d)", first_replica, first_replica + n_replicas, N);
  const size_t memory = (size_t)(table == 1 ? ctx->P.tile_actor.memory : ctx->P.tile.memory);
  if (count != (uint64_t)memory)                       // representation.h:247-252 "Configuration mismatch"
    return fail(GRLX_ERR_INVALID, "configuration mismatch: %llu weights given, the table has %zu", (unsigned long long)count, memory);
  if (ctx->cfg.agent == GRLX_AGENT_AC && ctx->trials_run != 0)
    return fail(GRLX_ERR_INVALID, "actor-critic: load before the first run (the critic's trace refers to table positions)");
  // a target network (table 0 of a SARSA / Q context): setParams is followed by synchronize() (representation.h:256-257) -- target <- tau * image
  // + (1 - tau) * target over the WHOLE parameter vector.  With tau = 0 the target is the image; otherwise every replica needs its own dense
  // vector of the result (its old target differs), `memory` doubles each, held until the next load or reset.
  const bool with_target = ctx->cfg.target_interval > 0 && table == 0;
  const bool target_images = with_target && ctx->cfg.target_tau != 0.;
  if (target_images && (double)n_replicas * (double)memory * sizeof(double) > 8.0 * 1024 * 1024 * 1024)
    return fail(GRLX_ERR_OOM, "loading into a target network with tau != 0 keeps one dense target vector per replica (%zu doubles each): %d replicas exceed the 8 GiB this entry point allows",
                memory, n_replicas);
  if (n_replicas == 0) return GRLX_OK;
  HIP_TRY(hipDeviceSynchronize());
  ctx->run_pending = false;
  double *img = nullptr;
  if (hipMalloc((void **)&img, sizeof(double) * memory) != hipSuccess) return fail(GRLX_ERR_OOM, "no device memory for a %zu-weight policy image", memory);
  // (a target network with tau != 0: one new dense vector per replica, allocated before anything is touched -- a failure leaves the context as it was)
  std::vector<double *> fresh((size_t)n_replicas, nullptr);
  if (target_images)
    for (int k = 0; k < n_replicas; ++k)
      if (hipMalloc((void **)&fresh[(size_t)k], sizeof(double) * memory) != hipSuccess)
      {
        (void)hipGetLastError();
        for (double *t : fresh)
          if (t) (void)hipFree(t);
        (void)hipFree(img);
        return fail(GRLX_ERR_OOM, "no device memory for the target vector of replica %d", first_replica + k);
      }
  ctx->images.push_back(img);
  {
    std::vector<int> &of = ctx->image_of[table];
    if (of.empty()) of.assign((size_t)N, -1);
    const int mine = (int)ctx->images.size() - 1;
    for (int r = first_replica; r < first_replica + n_replicas; ++r) of[(size_t)r] = mine;
  }
  HIP_TRY(hipMemcpy(img, dense, sizeof(double) * memory, hipMemcpyHostToDevice));
  if (with_target)
  { // the target's new values, from its values NOW (entries and lazily defined ones alike), BEFORE the tables are emptied
    if (ctx->target_images.empty()) ctx->target_images.assign((size_t)N, nullptr);
    std::vector<ReplicaState> hs((size_t)n_replicas);
    HIP_TRY(hipMemcpy(hs.data(), ctx->states + first_replica, sizeof(ReplicaState) * (size_t)n_replicas, hipMemcpyDeviceToHost));
    if (target_images)
      for (int k = 0; k < n_replicas; ++k) HIP_TRY(launch_target_after_load(ctx->P, first_replica + k, img, fresh[(size_t)k], nullptr));
    HIP_TRY(hipDeviceSynchronize());          // (the kernels read the vectors of an earlier load for the last time)
    for (int k = 0; k < n_replicas; ++k)
    {
      const int r = first_replica + k;
      if (ctx->target_images[(size_t)r]) (void)hipFree(ctx->target_images[(size_t)r]);
      ctx->target_images[(size_t)r] = fresh[(size_t)k];
      hs[(size_t)k].target_base = fresh[(size_t)k];
      hs[(size_t)k].syncs += 1;                     // the load's synchronize() (count_ = 0, representation.h:284-296)
      hs[(size_t)k].syncs_base = hs[(size_t)k].syncs;
      hs[(size_t)k].sync_count = 0;
    }
    HIP_TRY(hipMemcpy(ctx->states + first_replica, hs.data(), sizeof(ReplicaState) * (size_t)n_replicas, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(ctx->tvals + ((size_t)first_replica << ctx->P.logC), 0xFF, ((size_t)n_replicas * sizeof(double)) << ctx->P.logC));      // kTvalUnset
  }
  // setParams() overwrites every weight: forget the sparse tables of these replicas; every slot is
  // re-created on first touch from the image
  const size_t per_replica = sizeof(Entry) << ctx->P.logC;
  Entry *base = ctx->tables + (((size_t)table * (size_t)N + (size_t)first_replica) << ctx->P.logC);
  if (ctx->P.twin_tables)      // twin tables keep their common key layout: the entries stay and take their values from the image
    HIP_TRY(launch_reload_entries(ctx->P, table, first_replica, n_replicas, img, nullptr));
  else
    HIP_TRY(hipMemset(base, 0, per_replica * (size_t)n_replicas));
  HIP_TRY(launch_set_lazy_base(ctx->P, table, first_replica, n_replicas, img, nullptr));
  HIP_TRY(hipDeviceSynchronize());
  // an earlier image that no replica of any table refers to any more goes now (repeated loads do not accumulate)
  for (size_t k = 0; k + 1 < ctx->images.size(); ++k)
  {
    if (!ctx->images[k]) continue;
    bool used = false;
    for (int t = 0; t < 2 && !used; ++t)
      for (int v : ctx->image_of[t])
        if (v == (int)k) { used = true; break; }
    if (!used)
    {
      (void)hipFree(ctx->images[k]);
      ctx->images[k] = nullptr;
    }
  }
  return GRLX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The per-step plug-in interfaces on a context's replicas (grlx_step.h).  Host pointers; each call copies its arguments in,
// launches one kernel over all replicas and copies the results out.
static int step_buffers(grlx_ctx *ctx)
{
  if (ctx->stage_f64) return GRLX_OK;
  const size_t N = (size_t)ctx->P.n_replicas;
  if (hipMalloc((void **)&ctx->stage_f64, sizeof(double) * N * (GRLX_MAX_DIMS + 2)) != hipSuccess ||
      hipMalloc((void **)&ctx->stage_i32, sizeof(int32_t) * N * 2) != hipSuccess)
  {
    (void)hipGetLastError();
    if (ctx->stage_f64) (void)hipFree(ctx->stage_f64);
    ctx->stage_f64 = nullptr;
    return fail(GRLX_ERR_OOM, "no device memory for the staging buffers of the per-step entry points");
  }
  return GRLX_OK;
}

static const int32_t *stage_active(grlx_ctx *ctx, const int32_t *active, hipError_t *err)
{
  *err = hipSuccess;
  if (!active) return nullptr;
  *err = hipMemcpy(ctx->stage_i32, active, sizeof(int32_t) * (size_t)ctx->P.n_replicas, hipMemcpyHostToDevice);
  return ctx->stage_i32;
}

int grlx_env_start(grlx_ctx *ctx, int test, const int32_t *active, double *obs)
{
  if (!ctx || !obs) return fail(GRLX_ERR_INVALID, "bad argument");
  if (ctx->cfg.env == GRLX_ENV_EXTERNAL) return fail(GRLX_ERR_INVALID, "this context has no environment (GRLX_ENV_EXTERNAL)");
  DRAIN(ctx);
  int rc = step_buffers(ctx);
  if (rc != GRLX_OK) return rc;
  const size_t N = (size_t)ctx->P.n_replicas, D = (size_t)ctx->D;
  hipError_t e;
  const int32_t *act = stage_active(ctx, active, &e);
  HIP_TRY(e);
  if (active) HIP_TRY(hipMemcpy(ctx->stage_f64, obs, sizeof(double) * N * D, hipMemcpyHostToDevice));     // inactive rows keep the caller's values
  HIP_TRY(launch_env_start(ctx->P, test ? 1 : 0, act, ctx->stage_f64, nullptr));
  HIP_TRY(hipMemcpy(obs, ctx->stage_f64, sizeof(double) * N * D, hipMemcpyDeviceToHost));
  return GRLX_OK;
}

int grlx_env_advance(grlx_ctx *ctx, const int32_t *active, const double *action, double *obs, double *reward, int32_t *terminal)
{
  if (!ctx || !action || !obs || !reward || !terminal) return fail(GRLX_ERR_INVALID, "bad argument");
  if (ctx->cfg.env == GRLX_ENV_EXTERNAL) return fail(GRLX_ERR_INVALID, "this context has no environment (GRLX_ENV_EXTERNAL)");
  DRAIN(ctx);
  int rc = step_buffers(ctx);
  if (rc != GRLX_OK) return rc;
  const size_t N = (size_t)ctx->P.n_replicas, D = (size_t)ctx->D;
  double *d_obs = ctx->stage_f64, *d_reward = d_obs + N * GRLX_MAX_DIMS, *d_action = d_reward + N;
  int32_t *d_term = ctx->stage_i32 + N;
  hipError_t e;
  const int32_t *act = stage_active(ctx, active, &e);
  HIP_TRY(e);
  if (active)
  { // inactive rows keep the caller's values
    HIP_TRY(hipMemcpy(d_obs, obs, sizeof(double) * N * D, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_reward, reward, sizeof(double) * N, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_term, terminal, sizeof(int32_t) * N, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMemcpy(d_action, action, sizeof(double) * N, hipMemcpyHostToDevice));
  HIP_TRY(launch_env_advance(ctx->P, act, d_action, d_obs, d_reward, d_term, nullptr));
  HIP_TRY(hipMemcpy(obs, d_obs, sizeof(double) * N * D, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(reward, d_reward, sizeof(double) * N, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(terminal, d_term, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
  uint64_t h[3];                                        // a state outside the sine's domain is reported at once, as grlx_env_step does
  HIP_TRY(launch_step_counts(ctx->P, ctx->scratch, nullptr));
  HIP_TRY(hipMemcpy(h, ctx->scratch, sizeof(h), hipMemcpyDeviceToHost));
  return status_to_error(h[2] & ST_DOMAIN);
}

// the agent kinds the per-step kernels are built for, and the state they keep between calls
static int agent_ready(grlx_ctx *ctx)
{
  const grlx_config &c = ctx->cfg;
  const bool td = c.agent == GRLX_AGENT_SARSA || c.agent == GRLX_AGENT_Q || c.agent == GRLX_AGENT_EXPECTED_SARSA;
  if (!(td || c.agent == GRLX_AGENT_AC) || c.trace == GRLX_TRACE_ACCUMULATING || c.target_interval > 0 || c.projector.safe != 0 ||
      (td && c.action_steps != 3 && c.action_steps != 5))
    return fail(GRLX_ERR_INVALID, "the per-step agent entry points are built for agent/td with predictor/critic/{sarsa, q, expected_sarsa} (3 or 5 actions, "
                                  "replacing or no trace, no target network, safe = 0) and for the actor-critic agent");
  if (ctx->P.tap_capacity > 0 || ctx->P.diag_out) return fail(GRLX_ERR_INVALID, "the per-step agent entry points are not available with taps or diagnostics");
  if (ctx->agent_rep) return GRLX_OK;
  const size_t N = (size_t)ctx->P.n_replicas;
  int rc = step_buffers(ctx);
  if (rc != GRLX_OK) return rc;
  if (!ctx->trace_state)
  { // the predictor's trace between calls, in the layout the actor-critic kernels persist theirs in: empty
    const size_t words = N * 16 * kMaxTrace * 2;
    if (hipMalloc((void **)&ctx->trace_state, words * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); ctx->trace_state = nullptr; return fail(GRLX_ERR_OOM, "no device memory for the agents' traces"); }
    std::vector<uint32_t> init(words, 0u);
    for (size_t i = 0; i < words; i += 2) init[i] = kInvalidPos;
    HIP_TRY(hipMemcpy(ctx->trace_state, init.data(), words * sizeof(uint32_t), hipMemcpyHostToDevice));
    ctx->P.trace_state = ctx->trace_state;
  }
  if (hipMalloc((void **)&ctx->agent_lane, sizeof(uint32_t) * N * 16 * 2) != hipSuccess) { (void)hipGetLastError(); ctx->agent_lane = nullptr; return fail(GRLX_ERR_OOM, "no device memory for the agents' state"); }
  HIP_TRY(hipMemset(ctx->agent_lane, 0, sizeof(uint32_t) * N * 16 * 2));
  if (hipMalloc((void **)&ctx->agent_rep, sizeof(AgentRep) * N) != hipSuccess) { (void)hipGetLastError(); ctx->agent_rep = nullptr; return fail(GRLX_ERR_OOM, "no device memory for the agents' state"); }
  HIP_TRY(hipMemset(ctx->agent_rep, 0, sizeof(AgentRep) * N));
  ctx->P.agent_rep = ctx->agent_rep;
  ctx->P.agent_lane = ctx->agent_lane;
  return GRLX_OK;
}

static int agent_call(grlx_ctx *ctx, int mode, int test, const int32_t *active, double tau, const double *obs, const double *reward,
                      const int32_t *terminal, double *action)
{
  if (!ctx || !obs || (mode != STEP_START && !reward) || (mode != STEP_END && !action)) return fail(GRLX_ERR_INVALID, "bad argument");
  if (mode != STEP_START && tau != 1.) return fail(GRLX_ERR_INVALID, "Agent::step / end: tau must be 1 (environment/modeled:discrete_time = 1)");
  DRAIN(ctx);
  int rc = agent_ready(ctx);
  if (rc != GRLX_OK) return rc;
  // the tables these calls fill grow like the fused path's: looked at every 64 calls, between two calls
  if ((ctx->step_calls++ & 63u) == 63u && ctx->P.logC < ctx->logC_max)
  {
    uint32_t used = 0;
    HIP_TRY(launch_max_load(ctx->P, ctx->n_tables, ctx->max_load, nullptr));
    HIP_TRY(hipMemcpy(&used, ctx->max_load, sizeof(used), hipMemcpyDeviceToHost));
    if ((uint64_t)used * 4u > (1ull << ctx->P.logC))
    {
      uint32_t want = ctx->P.logC;
      while ((uint64_t)used * 8u > (1ull << want) && want < ctx->logC_max) ++want;
      rc = grow_tables(ctx, want);
      if (rc == GRLX_ERR_OOM) { g_err.clear(); ctx->logC_max = ctx->P.logC; }
      else if (rc != GRLX_OK) return rc;
    }
  }
  const size_t N = (size_t)ctx->P.n_replicas;
  const size_t D = (size_t)(ctx->cfg.agent == GRLX_AGENT_AC ? ctx->P.tile.D : ctx->P.tile.D - 1);
  double *d_obs = ctx->stage_f64, *d_reward = d_obs + N * GRLX_MAX_DIMS, *d_action = d_reward + N;
  int32_t *d_term = ctx->stage_i32 + N;
  StepArgs A;
  A.mode = mode;
  A.test = test ? 1 : 0;
  hipError_t e;
  A.active = stage_active(ctx, active, &e);
  HIP_TRY(e);
  HIP_TRY(hipMemcpy(d_obs, obs, sizeof(double) * N * D, hipMemcpyHostToDevice));
  A.obs = d_obs;
  A.reward = d_reward;
  if (mode != STEP_START) HIP_TRY(hipMemcpy(d_reward, reward, sizeof(double) * N, hipMemcpyHostToDevice));
  A.terminal = nullptr;
  if (mode == STEP_STEP && terminal)
  {
    HIP_TRY(hipMemcpy(d_term, terminal, sizeof(int32_t) * N, hipMemcpyHostToDevice));
    A.terminal = d_term;
  }
  A.action = d_action;
  if (mode != STEP_END) HIP_TRY(hipMemcpy(d_action, action, sizeof(double) * N, hipMemcpyHostToDevice));    // rows that do not act keep the caller's values
  if (ctx->poison) HIP_TRY(launch_poison_registers(ctx->poison_pattern, nullptr));
  HIP_TRY(launch_agent_step(ctx->P, A, nullptr));
  if (mode != STEP_END) HIP_TRY(hipMemcpy(action, d_action, sizeof(double) * N, hipMemcpyDeviceToHost));
  else HIP_TRY(hipDeviceSynchronize());
  return GRLX_OK;
}

int grlx_agent_start(grlx_ctx *ctx, int test, const int32_t *active, const double *obs, double *action)
{
  return agent_call(ctx, STEP_START, test, active, 0., obs, nullptr, nullptr, action);
}

int grlx_agent_step(grlx_ctx *ctx, int test, const int32_t *active, double tau, const double *obs, const double *reward, const int32_t *terminal, double *action)
{
  return agent_call(ctx, STEP_STEP, test, active, tau, obs, reward, terminal, action);
}

int grlx_agent_end(grlx_ctx *ctx, int test, const int32_t *active, double tau, const double *obs, const double *reward)
{
  return agent_call(ctx, STEP_END, test, active, tau, obs, reward, nullptr, nullptr);
}

int grlx_project(const grlx_tile_spec *spec, const double *in, int n, uint32_t *out)
{
  if (!spec || !in || !out || n < 0) return fail(GRLX_ERR_INVALID, "bad argument");
  TileParams tp;
  int rc = make_tile_params(*spec, &tp);
  if (rc != GRLX_OK) return rc;
  if (spec->safe != 0) return fail(GRLX_ERR_INVALID, "grlx_project is stateless: projector/tile_coding:safe needs the claim table of a context");
  if (!have_device()) return fail(GRLX_ERR_NO_DEVICE, "no HIP device: grlx has no CPU fallback");
  if (n == 0) return GRLX_OK;
  DevBuf din, dout;
  HIP_TRY(din.alloc(sizeof(double) * (size_t)n * (size_t)tp.D));
  HIP_TRY(dout.alloc(sizeof(uint32_t) * (size_t)n * (size_t)tp.T));
  HIP_TRY(hipMemcpy(din.p, in, sizeof(double) * (size_t)n * (size_t)tp.D, hipMemcpyHostToDevice));
  HIP_TRY(launch_project(tp, din.as<double>(), n, dout.as<uint32_t>(), nullptr));
  HIP_TRY(hipMemcpy(out, dout.p, sizeof(uint32_t) * (size_t)n * (size_t)tp.T, hipMemcpyDeviceToHost));
  return GRLX_OK;
}

int grlx_env_step(const grlx_config *cfg, double *state, const double *action, int n, double *obs, double *reward, int32_t *terminal)
{
  if (!cfg || !state || !action || !obs || !reward || !terminal || n < 0) return fail(GRLX_ERR_INVALID, "bad argument");
  if (cfg->env == GRLX_ENV_EXTERNAL) return fail(GRLX_ERR_INVALID, "GRLX_ENV_EXTERNAL is the caller's environment: nothing to step here");
  DevParams P;
  int rc = make_params(*cfg, &P);
  if (rc != GRLX_OK) return rc;
  if (!have_device()) return fail(GRLX_ERR_NO_DEVICE, "no HIP device: grlx has no CPU fallback");
  if (n == 0) return GRLX_OK;
  int S, D;
  env_dims(cfg->env, &S, &D);
  DevBuf ds, da, dobs, dr, dt, derr;
  HIP_TRY(ds.alloc(sizeof(double) * (size_t)n * S));
  HIP_TRY(da.alloc(sizeof(double) * (size_t)n));
  HIP_TRY(dobs.alloc(sizeof(double) * (size_t)n * D));
  HIP_TRY(dr.alloc(sizeof(double) * (size_t)n));
  HIP_TRY(dt.alloc(sizeof(int32_t) * (size_t)n));
  HIP_TRY(derr.alloc(sizeof(uint32_t)));
  HIP_TRY(hipMemset(derr.p, 0, sizeof(uint32_t)));
  HIP_TRY(hipMemcpy(ds.p, state, sizeof(double) * (size_t)n * S, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(da.p, action, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  HIP_TRY(launch_env_step(P, ds.as<double>(), da.as<double>(), n, dobs.as<double>(), dr.as<double>(), dt.as<int32_t>(), derr.as<uint32_t>(), nullptr));
  HIP_TRY(hipMemcpy(state, ds.p, sizeof(double) * (size_t)n * S, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(obs, dobs.p, sizeof(double) * (size_t)n * D, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(reward, dr.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(terminal, dt.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
  uint32_t err = 0;
  HIP_TRY(hipMemcpy(&err, derr.p, sizeof(err), hipMemcpyDeviceToHost));
  if (err & ST_DOMAIN) return fail(GRLX_ERR_DOMAIN, "environment state left the supported domain (NaN)");
  return GRLX_OK;
}

static int table_op(grlx_ctx *ctx, int table, int op, const int32_t *replica, const uint32_t *idx, int n,
                    const double *arg, double alpha, double *out)
{
  if (!ctx || !replica || !idx || n < 0 || table < 0 || table >= ctx->n_tables) return fail(GRLX_ERR_INVALID, "bad argument");
  if (n == 0) return GRLX_OK;
  for (int i = 0; i < n; ++i)
    if (replica[i] < 0 || replica[i] >= ctx->P.n_replicas) return fail(GRLX_ERR_INVALID, "replica index out of range");
  const int T = ctx->P.tile.T;
  const uint32_t mem = (uint32_t)(table == 1 ? ctx->P.tile_actor.memory : ctx->P.tile.memory);
  for (size_t i = 0; i < (size_t)n * (size_t)T; ++i)
    if (idx[i] != 0xFFFFFFFFu && idx[i] >= mem) return fail(GRLX_ERR_INVALID, "slot index out of range");
  DRAIN(ctx);
  DevBuf dr, di, da, dout;
  HIP_TRY(dr.alloc(sizeof(int32_t) * (size_t)n));
  HIP_TRY(di.alloc(sizeof(uint32_t) * (size_t)n * T));
  HIP_TRY(da.alloc(sizeof(double) * (size_t)n));
  HIP_TRY(dout.alloc(sizeof(double) * (size_t)n));
  HIP_TRY(hipMemcpy(dr.p, replica, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(di.p, idx, sizeof(uint32_t) * (size_t)n * T, hipMemcpyHostToDevice));
  if (arg) HIP_TRY(hipMemcpy(da.p, arg, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  HIP_TRY(launch_table_op(ctx->P, table, op, dr.as<int32_t>(), di.as<uint32_t>(), n, da.as<double>(), alpha, dout.as<double>(), nullptr));
  if (out) HIP_TRY(hipMemcpy(out, dout.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  HIP_TRY(hipDeviceSynchronize());
  return GRLX_OK;
}

int grlx_read(grlx_ctx *ctx, int table, const int32_t *replica, const uint32_t *idx, int n, double *out)
{
  if (!out) return fail(GRLX_ERR_INVALID, "bad argument");
  return table_op(ctx, table, 0, replica, idx, n, nullptr, 0, out);
}

int grlx_write(grlx_ctx *ctx, int table, const int32_t *replica, const uint32_t *idx, int n, const double *target, double alpha)
{
  if (!target) return fail(GRLX_ERR_INVALID, "bad argument");
  return table_op(ctx, table, 1, replica, idx, n, target, alpha, nullptr);
}

int grlx_update(grlx_ctx *ctx, int table, const int32_t *replica, const uint32_t *idx, int n, const double *delta)
{
  if (!delta) return fail(GRLX_ERR_INVALID, "bad argument");
  return table_op(ctx, table, 2, replica, idx, n, delta, 0, nullptr);
}

int grlx_math(int op, const double *x, const double *y, int n, double *out)
{
  if (!x || !out || n < 0 || op < 0 || op > 8 || (op == 3 && !y)) return fail(GRLX_ERR_INVALID, "bad argument");
  if (!have_device()) return fail(GRLX_ERR_NO_DEVICE, "no HIP device: grlx has no CPU fallback");
  if (n == 0) return GRLX_OK;
  DevBuf dx, dy, dout;
  HIP_TRY(dx.alloc(sizeof(double) * (size_t)n));
  HIP_TRY(dy.alloc(sizeof(double) * (size_t)n));
  HIP_TRY(dout.alloc(sizeof(double) * (size_t)n));
  HIP_TRY(hipMemcpy(dx.p, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  if (y) HIP_TRY(hipMemcpy(dy.p, y, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  HIP_TRY(launch_math(op, dx.as<double>(), dy.as<double>(), n, dout.as<double>(), nullptr));
  HIP_TRY(hipMemcpy(out, dout.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  return GRLX_OK;
}

int grlx_rand48_at(int64_t seed, const uint64_t *skip, int n, double *out)
{
  if (!skip || !out || n < 0) return fail(GRLX_ERR_INVALID, "bad argument");
  if (!have_device()) return fail(GRLX_ERR_NO_DEVICE, "no HIP device: grlx has no CPU fallback");
  if (n == 0) return GRLX_OK;
  DevBuf ds, dout;
  HIP_TRY(ds.alloc(sizeof(uint64_t) * (size_t)n));
  HIP_TRY(dout.alloc(sizeof(double) * (size_t)n));
  HIP_TRY(hipMemcpy(ds.p, skip, sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice));
  HIP_TRY(launch_rand48_at(h_seed((long)seed), ds.as<uint64_t>(), n, dout.as<double>(), nullptr));
  HIP_TRY(hipMemcpy(out, dout.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  return GRLX_OK;
}

} // extern "C"
