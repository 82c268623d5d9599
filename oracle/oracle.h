/* oracle.h -- CPU restatement of grl's OnlineLearningExperiment hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP path is compared
 * against; nothing under grl_amd/ (the product) may include, link or call it.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * It is a scalar, single-threaded, IEEE-double restatement (written fresh from
 * reading the reference; no reference source is copied) of these reference
 * files (paths relative to the wcaarls/grl checkout):
 *   base/src/experiments/online_learning.cpp:110-315   trial/step loop
 *   base/src/environments/modeled.cpp:132-276          environment glue + RK4
 *   base/src/environments/pendulum.cpp:40-145          dynamics + swing-up task
 *   base/src/discretizers/uniform.cpp:60-151           action grid
 *   base/src/projectors/tile_coding.cpp:45-149, include/grl/projectors/tile_coding.h:78-151
 *   base/include/grl/projection.h:68-113               IndexProjection::ssub
 *   base/src/representations/linear.cpp:104-268        weight table
 *   base/include/grl/representation.h:79-83            trace update loop
 *   base/include/grl/trace.h:135-263                   enumerated traces
 *   base/src/policies/q.cpp:94-155, base/src/samplers/greedy.cpp:47-218
 *   base/src/agents/td.cpp:50-81, base/src/agents/fixed.cpp
 *   base/src/predictors/sarsa.cpp:98-132, advantage.cpp:71-110, td.cpp:68-91, ac.cpp:72-110
 *   base/include/grl/utils.h:84-187, base/src/deployer.cpp:70-74   RNG streams
 *
 * PINNING.  In ORC_MATH_LIBM mode the pendulum SARSA(lambda) tile-coding
 * configuration with seed 1 reproduces the reference's own golden file
 * tests/template/pendulum-sarsa-tc-0.txt byte for byte (tests/test_oracle_golden.py;
 * fixture committed as tests/golden/pendulum-sarsa-tc-0.txt).  Everything the
 * reference's tests do not pin (Q-learning, actor-critic, cart-pole, acrobot,
 * compass walker) is marked "parity unpinned by reference tests" where it is
 * implemented.
 *
 * MATH MODES.  ORC_MATH_LIBM calls the host libm (sin/cos/fmod/pow/log) exactly
 * where the reference does.  ORC_MATH_PORTABLE replaces the transcendental calls by
 * the operation-by-operation specified routines of portable_math.c (IEEE
 * +,-,*,/,fma,rint only), which the HIP kernels restate independently; in that
 * mode the GPU path must agree with the oracle bit for bit.
 */
#ifndef GRL_ORACLE_H_
#define GRL_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ RNG -- */
/* 48-bit LCG of the drand48 family, restated so results do not depend on the
 * host libc (utils.h:84-137 uses srand48_r/drand48_r/lrand48). */
typedef struct { uint64_t x; } orc_rand48;
void     orc_srand48(orc_rand48 *g, long seed);      /* X = (seed<<16)|0x330E      */
double   orc_drand48(orc_rand48 *g);                 /* next X; X * 2^-48          */
uint32_t orc_lrand48(orc_rand48 *g);                 /* next X; X >> 17            */
/* state after n further draws (O(log n)); used to validate lazy weight init */
void     orc_rand48_jump(orc_rand48 *g, uint64_t n);

/* ------------------------------------------------------- portable math -- */
double orc_psin(double x);
double orc_pcos(double x);
double orc_plog(double x);
double orc_pexp(double x);     /* portable exp (policies; the batch path until round 4) */
double orc_logistic_exp(double x);  /* exp of the batch path's clamped logistic argument, |x| <= 690 (fqi.c D5) */

enum { ORC_MATH_LIBM = 0, ORC_MATH_PORTABLE = 1 };

/* ---------------------------------------------------------------- spec -- */
enum { ORC_ENV_PENDULUM = 0, ORC_ENV_CART_POLE = 1, ORC_ENV_ACROBOT = 2, ORC_ENV_COMPASS_WALKER = 3,
       ORC_ENV_CART_POLE_BALANCING = 4 };   /* dynamics/cart_pole + task/cart_pole/balancing (cart_pole.cpp:239-320) */
enum { ORC_AGENT_SARSA = 0, ORC_AGENT_Q = 1, ORC_AGENT_AC = 2, ORC_AGENT_EXPECTED_SARSA = 3, ORC_AGENT_ADVANTAGE = 4, ORC_AGENT_QV = 5,
       ORC_AGENT_PID = 6 };                 /* agent/fixed + policy/parameterized/pid, proportional gains (pid.cpp:136-179) */
enum { ORC_TRACE_NONE = 0, ORC_TRACE_REPLACING = 1, ORC_TRACE_ACCUMULATING = 2 };
enum { ORC_AC_PROPORTIONAL = 0, ORC_AC_CACLA = 1 };

#define ORC_MAX_DIMS 8      /* projector input dims (obs + action) */
#define ORC_MAX_STATE 12

typedef struct {
  int    tilings;                     /* projector/tile_coding:tilings          */
  int    memory;                      /* :memory (hash table size)              */
  int    dims;                        /* input dims = len(resolution)           */
  double resolution[ORC_MAX_DIMS];    /* :resolution                            */
  double wrapping[ORC_MAX_DIMS];      /* :wrapping (0 = none), unscaled         */
} orc_tile_spec;

typedef struct {
  double init_min, init_max;          /* representation/parameterized/linear    */
  double output_min, output_max;      /* +-DBL_MAX when the yaml gives []       */
  int    limit;                       /* default 1                              */
} orc_linear_spec;

typedef struct {
  /* experiment/online_learning */
  int    test_interval;               /* -1: no test trials                     */
  /* environment/modeled + model/dynamical */
  int    env;
  double control_step;
  int    integration_steps;
  double timeout;
  double randomization;
  int    end_stop_penalty;            /* task/cart_pole/swingup (class default 1; ac_tc.yaml: 0) */
  int    action_penalty;              /* task/cart_pole/swingup (default 0)                      */
  double slope_angle;                 /* model/compass_walker, task/compass_walker/walk (0.004)   */
  double initial_state_variation;     /* task/compass_walker/walk (0.2)                           */
  double negative_reward;             /* task/compass_walker/walk (-100)                          */
  /* discretizer/uniform over the action (dims = 1 for all supported envs) */
  double action_min, action_max;
  int    action_steps;
  /* agent */
  int    agent;                       /* ORC_AGENT_*                            */
  orc_tile_spec   projector;          /* Q: (obs,action) ; AC: critic (obs)     */
  orc_linear_spec representation;     /* Q table / critic V table               */
  double epsilon, decay_rate, decay_min;   /* sampler/epsilon_greedy            */
  double alpha, gamma, lambda;        /* predictor (critic for AC)              */
  int    trace;                       /* ORC_TRACE_*                            */
  /* actor (ORC_AGENT_AC only): policy/action + predictor/ac/action */
  orc_tile_spec   actor_projector;
  orc_linear_spec actor_representation;
  double actor_alpha;
  double sigma;                       /* policy/action:sigma (learning policy)  */
  double theta;                       /* policy/action:theta                    */
  double ac_decay_rate, ac_decay_min; /* policy/action decay                    */
  int    ac_update_method;            /* ORC_AC_*                               */
  double ac_step_limit;               /* <0: none                               */
  /* arithmetic */
  int    math;                        /* ORC_MATH_*                             */
  /* taps */
  int    tap_starts;                  /* 1: also record the start of every trial (terminal = -1) */
  /* predictor/critic/advantage */
  double kappa;                       /* advantage scaling factor (advantage.cpp:188, cfg: 0.2)  */
  /* predictor/critic/qv: Q = projector/representation (table 0), V = actor_projector/actor_representation (table 1) */
  double beta;                        /* state value learning rate (qv.cpp:40, cfg: 0.1)         */
  /* policy/parameterized/pid (ORC_AGENT_PID): `p` gains and `setpoint`, one per observation dimension, one output */
  double pid_p[ORC_MAX_DIMS];
  double pid_setpoint[ORC_MAX_DIMS];
  /* ParameterizedRepresentation target network of the Q table (representation.h:161-306; SARSA and Q-learning read their
   * targets from it, sarsa.cpp:107, advantage.cpp:88): `interval` counts LinearRepresentation::update calls (one per
   * write, one per trace entry: linear.cpp:267), 0 = no target network; `tau` is the synchronisation strength */
  int    target_interval;
  double target_tau;
  /* projector/tile_coding:safe (tile_coding.cpp:39, tile_coding.h:116-151): 0 = off, 1 = a slot is CLAIMED by the hash sum
   * of the first projection that is written through it (single projections claim, the policy's batch projections do
   * not); a later projection with another hash sum that lands on a claimed slot moves on to the next free one */
  int    safe;
  /* experiment/online_learning:test_trials (online_learning.cpp:160-225): greedy episodes per test trial, averaged in the row; 0, 1 = one */
  int    test_trials;
} orc_spec;

/* fill with the values of the reference's tests/pendulum-sarsa-tc.yaml */
void orc_spec_pendulum_sarsa(orc_spec *s);
/* fill with the values of the reference's tests/cart_pole_balancing-pid.yaml (its second golden file on this path) */
void orc_spec_cart_pole_balancing_pid(orc_spec *s);

/* --------------------------------------------------- fine-grained rows -- */
/* a7: TileCodingProjector::_project. in[dims] -> out[tilings]; returns 0, or
 * -1 when the spec is invalid (wrapping*scaling not an integer). */
int orc_tile_project(const orc_tile_spec *t, const double *in, uint32_t *out);
int orc_tile_project_hash(const orc_tile_spec *t, const double *in, uint32_t *out);   /* full hash sums (safe >= 1) */

/* a3/a4/a5: one ModeledEnvironment::step on an explicit state.
 * state[S] is updated in place; obs[D], *reward, *terminal are outputs.
 * Returns tau as the reference's step() would (1 for discrete_time). */
double orc_env_step(const orc_spec *s, double *state, double action,
                    double *obs, double *reward, int *terminal);
int    orc_env_state_dims(int env);
int    orc_env_obs_dims(int env);

/* --------------------------------------------------------- experiment -- */
typedef struct orc_exp orc_exp;

typedef struct {
  int64_t trial;      /* column 1: tt+1-(tt+1)/(test_interval+1)  (or tt)   */
  int64_t steps;      /* column 2: cumulative learning steps                */
  double  reward;     /* column 3: episode return                           */
  double  time;       /* column 4: episode time = sum of tau (online_learning.cpp:203,243) */
} orc_row;

/* per-step tap, for kernel unit tests (learning and test steps alike) */
typedef struct {
  int32_t  test;                 /* 0 learning step, 1 test step              */
  int32_t  action_index;         /* chosen discrete action a' (Q agents)      */
  double   obs[ORC_MAX_DIMS];    /* observation s' after the step             */
  double   action;               /* action value chosen at s'                 */
  double   reward;
  int32_t  terminal;
  int32_t  trace_len;            /* entries in the trace after the update     */
  double   q[8];                 /* Q(s', .) as seen by the policy            */
  double   delta;                /* TD error of the update (learning only)    */
  uint32_t p_idx[32];            /* indices of project(s, a) that was updated */
  double   state[ORC_MAX_STATE]; /* model state after this step (start record: the start state) */
} orc_tap;

typedef struct {
  uint64_t learn_steps, test_steps;
  uint64_t weight_reads;         /* 8-byte reads the algorithm performs       */
  uint64_t weight_rmws;          /* read-modify-writes                        */
  uint64_t trace_entries_sum;    /* sum over learn steps of trace length      */
  uint64_t explorations, ties;
} orc_stats;

/* Construct exactly as `grld -s seed cfg.yaml` instantiates the object tree:
 * srand48(seed), then RNG consumption in YAML order. */
orc_exp *orc_create(const orc_spec *spec, long seed);
void     orc_destroy(orc_exp *e);

/* Run `n_trials` further trials (learning and test trials both count, as in
 * online_learning.cpp:154).  Test-trial rows are appended to rows[] (at most
 * max_rows); returns the number of rows written.  With test_interval < 0 every
 * trial writes a row.  If tap != NULL the first tap_cap steps are recorded and
 * *tap_n receives the count. */
int orc_run(orc_exp *e, int n_trials, orc_row *rows, int max_rows,
            orc_tap *tap, int tap_cap, int *tap_n);

/* The same experiment behind the reference's per-step plug-in interfaces, one call per call of OnlineLearningExperiment::run
 * (online_learning.cpp:172-213); orc_run is written on top of them, so the golden files pin them.  The caller owns the loop:
 * trial / step counters, rows and the steps budget are NOT advanced by these calls.
 *   Environment::start / step (environment.h:48-51 -> ModeledEnvironment, modeled.cpp:132-213) on the experiment's own model state
 *   and random streams; step returns tau.
 *   Agent::start / step / end (agent.h:44-56): test = 0 the learning agent (agent/td, td.cpp:50-81), test = 1 the test agent
 *   (agent/fixed, fixed.cpp:47-65).  start and step return the action taken. */
void   orc_exp_env_start(orc_exp *e, int test, double *obs);
double orc_exp_env_step(orc_exp *e, double action, double *obs, double *reward, int *terminal);
double orc_exp_agent_start(orc_exp *e, int test, const double *obs);
double orc_exp_agent_step(orc_exp *e, int test, double tau, const double *obs, double reward);
void   orc_exp_agent_end(orc_exp *e, int test, double tau, const double *obs, double reward);
/* overwrite the model state (tests that run the environment elsewhere and only want the oracle's final state to compare) */
void   orc_set_state(orc_exp *e, const double *state);

/* Experiment::reset() between two runs of one process (online_learning.cpp:307-308): parameters re-drawn from the continuing
 * thread-local stream, traces cleared, exploration decay back to 1, the run's counters restart; the streams are NOT reseeded.
 * 0, or -1 for graphs whose reset is not restated (target network, safe >= 1, the PID agent). */
int           orc_reset_run(orc_exp *e);
/* experiment/online_learning:steps: orc_run starts no further trial once the learning steps of the run have reached `steps` (0: no budget) */
void          orc_set_steps_budget(orc_exp *e, uint64_t steps);
int64_t       orc_trials(const orc_exp *e);                    /* trials of the run so far (tt) */
void          orc_get_stats(const orc_exp *e, orc_stats *out);
const double *orc_weights(const orc_exp *e, int table);        /* table 0: Q/critic, 1: actor, 2: target network of table 0 */
int64_t       orc_target_syncs(const orc_exp *e);              /* synchronisations of the target network so far */
/* {action: load} of a representation (representation.h:231-263): overwrite all n = memory weights; 0 or -1 */
int           orc_set_weights(orc_exp *e, int table, const double *w, size_t n);
void          orc_get_state(const orc_exp *e, double *state);  /* current env state           */
void          orc_rng_states(const orc_exp *e, uint64_t out[4]); /* G, TL, S1, S2              */

/* Text of one row in the reference's golden 3-column layout (setw(15) x3,
 * default ostream precision), newline terminated. Returns bytes written. */
int orc_format_row(const orc_row *r, char *buf, size_t cap);

/* weight-init value of slot i for a table whose thread-local stream was seeded
 * by `tl_seed` (= the lrand48() draw) *before* any other TL draw. */
double orc_lazy_weight(uint32_t tl_seed, uint64_t draws_before, uint32_t slot,
                       double init_min, double init_max);

/* ------------------------------------------------------- batch path (FQI) -- */
/* experiment/batch_learning + predictor/fqi + representation/iterative + representation/parameterized/ann over
 * projector/pre/normalizing (tests/pendulum-fqi-ann.yaml; BASELINE configs[4]).  fqi.c: PARITY UNPINNED, deviations D1-D5. */
typedef struct {
  orc_spec base;          /* env (pendulum), control_step, integration_steps, timeout, action_min/max/steps, gamma, math */
  int      batch_size;    /* experiment/batch_learning:batch_size: transitions drawn per batch                          */
  int      iterations;    /* predictor/fqi:iterations                                                                   */
  int      epochs;        /* representation/iterative:epochs                                                            */
  int      hidden;        /* representation/parameterized/ann:hiddens = [hidden]                                        */
  int      sum_order;     /* 0: gradient summed in sample order (the reference); 1: the GPU's fixed tree (fqi.c D3)     */
  double   gamma_tau;     /* pow(gamma, control_step), evaluated once by the caller with libm                           */
  double   eta;           /* representation/parameterized/ann:eta: 0 = RPROP, > 0 = gradient descent, < 0 = RMSprop     */
} orc_fqi_spec;
typedef struct orc_fqi orc_fqi;
void     orc_fqi_spec_pendulum(orc_fqi_spec *s);                 /* the reference's tests/pendulum-fqi-ann.yaml */
orc_fqi *orc_fqi_create(const orc_fqi_spec *spec, long seed);
void     orc_fqi_destroy(orc_fqi *f);
/* one batch: draw batch_size transitions, FQIPredictor::rebuild, one greedy test trial -> *row (batch, transitions, return) */
int      orc_fqi_run_batch(orc_fqi *f, orc_row *row);
const double *orc_fqi_params(const orc_fqi *f, int *n);          /* layer 1 (n_in+1) x H column-major, then layer 2 (H+1) x 1 */
size_t   orc_fqi_transitions(const orc_fqi *f, const double **in, const double **next_obs, const double **reward, const double **targets);
void     orc_fqi_info(const orc_fqi *f, double *maxdelta, int *iterations, double *error);
double   orc_fqi_q(const orc_fqi *f, const double *obs, double action);
void     orc_fqi_rng(const orc_fqi *f, uint64_t out[3]);         /* G, TL, weight-init stream */

#ifdef __cplusplus
}
#endif
#endif /* GRL_ORACLE_H_ */
