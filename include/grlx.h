/* grlx.h -- C ABI of the MI355X-native runner for grl's online-learning hot path.
 *
 * This is the drop-in boundary: plain C, POD structs, caller-allocated buffers,
 * `int` status returns (0 = GRLX_OK, negative = error; grlx_last_error() has the
 * text), no exceptions, no torch/Eigen types.  One context owns the replicas of
 * one GPU; a context is not thread-safe.  Every entry point names the reference
 * interface it replaces (paths relative to the wcaarls/grl checkout); the
 * reference-side binding a grl maintainer would add is shown in INTEGRATION.md.
 *
 * There is NO CPU fallback: every compute entry point runs HIP kernels on the
 * current device and returns GRLX_ERR_NO_DEVICE when there is none.
 */
#ifndef GRLX_H_
#define GRLX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): grlx_config grew test_trials / reserved0 in round 3 and the per-step entry points were added; a binding compiled against
 * version 1's header must not drive this library (grlx_create would read past the end of its struct). */
#define GRLX_ABI_VERSION 2

enum {
  GRLX_OK = 0,
  GRLX_ERR_INVALID = -1,       /* bad argument / unsupported configuration (grl: bad_param)   */
  GRLX_ERR_NO_DEVICE = -2,     /* no HIP device: the product never computes on the CPU        */
  GRLX_ERR_HIP = -3,           /* HIP runtime error                                           */
  GRLX_ERR_TABLE_FULL = -4,    /* a replica's sparse weight table overflowed (raise capacity) */
  GRLX_ERR_DOMAIN = -5,        /* portable sin/cos argument outside |x| < 2^20                */
  GRLX_ERR_ROWS_FULL = -6,     /* more test rows than reserved at create                      */
  GRLX_ERR_OOM = -7
};

/* YAML type strings of the reference these enumerators stand for */
enum { GRLX_ENV_PENDULUM = 0,        /* dynamics/pendulum + task/pendulum/swingup   (pendulum.cpp)       */
       GRLX_ENV_CART_POLE = 1,       /* dynamics/cart_pole + task/cart_pole/swingup (cart_pole.cpp)      */
       GRLX_ENV_ACROBOT = 2,         /* dynamics/acrobot + task/acrobot/balancing   (acrobot.cpp)        */
       GRLX_ENV_COMPASS_WALKER = 3,  /* sandbox/compass_walker (walk task)          (compass_walker.cpp) */
       GRLX_ENV_CART_POLE_BALANCING = 4,/* dynamics/cart_pole + task/cart_pole/balancing (cart_pole.cpp:239-320): the task of the
                                        reference's tests/cart_pole_balancing-pid.yaml; grlx_env_step only (grlx_create refuses it) */
       GRLX_ENV_EXTERNAL = 5         /* no environment in the context: it is the caller's (any grl Environment: gym, a robot, a CPU
                                        simulator).  Observation dimensions = the projector's input dimensions (minus the action's for
                                        the Q agents).  Only the per-step agent entry points (grlx_agent_start / _step / _end) and the
                                        table operators serve such a context; grlx_run, grlx_env_start and grlx_env_advance refuse it.
                                        The environment fields of grlx_config (control_step, timeout, ...) are not read. */ };
enum { GRLX_AGENT_SARSA = 0,         /* agent/td + policy/discrete/q + predictor/critic/sarsa (sarsa.cpp)     */
       GRLX_AGENT_Q = 1,             /* ... + predictor/critic/q (advantage.cpp:71-110)                       */
       GRLX_AGENT_AC = 2,            /* policy/action + predictor/ac/action + predictor/critic/td (ac.cpp)    */
       GRLX_AGENT_EXPECTED_SARSA = 3,/* ... + predictor/critic/expected_sarsa (sarsa.cpp:167-194)              */
       GRLX_AGENT_ADVANTAGE = 4,     /* ... + predictor/critic/advantage (advantage.cpp:222-268), `kappa`     */
       GRLX_AGENT_QV = 5             /* ... + predictor/critic/qv (qv.cpp:74-108): table 0 = Q, table 1 = V
                                        (v_projector / v_representation in the actor_* fields), `beta`     */ };
enum { GRLX_TRACE_NONE = 0, GRLX_TRACE_REPLACING = 1, GRLX_TRACE_ACCUMULATING = 2 };   /* trace.h:208-263 */

#define GRLX_MAX_DIMS 8
#define GRLX_MAX_STATE 12
#define GRLX_MAX_ACTIONS 8
/* An episode ends on the task's terminal state, at the latest on time > timeout; the fused kernels have no
 * other exit, so grlx_create refuses a non-finite or negative timeout and more control steps per episode
 * (timeout / control_step; twice that for the compass walker's test episodes) than this. */
#define GRLX_MAX_EPISODE_STEPS 100000

/* projector/tile_coding (tile_coding.cpp:34-80): tilings, memory, resolution, wrapping; safe = 0.
 * HARD LIMIT of the fused rollout kernels (grlx_create): tilings == 16 -- one lane per tiling, 16 lanes per replica, four (or
 * eight) replicas per 64-lane wavefront; anything else is GRLX_ERR_INVALID.  The reference takes any count (tile_coding.cpp:47);
 * the stateless operator grlx_project takes 1..32.  memory: 1 .. 2^26-1 slots. */
typedef struct {
  int32_t tilings;
  int32_t memory;
  int32_t dims;
  int32_t safe;                       /* collision detection (tile_coding.cpp:39): 0 = off; 1 = claim on write (single projections
                                         claim their slots with their hash sum, tile_coding.h:116-151) -- served by the plain kernel
                                         for SARSA / Q-learning on the pendulum and the acrobot; 2 = claim always: the policy's batch
                                         projections (tile_coding.h:67-73) claim too, one variant after the other */
  double  resolution[GRLX_MAX_DIMS];
  double  wrapping[GRLX_MAX_DIMS];
} grlx_tile_spec;

/* representation/parameterized/linear (linear.cpp:34-101) */
typedef struct {
  double  init_min, init_max;
  double  output_min, output_max;     /* +-DBL_MAX for the yaml's empty vectors */
  int32_t limit;                      /* default 1 (linear.h:46-49)             */
  int32_t reserved;
} grlx_linear_spec;

/* One fused experiment graph:
 *   experiment/online_learning { environment/modeled { model/dynamical, task },
 *                                agent/td { policy, predictor }, test_agent agent/fixed }
 * Field names follow the reference's YAML parameter names. */
typedef struct {
  uint32_t struct_size;               /* = sizeof(grlx_config), ABI check                       */
  int32_t  n_replicas;                /* independent-seed replicas on this GPU (experiment/multi analogue, multi.cpp:44-75) */
  int32_t  test_interval;             /* experiment: -1 = none                                  */
  int32_t  env;                       /* GRLX_ENV_*                                             */
  double   control_step;              /* model/dynamical                                        */
  int32_t  integration_steps;
  int32_t  discrete_time;             /* environment/modeled: must be 1                         */
  double   timeout;                   /* task                                                   */
  double   randomization;
  double   action_min, action_max;    /* discretizer/uniform over task action range             */
  int32_t  action_steps;
  int32_t  agent;                     /* GRLX_AGENT_*                                           */
  grlx_tile_spec   projector;
  grlx_linear_spec representation;
  double   epsilon, decay_rate, decay_min;    /* sampler/epsilon_greedy                         */
  double   alpha, gamma, lambda;              /* predictor                                      */
  int32_t  trace;                             /* GRLX_TRACE_*                                   */
  int32_t  tap_starts;                /* 1: taps also record the start of every trial (terminal = -1): the rows of a transition log */
  /* actor-critic only */
  grlx_tile_spec   actor_projector;
  grlx_linear_spec actor_representation;
  double   actor_alpha, sigma, theta, ac_decay_rate, ac_decay_min, ac_step_limit;
  int32_t  ac_update_method;
  /* GPU-side sizing (no reference counterpart) */
  int32_t  table_log2_capacity;       /* sparse table slots per replica = 2^this; 0 = default   */
  int32_t  max_rows;                  /* test rows reserved per replica                          */
  int32_t  tap_replica;               /* -1: off; else record per-step taps of that replica      */
  int32_t  tap_capacity;
  /* task/cart_pole/swingup (cart_pole.cpp:110-130); class defaults 1 / 0, cfg/cart_pole/ac_tc.yaml: 0 / 0 */
  int32_t  end_stop_penalty;
  int32_t  action_penalty;
  int32_t  force_generic;             /* 1: never pick a compile-time specialised kernel (tests) */
  /* model/compass_walker + task/compass_walker/walk (compass_walker.cpp:41-60, 198-249) */
  double   slope_angle;               /* default 0.004 */
  double   initial_state_variation;   /* default 0.2   */
  double   negative_reward;           /* default -100  */
  /* predictor/critic/advantage (advantage.cpp:183-213) */
  double   kappa;                     /* advantage scaling factor (cfg/pendulum/advantage_tc.yaml: 0.2) */
  /* predictor/critic/qv (qv.cpp:35-64): state-value learning rate; V's projector / representation
   * are given in actor_projector / actor_representation (the second table of the context) */
  double   beta;
  /* GPU-side layout (no reference counterpart): replicas carried by one wavefront.  0 = automatic: 4 (16 lanes per
   * replica, lane = tiling) while the batch has no more than 4 replicas per SIMD of the device, else 8 (two sub-batches
   * of four share one environment phase; grlx_rollout_wide.h).  4 / 8 force the choice (tests); the actor-critic kernel also
   * takes 12 and 16 (three / four sub-batches, the lane state of those beyond the second parked in registers; with 12 a wave
   * owns ceil(replicas / waves) replicas and rotates them through its slots trial by trial -- grlx_rollout_ac_wide.h; batches that
   * would give a wave more than 64 replicas run with 8).  Automatic for the actor-critic: 16 from 15 replicas per SIMD up, 12 for 9-14.
   * 16 also for the TD agents on the acrobot / the compass walker with three actions (round 4; automatic from 15 replicas per SIMD up),
   * 32 for the TD agents on the compass walker (eight sub-batches; automatic from 30 replicas per SIMD up).  Results are identical. */
  int32_t  replicas_per_wave;
  /* 1: the taps are recorded by the PRODUCTION ordering of the rollout kernel (TD update applied one pass later, under the
   * next step's table loads) instead of the in-place diagnostic ordering: per-step parity of the kernel that is benchmarked.
   * Built for the pendulum and the acrobot with 3 actions (SARSA / Q / Expected SARSA, replacing or no trace). */
  int32_t  tap_deferred;
  /* representation/parameterized/linear: interval, tau (ParameterizedRepresentation, representation.h:161-306): a TARGET
   * NETWORK on the Q table.  target_interval > 0: SARSA / Q-learning read their targets from a second copy of the
   * parameters (sarsa.cpp:107, advantage.cpp:88), synchronised as tau*params + (1-tau)*target (tau = 0: plain copy) every
   * target_interval LinearRepresentation::update calls (linear.cpp:267: one per write and one per trace entry).  Served by
   * its own plain kernel (pendulum / acrobot, 3 actions, replacing or no trace); 0 = none. */
  int32_t  target_interval;
  /* GPU-side layout (no reference counterpart), wide actor-critic kernel: at most this many wavefronts are launched; each
   * carries replicas_per_wave replicas at a time and takes the next unstarted replica from a device-side counter whenever
   * one of its slots has finished its trials (episodes of a learning batch are ragged: without the queue a wave idles until
   * its slowest replica is done).  0 = automatic: one wave per SIMD of the device.  Results do not depend on it (tests). */
  int32_t  wave_limit;
  /* Sparse weight tables: table_log2_capacity (above) is the INITIAL size, 2^k entries per replica and table.  Between
   * launches -- at the start of a grlx_run whose predecessor has been waited for (grlx_sync or any read-back) -- the fullest
   * table of the context is looked at, and beyond a quarter of the capacity all tables are re-hashed into larger ones
   * (results do not depend on it: positions are internal).  table_log2_max bounds the growth: 0 = up to 2^26; a value equal
   * to the initial size = fixed tables, overflow is then the sticky error GRLX_ERR_TABLE_FULL, as within one launch. */
  int32_t  table_log2_max;
  double   target_tau;
  /* experiment/online_learning:test_trials (online_learning.cpp:160-225): a test trial is `test_trials` greedy episodes, each begun with
   * environment start and agent start; reward and time keep adding up across them and the row holds their means.  0 and 1 = one episode.
   * Every kernel family has it (round 4: QV, advantage, the accumulating trace, target networks / safe too); not with taps. */
  int32_t  test_trials;
  int32_t  reserved0;
} grlx_config;

typedef struct grlx_ctx grlx_ctx;

/* per-step record of the tapped replica (debug / parity tests / transition log).  A record with
 * terminal = -1 (only with tap_starts) is the start of a trial: first observation and first action. */
typedef struct {
  int32_t  test, action_index, terminal, trace_len;
  double   obs[GRLX_MAX_DIMS];
  double   action, reward, delta;
  double   q[GRLX_MAX_ACTIONS];
  uint32_t p_idx[32];
  double   state[GRLX_MAX_STATE];     /* model state after this step (start record: the start state) */
} grlx_tap;

const char *grlx_last_error(void);
int  grlx_abi_version(void);
/* How the library was built: "device-asm+mir-exec-prologue-fix/2" when grl_amd/_build.py built it (the only supported
 * build: it passes the device code through the work-around for a register-allocation bug of ROCm 7.2's compiler,
 * DESIGN.md 4.1f), "" for a plain `hipcc -shared`, whose kernels may read stale lanes.  Bindings refuse the latter
 * (grl_amd/capi.py: load(); the reference-side addon: integration/addons/grlx). */
const char *grlx_build_pipeline(void);
int  grlx_device_count(void);

/* Fill *cfg with the values of the reference's tests/pendulum-sarsa-tc.yaml. */
void grlx_config_pendulum_sarsa(grlx_config *cfg);
/* Fill *cfg with the values of the reference's cfg/cart_pole/ac_tc.yaml (actor-critic, two tables). */
void grlx_config_cart_pole_ac(grlx_config *cfg);

/* Replaces: Configurator::instantiate of the experiment subtree (configurable.cpp:603-715)
 * for n_replicas deep clones (multi.cpp:49-59) after `srand48(seeds[r])`
 * (deployer.cpp:70-74).  RNG streams are consumed exactly in the reference's
 * YAML instantiate order; the 8,388,608-draw weight
 * initialisation (linear.cpp:110-121) is performed lazily per touched slot by
 * LCG jump-ahead and is bit-identical. */
int  grlx_create(const grlx_config *cfg, const int64_t *seeds, grlx_ctx **out);
int  grlx_destroy(grlx_ctx *ctx);

/* Replaces: OnlineLearningExperiment::run (online_learning.cpp:110-315) for every
 * replica: advance each by n_trials trials (learning and test trials both
 * count).  Asynchronous on `stream` (a hipStream_t passed as void*; NULL = default). */
int  grlx_run(grlx_ctx *ctx, int n_trials, void *stream);
/* Wait for the stream, then report sticky per-replica error flags (table full, ...). */
int  grlx_sync(grlx_ctx *ctx, void *stream);
/* Stream contract: grlx_run and grlx_curve_stats are asynchronous on the caller's stream.  Every entry point
 * that reads a context back to the host or works on its tables (grlx_rows, grlx_read_rows, grlx_read_row_times,
 * grlx_step_counts, grlx_get_*, grlx_table_load, grlx_export_weights, grlx_load_weights, grlx_read_taps,
 * grlx_read_diag, grlx_read / _write / _update) first waits for the stream of the context's last grlx_run,
 * so it never observes rollouts in flight -- also on a hipStreamNonBlocking stream. */

/* Test rows written so far (same for every replica). */
int  grlx_rows(grlx_ctx *ctx);
/* Replaces the per-replica `<output>-<run>@<i>.txt` rows (online_learning.cpp:238-262):
 * copy rows [first, first+count) of `replica` to host arrays (columns 1-3). */
int  grlx_read_rows(grlx_ctx *ctx, int replica, int first, int count,
                    int64_t *trial, int64_t *steps, double *reward);
/* column 4 of the same rows (online_learning.cpp:243 `total_time`): the sum of the tau returned by
 * Environment::step over the trial -- under discrete_time (modeled.cpp:209-212) the number of steps. */
int  grlx_read_row_times(grlx_ctx *ctx, int replica, int first, int count, double *episode_time);
/* Device-side learning-curve statistics over this GPU's replicas, written to a
 * DEVICE buffer out[count][3] = {sum reward, sum reward^2, replica count};
 * fixed reduction order (bitwise reproducible).  Input of the one RCCL
 * all-reduce of the multi-GPU path. */
int  grlx_curve_stats(grlx_ctx *ctx, int first, int count, double *out_dev, void *stream);

/* Which instantiation of the rollout kernel the last grlx_run launched (tests and benchmarks check that
 * the configuration they mean to measure takes the path they mean to measure). */
enum { GRLX_KERNEL_NONE = 0,          /* nothing launched yet                                                  */
       GRLX_KERNEL_GENERIC = 1,       /* parameters read at run time                                           */
       GRLX_KERNEL_SPECIALISED = 2,   /* compile-time instantiation of a reference yaml (pendulum tile-coding
                                         SARSA / Q / Expected SARSA, cart-pole actor-critic)                   */
       GRLX_KERNEL_IN_PLACE = 3       /* diagnostic instantiation: taps / stamps, TD update applied in place   */ };
int  grlx_last_kernel(grlx_ctx *ctx);
/* Replicas per wavefront of the context's rollout kernels (4 or 8): the layout chosen at create. */
int  grlx_replicas_per_wave(grlx_ctx *ctx);

/* counters: total env steps executed by all replicas since create */
int  grlx_step_counts(grlx_ctx *ctx, uint64_t *learn_steps, uint64_t *test_steps);

/* --- diagnostics: a separately compiled, stamped build of the rollout kernel
 * (s_memtime per phase).  Its run time is NOT representative; only the shares.
 * enable = 1: the instantiation that applies each TD update in place (the one
 * that also records taps); enable = 2: the production ordering (update applied
 * one pass later, under the next step's table loads; pendulum with 3 actions
 * only); 0: off. */
int  grlx_set_diag(grlx_ctx *ctx, int enable);
int  grlx_read_diag(grlx_ctx *ctx, uint64_t *out /*[waves][8] cycle sums*/, int cap_waves, int *n_waves);

/* --- state inspection (parity tests) ------------------------------------ */
int  grlx_get_env_state(grlx_ctx *ctx, int replica, double *state /*[GRLX_MAX_STATE]*/);
int  grlx_get_rng(grlx_ctx *ctx, int replica, uint64_t out[4] /* G, TL, S1, S2 */);
/* Replaces reading LinearRepresentation::params_ (linear.cpp; .dat dump
 * representation.h:201-263): current weights of the given reference slots. */
int  grlx_get_weights(grlx_ctx *ctx, int table, int replica, const uint32_t *slots, int n, double *out);
int  grlx_table_load(grlx_ctx *ctx, int table, int replica, uint32_t *n_slots_used);
/* Current table size (log2 of the entries per replica and table), and explicit growth to 2^new_log2 (waits for the context's
 * work first; no-op if the tables are at least that large). */
int  grlx_table_capacity(grlx_ctx *ctx, uint32_t *log2_entries);
int  grlx_grow_tables(grlx_ctx *ctx, uint32_t new_log2);
/* The trial loop of OnlineLearningExperiment::run with BOTH of its bounds (online_learning.cpp:154: `(!trials_ || tt < trials_) &&
 * (!steps_ || ss < steps_)`): every replica runs at most max_trials further trials and starts none once its learning steps of the run
 * (since grlx_create / grlx_reset_run) have reached `steps` (> 0).  Replicas stop at trials of their own, so their rows are ragged:
 * grlx_replica_rows gives a replica's count, grlx_curve_stats counts per row.  One launch per call.  Every kernel family has it
 * (round 4); GRLX_ERR_INVALID with taps or diagnostics. */
int  grlx_run_steps(grlx_ctx *ctx, int max_trials, uint64_t steps, void *stream);
int  grlx_replica_rows(grlx_ctx *ctx, int replica);          /* rows replica `replica` has written (grlx_rows: replica 0) */
/* Experiment::reset() between two runs of `runs: N` (online_learning.cpp:307-308; Configurable::reset, configurable.h:770-776):
 * the representations' parameters are drawn again from the CONTINUING thread-local stream (linear.cpp:104-125), predictors clear
 * their traces (sarsa.cpp:60-66), epsilon-greedy and the action policy set decay_ = 1 (greedy.cpp:140-141, action.cpp:93-97); the
 * run's step / trial counters and rows start again; no stream is reseeded.  A target network draws again too, before its representation
 * (the walk visits the provided `target` object first: configurable.cpp:690-712, 754-757), and is synchronised from the two fresh vectors as
 * at construction; the claims of projector/tile_coding:safe are dropped (tile_coding.cpp:82-89). */
int  grlx_reset_run(grlx_ctx *ctx);
/* Replaces reading target()->params() (representation.h:266-282): the target network's current value of the given
 * reference slots of the Q table, and the number of synchronisations so far; contexts with target_interval > 0 only. */
int  grlx_get_target_weights(grlx_ctx *ctx, int replica, const uint32_t *slots, int n, double *out, uint32_t *n_syncs);
/* Replaces ParameterizedRepresentation's {action: load} (representation.h:231-263, driven by
 * experiment/online_learning:load_file, online_learning.cpp:140-150): setParams() with the raw
 * little-endian double[memory] image of a .dat file.  Every weight of `table` of the replicas
 * [first_replica, first_replica + n_replicas) becomes dense[slot]; RNG streams, environment state
 * and counters are untouched, as in the reference.  The image (count must equal the table's
 * memory, else GRLX_ERR_INVALID like the reference's "Configuration mismatch") is copied once to
 * the device and shared by those replicas; their sparse tables are cleared and re-created on
 * first touch from it.  Actor-critic contexts accept it only before the first grlx_run.  With a target network (table 0 of a context with
 * target_interval > 0) the load is followed by synchronize() as in the reference (representation.h:256-257): the target becomes
 * tau * image + (1 - tau) * target for EVERY slot -- one dense vector of `memory` doubles per replica on the device while tau != 0
 * (GRLX_ERR_OOM beyond 8 GiB of them). */
int  grlx_load_weights(grlx_ctx *ctx, int table, int first_replica, int n_replicas, const double *dense, uint64_t count);
/* Replaces ParameterizedRepresentation's {action: save} (representation.h:201-229): the DENSE
 * parameter vector double[memory] of one replica's table, little-endian as grl's .dat files hold it
 * (untouched slots carry their lazily computed initial value).  out: host buffer of `memory` doubles. */
int  grlx_export_weights(grlx_ctx *ctx, int table, int replica, double *out);
int  grlx_read_taps(grlx_ctx *ctx, grlx_tap *out, int cap, int *n);

/* --- fine-grained batched operators (host pointers; each call copies in, runs a
 *     HIP kernel, copies out).  They mirror the plug-in interfaces one to one. */

/* Projector::project -> TileCodingProjector::_project (tile_coding.cpp:103-149):
 * in[n][dims] -> out[n][tilings] reference slot indices. */
int  grlx_project(const grlx_tile_spec *spec, const double *in, int n, uint32_t *out);

/* Environment::step -> ModeledEnvironment::step (modeled.cpp:160-213):
 * state[n][S] updated in place; obs[n][D], reward[n], terminal[n]. */
int  grlx_env_step(const grlx_config *cfg, double *state, const double *action, int n,
                   double *obs, double *reward, int32_t *terminal);
int  grlx_env_dims(int env, int *state_dims, int *obs_dims);

/* Representation::read / write / update on a context's table
 * (linear.cpp:136-184, 186-196, 198-216): idx[n][tilings] reference slots,
 * replica[n] selects the table instance.  Rows are applied in order. */
int  grlx_read(grlx_ctx *ctx, int table, const int32_t *replica, const uint32_t *idx, int n, double *out);
int  grlx_write(grlx_ctx *ctx, int table, const int32_t *replica, const uint32_t *idx, int n,
                const double *target, double alpha);
int  grlx_update(grlx_ctx *ctx, int table, const int32_t *replica, const uint32_t *idx, int n,
                 const double *delta);

/* ---------------------------------------------------------------------------------------------------------------
 * The per-step plug-in interfaces on the replicas of a context: one call of the reference's interface for EVERY replica per
 * call, for graphs in which only one side of the loop of OnlineLearningExperiment::run (online_learning.cpp:172-213) lives on the
 * GPU -- the agent beside an environment of the caller's, or the environment beside an agent of the caller's -- from the first
 * step of a trial on.  (grlx_run stays the fast path: it fuses both sides.)  Host pointers, [n_replicas] rows each.
 *   active   NULL = every replica takes part; else only those with a non-zero entry (episodes of different replicas end at
 *            different steps); the output rows of the others are left as the caller passed them.
 * The loop is the caller's: trial / step counters, rows and the steps budget are not advanced by these calls.  State between
 * calls lives where the fused kernels keep it between launches (streams, decay, the predictor's trace), so grlx_run and these
 * calls can be mixed on one context at trial boundaries. */

/* Environment::start (environment.h:48 -> ModeledEnvironment::start, modeled.cpp:132-158): test = 0 a learning trial's start state,
 * 1 a test trial's; the start state is drawn from the replica's own random streams; obs[n_replicas][obs_dims]. */
int  grlx_env_start(grlx_ctx *ctx, int test, const int32_t *active, double *obs);
/* Environment::step (environment.h:49-51 -> ModeledEnvironment::step, modeled.cpp:160-213) on the replicas' model states:
 * action[n_replicas] -> obs[n_replicas][obs_dims], reward[n_replicas], terminal[n_replicas] (0, 1 = timed out, 2 = absorbing); tau = 1
 * (discrete_time).  (grlx_env_step above is the stateless form: explicit states, no context.) */
int  grlx_env_advance(grlx_ctx *ctx, const int32_t *active, const double *action, double *obs, double *reward, int32_t *terminal);
/* Agent::start (agent.h:44-47): test = 0 the learning agent (agent/td, td.cpp:50-61: the predictor's trace is cleared -- not the
 * actor-critic's critic, ac.cpp:170-173 --, time_ = 0, the policy acts), test = 1 the test agent (agent/fixed, fixed.cpp:47-51: the
 * greedy / noise-free policy over the same tables).  obs[n_replicas][obs_dims] -> action[n_replicas]. */
int  grlx_agent_start(grlx_ctx *ctx, int test, const int32_t *active, const double *obs, double *action);
/* Agent::step (agent.h:49-52; td.cpp:63-74: act at obs, then predictor->update(prev_obs, prev_action, tau, reward, obs, action);
 * fixed.cpp:53-61: act).  tau must be 1.  terminal: NULL, or [n_replicas]: a replica whose entry is 2 gets Agent::end instead
 * (online_learning.cpp:210-213) and no action. */
int  grlx_agent_step(grlx_ctx *ctx, int test, const int32_t *active, double tau, const double *obs, const double *reward,
                     const int32_t *terminal, double *action);
/* Agent::end (agent.h:54-56; td.cpp:76-81: the update with an empty next action, target = reward; agent/fixed: nothing). */
int  grlx_agent_end(grlx_ctx *ctx, int test, const int32_t *active, double tau, const double *obs, const double *reward);
/* Built for agent/td with predictor/critic/{sarsa, q, expected_sarsa} (3 or 5 actions, replacing or no trace, no target network,
 * safe = 0) and for the actor-critic agent; GRLX_ERR_INVALID otherwise. */

/* device math used by the environments (bit-identical to the documented
 * portable specification): op 0 sin, 1 cos, 2 log, 3 fmod(x, y[i]), 4 sqrt, 5 x/6 (the 3-operation exact form used by RK4),
 * 6 / 7 / 8 the small-angle-aware sin, cos and sin+cos the compass walker uses (bitwise equal to 0 / 1 / their sum) */
int  grlx_math(int op, const double *x, const double *y, int n, double *out);

/* Rand / RandGen streams (utils.h:84-137) evaluated on the device:
 * out[n] = drand48 of the stream seeded by srand48(seed) after `skip[i]` draws. */
int  grlx_rand48_at(int64_t seed, const uint64_t *skip, int n, double *out);

/* ---------------------------------------------------------------------------------------------------------------
 * The batch path (BASELINE.json configs[4]): experiment/batch_learning + predictor/fqi + representation/iterative +
 * representation/parameterized/ann over projector/pre/normalizing, as the reference's tests/pendulum-fqi-ann.yaml
 * composes them; n_replicas independent-seed copies per GPU.
 * PARITY UNPINNED: the reference's ANN initialisation (Eigen's Random over libc rand(), ann.cpp:97) and its
 * hidden-layer delta (mismatched Eigen dimensions, ann.cpp:249) cannot be reproduced; the four documented
 * deviations are listed at the top of oracle/fqi.c and in DESIGN.md.  Everything else follows the cited lines. */
typedef struct {
  uint32_t struct_size;               /* = sizeof(grlx_fqi_config)                                                */
  int32_t  n_replicas;
  int32_t  env;                       /* GRLX_ENV_PENDULUM (the task must support invert(), pendulum.cpp:147-155) */
  int32_t  integration_steps;         /* model/dynamical                                                          */
  double   control_step;
  double   timeout;                   /* task/pendulum/swingup                                                    */
  double   action_min, action_max;    /* discretizer/uniform over the task's action range                         */
  int32_t  action_steps;
  int32_t  batch_size;                /* experiment/batch_learning:batch_size (transitions drawn per batch)       */
  double   gamma;                     /* predictor/fqi                                                            */
  int32_t  iterations;
  int32_t  epochs;                    /* representation/iterative:epochs                                          */
  int32_t  hidden;                    /* representation/parameterized/ann:hiddens = [hidden]                      */
  int32_t  max_batches;               /* batches reserved (transition store and rows): experiment:batches         */
  double   eta;                       /* representation/parameterized/ann:eta (ann.cpp:198-221): 0 = RPROP        */
} grlx_fqi_config;
typedef struct grlx_fqi_ctx grlx_fqi_ctx;
/* Fill *cfg with the values of the reference's tests/pendulum-fqi-ann.yaml. */
void grlx_fqi_config_pendulum(grlx_fqi_config *cfg);
/* Replaces Configurator::instantiate of that experiment for n_replicas clones after srand48(seeds[r]). */
int  grlx_fqi_create(const grlx_fqi_config *cfg, const int64_t *seeds, grlx_fqi_ctx **out);
int  grlx_fqi_destroy(grlx_fqi_ctx *ctx);
/* Replaces one pass of the batch loop of BatchLearningExperiment::run (batch_learning.cpp:105-188) for every replica:
 * batch_size random transitions (one model step each), FQIPredictor::rebuild over the whole store (fqi.cpp:205-285),
 * one greedy test trial -> one row.  Asynchronous on `stream`. */
int  grlx_fqi_run_batch(grlx_fqi_ctx *ctx, void *stream);
int  grlx_fqi_sync(grlx_fqi_ctx *ctx, void *stream);
/* rows of `<output>-<run>.txt` (batch_learning.cpp:179): batch, batch*batch_size, return of the test trial */
int  grlx_fqi_read_rows(grlx_fqi_ctx *ctx, int replica, int first, int count, int64_t *batch, int64_t *transitions, double *reward);
/* ANNRepresentation::params_: layer 1 (inputs+1) x hidden column-major, then layer 2 (hidden+1) x 1; n must match */
int  grlx_fqi_get_params(grlx_fqi_ctx *ctx, int replica, double *out, int n);
/* the transition store (FQIPredictor::transitions_) as the kernels hold it: normalised (obs, action) inputs [count][3],
 * next observations [count][2], rewards, and the targets of the last iteration; any pointer may be NULL */
int  grlx_fqi_get_transitions(grlx_fqi_ctx *ctx, int replica, int first, int count, double *in, double *next_obs, double *reward, double *targets);
/* state of a replica after the last batch: transitions stored, L-infinity target change and iteration count of the last
 * rebuild (fqi.cpp:213), mean squared error of its last epoch (ann.cpp:201), RNG streams {global, thread-local} */
int  grlx_fqi_info(grlx_fqi_ctx *ctx, int replica, int64_t *n_transitions, double *maxdelta, int32_t *iterations, double *error, uint64_t rng[2]);

#ifdef __cplusplus
}
#endif
#endif /* GRLX_H_ */
