// grlx_internal.h -- structures shared by the HIP kernels and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/grlx.h"

namespace grlx {

constexpr int kLanesPerReplica = 16;     // one lane per tiling
constexpr size_t kEnvMailBytes = 1024;    // sizeof(EnvMail), grlx_env_server.h
constexpr size_t kEnvMailFlagOffset = 128 + 15 * 8;   // offsetof(EnvMail, stats[15]): served to the end 1 / fell back 2
constexpr size_t kWideMailBytes = 2048;   // >= sizeof(WideMail<ENV>), grlx_env_server_wide.h (the wide kernels' environment server)
constexpr size_t kAcParkBytes = 12 * 64 * 16 + 3 * 64 * 4;     // grlx_rollout_ac_wide.h: kWideQuads quads + three counters per lane
constexpr int kReplicasPerWave = 4;      // per sub-batch; a wide wave carries 4*B (grlx_rollout_wide.h)
constexpr int kMaxTrace = 10;            // replacing trace: (gamma*lambda)^n < 0.01 must hold for n <= 10
constexpr int kMaxActions = GRLX_MAX_ACTIONS;
constexpr uint32_t kInvalidPos = 0xFFFFFFFFu;
constexpr int kMaxProbe = 2048;

// status bits (sticky, per replica)
enum : uint32_t { ST_TABLE_FULL = 1u, ST_DOMAIN = 2u, ST_ROWS_FULL = 4u, ST_TRACE_OVERFLOW = 8u,
                  ST_BAD_POS = 16u };     // a TD update was queued for "no position": internal error, never a silent access

// One slot of a replica's sparse weight table.  The reference's table is a
// dense double[memory] (linear.cpp:86) of which a run touches ~0.2 %; here a
// slot exists only once touched and starts at the value the reference's dense
// initialisation would have given it (lazy_weight()).
struct __attribute__((aligned(16))) Entry {
  uint32_t key;      // reference slot index + 1; 0 = empty
  uint32_t aux;      // reserved (bit mask of tilings that touched the slot)
  double   val;
};

// Persistent per-replica state between launches (one cloned experiment of
// experiment/multi, multi.cpp:49-59).
struct __attribute__((aligned(16))) ReplicaState {
  double   x[GRLX_MAX_STATE];   // environment state (ModeledEnvironment::state_)
  uint64_t G;                   // global srand48 stream              (deployer.cpp:70-74)
  uint64_t TL;                  // thread-local RandGen stream        (utils.h:160-171)
  uint64_t S1, S2;              // samplers' private Rand             (greedy.cpp:38-41)
  uint64_t TL0;                 // TL state right after seeding: base of lazy weight init
  double   eps_decay;           // EpsilonGreedySampler::decay_
  double   ac_decay, ac_noise;  // ActionPolicy::decay_, n_
  int64_t  tt, ss;              // trial counter, cumulative learning steps
  uint64_t test_steps;
  uint32_t n_slots[2];          // occupied table slots
  uint32_t status;
  uint32_t rows;                // test rows written
  // actor-critic: the critic's trace survives episodes and launches (ac.cpp:170-173)
  int32_t  tr_len;
  int32_t  pad0;
  double   tr_total;
  // a loaded policy (ParameterizedRepresentation {action: load}, representation.h:231-263): dense
  // image double[memory] that replaces the drawn initial value of every slot not yet in the
  // sparse table; NULL = the reference's random initialisation.  Shared between replicas.
  const double *lazy_base[2];
  // target network of table 0 (rollout_tgt_kernel): ParameterizedRepresentation::count_ and the synchronisations so far
  int64_t  sync_count;
  uint32_t syncs;
  // ... after a load into it ({action: load}: setParams + synchronize, representation.h:231-263): the target's value of EVERY slot right
  // after that synchronisation (dense, per replica; NULL = none) and the number of synchronisations it already holds
  uint32_t syncs_base;
  const double *target_base;
};

// Per-replica state of the per-step agent entry points (grlx_step.h) between two calls: TDAgent::time_ (td.h) and the
// previous action.  The rest of an agent's state is where the fused kernels keep it (ReplicaState, trace_state).
struct __attribute__((aligned(16))) AgentRep {
  double  time;
  double  action;
  int32_t action_index;
  int32_t pad;
};

struct TileParams {
  int32_t  T, D, memory;
  double   scaling[GRLX_MAX_DIMS];
  int32_t  wrap[GRLX_MAX_DIMS];
};

struct LinearParams {
  double   init_min, init_range, out_min, out_max;
  int32_t  limit;
  uint64_t draws_before;        // TL draws consumed before this table's initialisation
};

struct DevParams {
  int32_t  n_replicas, test_interval, env, agent, trace_kind;
  int32_t  integration_steps;
  double   h;                   // control_step / integration_steps (modeled.cpp:257)
  double   timeout, randomization;
  int32_t  A;
  double   actions[kMaxActions];
  TileParams   tile;
  LinearParams lin;
  double   epsilon, decay_rate, decay_min, alpha, gamma, gl;   // gl = gamma*lambda
  // actor-critic (policy/action + predictor/ac/action); table 0 = critic, table 1 = actor
  TileParams   tile_actor;
  LinearParams lin_actor;
  double   actor_alpha, sigma, theta, ac_decay_rate, ac_decay_min, ac_step_limit;
  int32_t  ac_update_method;
  double   action_min, action_max;
  int32_t  end_stop_penalty, action_penalty;                   // task/cart_pole/swingup
  double   control_step, slope_angle, initial_state_variation, negative_reward, walker_dt;   // compass walker
  uint32_t *trace_state;                                       // [replica][16 lanes][kMaxTrace][2]: pos, cnt | wt << 16
  // sparse tables: table t of replica r starts at tables + ((t*n_replicas + r) << logC)
  Entry   *tables;
  uint32_t logC;
  ReplicaState *states;
  // rows: [row][replica]
  double  *row_reward;
  double  *row_time;            // episode time of the row's trial = sum of tau (= steps: discrete_time)
  int64_t *row_steps, *row_trial;
  int32_t  max_rows;
  // taps
  grlx_tap *taps;
  int32_t  tap_replica, tap_capacity;
  int32_t  tap_starts;          // also record the start pass of every trial (terminal = -1)
  uint32_t *tap_count;
  // diagnostic build only: per-wave cycle sums of 8 phases (NULL = production kernel)
  unsigned long long *diag_out;
  int32_t  no_specialisation;   // tests: force the generic kernel even when a specialised instantiation matches
  int32_t  diag_deferred;       // diagnostics: stamp the deferred-update instantiation (pendulum, 3 actions only)
  double   kappa;               // predictor/critic/advantage: advantage scaling factor
  double   beta;                // predictor/critic/qv: state-value learning rate
  // experiment/online_learning:steps (online_learning.cpp:154): a replica starts no further trial once its learning steps of the run
  // (ReplicaState::ss) have reached this budget; 0 = none.  Honoured by rollout_kernel, rollout_wide_kernel and the actor-critic kernels.
  uint64_t steps_budget;
  uint32_t env_tune;            // experiments (GRLX_ENV_SERVER_TUNE): bits 0-1 s_setprio of the rollout wave, 2-3 of the server wave, 4 no prefetch;
                                // tests: bit 6 the server leaves at once (every replica falls back to integrating itself)
  void    *park;                // rotating actor-critic kernel (12 slots): lane state of the third sub-batch, kAcParkBytes per wave
  struct EnvMail *env_mail;     // mailboxes of the environment server ([replica], grlx_env_server.h); null = the rollout kernel integrates itself
  int32_t  test_trials;         // experiment/online_learning:test_trials (>= 1): greedy episodes per test trial, averaged in the row
  // Actor-critic with EQUAL tile codings for actor and critic (cfg/cart_pole/ac_tc.yaml: the critic copies resolution and memory):
  // both tables are looked up with the same slots at every step, so the two sparse tables are kept as TWINS -- the same slot at the
  // same position in both, created together, re-hashed together -- and one key resolution / one creation path serves both.
  int32_t  twin_tables;
  int32_t  tile_safe;           // projector/tile_coding:safe: 1 = claim table, single projections claim; 2 = batch projections claim too (plain kernel)
  int32_t  target_interval;     // > 0: the Q table has a target network synchronised every so many update() calls
  double   target_tau;          // synchronisation strength (representation.h:284-296)
  double  *tvals;               // [replica][2^logC]: the target network's value per table position (all ones: not materialised)
  int32_t  tap_deferred;        // taps are recorded by the deferred-update (production) ordering instead of the in-place one
  int32_t  replicas_per_wave;   // 4 (one sub-batch) or 8 (two): chosen at create from the replica count and the SIMD count
  int32_t  wave_limit;          // wide actor-critic kernel: waves launched at most (> 0); further replicas come from `queue`
  uint32_t *queue;              // next unstarted replica (set by the launcher before every launch)
  // per-step agent entry points (grlx_step.h): created at the first grlx_agent_* call of a context
  AgentRep *agent_rep;          // [replica]
  uint32_t *agent_lane;         // [replica][16 lanes][2]: reference slots of project(prev_obs, prev_action) / critic's and actor's project(prev_obs)
};

// ---------------------------------------------------------------------------
// launchers implemented in grlx_kernels.hip
// *variant (optional) receives the GRLX_KERNEL_* instantiation that was launched
hipError_t launch_rollout(const DevParams &P, int n_trials, hipStream_t stream, int *variant);
bool       env_server_serves(const DevParams &P);                                   // is this context's rollout kernel one the environment server works for?
size_t     env_server_mail_bytes(const DevParams &P);                               // ... and the size of a replica's mailbox there (0: not served)
hipError_t launch_env_server(const DevParams &P, hipStream_t stream);
hipError_t launch_rollout_ac(const DevParams &P, int n_trials, hipStream_t stream, int *variant);
hipError_t launch_rollout_qv(const DevParams &P, int n_trials, hipStream_t stream, int *variant);
hipError_t launch_rollout_acc(const DevParams &P, int n_trials, hipStream_t stream, int *variant);
hipError_t launch_rollout_tgt(const DevParams &P, int n_trials, hipStream_t stream, int *variant);
// diagnostic: overwrite the whole vector register file (512 registers per lane) and the user SGPRs of every SIMD with
// `pattern`, so that a kernel launched next on `stream` that reads a register it never wrote computes with that
// pattern instead of with whatever the previous wave left (GRLX_POISON_REGISTERS)
hipError_t launch_poison_registers(uint32_t pattern, hipStream_t stream);
hipError_t launch_get_target_weights(const DevParams &P, int replica, const uint32_t *slots_dev, int n, double *out_dev, hipStream_t stream);
// per-step plug-in interfaces (grlx_step.h); the pointers of StepArgs are device pointers
enum : int { STEP_START = 0, STEP_STEP = 1, STEP_END = 2 };

struct StepArgs {
  int32_t        mode;        // STEP_*
  int32_t        test;        // 0: the learning agent (agent/td) / a learning trial's start state; 1: the test agent (agent/fixed) / a test start
  const int32_t *active;      // [n_replicas] or null: only replicas with a non-zero entry take part in the call
  const double  *obs;         // [n_replicas][D]   agent calls: the observation handed to the agent
  const double  *reward;      // [n_replicas]      agent step / end
  const int32_t *terminal;    // [n_replicas] or null; agent step: a replica whose entry is 2 gets Agent::end instead (td.cpp:76-81)
  double        *action;      // [n_replicas]      agent start / step: the action taken (untouched for replicas that ended or are inactive)
};

hipError_t launch_agent_step(const DevParams &P, const StepArgs &A, hipStream_t stream);
hipError_t launch_env_start(const DevParams &P, int test, const int32_t *active_dev, double *obs_dev, hipStream_t stream);
hipError_t launch_env_advance(const DevParams &P, const int32_t *active_dev, const double *action_dev, double *obs_dev, double *reward_dev,
                              int32_t *terminal_dev, hipStream_t stream);
hipError_t launch_project(const TileParams &tp, const double *in_dev, int n, uint32_t *out_dev, hipStream_t stream);
hipError_t launch_env_step(const DevParams &P, double *state_dev, const double *action_dev, int n,
                           double *obs_dev, double *reward_dev, int32_t *terminal_dev, uint32_t *err_dev, hipStream_t stream);
hipError_t launch_table_op(const DevParams &P, int table, int op, const int32_t *replica_dev, const uint32_t *idx_dev, int n,
                           const double *arg_dev, double alpha, double *out_dev, hipStream_t stream);
hipError_t launch_set_lazy_base(const DevParams &P, int table, int first, int count, const double *image_dev, hipStream_t stream);
hipError_t launch_target_after_load(const DevParams &P, int replica, const double *image_dev, double *out_dev, hipStream_t stream);
hipError_t launch_export_weights(const DevParams &P, int table, int replica, double *out_dev, hipStream_t stream);
hipError_t launch_get_weights(const DevParams &P, int table, int replica, const uint32_t *slots_dev, int n, double *out_dev, hipStream_t stream);
hipError_t launch_math(int op, const double *x, const double *y, int n, double *out, hipStream_t stream);
hipError_t launch_rand48_at(uint64_t x0, const uint64_t *skip, int n, double *out, hipStream_t stream);
hipError_t launch_curve_stats(const DevParams &P, int first, int count, double *out_dev, hipStream_t stream);
hipError_t launch_reload_entries(const DevParams &P, int table, int first_replica, int n_replicas, const double *image_dev, hipStream_t stream);
hipError_t launch_step_counts(const DevParams &P, uint64_t *out_dev /*[3]: learn, test, status-or*/, hipStream_t stream);
// sparse-table growth (grlx_api.cpp: grow_tables).  max over replicas and tables of the occupied slots; every entry of the
// old tables re-inserted into tables of 2^new_logC entries per replica (remap_dev, optional: [replica][2^old logC] new position of
// every old position of table 0); the positions persisted outside the tables translated (actor-critic trace, target values)
hipError_t launch_max_load(const DevParams &P, int n_tables, uint32_t *out_dev, hipStream_t stream);
hipError_t launch_rehash(const DevParams &P, int n_tables, Entry *new_tables, uint32_t new_logC, uint32_t *remap_dev, hipStream_t stream);
hipError_t launch_remap_positions(const DevParams &P, const uint32_t *remap_dev, uint32_t new_logC, double *new_tvals, hipStream_t stream);

} // namespace grlx
