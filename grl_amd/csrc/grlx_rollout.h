// grlx_rollout.h -- the fused rollout kernel of the discrete-action TD agents (SARSA, Q, Expected SARSA, advantage learning):
// diagnostic stamps, the compile-time specialisation of cfg/pendulum/{sarsa,q}_tc.yaml and rollout_kernel.
// Part of the single translation unit grlx_kernels.hip (included there, in order; not self-contained).
#pragma once

namespace grlx {

// in-kernel stamps (diagnostic instantiation only; cdna_hip_programming.md section 7)
__device__ __forceinline__ unsigned long long stamp()
{
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
// (GRLX_ENV_SERVER_STATS builds: the served kernel stamps too, and leaves its sums in its mailboxes -- tools/env_server_stats.py)
#ifdef GRLX_ENV_SERVER_STATS
#define GRLX_STAMPED (DIAG || SERVED)
#else
#define GRLX_STAMPED DIAG
#endif
#define DIAG_STAMP(slot)                                   \
  if (GRLX_STAMPED)                                        \
  {                                                        \
    unsigned long long now__ = stamp();                    \
    diag_sum[slot] += now__ - diag_last;                   \
    diag_last = now__;                                     \
  }

// Compile-time specialisation for the headline configuration (cfg/pendulum/sarsa_tc.yaml,
// tests/pendulum-sarsa-tc.yaml): the values below replace the corresponding fields of the
// parameter block, so they become literals (fewer live SGPRs, "% memory" becomes a mask, the
// wrap modulus a constant).  The launcher selects it only when every one of these runtime
// parameters equals the constant bit for bit, so results cannot differ from the generic kernel.
constexpr DevParams make_spec_pendulum_tc()
{
  DevParams P{};
  P.env = GRLX_ENV_PENDULUM;
  P.agent = GRLX_AGENT_SARSA;
  P.trace_kind = GRLX_TRACE_REPLACING;
  P.test_interval = 10;
  P.h = 0.03 / 5;
  P.integration_steps = 5;
  P.timeout = 2.99;
  P.randomization = 0;
  P.A = 3;
  P.actions[0] = -3; P.actions[1] = 0; P.actions[2] = 3;
  P.tile.T = 16; P.tile.D = 3; P.tile.memory = 8388608;
  P.tile.scaling[0] = 16 / 0.31415; P.tile.scaling[1] = 16 / 3.1415; P.tile.scaling[2] = 16 / 3.0;
  P.tile.wrap[0] = 320;
  P.lin.init_min = 0; P.lin.init_range = 1;
  P.lin.out_min = -1.7976931348623157e308; P.lin.out_max = 1.7976931348623157e308;
  P.lin.limit = 1; P.lin.draws_before = 0;
  P.epsilon = 0.05; P.decay_rate = 1; P.decay_min = 0;
  P.alpha = 0.2; P.gamma = 0.97; P.gl = 0.97 * 0.65;
  return P;
}
__device__ const DevParams d_spec_pendulum_tc = make_spec_pendulum_tc();

// AGENT: the predictor kind is a compile-time constant of the instantiation too (one per TD agent)
template <int AGENT>
struct SpecPendulumTcA {
  __device__ static __forceinline__ int agent(const DevParams &) { return AGENT; }
  // every numeric field the rollout kernel reads must equal the constant, bit for bit
  static bool matches(const DevParams &P)
  {
    constexpr DevParams C = make_spec_pendulum_tc();
    bool ok = P.env == C.env && P.trace_kind == C.trace_kind && P.test_interval == C.test_interval && P.h == C.h &&
              P.integration_steps == C.integration_steps && P.timeout == C.timeout && P.randomization == C.randomization && P.A == C.A &&
              P.tile.T == C.tile.T && P.tile.D == C.tile.D && P.tile.memory == C.tile.memory &&
              P.lin.init_min == C.lin.init_min && P.lin.init_range == C.lin.init_range && P.lin.out_min == C.lin.out_min &&
              P.lin.out_max == C.lin.out_max && P.lin.limit == C.lin.limit && P.lin.draws_before == C.lin.draws_before &&
              P.epsilon == C.epsilon && P.decay_rate == C.decay_rate && P.decay_min == C.decay_min && P.alpha == C.alpha &&
              P.gamma == C.gamma && P.gl == C.gl && P.agent == AGENT;
    for (int i = 0; i < 3; ++i)
      ok = ok && P.actions[i] == C.actions[i] && P.tile.scaling[i] == C.tile.scaling[i] && P.tile.wrap[i] == C.tile.wrap[i];
    return ok;
  }
  __device__ static __forceinline__ const DevParams &numeric(const DevParams &) { return d_spec_pendulum_tc; }
};
using SpecPendulumTc = SpecPendulumTcA<GRLX_AGENT_SARSA>;

// Every numeric field of the parameter block (what a kernel reads through SPEC::numeric) equal, bit for bit.  Pointers,
// sizes, taps and layout choices stay run-time values of P.
inline bool spec_numeric_equal(const DevParams &a, const DevParams &b)
{
  auto tile_eq = [](const TileParams &x, const TileParams &y) {
    bool ok = x.T == y.T && x.D == y.D && x.memory == y.memory;
    for (int i = 0; i < GRLX_MAX_DIMS; ++i) ok = ok && x.scaling[i] == y.scaling[i] && x.wrap[i] == y.wrap[i];
    return ok;
  };
  auto lin_eq = [](const LinearParams &x, const LinearParams &y) {
    return x.init_min == y.init_min && x.init_range == y.init_range && x.out_min == y.out_min && x.out_max == y.out_max &&
           x.limit == y.limit && x.draws_before == y.draws_before;
  };
  bool ok = a.test_interval == b.test_interval && a.env == b.env && a.agent == b.agent && a.trace_kind == b.trace_kind &&
            a.integration_steps == b.integration_steps && a.h == b.h && a.timeout == b.timeout && a.randomization == b.randomization &&
            a.A == b.A && tile_eq(a.tile, b.tile) && lin_eq(a.lin, b.lin) && a.epsilon == b.epsilon && a.decay_rate == b.decay_rate &&
            a.decay_min == b.decay_min && a.alpha == b.alpha && a.gamma == b.gamma && a.gl == b.gl &&
            tile_eq(a.tile_actor, b.tile_actor) && lin_eq(a.lin_actor, b.lin_actor) && a.actor_alpha == b.actor_alpha &&
            a.sigma == b.sigma && a.theta == b.theta && a.ac_decay_rate == b.ac_decay_rate && a.ac_decay_min == b.ac_decay_min &&
            a.ac_step_limit == b.ac_step_limit && a.ac_update_method == b.ac_update_method && a.action_min == b.action_min &&
            a.action_max == b.action_max && a.end_stop_penalty == b.end_stop_penalty && a.action_penalty == b.action_penalty &&
            a.control_step == b.control_step && a.slope_angle == b.slope_angle &&
            a.initial_state_variation == b.initial_state_variation && a.negative_reward == b.negative_reward &&
            a.walker_dt == b.walker_dt && a.kappa == b.kappa && a.beta == b.beta && a.tile_safe == b.tile_safe &&
            a.target_interval == b.target_interval && a.target_tau == b.target_tau;
  for (int i = 0; i < kMaxActions; ++i) ok = ok && a.actions[i] == b.actions[i];
  return ok;
}

// The agent block of the reference's Q-learning yamls (cfg/pendulum/q_tc.yaml, cfg/compass_walker/qlearning_walk.yaml):
// epsilon-greedy 0.05, alpha 0.2, gamma 0.97, lambda 0.65, replacing trace, test_interval 10, weights U(0, 1), no output limits
constexpr void spec_q_agent_block(DevParams &P)
{
  P.agent = GRLX_AGENT_Q;
  P.trace_kind = GRLX_TRACE_REPLACING;
  P.test_interval = 10;
  P.randomization = 0;
  P.A = 3;
  P.tile.T = 16; P.tile.memory = 8388608;
  P.lin.init_min = 0; P.lin.init_range = 1;
  P.lin.out_min = -1.7976931348623157e308; P.lin.out_max = 1.7976931348623157e308;
  P.lin.limit = 1; P.lin.draws_before = 0;
  P.epsilon = 0.05; P.decay_rate = 1; P.decay_min = 0;
  P.alpha = 0.2; P.gamma = 0.97; P.gl = 0.97 * 0.65;
  // fields these kernels do not read, at the values grlx_config_pendulum_sarsa leaves them (the comparison covers every field)
  P.end_stop_penalty = 1;
  P.slope_angle = 0.004; P.initial_state_variation = 0.2; P.negative_reward = -100.0;
}
// cfg/compass_walker/qlearning_walk.yaml (BASELINE config 4): every derived value by the expression of make_params (grlx_api.cpp)
constexpr DevParams make_spec_walker_q()
{
  DevParams P{};
  spec_q_agent_block(P);
  P.env = GRLX_ENV_COMPASS_WALKER;
  P.integration_steps = 20;
  P.control_step = 0.2;
  P.h = 0.2 / (double)(size_t)20;
  P.timeout = 100.0;
  P.slope_angle = 0.004; P.initial_state_variation = 0.2; P.negative_reward = -100.0;
  P.walker_dt = 1.0E-6 * (double)(uint64_t)200000 / 20;          // floor((0.2 + 0.5E-6) * 1E6) = 200000 whole microseconds
  P.action_min = -1.2; P.action_max = 1.2;
  P.actions[0] = -1.2 + (2.4 / 2.) * 0; P.actions[1] = -1.2 + (2.4 / 2.) * 1; P.actions[2] = -1.2 + (2.4 / 2.) * 2;
  P.tile.D = 6;
  P.tile.scaling[0] = 16 / 0.0838; P.tile.scaling[1] = 16 / 0.1047; P.tile.scaling[2] = 16 / 0.1111;
  P.tile.scaling[3] = 16 / 0.2222; P.tile.scaling[4] = 16 / 10.; P.tile.scaling[5] = 16 / 1.2;
  return P;
}
// the acrobot of SURVEY 8d config 4 (the reference ships no TD yaml for it): dynamics/acrobot + task/acrobot/balancing under the
// same agent block, 3 torques over [-1, 1], control step 0.05 s, tile resolution [0.05, 0.05, 0.2, 0.4 | 1.0]
constexpr DevParams make_spec_acrobot_q()
{
  DevParams P{};
  spec_q_agent_block(P);
  P.env = GRLX_ENV_ACROBOT;
  P.integration_steps = 5;
  P.control_step = 0.05;
  P.h = 0.05 / (double)(size_t)5;
  P.timeout = 20.0;
  P.walker_dt = 1.0E-6 * (double)(uint64_t)50000 / 5;
  P.action_min = -1.0; P.action_max = 1.0;
  P.actions[0] = -1.0 + (2.0 / 2.) * 0; P.actions[1] = -1.0 + (2.0 / 2.) * 1; P.actions[2] = -1.0 + (2.0 / 2.) * 2;
  P.tile.D = 5;
  P.tile.scaling[0] = 16 / 0.05; P.tile.scaling[1] = 16 / 0.05; P.tile.scaling[2] = 16 / 0.2; P.tile.scaling[3] = 16 / 0.4;
  P.tile.scaling[4] = 16 / 1.0;
  return P;
}
__device__ const DevParams d_spec_walker_q = make_spec_walker_q();
__device__ const DevParams d_spec_acrobot_q = make_spec_acrobot_q();
struct SpecWalkerQ {
  static constexpr int kEnv = GRLX_ENV_COMPASS_WALKER;
  __device__ static __forceinline__ int agent(const DevParams &) { return GRLX_AGENT_Q; }
  static bool matches(const DevParams &P) { constexpr DevParams C = make_spec_walker_q(); return spec_numeric_equal(P, C); }
  __device__ static __forceinline__ const DevParams &numeric(const DevParams &) { return d_spec_walker_q; }
};
struct SpecAcrobotQ {
  static constexpr int kEnv = GRLX_ENV_ACROBOT;
  __device__ static __forceinline__ int agent(const DevParams &) { return GRLX_AGENT_Q; }
  static bool matches(const DevParams &P) { constexpr DevParams C = make_spec_acrobot_q(); return spec_numeric_equal(P, C); }
  __device__ static __forceinline__ const DevParams &numeric(const DevParams &) { return d_spec_acrobot_q; }
};
struct SpecNone {
  __device__ static __forceinline__ const DevParams &numeric(const DevParams &P) { return P; }
  __device__ static __forceinline__ int agent(const DevParams &P) { return P.agent; }
};

// ADV: advantage learning (predictor/critic/advantage, advantage.cpp:222-268) also reads A(s, .) of the
// PREVIOUS state for every action with the current weights: NA more rows (their table positions are
// those of the previous pass), Q(s,a) being one of them.  Built without the deferred update.
// TAP: per-step records of the tapped replica.  Normally the instantiation that updates in place; TAP with DEFER
// records the PRODUCTION ordering step by step (grlx_config.tap_deferred), the trace length being the one the pending
// update will leave.
// GRLX_TWO_WAVES (experiment): limit the kernel to 256 registers so that two waves share a SIMD
#ifdef GRLX_TWO_WAVES
#define GRLX_ROLLOUT_OCCUPANCY __attribute__((amdgpu_waves_per_eu(2, 2)))
#else
#define GRLX_ROLLOUT_OCCUPANCY
#endif
// SERVED: the environment steps come from the environment server (grlx_env_server.h) -- its own kernel, rollout_served_kernel below.
template <int ENV, int NA, bool DIAG, typename SPEC, bool DEFER, bool ADV, bool TAP, bool SERVED>
__device__ __forceinline__ void rollout_body(const DevParams &P, int n_trials)
{
  static_assert(!(ADV && DEFER), "the advantage-learning instantiation updates in place");
  constexpr int NROWS = ADV ? 2 * NA : NA + 1;      // LDS rows of weights summed per pass
  // N: the numeric parameters -- the runtime block, or compile-time constants in a specialised build.
  // P keeps the pointers, the replica count and the buffer sizes.
  const DevParams &N = SPEC::numeric(P);
  constexpr int S = Env<ENV>::S, D = Env<ENV>::D, T = kLanesPerReplica;
  __shared__ double   sh_w[NROWS * 16 * 4];
  __shared__ uint32_t sh_ppos[4 * 16];
  __shared__ double   sh_fb[16 * 4];
  __shared__ uint32_t sh_fbflag[16 * 4];
  __shared__ uint32_t sh_mb[4 * NA * 16];      // parallel insert: claimed bucket per (action, tiling), ~0 = none
  __shared__ uint32_t sh_ms[4 * NA * 16];      //                  and the slot claiming it
  __shared__ uint32_t sh_mail[4];              // position of a slot that just became shared between tilings
  __shared__ double   sh_res[4 * 16];          // per-replica sums (row r in slot r)
  __shared__ uint64_t sh_jump[2048];           // LCG jump table (lazy weight initialisation)
  jump_table_to_lds(sh_jump);

  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, j = lane & 15;
  const int r_raw = blockIdx.x * kReplicasPerWave + g;
  const bool live = r_raw < P.n_replicas;
  const int r = live ? r_raw : 0;
  const bool tapped = live && (r == P.tap_replica);
  const unsigned long long gmask = 0xFFFFull << (16 * g);

  ReplicaState &RS = P.states[r];
  double x[S];
#pragma unroll
  for (int i = 0; i < S; ++i) x[i] = RS.x[i];
  uint64_t G = RS.G, TL = RS.TL, S1 = RS.S1;
  double eps_decay = RS.eps_decay;
  int64_t tt = RS.tt, ss = RS.ss;
  uint64_t test_steps = RS.test_steps;
  uint32_t status = RS.status, rows = RS.rows, inserted = 0;

  const Table tab = table_of(P, 0, r);
  // the two fields the slot-creation path reads (start of the initialisation stream, loaded policy image): kept in registers
  // instead of loaded behind a miss -- in the first few hundred trials most passes of a wave create a slot somewhere
  ReplicaState RSc;
  RSc.TL0 = RS.TL0;
  RSc.lazy_base[0] = RS.lazy_base[0];
  UpdateParams up;
  up.out_min = N.lin.out_min;
  up.out_max = N.lin.out_max;
  up.limit = N.lin.limit != 0;
  up.ee = N.gl;                                 // pow(gamma*lambda, tau), tau = 1 (discrete_time)
  up.cut = (N.trace_kind == GRLX_TRACE_REPLACING) ? 0.01 : 0.0001;
  up.use_trace = N.trace_kind == GRLX_TRACE_REPLACING;
  up.dW = up.dT = 0;

  double acts[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) acts[a] = N.actions[a];
  // The action coordinate of tiling j and the tiling index itself do not change: their murmur
  // key words are computed once (32-bit multiplies are quarter rate).
  uint32_t key_act[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a)
    key_act[a] = in_reg(murmur_key(tile_coord<T>(N.tile, D, tile_quant(N.tile, D, N.actions[a]), j)));
  const uint32_t key_j = in_reg(murmur_key(j));

  // the 16 lanes of a replica share out the independent sin/cos evaluations of its equations of motion (grlx_envs.h)
  LaneShare lshare;
  lshare.src[0] = g * 16; lshare.src[1] = g * 16 + 1; lshare.src[2] = g * 16 + 2;
  lshare.role3 = j % 3; lshare.role2 = j & 1;

  // the environment server (grlx_env_server.h): this replica's mailbox, the number of commands sent, and whether the server still answers
  constexpr bool EXT = SERVED;
  static_assert(!SERVED || (DEFER && !DIAG && !TAP && !ADV && ENV == GRLX_ENV_PENDULUM && NA == 3), "what the environment server works for");
  EnvMail *mail = nullptr;
  unsigned long long mseq = 0;
  bool srv = false;
#ifdef GRLX_ENV_SERVER_STATS
  unsigned long long st_wait = 0, st_polls = 0, st_fetch = 0, st_begin = mail_clock();
#endif
  MailBox mbox;
  mail_u32x4 mpre = {0u, 0u, 0u, 0u};      // the unit this lane loaded ahead
  if constexpr (EXT)
  {
    srv = live && P.env_mail != nullptr;
    mail = P.env_mail + r;
    mbox = mailbox_of(P, r, j);
    mail_setprio(P.env_tune & 3u);
  }

  TraceRegs tr;
  trace_init(tr);
  int tr_len_ref = 0;           // length as the reference reports it (its trace survives test trials)
  // DEFER: the TD update of a step is applied one pass later, between the next step's table loads
  // and their first use (same arithmetic, same order of updates; only its position in the
  // instruction stream moves).  The diagnostic instantiation (stamps, taps) updates in place
  // (DEFER = false) unless asked to stamp the production ordering.
  bool pd = false, pd_sh = false;
  double pd_dW = 0, pd_dT = 0, pd_wp = 0;
  uint32_t pd_pos = kInvalidPos;
  unsigned long long diag_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, diag_last = 0;
  if (GRLX_STAMPED) diag_last = stamp();

  for (int trial = 0; trial < n_trials; ++trial)
  {
    // online_learning.cpp:154: a replica whose learning steps have reached the steps budget starts no further trial
    const bool act = live && !(P.steps_budget != 0u && (uint64_t)ss >= P.steps_budget);
    if (!__any(act)) break;
    const int ti = N.test_interval;
    const int test = (ti >= 0 && tt % (ti + 1) == ti) ? 1 : 0;        // online_learning.cpp:160
    // a test trial is test_trials greedy episodes (online_learning.cpp:161-170): each starts the environment and the agent anew, while
    // reward and time keep adding up (:202-203); a learning trial is one episode.  `time` doubles as the agent's episode time, which
    // only learning episodes read (the sampler's decay at time 0).
    double total_reward = 0, time = 0;
    const int subtrials = (test && P.test_trials > 1) ? P.test_trials : 1;
    for (int st = 0; st < P.test_trials; ++st)
    {
    const bool episode = act && st < subtrials;
    if (!__any(episode)) break;
    double obs[D], reward = 0;
    int terminal = 0;
    bool running = episode;

    // environment_->start (modeled.cpp:132-158)
    if (episode)
    {
      Env<ENV>::start(N, test, TL, G, x);
      Env<ENV>::observe(N, x, obs);
      if constexpr (EXT)
        if (srv)
        { // the server starts on the first step of this episode, for every action, while this wave looks up Q(s0, .)
          ++mseq;
          if (j == 0) mail_send_reset(mail, mseq, x);
        }
    }
    // agent->start: TDAgent::start clears the trace (td.cpp:50-61, sarsa.cpp:126-132); the
    // trace was written back at the end of the previous learning trial, so it is empty here
    double action = 0;
    int    action_index = 0;
    uint32_t p_pos = kInvalidPos, p_slot = 0;
    bool   p_sh = false;
    double wp_seen = 0;                   // weight of p's slot as looked up (and forwarded from the trace) one pass ago
    uint32_t pos_prev[NA];                // ADV: positions of project(s, a_k) for every action
#pragma unroll
    for (int a = 0; a < NA; ++a) pos_prev[a] = 0u;
    if (!test) tr_len_ref = 0;          // TDAgent::start -> trace_->clear()
    bool first = true;                    // first pass = start(): act only, no env step / update

    for (;;)
    {
      if (!__any(running || pd)) break;
      // state that lives across the deferred-update site
      uint32_t slot[NA];
      Lookup lk[NA];
      BucketRegs br[NA];
      double wp = 0;
      double wprev[NA];
#pragma unroll
      for (int a = 0; a < NA; ++a) wprev[a] = 0;
      bool has_next = false, update = false;
      // (slot, lk, br are written and read only under running && has_next)
      if (running)
      {
        DIAG_STAMP(0)
        // -------- environment step (skipped on the start() pass)
        if (!first)
        {
          bool stepped = false;
          if constexpr (EXT)
            if (srv)
            { // the step was integrated by the server while this wave updated its table: the candidate of the action taken
#ifdef GRLX_ENV_SERVER_STATS
              const unsigned long long tf0 = mail_clock();
              stepped = mail_take<ENV>(N, mbox, mpre, mseq - 1, action_index, g, gmask, x, obs, reward, terminal, status, &st_polls);
              st_wait += mail_clock() - tf0;
              ++st_fetch;
#else
              stepped = mail_take<ENV>(N, mbox, mpre, mseq - 1, action_index, g, gmask, x, obs, reward, terminal, status);
#endif
              if (!stepped)
              { // no answer: integrate here, from now on (the server is told to stop waiting for this replica)
                srv = false;
                ++mseq;
                if (j == 0) mail_send(mail, mseq, kMailExit);
              }
            }
          if (!stepped)
            env_step<ENV, true, LaneShare>(N, x, action, obs, reward, terminal, status, lshare);   // online_learning.cpp:196
          total_reward += reward;                                          // :202
          time += 1;                                                       // tau = 1
        }
        has_next = first || !Env<ENV>::kAbsorbing || terminal != 2;      // (a compile-time `true` for the pendulum)
        update = !first && !test;                                          // a TD update follows
        DIAG_STAMP(1)

        // -------- policy: Q(s', .) for all actions (q.cpp:94-107): projections
        if (has_next)
        {
          uint32_t hpre = 449u ^ (uint32_t)(D + 2);
#pragma unroll
          for (int i = 0; i < D; ++i)
            hpre = murmur_mix(hpre, tile_coord<T>(N.tile, i, tile_quant(N.tile, i, obs[i]), j));
          const uint32_t hpm = hpre * 0x5bd1e995u;                           // shared by the NA projections
#pragma unroll
          for (int a = 0; a < NA; ++a)
          {
            uint32_t h = hpm ^ key_act[a];                                   // murmur_mix(hpre, coordinate of action a)
            h = murmur_absorb(h, key_j);                                     // murmur_mix(h, j)
            const uint32_t hm = murmur_final(h), mem = (uint32_t)N.tile.memory;
            slot[a] = ((mem & (mem - 1u)) == 0u) ? (hm & (mem - 1u)) : (hm % mem);
          }
        }
        DIAG_STAMP(2)
        // every store of the previous step precedes these loads in program order (issued a
        // full RK4 ago, so this wait is free; it makes the ordering explicit)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        DIAG_STAMP(6)
        // weights of project(s, a): with the deferred ordering the value this lane saw when it looked the slot up one pass
        // ago is still the table's (anything newer lives in the trace, is the held eviction, or belongs to a slot shared
        // between tilings -- all reconciled below); the in-place ordering may have evicted it meanwhile and loads it
        if (update) wp = DEFER ? wp_seen : value_load(tab, p_pos);
        if (ADV && update)
        {
#pragma unroll
          for (int a = 0; a < NA; ++a) wprev[a] = value_load(tab, pos_prev[a]);
        }
        if (has_next) table_issue<NA>(tab, slot, lk, br);                  // home buckets of Q(s', .): loads in flight
      }

      // -------- the PREVIOUS step's predictor update, in the shadow of the loads just issued.
      // It works on the register trace only; the one weight it evicts is handed back in `ev` and
      // stored at the end of this pass, so no store sits between the loads and their use.
      Evicted ev;
      ev.n = 0u; ev.pos = kInvalidPos; ev.val = 0;
      if (DEFER)
      {
        DIAG_STAMP(7)
        if (pd)
        {
          sh_ppos[g * 16 + j] = pd_pos;
          sh_fbflag[j * 4 + g] = 0u;
        }
        wave_sync();
        if (pd)
        {
          up.dW = pd_dW;
          up.dT = pd_dT;
          td_update_lane<true>(tr, tab, up, pd_pos, pd_sh, pd_wp, g, j, sh_ppos, sh_fb, sh_fbflag, status, ev);
          pd = false;
        }
        DIAG_STAMP(5)
      }

      if (running)
      {
        double q[NA];
        uint32_t pos[NA];
        double w[NA];
        bool sh[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) { q[a] = 0; pos[a] = kInvalidPos; w[a] = 0; sh[a] = false; }
        if (has_next)
        {
          bool shared_event = false;
          table_get_finish<NA>(tab, N.lin, RSc, 0, slot, lk, br, pos, w, sh, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump, status, inserted,
                               [&](uint32_t mp) {
                                 // a weight evicted a moment ago and not stored yet: store it now, the finder reads it
                                 if (DEFER && ev.pos != kInvalidPos && ev.pos == mp) value_store(tab, mp, ev.val);
                                 trace_share_event(tr, tab, mp);
                                 if (p_pos == mp) p_sh = true;
                                 shared_event = true;
                               });
          DIAG_STAMP(7)
          if (rarely(__any(shared_event)) && update)
          {
            wp = value_load(tab, p_pos);
            if (ADV)
            {
#pragma unroll
              for (int a = 0; a < NA; ++a) wprev[a] = value_load(tab, pos_prev[a]);
            }
          }
        }
        if (DEFER)
        { // Values loaded before the deferred update may be stale where that update wrote the table:
          // (1) slots shared between tilings (kept current in the table by their owners' lanes) and
          // paths that do not track their write-backs: load again, the stores precede these loads;
          // (2) the one held eviction: its value is in `ev`.
          bool risky = ev.n > 1u || (update && p_sh);
#pragma unroll
          for (int a = 0; a < NA; ++a) risky = risky || (has_next && sh[a]);
          if (rarely(__any(risky)))
          {
#pragma unroll
            for (int a = 0; a < NA; ++a)
              if (has_next) w[a] = value_load(tab, pos[a]);
            if (update) wp = value_load(tab, p_pos);
          }
          const bool held = ev.pos != kInvalidPos;
#pragma unroll
          for (int a = 0; a < NA; ++a) w[a] = (held && pos[a] == ev.pos) ? ev.val : w[a];
          wp = (held && p_pos == ev.pos) ? ev.val : wp;
        }
        if (has_next)
        {
#pragma unroll
          for (int a = 0; a < NA; ++a)
          {
            w[a] = trace_forward(tr, pos[a], w[a]);
            SHW(a, j, g) = w[a];
          }
        }
        if (update)
        {
          wp = trace_forward(tr, p_pos, wp);
          if (ADV)
          {
#pragma unroll
            for (int a = 0; a < NA; ++a)
            {
              wprev[a] = trace_forward(tr, pos_prev[a], wprev[a]);
              SHW(NA + a, j, g) = wprev[a];
            }
          }
          else
            SHW(NA, j, g) = wp;
        }
        DIAG_STAMP(3)
        if (!DEFER)
        {
          sh_ppos[g * 16 + j] = p_pos;
          sh_fbflag[j * 4 + g] = 0u;
        }
        wave_sync();
        // LinearRepresentation::read (linear.cpp:136-184): serial sum over the 16 tilings, mean, clamp.
        // Lane r of the replica sums row r (Q(s',a_r) for r < NA, Q(s,a) for r = NA) in the reference's
        // order; the NA+1 results are shared through LDS (lanes beyond NA repeat row 0, harmlessly).
        {
          const int row = (j < NROWS) ? j : 0;
          double sum = 0;
#pragma unroll
          for (int k = 0; k < 16; ++k) sum += SHW(row, k, g);
          sum /= 16;
          sh_res[g * 16 + j] = sum;
        }
        wave_sync();
        if (has_next)
        {
#pragma unroll
          for (int a = 0; a < NA; ++a) q[a] = clampd(sh_res[g * 16 + a], up.out_min, up.out_max);
        }
        double qsa = 0;
        double qprev[NA];                                  // ADV: A(s, a_k) with the current weights
#pragma unroll
        for (int a = 0; a < NA; ++a) qprev[a] = 0;
        if (update)
        {
          if (ADV)
          {
#pragma unroll
            for (int a = 0; a < NA; ++a) qprev[a] = clampd(sh_res[g * 16 + NA + a], up.out_min, up.out_max);
            qsa = pick<double, NA>(qprev, action_index);          // project(s, a) is the row of the action taken
          }
          else
            qsa = clampd(sh_res[g * 16 + NA], up.out_min, up.out_max);
        }

        // -------- sampler (greedy.cpp:63-86, 144-218)
        int a_next = 0;
        int mai = 0, man = 1;
        double best = 0;
        if (has_next)
        {
          findmax<NA>(q, mai, man, best);
          if (test)
          {
            a_next = (man > 1) ? tie_break<NA>(q, best, man, G) : mai;
          }
          else
          {
            if (time == 0.) eps_decay = fmax(eps_decay * N.decay_rate, N.decay_min);
            S1 = lcg_next(S1);
            double rnd = lcg_double(S1);
            if (rnd < eps_decay * N.epsilon)
            {
              G = lcg_next(G);
              a_next = (int)(lcg_long(G) % (uint32_t)NA);
            }
            else
              a_next = (man > 1) ? tie_break<NA>(q, best, man, G) : mai;
          }
        }
        if constexpr (EXT)
          if (srv && running && has_next && !(!first && terminal))
          { // the action of the step this pass looked ahead to: the server moves on to the step after it
            ++mseq;
            if (j == 0) mail_send(mail, mseq, (unsigned)a_next);
            if (!(P.env_tune & 16u))
              mpre = mail_prefetch(mbox, mseq - 1, a_next);        // what the next pass starts from: ready unless the server is late
          }

        DIAG_STAMP(4)
        // -------- predictor update (sarsa.cpp:98-124 / advantage.cpp:71-110)
        double delta = 0;
        if (update)
        {
          double target = reward;
          if (ADV)
          { // AdvantagePredictor::criticize (advantage.cpp:232-254)
            double v = -__builtin_inf();
#pragma unroll
            for (int kk = 0; kk < NA; ++kk) v = fmax(v, qprev[kk]);
            target = v + (reward - v) / N.kappa;
            if (has_next)
            {
              v = -__builtin_inf();
#pragma unroll
              for (int kk = 0; kk < NA; ++kk) v = fmax(v, q[kk]);
              target += N.gamma * v / N.kappa;
            }
          }
          else if (has_next)
          {
            if (SPEC::agent(P) == GRLX_AGENT_SARSA)
              target += N.gamma * pick<double, NA>(q, a_next);
            else if (SPEC::agent(P) == GRLX_AGENT_EXPECTED_SARSA)
            { // QPolicy::value (q.cpp:60-73) = sum_a Q(s',a) * EpsilonGreedySampler::distribution (greedy.cpp:220-238)
              const double de = eps_decay * N.epsilon;
              double v = 0;
#pragma unroll
              for (int kk = 0; kk < NA; ++kk)
              {
                double d = (q[kk] == best) ? 1. / man : 0.;
                if (d == 1) d = 1 - de;
                d += de / NA;
                v += q[kk] * d;
              }
              target += N.gamma * v;
            }
            else
            {
              double v = -__builtin_inf();
#pragma unroll
              for (int kk = 0; kk < NA; ++kk) v = fmax(v, q[kk]);
              target += N.gamma * v;
            }
          }
          delta = target - qsa;
          const double dW = N.alpha * (target - qsa);          // LinearRepresentation::write (linear.cpp:186-196)
          const double dT = N.alpha * delta;                   // VectorConstructor(alpha_*delta)
          if (DEFER)
          { // applied on the next pass, after that pass's loads are in flight
            if (TAP && up.use_trace)
            { // the length trace_->add will leave (trace.h:215-234), for the record of this step
              int l = tr.len;
              double tot = tr.total;
              if (up.ee < up.cut) { l = 0; tot = 1.; }
              l = (l < kMaxTrace) ? l + 1 : kMaxTrace;
              tot *= up.ee;
              while (tot < up.cut && l > 1) { tot /= up.ee; l--; }
              tr_len_ref = l;
            }
            pd = true;
            pd_dW = dW;
            pd_dT = dT;
            pd_pos = p_pos;
            status |= (p_pos == kInvalidPos) ? ST_BAD_POS : 0u;      // assert: an update always follows an action taken
            pd_sh = p_sh;
            pd_wp = wp;
          }
          else
          {
            up.dW = dW;
            up.dT = dT;
            Evicted none;
            td_update_lane<false>(tr, tab, up, p_pos, p_sh, wp, g, j, sh_ppos, sh_fb, sh_fbflag, status, none);
            tr_len_ref = tr.len;
          }
        }

        DIAG_STAMP(5)
        // -------- tap (debug / parity tests; only the immediate-update instantiation records taps)
        if (TAP && tapped && (!first || P.tap_starts))
        {
          uint32_t n = *P.tap_count;
          if (n < (uint32_t)P.tap_capacity)
          {
            grlx_tap *tp = &P.taps[n];
            tp->p_idx[j] = update ? p_slot : 0u;
            tp->p_idx[16 + j] = 0u;
            if (j == 0)
            {
              tp->test = test;
              tp->action_index = has_next ? a_next : action_index;
              tp->terminal = first ? -1 : terminal;
              tp->trace_len = tr_len_ref;
              for (int i = 0; i < GRLX_MAX_DIMS; ++i) tp->obs[i] = (i < D) ? obs[i] : 0.;
              tp->action = has_next ? pick<double, NA>(acts, a_next) : action;
              tp->reward = reward;
              for (int i = 0; i < GRLX_MAX_STATE; ++i) tp->state[i] = (i < S) ? x[i] : 0.;
              tp->delta = delta;
              for (int a = 0; a < kMaxActions; ++a) tp->q[a] = 0.;
#pragma unroll
              for (int a = 0; a < NA; ++a) tp->q[a] = has_next ? q[a] : 0.;
            }
          }
          wave_sync();
          if (j == 0) *P.tap_count = n + 1u;
        }

        // -------- bookkeeping
        if (!first)
        {
          if (test) test_steps++;
          else ss++;                                                       // online_learning.cpp:218
        }
        if (has_next)
        {
          action_index = a_next;
          action = pick<double, NA>(acts, a_next);                         // discretizer_->at(index), uniform.cpp:140-151
          p_pos = pick<uint32_t, NA>(pos, a_next);
          p_slot = pick<uint32_t, NA>(slot, a_next);
          p_sh = pick<bool, NA>(sh, a_next);
          wp_seen = pick<double, NA>(w, a_next);
          if (ADV)
          {
#pragma unroll
            for (int a = 0; a < NA; ++a) pos_prev[a] = pos[a];
          }
        }
        if (!first && terminal) running = false;
        first = false;
      }
      // the eviction held back by the deferred update: nothing reads the table before the next pass
      if (DEFER && ev.pos != kInvalidPos) value_store(tab, ev.pos, ev.val);
    }

    // end of the trial: the trace is cleared by the next TDAgent::start (td.cpp:54); write the
    // cached weights back now so that test trials and the host see them
    if (!test) trace_flush(tr, tab, true);
    }   // episodes of the trial

    // row of a test trial (online_learning.cpp:238-262) -- or of every trial when test_interval < 0
    if (act && (ti >= 0 ? test : 1))
    {
      if (rows < (uint32_t)P.max_rows)
      {
        if (j == 0)
        {
          size_t at = (size_t)rows * (size_t)P.n_replicas + (size_t)r;
          P.row_reward[at] = total_reward / (double)subtrials;              // :224-225
          P.row_time[at] = time / (double)subtrials;
          P.row_steps[at] = ss;
          P.row_trial[at] = (ti >= 0) ? (tt + 1 - (tt + 1) / (ti + 1)) : tt;
        }
        rows++;
      }
      else
        status |= ST_ROWS_FULL;
    }
    tt += act ? 1 : 0;
  }

  if (DIAG && P.diag_out && lane == 0)
    for (int k = 0; k < 8; ++k) P.diag_out[(size_t)blockIdx.x * 8 + k] = diag_sum[k];

  if constexpr (EXT)
  {
#ifdef GRLX_ENV_SERVER_STATS
    if (live && j == 0 && P.env_mail)
    {
      mail->stats[4] = mail_clock() - st_begin;
      mail->stats[5] = st_wait;
      mail->stats[6] = st_fetch;
      mail->stats[7] = st_polls;
      mail->stats[8] = srv ? 1 : 0;
      for (int k = 0; k < 8; ++k) mail->pad1[k] = diag_sum[k];
    }
#endif
    if (live && j == 0 && P.env_mail) mail->stats[15] = srv ? 1u : 2u;     // (grlx_env_server_counts: served to the end / fell back)
    if (srv)
    { // no further command in this launch
      ++mseq;
      if (j == 0) mail_send(mail, mseq, kMailExit);
    }
  }

  // write the replica back
  uint32_t ins = inserted;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) ins += __shfl_xor(ins, off, 16);
  if (live && j == 0)
  {
#pragma unroll
    for (int i = 0; i < S; ++i) RS.x[i] = x[i];
    RS.G = G;
    RS.TL = TL;
    RS.S1 = S1;
    RS.eps_decay = eps_decay;
    RS.tt = tt;
    RS.ss = ss;
    RS.test_steps = test_steps;
    RS.n_slots[0] += ins;
    RS.rows = rows;
  }
  // status may differ per lane (a probe failure is lane-local): OR over the replica
  uint32_t st = status;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) st |= __shfl_xor(st, off, 16);
  if (live && j == 0) RS.status = st;
}

template <int ENV, int NA, bool DIAG, typename SPEC, bool DEFER = !DIAG, bool ADV = false, bool TAP = !DEFER>
__global__ __launch_bounds__(64) GRLX_ROLLOUT_OCCUPANCY void rollout_kernel(DevParams P, int n_trials)
{
  rollout_body<ENV, NA, DIAG, SPEC, DEFER, ADV, TAP, false>(P, n_trials);
}

// The instantiation the environment server works for.  208 (x 2: vector + accumulation registers) = 416 of the SIMD's 512 registers,
// so that the server's wave (96) fits beside it.
template <int NA, typename SPEC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(208))) void rollout_served_kernel(DevParams P, int n_trials)
{
  rollout_body<GRLX_ENV_PENDULUM, NA, false, SPEC, true, false, false, true>(P, n_trials);
}

// cfg/cart_pole/ac_tc.yaml as compile-time constants (see SpecPendulumTcA): every field the actor-critic
// kernel reads, derived with the expressions of make_params (grlx_api.cpp)

} // namespace grlx
