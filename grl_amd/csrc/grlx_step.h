// grlx_step.h -- the reference's per-step plug-in interfaces on the replicas of a context: Environment::start / step
// (environment.h:48-51 -> ModeledEnvironment, modeled.cpp:132-213) and Agent::start / step / end (agent.h:44-56 -> agent/td,
// td.cpp:50-81; agent/fixed, fixed.cpp:47-65) as batched kernels, one call of the interface for every replica per launch.
// They serve graphs in which only ONE side of the loop of OnlineLearningExperiment::run (online_learning.cpp:172-213) lives on the
// GPU -- the agent beside an environment of the caller's (a robot, a simulator of grl's), or the environment beside an agent of the
// caller's -- from the first step of a trial on.  The fused kernels stay the fast path; these launch once per step.
// Part of the single translation unit grlx_kernels.hip (included there, in order; not self-contained).
//
// State between calls lives where the fused kernels keep it between launches, so the two can be mixed on one context:
//   ReplicaState      model state, the random streams, epsilon / noise decay, the trace's length and total decay
//   trace_state       the predictor's trace: table positions (+ occurrence count and shared flag) per tiling, newest first
//   AgentRep          TDAgent::time_, the previous action (value and index)
//   agent_lane        reference slots of project(prev_obs, prev_action) (Q agents) / of the critic's and the actor's
//                     project(prev_obs) (actor-critic), one per tiling: looked up again by the next call
// Nothing is cached across calls: every call restores the trace's weights from the table (which is current: the fused kernels
// write their trace back at the end of a trial or launch, these kernels at the end of every call), applies the update in place
// with the same td_update_lane the fused kernels run, and writes the weights back.  Trial and step counters, rows and the steps
// budget belong to the caller's loop and are not advanced.
#pragma once

namespace grlx {

// ------------------------------------------------------------------------------------------- environment ---
// Environment::start (modeled.cpp:132-158): one lane per replica; the start state is drawn from the replica's own streams
template <int ENV>
__global__ void env_start_kernel(DevParams P, int test, const int32_t *active, double *obs_out)
{
  constexpr int S = Env<ENV>::S, D = Env<ENV>::D;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= P.n_replicas || (active && active[r] == 0)) return;
  ReplicaState &RS = P.states[r];
  double x[S], o[D];
  uint64_t TL = RS.TL, G = RS.G;
  Env<ENV>::start(P, test, TL, G, x);
  Env<ENV>::observe(P, x, o);
#pragma unroll
  for (int k = 0; k < S; ++k) RS.x[k] = x[k];
  RS.TL = TL;
  RS.G = G;
#pragma unroll
  for (int k = 0; k < D; ++k) obs_out[(size_t)r * D + k] = o[k];
}

// Environment::step (modeled.cpp:160-213) on the replica's model state
template <int ENV>
__global__ void env_advance_kernel(DevParams P, const int32_t *active, const double *action, double *obs, double *reward, int32_t *terminal)
{
  constexpr int S = Env<ENV>::S, D = Env<ENV>::D;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= P.n_replicas || (active && active[r] == 0)) return;
  ReplicaState &RS = P.states[r];
  double x[S], o[D], rw;
  int term;
#pragma unroll
  for (int k = 0; k < S; ++k) x[k] = RS.x[k];
  uint32_t st = Env<ENV>::in_domain(x) ? 0u : ST_DOMAIN;
  env_step<ENV>(P, x, action[r], o, rw, term, st);
#pragma unroll
  for (int k = 0; k < S; ++k) RS.x[k] = x[k];
#pragma unroll
  for (int k = 0; k < D; ++k) obs[(size_t)r * D + k] = o[k];
  reward[r] = rw;
  terminal[r] = term;
  if (st) RS.status |= st;
}

hipError_t launch_env_start(const DevParams &P, int test, const int32_t *active_dev, double *obs_dev, hipStream_t stream)
{
  const int blocks = (P.n_replicas + 63) / 64;
  switch (P.env)
  {
#define GRLX_ENV_CASE(ENVID) case ENVID: hipLaunchKernelGGL(env_start_kernel<ENVID>, dim3(blocks), dim3(64), 0, stream, P, test, active_dev, obs_dev); break;
    GRLX_ENV_CASE(GRLX_ENV_PENDULUM)
    GRLX_ENV_CASE(GRLX_ENV_ACROBOT)
    GRLX_ENV_CASE(GRLX_ENV_CART_POLE)
    GRLX_ENV_CASE(GRLX_ENV_COMPASS_WALKER)
#undef GRLX_ENV_CASE
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_env_advance(const DevParams &P, const int32_t *active_dev, const double *action_dev, double *obs_dev, double *reward_dev,
                              int32_t *terminal_dev, hipStream_t stream)
{
  const int blocks = (P.n_replicas + 63) / 64;
  switch (P.env)
  {
#define GRLX_ENV_CASE(ENVID) case ENVID: hipLaunchKernelGGL(env_advance_kernel<ENVID>, dim3(blocks), dim3(64), 0, stream, P, active_dev, action_dev, obs_dev, reward_dev, terminal_dev); break;
    GRLX_ENV_CASE(GRLX_ENV_PENDULUM)
    GRLX_ENV_CASE(GRLX_ENV_ACROBOT)
    GRLX_ENV_CASE(GRLX_ENV_CART_POLE)
    GRLX_ENV_CASE(GRLX_ENV_COMPASS_WALKER)
#undef GRLX_ENV_CASE
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------- agents ---
// the predictor's trace of tiling j as the actor-critic kernels persist it: positions and flags from trace_state, weights from the table
__device__ __forceinline__ void step_trace_restore(TraceRegs &tr, const Table &tab, const uint32_t *ts, const ReplicaState &RS)
{
  tr.len = RS.tr_len;
  tr.total = RS.tr_total;
#pragma unroll
  for (int e = 0; e < kMaxTrace; ++e)
  {
    tr.pos[e] = ts[e * 2];
    const uint32_t cw = ts[e * 2 + 1];
    const uint32_t cn = cw & 0xFFFFu;
    tr.cnt2 |= ((cn > 0u ? cn - 1u : 0u) & 3u) << (2 * e);
    if (cw >> 16) tr.wt |= 1u << e;
    tr.dup = tr.dup || cn > 1u;
    if (tr.pos[e] != kInvalidPos) tr.val[e] = value_load(tab, tr.pos[e]);
  }
}

__device__ __forceinline__ void step_trace_persist(TraceRegs &tr, const Table &tab, uint32_t *ts)
{
  trace_flush(tr, tab, false);                          // the weights go back to the table: nothing is cached across calls
#pragma unroll
  for (int e = 0; e < kMaxTrace; ++e)
  {
    ts[e * 2] = tr.pos[e];
    ts[e * 2 + 1] = (trace_cnt(tr, e) & 0xFFFFu) | (((tr.wt >> e) & 1u) << 16);
  }
}

// One call of Agent::start / step / end of the discrete-action TD agents (agent/td with mapping/policy/discrete/value/q and
// predictor/critic/{sarsa, q, expected_sarsa}; test = 1: agent/fixed with the greedy policy over the same table) for the replicas
// of the context.  One wavefront = 4 replicas x 16 lanes, lane = tiling, as in rollout_kernel; the arithmetic of a step and its
// order are those of rollout_body's in-place instantiation (the policy's Q(s', .), the sampler, then criticize: write, trace
// update, trace add).  Observation dimensions are a run-time value (the agent does not know the environment).
template <int NA>
__global__ __launch_bounds__(64) void agent_step_kernel(DevParams P, StepArgs A)
{
  constexpr int T = kLanesPerReplica, NROWS = NA + 1;
  __shared__ double   sh_w[NROWS * 16 * 4];
  __shared__ uint32_t sh_ppos[4 * 16];
  __shared__ double   sh_fb[16 * 4];
  __shared__ uint32_t sh_fbflag[16 * 4];
  __shared__ uint32_t sh_mb[4 * NA * 16];
  __shared__ uint32_t sh_ms[4 * NA * 16];
  __shared__ uint32_t sh_mail[4];
  __shared__ double   sh_res[4 * 16];
  __shared__ uint64_t sh_jump[2048];
  jump_table_to_lds(sh_jump);

  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, j = lane & 15;
  const int r_raw = blockIdx.x * kReplicasPerWave + g;
  const bool live = r_raw < P.n_replicas;
  const int r = live ? r_raw : 0;
  const unsigned long long gmask = 0xFFFFull << (16 * g);
  const int D = P.tile.D - 1;                                    // observation dimensions: the projector's input is (obs, action)

  const bool on = live && (!A.active || A.active[r] != 0);
  int mode = A.mode;
  if (mode == STEP_STEP && A.terminal && on && A.terminal[r] == 2) mode = STEP_END;      // online_learning.cpp:210-213
  const int test = A.test;
  const bool first = mode == STEP_START;
  const bool has_next = on && mode != STEP_END;
  const bool update = on && !first && !test;

  ReplicaState &RS = P.states[r];
  AgentRep &AR = P.agent_rep[r];
  uint64_t G = RS.G, S1 = RS.S1;
  double eps_decay = RS.eps_decay;
  uint32_t status = RS.status, inserted = 0;
  double time = AR.time;
  int action_index = AR.action_index;
  const Table tab = table_of(P, 0, r);

  UpdateParams up;
  up.out_min = P.lin.out_min;
  up.out_max = P.lin.out_max;
  up.limit = P.lin.limit != 0;
  up.ee = P.gl;
  up.cut = 0.01;
  up.use_trace = P.trace_kind == GRLX_TRACE_REPLACING;
  up.dW = up.dT = 0;

  // the learning agent's trace: empty at Agent::start (TDAgent::start -> predictor_->finalize(), td.cpp:54, sarsa.cpp:126-132), else
  // as the previous call left it.  The test agent has none and leaves the learning agent's alone.
  TraceRegs tr;
  trace_init(tr);
  uint32_t *ts = P.trace_state + ((size_t)r * 16 + (size_t)j) * kMaxTrace * 2;
  const bool have_trace = on && up.use_trace && !(first && !test);
  if (have_trace) step_trace_restore(tr, tab, ts, RS);
  if (on && !test && first) time = 0;                          // td.cpp:55
  if (on && !test && !first) time += 1;                        // td.cpp:65 / :78: time_ += tau, tau = 1 (discrete_time)

  double obs[GRLX_MAX_DIMS];
#pragma unroll
  for (int i = 0; i < GRLX_MAX_DIMS; ++i) obs[i] = (on && i < D) ? A.obs[(size_t)r * D + i] : 0.;
  const double reward = (on && !first) ? A.reward[r] : 0.;

  // project(prev_obs, prev_action): the slot this tiling wrote down at the previous call, looked up again (it exists)
  uint32_t p_pos = kInvalidPos;
  bool p_sh = false;
  double wp = 0;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  if (update)
  {
    uint32_t ps[1] = {P.agent_lane[((size_t)r * 16 + (size_t)j) * 2]}, pp[1] = {kInvalidPos};
    double pw[1] = {0};
    bool psh[1] = {false};
    table_get<1>(tab, P.lin, RS, 0, ps, pp, pw, psh, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump, status, inserted,
                 [&](uint32_t mp) { trace_share_event(tr, tab, mp); });
    p_pos = pp[0];
    p_sh = psh[0];
    wp = pw[0];
  }

  // -------- policy: Q(s', .) for all actions (q.cpp:94-107)
  uint32_t slot[NA], pos[NA];
  double w[NA], q[NA];
  bool sh[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) { slot[a] = 0; pos[a] = kInvalidPos; w[a] = 0; q[a] = 0; sh[a] = false; }
  if (has_next)
  {
    uint32_t hpre = 449u ^ (uint32_t)(D + 2);
    for (int i = 0; i < D; ++i) hpre = murmur_mix(hpre, tile_coord<T>(P.tile, i, tile_quant(P.tile, i, obs[i]), j));
#pragma unroll
    for (int a = 0; a < NA; ++a)
    {
      uint32_t h = murmur_mix(hpre, tile_coord<T>(P.tile, D, tile_quant(P.tile, D, P.actions[a]), j));
      h = murmur_mix(h, j);
      slot[a] = murmur_final(h) % (uint32_t)P.tile.memory;
    }
    bool shared_event = false;
    table_get<NA>(tab, P.lin, RS, 0, slot, pos, w, sh, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump, status, inserted,
                  [&](uint32_t mp) {
                    trace_share_event(tr, tab, mp);
                    if (p_pos == mp) p_sh = true;
                    shared_event = true;
                  });
    if (rarely(__any(shared_event)) && update) wp = value_load(tab, p_pos);
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
    SHW(NA, j, g) = wp;
  }
  sh_ppos[g * 16 + j] = p_pos;
  sh_fbflag[j * 4 + g] = 0u;
  wave_sync();
  { // LinearRepresentation::read (linear.cpp:136-184): serial sum over the 16 tilings, mean, clamp
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
  const double qsa = update ? clampd(sh_res[g * 16 + NA], up.out_min, up.out_max) : 0.;

  // -------- sampler (greedy.cpp:63-86, 144-218)
  int a_next = 0, mai = 0, man = 1;
  double best = 0;
  if (has_next)
  {
    findmax<NA>(q, mai, man, best);
    if (test)
      a_next = (man > 1) ? tie_break<NA>(q, best, man, G) : mai;
    else
    {
      if (time == 0.) eps_decay = fmax(eps_decay * P.decay_rate, P.decay_min);
      S1 = lcg_next(S1);
      const double rnd = lcg_double(S1);
      if (rnd < eps_decay * P.epsilon)
      {
        G = lcg_next(G);
        a_next = (int)(lcg_long(G) % (uint32_t)NA);
      }
      else
        a_next = (man > 1) ? tie_break<NA>(q, best, man, G) : mai;
    }
  }

  // -------- predictor update (sarsa.cpp:98-124 / advantage.cpp:71-110 / sarsa.cpp:167-194)
  if (update)
  {
    double target = reward;
    if (has_next)
    {
      if (P.agent == GRLX_AGENT_SARSA)
        target += P.gamma * pick<double, NA>(q, a_next);
      else if (P.agent == GRLX_AGENT_EXPECTED_SARSA)
      {
        const double de = eps_decay * P.epsilon;
        double v = 0;
#pragma unroll
        for (int kk = 0; kk < NA; ++kk)
        {
          double d = (q[kk] == best) ? 1. / man : 0.;
          if (d == 1) d = 1 - de;
          d += de / NA;
          v += q[kk] * d;
        }
        target += P.gamma * v;
      }
      else
      {
        double v = -__builtin_inf();
#pragma unroll
        for (int kk = 0; kk < NA; ++kk) v = fmax(v, q[kk]);
        target += P.gamma * v;
      }
    }
    const double delta = target - qsa;
    up.dW = P.alpha * (target - qsa);                    // LinearRepresentation::write (linear.cpp:186-196)
    up.dT = P.alpha * delta;
    status |= (p_pos == kInvalidPos) ? ST_BAD_POS : 0u;
    Evicted none;
    td_update_lane<false>(tr, tab, up, p_pos, p_sh, wp, g, j, sh_ppos, sh_fb, sh_fbflag, status, none);
  }

  // -------- what the next call starts from
  // (a learning start persists the cleared trace; a test call only the flags a lookup of its own may have set)
  if (on && up.use_trace) step_trace_persist(tr, tab, ts);
  if (has_next && !test) P.agent_lane[((size_t)r * 16 + (size_t)j) * 2] = pick<uint32_t, NA>(slot, a_next);
  uint32_t ins = inserted;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) ins += __shfl_xor(ins, off, 16);
  uint32_t st = status;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) st |= __shfl_xor(st, off, 16);
  if (on && j == 0)
  {
    RS.G = G;
    RS.S1 = S1;
    RS.eps_decay = eps_decay;
    RS.n_slots[0] += ins;
    RS.status = st;
    if (!test)
    {
      if (up.use_trace) { RS.tr_len = tr.len; RS.tr_total = tr.total; }
      AR.time = time;
    }
    if (has_next)
    {
      const double act = P.actions[a_next];                      // discretizer_->at(index), uniform.cpp:140-151
      A.action[r] = act;
      if (!test) { AR.action = act; AR.action_index = a_next; }
    }
  }
  (void)action_index;
}

// Agent::start / step / end of the actor-critic agent (agent/td with mapping/policy/action and predictor/ac/action over a
// predictor/critic/td critic; test = 1: agent/fixed with the noise-free policy/action over the same actor table).  The
// arithmetic and its order are those of rollout_ac_kernel's in-place instantiation; the critic's trace is never cleared
// (ac.cpp:170-173), it continues from whatever the previous call -- or the previous fused launch -- left in trace_state.
__global__ __launch_bounds__(64) void agent_ac_step_kernel(DevParams P, StepArgs A)
{
  constexpr int T = kLanesPerReplica;
  __shared__ double   sh_w[4 * 16 * 4];        // rows: actor(s'), critic(s'), actor(s), critic(s)
  __shared__ uint32_t sh_ppos[4 * 16];
  __shared__ uint32_t sh_apos[4 * 16];
  __shared__ double   sh_fb[16 * 4];
  __shared__ uint32_t sh_fbflag[16 * 4];
  __shared__ uint32_t sh_mb[4 * 16];
  __shared__ uint32_t sh_ms[4 * 16];
  __shared__ uint32_t sh_mail[4];
  __shared__ double   sh_res[4 * 16];
  __shared__ uint64_t sh_jump[2048];
  jump_table_to_lds(sh_jump);

  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, j = lane & 15;
  const int r_raw = blockIdx.x * kReplicasPerWave + g;
  const bool live = r_raw < P.n_replicas;
  const int r = live ? r_raw : 0;
  const unsigned long long gmask = 0xFFFFull << (16 * g);
  const int D = P.tile.D;

  const bool on = live && (!A.active || A.active[r] != 0);
  int mode = A.mode;
  if (mode == STEP_STEP && A.terminal && on && A.terminal[r] == 2) mode = STEP_END;
  const int test = A.test;
  const bool first = mode == STEP_START;
  const bool has_next = on && mode != STEP_END;
  const bool update = on && !first && !test;
  const bool need_critic = has_next && !test;

  ReplicaState &RS = P.states[r];
  AgentRep &AR = P.agent_rep[r];
  uint64_t TL = RS.TL;
  double ac_decay = RS.ac_decay, ac_noise = RS.ac_noise;
  uint32_t status = RS.status, ins_c = 0, ins_a = 0;
  double time = AR.time;
  const double action = AR.action;                             // transition.prev_action
  const Table tabC = table_of(P, 0, r), tabA = table_of(P, 1, r);

  UpdateParams up;
  up.out_min = P.lin.out_min;
  up.out_max = P.lin.out_max;
  up.limit = P.lin.limit != 0;
  up.ee = P.gl;
  up.cut = 0.01;
  up.use_trace = P.trace_kind == GRLX_TRACE_REPLACING;
  up.dW = up.dT = 0;
  const double a_min = P.lin_actor.out_min, a_max = P.lin_actor.out_max;
  const bool a_limit = P.lin_actor.limit != 0;

  TraceRegs tr;
  trace_init(tr);
  uint32_t *ts = P.trace_state + ((size_t)r * 16 + (size_t)j) * kMaxTrace * 2;
  if (on && up.use_trace) step_trace_restore(tr, tabC, ts, RS);      // (never cleared: ac.cpp:170-173)
  if (on && !test && first) time = 0;
  if (on && !test && !first) time += 1;

  double obs[GRLX_MAX_DIMS];
#pragma unroll
  for (int i = 0; i < GRLX_MAX_DIMS; ++i) obs[i] = (on && i < D) ? A.obs[(size_t)r * D + i] : 0.;
  const double reward = (on && !first) ? A.reward[r] : 0.;

  // project(prev_obs) of the critic and of the actor: the slots written down by the previous call, looked up again
  uint32_t p_pos = kInvalidPos, ap_pos = kInvalidPos;
  bool p_sh = false, ap_sh = false;
  double wpc = 0, wap = 0;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  if (update)
  {
    const uint32_t *al = P.agent_lane + ((size_t)r * 16 + (size_t)j) * 2;
    uint32_t sc[1] = {al[0]}, sa[1] = {al[1]}, pc[1] = {kInvalidPos}, pa[1] = {kInvalidPos};
    double vc[1] = {0}, va[1] = {0};
    bool hc[1] = {false}, ha[1] = {false};
    if (P.twin_tables)
    { // the same slot at the same position in both tables: one resolution
      Lookup lk[1];
      BucketRegs brA[1], brC[1];
      table_issue<1>(tabA, sa, lk, brA);
      bucket_load_vals(tabC, table_home(tabC, sa[0]), brC[0]);
      table_get_finish_twin<1>(tabA, tabC, P.lin_actor, P.lin, RS, sa, lk, brA, brC, true, pa, va, vc[0], ha, g, j, gmask, sh_mb, sh_ms, sh_mail,
                               sh_jump, status, ins_a, ins_c, [&](uint32_t mp) { trace_share_event(tr, tabC, mp); });
      pc[0] = pa[0];
      hc[0] = ha[0];
    }
    else
    {
      table_get<1>(tabA, P.lin_actor, RS, 1, sa, pa, va, ha, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump, status, ins_a, [&](uint32_t) {});
      table_get<1>(tabC, P.lin, RS, 0, sc, pc, vc, hc, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump, status, ins_c,
                   [&](uint32_t mp) { trace_share_event(tr, tabC, mp); });
    }
    p_pos = pc[0]; p_sh = hc[0]; wpc = vc[0];
    ap_pos = pa[0]; ap_sh = ha[0]; wap = va[0];
  }

  // actor and critic at s'
  uint32_t slotA[1] = {0}, slotC[1] = {0}, posA[1] = {kInvalidPos}, posC[1] = {kInvalidPos};
  double wA[1] = {0}, wC[1] = {0};
  bool shA[1] = {false}, shC[1] = {false};
  if (has_next)
  {
    slotA[0] = tile_slot_obs<T>(P.tile_actor, obs, D, j);
    slotC[0] = tile_slot_obs<T>(P.tile, obs, D, j);
    bool shared_event = false;
    if (P.twin_tables)
    {
      Lookup lk[1];
      BucketRegs brA[1], brC[1];
      table_issue<1>(tabA, slotA, lk, brA);
      bucket_load_vals(tabC, table_home(tabC, slotC[0]), brC[0]);
      table_get_finish_twin<1>(tabA, tabC, P.lin_actor, P.lin, RS, slotA, lk, brA, brC, need_critic, posA, wA, wC[0], shA, g, j, gmask,
                               sh_mb, sh_ms, sh_mail, sh_jump, status, ins_a, ins_c,
                               [&](uint32_t mp) {
                                 if (ap_pos == mp) ap_sh = true;
                                 trace_share_event(tr, tabC, mp);
                                 if (p_pos == mp) p_sh = true;
                                 shared_event = true;
                               });
      posC[0] = posA[0];
      shC[0] = shA[0];
    }
    else
    {
      table_get<1>(tabA, P.lin_actor, RS, 1, slotA, posA, wA, shA, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump, status, ins_a,
                   [&](uint32_t mp) { if (ap_pos == mp) ap_sh = true; });
      if (need_critic)
        table_get<1>(tabC, P.lin, RS, 0, slotC, posC, wC, shC, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump, status, ins_c,
                     [&](uint32_t mp) {
                       trace_share_event(tr, tabC, mp);
                       if (p_pos == mp) p_sh = true;
                       shared_event = true;
                     });
    }
    if (rarely(__any(shared_event)) && update) wpc = value_load(tabC, p_pos);
  }
  if (need_critic) wC[0] = trace_forward(tr, posC[0], wC[0]);
  if (update) wpc = trace_forward(tr, p_pos, wpc);
  SHA(0, j, g) = wA[0];
  SHA(1, j, g) = wC[0];
  SHA(2, j, g) = wap;
  SHA(3, j, g) = wpc;
  sh_ppos[g * 16 + j] = p_pos;
  sh_fbflag[j * 4 + g] = 0u;
  sh_apos[g * 16 + j] = ap_pos;
  wave_sync();
  {
    const int row = j & 3;
    double sum = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += SHA(row, k, g);
    sh_res[g * 16 + j] = sum / 16;
  }
  wave_sync();
  const double u_next = clampd(sh_res[g * 16 + 0], a_min, a_max);
  const double v_next = clampd(sh_res[g * 16 + 1], up.out_min, up.out_max);
  const double u_prev = clampd(sh_res[g * 16 + 2], a_min, a_max);
  const double v_prev = clampd(sh_res[g * 16 + 3], up.out_min, up.out_max);

  // -------- policy (ActionPolicy::act, action.cpp:127-158)
  double a_next = 0;
  if (has_next)
  {
    double out = u_next;
    if (!test)
    {
      if (time == 0) ac_noise = 0;
      if (time == 0.) ac_decay = fmax(ac_decay * P.ac_decay_rate, P.ac_decay_min);
      if (P.sigma != 0)
      { // Rand::getNormal(0, decay*sigma): two thread-local draws (utils.h:120-125)
        TL = lcg_next(TL);
        const double U1 = lcg_double(TL);
        TL = lcg_next(TL);
        const double U2 = lcg_double(TL);
        const double sg = ac_decay * P.sigma;
        const double nrm = __builtin_sqrt(-2 * plog(U1)) * pcos(2 * GRLX_PI * U2) * sg + 0.;
        ac_noise = (1 - P.theta) * ac_noise + nrm;
        out += ac_noise;
      }
    }
    a_next = fmin(fmax(out, P.action_min), P.action_max);
  }

  // -------- predictor (ActionACPredictor::update, ac.cpp:72-110)
  if (update)
  {
    double target = reward;                                   // critic: TDPredictor::criticize (td.cpp:68-91)
    if (has_next) target += P.gamma * v_next;
    const double delta = target - v_prev;
    up.dW = P.alpha * (target - v_prev);
    up.dT = P.alpha * delta;
    status |= (p_pos == kInvalidPos) ? ST_BAD_POS : 0u;
    Evicted none;
    td_update_lane<false>(tr, tabC, up, p_pos, p_sh, wpc, g, j, sh_ppos, sh_fb, sh_fbflag, status, none);
    if (P.ac_update_method == 0 || delta > 0)
    { // actor
      double du = action - u_prev;
      if (P.ac_update_method == 0) du = delta * du;
      if (P.ac_step_limit >= 0) du = fmin(fmax(du, -P.ac_step_limit), P.ac_step_limit);
      const double target_u = u_prev + du;
      const double dA = P.actor_alpha * (target_u - u_prev);
      uint32_t cpa = 1;                                        // a slot that occurs twice in the projection is updated twice
      const uint32_t amask = (uint32_t)((__ballot(ap_sh) >> (16 * g)) & 0xFFFFull);
      for (uint32_t mm = amask; mm != 0u; mm &= mm - 1u)
      {
        const int k = __builtin_ctz(mm);
        if (k != j && sh_apos[g * 16 + k] == ap_pos) cpa++;
      }
      double nv = wap;
      for (uint32_t c = 0; c < cpa; ++c) nv = a_limit ? clampd(nv + dA, a_min, a_max) : nv + dA;
      value_store(tabA, ap_pos, nv);
    }
  }

  if (on && up.use_trace) step_trace_persist(tr, tabC, ts);
  if (has_next && !test)
  {
    uint32_t *al = P.agent_lane + ((size_t)r * 16 + (size_t)j) * 2;
    al[0] = slotC[0];
    al[1] = slotA[0];
  }
  uint32_t ic = ins_c, ia = ins_a;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) { ic += __shfl_xor(ic, off, 16); ia += __shfl_xor(ia, off, 16); }
  uint32_t st = status;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) st |= __shfl_xor(st, off, 16);
  if (on && j == 0)
  {
    RS.TL = TL;
    RS.ac_decay = ac_decay;
    RS.ac_noise = ac_noise;
    RS.n_slots[0] += ic;
    RS.n_slots[1] += ia;
    RS.status = st;
    if (up.use_trace) { RS.tr_len = tr.len; RS.tr_total = tr.total; }
    if (!test) AR.time = time;
    if (has_next)
    {
      A.action[r] = a_next;
      if (!test) AR.action = a_next;
    }
  }
}

hipError_t launch_agent_step(const DevParams &P, const StepArgs &A, hipStream_t stream)
{
  const int waves = (P.n_replicas + kReplicasPerWave - 1) / kReplicasPerWave;
  if (P.agent == GRLX_AGENT_AC)
    hipLaunchKernelGGL(agent_ac_step_kernel, dim3(waves), dim3(64), 0, stream, P, A);
  else if (P.A == 3)
    hipLaunchKernelGGL(agent_step_kernel<3>, dim3(waves), dim3(64), 0, stream, P, A);
  else if (P.A == 5)
    hipLaunchKernelGGL(agent_step_kernel<5>, dim3(waves), dim3(64), 0, stream, P, A);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

} // namespace grlx
