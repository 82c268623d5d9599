// grlx_rollout_ac.h -- actor-critic rollout (cfg/cart_pole/ac_tc.yaml): its compile-time specialisation and rollout_ac_kernel.
// Part of the single translation unit grlx_kernels.hip (included there, in order; not self-contained).
#pragma once

namespace grlx {

constexpr DevParams make_spec_cart_pole_ac()
{
  DevParams P = {};
  P.env = GRLX_ENV_CART_POLE;
  P.agent = GRLX_AGENT_AC;
  P.trace_kind = GRLX_TRACE_REPLACING;
  P.test_interval = 10;
  P.integration_steps = 5;
  P.h = 0.05 / 5.0;
  P.control_step = 0.05;
  P.timeout = 9.99;
  P.randomization = 0;
  P.end_stop_penalty = 0;
  P.action_penalty = 0;
  P.action_min = -15;
  P.action_max = 15;
  P.A = 0;
  const double res[4] = {2.5, 0.157075, 2.5, 1.57075};
  P.tile.T = 16; P.tile.D = 4; P.tile.memory = 8388608;
  P.tile_actor.T = 16; P.tile_actor.D = 4; P.tile_actor.memory = 8388608;
  for (int i = 0; i < 4; ++i) { P.tile.scaling[i] = 16 / res[i]; P.tile_actor.scaling[i] = 16 / res[i]; }
  P.tile.wrap[1] = 640; P.tile_actor.wrap[1] = 640;              // round(6.283 * 16 / 0.157075)
  P.lin.init_min = 0; P.lin.init_range = 1;
  P.lin.out_min = -1.7976931348623157e308; P.lin.out_max = 1.7976931348623157e308;
  P.lin.limit = 1; P.lin.draws_before = 8388608;
  P.lin_actor.init_min = 0; P.lin_actor.init_range = 1;
  P.lin_actor.out_min = -15; P.lin_actor.out_max = 15;
  P.lin_actor.limit = 1; P.lin_actor.draws_before = 0;
  P.actor_alpha = 0.01; P.sigma = 5; P.theta = 1; P.ac_decay_rate = 1; P.ac_decay_min = 0;
  P.ac_step_limit = -1; P.ac_update_method = 0;
  P.alpha = 0.2; P.gamma = 0.97; P.gl = 0.97 * 0.65;
  return P;
}
__device__ const DevParams d_spec_cart_pole_ac = make_spec_cart_pole_ac();

struct SpecCartPoleAc {
  static bool same_tile(const TileParams &a, const TileParams &b)
  {
    bool ok = a.T == b.T && a.D == b.D && a.memory == b.memory;
    for (int i = 0; i < GRLX_MAX_DIMS; ++i) ok = ok && a.scaling[i] == b.scaling[i] && a.wrap[i] == b.wrap[i];
    return ok;
  }
  static bool same_lin(const LinearParams &a, const LinearParams &b)
  {
    return a.init_min == b.init_min && a.init_range == b.init_range && a.out_min == b.out_min && a.out_max == b.out_max &&
           a.limit == b.limit && a.draws_before == b.draws_before;
  }
  static bool matches(const DevParams &P)
  {
    constexpr DevParams C = make_spec_cart_pole_ac();
    return P.env == C.env && P.agent == C.agent && P.trace_kind == C.trace_kind && P.test_interval == C.test_interval &&
           P.integration_steps == C.integration_steps && P.h == C.h && P.control_step == C.control_step && P.timeout == C.timeout &&
           P.randomization == C.randomization && P.end_stop_penalty == C.end_stop_penalty && P.action_penalty == C.action_penalty &&
           P.action_min == C.action_min && P.action_max == C.action_max && same_tile(P.tile, C.tile) && same_tile(P.tile_actor, C.tile_actor) &&
           same_lin(P.lin, C.lin) && same_lin(P.lin_actor, C.lin_actor) && P.actor_alpha == C.actor_alpha && P.sigma == C.sigma &&
           P.theta == C.theta && P.ac_decay_rate == C.ac_decay_rate && P.ac_decay_min == C.ac_decay_min && P.ac_step_limit == C.ac_step_limit &&
           P.ac_update_method == C.ac_update_method && P.alpha == C.alpha && P.gamma == C.gamma && P.gl == C.gl;
  }
  __device__ static __forceinline__ const DevParams &numeric(const DevParams &) { return d_spec_cart_pole_ac; }
};

// ------------------------------------------------------ actor-critic rollout ---
// agent/td { policy: mapping/policy/action, predictor: predictor/ac/action { critic:
// predictor/critic/td } } with agent/fixed for test trials (cfg/cart_pole/ac_tc.yaml).
// Table 0 = critic V(s) with the register trace, table 1 = actor u(s) (no trace: plain
// read-modify-write).  Lane j = tiling j of both projectors.  References:
//   ActionPolicy::act        base/src/policies/action.cpp:127-158
//   ActionACPredictor::update base/src/predictors/ac.cpp:72-110
//   TDPredictor::criticize    base/src/predictors/td.cpp:68-91
//   Rand::getNormal           base/include/grl/utils.h:120-125
// Quirk kept: ActionACPredictor::finalize (ac.cpp:170-173) does not reach the critic, so the
// critic's trace is NOT cleared at episode start; it survives test trials and launches.
template <int T>
__device__ __forceinline__ uint32_t tile_slot_obs(const TileParams &tp, const double *obs, int D, int j)
{
  uint32_t h = 449u ^ (uint32_t)(D + 1);
  for (int i = 0; i < D; ++i) h = murmur_mix(h, tile_coord<T>(tp, i, tile_quant(tp, i, obs[i]), j));
  h = murmur_mix(h, j);
  return murmur_final(h) % (uint32_t)tp.memory;
}

#define SHA(row, k, g) sh_w[(((row) * 16 + (k)) << 2) + (g)]

// DEFER: the critic's TD update of a step is applied one pass later, between the next step's table loads and
// their first use (as in rollout_kernel); the taps need the in-place ordering.
template <int ENV, typename SPEC, bool DEFER>
__global__ __launch_bounds__(64) void rollout_ac_kernel(DevParams P, int n_trials)
{
  // N: numeric parameters (compile-time constants in a specialised build); P: pointers and sizes
  const DevParams &N = SPEC::numeric(P);
  constexpr int S = Env<ENV>::S, D = Env<ENV>::D, T = kLanesPerReplica;
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
  const bool tapped = live && (r == P.tap_replica);
  const unsigned long long gmask = 0xFFFFull << (16 * g);

  ReplicaState &RS = P.states[r];
  double x[S];
#pragma unroll
  for (int i = 0; i < S; ++i) x[i] = RS.x[i];
  uint64_t G = RS.G, TL = RS.TL;
  double ac_decay = RS.ac_decay, ac_noise = RS.ac_noise;
  int64_t tt = RS.tt, ss = RS.ss;
  uint64_t test_steps = RS.test_steps;
  uint32_t status = RS.status, rows = RS.rows, ins_c = 0, ins_a = 0;

  const Table tabC = table_of(P, 0, r), tabA = table_of(P, 1, r);
  UpdateParams up;
  up.out_min = N.lin.out_min;
  up.out_max = N.lin.out_max;
  up.limit = N.lin.limit != 0;
  up.ee = N.gl;
  up.cut = 0.01;
  up.use_trace = N.trace_kind == GRLX_TRACE_REPLACING;
  up.dW = up.dT = 0;
  const double a_min = N.lin_actor.out_min, a_max = N.lin_actor.out_max;
  const bool a_limit = N.lin_actor.limit != 0;

  // restore the critic's trace: positions from HBM, weights from the (current) table
  TraceRegs tr;
  trace_init(tr);
  uint32_t *ts = P.trace_state + ((size_t)r * 16 + (size_t)j) * kMaxTrace * 2;
  if (live && up.use_trace)
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
      if (tr.pos[e] != kInvalidPos) tr.val[e] = value_load(tabC, tr.pos[e]);
    }
  }

  bool pd = false, pd_sh = false;          // pending critic update (DEFER)
  double pd_dW = 0, pd_dT = 0, pd_wp = 0;
  uint32_t pd_pos = kInvalidPos;

  for (int trial = 0; trial < n_trials; ++trial)
  {
    // online_learning.cpp:154: a replica whose learning steps have reached the steps budget starts no further trial
    const bool act = live && !(P.steps_budget != 0u && (uint64_t)ss >= P.steps_budget);
    if (!__any(act)) break;
    const int ti = N.test_interval;
    const int test = (ti >= 0 && tt % (ti + 1) == ti) ? 1 : 0;
    // a test trial is test_trials greedy episodes (online_learning.cpp:161-170), reward and time adding up across them; `time` doubles
    // as the agent's episode time, which only learning episodes read (noise reset and decay at time 0)
    double total_reward = 0, time = 0;
    const int subtrials = (test && P.test_trials > 1) ? P.test_trials : 1;
    for (int st = 0; st < P.test_trials; ++st)
    {
    const bool episode = act && st < subtrials;
    if (!__any(episode)) break;
    double obs[D], reward = 0;
    int terminal = 0;
    bool running = episode;
    if (episode)
    {
      Env<ENV>::start(N, test, TL, G, x);
      Env<ENV>::observe(N, x, obs);
    }
    double action = 0;
    uint32_t p_pos = kInvalidPos, p_slot = 0, ap_pos = kInvalidPos, ap_slot = 0;
    bool p_sh = false, ap_sh = false;
    double wap_seen = 0, wpc_seen = 0;
    bool first = true;

    for (;;)
    {
      if (!__any(running || pd)) break;
      // state that lives across the deferred-update site
      uint32_t slotA[1] = {0}, slotC[1] = {0};
      Lookup lkA[1], lkC[1];
      BucketRegs brA[1], brC[1];
      double wap = 0, wpc = 0;
      bool has_next = false, update = false, need_critic = false;
      if (running)
      {
        if (!first)
        {
          env_step<ENV>(N, x, action, obs, reward, terminal, status);
          total_reward += reward;
          time += 1;
        }
        has_next = first || terminal != 2;
        update = !first && !test;
        need_critic = has_next && !test;
        if (has_next)
        {
          slotA[0] = tile_slot_obs<T>(N.tile_actor, obs, D, j);
          slotC[0] = tile_slot_obs<T>(N.tile, obs, D, j);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (update)
        { // weights of project(prev_obs): with the deferred ordering as looked up one pass ago (the actor's: or as written by
          // the last actor update to the same slot; shared slots are loaded); the in-place ordering loads them
          wap = (DEFER && !ap_sh) ? wap_seen : value_load(tabA, ap_pos);
          wpc = DEFER ? wpc_seen : value_load(tabC, p_pos);
        }
        // both tables' home buckets in flight together: one memory round trip for the two lookups
        if (has_next) table_issue<1>(tabA, slotA, lkA, brA);
        if (need_critic)
        {
          if (P.twin_tables) bucket_load_vals(tabC, table_home(tabC, slotC[0]), brC[0]);     // same keys as the actor's bucket
          else table_issue<1>(tabC, slotC, lkC, brC);
        }
      }

      // -------- the PREVIOUS step's critic update, in the shadow of the loads just issued
      Evicted ev;
      ev.n = 0u; ev.pos = kInvalidPos; ev.val = 0;
      if (DEFER)
      {
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
          td_update_lane<true>(tr, tabC, up, pd_pos, pd_sh, pd_wp, g, j, sh_ppos, sh_fb, sh_fbflag, status, ev);
          pd = false;
        }
      }

      if (running)
      {
        uint32_t posA[1] = {kInvalidPos}, posC[1] = {kInvalidPos};
        double wA[1] = {0}, wC[1] = {0};
        bool shA[1] = {false}, shC[1] = {false};
        if (P.twin_tables)
        { // equal tile codings: one resolution, one creation path for both tables (table_get_finish_twin)
          if (has_next)
          {
            bool shared_event = false;
            table_get_finish_twin<1>(tabA, tabC, N.lin_actor, N.lin, RS, slotA, lkA, brA, brC, need_critic, posA, wA, wC[0], shA, g, j, gmask,
                                     sh_mb, sh_ms, sh_mail, sh_jump, status, ins_a, ins_c,
                                     [&](uint32_t mp) {
                                       if (ap_pos == mp) ap_sh = true;
                                       if (DEFER && ev.pos != kInvalidPos && ev.pos == mp) value_store(tabC, mp, ev.val);
                                       trace_share_event(tr, tabC, mp);
                                       if (p_pos == mp) p_sh = true;
                                       shared_event = true;
                                     });
            posC[0] = posA[0];
            shC[0] = shA[0];
            if (rarely(__any(shared_event)) && update) wpc = value_load(tabC, p_pos);
          }
        }
        else
        {
          if (has_next)
          {
            table_get_finish<1>(tabA, N.lin_actor, RS, 1, slotA, lkA, brA, posA, wA, shA, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump, status, ins_a,
                                [&](uint32_t mp) { if (ap_pos == mp) ap_sh = true; });
          }
          if (need_critic)
          {
            bool shared_event = false;
            table_get_finish<1>(tabC, N.lin, RS, 0, slotC, lkC, brC, posC, wC, shC, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump, status, ins_c,
                         [&](uint32_t mp) {
                           if (DEFER && ev.pos != kInvalidPos && ev.pos == mp) value_store(tabC, mp, ev.val);
                           trace_share_event(tr, tabC, mp);
                           if (p_pos == mp) p_sh = true;
                           shared_event = true;
                         });
            if (rarely(__any(shared_event)) && update) wpc = value_load(tabC, p_pos);
          }
        }
        if (DEFER)
        { // critic values loaded before the deferred update: reload where the update wrote the table, patch the held eviction
          const bool risky = ev.n > 1u || (update && p_sh) || (need_critic && shC[0]);
          if (rarely(__any(risky)))
          {
            if (need_critic) wC[0] = value_load(tabC, posC[0]);
            if (update) wpc = value_load(tabC, p_pos);
          }
          const bool held = ev.pos != kInvalidPos;
          wC[0] = (held && posC[0] == ev.pos) ? ev.val : wC[0];
          wpc = (held && p_pos == ev.pos) ? ev.val : wpc;
        }
        if (need_critic) wC[0] = trace_forward(tr, posC[0], wC[0]);
        if (update) wpc = trace_forward(tr, p_pos, wpc);
        SHA(0, j, g) = wA[0];
        SHA(1, j, g) = wC[0];
        SHA(2, j, g) = wap;
        SHA(3, j, g) = wpc;
        if (!DEFER)
        {
          sh_ppos[g * 16 + j] = p_pos;
          sh_fbflag[j * 4 + g] = 0u;
        }
        sh_apos[g * 16 + j] = ap_pos;
        wave_sync();
        double sums[4];
        { // lane r sums row r in the reference's order (linear.cpp:147-151); results shared through LDS
          const int row = j & 3;
          double sum = 0;
#pragma unroll
          for (int k = 0; k < 16; ++k) sum += SHA(row, k, g);
          sh_res[g * 16 + j] = sum / 16;
        }
        wave_sync();
#pragma unroll
        for (int row = 0; row < 4; ++row) sums[row] = sh_res[g * 16 + row];
        const double u_next = clampd(sums[0], a_min, a_max);           // actor at s'
        const double v_next = clampd(sums[1], up.out_min, up.out_max); // critic at s'
        const double u_prev = clampd(sums[2], a_min, a_max);           // actor at s (before its update)
        const double v_prev = clampd(sums[3], up.out_min, up.out_max); // critic at s

        // -------- policy (ActionPolicy::act, action.cpp:127-158)
        double a_next = 0;
        if (has_next)
        {
          double out = u_next;
          if (!test)
          {
            if (time == 0) ac_noise = 0;
            if (time == 0.) ac_decay = fmax(ac_decay * N.ac_decay_rate, N.ac_decay_min);
            if (N.sigma != 0)
            { // Rand::getNormal(0, decay*sigma): two thread-local draws (utils.h:120-125)
              TL = lcg_next(TL);
              const double U1 = lcg_double(TL);
              TL = lcg_next(TL);
              const double U2 = lcg_double(TL);
              const double sg = ac_decay * N.sigma;
              const double nrm = __builtin_sqrt(-2 * plog(U1)) * pcos(2 * GRLX_PI * U2) * sg + 0.;
              ac_noise = (1 - N.theta) * ac_noise + nrm;
              out += ac_noise;
            }
          }
          a_next = fmin(fmax(out, N.action_min), N.action_max);
        }

        // -------- predictor (ActionACPredictor::update, ac.cpp:72-110)
        double delta = 0;
        if (update)
        {
          // critic: TDPredictor::criticize (td.cpp:68-91)
          double target = reward;
          if (has_next) target += N.gamma * v_next;
          delta = target - v_prev;
          if (DEFER)
          { // applied on the next pass, after that pass's loads are in flight
            pd = true;
            pd_dW = N.alpha * (target - v_prev);
            pd_dT = N.alpha * delta;
            pd_pos = p_pos;
            status |= (p_pos == kInvalidPos) ? ST_BAD_POS : 0u;      // assert: an update always follows an action taken
            pd_sh = p_sh;
            pd_wp = wpc;
          }
          else
          {
            up.dW = N.alpha * (target - v_prev);
            up.dT = N.alpha * delta;
            Evicted ev_unused;
            td_update_lane<false>(tr, tabC, up, p_pos, p_sh, wpc, g, j, sh_ppos, sh_fb, sh_fbflag, status, ev_unused);
          }
          // actor
          if (N.ac_update_method == 0 || delta > 0)
          {
            double du = action - u_prev;                          // transition.prev_action - u
            if (N.ac_update_method == 0) du = delta * du;
            if (N.ac_step_limit >= 0) du = fmin(fmax(du, -N.ac_step_limit), N.ac_step_limit);
            const double target_u = u_prev + du;
            const double dA = N.actor_alpha * (target_u - u_prev);    // LinearRepresentation::write
            uint32_t cpa = 1;                                         // a slot that occurs twice is updated twice
            const uint32_t amask = (uint32_t)((__ballot(ap_sh) >> (16 * g)) & 0xFFFFull);
            for (uint32_t mm = amask; mm != 0u; mm &= mm - 1u)
            {
              const int k = __builtin_ctz(mm);
              if (k != j && sh_apos[g * 16 + k] == ap_pos) cpa++;
            }
            double nv = wap;
            for (uint32_t c = 0; c < cpa; ++c) nv = a_limit ? clampd(nv + dA, a_min, a_max) : nv + dA;
            value_store(tabA, ap_pos, nv);
            if (has_next && posA[0] == ap_pos) wA[0] = nv;          // the next step updates the same slot: it continues from this value
          }
        }

        // -------- tap
        if (!DEFER && tapped && (!first || P.tap_starts))
        {
          uint32_t n = *P.tap_count;
          if (n < (uint32_t)P.tap_capacity)
          {
            grlx_tap *tp = &P.taps[n];
            tp->p_idx[j] = update ? p_slot : 0u;
            tp->p_idx[16 + j] = update ? ap_slot : 0u;
            if (j == 0)
            {
              tp->test = test;
              tp->action_index = 0;
              tp->terminal = first ? -1 : terminal;
              tp->trace_len = tr.len;
              for (int i = 0; i < GRLX_MAX_DIMS; ++i) tp->obs[i] = (i < D) ? obs[i] : 0.;
              tp->action = has_next ? a_next : action;
              tp->reward = reward;
              for (int i = 0; i < GRLX_MAX_STATE; ++i) tp->state[i] = (i < S) ? x[i] : 0.;
              tp->delta = delta;
              for (int a = 0; a < kMaxActions; ++a) tp->q[a] = 0.;
              tp->q[0] = has_next ? u_next : 0.;
            }
          }
          wave_sync();
          if (j == 0) *P.tap_count = n + 1u;
        }

        if (!first)
        {
          if (test) test_steps++;
          else ss++;
        }
        if (has_next)
        {
          action = a_next;
          ap_pos = posA[0]; ap_slot = slotA[0]; ap_sh = shA[0];
          wap_seen = wA[0];
          if (need_critic) { p_pos = posC[0]; p_slot = slotC[0]; p_sh = shC[0]; wpc_seen = wC[0]; }
        }
        if (!first && terminal) running = false;
        first = false;
      }
      // the eviction held back by the deferred update: nothing reads the table before the next pass
      if (DEFER && ev.pos != kInvalidPos) value_store(tabC, ev.pos, ev.val);
    }

    // end of a learning trial: make the table current (test trials and the host read it); the
    // entries themselves stay -- the reference never clears the critic's trace
    if (!test) trace_flush(tr, tabC, false);
    }   // episodes of the trial

    if (act && (ti >= 0 ? test : 1))
    {
      if (rows < (uint32_t)P.max_rows)
      {
        if (j == 0)
        {
          size_t at = (size_t)rows * (size_t)P.n_replicas + (size_t)r;
          P.row_reward[at] = total_reward / (double)subtrials;              // online_learning.cpp:224-225
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

  // persist the critic's trace (weights are in the table already)
  trace_flush(tr, tabC, false);
  if (live && up.use_trace)
  {
#pragma unroll
    for (int e = 0; e < kMaxTrace; ++e)
    {
      ts[e * 2] = tr.pos[e];
      ts[e * 2 + 1] = (trace_cnt(tr, e) & 0xFFFFu) | (((tr.wt >> e) & 1u) << 16);
    }
  }
  uint32_t ic = ins_c, ia = ins_a;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) { ic += __shfl_xor(ic, off, 16); ia += __shfl_xor(ia, off, 16); }
  if (live && j == 0)
  {
#pragma unroll
    for (int i = 0; i < S; ++i) RS.x[i] = x[i];
    RS.G = G;
    RS.TL = TL;
    RS.ac_decay = ac_decay;
    RS.ac_noise = ac_noise;
    RS.tt = tt;
    RS.ss = ss;
    RS.test_steps = test_steps;
    RS.n_slots[0] += ic;
    RS.n_slots[1] += ia;
    RS.rows = rows;
    RS.tr_len = tr.len;
    RS.tr_total = tr.total;
  }
  uint32_t st = status;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) st |= __shfl_xor(st, off, 16);
  if (live && j == 0) RS.status = st;
}


} // namespace grlx
