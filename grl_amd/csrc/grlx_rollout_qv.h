// grlx_rollout_qv.h -- QV-learning rollout (cfg/pendulum/qv_tc.yaml): rollout_qv_kernel and its launcher.
// Part of the single translation unit grlx_kernels.hip (included there, in order; not self-contained).
#pragma once

namespace grlx {

// ----------------------------------------------------------- QV rollout ---
// agent/td { policy: mapping/policy/discrete/value/q, predictor: predictor/critic/qv } (cfg/pendulum/qv_tc.yaml).
// Table 0 = Q(s,a) read by the epsilon-greedy policy and written without a trace, table 1 = V(s) with the
// register trace; both move towards r + gamma V(s') (QVPredictor::criticize, qv.cpp:74-108).  Lane j = tiling j
// of both projectors.  LDS rows: Q(s',a_0..NA-1), V(s'), Q(s,a), V(s).  TD update applied in place.
template <int ENV, int NA>
__global__ __launch_bounds__(64) void rollout_qv_kernel(DevParams P, int n_trials)
{
  constexpr int S = Env<ENV>::S, D = Env<ENV>::D, T = kLanesPerReplica;
  constexpr int NROWS = NA + 3, RV = NA, RQP = NA + 1, RVP = NA + 2;
  __shared__ double   sh_w[NROWS * 16 * 4];
  __shared__ uint32_t sh_ppos[4 * 16];
  __shared__ uint32_t sh_qpos[4 * 16];
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
  uint32_t status = RS.status, rows = RS.rows, ins_q = 0, ins_v = 0;

  const Table tabQ = table_of(P, 0, r), tabV = table_of(P, 1, r);
  UpdateParams up;                                  // the V table's update (the one with the trace)
  up.out_min = P.lin_actor.out_min;
  up.out_max = P.lin_actor.out_max;
  up.limit = P.lin_actor.limit != 0;
  up.ee = P.gl;
  up.cut = 0.01;
  up.use_trace = P.trace_kind == GRLX_TRACE_REPLACING;
  up.dW = up.dT = 0;
  const double q_min = P.lin.out_min, q_max = P.lin.out_max;
  const bool q_limit = P.lin.limit != 0;

  double acts[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) acts[a] = P.actions[a];
  uint32_t key_act[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a)
    key_act[a] = in_reg(murmur_key(tile_coord<T>(P.tile, D, tile_quant(P.tile, D, P.actions[a]), j)));
  const uint32_t key_j = in_reg(murmur_key(j));

  TraceRegs tr;
  trace_init(tr);
  int tr_len_ref = 0;           // length as the reference reports it (its trace survives test trials)

  for (int trial = 0; trial < n_trials; ++trial)
  {
    // online_learning.cpp:154: a replica whose learning steps have reached the steps budget starts no further trial
    const bool act = live && !(P.steps_budget != 0u && (uint64_t)ss >= P.steps_budget);
    if (!__any(act)) break;
    const int ti = P.test_interval;
    const int test = (ti >= 0 && tt % (ti + 1) == ti) ? 1 : 0;
    // a test trial is test_trials greedy episodes (online_learning.cpp:161-170): each starts the environment and the agent anew, while
    // reward and time keep adding up (:202-203); a learning trial is one episode (its `time` = 0 is the sampler's moment to decay)
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
      Env<ENV>::start(P, test, TL, G, x);
      Env<ENV>::observe(P, x, obs);
    }
    double action = 0;
    int action_index = 0;
    uint32_t qp_pos = kInvalidPos, qp_slot = 0, vp_pos = kInvalidPos, vp_slot = 0;
    bool qp_sh = false, vp_sh = false;
    if (!test) tr_len_ref = 0;          // TDAgent::start -> QVPredictor::finalize -> trace_->clear()
    bool first = true;

    for (;;)
    {
      if (!__any(running)) break;
      if (running)
      {
        if (!first)
        {
          env_step<ENV>(P, x, action, obs, reward, terminal, status);
          total_reward += reward;
          time += 1;
        }
        const bool has_next = first || terminal != 2;
        const bool update = !first && !test;

        // projections of (s', a_k) for the policy and of s' for V (the latter also in test trials: unused there)
        uint32_t slotQ[NA], posQ[NA], slotV[1] = {0}, posV[1] = {kInvalidPos};
        double wQ[NA], wV[1] = {0};
        bool shQ[NA], shV[1] = {false};
#pragma unroll
        for (int a = 0; a < NA; ++a) { slotQ[a] = 0; posQ[a] = kInvalidPos; wQ[a] = 0; shQ[a] = false; }
        const bool need_v = has_next && !test;
        if (has_next)
        {
          uint32_t hpre = 449u ^ (uint32_t)(D + 2);
#pragma unroll
          for (int i = 0; i < D; ++i)
            hpre = murmur_mix(hpre, tile_coord<T>(P.tile, i, tile_quant(P.tile, i, obs[i]), j));
          const uint32_t hpm = hpre * 0x5bd1e995u;
#pragma unroll
          for (int a = 0; a < NA; ++a)
          {
            uint32_t h = murmur_absorb(hpm ^ key_act[a], key_j);
            const uint32_t hm = murmur_final(h), mem = (uint32_t)P.tile.memory;
            slotQ[a] = ((mem & (mem - 1u)) == 0u) ? (hm & (mem - 1u)) : (hm % mem);
          }
          slotV[0] = tile_slot_obs<T>(P.tile_actor, obs, D, j);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        double wqp = 0, wvp = 0;
        if (update)
        {
          wqp = value_load(tabQ, qp_pos);                // Q weights of project(s, a): the table is always current
          wvp = value_load(tabV, vp_pos);                // V weights of project(s), as stored
        }
        Lookup lkQ[NA], lkV[1];
        BucketRegs brQ[NA], brV[1];
        if (has_next) table_issue<NA>(tabQ, slotQ, lkQ, brQ);
        if (need_v) table_issue<1>(tabV, slotV, lkV, brV);
        if (has_next)
          table_get_finish<NA>(tabQ, P.lin, RS, 0, slotQ, lkQ, brQ, posQ, wQ, shQ, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump, status, ins_q,
                               [&](uint32_t mp) { if (qp_pos == mp) qp_sh = true; });
        if (need_v)
        {
          bool shared_event = false;
          table_get_finish<1>(tabV, P.lin_actor, RS, 1, slotV, lkV, brV, posV, wV, shV, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump, status, ins_v,
                              [&](uint32_t mp) {
                                trace_share_event(tr, tabV, mp);
                                if (vp_pos == mp) vp_sh = true;
                                shared_event = true;
                              });
          if (rarely(__any(shared_event)) && update) wvp = value_load(tabV, vp_pos);
          wV[0] = trace_forward(tr, posV[0], wV[0]);
        }
        if (update) wvp = trace_forward(tr, vp_pos, wvp);
#pragma unroll
        for (int a = 0; a < NA; ++a) SHW(a, j, g) = wQ[a];
        SHW(RV, j, g) = wV[0];
        SHW(RQP, j, g) = wqp;
        SHW(RVP, j, g) = wvp;
        sh_ppos[g * 16 + j] = vp_pos;
        sh_qpos[g * 16 + j] = qp_pos;
        sh_fbflag[j * 4 + g] = 0u;
        wave_sync();
        { // lane r sums row r in the reference's order (linear.cpp:147-151)
          const int row = (j < NROWS) ? j : 0;
          double sum = 0;
#pragma unroll
          for (int k = 0; k < 16; ++k) sum += SHW(row, k, g);
          sh_res[g * 16 + j] = sum / 16;
        }
        wave_sync();
        double q[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) q[a] = has_next ? clampd(sh_res[g * 16 + a], q_min, q_max) : 0.;
        const double v_next = clampd(sh_res[g * 16 + RV], up.out_min, up.out_max);
        const double q_prev = clampd(sh_res[g * 16 + RQP], q_min, q_max);
        const double v_prev = clampd(sh_res[g * 16 + RVP], up.out_min, up.out_max);

        // -------- policy: QPolicy::act over the Q table (q.cpp:143-155, greedy.cpp:63-86, 144-218)
        int a_next = 0;
        if (has_next)
        {
          int mai = 0, man = 1;
          double best = 0;
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

        // -------- predictor (QVPredictor::criticize, qv.cpp:74-108)
        double delta = 0;
        if (update)
        {
          double target = reward;
          if (has_next) target += P.gamma * v_next;
          delta = target - v_prev;
          { // Q update: LinearRepresentation::write(qp, target, alpha) (linear.cpp:186-216); a slot that occurs
            // twice in the projection (shared between tilings) is written twice
            const double dQ = P.alpha * (target - q_prev);
            uint32_t cpq = 1;
            const uint32_t qmask = (uint32_t)((__ballot(qp_sh) >> (16 * g)) & 0xFFFFull);
            for (uint32_t mm = qmask; mm != 0u; mm &= mm - 1u)
            {
              const int k = __builtin_ctz(mm);
              if (k != j && sh_qpos[g * 16 + k] == qp_pos) cpq++;
            }
            double nv = wqp;
            for (uint32_t c = 0; c < cpq; ++c) nv = q_limit ? clampd(nv + dQ, q_min, q_max) : nv + dQ;
            value_store(tabQ, qp_pos, nv);
          }
          // V update with the trace
          up.dW = P.beta * (target - v_prev);
          up.dT = P.beta * delta;
          Evicted ev_unused;
          td_update_lane<false>(tr, tabV, up, vp_pos, vp_sh, wvp, g, j, sh_ppos, sh_fb, sh_fbflag, status, ev_unused);
          tr_len_ref = tr.len;
        }

        // -------- tap
        if (tapped && (!first || P.tap_starts))
        {
          uint32_t n = *P.tap_count;
          if (n < (uint32_t)P.tap_capacity)
          {
            grlx_tap *tp = &P.taps[n];
            tp->p_idx[j] = update ? qp_slot : 0u;
            tp->p_idx[16 + j] = update ? vp_slot : 0u;
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

        if (!first)
        {
          if (test) test_steps++;
          else ss++;
        }
        if (has_next)
        {
          action_index = a_next;
          action = pick<double, NA>(acts, a_next);
          qp_pos = pick<uint32_t, NA>(posQ, a_next);
          qp_slot = pick<uint32_t, NA>(slotQ, a_next);
          qp_sh = pick<bool, NA>(shQ, a_next);
          if (need_v)
          {
            vp_pos = posV[0];
            vp_slot = slotV[0];
            vp_sh = shV[0];
          }
        }
        if (!first && terminal) running = false;
        first = false;
      }
    }

    // QVPredictor::finalize clears the trace at the next TDAgent::start (qv.cpp:110-116): write it back now
    if (!test) trace_flush(tr, tabV, true);

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

  uint32_t iq = ins_q, iv = ins_v;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) { iq += __shfl_xor(iq, off, 16); iv += __shfl_xor(iv, off, 16); }
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
    RS.n_slots[0] += iq;
    RS.n_slots[1] += iv;
    RS.rows = rows;
  }
  uint32_t st = status;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) st |= __shfl_xor(st, off, 16);
  if (live && j == 0) RS.status = st;
}

hipError_t launch_rollout_qv(const DevParams &P, int n_trials, hipStream_t stream, int *variant)
{
  if (variant) *variant = GRLX_KERNEL_IN_PLACE;
  int waves = (P.n_replicas + kReplicasPerWave - 1) / kReplicasPerWave;
  if (P.env == GRLX_ENV_PENDULUM && P.A == 3)
    hipLaunchKernelGGL((rollout_qv_kernel<GRLX_ENV_PENDULUM, 3>), dim3(waves), dim3(64), 0, stream, P, n_trials);
  else if (P.env == GRLX_ENV_ACROBOT && P.A == 3)
    hipLaunchKernelGGL((rollout_qv_kernel<GRLX_ENV_ACROBOT, 3>), dim3(waves), dim3(64), 0, stream, P, n_trials);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}


} // namespace grlx
