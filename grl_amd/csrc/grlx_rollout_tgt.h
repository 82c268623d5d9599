// grlx_rollout_tgt.h -- rollout with a TARGET NETWORK on the Q table (ParameterizedRepresentation `interval` / `tau`,
// representation.h:161-306): SARSA and Q-learning read their targets from a second copy of the parameters that is
// synchronised every `interval` LinearRepresentation::update calls (sarsa.cpp:107, advantage.cpp:88, linear.cpp:267).
// Part of the single translation unit grlx_kernels.hip (included there, in order; not self-contained).
//
// A plain path in the manner of rollout_acc_kernel: nothing is cached -- the trace holds table positions, every update
// is a read-modify-write of the table in the reference's order -- because a synchronisation falls BETWEEN two update
// calls of one step (the count advances once per call: the write of p, then one call per trace entry) and must see the
// table exactly as those calls left it.
//   * the target's parameters live in a parallel array tvals[position] of the sparse table;
//   * synchronise() = tau * params + (1 - tau) * target over the WHOLE parameter vector: here a pass of the replica's 16
//     lanes over its sparse table; a slot that is not in the table has never been written, so its parameter is still its
//     initial draw p0 and its target value follows t <- tau*p0 + (1-tau)*t once per synchronisation from the target's OWN
//     initial draw (the target is instantiated, and draws, BEFORE the main table: representation.h:186-190) -- evaluated
//     when the slot first gets an entry (K-fold recurrence, K = synchronisations so far; tau = 0 or 1: no recurrence);
//   * replacing trace (ssub across tilings through LDS) or no trace.
#pragma once

namespace grlx {

constexpr unsigned long long kTvalUnset = 0xFFFFFFFFFFFFFFFFull;       // tvals[] entry not materialised yet

// target value of a slot that has had no entry during the first K synchronisations (its parameter is its initial draw)
__device__ inline double target_from_init(const DevParams &P, const ReplicaState &rs, uint32_t slot, uint32_t K)
{
  const double tau = P.target_tau;
  const double p0 = initial_weight(rs, 0, P.lin, slot);                 // the slot's parameter: its draw, or a loaded image's value
  if (tau == 0.) return p0;                                             // setParams(params()) at reset, at a load and ever after
  if (rs.target_base)
  { // loaded (grlx_load_weights): the target of every slot right after the load's synchronize(), then the recurrence with the image's value
    double tl = rs.target_base[slot];
    if (tau == 1.) return tl;
    for (uint32_t k = rs.syncs_base; k < K; ++k) tl = tau * p0 + (1 - tau) * tl;
    return tl;
  }
  LinearParams first = P.lin;
  first.draws_before = 0;                                               // the target table's own draws come first
  double t = lazy_weight(rs.TL0, first, slot);
  t = tau * p0 + (1 - tau) * t;                                         // synchronize() at the end of reset (linear.cpp:122)
  if (tau == 1.) return t;                                              // (1 - tau) * t == +-0: every later round returns the same bits
  for (uint32_t k = 0; k < K; ++k) t = tau * p0 + (1 - tau) * t;
  return t;
}

__device__ __forceinline__ double tval_get(const DevParams &P, const ReplicaState &rs, double *tv, uint32_t pos, uint32_t slot, uint32_t K)
{
  double t = tv[pos];
  if ((unsigned long long)__double_as_longlong(t) == kTvalUnset)
  {
    t = target_from_init(P, rs, slot, K);
    tv[pos] = t;
  }
  return t;
}

// ParameterizedRepresentation::synchronize for the replicas flagged in `todo` (wave lanes of those 16-lane groups):
// every entry of the sparse table; K = synchronisations BEFORE this one
__device__ __noinline__ void target_sync_pass(const DevParams &P, const Table &tab, const ReplicaState &rs, double *tv, uint32_t K, bool todo, int j)
{
  if (!todo) return;
  const double tau = P.target_tau;
  const uint32_t nb = tab.bmask + 1u;
  for (uint32_t b = (uint32_t)j; b < nb; b += 16u)
  {
    const BucketRegs br = bucket_load(tab, b);
    const uint32_t keys[4] = {br.k.x, br.k.y, br.k.z, br.k.w};
#pragma unroll
    for (int way = 0; way < 4; ++way)
    {
      const uint32_t kk = keys[way] & kKeyMask;
      if (kk == 0u) continue;
      const uint32_t pos = (b << 2) | (uint32_t)way;
      const double t = tval_get(P, rs, tv, pos, kk - 1u, K);
      tv[pos] = (tau != 0.) ? tau * br.v[way] + (1 - tau) * t : br.v[way];
    }
  }
}

// SAFE: projector/tile_coding:safe = 1 (tile_coding.h:116-151).  A slot can be CLAIMED by the full 32-bit hash sum of a
// projection; a projection with another hash sum that lands on a claimed slot walks on to the next slot that is free or
// its own.  Single projections claim -- project(prev_obs, prev_action), SARSA's project(obs, action), the critique's
// project(prev_obs, action) -- the policy's batch projections do not (tile_coding.h:62-73).  The claim lives in the `aux`
// word of the sparse entry (hash sum + 1; 0 = free, the reference's -1).  Claims of one projection are made in tiling
// order: a lane finalises its slot only when every lower tiling of its replica has, and not on a slot a lower tiling is
// taking with another hash sum in the same round.
template <int ENV, int NA, bool TARGET, bool SAFE>
__global__ __launch_bounds__(64) void rollout_tgt_kernel(DevParams P, int n_trials)
{
  constexpr int S = Env<ENV>::S, D = Env<ENV>::D, T = kLanesPerReplica;
  constexpr int NROWS = 2 * NA + 2;                  // Q(s', .), Q(s, a), Q_target(s', .), SAFE: Q_target at the claimed project(s', a')
  static_assert(NROWS <= 16, "one lane per row of sums");
  __shared__ double   sh_w[NROWS * 16 * 4];
  __shared__ uint32_t sh_mb[4 * NA * 16];
  __shared__ uint32_t sh_ms[4 * NA * 16];
  __shared__ uint32_t sh_mail[4];
  __shared__ double   sh_res[4 * 16];
  __shared__ uint32_t sh_ppos[4 * 16];
  __shared__ uint32_t sh_pii[4 * 16], sh_ph[4 * 16];      // SAFE: tentative slot and hash sum of every tiling of a projection
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
  uint32_t status = RS.status, rows = RS.rows, inserted = 0;
  int64_t sync_count = RS.sync_count;
  uint32_t K = RS.syncs;

  const Table tab = table_of(P, 0, r);
  double *tv = TARGET ? P.tvals + ((size_t)r << P.logC) : nullptr;
  const double out_min = P.lin.out_min, out_max = P.lin.out_max;
  const bool limit = P.lin.limit != 0;
  const bool use_trace = P.trace_kind == GRLX_TRACE_REPLACING;
  const double ee = P.gl, cut = 0.01;

  double acts[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) acts[a] = P.actions[a];
  uint32_t key_act[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a)
    key_act[a] = in_reg(murmur_key(tile_coord<T>(P.tile, D, tile_quant(P.tile, D, P.actions[a]), j)));
  const uint32_t key_j = in_reg(murmur_key(j));

  // the trace of this lane's tiling: table positions, newest first (kInvalidPos: index removed by ssub); bit e of tsh:
  // entry e is a slot shared between tilings.  tlen is the reference's entry count (equal in all lanes).
  uint32_t tpos[kMaxTrace];
#pragma unroll
  for (int e = 0; e < kMaxTrace; ++e) tpos[e] = kInvalidPos;
  uint32_t tsh = 0;
  int tlen = 0;
  double ttotal = 1.;

  auto add_to = [&](uint32_t pos, double d) {       // LinearRepresentation::update of one index (linear.cpp:198-216)
    const double v = value_load(tab, pos) + d;
    value_store(tab, pos, limit ? clampd(v, out_min, out_max) : v);
  };
  // checkSynchronize after one update() call (linear.cpp:267, representation.h:298-305)
  auto count_call = [&](bool doit) {
    if (!TARGET) return;
    bool fire = false;
    if (doit)
    {
      sync_count++;
      fire = sync_count >= (int64_t)P.target_interval;
    }
    if (rarely(__any(fire)))
    {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");           // the pass reads what the calls so far have stored
      target_sync_pass(P, tab, RS, tv, K, fire, j);
      if (fire) { sync_count = 0; K++; }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
  };

  // SAFE: getFeatureLocation (tile_coding.h:116-151) for the lanes of the groups flagged `active` (group-uniform):
  // walk from h % memory over slots claimed by other hash sums; with `claim`, take the slot in tiling order.
  const uint32_t mem_u = (uint32_t)P.tile.memory;
  auto probe = [&](bool active, uint32_t h, bool claim, uint32_t &slot_out, uint32_t &pos_out, double &w_out, bool &sh_out, auto on_share) {
    uint32_t ii = h % mem_u;
    bool pend = active;
    uint32_t pos1[1] = {kInvalidPos};
    double w1[1] = {0};
    bool sh1[1] = {false};
    for (int round = 0; round < 4096; ++round)
    {
      if (!__any(pend)) break;
      // walk: every lane of an active group looks its current slot up (finished lanes repeat theirs: idempotent)
      bool moved = false;
      if (active)
      {
        uint32_t s1[1] = {ii};
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        table_get<1>(tab, P.lin, RS, 0, s1, pos1, w1, sh1, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump, status, inserted, on_share);
        const uint32_t aux = tab.base[(pos1[0] >> 2) & tab.bmask].aux[pos1[0] & 3u];
        moved = pend && aux != 0u && aux != h + 1u;                       // claimed by another hash sum
        if (moved) ii = (ii + 1u >= mem_u) ? 0u : ii + 1u;
      }
      if (__any(moved)) continue;                                          // someone is still walking
      if (!claim) { pend = false; continue; }
      // every pending lane stands on a slot that is free or its own: finalise in tiling order
      if (active) { sh_pii[g * 16 + j] = pend ? ii : 0xFFFFFFFFu; sh_ph[g * 16 + j] = h; }
      wave_sync();
      bool lose = false;
      if (pend)
        for (int k = 0; k < j; ++k)
          lose = lose || (sh_pii[g * 16 + k] == ii && sh_ph[g * 16 + k] != h);
      const uint32_t losers = (uint32_t)((__ballot(pend && lose) >> (16 * g)) & 0xFFFFull);
      const bool fin = pend && !lose && (losers & ((1u << j) - 1u)) == 0u;   // no lower tiling still has to move
      if (fin)
      {
        tab.base[(pos1[0] >> 2) & tab.bmask].aux[pos1[0] & 3u] = h + 1u;
        pend = false;
      }
      wave_sync();
      // a loser's slot is being claimed by a lower tiling with another hash sum: it walks on next round
    }
    if (pend) status |= ST_TABLE_FULL;
    if (active)
    {
      slot_out = ii;
      pos_out = pos1[0];
      w_out = w1[0];
      sh_out = sh1[0];
    }
  };

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
    uint32_t p_pos = kInvalidPos, p_slot = 0, hp = 0, hpm_prev = 0;
    bool p_sh = false;
    if (!test)
    { // TDAgent::start -> predictor->finalize() -> trace_->clear()
#pragma unroll
      for (int e = 0; e < kMaxTrace; ++e) tpos[e] = kInvalidPos;
      tsh = 0; tlen = 0; ttotal = 1.;
    }
    bool first = true;

    for (;;)
    {
      if (!__any(running)) break;
      bool has_next = false, update = false;
      uint32_t slot[NA], pos[NA], hfull[NA];
      double w[NA], tw[NA];
      bool sh[NA];
#pragma unroll
      for (int a = 0; a < NA; ++a) { slot[a] = 0; pos[a] = kInvalidPos; hfull[a] = 0; w[a] = 0; tw[a] = 0; sh[a] = false; }
      double q[NA], qt[NA], qsa = 0;
      int a_next = 0, mai = 0, man = 1;
      double best = 0, delta = 0, dW = 0, dT = 0;
      uint32_t hpm = 0;
      auto share_event = [&](uint32_t mp) {               // a slot became shared between tilings: its entries are updated serially from now on
        if (p_pos == mp) p_sh = true;
#pragma unroll
        for (int e = 0; e < kMaxTrace; ++e) tsh |= (tpos[e] == mp) ? (1u << e) : 0u;
      };
      // the target network's value of a looked-up slot (TARGET), else the main table's; shared slots one lane at a time
      // (same value either way; the serialisation only keeps the first store from racing a second lane's read)
      auto target_value = [&](bool active, uint32_t pa, uint32_t sa, bool shared, double main_value) -> double {
        if (!TARGET) return main_value;
        double got = 0;
        if (active && !shared) got = tval_get(P, RS, tv, pa, sa, K);
        if (rarely(__any(active && shared))) serial_lanes(active && shared, [&]() { got = tval_get(P, RS, tv, pa, sa, K); });
        return got;
      };
      if (running)
      {
        if (!first)
        {
          env_step<ENV>(P, x, action, obs, reward, terminal, status);
          total_reward += reward;
          time += 1;
        }
        has_next = first || terminal != 2;
        update = !first && !test;
        if (has_next)
        {
          uint32_t hpre = 449u ^ (uint32_t)(D + 2);
#pragma unroll
          for (int i = 0; i < D; ++i)
            hpre = murmur_mix(hpre, tile_coord<T>(P.tile, i, tile_quant(P.tile, i, obs[i]), j));
          hpm = hpre * 0x5bd1e995u;
#pragma unroll
          for (int a = 0; a < NA; ++a)
          {
            uint32_t h = murmur_absorb(hpm ^ key_act[a], key_j);
            hfull[a] = murmur_final(h);
            const uint32_t mem = (uint32_t)P.tile.memory;
            slot[a] = ((mem & (mem - 1u)) == 0u) ? (hfull[a] & (mem - 1u)) : (hfull[a] % mem);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      }
      // -------- policy: Q(s', .) -- the batch projections of QPolicy::values: no claims under safe = 1; under safe = 2 each
      // variant claims like a single projection, one variant after the other (tile_coding.h:67-73)
      const bool claim_batch = P.tile_safe > 1;
      if (SAFE)
      {
#pragma unroll
        for (int a = 0; a < NA; ++a) probe(running && has_next, hfull[a], claim_batch, slot[a], pos[a], w[a], sh[a], share_event);
      }
      else if (running && has_next)
        table_get<NA>(tab, P.lin, RS, 0, slot, pos, w, sh, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump, status, inserted, share_event);
      // -------- criticize: p = project(prev_obs, prev_action), a single projection (claims under safe = 1)
      if (SAFE)
      {
        uint32_t ps = 0;
        double unused = 0;
        probe(running && update, hp, true, ps, p_pos, unused, p_sh, share_event);
        if (running && update) p_slot = ps;
      }
      // the rows the TARGET is formed from.  Q-learning projects the variants of obs AGAIN inside criticize
      // (advantage.cpp:85-86), i.e. after p's claim, which may have moved one of them off the slot the policy saw.
      if (SAFE && P.agent != GRLX_AGENT_SARSA)
      {
#pragma unroll
        for (int a = 0; a < NA; ++a)
        {
          const bool act = running && update && has_next;
          uint32_t s2 = 0, p2 = kInvalidPos;
          double w2 = 0;
          bool sh2 = false;
          probe(act, hfull[a], claim_batch, s2, p2, w2, sh2, share_event);
          tw[a] = target_value(act, p2, s2, sh2, w2);
        }
      }
      else
      {
#pragma unroll
        for (int a = 0; a < NA; ++a) tw[a] = target_value(running && has_next, pos[a], slot[a], sh[a], w[a]);
      }
      if (running)
      {
        double wp = 0;
        if (update) wp = value_load(tab, p_pos);
#pragma unroll
        for (int a = 0; a < NA; ++a) { SHW(a, j, g) = w[a]; SHW(NA + 1 + a, j, g) = tw[a]; }
        SHW(NA, j, g) = wp;
      }
      wave_sync();
      if (running)
      {
        const int row = (j < NROWS - 1) ? j : 0;
        double sum = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += SHW(row, k, g);
        sh_res[g * 16 + j] = sum / 16;
      }
      wave_sync();
      if (running)
      {
#pragma unroll
        for (int a = 0; a < NA; ++a)
        {
          q[a] = has_next ? clampd(sh_res[g * 16 + a], out_min, out_max) : 0.;
          qt[a] = has_next ? clampd(sh_res[g * 16 + NA + 1 + a], out_min, out_max) : 0.;
        }
        qsa = clampd(sh_res[g * 16 + NA], out_min, out_max);

        // sampler (greedy.cpp:63-86, 144-218): the POLICY reads the main table
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
      }
      // SARSA under safe = 1: the target reads project(obs, action), a SINGLE projection -- it claims, and a claim made a
      // moment ago (p's) may have moved it off the slot the policy's batch projection saw
      double q_claimed = 0;
      if (SAFE && P.agent == GRLX_AGENT_SARSA)
      {
        const bool act = running && update && has_next;
        uint32_t ns = 0, npos = kInvalidPos;
        double nw = 0;
        bool nsh = false;
        probe(act, pick<uint32_t, NA>(hfull, a_next), true, ns, npos, nw, nsh, share_event);
        const double ntw = target_value(act, npos, ns, nsh, nw);
        if (act) SHW(NROWS - 1, j, g) = ntw;
        wave_sync();
        if (act)
        {
          double sum = 0;
#pragma unroll
          for (int k = 0; k < 16; ++k) sum += SHW(NROWS - 1, k, g);
          q_claimed = clampd(sum / 16, out_min, out_max);
        }
        wave_sync();
      }
      if (running && update)
      { // the TARGET comes from the target network when there is one (sarsa.cpp:107 / advantage.cpp:88)
        double target = reward;
        if (has_next)
        {
          if (P.agent == GRLX_AGENT_SARSA)
            target += P.gamma * (SAFE ? q_claimed : pick<double, NA>(qt, a_next));
          else
          {
            double v = -__builtin_inf();
#pragma unroll
            for (int kk = 0; kk < NA; ++kk) v = fmax(v, qt[kk]);
            target += P.gamma * v;
          }
        }
        delta = target - qsa;
        dW = P.alpha * (target - qsa);
        dT = P.alpha * delta;
      }
      // Q-learning's critique reads project(prev_obs, action) BEFORE the write (advantage.cpp:95-97); SARSA's after the trace
      // (sarsa.cpp:120-121): a single projection -- it claims its slots; its value is not used by agent/td
      auto critique_claim = [&]() {
        if (!SAFE) return;
        const bool act = running && update && has_next;
        uint32_t cs = 0, cpos = kInvalidPos;
        double cw = 0;
        bool csh = false;
        const uint32_t hc = murmur_final(murmur_absorb(hpm_prev ^ pick<uint32_t, NA>(key_act, a_next), key_j));
        probe(act, hc, true, cs, cpos, cw, csh, share_event);
      };
      if (P.agent != GRLX_AGENT_SARSA) critique_claim();

      // -------- the update calls, one after the other, each followed by checkSynchronize()
      if (__any(update))
      {
        // call 1: write(p, target, alpha) -- every index of p, in tiling order where tilings share the slot
        if (update && !p_sh) add_to(p_pos, dW);
        if (rarely(__any(update && p_sh))) serial_lanes(update && p_sh, [&]() { add_to(p_pos, dW); });
        count_call(update);
        if (use_trace)
        {
          // calls 2..: update(trace, alpha*delta, e), newest entry first while its weight exceeds 0.001
          double weight = 1.;
#pragma unroll
          for (int e = 0; e < kMaxTrace; ++e)
          {
            const bool go = update && e < tlen && weight > 0.001;
            if (__any(go))
            {
              const double de = weight * dT * ee;
              const bool valid = go && tpos[e] != kInvalidPos;
              const bool shared = ((tsh >> e) & 1u) != 0u;
              const uint32_t at = tpos[e];
              if (valid && !shared) add_to(at, de);
              if (rarely(__any(valid && shared))) serial_lanes(valid && shared, [&]() { add_to(at, de); });
              count_call(go);
            }
            weight *= ee;
          }
          // trace_->add(p, e) (trace.h:215-234): ssub against EVERY index of p (other tilings' through LDS), push, pop
          if (update) sh_ppos[g * 16 + j] = p_sh ? p_pos : kInvalidPos;
          wave_sync();
          if (update)
          {
            if (ee < cut) { tlen = 0; ttotal = 1.; tsh = 0; }
            const uint32_t shm = (uint32_t)((__ballot(update && p_sh) >> (16 * g)) & 0xFFFFull);
#pragma unroll
            for (int e = 0; e < kMaxTrace; ++e)
            {
              bool hit = tpos[e] == p_pos;
              for (uint32_t mm = shm; mm != 0u; mm &= mm - 1u)
                hit = hit || (tpos[e] != kInvalidPos && tpos[e] == sh_ppos[g * 16 + __builtin_ctz(mm)]);
              if (e < tlen && hit) { tpos[e] = kInvalidPos; tsh &= ~(1u << e); }
            }
            if (tlen >= kMaxTrace) status |= ST_TRACE_OVERFLOW;       // cannot happen: validated at create
#pragma unroll
            for (int e = kMaxTrace - 1; e > 0; --e) tpos[e] = tpos[e - 1];
            tsh = (tsh << 1) & ((1u << kMaxTrace) - 1u);
            tpos[0] = p_pos;
            if (p_sh) tsh |= 1u;
            tlen = (tlen < kMaxTrace) ? tlen + 1 : kMaxTrace;
            ttotal *= ee;
            while (ttotal < cut && tlen > 1)
            {
              ttotal /= ee;
              tlen--;
            }
#pragma unroll
            for (int e = 0; e < kMaxTrace; ++e)
              if (e >= tlen) { tpos[e] = kInvalidPos; tsh &= ~(1u << e); }
          }
          wave_sync();
        }
      }
      if (P.agent == GRLX_AGENT_SARSA) critique_claim();

      if (running)
      {
        // -------- tap
        if (tapped && (!first || P.tap_starts))
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
              tp->trace_len = tlen;
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
          p_pos = pick<uint32_t, NA>(pos, a_next);
          p_slot = pick<uint32_t, NA>(slot, a_next);
          p_sh = pick<bool, NA>(sh, a_next);
          hp = pick<uint32_t, NA>(hfull, a_next);          // SAFE: p is projected again, with a claim, when it is criticized
          hpm_prev = hpm;
        }
        if (!first && terminal) running = false;
        first = false;
      }
    }

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
    RS.sync_count = sync_count;
    RS.syncs = K;
  }
  uint32_t st = status;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) st |= __shfl_xor(st, off, 16);
  if (live && j == 0) RS.status = st;
}

hipError_t launch_rollout_tgt(const DevParams &P, int n_trials, hipStream_t stream, int *variant)
{
  if (variant) *variant = GRLX_KERNEL_IN_PLACE;
  int waves = (P.n_replicas + kReplicasPerWave - 1) / kReplicasPerWave;
  const bool target = P.target_interval > 0, safe = P.tile_safe != 0;
#define GRLX_LAUNCH_PLAIN(ENVID, TG, SF)                                                                              \
  if (P.env == ENVID && P.A == 3 && target == TG && safe == SF)                                                     \
  {                                                                                                                 \
    hipLaunchKernelGGL((rollout_tgt_kernel<ENVID, 3, TG, SF>), dim3(waves), dim3(64), 0, stream, P, n_trials);      \
    return hipGetLastError();                                                                                       \
  }
  GRLX_LAUNCH_PLAIN(GRLX_ENV_PENDULUM, true, false)
  GRLX_LAUNCH_PLAIN(GRLX_ENV_PENDULUM, false, true)
  GRLX_LAUNCH_PLAIN(GRLX_ENV_PENDULUM, true, true)
  GRLX_LAUNCH_PLAIN(GRLX_ENV_ACROBOT, true, false)
  GRLX_LAUNCH_PLAIN(GRLX_ENV_ACROBOT, false, true)
  // round 3: the other two environments of the path (a discretised cart-pole; cfg/compass_walker/qlearning_walk.yaml)
  GRLX_LAUNCH_PLAIN(GRLX_ENV_CART_POLE, true, false)
  GRLX_LAUNCH_PLAIN(GRLX_ENV_CART_POLE, false, true)
  GRLX_LAUNCH_PLAIN(GRLX_ENV_COMPASS_WALKER, true, false)
  GRLX_LAUNCH_PLAIN(GRLX_ENV_COMPASS_WALKER, false, true)
#undef GRLX_LAUNCH_PLAIN
  return hipErrorInvalidValue;
}

// current target-network value of reference slots (no insertion)
__global__ void get_target_weights_kernel(DevParams P, int replica, const uint32_t *slots, int n, double *out)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Table tab = table_of(P, 0, replica);
  const ReplicaState &rs = P.states[replica];
  const double *tv = P.tvals + ((size_t)replica << P.logC);
  const uint32_t slot = slots[i];
  uint32_t b = table_home(tab, slot);
  double v = target_from_init(P, rs, slot, rs.syncs);
  for (int it = 0; it < kMaxProbe; ++it)
  {
    const BucketRegs br = bucket_load(tab, b);
    uint32_t empty;
    const int way = bucket_find(br.k, slot, empty);
    if (way >= 0)
    {
      const double t = tv[(b << 2) | (uint32_t)way];
      if ((unsigned long long)__double_as_longlong(t) != kTvalUnset) v = t;
      break;
    }
    if (empty != 0u) break;
    b = (b + 1u) & tab.bmask;
  }
  out[i] = v;
}

// {action: load} into a representation with a target network (representation.h:231-263): setParams(image), then synchronize() --
// target <- tau * image + (1 - tau) * target over the WHOLE parameter vector.  out[slot] = that value for every slot of one replica, from the
// target's value NOW (its table entry, or what target_from_init gives an untouched slot); the caller then empties the replica's table.
__global__ void target_after_load_kernel(DevParams P, int replica, const double *image, double *out)
{
  const uint32_t n = (uint32_t)P.tile.memory;
  const Table tab = table_of(P, 0, replica);
  const ReplicaState &rs = P.states[replica];
  const double *tv = P.tvals + ((size_t)replica << P.logC);
  const double tau = P.target_tau;
  for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n; slot += gridDim.x * blockDim.x)
  {
    if (tau == 0.) { out[slot] = image[slot]; continue; }
    uint32_t b = table_home(tab, slot);
    double v = target_from_init(P, rs, slot, rs.syncs);
    for (int it = 0; it < kMaxProbe; ++it)
    {
      const BucketRegs br = bucket_load(tab, b);
      uint32_t empty;
      const int way = bucket_find(br.k, slot, empty);
      if (way >= 0)
      {
        const double t = tv[(b << 2) | (uint32_t)way];
        if ((unsigned long long)__double_as_longlong(t) != kTvalUnset) v = t;
        break;
      }
      if (empty != 0u) break;
      b = (b + 1u) & tab.bmask;
    }
    out[slot] = tau * image[slot] + (1 - tau) * v;
  }
}

hipError_t launch_target_after_load(const DevParams &P, int replica, const double *image_dev, double *out_dev, hipStream_t stream)
{
  hipLaunchKernelGGL(target_after_load_kernel, dim3(2048), dim3(256), 0, stream, P, replica, image_dev, out_dev);
  return hipGetLastError();
}

hipError_t launch_get_target_weights(const DevParams &P, int replica, const uint32_t *slots_dev, int n, double *out_dev, hipStream_t stream)
{
  int blocks = (n + 255) / 256;
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(get_target_weights_kernel, dim3(blocks), dim3(256), 0, stream, P, replica, slots_dev, n, out_dev);
  return hipGetLastError();
}

} // namespace grlx
