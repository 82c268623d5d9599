// grlx_rollout_acc.h -- accumulating-trace rollout: rollout_acc_kernel and its launcher.
// Part of the single translation unit grlx_kernels.hip (included there, in order; not self-contained).
#pragma once

namespace grlx {

// ------------------------------------------------- accumulating-trace rollout ---
// trace/enumerated/accumulating (trace.h:238-263): no ssub, cut 1e-4 -- up to 19 entries in which a slot may
// occur many times, every occurrence adding to the weight in the reference's order (entry-major, newest first;
// tiling-minor).  SARSA, Q-learning and Expected SARSA over one Q table; TD update in place.
// The weights of the slots in the trace are CACHED (write-back), one value per entry in LDS (sh_tv, entry-major, lane-minor):
//  * the value of a slot is authoritative at its HEAD -- its newest occurrence in this lane's trace (bit e of tnh = entry e
//    has a newer occurrence); lookups and the weight of p are forwarded from the heads, the table is stale meanwhile;
//  * one update walks the entries newest first; an entry continues from the result of the latest newer occurrence of its
//    slot (or from p's freshly written value), and every result is handed back to the newer occurrences, so that after the
//    walk all occurrences -- the head included -- hold the slot's final value: the additions of one step reach a slot one
//    after the other in entry order, as in the reference;
//  * a head that leaves the trace (dropped at the far end, trace cleared) is written to the table; the trace is flushed and
//    cleared at the end of every learning trial (TDAgent::start would clear it anyway), so test trials and the host see
//    the table current;
//  * slots shared between tilings are not cached: read-modify-write of the table, one lane at a time in tiling order; a
//    slot that becomes shared is written back first (on_share).
constexpr int kAccTrace = 20;

constexpr DevParams make_spec_pendulum_acc(int agent)
{
  DevParams P = make_spec_pendulum_tc();
  P.trace_kind = GRLX_TRACE_ACCUMULATING;
  P.agent = agent;
  P.end_stop_penalty = 1;                              // fields this kernel does not read, as grlx_config_pendulum_sarsa leaves them
  P.slope_angle = 0.004; P.initial_state_variation = 0.2; P.negative_reward = -100.0;
  P.control_step = 0.03;
  P.walker_dt = 1.0E-6 * (double)(uint64_t)30000 / 5;
  P.action_min = -3; P.action_max = 3;
  return P;
}
__device__ const DevParams d_spec_pendulum_acc = make_spec_pendulum_acc(GRLX_AGENT_SARSA);     // the agent is a template constant
template <int AGENT>
struct SpecPendulumAcc {
  __device__ static __forceinline__ int agent(const DevParams &) { return AGENT; }
  static bool matches(const DevParams &P) { constexpr DevParams C = make_spec_pendulum_acc(AGENT); return spec_numeric_equal(P, C); }
  __device__ static __forceinline__ const DevParams &numeric(const DevParams &) { return d_spec_pendulum_acc; }
};

// f() in the flagged lanes of every 16-lane group, one lane of a group at a time, ascending
template <typename F>
__device__ __forceinline__ void serial_lanes(bool flag, F f)
{
  const int lane = threadIdx.x & 63;
  unsigned long long pend = __ballot(flag);
  while (pend != 0ull)
  {
    unsigned long long sel = 0ull;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg)
    {
      unsigned long long grp = pend & (0xFFFFull << (16 * gg));
      sel |= grp & (~grp + 1ull);
    }
    if ((sel >> lane) & 1ull) f();
    pend &= ~sel;
    wave_sync();
  }
}

// SPEC: SpecNone, or the compile-time parameter block of the reference's cfg/pendulum/{sarsa,q}_tc.yaml family with the
// accumulating trace (SpecPendulumAcc; see SpecPendulumTcA in grlx_rollout.h)
template <int ENV, int NA, typename SPEC = SpecNone>
__global__ __launch_bounds__(64) void rollout_acc_kernel(DevParams P, int n_trials)
{
  const DevParams &N = SPEC::numeric(P);
  constexpr int S = Env<ENV>::S, D = Env<ENV>::D, T = kLanesPerReplica;
  __shared__ double   sh_w[(NA + 1) * 16 * 4];
  __shared__ uint32_t sh_mb[4 * NA * 16];
  __shared__ uint32_t sh_ms[4 * NA * 16];
  __shared__ uint32_t sh_mail[4];
  __shared__ double   sh_res[4 * 16];
  __shared__ uint64_t sh_jump[2048];
  __shared__ double   sh_tv[kAccTrace * 64];         // cached weight per trace entry and lane
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
  const double out_min = N.lin.out_min, out_max = N.lin.out_max;
  const bool limit = N.lin.limit != 0;
  const double ee = N.gl, cut = 0.0001;

  double acts[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) acts[a] = N.actions[a];
  uint32_t key_act[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a)
    key_act[a] = in_reg(murmur_key(tile_coord<T>(N.tile, D, tile_quant(N.tile, D, N.actions[a]), j)));
  const uint32_t key_j = in_reg(murmur_key(j));

  // the trace of this lane's tiling: positions newest first, bit e of tsh = entry e is a slot shared between tilings
  uint32_t tpos[kAccTrace];
#pragma unroll
  for (int e = 0; e < kAccTrace; ++e) tpos[e] = kInvalidPos;
  uint32_t tsh = 0, tnh = 0;
  int tlen = 0;
  double ttotal = 1.;
  int tr_len_ref = 0;
  constexpr uint32_t kAccMask = (1u << kAccTrace) - 1u;
  // entries whose weight this lane caches: valid and not shared between tilings; heads: those without a newer occurrence
  auto mine_mask = [&]() { return (tlen >= kAccTrace ? kAccMask : ((1u << tlen) - 1u)) & ~tsh; };

  auto add_to = [&](uint32_t pos, double d) {       // LinearRepresentation::update of one index (linear.cpp:198-216)
    const double v = value_load(tab, pos) + d;
    value_store(tab, pos, limit ? clampd(v, out_min, out_max) : v);
  };

  for (int trial = 0; trial < n_trials; ++trial)
  {
    // online_learning.cpp:154: a replica whose learning steps have reached the steps budget starts no further trial
    const bool act = live && !(P.steps_budget != 0u && (uint64_t)ss >= P.steps_budget);
    if (!__any(act)) break;
    const int ti = N.test_interval;
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
      Env<ENV>::start(N, test, TL, G, x);
      Env<ENV>::observe(N, x, obs);
    }
    double action = 0;
    int action_index = 0;
    uint32_t p_pos = kInvalidPos, p_slot = 0;
    bool p_sh = false;
    double wp_seen = 0;                                     // weight of p's slot as looked up (and forwarded) one pass ago
    if (!test)
    { // TDAgent::start -> predictor->finalize() -> trace_->clear() (td.cpp:54, sarsa.cpp:126-132); it is empty already
      // (flushed and cleared at the end of the previous learning trial)
#pragma unroll
      for (int e = 0; e < kAccTrace; ++e) tpos[e] = kInvalidPos;
      tsh = 0; tnh = 0; tlen = 0; ttotal = 1.; tr_len_ref = 0;
    }
    bool first = true;

    for (;;)
    {
      if (!__any(running)) break;
      if (running)
      {
        if (!first)
        {
          env_step<ENV>(N, x, action, obs, reward, terminal, status);
          total_reward += reward;
          time += 1;
        }
        const bool has_next = first || terminal != 2;
        const bool update = !first && !test;

        uint32_t slot[NA], pos[NA];
        double w[NA];
        bool sh[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) { slot[a] = 0; pos[a] = kInvalidPos; w[a] = 0; sh[a] = false; }
        if (has_next)
        {
          uint32_t hpre = 449u ^ (uint32_t)(D + 2);
#pragma unroll
          for (int i = 0; i < D; ++i)
            hpre = murmur_mix(hpre, tile_coord<T>(N.tile, i, tile_quant(N.tile, i, obs[i]), j));
          const uint32_t hpm = hpre * 0x5bd1e995u;
#pragma unroll
          for (int a = 0; a < NA; ++a)
          {
            uint32_t h = murmur_absorb(hpm ^ key_act[a], key_j);
            const uint32_t hm = murmur_final(h), mem = (uint32_t)N.tile.memory;
            slot[a] = ((mem & (mem - 1u)) == 0u) ? (hm & (mem - 1u)) : (hm % mem);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        bool shared_event = false;
        if (has_next)
          table_get<NA>(tab, N.lin, RS, 0, slot, pos, w, sh, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump, status, inserted,
                        [&](uint32_t mp) { // a slot became shared: its cached value goes to the table (the finder reads it), and
                          // every entry that refers to it is updated serially, on the table, from now on
                          if (p_pos == mp) p_sh = true;
                          shared_event = true;
                          const uint32_t heads = mine_mask() & ~tnh;
#pragma unroll
                          for (int e = 0; e < kAccTrace; ++e)
                          {
                            if (tpos[e] == mp && ((heads >> e) & 1u)) value_store(tab, mp, sh_tv[e * 64 + lane]);
                            tsh |= (tpos[e] == mp) ? (1u << e) : 0u;
                          }
                        });
        // a slot this lane had cached and looked up in the same pass was written back AFTER its bucket was loaded
        if (rarely(__any(shared_event)))
        {
#pragma unroll
          for (int a = 0; a < NA; ++a)
            if (has_next && sh[a]) w[a] = value_load(tab, pos[a]);
        }
        // the cached weights of this lane's trace; what the table returned for a cached slot is stale: the head's value counts
        double tv[kAccTrace];
        const uint32_t mine = mine_mask(), heads = mine & ~tnh;
#pragma unroll
        for (int e = 0; e < kAccTrace; ++e) tv[e] = sh_tv[e * 64 + lane];
        double wp = wp_seen;
#pragma unroll
        for (int e = kAccTrace - 1; e >= 0; --e)
        {
          const bool hd = ((heads >> e) & 1u) != 0u;
#pragma unroll
          for (int a = 0; a < NA; ++a) w[a] = (hd && tpos[e] == pos[a]) ? tv[e] : w[a];
          wp = (hd && tpos[e] == p_pos) ? tv[e] : wp;
        }
        if (rarely(__any(update && p_sh)))
          if (update && p_sh) wp = value_load(tab, p_pos);    // shared between tilings: the table is current
#pragma unroll
        for (int a = 0; a < NA; ++a) SHW(a, j, g) = w[a];
        SHW(NA, j, g) = wp;
        wave_sync();
        {
          const int row = (j <= NA) ? j : 0;
          double sum = 0;
#pragma unroll
          for (int k = 0; k < 16; ++k) sum += SHW(row, k, g);
          sh_res[g * 16 + j] = sum / 16;
        }
        wave_sync();
        double q[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) q[a] = has_next ? clampd(sh_res[g * 16 + a], out_min, out_max) : 0.;
        const double qsa = clampd(sh_res[g * 16 + NA], out_min, out_max);

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
            if (time == 0.) eps_decay = fmax(eps_decay * N.decay_rate, N.decay_min);
            S1 = lcg_next(S1);
            const double rnd = lcg_double(S1);
            if (rnd < eps_decay * N.epsilon)
            {
              G = lcg_next(G);
              a_next = (int)(lcg_long(G) % (uint32_t)NA);
            }
            else
              a_next = (man > 1) ? tie_break<NA>(q, best, man, G) : mai;
          }
        }

        // -------- predictor update (sarsa.cpp:98-124, 167-194 / advantage.cpp:71-110)
        double delta = 0;
        if (update)
        {
          double target = reward;
          if (has_next)
          {
            if (SPEC::agent(P) == GRLX_AGENT_SARSA)
              target += N.gamma * pick<double, NA>(q, a_next);
            else if (SPEC::agent(P) == GRLX_AGENT_EXPECTED_SARSA)
            {
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
          const double dW = N.alpha * (target - qsa);
          const double dT = N.alpha * delta;
          // write(p, target, alpha): every index of p, in tiling order where tilings share the slot.  p is about to enter the
          // trace as its newest entry, so its new weight stays in the cache (unless the slot is shared between tilings)
          const double newp = limit ? clampd(wp + dW, out_min, out_max) : wp + dW;
          serial_lanes(p_sh, [&]() { add_to(p_pos, dW); });
          // update(trace, alpha*delta, e): newest entry first while its weight exceeds 0.001.  tv[e] becomes the value of the
          // entry's slot after all additions processed so far: an entry starts from the latest newer occurrence of its slot
          // (tv[k] of ANY newer occurrence k: they are kept equal), or from p's new weight, or -- a head -- from its cache
          uint32_t dupp = 0;
          {
            // positions of the cached entries; an entry this lane does not cache gets a value no other entry has, so that one
            // compare decides "same slot, both cached" (the compare is made twice per pair rather than kept: 190 lane masks
            // do not fit the scalar registers)
            uint32_t tq[kAccTrace];
#pragma unroll
            for (int e = 0; e < kAccTrace; ++e) tq[e] = ((mine >> e) & 1u) ? tpos[e] : (kInvalidPos - 1u - (uint32_t)e);
            const uint32_t pq = p_sh ? kInvalidPos : p_pos;
            double weight = 1.;
#pragma unroll
            for (int e = 0; e < kAccTrace; ++e)
            {
              const bool me = ((mine >> e) & 1u) != 0u;
              const bool go = e < tlen && weight > 0.001;
              const double de = weight * dT * ee;
              const bool isp = tq[e] == pq;
              dupp |= isp ? (1u << e) : 0u;
              double base = isp ? newp : tv[e];
#pragma unroll
              for (int k = 0; k < e; ++k) base = (tq[k] == tq[e]) ? tv[k] : base;
              const double v = base + de;
              const double res = (go && me) ? (limit ? clampd(v, out_min, out_max) : v) : base;
              tv[e] = res;
#pragma unroll
              for (int k = 0; k < e; ++k) tv[k] = (tq[k] == tq[e]) ? res : tv[k];
              weight *= ee;
            }
          }
          // p's weight after the walk: that of its oldest occurrence in the trace, if it has one
          double p_fin = newp;
#pragma unroll
          for (int e = 0; e < kAccTrace; ++e) p_fin = ((dupp >> e) & 1u) ? tv[e] : p_fin;
          {
            double weight = 1.;
#pragma unroll
            for (int e = 0; e < kAccTrace; ++e)
            {
              const bool go = e < tlen && weight > 0.001;
              const bool shared = ((tsh >> e) & 1u) != 0u;
              const double d = weight * dT * ee;
              const uint32_t at = tpos[e];
              if (rarely(__any(go && shared)))
                serial_lanes(go && shared, [&]() { add_to(at, d); });
              weight *= ee;
            }
          }
          // trace_->add(p, e) (trace.h:245-262): the entries move one place down, p becomes entry 0; heads that fall off the
          // far end (or all of them, when the decay is below the cut) are written back
          const int old_len = (ee < cut) ? 0 : tlen;                 // entries that survive the `ee < cut` reset
          if (ee < cut) ttotal = 1.;
          if (old_len >= kAccTrace) status |= ST_TRACE_OVERFLOW;       // cannot happen: validated at create
          int new_len = (old_len < kAccTrace) ? old_len + 1 : kAccTrace;
          ttotal *= ee;
          while (ttotal < cut && new_len > 1)
          {
            ttotal /= ee;
            new_len--;
          }
          {
            const uint32_t nonhead = tnh | dupp;
#pragma unroll
            for (int e = 0; e < kAccTrace; ++e)
            { // old entry e moves to e + 1: it stays if e + 1 < new_len and the trace was not reset
              const bool dropped = e < tlen && !(e < old_len && e + 1 < new_len);
              const bool wb = dropped && ((mine >> e) & 1u) != 0u && ((nonhead >> e) & 1u) == 0u;
              if (rarely(__any(wb)))
                if (wb) value_store(tab, tpos[e], tv[e]);
            }
            if (ee < cut) { tsh = 0; tnh = 0; }
            else tnh = ((nonhead) << 1) & kAccMask;
          }
#pragma unroll
          for (int e = kAccTrace - 1; e > 0; --e) tpos[e] = tpos[e - 1];
          tsh = (tsh << 1) & kAccMask;
          tpos[0] = p_pos;
          if (p_sh) tsh |= 1u;
          tlen = new_len;
#pragma unroll
          for (int e = 0; e < kAccTrace; ++e)
            if (e >= tlen) { tpos[e] = kInvalidPos; tsh &= ~(1u << e); tnh &= ~(1u << e); }
          // the cache in its new order
          sh_tv[lane] = p_fin;
#pragma unroll
          for (int e = 0; e + 1 < kAccTrace; ++e) sh_tv[(e + 1) * 64 + lane] = tv[e];
          tr_len_ref = tlen;
        }

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
          p_pos = pick<uint32_t, NA>(pos, a_next);
          p_slot = pick<uint32_t, NA>(slot, a_next);
          p_sh = pick<bool, NA>(sh, a_next);
          wp_seen = pick<double, NA>(w, a_next);
        }
        if (!first && terminal) running = false;
        first = false;
      }
    }

    if (!test)
    { // end of a learning trial: the cached weights go to the table, the trace is emptied (the next TDAgent::start would)
      const uint32_t heads = mine_mask() & ~tnh;
#pragma unroll
      for (int e = 0; e < kAccTrace; ++e)
      {
        if ((heads >> e) & 1u) value_store(tab, tpos[e], sh_tv[e * 64 + lane]);
        tpos[e] = kInvalidPos;
      }
      tsh = 0; tnh = 0; tlen = 0; ttotal = 1.;
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
  }
  uint32_t st = status;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) st |= __shfl_xor(st, off, 16);
  if (live && j == 0) RS.status = st;
}

hipError_t launch_rollout_acc(const DevParams &P, int n_trials, hipStream_t stream, int *variant)
{
  if (variant) *variant = GRLX_KERNEL_IN_PLACE;
  int waves = (P.n_replicas + kReplicasPerWave - 1) / kReplicasPerWave;
  if (!P.no_specialisation && !(P.tap_replica >= 0 && P.tap_capacity > 0))
  {
#define GRLX_LAUNCH_ACC_SPEC(AGENT)                                                                                              \
    if (SpecPendulumAcc<AGENT>::matches(P))                                                                                      \
    {                                                                                                                            \
      if (variant) *variant = GRLX_KERNEL_SPECIALISED;                                                                           \
      hipLaunchKernelGGL((rollout_acc_kernel<GRLX_ENV_PENDULUM, 3, SpecPendulumAcc<AGENT>>), dim3(waves), dim3(64), 0, stream, P, n_trials); \
      return hipGetLastError();                                                                                                  \
    }
    GRLX_LAUNCH_ACC_SPEC(GRLX_AGENT_SARSA)
    GRLX_LAUNCH_ACC_SPEC(GRLX_AGENT_Q)
#undef GRLX_LAUNCH_ACC_SPEC
  }
  if (P.env == GRLX_ENV_PENDULUM && P.A == 3)
    hipLaunchKernelGGL((rollout_acc_kernel<GRLX_ENV_PENDULUM, 3>), dim3(waves), dim3(64), 0, stream, P, n_trials);
  else if (P.env == GRLX_ENV_ACROBOT && P.A == 3)
    hipLaunchKernelGGL((rollout_acc_kernel<GRLX_ENV_ACROBOT, 3>), dim3(waves), dim3(64), 0, stream, P, n_trials);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}


} // namespace grlx
