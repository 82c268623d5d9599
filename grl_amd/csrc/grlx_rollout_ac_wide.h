// grlx_rollout_ac_wide.h -- actor-critic rollout with 8 replicas per wave: rollout_ac_kernel's table phase (grlx_rollout_ac.h)
// under the wave layout and the per-replica trial sequencing of rollout_wide_kernel (grlx_rollout_wide.h).
// Work queue: a wave starts with the replicas blockIdx*R .. +R-1 in its R slots; a slot whose replica has run its n_trials
// writes it back and takes the next unstarted replica from the counter P.queue (set to gridDim*R by the launcher), until the
// counter passes n_replicas.  Episodes of a learning batch are ragged, so this is what keeps the 8 slots of a wave busy;
// every replica still runs its own trials in order with its own tables and streams: results do not depend on who runs it.
// B = 3 (12 slots): the environment phase is one instruction stream however many lanes it serves, so a third sub-batch makes a replica-step
// cheaper (20.8 + 3 x 20.3 k cycles for 12 instead of 20.8 + 2 x 20.3 k for 8).  Two things make it fit and pay:
//  * the lane state of the sub-batches beyond the second is parked in device memory (P.park: 13 KB per wave and sub-batch, rewritten
//    every pass: it lives in the L2) instead of LDS: 39 KB instead of 52 KB (B = 3) / 66 KB (B = 4) of LDS, four waves per CU as before.  (Keeping it in registers was tried first: the
//    copies in and out of the accumulation registers and 24 more scratch accesses per sub-batch made EVERY sub-batch 15 % slower.)
//  * ROTATION: a wave owns K = ceil(n / waves) consecutive replicas (K <= kAcOwnedMax) and passes its slots round per TRIAL, not per
//    launch -- a slot whose replica has finished a trial queues it behind the waiting ones (a ring in LDS) and takes the one that has
//    waited longest; remaining trials of the launch live in LDS.  16 replicas on 12 slots then keep all slots busy to the last round,
//    where a per-launch hand-over would leave the last four alone in a second round.  Who runs a replica when does not matter (own tables,
//    own streams; the critic's trace persists through slot_store / slot_load as it does between launches).
// Part of the single translation unit grlx_kernels.hip (included there, in order; not self-contained).
#pragma once

namespace grlx {

constexpr int kAcOwnedMax = 64;      // replicas a wave of the rotating (B = 3) kernel can own

// WideRep slots reused: S1 holds the bits of ActionPolicy::n_ (ac_noise), eps_decay holds ActionPolicy::decay_ (ac_decay).
template <int ENV, int B, typename SPEC>
__global__ __launch_bounds__(64) void rollout_ac_wide_kernel(DevParams P, int n_trials)
{
  static_assert(B >= 2 && B <= 4, "sub-batches per wave");
  constexpr int R = 4 * B;
  constexpr bool ROT = B == 3;               // rotation per trial over an owned set of replicas
  constexpr bool GLP = B >= 3;               // the sub-batches beyond the second park their lane state in registers (WideRegPark, grlx_rollout_wide.h), not in LDS
  constexpr int BP = GLP ? 2 : B;            // sub-batches parked in LDS
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
  __shared__ uint4    sh_ctx[BP * kWideQuads * 64];
  __shared__ uint32_t sh_ins[BP * 3 * 64];
  __shared__ uint32_t sh_left[ROT ? kAcOwnedMax : 1];     // ROT: trials of this launch the owned replica i still has to run
  __shared__ uint32_t sh_ring[ROT ? kAcOwnedMax : 1];     //      owned replicas waiting for a slot, oldest first
  __shared__ uint32_t sh_ring_ht[2];                      //      ring: pops so far, pushes so far
  __shared__ uint64_t sh_r64[WR_FIELDS64 * R];
  __shared__ uint32_t sh_r32[WR_FIELDS32 * R];
  __shared__ double   sh_x[S * R];
  __shared__ double   sh_obs[D * R];
  __shared__ double   sh_reward[R];
  __shared__ double   sh_act[R];
  __shared__ int      sh_term[R];
  __shared__ uint32_t sh_step[R];
  __shared__ uint32_t sh_est[R];
  __shared__ uint64_t sh_jump6[kJump6Words];   // LCG jump table, 6-bit windows (lazy weight initialisation)
  __shared__ uint32_t sh_rid[R];               // replica in slot q (kNoReplica: none)
  __shared__ uint32_t sh_xwb[R];               // replica the slot has just retired: its environment state is still in the env lanes
  constexpr uint32_t kNoReplica = 0xFFFFFFFFu;

  jump_table6_to_lds(sh_jump6);
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, j = lane & 15;
  const unsigned long long gmask = 0xFFFFull << (16 * g);
  // ROT: the wave's own replicas wave0 .. wave0 + own - 1, the first R of them in its slots
  const int owned_k = ROT ? (P.n_replicas + (int)gridDim.x - 1) / (int)gridDim.x : R;
  const int wave0 = blockIdx.x * owned_k;
  const int own = (P.n_replicas - wave0 < owned_k) ? ((P.n_replicas - wave0 > 0) ? P.n_replicas - wave0 : 0) : owned_k;

  // ---- environment role
  const int eq = lane % R;
  double x[S];
#pragma unroll
  for (int i = 0; i < S; ++i) x[i] = P.states[(eq < own) ? wave0 + eq : 0].x[i];
  uint32_t estatus = 0;
  if (lane < R)
  {
    sh_step[lane] = 0u; sh_est[lane] = 0u; sh_term[lane] = 0; sh_reward[lane] = 0; sh_act[lane] = 0;
    sh_rid[lane] = (lane < own) ? (uint32_t)(wave0 + lane) : kNoReplica;
    sh_xwb[lane] = kNoReplica;
  }
  if constexpr (ROT)
  {
    if (lane < kAcOwnedMax)
    {
      sh_left[lane] = (lane < own) ? (uint32_t)n_trials : 0u;
      sh_ring[lane] = (R + lane < own) ? (uint32_t)(R + lane) : 0u;          // the owned replicas beyond the slots wait, in order
    }
    if (lane == 0) { sh_ring_ht[0] = 0u; sh_ring_ht[1] = (own > R) ? (uint32_t)(own - R) : 0u; }
  }

  // ---- table role
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

  // replica r into a slot: the critic's trace is restored (positions from HBM, weights from the current table)
  auto slot_load = [&](int r, WideLane &c, WideRep &s, int left) {
    const ReplicaState &RS = P.states[r];
    const Table tabC = table_of(P, 0, r);
    trace_init(c.tr);
    if (up.use_trace)
    {
      const uint32_t *ts = P.trace_state + ((size_t)r * 16 + (size_t)j) * kMaxTrace * 2;
      c.tr.len = RS.tr_len;
      c.tr.total = RS.tr_total;
#pragma unroll
      for (int e = 0; e < kMaxTrace; ++e)
      {
        c.tr.pos[e] = ts[e * 2];
        const uint32_t cw = ts[e * 2 + 1];
        const uint32_t cn = cw & 0xFFFFu;
        c.tr.cnt2 |= ((cn > 0u ? cn - 1u : 0u) & 3u) << (2 * e);
        if (cw >> 16) c.tr.wt |= 1u << e;
        c.tr.dup = c.tr.dup || cn > 1u;
        if (c.tr.pos[e] != kInvalidPos) c.tr.val[e] = value_load(tabC, c.tr.pos[e]);
      }
    }
    c.pd = c.pd_sh = c.p_sh = false;
    c.pd_dW = c.pd_dT = c.pd_wp = 0;
    c.pd_pos = c.p_pos = kInvalidPos;
    c.status = RS.status;
    c.inserted = 0;
    c.ap_pos = kInvalidPos; c.inserted2 = 0; c.ap_sh = false;
    c.wp_seen = 0; c.wap_seen = 0;
    s.G = RS.G; s.TL = RS.TL;
    s.S1 = (uint64_t)__double_as_longlong(RS.ac_noise);
    s.eps_decay = RS.ac_decay;
    s.tt = RS.tt; s.ss = RS.ss;
    s.test_steps = RS.test_steps;
    s.total_reward = 0; s.time = 0; s.action = 0;
    s.action_index = 0;
    s.running = false; s.first = true; s.test = 0;
    s.ending = false;
    s.rows = RS.rows;
    s.trials_left = (P.steps_budget != 0u && (uint64_t)RS.ss >= P.steps_budget) ? 0 : left;
    s.sub_left = 0;
  };
  // ... and back: the critic's trace is persisted (its weights go to the table), the counters and streams to the replica
  // (the environment state follows from the env lanes, see sh_xwb)
  auto slot_store = [&](int r, int q, WideLane &c, WideRep &s) {
    const Table tabC = table_of(P, 0, r);
    trace_flush(c.tr, tabC, false);
    if (up.use_trace)
    {
      uint32_t *ts = P.trace_state + ((size_t)r * 16 + (size_t)j) * kMaxTrace * 2;
#pragma unroll
      for (int e = 0; e < kMaxTrace; ++e)
      {
        ts[e * 2] = c.tr.pos[e];
        ts[e * 2 + 1] = (trace_cnt(c.tr, e) & 0xFFFFu) | (((c.tr.wt >> e) & 1u) << 16);
      }
    }
    uint32_t ic = c.inserted, ia = c.inserted2;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) { ic += __shfl_xor(ic, off, 16); ia += __shfl_xor(ia, off, 16); }
    uint32_t st = c.status | sh_est[q];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) st |= __shfl_xor(st, off, 16);
    if (j == 0)
    {
      ReplicaState &RS = P.states[r];
      RS.G = s.G;
      RS.TL = s.TL;
      RS.ac_decay = s.eps_decay;
      RS.ac_noise = __longlong_as_double((long long)s.S1);
      RS.tt = s.tt;
      RS.ss = s.ss;
      RS.test_steps = s.test_steps;
      RS.n_slots[0] += ic;
      RS.n_slots[1] += ia;
      RS.rows = s.rows;
      RS.tr_len = c.tr.len;
      RS.tr_total = c.tr.total;
      RS.status = st;
    }
  };
  // a slot without a replica: nothing pending, nothing left to run
  auto slot_empty = [&](WideLane &c, WideRep &s) {
    trace_init(c.tr);
    c.pd = c.pd_sh = c.p_sh = false;
    c.pd_dW = c.pd_dT = c.pd_wp = 0;
    c.pd_pos = c.p_pos = kInvalidPos;
    c.status = 0;
    c.inserted = 0;
    c.ap_pos = kInvalidPos; c.inserted2 = 0; c.ap_sh = false;
    c.wp_seen = 0; c.wap_seen = 0;
    s.G = 0; s.TL = 0; s.S1 = 0;
    s.eps_decay = 0;
    s.tt = 0; s.ss = 0;
    s.test_steps = 0;
    s.total_reward = 0; s.time = 0; s.action = 0;
    s.action_index = 0;
    s.running = false; s.first = true; s.test = 0;
    s.ending = false;
    s.rows = 0;
    s.trials_left = 0;
    s.sub_left = 0;
  };
  // the env lanes of a retired slot still hold its last state (and status bits of its steps): to its replica, before the
  // slot's next replica starts
  auto env_write_back = [&]() {
    const uint32_t wb = sh_xwb[eq];
    if (rarely(__any(wb != kNoReplica)))
    {
      wave_sync();
      if (wb != kNoReplica) estatus = 0u;               // the slot's next replica starts clean (its bits went out through sh_est)
      if (wb != kNoReplica && lane < R)
      {
        ReplicaState &RS = P.states[wb];
#pragma unroll
        for (int i = 0; i < S; ++i) RS.x[i] = x[i];
        sh_xwb[eq] = kNoReplica;
      }
      wave_sync();
    }
  };

  // GLP: the lane state of sub-batch 2 / 3 between its turns (b is wave-uniform: the branches below are scalar).  Round 3 kept it in device
  // memory (P.park): 13 KB per wave, sub-batch and pass written and read back, 132 GB of the bench's 237 GB per launch.
  WideRegPark rp2, rp3;
  auto park_state = [&](const WideLane &c, int b) __attribute__((always_inline)) {
    if (GLP && b == 2) wide_pack<true>(c, rp2);
    else if (GLP && b == 3) wide_pack<true>(c, rp3);
    else wide_park<true>(c, sh_ctx + b * kWideQuads * 64, sh_ins + b * 3 * 64, lane);
  };
  for (int b = 0; b < B; ++b)
  {
    const int q = 4 * b + g;
    WideLane c;
    WideRep s;
    if (q < own) slot_load(wave0 + q, c, s, n_trials);
    else slot_empty(c, s);
    park_state(c, b);
    wide_rep_store<R>(s, sh_r64, sh_r32, q);
  }
  wave_sync();

  for (;;)
  {
#ifdef GRLX_WIDE_STAMPS
    const unsigned long long st0 = stamp();
#define GRLX_AC_STAMP(k) { const unsigned long long t__ = stamp(); if (P.diag_out && lane == 0) P.diag_out[(size_t)blockIdx.x * 8 + (k)] += t__ - stl; stl = t__; }
#else
#define GRLX_AC_STAMP(k)
#endif
    // ================= environment phase (see rollout_wide_kernel)
    {
      env_write_back();
      const uint32_t todo = sh_step[eq];               // 0 for a slot without a replica
      if (rarely(__any(todo == 2u)))
      {
        if (todo == 2u)
        {
#pragma unroll
          for (int i = 0; i < S; ++i) x[i] = sh_x[i * R + eq];
        }
      }
      const bool step = todo == 1u;
      if (__any(step))
      {
        double obs[D], reward = 0;
        int terminal = 0;
#pragma unroll
        for (int i = 0; i < D; ++i) obs[i] = 0;
        if (step)
        {
          const double action = sh_act[eq];
          env_step<ENV>(N, x, action, obs, reward, terminal, estatus);
#pragma unroll
          for (int i = 0; i < D; ++i) sh_obs[i * R + eq] = obs[i];
          sh_reward[eq] = reward;
          sh_term[eq] = terminal;
          if (rarely(estatus != 0u)) sh_est[eq] = estatus;       // sticky status bits of this replica's steps (ST_DOMAIN)
        }
      }
      wave_sync();
    }

#ifdef GRLX_WIDE_STAMPS
    const unsigned long long st1 = stamp();
    unsigned long long stl = st1;
    if (P.diag_out && lane == 0) { P.diag_out[(size_t)blockIdx.x * 8 + 0] += st1 - st0; P.diag_out[(size_t)blockIdx.x * 8 + 2] += 1; }
#endif
    // ================= table phase, one sub-batch after the other
    bool more = false;
    for (int b = 0; b < B; ++b)
    {
      const int q = 4 * b + g;
      const uint32_t rid = sh_rid[q];
      const bool live = rid != kNoReplica;
      const int r = live ? (int)rid : 0;
      const ReplicaState &RSg = P.states[r];
      // the three fields the slot-creation path reads (start of the initialisation stream, loaded policy images): loaded here,
      // with the buckets, instead of behind a miss -- early in learning nearly every pass of a sub-batch creates a slot
      ReplicaState RS;
      RS.TL0 = RSg.TL0;
      RS.lazy_base[0] = RSg.lazy_base[0];
      RS.lazy_base[1] = RSg.lazy_base[1];
      const Table tabC = table_of(P, 0, r), tabA = table_of(P, 1, r);
      WideLane c;
      // GLP: the LDS loads of the first two sub-batches now, the unpacking after the hashing; those beyond are unpacked from their registers
      uint4 raw[kWideQuads];
      uint32_t rawi[3];
      if constexpr (GLP)
      {
        if (b < 2) wide_unpark_load<true>(raw, rawi, sh_ctx + b * kWideQuads * 64, sh_ins + b * 3 * 64, lane);
      }
      else
        wide_unpark<true>(c, sh_ctx + b * kWideQuads * 64, sh_ins + b * 3 * 64, lane);
      auto decode_state = [&](WideLane &cc) __attribute__((always_inline)) {
        if (b == 2) wide_unpark_decode<true>(cc, rp2.q, rp2.i);
        else if (b == 3) wide_unpark_decode<true>(cc, rp3.q, rp3.i);
        else wide_unpark_decode<true>(cc, raw, rawi);
      };
      WideRep s;
      wide_rep_load<R>(s, sh_r64, sh_r32, q);
      if (!__any(live)) continue;                     // (a live slot always has something to do: it retires the moment it has not)
      double ac_noise = __longlong_as_double((long long)s.S1), ac_decay = s.eps_decay;

      uint32_t slotA[1] = {0}, slotC[1] = {0};
      Lookup lkA[1], lkC[1];
      BucketRegs brA[1], brC[1];
      double wap = 0, wpc = 0;
      bool has_next = false, update = false, need_critic = false;
      double obs[D], reward = 0;
      int terminal = 0;
#pragma unroll
      for (int i = 0; i < D; ++i) obs[i] = sh_obs[i * R + q];
      if (s.running)
      {
        if (!s.first)
        {
          reward = sh_reward[q];
          terminal = sh_term[q];
          s.total_reward += reward;
          s.time += 1;
        }
        has_next = s.first || terminal != 2;
        update = !s.first && !s.test;
        need_critic = has_next && !s.test;
        if (has_next)
        {
          slotA[0] = tile_slot_obs<T>(N.tile_actor, obs, D, j);
          slotC[0] = tile_slot_obs<T>(N.tile, obs, D, j);
        }
        if constexpr (GLP) decode_state(c);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (update)
        { // as looked up one pass ago (the actor's: or as written by the last actor update to the same slot); slots shared
          // between tilings are loaded (another lane may have written them)
          wap = c.ap_sh ? value_load(tabA, c.ap_pos) : c.wap_seen;
          wpc = c.wp_seen;
        }
        if (has_next) table_issue<1>(tabA, slotA, lkA, brA);
        if (need_critic)
        {
          if (P.twin_tables) bucket_load_vals(tabC, table_home(tabC, slotC[0]), brC[0]);     // same keys as the actor's bucket
          else table_issue<1>(tabC, slotC, lkC, brC);
        }
      }
      else if constexpr (GLP)
        decode_state(c);

      GRLX_AC_STAMP(3)
      // the PREVIOUS step's critic update, in the shadow of the loads just issued
      Evicted ev;
      ev.n = 0u; ev.pos = kInvalidPos; ev.val = 0;
      if (c.pd)
      {
        sh_ppos[g * 16 + j] = c.pd_pos;
        sh_fbflag[j * 4 + g] = 0u;
      }
      wave_sync();
      if (c.pd)
      {
        up.dW = c.pd_dW;
        up.dT = c.pd_dT;
        td_update_lane<true>(c.tr, tabC, up, c.pd_pos, c.pd_sh, c.pd_wp, g, j, sh_ppos, sh_fb, sh_fbflag, c.status, ev);
        c.pd = false;
      }

      GRLX_AC_STAMP(4)
      if (s.running)
      {
        uint32_t posA[1] = {kInvalidPos}, posC[1] = {kInvalidPos};
        double wA[1] = {0}, wC[1] = {0};
        bool shA[1] = {false}, shC[1] = {false};
        if (P.twin_tables)
        { // equal tile codings: one resolution, one creation path for both tables (table_get_finish_twin)
          if (has_next)
          {
            bool shared_event = false;
            table_get_finish_twin<2>(tabA, tabC, N.lin_actor, N.lin, RS, slotA, lkA, brA, brC, need_critic, posA, wA, wC[0], shA, g, j, gmask,
                                     sh_mb, sh_ms, sh_mail, sh_jump6, c.status, c.inserted2, c.inserted,
                                     [&](uint32_t mp) {
                                       if (c.ap_pos == mp) c.ap_sh = true;
                                       if (ev.pos != kInvalidPos && ev.pos == mp) value_store(tabC, mp, ev.val);
                                       trace_share_event(c.tr, tabC, mp);
                                       if (c.p_pos == mp) c.p_sh = true;
                                       shared_event = true;
                                     });
            posC[0] = posA[0];
            shC[0] = shA[0];
            if (rarely(__any(shared_event)) && update) wpc = value_load(tabC, c.p_pos);
          }
        }
        else
        {
        if (has_next)
        {
          table_get_finish<1, 2>(tabA, N.lin_actor, RS, 1, slotA, lkA, brA, posA, wA, shA, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump6,
                                     c.status, c.inserted2, [&](uint32_t mp) { if (c.ap_pos == mp) c.ap_sh = true; });
        }
        if (need_critic)
        {
          bool shared_event = false;
          table_get_finish<1, 2>(tabC, N.lin, RS, 0, slotC, lkC, brC, posC, wC, shC, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump6,
                                     c.status, c.inserted,
                                     [&](uint32_t mp) {
                                       if (ev.pos != kInvalidPos && ev.pos == mp) value_store(tabC, mp, ev.val);
                                       trace_share_event(c.tr, tabC, mp);
                                       if (c.p_pos == mp) c.p_sh = true;
                                       shared_event = true;
                                     });
          if (rarely(__any(shared_event)) && update) wpc = value_load(tabC, c.p_pos);
        }
        }
        {
          const bool risky = ev.n > 1u || (update && c.p_sh) || (need_critic && shC[0]);
          if (rarely(__any(risky)))
          {
            if (need_critic) wC[0] = value_load(tabC, posC[0]);
            if (update) wpc = value_load(tabC, c.p_pos);
          }
          const bool held = ev.pos != kInvalidPos;
          wC[0] = (held && posC[0] == ev.pos) ? ev.val : wC[0];
          wpc = (held && c.p_pos == ev.pos) ? ev.val : wpc;
        }
        if (need_critic) wC[0] = trace_forward(c.tr, posC[0], wC[0]);
        if (update) wpc = trace_forward(c.tr, c.p_pos, wpc);
        GRLX_AC_STAMP(5)
        SHA(0, j, g) = wA[0];
        SHA(1, j, g) = wC[0];
        SHA(2, j, g) = wap;
        SHA(3, j, g) = wpc;
        sh_apos[g * 16 + j] = c.ap_pos;
        wave_sync();
        double sums[4];
        {
          const int row = j & 3;
          double sum = 0;
#pragma unroll
          for (int k = 0; k < 16; ++k) sum += SHA(row, k, g);
          sh_res[g * 16 + j] = sum / 16;
        }
        wave_sync();
#pragma unroll
        for (int row = 0; row < 4; ++row) sums[row] = sh_res[g * 16 + row];
        const double u_next = clampd(sums[0], a_min, a_max);
        const double v_next = clampd(sums[1], up.out_min, up.out_max);
        const double u_prev = clampd(sums[2], a_min, a_max);
        const double v_prev = clampd(sums[3], up.out_min, up.out_max);

        // policy (ActionPolicy::act, action.cpp:127-158)
        double a_next = 0;
        if (has_next)
        {
          double out = u_next;
          if (!s.test)
          {
            if (s.time == 0) ac_noise = 0;
            if (s.time == 0.) ac_decay = fmax(ac_decay * N.ac_decay_rate, N.ac_decay_min);
            if (N.sigma != 0)
            { // Rand::getNormal(0, decay*sigma): two thread-local draws (utils.h:120-125)
              s.TL = lcg_next(s.TL);
              const double U1 = lcg_double(s.TL);
              s.TL = lcg_next(s.TL);
              const double U2 = lcg_double(s.TL);
              const double sg = ac_decay * N.sigma;
              const double nrm = __builtin_sqrt(-2 * plog(U1)) * pcos(2 * GRLX_PI * U2) * sg + 0.;
              ac_noise = (1 - N.theta) * ac_noise + nrm;
              out += ac_noise;
            }
          }
          a_next = fmin(fmax(out, N.action_min), N.action_max);
        }

        // predictor (ActionACPredictor::update, ac.cpp:72-110)
        if (update)
        {
          double target = reward;
          if (has_next) target += N.gamma * v_next;
          const double delta = target - v_prev;
          c.pd = true;
          c.pd_dW = N.alpha * (target - v_prev);
          c.pd_dT = N.alpha * delta;
          c.pd_pos = c.p_pos;
          c.status |= (c.p_pos == kInvalidPos) ? ST_BAD_POS : 0u;
          c.pd_sh = c.p_sh;
          c.pd_wp = wpc;
          if (N.ac_update_method == 0 || delta > 0)
          {
            double du = s.action - u_prev;
            if (N.ac_update_method == 0) du = delta * du;
            if (N.ac_step_limit >= 0) du = fmin(fmax(du, -N.ac_step_limit), N.ac_step_limit);
            const double target_u = u_prev + du;
            const double dA = N.actor_alpha * (target_u - u_prev);
            uint32_t cpa = 1;
            const uint32_t amask = (uint32_t)((__ballot(c.ap_sh) >> (16 * g)) & 0xFFFFull);
            for (uint32_t mm = amask; mm != 0u; mm &= mm - 1u)
            {
              const int k = __builtin_ctz(mm);
              if (k != j && sh_apos[g * 16 + k] == c.ap_pos) cpa++;
            }
            double nv = wap;
            for (uint32_t cc = 0; cc < cpa; ++cc) nv = a_limit ? clampd(nv + dA, a_min, a_max) : nv + dA;
            value_store(tabA, c.ap_pos, nv);
            if (has_next && posA[0] == c.ap_pos) wA[0] = nv;       // the next step updates the same slot: it continues from this value
          }
        }

        if (!s.first)
        {
          if (s.test) s.test_steps++;
          else s.ss++;
        }
        if (has_next)
        {
          s.action = a_next;
          c.ap_pos = posA[0]; c.ap_sh = shA[0];
          c.wap_seen = wA[0];
          if (need_critic) { c.p_pos = posC[0]; c.p_sh = shC[0]; c.wp_seen = wC[0]; }
        }
        if (!s.first && terminal) { s.running = false; s.ending = true; }
        s.first = false;
      }
      if (ev.pos != kInvalidPos) value_store(tabC, ev.pos, ev.val);
      GRLX_AC_STAMP(6)

      // ---- between trials
      uint32_t step_next = s.running ? 1u : 0u;
      bool now_live = live;
      const bool at_rest = live && !s.running && !c.pd;
      if (__any(at_rest))
      {
        bool between = at_rest && s.trials_left > 0;
        bool again = false;                                 // another greedy episode of the same test trial follows (test_trials)
        bool ended_now = false;                             // a trial of this slot's replica has just ended
        if (between && s.ending && s.test && s.sub_left > 0)
        {
          s.sub_left--;
          s.ending = false;
          again = true;
        }
        if (between && s.ending)
        { // end of a learning trial: make the table current; the entries stay -- the reference never clears the critic's trace
          if (!s.test) trace_flush(c.tr, tabC, false);
          const int ti = N.test_interval;
          if (ti >= 0 ? s.test : 1)
          {
            if (s.rows < (uint32_t)P.max_rows)
            {
              if (j == 0)
              {
                const size_t at = (size_t)s.rows * (size_t)P.n_replicas + (size_t)r;
                const double sub = (s.test && P.test_trials > 1) ? (double)P.test_trials : 1.;     // online_learning.cpp:224-225
                P.row_reward[at] = s.total_reward / sub;
                P.row_time[at] = s.time / sub;
                P.row_steps[at] = s.ss;
                P.row_trial[at] = (ti >= 0) ? (s.tt + 1 - (s.tt + 1) / (ti + 1)) : s.tt;
              }
              s.rows++;
            }
            else
              c.status |= ST_ROWS_FULL;
          }
          s.tt++;
          s.trials_left--;
          if (P.steps_budget != 0u && (uint64_t)s.ss >= P.steps_budget) s.trials_left = 0;        // online_learning.cpp:154: `ss < steps_`
          s.ending = false;
          ended_now = true;
        }
        bool leave = at_rest && s.trials_left == 0 && !again;      // this replica is done
        if constexpr (ROT)
        { // ... or has finished a trial and has more to run: to the back of the wave's ring, behind the ones that are waiting
          const bool rotate = at_rest && ended_now && !again && s.trials_left > 0;
          if (rotate && j == 0)
          {
            const uint32_t me = rid - (uint32_t)wave0;
            sh_left[me] = (uint32_t)s.trials_left;
            sh_ring[atomicAdd(&sh_ring_ht[1], 1u) % (uint32_t)kAcOwnedMax] = me;
          }
          leave = leave || rotate;
          wave_sync();                                             // every push of this pass before the first pop
        }
        if (leave)
        { // back to HBM, and the next one (group-uniform: all 16 lanes are here)
          uint32_t nr = (uint32_t)P.n_replicas;                    // (= none)
          int left = n_trials;
          if constexpr (ROT)
          {
            uint32_t nl = kNoReplica;
            if (j == 0)
            {
              const uint32_t h = atomicAdd(&sh_ring_ht[0], 1u);
              if (h < sh_ring_ht[1]) nl = sh_ring[h % (uint32_t)kAcOwnedMax];
              else atomicSub(&sh_ring_ht[0], 1u);
            }
            nl = __shfl(nl, 0, 16);
            if (nl != kNoReplica)
            {
              nr = (uint32_t)wave0 + nl;
              left = (int)sh_left[nl];
            }
          }
          else
          {
            if (j == 0) nr = __hip_atomic_fetch_add(P.queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            nr = __shfl(nr, 0, 16);
          }
          if (!(ROT && nr == rid))                                 // (nobody was waiting: the replica goes on in its slot)
          {
            slot_store(r, q, c, s);
            if (j == 0)
            {
              sh_xwb[q] = rid;
              sh_est[q] = 0u;
              sh_rid[q] = (nr < (uint32_t)P.n_replicas) ? nr : kNoReplica;
            }
            // ROT: the replica taken over may have been written back by another slot of this wave a moment ago (or in this very
            // pass): its state, its trace and the weights the trace flushed must be in memory before they are read
            if constexpr (ROT) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            if (nr < (uint32_t)P.n_replicas) slot_load((int)nr, c, s, left);
            else { slot_empty(c, s); now_live = false; }
          }
          between = now_live && s.trials_left > 0;
        }
        if (between && (s.trials_left > 0 || again))
        {
          const int ti = N.test_interval;
          if (!again)
          {
            s.test = (ti >= 0 && s.tt % (ti + 1) == ti) ? 1 : 0;
            s.sub_left = s.test ? P.test_trials - 1 : 0;
          }
          double xs[S], ob0[D];
          Env<ENV>::start(N, s.test, s.TL, s.G, xs);
          Env<ENV>::observe(N, xs, ob0);
#pragma unroll
          for (int i = 0; i < S; ++i) sh_x[i * R + q] = xs[i];
#pragma unroll
          for (int i = 0; i < D; ++i) sh_obs[i * R + q] = ob0[i];
          if (!again)
          { // (reward and time keep adding up across the episodes of one test trial)
            s.total_reward = 0;
            s.time = 0;
          }
          s.action = 0;
          s.running = true;
          s.first = true;
          step_next = 2u;
        }
      }

      if (j == 0 && live)
      {
        sh_act[q] = s.action;
        sh_step[q] = step_next;
      }
      more = more || now_live;
      s.S1 = (uint64_t)__double_as_longlong(ac_noise);
      s.eps_decay = ac_decay;
      wide_rep_store<R>(s, sh_r64, sh_r32, q);
      park_state(c, b);
      GRLX_AC_STAMP(7)
    }
    wave_sync();
#ifdef GRLX_WIDE_STAMPS
    if (P.diag_out && lane == 0) P.diag_out[(size_t)blockIdx.x * 8 + 1] += stamp() - st1;
#endif
    if (!__any(more)) break;
  }

  // every slot has retired its last replica inside the loop; the env lanes still owe the states of the last ones
  env_write_back();
}

#undef GRLX_AC_STAMP

} // namespace grlx
