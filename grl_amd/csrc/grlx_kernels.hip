// grlx_kernels.hip -- hand-written HIP kernels (gfx950) for grl's online-learning hot path.
//
// Layout of the fused rollout kernel
// ----------------------------------
//  * one wavefront (64 lanes) = 4 independent replicas x 16 lanes; lane j of a
//    replica owns tiling j of the hashed tile coding (T = 16).
//  * per-replica scalar work (RK4 integration of the dynamics, reward, RNG,
//    epsilon-greedy) is computed redundantly by the 16 lanes of the replica --
//    the lanes would otherwise idle, and it removes every broadcast.
//  * the eligibility trace lives in registers: lane j keeps, for each of the
//    <= 10 trace entries, the table position and the AUTHORITATIVE weight of tiling j
//    (write-back: stored when the slot leaves the trace; grlx_update.h).
//  * weights live in a per-replica open-addressing table in HBM (64-byte buckets of
//    four entries, grlx_table.h), lazily initialised to the value the reference's
//    8,388,608-draw initialisation gives that slot (LCG jump-ahead, grlx_rng.h).
//  * the TD update of a step is applied one pass later, under the next step's loads.
//  * sums over the 16 tilings are taken in the reference's serial order
//    (linear.cpp:147-151) through a 4-way interleaved LDS tile, so Q-values are
//    bit-identical to a scalar run.
//
// Compiled with -ffp-contract=off: the reference's arithmetic is plain IEEE
// double without fused multiply-add (x86-64 baseline); only grlx_math.h fuses.
#include "grlx_internal.h"
#include "grlx_math.h"
#include "grlx_rng.h"
#include "grlx_tile.h"
#include "grlx_table.h"
#include "grlx_envs.h"
#include "grlx_policy.h"

namespace grlx {


// ------------------------------------------------------- fused rollout -----
// LDS tile shared by the 4 replicas of the wave; index (row*16 + tiling)*4 + group
// makes both the per-lane writes and the 4 simultaneous broadcast reads conflict-free.
#define SHW(row, k, g) sh_w[(((row) * 16 + (k)) << 2) + (g)]


} // namespace grlx

#include "grlx_update.h"

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
#define DIAG_STAMP(slot)                                   \
  if (DIAG)                                                \
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
struct SpecNone {
  __device__ static __forceinline__ const DevParams &numeric(const DevParams &P) { return P; }
  __device__ static __forceinline__ int agent(const DevParams &P) { return P.agent; }
};

// ADV: advantage learning (predictor/critic/advantage, advantage.cpp:222-268) also reads A(s, .) of the
// PREVIOUS state for every action with the current weights: NA more rows (their table positions are
// those of the previous pass), Q(s,a) being one of them.  Built without the deferred update.
template <int ENV, int NA, bool DIAG, typename SPEC, bool DEFER = !DIAG, bool ADV = false>
__global__ __launch_bounds__(64) void rollout_kernel(DevParams P, int n_trials)
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
  if (DIAG) diag_last = stamp();

  for (int trial = 0; trial < n_trials; ++trial, ++tt)
  {
    const int ti = N.test_interval;
    const int test = (ti >= 0 && tt % (ti + 1) == ti) ? 1 : 0;        // online_learning.cpp:160
    double obs[D], reward = 0, total_reward = 0;
    int terminal = 0;
    bool running = live;

    // environment_->start (modeled.cpp:132-158)
    if (live)
    {
      Env<ENV>::start(N, test, TL, G, x);
      Env<ENV>::observe(N, x, obs);
    }
    // agent->start: TDAgent::start clears the trace (td.cpp:50-61, sarsa.cpp:126-132); the
    // trace was written back at the end of the previous learning trial, so it is empty here
    double time = 0;
    double action = 0;
    int    action_index = 0;
    uint32_t p_pos = kInvalidPos, p_slot = 0;
    bool   p_sh = false;
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
          env_step<ENV>(N, x, action, obs, reward, terminal, status);     // online_learning.cpp:196
          total_reward += reward;                                          // :202
          time += 1;                                                       // tau = 1
        }
        has_next = first || terminal != 2;
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
        if (update) wp = value_load(tab, p_pos);                           // weights of project(s, a) as stored
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
          table_get_finish<NA>(tab, N.lin, RS, 0, slot, lk, br, pos, w, sh, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump, status, inserted,
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
            pd = true;
            pd_dW = dW;
            pd_dT = dT;
            pd_pos = p_pos;
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
        if (!DEFER && tapped && (!first || P.tap_starts))
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

    // row of a test trial (online_learning.cpp:238-262) -- or of every trial when test_interval < 0
    if (live && (ti >= 0 ? test : 1))
    {
      if (rows < (uint32_t)P.max_rows)
      {
        if (j == 0)
        {
          size_t at = (size_t)rows * (size_t)P.n_replicas + (size_t)r;
          P.row_reward[at] = total_reward;
          P.row_time[at] = time;
          P.row_steps[at] = ss;
          P.row_trial[at] = (ti >= 0) ? (tt + 1 - (tt + 1) / (ti + 1)) : tt;
        }
        rows++;
      }
      else
        status |= ST_ROWS_FULL;
    }
  }

  if (DIAG && P.diag_out && lane == 0)
    for (int k = 0; k < 8; ++k) P.diag_out[(size_t)blockIdx.x * 8 + k] = diag_sum[k];

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

// cfg/cart_pole/ac_tc.yaml as compile-time constants (see SpecPendulumTcA): every field the actor-critic
// kernel reads, derived with the expressions of make_params (grlx_api.cpp)
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

  for (int trial = 0; trial < n_trials; ++trial, ++tt)
  {
    const int ti = N.test_interval;
    const int test = (ti >= 0 && tt % (ti + 1) == ti) ? 1 : 0;
    double obs[D], reward = 0, total_reward = 0;
    int terminal = 0;
    bool running = live;
    if (live)
    {
      Env<ENV>::start(N, test, TL, G, x);
      Env<ENV>::observe(N, x, obs);
    }
    double time = 0, action = 0;
    uint32_t p_pos = kInvalidPos, p_slot = 0, ap_pos = kInvalidPos, ap_slot = 0;
    bool p_sh = false, ap_sh = false;
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
        {
          wap = value_load(tabA, ap_pos);                // actor weights of project(prev_obs), current
          wpc = value_load(tabC, p_pos);                 // critic weights of project(prev_obs), as stored
        }
        // both tables' home buckets in flight together: one memory round trip for the two lookups
        if (has_next) table_issue<1>(tabA, slotA, lkA, brA);
        if (need_critic) table_issue<1>(tabC, slotC, lkC, brC);
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
          if (need_critic) { p_pos = posC[0]; p_slot = slotC[0]; p_sh = shC[0]; }
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

    if (live && (ti >= 0 ? test : 1))
    {
      if (rows < (uint32_t)P.max_rows)
      {
        if (j == 0)
        {
          size_t at = (size_t)rows * (size_t)P.n_replicas + (size_t)r;
          P.row_reward[at] = total_reward;
          P.row_time[at] = time;
          P.row_steps[at] = ss;
          P.row_trial[at] = (ti >= 0) ? (tt + 1 - (tt + 1) / (ti + 1)) : tt;
        }
        rows++;
      }
      else
        status |= ST_ROWS_FULL;
    }
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

  for (int trial = 0; trial < n_trials; ++trial, ++tt)
  {
    const int ti = P.test_interval;
    const int test = (ti >= 0 && tt % (ti + 1) == ti) ? 1 : 0;
    double obs[D], reward = 0, total_reward = 0;
    int terminal = 0;
    bool running = live;
    if (live)
    {
      Env<ENV>::start(P, test, TL, G, x);
      Env<ENV>::observe(P, x, obs);
    }
    double time = 0, action = 0;
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

    if (live && (ti >= 0 ? test : 1))
    {
      if (rows < (uint32_t)P.max_rows)
      {
        if (j == 0)
        {
          size_t at = (size_t)rows * (size_t)P.n_replicas + (size_t)r;
          P.row_reward[at] = total_reward;
          P.row_time[at] = time;
          P.row_steps[at] = ss;
          P.row_trial[at] = (ti >= 0) ? (tt + 1 - (tt + 1) / (ti + 1)) : tt;
        }
        rows++;
      }
      else
        status |= ST_ROWS_FULL;
    }
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

// ------------------------------------------------- accumulating-trace rollout ---
// trace/enumerated/accumulating (trace.h:238-263): no ssub, cut 1e-4 -- up to 19 entries in which a slot may
// occur many times, every occurrence adding to the weight in the reference's order (entry-major, newest first;
// tiling-minor).  Nothing is cached here: the trace holds table positions only and every update is a
// read-modify-write of the table, entry after entry (same-address accesses of one wave complete in issue
// order); slots shared between tilings are updated one lane at a time in tiling order.  SARSA, Q-learning and
// Expected SARSA over one Q table; TD update in place.  A plain, correct path -- about 3x the time of the
// replacing-trace kernel.
constexpr int kAccTrace = 20;

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

template <int ENV, int NA>
__global__ __launch_bounds__(64) void rollout_acc_kernel(DevParams P, int n_trials)
{
  constexpr int S = Env<ENV>::S, D = Env<ENV>::D, T = kLanesPerReplica;
  __shared__ double   sh_w[(NA + 1) * 16 * 4];
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
  uint32_t status = RS.status, rows = RS.rows, inserted = 0;

  const Table tab = table_of(P, 0, r);
  const double out_min = P.lin.out_min, out_max = P.lin.out_max;
  const bool limit = P.lin.limit != 0;
  const double ee = P.gl, cut = 0.0001;

  double acts[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) acts[a] = P.actions[a];
  uint32_t key_act[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a)
    key_act[a] = in_reg(murmur_key(tile_coord<T>(P.tile, D, tile_quant(P.tile, D, P.actions[a]), j)));
  const uint32_t key_j = in_reg(murmur_key(j));

  // the trace of this lane's tiling: positions newest first, bit e of tsh = entry e is a slot shared between tilings
  uint32_t tpos[kAccTrace];
#pragma unroll
  for (int e = 0; e < kAccTrace; ++e) tpos[e] = kInvalidPos;
  uint32_t tsh = 0;
  int tlen = 0;
  double ttotal = 1.;
  int tr_len_ref = 0;

  auto add_to = [&](uint32_t pos, double d) {       // LinearRepresentation::update of one index (linear.cpp:198-216)
    const double v = value_load(tab, pos) + d;
    value_store(tab, pos, limit ? clampd(v, out_min, out_max) : v);
  };

  for (int trial = 0; trial < n_trials; ++trial, ++tt)
  {
    const int ti = P.test_interval;
    const int test = (ti >= 0 && tt % (ti + 1) == ti) ? 1 : 0;
    double obs[D], reward = 0, total_reward = 0;
    int terminal = 0;
    bool running = live;
    if (live)
    {
      Env<ENV>::start(P, test, TL, G, x);
      Env<ENV>::observe(P, x, obs);
    }
    double time = 0, action = 0;
    int action_index = 0;
    uint32_t p_pos = kInvalidPos, p_slot = 0;
    bool p_sh = false;
    if (!test)
    { // TDAgent::start -> predictor->finalize() -> trace_->clear() (td.cpp:54, sarsa.cpp:126-132)
#pragma unroll
      for (int e = 0; e < kAccTrace; ++e) tpos[e] = kInvalidPos;
      tsh = 0; tlen = 0; ttotal = 1.; tr_len_ref = 0;
    }
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
            hpre = murmur_mix(hpre, tile_coord<T>(P.tile, i, tile_quant(P.tile, i, obs[i]), j));
          const uint32_t hpm = hpre * 0x5bd1e995u;
#pragma unroll
          for (int a = 0; a < NA; ++a)
          {
            uint32_t h = murmur_absorb(hpm ^ key_act[a], key_j);
            const uint32_t hm = murmur_final(h), mem = (uint32_t)P.tile.memory;
            slot[a] = ((mem & (mem - 1u)) == 0u) ? (hm & (mem - 1u)) : (hm % mem);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (has_next)
          table_get<NA>(tab, P.lin, RS, 0, slot, pos, w, sh, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump, status, inserted,
                        [&](uint32_t mp) { // a slot became shared: every entry that refers to it is updated serially from now on
                          if (p_pos == mp) p_sh = true;
#pragma unroll
                          for (int e = 0; e < kAccTrace; ++e) tsh |= (tpos[e] == mp) ? (1u << e) : 0u;
                        });
        double wp = 0;
        if (update) wp = value_load(tab, p_pos);          // the table is always current here
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

        // -------- predictor update (sarsa.cpp:98-124, 167-194 / advantage.cpp:71-110)
        double delta = 0;
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
          delta = target - qsa;
          const double dW = P.alpha * (target - qsa);
          const double dT = P.alpha * delta;
          // write(p, target, alpha): every index of p, in tiling order where tilings share the slot
          if (!p_sh) add_to(p_pos, dW);
          serial_lanes(p_sh, [&]() { add_to(p_pos, dW); });
          // update(trace, alpha*delta, e): newest entry first while its weight exceeds 0.001.  The slots this
          // tiling owns alone are loaded together (one round trip); an entry whose slot occurred in a newer
          // entry continues from that entry's result instead of its (stale) load, so every slot still
          // receives its additions one after the other in entry order.  Shared slots: one lane at a time.
          double cur[kAccTrace], de[kAccTrace];
          bool mine[kAccTrace];
          {
            double weight = 1.;
#pragma unroll
            for (int e = 0; e < kAccTrace; ++e)
            {
              const bool go = e < tlen && weight > 0.001;
              de[e] = weight * dT * ee;
              mine[e] = go && ((tsh >> e) & 1u) == 0u;
              cur[e] = mine[e] ? value_load(tab, tpos[e]) : 0.;
              weight *= ee;
            }
          }
#pragma unroll
          for (int e = 0; e < kAccTrace; ++e)
          {
            double base = cur[e];
#pragma unroll
            for (int k = 0; k < e; ++k) base = (mine[k] && tpos[k] == tpos[e]) ? cur[k] : base;   // the latest newer occurrence wins
            const double v = base + de[e];
            cur[e] = limit ? clampd(v, out_min, out_max) : v;
            if (mine[e]) value_store(tab, tpos[e], cur[e]);
          }
          {
            double weight = 1.;
#pragma unroll
            for (int e = 0; e < kAccTrace; ++e)
            {
              const bool go = e < tlen && weight > 0.001;
              const bool shared = ((tsh >> e) & 1u) != 0u;
              const double d = de[e];
              const uint32_t at = tpos[e];
              if (rarely(__any(go && shared)))
                serial_lanes(go && shared, [&]() { add_to(at, d); });
              weight *= ee;
            }
          }
          // trace_->add(p, e) (trace.h:245-262)
          if (ee < cut) { tlen = 0; ttotal = 1.; tsh = 0; }
          if (tlen >= kAccTrace) status |= ST_TRACE_OVERFLOW;       // cannot happen: validated at create
#pragma unroll
          for (int e = kAccTrace - 1; e > 0; --e) tpos[e] = tpos[e - 1];
          tsh = (tsh << 1) & ((1u << kAccTrace) - 1u);
          tpos[0] = p_pos;
          if (p_sh) tsh |= 1u;
          tlen = (tlen < kAccTrace) ? tlen + 1 : kAccTrace;
          ttotal *= ee;
          while (ttotal < cut && tlen > 1)
          {
            ttotal /= ee;
            tlen--;
          }
#pragma unroll
          for (int e = 0; e < kAccTrace; ++e)
            if (e >= tlen) { tpos[e] = kInvalidPos; tsh &= ~(1u << e); }
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
        }
        if (!first && terminal) running = false;
        first = false;
      }
    }

    if (live && (ti >= 0 ? test : 1))
    {
      if (rows < (uint32_t)P.max_rows)
      {
        if (j == 0)
        {
          size_t at = (size_t)rows * (size_t)P.n_replicas + (size_t)r;
          P.row_reward[at] = total_reward;
          P.row_time[at] = time;
          P.row_steps[at] = ss;
          P.row_trial[at] = (ti >= 0) ? (tt + 1 - (tt + 1) / (ti + 1)) : tt;
        }
        rows++;
      }
      else
        status |= ST_ROWS_FULL;
    }
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
  if (P.env == GRLX_ENV_PENDULUM && P.A == 3)
    hipLaunchKernelGGL((rollout_acc_kernel<GRLX_ENV_PENDULUM, 3>), dim3(waves), dim3(64), 0, stream, P, n_trials);
  else if (P.env == GRLX_ENV_ACROBOT && P.A == 3)
    hipLaunchKernelGGL((rollout_acc_kernel<GRLX_ENV_ACROBOT, 3>), dim3(waves), dim3(64), 0, stream, P, n_trials);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_rollout_ac(const DevParams &P, int n_trials, hipStream_t stream, int *variant)
{
  const bool taps = P.tap_replica >= 0 && P.tap_capacity > 0;          // recorded by the in-place instantiation
  if (variant) *variant = taps ? GRLX_KERNEL_IN_PLACE : GRLX_KERNEL_GENERIC;
  int waves = (P.n_replicas + kReplicasPerWave - 1) / kReplicasPerWave;
  switch (P.env)
  {
    case GRLX_ENV_CART_POLE:
      if (taps)
        hipLaunchKernelGGL((rollout_ac_kernel<GRLX_ENV_CART_POLE, SpecNone, false>), dim3(waves), dim3(64), 0, stream, P, n_trials);
      else if (!P.no_specialisation && SpecCartPoleAc::matches(P))
      {
        if (variant) *variant = GRLX_KERNEL_SPECIALISED;
        hipLaunchKernelGGL((rollout_ac_kernel<GRLX_ENV_CART_POLE, SpecCartPoleAc, true>), dim3(waves), dim3(64), 0, stream, P, n_trials);
      }
      else
        hipLaunchKernelGGL((rollout_ac_kernel<GRLX_ENV_CART_POLE, SpecNone, true>), dim3(waves), dim3(64), 0, stream, P, n_trials);
      break;
    case GRLX_ENV_PENDULUM:
      if (taps)
        hipLaunchKernelGGL((rollout_ac_kernel<GRLX_ENV_PENDULUM, SpecNone, false>), dim3(waves), dim3(64), 0, stream, P, n_trials);
      else
        hipLaunchKernelGGL((rollout_ac_kernel<GRLX_ENV_PENDULUM, SpecNone, true>), dim3(waves), dim3(64), 0, stream, P, n_trials);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_rollout(const DevParams &P, int n_trials, hipStream_t stream, int *variant)
{
  if (variant) *variant = GRLX_KERNEL_GENERIC;
  int waves = (P.n_replicas + kReplicasPerWave - 1) / kReplicasPerWave;
  // stamps and per-step taps are recorded by the instantiation that updates in place
  const bool inplace = P.diag_out != nullptr || (P.tap_replica >= 0 && P.tap_capacity > 0);
  if (P.diag_out && P.diag_deferred && P.env == GRLX_ENV_PENDULUM && P.A == 3 && !(P.tap_replica >= 0 && P.tap_capacity > 0))
  {
    if (variant) *variant = GRLX_KERNEL_GENERIC;
    hipLaunchKernelGGL((rollout_kernel<GRLX_ENV_PENDULUM, 3, true, SpecNone, true>), dim3(waves), dim3(64), 0, stream, P, n_trials);
    return hipGetLastError();
  }
#define GRLX_LAUNCH(ENVID, NACT)                                                                              \
  if (P.env == ENVID && P.A == NACT)                                                                        \
  {                                                                                                         \
    if (variant && inplace) *variant = GRLX_KERNEL_IN_PLACE;                                                \
    if (inplace)                                                                                            \
      hipLaunchKernelGGL((rollout_kernel<ENVID, NACT, true, SpecNone>), dim3(waves), dim3(64), 0, stream, P, n_trials); \
    else                                                                                                    \
      hipLaunchKernelGGL((rollout_kernel<ENVID, NACT, false, SpecNone>), dim3(waves), dim3(64), 0, stream, P, n_trials); \
    return hipGetLastError();                                                                               \
  }
  if (P.agent == GRLX_AGENT_ADVANTAGE)
  { // advantage learning: its own in-place instantiation (taps included)
    if (variant) *variant = GRLX_KERNEL_IN_PLACE;
#define GRLX_LAUNCH_ADV(ENVID, NACT)                                                                                  \
    if (P.env == ENVID && P.A == NACT)                                                                              \
    {                                                                                                               \
      hipLaunchKernelGGL((rollout_kernel<ENVID, NACT, true, SpecNone, false, true>), dim3(waves), dim3(64), 0, stream, P, n_trials); \
      return hipGetLastError();                                                                                     \
    }
    GRLX_LAUNCH_ADV(GRLX_ENV_PENDULUM, 3)
    GRLX_LAUNCH_ADV(GRLX_ENV_ACROBOT, 3)
#undef GRLX_LAUNCH_ADV
    return hipErrorInvalidValue;
  }
  if (!inplace && !P.no_specialisation)
  { // compile-time specialised instantiations of the reference's cfg/pendulum/{sarsa,q}_tc.yaml family
#define GRLX_LAUNCH_SPEC(AGENT)                                                                                        \
    if (SpecPendulumTcA<AGENT>::matches(P))                                                                            \
    {                                                                                                                  \
      if (variant) *variant = GRLX_KERNEL_SPECIALISED;                                                                 \
      hipLaunchKernelGGL((rollout_kernel<GRLX_ENV_PENDULUM, 3, false, SpecPendulumTcA<AGENT>>), dim3(waves), dim3(64), 0, stream, P, n_trials); \
      return hipGetLastError();                                                                                        \
    }
    GRLX_LAUNCH_SPEC(GRLX_AGENT_SARSA)
    GRLX_LAUNCH_SPEC(GRLX_AGENT_Q)
    GRLX_LAUNCH_SPEC(GRLX_AGENT_EXPECTED_SARSA)
#undef GRLX_LAUNCH_SPEC
  }
  GRLX_LAUNCH(GRLX_ENV_PENDULUM, 3)
  GRLX_LAUNCH(GRLX_ENV_PENDULUM, 5)
  GRLX_LAUNCH(GRLX_ENV_ACROBOT, 3)
  GRLX_LAUNCH(GRLX_ENV_CART_POLE, 3)
  GRLX_LAUNCH(GRLX_ENV_COMPASS_WALKER, 3)
#undef GRLX_LAUNCH
  return hipErrorInvalidValue;
}

// -------------------------------------------------- fine-grained kernels ---
// Projector::project, batched: one lane per (row, tiling)
__global__ void project_kernel(TileParams tp, const double *in, int n, uint32_t *out)
{
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  int row = gid / tp.T, j = gid % tp.T;
  if (row >= n) return;
  out[(size_t)row * tp.T + j] = tile_slot_generic(tp, in + (size_t)row * tp.D, j);
}

hipError_t launch_project(const TileParams &tp, const double *in_dev, int n, uint32_t *out_dev, hipStream_t stream)
{
  long total = (long)n * tp.T;
  int blocks = (int)((total + 255) / 256);
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(project_kernel, dim3(blocks), dim3(256), 0, stream, tp, in_dev, n, out_dev);
  return hipGetLastError();
}

// Environment::step, batched: one lane per environment instance
template <int ENV>
__global__ void env_step_kernel(DevParams P, double *state, const double *action, int n,
                                double *obs, double *reward, int32_t *terminal, uint32_t *err)
{
  constexpr int S = Env<ENV>::S, D = Env<ENV>::D;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x[S], o[D], rw;
  int term;
  for (int k = 0; k < S; ++k) x[k] = state[(size_t)i * S + k];
  uint32_t st = 0;
  if (!Env<ENV>::in_domain(x)) st |= ST_DOMAIN;
  env_step<ENV>(P, x, action[i], o, rw, term, st);
  for (int k = 0; k < S; ++k) state[(size_t)i * S + k] = x[k];
  for (int k = 0; k < D; ++k) obs[(size_t)i * D + k] = o[k];
  reward[i] = rw;
  terminal[i] = term;
  bool bad = st != 0;
  for (int k = 0; k < S; ++k) bad = bad || (x[k] != x[k]);
  if (bad) atomicOr(err, ST_DOMAIN);
}

hipError_t launch_env_step(const DevParams &P, double *state_dev, const double *action_dev, int n,
                           double *obs_dev, double *reward_dev, int32_t *terminal_dev, uint32_t *err_dev, hipStream_t stream)
{
  int blocks = (n + 63) / 64;
  if (blocks == 0) return hipSuccess;
  switch (P.env)
  {
    case GRLX_ENV_PENDULUM:
      hipLaunchKernelGGL(env_step_kernel<GRLX_ENV_PENDULUM>, dim3(blocks), dim3(64), 0, stream, P, state_dev, action_dev, n,
                         obs_dev, reward_dev, terminal_dev, err_dev);
      break;
    case GRLX_ENV_ACROBOT:
      hipLaunchKernelGGL(env_step_kernel<GRLX_ENV_ACROBOT>, dim3(blocks), dim3(64), 0, stream, P, state_dev, action_dev, n,
                         obs_dev, reward_dev, terminal_dev, err_dev);
      break;
    case GRLX_ENV_CART_POLE:
      hipLaunchKernelGGL(env_step_kernel<GRLX_ENV_CART_POLE>, dim3(blocks), dim3(64), 0, stream, P, state_dev, action_dev, n,
                         obs_dev, reward_dev, terminal_dev, err_dev);
      break;
    case GRLX_ENV_COMPASS_WALKER:
      hipLaunchKernelGGL(env_step_kernel<GRLX_ENV_COMPASS_WALKER>, dim3(blocks), dim3(64), 0, stream, P, state_dev, action_dev, n,
                         obs_dev, reward_dev, terminal_dev, err_dev);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// Representation::read / write / update on rows applied in order (one wave; lane j = tiling j).
// op 0 read, 1 write(target, alpha), 2 update(delta)
__global__ __launch_bounds__(64) void table_op_kernel(DevParams P, int table, int op, const int32_t *replica, const uint32_t *idx, int n,
                                                      const double *arg, double alpha, double *out)
{
  __shared__ double   sh[64];
  __shared__ uint32_t shp[64];
  const int lane = threadIdx.x & 63;
  const int Tn = P.tile.T;
  uint32_t status = 0, inserted = 0;
  for (int i = 0; i < n; ++i)
  {
    const int r = replica[i];
    const Table tab = table_of(P, table, r);
    const bool act = lane < Tn;
    uint32_t slot = act ? idx[(size_t)i * Tn + lane] : 0u, pos = 0;
    const bool valid = act && slot != 0xFFFFFFFFu;     // invalid_index(): skipped by update (linear.cpp:207)
    double w = 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const LinearParams &lp = table == 1 ? P.lin_actor : P.lin;
    table_probe(tab, lp, P.states[r], table, valid, slot, pos, w, status, inserted);
    sh[lane] = w;
    shp[lane] = valid ? pos : kInvalidPos;
    wave_sync();
    double s = 0;
    for (int k = 0; k < Tn; ++k) s += sh[k];
    s /= Tn;
    s = clampd(s, lp.out_min, lp.out_max);
    if (op == 0)
    {
      if (lane == 0) out[i] = s;
    }
    else
    {
      double d = (op == 1) ? alpha * (arg[i] - s) : arg[i];
      if (valid)
      { // sequential semantics for duplicate slots: c additions, highest lane's value is final
        uint32_t c = 0;
        for (int k = 0; k <= lane; ++k) c += (shp[k] == pos) ? 1u : 0u;
        bool last = true;
        for (int k = lane + 1; k < Tn; ++k) last = last && (shp[k] != pos);
        double v = w;
        for (uint32_t cc = 0; cc < c; ++cc)
          v = lp.limit ? clampd(v + d, lp.out_min, lp.out_max) : v + d;
        if (last) value_store(tab, pos, v);
      }
    }
    wave_sync();
    uint32_t ins = inserted;
    for (int off = 32; off > 0; off >>= 1) ins += __shfl_xor(ins, off, 64);
    if (lane == 0 && ins) P.states[r].n_slots[table] += ins;
    inserted = 0;
  }
  uint32_t st = status;
  for (int off = 32; off > 0; off >>= 1) st |= __shfl_xor(st, off, 64);
  if (lane == 0 && st && n > 0) atomicOr(&P.states[replica[0]].status, st);
}

hipError_t launch_table_op(const DevParams &P, int table, int op, const int32_t *replica_dev, const uint32_t *idx_dev, int n,
                           const double *arg_dev, double alpha, double *out_dev, hipStream_t stream)
{
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(table_op_kernel, dim3(1), dim3(64), 0, stream, P, table, op, replica_dev, idx_dev, n, arg_dev, alpha, out_dev);
  return hipGetLastError();
}

// current weight of reference slots (lazy value when the slot was never touched; no insertion)
__global__ void get_weights_kernel(DevParams P, int table, int replica, const uint32_t *slots, int n, double *out)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Table tab = table_of(P, table, replica);
  uint32_t slot = slots[i];
  uint32_t b = table_home(tab, slot);
  double v = initial_weight(P.states[replica], table, table == 1 ? P.lin_actor : P.lin, slot);
  for (int it = 0; it < kMaxProbe; ++it)
  {
    const BucketRegs br = bucket_load(tab, b);
    uint32_t empty;
    const int way = bucket_find(br.k, slot, empty);
    if (way >= 0) { v = (way == 0) ? br.v[0] : (way == 1) ? br.v[1] : (way == 2) ? br.v[2] : br.v[3]; break; }
    if (empty != 0u) break;
    b = (b + 1u) & tab.bmask;
  }
  out[i] = v;
}

// {action: load}: replicas [first, first+count) take `image` as the initial value of every slot of
// `table` and forget what they had learned in it (their sparse tables are cleared by the caller)
__global__ void set_lazy_base_kernel(DevParams P, int table, int first, int count, const double *image)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  ReplicaState &rs = P.states[first + i];
  rs.lazy_base[table] = image;
  rs.n_slots[table] = 0u;
}

hipError_t launch_set_lazy_base(const DevParams &P, int table, int first, int count, const double *image_dev, hipStream_t stream)
{
  if (count <= 0) return hipSuccess;
  hipLaunchKernelGGL(set_lazy_base_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, P, table, first, count, image_dev);
  return hipGetLastError();
}

// dense export of a table: out[slot] for every reference slot (grid-stride)
__global__ void export_weights_kernel(DevParams P, int table, int replica, double *out)
{
  const Table tab = table_of(P, table, replica);
  const LinearParams &lp = table == 1 ? P.lin_actor : P.lin;
  const uint32_t memory = (uint32_t)(table == 1 ? P.tile_actor.memory : P.tile.memory);
  for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < memory; slot += gridDim.x * blockDim.x)
  {
    uint32_t b = table_home(tab, slot);
    double v = 0;
    bool found = false;
    for (int it = 0; it < kMaxProbe; ++it)
    {
      const BucketRegs br = bucket_load(tab, b);
      uint32_t empty;
      const int way = bucket_find(br.k, slot, empty);
      if (way >= 0) { v = (way == 0) ? br.v[0] : (way == 1) ? br.v[1] : (way == 2) ? br.v[2] : br.v[3]; found = true; break; }
      if (empty != 0u) break;
      b = (b + 1u) & tab.bmask;
    }
    if (!found) v = initial_weight(P.states[replica], table, lp, slot);
    out[slot] = v;
  }
}

hipError_t launch_export_weights(const DevParams &P, int table, int replica, double *out_dev, hipStream_t stream)
{
  hipLaunchKernelGGL(export_weights_kernel, dim3(2048), dim3(256), 0, stream, P, table, replica, out_dev);
  return hipGetLastError();
}

hipError_t launch_get_weights(const DevParams &P, int table, int replica, const uint32_t *slots_dev, int n, double *out_dev, hipStream_t stream)
{
  int blocks = (n + 255) / 256;
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(get_weights_kernel, dim3(blocks), dim3(256), 0, stream, P, table, replica, slots_dev, n, out_dev);
  return hipGetLastError();
}

__global__ void math_kernel(int op, const double *x, const double *y, int n, double *out)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i], r;
  switch (op)
  {
    case 0: r = psin_checked(v); break;
    case 1: r = pcos_checked(v); break;
    case 2: r = plog(v); break;
    case 3: r = pfmod(v, y[i]); break;
    case 5: r = div6(v); break;
    case 6: r = psin_s(v, sin_consts<false>()); break;       // small-angle-aware forms (whole waves of small arguments take the short path)
    case 7: r = pcos_s(v, sin_consts<false>()); break;
    case 8: { double sn, cs; psincos_s(v, sin_consts<false>(), sn, cs); r = sn + cs; break; }
    default: r = __builtin_sqrt(v); break;
  }
  out[i] = r;
}

hipError_t launch_math(int op, const double *x, const double *y, int n, double *out, hipStream_t stream)
{
  int blocks = (n + 255) / 256;
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(math_kernel, dim3(blocks), dim3(256), 0, stream, op, x, y, n, out);
  return hipGetLastError();
}

__global__ void rand48_at_kernel(uint64_t x0, const uint64_t *skip, int n, double *out)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = lcg_double(lcg_next(lcg_jump(x0, skip[i])));
}

hipError_t launch_rand48_at(uint64_t x0, const uint64_t *skip, int n, double *out, hipStream_t stream)
{
  int blocks = (n + 255) / 256;
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(rand48_at_kernel, dim3(blocks), dim3(256), 0, stream, x0, skip, n, out);
  return hipGetLastError();
}

// learning-curve statistics over the replicas of this GPU: one block per row, fixed-order tree
__global__ __launch_bounds__(256) void curve_stats_kernel(DevParams P, int first, double *out)
{
  __shared__ double s1[256], s2[256];
  const int row = first + blockIdx.x;
  double a = 0, b = 0;
  for (int r = threadIdx.x; r < P.n_replicas; r += 256)
  {
    double v = P.row_reward[(size_t)row * P.n_replicas + r];
    a += v;
    b += v * v;
  }
  s1[threadIdx.x] = a;
  s2[threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1)
  {
    if ((int)threadIdx.x < off)
    {
      s1[threadIdx.x] += s1[threadIdx.x + off];
      s2[threadIdx.x] += s2[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
  {
    out[blockIdx.x * 3 + 0] = s1[0];
    out[blockIdx.x * 3 + 1] = s2[0];
    out[blockIdx.x * 3 + 2] = (double)P.n_replicas;
  }
}

hipError_t launch_curve_stats(const DevParams &P, int first, int count, double *out_dev, hipStream_t stream)
{
  if (count == 0) return hipSuccess;
  hipLaunchKernelGGL(curve_stats_kernel, dim3(count), dim3(256), 0, stream, P, first, out_dev);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void step_counts_kernel(DevParams P, uint64_t *out)
{
  __shared__ unsigned long long a[256], b[256], c[256];
  unsigned long long la = 0, lb = 0, lc = 0;
  for (int r = threadIdx.x; r < P.n_replicas; r += 256)
  {
    la += (unsigned long long)P.states[r].ss;
    lb += P.states[r].test_steps;
    lc |= P.states[r].status;
  }
  a[threadIdx.x] = la; b[threadIdx.x] = lb; c[threadIdx.x] = lc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1)
  {
    if ((int)threadIdx.x < off)
    {
      a[threadIdx.x] += a[threadIdx.x + off];
      b[threadIdx.x] += b[threadIdx.x + off];
      c[threadIdx.x] |= c[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = a[0]; out[1] = b[0]; out[2] = c[0]; }
}

hipError_t launch_step_counts(const DevParams &P, uint64_t *out_dev, hipStream_t stream)
{
  hipLaunchKernelGGL(step_counts_kernel, dim3(1), dim3(256), 0, stream, P, out_dev);
  return hipGetLastError();
}

} // namespace grlx
