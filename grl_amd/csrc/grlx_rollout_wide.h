// grlx_rollout_wide.h -- the rollout kernel of the discrete-action TD agents for batches that outnumber the SIMDs:
// one wave carries R = 4*B replicas (B sub-batches of four) instead of four.
// Part of the single translation unit grlx_kernels.hip (included there, in order; not self-contained).
//
// Why.  rollout_kernel gives a replica 16 lanes (lane = tiling); the per-replica scalar chain -- the RK4 integration,
// reward, observation: 40-90 % of a pass -- is then computed 16 times over.  That is free while there are no more
// waves than SIMDs (4096 replicas = 1024 waves), but 8192 or 16384 replicas just run two or four rounds of it.
// Here the ENVIRONMENT phase of a pass steps all R replicas of the wave at once (lane L integrates replica L mod R: the
// redundancy drops from 16x to 64/R), and the TABLE phase -- hashing, lookups, sums, sampler, TD update: lane = tiling,
// exactly the code of rollout_kernel -- runs B times, once per sub-batch.  Per replica the arithmetic and its order are
// unchanged (bit-identical results); a wave-pass costs E + B*Tb instead of B*(E + Tb).
//
// Replicas of a wave do not wait for each other at episode ends: each one runs its own sequence of trials (start,
// steps, drain of the pending update, row) and begins its next trial in the pass after its episode has drained, so a
// wave idles only at the very end of a launch -- episodes of the acrobot and the walker end at different steps.
//
// Between its turns a sub-batch's lane state (register trace, pending update, position of p) is PARKED in LDS
// (196 B per lane, lane-contiguous 16-byte quads: conflict-free ds_read/write_b128), its per-replica scalars (RNG
// streams, counters, action) in a small structure-of-arrays; observation / reward / terminal / action travel between
// the two lane roles through LDS as well.  B = 2: 25 KB of parked state per wave, four waves per CU fit.
#pragma once

namespace grlx {

struct WideLane {                 // table-role state of one lane for one sub-batch
  TraceRegs tr;
  bool      pd, pd_sh, p_sh;
  double    pd_dW, pd_dT, pd_wp;
  uint32_t  pd_pos, p_pos;
  uint32_t  status, inserted;
  // actor-critic only (AC = true): position of the actor's projection of s, its shared flag, slots created in the actor's table
  uint32_t  ap_pos, inserted2;
  bool      ap_sh;
  // the weights of p's slot(s) as looked up one pass ago (instead of loading them again): Q / critic table, actor table
  double    wp_seen, wap_seen;
};
constexpr int kWideQuads = 12;    // 16-byte quads per parked lane

__device__ __forceinline__ uint4 pack2d(double a, double b)
{
  const uint64_t ua = (uint64_t)__double_as_longlong(a), ub = (uint64_t)__double_as_longlong(b);
  return make_uint4((uint32_t)ua, (uint32_t)(ua >> 32), (uint32_t)ub, (uint32_t)(ub >> 32));
}
__device__ __forceinline__ void unpack2d(const uint4 &q, double &a, double &b)
{
  a = __longlong_as_double((long long)((uint64_t)q.x | ((uint64_t)q.y << 32)));
  b = __longlong_as_double((long long)((uint64_t)q.z | ((uint64_t)q.w << 32)));
}

// The parked form of a lane's state: kWideQuads 16-byte quads + three counters.  Held in LDS ([quad][lane]: conflict-free b128 accesses) for
// the first two sub-batches of a wave and in REGISTERS (WideRegPark) for the third and fourth: a wave that is alone on its SIMD has a hundred
// registers to spare, and what round 3 parked in device memory (13 KB per wave, sub-batch and pass, written and read back: 132 GB per launch
// of the actor-critic bench, 3 KB per env-step of the 16-replica walker waves) stays in the register file.
struct WideRegPark { uint4 q[kWideQuads]; uint32_t i[3]; };
template <bool AC = false>
__device__ __forceinline__ void wide_pack(const WideLane &c, WideRegPark &p)
{
  static_assert(kMaxTrace == 10 && kWideQuads == 12, "the parked layout holds ten trace entries");
  p.q[0] = make_uint4(c.tr.pos[0], c.tr.pos[1], c.tr.pos[2], c.tr.pos[3]);
  p.q[1] = make_uint4(c.tr.pos[4], c.tr.pos[5], c.tr.pos[6], c.tr.pos[7]);
  p.q[2] = make_uint4(c.tr.pos[8], c.tr.pos[9], c.tr.cnt2, c.tr.wt | (c.tr.dup ? 1u << 16 : 0u) | ((uint32_t)c.tr.len << 20));
#pragma unroll
  for (int k = 0; k < 5; ++k) p.q[3 + k] = pack2d(c.tr.val[2 * k], c.tr.val[2 * k + 1]);
  {
    const uint64_t ut = (uint64_t)__double_as_longlong(c.tr.total);
    p.q[8] = make_uint4((uint32_t)ut, (uint32_t)(ut >> 32), c.pd_pos, c.p_pos);
  }
  p.q[9] = pack2d(c.pd_dW, c.pd_dT);
  {
    const uint64_t uw = (uint64_t)__double_as_longlong(c.pd_wp);
    p.q[10] = make_uint4((uint32_t)uw, (uint32_t)(uw >> 32),
                         (c.pd ? 1u : 0u) | (c.pd_sh ? 2u : 0u) | (c.p_sh ? 4u : 0u) | ((AC && c.ap_sh) ? 8u : 0u), c.status);
  }
  p.q[11] = pack2d(c.wp_seen, c.wap_seen);
  p.i[0] = c.inserted;
  p.i[1] = AC ? c.inserted2 : 0u;
  p.i[2] = AC ? c.ap_pos : kInvalidPos;
}

// sh_ctx: [quad][lane] of this sub-batch; sh_ins: [lane] (AC: [3][lane])
template <bool AC = false>
__device__ __forceinline__ void wide_park(const WideLane &c, uint4 *sh_ctx, uint32_t *sh_ins, int lane)
{
  WideRegPark p;
  wide_pack<AC>(c, p);
#pragma unroll
  for (int k = 0; k < kWideQuads; ++k) sh_ctx[k * 64 + lane] = p.q[k];
  sh_ins[lane] = p.i[0];
  if (AC)
  {
    sh_ins[64 + lane] = p.i[1];
    sh_ins[128 + lane] = p.i[2];
  }
}

template <bool AC = false>
__device__ __forceinline__ void wide_unpark(WideLane &c, const uint4 *sh_ctx, const uint32_t *sh_ins, int lane)
{
  uint4 q = sh_ctx[0 * 64 + lane];
  c.tr.pos[0] = q.x; c.tr.pos[1] = q.y; c.tr.pos[2] = q.z; c.tr.pos[3] = q.w;
  q = sh_ctx[1 * 64 + lane];
  c.tr.pos[4] = q.x; c.tr.pos[5] = q.y; c.tr.pos[6] = q.z; c.tr.pos[7] = q.w;
  q = sh_ctx[2 * 64 + lane];
  c.tr.pos[8] = q.x; c.tr.pos[9] = q.y; c.tr.cnt2 = q.z;
  c.tr.wt = q.w & 0xFFFFu;
  c.tr.dup = ((q.w >> 16) & 1u) != 0u;
  c.tr.len = (int)(q.w >> 20);
#pragma unroll
  for (int k = 0; k < 5; ++k) unpack2d(sh_ctx[(3 + k) * 64 + lane], c.tr.val[2 * k], c.tr.val[2 * k + 1]);
  q = sh_ctx[8 * 64 + lane];
  c.tr.total = __longlong_as_double((long long)((uint64_t)q.x | ((uint64_t)q.y << 32)));
  c.pd_pos = q.z;
  c.p_pos = q.w;
  unpack2d(sh_ctx[9 * 64 + lane], c.pd_dW, c.pd_dT);
  q = sh_ctx[10 * 64 + lane];
  c.pd_wp = __longlong_as_double((long long)((uint64_t)q.x | ((uint64_t)q.y << 32)));
  c.pd = (q.z & 1u) != 0u;
  c.pd_sh = (q.z & 2u) != 0u;
  c.p_sh = (q.z & 4u) != 0u;
  c.status = q.w;
  unpack2d(sh_ctx[11 * 64 + lane], c.wp_seen, c.wap_seen);
  c.inserted = sh_ins[lane];
  c.ap_sh = AC && (q.z & 8u) != 0u;
  c.inserted2 = AC ? sh_ins[64 + lane] : 0u;
  c.ap_pos = AC ? sh_ins[128 + lane] : kInvalidPos;
}

// wide_unpark in two halves: the loads, and -- after whatever can be done without the lane state -- the unpacking
template <bool AC = false>
__device__ __forceinline__ void wide_unpark_load(uint4 *raw, uint32_t *rawi, const uint4 *ctx, const uint32_t *ins, int lane)
{
#pragma unroll
  for (int k = 0; k < kWideQuads; ++k) raw[k] = ctx[k * 64 + lane];
  rawi[0] = ins[lane];
  rawi[1] = AC ? ins[64 + lane] : 0u;
  rawi[2] = AC ? ins[128 + lane] : kInvalidPos;
}
template <bool AC = false>
__device__ __forceinline__ void wide_unpark_decode(WideLane &c, const uint4 *raw, const uint32_t *rawi)
{
  uint4 q = raw[0];
  c.tr.pos[0] = q.x; c.tr.pos[1] = q.y; c.tr.pos[2] = q.z; c.tr.pos[3] = q.w;
  q = raw[1];
  c.tr.pos[4] = q.x; c.tr.pos[5] = q.y; c.tr.pos[6] = q.z; c.tr.pos[7] = q.w;
  q = raw[2];
  c.tr.pos[8] = q.x; c.tr.pos[9] = q.y; c.tr.cnt2 = q.z;
  c.tr.wt = q.w & 0xFFFFu;
  c.tr.dup = ((q.w >> 16) & 1u) != 0u;
  c.tr.len = (int)(q.w >> 20);
#pragma unroll
  for (int k = 0; k < 5; ++k) unpack2d(raw[3 + k], c.tr.val[2 * k], c.tr.val[2 * k + 1]);
  q = raw[8];
  c.tr.total = __longlong_as_double((long long)((uint64_t)q.x | ((uint64_t)q.y << 32)));
  c.pd_pos = q.z;
  c.p_pos = q.w;
  unpack2d(raw[9], c.pd_dW, c.pd_dT);
  q = raw[10];
  c.pd_wp = __longlong_as_double((long long)((uint64_t)q.x | ((uint64_t)q.y << 32)));
  c.pd = (q.z & 1u) != 0u;
  c.pd_sh = (q.z & 2u) != 0u;
  c.p_sh = (q.z & 4u) != 0u;
  c.status = q.w;
  unpack2d(raw[11], c.wp_seen, c.wap_seen);
  c.inserted = rawi[0];
  c.ap_sh = AC && (q.z & 8u) != 0u;
  c.inserted2 = rawi[1];
  c.ap_pos = rawi[2];
}

// per-replica scalars of the table role, parked as a structure of arrays [field][replica in wave]
enum { WR_G = 0, WR_TL, WR_S1, WR_EPS, WR_TT, WR_SS, WR_TSTEPS, WR_TOTAL, WR_TIME, WR_ACTION, WR_FIELDS64 };
enum { WR_AIDX = 0, WR_FLAGS, WR_ROWS, WR_LEFT, WR_SUB, WR_FIELDS32 };
enum : uint32_t { WF_RUNNING = 1u, WF_FIRST = 2u, WF_TEST = 4u, WF_ENDING = 8u, WF_SERVED = 16u };

struct WideRep {
  uint64_t G, TL, S1;
  double   eps_decay;
  int64_t  tt, ss;
  uint64_t test_steps;
  double   total_reward, time, action;
  int      action_index;
  bool     running, first;
  bool     ending;              // the episode has seen its terminal state; its row is written once the pending update is applied
  int      test;
  uint32_t rows;
  int      trials_left;         // trials of this launch the replica has not finished yet
  int      sub_left;            // greedy episodes the running test trial still has to run after this one (test_trials, online_learning.cpp:161-170)
  bool     served;              // the replica still takes its steps from the environment server (grlx_env_server_wide.h) and sends it a command per pass
};

template <int R>
__device__ __forceinline__ void wide_rep_store(const WideRep &s, uint64_t *sh64, uint32_t *sh32, int q)
{
  sh64[WR_G * R + q] = s.G;
  sh64[WR_TL * R + q] = s.TL;
  sh64[WR_S1 * R + q] = s.S1;
  sh64[WR_EPS * R + q] = (uint64_t)__double_as_longlong(s.eps_decay);
  sh64[WR_TT * R + q] = (uint64_t)s.tt;
  sh64[WR_SS * R + q] = (uint64_t)s.ss;
  sh64[WR_TSTEPS * R + q] = s.test_steps;
  sh64[WR_TOTAL * R + q] = (uint64_t)__double_as_longlong(s.total_reward);
  sh64[WR_TIME * R + q] = (uint64_t)__double_as_longlong(s.time);
  sh64[WR_ACTION * R + q] = (uint64_t)__double_as_longlong(s.action);
  sh32[WR_AIDX * R + q] = (uint32_t)s.action_index;
  sh32[WR_FLAGS * R + q] = (s.running ? WF_RUNNING : 0u) | (s.first ? WF_FIRST : 0u) | (s.test ? WF_TEST : 0u) | (s.ending ? WF_ENDING : 0u) |
                           (s.served ? WF_SERVED : 0u);
  sh32[WR_ROWS * R + q] = s.rows;
  sh32[WR_LEFT * R + q] = (uint32_t)s.trials_left;
  sh32[WR_SUB * R + q] = (uint32_t)s.sub_left;
}

template <int R>
__device__ __forceinline__ void wide_rep_load(WideRep &s, const uint64_t *sh64, const uint32_t *sh32, int q)
{
  s.G = sh64[WR_G * R + q];
  s.TL = sh64[WR_TL * R + q];
  s.S1 = sh64[WR_S1 * R + q];
  s.eps_decay = __longlong_as_double((long long)sh64[WR_EPS * R + q]);
  s.tt = (int64_t)sh64[WR_TT * R + q];
  s.ss = (int64_t)sh64[WR_SS * R + q];
  s.test_steps = sh64[WR_TSTEPS * R + q];
  s.total_reward = __longlong_as_double((long long)sh64[WR_TOTAL * R + q]);
  s.time = __longlong_as_double((long long)sh64[WR_TIME * R + q]);
  s.action = __longlong_as_double((long long)sh64[WR_ACTION * R + q]);
  s.action_index = (int)sh32[WR_AIDX * R + q];
  const uint32_t f = sh32[WR_FLAGS * R + q];
  s.running = (f & WF_RUNNING) != 0u;
  s.first = (f & WF_FIRST) != 0u;
  s.test = (f & WF_TEST) ? 1 : 0;
  s.ending = (f & WF_ENDING) != 0u;
  s.served = (f & WF_SERVED) != 0u;
  s.rows = sh32[WR_ROWS * R + q];
  s.trials_left = (int)sh32[WR_LEFT * R + q];
  s.sub_left = (int)sh32[WR_SUB * R + q];
}

// B sub-batches of four replicas per wave.  Production ordering only (deferred TD update, no taps, no stamps): the
// diagnostic instantiations stay with rollout_kernel.
// SERVED: the environment steps come from the environment server of the wide kernels (grlx_env_server_wide.h) -- rollout_wide_served_kernel below.
template <int ENV, int NA, int B, typename SPEC, bool SERVED>
__device__ __forceinline__ void rollout_wide_body(const DevParams &P, int n_trials)
{
  static_assert(B == 2 || B == 3 || B == 4 || B == 8, "sub-batches per wave");
  static_assert(!SERVED || (B == 2 && NA == 3), "what the environment server of the wide kernels works for");
  constexpr int R = 4 * B;
  constexpr int NROWS = NA + 1;
  const DevParams &N = SPEC::numeric(P);
  constexpr int S = Env<ENV>::S, D = Env<ENV>::D, T = kLanesPerReplica;
  __shared__ double   sh_w[NROWS * 16 * 4];
  __shared__ uint32_t sh_ppos[4 * 16];
  __shared__ double   sh_fb[16 * 4];
  __shared__ uint32_t sh_fbflag[16 * 4];
  __shared__ uint32_t sh_mb[4 * NA * 16];
  __shared__ uint32_t sh_ms[4 * NA * 16];
  __shared__ uint32_t sh_mail[4];
  __shared__ uint64_t sh_jump6[kJump6Words];        // LCG jump table, 6-bit windows (lazy weight initialisation)
  __shared__ double   sh_res[4 * 16];
  // the sub-batches beyond the second park their lane state in registers (WideRegPark), not in LDS: four parked sub-batches would be 50 KB
  // per wave, two waves per CU instead of four
  constexpr bool GLP = B >= 3;
  constexpr bool MEMP = B > 4;                          // (32 replicas per wave: see below)
  constexpr int BP = MEMP ? 1 : GLP ? 2 : B;            // sub-batches parked in LDS (four waves per CU must keep fitting its 160 KB)
  __shared__ uint4    sh_ctx[BP * kWideQuads * 64];     // parked lane state
  __shared__ uint32_t sh_ins[BP * 64];
  __shared__ uint64_t sh_r64[WR_FIELDS64 * R];          // parked per-replica scalars
  __shared__ uint32_t sh_r32[WR_FIELDS32 * R];
  // exchange between the two lane roles, [field][replica in wave]
  __shared__ double   sh_x[S * R];                      // start state (table role -> environment role)
  __shared__ double   sh_obs[D * R];
  __shared__ double   sh_reward[R];
  __shared__ double   sh_act[R];                        // action to apply in the next environment phase
  __shared__ int      sh_term[R];
  __shared__ uint32_t sh_step[R];                       // 1: the replica takes an environment step in the next pass
  __shared__ uint32_t sh_est[R];                        // status bits raised by the environment role
  __shared__ uint32_t sh_aidx[R];                       // index of that action (the candidate the environment server holds for it)
  __shared__ uint32_t sh_srv[R];                        // 1: the replica is served by the environment server; 0: it integrates here

  jump_table6_to_lds(sh_jump6);
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, j = lane & 15;
  const unsigned long long gmask = 0xFFFFull << (16 * g);
  const int wave0 = blockIdx.x * R;                     // first replica of this wave

  // ---- environment role: lane L integrates replica wave0 + (L mod R)
  const int eq = lane % R;
  const bool elive = wave0 + eq < P.n_replicas;
  double x[S];
#pragma unroll
  for (int i = 0; i < S; ++i) x[i] = P.states[elive ? wave0 + eq : 0].x[i];
  uint32_t estatus = 0;
  // the 64/R lanes of a replica share out the independent sin/cos evaluations of its equations of motion (grlx_envs.h)
  LaneShare lshare;
  lshare.src[0] = eq; lshare.src[1] = eq + R; lshare.src[2] = eq + 2 * R;
  lshare.role3 = (lane / R) % 3; lshare.role2 = (lane / R) & 1;
  static_assert(64 / R >= (ENV == GRLX_ENV_COMPASS_WALKER ? 2 : 3), "the lanes of a replica share out an equation of motion's sines (walker: two, else three)");
  if (lane < R)
  {
    sh_step[lane] = 0u; sh_est[lane] = 0u; sh_term[lane] = 0; sh_reward[lane] = 0; sh_act[lane] = 0;
    sh_aidx[lane] = 0u;
    sh_srv[lane] = (SERVED && P.env_mail != nullptr && wave0 + lane < P.n_replicas) ? 1u : 0u;
  }
  unsigned long long pass = 0;                          // passes of this wave so far = the sequence number of the commands it sends
  if constexpr (SERVED) mail_setprio(P.env_tune & 3u);

  // ---- table role: lane (g, j) of sub-batch b serves tiling j of replica wave0 + 4b + g
  UpdateParams up;
  up.out_min = N.lin.out_min;
  up.out_max = N.lin.out_max;
  up.limit = N.lin.limit != 0;
  up.ee = N.gl;
  up.cut = (N.trace_kind == GRLX_TRACE_REPLACING) ? 0.01 : 0.0001;
  up.use_trace = N.trace_kind == GRLX_TRACE_REPLACING;
  up.dW = up.dT = 0;

  double acts[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) acts[a] = N.actions[a];
  uint32_t key_act[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a)
    key_act[a] = in_reg(murmur_key(tile_coord<T>(N.tile, D, tile_quant(N.tile, D, N.actions[a]), j)));
  const uint32_t key_j = in_reg(murmur_key(j));

  WideRegPark rp2, rp3;                                 // GLP: the parked state of sub-batch 2 / 3 (b is wave-uniform: the branches below are scalar)
  // MEMP (B = 8, 32 replicas per wave): sub-batch 0 parks in LDS, 2 and 3 in registers, 1 and 4..7 in device memory (P.park: kAcParkBytes per
  // wave and sub-batch, written at the end of a turn, requested again one turn before it is unpacked)
  constexpr int kParkQuads = (int)(kAcParkBytes / sizeof(uint4));
  constexpr int kMemAreas = MEMP ? B - 3 : 0;
  uint4 *gl_base = MEMP ? (uint4 *)P.park + (size_t)blockIdx.x * (size_t)(kParkQuads * kMemAreas) : nullptr;
  auto in_memory = [&](int b) __attribute__((always_inline)) { return MEMP && (b == 1 || b >= 4); };
  auto gl_ctx = [&](int b) __attribute__((always_inline)) { return gl_base + (b == 1 ? 0 : b - 3) * kParkQuads; };
  auto gl_ins = [&](int b) __attribute__((always_inline)) { return (uint32_t *)(gl_ctx(b) + kWideQuads * 64); };
  auto park_state = [&](const WideLane &c, int b) __attribute__((always_inline)) {
    if (GLP && b == 2) wide_pack(c, rp2);
    else if (GLP && b == 3) wide_pack(c, rp3);
    else if (in_memory(b)) wide_park(c, gl_ctx(b), gl_ins(b), lane);
    else wide_park(c, sh_ctx + b * kWideQuads * 64, sh_ins + b * 64, lane);
  };
  auto unpark_state = [&](WideLane &c, int b) __attribute__((always_inline)) {
    if (GLP && b == 2) wide_unpark_decode(c, rp2.q, rp2.i);
    else if (GLP && b == 3) wide_unpark_decode(c, rp3.q, rp3.i);
    else if (in_memory(b)) wide_unpark(c, gl_ctx(b), gl_ins(b), lane);
    else wide_unpark(c, sh_ctx + b * kWideQuads * 64, sh_ins + b * 64, lane);
  };
  for (int b = 0; b < B; ++b)
  { // initial parked state of every sub-batch
    const int q = 4 * b + g;
    const bool live = wave0 + q < P.n_replicas;
    const ReplicaState &RS = P.states[live ? wave0 + q : 0];
    WideLane c;
    trace_init(c.tr);
    c.pd = c.pd_sh = c.p_sh = false;
    c.pd_dW = c.pd_dT = c.pd_wp = 0;
    c.pd_pos = c.p_pos = kInvalidPos;
    c.status = RS.status;
    c.inserted = 0;
    c.ap_pos = kInvalidPos; c.inserted2 = 0; c.ap_sh = false;
    c.wp_seen = 0; c.wap_seen = 0;
    park_state(c, b);
    WideRep s;
    s.G = RS.G; s.TL = RS.TL; s.S1 = RS.S1;
    s.eps_decay = RS.eps_decay;
    s.tt = RS.tt; s.ss = RS.ss;
    s.test_steps = RS.test_steps;
    s.total_reward = 0; s.time = 0; s.action = 0;
    s.action_index = 0;
    s.running = false; s.first = true; s.test = 0;
    s.ending = false;
    s.rows = RS.rows;
    s.trials_left = (live && !(P.steps_budget != 0u && (uint64_t)RS.ss >= P.steps_budget)) ? n_trials : 0;
    s.sub_left = 0;
    s.served = SERVED && P.env_mail != nullptr && live;
    if constexpr (SERVED)
      if (s.served && s.trials_left <= 0)
      { // nothing to do in this launch (its steps budget was reached before): the server does not wait for this replica
        if (j == 0) wide_mail_send<ENV>(P, wave0 + q, 1u, kMailExit);
        s.served = false;
      }
    wide_rep_store<R>(s, sh_r64, sh_r32, q);
  }
  wave_sync();

  {
    for (;;)
    {
      ++pass;
#ifdef GRLX_WIDE_STAMPS
      const unsigned long long st0 = stamp();
#endif
      // ================= environment phase: every replica of the wave that is in mid-episode takes its step;
      // a replica whose trial has just started takes its start state over from the table role instead (sh_step = 2)
      {
        const uint32_t todo = elive ? sh_step[eq] : 0u;
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
          bool got = false;
          if constexpr (SERVED)
          { // the step was integrated by the environment server while this wave was in its table phases: the candidate of the action taken
            // (command `pass - 1` named it; the candidates it selects from are the ones of command `pass - 2`)
            const bool want = step && sh_srv[eq] != 0u;
            if (__any(want))
            {
              double xn[S], rw = 0;
              got = wide_mail_take<ENV, R>(P, elive ? wave0 + eq : 0, want, pass - 2u, (int)sh_aidx[eq], lane, xn, rw);
              if (want && !got) sh_srv[eq] = 0u;      // no answer: this replica integrates here from now on (its table lanes send kMailExit)
              if (got)
              {
                terminal = Env<ENV>::observe(N, xn, obs);                     // env_step: observe, domain check
                reward = rw;
                if (!Env<ENV>::in_domain(xn)) estatus |= ST_DOMAIN;
#pragma unroll
                for (int i = 0; i < S; ++i) x[i] = xn[i];
              }
            }
          }
          const bool local = step && !got;
          if (__any(local))
          {
            if (local)
            {
              const double action = sh_act[eq];
              env_step<ENV, true, LaneShare>(N, x, action, obs, reward, terminal, estatus, lshare);     // online_learning.cpp:196
            }
          }
          if (step)
          { // the 64/R lanes of a replica hold identical values: all of them store (same address, same bits)
#pragma unroll
            for (int i = 0; i < D; ++i) sh_obs[i * R + eq] = obs[i];
            sh_reward[eq] = reward;
            sh_term[eq] = terminal;
          }
        }
        wave_sync();
      }

#ifdef GRLX_WIDE_STAMPS
      const unsigned long long st1 = stamp();
      if (P.diag_out && lane == 0) { P.diag_out[(size_t)blockIdx.x * 8 + 0] += st1 - st0; P.diag_out[(size_t)blockIdx.x * 8 + 2] += 1; }
#endif
      // ================= table phase, one sub-batch after the other
      bool more = false;
      WideRegPark nxt;                                  // MEMP: the parked state of the sub-batch after the running one, on its way from memory
      for (int b = 0; b < B; ++b)
      {
        const int q = 4 * b + g;
        const bool live = wave0 + q < P.n_replicas;
        const int r = live ? wave0 + q : 0;
        const ReplicaState &RS = P.states[r];
        const Table tab = table_of(P, 0, r);
        WideLane c;
        if (in_memory(b)) wide_unpark_decode(c, nxt.q, nxt.i);               // requested during the previous sub-batch's turn
        else unpark_state(c, b);
        if (b + 1 < B && in_memory(b + 1)) wide_unpark_load(nxt.q, nxt.i, gl_ctx(b + 1), gl_ins(b + 1), lane);
        WideRep s;
        wide_rep_load<R>(s, sh_r64, sh_r32, q);
        if (!__any(s.running || c.pd || s.trials_left > 0)) continue;        // this sub-batch has finished its trials

        uint32_t slot[NA];
        Lookup lk[NA];
        BucketRegs br[NA];
        double wp = 0;
        bool has_next = false, update = false;
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
            s.total_reward += reward;                                          // online_learning.cpp:202
            s.time += 1;                                                       // tau = 1
          }
          has_next = s.first || terminal != 2;
          update = !s.first && !s.test;
          if (has_next)
          { // policy: projections of Q(s', .) (q.cpp:94-107)
            uint32_t hpre = 449u ^ (uint32_t)(D + 2);
#pragma unroll
            for (int i = 0; i < D; ++i)
              hpre = murmur_mix(hpre, tile_coord<T>(N.tile, i, tile_quant(N.tile, i, obs[i]), j));
            const uint32_t hpm = hpre * 0x5bd1e995u;
#pragma unroll
            for (int a = 0; a < NA; ++a)
            {
              uint32_t h = hpm ^ key_act[a];
              h = murmur_absorb(h, key_j);
              const uint32_t hm = murmur_final(h), mem = (uint32_t)N.tile.memory;
              slot[a] = ((mem & (mem - 1u)) == 0u) ? (hm & (mem - 1u)) : (hm % mem);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          if (update) wp = c.wp_seen;          // as looked up one pass ago; reconciled with the trace / eviction / shared slots below
          if (has_next) table_issue<NA>(tab, slot, lk, br);
        }

        // the PREVIOUS step's predictor update, in the shadow of the loads just issued (see rollout_kernel)
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
          td_update_lane<true>(c.tr, tab, up, c.pd_pos, c.pd_sh, c.pd_wp, g, j, sh_ppos, sh_fb, sh_fbflag, c.status, ev);
          c.pd = false;
        }

        if (s.running)
        {
          double qv[NA];
          uint32_t pos[NA];
          double w[NA];
          bool sh[NA];
#pragma unroll
          for (int a = 0; a < NA; ++a) { qv[a] = 0; pos[a] = kInvalidPos; w[a] = 0; sh[a] = false; }
          if (has_next)
          {
            bool shared_event = false;
            table_get_finish<NA, 2>(tab, N.lin, RS, 0, slot, lk, br, pos, w, sh, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump6,
                                 c.status, c.inserted,
                                 [&](uint32_t mp) {
                                   if (ev.pos != kInvalidPos && ev.pos == mp) value_store(tab, mp, ev.val);
                                   trace_share_event(c.tr, tab, mp);
                                   if (c.p_pos == mp) c.p_sh = true;
                                   shared_event = true;
                                 });
            if (rarely(__any(shared_event)) && update) wp = value_load(tab, c.p_pos);
          }
          {
            bool risky = ev.n > 1u || (update && c.p_sh);
#pragma unroll
            for (int a = 0; a < NA; ++a) risky = risky || (has_next && sh[a]);
            if (rarely(__any(risky)))
            {
#pragma unroll
              for (int a = 0; a < NA; ++a)
                if (has_next) w[a] = value_load(tab, pos[a]);
              if (update) wp = value_load(tab, c.p_pos);
            }
            const bool held = ev.pos != kInvalidPos;
#pragma unroll
            for (int a = 0; a < NA; ++a) w[a] = (held && pos[a] == ev.pos) ? ev.val : w[a];
            wp = (held && c.p_pos == ev.pos) ? ev.val : wp;
          }
          if (has_next)
          {
#pragma unroll
            for (int a = 0; a < NA; ++a)
            {
              w[a] = trace_forward(c.tr, pos[a], w[a]);
              SHW(a, j, g) = w[a];
            }
          }
          if (update)
          {
            wp = trace_forward(c.tr, c.p_pos, wp);
            SHW(NA, j, g) = wp;
          }
          wave_sync();
          { // LinearRepresentation::read (linear.cpp:136-184): lane r of the replica sums row r in the reference's order
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
            for (int a = 0; a < NA; ++a) qv[a] = clampd(sh_res[g * 16 + a], up.out_min, up.out_max);
          }
          double qsa = 0;
          if (update) qsa = clampd(sh_res[g * 16 + NA], up.out_min, up.out_max);

          // sampler (greedy.cpp:63-86, 144-218)
          int a_next = 0;
          int mai = 0, man = 1;
          double best = 0;
          if (has_next)
          {
            findmax<NA>(qv, mai, man, best);
            if (s.test)
              a_next = (man > 1) ? tie_break<NA>(qv, best, man, s.G) : mai;
            else
            {
              if (s.time == 0.) s.eps_decay = fmax(s.eps_decay * N.decay_rate, N.decay_min);
              s.S1 = lcg_next(s.S1);
              const double rnd = lcg_double(s.S1);
              if (rnd < s.eps_decay * N.epsilon)
              {
                s.G = lcg_next(s.G);
                a_next = (int)(lcg_long(s.G) % (uint32_t)NA);
              }
              else
                a_next = (man > 1) ? tie_break<NA>(qv, best, man, s.G) : mai;
            }
          }

          // predictor update (sarsa.cpp:98-124 / advantage.cpp:71-110), queued for the next pass
          if (update)
          {
            double target = reward;
            if (has_next)
            {
              if (SPEC::agent(P) == GRLX_AGENT_SARSA)
                target += N.gamma * pick<double, NA>(qv, a_next);
              else if (SPEC::agent(P) == GRLX_AGENT_EXPECTED_SARSA)
              {
                const double de = s.eps_decay * N.epsilon;
                double v = 0;
#pragma unroll
                for (int kk = 0; kk < NA; ++kk)
                {
                  double d = (qv[kk] == best) ? 1. / man : 0.;
                  if (d == 1) d = 1 - de;
                  d += de / NA;
                  v += qv[kk] * d;
                }
                target += N.gamma * v;
              }
              else
              {
                double v = -__builtin_inf();
#pragma unroll
                for (int kk = 0; kk < NA; ++kk) v = fmax(v, qv[kk]);
                target += N.gamma * v;
              }
            }
            const double delta = target - qsa;
            c.pd = true;
            c.pd_dW = N.alpha * (target - qsa);
            c.pd_dT = N.alpha * delta;
            c.pd_pos = c.p_pos;
            c.status |= (c.p_pos == kInvalidPos) ? ST_BAD_POS : 0u;
            c.pd_sh = c.p_sh;
            c.pd_wp = wp;
          }

          // bookkeeping
          if (!s.first)
          {
            if (s.test) s.test_steps++;
            else s.ss++;                                                       // online_learning.cpp:218
          }
          if (has_next)
          {
            s.action_index = a_next;
            s.action = pick<double, NA>(acts, a_next);                         // discretizer_->at(index)
            c.p_pos = pick<uint32_t, NA>(pos, a_next);
            c.p_sh = pick<bool, NA>(sh, a_next);
            c.wp_seen = pick<double, NA>(w, a_next);
          }
          if (!s.first && terminal) { s.running = false; s.ending = true; }
          s.first = false;
        }
        if (ev.pos != kInvalidPos) value_store(tab, ev.pos, ev.val);

        // ---- between trials: the episode has ended and its last update has been applied (or none was pending)
        uint32_t step_next = s.running ? 1u : 0u;
        const bool between = !s.running && !c.pd && s.trials_left > 0;
        if (__any(between))
        {
          bool again = false;                                 // another greedy episode of the same test trial follows (test_trials)
          if (between && s.ending && s.test && s.sub_left > 0)
          {
            s.sub_left--;
            s.ending = false;
            again = true;
          }
          if (between && s.ending)
          { // end of the trial: write the cached weights back (test trials and the host read the table); the row
            if (!s.test) trace_flush(c.tr, tab, true);       // the next TDAgent::start clears the trace (td.cpp:54)
            const int ti = N.test_interval;
            if (ti >= 0 ? s.test : 1)
            { // online_learning.cpp:238-262
              if (s.rows < (uint32_t)P.max_rows)
              {
                if (j == 0)
                {
                  const size_t at = (size_t)s.rows * (size_t)P.n_replicas + (size_t)r;
                  const double sub = (s.test && P.test_trials > 1) ? (double)P.test_trials : 1.;     // :224-225
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
            if (P.steps_budget != 0u && (uint64_t)s.ss >= P.steps_budget) s.trials_left = 0;      // online_learning.cpp:154: `ss < steps_`
            s.ending = false;
          }
          if (between && (s.trials_left > 0 || again))
          { // start of the next trial (the start state may draw from the replica's RNG streams)
            const int ti = N.test_interval;
            if (!again)
            {
              s.test = (ti >= 0 && s.tt % (ti + 1) == ti) ? 1 : 0;           // online_learning.cpp:160
              s.sub_left = s.test ? P.test_trials - 1 : 0;
            }
            double xs[S], ob0[D];
            Env<ENV>::start(N, s.test, s.TL, s.G, xs);                       // modeled.cpp:132-158
            Env<ENV>::observe(N, xs, ob0);
#pragma unroll
            for (int i = 0; i < S; ++i) sh_x[i * R + q] = xs[i];
#pragma unroll
            for (int i = 0; i < D; ++i) sh_obs[i * R + q] = ob0[i];
            if constexpr (SERVED)
              if (s.served && live)
              { // the environment server starts on the first step of this trial, for every action, while this wave looks up Q(s0, .)
#pragma unroll
                for (int i = 0; i < S; ++i)
                  if (j == i) wide_mail_reset_unit<ENV>(P, r, pass, i, xs[i]);
              }
            if (!again)
            { // (reward and time keep adding up across the episodes of one test trial, :202-203)
              s.total_reward = 0;
              s.time = 0;
            }
            s.action = 0;
            s.action_index = 0;
            s.running = true;
            s.first = true;
            step_next = 2u;                                                  // environment role: take sh_x over
          }
        }

        // hand the action to the environment role and park
        if (j == 0 && live)
        {
          sh_act[q] = s.action;
          sh_aidx[q] = (uint32_t)s.action_index;
          sh_step[q] = step_next;
        }
        if constexpr (SERVED)
          if (s.served)
          { // this pass's command to the environment server: exactly one per replica and pass
            const bool fin = !s.running && !c.pd && s.trials_left <= 0;
            const bool lost = sh_srv[q] == 0u;
            const unsigned op = (fin || lost) ? kMailExit : step_next == 1u ? (unsigned)s.action_index : step_next == 2u ? kMailReset : kWideSkip;
            if (j == 0 && live) wide_mail_send<ENV>(P, r, pass, op);
            if (fin || lost) s.served = false;
          }
        more = more || s.running || c.pd || s.trials_left > 0;
        wide_rep_store<R>(s, sh_r64, sh_r32, q);
        park_state(c, b);
      }
      wave_sync();
#ifdef GRLX_WIDE_STAMPS
      if (P.diag_out && lane == 0) P.diag_out[(size_t)blockIdx.x * 8 + 1] += stamp() - st1;
#endif
      if (!__any(more)) break;
    }

  }

  // ---- write the replicas back
  if (elive && lane < R)
  {
    ReplicaState &RS = P.states[wave0 + eq];
#pragma unroll
    for (int i = 0; i < S; ++i) RS.x[i] = x[i];
    sh_est[eq] = estatus;
  }
  wave_sync();
  for (int b = 0; b < B; ++b)
  {
    const int q = 4 * b + g;
    const bool live = wave0 + q < P.n_replicas;
    WideLane c;
    unpark_state(c, b);
    WideRep s;
    wide_rep_load<R>(s, sh_r64, sh_r32, q);
    uint32_t ins = c.inserted;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) ins += __shfl_xor(ins, off, 16);
    uint32_t st = c.status | sh_est[q];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) st |= __shfl_xor(st, off, 16);
    if (SERVED && live && j == 0 && P.env_mail)      // (grlx_env_server_counts: served to the end / fell back)
      wide_mail_of<ENV>(P, wave0 + q)->stats[15] = sh_srv[q] ? 1u : 2u;
    if (live && j == 0)
    {
      ReplicaState &RS = P.states[wave0 + q];
      RS.G = s.G;
      RS.TL = s.TL;
      RS.S1 = s.S1;
      RS.eps_decay = s.eps_decay;
      RS.tt = s.tt;
      RS.ss = s.ss;
      RS.test_steps = s.test_steps;
      RS.n_slots[0] += ins;
      RS.rows = s.rows;
      RS.status = st;
    }
  }
}

template <int ENV, int NA, int B, typename SPEC>
__global__ __launch_bounds__(64) void rollout_wide_kernel(DevParams P, int n_trials)
{
  rollout_wide_body<ENV, NA, B, SPEC, false>(P, n_trials);
}

// The instantiation the environment server of the wide kernels works for (grlx_env_server_wide.h): the same body, its environment
// phase fetching what the server integrated.  Its registers leave room for the server's wave on the same SIMD.
template <int ENV, typename SPEC>
__global__ __launch_bounds__(64) void rollout_wide_served_kernel(DevParams P, int n_trials)
{
  rollout_wide_body<ENV, 3, 2, SPEC, true>(P, n_trials);
}

} // namespace grlx
