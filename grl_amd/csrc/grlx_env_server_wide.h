// grlx_env_server_wide.h -- the environment server of the WIDE rollout kernels (acrobot, compass walker; 8 replicas per wave): the
// environment phase of rollout_wide_kernel moved to a second, co-resident kernel whose lanes are (replica, action) pairs.
// Part of the single translation unit grlx_kernels.hip (included there, in order; not self-contained).
//
// Why.  A wide wave's pass is E + 2 T: ONE environment phase for its 8 replicas (one instruction stream whatever the number of
// lanes it serves: 8 distinct replicas x 2-3 shared sin/cos roles on 64 lanes) and two table phases.  E is half of the pass on
// the acrobot and the walker (27.8 k of 56.5 k cycles; 58.8 k of 120 k), and the wave issues on 0.57-0.60 of its cycles.  The
// table phase cannot start before the step it looks at is integrated -- but WHICH step that is depends on one of three actions
// only.  So a small kernel (one 64-thread block per rollout wave, resident beside it: rollout_wide_kernel holds 336 of a SIMD's
// 512 registers) integrates the next control step of every replica of the wave for all three actions while the rollout wave is
// in its table phases: lane (role, action, replica), 2 x 3 x 8 = 48 lanes, the two lanes of a pair sharing the sin/cos
// evaluations of the equations of motion (PairShare).  The rollout wave's environment phase shrinks to fetching the candidate of
// the action its sampler chose.  Same env_step on the same arguments: same bits.
//
// Protocol (one WideMail per replica, 16-byte units {value, seq} as in grlx_env_server.h: a unit is the old or the new one and
// says which).  A rollout wave counts its passes; in pass p it sends EVERY replica exactly one command, cmd[p & 7] = p << 8 | op:
//   op 0..2  "the step of the next pass is taken with action op"        (the replica is in mid-episode)
//   kReset   "a trial starts in reset[] (units tagged p)"               (the next pass takes the start state, no step)
//   kSkip    "nothing to integrate" (episode over, update draining, trials finished for this launch)
//   kExit    "no further command" (the replica has finished its trials, or goes on without the server)
// The server works in lock step with that: iteration p waits until all its replicas show command p, then ONE instruction
// stream: base state of a replica = the reset state, or the candidate its lanes (op, .) hold from iteration p - 1; all three
// lanes integrate one control step from it and store cand[p & 1][a] = units {x[0..S), reward} tagged p.  The rollout wave's
// environment phase of pass p + 2 -- the pass after the one whose table phase chose the action `a` sent in command p + 1 --
// fetches cand[p & 1][a] and expects tag p; observation, terminal flag and domain check are recomputed from the state (cheap
// for these tasks).  The server therefore has a whole pass (two table phases) per stream.
// Neither side waits for the other beyond a bound: a fetch that is not answered within kFetchPolls polls is integrated by the
// rollout wave itself (its environment lanes keep every replica's state), the replica sends kExit and integrates itself for the
// rest of the launch; the server leaves when all its replicas have sent kExit or after kServerIdlePolls polls without a full set
// of commands.
#pragma once

namespace grlx {

constexpr unsigned kWideSkip = 5u;                  // (kMailReset = 3, kMailExit = 4: grlx_env_server.h)
constexpr int kWideUnitsMax = 12;                   // state (<= 11) + reward

template <int ENV>
struct __attribute__((aligned(256))) WideMail {
  static constexpr int S = Env<ENV>::S;
  unsigned long long cmd[8];                        //   0  (a ring of eight: a wave whose replicas are all between trials runs a few passes ahead of the server)
  unsigned long long pad0[8];                       //  64
  unsigned long long stats[16];                     // 128  ([15]: served to the end 1 / fell back 2, at kEnvMailFlagOffset as in EnvMail)
  MailUnit reset[S];                                // 256
  MailUnit cand[2][3][S + 1];
};
static_assert(sizeof(WideMail<GRLX_ENV_COMPASS_WALKER>) <= kWideMailBytes && sizeof(WideMail<GRLX_ENV_ACROBOT>) <= kWideMailBytes, "one mailbox = kWideMailBytes");
static_assert(offsetof(WideMail<GRLX_ENV_ACROBOT>, stats) + 15 * sizeof(unsigned long long) == kEnvMailFlagOffset, "the flag grlx_env_server_counts reads");

template <int ENV>
__device__ __forceinline__ WideMail<ENV> *wide_mail_of(const DevParams &P, int r)
{
  return reinterpret_cast<WideMail<ENV> *>(reinterpret_cast<char *>(P.env_mail) + (size_t)r * kWideMailBytes);
}

// The mailboxes of a launch as one buffer resource; a unit = one 16-byte buffer load with device scope (sc1), issued and waited for by
// the compiler (several in flight together).  `tag` receives the unit's sequence number.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wide_rsrc(const DevParams &P)
{
  return __builtin_amdgcn_make_buffer_rsrc((void *)P.env_mail, 0, (int)((size_t)P.n_replicas * kWideMailBytes), 0x00020000);
}
__device__ __forceinline__ double unit_get(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned long long &tag)
{
  const mail_u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);         // 16 = sc1: device scope
  tag = ((unsigned long long)d.w << 32) | d.z;
  return __longlong_as_double((long long)(((unsigned long long)d.y << 32) | d.x));
}
template <int ENV>
__device__ __forceinline__ unsigned wide_cand_off(int r, unsigned long long tag, int a, int unit)
{
  return (unsigned)r * (unsigned)kWideMailBytes + (unsigned)offsetof(WideMail<ENV>, cand) +
         ((((unsigned)tag & 1u) * 3u + (unsigned)a) * (unsigned)(Env<ENV>::S + 1) + (unsigned)unit) * (unsigned)sizeof(MailUnit);
}

// Rollout side, environment role (lane L: replica eq = L mod R, copy c = L / R, R = 8: eight lanes per replica).  Fetch the candidate
// {x[0..S), reward} of command `tag` followed by action `a` for the replicas flagged `want`: copy c loads units c and c + 8, the
// eight copies of a replica agree on whether all units carried `tag`, the values are handed round.  false: not there within the bound.
template <int ENV, int R>
__device__ __forceinline__ bool wide_mail_take(const DevParams &P, int r, bool want, unsigned long long tag, int a, int lane, double *xn, double &reward,
                                               unsigned long long *polled = nullptr)
{
  constexpr int S = Env<ENV>::S, U = S + 1;
  static_assert(R == 8 && U <= 16, "eight copies per replica, two units each");
  const int eq = lane % R, c = lane / R;
  const __amdgpu_buffer_rsrc_t rs = wide_rsrc(P);
  const unsigned o0 = wide_cand_off<ENV>(r, tag, a, c < U ? c : 0), o1 = wide_cand_off<ENV>(r, tag, a, c + 8 < U ? c + 8 : 0);
  double v0 = 0, v1 = 0;
  bool ok = false;
  // which lanes hold the copies of my replica: eq, eq + 8, ..., eq + 56
  const unsigned long long mine = 0x0101010101010101ull << eq;
  unsigned polls = 0;
  for (;;)
  {
    unsigned long long t0 = tag, t1 = tag;
    if (want)
    {
      v0 = unit_get(rs, o0, t0);
      v1 = unit_get(rs, o1, t1);
    }
    const bool good = !want || (t0 == tag && t1 == tag);
    const unsigned long long bad = __ballot(!good);
    ok = (bad & mine) == 0ull;
    if (bad == 0ull) break;                                   // every replica of the wave has its candidate
    if (++polls > (tag <= 2u ? kFirstFetchPolls : kFetchPolls)) break;
    asm volatile("" ::: "memory");                            // (the loads are issued again)
  }
  if (polled) *polled += polls;
  // hand the units round: unit i sits in copy (i mod 8), first or second slot
#pragma unroll
  for (int i = 0; i < S; ++i)
  {
    const double v = (i < 8) ? v0 : v1;
    xn[i] = lane_fetch(v, eq + 8 * (i & 7));
  }
  {
    const double v = (S < 8) ? v0 : v1;
    reward = lane_fetch(v, eq + 8 * (S & 7));
  }
  return want && ok;
}

// rollout side, ONE lane of the replica's table group
template <int ENV>
__device__ __forceinline__ void wide_mail_send(const DevParams &P, int r, unsigned long long seq, unsigned op)
{
  mail_store(&wide_mail_of<ENV>(P, r)->cmd[seq & 7u], (seq << 8) | op);
}
// ... lanes j < S of the group: the start state of a trial
template <int ENV>
__device__ __forceinline__ void wide_mail_reset_unit(const DevParams &P, int r, unsigned long long seq, int j, double v)
{
  unit_store(&wide_mail_of<ENV>(P, r)->reset[j], v, seq);
}

// The server: block b serves the 8 replicas of rollout wave b.  lane = 24 rho + 8 a + q: role rho (0, 1) of the pair that integrates
// action a of replica q; lanes 48..63 idle (kept in step, their states zero so that the small-angle tests of the walker's sines pass).
// PIN: the constants of the dynamics held in vector registers across the integration (where the register budget beside the rollout wave allows)
template <int ENV, int NA, typename SPEC, bool PIN>
__device__ __forceinline__ void env_server_wide_body(const DevParams &P)
{
  static_assert(NA == 3, "three actions per replica");
  constexpr int S = Env<ENV>::S, D = Env<ENV>::D, R = 8;
  const DevParams &N = SPEC::numeric(P);
  const int lane = threadIdx.x & 63;
  const int q = lane & 7, a = (lane >> 3) % NA, rho = lane / 24;
  const bool worker = lane < 48;
  const int r = blockIdx.x * R + q;
  bool done = !worker || r >= P.n_replicas || (P.env_tune & 64u) != 0;      // (64: tests -- a server that is not there)
  WideMail<ENV> *m = wide_mail_of<ENV>(P, done ? 0 : r);
  const double act = N.actions[a];
  PairShare share;
  share.src[0] = worker ? 8 * a + q : lane;
  share.src[1] = worker ? 24 + 8 * a + q : lane;
  share.role2 = worker ? rho : 0;
  unsigned long long expect = 1;
  double cx[S], creward = 0;
#pragma unroll
  for (int i = 0; i < S; ++i) cx[i] = 0;
  unsigned idle = 0;
  bool seen = false;
  mail_setprio((P.env_tune >> 2) & 3u);
  const unsigned long long quit_after = (P.env_tune >> 8) & 0xFFFFu;      // (tests: the server leaves, unannounced, after that many passes)
  for (;;)
  {
    if (quit_after != 0u && expect > quit_after) done = true;
    if (__all(done)) break;
    unsigned long long word = 0;
    bool ready = false;
    if (!done)
    {
      word = mail_load(&m->cmd[expect & 7u]);
      ready = (word >> 8) == expect;
      if ((word >> 8) > expect) done = true;            // the ring has moved on: this replica's wave is not in step with this server any more
    }
    if (!__all(ready || done))
    {
      __builtin_amdgcn_s_sleep(1);
      if (++idle > (seen ? kServerIdlePolls : kServerStartPolls)) break;
      continue;
    }
    if (__all(done)) break;
    seen = true;
    idle = 0;
    // ---- all replicas of the wave have their command of pass `expect`
    const unsigned op = done ? kWideSkip : (unsigned)(word & 0xFFu);
    if (!done && op == kMailExit) done = true;
    bool compute = !done && op < (unsigned)NA;
    double base[S];
    // the candidate the action taken leaves: held by the lanes (., op, q)
    const int src = (op < (unsigned)NA) ? 8 * (int)op + q : lane;
#pragma unroll
    for (int i = 0; i < S; ++i) base[i] = lane_fetch(cx[i], src);
    if (rarely(__any(!done && op == kMailReset)))
    { // a trial starts: every lane of the replica reads the start state (units tagged with this pass; all S loads in flight together)
      const bool rs_ = !done && op == kMailReset;
      const __amdgpu_buffer_rsrc_t rsrc = wide_rsrc(P);
      const unsigned off = (unsigned)(done ? 0 : r) * (unsigned)kWideMailBytes + (unsigned)offsetof(WideMail<ENV>, reset);
      bool have = !rs_;
      for (unsigned tries = 0; tries < kFetchPolls && !__all(have); ++tries)
      {
        if (rs_ && !have)
        {
          bool all = true;
#pragma unroll
          for (int i = 0; i < S; ++i)
          {
            unsigned long long tag = 0;
            base[i] = unit_get(rsrc, off + (unsigned)i * (unsigned)sizeof(MailUnit), tag);
            all = all && tag == expect;
          }
          have = all;
        }
        asm volatile("" ::: "memory");
      }
      if (rs_ && !have) done = true;                    // the start state never came: leave this replica to its wave
      compute = compute || (rs_ && have);
    }
    if (__any(compute))
    {
      double nx[S], obs[D], rw = 0;
      int terminal = 0;
      uint32_t st = 0;
#pragma unroll
      for (int i = 0; i < S; ++i) nx[i] = compute ? base[i] : cx[i];
      env_step<ENV, PIN, PairShare>(N, nx, act, obs, rw, terminal, st, share);     // online_learning.cpp:196, for every action
      if (compute)
      {
#pragma unroll
        for (int i = 0; i < S; ++i) cx[i] = nx[i];
        creward = rw;
        if (rho == 0)
        {
          MailUnit *u = &m->cand[expect & 1u][a][0];
#pragma unroll
          for (int i = 0; i < S; ++i) unit_store(u + i, cx[i], expect);
          unit_store(u + S, creward, expect);
        }
      }
    }
    ++expect;
  }
}

// Registers: the server's wave must fit beside a wave of rollout_wide_served_kernel on one SIMD (512 registers): the walker's server is
// capped at 2 x 80 = 160 (beside 346 -> 352), the acrobot's at 2 x 72 = 144 (beside 336 / 363).  launch_env_server checks the sum at run
// time (hipFuncGetAttributes) and leaves the server out where it does not fit (the generic walker instantiation).
template <typename SPEC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(80))) void env_server_walker_kernel(DevParams P)
{
  env_server_wide_body<GRLX_ENV_COMPASS_WALKER, 3, SPEC, false>(P);
}
template <typename SPEC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(64))) void env_server_acrobot_kernel(DevParams P)
{
  env_server_wide_body<GRLX_ENV_ACROBOT, 3, SPEC, false>(P);
}
// the acrobot's server beside the SPECIALISED rollout kernel (336 registers): room for the sine's constants in registers (2 x 88 = 176)
template <typename SPEC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(88))) void env_server_acrobot_pinned_kernel(DevParams P)
{
  env_server_wide_body<GRLX_ENV_ACROBOT, 3, SPEC, true>(P);
}

} // namespace grlx
