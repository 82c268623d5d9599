// grlx_env_server.h -- the environment step of the 4-replicas-per-wave pendulum kernels, moved to a SECOND kernel that shares the SIMDs.
// Part of the single translation unit grlx_kernels.hip (included there, in order; not self-contained).
//
// Why.  At 4096 replicas the rollout kernel runs exactly one wave per SIMD and issues on 0.64 of its cycles; the rest is latency of its
// own dependent instructions (f64 chains of the sine, LDS round trips), which only another wave on the SIMD could fill -- and a second
// 384-register wave does not fit.  But the RK4 integration (40 % of a pass) needs few registers, reads no table, and its inputs are known
// early: the next step is taken from the state this pass started in, with one of NA known torques.  So a small kernel -- one 64-thread
// block per rollout wave, 96 registers beside the 416 of rollout_served_kernel, resident on the same SIMDs (tools/microbench/coresident.hip:
// both kernels stay resident, a round trip between them through device memory costs 3.8 k cycles) -- integrates the steps AHEAD of the
// rollout wave, for every action it can choose (two levels: see env_server_kernel), while the rollout wave works through its table
// phase; the rollout wave then only fetches the result of the action its sampler chose.  Same operations on the same arguments (env_step
// is the same code): same bits.  DESIGN.md section 4.1g has the measurements.
//
// Protocol, one 1-KB mailbox per replica (EnvMail).  Every payload word travels in a 16-byte UNIT {value, seq} written by one lane with one
// 16-byte store and read with one 16-byte load (device scope, sc1: served by memory that every XCD sees): a unit is either the old or the
// new one, and says which -- no ordering between accesses is needed, a reader that finds a wrong seq reads again.
//   rollout wave -> server: cmd[seq & 3] = seq << 8 | op, seq = 1, 2, ...;  op 0..NA-1: "the step was taken with action op",
//                           op kReset: "a trial starts in reset[]" (units tagged seq), op kExit: "no further command".
//   server -> rollout wave: cand[seq & 1][a] = units {x0, x1, x2, obs0, reward} after command seq, stepped with action a (obs1 = x1;
//                           terminal and the domain check are comparisons on x the rollout wave redoes).  Two buffers: the server writes
//                           the candidates of command seq + 1 while the rollout wave may still be reading those of command seq.
// The rollout wave never depends on the server: a fetch that does not arrive within kFetchPolls polls is computed locally (it has the state
// and the action), the replica tells the server to stop and integrates by itself for the rest of the launch.  The server leaves when every
// replica of its block has sent kExit (or has gone on without it), or after kServerStartPolls polls without a first command /
// kServerIdlePolls polls (or iterations that consume no command) without a further one.  Neither side can hang the other.
#pragma once

namespace grlx {

struct __attribute__((aligned(16))) MailUnit { double v; unsigned long long seq; };
constexpr int kCandUnits = 5;
struct __attribute__((aligned(1024))) EnvMail {
  unsigned long long cmd[4];                    //   0
  MailUnit reset[3];                            //  32
  unsigned long long pad0[6];                   //  80
  unsigned long long stats[16];                 // 128  ([15]: served to the end 1 / fell back 2; the rest: GRLX_ENV_SERVER_STATS builds)
  MailUnit cand[2][3][kCandUnits];              // 256 .. 736
  unsigned long long pad1[36];
};
static_assert(sizeof(EnvMail) == kEnvMailBytes, "one mailbox = kEnvMailBytes");
static_assert(offsetof(EnvMail, stats) + 15 * sizeof(unsigned long long) == kEnvMailFlagOffset, "the flag grlx_env_server_counts reads");

typedef __attribute__((address_space(1))) unsigned long long env_gu64;
typedef unsigned int mail_u32x4 __attribute__((ext_vector_type(4)));
#define GRLX_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr unsigned kMailReset = 3u, kMailExit = 4u;
constexpr unsigned kFetchPolls = 400u;              // x (one round trip through memory) = ~0.5 ms: the server is not there
constexpr unsigned kFirstFetchPolls = 8000u;        // the first answer of a launch: the two kernels start on two streams, and the first
                                                    //   launch of a kernel in a process may wait for its scratch to be set up
constexpr unsigned kServerStartPolls = 20000u;      // x (s_sleep + one round trip) = tens of ms without a first command: the rollout kernel is
                                                    //   not running beside this one (a profiler that serialises kernels, a busy device)
constexpr unsigned kServerIdlePolls = 400000u;      // ... = half a second without a further command

__device__ __forceinline__ void mail_store(void *p, unsigned long long v) { __hip_atomic_store((env_gu64 *)p, v, GRLX_RLX_AGENT); }
__device__ __forceinline__ unsigned long long mail_load(const void *p) { return __hip_atomic_load((env_gu64 *)p, GRLX_RLX_AGENT); }
__device__ __forceinline__ void mail_setprio(unsigned p)
{
  if (p == 1) __builtin_amdgcn_s_setprio(1);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else if (p == 3) __builtin_amdgcn_s_setprio(3);
}
__device__ __forceinline__ unsigned long long mail_clock()
{
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

// one unit, one access
__device__ __forceinline__ void unit_store(MailUnit *p, double v, unsigned long long seq)
{
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  mail_u32x4 d;
  d.x = (unsigned)b; d.y = (unsigned)(b >> 32); d.z = (unsigned)seq; d.w = (unsigned)(seq >> 32);
  // (s_nop: a store of more than 8 bytes reads its data registers a cycle late, and the assembler does not see into this string)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
}
// server side: the three units of a start state, all loads in flight together, then ONE wait; true when every unit carries `seq`
// (inline assembly, waited for on the spot: nothing else is in flight in the server when it reads a start state)
template <int N>
__device__ __forceinline__ bool units_load(const MailUnit *p, unsigned long long seq, double *v)
{
  static_assert(N == 3, "");
  mail_u32x4 d[N];
  asm volatile("global_load_dwordx4 %0, %3, off sc1\n\t"
               "global_load_dwordx4 %1, %3, off offset:16 sc1\n\t"
               "global_load_dwordx4 %2, %3, off offset:32 sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2])
               : "v"(p)
               : "memory");
  bool ok = true;
#pragma unroll
  for (int i = 0; i < N; ++i)
  {
    ok = ok && (((unsigned long long)d[i].w << 32) | d[i].z) == seq;
    v[i] = __longlong_as_double((long long)(((unsigned long long)d[i].y << 32) | d[i].x));
  }
  return ok;
}

// rollout side, ONE lane of the replica's group
__device__ __forceinline__ void mail_send(EnvMail *m, unsigned long long seq, unsigned op) { mail_store(&m->cmd[seq & 3u], (seq << 8) | op); }
__device__ __forceinline__ void mail_send_reset(EnvMail *m, unsigned long long seq, const double *x)
{
  unit_store(&m->reset[0], x[0], seq);
  unit_store(&m->reset[1], x[1], seq);
  unit_store(&m->reset[2], x[2], seq);
  mail_send(m, seq, kMailReset);
}

// rollout side, all 16 lanes of the replica's group: what command `seq` followed by action `a` leaves.  Lane j < 5 loads unit j (4 registers
// instead of 20), the group agrees on whether all five were the ones of `seq`, and the values are handed round.  The load is a buffer
// load the compiler schedules and waits for itself, so that it can be issued early (mail_prefetch, right after the action is chosen: one
// trip through memory, 2.3 k cycles, hidden behind the rest of the pass) and looked at late (mail_take, when the next pass starts).
struct MailBox {
  __amdgpu_buffer_rsrc_t rs;      // all mailboxes of the launch
  unsigned at;                    // this lane's byte offset: its replica's mailbox + the unit it loads
};
__device__ __forceinline__ MailBox mailbox_of(const DevParams &P, int r, int j)
{
  MailBox b;
  b.rs = __builtin_amdgcn_make_buffer_rsrc((void *)P.env_mail, 0, (int)((size_t)P.n_replicas * kEnvMailBytes), 0x00020000);
  b.at = (unsigned)r * (unsigned)kEnvMailBytes + (unsigned)offsetof(EnvMail, cand) + (unsigned)(j < kCandUnits ? j : 0) * (unsigned)sizeof(MailUnit);
  return b;
}
__device__ __forceinline__ mail_u32x4 mail_prefetch(const MailBox &b, unsigned long long seq, int a)
{
  const unsigned off = b.at + (((unsigned)seq & 1u) * 3u + (unsigned)a) * (unsigned)(kCandUnits * sizeof(MailUnit));
  return __builtin_amdgcn_raw_buffer_load_b128(b.rs, off, 0, 16);         // 16 = sc1: device scope
}
template <int ENV>
__device__ __forceinline__ bool mail_take(const DevParams &N, const MailBox &b, mail_u32x4 d, unsigned long long seq, int a, int g,
                                          unsigned long long gmask, double *x, double *obs, double &reward, int &terminal, uint32_t &status,
                                          unsigned long long *polled = nullptr)
{
  static_assert(ENV == GRLX_ENV_PENDULUM, "the pendulum's mailbox");
  bool ok = false;
  unsigned polls = 0;
  for (;;)
  {
    const bool mine = (((unsigned long long)d.w << 32) | d.z) == seq;
    if ((__ballot(mine) & gmask) == gmask) { ok = true; break; }
    if (++polls > (seq <= 1u ? kFirstFetchPolls : kFetchPolls)) break;
    d = mail_prefetch(b, seq, a);
  }
  if (polled) *polled += polls;
  if (!ok) return false;
  const double v = __longlong_as_double((long long)(((unsigned long long)d.y << 32) | d.x));
  const int base = g * 16;
  x[0] = lane_fetch(v, base + 0);
  x[1] = lane_fetch(v, base + 1);
  x[2] = lane_fetch(v, base + 2);
  obs[0] = lane_fetch(v, base + 3);
  obs[1] = x[1];
  reward = lane_fetch(v, base + 4);
  terminal = x[2] > N.timeout ? 1 : 0;                            // Env::observe
  if (!Env<ENV>::in_domain(x)) status |= ST_DOMAIN;             // env_step
  return true;
}

// The server: block b serves the replicas 4b .. 4b+3 of rollout wave b.  It looks TWO steps ahead: lane 16 q + 3 a + a2 (a, a2 < NA) of
// replica q holds the state after the actions (a, a2) applied to the replica's last confirmed state -- nine per replica, one instruction
// stream.  When the command "the step was taken with action a*" arrives, the three lanes (a*, .) hold what the rollout wave asks for next and
// store it at once; only then do all nine integrate the level after it, each from the state lane (a*, its a) held.  The integration is off
// the path between a command and its answer (two trips through memory), at the same number of instructions per step.
// Registers: 48 (x 2: vector + accumulation) = 96 of a SIMD's 512, beside the 416 of rollout_served_kernel.  The constants of the dynamics
// stay literals here (PIN = false).
template <int ENV, int NA, typename SPEC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(48))) void env_server_kernel(DevParams P)
{
  static_assert(NA == 3, "sixteen lanes per replica: nine pairs of actions");
  constexpr int S = Env<ENV>::S, D = Env<ENV>::D;
  static_assert(S == 3 && D == 2, "the pendulum's mailbox");
  const DevParams &N = SPEC::numeric(P);
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, c = lane & 15;
  const int a = c / NA, a2 = c % NA;                     // (c >= 9: idle lanes, kept in step for the shuffles)
  const bool worker = c < NA * NA;
  const int r = blockIdx.x * kReplicasPerWave + q;
  bool done = r >= P.n_replicas || (P.env_tune & 64u) != 0;   // (64: tests -- a server that is not there)
  EnvMail *m = P.env_mail + (done ? 0 : r);
  const double act = N.actions[a2];
  unsigned long long expect = 1;
  // this lane's candidate: state, observed angle and reward after (a, a2)
  double cx[S], cobs0 = 0, creward = 0;
#pragma unroll
  for (int i = 0; i < S; ++i) cx[i] = 0;
  unsigned idle = 0;
  bool seen = false;
  mail_setprio((P.env_tune >> 2) & 3u);
#ifdef GRLX_ENV_SERVER_STATS
  unsigned long long t_begin = mail_clock(), t_busy = 0, n_cmd = 0, n_idle = 0, n_split = 0, n_batch = 0;
#endif
  // (tests: bits 8-23 of env_tune = the server leaves, unannounced, once it has answered that many commands of a replica -- a fetch in
  //  MID-episode then runs into its bound and the rollout wave goes on by itself from there)
  const unsigned long long quit_after = (P.env_tune >> 8) & 0xFFFFu;
  for (;;)
  {
    if (quit_after != 0u && expect > quit_after) done = true;
    if (__all(done)) break;
    unsigned long long word = 0;
    bool ready = false;
    if (!done)
    {
      word = mail_load(&m->cmd[expect & 3u]);
      ready = (word >> 8) == expect;
      if ((word >> 8) > expect) done = true;          // the replica went on without the server (its later commands overwrote this one)
    }
    if (!__any(ready))
    {
      __builtin_amdgcn_s_sleep(1);
#ifdef GRLX_ENV_SERVER_STATS
      ++n_idle;
#endif
      if (++idle > (seen ? kServerIdlePolls : kServerStartPolls)) break;
      continue;
    }
    seen = true;
    // (`idle` is reset only by an iteration that CONSUMED a command: a reset command whose start-state units never arrive -- it cannot
    //  happen while a reset is always followed by four more commands of the launch, but nothing here relies on that -- keeps `ready`
    //  true without progress, and still runs into kServerIdlePolls)
    bool progress = false;
#ifdef GRLX_ENV_SERVER_STATS
    const unsigned long long t0 = mail_clock();
    if (!__all(ready || done)) ++n_split;
    ++n_batch;
#endif
    if (ready)
    {
      unsigned op = (unsigned)(word & 0xFFu);
      if (op == kMailExit)
      {
        done = true;
        progress = true;
      }
      else
      {
        bool have = true;
        if (op == kMailReset)
        { // a trial starts: the first level is integrated here, from the start state (every lane (., a2) the same f(start, a2));
          // the lanes (0, .) then stand for "the action taken".  (A second copy of the integration: folding both levels into one loop
          // costs registers the kernel does not have -- 102 and scratch, and with scratch the two kernels no longer share the SIMDs.)
          double x[S];
          have = units_load<3>(&m->reset[0], expect, x);      // (not there yet: the command is looked at again)
          if (have)
          {
            double obs[D];
            int terminal = 0;
            uint32_t st = 0;
#pragma unroll
            for (int i = 0; i < S; ++i) cx[i] = x[i];
            env_step<ENV, false>(N, cx, act, obs, creward, terminal, st);
            cobs0 = obs[0];
            op = 0u;
          }
        }
        if (have)
        { // the answer: what the action taken leaves, for every next action
          if (worker && a == (int)op)
          {
            MailUnit *u = &m->cand[expect & 1u][a2][0];
            unit_store(u + 0, cx[0], expect);
            unit_store(u + 1, cx[1], expect);
            unit_store(u + 2, cx[2], expect);
            unit_store(u + 3, cobs0, expect);
            unit_store(u + 4, creward, expect);
          }
          // the level after it: lane (a, a2) continues from the state the lane (action taken, a) holds
          const int src = (lane & ~15) + (int)op * NA + (worker ? a : 0);
          double x[S];
#pragma unroll
          for (int i = 0; i < S; ++i) x[i] = lane_fetch(cx[i], src);
          double obs[D];
          int terminal = 0;
          uint32_t st = 0;
#pragma unroll
          for (int i = 0; i < S; ++i) cx[i] = x[i];
          env_step<ENV, false>(N, cx, act, obs, creward, terminal, st);
          cobs0 = obs[0];
          ++expect;
          progress = true;
#ifdef GRLX_ENV_SERVER_STATS
          ++n_cmd;
#endif
        }
      }
    }
    if (__any(progress)) idle = 0;
    else if (++idle > kServerIdlePolls) break;
#ifdef GRLX_ENV_SERVER_STATS
    t_busy += mail_clock() - t0;
#endif
  }
#ifdef GRLX_ENV_SERVER_STATS
  if (c == 0 && r < P.n_replicas)
  {
    EnvMail *mm = P.env_mail + r;
    mm->stats[0] = mail_clock() - t_begin;
    mm->stats[1] = t_busy;
    mm->stats[2] = n_cmd;
    mm->stats[3] = n_idle;
    mm->stats[9] = n_split;
    mm->stats[10] = n_batch;
  }
#endif
}

} // namespace grlx
