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
//    <= 10 trace entries, the table position and the current weight of tiling j
//    (a write-through cache, so trace updates need no loads).
//  * weights live in a per-replica open-addressing table in HBM/L2 (16-byte
//    {slot, weight} entries), lazily initialised to the value the reference's
//    8,388,608-draw initialisation gives that slot (LCG jump-ahead).
//  * sums over the 16 tilings are taken in the reference's serial order
//    (linear.cpp:147-151) through a 4-way interleaved LDS tile, so Q-values are
//    bit-identical to a scalar run.
//
// Compiled with -ffp-contract=off: the reference's arithmetic is plain IEEE
// double without fused multiply-add (x86-64 baseline); only grlx_math.h fuses.
#include "grlx_internal.h"
#include "grlx_math.h"

namespace grlx {

// ------------------------------------------------------------------ RNG ----
// drand48 family (utils.h:84-137): X' = (A*X + C) mod 2^48.
constexpr uint64_t kLcgA = 0x5DEECE66DULL, kLcgC = 0xBULL, kMask48 = (1ULL << 48) - 1;

// Jump-ahead x -> A^n x + C_n by byte windows of n: entry [w][b] is the affine map of
// b * 256^w draws, so a jump of up to 2^32 draws costs four multiply-adds.
struct JumpTable { uint64_t a[4][256], c[4][256]; };
constexpr JumpTable make_jump_table()
{
  JumpTable t{};
  uint64_t sa = kLcgA, sc = kLcgC;                 // map of 256^w draws
  for (int w = 0; w < 4; ++w)
  {
    uint64_t a = 1, c = 0;                         // identity = 0 draws
    for (int b = 0; b < 256; ++b)
    {
      t.a[w][b] = a;
      t.c[w][b] = c;
      c = (sa * c + sc) & kMask48;                 // compose with one more window step
      a = (sa * a) & kMask48;
    }
    sa = a;                                        // after 256 steps: map of 256^(w+1) draws
    sc = c;
  }
  return t;
}
__device__ const JumpTable d_jump = make_jump_table();

__device__ __forceinline__ uint64_t lcg_next(uint64_t x) { return (kLcgA * x + kLcgC) & kMask48; }
__device__ __forceinline__ double   lcg_double(uint64_t x) { return (double)x * 0x1p-48; }
__device__ __forceinline__ uint32_t lcg_long(uint64_t x) { return (uint32_t)(x >> 17); }

__device__ inline uint64_t lcg_step_pow2(uint64_t x, uint64_t n)
{ // generic O(log n) jump by repeated squaring (only for n >= 2^32)
  uint64_t a = kLcgA, c = kLcgC;
  while (n)
  {
    if (n & 1) x = (a * x + c) & kMask48;
    c = ((a + 1) * c) & kMask48;
    a = (a * a) & kMask48;
    n >>= 1;
  }
  return x;
}

__device__ inline uint64_t lcg_jump(uint64_t x, uint64_t n)
{
#pragma unroll
  for (int w = 0; w < 4; ++w)
  {
    const uint32_t b = (uint32_t)(n >> (8 * w)) & 0xFFu;
    x = (d_jump.a[w][b] * x + d_jump.c[w][b]) & kMask48;
  }
  if (n >> 32) x = lcg_step_pow2(x, (n >> 32) << 32);
  return x;
}

// value the reference's dense initialisation gives `slot` (linear.cpp:117-120:
// params_[ii] = rand->getUniform(init_min, init_max) in index order)
__device__ inline double lazy_weight(uint64_t tl0, const LinearParams &lp, uint32_t slot)
{
  uint64_t x = lcg_next(lcg_jump(tl0, lp.draws_before + (uint64_t)slot));
  return lp.init_min + lcg_double(x) * lp.init_range;
}

// ----------------------------------------------------------- tile coding ---
__device__ __forceinline__ int smod(int x, int y)
{ // utils.h:70-78
  int r = x % y;
  return r < 0 ? r + y : r;
}

__device__ __forceinline__ uint32_t murmur_mix(uint32_t h, int c)
{ // tile_coding.h:96-107
  const uint32_t m = 0x5bd1e995u;
  uint32_t k = (uint32_t)c;
  k *= m;
  k ^= k >> 24;
  k *= m;
  h *= m;
  h ^= k;
  return h;
}

__device__ __forceinline__ uint32_t murmur_final(uint32_t h)
{ // tile_coding.h:109-113
  const uint32_t m = 0x5bd1e995u;
  h ^= h >> 13;
  h *= m;
  h ^= h >> 15;
  return h;
}

// coordinate of dimension i in tiling j (tile_coding.cpp:128-141)
template <int T>
__device__ __forceinline__ int tile_coord(const TileParams &tp, int i, int q, int j)
{
  int c = q - smod(q - j * (1 + 2 * i), T);
  if (tp.wrap[i] != 0)
    c = smod(c, tp.wrap[i]);
  return c;
}

__device__ __forceinline__ int tile_quant(const TileParams &tp, int i, double x)
{ // tile_coding.cpp:121-125
  return (int)__builtin_floor(x * tp.scaling[i]);
}

// generic (runtime T) projection of one input for tiling j
__device__ inline uint32_t tile_slot_generic(const TileParams &tp, const double *in, int j)
{
  uint32_t h = 449u ^ (uint32_t)(tp.D + 1);
  for (int i = 0; i < tp.D; ++i)
  {
    int q = tile_quant(tp, i, in[i]);
    int c = q - smod(q - j * (1 + 2 * i), tp.T);
    if (tp.wrap[i] != 0)
      c = smod(c, tp.wrap[i]);
    h = murmur_mix(h, c);
  }
  h = murmur_mix(h, j);
  return murmur_final(h) % (uint32_t)tp.memory;
}

// ---------------------------------------------------------- sparse table ---
struct Table {
  Entry   *base;
  uint32_t mask, shift;
};

__device__ __forceinline__ Table table_of(const DevParams &P, int table, int replica)
{
  Table t;
  t.base = P.tables + (((size_t)table * (size_t)P.n_replicas + (size_t)replica) << P.logC);
  t.mask = (1u << P.logC) - 1u;
  t.shift = 32u - P.logC;
  return t;
}

__device__ __forceinline__ uint32_t table_home(const Table &t, uint32_t slot)
{
  return ((slot + 1u) * 0x9E3779B1u) >> t.shift;
}

__device__ __forceinline__ uint4 entry_load(const Table &t, uint32_t pos)
{
  return *reinterpret_cast<const uint4 *>(&t.base[pos]);
}

__device__ __forceinline__ double entry_val(const uint4 &raw)
{
  return __longlong_as_double((long long)(((unsigned long long)raw.w << 32) | raw.z));
}

__device__ __forceinline__ void entry_store(const Table &t, uint32_t pos, uint32_t key, double v)
{
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  uint4 raw;
  raw.x = key;
  raw.y = 0;
  raw.z = (uint32_t)b;
  raw.w = (uint32_t)(b >> 32);
  *reinterpret_cast<uint4 *>(&t.base[pos]) = raw;
}

__device__ __forceinline__ void value_store(const Table &t, uint32_t pos, double v) { t.base[pos].val = v; }
__device__ __forceinline__ double value_load(const Table &t, uint32_t pos) { return t.base[pos].val; }

// order LDS / global accesses of the lanes of one wave (no instruction beyond waits)
__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Probe sequence.  Entries are grouped in 128-byte buckets of 8 (one cache line): the
// t-th probe of a slot stays inside its home bucket for t < 8 (so every probe after
// the first is an L1 hit) and moves to the following buckets only when a bucket is full.
__device__ __forceinline__ uint32_t probe_pos(const Table &t, uint32_t home, uint32_t n)
{
  uint32_t bucket = ((home >> 3) + (n >> 3)) & (t.mask >> 3);
  return (bucket << 3) | ((home + n) & 7u);
}

// Resolve NP independent lookups of one lane at once: the first-round loads of all of
// them are in flight together (one memory round trip per step in the common case).
// found[i]: position of the entry (hit) or of the first empty position (miss).
template <int NP>
__device__ __forceinline__ void table_lookup(const Table &t, const uint32_t (&slot)[NP], uint32_t (&home)[NP], uint32_t (&tries)[NP],
                                             double (&val)[NP], bool (&miss)[NP], uint32_t &status)
{
  uint4 raw[NP];
  bool pending[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i)
  {
    home[i] = table_home(t, slot[i]);
    tries[i] = 0;
    raw[i] = entry_load(t, probe_pos(t, home[i], 0));
    pending[i] = true;
    miss[i] = false;
  }
  for (int it = 0; it < kMaxProbe; ++it)
  {
    bool any = false;
#pragma unroll
    for (int i = 0; i < NP; ++i)
      if (pending[i])
      {
        if (raw[i].x == slot[i] + 1u) { val[i] = entry_val(raw[i]); pending[i] = false; }
        else if (raw[i].x == 0u) { miss[i] = true; pending[i] = false; }
        else
        {
          tries[i]++;
          raw[i] = entry_load(t, probe_pos(t, home[i], tries[i]));
          any = true;
        }
      }
    if (!any) break;
  }
#pragma unroll
  for (int i = 0; i < NP; ++i)
    if (pending[i]) status |= ST_TABLE_FULL;
}

// Create the slots that were missing, with their lazy initial weights.  Lanes of one
// replica may miss on the same empty position, so inserts are serialised within each
// 16-lane group (one lane per group at a time); they are rare (about 17 k per replica
// over a whole pendulum run).
__device__ inline void table_insert(const Table &t, const LinearParams &lp, uint64_t tl0, bool miss,
                                    uint32_t slot, uint32_t home, uint32_t &tries, double &val, uint32_t &status, uint32_t &inserted)
{
  const int lane = threadIdx.x & 63;
  unsigned long long pend = __ballot(miss);
  if (pend == 0ull) return;
  double w0 = 0.0;
  if (miss) w0 = lazy_weight(tl0, lp, slot);
  while (pend != 0ull)
  {
    unsigned long long sel = 0ull;                 // lowest pending lane of every 16-lane group goes now
#pragma unroll
    for (int gg = 0; gg < 4; ++gg)
    {
      unsigned long long grp = pend & (0xFFFFull << (16 * gg));
      sel |= grp & (~grp + 1ull);
    }
    if ((sel >> lane) & 1ull)
    {
      bool done = false;
      for (int it = 0; it < kMaxProbe; ++it)
      {
        const uint32_t p = probe_pos(t, home, tries);
        uint4 raw = entry_load(t, p);
        if (raw.x == slot + 1u) { val = entry_val(raw); done = true; break; }   // a sibling lane created it
        if (raw.x == 0u)
        {
          entry_store(t, p, slot + 1u, w0);
          val = w0;
          inserted++;
          done = true;
          break;
        }
        tries++;
      }
      if (!done) status |= ST_TABLE_FULL;
    }
    pend &= ~sel;
    // The next lane's probe must observe this insert.  Both are vector memory operations
    // of the same wave issued in this order, which the hardware keeps for one address;
    // the fence only stops the compiler from reordering them.
    wave_sync();
  }
}

// single lookup-or-create (fine-grained operators)
__device__ inline void table_probe(const Table &t, const LinearParams &lp, uint64_t tl0, bool active,
                                   uint32_t slot, uint32_t &pos, double &val, uint32_t &status, uint32_t &inserted)
{
  uint32_t sl[1] = {slot}, home[1] = {0}, tries[1] = {0};
  double v[1] = {0};
  bool miss[1] = {false};
  if (active) table_lookup<1>(t, sl, home, tries, v, miss, status);
  table_insert(t, lp, tl0, active && miss[0], slot, home[0], tries[0], v[0], status, inserted);
  pos = probe_pos(t, home[0], tries[0]);
  val = v[0];
}

// ----------------------------------------------------------- environments --
template <int ENV> struct Env;

// dynamics/pendulum + task/pendulum/swingup (pendulum.cpp:40-145)
template <> struct Env<GRLX_ENV_PENDULUM> {
  static constexpr int S = 3, D = 2;
  __device__ static __forceinline__ void eom(const double *x, double u, double *xd)
  { // pendulum.cpp:40-49, 55-68
    const double J = 0.000191, m = 0.055, g = 9.81, l = 0.042, b = 0.000003, K = 0.0536, R = 9.5;
    double a = x[0], ad = x[1];
    double add = (1 / J) * (m * g * l * psin(a) - b * ad - (K * K / R) * ad + (K / R) * u);
    xd[0] = ad;
    xd[1] = add;
    xd[2] = 1;
  }
  __device__ static __forceinline__ void start(const DevParams &P, int test, uint64_t &TL, uint64_t &, double *x)
  { // pendulum.cpp:97-103 (the RandGen draw happens every episode)
    TL = lcg_next(TL);
    double r = lcg_double(TL);
    x[0] = GRLX_PI + P.randomization * (test == 0) * r * 2 * GRLX_PI;
    x[1] = 0;
    x[2] = 0;
  }
  __device__ static __forceinline__ double actuate(double a) { return fmin(fmax(a, -3.0), 3.0); }   // :105-109
  __device__ static __forceinline__ int observe(const DevParams &P, const double *x, double *obs)
  { // :111-129
    double a = pfmod(x[0] + GRLX_PI, GRLX_2PI);
    if (a < 0) a += GRLX_2PI;
    obs[0] = a;
    obs[1] = x[1];
    return x[2] > P.timeout ? 1 : 0;
  }
  __device__ static __forceinline__ double evaluate(const DevParams &, const double *x, double action, const double *next)
  { // :131-145; pow(v, 2) is v*v in the portable specification
    double a = pfmod(__builtin_fabs(next[0]), GRLX_2PI);
    if (a > GRLX_PI) a -= GRLX_2PI;
    double reward = -5 * (a * a) - 0.1 * (next[1] * next[1]) - 1 * (action * action);
    if ((next[2] - x[2]) != 1)
      reward *= (next[2] - x[2]) / 0.03;
    return reward;
  }
};

// DynamicalModel::step (modeled.cpp:254-276): classical RK4 sub-steps
template <int ENV>
__device__ __forceinline__ void rk4_step(const DevParams &P, const double *x, double u, double *next)
{
  constexpr int S = Env<ENV>::S;
  const double h = P.h;
  double xd[S], k1[S], k2[S], k3[S], k4[S], t[S];
#pragma unroll
  for (int i = 0; i < S; ++i) next[i] = x[i];
  for (int ii = 0; ii < P.integration_steps; ++ii)
  {
    Env<ENV>::eom(next, u, xd);
#pragma unroll
    for (int i = 0; i < S; ++i) { k1[i] = h * xd[i]; t[i] = next[i] + k1[i] / 2; }
    Env<ENV>::eom(t, u, xd);
#pragma unroll
    for (int i = 0; i < S; ++i) { k2[i] = h * xd[i]; t[i] = next[i] + k2[i] / 2; }
    Env<ENV>::eom(t, u, xd);
#pragma unroll
    for (int i = 0; i < S; ++i) { k3[i] = h * xd[i]; t[i] = next[i] + k3[i]; }
    Env<ENV>::eom(t, u, xd);
#pragma unroll
    for (int i = 0; i < S; ++i)
    {
      k4[i] = h * xd[i];
      next[i] = next[i] + (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) / 6;
    }
  }
}

// ModeledEnvironment::step (modeled.cpp:160-213), window 1, no delta, discrete_time 1
template <int ENV>
__device__ __forceinline__ void env_step(const DevParams &P, double *x, double action, double *obs, double &reward, int &terminal)
{
  constexpr int S = Env<ENV>::S;
  double next[S];
  rk4_step<ENV>(P, x, Env<ENV>::actuate(action), next);
  terminal = Env<ENV>::observe(P, next, obs);
  reward = Env<ENV>::evaluate(P, x, action, next);
#pragma unroll
  for (int i = 0; i < S; ++i) x[i] = next[i];
}

// ------------------------------------------------------------ samplers -----
// GreedySampler::findmax (greedy.cpp:47-61); loops are unrolled over the
// compile-time action count so Q-values stay in registers
template <int NA>
__device__ __forceinline__ void findmax(const double (&v)[NA], int &mai, int &man, double &best)
{
  best = v[0];
  mai = 0;
  man = 1;
#pragma unroll
  for (int i = 1; i < NA; ++i)
  {
    if (v[i] > best) { best = v[i]; mai = i; man = 1; }
    else if (v[i] == best) man++;
  }
}

// random tie break (greedy.cpp:77-85): the (jj+1)-th maximal entry, jj = lrand48() % man;
// getInteger draws from the GLOBAL stream (utils.h:127-130)
template <int NA>
__device__ __forceinline__ int tie_break(const double (&v)[NA], double best, int man, uint64_t &G)
{
  G = lcg_next(G);
  int jj = (int)(lcg_long(G) % (uint32_t)man);
  int res = 0;
#pragma unroll
  for (int i = 0; i < NA; ++i)
    if (v[i] == best)
    {
      if (jj == 0) res = i;
      --jj;
    }
  return res;
}

template <typename Tv, int NA>
__device__ __forceinline__ Tv pick(const Tv (&arr)[NA], int idx)
{
  Tv v = arr[0];
#pragma unroll
  for (int a = 1; a < NA; ++a) v = (a == idx) ? arr[a] : v;
  return v;
}

__device__ __forceinline__ double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// ------------------------------------------------------- fused rollout -----
// LDS tile shared by the 4 replicas of the wave; index (row*16 + tiling)*4 + group
// makes both the per-lane writes and the 4 simultaneous broadcast reads conflict-free.
#define SHW(row, k, g) sh_w[(((row) * 16 + (k)) << 2) + (g)]

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

template <int ENV, int NA, bool DIAG>
__global__ __launch_bounds__(64) void rollout_kernel(DevParams P, int n_trials)
{
  constexpr int S = Env<ENV>::S, D = Env<ENV>::D, T = kLanesPerReplica;
  __shared__ double   sh_w[(NA + 1) * 16 * 4];
  __shared__ uint32_t sh_ppos[4 * 16];
  __shared__ double   sh_fb[16 * 4];
  __shared__ uint32_t sh_fbflag[16 * 4];

  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, j = lane & 15;
  const int r_raw = blockIdx.x * kReplicasPerWave + g;
  const bool live = r_raw < P.n_replicas;
  const int r = live ? r_raw : 0;
  constexpr int A = NA;
  const bool tapped = live && (r == P.tap_replica);

  ReplicaState &RS = P.states[r];
  double x[S];
#pragma unroll
  for (int i = 0; i < S; ++i) x[i] = RS.x[i];
  uint64_t G = RS.G, TL = RS.TL, S1 = RS.S1;
  const uint64_t TL0 = RS.TL0;
  double eps_decay = RS.eps_decay;
  int64_t tt = RS.tt, ss = RS.ss;
  uint64_t test_steps = RS.test_steps;
  uint32_t status = RS.status, rows = RS.rows, inserted = 0;

  const Table tab = table_of(P, 0, r);
  const double out_min = P.lin.out_min, out_max = P.lin.out_max;
  const double ee = P.gl;                       // pow(gamma*lambda, tau), tau = 1 (discrete_time)
  const double cut = (P.trace_kind == GRLX_TRACE_REPLACING) ? 0.01 : 0.0001;

  // register-resident replacing trace (trace.h:208-235), newest first
  uint32_t tr_pos[kMaxTrace];
  double   tr_val[kMaxTrace];
  uint32_t tr_cnt[kMaxTrace];
#pragma unroll
  for (int e = 0; e < kMaxTrace; ++e) { tr_pos[e] = kInvalidPos; tr_val[e] = 0; tr_cnt[e] = 0; }
  int    tr_len = 0;
  double tr_total = 1.;
  unsigned long long diag_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, diag_last = 0;
  if (DIAG) diag_last = stamp();

  for (int trial = 0; trial < n_trials; ++trial, ++tt)
  {
    const int ti = P.test_interval;
    const int test = (ti >= 0 && tt % (ti + 1) == ti) ? 1 : 0;        // online_learning.cpp:160
    double obs[D], reward = 0, total_reward = 0;
    int terminal = 0;
    bool running = live;

    // environment_->start (modeled.cpp:132-158)
    if (live)
    {
      Env<ENV>::start(P, test, TL, G, x);
      Env<ENV>::observe(P, x, obs);
    }
    // agent->start: TDAgent::start clears the trace (td.cpp:50-61, sarsa.cpp:126-132)
    if (!test)
    {
#pragma unroll
      for (int e = 0; e < kMaxTrace; ++e) tr_pos[e] = kInvalidPos;
      tr_len = 0;
      tr_total = 1.;
    }
    double time = 0;
    double action = 0;
    int    action_index = 0;
    uint32_t p_pos = kInvalidPos, p_slot = 0;
    bool first = true;                    // first pass = start(): act only, no env step / update

    for (;;)
    {
      if (!__any(running)) break;
      if (running)
      {
        DIAG_STAMP(0)
        // -------- environment step (skipped on the start() pass)
        if (!first)
        {
          env_step<ENV>(P, x, action, obs, reward, terminal);             // online_learning.cpp:196
          total_reward += reward;                                          // :202
          time += 1;                                                       // tau = 1
        }
        const bool has_next = first || terminal != 2;
        DIAG_STAMP(1)

        // -------- policy: Q(s', .) for all actions (q.cpp:94-107)
        double q[NA];
        uint32_t slot[NA], pos[NA];
        double w[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) { q[a] = 0; slot[a] = 0; pos[a] = 0; w[a] = 0; }
        double wp = 0;
        const bool update = !first && !test;                               // a TD update follows
        if (has_next)
        {
          int qd[GRLX_MAX_DIMS];
          uint32_t hpre = 449u ^ (uint32_t)(D + 2);
#pragma unroll
          for (int i = 0; i < D; ++i)
          {
            qd[i] = tile_quant(P.tile, i, obs[i]);
            hpre = murmur_mix(hpre, tile_coord<T>(P.tile, i, qd[i], j));
          }
#pragma unroll
          for (int a = 0; a < NA; ++a)
          {
            int qa = tile_quant(P.tile, D, P.actions[a]);
            uint32_t h = murmur_mix(hpre, tile_coord<T>(P.tile, D, qa, j));
            h = murmur_mix(h, j);
            slot[a] = murmur_final(h) % (uint32_t)P.tile.memory;
          }
        }
        DIAG_STAMP(2)
        // previous step's stores precede these loads in program order; they were issued
        // a full RK4 ago, so this wait is free and makes the ordering explicit
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        DIAG_STAMP(6)
        if (update) wp = value_load(tab, p_pos);                           // current weights of project(s, a)
        if (has_next)
        {
          uint32_t home[NA], tries[NA];
          bool miss[NA];
          table_lookup<NA>(tab, slot, home, tries, w, miss, status);
          DIAG_STAMP(7)
#pragma unroll
          for (int a = 0; a < NA; ++a)
            table_insert(tab, P.lin, TL0, miss[a], slot[a], home[a], tries[a], w[a], status, inserted);
#pragma unroll
          for (int a = 0; a < NA; ++a)
          {
            pos[a] = probe_pos(tab, home[a], tries[a]);
            SHW(a, j, g) = w[a];
          }
        }
        DIAG_STAMP(3)
        if (update) SHW(NA, j, g) = wp;
        sh_ppos[g * 16 + j] = p_pos;
        sh_fbflag[j * 4 + g] = 0u;
        wave_sync();
        if (has_next)
        {
#pragma unroll
          for (int a = 0; a < NA; ++a)
            { // LinearRepresentation::read (linear.cpp:136-184): serial sum, mean, clamp
              double s = 0;
#pragma unroll
              for (int k = 0; k < 16; ++k) s += SHW(a, k, g);
              s /= 16;
              q[a] = clampd(s, out_min, out_max);
            }
        }
        double qsa = 0;
        if (update)
        {
          double s = 0;
#pragma unroll
          for (int k = 0; k < 16; ++k) s += SHW(NA, k, g);
          s /= 16;
          qsa = clampd(s, out_min, out_max);
        }

        // -------- sampler (greedy.cpp:63-86, 144-218)
        int a_next = 0;
        if (has_next)
        {
          int mai, man;
          double best;
          findmax<NA>(q, mai, man, best);
          if (test)
          {
            a_next = (man > 1) ? tie_break<NA>(q, best, man, G) : mai;
          }
          else
          {
            if (time == 0.) eps_decay = fmax(eps_decay * P.decay_rate, P.decay_min);
            S1 = lcg_next(S1);
            double rnd = lcg_double(S1);
            if (rnd < eps_decay * P.epsilon)
            {
              G = lcg_next(G);
              a_next = (int)(lcg_long(G) % (uint32_t)A);
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
          if (has_next)
          {
            if (P.agent == GRLX_AGENT_SARSA)
              target += P.gamma * pick<double, NA>(q, a_next);
            else
            {
              double v = -__builtin_inf();
#pragma unroll
              for (int kk = 0; kk < NA; ++kk) v = fmax(v, q[kk]);
              target += P.gamma * v;
            }
          }
          delta = target - qsa;
          const double dW = P.alpha * (target - qsa);          // LinearRepresentation::write (linear.cpp:186-196)
          const double dT = P.alpha * delta;                   // VectorConstructor(alpha_*delta)

          // positions of p for the 16 tilings of this replica
          uint32_t pp[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) pp[k] = sh_ppos[g * 16 + k];
          uint32_t cp = 0;                                     // duplicates of my slot inside p
#pragma unroll
          for (int k = 0; k < 16; ++k) cp += (pp[k] == p_pos) ? 1u : 0u;

          // trace entries, newest first (representation.h:79-83, trace.h:150-178)
          if (P.trace_kind == GRLX_TRACE_REPLACING)
          {
            double weight = 1.;
            bool upd = true;
#pragma unroll
            for (int e = 0; e < kMaxTrace; ++e)
            {
              if (e < tr_len)
              {
                upd = upd && (weight > 0.001);
                const double de = weight * dT * ee;
                if (tr_pos[e] != kInvalidPos)
                {
                  uint32_t m = 0;
#pragma unroll
                  for (int k = 0; k < 16; ++k) m |= (pp[k] == tr_pos[e]) ? (1u << k) : 0u;
                  if (m == 0u)
                  {
                    if (upd)
                    { // LinearRepresentation::update (linear.cpp:198-216), duplicates applied tr_cnt times
                      double v = tr_val[e];
                      for (uint32_t c = 0; c < tr_cnt[e]; ++c)
                        v = P.lin.limit ? clampd(v + de, out_min, out_max) : v + de;
                      tr_val[e] = v;
                      value_store(tab, tr_pos[e], v);
                    }
                  }
                  else
                  { // this slot is also written through p: p's write comes first, then this entry's update;
                    // the slot leaves the trace (IndexProjection::ssub, projection.h:94-104)
                    if (upd)
                    {
                      double v = tr_val[e];
                      const uint32_t cpx = (uint32_t)__builtin_popcount(m);
                      for (uint32_t c = 0; c < cpx; ++c)
                        v = P.lin.limit ? clampd(v + dW, out_min, out_max) : v + dW;
                      for (uint32_t c = 0; c < tr_cnt[e]; ++c)
                        v = P.lin.limit ? clampd(v + de, out_min, out_max) : v + de;
                      for (uint32_t mm = m; mm != 0u; mm &= mm - 1u)
                      {
                        int k = __builtin_ctz(mm);
                        sh_fb[k * 4 + g] = v;
                        sh_fbflag[k * 4 + g] = 1u;
                      }
                    }
                    tr_pos[e] = kInvalidPos;
                  }
                }
                weight *= ee;
              }
            }
          }
          wave_sync();
          // p's own write (linear.cpp:186-216): cp sequential additions of dW
          double v;
          if (sh_fbflag[j * 4 + g] != 0u)
            v = sh_fb[j * 4 + g];
          else
          {
            v = wp;
            for (uint32_t c = 0; c < cp; ++c)
              v = P.lin.limit ? clampd(v + dW, out_min, out_max) : v + dW;
          }
          value_store(tab, p_pos, v);

          // trace_->add(p, decay) (trace.h:215-234)
          if (P.trace_kind == GRLX_TRACE_REPLACING)
          {
            if (ee < cut)
            {
#pragma unroll
              for (int e = 0; e < kMaxTrace; ++e) tr_pos[e] = kInvalidPos;
              tr_len = 0;
              tr_total = 1.;
            }
            if (tr_len >= kMaxTrace) status |= ST_TRACE_OVERFLOW;
#pragma unroll
            for (int e = kMaxTrace - 1; e > 0; --e)
            {
              tr_pos[e] = tr_pos[e - 1];
              tr_val[e] = tr_val[e - 1];
              tr_cnt[e] = tr_cnt[e - 1];
            }
            tr_pos[0] = p_pos;
            tr_val[0] = v;
            tr_cnt[0] = cp;
            tr_len = (tr_len < kMaxTrace) ? tr_len + 1 : kMaxTrace;
            tr_total *= ee;
            while (tr_total < cut && tr_len > 1)
            {
              tr_total /= ee;
              tr_len--;
            }
#pragma unroll
            for (int e = 0; e < kMaxTrace; ++e)
              if (e >= tr_len) tr_pos[e] = kInvalidPos;
          }
        }

        DIAG_STAMP(5)
        // -------- tap (debug / parity tests)
        if (tapped && !first)
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
              tp->terminal = terminal;
              tp->trace_len = tr_len;
              for (int i = 0; i < GRLX_MAX_DIMS; ++i) tp->obs[i] = (i < D) ? obs[i] : 0.;
              tp->action = has_next ? P.actions[a_next] : action;
              tp->reward = reward;
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
          action = P.actions[a_next];                                      // discretizer_->at(index), uniform.cpp:140-151
          p_pos = pick<uint32_t, NA>(pos, a_next);
          p_slot = pick<uint32_t, NA>(slot, a_next);
        }
        if (!first && terminal) running = false;
        first = false;
      }
    }

    // row of a test trial (online_learning.cpp:238-262) -- or of every trial when test_interval < 0
    if (live && (ti >= 0 ? test : 1))
    {
      if (rows < (uint32_t)P.max_rows)
      {
        if (j == 0)
        {
          size_t at = (size_t)rows * (size_t)P.n_replicas + (size_t)r;
          P.row_reward[at] = total_reward;
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

hipError_t launch_rollout(const DevParams &P, int n_trials, hipStream_t stream)
{
  int waves = (P.n_replicas + kReplicasPerWave - 1) / kReplicasPerWave;
#define GRLX_LAUNCH(ENVID, NACT)                                                                              \
  if (P.env == ENVID && P.A == NACT)                                                                        \
  {                                                                                                         \
    if (P.diag_out)                                                                                         \
      hipLaunchKernelGGL((rollout_kernel<ENVID, NACT, true>), dim3(waves), dim3(64), 0, stream, P, n_trials); \
    else                                                                                                    \
      hipLaunchKernelGGL((rollout_kernel<ENVID, NACT, false>), dim3(waves), dim3(64), 0, stream, P, n_trials); \
    return hipGetLastError();                                                                               \
  }
  GRLX_LAUNCH(GRLX_ENV_PENDULUM, 3)
  GRLX_LAUNCH(GRLX_ENV_PENDULUM, 5)
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
  env_step<ENV>(P, x, action[i], o, rw, term);
  for (int k = 0; k < S; ++k) state[(size_t)i * S + k] = x[k];
  for (int k = 0; k < D; ++k) obs[(size_t)i * D + k] = o[k];
  reward[i] = rw;
  terminal[i] = term;
  bool bad = false;
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
    table_probe(tab, P.lin, P.states[r].TL0, valid, slot, pos, w, status, inserted);
    sh[lane] = w;
    shp[lane] = valid ? pos : kInvalidPos;
    wave_sync();
    double s = 0;
    for (int k = 0; k < Tn; ++k) s += sh[k];
    s /= Tn;
    s = clampd(s, P.lin.out_min, P.lin.out_max);
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
          v = P.lin.limit ? clampd(v + d, P.lin.out_min, P.lin.out_max) : v + d;
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
  const uint32_t home = table_home(tab, slot);
  double v = lazy_weight(P.states[replica].TL0, P.lin, slot);
  for (int it = 0; it < kMaxProbe; ++it)
  {
    uint4 raw = entry_load(tab, probe_pos(tab, home, (uint32_t)it));
    if (raw.x == slot + 1u) { v = entry_val(raw); break; }
    if (raw.x == 0u) break;
  }
  out[i] = v;
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
    case 0: r = psin(v); break;
    case 1: r = pcos(v); break;
    case 2: r = plog(v); break;
    case 3: r = pfmod(v, y[i]); break;
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
