// grlx_fqi.hip -- the batch path (BASELINE.json configs[4]): experiment/batch_learning + predictor/fqi +
// representation/iterative + representation/parameterized/ann over projector/pre/normalizing, R independent-seed
// replicas per GPU.  Reference: batch_learning.cpp:88-205, fqi.cpp:187-285, iterative.cpp:63-107, ann.cpp:133-263,
// normalizing.cpp:78-87, greedy.cpp:47-86.  The operation-by-operation specification (incl. the documented
// deviations D1-D4 from the unpinnable parts of the reference) is oracle/fqi.c; this file restates it for the GPU.
//
// Shape of the work.  A rebuild is `iterations` x (targets + `epochs` x (gradient over all samples + RPROP step)): a
// long chain of tiny dense passes (3 -> H -> 1 per sample, 2 x 101 parameters) whose only coupling is the gradient
// sum.  One lane = one sample; the network (101 doubles) sits in LDS; the per-sample gradient stays in registers
// only as long as it takes to park its factors in LDS, and is summed by the fixed two-level tree of the specification
// (chunks of 64 samples in sample order -> chunk sums strided over 64 lanes + wave shuffle), independent of the grid.  No contraction here is large enough for MFMA
// (K = 3 and K = 20, N = 1; a 16x16x4 f64 tile would be 80 % padding) and the chain is bound by launch latency and
// the reduction, not by FLOPs: see DESIGN.md section 4.3 for the measurement.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "grlx_internal.h"
#include "grlx_math.h"
#include "grlx_math_batch.h"
#include "grlx_rng.h"
#include "grlx_tile.h"
#include "grlx_table.h"
#include "grlx_envs.h"

namespace grlx {
void set_last_error(const char *msg);      // grlx_api.cpp

constexpr int kFqiNIn = 3;                 // task/pendulum/swingup: observation (2) ++ action (1)
constexpr int kFqiD = 2;
constexpr int kFqiS = 3;

struct FqiRep {                            // per-replica state between launches
  uint64_t G, TL;                          // global / thread-local streams (fqi.c: orc_fqi_create)
  int64_t  n;                              // transitions stored
  int64_t  batches_done;
  unsigned long long maxdelta_bits;        // max |target - previous target| of the running iteration (non-negative double)
  double   maxdelta_last;
  int32_t  done;                           // the rebuild's iteration loop has ended (maxdelta <= 0.001)
  int32_t  iterations;                     // iterations executed by the last rebuild
  uint32_t status;
  uint32_t rows;
  double   last_error;
};

struct FqiParams {
  DevParams env;                           // h, integration_steps, timeout, randomization of the pendulum
  int32_t  R, H, P, A, batch_size, cap, chunks_cap;
  double   actions[kMaxActions];
  double   in_min[kFqiNIn], in_scale[kFqiNIn], obs_min[kFqiD], obs_range[kFqiD];
  double   action_min, action_range;
  double   gamma_tau;
  double   eta;                                     // representation/parameterized/ann:eta (0: RPROP, > 0: gradient descent, < 0: RMSprop)
  FqiRep  *rep;
  double  *in, *next_obs, *reward, *targets;        // [R][cap][...]
  int32_t *absorbing;
  double  *net;                                     // [R][4][P]: params, eta, Delta (unused), prev_Delta
  double  *vL;                                      // [2][R][P + 1][64]: the 64 partial sums of level 2 per parameter, double-buffered by epoch parity
  unsigned long long *stamps;                       // diagnostic (GRLX_FQI_STAMPS=1): [R][16][4 epochs][8] shader-clock stamps, else null
  unsigned int *sync;                               // [R][4]: arrival counter of the replica's blocks (16-byte slots), zeroed per launch
  double  *row_reward;                              // [R][max_rows]
  int64_t *row_batch, *row_transitions;
  int32_t  max_rows;
};

// ANNRepresentation::read (ann.cpp:133-160): net input ((w0 a0 + w1 a1) + w2 a2) + bias, logistic hidden layer, linear output.
// Written stage by stage over the hidden units (per unit the operations and their order are those of the scalar text:
// oracle/fqi.c ann_forward): independent chains for the wave to overlap -- a branchy exp per unit serialises them.  Up to 20
// units are one batch; wider layers go 16 at a time (two batches of stage values would not fit the registers), the output sum
// still adds the units in their order.  a_out: registers (H <= 20) or the lane's row of the factor tile in LDS.
template <int H>
__device__ __forceinline__ double ann_forward(const double *w, const double (&in)[kFqiNIn], double *a_out)
{
  constexpr int CH = (H <= 20) ? H : 16;
  static_assert(H % CH == 0, "hidden units per batch");
  const double *W2 = w + (kFqiNIn + 1) * H;
  double out = 0;
#pragma unroll
  for (int h0 = 0; h0 < H; h0 += CH)
  {
    double net[CH], e[CH];
#pragma unroll
    for (int h = 0; h < CH; ++h)
    {
      double v = 0;
#pragma unroll
      for (int i = 0; i < kFqiNIn; ++i) v += w[(h0 + h) * (kFqiNIn + 1) + i] * in[i];
      v += w[(h0 + h) * (kFqiNIn + 1) + kFqiNIn];
      net[h] = -v;
    }
    plogistic_batch<CH>(net, e);                                  // a = 1 / (1 + exp(-net)), ann.h:108-111; -net clamped to +-690 (fqi.c D5)
#pragma unroll
    for (int h = 0; h < CH; ++h)
    {
      if (a_out) a_out[h0 + h] = e[h];
      out += W2[h0 + h] * e[h];
    }
  }
  out += W2[H];
  return out;
}

__device__ __forceinline__ void fqi_normalise(const FqiParams &F, const double *obs, double action, double (&in)[kFqiNIn])
{ // NormalizingProjector::project (normalizing.cpp:78-87), signed = 0
#pragma unroll
  for (int i = 0; i < kFqiD; ++i) in[i] = (obs[i] - F.in_min[i]) * F.in_scale[i] - 0;
  in[kFqiD] = (action - F.in_min[kFqiD]) * F.in_scale[kFqiD] - 0;
}

// weights: w[i] = 0.01 * (2 u_i - 1) from the initialisation stream (fqi.c D1); eta = Ones, x 0.1 for RPROP (ann.cpp:106-109)
__global__ void fqi_init_kernel(FqiParams F, const uint64_t *r0)
{
  const int r = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= F.P) return;
  double *net = F.net + (size_t)r * 4 * F.P;
  const uint64_t x = lcg_next(lcg_jump(r0[r], (uint64_t)k));
  net[k] = (2 * lcg_double(x) - 1) * 0.01;
  net[F.P + k] = (F.eta == 0) ? 1. * 0.1 : 1.;
  net[2 * F.P + k] = 0;
  net[3 * F.P + k] = 0;
}

// BatchLearningExperiment::run, the sampling loop (batch_learning.cpp:107-138): one lane per transition; the
// replica's thread-local stream is consumed 4 draws per transition in sample order (LCG jump-ahead)
__global__ __launch_bounds__(256) void fqi_generate_kernel(FqiParams F)
{
  const int r = blockIdx.y, s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= F.batch_size) return;
  const FqiRep &rep = F.rep[r];
  uint64_t x = lcg_jump(rep.TL, 4ull * (uint64_t)s);
  double obs[kFqiD];
#pragma unroll
  for (int i = 0; i < kFqiD; ++i) { x = lcg_next(x); obs[i] = 0 + lcg_double(x) * (1 - 0); }
  x = lcg_next(x);
  double action = 0 + lcg_double(x) * (1 - 0);
  // (the fourth draw is next_action: consumed, never read by FQI)
#pragma unroll
  for (int i = 0; i < kFqiD; ++i) obs[i] = F.obs_min[i] + obs[i] * F.obs_range[i];
  action = F.action_min + action * F.action_range;
  double st[kFqiS] = {obs[0] - GRLX_PI, obs[1], 0.};                          // PendulumSwingupTask::invert (pendulum.cpp:147-155)
  double nobs[kFqiD], reward;
  int terminal;
  uint32_t status = 0;
  env_step<GRLX_ENV_PENDULUM, false>(F.env, st, action, nobs, reward, terminal, status);
  const size_t at = (size_t)r * (size_t)F.cap + (size_t)rep.n + (size_t)s;
  double in[kFqiNIn];
  fqi_normalise(F, obs, action, in);
#pragma unroll
  for (int i = 0; i < kFqiNIn; ++i) F.in[at * kFqiNIn + i] = in[i];
#pragma unroll
  for (int i = 0; i < kFqiD; ++i) F.next_obs[at * kFqiD + i] = nobs[i];
  F.reward[at] = reward;
  F.absorbing[at] = terminal == 2 ? 1 : 0;
  F.targets[at] = 0.;
  if (status) atomicOr(&F.rep[r].status, status);
}

// after the generate kernel: the batch is part of the store, the stream has moved on, a rebuild begins
__global__ void fqi_batch_begin_kernel(FqiParams F)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= F.R) return;
  FqiRep &rep = F.rep[r];
  rep.TL = lcg_jump(rep.TL, 4ull * (uint64_t)F.batch_size);
  rep.n += F.batch_size;
  rep.done = 0;
  rep.iterations = 0;
  rep.maxdelta_bits = 0x7FF0000000000000ull;                                  // +inf (fqi.cpp:208)
}

// head of iteration ii of FQIPredictor::rebuild (fqi.cpp:213): `ii < iterations_ && maxdelta > 0.001`
__global__ void fqi_iter_begin_kernel(FqiParams F, int ii)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= F.R) return;
  FqiRep &rep = F.rep[r];
  if (rep.done) return;
  const double md = __longlong_as_double((long long)rep.maxdelta_bits);
  rep.maxdelta_last = md;
  if (!(md > 0.001)) { rep.done = 1; return; }
  rep.iterations = ii + 1;
  rep.maxdelta_bits = 0ull;
}

// targets of one iteration (fqi.cpp:227-262): r + gamma^tau max_a Q(s', a); L-infinity change against the previous targets
template <int H>
__global__ __launch_bounds__(256) void fqi_targets_kernel(FqiParams F, int first_iteration)
{
  __shared__ double sh_net[(kFqiNIn + 1) * H + H + 1];
  __shared__ double sh_max[4];
  const int r = blockIdx.y;
  const FqiRep &rep = F.rep[r];
  if (rep.done) return;
  const double *net = F.net + (size_t)r * 4 * F.P;
  for (int k = threadIdx.x; k < F.P; k += blockDim.x) sh_net[k] = net[k];
  __syncthreads();
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double diff = 0;
  if (s < rep.n)
  {
    const size_t at = (size_t)r * (size_t)F.cap + (size_t)s;
    double target = F.reward[at];
    if (!F.absorbing[at])
    {
      double nobs[kFqiD];
#pragma unroll
      for (int i = 0; i < kFqiD; ++i) nobs[i] = F.next_obs[at * kFqiD + i];
      double v = -__builtin_inf();
      for (int k = 0; k < F.A; ++k)
      {
        double in[kFqiNIn];
        fqi_normalise(F, nobs, F.actions[k], in);
        v = fmax(v, ann_forward<H>(sh_net, in, nullptr));
      }
      target += F.gamma_tau * v;
    }
    const double prev = first_iteration ? 0. : F.targets[at];
    diff = __builtin_fabs(prev - target);
    F.targets[at] = target;
  }
  // max is order-independent: any reduction gives the reference's value
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) diff = fmax(diff, __shfl_down(diff, off, 64));
  if ((threadIdx.x & 63) == 0) sh_max[threadIdx.x >> 6] = diff;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    const double m = fmax(fmax(sh_max[0], sh_max[1]), fmax(sh_max[2], sh_max[3]));
    atomicMax(&F.rep[r].maxdelta_bits, (unsigned long long)__double_as_longlong(m));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The epochs of one iteration in ONE launch: `fqi_epochs_kernel`.
//
// An epoch = the gradient of every stored transition (ann.cpp:224-263), summed, and one RPROP step (ann.cpp:186-192).  The
// specification's sum (oracle/fqi.c) is a fixed tree: level 1 = a chunk of 64 samples in sample order; level 2 = 64 partial
// sums per parameter, the L-th over the chunks L, L+64, L+128, ... in this order; level 3 = v[L] += v[L + off], off = 32..1.
// Level 2 is the mapping: wave (replica r, L) walks ITS chunks in order and keeps its partial sums in registers -- no
// per-chunk sums ever reach memory (round 2 wrote and re-read 41 MB of them per epoch at 16 x 200 000 transitions, in two
// launches per epoch whose dispatch gaps were a quarter of the time).  A block = 4 waves = 4 consecutive L of one replica;
// 16 blocks per replica; 16 replicas fill the 256 CUs with one block each (93 KB of LDS per block: four factor tiles) -- for the
// 20 hidden units of the reference's file; FqiShape<H> gives the numbers of the other instantiated widths (8, 16, 32, 64).
// One wave per SIMD means nothing but the wave's own instruction-level parallelism hides latency: the forward pass is
// written stage by stage over the 20 hidden units (pexp_batch, pdiv_batch: grlx_math.h), the next chunk's inputs are
// loaded while the current one is computed.
// Once per epoch the 64 waves of a replica meet (agent-scope hand-off, cdna_hip_programming.md Guideline 16): every wave
// publishes its 102 partial sums (write-through stores, double-buffered by epoch parity), the block's leader adds 1 to the
// replica's arrival counter and polls it; behind an agent-scope acquire EVERY block runs level 3 and the RPROP step for
// ALL parameters itself (wave w: parameters w, w+4, ...: lane l loads partial sum l, the tree runs in the wave, lane 0
// steps).  The 16 copies are bit-identical, so the network, eta and the previous gradient live in each block's LDS for
// the whole launch and nobody waits for a second hand-off; block 0 of the replica writes them back at the end.
// The replicas never wait for each other.  All blocks of a launch must be resident (they spin on each other): the launch is
// cooperative (the runtime refuses a grid that does not fit) and every spin is bounded -- a wait that runs out raises
// ST_SYNC_TIMEOUT and the block leaves, so the grid always drains.
typedef __attribute__((address_space(1))) unsigned int fqi_gu32;
typedef __attribute__((address_space(1))) unsigned long long fqi_gu64;
#define FQI_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr uint32_t ST_SYNC_TIMEOUT = 32u;
constexpr unsigned kFqiSpinLimit = 4u << 20;            // x (s_sleep 2 + one L2 round trip) = seconds

__device__ __forceinline__ void fqi_publish(double *p, double x)
{ // write-through (sc1) 8-byte store: visible to other XCDs without a release fence once drained
  __hip_atomic_store((fqi_gu64 *)p, (unsigned long long)__double_as_longlong(x), FQI_RLX_AGENT);
}

// All threads of the block call this after their publishing stores.  Returns false when the wait timed out.
__device__ __forceinline__ bool fqi_meet(unsigned int *counter, unsigned target, int *sh_ok)
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its stores ...
  __syncthreads();                                       // ... before the leader signals for all of them
  if (threadIdx.x == 0)
  {
    __hip_atomic_fetch_add((fqi_gu32 *)counter, 1u, FQI_RLX_AGENT);
    bool ok = true;
    for (unsigned spins = 0; __hip_atomic_load((fqi_gu32 *)counter, FQI_RLX_AGENT) < target;)
    {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > kFqiSpinLimit) { ok = false; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // ONE acquire after the match: drops this CU's stale lines
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *sh_ok = ok ? 1 : 0;
  }
  __syncthreads();
  return *sh_ok != 0;
}

// lane i of a row of 16 receives the value of lane i + N of the same row (DPP row_shl: N); lanes past the row's end keep their own
template <int N>
__device__ __forceinline__ double fqi_row_shl(double v)
{
  const long long b = __double_as_longlong(v);
  const int lo = (int)b, hi = (int)(b >> 32);
  const int rlo = __builtin_amdgcn_update_dpp(lo, lo, 0x100 + N, 0xF, 0xF, false);
  const int rhi = __builtin_amdgcn_update_dpp(hi, hi, 0x100 + N, 0xF, 0xF, false);
  return __longlong_as_double(((long long)rhi << 32) | (unsigned int)rlo);
}

// The shape of the epochs kernel for H hidden units.  A lane sums KP parameters (lane, lane + 64, ...: P + 1 <= 64 KP, the extra one
// is the squared error); a wave's factor tile is 64 samples x COLS doubles (COLS odd: a column read by 64 lanes hits 64 banks); as many
// waves per block as their tiles fit the CU's 160 KB of LDS (4 up to 32 hidden units, 2 for 64), 64 waves = the 64 partial sums of
// level 2 per replica.  LDS reads of level 1 go SPB samples (2 KP SPB <= 12 reads) at a time.
template <int H> struct FqiShape {
  static constexpr int P = (kFqiNIn + 1) * H + H + 1;
  static constexpr int KP = (P + 1 + 63) / 64;
  static constexpr int COLS = 2 * H + kFqiNIn + 2;        // in[3], d1[H], a[H], d2, 1.0
  static constexpr int WAVES = ((size_t)4 * 64 * COLS * 8 + 3 * P * 8 + 64 <= 160 * 1024) ? 4 : 2;
  static constexpr int BLOCKS = 64 / WAVES;               // per replica
  // KP == 2 (16 and 20 hidden units): a lane's two terms SHARE a factor (below), three LDS reads per sample instead of four
  static constexpr bool SHARED = KP == 2 && 2 * H + (H + 3) / 2 <= 64;
  static constexpr int RS = SHARED ? 3 : 2 * KP;          // LDS reads per sample and lane
  static constexpr int SPB = 12 / RS;
  static constexpr int QSPLIT = (63 * COLS * 8 <= 65535) ? 64 : 32;      // ds_read offsets are 16 bits: a second base address beyond
  static_assert(KP <= 6 && RS * SPB <= 12 && SPB >= 1, "reads per batch");
  static_assert((size_t)WAVES * 64 * COLS * 8 + 3 * P * 8 + 64 <= 160 * 1024, "factor tiles exceed the LDS");
};
constexpr int kFqiStampBlocks = 16;

// s_waitcnt lgkmcnt(N) that the compiler may not move the uses of x[] across
template <int N>
__device__ __forceinline__ void fqi_lds_wait(double (&x)[12])
{
  asm volatile("s_waitcnt lgkmcnt(%12)"
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11])
               : "n"(N));
}

template <int H>
__global__ __launch_bounds__(FqiShape<H>::WAVES * 64) void fqi_epochs_kernel(FqiParams F, int r_first, int epochs)
{
  typedef FqiShape<H> Sh;
  constexpr int P = Sh::P, KP = Sh::KP, COLS = Sh::COLS, WAVES = Sh::WAVES, BLOCKS = Sh::BLOCKS, SPB = Sh::SPB, QSPLIT = Sh::QSPLIT, NT = WAVES * 64;
  constexpr int C_IN = 0, C_D1 = kFqiNIn, C_A = kFqiNIn + H, C_D2 = kFqiNIn + 2 * H, C_ONE = C_D2 + 1;
  constexpr int TREES = (P + 1 + WAVES - 1) / WAVES;     // level-3 trees per wave: parameters w, w + WAVES, ... (and the squared error)
  constexpr int GROUPS = (TREES + 3) / 4;                // ... taken four at a time, one per row of 16 lanes
  constexpr int GB = 8;                                  // ... and at most 8 groups of leaves in registers together
  static_assert(COLS == C_ONE + 1, "factor tile width");
  __shared__ double sh_net[P], sh_eta[P], sh_prev[P];
  __shared__ double sh_tile[WAVES][64 * COLS];
  __shared__ int sh_ok;
  const int r = r_first + (int)blockIdx.x / BLOCKS, bl = (int)blockIdx.x % BLOCKS;
  if (r >= F.R || epochs <= 0) return;
  FqiRep &rep = F.rep[r];
  if (rep.done) return;                                  // uniform over the replica's blocks
  const int w = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  const int L = bl * WAVES + w;
  const int64_t n = rep.n, chunks = (n + 63) / 64;
  double *net = F.net + (size_t)r * 4 * F.P;
  unsigned int *counter = F.sync + (size_t)r * 4;        // one 16-byte slot per replica
  double *tile = sh_tile[w];
  double *row = tile + lane * COLS;
  for (int k = (int)threadIdx.x; k < P; k += NT)
  {
    sh_net[k] = net[k];
    sh_eta[k] = net[F.P + k];
    sh_prev[k] = net[3 * F.P + k];
  }
  // the two factors of the per-sample gradient terms this lane sums (x * 1.0 is exact) and the parameter each sum belongs to (P: the squared
  // error, ann.cpp:240; -1: none).  General layout: parameters lane, lane + 64, ...  SHARED layout (two terms per lane): the two share one
  // factor, so a sample costs three LDS reads instead of four -- layer 1: lane (h, pair) sums Delta1(2 pair, h) and Delta1(2 pair + 1, h),
  // both times d1[h]; layer 2, its bias and the error: neighbours two by two, all times d2.  Which lane sums a parameter does not touch the
  // order of its sum (samples of a chunk in order, then the fixed tree).
  constexpr bool SHARED = Sh::SHARED;
  int cx[KP], cy[KP], pidx[KP];
  if constexpr (SHARED)
  {
    constexpr int L1 = 2 * H, E2 = H + 2, L2 = (E2 + 1) / 2;
    static_assert(L1 + L2 <= 64, "lanes of the shared-factor layout");
#pragma unroll
    for (int k = 0; k < 2; ++k)
    {
      if (lane < L1)
      {
        const int h = lane >> 1, i = 2 * (lane & 1) + k;
        cx[k] = (i < kFqiNIn) ? C_IN + i : C_ONE;                  // Delta1(i, h) += a0[i] * d1[h]; bias row: d1[h]
        cy[k] = C_D1 + h;
        pidx[k] = h * (kFqiNIn + 1) + i;
      }
      else
      {
        const int q = 2 * (lane - L1) + k;
        cy[k] = C_D2;
        if (lane < L1 + L2 && q < H) { cx[k] = C_A + q; pidx[k] = (kFqiNIn + 1) * H + q; }   // Delta2(h) += a1[h] * d2
        else if (lane < L1 + L2 && q == H) { cx[k] = C_ONE; pidx[k] = P - 1; }                // bias: d2
        else if (lane < L1 + L2 && q == H + 1) { cx[k] = C_D2; pidx[k] = P; }                 // squared error
        else { cx[k] = C_D2; pidx[k] = -1; }
      }
    }
  }
  else
  {
#pragma unroll
    for (int k = 0; k < KP; ++k)
    {
      const int p = lane + 64 * k;
      pidx[k] = (p <= P) ? p : -1;
      if (p < (kFqiNIn + 1) * H)
      {
        const int h = p / (kFqiNIn + 1), i = p % (kFqiNIn + 1);
        cx[k] = (i < kFqiNIn) ? C_IN + i : C_ONE;                  // Delta1(i, h) += a0[i] * d1[h]; bias row: d1[h]
        cy[k] = C_D1 + h;
      }
      else if (p < P - 1) { cx[k] = C_A + (p - (kFqiNIn + 1) * H); cy[k] = C_D2; }   // Delta2(h) += a1[h] * d2
      else if (p == P - 1) { cx[k] = C_ONE; cy[k] = C_D2; }                          // bias: d2
      else { cx[k] = C_D2; cy[k] = C_D2; }                                           // p == P: squared error (ann.cpp:240); beyond: unused
    }
  }
  row[C_ONE] = 1.;                                        // the constant factor of the bias terms: written once
  const size_t base = (size_t)r * (size_t)F.cap;
  double last_error = 0.;
  bool has_error = false;                                 // this lane ran the tree of the squared error
  __syncthreads();
#define FQI_STAMP(k) do { if (F.stamps && e < 4 && bl < kFqiStampBlocks && threadIdx.x == 0) F.stamps[(((size_t)r * kFqiStampBlocks + bl) * 4 + e) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
  for (int e = 0; e < epochs; ++e)
  {
    FQI_STAMP(0);
    double acc[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) acc[k] = 0.;
    // inputs of the wave's first chunk; inside the loop the next chunk's are requested before the current one is worked on
    double nin[kFqiNIn] = {0., 0., 0.}, ntarget = 0.;
    if (L < chunks && (int64_t)L * 64 + lane < n)
    {
      const size_t at = base + (size_t)L * 64 + (size_t)lane;
#pragma unroll
      for (int i = 0; i < kFqiNIn; ++i) nin[i] = F.in[at * kFqiNIn + i];
      ntarget = F.targets[at];
    }
    for (int64_t c = L; c < chunks; c += 64)
    { // level 1: one chunk of 64 samples; lane = sample for the forward / backward pass (ann.cpp:224-263)
      const int64_t s = c * 64 + lane;
      double in[kFqiNIn];
#pragma unroll
      for (int i = 0; i < kFqiNIn; ++i) in[i] = nin[i];
      const double target = ntarget;
      if (c + 64 < chunks && s + 64 * 64 < n)
      {
        const size_t at = base + (size_t)s + 64 * 64;
#pragma unroll
        for (int i = 0; i < kFqiNIn; ++i) nin[i] = F.in[at * kFqiNIn + i];
        ntarget = F.targets[at];
      }
      if (c == L) FQI_STAMP(5);
      if (s < n)
      {
        const double *W2 = sh_net + (kFqiNIn + 1) * H;
        if constexpr (H <= 20)
        { // the activations stay in registers between the two passes
          double a[H];
          const double out = ann_forward<H>(sh_net, in, a);
          if (c == L) FQI_STAMP(6);
          const double d2 = out - target;
#pragma unroll
          for (int i = 0; i < kFqiNIn; ++i) row[C_IN + i] = in[i];
#pragma unroll
          for (int h = 0; h < H; ++h)
          {
            row[C_D1 + h] = (W2[h] * d2) * (a[h] * (1. - a[h]));       // ann.cpp:249 with deviation D2
            row[C_A + h] = a[h];
          }
          row[C_D2] = d2;
        }
        else
        { // the activations go to the lane's row as they are produced and are read back for the hidden deltas
          const double out = ann_forward<H>(sh_net, in, row + C_A);
          asm volatile("" ::: "memory");                               // (read back from the row: do not keep H activations live in registers)
          if (c == L) FQI_STAMP(6);
          const double d2 = out - target;
#pragma unroll
          for (int i = 0; i < kFqiNIn; ++i) row[C_IN + i] = in[i];
#pragma unroll
          for (int h = 0; h < H; ++h)
          {
            const double ah = row[C_A + h];
            row[C_D1 + h] = (W2[h] * d2) * (ah * (1. - ah));           // ann.cpp:249 with deviation D2
          }
          row[C_D2] = d2;
        }
      }
      // the tile belongs to this wave alone: its LDS writes are complete before its LDS reads issue (in-order LDS, one wave)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int n_here = (int)((n - c * 64 < 64) ? n - c * 64 : 64);
      if (c == L) FQI_STAMP(7);
      { // the lane turns into one lane per PARAMETER (KP of them): per-sample products added in sample order (level 1), then level 2
        double v[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) v[k] = 0.;
        if (n_here == 64)
        { // 128 KP LDS reads per wave and chunk: issued as plain ds_read_b64 (2 LDS cycles each; the compiler pairs neighbouring rows
          // into ds_read2_b64, 8 cycles per pair: MI355X_MICROARCH.md, LDS table), SPB samples (<= 12 reads) per batch, the next
          // batch in flight while the current one is multiplied and added (lgkmcnt counts at most 15 operations)
          constexpr int RS = Sh::RS, RPB = RS * SPB, NB = 64 / SPB, REM = 64 % SPB;
          double x[2][12];
          const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) double *)tile;
          unsigned a_cx[64 / QSPLIT][KP], a_cy[64 / QSPLIT][KP];
#pragma unroll
          for (int hf = 0; hf < 64 / QSPLIT; ++hf)
#pragma unroll
            for (int k = 0; k < KP; ++k)
            {
              a_cx[hf][k] = lds0 + (unsigned)(hf * QSPLIT * COLS + cx[k]) * 8u;
              a_cy[hf][k] = lds0 + (unsigned)(hf * QSPLIT * COLS + cy[k]) * 8u;
            }
#define FQI_RD(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define FQI_ISSUE(buf, q0, cnt)                                                                                           \
          _Pragma("unroll") for (int t = 0; t < (cnt); ++t)                                                                \
          {                                                                                                               \
            if constexpr (SHARED)                                                                                         \
            {                                                                                                             \
              FQI_RD(x[buf][3 * t + 0], a_cy[((q0) + t) / QSPLIT][0], (((q0) + t) % QSPLIT) * COLS * 8);                   \
              FQI_RD(x[buf][3 * t + 1], a_cx[((q0) + t) / QSPLIT][0], (((q0) + t) % QSPLIT) * COLS * 8);                   \
              FQI_RD(x[buf][3 * t + 2], a_cx[((q0) + t) / QSPLIT][1], (((q0) + t) % QSPLIT) * COLS * 8);                   \
            }                                                                                                             \
            else                                                                                                          \
            {                                                                                                             \
              _Pragma("unroll") for (int k = 0; k < KP; ++k)                                                               \
              {                                                                                                           \
                FQI_RD(x[buf][2 * (t * KP + k) + 0], a_cx[((q0) + t) / QSPLIT][k], (((q0) + t) % QSPLIT) * COLS * 8);        \
                FQI_RD(x[buf][2 * (t * KP + k) + 1], a_cy[((q0) + t) / QSPLIT][k], (((q0) + t) % QSPLIT) * COLS * 8);        \
              }                                                                                                           \
            }                                                                                                             \
          }
#define FQI_USE(buf, cnt)                                                                                                 \
          _Pragma("unroll") for (int t = 0; t < (cnt); ++t)                                                                \
          {                                                                                                               \
            if constexpr (SHARED)                                                                                         \
            {                                                                                                             \
              v[0] += x[buf][3 * t + 1] * x[buf][3 * t + 0];                                                              \
              v[1] += x[buf][3 * t + 2] * x[buf][3 * t + 0];                                                              \
            }                                                                                                             \
            else                                                                                                          \
            {                                                                                                             \
              _Pragma("unroll") for (int k = 0; k < KP; ++k) v[k] += x[buf][2 * (t * KP + k) + 0] * x[buf][2 * (t * KP + k) + 1]; \
            }                                                                                                             \
          }
#define FQI_BATCH(cur, nxt)                                                                                               \
          {                                                                                                               \
            if (b + 1 < NB) { FQI_ISSUE(nxt, SPB * (b + 1), SPB); fqi_lds_wait<RPB>(x[cur]); }                             \
            else if (REM) { FQI_ISSUE(nxt, SPB * NB, REM); fqi_lds_wait<RS * REM>(x[cur]); }                               \
            else fqi_lds_wait<0>(x[cur]);                                                                                 \
            FQI_USE(cur, SPB);                                                                                            \
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the tile writes of this wave have left for the LDS
          FQI_ISSUE(0, 0, SPB);
#pragma unroll
          for (int b = 0; b < NB; ++b)
          { // batch b = samples SPB b .. SPB b + SPB - 1 (and one last batch of the REM samples that remain)
            if (b & 1) FQI_BATCH(1, 0)
            else FQI_BATCH(0, 1)
          }
          if (REM)
          {
            fqi_lds_wait<0>(x[NB & 1]);
            FQI_USE(NB & 1, REM);
          }
#undef FQI_RD
#undef FQI_ISSUE
#undef FQI_USE
#undef FQI_BATCH
        }
        else
          for (int q = 0; q < n_here; ++q)
#pragma unroll
            for (int k = 0; k < KP; ++k) v[k] += tile[q * COLS + cx[k]] * tile[q * COLS + cy[k]];
#pragma unroll
        for (int k = 0; k < KP; ++k) acc[k] += v[k];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    FQI_STAMP(1);
    // the hand-off: the 64 partial sums of every parameter (buffer e mod 2: a block that is one epoch ahead writes the other one)
    double *vL = F.vL + ((size_t)(e & 1) * (size_t)F.R + (size_t)r) * 64 * (size_t)(P + 1);
#pragma unroll
    for (int k = 0; k < KP; ++k)                                     // [parameter][L]: the 64 leaves of a tree are 512 contiguous bytes
      if (pidx[k] >= 0) fqi_publish(vL + (size_t)pidx[k] * 64 + L, acc[k]);
    if (!fqi_meet(counter, (unsigned)(e + 1) * BLOCKS, &sh_ok))
    {
      if (threadIdx.x == 0) atomicOr(&rep.status, ST_SYNC_TIMEOUT);
      return;
    }
    FQI_STAMP(2);
    // level 3 and the step of ANNRepresentation::finalize (ann.cpp:198-221), in every block for all parameters
    // Four trees at a time, one per row of 16 lanes: lane j of row t loads leaves j, j+32, j+16, j+48 of its tree and adds them as
    // levels 32 and 16 do -- (v[j] + v[j+32]) + (v[j+16] + v[j+48]) --, levels 8..1 are row shifts (DPP); lane 0 of the row holds the
    // sum and steps the parameter, whose eta / previous gradient live in LDS beside the network.
    const int rowi = lane >> 4, j16 = lane & 15;
#pragma unroll
    for (int g0 = 0; g0 < GROUPS; g0 += GB)
    {
      constexpr int GBmax = GB;
      double leaf[GBmax][4];
#pragma unroll
      for (int gg = 0; gg < GBmax; ++gg)
        if (g0 + gg < GROUPS)
        {
          const int p = w + WAVES * (4 * (g0 + gg) + rowi);
          const double *t = vL + (size_t)(p <= P ? p : P) * 64 + j16;
#pragma unroll
          for (int k = 0; k < 4; ++k) leaf[gg][k] = t[16 * k];
        }
      double st_eta[GBmax], st_prev[GBmax], st_net[GBmax];
#pragma unroll
      for (int gg = 0; gg < GBmax; ++gg)
        if (g0 + gg < GROUPS)
        {
          const int p = w + WAVES * (4 * (g0 + gg) + rowi), pc = p < P ? p : 0;
          st_eta[gg] = sh_eta[pc]; st_prev[gg] = sh_prev[pc]; st_net[gg] = sh_net[pc];
        }
#pragma unroll
      for (int gg = 0; gg < GBmax; ++gg)
        if (g0 + gg < GROUPS)
        {
          double v = (leaf[gg][0] + leaf[gg][2]) + (leaf[gg][1] + leaf[gg][3]);
          v += fqi_row_shl<8>(v);
          v += fqi_row_shl<4>(v);
          v += fqi_row_shl<2>(v);
          v += fqi_row_shl<1>(v);
          const int p = w + WAVES * (4 * (g0 + gg) + rowi);
          if (j16 == 0)
          {
            if (p == P) { last_error = v / (double)n; has_error = true; }
            else if (p < P)
            {
              const double Delta = 0. + v;                             // Delta was zero before this epoch (ann.cpp:221)
              if (F.eta == 0)
              { // RPROP (ann.cpp:207-213)
                const double eta = (Delta * st_prev[gg] > 0) ? st_eta[gg] * 1.2 : st_eta[gg] * 0.5;
                sh_net[p] = st_net[gg] - ((Delta > 0) ? eta : -eta);
                sh_eta[p] = eta;
                sh_prev[p] = Delta;
              }
              else if (F.eta > 0)
                sh_net[p] = st_net[gg] - (F.eta * Delta) / (double)n;   // gradient descent (:202-206), samples_ = n
              else
              { // RMSprop (:214-219); sqrt and the divisions are the correctly rounded IEEE operations
                const double gm = Delta / (double)n;
                const double eta = 0.9 * st_eta[gg] + 0.1 * (gm * gm);
                sh_eta[p] = eta;
                sh_net[p] = st_net[gg] + F.eta * (Delta / __builtin_sqrt(eta));
              }
            }
          }
        }
    }
    FQI_STAMP(3);
    __syncthreads();
    FQI_STAMP(4);
  }
#undef FQI_STAMP
  if (bl == 0)
  { // the blocks hold the same bits: one of them writes the state back for the kernels and launches that follow
    for (int k = (int)threadIdx.x; k < P; k += NT)
    {
      net[k] = sh_net[k];
      net[F.P + k] = sh_eta[k];
      net[3 * F.P + k] = sh_prev[k];
    }
    if (has_error) rep.last_error = last_error;
  }
}

// the test trial after a batch (batch_learning.cpp:141-186): agent/fixed + policy/discrete/q + sampler/greedy; one lane per replica
template <int H>
__global__ __launch_bounds__(64) void fqi_test_kernel(FqiParams F)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= F.R) return;
  FqiRep &rep = F.rep[r];
  const double *net = F.net + (size_t)r * 4 * F.P;
  uint64_t TL = rep.TL, G = rep.G;
  double x[kFqiS], obs[kFqiD], reward = 0, total = 0;
  int terminal = 0;
  uint32_t status = 0;
  Env<GRLX_ENV_PENDULUM>::start(F.env, 1, TL, G, x);
  Env<GRLX_ENV_PENDULUM>::observe(F.env, x, obs);
  for (int guard = 0; guard < GRLX_MAX_EPISODE_STEPS + 2; ++guard)
  {
    double q[kMaxActions];
    for (int k = 0; k < F.A; ++k)
    {
      double in[kFqiNIn];
      fqi_normalise(F, obs, F.actions[k], in);
      q[k] = ann_forward<H>(net, in, nullptr);
    }
    int mai = 0, man = 1;                                                    // GreedySampler::sample (greedy.cpp:47-86)
    for (int k = 1; k < F.A; ++k)
    {
      if (q[k] > q[mai]) { mai = k; man = 1; }
      else if (q[k] == q[mai]) man++;
    }
    if (man > 1)
    {
      G = lcg_next(G);
      int jj = (int)(lcg_long(G) % (uint32_t)man);
      for (int k = 0; k < F.A; ++k)
        if (q[k] == q[mai])
        {
          if (jj == 0) { mai = k; break; }
          --jj;
        }
    }
    if (terminal) break;
    env_step<GRLX_ENV_PENDULUM, false>(F.env, x, F.actions[mai], obs, reward, terminal, status);
    total += reward;
    if (terminal == 2) break;
  }
  if (rep.rows < (uint32_t)F.max_rows)
  {
    const size_t at = (size_t)r * (size_t)F.max_rows + rep.rows;
    F.row_reward[at] = total;
    F.row_batch[at] = rep.batches_done;
    F.row_transitions[at] = rep.batches_done * (int64_t)F.batch_size;
    rep.rows++;
  }
  else
    status |= ST_ROWS_FULL;
  rep.batches_done++;
  rep.TL = TL;
  rep.G = G;
  rep.status |= status;
}

} // namespace grlx

using namespace grlx;

struct grlx_fqi_ctx {
  grlx_fqi_config cfg;
  FqiParams F;
  std::vector<void *> bufs;
  uint64_t *r0 = nullptr;
  int batches_run = 0;
  int replicas_per_launch = 1;       // how many replicas' blocks are resident at once
  int blocks_per_replica = 16, block_threads = 256;      // FqiShape<hidden>
  bool plain_launch = false;         // diagnostic: GRLX_FQI_NO_COOP=1 launches the epochs kernel without hipLaunchCooperativeKernel
};

namespace {
// the instantiated widths of the hidden layer (representation/parameterized/ann:hiddens = [H])
#define FQI_FOR_H(H_, ...)                                           \
  switch (H_)                                                        \
  {                                                                  \
    case 8:  { constexpr int HH = 8;  __VA_ARGS__; } break;          \
    case 16: { constexpr int HH = 16; __VA_ARGS__; } break;          \
    case 20: { constexpr int HH = 20; __VA_ARGS__; } break;          \
    case 32: { constexpr int HH = 32; __VA_ARGS__; } break;          \
    case 64: { constexpr int HH = 64; __VA_ARGS__; } break;          \
    default: break;                                                  \
  }
inline bool fqi_width_built(int H) { return H == 8 || H == 16 || H == 20 || H == 32 || H == 64; }

int ffail(int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  set_last_error(buf);
  return code;
}
#define FQI_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e__ = (expr);                                                                       \
    if (e__ != hipSuccess) return ffail(GRLX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
  } while (0)

constexpr uint64_t kA = 0x5DEECE66DULL, kC = 0xBULL, kM = (1ULL << 48) - 1;
inline uint64_t h_seed(long s) { return ((((uint64_t)s) & 0xFFFFFFFFULL) << 16) | 0x330EULL; }
inline uint64_t h_next(uint64_t x) { return (kA * x + kC) & kM; }

template <typename T> int dev_alloc(grlx_fqi_ctx *ctx, T **p, size_t n)
{
  void *q = nullptr;
  if (hipMalloc(&q, n * sizeof(T) ? n * sizeof(T) : 8) != hipSuccess) return ffail(GRLX_ERR_OOM, "out of device memory (%zu bytes)", n * sizeof(T));
  if (hipMemset(q, 0, n * sizeof(T)) != hipSuccess) return ffail(GRLX_ERR_HIP, "hipMemset failed");
  ctx->bufs.push_back(q);
  *p = (T *)q;
  return GRLX_OK;
}
} // namespace

extern "C" {

void grlx_fqi_config_pendulum(grlx_fqi_config *c)
{ // the reference's tests/pendulum-fqi-ann.yaml
  memset(c, 0, sizeof(*c));
  c->struct_size = sizeof(grlx_fqi_config);
  c->n_replicas = 1;
  c->env = GRLX_ENV_PENDULUM;
  c->control_step = 0.03;
  c->integration_steps = 5;
  c->timeout = 2.99;
  c->action_min = -3;
  c->action_max = 3;
  c->action_steps = 3;
  c->gamma = 0.97;
  c->batch_size = 1000;
  c->max_batches = 2;
  c->iterations = 10;
  c->epochs = 500;
  c->hidden = 20;
}

int grlx_fqi_create(const grlx_fqi_config *cfg, const int64_t *seeds, grlx_fqi_ctx **out)
{
  if (!cfg || !seeds || !out) return ffail(GRLX_ERR_INVALID, "null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(grlx_fqi_config)) return ffail(GRLX_ERR_INVALID, "grlx_fqi_config.struct_size mismatch");
  if (cfg->env != GRLX_ENV_PENDULUM) return ffail(GRLX_ERR_INVALID, "experiment/batch_learning: the task must support invert(); built for task/pendulum/swingup");
  if (!(cfg->eta >= -2. && cfg->eta <= 2.)) return ffail(GRLX_ERR_INVALID, "representation/parameterized/ann:eta = %g outside [-2, 2] (ann.cpp:61)", cfg->eta);
  if (!fqi_width_built(cfg->hidden))
    return ffail(GRLX_ERR_INVALID, "representation/parameterized/ann:hiddens = [%d] is not built (instantiated: [8], [16], [20], [32], [64])", cfg->hidden);
  if (cfg->n_replicas < 1 || cfg->batch_size < 1 || cfg->max_batches < 1 || cfg->iterations < 1 || cfg->epochs < 0)
    return ffail(GRLX_ERR_INVALID, "n_replicas, batch_size, max_batches, iterations must be >= 1");
  if (cfg->action_steps < 1 || cfg->action_steps > GRLX_MAX_ACTIONS) return ffail(GRLX_ERR_INVALID, "discretizer/uniform:steps");
  if (!(cfg->action_min < cfg->action_max) || !(cfg->control_step > 0) || cfg->integration_steps < 1 || !std::isfinite(cfg->timeout) || cfg->timeout < 0)
    return ffail(GRLX_ERR_INVALID, "model / task parameters");
  if (std::ceil(cfg->timeout / cfg->control_step) + 1 > (double)GRLX_MAX_EPISODE_STEPS) return ffail(GRLX_ERR_INVALID, "task:timeout exceeds GRLX_MAX_EPISODE_STEPS");
  if ((double)cfg->batch_size * cfg->max_batches > 2e8) return ffail(GRLX_ERR_INVALID, "transition store too large");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { (void)hipGetLastError(); return ffail(GRLX_ERR_NO_DEVICE, "no HIP device: grlx has no CPU fallback"); }

  grlx_fqi_ctx *ctx = new grlx_fqi_ctx();
  ctx->cfg = *cfg;
  FqiParams &F = ctx->F;
  memset(&F, 0, sizeof(F));
  F.env.env = GRLX_ENV_PENDULUM;
  F.env.integration_steps = cfg->integration_steps;
  F.env.h = cfg->control_step / (double)(size_t)cfg->integration_steps;
  F.env.timeout = cfg->timeout;
  F.env.randomization = 0;
  F.R = cfg->n_replicas;
  F.H = cfg->hidden;
  F.P = (kFqiNIn + 1) * F.H + F.H + 1;
  F.A = cfg->action_steps;
  F.batch_size = cfg->batch_size;
  F.cap = cfg->batch_size * cfg->max_batches;
  F.chunks_cap = (F.cap + 63) / 64;
  F.max_rows = cfg->max_batches;
  {
    const double range = cfg->action_max - cfg->action_min;
    double delta = range / ((double)cfg->action_steps - 1);
    if (std::isnan(delta)) delta = 0.;
    for (int k = 0; k < cfg->action_steps; ++k) F.actions[k] = cfg->action_min + delta * k;
  }
  const double omin[2] = {0., -12 * M_PI}, omax[2] = {2 * M_PI, 12 * M_PI};      // pendulum.cpp:84-85
  for (int i = 0; i < kFqiD; ++i)
  {
    F.obs_min[i] = omin[i];
    F.obs_range[i] = omax[i] - omin[i];
    F.in_min[i] = omin[i];
    F.in_scale[i] = 1. / (omax[i] - omin[i]) * (1 + 0);
  }
  F.action_min = cfg->action_min;
  F.action_range = cfg->action_max - cfg->action_min;
  F.in_min[kFqiD] = cfg->action_min;
  F.in_scale[kFqiD] = 1. / (cfg->action_max - cfg->action_min) * (1 + 0);
  F.gamma_tau = pow(cfg->gamma, cfg->control_step);                               // pow(gamma_, tau), tau = control_step
  F.eta = cfg->eta;

  const size_t R = (size_t)F.R, cap = (size_t)F.cap;
  int rc = GRLX_OK;
  if ((rc = dev_alloc(ctx, &F.rep, R)) != GRLX_OK || (rc = dev_alloc(ctx, &F.in, R * cap * kFqiNIn)) != GRLX_OK ||
      (rc = dev_alloc(ctx, &F.next_obs, R * cap * kFqiD)) != GRLX_OK || (rc = dev_alloc(ctx, &F.reward, R * cap)) != GRLX_OK ||
      (rc = dev_alloc(ctx, &F.targets, R * cap)) != GRLX_OK || (rc = dev_alloc(ctx, &F.absorbing, R * cap)) != GRLX_OK ||
      (rc = dev_alloc(ctx, &F.net, R * 4 * (size_t)F.P)) != GRLX_OK ||
      (rc = dev_alloc(ctx, &F.vL, 2 * R * 64 * (size_t)(F.P + 1))) != GRLX_OK || (rc = dev_alloc(ctx, &F.sync, R * 4)) != GRLX_OK ||
      (rc = dev_alloc(ctx, &F.row_reward, R * (size_t)F.max_rows)) != GRLX_OK || (rc = dev_alloc(ctx, &F.row_batch, R * (size_t)F.max_rows)) != GRLX_OK ||
      (rc = dev_alloc(ctx, &F.row_transitions, R * (size_t)F.max_rows)) != GRLX_OK || (rc = dev_alloc(ctx, &ctx->r0, R)) != GRLX_OK)
  {
    grlx_fqi_destroy(ctx);
    return rc;
  }
  // instantiate order of the yaml (oracle/fqi.c: orc_fqi_create): sampler/greedy Rand = global lrand48 #1, the
  // thread-local RandGen = global lrand48 #2; the weight-initialisation stream of deviation D1 = srand48_r(seed)
  std::vector<FqiRep> hr(R);
  std::vector<uint64_t> hr0(R);
  for (size_t r = 0; r < R; ++r)
  {
    memset(&hr[r], 0, sizeof(FqiRep));
    uint64_t G = h_seed((long)seeds[r]);
    G = h_next(G);                                   // sampler's private Rand (never drawn from: ties use the global stream)
    G = h_next(G);
    hr[r].TL = h_seed((long)(G >> 17));
    hr[r].G = G;
    hr0[r] = h_seed((long)seeds[r]);
  }
  if (hipMemcpy(F.rep, hr.data(), sizeof(FqiRep) * R, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(ctx->r0, hr0.data(), sizeof(uint64_t) * R, hipMemcpyHostToDevice) != hipSuccess)
  {
    grlx_fqi_destroy(ctx);
    return ffail(GRLX_ERR_HIP, "hipMemcpy failed");
  }
  {
    int dev = 0, cus = 0, per_cu = 0;
    hipError_t occ = hipErrorInvalidValue;
    FQI_FOR_H(F.H, ctx->blocks_per_replica = FqiShape<HH>::BLOCKS; ctx->block_threads = FqiShape<HH>::WAVES * 64;
              occ = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fqi_epochs_kernel<HH>, FqiShape<HH>::WAVES * 64, 0));
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        occ != hipSuccess || cus * per_cu < ctx->blocks_per_replica)
    {
      const int need = ctx->blocks_per_replica;
      grlx_fqi_destroy(ctx);
      return ffail(GRLX_ERR_HIP, "fqi_epochs_kernel: the device cannot hold the %d resident blocks of one replica", need);
    }
    ctx->replicas_per_launch = (cus * per_cu) / ctx->blocks_per_replica;
  }
  if (const char *nc = getenv("GRLX_FQI_NO_COOP")) ctx->plain_launch = nc[0] && nc[0] != '0';
  if (const char *st = getenv("GRLX_FQI_STAMPS"))
    if (st[0] && st[0] != '0' && (rc = dev_alloc(ctx, &F.stamps, R * kFqiStampBlocks * 4 * 8)) != GRLX_OK)
    {
      grlx_fqi_destroy(ctx);
      return rc;
    }
  hipLaunchKernelGGL(fqi_init_kernel, dim3((F.P + 127) / 128, F.R), dim3(128), 0, nullptr, F, ctx->r0);
  if (hipDeviceSynchronize() != hipSuccess)
  {
    grlx_fqi_destroy(ctx);
    return ffail(GRLX_ERR_HIP, "fqi_init_kernel failed");
  }
  *out = ctx;
  return GRLX_OK;
}

int grlx_fqi_destroy(grlx_fqi_ctx *ctx)
{
  if (!ctx) return GRLX_OK;
  for (void *p : ctx->bufs) (void)hipFree(p);
  delete ctx;
  return GRLX_OK;
}

int grlx_fqi_run_batch(grlx_fqi_ctx *ctx, void *stream_)
{
  if (!ctx) return ffail(GRLX_ERR_INVALID, "null ctx");
  if (ctx->batches_run >= ctx->cfg.max_batches) return ffail(GRLX_ERR_ROWS_FULL, "more batches than max_batches");
  hipStream_t stream = (hipStream_t)stream_;
  const FqiParams &F = ctx->F;
  const int n_after = (ctx->batches_run + 1) * F.batch_size;
  const int chunks = (n_after + 255) / 256;
  hipLaunchKernelGGL(fqi_generate_kernel, dim3((F.batch_size + 255) / 256, F.R), dim3(256), 0, stream, F);
  hipLaunchKernelGGL(fqi_batch_begin_kernel, dim3((F.R + 63) / 64), dim3(64), 0, stream, F);
  for (int ii = 0; ii < ctx->cfg.iterations; ++ii)
  {
    hipLaunchKernelGGL(fqi_iter_begin_kernel, dim3((F.R + 63) / 64), dim3(64), 0, stream, F, ii);
    FQI_FOR_H(F.H, hipLaunchKernelGGL(fqi_targets_kernel<HH>, dim3(chunks, F.R), dim3(256), 0, stream, F, ii == 0 ? 1 : 0));
    // the epochs of the iteration: one cooperative launch per group of replicas that fills the chip (16 blocks per replica for [20])
    for (int r0 = 0; r0 < F.R; r0 += ctx->replicas_per_launch)
    {
      const int nr = (F.R - r0 < ctx->replicas_per_launch) ? F.R - r0 : ctx->replicas_per_launch;
      FQI_TRY(hipMemsetAsync(F.sync + (size_t)r0 * 4, 0, (size_t)nr * 4 * sizeof(unsigned int), stream));
      FqiParams Fa = F;
      int r_first = r0, epochs = ctx->cfg.epochs;
      void *args[] = {&Fa, &r_first, &epochs};
      const dim3 grid(nr * ctx->blocks_per_replica), block(ctx->block_threads);
      hipError_t le = hipSuccess;
      if (ctx->plain_launch)      // diagnostic (GRLX_FQI_NO_COOP=1): the same grid without the runtime's residency check
      { FQI_FOR_H(F.H, hipLaunchKernelGGL(fqi_epochs_kernel<HH>, grid, block, 0, stream, Fa, r_first, epochs)); }
      else
      { FQI_FOR_H(F.H, le = hipLaunchCooperativeKernel((const void *)fqi_epochs_kernel<HH>, grid, block, args, 0, stream)); }
      FQI_TRY(le);
    }
  }
  FQI_FOR_H(F.H, hipLaunchKernelGGL(fqi_test_kernel<HH>, dim3((F.R + 63) / 64), dim3(64), 0, stream, F));
  FQI_TRY(hipGetLastError());
  ctx->batches_run++;
  return GRLX_OK;
}

int grlx_fqi_sync(grlx_fqi_ctx *ctx, void *stream)
{
  if (!ctx) return ffail(GRLX_ERR_INVALID, "null ctx");
  FQI_TRY(hipStreamSynchronize((hipStream_t)stream));
  std::vector<FqiRep> hr((size_t)ctx->F.R);
  FQI_TRY(hipMemcpy(hr.data(), ctx->F.rep, sizeof(FqiRep) * hr.size(), hipMemcpyDeviceToHost));
  uint32_t st = 0;
  for (const FqiRep &r : hr) st |= r.status;
  if (st & ST_DOMAIN) return ffail(GRLX_ERR_DOMAIN, "sin/cos argument outside |x| < 2^20");
  if (st & ST_ROWS_FULL) return ffail(GRLX_ERR_ROWS_FULL, "more batches than max_batches");
  if (st & ST_SYNC_TIMEOUT) return ffail(GRLX_ERR_HIP, "fqi_epochs_kernel: a wait between the blocks of a replica timed out (were all of its blocks resident?)");
  return GRLX_OK;
}

int grlx_fqi_read_rows(grlx_fqi_ctx *ctx, int replica, int first, int count, int64_t *batch, int64_t *transitions, double *reward)
{
  if (!ctx || replica < 0 || replica >= ctx->F.R || first < 0 || count < 0 || first + count > ctx->F.max_rows) return ffail(GRLX_ERR_INVALID, "bad argument");
  if (count == 0) return GRLX_OK;
  FQI_TRY(hipDeviceSynchronize());
  const size_t at = (size_t)replica * (size_t)ctx->F.max_rows + (size_t)first;
  if (batch) FQI_TRY(hipMemcpy(batch, ctx->F.row_batch + at, sizeof(int64_t) * (size_t)count, hipMemcpyDeviceToHost));
  if (transitions) FQI_TRY(hipMemcpy(transitions, ctx->F.row_transitions + at, sizeof(int64_t) * (size_t)count, hipMemcpyDeviceToHost));
  if (reward) FQI_TRY(hipMemcpy(reward, ctx->F.row_reward + at, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost));
  return GRLX_OK;
}

int grlx_fqi_get_params(grlx_fqi_ctx *ctx, int replica, double *out, int n)
{
  if (!ctx || !out || replica < 0 || replica >= ctx->F.R || n != ctx->F.P) return ffail(GRLX_ERR_INVALID, "bad argument (the network has %d parameters)", ctx ? ctx->F.P : 0);
  FQI_TRY(hipDeviceSynchronize());
  FQI_TRY(hipMemcpy(out, ctx->F.net + (size_t)replica * 4 * (size_t)ctx->F.P, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  return GRLX_OK;
}

int grlx_fqi_get_transitions(grlx_fqi_ctx *ctx, int replica, int first, int count, double *in, double *next_obs, double *reward, double *targets)
{
  if (!ctx || replica < 0 || replica >= ctx->F.R || first < 0 || count < 0 || first + count > ctx->F.cap) return ffail(GRLX_ERR_INVALID, "bad argument");
  if (count == 0) return GRLX_OK;
  FQI_TRY(hipDeviceSynchronize());
  const size_t at = (size_t)replica * (size_t)ctx->F.cap + (size_t)first;
  if (in) FQI_TRY(hipMemcpy(in, ctx->F.in + at * kFqiNIn, sizeof(double) * (size_t)count * kFqiNIn, hipMemcpyDeviceToHost));
  if (next_obs) FQI_TRY(hipMemcpy(next_obs, ctx->F.next_obs + at * kFqiD, sizeof(double) * (size_t)count * kFqiD, hipMemcpyDeviceToHost));
  if (reward) FQI_TRY(hipMemcpy(reward, ctx->F.reward + at, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost));
  if (targets) FQI_TRY(hipMemcpy(targets, ctx->F.targets + at, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost));
  return GRLX_OK;
}

// diagnostic, not part of include/grlx.h: the stamps of the LAST fqi_epochs_kernel launch (GRLX_FQI_STAMPS=1 at create)
int grlx_fqi_debug_stamps(grlx_fqi_ctx *ctx, unsigned long long *out, int count)
{
  if (!ctx || !out || !ctx->F.stamps || count != ctx->F.R * kFqiStampBlocks * 4 * 8) return ffail(GRLX_ERR_INVALID, "no stamps (GRLX_FQI_STAMPS=1 at create; count = R * 512)");
  FQI_TRY(hipDeviceSynchronize());
  FQI_TRY(hipMemcpy(out, ctx->F.stamps, sizeof(unsigned long long) * (size_t)count, hipMemcpyDeviceToHost));
  return GRLX_OK;
}

int grlx_fqi_info(grlx_fqi_ctx *ctx, int replica, int64_t *n_transitions, double *maxdelta, int32_t *iterations, double *error, uint64_t rng[2])
{
  if (!ctx || replica < 0 || replica >= ctx->F.R) return ffail(GRLX_ERR_INVALID, "bad argument");
  FQI_TRY(hipDeviceSynchronize());
  FqiRep r;
  FQI_TRY(hipMemcpy(&r, ctx->F.rep + replica, sizeof(r), hipMemcpyDeviceToHost));
  if (n_transitions) *n_transitions = r.n;
  if (maxdelta) *maxdelta = r.done ? r.maxdelta_last : __builtin_bit_cast(double, r.maxdelta_bits);
  if (iterations) *iterations = r.iterations;
  if (error) *error = r.last_error;
  if (rng) { rng[0] = r.G; rng[1] = r.TL; }
  return GRLX_OK;
}

} // extern "C"
