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

// the same jump with the table staged in LDS (rollout kernels: slot creation is frequent early in
// learning and the table's eight loads are the latency of lazy_weight)
__device__ __forceinline__ void jump_table_to_lds(uint64_t *sh_jump)
{
  const uint64_t *src = reinterpret_cast<const uint64_t *>(&d_jump);
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) sh_jump[i] = src[i];
  __syncthreads();
}

__device__ __forceinline__ uint64_t lcg_jump_lds(const uint64_t *sh_jump, uint64_t x, uint64_t n)
{
  uint64_t a[4], c[4];
#pragma unroll
  for (int w = 0; w < 4; ++w)
  {
    const uint32_t b = (uint32_t)(n >> (8 * w)) & 0xFFu;
    a[w] = sh_jump[w * 256 + b];
    c[w] = sh_jump[1024 + w * 256 + b];
  }
#pragma unroll
  for (int w = 0; w < 4; ++w) x = (a[w] * x + c[w]) & kMask48;
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

// initial value of a slot: the loaded image when there is one, else the reference's draw
__device__ inline double initial_weight(const ReplicaState &rs, int table, const LinearParams &lp, uint32_t slot)
{
  const double *img = rs.lazy_base[table];
  return img ? img[slot] : lazy_weight(rs.TL0, lp, slot);
}

__device__ __forceinline__ double lazy_weight_lds(const uint64_t *sh_jump, uint64_t tl0, const LinearParams &lp, uint32_t slot)
{
  uint64_t x = lcg_next(lcg_jump_lds(sh_jump, tl0, lp.draws_before + (uint64_t)slot));
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

// the two halves of murmur_mix: the key's own scramble (depends on the coordinate only) and
// its absorption into the running hash; murmur_mix(h, c) == murmur_absorb(h, murmur_key(c))
__device__ __forceinline__ uint32_t murmur_key(int c)
{
  const uint32_t m = 0x5bd1e995u;
  uint32_t k = (uint32_t)c;
  k *= m;
  k ^= k >> 24;
  k *= m;
  return k;
}
__device__ __forceinline__ uint32_t murmur_absorb(uint32_t h, uint32_t k) { return (h * 0x5bd1e995u) ^ k; }

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
// One replica's weights: an open-addressing table of 64-byte buckets, 4 entries each,
//   { key[4] (16 B) | aux[4] (16 B) | val[4] (32 B) }   = one fabric request per lookup.
// key word: bits 0..25 reference slot index + 1 (0 = empty), bits 26..30 tiling that
// created the entry, bit 31 "touched by a second tiling" (hash collision across tilings).
// A lookup fetches the whole home bucket at once (3 x 16-byte loads in flight), so it
// completes in ONE memory round trip unless the bucket is full (then: next bucket).
// position = bucket * 4 + way; a position is stable for the life of the table.
struct __attribute__((aligned(64))) Bucket {
  uint32_t key[4];
  uint32_t aux[4];
  double   val[4];
};
static_assert(sizeof(Bucket) == 4 * sizeof(Entry), "a bucket is four 16-byte entries");

constexpr uint32_t kKeyMask = 0x03FFFFFFu, kOwnerShift = 26, kSharedBit = 0x80000000u;

struct Table {
  Bucket  *base;
  uint32_t bmask, shift;
};

__device__ __forceinline__ Table table_of(const DevParams &P, int table, int replica)
{
  Table t;
  Entry *e = P.tables + (((size_t)table * (size_t)P.n_replicas + (size_t)replica) << P.logC);
  t.base = reinterpret_cast<Bucket *>(e);
  t.bmask = (1u << (P.logC - 2)) - 1u;
  t.shift = 32u - (P.logC - 2);
  return t;
}

__device__ __forceinline__ uint32_t table_home(const Table &t, uint32_t slot)
{
  return ((slot + 1u) * 0x9E3779B1u) >> t.shift;
}

struct BucketRegs { uint4 k; double v[4]; };

__device__ __forceinline__ BucketRegs bucket_load(const Table &t, uint32_t b)
{
  BucketRegs r;
  const Bucket *bp = &t.base[b];
  r.k = *reinterpret_cast<const uint4 *>(bp->key);
  const double2 v01 = *reinterpret_cast<const double2 *>(&bp->val[0]);
  const double2 v23 = *reinterpret_cast<const double2 *>(&bp->val[2]);
  r.v[0] = v01.x; r.v[1] = v01.y; r.v[2] = v23.x; r.v[3] = v23.y;
  return r;
}

__device__ __forceinline__ uint4 bucket_keys(const Table &t, uint32_t b)
{
  return *reinterpret_cast<const uint4 *>(t.base[b].key);
}

__device__ __forceinline__ void value_store(const Table &t, uint32_t pos, double v) { t.base[pos >> 2].val[pos & 3u] = v; }
__device__ __forceinline__ double value_load(const Table &t, uint32_t pos) { return t.base[pos >> 2].val[pos & 3u]; }

__device__ __forceinline__ void entry_create(const Table &t, uint32_t pos, uint32_t slot, uint32_t owner, double v)
{
  Bucket *bp = &t.base[pos >> 2];
  bp->key[pos & 3u] = (slot + 1u) | (owner << kOwnerShift);
  bp->val[pos & 3u] = v;
}

// branch hint: the rare side is laid out of line, the common path falls through (a lone wave
// has nothing to hide the fetch bubble of a taken branch behind)
__device__ __forceinline__ bool rarely(bool c) { return __builtin_expect(c, false); }

// order LDS / global accesses of the lanes of one wave (no instruction beyond waits)
__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// way of `slot` in a loaded bucket (0..3) or -1; *empty = bit mask of empty ways
__device__ __forceinline__ int bucket_find(const uint4 &k, uint32_t slot, uint32_t &empty)
{
  const uint32_t want = slot + 1u;
  const uint32_t k0 = k.x & kKeyMask, k1 = k.y & kKeyMask, k2 = k.z & kKeyMask, k3 = k.w & kKeyMask;
  empty = (k0 == 0u ? 1u : 0u) | (k1 == 0u ? 2u : 0u) | (k2 == 0u ? 4u : 0u) | (k3 == 0u ? 8u : 0u);
  int way = -1;
  way = (k3 == want) ? 3 : way;
  way = (k2 == want) ? 2 : way;
  way = (k1 == want) ? 1 : way;
  way = (k0 == want) ? 0 : way;
  return way;
}

// State of one lookup.  hit: pos/val valid.  miss: `bucket` is the first bucket of the
// probe sequence with an empty way and `empty` its empty-way mask (as loaded).
struct Lookup {
  uint32_t bucket, empty, pos;
  uint32_t kw;                  // key word of the entry found (owner tiling, shared bit); 0 when created by this lane
  bool     miss;
};

__device__ __forceinline__ uint32_t bucket_kw(const uint4 &k, int way)
{
  return (way == 0) ? k.x : (way == 1) ? k.y : (way == 2) ? k.z : k.w;
}

// NP independent lookups of one lane, in two halves so that a caller can put independent work
// between the loads and their first use: table_issue starts the home-bucket loads (all in
// flight together), table_resolve consumes them.
template <int NP>
__device__ __forceinline__ void table_issue(const Table &t, const uint32_t (&slot)[NP], Lookup (&lk)[NP], BucketRegs (&br)[NP])
{
#pragma unroll
  for (int i = 0; i < NP; ++i)
  {
    lk[i].bucket = table_home(t, slot[i]);
    lk[i].pos = 0u;
    br[i] = bucket_load(t, lk[i].bucket);
  }
}

// way of `slot` in a loaded bucket, branch-free: *hit, and for a hit the way (0..3), the key
// word and the value.  A slot occupies at most one way.
__device__ __forceinline__ void bucket_select(const BucketRegs &b, uint32_t slot, bool &hit, uint32_t &way, uint32_t &kw, double &val)
{
  const uint32_t want = slot + 1u;
  const uint32_t k0 = b.k.x & kKeyMask, k1 = b.k.y & kKeyMask, k2 = b.k.z & kKeyMask, k3 = b.k.w & kKeyMask;
  const bool m0 = k0 == want, m1 = k1 == want, m2 = k2 == want, m3 = k3 == want;
  way = m1 ? 1u : 0u;
  kw = m1 ? b.k.y : b.k.x;
  val = m1 ? b.v[1] : b.v[0];
  way = m2 ? 2u : way;
  kw = m2 ? b.k.z : kw;
  val = m2 ? b.v[2] : val;
  way = m3 ? 3u : way;
  kw = m3 ? b.k.w : kw;
  val = m3 ? b.v[3] : val;
  hit = m0 || m1 || m2 || m3;
}

// bit mask of the empty ways of a loaded bucket
__device__ __forceinline__ uint32_t bucket_empty(const uint4 &k)
{
  return ((k.x & kKeyMask) == 0u ? 1u : 0u) | ((k.y & kKeyMask) == 0u ? 2u : 0u) | ((k.z & kKeyMask) == 0u ? 4u : 0u) |
         ((k.w & kKeyMask) == 0u ? 8u : 0u);
}

template <int NP>
__device__ __forceinline__ void table_resolve(const Table &t, const uint32_t (&slot)[NP], Lookup (&lk)[NP], const BucketRegs (&br)[NP],
                                              double (&val)[NP], uint32_t &status)
{
  bool pending[NP];
  bool any = false, anynot = false;
#pragma unroll
  for (int i = 0; i < NP; ++i)
  { // straight-line selects: nothing here is worth a branch
    bool hit;
    uint32_t way, kw;
    double v;
    bucket_select(br[i], slot[i], hit, way, kw, v);
    lk[i].pos = hit ? ((lk[i].bucket << 2) | way) : lk[i].pos;
    lk[i].kw = hit ? kw : 0u;
    val[i] = hit ? v : val[i];
    lk[i].miss = !hit;                                   // refined below
    lk[i].empty = 0u;
    pending[i] = false;
    anynot = anynot || !hit;
  }
  if (rarely(__any(anynot)))
  { // some lane did not find its slot: empty ways decide between "create here" and "walk on"
#pragma unroll
    for (int i = 0; i < NP; ++i)
    {
      const bool nohit = lk[i].miss;
      lk[i].empty = bucket_empty(br[i].k);
      lk[i].miss = nohit && lk[i].empty != 0u;
      pending[i] = nohit && lk[i].empty == 0u;           // home bucket full of other slots: overflow chain
      any = any || pending[i];
    }
  }
  if (rarely(__any(any)))
  { // rare: walk the following buckets
    for (int it = 1; it < kMaxProbe; ++it)
    {
      bool more = false;
#pragma unroll
      for (int i = 0; i < NP; ++i)
        if (pending[i])
        {
          lk[i].bucket = (lk[i].bucket + 1u) & t.bmask;
          const BucketRegs b2 = bucket_load(t, lk[i].bucket);
          const int way = bucket_find(b2.k, slot[i], lk[i].empty);
          if (way >= 0)
          {
            lk[i].pos = (lk[i].bucket << 2) | (uint32_t)way;
            lk[i].kw = bucket_kw(b2.k, way);
            val[i] = (way == 0) ? b2.v[0] : (way == 1) ? b2.v[1] : (way == 2) ? b2.v[2] : b2.v[3];
            pending[i] = false;
          }
          else if (lk[i].empty != 0u) { lk[i].miss = true; pending[i] = false; }
          else more = true;
        }
      if (!__any(more)) break;
    }
#pragma unroll
    for (int i = 0; i < NP; ++i)
      if (pending[i]) status |= ST_TABLE_FULL;
  }
}

template <int NP>
__device__ __forceinline__ void table_lookup(const Table &t, const uint32_t (&slot)[NP], Lookup (&lk)[NP], double (&val)[NP], uint32_t &status)
{
  BucketRegs br[NP];
  table_issue<NP>(t, slot, lk, br);
  table_resolve<NP>(t, slot, lk, br, val, status);
}

// Serialised insert (one lane per 16-lane group at a time), re-reading the bucket: used for
// the lanes the parallel path could not place (conflicts), and by the fine-grained operators.
__device__ __noinline__ void table_insert_serial(const Table &t, bool todo, uint32_t slot, uint32_t owner, double w0,
                                           Lookup &lk, double &val, uint32_t &status, uint32_t &inserted)
{
  const int lane = threadIdx.x & 63;
  unsigned long long pend = __ballot(todo);
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
      uint32_t b = lk.bucket;
      for (int it = 0; it < kMaxProbe; ++it)
      {
        const BucketRegs br = bucket_load(t, b);
        uint32_t empty;
        const int way = bucket_find(br.k, slot, empty);
        if (way >= 0)
        { // a sibling lane created it meanwhile
          lk.pos = (b << 2) | (uint32_t)way;
          lk.kw = bucket_kw(br.k, way);
          val = (way == 0) ? br.v[0] : (way == 1) ? br.v[1] : (way == 2) ? br.v[2] : br.v[3];
          done = true;
          break;
        }
        if (empty != 0u)
        {
          const uint32_t w = (uint32_t)__builtin_ctz(empty);
          lk.pos = (b << 2) | w;
          lk.kw = 0u;
          entry_create(t, lk.pos, slot, owner, w0);
          val = w0;
          inserted++;
          done = true;
          break;
        }
        b = (b + 1u) & t.bmask;
      }
      if (!done) status |= ST_TABLE_FULL;
    }
    pend &= ~sel;
    // The next lane's probe must observe this insert.  Both are vector memory operations of
    // the same wave issued in this order, which the hardware keeps for one address; the
    // fence only stops the compiler from reordering them.
    wave_sync();
  }
}

// single lookup-or-create (fine-grained operators): lane = tiling
__device__ inline void table_probe(const Table &t, const LinearParams &lp, const ReplicaState &rs, int table, bool active,
                                   uint32_t slot, uint32_t &pos, double &val, uint32_t &status, uint32_t &inserted)
{
  uint32_t sl[1] = {slot};
  Lookup lk[1];
  lk[0].bucket = 0; lk[0].empty = 0; lk[0].pos = 0; lk[0].kw = 0; lk[0].miss = false;
  double v[1] = {0};
  if (active) table_lookup<1>(t, sl, lk, v, status);
  const bool miss = active && lk[0].miss;
  double w0 = 0;
  if (__any(miss))
  {
    if (miss) w0 = initial_weight(rs, table, lp, slot);
    table_insert_serial(t, miss, slot, (uint32_t)(threadIdx.x & 31), w0, lk[0], v[0], status, inserted);
  }
  // keep the "touched by a second tiling" bit current (the fused kernel relies on it)
  if (active && lk[0].kw != 0u && ((lk[0].kw >> kOwnerShift) & 31u) != (uint32_t)(threadIdx.x & 31) && !(lk[0].kw & kSharedBit))
    t.base[lk[0].pos >> 2].key[lk[0].pos & 3u] = lk[0].kw | kSharedBit;
  pos = lk[0].pos;
  val = v[0];
}

// ----------------------------------------------------------- environments --
template <int ENV> struct Env;

// dynamics/pendulum + task/pendulum/swingup (pendulum.cpp:40-145)
template <> struct Env<GRLX_ENV_PENDULUM> {
  static constexpr int S = 3, D = 2;
  // pendulum.cpp:40-49, 55-68; the constants are held in registers by the caller (rk4_step)
  struct Consts { SinConsts k; double invJ, mgl, b, kkr, kr; };
  template <bool PIN> __device__ static __forceinline__ Consts consts()
  {
    const double J = 0.000191, m = 0.055, g = 9.81, l = 0.042, b = 0.000003, K = 0.0536, R = 9.5;
    Consts c;
    c.k = sin_consts<PIN>();
    c.invJ = math_const<PIN>(1 / J);
    c.mgl = math_const<PIN>(m * g * l);
    c.b = math_const<PIN>(b);
    c.kkr = math_const<PIN>(K * K / R);
    c.kr = math_const<PIN>(K / R);
    return c;
  }
  __device__ static __forceinline__ void eom(const Consts &c, const double *x, double u, double *xd)
  {
    double a = x[0], ad = x[1];
    double add = c.invJ * (c.mgl * psin(a, c.k) - c.b * ad - c.kkr * ad + c.kr * u);
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
  __device__ static __forceinline__ bool in_domain(const double *x) { return __builtin_fabs(x[0]) < 0x1p19; }
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

// dynamics/acrobot + task/acrobot/balancing (acrobot.cpp:48-151); state = [theta1, theta2,
// thetad1, thetad2, time].  No reference test pins it: parity is against the oracle only.
template <> struct Env<GRLX_ENV_ACROBOT> {
  static constexpr int S = 5, D = 4;
  using Consts = SinConsts;                     // held in registers across the integration loop
  template <bool PIN> __device__ static __forceinline__ Consts consts() { return sin_consts<PIN>(); }
  __device__ static __forceinline__ void eom(const Consts &k, const double *x, double u, double *xd)
  { // acrobot.cpp:48-79, expression for expression
    const double l1 = 1, m1 = 1, m2 = 1, lc1 = 0.5, lc2 = 0.5, I1 = 1, I2 = 1, g = 9.8;
    const double theta1 = x[0], theta2 = x[1], thetad1 = x[2], thetad2 = x[3];
    const double tau = u;
    double sin2, cos2;
    psincos(theta2, k, sin2, cos2);

    double phi2 = m2*lc2*g*pcos(theta1+theta2-GRLX_PI/2, k);
    double phi1 = -m2*l1*lc2*thetad2*thetad2*sin2-2*m2*l1*lc2*thetad2*thetad1*sin2 +
                  (m1*lc1+m2*l1)*g*pcos(theta1-GRLX_PI/2, k)+phi2;
    double d2 = m2*(lc2*lc2+l1*lc2*cos2)+I2;
    double d1 = m1*lc1*lc1 + m2*(l1*l1+lc2*lc2+2*l1*lc2*cos2)+I1+I2;
    double thetadd2 = (tau+d2*phi1/d1-m2*l1*lc2*thetad2*thetad2*sin2-phi2)/
                      (m2*lc2*lc2+I2-d2*d2/d1);
    double thetadd1 = -(d2*thetadd2+phi1)/d1;

    if (thetad1 >  4*GRLX_PI) thetadd1 = fmin(thetadd1, 0.);
    if (thetad1 < -4*GRLX_PI) thetadd1 = fmax(thetadd1, 0.);
    if (thetad2 >  9*GRLX_PI) thetadd2 = fmin(thetadd2, 0.);
    if (thetad2 < -9*GRLX_PI) thetadd2 = fmax(thetadd2, 0.);

    xd[0] = thetad1;
    xd[1] = thetad2;
    xd[2] = thetadd1;
    xd[3] = thetadd2;
    xd[4] = 1;
  }
  __device__ static __forceinline__ bool failed(const double *x)
  { // :147-151
    return __builtin_fabs(x[0]-GRLX_PI) > 12*GRLX_PI/180 || __builtin_fabs(x[1]) > 12*GRLX_PI/180;
  }
  __device__ static __forceinline__ void start(const DevParams &, int, uint64_t &TL, uint64_t &, double *x)
  { // :102-107
    TL = lcg_next(TL);
    const double r1 = lcg_double(TL);
    TL = lcg_next(TL);
    const double r2 = lcg_double(TL);
    x[0] = GRLX_PI+r1*0.01-0.005;
    x[1] = r2*0.01-0.005;
    x[2] = 0; x[3] = 0; x[4] = 0;
  }
  __device__ static __forceinline__ double actuate(double a) { return a; }                  // Task::actuate default (environment.h:94)
  __device__ static __forceinline__ bool in_domain(const double *x) { return __builtin_fabs(x[0]) < 0x1p18 && __builtin_fabs(x[1]) < 0x1p18; }
  __device__ static __forceinline__ int observe(const DevParams &, const double *x, double *obs)
  { // :109-125
#pragma unroll
    for (int i = 0; i < 4; ++i) obs[i] = x[i];
    if (failed(x)) return 2;
    return x[4] > 20 ? 1 : 0;
  }
  __device__ static __forceinline__ double evaluate(const DevParams &, const double *, double, const double *next)
  { // :127-133
    return failed(next) ? 0. : 1.;
  }
};

// dynamics/cart_pole (end_stop = 1) + task/cart_pole/swingup (cart_pole.cpp:41-237);
// state = [x, theta, xd, thetad, time].  parity unpinned by reference tests.
template <> struct Env<GRLX_ENV_CART_POLE> {
  static constexpr int S = 5, D = 4;
  using Consts = SinConsts;                     // held in registers across the integration loop
  template <bool PIN> __device__ static __forceinline__ Consts consts() { return sin_consts<PIN>(); }
  __device__ static __forceinline__ void eom(const Consts &k, const double *x, double u, double *xd)
  { // cart_pole.cpp:58-108.  QUIRK reproduced on purpose: :65 reads dtheta = state[3-2*end_stop_],
    // which for end_stop = 1 is state[1] -- the ANGLE, not its rate.
    const double g = 9.8, mass_cart = 1.0, mass_pole = 0.1, length = 0.5;
    const double total_mass = mass_cart + mass_pole, pole_mass_length = mass_pole * length;
    const double theta = x[1], dtheta = x[3 - 2 * 1];
    double costheta, sintheta;
    psincos(theta, k, sintheta, costheta);
    const double temp = (u + pole_mass_length * dtheta * dtheta * sintheta) / total_mass;
    const double thetaacc = (g * sintheta - costheta * temp) /
                            (length * ((4. / 3.) - mass_pole * costheta * costheta / total_mass));
    const double acc = temp - pole_mass_length * thetaacc * costheta / total_mass;
    xd[0] = x[2];
    xd[1] = x[3];
    xd[2] = acc;
    xd[3] = thetaacc;
    xd[4] = 1;
    if (x[0] > 2.4 && x[2] > 0)
    { // end stops, :93-105
      xd[0] = 0;
      if (acc > 0) xd[2] = 0;
    }
    else if (x[0] < -2.4 && x[2] < 0)
    {
      xd[0] = 0;
      if (acc < 0) xd[2] = 0;
    }
  }
  __device__ static __forceinline__ bool failed(const double *x) { return __builtin_fabs(x[0]) > 2.4; }   // :212-215
  __device__ static __forceinline__ double potential(const double *x)
  { // :232-238
    double a = pfmod(__builtin_fabs(x[1]), GRLX_2PI);
    if (a > GRLX_PI) a -= GRLX_2PI;
    return -2 * (x[0] * x[0]) - 0.1 * (x[2] * x[2]) - (a * a) - 0.1 * (x[3] * x[3]);
  }
  __device__ static __forceinline__ void start(const DevParams &P, int, uint64_t &TL, uint64_t &, double *x)
  { // :155-164
    TL = lcg_next(TL);
    const double r = lcg_double(TL);
    x[0] = 0;
    x[1] = GRLX_PI + P.randomization * ((r * 0.1) - 0.05);
    x[2] = 0; x[3] = 0; x[4] = 0;
  }
  __device__ static __forceinline__ double actuate(double a) { return a; }                  // Task::actuate default (environment.h:94)
  __device__ static __forceinline__ bool in_domain(const double *x) { return __builtin_fabs(x[1]) < 0x1p19; }
  __device__ static __forceinline__ int observe(const DevParams &P, const double *x, double *obs)
  { // :166-190
    double a = pfmod(x[1] + GRLX_PI, GRLX_2PI);
    if (a < 0) a += GRLX_2PI;
    obs[0] = x[0];
    obs[1] = a;
    obs[2] = x[2];
    obs[3] = x[3];
    if (P.end_stop_penalty && failed(x)) return 2;
    return x[4] > P.timeout ? 1 : 0;
  }
  __device__ static __forceinline__ double evaluate(const DevParams &P, const double *, double action, const double *next)
  { // :192-201, shaping = 0
    const double a15 = action / 15;
    return potential(next) - P.action_penalty * (a15 * a15) * 2 - P.end_stop_penalty * (failed(next) ? 1 : 0) * 10000;
  }
};

// model/compass_walker + task/compass_walker/walk: the simplest walking model with its own
// RK4 (velocities and angles staged separately), angle wrapping and heel-strike events located
// by a secant search (SWModel.cpp:15-258, SWModel.h:40-59, compass_walker.cpp:63-94, 251-344).
// state vector (compass_walker.h:40-42).  parity unpinned by reference tests.
template <> struct Env<GRLX_ENV_COMPASS_WALKER> {
  static constexpr int S = 11, D = 5;
  static constexpr bool kCustomModel = true;
  enum { SLA = 0, HA, SLAR, HAR, CHANGED, SFX, LASTHIPX, HIPVEL, STEPDIST, TIME, TIMEOUT };
  struct St { double sla, slar, ha, har, sfx; };

  __device__ static __forceinline__ double hip_x(const St &m) { return m.sfx - psin(m.sla); }
  __device__ static __forceinline__ double swing_y(const St &m) { return pcos(m.sla) - pcos(m.sla - m.ha); }
  // the same with the sine constants held in registers by model_step (20 sub-steps x 10 evaluations)
  __device__ static __forceinline__ double hip_x(const SinConsts &k, const St &m) { return m.sfx - psin_s(m.sla, k); }
  __device__ static __forceinline__ double swing_y(const SinConsts &k, const St &m) { return pcos_s(m.sla, k) - pcos_s(m.sla - m.ha, k); }
  __device__ static __forceinline__ void wrap(St &m)
  { // SWModel.h:48-59
    if (m.sla >= GRLX_PI) m.sla -= 2*GRLX_PI;
    if (m.sla < -GRLX_PI) m.sla += 2*GRLX_PI;
    if (m.ha >= GRLX_PI) m.ha -= 2*GRLX_PI;
    if (m.ha < -GRLX_PI) m.ha += 2*GRLX_PI;
  }
  __device__ static __forceinline__ void accel(const DevParams &P, const SinConsts &k, const St &m, double torque, double &asl, double &ahip)
  { // SWModel.cpp:212-218
    double sn, cs;
    psincos_s(m.sla - P.slope_angle, k, sn, cs);
    asl = sn;
    ahip = psin_s(m.ha, k) * (m.slar*m.slar - cs) + asl;
    ahip += torque;
  }
  __device__ static __forceinline__ void rk4(const DevParams &P, const SinConsts &k, St &state, double torque, double dt)
  { // SWModel.cpp:220-258
    St s1 = state, s2 = state, s3 = state, s4 = state;
    double k1s, k1h, k2s, k2h, k3s, k3h, k4s, k4h;
    accel(P, k, s1, torque, k1s, k1h);
    s2.slar = s1.slar + (dt/2)*k1s;
    s2.har  = s1.har  + (dt/2)*k1h;
    s2.sla  = s1.sla  + (dt/2)*s1.slar;
    s2.ha   = s1.ha   + (dt/2)*s1.har;
    accel(P, k, s2, torque, k2s, k2h);
    s3.slar = s1.slar + (dt/2)*k2s;
    s3.har  = s1.har  + (dt/2)*k2h;
    s3.sla  = s1.sla  + (dt/2)*s2.slar;
    s3.ha   = s1.ha   + (dt/2)*s2.har;
    accel(P, k, s3, torque, k3s, k3h);
    s4.slar = s1.slar + (dt)*k3s;
    s4.har  = s1.har  + (dt)*k3h;
    s4.sla  = s1.sla  + (dt)*s3.slar;
    s4.ha   = s1.ha   + (dt)*s3.har;
    accel(P, k, s4, torque, k4s, k4h);
    state.slar = s1.slar + (dt/6)*(k1s + 2*k2s + 2*k3s + k4s);
    state.har  = s1.har  + (dt/6)*(k1h + 2*k2h + 2*k3h + k4h);
    state.sla  = s1.sla  + (dt/6)*(s1.slar + 2*s2.slar + 2*s3.slar + s4.slar);
    state.ha   = s1.ha   + (dt/6)*(s1.har + 2*s2.har + 2*s3.har + s4.har);
  }
  __device__ static __forceinline__ double heelstrike_moment(const DevParams &P, const SinConsts &k, const St &t0, const St &t1, St &hs, double torque, double precision, double dt)
  { // SWModel.cpp:53-104
    double timeLeft = 0;
    St s0 = t0, s1 = t1;
    double s0time = 0, s1time = dt;
    const int maxIterations = 10;
    int iIter;
    for (iIter = 0; iIter < maxIterations; iIter++)
    {
      hs = s0;
      const double y0 = swing_y(k, s0);
      double newDt = (s1time - s0time) * y0 / (y0 - swing_y(k, s1));
      rk4(P, k, hs, torque, newDt);
      if (swing_y(k, hs) > 0)
      {
        s0 = hs;
        s0time = s0time + newDt;
      }
      else
      {
        s1 = hs;
        s1time = s0time + newDt;
      }
      if (swing_y(k, s0) < precision)
      {
        hs = s0;
        timeLeft = dt - s0time;
        break;
      }
      else if (-swing_y(k, s1) < precision)
      {
        hs = s1;
        timeLeft = dt - s1time;
        break;
      }
    }
    if (iIter >= maxIterations)
    {
      if (swing_y(k, hs) > 0) timeLeft = dt - s0time;
      else timeLeft = dt - s1time;
    }
    return timeLeft;
  }
  __device__ static __forceinline__ void model_step(const DevParams &P, const double *x, double torque, double *next)
  { // CompassWalkerModel::step (compass_walker.cpp:63-94) around CSWModel::singleStep (SWModel.cpp:142-210)
    St st, prev, hs;
    st.sfx = x[SFX]; st.sla = x[SLA]; st.slar = x[SLAR]; st.ha = x[HA]; st.har = x[HAR];
    prev = st;
    hs = st;
    bool changed = false;
    const double partial = P.walker_dt;
    const SinConsts k = sin_consts<true>();
    double y_prev = swing_y(k, prev);            // swing_y(prev) of the next sub-step is this sub-step's swing_y(st)
    for (int i = 0; i < P.integration_steps; i++)
    {
      rk4(P, k, st, torque, partial);
      wrap(st);
      // detectEvents (SWModel.cpp:30-45)
      double timeleft = 0;
      bool struck = false;
      const double y_now = swing_y(k, st);
      if ((y_prev >= 0) && (y_now < 0))
        if (((prev.ha < 0) && (st.ha < 0)) || ((prev.ha > 0) && (st.ha > 0)))
          if ((st.slar < 0) && (st.ha < 0))
          { // processStanceLegChange (:106-124)
            struck = true;
            timeleft = heelstrike_moment(P, k, prev, st, hs, torque, 1.0E-11, partial);
            const double c2 = pcos(2.0*hs.sla, k);
            st.har  = hs.slar*(c2*(1.0 - c2));
            st.slar = hs.slar*(c2);
            st.sfx  = hip_x(k, hs) + psin(hs.sla - hs.ha, k);
            st.sla  = -hs.sla;
            st.ha   = -2.0*hs.sla;
          }
      changed = changed || (timeleft > 0);
      if (timeleft > 0)
      {
        rk4(P, k, st, torque, timeleft);
        wrap(st);
      }
      // a pure function of the state: recomputed only where a heel strike replaced the state
      y_prev = struck ? swing_y(k, st) : y_now;
      prev = st;
    }
#pragma unroll
    for (int i = 0; i < S; ++i) next[i] = x[i];
    next[SLA] = st.sla;
    next[HA] = st.ha;
    next[SLAR] = st.slar;
    next[HAR] = st.har;
    next[SFX] = st.sfx;
    next[CHANGED] = changed ? 1. : 0.;
    next[LASTHIPX] = changed ? hip_x(k, st) : x[LASTHIPX];
    next[HIPVEL] = - st.slar * pcos(st.sla, k);
    next[TIME] = x[TIME] + P.control_step;
    next[TIMEOUT] = x[TIMEOUT];
  }
  __device__ static __forceinline__ void eom(const double *, double, double *) {}
  __device__ static __forceinline__ void start(const DevParams &P, int test, uint64_t &, uint64_t &G, double *x)
  { // compass_walker.cpp:251-290: rejection sampling on the GLOBAL drand48 stream
    const double i_sla = 0.1534, i_slar = -0.1561, i_ha = 2.0*0.1534, i_har = -0.0073;
    const double variation = (!test) ? P.initial_state_variation : 0;
    const double cslope = pcos(P.slope_angle);
    St sw;
    sw.sfx = 0;
    for (int guard = 0; guard < 100000; ++guard)
    {
      G = lcg_next(G); sw.sla  = i_sla  * (1.0 - variation + 2.0*variation*lcg_double(G));
      G = lcg_next(G); sw.ha   = i_ha   * (1.0 - variation + 2.0*variation*lcg_double(G));
      G = lcg_next(G); sw.slar = i_slar * (1.0 - variation + 2.0*variation*lcg_double(G));
      G = lcg_next(G); sw.har  = i_har  * (1.0 - variation + 2.0*variation*lcg_double(G));
      if (!(sw.slar*sw.slar/2.0 + pcos(sw.sla)*cslope < cslope)) break;
    }
#pragma unroll
    for (int i = 0; i < S; ++i) x[i] = 0;
    x[SLA] = sw.sla;
    x[HA] = sw.ha;
    x[SLAR] = sw.slar;
    x[HAR] = sw.har;
    x[SFX] = sw.sfx;
    x[LASTHIPX] = hip_x(sw);
    x[HIPVEL] = -sw.slar * pcos(sw.sla);
    x[TIMEOUT] = test ? 2*P.timeout : P.timeout;
  }
  __device__ static __forceinline__ double actuate(double a) { return a; }
  __device__ static __forceinline__ bool in_domain(const double *x) { return __builtin_fabs(x[SLA]) < 8. && __builtin_fabs(x[HA]) < 8. && __builtin_fabs(x[SLAR]) < 1e6; }
  __device__ static __forceinline__ bool fallen(const double *x)
  {
    return __builtin_fabs(x[SLA]) > GRLX_PI/8 || __builtin_fabs(x[HA] - 2 * x[SLA]) > GRLX_PI/4;
  }
  __device__ static __forceinline__ int observe(const DevParams &, const double *x, double *obs)
  { // :292-329, observe = [1,1,1,1,1,0,0], steps = 0
    obs[0] = x[SLA];
    obs[1] = x[HA] - 2 * x[SLA];
    obs[2] = x[SLAR];
    obs[3] = x[HAR] - 2 * x[SLAR];
    obs[4] = x[CHANGED] > 0.5 ? 1. : 0.;
    if (fallen(x)) return 2;
    if (x[TIME] > x[TIMEOUT]) return 1;
    return 0;
  }
  __device__ static __forceinline__ double evaluate(const DevParams &P, const double *, double, const double *next)
  { // :331-344
    double reward = -1;
    if (next[CHANGED] > 0.5) reward = fmin(50 * 4 * psin(next[SLA]), 30.);
    if (fallen(next))
      if (P.negative_reward != 0) reward = P.negative_reward;
    return reward;
  }
};

template <int ENV> struct HasCustomModel { static constexpr bool value = false; };
template <> struct HasCustomModel<GRLX_ENV_COMPASS_WALKER> { static constexpr bool value = true; };

// DynamicalModel::step (modeled.cpp:254-276): classical RK4 sub-steps.
// The last state component is time (xd = 1 in every supported dynamics, and no eom reads
// it), so its stage values are the constant h and its update the constant
// (h + 2h + 2h + h)/6 -- the same operations the reference performs, hoisted.
template <int ENV, bool PIN>
__device__ __forceinline__ void rk4_step(const DevParams &P, const double *x, double u, double *next)
{
  constexpr int S = Env<ENV>::S, SD = S - 1;
  const double h = P.h;
  const double tinc = (((h + 2 * h) + 2 * h) + h) / 6;
  double xd[S], k1[SD], k2[SD], k3[SD], k4[SD], t[S];
#pragma unroll
  for (int i = 0; i < S; ++i) { next[i] = x[i]; t[i] = x[i]; }
  const typename Env<ENV>::Consts ec = Env<ENV>::template consts<PIN>();   // PIN: constants held in vector registers
  for (int ii = 0; ii < P.integration_steps; ++ii)
  {
    Env<ENV>::eom(ec, next, u, xd);
#pragma unroll
    for (int i = 0; i < SD; ++i) { k1[i] = h * xd[i]; t[i] = next[i] + k1[i] / 2; }
    Env<ENV>::eom(ec, t, u, xd);
#pragma unroll
    for (int i = 0; i < SD; ++i) { k2[i] = h * xd[i]; t[i] = next[i] + k2[i] / 2; }
    Env<ENV>::eom(ec, t, u, xd);
#pragma unroll
    for (int i = 0; i < SD; ++i) { k3[i] = h * xd[i]; t[i] = next[i] + k3[i]; }
    Env<ENV>::eom(ec, t, u, xd);
#pragma unroll
    for (int i = 0; i < SD; ++i)
    {
      k4[i] = h * xd[i];
      next[i] = next[i] + div6(k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
    }
    next[SD] = next[SD] + tinc;
  }
}

// ModeledEnvironment::step (modeled.cpp:160-213), window 1, no delta, discrete_time 1
// PIN: hold the dynamics' constants in vector registers across the integration loop (pays at one
// wave per SIMD, costs registers)
template <int ENV, bool PIN = true>
__device__ __forceinline__ void env_step(const DevParams &P, double *x, double action, double *obs, double &reward, int &terminal, uint32_t &status)
{
  constexpr int S = Env<ENV>::S;
  double next[S];
  if constexpr (HasCustomModel<ENV>::value)
    Env<ENV>::model_step(P, x, Env<ENV>::actuate(action), next);     // model/compass_walker integrates itself
  else
    rk4_step<ENV, PIN>(P, x, Env<ENV>::actuate(action), next);
  terminal = Env<ENV>::observe(P, next, obs);
  reward = Env<ENV>::evaluate(P, x, action, next);
  // the branch-free sin/cos need |angle| < 2^20; 2^19 at step ends leaves room for the stages
  if (!Env<ENV>::in_domain(next)) status |= ST_DOMAIN;
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

__device__ __forceinline__ double   in_reg(double v)   { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ uint32_t in_reg(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ bool     in_reg(bool v)     { uint32_t t = v ? 1u : 0u; asm volatile("" : "+v"(t)); return t != 0u; }

// arr[idx] for a register array: every candidate is pinned in a register first, otherwise the
// compiler rewrites the select chain as a dynamically indexed load from a stack copy (scratch memory)
template <typename Tv, int NA>
__device__ __forceinline__ Tv pick(const Tv (&arr)[NA], int idx)
{
  Tv v = in_reg(arr[0]);
#pragma unroll
  for (int a = 1; a < NA; ++a)
  {
    const Tv c = in_reg(arr[a]);
    v = (a == idx) ? c : v;
  }
  return v;
}

__device__ __forceinline__ double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// ------------------------------------------------------- fused rollout -----
// LDS tile shared by the 4 replicas of the wave; index (row*16 + tiling)*4 + group
// makes both the per-lane writes and the 4 simultaneous broadcast reads conflict-free.
#define SHW(row, k, g) sh_w[(((row) * 16 + (k)) << 2) + (g)]

// ------------------------------------------------------- register trace ----
// Replacing eligibility trace (trace.h:208-235) of one tiling, newest first, kept in
// registers.  val is the AUTHORITATIVE weight of the slot while it is in the trace: it is
// written back to the table only when the slot leaves the trace (write-back), unless the
// slot is shared with another tiling (bit e of wt): then every update is also stored
// (write-through) so that the other lane's loads see it.
struct TraceRegs {
  uint32_t pos[kMaxTrace];
  double   val[kMaxTrace];
  uint32_t cnt2;                // occurrences of the slot in its projection minus 1, two bits per entry
  uint32_t wt;
  bool     dup;                 // some entry occurs twice in its projection (sticky until cleared)
  int      len;
  double   total;
};

__device__ __forceinline__ uint32_t trace_cnt(const TraceRegs &tr, int e) { return ((tr.cnt2 >> (2 * e)) & 3u) + 1u; }

__device__ __forceinline__ void trace_init(TraceRegs &tr)
{
#pragma unroll
  for (int e = 0; e < kMaxTrace; ++e) { tr.pos[e] = kInvalidPos; tr.val[e] = 0; }
  tr.cnt2 = 0;
  tr.wt = 0;
  tr.dup = false;
  tr.len = 0;
  tr.total = 1.;
}

// write every cached weight back; optionally forget the entries (EnumeratedTrace::clear)
__device__ __forceinline__ void trace_flush(TraceRegs &tr, const Table &tab, bool clear)
{
#pragma unroll
  for (int e = 0; e < kMaxTrace; ++e)
  {
    if (tr.pos[e] != kInvalidPos && !((tr.wt >> e) & 1u)) value_store(tab, tr.pos[e], tr.val[e]);
    if (clear) tr.pos[e] = kInvalidPos;
  }
  if (clear)
  {
    tr.wt = 0;
    tr.dup = false;
    tr.len = 0;
    tr.total = 1.;
  }
}

// a slot that is in this lane's trace has its current weight in val, not (yet) in the table
__device__ __forceinline__ double trace_forward(const TraceRegs &tr, uint32_t pos, double w)
{
#pragma unroll
  for (int e = 0; e < kMaxTrace; ++e) w = (tr.pos[e] == pos) ? tr.val[e] : w;
  return w;
}

// slot `mp` has just become shared between tilings: the owner writes its cached weight back
// and keeps the table current from now on
__device__ __forceinline__ void trace_share_event(TraceRegs &tr, const Table &tab, uint32_t mp)
{
#pragma unroll
  for (int e = 0; e < kMaxTrace; ++e)
    if (tr.pos[e] == mp && !((tr.wt >> e) & 1u))
    {
      value_store(tab, mp, tr.val[e]);
      tr.wt |= 1u << e;
    }
}

struct UpdateParams {
  double dW, dT, ee, cut, out_min, out_max;
  bool   limit, use_trace;
};

__device__ __forceinline__ double add_clamped(const UpdateParams &u, double v, double d)
{
  return u.limit ? clampd(v + d, u.out_min, u.out_max) : v + d;
}

// One TD update of a linear representation with a replacing trace, as the reference orders it:
//   write(p, target, alpha)            -> every slot of p gets +dW          (linear.cpp:186-216)
//   update(trace, alpha*delta, e)      -> entry k gets +weight_k*dT*ee      (representation.h:79-83)
//   trace->add(p, e)                   -> ssub, push, pop                   (trace.h:215-234)
// Lane j handles tiling j.  Returns nothing; p's final weight becomes trace entry 0.
// sh_ppos / sh_fb / sh_fbflag: LDS scratch of the wave (see rollout kernels).
// Eviction: a weight that leaves the trace is written back to the table.  With HOLD the first
// write-back of the call is handed to the caller instead ({pos, val} in ev; the caller stores it
// later; pos = kInvalidPos: nothing held); ev.n counts the write-backs of the call (n > 1, or a
// path that does not count: n = 2, tells the caller that table values it loaded before this call
// may be stale).
struct Evicted { uint32_t n, pos; double val; };

template <bool HOLD>
__device__ __forceinline__ void td_update_lane(TraceRegs &tr, const Table &tab, const UpdateParams &u, uint32_t p_pos, bool p_sh, double wp,
                                               int g, int j, const uint32_t *sh_ppos, double *sh_fb, uint32_t *sh_fbflag, uint32_t &status,
                                               Evicted &ev)
{
  // Aliasing between p and the trace (IndexProjection::ssub, projection.h:94-104).  Inside a
  // lane it is a register compare.  Across lanes it needs a p that is a slot shared between
  // tilings (only such a slot can sit in another lane's trace, or twice in p): those lanes'
  // positions are compared through LDS -- usually none.
  const uint32_t shmask = (uint32_t)((__ballot(p_sh) >> (16 * g)) & 0xFFFFull);
  uint32_t cp = 1;                                     // occurrences of my slot inside p
  double v;                                            // final weight of p's slot after this step
  bool cross = tr.dup;                                 // does this lane see an alias that crosses lanes?
  if (rarely(__any(shmask != 0u)))
  { // some lane's p is a shared slot: compare those few positions (usually one) with my trace and my p
    for (uint32_t mm = shmask; mm != 0u; mm &= mm - 1u)
    {
      const int k = __builtin_ctz(mm);
      const uint32_t ppk = sh_ppos[g * 16 + k];
      if (k != j)
      {
        if (ppk == p_pos) cross = true;
#pragma unroll
        for (int e = 0; e < kMaxTrace; ++e)
          if (e < tr.len && tr.pos[e] == ppk) cross = true;
      }
    }
  }
  if (!rarely(__any(cross)))
  { // ---- common case: no alias crosses lanes in this wave; aliasing is a register compare inside
    // the lane; straight-line code, no exec-mask branches
    // Entries at e >= len are always invalid (pos == kInvalidPos), so validity alone decides; the
    // weight sequence 1, ee, ee^2, ... does not depend on the data (a compile-time table in a
    // specialised build).
    double a_val = 0, a_de = 0;
    bool a_upd = false;
    uint32_t doitmask = 0, ownmask = 0;
    if (u.use_trace)
    {
      double weight = 1.;
      bool upd = true;
#pragma unroll
      for (int e = 0; e < kMaxTrace; ++e)
      {
        upd = upd && (weight > 0.001);                   // representation.h:81
        const double de = weight * u.dT * u.ee;
        const bool own = tr.pos[e] == p_pos;             // p_pos is a valid position
        const bool doit = tr.pos[e] != kInvalidPos && !own && upd;
        const double vv = add_clamped(u, tr.val[e], de);
        tr.val[e] = doit ? vv : tr.val[e];
        doitmask |= doit ? (1u << e) : 0u;
        ownmask |= own ? (1u << e) : 0u;
        a_val = own ? tr.val[e] : a_val;
        a_de = own ? de : a_de;
        a_upd = own ? upd : a_upd;
        tr.pos[e] = own ? kInvalidPos : tr.pos[e];       // ssub: the slot leaves the trace
        weight *= u.ee;
      }
    }
    const bool aliased = ownmask != 0u;
    const uint32_t stmask = doitmask & tr.wt;            // write-through entries that changed
    tr.wt &= ~ownmask;
    // p's write first, then the aliased entry's update (if it is still being updated)
    const double base = aliased ? a_val : wp;
    const double v1 = add_clamped(u, base, u.dW);
    const double v2 = add_clamped(u, v1, a_de);
    v = (aliased && a_upd) ? v2 : v1;
    if (rarely(__any(stmask != 0u)))
    {
#pragma unroll
      for (int e = 0; e < kMaxTrace; ++e)
        if ((stmask >> e) & 1u) value_store(tab, tr.pos[e], tr.val[e]);   // shared slot: keep the table current
    }
  }
  else
  { // ---- general case: some alias crosses lanes (a slot shared between tilings is involved)
    uint32_t xm[kMaxTrace];                              // lanes k != j whose p equals my trace slot e
#pragma unroll
    for (int e = 0; e < kMaxTrace; ++e) xm[e] = 0u;
    for (uint32_t mm = shmask; mm != 0u; mm &= mm - 1u)
    {
      const int k = __builtin_ctz(mm);
      const uint32_t ppk = sh_ppos[g * 16 + k];
      if (k != j)
      {
        if (ppk == p_pos) cp++;
#pragma unroll
        for (int e = 0; e < kMaxTrace; ++e) xm[e] |= (e < tr.len && tr.pos[e] == ppk) ? (1u << k) : 0u;
      }
    }
    double v_alias = 0;
    bool aliased = false;
    if (u.use_trace)
    { // trace entries, newest first (representation.h:79-83, trace.h:150-178)
      double weight = 1.;
      bool upd = true;
#pragma unroll
      for (int e = 0; e < kMaxTrace; ++e)
        if (e < tr.len)
        {
          upd = upd && (weight > 0.001);
          const double de = weight * u.dT * u.ee;
          if (tr.pos[e] != kInvalidPos)
          {
            const bool own = tr.pos[e] == p_pos;
            if (!own && xm[e] == 0u)
            {
              if (upd)
              { // LinearRepresentation::update (linear.cpp:198-216); a slot that occurs twice in
                // its projection is updated twice
                double vv = add_clamped(u, tr.val[e], de);
                if ((tr.wt >> e) & 1u)
                {
                  for (uint32_t c = 1; c < trace_cnt(tr, e); ++c) vv = add_clamped(u, vv, de);
                  value_store(tab, tr.pos[e], vv);
                }
                tr.val[e] = vv;
              }
            }
            else
            { // the slot is also written through p: p's write comes first, then this entry's
              // update; the slot leaves the trace
              if (upd)
              {
                double vv = tr.val[e];
                const uint32_t cpx = (own ? 1u : 0u) + (uint32_t)__builtin_popcount(xm[e]);
                for (uint32_t c = 0; c < cpx; ++c) vv = add_clamped(u, vv, u.dW);
                for (uint32_t c = 0; c < trace_cnt(tr, e); ++c) vv = add_clamped(u, vv, de);
                if (own) { v_alias = vv; aliased = true; }
                for (uint32_t mm = xm[e]; mm != 0u; mm &= mm - 1u)
                {
                  const int k = __builtin_ctz(mm);
                  sh_fb[k * 4 + g] = vv;
                  sh_fbflag[k * 4 + g] = 1u;
                }
              }
              tr.pos[e] = kInvalidPos;
              tr.wt &= ~(1u << e);
            }
          }
          weight *= u.ee;
        }
    }
    wave_sync();
    if (aliased)
      v = v_alias;
    else if (shmask != 0u && sh_fbflag[j * 4 + g] != 0u)
      v = sh_fb[j * 4 + g];
    else
    {
      v = wp;
      for (uint32_t c = 0; c < cp; ++c) v = add_clamped(u, v, u.dW);
    }
    if (HOLD) ev.n = 2u;                                   // weights moved between lanes: not tracked
  }
  // a shared slot is kept current in the table; an exclusive one only if no trace follows.  (Rare per-lane
  // blocks sit behind a wave-uniform test: skipping an exec-masked block is a TAKEN branch, ~30 cycles for a
  // lone wave; a not-taken scalar branch is one issue slot.)
  if (rarely(__any(p_sh || !u.use_trace)))
    if (p_sh || !u.use_trace) value_store(tab, p_pos, v);

  // trace_->add(p, decay) (trace.h:215-234)
  if (u.use_trace)
  {
    if (u.ee < u.cut)
    { // decay below the cut: clear() first
      trace_flush(tr, tab, true);
      if (HOLD) ev.n = 2u;
    }
    if (tr.len >= kMaxTrace) status |= ST_TRACE_OVERFLOW;  // cannot happen: validated at create
#pragma unroll
    for (int e = kMaxTrace - 1; e > 0; --e)
    {
      tr.pos[e] = tr.pos[e - 1];
      tr.val[e] = tr.val[e - 1];
    }
    tr.wt = (tr.wt << 1) & ((1u << kMaxTrace) - 1u);
    tr.cnt2 = (tr.cnt2 << 2) & ((1u << (2 * kMaxTrace)) - 1u);
    tr.pos[0] = p_pos;
    tr.val[0] = v;
    if (cp > 4u) status |= ST_TRACE_OVERFLOW;              // more than four tilings on one slot: not representable
    tr.cnt2 |= (cp - 1u) & 3u;
    tr.dup = tr.dup || cp > 1u;
    if (p_sh) tr.wt |= 1u;
    tr.len = (tr.len < kMaxTrace) ? tr.len + 1 : kMaxTrace;
    tr.total *= u.ee;
    { // pop while the total decay is below the cut (trace.h:227-231): the first pop as selects, more in a rare loop
      const bool pop = tr.total < u.cut && tr.len > 1;
      const double undone = tr.total / u.ee;
      tr.total = pop ? undone : tr.total;
      tr.len = pop ? tr.len - 1 : tr.len;
      if (rarely(__any(tr.total < u.cut && tr.len > 1)))
        while (tr.total < u.cut && tr.len > 1)
        {
          tr.total /= u.ee;
          tr.len--;
        }
    }
    // entries popped off the front of the reference's deque: write their weights back.  In the steady state
    // of a full trace exactly the entry that was shifted into the last register falls off.
    if (!rarely(__any(tr.len != kMaxTrace - 1)))
    {
      constexpr int e = kMaxTrace - 1;
      const bool wb = tr.pos[e] != kInvalidPos && !((tr.wt >> e) & 1u);
      if (HOLD)
      {
        const bool hold = wb && ev.n == 0u;
        if (rarely(__any(wb && !hold)))
          if (wb && !hold) value_store(tab, tr.pos[e], tr.val[e]);
        ev.pos = hold ? tr.pos[e] : ev.pos;
        ev.val = hold ? tr.val[e] : ev.val;
        ev.n += wb ? 1u : 0u;
      }
      else if (wb)
        value_store(tab, tr.pos[e], tr.val[e]);
      tr.pos[e] = kInvalidPos;
      tr.wt &= ~(1u << e);
    }
    else
#pragma unroll
    for (int e = 0; e < kMaxTrace; ++e)
      if (e >= tr.len)
      {
        const bool wb = tr.pos[e] != kInvalidPos && !((tr.wt >> e) & 1u);
        if (HOLD)
        {
          const bool hold = wb && ev.n == 0u;
          if (wb && !hold) value_store(tab, tr.pos[e], tr.val[e]);
          ev.pos = hold ? tr.pos[e] : ev.pos;
          ev.val = hold ? tr.val[e] : ev.val;
          ev.n += wb ? 1u : 0u;
        }
        else if (wb)
          value_store(tab, tr.pos[e], tr.val[e]);
        tr.pos[e] = kInvalidPos;
        tr.wt &= ~(1u << e);
      }
  }
}

// Lookup-or-create of NP slots of one lane in one table, all first-round loads in flight
// together; creates missing slots (parallel LDS-ranked claims, serialised fallback) and
// resolves new cross-tiling sharing events.  sh[i]: the slot is shared between tilings.
// on_share(mp): called in every lane of the group for each slot position that just became shared.
// table_get_finish: the part after the loads of table_issue (lk, br).
template <int NP, typename OnShare>
__device__ __forceinline__ void table_get_finish(const Table &tab, const LinearParams &lp, const ReplicaState &rs, int table, const uint32_t (&slot)[NP],
                                                 Lookup (&lk)[NP], const BucketRegs (&br)[NP],
                                                 uint32_t (&pos)[NP], double (&w)[NP], bool (&sh)[NP], int g, int j, unsigned long long gmask,
                                                 uint32_t *sh_mb, uint32_t *sh_ms, uint32_t *sh_mail, const uint64_t *sh_jump,
                                                 uint32_t &status, uint32_t &inserted, OnShare on_share)
{
  const int lane = threadIdx.x & 63;
  table_resolve<NP>(tab, slot, lk, br, w, status);
  bool anymiss = false;
#pragma unroll
  for (int a = 0; a < NP; ++a) anymiss = anymiss || lk[a].miss;
  if (rarely(__any(anymiss)))
  { // Create the missing slots.  All lookups of this call are complete, so every lane that
    // misses into bucket B saw the same empty ways of B.  Claims are ranked in the fixed order
    // (index, tiling) through LDS: the r-th claimant of a bucket takes its r-th empty way -- no
    // reload, all lanes in parallel.  Equal slots claimed twice (a hash collision inside one
    // state) or a bucket with too few empty ways fall back to the serialised path.
    double w0[NP];
    bool slow[NP];
    uint32_t claims[NP];                                  // lanes of my group that claim a bucket, per index
#pragma unroll
    for (int a = 0; a < NP; ++a)
    {
      sh_mb[g * (NP * 16) + a * 16 + j] = lk[a].miss ? lk[a].bucket : 0xFFFFFFFFu;
      sh_ms[g * (NP * 16) + a * 16 + j] = slot[a];
      claims[a] = (uint32_t)((__ballot(lk[a].miss) >> (16 * g)) & 0xFFFFull);
      w0[a] = 0;
      slow[a] = false;
      if (lk[a].miss)
      { // a loaded policy image replaces the drawn initial value (read here, on the rare path, so
        // that the hot path carries no pointer for it)
        const double *img = rs.lazy_base[table];
        w0[a] = img ? img[slot[a]] : lazy_weight_lds(sh_jump, rs.TL0, lp, slot[a]);
      }
    }
    wave_sync();
#pragma unroll
    for (int a = 0; a < NP; ++a)
      if (lk[a].miss)
      {
        const int me = a * 16 + j;
        uint32_t rank = 0;
        bool dup = false;
#pragma unroll
        for (int a2 = 0; a2 < NP; ++a2)
          for (uint32_t mm = claims[a2]; mm != 0u; mm &= mm - 1u)
          { // only the (index, tiling) pairs that actually claim something
            const int k = a2 * 16 + __builtin_ctz(mm);
            const uint32_t ob = sh_mb[g * (NP * 16) + k], os = sh_ms[g * (NP * 16) + k];
            if (ob == lk[a].bucket && k != me)
            {
              if (os == slot[a]) dup = true;
              else if (k < me) rank++;
            }
          }
        uint32_t e = lk[a].empty;
        for (uint32_t c = 0; c < rank; ++c) e &= e - 1u;      // drop the ways taken by earlier claimants
        if (dup || e == 0u)
          slow[a] = true;
        else
        {
          lk[a].pos = (lk[a].bucket << 2) | (uint32_t)__builtin_ctz(e);
          lk[a].kw = 0u;
          entry_create(tab, lk[a].pos, slot[a], (uint32_t)j, w0[a]);
          w[a] = w0[a];
          inserted++;
        }
      }
    wave_sync();
#pragma unroll
    for (int a = 0; a < NP; ++a)
      if (rarely(__any(slow[a])))
      { // out-of-line and rare: work on copies so that nothing of the hot path has its address taken
        Lookup tmp = lk[a];
        double tv = w[a];
        uint32_t tst = 0, tins = 0;
        table_insert_serial(tab, slow[a], slot[a], (uint32_t)j, w0[a], tmp, tv, tst, tins);
        lk[a] = tmp;
        w[a] = tv;
        status |= tst;
        inserted += tins;
      }
  }

  // ---- slots shared between tilings (a collision of the reference's hash across tilings,
  // ~70 per replica and run).  A slot found with a foreign owner and no shared bit yet is a
  // NEW sharing event: mark it in the table and tell the owner's lane, whose trace may hold
  // the only current copy of the weight.
  bool fresh[NP];
  bool anyfresh = false;
#pragma unroll
  for (int a = 0; a < NP; ++a)
  {
    pos[a] = lk[a].pos;
    const bool found = lk[a].kw != 0u;
    const bool foreign = found && ((lk[a].kw >> kOwnerShift) & 31u) != (uint32_t)j;
    sh[a] = found && (foreign || (lk[a].kw & kSharedBit) != 0u);
    fresh[a] = foreign && (lk[a].kw & kSharedBit) == 0u;
    anyfresh = anyfresh || fresh[a];
  }
  if (rarely(__any(anyfresh)))
  {
#pragma unroll
    for (int a = 0; a < NP; ++a)
    {
      if (fresh[a]) tab.base[pos[a] >> 2].key[pos[a] & 3u] = lk[a].kw | kSharedBit;
      unsigned long long pend = __ballot(fresh[a]);
      while (pend != 0ull)
      { // one event per 16-lane group at a time
        unsigned long long sel = 0ull;
#pragma unroll
        for (int gg = 0; gg < 4; ++gg)
        {
          unsigned long long grp = pend & (0xFFFFull << (16 * gg));
          sel |= grp & (~grp + 1ull);
        }
        const bool mine = ((sel >> lane) & 1ull) != 0ull;
        const bool grp_has = (sel & gmask) != 0ull;
        if (mine) sh_mail[g] = pos[a];
        wave_sync();
        if (grp_has)
        {
          const uint32_t mp = sh_mail[g];
          on_share(mp);                                   // owner side: write back, switch to write-through
#pragma unroll
          for (int b2 = 0; b2 < NP; ++b2)
            if (pos[b2] == mp) sh[b2] = true;
        }
        wave_sync();
        if (mine) w[a] = value_load(tab, pos[a]);         // the value the owner just wrote back
        pend &= ~sel;
      }
    }
  }
}

template <int NP, typename OnShare>
__device__ __forceinline__ void table_get(const Table &tab, const LinearParams &lp, const ReplicaState &rs, int table, const uint32_t (&slot)[NP],
                                          uint32_t (&pos)[NP], double (&w)[NP], bool (&sh)[NP], int g, int j, unsigned long long gmask,
                                          uint32_t *sh_mb, uint32_t *sh_ms, uint32_t *sh_mail, const uint64_t *sh_jump,
                                          uint32_t &status, uint32_t &inserted, OnShare on_share)
{
  Lookup lk[NP];
  BucketRegs br[NP];
  table_issue<NP>(tab, slot, lk, br);
  table_get_finish<NP>(tab, lp, rs, table, slot, lk, br, pos, w, sh, g, j, gmask, sh_mb, sh_ms, sh_mail, sh_jump, status, inserted, on_share);
}

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
