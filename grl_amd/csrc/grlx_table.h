// grlx_table.h -- per-replica sparse weight table: 64-byte buckets of four {key, value} entries, single-round lookups,
// serialised inserts, fine-grained probe.
// Part of the single translation unit grlx_kernels.hip (included there, in order; not self-contained).
#pragma once

namespace grlx {

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

// the values of a bucket only (twin tables: the keys are the other table's)
__device__ __forceinline__ void bucket_load_vals(const Table &t, uint32_t b, BucketRegs &r)
{
  const Bucket *bp = &t.base[b];
  const double2 v01 = *reinterpret_cast<const double2 *>(&bp->val[0]);
  const double2 v23 = *reinterpret_cast<const double2 *>(&bp->val[2]);
  r.v[0] = v01.x; r.v[1] = v01.y; r.v[2] = v23.x; r.v[3] = v23.y;
}

__device__ __forceinline__ uint4 bucket_keys(const Table &t, uint32_t b)
{
  return *reinterpret_cast<const uint4 *>(t.base[b].key);
}

// Positions are masked into the replica's own table: whatever a caller passes (kInvalidPos included -- "nothing
// held", "no previous step"), an access can never leave the allocation.  A wrong position is then a parity
// failure (and ST_BAD_POS where the kernels check it), not a memory fault that takes the process down.
__device__ __forceinline__ void value_store(const Table &t, uint32_t pos, double v) { t.base[(pos >> 2) & t.bmask].val[pos & 3u] = v; }
__device__ __forceinline__ double value_load(const Table &t, uint32_t pos) { return t.base[(pos >> 2) & t.bmask].val[pos & 3u]; }

__device__ __forceinline__ void entry_create(const Table &t, uint32_t pos, uint32_t slot, uint32_t owner, double v)
{
  Bucket *bp = &t.base[(pos >> 2) & t.bmask];
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
// twin (DevParams::twin_tables): the other table of the pair and its parameters -- an entry created in one is created in both
__device__ inline void table_probe(const Table &t, const LinearParams &lp, const ReplicaState &rs, int table, bool active,
                                   uint32_t slot, uint32_t &pos, double &val, uint32_t &status, uint32_t &inserted,
                                   const Table *twin = nullptr, const LinearParams *twin_lp = nullptr, uint32_t *twin_inserted = nullptr)
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
    const uint32_t before = inserted;
    table_insert_serial(t, miss, slot, (uint32_t)(threadIdx.x & 31), w0, lk[0], v[0], status, inserted);
    if (twin && miss && inserted != before)
    {
      entry_create(*twin, lk[0].pos, slot, (uint32_t)(threadIdx.x & 31), initial_weight(rs, 1 - table, *twin_lp, slot));
      (*twin_inserted)++;
    }
    if (twin) wave_sync();
  }
  // keep the "touched by a second tiling" bit current (the fused kernel relies on it)
  if (active && lk[0].kw != 0u && ((lk[0].kw >> kOwnerShift) & 31u) != (uint32_t)(threadIdx.x & 31) && !(lk[0].kw & kSharedBit))
  {
    t.base[(lk[0].pos >> 2) & t.bmask].key[lk[0].pos & 3u] = lk[0].kw | kSharedBit;
    if (twin) twin->base[(lk[0].pos >> 2) & twin->bmask].key[lk[0].pos & 3u] = lk[0].kw | kSharedBit;
  }
  pos = lk[0].pos;
  val = v[0];
}


} // namespace grlx
