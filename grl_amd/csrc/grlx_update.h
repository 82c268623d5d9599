// grlx_update.h -- register eligibility trace, the TD update of one tiling (td_update_lane) and the lookup-or-create front end
// of the table (table_get / table_get_finish).
// Part of the single translation unit grlx_kernels.hip (included there, in order; not self-contained).
#pragma once

namespace grlx {

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

// Twin tables (DevParams::twin_tables): ONE lookup-or-create of `slot` for the actor's table A and the critic's table C, whose keys
// are identical by construction.  A's bucket (brA, loaded by table_issue) is resolved as in table_get_finish<1>; the critic's value is
// taken from its own bucket brC at the same way (want_c: the caller loaded it and wants the value), or loaded from C where the entry
// sits in an overflow bucket.  A missing slot is created in BOTH tables at the same position (each with its own lazy initial weight);
// a new cross-tiling sharing event marks the key word in both and calls on_share once.  insA / insC count the entries created.
template <int LDSJ, typename OnShare>
__device__ __forceinline__ void table_get_finish_twin(const Table &tabA, const Table &tabC, const LinearParams &lpA, const LinearParams &lpC,
                                                      const ReplicaState &rs, const uint32_t (&slot)[1], Lookup (&lk)[1], const BucketRegs (&brA)[1],
                                                      const BucketRegs (&brC)[1], bool want_c, uint32_t (&pos)[1], double (&wA)[1], double &wC, bool (&sh)[1],
                                                      int g, int j, unsigned long long gmask, uint32_t *sh_mb, uint32_t *sh_ms, uint32_t *sh_mail,
                                                      const uint64_t *sh_jump, uint32_t &status, uint32_t &insA, uint32_t &insC, OnShare on_share)
{
  const int lane = threadIdx.x & 63;
  const uint32_t home = lk[0].bucket;
  table_resolve<1>(tabA, slot, lk, brA, wA, status);
  bool created = false;
  double w0C = 0;
  if (rarely(__any(lk[0].miss)))
  { // (the claim protocol of table_get_finish, for one slot per lane; both tables have the same empty ways)
    sh_mb[g * 16 + j] = lk[0].miss ? lk[0].bucket : 0xFFFFFFFFu;
    sh_ms[g * 16 + j] = slot[0];
    const uint32_t claims = (uint32_t)((__ballot(lk[0].miss) >> (16 * g)) & 0xFFFFull);
    double w0A;
    if constexpr (LDSJ == 1) { w0A = lazy_weight_lds(sh_jump, rs.TL0, lpA, slot[0]); w0C = lazy_weight_lds(sh_jump, rs.TL0, lpC, slot[0]); }
    else if constexpr (LDSJ == 2) { w0A = lazy_weight_lds6(sh_jump, rs.TL0, lpA, slot[0]); w0C = lazy_weight_lds6(sh_jump, rs.TL0, lpC, slot[0]); }
    else { w0A = lazy_weight(rs.TL0, lpA, slot[0]); w0C = lazy_weight(rs.TL0, lpC, slot[0]); }
    {
      const double *imgA = rs.lazy_base[1], *imgC = rs.lazy_base[0];
      if (rarely(imgA != nullptr) && lk[0].miss) w0A = imgA[slot[0]];
      if (rarely(imgC != nullptr) && lk[0].miss) w0C = imgC[slot[0]];
    }
    wave_sync();
    uint32_t rank = 0u;
    bool dup = false;
    for (uint32_t mm = claims; mm != 0u; mm &= mm - 1u)
    {
      const int k = __builtin_ctz(mm);
      const uint32_t ob = sh_mb[g * 16 + k], os = sh_ms[g * 16 + k];
      const bool same_bucket = lk[0].miss && ob == lk[0].bucket && k != j;
      dup = dup || (same_bucket && os == slot[0]);
      rank += (same_bucket && os != slot[0] && k < j) ? 1u : 0u;
    }
    bool slow = false;
    if (lk[0].miss)
    {
      uint32_t e = lk[0].empty;
      for (uint32_t c = 0; c < rank; ++c) e &= e - 1u;
      if (dup || e == 0u)
        slow = true;
      else
      {
        lk[0].pos = (lk[0].bucket << 2) | (uint32_t)__builtin_ctz(e);
        lk[0].kw = 0u;
        entry_create(tabA, lk[0].pos, slot[0], (uint32_t)j, w0A);
        entry_create(tabC, lk[0].pos, slot[0], (uint32_t)j, w0C);
        wA[0] = w0A;
        insA++;
        insC++;
        created = true;
      }
    }
    wave_sync();
    if (rarely(__any(slow)))
    { // out of line and rare: the serialised insert works on table A; whoever creates there creates the twin entry too
      Lookup tmp = lk[0];
      double tv = wA[0];
      uint32_t tst = 0, tins = 0;
      table_insert_serial(tabA, slow, slot[0], (uint32_t)j, w0A, tmp, tv, tst, tins);
      if (slow && tins != 0u)
      {
        entry_create(tabC, tmp.pos, slot[0], (uint32_t)j, w0C);
        insC++;
        created = true;
      }
      wave_sync();
      lk[0] = tmp;
      wA[0] = tv;
      status |= tst;
      insA += tins;
    }
  }
  pos[0] = lk[0].pos;
  if (want_c)
  { // the critic's value: its bucket holds the same keys, so the way is the one found in A
    const uint32_t way = pos[0] & 3u;
    const double v = (way == 0u) ? brC[0].v[0] : (way == 1u) ? brC[0].v[1] : (way == 2u) ? brC[0].v[2] : brC[0].v[3];
    const bool at_home = (pos[0] >> 2) == home;
    wC = created ? w0C : v;
    if (rarely(__any(!created && !at_home)))
      if (!created && !at_home) wC = value_load(tabC, pos[0]);       // overflow chain: the entry is not in the bucket that was loaded
  }
  // ---- slots shared between tilings: as in table_get_finish, the mark goes into both key words
  const bool found = lk[0].kw != 0u;
  const bool foreign = found && ((lk[0].kw >> kOwnerShift) & 31u) != (uint32_t)j;
  sh[0] = found && (foreign || (lk[0].kw & kSharedBit) != 0u);
  const bool fresh = foreign && (lk[0].kw & kSharedBit) == 0u;
  if (rarely(__any(fresh)))
  {
    if (fresh)
    {
      tabA.base[(pos[0] >> 2) & tabA.bmask].key[pos[0] & 3u] = lk[0].kw | kSharedBit;
      tabC.base[(pos[0] >> 2) & tabC.bmask].key[pos[0] & 3u] = lk[0].kw | kSharedBit;
    }
    unsigned long long pend = __ballot(fresh);
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
      if (mine) sh_mail[g] = pos[0];
      wave_sync();
      if (grp_has)
      {
        const uint32_t mp = sh_mail[g];
        on_share(mp);                                     // owner side: write back, switch to write-through
        if (pos[0] == mp) sh[0] = true;
      }
      wave_sync();
      if (mine)
      { // the values the owner just wrote back
        wA[0] = value_load(tabA, pos[0]);
        if (want_c) wC = value_load(tabC, pos[0]);
      }
      pend &= ~sel;
    }
  }
}

// Lookup-or-create of NP slots of one lane in one table, all first-round loads in flight
// together; creates missing slots (parallel LDS-ranked claims, serialised fallback) and
// resolves new cross-tiling sharing events.  sh[i]: the slot is shared between tilings.
// on_share(mp): called in every lane of the group for each slot position that just became shared.
// table_get_finish: the part after the loads of table_issue (lk, br).
// LDSJ: where the LCG jump table is read from: 1 = the byte-window table staged in LDS (sh_jump, 16 KB), 0 = device memory
// (sh_jump ignored), 2 = the 6-bit-window table staged in LDS (sh_jump, 6 KB: the wide kernels, whose LDS holds parked state)
template <int NP, int LDSJ = 1, typename OnShare>
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
      slow[a] = false;
    }
    // the initial weights of all NP slots in straight-line code, needed or not: their jump-ahead chains (four dependent table
    // reads each) then overlap instead of running one after the other under NP branches.  A loaded policy image replaces the draw.
#pragma unroll
    for (int a = 0; a < NP; ++a)
    {
      if constexpr (LDSJ == 1)
        w0[a] = lazy_weight_lds(sh_jump, rs.TL0, lp, slot[a]);
      else if constexpr (LDSJ == 2)
        w0[a] = lazy_weight_lds6(sh_jump, rs.TL0, lp, slot[a]);
      else
        w0[a] = lazy_weight(rs.TL0, lp, slot[a]);
    }
    {
      const double *img = rs.lazy_base[table];
      if (rarely(img != nullptr))
      {
#pragma unroll
        for (int a = 0; a < NP; ++a)
          if (lk[a].miss) w0[a] = img[slot[a]];
      }
    }
    wave_sync();
    // one walk over the claims of my group serves all my NP lookups: an earlier claimant of the same bucket with another
    // slot moves me one empty way on, the same slot claimed twice goes to the serialised path
    uint32_t rank[NP];
    bool dup[NP];
#pragma unroll
    for (int a = 0; a < NP; ++a) { rank[a] = 0u; dup[a] = false; }
#pragma unroll
    for (int a2 = 0; a2 < NP; ++a2)
      for (uint32_t mm = claims[a2]; mm != 0u; mm &= mm - 1u)
      { // only the (index, tiling) pairs that actually claim something
        const int k = a2 * 16 + __builtin_ctz(mm);
        const uint32_t ob = sh_mb[g * (NP * 16) + k], os = sh_ms[g * (NP * 16) + k];
#pragma unroll
        for (int a = 0; a < NP; ++a)
        {
          const int me = a * 16 + j;
          const bool same_bucket = lk[a].miss && ob == lk[a].bucket && k != me;
          dup[a] = dup[a] || (same_bucket && os == slot[a]);
          rank[a] += (same_bucket && os != slot[a] && k < me) ? 1u : 0u;
        }
      }
#pragma unroll
    for (int a = 0; a < NP; ++a)
      if (lk[a].miss)
      {
        uint32_t e = lk[a].empty;
        for (uint32_t c = 0; c < rank[a]; ++c) e &= e - 1u;   // drop the ways taken by earlier claimants
        if (dup[a] || e == 0u)
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
    bool anyslow = false;
#pragma unroll
    for (int a = 0; a < NP; ++a) anyslow = anyslow || slow[a];
    if (rarely(__any(anyslow)))
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
      if (fresh[a]) tab.base[(pos[a] >> 2) & tab.bmask].key[pos[a] & 3u] = lk[a].kw | kSharedBit;
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


} // namespace grlx
