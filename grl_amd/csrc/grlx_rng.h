// grlx_rng.h -- 48-bit LCG of the drand48 family with jump-ahead: the reference's RNG streams (utils.h:84-187) and the lazy
// initialisation of weight-table slots (linear.cpp:110-121).
// Part of the single translation unit grlx_kernels.hip (included there, in order; not self-contained).
#pragma once

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

// The wide kernels have no 16 KB of LDS to spare (their parked state already takes 24 KB of the 40 KB a wave may use with four
// waves per CU): a table of 6-bit windows is 6 KB -- six multiply-adds instead of four, all reads from LDS instead of eight from
// device memory.  The maps are powers of one affine map, so any window width gives the same x.
struct JumpTable6 { uint64_t a[6][64], c[6][64]; };
constexpr JumpTable6 make_jump_table6()
{
  JumpTable6 t{};
  uint64_t sa = kLcgA, sc = kLcgC;                 // map of 64^w draws
  for (int w = 0; w < 6; ++w)
  {
    uint64_t a = 1, c = 0;
    for (int b = 0; b < 64; ++b)
    {
      t.a[w][b] = a;
      t.c[w][b] = c;
      c = (sa * c + sc) & kMask48;
      a = (sa * a) & kMask48;
    }
    sa = a;
    sc = c;
  }
  return t;
}
__device__ const JumpTable6 d_jump6 = make_jump_table6();
constexpr int kJump6Words = 2 * 6 * 64;

__device__ __forceinline__ void jump_table6_to_lds(uint64_t *sh_jump6)
{
  const uint64_t *src = reinterpret_cast<const uint64_t *>(&d_jump6);
  for (int i = threadIdx.x; i < kJump6Words; i += blockDim.x) sh_jump6[i] = src[i];
  __syncthreads();
}

__device__ __forceinline__ uint64_t lcg_jump_lds6(const uint64_t *sh_jump6, uint64_t x, uint64_t n)
{
  uint64_t a[6], c[6];
#pragma unroll
  for (int w = 0; w < 6; ++w)
  {
    const uint32_t b = (uint32_t)(n >> (6 * w)) & 63u;
    a[w] = sh_jump6[w * 64 + b];
    c[w] = sh_jump6[6 * 64 + w * 64 + b];
  }
#pragma unroll
  for (int w = 0; w < 6; ++w) x = (a[w] * x + c[w]) & kMask48;
  if (n >> 36) x = lcg_step_pow2(x, (n >> 36) << 36);
  return x;
}

__device__ __forceinline__ double lazy_weight_lds6(const uint64_t *sh_jump6, uint64_t tl0, const LinearParams &lp, uint32_t slot)
{
  uint64_t x = lcg_next(lcg_jump_lds6(sh_jump6, tl0, lp.draws_before + (uint64_t)slot));
  return lp.init_min + lcg_double(x) * lp.init_range;
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


} // namespace grlx
