// grlx_tile.h -- hashed tile coding (tile_coding.cpp:103-149, tile_coding.h:78-151): MurmurHash2 over tile coordinates.
// Part of the single translation unit grlx_kernels.hip (included there, in order; not self-contained).
#pragma once

namespace grlx {

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


} // namespace grlx
