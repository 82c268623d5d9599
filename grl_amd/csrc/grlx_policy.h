// grlx_policy.h -- greedy / epsilon-greedy sampling helpers (greedy.cpp:47-218) and register-pinning utilities.
// Part of the single translation unit grlx_kernels.hip (included there, in order; not self-contained).
#pragma once

namespace grlx {

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


} // namespace grlx
