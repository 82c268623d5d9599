/* rand48.c -- the drand48-family 48-bit LCG, restated (TEST INFRASTRUCTURE).
 *
 * Follows the use the reference makes of glibc's srand48_r/drand48_r/lrand48
 * (base/include/grl/utils.h:84-137):
 *   X(n+1) = (0x5DEECE66D * X(n) + 0xB) mod 2^48
 *   srand48(seed):  X = (low 32 bits of seed) << 16 | 0x330E
 *   drand48():      advance, return X * 2^-48   (exact: 48 bits fit a double)
 *   lrand48():      advance, return X >> 17     (31 bits)
 * tests/test_oracle_rng.py checks these against the host libc where present.
 */
#include "oracle.h"

#define LCG_A 0x5DEECE66DULL
#define LCG_C 0xBULL
#define MASK48 ((1ULL << 48) - 1)

void orc_srand48(orc_rand48 *g, long seed)
{
  g->x = ((((uint64_t)seed) & 0xFFFFFFFFULL) << 16) | 0x330EULL;
}

static inline void advance(orc_rand48 *g)
{
  g->x = (LCG_A * g->x + LCG_C) & MASK48;
}

double orc_drand48(orc_rand48 *g)
{
  advance(g);
  return (double)g->x * 0x1p-48;
}

uint32_t orc_lrand48(orc_rand48 *g)
{
  advance(g);
  return (uint32_t)(g->x >> 17);
}

void orc_rand48_jump(orc_rand48 *g, uint64_t n)
{
  /* compose the affine map x -> a*x + c with itself by repeated squaring */
  uint64_t a = LCG_A, c = LCG_C, x = g->x;
  while (n)
  {
    if (n & 1)
      x = (a * x + c) & MASK48;
    c = ((a + 1) * c) & MASK48;
    a = (a * a) & MASK48;
    n >>= 1;
  }
  g->x = x;
}

double orc_lazy_weight(uint32_t tl_seed, uint64_t draws_before, uint32_t slot,
                       double init_min, double init_max)
{
  /* linear.cpp:117-120: params_[ii] = rand->getUniform(init_min, init_max) in
   * index order; utils.h:110-113: a + get()*(b-a). */
  orc_rand48 g;
  orc_srand48(&g, (long)tl_seed);
  orc_rand48_jump(&g, draws_before + (uint64_t)slot);
  return init_min + orc_drand48(&g) * (init_max - init_min);
}
