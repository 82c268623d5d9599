/* tile_coding.c -- hashed tile coding, restated (TEST INFRASTRUCTURE).
 *
 * Follows base/src/projectors/tile_coding.cpp:45-80 (configure: scaling,
 * integer wrapping), :103-149 (_project) and
 * base/include/grl/projectors/tile_coding.h:78-118 (MurmurHash2 over the
 * coordinate tuple with seed 449, index = hash % memory), safe = 0 only.
 */
#include <math.h>
#include "oracle.h"

static int smod(int x, int y)
{ /* utils.h:70-78 safe_mod: C remainder lifted into [0, y) */
  int r = x % y;
  return r < 0 ? r + y : r;
}

static uint32_t murmur2_ints(const int *v, uint32_t n, uint32_t seed)
{ /* tile_coding.h:78-114: MurmurHash2 (Appleby), 4 bytes per coordinate */
  const uint32_t m = 0x5bd1e995u;
  uint32_t h = seed ^ n;
  for (uint32_t i = 0; i < n; ++i)
  {
    uint32_t k = (uint32_t)v[i];
    k *= m;
    k ^= k >> 24;
    k *= m;
    h *= m;
    h ^= k;
  }
  h ^= h >> 13;
  h *= m;
  h ^= h >> 15;
  return h;
}

static int tile_project_impl(const orc_tile_spec *t, const double *in, uint32_t *out, int full_hash);

int orc_tile_project(const orc_tile_spec *t, const double *in, uint32_t *out) { return tile_project_impl(t, in, out, 0); }
/* the same with the full 32-bit hash sums (createHashSum, tile_coding.h:78-113) instead of `% memory`: what the collision
 * table of safe >= 1 keys its claims with (getFeatureLocation, tile_coding.h:116-151) */
int orc_tile_project_hash(const orc_tile_spec *t, const double *in, uint32_t *out) { return tile_project_impl(t, in, out, 1); }

static int tile_project_impl(const orc_tile_spec *t, const double *in, uint32_t *out, int full_hash)
{
  int q[ORC_MAX_DIMS], base[ORC_MAX_DIMS], wrap[ORC_MAX_DIMS], c[ORC_MAX_DIMS + 1];
  double scaling[ORC_MAX_DIMS];
  int D = t->dims;

  if (D < 1 || D > ORC_MAX_DIMS || t->tilings < 1 || t->memory < 1)
    return -1;

  for (int i = 0; i < D; ++i)
  { /* tile_coding.cpp:66-78 */
    scaling[i] = t->tilings / t->resolution[i];
    double w = t->wrapping[i] * scaling[i];
    if (fabs(w - round(w)) > 0.001)
      return -1;
    wrap[i] = (int)round(w);
  }

  for (int i = 0; i < D; ++i)
  { /* :121-125 */
    q[i] = (int)floor(in[i] * scaling[i]);
    base[i] = 0;
  }

  for (int j = 0; j < t->tilings; ++j)
  { /* :128-146 */
    int i;
    for (i = 0; i < D; ++i)
    {
      c[i] = q[i] - smod(q[i] - base[i], t->tilings);
      if (wrap[i] != 0)
        c[i] = smod(c[i], wrap[i]);
      base[i] += 1 + 2 * i;
    }
    c[i] = j;
    /* tile_coding.h:118: unsigned % int -> unsigned arithmetic */
    out[j] = murmur2_ints(c, (uint32_t)(D + 1), 449u);
    if (!full_hash) out[j] %= (uint32_t)t->memory;
  }
  return 0;
}
