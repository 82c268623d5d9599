/* portable_math.c -- operation-by-operation specified sin/cos/log
 * (TEST INFRASTRUCTURE; the HIP kernels restate the same specification in
 * grl_amd/csrc/grlx_math.h and must agree with this file bit for bit).
 *
 * Why: the reference calls glibc's sin/cos/log (pendulum.cpp:62, cart_pole.cpp,
 * acrobot.cpp, utils.h:120-125).  glibc's results are not specified bit-wise and
 * are not available on the GPU, so the GPU path and the oracle's
 * ORC_MATH_PORTABLE mode share this specification instead; ORC_MATH_LIBM keeps
 * the libm calls to pin the reference's golden file.  tests/test_oracle_math.py
 * measures how often the two differ (<= 1 ulp, a few percent of arguments).
 *
 * Specification (all operations IEEE-754 binary64, round-to-nearest-even;
 * fma = correctly rounded fused multiply-add; no other contraction allowed --
 * compile with -ffp-contract=off):
 *
 *  reduce(x): fn = rint(x*INVPIO2); r0 = fma(-fn,P1,x)  [exact for |x|<2^20]
 *             p = fn*P2; pl = fma(fn,P2,-p); r = r0-p; e = (r0-r)-p  [Fast2Sum]
 *             t = (e-pl) - fn*P3;  reduced argument = r + t (NOT renormalised: the
 *             kernels only need t to first order); quadrant = fn mod 4
 *  poly(z; c1..c8) = Estrin: z2=z*z; z4=z2*z2;
 *             a=fma(z,c2,c1); b=fma(z,c4,c3); c=fma(z,c6,c5); d=fma(z,c8,c7);
 *             lo=fma(z2,b,a); hi=fma(z2,d,c); result fma(z4,hi,lo)
 *  ksin(r,t): z=r*r; P=poly(z;S1..S8); result r + fma(z*r, P, fma(-0.5*z, t, t))
 *  kcos(r,t): z=r*r; hz=0.5*z; w=1-hz; tail=(1-w)-hz; Q=poly(z;C1..C8);
 *             result w + fma(z*z, Q, fma(-r, t, tail))
 *  Constants: tools/gen_math_constants.py (Taylor coefficients 1/k! and a
 *  three-double split of pi/2, all correctly rounded from exact rationals).
 *  No special case for tiny |x| (the general path returns x resp. 1 there; sin(-0) = +0).
 *  Domain: |x| < 2^20 (NaN outside; the environments never get there).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "oracle.h"

#define PM_PIO2_1  0x1.921fb54442d18p+0
#define PM_PIO2_2  0x1.1a62633145c07p-54
#define PM_PIO2_3  -0x1.f1976b7ed8fbcp-110
#define PM_INVPIO2 0x1.45f306dc9c883p-1
#define PM_S1      -0x1.5555555555555p-3
#define PM_S2      0x1.1111111111111p-7
#define PM_S3      -0x1.a01a01a01a01ap-13
#define PM_S4      0x1.71de3a556c734p-19
#define PM_S5      -0x1.ae64567f544e4p-26
#define PM_S6      0x1.6124613a86d09p-33
#define PM_S7      -0x1.ae7f3e733b81fp-41
#define PM_S8      0x1.952c77030ad4ap-49
#define PM_C1      0x1.5555555555555p-5
#define PM_C2      -0x1.6c16c16c16c17p-10
#define PM_C3      0x1.a01a01a01a01ap-16
#define PM_C4      -0x1.27e4fb7789f5cp-22
#define PM_C5      0x1.1eed8eff8d898p-29
#define PM_C6      -0x1.93974a8c07c9dp-37
#define PM_C7      0x1.ae7f3e733b81fp-45
#define PM_C8      -0x1.6827863b97d97p-53
#define PM_LN2_HI  0x1.62e4200000000p-1
#define PM_LN2_LO  0x1.fdf473de6af28p-22
#define PM_SQRT2   0x1.6a09e667f3bcdp+0
#define PM_L1      0x1.5555555555555p-1
#define PM_L2      0x1.999999999999ap-2
#define PM_L3      0x1.2492492492492p-2
#define PM_L4      0x1.c71c71c71c71cp-3
#define PM_L5      0x1.745d1745d1746p-3
#define PM_L6      0x1.3b13b13b13b14p-3
#define PM_L7      0x1.1111111111111p-3
#define PM_L8      0x1.e1e1e1e1e1e1ep-4
#define PM_L9      0x1.af286bca1af28p-4
#define PM_L10     0x1.8618618618618p-4
#define PM_L11     0x1.642c8590b2164p-4

static int reduce(double x, double *rh, double *rl)
{
  double fn = rint(x * PM_INVPIO2);
  double r0 = fma(-fn, PM_PIO2_1, x);
  double p  = fn * PM_PIO2_2;
  double pl = fma(fn, PM_PIO2_2, -p);
  double r  = r0 - p;
  double e  = (r0 - r) - p;
  double t  = (e - pl) - fn * PM_PIO2_3;
  *rh = r;
  *rl = t;
  return (int)((int64_t)fn & 3);
}

static double estrin8(double z, double c1, double c2, double c3, double c4, double c5, double c6, double c7, double c8)
{
  double z2 = z * z, z4 = z2 * z2;
  double a = fma(z, c2, c1), b = fma(z, c4, c3), c = fma(z, c6, c5), d = fma(z, c8, c7);
  double lo = fma(z2, b, a), hi = fma(z2, d, c);
  return fma(z4, hi, lo);
}

static double ksin(double r, double rl)
{
  double z = r * r;
  double P = estrin8(z, PM_S1, PM_S2, PM_S3, PM_S4, PM_S5, PM_S6, PM_S7, PM_S8);
  return r + fma(z * r, P, fma(-0.5 * z, rl, rl));
}

static double kcos(double r, double rl)
{
  double z = r * r, hz = 0.5 * z;
  double w = 1.0 - hz;
  double tail = (1.0 - w) - hz;
  double Q = estrin8(z, PM_C1, PM_C2, PM_C3, PM_C4, PM_C5, PM_C6, PM_C7, PM_C8);
  return w + fma(z * z, Q, fma(-r, rl, tail));
}

double orc_psin(double x)
{
  double ax = fabs(x), rh, rl;
  if (!(ax < 0x1p20)) return NAN;
  switch (reduce(x, &rh, &rl))
  {
    case 0:  return ksin(rh, rl);
    case 1:  return kcos(rh, rl);
    case 2:  return -ksin(rh, rl);
    default: return -kcos(rh, rl);
  }
}

double orc_pcos(double x)
{
  double ax = fabs(x), rh, rl;
  if (!(ax < 0x1p20)) return NAN;
  switch (reduce(x, &rh, &rl))
  {
    case 0:  return kcos(rh, rl);
    case 1:  return -ksin(rh, rl);
    case 2:  return -kcos(rh, rl);
    default: return ksin(rh, rl);
  }
}

/* plog(x), x > 0 finite:  x = m*2^k, m in [sqrt2/2, sqrt2);  f = m-1 (exact);
 * d = f+2; s = f/d; sl = (fma(-s,d,f) - s*((2-(d-f)))) / d   [s+sl ~ f/(f+2)]
 * z = s*s; R = L1+z*(L2+...+z*L11) by fma-Horner;
 * lo = fma(k, LN2_LO, fma(s*z, R, 2*sl)); result = fma(k, LN2_HI, 2*s + lo)
 * where the final sum is evaluated as (k*LN2_HI exact) + (2*s + lo).
 * x == 0 -> -inf; x < 0 or NaN -> NaN; +inf -> +inf. */
double orc_plog(double x)
{
  uint64_t b;
  int k;
  if (x != x || x < 0.0) return NAN;
  if (x == 0.0) return -INFINITY;
  if (x == INFINITY) return x;
  memcpy(&b, &x, 8);
  k = 0;
  if ((b >> 52) == 0)
  { /* subnormal: scale by 2^54 */
    x *= 0x1p54;
    memcpy(&b, &x, 8);
    k = -54;
  }
  k += (int)(b >> 52) - 1023;
  b = (b & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL;
  double m;
  memcpy(&m, &b, 8);
  if (m > PM_SQRT2) { m *= 0.5; k += 1; }
  double f = m - 1.0;
  double d = f + 2.0;
  double dl = 2.0 - (d - f);
  double s = f / d;
  double sl = (fma(-s, d, f) - s * dl) / d;
  double z = s * s;
  double R = fma(z, PM_L11, PM_L10);
  R = fma(z, R, PM_L9);
  R = fma(z, R, PM_L8);
  R = fma(z, R, PM_L7);
  R = fma(z, R, PM_L6);
  R = fma(z, R, PM_L5);
  R = fma(z, R, PM_L4);
  R = fma(z, R, PM_L3);
  R = fma(z, R, PM_L2);
  R = fma(z, R, PM_L1);
  double dk = (double)k;
  double lo = fma(dk, PM_LN2_LO, fma(s * z, R, 2.0 * sl));
  return fma(dk, PM_LN2_HI, 2.0 * s + lo);
}

/* pexp(x) -- the logistic activation of representation/parameterized/ann needs exp (ann.h:108-111).
 * Specification: k = rint(x*LOG2E); r = fma(-k, LN2_LO, fma(-k, LN2_HI, x))  [k*LN2_HI is exact: LN2_HI has 21
 * trailing zero bits]; q = Horner with fma over the Taylor coefficients 1/13! .. 1/2! (each the correctly rounded
 * quotient 1.0/n!, n! exact in binary64); e = 1 + fma(r*r, q, r); result = (e * 2^k1) * 2^k2 with k1 = k/2 (C integer
 * division), k2 = k - k1 (both powers of two are normal numbers; only the last multiplication can round).
 * NaN -> NaN; x > 709.782712893384 -> +inf; x < -745.2 -> +0. */
static double pm_pow2(int k)
{
  uint64_t b = (uint64_t)(k + 1023) << 52;
  double d;
  memcpy(&d, &b, 8);
  return d;
}

double orc_pexp(double x)
{
  if (x != x) return NAN;
  if (x > 709.782712893384) return INFINITY;
  if (x < -745.2) return 0.0;
  const double kd = rint(x * 0x1.71547652b82fep+0);
  double r = fma(-kd, PM_LN2_HI, x);
  r = fma(-kd, PM_LN2_LO, r);
  double q = 1.0 / 6227020800.0;
  q = fma(r, q, 1.0 / 479001600.0);
  q = fma(r, q, 1.0 / 39916800.0);
  q = fma(r, q, 1.0 / 3628800.0);
  q = fma(r, q, 1.0 / 362880.0);
  q = fma(r, q, 1.0 / 40320.0);
  q = fma(r, q, 1.0 / 5040.0);
  q = fma(r, q, 1.0 / 720.0);
  q = fma(r, q, 1.0 / 120.0);
  q = fma(r, q, 1.0 / 24.0);
  q = fma(r, q, 1.0 / 6.0);
  q = fma(r, q, 0.5);
  const double e = 1.0 + fma(r * r, q, r);
  const int k = (int)kd, k1 = k / 2, k2 = k - k1;
  return (e * pm_pow2(k1)) * pm_pow2(k2);
}
