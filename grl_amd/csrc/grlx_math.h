// grlx_math.h -- device transcendental functions with a bit-exact specification.
//
// The reference calls glibc's sin/cos/log from its dynamics and noise code
// (base/src/environments/pendulum.cpp:62, cart_pole.cpp:64-65, acrobot.cpp,
// base/include/grl/utils.h:120-125).  Those results are not specified bit-wise,
// so this path defines its own routines out of IEEE-754 binary64 +,-,*,/, fma
// and rint only; every operation below is a single correctly rounded
// instruction on gfx950 (v_fma_f64, v_mul_f64, v_add_f64, v_rndne_f64).  The
// translation unit MUST be compiled with -ffp-contract=off so that nothing but
// the explicit __builtin_fma calls is fused.
//
// Specification (checked bit for bit against an independent CPU restatement in
// tests/test_gpu_math.py):
//   reduce(x): fn = rint(x*INVPIO2); r0 = fma(-fn,P1,x)  (exact for |x| < 2^20)
//              p = fn*P2; pl = fma(fn,P2,-p); r = r0-p; e = (r0-r)-p   [Fast2Sum]
//              t = (e-pl) - fn*P3; reduced argument = r + t, NOT renormalised (the
//              kernels need t to first order only); quadrant = fn mod 4
//   poly(z; c1..c8), Estrin: z2=z*z; z4=z2*z2; a=fma(z,c2,c1); b=fma(z,c4,c3);
//              c=fma(z,c6,c5); d=fma(z,c8,c7); lo=fma(z2,b,a); hi=fma(z2,d,c); fma(z4,hi,lo)
//   ksin(r,t): z=r*r; P=poly(z;S1..S8); r + fma(z*r, P, fma(-0.5*z, t, t))
//   kcos(r,t): z=r*r; hz=0.5*z; w=1-hz; tail=(1-w)-hz; Q=poly(z;C1..C8);
//              w + fma(z*z, Q, fma(-r, t, tail))
// The dependent chain of one sine is ~13 operations; it bounds the RK4 latency at one
// wave per SIMD, where a dependent f64 operation costs ~19 cycles.
// Constants are Taylor coefficients 1/k! and a three-double split of pi/2, all
// correctly rounded from exact rationals (tools/gen_math_constants.py).
// Accuracy: <= 1 ulp from glibc on 3 % of arguments, identical elsewhere.
// No tiny-argument shortcut (the general path returns x resp. 1; sin(-0) = +0).
// Domain: |x| < 2^20: psin/pcos are branch-free and unchecked, psin_checked/pcos_checked
// return NaN outside; the rollout kernel checks its states once per step (GRLX_ERR_DOMAIN).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace grlx {

#define GRLX_PIO2_1  0x1.921fb54442d18p+0
#define GRLX_PIO2_2  0x1.1a62633145c07p-54
#define GRLX_PIO2_3  -0x1.f1976b7ed8fbcp-110
#define GRLX_INVPIO2 0x1.45f306dc9c883p-1
#define GRLX_PI      0x1.921fb54442d18p+1
#define GRLX_2PI     0x1.921fb54442d18p+2

__device__ __forceinline__ int math_reduce(double x, double &rh, double &rl)
{
  double fn = __builtin_rint(x * GRLX_INVPIO2);
  double r0 = __builtin_fma(-fn, GRLX_PIO2_1, x);
  double p  = fn * GRLX_PIO2_2;
  double pl = __builtin_fma(fn, GRLX_PIO2_2, -p);
  double r  = r0 - p;
  double e  = (r0 - r) - p;
  double t  = (e - pl) - fn * GRLX_PIO2_3;
  rh = r;
  rl = t;
  return (int)((long long)fn & 3);
}

__device__ __forceinline__ double math_estrin8(double z, double c1, double c2, double c3, double c4,
                                               double c5, double c6, double c7, double c8)
{
  double z2 = z * z, z4 = z2 * z2;
  double a = __builtin_fma(z, c2, c1), b = __builtin_fma(z, c4, c3), c = __builtin_fma(z, c6, c5), d = __builtin_fma(z, c8, c7);
  double lo = __builtin_fma(z2, b, a), hi = __builtin_fma(z2, d, c);
  return __builtin_fma(z4, hi, lo);
}

__device__ __forceinline__ double math_ksin(double r, double rl)
{
  double z = r * r;
  double P = math_estrin8(z, -0x1.5555555555555p-3, 0x1.1111111111111p-7, -0x1.a01a01a01a01ap-13, 0x1.71de3a556c734p-19,
                          -0x1.ae64567f544e4p-26, 0x1.6124613a86d09p-33, -0x1.ae7f3e733b81fp-41, 0x1.952c77030ad4ap-49);
  return r + __builtin_fma(z * r, P, __builtin_fma(-0.5 * z, rl, rl));
}

__device__ __forceinline__ double math_kcos(double r, double rl)
{
  double z = r * r, hz = 0.5 * z;
  double w = 1.0 - hz;
  double tail = (1.0 - w) - hz;
  double Q = math_estrin8(z, 0x1.5555555555555p-5, -0x1.6c16c16c16c17p-10, 0x1.a01a01a01a01ap-16, -0x1.27e4fb7789f5cp-22,
                          0x1.1eed8eff8d898p-29, -0x1.93974a8c07c9dp-37, 0x1.ae7f3e733b81fp-45, -0x1.6827863b97d97p-53);
  return w + __builtin_fma(z * z, Q, __builtin_fma(-r, rl, tail));
}

// Both kernels of one reduced argument, sharing z, z^2, z^4 (the selects a per-lane choice
// of coefficients would need cost more than the second set of seven fma).
__device__ __forceinline__ void math_kboth(double r, double t, double &sn, double &cs)
{
  const double z = r * r, z2 = z * z, z4 = z2 * z2;
  const double sa = __builtin_fma(z, 0x1.1111111111111p-7, -0x1.5555555555555p-3);
  const double sb = __builtin_fma(z, 0x1.71de3a556c734p-19, -0x1.a01a01a01a01ap-13);
  const double sc = __builtin_fma(z, 0x1.6124613a86d09p-33, -0x1.ae64567f544e4p-26);
  const double sd = __builtin_fma(z, 0x1.952c77030ad4ap-49, -0x1.ae7f3e733b81fp-41);
  const double P = __builtin_fma(z4, __builtin_fma(z2, sd, sc), __builtin_fma(z2, sb, sa));
  const double ca = __builtin_fma(z, -0x1.6c16c16c16c17p-10, 0x1.5555555555555p-5);
  const double cb = __builtin_fma(z, -0x1.27e4fb7789f5cp-22, 0x1.a01a01a01a01ap-16);
  const double cc = __builtin_fma(z, -0x1.93974a8c07c9dp-37, 0x1.1eed8eff8d898p-29);
  const double cd = __builtin_fma(z, -0x1.6827863b97d97p-53, 0x1.ae7f3e733b81fp-45);
  const double Q = __builtin_fma(z4, __builtin_fma(z2, cd, cc), __builtin_fma(z2, cb, ca));
  const double hz = 0.5 * z;
  sn = r + __builtin_fma(z * r, P, __builtin_fma(-hz, t, t));
  const double w = 1.0 - hz;
  const double tail = (1.0 - w) - hz;
  cs = w + __builtin_fma(z2, Q, __builtin_fma(-r, t, tail));
}

// The constants of reduce + both kernels held in vector registers.  A lone wave issues one
// instruction every four cycles whatever its kind, so re-materialising 64-bit literals into
// scalar registers inside a loop (what the compiler does when it runs out of them) costs as
// much as the arithmetic; hoisted into VGPRs once (pin()), the loop has no such moves.
struct SinConsts {
  double invpio2, p1, p2, p3;
  double s[8], c[8];
};

__device__ __forceinline__ double math_pin(double v) { asm volatile("" : "+v"(v)); return v; }
// PIN = false leaves the constants to the compiler (literals): the choice for kernels that run
// several waves per SIMD, where registers are scarcer than scalar issue slots.
template <bool PIN> __device__ __forceinline__ double math_const(double v) { return PIN ? math_pin(v) : v; }

template <bool PIN = true>
__device__ __forceinline__ SinConsts sin_consts()
{
  SinConsts k;
  k.invpio2 = math_const<PIN>(GRLX_INVPIO2);
  k.p1 = math_const<PIN>(GRLX_PIO2_1);
  k.p2 = math_const<PIN>(GRLX_PIO2_2);
  k.p3 = math_const<PIN>(GRLX_PIO2_3);
  const double sv[8] = {-0x1.5555555555555p-3, 0x1.1111111111111p-7, -0x1.a01a01a01a01ap-13, 0x1.71de3a556c734p-19,
                        -0x1.ae64567f544e4p-26, 0x1.6124613a86d09p-33, -0x1.ae7f3e733b81fp-41, 0x1.952c77030ad4ap-49};
  const double cv[8] = {0x1.5555555555555p-5, -0x1.6c16c16c16c17p-10, 0x1.a01a01a01a01ap-16, -0x1.27e4fb7789f5cp-22,
                        0x1.1eed8eff8d898p-29, -0x1.93974a8c07c9dp-37, 0x1.ae7f3e733b81fp-45, -0x1.6827863b97d97p-53};
#pragma unroll
  for (int i = 0; i < 8; ++i) { k.s[i] = math_const<PIN>(sv[i]); k.c[i] = math_const<PIN>(cv[i]); }
  return k;
}

// psin with the constants supplied: the same operations as psin below, in the same order
__device__ __forceinline__ double psin(double x, const SinConsts &k)
{
  const double fn = __builtin_rint(x * k.invpio2);
  const double r0 = __builtin_fma(-fn, k.p1, x);
  const double p  = fn * k.p2;
  const double pl = __builtin_fma(fn, k.p2, -p);
  const double r = r0 - p;
  const double e = (r0 - r) - p;
  const double t = (e - pl) - fn * k.p3;
  const int q = (int)fn;
  const double z = r * r, z2 = z * z, z4 = z2 * z2;
  const double sa = __builtin_fma(z, k.s[1], k.s[0]);
  const double sb = __builtin_fma(z, k.s[3], k.s[2]);
  const double sc = __builtin_fma(z, k.s[5], k.s[4]);
  const double sd = __builtin_fma(z, k.s[7], k.s[6]);
  const double P = __builtin_fma(z4, __builtin_fma(z2, sd, sc), __builtin_fma(z2, sb, sa));
  const double ca = __builtin_fma(z, k.c[1], k.c[0]);
  const double cb = __builtin_fma(z, k.c[3], k.c[2]);
  const double cc = __builtin_fma(z, k.c[5], k.c[4]);
  const double cd = __builtin_fma(z, k.c[7], k.c[6]);
  const double Q = __builtin_fma(z4, __builtin_fma(z2, cd, cc), __builtin_fma(z2, cb, ca));
  const double hz = 0.5 * z;
  const double sn = r + __builtin_fma(z * r, P, __builtin_fma(-hz, t, t));
  const double w = 1.0 - hz;
  const double tail = (1.0 - w) - hz;
  const double cs = w + __builtin_fma(z2, Q, __builtin_fma(-r, t, tail));
  const double v = (q & 1) ? cs : sn;
  return (q & 2) ? -v : v;
}

// reduce + both kernels with supplied constants (the operations of math_reduce / math_kboth)
__device__ __forceinline__ int math_both(double x, const SinConsts &k, double &sn, double &cs)
{
  const double fn = __builtin_rint(x * k.invpio2);
  const double r0 = __builtin_fma(-fn, k.p1, x);
  const double p  = fn * k.p2;
  const double pl = __builtin_fma(fn, k.p2, -p);
  const double r = r0 - p;
  const double e = (r0 - r) - p;
  const double t = (e - pl) - fn * k.p3;
  const double z = r * r, z2 = z * z, z4 = z2 * z2;
  const double sa = __builtin_fma(z, k.s[1], k.s[0]);
  const double sb = __builtin_fma(z, k.s[3], k.s[2]);
  const double sc = __builtin_fma(z, k.s[5], k.s[4]);
  const double sd = __builtin_fma(z, k.s[7], k.s[6]);
  const double P = __builtin_fma(z4, __builtin_fma(z2, sd, sc), __builtin_fma(z2, sb, sa));
  const double ca = __builtin_fma(z, k.c[1], k.c[0]);
  const double cb = __builtin_fma(z, k.c[3], k.c[2]);
  const double cc = __builtin_fma(z, k.c[5], k.c[4]);
  const double cd = __builtin_fma(z, k.c[7], k.c[6]);
  const double Q = __builtin_fma(z4, __builtin_fma(z2, cd, cc), __builtin_fma(z2, cb, ca));
  const double hz = 0.5 * z;
  sn = r + __builtin_fma(z * r, P, __builtin_fma(-hz, t, t));
  const double w = 1.0 - hz;
  const double tail = (1.0 - w) - hz;
  cs = w + __builtin_fma(z2, Q, __builtin_fma(-r, t, tail));
  return (int)((long long)fn & 3);
}

__device__ __forceinline__ double pcos(double x, const SinConsts &k)
{
  double sn, cs;
  const int q = math_both(x, k, sn, cs) + 1;    // cos(x) = sin(x + pi/2)
  const double v = (q & 1) ? cs : sn;
  return (q & 2) ? -v : v;
}

__device__ __forceinline__ void psincos(double x, const SinConsts &k, double &sn_out, double &cs_out)
{
  double sn, cs;
  const int q = math_both(x, k, sn, cs);
  const double vs = (q & 1) ? cs : sn;
  sn_out = (q & 2) ? -vs : vs;
  const double vc = (q & 1) ? sn : cs;
  cs_out = ((q + 1) & 2) ? -vc : vc;
}

// Small-angle-aware forms for code whose angles mostly stay within a quarter turn (the compass walker):
// when rint(x * 2/pi) is zero in EVERY lane of the wave the reduction is the identity (r = fma(-fn,P1,x),
// t = +0: every term of it is a signed zero that sums to +0) and only the kernel that is asked for is
// evaluated -- the same operations the general path would perform on (r, +0), hence the same bits;
// fma(-hz,t,t) = +0 and fma(-r,t,tail) = tail are the values those operations have for t = +0.
// Anything else takes the general path.  One wave-uniform test per call.
__device__ __forceinline__ double psin_s(double x, const SinConsts &k)
{
  const double fn = __builtin_rint(x * k.invpio2);
  if (__builtin_expect(__all(fn == 0.0), 1))
  {
    const double r = __builtin_fma(-fn, k.p1, x);
    const double z = r * r, z2 = z * z, z4 = z2 * z2;
    const double sa = __builtin_fma(z, k.s[1], k.s[0]);
    const double sb = __builtin_fma(z, k.s[3], k.s[2]);
    const double sc = __builtin_fma(z, k.s[5], k.s[4]);
    const double sd = __builtin_fma(z, k.s[7], k.s[6]);
    const double P = __builtin_fma(z4, __builtin_fma(z2, sd, sc), __builtin_fma(z2, sb, sa));
    return r + __builtin_fma(z * r, P, 0.0);
  }
  return psin(x, k);
}

__device__ __forceinline__ double pcos_s(double x, const SinConsts &k)
{
  const double fn = __builtin_rint(x * k.invpio2);
  if (__builtin_expect(__all(fn == 0.0), 1))
  {
    const double r = __builtin_fma(-fn, k.p1, x);
    const double z = r * r, z2 = z * z, z4 = z2 * z2;
    const double ca = __builtin_fma(z, k.c[1], k.c[0]);
    const double cb = __builtin_fma(z, k.c[3], k.c[2]);
    const double cc = __builtin_fma(z, k.c[5], k.c[4]);
    const double cd = __builtin_fma(z, k.c[7], k.c[6]);
    const double Q = __builtin_fma(z4, __builtin_fma(z2, cd, cc), __builtin_fma(z2, cb, ca));
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double tail = (1.0 - w) - hz;
    return w + __builtin_fma(z2, Q, tail);
  }
  return pcos(x, k);
}

__device__ __forceinline__ void psincos_s(double x, const SinConsts &k, double &sn_out, double &cs_out)
{
  const double fn = __builtin_rint(x * k.invpio2);
  if (__builtin_expect(__all(fn == 0.0), 1))
  {
    const double r = __builtin_fma(-fn, k.p1, x);
    const double z = r * r, z2 = z * z, z4 = z2 * z2;
    const double sa = __builtin_fma(z, k.s[1], k.s[0]);
    const double sb = __builtin_fma(z, k.s[3], k.s[2]);
    const double sc = __builtin_fma(z, k.s[5], k.s[4]);
    const double sd = __builtin_fma(z, k.s[7], k.s[6]);
    const double P = __builtin_fma(z4, __builtin_fma(z2, sd, sc), __builtin_fma(z2, sb, sa));
    const double ca = __builtin_fma(z, k.c[1], k.c[0]);
    const double cb = __builtin_fma(z, k.c[3], k.c[2]);
    const double cc = __builtin_fma(z, k.c[5], k.c[4]);
    const double cd = __builtin_fma(z, k.c[7], k.c[6]);
    const double Q = __builtin_fma(z4, __builtin_fma(z2, cd, cc), __builtin_fma(z2, cb, ca));
    const double hz = 0.5 * z;
    sn_out = r + __builtin_fma(z * r, P, 0.0);
    const double w = 1.0 - hz;
    const double tail = (1.0 - w) - hz;
    cs_out = w + __builtin_fma(z2, Q, tail);
    return;
  }
  psincos(x, k, sn_out, cs_out);
}

// Branch-free forms: callers guarantee |x| < 2^20 (checked once per environment step;
// outside the domain the result is unspecified and the replica is flagged).
__device__ __forceinline__ double psin(double x)
{
  double r, t, sn, cs;
  const double fn = __builtin_rint(x * GRLX_INVPIO2);
  {
    const double r0 = __builtin_fma(-fn, GRLX_PIO2_1, x);
    const double p  = fn * GRLX_PIO2_2;
    const double pl = __builtin_fma(fn, GRLX_PIO2_2, -p);
    r = r0 - p;
    const double e = (r0 - r) - p;
    t = (e - pl) - fn * GRLX_PIO2_3;
  }
  const int q = (int)fn;                        // |fn| < 2^20
  math_kboth(r, t, sn, cs);
  const double v = (q & 1) ? cs : sn;
  return (q & 2) ? -v : v;
}

__device__ __forceinline__ double pcos(double x)
{
  double rh, rl, sn, cs;
  const int q = math_reduce(x, rh, rl) + 1;     // cos(x) = sin(x + pi/2)
  math_kboth(rh, rl, sn, cs);
  const double v = (q & 1) ? cs : sn;
  return (q & 2) ? -v : v;
}

// sin and cos of the same argument share everything but the final selection
__device__ __forceinline__ void psincos(double x, double &sn_out, double &cs_out)
{
  double rh, rl, sn, cs;
  const int q = math_reduce(x, rh, rl);
  math_kboth(rh, rl, sn, cs);
  const double vs = (q & 1) ? cs : sn;
  sn_out = (q & 2) ? -vs : vs;
  const double vc = (q & 1) ? sn : cs;
  cs_out = ((q + 1) & 2) ? -vc : vc;
}

// checked entry points (fine-grained operators): NaN outside the domain
__device__ __forceinline__ double psin_checked(double x) { return (__builtin_fabs(x) < 0x1p20) ? psin(x) : __builtin_nan(""); }
__device__ __forceinline__ double pcos_checked(double x) { return (__builtin_fabs(x) < 0x1p20) ? pcos(x) : __builtin_nan(""); }

// x / 6, correctly rounded, in three operations: q0 = x*c, r = fma(-6,q0,x) (exact),
// q = fma(r,c,q0) = RN(x/6 * (1 + 2^-107)).  x/6 = (x/3)/2 and the quotient of a double by 3
// is either exact or 1/3 resp. 2/3 of an ulp beyond a representable number -- never within
// 2^-54 ulp of a rounding midpoint -- so the perturbation cannot change the rounding:
// the result equals the IEEE division bit for bit (checked against v_div on random and
// structured inputs in tests/test_gpu_parity.py).  Tiny |x| takes the true division.
__device__ __forceinline__ double div6(double x)
{
  const double c = 0x1.5555555555555p-3;        // RN(1/6)
  const double q0 = x * c;
  const double r = __builtin_fma(-6.0, q0, x);
  double q = __builtin_fma(r, c, q0);
  // tiny |x| (never seen in practice): one wave-uniform test instead of an exec-mask region per call
  const bool tiny = !(__builtin_fabs(x) > 0x1p-900);
  if (__builtin_expect(__any(tiny), 0))
    q = tiny ? x / 6.0 : q;
  return q;
}

// plog(x): x = m*2^k, m in [sqrt2/2, sqrt2); f = m-1; d = f+2; s = f/d;
// sl = (fma(-s,d,f) - s*(2-(d-f)))/d; z = s*s; R = L1+z*(L2+..+z*L11);
// lo = fma(k, LN2_LO, fma(s*z, R, 2*sl)); result fma(k, LN2_HI, 2*s+lo).
__device__ __forceinline__ double plog(double x)
{
  if (x != x || x < 0.0) return __builtin_nan("");
  if (x == 0.0) return -__builtin_inf();
  if (x == __builtin_inf()) return x;
  unsigned long long b = (unsigned long long)__double_as_longlong(x);
  int k = 0;
  if ((b >> 52) == 0)
  {
    x *= 0x1p54;
    b = (unsigned long long)__double_as_longlong(x);
    k = -54;
  }
  k += (int)(b >> 52) - 1023;
  b = (b & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL;
  double m = __longlong_as_double((long long)b);
  if (m > 0x1.6a09e667f3bcdp+0) { m *= 0.5; k += 1; }
  double f = m - 1.0;
  double d = f + 2.0;
  double dl = 2.0 - (d - f);
  double s = f / d;
  double sl = (__builtin_fma(-s, d, f) - s * dl) / d;
  double z = s * s;
  double R = __builtin_fma(z, 0x1.642c8590b2164p-4, 0x1.8618618618618p-4);
  R = __builtin_fma(z, R, 0x1.af286bca1af28p-4);
  R = __builtin_fma(z, R, 0x1.e1e1e1e1e1e1ep-4);
  R = __builtin_fma(z, R, 0x1.1111111111111p-3);
  R = __builtin_fma(z, R, 0x1.3b13b13b13b14p-3);
  R = __builtin_fma(z, R, 0x1.745d1745d1746p-3);
  R = __builtin_fma(z, R, 0x1.c71c71c71c71cp-3);
  R = __builtin_fma(z, R, 0x1.2492492492492p-2);
  R = __builtin_fma(z, R, 0x1.999999999999ap-2);
  R = __builtin_fma(z, R, 0x1.5555555555555p-1);
  double dk = (double)k;
  double lo = __builtin_fma(dk, 0x1.fdf473de6af28p-22, __builtin_fma(s * z, R, 2.0 * sl));
  return __builtin_fma(dk, 0x1.62e4200000000p-1, 2.0 * s + lo);
}

// pexp(x) (the logistic activation of representation/parameterized/ann, ann.h:108-111): k = rint(x*LOG2E);
// r = fma(-k, LN2_LO, fma(-k, LN2_HI, x)); q = Horner (fma) over 1/13! .. 1/2!; e = 1 + fma(r*r, q, r);
// result (e * 2^(k/2)) * 2^(k - k/2).  The specification is oracle/portable_math.c: orc_pexp.
__device__ __forceinline__ double ppow2(int k)
{
  return __longlong_as_double((long long)((unsigned long long)(k + 1023) << 52));
}
__device__ __forceinline__ double pexp(double x)
{
  if (x != x) return __builtin_nan("");
  if (x > 709.782712893384) return __builtin_inf();
  if (x < -745.2) return 0.0;
  const double kd = __builtin_rint(x * 0x1.71547652b82fep+0);
  double r = __builtin_fma(-kd, 0x1.62e4200000000p-1, x);
  r = __builtin_fma(-kd, 0x1.fdf473de6af28p-22, r);
  double q = 1.0 / 6227020800.0;
  q = __builtin_fma(r, q, 1.0 / 479001600.0);
  q = __builtin_fma(r, q, 1.0 / 39916800.0);
  q = __builtin_fma(r, q, 1.0 / 3628800.0);
  q = __builtin_fma(r, q, 1.0 / 362880.0);
  q = __builtin_fma(r, q, 1.0 / 40320.0);
  q = __builtin_fma(r, q, 1.0 / 5040.0);
  q = __builtin_fma(r, q, 1.0 / 720.0);
  q = __builtin_fma(r, q, 1.0 / 120.0);
  q = __builtin_fma(r, q, 1.0 / 24.0);
  q = __builtin_fma(r, q, 1.0 / 6.0);
  q = __builtin_fma(r, q, 0.5);
  const double e = 1.0 + __builtin_fma(r * r, q, r);
  const int k = (int)kd, k1 = k / 2, k2 = k - k1;
  return (e * ppow2(k1)) * ppow2(k2);
}


// fmod(x, y) for finite x, y > 0: exact by definition (IEEE remainder toward
// zero); long division by exactly representable multiples of y.  Each
// subtraction is exact (Sterbenz), so the result equals libm's fmod bit for bit.
__device__ __forceinline__ double pfmod(double x, double y)
{
  double ax = __builtin_fabs(x);
  // every lane within two periods (the wrapped angles of the environments): the long division below
  // is then its last step alone -- one exact conditional subtraction
  if (__builtin_expect(__all(ax < y + y && y > 0.0), 1))
    return __builtin_copysign((ax >= y) ? ax - y : ax, x);
  if (!(ax < __builtin_inf()) || !(y > 0.0)) return __builtin_nan("");
  if (ax < y) return x;
  int ex = (int)(((unsigned long long)__double_as_longlong(ax) >> 52) & 0x7FF);
  int ey = (int)(((unsigned long long)__double_as_longlong(y) >> 52) & 0x7FF);
  if (ey == 0 || ex - ey > 1000)
    return fmod(x, y);                               // subnormal divisor / huge ratio: library path
  // t = y * 2^(ex-ey): same exponent as ax
  double t = __longlong_as_double(__double_as_longlong(y) + ((long long)(ex - ey) << 52));
  for (int i = ex - ey; i >= 0; --i)
  {
    if (ax >= t) ax -= t;
    t *= 0.5;
  }
  return __builtin_copysign(ax, x);
}

} // namespace grlx
