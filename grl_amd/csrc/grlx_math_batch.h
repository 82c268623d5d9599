// grlx_math_batch.h -- the portable exp and the IEEE division for N independent arguments, stage by stage (the batch path's logistic,
// grlx_fqi.hip).  Same operations per element as grlx_math.h's scalar forms: same bits.  Included by grlx_fqi.hip only, so that tuning
// these does not recompile the rollout kernels.
#pragma once
#include "grlx_math.h"

namespace grlx {

// The same function for N independent arguments, written stage by stage and without branches: every element sees exactly the
// operations of pexp (same bits), but the N dependent chains (rint, two reduction fmas, 12 Horner fmas, scaling) stand side by side, so
// a wave that is alone on its SIMD can fill the latency of one with the others.  The three special ranges are selected at the end;
// what the main path computes from such an argument is discarded (no instruction here traps).
template <int N>
__device__ __forceinline__ void pexp_batch(const double (&x)[N], double (&out)[N])
{ // __builtin_amdgcn_sched_barrier(0): nothing is scheduled across it -- the stages stay stages (N independent instructions each)
  double kd[N], r[N], q[N];
#pragma unroll
  for (int i = 0; i < N; ++i) kd[i] = __builtin_rint(x[i] * 0x1.71547652b82fep+0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = __builtin_fma(-kd[i], 0x1.62e4200000000p-1, x[i]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = __builtin_fma(-kd[i], 0x1.fdf473de6af28p-22, r[i]);
  __builtin_amdgcn_sched_barrier(0);
  const double c[12] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0,
                        1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5};
#pragma unroll
  for (int i = 0; i < N; ++i) q[i] = __builtin_fma(r[i], c[0], c[1]);
#pragma unroll
  for (int s = 2; s < 12; ++s)
  {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = __builtin_fma(r[i], q[i], c[s]);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) q[i] = 1.0 + __builtin_fma(r[i] * r[i], q[i], r[i]);
  __builtin_amdgcn_sched_barrier(0);
  // the three special ranges (NaN, overflow, underflow) are rare: one wave-uniform test decides whether anybody needs the selects
  // (and a result that may be subnormal, x < -708: the two-step scaling below is the specification there; for normal results ONE exact
  //  scaling by 2^k -- v_ldexp_f64 -- gives the same bits as the two exact-then-exact multiplications, in 2 instructions instead of 13)
  bool special = false;
#pragma unroll
  for (int i = 0; i < N; ++i) special |= !(x[i] >= -708.0 && x[i] <= 709.782712893384);
  if (__builtin_expect(__any(special), 0))
  {
#pragma unroll
    for (int i = 0; i < N; ++i)
    { // kd is within +-1100 wherever the main path is kept; clamped so that the exponent arithmetic of the discarded ones stays defined
      const int k = (int)__builtin_fmax(__builtin_fmin(kd[i], 2000.0), -2000.0), k1 = k / 2, k2 = k - k1;
      double v = (q[i] * ppow2(k1)) * ppow2(k2);
      v = (x[i] < -745.2) ? 0.0 : v;
      v = (x[i] > 709.782712893384) ? __builtin_inf() : v;
      v = (x[i] != x[i]) ? __builtin_nan("") : v;
      out[i] = v;
    }
  }
  else
  {
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = __builtin_ldexp(q[i], (int)kd[i]);      // |k| <= 1024, q in [0.7, 1.42], the result normal: exact
  }
  __builtin_amdgcn_sched_barrier(0);
}

// x / y[i] for N independent divisors, stage by stage: the correctly rounded IEEE quotient, by the operations the compiler itself emits
// for an f64 division on this target (LLVM AMDGPU LowerFDIV64: v_div_scale x2, v_rcp, two Newton steps, the scaled quotient with one
// residual correction in v_div_fmas, v_div_fixup) -- but with the N dependent chains side by side instead of one after the other.
template <int N>
__device__ __forceinline__ void pdiv_batch(double x, const double (&y)[N], double (&out)[N])
{
  double sd[N], sn[N], rcp[N], t[N], u[N];
  bool flag[N];
#pragma unroll
  for (int i = 0; i < N; ++i)
  {
    bool unused;
    sd[i] = __builtin_amdgcn_div_scale(x, y[i], false, &unused);      // the divisor, scaled
    sn[i] = __builtin_amdgcn_div_scale(x, y[i], true, &flag[i]);       // the dividend, scaled; its flag steers v_div_fmas
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) rcp[i] = __builtin_amdgcn_rcp(sd[i]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) t[i] = __builtin_fma(-sd[i], rcp[i], 1.0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) rcp[i] = __builtin_fma(rcp[i], t[i], rcp[i]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) t[i] = __builtin_fma(-sd[i], rcp[i], 1.0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) rcp[i] = __builtin_fma(rcp[i], t[i], rcp[i]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) u[i] = sn[i] * rcp[i];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) t[i] = __builtin_fma(-sd[i], u[i], sn[i]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) out[i] = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(t[i], rcp[i], u[i], flag[i]), y[i], x);
  __builtin_amdgcn_sched_barrier(0);
}

// The batch path's logistic, a = 1 / (1 + exp(x)) with x = -net clamped to [-690, 690] (oracle/fqi.c, deviation D5: beyond the clamp the
// reference's value is within 1e-299 of 0 or of 1), for N independent arguments.  Inside the clamp exp(x) is normal and below 2^996, so
// (i) 2^k is applied by one exact v_ldexp_f64 and the special ranges of exp need no selects, and (ii) the divisor y = 1 + e lies in
// [1, 2^997): for a numerator of 1 and such a divisor V_DIV_SCALE_F64 scales neither operand and raises no flag, V_DIV_FMAS_F64 is a plain
// fma and V_DIV_FIXUP_F64 returns its first operand -- the compiler's correctly rounded division sequence (pdiv_batch) reduces to v_rcp, two
// Newton steps and one residual correction, 7 instructions instead of 12, the same bits as `1.0 / y` on the host.  The clamp is C's
// fmin(fmax(x, -690), 690) -- v_max_f64, v_min_f64 -- so a NaN net input gives a = 1 here and in the oracle (compares and selects: 6
// instructions per unit instead of 2, 13.95 instead of 15.87 G).  12.7 -> 15.9 G sample-epochs/s on the bench (the same arithmetic beside
// a second, general path in one function was 20 % SLOWER than pexp_batch + pdiv_batch; the clamp makes the second path unnecessary).
template <int N>
__device__ __forceinline__ void plogistic_batch(const double (&xin)[N], double (&a)[N])
{
  double x[N], kd[N], r[N], q[N];
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = __builtin_fmin(__builtin_fmax(xin[i], -690.0), 690.0);      // (C's fmax / fmin: a NaN argument becomes -690)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) kd[i] = __builtin_rint(x[i] * 0x1.71547652b82fep+0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = __builtin_fma(-kd[i], 0x1.62e4200000000p-1, x[i]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = __builtin_fma(-kd[i], 0x1.fdf473de6af28p-22, r[i]);
  __builtin_amdgcn_sched_barrier(0);
  // the kernel 1 + r + r^2 q(r), q of degree 9: a Chebyshev fit of (e^r - 1 - r) / r^2 on |r| <= ln2 / 2, 0.07 ulp from exp before rounding
  // (oracle/fqi.c: logistic_exp; pexp_batch keeps the Taylor polynomial of degree 11 the other paths' parity is pinned on)
  const double c[10] = {0x1.af389ecfc4b9cp-26, 0x1.28917c89a43a7p-22, 0x1.71de0db2f6b19p-19, 0x1.a019b9149a41cp-16, 0x1.a01a01a7c2efep-13,
                        0x1.6c16c17889ef1p-10, 0x1.11111111109b5p-7, 0x1.5555555553d68p-5, 0x1.5555555555556p-3, 0x1.0000000000001p-1};
#pragma unroll
  for (int i = 0; i < N; ++i) q[i] = __builtin_fma(r[i], c[0], c[1]);
#pragma unroll
  for (int s = 2; s < 10; ++s)
  {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = __builtin_fma(r[i], q[i], c[s]);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) q[i] = 1.0 + __builtin_fma(r[i] * r[i], q[i], r[i]);
  __builtin_amdgcn_sched_barrier(0);
  double y[N], rcp[N], t[N];
#pragma unroll
  for (int i = 0; i < N; ++i) y[i] = 1. + __builtin_ldexp(q[i], (int)kd[i]);      // 1 + exp(x); |k| <= 996: the scaling is exact
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) rcp[i] = __builtin_amdgcn_rcp(y[i]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) t[i] = __builtin_fma(-y[i], rcp[i], 1.0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) rcp[i] = __builtin_fma(rcp[i], t[i], rcp[i]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) t[i] = __builtin_fma(-y[i], rcp[i], 1.0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) rcp[i] = __builtin_fma(rcp[i], t[i], rcp[i]);
  __builtin_amdgcn_sched_barrier(0);
  // (the scaled quotient u = sn * rcp with sn = 1 is rcp itself, exactly)  the residual of the quotient and its correction
#pragma unroll
  for (int i = 0; i < N; ++i) t[i] = __builtin_fma(-y[i], rcp[i], 1.0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = __builtin_fma(t[i], rcp[i], rcp[i]);
  __builtin_amdgcn_sched_barrier(0);
}

} // namespace grlx
