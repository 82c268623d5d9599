// grlx_kernels.hip -- hand-written HIP kernels (gfx950) for grl's online-learning hot path.
//
// Layout of the fused rollout kernel
// ----------------------------------
//  * one wavefront (64 lanes) = 4 independent replicas x 16 lanes; lane j of a
//    replica owns tiling j of the hashed tile coding (T = 16).
//  * per-replica scalar work (RK4 integration of the dynamics, reward, RNG,
//    epsilon-greedy) is computed redundantly by the 16 lanes of the replica --
//    the lanes would otherwise idle, and it removes every broadcast.
//  * the eligibility trace lives in registers: lane j keeps, for each of the
//    <= 10 trace entries, the table position and the AUTHORITATIVE weight of tiling j
//    (write-back: stored when the slot leaves the trace; grlx_update.h).
//  * weights live in a per-replica open-addressing table in HBM (64-byte buckets of
//    four entries, grlx_table.h), lazily initialised to the value the reference's
//    8,388,608-draw initialisation gives that slot (LCG jump-ahead, grlx_rng.h).
//  * the TD update of a step is applied one pass later, under the next step's loads.
//  * sums over the 16 tilings are taken in the reference's serial order
//    (linear.cpp:147-151) through a 4-way interleaved LDS tile, so Q-values are
//    bit-identical to a scalar run.
//
// Compiled with -ffp-contract=off: the reference's arithmetic is plain IEEE
// double without fused multiply-add (x86-64 baseline); only grlx_math.h fuses.
#include <cstdio>
#include <cstdlib>
#include "grlx_internal.h"
#include "grlx_math.h"
#include "grlx_rng.h"
#include "grlx_tile.h"
#include "grlx_table.h"
#include "grlx_envs.h"
#include "grlx_policy.h"

namespace grlx {


// ------------------------------------------------------- fused rollout -----
// LDS tile shared by the 4 replicas of the wave; index (row*16 + tiling)*4 + group
// makes both the per-lane writes and the 4 simultaneous broadcast reads conflict-free.
#define SHW(row, k, g) sh_w[(((row) * 16 + (k)) << 2) + (g)]


} // namespace grlx

#include "grlx_update.h"
#include "grlx_env_server.h"
#include "grlx_env_server_wide.h"
#include "grlx_rollout.h"
#include "grlx_rollout_wide.h"
#include "grlx_rollout_ac.h"
#include "grlx_rollout_ac_wide.h"
#include "grlx_rollout_qv.h"
#include "grlx_rollout_acc.h"
#include "grlx_rollout_tgt.h"
#include "grlx_step.h"

namespace grlx {

// ------------------------------------------------------ register poisoning ---
// Every vector register (v0..v255, a0..a255: one wave owns a SIMD's whole file) and the SGPRs a kernel may be given
// are set to `pattern`.  The clobber lists make the compiler declare all of them for this kernel; the writes
// themselves are assembler repeat blocks.
#define GRLX_D10(p) p "0", p "1", p "2", p "3", p "4", p "5", p "6", p "7", p "8", p "9"
#define GRLX_D100(p) GRLX_D10(p "0"), GRLX_D10(p "1"), GRLX_D10(p "2"), GRLX_D10(p "3"), GRLX_D10(p "4"), GRLX_D10(p "5"), GRLX_D10(p "6"), GRLX_D10(p "7"), GRLX_D10(p "8"), GRLX_D10(p "9")
#define GRLX_ALL256(p) GRLX_D10(p), GRLX_D10(p "1"), GRLX_D10(p "2"), GRLX_D10(p "3"), GRLX_D10(p "4"), GRLX_D10(p "5"), GRLX_D10(p "6"), GRLX_D10(p "7"), GRLX_D10(p "8"), GRLX_D10(p "9"), \
  GRLX_D100(p "1"), GRLX_D10(p "20"), GRLX_D10(p "21"), GRLX_D10(p "22"), GRLX_D10(p "23"), GRLX_D10(p "24"), p "250", p "251", p "252", p "253", p "254", p "255"
__global__ __launch_bounds__(64) void poison_registers_kernel(uint32_t pattern)
{
  asm volatile(
      "v_mov_b32 v0, %0\n"
      ".set grlx_i, 0\n"
      ".rept 256\n"
      "  v_accvgpr_write_b32 a[grlx_i], v0\n"
      "  .set grlx_i, grlx_i + 1\n"
      ".endr\n"
      ".set grlx_i, 1\n"
      ".rept 255\n"
      "  v_mov_b32 v[grlx_i], %0\n"
      "  .set grlx_i, grlx_i + 1\n"
      ".endr\n"
      ".set grlx_i, 20\n"
      ".rept 12\n"                      // s20..s31; s32..s34 are the ABI's stack / frame / base pointer registers: not clobbered
      "  s_mov_b32 s[grlx_i], %0\n"
      "  .set grlx_i, grlx_i + 1\n"
      ".endr\n"
      ".set grlx_i, 35\n"
      ".rept 65\n"                      // s35..s99
      "  s_mov_b32 s[grlx_i], %0\n"
      "  .set grlx_i, grlx_i + 1\n"
      ".endr\n"
      :: "s"(pattern) : "memory", GRLX_ALL256("v"), GRLX_ALL256("a"), GRLX_D10("s2"), "s30", "s31", "s35", "s36", "s37", "s38", "s39", GRLX_D10("s4"), GRLX_D10("s5"),
         GRLX_D10("s6"), GRLX_D10("s7"), GRLX_D10("s8"), GRLX_D10("s9"));
}
#undef GRLX_ALL256
#undef GRLX_D100
#undef GRLX_D10

__global__ void set_u32_kernel(uint32_t *p, uint32_t v) { *p = v; }

hipError_t launch_poison_registers(uint32_t pattern, hipStream_t stream)
{ // one wave fills a SIMD; several rounds over the chip's SIMDs so that none is missed
  hipLaunchKernelGGL(poison_registers_kernel, dim3(8192), dim3(64), 0, stream, pattern);
  return hipGetLastError();
}

// ------------------------------------------------------------- launchers ---

hipError_t launch_rollout_ac(const DevParams &P, int n_trials, hipStream_t stream, int *variant)
{
  const bool taps = P.tap_replica >= 0 && P.tap_capacity > 0;          // recorded by the in-place instantiation
  if (variant) *variant = taps ? GRLX_KERNEL_IN_PLACE : GRLX_KERNEL_GENERIC;
  int waves = (P.n_replicas + kReplicasPerWave - 1) / kReplicasPerWave;
  if (!taps && P.replicas_per_wave >= 8)
  { // two (four) sub-batches per wave share one environment phase (grlx_rollout_ac_wide.h); at most wave_limit waves, the
    // replicas beyond their first load are handed out by a device-side counter as slots fall idle
    int R = P.replicas_per_wave;
    int all_waves = (P.n_replicas + R - 1) / R;
    int wwaves = all_waves < P.wave_limit ? all_waves : P.wave_limit;
    // 12 slots: every wave owns ceil(n / waves) consecutive replicas and rotates them through its slots (no device-wide queue); a batch
    // that would give a wave more than kAcOwnedMax runs in the 8-slot kernel
    if (R == 12 && (P.n_replicas + wwaves - 1) / wwaves > kAcOwnedMax)
    {
      R = 8;
      all_waves = (P.n_replicas + R - 1) / R;
      wwaves = all_waves < P.wave_limit ? all_waves : P.wave_limit;
    }
    if (R == 12)
    { // (with K = ceil(n / waves) replicas per wave the last waves may own none: launch only the ones that own some)
      const int k = (P.n_replicas + wwaves - 1) / wwaves;
      wwaves = (P.n_replicas + k - 1) / k;
    }
    hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, stream, P.queue, (uint32_t)wwaves * (uint32_t)R);
#define GRLX_LAUNCH_AC_WIDE(NB)                                                                                                  \
    if (P.env == GRLX_ENV_CART_POLE && !P.no_specialisation && SpecCartPoleAc::matches(P))                                       \
    {                                                                                                                            \
      if (variant) *variant = GRLX_KERNEL_SPECIALISED;                                                                           \
      hipLaunchKernelGGL((rollout_ac_wide_kernel<GRLX_ENV_CART_POLE, NB, SpecCartPoleAc>), dim3(wwaves), dim3(64), 0, stream, P, n_trials); \
    }                                                                                                                            \
    else if (P.env == GRLX_ENV_CART_POLE)                                                                                        \
      hipLaunchKernelGGL((rollout_ac_wide_kernel<GRLX_ENV_CART_POLE, NB, SpecNone>), dim3(wwaves), dim3(64), 0, stream, P, n_trials); \
    else if (P.env == GRLX_ENV_PENDULUM)                                                                                         \
      hipLaunchKernelGGL((rollout_ac_wide_kernel<GRLX_ENV_PENDULUM, NB, SpecNone>), dim3(wwaves), dim3(64), 0, stream, P, n_trials); \
    else                                                                                                                         \
      return hipErrorInvalidValue;
    if (R == 16) { GRLX_LAUNCH_AC_WIDE(4) }
    else if (R == 12) { GRLX_LAUNCH_AC_WIDE(3) }
    else { GRLX_LAUNCH_AC_WIDE(2) }
#undef GRLX_LAUNCH_AC_WIDE
    return hipGetLastError();
  }
  switch (P.env)
  {
    case GRLX_ENV_CART_POLE:
      if (taps)
        hipLaunchKernelGGL((rollout_ac_kernel<GRLX_ENV_CART_POLE, SpecNone, false>), dim3(waves), dim3(64), 0, stream, P, n_trials);
      else if (!P.no_specialisation && SpecCartPoleAc::matches(P))
      {
        if (variant) *variant = GRLX_KERNEL_SPECIALISED;
        hipLaunchKernelGGL((rollout_ac_kernel<GRLX_ENV_CART_POLE, SpecCartPoleAc, true>), dim3(waves), dim3(64), 0, stream, P, n_trials);
      }
      else
        hipLaunchKernelGGL((rollout_ac_kernel<GRLX_ENV_CART_POLE, SpecNone, true>), dim3(waves), dim3(64), 0, stream, P, n_trials);
      break;
    case GRLX_ENV_PENDULUM:
      if (taps)
        hipLaunchKernelGGL((rollout_ac_kernel<GRLX_ENV_PENDULUM, SpecNone, false>), dim3(waves), dim3(64), 0, stream, P, n_trials);
      else
        hipLaunchKernelGGL((rollout_ac_kernel<GRLX_ENV_PENDULUM, SpecNone, true>), dim3(waves), dim3(64), 0, stream, P, n_trials);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// The environment server works for the deferred-update pendulum instantiations of rollout_kernel with three actions (EXT in grlx_rollout.h),
// and -- its own kernel, grlx_env_server_wide.h -- for the wide kernels of the acrobot and the compass walker (8 replicas per wave, three actions).
// do a wave of the rollout kernel and a wave of its server fit on one SIMD together (512 registers)?  asked of the runtime once per pair
template <typename KA, typename KB>
static bool waves_fit_together(KA rollout, KB server)
{
  hipFuncAttributes a, b;
  if (hipFuncGetAttributes(&a, reinterpret_cast<const void *>(rollout)) != hipSuccess || hipFuncGetAttributes(&b, reinterpret_cast<const void *>(server)) != hipSuccess)
  {
    (void)hipGetLastError();
    return false;
  }
  const int gran = 8;                                  // allocation granule of the unified register file
  const int ra = (a.numRegs + gran - 1) / gran * gran, rb = (b.numRegs + gran - 1) / gran * gran;
  if (getenv("GRLX_ENV_SERVER_DEBUG"))
    fprintf(stderr, "grlx: rollout wave %d registers (%zu B scratch, %zu B LDS) + server wave %d registers (%zu B scratch): %s\n", a.numRegs, a.localSizeBytes,
            a.sharedSizeBytes, b.numRegs, b.localSizeBytes, ra + rb <= 512 ? "resident together" : "do not fit one SIMD");
  return ra + rb <= 512;
}
static bool env_server_wide(const DevParams &P)
{
  const bool inplace = P.diag_out != nullptr || (P.tap_replica >= 0 && P.tap_capacity > 0);
  const bool td = P.agent == GRLX_AGENT_SARSA || P.agent == GRLX_AGENT_Q || P.agent == GRLX_AGENT_EXPECTED_SARSA;
  if (!(!inplace && P.replicas_per_wave == 8 && (P.env == GRLX_ENV_ACROBOT || P.env == GRLX_ENV_COMPASS_WALKER) && P.A == 3 && td &&
        P.trace_kind != GRLX_TRACE_ACCUMULATING && P.target_interval == 0 && P.tile_safe == 0))
    return false;
  // The walker's server is built and tested but NOT the default: beside the 346-register rollout wave it has 160 registers, too few to hold
  // the sine's constants and the integrator's stages, and the code it becomes issues more vector instructions than the SIMD has slots left
  // (measured: 180-213 M env-steps/s with it against 220 M without).  GRLX_ENV_SERVER_WALKER=1 turns it on.
  if (P.env == GRLX_ENV_COMPASS_WALKER)
  {
    const char *w = getenv("GRLX_ENV_SERVER_WALKER");
    if (!w || atoi(w) == 0) return false;
  }
  // a server that cannot be resident beside its rollout wave would only be waited for in vain (8000 polls at the first step)
  static const bool fit_walker_spec = waves_fit_together(rollout_wide_served_kernel<GRLX_ENV_COMPASS_WALKER, SpecWalkerQ>, env_server_walker_kernel<SpecWalkerQ>);
  static const bool fit_acrobot_spec = waves_fit_together(rollout_wide_served_kernel<GRLX_ENV_ACROBOT, SpecAcrobotQ>, env_server_acrobot_pinned_kernel<SpecAcrobotQ>);
  static const bool fit_walker = waves_fit_together(rollout_wide_served_kernel<GRLX_ENV_COMPASS_WALKER, SpecNone>, env_server_walker_kernel<SpecNone>);
  static const bool fit_acrobot = waves_fit_together(rollout_wide_served_kernel<GRLX_ENV_ACROBOT, SpecNone>, env_server_acrobot_kernel<SpecNone>);
  if (!P.no_specialisation && SpecWalkerQ::matches(P)) return fit_walker_spec;
  if (!P.no_specialisation && SpecAcrobotQ::matches(P)) return fit_acrobot_spec;
  return P.env == GRLX_ENV_ACROBOT ? fit_acrobot : fit_walker;
}
bool env_server_serves(const DevParams &P)
{
  const bool inplace = P.diag_out != nullptr || (P.tap_replica >= 0 && P.tap_capacity > 0);
  if (env_server_wide(P)) return true;
  return !inplace && P.replicas_per_wave == 4 && P.env == GRLX_ENV_PENDULUM && P.A == 3 && P.agent != GRLX_AGENT_ADVANTAGE;
}
size_t env_server_mail_bytes(const DevParams &P)
{
  return env_server_wide(P) ? kWideMailBytes : env_server_serves(P) ? kEnvMailBytes : 0;
}

// One block per rollout wave, with the numeric parameters of the instantiation launch_rollout picks (same constants, same folding).
hipError_t launch_env_server(const DevParams &P, hipStream_t stream)
{
  if (!P.env_mail || !env_server_serves(P)) return hipErrorInvalidValue;
  if (env_server_wide(P))
  { // one block per wide rollout wave (8 replicas), with the numeric parameters of the instantiation launch_rollout picks
    const int wwaves = (P.n_replicas + 7) / 8;
    if (!P.no_specialisation && SpecWalkerQ::matches(P))
      hipLaunchKernelGGL((env_server_walker_kernel<SpecWalkerQ>), dim3(wwaves), dim3(64), 0, stream, P);
    else if (!P.no_specialisation && SpecAcrobotQ::matches(P))
      hipLaunchKernelGGL((env_server_acrobot_pinned_kernel<SpecAcrobotQ>), dim3(wwaves), dim3(64), 0, stream, P);
    else if (P.env == GRLX_ENV_ACROBOT)
      hipLaunchKernelGGL((env_server_acrobot_kernel<SpecNone>), dim3(wwaves), dim3(64), 0, stream, P);
    else
      hipLaunchKernelGGL((env_server_walker_kernel<SpecNone>), dim3(wwaves), dim3(64), 0, stream, P);
    return hipGetLastError();
  }
  const int waves = (P.n_replicas + kReplicasPerWave - 1) / kReplicasPerWave;
#define GRLX_LAUNCH_SERVER(AGENT)                                                                                      \
  if (!P.no_specialisation && SpecPendulumTcA<AGENT>::matches(P))                                                    \
  {                                                                                                                  \
    hipLaunchKernelGGL((env_server_kernel<GRLX_ENV_PENDULUM, 3, SpecPendulumTcA<AGENT>>), dim3(waves), dim3(64), 0, stream, P); \
    return hipGetLastError();                                                                                        \
  }
  GRLX_LAUNCH_SERVER(GRLX_AGENT_SARSA)
  GRLX_LAUNCH_SERVER(GRLX_AGENT_Q)
  GRLX_LAUNCH_SERVER(GRLX_AGENT_EXPECTED_SARSA)
#undef GRLX_LAUNCH_SERVER
  hipLaunchKernelGGL((env_server_kernel<GRLX_ENV_PENDULUM, 3, SpecNone>), dim3(waves), dim3(64), 0, stream, P);
  return hipGetLastError();
}

hipError_t launch_rollout(const DevParams &P, int n_trials, hipStream_t stream, int *variant)
{
  if (variant) *variant = GRLX_KERNEL_GENERIC;
  int waves = (P.n_replicas + kReplicasPerWave - 1) / kReplicasPerWave;
  if (P.env_mail && env_server_wide(P))
  { // the wide kernels' environment server (launch_env_server picks the same numeric parameters)
    const int wwaves = (P.n_replicas + 7) / 8;
#define GRLX_LAUNCH_WSERVED(SPECQ)                                                                                     \
    if (!P.no_specialisation && SPECQ::matches(P))                                                                   \
    {                                                                                                                \
      if (variant) *variant = GRLX_KERNEL_SPECIALISED;                                                               \
      hipLaunchKernelGGL((rollout_wide_served_kernel<SPECQ::kEnv, SPECQ>), dim3(wwaves), dim3(64), 0, stream, P, n_trials); \
      return hipGetLastError();                                                                                      \
    }
    GRLX_LAUNCH_WSERVED(SpecWalkerQ)
    GRLX_LAUNCH_WSERVED(SpecAcrobotQ)
#undef GRLX_LAUNCH_WSERVED
    if (P.env == GRLX_ENV_ACROBOT)
      hipLaunchKernelGGL((rollout_wide_served_kernel<GRLX_ENV_ACROBOT, SpecNone>), dim3(wwaves), dim3(64), 0, stream, P, n_trials);
    else
      hipLaunchKernelGGL((rollout_wide_served_kernel<GRLX_ENV_COMPASS_WALKER, SpecNone>), dim3(wwaves), dim3(64), 0, stream, P, n_trials);
    return hipGetLastError();
  }
  if (P.env_mail)
  { // with the environment server (launch_env_server picks the same numeric parameters)
    if (!env_server_serves(P)) return hipErrorInvalidValue;
#define GRLX_LAUNCH_SERVED(AGENT)                                                                                      \
    if (!P.no_specialisation && SpecPendulumTcA<AGENT>::matches(P))                                                  \
    {                                                                                                                \
      if (variant) *variant = GRLX_KERNEL_SPECIALISED;                                                               \
      hipLaunchKernelGGL((rollout_served_kernel<3, SpecPendulumTcA<AGENT>>), dim3(waves), dim3(64), 0, stream, P, n_trials); \
      return hipGetLastError();                                                                                      \
    }
    GRLX_LAUNCH_SERVED(GRLX_AGENT_SARSA)
    GRLX_LAUNCH_SERVED(GRLX_AGENT_Q)
    GRLX_LAUNCH_SERVED(GRLX_AGENT_EXPECTED_SARSA)
#undef GRLX_LAUNCH_SERVED
    hipLaunchKernelGGL((rollout_served_kernel<3, SpecNone>), dim3(waves), dim3(64), 0, stream, P, n_trials);
    return hipGetLastError();
  }
  // stamps and per-step taps are recorded by the instantiation that updates in place
#ifdef GRLX_WIDE_STAMPS
  const bool inplace = P.tap_replica >= 0 && P.tap_capacity > 0;      // stamped wide build: diag_out feeds the wide kernel
#else
  const bool inplace = P.diag_out != nullptr || (P.tap_replica >= 0 && P.tap_capacity > 0);
#endif
  if (P.diag_out && P.diag_deferred && P.env == GRLX_ENV_PENDULUM && P.A == 3 && !(P.tap_replica >= 0 && P.tap_capacity > 0))
  {
    if (variant) *variant = GRLX_KERNEL_GENERIC;
    hipLaunchKernelGGL((rollout_kernel<GRLX_ENV_PENDULUM, 3, true, SpecNone, true>), dim3(waves), dim3(64), 0, stream, P, n_trials);
    return hipGetLastError();
  }
  if (P.tap_deferred && P.tap_replica >= 0 && P.tap_capacity > 0 && !P.diag_out && P.agent != GRLX_AGENT_ADVANTAGE)
  { // per-step records of the production ordering (tests): generic instantiation, deferred update, taps
    if (variant) *variant = GRLX_KERNEL_GENERIC;
#define GRLX_LAUNCH_TAPDEF(ENVID, NACT)                                                                               \
    if (P.env == ENVID && P.A == NACT)                                                                              \
    {                                                                                                               \
      hipLaunchKernelGGL((rollout_kernel<ENVID, NACT, false, SpecNone, true, false, true>), dim3(waves), dim3(64), 0, stream, P, n_trials); \
      return hipGetLastError();                                                                                     \
    }
    GRLX_LAUNCH_TAPDEF(GRLX_ENV_PENDULUM, 3)
    GRLX_LAUNCH_TAPDEF(GRLX_ENV_PENDULUM, 5)
    GRLX_LAUNCH_TAPDEF(GRLX_ENV_ACROBOT, 3)
#undef GRLX_LAUNCH_TAPDEF
    return hipErrorInvalidValue;
  }
#define GRLX_LAUNCH(ENVID, NACT)                                                                              \
  if (P.env == ENVID && P.A == NACT)                                                                        \
  {                                                                                                         \
    if (variant && inplace) *variant = GRLX_KERNEL_IN_PLACE;                                                \
    if (inplace)                                                                                            \
      hipLaunchKernelGGL((rollout_kernel<ENVID, NACT, true, SpecNone>), dim3(waves), dim3(64), 0, stream, P, n_trials); \
    else                                                                                                    \
      hipLaunchKernelGGL((rollout_kernel<ENVID, NACT, false, SpecNone>), dim3(waves), dim3(64), 0, stream, P, n_trials); \
    return hipGetLastError();                                                                               \
  }
  if (P.agent == GRLX_AGENT_ADVANTAGE)
  { // advantage learning: its own in-place instantiation (taps included)
    if (variant) *variant = GRLX_KERNEL_IN_PLACE;
#define GRLX_LAUNCH_ADV(ENVID, NACT)                                                                                  \
    if (P.env == ENVID && P.A == NACT)                                                                              \
    {                                                                                                               \
      hipLaunchKernelGGL((rollout_kernel<ENVID, NACT, true, SpecNone, false, true>), dim3(waves), dim3(64), 0, stream, P, n_trials); \
      return hipGetLastError();                                                                                     \
    }
    GRLX_LAUNCH_ADV(GRLX_ENV_PENDULUM, 3)
    GRLX_LAUNCH_ADV(GRLX_ENV_ACROBOT, 3)
#undef GRLX_LAUNCH_ADV
    return hipErrorInvalidValue;
  }
  if (!inplace && P.replicas_per_wave == 32)
  { // 30 or more compass walkers per SIMD: EIGHT sub-batches per wave (two lanes per replica in the environment phase: the walker's two sines);
    // the sub-batches beyond the fourth park in P.park
    const int wwaves = (P.n_replicas + 31) / 32;
    if (!P.park || P.env != GRLX_ENV_COMPASS_WALKER || P.A != 3) return hipErrorInvalidValue;
    if (!P.no_specialisation && SpecWalkerQ::matches(P))
    {
      if (variant) *variant = GRLX_KERNEL_SPECIALISED;
      hipLaunchKernelGGL((rollout_wide_kernel<GRLX_ENV_COMPASS_WALKER, 3, 8, SpecWalkerQ>), dim3(wwaves), dim3(64), 0, stream, P, n_trials);
      return hipGetLastError();
    }
    hipLaunchKernelGGL((rollout_wide_kernel<GRLX_ENV_COMPASS_WALKER, 3, 8, SpecNone>), dim3(wwaves), dim3(64), 0, stream, P, n_trials);
    return hipGetLastError();
  }
  if (!inplace && P.replicas_per_wave == 16)
  { // 15 or more replicas per SIMD: FOUR sub-batches per wave share one environment phase (E + 4 T per 16 replicas instead of 2 (E + 2 T));
    // instantiated where the environment phase is half of a pass: the acrobot and the compass walker with three actions
    const int wwaves = (P.n_replicas + 15) / 16;
#define GRLX_LAUNCH_WIDE4_SPECQ(SPECQ)                                                                                \
    if (!P.no_specialisation && SPECQ::matches(P))                                                                  \
    {                                                                                                               \
      if (variant) *variant = GRLX_KERNEL_SPECIALISED;                                                              \
      hipLaunchKernelGGL((rollout_wide_kernel<SPECQ::kEnv, 3, 4, SPECQ>), dim3(wwaves), dim3(64), 0, stream, P, n_trials); \
      return hipGetLastError();                                                                                     \
    }
    GRLX_LAUNCH_WIDE4_SPECQ(SpecWalkerQ)
    GRLX_LAUNCH_WIDE4_SPECQ(SpecAcrobotQ)
#undef GRLX_LAUNCH_WIDE4_SPECQ
    if (P.env == GRLX_ENV_ACROBOT && P.A == 3)
    {
      hipLaunchKernelGGL((rollout_wide_kernel<GRLX_ENV_ACROBOT, 3, 4, SpecNone>), dim3(wwaves), dim3(64), 0, stream, P, n_trials);
      return hipGetLastError();
    }
    if (P.env == GRLX_ENV_COMPASS_WALKER && P.A == 3)
    {
      hipLaunchKernelGGL((rollout_wide_kernel<GRLX_ENV_COMPASS_WALKER, 3, 4, SpecNone>), dim3(wwaves), dim3(64), 0, stream, P, n_trials);
      return hipGetLastError();
    }
    return hipErrorInvalidValue;
  }
  if (!inplace && P.replicas_per_wave == 8)
  { // more replicas than 4 x SIMDs: two sub-batches per wave share one environment phase (grlx_rollout_wide.h)
    const int wwaves = (P.n_replicas + 7) / 8;
#define GRLX_LAUNCH_WIDE(ENVID, NACT)                                                                                 \
    if (P.env == ENVID && P.A == NACT)                                                                              \
    {                                                                                                               \
      hipLaunchKernelGGL((rollout_wide_kernel<ENVID, NACT, 2, SpecNone>), dim3(wwaves), dim3(64), 0, stream, P, n_trials); \
      return hipGetLastError();                                                                                     \
    }
#define GRLX_LAUNCH_WIDE_SPEC(AGENT)                                                                                  \
    if (!P.no_specialisation && SpecPendulumTcA<AGENT>::matches(P))                                                 \
    {                                                                                                               \
      if (variant) *variant = GRLX_KERNEL_SPECIALISED;                                                              \
      hipLaunchKernelGGL((rollout_wide_kernel<GRLX_ENV_PENDULUM, 3, 2, SpecPendulumTcA<AGENT>>), dim3(wwaves), dim3(64), 0, stream, P, n_trials); \
      return hipGetLastError();                                                                                     \
    }
    GRLX_LAUNCH_WIDE_SPEC(GRLX_AGENT_SARSA)
    GRLX_LAUNCH_WIDE_SPEC(GRLX_AGENT_Q)
#define GRLX_LAUNCH_WIDE_SPECQ(SPECQ)                                                                                 \
    if (!P.no_specialisation && SPECQ::matches(P))                                                                  \
    {                                                                                                               \
      if (variant) *variant = GRLX_KERNEL_SPECIALISED;                                                              \
      hipLaunchKernelGGL((rollout_wide_kernel<SPECQ::kEnv, 3, 2, SPECQ>), dim3(wwaves), dim3(64), 0, stream, P, n_trials); \
      return hipGetLastError();                                                                                     \
    }
    GRLX_LAUNCH_WIDE_SPECQ(SpecWalkerQ)
    GRLX_LAUNCH_WIDE_SPECQ(SpecAcrobotQ)
#undef GRLX_LAUNCH_WIDE_SPECQ
    GRLX_LAUNCH_WIDE(GRLX_ENV_PENDULUM, 3)
    GRLX_LAUNCH_WIDE(GRLX_ENV_PENDULUM, 5)
    GRLX_LAUNCH_WIDE(GRLX_ENV_ACROBOT, 3)
    GRLX_LAUNCH_WIDE(GRLX_ENV_CART_POLE, 3)
    GRLX_LAUNCH_WIDE(GRLX_ENV_COMPASS_WALKER, 3)
#undef GRLX_LAUNCH_WIDE
#undef GRLX_LAUNCH_WIDE_SPEC
    return hipErrorInvalidValue;
  }
  if (!inplace && !P.no_specialisation)
  { // compile-time specialised instantiations of the reference's cfg/pendulum/{sarsa,q}_tc.yaml family
#define GRLX_LAUNCH_SPEC(AGENT)                                                                                        \
    if (SpecPendulumTcA<AGENT>::matches(P))                                                                            \
    {                                                                                                                  \
      if (variant) *variant = GRLX_KERNEL_SPECIALISED;                                                                 \
      hipLaunchKernelGGL((rollout_kernel<GRLX_ENV_PENDULUM, 3, false, SpecPendulumTcA<AGENT>>), dim3(waves), dim3(64), 0, stream, P, n_trials); \
      return hipGetLastError();                                                                                        \
    }
    GRLX_LAUNCH_SPEC(GRLX_AGENT_SARSA)
    GRLX_LAUNCH_SPEC(GRLX_AGENT_Q)
    GRLX_LAUNCH_SPEC(GRLX_AGENT_EXPECTED_SARSA)
#undef GRLX_LAUNCH_SPEC
#define GRLX_LAUNCH_SPECQ(SPECQ)                                                                                       \
    if (SPECQ::matches(P))                                                                                             \
    {                                                                                                                  \
      if (variant) *variant = GRLX_KERNEL_SPECIALISED;                                                                 \
      hipLaunchKernelGGL((rollout_kernel<SPECQ::kEnv, 3, false, SPECQ>), dim3(waves), dim3(64), 0, stream, P, n_trials); \
      return hipGetLastError();                                                                                        \
    }
    GRLX_LAUNCH_SPECQ(SpecWalkerQ)
    GRLX_LAUNCH_SPECQ(SpecAcrobotQ)
#undef GRLX_LAUNCH_SPECQ
  }
  GRLX_LAUNCH(GRLX_ENV_PENDULUM, 3)
  GRLX_LAUNCH(GRLX_ENV_PENDULUM, 5)
  GRLX_LAUNCH(GRLX_ENV_ACROBOT, 3)
  GRLX_LAUNCH(GRLX_ENV_CART_POLE, 3)
  GRLX_LAUNCH(GRLX_ENV_COMPASS_WALKER, 3)
#undef GRLX_LAUNCH
  return hipErrorInvalidValue;
}

// -------------------------------------------------- fine-grained kernels ---
// Projector::project, batched: one lane per (row, tiling)
__global__ void project_kernel(TileParams tp, const double *in, int n, uint32_t *out)
{
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  int row = gid / tp.T, j = gid % tp.T;
  if (row >= n) return;
  out[(size_t)row * tp.T + j] = tile_slot_generic(tp, in + (size_t)row * tp.D, j);
}

hipError_t launch_project(const TileParams &tp, const double *in_dev, int n, uint32_t *out_dev, hipStream_t stream)
{
  long total = (long)n * tp.T;
  int blocks = (int)((total + 255) / 256);
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(project_kernel, dim3(blocks), dim3(256), 0, stream, tp, in_dev, n, out_dev);
  return hipGetLastError();
}

// Environment::step, batched: one lane per environment instance
template <int ENV>
__global__ void env_step_kernel(DevParams P, double *state, const double *action, int n,
                                double *obs, double *reward, int32_t *terminal, uint32_t *err)
{
  constexpr int S = Env<ENV>::S, D = Env<ENV>::D;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x[S], o[D], rw;
  int term;
  for (int k = 0; k < S; ++k) x[k] = state[(size_t)i * S + k];
  uint32_t st = 0;
  if (!Env<ENV>::in_domain(x)) st |= ST_DOMAIN;
  env_step<ENV>(P, x, action[i], o, rw, term, st);
  for (int k = 0; k < S; ++k) state[(size_t)i * S + k] = x[k];
  for (int k = 0; k < D; ++k) obs[(size_t)i * D + k] = o[k];
  reward[i] = rw;
  terminal[i] = term;
  bool bad = st != 0;
  for (int k = 0; k < S; ++k) bad = bad || (x[k] != x[k]);
  if (bad) atomicOr(err, ST_DOMAIN);
}

hipError_t launch_env_step(const DevParams &P, double *state_dev, const double *action_dev, int n,
                           double *obs_dev, double *reward_dev, int32_t *terminal_dev, uint32_t *err_dev, hipStream_t stream)
{
  int blocks = (n + 63) / 64;
  if (blocks == 0) return hipSuccess;
  switch (P.env)
  {
    case GRLX_ENV_PENDULUM:
      hipLaunchKernelGGL(env_step_kernel<GRLX_ENV_PENDULUM>, dim3(blocks), dim3(64), 0, stream, P, state_dev, action_dev, n,
                         obs_dev, reward_dev, terminal_dev, err_dev);
      break;
    case GRLX_ENV_ACROBOT:
      hipLaunchKernelGGL(env_step_kernel<GRLX_ENV_ACROBOT>, dim3(blocks), dim3(64), 0, stream, P, state_dev, action_dev, n,
                         obs_dev, reward_dev, terminal_dev, err_dev);
      break;
    case GRLX_ENV_CART_POLE:
      hipLaunchKernelGGL(env_step_kernel<GRLX_ENV_CART_POLE>, dim3(blocks), dim3(64), 0, stream, P, state_dev, action_dev, n,
                         obs_dev, reward_dev, terminal_dev, err_dev);
      break;
    case GRLX_ENV_COMPASS_WALKER:
      hipLaunchKernelGGL(env_step_kernel<GRLX_ENV_COMPASS_WALKER>, dim3(blocks), dim3(64), 0, stream, P, state_dev, action_dev, n,
                         obs_dev, reward_dev, terminal_dev, err_dev);
      break;
    case GRLX_ENV_CART_POLE_BALANCING:
      hipLaunchKernelGGL(env_step_kernel<GRLX_ENV_CART_POLE_BALANCING>, dim3(blocks), dim3(64), 0, stream, P, state_dev, action_dev, n,
                         obs_dev, reward_dev, terminal_dev, err_dev);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// Representation::read / write / update on rows applied in order (one wave; lane j = tiling j).
// op 0 read, 1 write(target, alpha), 2 update(delta)
__global__ __launch_bounds__(64) void table_op_kernel(DevParams P, int table, int op, const int32_t *replica, const uint32_t *idx, int n,
                                                      const double *arg, double alpha, double *out)
{
  __shared__ double   sh[64];
  __shared__ uint32_t shp[64];
  const int lane = threadIdx.x & 63;
  const int Tn = P.tile.T;
  uint32_t status = 0, inserted = 0, inserted_twin = 0;
  for (int i = 0; i < n; ++i)
  {
    const int r = replica[i];
    const Table tab = table_of(P, table, r);
    const bool act = lane < Tn;
    uint32_t slot = act ? idx[(size_t)i * Tn + lane] : 0u, pos = 0;
    const bool valid = act && slot != 0xFFFFFFFFu;     // invalid_index(): skipped by update (linear.cpp:207)
    double w = 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const LinearParams &lp = table == 1 ? P.lin_actor : P.lin;
    if (P.twin_tables)
    {
      const Table other = table_of(P, 1 - table, r);
      const LinearParams &olp = table == 1 ? P.lin : P.lin_actor;
      table_probe(tab, lp, P.states[r], table, valid, slot, pos, w, status, inserted, &other, &olp, &inserted_twin);
    }
    else
      table_probe(tab, lp, P.states[r], table, valid, slot, pos, w, status, inserted);
    sh[lane] = w;
    shp[lane] = valid ? pos : kInvalidPos;
    wave_sync();
    double s = 0;
    for (int k = 0; k < Tn; ++k) s += sh[k];
    s /= Tn;
    s = clampd(s, lp.out_min, lp.out_max);
    if (op == 0)
    {
      if (lane == 0) out[i] = s;
    }
    else
    {
      double d = (op == 1) ? alpha * (arg[i] - s) : arg[i];
      if (valid)
      { // sequential semantics for duplicate slots: c additions, highest lane's value is final
        uint32_t c = 0;
        for (int k = 0; k <= lane; ++k) c += (shp[k] == pos) ? 1u : 0u;
        bool last = true;
        for (int k = lane + 1; k < Tn; ++k) last = last && (shp[k] != pos);
        double v = w;
        for (uint32_t cc = 0; cc < c; ++cc)
          v = lp.limit ? clampd(v + d, lp.out_min, lp.out_max) : v + d;
        if (last) value_store(tab, pos, v);
      }
    }
    wave_sync();
    uint32_t ins = inserted;
    for (int off = 32; off > 0; off >>= 1) ins += __shfl_xor(ins, off, 64);
    if (lane == 0 && ins) P.states[r].n_slots[table] += ins;
    inserted = 0;
    if (P.twin_tables)
    {
      uint32_t it = inserted_twin;
      for (int off = 32; off > 0; off >>= 1) it += __shfl_xor(it, off, 64);
      if (lane == 0 && it) P.states[r].n_slots[1 - table] += it;
      inserted_twin = 0;
    }
    // sticky status of the replica this row worked on (not of the first row's)
    uint32_t st = status;
    for (int off = 32; off > 0; off >>= 1) st |= __shfl_xor(st, off, 64);
    if (lane == 0 && st) atomicOr(&P.states[r].status, st);
    status = 0;
  }
}

hipError_t launch_table_op(const DevParams &P, int table, int op, const int32_t *replica_dev, const uint32_t *idx_dev, int n,
                           const double *arg_dev, double alpha, double *out_dev, hipStream_t stream)
{
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(table_op_kernel, dim3(1), dim3(64), 0, stream, P, table, op, replica_dev, idx_dev, n, arg_dev, alpha, out_dev);
  return hipGetLastError();
}

// current weight of reference slots (lazy value when the slot was never touched; no insertion)
__global__ void get_weights_kernel(DevParams P, int table, int replica, const uint32_t *slots, int n, double *out)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Table tab = table_of(P, table, replica);
  uint32_t slot = slots[i];
  uint32_t b = table_home(tab, slot);
  double v = initial_weight(P.states[replica], table, table == 1 ? P.lin_actor : P.lin, slot);
  for (int it = 0; it < kMaxProbe; ++it)
  {
    const BucketRegs br = bucket_load(tab, b);
    uint32_t empty;
    const int way = bucket_find(br.k, slot, empty);
    if (way >= 0) { v = (way == 0) ? br.v[0] : (way == 1) ? br.v[1] : (way == 2) ? br.v[2] : br.v[3]; break; }
    if (empty != 0u) break;
    b = (b + 1u) & tab.bmask;
  }
  out[i] = v;
}

// {action: load}: replicas [first, first+count) take `image` as the initial value of every slot of
// `table` and forget what they had learned in it (their sparse tables are cleared by the caller)
__global__ void set_lazy_base_kernel(DevParams P, int table, int first, int count, const double *image)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  ReplicaState &rs = P.states[first + i];
  rs.lazy_base[table] = image;
  rs.n_slots[table] = 0u;
}

hipError_t launch_set_lazy_base(const DevParams &P, int table, int first, int count, const double *image_dev, hipStream_t stream)
{
  if (count <= 0) return hipSuccess;
  hipLaunchKernelGGL(set_lazy_base_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, P, table, first, count, image_dev);
  return hipGetLastError();
}

// dense export of a table: out[slot] for every reference slot (grid-stride)
__global__ void export_weights_kernel(DevParams P, int table, int replica, double *out)
{
  const Table tab = table_of(P, table, replica);
  const LinearParams &lp = table == 1 ? P.lin_actor : P.lin;
  const uint32_t memory = (uint32_t)(table == 1 ? P.tile_actor.memory : P.tile.memory);
  for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < memory; slot += gridDim.x * blockDim.x)
  {
    uint32_t b = table_home(tab, slot);
    double v = 0;
    bool found = false;
    for (int it = 0; it < kMaxProbe; ++it)
    {
      const BucketRegs br = bucket_load(tab, b);
      uint32_t empty;
      const int way = bucket_find(br.k, slot, empty);
      if (way >= 0) { v = (way == 0) ? br.v[0] : (way == 1) ? br.v[1] : (way == 2) ? br.v[2] : br.v[3]; found = true; break; }
      if (empty != 0u) break;
      b = (b + 1u) & tab.bmask;
    }
    if (!found) v = initial_weight(P.states[replica], table, lp, slot);
    out[slot] = v;
  }
}

hipError_t launch_export_weights(const DevParams &P, int table, int replica, double *out_dev, hipStream_t stream)
{
  hipLaunchKernelGGL(export_weights_kernel, dim3(2048), dim3(256), 0, stream, P, table, replica, out_dev);
  return hipGetLastError();
}

hipError_t launch_get_weights(const DevParams &P, int table, int replica, const uint32_t *slots_dev, int n, double *out_dev, hipStream_t stream)
{
  int blocks = (n + 255) / 256;
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(get_weights_kernel, dim3(blocks), dim3(256), 0, stream, P, table, replica, slots_dev, n, out_dev);
  return hipGetLastError();
}

__global__ void math_kernel(int op, const double *x, const double *y, int n, double *out)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i], r;
  switch (op)
  {
    case 0: r = psin_checked(v); break;
    case 1: r = pcos_checked(v); break;
    case 2: r = plog(v); break;
    case 3: r = pfmod(v, y[i]); break;
    case 5: r = div6(v); break;
    case 6: r = psin_s(v, sin_consts<false>()); break;       // small-angle-aware forms (whole waves of small arguments take the short path)
    case 7: r = pcos_s(v, sin_consts<false>()); break;
    case 8: { double sn, cs; psincos_s(v, sin_consts<false>(), sn, cs); r = sn + cs; break; }
    default: r = __builtin_sqrt(v); break;
  }
  out[i] = r;
}

hipError_t launch_math(int op, const double *x, const double *y, int n, double *out, hipStream_t stream)
{
  int blocks = (n + 255) / 256;
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(math_kernel, dim3(blocks), dim3(256), 0, stream, op, x, y, n, out);
  return hipGetLastError();
}

__global__ void rand48_at_kernel(uint64_t x0, const uint64_t *skip, int n, double *out)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = lcg_double(lcg_next(lcg_jump(x0, skip[i])));
}

hipError_t launch_rand48_at(uint64_t x0, const uint64_t *skip, int n, double *out, hipStream_t stream)
{
  int blocks = (n + 255) / 256;
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(rand48_at_kernel, dim3(blocks), dim3(256), 0, stream, x0, skip, n, out);
  return hipGetLastError();
}

// learning-curve statistics over the replicas of this GPU: one block per row, fixed-order tree
__global__ __launch_bounds__(256) void curve_stats_kernel(DevParams P, int first, double *out)
{
  __shared__ double s1[256], s2[256];
  const int row = first + blockIdx.x;
  __shared__ double s3[256];
  double a = 0, b = 0, n = 0;
  for (int r = threadIdx.x; r < P.n_replicas; r += 256)
    if ((uint32_t)row < P.states[r].rows)          // rows are ragged when replicas stop at a steps budget (grlx_run_steps)
    {
      double v = P.row_reward[(size_t)row * P.n_replicas + r];
      a += v;
      b += v * v;
      n += 1;
    }
  s1[threadIdx.x] = a;
  s2[threadIdx.x] = b;
  s3[threadIdx.x] = n;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1)
  {
    if ((int)threadIdx.x < off)
    {
      s1[threadIdx.x] += s1[threadIdx.x + off];
      s2[threadIdx.x] += s2[threadIdx.x + off];
      s3[threadIdx.x] += s3[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
  {
    out[blockIdx.x * 3 + 0] = s1[0];
    out[blockIdx.x * 3 + 1] = s2[0];
    out[blockIdx.x * 3 + 2] = s3[0];
  }
}

hipError_t launch_curve_stats(const DevParams &P, int first, int count, double *out_dev, hipStream_t stream)
{
  if (count == 0) return hipSuccess;
  hipLaunchKernelGGL(curve_stats_kernel, dim3(count), dim3(256), 0, stream, P, first, out_dev);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void step_counts_kernel(DevParams P, uint64_t *out)
{
  __shared__ unsigned long long a[256], b[256], c[256];
  unsigned long long la = 0, lb = 0, lc = 0;
  for (int r = threadIdx.x; r < P.n_replicas; r += 256)
  {
    la += (unsigned long long)P.states[r].ss;
    lb += P.states[r].test_steps;
    lc |= P.states[r].status;
  }
  a[threadIdx.x] = la; b[threadIdx.x] = lb; c[threadIdx.x] = lc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1)
  {
    if ((int)threadIdx.x < off)
    {
      a[threadIdx.x] += a[threadIdx.x + off];
      b[threadIdx.x] += b[threadIdx.x + off];
      c[threadIdx.x] |= c[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = a[0]; out[1] = b[0]; out[2] = c[0]; }
}

hipError_t launch_step_counts(const DevParams &P, uint64_t *out_dev, hipStream_t stream)
{
  hipLaunchKernelGGL(step_counts_kernel, dim3(1), dim3(256), 0, stream, P, out_dev);
  return hipGetLastError();
}

// ------------------------------------------------------------ table growth ---
__global__ __launch_bounds__(256) void max_load_kernel(DevParams P, int n_tables, uint32_t *out)
{
  __shared__ uint32_t m[256];
  uint32_t v = 0;
  for (int r = threadIdx.x; r < P.n_replicas; r += 256)
    for (int t = 0; t < n_tables; ++t) v = max(v, P.states[r].n_slots[t]);
  m[threadIdx.x] = v;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1)
  {
    if ((int)threadIdx.x < off) m[threadIdx.x] = max(m[threadIdx.x], m[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = m[0];
}

hipError_t launch_max_load(const DevParams &P, int n_tables, uint32_t *out_dev, hipStream_t stream)
{
  hipLaunchKernelGGL(max_load_kernel, dim3(1), dim3(256), 0, stream, P, n_tables, out_dev);
  return hipGetLastError();
}

// One wave per (table, replica): the 64 lanes walk the old buckets and claim ways of the new table with a compare-and-swap
// on the key word (the new table is at most a quarter full, probes are short).  Key word (slot, creating tiling, shared bit),
// claim word and value move unchanged; only the position changes.
__global__ __launch_bounds__(64) void rehash_kernel(DevParams P, int n_tables, Entry *new_tables, uint32_t new_logC, uint32_t *remap)
{
  const int r = blockIdx.x % P.n_replicas, table = blockIdx.x / P.n_replicas;
  if (table >= n_tables) return;
  // twin tables keep the same slot at the same position: table 0's placement decides, table 1's entries follow it
  const bool twin = P.twin_tables != 0 && n_tables == 2;
  if (twin && table == 1) return;
  const Table ot = table_of(P, table, r);
  const Table ot2 = table_of(P, twin ? 1 : table, r);
  Table nt2 = ot2;
  nt2.base = reinterpret_cast<Bucket *>(new_tables + ((((size_t)(twin ? 1 : table)) * (size_t)P.n_replicas + (size_t)r) << new_logC));
  nt2.bmask = (1u << (new_logC - 2)) - 1u;
  nt2.shift = 32u - (new_logC - 2);
  Table nt;
  nt.base = reinterpret_cast<Bucket *>(new_tables + (((size_t)table * (size_t)P.n_replicas + (size_t)r) << new_logC));
  nt.bmask = (1u << (new_logC - 2)) - 1u;
  nt.shift = 32u - (new_logC - 2);
  uint32_t *map = (remap && table == 0) ? remap + ((size_t)r << P.logC) : nullptr;
  const uint32_t n_buckets = 1u << (P.logC - 2);
  for (uint32_t b = threadIdx.x; b < n_buckets; b += 64)
  {
    const Bucket ob = ot.base[b];
    for (int way = 0; way < 4; ++way)
    {
      const uint32_t kw = ob.key[way];
      if (map) map[b * 4 + way] = kInvalidPos;
      if ((kw & kKeyMask) == 0u) continue;
      uint32_t nb = table_home(nt, (kw & kKeyMask) - 1u);
      uint32_t at = kInvalidPos;
      for (int it = 0; it < kMaxProbe && at == kInvalidPos; ++it)
      {
        for (int w = 0; w < 4; ++w)
          if (atomicCAS(&nt.base[nb].key[w], 0u, kw) == 0u) { at = (nb << 2) | (uint32_t)w; break; }
        if (at == kInvalidPos) nb = (nb + 1u) & nt.bmask;
      }
      if (at == kInvalidPos) { atomicOr(&P.states[r].status, ST_TABLE_FULL); continue; }
      nt.base[at >> 2].aux[at & 3u] = ob.aux[way];
      nt.base[at >> 2].val[at & 3u] = ob.val[way];
      if (twin)
      {
        nt2.base[at >> 2].key[at & 3u] = kw;
        nt2.base[at >> 2].aux[at & 3u] = ot2.base[b].aux[way];
        nt2.base[at >> 2].val[at & 3u] = ot2.base[b].val[way];
      }
      if (map) map[b * 4 + way] = at;
    }
  }
}

// setParams() on a table whose entries must stay where they are (twin tables): every existing entry takes the image's value of its slot
__global__ __launch_bounds__(64) void reload_entries_kernel(DevParams P, int table, int first_replica, const double *image)
{
  const int r = first_replica + blockIdx.x;
  const Table t = table_of(P, table, r);
  const uint32_t n_buckets = 1u << (P.logC - 2);
  for (uint32_t b = threadIdx.x; b < n_buckets; b += 64)
    for (int way = 0; way < 4; ++way)
    {
      const uint32_t kw = t.base[b].key[way] & kKeyMask;
      if (kw != 0u) t.base[b].val[way] = image[kw - 1u];
    }
}

hipError_t launch_reload_entries(const DevParams &P, int table, int first_replica, int n_replicas, const double *image_dev, hipStream_t stream)
{
  if (n_replicas <= 0) return hipSuccess;
  hipLaunchKernelGGL(reload_entries_kernel, dim3(n_replicas), dim3(64), 0, stream, P, table, first_replica, image_dev);
  return hipGetLastError();
}

hipError_t launch_rehash(const DevParams &P, int n_tables, Entry *new_tables, uint32_t new_logC, uint32_t *remap_dev, hipStream_t stream)
{
  hipLaunchKernelGGL(rehash_kernel, dim3(P.n_replicas * n_tables), dim3(64), 0, stream, P, n_tables, new_tables, new_logC, remap_dev);
  return hipGetLastError();
}

// positions kept outside the tables: the actor-critic's persisted critic trace and the target network's values
__global__ __launch_bounds__(64) void remap_positions_kernel(DevParams P, const uint32_t *remap, uint32_t new_logC, double *new_tvals)
{
  const int r = blockIdx.x;
  const uint32_t *map = remap + ((size_t)r << P.logC);
  if (P.trace_state)
  {
    uint32_t *ts = P.trace_state + (size_t)r * 16 * kMaxTrace * 2;
    for (int k = threadIdx.x; k < 16 * kMaxTrace; k += 64)
    {
      const uint32_t pos = ts[k * 2];
      if (pos != kInvalidPos) ts[k * 2] = map[pos & ((1u << P.logC) - 1u)];
    }
  }
  if (P.tvals && new_tvals)
  {
    const double *otv = P.tvals + ((size_t)r << P.logC);
    double *ntv = new_tvals + ((size_t)r << new_logC);
    for (uint32_t p = threadIdx.x; p < (1u << P.logC); p += 64)
      if (map[p] != kInvalidPos) ntv[map[p]] = otv[p];
  }
}

hipError_t launch_remap_positions(const DevParams &P, const uint32_t *remap_dev, uint32_t new_logC, double *new_tvals, hipStream_t stream)
{
  hipLaunchKernelGGL(remap_positions_kernel, dim3(P.n_replicas), dim3(64), 0, stream, P, remap_dev, new_logC, new_tvals);
  return hipGetLastError();
}

} // namespace grlx
