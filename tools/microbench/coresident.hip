// Can a 384-register kernel (one wave per SIMD, the rollout kernels' shape) and a small second kernel be resident on the same SIMDs
// at once, and what does a hand-off between them through device memory cost?
//   hipcc --offload-arch=gfx950 -O3 -o coresident coresident.hip && ./coresident
// T ("table" role): 1024 blocks x 64 threads, ~380 registers, plays ping-pong with E through one 128-byte mailbox per T-wave group.
// E ("env server" role): 256 blocks x 64 threads, few registers; lane l of block b serves T-block 4b + l/16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((address_space(1))) unsigned int gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Mail { unsigned int ping, pong, pad[30]; };

__global__ __launch_bounds__(64) void t_kernel(Mail *mail, int rounds, unsigned long long *out, unsigned int spin_limit)
{
  // hold ~380 registers live across the loop
  double keep[150];
#pragma unroll
  for (int i = 0; i < 150; ++i) keep[i] = threadIdx.x * 1e-3 + i;
  Mail *m = mail + blockIdx.x;
  unsigned long long t0 = __builtin_readcyclecounter(), worst = 0, sum = 0;
  int done = 0;
  for (int r = 1; r <= rounds; ++r)
  {
    const unsigned long long a = __builtin_readcyclecounter();
    if (threadIdx.x == 0) __hip_atomic_store((gu32 *)&m->ping, (unsigned)r, RLX_AGENT);
    unsigned spins = 0;
    bool ok = true;
    if (threadIdx.x == 0)
      while (__hip_atomic_load((gu32 *)&m->pong, RLX_AGENT) != (unsigned)r) { if (++spins > spin_limit) { ok = false; break; } }
    ok = __shfl((int)ok, 0, 64) != 0;
    const unsigned long long b = __builtin_readcyclecounter();
    if (!ok) break;
    done = r;
    sum += b - a;
    if (b - a > worst) worst = b - a;
#pragma unroll
    for (int i = 0; i < 150; ++i) keep[i] = keep[i] * 1.0000001 + 1e-9;      // some arithmetic on the live set
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 150; ++i) s += keep[i];
  if (threadIdx.x == 0)
  {
    out[blockIdx.x * 4 + 0] = done;
    out[blockIdx.x * 4 + 1] = done ? sum / done : 0;
    out[blockIdx.x * 4 + 2] = worst;
    out[blockIdx.x * 4 + 3] = (unsigned long long)s + (__builtin_readcyclecounter() - t0);
  }
}

__global__ __launch_bounds__(64) void e_kernel(Mail *mail, int rounds, unsigned int spin_limit, unsigned int *seen)
{
  const int lane = threadIdx.x;
  Mail *m = mail + blockIdx.x * 4 + lane / 16;
  if (lane % 16 != 0) return;
  for (int r = 1; r <= rounds; ++r)
  {
    unsigned spins = 0;
    while (__hip_atomic_load((gu32 *)&m->ping, RLX_AGENT) != (unsigned)r) { if (++spins > spin_limit) return; }
    __hip_atomic_store((gu32 *)&m->pong, (unsigned)r, RLX_AGENT);
    if (r == 1) seen[blockIdx.x * 4 + lane / 16] = 1;
  }
}

int main(int argc, char **argv)
{
  const int nt = 1024, ne = 256, rounds = argc > 1 ? atoi(argv[1]) : 2000;
  Mail *mail; unsigned long long *out; unsigned int *seen;
  CHECK(hipMalloc(&mail, sizeof(Mail) * nt)); CHECK(hipMemset(mail, 0, sizeof(Mail) * nt));
  CHECK(hipMalloc(&out, sizeof(unsigned long long) * nt * 4)); CHECK(hipMemset(out, 0, sizeof(unsigned long long) * nt * 4));
  CHECK(hipMalloc(&seen, sizeof(unsigned) * nt)); CHECK(hipMemset(seen, 0, sizeof(unsigned) * nt));
  hipStream_t sa, sb; CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipFuncAttributes fa; CHECK(hipFuncGetAttributes(&fa, (const void *)t_kernel)); printf("t_kernel: %d registers\n", fa.numRegs);
  CHECK(hipFuncGetAttributes(&fa, (const void *)e_kernel)); printf("e_kernel: %d registers\n", fa.numRegs);
  for (int order = 0; order < 2; ++order)
  {
    CHECK(hipMemset(mail, 0, sizeof(Mail) * nt)); CHECK(hipMemset(out, 0, sizeof(unsigned long long) * nt * 4)); CHECK(hipDeviceSynchronize());
    const unsigned limit = 20000000u;       // bounded spins: ~ a second
    if (order == 0) { hipLaunchKernelGGL(e_kernel, dim3(ne), dim3(64), 0, sb, mail, rounds, limit, seen); hipLaunchKernelGGL(t_kernel, dim3(nt), dim3(64), 0, sa, mail, rounds, out, limit); }
    else { hipLaunchKernelGGL(t_kernel, dim3(nt), dim3(64), 0, sa, mail, rounds, out, limit); hipLaunchKernelGGL(e_kernel, dim3(ne), dim3(64), 0, sb, mail, rounds, limit, seen); }
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nt * 4);
    CHECK(hipMemcpy(h.data(), out, sizeof(unsigned long long) * nt * 4, hipMemcpyDeviceToHost));
    unsigned long long complete = 0, mean = 0, worst = 0;
    for (int b = 0; b < nt; ++b) { complete += h[b * 4] == (unsigned long long)rounds; mean += h[b * 4 + 1]; if (h[b * 4 + 2] > worst) worst = h[b * 4 + 2]; }
    printf("%s first: %llu of %d T-waves completed %d round trips; mean round trip %llu cycles, worst %llu\n", order == 0 ? "E" : "T", complete, nt, rounds, mean / nt, worst);
  }
  return 0;
}
