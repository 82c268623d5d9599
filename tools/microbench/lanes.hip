// Does a wave64 VALU instruction cost less when only some 16-lane groups are active?
// Each wave runs a chain of independent+dependent f64 FMAs with `active` lanes enabled.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int ILP>
__global__ __launch_bounds__(64) void chain(double *out, int iters, int active)
{
  const int lane = threadIdx.x & 63;
  if (lane >= active) return;
  double a[ILP];
  for (int k = 0; k < ILP; ++k) a[k] = 1.0 + lane * 1e-3 + k;
  const double m = 1.0000001, c = 1e-9;
  for (int i = 0; i < iters; ++i)
  {
#pragma unroll
    for (int u = 0; u < 64; ++u)        // 64 x ILP FMAs per trip: the loop branch no longer matters
    {
#pragma unroll
      for (int k = 0; k < ILP; ++k) a[k] = __builtin_fma(a[k], m, c);
    }
  }
  double s = 0;
  for (int k = 0; k < ILP; ++k) s += a[k];
  out[blockIdx.x * 64 + lane] = s;
}

template <int ILP>
static void run(int waves, int iters)
{
  double *out;
  hipMalloc(&out, sizeof(double) * waves * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int active : {64, 48, 32, 16, 8, 1})
  {
    hipLaunchKernelGGL(chain<ILP>, dim3(waves), dim3(64), 0, 0, out, iters, active);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain<ILP>, dim3(waves), dim3(64), 0, 0, out, iters, active);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("ILP %d waves %5d active lanes %2d: %8.3f ms  (%.2f ns per FMA instruction per wave)\n", ILP, waves, active, ms,
           ms * 1e6 / ((double)iters * ILP * 64));
  }
  hipFree(out);
}

int main(int argc, char **argv)
{
  int iters = 4000;
  run<1>(1024, iters);
  run<4>(1024, iters);
  run<8>(1024, iters);
  run<4>(2048, iters);
  run<4>(4096, iters);
  return 0;
}
