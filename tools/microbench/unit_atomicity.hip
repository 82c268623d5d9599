// Are 16-byte units {value, sequence number} written by ONE lane with ONE global_store_dwordx4 and read with ONE dwordx4 load ever seen torn
// by a reader on another CU / XCD?  (The assumption of the environment server's mailboxes, grl_amd/csrc/grlx_env_server.h: a unit is either
// the old or the new one.)  Writer blocks rewrite their units as fast as they can with {double(k), k}; reader blocks load them -- with the
// global load (sc1) the server uses and with the buffer load (sc1) the rollout wave uses -- and count units whose two halves disagree.
//   hipcc --offload-arch=gfx950 -O3 -o unit_atomicity unit_atomicity.hip && ./unit_atomicity [seconds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((aligned(16))) Unit { double v; unsigned long long seq; };
typedef __attribute__((address_space(1))) unsigned int gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr int kUnitsPerBlock = 64;      // one unit per lane: 4 units per 64-byte line, 16 lines per block

__global__ __launch_bounds__(64) void writer(Unit *units, unsigned int *stop, unsigned long long *written, unsigned long long ticks)
{
  const unsigned long long deadline = __builtin_amdgcn_s_memrealtime() + ticks;     // 100 MHz: the exit every wave reaches
  Unit *p = units + blockIdx.x * kUnitsPerBlock + threadIdx.x;
  unsigned long long k = 0;
  for (;;)
  {
    ++k;
    const unsigned long long b = (unsigned long long)__double_as_longlong((double)k);
    u32x4 d;
    d.x = (unsigned)b; d.y = (unsigned)(b >> 32); d.z = (unsigned)k; d.w = (unsigned)(k >> 32);
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
    if ((k & 255u) == 0u && (__hip_atomic_load((gu32 *)stop, RLX_AGENT) != 0u || __builtin_amdgcn_s_memrealtime() > deadline)) break;
  }
  if (threadIdx.x == 0) written[blockIdx.x] = k;
}

__global__ __launch_bounds__(64) void reader(const Unit *units, int n_writer_blocks, unsigned int *stop, unsigned long long *out, int use_buffer, unsigned long long ticks)
{
  const unsigned long long deadline = __builtin_amdgcn_s_memrealtime() + ticks;
  // every reader block walks over all writer blocks' units, lane l reading unit l of the block it looks at
  unsigned long long torn = 0, reads = 0, backwards = 0, last = 0;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)units, 0, n_writer_blocks * kUnitsPerBlock * (int)sizeof(Unit), 0x00020000);
  int wb = blockIdx.x % n_writer_blocks;
  for (unsigned long long it = 0;; ++it)
  {
    const Unit *p = units + wb * kUnitsPerBlock + threadIdx.x;
    u32x4 d;
    if (use_buffer)
      d = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((wb * kUnitsPerBlock + (int)threadIdx.x) * sizeof(Unit)), 0, 16);
    else
      asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(d) : "v"(p) : "memory");
    const unsigned long long seq = ((unsigned long long)d.w << 32) | d.z;
    const double v = __longlong_as_double((long long)(((unsigned long long)d.y << 32) | d.x));
    ++reads;
    if (v != (double)seq) ++torn;
    if ((it & 7u) == 0u) { wb = (wb + 1) % n_writer_blocks; last = 0; }      // stay on a block for 8 reads: the sequence must not go backwards
    else { if (seq < last) ++backwards; last = seq; }
    if ((it & 255u) == 0u && (__hip_atomic_load((gu32 *)stop, RLX_AGENT) != 0u || __builtin_amdgcn_s_memrealtime() > deadline)) break;
  }
  atomicAdd(&out[0], reads);
  atomicAdd(&out[1], torn);
  atomicAdd(&out[2], backwards);
}

int main(int argc, char **argv)
{
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
  const int n_w = 256, n_r = 768;
  Unit *units; unsigned int *stop; unsigned long long *written, *out;
  CHECK(hipMalloc(&units, n_w * kUnitsPerBlock * sizeof(Unit)));
  CHECK(hipMemset(units, 0, n_w * kUnitsPerBlock * sizeof(Unit)));
  CHECK(hipMalloc(&stop, 4)); CHECK(hipMalloc(&written, n_w * 8)); CHECK(hipMalloc(&out, 3 * 8));
  for (int use_buffer = 0; use_buffer < 2; ++use_buffer)
  {
    CHECK(hipMemset(stop, 0, 4)); CHECK(hipMemset(out, 0, 24)); CHECK(hipMemset(written, 0, n_w * 8));
    hipStream_t sw, sr, sc;
    CHECK(hipStreamCreateWithFlags(&sw, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sr, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    const unsigned long long ticks = (unsigned long long)((seconds + 1.0) * 1e8);
    hipLaunchKernelGGL(writer, dim3(n_w), dim3(64), 0, sw, units, stop, written, ticks);
    hipLaunchKernelGGL(reader, dim3(n_r), dim3(64), 0, sr, units, n_w, stop, out, use_buffer, ticks);
    CHECK(hipGetLastError());
    // both kernels poll `stop` and leave by themselves a second after the measuring time (s_memrealtime deadline): set it from a third stream
    timespec ts = {(time_t)seconds, (long)((seconds - (long)seconds) * 1e9)};
    nanosleep(&ts, nullptr);
    unsigned int one = 1;
    CHECK(hipMemcpyAsync(stop, &one, 4, hipMemcpyHostToDevice, sc));
    CHECK(hipStreamSynchronize(sc)); CHECK(hipStreamSynchronize(sw)); CHECK(hipStreamSynchronize(sr));
    unsigned long long h[3]; std::vector<unsigned long long> w(n_w);
    CHECK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(w.data(), written, n_w * 8, hipMemcpyDeviceToHost));
    unsigned long long wsum = 0; for (auto x : w) wsum += x;
    printf("%s loads: %.3g unit stores by %d writer waves (x 64 lanes), %.3g unit loads by %d reader waves: %llu torn, %llu going backwards\n",
           use_buffer ? "buffer (sc1)" : "global (sc1)", (double)wsum * 64, n_w, (double)h[0], n_r, h[1], h[2]);
    CHECK(hipStreamDestroy(sw)); CHECK(hipStreamDestroy(sr)); CHECK(hipStreamDestroy(sc));
  }
  return 0;
}
