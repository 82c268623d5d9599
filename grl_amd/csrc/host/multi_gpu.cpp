// multi_gpu.cpp -- the RCCL side of `grlxd -g N` (multi_gpu.h).  Plain C++ against the HIP runtime API and RCCL: no kernels here.
#include "multi_gpu.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <thread>

#include "configurable.h"

namespace grlx_host {

namespace {
void hip_check(hipError_t e, const char *what)
{
  if (e != hipSuccess) throw Exception(std::string("multi-GPU: ") + what + " failed: " + hipGetErrorString(e));
}
void nccl_check(ncclResult_t r, const char *what)
{
  if (r != ncclSuccess) throw Exception(std::string("multi-GPU: ") + what + " failed: " + ncclGetErrorString(r));
}

struct RcclReducer : CurveReducer {
  int rank_ = 0, world_ = 1;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  double *buf = nullptr;
  size_t cap = 0;
  ~RcclReducer() override
  {
    if (buf) (void)hipFree(buf);
    if (comm) (void)ncclCommDestroy(comm);
    if (stream) (void)hipStreamDestroy(stream);
  }
  int rank() const override { return rank_; }
  int world() const override { return world_; }
  double *device_buffer(size_t n) override
  {
    if (n > cap)
    {
      if (buf) hip_check(hipFree(buf), "hipFree");
      buf = nullptr;
      hip_check(hipMalloc((void **)&buf, n * sizeof(double)), "hipMalloc");
      cap = n;
    }
    return buf;
  }
  void all_reduce_sum(double *dev, size_t n) override
  {
    nccl_check(ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, comm, stream), "ncclAllReduce");
    hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
  }
  void to_host(double *host, const double *dev, size_t n) override
  {
    hip_check(hipMemcpy(host, dev, n * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy");
  }
};
} // namespace

std::string device_for_rank(int rank)
{
  const char *v = getenv("HIP_VISIBLE_DEVICES");
  if (!v || !*v) return std::to_string(rank);
  std::stringstream ss(v);
  std::string item;
  for (int i = 0; std::getline(ss, item, ','); ++i)
    if (i == rank) return item;
  throw Exception("multi-GPU: HIP_VISIBLE_DEVICES = '" + std::string(v) + "' has no entry for rank " + std::to_string(rank));
}

CurveReducer *make_rccl_reducer(int rank, int world, const std::string &id_file)
{
  if (world < 1 || rank < 0 || rank >= world) throw Exception("multi-GPU: bad rank / world");
  RcclReducer *r = new RcclReducer();
  r->rank_ = rank;
  r->world_ = world;
  try
  {
    int ndev = 0;
    hip_check(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
    if (ndev < 1) throw Exception("multi-GPU: no HIP device visible to rank " + std::to_string(rank));
    hip_check(hipSetDevice(0), "hipSetDevice");            // every rank sees exactly its own device (HIP_VISIBLE_DEVICES, set before HIP started)
    ncclUniqueId id;
    memset(&id, 0, sizeof(id));
    if (rank == 0)
    {
      nccl_check(ncclGetUniqueId(&id), "ncclGetUniqueId");
      if (world > 1)
      {
        const std::string tmp = id_file + ".tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(&id, sizeof(id), 1, f) != 1) { if (f) fclose(f); throw Exception("multi-GPU: cannot write " + tmp); }
        fclose(f);
        if (rename(tmp.c_str(), id_file.c_str()) != 0) throw Exception("multi-GPU: cannot publish " + id_file);
      }
    }
    else
    {
      bool have = false;
      for (int tries = 0; tries < 1200 && !have; ++tries)          // two minutes: rank 0 may still be paging its libraries in
      {
        FILE *f = fopen(id_file.c_str(), "rb");
        if (f)
        {
          have = fread(&id, sizeof(id), 1, f) == 1;
          fclose(f);
        }
        if (!have) std::this_thread::sleep_for(std::chrono::milliseconds(100));
      }
      if (!have) throw Exception("multi-GPU: rank " + std::to_string(rank) + " never saw the communicator id of rank 0 (" + id_file + ")");
    }
    nccl_check(ncclCommInitRank(&r->comm, world, id, rank), "ncclCommInitRank");
    hip_check(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking), "hipStreamCreate");
  }
  catch (...)
  {
    delete r;
    throw;
  }
  return r;
}

} // namespace grlx_host
