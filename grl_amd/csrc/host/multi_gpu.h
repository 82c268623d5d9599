// multi_gpu.h -- `grlxd -g N`: one process per GPU, the learning curve reduced over all of them with ONE RCCL all-reduce per run.
// The reference's model is experiment/multi (multi.cpp:44-75: clones of an experiment side by side, identities "@i"); here the clones of a
// rank are the replicas of its device context and the ranks are processes, as north_star asks ("host side stays C++ ... RCCL all-reduce").
#pragma once
#include <cstddef>
#include <string>

namespace grlx_host {

// A communicator over the ranks of one `grlxd -g N` invocation plus the little device memory the reduction needs.
struct CurveReducer {
  virtual ~CurveReducer() {}
  virtual int rank() const = 0;
  virtual int world() const = 0;
  virtual double *device_buffer(size_t n_doubles) = 0;                    // grows on demand; owned by the reducer
  virtual void all_reduce_sum(double *dev, size_t n_doubles) = 0;         // in place, on the reducer's stream; returns when done
  virtual void to_host(double *host, const double *dev, size_t n_doubles) = 0;
};

// rank 0 creates the ncclUniqueId and publishes it in `id_file` (written under another name, then renamed); the others wait for the file.
// Throws Exception on any HIP / RCCL error.
CurveReducer *make_rccl_reducer(int rank, int world, const std::string &id_file);

// The r-th entry of HIP_VISIBLE_DEVICES if the variable is set (a launcher may have narrowed the node already), else r.
std::string device_for_rank(int rank);

} // namespace grlx_host
