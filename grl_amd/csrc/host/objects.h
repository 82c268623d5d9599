// objects.h -- what the deployer needs to know about the experiment object.
#pragma once
#include <cstdint>
#include <vector>
#include "configurable.h"

namespace grlx_host {

struct CurveReducer;            // multi_gpu.h
struct RunOptions {
  int64_t seed = 1;             // deployer -s (replica i uses seed + i)
  int replicas = 1;             // clones of the experiment (experiment/multi analogue)
  // `grlxd -g N`: this process is rank `rank` of `world`, one per GPU.  Its replicas are the clones rank * replicas .. + replicas - 1 of the
  // whole job (seeds and "@i" identities count over all ranks); after every run the ranks reduce their learning curves with ONE all-reduce
  // (reducer->all_reduce_sum over grlx_curve_stats' device buffer) and rank 0 writes <output>-<run>-mean.txt.
  int rank = 0, world = 1;
  CurveReducer *reducer = nullptr;
  int table_log2_capacity = 0;
  bool legacy_rows = false;     // 3-column rows as in the reference's committed golden files
  bool print_rows = true;
};

// Environment::step (environment.h:48-51 -> ModeledEnvironment::step, modeled.cpp:160-213) for a batch of instances:
// state[n][S] is advanced in place; obs[n][D], reward[n], terminal[n].  Forwards to grlx_env_step (HIP kernel).
struct Environment : Configurable {
  virtual void dims(int *state_dims, int *obs_dims) const = 0;
  virtual void step(double *state, const double *action, int n, double *obs, double *reward, int32_t *terminal) const = 0;
};

// Projector::project (projector.h:55-69 -> TileCodingProjector::_project, tile_coding.cpp:103-149) for a batch of
// inputs in[n][dims] -> out[n][tilings] reference slot indices.  Forwards to grlx_project (HIP kernel).
struct Projector : Configurable {
  virtual int n_tilings() const = 0;
  virtual int n_dims() const = 0;
  virtual void project(const double *in, int n, uint32_t *out) const = 0;
};

// Representation::read / write / update (representation.h:60-83 -> LinearRepresentation, linear.cpp:136-216) for rows of
// projections idx[n][16] (reference slot indices, as Projector::project returns them), applied in row order.  The parameter
// vector lives on the GPU: the object owns a one-replica context whose table it is -- drawn from srand48(seed) the way
// LinearRepresentation::configure draws it as the first user of a fresh process -- and forwards to grlx_read / grlx_write /
// grlx_update (HIP kernel table_op_kernel).  target[n] / delta[n]: one output per row (outputs = 1).
struct Representation : Configurable {
  virtual void reset(int64_t seed) = 0;
  virtual void read(const uint32_t *idx, int n, double *out) = 0;
  virtual void write(const uint32_t *idx, int n, const double *target, double alpha) = 0;
  virtual void update(const uint32_t *idx, int n, const double *delta) = 0;
};

// Experiment::run (experiment.h:44) -> learning curve of replica 0
struct Experiment : Configurable {
  virtual std::vector<double> run(const RunOptions &opt) = 0;
};
struct OnlineLearningExperiment : Experiment {};

// The per-step side of the plug-in API: the experiment's environment and agents as objects a caller steps itself -- the loop of
// OnlineLearningExperiment::run (online_learning.cpp:172-213) stays with the caller, one side (or both) runs on the GPU, every call
// serves all replicas of the experiment's device context (rows of [replicas]; `active`: NULL or a mask of the replicas taking part).
//   Environment::start / step  (environment.h:48-51)  -> grlx_env_start / grlx_env_advance
//   Agent::start / step / end  (agent.h:44-56)        -> grlx_agent_start / _step / _end; agent/td is the learning agent, agent/fixed the test agent
struct StepwiseEnvironment {
  virtual ~StepwiseEnvironment() {}
  virtual void start(int test, const int32_t *active, double *obs) = 0;
  virtual void step(const int32_t *active, const double *action, double *obs, double *reward, int32_t *terminal) = 0;
};
struct StepwiseAgent {
  virtual ~StepwiseAgent() {}
  virtual void start(const int32_t *active, const double *obs, double *action) = 0;
  virtual void step(const int32_t *active, double tau, const double *obs, const double *reward, const int32_t *terminal, double *action) = 0;
  virtual void end(const int32_t *active, double tau, const double *obs, const double *reward) = 0;
};
// experiment/online_learning as the owner of the device context its objects step on
struct StepwiseExperiment {
  virtual ~StepwiseExperiment() {}
  virtual void open(const RunOptions &opt) = 0;                 // instantiate on the GPU (grlx_create of the lowered graph)
  virtual void close() = 0;
  virtual int replicas() const = 0;
  virtual int obs_dims() const = 0;
  virtual int test_interval_of() const = 0;
  virtual StepwiseEnvironment *stepwise_environment() = 0;
  virtual StepwiseAgent *stepwise_agent() = 0;
  virtual StepwiseAgent *stepwise_test_agent() = 0;
};

} // namespace grlx_host
