// objects.h -- what the deployer needs to know about the experiment object.
#pragma once
#include <cstdint>
#include <vector>
#include "configurable.h"

namespace grlx_host {

struct RunOptions {
  int64_t seed = 1;             // deployer -s (replica i uses seed + i)
  int replicas = 1;             // clones of the experiment (experiment/multi analogue)
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

// Experiment::run (experiment.h:44) -> learning curve of replica 0
struct OnlineLearningExperiment : Configurable {
  virtual std::vector<double> run(const RunOptions &opt) = 0;
};

} // namespace grlx_host
