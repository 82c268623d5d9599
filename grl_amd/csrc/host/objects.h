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

// Experiment::run (experiment.h:44) -> learning curve of replica 0
struct OnlineLearningExperiment : Configurable {
  virtual std::vector<double> run(const RunOptions &opt) = 0;
};

} // namespace grlx_host
