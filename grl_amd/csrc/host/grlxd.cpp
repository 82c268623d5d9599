// grlxd.cpp -- deployer for the accelerated path, command-line compatible with the
// reference's `grld [-v] [-s seed] <yaml file> [yaml file...]` (base/src/deployer.cpp:38-150),
// plus -r replicas, -t trials (override), -l (3-column golden layout), -q (no rows on stdout).
#include <unistd.h>

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

#include "configurable.h"
#include "objects.h"

using namespace grlx_host;

int main(int argc, char **argv)
{
  RunOptions opt;
  int trials_override = -1;
  int c;
  while ((c = getopt(argc, argv, "vs:r:t:lqc:")) != -1)
  {
    switch (c)
    {
      case 'v': log_verbosity++; break;
      case 's': opt.seed = atol(optarg); break;
      case 'r': opt.replicas = atoi(optarg); break;
      case 't': trials_override = atoi(optarg); break;
      case 'l': opt.legacy_rows = true; break;
      case 'q': opt.print_rows = false; break;
      case 'c': opt.table_log2_capacity = atoi(optarg); break;
      default: return 1;
    }
  }
  if (optind > argc - 1)
  {
    log(0, std::string("Usage: \n  ") + argv[0] + " [-v] [-s seed] [-r replicas] [-t trials] [-l] [-q] <yaml file> [yaml file...]");
    return 1;
  }
  if (opt.seed == 0)
  { // deployer.cpp:75-83: seed 0 means "from the clock"; the accelerated path wants reproducible runs
    log(0, "seed 0 (time-based seeding) is not supported; pass -s <seed>");
    return 1;
  }
  try
  {
    YamlNode root;
    for (; optind < argc; ++optind)
    {
      log(2, std::string("Loading configuration from '") + argv[optind] + "'");
      std::ifstream ifs(argv[optind]);
      if (!ifs) { log(0, std::string("Could not load configuration '") + argv[optind] + "'"); return 1; }
      std::stringstream ss;
      ss << ifs.rdbuf();
      merge_yaml(root, parse_yaml(ss.str()));
    }
    if (trials_override >= 0)
      for (auto &kv : root.children)
        if (kv.first == "experiment")
          for (auto &p : kv.second.children)
          {
            if (p.first == "trials") p.second.scalar = std::to_string(trials_override);
            if (p.first == "experiment")                       // experiment/multi { experiment: experiment/online_learning }
              for (auto &q : p.second.children)
                if (q.first == "trials") q.second.scalar = std::to_string(trials_override);
          }
    log(2, "Instantiating configuration");
    std::unique_ptr<Configurator> tree = instantiate(root);
    Configurator *expconf = tree->child("experiment");
    if (!expconf || !expconf->is_object) { log(0, "YAML configuration does not specify an experiment"); return 1; }
    Experiment *experiment = dynamic_cast<Experiment *>(expconf->object.get());
    if (!experiment) { log(0, "Specified experiment has wrong type"); return 1; }
    log(2, "Starting experiment");
    experiment->run(opt);
    log(2, "Cleaning up");
  }
  catch (Exception &e)
  {
    log(0, e.what());
    return 1;
  }
  return 0;
}
