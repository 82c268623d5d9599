// grlxd.cpp -- deployer for the accelerated path, command-line compatible with the
// reference's `grld [-v] [-s seed] <yaml file> [yaml file...]` (base/src/deployer.cpp:38-150),
// plus -r replicas, -t trials (override), -l (3-column golden layout), -q (no rows on stdout), and
// -g N: one process per GPU (rank r on the r-th visible device, started before anything touches HIP), every rank running -r replicas
// (clones r * replicas ..., seeds and "@i" identities counted over the whole job), the learning curves reduced with one RCCL all-reduce
// per run (multi_gpu.h); rank 0 prints the rows and writes <output>-<run>-mean.txt.  The model is experiment/multi (multi.cpp:44-75).
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <vector>

#include "configurable.h"
#include "multi_gpu.h"
#include "objects.h"

using namespace grlx_host;

int main(int argc, char **argv)
{
  RunOptions opt;
  int trials_override = -1;
  int gpus = 0;
  int c;
  while ((c = getopt(argc, argv, "vs:r:t:lqc:g:")) != -1)
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
      case 'g': gpus = atoi(optarg); break;
      default: return 1;
    }
  }
  if (optind > argc - 1)
  {
    log(0, std::string("Usage: \n  ") + argv[0] + " [-v] [-s seed] [-r replicas] [-g gpus] [-t trials] [-l] [-q] <yaml file> [yaml file...]");
    return 1;
  }
  if (opt.seed == 0)
  { // deployer.cpp:75-83: seed 0 means "from the clock"; the accelerated path wants reproducible runs
    log(0, "seed 0 (time-based seeding) is not supported; pass -s <seed>");
    return 1;
  }
  std::string id_file;
  if (gpus < 0 || gpus > 64) { log(0, "-g: between 1 and 64 processes"); return 1; }
  if (gpus >= 1)
  { // one process per GPU, forked BEFORE anything initialises HIP; each child narrows itself to its device
    std::ostringstream idn;
    idn << "/tmp/grlxd-" << getpid() << ".ncclid";
    id_file = idn.str();
    opt.world = gpus;
    if (gpus > 1)
    {
      std::vector<pid_t> kids;
      int rank = -1;
      for (int r = 0; r < gpus; ++r)
      {
        const pid_t pid = fork();
        if (pid < 0) { log(0, "fork failed"); return 1; }
        if (pid == 0) { rank = r; break; }
        kids.push_back(pid);
      }
      if (rank < 0)
      { // the parent: wait for the ranks in whatever order they end; the first failure ends the others (a rank that waits for a dead peer
        // inside the communicator's rendezvous would wait for ever)
        int bad = 0;
        for (size_t left = kids.size(); left > 0; --left)
        {
          int st = 0;
          const pid_t done = waitpid(-1, &st, 0);
          if (done < 0) { bad++; break; }
          if (!WIFEXITED(st) || WEXITSTATUS(st) != 0)
          {
            if (bad++ == 0)
              for (pid_t k : kids)
                if (k != done) kill(k, SIGTERM);
          }
        }
        unlink(id_file.c_str());
        if (bad) log(0, std::to_string(bad) + " of " + std::to_string(gpus) + " ranks failed");
        return bad ? 1 : 0;
      }
      opt.rank = rank;
    }
    try { setenv("HIP_VISIBLE_DEVICES", device_for_rank(opt.rank).c_str(), 1); }
    catch (Exception &e) { log(0, e.what()); return 1; }
  }
  std::unique_ptr<CurveReducer> reducer;
  try
  {
    if (gpus >= 1)
    {
      reducer.reset(make_rccl_reducer(opt.rank, opt.world, id_file));
      opt.reducer = reducer.get();
      log(2, "rank " + std::to_string(opt.rank) + " of " + std::to_string(opt.world) + ": communicator ready");
    }
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
