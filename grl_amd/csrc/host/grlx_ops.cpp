// grlx_ops.cpp -- the plug-in interfaces of the host layer, driven from the command line (parity checks, mixed graphs):
//   grlx_ops project <yaml> <path of a projector/tile_coding> <input file>     Projector::project for every input row
//   grlx_ops envstep <yaml> <path of an environment/modeled> <input file>      Environment::step for every (state, action) row
//   grlx_ops stepwise <yaml> <path of an experiment/online_learning> <input>   the experiment's loop on the host over per-step objects
// Rows are whitespace-separated numbers; results are printed with 17 significant digits (round-trip exact).
// The objects are instantiated from the reference's yaml exactly as grlxd does; the work happens in the HIP kernels
// behind grlx_project / grlx_env_step.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "configurable.h"
#include "objects.h"

using namespace grlx_host;

static std::vector<std::vector<double>> read_rows(const char *file)
{
  std::ifstream ifs(file);
  if (!ifs) throw Exception(std::string("cannot read '") + file + "'");
  std::vector<std::vector<double>> rows;
  std::string line;
  while (std::getline(ifs, line))
  {
    std::istringstream iss(line);
    std::vector<double> row;
    double v;
    while (iss >> v) row.push_back(v);
    if (!row.empty()) rows.push_back(row);
  }
  return rows;
}

int main(int argc, char **argv)
{
  if (argc != 5)
  {
    std::cerr << "usage: " << argv[0] << " project|envstep|represent|stepwise <yaml> <object path> <input file>" << std::endl;
    return 1;
  }
  try
  {
    std::ifstream ifs(argv[2]);
    if (!ifs) throw Exception(std::string("Could not load configuration '") + argv[2] + "'");
    std::stringstream ss;
    ss << ifs.rdbuf();
    YamlNode root;
    merge_yaml(root, parse_yaml(ss.str()));
    std::unique_ptr<Configurator> tree = instantiate(root);
    Configurator *node = tree->find(argv[3]);
    if (!node || !node->ptr()) throw Exception(std::string(argv[3]) + ": does not name an object");
    const std::vector<std::vector<double>> rows = read_rows(argv[4]);
    const int n = (int)rows.size();
    if (std::string(argv[1]) == "project")
    {
      const Projector *p = dynamic_cast<const Projector *>(node->ptr());
      if (!p) throw Exception(std::string(argv[3]) + ": not a projector");
      const int D = p->n_dims(), T = p->n_tilings();
      std::vector<double> in((size_t)n * D);
      for (int i = 0; i < n; ++i)
      {
        if ((int)rows[i].size() != D) throw Exception("input row does not match the projector's resolution vector");
        for (int k = 0; k < D; ++k) in[(size_t)i * D + k] = rows[i][k];
      }
      std::vector<uint32_t> out((size_t)n * T);
      p->project(in.data(), n, out.data());
      for (int i = 0; i < n; ++i)
      {
        for (int k = 0; k < T; ++k) printf("%s%u", k ? " " : "", out[(size_t)i * T + k]);
        printf("\n");
      }
    }
    else if (std::string(argv[1]) == "envstep")
    {
      const Environment *e = dynamic_cast<const Environment *>(node->ptr());
      if (!e) throw Exception(std::string(argv[3]) + ": not an environment");
      int S = 0, D = 0;
      e->dims(&S, &D);
      std::vector<double> state((size_t)n * S), action(n), obs((size_t)n * D), reward(n);
      std::vector<int32_t> terminal(n);
      for (int i = 0; i < n; ++i)
      {
        if ((int)rows[i].size() != S + 1) throw Exception("input row must hold the state followed by the action");
        for (int k = 0; k < S; ++k) state[(size_t)i * S + k] = rows[i][k];
        action[i] = rows[i][S];
      }
      e->step(state.data(), action.data(), n, obs.data(), reward.data(), terminal.data());
      for (int i = 0; i < n; ++i)
      {
        for (int k = 0; k < S; ++k) printf("%.17g ", state[(size_t)i * S + k]);
        for (int k = 0; k < D; ++k) printf("%.17g ", obs[(size_t)i * D + k]);
        printf("%.17g %d\n", reward[i], terminal[i]);
      }
    }
    else if (std::string(argv[1]) == "represent")
    { // rows: <op> <argument> <16 slot indices>; op 0 = read (argument ignored), 1 = write (argument = target, alpha 0.2),
      // 2 = update (argument = delta); applied in order on a representation seeded with 1; prints what each read returns
      Representation *rp = dynamic_cast<Representation *>(node->ptr());
      if (!rp) throw Exception(std::string(argv[3]) + ": not a representation");
      rp->reset(1);
      for (int i = 0; i < n; ++i)
      {
        if (rows[i].size() != 18) throw Exception("input row must hold op, argument and 16 slot indices");
        uint32_t idx[16];
        for (int k = 0; k < 16; ++k) idx[k] = (uint32_t)rows[i][2 + k];
        const int op = (int)rows[i][0];
        const double arg = rows[i][1];
        if (op == 0)
        {
          double out = 0;
          rp->read(idx, 1, &out);
          printf("%.17g\n", out);
        }
        else if (op == 1)
          rp->write(idx, 1, &arg, 0.2);
        else if (op == 2)
          rp->update(idx, 1, &arg);
        else
          throw Exception("unknown representation op");
      }
    }
    else if (std::string(argv[1]) == "stepwise")
    { // grlx_ops stepwise <yaml> <path of an experiment/online_learning> <file: "<seed> <replicas> <trials>">
      // The loop of OnlineLearningExperiment::run (online_learning.cpp:154-262) written HERE, on the host, over the experiment's
      // environment and agents as per-step objects (Environment::start / step, Agent::start / step / end): every call serves all
      // replicas on the GPU.  Prints the rows of replica 0 in the layout of the reference's golden files.
      StepwiseExperiment *ex = dynamic_cast<StepwiseExperiment *>(node->ptr());
      if (!ex) throw Exception(std::string(argv[3]) + ": not an experiment/online_learning");
      if (n != 1 || rows[0].size() != 3) throw Exception("input: one row `<seed> <replicas> <trials>`");
      RunOptions opt;
      opt.seed = (int64_t)rows[0][0];
      opt.replicas = (int)rows[0][1];
      const int trials = (int)rows[0][2];
      ex->open(opt);
      const int N = ex->replicas(), D = ex->obs_dims(), ti = ex->test_interval_of();
      std::vector<double> obs((size_t)N * D), action(N), reward(N), total(N);
      std::vector<int32_t> terminal(N), active(N);
      std::vector<long long> ss(N, 0);
      StepwiseEnvironment *env = ex->stepwise_environment();
      for (int tt = 0; tt < trials; ++tt)
      {
        const int test = (ti >= 0 && tt % (ti + 1) == ti) ? 1 : 0;                         // online_learning.cpp:160
        StepwiseAgent *agent = test ? ex->stepwise_test_agent() : ex->stepwise_agent();     // :167-168
        std::fill(active.begin(), active.end(), 1);
        std::fill(total.begin(), total.end(), 0.);
        env->start(test, active.data(), obs.data());                                        // :172
        agent->start(active.data(), obs.data(), action.data());                             // :178
        bool any = true;
        while (any)
        {
          env->step(active.data(), action.data(), obs.data(), reward.data(), terminal.data());     // :196
          agent->step(active.data(), 1., obs.data(), reward.data(), terminal.data(), action.data());   // :210-213 (terminal == 2: Agent::end)
          any = false;
          for (int k = 0; k < N; ++k)
            if (active[k])
            {
              total[k] += reward[k];                                                        // :202
              if (!test) ss[k]++;                                                           // :218
              if (terminal[k]) active[k] = 0;
              else any = true;
            }
        }
        if (ti >= 0 ? test : 1)
          printf("%15lld%15lld%15g\n", (long long)(ti >= 0 ? tt + 1 - (tt + 1) / (ti + 1) : tt), ss[0], total[0]);
      }
      ex->close();
    }
    else
      throw Exception(std::string("unknown operator '") + argv[1] + "'");
  }
  catch (Exception &ex)
  {
    log(0, ex.what());
    return 1;
  }
  return 0;
}
