// grlx_ops.cpp -- the plug-in interfaces of the host layer, driven from the command line (parity checks, mixed graphs):
//   grlx_ops project <yaml> <path of a projector/tile_coding> <input file>     Projector::project for every input row
//   grlx_ops envstep <yaml> <path of an environment/modeled> <input file>      Environment::step for every (state, action) row
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
    std::cerr << "usage: " << argv[0] << " project|envstep|represent <yaml> <object path> <input file>" << std::endl;
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
