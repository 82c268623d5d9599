// objects.cpp -- the reference's plug-in classes on the hot path, as configuration
// descriptors with the same TYPEINFO strings, parameter names, defaults and bad_param
// conditions; `experiment/online_learning` lowers the instantiated graph to a grlx_config
// and runs it on the GPU through the C ABI (include/grlx.h).  Nothing here computes the
// algorithm on the CPU: a graph the fused kernels do not implement is refused.
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <sstream>

#include "../../../include/grlx.h"
#include "configurable.h"
#include "objects.h"
#include "multi_gpu.h"

namespace grlx_host {

using VecD = std::vector<double>;
static const double kPi = 3.14159265358979323846;

// ------------------------------------------------------------ environment ---
struct Dynamics : Configurable { virtual int env_id() const = 0; };
struct Task : Configurable {
  virtual int env_id() const = 0;
  double timeout = 0, randomization = 0;
};

// dynamics/pendulum (pendulum.cpp:36-49): no parameters
struct PendulumDynamics : Dynamics {
  GRLX_TYPEINFO("dynamics/pendulum")
  int env_id() const override { return GRLX_ENV_PENDULUM; }
};
GRLX_REGISTER(PendulumDynamics)

// task/pendulum/swingup (pendulum.cpp:70-95)
struct PendulumSwingupTask : Task {
  GRLX_TYPEINFO("task/pendulum/swingup")
  int env_id() const override { return GRLX_ENV_PENDULUM; }
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("timeout", "Episode timeout", 2.99));
    config->push_back(CRP("randomization", "Level of start state randomization", 0.));
    config->push_back(CRP::provided("observation_dims", "int.observation_dims", "Number of observation dimensions"));
    config->push_back(CRP::provided("observation_min", "vector.observation_min", "Lower limit on observations"));
    config->push_back(CRP::provided("observation_max", "vector.observation_max", "Upper limit on observations"));
    config->push_back(CRP::provided("action_dims", "int.action_dims", "Number of action dimensions"));
    config->push_back(CRP::provided("action_min", "vector.action_min", "Lower limit on actions"));
    config->push_back(CRP::provided("action_max", "vector.action_max", "Upper limit on actions"));
    config->push_back(CRP::provided("reward_min", "double.reward_min", "Lower limit on immediate reward"));
    config->push_back(CRP::provided("reward_max", "double.reward_max", "Upper limit on immediate reward"));
  }
  void configure(Configuration &config) override
  {
    timeout = config["timeout"];
    randomization = config["randomization"];
    if (timeout < 0) throw bad_param("task/pendulum/swingup:timeout");
    if (randomization < 0 || randomization > 1) throw bad_param("task/pendulum/swingup:randomization");
    config.set("observation_dims", 2);
    config.set("observation_min", VecD{0., -12 * kPi});
    config.set("observation_max", VecD{2 * kPi, 12 * kPi});
    config.set("action_dims", 1);
    config.set("action_min", VecD{-3});
    config.set("action_max", VecD{3});
    config.set("reward_min", -5 * std::pow(kPi, 2) - 0.1 * std::pow(12 * kPi, 2) - 1 * std::pow(3, 2));
    config.set("reward_max", 0.);
  }
};
GRLX_REGISTER(PendulumSwingupTask)

// dynamics/acrobot, task/acrobot/balancing (acrobot.cpp:36-100): no parameters; the class
// default control step of model/dynamical (0.05 s) applies
struct AcrobotDynamics : Dynamics {
  GRLX_TYPEINFO("dynamics/acrobot")
  int env_id() const override { return GRLX_ENV_ACROBOT; }
};
GRLX_REGISTER(AcrobotDynamics)
struct AcrobotBalancingTask : Task {
  GRLX_TYPEINFO("task/acrobot/balancing")
  int env_id() const override { return GRLX_ENV_ACROBOT; }
  void request(const std::string &, ConfigurationRequest *config) override
  {
    for (const char *n : {"observation_dims", "observation_min", "observation_max", "action_dims", "action_min", "action_max", "reward_min", "reward_max"})
      config->push_back(CRP::provided(n, std::string("vector.") + n, "Task limits"));
  }
  void configure(Configuration &config) override
  {
    timeout = 20;                                       // acrobot.cpp:125 (fixed)
    config.set("observation_dims", 4);
    config.set("observation_min", VecD{kPi - 12 * kPi / 180, -12 * kPi / 180, -0.6, -1.1});
    config.set("observation_max", VecD{kPi + 12 * kPi / 180, 12 * kPi / 180, 0.6, 1.1});
    config.set("action_dims", 1);
    config.set("action_min", VecD{-1});
    config.set("action_max", VecD{1});
    config.set("reward_min", 1.);
    config.set("reward_max", 1.);
  }
};
GRLX_REGISTER(AcrobotBalancingTask)

// dynamics/cart_pole, task/cart_pole/swingup (cart_pole.cpp:36-56, 110-153)
struct CartPoleDynamics : Dynamics {
  GRLX_TYPEINFO("dynamics/cart_pole")
  int env_id() const override { return GRLX_ENV_CART_POLE; }
  void request(const std::string &, ConfigurationRequest *config) override
  { config->push_back(CRP("end_stop", "Simulate end stops (adds position and velocity to state)", 1)); }
  void configure(Configuration &config) override
  { if ((int)config["end_stop"] != 1) throw Exception(path() + ": end_stop must be 1 (task/cart_pole/swingup needs the 5-dimensional state)"); }
};
GRLX_REGISTER(CartPoleDynamics)
struct CartPoleSwingupTask : Task {
  GRLX_TYPEINFO("task/cart_pole/swingup")
  int end_stop_penalty = 1, action_penalty = 0;
  int env_id() const override { return GRLX_ENV_CART_POLE; }
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("timeout", "Episode timeout", 9.99));
    config->push_back(CRP("randomization", "Start state randomization", 0., CRP::Online));
    config->push_back(CRP("shaping", "Whether to use reward shaping", 0));
    config->push_back(CRP("gamma", "Discount rate for reward shaping", 1.));
    config->push_back(CRP("end_stop_penalty", "Terminate episode with penalty when end stop is reached", 1));
    config->push_back(CRP("action_penalty", "Penalize applied torque", 0));
    for (const char *n : {"observation_dims", "observation_min", "observation_max", "action_dims", "action_min", "action_max", "reward_min", "reward_max"})
      config->push_back(CRP::provided(n, std::string("vector.") + n, "Task limits"));
  }
  void configure(Configuration &config) override
  {
    timeout = config["timeout"]; randomization = config["randomization"];
    end_stop_penalty = config["end_stop_penalty"]; action_penalty = config["action_penalty"];
    if ((int)config["shaping"] != 0) throw Exception(path() + ": reward shaping is outside the accelerated path");
    config.set("observation_dims", 4);
    config.set("observation_min", VecD{-2.4, 0., -10.0, -5 * kPi});
    config.set("observation_max", VecD{2.4, 2 * kPi, 10.0, 5 * kPi});
    config.set("action_dims", 1);
    config.set("action_min", VecD{-15.});
    config.set("action_max", VecD{15.});
    config.set("reward_min", -2 * std::pow(2.4, 2) - 0.1 * std::pow(10, 2) - std::pow(kPi, 2) - 0.1 * std::pow(5 * kPi, 2) - action_penalty * 2 - end_stop_penalty * 10000);
    config.set("reward_max", 0.);
  }
};
GRLX_REGISTER(CartPoleSwingupTask)

struct Model : Configurable {
  double control_step = 0.05;
  int integration_steps = 5;
  virtual int env_id() const = 0;
};

// model/compass_walker (compass_walker.cpp:41-60), task/compass_walker/walk (:198-249)
struct CompassWalkerModel : Model {
  GRLX_TYPEINFO("model/compass_walker")
  double slope_angle = 0.004;
  int env_id() const override { return GRLX_ENV_COMPASS_WALKER; }
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("control_step", "Control step time", 0.2));
    config->push_back(CRP("integration_steps", "Number of integration steps per control step", 20));
    config->push_back(CRP("slope_angle", "Inclination of the slope", 0.004));
  }
  void configure(Configuration &config) override
  {
    control_step = config["control_step"]; integration_steps = config["integration_steps"]; slope_angle = config["slope_angle"];
    if (!(control_step >= 0.001)) throw bad_param("model/compass_walker:control_step");
    if (integration_steps < 1) throw bad_param("model/compass_walker:integration_steps");
  }
};
GRLX_REGISTER(CompassWalkerModel)
struct CompassWalkerWalkTask : Task {
  GRLX_TYPEINFO("task/compass_walker/walk")
  double initial_state_variation = 0.2, slope_angle = 0.004, negative_reward = -100;
  int env_id() const override { return GRLX_ENV_COMPASS_WALKER; }
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("timeout", "Learning episode timeout", 100.));
    config->push_back(CRP("initial_state_variation", "Variation of initial state", 0.2));
    config->push_back(CRP("slope_angle", "Inclination of the slope", 0.004, CRP::System));
    config->push_back(CRP("negative_reward", "Negative reward", -100.));
    config->push_back(CRP("observe", "State elements observed by an agent", VecD{1, 1, 1, 1, 1, 0, 0}));
    config->push_back(CRP("steps", "number of steps after which task is terminated", 0));
    for (const char *n : {"observation_dims", "observation_min", "observation_max", "action_dims", "action_min", "action_max", "reward_min", "reward_max"})
      config->push_back(CRP::provided(n, std::string("vector.") + n, "Task limits"));
  }
  void configure(Configuration &config) override
  {
    timeout = config["timeout"]; initial_state_variation = config["initial_state_variation"];
    slope_angle = config["slope_angle"]; negative_reward = config["negative_reward"];
    const VecD observe = config["observe"].v();
    if (observe.size() != 7) throw bad_param("task/walk:observe");
    if (observe != VecD{1, 1, 1, 1, 1, 0, 0}) throw Exception(path() + ": only the default observation mask [1,1,1,1,1,0,0] is on the accelerated path");
    if ((int)config["steps"] != 0) throw Exception(path() + ": steps > 0 is outside the accelerated path");
    if (negative_reward > 0) throw bad_param("task/compass_walker/walk:negative_reward");
    config.set("observation_dims", 5);
    config.set("observation_min", VecD{-kPi / 8, -kPi / 4, -kPi, -kPi, 0});
    config.set("observation_max", VecD{kPi / 8, kPi / 4, kPi, kPi, 0.5});
    config.set("action_dims", 1);
    config.set("action_min", VecD{-1.2});
    config.set("action_max", VecD{1.2});
    config.set("reward_min", -101.);
    config.set("reward_max", 50.);
  }
};
GRLX_REGISTER(CompassWalkerWalkTask)

// model/dynamical (modeled.cpp:234-252)
struct DynamicalModel : Model {
  GRLX_TYPEINFO("model/dynamical")
  Dynamics *dynamics = nullptr;
  int env_id() const override { return dynamics ? dynamics->env_id() : -1; }
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("control_step", "Control step time", 0.05));
    config->push_back(CRP("integration_steps", "Number of integration steps per control step", 5));
    config->push_back(CRP("dynamics", "dynamics", "Equations of motion", (Configurable *)nullptr));
  }
  void configure(Configuration &config) override
  {
    dynamics = dynamic_cast<Dynamics *>(config["dynamics"].ptr());
    control_step = config["control_step"];
    integration_steps = config["integration_steps"];
    if (!(control_step >= 0.00001)) throw bad_param("model/dynamical:control_step");
    if (integration_steps < 1) throw bad_param("model/dynamical:integration_steps");
    if (!dynamics) throw Exception(path() + ": dynamics outside the accelerated path");
  }
};
GRLX_REGISTER(DynamicalModel)

// environment/modeled (modeled.cpp:35-119)
struct ModeledEnvironment : Environment {
  GRLX_TYPEINFO("environment/modeled")
  void lower_env(grlx_config *c) const;                 // model + task -> the environment fields of a grlx_config
  void dims(int *state_dims, int *obs_dims) const override
  {
    if (grlx_env_dims(task->env_id(), state_dims, obs_dims) != GRLX_OK) throw Exception(path() + ": " + grlx_last_error());
  }
  void step(double *state, const double *action, int n, double *obs, double *reward, int32_t *terminal) const override
  { // ModeledEnvironment::step for n independent instances, on the GPU
    grlx_config c;
    grlx_config_pendulum_sarsa(&c);
    lower_env(&c);
    int S = 0, D = 0;
    dims(&S, &D);
    c.projector.dims = D + 1;                           // the fine-grained operator only reads the environment fields
    for (int i = 0; i < GRLX_MAX_DIMS; ++i) { c.projector.resolution[i] = 1.; c.projector.wrapping[i] = 0.; }
    if (grlx_env_step(&c, state, action, n, obs, reward, terminal) != GRLX_OK) throw Exception(path() + ": " + grlx_last_error());
  }
  Model *model = nullptr;
  Task *task = nullptr;
  int discrete_time = 1;
  Configurable *exporter = nullptr;     // optional exporter/csv for the transition log with the model state (modeled.cpp:67-71)
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("discrete_time", "Always report unit step time", 1));
    config->push_back(CRP("window", "Number of observations to concatenate", 1));
    config->push_back(CRP("stride", "Time steps between concatenated observations", 1));
    config->push_back(CRP("delta", "Action delta for differential actions", VecD{}));
    config->push_back(CRP("model", "model", "Environment model", (Configurable *)nullptr));
    config->push_back(CRP("task", "task", "Task to perform in the environment (should match model)", (Configurable *)nullptr));
    config->push_back(CRP("exporter", "exporter", "Optional exporter for transition log", (Configurable *)nullptr, true));
    for (const char *n : {"observation_dims", "observation_min", "observation_max", "action_dims", "action_min", "action_max", "reward_min", "reward_max"})
      config->push_back(CRP::provided(n, std::string("vector.") + n, "Forwarded task parameter"));
  }
  void configure(Configuration &config) override
  {
    model = dynamic_cast<Model *>(config["model"].ptr());
    task = dynamic_cast<Task *>(config["task"].ptr());
    discrete_time = config["discrete_time"];
    if (!model || !task) throw Exception(path() + ": model/task outside the accelerated path");
    if ((int)config["window"] != 1 || (int)config["stride"] != 1 || !config["delta"].v().empty())
      throw Exception(path() + ": window/stride/delta are not supported by the accelerated path");
    exporter = config["exporter"].ptr();
    if (discrete_time != 1) throw Exception(path() + ": discrete_time must be 1 on the accelerated path");
    if (model->env_id() != task->env_id()) throw Exception(path() + ": task does not match the model");
    // forward the task's provided parameters (modeled.cpp:79-114)
    Configurator *t = task->configurator;
    for (const char *n : {"observation_dims", "observation_min", "observation_max", "action_dims", "action_min", "action_max", "reward_min", "reward_max"})
      config.set(n, t->child(n)->value);
  }
};
GRLX_REGISTER(ModeledEnvironment)

void ModeledEnvironment::lower_env(grlx_config *c) const
{
  const Model *m = model;
  const Task *t = task;
  c->env = t->env_id();
  c->control_step = m->control_step;
  c->integration_steps = m->integration_steps;
  c->discrete_time = discrete_time;
  c->timeout = t->timeout;
  c->randomization = t->randomization;
  if (const CartPoleSwingupTask *cp = dynamic_cast<const CartPoleSwingupTask *>(t))
  { c->end_stop_penalty = cp->end_stop_penalty; c->action_penalty = cp->action_penalty; }
  if (const CompassWalkerWalkTask *w = dynamic_cast<const CompassWalkerWalkTask *>(t))
  {
    const CompassWalkerModel *wm = dynamic_cast<const CompassWalkerModel *>(m);
    if (!wm || wm->slope_angle != w->slope_angle) throw Exception(w->path() + ": slope_angle must match model/compass_walker");
    c->slope_angle = w->slope_angle; c->initial_state_variation = w->initial_state_variation; c->negative_reward = w->negative_reward;
  }
}

// ------------------------------------------------------------------ agent ---
// discretizer/uniform (uniform.cpp:34-95)
struct UniformDiscretizer : Configurable {
  GRLX_TYPEINFO("discretizer/uniform")
  VecD min, max, steps;
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("min", "Lower limit", VecD{}, CRP::System));
    config->push_back(CRP("max", "Upper limit", VecD{}, CRP::System));
    config->push_back(CRP("steps", "Discretization steps per dimension", VecD{}));
  }
  void configure(Configuration &config) override
  {
    min = config["min"].v(); max = config["max"].v(); steps = config["steps"].v();
    if (min.size() != max.size() || min.size() != steps.size()) throw bad_param("discretizer/uniform:{min,max,steps}");
    for (double s : steps) if (s < 1) throw bad_param("discretizer/uniform:steps");
  }
};
GRLX_REGISTER(UniformDiscretizer)

// projector/tile_coding (tile_coding.cpp:34-80)
struct TileCodingProjector : Projector {
  GRLX_TYPEINFO("projector/tile_coding")
  int n_tilings() const override { return tilings; }
  int n_dims() const override { return (int)resolution.size(); }
  void project(const double *in, int n, uint32_t *out) const override
  { // TileCodingProjector::_project for n inputs, on the GPU
    grlx_tile_spec t;
    memset(&t, 0, sizeof(t));
    t.tilings = tilings; t.memory = memory; t.dims = (int)resolution.size();
    if (resolution.size() > GRLX_MAX_DIMS) throw bad_param("projector/tile_coding:resolution");
    for (size_t i = 0; i < resolution.size(); ++i) { t.resolution[i] = resolution[i]; t.wrapping[i] = wrapping[i]; }
    // safe >= 1 returns CLAIMED slots, which depend on what was written before (tile_coding.h:116-151): that state lives in an
    // experiment's tables, not in a stand-alone projector -- refuse rather than return the unclaimed `hash % memory`
    if (safe != 0) throw Exception(path() + ": Projector::project as a stand-alone operator serves safe = 0 only (claims live in the experiment's tables)");
    if (grlx_project(&t, in, n, out) != GRLX_OK) throw Exception(path() + ": " + grlx_last_error());
  }
  int tilings = 16, memory = 8 * 1024 * 1024, safe = 0;
  VecD resolution, wrapping;
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("tilings", "Number of tilings", 16));
    config->push_back(CRP("memory", "Hash table size", 8 * 1024 * 1024));
    config->push_back(CRP("safe", "Collision detection (0=off, 1=claim on write, 2=claim always)", 0));
    config->push_back(CRP("resolution", "Size of a single tile", VecD{}));
    config->push_back(CRP("wrapping", "Wrapping boundaries (must be multiple of resolution)", VecD{}));
  }
  void configure(Configuration &config) override
  {
    tilings = config["tilings"]; memory = config["memory"]; safe = config["safe"];
    resolution = config["resolution"].v(); wrapping = config["wrapping"].v();
    if (wrapping.empty()) wrapping.assign(resolution.size(), 0.);
    if (wrapping.size() != resolution.size()) throw bad_param("projector/tile_coding:wrapping");
    for (size_t ii = 0; ii < resolution.size(); ++ii)
    { // tile_coding.cpp:66-78
      double w = wrapping[ii] * (tilings / resolution[ii]);
      if (std::fabs(w - std::round(w)) > 0.001) throw bad_param("projector/tile_coding:wrapping");
    }
    if (safe < 0 || safe > 2) throw Exception(path() + ": safe must be 0, 1 or 2");
  }
};
GRLX_REGISTER(TileCodingProjector)

// representation/parameterized/linear (linear.cpp:34-101)
struct LinearRepresentation : Representation {
  GRLX_TYPEINFO("representation/parameterized/linear")
  VecD init_min, init_max, output_min, output_max;
  int memory = 8 * 1024 * 1024, outputs = 1, limit = 1, interval = 0;
  double tau = 1;
  // the stand-alone use of the object (Representation interface): a private one-replica context holds the parameters
  grlx_ctx *own = nullptr;
  ~LinearRepresentation() override { if (own) grlx_destroy(own); }
  void reset(int64_t seed) override
  {
    if (outputs != 1) throw Exception(path() + ": the GPU operators serve representations with one output");
    // a representation with a target network (interval) reads through target() and counts updates towards the next
    // synchronisation (representation.h:266-306): the stand-alone operators do not carry that state -- refuse
    if (interval != 0) throw Exception(path() + ": Representation::read/write/update as stand-alone operators serve interval = 0 only (no target network)");
    if (own) { grlx_destroy(own); own = nullptr; }
    grlx_config c;
    grlx_config_pendulum_sarsa(&c);
    c.n_replicas = 1;
    c.projector.memory = memory;
    c.representation.init_min = init_min[0];
    c.representation.init_max = init_max[0];
    c.representation.output_min = output_min[0];
    c.representation.output_max = output_max[0];
    c.representation.limit = limit;
    if (grlx_create(&c, &seed, &own) != GRLX_OK) { own = nullptr; throw Exception(path() + ": " + grlx_last_error()); }
  }
  grlx_ctx *context()
  {
    if (!own) reset(1);
    return own;
  }
  void read(const uint32_t *idx, int n, double *out) override
  {
    const std::vector<int32_t> rep((size_t)n, 0);
    if (grlx_read(context(), 0, rep.data(), idx, n, out) != GRLX_OK) throw Exception(path() + ": " + grlx_last_error());
  }
  void write(const uint32_t *idx, int n, const double *target, double alpha) override
  {
    const std::vector<int32_t> rep((size_t)n, 0);
    if (grlx_write(context(), 0, rep.data(), idx, n, target, alpha) != GRLX_OK) throw Exception(path() + ": " + grlx_last_error());
  }
  void update(const uint32_t *idx, int n, const double *delta) override
  {
    const std::vector<int32_t> rep((size_t)n, 0);
    if (grlx_update(context(), 0, rep.data(), idx, n, delta) != GRLX_OK) throw Exception(path() + ": " + grlx_last_error());
  }
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("init_min", "Lower initial value limit", VecD{0.}));
    config->push_back(CRP("init_max", "Upper initial value limit", VecD{1.}));
    config->push_back(CRP("memory", "Feature vector size", 8 * 1024 * 1024, CRP::System));
    config->push_back(CRP("outputs", "Number of outputs", 1, CRP::System));
    config->push_back(CRP("output_min", "Lower output limit", VecD{}, CRP::System));
    config->push_back(CRP("output_max", "Upper output limit", VecD{}, CRP::System));
    config->push_back(CRP("limit", "Limit parameters to same range as outputs", 1));
    config->push_back(CRP("interval", "Target network update interval", 0.));     // ParameterizedRepresentation (unused here)
    config->push_back(CRP("tau", "Target network update rate", 1.));
  }
  void configure(Configuration &config) override
  {
    memory = config["memory"]; outputs = config["outputs"]; limit = config["limit"];
    init_min = config["init_min"].v(); init_max = config["init_max"].v();
    if (!init_min.empty() && (int)init_min.size() < outputs) init_min.assign((size_t)outputs, init_min[0]);
    if ((int)init_min.size() != outputs) throw bad_param("representation/parameterized/linear:init_min");
    if (!init_max.empty() && (int)init_max.size() < outputs) init_max.assign((size_t)outputs, init_max[0]);
    if ((int)init_max.size() != outputs) throw bad_param("representation/parameterized/linear:max");
    output_min = config["output_min"].v();
    if (output_min.empty()) output_min.assign((size_t)outputs, -DBL_MAX);
    if ((int)output_min.size() != outputs) throw bad_param("representation/parameterized/linear:output_min");
    output_max = config["output_max"].v();
    if (output_max.empty()) output_max.assign((size_t)outputs, DBL_MAX);
    if ((int)output_max.size() != outputs) throw bad_param("representation/parameterized/linear:output_max");
    interval = (int)(double)config["interval"];                   // ParameterizedRepresentation (representation.h:173-190)
    tau = config["tau"];
    if (interval < 0 || !(tau >= 0 && tau <= 1)) throw bad_param("representation/parameterized/linear:{interval,tau}");
  }
};
GRLX_REGISTER(LinearRepresentation)

// sampler/greedy, sampler/epsilon_greedy (greedy.cpp:34-130)
struct GreedySampler : Configurable {
  GRLX_TYPEINFO("sampler/greedy")
  virtual bool explores() const { return false; }
};
GRLX_REGISTER(GreedySampler)
struct EpsilonGreedySampler : GreedySampler {
  GRLX_TYPEINFO("sampler/epsilon_greedy")
  VecD epsilon; double decay_rate = 1, decay_min = 0;
  bool explores() const override { return true; }
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("epsilon", "Exploration rate (can be defined per action)", VecD{0.05}, CRP::Online));
    config->push_back(CRP("decay_rate", "Multiplicative decay factor per episode", 1.));
    config->push_back(CRP("decay_min", "Minimum decay (eps_min = eps*decay_min)", 0.));
  }
  void configure(Configuration &config) override
  {
    epsilon = config["epsilon"].v(); decay_rate = config["decay_rate"]; decay_min = config["decay_min"];
    if (epsilon.size() < 1) throw bad_param("sampler/epsilon_greedy:epsilon");
    if (epsilon.size() > 1) throw Exception(path() + ": per-action epsilon is outside the accelerated path");
  }
};
GRLX_REGISTER(EpsilonGreedySampler)

struct Policy : Configurable {};

// mapping/policy/action (action.cpp:38-91)
struct ActionPolicy : Policy {
  GRLX_TYPEINFO("mapping/policy/action")
  VecD sigma, theta, min, max; double decay_rate = 1, decay_min = 0;
  TileCodingProjector *projector = nullptr; LinearRepresentation *representation = nullptr;
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("sigma", "Standard deviation of Gaussian exploration distribution", VecD{}));
    config->push_back(CRP("theta", "Ornstein-Uhlenbeck friction term (1=pure Gaussian noise)", VecD{}));
    config->push_back(CRP("decay_rate", "Multiplicative decay factor per episode", 1.));
    config->push_back(CRP("decay_min", "Minimum decay (sigma_min = sigma*decay_min)", 0.));
    config->push_back(CRP("renormalize", "Renormalize representation output from [-1, 1] to [min, max]", 0));
    config->push_back(CRP("output_min", "Lower limit on outputs", VecD{}, CRP::System));
    config->push_back(CRP("output_max", "Upper limit on outputs", VecD{}, CRP::System));
    config->push_back(CRP("projector", "projector.observation", "Projects observations onto representation space", (Configurable *)nullptr));
    config->push_back(CRP("representation", "representation.action", "Action representation", (Configurable *)nullptr));
  }
  void configure(Configuration &config) override
  {
    projector = dynamic_cast<TileCodingProjector *>(config["projector"].ptr());
    representation = dynamic_cast<LinearRepresentation *>(config["representation"].ptr());
    sigma = config["sigma"].v(); theta = config["theta"].v();
    decay_rate = config["decay_rate"]; decay_min = config["decay_min"];
    min = config["output_min"].v(); max = config["output_max"].v();
    if (min.size() != max.size() || min.empty()) throw bad_param("policy/action:{output_min,output_max}");
    if (sigma.empty()) sigma = VecD{0.};
    if (sigma.size() == 1) sigma.assign(min.size(), sigma[0]);
    if (sigma.size() != min.size()) throw bad_param("policy/action:sigma");
    if (theta.empty()) theta = VecD{1.};
    if (theta.size() == 1) theta.assign(min.size(), theta[0]);
    if (theta.size() != min.size()) throw bad_param("policy/action:theta");
    if ((int)config["renormalize"] != 0) throw Exception(path() + ": renormalize is outside the accelerated path");
    if (!projector || !representation) throw Exception(path() + ": projector/representation outside the accelerated path");
    if (min.size() != 1) throw Exception(path() + ": one action dimension supported");
  }
};
GRLX_REGISTER(ActionPolicy)

// mapping/policy/discrete/value/q (q.cpp:35-52); the reference's own test yaml still says policy/discrete/q
struct QPolicy : Policy {
  GRLX_TYPEINFO("mapping/policy/discrete/value/q")
  UniformDiscretizer *discretizer = nullptr; TileCodingProjector *projector = nullptr;
  LinearRepresentation *representation = nullptr; GreedySampler *sampler = nullptr;
  Configurable *any_projector = nullptr, *any_representation = nullptr;     // whatever the yaml gave (the batch path: normalizing projector, iterative ANN)
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("discretizer", "discretizer.action", "Action discretizer", (Configurable *)nullptr));
    config->push_back(CRP("projector", "projector.pair", "Projects observation-action pairs onto representation space", (Configurable *)nullptr));
    config->push_back(CRP("representation", "representation.value/action", "Action-value representation", (Configurable *)nullptr));
    config->push_back(CRP("sampler", "sampler", "Samples actions from action-values", (Configurable *)nullptr));
  }
  void configure(Configuration &config) override
  {
    discretizer = dynamic_cast<UniformDiscretizer *>(config["discretizer"].ptr());
    projector = dynamic_cast<TileCodingProjector *>(config["projector"].ptr());
    representation = dynamic_cast<LinearRepresentation *>(config["representation"].ptr());
    sampler = dynamic_cast<GreedySampler *>(config["sampler"].ptr());
    any_projector = config["projector"].ptr();
    any_representation = config["representation"].ptr();
    // (projector / representation kinds are checked where the graph is lowered: tile coding + linear for experiment/online_learning,
    //  projector/pre/normalizing + representation/iterative over an ANN for experiment/batch_learning)
    if (!discretizer || !any_projector || !any_representation || !sampler)
      throw Exception(path() + ": the accelerated path needs discretizer/uniform, a projector, a representation and a greedy sampler");
  }
};
GRLX_REGISTER(QPolicy)
struct QPolicyLegacy : QPolicy { GRLX_TYPEINFO("mapping/policy/discrete/q") };      // name used by tests/pendulum-sarsa-tc.yaml
GRLX_REGISTER(QPolicyLegacy)

struct Trace : Configurable { virtual int kind() const = 0; };
struct ReplacingTrace : Trace { GRLX_TYPEINFO("trace/enumerated/replacing") int kind() const override { return GRLX_TRACE_REPLACING; } };
GRLX_REGISTER(ReplacingTrace)
struct AccumulatingTrace : Trace { GRLX_TYPEINFO("trace/enumerated/accumulating") int kind() const override { return GRLX_TRACE_ACCUMULATING; } };
GRLX_REGISTER(AccumulatingTrace)

struct Predictor : Configurable {};

// predictor/critic/sarsa (sarsa.cpp:35-60), predictor/critic/q (advantage.cpp:35-62)
struct TDPredictorBase : Predictor {
  double alpha = 0.2, gamma = 0.97, lambda = 0.65, kappa = 0;
  TileCodingProjector *projector = nullptr; LinearRepresentation *representation = nullptr; Trace *trace = nullptr;
  virtual int agent_id() const = 0;
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("alpha", "Learning rate", 0.2));
    config->push_back(CRP("gamma", "Discount rate", 0.97));
    config->push_back(CRP("lambda", "Trace decay rate", 0.65));
    if (agent_id() == GRLX_AGENT_ADVANTAGE)
      config->push_back(CRP("kappa", "Advantage scaling factor", 0.2));          // advantage.cpp:188
    if (agent_id() == GRLX_AGENT_Q || agent_id() == GRLX_AGENT_ADVANTAGE)
      config->push_back(CRP("discretizer", "discretizer.action", "Action discretizer", (Configurable *)nullptr));
    config->push_back(CRP("projector", "projector.pair", "Projects observation-action pairs onto representation space", (Configurable *)nullptr));
    config->push_back(CRP("representation", "representation.value/action", "Q-value representation", (Configurable *)nullptr));
    config->push_back(CRP("trace", "trace", "Trace of projections", (Configurable *)nullptr, true));
    config->push_back(CRP("importer", "importer", "Optional importer", (Configurable *)nullptr, true));
    config->push_back(CRP("exporter", "exporter", "Optional exporter", (Configurable *)nullptr, true));
  }
  void configure(Configuration &config) override
  {
    alpha = config["alpha"]; gamma = config["gamma"]; lambda = config["lambda"];
    if (agent_id() == GRLX_AGENT_ADVANTAGE) kappa = config["kappa"];
    projector = dynamic_cast<TileCodingProjector *>(config["projector"].ptr());
    representation = dynamic_cast<LinearRepresentation *>(config["representation"].ptr());
    trace = dynamic_cast<Trace *>(config["trace"].ptr());
    if (!projector || !representation) throw Exception(path() + ": projector/representation outside the accelerated path");
    if (config["importer"].ptr() || config["exporter"].ptr()) throw Exception(path() + ": importer/exporter are outside the accelerated path");
  }
};
struct SARSAPredictor : TDPredictorBase { GRLX_TYPEINFO("predictor/critic/sarsa") int agent_id() const override { return GRLX_AGENT_SARSA; } };
GRLX_REGISTER(SARSAPredictor)
struct QPredictor : TDPredictorBase { GRLX_TYPEINFO("predictor/critic/q") int agent_id() const override { return GRLX_AGENT_Q; } };
GRLX_REGISTER(QPredictor)
// predictor/critic/advantage (advantage.cpp:181-268): advantage learning, scaling factor kappa
struct AdvantagePredictor : TDPredictorBase { GRLX_TYPEINFO("predictor/critic/advantage") int agent_id() const override { return GRLX_AGENT_ADVANTAGE; } };
GRLX_REGISTER(AdvantagePredictor)
// predictor/critic/expected_sarsa (sarsa.cpp:134-165): the target policy must be the learning policy
struct ExpectedSARSAPredictor : TDPredictorBase {
  GRLX_TYPEINFO("predictor/critic/expected_sarsa")
  Configurable *target_policy = nullptr;
  int agent_id() const override { return GRLX_AGENT_EXPECTED_SARSA; }
  void request(const std::string &role, ConfigurationRequest *config) override
  {
    TDPredictorBase::request(role, config);
    config->push_back(CRP("policy", "mapping/policy/discrete/value", "Value based target policy", (Configurable *)nullptr));
  }
  void configure(Configuration &config) override
  {
    TDPredictorBase::configure(config);
    target_policy = config["policy"].ptr();
  }
};
GRLX_REGISTER(ExpectedSARSAPredictor)
// predictor/critic/qv (qv.cpp:35-64): Q(s,a) and V(s) tables, the trace on V
struct QVPredictor : TDPredictorBase {
  GRLX_TYPEINFO("predictor/critic/qv")
  double beta = 0.1;
  TileCodingProjector *v_projector = nullptr; LinearRepresentation *v_representation = nullptr;
  int agent_id() const override { return GRLX_AGENT_QV; }
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("alpha", "State-action value learning rate", 0.2));
    config->push_back(CRP("beta", "State value learning rate", 0.1));
    config->push_back(CRP("gamma", "Discount rate", 0.97));
    config->push_back(CRP("lambda", "Trace decay rate", 0.65));
    config->push_back(CRP("q_projector", "projector.pair", "Projects observation-action pairs onto representation space", (Configurable *)nullptr));
    config->push_back(CRP("q_representation", "representation.value/action", "State-action value representation (Q)", (Configurable *)nullptr));
    config->push_back(CRP("v_projector", "projector.observation", "Projects observations onto representation space", (Configurable *)nullptr));
    config->push_back(CRP("v_representation", "representation.value/state", "State value representation (V)", (Configurable *)nullptr));
    config->push_back(CRP("trace", "trace", "Trace of projections", (Configurable *)nullptr, true));
    config->push_back(CRP("importer", "importer", "Optional importer", (Configurable *)nullptr, true));
    config->push_back(CRP("exporter", "exporter", "Optional exporter", (Configurable *)nullptr, true));
  }
  void configure(Configuration &config) override
  {
    alpha = config["alpha"]; beta = config["beta"]; gamma = config["gamma"]; lambda = config["lambda"];
    projector = dynamic_cast<TileCodingProjector *>(config["q_projector"].ptr());
    representation = dynamic_cast<LinearRepresentation *>(config["q_representation"].ptr());
    v_projector = dynamic_cast<TileCodingProjector *>(config["v_projector"].ptr());
    v_representation = dynamic_cast<LinearRepresentation *>(config["v_representation"].ptr());
    trace = dynamic_cast<Trace *>(config["trace"].ptr());
    if (!projector || !representation || !v_projector || !v_representation) throw Exception(path() + ": projectors/representations outside the accelerated path");
    if (config["importer"].ptr() || config["exporter"].ptr()) throw Exception(path() + ": importer/exporter are outside the accelerated path");
  }
};
GRLX_REGISTER(QVPredictor)
// names from before the reference's predictor/critic/* rename, still used by its tests/pendulum-sarsa-tc.yaml
struct SARSAPredictorLegacy : SARSAPredictor { GRLX_TYPEINFO("predictor/sarsa") };
GRLX_REGISTER(SARSAPredictorLegacy)
struct QPredictorLegacy : QPredictor { GRLX_TYPEINFO("predictor/q") };
GRLX_REGISTER(QPredictorLegacy)

// predictor/critic/td (predictors/td.cpp:35-58): V(s) critic with a trace
struct VPredictor : Predictor {
  GRLX_TYPEINFO("predictor/critic/td")
  double alpha = 0.2, gamma = 0.97, lambda = 0.65;
  TileCodingProjector *projector = nullptr; LinearRepresentation *representation = nullptr; Trace *trace = nullptr;
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("alpha", "Learning rate", 0.2));
    config->push_back(CRP("gamma", "Discount rate", 0.97));
    config->push_back(CRP("lambda", "Trace decay rate", 0.65));
    config->push_back(CRP("projector", "projector.observation", "Projects observations onto representation space", (Configurable *)nullptr));
    config->push_back(CRP("representation", "representation.value/state", "State value representation", (Configurable *)nullptr));
    config->push_back(CRP("trace", "trace", "Trace of projections", (Configurable *)nullptr, true));
    config->push_back(CRP("importer", "importer", "Optional importer", (Configurable *)nullptr, true));
    config->push_back(CRP("exporter", "exporter", "Optional exporter", (Configurable *)nullptr, true));
  }
  void configure(Configuration &config) override
  {
    alpha = config["alpha"]; gamma = config["gamma"]; lambda = config["lambda"];
    projector = dynamic_cast<TileCodingProjector *>(config["projector"].ptr());
    representation = dynamic_cast<LinearRepresentation *>(config["representation"].ptr());
    trace = dynamic_cast<Trace *>(config["trace"].ptr());
    if (!projector || !representation) throw Exception(path() + ": projector/representation outside the accelerated path");
    if (config["importer"].ptr() || config["exporter"].ptr()) throw Exception(path() + ": importer/exporter are outside the accelerated path");
  }
};
GRLX_REGISTER(VPredictor)

// predictor/ac/action (ac.cpp:36-70)
struct ActionACPredictor : Predictor {
  GRLX_TYPEINFO("predictor/ac/action")
  double alpha = 0.01; std::string update_method = "proportional"; VecD step_limit;
  TileCodingProjector *projector = nullptr; LinearRepresentation *representation = nullptr; VPredictor *critic = nullptr;
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("alpha", "Critic learning rate", 0.01));
    config->push_back(CRP("update_method", "Actor update method", std::string("proportional")));
    config->push_back(CRP("step_limit", "Actor exploration step limit", VecD{}));
    config->push_back(CRP("projector", "projector.observation", "Projects observations onto actor representation space", (Configurable *)nullptr));
    config->push_back(CRP("representation", "representation.action", "Action representation", (Configurable *)nullptr));
    config->push_back(CRP("critic", "predictor/critic", "Critic predictor", (Configurable *)nullptr));
    config->push_back(CRP("importer", "importer", "Optional importer", (Configurable *)nullptr, true));
    config->push_back(CRP("exporter", "exporter", "Optional exporter", (Configurable *)nullptr, true));
  }
  void configure(Configuration &config) override
  {
    alpha = config["alpha"]; update_method = config["update_method"].str(); step_limit = config["step_limit"].v();
    projector = dynamic_cast<TileCodingProjector *>(config["projector"].ptr());
    representation = dynamic_cast<LinearRepresentation *>(config["representation"].ptr());
    critic = dynamic_cast<VPredictor *>(config["critic"].ptr());
    if (update_method != "proportional" && update_method != "cacla") throw bad_param("predictor/ac/action:update_method");
    if (step_limit.size() > 1) throw bad_param("predictor/ac:step_limit");
    if (!projector || !representation || !critic) throw Exception(path() + ": the accelerated path needs tile coding, a linear actor and a predictor/critic/td critic");
  }
};
GRLX_REGISTER(ActionACPredictor)

// agent/td (td.cpp:34-48), agent/fixed (fixed.cpp:34-45)
struct TDAgent : Configurable {
  GRLX_TYPEINFO("agent/td")
  Policy *policy = nullptr; Predictor *predictor = nullptr;
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("policy", "mapping/policy", "Control policy", (Configurable *)nullptr));
    config->push_back(CRP("predictor", "predictor", "Value function predictor", (Configurable *)nullptr));
  }
  void configure(Configuration &config) override
  {
    policy = dynamic_cast<Policy *>(config["policy"].ptr());
    predictor = dynamic_cast<Predictor *>(config["predictor"].ptr());
    if (!policy || !predictor) throw Exception(path() + ": policy/predictor outside the accelerated path");
  }
};
GRLX_REGISTER(TDAgent)
struct FixedAgent : Configurable {
  GRLX_TYPEINFO("agent/fixed")
  Policy *policy = nullptr;
  void request(const std::string &, ConfigurationRequest *config) override
  { config->push_back(CRP("policy", "mapping/policy", "Control policy", (Configurable *)nullptr)); }
  void configure(Configuration &config) override
  {
    policy = dynamic_cast<Policy *>(config["policy"].ptr());
    if (!policy) throw Exception(path() + ": policy outside the accelerated path");
  }
};
GRLX_REGISTER(FixedAgent)

// exporter/csv (exporters/csv.cpp:37-206, exporter.h:62-95): comma-separated transition log.
// Same parameters, file naming (<file>-<variant>-<run counter>.csv), header styles and number
// format; fed from the device's per-step taps after a run instead of once per step.
struct CSVExporter : Configurable {
  GRLX_TYPEINFO("exporter/csv")
  std::string file, fields, style = "line", variant = "all";
  int enabled = 1, offset = 0;
  std::vector<std::string> headers;
  std::vector<size_t> order;
  std::ofstream stream;
  bool header_due = true;
  std::map<std::string, int> counter;

  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("file", "Output base filename", file));
    config->push_back(CRP("fields", "Comma-separated list of fields to write", fields));
    config->push_back(CRP("style", "Header style", style));
    config->push_back(CRP("variant", "Variant to export", variant));
    config->push_back(CRP("enabled", "Enable writing to output file", enabled, CRP::Online));
    config->push_back(CRP("offset", "Start of numbering", offset));
  }
  void configure(Configuration &config) override
  {
    file = config["file"].str(); fields = config["fields"].str(); style = config["style"].str(); variant = config["variant"].str();
    enabled = config["enabled"]; offset = config["offset"];
    if (file.empty()) throw bad_param("exporter/csv:file");
    if (style != "none" && style != "line" && style != "meshup") throw bad_param("exporter/csv:style");
    if (variant != "test" && variant != "learn" && variant != "all") throw bad_param("exporter/csv:variant");
    if (enabled != 0 && enabled != 1) throw bad_param("exporter/csv:enabled");
  }
  // register the columns the caller will supply and resolve `fields` against them (csv.cpp:66-121)
  void init(const std::vector<std::string> &h)
  {
    headers = h;
    order.clear();
    if (fields.empty())
    {
      for (size_t i = 0; i < h.size(); ++i) order.push_back(i);
      return;
    }
    std::stringstream list(fields);
    std::string item;
    while (std::getline(list, item, ','))
    {
      item.erase(std::remove_if(item.begin(), item.end(), [](char ch) { return ch == ' ' || ch == '\t'; }), item.end());
      bool known = false;
      for (size_t i = 0; i < h.size(); ++i)
        if (h[i] == item) { order.push_back(i); known = true; }
      if (!known)
      {
        log(0, "Requested unregistered field '" + item + "'");
        throw bad_param("exporter/csv:fields");
      }
    }
  }
  // csv.cpp:123-157: a non-appending open starts the next numbered file of the variant
  void open(const std::string &which, bool append)
  {
    if (stream.is_open()) stream.close();
    if (variant != "all" && variant != which) return;
    std::string name = which.empty() ? file : file + "-" + which;
    int &runs_seen = counter[name];
    if (runs_seen == 0) runs_seen = offset;
    if (!append) runs_seen++;
    name += "-" + std::to_string(runs_seen - 1) + ".csv";
    std::ifstream probe(name, std::ios::binary | std::ios::ate);
    header_due = !append || !probe || probe.tellg() == std::streampos(0);
    probe.close();
    stream.open(name, append ? (std::ios::out | std::ios::app) : (std::ios::out | std::ios::trunc));
    if (!stream.good())
    {
      log(0, "Could not open '" + name + "' for writing");
      throw bad_param("exporter/csv:file");
    }
  }
  // csv.cpp:159-206
  void write(const std::vector<std::vector<double>> &vars)
  {
    if (!enabled || !stream.is_open()) return;
    if (vars.size() != headers.size()) { log(0, "Variable list does not match header list"); return; }
    if (header_due && style != "none")
    {
      const bool mesh = style == "meshup";
      if (mesh) stream << "COLUMNS:" << std::endl;
      for (size_t i = 0; i < order.size(); ++i)
        for (size_t k = 0; k < vars[order[i]].size(); ++k)
        {
          stream << headers[order[i]] << "[" << k << "]";
          if (i + 1 < order.size() || k + 1 < vars[order[i]].size()) stream << ", ";
          if (mesh) stream << std::endl;
        }
      if (mesh) stream << "DATA:" << std::endl;
      else stream << std::endl;
    }
    header_due = false;
    for (size_t i = 0; i < order.size(); ++i)
      for (size_t k = 0; k < vars[order[i]].size(); ++k)
      {
        stream << std::fixed << std::setw(11) << std::setprecision(6) << vars[order[i]][k];
        if (i + 1 < order.size() || k + 1 < vars[order[i]].size()) stream << ", ";
      }
    stream << std::endl;
  }
  void close() { if (stream.is_open()) stream.close(); }
};
GRLX_REGISTER(CSVExporter)

// ------------------------------------------------------------- batch path ---
// The classes of the reference's tests/pendulum-fqi-ann.yaml (BASELINE configs[4]): descriptors, lowered to a grlx_fqi_config.
// projector/identity (projector.h:79-101): no parameters
struct IdentityProjector : Configurable { GRLX_TYPEINFO("projector/identity") };
GRLX_REGISTER(IdentityProjector)

// projector/pre/normalizing (normalizing.cpp:34-75)
struct NormalizingProjector : Configurable {
  GRLX_TYPEINFO("projector/pre/normalizing")
  VecD input_min, input_max;
  int signed_ = 0;
  Configurable *projector = nullptr;
  void request(const std::string &role, ConfigurationRequest *config) override
  {
    config->push_back(CRP("signed", "If true, project onto [-1, 1] instead of [0, 1]", 0));
    config->push_back(CRP("input_min", "Lower input dimension limit (for scaling)", VecD{}, CRP::System));
    config->push_back(CRP("input_max", "Upper input dimension limit (for scaling)", VecD{}, CRP::System));
    config->push_back(CRP("projector", "projector." + role, "Downstream projector", (Configurable *)nullptr));
  }
  void configure(Configuration &config) override
  {
    projector = config["projector"].ptr();
    input_min = config["input_min"].v(); input_max = config["input_max"].v();
    signed_ = config["signed"];
    if (signed_ != 0 && signed_ != 1) throw bad_param("projector/pre/normalizing:signed");
    if (input_min.size() != input_max.size()) throw bad_param("projector/normalizing:{input_min,input_max}");
    if (!dynamic_cast<IdentityProjector *>(projector)) throw Exception(path() + ": the accelerated path runs projector/pre/normalizing over projector/identity");
  }
};
GRLX_REGISTER(NormalizingProjector)

// representation/parameterized/ann (ann.cpp:36-96)
struct ANNRepresentation : Configurable {
  GRLX_TYPEINFO("representation/parameterized/ann")
  int inputs = 1, outputs = 1, interval = 0;
  VecD hiddens;
  double eta = 0.7, tau = 1;
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("interval", "Target network update interval", 0.));          // ParameterizedRepresentation (representation.h:173-184)
    config->push_back(CRP("tau", "Target network update rate", 1.));
    config->push_back(CRP("inputs", "Number of input dimensions", 1, CRP::System));
    config->push_back(CRP("outputs", "Number of output dimensions", 1, CRP::System));
    config->push_back(CRP("hiddens", "Number of hidden nodes per layer", VecD{5}));
    config->push_back(CRP("eta", "Learning rate (0=RPROP, <0=RMSPROP)", 0.7));
  }
  void configure(Configuration &config) override
  {
    inputs = config["inputs"]; outputs = config["outputs"]; hiddens = config["hiddens"].v(); eta = config["eta"];
    interval = (int)(double)config["interval"]; tau = config["tau"];
    if (inputs < 1) throw bad_param("representation/parameterized/ann:inputs");
    if (outputs < 1) throw bad_param("representation/parameterized/ann:outputs");
    for (double h : hiddens) if (std::round(h) <= 0) throw bad_param("representation/parameterized/ann:hiddens");     // ann.cpp:74-75
    if (eta < -2 || eta > 2) throw bad_param("representation/parameterized/ann:eta");
  }
};
GRLX_REGISTER(ANNRepresentation)

// representation/iterative (iterative.cpp:34-53)
struct IterativeRepresentation : Configurable {
  GRLX_TYPEINFO("representation/iterative")
  int epochs = 5000, cumulative = 1, batch_size = 0;
  Configurable *representation = nullptr;
  void request(const std::string &role, ConfigurationRequest *config) override
  {
    config->push_back(CRP("epochs", "Learning epochs", 5000));
    config->push_back(CRP("cumulative", "Add to training set instead of replacing it", 1));
    config->push_back(CRP("batch_size", "Batch size for gradient estimation (0=entire dataset)", 0));
    config->push_back(CRP("representation", "representation." + role, "Downstream representation", (Configurable *)nullptr));
  }
  void configure(Configuration &config) override
  {
    epochs = config["epochs"]; cumulative = config["cumulative"]; batch_size = config["batch_size"];
    representation = config["representation"].ptr();
    if (cumulative != 0 && cumulative != 1) throw bad_param("representation/iterative:cumulative");
    if (batch_size < 0) throw bad_param("representation/iterative:batch_size");
  }
};
GRLX_REGISTER(IterativeRepresentation)

// predictor/fqi (fqi.cpp:34-77)
struct FQIPredictor : Predictor {
  GRLX_TYPEINFO("predictor/fqi")
  double gamma = 0.97;
  int transitions = 100000, iterations = 10, macro_batch_size = 1;
  std::string reset_strategy = "iteration";
  UniformDiscretizer *discretizer = nullptr;
  NormalizingProjector *projector = nullptr;
  IterativeRepresentation *representation = nullptr;
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("importer", "importer", "Optional importer", (Configurable *)nullptr, true));     // Predictor::request
    config->push_back(CRP("exporter", "exporter", "Optional exporter", (Configurable *)nullptr, true));
    config->push_back(CRP("gamma", "Discount rate", 0.97));
    config->push_back(CRP("transitions", "Maximum number of transitions to store", 100000));
    config->push_back(CRP("iterations", "Number of policy improvement rounds per episode", 10));
    config->push_back(CRP("reset_strategy", "At which point to reset the representation", std::string("iteration")));
    config->push_back(CRP("macro_batch_size", "Number of episodes/batches after which prediction is rebuilt. Use 0 for no rebuilds.", 1));
    config->push_back(CRP("discretizer", "discretizer.action", "Action discretizer", (Configurable *)nullptr));
    config->push_back(CRP("projector", "projector.pair", "Projects observations onto critic representation space", (Configurable *)nullptr));
    config->push_back(CRP("representation", "representation.value/action", "Value function representation", (Configurable *)nullptr));
  }
  void configure(Configuration &config) override
  {
    gamma = config["gamma"]; transitions = config["transitions"]; iterations = config["iterations"];
    macro_batch_size = config["macro_batch_size"]; reset_strategy = config["reset_strategy"].str();
    discretizer = dynamic_cast<UniformDiscretizer *>(config["discretizer"].ptr());
    projector = dynamic_cast<NormalizingProjector *>(config["projector"].ptr());
    representation = dynamic_cast<IterativeRepresentation *>(config["representation"].ptr());
    if (transitions < 1) throw bad_param("predictor/fqi:transitions");
    if (iterations < 1) throw bad_param("predictor/fqi:iterations");
    if (reset_strategy != "never" && reset_strategy != "batch" && reset_strategy != "iteration") throw bad_param("predictor/fqi:reset_strategy");
    if (config["importer"].ptr() || config["exporter"].ptr()) throw Exception(path() + ": importer/exporter are outside the accelerated path");
    if (!discretizer || !projector || !representation)
      throw Exception(path() + ": the accelerated path needs discretizer/uniform, projector/pre/normalizing and representation/iterative");
  }
};
GRLX_REGISTER(FQIPredictor)

// experiment/batch_learning (batch_learning.cpp:36-205): batches of uniformly drawn transitions, FQIPredictor::rebuild after each, one greedy
// test trial per batch.  Lowered to a grlx_fqi_config and run by the kernels of grlx_fqi.hip (grlx_fqi_*): PARITY UNPINNED, oracle/fqi.c D1-D4.
struct BatchLearningExperimentImpl : Experiment {
  GRLX_TYPEINFO("experiment/batch_learning")
  int runs = 1, batches = 0, batch_size = 100;
  std::string output;
  Model *model = nullptr; Task *task = nullptr; FQIPredictor *predictor = nullptr; FixedAgent *test_agent = nullptr;
  VecD observation_min, observation_max, action_min, action_max;
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("runs", "Number of separate learning runs to perform", 1));
    config->push_back(CRP("batches", "Number of batches per learning run", 0));
    config->push_back(CRP("batch_size", "Number of transitions per batch", 100));
    config->push_back(CRP("rate", "Test trial control step frequency in Hz", 0, CRP::Online));
    config->push_back(CRP("output", "Output base filename", std::string()));
    config->push_back(CRP("model", "model", "Model in which the task is set", (Configurable *)nullptr));
    config->push_back(CRP("task", "task", "Task to be solved", (Configurable *)nullptr));
    config->push_back(CRP("predictor", "predictor", "Learner", (Configurable *)nullptr));
    config->push_back(CRP("test_agent", "agent", "Agent to use in test trials after each batch", (Configurable *)nullptr));
    config->push_back(CRP("observation_min", "Lower limit for observations", VecD{}, CRP::System));
    config->push_back(CRP("observation_max", "Upper limit for observations", VecD{}, CRP::System));
    config->push_back(CRP("action_min", "Lower limit for actions", VecD{}, CRP::System));
    config->push_back(CRP("action_max", "Upper limit for actions", VecD{}, CRP::System));
  }
  void configure(Configuration &config) override
  {
    runs = config["runs"]; batches = config["batches"]; batch_size = config["batch_size"]; output = config["output"].str();
    model = dynamic_cast<Model *>(config["model"].ptr());
    task = dynamic_cast<Task *>(config["task"].ptr());
    predictor = dynamic_cast<FQIPredictor *>(config["predictor"].ptr());
    test_agent = dynamic_cast<FixedAgent *>(config["test_agent"].ptr());
    observation_min = config["observation_min"].v(); observation_max = config["observation_max"].v();
    action_min = config["action_min"].v(); action_max = config["action_max"].v();
    if (runs < 1) throw bad_param("experiment/batch_learning:runs");
    if (batch_size < 1) throw bad_param("experiment/batch_learning:batch_size");
    if ((int)config["rate"] != 0) throw Exception(path() + ": rate (wall-clock pacing) is outside the accelerated path");
    if (!model || !task || !predictor || !test_agent)
      throw Exception(path() + ": the accelerated path needs model/dynamical, a task, predictor/fqi and agent/fixed");
  }

  // every assumption the batch kernels make is checked here; what they do not implement is refused, never emulated
  void lower(grlx_fqi_config *c) const
  {
    grlx_fqi_config_pendulum(c);
    const DynamicalModel *dm = dynamic_cast<const DynamicalModel *>(model);
    if (!dm || dm->env_id() != GRLX_ENV_PENDULUM || task->env_id() != GRLX_ENV_PENDULUM)
      throw Exception(path() + ": the batch path is built for model/dynamical with dynamics/pendulum and task/pendulum/swingup (the task must support invert(), pendulum.cpp:147-155)");
    if (task->randomization != 0) throw Exception(task->path() + ": randomization must be 0 on the batch path");
    c->control_step = dm->control_step; c->integration_steps = dm->integration_steps; c->timeout = task->timeout;
    // the experiment's sampling box must be the task's (the kernels draw observations and actions over the task's limits)
    const Configurator *t = task->configurator;
    auto task_vec = [&](const char *n) { return parse_vector(t->child(n)->value, n); };
    if (observation_min != task_vec("observation_min") || observation_max != task_vec("observation_max") ||
        action_min != task_vec("action_min") || action_max != task_vec("action_max"))
      throw Exception(path() + ": observation_min/max and action_min/max must be the task's limits on the accelerated path");
    const UniformDiscretizer *d = predictor->discretizer;
    if (d->min.size() != 1 || d->min != action_min || d->max != action_max) throw Exception(d->path() + ": one action dimension over the task's action range");
    c->action_min = d->min[0]; c->action_max = d->max[0]; c->action_steps = (int)d->steps[0];
    // projector/pre/normalizing over (observation ++ action).  The reference's yaml writes `observation_min+action_min`, which its
    // own parser adds element-wise with the scalar broadcast (parser.cpp:76-83) -- two entries for a three-dimensional input, which
    // NormalizingProjector::project then refuses (normalizing.cpp:78-83).  Deviation D4 (oracle/fqi.c): a limit
    // vector that does not have observation + action entries is replaced by the role's default, observation_min ++ action_min
    // (normalizing.cpp:50-51); one that has them must BE that concatenation.
    const NormalizingProjector *np = predictor->projector;
    VecD want_min = observation_min, want_max = observation_max;
    want_min.insert(want_min.end(), action_min.begin(), action_min.end());
    want_max.insert(want_max.end(), action_max.begin(), action_max.end());
    if (np->input_min.size() != want_min.size())
      log(1, np->path() + ": input_min / input_max have " + std::to_string(np->input_min.size()) + " entries for a " + std::to_string(want_min.size()) +
                 "-dimensional (observation, action) input; using observation_min++action_min, observation_max++action_max (deviation D4)");
    else if (np->input_min != want_min || np->input_max != want_max)
      throw Exception(np->path() + ": input_min / input_max must be the task's observation and action limits on the accelerated path");
    if (np->signed_ != 0) throw Exception(np->path() + ": signed = 1 is outside the accelerated path");
    // representation/iterative over representation/parameterized/ann
    const IterativeRepresentation *it = predictor->representation;
    const ANNRepresentation *ann = dynamic_cast<const ANNRepresentation *>(it->representation);
    if (!ann) throw Exception(it->path() + ": the accelerated path runs representation/iterative over representation/parameterized/ann");
    if (it->cumulative != 0 || it->batch_size != 0) throw Exception(it->path() + ": cumulative = 0 and batch_size = 0 (the whole data set per epoch) on the accelerated path");
    if (ann->inputs != (int)want_min.size() || ann->outputs != 1) throw Exception(ann->path() + ": inputs must be observation_dims+action_dims and outputs 1");
    if (ann->hiddens.size() != 1) throw Exception(ann->path() + ": one hidden layer on the accelerated path");
    if (ann->interval != 0) throw Exception(ann->path() + ": a target network (interval) is outside the accelerated path");
    c->hidden = (int)std::round(ann->hiddens[0]);
    c->eta = ann->eta;
    c->epochs = it->epochs;
    c->gamma = predictor->gamma; c->iterations = predictor->iterations;
    if (predictor->reset_strategy != "never") throw Exception(predictor->path() + ": reset_strategy must be never on the accelerated path");
    if (predictor->macro_batch_size != 1) throw Exception(predictor->path() + ": macro_batch_size must be 1 on the accelerated path");
    // the test agent: agent/fixed with the greedy Q policy over the predictor's own discretizer, projector and representation
    const QPolicy *tp = dynamic_cast<const QPolicy *>(test_agent->policy);
    if (!tp || tp->discretizer != d || tp->any_projector != np || tp->any_representation != it || tp->sampler->explores())
      throw Exception(test_agent->path() + ": the test agent must be agent/fixed with policy/discrete/q over the predictor's discretizer, projector and representation and sampler/greedy");
    c->batch_size = batch_size;
    if (batches < 1) throw Exception(path() + ": batches must be > 0 (the reference's batches: 0 runs forever)");
    c->max_batches = batches;
    if ((long long)batches * batch_size > (long long)predictor->transitions)
      throw Exception(predictor->path() + ": transitions (the store's size) is smaller than batches x batch_size; the accelerated path keeps every transition");
  }

  std::vector<double> run(const RunOptions &opt) override
  {
    if (runs != 1) throw Exception(path() + ": runs > 1 (Experiment::reset between runs) is not built for the batch path");
    grlx_fqi_config c;
    lower(&c);
    c.n_replicas = opt.replicas;
    if (!output.empty())
    {
      std::ofstream ofs(output + ".yaml");
      ofs << configurator->root()->yaml();
    }
    std::vector<int64_t> seeds((size_t)opt.replicas);
    for (int i = 0; i < opt.replicas; ++i) seeds[(size_t)i] = opt.seed + i;
    grlx_fqi_ctx *ctx = nullptr;
    if (grlx_fqi_create(&c, seeds.data(), &ctx) != GRLX_OK) throw Exception(grlx_last_error());
    std::vector<std::ofstream> files((size_t)opt.replicas);
    if (!output.empty())
      for (int i = 0; i < opt.replicas; ++i)
      { // <output>-<run>.txt (batch_learning.cpp:96-101); clones carry the identity "@i" (multi.cpp:52-56)
        std::ostringstream name;
        name << output << "-" << 0;
        if (opt.replicas > 1) name << "@" << i;
        name << ".txt";
        files[(size_t)i].open(name.str());
      }
    std::vector<double> curve;
    for (int bb = 0; bb < batches; ++bb)
    { // the rows are written batch by batch, as the reference does (a run can be watched)
      auto start = std::chrono::steady_clock::now();
      int rc = grlx_fqi_run_batch(ctx, nullptr);
      if (rc == GRLX_OK) rc = grlx_fqi_sync(ctx, nullptr);
      if (rc != GRLX_OK) { std::string e = grlx_last_error(); grlx_fqi_destroy(ctx); throw Exception(e); }
      const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
      for (int i = 0; i < opt.replicas; ++i)
      {
        int64_t batch = 0, transitions = 0;
        double reward = 0;
        if (grlx_fqi_read_rows(ctx, i, bb, 1, &batch, &transitions, &reward) != GRLX_OK) { std::string e = grlx_last_error(); grlx_fqi_destroy(ctx); throw Exception(e); }
        std::ostringstream oss;                          // batch_learning.cpp:179
        oss << std::setw(15) << batch << std::setw(15) << transitions << std::setw(15) << reward;
        if (files[(size_t)i].is_open()) files[(size_t)i] << oss.str() << std::endl;
        if (i == 0 && opt.print_rows) std::cout << oss.str() << std::endl;
        if (i == 0) curve.push_back(reward);
      }
      int32_t its = 0;
      grlx_fqi_info(ctx, 0, nullptr, nullptr, &its, nullptr, nullptr);
      std::ostringstream msg;
      msg << "batch " << bb << ": " << opt.replicas << " replicas, " << (bb + 1) * batch_size << " stored transitions, " << its << " iterations x " << c.epochs
          << " epochs (replica 0) in " << wall << " s";
      log(2, msg.str());
    }
    grlx_fqi_destroy(ctx);
    return curve;
  }
};
GRLX_REGISTER(BatchLearningExperimentImpl)

// ------------------------------------------------------------- experiment ---
// The experiment's objects stepped by the caller: thin forwards to the per-step entry points of the C ABI on the experiment's context
struct ContextEnvironment : StepwiseEnvironment {
  grlx_ctx *ctx = nullptr;
  void start(int test, const int32_t *active, double *obs) override
  { if (grlx_env_start(ctx, test, active, obs) != GRLX_OK) throw Exception(std::string("environment/modeled: ") + grlx_last_error()); }
  void step(const int32_t *active, const double *action, double *obs, double *reward, int32_t *terminal) override
  { if (grlx_env_advance(ctx, active, action, obs, reward, terminal) != GRLX_OK) throw Exception(std::string("environment/modeled: ") + grlx_last_error()); }
};
struct ContextAgent : StepwiseAgent {
  grlx_ctx *ctx = nullptr;
  int test = 0;                                     // 0: agent/td, 1: agent/fixed (the test agent)
  void start(const int32_t *active, const double *obs, double *action) override
  { if (grlx_agent_start(ctx, test, active, obs, action) != GRLX_OK) throw Exception(std::string("agent: ") + grlx_last_error()); }
  void step(const int32_t *active, double tau, const double *obs, const double *reward, const int32_t *terminal, double *action) override
  { if (grlx_agent_step(ctx, test, active, tau, obs, reward, terminal, action) != GRLX_OK) throw Exception(std::string("agent: ") + grlx_last_error()); }
  void end(const int32_t *active, double tau, const double *obs, const double *reward) override
  { if (grlx_agent_end(ctx, test, active, tau, obs, reward) != GRLX_OK) throw Exception(std::string("agent: ") + grlx_last_error()); }
};

struct OnlineLearningExperimentImpl : OnlineLearningExperiment, StepwiseExperiment {
  GRLX_TYPEINFO("experiment/online_learning")
  // ---- StepwiseExperiment: the graph instantiated on the GPU, its environment and agents stepped by the caller
  grlx_ctx *step_ctx = nullptr;
  grlx_config step_cfg;
  ContextEnvironment step_env;
  ContextAgent step_agent, step_test_agent;
  ~OnlineLearningExperimentImpl() override { close(); }
  void open(const RunOptions &opt) override
  {
    close();
    lower(&step_cfg);
    step_cfg.n_replicas = opt.replicas;
    if (opt.table_log2_capacity) step_cfg.table_log2_capacity = opt.table_log2_capacity;
    std::vector<int64_t> seeds((size_t)opt.replicas);
    for (int i = 0; i < opt.replicas; ++i) seeds[(size_t)i] = opt.seed + i;
    if (grlx_create(&step_cfg, seeds.data(), &step_ctx) != GRLX_OK) { step_ctx = nullptr; throw Exception(grlx_last_error()); }
    step_env.ctx = step_agent.ctx = step_test_agent.ctx = step_ctx;
    step_agent.test = 0;
    step_test_agent.test = 1;
  }
  void close() override
  {
    if (step_ctx) grlx_destroy(step_ctx);
    step_ctx = nullptr;
  }
  int replicas() const override { return step_cfg.n_replicas; }
  int obs_dims() const override { int s = 0, d = 0; grlx_env_dims(step_cfg.env, &s, &d); return d; }
  int test_interval_of() const override { return test_interval; }
  StepwiseEnvironment *stepwise_environment() override { return &step_env; }
  StepwiseAgent *stepwise_agent() override { return &step_agent; }
  StepwiseAgent *stepwise_test_agent() override { return &step_test_agent; }

  int runs = 1, run_offset = 0, trials = 0, steps = 0, test_interval = -1, test_trials = 1;
  std::string output, load_file, save_every;
  ModeledEnvironment *environment = nullptr; TDAgent *agent = nullptr; FixedAgent *test_agent = nullptr;
  CSVExporter *exporter = nullptr, *env_exporter = nullptr;

  void request(const std::string &, ConfigurationRequest *config) override
  { // online_learning.cpp:40-62
    config->push_back(CRP("runs", "Number of separate learning runs to perform", 1));
    config->push_back(CRP("run_offset", "Run offset to start at", 0));
    config->push_back(CRP("trials", "Number of episodes per learning run", 0));
    config->push_back(CRP("steps", "Number of steps per learning run", 0));
    config->push_back(CRP("rate", "Control step frequency in Hz", 0, CRP::Online));
    config->push_back(CRP("test_interval", "Number of episodes in between test trials", -1));
    config->push_back(CRP("test_trials", "Number of test trials per interval", 1));
    config->push_back(CRP("output", "Output base filename", std::string()));
    config->push_back(CRP("environment", "environment", "Environment in which the agent acts", (Configurable *)nullptr));
    config->push_back(CRP("agent", "agent", "Agent", (Configurable *)nullptr));
    config->push_back(CRP("test_agent", "agent", "Agent to use in test trials", (Configurable *)nullptr, true));
    config->push_back(CRP("exporter", "exporter", "Optional exporter for transition log", (Configurable *)nullptr, true));
    config->push_back(CRP("load_file", "Load policy filename", std::string()));
    config->push_back(CRP("save_every", "Save policy to 'output' at the end of event", std::string("never")));
  }
  void configure(Configuration &config) override
  {
    runs = config["runs"]; run_offset = config["run_offset"]; trials = config["trials"]; steps = config["steps"];
    test_interval = config["test_interval"]; test_trials = config["test_trials"];
    output = config["output"].str(); load_file = config["load_file"].str(); save_every = config["save_every"].str();
    environment = dynamic_cast<ModeledEnvironment *>(config["environment"].ptr());
    agent = dynamic_cast<TDAgent *>(config["agent"].ptr());
    test_agent = dynamic_cast<FixedAgent *>(config["test_agent"].ptr());
    if (runs < 1) throw bad_param("experiment/online_learning:runs");
    if (test_interval < -1) throw bad_param("experiment/online_learning:test_interval");
    if (test_interval >= 0 && !config["test_agent"].ptr()) throw bad_param("experiment/online_learning:test_agent");   // :100-101
    if (!environment || !agent || (config["test_agent"].ptr() && !test_agent))
      throw Exception(path() + ": the accelerated path needs environment/modeled, agent/td and agent/fixed");
    if (save_every != "never" && save_every != "run" && save_every != "test" && save_every != "trial")
      throw bad_param("experiment/online_learning:save_every");
    if (steps < 0) throw bad_param("experiment/online_learning:steps");
    if (test_trials < 1) throw bad_param("experiment/online_learning:test_trials");
    if ((int)config["rate"] != 0)
      throw Exception(path() + ": rate (wall-clock pacing) is outside the accelerated path");
    exporter = dynamic_cast<CSVExporter *>(config["exporter"].ptr());
    if (config["exporter"].ptr() && !exporter) throw Exception(path() + ": only exporter/csv is available on the accelerated path");
    if (exporter) exporter->init({"time", "observation", "action", "reward", "terminal"});      // online_learning.cpp:74-75
    env_exporter = dynamic_cast<CSVExporter *>(environment->exporter);
    if (environment->exporter && !env_exporter) throw Exception(environment->path() + ": only exporter/csv is available on the accelerated path");
    if (env_exporter) env_exporter->init({"time", "state", "observation", "action", "reward", "terminal"});   // modeled.cpp:70-71
  }

  // lower the instantiated graph to the C ABI's grlx_config; every assumption the fused kernels make is checked
  static void lower_tile(const TileCodingProjector *p, grlx_tile_spec *t)
  {
    memset(t, 0, sizeof(*t));
    t->tilings = p->tilings; t->memory = p->memory; t->dims = (int)p->resolution.size();
    t->safe = p->safe;
    if (p->resolution.size() > GRLX_MAX_DIMS) throw bad_param("projector/tile_coding:resolution");
    for (size_t i = 0; i < p->resolution.size(); ++i) { t->resolution[i] = p->resolution[i]; t->wrapping[i] = p->wrapping[i]; }
  }
  static void lower_linear(const LinearRepresentation *r, const TileCodingProjector *p, grlx_linear_spec *l)
  {
    if (r->outputs != 1) throw Exception(r->path() + ": outputs must be 1");
    if (r->memory != p->memory) throw bad_param("representation/parameterized/linear:memory (or matching projector)");
    l->init_min = r->init_min[0]; l->init_max = r->init_max[0];
    l->output_min = r->output_min[0]; l->output_max = r->output_max[0];
    l->limit = r->limit;
  }
  int order_of(const Configurable *o) const
  {
    int k = 0;
    for (Configurator *n : instantiate_order()) { if (n->object.get() == o) return k; ++k; }
    return -1;
  }

  void lower(grlx_config *c) const
  {
    grlx_config_pendulum_sarsa(c);
    c->test_interval = test_interval;
    c->test_trials = test_trials;
    environment->lower_env(c);

    if (const ActionACPredictor *ac = dynamic_cast<const ActionACPredictor *>(agent->predictor))
    { // ---- actor-critic (cfg/cart_pole/ac_tc.yaml)
      const ActionPolicy *pol = dynamic_cast<const ActionPolicy *>(agent->policy);
      const ActionPolicy *tpol = test_agent ? dynamic_cast<const ActionPolicy *>(test_agent->policy) : nullptr;
      if (!pol || (test_agent && !tpol)) throw Exception(path() + ": predictor/ac/action needs mapping/policy/action policies");
      if (ac->projector != pol->projector || ac->representation != pol->representation)
        throw Exception(ac->path() + ": actor predictor and policy must share projector and representation");
      if (tpol && (tpol->projector != pol->projector || tpol->representation != pol->representation))
        throw Exception(test_agent->path() + ": the test policy must share projector and representation with the learning policy");
      if (tpol && tpol->sigma[0] != 0) throw Exception(tpol->path() + ": the test policy must be noise-free (sigma: [])");
      const VPredictor *cr = ac->critic;
      if (cr->representation == pol->representation) throw Exception(cr->path() + ": actor and critic need separate tables");
      // both tables draw their initial weights from one thread-local stream: actor first
      if (!(order_of(pol->representation) >= 0 && order_of(pol->representation) < order_of(cr->representation)))
        throw Exception(path() + ": this yaml instantiates the actor/critic tables in an order the fused kernel does not reproduce");
      c->agent = GRLX_AGENT_AC;
      c->action_min = pol->min[0]; c->action_max = pol->max[0]; c->action_steps = 0;
      lower_tile(cr->projector, &c->projector);
      lower_linear(cr->representation, cr->projector, &c->representation);
      lower_tile(pol->projector, &c->actor_projector);
      lower_linear(pol->representation, pol->projector, &c->actor_representation);
      if (cr->representation->interval || pol->representation->interval)
        throw Exception(path() + ": target networks (interval) are built for predictor/critic/sarsa and predictor/critic/q only");
      c->alpha = cr->alpha; c->gamma = cr->gamma; c->lambda = cr->lambda;
      c->trace = cr->trace ? cr->trace->kind() : GRLX_TRACE_NONE;
      c->actor_alpha = ac->alpha;
      c->sigma = pol->sigma[0]; c->theta = pol->theta[0];
      c->ac_decay_rate = pol->decay_rate; c->ac_decay_min = pol->decay_min;
      c->ac_update_method = ac->update_method[0] == 'p' ? 0 : 1;
      c->ac_step_limit = ac->step_limit.empty() ? -1. : ac->step_limit[0];
      c->table_log2_capacity = 16;                   // initial size: the tables grow between launches (grlx_config_cart_pole_ac)
      return;
    }

    // ---- discrete Q policies (cfg/pendulum/sarsa_tc.yaml, q_tc.yaml)
    const QPolicy *pol = dynamic_cast<const QPolicy *>(agent->policy);
    const TDPredictorBase *pred = dynamic_cast<const TDPredictorBase *>(agent->predictor);
    const QPolicy *tpol = test_agent ? dynamic_cast<const QPolicy *>(test_agent->policy) : nullptr;
    if (!pol || !pred || (test_agent && !tpol))
      throw Exception(path() + ": the accelerated path needs mapping/policy/discrete/value/q with predictor/critic/sarsa|q, or policy/action with predictor/ac/action");
    if (!pol->projector || !pol->representation || (tpol && (!tpol->projector || !tpol->representation)))
      throw Exception(path() + ": experiment/online_learning runs projector/tile_coding with representation/parameterized/linear on the accelerated path");
    if (pred->projector != pol->projector || pred->representation != pol->representation)
      throw Exception(pred->path() + ": predictor and policy must share projector and representation on the accelerated path");
    if (const ExpectedSARSAPredictor *es = dynamic_cast<const ExpectedSARSAPredictor *>(pred))
      if (es->target_policy != pol) throw Exception(es->path() + ": the target policy must be the agent's own policy on the accelerated path");
    if (tpol && (tpol->projector != pol->projector || tpol->representation != pol->representation || tpol->discretizer != pol->discretizer))
      throw Exception(test_agent->path() + ": the test policy must share discretizer, projector and representation with the learning policy");
    if (!pol->sampler->explores() || (tpol && tpol->sampler->explores()))
      throw Exception(path() + ": the accelerated path needs sampler/epsilon_greedy for learning and sampler/greedy for testing");
    // RNG streams are consumed in instantiate order: representation, learning sampler, test sampler
    const int at_repr = order_of(pol->representation), at_s1 = order_of(pol->sampler), at_s2 = tpol ? order_of(tpol->sampler) : -1;
    if (!(at_repr >= 0 && at_repr < at_s1 && (!tpol || at_s1 < at_s2)))
      throw Exception(path() + ": this yaml instantiates representation/samplers in an order the fused kernel does not reproduce");
    const UniformDiscretizer *d = pol->discretizer;
    if (d->min.size() != 1) throw Exception(d->path() + ": one action dimension supported");
    c->action_min = d->min[0]; c->action_max = d->max[0]; c->action_steps = (int)d->steps[0];
    lower_tile(pol->projector, &c->projector);
    lower_linear(pol->representation, pol->projector, &c->representation);
    const EpsilonGreedySampler *sm = static_cast<const EpsilonGreedySampler *>(pol->sampler);
    c->epsilon = sm->epsilon[0]; c->decay_rate = sm->decay_rate; c->decay_min = sm->decay_min;
    c->agent = pred->agent_id();
    c->target_interval = pol->representation->interval;              // target network of the Q table (representation.h:161-306)
    c->target_tau = pol->representation->tau;
    c->alpha = pred->alpha; c->gamma = pred->gamma; c->lambda = pred->lambda; c->kappa = pred->kappa;
    c->trace = pred->trace ? pred->trace->kind() : GRLX_TRACE_NONE;
    if (const QVPredictor *qv = dynamic_cast<const QVPredictor *>(pred))
    { // second table: V(s); its weights are drawn from the thread-local stream after the Q table's
      if (!(order_of(qv->v_representation) > at_repr))
        throw Exception(qv->path() + ": v_representation must be instantiated after the policy's representation");
      lower_tile(qv->v_projector, &c->actor_projector);
      lower_linear(qv->v_representation, qv->v_projector, &c->actor_representation);
      c->beta = qv->beta;
    }
  }

  // the parameterized representations of the agent in table order: 0 = Q / critic, 1 = actor / V
  std::vector<const Configurable *> representations() const
  {
    std::vector<const Configurable *> reprs;
    if (const ActionACPredictor *ac = dynamic_cast<const ActionACPredictor *>(agent->predictor))
    { reprs.push_back(ac->critic->representation); reprs.push_back(ac->representation); }
    else
    {
      reprs.push_back(dynamic_cast<const QPolicy *>(agent->policy)->representation);
      if (const QVPredictor *qv = dynamic_cast<const QVPredictor *>(agent->predictor)) reprs.push_back(qv->v_representation);
    }
    return reprs;
  }

  // ParameterizedRepresentation {action: save} (representation.h:201-229) of every representation of the agent: raw double[memory] to
  // <base>[@clone]-<config path with '/'->'_'>.dat
  void save_policy(grlx_ctx *ctx, const grlx_config &c, const std::string &base, int replicas)
  {
    const std::vector<const Configurable *> reprs = representations();
    for (size_t tb = 0; tb < reprs.size(); ++tb)
      for (int i = 0; i < replicas; ++i)
      {
        const int memory = tb == 1 ? c.actor_projector.memory : c.projector.memory;
        std::vector<double> dense((size_t)memory);
        if (grlx_export_weights(ctx, (int)tb, i, dense.data()) != GRLX_OK) { std::string e = grlx_last_error(); grlx_destroy(ctx); throw Exception(e); }
        std::string cfg_path = reprs[tb]->path();
        std::replace(cfg_path.begin(), cfg_path.end(), '/', '_');
        std::ostringstream name;
        name << base;
        if (replicas > 1) name << "@" << i;
        name << "-" << cfg_path << ".dat";
        std::ofstream f(name.str(), std::ios::binary);
        if (!f) { log(1, "Could not open '" + name.str() + "' for writing"); continue; }
        f.write(reinterpret_cast<const char *>(dense.data()), (std::streamsize)(dense.size() * sizeof(double)));
      }
  }

  std::vector<double> run(const RunOptions &opt) override
  {
    if (trials <= 0 && steps <= 0) throw Exception(path() + ": trials or steps must be > 0 (the reference's trials: 0, steps: 0 runs forever)");
    // the trial loop of online_learning.cpp:154 ends a run at the first trial boundary with `ss >= steps`, and save_every: test | trial
    // writes the policy between trials: both need the host between trials, so such runs launch one trial at a time
    const bool by_trial = steps > 0 || save_every == "test" || save_every == "trial";
    if (steps > 0 && opt.replicas * opt.world > 1)
      throw Exception(path() + ": a steps budget ends every clone at a trial of its own; on the accelerated path it is built for one replica");
    // rows a run can write: with a steps budget alone, a trial has at least one learning step
    const int trial_cap = trials > 0 ? trials : steps * (test_interval >= 0 ? 2 : 1) + 1;
    grlx_config c;
    lower(&c);
    c.n_replicas = opt.replicas;
    c.max_rows = (test_interval >= 0 ? trial_cap / (test_interval + 1) : trial_cap) + 1;
    if (opt.table_log2_capacity) c.table_log2_capacity = opt.table_log2_capacity;
    std::vector<double> curve;
    int obs_dims = 0, state_dims = 0;
    grlx_env_dims(c.env, &state_dims, &obs_dims);
    if (exporter || env_exporter)
    { // transition log of replica 0: every step and every trial start is tapped on the device and
      // written after the run (the reference writes one row per step, online_learning.cpp:183-206)
      const double per_trial = std::floor(c.timeout / c.control_step) + 3;
      const double want = per_trial * (double)trial_cap;
      if (want > 4e6) throw Exception(path() + ": transition log of " + std::to_string((long long)want) + " rows is too large for the device-side tap buffer");
      c.tap_replica = 0;
      c.tap_capacity = (int)want;
      c.tap_starts = 1;
    }

    if (!output.empty())
    { // store the resolved configuration with the output (online_learning.cpp:117-122)
      std::ofstream ofs(output + ".yaml");
      ofs << configurator->root()->yaml();
    }
    // ONE instantiation for all runs, as in the reference: between two runs the experiment is reset (online_learning.cpp:307-308),
    // not re-created -- the random streams continue, so run 1 differs from a fresh process with another seed
    const int first_clone = opt.rank * opt.replicas;              // `grlxd -g N`: this rank's clones among those of the whole job
    const bool many = opt.replicas * opt.world > 1;
    std::vector<int64_t> seeds((size_t)opt.replicas);
    for (int i = 0; i < opt.replicas; ++i) seeds[(size_t)i] = opt.seed + first_clone + i;
    grlx_ctx *ctx = nullptr;
    if (grlx_create(&c, seeds.data(), &ctx) != GRLX_OK) throw Exception(grlx_last_error());
    for (int rr = run_offset; rr < runs + run_offset; ++rr)
    {
      // Load policy every run (online_learning.cpp:140-150 -> ParameterizedRepresentation {action: load},
      // representation.h:231-263): <load_file with $run -> rr>-<config path with '/'->'_'>.dat, raw
      // double[memory]; a missing or wrong-sized file is a warning, not an error, as in the reference.
      // All clones of a run read the same file (multi.cpp does not touch load_file).
      if (!load_file.empty())
      {
        std::string base = load_file + "-";
        for (size_t at; (at = base.find("$run")) != std::string::npos;) base.replace(at, 4, std::to_string(rr));
        log(2, "Loading policy: " + base);
        const std::vector<const Configurable *> reprs = representations();
        for (size_t tb = 0; tb < reprs.size(); ++tb)
        {
          std::string cfg_path = reprs[tb]->path();
          std::replace(cfg_path.begin(), cfg_path.end(), '/', '_');
          const std::string file = base + cfg_path + ".dat";
          const size_t memory = (size_t)(tb == 1 ? c.actor_projector.memory : c.projector.memory);
          std::ifstream f(file, std::ios::binary | std::ios::ate);
          if (!f) { log(1, "Could not open '" + file + "' for reading"); continue; }
          if ((size_t)f.tellg() != memory * sizeof(double)) { log(1, "Configuration mismatch for '" + file + "'"); continue; }
          std::vector<double> dense(memory);
          f.seekg(0);
          if (!f.read(reinterpret_cast<char *>(dense.data()), (std::streamsize)(memory * sizeof(double)))) { log(1, "Could not read '" + file + "'"); continue; }
          if (grlx_load_weights(ctx, (int)tb, 0, opt.replicas, dense.data(), (uint64_t)memory) != GRLX_OK)
          { std::string e = grlx_last_error(); grlx_destroy(ctx); throw Exception(e); }
        }
      }
      auto start = std::chrono::steady_clock::now();
      int rc = GRLX_OK;
      if (!by_trial)
      {
        rc = grlx_run(ctx, trials, nullptr);
        if (rc == GRLX_OK) rc = grlx_sync(ctx, nullptr);
      }
      else
      { // for (ss = 0, tt = 0; (!trials || tt < trials) && (!steps || ss < steps); ++tt)   -- online_learning.cpp:154
        uint64_t learn0 = 0, test0 = 0;
        grlx_step_counts(ctx, &learn0, &test0);
        uint64_t ss = 0;
        for (int tt = 0; (trials <= 0 || tt < trials) && (steps <= 0 || ss < (uint64_t)steps) && rc == GRLX_OK; ++tt)
        {
          rc = grlx_run(ctx, 1, nullptr);
          if (rc == GRLX_OK) rc = grlx_sync(ctx, nullptr);
          if (rc != GRLX_OK) break;
          uint64_t learn = 0, test = 0;
          grlx_step_counts(ctx, &learn, &test);
          ss = learn - learn0;                         // one replica when a budget is set; unused otherwise
          const bool was_test = test_interval >= 0 && tt % (test_interval + 1) == test_interval;
          if ((save_every == "trial" || (was_test && save_every == "test")) && !output.empty())
          { // online_learning.cpp:281-290: <output>-run<rr>-trial<tt>-<config path>.dat
            std::ostringstream base;
            base << output << "-run" << rr << "-trial" << tt;
            save_policy(ctx, c, base.str(), opt.replicas);
          }
        }
      }
      if (rc != GRLX_OK) { std::string e = grlx_last_error(); grlx_destroy(ctx); throw Exception(e); }
      double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();

      std::vector<grlx_tap> taps;
      int ntaps = 0;
      if (exporter || env_exporter)
      {
        taps.resize((size_t)c.tap_capacity);
        if (grlx_read_taps(ctx, taps.data(), c.tap_capacity, &ntaps) != GRLX_OK) { std::string e = grlx_last_error(); grlx_destroy(ctx); throw Exception(e); }
        if (ntaps == c.tap_capacity) log(1, "transition log truncated at " + std::to_string(ntaps) + " rows");
      }
      if (env_exporter)
      { // ModeledEnvironment's own log (modeled.cpp:156-157, 200-203): one row per step with the state the step started
        // from and the environment's cumulative learn / test time BEFORE the step; a numbered file per variant is
        // started by the first episode of that variant, later episodes append
        double time_learn = 0, time_test = 0, applied_time_step = c.control_step;
        std::vector<double> before((size_t)state_dims, 0.);
        bool is_test = false;
        for (int k = 0; k < ntaps; ++k)
        {
          const grlx_tap &tp = taps[(size_t)k];
          if (tp.terminal == -1)
          {
            is_test = tp.test != 0;
            env_exporter->open(is_test ? "test" : "learn", (is_test ? time_test : time_learn) != 0.0);
          }
          else
          {
            double &time = is_test ? time_test : time_learn;
            const std::vector<double> obs(tp.obs, tp.obs + obs_dims);
            // the action of this row is the one the step was taken with: the previous record's
            env_exporter->write({{time}, before, obs, {taps[(size_t)k - 1].action}, {tp.reward}, {(double)tp.terminal}});
            time += applied_time_step;                 // model tau = control_step (modeled.cpp:176, 203)
          }
          before.assign(tp.state, tp.state + state_dims);
        }
        env_exporter->close();
      }
      if (exporter)
      { // online_learning.cpp:127-131 (new numbered files per run), :164-165 (append per trial), :183-206 (rows)
        exporter->open("test", false);
        exporter->open("learn", false);
        double total_time = 0, applied = 0;
        for (int k = 0; k < ntaps; ++k)
        {
          const grlx_tap &tp = taps[(size_t)k];
          const std::vector<double> obs(tp.obs, tp.obs + obs_dims);
          if (tp.terminal == -1)
          { // start of a trial: first observation, first action
            exporter->open(tp.test ? "test" : "learn", true);
            total_time = 0;
            exporter->write({{total_time}, obs, {tp.action}, {0.}, {0.}});
          }
          else
          { // a step: the action that was applied, the observation and reward it led to
            total_time += 1;                          // tau = 1 (discrete_time)
            exporter->write({{total_time}, obs, {applied}, {tp.reward}, {(double)tp.terminal}});
          }
          applied = tp.action;
        }
        exporter->close();
      }

      const int n = grlx_rows(ctx);
      std::vector<int64_t> trial((size_t)n), stp((size_t)n);
      std::vector<double> reward((size_t)n);
      std::vector<double> episode_time((size_t)n);     // total_time: sum of tau = steps of the trial (discrete_time, modeled.cpp:209-212)
      for (int i = 0; i < opt.replicas; ++i)
      {
        grlx_read_rows(ctx, i, 0, n, trial.data(), stp.data(), reward.data());
        grlx_read_row_times(ctx, i, 0, n, episode_time.data());
        std::ostringstream name;                       // <output>-<run><identity>.txt, identity "@i" for clones (multi.cpp:52-56)
        name << output << "-" << rr;
        if (many) name << "@" << first_clone + i;
        name << ".txt";
        std::ofstream ofs;
        if (!output.empty()) ofs.open(name.str());
        for (int k = 0; k < n; ++k)
        {
          std::ostringstream oss;
          if (opt.legacy_rows)                         // the 3-column layout of the reference's committed golden files
            oss << std::setw(15) << trial[(size_t)k] << std::setw(15) << stp[(size_t)k] << std::setw(15) << reward[(size_t)k];
          else                                         // online_learning.cpp:243
            oss << std::setw(15) << trial[(size_t)k] << std::setw(15) << stp[(size_t)k] << std::setw(15) << std::setprecision(3) << std::fixed
                << reward[(size_t)k] << std::setw(15) << std::setprecision(3) << episode_time[(size_t)k] << std::setw(15) << std::setprecision(3)
                << reward[(size_t)k] / episode_time[(size_t)k] << std::setw(15) << std::setprecision(3) << wall;
          if (ofs.is_open()) ofs << oss.str() << std::endl;
          if (i == 0 && opt.rank == 0 && opt.print_rows) std::cout << oss.str() << std::endl;
        }
        if (i == 0) curve = reward;
      }
      if (opt.reducer && n > 0)
      { // the learning curve over the clones of ALL ranks: per row {sum, sum of squares, count} of this device's replicas (grlx_curve_stats,
        // fixed reduction order), ONE all-reduce over the ranks, mean and standard deviation on rank 0.  (Trial-bounded runs write the same
        // rows on every rank; a steps budget with more than one clone is refused above.)
        double *dev = opt.reducer->device_buffer((size_t)n * 3);
        if (grlx_curve_stats(ctx, 0, n, dev, nullptr) != GRLX_OK || grlx_sync(ctx, nullptr) != GRLX_OK)
        { std::string e = grlx_last_error(); grlx_destroy(ctx); throw Exception(e); }
        opt.reducer->all_reduce_sum(dev, (size_t)n * 3);
        std::vector<double> st((size_t)n * 3);
        opt.reducer->to_host(st.data(), dev, st.size());
        if (opt.rank == 0 && !output.empty())
        {
          std::ostringstream name;
          name << output << "-" << rr << "-mean.txt";
          std::ofstream ofs(name.str());
          grlx_read_rows(ctx, 0, 0, n, trial.data(), stp.data(), reward.data());
          for (int k = 0; k < n; ++k)
          { // trial, clones that wrote the row, mean return, standard deviation over the clones (population)
            const double sum = st[(size_t)k * 3], sq = st[(size_t)k * 3 + 1], cntk = st[(size_t)k * 3 + 2];
            const double mean = cntk > 0 ? sum / cntk : 0., var = cntk > 0 ? std::max(0., sq / cntk - mean * mean) : 0.;
            ofs << std::setw(15) << trial[(size_t)k] << std::setw(15) << (long long)cntk << std::setw(20) << std::setprecision(12) << mean
                << std::setw(20) << std::setprecision(12) << std::sqrt(var) << std::endl;
          }
        }
      }
      // Save policy every run (online_learning.cpp:294-302 -> ParameterizedRepresentation {action: save},
      // representation.h:201-229): raw double[memory] to <output>-run<rr>-<config path with '/'->'_'>.dat
      if (save_every == "run" && !output.empty())
      {
        std::ostringstream base;
        base << output << "-run" << rr;
        save_policy(ctx, c, base.str(), opt.replicas);
      }
      uint64_t learn = 0, test = 0;
      grlx_step_counts(ctx, &learn, &test);
      std::ostringstream msg;
      msg << "run " << rr << ": " << opt.replicas << " replicas, " << (learn + test) << " env-steps in " << wall << " s = "
          << (double)(learn + test) / wall / 1e6 << " M env-steps/s";
      log(2, msg.str());
      if (rr < runs + run_offset - 1 && grlx_reset_run(ctx) != GRLX_OK)        // online_learning.cpp:307-308
      { std::string e = grlx_last_error(); grlx_destroy(ctx); throw Exception(e); }
    }
    grlx_destroy(ctx);
    return curve;
  }
};
GRLX_REGISTER(OnlineLearningExperimentImpl)

// experiment/multi (multi.cpp:36-75): `instances` clones of an experiment running side by side, each
// with the identity "@i" in its output names.  Here the clones are the replicas of ONE device
// context (instance i is seeded seed + i; the reference's clones draw their seeds from the shared
// global stream in instantiation order and then race on it, which no golden file pins).
struct MultiExperiment : OnlineLearningExperiment {
  GRLX_TYPEINFO("experiment/multi")
  int instances = 1;
  OnlineLearningExperiment *prototype = nullptr;
  void request(const std::string &, ConfigurationRequest *config) override
  {
    config->push_back(CRP("instances", "Number of experiments to run in parallel", instances));
    config->push_back(CRP("experiment", "experiment", "Experiment to run", (Configurable *)nullptr));
  }
  void configure(Configuration &config) override
  {
    instances = config["instances"];
    prototype = dynamic_cast<OnlineLearningExperiment *>(config["experiment"].ptr());
    if (instances < 1) throw bad_param("experiment/multi:instances");
    if (!prototype || dynamic_cast<MultiExperiment *>(prototype)) throw Exception(path() + ": experiment must be an experiment/online_learning");
  }
  std::vector<double> run(const RunOptions &opt) override
  {
    RunOptions clones = opt;
    clones.replicas = instances * std::max(1, opt.replicas);
    return prototype->run(clones);
  }
};
GRLX_REGISTER(MultiExperiment)

} // namespace grlx_host
