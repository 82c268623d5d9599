/** \file grlx.cpp
 * \brief Reference-side binding of grl's plug-in API to libgrlx.so (see include/grl/experiments/grlx.h).
 *
 * A yaml switches to the GPU by one line: `type: experiment/online_learning` -> `type: experiment/online_learning/grlx`
 * (plus `replicas: 4096`).  The experiment reads its own, already instantiated subtree through the reference's
 * Configurable::operator[] (configurable.h:798-804) -- same parameter names as the yaml -- decides which fused kernel
 * family serves the graph from the objects' d_type() strings, and fills the grlx_config of include/grlx.h.
 * Errors of the library (GRLX_ERR_INVALID = the reference's bad_param conditions) are re-thrown as grl exceptions.
 */
#include <cstring>
#include <fstream>
#include <iomanip>

#include <grl/experiments/grlx.h>

using namespace grl;

REGISTER_CONFIGURABLE(GrlxOnlineLearningExperiment)
REGISTER_CONFIGURABLE(GrlxBatchLearningExperiment)
REGISTER_CONFIGURABLE(GrlxTDAgent)
REGISTER_CONFIGURABLE(GrlxFixedAgent)
REGISTER_CONFIGURABLE(GrlxModeledEnvironment)
REGISTER_CONFIGURABLE(GrlxTileCodingProjector)

namespace {

bool typeIs(const Configurable *obj, const std::string &suffix)
{ // YAML type names may be given by unique suffix (configurable.cpp:106-125); d_type() is always the full name
  const std::string t = obj->d_type();
  return t.size() >= suffix.size() && t.compare(t.size() - suffix.size(), suffix.size(), suffix) == 0;
}

double first(const LargeVector &v, double otherwise)
{
  return v.size() ? v[0] : otherwise;
}

void check(int rc)
{
  if (rc == GRLX_ERR_INVALID) throw bad_param(grlx_last_error());
  if (rc < 0) throw Exception(grlx_last_error());
}

}

// ---------------------------------------------------------------------------------------------------------------

void GrlxOnlineLearningExperiment::request(ConfigurationRequest *config)
{ // mirrors OnlineLearningExperiment::request (online_learning.cpp:39-67); `rate`, signals and the exporter have no counterpart
  // on the fused path
  config->push_back(CRP("runs", "Number of separate learning runs to perform", runs_, CRP::Configuration, 1));
  config->push_back(CRP("trials", "Number of episodes per learning run (0 = bounded by steps)", trials_, CRP::Configuration, 0));
  config->push_back(CRP("steps", "Number of steps per learning run (0 = bounded by trials)", steps_, CRP::Configuration, 0));
  config->push_back(CRP("test_interval", "Number of episodes in between test trials", test_interval_, CRP::Configuration, -1));
  config->push_back(CRP("test_trials", "Number of test trials per interval", test_trials_, CRP::Configuration, 1));
  config->push_back(CRP("output", "Output base filename", output_));
  config->push_back(CRP("replicas", "Independent-seed replicas run side by side on the GPU (experiment/multi analogue)", replicas_, CRP::Configuration, 1));
  config->push_back(CRP("seed", "Seed of replica 0 (replica i uses seed+i, as `grld -s` would for separate processes)", seed_));
  config->push_back(CRP("table_log2_capacity", "Sparse weight table slots per replica = 2^this (0 = default)", table_log2_capacity_, CRP::Configuration, 0, 26));

  config->push_back(CRP("environment", "environment", "Environment in which the agent acts (environment/modeled)", environment_));
  config->push_back(CRP("agent", "agent", "Agent (agent/td)", agent_));
  config->push_back(CRP("test_agent", "agent", "Agent to use in test trials (agent/fixed)", test_agent_, true));

  config->push_back(CRP("load_file", "Load policy filename", load_file_));
  std::vector<std::string> options;
  options.push_back("never");
  options.push_back("run");
  config->push_back(CRP("save_every", "Save policy to 'output-run<run>-*.dat' at the end of every run", save_every_, CRP::Configuration, options));
}

// The only supported build of libgrlx.so is grl_amd/_build.py: it passes the device code through the work-around for a
// register-allocation bug of ROCm 7.2's compiler.  A plain `hipcc -shared` build carries no tag.
static void requireFilteredBuild()
{
  // the structs of include/grlx.h are passed by address: a library of another ABI version would read them with another layout
  if (grlx_abi_version() != GRLX_ABI_VERSION)
    throw Exception("libgrlx.so has another ABI version than the grlx.h this addon was compiled against: rebuild the addon");
  const char *tag = grlx_build_pipeline();
  if (!tag || strncmp(tag, "device-asm+mir-exec-prologue-fix/", 33))
    throw Exception(std::string("libgrlx.so was not built by `python -m grl_amd._build` (pipeline tag '") + (tag ? tag : "") +
                    "'): its kernels may read stale lanes; rebuild it");
}

void GrlxOnlineLearningExperiment::configure(Configuration &config)
{
  requireFilteredBuild();
  agent_ = config["agent"].ptr();
  test_agent_ = config["test_agent"].ptr();
  environment_ = config["environment"].ptr();

  runs_ = config["runs"];
  trials_ = config["trials"];
  test_interval_ = config["test_interval"];
  steps_ = config["steps"];
  test_trials_ = config["test_trials"];
  if (!trials_ && !steps_) throw bad_param("experiment/online_learning/grlx:{trials,steps} (one of them must bound the run)");
  output_ = config["output"].str();
  replicas_ = config["replicas"];
  seed_ = config["seed"];
  table_log2_capacity_ = config["table_log2_capacity"];
  load_file_ = config["load_file"].str();
  save_every_ = config["save_every"].str();

  if (!typeIs(environment_, "environment/modeled"))
    throw bad_param("experiment/online_learning/grlx:environment (must be environment/modeled)");
  if (!typeIs(agent_, "agent/td"))
    throw bad_param("experiment/online_learning/grlx:agent (must be agent/td)");
  if (test_agent_ && !typeIs(test_agent_, "agent/fixed"))
    throw bad_param("experiment/online_learning/grlx:test_agent (must be agent/fixed)");
}

void GrlxOnlineLearningExperiment::reconfigure(const Configuration &config)
{
}

void GrlxOnlineLearningExperiment::lowerTile(const Configurable *projector, grlx_tile_spec *t)
{ // projector/tile_coding: tilings, memory, safe, resolution, wrapping (tile_coding.cpp:34-42)
  if (!typeIs(projector, "projector/tile_coding"))
    throw bad_param(projector->path() + ": the fused path needs projector/tile_coding");
  if ((*projector)["safe"].i() < 0 || (*projector)["safe"].i() > 2)
    throw bad_param(projector->path() + ":safe (0 = off, 1 = claim on write, 2 = claim always; tile_coding.cpp:38)");
  memset(t, 0, sizeof(*t));
  t->safe = (*projector)["safe"];
  t->tilings = (*projector)["tilings"];
  t->memory = (*projector)["memory"];
  const LargeVector res = (*projector)["resolution"].v(), wrap = (*projector)["wrapping"].v();
  if (res.size() > GRLX_MAX_DIMS) throw bad_param(projector->path() + ":resolution");
  t->dims = res.size();
  for (size_t i = 0; i < (size_t)res.size(); ++i)
  {
    t->resolution[i] = res[i];
    t->wrapping[i] = i < (size_t)wrap.size() ? wrap[i] : 0.;       // tile_coding.cpp:62-66: missing entries = no wrapping
  }
}

void GrlxOnlineLearningExperiment::lowerLinear(const Configurable *representation, grlx_linear_spec *l)
{ // representation/parameterized/linear: init_min/max, output_min/max ([] = unbounded, linear.cpp:86-96), limit
  if (!typeIs(representation, "representation/parameterized/linear"))
    throw bad_param(representation->path() + ": the fused path needs representation/parameterized/linear");
  if ((*representation)["outputs"].i() != 1) throw bad_param(representation->path() + ":outputs (must be 1)");
  l->init_min = first((*representation)["init_min"].v(), 0.);
  l->init_max = first((*representation)["init_max"].v(), 1.);
  l->output_min = first((*representation)["output_min"].v(), -std::numeric_limits<double>::max());
  l->output_max = first((*representation)["output_max"].v(), std::numeric_limits<double>::max());
  l->limit = (*representation)["limit"];
  l->reserved = 0;
}

void GrlxOnlineLearningExperiment::lower(grlx_config *c) const
{
  grlx_config_pendulum_sarsa(c);                                    // defaults of every field, then the tree's values
  c->n_replicas = replicas_;
  c->test_interval = test_interval_;
  c->test_trials = test_trials_;
  c->table_log2_capacity = table_log2_capacity_;
  {
    // rows a run can write: with a steps budget alone a trial has at least one learning step
    const int trial_cap = trials_ ? trials_ : steps_ * (test_interval_ >= 0 ? 2 : 1) + 1;
    c->max_rows = test_interval_ >= 0 ? trial_cap / (test_interval_ + 1) + 1 : trial_cap + 1;
  }

  lowerEnvironment((*environment_)["model"].ptr(), (*environment_)["task"].ptr(), environment_, c);
  lowerAgent((*agent_)["policy"].ptr(), (*agent_)["predictor"].ptr(), test_agent_ ? (*test_agent_)["policy"].ptr() : NULL, c);
  if (c->agent == GRLX_AGENT_AC && !table_log2_capacity_) c->table_log2_capacity = 18;
}

void GrlxOnlineLearningExperiment::lowerEnvironment(const Configurable *model, const Configurable *task, const Configurable *environment, grlx_config *c)
{
  // ---- environment/modeled { model, task }
  c->discrete_time = environment ? (*environment)["discrete_time"].i() : 1;
  c->control_step = (*model)["control_step"];
  c->integration_steps = (*model)["integration_steps"];
  if (typeIs(model, "model/dynamical"))
  {
    const Configurable *dynamics = (*model)["dynamics"].ptr();
    if (typeIs(dynamics, "dynamics/pendulum") && typeIs(task, "task/pendulum/swingup")) c->env = GRLX_ENV_PENDULUM;
    else if (typeIs(dynamics, "dynamics/cart_pole") && typeIs(task, "task/cart_pole/swingup"))
    {
      if ((*dynamics)["end_stop"].i() != 1) throw bad_param(dynamics->path() + ":end_stop (must be 1)");
      c->env = GRLX_ENV_CART_POLE;
      c->end_stop_penalty = (*task)["end_stop_penalty"];
      c->action_penalty = (*task)["action_penalty"];
      if ((*task)["shaping"].i() != 0) throw bad_param(task->path() + ":shaping");
    }
    else if (typeIs(dynamics, "dynamics/acrobot") && typeIs(task, "task/acrobot/balancing")) c->env = GRLX_ENV_ACROBOT;
    else throw bad_param(model->path() + ": dynamics / task pair not implemented by the fused kernels");
  }
  else if (typeIs(model, "model/compass_walker") && typeIs(task, "task/compass_walker/walk"))
  {
    c->env = GRLX_ENV_COMPASS_WALKER;
    c->slope_angle = (*task)["slope_angle"];
    if ((*model)["slope_angle"].d() != c->slope_angle) throw bad_param(task->path() + ":slope_angle (must match the model's)");
    c->initial_state_variation = (*task)["initial_state_variation"];
    c->negative_reward = (*task)["negative_reward"];
  }
  else
    throw bad_param(model->path() + ": model not implemented by the fused kernels");
  if (c->env != GRLX_ENV_ACROBOT) c->timeout = (*task)["timeout"];
  if (c->env == GRLX_ENV_PENDULUM || c->env == GRLX_ENV_CART_POLE) c->randomization = (*task)["randomization"];
}

void GrlxOnlineLearningExperiment::lowerAgent(const Configurable *policy, const Configurable *predictor, const Configurable *test_policy, grlx_config *c)
{
  // ---- agent/td { policy, predictor }

  if (typeIs(predictor, "predictor/ac/action"))
  { // actor-critic (cfg/cart_pole/ac_tc.yaml): mapping/policy/action + predictor/ac/action { critic: predictor/critic/td }
    if (!typeIs(policy, "policy/action") || (test_policy && !typeIs(test_policy, "policy/action")))
      throw bad_param(predictor->path() + ": predictor/ac/action needs mapping/policy/action policies");
    const Configurable *critic = (*predictor)["critic"].ptr();
    if (!typeIs(critic, "predictor/critic/td")) throw bad_param(critic->path() + ": the critic must be predictor/critic/td");
    if ((*predictor)["projector"].ptr() != (*policy)["projector"].ptr() || (*predictor)["representation"].ptr() != (*policy)["representation"].ptr())
      throw bad_param(predictor->path() + ": actor predictor and policy must share projector and representation");
    if (test_policy && first((*test_policy)["sigma"].v(), 0.) != 0) throw bad_param(test_policy->path() + ":sigma (the test policy must be noise-free)");
    c->agent = GRLX_AGENT_AC;
    c->action_min = first((*policy)["output_min"].v(), 0.);
    c->action_max = first((*policy)["output_max"].v(), 0.);
    c->action_steps = 0;
    lowerTile((*critic)["projector"].ptr(), &c->projector);                 // table 0: the critic's V(s)
    lowerLinear((*critic)["representation"].ptr(), &c->representation);
    lowerTile((*policy)["projector"].ptr(), &c->actor_projector);           // table 1: the actor's u(s)
    lowerLinear((*policy)["representation"].ptr(), &c->actor_representation);
    c->alpha = (*critic)["alpha"];
    c->gamma = (*critic)["gamma"];
    c->lambda = (*critic)["lambda"];
    const Configurable *trace = (*critic)["trace"].ptr();
    c->trace = !trace ? GRLX_TRACE_NONE : typeIs(trace, "trace/enumerated/replacing") ? GRLX_TRACE_REPLACING : -1;
    c->actor_alpha = (*predictor)["alpha"];
    c->sigma = first((*policy)["sigma"].v(), 0.);
    c->theta = first((*policy)["theta"].v(), 1.);
    c->ac_decay_rate = (*policy)["decay_rate"];
    c->ac_decay_min = (*policy)["decay_min"];
    c->ac_update_method = (*predictor)["update_method"].str() == "proportional" ? 0 : 1;
    c->ac_step_limit = first((*predictor)["step_limit"].v(), -1.);
    return;
  }

  // discrete-action TD agents (cfg/pendulum/{sarsa,q,qv,advantage}_tc.yaml, cfg/compass_walker/qlearning_walk.yaml):
  // mapping/policy/discrete/value/q { discretizer/uniform, projector/tile_coding, representation/parameterized/linear,
  // sampler/epsilon_greedy } + predictor/critic/{sarsa, q, expected_sarsa, advantage, qv}
  if (!typeIs(policy, "policy/discrete/value/q") && !typeIs(policy, "policy/discrete/q"))
    throw bad_param(policy->path() + ": the fused path needs mapping/policy/discrete/value/q or mapping/policy/action");
  if (typeIs(predictor, "predictor/critic/sarsa")) c->agent = GRLX_AGENT_SARSA;
  else if (typeIs(predictor, "predictor/critic/expected_sarsa")) c->agent = GRLX_AGENT_EXPECTED_SARSA;
  else if (typeIs(predictor, "predictor/critic/q")) c->agent = GRLX_AGENT_Q;
  else if (typeIs(predictor, "predictor/critic/advantage")) { c->agent = GRLX_AGENT_ADVANTAGE; c->kappa = (*predictor)["kappa"]; }
  else if (typeIs(predictor, "predictor/critic/qv")) c->agent = GRLX_AGENT_QV;
  else throw bad_param(predictor->path() + ": predictor not implemented by the fused kernels");

  const Configurable *projector = (*policy)["projector"].ptr(), *representation = (*policy)["representation"].ptr();
  const char *pkey = c->agent == GRLX_AGENT_QV ? "q_projector" : "projector", *rkey = c->agent == GRLX_AGENT_QV ? "q_representation" : "representation";
  if ((*predictor)[pkey].ptr() != projector || (*predictor)[rkey].ptr() != representation)
    throw bad_param(predictor->path() + ": predictor and policy must share projector and representation");
  if (test_policy && ((*test_policy)["projector"].ptr() != projector || (*test_policy)["representation"].ptr() != representation))
    throw bad_param(test_policy->path() + ": the test policy must share projector and representation with the learning policy");

  const Configurable *discretizer = (*policy)["discretizer"].ptr(), *sampler = (*policy)["sampler"].ptr();
  if (!typeIs(discretizer, "discretizer/uniform")) throw bad_param(discretizer->path() + ": discretizer/uniform expected");
  if (!typeIs(sampler, "sampler/epsilon_greedy")) throw bad_param(sampler->path() + ": sampler/epsilon_greedy expected for the learning policy");
  if (test_policy && !typeIs((*test_policy)["sampler"].ptr(), "sampler/greedy"))
    throw bad_param(test_policy->path() + ":sampler (sampler/greedy expected for the test policy)");
  const LargeVector dmin = (*discretizer)["min"].v(), dmax = (*discretizer)["max"].v(), dsteps = (*discretizer)["steps"].v();
  if (dmin.size() != 1 || dmax.size() != 1 || dsteps.size() != 1) throw bad_param(discretizer->path() + ": one action dimension supported");
  c->action_min = dmin[0];
  c->action_max = dmax[0];
  c->action_steps = (int)dsteps[0];
  lowerTile(projector, &c->projector);
  lowerLinear(representation, &c->representation);
  c->target_interval = (*representation)["interval"];               // target network of the Q table (representation.h:173-190)
  c->target_tau = (*representation)["tau"];
  c->epsilon = first((*sampler)["epsilon"].v(), 0.);
  c->decay_rate = (*sampler)["decay_rate"];
  c->decay_min = (*sampler)["decay_min"];
  c->alpha = (*predictor)["alpha"];
  c->gamma = (*predictor)["gamma"];
  c->lambda = (*predictor)["lambda"];
  const Configurable *trace = (*predictor)["trace"].ptr();
  c->trace = !trace ? GRLX_TRACE_NONE : typeIs(trace, "trace/enumerated/replacing") ? GRLX_TRACE_REPLACING
           : typeIs(trace, "trace/enumerated/accumulating") ? GRLX_TRACE_ACCUMULATING : -1;
  if (c->agent == GRLX_AGENT_QV)
  { // second table: V(s) (v_projector / v_representation), learning rate beta (qv.cpp:35-64)
    lowerTile((*predictor)["v_projector"].ptr(), &c->actor_projector);
    lowerLinear((*predictor)["v_representation"].ptr(), &c->actor_representation);
    c->beta = (*predictor)["beta"];
  }
}

LargeVector GrlxOnlineLearningExperiment::run()
{
  grlx_config c;
  lower(&c);

  std::vector<double> curve;
  // ONE instantiation for all runs, as in the reference: between two runs the experiment is reset, not re-created
  // (online_learning.cpp:307-308 -> grlx_reset_run: parameters re-drawn from the continuing thread-local stream, traces cleared,
  // decay back to 1, counters restart; no stream is reseeded).  Replica i is seeded seed + i.
  std::vector<int64_t> seeds(replicas_);
  for (int i = 0; i < replicas_; ++i) seeds[i] = (int64_t)seed_ + i;
  grlx_ctx *ctx = NULL;
  check(grlx_create(&c, &seeds[0], &ctx));
  for (int rr = 0; rr < runs_; ++rr)
  {
    try
    {
      // load_file: the .dat parameter files a (CPU or GPU) grl saved; names as ParameterizedRepresentation builds them
      // (representation.h:201-229: <file><config path with '/' -> '_'>.dat), `$run` replaced by the run number
      // ... one grlx_load_weights(ctx, table, 0, replicas_, data, count) per representation of the agent (table 0 =
      // Q / critic, table 1 = actor / V) exactly as grl_amd/csrc/host/objects.cpp does; omitted when load_file is empty.

      // the trial loop with both of its bounds (online_learning.cpp:154); with a steps budget the clones stop at trials of their own
      if (steps_) check(grlx_run_steps(ctx, trials_ ? trials_ : (1 << 30), (uint64_t)steps_, NULL));
      else check(grlx_run(ctx, trials_, NULL));
      check(grlx_sync(ctx, NULL));

      for (int i = 0; i < replicas_; ++i)
      { // `<output>-<run>@<i>.txt` as experiment/multi names its clones' files (multi.cpp:52-56); single replica: `<output>-<run>.txt`
        const int n = grlx_replica_rows(ctx, i);
        std::vector<int64_t> trial(n), steps(n);
        std::vector<double> reward(n), time(n);
        check(grlx_read_rows(ctx, i, 0, n, &trial[0], &steps[0], &reward[0]));
        check(grlx_read_row_times(ctx, i, 0, n, &time[0]));
        if (!output_.empty())
        {
          std::ostringstream name;
          name << output_ << "-" << rr;
          if (replicas_ > 1) name << "@" << i;
          name << ".txt";
          std::ofstream ofs(name.str().c_str());
          for (int k = 0; k < n; ++k)
            ofs << std::setw(15) << trial[k] << std::setw(15) << steps[k] << std::setw(15) << std::setprecision(3) << std::fixed << reward[k]
                << std::setw(15) << std::setprecision(3) << time[k] << std::setw(15) << reward[k] / time[k] << std::endl;   // online_learning.cpp:243-247
        }
        if (i == replicas_ - 1 && rr == runs_ - 1)
          curve.assign(reward.begin(), reward.end());                // run() returns the learning curve of the last run
      }
      if (save_every_ == "run" && !output_.empty())
      { // raw little-endian double[memory] per table, the reference's .dat layout: a CPU grl can load them
        const int tables = (c.agent == GRLX_AGENT_AC || c.agent == GRLX_AGENT_QV) ? 2 : 1;
        for (int t = 0; t < tables; ++t)
        {
          const int memory = t == 1 ? c.actor_projector.memory : c.projector.memory;
          std::vector<double> dense(memory);
          check(grlx_export_weights(ctx, t, 0, &dense[0]));
          std::ostringstream name;
          name << output_ << "-run" << rr << "-table" << t << ".dat";
          std::ofstream ofs(name.str().c_str(), std::ios::binary);
          ofs.write((const char *)&dense[0], sizeof(double) * dense.size());
        }
      }
    }
    catch (...)
    {
      grlx_destroy(ctx);
      throw;
    }
    if (rr < runs_ - 1 && grlx_reset_run(ctx) != GRLX_OK)
    {
      const std::string e = grlx_last_error();
      grlx_destroy(ctx);
      throw Exception(e);
    }
  }
  grlx_destroy(ctx);

  LargeVector result;
  toVector(curve, result);
  return result;
}

// ---------------------------------------------------------------------------------------------------------------
// experiment/batch_learning/grlx: the reference's BatchLearningExperiment (batch_learning.cpp:36-205) with predictor/fqi over
// representation/iterative + representation/parameterized/ann and projector/pre/normalizing, lowered to a grlx_fqi_config.

void GrlxBatchLearningExperiment::request(ConfigurationRequest *config)
{ // mirrors BatchLearningExperiment::request (batch_learning.cpp:38-58); the sampling box is the task's (observation / action limits)
  config->push_back(CRP("runs", "Number of separate learning runs to perform", runs_, CRP::Configuration, 1, 1));
  config->push_back(CRP("batches", "Number of batches per learning run", batches_, CRP::Configuration, 1));
  config->push_back(CRP("batch_size", "Number of transitions per batch", batch_size_, CRP::Configuration, 1));
  config->push_back(CRP("output", "Output base filename", output_));
  config->push_back(CRP("replicas", "Independent-seed replicas run side by side on the GPU", replicas_, CRP::Configuration, 1));
  config->push_back(CRP("seed", "Seed of replica 0 (replica i uses seed+i)", seed_));
  config->push_back(CRP("model", "model", "Model in which the task is set (model/dynamical over dynamics/pendulum)", model_));
  config->push_back(CRP("task", "task", "Task to be solved (task/pendulum/swingup: it must support invert())", task_));
  config->push_back(CRP("predictor", "predictor", "Learner (predictor/fqi)", predictor_));
  config->push_back(CRP("test_agent", "agent", "Agent to use in test trials after each batch (agent/fixed over the predictor's objects)", test_agent_));
}

void GrlxBatchLearningExperiment::configure(Configuration &config)
{
  requireFilteredBuild();
  model_ = config["model"].ptr();
  task_ = config["task"].ptr();
  predictor_ = config["predictor"].ptr();
  test_agent_ = config["test_agent"].ptr();
  runs_ = config["runs"];
  batches_ = config["batches"];
  batch_size_ = config["batch_size"];
  output_ = config["output"].str();
  replicas_ = config["replicas"];
  seed_ = config["seed"];
  if (!typeIs(predictor_, "predictor/fqi")) throw bad_param("experiment/batch_learning/grlx:predictor (must be predictor/fqi)");
  if (!typeIs(test_agent_, "agent/fixed")) throw bad_param("experiment/batch_learning/grlx:test_agent (must be agent/fixed)");
}

void GrlxBatchLearningExperiment::lower(grlx_fqi_config *c) const
{
  grlx_fqi_config_pendulum(c);
  c->n_replicas = replicas_;
  if (!typeIs(model_, "model/dynamical") || !typeIs((*model_)["dynamics"].ptr(), "dynamics/pendulum") || !typeIs(task_, "task/pendulum/swingup"))
    throw bad_param(model_->path() + ": the batch path is built for model/dynamical with dynamics/pendulum and task/pendulum/swingup");
  if ((*task_)["randomization"].d() != 0) throw bad_param(task_->path() + ":randomization (must be 0)");
  c->control_step = (*model_)["control_step"];
  c->integration_steps = (*model_)["integration_steps"];
  c->timeout = (*task_)["timeout"];
  const Configurable *discretizer = (*predictor_)["discretizer"].ptr(), *projector = (*predictor_)["projector"].ptr(),
                     *representation = (*predictor_)["representation"].ptr();
  if (!typeIs(discretizer, "discretizer/uniform")) throw bad_param(discretizer->path() + ": discretizer/uniform expected");
  const LargeVector dmin = (*discretizer)["min"].v(), dmax = (*discretizer)["max"].v(), dsteps = (*discretizer)["steps"].v();
  if (dmin.size() != 1 || dmax.size() != 1 || dsteps.size() != 1) throw bad_param(discretizer->path() + ": one action dimension supported");
  c->action_min = dmin[0];
  c->action_max = dmax[0];
  c->action_steps = (int)dsteps[0];
  if (!typeIs(projector, "projector/pre/normalizing") || !typeIs((*projector)["projector"].ptr(), "projector/identity") || (*projector)["signed"].i() != 0)
    throw bad_param(projector->path() + ": projector/pre/normalizing (signed = 0) over projector/identity expected");
  // (the limits are the task's observation ++ action limits: a yaml that adds them element-wise, as tests/pendulum-fqi-ann.yaml does under
  //  today's parser, would be refused by NormalizingProjector::project at the first sample anyway; the kernels scale by the task's limits)
  if (!typeIs(representation, "representation/iterative")) throw bad_param(representation->path() + ": representation/iterative expected");
  const Configurable *ann = (*representation)["representation"].ptr();
  if (!typeIs(ann, "representation/parameterized/ann")) throw bad_param(ann->path() + ": representation/parameterized/ann expected");
  if ((*representation)["cumulative"].i() != 0 || (*representation)["batch_size"].i() != 0)
    throw bad_param(representation->path() + ": cumulative = 0 and batch_size = 0 (the whole data set per epoch)");
  const LargeVector hiddens = (*ann)["hiddens"].v();
  if (hiddens.size() != 1 || (*ann)["outputs"].i() != 1 || (*ann)["inputs"].i() != 3) throw bad_param(ann->path() + ": a 3-H-1 network (one hidden layer)");
  c->hidden = (int)round(hiddens[0]);
  c->eta = (*ann)["eta"];
  c->epochs = (*representation)["epochs"];
  c->gamma = (*predictor_)["gamma"];
  c->iterations = (*predictor_)["iterations"];
  if ((*predictor_)["reset_strategy"].str() != "never" || (*predictor_)["macro_batch_size"].i() != 1)
    throw bad_param(predictor_->path() + ": reset_strategy never and macro_batch_size 1");
  c->batch_size = batch_size_;
  c->max_batches = batches_;
  if ((long long)batches_ * batch_size_ > (long long)(*predictor_)["transitions"].i())
    throw bad_param(predictor_->path() + ":transitions (smaller than batches x batch_size)");
  const Configurable *tp = (*test_agent_)["policy"].ptr();
  if ((*tp)["discretizer"].ptr() != discretizer || (*tp)["projector"].ptr() != projector || (*tp)["representation"].ptr() != representation ||
      !typeIs((*tp)["sampler"].ptr(), "sampler/greedy"))
    throw bad_param(test_agent_->path() + ": the test policy must be the greedy Q policy over the predictor's discretizer, projector and representation");
}

LargeVector GrlxBatchLearningExperiment::run()
{
  grlx_fqi_config c;
  lower(&c);
  std::vector<int64_t> seeds(replicas_);
  for (int i = 0; i < replicas_; ++i) seeds[i] = (int64_t)seed_ + i;
  grlx_fqi_ctx *ctx = NULL;
  check(grlx_fqi_create(&c, &seeds[0], &ctx));
  std::vector<double> curve;
  try
  {
    std::vector<std::ofstream> files(replicas_);
    if (!output_.empty())
      for (int i = 0; i < replicas_; ++i)
      {
        std::ostringstream name;
        name << output_ << "-0";
        if (replicas_ > 1) name << "@" << i;
        name << ".txt";
        files[i].open(name.str().c_str());
      }
    for (int bb = 0; bb < batches_; ++bb)
    { // batch_learning.cpp:105-188: one batch of samples, FQIPredictor::rebuild, one greedy test trial -> one row
      check(grlx_fqi_run_batch(ctx, NULL));
      check(grlx_fqi_sync(ctx, NULL));
      for (int i = 0; i < replicas_; ++i)
      {
        int64_t batch = 0, transitions = 0;
        double reward = 0;
        check(grlx_fqi_read_rows(ctx, i, bb, 1, &batch, &transitions, &reward));
        std::ostringstream oss;
        oss << std::setw(15) << batch << std::setw(15) << transitions << std::setw(15) << reward;        // :179
        if (files[i].is_open()) files[i] << oss.str() << std::endl;
        if (i == 0) { INFO(oss.str()); curve.push_back(reward); }
      }
    }
  }
  catch (...)
  {
    grlx_fqi_destroy(ctx);
    throw;
  }
  grlx_fqi_destroy(ctx);
  LargeVector result;
  toVector(curve, result);
  return result;
}

// ---------------------------------------------------------------------------------------------------------------
// The per-step side: a grl graph keeps its own loop (experiment/online_learning, or anything that drives Agent / Environment) and puts
// ONE of the two objects on the GPU.  One replica per object; every call is one kernel launch plus the copies of its arguments.

void GrlxTDAgent::request(ConfigurationRequest *config)
{ // agent/td (td.cpp:36-41) + what the device context needs
  config->push_back(CRP("policy", "mapping/policy", "Control policy (mapping/policy/discrete/value/q or mapping/policy/action over tile coding)", policy_));
  config->push_back(CRP("predictor", "predictor", "Value function predictor (predictor/critic/{sarsa,q,expected_sarsa} or predictor/ac/action)", predictor_));
  config->push_back(CRP("seed", "Seed of the agent's random streams (what `grld -s` seeds)", seed_));
  config->push_back(CRP("table_log2_capacity", "Sparse weight table slots = 2^this (0 = default)", table_log2_capacity_, CRP::Configuration, 0, 26));
}

void GrlxTDAgent::configure(Configuration &config)
{
  requireFilteredBuild();
  policy_ = config["policy"].ptr();
  predictor_ = config["predictor"].ptr();
  seed_ = config["seed"];
  table_log2_capacity_ = config["table_log2_capacity"];
  grlx_config c;
  grlx_config_pendulum_sarsa(&c);
  c.env = GRLX_ENV_EXTERNAL;                         // the environment is another object of the graph
  c.n_replicas = 1;
  c.table_log2_capacity = table_log2_capacity_;
  GrlxOnlineLearningExperiment::lowerAgent(policy_, predictor_, NULL, &c);
  const int64_t seed = seed_;
  check(grlx_create(&c, &seed, &ctx_));
}

static void agentCall(grlx_ctx *ctx, int mode, int test, double tau, const Observation &obs, double reward, Action *action)
{
  std::vector<double> o(obs.v.size());
  for (size_t i = 0; i < (size_t)obs.v.size(); ++i) o[i] = obs.v[i];
  double a = (action && action->v.size()) ? action->v[0] : 0.;
  if (mode == 0) check(grlx_agent_start(ctx, test, NULL, &o[0], &a));
  else if (mode == 1) check(grlx_agent_step(ctx, test, NULL, tau, &o[0], &reward, NULL, &a));
  else check(grlx_agent_end(ctx, test, NULL, tau, &o[0], &reward));
  if (action)
  {
    action->v = VectorConstructor(a);
    action->type = atUndefined;                      // (exploration is decided on the device; the type flag is informational)
  }
}

void GrlxTDAgent::start(const Observation &obs, Action *action) { agentCall(ctx_, 0, 0, 0., obs, 0., action); }
void GrlxTDAgent::step(double tau, const Observation &obs, double reward, Action *action) { agentCall(ctx_, 1, 0, tau, obs, reward, action); }
void GrlxTDAgent::end(double tau, const Observation &obs, double reward) { agentCall(ctx_, 2, 0, tau, obs, reward, NULL); }

void GrlxFixedAgent::request(ConfigurationRequest *config)
{
  config->push_back(CRP("agent", "agent/td/grlx", "The learning agent whose tables the test policy reads", agent_));
}

void GrlxFixedAgent::configure(Configuration &config)
{
  agent_ = dynamic_cast<GrlxTDAgent *>((Configurable *)config["agent"].ptr());
  if (!agent_) throw bad_param("agent/fixed/grlx:agent (must be an agent/td/grlx)");
}

void GrlxFixedAgent::start(const Observation &obs, Action *action) { agentCall(agent_->context(), 0, 1, 0., obs, 0., action); }
void GrlxFixedAgent::step(double tau, const Observation &obs, double reward, Action *action) { agentCall(agent_->context(), 1, 1, tau, obs, reward, action); }
void GrlxFixedAgent::end(double tau, const Observation &obs, double reward) { agentCall(agent_->context(), 2, 1, tau, obs, reward, NULL); }

void GrlxModeledEnvironment::request(ConfigurationRequest *config)
{ // environment/modeled (modeled.cpp:37-65) without window / delta / exporter
  config->push_back(CRP("model", "model", "Environment model", model_));
  config->push_back(CRP("task", "task", "Task to perform in the environment (should match model)", task_));
  config->push_back(CRP("seed", "Seed of the environment's random streams", seed_));
}

void GrlxModeledEnvironment::configure(Configuration &config)
{
  requireFilteredBuild();
  model_ = config["model"].ptr();
  task_ = config["task"].ptr();
  seed_ = config["seed"];
  grlx_config c;
  grlx_config_pendulum_sarsa(&c);                    // a context needs an agent block: the default one, never stepped here
  GrlxOnlineLearningExperiment::lowerEnvironment(model_, task_, NULL, &c);
  int sd = 0;
  check(grlx_env_dims(c.env, &sd, &obs_dims_));
  c.projector.dims = obs_dims_ + 1;
  for (int i = 0; i < GRLX_MAX_DIMS; ++i) { c.projector.resolution[i] = 1.; c.projector.wrapping[i] = 0.; }
  c.n_replicas = 1;
  const int64_t seed = seed_;
  check(grlx_create(&c, &seed, &ctx_));
}

void GrlxModeledEnvironment::start(int test, Observation *obs)
{
  std::vector<double> o(obs_dims_);
  check(grlx_env_start(ctx_, test, NULL, &o[0]));
  toVector(o, obs->v);
  obs->absorbing = false;
}

double GrlxModeledEnvironment::step(const Action &action, Observation *obs, double *reward, int *terminal)
{
  std::vector<double> o(obs_dims_);
  const double a = action.v[0];
  int32_t term = 0;
  check(grlx_env_advance(ctx_, NULL, &a, &o[0], reward, &term));
  toVector(o, obs->v);
  obs->absorbing = term == 2;
  *terminal = term;
  return 1.;                                         // discrete_time (modeled.cpp:209-212)
}

// ---------------------------------------------------------------------------------------------------------------

void GrlxTileCodingProjector::request(const std::string &role, ConfigurationRequest *config)
{ // the parameters of projector/tile_coding (tile_coding.cpp:34-42); safe must stay 0
  config->push_back(CRP("tilings", "Number of tilings", (int)spec_.tilings));
  config->push_back(CRP("memory", "int.memory", "Hash table size", (int)spec_.memory));
  config->push_back(CRP("resolution", "Size of a single tile", Vector()));
  config->push_back(CRP("wrapping", "vector.wrapping", "Wrapping boundaries (must be multiple of resolution)", Vector()));
}

void GrlxTileCodingProjector::configure(Configuration &config)
{
  requireFilteredBuild();
  memset(&spec_, 0, sizeof(spec_));
  spec_.tilings = config["tilings"];
  spec_.memory = config["memory"];
  const LargeVector res = config["resolution"].v(), wrap = config["wrapping"].v();
  if (!res.size() || res.size() > GRLX_MAX_DIMS) throw bad_param("projector/tile_coding/grlx:resolution");
  spec_.dims = res.size();
  for (size_t i = 0; i < (size_t)res.size(); ++i)
  {
    spec_.resolution[i] = res[i];
    spec_.wrapping[i] = i < (size_t)wrap.size() ? wrap[i] : 0.;
  }
  uint32_t probe[32];
  std::vector<double> zero(spec_.dims, 0.);
  check(grlx_project(&spec_, &zero[0], 1, probe));                   // validates the spec (non-integer wrapping etc.) once
}

ProjectionPtr GrlxTileCodingProjector::project(const Vector &in) const
{
  if ((int)in.size() != spec_.dims) throw bad_param("projector/tile_coding/grlx:resolution (input size mismatch)");
  IndexProjection *p = new IndexProjection();
  std::vector<double> x(in.size());
  std::vector<uint32_t> out(spec_.tilings);
  for (size_t i = 0; i < (size_t)in.size(); ++i) x[i] = in[i];
  check(grlx_project(&spec_, &x[0], 1, &out[0]));
  p->indices.assign(out.begin(), out.end());
  return ProjectionPtr(p);
}

void GrlxTileCodingProjector::project(const Vector &base, const std::vector<Vector> &variants, std::vector<ProjectionPtr> *out) const
{ // all variants (e.g. the discrete actions of one state) in ONE call of the batched operator
  const size_t n = variants.size(), nb = base.size();
  std::vector<double> x(n * spec_.dims);
  for (size_t k = 0; k < n; ++k)
  {
    if ((int)(nb + variants[k].size()) != spec_.dims) throw bad_param("projector/tile_coding/grlx:resolution (input size mismatch)");
    for (size_t i = 0; i < nb; ++i) x[k * spec_.dims + i] = base[i];
    for (size_t i = 0; i < (size_t)variants[k].size(); ++i) x[k * spec_.dims + nb + i] = variants[k][i];
  }
  std::vector<uint32_t> idx(n * spec_.tilings);
  check(grlx_project(&spec_, n ? &x[0] : NULL, (int)n, n ? &idx[0] : NULL));
  out->clear();
  for (size_t k = 0; k < n; ++k)
  {
    IndexProjection *p = new IndexProjection();
    p->indices.assign(idx.begin() + k * spec_.tilings, idx.begin() + (k + 1) * spec_.tilings);
    out->push_back(ProjectionPtr(p));
  }
}
