/** \file grlx.h
 * \brief Reference-side binding of grl's plug-in API to libgrlx.so (the MI355X runner of the online-learning hot path).
 *
 * Written against grl's public headers (base/include/grl/{experiment,projector,configurable}.h); it is NOT built in the
 * grl_amd repository (grl needs Eigen3, which that repository's build image lacks) -- it is the source a grl
 * maintainer adds as addons/grlx/.  Every class is a thin adapter: parameters are read from the instantiated
 * configuration tree by the reference's own parameter names and handed to the C ABI of include/grlx.h.
 */
#ifndef GRL_GRLX_EXPERIMENT_H_
#define GRL_GRLX_EXPERIMENT_H_

#include <grl/experiment.h>
#include <grl/environment.h>
#include <grl/agent.h>
#include <grl/projector.h>
#include <grlx.h>

namespace grl
{

/// experiment/online_learning on the GPU: N independent-seed replicas of the configured graph in one launch sequence.
class GrlxOnlineLearningExperiment : public Experiment
{
  public:
    TYPEINFO("experiment/online_learning/grlx", "Online learning experiment run on an MI355X through libgrlx (fused rollout kernels)")

  protected:
    Configurable *agent_, *test_agent_, *environment_;
    int runs_, trials_, steps_, test_interval_, test_trials_, replicas_, seed_, table_log2_capacity_;
    std::string output_, load_file_, save_every_;

  public:
    GrlxOnlineLearningExperiment() : agent_(NULL), test_agent_(NULL), environment_(NULL), runs_(1), trials_(0), steps_(0), test_interval_(-1), test_trials_(1),
                                     replicas_(1), seed_(1), table_log2_capacity_(0), save_every_("never") { }

    // From Configurable
    virtual void request(ConfigurationRequest *config);
    virtual void configure(Configuration &config);
    virtual void reconfigure(const Configuration &config);

    // From Experiment
    virtual LargeVector run();

  protected:
    /// Fill a grlx_config from the instantiated subtree; throws bad_param for graphs the fused kernels do not implement.
    void lower(grlx_config *c) const;

  public:
    /// environment/modeled { model, task } -> the environment fields of a grlx_config
    static void lowerEnvironment(const Configurable *model, const Configurable *task, const Configurable *environment, grlx_config *c);
    /// agent/td { policy, predictor } (+ the test agent's policy, may be NULL) -> the agent fields of a grlx_config
    static void lowerAgent(const Configurable *policy, const Configurable *predictor, const Configurable *test_policy, grlx_config *c);
    static void lowerTile(const Configurable *projector, grlx_tile_spec *t);
    static void lowerLinear(const Configurable *representation, grlx_linear_spec *l);
};

/// experiment/batch_learning on the GPU (tests/pendulum-fqi-ann.yaml: predictor/fqi over representation/iterative + parameterized/ann).
class GrlxBatchLearningExperiment : public Experiment
{
  public:
    TYPEINFO("experiment/batch_learning/grlx", "Batch learning experiment (fitted Q-iteration over an ANN) run on an MI355X through libgrlx")

  protected:
    Configurable *model_, *task_, *predictor_, *test_agent_;
    int runs_, batches_, batch_size_, replicas_, seed_;
    std::string output_;

  public:
    GrlxBatchLearningExperiment() : model_(NULL), task_(NULL), predictor_(NULL), test_agent_(NULL), runs_(1), batches_(0), batch_size_(100), replicas_(1), seed_(1) { }

    // From Configurable
    virtual void request(ConfigurationRequest *config);
    virtual void configure(Configuration &config);
    virtual void reconfigure(const Configuration &config) { }

    // From Experiment
    virtual LargeVector run();

  protected:
    void lower(grlx_fqi_config *c) const;
};

/// agent/td whose policy and predictor run on the GPU, one Agent call per kernel launch: for graphs whose ENVIRONMENT stays a grl
/// object on the CPU (environment/gym, a robot, any simulator).  The context has no environment (GRLX_ENV_EXTERNAL).
class GrlxTDAgent : public Agent
{
  public:
    TYPEINFO("agent/td/grlx", "TD agent (tile-coding SARSA / Q / expected SARSA or actor-critic) stepped on an MI355X through libgrlx")

  protected:
    Configurable *policy_, *predictor_;
    int seed_, table_log2_capacity_;
    grlx_ctx *ctx_;

  public:
    GrlxTDAgent() : policy_(NULL), predictor_(NULL), seed_(1), table_log2_capacity_(0), ctx_(NULL) { }
    ~GrlxTDAgent() { if (ctx_) grlx_destroy(ctx_); }

    // From Configurable
    virtual void request(ConfigurationRequest *config);
    virtual void configure(Configuration &config);
    virtual void reconfigure(const Configuration &config) { }

    // From Agent
    virtual void start(const Observation &obs, Action *action);
    virtual void step(double tau, const Observation &obs, double reward, Action *action);
    virtual void end(double tau, const Observation &obs, double reward);

    grlx_ctx *context() { return ctx_; }
};

/// agent/fixed over the tables of an agent/td/grlx: the greedy (noise-free) test agent on the same device context.
class GrlxFixedAgent : public Agent
{
  public:
    TYPEINFO("agent/fixed/grlx", "Fixed-policy test agent over the tables of an agent/td/grlx")

  protected:
    GrlxTDAgent *agent_;

  public:
    GrlxFixedAgent() : agent_(NULL) { }

    // From Configurable
    virtual void request(ConfigurationRequest *config);
    virtual void configure(Configuration &config);
    virtual void reconfigure(const Configuration &config) { }

    // From Agent
    virtual void start(const Observation &obs, Action *action);
    virtual void step(double tau, const Observation &obs, double reward, Action *action);
    virtual void end(double tau, const Observation &obs, double reward);
};

/// environment/modeled integrated on the GPU, one Environment call per kernel launch: for graphs whose AGENT stays a grl object on the CPU.
class GrlxModeledEnvironment : public Environment
{
  public:
    TYPEINFO("environment/modeled/grlx", "Modeled environment (pendulum, cart-pole, acrobot, compass walker) stepped on an MI355X through libgrlx")

  protected:
    Configurable *model_, *task_;
    int seed_;
    grlx_ctx *ctx_;
    int obs_dims_;

  public:
    GrlxModeledEnvironment() : model_(NULL), task_(NULL), seed_(1), ctx_(NULL), obs_dims_(0) { }
    ~GrlxModeledEnvironment() { if (ctx_) grlx_destroy(ctx_); }

    // From Configurable
    virtual void request(ConfigurationRequest *config);
    virtual void configure(Configuration &config);
    virtual void reconfigure(const Configuration &config) { }

    // From Environment
    virtual void start(int test, Observation *obs);
    virtual double step(const Action &action, Observation *obs, double *reward, int *terminal);
};

/// projector/tile_coding evaluated by the GPU library (parity checks inside grl, mixed CPU/GPU graphs).
class GrlxTileCodingProjector : public Projector
{
  public:
    TYPEINFO("projector/tile_coding/grlx", "Hashed tile coding evaluated by libgrlx (bit-identical indices)")

  protected:
    grlx_tile_spec spec_;

  public:
    virtual void request(const std::string &role, ConfigurationRequest *config);
    virtual void configure(Configuration &config);
    virtual void reconfigure(const Configuration &config) { }
    virtual ProjectionLifetime lifetime() const { return plIndefinite; }
    virtual ProjectionPtr project(const Vector &in) const;
    virtual void project(const Vector &base, const std::vector<Vector> &variants, std::vector<ProjectionPtr> *out) const;
};

}

#endif /* GRL_GRLX_EXPERIMENT_H_ */
