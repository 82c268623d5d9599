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
    static void lowerTile(const Configurable *projector, grlx_tile_spec *t);
    static void lowerLinear(const Configurable *representation, grlx_linear_spec *l);
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
