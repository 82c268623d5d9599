// configurable.h -- host-side mirror of grl's configuration runtime for the hot path.
//
// Same roles and vocabulary as the reference (base/include/grl/configurable.h,
// configuration.h, factory.h; base/src/configurable.cpp): a YAML file becomes a tree of
// Configurators; object nodes are created through a factory keyed by the reference's
// TYPEINFO strings, `request()` declares parameters (name / type / default / optional),
// children are instantiated in YAML key order (configurable.cpp:627-654 -- the order IS
// the order RNG streams are consumed in), parameter values are strings resolved by path
// (relative to the parameter node, then from the root; configurable.cpp:355-432) and
// `configure()` may publish provided parameters (configurable.cpp:691-712).
// Written fresh for this path; the objects are descriptors (no CPU compute): the
// experiment lowers the validated graph to a grlx_config and runs it through the C ABI.
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace grlx_host {

struct Exception : std::runtime_error { using std::runtime_error::runtime_error; };
struct bad_param : Exception { explicit bad_param(const std::string &w) : Exception("Parameter '" + w + "' has an illegal value") {} };

extern int log_verbosity;     // deployer -v
void log(int level, const std::string &msg);

// ---------------------------------------------------------------- YAML subset ----
struct YamlNode {
  bool is_map = false;
  std::string scalar;                                            // raw text of a scalar / flow sequence
  std::vector<std::pair<std::string, YamlNode>> children;        // key order preserved
  const YamlNode *find(const std::string &k) const;
};
YamlNode parse_yaml(const std::string &text);                    // throws Exception with line number
void merge_yaml(YamlNode &into, const YamlNode &from);           // several files on the command line (deployer.cpp:88-93)

std::vector<double> parse_vector(const std::string &s, const std::string &what);   // "[a, b]" or a bare scalar (vector.h:61-109)
std::string format_vector(const std::vector<double> &v);                           // max_digits10, round-trips exactly

class Configurable;

// one requested parameter (configurable.h:97-237)
struct CRP {
  enum Mutability { Configuration, System, Online, Provided };
  std::string name, type, description, def;
  Mutability mutability = Configuration;
  bool optional = false, is_object = false;
  CRP(std::string n, std::string desc, double v, Mutability m = Configuration);
  CRP(std::string n, std::string desc, int v, Mutability m = Configuration);
  CRP(std::string n, std::string desc, const std::string &v, Mutability m = Configuration);
  CRP(std::string n, std::string desc, const std::vector<double> &v, Mutability m = Configuration);
  // object parameter: `type` is the requested base type ("projector.pair" -> base "projector", role "pair")
  CRP(std::string n, std::string type, std::string desc, Configurable *ptr, bool optional = false);
  // provided parameter (published by configure()): type like "vector.action_min", "int.memory"
  static CRP provided(std::string n, std::string type, std::string desc);
};
using ConfigurationRequest = std::vector<CRP>;

// string-typed parameter bag handed to configure() (configuration.h:84-206)
class Configuration {
 public:
  struct Value {
    std::string s;
    Configurable *p = nullptr;
    const std::string &str() const { return s; }
    operator double() const;
    operator int() const;
    std::vector<double> v() const;
    Configurable *ptr() const { return p; }
  };
  bool has(const std::string &k) const { return values_.count(k) != 0; }
  const Value &operator[](const std::string &k) const;
  void set(const std::string &k, const std::string &v) { values_[k].s = v; provided_.push_back(k); }
  void set(const std::string &k, double v);
  void set(const std::string &k, int v);
  void set(const std::string &k, const std::vector<double> &v) { set(k, format_vector(v)); }
  void put(const std::string &k, const std::string &s, Configurable *p) { values_[k].s = s; values_[k].p = p; }
  const std::vector<std::string> &provided() const { return provided_; }
  void clear_provided() { provided_.clear(); }
 private:
  std::map<std::string, Value> values_;
  std::vector<std::string> provided_;
};

class Configurator;

class Configurable {
 public:
  virtual ~Configurable() {}
  virtual std::string d_type() const = 0;                       // TYPEINFO string
  virtual void request(const std::string &role, ConfigurationRequest *config) { (void)role; (void)config; }
  virtual void configure(Configuration &config) { (void)config; }
  Configurator *configurator = nullptr;
  std::string path() const;
};

using Creator = Configurable *(*)();
class ConfigurableFactory {
 public:
  static std::map<std::string, Creator> &factories();
  static Configurable *create(const std::string &type);
  // unique-suffix normalisation of a yaml type (configurable.cpp:106-125): "predictor/sarsa" -> "predictor/critic/sarsa"
  static std::string normalise(const std::string &type);
};
struct Registrar { Registrar(const std::string &type, Creator c) { ConfigurableFactory::factories()[type] = c; } };
#define GRLX_TYPEINFO(t) static std::string s_type() { return t; } std::string d_type() const override { return t; }
#define GRLX_REGISTER(cls) static ::grlx_host::Registrar registrar_##cls(cls::s_type(), []() -> ::grlx_host::Configurable * { return new cls(); });

// instantiated tree (ObjectConfigurator / ParameterConfigurator of the reference folded into one node type)
class Configurator {
 public:
  std::string name;
  Configurator *parent = nullptr;
  std::vector<std::unique_ptr<Configurator>> children;          // YAML order, then defaults, then provided
  bool is_object = false;
  std::string value;                                            // resolved string of a parameter
  std::unique_ptr<Configurable> object;                         // owned object of an object node
  Configurable *ref = nullptr;                                  // object a reference parameter points to
  Configuration config;                                         // what configure() received (object nodes)

  Configurator *child(const std::string &n) const;
  Configurator *root();
  std::string path() const;
  // path lookup: relative (with ..) from this node, then from the root (configurable.cpp:355-376)
  Configurator *find(const std::string &path);
  Configurable *ptr() const { return is_object ? object.get() : ref; }
  std::string yaml(int indent = 0) const;                       // resolved configuration dump (online_learning.cpp:117-122)
};

// loadYAML + instantiate (configurable.cpp:68-193, 603-715).  Returns the root of the instantiated tree.
std::unique_ptr<Configurator> instantiate(const YamlNode &root);
// order in which object nodes were instantiated (type strings with paths) -- the RNG consumption order
const std::vector<Configurator *> &instantiate_order();

} // namespace grlx_host
