// configurable.cpp -- see configurable.h.  Follows base/src/configurable.cpp (loadYAML :68-193,
// path resolution :355-432, instantiate :603-715) for the YAML subset the BASELINE configs use.
#include "configurable.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <limits>
#include <sstream>

namespace grlx_host {

int log_verbosity = 3;
void log(int level, const std::string &msg)
{
  static const char *names[] = {"ERR", "WRN", "NTC", "INF", "TRC", "DBG", "CRL"};
  if (level <= log_verbosity) std::cerr << "[" << names[std::min(level, 6)] << "] " << msg << std::endl;
}

// ------------------------------------------------------------------ YAML ----
const YamlNode *YamlNode::find(const std::string &k) const
{
  for (auto &c : children)
    if (c.first == k) return &c.second;
  return nullptr;
}

namespace {
std::string trim(const std::string &s)
{
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}

struct Line { int indent; std::string key, value; bool has_value; int no; };

std::string strip_comment(const std::string &s)
{
  bool inq = false;
  for (size_t i = 0; i < s.size(); ++i)
  {
    if (s[i] == '"') inq = !inq;
    if (s[i] == '#' && !inq && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) return s.substr(0, i);
  }
  return s;
}

void build(const std::vector<Line> &lines, size_t &i, int indent, YamlNode &node)
{
  node.is_map = true;
  while (i < lines.size() && lines[i].indent == indent)
  {
    const Line &l = lines[i++];
    YamlNode child;
    if (l.has_value)
      child.scalar = l.value;
    else if (i < lines.size() && lines[i].indent > indent)
      build(lines, i, lines[i].indent, child);
    else
      child.scalar = "";
    for (auto &c : node.children)
      if (c.first == l.key) throw Exception("yaml line " + std::to_string(l.no) + ": duplicate key '" + l.key + "'");
    node.children.emplace_back(l.key, std::move(child));
  }
  if (i < lines.size() && lines[i].indent > indent)
    throw Exception("yaml line " + std::to_string(lines[i].no) + ": unexpected indentation");
}
} // namespace

YamlNode parse_yaml(const std::string &text)
{
  std::vector<Line> lines;
  std::istringstream iss(text);
  std::string raw;
  int no = 0;
  while (std::getline(iss, raw))
  {
    ++no;
    std::string s = strip_comment(raw);
    if (trim(s).empty() || trim(s) == "---") continue;
    if (s.find('\t') != std::string::npos && s.find_first_not_of(" \t") > s.find('\t'))
      throw Exception("yaml line " + std::to_string(no) + ": tab indentation");
    int indent = (int)s.find_first_not_of(' ');
    std::string body = trim(s);
    if (body[0] == '-') throw Exception("yaml line " + std::to_string(no) + ": block sequences are not supported by this loader");
    size_t colon = body.find(':');
    if (colon == std::string::npos) throw Exception("yaml line " + std::to_string(no) + ": expected 'key: value'");
    Line l;
    l.indent = indent;
    l.key = trim(body.substr(0, colon));
    std::string v = trim(body.substr(colon + 1));
    l.has_value = !v.empty();
    if (v.size() >= 2 && v.front() == '"' && v.back() == '"') { v = v.substr(1, v.size() - 2); l.has_value = true; }
    if (v == "|") throw Exception("yaml line " + std::to_string(no) + ": block scalars are not supported by this loader");
    l.value = v;
    l.no = no;
    lines.push_back(l);
  }
  YamlNode root;
  root.is_map = true;
  size_t i = 0;
  if (!lines.empty()) build(lines, i, lines[0].indent, root);
  if (i != lines.size()) throw Exception("yaml line " + std::to_string(lines[i].no) + ": inconsistent indentation");
  return root;
}

void merge_yaml(YamlNode &into, const YamlNode &from)
{
  for (auto &c : from.children)
  {
    bool found = false;
    for (auto &d : into.children)
      if (d.first == c.first)
      {
        found = true;
        if (d.second.is_map && c.second.is_map) merge_yaml(d.second, c.second);
        else d.second = c.second;
      }
    if (!found) into.children.push_back(c);
  }
  into.is_map = true;
}

std::vector<double> parse_vector(const std::string &str, const std::string &what)
{
  std::string s = trim(str);
  std::vector<double> out;
  if (s.empty()) return out;
  if (s.front() == '[')
  {
    if (s.back() != ']') throw bad_param(what);
    s = s.substr(1, s.size() - 2);
  }
  std::replace(s.begin(), s.end(), ',', ' ');
  std::istringstream iss(s);
  std::string tok;
  while (iss >> tok)
  {
    char *end = nullptr;
    double v = std::strtod(tok.c_str(), &end);
    if (end == tok.c_str() || *end) throw bad_param(what);
    out.push_back(v);
  }
  return out;
}

std::string format_vector(const std::vector<double> &v)
{
  std::ostringstream oss;
  oss << std::setprecision(std::numeric_limits<double>::max_digits10) << "[ ";
  for (size_t i = 0; i < v.size(); ++i) oss << (i ? ", " : "") << v[i];
  oss << " ]";
  return oss.str();
}

// ------------------------------------------------------------------ CRP -----
static std::string num(double v)
{
  std::ostringstream oss;
  oss << std::setprecision(std::numeric_limits<double>::max_digits10) << v;
  return oss.str();
}
CRP::CRP(std::string n, std::string desc, double v, Mutability m) : name(std::move(n)), type("double"), description(std::move(desc)), def(num(v)), mutability(m) {}
CRP::CRP(std::string n, std::string desc, int v, Mutability m) : name(std::move(n)), type("int"), description(std::move(desc)), def(std::to_string(v)), mutability(m) {}
CRP::CRP(std::string n, std::string desc, const std::string &v, Mutability m) : name(std::move(n)), type("string"), description(std::move(desc)), def(v), mutability(m) {}
CRP::CRP(std::string n, std::string desc, const std::vector<double> &v, Mutability m) : name(std::move(n)), type("vector"), description(std::move(desc)), def(format_vector(v)), mutability(m) {}
CRP::CRP(std::string n, std::string t, std::string desc, Configurable *, bool opt) : name(std::move(n)), type(std::move(t)), description(std::move(desc)), optional(opt), is_object(true) {}
CRP CRP::provided(std::string n, std::string t, std::string desc)
{
  CRP p(std::move(n), std::move(desc), std::string(), Provided);
  p.type = std::move(t);
  return p;
}

// --------------------------------------------------------- Configuration ----
const Configuration::Value &Configuration::operator[](const std::string &k) const
{
  auto it = values_.find(k);
  if (it == values_.end()) throw Exception("Parameter '" + k + "' not set");
  return it->second;
}
Configuration::Value::operator double() const
{ // istream >> double (configuration.h:123-137)
  const std::string t = trim(s);
  char *end = nullptr;
  double v = std::strtod(t.c_str(), &end);
  if (t.empty() || end == t.c_str() || *end) throw Exception("Parameter value '" + s + "' is not a number");
  return v;
}
Configuration::Value::operator int() const { double d = *this; return (int)d; }
std::vector<double> Configuration::Value::v() const { return parse_vector(s, s); }
void Configuration::set(const std::string &k, double v) { set(k, num(v)); }
void Configuration::set(const std::string &k, int v) { set(k, std::to_string(v)); }

// --------------------------------------------------------------- factory ----
std::map<std::string, Creator> &ConfigurableFactory::factories()
{
  static std::map<std::string, Creator> f;
  return f;
}

std::string ConfigurableFactory::normalise(const std::string &type)
{
  auto &f = factories();
  if (f.count(type)) return type;
  std::vector<std::string> hits;
  for (auto &kv : f)
  { // configurable.cpp:106-125: any registered type that ends with the given string
    const std::string &full = kv.first;
    if (full.size() > type.size() && full.compare(full.size() - type.size(), std::string::npos, type) == 0) hits.push_back(full);
  }
  if (hits.size() > 1) log(1, type + " does not specify a unique type. Expanded to " + hits.back());
  if (!hits.empty()) return hits.back();
  return type;
}

Configurable *ConfigurableFactory::create(const std::string &type)
{
  auto &f = factories();
  auto it = f.find(normalise(type));
  if (it == f.end()) return nullptr;
  return it->second();
}

std::string Configurable::path() const { return configurator ? configurator->path() : std::string(); }

// ---------------------------------------------------------- Configurator ----
Configurator *Configurator::child(const std::string &n) const
{
  for (auto &c : children)
    if (c->name == n) return c.get();
  return nullptr;
}
Configurator *Configurator::root() { Configurator *c = this; while (c->parent) c = c->parent; return c; }
std::string Configurator::path() const
{
  if (!parent) return "";
  std::string p = parent->path();
  return p.empty() ? name : p + "/" + name;
}

static Configurator *walk(Configurator *from, const std::string &path)
{
  Configurator *c = from;
  std::istringstream iss(path);
  std::string tok;
  while (c && std::getline(iss, tok, '/'))
  {
    if (tok.empty() || tok == ".") continue;
    if (tok == "..") c = c->parent;
    else
    { // follow references to objects
      Configurator *n = c->child(tok);
      if (!n && c->ref && c->ref->configurator) n = c->ref->configurator->child(tok);
      c = n;
    }
  }
  return c;
}

Configurator *Configurator::find(const std::string &p)
{ // configurable.h:418-449: a leading '/' addresses the root; otherwise relative to this node first, then from the root
  if (!p.empty() && p[0] == '/') return walk(root(), p.substr(1));
  Configurator *c = walk(this, p);
  if (!c) c = walk(root(), p);
  return c;
}

std::string Configurator::yaml(int indent) const
{
  std::ostringstream oss;
  std::string pad((size_t)indent, ' ');
  for (auto &c : children)
  {
    if (c->is_object)
    {
      oss << pad << c->name << ":\n" << pad << "  type: " << c->object->d_type() << "\n" << c->yaml(indent + 2);
    }
    else if (c->ref && c->ref->configurator)
      oss << pad << c->name << ": " << c->ref->configurator->path() << "\n";
    else
      oss << pad << c->name << ": " << (c->value.empty() ? "\"\"" : c->value) << "\n";
  }
  return oss.str();
}

static std::vector<Configurator *> g_order;
const std::vector<Configurator *> &instantiate_order() { return g_order; }

static bool looks_like_path(const std::string &v)
{ // an identifier path such as experiment/agent/policy/projector or ../../projector/memory
  if (v.empty()) return false;
  if (!(std::isalpha((unsigned char)v[0]) || v[0] == '.' || v[0] == '_' || (v[0] == '/' && v.size() > 1))) return false;
  for (char ch : v)
    if (!(std::isalnum((unsigned char)ch) || ch == '/' || ch == '_' || ch == '.')) return false;
  return true;
}

static void instantiate_object(Configurator *node, const YamlNode &y);

// ---- expressions on parameter values (parser.cpp:49-196): scalars and vectors combined left to right by
//   +  -  *   element-wise, a scalar broadcast over a vector (vector sizes must match)
//   ++        concatenation,   --  integer range [a, b),   **  replication
// anything that is not numeric on both sides is put back together as text (a comma-separated list inside [ ] stays a list).
static bool expr_numbers(const std::string &s, std::vector<double> &out)
{
  out.clear();
  std::string t = trim(s);
  if (t.empty()) return false;
  if (t.front() == '[')
  {
    if (t.back() != ']') return false;
    t = t.substr(1, t.size() - 2);
    std::replace(t.begin(), t.end(), ',', ' ');
  }
  else if (t.find_first_of(", \t") != std::string::npos)
    return false;                                      // a bare operand is ONE number (vector.h:61-109); "0," is text
  std::istringstream iss(t);
  std::string tok;
  while (iss >> tok)
  {
    char *end = nullptr;
    const double v = std::strtod(tok.c_str(), &end);
    if (end == tok.c_str() || *end) return false;
    out.push_back(v);
  }
  return !out.empty();
}

static std::string expr_format(const std::vector<double> &z)
{
  std::ostringstream oss;
  oss << std::setprecision(std::numeric_limits<double>::max_digits10);
  if (z.size() == 1) { oss << z[0]; return oss.str(); }
  return format_vector(z);
}

static std::string expr_apply(const std::string &op, const std::string &left, const std::string &right)
{ // ASTNode::evaluate (parser.cpp:49-134)
  std::vector<double> x, y, z;
  if (!expr_numbers(left, x) || !expr_numbers(right, y)) return left + op + right;
  auto each = [&](double (*f)(double, double)) -> bool {
    if (x.size() == 1) { for (double v : y) z.push_back(f(x[0], v)); return true; }
    if (y.size() == 1) { for (double v : x) z.push_back(f(v, y[0])); return true; }
    if (x.size() == y.size()) { for (size_t i = 0; i < x.size(); ++i) z.push_back(f(x[i], y[i])); return true; }
    return false;
  };
  bool ok = true;
  if (op == "+") ok = each([](double a, double b) { return a + b; });
  else if (op == "-") ok = each([](double a, double b) { return a - b; });
  else if (op == "*") ok = each([](double a, double b) { return a * b; });
  else if (op == "++") { z = x; z.insert(z.end(), y.begin(), y.end()); }
  else if (op == "--")
  {
    if (x.size() != 1 || y.size() != 1) throw Exception("Cannot create list from " + left + " to " + right + ": requires scalar operands");
    for (double v = x[0]; v < y[0]; v += 1) z.push_back(v);
  }
  else if (op == "**")
  {
    if (y.size() != 1 || !(y[0] > 0)) throw Exception("Cannot replicate " + left + " " + right + " times: vector size mismatch");
    for (int k = 0; k < (int)y[0]; ++k) z.insert(z.end(), x.begin(), x.end());
  }
  else return left + op + right;
  if (!ok) throw Exception("Cannot combine " + left + " and " + right + " with '" + op + "': vector size mismatch");
  return expr_format(z);
}

static std::string evaluate_expression(const std::string &str)
{ // parseExpression (parser.cpp:136-196): left-associative, sub-expressions in [ ] and ( )
  static const std::string operators = "+*,-", space = " \t";
  if (str.empty()) return str;
  size_t at = 0;
  std::string op, acc;
  bool have = false;
  for (;;)
  {
    while (at < str.size() && space.find(str[at]) != std::string::npos) ++at;
    std::string expr;
    int paren = 0;
    for (; at < str.size() && ((operators.find(str[at]) == std::string::npos && space.find(str[at]) == std::string::npos) || paren); ++at)
    {
      expr.push_back(str[at]);
      if (str[at] == '[' || str[at] == '(') paren++;
      else if ((str[at] == ']' || str[at] == ')') && !--paren) { ++at; break; }
    }
    std::string value;
    if (expr.size() >= 2 && expr.front() == '[' && expr.back() == ']') value = "[ " + evaluate_expression(expr.substr(1, expr.size() - 2)) + " ]";
    else if (expr.size() >= 2 && expr.front() == '(' && expr.back() == ')') value = evaluate_expression(expr.substr(1, expr.size() - 2));
    else value = expr;
    acc = have ? expr_apply(op, acc, value) : value;
    have = true;
    while (at < str.size() && space.find(str[at]) != std::string::npos) ++at;
    if (at < str.size() && operators.find(str[at]) != std::string::npos)
    {
      op.clear();
      for (; at < str.size() && operators.find(str[at]) != std::string::npos; ++at) op.push_back(str[at]);
    }
    else if (at < str.size())
      return expr_apply(" ", acc, evaluate_expression(str.substr(at)));
    else
      return acc;
  }
}

// an expression over references (ParameterConfigurator::str, configurable.cpp:391-432): every identifier between the separators
// ' ' \t [ ] + , - * ( ) that names a parameter node is replaced by that node's value, then the text is evaluated
static bool resolve_expression(Configurator *node, const std::string &raw)
{
  static const std::string seps = " \t[]+,-*()";
  if (raw.find_first_of("+*") == std::string::npos) return false;
  std::string out, id;
  bool any = false;
  auto flush = [&]() {
    if (id.empty()) return;
    Configurator *target = nullptr;
    if (looks_like_path(id) && !std::isdigit((unsigned char)id[0]))
    {
      target = node->parent ? node->parent->find(id) : nullptr;
      if (!target && node->parent) target = node->find(id);
    }
    if (target)
    {
      if (target->is_object || target->ref) throw Exception(node->path() + ": '" + id + "' names an object inside the expression '" + raw + "'");
      out += target->value;
      any = true;
    }
    else
      out += id;
    id.clear();
  };
  for (char ch : raw)
  {
    if (seps.find(ch) != std::string::npos) { flush(); out.push_back(ch); }
    else id.push_back(ch);
  }
  flush();
  if (!any) return false;
  node->value = evaluate_expression(out);
  return true;
}

// value of a parameter node: literal, or the value / object another node holds, or an expression over such values
static void resolve_parameter(Configurator *node, const std::string &raw)
{
  node->value = raw;
  if (resolve_expression(node, raw)) return;
  if (!looks_like_path(raw)) return;
  Configurator *target = node->parent ? node->parent->find(raw) : nullptr;     // relative to the owning object first
  if (!target && node->parent) target = node->find(raw);
  if (!target) return;                                                         // a plain string (e.g. save_every: never)
  if (target->is_object) { node->ref = target->object.get(); node->value = ""; }
  else if (target->ref) { node->ref = target->ref; node->value = ""; }
  else node->value = target->value;
}

static void instantiate_object(Configurator *node, const YamlNode &y)
{
  const YamlNode *ty = y.find("type");
  if (!ty) throw Exception(node->path() + ": object has no type");
  const std::string type = ConfigurableFactory::normalise(ty->scalar);
  Configurable *obj = ConfigurableFactory::create(type);
  if (!obj) throw Exception(node->path() + ": unknown object type '" + ty->scalar + "' (not part of the accelerated path)");
  node->is_object = true;
  node->object.reset(obj);
  obj->configurator = node;

  // role = suffix of the type the PARENT requested for this parameter (configurable.h:191-204)
  std::string role;
  if (node->parent && node->parent->is_object)
  {
    ConfigurationRequest preq;
    node->parent->object->request("", &preq);   // role-independent for the classes that request objects
    for (auto &p : preq)
      if (p.name == node->name)
      {
        size_t dot = p.type.find('.');
        if (dot != std::string::npos) role = p.type.substr(dot + 1);
        std::string base = p.type.substr(0, dot);
        if (obj->d_type().compare(0, base.size(), base) != 0)
          throw Exception(node->path() + ": object of type '" + obj->d_type() + "' given where '" + base + "' is required");
      }
  }
  ConfigurationRequest req;
  obj->request(role, &req);

  // children in YAML order (configurable.cpp:627-654); unknown keys are ignored silently (:640-653)
  for (auto &kv : y.children)
  {
    if (kv.first == "type") continue;
    const CRP *crp = nullptr;
    for (auto &p : req)
      if (p.name == kv.first) crp = &p;
    if (!crp) { log(4, node->path() + ": ignoring unknown parameter '" + kv.first + "'"); continue; }
    auto ch = std::make_unique<Configurator>();
    ch->name = kv.first;
    ch->parent = node;
    Configurator *raw = ch.get();
    node->children.push_back(std::move(ch));
    if (kv.second.is_map)
      instantiate_object(raw, kv.second);
    else
      resolve_parameter(raw, kv.second.scalar);
  }
  // defaults (configurable.cpp:657-685)
  for (auto &p : req)
  {
    if (p.mutability == CRP::Provided || node->child(p.name)) continue;
    if (p.is_object)
    {
      if (!p.optional) throw Exception(node->path() + ": required parameter '" + p.name + "' is undefined");
      continue;
    }
    auto ch = std::make_unique<Configurator>();
    ch->name = p.name;
    ch->parent = node;
    ch->value = p.def;
    node->children.push_back(std::move(ch));
  }
  // configure (configurable.cpp:688)
  Configuration &cfg = node->config;
  for (auto &p : req)
  {
    Configurator *c = node->child(p.name);
    if (!c) { if (p.is_object) cfg.put(p.name, "", nullptr); continue; }
    if (p.is_object)
    {
      Configurable *ptr = c->ptr();
      if (!ptr && !(c->value == "0" || c->value.empty())) throw Exception(c->path() + ": does not name an object");
      if (!ptr && !p.optional) throw Exception(node->path() + ": required parameter '" + p.name + "' is undefined");
      cfg.put(p.name, "", ptr);
    }
    else
    {
      if (c->ref) throw Exception(c->path() + ": names an object where a value is required");
      cfg.put(p.name, c->value, nullptr);
    }
  }
  cfg.clear_provided();
  g_order.push_back(node);
  obj->configure(cfg);
  // provided parameters become ordinary nodes (configurable.cpp:691-712)
  for (auto &k : cfg.provided())
  {
    if (node->child(k)) { node->child(k)->value = cfg[k].str(); continue; }
    auto ch = std::make_unique<Configurator>();
    ch->name = k;
    ch->parent = node;
    ch->value = cfg[k].str();
    node->children.push_back(std::move(ch));
  }
}

std::unique_ptr<Configurator> instantiate(const YamlNode &y)
{
  g_order.clear();
  auto root = std::make_unique<Configurator>();
  for (auto &kv : y.children)
  {
    auto ch = std::make_unique<Configurator>();
    ch->name = kv.first;
    ch->parent = root.get();
    Configurator *raw = ch.get();
    root->children.push_back(std::move(ch));
    if (kv.second.is_map)
    {
      const YamlNode *ty = kv.second.find("type");
      // top-level sections the accelerated path does not implement (visualizer, visualization...) are skipped loudly
      if (ty && !ConfigurableFactory::factories().count(ConfigurableFactory::normalise(ty->scalar)))
      {
        log(1, "skipping top-level section '" + kv.first + "' of type '" + ty->scalar + "' (outside the accelerated path)");
        root->children.pop_back();
        continue;
      }
      instantiate_object(raw, kv.second);
    }
    else
      resolve_parameter(raw, kv.second.scalar);
  }
  return root;
}

} // namespace grlx_host
