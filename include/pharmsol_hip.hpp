// pharmsol_hip.hpp — header-only C++17 host facade over the C ABI (include/pmx.h).
//
// Mirrors the reference's interface for the prediction path so that C++ callers (and the parity tests in
// tests/cpp/) read like pharmsol code:
//   Subject::builder("id").bolus(..).infusion(..).observation(..).covariate(..).repeat(..).reset().build()
//                                                                  src/data/builder.rs:38-50,113-361
//   Parameters::dense({..}) / Parameters::with_model(model, {{"ke", 0.1}, ..})   src/parameters.rs:74-102
//   equation::Analytical / equation::ODE  + with_nstates/with_ndrugs/with_nout    analytical/mod.rs:102-138
//   model.estimate_predictions(subject, parameters) -> SubjectPredictions         equation/mod.rs:526-532
//   model.estimate_predictions_matrix(data, theta)  (the loop nest of likelihood/matrix.rs:79-98)
// Errors are exceptions carrying the pmx_status and the library's message (PharmsolError, error/mod.rs:13-49).
// All compute happens in libpmx_hip.so on the GPU; there is no CPU path behind this header.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "pmx.h"

namespace pharmsol {

struct Error : std::runtime_error {
  int32_t status;
  Error(int32_t st, const std::string& msg) : std::runtime_error(msg), status(st) {}
};
inline void check(int32_t rc, bool allow_pair_failures = false) {
  if (rc == PMX_OK || (allow_pair_failures && rc == PMX_ERR_PAIR_FAILED)) return;
  throw Error(rc, pmx_last_error());
}

// ---------------------------------------------------------------- data model (src/data)
struct Event {
  uint8_t kind;  // PMX_EV_*
  double time, value, duration;
  std::string label;  // input label (doses) / output label (observations): public labels, resolved by the model
  bool missing = false;
};

struct Occasion {
  int index = 0;
  std::vector<Event> events;
  std::map<std::string, std::vector<std::pair<double, double>>> covariates;  // BTreeMap order
  void sort() {  // Occasion::sort: time.total_cmp, then Observation < Bolus < Infusion; stable (event.rs:292-304)
    auto key = [](double v) {
      int64_t b;
      std::memcpy(&b, &v, 8);
      return b ^ static_cast<int64_t>(static_cast<uint64_t>(b >> 63) >> 1);
    };
    std::stable_sort(events.begin(), events.end(), [&](const Event& a, const Event& b) {
      const int64_t ka = key(a.time), kb = key(b.time);
      return ka != kb ? ka < kb : a.kind < b.kind;
    });
  }
};

class SubjectBuilder;
class Subject {
 public:
  static SubjectBuilder builder(const std::string& id);
  const std::string& id() const { return id_; }
  const std::vector<Occasion>& occasions() const { return occasions_; }

 private:
  friend class SubjectBuilder;
  Subject(std::string id, std::vector<Occasion> occ) : id_(std::move(id)), occasions_(std::move(occ)) {
    for (auto& o : occasions_) o.sort();  // Subject::new (structs.rs:363-369)
  }
  std::string id_;
  std::vector<Occasion> occasions_;
};

class SubjectBuilder {
 public:
  explicit SubjectBuilder(std::string id) : id_(std::move(id)) {}
  SubjectBuilder& bolus(double t, double amount, const std::string& input) {
    return push({PMX_EV_BOLUS, t, amount, 0.0, input});
  }
  SubjectBuilder& bolus(double t, double amount, int input) { return bolus(t, amount, std::to_string(input)); }
  SubjectBuilder& infusion(double t, double amount, const std::string& input, double duration) {
    return push({PMX_EV_INFUSION, t, amount, duration, input});
  }
  SubjectBuilder& infusion(double t, double amount, int input, double duration) {
    return infusion(t, amount, std::to_string(input), duration);
  }
  SubjectBuilder& observation(double t, double value, const std::string& outeq) {
    return push({PMX_EV_OBSERVATION, t, value, 0.0, outeq});
  }
  SubjectBuilder& observation(double t, double value, int outeq) { return observation(t, value, std::to_string(outeq)); }
  SubjectBuilder& missing_observation(double t, const std::string& outeq) {
    Event e{PMX_EV_OBSERVATION, t, std::numeric_limits<double>::quiet_NaN(), 0.0, outeq};
    e.missing = true;
    return push(e);
  }
  SubjectBuilder& missing_observation(double t, int outeq) { return missing_observation(t, std::to_string(outeq)); }
  SubjectBuilder& repeat(int n, double delta) {  // builder.rs:251-313
    if (!last_) return *this;
    const Event base = *last_;
    for (int i = 1; i <= n; ++i) {
      Event e = base;
      e.time = base.time + delta * static_cast<double>(i);
      push(e);
    }
    return *this;
  }
  SubjectBuilder& covariate(const std::string& name, double t, double value) {
    covariates_[name].emplace_back(t, value);
    return *this;
  }
  SubjectBuilder& reset() {  // builder.rs:315-326
    current_.sort();
    current_.covariates = covariates_;
    occasions_.push_back(current_);
    const int next = current_.index + 1;
    current_ = Occasion{};
    current_.index = next;
    covariates_.clear();
    last_.reset();
    return *this;
  }
  Subject build() {
    reset();
    return Subject(id_, occasions_);
  }

 private:
  SubjectBuilder& push(const Event& e) {
    last_ = e;
    current_.events.push_back(e);
    current_.sort();  // add_event re-sorts (structs.rs:713-716)
    return *this;
  }
  std::string id_;
  std::vector<Occasion> occasions_;
  Occasion current_;
  std::map<std::string, std::vector<std::pair<double, double>>> covariates_;
  std::optional<Event> last_;
};
inline SubjectBuilder Subject::builder(const std::string& id) { return SubjectBuilder(id); }

using Data = std::vector<Subject>;

// ---------------------------------------------------------------- predictions (likelihood/prediction.rs, subject.rs)
struct Prediction {
  double time;
  std::optional<double> observation;
  double prediction;
  int outeq;
  int occasion;
};
class SubjectPredictions {
 public:
  std::vector<Prediction> predictions;
  std::vector<double> flat_predictions() const {
    std::vector<double> v;
    for (const auto& p : predictions) v.push_back(p.prediction);
    return v;
  }
  std::vector<double> flat_times() const {
    std::vector<double> v;
    for (const auto& p : predictions) v.push_back(p.time);
    return v;
  }
};

// ---------------------------------------------------------------- error models (error_model.rs:1045-1080)
/// AssayErrorModel::additive(ErrorPoly::new(c0..c3), lambda) / ::proportional(.., gamma): sigma from the OBSERVATION,
/// alpha = c0 + c1 y + c2 y^2 + c3 y^3; additive sqrt(alpha^2 + lambda^2), proportional gamma * alpha.
struct AssayErrorModel {
  static pmx_error_model additive(double c0, double c1, double c2, double c3, double lambda) {
    return pmx_error_model{PMX_EM_ADDITIVE, 0, {c0, c1, c2, c3}, lambda};
  }
  static pmx_error_model proportional(double c0, double c1, double c2, double c3, double gamma) {
    return pmx_error_model{PMX_EM_PROPORTIONAL, 0, {c0, c1, c2, c3}, gamma};
  }
  static pmx_error_model none() { return pmx_error_model{PMX_EM_NONE, 0, {0, 0, 0, 0}, 0}; }
};

/// ParameterOrder (src/parameter_order.rs:12-16; parameters.rs:125-145): the permutation between a caller's parameter
/// columns (e.g. the order of an NPAG support-point table) and the model's declaration order.
class ParameterOrder {
 public:
  /// `source` = the caller's column names, `model` = the model's parameter names; both must hold the same set.
  ParameterOrder(const std::vector<std::string>& source, const std::vector<std::string>& model) {
    if (source.size() != model.size()) throw Error(PMX_ERR_INVALID_ARGUMENT, "parameter count differs from the model's");
    perm_.resize(model.size());
    for (size_t i = 0; i < model.size(); ++i) {
      auto it = std::find(source.begin(), source.end(), model[i]);
      if (it == source.end()) throw Error(PMX_ERR_INVALID_ARGUMENT, "missing parameter '" + model[i] + "'");
      if (std::count(source.begin(), source.end(), model[i]) != 1)
        throw Error(PMX_ERR_INVALID_ARGUMENT, "parameter '" + model[i] + "' given twice");
      perm_[i] = static_cast<size_t>(it - source.begin());
    }
  }
  bool is_identity() const {
    for (size_t i = 0; i < perm_.size(); ++i)
      if (perm_[i] != i) return false;
    return true;
  }
  /// rows of `n` source-order values -> model order, row by row
  std::vector<double> reorder(const std::vector<double>& theta_source) const {
    const size_t k = perm_.size();
    if (k == 0 || theta_source.size() % k) throw Error(PMX_ERR_INVALID_ARGUMENT, "theta is not a whole number of rows");
    std::vector<double> out(theta_source.size());
    for (size_t r = 0; r < theta_source.size() / k; ++r)
      for (size_t i = 0; i < k; ++i) out[r * k + i] = theta_source[r * k + perm_[i]];
    return out;
  }

 private:
  std::vector<size_t> perm_;
};

namespace equation {

struct Route {
  enum Kind { Bolus, Infusion } kind;
  std::string name;
  int dest;
  static Route bolus(std::string n, int dest) { return {Bolus, std::move(n), dest}; }
  static Route infusion(std::string n, int dest) { return {Infusion, std::move(n), dest}; }
};

// Shared model state; Analytical and ODE differ in eq_kind / kernel tables only.
class Equation {
 public:
  ~Equation() {
    if (handle_) pmx_model_destroy(handle_);
  }
  Equation(const Equation& o) : desc_(o.desc_), source_(o.source_), has_init_(o.has_init_), user_functions_(o.user_functions_), params_(o.params_),
                                outputs_(o.outputs_), routes_(o.routes_), covariates_(o.covariates_),
                                has_metadata_(o.has_metadata_) {}
  Equation& operator=(const Equation&) = delete;

  Equation& with_nstates(int n) { desc_.nstates = n; return invalidate(); }
  Equation& with_ndrugs(int n) { desc_.ndrugs = n; return invalidate(); }
  Equation& with_nout(int n) { desc_.nout = n; return invalidate(); }
  /// y[outeq] = x[state] / theta[vol_param]   (vol_param < 0: y = x[state])
  Equation& with_output(int outeq, int state, int vol_param) {
    desc_.out[outeq].state = state;
    desc_.out[outeq].vol_src = vol_param >= 0 ? PMX_SRC_PRIMARY : PMX_SRC_NONE;
    desc_.out[outeq].vol_index = vol_param >= 0 ? vol_param : 0;
    return invalidate();
  }
  /// derived[d] = theta[src_param] * (cov/ref)^coef  — one line of a `derive:` block (examples/analytical_readme.rs:18-20)
  Equation& with_derived_pow(int d, int src_param, int cov, double ref, double coef) {
    desc_.n_derived = std::max(desc_.n_derived, d + 1);
    desc_.derived[d].src_param = src_param;
    desc_.derived[d].n_factors = 1;
    desc_.derived[d].f[0] = pmx_factor{PMX_F_POW, cov, ref, coef};
    return invalidate();
  }
  /// Kernel-order parameter j <- (PMX_SRC_PRIMARY | PMX_SRC_DERIVED, index): the macro's projection wrapper
  /// (pharmsol-macros/src/expand/analytical.rs:208-294).
  Equation& with_bind(const std::vector<std::pair<int, int>>& binds) {
    desc_.n_bind = static_cast<int32_t>(binds.size());
    for (size_t j = 0; j < binds.size(); ++j) desc_.bind[j] = pmx_bind{binds[j].first, binds[j].second};
    return invalidate();
  }
  /// Attach names: parameters, outputs and routes (metadata::new(..).parameters(..).outputs(..).routes(..)).
  Equation& with_metadata(std::vector<std::string> params, std::vector<std::string> outputs, std::vector<Route> routes,
                          std::vector<std::string> covariates = {}) {
    params_ = std::move(params);
    outputs_ = std::move(outputs);
    routes_ = std::move(routes);
    covariates_ = std::move(covariates);
    has_metadata_ = true;
    desc_.nparams = static_cast<int32_t>(params_.size());
    desc_.n_covariates = static_cast<int32_t>(covariates_.size());
    int nb = 0, ni = 0;
    for (const auto& r : routes_) {  // per-kind numbering in declaration order (metadata.rs:926-946)
      if (r.kind == Route::Bolus) desc_.bolus_dest[nb++] = r.dest; else desc_.infusion_dest[ni++] = r.dest;
    }
    return invalidate();
  }
  const std::vector<std::string>& params() const { return params_; }
  const pmx_model_desc& desc() const { return desc_; }

  // EquationPriv::resolve_input_label / resolve_output_label (equation/mod.rs:195-245)
  int resolve_input_label(const std::string& label, Route::Kind kind) const {
    if (has_metadata_) {
      int nb = 0, ni = 0;
      for (const auto& r : routes_) {
        const int idx = r.kind == Route::Bolus ? nb++ : ni++;
        if (r.kind == kind && (r.name == label || (is_numeric(label) && r.name == "input_" + label))) return idx;
      }
      throw Error(PMX_ERR_INVALID_ARGUMENT, "unknown input label '" + label + "'");
    }
    if (!is_numeric(label)) throw Error(PMX_ERR_INVALID_ARGUMENT, "unknown input label '" + label + "' (no metadata)");
    return std::stoi(label);
  }
  int resolve_output_label(const std::string& label) const {
    if (has_metadata_) {
      for (size_t i = 0; i < outputs_.size(); ++i)
        if (outputs_[i] == label || (is_numeric(label) && outputs_[i] == "outeq_" + label)) return static_cast<int>(i);
      throw Error(PMX_ERR_INVALID_ARGUMENT, "unknown output label '" + label + "'");
    }
    if (!is_numeric(label)) throw Error(PMX_ERR_INVALID_ARGUMENT, "unknown output label '" + label + "' (no metadata)");
    return std::stoi(label);
  }

  /// Equation::estimate_predictions (equation/mod.rs:526-532): one subject, one support point.
  SubjectPredictions estimate_predictions(const Subject& subject, const std::vector<double>& parameters, int device = 0) {
    std::vector<double> pred;
    std::vector<uint8_t> status;
    predict_matrix({subject}, parameters, 1, device, &pred, &status, /*throw_on_pair_failure=*/true);
    SubjectPredictions out;
    size_t row = 0;
    for (const auto& occ : subject.occasions())
      for (const auto& e : occ.events)
        if (e.kind == PMX_EV_OBSERVATION)
          out.predictions.push_back({e.time, e.missing ? std::nullopt : std::optional<double>(e.value), pred[row++],
                                     resolve_output_label(e.label), occ.index});
    return out;
  }

  /// Every subject x every support point; pred is [n_observations x n_support] row-major,
  /// status [n_subjects x n_support].  theta is [n_support x nparams] row-major (matrix.rs:62-65).
  void predict_matrix(const Data& data, const std::vector<double>& theta, int64_t n_support, int device,
                      std::vector<double>* pred, std::vector<uint8_t>* status, bool throw_on_pair_failure = false) {
    if (static_cast<int64_t>(theta.size()) != n_support * desc_.nparams)
      throw Error(PMX_ERR_INVALID_ARGUMENT, "theta must hold n_support x nparams values");
    Flat flat = flatten(data);
    pmx_population* pop = nullptr;
    pmx_population_desc d = flat.desc();
    check(pmx_population_create(&d, device, &pop));
    struct Guard { pmx_population* p; ~Guard() { pmx_population_destroy(p); } } guard{pop};
    const int64_t n_obs = pmx_population_n_observations(pop);
    pred->assign(static_cast<size_t>(n_obs * n_support), std::numeric_limits<double>::quiet_NaN());
    status->assign(data.size() * static_cast<size_t>(n_support), 0);
    check(pmx_predict(handle(), pop, theta.data(), n_support, pred->data(), n_support, status->data()),
          !throw_on_pair_failure);
  }

  /// log_likelihood_matrix(&eq, &data, &theta, &error_models, _) (likelihood/matrix.rs:52-106): ll is
  /// [n_subjects x n_support] row-major.  `error_models[o]` = the model of output equation o (kind PMX_EM_NONE for
  /// outputs without one); observed values come from the subjects' `observation(..)` events.
  void log_likelihood_matrix(const Data& data, const std::vector<double>& theta, int64_t n_support,
                             const std::vector<pmx_error_model>& error_models, int device, std::vector<double>* ll,
                             std::vector<uint8_t>* status, bool throw_on_pair_failure = false) {
    if (static_cast<int64_t>(theta.size()) != n_support * desc_.nparams)
      throw Error(PMX_ERR_INVALID_ARGUMENT, "theta must hold n_support x nparams values");
    if (static_cast<int>(error_models.size()) != desc_.nout)
      throw Error(PMX_ERR_INVALID_ARGUMENT, "one error model per output equation");
    Flat flat = flatten(data);
    pmx_population* pop = nullptr;
    pmx_population_desc d = flat.desc();
    check(pmx_population_create(&d, device, &pop));
    struct Guard { pmx_population* p; ~Guard() { pmx_population_destroy(p); } } guard{pop};
    ll->assign(data.size() * static_cast<size_t>(n_support), std::numeric_limits<double>::quiet_NaN());
    status->assign(data.size() * static_cast<size_t>(n_support), 0);
    check(pmx_loglik(handle(), pop, error_models.data(), theta.data(), n_support, ll->data(), n_support, status->data()),
          !throw_on_pair_failure);
  }

  /// Equation::estimate_log_likelihood (equation/mod.rs:468-477): one subject, one support point.
  double estimate_log_likelihood(const Subject& subject, const std::vector<double>& parameters,
                                 const std::vector<pmx_error_model>& error_models, int device = 0) {
    std::vector<double> ll;
    std::vector<uint8_t> status;
    log_likelihood_matrix({subject}, parameters, 1, error_models, device, &ll, &status, /*throw_on_pair_failure=*/true);
    return ll[0];
  }

  /// log_likelihood_batch(&eq, &data, &parameters, &error_models) (likelihood/mod.rs:119-177): subject s under ITS OWN
  /// row of `theta` ([n_subjects x nparams]); a subject whose simulation or likelihood fails gets -inf
  /// (likelihood/mod.rs:137-140) and a status byte saying why - the call itself succeeds.
  std::vector<double> log_likelihood_batch(const Data& data, const std::vector<double>& theta,
                                           const std::vector<pmx_error_model>& error_models, int device = 0,
                                           std::vector<uint8_t>* status = nullptr) {
    if (theta.size() != data.size() * static_cast<size_t>(desc_.nparams))
      throw Error(PMX_ERR_INVALID_ARGUMENT, "theta must hold one row of nparams values per subject");
    if (static_cast<int>(error_models.size()) != desc_.nout)
      throw Error(PMX_ERR_INVALID_ARGUMENT, "one error model per output equation");
    Flat flat = flatten(data);
    pmx_population* pop = nullptr;
    pmx_population_desc d = flat.desc();
    check(pmx_population_create(&d, device, &pop));
    struct Guard { pmx_population* p; ~Guard() { pmx_population_destroy(p); } } guard{pop};
    std::vector<double> ll(data.size(), std::numeric_limits<double>::quiet_NaN());
    std::vector<uint8_t> st(data.size(), 0);
    check(pmx_loglik_batch(handle(), pop, error_models.data(), theta.data(), ll.data(), st.data()), true);
    if (status) *status = std::move(st);
    return ll;
  }

  /// One theta row per subject, predictions only (the batch twin of predict_matrix; pmx_predict_batch).
  std::vector<double> predict_batch(const Data& data, const std::vector<double>& theta, int device = 0,
                                    std::vector<uint8_t>* status = nullptr) {
    if (theta.size() != data.size() * static_cast<size_t>(desc_.nparams))
      throw Error(PMX_ERR_INVALID_ARGUMENT, "theta must hold one row of nparams values per subject");
    Flat flat = flatten(data);
    pmx_population* pop = nullptr;
    pmx_population_desc d = flat.desc();
    check(pmx_population_create(&d, device, &pop));
    struct Guard { pmx_population* p; ~Guard() { pmx_population_destroy(p); } } guard{pop};
    std::vector<double> pred(static_cast<size_t>(pmx_population_n_observations(pop)), std::numeric_limits<double>::quiet_NaN());
    std::vector<uint8_t> st(data.size(), 0);
    check(pmx_predict_batch(handle(), pop, theta.data(), pred.data(), st.data()), true);
    if (status) *status = std::move(st);
    return pred;
  }

  /// A population kept on the device across calls - what an NPAG loop holds: flattened, label-resolved and compiled
  /// once; every cycle calls log_likelihood_matrix with a new support-point table.  (`Data` in the reference is borrowed
  /// by every `log_likelihood_matrix` call; here the borrow is the handle.)
  class Resident {
   public:
    Resident(Equation& eq, const Data& data, int device = 0) : eq_(eq), n_subjects_(static_cast<int64_t>(data.size())) {
      Flat flat = eq.flatten(data);
      pmx_population_desc d = flat.desc();
      check(pmx_population_create(&d, device, &pop_));
    }
    ~Resident() {
      if (pop_) pmx_population_destroy(pop_);
    }
    Resident(const Resident&) = delete;
    Resident& operator=(const Resident&) = delete;
    int64_t n_subjects() const { return n_subjects_; }
    int64_t n_observations() const { return pmx_population_n_observations(pop_); }
    int64_t n_events() const { return pmx_population_n_events(pop_); }
    /// ll [n_subjects x n_support] row-major (pass a pmx_host_alloc'ed buffer for a copy at link rate)
    void log_likelihood_matrix(const double* theta, int64_t n_support, const std::vector<pmx_error_model>& error_models,
                               double* ll, uint8_t* status, bool throw_on_pair_failure = false) {
      check(pmx_loglik(eq_.handle(), pop_, error_models.data(), theta, n_support, ll, n_support, status), !throw_on_pair_failure);
    }
    /// pred [n_observations x n_support] row-major
    void predict_matrix(const double* theta, int64_t n_support, double* pred, uint8_t* status, bool throw_on_pair_failure = false) {
      check(pmx_predict(eq_.handle(), pop_, theta, n_support, pred, n_support, status), !throw_on_pair_failure);
    }
    const pmx_population* handle() const { return pop_; }

   private:
    Equation& eq_;
    pmx_population* pop_ = nullptr;
    int64_t n_subjects_;
  };

  // Flattened population (the pmx_population_desc arrays), exposed for tests.
  struct Flat {
    std::vector<int64_t> subj_occ_off{0}, occ_ev_off{0}, cov_knot_off{0};
    std::vector<int32_t> occ_index;
    std::vector<double> t, v, dur, knot_t, knot_v;
    std::vector<uint8_t> kind, fixed;
    std::vector<uint16_t> io;
    int32_t n_cov = 0;
    pmx_population_desc desc() const {
      pmx_population_desc d{};
      d.n_subjects = static_cast<int64_t>(subj_occ_off.size()) - 1;
      d.n_occasions = static_cast<int64_t>(occ_ev_off.size()) - 1;
      d.n_events = static_cast<int64_t>(t.size());
      d.subj_occ_off = subj_occ_off.data();
      d.occ_ev_off = occ_ev_off.data();
      d.occ_index = occ_index.data();
      d.ev_time = t.data();
      d.ev_value = v.data();
      d.ev_duration = dur.data();
      d.ev_kind = kind.data();
      d.ev_io = io.data();
      d.n_covariates = n_cov;
      d.presorted = 0;
      d.cov_knot_off = n_cov ? cov_knot_off.data() : nullptr;
      d.cov_knot_time = n_cov ? knot_t.data() : nullptr;
      d.cov_knot_value = n_cov ? knot_v.data() : nullptr;
      d.cov_fixed = nullptr;
      return d;
    }
  };
  Flat flatten(const Data& data) const {
    Flat f;
    f.n_cov = static_cast<int32_t>(covariates_.size());
    for (const auto& s : data) {
      for (const auto& occ : s.occasions()) {
        f.occ_index.push_back(occ.index);
        for (const auto& e : occ.events) {
          f.t.push_back(e.time);
          f.v.push_back(e.value);
          f.dur.push_back(e.duration);
          f.kind.push_back(e.kind);
          f.io.push_back(static_cast<uint16_t>(
              e.kind == PMX_EV_OBSERVATION ? resolve_output_label(e.label)
                                           : resolve_input_label(e.label, e.kind == PMX_EV_BOLUS ? Route::Bolus
                                                                                                 : Route::Infusion)));
        }
        f.occ_ev_off.push_back(static_cast<int64_t>(f.t.size()));
        for (const auto& name : covariates_) {
          auto it = occ.covariates.find(name);
          if (it == occ.covariates.end() || it->second.empty())
            throw Error(PMX_ERR_INVALID_ARGUMENT, "Covariate " + name + " not found");  // fetch_cov! (lib.rs:433-443)
          for (const auto& kv : it->second) {
            f.knot_t.push_back(kv.first);
            f.knot_v.push_back(kv.second);
          }
          f.cov_knot_off.push_back(static_cast<int64_t>(f.knot_t.size()));
        }
      }
      f.subj_occ_off.push_back(static_cast<int64_t>(f.occ_index.size()));
    }
    return f;
  }

 protected:
  Equation(int32_t eq_kind, int32_t kernel, int32_t nparams) {
    std::memset(&desc_, 0, sizeof desc_);
    desc_.eq_kind = eq_kind;
    desc_.kernel = kernel;
    desc_.nstates = desc_.ndrugs = desc_.nout = 5;  // Neqs::default (analytical/mod.rs:93)
    desc_.nparams = nparams;
    desc_.rk4_h_max = 0.02;
    desc_.ode_solver = PMX_SOLVER_RK4;
    desc_.ode_rtol = desc_.ode_atol = 1e-4;  // the reference's defaults (ode/mod.rs:126-127)
    for (int i = 0; i < PMX_MAX_STATES; ++i) desc_.init_param[i] = -1;
    for (int i = 0; i < PMX_MAX_INPUTS; ++i)
      desc_.lag_param[i] = desc_.fa_param[i] = desc_.bolus_dest[i] = desc_.infusion_dest[i] = -1;
  }
  pmx_model_desc desc_;
  std::string source_;  // user closures as source text (ODE::custom, Analytical::with_closures)
  bool has_init_ = false;
  uint32_t user_functions_ = 0;  // PMX_FN_* bits of the closures `source_` defines (pmx_model_create_user)
  Equation& invalidate() {
    if (handle_) pmx_model_destroy(handle_);
    handle_ = nullptr;
    return *this;
  }

 private:
  static bool is_numeric(const std::string& s) {
    return !s.empty() && std::all_of(s.begin(), s.end(), [](char c) { return c >= '0' && c <= '9'; });
  }
  pmx_model* handle() {
    if (!handle_) {
      if (!source_.empty() && user_functions_ != 0)
        check(pmx_model_create_user(&desc_, source_.c_str(), user_functions_, &handle_));
      else if (!source_.empty())
        check(pmx_model_create_custom(&desc_, source_.c_str(), has_init_ ? 1 : 0, &handle_));
      else
        check(pmx_model_create(&desc_, &handle_));
    }
    return handle_;
  }

  std::vector<std::string> params_, outputs_;
  std::vector<Route> routes_;
  std::vector<std::string> covariates_;
  bool has_metadata_ = false;
  pmx_model* handle_ = nullptr;
};

/// Analytical::new(eq, ..) with a built-in structure (analytical/mod.rs:102; kernels: PMX_K_*).
class Analytical : public Equation {
 public:
  Analytical(int32_t kernel, int32_t nparams) : Equation(PMX_EQ_ANALYTICAL, kernel, nparams) {}
  /// Analytical::new(eq, seq_eq, lag, fa, init, out) with the closures given as source text (pmx.h, "user closures"):
  /// `functions` = PMX_FN_* bits of the roles `source` defines (pmx_derive / pmx_route_lag / pmx_route_bioavailability /
  /// pmx_init / pmx_outputs / pmx_seq_eq / pmx_eq), all taking (t, x, p, cov, rateiv, derived, out) like the reference's
  /// compiled kernels (src/dsl/native.rs:45-53); compiled for gfx950 with hiprtc on first use.  `kernel` stays the
  /// built-in structure the closures wrap, or PMX_K_CUSTOM with PMX_FN_EQ.  `cov[c]` follows the covariate order of
  /// with_metadata(.., covariates).
  Analytical& with_closures(const std::string& source, uint32_t functions, int n_derived = 0) {
    source_ = source;
    user_functions_ = functions;
    desc_.n_derived = n_derived;
    invalidate();
    return *this;
  }
};
/// ODE::new(diffeq, ..) with a built-in diffeq body (ode/mod.rs:115; PMX_ODE_*), integrated with fixed-step RK4.
class ODE : public Equation {
 public:
  ODE(int32_t model, int32_t nparams, double h_max = 0.02) : Equation(PMX_EQ_ODE, model, nparams) { desc_.rk4_h_max = h_max; }
  /// ODE::new(diffeq, lag, fa, init, out) with USER bodies: `source` defines pmx_dynamics / pmx_outputs (/ pmx_init)
  /// as described in pmx.h; the library compiles it for gfx950 with hiprtc.
  static ODE custom(const std::string& source, int nstates, int nparams, int ndrugs = 1, int nout = 1, bool has_init = false,
                    double h_max = 0.02) {
    ODE m(PMX_ODE_CUSTOM, nparams, h_max);
    m.with_nstates(nstates).with_ndrugs(ndrugs).with_nout(nout);
    m.source_ = source;
    m.has_init_ = has_init;
    return m;
  }
  /// ODE::with_solver / with_tolerances (ode/mod.rs:134-166): PMX_SOLVER_RK4 (fixed step) | PMX_SOLVER_DOPRI5 (adaptive,
  /// explicit) | PMX_SOLVER_ROS2 (adaptive, L-stable: the stiff option, the role of OdeSolver::Bdf / Sdirk).
  ODE& with_solver(int32_t solver) { desc_.ode_solver = solver; invalidate(); return *this; }
  ODE& with_tolerances(double rtol, double atol) { desc_.ode_rtol = rtol; desc_.ode_atol = atol; invalidate(); return *this; }
  ODE& with_step(double h_max) { desc_.rk4_h_max = h_max; invalidate(); return *this; }
};

}  // namespace equation

/// Parameters: one support point in model order (src/parameters.rs:51,74-102).
class Parameters {
 public:
  static std::vector<double> dense(std::initializer_list<double> v) { return std::vector<double>(v); }
  static std::vector<double> with_model(const equation::Equation& model,
                                        std::initializer_list<std::pair<std::string, double>> named) {
    const auto& names = model.params();
    if (names.empty()) throw Error(PMX_ERR_INVALID_ARGUMENT, "model declares no parameter names");
    std::vector<double> out(names.size(), std::numeric_limits<double>::quiet_NaN());
    std::vector<bool> seen(names.size(), false);
    for (const auto& kv : named) {
      auto it = std::find(names.begin(), names.end(), kv.first);
      if (it == names.end()) throw Error(PMX_ERR_INVALID_ARGUMENT, "unknown parameter '" + kv.first + "'");
      const size_t i = static_cast<size_t>(it - names.begin());
      if (seen[i]) throw Error(PMX_ERR_INVALID_ARGUMENT, "parameter '" + kv.first + "' given twice");
      seen[i] = true;
      out[i] = kv.second;
    }
    for (size_t i = 0; i < names.size(); ++i)
      if (!seen[i]) throw Error(PMX_ERR_INVALID_ARGUMENT, "missing parameter '" + names[i] + "'");
    return out;
  }
};

}  // namespace pharmsol
