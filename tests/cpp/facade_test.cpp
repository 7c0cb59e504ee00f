// C++ host-facade tests (include/pharmsol_hip.hpp over the C ABI).  Written to read like the reference's own
// tests/examples.  `facade_test cpu` needs no GPU (data model + population compiler through
// pmx_debug_compile); `facade_test gpu` runs predictions on device 0 and checks them against the CPU oracle
// (oracle/pmx_oracle.h — test infrastructure) and against the reference's known values.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/pharmsol_hip.hpp"
#include "../../oracle/pmx_oracle.h"

using namespace pharmsol;
using equation::Analytical;
using equation::ODE;
using equation::Route;

static int failures = 0;
#define CHECK(cond)                                                          \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);            \
      ++failures;                                                            \
    }                                                                        \
  } while (0)

// examples/analytical_readme.rs:7-24
static Analytical readme_model() {
  Analytical m(PMX_K_ONE_COMPARTMENT_WITH_ABSORPTION, 3);
  m.with_nstates(2).with_ndrugs(1).with_nout(1);
  m.with_metadata({"ka", "ke0", "v"}, {"cp"}, {Route::bolus("oral", 0)}, {"wt"});
  m.with_derived_pow(0, /*ke0*/ 1, /*wt*/ 0, 70.0, 0.75);
  m.with_bind({{PMX_SRC_PRIMARY, 0}, {PMX_SRC_DERIVED, 0}});  // structure wants [ka, ke]
  m.with_output(0, /*central*/ 1, /*v*/ 2);
  return m;
}
static Subject readme_subject() {  // examples/analytical_readme.rs:26-33
  return Subject::builder("analytical_readme")
      .bolus(0.0, 500.0, "oral")
      .missing_observation(0.5, "cp")
      .missing_observation(1.0, "cp")
      .missing_observation(2.0, "cp")
      .missing_observation(4.0, "cp")
      .covariate("wt", 0.0, 75.0)
      .build();
}

static void test_builder_and_sort() {
  // builder.rs:369-391 / event.rs:292-304
  Subject s = Subject::builder("s1").infusion(1.0, 1.0, 0, 1.0).bolus(1.0, 1.0, 0).observation(1.0, 0.0, 0)
                  .observation(0.5, 0.0, 0).bolus(0.5, 2.0, 0).repeat(2, 12.0).reset().observation(10.0, 1.0, 0).build();
  CHECK(s.occasions().size() == 2);
  const auto& ev = s.occasions()[0].events;
  CHECK(ev.size() == 7);
  CHECK(ev[0].kind == PMX_EV_OBSERVATION && ev[0].time == 0.5);
  CHECK(ev[1].kind == PMX_EV_BOLUS && ev[1].time == 0.5);
  CHECK(ev[2].kind == PMX_EV_OBSERVATION && ev[3].kind == PMX_EV_BOLUS && ev[4].kind == PMX_EV_INFUSION);
  CHECK(ev[5].time == 12.5 && ev[6].time == 24.5);  // repeat(2, 12.0) of the last bolus
  CHECK(s.occasions()[1].index == 1);
}

static void test_labels_and_parameters() {
  Analytical m = readme_model();
  CHECK(m.resolve_input_label("oral", Route::Bolus) == 0);
  CHECK(m.resolve_output_label("cp") == 0);
  bool threw = false;
  try { m.resolve_input_label("iv", Route::Infusion); } catch (const Error&) { threw = true; }
  CHECK(threw);
  auto p = Parameters::with_model(m, {{"v", 194.0}, {"ka", 1.2}, {"ke0", 0.08}});
  CHECK(p.size() == 3 && p[0] == 1.2 && p[1] == 0.08 && p[2] == 194.0);
  threw = false;
  try { Parameters::with_model(m, {{"ka", 1.2}}); } catch (const Error&) { threw = true; }
  CHECK(threw);
}

static void test_compile_without_gpu() {
  Analytical m = readme_model();
  auto flat = m.flatten({readme_subject()});
  pmx_population_desc d = flat.desc();
  pmx_op_stream_view v{};
  check(pmx_debug_compile(&d, &m.desc(), &v));
  // RESET, BOLUS, PROP(0.5), OBS, PROP(0.5), OBS, PROP(1), OBS, PROP(2), OBS
  CHECK(v.n_ops == 10);
  CHECK((v.op_meta[1] & 0xff) == PMX_OP_BOLUS && v.op_a[1] == 500.0);
  CHECK((v.op_meta[2] & 0xff) == PMX_OP_PROP && v.op_a[2] == 0.5 && v.op_cov[2] == 75.0);
  CHECK((v.op_meta[8] & 0xff) == PMX_OP_PROP && v.op_a[8] == 2.0);
  pmx_debug_free(&v);
}

static void test_gpu_readme() {
  Analytical m = readme_model();
  auto params = Parameters::with_model(m, {{"ka", 1.2}, {"ke0", 0.08}, {"v", 194.0}});
  SubjectPredictions pr = m.estimate_predictions(readme_subject(), params);
  const double want[4] = {1.1363216631314599, 1.7130756583758835, 2.0906323551896495, 1.956103669112038};
  auto got = pr.flat_predictions();
  CHECK(got.size() == 4);
  for (size_t i = 0; i < got.size() && i < 4; ++i) CHECK(std::fabs(got[i] - want[i]) / want[i] < 1e-6);
  auto t = pr.flat_times();
  CHECK(t.size() == 4 && t[0] == 0.5 && t[3] == 4.0);
}

static void test_gpu_two_compartment_matrix_vs_oracle() {
  // examples/analytical_vs_ode.rs subject_iv + a 40-point grid, against the CPU oracle at 1e-6
  Analytical m(PMX_K_TWO_COMPARTMENTS, 4);
  m.with_nstates(2).with_ndrugs(1).with_nout(1).with_output(0, 0, 3);
  m.with_metadata({"ke", "kcp", "kpc", "v"}, {"cp"}, {Route::infusion("iv", 0)});
  Data data;
  for (int s = 0; s < 50; ++s) {
    auto b = Subject::builder(std::to_string(s)).infusion(0.0, 500.0 + s, "iv", 0.5);
    for (double t : {0.5, 1.0, 2.0, 4.0, 8.0, 12.0, 24.0}) b.observation(t, 0.0, "cp");
    data.push_back(b.build());
  }
  const int P = 40;
  std::vector<double> theta;
  for (int p = 0; p < P; ++p) {
    theta.push_back(0.05 + 0.01 * p);
    theta.push_back(0.1 + 0.005 * p);
    theta.push_back(0.08 + 0.003 * p);
    theta.push_back(20.0 + p);
  }
  std::vector<double> pred;
  std::vector<uint8_t> status;
  m.predict_matrix(data, theta, P, 0, &pred, &status);
  auto flat = m.flatten(data);
  pmx_population_desc d = flat.desc();
  std::vector<double> want(pred.size());
  std::vector<uint8_t> wst(status.size());
  CHECK(pmx_oracle_predict(&m.desc(), &d, theta.data(), P, want.data(), P, wst.data(), 1) == PMX_OK);
  double worst = 0.0;
  for (size_t i = 0; i < pred.size(); ++i) worst = std::fmax(worst, std::fabs(pred[i] - want[i]) / std::fmax(std::fabs(want[i]), 1e-12));
  CHECK(pred.size() == 50u * 7u * P);
  CHECK(worst < 1e-6);
  std::printf("two-compartment matrix: max rel err vs oracle %.3e\n", worst);
}

static void test_gpu_ode_dose_conservation() {
  // ode/mod.rs:1336-1347: back-to-back infusions conserve the dose (dx = rateiv[0], y = x[0])
  ODE m(PMX_ODE_ONE_CMT_IV, 1, 0.01);
  m.with_nstates(1).with_ndrugs(1).with_nout(1).with_output(0, 0, -1);
  m.with_metadata({"ke"}, {"cp"}, {Route::infusion("iv", 0)});
  Subject s = Subject::builder("b2b").infusion(0.0, 100.0, "iv", 0.5).infusion(0.5, 100.0, "iv", 0.5)
                  .observation(1.0, 0.0, "cp").build();
  auto pr = m.estimate_predictions(s, Parameters::dense({0.0}));
  CHECK(pr.predictions.size() == 1 && std::fabs(pr.predictions[0].prediction - 200.0) / 200.0 < 1e-4);
}

#define CHECK_CLOSE(a, b, rtol)                                                                      \
  do {                                                                                               \
    const double a_ = (a), b_ = (b);                                                                 \
    if (!(std::fabs(a_ - b_) <= (rtol) * std::fmax(std::fabs(b_), 1e-300))) {                        \
      std::printf("FAIL %s:%d  %s = %.17g, expected %.17g\n", __FILE__, __LINE__, #a, a_, b_);      \
      ++failures;                                                                                    \
    }                                                                                                \
  } while (0)

// user ODE bodies + adaptive solver + fused log-likelihood through the C++ facade, against the oracle / closed forms
static const char* kOneCmtSrc = R"SRC(
PMX_DEVICE void pmx_dynamics(double t, const double* x, const double* p, const double* cov, const double* rateiv,
                             const double* derived, double* dx) { dx[0] = -p[0] * x[0] + rateiv[0]; }
PMX_DEVICE void pmx_outputs(double t, const double* x, const double* p, const double* cov, const double* rateiv,
                            const double* derived, double* y) { y[0] = x[0] / p[1]; }
)SRC";

static void test_custom_source_compiles_without_gpu() {
  // hiprtc needs no device: creation compiles the source; a typo comes back with the compiler's message
  pmx_model_desc d{};
  d.eq_kind = PMX_EQ_ODE;
  d.kernel = PMX_ODE_CUSTOM;
  d.nstates = 1; d.ndrugs = 1; d.nout = 1; d.nparams = 2; d.rk4_h_max = 0.02;
  d.ode_rtol = d.ode_atol = 1e-4;
  for (int i = 0; i < PMX_MAX_STATES; ++i) d.init_param[i] = -1;
  for (int i = 0; i < PMX_MAX_INPUTS; ++i) d.lag_param[i] = d.fa_param[i] = d.bolus_dest[i] = d.infusion_dest[i] = -1;
  pmx_model* m = nullptr;
  CHECK(pmx_model_create_custom(&d, kOneCmtSrc, 0, &m) == PMX_OK);
  pmx_model_destroy(m);
  std::string bad = kOneCmtSrc;
  bad.replace(bad.find("x[0] +"), 6, "q[0] +");
  CHECK(pmx_model_create_custom(&d, bad.c_str(), 0, &m) == PMX_ERR_INVALID_ARGUMENT);
  CHECK(std::string(pmx_last_error()).find("undeclared identifier 'q'") != std::string::npos);
}

static void test_gpu_custom_ode_dopri5_and_loglik() {
  ODE m = ODE::custom(kOneCmtSrc, 1, 2);
  m.with_step(4.0).with_solver(PMX_SOLVER_DOPRI5).with_tolerances(1e-9, 1e-9);
  Subject s = Subject::builder("c").bolus(0.0, 100.0, 0).observation(1.0, 7.5, 0).observation(4.0, 4.0, 0)
                  .missing_observation(8.0, 0).build();
  const double ke = 0.2, v = 10.0;
  auto pred = m.estimate_predictions(s, Parameters::dense({ke, v})).flat_predictions();
  CHECK(pred.size() == 3);
  for (size_t i = 0; i < 3; ++i) {
    const double t = (i == 0 ? 1.0 : (i == 1 ? 4.0 : 8.0));
    CHECK_CLOSE(pred[i], 100.0 / v * std::exp(-ke * t), 1e-7);
  }
  // log-likelihood: additive error, sigma^2 = (0.1 + 0.1 y)^2 + 0.2^2, the missing observation contributes nothing
  std::vector<pmx_error_model> em(1);
  em[0].kind = PMX_EM_ADDITIVE;
  em[0].reserved = 0;
  em[0].c[0] = 0.1; em[0].c[1] = 0.1; em[0].c[2] = 0.0; em[0].c[3] = 0.0;
  em[0].scalar = 0.2;
  std::vector<double> ll;
  std::vector<uint8_t> st;
  m.log_likelihood_matrix({s}, {ke, v, 0.3, 12.0}, 2, em, 0, &ll, &st);
  CHECK(ll.size() == 2 && st[0] == 0 && st[1] == 0);
  auto lnpdf = [](double obs, double p, double sig) {
    return -0.5 * 1.8378770664093453 - std::log(sig) - (obs - p) * (obs - p) / (2.0 * sig * sig);
  };
  for (int k = 0; k < 2; ++k) {
    const double kk = k ? 0.3 : ke, vv = k ? 12.0 : v;
    const double p1 = 100.0 / vv * std::exp(-kk * 1.0), p4 = 100.0 / vv * std::exp(-kk * 4.0);
    const double want = lnpdf(7.5, p1, std::sqrt(std::pow(0.1 + 0.75, 2) + 0.04)) +
                        lnpdf(4.0, p4, std::sqrt(std::pow(0.1 + 0.4, 2) + 0.04));
    CHECK_CLOSE(ll[static_cast<size_t>(k)], want, 1e-6);
  }
}

// ---- round 2: user closures, ParameterOrder, per-subject / batch likelihoods, the resident population
// The covariate model of the reference's own parity test (tests/analytical_macro_lowering.rs:225-260): lag, fa, init and
// the output volume are functions of (theta, t, covariates); same bodies as tests/test_user_analytical.py.
#define PMX_SIG "double t, const double* x, const double* p, const double* cov, const double* rateiv, const double* derived, double* "
static const char* kCovariateSrc =
    "enum { P_ka, P_ke0, P_v, P_tlag, P_f_oral, P_base_gut, P_base_central };\n"
    "enum { D_ke, D_adjusted_v };  enum { COV_wt, COV_renal };  enum { X_gut, X_central };\n"
    "PMX_DEVICE void pmx_derive(" PMX_SIG "d) {\n"
    "  const double wt = cov[COV_wt], renal = cov[COV_renal];\n"
    "  d[D_ke] = p[P_ke0] * pow(wt / 70.0, 0.75) * pow(renal / 90.0, 0.25);\n"
    "  d[D_adjusted_v] = p[P_v] * (wt / 70.0) * (1.0 + 0.001 * (renal - 90.0));\n"
    "}\n"
    "PMX_DEVICE void pmx_route_lag(" PMX_SIG "lag) {\n"
    "  lag[0] = p[P_tlag] * sqrt(cov[COV_wt] / 70.0) * pow(90.0 / cov[COV_renal], 0.1);\n"
    "}\n"
    "PMX_DEVICE void pmx_route_bioavailability(" PMX_SIG "fa) {\n"
    "  fa[0] = fmin(fmax(p[P_f_oral] * pow(cov[COV_renal] / 90.0, 0.1), 0.0), 1.0);\n"
    "}\n"
    "PMX_DEVICE void pmx_init(" PMX_SIG "xi) {\n"
    "  xi[X_gut] = p[P_base_gut] + 0.03 * cov[COV_wt];\n"
    "  xi[X_central] = p[P_base_central] + 0.08 * cov[COV_renal];\n"
    "}\n"
    "PMX_DEVICE void pmx_outputs(" PMX_SIG "y) { y[0] = x[X_central] / derived[D_adjusted_v]; }\n";
static const uint32_t kCovariateFns =
    PMX_FN_DERIVE | PMX_FN_ROUTE_LAG | PMX_FN_ROUTE_BIOAVAILABILITY | PMX_FN_INIT | PMX_FN_OUTPUTS;

static Analytical covariate_model() {
  Analytical m(PMX_K_ONE_COMPARTMENT_WITH_ABSORPTION, 7);
  m.with_nstates(2).with_ndrugs(1).with_nout(1);
  m.with_bind({{PMX_SRC_PRIMARY, 0}, {PMX_SRC_DERIVED, 0}});  // structure order [ka, ke] <- theta ka, derived ke
  m.with_metadata({"ka", "ke0", "v", "tlag", "f_oral", "base_gut", "base_central"}, {"cp"},
                  {Route::bolus("oral", 0), Route::infusion("iv", 1)}, {"wt", "renal"});
  m.with_closures(kCovariateSrc, kCovariateFns, /*n_derived=*/2);
  return m;
}
static Subject covariate_subject() {  // tests/analytical_macro_lowering.rs:35-51
  auto b = Subject::builder("analytical-macro-covariates").bolus(1.0, 100.0, "oral").infusion(6.0, 140.0, "iv", 2.0);
  for (double t : {0.25, 0.75, 1.5, 3.0, 6.5, 7.0, 8.0}) b.missing_observation(t, "cp");
  b.covariate("wt", 0.0, 68.0).covariate("wt", 8.0, 74.0).covariate("renal", 0.0, 95.0).covariate("renal", 8.0, 72.0);
  return b.build();
}

static void test_parameter_order_and_error_model_helpers() {
  ParameterOrder ord({"v", "ke", "ka"}, {"ka", "ke", "v"});
  CHECK(!ord.is_identity());
  auto r = ord.reorder({194.0, 0.08, 1.2, 200.0, 0.09, 1.3});
  CHECK(r.size() == 6 && r[0] == 1.2 && r[1] == 0.08 && r[2] == 194.0 && r[3] == 1.3 && r[5] == 200.0);
  CHECK(ParameterOrder({"a", "b"}, {"a", "b"}).is_identity());
  bool threw = false;
  try { ParameterOrder({"a", "c"}, {"a", "b"}); } catch (const Error&) { threw = true; }
  CHECK(threw);
  threw = false;
  try { ParameterOrder({"a", "a"}, {"a", "b"}); } catch (const Error&) { threw = true; }
  CHECK(threw);
  const pmx_error_model em = AssayErrorModel::proportional(0.02, 0.15, 0.001, 0.0, 1.3);
  CHECK(em.kind == PMX_EM_PROPORTIONAL && em.c[1] == 0.15 && em.scalar == 1.3);
  CHECK(AssayErrorModel::additive(0.1, 0.1, 0, 0, 0.2).kind == PMX_EM_ADDITIVE && AssayErrorModel::none().kind == PMX_EM_NONE);
}

static void test_user_closures_compile_without_gpu() {
  Analytical m = covariate_model();
  pmx_model* h = nullptr;
  CHECK(pmx_model_create_user(&m.desc(), kCovariateSrc, kCovariateFns, &h) == PMX_OK);
  if (h) pmx_model_destroy(h);
  // a closure the source does not define is a link-time hole, reported with the compiler's text
  h = nullptr;
  CHECK(pmx_model_create_user(&m.desc(), kCovariateSrc, kCovariateFns | PMX_FN_SEQ_EQ, &h) == PMX_ERR_INVALID_ARGUMENT);
  CHECK(std::string(pmx_last_error()).find("pmx_seq_eq") != std::string::npos);
}

static void test_gpu_reference_covariate_fixture() {
  // expected: the fixture marched in plain Python from the reference's rules
  // (tests/test_user_analytical.py independent_fixture_predictions, support point :470-483)
  const double want[7] = {0.6899882426482612, 0.6952166346374732, 0.6754383593029405, 2.2271922879272488,
                          2.6237756574341624, 3.4411825895032986, 4.868892669547507};
  Analytical m = covariate_model();
  const auto theta = Parameters::with_model(m, {{"ka", 1.0}, {"ke0", 0.16}, {"v", 32.0}, {"tlag", 0.5}, {"f_oral", 0.8},
                                                {"base_gut", 3.0}, {"base_central", 14.0}});
  auto got = m.estimate_predictions(covariate_subject(), theta).flat_predictions();
  CHECK(got.size() == 7);
  for (size_t i = 0; i < got.size() && i < 7; ++i) CHECK_CLOSE(got[i], want[i], 1e-9);
}

static void test_gpu_batch_and_resident_likelihoods() {
  Analytical m(PMX_K_TWO_COMPARTMENTS, 4);
  m.with_nstates(2).with_ndrugs(1).with_nout(1).with_output(0, 0, 3);
  m.with_metadata({"ke", "kcp", "kpc", "v"}, {"cp"}, {Route::infusion("iv", 0)});
  Data data;
  for (int s = 0; s < 12; ++s) {
    auto b = Subject::builder(std::to_string(s)).infusion(0.0, 500.0 + 10.0 * s, "iv", 0.5);
    for (double t : {0.5, 1.0, 2.0, 4.0, 8.0}) b.observation(t, 8.0 / (1.0 + t) + 0.1 * s, "cp");
    data.push_back(b.build());
  }
  const std::vector<pmx_error_model> em = {AssayErrorModel::additive(0.05, 0.1, 0.0, 0.0, 0.1)};
  const int P = 3;
  const std::vector<double> theta = {0.10, 0.30, 0.20, 50.0, 0.20, 0.10, 0.15, 40.0, 0.05, 0.20, 0.10, 60.0};
  std::vector<double> ll;
  std::vector<uint8_t> st;
  m.log_likelihood_matrix(data, theta, P, em, 0, &ll, &st);
  // per-subject entry == estimate_log_likelihood
  const double one = m.estimate_log_likelihood(data[4], {0.20, 0.10, 0.15, 40.0}, em);
  CHECK_CLOSE(one, ll[4 * P + 1], 1e-12);
  // the resident population gives the same table, call after call
  equation::Equation::Resident pop(m, data);
  CHECK(pop.n_subjects() == 12 && pop.n_observations() == 60 && pop.n_events() == 72);
  std::vector<double> ll2(ll.size());
  std::vector<uint8_t> st2(st.size());
  for (int rep = 0; rep < 2; ++rep) {
    pop.log_likelihood_matrix(theta.data(), P, em, ll2.data(), st2.data());
    for (size_t i = 0; i < ll.size(); ++i) CHECK(ll2[i] == ll[i]);
  }
  // batch: subject s under its own row; the diagonal of a (subject x row) table; a failing row -> -inf, not an error
  std::vector<double> tb;
  for (int s = 0; s < 12; ++s)
    for (int k = 0; k < 4; ++k) tb.push_back(theta[static_cast<size_t>((s % P) * 4 + k)]);
  tb[7 * 4 + 0] = 1.0;   // subject 7: ke = 1, kcp = -0.5, kpc = 1 -> complex eigenvalues (two_compartment_models.rs:20-22)
  tb[7 * 4 + 1] = -0.5;
  tb[7 * 4 + 2] = 1.0;
  std::vector<uint8_t> stb;
  const auto llb = m.log_likelihood_batch(data, tb, em, 0, &stb);
  CHECK(llb.size() == 12 && stb.size() == 12);
  for (int s = 0; s < 12; ++s) {
    if (s == 7) {
      CHECK(std::isinf(llb[7]) && llb[7] < 0 && stb[7] == PMX_PAIR_COMPLEX_ROOTS);
    } else {
      CHECK_CLOSE(llb[static_cast<size_t>(s)], ll[static_cast<size_t>(s * P + s % P)], 1e-12);
      CHECK(stb[static_cast<size_t>(s)] == PMX_PAIR_OK);
    }
  }
  const auto pb = m.predict_batch(data, tb, 0);
  CHECK(pb.size() == 60 && std::isnan(pb[7 * 5]) && std::isfinite(pb[0]));
}

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "cpu";
  try {
    test_builder_and_sort();
    test_labels_and_parameters();
    test_compile_without_gpu();
    test_custom_source_compiles_without_gpu();
    test_parameter_order_and_error_model_helpers();
    test_user_closures_compile_without_gpu();
    if (mode == "gpu") {
      test_gpu_reference_covariate_fixture();
      test_gpu_batch_and_resident_likelihoods();
      test_gpu_readme();
      test_gpu_two_compartment_matrix_vs_oracle();
      test_gpu_ode_dose_conservation();
      test_gpu_custom_ode_dopri5_and_loglik();
    } else {
      // no device: the facade must fail loudly, never fall back to a CPU path
      bool threw = false;
      try {
        Analytical m = readme_model();
        m.estimate_predictions(readme_subject(), Parameters::dense({1.2, 0.08, 194.0}));
      } catch (const Error& e) {
        threw = (e.status == PMX_ERR_NO_DEVICE);
      }
      if (pmx_device_count() == 0) CHECK(threw);
    }
  } catch (const std::exception& e) {
    std::printf("FAIL unexpected exception: %s\n", e.what());
    ++failures;
  }
  std::printf("%s: %d failure(s)\n", mode.c_str(), failures);
  return failures == 0 ? 0 : 1;
}
