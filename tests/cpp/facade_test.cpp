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

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "cpu";
  try {
    test_builder_and_sort();
    test_labels_and_parameters();
    test_compile_without_gpu();
    if (mode == "gpu") {
      test_gpu_readme();
      test_gpu_two_compartment_matrix_vs_oracle();
      test_gpu_ode_dose_conservation();
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
