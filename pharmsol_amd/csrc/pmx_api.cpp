// pmx_api.cpp — the extern "C" boundary of libpmx_hip.so (include/pmx.h).
//
// No CPU compute path exists in this library: if there is no HIP device the
// entry points fail with PMX_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <memory>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pmx.h"
#include "pmx_compile.hpp"
#include "pmx_jit.hpp"
#include <dlfcn.h>
#include "pmx_kernels.hpp"
#include "pmx_structures.hpp"  // kernel_nparams()

namespace {

thread_local std::string g_err;
thread_local const char* g_kernel_name = "";

int32_t fail(int32_t code, const std::string& msg) {
  g_err = msg;
  return code;
}
int32_t create_user_ode(const pmx_model_desc* d, const char* source, uint32_t fns, pmx_model** out);  // (after check_user_ode)

// Developer switches (INTEGRATION.md "environment switches").  Read ONCE, at the first call that needs them: a
// std::getenv per launch is measurable on the 11 us C2 pass.  pmx_debug_reload_env() re-reads them (tuning scripts
// and tests that flip a switch inside one process).
struct Tunables {
  bool disable_ladder = false, disable_classing = false, ll_old = false;
  int32_t steps_per_trip = 0, grid_min_p = 0, min_class = 0, cpb = 0;
  int32_t spread = -1, loose = -1;  // -1 = library default
  int32_t prop_slots = -1, dyn_tile = 0;
  void load() {
    auto flag = [](const char* n) {
      const char* e = std::getenv(n);
      return e && e[0] && e[0] != '0';
    };
    auto num = [](const char* n) {
      const char* e = std::getenv(n);
      const int v = e ? std::atoi(e) : 0;
      return v > 0 ? v : 0;
    };
    auto tri = [](const char* n) {
      const char* e = std::getenv(n);
      return e ? ((e[0] && e[0] != '0') ? 1 : 0) : -1;
    };
    disable_ladder = flag("PMX_DISABLE_LADDER");
    disable_classing = flag("PMX_DISABLE_CLASSING");
    ll_old = flag("PMX_TUNE_LL_OLD");
    steps_per_trip = num("PMX_TUNE_STEPS_PER_TRIP");
    grid_min_p = num("PMX_TUNE_GRID_MIN_P");
    min_class = num("PMX_TUNE_MIN_CLASS");
    cpb = num("PMX_TUNE_CPB");
    {
      const char* e = std::getenv("PMX_TUNE_PROP_SLOTS");
      prop_slots = e ? std::atoi(e) : -1;
    }
    dyn_tile = num("PMX_TUNE_DYN_TILE");
    spread = tri("PMX_TUNE_SPREAD");
    loose = tri("PMX_TUNE_LOOSE");
  }
};
std::mutex g_tun_mu;
Tunables g_tun;
bool g_tun_loaded = false;
Tunables tunables() {
  std::lock_guard<std::mutex> lock(g_tun_mu);
  if (!g_tun_loaded) {
    g_tun.load();
    g_tun_loaded = true;
  }
  return g_tun;
}

#define PMX_HIP(call)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(e_ == hipErrorOutOfMemory ? PMX_ERR_OUT_OF_MEMORY : PMX_ERR_HIP,                 \
                  std::string(#call) + ": " + hipGetErrorString(e_));                              \
  } while (0)

// Restores the caller's current device on scope exit.
struct DeviceGuard {
  int prev = -1;
  bool active = false;
  hipError_t enter(int dev) {
    hipError_t e = hipGetDevice(&prev);
    if (e != hipSuccess) return e;
    if (prev != dev) {
      e = hipSetDevice(dev);
      active = (e == hipSuccess);
    }
    return e;
  }
  ~DeviceGuard() {
    if (active) (void)hipSetDevice(prev);
  }
};

struct DeviceStream {
  pmx::CompileKey key;
  pmx::DevOps dev{};
  pmx::DevClassPlan cls{};
  pmx::DevSteps steps{};  // fused step programs of the lean generic walker (analytical streams without lag / covariates)
  int64_t n_classed_subjects = 0;
  // host copies for the log-likelihood's per-chunk observation blocks
  std::vector<int64_t> h_chunk_row;      // [n_chunks*G]
  std::vector<int32_t> h_chunk_nobs;     // [n_chunks] observations per member of the chunk's class
  std::vector<int32_t> h_chunk_n;        // [n_chunks] live members
  // Sigma tables of the log-likelihood, per set of error models.  They are filled ON THE DEVICE
  // (pmx_kernels.hip pmx_ll_prepare_*), stream-ordered before the kernel that reads them: an optimiser that changes
  // gamma / lambda every call pays two ~10 us kernels, not a host pass over every observation plus a 40 MB upload.
  // A small LRU of slots; uses of one slot are chained through its event so that a slot is never rewritten while a
  // kernel on another stream still reads it.
  struct LLCache {
    std::vector<pmx_error_model> em;
    double* d_obs = nullptr;   // [n_obs][4]
    double* d_cobs = nullptr;  // classed blocks
    int32_t* d_err = nullptr;  // invalid-sigma counter of the last fill
    hipEvent_t ev = nullptr;   // last use (fill or read) of this slot
    int64_t stamp = 0;         // LRU
    int32_t host_users = 0;    // host threads between "picked" and "launched"
  };
  std::deque<LLCache> ll_cache;  // (deque: slots handed out by pointer must survive later push_backs)
  int64_t ll_stamp = 0;
  const int32_t* d_chunk_nobs = nullptr;     // [n_chunks]
  const int64_t* d_chunk_obs_off = nullptr;  // [n_chunks] offsets into a slot's cobs
  int64_t cobs_size = 0;
  std::vector<void*> allocs;
  int32_t max_input_used = -1;
  int64_t n_ops = 0, n_prop = 0;
  int64_t max_lagb_per_list = 0;
  int32_t prop_cache_used = 0;        // LDS slots the stream's propagator-cache codes use
  bool no_rates = false;              // no PROP of the stream has an active infusion
  bool eig_reuse = false;             // some PROP repeats the previous built segment's covariate factor row (bit 27)
  double prop_reuse_fraction = 0.0;   // share of PROP ops that take a kept propagator
  ~DeviceStream() {
    for (void* p : allocs) (void)hipFree(p);
    for (auto& c : ll_cache)
      if (c.ev) (void)hipEventDestroy(c.ev);
  }
};

template <class T>
int32_t upload(const std::vector<T>& v, const T** out, std::vector<void*>* allocs) {
  *out = nullptr;
  if (v.empty()) return PMX_OK;
  void* p = nullptr;
  PMX_HIP(hipMalloc(&p, v.size() * sizeof(T)));
  allocs->push_back(p);
  PMX_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = static_cast<const T*>(p);
  return PMX_OK;
}

int ode_nstates(int model) {
  static const int n[PMX_ODE_MODEL_COUNT] = {1, 2, 2, 3, 3, 4, 1};
  return (model >= 0 && model < PMX_ODE_MODEL_COUNT) ? n[model] : -1;
}
int ode_nparams(int model) {
  static const int n[PMX_ODE_MODEL_COUNT] = {1, 2, 3, 4, 5, 6, 3};
  return (model >= 0 && model < PMX_ODE_MODEL_COUNT) ? n[model] : -1;
}

}  // namespace

namespace pmx {
// the calling thread's error text, for the other translation units of the C ABI (pmx_shard.cpp)
int32_t set_error(int32_t code, const std::string& msg) { return fail(code, msg); }
}  // namespace pmx

// What the HOST-pointer entry points (pmx_predict, pmx_predict_batch, pmx_loglik, pmx_loglik_batch) keep between
// calls, per population: device buffers for theta / output / status (grown, never shrunk), a private stream pair and
// two pinned bounce buffers.  An NPAG loop calls these entry points thousands of times; allocating, page-locking and
// freeing per call cost ~700x the kernel (profiles/r01/pcie_inclusive.txt).  Calls on one population take turns.
struct HostWorkspace {
  std::mutex mu;
  hipStream_t compute = nullptr, copy = nullptr;
  void* d_theta = nullptr;
  void* d_out = nullptr;
  void* d_status = nullptr;
  int32_t* d_flag = nullptr;  // "any pair failed" (pmx_status_any)
  size_t theta_cap = 0, out_cap = 0, status_cap = 0;
  static constexpr size_t kBounce = 32u << 20;
  void* bounce[2] = {nullptr, nullptr};
  hipEvent_t bev[2] = {nullptr, nullptr};
  int32_t* h_flag = nullptr;  // pinned
  ~HostWorkspace() {
    if (d_theta) (void)hipFree(d_theta);
    if (d_out) (void)hipFree(d_out);
    if (d_status) (void)hipFree(d_status);
    if (d_flag) (void)hipFree(d_flag);
    for (int i = 0; i < 2; ++i) {
      if (bounce[i]) (void)hipHostFree(bounce[i]);
      if (bev[i]) (void)hipEventDestroy(bev[i]);
    }
    if (h_flag) (void)hipHostFree(h_flag);
    if (compute) (void)hipStreamDestroy(compute);
    if (copy) (void)hipStreamDestroy(copy);
  }
};

struct pmx_population {
  int device = 0;
  pmx::HostPopulation hp;
  std::mutex mu;
  std::unique_ptr<HostWorkspace> ws;  // created by the first host-pointer call
  // what the log-likelihood tables are computed from, uploaded at the first pmx_loglik* call
  bool ll_ready = false;
  const double* d_obs_y = nullptr;
  const int32_t* d_obs_outeq = nullptr;
  const double* d_obs_poly = nullptr;
  const int8_t* d_obs_cens = nullptr;
  uint32_t valued_outeq_mask = 0;  // bit q: some observation on output q carries a value
  bool any_censored = false;
  std::vector<void*> ll_allocs;
  ~pmx_population() {
    for (void* p : ll_allocs) (void)hipFree(p);
  }
  std::vector<std::unique_ptr<DeviceStream>> streams;  // one per model flavour, built lazily
};

struct pmx_model {
  pmx_model_desc d;
  bool dyn = false;  // kernel parameters depend on covariates
  bool has_init = false;
  // custom (hiprtc) models: the code object and its per-device modules
  bool custom = false;
  uint32_t user_fns = 0;  // PMX_FN_* the user's source defines (pmx_model_create_user)
  bool user_lag = false, user_eq = false;  // user model: any lag closure (user's or descriptor's) / own propagator
  bool user_ode = false;                   // ODE model on the general walker (pmx_ode_user.hpp): lag / fa / derive closures, bolus[]
  std::vector<char> jit_code;
  pmx::JitSpec jit_spec;  // what jit_code was compiled from (the big-lists build below is made from it on demand)
  mutable std::vector<char> jit_code_big;  // closure walkers: the PMX_USER_BIG_LISTS build, compiled at the first launch on a
                                           // population with more than 64 lagged boluses in one occasion (pmx_userlag.hpp)
  mutable std::mutex jit_mu;
  mutable std::map<int, pmx::JitModule> jit_modules;
  mutable std::map<int, pmx::JitModule> jit_modules_big;
  ~pmx_model() {
    for (auto& kv : jit_modules) pmx::jit_unload(&kv.second);
    for (auto& kv : jit_modules_big) pmx::jit_unload(&kv.second);
  }
};

extern "C" {

int32_t pmx_abi_version(void) { return PMX_ABI_VERSION; }
int64_t pmx_sizeof_model_desc(void) { return static_cast<int64_t>(sizeof(pmx_model_desc)); }
int64_t pmx_sizeof_population_desc(void) { return static_cast<int64_t>(sizeof(pmx_population_desc)); }
int64_t pmx_sizeof_struct(const char* name) {
  if (!name) return -1;
#define PMX_SZ(T) \
  if (std::strcmp(name, #T) == 0) return static_cast<int64_t>(sizeof(T));
  PMX_SZ(pmx_population_desc)
  PMX_SZ(pmx_factor)
  PMX_SZ(pmx_derived)
  PMX_SZ(pmx_bind)
  PMX_SZ(pmx_out)
  PMX_SZ(pmx_model_desc)
  PMX_SZ(pmx_error_model)
  PMX_SZ(pmx_op_stream_view)
#undef PMX_SZ
  return -1;
}

int32_t pmx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* pmx_last_error(void) { return g_err.c_str(); }

void pmx_debug_reload_env(void) {
  std::lock_guard<std::mutex> lock(g_tun_mu);
  g_tun.load();
  g_tun_loaded = true;
}
const char* pmx_last_kernel_name(void) { return g_kernel_name; }

int32_t pmx_population_create(const pmx_population_desc* desc, int32_t device, pmx_population** out) {
  g_err.clear();
  if (!out) return fail(PMX_ERR_INVALID_ARGUMENT, "out is null");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(PMX_ERR_NO_DEVICE, "no HIP device visible (libpmx_hip has no CPU path)");
  if (device < 0 || device >= n) return fail(PMX_ERR_INVALID_ARGUMENT, "device ordinal out of range");
  auto pop = std::make_unique<pmx_population>();
  pop->device = device;
  std::string err;
  int32_t rc = pmx::build_host_population(desc, &pop->hp, &err);
  if (rc != PMX_OK) return fail(rc, err);
  *out = pop.release();
  return PMX_OK;
}

int32_t pmx_population_create_shard(const pmx_population_desc* desc, int64_t subject_begin, int64_t subject_end,
                                    int32_t device, pmx_population** out) {
  g_err.clear();
  if (!out) return fail(PMX_ERR_INVALID_ARGUMENT, "out is null");
  *out = nullptr;
  if (!desc || !desc->subj_occ_off || !desc->occ_ev_off) return fail(PMX_ERR_INVALID_ARGUMENT, "null population descriptor");
  if (subject_begin < 0 || subject_end < subject_begin || subject_end > desc->n_subjects)
    return fail(PMX_ERR_INVALID_ARGUMENT, "subject range out of bounds");
  // the descriptor of subjects [begin, end): the caller's arrays addressed in place, the three CSR offset arrays re-based
  const int64_t s0 = subject_begin, s1 = subject_end;
  const int64_t o0 = desc->subj_occ_off[s0], o1 = desc->subj_occ_off[s1];
  const int64_t e0 = desc->occ_ev_off[o0], e1 = desc->occ_ev_off[o1];
  std::vector<int64_t> subj_occ(static_cast<size_t>(s1 - s0) + 1), occ_ev(static_cast<size_t>(o1 - o0) + 1), knot_off;
  for (int64_t s = s0; s <= s1; ++s) subj_occ[static_cast<size_t>(s - s0)] = desc->subj_occ_off[s] - o0;
  for (int64_t o = o0; o <= o1; ++o) occ_ev[static_cast<size_t>(o - o0)] = desc->occ_ev_off[o] - e0;
  pmx_population_desc d = *desc;
  d.n_subjects = s1 - s0;
  d.n_occasions = o1 - o0;
  d.n_events = e1 - e0;
  d.subj_occ_off = subj_occ.data();
  d.occ_ev_off = occ_ev.data();
  if (desc->occ_index) d.occ_index = desc->occ_index + o0;
  if (desc->ev_time) d.ev_time = desc->ev_time + e0;
  if (desc->ev_value) d.ev_value = desc->ev_value + e0;
  if (desc->ev_duration) d.ev_duration = desc->ev_duration + e0;
  if (desc->ev_kind) d.ev_kind = desc->ev_kind + e0;
  if (desc->ev_io) d.ev_io = desc->ev_io + e0;
  if (desc->ev_errorpoly) d.ev_errorpoly = desc->ev_errorpoly + 4 * e0;
  if (desc->ev_censor) d.ev_censor = desc->ev_censor + e0;
  const int32_t nc = desc->n_covariates;
  if (nc > 0) {
    if (!desc->cov_knot_off || !desc->cov_knot_time || !desc->cov_knot_value)
      return fail(PMX_ERR_INVALID_ARGUMENT, "covariate arrays missing");
    const int64_t c0 = o0 * nc, c1 = o1 * nc, k0 = desc->cov_knot_off[c0];
    knot_off.resize(static_cast<size_t>(c1 - c0) + 1);
    for (int64_t c = c0; c <= c1; ++c) knot_off[static_cast<size_t>(c - c0)] = desc->cov_knot_off[c] - k0;
    d.cov_knot_off = knot_off.data();
    d.cov_knot_time = desc->cov_knot_time + k0;
    d.cov_knot_value = desc->cov_knot_value + k0;
    if (desc->cov_fixed) d.cov_fixed = desc->cov_fixed + c0;
  }
  return pmx_population_create(&d, device, out);
}

void pmx_population_destroy(pmx_population* pop) {
  if (!pop) return;
  {
    DeviceGuard g;
    (void)g.enter(pop->device);
    pop->streams.clear();
    pop->ws.reset();
  }
  delete pop;
}

int64_t pmx_population_n_subjects(const pmx_population* pop) { return pop ? pop->hp.n_subjects : -1; }
int64_t pmx_population_n_observations(const pmx_population* pop) { return pop ? pop->hp.n_obs : -1; }
int64_t pmx_population_n_events(const pmx_population* pop) { return pop ? pop->hp.n_events : -1; }
int32_t pmx_population_device(const pmx_population* pop) { return pop ? pop->device : -1; }

int32_t pmx_population_observation_offsets(const pmx_population* pop, int64_t* obs_off) {
  if (!pop || !obs_off) return fail(PMX_ERR_INVALID_ARGUMENT, "null argument");
  std::memcpy(obs_off, pop->hp.subj_obs_off.data(), sizeof(int64_t) * pop->hp.subj_obs_off.size());
  return PMX_OK;
}

int32_t pmx_population_observation_info(const pmx_population* pop, double* time, int32_t* outeq, int64_t* subject) {
  if (!pop) return fail(PMX_ERR_INVALID_ARGUMENT, "null population");
  const auto& hp = pop->hp;
  if (time) std::memcpy(time, hp.obs_time.data(), sizeof(double) * hp.obs_time.size());
  if (outeq) std::memcpy(outeq, hp.obs_outeq.data(), sizeof(int32_t) * hp.obs_outeq.size());
  if (subject) std::memcpy(subject, hp.obs_subject.data(), sizeof(int64_t) * hp.obs_subject.size());
  return PMX_OK;
}

int32_t pmx_model_create(const pmx_model_desc* d, pmx_model** out) {
  g_err.clear();
  if (!d || !out) return fail(PMX_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  if (d->eq_kind == PMX_EQ_ODE && d->kernel == PMX_ODE_CUSTOM)
    return fail(PMX_ERR_INVALID_ARGUMENT, "PMX_ODE_CUSTOM models are created with pmx_model_create_custom");
  if (d->eq_kind == PMX_EQ_ANALYTICAL && d->kernel == PMX_K_CUSTOM)
    return fail(PMX_ERR_INVALID_ARGUMENT, "PMX_K_CUSTOM models are created with pmx_model_create_user");
  if (d->nstates < 1 || d->nstates > PMX_MAX_STATES) return fail(PMX_ERR_INVALID_ARGUMENT, "nstates out of range");
  if (d->ndrugs < 0 || d->ndrugs > PMX_MAX_INPUTS) return fail(PMX_ERR_INVALID_ARGUMENT, "ndrugs out of range");
  if (d->nout < 1 || d->nout > PMX_MAX_OUT) return fail(PMX_ERR_INVALID_ARGUMENT, "nout out of range");
  if (d->nparams < 0 || d->nparams > PMX_MAX_PARAMS) return fail(PMX_ERR_INVALID_ARGUMENT, "nparams out of range");
  if (d->n_covariates < 0 || d->n_covariates > PMX_MAX_COVARIATES)
    return fail(PMX_ERR_INVALID_ARGUMENT, "n_covariates out of range");
  if (d->n_derived < 0 || d->n_derived > PMX_MAX_DERIVED) return fail(PMX_ERR_INVALID_ARGUMENT, "n_derived out of range");
  for (int i = 0; i < d->n_derived; ++i) {
    const pmx_derived& dd = d->derived[i];
    if (dd.src_param < 0 || dd.src_param >= d->nparams) return fail(PMX_ERR_INVALID_ARGUMENT, "derived.src_param out of range");
    if (dd.n_factors < 0 || dd.n_factors > PMX_MAX_FACTORS) return fail(PMX_ERR_INVALID_ARGUMENT, "derived.n_factors out of range");
    for (int k = 0; k < dd.n_factors; ++k)
      if (dd.f[k].op != PMX_F_NONE && (dd.f[k].cov < 0 || dd.f[k].cov >= d->n_covariates))
        return fail(PMX_ERR_INVALID_ARGUMENT, "derived factor covariate out of range");
  }
  auto m = std::make_unique<pmx_model>();
  m->d = *d;
  const int pm = d->pmetrics_indexing ? 1 : 0;
  bool to_user_walker = false, ode_many_lags = false;
  if (d->eq_kind == PMX_EQ_ANALYTICAL) {
    if (d->kernel < 0 || d->kernel >= PMX_K_ANALYTICAL_COUNT) return fail(PMX_ERR_INVALID_ARGUMENT, "unknown analytical kernel");
    static const int kNS[12] = {1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4};  // AnalyticalKernel::state_count analysis.rs:259-270
    if (d->nstates < kNS[d->kernel] + pm) return fail(PMX_ERR_INVALID_ARGUMENT, "model has fewer states than its structure");
    const int np = pmx::kernel_nparams(d->kernel);
    if (d->n_bind == 0 && d->nparams < np) return fail(PMX_ERR_INVALID_ARGUMENT, "too few parameters for the structure");
    if (d->n_bind != 0 && d->n_bind != np) return fail(PMX_ERR_INVALID_ARGUMENT, "n_bind must equal the structure's parameter count");
    for (int j = 0; j < d->n_bind; ++j) {
      const pmx_bind& b = d->bind[j];
      if (b.src == PMX_SRC_PRIMARY) {
        if (b.index < 0 || b.index >= d->nparams) return fail(PMX_ERR_INVALID_ARGUMENT, "bind index out of range");
      } else if (b.src == PMX_SRC_DERIVED) {
        if (b.index < 0 || b.index >= d->n_derived) return fail(PMX_ERR_INVALID_ARGUMENT, "bind derived index out of range");
        if (d->derived[b.index].n_factors > 0) m->dyn = true;
      } else
        return fail(PMX_ERR_INVALID_ARGUMENT, "bind.src must be PRIMARY or DERIVED");
    }
    int n_lag = 0;
    for (int i = 0; i < PMX_MAX_INPUTS; ++i) {
      if (d->lag_param[i] >= d->nparams || d->fa_param[i] >= d->nparams)
        return fail(PMX_ERR_INVALID_ARGUMENT, "lag_param / fa_param out of range");
      if (d->lag_param[i] >= 0) ++n_lag;
    }
    // Descriptor forms the library's own kernels do not take - lag time together with covariate-derived rate constants
    // or pm_* indexing, more than four lagged inputs - run on the user-closure walker instead (pmx_analytical.hpp): the
    // derived values are written out as source, every other closure is generated from the descriptor as for any user model.
    to_user_walker = n_lag > pmx::kMaxLagSlots || (n_lag > 0 && (m->dyn || pm));
  } else if (d->eq_kind == PMX_EQ_ODE) {
    if (ode_nstates(d->kernel) < 0) return fail(PMX_ERR_INVALID_ARGUMENT, "unknown ODE model");
    if (d->nstates < ode_nstates(d->kernel)) return fail(PMX_ERR_INVALID_ARGUMENT, "model has fewer states than its diffeq");
    if (d->nparams < ode_nparams(d->kernel)) return fail(PMX_ERR_INVALID_ARGUMENT, "too few parameters for the diffeq");
    if (!(d->rk4_h_max > 0.0)) return fail(PMX_ERR_INVALID_ARGUMENT, "rk4_h_max must be > 0");
    if (d->ode_solver != PMX_SOLVER_RK4 && d->ode_solver != PMX_SOLVER_DOPRI5 && d->ode_solver != PMX_SOLVER_ROS2)
      return fail(PMX_ERR_INVALID_ARGUMENT, "unknown ode_solver");
    if (d->ode_solver != PMX_SOLVER_RK4 && !(d->ode_rtol > 0.0 && d->ode_atol > 0.0))
      return fail(PMX_ERR_INVALID_ARGUMENT, "the adaptive solver needs ode_rtol > 0 and ode_atol > 0");
    if (pm) return fail(PMX_ERR_INVALID_ARGUMENT, "pm_* indexing is a wrapper of the analytical structures (analytical/mod.rs:62-90): it does not apply to ODE models");
    if (d->n_bind != 0 && d->n_bind != ode_nparams(d->kernel))
      return fail(PMX_ERR_INVALID_ARGUMENT, "n_bind must equal the diffeq's parameter count");
    for (int j = 0; j < d->n_bind; ++j) {
      const pmx_bind& b = d->bind[j];
      if (b.src == PMX_SRC_PRIMARY ? (b.index < 0 || b.index >= d->nparams)
                                   : (b.src != PMX_SRC_DERIVED || b.index < 0 || b.index >= d->n_derived))
        return fail(PMX_ERR_INVALID_ARGUMENT, "bind entry out of range");
    }
    {
      int n_lag = 0;
      for (int i = 0; i < PMX_MAX_INPUTS; ++i) n_lag += d->lag_param[i] >= 0;
      ode_many_lags = n_lag > pmx::kMaxLagSlots;  // (the state-machine kernels keep four lag cursors: the general ODE walker takes over)
    }
    for (int i = 0; i < PMX_MAX_INPUTS; ++i) {
      if (d->lag_param[i] >= d->nparams || d->fa_param[i] >= d->nparams)
        return fail(PMX_ERR_INVALID_ARGUMENT, "lag_param / fa_param out of range");
      if (d->bolus_dest[i] >= d->nstates || d->infusion_dest[i] >= d->nstates)
        return fail(PMX_ERR_INVALID_ARGUMENT, "route destination out of range");
    }
  } else
    return fail(PMX_ERR_INVALID_ARGUMENT, "unknown eq_kind");
  for (int o = 0; o < d->nout; ++o) {
    const pmx_out& oo = d->out[o];
    if (oo.state < 0 || oo.state >= d->nstates) return fail(PMX_ERR_INVALID_ARGUMENT, "out.state out of range");
    if (oo.vol_src == PMX_SRC_PRIMARY && (oo.vol_index < 0 || oo.vol_index >= d->nparams))
      return fail(PMX_ERR_INVALID_ARGUMENT, "out.vol_index out of range");
    if (oo.vol_src == PMX_SRC_DERIVED && (oo.vol_index < 0 || oo.vol_index >= d->n_derived))
      return fail(PMX_ERR_INVALID_ARGUMENT, "out.vol_index (derived) out of range");
  }
  for (int i = 0; i < PMX_MAX_STATES; ++i) {
    if (d->init_param[i] >= d->nparams) return fail(PMX_ERR_INVALID_ARGUMENT, "init_param out of range");
    if (d->init_param[i] >= 0 && i < d->nstates) m->has_init = true;
  }
  if (to_user_walker) {
    pmx::JitSpec sp;
    sp.analytical = true;
    sp.fns = d->n_derived > 0 ? static_cast<uint32_t>(PMX_FN_DERIVE) : 0u;
    sp.desc = *d;
    sp.source = d->n_derived > 0 ? pmx::analytical_descriptor_source(*d) : std::string();
    std::string log;
    m->jit_spec = sp;
    if (!pmx::jit_compile(sp, &m->jit_code, &log))
      return fail(PMX_ERR_HIP, "hiprtc could not compile the generated closures:\n" + log);
    m->custom = true;
    m->dyn = false;
    m->user_fns = sp.fns;
    m->user_lag = true;
    m->has_init = true;  // (RESET ops always carry the occasion-index flag; the policy's init may be empty)
  }
  if (ode_many_lags) {  // built-in diffeq body, more than four lagged inputs: body written out as source, the general ODE walker
    pmx_model_desc dd = *d;
    dd.kernel = PMX_ODE_CUSTOM;
    dd.n_derived = 0;
    dd.n_bind = 0;
    const std::string src = pmx::ode_descriptor_source(*d);
    return create_user_ode(&dd, src.c_str(), PMX_FN_DYNAMICS | PMX_FN_OUTPUTS | PMX_FN_INIT, out);
  }
  if (d->eq_kind == PMX_EQ_ODE && (d->n_derived > 0 || d->n_bind > 0)) {
    // covariate-derived parameters of a built-in diffeq body (expand/ode.rs:126-185): the body is written out as source
    // and takes the run-time-compiled path, where covariates are looked up on the device at every stage time
    pmx_model_desc dd = *d;
    dd.kernel = PMX_ODE_CUSTOM;
    dd.n_derived = 0;
    dd.n_bind = 0;
    pmx::JitSpec sp;
    sp.nstates = d->nstates;
    sp.nparams = d->nparams;
    sp.nout = d->nout;
    sp.ninputs = d->ndrugs > 0 ? d->ndrugs : 1;
    sp.has_init = m->has_init;
    sp.ncov = d->n_covariates;
    sp.source = pmx::ode_descriptor_source(*d);
    std::string log;
    m->jit_spec = sp;
    if (!pmx::jit_compile(sp, &m->jit_code, &log))
      return fail(PMX_ERR_HIP, "hiprtc could not compile the generated diffeq body:\n" + log);
    m->d = dd;
    m->custom = true;
  }
  *out = m.release();
  return PMX_OK;
}

}  // extern "C"

namespace {
int32_t check_custom_desc(const pmx_model_desc* d, const char* source) {
  if (!d || !source) return fail(PMX_ERR_INVALID_ARGUMENT, "null argument");
  if (d->eq_kind != PMX_EQ_ODE || d->kernel != PMX_ODE_CUSTOM)
    return fail(PMX_ERR_INVALID_ARGUMENT, "custom models need eq_kind = PMX_EQ_ODE and kernel = PMX_ODE_CUSTOM");
  if (d->nstates < 1 || d->nstates > PMX_MAX_STATES) return fail(PMX_ERR_INVALID_ARGUMENT, "nstates out of range");
  if (d->ndrugs < 0 || d->ndrugs > PMX_MAX_INPUTS) return fail(PMX_ERR_INVALID_ARGUMENT, "ndrugs out of range");
  if (d->nout < 1 || d->nout > PMX_MAX_OUT) return fail(PMX_ERR_INVALID_ARGUMENT, "nout out of range");
  if (d->nparams < 1 || d->nparams > PMX_MAX_PARAMS) return fail(PMX_ERR_INVALID_ARGUMENT, "nparams out of range");
  if (!(d->rk4_h_max > 0.0)) return fail(PMX_ERR_INVALID_ARGUMENT, "rk4_h_max must be > 0");
  if (d->ode_solver != PMX_SOLVER_RK4 && d->ode_solver != PMX_SOLVER_DOPRI5 && d->ode_solver != PMX_SOLVER_ROS2)
    return fail(PMX_ERR_INVALID_ARGUMENT, "unknown ode_solver");
  if (d->ode_solver != PMX_SOLVER_RK4 && !(d->ode_rtol > 0.0 && d->ode_atol > 0.0))
    return fail(PMX_ERR_INVALID_ARGUMENT, "the adaptive solver needs ode_rtol > 0 and ode_atol > 0");
  if (d->n_covariates < 0 || d->n_covariates > PMX_MAX_COVARIATES)
    return fail(PMX_ERR_INVALID_ARGUMENT, "n_covariates out of range");
  if (d->n_derived != 0 || d->n_bind != 0 || d->pmetrics_indexing)
    return fail(PMX_ERR_INVALID_ARGUMENT, "derived-parameter descriptors / pm indexing do not apply to custom ODE bodies (compute them in the body)");
  int n_lag = 0;
  for (int i = 0; i < PMX_MAX_INPUTS; ++i) {
    if (d->lag_param[i] >= d->nparams || d->fa_param[i] >= d->nparams)
      return fail(PMX_ERR_INVALID_ARGUMENT, "lag_param / fa_param out of range");
    if (d->bolus_dest[i] >= d->nstates) return fail(PMX_ERR_INVALID_ARGUMENT, "route destination out of range");
    n_lag += d->lag_param[i] >= 0;
  }
  (void)n_lag;  // (more than four lagged inputs: pmx_model_create_custom hands the model to the general ODE walker)
  return PMX_OK;
}
pmx::JitSpec spec_of(const pmx_model_desc* d, const char* source, int32_t has_init) {
  pmx::JitSpec sp;
  sp.nstates = d->nstates;
  sp.nparams = d->nparams;
  sp.nout = d->nout;
  sp.ninputs = d->ndrugs > 0 ? d->ndrugs : 1;
  sp.has_init = has_init != 0;
  sp.ncov = d->n_covariates;
  sp.source = source;
  return sp;
}
}  // namespace

extern "C" {

int32_t pmx_model_create_custom(const pmx_model_desc* d, const char* source, int32_t has_init, pmx_model** out) {
  g_err.clear();
  if (!out) return fail(PMX_ERR_INVALID_ARGUMENT, "out is null");
  *out = nullptr;
  const int32_t rc = check_custom_desc(d, source);
  if (rc != PMX_OK) return rc;
  {
    int n_lag = 0;
    for (int i = 0; i < PMX_MAX_INPUTS; ++i) n_lag += d->lag_param[i] >= 0;
    if (n_lag > pmx::kMaxLagSlots)  // the state-machine kernels keep four lag cursors: the general ODE walker sorts any number
      return create_user_ode(d, source, PMX_FN_DYNAMICS | PMX_FN_OUTPUTS | (has_init ? PMX_FN_INIT : 0u), out);
  }
  auto m = std::make_unique<pmx_model>();
  m->d = *d;
  m->custom = true;
  m->has_init = has_init != 0;
  std::string log;
  m->jit_spec = spec_of(d, source, has_init);
  if (!pmx::jit_compile(m->jit_spec, &m->jit_code, &log))
    return fail(PMX_ERR_INVALID_ARGUMENT, "hiprtc could not compile the model source:\n" + log);
  *out = m.release();
  return PMX_OK;
}

int32_t pmx_debug_jit_source(const pmx_model_desc* d, const char* source, int32_t has_init, char** out_text) {
  g_err.clear();
  if (!out_text) return fail(PMX_ERR_INVALID_ARGUMENT, "out_text is null");
  *out_text = nullptr;
  const int32_t rc = check_custom_desc(d, source);
  if (rc != PMX_OK) return rc;
  const std::string tu = pmx::jit_translation_unit(spec_of(d, source, has_init));
  char* buf = static_cast<char*>(std::malloc(tu.size() + 1));
  if (!buf) return fail(PMX_ERR_OUT_OF_MEMORY, "malloc");
  std::memcpy(buf, tu.c_str(), tu.size() + 1);
  *out_text = buf;
  return PMX_OK;
}

void pmx_free_text(char* text) { std::free(text); }

}  // extern "C"

namespace {
// Analytical model with user closures: what of the descriptor must hold (pmx.h "user closures for either back-end")
int32_t check_user_analytical(const pmx_model_desc* d, const char* source, uint32_t fns) {
  if (!d || !source) return fail(PMX_ERR_INVALID_ARGUMENT, "null argument");
  if (d->nstates < 1 || d->nstates > PMX_MAX_STATES) return fail(PMX_ERR_INVALID_ARGUMENT, "nstates out of range");
  if (d->ndrugs < 1 || d->ndrugs > PMX_MAX_INPUTS) return fail(PMX_ERR_INVALID_ARGUMENT, "ndrugs out of range (1..8 for a model with user closures)");
  if (d->nout < 1 || d->nout > PMX_MAX_OUT) return fail(PMX_ERR_INVALID_ARGUMENT, "nout out of range");
  if (d->nparams < 1 || d->nparams > PMX_MAX_PARAMS) return fail(PMX_ERR_INVALID_ARGUMENT, "nparams out of range");
  if (d->n_covariates < 0 || d->n_covariates > PMX_MAX_COVARIATES) return fail(PMX_ERR_INVALID_ARGUMENT, "n_covariates out of range");
  if (d->n_derived < 0 || d->n_derived > PMX_MAX_USER_DERIVED) return fail(PMX_ERR_INVALID_ARGUMENT, "n_derived out of range");
  if (d->n_derived > 0 && !(fns & PMX_FN_DERIVE)) return fail(PMX_ERR_INVALID_ARGUMENT, "n_derived > 0 needs PMX_FN_DERIVE (desc.derived[] is not read for user models)");
  if (d->pmetrics_indexing && d->kernel == PMX_K_CUSTOM)
    return fail(PMX_ERR_INVALID_ARGUMENT, "pm_* indexing wraps a built-in structure: it does not apply to a user propagator (pmx_eq)");
  if (fns & PMX_FN_DYNAMICS) return fail(PMX_ERR_INVALID_ARGUMENT, "PMX_FN_DYNAMICS belongs to ODE models");
  if (d->kernel == PMX_K_CUSTOM) {
    if (!(fns & PMX_FN_EQ)) return fail(PMX_ERR_INVALID_ARGUMENT, "kernel = PMX_K_CUSTOM needs PMX_FN_EQ");
  } else {
    if (fns & PMX_FN_EQ) return fail(PMX_ERR_INVALID_ARGUMENT, "PMX_FN_EQ needs kernel = PMX_K_CUSTOM");
    if (d->kernel < 0 || d->kernel >= PMX_K_ANALYTICAL_COUNT) return fail(PMX_ERR_INVALID_ARGUMENT, "unknown analytical kernel");
    static const int kNS[12] = {1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4};
    if (d->nstates < kNS[d->kernel] + (d->pmetrics_indexing ? 1 : 0)) return fail(PMX_ERR_INVALID_ARGUMENT, "model has fewer states than its structure");
    const int np = pmx::kernel_nparams(d->kernel);
    if (d->n_bind == 0 && d->nparams < np) return fail(PMX_ERR_INVALID_ARGUMENT, "too few parameters for the structure");
    if (d->n_bind != 0 && d->n_bind != np) return fail(PMX_ERR_INVALID_ARGUMENT, "n_bind must equal the structure's parameter count");
    for (int j = 0; j < d->n_bind; ++j) {
      const pmx_bind& b = d->bind[j];
      if (b.src == PMX_SRC_PRIMARY ? (b.index < 0 || b.index >= d->nparams)
                                   : (b.src != PMX_SRC_DERIVED || b.index < 0 || b.index >= d->n_derived))
        return fail(PMX_ERR_INVALID_ARGUMENT, "bind entry out of range");
    }
  }
  for (int i = 0; i < PMX_MAX_INPUTS; ++i)
    if (d->lag_param[i] >= d->nparams || d->fa_param[i] >= d->nparams)
      return fail(PMX_ERR_INVALID_ARGUMENT, "lag_param / fa_param out of range");
  for (int i = 0; i < PMX_MAX_STATES; ++i)
    if (d->init_param[i] >= d->nparams) return fail(PMX_ERR_INVALID_ARGUMENT, "init_param out of range");
  if (!(fns & PMX_FN_OUTPUTS))
    for (int o = 0; o < d->nout; ++o) {
      const pmx_out& oo = d->out[o];
      if (oo.state < 0 || oo.state >= d->nstates) return fail(PMX_ERR_INVALID_ARGUMENT, "out.state out of range");
      if (oo.vol_src == PMX_SRC_PRIMARY && (oo.vol_index < 0 || oo.vol_index >= d->nparams))
        return fail(PMX_ERR_INVALID_ARGUMENT, "out.vol_index out of range");
      if (oo.vol_src == PMX_SRC_DERIVED && (oo.vol_index < 0 || oo.vol_index >= d->n_derived))
        return fail(PMX_ERR_INVALID_ARGUMENT, "out.vol_index (derived) out of range");
    }
  return PMX_OK;
}
// ODE model with user closures beyond the dynamics (pmx.h "user closures for either back-end", ODE models)
int32_t check_user_ode(const pmx_model_desc* d, const char* source, uint32_t fns) {
  if (!d || !source) return fail(PMX_ERR_INVALID_ARGUMENT, "null argument");
  if (d->kernel != PMX_ODE_CUSTOM) return fail(PMX_ERR_INVALID_ARGUMENT, "ODE models with user closures need kernel = PMX_ODE_CUSTOM");
  const bool dyn = (fns & PMX_FN_DYNAMICS) != 0, dynb = (fns & PMX_FN_DYNAMICS_BOLUS) != 0;
  if (dyn == dynb) return fail(PMX_ERR_INVALID_ARGUMENT, "ODE models define pmx_dynamics OR pmx_dynamics_bolus (PMX_FN_DYNAMICS | PMX_FN_DYNAMICS_BOLUS)");
  if (!(fns & PMX_FN_OUTPUTS)) return fail(PMX_ERR_INVALID_ARGUMENT, "ODE models define pmx_outputs (PMX_FN_OUTPUTS)");
  if (fns & (PMX_FN_SEQ_EQ | PMX_FN_EQ)) return fail(PMX_ERR_INVALID_ARGUMENT, "PMX_FN_SEQ_EQ / PMX_FN_EQ belong to analytical models");
  if (d->nstates < 1 || d->nstates > PMX_MAX_STATES) return fail(PMX_ERR_INVALID_ARGUMENT, "nstates out of range");
  if (d->ndrugs < 1 || d->ndrugs > PMX_MAX_INPUTS) return fail(PMX_ERR_INVALID_ARGUMENT, "ndrugs out of range (1..8 for a model with user closures)");
  if (d->nout < 1 || d->nout > PMX_MAX_OUT) return fail(PMX_ERR_INVALID_ARGUMENT, "nout out of range");
  if (d->nparams < 1 || d->nparams > PMX_MAX_PARAMS) return fail(PMX_ERR_INVALID_ARGUMENT, "nparams out of range");
  if (d->n_covariates < 0 || d->n_covariates > PMX_MAX_COVARIATES) return fail(PMX_ERR_INVALID_ARGUMENT, "n_covariates out of range");
  if (d->n_derived < 0 || d->n_derived > PMX_MAX_USER_DERIVED) return fail(PMX_ERR_INVALID_ARGUMENT, "n_derived out of range");
  if (d->n_derived > 0 && !(fns & PMX_FN_DERIVE)) return fail(PMX_ERR_INVALID_ARGUMENT, "n_derived > 0 needs PMX_FN_DERIVE (desc.derived[] is not read for user models)");
  if (d->n_bind != 0 || d->pmetrics_indexing) return fail(PMX_ERR_INVALID_ARGUMENT, "bind[] / pm indexing do not apply to ODE models with user closures");
  if (!(d->rk4_h_max > 0.0)) return fail(PMX_ERR_INVALID_ARGUMENT, "rk4_h_max must be > 0");
  if (d->ode_solver != PMX_SOLVER_RK4 && d->ode_solver != PMX_SOLVER_DOPRI5 && d->ode_solver != PMX_SOLVER_ROS2) return fail(PMX_ERR_INVALID_ARGUMENT, "unknown ode_solver");
  if (d->ode_solver != PMX_SOLVER_RK4 && !(d->ode_rtol > 0.0 && d->ode_atol > 0.0))
    return fail(PMX_ERR_INVALID_ARGUMENT, "the adaptive solver needs ode_rtol > 0 and ode_atol > 0");
  for (int i = 0; i < PMX_MAX_INPUTS; ++i) {
    if (d->lag_param[i] >= d->nparams || d->fa_param[i] >= d->nparams)
      return fail(PMX_ERR_INVALID_ARGUMENT, "lag_param / fa_param out of range");
    if (d->bolus_dest[i] >= d->nstates) return fail(PMX_ERR_INVALID_ARGUMENT, "route destination out of range");
  }
  for (int i = 0; i < PMX_MAX_STATES; ++i)
    if (d->init_param[i] >= d->nparams) return fail(PMX_ERR_INVALID_ARGUMENT, "init_param out of range");
  return PMX_OK;
}
pmx::JitSpec user_spec_of(const pmx_model_desc* d, const char* source, uint32_t fns) {
  pmx::JitSpec sp;
  sp.analytical = d->eq_kind == PMX_EQ_ANALYTICAL;
  sp.ode_user = d->eq_kind == PMX_EQ_ODE;
  sp.fns = fns;
  sp.desc = *d;
  sp.source = source;
  return sp;
}
int32_t create_user_ode(const pmx_model_desc* d, const char* source, uint32_t fns, pmx_model** out) {
  const int32_t rc = check_user_ode(d, source, fns);
  if (rc != PMX_OK) return rc;
  auto m = std::make_unique<pmx_model>();
  m->d = *d;
  m->custom = true;
  m->user_ode = true;
  m->user_fns = fns;
  m->user_lag = (fns & PMX_FN_ROUTE_LAG) != 0;
  for (int i = 0; i < PMX_MAX_INPUTS; ++i) m->user_lag |= d->lag_param[i] >= 0;
  m->has_init = true;  // (RESET ops always carry the occasion-index flag; the policy's init may be empty)
  std::string log;
  m->jit_spec = user_spec_of(d, source, fns);
  if (!pmx::jit_compile(m->jit_spec, &m->jit_code, &log))
    return fail(PMX_ERR_INVALID_ARGUMENT, "hiprtc could not compile the model source:\n" + log);
  *out = m.release();
  return PMX_OK;
}
}  // namespace

extern "C" {

int32_t pmx_model_create_user(const pmx_model_desc* d, const char* source, uint32_t functions, pmx_model** out) {
  g_err.clear();
  if (!out) return fail(PMX_ERR_INVALID_ARGUMENT, "out is null");
  *out = nullptr;
  if (!d) return fail(PMX_ERR_INVALID_ARGUMENT, "null argument");
  if (d->eq_kind == PMX_EQ_ODE) {
    if ((functions & ~static_cast<uint32_t>(PMX_FN_INIT)) == (PMX_FN_DYNAMICS | PMX_FN_OUTPUTS) && d->n_derived == 0)
      return pmx_model_create_custom(d, source, (functions & PMX_FN_INIT) ? 1 : 0, out);  // theta-indexed lag / fa: the state-machine kernels
    return create_user_ode(d, source, functions, out);
  }
  if (d->eq_kind != PMX_EQ_ANALYTICAL) return fail(PMX_ERR_INVALID_ARGUMENT, "unknown eq_kind");
  const int32_t rc = check_user_analytical(d, source, functions);
  if (rc != PMX_OK) return rc;
  auto m = std::make_unique<pmx_model>();
  m->d = *d;
  m->custom = true;
  m->user_fns = functions;
  m->user_eq = d->kernel == PMX_K_CUSTOM;
  m->user_lag = (functions & PMX_FN_ROUTE_LAG) != 0;
  for (int i = 0; i < PMX_MAX_INPUTS; ++i) m->user_lag |= d->lag_param[i] >= 0;
  m->has_init = true;  // (RESET ops always carry the occasion-index flag; the policy's init may be empty)
  std::string log;
  m->jit_spec = user_spec_of(d, source, functions);
  if (!pmx::jit_compile(m->jit_spec, &m->jit_code, &log))
    return fail(PMX_ERR_INVALID_ARGUMENT, "hiprtc could not compile the model source:\n" + log);
  *out = m.release();
  return PMX_OK;
}

int32_t pmx_debug_jit_source_user(const pmx_model_desc* d, const char* source, uint32_t functions, char** out_text) {
  g_err.clear();
  if (!out_text) return fail(PMX_ERR_INVALID_ARGUMENT, "out_text is null");
  *out_text = nullptr;
  if (!d) return fail(PMX_ERR_INVALID_ARGUMENT, "null argument");
  if (d->eq_kind == PMX_EQ_ODE && (functions & ~static_cast<uint32_t>(PMX_FN_INIT)) == (PMX_FN_DYNAMICS | PMX_FN_OUTPUTS) &&
      d->n_derived == 0)
    return pmx_debug_jit_source(d, source, (functions & PMX_FN_INIT) ? 1 : 0, out_text);
  const int32_t rc = d->eq_kind == PMX_EQ_ODE ? check_user_ode(d, source, functions) : check_user_analytical(d, source, functions);
  if (rc != PMX_OK) return rc;
  const std::string tu = pmx::jit_translation_unit(user_spec_of(d, source, functions));
  char* buf = static_cast<char*>(std::malloc(tu.size() + 1));
  if (!buf) return fail(PMX_ERR_OUT_OF_MEMORY, "malloc");
  std::memcpy(buf, tu.c_str(), tu.size() + 1);
  *out_text = buf;
  return PMX_OK;
}

void pmx_model_destroy(pmx_model* m) { delete m; }

}  // extern "C"

namespace {

pmx::CompileKey key_for(const pmx_model* m) {
  pmx::CompileKey k;
  k.eq_kind = m->d.eq_kind;
  if (m->d.eq_kind == PMX_EQ_ANALYTICAL && m->custom) {
    // user closures (pmx_analytical.hpp): covariates are looked up on the device, so the stream carries no factors;
    // absolute times on every PROP, solve marks for seq_eq, every input's rate for a user propagator, and - when the
    // model has any lag closure - ALL boluses leave the stream into one list per occasion that each lane sorts itself
    k.cov_time_mode = PMX_COV_TIME_SEGMENT_DT;  // (unused: no host-side covariate evaluation)
    k.rk4_h_max = 0.0;
    k.rate_input = (m->d.pmetrics_indexing && !m->user_eq) ? 1 : 0;  // (pm_* wrappers read rateiv[1]: analytical/mod.rs:86-88)
    k.full_rates = m->user_eq;
    k.n_rate = m->user_eq ? (m->d.ndrugs > 0 ? m->d.ndrugs : 1) : 1;
    k.want_times = true;
    k.solve_marks = true;
    k.user_cov = true;
    if (m->user_lag) {
      k.lag_merge = true;
      for (int i = 0; i < m->d.ndrugs && i < PMX_MAX_INPUTS; ++i) k.lag_mask |= (1u << i);
    }
  } else if (m->d.eq_kind == PMX_EQ_ANALYTICAL) {
    k.cov_time_mode = m->d.cov_time_mode;
    k.rk4_h_max = 0.0;
    k.n_rate = 1;
    k.rate_input = m->d.pmetrics_indexing ? 1 : 0;
    // classed fast path: theta-only coefficients, no covariates, plain indexing
    for (int i = 0; i < PMX_MAX_INPUTS; ++i)
      if (m->d.lag_param[i] >= 0) k.lag_mask |= (1u << i);
    // (bioavailability does not stop classing: the amounts in the plan are the recorded ones, each lane scales them)
    const Tunables tun = tunables();
    k.ladder = !m->dyn && k.lag_mask == 0 && !tun.disable_ladder;  // (switch: fresh exp() on every step, A/B and parity checks)
    k.n_derived = m->d.n_derived;
    std::memcpy(k.derived, m->d.derived, sizeof(k.derived));
    const bool disabled = tun.disable_classing;
    bool reads_pad = false;  // pm_ indexing: an output on model state 0 reads the wrapper's pad slot (generic walker only)
    if (m->d.pmetrics_indexing)
      for (int o = 0; o < m->d.nout && o < PMX_MAX_OUT; ++o)
        if (m->d.out[o].state == 0) reads_pad = true;
    // (one lagged input is classed too: in an exact class the bolus times are shared, so a lane's split points and the
    // propagator of every sub-interval serve all G members)
    const bool lag_ok = (k.lag_mask & (k.lag_mask - 1u)) == 0u;
    // (covariate-derived constants are classed by program shape alone, each member with its own factor rows)
    const bool plain = !m->dyn && m->d.n_covariates == 0;
    // ... where it pays: the one- and two-state structures (1-cpt + absorption 3.41 -> 2.87 ms on the C5 design); from
    // three states up the rebuild is so dominated by its own arithmetic that the batch gains nothing (C5: 19.4 -> 19.9 ms)
    const int st = pmx::kernel_structure(m->d.kernel);
    const bool dyn_ok = m->dyn && !m->d.pmetrics_indexing && k.lag_mask == 0 &&
                        (st == pmx::S_ONE || st == pmx::S_ONE_ABS || st == pmx::S_TWO);
    if (!disabled && !reads_pad && lag_ok && (plain || dyn_ok)) {
      k.class_g = (st == pmx::S_ONE || st == pmx::S_ONE_ABS || st == pmx::S_TWO) ? 8 : 4;  // == ClassBatch<KID>::G
    }
    // covariate models that take the generic walker: equal (length, factors) PROPs of an occasion share a propagator
    if (m->dyn && k.lag_mask == 0 && k.class_g == 0 && (st == pmx::S_THREE || st == pmx::S_THREE_ABS) && !m->d.pmetrics_indexing) {
      k.kfac_n = pmx::kernel_nparams(m->d.kernel);  // (the matrix-free walker, pmx_analytical_dyn3)
      for (int j = 0; j < k.kfac_n && j < 8; ++j)
        k.kfac_map[j] = (m->d.n_bind > 0 && m->d.bind[j].src == PMX_SRC_DERIVED) ? static_cast<int8_t>(m->d.bind[j].index) : int8_t(-1);
    }
    if (m->dyn && k.lag_mask == 0 && k.class_g == 0) {
      k.prop_cache_slots = tun.prop_slots >= 0 ? (tun.prop_slots > 3 ? 3 : tun.prop_slots) : 1;
      // the kept propagators live in LDS, [slot][component][256 lanes]: stay inside the 64 KB a block may take without an
      // opt-in attribute (the stream's cache codes are written for THIS number of slots, so it is fixed here)
      static const int kPropDoubles[6] = {2, 4, 6, 9, 12, 16};  // sizeof(Structure<ST>::Prop) / 8, ST = S_ONE .. S_THREE_ABS
      const int per_slot = kPropDoubles[st] * 8 * 256;
      while (k.prop_cache_slots > 0 && k.prop_cache_slots * per_slot > (64 << 10)) --k.prop_cache_slots;
    }  // (one slot: a second costs more occupancy
    // than its extra reuse returns - C5: 1 slot 16.8 ms, 2 slots 19.4 ms, none 20.2 ms; tools/c5 notes in DESIGN.md)
  } else if (m->user_ode) {
    // ODE with user lag / fa / derive closures (pmx_ode_user.hpp): covariates are looked up on the device, every PROP
    // carries its absolute [t0, t1), every input's rate rides along, and - when the model has any lag closure - ALL
    // boluses leave the stream into one list per occasion that each lane sorts itself
    k.cov_time_mode = PMX_COV_TIME_SEGMENT_END_ABS;
    k.rk4_h_max = m->d.rk4_h_max;
    k.n_rate = m->d.ndrugs > 0 ? m->d.ndrugs : 1;
    k.rate_input = 0;
    k.want_times = true;
    k.user_cov = true;
    if (m->user_lag) {
      k.lag_merge = true;
      for (int i = 0; i < m->d.ndrugs && i < PMX_MAX_INPUTS; ++i) k.lag_mask |= (1u << i);
    }
  } else {
    k.cov_time_mode = PMX_COV_TIME_SEGMENT_END_ABS;
    k.rk4_h_max = m->d.rk4_h_max;
    k.n_rate = m->d.ndrugs > 0 ? m->d.ndrugs : 1;
    k.rate_input = 0;
    for (int i = 0; i < PMX_MAX_INPUTS; ++i)
      if (m->d.lag_param[i] >= 0) k.lag_mask |= (1u << i);
    // absolute piece times: a user body may be non-autonomous; the adaptive solver steps on [t0, t1] itself
    k.want_times = m->custom || m->d.ode_solver != PMX_SOLVER_RK4;
  }
  return k;
}

// Find or build (compile + upload) the device op stream for this model flavour.
int32_t get_stream(pmx_population* pop, const pmx::CompileKey& key_in, DeviceStream** out) {
  std::lock_guard<std::mutex> lock(pop->mu);
  pmx::CompileKey key = key_in;
  if (key.kfac_n > 0 && key.prop_cache_slots == 1 && tunables().prop_slots < 0 && std::getenv("PMX_DISABLE_DYN3") == nullptr) {
    // a population without infusions takes the matrix-free walker, whose kept segment is 6-7 numbers per lane: two slots
    // fit where the matrix form held one (C5: 8 rebuilds per subject instead of 9)
    bool infusions = false;
    for (uint8_t kd : pop->hp.ev_kind) infusions |= (kd == PMX_EV_INFUSION);
    if (!infusions) key.prop_cache_slots = 2;
  }
  for (auto& s : pop->streams)
    if (s->key == key) {
      *out = s.get();
      return PMX_OK;
    }
  pmx::OpStream os;
  std::string err;
  int32_t rc = pmx::compile_ops(pop->hp, key, &os, &err);
  if (rc != PMX_OK) return fail(rc, err);
  auto ds = std::make_unique<DeviceStream>();
  ds->key = key;
  ds->max_input_used = os.max_input_used;
  ds->n_ops = os.n_ops;
  ds->n_prop = os.n_prop;
  if ((rc = upload(os.subj_op_off, &ds->dev.subj_op_off, &ds->allocs)) != PMX_OK) return rc;
  if ((rc = upload(pop->hp.subj_obs_off, &ds->dev.subj_obs_off, &ds->allocs)) != PMX_OK) return rc;
  if ((rc = upload(os.subj_order, &ds->dev.subj_order, &ds->allocs)) != PMX_OK) return rc;
  if ((rc = upload(os.op_meta, &ds->dev.op_meta, &ds->allocs)) != PMX_OK) return rc;
  if ((rc = upload(os.op_a, &ds->dev.op_a, &ds->allocs)) != PMX_OK) return rc;
  if ((rc = upload(os.op_b, &ds->dev.op_b, &ds->allocs)) != PMX_OK) return rc;
  if ((rc = upload(os.op_n, &ds->dev.op_n, &ds->allocs)) != PMX_OK) return rc;
  if ((rc = upload(os.op_rate, &ds->dev.op_rate, &ds->allocs)) != PMX_OK) return rc;
  if (key.eq_kind == PMX_EQ_ODE) {  // packed per-op records (pmx_devtypes.hpp DevOps::op_rec)
    std::vector<double> rec(static_cast<size_t>(os.n_ops) * 6, 0.0);
    const bool times = !os.op_t0.empty();
    for (int64_t o = 0; o < os.n_ops; ++o) {
      const uint64_t w = static_cast<uint64_t>(os.op_meta[o]) | (static_cast<uint64_t>(static_cast<uint32_t>(os.op_n[o])) << 32);
      std::memcpy(&rec[6 * o], &w, 8);
      rec[6 * o + 1] = os.op_a[o];
      rec[6 * o + 2] = os.op_b[o];
      rec[6 * o + 3] = key.n_rate > 0 ? os.op_rate[o * key.n_rate] : 0.0;
      rec[6 * o + 4] = times ? os.op_t0[o] : 0.0;
      rec[6 * o + 5] = times ? os.op_t1[o] : 0.0;
    }
    if ((rc = upload(rec, &ds->dev.op_rec, &ds->allocs)) != PMX_OK) return rc;
  } else {  // analytical: {meta (bits), a, b, t0} = 32 bytes
    std::vector<double> rec(static_cast<size_t>(os.n_ops) * 4, 0.0);
    const bool times = !os.op_t0.empty();
    for (int64_t o = 0; o < os.n_ops; ++o) {
      const uint64_t w = static_cast<uint64_t>(os.op_meta[o]);
      std::memcpy(&rec[4 * o], &w, 8);
      rec[4 * o + 1] = os.op_a[o];
      rec[4 * o + 2] = os.op_b[o];
      rec[4 * o + 3] = times ? os.op_t0[o] : 0.0;
    }
    if ((rc = upload(rec, &ds->dev.op_rec, &ds->allocs)) != PMX_OK) return rc;
  }
  if ((rc = upload(os.op_fac, &ds->dev.op_fac, &ds->allocs)) != PMX_OK) return rc;
  if (key.kfac_n > 0 && !os.op_fac.empty()) {  // one 64-byte record per op for the matrix-free walker (DevOps::op_kfac)
    const size_t n_ops = os.op_meta.size();
    const size_t fw = static_cast<size_t>(key.n_derived) * PMX_MAX_FACTORS;
    std::vector<double> kf(n_ops * 8, 1.0);
    for (size_t o = 0; o < n_ops; ++o) {
      for (int j = 0; j < key.kfac_n && j < 7; ++j) {
        const int dd = key.kfac_map[j];
        if (dd < 0 || dd >= key.n_derived) continue;
        double f = 1.0;  // the parameter's factors multiplied out: theta * (f0 * f1) for the descriptor's (theta * f0) * f1
        for (int q = 0; q < key.derived[dd].n_factors && q < PMX_MAX_FACTORS; ++q) f *= os.op_fac[o * fw + static_cast<size_t>(dd) * PMX_MAX_FACTORS + q];
        kf[o * 8 + j] = f;
      }
      kf[o * 8 + 7] = os.op_a[o];
    }
    if ((rc = upload(kf, &ds->dev.op_kfac, &ds->allocs)) != PMX_OK) return rc;
  }
  if ((rc = upload(os.op_t0, &ds->dev.op_t0, &ds->allocs)) != PMX_OK) return rc;
  if ((rc = upload(os.op_t1, &ds->dev.op_t1, &ds->allocs)) != PMX_OK) return rc;
  if ((rc = upload(os.lagb_off, &ds->dev.lagb_off, &ds->allocs)) != PMX_OK) return rc;
  if ((rc = upload(os.lagb_time, &ds->dev.lagb_time, &ds->allocs)) != PMX_OK) return rc;
  if ((rc = upload(os.lagb_amount, &ds->dev.lagb_amount, &ds->allocs)) != PMX_OK) return rc;
  if (key.lag_merge && (rc = upload(os.lagb_input, &ds->dev.lagb_input, &ds->allocs)) != PMX_OK) return rc;
  ds->max_lagb_per_list = os.max_lagb_per_list;
  ds->prop_cache_used = os.prop_cache_used;
  ds->no_rates = true;  // (analytical streams: a PROP's op_b is its rate)
  for (size_t o = 0; o < os.op_meta.size() && ds->no_rates; ++o)
    if ((os.op_meta[o] & 0xffu) == pmx::OP_PROP && os.op_b[o] != 0.0) ds->no_rates = false;
  ds->eig_reuse = false;  // (covariate streams: bit 27 of a PROP = "same rate constants as the previous built segment")
  if (key.prop_cache_slots > 0 && !os.op_fac.empty())
    for (size_t o = 0; o < os.op_meta.size() && !ds->eig_reuse; ++o)
      if ((os.op_meta[o] & 0xffu) == pmx::OP_PROP && (os.op_meta[o] & (1u << 27))) ds->eig_reuse = true;
  ds->prop_reuse_fraction = os.n_prop > 0 ? static_cast<double>(os.n_prop_reused) / static_cast<double>(os.n_prop) : 0.0;
  ds->dev.n_rate = key.n_rate;
  ds->dev.n_cov = 0;
  if ((key.eq_kind == PMX_EQ_ODE || key.user_cov) && pop->hp.n_cov > 0) {  // covariate segments for bodies that read them on the device
    const auto& hp = pop->hp;
    ds->dev.n_cov = hp.n_cov;
    if ((rc = upload(hp.cov_seg_off, &ds->dev.cov_seg_off, &ds->allocs)) != PMX_OK) return rc;
    if ((rc = upload(hp.seg_from, &ds->dev.seg_from, &ds->allocs)) != PMX_OK) return rc;
    if ((rc = upload(hp.seg_to, &ds->dev.seg_to, &ds->allocs)) != PMX_OK) return rc;
    if ((rc = upload(hp.seg_slope, &ds->dev.seg_slope, &ds->allocs)) != PMX_OK) return rc;
    if ((rc = upload(hp.seg_icpt, &ds->dev.seg_icpt, &ds->allocs)) != PMX_OK) return rc;
    if ((rc = upload(hp.cov_first_t, &ds->dev.cov_first_t, &ds->allocs)) != PMX_OK) return rc;
    if ((rc = upload(hp.cov_first_v, &ds->dev.cov_first_v, &ds->allocs)) != PMX_OK) return rc;
  }
  if (key.eq_kind == PMX_EQ_ANALYTICAL && !key.user_cov && key.lag_mask == 0 && os.op_fac.empty()) {
    std::vector<int64_t> off;
    std::vector<double> rec;
    pmx::build_step_stream(os, &off, &rec);
    if ((rc = upload(off, &ds->steps.subj_step_off, &ds->allocs)) != PMX_OK) return rc;
    if ((rc = upload(rec, &ds->steps.step_rec, &ds->allocs)) != PMX_OK) return rc;
  }
  if (key.class_g > 0) {
    pmx::ClassPlan cp;
    const Tunables tun = tunables();  // (tuning experiments)
    const int32_t min_class = tun.min_class > 0 ? tun.min_class : key.class_g / 2;
    const bool spread = tun.spread < 0 ? true : tun.spread != 0;  // (0.94-0.97 vs 1.05-1.11 ms on C3 in most allocations, never slower: tools/experiments/alloc_tune.py)
    const bool loose = tun.loose < 0 ? true : tun.loose != 0;  // subjects without a shared design still share a program shape: batched with per-member step lengths
    pmx::build_class_plan(pop->hp, os, key.class_g, min_class, &cp, key.ladder, spread, loose);
    if (cp.n_chunks > 0) {
      if ((rc = upload(cp.prog_meta, &ds->cls.prog_meta, &ds->allocs)) != PMX_OK) return rc;
      if ((rc = upload(cp.prog_dt, &ds->cls.prog_dt, &ds->allocs)) != PMX_OK) return rc;
      {
        std::vector<double> prec((cp.prog_meta.size() + 1) * 2, 0.0);
        for (size_t i = 0; i < cp.prog_meta.size(); ++i) {
          const uint64_t w = cp.prog_meta[i];
          std::memcpy(&prec[2 * i], &w, 8);
          prec[2 * i + 1] = cp.prog_dt[i];
        }
        if ((rc = upload(prec, &ds->cls.prog_rec, &ds->allocs)) != PMX_OK) return rc;
        if ((rc = upload(cp.chunk_rate_mask, &ds->cls.chunk_rate_mask, &ds->allocs)) != PMX_OK) return rc;
        if ((rc = upload(cp.cls_fast_mask, &ds->cls.cls_fast_mask, &ds->allocs)) != PMX_OK) return rc;
      }
      if ((rc = upload(cp.prog_t0, &ds->cls.prog_t0, &ds->allocs)) != PMX_OK) return rc;
      if ((rc = upload(cp.prog_t1, &ds->cls.prog_t1, &ds->allocs)) != PMX_OK) return rc;
      if ((rc = upload(cp.cls_prog_off, &ds->cls.cls_prog_off, &ds->allocs)) != PMX_OK) return rc;
      if ((rc = upload(cp.chunk_cls, &ds->cls.chunk_cls, &ds->allocs)) != PMX_OK) return rc;
      if ((rc = upload(cp.chunk_n, &ds->cls.chunk_n, &ds->allocs)) != PMX_OK) return rc;
      if ((rc = upload(cp.chunk_val_off, &ds->cls.chunk_val_off, &ds->allocs)) != PMX_OK) return rc;
      if ((rc = upload(cp.chunk_subj, &ds->cls.chunk_subj, &ds->allocs)) != PMX_OK) return rc;
      if ((rc = upload(cp.chunk_row, &ds->cls.chunk_row, &ds->allocs)) != PMX_OK) return rc;
      if ((rc = upload(cp.val, &ds->cls.val, &ds->allocs)) != PMX_OK) return rc;
      if ((rc = upload(cp.dtv, &ds->cls.dtv, &ds->allocs)) != PMX_OK) return rc;
      if ((rc = upload(cp.facp, &ds->cls.facp, &ds->allocs)) != PMX_OK) return rc;
      if ((rc = upload(cp.faco, &ds->cls.faco, &ds->allocs)) != PMX_OK) return rc;
      ds->cls.n_fac = cp.n_fac;
      if ((rc = upload(cp.generic_subjects, &ds->cls.generic_subjects, &ds->allocs)) != PMX_OK) return rc;
      ds->h_chunk_row = cp.chunk_row;
      ds->h_chunk_n = cp.chunk_n;
      ds->h_chunk_nobs.resize(static_cast<size_t>(cp.n_chunks));
      for (int64_t c = 0; c < cp.n_chunks; ++c) {
        const int32_t cl = cp.chunk_cls[static_cast<size_t>(c)];
        int32_t nobs = 0;
        for (int64_t o = cp.cls_prog_off[cl]; o < cp.cls_prog_off[cl + 1]; ++o) nobs += (cp.prog_meta[static_cast<size_t>(o)] >> 24) & 1u;
        ds->h_chunk_nobs[static_cast<size_t>(c)] = nobs;
      }
      {  // where each chunk's block {[G] constant sums, [G] flags, [observation][value|weight][G]} starts in a slot's cobs
        std::vector<int64_t> off(static_cast<size_t>(cp.n_chunks) + 1);
        int64_t at = 0;
        for (int64_t c = 0; c < cp.n_chunks; ++c) {
          off[static_cast<size_t>(c)] = at;
          at += (static_cast<int64_t>(ds->h_chunk_nobs[static_cast<size_t>(c)]) * 2 + 2) * cp.G;
        }
        off[static_cast<size_t>(cp.n_chunks)] = at;  // sentinel
        ds->cobs_size = at;
        if ((rc = upload(ds->h_chunk_nobs, &ds->d_chunk_nobs, &ds->allocs)) != PMX_OK) return rc;
        if ((rc = upload(off, &ds->d_chunk_obs_off, &ds->allocs)) != PMX_OK) return rc;
        {
          // one 64-byte record per chunk for pmx_analytical_classed_ll (pmx_kernels.hpp DevClassPlan::chunk_hdr): everything
          // the kernel needs of a chunk in ONE scalar fetch.  32-bit offsets and 16-bit counts: a plan outside those
          // limits simply keeps the round-2 kernel (chunk_hdr stays null).
          bool fits = cp.G <= 8 && at < (int64_t{1} << 32) && static_cast<int64_t>(cp.val.size()) < (int64_t{1} << 32) &&
                      static_cast<int64_t>(cp.prog_meta.size()) < (int64_t{1} << 32);
          for (size_t cl = 0; cl + 1 < cp.cls_prog_off.size() && fits; ++cl)
            fits = cp.cls_prog_off[cl + 1] - cp.cls_prog_off[cl] < 65536;
          if (fits) {
            std::vector<uint32_t> hdr((static_cast<size_t>(cp.n_chunks) + 1) * 16, 0);
            for (int64_t c = 0; c < cp.n_chunks; ++c) {
              uint32_t* q = &hdr[static_cast<size_t>(c) * 16];
              const int32_t cl = cp.chunk_cls[static_cast<size_t>(c)];
              q[0] = static_cast<uint32_t>(cp.chunk_n[static_cast<size_t>(c)]) |
                     (static_cast<uint32_t>(cp.cls_prog_off[cl + 1] - cp.cls_prog_off[cl]) << 16);
              q[1] = static_cast<uint32_t>(cp.cls_prog_off[cl]);
              q[2] = static_cast<uint32_t>(cp.chunk_val_off[static_cast<size_t>(c)]);
              q[3] = static_cast<uint32_t>(off[static_cast<size_t>(c)]);
              const uint64_t rm = cp.chunk_rate_mask[static_cast<size_t>(c)], fm = cp.cls_fast_mask[cl];
              q[4] = static_cast<uint32_t>(rm);
              q[5] = static_cast<uint32_t>(rm >> 32);
              q[6] = static_cast<uint32_t>(fm);
              q[7] = static_cast<uint32_t>(fm >> 32);
              for (int32_t j = 0; j < cp.G; ++j) q[8 + j] = static_cast<uint32_t>(cp.chunk_subj[static_cast<size_t>(c) * cp.G + j]);
            }
            const uint32_t* d_hdr = nullptr;
            if ((rc = upload(hdr, &d_hdr, &ds->allocs)) != PMX_OK) return rc;
            ds->cls.chunk_hdr = d_hdr;
          }
        }
      }
      ds->cls.n_chunks = cp.n_chunks;
      ds->cls.n_chunks_exact = cp.n_chunks_exact;
      ds->cls.n_generic = static_cast<int64_t>(cp.generic_subjects.size());
      ds->cls.G = cp.G;
      ds->n_classed_subjects = cp.n_classed_subjects;
    }
  }
  *out = ds.get();
  pop->streams.push_back(std::move(ds));
  return PMX_OK;
}

// Pick (or fill) the slot holding the sigma tables for `em`.  On return the slot is pinned (host_users) and `stream`
// is ordered after the slot's last use; the caller launches its kernel and then calls release_ll_slot.
int32_t acquire_ll_slot(const pmx_model* model, pmx_population* pop, DeviceStream* ds, const pmx_error_model* em,
                        void* stream, DeviceStream::LLCache** out, bool batch = false) {
  const int nout = model->d.nout;
  const auto& hp = pop->hp;
  std::lock_guard<std::mutex> lock(pop->mu);
  if (!pop->ll_ready) {  // observation-side inputs, once per population
    int32_t rc;
    std::vector<int32_t> oq(hp.obs_outeq.begin(), hp.obs_outeq.end());
    if ((rc = upload(hp.obs_value, &pop->d_obs_y, &pop->ll_allocs)) != PMX_OK) return rc;
    if ((rc = upload(oq, &pop->d_obs_outeq, &pop->ll_allocs)) != PMX_OK) return rc;
    if ((rc = upload(hp.obs_errorpoly, &pop->d_obs_poly, &pop->ll_allocs)) != PMX_OK) return rc;
    if ((rc = upload(hp.obs_censor, &pop->d_obs_cens, &pop->ll_allocs)) != PMX_OK) return rc;
    for (int64_t r = 0; r < hp.n_obs; ++r) {
      if (std::isnan(hp.obs_value[static_cast<size_t>(r)])) continue;
      const int q = hp.obs_outeq[static_cast<size_t>(r)];
      if (q >= 0 && q < 32) pop->valued_outeq_mask |= (1u << q);
      if (!hp.obs_censor.empty() && hp.obs_censor[static_cast<size_t>(r)] != PMX_CENSOR_NONE) pop->any_censored = true;
    }
    pop->ll_ready = true;
  }
  for (int q = 0; q < 32; ++q) {
    if (!((pop->valued_outeq_mask >> q) & 1u)) continue;
    if (q >= nout) return fail(PMX_ERR_OUTEQ_OUT_OF_RANGE, "observation outeq >= nout");
    // log_likelihood_matrix: MissingErrorModel fails the call (error_model.rs:1045-1080 through matrix.rs:83,104).
    // log_likelihood_batch: ResidualErrorModels::total_log_likelihood gives such a SUBJECT -inf and the call succeeds
    // (residual_error.rs:413-425): the table fill poisons the rows of that output (pmx_ll_prepare_obs), the subject's sum
    // comes out NaN with PMX_PAIR_NONFINITE, the batch entry points map that to -inf.
    if (!batch && (em[q].kind < PMX_EM_ADDITIVE || em[q].kind > PMX_EM_RES_EXPONENTIAL))
      return fail(PMX_ERR_ERROR_MODEL, "MissingErrorModel: output " + std::to_string(q) + " has observations but no error model");
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  DeviceStream::LLCache* slot = nullptr;
  for (auto& c : ds->ll_cache)
    if (static_cast<int>(c.em.size()) == nout && std::memcmp(c.em.data(), em, sizeof(pmx_error_model) * nout) == 0) {
      slot = &c;
      break;
    }
  const bool hit = slot != nullptr;
  if (!hit) {
    constexpr size_t kSlots = 4;
    if (ds->ll_cache.size() >= kSlots)  // least recently used slot nobody is about to launch on
      for (auto& c : ds->ll_cache)
        if (c.host_users == 0 && (slot == nullptr || c.stamp < slot->stamp)) slot = &c;
    if (slot == nullptr) {
      ds->ll_cache.emplace_back();
      slot = &ds->ll_cache.back();
      void* p = nullptr;
      PMX_HIP(hipMalloc(&p, static_cast<size_t>(hp.n_obs > 0 ? hp.n_obs : 1) * 4 * sizeof(double)));
      ds->allocs.push_back(p);
      slot->d_obs = static_cast<double*>(p);
      if (ds->cobs_size > 0) {
        // (+ 2 G doubles of slack: the kernel requests a step's observation block before it knows the step has one)
        PMX_HIP(hipMalloc(&p, static_cast<size_t>(ds->cobs_size + 2 * ds->cls.G) * sizeof(double)));
        ds->allocs.push_back(p);
        slot->d_cobs = static_cast<double*>(p);
      }
      PMX_HIP(hipMalloc(&p, sizeof(int32_t)));
      ds->allocs.push_back(p);
      slot->d_err = static_cast<int32_t*>(p);
      PMX_HIP(hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming));
      PMX_HIP(hipEventRecord(slot->ev, st));
    }
  }
  PMX_HIP(hipStreamWaitEvent(st, slot->ev, 0));  // after the slot's last fill / read, whatever stream that was on
  if (!hit) {
    pmx::LLPrepareArgs a{};
    a.obs_y = pop->d_obs_y;
    a.obs_outeq = pop->d_obs_outeq;
    a.obs_poly = pop->d_obs_poly;
    a.obs_cens = pop->d_obs_cens;
    for (int q = 0; q < PMX_MAX_OUT; ++q) a.em[q] = q < nout ? em[q] : pmx_error_model{};
    a.n_obs = hp.n_obs;
    a.obs4 = slot->d_obs;
    a.err = slot->d_err;
    a.chunk_row = ds->cls.chunk_row;
    a.chunk_n = ds->cls.chunk_n;
    a.chunk_nobs = ds->d_chunk_nobs;
    a.chunk_obs_off = ds->d_chunk_obs_off;
    a.chunk_cls = ds->cls.chunk_cls;
    a.cls_prog_off = ds->cls.cls_prog_off;
    a.prog_meta = ds->cls.prog_meta;
    a.n_chunks = ds->cobs_size > 0 ? ds->cls.n_chunks : 0;
    a.G = ds->cls.G;
    a.cobs = slot->d_cobs;
    a.stream = stream;
    // Filling: until the event below is recorded behind the fill, the slot must not be hit by another host thread (its
    // stream would only wait for the slot's PREVIOUS use and read a half-written table).  pop->mu is held throughout; a
    // failed fill leaves the slot keyless.
    slot->em.clear();
    hipError_t fe = hipMemsetAsync(slot->d_err, 0, sizeof(int32_t), st);
    if (fe == hipSuccess) fe = pmx::launch_ll_prepare(a);
    if (fe == hipSuccess) fe = hipEventRecord(slot->ev, st);
    if (fe != hipSuccess) return fail(PMX_ERR_HIP, std::string("log-likelihood table fill: ") + hipGetErrorString(fe));
    slot->em.assign(em, em + nout);
  }
  slot->stamp = ++ds->ll_stamp;
  slot->host_users++;
  *out = slot;
  return PMX_OK;
}

void release_ll_slot(pmx_population* pop, DeviceStream::LLCache* slot, void* stream) {
  std::lock_guard<std::mutex> lock(pop->mu);
  (void)hipEventRecord(slot->ev, static_cast<hipStream_t>(stream));
  slot->host_users--;
}

struct LLRequest {
  const pmx_error_model* em = nullptr;
  double* d_ll = nullptr;
  int64_t ld = 0;
  const int32_t** d_sigma_err = nullptr;  // out (host form): the slot's invalid-sigma counter
};

int32_t enqueue(const pmx_model* model, pmx_population* pop, const double* d_theta, int64_t P, int batch,
                double* d_pred, int64_t ld, uint8_t* d_status, void* stream, const LLRequest* llreq = nullptr,
                int state_override = -1) {
  const pmx_model_desc& d = model->d;
  if (d.n_covariates != pop->hp.n_cov)
    return fail(PMX_ERR_INVALID_ARGUMENT, "model declares " + std::to_string(d.n_covariates) +
                                              " covariates, population carries " + std::to_string(pop->hp.n_cov));
  DeviceStream* ds = nullptr;
  int32_t rc = get_stream(pop, key_for(model), &ds);
  if (rc != PMX_OK) return rc;
  // range checks the reference performs inside the event loop
  if (ds->max_input_used >= d.ndrugs)
    return fail(PMX_ERR_INPUT_OUT_OF_RANGE, "input " + std::to_string(ds->max_input_used) + " >= ndrugs " +
                                                std::to_string(d.ndrugs));  // equation/mod.rs:322-327
  if (pop->hp.max_outeq >= d.nout)
    return fail(PMX_ERR_OUTEQ_OUT_OF_RANGE,
                "outeq " + std::to_string(pop->hp.max_outeq) + " >= nout " + std::to_string(d.nout));
  if (pop->hp.n_subjects == 0) return PMX_OK;

  pmx::LaunchArgs a{};
  a.m.eq_kind = d.eq_kind;
  a.m.kernel = d.kernel;
  a.m.nparams = d.nparams;
  a.m.n_cov = d.n_covariates;
  a.m.n_derived = d.n_derived;
  a.m.n_bind = d.n_bind;
  a.m.nout = d.nout;
  a.m.pm = d.pmetrics_indexing ? 1 : 0;
  a.m.has_init = model->has_init ? 1 : 0;
  a.m.rk4_h_max = d.rk4_h_max;
  a.m.ode_rtol = d.ode_rtol;
  a.m.ode_atol = d.ode_atol;
  a.adaptive = (d.eq_kind == PMX_EQ_ODE && d.ode_solver != PMX_SOLVER_RK4) ? 1 : 0;
  a.m.ode_stiff = (d.eq_kind == PMX_EQ_ODE && d.ode_solver == PMX_SOLVER_ROS2) ? 1 : 0;
  std::memcpy(a.m.derived, d.derived, sizeof(d.derived));
  std::memcpy(a.m.bind, d.bind, sizeof(d.bind));
  std::memcpy(a.m.out, d.out, sizeof(d.out));
  a.m.state_override = state_override;  // (the run-time-compiled walkers read it)
  if (state_override >= 0)  // Prediction::state: every output equation reads the raw amount of one state
    for (int o = 0; o < PMX_MAX_OUT; ++o) a.m.out[o] = pmx_out{state_override, PMX_SRC_NONE, 0};
  for (int o = 0; o < PMX_MAX_OUT; ++o) {
    a.m.out_vol_theta[o] = -1;
    if (a.m.out[o].vol_src == PMX_SRC_PRIMARY) a.m.out_vol_theta[o] = a.m.out[o].vol_index;
    if (a.m.out[o].vol_src == PMX_SRC_DERIVED && a.m.out[o].vol_index >= 0 && a.m.out[o].vol_index < PMX_MAX_DERIVED)
      a.m.out_vol_theta[o] = d.derived[a.m.out[o].vol_index].src_param;
  }
  std::memcpy(a.m.init_param, d.init_param, sizeof(d.init_param));
  std::memcpy(a.m.bolus_dest, d.bolus_dest, sizeof(d.bolus_dest));
  std::memcpy(a.m.infusion_dest, d.infusion_dest, sizeof(d.infusion_dest));
  std::memcpy(a.m.fa_param, d.fa_param, sizeof(d.fa_param));
  for (int i = 0; i < PMX_MAX_INPUTS; ++i) {
    if (d.fa_param[i] >= 0) a.m.has_fa = 1;
    if (d.lag_param[i] >= 0 && a.m.n_lag_slots < pmx::kMaxLagSlots && !model->user_ode) {
      a.m.lag_input[a.m.n_lag_slots] = i;
      a.m.lag_param[a.m.n_lag_slots] = d.lag_param[i];
      a.m.lag_dest[a.m.n_lag_slots] = (d.eq_kind == PMX_EQ_ODE && d.bolus_dest[i] >= 0) ? d.bolus_dest[i] : i;
      a.m.n_lag_slots++;
    }
  }
  a.ops = ds->dev;
  {
    // ODE PAIR kernel, steps per trip of the lane state machine (pmx_ode.hpp ode_pair_body): tools/experiments/steps_per_trip_sweep.sh
    const int64_t n_pairs = batch ? pop->hp.n_subjects : pop->hp.n_subjects * P;
    const int32_t spt_tuned = tunables().steps_per_trip;
    a.ops.steps_per_trip = spt_tuned > 0 ? spt_tuned : (n_pairs <= 131072 ? 48 : 32);
  }
  a.theta = d_theta;
  a.P = batch ? 1 : P;
  a.S = pop->hp.n_subjects;
  a.pred = d_pred;
  a.ld = batch ? 1 : ld;
  a.status = d_status;
  a.batch = batch;
  a.dyn = model->dyn ? 1 : 0;
  a.stream = stream;
  a.cls = ds->cls;
  a.use_classes = ds->cls.n_chunks > 0 ? 1 : 0;
  {
    // the lean walker serves the plain models: rate constants and volumes fixed per lane, no lag, no pm_ pad slot
    bool plain = d.eq_kind == PMX_EQ_ANALYTICAL && !model->dyn && !model->custom && a.m.n_lag_slots == 0 && !d.pmetrics_indexing &&
                 std::getenv("PMX_DISABLE_STEPS") == nullptr;
    for (int o = 0; o < d.nout && o < PMX_MAX_OUT; ++o)
      if (a.m.out[o].vol_src == PMX_SRC_DERIVED && d.derived[a.m.out[o].vol_index].n_factors > 0) plain = false;
    if (plain) a.steps = ds->steps;
  }
  // the stream's codes were written for key.prop_cache_slots slots; the kernel decodes them with the same number
  a.prop_slots = ds->prop_cache_used > 0 ? ds->key.prop_cache_slots : 0;
  a.no_rates = (ds->no_rates && d.eq_kind == PMX_EQ_ANALYTICAL && ds->dev.op_kfac != nullptr && std::getenv("PMX_DISABLE_DYN3") == nullptr) ? 1 : 0;
  a.eig_reuse = ds->eig_reuse ? 1 : 0;
  a.dyn_tile = tunables().dyn_tile;  // (0 = the default tile; 64 and 256 measured the same with one slot)
  DeviceStream::LLCache* slot = nullptr;
  struct SlotGuard {  // the slot is released (event recorded on the stream) however this function leaves
    pmx_population* pop;
    DeviceStream::LLCache** slot;
    void* stream;
    ~SlotGuard() {
      if (*slot) release_ll_slot(pop, *slot, stream);
    }
  } slot_guard{pop, &slot, stream};
  if (llreq != nullptr) {
    rc = acquire_ll_slot(model, pop, ds, llreq->em, stream, &slot, batch != 0);
    if (rc != PMX_OK) return rc;
    a.ops.ll_obs = slot->d_obs;
    a.ops.ll_out = llreq->d_ll;
    a.ops.ll_ld = llreq->ld;
    a.cls.cobs = slot->d_cobs;
    a.cls.chunk_obs_off = ds->d_chunk_obs_off;
    if (llreq->d_sigma_err) *llreq->d_sigma_err = slot->d_err;
    a.ll_censored = pop->any_censored ? 1 : 0;  // (known once the population's observation arrays are on the device)
    for (int q = 0; q < d.nout && q < PMX_MAX_OUT; ++q)
      if (llreq->em[q].kind >= PMX_EM_RES_CONSTANT) a.ll_censored = 1;  // residual models fold from the full records too
  }
  // GRID (lane = support point, wave-uniform op stream) vs PAIR (lane = pair, divergent streams): measured crossovers
  // (tools/experiments/pairgrid_sweep.sh) are 8 support points when the classed kernel serves most subjects, ~48 when every
  // subject goes through the generic walker (a GRID wave with few live lanes still pays the whole walk); ODE: 32.
  int64_t grid_min_p = 32;
  if (d.eq_kind == PMX_EQ_ANALYTICAL)
    grid_min_p = (a.use_classes && 2 * ds->n_classed_subjects >= a.S) ? 8 : 48;
  if (const int32_t g = tunables().grid_min_p; g > 0) grid_min_p = g;  // tuning experiments
  a.tune_cpb = tunables().cpb;
  a.tune_ll_old = tunables().ll_old ? 1 : 0;
  if (!batch && P >= grid_min_p) {
    a.mode = pmx::MODE_GRID;
    a.n_ptiles = static_cast<int32_t>((P + 255) / 256);
    // enough blocks to fill 256 CUs several times over, few enough that the per-block
    // rate-constant setup stays amortised
    int64_t chunk = (a.S * a.n_ptiles) / 8192;
    if (chunk < 1) chunk = 1;
    if (chunk > 64) chunk = 64;
    a.s_chunk = static_cast<int32_t>(chunk);
  } else {
    a.mode = pmx::MODE_PAIR;
    a.n_ptiles = 1;
    a.s_chunk = 1;
  }
  const int64_t blocks = a.mode == pmx::MODE_GRID ? ((a.S + a.s_chunk - 1) / a.s_chunk) * a.n_ptiles
                                                  : ((batch ? a.S : a.S * a.P) + 255) / 256;
  if (blocks > 0x7fffffffLL) return fail(PMX_ERR_INVALID_ARGUMENT, "grid too large for one launch");
  // Status bytes need no memset before the launch (it cost ~70 us of serialisation per pass): the PAIR and ODE kernels
  // write every pair's byte; the analytical GRID kernels clear a subject's bytes with 8-byte stores when the row
  // length allows (mode 1) and otherwise write every byte too (mode 2).  Every subject is visited: the generic walker
  // owns the subjects no class holds, empty ones included.
  a.cls.zero_status = 0;
  if (d_status != nullptr && a.mode == pmx::MODE_GRID && d.eq_kind == PMX_EQ_ANALYTICAL)
    a.cls.zero_status = (P % 8 == 0 && reinterpret_cast<uintptr_t>(d_status) % 8 == 0 && (ds->cls.n_chunks == 0 || ds->cls.G <= 8)) ? 1 : 2;
  const char* name = "";
  hipError_t e;
  if (model->custom) {
    // hiprtc-compiled model: resolve (once per device) and launch the matching entry point of its module
    const pmx::JitModule* jm = nullptr;
    {
      std::lock_guard<std::mutex> lock(model->jit_mu);
      // closure walkers keep 64 landing times per lane; an occasion with more takes the build with the scan path in
      const bool big = (d.eq_kind == PMX_EQ_ANALYTICAL || model->user_ode) && ds->max_lagb_per_list > pmx::kUserLagKept;
      if (big && model->jit_code_big.empty()) {
        pmx::JitSpec sp = model->jit_spec;
        sp.big_lists = true;
        std::string log;
        if (!pmx::jit_compile(sp, &model->jit_code_big, &log))
          return fail(PMX_ERR_HIP, "hiprtc could not compile the big-lists build of the model:\n" + log);
      }
      auto& modules = big ? model->jit_modules_big : model->jit_modules;
      auto it = modules.find(pop->device);
      if (it == modules.end()) {
        pmx::JitModule mod;
        const hipError_t le = pmx::jit_load(big ? model->jit_code_big : model->jit_code, &mod,
                                            d.eq_kind == PMX_EQ_ANALYTICAL ? pmx::JIT_ANALYTICAL
                                                                           : (model->user_ode ? pmx::JIT_ODE_USER : pmx::JIT_ODE));
        if (le != hipSuccess) return fail(PMX_ERR_HIP, std::string("loading the compiled model: ") + hipGetErrorString(le));
        it = modules.emplace(pop->device, mod).first;
      }
      jm = &it->second;
    }
    const bool ua = d.eq_kind == PMX_EQ_ANALYTICAL;  // user analytical model: [mode][0][LL][0]; general ODE walker: [mode][0][LL][ADAPT]
    const int lag = (!ua && !model->user_ode && a.m.n_lag_slots > 0) ? 1 : 0, ll = a.ops.ll_obs != nullptr ? 1 : 0,
              ad = (!ua && a.adaptive) ? 1 : 0;
    const int mode = a.mode == pmx::MODE_GRID ? 0 : 1;
    static const char* const kNames[3][2][2] = {
        {{"pmx_jit_ode_rk4_grid", "pmx_jit_ode_rk4_grid<lag>"}, {"pmx_jit_ode_rk4_pair", "pmx_jit_ode_rk4_pair<lag>"}},
        {{"pmx_jit_ode_dopri5_grid", "pmx_jit_ode_dopri5_grid<lag>"}, {"pmx_jit_ode_dopri5_pair", "pmx_jit_ode_dopri5_pair<lag>"}},
        {{"pmx_jit_ode_ros2_grid", "pmx_jit_ode_ros2_grid<lag>"}, {"pmx_jit_ode_ros2_pair", "pmx_jit_ode_ros2_pair<lag>"}}};
    const int sv = ad ? (a.m.ode_stiff ? 2 : 1) : 0;  // (the same compiled entry point serves both adaptive steppers)
    name = kNames[sv][mode][lag];
    if (ua) name = mode == 0 ? "pmx_jit_analytical_grid" : "pmx_jit_analytical_pair";
    static const char* const kUser[3][2] = {{"pmx_jit_ode_user_rk4_grid", "pmx_jit_ode_user_rk4_pair"},
                                            {"pmx_jit_ode_user_dopri5_grid", "pmx_jit_ode_user_dopri5_pair"},
                                            {"pmx_jit_ode_user_ros2_grid", "pmx_jit_ode_user_ros2_pair"}};
    if (model->user_ode) name = kUser[sv][mode];
    // (experiment hook, tools/experiments/user_static: PMX_DEBUG_STATIC_SO names a shared object holding the SAME translation
    // unit compiled ahead of time by hipcc, with a launcher for its GRID prediction entry point)
    typedef int (*static_launch_t)(const void*, const void*, const double*, int64_t, int64_t, int32_t, int32_t, double*, int64_t,
                                   uint8_t*, uint32_t, uint32_t, void*);
    static static_launch_t s_static = []() -> static_launch_t {
      const char* so = std::getenv("PMX_DEBUG_STATIC_SO");
      if (!so) return nullptr;
      void* h = dlopen(so, RTLD_NOW | RTLD_LOCAL);
      return h ? reinterpret_cast<static_launch_t>(dlsym(h, "pmx_static_launch")) : nullptr;
    }();
    if (s_static && mode == 0 && !ll && a.S > 0 && a.P > 0) {
      const int64_t n_chunks = (a.S + a.s_chunk - 1) / a.s_chunk;
      const int rc_s = s_static(&a.m, &a.ops, a.theta, a.P, a.S, a.s_chunk, a.n_ptiles, a.pred, a.ld, a.status,
                                static_cast<uint32_t>(n_chunks * a.n_ptiles), a.P <= 64 ? 64u : (a.P <= 128 ? 128u : 256u), stream);
      e = rc_s == 0 ? hipSuccess : hipErrorUnknown;
      name = "pmx_static_agrid";
    } else if (a.S <= 0 || (a.P <= 0 && !a.batch)) {
      e = hipSuccess;
    } else if (mode == 0) {
      const int64_t n_chunks = (a.S + a.s_chunk - 1) / a.s_chunk;
      void* args[] = {&a.m, &a.ops, &a.theta, &a.P, &a.S, &a.s_chunk, &a.n_ptiles, &a.pred, &a.ld, &a.status};
      e = hipModuleLaunchKernel(jm->fn[0][lag][ll][ad], static_cast<uint32_t>(n_chunks * a.n_ptiles), 1, 1,
                                a.P <= 64 ? 64u : (a.P <= 128 ? 128u : 256u), 1, 1, 0,
                                static_cast<hipStream_t>(stream), args, nullptr);
    } else {
      const int64_t n_pairs = a.batch ? a.S : a.S * a.P;
      void* args[] = {&a.m, &a.ops, &a.theta, &a.P, &a.S, &a.batch, &a.pred, &a.ld, &a.status};
      e = hipModuleLaunchKernel(jm->fn[1][lag][ll][ad], static_cast<uint32_t>((n_pairs + 255) / 256), 1, 1, 256, 1, 1, 0,
                                static_cast<hipStream_t>(stream), args, nullptr);
    }
  } else {
    e = pmx::launch_predict(a, &name);
  }
  g_kernel_name = name;
  if (e != hipSuccess) return fail(PMX_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  return PMX_OK;
}

// ---- host-pointer forms -------------------------------------------------------------------------------------------
int32_t ws_get(pmx_population* pop, HostWorkspace** out) {
  std::lock_guard<std::mutex> lock(pop->mu);
  if (!pop->ws) {
    auto ws = std::make_unique<HostWorkspace>();
    PMX_HIP(hipStreamCreateWithFlags(&ws->compute, hipStreamNonBlocking));
    PMX_HIP(hipStreamCreateWithFlags(&ws->copy, hipStreamNonBlocking));
    PMX_HIP(hipMalloc(reinterpret_cast<void**>(&ws->d_flag), sizeof(int32_t)));
    PMX_HIP(hipHostMalloc(reinterpret_cast<void**>(&ws->h_flag), sizeof(int32_t), hipHostMallocDefault));
    for (int i = 0; i < 2; ++i) {
      PMX_HIP(hipHostMalloc(&ws->bounce[i], HostWorkspace::kBounce, hipHostMallocDefault));
      PMX_HIP(hipEventCreateWithFlags(&ws->bev[i], hipEventDisableTiming));
    }
    pop->ws = std::move(ws);
  }
  *out = pop->ws.get();
  return PMX_OK;
}

int32_t ws_reserve(void** p, size_t* cap, size_t need) {
  if (need <= *cap && *p) return PMX_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  const size_t want = need > 0 ? ((need + (1u << 20) - 1) >> 20) << 20 : (1u << 20);  // whole MiB
  PMX_HIP(hipMalloc(p, want));
  *cap = want;
  return PMX_OK;
}

bool is_pinned_host(const void* p) {
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // (an ordinary malloc'ed pointer: "invalid value", not an error of ours)
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

// rows x row_bytes from a dense device buffer to host rows `dst_pitch` apart, after everything enqueued on ws->compute.
// Page-locked destinations (pmx_host_alloc, hipHostMalloc, hipHostRegister) take ONE DMA at link rate; pageable ones go
// through the two pinned bounce buffers, the DMA of one piece overlapping the CPU copy of the previous one.
int32_t ws_copy_out(HostWorkspace* ws, void* dst, size_t dst_pitch, const void* src, size_t row_bytes, size_t rows) {
  if (rows == 0 || row_bytes == 0) {
    PMX_HIP(hipStreamSynchronize(ws->compute));
    return PMX_OK;
  }
  if (is_pinned_host(dst)) {
    if (dst_pitch == row_bytes)
      PMX_HIP(hipMemcpyAsync(dst, src, row_bytes * rows, hipMemcpyDeviceToHost, ws->compute));
    else
      PMX_HIP(hipMemcpy2DAsync(dst, dst_pitch, src, row_bytes, row_bytes, rows, hipMemcpyDeviceToHost, ws->compute));
    PMX_HIP(hipStreamSynchronize(ws->compute));
    return PMX_OK;
  }
  PMX_HIP(hipStreamSynchronize(ws->compute));
  const size_t total = row_bytes * rows;
  const size_t piece = HostWorkspace::kBounce;
  const size_t n_pieces = (total + piece - 1) / piece;
  auto land = [&](size_t k) {  // bounce[k % 2] -> the caller's rows
    const size_t off = k * piece, len = (off + piece <= total) ? piece : total - off;
    const char* b = static_cast<const char*>(ws->bounce[k % 2]);
    if (dst_pitch == row_bytes) {
      std::memcpy(static_cast<char*>(dst) + off, b, len);
      return;
    }
    size_t done = 0;
    while (done < len) {  // split at row ends
      const size_t at = off + done, r = at / row_bytes, c = at % row_bytes;
      const size_t n = (row_bytes - c < len - done) ? row_bytes - c : len - done;
      std::memcpy(static_cast<char*>(dst) + r * dst_pitch + c, b + done, n);
      done += n;
    }
  };
  for (size_t k = 0; k < n_pieces; ++k) {
    const size_t off = k * piece, len = (off + piece <= total) ? piece : total - off;
    PMX_HIP(hipMemcpyAsync(ws->bounce[k % 2], static_cast<const char*>(src) + off, len, hipMemcpyDeviceToHost, ws->copy));
    PMX_HIP(hipEventRecord(ws->bev[k % 2], ws->copy));
    if (k > 0) {
      PMX_HIP(hipEventSynchronize(ws->bev[(k - 1) % 2]));
      land(k - 1);
    }
  }
  PMX_HIP(hipEventSynchronize(ws->bev[(n_pieces - 1) % 2]));
  land(n_pieces - 1);
  return PMX_OK;
}

// Every exit path of a host-pointer entry point leaves nothing in flight: the caller may free or reuse theta and its
// (possibly page-locked) outputs as soon as the call returns - also when it returns an error half-way - and the next
// call reuses the workspace buffers.
struct WsDrain {
  HostWorkspace* ws;
  ~WsDrain() {
    (void)hipStreamSynchronize(ws->compute);
    (void)hipStreamSynchronize(ws->copy);
  }
};

// the per-pair status bytes: "did any pair fail" comes back as ONE flag reduced on the device (the array itself is only
// copied when the caller asked for it: 100 MB for C3)
int32_t ws_finish_status(HostWorkspace* ws, uint8_t* status, size_t n_status, bool* any_failed) {
  PMX_HIP(hipMemsetAsync(ws->d_flag, 0, sizeof(int32_t), ws->compute));
  PMX_HIP(pmx::launch_status_any(static_cast<const uint8_t*>(ws->d_status), static_cast<int64_t>(n_status), ws->d_flag, ws->compute));
  PMX_HIP(hipMemcpyAsync(ws->h_flag, ws->d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, ws->compute));
  if (status) {
    const int32_t rc = ws_copy_out(ws, status, n_status, ws->d_status, n_status, 1);
    if (rc != PMX_OK) return rc;
  } else {
    PMX_HIP(hipStreamSynchronize(ws->compute));
  }
  *any_failed = *ws->h_flag != 0;
  return PMX_OK;
}

int32_t predict_host(const pmx_model* model, const pmx_population* cpop, const double* theta, int64_t P, int batch,
                     double* pred, int64_t ld, uint8_t* status) {
  g_err.clear();
  if (!model || !cpop || !theta || !pred) return fail(PMX_ERR_INVALID_ARGUMENT, "null argument");
  pmx_population* pop = const_cast<pmx_population*>(cpop);
  const int64_t S = pop->hp.n_subjects, NO = pop->hp.n_obs;
  if (!batch && (P <= 0 || ld < P)) return fail(PMX_ERR_INVALID_ARGUMENT, "n_support must be > 0 and ld_pred >= n_support");
  DeviceGuard g;
  PMX_HIP(g.enter(pop->device));
  HostWorkspace* ws = nullptr;
  int32_t rc = ws_get(pop, &ws);
  if (rc != PMX_OK) return rc;
  std::lock_guard<std::mutex> turn(ws->mu);
  WsDrain drain{ws};
  const int64_t rows_theta = batch ? S : P;
  const int64_t Pd = batch ? 1 : P;  // the device matrix is dense; the caller's padding columns are never touched
  const size_t n_status = static_cast<size_t>(batch ? S : S * P);
  const size_t theta_bytes = static_cast<size_t>(rows_theta) * model->d.nparams * sizeof(double);
  const size_t pred_bytes = static_cast<size_t>(NO) * Pd * sizeof(double);
  if ((rc = ws_reserve(&ws->d_theta, &ws->theta_cap, theta_bytes)) != PMX_OK) return rc;
  if ((rc = ws_reserve(&ws->d_out, &ws->out_cap, pred_bytes)) != PMX_OK) return rc;
  if ((rc = ws_reserve(&ws->d_status, &ws->status_cap, n_status)) != PMX_OK) return rc;
  PMX_HIP(hipMemcpyAsync(ws->d_theta, theta, theta_bytes, hipMemcpyHostToDevice, ws->compute));
  if (n_status > 0) PMX_HIP(hipMemsetAsync(ws->d_status, 0, n_status, ws->compute));
  rc = enqueue(model, pop, static_cast<const double*>(ws->d_theta), P, batch, static_cast<double*>(ws->d_out), Pd,
               static_cast<uint8_t*>(ws->d_status), ws->compute);
  if (rc != PMX_OK) return rc;
  bool any_failed = false;
  if ((rc = ws_finish_status(ws, status, n_status, &any_failed)) != PMX_OK) return rc;
  if ((rc = ws_copy_out(ws, pred, static_cast<size_t>(batch ? 1 : ld) * sizeof(double), ws->d_out,
                        static_cast<size_t>(Pd) * sizeof(double), static_cast<size_t>(NO))) != PMX_OK)
    return rc;
  if (any_failed)
    return fail(PMX_ERR_PAIR_FAILED, "at least one (subject, support point) pair failed; see the status array");
  return PMX_OK;
}

// ll[s][p] (matrix shape) or ll[s] (batch shape: subject s with theta row s, likelihood/mod.rs:119-177)
int32_t loglik_host(const pmx_model* model, const pmx_population* cpop, const pmx_error_model* em, const double* theta,
                    int64_t P, int batch, double* ll, int64_t ld_ll, uint8_t* status) {
  pmx_population* pop = const_cast<pmx_population*>(cpop);
  const int64_t S = pop->hp.n_subjects;
  DeviceGuard g;
  PMX_HIP(g.enter(pop->device));
  HostWorkspace* ws = nullptr;
  int32_t rc = ws_get(pop, &ws);
  if (rc != PMX_OK) return rc;
  std::lock_guard<std::mutex> turn(ws->mu);
  WsDrain drain{ws};
  const int64_t Pd = batch ? 1 : P;
  const size_t theta_bytes = static_cast<size_t>(batch ? S : P) * model->d.nparams * sizeof(double);
  const size_t ll_bytes = static_cast<size_t>(S) * Pd * sizeof(double);
  const size_t n_status = static_cast<size_t>(S) * Pd;
  if ((rc = ws_reserve(&ws->d_theta, &ws->theta_cap, theta_bytes)) != PMX_OK) return rc;
  if ((rc = ws_reserve(&ws->d_out, &ws->out_cap, ll_bytes)) != PMX_OK) return rc;
  if ((rc = ws_reserve(&ws->d_status, &ws->status_cap, n_status)) != PMX_OK) return rc;
  PMX_HIP(hipMemcpyAsync(ws->d_theta, theta, theta_bytes, hipMemcpyHostToDevice, ws->compute));
  if (n_status > 0) PMX_HIP(hipMemsetAsync(ws->d_status, 0, n_status, ws->compute));
  const int32_t* d_sigma_err = nullptr;
  LLRequest req{em, static_cast<double*>(ws->d_out), Pd, &d_sigma_err};
  rc = enqueue(model, pop, static_cast<const double*>(ws->d_theta), P, batch, static_cast<double*>(ws->d_out), Pd,
               static_cast<uint8_t*>(ws->d_status), ws->compute, &req);
  if (rc != PMX_OK) return rc;
  if (d_sigma_err) {  // ErrorModelError::NegativeSigma / NonFiniteSigma (error_model.rs:1073-1077), found on the device
    PMX_HIP(hipMemcpyAsync(ws->h_flag, d_sigma_err, sizeof(int32_t), hipMemcpyDeviceToHost, ws->compute));
    PMX_HIP(hipStreamSynchronize(ws->compute));
    const int32_t n_bad = *ws->h_flag;
    if (n_bad > 0)
      return fail(PMX_ERR_ERROR_MODEL, "NegativeSigma / NonFiniteSigma for " + std::to_string(n_bad) + " observation(s)");
  }
  bool any_failed = false;
  if ((rc = ws_finish_status(ws, status, n_status, &any_failed)) != PMX_OK) return rc;
  if ((rc = ws_copy_out(ws, ll, static_cast<size_t>(batch ? 1 : ld_ll) * sizeof(double), ws->d_out,
                        static_cast<size_t>(Pd) * sizeof(double), static_cast<size_t>(S))) != PMX_OK)
    return rc;
  if (any_failed) {
    if (batch) {
      // log_likelihood_batch maps a failed subject to -inf instead of failing the call (likelihood/mod.rs:137-140):
      // the status array names them, the caller's row is overwritten here
      std::vector<uint8_t> hst(n_status);
      PMX_HIP(hipMemcpy(hst.data(), ws->d_status, n_status, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < n_status; ++i)
        if (hst[i] != PMX_PAIR_OK) ll[i] = -std::numeric_limits<double>::infinity();
      return PMX_OK;
    }
    return fail(PMX_ERR_PAIR_FAILED, "at least one (subject, support point) pair failed; see the status array");
  }
  return PMX_OK;
}

}  // namespace

extern "C" {

int32_t pmx_predict(const pmx_model* model, const pmx_population* pop, const double* theta, int64_t n_support,
                    double* pred, int64_t ld_pred, uint8_t* status) {
  return predict_host(model, pop, theta, n_support, 0, pred, ld_pred, status);
}

int32_t pmx_predict_batch(const pmx_model* model, const pmx_population* pop, const double* theta, double* pred,
                          uint8_t* status) {
  return predict_host(model, pop, theta, 1, 1, pred, 1, status);
}

int32_t pmx_predict_device(const pmx_model* model, const pmx_population* cpop, const double* d_theta,
                           int64_t n_support, double* d_pred, int64_t ld_pred, uint8_t* d_status, void* stream) {
  g_err.clear();
  if (!model || !cpop || !d_theta || !d_pred) return fail(PMX_ERR_INVALID_ARGUMENT, "null argument");
  if (n_support <= 0 || ld_pred < n_support) return fail(PMX_ERR_INVALID_ARGUMENT, "n_support must be > 0 and ld_pred >= n_support");
  pmx_population* pop = const_cast<pmx_population*>(cpop);
  DeviceGuard g;
  PMX_HIP(g.enter(pop->device));
  return enqueue(model, pop, d_theta, n_support, 0, d_pred, ld_pred, d_status, stream);
}

int32_t pmx_time_predict_device(const pmx_model* model, const pmx_population* cpop, const double* d_theta,
                                int64_t n_support, double* d_pred, int64_t ld_pred, int32_t reps, void* stream,
                                double* ms_per_pass) {
  g_err.clear();
  if (!model || !cpop || !d_theta || !d_pred || !ms_per_pass) return fail(PMX_ERR_INVALID_ARGUMENT, "null argument");
  if (n_support <= 0 || ld_pred < n_support || reps < 1) return fail(PMX_ERR_INVALID_ARGUMENT, "n_support, ld_pred or reps out of range");
  pmx_population* pop = const_cast<pmx_population*>(cpop);
  DeviceGuard g;
  PMX_HIP(g.enter(pop->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  int32_t rc = enqueue(model, pop, d_theta, n_support, 0, d_pred, ld_pred, nullptr, stream);
  if (rc != PMX_OK) return rc;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  PMX_HIP(hipEventCreate(&e0));
  PMX_HIP(hipEventCreate(&e1));
  PMX_HIP(hipEventRecord(e0, st));
  for (int32_t i = 0; i < reps && rc == PMX_OK; ++i) rc = enqueue(model, pop, d_theta, n_support, 0, d_pred, ld_pred, nullptr, stream);
  if (rc == PMX_OK) {
    float ms = 0.0f;
    hipError_t e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e != hipSuccess) rc = fail(PMX_ERR_HIP, hipGetErrorString(e));
    *ms_per_pass = static_cast<double>(ms) / reps;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

int32_t pmx_predict_state_device(const pmx_model* model, const pmx_population* cpop, const double* d_theta,
                                 int64_t n_support, int32_t state, double* d_out, int64_t ld_out, uint8_t* d_status,
                                 void* stream) {
  g_err.clear();
  if (!model || !cpop || !d_theta || !d_out) return fail(PMX_ERR_INVALID_ARGUMENT, "null argument");
  if (n_support <= 0 || ld_out < n_support) return fail(PMX_ERR_INVALID_ARGUMENT, "n_support must be > 0 and ld_out >= n_support");
  if (state < 0 || state >= model->d.nstates) return fail(PMX_ERR_INVALID_ARGUMENT, "state out of range");
  pmx_population* pop = const_cast<pmx_population*>(cpop);
  DeviceGuard g;
  PMX_HIP(g.enter(pop->device));
  return enqueue(model, pop, d_theta, n_support, 0, d_out, ld_out, d_status, stream, nullptr, state);
}

int32_t pmx_predict_batch_device(const pmx_model* model, const pmx_population* cpop, const double* d_theta,
                                 double* d_pred, uint8_t* d_status, void* stream) {
  g_err.clear();
  if (!model || !cpop || !d_theta || !d_pred) return fail(PMX_ERR_INVALID_ARGUMENT, "null argument");
  pmx_population* pop = const_cast<pmx_population*>(cpop);
  DeviceGuard g;
  PMX_HIP(g.enter(pop->device));
  return enqueue(model, pop, d_theta, 1, 1, d_pred, 1, d_status, stream);
}

}  // extern "C"

// ---- host-side introspection ---------------------------------------------------
namespace {
struct DebugOwner {
  pmx::HostPopulation hp;
  pmx::OpStream os;
};
}  // namespace

extern "C" {

int32_t pmx_debug_compile(const pmx_population_desc* pop, const pmx_model_desc* model, pmx_op_stream_view* out) {
  g_err.clear();
  if (!pop || !model || !out) return fail(PMX_ERR_INVALID_ARGUMENT, "null argument");
  std::memset(out, 0, sizeof(*out));
  pmx_model* m = nullptr;
  int32_t rc = pmx_model_create(model, &m);
  if (rc != PMX_OK) return rc;
  std::unique_ptr<pmx_model> mg(m);
  auto own = std::make_unique<DebugOwner>();
  std::string err;
  rc = pmx::build_host_population(pop, &own->hp, &err);
  if (rc != PMX_OK) return fail(rc, err);
  rc = pmx::compile_ops(own->hp, key_for(m), &own->os, &err);
  if (rc != PMX_OK) return fail(rc, err);
  const pmx::OpStream& os = own->os;
  out->n_subjects = own->hp.n_subjects;
  out->n_ops = os.n_ops;
  out->n_cov = own->hp.n_cov;
  out->n_rate = os.key.n_rate;
  out->max_input_used = os.max_input_used;
  out->max_outeq = own->hp.max_outeq;
  out->subj_op_off = os.subj_op_off.data();
  out->op_meta = os.op_meta.data();
  out->op_a = os.op_a.data();
  out->op_b = os.op_b.data();
  out->op_n = os.op_n.empty() ? nullptr : os.op_n.data();
  out->op_rate = os.op_rate.empty() ? nullptr : os.op_rate.data();
  out->op_cov = os.op_cov.empty() ? nullptr : os.op_cov.data();
  out->subj_order = os.subj_order.data();
  out->owner = own.release();
  return PMX_OK;
}

int32_t pmx_debug_class_plan(const pmx_population_desc* pop, const pmx_model_desc* model, int64_t* counts) {
  g_err.clear();
  if (!pop || !model || !counts) return fail(PMX_ERR_INVALID_ARGUMENT, "null argument");
  pmx_model* m = nullptr;
  int32_t rc = pmx_model_create(model, &m);
  if (rc != PMX_OK) return rc;
  std::unique_ptr<pmx_model> mg(m);
  pmx::HostPopulation hp;
  pmx::OpStream os;
  std::string err;
  rc = pmx::build_host_population(pop, &hp, &err);
  if (rc != PMX_OK) return fail(rc, err);
  const pmx::CompileKey key = key_for(m);
  rc = pmx::compile_ops(hp, key, &os, &err);
  if (rc != PMX_OK) return fail(rc, err);
  for (int i = 0; i < 5; ++i) counts[i] = 0;
  counts[3] = hp.n_subjects;
  if (key.class_g > 0) {
    pmx::ClassPlan cp;
    pmx::build_class_plan(hp, os, key.class_g, key.class_g / 2, &cp, key.ladder, true, true);
    counts[0] = cp.n_chunks_exact;
    counts[1] = cp.n_chunks - cp.n_chunks_exact;
    counts[2] = cp.n_classed_subjects;
    counts[3] = static_cast<int64_t>(cp.generic_subjects.size());
    counts[4] = cp.G;
  }
  return PMX_OK;
}

void pmx_debug_free(pmx_op_stream_view* view) {
  if (!view || !view->owner) return;
  delete static_cast<DebugOwner*>(view->owner);
  std::memset(view, 0, sizeof(*view));
}

}  // extern "C"

// ---- fused log-likelihood -------------------------------------------------------------------------
extern "C" {

int32_t pmx_loglik_device(const pmx_model* model, const pmx_population* cpop, const pmx_error_model* em,
                          const double* d_theta, int64_t n_support, double* d_ll, int64_t ld_ll, uint8_t* d_status,
                          void* stream) {
  g_err.clear();
  if (!model || !cpop || !em || !d_theta || !d_ll) return fail(PMX_ERR_INVALID_ARGUMENT, "null argument");
  if (n_support <= 0 || ld_ll < n_support) return fail(PMX_ERR_INVALID_ARGUMENT, "n_support must be > 0 and ld_ll >= n_support");
  pmx_population* pop = const_cast<pmx_population*>(cpop);
  DeviceGuard g;
  PMX_HIP(g.enter(pop->device));
  LLRequest req{em, d_ll, ld_ll};
  // pred is not written in log-likelihood mode; pass the ll buffer as a non-null placeholder
  return enqueue(model, pop, d_theta, n_support, 0, d_ll, ld_ll, d_status, stream, &req);
}

int32_t pmx_loglik(const pmx_model* model, const pmx_population* cpop, const pmx_error_model* em, const double* theta,
                   int64_t n_support, double* ll, int64_t ld_ll, uint8_t* status) {
  g_err.clear();
  if (!model || !cpop || !em || !theta || !ll) return fail(PMX_ERR_INVALID_ARGUMENT, "null argument");
  if (n_support <= 0 || ld_ll < n_support) return fail(PMX_ERR_INVALID_ARGUMENT, "n_support must be > 0 and ld_ll >= n_support");
  return loglik_host(model, cpop, em, theta, n_support, 0, ll, ld_ll, status);
}

int32_t pmx_loglik_batch(const pmx_model* model, const pmx_population* cpop, const pmx_error_model* em, const double* theta,
                         double* ll, uint8_t* status) {
  g_err.clear();
  if (!model || !cpop || !em || !theta || !ll) return fail(PMX_ERR_INVALID_ARGUMENT, "null argument");
  return loglik_host(model, cpop, em, theta, 1, 1, ll, 1, status);
}

int32_t pmx_loglik_batch_device(const pmx_model* model, const pmx_population* cpop, const pmx_error_model* em,
                                const double* d_theta, double* d_ll, uint8_t* d_status, void* stream) {
  g_err.clear();
  if (!model || !cpop || !em || !d_theta || !d_ll) return fail(PMX_ERR_INVALID_ARGUMENT, "null argument");
  pmx_population* pop = const_cast<pmx_population*>(cpop);
  DeviceGuard g;
  PMX_HIP(g.enter(pop->device));
  LLRequest req{em, d_ll, 1};
  return enqueue(model, pop, d_theta, 1, 1, d_ll, 1, d_status, stream, &req);
}

int64_t pmx_recommended_ld(int64_t n_support) { return n_support <= 0 ? 0 : (n_support + 15) / 16 * 16; }

int32_t pmx_measure_write_ceiling(double* d_buf, int64_t n_doubles, int32_t reps, void* stream, double* gb_per_s) {
  g_err.clear();
  if (!d_buf || !gb_per_s || n_doubles < 2 || reps < 1) return fail(PMX_ERR_INVALID_ARGUMENT, "bad argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  PMX_HIP(hipEventCreate(&e0));
  PMX_HIP(hipEventCreate(&e1));
  hipError_t e = hipSuccess;
  float best_ms = 0.0f;
  for (int shape = 0; shape < 4 && e == hipSuccess; ++shape) {  // the best of four store shapes (pmx_kernels.hip)
    e = pmx::launch_fill_linear(d_buf, n_doubles, 0.0, stream, shape);  // untimed
    if (e == hipSuccess) e = hipEventRecord(e0, st);
    for (int32_t i = 0; i < reps && e == hipSuccess; ++i) e = pmx::launch_fill_linear(d_buf, n_doubles, 0.0, stream, shape);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess && (shape == 0 || ms < best_ms)) best_ms = ms;
  }
  const float ms = best_ms;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (e != hipSuccess) return fail(PMX_ERR_HIP, hipGetErrorString(e));
  *gb_per_s = static_cast<double>(n_doubles / 2 * 16) * reps / (static_cast<double>(ms) * 1.0e-3) / 1.0e9;
  return PMX_OK;
}

int32_t pmx_host_alloc(int64_t bytes, void** out) {
  g_err.clear();
  if (!out || bytes < 0) return fail(PMX_ERR_INVALID_ARGUMENT, "bad argument");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(PMX_ERR_NO_DEVICE, "no HIP device visible");
  PMX_HIP(hipHostMalloc(out, static_cast<size_t>(bytes > 0 ? bytes : 1), hipHostMallocDefault));
  return PMX_OK;
}

void pmx_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

}  // extern "C"
