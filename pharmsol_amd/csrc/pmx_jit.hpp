// pmx_jit.hpp — user ODE models compiled at run time for gfx950 with hiprtc.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/pmx.h"

namespace pmx {

struct JitSpec {
  int32_t nstates = 1, nparams = 1, nout = 1, ninputs = 1, ncov = 0;
  bool has_init = false;
  std::string source;  // definitions of pmx_dynamics / pmx_outputs (/ pmx_init), see include/pmx.h
  // user ANALYTICAL models (pmx_analytical.hpp): which closures `source` defines (PMX_FN_*), and the descriptor the
  // generator turns into code for the closures it leaves out
  bool analytical = false;
  // user ODE models whose lag / fa / derive are closures too, or whose dynamics takes the bolus vector
  // (pmx_ode_user.hpp): same two fields
  bool ode_user = false;
  uint32_t fns = 0;
  pmx_model_desc desc{};
  bool big_lists = false;  // closure walkers: compile the > 64-boluses-per-occasion path in (PMX_USER_BIG_LISTS, pmx_userlag.hpp)
};
enum JitKind { JIT_ODE = 0, JIT_ANALYTICAL = 1, JIT_ODE_USER = 2 };
inline JitKind jit_kind(const JitSpec& s) { return s.analytical ? JIT_ANALYTICAL : (s.ode_user ? JIT_ODE_USER : JIT_ODE); }

// The translation unit handed to hiprtc (user source + policy + the kernel wrappers: 16 for an ODE model - GRID/PAIR x
// lag x log-likelihood x solver -, 4 for an analytical one - GRID/PAIR x log-likelihood -, 8 for an ODE model with
// user lag / fa closures - GRID/PAIR x log-likelihood x solver).
std::string jit_translation_unit(const JitSpec& spec);

// Source text (pmx_derive) of an ANALYTICAL descriptor's derived values (desc.derived[]: theta * covariate factors), for
// descriptor models the library's own kernels do not take (lag time together with covariate-derived rate constants or
// pm_* indexing, more than four lagged inputs): such a model runs on the user-closure walker, every other closure
// generated from the descriptor as usual.
std::string analytical_descriptor_source(const pmx_model_desc& d);

// Source text (pmx_dynamics / pmx_outputs / pmx_init) of a BUILT-IN diffeq body whose parameters / volumes are derived
// from covariates through the descriptor (desc.derived[], desc.bind[]): the covariates are bound at the stage time.
std::string ode_descriptor_source(const pmx_model_desc& d);

// Compile for gfx950 (needs no device).  Returns true and fills *code (an AMDGPU code object); on failure *log has
// the compiler's diagnostics.
bool jit_compile(const JitSpec& spec, std::vector<char>* code, std::string* log);

struct JitModule {
  hipModule_t module = nullptr;
  hipFunction_t fn[2][2][2][2] = {};  // [mode: 0 GRID, 1 PAIR][LAG][LL][ADAPT]
};
// Load a compiled code object on the CURRENT device and resolve the kernel entry points.
hipError_t jit_load(const std::vector<char>& code, JitModule* out, JitKind kind = JIT_ODE);
void jit_unload(JitModule* m);

}  // namespace pmx
