// pmx_jit.hpp — user ODE models compiled at run time for gfx950 with hiprtc.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

namespace pmx {

struct JitSpec {
  int32_t nstates = 1, nparams = 1, nout = 1, ninputs = 1, ncov = 0;
  bool has_init = false;
  std::string source;  // definitions of pmx_dynamics / pmx_outputs (/ pmx_init), see include/pmx.h
};

// The translation unit handed to hiprtc (user source + policy + the 8 kernel wrappers).
std::string jit_translation_unit(const JitSpec& spec);

// Compile for gfx950 (needs no device).  Returns true and fills *code (an AMDGPU code object); on failure *log has
// the compiler's diagnostics.
bool jit_compile(const JitSpec& spec, std::vector<char>* code, std::string* log);

struct JitModule {
  hipModule_t module = nullptr;
  hipFunction_t fn[2][2][2][2] = {};  // [mode: 0 GRID, 1 PAIR][LAG][LL][ADAPT]
};
// Load a compiled code object on the CURRENT device and resolve the kernel entry points.
hipError_t jit_load(const std::vector<char>& code, JitModule* out);
void jit_unload(JitModule* m);

}  // namespace pmx
