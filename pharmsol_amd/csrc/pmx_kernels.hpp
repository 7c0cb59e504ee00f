// pmx_kernels.hpp — host <-> device launch contract (internal to libpmx_hip.so).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "pmx_compile.hpp"
#include "pmx_devtypes.hpp"

namespace pmx {

// Device mirror of a ClassPlan (pmx_compile.hpp).
struct DevClassPlan {
  const uint32_t* prog_meta;
  const double* prog_dt;
  const double* prog_t0;            // lag models: absolute start of a PROP step / first remaining event time of a RESET step
  const double* prog_t1;            // lag models: absolute end of a PROP step
  const int64_t* cls_prog_off;
  const int32_t* chunk_cls;
  const int32_t* chunk_n;
  const int64_t* chunk_val_off;
  const int32_t* chunk_subj;
  const int64_t* chunk_row;  // [n_chunks*G] first prediction row of each member (0 for padding)
  const double* val;
  const double* prog_rec;           // the programs again, packed for the log-likelihood kernel: [n_prog_steps + 1][2] =
                                    //   {meta (u64 bits), dt} - ONE scalar fetch per step, requested a step ahead
  const uint64_t* chunk_rate_mask;  // [n_chunks] ClassPlan::chunk_rate_mask
  const uint64_t* cls_fast_mask;    // [n_classes] ClassPlan::cls_fast_mask
  const uint32_t* chunk_hdr;        // [n_chunks + 1][16]: everything pmx_analytical_classed_ll needs of a chunk in ONE 64-byte
                                    //   record {n_live | n_steps << 16, program offset, val offset, cobs offset, rate mask (2),
                                    //   class fast mask (2), subject ids (8)} (the last record is padding; null: plan too large)
  const double* dtv;                // loose chunks: each member's own PROP lengths, laid out like val
  const double* facp;               // covariate models: [..][G][n_fac] covariate factors of each member's PROP ...
  const double* faco;               // ... and of the observation fused into the step
  int32_t n_fac;
  int32_t pad_;
  const double* cobs;               // log-likelihood mode: per chunk {[G] sum of the members' constant terms, [G] flags
                                    //   (slot 0: bit k = observation k is plain for every live member),
                                    //   [observation k][2][G] = observed value, weight} (pmx_ll_prepare_chunks)
  const int64_t* chunk_obs_off;     // [n_chunks] offset of the chunk's block in cobs
  const int32_t* generic_subjects;  // subjects the generic GRID kernel still has to walk
  int64_t n_chunks;
  int64_t n_chunks_exact;           // chunks [0, n_chunks_exact): shared step lengths; the rest: loose (pmx_compile.hpp)
  int64_t n_generic;
  int32_t G;
  int32_t zero_status;  // how the analytical GRID kernels own their status bytes (no memset precedes a launch):
                        // 1 = clear with 8-byte stores, then write failures only; 2 = write every pair's byte
};

// Fused per-subject step programs for the lean generic walker (pmx_analytical_steps): what the class plan builds per
// CLASS, built per SUBJECT - every OBS op folded into the step in front of it, one packed 32-byte record per step
// {meta (u64 bits: kind | io << 8 | obs-after << 24 | outeq << 25 | ladder rung << 27), a, b, 0}, one record of padding
// behind the last step (the walker requests step o + 1 while it works on step o).
struct DevSteps {
  const int64_t* subj_step_off;  // [S+1]
  const double* step_rec;        // [(n_steps + 1) * 4]
};

enum LaneMode : int32_t { MODE_GRID = 0, MODE_PAIR = 1 };

struct LaunchArgs {
  DevModel m;
  DevOps ops;
  const double* theta;
  int64_t P, S;
  double* pred;
  int64_t ld;
  uint8_t* status;
  int32_t mode;      // LaneMode
  int32_t batch;     // PAIR only: subject s uses theta row s
  int32_t s_chunk;   // GRID: subjects walked by one block
  int32_t n_ptiles;  // GRID: ceil(P / 256)
  int32_t dyn;       // analytical: kernel parameters depend on covariates (re-prepare per PROP)
  int32_t adaptive;     // ODE: PMX_SOLVER_DOPRI5
  int32_t ll_censored;  // log-likelihood mode: the population holds censored observations (classed kernel's CENS variant)
  int32_t use_classes;  // GRID analytical: run the classed kernel on cls.n_chunks, the generic one on the rest
  int32_t prop_slots;   // DYN GRID: LDS slots for kept propagators (OpStream::prop_cache_used; 0 = none)
  int32_t dyn_tile;     // DYN GRID with kept propagators: support points per block (0 = the default tile)
  int32_t no_rates;     // the compiled stream holds no PROP with an active infusion (three-compartment DYN: matrix-free walker)
  int32_t eig_reuse;    // ... and marks segments that repeat the previous built segment's rate constants (bit 27: the EIGR variant)
  int32_t tune_cpb;     // > 0: chunks per block of the classed kernel forced by PMX_TUNE_CPB (tuning experiments)
  int32_t tune_ll_old;  // != 0: PMX_TUNE_LL_OLD - the round-2 log-likelihood kernel for exact classes (A/B)
  DevClassPlan cls;
  DevSteps steps;       // analytical GRID, plain models (no covariate factors, no lag, no pm_ indexing): nullptr = none
  const int32_t* subj_list;  // GRID: walk these subjects instead of 0..S-1 (nullptr = all)
  int64_t n_list;
  void* stream;
};

// Sigma terms of every observation for one set of error models, computed on the device (AssayErrorModel::sigma,
// error_model.rs:1045-1080; the sigma-only parts of lognormpdf / lognormcdf, distributions.rs:31-103): fills
// obs4[n_obs][4] = {y, -0.5 ln(2 pi) - ln sigma, 1/(2 sigma^2), +-1/(sigma sqrt 2) | 0} and, for a class plan,
// the per-chunk [observation k][value | const | weight][G] blocks.  An invalid sigma (negative, non-finite, or not
// positive on a censored row) poisons that row with NaN and bumps *err.
struct LLPrepareArgs {
  const double* obs_y;        // [n_obs] observed value, NaN = missing
  const int32_t* obs_outeq;   // [n_obs]
  const double* obs_poly;     // [n_obs*4] or nullptr: the observation's own ErrorPoly (c0 NaN = none)
  const int8_t* obs_cens;     // [n_obs] or nullptr
  pmx_error_model em[PMX_MAX_OUT];
  int64_t n_obs;
  double* obs4;
  int32_t* err;
  // classed blocks (n_chunks == 0: none)
  const int64_t* chunk_row;
  const int32_t* chunk_n;
  const int32_t* chunk_nobs;
  const int64_t* chunk_obs_off;
  int64_t n_chunks;
  int32_t G;
  double* cobs;
  // the chunks' programs (flag slot 1 of a chunk's block = the plain-row mask indexed by program STEP)
  const int32_t* chunk_cls;
  const int64_t* cls_prog_off;
  const uint32_t* prog_meta;
  void* stream;
};
hipError_t launch_ll_prepare(const LLPrepareArgs& a);

// *d_flag |= 1 iff any of the n status bytes is non-zero (the host forms' "did any pair fail", without copying the array)
hipError_t launch_status_any(const uint8_t* d_status, int64_t n, int32_t* d_flag, void* stream);

// a linear streaming fill of n_doubles (pmx_measure_write_ceiling)
hipError_t launch_fill_linear(double* d_dst, int64_t n_doubles, double v, void* stream, int shape = 0);  // shape 0..2 (pmx_kernels.hip)

// Enqueue the prediction kernel; *name receives a static string naming the kernel family.
hipError_t launch_predict(const LaunchArgs& a, const char** name);

}  // namespace pmx
