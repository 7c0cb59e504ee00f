// pmx_kernels.hpp — host <-> device launch contract (internal to libpmx_hip.so).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/pmx.h"
#include "pmx_compile.hpp"

namespace pmx {

// Model description as the kernels see it (passed by value in the kernarg segment).
struct DevModel {
  int32_t eq_kind, kernel;
  int32_t nparams, n_cov, n_derived, n_bind, nout, pm;
  int32_t has_init;
  int32_t pad_;
  pmx_derived derived[PMX_MAX_DERIVED];
  pmx_bind bind[PMX_MAX_KPARAMS];
  pmx_out out[PMX_MAX_OUT];
  int32_t init_param[PMX_MAX_STATES];
  int32_t bolus_dest[PMX_MAX_INPUTS];
  int32_t infusion_dest[PMX_MAX_INPUTS];
  int32_t fa_param[PMX_MAX_INPUTS];   // bioavailability: amount *= theta[fa_param[input]]  (structs.rs:645-666)
  int32_t n_lag_slots;                // lagged inputs (<= kMaxLagSlots)
  int32_t has_fa;
  int32_t lag_input[4];               // slot -> input
  int32_t lag_param[4];               // slot -> theta index of the lag time
  int32_t lag_dest[4];                // slot -> state that receives the bolus
  double rk4_h_max;                   // ODE + lag: pieces split on the device recompute n = ceil(dt / h_max)
};
constexpr int kMaxLagSlots = 4;

// Device mirror of an OpStream (all pointers are device pointers).
struct DevOps {
  const int64_t* subj_op_off;   // [S+1]
  const int64_t* subj_obs_off;  // [S+1]
  const int32_t* subj_order;    // [S]
  const uint32_t* op_meta;      // [n_ops]
  const double* op_a;           // [n_ops]
  const double* op_b;           // [n_ops]
  const int32_t* op_n;          // [n_ops] (ODE)
  const double* op_rate;        // [n_ops*n_rate] (ODE)
  const double* op_cov;         // [n_ops*n_cov]
  const double* op_t0;          // lag models: absolute start of each PROP / first event time of a RESET's occasion
  const double* op_t1;          // lag models: absolute end of each PROP
  const int64_t* lagb_off;      // [(n_occasions*n_lag_slots)+1]
  const double* lagb_time;
  const double* lagb_amount;
  // fused log-likelihood (pmx_loglik): nullptr = prediction mode
  const double* ll_obs;         // [n_observations][4] = {observed value, -0.5 ln(2pi) - ln(sigma), 1/(2 sigma^2), 0};
                                //   weight 0 marks a missing observation (contributes nothing)
  double* ll_out;               // [n_subjects x ll_ld]
  int64_t ll_ld;
  int32_t n_rate;
  int32_t pad_;
};

// Device mirror of a ClassPlan (pmx_compile.hpp).
struct DevClassPlan {
  const uint32_t* prog_meta;
  const double* prog_dt;
  const int64_t* cls_prog_off;
  const int32_t* chunk_cls;
  const int32_t* chunk_n;
  const int64_t* chunk_val_off;
  const int32_t* chunk_subj;
  const int64_t* chunk_row;  // [n_chunks*G] first prediction row of each member (0 for padding)
  const double* val;
  const double* cobs;               // log-likelihood mode: per chunk [observation k][3][G] = value, const term, weight
  const int64_t* chunk_obs_off;     // [n_chunks] offset of the chunk's block in cobs
  const int32_t* generic_subjects;  // subjects the generic GRID kernel still has to walk
  int64_t n_chunks;
  int64_t n_generic;
  int32_t G;
  int32_t pad_;
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
  int32_t use_classes;  // GRID analytical: run the classed kernel on cls.n_chunks, the generic one on the rest
  DevClassPlan cls;
  const int32_t* subj_list;  // GRID: walk these subjects instead of 0..S-1 (nullptr = all)
  int64_t n_list;
  void* stream;
};

// Enqueue the prediction kernel; *name receives a static string naming the kernel family.
hipError_t launch_predict(const LaunchArgs& a, const char** name);

}  // namespace pmx
