// pmx_devtypes.hpp — the structs the kernels receive by value (kernarg segment).  Shared by the library's own
// translation unit and by the sources hiprtc compiles at run time for user models (pmx_jit.cpp), so it must
// stay free of host-only headers.
#pragma once

#if !defined(__HIPCC_RTC__)
#include <cstdint>
#endif

#include "pmx.h"

namespace pmx {

// op kinds of the flattened per-subject stream (pmx_compile.hpp OpStream)
enum OpKind : uint32_t { OP_RESET = 0, OP_BOLUS = 1, OP_OBS = 2, OP_PROP = 3 };

// Model description as the kernels see it (passed by value in the kernarg segment).
struct DevModel {
  int32_t eq_kind, kernel;
  int32_t nparams, n_cov, n_derived, n_bind, nout, pm;
  int32_t has_init;
  int32_t state_override;  // >= 0: every output reads the raw amount of this state (Prediction::state, pmx_predict_state_device);
                           // read by the run-time-compiled walkers - the library's own kernels get their out[] rewritten
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
  int32_t out_vol_theta[PMX_MAX_OUT]; // theta index behind each output's volume when it is lane-constant (a primary
                                      // parameter, or a derived value without covariate factors: its base parameter); -1 = none
  double rk4_h_max;                   // ODE + lag: pieces split on the device recompute n = ceil(dt / h_max)
  double ode_rtol, ode_atol;          // adaptive solvers (PMX_SOLVER_DOPRI5, PMX_SOLVER_ROS2)
  int32_t ode_stiff;                  // the adaptive step is ROS2 (PMX_SOLVER_ROS2) instead of DOPRI5
  int32_t pad2_;
};
constexpr int kMaxLagSlots = 4;
// closure walkers (pmx_userlag.hpp): lagged boluses of one occasion whose landing times a lane keeps sorted in a private
// array; an occasion with more takes the model's PMX_USER_BIG_LISTS build (compiled on demand, pmx_api.cpp)
#ifndef PMX_USER_LAG_KEPT
#define PMX_USER_LAG_KEPT 64
#endif
constexpr int kUserLagKept = PMX_USER_LAG_KEPT;

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
  const double* op_rec;         // the ops packed for the PAIR kernels, whose lanes each fetch their own op: one record
                                //   instead of up to five scattered array reads.  ODE: [n_ops][6] = {meta | n << 32
                                //   (bits), a, b, rate[0], t0, t1}; analytical: [n_ops][4] = {meta (bits), a, b, t0}
  const double* op_fac;         // [n_ops*n_derived*PMX_MAX_FACTORS] covariate factors of the derived values (host-evaluated)
  const double* op_kfac;        // three-compartment covariate models (pmx_analytical_dyn3): one 64-byte record per op,
                                //   [n_ops][8] = {covariate factor of kernel parameter 0..6 (1.0 = none; a parameter's
                                //   factors multiplied out), op_a} - ONE wide scalar fetch per op, requested an op ahead
                                //   (nullptr = none)
  const double* op_t0;          // lag models: absolute start of each PROP / first event time of a RESET's occasion
  const double* op_t1;          // lag models: absolute end of each PROP
  const int64_t* lagb_off;      // [(n_occasions*n_lag_slots)+1]
  const double* lagb_time;
  const double* lagb_amount;
  // fused log-likelihood (pmx_loglik): nullptr = prediction mode
  const double* ll_obs;         // [n_observations][4] = {observed value, -0.5 ln(2pi) - ln(sigma), 1/(2 sigma^2),
                                //   censor scale: 0 | +1/(sigma sqrt 2) BLOQ | -1/(sigma sqrt 2) ALOQ};
                                //   weight 0 marks a missing observation (contributes nothing)
  double* ll_out;               // [n_subjects x ll_ld]
  int64_t ll_ld;
  int32_t n_rate;
  int32_t steps_per_trip;       // ODE PAIR kernel: RK4 steps (adaptive attempts) a lane takes per trip of its state machine
  int32_t n_cov;                // covariates per occasion (the segment tables below; custom ODE bodies read them)
  // covariate segments of every (occasion, covariate) cell, as HostPopulation holds them (covariate.rs:189-214)
  const int64_t* cov_seg_off;   // [n_occasions*n_cov + 1]
  const double* seg_from;
  const double* seg_to;
  const double* seg_slope;      // NaN = carry-forward segment
  const double* seg_icpt;
  const double* cov_first_t;    // [n_occasions*n_cov]
  const double* cov_first_v;
  const int32_t* lagb_input;    // user analytical models: the input of each listed bolus (one merged list per occasion)
};

}  // namespace pmx
