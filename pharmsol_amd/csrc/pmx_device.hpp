// pmx_device.hpp — device helpers shared by every kernel family (and by hiprtc-compiled user models).
#pragma once

#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#endif

#include "pmx_devtypes.hpp"

namespace pmx {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ int64_t uniform64(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32));
  return static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo);
}
__device__ __forceinline__ uint32_t uniform32(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// Read-only-for-the-launch data addressed with wave-uniform indices: a constant-address-space
// pointer lets the backend use the scalar unit (s_load_*) instead of 64 identical vector loads.
template <class T>
using cptr = const __attribute__((address_space(4))) T*;
template <class T>
__device__ __forceinline__ cptr<T> as_const(const T* p) {
  return (cptr<T>)(p);
}
__device__ __forceinline__ double uniformf64(double v) {
  return __longlong_as_double(uniform64(__double_as_longlong(v)));
}

// derive: derived[d] = ((theta[src] * f0) * f1), covariates as seen by the op

template <int N>
__device__ __forceinline__ double select_state(const double (&x)[N], int idx) {
  // The empty asm pins each element in a VGPR pair: without it LLVM rewrites the select chain into
  // ONE load from a selected address, which forces the whole state array out of registers (it was
  // promoted to LDS/scratch for every 3- and 4-state structure).
  double v = x[0];
#pragma unroll
  for (int i = 1; i < N; ++i) {
    double xi = x[i];
    asm("" : "+v"(xi));  // (NOT volatile: a side-effecting statement is a memory barrier to the compiler's alias analysis, and
                         // the run-time-compiled walkers - whose op-stream pointers are plain global ones - then fetch
                         // every wave-uniform word through the vector unit: 158 vector loads instead of 36, the user-closure
                         // workload 3.4 -> 3.9 ms)
    v = (idx == i) ? xi : v;
  }
  return v;
}

// Covariate c of occasion `occ` at time t: first segment with from <= t < to (linear: slope * t + intercept, two
// roundings like the reference; carry-forward: the stored value); before the first observation its value; the last
// segment is open-ended.  NaN when nothing matches (the reference's MissingSegments error).
__device__ __forceinline__ double cov_at(const DevOps& ops, int64_t occ, int c, double t) {
  const int64_t cell = occ * ops.n_cov + c;
  const int64_t s0 = as_const(ops.cov_seg_off)[cell], s1 = as_const(ops.cov_seg_off)[cell + 1];
  if (t < as_const(ops.cov_first_t)[cell]) return as_const(ops.cov_first_v)[cell];
  double v = __longlong_as_double(0x7ff8000000000000LL);
  for (int64_t sg = s0; sg < s1; ++sg) {
    if (as_const(ops.seg_from)[sg] <= t && t < as_const(ops.seg_to)[sg]) {
      const double sl = as_const(ops.seg_slope)[sg], ic = as_const(ops.seg_icpt)[sg];
      v = (sl != sl) ? ic : __dadd_rn(__dmul_rn(sl, t), ic);
      break;
    }
  }
  return v;
}

// ------------------------------------------------------------------------------------
// theta-dependent event rewrite on the device: lag time and bioavailability
// (Occasion::add_lagtime / add_bioavailability, src/data/structs.rs:611-666).
// Lagged boluses are NOT in the op stream; each lane merges them at t + lag(theta): a bolus that lands
// inside a PROP [t0, t1) splits it exactly where the reference's re-sorted event list would
// (solve(prev, tau), bolus, solve(tau, next)), a bolus that lands before the occasion's first remaining
// event opens the occasion.  At equal times an observation precedes the bolus (event.rs:292-304): a
// bolus landing exactly on an event time is applied at the START of the next PROP.
// ------------------------------------------------------------------------------------
struct LagState {
  double lag[kMaxLagSlots];
  int32_t cur[kMaxLagSlots];
  int32_t end[kMaxLagSlots];
};

// table[input] of a per-input model table for a lane-varying input: a select chain over the (scalar) entries, where
// indexing the kernel argument would be a vector load from memory
__device__ __forceinline__ int input_entry(const int32_t (&table)[PMX_MAX_INPUTS], int input) {
  int v = -1;
#pragma unroll
  for (int i = 0; i < PMX_MAX_INPUTS; ++i) v = (i == input) ? table[i] : v;
  return v;
}

__device__ __forceinline__ double fa_of(const DevModel& m, const double* __restrict__ th, int input) {
  double f = 1.0;
  if (m.has_fa) {  // wave-uniform; models without bioavailability never enter
    int fp = -1;
#pragma unroll
    for (int i = 0; i < PMX_MAX_INPUTS; ++i) fp = (i == input) ? m.fa_param[i] : fp;
    if (fp >= 0) f = th[fp];
  }
  return f;
}

// earliest pending lagged bolus: returns its landing time (inf if none) and slot
__device__ __forceinline__ double lag_next(const DevModel& m, const DevOps& ops, const LagState& ls, int& which) {
  double tau = __longlong_as_double(0x7ff0000000000000LL);
  which = -1;
#pragma unroll
  for (int k = 0; k < kMaxLagSlots; ++k) {
    if (k < m.n_lag_slots && ls.cur[k] < ls.end[k]) {
      const double tk = as_const(ops.lagb_time)[ls.cur[k]] + ls.lag[k];
      if (tk < tau) {
        tau = tk;
        which = k;
      }
    }
  }
  return tau;
}

template <int NS>
__device__ __forceinline__ void lag_apply_bolus(const DevModel& m, const DevOps& ops, LagState& ls, int which,
                                                const double* __restrict__ th, double (&x)[NS]);

// An observation that no PROP step precedes (OBS op bit 31, pmx_compile.cpp): the lagged boluses landing before its
// time are taken first, without propagation - the reference's solve does not advance over gaps below 1e-12 either.
template <int NS>
__device__ __forceinline__ void lag_flush_before(const DevModel& m, const DevOps& ops, LagState& ls, double t_obs,
                                                 const double* __restrict__ th, double (&x)[NS]) {
  for (;;) {
    int which;
    const double tau = lag_next(m, ops, ls, which);
    if (!(tau < t_obs)) break;
    lag_apply_bolus<NS>(m, ops, ls, which, th, x);
  }
}

template <int NS>
__device__ __forceinline__ void lag_apply_bolus(const DevModel& m, const DevOps& ops, LagState& ls, int which,
                                                const double* __restrict__ th, double (&x)[NS]) {
  int32_t idx = 0;
  int input = 0, dest = 0;
#pragma unroll
  for (int k = 0; k < kMaxLagSlots; ++k) {
    if (k == which) {
      idx = ls.cur[k];
      input = m.lag_input[k];
      dest = m.lag_dest[k];
      ls.cur[k] += 1;
    }
  }
  const double amt = as_const(ops.lagb_amount)[idx] * fa_of(m, th, input);
#pragma unroll
  for (int i = 0; i < NS; ++i) x[i] += (i == dest) ? amt : 0.0;
}

// One observation in log-likelihood mode: acc += lognormpdf(obs, y, sigma) with the sigma-only parts
// precomputed on the host (likelihood/distributions.rs:31-34; sigma from the observation,
// error_model.rs:1045-1080).  `q` = {obs, -0.5 ln(2 pi) - ln sigma, 1/(2 sigma^2), censor scale}.
// Censored rows (q[3] = +1/(sigma sqrt 2): BLOQ, -1/(sigma sqrt 2): ALOQ) take the log CDF / log survival
// function of distributions.rs:52-103, with statrs' Normal::cdf = 0.5 erfc((mean - x)/(sigma sqrt 2)); a tail
// that underflows falls back to the reference's asymptote (|z| > 37) or poisons the sum with NaN (its Err).
// censored row, out of line: erfc + two logs would otherwise sit in every log-likelihood kernel's register budget
// (the generic GRID kernel spilled 140 bytes and lost a wave of occupancy for a branch most datasets never take)
__device__ __noinline__ double ll_censored_term(double obs, double y, double pdf, double cs) {
  const double inv = fabs(cs);
  const double d = obs - y;
  const double cdf = 0.5 * erfc((y - obs) * inv);
  const double z = d * (inv * 1.4142135623730951);  // (obs - pred) / sigma
  const double nanv = __longlong_as_double(0x7ff8000000000000LL);
  if (cs > 0.0)  // BLOQ: ln P(X <= obs)
    return (cdf > 0.0) ? log(cdf) : ((z < -37.0) ? pdf - log(fabs(z)) : nanv);
  const double sf = 1.0 - cdf;  // ALOQ: ln P(X > obs), computed as 1 - cdf like the reference
  return (sf > 0.0) ? log(sf) : ((z > 37.0) ? pdf - log(z) : nanv);
}

// ResidualErrorModel::log_likelihood (src/data/residual_error.rs:178-191,265-271): sigma from the PREDICTION, floored at
// sqrt(f64::EPSILON).  Out of line like the censored term (rare: parametric-algorithm callers, log_likelihood_batch).
// kind = PMX_EM_RES_*; a = `scalar`, b = c[0].
__device__ __noinline__ double ll_residual_term(double obs, double y, double a, double b, int kind) {
  double raw = a;                                                     // constant | exponential
  if (kind == PMX_EM_RES_PROPORTIONAL) raw = a * fabs(y);             // b |f|  (b travels in `a`)
  if (kind == PMX_EM_RES_COMBINED) raw = sqrt(a * a + (b * b) * (y * y));
  const double sigma = fmax(raw, 1.4901161193847656e-08);
  const double z = (obs - y) / sigma;
  return -0.5 * (1.8378770664093453 + 2.0 * log(sigma) + z * z);
}

// (Q = const double* for per-lane rows, cptr<double> where the row is wave-uniform: scalar fetches)
template <class Q>
__device__ __forceinline__ void ll_accumulate(Q q, double y, double& acc) {
  const double w = q[2];
  if (w < 0.0) {  // residual (prediction-based) error model: {obs, a, -kind, b}
    acc += ll_residual_term(q[0], y, q[1], q[3], static_cast<int>(-w));
  } else if (w != 0.0) {  // weight 0 = missing observation: contributes 0 whatever the prediction (prediction.rs:107-111)
    const double d = q[0] - y;
    const double pdf = q[1] - (d * d) * w;
    const double cs = q[3];
    acc += (cs == 0.0) ? pdf : ll_censored_term(q[0], y, pdf, cs);
  }
}

}  // namespace
}  // namespace pmx
