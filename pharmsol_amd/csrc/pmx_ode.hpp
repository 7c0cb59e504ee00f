// pmx_ode.hpp — the ODE back-end's device code, generic over a model policy: fixed-step classic RK4 per constant-
// rate piece, the GRID and PAIR walkers (ode/mod.rs:609-823 semantics).  Included by
// pmx_kernels.hip with the built-in diffeq bodies and by the source hiprtc compiles for a user model (pmx_jit.cpp).
#pragma once

#include "pmx_device.hpp"

namespace pmx {
namespace {

// ------------------------------------------------------------------------------------
// ODE: built-in diffeq bodies + classic RK4 (fixed step per constant-rate piece)
// ------------------------------------------------------------------------------------
// Everything a lane needs besides its state.  `ops` / `occ` let a custom body read its covariates at any time t
// (device-side Covariate::interpolate over the occasion's segments, covariate.rs:216-241).
template <class M>
struct OdeLane {
  double kp[M::NP];
  double inv_vol[PMX_MAX_OUT];
  double xinit[M::NS];
  const DevOps* ops;
  int64_t occ;  // global occasion index of the occasion being walked
};

// Model policy M (a built-in diffeq body below, or the wrapper pmx_jit.cpp generates around a user's source):
//   NS, NP, CENTRAL, CUSTOM, NR (length of the rate vector: NS per-state rates for built-ins, the model's inputs
//   for custom bodies, which add rateiv themselves like a hand-written ODE::new closure)
//   built-in:  rhs(p, x, dx)                       autonomous; the walker adds the per-state rates
//   custom:    rhs(t, p, x, rateiv, dx), out(t, p, x, y), init(p, x); NOUT, HAS_INIT
template <class M>
__device__ __forceinline__ void ode_eval(double t, const OdeLane<M>& L, const double (&x)[M::NS], const double (&rs)[M::NR],
                                         double (&dx)[M::NS]) {
  if constexpr (M::CUSTOM) {
    if constexpr (M::NCOV > 0) {
      double cov[M::NCOV];
#pragma unroll
      for (int c = 0; c < M::NCOV; ++c) cov[c] = cov_at(*L.ops, L.occ, c, t);
      M::rhs(t, L.kp, x, rs, cov, dx);
    } else {
      M::rhs(t, L.kp, x, rs, nullptr, dx);
    }
  } else {
    M::rhs(L.kp, x, dx);
#pragma unroll
    for (int i = 0; i < M::NS; ++i) dx[i] += rs[i];
  }
}

// one classic RK4 step from t to t + h (rs constant over the piece)
template <class M>
__device__ __forceinline__ void rk4_step(const OdeLane<M>& L, double (&x)[M::NS], const double (&rs)[M::NR], double t,
                                         double h) {
  constexpr int NS = M::NS;
  double k1[NS], k2[NS], k3[NS], k4[NS], xt[NS];
  ode_eval<M>(t, L, x, rs, k1);
#pragma unroll
  for (int i = 0; i < NS; ++i) xt[i] = x[i] + (0.5 * h) * k1[i];
  ode_eval<M>(t + 0.5 * h, L, xt, rs, k2);
#pragma unroll
  for (int i = 0; i < NS; ++i) xt[i] = x[i] + (0.5 * h) * k2[i];
  ode_eval<M>(t + 0.5 * h, L, xt, rs, k3);
#pragma unroll
  for (int i = 0; i < NS; ++i) xt[i] = x[i] + h * k3[i];
  ode_eval<M>(t + h, L, xt, rs, k4);
#pragma unroll
  for (int i = 0; i < NS; ++i) x[i] = x[i] + (h / 6.0) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
}

// ---- adaptive: Dormand-Prince 5(4) ("dopri5" / ode45), PMX_SOLVER_DOPRI5 --------------------------------------
// One ATTEMPTED step of length h from (t, x): fills xn with the 5th-order solution and returns the scaled error
// norm rms(e_i / (atol + rtol max(|x_i|, |xn_i|))); the step is acceptable iff the result is <= 1.
template <class M>
__device__ __forceinline__ double dopri5_try(const DevModel& m, const OdeLane<M>& L, const double (&x)[M::NS],
                                             const double (&rs)[M::NR], double t, double h, double (&xn)[M::NS]) {
  constexpr int NS = M::NS;
  double k1[NS], k2[NS], k3[NS], k4[NS], k5[NS], k6[NS], k7[NS], xt[NS];
  ode_eval<M>(t, L, x, rs, k1);
#pragma unroll
  for (int i = 0; i < NS; ++i) xt[i] = x[i] + h * (0.2 * k1[i]);
  ode_eval<M>(t + 0.2 * h, L, xt, rs, k2);
#pragma unroll
  for (int i = 0; i < NS; ++i) xt[i] = x[i] + h * ((3.0 / 40.0) * k1[i] + (9.0 / 40.0) * k2[i]);
  ode_eval<M>(t + 0.3 * h, L, xt, rs, k3);
#pragma unroll
  for (int i = 0; i < NS; ++i) xt[i] = x[i] + h * ((44.0 / 45.0) * k1[i] - (56.0 / 15.0) * k2[i] + (32.0 / 9.0) * k3[i]);
  ode_eval<M>(t + 0.8 * h, L, xt, rs, k4);
#pragma unroll
  for (int i = 0; i < NS; ++i)
    xt[i] = x[i] + h * ((19372.0 / 6561.0) * k1[i] - (25360.0 / 2187.0) * k2[i] + (64448.0 / 6561.0) * k3[i] -
                        (212.0 / 729.0) * k4[i]);
  ode_eval<M>(t + (8.0 / 9.0) * h, L, xt, rs, k5);
#pragma unroll
  for (int i = 0; i < NS; ++i)
    xt[i] = x[i] + h * ((9017.0 / 3168.0) * k1[i] - (355.0 / 33.0) * k2[i] + (46732.0 / 5247.0) * k3[i] +
                        (49.0 / 176.0) * k4[i] - (5103.0 / 18656.0) * k5[i]);
  ode_eval<M>(t + h, L, xt, rs, k6);
#pragma unroll
  for (int i = 0; i < NS; ++i)
    xn[i] = x[i] + h * ((35.0 / 384.0) * k1[i] + (500.0 / 1113.0) * k3[i] + (125.0 / 192.0) * k4[i] -
                        (2187.0 / 6784.0) * k5[i] + (11.0 / 84.0) * k6[i]);
  ode_eval<M>(t + h, L, xn, rs, k7);
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const double e = h * ((71.0 / 57600.0) * k1[i] - (71.0 / 16695.0) * k3[i] + (71.0 / 1920.0) * k4[i] -
                          (17253.0 / 339200.0) * k5[i] + (22.0 / 525.0) * k6[i] - (1.0 / 40.0) * k7[i]);
    const double sc = m.ode_atol + m.ode_rtol * fmax(fabs(x[i]), fabs(xn[i]));
    const double q = e / sc;
    acc += q * q;
  }
  return sqrt(acc / static_cast<double>(NS));
}

// ---- stiff: ROS2, a linearly implicit (Rosenbrock) method, PMX_SOLVER_ROS2 ------------------------------------------
// The reference's default solver is diffsol's BDF - implicit, for stiff systems (ode/mod.rs:60-68); fast absorption or
// distribution next to slow elimination makes the explicit steppers crawl (DOPRI5 is stability-bound at h ~ 3.3 / |lambda|).
// ROS2 (Verwer, Spee, Blom, Hundsdorfer 1999), gamma = 1 + 1/sqrt(2), L-stable, second order for ANY approximation W of
// the Jacobian:
//     (I - gamma h J) k1 = f(t, y) + gamma h f_t
//     (I - gamma h J) k2 = f(t + h, y + h k1) - gamma h f_t - 2 k1
//     y+ = y + 3/2 h k1 + 1/2 h k2,     error estimate = y+ - (y + h k1) = h/2 (k1 + k2)   (the embedded first-order solution)
// One lane = one system of NS <= 8 states: J and f_t by forward differences (NS + 1 extra right-hand sides; exact up to
// rounding for the linear compartmental bodies), the NS x NS factorisation fully unrolled in registers WITHOUT pivoting
// (I - gamma h J of a compartmental system is a column-diagonally-dominant M-matrix, for which elimination in the natural
// order is stable).  Same try/advance contract as DOPRI5, error exponent -1/2.
template <class M>
__device__ __forceinline__ double ros2_try(const DevModel& m, const OdeLane<M>& L, const double (&x)[M::NS],
                                           const double (&rs)[M::NR], double t, double h, double (&xn)[M::NS]) {
  constexpr int NS = M::NS;
  constexpr double kGamma = 1.7071067811865475;
  constexpr double kSqrtEps = 1.4901161193847656e-08;
  const double gh = kGamma * h;
  double f0[NS], f1[NS], xt[NS], W[NS][NS], ft[NS], k1[NS], k2[NS];
  ode_eval<M>(t, L, x, rs, f0);
#pragma unroll
  for (int j = 0; j < NS; ++j) {
#pragma unroll
    for (int i = 0; i < NS; ++i) xt[i] = x[i];
    const double d = kSqrtEps * fmax(fabs(x[j]), 1.0);
    xt[j] = x[j] + d;
    ode_eval<M>(t, L, xt, rs, f1);
    const double s = -gh / d;
#pragma unroll
    for (int i = 0; i < NS; ++i) W[i][j] = (f1[i] - f0[i]) * s + ((i == j) ? 1.0 : 0.0);
  }
  {
    const double dt = kSqrtEps * fmax(fabs(t), 1.0);
    ode_eval<M>(t + dt, L, x, rs, f1);
    const double s = gh / dt;
#pragma unroll
    for (int i = 0; i < NS; ++i) ft[i] = (f1[i] - f0[i]) * s;
  }
  // W = L U in place (unit lower triangle below the diagonal), natural order
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const double inv = 1.0 / W[k][k];
#pragma unroll
    for (int i = k + 1; i < NS; ++i) {
      const double l = W[i][k] * inv;
      W[i][k] = l;
#pragma unroll
      for (int j = k + 1; j < NS; ++j) W[i][j] -= l * W[k][j];
    }
    W[k][k] = inv;  // (the diagonal keeps its reciprocal for the back substitutions)
  }
  auto solve = [&W](double (&b)[NS]) {
#pragma unroll
    for (int i = 1; i < NS; ++i) {
#pragma unroll
      for (int j = 0; j < i; ++j) b[i] -= W[i][j] * b[j];
    }
#pragma unroll
    for (int i = NS - 1; i >= 0; --i) {
#pragma unroll
      for (int j = i + 1; j < NS; ++j) b[i] -= W[i][j] * b[j];
      b[i] *= W[i][i];
    }
  };
#pragma unroll
  for (int i = 0; i < NS; ++i) k1[i] = f0[i] + ft[i];
  solve(k1);
#pragma unroll
  for (int i = 0; i < NS; ++i) xt[i] = x[i] + h * k1[i];
  ode_eval<M>(t + h, L, xt, rs, f1);
#pragma unroll
  for (int i = 0; i < NS; ++i) k2[i] = f1[i] - ft[i] - 2.0 * k1[i];
  solve(k2);
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    xn[i] = x[i] + h * (1.5 * k1[i] + 0.5 * k2[i]);
    const double e = (0.5 * h) * (k1[i] + k2[i]);
    const double sc = m.ode_atol + m.ode_rtol * fmax(fabs(x[i]), fabs(xn[i]));
    const double q = e / sc;
    acc += q * q;
  }
  return sqrt(acc / static_cast<double>(NS));
}

// Step-size controller state of a lane: `h` = the controller's current proposal (carried from piece to piece).
struct AdaptState {
  double h;
  uint8_t failed;
};

// One attempt inside the piece [.., t1]: tries min(h, t1 - t, h_max); on acceptance advances (t, x).  Returns true
// while the piece is unfinished.  A step that underflows (h < 1e-13 max(1,|t|)) marks the lane failed and jumps to
// the end of the piece so that every lane terminates.
template <class M>
__device__ __forceinline__ bool dopri5_advance(const DevModel& m, const OdeLane<M>& L, double (&x)[M::NS],
                                               const double (&rs)[M::NR], double& t, double t1, AdaptState& as) {
  constexpr int NS = M::NS;
  const double left = t1 - t;
  if (!(left > 0.0)) return false;
  double h = fmin(as.h, m.rk4_h_max);
  const bool clipped = h >= left;
  if (clipped) h = left;
  double xn[NS];
  const bool stiff = m.ode_stiff != 0;  // (wave-uniform: PMX_SOLVER_ROS2)
  const double err = stiff ? ros2_try<M>(m, L, x, rs, t, h, xn) : dopri5_try<M>(m, L, x, rs, t, h, xn);
  const bool ok = err <= 1.0;  // (false for NaN)
  // factor 0.9 err^(-1/(p+1)) in [0.2, 5], p = the order of the error estimate (4 | 1); no growth right after a rejection
  double fac = (err > 0.0) ? 0.9 * pow(err, stiff ? -0.5 : -0.2) : 5.0;
  if (!(fac >= 0.2)) fac = 0.2;  // also catches NaN
  if (fac > 5.0) fac = 5.0;
  if (!ok && fac > 1.0) fac = 1.0;
  const double h_next = h * fac;
  if (ok) {
#pragma unroll
    for (int i = 0; i < NS; ++i) x[i] = xn[i];
    t = clipped ? t1 : (t + h);
    as.h = clipped ? fmax(as.h, h_next) : h_next;  // a step cut short by the piece end must not shrink the proposal
    return !clipped;
  }
  as.h = h_next;
  if (!(h_next > 1.0e-13 * fmax(1.0, fabs(t)))) {  // step-size underflow (or NaN): give up on this piece
    as.failed = 1;
    t = t1;
    return false;
  }
  return true;
}

template <class M>
__device__ __forceinline__ void ode_lane_setup(const DevModel& m, const double* __restrict__ th, OdeLane<M>& L) {
#pragma unroll
  for (int j = 0; j < M::NP; ++j) L.kp[j] = th[j];
  if constexpr (M::CUSTOM) {
#pragma unroll
    for (int o = 0; o < PMX_MAX_OUT; ++o) L.inv_vol[o] = 1.0;
#pragma unroll
    for (int i = 0; i < M::NS; ++i) L.xinit[i] = 0.0;  // (custom bodies initialise at RESET: ode_reset)
    return;
  }
#pragma unroll
  for (int o = 0; o < PMX_MAX_OUT; ++o) {
    double v = 1.0;
    if (o < m.nout && m.out[o].vol_src == PMX_SRC_PRIMARY) v = th[m.out[o].vol_index];
    L.inv_vol[o] = 1.0 / v;
  }
#pragma unroll
  for (int i = 0; i < M::NS; ++i) L.xinit[i] = (m.has_init && m.init_param[i] >= 0) ? th[m.init_param[i]] : 0.0;
}

// RESET: x = 0, plus the model's init for an occasion of index 0 (`with_init`; ode/mod.rs:536-549).  A custom body
// is evaluated here, per subject, because it may read covariates (at t = 0).
template <class M>
__device__ __forceinline__ void ode_reset(const OdeLane<M>& L, int with_init, double (&x)[M::NS]) {
#pragma unroll
  for (int i = 0; i < M::NS; ++i) x[i] = with_init ? L.xinit[i] : 0.0;
  if constexpr (M::CUSTOM) {
    if constexpr (M::HAS_INIT) {
      if (with_init) {
        if constexpr (M::NCOV > 0) {
          double cov[M::NCOV];
#pragma unroll
          for (int c = 0; c < M::NCOV; ++c) cov[c] = cov_at(*L.ops, L.occ, c, 0.0);
          M::init(L.kp, cov, x);
        } else {
          M::init(L.kp, nullptr, x);
        }
      }
    }
  }
}

template <class M>
__device__ __forceinline__ double ode_out(const DevModel& m, const OdeLane<M>& L,
                                          const double (&x)[M::NS], int outeq, double t) {
  if constexpr (M::CUSTOM) {
    if (m.state_override >= 0) return select_state<M::NS>(x, m.state_override);  // (wave-uniform)
    double y[M::NOUT];
#pragma unroll
    for (int o = 0; o < M::NOUT; ++o) y[o] = 0.0;
    if constexpr (M::NCOV > 0) {
      double cov[M::NCOV];
#pragma unroll
      for (int c = 0; c < M::NCOV; ++c) cov[c] = cov_at(*L.ops, L.occ, c, t);
      M::out(t, L.kp, x, cov, y);
    } else {
      M::out(t, L.kp, x, nullptr, y);
    }
    double v = y[0];
#pragma unroll
    for (int o = 1; o < M::NOUT; ++o) v = (o == outeq) ? y[o] : v;
    return v;
  }
  int state = 0;
  double inv = 1.0;
#pragma unroll
  for (int o = 0; o < PMX_MAX_OUT; ++o) {
    if (o == outeq) {
      state = m.out[o].state;
      inv = L.inv_vol[o];
    }
  }
  return select_state<M::NS>(x, state) * inv;
}

// per-state rate vector of a PROP op: dx[dest(input)] += rateiv[input]  (expand/ode.rs:380-406)
template <class M>
__device__ __forceinline__ void ode_rates(const DevModel& m, const double* __restrict__ op_rate, int64_t o, int n_rate,
                                          double (&rs)[M::NR]) {
  if constexpr (M::CUSTOM) {  // rateiv[input], handed to the user's body as it is
#pragma unroll
    for (int k = 0; k < M::NR; ++k) rs[k] = (k < n_rate) ? op_rate[o * n_rate + k] : 0.0;
    return;
  }
#pragma unroll
  for (int i = 0; i < M::NS; ++i) rs[i] = 0.0;
  for (int k = 0; k < n_rate; ++k) {
    const double r = op_rate[o * n_rate + k];
    const int dest = (m.infusion_dest[k] >= 0) ? m.infusion_dest[k] : M::CENTRAL;
#pragma unroll
    for (int i = 0; i < M::NS; ++i) rs[i] += (i == dest) ? r : 0.0;
  }
}
// ... with the first input's rate (r0 = op_rate[o * n_rate]) already in a register
template <class M>
__device__ __forceinline__ void ode_rates(const DevModel& m, const double* __restrict__ op_rate, int64_t o, int n_rate,
                                          double r0, double (&rs)[M::NR]) {
  if constexpr (M::CUSTOM) {
#pragma unroll
    for (int k = 0; k < M::NR; ++k) rs[k] = (k == 0) ? ((n_rate > 0) ? r0 : 0.0) : ((k < n_rate) ? op_rate[o * n_rate + k] : 0.0);
    return;
  }
#pragma unroll
  for (int i = 0; i < M::NS; ++i) rs[i] = 0.0;
  for (int k = 0; k < n_rate; ++k) {
    const double r = (k == 0) ? r0 : op_rate[o * n_rate + k];
    const int dest = (m.infusion_dest[k] >= 0) ? m.infusion_dest[k] : M::CENTRAL;
#pragma unroll
    for (int i = 0; i < M::NS; ++i) rs[i] += (i == dest) ? r : 0.0;
  }
}

// One constant-rate piece [t0, t1] whose length is only known on the device (a lagged bolus split it):
// n = ceil(dt / h_max) classic RK4 steps, the host compiler's rule (pmx_compile.cpp, ODE PROP ops).
template <class M, bool ADAPT>
__device__ __forceinline__ void ode_piece(const DevModel& m, const OdeLane<M>& L, double (&x)[M::NS],
                                          const double (&rs)[M::NR], double t0, double t1, AdaptState& as) {
  const double dt = t1 - t0;
  if (!(dt > 0.0)) return;
  if constexpr (ADAPT) {
    double t = t0;
    for (int32_t guard = 0; guard < 10000000 && dopri5_advance<M>(m, L, x, rs, t, t1, as); ++guard) {
    }
    return;
  }
  double nf = ceil(dt / m.rk4_h_max);
  if (!(nf >= 1.0)) nf = 1.0;
  if (nf > 1.0e7) nf = 1.0e7;  // a lane with an absurd lag must still terminate
  const int32_t n = static_cast<int32_t>(nf);
  const double h = dt / static_cast<double>(n);
  for (int32_t k = 0; k < n; ++k) rk4_step<M>(L, x, rs, t0 + static_cast<double>(k) * h, h);
}

// A lagged bolus of slot `which` lands exactly on the time of the occasion's first remaining event (kind k_first): is it
// in front of that event in the re-sorted list?  Observation < Bolus < Infusion at equal times, and the sort is stable
// (event.rs:292-304, structs.rs:669-671): against another bolus it keeps its recorded place - earlier iff its lag is > 0.
__device__ __forceinline__ bool lag_lands_first(const LagState& ls, int which, uint32_t k_first) {
  double lg = 0.0;
#pragma unroll
  for (int k = 0; k < kMaxLagSlots; ++k) lg = (k == which) ? ls.lag[k] : lg;
  return k_first == PMX_EV_INFUSION || (k_first == PMX_EV_BOLUS && lg > 0.0);
}

// lag_open_occasion / lag_prop of the ODE back-end: same merge rule, RK4 pieces instead of closed forms.
//
// The solver clock (ode/mod.rs:348,719-721): the reference's solver starts at the occasion's RECORDED initial time
// `t_rec` (Occasion::initial_time, lagged boluses at their recorded times included) and only advances
// `while next_event_time > solver.state().t`.  The first event of the lag-rewritten list is applied at that clock
// without integration; every later event is reached by integrating from the clock if its time lies ahead.  Between an
// early lagged bolus and the occasion's first remaining event no infusion can be active (infusions are events of the
// occasion), so those pieces run with zero rates.  Returns the clock on arrival at the first remaining event.
template <class M, bool ADAPT>
__device__ __forceinline__ double ode_lag_open_occasion(const DevModel& m, const DevOps& ops, LagState& ls, int64_t occ,
                                                        double t_first, uint32_t k_first, double t_rec, const OdeLane<M>& L,
                                                        const double* __restrict__ th, double (&x)[M::NS], AdaptState& as) {
  constexpr int NS = M::NS;
#pragma unroll
  for (int k = 0; k < kMaxLagSlots; ++k) {
    if (k < m.n_lag_slots) {
      ls.cur[k] = static_cast<int32_t>(as_const(ops.lagb_off)[occ * m.n_lag_slots + k]);
      ls.end[k] = static_cast<int32_t>(as_const(ops.lagb_off)[occ * m.n_lag_slots + k + 1]);
    } else {
      ls.cur[k] = ls.end[k] = 0;
    }
  }
  double zero[M::NR];
#pragma unroll
  for (int i = 0; i < M::NR; ++i) zero[i] = 0.0;
  bool first = true;
  double clk = t_rec;
  for (;;) {
    int which;
    const double tau = lag_next(m, ops, ls, which);
    if (!(tau < t_first)) break;
    if (!first && tau > clk) {
      ode_piece<M, ADAPT>(m, L, x, zero, clk, tau, as);
      clk = tau;
    }
    first = false;
    lag_apply_bolus<NS>(m, ops, ls, which, th, x);
  }
  if (first) {  // a bolus landing exactly ON the first remaining event's time may still be the first event of the list
    int which;
    const double tau = lag_next(m, ops, ls, which);
    if (tau == t_first && lag_lands_first(ls, which, k_first)) {
      first = false;
      lag_apply_bolus<NS>(m, ops, ls, which, th, x);
    }
  }
  if (!first && t_first > clk && t_first < __longlong_as_double(0x7ff0000000000000LL)) {
    ode_piece<M, ADAPT>(m, L, x, zero, clk, t_first, as);
    clk = t_first;
  }
  return clk;
}

template <class M, bool ADAPT>
__device__ __forceinline__ void ode_lag_prop(const DevModel& m, const DevOps& ops, LagState& ls, double t0, double t1,
                                             const OdeLane<M>& L, const double (&rs)[M::NR],
                                             const double* __restrict__ th, double (&x)[M::NS], AdaptState& as) {
  constexpr int NS = M::NS;
  double t = t0;
  for (;;) {
    int which;
    const double tau = lag_next(m, ops, ls, which);
    if (!(tau < t1)) break;
    if (tau > t) {
      ode_piece<M, ADAPT>(m, L, x, rs, t, tau, as);
      t = tau;
    }
    lag_apply_bolus<NS>(m, ops, ls, which, th, x);
  }
  ode_piece<M, ADAPT>(m, L, x, rs, t, t1, as);
}

template <class M, bool LAG, bool LL, bool ADAPT>
__device__ __forceinline__ void ode_grid_body(const DevModel& m, const DevOps& ops, const double* __restrict__ theta,
                                              int64_t P, int64_t S, int32_t s_chunk, int32_t n_ptiles,
                                              double* __restrict__ pred, int64_t ld, uint8_t* __restrict__ status) {
  constexpr int NS = M::NS;
  const int64_t b = blockIdx.x;
  const int32_t ptile = static_cast<int32_t>(b % n_ptiles);
  const int64_t chunk = b / n_ptiles;
  const int64_t p = static_cast<int64_t>(ptile) * kBlock + threadIdx.x;
  const bool lane_ok = p < P;
  const int64_t pc = lane_ok ? p : (P - 1);
  const double* __restrict__ th = theta + pc * m.nparams;
  OdeLane<M> L;
  ode_lane_setup<M>(m, th, L);
  L.ops = &ops;
  L.occ = 0;
  uint8_t st_lane = PMX_PAIR_OK;
  LagState ls;
  if constexpr (LAG) {
#pragma unroll
    for (int k = 0; k < kMaxLagSlots; ++k) {
      ls.lag[k] = (k < m.n_lag_slots) ? th[m.lag_param[k]] : 0.0;
      ls.cur[k] = ls.end[k] = 0;
      if (k < m.n_lag_slots && ls.lag[k] != ls.lag[k]) {  // NaN; a negative lag shifts the bolus earlier (structs.rs:629-634)
        st_lane = PMX_PAIR_BAD_LAG;
        ls.lag[k] = 0.0;  // keep the walk finite; every output of this lane is NaN anyway
      }
    }
  }
  const double nanv = __longlong_as_double(0x7ff8000000000000LL);
  const int64_t s_begin = chunk * s_chunk;
  const int64_t s_end = (s_begin + s_chunk < S) ? (s_begin + s_chunk) : S;
  for (int64_t s = s_begin; s < s_end; ++s) {
    const int64_t o0 = uniform64(as_const(ops.subj_op_off)[s]);
    const int64_t o1 = uniform64(as_const(ops.subj_op_off)[s + 1]);
    int64_t row = uniform64(as_const(ops.subj_obs_off)[s]);
    double x[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) x[i] = 0.0;
    uint8_t st = st_lane;
    double ll_acc = 0.0;
    AdaptState as;  // adaptive solver: the step-size proposal restarts with every subject
    as.h = m.rk4_h_max;
    as.failed = 0;
    double clk = 0.0;  // LAG: the lane's solver clock (see ode_lag_open_occasion)
    for (int64_t o = o0; o < o1; ++o) {
      const uint32_t meta = uniform32(as_const(ops.op_meta)[o]);
      const uint32_t kind = meta & 0xffu;
      const int io = static_cast<int>((meta >> 8) & 0xffffu);
      const double a = uniformf64(as_const(ops.op_a)[o]);
      if (kind == OP_PROP) {
        double rs[M::NR];
        ode_rates<M>(m, ops.op_rate, o, ops.n_rate, rs);
        if constexpr (LAG) {
          const double t0 = uniformf64(as_const(ops.op_t0)[o]), t1 = uniformf64(as_const(ops.op_t1)[o]);
          ode_lag_prop<M, ADAPT>(m, ops, ls, (clk > t0) ? clk : t0, t1, L, rs, th, x, as);
          if (t1 > clk) clk = t1;
        } else if constexpr (ADAPT) {
          ode_piece<M, true>(m, L, x, rs, uniformf64(as_const(ops.op_t0)[o]), uniformf64(as_const(ops.op_t1)[o]), as);
        } else {
          const double h = uniformf64(as_const(ops.op_b)[o]);
          const int32_t n = static_cast<int32_t>(uniform32(static_cast<uint32_t>(as_const(ops.op_n)[o])));
          double t0 = 0.0;  // only a custom (possibly non-autonomous) body reads the time
          if constexpr (M::CUSTOM) t0 = uniformf64(as_const(ops.op_t0)[o]);
          for (int32_t k = 0; k < n; ++k) rk4_step<M>(L, x, rs, t0 + static_cast<double>(k) * h, h);
        }
      } else if (kind == OP_OBS) {
        double y = ode_out<M>(m, L, x, io, a);
        if (LAG && st == PMX_PAIR_BAD_LAG) y = nanv;
        if (ADAPT && as.failed) {  // step-size underflow somewhere before this row
          if (st == PMX_PAIR_OK) st = PMX_PAIR_SOLVER_FAIL;
          y = nanv;
        }
        if constexpr (LL) {
          ll_accumulate(as_const(ops.ll_obs) + row * 4, y, ll_acc);
        } else {
          if (st == PMX_PAIR_OK && !isfinite(y)) st = PMX_PAIR_NONFINITE;
          if (lane_ok) pred[row * ld + p] = y;
        }
        ++row;
      } else if (kind == OP_BOLUS) {
        const int bd = input_entry(m.bolus_dest, io);
        const int dest = (bd >= 0) ? bd : io;
        const double amt = a * fa_of(m, th, io);
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] += (i == dest) ? amt : 0.0;
      } else {
        L.occ = static_cast<int64_t>(a);
        ode_reset<M>(L, io, x);
        if constexpr (LAG)
          clk = ode_lag_open_occasion<M, ADAPT>(m, ops, ls, static_cast<int64_t>(a), uniformf64(as_const(ops.op_t0)[o]),
                                                (meta >> 25) & 3u, uniformf64(as_const(ops.op_b)[o]), L, th, x, as);
      }
    }
    if constexpr (LL) {
      if (st == PMX_PAIR_OK && !isfinite(ll_acc)) st = PMX_PAIR_NONFINITE;
      if (lane_ok) ops.ll_out[s * ops.ll_ld + p] = (st == PMX_PAIR_OK || st == PMX_PAIR_NONFINITE) ? ll_acc : nanv;
    }
    if (status != nullptr && lane_ok) status[s * P + p] = st;  // every pair writes its byte: no memset before the launch
  }
}

// PAIR: each lane is a small state machine {cursor o, remaining RK4 steps, ring of fetched ops}.  One trip of the
// wave loop = a top-up of the rings when a lane ran dry, an op phase (idle lanes take up to two ops) and a stepping
// phase (lanes inside a piece take up to ops.steps_per_trip steps), so lanes in different segments of different
// subjects advance together (divergent timelines, C4) and none waits longer than a bounded number of steps.
template <class M, bool LAG, bool LL, bool ADAPT>
__device__ __forceinline__ void ode_pair_body(const DevModel& m, const DevOps& ops, const double* __restrict__ theta,
                                              int64_t P, int64_t S, int32_t batch, double* __restrict__ pred,
                                              int64_t ld, uint8_t* __restrict__ status) {
  constexpr int NS = M::NS;
  const int64_t n_pairs = batch ? S : S * P;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const bool lane_ok = i < n_pairs;
  const int64_t ic = lane_ok ? i : (n_pairs - 1);
  const int64_t s = as_const(ops.subj_order)[batch ? ic : (ic / P)];
  const int64_t p = batch ? 0 : (ic % P);
  const double* __restrict__ th = theta + (batch ? s : p) * m.nparams;
  OdeLane<M> L;
  ode_lane_setup<M>(m, th, L);
  L.ops = &ops;
  L.occ = 0;
  uint8_t st = PMX_PAIR_OK;
  LagState ls;
  if constexpr (LAG) {
#pragma unroll
    for (int k = 0; k < kMaxLagSlots; ++k) {
      ls.lag[k] = (k < m.n_lag_slots) ? th[m.lag_param[k]] : 0.0;
      ls.cur[k] = ls.end[k] = 0;
      if (k < m.n_lag_slots && ls.lag[k] != ls.lag[k]) {  // NaN; a negative lag shifts the bolus earlier (structs.rs:629-634)
        st = PMX_PAIR_BAD_LAG;
        ls.lag[k] = 0.0;
      }
    }
  }
  const double nanv = __longlong_as_double(0x7ff8000000000000LL);
  const double inf = __longlong_as_double(0x7ff0000000000000LL);

  int64_t o = as_const(ops.subj_op_off)[s];
  const int64_t o1 = lane_ok ? as_const(ops.subj_op_off)[s + 1] : o;
  int64_t row = as_const(ops.subj_obs_off)[s];
  // The lane's next ops wait in LDS: a ring of kRing packed records per lane (DevOps::op_rec), topped up for EVERY
  // lane of the wave whenever one lane runs dry.  Lanes consume their ops at their own pace, so without the ring
  // nearly every trip had some lane waiting for a gather from HBM/L2 (~0.7 us with one wave per SIMD) and the whole
  // wave with it; with it a wave waits for memory a handful of times per kernel and an op costs an LDS read.
  constexpr int kRing = 4;
  constexpr int kOpsPerTrip = 2;  // (3 or 4 measure the same on C4)
  constexpr int kParts = (LAG || ADAPT || M::CUSTOM) ? 3 : 2;  // 16-byte parts of a record this variant reads
  __shared__ double2 ring[kRing * kParts * kBlock];
  int64_t filled = o;  // ops [o, filled) are in the ring, op k in slot k % kRing
  double x[NS], rs[M::NR];
#pragma unroll
  for (int k = 0; k < NS; ++k) x[k] = 0.0;
#pragma unroll
  for (int k = 0; k < M::NR; ++k) rs[k] = 0.0;
  int32_t rem = 0;
  double h = 0.0;
  // custom bodies may read the time: the open piece's start and step count (stage time = t_piece + k h)
  double t_piece = 0.0;
  int32_t n_piece = 0;
  // adaptive solver: the running piece [t_run, t_run_end] and the lane's step-size proposal
  bool stepping = false;
  double t_run = 0.0, t_run_end = 0.0;
  AdaptState as;
  as.h = m.rk4_h_max;
  as.failed = 0;
  double ll_acc = 0.0;
  // LAG: an open PROP (or occasion opening) [t_cur, t_stop) that lagged boluses may still split
  bool in_prop = false;
  double t_cur = 0.0, t_stop = 0.0;
  double clk = 0.0;  // LAG: the lane's solver clock (see ode_lag_open_occasion)
  // Every phase is bounded and the op phase is a single if / else chain per action: with `continue`s the compiler
  // rotated the stepping branch into an unbounded inner per-lane loop and lanes that needed an op waited for the
  // longest piece in the wave (measured on the first version of this loop: C4 2.4 -> 3.9 ms).
  while ((ADAPT ? stepping : rem > 0) || o < o1) {
    {
      const bool busy = (ADAPT ? stepping : rem > 0) || (LAG && in_prop);
      const bool dry = !busy && o < o1 && o == filled;
      if (__ballot(dry) != 0ull) {  // wave-uniform: every lane fills its free slots, all gathers in flight together
        const double2* __restrict__ recs = reinterpret_cast<const double2*>(ops.op_rec);
        const int64_t stop = (o + kRing < o1) ? (o + kRing) : o1;
        // (named values and clamped addresses instead of an array under per-lane conditions: that form ended up in
        // scratch memory; a lane with nothing to fetch re-reads a record it already holds, or op 0 of the stream)
        static_assert(kRing == 4, "the top-up below is written out for four slots");
        const int64_t k0 = filled, k1 = filled + 1, k2 = filled + 2, k3 = filled + 3;
        const int64_t safe = (stop > 0) ? (stop - 1) : 0;
        const int64_t c0 = (k0 < stop) ? k0 : safe, c1 = (k1 < stop) ? k1 : safe, c2 = (k2 < stop) ? k2 : safe,
                      c3 = (k3 < stop) ? k3 : safe;
        const double2 a0 = recs[c0 * 3], b0 = recs[c0 * 3 + 1];
        const double2 a1 = recs[c1 * 3], b1 = recs[c1 * 3 + 1];
        const double2 a2 = recs[c2 * 3], b2 = recs[c2 * 3 + 1];
        const double2 a3 = recs[c3 * 3], b3 = recs[c3 * 3 + 1];
        double2 e0 = a0, e1 = a0, e2 = a0, e3 = a0;
        if constexpr (kParts == 3) {
          e0 = recs[c0 * 3 + 2];
          e1 = recs[c1 * 3 + 2];
          e2 = recs[c2 * 3 + 2];
          e3 = recs[c3 * 3 + 2];
        }
        const auto stash = [&](const double2& ra, const double2& rb, const double2& rc, int64_t k) {
          if (k < stop) {
            const int slot = static_cast<int>(k & (kRing - 1));
            ring[(slot * kParts + 0) * kBlock + threadIdx.x] = ra;
            ring[(slot * kParts + 1) * kBlock + threadIdx.x] = rb;
            if constexpr (kParts == 3) ring[(slot * kParts + 2) * kBlock + threadIdx.x] = rc;
          }
        };
        stash(a0, b0, e0, k0);
        stash(a1, b1, e1, k1);
        stash(a2, b2, e2, k2);
        stash(a3, b3, e3, k3);
        if (stop > filled) filled = stop;
      }
    }
    // Op phase: a lane that is not inside a piece takes up to kOpsPerTrip actions (ops from its ring, lag sub-piece
    // decisions), so an observation followed by a PROP reaches the stepping phase of the SAME trip; with one action
    // per trip every op of a lane cost it a whole trip of its neighbours' stepping (C4: 45 ops against ~190 stepping
    // trips per lane).  Bounded like the stepping, and again one if / else chain per action.
#pragma unroll 1
    for (int act = 0; act < kOpsPerTrip; ++act) {
      const bool idle = !(ADAPT ? stepping : rem > 0);
      if (idle && LAG && in_prop) {
        int which;
        const double tau = lag_next(m, ops, ls, which);
        const bool bol = tau < t_stop;
        const double stop = bol ? tau : t_stop;
        if (stop > t_cur) {  // next sub-piece; n = ceil(dt / h_max) as ode_piece
          if constexpr (ADAPT) {
            stepping = true;
            t_run = t_cur;
            t_run_end = stop;
          } else {
            const double dt = stop - t_cur;
            double nf = ceil(dt / m.rk4_h_max);
            if (!(nf >= 1.0)) nf = 1.0;
            if (nf > 1.0e7) nf = 1.0e7;
            rem = static_cast<int32_t>(nf);
            h = dt / static_cast<double>(rem);
            t_piece = t_cur;
            n_piece = rem;
          }
          t_cur = stop;
        } else if (bol) {
          lag_apply_bolus<NS>(m, ops, ls, which, th, x);
        } else {
          in_prop = false;
          if (t_stop > clk) clk = t_stop;
          ++o;
        }
      } else if (idle && o < filled) {
        // every field of the op is fetched before its kind is looked at: independent loads, one memory latency per
        // op instead of the chain meta -> kind -> payload (a lone wave per SIMD has nothing to hide either behind)
        // (and from one packed 48-byte record, DevOps::op_rec, that the top-up above left in the lane's ring: the lanes of
        // a wave read 64 different ops, so every separate array would be another 64-line gather)
        const int slot = static_cast<int>(o & (kRing - 1));
        const double2 rec0 = ring[(slot * kParts + 0) * kBlock + threadIdx.x];
        const double2 rec1 = ring[(slot * kParts + 1) * kBlock + threadIdx.x];
        const uint64_t mw = static_cast<uint64_t>(__double_as_longlong(rec0.x));
        const uint32_t meta = static_cast<uint32_t>(mw);
        const int32_t op_steps = static_cast<int32_t>(mw >> 32);
        const double a = rec0.y;
        const double op_h = rec1.x;
        const double op_r0 = rec1.y;
        double op_t0 = 0.0, op_t1 = 0.0;
        if constexpr (LAG || ADAPT || M::CUSTOM) {
          const double2 rec2 = ring[(slot * kParts + (kParts - 1)) * kBlock + threadIdx.x];
          op_t0 = rec2.x;
          op_t1 = rec2.y;
        }
        const uint32_t kind = meta & 0xffu;
        const int io = static_cast<int>((meta >> 8) & 0xffffu);
        bool next_op = true;  // LAG: a PROP / an occasion opening stays the current op until the branch above closes it
        if (kind == OP_PROP) {
          ode_rates<M>(m, ops.op_rate, o, ops.n_rate, op_r0, rs);
          if constexpr (LAG) {
            in_prop = true;
            t_cur = (clk > op_t0) ? clk : op_t0;
            t_stop = op_t1;
            next_op = false;
          } else if constexpr (ADAPT) {
            t_run = op_t0;
            t_run_end = op_t1;
            stepping = t_run_end > t_run;
          } else {
            h = op_h;
            rem = op_steps;
            if constexpr (M::CUSTOM) {
              t_piece = op_t0;
              n_piece = rem;
            }
          }
        } else if (kind == OP_OBS) {
          double y = ode_out<M>(m, L, x, io, a);
          if (LAG && st == PMX_PAIR_BAD_LAG) y = nanv;
          if (ADAPT && as.failed) {
            if (st == PMX_PAIR_OK) st = PMX_PAIR_SOLVER_FAIL;
            y = nanv;
          }
          if constexpr (LL) {
            ll_accumulate(ops.ll_obs + row * 4, y, ll_acc);
          } else {
            if (st == PMX_PAIR_OK && !isfinite(y)) st = PMX_PAIR_NONFINITE;
            pred[row * ld + p] = y;
          }
          ++row;
        } else if (kind == OP_BOLUS) {
          const int bd = input_entry(m.bolus_dest, io);
          const int dest = (bd >= 0) ? bd : io;
          const double amt = a * fa_of(m, th, io);
  #pragma unroll
          for (int j = 0; j < NS; ++j) x[j] += (j == dest) ? amt : 0.0;
        } else {
          L.occ = static_cast<int64_t>(a);
          ode_reset<M>(L, io, x);
          if constexpr (LAG) {
            const int64_t occ = static_cast<int64_t>(a);
  #pragma unroll
            for (int k = 0; k < kMaxLagSlots; ++k) {
              if (k < m.n_lag_slots) {
                ls.cur[k] = static_cast<int32_t>(as_const(ops.lagb_off)[occ * m.n_lag_slots + k]);
                ls.end[k] = static_cast<int32_t>(as_const(ops.lagb_off)[occ * m.n_lag_slots + k + 1]);
              }
            }
            // boluses landing before the occasion's first remaining event open the occasion (zero rates there).  The
            // first of them is the first event of the re-sorted list: applied here, at the clock as it stands (the
            // occasion's recorded initial time, op_b of the RESET); the rest is reached from the clock.
            clk = op_h;
            int which;
            const double tau = lag_next(m, ops, ls, which);
            const double t_first = op_t0;
            if ((tau < t_first || (tau == t_first && lag_lands_first(ls, which, (meta >> 25) & 3u))) && t_first < inf) {
              lag_apply_bolus<NS>(m, ops, ls, which, th, x);
  #pragma unroll
              for (int j = 0; j < M::NR; ++j) rs[j] = 0.0;
              in_prop = true;
              t_cur = clk;
              t_stop = t_first;
              next_op = false;
            }
          }
        }
        if (next_op) ++o;
      }
    }  // op phase
    // Stepping phase
    if (ADAPT ? stepping : rem > 0) {
      // up to ops.steps_per_trip steps of the open piece: a trip through this state machine costs ~7 bare RK4 steps
      // of dx/dt = -ke x + r (tools/experiments/rk4_latency_probe.hip: 46 ns/step for a lone wave, 333 ns/trip here), and a
      // batch of a few 10k pairs is one wave per SIMD, i.e. latency-bound.  Bounded, so a lane that needs its next
      // op waits for at most that many steps of its neighbours, not for the longest piece in the wave; the host
      // picks the bound from the batch size (pmx_api.cpp).
      const int32_t spt = ops.steps_per_trip;
      if constexpr (ADAPT) {
        for (int32_t j = 0; j < spt && stepping; ++j) stepping = dopri5_advance<M>(m, L, x, rs, t_run, t_run_end, as);
      } else {
        const int32_t kk = rem < spt ? rem : spt;
        for (int32_t j = 0; j < kk; ++j) {
          double t = 0.0;
          if constexpr (M::CUSTOM) t = t_piece + static_cast<double>(n_piece - rem + j) * h;
          rk4_step<M>(L, x, rs, t, h);
        }
        rem -= kk;
      }
    }
  }
  if constexpr (LL) {
    if (st == PMX_PAIR_OK && !isfinite(ll_acc)) st = PMX_PAIR_NONFINITE;
    if (lane_ok) ops.ll_out[batch ? s : (s * ops.ll_ld + p)] = (st == PMX_PAIR_OK || st == PMX_PAIR_NONFINITE) ? ll_acc : nanv;
  }
  if (status != nullptr && lane_ok) status[batch ? s : (s * P + p)] = st;  // every pair writes its byte
}

}  // namespace
}  // namespace pmx
