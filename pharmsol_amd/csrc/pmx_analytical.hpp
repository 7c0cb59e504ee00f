// pmx_analytical.hpp — the Analytical back-end walked with USER closures (hiprtc-compiled), generic over a model
// policy.  The reference's `Analytical::new(eq, seq_eq, lag, fa, init, out)` takes arbitrary functions of
// (theta, t, covariates) (src/simulator/mod.rs:41-197); the library's own kernels (pmx_kernels.hip) only know closed
// descriptor forms.  This header is the general case: every closure is a device function the policy M forwards to —
// the user's source text, or code pmx_jit.cpp generates from the descriptor when the user leaves a closure out — and
// every covariate is looked up on the device at the time the reference's closure would see:
//
//   lag    at the bolus' recorded time             Occasion::add_lagtime          src/data/structs.rs:611-643
//   fa     at the bolus' time AFTER the lag shift  Occasion::add_bioavailability  src/data/structs.rs:645-666
//   init   at 0.0, occasion index 0 only           Analytical::initial_state      analytical/mod.rs:409-426
//   out    at the observation time                 Analytical::process_observation analytical/mod.rs:373-407
//   seq_eq at the sub-segment's absolute end, on a parameter vector rebuilt per solve   analytical/mod.rs:331,360
//   eq     at the sub-segment LENGTH dt (the macro lowering's derive(p, dt, cov), expand/analytical.rs:254,286) or,
//          PMX_COV_TIME_SEGMENT_END_ABS, at its absolute end                            analytical/mod.rs:363-364
//
// Lag times may differ from bolus to bolus (covariates move), so an occasion's lagged boluses are re-sorted per lane:
// landing times in a private array, insertion-sorted (the reference re-sorts the whole event list, structs.rs:638-642;
// only the order of the lagged boluses among themselves and against the fixed events matters, and the latter is what
// the PROP splitting below reproduces).
#pragma once

#include "pmx_device.hpp"
#include "pmx_structures.hpp"
#include "pmx_userlag.hpp"

namespace pmx {
namespace {

// Policy M:
//   NS, NP, NIN (ndrugs), NOUT, NCOV, NDER       sizes
//   KID            built-in structure (PMX_K_*), or -1: the user's own propagator `eq`
//   RATE_INPUT     the input whose infusion rate a built-in structure reads (rateiv[0]; rateiv[1] under pm_* indexing)
//   PM             1 = Pmetrics indexing: model state / input 0 is the wrapper's dead pad slot (analytical/mod.rs:62-90)
//   COV_ABS        eq's derive sees covariates at the absolute segment end instead of at dt
//   HAS_LAG, HAS_FA, HAS_SEQ, HAS_DERIVE, STATIC_COEF (built-in structure whose rate constants never change in a lane)
//   derive(t, p, cov, der) / lag(t, p, cov, der, lag[NIN]) / fa(.., fa[NIN]) / init(p, cov, der, x) /
//   out(t, x, p, cov, der, y) / seq(t, theta, cov, pw) / eq(dt, x, pw, cov, rate, der, xn) / kernel_params(pw, der, kp)
// (UserCov / UserDer / UserLag: pmx_userlag.hpp)

template <class M>
struct UserState {
  double x[M::NS];
  double pw[M::NP];  // the parameter vector of the running solve (seq_eq mutates it; rebuilt per solve)
  bool fresh;        // the next sub-segment starts a new solve
  bool cplx;         // complex eigenvalues somewhere in the current occasion: its remaining rows are NaN
};

template <class M>
__device__ __forceinline__ void user_add(double (&x)[M::NS], int state, double amt) {
#pragma unroll
  for (int i = 0; i < M::NS; ++i) x[i] += (i == state) ? amt : 0.0;
}

// one sub-segment [tA, tB] of a solve (analytical/mod.rs:334-367)
template <class M>
__device__ __forceinline__ void user_piece(const DevOps& ops, int64_t occ, const double* __restrict__ th,
                                           const typename Structure<kernel_structure(M::KID < 0 ? 0 : M::KID)>::Coef& coef,
                                           UserState<M>& s, double tA, double tB, const double* rate) {
  const double dt = tB - tA;
  if (s.fresh) {  // parameters_v is rebuilt from the support point for every solve (:331)
#pragma unroll
    for (int i = 0; i < M::NP; ++i) s.pw[i] = th[i];
    s.fresh = false;
  }
  UserCov<M> cov;
  UserDer<M> der;
  if constexpr (M::HAS_SEQ) {  // (self.seq_eq)(&mut parameters_v, next_t, covariates)  (:360)
    user_cov<M>(ops, occ, tB, cov);
    M::seq(tB, th, cov.v, s.pw);
  }
  const double t_cov = M::COV_ABS ? tB : dt;
  if constexpr (M::KID >= 0) {
    constexpr int ST = kernel_structure(M::KID);
    using S = Structure<ST>;
    double xk[S::NS];
#pragma unroll
    for (int i = 0; i < S::NS; ++i) xk[i] = s.x[i + M::PM];  // (PM = 1: the pm_* wrapper drops slot 0, analytical/mod.rs:62-90)
    if constexpr (M::STATIC_COEF) {
      advance<ST>(coef, xk, dt, rate[M::RATE_INPUT]);
    } else {
      user_cov_der<M>(ops, occ, t_cov, s.pw, cov, der);
      double kp[kernel_nparams(M::KID)], q[kernel_nparams(M::KID)];
      M::kernel_params(s.pw, der.v, kp);
      to_native_params<M::KID>(kp, q);
      typename S::Prop pr;
      if (!make_prop_dyn<ST>(q, dt, pr)) s.cplx = true;
      S::apply(pr, xk, rate[M::RATE_INPUT]);
    }
#pragma unroll
    for (int i = 0; i < S::NS; ++i) s.x[i + M::PM] = xk[i];
    if constexpr (M::PM != 0) s.x[0] = 0.0;  // ... and re-pads it with 0 after every kernel call
  } else {
    user_cov_der<M>(ops, occ, t_cov, s.pw, cov, der);
    double xn[M::NS];
#pragma unroll
    for (int i = 0; i < M::NS; ++i) xn[i] = s.x[i];
    M::eq(dt, s.x, s.pw, cov.v, rate, der.v, xn);
#pragma unroll
    for (int i = 0; i < M::NS; ++i) s.x[i] = xn[i];
  }
}

// bolus `amount` on `input`, in effect at time t (its time after the lag shift): scaled by fa, added to x[input]
// (structs.rs:645-666; equation/mod.rs:328)
template <class M>
__device__ __forceinline__ void user_bolus(const DevOps& ops, int64_t occ, const double* __restrict__ th, UserState<M>& s,
                                           double t, int input, double amount) {
  double f = 1.0;
  if constexpr (M::HAS_FA) {
    UserCov<M> cov;
    UserDer<M> der;
    user_cov_der<M>(ops, occ, t, th, cov, der);
    double fa[M::NIN];
#pragma unroll
    for (int i = 0; i < M::NIN; ++i) fa[i] = 1.0;
    M::fa(t, th, cov.v, der.v, fa);
#pragma unroll
    for (int i = 0; i < M::NIN; ++i) f = (i == input) ? fa[i] : f;
  }
  user_add<M>(s.x, input, amount * f);
}

template <class M>
__device__ __forceinline__ void user_lag_apply(const DevOps& ops, int64_t occ, const double* __restrict__ th,
                                               UserState<M>& s, UserLag& L) {
  double tau, amount;
  int input;
  user_lag_take<M>(ops, occ, th, L, &tau, &input, &amount);
  user_bolus<M>(ops, occ, th, s, tau, input, amount);
  s.fresh = true;  // an event: whatever follows is another solve
}

// Everything one lane does for one subject.  `UNIFORM`: the op stream is wave-uniform (GRID mapping) and is fetched
// through the scalar unit.
template <class M, bool UNIFORM, bool LL>
__device__ __forceinline__ void user_walk_subject(const DevOps& ops, const double* __restrict__ th, int64_t subj,
                                                  bool walk, bool store_ok, double* __restrict__ pred, int64_t ld,
                                                  int64_t p, double* ll_slot, uint8_t* status_slot, int state_override = -1) {
  constexpr int NS = M::NS;
  using Coef = typename Structure<kernel_structure(M::KID < 0 ? 0 : M::KID)>::Coef;
  const double nanv = __longlong_as_double(0x7ff8000000000000LL);
  const double inf = __longlong_as_double(0x7ff0000000000000LL);
  auto u64 = [](int64_t v) { return UNIFORM ? uniform64(v) : v; };
  auto u32 = [](uint32_t v) { return UNIFORM ? uniform32(v) : v; };
  auto uf = [](double v) { return UNIFORM ? uniformf64(v) : v; };
  const int64_t o0 = u64(as_const(ops.subj_op_off)[subj]);
  const int64_t o1 = walk ? u64(as_const(ops.subj_op_off)[subj + 1]) : o0;  // (GRID: idle lanes shadow the last support point, stores masked)
  int64_t row = u64(as_const(ops.subj_obs_off)[subj]);

  Coef coef;
  bool lane_cplx = false;
  if constexpr (M::KID >= 0 && M::STATIC_COEF) {
    double kp[kernel_nparams(M::KID)], q[kernel_nparams(M::KID)];
    UserDer<M> none;
    none.v[0] = 0.0;
    M::kernel_params(th, none.v, kp);
    to_native_params<M::KID>(kp, q);
    lane_cplx = !Structure<kernel_structure(M::KID)>::prepare(q, coef);
  }
  UserState<M> s;
#pragma unroll
  for (int i = 0; i < NS; ++i) s.x[i] = 0.0;
#pragma unroll
  for (int i = 0; i < M::NP; ++i) s.pw[i] = th[i];
  s.fresh = true;
  s.cplx = lane_cplx;
  UserLag lagst;
  lagst.n = lagst.cur = 0;
  lagst.base = 0;
  lagst.big = false;
  lagst.nxt_tau = inf;
  lagst.nxt_idx = -1;
  int64_t occ = 0;
  uint8_t st = PMX_PAIR_OK;
  bool bad_lag = false;
  double ll_acc = 0.0;
  double zero_rate[M::NIN];
#pragma unroll
  for (int i = 0; i < M::NIN; ++i) zero_rate[i] = 0.0;

#pragma unroll 1
  for (int64_t o = o0; o < o1; ++o) {
    const uint32_t meta = u32(as_const(ops.op_meta)[o]);
    const uint32_t kind = meta & 0xffu;
    const int io = static_cast<int>((meta >> 8) & 0xffffu);
    const double a = uf(as_const(ops.op_a)[o]);
    if (kind == OP_PROP) {
      double rate[M::NIN];
#pragma unroll
      for (int i = 0; i < M::NIN; ++i) rate[i] = 0.0;
      if constexpr (M::KID >= 0) {
        rate[M::RATE_INPUT] = uf(as_const(ops.op_b)[o]);
      } else {
#pragma unroll
        for (int i = 0; i < M::NIN; ++i) rate[i] = (i < ops.n_rate) ? uf(as_const(ops.op_rate)[o * ops.n_rate + i]) : 0.0;
      }
      const double t0 = uf(as_const(ops.op_t0)[o]), t1 = uf(as_const(ops.op_t1)[o]);
      if (((meta >> 24) & 1u) == 0u) s.fresh = true;  // first sub-segment of a solve
      double t = t0;
      if constexpr (M::HAS_LAG) {
#pragma unroll 1
        for (;;) {  // a lagged bolus landing inside [t0, t1) splits it where the reference's re-sorted list would
          const double tau = user_lag_next(lagst);
          if (!(tau < t1)) break;
          if (tau > t) {
            user_piece<M>(ops, occ, th, coef, s, t, tau, rate);
            t = tau;
          }
          user_lag_apply<M>(ops, occ, th, s, lagst);
        }
      }
      if (t1 > t) user_piece<M>(ops, occ, th, coef, s, t, t1, rate);
    } else if (kind == OP_OBS) {
      if constexpr (M::HAS_LAG) {
        // lagged boluses landing before this observation with no PROP step in between (events closer than the solve's
        // 1e-12 dedup): taken first, without propagation.  (Behind a PROP step nothing is pending below its end.)
#pragma unroll 1
        for (;;) {
          const double tau = user_lag_next(lagst);
          if (!(tau < a)) break;
          user_lag_apply<M>(ops, occ, th, s, lagst);
        }
      }
      UserCov<M> cov;
      UserDer<M> der;
      user_cov_der<M>(ops, occ, a, th, cov, der);
      double yv[M::NOUT];
#pragma unroll
      for (int q = 0; q < M::NOUT; ++q) yv[q] = 0.0;
      M::out(a, s.x, th, cov.v, der.v, yv);
      double y = yv[0];
#pragma unroll
      for (int q = 1; q < M::NOUT; ++q) y = (q == io) ? yv[q] : y;
      if (state_override >= 0) y = select_state<NS>(s.x, state_override);  // Prediction::state read-out (wave-uniform)
      if (s.cplx || bad_lag) y = nanv;
      if (s.cplx && st == PMX_PAIR_OK) st = PMX_PAIR_COMPLEX_ROOTS;
      if constexpr (LL) {
        if constexpr (UNIFORM)
          ll_accumulate(as_const(ops.ll_obs) + row * 4, y, ll_acc);
        else
          ll_accumulate(ops.ll_obs + row * 4, y, ll_acc);
      } else {
        if (st == PMX_PAIR_OK && !isfinite(y)) st = PMX_PAIR_NONFINITE;
        if (store_ok) pred[row * ld + p] = y;
      }
      ++row;
    } else if (kind == OP_BOLUS) {
      user_bolus<M>(ops, occ, th, s, uf(as_const(ops.op_b)[o]), io, a);
      s.fresh = true;
    } else {  // OP_RESET: initial_state (analytical/mod.rs:409-426), then this lane's view of the lagged boluses
      occ = static_cast<int64_t>(a);
#pragma unroll
      for (int i = 0; i < NS; ++i) s.x[i] = 0.0;
      s.fresh = true;
      if (s.cplx && st == PMX_PAIR_OK) st = PMX_PAIR_COMPLEX_ROOTS;
      s.cplx = lane_cplx;  // a new occasion re-derives its coefficients; the pair stays failed
      if (io) {
        UserCov<M> cov;
        UserDer<M> der;
        user_cov_der<M>(ops, occ, 0.0, th, cov, der);
        M::init(th, cov.v, der.v, s.x);
      }
      if constexpr (M::HAS_LAG) {
        if (!user_lag_open<M>(ops, occ, th, lagst)) bad_lag = true;
        // boluses landing before the occasion's first remaining event open the occasion; no infusion can be active yet
        const double t_first = uf(as_const(ops.op_t0)[o]);
        bool started = false;
        double t = 0.0;
#pragma unroll 1
        for (;;) {
          const double tau = user_lag_next(lagst);
          if (!(tau < t_first)) break;
          if (started && tau > t) user_piece<M>(ops, occ, th, coef, s, t, tau, zero_rate);
          t = tau;
          started = true;
          user_lag_apply<M>(ops, occ, th, s, lagst);
        }
        if (started && t_first > t && t_first < inf) {
          s.fresh = true;
          user_piece<M>(ops, occ, th, coef, s, t, t_first, zero_rate);
        }
        s.fresh = true;
      }
    }
  }
  if (s.cplx && st == PMX_PAIR_OK) st = PMX_PAIR_COMPLEX_ROOTS;
  if (bad_lag) st = PMX_PAIR_BAD_LAG;
  if constexpr (LL) {
    if (st == PMX_PAIR_OK && !isfinite(ll_acc)) st = PMX_PAIR_NONFINITE;
    if (store_ok && ll_slot != nullptr) *ll_slot = (st == PMX_PAIR_OK || st == PMX_PAIR_NONFINITE) ? ll_acc : nanv;
  }
  if (store_ok && status_slot != nullptr) *status_slot = st;  // every pair writes its byte: no memset before the launch
}

// GRID: lane = support point, a block walks a chunk of subjects (wave-uniform op stream)
template <class M, bool LL>
__device__ __forceinline__ void user_grid_body(const DevModel& m, const DevOps& ops, const double* __restrict__ theta,
                                               int64_t P, int64_t S, int32_t s_chunk, int32_t n_ptiles,
                                               double* __restrict__ pred, int64_t ld, uint8_t* __restrict__ status) {
  const int64_t b = blockIdx.x;
  const int32_t ptile = static_cast<int32_t>(b % n_ptiles);
  const int64_t chunk = b / n_ptiles;
  const int64_t p = static_cast<int64_t>(ptile) * kBlock + threadIdx.x;
  const bool lane_ok = p < P;
  const int64_t pc = lane_ok ? p : (P - 1);
  const double* __restrict__ th = theta + pc * m.nparams;
  const int64_t s_begin = chunk * s_chunk;
  const int64_t s_end = (s_begin + s_chunk < S) ? (s_begin + s_chunk) : S;
#pragma unroll 1
  for (int64_t s = s_begin; s < s_end; ++s)
    user_walk_subject<M, true, LL>(ops, th, s, true, lane_ok, pred, ld, pc, LL ? (ops.ll_out + s * ops.ll_ld + pc) : nullptr,
                                   status ? (status + s * P + pc) : nullptr, m.state_override);
}

// PAIR: lane = one (subject, support point) pair; batch: subject s with theta row s
template <class M, bool LL>
__device__ __forceinline__ void user_pair_body(const DevModel& m, const DevOps& ops, const double* __restrict__ theta,
                                               int64_t P, int64_t S, int32_t batch, double* __restrict__ pred, int64_t ld,
                                               uint8_t* __restrict__ status) {
  const int64_t n_pairs = batch ? S : S * P;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const bool lane_ok = i < n_pairs;
  const int64_t ic = lane_ok ? i : (n_pairs - 1);
  const int64_t s = as_const(ops.subj_order)[batch ? ic : (ic / P)];
  const int64_t p = batch ? 0 : (ic % P);
  const double* __restrict__ th = theta + (batch ? s : p) * m.nparams;
  user_walk_subject<M, false, LL>(ops, th, s, lane_ok, lane_ok, pred, ld, p, LL ? (ops.ll_out + (batch ? s : (s * ops.ll_ld + p))) : nullptr,
                                  status ? (status + (batch ? s : (s * P + p))) : nullptr, m.state_override);
}

}  // namespace
}  // namespace pmx
