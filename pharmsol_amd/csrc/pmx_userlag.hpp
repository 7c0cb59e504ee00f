// pmx_userlag.hpp — what the two walkers for USER closures (pmx_analytical.hpp, pmx_ode_user.hpp) share: covariates and
// derived values at a time t, and a lane's own view of an occasion's lagged boluses.
//
// The reference rewrites the event list per (subject, support point): every bolus moves by lag(theta, t_recorded, cov)
// and the list is re-sorted (Occasion::add_lagtime + sort, src/data/structs.rs:611-643).  A user lag closure may return a
// different value for every bolus (covariates move), so ALL boluses of a model with a lag closure leave the op stream
// into one list per occasion (lagb_time / lagb_amount / lagb_input) and each lane sorts ITS landing times.
#pragma once

#include "pmx_device.hpp"

namespace pmx {
namespace {

// Policy members read here: NCOV, NDER, NIN, HAS_DERIVE, derive(t, p, cov, der), lag(t, p, cov, der, lag[NIN])
template <class M>
struct UserCov {
  double v[M::NCOV > 0 ? M::NCOV : 1];
};
template <class M>
struct UserDer {
  double v[M::NDER > 0 ? M::NDER : 1];
};

template <class M>
__device__ __forceinline__ void user_cov(const DevOps& ops, int64_t occ, double t, UserCov<M>& c) {
  if constexpr (M::NCOV > 0) {
#pragma unroll
    for (int i = 0; i < M::NCOV; ++i) c.v[i] = cov_at(ops, occ, i, t);
  } else {
    c.v[0] = 0.0;
  }
}
// covariates at t and the derived values there (`derive` runs first in every macro-lowered closure)
template <class M>
__device__ __forceinline__ void user_cov_der(const DevOps& ops, int64_t occ, double t, const double* p, UserCov<M>& c,
                                             UserDer<M>& d) {
  user_cov<M>(ops, occ, t, c);
#pragma unroll
  for (int i = 0; i < (M::NDER > 0 ? M::NDER : 1); ++i) d.v[i] = 0.0;
  if constexpr (M::HAS_DERIVE) M::derive(t, p, c.v, d.v);
}

// (The scan below is compiled only into the PMX_USER_BIG_LISTS build of a model, which the library makes - lazily - for
// populations that hold such an occasion: inlined into the ordinary build it cost the walker 26 vector registers, 142
// instead of 116, i.e. three waves per SIMD instead of four, and the user-closure workload ran 3.4 -> 4.9 ms.)
// The lane's view of the occasion's lagged boluses.  Up to kUserLagKept of them: landing times in a private array,
// insertion-sorted, `idx` = position in the occasion's list.  A longer list (`big`) is not stored at all: the next
// bolus to land is found by scanning the list for the smallest (landing time, position) after the last one taken -
// n lag evaluations per bolus instead of one, no memory, no cap (a rare shape: > 64 boluses in ONE occasion).
struct UserLag {
  double tau[kUserLagKept];
  uint16_t idx[kUserLagKept];
  int32_t n, cur;
  int64_t base;
  bool big;
  double nxt_tau;   // big: the next bolus to land (inf: none left) ...
  int32_t nxt_idx;  // ... and its position
};

// landing time of bolus j of the list: `if l != 0.0 { *bolus.mut_time() += l }` (structs.rs:631-634); NaN -> +inf, *ok = false
template <class M>
__device__ __forceinline__ double user_lag_landing(const DevOps& ops, int64_t occ, const double* __restrict__ th,
                                                   int64_t base, int32_t j, bool* ok) {
  const double t = as_const(ops.lagb_time)[base + j];
  const int input = as_const(ops.lagb_input)[base + j];
  UserCov<M> cov;
  UserDer<M> der;
  user_cov_der<M>(ops, occ, t, th, cov, der);
  double lag[M::NIN];
#pragma unroll
  for (int i = 0; i < M::NIN; ++i) lag[i] = 0.0;
  M::lag(t, th, cov.v, der.v, lag);
  double l = 0.0;
#pragma unroll
  for (int i = 0; i < M::NIN; ++i) l = (i == input) ? lag[i] : l;
  double tau = (l != 0.0) ? (t + l) : t;
  if (tau != tau) {
    *ok = false;
    tau = __longlong_as_double(0x7ff0000000000000LL);
  }
  return tau;
}

// big lists: the smallest (landing time, position) strictly after (after_tau, after_idx); position -1 = from the start
template <class M>
__device__ __forceinline__ void user_lag_scan(const DevOps& ops, int64_t occ, const double* __restrict__ th, UserLag& L,
                                              double after_tau, int32_t after_idx, bool* ok) {
  double best = __longlong_as_double(0x7ff0000000000000LL);
  int32_t best_j = -1;
#pragma unroll 1
  for (int32_t j = 0; j < L.n; ++j) {
    const double tau = user_lag_landing<M>(ops, occ, th, L.base, j, ok);
    const bool later = after_idx < 0 || tau > after_tau || (tau == after_tau && j > after_idx);
    if (later && (best_j < 0 || tau < best)) {  // (equal landing times keep the list order: first position wins)
      best = tau;
      best_j = j;
    }
  }
  L.nxt_tau = best_j >= 0 ? best : __longlong_as_double(0x7ff0000000000000LL);
  L.nxt_idx = best_j;
}

// RESET of a model with lag: this lane's landing times of the occasion's lagged boluses, sorted (add_lagtime + sort,
// structs.rs:611-643).  Returns false when a lag time is NaN (the reference panics in its sort).
template <class M>
__device__ __forceinline__ bool user_lag_open(const DevOps& ops, int64_t occ, const double* __restrict__ th, UserLag& L) {
  bool ok = true;
  L.base = as_const(ops.lagb_off)[occ];
  const int64_t n = as_const(ops.lagb_off)[occ + 1] - L.base;
  L.n = static_cast<int32_t>(n);
  L.cur = 0;
  L.big = false;
  L.nxt_tau = __longlong_as_double(0x7ff0000000000000LL);
  L.nxt_idx = -1;
#ifdef PMX_USER_BIG_LISTS
  L.big = n > kUserLagKept;
  if (L.big) {
    user_lag_scan<M>(ops, occ, th, L, 0.0, -1, &ok);
    return ok;
  }
#else
  if (n > kUserLagKept) L.n = kUserLagKept;  // (never: the host launches the PMX_USER_BIG_LISTS build for such populations)
#endif
#pragma unroll 1
  for (int32_t j = 0; j < L.n; ++j) {
    const double tau = user_lag_landing<M>(ops, occ, th, L.base, j, &ok);
    int32_t k = j;  // stable insertion: equal landing times keep the list order
#pragma unroll 1
    while (k > 0 && L.tau[k - 1] > tau) {
      L.tau[k] = L.tau[k - 1];
      L.idx[k] = L.idx[k - 1];
      --k;
    }
    L.tau[k] = tau;
    L.idx[k] = static_cast<uint16_t>(j);
  }
  return ok;
}

// landing time of the next pending bolus (+inf: none)
__device__ __forceinline__ double user_lag_next(const UserLag& L) {
#ifdef PMX_USER_BIG_LISTS
  if (L.big) return L.nxt_tau;
#endif
  return (L.cur < L.n) ? L.tau[L.cur] : __longlong_as_double(0x7ff0000000000000LL);
}

// take the next pending bolus off the list: its landing time, input and recorded amount
template <class M>
__device__ __forceinline__ void user_lag_take(const DevOps& ops, int64_t occ, const double* __restrict__ th, UserLag& L,
                                              double* tau, int* input, double* amount) {
  int32_t j;
#ifdef PMX_USER_BIG_LISTS
  if (L.big) {
    j = L.nxt_idx;
    *tau = L.nxt_tau;
    bool ok = true;
    user_lag_scan<M>(ops, occ, th, L, *tau, j, &ok);
  } else
#endif
  {
    j = L.idx[L.cur];
    *tau = L.tau[L.cur];
    L.cur += 1;
  }
  *input = as_const(ops.lagb_input)[L.base + j];
  *amount = as_const(ops.lagb_amount)[L.base + j];
}

}  // namespace
}  // namespace pmx
