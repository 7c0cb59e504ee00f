// pmx_kernels.hip — hand-written gfx950 kernels for the (subject x support point) prediction grid.
//
// One wavefront lane per (subject, support point) pair, two lane mappings:
//
//  GRID  lane = support point (fastest index), a block walks a chunk of subjects.
//        The op stream of a subject is WAVE-UNIFORM: every lane of every wave in the
//        block executes the same BOLUS/OBS/PROP sequence, so op fetches are scalar
//        (s_load through the scalar cache), branches are scalar, and there is no
//        divergence at all.  Stores are pred[row][p0..p0+63]: 512 contiguous bytes per
//        wave-instruction.  Used when n_support >= 32 (NPAG-style grids, C3/C5).
//
//  PAIR  lane = one (subject, support point) pair with its own op cursor; lanes of a
//        wave run different schedules (divergent timelines), the wave loops until every
//        lane's cursor reaches its end (exec-masked loop == ballot of "any lane active").
//        Subjects are pre-sorted by work so neighbouring lanes finish together.
//        Used for n_support < 32 (C2) and for the batch shape (C4: one theta per subject).
//
// States live in registers (1-4 doubles; LDS staging would only add latency), the
// rate-constant-only part of every closed form is hoisted out of the event loop
// (pmx_structures.hpp).  No MFMA: 2-6-state systems have no dense contraction.
//
// Reference contracts: equation/mod.rs:300-358,480-516 (event loop), analytical/mod.rs:299-426,
// ode/mod.rs:609-823 (ODE event loop; diffsol replaced by fixed-step RK4).
#include <hip/hip_runtime.h>

#include <type_traits>

#include <cmath>
#include <cstdint>

#include "pmx_kernels.hpp"
#include "pmx_structures.hpp"
#include "pmx_device.hpp"
#include "pmx_ode.hpp"

namespace pmx {

namespace {

// derive: derived[d] = ((theta[src] * f0) * f1) (expand/analytical.rs:254,286; bindings.rs:98-117).  The factors
// depend on the op's covariates only, not on the lane: the host evaluated them (pmx_compile.cpp op_fac); `fac` points
// at this op's [n_derived][PMX_MAX_FACTORS] block, `d` = index of the derived value, `base` = theta[src_param].
__device__ __forceinline__ double apply_factors(const DevModel& m, int d, double base, const double* __restrict__ fac) {
  double v = base;
#pragma unroll
  for (int k = 0; k < PMX_MAX_FACTORS; ++k) {
    double f = 1.0;
#pragma unroll
    for (int dd = 0; dd < PMX_MAX_DERIVED; ++dd)
      if (dd == d && k < m.derived[dd].n_factors) f = fac[dd * PMX_MAX_FACTORS + k];
    v = v * f;
  }
  return v;
}

// wide wave-uniform fetches through the scalar unit, placed where they are written (volatile: the compiler neither
// moves nor merges them, and tracks their completion itself)
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
template <class V>
__device__ __forceinline__ V sload_here(const void* p_) {
  const void* p = reinterpret_cast<const void*>(uniform64(reinterpret_cast<int64_t>(p_)));  // (wave-uniform: a scalar address)
  return *(const volatile __attribute__((address_space(4))) V*)(p);
}

// Everything a lane needs besides its state; filled once per lane.
template <int KID>
struct LaneModel {
  static constexpr int ST = kernel_structure(KID);
  using S = Structure<ST>;
  static constexpr int NS = S::NS;
  static constexpr int NKP = kernel_nparams(KID);
  typename S::Coef coef;
  double kp_base[NKP];         // theta[...] for each kernel-order parameter (base value when derived)
  double vol_base[PMX_MAX_OUT];  // theta[...] behind each output's volume (1.0 when none)
  double inv_vol[PMX_MAX_OUT];
  double xinit[NS];
  bool ok;
};

template <int KID, bool DYN>
__device__ __forceinline__ void lane_setup(const DevModel& m, const double* __restrict__ th, LaneModel<KID>& L) {
  using LM = LaneModel<KID>;
#pragma unroll
  for (int j = 0; j < LM::NKP; ++j) {
    int idx = j;
    if (m.n_bind > 0) idx = (m.bind[j].src == PMX_SRC_DERIVED) ? m.derived[m.bind[j].index].src_param : m.bind[j].index;
    L.kp_base[j] = th[idx];
  }
#pragma unroll
  for (int o = 0; o < PMX_MAX_OUT; ++o) {
    double v = 1.0;
    if (o < m.nout) {
      if (m.out[o].vol_src == PMX_SRC_PRIMARY) v = th[m.out[o].vol_index];
      if (m.out[o].vol_src == PMX_SRC_DERIVED) v = th[m.derived[m.out[o].vol_index].src_param];
    }
    L.vol_base[o] = v;
    L.inv_vol[o] = 1.0 / v;
  }
#pragma unroll
  for (int i = 0; i < LM::NS; ++i) {
    const int st = i + m.pm;  // model state index of kernel state i
    L.xinit[i] = (m.has_init && m.init_param[st] >= 0) ? th[m.init_param[st]] : 0.0;
  }
  L.ok = true;
  if constexpr (!DYN) {
    double q[LM::NKP];
    to_native_params<KID>(L.kp_base, q);
    L.ok = LM::S::prepare(q, L.coef);
  }
}

// PROP with covariate-derived kernel parameters: this op's rate constants in the structure's native order.
template <int KID>
__device__ __forceinline__ void lane_params_dyn(const DevModel& m, const LaneModel<KID>& L, const double* cov,
                                                double (&q)[LaneModel<KID>::NKP]) {
  using LM = LaneModel<KID>;
  double kp[LM::NKP];
#pragma unroll
  for (int j = 0; j < LM::NKP; ++j) {
    double v = L.kp_base[j];
    if (m.bind[j].src == PMX_SRC_DERIVED) v = apply_factors(m, m.bind[j].index, v, cov);
    kp[j] = v;
  }
  to_native_params<KID>(kp, q);
}
// ... and the segment's propagator applied (fused prepare + make, pmx_structures.hpp make_prop_dyn)
// UNIFORM_R: the rate is wave-uniform (GRID / classed kernels), so a segment without an active infusion can skip the
// response J altogether (a scalar branch); the PAIR kernels' lanes carry their own rates and always build it
template <int KID, bool UNIFORM_R = false>
__device__ __forceinline__ bool lane_advance_dyn(const DevModel& m, const LaneModel<KID>& L, const double* cov,
                                                 double (&x)[LaneModel<KID>::NS], double dt, double r) {
  using LM = LaneModel<KID>;
  double q[LM::NKP];
  lane_params_dyn<KID>(m, L, cov, q);
  typename LM::S::Prop pr;
  bool ok;
  if (UNIFORM_R && r == 0.0) {
    ok = make_prop_dyn<LM::ST, false>(q, dt, pr);
    LM::S::apply0(pr, x);
  } else {
    ok = make_prop_dyn<LM::ST, true>(q, dt, pr);
    LM::S::apply(pr, x, r);
  }
  return ok;
}

template <int KID>
__device__ __forceinline__ double lane_out(const DevModel& m, const LaneModel<KID>& L,
                                           const double (&x)[LaneModel<KID>::NS], double xpad, int outeq,
                                           const double* cov) {
  using LM = LaneModel<KID>;
  // y[o] = x[state] / vol  (e.g. examples/analytical_vs_ode.rs:82-84)
  int state = 0, vsrc = PMX_SRC_NONE, vidx = 0;
  double inv = 1.0, vbase = 1.0;
#pragma unroll
  for (int o = 0; o < PMX_MAX_OUT; ++o) {
    if (o == outeq) {
      state = m.out[o].state;
      vsrc = m.out[o].vol_src;
      vidx = m.out[o].vol_index;
      inv = L.inv_vol[o];
      vbase = L.vol_base[o];
    }
  }
  double xs = select_state<LM::NS>(x, state - m.pm);
  if (m.pm && state == 0) xs = xpad;
  if (vsrc == PMX_SRC_DERIVED) return xs * pmx_rcp(apply_factors(m, vidx, vbase, cov));  // (pmx_structures.hpp: 8 issue slots for 15)
  return xs * inv;
}

// The same for a compile-time output O behind a wave-uniform branch on the op's output index (the lane-valued volume
// terms need no select chain then); no pm_ pad slot.
template <int KID, int O>
__device__ __forceinline__ double lane_out_at(const DevModel& m, const LaneModel<KID>& L, const double (&x)[LaneModel<KID>::NS],
                                              const double* cov) {
  using LM = LaneModel<KID>;
  const double xs = select_state<LM::NS>(x, m.out[O].state);
  if (m.out[O].vol_src == PMX_SRC_DERIVED) return xs * pmx_rcp(apply_factors(m, m.out[O].vol_index, L.vol_base[O], cov));
  return xs * L.inv_vol[O];
}
template <int KID>
__device__ __forceinline__ double lane_out_uniform(const DevModel& m, const LaneModel<KID>& L,
                                                   const double (&x)[LaneModel<KID>::NS], int outeq, const double* cov) {
  static_assert(PMX_MAX_OUT == 4, "one branch per output");
  if (outeq == 0) return lane_out_at<KID, 0>(m, L, x, cov);
  if (outeq == 1) return lane_out_at<KID, 1>(m, L, x, cov);
  if (outeq == 2) return lane_out_at<KID, 2>(m, L, x, cov);
  return lane_out_at<KID, 3>(m, L, x, cov);
}

// (lag / bioavailability helpers shared with the ODE back-end: pmx_device.hpp)

// RESET of a lag model: point the cursors at this occasion's lists and run the boluses that land before the
// occasion's first remaining event (they become the first events of the re-sorted list).
template <int ST, int NS>
__device__ __forceinline__ void lag_open_occasion(const DevModel& m, const DevOps& ops, LagState& ls, int64_t occ,
                                                  double t_first, const typename Structure<ST>::Coef& coef,
                                                  const double* __restrict__ th, double (&x)[NS]) {
#pragma unroll
  for (int k = 0; k < kMaxLagSlots; ++k) {
    if (k < m.n_lag_slots) {
      ls.cur[k] = static_cast<int32_t>(ops.lagb_off[occ * m.n_lag_slots + k]);
      ls.end[k] = static_cast<int32_t>(ops.lagb_off[occ * m.n_lag_slots + k + 1]);
    } else {
      ls.cur[k] = ls.end[k] = 0;
    }
  }
  bool started = false;
  double t = 0.0;
  for (;;) {
    int which;
    const double tau = lag_next(m, ops, ls, which);
    if (!(tau < t_first)) break;
    if (started && tau > t) advance<ST>(coef, x, tau - t, 0.0);
    t = tau;
    started = true;
    lag_apply_bolus<NS>(m, ops, ls, which, th, x);
  }
  if (started && t_first > t && t_first < __longlong_as_double(0x7ff0000000000000LL)) advance<ST>(coef, x, t_first - t, 0.0);
}

// PROP [t0, t1) of a lag model
template <int ST, int NS>
__device__ __forceinline__ void lag_prop(const DevModel& m, const DevOps& ops, LagState& ls, double t0, double t1, double r,
                                         const typename Structure<ST>::Coef& coef, const double* __restrict__ th,
                                         double (&x)[NS]) {
  double t = t0;
  for (;;) {
    int which;
    const double tau = lag_next(m, ops, ls, which);
    if (!(tau < t1)) break;
    if (tau > t) {
      advance<ST>(coef, x, tau - t, r);
      t = tau;
    }
    lag_apply_bolus<NS>(m, ops, ls, which, th, x);
  }
  if (t1 > t) advance<ST>(coef, x, t1 - t, r);
}


// ------------------------------------------------------------------------------------
// GRID kernel (analytical)
// ------------------------------------------------------------------------------------
template <int KID, bool DYN, bool LAG, bool LL>
// (covariate walkers: 3 waves per SIMD = 168 VGPRs; at 4 the three-compartment rebuild spilled 324 bytes per lane to scratch.
// C5, same box: 4 -> 16.9 ms, 3 -> 16.0 ms, 2 -> 19.8 ms)
#ifndef PMX_DYN_WAVES
#define PMX_DYN_WAVES 3
#endif
__global__ __launch_bounds__(kBlock, (!LAG && LaneModel<KID>::NS <= 2) ? 4 : ((!LAG && DYN) ? PMX_DYN_WAVES : 1)) void pmx_analytical_grid(DevModel m, DevOps ops, const double* __restrict__ theta,
                                                              int64_t P, int64_t S, int32_t s_chunk, int32_t n_ptiles,
                                                              double* __restrict__ pred, int64_t ld,
                                                              uint8_t* __restrict__ status,
                                                              const int32_t* __restrict__ subj_list, int32_t zero_status,
                                                              int32_t prop_slots) {
  using LM = LaneModel<KID>;
  constexpr int NS = LM::NS;
  const int64_t b = blockIdx.x;
  const int32_t ptile = static_cast<int32_t>(b % n_ptiles);
  const int64_t chunk = b / n_ptiles;
  const uint32_t tile = blockDim.x;  // support points per block (64 / 128 / 256: launch_analytical)
  const int64_t p = static_cast<int64_t>(ptile) * tile + threadIdx.x;
  const bool lane_ok = p < P;
  const int64_t pc = lane_ok ? p : (P - 1);  // idle lanes shadow the last support point; their stores are masked
  const double* __restrict__ th = theta + pc * m.nparams;
  // DYN: propagators the host marked for reuse wait in LDS, [slot][component][lane] (pmx_compile.cpp, prop cache codes)
  extern __shared__ double prop_cache[];
  constexpr int NPD = static_cast<int>(sizeof(typename LM::S::Prop) / sizeof(double));
  (void)prop_cache;

  LM L;
  lane_setup<KID, DYN>(m, th, L);
  uint8_t st_lane0 = L.ok ? PMX_PAIR_OK : PMX_PAIR_COMPLEX_ROOTS;
  LagState ls;
  if constexpr (LAG) {
#pragma unroll
    for (int k = 0; k < kMaxLagSlots; ++k) {
      ls.lag[k] = (k < m.n_lag_slots) ? th[m.lag_param[k]] : 0.0;
      ls.cur[k] = ls.end[k] = 0;
      // a negative lag moves the bolus EARLIER, like the reference's `time += l` (structs.rs:629-634); NaN is flagged
      if (k < m.n_lag_slots && ls.lag[k] != ls.lag[k] && st_lane0 == PMX_PAIR_OK) st_lane0 = PMX_PAIR_BAD_LAG;
    }
  }
  const uint8_t st_lane = st_lane0;
  const double nanv = __longlong_as_double(0x7ff8000000000000LL);
  double ex[LM::S::NE];  // the lane's exponentials of the last PROP (ladder)
#pragma unroll
  for (int i = 0; i < LM::S::NE; ++i) ex[i] = 0.0;

  // the op stream is read-only for the launch and indexed wave-uniformly: constant-address-space pointers
  // turn these into scalar (s_load) fetches
  const auto c_subj_op_off = as_const(ops.subj_op_off);
  const auto c_subj_obs_off = as_const(ops.subj_obs_off);
  const auto c_op_meta = as_const(ops.op_meta);
  const auto c_op_a = as_const(ops.op_a);
  const auto c_op_b = as_const(ops.op_b);
  const auto c_op_t0 = as_const(ops.op_t0);
  const auto c_op_t1 = as_const(ops.op_t1);
  (void)c_op_t0;
  (void)c_op_t1;

  const int64_t s_begin = chunk * s_chunk;
  const int64_t s_end = (s_begin + s_chunk < S) ? (s_begin + s_chunk) : S;
  for (int64_t si = s_begin; si < s_end; ++si) {
    const int64_t s = subj_list ? static_cast<int64_t>(as_const(subj_list)[si]) : si;
    const int64_t o0 = c_subj_op_off[s];
    const int64_t o1 = c_subj_op_off[s + 1];
    int64_t row = c_subj_obs_off[s];
    double x[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) x[i] = 0.0;
    double xpad = 0.0;
    double ll_acc = 0.0;
    uint8_t st = st_lane;
    uint8_t st_sticky = PMX_PAIR_OK;  // DYN: first failure of an EARLIER occasion (the reference errors out for the whole subject)
    (void)st_sticky;
    // status bytes: no memset precedes the launch.  mode 1 (n_support % 8 == 0, aligned array): the wave clears this
    // subject's 64 bytes with 8 lanes x 8 bytes and only failures are written later; mode 2: every pair's byte is written.
    if (zero_status == 1 && status != nullptr) {
      const uint32_t zl = threadIdx.x & 63u;
      const int64_t zp = static_cast<int64_t>(ptile) * tile + (threadIdx.x & ~63u) + 8 * zl;
      if (zl < 8u && zp < P) *reinterpret_cast<uint64_t*>(status + s * P + zp) = 0ull;
    }
    for (int64_t o = o0; o < o1; ++o) {
      const uint32_t meta = c_op_meta[o];
      const uint32_t kind = meta & 0xffu;
      const int io = static_cast<int>((meta >> 8) & 0xffffu);
      const double a = c_op_a[o];
      const double* cov = ops.op_fac + o * (m.n_derived * PMX_MAX_FACTORS);  // this op's covariate factors
      if (kind == OP_PROP) {
        const double r = c_op_b[o];
        if constexpr (LAG) {
          lag_prop<LM::ST, NS>(m, ops, ls, c_op_t0[o], c_op_t1[o], r, L.coef, th, x);
        } else if constexpr (DYN) {
          // bits 24-26: 0 = build; 1 + k = build and keep in slot k; 1 + S + k = take slot k (same length, same
          // covariate factors earlier in this occasion: the same transition matrix).  Wave-uniform: scalar branches.
          const uint32_t rc = (meta >> 24) & 7u;
          const uint32_t n_slots = static_cast<uint32_t>(prop_slots);
          typename LM::S::Prop pr;
          if (rc > n_slots) {  // (kept by a segment of the same kind: with a rate -> F and J, without -> F only)
            double tmp[NPD];
#pragma unroll
            for (int k = 0; k < NPD; ++k) tmp[k] = prop_cache[((rc - 1u - n_slots) * NPD + k) * tile + threadIdx.x];
            __builtin_memcpy(&pr, tmp, sizeof(pr));
          } else {
            double q[LM::NKP];
            lane_params_dyn<KID>(m, L, cov, q);
            const bool ok = (r != 0.0) ? make_prop_dyn<LM::ST, true>(q, a, pr) : make_prop_dyn<LM::ST, false>(q, a, pr);
            if (!ok) st = PMX_PAIR_COMPLEX_ROOTS;
            if (rc != 0u) {
              double tmp[NPD];
              __builtin_memcpy(tmp, &pr, sizeof(pr));
#pragma unroll
              for (int k = 0; k < NPD; ++k) prop_cache[((rc - 1u) * NPD + k) * tile + threadIdx.x] = tmp[k];
            }
          }
          if (r != 0.0) LM::S::apply(pr, x, r);
          else LM::S::apply0(pr, x);
        } else {
          // exponential ladder (pmx_compile.cpp ladder_code): bits 27-29 relate this PROP's length to the previous one's
          const uint32_t rung = (meta >> 27) & 7u;
          if (rung == 0u) {
            LM::S::exps(L.coef, a, ex);
          } else if (rung != 1u) {
            ladder_pow<LM::S::NE>(ex, rung);
          }
          step_from_exps<LM::ST>(L.coef, ex, x, r);
        }
        xpad = 0.0;  // pm_* wrappers re-pad slot 0 with 0 after every kernel call (analytical/mod.rs:70-75)
      } else if (kind == OP_OBS) {
        if constexpr (LAG) {  // no PROP step in front of this observation: lagged boluses may land before it (bit 31)
          if (meta >> 31) lag_flush_before<NS>(m, ops, ls, a, th, x);
        }
        double y = lane_out<KID>(m, L, x, xpad, io, cov);
        if (st == PMX_PAIR_COMPLEX_ROOTS || st == PMX_PAIR_BAD_LAG) y = nanv;
        if constexpr (LL) {
          ll_accumulate(as_const(ops.ll_obs) + row * 4, y, ll_acc);  // row is wave-uniform: scalar fetches
        } else {
          if (st == PMX_PAIR_OK && !isfinite(y)) st = PMX_PAIR_NONFINITE;
          if (lane_ok) pred[row * ld + p] = y;
        }
        ++row;
      } else if (kind == OP_BOLUS) {
        const int k = io - m.pm;
        const double amt = a * fa_of(m, th, io);
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] += (i == k) ? amt : 0.0;
        if (m.pm && io == 0) xpad += amt;
      } else {  // OP_RESET
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] = io ? L.xinit[i] : 0.0;
        xpad = 0.0;
        if constexpr (DYN) {  // a new occasion re-derives its coefficients: its rows are finite again, the pair stays failed
          if (st_sticky == PMX_PAIR_OK) st_sticky = st;
          st = st_lane;
        }
        if constexpr (LAG)
          lag_open_occasion<LM::ST, NS>(m, ops, ls, static_cast<int64_t>(a), c_op_t0[o], L.coef, th, x);
      }
    }
    if constexpr (DYN) {
      if (st_sticky != PMX_PAIR_OK) st = st_sticky;
    }
    if constexpr (LL) {
      if (st == PMX_PAIR_OK && !isfinite(ll_acc)) st = PMX_PAIR_NONFINITE;  // NonFiniteLikelihood (prediction.rs:119-124)
      if (lane_ok) ops.ll_out[s * ops.ll_ld + p] = (st == PMX_PAIR_OK || st == PMX_PAIR_NONFINITE) ? ll_acc : nanv;
    }
    if (status != nullptr && lane_ok && (st != PMX_PAIR_OK || zero_status == 2)) {
      if (zero_status == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clearing store above lands first
      status[s * P + p] = st;
    }
  }
}

// ------------------------------------------------------------------------------------
// Three-compartment structures with covariate-derived rate constants on a population WITHOUT infusions (oral / bolus
// dosing: C5).  Every segment rebuilds its propagator and applies it once, so no transition matrix is formed: the
// matrix-free step of pmx_structures.hpp (ThreeNewton) - eigenvalues, the divided differences of exp(-l dt) on them and
// three sparse matrix-vector products.  Same lane mapping, op stream, kept-propagator codes and status rules as
// pmx_analytical_grid<KID, dyn>; a kept segment holds 6-7 numbers per lane in LDS instead of 12-16.  The host picks it
// when the compiled stream holds no PROP with a rate (LaunchArgs::no_rates) and the model has no pm_ pad slot.
// ------------------------------------------------------------------------------------
#ifndef PMX_DYN3_WAVES
#define PMX_DYN3_WAVES 3
#endif
// EIGR: the stream marks segments whose rate constants equal those of the occasion's previous built segment (bit 27: a
// subject-constant covariate) - the eigenvalues are kept in registers and only the divided differences are rebuilt
// (C5 with one wt per subject 10.6 -> 9.5 ms).  Its own instantiation: carrying the six registers and the second copy of
// the rebuild through the time-varying case cost that one 5 % (10.57 -> 11.09 ms).
template <int KID, bool LL, bool EIGR>
__global__ __launch_bounds__(kBlock, PMX_DYN3_WAVES) void pmx_analytical_dyn3(DevModel m, DevOps ops, const double* __restrict__ theta,
                                                                             int64_t P, int64_t S, int32_t s_chunk, int32_t n_ptiles,
                                                                             double* __restrict__ pred, int64_t ld,
                                                                             uint8_t* __restrict__ status,
                                                                             const int32_t* __restrict__ subj_list, int32_t zero_status,
                                                                             int32_t prop_slots) {
  using LM = LaneModel<KID>;
  constexpr int NS = LM::NS;
  constexpr int ND0 = LM::S::ND0;
  const int64_t b = blockIdx.x;
  const int32_t ptile = static_cast<int32_t>(b % n_ptiles);
  const int64_t chunk = b / n_ptiles;
  const uint32_t tile = blockDim.x;
  const int64_t p = static_cast<int64_t>(ptile) * tile + threadIdx.x;
  const bool lane_ok = p < P;
  const int64_t pc = lane_ok ? p : (P - 1);
  const double* __restrict__ th = theta + pc * m.nparams;
  extern __shared__ double prop_cache[];  // [slot][ND0][lane]
  (void)prop_cache;

  LM L;
  lane_setup<KID, true>(m, th, L);
  const double nanv = __longlong_as_double(0x7ff8000000000000LL);
  const auto c_subj_op_off = as_const(ops.subj_op_off);
  const auto c_subj_obs_off = as_const(ops.subj_obs_off);

  const int64_t s_begin = chunk * s_chunk;
  const int64_t s_end = (s_begin + s_chunk < S) ? (s_begin + s_chunk) : S;
  // An op = its 4-byte meta word + its 64-byte record {factor of kernel parameter 0..6, op_a} (DevOps::op_kfac), both
  // requested ONE OP AHEAD: a scalar fetch of a line nobody touched before costs about a microsecond, and with one in
  // front of every rebuild the walker waited on the scalar cache more than it computed.  Without a subject list the
  // stream is walked in order, so the look-ahead runs across subjects (a subject's last op requests the next one's first).
  const bool chained = subj_list == nullptr;
  const int64_t o_blk_end = chained ? c_subj_op_off[s_end] : 0;  // (the look-ahead never leaves the ops of this block's subjects:
                                                                 // trailing subjects without ops would otherwise send it one past the stream)
  uint32_t meta_n = 0u;
  u32x16 rec_n = {};
  // ... and so does the subject's header {first op, end op, first row}: the next subject's end op and first row are
  // requested while this one is walked (its first op is this one's end op)
  int64_t o0_n = 0, o1_n = 0, row_n = 0;
  bool primed = false;
  if (chained && s_begin < s_end) {
    o0_n = sload_here<int64_t>(ops.subj_op_off + s_begin);
    o1_n = sload_here<int64_t>(ops.subj_op_off + s_begin + 1);
    row_n = sload_here<int64_t>(ops.subj_obs_off + s_begin);
  }
  for (int64_t si = s_begin; si < s_end; ++si) {
    const int64_t s = subj_list ? static_cast<int64_t>(as_const(subj_list)[si]) : si;
    int64_t o0, o1, row;
    if (chained) {
      o0 = o0_n;
      o1 = o1_n;
      row = row_n;
      asm volatile("" : "+s"(o0), "+s"(o1), "+s"(row));
      const int64_t sn = (si + 1 < s_end) ? s + 1 : s;  // (the last subject of the block requests itself again)
      o0_n = o1;
      o1_n = sload_here<int64_t>(ops.subj_op_off + sn + 1);
      row_n = sload_here<int64_t>(ops.subj_obs_off + sn);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      o0 = c_subj_op_off[s];
      o1 = c_subj_op_off[s + 1];
      row = c_subj_obs_off[s];
    }
    if ((!chained || !primed) && o0 < o1) {  // (chained: once, at the block's first subject that has any op)
      meta_n = sload_here<uint32_t>(ops.op_meta + o0);
      rec_n = sload_here<u32x16>(ops.op_kfac + o0 * 8);
      primed = true;
    }
    double x[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) x[i] = 0.0;
    double ll_acc = 0.0;
    double lprev[3] = {0.0, 0.0, 0.0};  // eigenvalues of the occasion's last built segment (bit 27 of a PROP reuses them)
    bool okprev = true;
    uint8_t st = PMX_PAIR_OK;
    uint8_t st_sticky = PMX_PAIR_OK;  // first failure of an EARLIER occasion (the reference errors out for the whole subject)
    if (zero_status == 1 && status != nullptr) {  // (status protocol: pmx_analytical_grid)
      const uint32_t zl = threadIdx.x & 63u;
      const int64_t zp = static_cast<int64_t>(ptile) * tile + (threadIdx.x & ~63u) + 8 * zl;
      if (zl < 8u && zp < P) *reinterpret_cast<uint64_t*>(status + s * P + zp) = 0ull;
    }
    for (int64_t o = o0; o < o1; ++o) {
      uint32_t meta = meta_n;
      u32x16 rec = rec_n;
      asm volatile("" : "+s"(meta), "+s"(rec));  // (this op's words are in scalar registers from here on)
      {
        int64_t on = o + 1;
        if (on >= o1 && (!chained || on >= o_blk_end)) on = o;  // nothing follows (in this block's range of the stream): request this op again
        meta_n = sload_here<uint32_t>(ops.op_meta + on);
        rec_n = sload_here<u32x16>(ops.op_kfac + on * 8);
        __builtin_amdgcn_sched_barrier(0);
      }
      const uint32_t kind = meta & 0xffu;
      const int io = static_cast<int>((meta >> 8) & 0xffffu);
      auto rec_f64 = [&rec](int k) {
        return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(rec[2 * k + 1]) << 32) | rec[2 * k]));
      };
      const double a = rec_f64(7);
      const double* cov = ops.op_fac + o * (m.n_derived * PMX_MAX_FACTORS);  // (derived volumes: lane_out)
      if (kind == OP_PROP) {
        // bits 24-26: 0 = build; 1 + k = build and keep in slot k; 1 + S + k = take slot k (pmx_compile.cpp)
        const uint32_t rc = (meta >> 24) & 7u;
        const uint32_t n_slots = static_cast<uint32_t>(prop_slots);
        double q[LM::NKP], keep[ND0];
        {
          double kp[LM::NKP];
#pragma unroll
          for (int j = 0; j < LM::NKP; ++j) kp[j] = L.kp_base[j] * rec_f64(j);
          to_native_params<KID>(kp, q);
        }
        if (rc > n_slots) {
#pragma unroll
          for (int k = 0; k < ND0; ++k) keep[k] = prop_cache[((rc - 1u - n_slots) * ND0 + k) * tile + threadIdx.x];
        } else {
          // bit 27: same covariate factor row as the occasion's previous built segment - its eigenvalues still hold
          bool ok;
          if constexpr (EIGR) {
            ok = (meta & (1u << 27)) ? LM::S::template direct0_make<true>(q, a, keep, lprev, okprev)
                                     : LM::S::template direct0_make<false>(q, a, keep, lprev, okprev);
          } else {
            ok = LM::S::template direct0_make<false>(q, a, keep, lprev, okprev);
          }
          if (!ok) st = PMX_PAIR_COMPLEX_ROOTS;
          if (rc != 0u) {
#pragma unroll
            for (int k = 0; k < ND0; ++k) prop_cache[((rc - 1u) * ND0 + k) * tile + threadIdx.x] = keep[k];
          }
        }
        LM::S::direct0_apply(q, keep, x);
      } else if (kind == OP_OBS) {
        double y = lane_out_uniform<KID>(m, L, x, io, cov);
        if (st == PMX_PAIR_COMPLEX_ROOTS) y = nanv;
        if constexpr (LL) {
          ll_accumulate(as_const(ops.ll_obs) + row * 4, y, ll_acc);
        } else {
          if (st == PMX_PAIR_OK && !isfinite(y)) st = PMX_PAIR_NONFINITE;
          if (lane_ok) pred[row * ld + p] = y;
        }
        ++row;
      } else if (kind == OP_BOLUS) {
        const double amt = a * fa_of(m, th, io);
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] += (i == io) ? amt : 0.0;
      } else {  // OP_RESET
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] = io ? L.xinit[i] : 0.0;
        if (st_sticky == PMX_PAIR_OK) st_sticky = st;
        st = PMX_PAIR_OK;
      }
    }
    if (st_sticky != PMX_PAIR_OK) st = st_sticky;
    if constexpr (LL) {
      if (st == PMX_PAIR_OK && !isfinite(ll_acc)) st = PMX_PAIR_NONFINITE;
      if (lane_ok) ops.ll_out[s * ops.ll_ld + p] = (st == PMX_PAIR_OK || st == PMX_PAIR_NONFINITE) ? ll_acc : nanv;
    }
    if (status != nullptr && lane_ok && (st != PMX_PAIR_OK || zero_status == 2)) {
      if (zero_status == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      status[s * P + p] = st;
    }
  }
}

// ------------------------------------------------------------------------------------
// LEAN GRID walker (analytical, plain models): the subjects no class holds - populations without a shared program shape -
// walked from FUSED step records (DevSteps).  Same lane mapping and arithmetic as pmx_analytical_grid<KID, false, false>;
// what goes is everything that kernel carries for the cases it also serves (covariate factors, lag cursors, pm_ pads,
// per-output descriptor look-ups): an observation costs no trip of its own, a step is ONE packed scalar fetch, requested
// a step ahead (the kernel the round-2 profile showed scalar-bound: ~88 SALU + 3 dependent s_loads per event), the
// prediction's address is a scalar row base + the lane's constant byte offset.
// ------------------------------------------------------------------------------------
template <int KID, bool LL>
__global__ __launch_bounds__(kBlock, (LaneModel<KID>::NS <= 2) ? 4 : 2) void pmx_analytical_steps(
    DevModel m, DevOps ops, DevSteps sp, const double* __restrict__ theta, int64_t P, int64_t S, int32_t s_chunk,
    int32_t n_ptiles, double* __restrict__ pred, int64_t ld, uint8_t* __restrict__ status,
    const int32_t* __restrict__ subj_list, int32_t zero_status) {
  using LM = LaneModel<KID>;
  constexpr int NS = LM::NS;
  const int64_t b = blockIdx.x;
  const int32_t ptile = static_cast<int32_t>(b % n_ptiles);
  const int64_t chunk = b / n_ptiles;
  const uint32_t tile = blockDim.x;
  const int64_t p = static_cast<int64_t>(ptile) * tile + threadIdx.x;
  const bool lane_ok = p < P;
  const int64_t pc = lane_ok ? p : (P - 1);  // idle lanes shadow the last support point; their stores are masked
  const double* __restrict__ th = theta + pc * m.nparams;
  const uint32_t poff = static_cast<uint32_t>(pc) * 8u;  // the lane's byte offset inside a prediction row
  const double nanv = __longlong_as_double(0x7ff8000000000000LL);

  LM L;
  lane_setup<KID, false>(m, th, L);
  const uint8_t st_lane = L.ok ? PMX_PAIR_OK : PMX_PAIR_COMPLEX_ROOTS;
  const double inv_vol0 = L.ok ? L.inv_vol[0] : nanv;  // (a lane with complex roots: every prediction NaN)
  const int out_state0 = m.out[0].state;
  double ex[LM::S::NE];  // the lane's exponentials of the last PROP (ladder)
#pragma unroll
  for (int i = 0; i < LM::S::NE; ++i) ex[i] = 0.0;

  const auto c_step_off = as_const(sp.subj_step_off);
  const auto c_obs_off = as_const(ops.subj_obs_off);
  const auto c_rec = as_const(reinterpret_cast<const uint64_t*>(sp.step_rec));

  const int64_t s_begin = chunk * s_chunk;
  const int64_t s_end = (s_begin + s_chunk < S) ? (s_begin + s_chunk) : S;
  for (int64_t si = s_begin; si < s_end; ++si) {
    const int64_t s = subj_list ? static_cast<int64_t>(as_const(subj_list)[si]) : si;
    const int64_t o0 = c_step_off[s];
    const int64_t o1 = c_step_off[s + 1];
    int64_t row = c_obs_off[s];
    char* rowp = reinterpret_cast<char*>(pred + row * ld);  // wave-uniform: stays in scalar registers
    double x[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) x[i] = 0.0;
    double ll_acc = 0.0, nanacc = 0.0;
    if (zero_status == 1 && status != nullptr) {  // (see pmx_analytical_grid)
      const uint32_t zl = threadIdx.x & 63u;
      const int64_t zp = static_cast<int64_t>(ptile) * tile + (threadIdx.x & ~63u) + 8 * zl;
      if (zl < 8u && zp < P) *reinterpret_cast<uint64_t*>(status + s * P + zp) = 0ull;
    }
    // the record of step o is requested while step o - 1 is worked on
    auto recp = c_rec + 4 * o0;
    uint64_t w_n = recp[0], a_n = recp[1], b_n = recp[2];
    const int32_t n_steps = static_cast<int32_t>(o1 - o0);  // (32-bit trip count: the 64-bit compare is a vector instruction)
    for (int32_t k = 0; k < n_steps; ++k) {
      uint64_t w = w_n, ab = a_n, bb = b_n;
      asm volatile("" : "+s"(w), "+s"(ab), "+s"(bb));  // (the wait for this step's record sits here, the next request behind it)
      recp += 4;
      w_n = recp[0];
      a_n = recp[1];
      b_n = recp[2];
      const uint32_t meta = static_cast<uint32_t>(w);
      const uint32_t kind = meta & 0xffu;
      const int io = static_cast<int>((meta >> 8) & 0xffffu);
      const double a = __longlong_as_double(static_cast<int64_t>(ab));
      if (kind == OP_PROP) {
        const double r = __longlong_as_double(static_cast<int64_t>(bb));
        const uint32_t rung = (meta >> 27) & 7u;
        if (rung == 0u) {
          LM::S::exps(L.coef, a, ex);
        } else if (rung != 1u) {
          ladder_pow<LM::S::NE>(ex, rung);
        }
        step_from_exps<LM::ST>(L.coef, ex, x, r);
      } else if (kind == OP_BOLUS) {
        double amt = a;
        if (m.has_fa) amt = a * fa_of(m, th, io);
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] += (i == io) ? amt : 0.0;
      } else if (kind == OP_RESET) {
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] = io ? L.xinit[i] : 0.0;
      }  // (OP_OBS: an observation with no step to ride on - the first op of nothing, or a second one at the same instant)
      if ((meta >> 24) & 1u) {
        const int oq = static_cast<int>((meta >> 25) & 3u);
        int out_state = out_state0;
        double inv_vol = inv_vol0;
        if (oq != 0) {  // outputs beyond the first: rare (see pmx_analytical_classed)
          out_state = m.out[oq].state;
          const int vp = m.out_vol_theta[oq];
          double v = 1.0;
          if (vp >= 0) v = th[vp];
          double iv = 1.0 / v;
          asm volatile("" : "+v"(iv));
          inv_vol = L.ok ? iv : nanv;
        }
        double xs = x[0];
#pragma unroll
        for (int i = 1; i < NS; ++i) xs = (out_state == i) ? x[i] : xs;  // (wave-uniform condition: scalar selects)
        const double y = xs * inv_vol;
        if constexpr (LL) {
          ll_accumulate(as_const(ops.ll_obs) + row * 4, y, ll_acc);
          ++row;
        } else {
          nanacc = fma(y, 0.0, nanacc);  // 0 * y is NaN iff y is not finite: resolved once per subject
          if (lane_ok) __builtin_nontemporal_store(y, reinterpret_cast<double*>(rowp + poff));
          rowp += ld * 8;
        }
      }
    }
    uint8_t st = st_lane;
    if constexpr (LL) {
      if (st == PMX_PAIR_OK && !isfinite(ll_acc)) st = PMX_PAIR_NONFINITE;  // NonFiniteLikelihood (prediction.rs:119-124)
      if (lane_ok) ops.ll_out[s * ops.ll_ld + p] = (st == PMX_PAIR_OK || st == PMX_PAIR_NONFINITE) ? ll_acc : nanv;
    } else {
      if (st == PMX_PAIR_OK && nanacc != nanacc) st = PMX_PAIR_NONFINITE;
    }
    if (status != nullptr && lane_ok && (st != PMX_PAIR_OK || zero_status == 2)) {
      if (zero_status == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clearing store above lands first
      status[s * P + p] = st;
    }
  }
}

// ------------------------------------------------------------------------------------
// CLASSED GRID kernel (analytical): one propagator per (lane, program step), applied to a
// register-resident batch of G subjects that share a dosing/sampling design (pmx_compile.hpp
// ClassPlan).  Per (subject, support point) the arithmetic is the generic kernel's; what goes away
// is recomputing exp(-lambda*dt) for every member, and fetching/decoding the op stream per subject.
// ------------------------------------------------------------------------------------
template <int KID>
struct ClassBatch {
  static constexpr int G = (LaneModel<KID>::NS <= 2) ? 8 : 4;
};

// Emit one observation row for every member of the chunk: y = x[ST] * inv_vol, stored in PAIRS.
// Lanes 0-31 of a wave own the even support points of the wave's 64, lanes 32-63 the odd ones; one
// v_permlane32_swap per dword turns (member A value, member B value) into (two adjacent doubles of A's
// row | two adjacent doubles of B's row), so a pair of members leaves in ONE 16-byte store per lane.
// `slot[h]` is this lane's 16-byte slot in the first prediction row of pair h (lower half-wave: member
// 2h, upper half-wave: member 2h+1).  Non-finite predictions are caught with one FMA per value
// (0*y is NaN iff y is inf/NaN) and resolved to members only in the rare wave that saw one.
template <int ST, int G, int NS>
__device__ __forceinline__ void classed_emit(const double (&x)[G][NS], double inv_vol, double* const (&slot)[G / 2],
                                             int64_t kld, bool upper, bool pair_full, bool pair_half, bool any_half,
                                             int32_t n_live, uint32_t& bad) {
  double nanacc = 0.0;
#pragma unroll
  for (int h = 0; h < G / 2; ++h) {
    if (2 * h < n_live) {  // wave-uniform
      const double ya = x[2 * h][ST] * inv_vol;
      const double yb = x[2 * h + 1][ST] * inv_vol;
      nanacc = fma(ya, 0.0, nanacc);
      nanacc = fma(yb, 0.0, nanacc);
      uint32_t alo = static_cast<uint32_t>(__double_as_longlong(ya));
      uint32_t ahi = static_cast<uint32_t>(static_cast<uint64_t>(__double_as_longlong(ya)) >> 32);
      uint32_t blo = static_cast<uint32_t>(__double_as_longlong(yb));
      uint32_t bhi = static_cast<uint32_t>(static_cast<uint64_t>(__double_as_longlong(yb)) >> 32);
      const auto r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
      double2 v;
      v.x = __longlong_as_double(static_cast<int64_t>((static_cast<uint64_t>(r1[0]) << 32) | r0[0]));
      v.y = __longlong_as_double(static_cast<int64_t>((static_cast<uint64_t>(r1[1]) << 32) | r0[1]));
      const bool row_live = (2 * h + 1 < n_live) || !upper;  // the last pair of a partial chunk has no member B
      double* dst = slot[h] + kld;
      if (row_live && pair_full) {
        typedef double dbl2 __attribute__((ext_vector_type(2)));
        dbl2 vv;
        vv.x = v.x;
        vv.y = v.y;
        // streaming store, cache policy `sc1 nt` (tools/experiments/store_pattern_probe.hip: plain 1.23 ms, nt 1.12, sc1 nt 1.07
        // for this address map); no builtin carries sc1, hence the asm
        asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(dst), "v"(vv) : "memory");
      }
      if (any_half) {  // wave-uniform: only the wave holding the last slot of an odd-length row
        if (row_live && pair_half) *dst = v.x;
      }
    }
  }
  if (__any((nanacc != nanacc) ? 1 : 0)) {  // rare: find out which members
#pragma unroll
    for (int j = 0; j < G; ++j)
      if (j < n_live && !isfinite(x[j][ST] * inv_vol)) bad |= (1u << j);
  }
}

template <int ST, int G, int NS>
__device__ __forceinline__ void classed_emit_state(int out_state, const double (&x)[G][NS], double inv_vol,
                                                   double* const (&slot)[G / 2], int64_t kld, bool upper, bool pair_full,
                                                   bool pair_half, bool any_half, int32_t n_live, uint32_t& bad) {
  if (out_state == ST) {
    classed_emit<ST, G, NS>(x, inv_vol, slot, kld, upper, pair_full, pair_half, any_half, n_live, bad);
  } else if constexpr (ST + 1 < NS) {
    classed_emit_state<ST + 1, G, NS>(out_state, x, inv_vol, slot, kld, upper, pair_full, pair_half, any_half, n_live,
                                      bad);
  }
}

// __launch_bounds__ 2nd argument = waves per SIMD the register allocator must leave room for
// (4 -> at most 128 VGPRs): the kernel is a latency/bandwidth mix and wants the occupancy.
// LAGC: one lagged input (exact classes only).  The members of a class share every bolus TIME, so a lane's lagged
// landing times t + lag(theta) - its split points inside a PROP step - are the same for all G members: one propagator
// per sub-interval still serves the whole batch; only the amounts are the members' own (lag_prop / lag_open_occasion
// of the generic walker, over G states at once).
// CENS (log-likelihood mode): the population holds censored observations; their rows are marked in the chunk blocks
// and folded from their full records (a separate instantiation: the extra branch costs the uncensored kernel 9 %).
// DYNC (with PERDT): covariate-derived rate constants / volumes.  The members share the program shape only; each
// rebuilds its propagator from its own covariate factors (the generic walker's lane_advance_dyn, the plan's facp
// rows) and scales its output by its own volume (lane_out, faco rows).  Complex roots are a member's, per occasion.
template <int KID, bool LL, bool PERDT, bool LAGC = false, bool CENS = false, bool DYNC = false>
__global__ __launch_bounds__(kBlock, (LaneModel<KID>::NS <= 2) ? 4 : 2) void pmx_analytical_classed(
    DevModel m, DevOps ops, DevClassPlan cp, const double* __restrict__ theta, int64_t P, int32_t chunks_per_block,
    int32_t n_ptiles, double* __restrict__ pred, int64_t ld, uint8_t* __restrict__ status) {
  using LM = LaneModel<KID>;
  constexpr int NS = LM::NS;
  constexpr int G = ClassBatch<KID>::G;
  static_assert(G % 2 == 0, "members are stored in pairs");
  // XCD-aware block -> tile map.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8
  // share an XCD and its L2).  The n_ptiles column tiles of one chunk-block together write whole
  // prediction rows; placing them on ONE XCD lets that L2 assemble full rows / whole per-subject
  // regions before write-back instead of scattering 2 KB pieces of every row over n_ptiles L2s.
  // (speed only: any placement is correct)
  const int64_t b = blockIdx.x;
  const int64_t group = b / (8 * n_ptiles);
  const int32_t local = static_cast<int32_t>(b % (8 * n_ptiles));
  const int32_t ptile = local / 8;
  const int64_t cblock = group * 8 + (local % 8);
  // chunk-blocks take chunks cblock, cblock + n_cblocks, ... (grid stride): the blocks resident at one moment then
  // work on neighbouring chunks, which keeps each of the G write fronts compact (see build_class_plan `spread`)
  const int64_t n_cblocks = static_cast<int64_t>(gridDim.x) / n_ptiles;
  // this launch's share of the plan: the chunks with shared step lengths, or (PERDT) the loose ones behind them
  const int64_t c_begin = PERDT ? cp.n_chunks_exact : 0;
  const int64_t c_end = PERDT ? cp.n_chunks : cp.n_chunks_exact;
  if (c_begin + cblock >= c_end) return;
  const uint32_t lane = threadIdx.x & 63u;
  const bool upper = lane >= 32u;
  const int64_t p_even = static_cast<int64_t>(ptile) * kBlock + (threadIdx.x & ~63u) + 2u * (lane & 31u);
  const int64_t p = p_even + (upper ? 1 : 0);
  const bool lane_ok = p < P;
  const int64_t pc = lane_ok ? p : (P - 1);
  const bool pair_full = (p_even + 1) < P;   // this lane's 16-byte slot [p_even, p_even+1] is inside the row
  const bool pair_half = (p_even + 1) == P;  // only its first 8 bytes are (odd n_support, last slot of a row)
  const bool any_half = __any(pair_half ? 1 : 0) != 0;

  typename LM::S::Coef coef;
  double inv_vol0;  // 1/volume of output 0 (NaN for a lane with complex roots: all its predictions are NaN)
  bool lane_good;
  LM Ld;  // DYNC: the lane's base parameters, kept for the per-member rebuilds
  if constexpr (DYNC) {
    lane_setup<KID, true>(m, theta + pc * m.nparams, Ld);
    lane_good = true;
    inv_vol0 = Ld.inv_vol[0];
  } else {
    LM L;
    lane_setup<KID, false>(m, theta + pc * m.nparams, L);
    coef = L.coef;
    lane_good = L.ok;
    inv_vol0 = L.ok ? L.inv_vol[0] : __longlong_as_double(0x7ff8000000000000LL);
  }
  (void)Ld;
  double lagv = 0.0;      // LAGC: this lane's lag time of the lagged input
  bool lane_badlag = false;
  if constexpr (LAGC) {
    lagv = theta[pc * m.nparams + m.lag_param[0]];
    if (lagv != lagv) {  // NaN lag: PMX_PAIR_BAD_LAG, rows NaN (the generic walker's rule; a negative lag is a shift to earlier)
      lane_badlag = true;
      inv_vol0 = __longlong_as_double(0x7ff8000000000000LL);
    }
  }
  const double kInf = __longlong_as_double(0x7ff0000000000000LL);
  (void)kInf;
  // the plan arrays are read-only for the whole launch and every index below is wave-uniform:
  // constant-address-space pointers make these scalar (s_load) fetches
  const auto prog_meta = as_const(cp.prog_meta);
  const auto prog_dt = as_const(cp.prog_dt);
  const auto cls_prog_off = as_const(cp.cls_prog_off);
  const auto chunk_cls = as_const(cp.chunk_cls);
  const auto chunk_n = as_const(cp.chunk_n);
  const auto chunk_val_off = as_const(cp.chunk_val_off);
  const auto chunk_subj = as_const(cp.chunk_subj);
  const auto chunk_row = as_const(cp.chunk_row);
  const auto val = as_const(cp.val);
  const auto dtv = as_const(cp.dtv);
  (void)dtv;
  const double* __restrict__ th = theta + pc * m.nparams;

  (void)chunks_per_block;
  for (int64_t c = c_begin + cblock; c < c_end; c += n_cblocks) {
    const int32_t cls = chunk_cls[c];
    const int32_t n_live = chunk_n[c];
    int64_t voff = chunk_val_off[c];
    const int64_t pb = cls_prog_off[cls];
    const int64_t pe = cls_prog_off[cls + 1];
    int64_t kld = 0;   // (observations emitted so far) * ld
    if (cp.zero_status == 1 && status != nullptr) {
      // The wave clears the status bytes it owns (G members x its 64 support points) with ONE 8-byte store per lane:
      // lane = 8 * member + piece.  A separate memset between two passes cost ~70 us of serialisation per pass, this
      // costs one store per chunk.  (Launcher guarantees n_support % 8 == 0 and G <= 8 when the flag is set.)
      const int zj = static_cast<int>(lane >> 3);
      int64_t zsid = -1;
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int64_t sj = chunk_subj[c * G + j];
        zsid = (zj == j && j < n_live) ? sj : zsid;
      }
      const int64_t zp = static_cast<int64_t>(ptile) * kBlock + (threadIdx.x & ~63u) + 8 * (lane & 7u);
      if (zsid >= 0 && zp < P) *reinterpret_cast<uint64_t*>(status + zsid * P + zp) = 0ull;
    }
    uint64_t plain_obs = 0;  // log-likelihood mode: bit k = observation k is a plain row for every live member
    (void)plain_obs;
    double* slot[G / 2];  // this lane's 16-byte slot in the first prediction row of each member pair
    double ll_acc[G];     // log-likelihood mode: running sum of each member
    int64_t cobs_off = 0;  // log-likelihood mode: the chunk's {value, const, weight} block, advanced per observation
    int64_t kobs = 0;      // ... and how many observations of the program have been folded
    (void)kobs;
    if constexpr (LL) {
      cobs_off = as_const(cp.chunk_obs_off)[c] + 2 * G;  // (behind the chunk's [G] constant sums and [G] flags)
      plain_obs = static_cast<uint64_t>(__double_as_longlong(as_const(cp.cobs)[cobs_off - G]));
#pragma unroll
      for (int j = 0; j < G; ++j) ll_acc[j] = 0.0;
    } else {
      const auto rows = chunk_row + c * G;
#pragma unroll
      for (int h = 0; h < G / 2; ++h) {
        int64_t ra = rows[2 * h] * ld, rb = rows[2 * h + 1] * ld;
        asm volatile("" : "+s"(ra), "+s"(rb));  // (keeps LLVM from selecting between the two ADDRESSES)
        slot[h] = pred + ((upper ? rb : ra) + p_even);
      }
    }
    double x[G][NS];
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
      for (int i = 0; i < NS; ++i) x[j][i] = 0.0;
    uint32_t bad = 0;  // bit j: member j emitted a non-finite prediction
    uint32_t cplx = 0;  // DYNC: bit j: a rebuild of member j found complex eigenvalues in the current occasion
    uint32_t cplx_any = 0, bad_any = 0;  // DYNC: ... in an earlier occasion (status is sticky, the rows are not)
    (void)cplx;
    (void)cplx_any;
    (void)bad_any;
    // the lane's exponentials outlive a step: bits 27-29 of a PROP step say how this step's length relates
    // to the previous PROP's (0 = unrelated: exp(); 1 = equal; n = 2..4: n times as long: ladder_pow)
    double ex[LM::S::NE];
    // LAGC: cursor into the current occasion's list of lagged boluses (relative: the members' lists run in parallel),
    // the list's length and member 0's list (the shared times); reset_voff = the val row with the members' occasions
    int32_t lcur = 0, lcnt = 0;
    int64_t lbase0 = 0, reset_voff = 0;
    // every member advances by dt (per lane) under its own rate / takes its own amount of the lagged bolus at lcur
    auto advance_all = [&](double dt, int64_t rate_off, bool with_rate) {
      typename LM::S::Prop pr;
      make_prop<LM::ST>(coef, dt, pr);
#pragma unroll
      for (int j = 0; j < G; ++j) LM::S::apply(pr, x[j], with_rate ? val[rate_off + j] : 0.0);
    };
    auto bolus_all = [&]() {
      const double f = fa_of(m, th, m.lag_input[0]);
      const int dest = m.lag_dest[0];
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int64_t occ = static_cast<int64_t>(val[reset_voff + j]);  // (padding members: occasion 0, amounts unused)
        const double amt = ops.lagb_amount[as_const(ops.lagb_off)[occ] + lcur] * f;
#pragma unroll
        for (int i = 0; i < NS; ++i) x[j][i] += (i == dest) ? amt : 0.0;
      }
      ++lcur;
    };
    auto lag_tau = [&]() { return (lcur < lcnt) ? (ops.lagb_time[lbase0 + lcur] + lagv) : kInf; };
    (void)advance_all;
    (void)bolus_all;
    (void)lag_tau;
    for (int64_t o = pb; o < pe; ++o, voff += G) {
      const uint32_t meta = prog_meta[o];
      // Log-likelihood mode stores nothing inside this loop; what it waits for is scalar fetches, and fetched where they
      // are used a step has four of them one behind the other (meta -> lengths / rates -> descriptor -> observed values
      // and weights; SQ_WAIT_ANY: ~2000 cycles per wave-step).  So every scalar of the step is requested here, in one go,
      // whether or not the step turns out to need it (a step without an observation reads the next one's values; the
      // chunk's block is followed by the next chunk's, the array by 2 G doubles of slack), and pinned, so that the
      // compiler neither sinks the fetches back to their uses nor splits the wait.
#ifdef PMX_EXP_NO_UPFRONT
      constexpr bool kUpfront = false;
#else
      constexpr bool kUpfront = LL && !LAGC && !DYNC && !PERDT;
#endif
      double up_dt = 0.0, up_v[G], up_l[G], up_y[G], up_w[G];
      (void)up_dt;
      (void)up_v;
      (void)up_l;
      (void)up_y;
      (void)up_w;
      if constexpr (kUpfront) {
        const auto ov = as_const(cp.cobs) + cobs_off;
        if constexpr (!PERDT) up_dt = prog_dt[o];
#pragma unroll
        for (int j = 0; j < G; ++j) {
          up_v[j] = val[voff + j];
          if constexpr (PERDT) up_l[j] = dtv[voff + j];
          up_y[j] = ov[j];
          up_w[j] = ov[G + j];
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
          int64_t bv = __double_as_longlong(up_v[j]), by = __double_as_longlong(up_y[j]), bw = __double_as_longlong(up_w[j]);
          asm volatile("" : "+s"(bv), "+s"(by), "+s"(bw));
          up_v[j] = __longlong_as_double(bv);
          up_y[j] = __longlong_as_double(by);
          up_w[j] = __longlong_as_double(bw);
          if constexpr (PERDT) {
            int64_t bl = __double_as_longlong(up_l[j]);
            asm volatile("" : "+s"(bl));
            up_l[j] = __longlong_as_double(bl);
          }
        }
      }
      const uint32_t kind = meta & 0xffu;
      const int io = static_cast<int>((meta >> 8) & 0xffffu);
      if (kind == OP_PROP) {
        if constexpr (LAGC) {
          // lag_prop: split [t0, t1) at this lane's lagged landing times
          const double t1 = as_const(cp.prog_t1)[o];
          double t = as_const(cp.prog_t0)[o];
          for (;;) {
            const double tau = lag_tau();
            if (!(tau < t1)) break;
            if (tau > t) {
              advance_all(tau - t, voff, true);
              t = tau;
            }
            bolus_all();
          }
          if (t1 > t) advance_all(t1 - t, voff, true);
        } else if constexpr (DYNC) {
          const int64_t nf = cp.n_fac;
#pragma unroll
          for (int j = 0; j < G; ++j) {
            if (!lane_advance_dyn<KID, true>(m, Ld, cp.facp + (voff + j) * nf, x[j], dtv[voff + j], val[voff + j])) cplx |= (1u << j);
            if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
          }
        } else if constexpr (PERDT) {
          // loose chunk: every member has its own step length, hence its own propagator; the members still share
          // the walk through the program (one scalar decode per step instead of G) and the paired stores
          // the step's 2 G scalars (lengths, rates) come in up front with two wide scalar loads: fetched member by member
          // each of the G blocks below began by waiting for its own s_load
          double m_dt[G], m_r[G];
#pragma unroll
          for (int j = 0; j < G; ++j) {
            m_dt[j] = kUpfront ? up_l[j] : dtv[voff + j];
            m_r[j] = kUpfront ? up_v[j] : val[voff + j];
          }
#pragma unroll
          for (int j = 0; j < G; ++j) {
            LM::S::exps(coef, m_dt[j], ex);
            step_from_exps<LM::ST>(coef, ex, x[j], m_r[j]);  // (the member's rate is a scalar: no infusion, no J)
            if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          const uint32_t rung = (meta >> 27) & 7u;
          if (rung == 0u) {
            LM::S::exps(coef, kUpfront ? up_dt : prog_dt[o], ex);
          } else if (rung != 1u) {
            ladder_pow<LM::S::NE>(ex, rung);
          }
          typename LM::S::Prop pr;
          LM::S::from_exps(coef, ex, pr);  // (one propagator per step for G members: splitting off J does not pay here)
#pragma unroll
          for (int j = 0; j < G; ++j) {
            LM::S::apply(pr, x[j], kUpfront ? up_v[j] : val[voff + j]);
            // keep the scheduler from interleaving all G updates (it would hold old and new state of
            // every member at once: +2*NS*G registers, one wave per SIMD less)
            if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else if (kind == OP_BOLUS) {
        const double f = fa_of(m, th, io);  // the lane's bioavailability of this input (1.0 when the model has none)
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const double a = (kUpfront ? up_v[j] : val[voff + j]) * f;
#pragma unroll
          for (int i = 0; i < NS; ++i) x[j][i] += (i == io - m.pm) ? a : 0.0;  // (pm_: model input 1 = kernel state 0)
        }
      } else if (kind == OP_RESET) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          double xi = 0.0;
          if (io && m.has_init && m.init_param[i + m.pm] >= 0) xi = th[m.init_param[i + m.pm]];
          if constexpr (DYNC) {  // a new occasion re-derives its coefficients: its rows are finite again, but the pair
            cplx_any |= cplx;    // stays failed (the reference errors out for the whole subject)
            bad_any |= bad;
            cplx = 0;
            bad = 0;
          }
#pragma unroll
          for (int j = 0; j < G; ++j) x[j][i] = xi;
        }
        if constexpr (LAGC) {
          // lag_open_occasion: point the cursor at this occasion's list, run the boluses that land before the
          // occasion's first remaining event (no infusion can be active there)
          reset_voff = voff;
          const int64_t occ0 = static_cast<int64_t>(val[voff]);
          lbase0 = as_const(ops.lagb_off)[occ0];
          lcnt = static_cast<int32_t>(as_const(ops.lagb_off)[occ0 + 1] - lbase0);
          lcur = 0;
          const double t_first = as_const(cp.prog_t0)[o];
          bool started = false;
          double t = 0.0;
          for (;;) {
            const double tau = lag_tau();
            if (!(tau < t_first)) break;
            if (started && tau > t) advance_all(tau - t, voff, false);
            t = tau;
            started = true;
            bolus_all();
          }
          if (started && t_first > t && t_first < kInf) advance_all(t_first - t, voff, false);
        }
      }  // (kind == OP_OBS: a second observation at the same instant, no state change)
      if ((meta >> 24) & 1u) {  // the observation fused into this step (pmx_compile.cpp build_class_plan)
        if constexpr (LAGC) {
          // no PROP step in front of this observation (bit 31; its time sits in the step's t1 slot): the lagged boluses
          // landing before it come first, without propagation (the members share the landing times)
          if (meta >> 31) {
            const double t_obs = as_const(cp.prog_t1)[o];
            while (lag_tau() < t_obs) bolus_all();
          }
        }
        const int oq = static_cast<int>((meta >> 25) & 3u);
        // (pm_ models: the plan only holds subjects that never dose the pad slot and models that never read it, so
        // kernel state = model state - 1 is all the wrapper amounts to; pmx_compile.cpp build_class_plan)
        int out_state = m.out[0].state - m.pm;
        double inv_vol = inv_vol0;
        if (oq != 0) {  // outputs beyond the first: rare, re-derive the volume instead of keeping 4 live
          // ONE descriptor fetch each for state and volume (the host resolved "theta index behind the volume", derived
          // values without covariate factors included: DevModel::out_vol_theta).  The compiler hoists these scalar
          // fetches in front of the branch, into every observation step of the single-output case: with the six
          // fetches + select chain of a device-side resolution C3 went from 0.83 to 0.97 ms.
          //
          // The volume's load must be CONSUMED inside this block on every path: when the division sat behind an
          // exec-masked skip (lanes with complex roots), the load was still pending at the join and the waitcnt pass
          // put `s_waitcnt vmcnt(0)` into the common emit path - every observation step then waited for all earlier
          // prediction stores (C3 0.83 -> 0.98 ms).  Hence: divide unconditionally, pin the quotient, select after.
          out_state = m.out[oq].state - m.pm;
          const int vp = m.out_vol_theta[oq];
          double v = 1.0;
          if (vp >= 0) v = th[vp];  // (scalar condition)
          double iv = 1.0 / v;
          asm volatile("" : "+v"(iv));
          inv_vol = (lane_good && !lane_badlag) ? iv : __longlong_as_double(0x7ff8000000000000LL);
        }
        if constexpr (LL) {
          // fold the G predictions into the members' sums instead of storing them (ll_accumulate, per member;
          // the observed values and sigma terms are wave-uniform scalar fetches)
          // the step's 3 x G scalars are fetched unconditionally and up front (a few wide s_loads instead of 3 G
          // dependent ones behind the weight test: the kernel was scalar-fetch-latency bound)
          const auto ov = as_const(cp.cobs) + cobs_off;
          double ov_y[G], ov_w[G];
#pragma unroll
          for (int j = 0; j < G; ++j) {
            ov_y[j] = kUpfront ? up_y[j] : ov[j];
            ov_w[j] = kUpfront ? up_w[j] : ov[G + j];
          }
          // one member-observation: d = y_obs - pred ; sum -= w d^2   (the constants come in at the end: csum).  The
          // weight tests are on the BITS (scalar integer compares; a floating-point compare of two SGPR values is a
          // vector instruction); the output's state is picked by a scalar branch around the whole member loop
          auto fold = [&](auto st_c) {
            constexpr int ST = decltype(st_c)::value;
            if constexpr (!DYNC) {
              // the common step: a plain row for every live member - no tests (a scalar branch per member costs more
              // than the three instructions it guards); padding members carry weight 0 and finite states: they add -0
              if (kobs < 63 && ((plain_obs >> kobs) & 1ull)) {
#pragma unroll
                for (int j = 0; j < G; ++j) {
                  const double d = fma(-inv_vol, x[j][ST], ov_y[j]);
                  ll_acc[j] = fma(-(d * ov_w[j]), d, ll_acc[j]);
                }
                return;
              }
            }
#pragma unroll
            for (int j = 0; j < G; ++j) {
              const int64_t wb = __double_as_longlong(ov_w[j]);
              if (wb != 0) {  // wave-uniform; weight 0 = missing observation (or chunk padding)
                if (CENS && wb < 0) {  // censored row (marker from pmx_ll_prepare_chunks): the generic fold on its full record
                  double y = x[j][ST] * inv_vol;
                  if constexpr (DYNC) {
                    y = ((cplx >> j) & 1u) ? __longlong_as_double(0x7ff8000000000000LL)
                                           : lane_out<KID>(m, Ld, x[j], 0.0, oq, cp.faco + (voff + j) * cp.n_fac);
                  }
                  ll_accumulate(as_const(ops.ll_obs) + (chunk_row[c * G + j] + kobs) * 4, y, ll_acc[j]);
                } else {
                  double d;
                  if constexpr (DYNC) {
                    const double y = ((cplx >> j) & 1u) ? __longlong_as_double(0x7ff8000000000000LL)
                                                        : lane_out<KID>(m, Ld, x[j], 0.0, oq, cp.faco + (voff + j) * cp.n_fac);
                    d = ov_y[j] - y;
                  } else {
                    d = fma(-inv_vol, x[j][ST], ov_y[j]);
                  }
                  ll_acc[j] = fma(-(d * ov_w[j]), d, ll_acc[j]);
                }
              }
            }
          };
          if constexpr (DYNC) {
            fold(std::integral_constant<int, 0>{});  // (lane_out picks the state itself)
          } else {
            if (out_state == 0) fold(std::integral_constant<int, 0>{});
            if constexpr (NS > 1) {
              if (out_state == 1) fold(std::integral_constant<int, 1>{});
            }
            if constexpr (NS > 2) {
              if (out_state == 2) fold(std::integral_constant<int, 2>{});
            }
            if constexpr (NS > 3) {
              if (out_state == 3) fold(std::integral_constant<int, 3>{});
            }
          }
          cobs_off += 2 * G;
          ++kobs;
        } else {
          if constexpr (DYNC) {
            double ys[G][1];  // each member's prediction under its own volume (NaN while its occasion has complex roots)
#pragma unroll
            for (int j = 0; j < G; ++j)
              ys[j][0] = ((cplx >> j) & 1u) ? __longlong_as_double(0x7ff8000000000000LL)
                                            : lane_out<KID>(m, Ld, x[j], 0.0, oq, cp.faco + (voff + j) * cp.n_fac);
            classed_emit<0, G, 1>(ys, 1.0, slot, kld, upper, pair_full, pair_half, any_half, n_live, bad);
          } else {
            // wave-uniform: the state is picked by a scalar branch, not per-lane selects
            classed_emit_state<0, G, NS>(out_state, x, inv_vol, slot, kld, upper, pair_full, pair_half, any_half, n_live,
                                         bad);
          }
          kld += ld;
        }
      }
    }
    if constexpr (DYNC) {
      cplx |= cplx_any;
      bad |= bad_any;
    }
    if constexpr (LL) {
#pragma unroll
      for (int j = 0; j < G; ++j) {
        if (j < n_live) {
          const int64_t sid = chunk_subj[c * G + j];
          const double llj = ll_acc[j] + as_const(cp.cobs)[as_const(cp.chunk_obs_off)[c] + j];  // + the member's constants
          if (!isfinite(llj)) bad |= (1u << j);  // NonFiniteLikelihood (prediction.rs:119-124)
          if (lane_ok) ops.ll_out[sid * ops.ll_ld + p] = llj;  // (NaN already for a lane with complex roots)
        }
      }
    }
    // status bytes: the library zeroes the array before the launch (PMX_PAIR_OK == 0); only failures are
    // written here, so the healthy case issues no byte stores at all
    if (status != nullptr && (cp.zero_status == 2 || __any(((bad != 0u || cplx != 0u || !lane_good || lane_badlag) && lane_ok) ? 1 : 0))) {
      if (cp.zero_status == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clearing store above lands first
#pragma unroll
      for (int j = 0; j < G; ++j) {
        if (j < n_live) {
          const int64_t sid = chunk_subj[c * G + j];
          const uint8_t st = (!lane_good || ((cplx >> j) & 1u)) ? PMX_PAIR_COMPLEX_ROOTS
                             : (lane_badlag ? PMX_PAIR_BAD_LAG : (((bad >> j) & 1u) ? PMX_PAIR_NONFINITE : PMX_PAIR_OK));
          if (lane_ok && (st != PMX_PAIR_OK || cp.zero_status == 2)) status[sid * P + p] = st;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// CLASSED log-likelihood kernel, exact classes of plain models - the entry NPAG calls (log_likelihood_matrix,
// likelihood/matrix.rs:52-106).  Same arithmetic per (subject, support point) as pmx_analytical_classed<KID, true>; what
// is different is how a wave gets its scalars.  Stamped with s_memtime (tools/ll_stamps.py), the round-2 shape spent 46 %
// of its wave time in the chunk epilogue and 7 % in the chunk header - dependent scalar fetches of data that streams
// from HBM once (subject ids, offsets), one after the other, each a full memory latency - and waited on five scalar
// blocks at the top of every step.  Here
//   * everything a chunk needs is ONE 128-byte record (DevClassPlan::chunk_hdr: program, offsets, masks, the G subject
//     ids), requested a chunk ahead - at the start of the previous chunk's epilogue;
//   * a step's {meta, dt} is one 16-byte record, requested a step ahead; its observation block (G observed values + G
//     weights) one pair of wide fetches requested at the top of the step and first touched behind the state update
//     (volatile fetches pinned by scheduling barriers: the compiler sinks plain ones to their use);
//   * the members' constant sums are requested in front of the run that closes the chunk and added behind it;
//   * runs of steps that are on the exponential ladder, carry a row of output 0 and see no infusion in this chunk are
//     straight-line code (one basic block per step, every value updated in place); a missing observation's weight 0
//     makes its term vanish, so without censored rows (CENS = false) every row qualifies;
//   * a step in which no live member has an infusion running builds F only and advances with apply0;
//   * the lane's initial state and every other per-lane value is in registers before the chunk loop: no vector load
//     (and so no s_waitcnt vmcnt behind the previous chunk's stores) on the common path.
// ------------------------------------------------------------------------------------
#ifdef PMX_LL_STAMPS
__device__ uint64_t g_ll_stamps[5];  // diagnostic build only (tools/ll_stamps.py): cycles per phase, summed over waves
#endif
// Scalar fetches that stay where they are written.  Left to the compiler a request whose only use is the next trip of a
// loop sinks to the end of the trip, a dozen instructions in front of its wait; a VOLATILE fetch is not moved by the
// optimiser, and a scheduling barrier behind it keeps the instruction scheduler from moving it either.  (The waits are the
// compiler's own: it knows these registers are pending.  An earlier form issued the fetches from inline assembly - faster
// to write, but the register allocator may spill or copy an output it believes is already there.)
template <int G>
struct ObsRequest;
template <>
struct ObsRequest<8> {
  typedef u32x16 V;
};
template <>
struct ObsRequest<4> {
  typedef u32x8 V;
};

#ifndef PMX_LL_WAVES
#define PMX_LL_WAVES 3
#endif
template <int KID, bool CENS>
__global__ __launch_bounds__(kBlock, (LaneModel<KID>::NS <= 2) ? PMX_LL_WAVES : 2) void pmx_analytical_classed_ll(
    DevModel m, DevOps ops, DevClassPlan cp, const double* __restrict__ theta, int64_t P, int32_t n_ptiles,
    uint8_t* __restrict__ status) {
  using LM = LaneModel<KID>;
  constexpr int NS = LM::NS;
  constexpr int G = ClassBatch<KID>::G;
  using Req = ObsRequest<G>;
  const int64_t b = blockIdx.x;
  const int64_t group = b / (8 * n_ptiles);
  const int32_t local = static_cast<int32_t>(b % (8 * n_ptiles));
  const int32_t ptile = local / 8;
  const int64_t cblock = group * 8 + (local % 8);
  const int64_t n_cblocks = static_cast<int64_t>(gridDim.x) / n_ptiles;
  const int64_t c_end = cp.n_chunks_exact;
  if (cblock >= c_end) return;
  const uint32_t lane = threadIdx.x & 63u;
  const int64_t p = static_cast<int64_t>(ptile) * kBlock + threadIdx.x;
  const bool lane_ok = p < P;
  const int64_t pc = lane_ok ? p : (P - 1);
  const double* __restrict__ th = theta + pc * m.nparams;
  const double nanv = __longlong_as_double(0x7ff8000000000000LL);

  typename LM::S::Coef coef;
  double inv_vol0;
  bool lane_good;
  double xinit[NS];  // (in registers for the whole launch: see the header comment)
  {
    LM L;
    lane_setup<KID, false>(m, th, L);
    coef = L.coef;
    lane_good = L.ok;
    inv_vol0 = L.ok ? L.inv_vol[0] : nanv;
#pragma unroll
    for (int i = 0; i < NS; ++i) xinit[i] = L.xinit[i];
  }
  const auto chunk_row = as_const(cp.chunk_row);
  const auto val = as_const(cp.val);
  const auto cobs = as_const(cp.cobs);
  (void)chunk_row;
  const int out_state0 = m.out[0].state - m.pm;
  const char* hdr_base = reinterpret_cast<const char*>(cp.chunk_hdr);
  // the lane's slot in its row of the output, ll_out[sid][p], and the row pitch in bytes - the pitch parked in a VECTOR
  // register: as a scalar it does not survive the register pressure of the step loop, and re-fetched from the kernel
  // arguments in front of every store (s_load + s_waitcnt, which also waits for the header request in flight) it made
  // the epilogue half of the wave's time (tools/ll_stamps.py)
  char* const ll_lane = reinterpret_cast<char*>(ops.ll_out + p);
  uint32_t ll_pitch = static_cast<uint32_t>(ops.ll_ld * 8);  // (host-checked: fits 32 bits)
  asm volatile("" : "+v"(ll_pitch));

#ifdef PMX_LL_STAMPS
  uint64_t tp[5] = {0, 0, 0, 0, 0};  // diagnostic build only: shader cycles per phase (header, slow steps, fast runs, epilogue, whole wave)
  const uint64_t t_wave0 = __builtin_amdgcn_s_memtime();
#define PMX_STAMP(i, t_prev)                              \
  {                                                       \
    const uint64_t t_now_ = __builtin_amdgcn_s_memtime(); \
    tp[i] += t_now_ - t_prev;                             \
    t_prev = t_now_;                                      \
  }
#else
#define PMX_STAMP(i, t_prev)
#endif

  // the NEXT chunk's header record, requested a chunk ahead
  u32x16 h_n = sload_here<u32x16>(hdr_base + cblock * 64);
  for (int64_t c = cblock; c < c_end; c += n_cblocks) {
#ifdef PMX_LL_STAMPS
    uint64_t t_ph = __builtin_amdgcn_s_memtime();
#endif
    u32x16 h = h_n;
    asm volatile("" : "+s"(h));
#ifdef PMX_LL_HDR_EARLY
    {
      const int64_t c_next = (c + n_cblocks < c_end) ? (c + n_cblocks) : c;
      h_n = sload_here<u32x16>(hdr_base + c_next * 64);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    // {n_live | n_steps << 16, program offset, val offset, cobs offset, rate mask, class fast mask, subject ids}
    const int32_t n_live = static_cast<int32_t>(h[0] & 0xffffu);
    const int32_t n_steps = static_cast<int32_t>(h[0] >> 16);
    const int64_t pb = static_cast<int64_t>(h[1]);
    int64_t voff = static_cast<int64_t>(h[2]);
    const int64_t cbase = static_cast<int64_t>(h[3]);
    const uint64_t rate_mask = (static_cast<uint64_t>(h[5]) << 32) | h[4];
    uint64_t fast_mask = ((static_cast<uint64_t>(h[7]) << 32) | h[6]) & ~rate_mask & 0x7fffffffffffffffull;
    // the G subject ids wait for the epilogue in ONE vector register (lane j holds member j's): eight scalar registers
    // less across the step loop, whose straight-line runs hold two observation blocks at a time
    int32_t sid_park = 0;
#pragma unroll
    for (int j = 0; j < G; ++j) sid_park = (static_cast<int>(lane) == j) ? static_cast<int32_t>(h[8 + j]) : sid_park;
    int64_t cobs_off = cbase + 2 * G;  // (behind the chunk's [G] constant sums and [G] flags)
    if constexpr (CENS) {
      // censored / residual-model rows take the general fold: only the steps whose row is plain for every live member
      // (the mask by program step, pmx_ll_prepare_chunks) run straight-line.  (a dependent fetch: this variant is the rare one)
      fast_mask &= static_cast<uint64_t>(__double_as_longlong(cobs[cbase + G + 1]));
    }
    if (cp.zero_status == 1 && status != nullptr) {  // (see pmx_analytical_classed)
      const int zj = static_cast<int>(lane >> 3);
      int64_t zsid = -1;
#pragma unroll
      for (int j = 0; j < G; ++j) zsid = (zj == j && j < n_live) ? static_cast<int64_t>(static_cast<int32_t>(h[8 + j])) : zsid;
      const int64_t zp = static_cast<int64_t>(ptile) * kBlock + (threadIdx.x & ~63u) + 8 * (lane & 7u);
      if (zsid >= 0 && zp < P) *reinterpret_cast<uint64_t*>(status + zsid * P + zp) = 0ull;
    }
    double ll_acc[G], x[G][NS];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      ll_acc[j] = 0.0;
#pragma unroll
      for (int i = 0; i < NS; ++i) x[j][i] = 0.0;
    }
    uint32_t bad = 0;
    double ex[LM::S::NE];
#pragma unroll
    for (int i = 0; i < LM::S::NE; ++i) ex[i] = 0.0;
    int32_t kobs = 0;
    bool csum_in = false;  // the members' constant sums (the block behind the chunk's last observation block) are in the sums
    const double* recs = cp.prog_rec + 2 * pb;
    int32_t k = 0;
    PMX_STAMP(0, t_ph)
    while (k < n_steps) {
      const uint64_t run_bits = (k < 63) ? (fast_mask >> k) : 0ull;
      if (run_bits & 1ull) {
        // ---- a run of straight-line steps: x' = F x, fold the row.  The output's state is picked OUTSIDE the loop (one
        // copy of the loop per state) and the ladder is a one-sided branch, so a step is straight-line code that updates
        // every value in place; the record of the NEXT step and its observation block are requested at the step's top.
        int32_t run = __builtin_ctzll(~run_bits);
        if (run > n_steps - k) run = n_steps - k;
        const bool closes = (k + run == n_steps);  // this run holds the chunk's last step
        auto fast_run = [&](auto st_c) {
          constexpr int ST = decltype(st_c)::value;
          // the members' constant sums (head of the chunk's block) ride along when this run closes the chunk: requested
          // here, added behind the loop
          typename Req::V cs_v;
          if (closes) cs_v = sload_here<typename Req::V>(cp.cobs + cbase);
          uint64_t w_n = sload_here<uint64_t>(recs + 2 * k);
#pragma unroll 1
          for (int32_t i = 0; i < run; ++i) {
            // this step's observation block goes out at the top of the step and is first touched behind the state update
            // (keeping TWO blocks in flight - a whole step ahead - cost 32 more scalar registers and the spills that came
            // with them: 0.53 ms against 0.48 on C3); the next step's record goes out a step ahead
            uint64_t w = w_n;
            asm volatile("" : "+s"(w));
            const typename Req::V yc = sload_here<typename Req::V>(cp.cobs + cobs_off);
            const typename Req::V wc = sload_here<typename Req::V>(cp.cobs + cobs_off + G);
            w_n = sload_here<uint64_t>(recs + 2 * (k + i + 1));  // (behind the last program record: one record of padding)
            __builtin_amdgcn_sched_barrier(0);
            cobs_off += 2 * G;
            double ov_y[G], ov_w[G];
#pragma unroll
            for (int j = 0; j < G; ++j) {
              ov_y[j] = __longlong_as_double(static_cast<int64_t>((static_cast<uint64_t>(yc[2 * j + 1]) << 32) | yc[2 * j]));
              ov_w[j] = __longlong_as_double(static_cast<int64_t>((static_cast<uint64_t>(wc[2 * j + 1]) << 32) | wc[2 * j]));
            }
            const uint32_t rung = (static_cast<uint32_t>(w) >> 27) & 7u;  // 1..4
            if (rung != 1u) {
#pragma unroll
              for (int e = 0; e < LM::S::NE; ++e) {
                const double bse = ex[e];
                const double sq = bse * bse;
                double r = sq;
                if (rung != 2u) r = sq * ((rung == 3u) ? bse : sq);
                ex[e] = r;
              }
            }
            typename LM::S::Prop pr;
            LM::S::from_exps_f(coef, ex, pr);
#pragma unroll
            for (int j = 0; j < G; ++j) LM::S::apply0(pr, x[j]);
#pragma unroll
            for (int j = 0; j < G; ++j) {
              const double d = fma(-inv_vol0, x[j][ST], ov_y[j]);
              ll_acc[j] = fma(-(d * ov_w[j]), d, ll_acc[j]);  // (weight 0 = a missing observation: the term vanishes)
            }
          }
          if (closes) {
#pragma unroll
            for (int j = 0; j < G; ++j)
              ll_acc[j] += __longlong_as_double(static_cast<int64_t>((static_cast<uint64_t>(cs_v[2 * j + 1]) << 32) | cs_v[2 * j]));
          }
        };
        if (out_state0 == 0) fast_run(std::integral_constant<int, 0>{});
        if constexpr (NS > 1) {
          if (out_state0 == 1) fast_run(std::integral_constant<int, 1>{});
        }
        if constexpr (NS > 2) {
          if (out_state0 == 2) fast_run(std::integral_constant<int, 2>{});
        }
        if constexpr (NS > 3) {
          if (out_state0 == 3) fast_run(std::integral_constant<int, 3>{});
        }
        kobs += run;
        voff += static_cast<int64_t>(run) * G;
        k += run;
        csum_in = closes;
        PMX_STAMP(2, t_ph)
        continue;
      }
      // ---- any other step: the general form
      const auto rp = as_const(reinterpret_cast<const uint64_t*>(recs)) + 2 * k;
      const uint64_t w = rp[0], dtb = rp[1];
      const uint32_t meta = static_cast<uint32_t>(w);
      const uint32_t kind = meta & 0xffu;
      const int io = static_cast<int>((meta >> 8) & 0xffffu);
      const bool has_val = ((rate_mask >> (k < 63 ? k : 63)) & 1ull) != 0ull;
      if (kind == OP_PROP) {
        const uint32_t rung = (meta >> 27) & 7u;
        if (rung == 0u) {
          LM::S::exps(coef, __longlong_as_double(static_cast<int64_t>(dtb)), ex);
        } else if (rung != 1u) {
          ladder_pow<LM::S::NE>(ex, rung);
        }
        typename LM::S::Prop pr;
        LM::S::from_exps_f(coef, ex, pr);
#pragma unroll
        for (int j = 0; j < G; ++j) LM::S::apply0(pr, x[j]);
        if (has_val) {  // wave-uniform: somebody infuses - the response to the members' rates on top
          LM::S::from_exps_j(coef, ex, pr);
#pragma unroll
          for (int j = 0; j < G; ++j) LM::S::add_j(pr, x[j], val[voff + j]);
        }
      } else if (kind == OP_BOLUS) {
        double f = fa_of(m, th, io);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(f)::"memory");  // (the lane's fa is consumed HERE, not behind the join)
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const double a = val[voff + j] * f;
#pragma unroll
          for (int i = 0; i < NS; ++i) x[j][i] += (i == io - m.pm) ? a : 0.0;
        }
      } else if (kind == OP_RESET) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const double xi = io ? xinit[i] : 0.0;
#pragma unroll
          for (int j = 0; j < G; ++j) x[j][i] = xi;
        }
      }
      if ((meta >> 24) & 1u) {  // the observation fused into this step
        const auto ov = cobs + cobs_off;
        double ov_y[G], ov_w[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
          ov_y[j] = ov[j];
          ov_w[j] = ov[G + j];
        }
        const int oq = static_cast<int>((meta >> 25) & 3u);
        int out_state = out_state0;
        double inv_vol = inv_vol0;
        if (oq != 0) {  // outputs beyond the first: rare (see pmx_analytical_classed for why it is written this way)
          out_state = m.out[oq].state - m.pm;
          const int vp = m.out_vol_theta[oq];
          double v = 1.0;
          if (vp >= 0) v = th[vp];
          double iv = 1.0 / v;
          asm volatile("" : "+v"(iv));
          inv_vol = lane_good ? iv : nanv;
        }
        auto fold = [&](auto st_c) {
          constexpr int ST = decltype(st_c)::value;
#pragma unroll
          for (int j = 0; j < G; ++j) {
            const int64_t wb = __double_as_longlong(ov_w[j]);
            if (wb != 0) {  // wave-uniform; weight 0 = missing observation (or chunk padding)
              if (CENS && wb < 0) {  // censored / residual-model row: the generic fold on its full record
                ll_accumulate(as_const(ops.ll_obs) + (chunk_row[c * G + j] + kobs) * 4, x[j][ST] * inv_vol, ll_acc[j]);
              } else {
                const double d = fma(-inv_vol, x[j][ST], ov_y[j]);
                ll_acc[j] = fma(-(d * ov_w[j]), d, ll_acc[j]);
              }
            }
          }
        };
        if (out_state == 0) fold(std::integral_constant<int, 0>{});
        if constexpr (NS > 1) {
          if (out_state == 1) fold(std::integral_constant<int, 1>{});
        }
        if constexpr (NS > 2) {
          if (out_state == 2) fold(std::integral_constant<int, 2>{});
        }
        if constexpr (NS > 3) {
          if (out_state == 3) fold(std::integral_constant<int, 3>{});
        }
        cobs_off += 2 * G;
        ++kobs;
      }
      voff += G;
      ++k;
      PMX_STAMP(1, t_ph)
    }
    // ---- epilogue.  The next chunk's header goes out first: the stores below cover its fetch.
#ifndef PMX_LL_HDR_EARLY
    {
      const int64_t c_next = (c + n_cblocks < c_end) ? (c + n_cblocks) : c;
      h_n = sload_here<u32x16>(hdr_base + c_next * 64);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    if (!csum_in) {  // (the chunk ended in a general step: fetch the constant sums now)
      const auto cs = cobs + cbase;
#pragma unroll
      for (int j = 0; j < G; ++j) ll_acc[j] += cs[j];
    }
    double nanacc = 0.0;
#pragma unroll
    for (int j = 0; j < G; ++j) {
      if (j < n_live) {  // wave-uniform
        const double llj = ll_acc[j];
        nanacc = fma(llj, 0.0, nanacc);  // 0 * v is NaN iff v is not finite: resolved to members only in the rare wave that saw one
        const uint32_t sidj = static_cast<uint32_t>(__builtin_amdgcn_readlane(sid_park, j));
        double* const dst = reinterpret_cast<double*>(ll_lane + static_cast<uint64_t>(sidj) * ll_pitch);  // one v_mad_u64_u32
#if defined(PMX_LL_STORE_PLAIN)
        if (lane_ok) *dst = llj;
#elif !defined(PMX_ABL_NOSTORE)
        // streaming store: the matrix is written once and read by nobody on this device (plain stores, which allocate in L2:
        // 0.600 ms on C3; nt: 0.530 - the wave sat in front of its eight stores for half its time, tools/ll_stamps.py)
        if (lane_ok) __builtin_nontemporal_store(llj, dst);  // (NaN already for a lane with complex roots)
#else
        if (lane_ok && llj == 1.2345e300) *dst = llj;
#endif
      }
    }
    if (status != nullptr) {
      if (__any((nanacc != nanacc) ? 1 : 0)) {  // NonFiniteLikelihood (prediction.rs:119-124)
#pragma unroll
        for (int j = 0; j < G; ++j)
          if (j < n_live && !isfinite(ll_acc[j])) bad |= (1u << j);
      }
      if (cp.zero_status == 2 || __any(((bad != 0u || !lane_good) && lane_ok) ? 1 : 0)) {
        if (cp.zero_status == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clearing store above lands first
#pragma unroll
        for (int j = 0; j < G; ++j) {
          if (j < n_live) {
            const uint8_t st = !lane_good ? PMX_PAIR_COMPLEX_ROOTS : (((bad >> j) & 1u) ? PMX_PAIR_NONFINITE : PMX_PAIR_OK);
            if (lane_ok && (st != PMX_PAIR_OK || cp.zero_status == 2))
              status[static_cast<int64_t>(__builtin_amdgcn_readlane(sid_park, j)) * P + p] = st;
          }
        }
      }
    }
    PMX_STAMP(3, t_ph)
  }
#ifdef PMX_LL_STAMPS
  tp[4] = __builtin_amdgcn_s_memtime() - t_wave0;
  if ((threadIdx.x & 63u) == 0u)
    for (int i = 0; i < 5; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(&g_ll_stamps[i]), static_cast<unsigned long long>(tp[i]));
#endif
#undef PMX_STAMP
}

// ------------------------------------------------------------------------------------
// PAIR kernel (analytical): lane = (subject, support point), divergent schedules
// ------------------------------------------------------------------------------------
template <int KID, bool DYN, bool LAG, bool LL>
__global__ __launch_bounds__(kBlock) void pmx_analytical_pair(DevModel m, DevOps ops, const double* __restrict__ theta,
                                                              int64_t P, int64_t S, int32_t batch,
                                                              double* __restrict__ pred, int64_t ld,
                                                              uint8_t* __restrict__ status) {
  using LM = LaneModel<KID>;
  constexpr int NS = LM::NS;
  const int64_t n_pairs = batch ? S : S * P;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const bool lane_ok = i < n_pairs;
  const int64_t ic = lane_ok ? i : (n_pairs - 1);
  const int64_t s = ops.subj_order[batch ? ic : (ic / P)];
  const int64_t p = batch ? 0 : (ic % P);
  const double* __restrict__ th = theta + (batch ? s : p) * m.nparams;

  LM L;
  lane_setup<KID, DYN>(m, th, L);
  uint8_t st_lane0 = L.ok ? PMX_PAIR_OK : PMX_PAIR_COMPLEX_ROOTS;
  LagState ls;
  if constexpr (LAG) {
#pragma unroll
    for (int k = 0; k < kMaxLagSlots; ++k) {
      ls.lag[k] = (k < m.n_lag_slots) ? th[m.lag_param[k]] : 0.0;
      ls.cur[k] = ls.end[k] = 0;
      // a negative lag moves the bolus EARLIER, like the reference's `time += l` (structs.rs:629-634); NaN is flagged
      if (k < m.n_lag_slots && ls.lag[k] != ls.lag[k] && st_lane0 == PMX_PAIR_OK) st_lane0 = PMX_PAIR_BAD_LAG;
    }
  }
  const uint8_t st_lane = st_lane0;
  const double nanv = __longlong_as_double(0x7ff8000000000000LL);

  int64_t o = ops.subj_op_off[s];
  const int64_t o1 = lane_ok ? ops.subj_op_off[s + 1] : o;  // idle lanes have an empty stream
  int64_t row = ops.subj_obs_off[s];
  double x[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) x[k] = 0.0;
  double xpad = 0.0;
  double ll_acc = 0.0;
  uint8_t st = st_lane;
  uint8_t st_sticky = PMX_PAIR_OK;  // DYN: first failure of an earlier occasion (see the GRID kernel)
  (void)st_sticky;
  // exec-masked loop: runs while ANY lane of the wave still has ops (each lane exits at its own o1).  Every lane reads
  // its own op, so a fetch is a 64-line gather with nothing to hide its latency behind when the batch is a few
  // thousand pairs (C2: 157 waves on 1024 SIMDs); the ops come as packed 32-byte records (DevOps::op_rec), four
  // at a time: one memory latency per four ops.
  const double4* __restrict__ recs = reinterpret_cast<const double4*>(ops.op_rec);
  for (int64_t og = o; og < o1; og += 4) {
    const int64_t last = o1 - 1;
    double4 q = recs[og];
    double4 q1 = recs[(og + 1 < o1) ? og + 1 : last];
    double4 q2 = recs[(og + 2 < o1) ? og + 2 : last];
    double4 q3 = recs[(og + 3 < o1) ? og + 3 : last];
#pragma unroll 1
    for (int j = 0; j < 4; ++j, q = q1, q1 = q2, q2 = q3) {  // (rotating the records keeps them in registers)
      o = og + j;
      if (o >= o1) break;
      const uint32_t meta = static_cast<uint32_t>(__double_as_longlong(q.x));
      const uint32_t kind = meta & 0xffu;
      const int io = static_cast<int>((meta >> 8) & 0xffffu);
      const double a = q.y;
      const double* cov = ops.op_fac + o * (m.n_derived * PMX_MAX_FACTORS);  // this op's covariate factors
      if (kind == OP_PROP) {
        const double r = q.z;
        if constexpr (LAG) {
          lag_prop<LM::ST, NS>(m, ops, ls, q.w, ops.op_t1[o], r, L.coef, th, x);
        } else if constexpr (DYN) {
          if (!lane_advance_dyn<KID>(m, L, cov, x, a, r)) st = PMX_PAIR_COMPLEX_ROOTS;
        } else {
          advance<LM::ST>(L.coef, x, a, r);
        }
        xpad = 0.0;
      } else if (kind == OP_OBS) {
        if constexpr (LAG) {  // (see the GRID kernel)
          if (meta >> 31) lag_flush_before<NS>(m, ops, ls, a, th, x);
        }
        double y = lane_out<KID>(m, L, x, xpad, io, cov);
        if (st == PMX_PAIR_COMPLEX_ROOTS || st == PMX_PAIR_BAD_LAG) y = nanv;
        if constexpr (LL) {
          ll_accumulate(ops.ll_obs + row * 4, y, ll_acc);
        } else {
          if (st == PMX_PAIR_OK && !isfinite(y)) st = PMX_PAIR_NONFINITE;
          pred[row * ld + p] = y;
        }
        ++row;
      } else if (kind == OP_BOLUS) {
        const int k = io - m.pm;
        const double amt = a * fa_of(m, th, io);
#pragma unroll
        for (int jj = 0; jj < NS; ++jj) x[jj] += (jj == k) ? amt : 0.0;
        if (m.pm && io == 0) xpad += amt;
      } else {
#pragma unroll
        for (int jj = 0; jj < NS; ++jj) x[jj] = io ? L.xinit[jj] : 0.0;
        xpad = 0.0;
        if constexpr (DYN) {
          if (st_sticky == PMX_PAIR_OK) st_sticky = st;
          st = st_lane;
        }
        if constexpr (LAG) lag_open_occasion<LM::ST, NS>(m, ops, ls, static_cast<int64_t>(a), q.w, L.coef, th, x);
      }
    }
  }
  if constexpr (DYN) {
    if (st_sticky != PMX_PAIR_OK) st = st_sticky;
  }
  if constexpr (LL) {
    if (st == PMX_PAIR_OK && !isfinite(ll_acc)) st = PMX_PAIR_NONFINITE;
    if (lane_ok) ops.ll_out[batch ? s : (s * ops.ll_ld + p)] = (st == PMX_PAIR_OK || st == PMX_PAIR_NONFINITE) ? ll_acc : nanv;
  }
  if (status != nullptr && lane_ok) status[batch ? s : (s * P + p)] = st;  // every pair writes its byte: no memset before the launch
}

// ------------------------------------------------------------------------------------
// ODE: built-in diffeq bodies (the walkers and the RK4 stepper are in pmx_ode.hpp)
// ------------------------------------------------------------------------------------
template <int MODEL>
struct OdeModel;

template <>
struct OdeModel<PMX_ODE_ONE_CMT_IV> {  // examples/ode_readme.rs:17-19
  static constexpr bool CUSTOM = false;
  static constexpr int NS = 1, NP = 1, CENTRAL = 0;
  static constexpr int NR = 1;
  __device__ __forceinline__ static void rhs(const double* p, const double (&x)[NS], double (&dx)[NS]) {
    dx[0] = -p[0] * x[0];
  }
};
template <>
struct OdeModel<PMX_ODE_ONE_CMT_ORAL> {
  static constexpr bool CUSTOM = false;
  static constexpr int NS = 2, NP = 2, CENTRAL = 1;
  static constexpr int NR = 2;
  __device__ __forceinline__ static void rhs(const double* p, const double (&x)[NS], double (&dx)[NS]) {
    dx[0] = -p[0] * x[0];
    dx[1] = p[0] * x[0] - p[1] * x[1];
  }
};
template <>
struct OdeModel<PMX_ODE_TWO_CMT_IV> {  // two_compartment_models.rs:131-136
  static constexpr bool CUSTOM = false;
  static constexpr int NS = 2, NP = 3, CENTRAL = 0;
  static constexpr int NR = 2;
  __device__ __forceinline__ static void rhs(const double* p, const double (&x)[NS], double (&dx)[NS]) {
    dx[0] = -p[0] * x[0] - p[1] * x[0] + p[2] * x[1];
    dx[1] = p[1] * x[0] - p[2] * x[1];
  }
};
template <>
struct OdeModel<PMX_ODE_TWO_CMT_ORAL> {  // two_compartment_models.rs:188-194, p=[ke,ka,kcp,kpc]
  static constexpr bool CUSTOM = false;
  static constexpr int NS = 3, NP = 4, CENTRAL = 1;
  static constexpr int NR = 3;
  __device__ __forceinline__ static void rhs(const double* p, const double (&x)[NS], double (&dx)[NS]) {
    dx[0] = -p[1] * x[0];
    dx[1] = -p[0] * x[1] + p[1] * x[0] - p[2] * x[1] + p[3] * x[2];
    dx[2] = p[2] * x[1] - p[3] * x[2];
  }
};
template <>
struct OdeModel<PMX_ODE_THREE_CMT_IV> {
  static constexpr bool CUSTOM = false;
  static constexpr int NS = 3, NP = 5, CENTRAL = 0;
  static constexpr int NR = 3;
  __device__ __forceinline__ static void rhs(const double* p, const double (&x)[NS], double (&dx)[NS]) {
    dx[0] = -(p[0] + p[1] + p[2]) * x[0] + p[3] * x[1] + p[4] * x[2];
    dx[1] = p[1] * x[0] - p[3] * x[1];
    dx[2] = p[2] * x[0] - p[4] * x[2];
  }
};
template <>
struct OdeModel<PMX_ODE_THREE_CMT_ORAL> {
  static constexpr bool CUSTOM = false;
  static constexpr int NS = 4, NP = 6, CENTRAL = 1;
  static constexpr int NR = 4;
  __device__ __forceinline__ static void rhs(const double* p, const double (&x)[NS], double (&dx)[NS]) {
    dx[0] = -p[0] * x[0];
    dx[1] = p[0] * x[0] - (p[1] + p[2] + p[3]) * x[1] + p[4] * x[2] + p[5] * x[3];
    dx[2] = p[2] * x[1] - p[4] * x[2];
    dx[3] = p[3] * x[1] - p[5] * x[3];
  }
};
template <>
struct OdeModel<PMX_ODE_ONE_CMT_MM> {  // p=[vmax,km,v]
  static constexpr bool CUSTOM = false;
  static constexpr int NS = 1, NP = 3, CENTRAL = 0;
  static constexpr int NR = 1;
  __device__ __forceinline__ static void rhs(const double* p, const double (&x)[NS], double (&dx)[NS]) {
    const double cc = x[0] / p[2];
    dx[0] = -p[0] * cc / (p[1] + cc);
  }
};


template <int MODEL, bool LAG, bool LL, bool ADAPT>
__global__ __launch_bounds__(kBlock) void pmx_ode_rk4_grid(DevModel m, DevOps ops, const double* __restrict__ theta,
                                                           int64_t P, int64_t S, int32_t s_chunk, int32_t n_ptiles,
                                                           double* __restrict__ pred, int64_t ld,
                                                           uint8_t* __restrict__ status) {
  ode_grid_body<OdeModel<MODEL>, LAG, LL, ADAPT>(m, ops, theta, P, S, s_chunk, n_ptiles, pred, ld, status);
}

template <int MODEL, bool LAG, bool LL, bool ADAPT>
__global__ __launch_bounds__(kBlock) void pmx_ode_rk4_pair(DevModel m, DevOps ops, const double* __restrict__ theta,
                                                           int64_t P, int64_t S, int32_t batch,
                                                           double* __restrict__ pred, int64_t ld,
                                                           uint8_t* __restrict__ status) {
  ode_pair_body<OdeModel<MODEL>, LAG, LL, ADAPT>(m, ops, theta, P, S, batch, pred, ld, status);
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
// GRID launches: a support grid smaller than one 256-lane tile runs with just the waves it needs (whole waves of
// lanes beyond n_support would otherwise walk every subject for nothing: P = 64 wasted 3 of 4 waves).
inline uint32_t grid_threads(int64_t P) {
  return P <= 64 ? 64u : (P <= 128 ? 128u : static_cast<uint32_t>(kBlock));  // (192-thread blocks measured slower than 256)
}

template <int KID, bool DYN, bool LAG>
hipError_t launch_analytical(const LaunchArgs& a, const char** name) {
  static const char* const kNameGrid = DYN ? "pmx_analytical_grid<dyn>" : (LAG ? "pmx_analytical_grid<lag>" : "pmx_analytical_grid");
  static const char* const kNamePair = DYN ? "pmx_analytical_pair<dyn>" : (LAG ? "pmx_analytical_pair<lag>" : "pmx_analytical_pair");
  hipStream_t st = static_cast<hipStream_t>(a.stream);
  if (a.mode == MODE_GRID) {
    int64_t n_walk = a.S;
    const int32_t* list = nullptr;
    *name = kNameGrid;
    if constexpr (!(DYN && LAG)) {
      if (a.use_classes && a.cls.n_chunks > 0) {
        *name = "pmx_analytical_classed";
        // enough blocks to fill the chip several times over, few enough that lane_setup stays amortised
        // (chunks are taken in grid-stride order; one chunk per block up to 32k blocks measured best: tools/experiments/cpb_on_one_allocation.py)
        // (the log-likelihood variant writes almost nothing: it prefers fewer, longer blocks that amortise the lane setup)
        const bool ll = a.ops.ll_obs != nullptr;
        const int64_t n_exact = a.cls.n_chunks_exact, n_loose = a.cls.n_chunks - a.cls.n_chunks_exact;
        // (the loose launch is FP64-bound too and behaves the same: 4 chunks per block 1.83 ms, one 1.88 ms, eight 1.84 ms
        // on jittered C3, profiles/r02/loose_chunks_per_block.txt)
        auto blocks_for = [&](int64_t n, bool loose, int64_t* cpb_out) {
          int64_t cpb = (n * a.n_ptiles) / ((ll || loose) ? 8192 : 32768);
          if (cpb < 1) cpb = 1;
          if (cpb > (loose && !ll ? 4 : 8)) cpb = loose && !ll ? 4 : 8;
          if (a.tune_cpb > 0) cpb = a.tune_cpb;  // tuning experiments (PMX_TUNE_CPB, read once by pmx_api.cpp)
          *cpb_out = cpb;
          return ((n + cpb - 1) / cpb + 7) / 8 * 8;  // whole XCD groups
        };
        auto launch_cls = [&](auto ll_c, auto perdt_c, auto cens_c, int64_t n) {
          int64_t cpb = 1;
          const int64_t cblocks = blocks_for(n, decltype(perdt_c)::value, &cpb);
          hipLaunchKernelGGL((pmx_analytical_classed<KID, decltype(ll_c)::value, decltype(perdt_c)::value, LAG, decltype(cens_c)::value,
                                                     (DYN && decltype(perdt_c)::value)>),
                             dim3(static_cast<uint32_t>(cblocks * a.n_ptiles)), dim3(grid_threads(a.P)), 0, st, a.m, a.ops, a.cls,
                             a.theta, a.P, static_cast<int32_t>(cpb), a.n_ptiles, a.pred, a.ld, a.status);
        };
        using T = std::true_type;
        using F = std::false_type;
        const bool cens = ll && a.ll_censored != 0;
        if (n_exact > 0) {
          if (LAG) *name = ll ? "pmx_analytical_classed<ll,lag>" : "pmx_analytical_classed<lag>";
          else if (ll) *name = "pmx_analytical_classed<ll>";
          bool done = false;
          if constexpr (!LAG && !DYN) {
            if (ll && a.cls.prog_rec != nullptr && a.cls.chunk_hdr != nullptr && a.tune_ll_old == 0 &&
                a.ops.ll_ld < (int64_t{1} << 28)) {  // exact classes of a plain model: the pipelined kernel
              int64_t cpb = 1;
              const int64_t cblocks = blocks_for(n_exact, false, &cpb);
              if (cens)
                hipLaunchKernelGGL((pmx_analytical_classed_ll<KID, true>), dim3(static_cast<uint32_t>(cblocks * a.n_ptiles)),
                                   dim3(grid_threads(a.P)), 0, st, a.m, a.ops, a.cls, a.theta, a.P, a.n_ptiles, a.status);
              else
                hipLaunchKernelGGL((pmx_analytical_classed_ll<KID, false>), dim3(static_cast<uint32_t>(cblocks * a.n_ptiles)),
                                   dim3(grid_threads(a.P)), 0, st, a.m, a.ops, a.cls, a.theta, a.P, a.n_ptiles, a.status);
              *name = "pmx_analytical_classed_ll";
              done = true;
            }
          }
          if (done) {
          } else if (!ll) launch_cls(F{}, F{}, F{}, n_exact);
          else if (cens) launch_cls(T{}, F{}, T{}, n_exact);
          else launch_cls(T{}, F{}, F{}, n_exact);
        }
        if constexpr (!LAG) {
          if (n_loose > 0) {  // subjects that share a program shape but not its step lengths
            if (n_exact == 0) *name = ll ? "pmx_analytical_classed<ll,loose>" : "pmx_analytical_classed<loose>";
            if (DYN) *name = ll ? "pmx_analytical_classed<ll,dyn>" : "pmx_analytical_classed<dyn>";
            if (!ll) launch_cls(F{}, T{}, F{}, n_loose);
            else if (cens) launch_cls(T{}, T{}, T{}, n_loose);
            else launch_cls(T{}, T{}, F{}, n_loose);
          }
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        n_walk = a.cls.n_generic;
        list = a.cls.generic_subjects;
        if (n_walk == 0) return hipSuccess;
      }
    }
    int32_t s_chunk = a.s_chunk;
    if (list != nullptr) {
      int64_t ch = (n_walk * a.n_ptiles) / 8192;
      s_chunk = static_cast<int32_t>(ch < 1 ? 1 : (ch > 64 ? 64 : ch));
    }
    // tile = support points per block.  With kept propagators (DYN) the LDS cache is sized per lane, so the tile also
    // sets the occupancy: a.dyn_tile (64 / 128 / 256, pmx_api.cpp) was picked for that.
    uint32_t threads = grid_threads(a.P);
    int32_t n_ptiles = a.n_ptiles;
    size_t lds = 0;
    if (DYN && a.prop_slots > 0) {
      if (a.dyn_tile > 0 && static_cast<uint32_t>(a.dyn_tile) < threads) threads = static_cast<uint32_t>(a.dyn_tile);
      n_ptiles = static_cast<int32_t>((a.P + threads - 1) / threads);
      lds = static_cast<size_t>(a.prop_slots) * sizeof(typename LaneModel<KID>::S::Prop) * threads;
    }
    const int64_t n_chunks = (n_walk + s_chunk - 1) / s_chunk;
    const int64_t blocks = n_chunks * n_ptiles;
    if constexpr (!DYN && !LAG) {
      if (a.steps.step_rec != nullptr) {  // plain model: the lean walker over fused step records
        *name = (list != nullptr) ? *name : "pmx_analytical_steps";
        if (a.ops.ll_obs != nullptr)
          hipLaunchKernelGGL((pmx_analytical_steps<KID, true>), dim3(static_cast<uint32_t>(blocks)), dim3(threads), 0, st, a.m, a.ops,
                             a.steps, a.theta, a.P, n_walk, s_chunk, n_ptiles, a.pred, a.ld, a.status, list, a.cls.zero_status);
        else
          hipLaunchKernelGGL((pmx_analytical_steps<KID, false>), dim3(static_cast<uint32_t>(blocks)), dim3(threads), 0, st, a.m, a.ops,
                             a.steps, a.theta, a.P, n_walk, s_chunk, n_ptiles, a.pred, a.ld, a.status, list, a.cls.zero_status);
        return hipGetLastError();
      }
    }
    if constexpr (DYN && !LAG && kHasDirect0<kernel_structure(KID)>) {
      if (a.no_rates && a.m.pm == 0) {  // three-compartment covariate model, no infusion anywhere: the matrix-free walker
        *name = (list != nullptr) ? *name : "pmx_analytical_dyn3";
        const size_t lds3 = (a.prop_slots > 0) ? static_cast<size_t>(a.prop_slots) * LaneModel<KID>::S::ND0 * sizeof(double) * threads : 0;
        const bool ll = a.ops.ll_obs != nullptr, er = a.eig_reuse != 0;
        auto go = [&](auto kern) {
          hipLaunchKernelGGL(kern, dim3(static_cast<uint32_t>(blocks)), dim3(threads), lds3, st, a.m, a.ops, a.theta, a.P, n_walk,
                             s_chunk, n_ptiles, a.pred, a.ld, a.status, list, a.cls.zero_status, a.prop_slots);
        };
        if (ll && er) go(pmx_analytical_dyn3<KID, true, true>);
        else if (ll) go(pmx_analytical_dyn3<KID, true, false>);
        else if (er) go(pmx_analytical_dyn3<KID, false, true>);
        else go(pmx_analytical_dyn3<KID, false, false>);
        return hipGetLastError();
      }
    }
    if (a.ops.ll_obs != nullptr)
      hipLaunchKernelGGL((pmx_analytical_grid<KID, DYN, LAG, true>), dim3(static_cast<uint32_t>(blocks)), dim3(threads), lds, st,
                         a.m, a.ops, a.theta, a.P, n_walk, s_chunk, n_ptiles, a.pred, a.ld, a.status, list, a.cls.zero_status,
                         a.prop_slots);
    else
      hipLaunchKernelGGL((pmx_analytical_grid<KID, DYN, LAG, false>), dim3(static_cast<uint32_t>(blocks)), dim3(threads), lds, st,
                         a.m, a.ops, a.theta, a.P, n_walk, s_chunk, n_ptiles, a.pred, a.ld, a.status, list, a.cls.zero_status,
                         a.prop_slots);
  } else {
    *name = kNamePair;
    const int64_t n_pairs = a.batch ? a.S : a.S * a.P;
    const int64_t blocks = (n_pairs + kBlock - 1) / kBlock;
    if (a.ops.ll_obs != nullptr)
      hipLaunchKernelGGL((pmx_analytical_pair<KID, DYN, LAG, true>), dim3(static_cast<uint32_t>(blocks)), dim3(kBlock), 0, st,
                         a.m, a.ops, a.theta, a.P, a.S, a.batch, a.pred, a.ld, a.status);
    else
      hipLaunchKernelGGL((pmx_analytical_pair<KID, DYN, LAG, false>), dim3(static_cast<uint32_t>(blocks)), dim3(kBlock), 0, st,
                         a.m, a.ops, a.theta, a.P, a.S, a.batch, a.pred, a.ld, a.status);
  }
  return hipGetLastError();
}

template <int MODEL, bool LAG, bool LL, bool ADAPT>
hipError_t launch_ode_v(const LaunchArgs& a, const char** name) {
  hipStream_t st = static_cast<hipStream_t>(a.stream);
  if (a.mode == MODE_GRID) {
    *name = ADAPT ? (LAG ? "pmx_ode_dopri5_grid<lag>" : "pmx_ode_dopri5_grid") : (LAG ? "pmx_ode_rk4_grid<lag>" : "pmx_ode_rk4_grid");
    if (ADAPT && a.m.ode_stiff) *name = LAG ? "pmx_ode_ros2_grid<lag>" : "pmx_ode_ros2_grid";
    const int64_t n_chunks = (a.S + a.s_chunk - 1) / a.s_chunk;
    const int64_t blocks = n_chunks * a.n_ptiles;
    hipLaunchKernelGGL((pmx_ode_rk4_grid<MODEL, LAG, LL, ADAPT>), dim3(static_cast<uint32_t>(blocks)), dim3(grid_threads(a.P)), 0, st,
                       a.m, a.ops, a.theta, a.P, a.S, a.s_chunk, a.n_ptiles, a.pred, a.ld, a.status);
  } else {
    *name = ADAPT ? (LAG ? "pmx_ode_dopri5_pair<lag>" : "pmx_ode_dopri5_pair") : (LAG ? "pmx_ode_rk4_pair<lag>" : "pmx_ode_rk4_pair");
    if (ADAPT && a.m.ode_stiff) *name = LAG ? "pmx_ode_ros2_pair<lag>" : "pmx_ode_ros2_pair";
    const int64_t n_pairs = a.batch ? a.S : a.S * a.P;
    const int64_t blocks = (n_pairs + kBlock - 1) / kBlock;
    hipLaunchKernelGGL((pmx_ode_rk4_pair<MODEL, LAG, LL, ADAPT>), dim3(static_cast<uint32_t>(blocks)), dim3(kBlock), 0, st,
                       a.m, a.ops, a.theta, a.P, a.S, a.batch, a.pred, a.ld, a.status);
  }
  return hipGetLastError();
}

template <int MODEL>
hipError_t launch_ode(const LaunchArgs& a, const char** name) {
  const bool lag = a.m.n_lag_slots > 0, ll = a.ops.ll_obs != nullptr, ad = a.adaptive != 0;
  if (lag) {
    if (ll) return ad ? launch_ode_v<MODEL, true, true, true>(a, name) : launch_ode_v<MODEL, true, true, false>(a, name);
    return ad ? launch_ode_v<MODEL, true, false, true>(a, name) : launch_ode_v<MODEL, true, false, false>(a, name);
  }
  if (ll) return ad ? launch_ode_v<MODEL, false, true, true>(a, name) : launch_ode_v<MODEL, false, true, false>(a, name);
  return ad ? launch_ode_v<MODEL, false, false, true>(a, name) : launch_ode_v<MODEL, false, false, false>(a, name);
}

template <int KID>
hipError_t launch_analytical_k(const LaunchArgs& a, const char** name) {
  if (a.m.n_lag_slots > 0) return launch_analytical<KID, false, true>(a, name);  // (lag + covariate-derived constants is rejected at model_create)
  return a.dyn ? launch_analytical<KID, true, false>(a, name) : launch_analytical<KID, false, false>(a, name);
}

}  // namespace

// ------------------------------------------------------------------------------------
// log-likelihood tables (one thread per observation / per chunk slot)
// ------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void pmx_ll_prepare_obs(LLPrepareArgs a) {
  const int64_t r = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (r >= a.n_obs) return;
  double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
  const double y = a.obs_y[r];
  if (y == y) {  // a valued observation (missing ones keep weight 0)
    const int q = a.obs_outeq[r];
    const pmx_error_model& e = a.em[q < PMX_MAX_OUT ? q : 0];
    if (e.kind < PMX_EM_ADDITIVE || e.kind > PMX_EM_RES_EXPONENTIAL) {
      // no error model for this output (only the batch entry points get here: log_likelihood_batch scores such a subject
      // -inf instead of failing, residual_error.rs:413-425): the row poisons its subject's sum
      const double nanq = __longlong_as_double(0x7ff8000000000000LL);
      a.obs4[r * 4 + 0] = nanq;
      a.obs4[r * 4 + 1] = nanq;
      a.obs4[r * 4 + 2] = 1.0;
      a.obs4[r * 4 + 3] = 0.0;
      return;
    }
    double c0 = e.c[0], c1 = e.c[1], c2 = e.c[2], c3 = e.c[3];
    if (a.obs_poly != nullptr) {  // the observation's own polynomial wins (error_model.rs:1051-1054)
      const double p0 = a.obs_poly[r * 4];
      if (p0 == p0) {
        c0 = p0;
        c1 = a.obs_poly[r * 4 + 1];
        c2 = a.obs_poly[r * 4 + 2];
        c3 = a.obs_poly[r * 4 + 3];
      }
    }
    if (e.kind >= PMX_EM_RES_CONSTANT) {  // residual models: the fold derives sigma from the prediction (ll_residual_term)
      a.obs4[r * 4 + 0] = y;
      a.obs4[r * 4 + 1] = e.scalar;
      a.obs4[r * 4 + 2] = -static_cast<double>(e.kind);
      a.obs4[r * 4 + 3] = e.c[0];
      return;
    }
    const double alpha = c0 + c1 * y + c2 * (y * y) + c3 * (y * y * y);
    const double sigma = (e.kind == PMX_EM_ADDITIVE) ? sqrt(alpha * alpha + e.scalar * e.scalar) : e.scalar * alpha;
    const int cz = a.obs_cens != nullptr ? a.obs_cens[r] : 0;
    const bool bad = !(sigma >= 0.0) || !isfinite(sigma) || (cz != 0 && !(sigma > 0.0));
    q0 = y;
    q1 = -0.5 * 1.8378770664093453 - log(sigma);
    q2 = 1.0 / (2.0 * sigma * sigma);
    q3 = (cz == 0) ? 0.0 : ((cz > 0 ? 1.0 : -1.0) / (sigma * 1.4142135623730951));
    if (bad) {  // NegativeSigma / NonFiniteSigma: the row (and so the subject's sum) becomes NaN
      q0 = q1 = __longlong_as_double(0x7ff8000000000000LL);  // (q0 too: the censored fold reads value and scale only)
      q2 = 1.0;
      atomicAdd(a.err, 1);
    }
  }
  a.obs4[r * 4 + 0] = q0;
  a.obs4[r * 4 + 1] = q1;
  a.obs4[r * 4 + 2] = q2;
  a.obs4[r * 4 + 3] = q3;
}

// A chunk's block of cobs: [G] csum, [G] flags, then [k][2][G] = {observed value, weight} of observation k of member j (0 for padding
// members and missing observations).  A plain row's term is  c - w (y - pred)^2 ; the constants c depend on nothing the
// kernel computes, so they are summed here, once per (error model, population), and the kernel adds csum[j] at the end.
// A censored row (BLOQ / ALOQ) or a residual-model row carries weight -1 as a marker: the kernel takes the row's full
// record {value, const, weight, censor scale} from obs4 (rare, out of the main path; its constant is not in csum).
__global__ __launch_bounds__(256) void pmx_ll_prepare_chunks(LLPrepareArgs a) {
  const int64_t ch = blockIdx.x;
  if (ch >= a.n_chunks) return;
  const int32_t nobs = a.chunk_nobs[ch], n_live = a.chunk_n[ch];
  const int64_t base = a.chunk_obs_off[ch];
  const int32_t total = nobs * 2 * a.G;
  for (int32_t i = threadIdx.x; i < total; i += 256) {
    const int32_t j = i % a.G, f = (i / a.G) % 2, k = i / (2 * a.G);
    double v = 0.0;
    if (j < n_live) {
      const double* rec = a.obs4 + (a.chunk_row[ch * a.G + j] + k) * 4;
      v = f == 0 ? rec[0] : rec[2];
      if (f == 1 && (rec[3] != 0.0 || rec[2] < 0.0) && v == v) v = -1.0;
    }
    a.cobs[base + 2 * a.G + i] = v;
  }
  if (threadIdx.x == 0) {
    // bit k: every live member's row of observation k is a plain one (weight neither 0 = missing nor the detour marker):
    // the kernel then folds the G members without a test per member
    uint64_t plain = 0;
    for (int32_t k = 0; k < nobs && k < 63; ++k) {
      bool all = true;
      for (int32_t j = 0; j < n_live; ++j) {
        const double* rec = a.obs4 + (a.chunk_row[ch * a.G + j] + k) * 4;
        const bool detour = (rec[3] != 0.0 || rec[2] < 0.0) && rec[2] == rec[2];
        all = all && !detour && __double_as_longlong(rec[2]) != 0;
      }
      if (all) plain |= (1ull << k);
    }
    a.cobs[base + a.G] = __longlong_as_double(static_cast<int64_t>(plain));
    // ... and the same mask indexed by program STEP (bit s < 63: step s carries an observation that is plain for every
    // live member), for the kernel that picks its straight-line steps by step number (pmx_analytical_classed_ll)
    uint64_t plain_step = 0;
    {
      const int32_t cl = a.chunk_cls[ch];
      int32_t k = 0;
      for (int64_t o = a.cls_prog_off[cl], s = 0; o < a.cls_prog_off[cl + 1]; ++o, ++s) {
        if ((a.prog_meta[o] >> 24) & 1u) {
          if (s < 63 && k < 63 && ((plain >> k) & 1ull)) plain_step |= 1ull << s;
          ++k;
        }
      }
    }
    a.cobs[base + a.G + 1] = __longlong_as_double(static_cast<int64_t>(plain_step));
    for (int32_t j = 2; j < a.G; ++j) a.cobs[base + a.G + j] = 0.0;
  }
  for (int32_t j = threadIdx.x; j < a.G; j += 256) {
    double csum = 0.0;
    if (j < n_live) {
      for (int32_t k = 0; k < nobs; ++k) {
        const double* rec = a.obs4 + (a.chunk_row[ch * a.G + j] + k) * 4;
        const bool detour = (rec[3] != 0.0 || rec[2] < 0.0) && rec[2] == rec[2];
        if (rec[2] != 0.0 && !detour) csum += rec[1];
      }
    }
    a.cobs[base + j] = csum;
  }
}
}  // namespace

hipError_t launch_ll_prepare(const LLPrepareArgs& a) {
  hipStream_t st = static_cast<hipStream_t>(a.stream);
  if (a.n_obs > 0) {
    hipLaunchKernelGGL(pmx_ll_prepare_obs, dim3(static_cast<uint32_t>((a.n_obs + 255) / 256)), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (a.n_chunks > 0) {
    hipLaunchKernelGGL(pmx_ll_prepare_chunks, dim3(static_cast<uint32_t>(a.n_chunks)), dim3(256), 0, st, a);
  }
  return hipGetLastError();
}

namespace {
// any non-zero status byte -> *flag = 1 (16 bytes per lane per trip; n is tens of MB at most)
__global__ __launch_bounds__(256) void pmx_status_any(const uint8_t* __restrict__ st, int64_t n, int32_t* __restrict__ flag) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256 * 16;
  uint32_t acc = 0;
  for (int64_t i = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 16; i < n; i += stride) {
    if (i + 16 <= n && (reinterpret_cast<uintptr_t>(st + i) & 15u) == 0) {
      const uint4 v = *reinterpret_cast<const uint4*>(st + i);
      acc |= v.x | v.y | v.z | v.w;
    } else {
      for (int64_t j = i; j < n && j < i + 16; ++j) acc |= st[j];
    }
  }
  if (__any(acc != 0u ? 1 : 0) && (threadIdx.x & 63u) == 0u) atomicOr(flag, 1);
}
}  // namespace

namespace {
// Streaming fills: what the device's write path takes when nothing else is asked of it (the measured ceiling bench.py
// prints beside the 8 TB/s datasheet peak: roofline.attainable).  Four shapes, the entry point reports the best:
//   0  grid-stride, 16 bytes per lane, streaming (nt) stores
//   1  the same with plain stores
//   2  the prediction kernels' own shape: one wave = 512 contiguous bytes per store (8 bytes per lane, nt), each
//      workgroup walking its own contiguous 64 KiB piece
//   3  shape 0 without the loop: one store per lane, as many workgroups as that takes
template <int SHAPE>
__global__ __launch_bounds__(256) void pmx_fill_linear(double* __restrict__ dst, int64_t n_pairs, double v) {
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  if constexpr (SHAPE == 2) {
    constexpr int64_t kPiece = 8192;  // doubles per workgroup piece
    const int64_t n = n_pairs * 2;
    for (int64_t base = static_cast<int64_t>(blockIdx.x) * kPiece; base < n; base += static_cast<int64_t>(gridDim.x) * kPiece) {
#pragma unroll 4
      for (int64_t i = threadIdx.x; i < kPiece; i += 256)
        if (base + i < n) __builtin_nontemporal_store(v, dst + base + i);
    }
  } else {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
    dbl2 vv;
    vv.x = v;
    vv.y = v;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n_pairs; i += stride) {
      if constexpr (SHAPE == 0)
        __builtin_nontemporal_store(vv, reinterpret_cast<dbl2*>(dst) + i);
      else
        reinterpret_cast<dbl2*>(dst)[i] = vv;
    }
  }
}
}  // namespace

hipError_t launch_fill_linear(double* d_dst, int64_t n_doubles, double v, void* stream, int shape) {
  const int64_t n_pairs = n_doubles / 2;
  if (n_pairs <= 0) return hipSuccess;
  int64_t blocks = shape == 2 ? (n_doubles + 8191) / 8192 : (n_pairs + 255) / 256;
  if (shape == 3) {  // one 16-byte streaming store per lane, no loop: the fastest of the shapes tried (tools/experiments/fill_probe.hip:
    shape = 0;       // 6.7 TB/s where the grid-stride forms reach 5.6-6.2 and hipMemsetAsync 6.4)
    if (blocks > 0x7fffffff) blocks = 0x7fffffff;
  } else if (blocks > 256 * 64) {
    blocks = 256 * 64;
  }
  const dim3 g(static_cast<uint32_t>(blocks)), b(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (shape == 0) hipLaunchKernelGGL(pmx_fill_linear<0>, g, b, 0, st, d_dst, n_pairs, v);
  if (shape == 1) hipLaunchKernelGGL(pmx_fill_linear<1>, g, b, 0, st, d_dst, n_pairs, v);
  if (shape == 2) hipLaunchKernelGGL(pmx_fill_linear<2>, g, b, 0, st, d_dst, n_pairs, v);
  return hipGetLastError();
}

hipError_t launch_status_any(const uint8_t* d_status, int64_t n, int32_t* d_flag, void* stream) {
  if (n <= 0) return hipSuccess;
  int64_t blocks = (n + 256 * 16 - 1) / (256 * 16);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pmx_status_any, dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), d_status, n, d_flag);
  return hipGetLastError();
}

#ifdef PMX_LL_STAMPS
}  // namespace pmx
extern "C" int32_t pmx_debug_ll_stamps(uint64_t* out5, int32_t reset) {  // diagnostic build only
  uint64_t z[5] = {0, 0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (out5 && hipMemcpyFromSymbol(out5, HIP_SYMBOL(pmx::g_ll_stamps), sizeof(z)) != hipSuccess) return 2;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(pmx::g_ll_stamps), z, sizeof(z)) != hipSuccess) return 3;
  return 0;
}
namespace pmx {
#endif

hipError_t launch_predict(const LaunchArgs& a, const char** name) {
  if (a.S <= 0 || (a.P <= 0 && !a.batch)) return hipSuccess;
  if (a.m.eq_kind == PMX_EQ_ANALYTICAL) {
    switch (a.m.kernel) {
      case 0: return launch_analytical_k<0>(a, name);
      case 1: return launch_analytical_k<1>(a, name);
      case 2: return launch_analytical_k<2>(a, name);
      case 3: return launch_analytical_k<3>(a, name);
      case 4: return launch_analytical_k<4>(a, name);
      case 5: return launch_analytical_k<5>(a, name);
      case 6: return launch_analytical_k<6>(a, name);
      case 7: return launch_analytical_k<7>(a, name);
      case 8: return launch_analytical_k<8>(a, name);
      case 9: return launch_analytical_k<9>(a, name);
      case 10: return launch_analytical_k<10>(a, name);
      case 11: return launch_analytical_k<11>(a, name);
      default: return hipErrorInvalidValue;
    }
  }
  switch (a.m.kernel) {
    case PMX_ODE_ONE_CMT_IV: return launch_ode<PMX_ODE_ONE_CMT_IV>(a, name);
    case PMX_ODE_ONE_CMT_ORAL: return launch_ode<PMX_ODE_ONE_CMT_ORAL>(a, name);
    case PMX_ODE_TWO_CMT_IV: return launch_ode<PMX_ODE_TWO_CMT_IV>(a, name);
    case PMX_ODE_TWO_CMT_ORAL: return launch_ode<PMX_ODE_TWO_CMT_ORAL>(a, name);
    case PMX_ODE_THREE_CMT_IV: return launch_ode<PMX_ODE_THREE_CMT_IV>(a, name);
    case PMX_ODE_THREE_CMT_ORAL: return launch_ode<PMX_ODE_THREE_CMT_ORAL>(a, name);
    case PMX_ODE_ONE_CMT_MM: return launch_ode<PMX_ODE_ONE_CMT_MM>(a, name);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace pmx
