// pmx_structures.hpp — closed-form compartment propagators as device functors.
//
// Arithmetic contracts (f64 throughout, src/simulator/mod.rs:14):
//   one_compartment                      one_compartment_models.rs:12-19
//   one_compartment_with_absorption      one_compartment_models.rs:32-44
//   two_compartments                     two_compartment_models.rs:14-48
//   two_compartments_with_absorption     two_compartment_models.rs:61-112
//   three_compartments                   three_compartment_models.rs:17-109
//   three_compartments_with_absorption   three_compartment_models.rs:126-240
//   CL re-parameterisations              {one,two,three}_compartment_cl_models.rs
//
// The reference recomputes eigenvalues and all coefficient quotients on EVERY
// sub-segment.  Here each structure is split into
//   prepare(kp)            -> Coef : everything that depends on the rate constants only
//                                    (eigenvalues, coefficient quotients as reciprocals)
//   make_prop(Coef, dt)    -> Prop : the exp() calls: the one-step propagator of the linear system,
//                                    x' = F x + J r  (F = transition matrix over dt, J = response to a
//                                    unit infusion rate)
//   apply(Prop, x, r)              : the state update, a handful of FMAs
// A lane whose rate constants are fixed (no covariate-derived parameter) prepares once; subjects that
// share a segment length dt (shared dosing/sampling design) can also share the lane's Prop — the
// "classed" kernel in pmx_kernels.hip applies one Prop to a register-resident batch of subjects.
// Divisions by (l1-l2), l_i, (ka-l_i) become multiplications by reciprocals
// computed in prepare(): results differ from the reference by a few ulp
// (parity budget 1e-6 relative, tests/test_gpu_parity.py).
#pragma once

#if !defined(__HIPCC_RTC__)  // (also embedded into the sources hiprtc compiles for user analytical models)
#include <hip/hip_runtime.h>

#include <cstdint>
#endif

namespace pmx {

// exp() for the propagators' arguments (-lambda dt: finite, almost always <= 0).  Same scheme as the library's
// (n = rint(x log2 e), r = x - n ln2 in two pieces, polynomial, scale by 2^n) without its special-case selects:
// a degree-11 near-minimax polynomial on |r| <= ln2/2 (Chebyshev fit of e^r in 60-digit arithmetic, tools/exp_poly_fit.py:
// approximation error 3.2e-18, i.e. the result is as good as the Horner evaluation in FMAs, within 1-2 ulp of the
// library's; the degree-13 Taylor series this replaced was no more accurate and two FMAs longer); v_ldexp saturates to
// 0 / inf by itself and NaN propagates.  ~17 VALU instructions instead of ~30.  (Keeping the constants resident in VGPRs
// instead of literals was tried for the generic kernel: one wave of occupancy less, 6 % slower.)
__device__ __forceinline__ double pmx_exp_poly(double x) {
  const double n = __builtin_rint(x * 1.4426950408889634074);
  double r = fma(n, -6.93147180369123816490e-01, x);  // ln2 high part: 21 trailing zero bits, n * hi is exact
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = 2.5110180394444085e-08;
  p = fma(p, r, 2.763282532538488e-07);
  p = fma(p, r, 2.7557240532217283e-06);
  p = fma(p, r, 2.480148497989361e-05);
  p = fma(p, r, 0.00019841269890408497);
  p = fma(p, r, 0.0013888888952784725);
  p = fma(p, r, 0.008333333333319464);
  p = fma(p, r, 0.04166666666648633);
  p = fma(p, r, 0.1666666666666668);
  p = fma(p, r, 0.5000000000000019);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, static_cast<int>(n));
}

// 2^t for the propagators: every exponential there is exp(-lambda dt), and lambda can carry the factor -log2(e) from the
// once-per-lane (or once-per-rebuild) set-up, so the argument arrives in base 2: n = rint(t), r = t - n EXACTLY (no
// two-piece ln2 reduction), degree-11 near-minimax polynomial of 2^r on |r| <= 1/2 (same fit: approximation error 3.2e-18,
// 1.7e-16 worst relative error with the double coefficients), scale by 2^n.  15 FP64-rate instructions; the rounding of
// t = lambda' dt is the same one ulp the product -lambda dt had.
__device__ __forceinline__ double pmx_exp2(double t) {
  const double n = __builtin_rint(t);
  const double r = t - n;
  double p = 4.4558179083360645e-10;
  p = fma(p, r, 7.074194297288521e-09);
  p = fma(p, r, 1.0178057087733941e-07);
  p = fma(p, r, 1.3215432535912375e-06);
  p = fma(p, r, 1.5252733841556773e-05);
  p = fma(p, r, 0.00015403530463724353);
  p = fma(p, r, 0.001333355814640647);
  p = fma(p, r, 0.009618129107587256);
  p = fma(p, r, 0.055504108664821625);
  p = fma(p, r, 0.24022650695910158);
  p = fma(p, r, 0.6931471805599453);
  p = fma(p, r, 1.0);
  return ldexp(p, static_cast<int>(n));
}
// N of them at once, the Horner chains interleaved coefficient by coefficient: one chain alone is eleven FMAs that each
// wait for the one before (the compiler emits N calls back to back as N such chains)
template <int N>
__device__ __forceinline__ void pmx_exp2_n(const double (&t)[N], double (&e)[N]) {
  constexpr double kC[11] = {7.074194297288521e-09, 1.0178057087733941e-07, 1.3215432535912375e-06, 1.5252733841556773e-05,
                             0.00015403530463724353, 0.001333355814640647,  0.009618129107587256,  0.055504108664821625,
                             0.24022650695910158,    0.6931471805599453,    1.0};
  double n[N], r[N], p[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    n[k] = __builtin_rint(t[k]);
    r[k] = t[k] - n[k];
    p[k] = 4.4558179083360645e-10;
  }
#pragma unroll
  for (int i = 0; i < 11; ++i)
#pragma unroll
    for (int k = 0; k < N; ++k) p[k] = fma(p[k], r[k], kC[i]);
#pragma unroll
  for (int k = 0; k < N; ++k) e[k] = ldexp(p[k], static_cast<int>(n[k]));
}
constexpr double kNegLog2e = -1.4426950408889634074;  // lambda' = -lambda log2(e):  exp(-lambda dt) = 2^(lambda' dt)

// (Tried: a 64-entry table of 2^(j/64) + a degree-5 polynomial, 11 FP64-rate instructions instead of 19.  The per-lane
// table fetch cost more than the eight FMAs it replaced: jittered C3 1.99 -> 2.43 ms, C5 and the generic walker unchanged.)
__device__ __forceinline__ double pmx_exp(double x) { return pmx_exp_poly(x); }

// 1/x to within an ulp or two: the hardware estimate polished by two Newton steps (what the IEEE division expands to,
// minus its scaling and fix-up instructions: ~8 issue slots instead of ~15).  For the per-segment coefficient rebuild
// of covariate models, which spends a third of its time dividing; not correctly rounded, 0 -> NaN instead of inf
// (either way the lane's predictions are non-finite and flagged).
__device__ __forceinline__ double pmx_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// FAST = the per-segment rebuild of a covariate model (make_prop_dyn); the once-per-lane set-up keeps the IEEE division
template <bool FAST>
__device__ __forceinline__ double rcp_of(double x) {
  if constexpr (FAST) return pmx_rcp(x);
  return 1.0 / x;
}

enum StructId : int { S_ONE = 0, S_ONE_ABS = 1, S_TWO = 2, S_TWO_ABS = 3, S_THREE = 4, S_THREE_ABS = 5 };

// kernel id (include/pmx.h PMX_K_*) -> structure / CL flag, usable on host and device
__host__ __device__ constexpr int kernel_structure(int k) {
  return (k == 0 || k == 1) ? S_ONE
         : (k == 2 || k == 3) ? S_ONE_ABS
         : (k == 4 || k == 5) ? S_TWO
         : (k == 6 || k == 7) ? S_TWO_ABS
         : (k == 8 || k == 9) ? S_THREE
                              : S_THREE_ABS;
}
__host__ __device__ constexpr bool kernel_is_cl(int k) { return k == 1 || k == 2 || k == 5 || k == 6 || k == 9 || k == 10; }
__host__ __device__ constexpr int kernel_nparams(int k) {
  return k == 0 ? 1 : k == 1 ? 2 : k == 2 ? 3 : k == 3 ? 2 : k == 4 ? 3 : k == 5 ? 4 : k == 6 ? 5 : k == 7 ? 4
         : k == 8 ? 5 : k == 9 ? 6 : k == 10 ? 7 : 6;
}

// CL -> micro-constant conversion, in the structure's native parameter order.
template <int KID>
__device__ __forceinline__ void to_native_params(const double* p, double* q) {
  if constexpr (KID == 1) {  // one_compartment_cl_models.rs:16-22
    q[0] = p[0] / p[1];
  } else if constexpr (KID == 2) {  // :38-45  [ka, ke]
    q[0] = p[0];
    q[1] = p[1] / p[2];
  } else if constexpr (KID == 5) {  // two_compartment_cl_models.rs:16-26  [ke,kcp,kpc]
    q[0] = p[0] / p[2];
    q[1] = p[1] / p[2];
    q[2] = p[1] / p[3];
  } else if constexpr (KID == 6) {  // :41-53  [ke,ka,kcp,kpc]
    q[0] = p[1] / p[3];
    q[1] = p[0];
    q[2] = p[2] / p[3];
    q[3] = p[2] / p[4];
  } else if constexpr (KID == 9) {  // three_compartment_cl_models.rs:16-31  [k10,k12,k13,k21,k31]
    q[0] = p[0] / p[3];
    q[1] = p[1] / p[3];
    q[2] = p[2] / p[3];
    q[3] = p[1] / p[4];
    q[4] = p[2] / p[5];
  } else if constexpr (KID == 10) {  // :46-67  [ka,k10,k12,k13,k21,k31]
    q[0] = p[0];
    q[1] = p[1] / p[4];
    q[2] = p[2] / p[4];
    q[3] = p[3] / p[4];
    q[4] = p[2] / p[5];
    q[5] = p[3] / p[6];
  } else {
    constexpr int n = kernel_nparams(KID);
#pragma unroll
    for (int i = 0; i < n; ++i) q[i] = p[i];
  }
}

template <int ST>
struct Structure;

// ---------------------------------------------------------------- one compartment
template <>
struct Structure<S_ONE> {
  static constexpr int NS = 1;
  struct Coef {
    double ke2, inv_ke;  // ke2 = -ke log2(e)
  };
  struct Prop {
    double e, j;
  };
  template <bool FAST = false>
  __device__ __forceinline__ static bool prepare(const double* kp, Coef& c) {
    c.ke2 = kp[0] * kNegLog2e;
    c.inv_ke = rcp_of<FAST>(kp[0]);
    return true;
  }
  static constexpr int NE = 1;
  __device__ __forceinline__ static void exps(const Coef& c, double dt, double (&e)[NE]) { e[0] = pmx_exp2(c.ke2 * dt); }
  // from_exps = from_exps_f (the transition part F) + from_exps_j (the response J to a unit infusion rate); a segment
  // without an active infusion (rate 0: wave-uniform in the GRID kernels) needs F only and advances with apply0
  __device__ __forceinline__ static void from_exps_f(const Coef&, const double (&e)[NE], Prop& p) { p.e = e[0]; }
  __device__ __forceinline__ static void from_exps_j(const Coef& c, const double (&e)[NE], Prop& p) { p.j = c.inv_ke * (1.0 - e[0]); }
  __device__ __forceinline__ static void from_exps(const Coef& c, const double (&e)[NE], Prop& p) {
    from_exps_f(c, e, p);
    from_exps_j(c, e, p);
  }
  __device__ __forceinline__ static void apply(const Prop& p, double (&x)[NS], double r) {
    x[0] = x[0] * p.e + p.j * r;
  }
  __device__ __forceinline__ static void apply0(const Prop& p, double (&x)[NS]) { x[0] = x[0] * p.e; }
  // x += J r: the response to an infusion rate on top of a state already advanced with apply0
  __device__ __forceinline__ static void add_j(const Prop& p, double (&x)[NS], double r) { x[0] += p.j * r; }
};

template <>
struct Structure<S_ONE_ABS> {
  static constexpr int NS = 2;
  struct Coef {
    double ka2, ke2, inv_ke, ka_over;  // ka_over = ka / (ka - ke); ka2, ke2 = -k log2(e)
  };
  struct Prop {
    double ea, ee, j, g;
  };
  template <bool FAST = false>
  __device__ __forceinline__ static bool prepare(const double* kp, Coef& c) {
    c.ka2 = kp[0] * kNegLog2e;
    c.ke2 = kp[1] * kNegLog2e;
    c.inv_ke = rcp_of<FAST>(kp[1]);
    c.ka_over = FAST ? kp[0] * pmx_rcp(kp[0] - kp[1]) : kp[0] / (kp[0] - kp[1]);
    return true;
  }
  static constexpr int NE = 2;
  __device__ __forceinline__ static void exps(const Coef& c, double dt, double (&e)[NE]) {
    const double t[2] = {c.ka2 * dt, c.ke2 * dt};
    pmx_exp2_n<2>(t, e);
  }
  __device__ __forceinline__ static void from_exps_f(const Coef& c, const double (&e)[NE], Prop& p) {
    p.ea = e[0];
    p.ee = e[1];
    p.g = c.ka_over * (p.ee - p.ea);
  }
  __device__ __forceinline__ static void from_exps_j(const Coef& c, const double (&e)[NE], Prop& p) { p.j = c.inv_ke * (1.0 - e[1]); }
  __device__ __forceinline__ static void from_exps(const Coef& c, const double (&e)[NE], Prop& p) {
    from_exps_f(c, e, p);
    from_exps_j(c, e, p);
  }
  __device__ __forceinline__ static void apply(const Prop& p, double (&x)[NS], double r) {
    const double g = x[0];
    x[0] = g * p.ea;
    x[1] = x[1] * p.ee + p.j * r + p.g * g;
  }
  __device__ __forceinline__ static void apply0(const Prop& p, double (&x)[NS]) {
    const double g = x[0];
    x[0] = g * p.ea;
    x[1] = x[1] * p.ee + p.g * g;
  }
  __device__ __forceinline__ static void add_j(const Prop& p, double (&x)[NS], double r) { x[1] += p.j * r; }
};

// ---------------------------------------------------------------- two compartments
struct TwoCore {
  double l1, l2, inv_d;       // inv_d = 1/(l1-l2)
  double l1s, l2s;            // -l log2(e): the exponents of make()'s 2^(ls dt)
  // every coefficient below already carries the 1/(l1-l2) of the reference's final division (two_compartment_models.rs:35-44)
  double a11, b11, kpc, kcp;  // M11 = a11 E1 + b11 E2 ; M12 = kpc (E2-E1) ; M21 = kcp (E2-E1)
  double a22, b22;            // M22 = a22 E1 + b22 E2
  double i0a, i0b, i1a, i1b;  // infusion vector: I0 = i0a(1-E1)+i0b(1-E2), I1 = i1a(1-E1)+i1b(1-E2)
  // `un` receives the UNscaled (l1-kpc, kpc-l2, kcp) for the absorption quotients of the 3-state structure
  template <bool FAST = false>
  __device__ __forceinline__ bool prepare(double ke, double kcp_, double kpc_, double (&un)[3]) {
    const double s = ke + kcp_ + kpc_;
    double disc = s * s - 4.0 * ke * kpc_;
    const bool ok = !(disc < 0.0);  // reference panics on disc < 0 (two_compartment_models.rs:20-22)
    disc = sqrt(disc);
    l1 = (s + disc) / 2.0;
    l2 = (s - disc) / 2.0;
    l1s = l1 * kNegLog2e;
    l2s = l2 * kNegLog2e;
    inv_d = rcp_of<FAST>(l1 - l2);
    const double ua11 = l1 - kpc_, ub11 = kpc_ - l2;
    un[0] = ua11;
    un[1] = ub11;
    un[2] = kcp_;
    a11 = ua11 * inv_d;
    b11 = ub11 * inv_d;
    kpc = kpc_ * inv_d;
    kcp = kcp_ * inv_d;
    a22 = (l1 - ke - kcp_) * inv_d;
    b22 = (ke + kcp_ - l2) * inv_d;
    double r1, r2;
    if constexpr (FAST) {
      r1 = pmx_rcp(l1);
      r2 = pmx_rcp(l2);
    } else {
      r1 = 1.0 / l1;
      r2 = 1.0 / l2;
    }
    i0a = a11 * r1;
    i0b = b11 * r2;
    i1a = -(kcp * r1);
    i1b = kcp * r2;
    return ok;
  }
};
struct TwoProp {
  double f00, f01, f10, f11, j0, j1;
  __device__ __forceinline__ void make_f(const TwoCore& t, double e1, double e2) {
    const double de = e2 - e1;
    f00 = t.a11 * e1 + t.b11 * e2;
    f01 = t.kpc * de;
    f10 = t.kcp * de;
    f11 = t.a22 * e1 + t.b22 * e2;
  }
  __device__ __forceinline__ void make_j(const TwoCore& t, double e1, double e2) {
    const double o1 = 1.0 - e1, o2 = 1.0 - e2;
    j0 = t.i0a * o1 + t.i0b * o2;
    j1 = t.i1a * o1 + t.i1b * o2;
  }
  __device__ __forceinline__ void make(const TwoCore& t, double e1, double e2) {
    make_f(t, e1, e2);
    make_j(t, e1, e2);
  }
};

template <>
struct Structure<S_TWO> {
  static constexpr int NS = 2;
  struct Coef {
    TwoCore t;
  };
  struct Prop {
    TwoProp p;
  };
  template <bool FAST = false>
  __device__ __forceinline__ static bool prepare(const double* kp, Coef& c) {
    double un[3];
    return c.t.template prepare<FAST>(kp[0], kp[1], kp[2], un);
  }
  static constexpr int NE = 2;
  __device__ __forceinline__ static void exps(const Coef& c, double dt, double (&e)[NE]) {
    const double t[2] = {c.t.l1s * dt, c.t.l2s * dt};
    pmx_exp2_n<2>(t, e);
  }
  __device__ __forceinline__ static void from_exps_f(const Coef& c, const double (&e)[NE], Prop& p) { p.p.make_f(c.t, e[0], e[1]); }
  __device__ __forceinline__ static void from_exps_j(const Coef& c, const double (&e)[NE], Prop& p) { p.p.make_j(c.t, e[0], e[1]); }
  __device__ __forceinline__ static void from_exps(const Coef& c, const double (&e)[NE], Prop& p) {
    p.p.make(c.t, e[0], e[1]);
  }
  __device__ __forceinline__ static void apply(const Prop& q, double (&x)[NS], double r) {
    const TwoProp& p = q.p;
    const double n0 = p.f00 * x[0] + p.f01 * x[1] + p.j0 * r;
    const double n1 = p.f10 * x[0] + p.f11 * x[1] + p.j1 * r;
    x[0] = n0;
    x[1] = n1;
  }
  __device__ __forceinline__ static void apply0(const Prop& q, double (&x)[NS]) {
    const TwoProp& p = q.p;
    const double n0 = p.f00 * x[0] + p.f01 * x[1];
    const double n1 = p.f10 * x[0] + p.f11 * x[1];
    x[0] = n0;
    x[1] = n1;
  }
  __device__ __forceinline__ static void add_j(const Prop& q, double (&x)[NS], double r) {
    x[0] += q.p.j0 * r;
    x[1] += q.p.j1 * r;
  }
};

template <>
struct Structure<S_TWO_ABS> {
  static constexpr int NS = 3;
  struct Coef {
    TwoCore t;
    double ka2, a0a, a0b, a1a, a1b;  // ka2 = -ka log2(e); absorption vector quotients (l1-kpc)/(ka-l1) ... times ka/(l1-l2)  (:97-103)
  };
  struct Prop {
    TwoProp p;
    double ea, g0, g1;
  };
  template <bool FAST = false>
  __device__ __forceinline__ static bool prepare(const double* kp, Coef& c) {
    // native order [ke, ka, kcp, kpc] (two_compartment_models.rs:62-65)
    double un[3];
    const bool ok = c.t.template prepare<FAST>(kp[0], kp[2], kp[3], un);
    const double ka = kp[1];
    c.ka2 = ka * kNegLog2e;
    const double h = ka * c.t.inv_d;  // ka * x[0] / (l1 - l2)  (:103)
    const double r1 = rcp_of<FAST>(ka - c.t.l1) * h, r2 = rcp_of<FAST>(ka - c.t.l2) * h;
    c.a0a = un[0] * r1;
    c.a0b = un[1] * r2;
    c.a1a = -(un[2] * r1);
    c.a1b = un[2] * r2;
    return ok;
  }
  static constexpr int NE = 3;
  __device__ __forceinline__ static void exps(const Coef& c, double dt, double (&e)[NE]) {
    const double t[3] = {c.t.l1s * dt, c.t.l2s * dt, c.ka2 * dt};
    pmx_exp2_n<3>(t, e);
  }
  __device__ __forceinline__ static void from_exps_f(const Coef& c, const double (&e)[NE], Prop& p) {
    const double e1 = e[0];
    const double e2 = e[1];
    p.ea = e[2];
    p.p.make_f(c.t, e1, e2);
    const double d1 = e1 - p.ea, d2 = e2 - p.ea;
    p.g0 = c.a0a * d1 + c.a0b * d2;
    p.g1 = c.a1a * d1 + c.a1b * d2;
  }
  __device__ __forceinline__ static void from_exps_j(const Coef& c, const double (&e)[NE], Prop& p) { p.p.make_j(c.t, e[0], e[1]); }
  __device__ __forceinline__ static void from_exps(const Coef& c, const double (&e)[NE], Prop& p) {
    from_exps_f(c, e, p);
    from_exps_j(c, e, p);
  }
  __device__ __forceinline__ static void apply(const Prop& q, double (&x)[NS], double r) {
    const TwoProp& p = q.p;
    const double g = x[0];
    const double n0 = p.f00 * x[1] + p.f01 * x[2] + p.j0 * r + q.g0 * g;
    const double n1 = p.f10 * x[1] + p.f11 * x[2] + p.j1 * r + q.g1 * g;
    x[0] = g * q.ea;
    x[1] = n0;
    x[2] = n1;
  }
  __device__ __forceinline__ static void apply0(const Prop& q, double (&x)[NS]) {
    const TwoProp& p = q.p;
    const double g = x[0];
    const double n0 = p.f00 * x[1] + p.f01 * x[2] + q.g0 * g;
    const double n1 = p.f10 * x[1] + p.f11 * x[2] + q.g1 * g;
    x[0] = g * q.ea;
    x[1] = n0;
    x[2] = n1;
  }
  __device__ __forceinline__ static void add_j(const Prop& q, double (&x)[NS], double r) {
    x[1] += q.p.j0 * r;
    x[2] += q.p.j1 * r;
  }
};

// ---------------------------------------------------------------- three compartments
struct ThreeCore {
  double l[3];
  double c[27];  // c[3*j + i]: row-major matrix entry j (0..8), eigenvalue i  == reference c_{3j+i+1}
  double d[9];   // d[3*row + i] = C_{row,1}^{(i)} / l_i        (infusion vector quotients)
  // the cubic's three real roots by the trigonometric method: three_compartment_models.rs:24-45
  __device__ __forceinline__ static bool eigen(double k10, double k12, double k13, double k21, double k31,
                                               double (&l)[3]) {
    const double a = k10 + k12 + k13 + k21 + k31;
    const double b = k10 * k21 + k13 * k21 + k10 * k31 + k12 * k31 + k21 * k31;
    const double cc = k10 * k21 * k31;
    const double m = (3.0 * b - a * a) * (1.0 / 3.0);
    const double n = (2.0 * (a * a * a) - 9.0 * a * b + 27.0 * cc) * (1.0 / 27.0);
    const double q = (n * n) * 0.25 + (m * m * m) * (1.0 / 27.0);
    const bool ok = !(q > 0.0);  // reference panics on q > 0 (:32-34)
    auto beta_of = [](double nn) { return -0.5 * nn; };
    (void)beta_of;
    // The reference takes gamma = |beta + i alpha|, theta = atan2(alpha, beta), gamma^(1/3) and cos/sin(theta/3)
    // (:36-45), i.e. the principal cube root z = cr (cs + i sn) of w = beta + i alpha.  Three f64 transcendentals
    // per segment dominate a covariate model's cost, so z is seeded in single precision and polished with two
    // Newton steps z <- (2 z + w / z^2) / 3 in f64 (quadratic: 2e-6 -> 4e-12 -> rounding).  The seed needs no
    // library call: gamma^2 = -m^3/27, so gamma^(1/3) = sqrt(-m/3); atan2 on alpha >= 0 is a degree-8 polynomial in
    // min/max (Abramowitz & Stegun 4.4.49, 2e-8) plus two reflections; theta/3 <= pi/3 goes straight to the hardware
    // sine/cosine.  Outside the float range the f64 functions are used directly.
    const double p3 = -m * (1.0 / 3.0);  // gamma^(2/3)
#ifndef PMX_EIGEN_ZDOMAIN
    // Real-root form (the default): with x = -lambda the reference's cubic is x^3 + a x^2 + b x + c, depressed by
    // x = t - a/3 to t^3 + m t + n = 0, whose roots are t_k = 2 sqrt(p3) cos((theta - 2 pi k) / 3) - the same angle theta.
    // The trigonometric seed is taken in single precision (alpha = sqrt(-q) included: no f64 square root at all) and two
    // of the roots are polished by Newton steps on the depressed cubic itself, t <- t - f(t) / f'(t) with the quotient's
    // reciprocal in single precision (the correction only needs its leading digits); the third follows from
    // t0 + t1 + t2 = 0.  Polished: the middle root (it can be near zero) and the largest (the SMALLEST eigenvalue, the one
    // whose relative accuracy the long segments feel); derived: the most negative one, the largest in magnitude.  Against an
    // 80-bit evaluation over the C5 parameter ranges two steps reach 1e-15 / 5e-14 / 3e-12 relative (largest / middle /
    // smallest eigenvalue; the last is the cancellation in a/3 - t every formula shares), a third changes nothing.
    if (p3 > 1.0e-20 && p3 < 1.0e20) {
      const float af = __builtin_amdgcn_sqrtf(static_cast<float>(-q)), bf = static_cast<float>(beta_of(n));
      const float ax = fabsf(bf);
      const float mx = fmaxf(ax, af), mn = fminf(ax, af);
      const float t = mn * __builtin_amdgcn_rcpf(mx);
      const float s2 = t * t;
      float pl = 0.0028662257f;
      pl = fmaf(pl, s2, -0.0161657367f);
      pl = fmaf(pl, s2, 0.0429096138f);
      pl = fmaf(pl, s2, -0.0752896400f);
      pl = fmaf(pl, s2, 0.1065626393f);
      pl = fmaf(pl, s2, -0.1420889944f);
      pl = fmaf(pl, s2, 0.1999355085f);
      pl = fmaf(pl, s2, -0.3333314528f);
      float thf = fmaf(pl * s2, t, t);
      thf = (af > ax) ? (1.57079632679f - thf) : thf;
      thf = (bf < 0.0f) ? (3.14159265359f - thf) : thf;
      const float ph = thf * (1.0f / 3.0f);
      const float crf = __builtin_amdgcn_sqrtf(static_cast<float>(p3));
      const float zrf = crf * __cosf(ph), zif = crf * __sinf(ph);
      double t2 = static_cast<double>(2.0f * zrf);                      // in [cr, 2 cr]
      double t1 = static_cast<double>(fmaf(1.7320508f, zif, -zrf));    // in [-cr, cr]
      // (one Newton step, then two chord steps on the same slope: the seed is good to ~1e-6, so f' moved by that much and
      // each chord step gains another ~5 digits - 80-bit check as above: 1e-15 / 5e-14 / 3e-12, where a single chord step
      // leaves 9e-11 on the smallest eigenvalue.  Two quarter-rate reciprocals less than a second Newton step.)
      double i1, i2;
      {
        const double g1 = fma(t1, t1, m), g2 = fma(t2, t2, m);
        const double f1 = fma(g1, t1, n), f2 = fma(g2, t2, n);
        const double d1 = fma(2.0 * t1, t1, g1), d2 = fma(2.0 * t2, t2, g2);  // 3 t^2 + m
        i1 = static_cast<double>(__builtin_amdgcn_rcpf(static_cast<float>(d1)));
        i2 = static_cast<double>(__builtin_amdgcn_rcpf(static_cast<float>(d2)));
        t1 = fma(-f1, i1, t1);
        t2 = fma(-f2, i2, t2);
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const double f1 = fma(fma(t1, t1, m), t1, n), f2 = fma(fma(t2, t2, m), t2, n);
        t1 = fma(-f1, i1, t1);
        t2 = fma(-f2, i2, t2);
      }
      const double a3 = a * (1.0 / 3.0);
      l[0] = a3 + (t1 + t2);  // a/3 - t0,  t0 = -(t1 + t2)
      l[1] = a3 - t1;
      l[2] = a3 - t2;
      return ok;
    }
#endif
    const double alpha = sqrt(-q);
    const double beta = -0.5 * n;
    double zr, zi;
#ifdef PMX_EIGEN_ZDOMAIN
    if (p3 > 1.0e-20 && p3 < 1.0e20) {
      const float af = static_cast<float>(alpha), bf = static_cast<float>(beta);
      const float ax = fabsf(bf);
      const float mx = fmaxf(ax, af), mn = fminf(ax, af);
      const float t = mn * __builtin_amdgcn_rcpf(mx);
      const float s2 = t * t;
      float pl = 0.0028662257f;
      pl = fmaf(pl, s2, -0.0161657367f);
      pl = fmaf(pl, s2, 0.0429096138f);
      pl = fmaf(pl, s2, -0.0752896400f);
      pl = fmaf(pl, s2, 0.1065626393f);
      pl = fmaf(pl, s2, -0.1420889944f);
      pl = fmaf(pl, s2, 0.1999355085f);
      pl = fmaf(pl, s2, -0.3333314528f);
      float thf = fmaf(pl * s2, t, t);
      thf = (af > ax) ? (1.57079632679f - thf) : thf;
      thf = (bf < 0.0f) ? (3.14159265359f - thf) : thf;
      const float ph = thf * (1.0f / 3.0f);
      const float crf = __builtin_amdgcn_sqrtf(static_cast<float>(p3));
      zr = static_cast<double>(crf * __cosf(ph));
      zi = static_cast<double>(crf * __sinf(ph));
// (two steps: the seed is good to ~1e-6 and each step squares the relative error - 1e-12, then rounding level; a third
// step bought nothing measurable, max rel err vs the oracle on C5 stays 1.5e-11, and cost 4 % of the kernel)
#ifndef PMX_EIGEN_NEWTON
#define PMX_EIGEN_NEWTON 2
#endif
#pragma unroll
      for (int it = 0; it < PMX_EIGEN_NEWTON; ++it) {
        const double z2r = zr * zr - zi * zi, z2i = 2.0 * zr * zi;
        const double inv = pmx_rcp(z2r * z2r + z2i * z2i);
        const double qr = (beta * z2r + alpha * z2i) * inv, qi = (alpha * z2r - beta * z2i) * inv;
        zr = (2.0 * zr + qr) * (1.0 / 3.0);
        zi = (2.0 * zi + qi) * (1.0 / 3.0);
      }
    } else
#endif
    {
      const double gamma = sqrt(beta * beta + alpha * alpha);
      const double theta = atan2(alpha, beta);
      const double cr = cbrt(gamma);  // reference: gamma.powf(1.0/3.0)
      double sn, cs;
      sincos(theta / 3.0, &sn, &cs);
      zr = cr * cs;
      zi = cr * sn;
    }
    const double rt3 = 1.7320508075688772;
    const double a3 = a * (1.0 / 3.0);
    l[0] = a3 + (zr + rt3 * zi);
    l[1] = a3 + (zr - rt3 * zi);
    l[2] = a3 - 2.0 * zr;
    return ok;
  }
  // eigen-solve + coefficients: three_compartment_models.rs:24-77
  __device__ __forceinline__ bool prepare(double k10, double k12, double k13, double k21, double k31) {
    const bool ok = eigen(k10, k12, k13, k21, k31, l);
    return prepare_tables(k10, k12, k13, k21, k31) && ok;
  }
  __device__ __forceinline__ bool prepare_tables(double k10, double k12, double k13, double k21, double k31) {
    const double K = k10 + k12 + k13;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double li = l[i];
      const double lo1 = l[(i + 1) % 3], lo2 = l[(i + 2) % 3];
      const double inv = 1.0 / ((lo1 - li) * (lo2 - li));  // d_i
      const double u = k21 - li, v = k31 - li, w = K - li;
      c[0 * 3 + i] = u * v * inv;                // c1..3
      c[1 * 3 + i] = k21 * v * inv;              // c4..6
      c[2 * 3 + i] = k31 * u * inv;              // c7..9
      c[3 * 3 + i] = k12 * v * inv;              // c10..12
      c[4 * 3 + i] = (w * v - k13 * k31) * inv;  // c13..15
      c[5 * 3 + i] = k12 * k31 * inv;            // c16..18
      c[6 * 3 + i] = k13 * u * inv;              // c19..21
      c[7 * 3 + i] = k21 * k13 * inv;            // c22..24
      c[8 * 3 + i] = (w * u - k12 * k21) * inv;  // c25..27
      const double il = 1.0 / li;
      d[0 * 3 + i] = c[0 * 3 + i] * il;
      d[1 * 3 + i] = c[3 * 3 + i] * il;
      d[2 * 3 + i] = c[6 * 3 + i] * il;
    }
    return true;
  }
};
struct ThreeProp {
  double m[9], j[3];
  __device__ __forceinline__ void make_f(const ThreeCore& t, const double (&e)[3]) {
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = t.c[3 * k] * e[0] + t.c[3 * k + 1] * e[1] + t.c[3 * k + 2] * e[2];
  }
  __device__ __forceinline__ void make_j(const ThreeCore& t, const double (&e)[3]) {
    const double o0 = 1.0 - e[0], o1 = 1.0 - e[1], o2 = 1.0 - e[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) j[k] = o0 * t.d[3 * k] + o1 * t.d[3 * k + 1] + o2 * t.d[3 * k + 2];
  }
  __device__ __forceinline__ void make(const ThreeCore& t, const double (&e)[3]) {
    make_f(t, e);
    make_j(t, e);
  }
};

// Fused prepare + make for lanes whose rate constants change with EVERY segment (covariate-derived parameters):
// one pass over the three eigenvalues that accumulates the transition matrix, the infusion response and (ABS) the
// absorption vector directly, never holding the 27 + 9 (+ 9) coefficient tables (the tabled form needs 255 VGPRs
// and runs one wave per SIMD).  Same terms as ThreeCore::prepare_tables + ThreeProp::make, summed in the same
// eigenvalue order.
// WITH_J = false: the segment has no active infusion, the response J (and the 1/l_i it needs) is left out.
template <bool ABS, bool WITH_J = true>
__device__ __forceinline__ bool three_direct(double k10, double k12, double k13, double k21, double k31, double ka,
                                             double dt, ThreeProp& p, double& ea, double (&g)[3]) {
  double l[3];
  const bool ok = ThreeCore::eigen(k10, k12, k13, k21, k31, l);
  const double K = k10 + k12 + k13;
  const double k1331 = k13 * k31, k1221 = k12 * k21, k1231 = k12 * k31, k2113 = k21 * k13;
  const double dts = dt * kNegLog2e;  // exp(-l dt) = 2^(l dts)
#ifdef PMX_THREE_SEQ_EXP
  if constexpr (ABS) ea = pmx_exp2(ka * dts);
#else
  // the segment's exponentials as one interleaved batch (pmx_exp2_n), ahead of the coefficient loop
  double ev[3];
  if constexpr (ABS) {
    const double t4[4] = {l[0] * dts, l[1] * dts, l[2] * dts, ka * dts};
    double e4[4];
    pmx_exp2_n<4>(t4, e4);
    ev[0] = e4[0];
    ev[1] = e4[1];
    ev[2] = e4[2];
    ea = e4[3];
  } else {
    const double t3[3] = {l[0] * dts, l[1] * dts, l[2] * dts};
    pmx_exp2_n<3>(t3, ev);
  }
#endif
#pragma unroll
  for (int k = 0; k < 9; ++k) p.m[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    p.j[k] = 0.0;
    g[k] = 0.0;
  }
#ifdef PMX_THREE_ROLLED
  double r0 = l[0], r1 = l[1], r2 = l[2];
#pragma unroll 1
  for (int i = 0; i < 3; ++i) {
    const double li = r0, lo1 = r1, lo2 = r2;
    r0 = lo1;
    r1 = lo2;
    r2 = li;
#else
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double li = l[i];
    const double lo1 = l[(i + 1) % 3], lo2 = l[(i + 2) % 3];
#endif
    // the three reciprocals of this eigenvalue - 1/d_i, 1/l_i and (ABS) 1/(ka - l_i) - from ONE Newton reciprocal of
    // their product (a third of this loop's instructions were reciprocals)
    const double di = (lo1 - li) * (lo2 - li);
    double inv, il = 0.0, ia = 0.0;
    if constexpr (ABS && WITH_J) {
      const double kl = ka - li;
      const double dl = di * li;
      const double r = pmx_rcp(dl * kl);
      inv = r * (li * kl);
      il = r * (di * kl);
      ia = r * dl;
    } else if constexpr (ABS) {
      const double kl = ka - li;
      const double r = pmx_rcp(di * kl);
      inv = r * kl;
      ia = r * di;
    } else if constexpr (WITH_J) {
      const double r = pmx_rcp(di * li);
      inv = r * li;
      il = r * di;
    } else {
      inv = pmx_rcp(di);
    }
    const double u = k21 - li, v = k31 - li, w = K - li;
    const double ui = u * inv, vi = v * inv;
#ifdef PMX_THREE_SEQ_EXP
    const double e = pmx_exp2(li * dts);
#else
    const double e = ev[i];
#endif
    const double c0 = u * vi, c3 = k12 * vi, c6 = k13 * ui;
    p.m[0] = fma(c0, e, p.m[0]);
    p.m[1] = fma(k21 * vi, e, p.m[1]);
    p.m[2] = fma(k31 * ui, e, p.m[2]);
    p.m[3] = fma(c3, e, p.m[3]);
    p.m[4] = fma(fma(w, v, -k1331) * inv, e, p.m[4]);
    p.m[5] = fma(k1231 * inv, e, p.m[5]);
    p.m[6] = fma(c6, e, p.m[6]);
    p.m[7] = fma(k2113 * inv, e, p.m[7]);
    p.m[8] = fma(fma(w, u, -k1221) * inv, e, p.m[8]);
    if constexpr (WITH_J) {
      const double o = (1.0 - e) * il;
      p.j[0] = fma(c0, o, p.j[0]);
      p.j[1] = fma(c3, o, p.j[1]);
      p.j[2] = fma(c6, o, p.j[2]);
    }
    if constexpr (ABS) {
      const double q = (e - ea) * ia;
      g[0] = fma(c0, q, g[0]);
      g[1] = fma(c3, q, g[1]);
      g[2] = fma(c6, q, g[2]);
    }
  }
  if constexpr (ABS) {
#pragma unroll
    for (int k = 0; k < 3; ++k) g[k] *= ka;
  }
  return ok;
}

// ---------------------------------------------------------------- rate-free segment of a covariate model, matrix-free
// A covariate model rebuilds its propagator for (almost) every segment and applies it ONCE, so the nine matrix entries
// (and the absorption vector) of three_direct are never needed as such.  With x' = -B x,
//     B = [ ka    0     0     0  ]      (ABS: gut first; without absorption the lower-right 3 x 3 block)
//         [-ka    K   -k21  -k31 ]      K = k10 + k12 + k13
//         [ 0   -k12   k21    0  ]
//         [ 0   -k13    0    k31 ]
// exp(-B dt) x = p(B) x for the polynomial p that interpolates f(l) = exp(-l dt) on B's eigenvalues {l0, l1, l2 (, ka)}
// (Lagrange-Sylvester; the same spectral sum as three_compartment_models.rs:47-77 / :218-236, regrouped).  In Newton form
//     p(B) x = c0 x + c1 (B - l0) x + c2 (B - l1)(B - l0) x + c3 (B - l2)(B - l1)(B - l0) x,
// c_k = f[l0..lk] the divided differences: three sparse matrix-vector products (13 instructions each) and six
// differences whose reciprocals come from ONE Newton reciprocal of their product - about 100 vector instructions where
// the table-free matrix form takes about 165, and 7 numbers to keep for a segment that repeats (l0..l2, c0..c3).
// Conditioning is that of the partial-fraction form: both divide by the same eigenvalue gaps.
template <bool ABS>
struct ThreeNewton {
  static constexpr int NC = ABS ? 4 : 3;
  static constexpr int NKEEP = 3 + NC + (ABS ? 1 : 0);
  // keep = {l0, l1, l2, c0.. (, exp(-ka dt): the gut decouples, its own step is that one product)}
  // REUSE: the eigenvalues of the previous build are still valid (same rate constants, another step length): `lprev`
  // (descending, as ThreeCore::eigen returns them) and `okprev` are read instead of solved for, and written otherwise
  template <bool REUSE = false>
  __device__ __forceinline__ static bool make(double k10, double k12, double k13, double k21, double k31, double ka, double dt,
                                              double (&keep)[NKEEP], double (&lprev)[3], bool& okprev) {
    double le[3];
    bool ok;
    if constexpr (REUSE) {
      le[0] = lprev[0];
      le[1] = lprev[1];
      le[2] = lprev[2];
      ok = okprev;
    } else {
      ok = ThreeCore::eigen(k10, k12, k13, k21, k31, le);
      lprev[0] = le[0];
      lprev[1] = le[1];
      lprev[2] = le[2];
      okprev = ok;
    }
    // nodes in ASCENDING order (eigen returns l0 >= l1 >= l2): the slowest mode first.  In f64 against an 80-bit evaluation
    // of the spectral sum, 3000 draws of the C5 parameter ranges: this order 1.8e-13 worst relative error (the partial-
    // fraction form 2.4e-13), descending order 4.3e-12.
    const double l[3] = {le[2], le[1], le[0]};
    const double dts = dt * kNegLog2e;  // exp(-l dt) = 2^(l dts)
    const double d01 = l[1] - l[0], d12 = l[2] - l[1], d02 = l[2] - l[0];
    keep[0] = l[0];
    keep[1] = l[1];
    keep[2] = l[2];
    if constexpr (ABS) {
      const double t4[4] = {l[0] * dts, l[1] * dts, l[2] * dts, ka * dts};
      double e[4];
      pmx_exp2_n<4>(t4, e);
      const double d23 = ka - l[2], d13 = ka - l[1], d03 = ka - l[0];
      // six reciprocals from one: pair products, the reciprocal of all six, then peel
      const double p1 = d01 * d12, p2 = d23 * d02, p3 = d13 * d03;
      const double p23 = p2 * p3, p13 = p1 * p3, p12 = p1 * p2;
      const double R = pmx_rcp(p1 * p23);
      const double i1 = R * p23, i2 = R * p13, i3 = R * p12;  // 1/p1, 1/p2, 1/p3
      const double r01 = i1 * d12, r12 = i1 * d01, r23 = i2 * d02, r02 = i2 * d23, r13 = i3 * d03, r03 = i3 * d13;
      const double f01 = (e[1] - e[0]) * r01, f12 = (e[2] - e[1]) * r12, f23 = (e[3] - e[2]) * r23;
      const double f012 = (f12 - f01) * r02, f123 = (f23 - f12) * r13;
      keep[3] = e[0];
      keep[4] = f01;
      keep[5] = f012;
      keep[6] = (f123 - f012) * r03;
      keep[7] = e[3];
    } else {
      const double t3[3] = {l[0] * dts, l[1] * dts, l[2] * dts};
      double e[3];
      pmx_exp2_n<3>(t3, e);
      const double q12 = d12 * d02, q02 = d01 * d02, q01 = d01 * d12;
      const double R = pmx_rcp(d01 * q12);
      const double r01 = R * q12, r12 = R * q02, r02 = R * q01;
      const double f01 = (e[1] - e[0]) * r01, f12 = (e[2] - e[1]) * r12;
      keep[3] = e[0];
      keep[4] = f01;
      keep[5] = (f12 - f01) * r02;
    }
    return ok;
  }
  // w <- (B - n) w on the three-compartment block (c, p2, p3) + (ABS) the gut in front
  __device__ __forceinline__ static void shift_mul(double k12, double k13, double k21, double k31, double K, double ka, double n,
                                                   double& g, double& c, double& p2, double& p3) {
    double nc = fma(K - n, c, -(k21 * p2));
    nc = fma(-k31, p3, nc);
    if constexpr (ABS) nc = fma(-ka, g, nc);
    const double n2 = fma(k21 - n, p2, -(k12 * c));
    const double n3 = fma(k31 - n, p3, -(k13 * c));
    if constexpr (ABS) g = (ka - n) * g;  // (feeds nc of the NEXT product: the gut's own result is taken from exp(-ka dt))
    c = nc;
    p2 = n2;
    p3 = n3;
  }
  // x <- p(B) x
  __device__ __forceinline__ static void apply(double k10, double k12, double k13, double k21, double k31, double ka,
                                               const double (&keep)[NKEEP], double& g, double& c, double& p2, double& p3) {
    const double K = k10 + k12 + k13;
    double wg = g, wc = c, w2 = p2, w3 = p3;
    double yg = 0.0, yc = keep[3] * wc, y2 = keep[3] * w2, y3 = keep[3] * w3;
    if constexpr (ABS) yg = keep[NKEEP - 1] * wg;  // p(B) x restricted to the gut = p(ka) g = exp(-ka dt) g
#pragma unroll
    for (int k = 1; k < NC; ++k) {
      shift_mul(k12, k13, k21, k31, K, ka, keep[k - 1], wg, wc, w2, w3);
      const double ck = keep[3 + k];
      yc = fma(ck, wc, yc);
      y2 = fma(ck, w2, y2);
      y3 = fma(ck, w3, y3);
    }
    g = yg;
    c = yc;
    p2 = y2;
    p3 = y3;
  }
};

template <>
struct Structure<S_THREE> {
  static constexpr int NS = 3;
  struct Coef {
    ThreeCore t;
  };
  struct Prop {
    ThreeProp p;
  };
  __device__ __forceinline__ static bool prepare(const double* kp, Coef& c) {
    return c.t.prepare(kp[0], kp[1], kp[2], kp[3], kp[4]);
  }
  static constexpr int NE = 3;
  __device__ __forceinline__ static void exps(const Coef& c, double dt, double (&e)[NE]) {
    const double dts = dt * kNegLog2e;
    const double t[3] = {c.t.l[0] * dts, c.t.l[1] * dts, c.t.l[2] * dts};
    pmx_exp2_n<3>(t, e);
  }
  __device__ __forceinline__ static void from_exps_f(const Coef& c, const double (&e)[NE], Prop& p) { p.p.make_f(c.t, e); }
  __device__ __forceinline__ static void from_exps_j(const Coef& c, const double (&e)[NE], Prop& p) { p.p.make_j(c.t, e); }
  __device__ __forceinline__ static void from_exps(const Coef& c, const double (&e)[NE], Prop& p) { p.p.make(c.t, e); }
  template <bool WITH_J = true>
  __device__ __forceinline__ static bool make_prop_dyn(const double* kp, double dt, Prop& p) {
    double ea_unused, g_unused[3];
    return three_direct<false, WITH_J>(kp[0], kp[1], kp[2], kp[3], kp[4], 0.0, dt, p.p, ea_unused, g_unused);
  }
  // rate-free segment, matrix-free (ThreeNewton): what to keep for a repeat, and the step itself
  static constexpr int ND0 = ThreeNewton<false>::NKEEP;
  template <bool REUSE = false>
  __device__ __forceinline__ static bool direct0_make(const double* kp, double dt, double (&keep)[ND0], double (&lprev)[3], bool& okprev) {
    return ThreeNewton<false>::template make<REUSE>(kp[0], kp[1], kp[2], kp[3], kp[4], 0.0, dt, keep, lprev, okprev);
  }
  __device__ __forceinline__ static void direct0_apply(const double* kp, const double (&keep)[ND0], double (&x)[NS]) {
    double g_unused = 0.0;
    ThreeNewton<false>::apply(kp[0], kp[1], kp[2], kp[3], kp[4], 0.0, keep, g_unused, x[0], x[1], x[2]);
  }
  __device__ __forceinline__ static void apply(const Prop& q, double (&x)[NS], double r) {
    const ThreeProp& p = q.p;
    const double y0 = p.m[0] * x[0] + p.m[1] * x[1] + p.m[2] * x[2] + p.j[0] * r;
    const double y1 = p.m[3] * x[0] + p.m[4] * x[1] + p.m[5] * x[2] + p.j[1] * r;
    const double y2 = p.m[6] * x[0] + p.m[7] * x[1] + p.m[8] * x[2] + p.j[2] * r;
    x[0] = y0;
    x[1] = y1;
    x[2] = y2;
  }
  __device__ __forceinline__ static void apply0(const Prop& q, double (&x)[NS]) {
    const ThreeProp& p = q.p;
    const double y0 = p.m[0] * x[0] + p.m[1] * x[1] + p.m[2] * x[2];
    const double y1 = p.m[3] * x[0] + p.m[4] * x[1] + p.m[5] * x[2];
    const double y2 = p.m[6] * x[0] + p.m[7] * x[1] + p.m[8] * x[2];
    x[0] = y0;
    x[1] = y1;
    x[2] = y2;
  }
  __device__ __forceinline__ static void add_j(const Prop& q, double (&x)[NS], double r) {
#pragma unroll
    for (int i = 0; i < 3; ++i) x[i] += q.p.j[i] * r;
  }
};

template <>
struct Structure<S_THREE_ABS> {
  static constexpr int NS = 4;
  struct Coef {
    ThreeCore t;
    double ka;
    double f[9];  // f[3*row + i] = C_{row,1}^{(i)} / (ka - l_i)   (:218-228)
  };
  struct Prop {
    ThreeProp p;
    double ea, g[3];
  };
  __device__ __forceinline__ static bool prepare(const double* kp, Coef& c) {
    const bool ok = c.t.prepare(kp[1], kp[2], kp[3], kp[4], kp[5]);
    c.ka = kp[0];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double inv = 1.0 / (c.ka - c.t.l[i]);
      c.f[0 * 3 + i] = c.t.c[0 * 3 + i] * inv;
      c.f[1 * 3 + i] = c.t.c[3 * 3 + i] * inv;
      c.f[2 * 3 + i] = c.t.c[6 * 3 + i] * inv;
    }
    return ok;
  }
  static constexpr int NE = 4;
  __device__ __forceinline__ static void exps(const Coef& c, double dt, double (&e4)[NE]) {
    const double dts = dt * kNegLog2e;
    const double t[4] = {c.t.l[0] * dts, c.t.l[1] * dts, c.t.l[2] * dts, c.ka * dts};
    pmx_exp2_n<4>(t, e4);
  }
  __device__ __forceinline__ static void from_exps_f(const Coef& c, const double (&e4)[NE], Prop& p) {
    const double e[3] = {e4[0], e4[1], e4[2]};
    p.ea = e4[3];
    p.p.make_f(c.t, e);
    const double d0 = e[0] - p.ea, d1 = e[1] - p.ea, d2 = e[2] - p.ea;
#pragma unroll
    for (int k = 0; k < 3; ++k) p.g[k] = (d0 * c.f[3 * k] + d1 * c.f[3 * k + 1] + d2 * c.f[3 * k + 2]) * c.ka;  // (:230)
  }
  __device__ __forceinline__ static void from_exps_j(const Coef& c, const double (&e4)[NE], Prop& p) {
    const double e[3] = {e4[0], e4[1], e4[2]};
    p.p.make_j(c.t, e);
  }
  __device__ __forceinline__ static void from_exps(const Coef& c, const double (&e4)[NE], Prop& p) {
    from_exps_f(c, e4, p);
    from_exps_j(c, e4, p);
  }
  template <bool WITH_J = true>
  __device__ __forceinline__ static bool make_prop_dyn(const double* kp, double dt, Prop& p) {
    return three_direct<true, WITH_J>(kp[1], kp[2], kp[3], kp[4], kp[5], kp[0], dt, p.p, p.ea, p.g);
  }
  static constexpr int ND0 = ThreeNewton<true>::NKEEP;
  template <bool REUSE = false>
  __device__ __forceinline__ static bool direct0_make(const double* kp, double dt, double (&keep)[ND0], double (&lprev)[3], bool& okprev) {
    return ThreeNewton<true>::template make<REUSE>(kp[1], kp[2], kp[3], kp[4], kp[5], kp[0], dt, keep, lprev, okprev);
  }
  __device__ __forceinline__ static void direct0_apply(const double* kp, const double (&keep)[ND0], double (&x)[NS]) {
    ThreeNewton<true>::apply(kp[1], kp[2], kp[3], kp[4], kp[5], kp[0], keep, x[0], x[1], x[2], x[3]);
  }
  __device__ __forceinline__ static void apply(const Prop& q, double (&x)[NS], double r) {
    const ThreeProp& p = q.p;
    const double g = x[0];
    const double y0 = p.m[0] * x[1] + p.m[1] * x[2] + p.m[2] * x[3] + p.j[0] * r + q.g[0] * g;
    const double y1 = p.m[3] * x[1] + p.m[4] * x[2] + p.m[5] * x[3] + p.j[1] * r + q.g[1] * g;
    const double y2 = p.m[6] * x[1] + p.m[7] * x[2] + p.m[8] * x[3] + p.j[2] * r + q.g[2] * g;
    x[0] = g * q.ea;
    x[1] = y0;
    x[2] = y1;
    x[3] = y2;
  }
  __device__ __forceinline__ static void apply0(const Prop& q, double (&x)[NS]) {
    const ThreeProp& p = q.p;
    const double g = x[0];
    const double y0 = p.m[0] * x[1] + p.m[1] * x[2] + p.m[2] * x[3] + q.g[0] * g;
    const double y1 = p.m[3] * x[1] + p.m[4] * x[2] + p.m[5] * x[3] + q.g[1] * g;
    const double y2 = p.m[6] * x[1] + p.m[7] * x[2] + p.m[8] * x[3] + q.g[2] * g;
    x[0] = g * q.ea;
    x[1] = y0;
    x[2] = y1;
    x[3] = y2;
  }
  __device__ __forceinline__ static void add_j(const Prop& q, double (&x)[NS], double r) {
#pragma unroll
    for (int i = 0; i < 3; ++i) x[1 + i] += q.p.j[i] * r;
  }
};

// make_prop = the pmx_exp() calls + the coefficient combination
template <int ST>
__device__ __forceinline__ void make_prop(const typename Structure<ST>::Coef& c, double dt, typename Structure<ST>::Prop& p) {
  double e[Structure<ST>::NE];
  Structure<ST>::exps(c, dt, e);
  Structure<ST>::from_exps(c, e, p);
}

// The exponential ladder: when a step's length is n x the previous step's (n = 2, 3, 4; sampling designs on
// 0.5/1/2/4/8/12/24 h grids are exactly that), pmx_exp(-lambda n dt) = pmx_exp(-lambda dt)^n costs n-1 multiplies instead of
// an pmx_exp() call.  Each rung multiplies the relative error of the previous one by n; the host caps the
// cumulative factor (pmx_compile.cpp ladder_codes), which keeps the deviation from a fresh pmx_exp() below 1e-12.
template <int NE>
__device__ __forceinline__ void ladder_pow(double (&e)[NE], uint32_t n) {
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const double b = e[i];
    const double sq = b * b;
    e[i] = (n == 2u) ? sq : ((n == 3u) ? sq * b : sq * sq);
  }
}

// structures with the matrix-free rate-free step (direct0_make / direct0_apply)
template <int ST>
inline constexpr bool kHasDirect0 =
#ifdef PMX_NO_DIRECT0
    false;
#else
    (ST == S_THREE || ST == S_THREE_ABS);
#endif

// covariate-derived rate constants: prepare + make for ONE segment.  Structures with a fused form provide
// make_prop_dyn; the others go through their Coef.
template <int ST, bool WITH_J = true>
__device__ __forceinline__ bool make_prop_dyn(const double* kp, double dt, typename Structure<ST>::Prop& p) {
  if constexpr (ST == S_THREE || ST == S_THREE_ABS) {
    return Structure<ST>::template make_prop_dyn<WITH_J>(kp, dt, p);
  } else {
    typename Structure<ST>::Coef c;
    const bool ok = Structure<ST>::template prepare<true>(kp, c);
    double e[Structure<ST>::NE];
    Structure<ST>::exps(c, dt, e);
    Structure<ST>::from_exps_f(c, e, p);
    if constexpr (WITH_J) Structure<ST>::from_exps_j(c, e, p);
    return ok;
  }
}

// x' = F x + J r from the segment's exponentials, for a WAVE-UNIFORM rate r: a segment without an active infusion skips J
template <int ST>
__device__ __forceinline__ void step_from_exps(const typename Structure<ST>::Coef& c, const double (&e)[Structure<ST>::NE],
                                               double (&x)[Structure<ST>::NS], double r) {
  typename Structure<ST>::Prop p;
  Structure<ST>::from_exps_f(c, e, p);
  if (r != 0.0) {
    Structure<ST>::from_exps_j(c, e, p);
    Structure<ST>::apply(p, x, r);
  } else {
    Structure<ST>::apply0(p, x);
  }
}

// advance = make_prop + apply (one sub-segment of Analytical::solve, analytical/mod.rs:363-364)
template <int ST>
__device__ __forceinline__ void advance(const typename Structure<ST>::Coef& c, double (&x)[Structure<ST>::NS], double dt,
                                        double r) {
  typename Structure<ST>::Prop p;
  make_prop<ST>(c, dt, p);
  Structure<ST>::apply(p, x, r);
}

}  // namespace pmx
