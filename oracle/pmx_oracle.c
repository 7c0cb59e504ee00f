/*
 * pmx_oracle.c — CPU ORACLE (test infrastructure; see pmx_oracle.h for the rules).
 *
 * Plain-C restatement of pharmsol v0.28.8's prediction path, one (subject,
 * support point) at a time, in the reference's operation order.  Build with
 * -ffp-contract=off: Rust never contracts a*b+c into an FMA, and the libm calls
 * (exp, pow, sqrt, sin, cos, atan2) are the same glibc functions Rust's f64
 * methods call on Linux.
 *
 * Citations are file:line under /root/reference/.
 */
#include "pmx_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static __thread char g_err[512];
const char* pmx_oracle_last_error(void) { return g_err; }
#define FAIL(code, ...)                       \
  do {                                        \
    snprintf(g_err, sizeof g_err, __VA_ARGS__); \
    return (code);                            \
  } while (0)

int32_t pmx_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* f64::total_cmp key (event.rs:301 uses time.total_cmp). */
static int64_t total_key(double v) {
  int64_t b;
  memcpy(&b, &v, 8);
  b ^= (int64_t)(((uint64_t)(b >> 63)) >> 1);
  return b;
}

/* ------------------------------------------------------------------------- */
/* Covariates — src/data/covariate.rs                                         */
/* ------------------------------------------------------------------------- */
typedef struct {
  double from, to; /* to = +inf encodes `None` */
  int has_to;
  int linear;
  double slope, intercept, value;
} cov_seg;

typedef struct {
  int64_t n; /* knots == segments */
  cov_seg* seg;
  double first_t, first_v, last_t, last_v;
} cov_track;

/* Covariate::build_segments, covariate.rs:189-214 */
static int cov_build(const double* kt, const double* kv, int64_t n, int fixed, cov_track* out) {
  out->n = n;
  out->seg = NULL;
  if (n == 0) return 0;
  /* observations.sort_by(total_cmp) — stable insertion sort on (time) */
  double* t = (double*)malloc(sizeof(double) * (size_t)n);
  double* v = (double*)malloc(sizeof(double) * (size_t)n);
  for (int64_t i = 0; i < n; i++) {
    double ti = kt[i], vi = kv[i];
    int64_t j = i;
    while (j > 0 && total_key(t[j - 1]) > total_key(ti)) {
      t[j] = t[j - 1];
      v[j] = v[j - 1];
      j--;
    }
    t[j] = ti;
    v[j] = vi;
  }
  out->seg = (cov_seg*)malloc(sizeof(cov_seg) * (size_t)n);
  for (int64_t i = 0; i < n; i++) {
    cov_seg* s = &out->seg[i];
    s->from = t[i];
    s->has_to = (i + 1 < n);
    s->to = s->has_to ? t[i + 1] : INFINITY;
    if (fixed || !s->has_to) {
      s->linear = 0;
      s->value = v[i]; /* CarryForward */
      s->slope = s->intercept = 0.0;
    } else {
      s->linear = 1;
      s->slope = (v[i + 1] - v[i]) / (t[i + 1] - t[i]);
      s->intercept = v[i] - s->slope * t[i];
      s->value = 0.0;
    }
  }
  out->first_t = t[0];
  out->first_v = v[0];
  out->last_t = t[n - 1];
  out->last_v = v[n - 1];
  free(t);
  free(v);
  return 0;
}

/* Covariate::interpolate, covariate.rs:216-241 (+ CovariateSegment::interpolate :50-65) */
static int cov_interp(const cov_track* c, double time, double* value) {
  if (c->n == 0) return -1; /* CovariateError::MissingSegments */
  for (int64_t i = 0; i < c->n; i++) {
    const cov_seg* s = &c->seg[i];
    if (s->from <= time && (!s->has_to || time < s->to)) {
      *value = s->linear ? (s->slope * time + s->intercept) : s->value;
      return 0;
    }
  }
  if (time < c->first_t) {
    *value = c->first_v;
    return 0;
  }
  if (time >= c->last_t) {
    *value = c->last_v;
    return 0;
  }
  return -1;
}

int32_t pmx_oracle_cov_interpolate(const double* kt, const double* kv, int64_t n, int32_t fixed, double t,
                                   double* value) {
  cov_track c;
  cov_build(kt, kv, n, fixed, &c);
  int r = cov_interp(&c, t, value);
  free(c.seg);
  return r == 0 ? PMX_OK : PMX_ERR_INVALID_ARGUMENT;
}

/* ------------------------------------------------------------------------- */
/* Closed-form kernels — src/simulator/equation/analytical/{one,two,three}_compartment_models.rs       */
/* ------------------------------------------------------------------------- */

/* one_compartment, one_compartment_models.rs:12-19 */
static int k_one_compartment(const double* x, const double* p, double t, const double* rateiv, double* xo) {
  double ke = p[0];
  xo[0] = x[0] * exp(-ke * t) + rateiv[0] / ke * (1.0 - exp(-ke * t));
  return 0;
}

/* one_compartment_with_absorption, one_compartment_models.rs:32-44 */
static int k_one_compartment_abs(const double* x, const double* p, double t, const double* rateiv, double* xo) {
  double ka = p[0];
  double ke = p[1];
  double x0 = x[0], x1 = x[1];
  xo[0] = x0 * exp(-ka * t);
  xo[1] = x1 * exp(-ke * t) + rateiv[0] / ke * (1.0 - exp(-ke * t)) +
          ((ka * x0) / (ka - ke)) * (exp(-ke * t) - exp(-ka * t));
  return 0;
}

/* two_compartments, two_compartment_models.rs:14-48 */
static int k_two_compartments(const double* x, const double* p, double t, const double* rateiv, double* xo) {
  double ke = p[0], kcp = p[1], kpc = p[2];
  double s = (ke + kcp + kpc);
  double sq = s * s - 4.0 * ke * kpc; /* .powi(2) */
  if (sq < 0.0) return PMX_PAIR_COMPLEX_ROOTS; /* panic!("Imaginary solutions") :20-22 */
  sq = sqrt(sq);
  double l1 = (ke + kcp + kpc + sq) / 2.0;
  double l2 = (ke + kcp + kpc - sq) / 2.0;
  double e1 = exp(-l1 * t);
  double e2 = exp(-l2 * t);
  /* Matrix2::new is row-major in its arguments :28-33 */
  double m11 = (l1 - kpc) * e1 + (kpc - l2) * e2;
  double m12 = -kpc * e1 + kpc * e2;
  double m21 = -kcp * e1 + kcp * e2;
  double m22 = (l1 - ke - kcp) * e1 + (ke + kcp - l2) * e2;
  /* (M * x) / (l1 - l2) :35 — nalgebra gemv accumulates column by column */
  double nz0 = (m11 * x[0] + m12 * x[1]) / (l1 - l2);
  double nz1 = (m21 * x[0] + m22 * x[1]) / (l1 - l2);
  double iv0 = ((l1 - kpc) / l1) * (1.0 - e1) + ((kpc - l2) / l2) * (1.0 - e2);
  double iv1 = (-kcp / l1) * (1.0 - e1) + (kcp / l2) * (1.0 - e2);
  double f = rateiv[0] / (l1 - l2); /* :42 */
  xo[0] = nz0 + iv0 * f;
  xo[1] = nz1 + iv1 * f;
  return 0;
}

/* two_compartments_with_absorption, two_compartment_models.rs:61-112 */
static int k_two_compartments_abs(const double* x, const double* p, double t, const double* rateiv, double* xo) {
  double ke = p[0], ka = p[1], kcp = p[2], kpc = p[3];
  double x0 = x[0], x1 = x[1], x2 = x[2];
  double s = (ke + kcp + kpc);
  double sq = s * s - 4.0 * ke * kpc;
  if (sq < 0.0) return PMX_PAIR_COMPLEX_ROOTS;
  sq = sqrt(sq);
  double l1 = (ke + kcp + kpc + sq) / 2.0;
  double l2 = (ke + kcp + kpc - sq) / 2.0;
  double e1 = exp(-l1 * t);
  double e2 = exp(-l2 * t);
  double m11 = (l1 - kpc) * e1 + (kpc - l2) * e2;
  double m12 = -kpc * e1 + kpc * e2;
  double m21 = -kcp * e1 + kcp * e2;
  double m22 = (l1 - ke - kcp) * e1 + (ke + kcp - l2) * e2;
  double nz0 = (m11 * x1 + m12 * x2) / (l1 - l2);
  double nz1 = (m21 * x1 + m22 * x2) / (l1 - l2);
  double iv0 = ((l1 - kpc) / l1) * (1.0 - e1) + ((kpc - l2) / l2) * (1.0 - e2);
  double iv1 = (-kcp / l1) * (1.0 - e1) + (kcp / l2) * (1.0 - e2);
  double f = rateiv[0] / (l1 - l2);
  double ea = exp(-ka * t);
  double av0 = ((l1 - kpc) / (ka - l1)) * (e1 - ea) + ((kpc - l2) / (ka - l2)) * (e2 - ea);
  double av1 = (-kcp / (ka - l1)) * (e1 - ea) + (kcp / (ka - l2)) * (e2 - ea);
  double g = ka * x0 / (l1 - l2); /* :103 */
  /* aux = non_zero + infusion + absorption :105 (left to right) */
  xo[0] = x0 * ea;
  xo[1] = (nz0 + iv0 * f) + av0 * g;
  xo[2] = (nz1 + iv1 * f) + av1 * g;
  return 0;
}

/* Shared eigen-solve + coefficient block of the three-compartment kernels,
 * three_compartment_models.rs:24-77 (identical text at :135-188). */
typedef struct {
  double l1, l2, l3;
  double c[28]; /* c[1..27] */
} tc3;

static int three_cpt_coeffs(double k10, double k12, double k13, double k21, double k31, tc3* o) {
  double a = k10 + k12 + k13 + k21 + k31;
  double b = k10 * k21 + k13 * k21 + k10 * k31 + k12 * k31 + k21 * k31;
  double c = k10 * k21 * k31;
  double m = (3.0 * b - a * a) / 3.0;
  double n = (2.0 * (a * a * a) - 9.0 * a * b + 27.0 * c) / 27.0;
  double q = (n * n) / 4.0 + (m * m * m) / 27.0;
  if (q > 0.0) return PMX_PAIR_COMPLEX_ROOTS; /* panic! :32-34 */
  double alpha = sqrt(-q);
  double beta = -n / 2.0;
  double gamma = sqrt(beta * beta + alpha * alpha);
  double theta = atan2(alpha, beta);
  double cr = pow(gamma, 1.0 / 3.0);
  double l1 = a / 3.0 + cr * (cos(theta / 3.0) + sqrt(3.0) * sin(theta / 3.0));
  double l2 = a / 3.0 + cr * (cos(theta / 3.0) - sqrt(3.0) * sin(theta / 3.0));
  double l3 = a / 3.0 - (2.0 * cr * cos(theta / 3.0));
  o->l1 = l1;
  o->l2 = l2;
  o->l3 = l3;
  double d1 = ((l2 - l1) * (l3 - l1));
  double d2 = ((l1 - l2) * (l3 - l2));
  double d3 = ((l1 - l3) * (l2 - l3));
  double* C = o->c;
  C[1] = (k21 - l1) * (k31 - l1) / d1;
  C[2] = (k21 - l2) * (k31 - l2) / d2;
  C[3] = (k21 - l3) * (k31 - l3) / d3;
  C[4] = k21 * (k31 - l1) / d1;
  C[5] = k21 * (k31 - l2) / d2;
  C[6] = k21 * (k31 - l3) / d3;
  C[7] = k31 * (k21 - l1) / d1;
  C[8] = k31 * (k21 - l2) / d2;
  C[9] = k31 * (k21 - l3) / d3;
  C[10] = k12 * (k31 - l1) / d1;
  C[11] = k12 * (k31 - l2) / d2;
  C[12] = k12 * (k31 - l3) / d3;
  C[13] = ((k10 + k12 + k13 - l1) * (k31 - l1) - (k13 * k31)) / d1;
  C[14] = ((k10 + k12 + k13 - l2) * (k31 - l2) - (k13 * k31)) / d2;
  C[15] = ((k10 + k12 + k13 - l3) * (k31 - l3) - (k13 * k31)) / d3;
  C[16] = k12 * k31 / d1;
  C[17] = k12 * k31 / d2;
  C[18] = k12 * k31 / d3;
  C[19] = k13 * (k21 - l1) / d1;
  C[20] = k13 * (k21 - l2) / d2;
  C[21] = k13 * (k21 - l3) / d3;
  C[22] = k21 * k13 / d1;
  C[23] = k21 * k13 / d2;
  C[24] = k21 * k13 / d3;
  C[25] = ((k10 + k12 + k13 - l1) * (k21 - l1) - (k12 * k21)) / d1;
  C[26] = ((k10 + k12 + k13 - l2) * (k21 - l2) - (k12 * k21)) / d2;
  C[27] = ((k10 + k12 + k13 - l3) * (k21 - l3) - (k12 * k21)) / d3;
  return 0;
}

/* three_compartments, three_compartment_models.rs:17-109 */
static int k_three_compartments(const double* x, const double* p, double t, const double* rateiv, double* xo) {
  tc3 k;
  int rc = three_cpt_coeffs(p[0], p[1], p[2], p[3], p[4], &k);
  if (rc) return rc;
  const double* C = k.c;
  double e1 = exp(-(k.l1 * t)), e2 = exp(-(k.l2 * t)), e3 = exp(-(k.l3 * t));
  double m11 = C[1] * e1 + C[2] * e2 + C[3] * e3;
  double m12 = C[4] * e1 + C[5] * e2 + C[6] * e3;
  double m13 = C[7] * e1 + C[8] * e2 + C[9] * e3;
  double m21 = C[10] * e1 + C[11] * e2 + C[12] * e3;
  double m22 = C[13] * e1 + C[14] * e2 + C[15] * e3;
  double m23 = C[16] * e1 + C[17] * e2 + C[18] * e3;
  double m31 = C[19] * e1 + C[20] * e2 + C[21] * e3;
  double m32 = C[22] * e1 + C[23] * e2 + C[24] * e3;
  double m33 = C[25] * e1 + C[26] * e2 + C[27] * e3;
  double x0 = x[0], x1 = x[1], x2 = x[2];
  double nz0 = m11 * x0 + m12 * x1 + m13 * x2;
  double nz1 = m21 * x0 + m22 * x1 + m23 * x2;
  double nz2 = m31 * x0 + m32 * x1 + m33 * x2;
  double iv0 = ((1.0 - e1) * C[1] / k.l1) + ((1.0 - e2) * C[2] / k.l2) + ((1.0 - e3) * C[3] / k.l3);
  double iv1 = ((1.0 - e1) * C[10] / k.l1) + ((1.0 - e2) * C[11] / k.l2) + ((1.0 - e3) * C[12] / k.l3);
  double iv2 = ((1.0 - e1) * C[19] / k.l1) + ((1.0 - e2) * C[20] / k.l2) + ((1.0 - e3) * C[21] / k.l3);
  double r = rateiv[0];
  xo[0] = nz0 + iv0 * r;
  xo[1] = nz1 + iv1 * r;
  xo[2] = nz2 + iv2 * r;
  return 0;
}

/* three_compartments_with_absorption, three_compartment_models.rs:126-240 */
static int k_three_compartments_abs(const double* x, const double* p, double t, const double* rateiv, double* xo) {
  double ka = p[0];
  tc3 k;
  int rc = three_cpt_coeffs(p[1], p[2], p[3], p[4], p[5], &k);
  if (rc) return rc;
  const double* C = k.c;
  double e1 = exp(-(k.l1 * t)), e2 = exp(-(k.l2 * t)), e3 = exp(-(k.l3 * t));
  double m11 = C[1] * e1 + C[2] * e2 + C[3] * e3;
  double m12 = C[4] * e1 + C[5] * e2 + C[6] * e3;
  double m13 = C[7] * e1 + C[8] * e2 + C[9] * e3;
  double m21 = C[10] * e1 + C[11] * e2 + C[12] * e3;
  double m22 = C[13] * e1 + C[14] * e2 + C[15] * e3;
  double m23 = C[16] * e1 + C[17] * e2 + C[18] * e3;
  double m31 = C[19] * e1 + C[20] * e2 + C[21] * e3;
  double m32 = C[22] * e1 + C[23] * e2 + C[24] * e3;
  double m33 = C[25] * e1 + C[26] * e2 + C[27] * e3;
  double g = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
  double nz0 = m11 * x1 + m12 * x2 + m13 * x3;
  double nz1 = m21 * x1 + m22 * x2 + m23 * x3;
  double nz2 = m31 * x1 + m32 * x2 + m33 * x3;
  double iv0 = ((1.0 - e1) * C[1] / k.l1) + ((1.0 - e2) * C[2] / k.l2) + ((1.0 - e3) * C[3] / k.l3);
  double iv1 = ((1.0 - e1) * C[10] / k.l1) + ((1.0 - e2) * C[11] / k.l2) + ((1.0 - e3) * C[12] / k.l3);
  double iv2 = ((1.0 - e1) * C[19] / k.l1) + ((1.0 - e2) * C[20] / k.l2) + ((1.0 - e3) * C[21] / k.l3);
  double r = rateiv[0];
  double ea = exp(-ka * t);
  double av0 = (e1 - ea) * C[1] / (ka - k.l1) + (e2 - ea) * C[2] / (ka - k.l2) + (e3 - ea) * C[3] / (ka - k.l3);
  double av1 = (e1 - ea) * C[10] / (ka - k.l1) + (e2 - ea) * C[11] / (ka - k.l2) + (e3 - ea) * C[12] / (ka - k.l3);
  double av2 = (e1 - ea) * C[19] / (ka - k.l1) + (e2 - ea) * C[20] / (ka - k.l2) + (e3 - ea) * C[21] / (ka - k.l3);
  /* absorption = absorption_vector * ka * x[0] :230 => (v*ka)*x0 ; aux = nz + inf + abs :232 */
  xo[0] = g * ea;
  xo[1] = (nz0 + iv0 * r) + (av0 * ka) * g;
  xo[2] = (nz1 + iv1 * r) + (av1 * ka) * g;
  xo[3] = (nz2 + iv2 * r) + (av2 * ka) * g;
  return 0;
}

static int kernel_nstates(int kernel) {
  /* AnalyticalKernel::state_count, pharmsol-dsl/src/analysis.rs:259-270 */
  switch (kernel) {
    case PMX_K_ONE_COMPARTMENT:
    case PMX_K_ONE_COMPARTMENT_CL:
      return 1;
    case PMX_K_ONE_COMPARTMENT_CL_WITH_ABSORPTION:
    case PMX_K_ONE_COMPARTMENT_WITH_ABSORPTION:
    case PMX_K_TWO_COMPARTMENTS:
    case PMX_K_TWO_COMPARTMENTS_CL:
      return 2;
    case PMX_K_TWO_COMPARTMENTS_CL_WITH_ABSORPTION:
    case PMX_K_TWO_COMPARTMENTS_WITH_ABSORPTION:
    case PMX_K_THREE_COMPARTMENTS:
    case PMX_K_THREE_COMPARTMENTS_CL:
      return 3;
    case PMX_K_THREE_COMPARTMENTS_CL_WITH_ABSORPTION:
    case PMX_K_THREE_COMPARTMENTS_WITH_ABSORPTION:
      return 4;
    default:
      return -1;
  }
}

static int kernel_nparams(int kernel) {
  /* AnalyticalKernel::required_parameter_names, analysis.rs:240-255 */
  static const int n[12] = {1, 2, 3, 2, 3, 4, 5, 4, 5, 6, 7, 6};
  return (kernel >= 0 && kernel < 12) ? n[kernel] : -1;
}

/* Native (0-indexed) kernel dispatch incl. the CL re-parameterisations
 * (*_cl_models.rs): convert, then delegate. */
static int kernel_native(int kernel, const double* x, const double* p, double t, const double* rateiv, double* xo) {
  double q[8];
  switch (kernel) {
    case PMX_K_ONE_COMPARTMENT:
      return k_one_compartment(x, p, t, rateiv, xo);
    case PMX_K_ONE_COMPARTMENT_CL: /* one_compartment_cl_models.rs:16-22 */
      q[0] = p[0] / p[1];
      return k_one_compartment(x, q, t, rateiv, xo);
    case PMX_K_ONE_COMPARTMENT_CL_WITH_ABSORPTION: /* :38-45 */
      q[0] = p[0];
      q[1] = p[1] / p[2];
      return k_one_compartment_abs(x, q, t, rateiv, xo);
    case PMX_K_ONE_COMPARTMENT_WITH_ABSORPTION:
      return k_one_compartment_abs(x, p, t, rateiv, xo);
    case PMX_K_TWO_COMPARTMENTS:
      return k_two_compartments(x, p, t, rateiv, xo);
    case PMX_K_TWO_COMPARTMENTS_CL: /* two_compartment_cl_models.rs:16-26 */
      q[0] = p[0] / p[2];
      q[1] = p[1] / p[2];
      q[2] = p[1] / p[3];
      return k_two_compartments(x, q, t, rateiv, xo);
    case PMX_K_TWO_COMPARTMENTS_CL_WITH_ABSORPTION: /* :41-53 (note the [ke,ka,kcp,kpc] order) */
      q[0] = p[1] / p[3];
      q[1] = p[0];
      q[2] = p[2] / p[3];
      q[3] = p[2] / p[4];
      return k_two_compartments_abs(x, q, t, rateiv, xo);
    case PMX_K_TWO_COMPARTMENTS_WITH_ABSORPTION:
      return k_two_compartments_abs(x, p, t, rateiv, xo);
    case PMX_K_THREE_COMPARTMENTS:
      return k_three_compartments(x, p, t, rateiv, xo);
    case PMX_K_THREE_COMPARTMENTS_CL: /* three_compartment_cl_models.rs:16-31 */
      q[0] = p[0] / p[3];
      q[1] = p[1] / p[3];
      q[2] = p[2] / p[3];
      q[3] = p[1] / p[4];
      q[4] = p[2] / p[5];
      return k_three_compartments(x, q, t, rateiv, xo);
    case PMX_K_THREE_COMPARTMENTS_CL_WITH_ABSORPTION: /* :46-67 */
      q[0] = p[0];
      q[1] = p[1] / p[4];
      q[2] = p[2] / p[4];
      q[3] = p[3] / p[4];
      q[4] = p[2] / p[5];
      q[5] = p[3] / p[6];
      return k_three_compartments_abs(x, q, t, rateiv, xo);
    case PMX_K_THREE_COMPARTMENTS_WITH_ABSORPTION:
      return k_three_compartments_abs(x, p, t, rateiv, xo);
    default:
      return -1;
  }
}

/* wrap_pmetrics_analytical, analytical/mod.rs:78-90: drop slot 0 of x and
 * rateiv, run the native kernel, re-pad the result with a leading 0. */
static int kernel_pm(int kernel, const double* x, const double* p, double t, const double* rateiv, double* xo) {
  int ns = kernel_nstates(kernel);
  double tmp[PMX_MAX_STATES];
  int rc = kernel_native(kernel, x + 1, p, t, rateiv + 1, tmp);
  if (rc) return rc;
  xo[0] = 0.0;
  for (int i = 0; i < ns; i++) xo[i + 1] = tmp[i];
  return 0;
}

int32_t pmx_oracle_kernel(int32_t kernel, int32_t pm, const double* x, const double* p, double t,
                          const double* rateiv, double* xout) {
  return pm ? kernel_pm(kernel, x, p, t, rateiv, xout) : kernel_native(kernel, x, p, t, rateiv, xout);
}

/* ------------------------------------------------------------------------- */
/* Events                                                                     */
/* ------------------------------------------------------------------------- */
typedef struct {
  double time, value, duration;
  uint8_t kind;
  uint16_t io;
  int64_t src; /* index of the event in the caller's arrays (prediction row bookkeeping) */
} ev_t;

/* Event::cmp_time_then_type, event.rs:292-304 */
static int ev_less(const ev_t* a, const ev_t* b) {
  int64_t ka = total_key(a->time), kb = total_key(b->time);
  if (ka != kb) return ka < kb;
  return a->kind < b->kind; /* PMX_EV_* values are the reference ranks */
}

/* Occasion::sort (stable), structs.rs:669-671 */
static void ev_sort(ev_t* e, int64_t n) {
  for (int64_t i = 1; i < n; i++) {
    ev_t cur = e[i];
    int64_t j = i;
    while (j > 0 && ev_less(&cur, &e[j - 1])) {
      e[j] = e[j - 1];
      j--;
    }
    e[j] = cur;
  }
}

typedef struct {
  double time, amount, duration;
  int input;
} inf_t;

/* ------------------------------------------------------------------------- */
/* Model closures                                                             */
/* ------------------------------------------------------------------------- */
typedef struct {
  const pmx_model_desc* m;
  const cov_track* cov; /* [n_covariates] for the current occasion */
  const double* theta;  /* the support point as given (a user seq_eq sees it beside the solve's working vector) */
} ctx_t;

/* User closures of an Analytical model (Analytical::new(eq, seq_eq, lag, fa, init, out), analytical/mod.rs:102-118;
 * argument order of the reference's compiled kernels, src/dsl/native.rs:45-53).  Registered by the test harness,
 * which builds the very source text the device compiles with gcc (oracle/__init__.py compile_user).  g_user_mask =
 * PMX_FN_* bits of the registered closures; 0 = descriptor model. */
typedef void (*pmx_user_fn)(double t, const double* x, const double* p, const double* cov, const double* rateiv,
                            const double* derived, double* out);
/* DiffEq with its bolus argument (ode/mod.rs:115-132: diffeq(x, p, t, dx, bolus, rateiv, cov)) */
typedef void (*pmx_user_fn8)(double t, const double* x, const double* p, const double* cov, const double* rateiv,
                             const double* bolus, const double* derived, double* out);
static uint32_t g_user_mask = 0;
static pmx_user_fn g_user_outputs = 0, g_user_init = 0, g_user_derive = 0, g_user_lag = 0, g_user_fa = 0, g_user_seq = 0,
                   g_user_eq = 0, g_user_dynamics = 0;
static pmx_user_fn8 g_user_dynamics_bolus = 0;
void pmx_oracle_set_user(uint32_t mask, void** fns) {
  g_user_mask = mask;
  g_user_dynamics = (mask & PMX_FN_DYNAMICS) ? (pmx_user_fn)fns[0] : 0;
  g_user_outputs = (mask & PMX_FN_OUTPUTS) ? (pmx_user_fn)fns[1] : 0;
  g_user_init = (mask & PMX_FN_INIT) ? (pmx_user_fn)fns[2] : 0;
  g_user_derive = (mask & PMX_FN_DERIVE) ? (pmx_user_fn)fns[3] : 0;
  g_user_lag = (mask & PMX_FN_ROUTE_LAG) ? (pmx_user_fn)fns[4] : 0;
  g_user_fa = (mask & PMX_FN_ROUTE_BIOAVAILABILITY) ? (pmx_user_fn)fns[5] : 0;
  g_user_seq = (mask & PMX_FN_SEQ_EQ) ? (pmx_user_fn)fns[6] : 0;
  g_user_eq = (mask & PMX_FN_EQ) ? (pmx_user_fn)fns[7] : 0;
  g_user_dynamics_bolus = (mask & PMX_FN_DYNAMICS_BOLUS) ? (pmx_user_fn8)fns[8] : 0;
}
static void cov_values(const ctx_t* c, double t, double* cv);
static int is_user_analytical(const pmx_model_desc* m) { return m->eq_kind == PMX_EQ_ANALYTICAL && g_user_mask != 0; }
/* ODE::new(diffeq, lag, fa, init, out) with every closure a registered body (pmx_model_create_user, general ODE walker) */
static int is_user_ode(const pmx_model_desc* m) {
  return m->eq_kind == PMX_EQ_ODE && m->kernel == PMX_ODE_CUSTOM && (g_user_mask & (PMX_FN_DYNAMICS | PMX_FN_DYNAMICS_BOLUS)) != 0;
}
static int is_user_model(const pmx_model_desc* m) { return is_user_analytical(m) || is_user_ode(m); }

/* the `derive:` block: derived[d] = theta[src] * f0 * f1, covariates at `t`
 * (bindings.rs:98-117 -> fetch_cov!(cov, t, ...) src/lib.rs:433-443) */
static int eval_derived(const ctx_t* c, const double* theta, double t, double* derived) {
  const pmx_model_desc* m = c->m;
  if (is_user_model(m)) { /* the user's derive closure at t (every macro-lowered closure starts with it) */
    double cv[PMX_MAX_COVARIATES];
    for (int d = 0; d < m->n_derived; d++) derived[d] = 0.0;
    if (g_user_derive) {
      cov_values(c, t, cv);
      g_user_derive(t, 0, theta, m->n_covariates ? cv : 0, 0, 0, derived);
    }
    return 0;
  }
  for (int d = 0; d < m->n_derived; d++) {
    const pmx_derived* dd = &m->derived[d];
    double v = theta[dd->src_param];
    for (int k = 0; k < dd->n_factors; k++) {
      const pmx_factor* f = &dd->f[k];
      if (f->op == PMX_F_NONE) continue;
      double cv;
      if (cov_interp(&c->cov[f->cov], t, &cv)) return -1;
      double fac = (f->op == PMX_F_POW) ? pow(cv / f->ref, f->coef) : (1.0 + f->coef * (cv - f->ref));
      v = v * fac;
    }
    derived[d] = v;
  }
  return 0;
}

/* `eq` as the analytical! macro lowers it (expand/analytical.rs:208-294):
 * derive at the kernel's time argument, project into kernel order, call the
 * structure.  `t_cov` is dt under SEGMENT_DT and absolute next_t under
 * SEGMENT_END_ABS (include/pmx.h). */
static int model_eq(const ctx_t* c, const double* x, const double* pv, double dt, double t_cov, const double* rateiv,
                    double* xo) {
  const pmx_model_desc* m = c->m;
  if (m->kernel == PMX_ORACLE_K_TEST_SEQ_ACCUM) { /* analytical/mod.rs:494-498 */
    xo[0] = x[0] + pv[0] * dt;
    return 0;
  }
  if (m->kernel == PMX_ORACLE_K_TEST_RATEIV3) { /* analytical/mod.rs:531-535 */
    for (int i = 0; i < m->nstates; i++) xo[i] = x[i];
    xo[0] = x[0] + rateiv[3] * dt;
    return 0;
  }
  double derived[PMX_MAX_USER_DERIVED];
  double kp[PMX_MAX_KPARAMS];
  const double* p = pv;
  if (m->kernel == PMX_K_CUSTOM) { /* the user's own propagator: eq(x, p, dt, rateiv, cov) (analytical/mod.rs:363-364) */
    double cv[PMX_MAX_COVARIATES];
    if (!g_user_eq) return -2;
    cov_values(c, t_cov, cv);
    eval_derived(c, pv, t_cov, derived);
    for (int i = 0; i < m->nstates; i++) xo[i] = x[i];
    g_user_eq(dt, x, pv, m->n_covariates ? cv : 0, rateiv, (g_user_mask & PMX_FN_DERIVE) ? derived : 0, xo);
    return 0;
  }
  if (m->n_bind > 0) {
    if (m->n_derived > 0 && eval_derived(c, pv, t_cov, derived)) return -2;
    for (int j = 0; j < m->n_bind; j++)
      kp[j] = (m->bind[j].src == PMX_SRC_DERIVED) ? derived[m->bind[j].index] : pv[m->bind[j].index];
    p = kp;
  }
  return m->pmetrics_indexing ? kernel_pm(m->kernel, x, p, dt, rateiv, xo)
                              : kernel_native(m->kernel, x, p, dt, rateiv, xo);
}

/* `seq_eq` — empty for every macro model (expand/analytical.rs:121). */
static void cov_values(const ctx_t* c, double t, double* cv);
static void model_seq_eq(const ctx_t* c, double* pv, double t) {
  if (c->m->kernel == PMX_ORACLE_K_TEST_SEQ_ACCUM) pv[0] += 1.0; /* analytical/mod.rs:499-501 */
  if (is_user_analytical(c->m) && g_user_seq) { /* (self.seq_eq)(&mut parameters_v, next_t, covariates) :360 */
    double cv[PMX_MAX_COVARIATES];
    cov_values(c, t, cv);
    g_user_seq(t, 0, c->theta, c->m->n_covariates ? cv : 0, 0, 0, pv);
  }
}

/* covariates of the current occasion at time t (fetch_cov!, src/lib.rs:433-443); NaN where interpolation fails */
static void cov_values(const ctx_t* c, double t, double* cv) {
  for (int i = 0; i < c->m->n_covariates; i++)
    if (cov_interp(&c->cov[i], t, &cv[i])) cv[i] = NAN;
}

/* User bodies of a PMX_ODE_CUSTOM model (the closures of ODE::new, ode/mod.rs:115-132), argument order of the
 * reference's compiled kernels (src/dsl/native.rs:45-53).  Registered by the test harness, which builds the very
 * source text the device compiles with gcc (oracle/__init__.py compile_custom). */
typedef void (*pmx_custom_fn)(double t, const double* x, const double* p, const double* cov, const double* rateiv,
                              const double* derived, double* out);
static pmx_custom_fn g_custom_dynamics = 0, g_custom_outputs = 0, g_custom_init = 0;
void pmx_oracle_set_custom(void* dynamics, void* outputs, void* init) {
  g_custom_dynamics = (pmx_custom_fn)dynamics;
  g_custom_outputs = (pmx_custom_fn)outputs;
  g_custom_init = (pmx_custom_fn)init;
}

/* `out`: y[o] = x[state] / vol, derive at the observation time
 * (expand/analytical.rs:320-326; e.g. examples/analytical_readme.rs:21-23) */
static int model_out(const ctx_t* c, const double* x, const double* theta, double t_obs, double* y) {
  const pmx_model_desc* m = c->m;
  if (m->eq_kind == PMX_EQ_ODE && m->kernel == PMX_ODE_CUSTOM && !is_user_ode(m)) {
    double cv[PMX_MAX_COVARIATES];
    cov_values(c, t_obs, cv);
    g_custom_outputs(t_obs, x, theta, m->n_covariates ? cv : 0, 0, 0, y); /* y zeroed by the caller */
    return 0;
  }
  double derived[PMX_MAX_USER_DERIVED];
  if (is_user_model(m) && g_user_outputs) { /* out(x, p, t_obs, cov, y), derive first (expand/analytical.rs:320-326) */
    double cv[PMX_MAX_COVARIATES];
    cov_values(c, t_obs, cv);
    eval_derived(c, theta, t_obs, derived);
    g_user_outputs(t_obs, x, theta, m->n_covariates ? cv : 0, 0, (g_user_mask & PMX_FN_DERIVE) ? derived : 0, y);
    return 0;
  }
  int need = 0;
  for (int o = 0; o < m->nout; o++) need |= (m->out[o].vol_src == PMX_SRC_DERIVED);
  if (need && eval_derived(c, theta, t_obs, derived)) return -2;
  for (int o = 0; o < m->nout; o++) {
    const pmx_out* oo = &m->out[o];
    double xs = x[oo->state];
    if (oo->vol_src == PMX_SRC_PRIMARY)
      y[o] = xs / theta[oo->vol_index];
    else if (oo->vol_src == PMX_SRC_DERIVED)
      y[o] = xs / derived[oo->vol_index];
    else
      y[o] = xs;
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Analytical::solve — analytical/mod.rs:299-370                              */
/* ------------------------------------------------------------------------- */
static int cmp_double(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}

static int analytical_solve(const ctx_t* c, double* x, const double* theta, const inf_t* infusions, int n_inf,
                            double ti, double tf, double* ts_buf) {
  const pmx_model_desc* m = c->m;
  if (ti == tf) return 0; /* :308-310 */
  int nts = 0;
  ts_buf[nts++] = ti;
  ts_buf[nts++] = tf;
  for (int i = 0; i < n_inf; i++) { /* :316-325 */
    double t0 = infusions[i].time;
    double t1 = t0 + infusions[i].duration;
    if (t0 > ti && t0 < tf) ts_buf[nts++] = t0;
    if (t1 > ti && t1 < tf) ts_buf[nts++] = t1;
  }
  qsort(ts_buf, (size_t)nts, sizeof(double), cmp_double); /* :326 (no ties matter: equal values) */
  { /* dedup_by(|a,b| (a-b).abs() < 1e-12) :327 — compare against the last retained */
    int w = 1;
    for (int r = 1; r < nts; r++)
      if (!(fabs(ts_buf[r] - ts_buf[w - 1]) < 1e-12)) ts_buf[w++] = ts_buf[r];
    nts = w;
  }
  double current_t = ts_buf[0];
  double pv[PMX_MAX_PARAMS];
  for (int i = 0; i < m->nparams; i++) pv[i] = theta[i]; /* :331 rebuilt per solve */
  double rateiv[PMX_MAX_INPUTS + 1];
  double xo[PMX_MAX_STATES + 1];
  for (int k = 1; k < nts; k++) { /* :334-367 */
    double next_t = ts_buf[k];
    for (int i = 0; i < m->ndrugs; i++) rateiv[i] = 0.0;
    for (int i = 0; i < n_inf; i++) {
      double s = infusions[i].time;
      double e = s + infusions[i].duration;
      if (current_t >= s && next_t <= e) {
        int input = infusions[i].input;
        if (input >= m->ndrugs) return -PMX_ERR_INPUT_OUT_OF_RANGE; /* :349-354 */
        rateiv[input] += infusions[i].amount / infusions[i].duration;
      }
    }
    model_seq_eq(c, pv, next_t); /* :360 */
    double dt = next_t - current_t;
    double t_cov = (m->cov_time_mode == PMX_COV_TIME_SEGMENT_END_ABS) ? next_t : dt;
    int rc = model_eq(c, x, pv, dt, t_cov, rateiv, xo); /* :363-364 */
    if (rc) return rc;
    for (int i = 0; i < m->nstates; i++) x[i] = xo[i];
    current_t = next_t;
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* ODE back-end: fixed-step RK4 over the reference's event semantics          */
/* (ode/mod.rs:609-823, ode/closure.rs:16-195).  diffsol's adaptive BDF is    */
/* REPLACED by RK4 (north star); step-level parity with diffsol is unpinned.  */
/* ------------------------------------------------------------------------- */
static int ode_central_state(int model) {
  switch (model) {
    case PMX_ODE_ONE_CMT_IV:
    case PMX_ODE_TWO_CMT_IV:
    case PMX_ODE_THREE_CMT_IV:
    case PMX_ODE_ONE_CMT_MM:
      return 0;
    default:
      return 1;
  }
}
static int ode_nstates(int model) {
  static const int n[PMX_ODE_MODEL_COUNT] = {1, 2, 2, 3, 3, 4, 1};
  return (model >= 0 && model < PMX_ODE_MODEL_COUNT) ? n[model] : -1;
}

static int ode_nparams_of(int model) {
  static const int n[PMX_ODE_MODEL_COUNT] = {1, 2, 3, 4, 5, 6, 3};
  return (model >= 0 && model < PMX_ODE_MODEL_COUNT) ? n[model] : 0;
}

/* The user `diffeq` bodies (device functor registry mirrored here). */
static void ode_rhs(int model, const double* x, const double* p, double* dx) {
  switch (model) {
    case PMX_ODE_ONE_CMT_IV: /* examples/ode_readme.rs:17-19 */
      dx[0] = -p[0] * x[0];
      break;
    case PMX_ODE_ONE_CMT_ORAL:
      dx[0] = -p[0] * x[0];
      dx[1] = p[0] * x[0] - p[1] * x[1];
      break;
    case PMX_ODE_TWO_CMT_IV: /* two_compartment_models.rs:131-136 */
      dx[0] = -p[0] * x[0] - p[1] * x[0] + p[2] * x[1];
      dx[1] = p[1] * x[0] - p[2] * x[1];
      break;
    case PMX_ODE_TWO_CMT_ORAL: /* two_compartment_models.rs:188-194, p=[ke,ka,kcp,kpc] */
      dx[0] = -p[1] * x[0];
      dx[1] = -p[0] * x[1] + p[1] * x[0] - p[2] * x[1] + p[3] * x[2];
      dx[2] = p[2] * x[1] - p[3] * x[2];
      break;
    case PMX_ODE_THREE_CMT_IV: /* p=[k10,k12,k13,k21,k31] */
      dx[0] = -(p[0] + p[1] + p[2]) * x[0] + p[3] * x[1] + p[4] * x[2];
      dx[1] = p[1] * x[0] - p[3] * x[1];
      dx[2] = p[2] * x[0] - p[4] * x[2];
      break;
    case PMX_ODE_THREE_CMT_ORAL: /* p=[ka,k10,k12,k13,k21,k31] */
      dx[0] = -p[0] * x[0];
      dx[1] = p[0] * x[0] - (p[1] + p[2] + p[3]) * x[1] + p[4] * x[2] + p[5] * x[3];
      dx[2] = p[2] * x[1] - p[4] * x[2];
      dx[3] = p[3] * x[1] - p[5] * x[3];
      break;
    case PMX_ODE_ONE_CMT_MM: { /* p=[vmax,km,v] */
      double cc = x[0] / p[2];
      dx[0] = -p[0] * cc / (p[1] + cc);
    } break;
  }
}

/* PmRhs::call_inplace (closure.rs:344-357) + the ode! route injection
 * dx[dest] += rateiv[i] (expand/ode.rs:380-406). */
static void ode_f(const ctx_t* c, double t, const double* x, const double* p, const double* rate, double* dx) {
  const pmx_model_desc* m = c->m;
  if (is_user_ode(m)) { /* diffeq(x, p, t, dx, bolus = 0, rateiv, cov), derive first (ode/closure.rs:344-357) */
    double cv[PMX_MAX_COVARIATES], der[PMX_MAX_USER_DERIVED], zero[PMX_MAX_INPUTS] = {0};
    cov_values(c, t, cv);
    eval_derived(c, p, t, der);
    for (int i = 0; i < m->nstates; i++) dx[i] = 0.0;
    if (g_user_dynamics_bolus)
      g_user_dynamics_bolus(t, x, p, m->n_covariates ? cv : 0, rate, zero, (g_user_mask & PMX_FN_DERIVE) ? der : 0, dx);
    else
      g_user_dynamics(t, x, p, m->n_covariates ? cv : 0, rate, (g_user_mask & PMX_FN_DERIVE) ? der : 0, dx);
    return;
  }
  if (m->kernel == PMX_ODE_CUSTOM) { /* the body adds rateiv itself, like a hand-written closure */
    double cv[PMX_MAX_COVARIATES];
    cov_values(c, t, cv);
    for (int i = 0; i < m->nstates; i++) dx[i] = 0.0;
    g_custom_dynamics(t, x, p, m->n_covariates ? cv : 0, rate, 0, dx);
    return;
  }
  if (m->n_derived > 0 || m->n_bind > 0) { /* derive-style lines inside the body, covariates bound at t (expand/ode.rs:126-185) */
    double derived[PMX_MAX_USER_DERIVED], kp[PMX_MAX_KPARAMS];
    int np = ode_nparams_of(m->kernel);
    if (eval_derived(c, p, t, derived)) {
      for (int i = 0; i < m->nstates; i++) dx[i] = NAN;
      return;
    }
    for (int j = 0; j < np; j++)
      kp[j] = m->n_bind > 0 ? ((m->bind[j].src == PMX_SRC_DERIVED) ? derived[m->bind[j].index] : p[m->bind[j].index]) : p[j];
    ode_rhs(m->kernel, x, kp, dx);
  } else
    ode_rhs(m->kernel, x, p, dx);
  for (int i = 0; i < m->ndrugs; i++) {
    if (rate[i] != 0.0) {
      int dest = m->infusion_dest[i] >= 0 ? m->infusion_dest[i] : ode_central_state(m->kernel);
      dx[dest] += rate[i];
    }
  }
}

/* One constant-rate piece [t0, t1] with n = ceil((t1-t0)/h_max) classic RK4 steps. */
static void rk4_piece(const ctx_t* c, double* x, const double* p, const double* rate, double t0, double t1) {
  const pmx_model_desc* m = c->m;
  double dt = t1 - t0;
  if (!(dt > 0.0)) return;
  double nf = ceil(dt / m->rk4_h_max);
  if (nf < 1.0) nf = 1.0;
  int64_t n = (int64_t)nf;
  double h = dt / (double)n;
  int ns = m->nstates;
  double k1[PMX_MAX_STATES], k2[PMX_MAX_STATES], k3[PMX_MAX_STATES], k4[PMX_MAX_STATES], xt[PMX_MAX_STATES];
  for (int64_t s = 0; s < n; s++) {
    double t = t0 + (double)s * h; /* stage times t, t + h/2, t + h/2, t + h */
    ode_f(c, t, x, p, rate, k1);
    for (int i = 0; i < ns; i++) xt[i] = x[i] + (0.5 * h) * k1[i];
    ode_f(c, t + 0.5 * h, xt, p, rate, k2);
    for (int i = 0; i < ns; i++) xt[i] = x[i] + (0.5 * h) * k2[i];
    ode_f(c, t + 0.5 * h, xt, p, rate, k3);
    for (int i = 0; i < ns; i++) xt[i] = x[i] + h * k3[i];
    ode_f(c, t + h, xt, p, rate, k4);
    for (int i = 0; i < ns; i++) x[i] = x[i] + (h / 6.0) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
  }
}

/* ---- PMX_SOLVER_DOPRI5: Dormand-Prince 5(4) with step-size control (the build's stand-in for the reference's
 * adaptive diffsol solvers; same rule set as pmx_ode.hpp dopri5_try / dopri5_advance). ---------------------- */
static double dopri5_try(const ctx_t* c, const double* x, const double* p, const double* rate, double t, double h,
                         double* xn) {
  const pmx_model_desc* m = c->m;
  int ns = m->nstates;
  double k1[PMX_MAX_STATES], k2[PMX_MAX_STATES], k3[PMX_MAX_STATES], k4[PMX_MAX_STATES], k5[PMX_MAX_STATES],
      k6[PMX_MAX_STATES], k7[PMX_MAX_STATES], xt[PMX_MAX_STATES];
  ode_f(c, t, x, p, rate, k1);
  for (int i = 0; i < ns; i++) xt[i] = x[i] + h * (0.2 * k1[i]);
  ode_f(c, t + 0.2 * h, xt, p, rate, k2);
  for (int i = 0; i < ns; i++) xt[i] = x[i] + h * ((3.0 / 40.0) * k1[i] + (9.0 / 40.0) * k2[i]);
  ode_f(c, t + 0.3 * h, xt, p, rate, k3);
  for (int i = 0; i < ns; i++) xt[i] = x[i] + h * ((44.0 / 45.0) * k1[i] - (56.0 / 15.0) * k2[i] + (32.0 / 9.0) * k3[i]);
  ode_f(c, t + 0.8 * h, xt, p, rate, k4);
  for (int i = 0; i < ns; i++)
    xt[i] = x[i] + h * ((19372.0 / 6561.0) * k1[i] - (25360.0 / 2187.0) * k2[i] + (64448.0 / 6561.0) * k3[i] -
                        (212.0 / 729.0) * k4[i]);
  ode_f(c, t + (8.0 / 9.0) * h, xt, p, rate, k5);
  for (int i = 0; i < ns; i++)
    xt[i] = x[i] + h * ((9017.0 / 3168.0) * k1[i] - (355.0 / 33.0) * k2[i] + (46732.0 / 5247.0) * k3[i] +
                        (49.0 / 176.0) * k4[i] - (5103.0 / 18656.0) * k5[i]);
  ode_f(c, t + h, xt, p, rate, k6);
  for (int i = 0; i < ns; i++)
    xn[i] = x[i] + h * ((35.0 / 384.0) * k1[i] + (500.0 / 1113.0) * k3[i] + (125.0 / 192.0) * k4[i] -
                        (2187.0 / 6784.0) * k5[i] + (11.0 / 84.0) * k6[i]);
  ode_f(c, t + h, xn, p, rate, k7);
  double acc = 0.0;
  for (int i = 0; i < ns; i++) {
    double e = h * ((71.0 / 57600.0) * k1[i] - (71.0 / 16695.0) * k3[i] + (71.0 / 1920.0) * k4[i] -
                    (17253.0 / 339200.0) * k5[i] + (22.0 / 525.0) * k6[i] - (1.0 / 40.0) * k7[i]);
    double sc = m->ode_atol + m->ode_rtol * fmax(fabs(x[i]), fabs(xn[i]));
    double q = e / sc;
    acc += q * q;
  }
  return sqrt(acc / (double)ns);
}

/* ---- PMX_SOLVER_ROS2: the stiff option (the role of the reference's OdeSolver::Bdf / Sdirk, ode/mod.rs:60-77); ROS2
 * (Verwer et al. 1999), same rule set as pmx_ode.hpp ros2_try: forward-difference Jacobian and time derivative,
 * elimination in the natural order, error estimate from the embedded first-order solution. ------------------------- */
static double ros2_try(const ctx_t* c, const double* x, const double* p, const double* rate, double t, double h,
                       double* xn) {
  const pmx_model_desc* m = c->m;
  const int ns = m->nstates;
  const double kGamma = 1.7071067811865475, kSqrtEps = 1.4901161193847656e-08;
  const double gh = kGamma * h;
  double f0[PMX_MAX_STATES], f1[PMX_MAX_STATES], xt[PMX_MAX_STATES], W[PMX_MAX_STATES][PMX_MAX_STATES],
      ft[PMX_MAX_STATES], k1[PMX_MAX_STATES], k2[PMX_MAX_STATES];
  ode_f(c, t, x, p, rate, f0);
  for (int j = 0; j < ns; j++) {
    for (int i = 0; i < ns; i++) xt[i] = x[i];
    double d = kSqrtEps * fmax(fabs(x[j]), 1.0);
    xt[j] = x[j] + d;
    ode_f(c, t, xt, p, rate, f1);
    double s = -gh / d;
    for (int i = 0; i < ns; i++) W[i][j] = (f1[i] - f0[i]) * s + ((i == j) ? 1.0 : 0.0);
  }
  {
    double dt = kSqrtEps * fmax(fabs(t), 1.0);
    ode_f(c, t + dt, x, p, rate, f1);
    double s = gh / dt;
    for (int i = 0; i < ns; i++) ft[i] = (f1[i] - f0[i]) * s;
  }
  for (int k = 0; k < ns; k++) {
    double inv = 1.0 / W[k][k];
    for (int i = k + 1; i < ns; i++) {
      double l = W[i][k] * inv;
      W[i][k] = l;
      for (int j = k + 1; j < ns; j++) W[i][j] -= l * W[k][j];
    }
    W[k][k] = inv;
  }
#define PMX_ROS2_SOLVE(b)                                          \
  do {                                                             \
    for (int i = 1; i < ns; i++)                                   \
      for (int j = 0; j < i; j++) (b)[i] -= W[i][j] * (b)[j];      \
    for (int i = ns - 1; i >= 0; i--) {                            \
      for (int j = i + 1; j < ns; j++) (b)[i] -= W[i][j] * (b)[j]; \
      (b)[i] *= W[i][i];                                           \
    }                                                              \
  } while (0)
  for (int i = 0; i < ns; i++) k1[i] = f0[i] + ft[i];
  PMX_ROS2_SOLVE(k1);
  for (int i = 0; i < ns; i++) xt[i] = x[i] + h * k1[i];
  ode_f(c, t + h, xt, p, rate, f1);
  for (int i = 0; i < ns; i++) k2[i] = f1[i] - ft[i] - 2.0 * k1[i];
  PMX_ROS2_SOLVE(k2);
#undef PMX_ROS2_SOLVE
  double acc = 0.0;
  for (int i = 0; i < ns; i++) {
    xn[i] = x[i] + h * (1.5 * k1[i] + 0.5 * k2[i]);
    double e = (0.5 * h) * (k1[i] + k2[i]);
    double sc = m->ode_atol + m->ode_rtol * fmax(fabs(x[i]), fabs(xn[i]));
    double q = e / sc;
    acc += q * q;
  }
  return sqrt(acc / (double)ns);
}

typedef struct {
  double h;
  int failed;
} adapt_t;

static void dopri5_piece(const ctx_t* c, double* x, const double* p, const double* rate, double t0, double t1,
                         adapt_t* as) {
  const pmx_model_desc* m = c->m;
  double t = t0;
  for (int64_t guard = 0; guard < 10000000; guard++) {
    double left = t1 - t;
    if (!(left > 0.0)) return;
    double h = fmin(as->h, m->rk4_h_max);
    int clipped = h >= left;
    if (clipped) h = left;
    double xn[PMX_MAX_STATES];
    int stiff = m->ode_solver == PMX_SOLVER_ROS2;
    double err = stiff ? ros2_try(c, x, p, rate, t, h, xn) : dopri5_try(c, x, p, rate, t, h, xn);
    int ok = err <= 1.0;
    double fac = (err > 0.0) ? 0.9 * pow(err, stiff ? -0.5 : -0.2) : 5.0;
    if (!(fac >= 0.2)) fac = 0.2;
    if (fac > 5.0) fac = 5.0;
    if (!ok && fac > 1.0) fac = 1.0;
    double h_next = h * fac;
    if (ok) {
      for (int i = 0; i < m->nstates; i++) x[i] = xn[i];
      t = clipped ? t1 : (t + h);
      as->h = clipped ? fmax(as->h, h_next) : h_next;
      if (clipped) return;
      continue;
    }
    as->h = h_next;
    if (!(h_next > 1.0e-13 * fmax(1.0, fabs(t)))) {
      as->failed = 1;
      return;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* One (subject, support point): Equation::simulate_subject_dense             */
/* equation/mod.rs:480-516 (analytical) / ode/mod.rs:306-461 (ODE)            */
/* ------------------------------------------------------------------------- */
typedef struct {
  ev_t* ev;        /* scratch: events of one occasion */
  inf_t* inf;      /* scratch: infusions pushed so far */
  double* ts;      /* scratch: breakpoints */
  double* bounds;  /* scratch: ODE infusion boundaries */
  cov_track* cov;  /* [n_occ_of_subject * n_cov] prepared once per subject */
  int64_t cap;
} scratch_t;

static int simulate_pair(const pmx_model_desc* m, const pmx_population_desc* pop, int64_t subj, const double* theta,
                         double* pred, int64_t pred_stride, uint8_t* status, scratch_t* sc) {
  int64_t occ0 = pop->subj_occ_off[subj], occ1 = pop->subj_occ_off[subj + 1];
  int ncov = pop->n_covariates;
  int64_t row = 0; /* prediction row within the subject */
  uint8_t st = PMX_PAIR_OK;
  adapt_t adapt = {m->rk4_h_max, 0}; /* adaptive solver: the proposal restarts with every subject */
  ctx_t ctx;
  ctx.m = m;
  ctx.theta = theta;
  const int user = is_user_model(m);
  for (int64_t oc = occ0; oc < occ1; oc++) { /* for occasion in subject.occasions() :494 */
    ctx.cov = sc->cov + (oc - occ0) * ncov;
    int occ_index = pop->occ_index ? pop->occ_index[oc] : (int)(oc - occ0);
    /* initial_state: zeros; init(theta, 0.0, cov, x) only when occasion_index == 0
     * (analytical/mod.rs:409-426, ode/mod.rs:536-549) */
    double x[PMX_MAX_STATES + 1];
    for (int i = 0; i <= PMX_MAX_STATES; i++) x[i] = 0.0;
    if (occ_index == 0) {
      for (int i = 0; i < m->nstates; i++)
        if (m->init_param[i] >= 0) x[i] = theta[m->init_param[i]];
      if (m->eq_kind == PMX_EQ_ODE && m->kernel == PMX_ODE_CUSTOM && !is_user_ode(m) && g_custom_init) {
        double cv[PMX_MAX_COVARIATES];
        cov_values(&ctx, 0.0, cv);
        g_custom_init(0.0, x, theta, m->n_covariates ? cv : 0, 0, 0, x);
      }
      if (user && g_user_init) { /* init(p, 0.0, cov, x), analytical/mod.rs:417-423 */
        double cv[PMX_MAX_COVARIATES], der[PMX_MAX_USER_DERIVED];
        cov_values(&ctx, 0.0, cv);
        eval_derived(&ctx, theta, 0.0, der);
        for (int i = 0; i < m->nstates; i++) x[i] = 0.0;
        g_user_init(0.0, x, theta, m->n_covariates ? cv : 0, 0, (g_user_mask & PMX_FN_DERIVE) ? der : 0, x);
      }
    }
    /* resolve_occasion_events: clone + process_events (equation/mod.rs:247-273, structs.rs:681-690) */
    int64_t e0 = pop->occ_ev_off[oc], e1 = pop->occ_ev_off[oc + 1];
    int64_t n = e1 - e0;
    ev_t* ev = sc->ev;
    for (int64_t i = 0; i < n; i++) {
      ev[i].time = pop->ev_time[e0 + i];
      ev[i].value = pop->ev_value[e0 + i];
      ev[i].duration = pop->ev_duration[e0 + i];
      ev[i].kind = pop->ev_kind[e0 + i];
      ev[i].io = pop->ev_io[e0 + i];
      ev[i].src = e0 + i;
    }
    if (!pop->presorted) ev_sort(ev, n); /* Subject::new sorts every occasion, structs.rs:363-369 */
    /* Occasion::initial_time(): the earliest RECORDED event time - the ODE solver's t0 is taken from the occasion as the
     * data holds it, before lag moves any bolus (ode/mod.rs:348 `.t0(occasion.initial_time())`, structs.rs:782-793) */
    double t0_recorded = 0.0;
    for (int64_t i = 0; i < n; i++)
      if (i == 0 || ev[i].time < t0_recorded) t0_recorded = ev[i].time;
    { /* add_lagtime, structs.rs:611-643 */
      int shifted = 0;
      for (int64_t i = 0; i < n; i++) {
        if (ev[i].kind != PMX_EV_BOLUS) continue;
        int input = ev[i].io;
        if (user && g_user_lag) { /* fn_lag(&parameters, bolus.time(), covariates), structs.rs:629 */
          double cv[PMX_MAX_COVARIATES], der[PMX_MAX_USER_DERIVED], lagv[PMX_MAX_INPUTS];
          cov_values(&ctx, ev[i].time, cv);
          eval_derived(&ctx, theta, ev[i].time, der);
          for (int k = 0; k < PMX_MAX_INPUTS; k++) lagv[k] = 0.0;
          g_user_lag(ev[i].time, 0, theta, m->n_covariates ? cv : 0, 0, (g_user_mask & PMX_FN_DERIVE) ? der : 0, lagv);
          double l = input < PMX_MAX_INPUTS ? lagv[input] : 0.0;
          if (l != 0.0) {
            ev[i].time += l;
            shifted = 1;
          }
        } else if (input < PMX_MAX_INPUTS && m->lag_param[input] >= 0) {
          double l = theta[m->lag_param[input]];
          if (l != 0.0) {
            ev[i].time += l;
            shifted = 1;
          }
        }
      }
      if (shifted) ev_sort(ev, n);
    }
    /* add_bioavailability, structs.rs:645-666 */
    for (int64_t i = 0; i < n; i++) {
      if (ev[i].kind != PMX_EV_BOLUS) continue;
      int input = ev[i].io;
      if (user && g_user_fa) { /* fn_fa(&parameters, bolus.time() [already shifted], covariates), structs.rs:661 */
        double cv[PMX_MAX_COVARIATES], der[PMX_MAX_USER_DERIVED], fav[PMX_MAX_INPUTS];
        cov_values(&ctx, ev[i].time, cv);
        eval_derived(&ctx, theta, ev[i].time, der);
        for (int k = 0; k < PMX_MAX_INPUTS; k++) fav[k] = 1.0;
        g_user_fa(ev[i].time, 0, theta, m->n_covariates ? cv : 0, 0, (g_user_mask & PMX_FN_DERIVE) ? der : 0, fav);
        if (input < PMX_MAX_INPUTS) ev[i].value = ev[i].value * fav[input];
      } else if (input < PMX_MAX_INPUTS && m->fa_param[input] >= 0) ev[i].value = ev[i].value * theta[m->fa_param[input]];
    }

    if (m->eq_kind == PMX_EQ_ANALYTICAL) {
      int n_inf = 0;
      for (int64_t i = 0; i < n; i++) { /* simulate_event, equation/mod.rs:300-358 */
        const ev_t* e = &ev[i];
        if (e->kind == PMX_EV_BOLUS) {
          if ((int)e->io >= m->ndrugs) return PMX_ERR_INPUT_OUT_OF_RANGE; /* :322-327 */
          x[e->io] += e->value;                                            /* :328 */
        } else if (e->kind == PMX_EV_INFUSION) {
          sc->inf[n_inf].time = e->time;
          sc->inf[n_inf].amount = e->value;
          sc->inf[n_inf].duration = e->duration;
          sc->inf[n_inf].input = e->io;
          n_inf++; /* :330-332 */
        } else {   /* process_observation, analytical/mod.rs:373-407 */
          double y[PMX_MAX_OUT] = {0.0, 0.0, 0.0, 0.0};
          if ((int)e->io >= m->nout) return PMX_ERR_OUTEQ_OUT_OF_RANGE;
          if (model_out(&ctx, x, theta, e->time, y)) return PMX_ERR_INVALID_ARGUMENT;
          double pr = y[e->io];
          if (st == PMX_PAIR_OK && !isfinite(pr)) st = PMX_PAIR_NONFINITE;
          pred[row * pred_stride] = pr;
          row++;
        }
        if (i + 1 < n) { /* :347-356 */
          int rc = analytical_solve(&ctx, x, theta, sc->inf, n_inf, e->time, ev[i + 1].time, sc->ts);
          if (rc == PMX_PAIR_COMPLEX_ROOTS) {
            /* the reference panics here; the pair is flagged and its remaining rows are NaN */
            st = PMX_PAIR_COMPLEX_ROOTS;
            for (int s2 = 0; s2 <= PMX_MAX_STATES; s2++) x[s2] = NAN;
          } else if (rc < 0) {
            return -rc == PMX_ERR_INPUT_OUT_OF_RANGE ? PMX_ERR_INPUT_OUT_OF_RANGE : PMX_ERR_INVALID_ARGUMENT;
          }
        }
      }
    } else { /* ODE::run_events, ode/mod.rs:609-823 */
      /* InfusionSchedule::new over ALL infusions of the occasion, closure.rs:109-180 */
      int n_inf = 0, nb = 0;
      for (int64_t i = 0; i < n; i++) {
        if (ev[i].kind != PMX_EV_INFUSION) continue;
        if (ev[i].duration <= 0.0) continue; /* closure.rs:127-129 */
        if ((int)ev[i].io >= m->ndrugs) return PMX_ERR_INPUT_OUT_OF_RANGE;
        sc->inf[n_inf].time = ev[i].time;
        sc->inf[n_inf].amount = ev[i].value;
        sc->inf[n_inf].duration = ev[i].duration;
        sc->inf[n_inf].input = ev[i].io;
        sc->bounds[nb++] = ev[i].time;
        sc->bounds[nb++] = ev[i].time + ev[i].duration;
        n_inf++;
      }
      qsort(sc->bounds, (size_t)nb, sizeof(double), cmp_double);
      { /* boundary_times.dedup() (exact) closure.rs:143-148 */
        int w = 0;
        for (int r = 0; r < nb; r++)
          if (w == 0 || sc->bounds[r] != sc->bounds[w - 1]) sc->bounds[w++] = sc->bounds[r];
        nb = w;
      }
      /* The solver clock: starts at the occasion's recorded initial time and only ever moves forward to the time of
       * the NEXT event (`while next_event_time > solver.state().t`, :719-721).  The first event of the re-sorted list is
       * therefore applied at t0 without integration, whatever its own (lagged) time is. */
      double t = t0_recorded;
      int bcur = 0;
      for (int64_t i = 0; i < n; i++) {
        const ev_t* e = &ev[i];
        if (e->kind == PMX_EV_BOLUS) {
          /* y += f(y, bolus) - f(y, 0) == amount at the route's destination, :659-686 */
          if ((int)e->io >= m->ndrugs) return PMX_ERR_INPUT_OUT_OF_RANGE;
          if (is_user_ode(m) && g_user_dynamics_bolus) {
            /* state += diffeq(y, t_event, bolus_v, zero_rateiv) - diffeq(y, t_event, zero_bolus, zero_rateiv), :647-686 */
            double cv[PMX_MAX_COVARIATES], der[PMX_MAX_USER_DERIVED], zero[PMX_MAX_INPUTS] = {0}, bv[PMX_MAX_INPUTS] = {0};
            double f0[PMX_MAX_STATES], f1[PMX_MAX_STATES];
            cov_values(&ctx, e->time, cv);
            eval_derived(&ctx, theta, e->time, der);
            const double* dp = (g_user_mask & PMX_FN_DERIVE) ? der : 0;
            bv[e->io] = e->value;
            for (int k = 0; k < m->nstates; k++) f0[k] = f1[k] = 0.0;
            g_user_dynamics_bolus(e->time, x, theta, m->n_covariates ? cv : 0, zero, zero, dp, f0);
            g_user_dynamics_bolus(e->time, x, theta, m->n_covariates ? cv : 0, zero, bv, dp, f1);
            for (int k = 0; k < m->nstates; k++) x[k] += f1[k] - f0[k]; /* axpy(-1, without, 1) then y += */
          } else {
            int dest = m->bolus_dest[e->io] >= 0 ? m->bolus_dest[e->io] : (int)e->io;
            x[dest] += e->value;
          }
        } else if (e->kind == PMX_EV_OBSERVATION) { /* :692-715 */
          double y[PMX_MAX_OUT] = {0.0, 0.0, 0.0, 0.0};
          if ((int)e->io >= m->nout) return PMX_ERR_OUTEQ_OUT_OF_RANGE;
          if (model_out(&ctx, x, theta, e->time, y)) return PMX_ERR_INVALID_ARGUMENT;
          double pr = y[e->io];
          if (adapt.failed) { /* step-size underflow before this row */
            if (st == PMX_PAIR_OK) st = PMX_PAIR_SOLVER_FAIL;
            pr = NAN;
          }
          if (st == PMX_PAIR_OK && !isfinite(pr)) st = PMX_PAIR_NONFINITE;
          pred[row * pred_stride] = pr;
          row++;
        }
        if (i + 1 < n) { /* advance, :719-819 */
          double next_t = ev[i + 1].time;
          while (next_t > t) {
            while (bcur < nb && sc->bounds[bcur] <= t) bcur++; /* :722-726 */
            double stop = next_t;
            if (bcur < nb && sc->bounds[bcur] <= next_t) stop = sc->bounds[bcur++]; /* :728-739 */
            /* rate on [t, stop): right-continuous at t (closure.rs:80-99) == sum of
             * infusions with s <= t < e; left-continuous at the stop, i.e. the same value. */
            double rate[PMX_MAX_INPUTS];
            for (int k = 0; k < m->ndrugs; k++) rate[k] = 0.0;
            for (int k = 0; k < n_inf; k++) {
              double s = sc->inf[k].time, en = s + sc->inf[k].duration;
              if (s <= t && t < en) rate[sc->inf[k].input] += sc->inf[k].amount / sc->inf[k].duration;
            }
            if (m->ode_solver != PMX_SOLVER_RK4)
              dopri5_piece(&ctx, x, theta, rate, t, stop, &adapt);
            else
              rk4_piece(&ctx, x, theta, rate, t, stop);
            t = stop;
          }
        }
      }
    }
  }
  if (status) *status = st;
  return PMX_OK;
}

/* ------------------------------------------------------------------------- */
static int validate(const pmx_model_desc* m, const pmx_population_desc* pop) {
  if (!m || !pop) FAIL(PMX_ERR_INVALID_ARGUMENT, "null descriptor");
  if (m->nstates < 1 || m->nstates > PMX_MAX_STATES) FAIL(PMX_ERR_INVALID_ARGUMENT, "nstates out of range");
  if (m->ndrugs < 0 || m->ndrugs > PMX_MAX_INPUTS) FAIL(PMX_ERR_INVALID_ARGUMENT, "ndrugs out of range");
  if (m->nout < 1 || m->nout > PMX_MAX_OUT) FAIL(PMX_ERR_INVALID_ARGUMENT, "nout out of range");
  if (m->nparams < 0 || m->nparams > PMX_MAX_PARAMS) FAIL(PMX_ERR_INVALID_ARGUMENT, "nparams out of range");
  if (m->n_covariates != pop->n_covariates)
    FAIL(PMX_ERR_INVALID_ARGUMENT, "model declares %d covariates, population carries %d", m->n_covariates,
         pop->n_covariates);
  if (m->eq_kind == PMX_EQ_ANALYTICAL) {
    if (m->kernel == PMX_K_CUSTOM) {
      if (!g_user_eq) FAIL(PMX_ERR_INVALID_ARGUMENT, "PMX_K_CUSTOM: no user eq registered (pmx_oracle_set_user)");
    } else if (m->kernel < PMX_ORACLE_K_TEST_SEQ_ACCUM) {
      int ns = kernel_nstates(m->kernel);
      if (ns < 0) FAIL(PMX_ERR_INVALID_ARGUMENT, "unknown analytical kernel %d", m->kernel);
      if (m->nstates < ns + (m->pmetrics_indexing ? 1 : 0))
        FAIL(PMX_ERR_INVALID_ARGUMENT, "kernel needs %d states, model has %d", ns, m->nstates);
      int np = kernel_nparams(m->kernel);
      if (m->n_bind == 0 && m->nparams < np)
        FAIL(PMX_ERR_INVALID_ARGUMENT, "kernel needs %d leading params, model has %d", np, m->nparams);
      if (m->n_bind != 0 && m->n_bind != np)
        FAIL(PMX_ERR_INVALID_ARGUMENT, "kernel needs %d bindings, got %d", np, m->n_bind);
    }
  } else if (m->eq_kind == PMX_EQ_ODE) {
    if (m->kernel == PMX_ODE_CUSTOM) {
      if (is_user_ode(m)) {
        if (!g_user_outputs) FAIL(PMX_ERR_INVALID_ARGUMENT, "user ODE model: no outputs closure registered");
      } else if (!g_custom_dynamics || !g_custom_outputs)
        FAIL(PMX_ERR_INVALID_ARGUMENT, "custom ODE bodies not registered");
    } else {
      if (ode_nstates(m->kernel) < 0) FAIL(PMX_ERR_INVALID_ARGUMENT, "unknown ODE model %d", m->kernel);
      if (m->nstates < ode_nstates(m->kernel)) FAIL(PMX_ERR_INVALID_ARGUMENT, "ODE model needs more states");
    }
    if (!(m->rk4_h_max > 0.0)) FAIL(PMX_ERR_INVALID_ARGUMENT, "rk4_h_max must be > 0");
  } else
    FAIL(PMX_ERR_INVALID_ARGUMENT, "unknown eq_kind %d", m->eq_kind);
  return PMX_OK;
}

int64_t pmx_oracle_n_observations(const pmx_population_desc* pop) {
  int64_t n = 0;
  for (int64_t i = 0; i < pop->n_events; i++) n += (pop->ev_kind[i] == PMX_EV_OBSERVATION);
  return n;
}

static int scratch_init(scratch_t* sc, const pmx_population_desc* pop) {
  int64_t cap = 1;
  for (int64_t oc = 0; oc < pop->n_occasions; oc++) {
    int64_t n = pop->occ_ev_off[oc + 1] - pop->occ_ev_off[oc];
    if (n > cap) cap = n;
  }
  sc->cap = cap;
  sc->ev = (ev_t*)malloc(sizeof(ev_t) * (size_t)cap);
  sc->inf = (inf_t*)malloc(sizeof(inf_t) * (size_t)cap);
  sc->ts = (double*)malloc(sizeof(double) * (size_t)(2 * cap + 4));
  sc->bounds = (double*)malloc(sizeof(double) * (size_t)(2 * cap + 4));
  sc->cov = NULL;
  return 0;
}
static void scratch_free(scratch_t* sc) {
  free(sc->ev);
  free(sc->inf);
  free(sc->ts);
  free(sc->bounds);
}

/* Covariate tracks of one subject (per occasion x covariate). */
static cov_track* subject_cov(const pmx_population_desc* pop, int64_t subj) {
  int ncov = pop->n_covariates;
  int64_t occ0 = pop->subj_occ_off[subj], occ1 = pop->subj_occ_off[subj + 1];
  if (ncov == 0) return NULL;
  cov_track* tr = (cov_track*)malloc(sizeof(cov_track) * (size_t)((occ1 - occ0) * ncov + 1));
  for (int64_t oc = occ0; oc < occ1; oc++)
    for (int c = 0; c < ncov; c++) {
      int64_t idx = oc * ncov + c;
      int64_t k0 = pop->cov_knot_off[idx], k1 = pop->cov_knot_off[idx + 1];
      cov_build(pop->cov_knot_time + k0, pop->cov_knot_value + k0, k1 - k0, pop->cov_fixed ? pop->cov_fixed[idx] : 0,
                &tr[(oc - occ0) * ncov + c]);
    }
  return tr;
}
static void subject_cov_free(const pmx_population_desc* pop, int64_t subj, cov_track* tr) {
  if (!tr) return;
  int64_t n = (pop->subj_occ_off[subj + 1] - pop->subj_occ_off[subj]) * pop->n_covariates;
  for (int64_t i = 0; i < n; i++) free(tr[i].seg);
  free(tr);
}

static int64_t* obs_offsets(const pmx_population_desc* pop) {
  int64_t* off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(pop->n_subjects + 1));
  off[0] = 0;
  for (int64_t s = 0; s < pop->n_subjects; s++) {
    int64_t e0 = pop->occ_ev_off[pop->subj_occ_off[s]], e1 = pop->occ_ev_off[pop->subj_occ_off[s + 1]];
    int64_t n = 0;
    for (int64_t i = e0; i < e1; i++) n += (pop->ev_kind[i] == PMX_EV_OBSERVATION);
    off[s + 1] = off[s] + n;
  }
  return off;
}

/* log_likelihood_matrix loop nest, likelihood/matrix.rs:79-98:
 * rayon over SUBJECTS, serial over SUPPORT POINTS. */
int32_t pmx_oracle_predict(const pmx_model_desc* model, const pmx_population_desc* pop, const double* theta,
                           int64_t n_support, double* pred, int64_t ld_pred, uint8_t* status, int32_t nthreads) {
  g_err[0] = 0;
  int rc = validate(model, pop);
  if (rc) return rc;
  if (ld_pred < n_support) FAIL(PMX_ERR_INVALID_ARGUMENT, "ld_pred < n_support");
  int64_t* off = obs_offsets(pop);
  int failed = PMX_OK;
  int any_pair_failed = 0;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
  nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
  {
    scratch_t sc;
    scratch_init(&sc, pop);
#pragma omp for schedule(dynamic)
    for (int64_t s = 0; s < pop->n_subjects; s++) {
      sc.cov = subject_cov(pop, s);
      for (int64_t p = 0; p < n_support; p++) {
        uint8_t st = 0;
        int r = simulate_pair(model, pop, s, theta + p * model->nparams, pred + off[s] * ld_pred + p, ld_pred, &st,
                              &sc);
        if (status) status[s * n_support + p] = st;
        if (st) {
#pragma omp atomic write
          any_pair_failed = 1;
        }
        if (r != PMX_OK) {
#pragma omp atomic write
          failed = r;
        }
      }
      subject_cov_free(pop, s, sc.cov);
    }
    scratch_free(&sc);
  }
  free(off);
  if (failed != PMX_OK) FAIL(failed, "simulation failed with status %d (input/outeq out of range?)", failed);
  if (any_pair_failed) FAIL(PMX_ERR_PAIR_FAILED, "at least one (subject, support point) pair failed");
  return PMX_OK;
}

int32_t pmx_oracle_predict_batch(const pmx_model_desc* model, const pmx_population_desc* pop, const double* theta,
                                 double* pred, uint8_t* status, int32_t nthreads) {
  g_err[0] = 0;
  int rc = validate(model, pop);
  if (rc) return rc;
  int64_t* off = obs_offsets(pop);
  int failed = PMX_OK;
  int any_pair_failed = 0;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
  nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
  {
    scratch_t sc;
    scratch_init(&sc, pop);
#pragma omp for schedule(dynamic, 16)
    for (int64_t s = 0; s < pop->n_subjects; s++) {
      sc.cov = subject_cov(pop, s);
      uint8_t st = 0;
      int r = simulate_pair(model, pop, s, theta + s * model->nparams, pred + off[s], 1, &st, &sc);
      if (status) status[s] = st;
      if (st) {
#pragma omp atomic write
        any_pair_failed = 1;
      }
      if (r != PMX_OK) {
#pragma omp atomic write
        failed = r;
      }
      subject_cov_free(pop, s, sc.cov);
    }
    scratch_free(&sc);
  }
  free(off);
  if (failed != PMX_OK) FAIL(failed, "simulation failed with status %d", failed);
  if (any_pair_failed) FAIL(PMX_ERR_PAIR_FAILED, "at least one subject failed");
  return PMX_OK;
}

/* ------------------------------------------------------------------------- */
/* log-likelihood: likelihood/{distributions,prediction,subject,matrix}.rs      */
/* ------------------------------------------------------------------------- */
#define PMX_LOG_2PI 1.8378770664093453 /* distributions.rs:12 */

/* lognormpdf, distributions.rs:31-34 */
double pmx_oracle_lognormpdf(double obs, double pred, double sigma) {
  double diff = obs - pred;
  return -0.5 * PMX_LOG_2PI - log(sigma) - (diff * diff) / (2.0 * sigma * sigma);
}

/* AssayErrorModel::sigma, error_model.rs:1045-1080 */
int32_t pmx_oracle_sigma(const pmx_error_model* em, double y, double* sigma) {
  double alpha = em->c[0] + em->c[1] * y + em->c[2] * (y * y) + em->c[3] * (y * y * y);
  double s;
  if (em->kind == PMX_EM_ADDITIVE)
    s = sqrt(alpha * alpha + em->scalar * em->scalar);
  else if (em->kind == PMX_EM_PROPORTIONAL)
    s = em->scalar * alpha;
  else
    return PMX_ERR_ERROR_MODEL; /* MissingErrorModel */
  if (s < 0.0) return PMX_ERR_ERROR_MODEL;   /* NegativeSigma */
  if (!isfinite(s)) return PMX_ERR_ERROR_MODEL; /* NonFiniteSigma */
  *sigma = s;
  return PMX_OK;
}

/* lognormcdf (upper = 0) / lognormccdf (upper = 1), distributions.rs:52-103.  The normal CDF is statrs 0.19's
 * (Cargo.toml; un-vendored): Normal::cdf(x) = 0.5 * erfc((mean - x) / (std_dev * SQRT_2)), Normal::new rejecting
 * std_dev <= 0 or NaN.  Its erfc is a rational approximation good to ~1e-15; glibc's erfc stands in here. */
int32_t pmx_oracle_lognormcdf(double obs, double pred, double sigma, int32_t upper, double* out) {
  if (!(sigma > 0.0) || isnan(pred)) return PMX_ERR_ERROR_MODEL; /* Normal::new -> Err -> NegativeSigma */
  double cdf = 0.5 * erfc((pred - obs) / (sigma * 1.4142135623730951));
  double z = (obs - pred) / sigma;
  if (!upper) {
    if (cdf <= 0.0) {
      if (z < -37.0) {
        *out = pmx_oracle_lognormpdf(obs, pred, sigma) - log(fabs(z));
        return PMX_OK;
      }
      return PMX_ERR_ERROR_MODEL;
    }
    *out = log(cdf);
    return PMX_OK;
  }
  double sf = 1.0 - cdf;
  if (sf <= 0.0) {
    if (z > 37.0) {
      *out = pmx_oracle_lognormpdf(obs, pred, sigma) - log(z);
      return PMX_OK;
    }
    return PMX_ERR_ERROR_MODEL;
  }
  *out = log(sf);
  return PMX_OK;
}

int32_t pmx_oracle_loglik(const pmx_model_desc* model, const pmx_population_desc* pop, const pmx_error_model* em,
                          const double* theta, int64_t n_support, double* ll, int64_t ld_ll, uint8_t* status,
                          int32_t nthreads) {
  g_err[0] = 0;
  int rc = validate(model, pop);
  if (rc) return rc;
  if (ld_ll < n_support) FAIL(PMX_ERR_INVALID_ARGUMENT, "ld_ll < n_support");
  int64_t* off = obs_offsets(pop);
  int64_t n_obs = off[pop->n_subjects];
  /* observation values / outeqs in prediction order (occasions are sorted like simulate does) */
  double* yv = (double*)malloc(sizeof(double) * (size_t)(n_obs + 1));
  int* oq = (int*)malloc(sizeof(int) * (size_t)(n_obs + 1));
  int64_t* osrc = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_obs + 1)); /* row -> caller's event index */
  {
    scratch_t sc;
    scratch_init(&sc, pop);
    int64_t row = 0;
    for (int64_t oc = 0; oc < pop->n_occasions; oc++) {
      int64_t e0 = pop->occ_ev_off[oc], n = pop->occ_ev_off[oc + 1] - e0;
      for (int64_t i = 0; i < n; i++) {
        sc.ev[i].time = pop->ev_time[e0 + i];
        sc.ev[i].value = pop->ev_value[e0 + i];
        sc.ev[i].kind = pop->ev_kind[e0 + i];
        sc.ev[i].io = pop->ev_io[e0 + i];
        sc.ev[i].src = e0 + i;
      }
      if (!pop->presorted) ev_sort(sc.ev, n);
      for (int64_t i = 0; i < n; i++)
        if (sc.ev[i].kind == PMX_EV_OBSERVATION) {
          yv[row] = sc.ev[i].value;
          oq[row] = sc.ev[i].io;
          osrc[row] = sc.ev[i].src;
          row++;
        }
    }
    scratch_free(&sc);
  }
  int failed = PMX_OK, any_pair_failed = 0;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
  nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
  {
    scratch_t sc;
    scratch_init(&sc, pop);
    int64_t max_rows = 1;
    for (int64_t s = 0; s < pop->n_subjects; s++)
      if (off[s + 1] - off[s] > max_rows) max_rows = off[s + 1] - off[s];
    double* pr = (double*)malloc(sizeof(double) * (size_t)max_rows);
#pragma omp for schedule(dynamic)
    for (int64_t s = 0; s < pop->n_subjects; s++) {
      sc.cov = subject_cov(pop, s);
      for (int64_t p = 0; p < n_support; p++) { /* estimate_log_likelihood_dense, equation/mod.rs:468-477 */
        uint8_t st = 0;
        int r = simulate_pair(model, pop, s, theta + p * model->nparams, pr, 1, &st, &sc);
        double total = 0.0; /* SubjectPredictions::log_likelihood, subject.rs:63-78 */
        for (int64_t k = 0; k < off[s + 1] - off[s] && r == PMX_OK; k++) {
          double y = yv[off[s] + k];
          if (isnan(y)) continue; /* observation is None: contributes 0, prediction.rs:107-111 */
          int q = oq[off[s] + k];
          double sigma;
          if (q >= model->nout) {
            r = PMX_ERR_ERROR_MODEL;
            break;
          }
          if (em[q].kind >= PMX_EM_RES_CONSTANT) { /* ResidualErrorModel::log_likelihood, residual_error.rs:178-191,265-271 */
            double f = pr[k], raw = em[q].scalar;
            if (em[q].kind == PMX_EM_RES_PROPORTIONAL) raw = em[q].scalar * fabs(f);
            if (em[q].kind == PMX_EM_RES_COMBINED) raw = sqrt(em[q].scalar * em[q].scalar + (em[q].c[0] * em[q].c[0]) * (f * f));
            double sg = fmax(raw, sqrt(2.220446049250313e-16));
            double nr = (y - f) / sg;
            total += -0.5 * (log(6.283185307179586) + 2.0 * log(sg) + nr * nr);
            continue;
          }
          pmx_error_model e = em[q]; /* the observation's own ErrorPoly wins, error_model.rs:1051-1054 */
          if (pop->ev_errorpoly && !isnan(pop->ev_errorpoly[osrc[off[s] + k] * 4]))
            for (int c = 0; c < 4; c++) e.c[c] = pop->ev_errorpoly[osrc[off[s] + k] * 4 + c];
          if (pmx_oracle_sigma(&e, y, &sigma) != PMX_OK) {
            r = PMX_ERR_ERROR_MODEL;
            break;
          }
          int cz = pop->ev_censor ? pop->ev_censor[osrc[off[s] + k]] : PMX_CENSOR_NONE;
          double term; /* Prediction::log_likelihood, prediction.rs:113-117 */
          if (cz == PMX_CENSOR_NONE) {
            term = pmx_oracle_lognormpdf(y, pr[k], sigma);
          } else if (pmx_oracle_lognormcdf(y, pr[k], sigma, cz == PMX_CENSOR_ALOQ, &term) != PMX_OK) {
            term = NAN; /* Err(..) in the reference: the pair is flagged non-finite */
          }
          total += term;
        }
        if (st == PMX_PAIR_OK && !isfinite(total)) st = PMX_PAIR_NONFINITE; /* NonFiniteLikelihood, prediction.rs:119-124 */
        if (st == PMX_PAIR_COMPLEX_ROOTS) total = NAN;
        ll[s * ld_ll + p] = total;
        if (status) status[s * n_support + p] = st;
        if (st) {
#pragma omp atomic write
          any_pair_failed = 1;
        }
        if (r != PMX_OK) {
#pragma omp atomic write
          failed = r;
        }
      }
      subject_cov_free(pop, s, sc.cov);
    }
    free(pr);
    scratch_free(&sc);
  }
  free(off);
  free(yv);
  free(oq);
  free(osrc);
  if (failed != PMX_OK) FAIL(failed, "log-likelihood failed with status %d", failed);
  if (any_pair_failed) FAIL(PMX_ERR_PAIR_FAILED, "at least one (subject, support point) pair failed");
  return PMX_OK;
}

int64_t pmx_oracle_sizeof_model_desc(void) { return (int64_t)sizeof(pmx_model_desc); }
int64_t pmx_oracle_sizeof_population_desc(void) { return (int64_t)sizeof(pmx_population_desc); }
