/*
 * pmx.h — C ABI of libpmx_hip.so: MI355X-native batched PK/PD prediction
 * (pharmsol's `Equation::estimate_predictions` hot path, subject x support-point).
 *
 * The reference (LAPKB/pharmsol, Rust) has no FFI on this path; the seam this
 * library replaces is the `Equation` trait surface:
 *   - Equation::estimate_predictions        src/simulator/equation/mod.rs:526-532
 *   - Equation::estimate_predictions_dense  src/simulator/equation/mod.rs:459-465
 *   - the population double loop            src/simulator/likelihood/matrix.rs:79-98
 * A pharmsol maintainer binds these entry points from an `extern "C"` block
 * (INTEGRATION.md shows the Rust stub).  Plain pointers and sizes only.
 *
 * Ownership: the caller owns every buffer it passes in; the library copies what
 * it needs at *_create time and owns the device mirrors behind opaque handles.
 * Handles are immutable after creation and may be shared by host threads
 * (the reference's `Equation: Sync` bound, equation/mod.rs:377).
 * Every function returns a pmx_status (0 = OK); pmx_last_error() returns the
 * message of the last failing call made by the calling thread.
 */
#ifndef PMX_H
#define PMX_H

#if !defined(__HIPCC_RTC__) /* hiprtc translation units bring their own fixed-width typedefs */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define PMX_ABI_VERSION 3

/* ---- limits (fixed-size descriptor arrays) ------------------------------- */
#define PMX_MAX_STATES 8
#define PMX_MAX_INPUTS 8
#define PMX_MAX_OUT 4
#define PMX_MAX_KPARAMS 8
#define PMX_MAX_DERIVED 4
#define PMX_MAX_USER_DERIVED 16 /* values a user `pmx_derive` may write (pmx_model_create_user) */
#define PMX_MAX_FACTORS 2
#define PMX_MAX_PARAMS 16
#define PMX_MAX_COVARIATES 8

/* ---- status codes --------------------------------------------------------- */
/* Call-level errors mirror PharmsolError variants (src/error/mod.rs:13-49). */
typedef enum pmx_status {
  PMX_OK = 0,
  PMX_ERR_INVALID_ARGUMENT = 1,
  PMX_ERR_INPUT_OUT_OF_RANGE = 2,  /* PharmsolError::InputOutOfRange  (error/mod.rs:43; equation/mod.rs:322-327) */
  PMX_ERR_OUTEQ_OUT_OF_RANGE = 3,  /* PharmsolError::OuteqOutOfRange  (error/mod.rs:45) */
  PMX_ERR_UNSUPPORTED = 4,         /* reserved: since ABI 3 no entry point returns it - every model shape the reference
                                      accepts runs on the device (tests/test_formerly_refused_shapes.py) */
  PMX_ERR_NO_DEVICE = 5,           /* no HIP device / kernel image: the library never falls back to a CPU path */
  PMX_ERR_HIP = 6,                 /* a HIP runtime call failed (message has the hipError string) */
  PMX_ERR_OUT_OF_MEMORY = 7,
  PMX_ERR_ERROR_MODEL = 9,         /* ErrorModelError: MissingErrorModel / NegativeSigma / NonFiniteSigma (error_model.rs:1045-1080) */
  PMX_ERR_PAIR_FAILED = 8          /* at least one (subject, support point) pair failed; see the status array.
                                      Mirrors log_likelihood_matrix aborting on the first error (matrix.rs:83,104);
                                      predictions of the healthy pairs are still written. */
} pmx_status;

/* Per-(subject, support point) status byte written beside the predictions. */
enum {
  PMX_PAIR_OK = 0,
  PMX_PAIR_COMPLEX_ROOTS = 1, /* reference panics: two_compartment_models.rs:20-22, three_compartment_models.rs:32-34 */
  PMX_PAIR_NONFINITE = 2,     /* a prediction is NaN/inf (PharmsolError::NonFiniteLikelihood-style guard) */
  PMX_PAIR_BAD_LAG = 3,       /* the support point gives a NaN lag time: its predictions are NaN (the reference panics in
                                 its sort).  A NEGATIVE lag is not an error: the bolus moves earlier, as in the reference
                                 (`if l != 0.0 { time += l }`, src/data/structs.rs:629-634) */
  PMX_PAIR_SOLVER_FAIL = 4    /* adaptive ODE solver: step size underflow (PharmsolError::DiffsolError, error/mod.rs:25) */
};

/* ---- events ---------------------------------------------------------------- */
/* Numeric value == the reference's sort rank at equal times
 * (Observation < Bolus < Infusion, src/data/event.rs:292-304). */
enum { PMX_EV_OBSERVATION = 0, PMX_EV_BOLUS = 1, PMX_EV_INFUSION = 2 };

/*
 * Flattened Data -> Subject -> Occasion -> Event (src/data/structs.rs:352-355,556-560;
 * src/data/event.rs:107-114) as structure-of-arrays with CSR offsets.  Labels are
 * already resolved to dense indices (equation/mod.rs:247-273 does this per
 * (subject, theta) in the reference; here it happens once on the host).
 */
typedef struct pmx_population_desc {
  int64_t n_subjects;
  int64_t n_occasions;            /* total over all subjects */
  int64_t n_events;               /* total over all occasions */
  const int64_t* subj_occ_off;    /* [n_subjects+1]  subject  -> occasions */
  const int64_t* occ_ev_off;      /* [n_occasions+1] occasion -> events */
  const int32_t* occ_index;       /* [n_occasions] Occasion::index (init runs only for index 0,
                                     analytical/mod.rs:417); NULL = position within the subject */
  const double* ev_time;          /* [n_events] */
  const double* ev_value;         /* [n_events] dose amount; observed value for observations (NaN = missing; unused here) */
  const double* ev_duration;      /* [n_events] infusion duration, 0 otherwise */
  const uint8_t* ev_kind;         /* [n_events] PMX_EV_* */
  const uint16_t* ev_io;          /* [n_events] dense input index (doses) / output-equation index (observations) */
  int32_t n_covariates;           /* dense covariate columns in model order */
  int32_t presorted;              /* 0: the library sorts each occasion like Occasion::sort (structs.rs:669-671) */
  const int64_t* cov_knot_off;    /* [n_occasions*n_covariates+1] (occasion, covariate) -> knots; NULL iff n_covariates==0 */
  const double* cov_knot_time;    /* covariate observations (src/data/covariate.rs:189-214) */
  const double* cov_knot_value;
  const uint8_t* cov_fixed;       /* [n_occasions*n_covariates] 1 = carry-forward only; NULL = all interpolated */
  /* log-likelihood only (Observation::errorpoly / ::censoring, src/data/event.rs:575-582); both may be NULL */
  const double* ev_errorpoly;     /* [n_events*4] the observation's own ErrorPoly c0..c3, overriding the error
                                     model's (error_model.rs:1051-1054); c0 = NaN: none */
  const int8_t* ev_censor;        /* [n_events] PMX_CENSOR_* (Censor, event.rs:557-567) */
} pmx_population_desc;

/* Censor: NONE -> lognormpdf, BLOQ -> log CDF, ALOQ -> log survival (prediction.rs:113-117) */
enum { PMX_CENSOR_NONE = 0, PMX_CENSOR_BLOQ = 1, PMX_CENSOR_ALOQ = -1 };

/* ---- models ---------------------------------------------------------------- */
/* EqnKind, src/simulator/equation/mod.rs:580-586 */
enum { PMX_EQ_ODE = 0, PMX_EQ_ANALYTICAL = 1 };

/* AnalyticalKernel, pharmsol-dsl/src/analysis.rs:187-200 (same order). */
enum {
  PMX_K_ONE_COMPARTMENT = 0,                        /* p = [ke]                       one_compartment_models.rs:12-19 */
  PMX_K_ONE_COMPARTMENT_CL = 1,                     /* p = [cl, v]                    one_compartment_cl_models.rs:16-22 */
  PMX_K_ONE_COMPARTMENT_CL_WITH_ABSORPTION = 2,     /* p = [ka, cl, v]                one_compartment_cl_models.rs:38-45 */
  PMX_K_ONE_COMPARTMENT_WITH_ABSORPTION = 3,        /* p = [ka, ke]                   one_compartment_models.rs:32-44 */
  PMX_K_TWO_COMPARTMENTS = 4,                       /* p = [ke, kcp, kpc]             two_compartment_models.rs:14-48 */
  PMX_K_TWO_COMPARTMENTS_CL = 5,                    /* p = [cl, q, vc, vp]            two_compartment_cl_models.rs:16-26 */
  PMX_K_TWO_COMPARTMENTS_CL_WITH_ABSORPTION = 6,    /* p = [ka, cl, q, vc, vp]        two_compartment_cl_models.rs:41-53 */
  PMX_K_TWO_COMPARTMENTS_WITH_ABSORPTION = 7,       /* p = [ke, ka, kcp, kpc]         two_compartment_models.rs:61-112 */
  PMX_K_THREE_COMPARTMENTS = 8,                     /* p = [k10,k12,k13,k21,k31]      three_compartment_models.rs:17-109 */
  PMX_K_THREE_COMPARTMENTS_CL = 9,                  /* p = [cl,q2,q3,vc,v2,v3]        three_compartment_cl_models.rs:16-31 */
  PMX_K_THREE_COMPARTMENTS_CL_WITH_ABSORPTION = 10, /* p = [ka,cl,q2,q3,vc,v2,v3]     three_compartment_cl_models.rs:46-67 */
  PMX_K_THREE_COMPARTMENTS_WITH_ABSORPTION = 11,    /* p = [ka,k10,k12,k13,k21,k31]   three_compartment_models.rs:126-240 */
  PMX_K_ANALYTICAL_COUNT = 12,
  PMX_K_CUSTOM = 100                                /* the user's own propagator `pmx_eq` (pmx_model_create_user) */
};

/* Built-in `diffeq` bodies for the ODE back-end (device functor registry; the
 * reference takes a Rust closure, src/simulator/mod.rs:41).  Route injection
 * follows the `ode!` lowering: dx[dest] += rateiv[i] per infusion route
 * (pharmsol-macros/src/expand/ode.rs:380-406). */
enum {
  PMX_ODE_ONE_CMT_IV = 0,     /* dx0 = -ke x0 + r0                         p=[ke,...]   examples/ode_readme.rs:9-23 */
  PMX_ODE_ONE_CMT_ORAL = 1,   /* dx0 = -ka x0; dx1 = ka x0 - ke x1 + r0    p=[ka,ke,...] */
  PMX_ODE_TWO_CMT_IV = 2,     /* two_compartment_models.rs:131-136 test ODE p=[ke,kcp,kpc,...] */
  PMX_ODE_TWO_CMT_ORAL = 3,   /* two_compartment_models.rs:188-194 test ODE p=[ke,ka,kcp,kpc,...] */
  PMX_ODE_THREE_CMT_IV = 4,   /* three_compartment_models.rs test ODE      p=[k10,k12,k13,k21,k31,...] */
  PMX_ODE_THREE_CMT_ORAL = 5, /* p=[ka,k10,k12,k13,k21,k31,...] */
  PMX_ODE_ONE_CMT_MM = 6,     /* nonlinear: dx0 = -vmax*(x0/v)/(km + x0/v) + r0   p=[vmax,km,v] */
  PMX_ODE_MODEL_COUNT = 7,
  PMX_ODE_CUSTOM = 100        /* user source compiled at run time: pmx_model_create_custom */
};

/* Where a value is read from. */
enum { PMX_SRC_NONE = 0, PMX_SRC_PRIMARY = 1 /* theta[index] */, PMX_SRC_DERIVED = 2 /* derived[index] */ };

/* One multiplicative covariate factor of a derived parameter. */
enum { PMX_F_NONE = 0, PMX_F_POW = 1 /* (cov/ref)^coef */, PMX_F_LIN = 2 /* 1 + coef*(cov-ref) */ };
typedef struct pmx_factor {
  int32_t op;
  int32_t cov;  /* dense covariate index */
  double ref;
  double coef;
} pmx_factor;

/* derived[d] = ((theta[src_param] * f[0]) * f[1])  — the `derive:` block of
 * analytical!/ode! restricted to allometric form
 * (examples/analytical_readme.rs:18-20: ke = ke0 * (wt/70)^0.75). */
typedef struct pmx_derived {
  int32_t src_param;
  int32_t n_factors;
  pmx_factor f[PMX_MAX_FACTORS];
} pmx_derived;

/* Kernel-order parameter j <- theta[index] | derived[index]
 * (the macro's projection wrapper, pharmsol-macros/src/expand/analytical.rs:208-294). */
typedef struct pmx_bind {
  int32_t src;
  int32_t index;
} pmx_bind;

/* y[o] = x[state] / vol,   vol = theta[vol_index] | derived[vol_index] | 1
 * (every reference example/bench: y = x[central]/v, e.g. examples/analytical_vs_ode.rs:82-84). */
typedef struct pmx_out {
  int32_t state;
  int32_t vol_src;
  int32_t vol_index;
  int32_t reserved;
} pmx_out;

/* Which time the analytical `derive` block sees:
 *  SEGMENT_DT      = the macro lowering: derive(p, dt, cov) — covariates at the
 *                    segment LENGTH (expand/analytical.rs:254,286; analytical/mod.rs:363-364)
 *  SEGMENT_END_ABS = the DSL runtime: derived refreshed at absolute next_t (src/dsl/native.rs:1907-1916) */
enum { PMX_COV_TIME_SEGMENT_DT = 0, PMX_COV_TIME_SEGMENT_END_ABS = 1 };

typedef struct pmx_model_desc {
  int32_t eq_kind;       /* PMX_EQ_ANALYTICAL | PMX_EQ_ODE */
  int32_t kernel;        /* PMX_K_* or PMX_ODE_* */
  int32_t nstates;       /* Analytical::with_nstates  analytical/mod.rs:120 */
  int32_t ndrugs;        /* Analytical::with_ndrugs   analytical/mod.rs:127 */
  int32_t nout;          /* Analytical::with_nout     analytical/mod.rs:134 */
  int32_t nparams;       /* length of one support point (Parameters::as_slice, parameters.rs:94) */
  int32_t n_covariates;
  int32_t n_derived;
  pmx_derived derived[PMX_MAX_DERIVED];
  int32_t n_bind;        /* 0 = identity (kernel param j = theta[j]) */
  pmx_bind bind[PMX_MAX_KPARAMS];
  pmx_out out[PMX_MAX_OUT];
  int32_t cov_time_mode; /* PMX_COV_TIME_* (analytical only; ODE covariates use absolute t, expand/ode.rs:150) */
  int32_t pmetrics_indexing; /* 1 = pm_* wrapper: state/rateiv slot 0 is a dead pad (analytical/mod.rs:62-90) */
  /* init: x[i] = theta[init_param[i]] for occasion index 0 only (analytical/mod.rs:409-426); -1 = 0.0 */
  int32_t init_param[PMX_MAX_STATES];
  /* lag[input] = theta[lag_param[input]], fa[input] = theta[fa_param[input]]; -1 = absent
   * (Occasion::process_events, src/data/structs.rs:611-690). */
  int32_t lag_param[PMX_MAX_INPUTS];
  int32_t fa_param[PMX_MAX_INPUTS];
  /* ODE route destinations (`ode!` routes, e.g. bolus(oral) -> gut, infusion(iv) -> central;
   * pharmsol-macros/src/expand/ode.rs:380-406): the state that receives bolus input i /
   * infusion input i.  -1 = default (bolus: state i; infusion: the model's central state).
   * Analytical ignores these: a bolus goes to x[input] and the closed forms read rateiv[0]
   * only (equation/mod.rs:328; one_compartment_models.rs:16). */
  int32_t bolus_dest[PMX_MAX_INPUTS];
  int32_t infusion_dest[PMX_MAX_INPUTS];
  double rk4_h_max;     /* ODE: fixed-step RK4, h = dt/ceil(dt/h_max) per constant-rate piece; adaptive: largest step */
  /* ODE solver (`ODE::with_solver` / `with_tolerances`, ode/mod.rs:134-166; the reference's diffsol solvers are
   * replaced: SURVEY.md §8 a23).  PMX_SOLVER_RK4 = the fixed-step default.  PMX_SOLVER_DOPRI5 = embedded
   * Dormand-Prince 5(4) with step-size control per lane: err = rms(e_i / (atol + rtol max(|x_i|, |x'_i|))) <= 1.
   * PMX_SOLVER_ROS2 = the stiff option (the role of the reference's default OdeSolver::Bdf / Sdirk, ode/mod.rs:60-77):
   * ROS2, a second-order L-stable Rosenbrock method with the same per-lane step control (error estimate from its embedded
   * first-order solution), Jacobian by forward differences, NS x NS elimination in registers (csrc/pmx_ode.hpp ros2_try). */
  int32_t ode_solver;
  int32_t reserved_;
  double ode_rtol, ode_atol;
} pmx_model_desc;

enum { PMX_SOLVER_RK4 = 0, PMX_SOLVER_DOPRI5 = 1, PMX_SOLVER_ROS2 = 2 /* stiff: L-stable Rosenbrock, adaptive */ };

typedef struct pmx_population pmx_population; /* opaque */
typedef struct pmx_model pmx_model;           /* opaque */

/* Layout of the prediction tensor written by pmx_predict*:
 *   pred[(obs_row) * ld_pred + p],  obs_row = running index of the observation over
 *   all subjects in event order (SubjectPredictions::flat_predictions order,
 *   likelihood/subject.rs:145-148), p = support point.  ld_pred >= n_support. */

/* ---- entry points ---------------------------------------------------------- */

int32_t pmx_abi_version(void);

/* sizeof() of the descriptor structs as the library was built (binding layout check). */
int64_t pmx_sizeof_model_desc(void);
int64_t pmx_sizeof_population_desc(void);
/* sizeof() of any public struct of this header by name ("pmx_model_desc", "pmx_factor", ...); -1 for an unknown name.
 * Generated bindings (bindings/rust/pmx_sys.rs, pharmsol_amd/_abi.py) are checked against it. */
int64_t pmx_sizeof_struct(const char* name);

/* Number of visible HIP devices (0 when none). */
int32_t pmx_device_count(void);

/* Flatten + upload a population.  Replaces Data/Subject construction + the
 * per-(subject,theta) Occasion::clone/sort of equation/mod.rs:247-273.
 * `device` = HIP device ordinal the mirrors live on. */
int32_t pmx_population_create(const pmx_population_desc* desc, int32_t device, pmx_population** out);
void pmx_population_destroy(pmx_population* pop);

/* ---- sharding across GPUs ------------------------------------------------------------------------
 * The reference's population loop (likelihood/matrix.rs:79-98: rayon over subjects, serial over support points) has no
 * cross-iteration state: the path shards by SUBJECT, theta replicated, with no data-path collective.  One process (or
 * host thread) per GPU:
 *
 *   pmx_shard_bounds(desc, n, bounds);                         // the same answer on every rank
 *   pmx_population_create_shard(desc, bounds[r], bounds[r + 1], device_r, &pop_r);
 *   pmx_predict_device / pmx_loglik_device(model, pop_r, ...)  // rank r's rows: [rows[r], rows[r + 1]) of the full matrix
 *
 * bounds[n_shards + 1]: contiguous subject ranges holding equal shares of the EVENTS (= subject-event-steps per
 * support point), so ragged populations balance by work, not by head count. */
int32_t pmx_shard_bounds(const pmx_population_desc* desc, int32_t n_shards, int64_t* bounds);
/* rows[n_shards + 1]: first prediction row of every shard (rows[n_shards] = all observations). */
int32_t pmx_shard_rows(const pmx_population_desc* desc, int32_t n_shards, const int64_t* bounds, int64_t* rows);
/* pmx_population_create over subjects [subject_begin, subject_end) of `desc`, without the caller re-basing its arrays. */
int32_t pmx_population_create_shard(const pmx_population_desc* desc, int64_t subject_begin, int64_t subject_end,
                                    int32_t device, pmx_population** out);

/* The one optional exchange: every rank's prediction rows on every device (a caller that post-processes the whole
 * matrix on each GPU; NPAG-style callers consume rows where they were produced and never need it).  RCCL over xGMI,
 * one communicator rank per GPU:
 *
 *   rank 0: pmx_comm_unique_id(id) -> the caller ships the 128 bytes to the other ranks (MPI, a socket, torch.distributed ...)
 *   every rank: pmx_comm_create(id, n_ranks, rank, device, &comm)          // collective: ncclCommInitRank
 *   every rank: d_full = [rows[n_ranks] x ld] doubles; pmx_predict_device(..., d_full + rows[rank] * ld, ld, ...)
 *   every rank: pmx_allgather_predictions(comm, d_full, rows, ld, stream)  // in place, stream-ordered, not synchronised
 *
 * Equal row blocks travel as ONE in-place ncclAllGather; unequal ones (events-balanced shards) as one grouped set of
 * in-place broadcasts, one per owner - no padding, no staging copy.  librccl.so is opened on first use
 * (PMX_ERR_NO_DEVICE when it cannot be). */
#define PMX_COMM_ID_BYTES 128
typedef struct pmx_comm pmx_comm; /* opaque */
int32_t pmx_comm_unique_id(uint8_t* id /* [PMX_COMM_ID_BYTES] */);
int32_t pmx_comm_create(const uint8_t* id, int32_t n_ranks, int32_t rank, int32_t device, pmx_comm** out);
void pmx_comm_destroy(pmx_comm* comm);
int32_t pmx_comm_size(const pmx_comm* comm);
int32_t pmx_comm_rank(const pmx_comm* comm);
int32_t pmx_allgather_predictions(pmx_comm* comm, double* d_full, const int64_t* rows, int64_t ld, void* stream);

/* Sizes a caller needs to allocate outputs. */
int64_t pmx_population_n_subjects(const pmx_population* pop);
int64_t pmx_population_n_observations(const pmx_population* pop); /* rows of pred */
int64_t pmx_population_n_events(const pmx_population* pop);       /* subject-event-steps per support point */
int32_t pmx_population_device(const pmx_population* pop);         /* the device ordinal it was created on (-1: NULL) */
/* obs_off[n_subjects+1]: first prediction row of each subject. */
int32_t pmx_population_observation_offsets(const pmx_population* pop, int64_t* obs_off);
/* Per prediction row: time / outeq / subject of the observation, in prediction
 * order (what Observation::to_prediction re-attaches, event.rs:698-711). Any pointer may be NULL. */
int32_t pmx_population_observation_info(const pmx_population* pop, double* time, int32_t* outeq, int64_t* subject);

/* Replaces Analytical::new(...).with_nstates()... / ODE::new(...) (analytical/mod.rs:102-152, ode/mod.rs:115-166). */
int32_t pmx_model_create(const pmx_model_desc* desc, pmx_model** out);
void pmx_model_destroy(pmx_model* model);

/* Population prediction: every subject x every support point.
 *   theta  [n_support x nparams] row-major (ParameterOrder::matrix rows, matrix.rs:62-65)
 *   pred   [n_observations x ld_pred]
 *   status [n_subjects x n_support] bytes (PMX_PAIR_*), may be NULL
 * Host-pointer form: copies theta in and pred/status out (PCIe inclusive). */
int32_t pmx_predict(const pmx_model* model, const pmx_population* pop, const double* theta, int64_t n_support,
                    double* pred, int64_t ld_pred, uint8_t* status);

/* Device-pointer form: theta/pred/status are device pointers on the population's
 * device; the launch is enqueued on `stream` (a hipStream_t, NULL = default
 * stream) and NOT synchronised.  This is the form bench.py times. */
int32_t pmx_predict_device(const pmx_model* model, const pmx_population* pop, const double* d_theta,
                           int64_t n_support, double* d_pred, int64_t ld_pred, uint8_t* d_status, void* stream);

/* Average device time (ms, HIP events on `stream`) of `reps` prediction passes into `d_pred`, after one untimed pass;
 * synchronises `stream`.  For choosing among candidate output buffers: on MI355X the row-strided write stream runs at one
 * of two speeds depending on the allocation it lands in (DESIGN.md §5 "Where the matrix lives"); a caller that reuses
 * one buffer across passes allocates a few, times each with this call and keeps the fastest. */
int32_t pmx_time_predict_device(const pmx_model* model, const pmx_population* pop, const double* d_theta,
                                int64_t n_support, double* d_pred, int64_t ld_pred, int32_t reps, void* stream,
                                double* ms_per_pass);

/* A prediction matrix [n_observations x n_support] (ld = n_support) placed where the kernel writes it fastest: an
 * arena of up to `search_bytes` (at least the matrix; 0 = just the matrix) is mapped window by window from separately
 * allocated physical chunks and the real kernel is timed into each; the search stops inside the first plateau of the
 * fast kind (three neighbouring windows at >= 6.4 TB/s, or 15 % faster than the slowest seen) or at `search_bytes`; the
 * chunks under the chosen window are kept and all others are returned to the device.  A NEGATIVE `search_bytes` asks
 * for the exhaustive form: every window of an arena of |search_bytes| is timed and the best one kept (about 1 s for
 * 96 GiB; typically 3-4 % faster than the first plateau).  A plain allocation is timed as a candidate too (boxes exist
 * whose arenas hold no fast window) and returned when it beats the chosen window.  *ms_per_pass receives the pass
 * time measured in the chosen window.  The arena is allocated on the population's device (the caller's current device
 * is restored on return).  Free with pmx_prediction_buffer_destroy. */
int32_t pmx_prediction_buffer_create(const pmx_model* model, const pmx_population* pop, const double* d_theta,
                                     int64_t n_support, int64_t search_bytes, void* stream, double** d_pred,
                                     double* ms_per_pass);
/* ... with rows `ld` doubles apart (ld >= n_support; pmx_recommended_ld): [n_observations x ld] doubles are placed. */
int32_t pmx_prediction_buffer_create_pitched(const pmx_model* model, const pmx_population* pop, const double* d_theta,
                                     int64_t n_support, int64_t ld, int64_t search_bytes, void* stream, double** d_pred,
                                     double* ms_per_pass);
void pmx_prediction_buffer_destroy(double* d_pred);

/* Prediction::state (likelihood/prediction.rs:18-27, a19): the reference records the full state vector beside every
 * prediction.  This entry point writes the amount in model state `state` at every observation time, for every
 * support point, in the layout of pmx_predict_device (one call per state of interest; same kernels, the output
 * equations replaced by y = x[state]). */
int32_t pmx_predict_state_device(const pmx_model* model, const pmx_population* pop, const double* d_theta,
                                 int64_t n_support, int32_t state, double* d_out, int64_t ld_out, uint8_t* d_status,
                                 void* stream);

/* "Batch" form (log_likelihood_batch shape, likelihood/mod.rs:119-177): subject s
 * is simulated with its own row theta[s] only.  pred is [n_observations] (ld 1),
 * status [n_subjects]. */
int32_t pmx_predict_batch(const pmx_model* model, const pmx_population* pop, const double* theta, double* pred,
                          uint8_t* status);
int32_t pmx_predict_batch_device(const pmx_model* model, const pmx_population* pop, const double* d_theta,
                                 double* d_pred, uint8_t* d_status, void* stream);

/* ---- fused log-likelihood ----------------------------------------------
 * log_likelihood_matrix(eq, &Data, &theta, &AssayErrorModels, progress) (likelihood/matrix.rs:52-106):
 *   ll[s][p] = sum over the subject's observations that carry a value of
 *              lognormpdf(obs, pred, sigma) = -0.5*ln(2 pi) - ln(sigma) - (obs-pred)^2 / (2 sigma^2)
 *              (likelihood/distributions.rs:31-34; SubjectPredictions::log_likelihood, subject.rs:63-78;
 *               missing observations contribute 0, prediction.rs:105-111)
 * sigma comes from the OBSERVATION through the assay error polynomial of its output equation
 * (AssayErrorModel::sigma, src/data/error_model.rs:1045-1080):
 *   alpha = c0 + c1 y + c2 y^2 + c3 y^3;  additive: sqrt(alpha^2 + lambda^2);  proportional: gamma * alpha
 * It does not depend on the support point, so one small device kernel derives it once per observation whenever the
 * error models change (stream-ordered before the likelihood kernel); the main kernel folds each prediction into its
 * subject's sum instead of storing it (output S x P doubles instead of S x O x P).  Censored observations
 * (pmx_population_desc::ev_censor) take the log CDF / log survival function (prediction.rs:113-117), an observation's
 * own polynomial (ev_errorpoly) overrides the model's. */
enum {
  PMX_EM_NONE = 0,
  PMX_EM_ADDITIVE = 1,      /* AssayErrorModel::additive:     sigma = sqrt(alpha(obs)^2 + lambda^2) */
  PMX_EM_PROPORTIONAL = 2,  /* AssayErrorModel::proportional: sigma = gamma * alpha(obs) */
  /* ResidualErrorModel (src/data/residual_error.rs:69-136): sigma from the PREDICTION f, floored at sqrt(f64::EPSILON)
   * (:178-191); term = -0.5 (ln 2 pi + 2 ln sigma + ((y - f) / sigma)^2) (:265-271); no censoring, no per-observation
   * polynomial.  What log_likelihood_batch takes (likelihood/mod.rs:119-124).  scalar = a | b | a | sigma, c[0] = b
   * of the combined model. */
  PMX_EM_RES_CONSTANT = 3,     /* sigma = a */
  PMX_EM_RES_PROPORTIONAL = 4, /* sigma = b |f| */
  PMX_EM_RES_COMBINED = 5,     /* sigma = sqrt(a^2 + b^2 f^2) */
  PMX_EM_RES_EXPONENTIAL = 6   /* sigma = sigma_exp */
};
typedef struct pmx_error_model {
  int32_t kind;   /* PMX_EM_* ; NONE + an observation on that outeq = MissingErrorModel error */
  int32_t reserved;
  double c[4];    /* ErrorPoly c0..c3 (assay models); c[0] = b of PMX_EM_RES_COMBINED */
  double scalar;  /* lambda (additive) | gamma (proportional) | a / b / sigma of the residual models */
} pmx_error_model;

/* ---- user ODE models compiled at run time (hiprtc) ---------------------------
 * The reference takes `diffeq` / `out` / `init` as Rust closures (ODE::new, ode/mod.rs:115-132) or as DSL text it
 * JIT-compiles for the CPU (src/dsl/jit.rs).  Here `source` is C/HIP text defining device functions with the
 * reference's compiled-kernel argument order (src/dsl/native.rs:45-53: t, states, params, covariates, routes,
 * derived, out):
 *
 *   PMX_DEVICE void pmx_dynamics(double t, const double* x, const double* p, const double* cov,
 *                                const double* rateiv, const double* derived, double* dx);   // dx pre-zeroed
 *   PMX_DEVICE void pmx_outputs (double t, const double* x, const double* p, const double* cov,
 *                                const double* rateiv, const double* derived, double* y);    // y pre-zeroed
 *   PMX_DEVICE void pmx_init    (double t, const double* x0, const double* p, const double* cov,
 *                                const double* rateiv, const double* derived, double* x);    // iff has_init
 *
 * `rateiv[i]` is the active infusion rate of input i (the body adds it where the route goes, like a hand-written
 * closure), `p` the support point in model order, `cov[c]` covariate c of the subject's current occasion
 * interpolated at `t` (every stage of every step, like `fetch_cov!(cov, t, ..)`; NULL when n_covariates = 0);
 * `derived` is NULL (compute derived values in the body).  desc: eq_kind = PMX_EQ_ODE, kernel = PMX_ODE_CUSTOM,
 * nstates/ndrugs/nout/nparams/n_covariates, rk4_h_max, ode_solver, lag_param/fa_param/bolus_dest as for built-in
 * models; `out[]` is ignored.  The source is
 * compiled for gfx950 at creation time (no device needed; PMX_ERR_INVALID_ARGUMENT + the compiler log in
 * pmx_last_error() on a compile error) and runs through the same walkers, RK4 stepper, lag/fa handling and fused
 * log-likelihood as the built-in bodies. */
int32_t pmx_model_create_custom(const pmx_model_desc* desc, const char* source, int32_t has_init, pmx_model** out);

/* ---- user closures for either back-end --------------------------------------------------------------------------
 * `Analytical::new(eq, seq_eq, lag, fa, init, out)` and `ODE::new(diffeq, lag, fa, init, out)` take arbitrary
 * functions of (theta, t, covariates) (src/simulator/mod.rs:41-197; analytical/mod.rs:102-118; ode/mod.rs:115-132).
 * `source` defines the ones named in `functions` (PMX_FN_* bits), all with the argument order of the reference's
 * compiled kernels (src/dsl/native.rs:45-53; symbol roles src/dsl/compiled_backend_abi.rs:13-27):
 *
 *   PMX_DEVICE void pmx_<role>(double t, const double* x, const double* p, const double* cov,
 *                              const double* rateiv, const double* derived, double* out);
 *
 *   role                     t                        out (pre-set)                    reference closure
 *   derive                   see below                derived[n_derived] (zeros)       `derive:` block of analytical!/ode!
 *   route_lag                the bolus' recorded time lag[ndrugs] (zeros)              Lag,  structs.rs:611-643
 *   route_bioavailability    its time AFTER the lag   fa[ndrugs] (ones)                Fa,   structs.rs:645-666
 *   init                     0.0                      x[nstates] (zeros)               Init, analytical/mod.rs:409-426
 *   outputs                  the observation time     y[nout] (zeros)                  Out,  analytical/mod.rs:373-407
 *   seq_eq  (analytical)     the sub-segment's end    the solve's parameter vector,    SecEq, analytical/mod.rs:331,360
 *                                                     modified in place (p = theta)
 *   eq      (analytical,     the sub-segment LENGTH   x_next[nstates] (copy of x)      AnalyticalEq, analytical/mod.rs:363-364
 *            kernel = PMX_K_CUSTOM)                   p = the solve's vector, rateiv[ndrugs]
 *   dynamics (ODE)           stage time               dx[nstates] (zeros)              DiffEq, ode/mod.rs:115-132
 *   dynamics_bolus (ODE)     stage time / bolus time  dx[nstates] (zeros)              DiffEq with its `bolus` argument
 *
 * ODE models.  PMX_FN_DYNAMICS | PMX_FN_OUTPUTS (| PMX_FN_INIT) alone == pmx_model_create_custom (theta-indexed lag / fa
 * through lag_param / fa_param, the fast state-machine kernels).  Adding PMX_FN_ROUTE_LAG / PMX_FN_ROUTE_BIOAVAILABILITY /
 * PMX_FN_DERIVE selects the general ODE walker (csrc/pmx_ode_user.hpp): `lag` and `fa` are closures of (theta, t, cov)
 * evaluated per bolus on the device at the reference's times, every lagged bolus is re-sorted per lane
 * (Occasion::process_events, structs.rs:611-690), `derived` = pmx_derive at the stage / event time.  The reference's
 * DiffEq also receives the dose being given: `diffeq(x, p, t, dx, bolus, rateiv, cov)` and a bolus event adds
 * f(x, bolus) - f(x, 0) (both with rateiv = 0, at the bolus' time after the lag shift) to the state (ode/mod.rs:659-686).
 * A source that defines
 *
 *   PMX_DEVICE void pmx_dynamics_bolus(double t, const double* x, const double* p, const double* cov,
 *                                      const double* rateiv, const double* bolus, const double* derived, double* dx);
 *
 * (PMX_FN_DYNAMICS_BOLUS instead of PMX_FN_DYNAMICS) gets exactly that: bolus[ndrugs] is zero during integration and
 * holds the (fa-scaled) amount at its input for the jump.  With plain pmx_dynamics the jump is `amount` into
 * bolus_dest[input] (what the jump rule gives for every `ode!` model, expand/ode.rs:394-399).
 * The solver clock of an occasion starts at the earliest RECORDED event time (Occasion::initial_time, structs.rs:782-793,
 * ode/mod.rs:348) and only moves forward to the time of the NEXT event (ode/mod.rs:719-721): the occasion's first event
 * after the lag rewrite is applied at that clock without integration - a lagged bolus that is the first event of its
 * occasion acts from the clock's start, exactly as in the reference.
 *
 * `cov[c]` = covariate c of the subject's current occasion interpolated AT THAT t on the device (fetch_cov!(cov, t, ..));
 * `derived` = pmx_derive evaluated at the same t first, the way every macro-lowered closure starts
 * (pharmsol-macros/src/expand/analytical.rs:333-420), NULL without PMX_FN_DERIVE; pointers that have no meaning for a
 * role are NULL.  Analytical models: `kernel` is a built-in structure (its parameters bound through desc->bind[] to
 * theta / derived values; its derive runs at the segment length dt or, PMX_COV_TIME_SEGMENT_END_ABS, at the absolute
 * segment end) or PMX_K_CUSTOM + PMX_FN_EQ.  A closure left out of `functions` falls back to the descriptor's closed
 * form (lag_param / fa_param / init_param / out[]); desc->derived[] is ignored, n_derived <= PMX_MAX_USER_DERIVED.
 * The reference fixture tests/analytical_macro_lowering.rs:225-260 (covariate-dependent lag, clamped fa, init and
 * volume) is written this way in tests/test_user_analytical.py. */
enum {
  PMX_FN_DYNAMICS = 1,
  PMX_FN_OUTPUTS = 2,
  PMX_FN_INIT = 4,
  PMX_FN_DERIVE = 8,
  PMX_FN_ROUTE_LAG = 16,
  PMX_FN_ROUTE_BIOAVAILABILITY = 32,
  PMX_FN_SEQ_EQ = 64,
  PMX_FN_EQ = 128,
  PMX_FN_DYNAMICS_BOLUS = 256
};
int32_t pmx_model_create_user(const pmx_model_desc* desc, const char* source, uint32_t functions, pmx_model** out);
/* The translation unit handed to hiprtc for such a model; free with pmx_free_text. */
int32_t pmx_debug_jit_source_user(const pmx_model_desc* desc, const char* source, uint32_t functions, char** out_text);
/* The translation unit that was (or would be) handed to hiprtc for this source; free with pmx_free_text. */
int32_t pmx_debug_jit_source(const pmx_model_desc* desc, const char* source, int32_t has_init, char** out_text);
void pmx_free_text(char* text);

/*   em   [model.nout] error model per output equation
 *   ll   [n_subjects x ld_ll], ll[s*ld_ll + p]  (the reference's Array2 (n_subjects, n_support); it stores
 *        that matrix column-major, matrix.rs:60 — same logical matrix, support point fastest here)
 *   status [n_subjects x n_support] as for pmx_predict; a non-finite sum sets PMX_PAIR_NONFINITE
 *        (PharmsolError::NonFiniteLikelihood, prediction.rs:119-124)
 * The sigma terms are derived from `em` on the device, stream-ordered before the kernel (changing the error model
 * between calls costs two small kernels, no host pass and no upload).  MissingErrorModel is reported by both
 * forms (PMX_ERR_ERROR_MODEL).  An invalid sigma (NegativeSigma / NonFiniteSigma, error_model.rs:1073-1077) is
 * found on the device: pmx_loglik returns PMX_ERR_ERROR_MODEL; pmx_loglik_device cannot fail after the fact, the
 * affected subjects' rows come back NaN with PMX_PAIR_NONFINITE. */
int32_t pmx_loglik(const pmx_model* model, const pmx_population* pop, const pmx_error_model* em,
                   const double* theta, int64_t n_support, double* ll, int64_t ld_ll, uint8_t* status);
int32_t pmx_loglik_device(const pmx_model* model, const pmx_population* pop, const pmx_error_model* em,
                          const double* d_theta, int64_t n_support, double* d_ll, int64_t ld_ll,
                          uint8_t* d_status, void* stream);

/* log_likelihood_batch(eq, &data, &parameters, &error_models) (likelihood/mod.rs:119-177): subject s under ITS OWN
 * parameter row theta[s] (n_subjects rows); ll[n_subjects], status[n_subjects].  Like the reference, a subject whose
 * simulation or likelihood fails does not fail the call: its entry is -inf (`Err(_) => f64::NEG_INFINITY`,
 * likelihood/mod.rs:137-140) and its status byte says why.  That includes a subject with an observation on an output
 * that has no error model (em[q].kind = PMX_EM_NONE): ResidualErrorModels::total_log_likelihood gives it -inf and the
 * call succeeds (residual_error.rs:413-425) - here its rows poison its sum (PMX_PAIR_NONFINITE), where pmx_loglik fails
 * the whole call with PMX_ERR_ERROR_MODEL like log_likelihood_matrix.  The device form leaves NaN in the failed entries
 * (nothing can be rewritten after the fact on a stream); map status != PMX_PAIR_OK to -inf on the caller's side. */
int32_t pmx_loglik_batch(const pmx_model* model, const pmx_population* pop, const pmx_error_model* em,
                         const double* theta, double* ll, uint8_t* status);
int32_t pmx_loglik_batch_device(const pmx_model* model, const pmx_population* pop, const pmx_error_model* em,
                                const double* d_theta, double* d_ll, uint8_t* d_status, void* stream);

/* Row pitch (ld_pred, in doubles) at which the prediction kernels write fastest: n_support rounded up to a multiple of 16,
 * so every row starts on a 128-byte boundary.  Measured on C3 (profiles/r03/ld_by_allocation.txt): 1000 -> 1008 doubles
 * gives 6.45 -> 6.8 TB/s in fast allocations and 5.3 -> 5.5 in slow ones; a pitch that is a multiple of 8 but not 16
 * doubles is slower than the dense one.  The padding doubles are never written. */
int64_t pmx_recommended_ld(int64_t n_support);

/* The device's write ceiling as measured, for roofline reports: average rate (GB/s, HIP events on `stream`) of `reps`
 * linear streaming fills of d_buf[0 .. n_doubles), after one untimed fill - the best of four store shapes (16 bytes per
 * lane non-temporal / plain in a grid-stride loop, the prediction kernels' own 512 contiguous bytes per wave, and one
 * 16-byte store per lane with no loop - the fastest on the boxes measured, tools/experiments/fill_probe.hip).  The buffer's contents are
 * overwritten with zeros.  bench.py prints it as roofline.attainable beside the 8 TB/s datasheet peak. */
int32_t pmx_measure_write_ceiling(double* d_buf, int64_t n_doubles, int32_t reps, void* stream, double* gb_per_s);

/* Page-locked host memory for the outputs of the HOST-pointer entry points (pmx_predict, pmx_loglik, ...).  Those entry
 * points keep their device buffers, streams and staging areas on the population handle between calls; an output array
 * in page-locked memory (this allocator, hipHostMalloc or hipHostRegister) is filled by one DMA at link rate, a
 * pageable one goes through pinned bounce buffers (the DMA overlapping the CPU copy; about a quarter of the rate). */
int32_t pmx_host_alloc(int64_t bytes, void** out);
void pmx_host_free(void* p);

/* Name of the device kernel family the last pmx_predict* call on this thread launched
 * (for matching rocprofv3 rows). */
const char* pmx_last_kernel_name(void);

/* Message of the last failing call made by this thread ("" if none). */
const char* pmx_last_error(void);

/* The library's developer switches (PMX_DISABLE_CLASSING, PMX_DISABLE_LADDER, PMX_TUNE_*) are read
 * from the environment ONCE, at the first call that needs them.  This re-reads them (tuning scripts, tests). */
void pmx_debug_reload_env(void);

/* ---- host-side introspection (needs no device) ------------------------------ */
/* The flattened op stream the device walks for (population, model): what the
 * host-side population compiler (csrc/pmx_compile.cpp) produces in place of the
 * reference's per-(subject, theta) Occasion::process_events + Analytical::solve
 * splitting.  Lets the CPU test-suite check that logic without a GPU. */
enum { PMX_OP_RESET = 0, PMX_OP_BOLUS = 1, PMX_OP_OBS = 2, PMX_OP_PROP = 3 };
typedef struct pmx_op_stream_view {
  int64_t n_subjects, n_ops;
  int32_t n_cov, n_rate;
  int32_t max_input_used, max_outeq;
  const int64_t* subj_op_off; /* [n_subjects+1] */
  const uint32_t* op_meta;    /* kind | io<<8 */
  const double* op_a;         /* BOLUS amount | OBS time | PROP dt */
  const double* op_b;         /* PROP rateiv[0] (analytical) / RK4 h (ODE) */
  const int32_t* op_n;        /* ODE: RK4 steps per PROP (NULL for analytical) */
  const double* op_rate;      /* ODE: [n_ops*n_rate] (NULL for analytical) */
  const double* op_cov;       /* [n_ops*n_cov] (NULL if n_cov == 0) */
  const int32_t* subj_order;  /* [n_subjects] lane order of the PAIR kernels */
  void* owner;                /* library-owned storage; release with pmx_debug_free */
} pmx_op_stream_view;
int32_t pmx_debug_compile(const pmx_population_desc* pop, const pmx_model_desc* model, pmx_op_stream_view* out);
void pmx_debug_free(pmx_op_stream_view* view);

/* Host-side introspection of the class planner (no GPU): how the analytical GRID kernels would split this
 * population for this model.  counts[0] = chunks of exact classes (members share the whole program, step lengths
 * included), counts[1] = chunks of loose classes (same op kinds / inputs / outputs in the same order, own step
 * lengths), counts[2] = subjects in either kind of class, counts[3] = subjects left to the generic walker,
 * counts[4] = members per chunk (0: the model is not classed at all - covariates, lag, bioavailability, ODE). */
int32_t pmx_debug_class_plan(const pmx_population_desc* pop, const pmx_model_desc* model, int64_t* counts);

#ifdef __cplusplus
}
#endif
#endif /* PMX_H */
