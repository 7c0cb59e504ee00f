// pmx_compile.hpp — host-side "population compiler".
//
// The reference re-derives, for EVERY (subject, support point), the processed
// event list (clone + label parse + sort: equation/mod.rs:247-273, structs.rs:669-690)
// and, inside every `solve`, the infusion sub-segments (analytical/mod.rs:313-357).
// None of that depends on the support point unless the model has lag/fa, so here
// it is done once per subject on the host and flattened into a linear OP STREAM
// that the device walks:
//
//   RESET  start of an occasion: x = 0 (+ init for occasion index 0)   analytical/mod.rs:409-426
//   BOLUS  x[input] += amount                                          equation/mod.rs:313-329
//   OBS    pred[row] = out(x, theta, t_obs)                            analytical/mod.rs:373-407
//   PROP   x = eq(x, theta, dt, rateiv)   one constant-rate sub-segment analytical/mod.rs:334-367
//
// Infusion events need no op of their own: their effect is the rate carried by
// the PROP ops they cover.
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "pmx_devtypes.hpp"

namespace pmx {

// enum OpKind { OP_RESET, OP_BOLUS, OP_OBS, OP_PROP }: pmx_devtypes.hpp

// op_meta layout: bits 0..7 kind, 8..23 io (input / outeq / reset: 1 = run init), 24..31 unused
inline uint32_t make_meta(uint32_t kind, uint32_t io) { return kind | (io << 8); }

// Sorted, validated copy of the caller's events (model independent).
struct HostPopulation {
  int64_t n_subjects = 0, n_occasions = 0, n_events = 0, n_obs = 0;
  int32_t n_cov = 0;
  std::vector<int64_t> subj_occ_off, occ_ev_off;
  std::vector<int32_t> occ_index;
  std::vector<double> ev_time, ev_value, ev_dur;
  std::vector<uint8_t> ev_kind;
  std::vector<uint16_t> ev_io;
  std::vector<int64_t> subj_obs_off;  // [S+1]
  // observation bookkeeping in prediction order
  std::vector<double> obs_time;
  std::vector<double> obs_value;   // observed value (NaN = missing), for the log-likelihood
  std::vector<int32_t> obs_outeq;
  std::vector<double> obs_errorpoly;  // [n_obs*4] the observation's own ErrorPoly (c0 = NaN: none); empty = none anywhere
  std::vector<int8_t> obs_censor;     // [n_obs] PMX_CENSOR_*; empty = none anywhere
  std::vector<int64_t> obs_subject;
  int32_t max_outeq = -1;  // over all observations (range check vs model.nout)
  // covariates: per (occasion, covariate) segments as covariate.rs stores them
  std::vector<int64_t> cov_seg_off;  // [n_occ*n_cov+1]
  std::vector<double> seg_from, seg_to /* +inf = open */, seg_slope, seg_icpt /* slope==NaN => carry value in icpt */;
  std::vector<double> cov_first_t, cov_first_v, cov_last_t, cov_last_v;  // [n_occ*n_cov]

  // Covariate::interpolate (covariate.rs:216-241); returns false on MissingSegments.
  bool interpolate(int64_t occ, int32_t cov, double t, double* out) const;
};

// What the model contributes to the op stream.
struct CompileKey {
  int32_t eq_kind = PMX_EQ_ANALYTICAL;
  int32_t cov_time_mode = PMX_COV_TIME_SEGMENT_DT;
  double rk4_h_max = 0.0;  // ODE only
  int32_t n_rate = 1;      // rate columns carried per PROP op (1 for analytical: closed forms read rateiv[0])
  int32_t rate_input = 0;  // analytical: the input whose rate the closed form reads (1 under pm_* indexing,
                           // where rateiv slot 0 is a dead pad: analytical/mod.rs:86-88)
  int32_t class_g = 0;     // analytical GRID: members per chunk of the classed kernel (0 = no class plan)
  uint32_t lag_mask = 0;   // bit i: boluses on input i are delayed by a theta-dependent lag -> kept OUT of the
                           // op stream and merged per lane on the device (Occasion::add_lagtime, structs.rs:611-643)
  // covariate factors of the model's derived values (theta * f0 * f1): lane-independent, so the host evaluates
  // pow((cov/ref), e) / 1 + a (cov - ref) once per op instead of every lane on every op (op_fac)
  int32_t n_derived = 0;
  pmx_derived derived[PMX_MAX_DERIVED] = {};
  bool want_times = false; // PROP ops carry absolute [t0, t1) even without lag (custom ODE bodies may read the time)
  // user (hiprtc) analytical models, pmx_analytical.hpp:
  bool lag_merge = false;   // every input of lag_mask shares ONE list per occasion (lagb_input says which); the lane
                            // sorts it by its own landing times (a user lag may differ from bolus to bolus)
  bool solve_marks = false; // bit 24 of a PROP op = it continues the previous PROP's solve (seq_eq's parameter vector
                            // lives for one solve, analytical/mod.rs:331)
  bool full_rates = false;  // analytical: op_rate carries rateiv of EVERY input (a user `eq` may read any of them)
  bool user_cov = false;    // covariates are looked up on the device (segment tables uploaded; no op_cov / op_fac)
  int32_t prop_cache_slots = 0;  // covariate-derived rate constants: PROP ops of one occasion with the same (length,
                                 // covariate factors) have the same propagator; bits 24-26 of such ops say
                                 // "compute and keep in slot k" / "take slot k" (prop_cache_codes, pmx_compile.cpp)
  // three-compartment covariate models: op_kfac rows (DevOps) are built for this kernel-parameter -> derived-value map
  int32_t kfac_n = 0;                 // kernel parameters (0 = no op_kfac)
  int8_t kfac_map[8] = {-1, -1, -1, -1, -1, -1, -1, -1};  // derived index behind kernel parameter j, -1 = a primary parameter
  bool ladder = false;     // analytical, theta-only coefficients, no lag: PROP ops carry the exponential-ladder code
                           // (bits 27-29 of op_meta, pmx_structures.hpp ladder_pow)
  bool operator==(const CompileKey& o) const {
    return eq_kind == o.eq_kind && cov_time_mode == o.cov_time_mode && rk4_h_max == o.rk4_h_max &&
           n_rate == o.n_rate && rate_input == o.rate_input && class_g == o.class_g && lag_mask == o.lag_mask &&
           ladder == o.ladder && want_times == o.want_times && lag_merge == o.lag_merge && solve_marks == o.solve_marks &&
           full_rates == o.full_rates && user_cov == o.user_cov && prop_cache_slots == o.prop_cache_slots &&
           n_derived == o.n_derived && kfac_n == o.kfac_n && std::memcmp(kfac_map, o.kfac_map, sizeof(kfac_map)) == 0 &&
           std::memcmp(derived, o.derived, sizeof(derived)) == 0;
  }
};

struct OpStream {
  CompileKey key;
  int64_t n_ops = 0;
  std::vector<int64_t> subj_op_off;  // [S+1]
  std::vector<uint32_t> op_meta;     // kind | io<<8
  std::vector<double> op_a;          // BOLUS amount | OBS t_obs | PROP dt
  std::vector<double> op_b;          // PROP: rateiv[0] (analytical) / h (ODE)
  std::vector<int32_t> op_n;         // ODE PROP: RK4 step count (empty for analytical)
  std::vector<double> op_rate;       // ODE: [n_ops * n_rate] rateiv per PROP (empty for analytical)
  std::vector<double> op_cov;        // [n_ops * n_cov] covariates seen by the op (empty if n_cov == 0)
  std::vector<double> op_fac;        // [n_ops * n_derived * PMX_MAX_FACTORS] the derived values' factors at those covariates
  // lag models only (key.lag_mask != 0):
  std::vector<double> op_t0, op_t1;  // absolute [start, end] of every PROP; RESET: op_t0 = time of the occasion's
                                     // first remaining event (+inf if none), op_a = global occasion index
  int32_t n_lag_slots = 0;           // number of lagged inputs (slot k = k-th set bit of lag_mask)
  std::vector<int64_t> lagb_off;     // [(n_occasions * n_lag_slots) + 1] lagged boluses per (occasion, slot)
  std::vector<double> lagb_time;     // original (un-lagged) bolus time, sorted per (occasion, slot)
  std::vector<double> lagb_amount;
  std::vector<int32_t> lagb_input;   // the bolus' input (lag_merge lists mix inputs)
  int64_t max_lagb_per_list = 0;     // longest (occasion, slot) list
  int32_t prop_cache_used = 0;       // slots the stream's cache codes actually use (0: no reuse anywhere)
  int64_t n_prop_reused = 0;         // PROP ops that take a kept propagator instead of rebuilding it
  std::vector<int32_t> subj_order;   // subjects sorted by op count (desc), for the lane-per-pair kernels
  int32_t max_ops_per_subject = 0;
  int64_t n_prop = 0;
  // Largest dose input that reaches the state: every bolus, and every infusion that is active in
  // some PROP — what the reference range-checks against ndrugs (equation/mod.rs:322-327,
  // analytical/mod.rs:349-354, closure.rs:131-134).
  int32_t max_input_used = -1;
};

// "Classes" = subjects whose op streams are identical except for dose amounts / infusion rates
// (same op kinds, inputs, outeqs and the same PROP lengths dt): a shared dosing/sampling design,
// the normal case for trial protocols and simulation studies (C2/C3).  For such subjects the
// propagator exp(-lambda*dt) of a support point is the same value for every member, so the classed
// kernel computes it once per (lane, program step) and applies it to a register-resident batch of
// G members.  Identical arithmetic per (subject, support point); only the redundant exp() calls go.
struct ClassPlan {
  int32_t G = 0;                       // members per chunk (register batch of the classed kernel)
  int64_t n_chunks = 0;
  int64_t n_classed_subjects = 0;
  std::vector<uint32_t> prog_meta;     // concatenated class programs: kind | io<<8 | obs_after<<24 | outeq<<25
  std::vector<double> prog_dt;         // PROP: dt
  // lag models (one lagged input; exact classes only): the absolute [t0, t1) of every PROP step and, on a RESET step,
  // the time of the occasion's first event that stays in the list - what a lane compares its lagged bolus times with
  std::vector<double> prog_t0, prog_t1;
  std::vector<int64_t> cls_prog_off;   // [n_classes+1]
  std::vector<uint64_t> cls_fast_mask; // [n_classes] bit s (s < 63): step s is a PROP on the exponential ladder (rung 1..4) with
                                       // an observation of output 0 fused into it - the steps the log-likelihood kernel runs
                                       // as straight-line code when, in the chunk at hand, nobody infuses and every row is plain
  std::vector<int32_t> chunk_cls;      // [n_chunks]
  std::vector<int32_t> chunk_n;        // [n_chunks] live members (<= G)
  std::vector<int64_t> chunk_val_off;  // [n_chunks + 1] offset of the chunk's value block in `val` (+ the total)
  std::vector<int32_t> chunk_subj;     // [n_chunks*G] subject ids, -1 = padding
  std::vector<int64_t> chunk_row;      // [n_chunks*G] first prediction row of each member
  std::vector<double> val;             // per chunk: [program length][G]  BOLUS amount / PROP rate
  std::vector<uint64_t> chunk_rate_mask;  // [n_chunks] bit k: step k of the chunk's program carries a non-zero value for
                                       // some live member (an infusion running / a bolus); bit 63 also stands for every
                                       // step from 63 on.  A clear bit = a PROP step nobody infuses in: F only, no val fetch
  // Chunks [0, n_chunks_exact) belong to classes whose members share the whole program, step lengths included: one
  // propagator per step serves all G members.  Chunks [n_chunks_exact, n_chunks) belong to LOOSE classes: same op
  // kinds, inputs and outputs in the same order, but every member has its own step lengths (recorded sampling
  // times instead of protocol times) - `dtv` holds them, laid out like `val`, and the kernel builds a propagator
  // per member; what the members still share is the program walk and the paired stores.
  int64_t n_chunks_exact = 0;
  std::vector<double> dtv;             // per chunk: [program length][G]  PROP length of each member (loose chunks)
  // covariate-derived rate constants / volumes (every class is loose then): the members' own covariate factors
  // (OpStream::op_fac rows) of the step's PROP and of the observation fused into it, [program length][G][n_fac]
  int32_t n_fac = 0;
  std::vector<double> facp, faco;
  std::vector<int32_t> generic_subjects;  // subjects left to the generic kernel (ascending)
};

// Group the subjects of an analytical op stream into classes; classes with fewer than
// `min_class_size` members stay generic.
// Exponential-ladder code of a PROP of length dt that follows a PROP whose exponentials are live:
// n in 1..4 when dt == n * prev (within 8 ulp) and the accumulated error factor stays <= 1024, else 0 (fresh
// exp()).  Updates prev/span to describe the exponentials after this step.
uint32_t ladder_code(double dt, double* prev, double* span);

void build_class_plan(const HostPopulation& hp, const OpStream& os, int32_t G, int32_t min_class_size, ClassPlan* out,
                      bool ladder = true, bool spread = false, bool loose_classes = false);

// Fused per-subject step programs for the lean generic walker (pmx_kernels.hpp DevSteps): every OBS op rides on the step
// in front of it (bit 24 + outeq in bits 25-26, like the class plan's programs); an observation with nothing in front of
// it in its subject, or a second one at the same instant, is a step of kind OP_OBS.  rec = [n_steps + 1][4] doubles
// {meta bits, a, b, 0} (the last record is padding).  Analytical streams without lag only.
void build_step_stream(const OpStream& os, std::vector<int64_t>* subj_step_off, std::vector<double>* rec);

// Validate + copy + sort (Occasion::sort, structs.rs:669-671) + build covariate segments.
// Returns PMX_OK or an error with `err` filled.
int32_t build_host_population(const pmx_population_desc* d, HostPopulation* out, std::string* err);

// Flatten into an op stream for one model flavour.
int32_t compile_ops(const HostPopulation& hp, const CompileKey& key, OpStream* out, std::string* err);

}  // namespace pmx
