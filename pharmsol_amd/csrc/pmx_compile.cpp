// pmx_compile.cpp — see pmx_compile.hpp.  Host C++ only (no HIP).
#include "pmx_compile.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>

namespace pmx {

namespace {

// f64::total_cmp as an integer key (event.rs:301, covariate.rs:191).
inline int64_t total_key(double v) {
  int64_t b;
  std::memcpy(&b, &v, 8);
  b ^= static_cast<int64_t>(static_cast<uint64_t>(b >> 63) >> 1);
  return b;
}

struct ActiveInfusion {
  double time, amount, duration;
  int32_t input;
};

}  // namespace

bool HostPopulation::interpolate(int64_t occ, int32_t cov, double t, double* out) const {
  const int64_t idx = occ * n_cov + cov;
  const int64_t s0 = cov_seg_off[idx], s1 = cov_seg_off[idx + 1];
  if (s0 == s1) return false;
  // linear scan: first segment with from <= t < to (covariate.rs:221-227, :60-64)
  for (int64_t s = s0; s < s1; ++s) {
    if (seg_from[s] <= t && t < seg_to[s]) {
      *out = std::isnan(seg_slope[s]) ? seg_icpt[s] : (seg_slope[s] * t + seg_icpt[s]);
      return true;
    }
  }
  if (t < cov_first_t[idx]) {
    *out = cov_first_v[idx];
    return true;
  }
  if (t >= cov_last_t[idx]) {
    *out = cov_last_v[idx];
    return true;
  }
  return false;
}

int32_t build_host_population(const pmx_population_desc* d, HostPopulation* hp, std::string* err) {
  auto fail = [&](int32_t code, const std::string& m) {
    *err = m;
    return code;
  };
  if (!d) return fail(PMX_ERR_INVALID_ARGUMENT, "null population descriptor");
  if (d->n_subjects < 0 || d->n_occasions < 0 || d->n_events < 0)
    return fail(PMX_ERR_INVALID_ARGUMENT, "negative sizes");
  if (!d->subj_occ_off || !d->occ_ev_off) return fail(PMX_ERR_INVALID_ARGUMENT, "null offset arrays");
  if (d->n_events > 0 && (!d->ev_time || !d->ev_value || !d->ev_duration || !d->ev_kind || !d->ev_io))
    return fail(PMX_ERR_INVALID_ARGUMENT, "null event arrays");
  if (d->n_covariates < 0 || d->n_covariates > PMX_MAX_COVARIATES)
    return fail(PMX_ERR_INVALID_ARGUMENT, "n_covariates out of range");
  const int64_t S = d->n_subjects, NO = d->n_occasions, NE = d->n_events;
  if (d->subj_occ_off[0] != 0 || d->subj_occ_off[S] != NO)
    return fail(PMX_ERR_INVALID_ARGUMENT, "subj_occ_off must span [0, n_occasions]");
  if (d->occ_ev_off[0] != 0 || d->occ_ev_off[NO] != NE)
    return fail(PMX_ERR_INVALID_ARGUMENT, "occ_ev_off must span [0, n_events]");
  for (int64_t s = 0; s < S; ++s)
    if (d->subj_occ_off[s + 1] < d->subj_occ_off[s]) return fail(PMX_ERR_INVALID_ARGUMENT, "subj_occ_off not monotone");
  for (int64_t o = 0; o < NO; ++o)
    if (d->occ_ev_off[o + 1] < d->occ_ev_off[o]) return fail(PMX_ERR_INVALID_ARGUMENT, "occ_ev_off not monotone");

  hp->n_subjects = S;
  hp->n_occasions = NO;
  hp->n_events = NE;
  hp->n_cov = d->n_covariates;
  hp->subj_occ_off.assign(d->subj_occ_off, d->subj_occ_off + S + 1);
  hp->occ_ev_off.assign(d->occ_ev_off, d->occ_ev_off + NO + 1);
  hp->occ_index.resize(NO);
  for (int64_t s = 0; s < S; ++s)
    for (int64_t o = hp->subj_occ_off[s]; o < hp->subj_occ_off[s + 1]; ++o)
      hp->occ_index[o] = d->occ_index ? d->occ_index[o] : static_cast<int32_t>(o - hp->subj_occ_off[s]);

  hp->ev_time.resize(NE);
  hp->ev_value.resize(NE);
  hp->ev_dur.resize(NE);
  hp->ev_kind.resize(NE);
  hp->ev_io.resize(NE);
  std::vector<int64_t> ev_src(static_cast<size_t>(NE), 0);  // sorted position -> caller's event index
  std::vector<int64_t> order;
  for (int64_t o = 0; o < NO; ++o) {
    const int64_t e0 = hp->occ_ev_off[o], e1 = hp->occ_ev_off[o + 1];
    order.resize(e1 - e0);
    std::iota(order.begin(), order.end(), e0);
    if (!d->presorted) {
      // Occasion::sort: stable, time.total_cmp then Observation < Bolus < Infusion (event.rs:292-304)
      std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        const int64_t ka = total_key(d->ev_time[a]), kb = total_key(d->ev_time[b]);
        if (ka != kb) return ka < kb;
        return d->ev_kind[a] < d->ev_kind[b];
      });
    }
    for (int64_t i = 0; i < e1 - e0; ++i) {
      const int64_t src = order[i], dst = e0 + i;
      const uint8_t k = d->ev_kind[src];
      if (k > PMX_EV_INFUSION) return fail(PMX_ERR_INVALID_ARGUMENT, "unknown event kind");
      hp->ev_time[dst] = d->ev_time[src];
      hp->ev_value[dst] = d->ev_value[src];
      hp->ev_dur[dst] = d->ev_duration[src];
      hp->ev_kind[dst] = k;
      hp->ev_io[dst] = d->ev_io[src];
      ev_src[dst] = src;
      if (k == PMX_EV_OBSERVATION)
        hp->max_outeq = std::max<int32_t>(hp->max_outeq, d->ev_io[src]);
    }
  }
  // prediction rows (event order == flat_predictions order, subject.rs:145-148)
  hp->subj_obs_off.assign(S + 1, 0);
  for (int64_t s = 0; s < S; ++s) {
    const int64_t e0 = hp->occ_ev_off[hp->subj_occ_off[s]], e1 = hp->occ_ev_off[hp->subj_occ_off[s + 1]];
    for (int64_t e = e0; e < e1; ++e)
      if (hp->ev_kind[e] == PMX_EV_OBSERVATION) {
        hp->obs_time.push_back(hp->ev_time[e]);
        hp->obs_value.push_back(hp->ev_value[e]);
        hp->obs_outeq.push_back(hp->ev_io[e]);
        hp->obs_subject.push_back(s);
        if (d->ev_errorpoly)
          for (int c = 0; c < 4; ++c) hp->obs_errorpoly.push_back(d->ev_errorpoly[ev_src[e] * 4 + c]);
        if (d->ev_censor) {
          const int8_t cz = d->ev_censor[ev_src[e]];
          if (cz != PMX_CENSOR_NONE && cz != PMX_CENSOR_BLOQ && cz != PMX_CENSOR_ALOQ)
            return fail(PMX_ERR_INVALID_ARGUMENT, "unknown censoring code");
          hp->obs_censor.push_back(cz);
        }
      }
    hp->subj_obs_off[s + 1] = static_cast<int64_t>(hp->obs_time.size());
  }
  hp->n_obs = static_cast<int64_t>(hp->obs_time.size());

  // covariate segments (Covariate::build_segments, covariate.rs:189-214)
  const int32_t nc = hp->n_cov;
  if (nc > 0) {
    if (!d->cov_knot_off || !d->cov_knot_time || !d->cov_knot_value)
      return fail(PMX_ERR_INVALID_ARGUMENT, "covariate arrays missing");
    const int64_t ncell = NO * nc;
    hp->cov_seg_off.assign(ncell + 1, 0);
    hp->cov_first_t.resize(ncell);
    hp->cov_first_v.resize(ncell);
    hp->cov_last_t.resize(ncell);
    hp->cov_last_v.resize(ncell);
    std::vector<std::pair<double, double>> obs;
    for (int64_t c = 0; c < ncell; ++c) {
      const int64_t k0 = d->cov_knot_off[c], k1 = d->cov_knot_off[c + 1];
      if (k1 <= k0) return fail(PMX_ERR_INVALID_ARGUMENT, "a covariate has no observations in some occasion");
      obs.clear();
      for (int64_t k = k0; k < k1; ++k) obs.emplace_back(d->cov_knot_time[k], d->cov_knot_value[k]);
      std::stable_sort(obs.begin(), obs.end(),
                       [](const auto& a, const auto& b) { return total_key(a.first) < total_key(b.first); });
      const bool fixed = d->cov_fixed && d->cov_fixed[c];
      const size_t n = obs.size();
      for (size_t i = 0; i < n; ++i) {
        const bool has_next = i + 1 < n;
        hp->seg_from.push_back(obs[i].first);
        hp->seg_to.push_back(has_next ? obs[i + 1].first : std::numeric_limits<double>::infinity());
        if (fixed || !has_next) {
          hp->seg_slope.push_back(std::numeric_limits<double>::quiet_NaN());  // CarryForward
          hp->seg_icpt.push_back(obs[i].second);
        } else {
          const double slope = (obs[i + 1].second - obs[i].second) / (obs[i + 1].first - obs[i].first);
          hp->seg_slope.push_back(slope);
          hp->seg_icpt.push_back(obs[i].second - slope * obs[i].first);
        }
      }
      hp->cov_seg_off[c + 1] = static_cast<int64_t>(hp->seg_from.size());
      hp->cov_first_t[c] = obs.front().first;
      hp->cov_first_v[c] = obs.front().second;
      hp->cov_last_t[c] = obs.back().first;
      hp->cov_last_v[c] = obs.back().second;
    }
  }
  return PMX_OK;
}

int32_t compile_ops(const HostPopulation& hp, const CompileKey& key, OpStream* os, std::string* err) {
  const int32_t nc = hp.n_cov;
  const bool ode = key.eq_kind == PMX_EQ_ODE;
  const int32_t n_rate = key.n_rate;
  os->key = key;
  os->subj_op_off.assign(hp.n_subjects + 1, 0);
  os->op_meta.clear();
  os->op_a.clear();
  os->op_b.clear();
  os->op_n.clear();
  os->op_rate.clear();
  os->op_cov.clear();
  os->op_fac.clear();
  os->op_t0.clear();
  os->op_t1.clear();
  os->lagb_off.clear();
  os->lagb_time.clear();
  os->lagb_amount.clear();
  os->lagb_input.clear();
  os->max_lagb_per_list = 0;
  os->prop_cache_used = 0;
  os->n_prop_reused = 0;
  os->n_prop = 0;
  const uint32_t lag_mask = key.lag_mask;
  int32_t slot_of_input[PMX_MAX_INPUTS];
  int32_t n_slots = 0;
  for (int i = 0; i < PMX_MAX_INPUTS; ++i) slot_of_input[i] = ((lag_mask >> i) & 1u) ? (key.lag_merge ? 0 : n_slots++) : -1;
  if (key.lag_merge && lag_mask != 0) n_slots = 1;
  os->n_lag_slots = n_slots;
  struct LagBolus {
    double first, second;  // recorded time, amount
    int32_t input;
  };
  std::vector<std::vector<LagBolus>> lagb(n_slots > 0 ? n_slots : 1);  // per slot, this occasion
  if (n_slots > 0) os->lagb_off.push_back(0);
  const bool times = n_slots > 0 || key.want_times;  // PROP ops carry their absolute [t0, t1)

  std::vector<double> covv(nc > 0 ? nc : 1, 0.0);
  std::vector<double> rate(n_rate > 0 ? n_rate : 1, 0.0);
  bool cov_missing = false;

  auto push = [&](uint32_t kind, uint32_t io, double a, double b, int32_t n, const double* rates, int64_t occ,
                  double t_cov, bool want_cov) {
    os->op_meta.push_back(make_meta(kind, io));
    os->op_a.push_back(a);
    os->op_b.push_back(b);
    if (times) {
      os->op_t0.push_back(0.0);
      os->op_t1.push_back(0.0);
    }
    if (ode) os->op_n.push_back(n);
    if (ode || key.full_rates)
      for (int32_t r = 0; r < n_rate; ++r) os->op_rate.push_back(rates ? rates[r] : 0.0);
    if (nc > 0 && !key.user_cov) {
      for (int32_t c = 0; c < nc; ++c) {
        double v = 0.0;
        if (want_cov && !hp.interpolate(occ, c, t_cov, &v)) cov_missing = true;
        os->op_cov.push_back(v);
        covv[c] = v;
      }
      // the factors of every derived value at these covariates (expand/analytical.rs:254,286; bindings.rs:98-117),
      // written exactly like the per-lane expression they replace
      for (int32_t d = 0; d < key.n_derived; ++d)
        for (int32_t k = 0; k < PMX_MAX_FACTORS; ++k) {
          double fac = 1.0;
          if (want_cov && k < key.derived[d].n_factors) {
            const pmx_factor& f = key.derived[d].f[k];
            const double cv = covv[f.cov];
            fac = (f.op == PMX_F_POW) ? std::pow(cv / f.ref, f.coef) : (1.0 + f.coef * (cv - f.ref));
          }
          os->op_fac.push_back(fac);
        }
    }
  };

  std::vector<ActiveInfusion> inf;
  std::vector<double> ts, bounds;
  int32_t max_input_used = -1;

  for (int64_t s = 0; s < hp.n_subjects; ++s) {
    for (int64_t oc = hp.subj_occ_off[s]; oc < hp.subj_occ_off[s + 1]; ++oc) {
      const int64_t e0 = hp.occ_ev_off[oc], e1 = hp.occ_ev_off[oc + 1];
      // initial_state: zeros, init only for occasion index 0 (analytical/mod.rs:409-426)
      push(OP_RESET, hp.occ_index[oc] == 0 ? 1u : 0u, static_cast<double>(oc), 0.0, 0, nullptr, oc, 0.0, false);
      const size_t reset_op = os->op_meta.size() - 1;
      inf.clear();
      if (n_slots > 0) {
        // Lagged boluses leave the event list (the device merges them at t + lag(theta)); what remains is
        // walked exactly like before.  ev_keep = indices of the remaining events, still sorted.
        for (auto& v : lagb) v.clear();
        double t_first = std::numeric_limits<double>::infinity();
        uint32_t k_first = 3u;  // kind of the first remaining event (PMX_EV_*; 3 = none)
        for (int64_t e = e0; e < e1; ++e) {
          const bool lagged = hp.ev_kind[e] == PMX_EV_BOLUS && hp.ev_io[e] < PMX_MAX_INPUTS && slot_of_input[hp.ev_io[e]] >= 0;
          if (lagged) {
            max_input_used = std::max<int32_t>(max_input_used, hp.ev_io[e]);
            lagb[slot_of_input[hp.ev_io[e]]].push_back({hp.ev_time[e], hp.ev_value[e], static_cast<int32_t>(hp.ev_io[e])});
          } else if (k_first == 3u) {
            t_first = hp.ev_time[e];
            k_first = hp.ev_kind[e];
          }
        }
        os->op_t0[reset_op] = t_first;
        // ODE: which event is FIRST in the re-sorted list decides what is applied at the solver clock without integration
        // (pmx_ode.hpp "the solver clock"); at equal times a bolus sorts behind an observation and in front of an infusion
        // (event.rs:292-304).  Bits 25-26 of the RESET op = kind of the first remaining event.
        os->op_meta[reset_op] |= k_first << 25;
        for (int32_t k = 0; k < n_slots; ++k) {
          for (const auto& tb : lagb[k]) {
            os->lagb_time.push_back(tb.first);
            os->lagb_amount.push_back(tb.second);
            os->lagb_input.push_back(tb.input);
          }
          os->max_lagb_per_list = std::max<int64_t>(os->max_lagb_per_list, static_cast<int64_t>(lagb[k].size()));
          os->lagb_off.push_back(static_cast<int64_t>(os->lagb_time.size()));
        }
      }
      if (!ode) {
        int64_t prev = -1;  // previous event that stays in the list
        for (int64_t e = e0; e < e1; ++e) {  // simulate_event, equation/mod.rs:300-358
          const uint8_t k = hp.ev_kind[e];
          if (n_slots > 0 && k == PMX_EV_BOLUS && hp.ev_io[e] < PMX_MAX_INPUTS && slot_of_input[hp.ev_io[e]] >= 0) continue;
          (void)prev;
          if (k == PMX_EV_BOLUS) {
            max_input_used = std::max<int32_t>(max_input_used, hp.ev_io[e]);
            push(OP_BOLUS, hp.ev_io[e], hp.ev_value[e], hp.ev_time[e], 0, nullptr, oc, 0.0, false);  // (op_b: its time, for a user fa)
          } else if (k == PMX_EV_INFUSION) {
            inf.push_back({hp.ev_time[e], hp.ev_value[e], hp.ev_dur[e], static_cast<int32_t>(hp.ev_io[e])});
          } else {
            // Lag models: the lagged boluses are not in this list; the device merges them into the PROP steps at their
            // landing times.  An observation that no PROP step precedes (events closer than the solve's 1e-12 dedup, or
            // at the same instant) can still have a lagged bolus landing in front of it - a zero lag leaves the bolus
            // where it was recorded, between two observations one ulp apart (found by the fuzz suite: seed 2235).  Bit 31
            // tells the device to take the boluses landing before this observation's time first.
            const bool flush = n_slots > 0 && !os->op_meta.empty() && (os->op_meta.back() & 0xffu) != OP_PROP;
            push(OP_OBS, hp.ev_io[e], hp.ev_time[e], 0.0, 0, nullptr, oc, hp.ev_time[e], true);
            if (flush) os->op_meta.back() |= (1u << 31);
          }
          int64_t en = e + 1;  // next event that stays in the list
          if (n_slots > 0)
            while (en < e1 && hp.ev_kind[en] == PMX_EV_BOLUS && hp.ev_io[en] < PMX_MAX_INPUTS &&
                   slot_of_input[hp.ev_io[en]] >= 0)
              ++en;
          if (en < e1) {  // Analytical::solve, analytical/mod.rs:299-370
            const double ti = hp.ev_time[e], tf = hp.ev_time[en];
            if (ti == tf) continue;  // :308-310
            ts.clear();
            ts.push_back(ti);
            ts.push_back(tf);
            for (const auto& f : inf) {  // :316-325 strictly-inside breakpoints
              const double t0 = f.time, t1 = t0 + f.duration;
              if (t0 > ti && t0 < tf) ts.push_back(t0);
              if (t1 > ti && t1 < tf) ts.push_back(t1);
            }
            std::sort(ts.begin(), ts.end());  // :326
            {                                 // dedup_by |a-b| < 1e-12 against the last retained, :327
              size_t w = 1;
              for (size_t r = 1; r < ts.size(); ++r)
                if (!(std::fabs(ts[r] - ts[w - 1]) < 1e-12)) ts[w++] = ts[r];
              ts.resize(w);
            }
            double cur = ts[0];
            for (size_t i = 1; i < ts.size(); ++i) {  // :334-367
              const double nxt = ts[i];
              double r0 = 0.0;  // rateiv[0]: the only slot the closed forms read
              if (key.full_rates) std::fill(rate.begin(), rate.end(), 0.0);
              for (const auto& f : inf) {
                const double st = f.time, en = st + f.duration;
                if (cur >= st && nxt <= en) {
                  max_input_used = std::max(max_input_used, f.input);
                  if (f.input == key.rate_input) r0 += f.amount / f.duration;  // :355
                  if (key.full_rates && f.input < n_rate) rate[f.input] += f.amount / f.duration;
                }
              }
              const double dt = nxt - cur;
              const double t_cov = key.cov_time_mode == PMX_COV_TIME_SEGMENT_END_ABS ? nxt : dt;
              push(OP_PROP, 0, dt, r0, 0, key.full_rates ? rate.data() : nullptr, oc, t_cov, true);
              if (key.solve_marks && i > 1) os->op_meta.back() |= (1u << 24);  // a later sub-segment of the same solve
              if (times) {
                os->op_t0.back() = cur;
                os->op_t1.back() = nxt;
              }
              os->n_prop++;
              cur = nxt;
            }
          }
        }
      } else {
        // ODE::run_events (ode/mod.rs:609-823) with InfusionSchedule over ALL infusions
        // of the occasion (closure.rs:109-180).
        bounds.clear();
        for (int64_t e = e0; e < e1; ++e) {
          if (hp.ev_kind[e] != PMX_EV_INFUSION) continue;
          if (hp.ev_dur[e] <= 0.0) continue;  // closure.rs:127-129
          max_input_used = std::max<int32_t>(max_input_used, hp.ev_io[e]);
          inf.push_back({hp.ev_time[e], hp.ev_value[e], hp.ev_dur[e], static_cast<int32_t>(hp.ev_io[e])});
          bounds.push_back(hp.ev_time[e]);
          bounds.push_back(hp.ev_time[e] + hp.ev_dur[e]);
        }
        std::sort(bounds.begin(), bounds.end());
        bounds.erase(std::unique(bounds.begin(), bounds.end()), bounds.end());  // exact dedup, closure.rs:143-148
        auto is_lagged = [&](int64_t e) {
          return n_slots > 0 && hp.ev_kind[e] == PMX_EV_BOLUS && hp.ev_io[e] < PMX_MAX_INPUTS && slot_of_input[hp.ev_io[e]] >= 0;
        };
        // The solver clock starts at Occasion::initial_time() of the occasion AS RECORDED - lagged boluses at their
        // recorded times included (ode/mod.rs:348 takes it from the occasion, not from the lag-rewritten event list;
        // structs.rs:782-793) - and only moves forward to the time of the next event (ode/mod.rs:719-721).  PROP ops
        // therefore start there; a lane whose own clock is ahead (boluses that landed before the first remaining event
        // took it there) starts its piece at its clock (pmx_ode.hpp / pmx_ode_user.hpp).  RESET: op_b = that time.
        double t = 0.0;
        for (int64_t e = e0; e < e1; ++e) t = (e == e0) ? hp.ev_time[e] : std::min(t, hp.ev_time[e]);
        os->op_b[reset_op] = t;
        size_t bcur = 0;
        for (int64_t e = e0; e < e1; ++e) {
          const uint8_t k = hp.ev_kind[e];
          if (is_lagged(e)) continue;
          if (k == PMX_EV_BOLUS) {
            max_input_used = std::max<int32_t>(max_input_used, hp.ev_io[e]);
            push(OP_BOLUS, hp.ev_io[e], hp.ev_value[e], hp.ev_time[e], 0, nullptr, oc, 0.0, false);  // (op_b: its time, for a user fa / bolus jump)
          } else if (k == PMX_EV_OBSERVATION) {
            push(OP_OBS, hp.ev_io[e], hp.ev_time[e], 0.0, 0, nullptr, oc, hp.ev_time[e], true);
          }
          int64_t en = e + 1;
          while (en < e1 && is_lagged(en)) ++en;
          if (en < e1) {
            const double next_t = hp.ev_time[en];
            while (next_t > t) {  // ode/mod.rs:721-739
              while (bcur < bounds.size() && bounds[bcur] <= t) ++bcur;
              double stop = next_t;
              if (bcur < bounds.size() && bounds[bcur] <= next_t) stop = bounds[bcur++];
              std::fill(rate.begin(), rate.end(), 0.0);
              for (const auto& f : inf) {  // right-continuous rate at t (closure.rs:80-99)
                const double st = f.time, en = st + f.duration;
                if (st <= t && t < en && f.input < n_rate) rate[f.input] += f.amount / f.duration;
              }
              const double dt = stop - t;
              if (dt > 0.0) {
                double nf = std::ceil(dt / key.rk4_h_max);
                if (nf < 1.0) nf = 1.0;
                if (nf > 2.0e9) {
                  *err = "RK4 step count overflow (dt / rk4_h_max too large)";
                  return PMX_ERR_INVALID_ARGUMENT;
                }
                const int32_t n = static_cast<int32_t>(nf);
                push(OP_PROP, 0, dt, dt / static_cast<double>(n), n, rate.data(), oc, t, true);
                if (times) {
                  os->op_t0.back() = t;
                  os->op_t1.back() = stop;
                }
                os->n_prop++;
              }
              t = stop;
            }
          }
        }
      }
    }
    os->subj_op_off[s + 1] = static_cast<int64_t>(os->op_meta.size());
    if (key.prop_cache_slots > 0 && !ode && !os->op_fac.empty()) {
      // Propagator reuse for covariate-derived rate constants.  The macro lowering evaluates `derive` at the segment
      // LENGTH (expand/analytical.rs:254,286), so two PROPs of equal length see equal covariates, hence equal rate
      // constants and the same transition matrix (a subject-constant covariate gives the same under either rule).
      // Per occasion: key = (dt, factor row) bitwise; a key with a later use gets a slot (furthest-next-use eviction);
      // code 1 + k = build and keep in slot k, 1 + S + k = take slot k (S = prop_cache_slots), 0 = build.
      const int32_t NS_ = key.prop_cache_slots;
      const size_t nfac = static_cast<size_t>(key.n_derived) * PMX_MAX_FACTORS;
      const int64_t o0 = os->subj_op_off[s], o1 = os->subj_op_off[s + 1];
      auto same = [&](int64_t a, int64_t b) {
        return std::memcmp(&os->op_a[a], &os->op_a[b], 8) == 0 && (os->op_b[a] != 0.0) == (os->op_b[b] != 0.0) &&  // (rate-free
               // segments keep the transition part only: they share among themselves, segments under an infusion likewise)
               std::memcmp(&os->op_fac[static_cast<size_t>(a) * nfac], &os->op_fac[static_cast<size_t>(b) * nfac], nfac * 8) == 0;
      };
      int64_t seg0 = o0;
      while (seg0 < o1) {  // one occasion at a time (a RESET re-derives the lane's failure state)
        int64_t seg1 = seg0 + 1;
        while (seg1 < o1 && (os->op_meta[seg1] & 0xffu) != OP_RESET) ++seg1;
        std::vector<int64_t> props;
        for (int64_t o = seg0; o < seg1; ++o)
          if ((os->op_meta[o] & 0xffu) == OP_PROP) props.push_back(o);
        const size_t n = props.size();
        std::vector<int64_t> next(n, -1);  // index (into props) of the next PROP with the same key
        for (size_t i = 0; i < n; ++i)
          for (size_t j = i + 1; j < n; ++j)
            if (same(props[i], props[j])) {
              next[i] = static_cast<int64_t>(j);
              break;
            }
        int64_t last_built = -1;
        std::vector<int64_t> holder(static_cast<size_t>(NS_), -1);  // slot -> index of the PROP whose propagator it holds
        std::vector<int32_t> slot_of(n, -1);
        for (size_t i = 0; i < n; ++i) {
          int32_t from = -1;
          for (int32_t k = 0; k < NS_; ++k)
            if (holder[static_cast<size_t>(k)] >= 0 && next[static_cast<size_t>(holder[static_cast<size_t>(k)])] == static_cast<int64_t>(i)) from = k;
          uint32_t code = 0;
          if (from >= 0) {  // take it; the slot now stands for this PROP (same key, next use continues the chain)
            code = static_cast<uint32_t>(1 + NS_ + from);
            holder[static_cast<size_t>(from)] = next[i] >= 0 ? static_cast<int64_t>(i) : -1;
            os->n_prop_reused++;
            os->prop_cache_used = std::max(os->prop_cache_used, from + 1);
          } else if (next[i] >= 0) {  // first of several: keep it if a slot is free or holds something needed later than this
            int32_t pick = -1;
            int64_t worst = -1;
            for (int32_t k = 0; k < NS_; ++k) {
              const int64_t h = holder[static_cast<size_t>(k)];
              if (h < 0) {
                pick = k;
                worst = std::numeric_limits<int64_t>::max();
                break;
              }
              if (next[static_cast<size_t>(h)] > worst) {
                worst = next[static_cast<size_t>(h)];
                pick = k;
              }
            }
            if (pick >= 0 && (holder[static_cast<size_t>(pick)] < 0 || worst > next[i])) {
              holder[static_cast<size_t>(pick)] = static_cast<int64_t>(i);
              code = static_cast<uint32_t>(1 + pick);
            }
          }
          os->op_meta[props[i]] |= code << 24;
          // bit 27: this PROP's covariate factor row equals the one of the occasion's previous BUILT PROP (a subject-constant
          // covariate: every segment) - the rate constants, hence the eigenvalues, are the same and only the step length
          // differs (pmx_analytical_dyn3 keeps the last eigenvalues in registers).  "Built" = not taken from a slot.
          if (code <= static_cast<uint32_t>(NS_)) {
            if (last_built >= 0 && std::memcmp(&os->op_fac[static_cast<size_t>(props[i]) * nfac],
                                               &os->op_fac[static_cast<size_t>(last_built) * nfac], nfac * 8) == 0)
              os->op_meta[props[i]] |= 1u << 27;
            last_built = props[i];
          }
        }
        seg0 = seg1;
      }
    }
    if (key.ladder && !ode) {  // exponential ladder along the subject's PROP ops (the lane's rate constants never change)
      double prev = 0.0, span = 1.0;
      for (int64_t o = os->subj_op_off[s]; o < os->subj_op_off[s + 1]; ++o)
        if ((os->op_meta[o] & 0xffu) == OP_PROP) os->op_meta[o] |= ladder_code(os->op_a[o], &prev, &span) << 27;
    }
  }
  if (cov_missing) {
    *err = "covariate interpolation failed (MissingSegments)";
    return PMX_ERR_INVALID_ARGUMENT;
  }
  os->n_ops = static_cast<int64_t>(os->op_meta.size());
  // lane-per-pair kernels: neighbours in a wavefront should have similar op counts
  os->subj_order.resize(hp.n_subjects);
  std::iota(os->subj_order.begin(), os->subj_order.end(), 0);
  auto work = [&](int32_t s) -> int64_t {
    if (!ode) return os->subj_op_off[s + 1] - os->subj_op_off[s];
    int64_t w = 0;
    for (int64_t o = os->subj_op_off[s]; o < os->subj_op_off[s + 1]; ++o) w += 1 + os->op_n[o];
    return w;
  };
  std::vector<int64_t> wk(hp.n_subjects);
  for (int64_t s = 0; s < hp.n_subjects; ++s) wk[s] = work(static_cast<int32_t>(s));
  std::stable_sort(os->subj_order.begin(), os->subj_order.end(), [&](int32_t a, int32_t b) { return wk[a] > wk[b]; });
  int64_t mx = 0;
  for (int64_t s = 0; s < hp.n_subjects; ++s) mx = std::max(mx, os->subj_op_off[s + 1] - os->subj_op_off[s]);
  os->max_ops_per_subject = static_cast<int32_t>(mx);
  os->max_input_used = max_input_used;
  return PMX_OK;
}

}  // namespace pmx

// ------------------------------------------------------------------------------------
// class plan
// ------------------------------------------------------------------------------------
#include <unordered_map>

namespace pmx {

namespace {
inline uint64_t mix64(uint64_t h, uint64_t v) {
  h ^= v + 0x9E3779B97F4A7C15ULL + (h << 6) + (h >> 2);
  h *= 0xBF58476D1CE4E5B9ULL;
  return h ^ (h >> 31);
}
}  // namespace

void build_step_stream(const OpStream& os, std::vector<int64_t>* subj_step_off, std::vector<double>* rec) {
  const int64_t S = static_cast<int64_t>(os.subj_op_off.size()) - 1;
  subj_step_off->assign(static_cast<size_t>(S) + 1, 0);
  rec->clear();
  rec->reserve(static_cast<size_t>(os.n_ops) * 2 + 4);
  auto push = [&](uint64_t meta, double a, double b) {
    double w;
    std::memcpy(&w, &meta, 8);
    rec->push_back(w);
    rec->push_back(a);
    rec->push_back(b);
    rec->push_back(0.0);
  };
  int64_t n_steps = 0;
  for (int64_t s = 0; s < S; ++s) {
    int64_t last = -1;  // this subject's last step, if it can still take an observation
    for (int64_t o = os.subj_op_off[s]; o < os.subj_op_off[s + 1]; ++o) {
      const uint32_t meta = os.op_meta[o];
      const uint32_t kind = meta & 0xffu, io = (meta >> 8) & 0xffffu;
      if (kind == OP_OBS) {
        const uint64_t tag = (1ull << 24) | (static_cast<uint64_t>(io & 3u) << 25);
        if (last >= 0) {
          uint64_t w;
          std::memcpy(&w, &(*rec)[static_cast<size_t>(last) * 4], 8);
          w |= tag;
          std::memcpy(&(*rec)[static_cast<size_t>(last) * 4], &w, 8);
          last = -1;
        } else {
          push(static_cast<uint64_t>(OP_OBS) | tag, 0.0, 0.0);
          ++n_steps;
        }
      } else {
        // kind | io | the PROP's ladder rung (bits 27-29 of the op, pmx_compile.cpp ladder_code)
        push(static_cast<uint64_t>(kind) | (static_cast<uint64_t>(io) << 8) | (static_cast<uint64_t>(meta) & (7ull << 27)), os.op_a[o],
             os.op_b[o]);
        last = n_steps++;
      }
    }
    (*subj_step_off)[static_cast<size_t>(s) + 1] = n_steps;
  }
  push(static_cast<uint64_t>(OP_OBS), 0.0, 0.0);  // padding: the walker requests one record past a subject's last step
}

uint32_t ladder_code(double dt, double* prev, double* span) {
  uint32_t code = 0;
  if (*prev > 0.0 && dt > 0.0) {
    for (uint32_t n = 1; n <= 4; ++n) {
      if (std::fabs(dt - n * *prev) <= 8.0 * std::numeric_limits<double>::epsilon() * dt && *span * n <= 1024.0) {
        code = n;
        break;
      }
    }
  }
  if (code) {
    *span *= code;
    *prev = code * *prev;
  } else {
    *span = 1.0;
    *prev = dt;
  }
  return code;
}

void build_class_plan(const HostPopulation& hp, const OpStream& os, int32_t G, int32_t min_class_size, ClassPlan* cp,
                      bool ladder, bool spread, bool loose_classes) {
  *cp = ClassPlan{};
  cp->G = G;
  const int64_t S = hp.n_subjects;
  const bool dyn = os.key.n_derived > 0 && !os.op_fac.empty();  // covariate-derived constants: nothing to share but the program shape
  const size_t nfac = static_cast<size_t>(os.key.n_derived) * PMX_MAX_FACTORS;
  cp->n_fac = static_cast<int32_t>(nfac);
  if (dyn) loose_classes = true;
  const bool lagged = os.n_lag_slots == 1;  // (the caller only asks for a plan of a lag model when one input is lagged)
  if (os.n_lag_slots > 1) return;
  if (lagged) loose_classes = false;  // loose members do not share the times
  auto sig_key = [&](int64_t o) -> uint64_t {  // what must match between class members, per op
    const uint32_t kind = os.op_meta[o] & 0xffu;
    uint64_t bits = 0;
    if (kind == OP_PROP) std::memcpy(&bits, &os.op_a[o], 8);  // dt; BOLUS amount / PROP rate / OBS time are free
    return (static_cast<uint64_t>(os.op_meta[o]) << 1) ^ (bits * 0x9E3779B97F4A7C15ULL) ^ bits;
  };
  auto same_program = [&](int64_t a, int64_t b) {
    const int64_t a0 = os.subj_op_off[a], a1 = os.subj_op_off[a + 1];
    const int64_t b0 = os.subj_op_off[b], b1 = os.subj_op_off[b + 1];
    if (a1 - a0 != b1 - b0) return false;
    for (int64_t i = 0; i < a1 - a0; ++i) {
      if (os.op_meta[a0 + i] != os.op_meta[b0 + i]) return false;
      const uint32_t kind = os.op_meta[a0 + i] & 0xffu;
      if (kind == OP_PROP && std::memcmp(&os.op_a[a0 + i], &os.op_a[b0 + i], 8) != 0) return false;
      if (lagged) {  // members of a lag class also share every absolute time a lane's lagged boluses are compared with
        if (kind == OP_PROP && (std::memcmp(&os.op_t0[a0 + i], &os.op_t0[b0 + i], 8) != 0 ||
                                std::memcmp(&os.op_t1[a0 + i], &os.op_t1[b0 + i], 8) != 0))
          return false;
        if (kind == OP_RESET) {
          if (std::memcmp(&os.op_t0[a0 + i], &os.op_t0[b0 + i], 8) != 0) return false;
          const int64_t oa = static_cast<int64_t>(os.op_a[a0 + i]), ob = static_cast<int64_t>(os.op_a[b0 + i]);
          const int64_t la = os.lagb_off[oa], lb = os.lagb_off[ob];
          const int64_t na = os.lagb_off[oa + 1] - la;
          if (na != os.lagb_off[ob + 1] - lb) return false;
          if (na > 0 && std::memcmp(&os.lagb_time[la], &os.lagb_time[lb], static_cast<size_t>(na) * 8) != 0) return false;
        }
      }
    }
    return true;
  };
  // class id per subject (representative = first subject seen with that program)
  std::unordered_map<uint64_t, std::vector<int32_t>> buckets;  // hash -> class ids
  std::vector<int32_t> cls_rep;                                 // class -> representative subject
  std::vector<std::vector<int32_t>> members;
  for (int64_t s = 0; s < S; ++s) {
    const int64_t o0 = os.subj_op_off[s], o1 = os.subj_op_off[s + 1];
    if (o1 == o0) {  // an empty subject: nothing to compute, but the generic walker still owns its status bytes
      cp->generic_subjects.push_back(static_cast<int32_t>(s));
      continue;
    }
    if (os.key.rate_input == 1) {  // pm_ indexing: a bolus into input 0 lands in the wrapper's pad slot (generic walker only)
      bool pad_dose = false;
      for (int64_t o = o0; o < o1; ++o)
        if ((os.op_meta[o] & 0xffu) == OP_BOLUS && ((os.op_meta[o] >> 8) & 0xffffu) == 0u) pad_dose = true;
      if (pad_dose) {
        cp->generic_subjects.push_back(static_cast<int32_t>(s));
        continue;
      }
    }
    uint64_t h = static_cast<uint64_t>(o1 - o0);
    for (int64_t o = o0; o < o1; ++o) {
      h = mix64(h, sig_key(o));
      if (lagged) {
        uint64_t bits = 0;
        std::memcpy(&bits, &os.op_t0[o], 8);
        h = mix64(h, bits);
        if ((os.op_meta[o] & 0xffu) == OP_RESET) {
          const int64_t oc = static_cast<int64_t>(os.op_a[o]);
          for (int64_t q = os.lagb_off[oc]; q < os.lagb_off[oc + 1]; ++q) {
            std::memcpy(&bits, &os.lagb_time[q], 8);
            h = mix64(h, bits);
          }
        }
      }
    }
    auto& ids = buckets[h];
    int32_t cls = -1;
    for (int32_t c : ids)
      if (same_program(cls_rep[c], s)) {
        cls = c;
        break;
      }
    if (cls < 0) {
      cls = static_cast<int32_t>(cls_rep.size());
      cls_rep.push_back(static_cast<int32_t>(s));
      members.emplace_back();
      ids.push_back(cls);
    }
    members[cls].push_back(static_cast<int32_t>(s));
  }
  cp->cls_prog_off.push_back(0);
  int32_t out_cls = 0;
  // one class -> its program and its chunks; `loose`: the members' PROP lengths differ (dtv), no ladder
  auto emit_class = [&](const std::vector<int32_t>& mem, int32_t rep, bool loose) {
    const int64_t r0 = os.subj_op_off[rep], r1 = os.subj_op_off[rep + 1];
    // Program steps: every OBS op is FUSED into the step before it (bit 24 = "emit a row after this step",
    // bits 25-26 = its outeq), so a PROP+OBS pair costs one trip of the device loop.  A second observation
    // at the same instant gets a step of its own (kind OP_OBS = no state change).
    std::vector<int32_t> step_of_op(static_cast<size_t>(r1 - r0), -1);
    std::vector<int32_t> obs_step_of_op(static_cast<size_t>(r1 - r0), -1);  // OBS op -> the step that emits its row
    std::vector<uint32_t> step_meta;
    std::vector<double> step_dt, step_t0, step_t1;
    for (int64_t o = r0; o < r1; ++o) {
      const uint32_t kind = os.op_meta[o] & 0xffu;
      const uint32_t io = (os.op_meta[o] >> 8) & 0xffffu;
      if (kind == OP_OBS) {
        const uint32_t flush = os.op_meta[o] & (1u << 31);  // lag models: lagged boluses may land in front of it (see above)
        if (!step_meta.empty() && ((step_meta.back() >> 24) & 1u) == 0u) {
          step_meta.back() |= (1u << 24) | ((io & 3u) << 25) | flush;
          if (flush) step_t1.back() = os.op_a[o];  // (never a PROP step: its t1 slot is free) the observation's time
          obs_step_of_op[static_cast<size_t>(o - r0)] = static_cast<int32_t>(step_meta.size()) - 1;
        } else {
          obs_step_of_op[static_cast<size_t>(o - r0)] = static_cast<int32_t>(step_meta.size());
          step_meta.push_back(make_meta(OP_OBS, 0) | (1u << 24) | ((io & 3u) << 25) | flush);
          step_dt.push_back(0.0);
          step_t0.push_back(0.0);
          step_t1.push_back(flush ? os.op_a[o] : 0.0);
        }
      } else {
        step_meta.push_back(make_meta(kind, io));
        step_dt.push_back(kind == OP_PROP ? os.op_a[o] : 0.0);
        step_t0.push_back(lagged ? os.op_t0[o] : 0.0);
        step_t1.push_back(lagged ? os.op_t1[o] : 0.0);
        step_of_op[static_cast<size_t>(o - r0)] = static_cast<int32_t>(step_meta.size()) - 1;
      }
    }
    const int64_t L = static_cast<int64_t>(step_meta.size());
    if (ladder && !loose) {
      // Exponential ladder (pmx_structures.hpp ladder_pow): bits 27-29 of a PROP step = n when its length is
      // n x the previous PROP's (n = 1: same propagator again).  `span` = how many times the first rung's
      // rounding error has been multiplied; past 1024 the next step starts a fresh ladder.
      double prev = 0.0, span = 1.0;
      for (int64_t i = 0; i < L; ++i) {
        if ((step_meta[static_cast<size_t>(i)] & 0xffu) != OP_PROP) continue;
        step_meta[static_cast<size_t>(i)] |= ladder_code(step_dt[static_cast<size_t>(i)], &prev, &span) << 27;
      }
    }
    for (int64_t i = 0; i < L; ++i) {
      cp->prog_meta.push_back(step_meta[static_cast<size_t>(i)]);
      cp->prog_dt.push_back(loose ? 0.0 : step_dt[static_cast<size_t>(i)]);
      cp->prog_t0.push_back(step_t0[static_cast<size_t>(i)]);
      cp->prog_t1.push_back(step_t1[static_cast<size_t>(i)]);
    }
    cp->cls_prog_off.push_back(static_cast<int64_t>(cp->prog_meta.size()));
    {
      uint64_t fast = 0;
      for (int64_t i = 0; i < L && i < 63; ++i) {
        const uint32_t sm = step_meta[static_cast<size_t>(i)];
        const bool prop = (sm & 0xffu) == OP_PROP, obs0 = ((sm >> 24) & 1u) != 0u && ((sm >> 25) & 3u) == 0u;
        const uint32_t rung = (sm >> 27) & 7u;
        if (prop && obs0 && rung >= 1u && rung <= 4u && (sm >> 31) == 0u) fast |= 1ull << i;
      }
      cp->cls_fast_mask.push_back(fast);
    }
    // Which members share a chunk is free (any G subjects of the class may share a propagator).  `spread`: member j of
    // chunk c is the (c + j * n_chunks)-th subject of the class, so the G rows a block writes at one step are far
    // apart while neighbouring blocks write neighbouring subjects: G slowly advancing write fronts instead of every
    // block covering its own 8-subject region (tools/experiments/store_pattern_probe.hip, rows B vs H).
    const size_t n_chunks_cls = (mem.size() + static_cast<size_t>(G) - 1) / static_cast<size_t>(G);
    std::vector<int32_t> pick(static_cast<size_t>(G));
    for (size_t c = 0; c < n_chunks_cls; ++c) {
      int32_t n = 0;
      for (int32_t j = 0; j < G; ++j) {
        const size_t idx = spread ? (c + static_cast<size_t>(j) * n_chunks_cls) : (c * static_cast<size_t>(G) + static_cast<size_t>(j));
        if (idx < mem.size()) pick[static_cast<size_t>(n++)] = mem[idx];
      }
      cp->chunk_cls.push_back(out_cls);
      cp->chunk_n.push_back(n);
      cp->chunk_val_off.push_back(static_cast<int64_t>(cp->val.size()));
      for (int32_t j = 0; j < G; ++j) {
        cp->chunk_subj.push_back(j < n ? pick[static_cast<size_t>(j)] : -1);
        cp->chunk_row.push_back(j < n ? hp.subj_obs_off[pick[static_cast<size_t>(j)]] : 0);
      }
      const size_t base = cp->val.size();
      cp->val.resize(base + static_cast<size_t>(L) * G, 0.0);
      cp->dtv.resize(base + static_cast<size_t>(L) * G, 0.0);
      if (dyn) {
        cp->facp.resize((base + static_cast<size_t>(L) * G) * nfac, 1.0);
        cp->faco.resize((base + static_cast<size_t>(L) * G) * nfac, 1.0);
      }
      for (int32_t j = 0; j < n; ++j) {
        const int64_t s0 = os.subj_op_off[pick[static_cast<size_t>(j)]];
        for (int64_t i = 0; i < r1 - r0; ++i) {
          const int32_t st = step_of_op[static_cast<size_t>(i)];
          if (dyn) {  // this member's covariate factors at the op: the PROP's rate constants, the observation's volume
            const int32_t so = obs_step_of_op[static_cast<size_t>(i)];
            const double* src = &os.op_fac[static_cast<size_t>(s0 + i) * nfac];
            if (so >= 0)
              std::memcpy(&cp->faco[(base + static_cast<size_t>(so) * G + j) * nfac], src, nfac * sizeof(double));
            else if (st >= 0 && (os.op_meta[s0 + i] & 0xffu) == OP_PROP)
              std::memcpy(&cp->facp[(base + static_cast<size_t>(st) * G + j) * nfac], src, nfac * sizeof(double));
          }
          if (st < 0) continue;
          const uint32_t kind = os.op_meta[s0 + i] & 0xffu;
          double v = 0.0;
          if (kind == OP_BOLUS) v = os.op_a[s0 + i];
          if (kind == OP_PROP) v = os.op_b[s0 + i];
          if (kind == OP_RESET && lagged) v = os.op_a[s0 + i];  // this member's occasion: where its lagged boluses are listed
          cp->val[base + static_cast<size_t>(st) * G + j] = v;
          if (loose && kind == OP_PROP) cp->dtv[base + static_cast<size_t>(st) * G + j] = os.op_a[s0 + i];
        }
      }
      {
        uint64_t mask = 0;
        for (int64_t st = 0; st < L; ++st) {
          bool any = false;
          for (int32_t j = 0; j < n; ++j) any |= cp->val[base + static_cast<size_t>(st) * G + j] != 0.0;
          if (any) mask |= 1ull << (st < 63 ? st : 63);
        }
        if (L > 63) mask |= 1ull << 63;  // (steps past the mask's width always fetch their values)
        cp->chunk_rate_mask.push_back(mask);
      }
      cp->n_classed_subjects += n;
    }
    ++out_cls;
  };
  std::vector<int32_t> leftover;  // members of classes too small to batch: second chance as loose classes
  for (size_t c = 0; c < members.size(); ++c) {
    if (dyn || static_cast<int32_t>(members[c].size()) < min_class_size) {
      leftover.insert(leftover.end(), members[c].begin(), members[c].end());
      continue;
    }
    emit_class(members[c], cls_rep[c], false);
  }
  cp->n_chunks_exact = static_cast<int64_t>(cp->chunk_cls.size());
  if (loose_classes && !leftover.empty()) {
    std::sort(leftover.begin(), leftover.end());
    constexpr uint32_t kShape = 0x00ffffffu;  // kind | io: what a loose class shares (ladder bits and lengths are free)
    auto same_shape = [&](int64_t x, int64_t y) {
      const int64_t x0 = os.subj_op_off[x], x1 = os.subj_op_off[x + 1];
      const int64_t y0 = os.subj_op_off[y], y1 = os.subj_op_off[y + 1];
      if (x1 - x0 != y1 - y0) return false;
      for (int64_t i = 0; i < x1 - x0; ++i)
        if ((os.op_meta[x0 + i] & kShape) != (os.op_meta[y0 + i] & kShape)) return false;
      return true;
    };
    std::unordered_map<uint64_t, std::vector<int32_t>> lbuckets;
    std::vector<int32_t> lrep;
    std::vector<std::vector<int32_t>> lmembers;
    for (int32_t s : leftover) {
      const int64_t o0 = os.subj_op_off[s], o1 = os.subj_op_off[s + 1];
      uint64_t h = static_cast<uint64_t>(o1 - o0);
      for (int64_t o = o0; o < o1; ++o) h = mix64(h, static_cast<uint64_t>(os.op_meta[o] & kShape));
      auto& ids = lbuckets[h];
      int32_t cls = -1;
      for (int32_t c : ids)
        if (same_shape(lrep[c], s)) {
          cls = c;
          break;
        }
      if (cls < 0) {
        cls = static_cast<int32_t>(lrep.size());
        lrep.push_back(s);
        lmembers.emplace_back();
        ids.push_back(cls);
      }
      lmembers[cls].push_back(s);
    }
    // a loose chunk does G members' arithmetic whatever it holds (no propagator to share): below ~3/4 full the
    // generic walker is the cheaper way to serve its subjects
    const int32_t min_loose = std::max(min_class_size, (3 * G + 3) / 4);
    for (size_t c = 0; c < lmembers.size(); ++c) {
      if (static_cast<int32_t>(lmembers[c].size()) < min_loose)
        cp->generic_subjects.insert(cp->generic_subjects.end(), lmembers[c].begin(), lmembers[c].end());
      else
        emit_class(lmembers[c], lrep[c], true);
    }
  } else {
    cp->generic_subjects.insert(cp->generic_subjects.end(), leftover.begin(), leftover.end());
  }
  cp->n_chunks = static_cast<int64_t>(cp->chunk_cls.size());
  cp->chunk_val_off.push_back(static_cast<int64_t>(cp->val.size()));  // sentinel: chunk c's block is [off[c], off[c+1])
  std::sort(cp->generic_subjects.begin(), cp->generic_subjects.end());
}

}  // namespace pmx
