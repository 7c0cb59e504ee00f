"""Deterministic synthetic populations for the BASELINE.json configs.

All draws come from SplitMix64 (seed 0x9E3779B97F4A7C15), generated on the host
and identical for the GPU path and the CPU oracle.  Populations are produced
directly as ``FlatPopulation`` (numpy) — building 100k+ Python ``Subject`` objects
would dominate set-up time.

  C2  two_compartments, 10k subjects x 1 support point, identical 8-event schedule
  C3  two_compartments, 100k subjects x 1000 support points (NPAG-style grid)
  C4  ode one_cmt_iv RK4, 50k subjects, irregular schedules, one theta per subject
  C5  three_compartments_with_absorption + time-varying wt, 200k x 512
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

from . import _abi
from .equation import (Pow, Ratio, Scaled, analytical, bolus, infusion, ode)
from .flatten import FlatPopulation

SEED = 0x9E3779B97F4A7C15
_M64 = (1 << 64) - 1


class SplitMix64:
    """Vectorised SplitMix64: ``uniform(n)`` returns the next n doubles in [0, 1)."""

    def __init__(self, seed: int = SEED):
        self.state = np.uint64(seed & _M64)

    def next_u64(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            inc = np.uint64(0x9E3779B97F4A7C15)
            z = self.state + inc * np.arange(1, n + 1, dtype=np.uint64)
            self.state = self.state + inc * np.uint64(n)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))

    def uniform(self, n: int) -> np.ndarray:
        return (self.next_u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))

    def log_uniform(self, n: int, lo: float, hi: float) -> np.ndarray:
        return np.exp(np.log(lo) + self.uniform(n) * (np.log(hi) - np.log(lo)))


def _shard_seed(seed: int, shard: int) -> int:
    """Seed of shard `shard` of a weak-scaling run (bench.py --gpus N: every rank draws its OWN subjects; shard 0 is the
    single-GPU population)."""
    return (seed ^ (shard * 0xD1B54A32D192ED03)) & _M64


# ---------------------------------------------------------------------------- models
def model_two_cpt_iv():
    """`two_compartments`, out = x[central]/v, theta = [ke, kcp, kpc, v] (examples/analytical_vs_ode.rs:180-197)."""
    return analytical(name="two_cmt_iv", params=["ke", "kcp", "kpc", "v"], structure="two_compartments",
                      states=["central", "peripheral"], outputs=["cp"], routes=[infusion("iv", "central")],
                      out={"cp": Ratio("central", "v")})


def model_one_cmt_iv_ode(h_max: float = 0.02):
    """`ode!` one_cmt_iv (examples/ode_readme.rs:9-23), theta = [ke, v]."""
    return ode(name="one_cmt_iv", params=["ke", "v"], diffeq="one_cmt_iv", states=["central"], outputs=["cp"],
               routes=[infusion("iv", "central")], out={"cp": Ratio("central", "v")}, h_max=h_max)


def model_three_cpt_abs_wt(cov_time: str = "segment_dt"):
    """C5: `three_compartments_with_absorption`, params [ka,k10_0,k12,k13,k21,k31,v], derived k10 = k10_0*(wt/70)^0.75."""
    return analytical(name="three_cmt_oral_wt", params=["ka", "k10_0", "k12", "k13", "k21", "k31", "v"],
                      derived={"k10": Scaled("k10_0", (Pow("wt", 70.0, 0.75),))}, covariates=["wt"],
                      structure="three_compartments_with_absorption",
                      states=["gut", "central", "periph1", "periph2"], outputs=["cp"],
                      routes=[bolus("oral", "gut")], out={"cp": Ratio("central", "v")}, cov_time=cov_time)


_SIG = ("double t, const double* x, const double* p, const double* cov, const double* rateiv, "
        "const double* derived, double* ")
# the bodies of the derive / lag / fa / init / out blocks of the reference's covariate parity model
# (tests/analytical_macro_lowering.rs:236-258), as closure source text (pmx.h "user closures")
USER_COVARIATE_SRC = f"""
PMX_DEVICE void pmx_derive({_SIG}d) {{
  const double wt = cov[COV_wt], renal = cov[COV_renal];
  const double wt_scale = pow(wt / 70.0, 0.75);
  const double renal_scale = pow(renal / 90.0, 0.25);
  d[D_ke] = p[P_ke0] * wt_scale * renal_scale;
  d[D_adjusted_v] = p[P_v] * (wt / 70.0) * (1.0 + 0.001 * (renal - 90.0));
}}
PMX_DEVICE void pmx_route_lag({_SIG}lag) {{
  const double lag_scale = sqrt(cov[COV_wt] / 70.0) * pow(90.0 / cov[COV_renal], 0.1);
  lag[R_oral] = p[P_tlag] * lag_scale;
}}
PMX_DEVICE void pmx_route_bioavailability({_SIG}fa) {{
  const double fa_scale = pow(cov[COV_renal] / 90.0, 0.1);
  fa[R_oral] = fmin(fmax(p[P_f_oral] * fa_scale, 0.0), 1.0);
}}
PMX_DEVICE void pmx_init({_SIG}xi) {{
  xi[X_gut] = p[P_base_gut] + 0.03 * cov[COV_wt];
  xi[X_central] = p[P_base_central] + 0.08 * cov[COV_renal];
}}
PMX_DEVICE void pmx_outputs({_SIG}y) {{ y[Y_cp] = x[X_central] / derived[D_adjusted_v]; }}
"""


def model_user_covariates(cov_time: str = "segment_dt"):
    """macro_covariate_analytical(), tests/analytical_macro_lowering.rs:225-260: lag, fa, init and the output volume are
    functions of (theta, t, covariates) - user closures compiled for the device at run time."""
    return analytical(name="one_cmt_abs_covariates",
                      params=["ka", "ke0", "v", "tlag", "f_oral", "base_gut", "base_central"],
                      derived=["ke", "adjusted_v"], covariates=["wt", "renal"], states=["gut", "central"], outputs=["cp"],
                      routes=[bolus("oral", "gut"), infusion("iv", "central")],
                      structure="one_compartment_with_absorption", source=USER_COVARIATE_SRC, cov_time=cov_time)


def theta_user(n_support: int = 256) -> np.ndarray:
    """log-uniform around the fixture's support point [1.0, 0.16, 32, 0.5, 0.8, 3, 14] (:470-483); ka > ke always."""
    rng = SplitMix64(SEED ^ 0x05E7)
    lo = np.array([0.8, 0.05, 20.0, 0.1, 0.5, 1.0, 5.0])
    hi = np.array([2.5, 0.35, 60.0, 1.0, 1.2, 5.0, 20.0])
    u = rng.uniform(7 * n_support).reshape(n_support, 7)
    return np.exp(np.log(lo) + u * (np.log(hi) - np.log(lo)))


def population_user(n_subjects: int, shard: int = 0) -> FlatPopulation:
    """The fixture's subject (tests/analytical_macro_lowering.rs:35-51: oral bolus at 1 h, 2 h infusion from 6 h, seven
    observations, wt and renal as two-knot lines) varied per subject: doses, recorded sampling times, covariate values."""
    rng = SplitMix64(_shard_seed(SEED ^ 0x0B5E, shard))
    S, E = n_subjects, 9
    obs_t = np.array([0.25, 0.75, 1.5, 3.0, 6.5, 7.0, 8.0])
    scale = 0.5 + rng.uniform(S)
    t = np.tile(np.concatenate([[1.0, 6.0], obs_t]), S).reshape(S, E)
    t[:, 2:] += 0.2 * (rng.uniform(S * 7).reshape(S, 7) - 0.5) * np.array([0.2, 0.2, 0.4, 1.0, 0.4, 0.4, 0.4])
    kind = np.tile(np.array([_abi.PMX_EV_BOLUS, _abi.PMX_EV_INFUSION] + [_abi.PMX_EV_OBSERVATION] * 7, dtype=np.uint8), S)
    v = np.full((S, E), np.nan)
    v[:, 0] = 100.0 * scale
    v[:, 1] = 140.0 * scale
    dur = np.zeros((S, E))
    dur[:, 1] = 2.0
    # covariates per occasion in declaration order (wt, renal), two knots each at 0 and 8 h
    kt = np.tile(np.array([0.0, 8.0, 0.0, 8.0]), S)
    kv = np.empty((S, 4))
    kv[:, 0] = 55.0 + 30.0 * rng.uniform(S)
    kv[:, 1] = kv[:, 0] + 8.0 * (rng.uniform(S) - 0.3)
    kv[:, 2] = 70.0 + 40.0 * rng.uniform(S)
    kv[:, 3] = kv[:, 2] - 25.0 * rng.uniform(S)
    return FlatPopulation(subj_occ_off=np.arange(S + 1), occ_ev_off=np.arange(S + 1) * E,
                          occ_index=np.zeros(S, dtype=np.int32), ev_time=t.reshape(-1), ev_value=v.reshape(-1),
                          ev_duration=dur.reshape(-1), ev_kind=kind, ev_io=np.zeros(S * E, dtype=np.uint16),
                          n_covariates=2, cov_knot_off=np.arange(2 * S + 1) * 2, cov_knot_time=kt,
                          cov_knot_value=kv.reshape(-1), presorted=False)


# ---------------------------------------------------------------------------- populations
_C23_OBS_T = np.array([0.5, 1.0, 2.0, 4.0, 8.0, 12.0, 24.0])


def population_c23(n_subjects: int, ragged: bool = False, shard: int = 0) -> FlatPopulation:
    """C2/C3 schedule: infusion(t=0, amt=500*(1+0.001*(s mod 1000)), dur=0.5) + 7 missing observations.
    ``ragged``: every subject's sampling times are jittered by up to +-10 % (a clinical dataset with recorded
    times instead of protocol times): no two subjects share a design, no step is a multiple of another."""
    S = n_subjects
    E = 8
    s = np.arange(S) + shard * S  # (shard r of a weak-scaling run = subjects [r S, (r + 1) S) of the N x S population)
    t = np.tile(np.concatenate([[0.0], _C23_OBS_T]), S)
    if ragged:
        jit = 1.0 + 0.2 * (SplitMix64(_shard_seed(SEED ^ 0x7A66, shard)).uniform(S * E) - 0.5)
        jit[0::E] = 1.0
        t = t * jit
    v = np.full(S * E, np.nan)
    v[0::E] = 500.0 * (1.0 + 0.001 * (s % 1000))
    dur = np.zeros(S * E)
    dur[0::E] = 0.5
    kind = np.tile(np.array([_abi.PMX_EV_INFUSION] + [_abi.PMX_EV_OBSERVATION] * 7, dtype=np.uint8), S)
    io = np.zeros(S * E, dtype=np.uint16)
    return FlatPopulation(subj_occ_off=np.arange(S + 1), occ_ev_off=np.arange(S + 1) * E,
                          occ_index=np.zeros(S, dtype=np.int32), ev_time=t, ev_value=v, ev_duration=dur,
                          ev_kind=kind, ev_io=io)


def theta_c2() -> np.ndarray:
    return np.array([[0.1, 0.3, 0.2, 50.0]])  # examples/analytical_vs_ode.rs:193-197


def theta_c3(n_support: int = 1000, rng: SplitMix64 = None) -> np.ndarray:
    """log-uniform grid ke in [0.02,0.5], kcp,kpc in [0.01,0.5], v in [10,100]."""
    rng = rng or SplitMix64()
    u = rng.uniform(4 * n_support).reshape(n_support, 4)
    lo = np.array([0.02, 0.01, 0.01, 10.0])
    hi = np.array([0.5, 0.5, 0.5, 100.0])
    return np.exp(np.log(lo) + u * (np.log(hi) - np.log(lo)))


def config_c2(n_subjects: int = 10_000):
    return model_two_cpt_iv(), population_c23(n_subjects), theta_c2()


def config_c3(n_subjects: int = 100_000, n_support: int = 1000):
    return model_two_cpt_iv(), population_c23(n_subjects), theta_c3(n_support)


def config_c4(n_subjects: int = 50_000, h_max: float = 0.02, shard: int = 0) -> Tuple[object, FlatPopulation, np.ndarray]:
    """`ode!` one_cmt_iv, per-subject theta (batch shape), E_s ~ U{6..40}: 1-6 infusions + observations."""
    rng = SplitMix64(_shard_seed(SEED, shard))
    S = n_subjects
    n_ev = 6 + (rng.next_u64(S) % np.uint64(35)).astype(np.int64)  # 6..40
    n_inf = 1 + (rng.next_u64(S) % np.uint64(6)).astype(np.int64)  # 1..6
    n_inf = np.minimum(n_inf, n_ev - 1)
    n_obs = n_ev - n_inf
    tot = int(n_ev.sum())
    off = np.concatenate([[0], np.cumsum(n_ev)])
    u_t = rng.uniform(tot)
    u_d = rng.uniform(tot)
    u_a = rng.uniform(tot)
    # position of every event inside its subject
    subj = np.repeat(np.arange(S), n_ev)
    pos = np.arange(tot) - off[subj]
    is_inf = pos < n_inf[subj]
    t = np.where(is_inf, u_t * 96.0, u_t * 120.0)
    dur = np.where(is_inf, 0.1 + u_d * 3.9, 0.0)
    val = np.where(is_inf, 50.0 + u_a * 450.0, np.nan)
    kind = np.where(is_inf, _abi.PMX_EV_INFUSION, _abi.PMX_EV_OBSERVATION).astype(np.uint8)
    flat = FlatPopulation(subj_occ_off=np.arange(S + 1), occ_ev_off=off, occ_index=np.zeros(S, dtype=np.int32),
                          ev_time=t, ev_value=val, ev_duration=dur, ev_kind=kind, ev_io=np.zeros(tot, dtype=np.uint16),
                          presorted=False)
    theta = np.stack([rng.log_uniform(S, 0.05, 1.5), rng.log_uniform(S, 20.0, 300.0)], axis=1)
    return model_one_cmt_iv_ode(h_max), flat, theta


def _cubic_q(k10, k12, k13, k21, k31):
    a = k10 + k12 + k13 + k21 + k31
    b = k10 * k21 + k13 * k21 + k10 * k31 + k12 * k31 + k21 * k31
    c = k10 * k21 * k31
    m = (3.0 * b - a * a) / 3.0
    n = (2.0 * a ** 3 - 9.0 * a * b + 27.0 * c) / 27.0
    return n * n / 4.0 + m ** 3 / 27.0, n


def theta_c5(n_support: int = 512, rng: SplitMix64 = None) -> np.ndarray:
    """log-uniform ka[0.5,3] k10_0[0.05,0.5] k12,k13[0.1,3] k21,k31[0.1,2] v[10,100]; reject draws whose cubic
    discriminant is not safely negative at either end of the wt range (k10 scales by 0.78..1.40)."""
    rng = rng or SplitMix64(SEED ^ 0xC5)
    lo = np.array([0.5, 0.05, 0.1, 0.1, 0.1, 0.1, 10.0])
    hi = np.array([3.0, 0.5, 3.0, 3.0, 2.0, 2.0, 100.0])
    out = []
    while len(out) < n_support:
        u = rng.uniform(7 * n_support).reshape(n_support, 7)
        th = np.exp(np.log(lo) + u * (np.log(hi) - np.log(lo)))
        ok = np.ones(n_support, dtype=bool)
        for scale in (0.70, 0.78, 1.0, 1.40, 1.45):
            q, n = _cubic_q(th[:, 1] * scale, th[:, 2], th[:, 3], th[:, 4], th[:, 5])
            ok &= q < -1e-9 * np.abs(n * n / 4.0)
        out.extend(th[ok])
    return np.asarray(out[:n_support])


def population_c5(n_subjects: int, rng: SplitMix64 = None, constant_wt: bool = False, shard: int = 0) -> FlatPopulation:
    """oral bolus(0, 100..500) q24h x3 + 10 observations; `wt`: 2-4 linear knots per subject in [50,110] kg
    (``constant_wt``: one knot, i.e. a subject-constant covariate, the usual allometric-scaling case)."""
    rng = rng or SplitMix64(_shard_seed(SEED ^ 0x5C, shard))
    S = n_subjects
    E = 13
    obs_t = np.array([1.0, 2.0, 4.0, 8.0, 12.0, 23.5, 26.0, 36.0, 50.0, 72.0])
    amt = 100.0 + 400.0 * rng.uniform(S)
    times = np.concatenate([[0.0, 24.0, 48.0], obs_t])
    kinds = np.array([_abi.PMX_EV_BOLUS] * 3 + [_abi.PMX_EV_OBSERVATION] * 10, dtype=np.uint8)
    t = np.tile(times, S)
    kind = np.tile(kinds, S)
    v = np.full(S * E, np.nan)
    for j in range(3):
        v[j::E] = amt
    nk = 2 + (rng.next_u64(S) % np.uint64(3)).astype(np.int64)  # 2..4 knots
    if constant_wt:
        nk = np.ones(S, dtype=np.int64)
    koff = np.concatenate([[0], np.cumsum(nk)])
    tot = int(nk.sum())
    ksub = np.repeat(np.arange(S), nk)
    kpos = np.arange(tot) - koff[ksub]
    # knots spread over [0, 72] h: first at 0, the rest at increasing random times
    kt = np.where(kpos == 0, 0.0, (kpos + rng.uniform(tot) * 0.9) * (72.0 / 4.0))
    kv = 50.0 + 60.0 * rng.uniform(tot)
    return FlatPopulation(subj_occ_off=np.arange(S + 1), occ_ev_off=np.arange(S + 1) * E,
                          occ_index=np.zeros(S, dtype=np.int32), ev_time=t, ev_value=v, ev_duration=np.zeros(S * E),
                          ev_kind=kind, ev_io=np.zeros(S * E, dtype=np.uint16), n_covariates=1, cov_knot_off=koff,
                          cov_knot_time=kt, cov_knot_value=kv, presorted=False)


def config_c5(n_subjects: int = 200_000, n_support: int = 512, cov_time: str = "segment_dt"):
    return model_three_cpt_abs_wt(cov_time), population_c5(n_subjects), theta_c5(n_support)
