"""Flatten ``Data`` into the structure-of-arrays the C ABI takes (``pmx_population_desc``).

This is the label-resolution half of ``EquationPriv::resolve_occasion_events``
(src/simulator/equation/mod.rs:247-273, labels -> dense indices via the model's
routes/outputs, src/simulator/equation/metadata.rs:248-275) done ONCE per
population instead of once per (subject, support point).  Sorting, sub-segment
splitting and covariate evaluation stay inside the native library
(csrc/pmx_compile.cpp).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _abi
from .data import Bolus, Data, Infusion, Observation, Subject


class FlatPopulation:
    """Numpy-backed ``pmx_population_desc`` (arrays are kept alive by this object)."""

    def __init__(self, *, subj_occ_off, occ_ev_off, occ_index, ev_time, ev_value, ev_duration, ev_kind, ev_io,
                 n_covariates=0, cov_knot_off=None, cov_knot_time=None, cov_knot_value=None, cov_fixed=None,
                 presorted=False, subject_ids: Optional[List[str]] = None, ev_errorpoly=None, ev_censor=None):
        self.subj_occ_off = np.ascontiguousarray(subj_occ_off, dtype=np.int64)
        self.occ_ev_off = np.ascontiguousarray(occ_ev_off, dtype=np.int64)
        self.occ_index = np.ascontiguousarray(occ_index, dtype=np.int32)
        self.ev_time = np.ascontiguousarray(ev_time, dtype=np.float64)
        self.ev_value = np.ascontiguousarray(ev_value, dtype=np.float64)
        self.ev_duration = np.ascontiguousarray(ev_duration, dtype=np.float64)
        self.ev_kind = np.ascontiguousarray(ev_kind, dtype=np.uint8)
        self.ev_io = np.ascontiguousarray(ev_io, dtype=np.uint16)
        self.n_covariates = int(n_covariates)
        self.presorted = bool(presorted)
        self.subject_ids = subject_ids
        n_occ = self.occ_ev_off.shape[0] - 1
        if self.n_covariates > 0:
            self.cov_knot_off = np.ascontiguousarray(cov_knot_off, dtype=np.int64)
            self.cov_knot_time = np.ascontiguousarray(cov_knot_time, dtype=np.float64)
            self.cov_knot_value = np.ascontiguousarray(cov_knot_value, dtype=np.float64)
            self.cov_fixed = (np.zeros(n_occ * self.n_covariates, dtype=np.uint8) if cov_fixed is None else
                              np.ascontiguousarray(cov_fixed, dtype=np.uint8))
            assert self.cov_knot_off.shape[0] == n_occ * self.n_covariates + 1
        else:
            self.cov_knot_off = self.cov_knot_time = self.cov_knot_value = self.cov_fixed = None
        n_ev = self.ev_time.shape[0]
        # likelihood-only attributes of observations (None = absent everywhere)
        self.ev_errorpoly = None if ev_errorpoly is None else np.ascontiguousarray(ev_errorpoly, dtype=np.float64).reshape(n_ev, 4)
        self.ev_censor = None if ev_censor is None else np.ascontiguousarray(ev_censor, dtype=np.int8)
        assert self.ev_censor is None or self.ev_censor.shape[0] == n_ev
        for a in (self.ev_value, self.ev_duration, self.ev_kind, self.ev_io):
            assert a.shape[0] == n_ev
        assert self.occ_ev_off[-1] == n_ev and self.subj_occ_off[-1] == n_occ

    # -- sizes ---------------------------------------------------------------
    @property
    def n_subjects(self) -> int:
        return int(self.subj_occ_off.shape[0] - 1)

    @property
    def n_occasions(self) -> int:
        return int(self.occ_ev_off.shape[0] - 1)

    @property
    def n_events(self) -> int:
        return int(self.ev_time.shape[0])

    @property
    def n_observations(self) -> int:
        return int((self.ev_kind == _abi.PMX_EV_OBSERVATION).sum())

    def events_per_subject(self) -> np.ndarray:
        ev_off = self.occ_ev_off[self.subj_occ_off]
        return np.diff(ev_off)

    def observation_offsets(self) -> np.ndarray:
        """First prediction row of every subject (+ total), event order."""
        is_obs = (self.ev_kind == _abi.PMX_EV_OBSERVATION).astype(np.int64)
        csum = np.concatenate([[0], np.cumsum(is_obs)])
        ev_off = self.occ_ev_off[self.subj_occ_off]
        return csum[ev_off]

    # -- C view --------------------------------------------------------------
    def desc(self) -> _abi.pmx_population_desc:
        d = _abi.pmx_population_desc()
        d.n_subjects = self.n_subjects
        d.n_occasions = self.n_occasions
        d.n_events = self.n_events

        def p(a, ct):
            return a.ctypes.data_as(C.POINTER(ct)) if a is not None and a.size > 0 else C.cast(None, C.POINTER(ct))

        d.subj_occ_off = p(self.subj_occ_off, C.c_int64)
        d.occ_ev_off = p(self.occ_ev_off, C.c_int64)
        d.occ_index = p(self.occ_index, C.c_int32)
        d.ev_time = p(self.ev_time, C.c_double)
        d.ev_value = p(self.ev_value, C.c_double)
        d.ev_duration = p(self.ev_duration, C.c_double)
        d.ev_kind = p(self.ev_kind, C.c_uint8)
        d.ev_io = p(self.ev_io, C.c_uint16)
        d.n_covariates = self.n_covariates
        d.presorted = 1 if self.presorted else 0
        d.cov_knot_off = p(self.cov_knot_off, C.c_int64)
        d.cov_knot_time = p(self.cov_knot_time, C.c_double)
        d.cov_knot_value = p(self.cov_knot_value, C.c_double)
        d.cov_fixed = p(self.cov_fixed, C.c_uint8)
        d.ev_errorpoly = p(self.ev_errorpoly, C.c_double)
        d.ev_censor = p(self.ev_censor, C.c_int8)
        return d

    # -- sharding (partition subjects, replicate theta) -----------
    def subject_slice(self, s0: int, s1: int) -> "FlatPopulation":
        """Contiguous sub-population [s0, s1)."""
        o0, o1 = int(self.subj_occ_off[s0]), int(self.subj_occ_off[s1])
        e0, e1 = int(self.occ_ev_off[o0]), int(self.occ_ev_off[o1])
        kw = dict(
            subj_occ_off=self.subj_occ_off[s0:s1 + 1] - o0,
            occ_ev_off=self.occ_ev_off[o0:o1 + 1] - e0,
            occ_index=self.occ_index[o0:o1],
            ev_time=self.ev_time[e0:e1], ev_value=self.ev_value[e0:e1], ev_duration=self.ev_duration[e0:e1],
            ev_kind=self.ev_kind[e0:e1], ev_io=self.ev_io[e0:e1],
            n_covariates=self.n_covariates, presorted=self.presorted,
            subject_ids=None if self.subject_ids is None else self.subject_ids[s0:s1],
            ev_errorpoly=None if self.ev_errorpoly is None else self.ev_errorpoly[e0:e1],
            ev_censor=None if self.ev_censor is None else self.ev_censor[e0:e1],
        )
        if self.n_covariates > 0:
            nc = self.n_covariates
            k0, k1 = int(self.cov_knot_off[o0 * nc]), int(self.cov_knot_off[o1 * nc])
            kw.update(cov_knot_off=self.cov_knot_off[o0 * nc:o1 * nc + 1] - k0,
                      cov_knot_time=self.cov_knot_time[k0:k1], cov_knot_value=self.cov_knot_value[k0:k1],
                      cov_fixed=self.cov_fixed[o0 * nc:o1 * nc])
        return FlatPopulation(**kw)


def flatten(model, data) -> FlatPopulation:
    """Resolve labels through ``model`` and flatten ``data`` (a ``Data``, a list of
    subjects or one ``Subject``)."""
    if isinstance(data, Subject):
        subjects = [data]
    elif isinstance(data, Data):
        subjects = data.subjects
    else:
        subjects = list(data)
    cov_names = list(model.covariates)
    nc = len(cov_names)
    subj_occ_off = [0]
    occ_ev_off = [0]
    occ_index: List[int] = []
    t: List[float] = []
    v: List[float] = []
    dur: List[float] = []
    kind: List[int] = []
    io: List[int] = []
    poly: List[tuple] = []
    cens: List[int] = []
    any_poly = any_cens = False
    nanpoly = (float("nan"),) * 4
    knot_off = [0]
    knot_t: List[float] = []
    knot_v: List[float] = []
    fixed: List[int] = []
    for subj in subjects:
        for occ in subj.occasions:
            occ_index.append(occ.index)
            for ev in occ.events:
                t.append(ev.time)
                kind.append(ev.kind)
                if isinstance(ev, Observation) and ev.errorpoly is not None:
                    poly.append(tuple(ev.errorpoly))
                    any_poly = True
                else:
                    poly.append(nanpoly)
                cz = ev.censoring if isinstance(ev, Observation) else 0
                cens.append(cz)
                any_cens = any_cens or cz != 0
                if isinstance(ev, Bolus):
                    v.append(ev.amount)
                    dur.append(0.0)
                    io.append(model.resolve_input_label(ev.input, "bolus"))
                elif isinstance(ev, Infusion):
                    v.append(ev.amount)
                    dur.append(ev.duration)
                    io.append(model.resolve_input_label(ev.input, "infusion"))
                else:
                    v.append(float("nan") if ev.value is None else ev.value)
                    dur.append(0.0)
                    io.append(model.resolve_output_label(ev.outeq))
            occ_ev_off.append(len(t))
            for name in cov_names:
                kn = occ.covariates.knots.get(name)
                if not kn:
                    # fetch_cov! panics "Covariate {} not found" (src/lib.rs:433-443)
                    raise KeyError(f"Covariate {name} not found for subject {subj.id} occasion {occ.index}")
                for (kt, kv) in kn:
                    knot_t.append(kt)
                    knot_v.append(kv)
                knot_off.append(len(knot_t))
                fixed.append(1 if occ.covariates.fixed.get(name, False) else 0)
        subj_occ_off.append(len(occ_index))
    return FlatPopulation(
        subj_occ_off=subj_occ_off, occ_ev_off=occ_ev_off, occ_index=occ_index, ev_time=t, ev_value=v,
        ev_duration=dur, ev_kind=kind, ev_io=io, n_covariates=nc,
        cov_knot_off=knot_off if nc else None, cov_knot_time=knot_t if nc else None,
        cov_knot_value=knot_v if nc else None, cov_fixed=fixed if nc else None,
        presorted=False, subject_ids=[s.id for s in subjects],
        ev_errorpoly=np.asarray(poly, dtype=np.float64).reshape(len(t), 4) if any_poly else None,
        ev_censor=cens if any_cens else None)
