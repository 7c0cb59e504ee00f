"""Host-side data model: ``Data -> Subject -> Occasion -> Event``.

Python mirror of the reference's input types so that callers (and the parity
tests) read like pharmsol code:

* ``Subject.builder(id).bolus(..).infusion(..).observation(..).covariate(..).repeat(..).reset().build()``
  — src/data/builder.rs:38-50,113-361
* ``Event`` = ``Bolus | Infusion | Observation`` — src/data/event.rs:107-114,354-359,445-451,575-582
* sort order: time (total order) then Observation < Bolus < Infusion, stable — src/data/event.rs:292-304
* ``Covariates``: named knot lists per occasion — src/data/covariate.rs:297-299

Observations also carry what the likelihood consumes: an optional per-observation ``ErrorPoly`` and the
``Censor`` state (src/data/event.rs:557-582).
"""
from __future__ import annotations

import math
import struct
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

from ._abi import PMX_CENSOR_ALOQ, PMX_CENSOR_BLOQ, PMX_CENSOR_NONE, PMX_EV_BOLUS, PMX_EV_INFUSION, PMX_EV_OBSERVATION

Label = Union[str, int]


def _total_key(x: float) -> int:
    """Sort key equal to Rust's ``f64::total_cmp`` ordering."""
    (b,) = struct.unpack("<q", struct.pack("<d", float(x)))
    return b ^ (((b >> 63) & 0xFFFFFFFFFFFFFFFF) >> 1)


@dataclass
class Bolus:
    """Instantaneous dose (src/data/event.rs:354-359)."""

    time: float
    amount: float
    input: Label
    occasion: int = 0
    kind = PMX_EV_BOLUS


@dataclass
class Infusion:
    """Constant-rate dose over ``duration`` (src/data/event.rs:445-451)."""

    time: float
    amount: float
    input: Label
    duration: float
    occasion: int = 0
    kind = PMX_EV_INFUSION


class Censor:
    """``Censor`` (src/data/event.rs:557-567): how the likelihood reads the observed value."""

    NONE = PMX_CENSOR_NONE  # lognormpdf
    BLOQ = PMX_CENSOR_BLOQ  # below the limit of quantification: log CDF
    ALOQ = PMX_CENSOR_ALOQ  # above it: log survival function


@dataclass
class Observation:
    """Observation slot; ``value is None`` = prediction-only (src/data/event.rs:575-582).  ``errorpoly`` =
    (c0, c1, c2, c3) overriding the error model's polynomial for this observation."""

    time: float
    value: Optional[float]
    outeq: Label
    occasion: int = 0
    errorpoly: Optional[Tuple[float, float, float, float]] = None
    censoring: int = PMX_CENSOR_NONE
    kind = PMX_EV_OBSERVATION


Event = Union[Bolus, Infusion, Observation]


def _event_sort_key(ev: Event) -> Tuple[int, int]:
    # Event::cmp_time_then_type (event.rs:292-304); list.sort is stable like sort_by.
    return (_total_key(ev.time), ev.kind)


@dataclass
class Covariates:
    """Named covariate observations of one occasion (src/data/covariate.rs:297-299)."""

    knots: Dict[str, List[Tuple[float, float]]] = field(default_factory=dict)
    fixed: Dict[str, bool] = field(default_factory=dict)

    def add_observation(self, name: str, time: float, value: float) -> None:
        # Covariate::add_observation (covariate.rs:143-154): a value at an existing time replaces it
        kn = self.knots.setdefault(name, [])
        for i, (t, _) in enumerate(kn):
            if t == float(time):
                kn[i] = (t, float(value))
                return
        kn.append((float(time), float(value)))

    def set_fixed(self, name: str, fixed: bool = True) -> None:
        self.fixed[name] = fixed

    def names(self) -> List[str]:
        return sorted(self.knots)  # BTreeMap iteration order


@dataclass
class Occasion:
    """One dosing/observation block; state resets at its start (src/data/structs.rs:556-560)."""

    index: int
    events: List[Event] = field(default_factory=list)
    covariates: Covariates = field(default_factory=Covariates)

    def add_event(self, ev: Event) -> None:
        # Occasion::add_event pushes then re-sorts (structs.rs:713-716)
        self.events.append(ev)
        self.sort()

    def sort(self) -> None:
        self.events.sort(key=_event_sort_key)


class Subject:
    """A subject = id + occasions (src/data/structs.rs:352-355)."""

    def __init__(self, id: str, occasions: List[Occasion]):
        self.id = id
        self.occasions = occasions
        for occ in self.occasions:  # Subject::new sorts every occasion (structs.rs:363-369)
            occ.sort()

    @staticmethod
    def builder(id: str) -> "SubjectBuilder":
        return SubjectBuilder(id)

    def n_events(self) -> int:
        return sum(len(o.events) for o in self.occasions)

    def n_observations(self) -> int:
        return sum(1 for o in self.occasions for e in o.events if isinstance(e, Observation))


class SubjectBuilder:
    """Fluent builder (src/data/builder.rs:84-361)."""

    def __init__(self, id: str):
        self._id = id
        self._occasions: List[Occasion] = []
        self._current = Occasion(0)
        self._covariates = Covariates()
        self._last: Optional[Event] = None

    def event(self, ev: Event) -> "SubjectBuilder":
        self._last = ev
        self._current.add_event(ev)
        return self

    def bolus(self, time: float, amount: float, input: Label) -> "SubjectBuilder":
        return self.event(Bolus(float(time), float(amount), input, self._current.index))

    def infusion(self, time: float, amount: float, input: Label, duration: float) -> "SubjectBuilder":
        return self.event(Infusion(float(time), float(amount), input, float(duration), self._current.index))

    def observation(self, time: float, value: float, outeq: Label) -> "SubjectBuilder":
        return self.event(Observation(float(time), float(value), outeq, self._current.index))

    def missing_observation(self, time: float, outeq: Label) -> "SubjectBuilder":
        return self.event(Observation(float(time), None, outeq, self._current.index))

    def censored_observation(self, time: float, value: float, outeq: Label, censoring: int) -> "SubjectBuilder":
        # builder.rs:161-178
        return self.event(Observation(float(time), float(value), outeq, self._current.index, None, int(censoring)))

    def observation_with_error(self, time: float, value: float, outeq: Label, errorpoly, censored: int = PMX_CENSOR_NONE
                               ) -> "SubjectBuilder":
        # builder.rs:211-229; errorpoly = ErrorPoly or (c0, c1, c2, c3)
        poly = tuple(errorpoly) if not hasattr(errorpoly, "c0") else (errorpoly.c0, errorpoly.c1, errorpoly.c2, errorpoly.c3)
        return self.event(Observation(float(time), float(value), outeq, self._current.index,
                                      tuple(float(c) for c in poly), int(censored)))

    def repeat(self, n: int, delta: float) -> "SubjectBuilder":
        # builder.rs:251-313: clones of the LAST added event at time + delta*i
        last = self._last
        if last is None:
            return self
        for i in range(1, n + 1):
            t = last.time + delta * float(i)
            if isinstance(last, Bolus):
                self.bolus(t, last.amount, last.input)
            elif isinstance(last, Infusion):
                self.infusion(t, last.amount, last.input, last.duration)
            else:  # value, censoring and error polynomial are kept (builder.rs:253-254)
                self.event(Observation(t, last.value, last.outeq, self._current.index, last.errorpoly, last.censoring))
        return self

    def covariate(self, name: str, time: float, value: float) -> "SubjectBuilder":
        self._covariates.add_observation(name, time, value)
        return self

    def reset(self) -> "SubjectBuilder":
        # builder.rs:315-326: close the occasion, attach the covariates collected so far
        nxt = self._current.index + 1
        self._current.sort()
        self._current.covariates = self._covariates
        self._occasions.append(self._current)
        self._current = Occasion(nxt)
        self._covariates = Covariates()
        self._last = None
        return self

    def build(self) -> Subject:
        self.reset()
        return Subject(self._id, self._occasions)


class Data:
    """A population (src/data/structs.rs:38)."""

    def __init__(self, subjects: Sequence[Subject]):
        self.subjects = list(subjects)

    def __len__(self) -> int:
        return len(self.subjects)

    def __iter__(self):
        return iter(self.subjects)

    @staticmethod
    def from_pmetrics_csv_bytes(data: bytes) -> "Data":
        """``Data::from_pmetrics_csv_bytes`` (src/data/parser/pmetrics/mod.rs:170-232)."""
        from .pmetrics import from_pmetrics_csv_bytes

        return from_pmetrics_csv_bytes(data)


def interpolate(knots: Sequence[Tuple[float, float]], t: float, fixed: bool = False) -> float:
    """``Covariate::interpolate`` (src/data/covariate.rs:189-241) on the host.

    Used only by host-side utilities; the device path receives raw knots and the
    C++ compiler in csrc/ evaluates them the same way.
    """
    obs = sorted(knots, key=lambda kv: _total_key(kv[0]))
    if not obs:
        raise ValueError("MissingSegments")
    n = len(obs)
    for i, (t0, v0) in enumerate(obs):
        t1 = obs[i + 1][0] if i + 1 < n else math.inf
        if t0 <= t and (i + 1 == n or t < t1):
            if fixed or i + 1 == n:
                return v0
            slope = (obs[i + 1][1] - v0) / (t1 - t0)
            intercept = v0 - slope * t0
            return slope * t + intercept
    if t < obs[0][0]:
        return obs[0][1]
    return obs[-1][1]
