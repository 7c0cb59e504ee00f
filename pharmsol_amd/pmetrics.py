"""Pmetrics CSV ingest: rows -> ``Data`` (the data format on the input side of the path).

Mirror of the reference's reader (``src/data/parser/pmetrics/mod.rs:26-232``: header rules, ``Row`` field parsing,
``OUT=-99`` = missing) and of its row ingestion (``row.rs:143-376``: validation, EVID 0/1/4, bolus vs infusion by
``DUR``, ``ADDL``/``II`` expansion in both directions; ``row.rs:593-674`` ``build_data``: subjects by ID, occasions
split at ``EVID=4``, covariates per occasion with a trailing ``!`` selecting carry-forward, subjects sorted by ID).
``INPUT`` / ``OUTEQ`` stay labels (strings); the model resolves them when the population is flattened
(``Equation.flatten``), as in the reference.  Export (``to_pmetrics_csv_bytes``) is not part of this build.
"""
from __future__ import annotations

import csv
import io
import math
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Tuple

from .data import Bolus, Censor, Covariates, Data, Event, Infusion, Observation, Occasion, Subject

CORE_HEADERS = ["ID", "EVID", "TIME", "DUR", "DOSE", "ADDL", "II", "INPUT", "OUT", "OUTEQ", "CENS", "C0", "C1", "C2",
                "C3"]  # CoreColumn::ALL, mod.rs:26-60
REQUIRED_HEADERS = ["ID", "EVID", "TIME"]
_I64_MIN, _I64_MAX = -(2 ** 63), 2 ** 63 - 1
_MAX_ADDL = 10 ** 8  # stand-in for Vec::try_reserve failing ("too large to expand", row.rs:332-345)


class DataError(Exception):
    """``DataError`` (row.rs:676-715); ``kind`` = the variant name."""

    kind = "DataError"

    def __init__(self, message: str, **fields):
        super().__init__(message)
        self.message = message
        self.fields = fields


def _variant(name: str):
    return type(name, (DataError,), {"kind": name})


CSVError = _variant("CSVError")
UnknownEvid = _variant("UnknownEvid")
MissingObservationOuteq = _variant("MissingObservationOuteq")
MissingInfusionDose = _variant("MissingInfusionDose")
MissingInfusionDur = _variant("MissingInfusionDur")
MissingBolusDose = _variant("MissingBolusDose")
MissingBolusInput = _variant("MissingBolusInput")
NonFiniteValue = _variant("NonFiniteValue")
InvalidDataRow = _variant("InvalidDataRow")
InvalidPmetricsData = _variant("InvalidPmetricsData")


def _fmt(x: float) -> str:
    """Rust's ``{}`` for f64 (``0`` not ``0.0``)."""
    if math.isfinite(x) and x == int(x) and abs(x) < 1e16:
        return str(int(x))
    return repr(float(x))


@dataclass
class DataRow:
    """One row-shaped record (row.rs:86-118)."""

    id: str
    time: float
    evid: int = 0
    dose: Optional[float] = None
    dur: Optional[float] = None
    addl: Optional[int] = None
    ii: Optional[float] = None
    input: Optional[str] = None
    out: Optional[float] = None
    outeq: Optional[str] = None
    cens: Optional[int] = None
    c0: Optional[float] = None
    c1: Optional[float] = None
    c2: Optional[float] = None
    c3: Optional[float] = None
    covariates: Dict[str, float] = field(default_factory=dict)

    @staticmethod
    def builder(id: str, time: float) -> "DataRowBuilder":
        return DataRowBuilder(id, time)

    # row.rs:143-212
    def validate(self) -> None:
        if self.id == "":
            raise InvalidDataRow("subject ID cannot be empty")
        self._finite(self.time, "TIME")
        for name, v in (("DUR", self.dur), ("DOSE", self.dose), ("II", self.ii), ("OUT", self.out), ("C0", self.c0),
                        ("C1", self.c1), ("C2", self.c2), ("C3", self.c3)):
            if v is not None:
                self._finite(v, name)
        for name, v in self.covariates.items():
            self._finite(v, name)
        present = sum(c is not None for c in (self.c0, self.c1, self.c2, self.c3))
        if present not in (0, 4):
            raise InvalidDataRow(f"partial error polynomial for {self.id} at time {_fmt(self.time)}")
        if self.addl is not None and self.addl != 0:
            if self.evid not in (1, 4):
                raise InvalidDataRow(f"nonzero ADDL for {self.id} at time {_fmt(self.time)} requires a dose row")
            if not (self.ii is not None and self.ii > 0.0):
                raise InvalidDataRow(f"nonzero ADDL for {self.id} at time {_fmt(self.time)} requires a positive II")
        if self.evid == 4 and (self.dose is None or self.input is None):
            raise InvalidDataRow(f"EVID=4 row for {self.id} at time {_fmt(self.time)} must contain a dose and INPUT")
        if self.evid in (1, 4) and self.dur is not None and self.dur < 0.0:
            raise InvalidDataRow(f"dose row for {self.id} at time {_fmt(self.time)} contains a negative duration")
        if self.evid not in (0, 1, 4):
            raise UnknownEvid(f"Unsupported EVID={self.evid} for subject {self.id} at time {_fmt(self.time)}",
                              evid=self.evid, id=self.id, time=self.time)

    def _finite(self, v: float, name: str) -> None:
        if not math.isfinite(v):
            raise NonFiniteValue(f"Nonfinite value in {name} for {self.id}", field=name, id=self.id)

    def errorpoly(self) -> Optional[Tuple[float, float, float, float]]:
        if None in (self.c0, self.c1, self.c2, self.c3):
            return None
        return (self.c0, self.c1, self.c2, self.c3)

    # row.rs:269-376
    def into_events(self) -> List[Event]:
        self.validate()
        events: List[Event] = []
        if self.evid == 0:
            if self.outeq is None:
                raise MissingObservationOuteq(f"Observation OUTEQ is missing for {self.id} at time {_fmt(self.time)}")
            events.append(Observation(self.time, self.out, self.outeq, 0, self.errorpoly(),
                                      Censor.NONE if self.cens is None else self.cens))
            return events
        if self.input is None:
            raise MissingBolusInput(f"Bolus input label (INPUT) is missing for {self.id} at time {_fmt(self.time)}")
        if (self.dur or 0.0) > 0.0:
            if self.dose is None:
                raise MissingInfusionDose(f"Infusion amount (DOSE) is missing for {self.id} at time {_fmt(self.time)}")
            base: Event = Infusion(self.time, self.dose, self.input, self.dur, 0)
        else:
            if self.dose is None:
                raise MissingBolusDose(f"Bolus amount (DOSE) is missing for {self.id} at time {_fmt(self.time)}")
            base = Bolus(self.time, self.dose, self.input, 0)
        if self.addl is not None and self.ii is not None and self.addl != 0:
            too_large = InvalidDataRow(f"ADDL for {self.id} at time {_fmt(self.time)} is too large to expand")
            if self.addl == _I64_MIN or abs(self.addl) > _MAX_ADDL:
                raise too_large
            interval = abs(self.ii)
            direction = 1.0 if self.addl > 0 else -1.0
            for rep in range(1, abs(self.addl) + 1):  # the additional doses come first, then the row's own (row.rs:347-364)
                offset = direction * interval * float(rep)
                if not math.isfinite(base.time + offset):
                    raise NonFiniteValue(f"Nonfinite value in expanded TIME for {self.id}", field="expanded TIME", id=self.id)
                if isinstance(base, Infusion):
                    events.append(Infusion(base.time + offset, base.amount, base.input, base.duration, 0))
                else:
                    events.append(Bolus(base.time + offset, base.amount, base.input, 0))
        events.append(base)
        return events

    def is_occasion_reset(self) -> bool:
        return self.evid == 4


class DataRowBuilder:
    """``DataRowBuilder`` (row.rs:420-560)."""

    def __init__(self, id: str, time: float):
        self._row = DataRow(str(id), float(time))

    def evid(self, evid: int) -> "DataRowBuilder":
        self._row.evid = int(evid)
        return self

    def dose(self, dose: float) -> "DataRowBuilder":
        self._row.dose = float(dose)
        return self

    def dur(self, dur: float) -> "DataRowBuilder":
        self._row.dur = float(dur)
        return self

    def addl(self, addl: int) -> "DataRowBuilder":
        self._row.addl = int(addl)
        return self

    def ii(self, ii: float) -> "DataRowBuilder":
        self._row.ii = float(ii)
        return self

    def input(self, input) -> "DataRowBuilder":
        self._row.input = str(input)
        return self

    def out(self, out: float) -> "DataRowBuilder":
        self._row.out = float(out)
        return self

    def outeq(self, outeq) -> "DataRowBuilder":
        self._row.outeq = str(outeq)
        return self

    def cens(self, cens: int) -> "DataRowBuilder":
        self._row.cens = int(cens)
        return self

    def error_poly(self, c0: float, c1: float, c2: float, c3: float) -> "DataRowBuilder":
        self._row.c0, self._row.c1, self._row.c2, self._row.c3 = float(c0), float(c1), float(c2), float(c3)
        return self

    def covariate(self, name: str, value: float) -> "DataRowBuilder":
        self._row.covariates[name] = float(value)
        return self

    def build(self) -> DataRow:
        return self._row


def build_data(rows: Iterable[DataRow]) -> Data:
    """``build_data`` (row.rs:593-674)."""
    by_id: Dict[str, List[DataRow]] = {}
    for row in rows:
        by_id.setdefault(row.id, []).append(row)
    subjects: List[Subject] = []
    for sid, srows in by_id.items():
        splits = [i for i, r in enumerate(srows) if r.evid == 4]
        blocks: List[List[DataRow]] = []
        start = 0
        for sp in splits:
            if start < sp:
                blocks.append(srows[start:sp])
            start = sp
        if start < len(srows):
            blocks.append(srows[start:])
        occasions: List[Occasion] = []
        for bi, block in enumerate(blocks):
            events: List[Event] = []
            observed: Dict[str, List[Tuple[float, float]]] = {}
            for row in block:
                events.extend(row.into_events())
                for name, value in row.covariates.items():
                    obs = observed.setdefault(name, [])
                    same = [v for (t, v) in obs if t == row.time]
                    if same:
                        if same[0] != value:
                            raise InvalidDataRow(f"conflicting covariate `{name}` values for subject `{sid}` occasion {bi} "
                                                 f"at time {_fmt(row.time)}")
                    else:
                        obs.append((row.time, value))
            for ev in events:
                ev.occasion = bi
            cov = Covariates()  # Covariates::from_row_observations, covariate.rs:316-333
            for key, obs in observed.items():
                fixed = key.endswith("!")
                name = key[:-1] if fixed else key
                for (t, v) in obs:
                    cov.add_observation(name, t, v)
                cov.set_fixed(name, fixed)
            occ = Occasion(bi, events, cov)
            occ.sort()
            occasions.append(occ)
        subjects.append(Subject(sid, occasions))
    subjects.sort(key=lambda s: s.id.encode("utf-8"))  # a.id().cmp(b.id()): byte order
    return Data(subjects)


# --------------------------------------------------------------------------- CSV layer (mod.rs:165-232, 258-420)
def _core_of(header: str) -> Optional[str]:
    for h in CORE_HEADERS:
        if h.lower() == header.lower() and header.isascii():  # eq_ignore_ascii_case
            return h
    return None


def _validate_covariate_header(header: str) -> None:
    base = header[:-1] if header.endswith("!") else header
    if base == "" or "!" in base or any(ord(ch) < 32 or 127 <= ord(ch) < 160 for ch in base) or _core_of(base) is not None:
        raise InvalidPmetricsData(f"reserved or ambiguous covariate column `{header}`")


def _opt(s: str) -> Optional[str]:
    return None if s in ("", ".", "NA") else s


def _parse_f64(s: str, what: str) -> float:
    # f64::from_str: no surrounding whitespace, no digit separators
    if s != s.strip() or "_" in s or s == "":
        raise CSVError(f"CSV error: invalid float literal `{s}` in {what}")
    try:
        return float(s)
    except ValueError:
        raise CSVError(f"CSV error: invalid float literal `{s}` in {what}") from None


def _parse_i64(s: str, what: str) -> int:
    ok = s != "" and s == s.strip() and "_" not in s and (s.lstrip("+-").isdigit() and s.lstrip("+-").isascii())
    if not ok or s.count("+") + s.count("-") > 1:
        raise CSVError(f"CSV error: invalid digit found in string `{s}` in {what}")
    v = int(s)
    if not (_I64_MIN <= v <= _I64_MAX):
        raise CSVError(f"CSV error: number too large to fit in target type `{s}` in {what}")
    return v


def _parse_cens(s: str) -> int:
    if s in ("1", "bloq"):
        return Censor.BLOQ
    if s in ("0", "none"):
        return Censor.NONE
    if s in ("-1", "aloq"):
        return Censor.ALOQ
    raise CSVError(f"CSV error: Expected one of 1/-1/0 or bloq/aloq/none), got {s}")


def from_pmetrics_csv_bytes(data: bytes) -> Data:
    """``Data::from_pmetrics_csv_bytes`` (mod.rs:170-232)."""
    try:
        text = data.decode("utf-8")
    except UnicodeDecodeError as e:
        raise CSVError(f"CSV error: {e}") from None
    # the csv crate's `comment(Some(b'#'))`: records whose first byte is '#' are skipped; so are empty lines
    lines = [ln for ln in text.splitlines(keepends=True) if not ln.startswith("#")]
    records = [r for r in csv.reader(io.StringIO("".join(lines))) if r != []]
    original = records[0] if records else []
    core_seen = set()
    forms: Dict[str, bool] = {}
    headers: List[Tuple[str, Optional[str]]] = []  # (kind, name): ("core", "ID") | ("cov", "wt" / "wt!")
    for h in original:
        core = _core_of(h)
        if core is not None:
            if core in core_seen:
                raise InvalidPmetricsData(f"duplicate core header `{core.lower()}`")
            core_seen.add(core)
            headers.append(("core", core))
            continue
        _validate_covariate_header(h)
        fixed = h.endswith("!")
        name = (h[:-1] if fixed else h).lower()
        if name in forms:
            raise InvalidPmetricsData(f"duplicate covariate column `{name}`" if forms[name] == fixed else
                                      f"covariate `{name}` is declared both with and without trailing !")
        forms[name] = fixed
        headers.append(("cov", name + "!" if fixed else name))
    for req in REQUIRED_HEADERS:
        if req not in core_seen:
            raise InvalidPmetricsData(f"missing required core header `{req}`")
    rows: List[DataRow] = []
    for ln, rec in enumerate(records[1:], start=2):
        if len(rec) != len(headers):
            raise CSVError(f"CSV error: record {ln} has {len(rec)} fields, the header has {len(headers)}")
        f: Dict[str, str] = {}
        covs: Dict[str, float] = {}
        for (kind, name), value in zip(headers, rec):
            if kind == "core":
                f[name] = value
            else:
                v = _opt(value)
                if v is not None:
                    covs[name] = _parse_f64(v, f"covariate {name}")
        g = lambda k: _opt(f.get(k, ""))  # noqa: E731  (#[serde(default)]: an absent column reads as None)
        num = lambda k: None if g(k) is None else _parse_f64(g(k), k)  # noqa: E731
        out = num("OUT")
        rows.append(DataRow(
            id=f["ID"], evid=_parse_i64(f["EVID"], "EVID"), time=_parse_f64(f["TIME"], "TIME"),
            dur=num("DUR"), dose=num("DOSE"), addl=None if g("ADDL") is None else _parse_i64(g("ADDL"), "ADDL"),
            ii=num("II"), input=g("INPUT"), out=None if out == -99.0 else out,  # OUT=-99: missing (mod.rs:293-294)
            outeq=g("OUTEQ"), cens=None if g("CENS") is None else _parse_cens(g("CENS")),
            c0=num("C0"), c1=num("C1"), c2=num("C2"), c3=num("C3"), covariates=covs))
    return build_data(rows)


def read_pmetrics(path: str) -> Data:
    """``read_pmetrics(path)`` (mod.rs:165-169)."""
    try:
        with open(path, "rb") as fh:
            data = fh.read()
    except OSError as e:
        raise CSVError(f"CSV error: {e}") from None
    return from_pmetrics_csv_bytes(data)
