"""``Equation`` surface: ``Analytical`` and ``ODE`` models + ``estimate_predictions``.

Python mirror of the reference's model-authoring and simulation entry points:

* ``analytical(...)`` / ``ode(...)``  ~ the ``analytical!`` / ``ode!`` declarations
  (pharmsol-macros/src/expand/analytical.rs:31-135, expand/ode.rs:126-185)
* ``Analytical.new(...)`` / ``ODE.new(...)`` + ``with_nstates/ndrugs/nout``
  ~ src/simulator/equation/analytical/mod.rs:102-152, ode/mod.rs:115-166
* ``Equation.estimate_predictions(subject, parameters)`` ~ equation/mod.rs:526-532
* ``Equation.estimate_predictions_matrix(data, theta)`` ~ the subjects x support
  points loop nest of likelihood/matrix.rs:79-98 (one HIP launch here)

The reference takes Rust closures for eq/derive/out/init/lag/fa.  The device needs
closed descriptions, so closures are restricted to the declarative forms of
``include/pmx.h`` (built-in structures, allometric ``derive``, ``x[state]/v`` outputs).

All compute goes through ``libpmx_hip.so`` (``_ffi``); there is no CPU fallback.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _abi
from .data import Data, Subject
from .flatten import FlatPopulation, flatten
from .parameters import Parameters
from .predictions import SubjectPredictions, Prediction


# ---------------------------------------------------------------------------
# declarative closure forms
# ---------------------------------------------------------------------------
@dataclass(frozen=True)
class Pow:
    """``(cov / ref).powf(coef)``"""
    cov: str
    ref: float
    coef: float


@dataclass(frozen=True)
class Lin:
    """``1.0 + coef * (cov - ref)``"""
    cov: str
    ref: float
    coef: float


@dataclass(frozen=True)
class Scaled:
    """``derived = param * f0 * f1`` — one line of a ``derive:`` block."""
    param: str
    factors: Tuple[Union[Pow, Lin], ...] = ()


@dataclass(frozen=True)
class Ratio:
    """``y[out] = x[state] / vol`` — one line of an ``out:`` block (vol None = 1)."""
    state: Union[str, int]
    vol: Optional[Union[str, int]] = None


@dataclass(frozen=True)
class Route:
    """``bolus(oral) -> gut`` / ``infusion(iv) -> central`` (equation/metadata.rs Route)."""
    kind: str  # "bolus" | "infusion"
    name: str
    dest: Union[str, int]


def bolus(name: str, dest) -> Route:
    return Route("bolus", name, dest)


def infusion(name: str, dest) -> Route:
    return Route("infusion", name, dest)


class LabelError(KeyError):
    """PharmsolError::UnknownInputLabel / UnknownOutputLabel / UnsupportedInputRouteKind."""


def _is_bare_numeric(label: str) -> bool:
    return label.isdigit()


class Equation:
    """Common model state + the simulation entry points."""

    eq_kind: int = -1

    def __init__(self):
        self.name = ""
        self.params: List[str] = []
        self.derived: Dict[str, Scaled] = {}
        self.covariates: List[str] = []
        self.states: List[str] = []
        self.outputs: List[str] = []
        self.routes: List[Route] = []
        self.out: Dict[Union[str, int], Ratio] = {}
        self.init: Dict[Union[str, int], str] = {}
        self.lag: Dict[str, str] = {}
        self.fa: Dict[str, str] = {}
        self.has_metadata = False
        self.nstates = 5  # Neqs::default(): all sizes = 5 (analytical/mod.rs:93)
        self.ndrugs = 5
        self.nout = 5
        self.nparams: Optional[int] = None
        self.kernel_name = ""
        self.cov_time = "segment_dt"
        self.pmetrics = False
        self.rk4_h_max = 0.02
        self._handle = None  # lazily created pmx_model*
        # user closures (Analytical.user / analytical(..., source=...)): C/HIP source text + which pmx_<role> it defines
        self.user_fns = 0
        self.user_derived: List[str] = []

    # -- builder methods (analytical/mod.rs:120-138) ---------------------------
    def with_nstates(self, n: int):
        self.nstates = int(n)
        self._handle = None
        return self

    def with_ndrugs(self, n: int):
        self.ndrugs = int(n)
        self._handle = None
        return self

    def with_nout(self, n: int):
        self.nout = int(n)
        self._handle = None
        return self

    # -- metadata lookups ------------------------------------------------------
    def parameter_index(self, name: str) -> Optional[int]:
        return self.params.index(name) if name in self.params else None

    def state_index(self, name) -> Optional[int]:
        if isinstance(name, int):
            return name
        return self.states.index(name) if name in self.states else None

    def covariate_index(self, name: str) -> Optional[int]:
        return self.covariates.index(name) if name in self.covariates else None

    def _route_inputs(self) -> List[Tuple[Route, int]]:
        # bolus and infusion routes are numbered independently, in declaration order
        # (metadata.rs:926-946; symbols.rs:189-211)
        nb = ni = 0
        out = []
        for r in self.routes:
            if r.kind == "bolus":
                out.append((r, nb))
                nb += 1
            else:
                out.append((r, ni))
                ni += 1
        return out

    def resolve_input_label(self, label, kind: str) -> int:
        """``EquationPriv::resolve_input_label`` (equation/mod.rs:195-233)."""
        s = str(label)
        if self.has_metadata:
            routes = self._route_inputs()
            for r, idx in routes:
                if r.kind == kind and r.name == s:
                    return idx
            if _is_bare_numeric(s):  # canonical alias input_<n> (metadata.rs:248-262)
                for r, idx in routes:
                    if r.kind == kind and r.name == f"input_{s}":
                        return idx
            for r, idx in routes:
                if r.kind != kind and (r.name == s or (_is_bare_numeric(s) and r.name == f"input_{s}")):
                    raise LabelError(f"UnsupportedInputRouteKind: input {idx} is not a {kind} route")
            raise LabelError(f"unknown input label '{s}'; available: {[r.name for r in self.routes]}")
        if not s.isdigit():
            raise LabelError(f"unknown input label '{s}' (no metadata: labels must be dense indices)")
        return int(s)

    def resolve_output_label(self, label) -> int:
        """``EquationPriv::resolve_output_label`` (equation/mod.rs:235-245)."""
        s = str(label)
        if self.has_metadata:
            if s in self.outputs:
                return self.outputs.index(s)
            if _is_bare_numeric(s) and f"outeq_{s}" in self.outputs:
                return self.outputs.index(f"outeq_{s}")
            raise LabelError(f"unknown output label '{s}'; available: {self.outputs}")
        if not s.isdigit():
            raise LabelError(f"unknown output label '{s}' (no metadata: labels must be dense indices)")
        return int(s)

    # -- lowering to the C descriptor -----------------------------------------
    def _value_src(self, name) -> Tuple[int, int]:
        if isinstance(name, int):
            return (_abi.PMX_SRC_PRIMARY, name)
        if name in self.params:
            return (_abi.PMX_SRC_PRIMARY, self.params.index(name))
        if name in self.derived:
            return (_abi.PMX_SRC_DERIVED, list(self.derived).index(name))
        if name in self.user_derived:
            return (_abi.PMX_SRC_DERIVED, self.user_derived.index(name))
        raise KeyError(f"'{name}' is neither a parameter nor a derived value")

    def _required_names(self) -> List[str]:
        raise NotImplementedError

    def desc(self) -> _abi.pmx_model_desc:
        d = _abi.pmx_model_desc()
        d.eq_kind = self.eq_kind
        d.kernel = self._kernel_id()
        d.nstates, d.ndrugs, d.nout = self.nstates, self.ndrugs, self.nout
        nparams = self.nparams if self.nparams is not None else len(self.params)
        d.nparams = nparams
        d.n_covariates = len(self.covariates)
        if len(self.derived) > _abi.PMX_MAX_DERIVED:
            raise ValueError("too many derived values")
        d.n_derived = len(self.derived)
        if self.user_fns:  # the user's pmx_derive writes them; no descriptors
            if len(self.user_derived) > _abi.PMX_MAX_USER_DERIVED:
                raise ValueError("too many derived values")
            d.n_derived = len(self.user_derived)
        for i, (name, sc) in enumerate(self.derived.items()):
            dd = d.derived[i]
            dd.src_param = self.params.index(sc.param)
            dd.n_factors = len(sc.factors)
            if len(sc.factors) > _abi.PMX_MAX_FACTORS:
                raise ValueError("too many covariate factors in one derived value")
            for k, f in enumerate(sc.factors):
                dd.f[k].op = _abi.PMX_F_POW if isinstance(f, Pow) else _abi.PMX_F_LIN
                dd.f[k].cov = self.covariates.index(f.cov)
                dd.f[k].ref = f.ref
                dd.f[k].coef = f.coef
        # kernel-order binding (expand/analytical.rs:208-294): identity when the
        # required names are the leading params in kernel order.
        req = self._required_names()
        d.n_bind = 0
        if req and self.params:
            binds = [self._value_src(n) for n in req]
            identity = all(src == _abi.PMX_SRC_PRIMARY and idx == j for j, (src, idx) in enumerate(binds))
            if not identity:
                d.n_bind = len(binds)
                for j, (src, idx) in enumerate(binds):
                    d.bind[j].src, d.bind[j].index = src, idx
        # outputs
        for o in range(_abi.PMX_MAX_OUT):
            d.out[o].state, d.out[o].vol_src, d.out[o].vol_index = 0, _abi.PMX_SRC_NONE, 0
        for key, r in self.out.items():
            o = self.outputs.index(key) if isinstance(key, str) else int(key)
            st = self.state_index(r.state)
            if st is None:
                raise KeyError(f"unknown state {r.state}")
            d.out[o].state = st
            if r.vol is None:
                d.out[o].vol_src = _abi.PMX_SRC_NONE
            else:
                d.out[o].vol_src, d.out[o].vol_index = self._value_src(r.vol)
        d.cov_time_mode = (_abi.PMX_COV_TIME_SEGMENT_DT if self.cov_time == "segment_dt" else
                           _abi.PMX_COV_TIME_SEGMENT_END_ABS)
        d.pmetrics_indexing = 1 if self.pmetrics else 0
        for i in range(_abi.PMX_MAX_STATES):
            d.init_param[i] = -1
        for key, pname in self.init.items():
            d.init_param[self.state_index(key)] = self.params.index(pname) if isinstance(pname, str) else int(pname)
        for i in range(_abi.PMX_MAX_INPUTS):
            d.lag_param[i] = d.fa_param[i] = -1
            d.bolus_dest[i] = d.infusion_dest[i] = -1
        for label, pname in self.lag.items():
            d.lag_param[self.resolve_input_label(label, "bolus")] = (
                self.params.index(pname) if isinstance(pname, str) else int(pname))
        for label, pname in self.fa.items():
            d.fa_param[self.resolve_input_label(label, "bolus")] = (
                self.params.index(pname) if isinstance(pname, str) else int(pname))
        for r, idx in self._route_inputs():
            dest = self.state_index(r.dest)
            if r.kind == "bolus":
                d.bolus_dest[idx] = dest
            else:
                d.infusion_dest[idx] = dest
        d.rk4_h_max = self.rk4_h_max
        d.ode_solver = getattr(self, "ode_solver", _abi.PMX_SOLVER_RK4)
        d.ode_rtol, d.ode_atol = getattr(self, "ode_rtol", 1e-4), getattr(self, "ode_atol", 1e-4)
        return d

    def _kernel_id(self) -> int:
        raise NotImplementedError

    # -- simulation -------------------------------------------------------------
    def flatten(self, data) -> FlatPopulation:
        return flatten(self, data)

    def estimate_predictions(self, subject: Subject, parameters, with_state: bool = False) -> SubjectPredictions:
        """``Equation::estimate_predictions`` (equation/mod.rs:526-532) for one subject and one
        support point, on the GPU.  ``with_state``: also fill ``Prediction.state`` (the reference records the state
        vector beside every prediction, analytical/mod.rs:401; here one extra device pass per state)."""
        theta = parameters.as_slice() if isinstance(parameters, Parameters) else np.asarray(parameters, dtype=np.float64)
        flat = self.flatten(subject)
        from . import runtime
        pred, status = runtime.predict_host(self, flat, theta.reshape(1, -1))
        states = None
        if with_state:
            pop = runtime.DevicePopulation(flat, 0)
            states = runtime.predict_states(self, pop, theta.reshape(1, -1))[:, :, 0].cpu().numpy()
        return SubjectPredictions.from_flat(subject, self, pred[:, 0], states)

    def estimate_log_likelihood(self, subject: Subject, parameters, error_models) -> float:
        """``Equation::estimate_log_likelihood`` (equation/mod.rs:468-477, 534-547): the subject's summed log-likelihood
        under one support point (fused on the device: the predictions are never stored).  A failed pair raises, like the
        reference's ``Err``."""
        theta = parameters.as_slice() if isinstance(parameters, Parameters) else np.asarray(parameters, dtype=np.float64)
        from . import runtime
        ll, _ = runtime.loglik_host(self, self.flatten(subject), error_models, theta.reshape(1, -1), raise_on_pair_failure=True)
        return float(ll[0, 0])

    def simulate_subject(self, subject: Subject, parameters, error_models=None):
        """``Equation::simulate_subject`` (equation/mod.rs:569-576): ``(predictions, Some(likelihood) | None)`` - the
        likelihood is the PRODUCT of the observations' likelihoods (equation/mod.rs:514, analytical/mod.rs:403-405)."""
        import math

        preds = self.estimate_predictions(subject, parameters)
        if error_models is None:
            return preds, None
        return preds, math.exp(self.estimate_log_likelihood(subject, parameters, error_models))

    def population_predictions(self, data, theta: np.ndarray):
        """Every subject x every support point as the reference's ``PopulationPredictions`` (subject.rs:140-165), from one
        device pass."""
        from .predictions import PopulationPredictions

        flat = self.flatten(data)
        pred, _ = self.estimate_predictions_matrix(flat, theta)
        return PopulationPredictions.from_matrix(data, self, pred, flat.observation_offsets())

    def log_likelihood_batch(self, data, parameters: np.ndarray, residual_error_models) -> np.ndarray:
        """``log_likelihood_batch(&eq, &data, &parameters, &residual_error_models)`` (likelihood/mod.rs:119-177): subject
        i under parameter row i, sigma from the prediction; a subject that fails scores ``-inf`` (:137-140)."""
        flat = data if isinstance(data, FlatPopulation) else self.flatten(data)
        parameters = np.ascontiguousarray(parameters, dtype=np.float64)
        if parameters.ndim != 2 or parameters.shape[0] != flat.n_subjects:
            raise ValueError(f"parameters has {parameters.shape[0] if parameters.ndim == 2 else '?'} rows but there are "
                             f"{flat.n_subjects} subjects")
        from . import runtime
        ll, _ = runtime.loglik_batch_host(self, flat, residual_error_models, parameters)
        return ll

    def estimate_predictions_matrix(self, data, theta: np.ndarray):
        """All subjects x all support points (likelihood/matrix.rs:79-98 loop nest).

        Returns ``(pred[n_observations, n_support], status[n_subjects, n_support])`` as numpy arrays
        (host-pointer ABI form).  Use ``runtime.DevicePopulation`` for the resident form.
        """
        flat = data if isinstance(data, FlatPopulation) else self.flatten(data)
        from . import runtime
        return runtime.predict_host(self, flat, np.ascontiguousarray(theta, dtype=np.float64))


    def log_likelihood_matrix(self, data, theta: np.ndarray, error_models):
        """``log_likelihood_matrix(eq, &data, &theta, &error_models, progress)`` (likelihood/matrix.rs:52-106):
        ``(ll[n_subjects, n_support], status)``; the predictions never leave the GPU."""
        flat = data if isinstance(data, FlatPopulation) else self.flatten(data)
        from . import runtime
        return runtime.loglik_host(self, flat, error_models, np.ascontiguousarray(theta, dtype=np.float64))


class Analytical(Equation):
    """Closed-form model (src/simulator/equation/analytical/mod.rs:48-59)."""

    eq_kind = _abi.PMX_EQ_ANALYTICAL

    @staticmethod
    def new(eq: str, out: Dict[int, Ratio], *, nparams: int, init: Optional[Dict[int, int]] = None,
            lag: Optional[Dict[int, int]] = None, fa: Optional[Dict[int, int]] = None) -> "Analytical":
        """``Analytical::new(eq, seq_eq, lag, fa, init, out)`` with a built-in ``eq`` (by name; a ``pm_``
        prefix selects the Pmetrics 1-indexed wrapper) and index-based closures.  No metadata:
        data labels must be dense numeric indices."""
        m = Analytical()
        if eq.startswith("pm_"):
            m.pmetrics = True
            eq = eq[3:]
        if eq not in _abi.ANALYTICAL_KERNELS:
            raise KeyError(f"unknown analytical structure '{eq}'")
        m.kernel_name = eq
        m.out = dict(out)
        m.nparams = int(nparams)
        m.init = dict(init or {})
        m.lag = {str(k): v for k, v in (lag or {}).items()}
        m.fa = {str(k): v for k, v in (fa or {}).items()}
        return m

    source: Optional[str] = None  # user closures (Analytical.user / analytical(..., source=...))

    @staticmethod
    def user(source: str, *, eq: Optional[str] = None, nstates: int, nparams: int, ndrugs: int = 1, nout: int = 1,
             covariates: Optional[Sequence[str]] = None, n_derived: int = 0, bind: Optional[Sequence] = None,
             out: Optional[Dict[int, Ratio]] = None, init: Optional[Dict[int, int]] = None,
             lag: Optional[Dict[int, int]] = None, fa: Optional[Dict[int, int]] = None,
             cov_time: str = "segment_dt") -> "Analytical":
        """``Analytical::new(eq, seq_eq, lag, fa, init, out)`` with USER closures (analytical/mod.rs:102-118), index
        based like the reference's hand-written form.  ``source`` is C/HIP text defining any of ``pmx_derive``,
        ``pmx_route_lag``, ``pmx_route_bioavailability``, ``pmx_init``, ``pmx_outputs``, ``pmx_seq_eq``, ``pmx_eq``
        (include/pmx.h "user closures"); the library compiles it for gfx950 with hiprtc.  ``eq`` names a built-in
        structure, or is None when the source brings its own propagator ``pmx_eq``.  ``bind``: kernel-order parameter
        j <- ``("p", k)`` theta[k] or ``("d", k)`` derived[k] (default: the leading parameters).  Closures the source
        leaves out fall back to the index forms (``out`` / ``init`` / ``lag`` / ``fa``) of ``Analytical.new``."""
        m = Analytical()
        if eq is not None and eq.startswith("pm_"):  # Pmetrics 1-indexed wrapper around the structure (analytical/mod.rs:62-90)
            m.pmetrics = True
            eq = eq[3:]
        m.kernel_name = eq or "custom"
        if eq is not None and eq not in _abi.ANALYTICAL_KERNELS:
            raise KeyError(f"unknown analytical structure '{eq}'")
        m.covariates = list(covariates or [])
        m._set_user_source(source)
        m.user_derived = [f"d{i}" for i in range(int(n_derived))]
        m.user_bind = [tuple(b) for b in (bind or [])]
        m.out = dict(out or {})
        m.nparams = int(nparams)
        m.init = dict(init or {})
        m.lag = {str(k): v for k, v in (lag or {}).items()}
        m.fa = {str(k): v for k, v in (fa or {}).items()}
        m.cov_time = cov_time
        return m.with_nstates(nstates).with_ndrugs(ndrugs).with_nout(nout)

    user_bind: Sequence = ()

    def _set_user_source(self, source: str):
        fns = _abi.user_functions_of(source)
        if not fns:
            raise ValueError("the source defines none of pmx_derive / pmx_route_lag / pmx_route_bioavailability / pmx_init / "
                             "pmx_outputs / pmx_seq_eq / pmx_eq")
        if (fns & _abi.PMX_FN_EQ) and self.kernel_name != "custom":
            raise ValueError("pmx_eq replaces the structure: declare no built-in structure with it")
        if self.kernel_name == "custom" and not (fns & _abi.PMX_FN_EQ):
            raise ValueError("no structure named and no pmx_eq in the source")
        self.source, self.user_fns = str(source), fns
        self._handle = None

    def desc(self) -> _abi.pmx_model_desc:
        d = super().desc()
        if self.user_fns and self.user_bind:  # index-based binding of Analytical.user
            d.n_bind = len(self.user_bind)
            for j, (src, idx) in enumerate(self.user_bind):
                d.bind[j].src = _abi.PMX_SRC_DERIVED if src == "d" else _abi.PMX_SRC_PRIMARY
                d.bind[j].index = int(idx)
        return d

    def _kernel_id(self) -> int:
        return _abi.PMX_K_CUSTOM if self.kernel_name == "custom" else _abi.ANALYTICAL_KERNELS[self.kernel_name]

    def _required_names(self) -> List[str]:
        if self.kernel_name == "custom":
            return []
        return _abi.KERNEL_PARAMETER_NAMES[self.kernel_name] if self.has_metadata else []


class ODE(Equation):
    """ODE model integrated with fixed-step RK4 on the device (src/simulator/equation/ode/mod.rs:98-132;
    the reference's diffsol solvers are replaced)."""

    eq_kind = _abi.PMX_EQ_ODE

    @staticmethod
    def new(diffeq: str, out: Dict[int, Ratio], *, nparams: int, init: Optional[Dict[int, int]] = None,
            lag: Optional[Dict[int, int]] = None, fa: Optional[Dict[int, int]] = None, h_max: float = 0.02) -> "ODE":
        """``ODE::new(diffeq, lag, fa, init, out)`` (ode/mod.rs:115-132) with a built-in ``diffeq`` body."""
        m = ODE()
        if diffeq not in _abi.ODE_MODELS:
            raise KeyError(f"unknown built-in diffeq '{diffeq}'")
        m.kernel_name = diffeq
        m.out = dict(out)
        m.nparams = int(nparams)
        m.init = dict(init or {})
        m.lag = {str(k): v for k, v in (lag or {}).items()}
        m.fa = {str(k): v for k, v in (fa or {}).items()}
        m.rk4_h_max = float(h_max)
        return m

    @staticmethod
    def custom(source: str, *, nstates: int, nparams: int, ndrugs: int = 1, nout: int = 1, has_init: bool = False,
               covariates: Optional[Sequence[str]] = None, lag: Optional[Dict[int, int]] = None,
               fa: Optional[Dict[int, int]] = None, h_max: float = 0.02) -> "ODE":
        """``ODE::new(diffeq, lag, fa, init, out)`` with USER bodies (ode/mod.rs:115-132): ``source`` is C/HIP text
        defining ``pmx_dynamics`` / ``pmx_outputs`` (/ ``pmx_init``) as described in include/pmx.h; the library
        compiles it for gfx950 with hiprtc.  Index-based like ``ODE::new``: data labels are dense numeric indices,
        a bolus on input i goes to state i.  ``covariates`` names the subject covariates the bodies read as
        ``cov[0..]`` (interpolated at the time of every right-hand-side evaluation, like ``fetch_cov!``)."""
        m = ODE()
        m.covariates = list(covariates or [])
        m.kernel_name = "custom"
        m.source = str(source)
        m.has_init = bool(has_init)
        m.out = {}
        m.nparams = int(nparams)
        m.init = {}
        m.lag = {str(k): v for k, v in (lag or {}).items()}
        m.fa = {str(k): v for k, v in (fa or {}).items()}
        m.rk4_h_max = float(h_max)
        return m.with_nstates(nstates).with_ndrugs(ndrugs).with_nout(nout)

    @staticmethod
    def user(source: str, *, nstates: int, nparams: int, ndrugs: int = 1, nout: int = 1,
             covariates: Optional[Sequence[str]] = None, n_derived: int = 0, bolus_dest: Optional[Dict[int, int]] = None,
             lag: Optional[Dict[int, int]] = None, fa: Optional[Dict[int, int]] = None, h_max: float = 0.02) -> "ODE":
        """``ODE::new(diffeq, lag, fa, init, out)`` with EVERY closure a function of (theta, t, covariates)
        (ode/mod.rs:115-132; closure types src/simulator/mod.rs:41-197), index based like the reference's hand-written
        form.  ``source`` defines ``pmx_dynamics`` - or ``pmx_dynamics_bolus``, the DiffEq with its ``bolus`` argument -
        and ``pmx_outputs``, and any of ``pmx_init`` / ``pmx_derive`` / ``pmx_route_lag`` / ``pmx_route_bioavailability``
        (include/pmx.h "user closures").  ``bolus_dest``: input -> state for plain ``pmx_dynamics`` bodies (default:
        state = input).  ``lag`` / ``fa``: theta-index fall-backs for closures the source leaves out."""
        m = ODE()
        m.covariates = list(covariates or [])
        m.kernel_name = "custom"
        m.nparams = int(nparams)
        m.user_derived = [f"d{i}" for i in range(int(n_derived))]
        m.lag = {str(k): v for k, v in (lag or {}).items()}
        m.fa = {str(k): v for k, v in (fa or {}).items()}
        m.user_bolus_dest = dict(bolus_dest or {})
        m.rk4_h_max = float(h_max)
        m._set_user_source(source)
        return m.with_nstates(nstates).with_ndrugs(ndrugs).with_nout(nout)

    user_bolus_dest: Dict[int, int] = {}

    def _set_user_source(self, source: str):
        fns = _abi.user_functions_of(source)
        if not (fns & (_abi.PMX_FN_DYNAMICS | _abi.PMX_FN_DYNAMICS_BOLUS)) or not (fns & _abi.PMX_FN_OUTPUTS):
            raise ValueError("an ODE source defines pmx_dynamics (or pmx_dynamics_bolus) and pmx_outputs")
        if (fns & _abi.PMX_FN_DYNAMICS) and (fns & _abi.PMX_FN_DYNAMICS_BOLUS):
            raise ValueError("define pmx_dynamics OR pmx_dynamics_bolus, not both")
        if fns & (_abi.PMX_FN_SEQ_EQ | _abi.PMX_FN_EQ):
            raise ValueError("pmx_seq_eq / pmx_eq belong to analytical models")
        self.source, self.user_fns = str(source), fns
        self.has_init = bool(fns & _abi.PMX_FN_INIT)
        self._handle = None

    def desc(self) -> _abi.pmx_model_desc:
        d = super().desc()
        for i, st in self.user_bolus_dest.items():
            d.bolus_dest[int(i)] = int(st)
        return d

    ode_solver = _abi.PMX_SOLVER_RK4
    ode_rtol = 1e-4  # the reference's defaults (ode/mod.rs:126-127)
    ode_atol = 1e-4

    def with_solver(self, solver: str) -> "ODE":
        """``ODE::with_solver`` (ode/mod.rs:134-150).  The reference's diffsol solvers are replaced: ``"rk4"`` = fixed
        step (default), ``"dopri5"`` = adaptive Dormand-Prince 5(4) with per-lane step control (the role of
        ``ExplicitRk(Tsit45)``), ``"ros2"`` (alias ``"stiff"``) = the L-stable Rosenbrock method ROS2 with the same step
        control, for stiff systems (the role of ``Bdf`` / ``Sdirk``, ode/mod.rs:60-77)."""
        self.ode_solver = {"rk4": _abi.PMX_SOLVER_RK4, "dopri5": _abi.PMX_SOLVER_DOPRI5, "ros2": _abi.PMX_SOLVER_ROS2,
                           "stiff": _abi.PMX_SOLVER_ROS2}[solver]
        self._handle = None
        return self

    def with_tolerances(self, rtol: float, atol: float) -> "ODE":
        """``ODE::with_tolerances`` (ode/mod.rs:152-166); read by the adaptive solver."""
        self.ode_rtol, self.ode_atol = float(rtol), float(atol)
        self._handle = None
        return self

    def with_step(self, h_max: float) -> "ODE":
        self.rk4_h_max = float(h_max)
        self._handle = None
        return self

    source: Optional[str] = None  # custom bodies (ODE.custom)
    has_init = False

    def _kernel_id(self) -> int:
        return _abi.PMX_ODE_CUSTOM if self.source is not None else _abi.ODE_MODELS[self.kernel_name]

    def _required_names(self) -> List[str]:
        # a declared ode(...) with derived values binds the body's parameters by name, like the analytical structures
        if self.has_metadata and self.source is None and self.derived:
            return _abi.ODE_PARAMETER_NAMES[self.kernel_name]
        return []


def _declare(m: Equation, name, params, derived, covariates, states, outputs, routes, out, init, lag, fa):
    m.name = name
    m.params = list(params)
    m.derived = dict(derived or {})
    m.covariates = list(covariates or [])
    m.states = list(states)
    m.outputs = list(outputs)
    m.routes = list(routes or [])
    m.out = dict(out or {})
    m.init = dict(init or {})
    m.lag = dict(lag or {})
    m.fa = dict(fa or {})
    m.has_metadata = True
    m.nstates = len(m.states)
    m.nout = len(m.outputs)
    nb = sum(1 for r in m.routes if r.kind == "bolus")
    ni = sum(1 for r in m.routes if r.kind == "infusion")
    m.ndrugs = max(nb, ni, 1)  # validate_routes returns max(bolus_inputs, infusion_inputs) (metadata.rs:926-946)
    return m


def analytical(*, name: str, params: Sequence[str], structure: Optional[str], states: Sequence[str], outputs: Sequence[str],
               routes: Sequence[Route], out: Optional[Dict[str, Ratio]] = None, derived=None,
               covariates: Optional[Sequence[str]] = None, init=None, lag=None, fa=None,
               cov_time: str = "segment_dt", source: Optional[str] = None) -> Analytical:
    """The ``analytical!`` declaration (e.g. examples/analytical_readme.rs:7-24).

    Closures come in two forms.  Declarative (``derived={name: Scaled(..)}``, ``out={..: Ratio(..)}``, ``lag`` / ``fa`` /
    ``init`` = parameter names): the closed forms the library's own kernels know.  Or ``source=``: C/HIP text with the
    bodies of the macro's ``derive:`` / ``lag:`` / ``fa:`` / ``init:`` / ``out:`` blocks as ``pmx_derive`` / ``pmx_route_lag`` /
    ``pmx_route_bioavailability`` / ``pmx_init`` / ``pmx_outputs`` (+ ``pmx_seq_eq``, ``pmx_eq``), compiled for gfx950 at run time
    (include/pmx.h "user closures"); then ``derived`` is the LIST of names ``pmx_derive`` writes, and the source may use
    the generated index constants ``P_<param>``, ``D_<derived>``, ``COV_<covariate>``, ``X_<state>``, ``Y_<output>``,
    ``R_<route>`` the way the macro binds names (tests/test_user_analytical.py re-creates
    tests/analytical_macro_lowering.rs:225-260 with it)."""
    m = Analytical()
    if structure is not None and structure not in _abi.ANALYTICAL_KERNELS:
        raise KeyError(f"unknown analytical structure '{structure}'")
    m.kernel_name = structure or "custom"
    user_derived: List[str] = []
    if source is not None:
        user_derived = list(derived or [])
        derived = None
    _declare(m, name, params, derived, covariates, states, outputs, routes, out, init, lag, fa)
    if source is not None:
        m.user_derived = user_derived
        m._set_user_source(_name_constants(m, user_derived) + source)
    if structure is None:
        if source is None:
            raise ValueError("no structure and no source")
        m.cov_time = cov_time
        return m
    if len(m.states) != _abi.KERNEL_STATE_COUNT[structure]:
        raise ValueError(f"structure {structure} has {_abi.KERNEL_STATE_COUNT[structure]} states, "
                         f"{len(m.states)} declared")
    for n in _abi.KERNEL_PARAMETER_NAMES[structure]:
        if n not in m.params and n not in m.derived and n not in m.user_derived:
            raise KeyError(f"structure {structure} requires '{n}' in params or derived")
    if cov_time not in ("segment_dt", "segment_end_abs"):
        raise ValueError("cov_time must be 'segment_dt' or 'segment_end_abs'")
    m.cov_time = cov_time
    # Like the reference, an analytical model puts a dose into x[input index] (`x.add_bolus(input, amount)`,
    # equation/mod.rs:313-328; a route's `to_state` is descriptive metadata for this back-end,
    # ode/mod.rs:1141 "input_policy_is_descriptive_only"; the analytical! macro lowers routes to metadata only,
    # expand/analytical.rs:480-522) and an infusion's rate into rateiv[input index], which the structure adds to its
    # central state.  Routes declared in another order than the structure's states would silently dose another state
    # than their `to_state` says: say so.
    for r, idx in m._route_inputs():
        if r.kind == "bolus" and m.state_index(r.dest) != idx:
            import warnings

            warnings.warn(f"analytical model '{name}': bolus route '{r.name}' is input {idx}, so its doses go to state "
                          f"{idx} ('{m.states[idx] if idx < len(m.states) else '?'}'), not to '{r.dest}' - declare the bolus "
                          "routes in the order of the states they dose (the reference behaves the same way)", stacklevel=2)
    return m


def _name_constants(m: Equation, user_derived: Sequence[str]) -> str:
    """``enum { P_<param>, D_<derived>, COV_<covariate>, X_<state>, Y_<output>, R_<route> }``: the index constants a
    ``source=`` text may use the way the macros bind names (symbols.rs:189-239)."""
    consts = ([f"P_{n} = {i}" for i, n in enumerate(m.params)] + [f"D_{n} = {i}" for i, n in enumerate(user_derived)] +
              [f"COV_{n} = {i}" for i, n in enumerate(m.covariates)] + [f"X_{n} = {i}" for i, n in enumerate(m.states)] +
              [f"Y_{n} = {i}" for i, n in enumerate(m.outputs)] +
              [f"R_{n} = {i}" for n, i in dict((r.name, i) for r, i in m._route_inputs()).items()])
    return "enum { " + ", ".join(consts) + " };\n"


def ode(*, name: str, params: Sequence[str], diffeq: Optional[str] = None, states: Sequence[str], outputs: Sequence[str],
        routes: Sequence[Route], out: Optional[Dict[str, Ratio]] = None, derived=None, covariates=None, init=None, lag=None,
        fa=None, h_max: float = 0.02, source: Optional[str] = None) -> ODE:
    """The ``ode!`` declaration: a built-in ``diffeq`` body by name (e.g. examples/ode_readme.rs:9-23), or ``source=``:
    C/HIP text with the bodies of the macro's ``diffeq:`` / ``lag:`` / ``fa:`` / ``init:`` / ``out:`` blocks as
    ``pmx_dynamics`` (routes are injected like the macro does: ``dx[dest] += rateiv[input]``, a bolus lands in its
    route's destination state, expand/ode.rs:380-406) - or ``pmx_dynamics_bolus``, the hand-written DiffEq that adds
    ``bolus[..]`` / ``rateiv[..]`` itself - plus ``pmx_route_lag`` / ``pmx_route_bioavailability`` / ``pmx_init`` /
    ``pmx_outputs`` (/ ``pmx_derive``, then ``derived`` lists the names it writes), compiled for gfx950 at run time.
    tests/test_full_feature_parity.py re-creates tests/full_feature_macro_parity.rs:10-53 with it."""
    m = ODE()
    if source is not None:
        if diffeq is not None:
            raise ValueError("a source brings its own pmx_dynamics: name no built-in diffeq with it")
        user_derived = list(derived or [])
        m.kernel_name = "custom"
        _declare(m, name, params, None, covariates, states, outputs, routes, out, init, lag, fa)
        m.user_derived = user_derived
        m.rk4_h_max = float(h_max)
        text = _name_constants(m, user_derived) + source
        fns = _abi.user_functions_of(text)
        if fns & _abi.PMX_FN_DYNAMICS:
            # the macro appends the route injection to the diffeq body (expand/ode.rs:380-406): infusion routes only -
            # a bolus is a jump of `amount` in its destination state (bolus_dest, from the routes)
            inj = "".join(f"  dx[{m.state_index(r.dest)}] += rateiv[{idx}];\n" for r, idx in m._route_inputs() if r.kind == "infusion")
            text = text.replace("pmx_dynamics(", "pmx_dynamics_body_(") + (
                "\nPMX_DEVICE void pmx_dynamics(double t, const double* x, const double* p, const double* cov, "
                "const double* rateiv, const double* derived, double* dx) {\n"
                "  pmx_dynamics_body_(t, x, p, cov, rateiv, derived, dx);\n" + inj + "}\n")
        m._set_user_source(text)
        return m
    if diffeq not in _abi.ODE_MODELS:
        raise KeyError(f"unknown built-in diffeq '{diffeq}'")
    m.kernel_name = diffeq
    _declare(m, name, params, derived, covariates, states, outputs, routes, out or {}, init, lag, fa)
    if len(m.states) != _abi.ODE_STATE_COUNT[diffeq]:
        raise ValueError(f"diffeq {diffeq} has {_abi.ODE_STATE_COUNT[diffeq]} states, {len(m.states)} declared")
    m.rk4_h_max = float(h_max)
    return m
