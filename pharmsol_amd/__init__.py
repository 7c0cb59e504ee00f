"""pharmsol_amd — MI355X-native batched PK/PD prediction (pharmsol's
``Equation::estimate_predictions`` hot path, subject x support-point).

Host-side mirror of the reference interface for that path (``Subject`` builder,
``Parameters``, ``Analytical`` / ``ODE`` equations) over the C ABI of
``libpmx_hip.so`` (``include/pmx.h``).  All compute is hand-written HIP for
gfx950; there is no CPU fallback (the CPU oracle lives under ``oracle/`` and is
test infrastructure only).
"""
from .data import (Bolus, Censor, Covariates, Data, Event, Infusion, Observation, Occasion, Subject, SubjectBuilder,
                   interpolate)
from .equation import (ODE, Analytical, Equation, LabelError, Lin, Pow, Ratio, Route, Scaled, analytical, bolus,
                       infusion, ode)
from .error_model import AssayErrorModel, AssayErrorModels, ErrorPoly, ResidualErrorModel, ResidualErrorModels
from .flatten import FlatPopulation, flatten
from .parameters import ParameterError, ParameterOrder, Parameters
from .pmetrics import DataError, DataRow, build_data, read_pmetrics
from .predictions import PopulationPredictions, Prediction, SubjectPredictions
from ._abi import PmxError

__all__ = [
    "Bolus", "Censor", "Covariates", "Data", "DataError", "DataRow", "build_data", "read_pmetrics", "interpolate", "Event", "Infusion", "Observation", "Occasion", "Subject", "SubjectBuilder",
    "ODE", "Analytical", "Equation", "LabelError", "Lin", "Pow", "Ratio", "Route", "Scaled", "analytical", "bolus",
    "infusion", "ode", "AssayErrorModel", "AssayErrorModels", "ErrorPoly", "ResidualErrorModel", "ResidualErrorModels", "FlatPopulation", "flatten",
    "Parameters", "ParameterOrder", "ParameterError", "Prediction", "SubjectPredictions", "PopulationPredictions", "PmxError",
]
