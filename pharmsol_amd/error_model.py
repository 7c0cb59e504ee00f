"""Assay error models (src/data/error_model.rs): sigma from the OBSERVATION through an error polynomial.

Mirror of ``ErrorPoly`` / ``AssayErrorModel::{additive, proportional}`` / ``AssayErrorModels::empty().add(outeq, ..)``
for the fused log-likelihood entry point (``pmx_loglik``).  Holds no numerics: the library evaluates
``AssayErrorModel::sigma`` (error_model.rs:1045-1080) per observation.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Union

from . import _abi


@dataclass(frozen=True)
class ErrorPoly:
    """``ErrorPoly::new(c0, c1, c2, c3)``: alpha = c0 + c1*y + c2*y^2 + c3*y^3."""
    c0: float
    c1: float
    c2: float
    c3: float


@dataclass(frozen=True)
class AssayErrorModel:
    kind: int
    poly: ErrorPoly
    scalar: float

    @staticmethod
    def additive(poly: ErrorPoly, lam: float) -> "AssayErrorModel":
        """sigma = sqrt(alpha^2 + lambda^2)"""
        return AssayErrorModel(_abi.PMX_EM_ADDITIVE, poly, float(lam))

    @staticmethod
    def proportional(poly: ErrorPoly, gamma: float) -> "AssayErrorModel":
        """sigma = gamma * alpha"""
        return AssayErrorModel(_abi.PMX_EM_PROPORTIONAL, poly, float(gamma))


class AssayErrorModels:
    """``AssayErrorModels::empty().add(outeq, model)`` — one model per output equation."""

    def __init__(self):
        self._m: Dict[Union[int, str], AssayErrorModel] = {}

    @staticmethod
    def empty() -> "AssayErrorModels":
        return AssayErrorModels()

    def add(self, outeq: Union[int, str], model: AssayErrorModel) -> "AssayErrorModels":
        if outeq in self._m:
            raise KeyError(f"error model for output {outeq} already present")
        self._m[outeq] = model
        return self

    def to_c(self, equation):
        """``pmx_error_model[nout]`` in the equation's dense output order."""
        nout = equation.desc().nout
        arr = (_abi.pmx_error_model * max(nout, 1))()
        for i in range(nout):
            arr[i].kind = _abi.PMX_EM_NONE
        for key, em in self._m.items():
            # an integer is the dense output slot itself (AssayErrorModels::add(0, ..)); a string is a public label
            idx = key if isinstance(key, int) else equation.resolve_output_label(key)
            if idx >= nout:
                raise _abi.PmxError(_abi.PMX_ERR_OUTEQ_OUT_OF_RANGE, f"error model for outeq {idx} >= nout {nout}")
            arr[idx].kind = em.kind
            arr[idx].c[0], arr[idx].c[1], arr[idx].c[2], arr[idx].c[3] = em.poly.c0, em.poly.c1, em.poly.c2, em.poly.c3
            arr[idx].scalar = em.scalar
        return arr


@dataclass(frozen=True)
class ResidualErrorModel:
    """``ResidualErrorModel`` (src/data/residual_error.rs:69-136): sigma from the PREDICTION ``f`` - what the parametric
    algorithms and ``log_likelihood_batch`` (likelihood/mod.rs:119-177) use.  The library evaluates
    ``sigma(f)`` (:178-191, floored at sqrt(f64::EPSILON)) and ``log_likelihood`` (:265-271) inside the fused kernel."""
    kind: int
    a: float
    b: float = 0.0

    @staticmethod
    def constant(a: float) -> "ResidualErrorModel":
        """sigma = a"""
        return ResidualErrorModel(_abi.PMX_EM_RES_CONSTANT, float(a))

    @staticmethod
    def proportional(b: float) -> "ResidualErrorModel":
        """sigma = b |f|"""
        return ResidualErrorModel(_abi.PMX_EM_RES_PROPORTIONAL, float(b))

    @staticmethod
    def combined(a: float, b: float) -> "ResidualErrorModel":
        """sigma = sqrt(a^2 + b^2 f^2)"""
        return ResidualErrorModel(_abi.PMX_EM_RES_COMBINED, float(a), float(b))

    @staticmethod
    def exponential(sigma: float) -> "ResidualErrorModel":
        """sigma = sigma_exp"""
        return ResidualErrorModel(_abi.PMX_EM_RES_EXPONENTIAL, float(sigma))

    def sigma(self, prediction: float) -> float:
        import math

        raw = {_abi.PMX_EM_RES_CONSTANT: self.a, _abi.PMX_EM_RES_PROPORTIONAL: self.a * abs(prediction),
               _abi.PMX_EM_RES_COMBINED: math.sqrt(self.a ** 2 + self.b ** 2 * prediction ** 2),
               _abi.PMX_EM_RES_EXPONENTIAL: self.a}[self.kind]
        return max(raw, math.sqrt(2.220446049250313e-16))


class ResidualErrorModels:
    """``ResidualErrorModels::new().add(outeq, model)`` (residual_error.rs:341-359)."""

    def __init__(self):
        self._m: Dict[int, ResidualErrorModel] = {}

    @staticmethod
    def new() -> "ResidualErrorModels":
        return ResidualErrorModels()

    def add(self, outeq: int, model: ResidualErrorModel) -> "ResidualErrorModels":
        self._m[int(outeq)] = model
        return self

    def to_c(self, equation):
        nout = equation.desc().nout
        arr = (_abi.pmx_error_model * max(nout, 1))()
        for i in range(nout):
            arr[i].kind = _abi.PMX_EM_NONE
        for idx, em in self._m.items():
            if idx >= nout:
                raise _abi.PmxError(_abi.PMX_ERR_OUTEQ_OUT_OF_RANGE, f"error model for outeq {idx} >= nout {nout}")
            arr[idx].kind = em.kind
            arr[idx].scalar = em.a
            arr[idx].c[0] = em.b
        return arr
