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
