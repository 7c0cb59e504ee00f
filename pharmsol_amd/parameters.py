"""``Parameters``: one support point in model order; ``ParameterOrder``: a validated column permutation.

Mirror of src/parameters.rs:51,74-165 and src/parameter_order.rs:12-116 — a dense ``Vec<f64>`` in the model's
declared parameter order with named ingress through the model's metadata, and the plan an NPAG-style caller uses to
bring ITS column order (e.g. the columns of a prior file) into model order once, for single support points and for
whole support-point matrices (the rows ``log_likelihood_matrix`` walks, matrix.rs:62-65).
"""
from __future__ import annotations

from typing import Iterable, Sequence, Tuple

import numpy as np


class Parameters:
    def __init__(self, values: Sequence[float]):
        self._v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)

    @staticmethod
    def dense(values: Sequence[float]) -> "Parameters":
        """``pharmsol::parameters::dense([..])`` — already in model order."""
        return Parameters(values)

    @staticmethod
    def with_model(model, named: Iterable[Tuple[str, float]]) -> "Parameters":
        """``Parameters::with_model(&model, [(name, value), ..])`` (parameters.rs:74-91).

        Every declared parameter must be given exactly once; unknown names fail.
        """
        names = list(model.params)
        if not names:
            raise ValueError("model declares no parameter names; use Parameters.dense")
        out = np.full(len(names), np.nan)
        seen = set()
        for name, value in named:
            if name not in names:
                raise KeyError(f"unknown parameter '{name}'; model declares {names}")
            if name in seen:
                raise KeyError(f"parameter '{name}' given twice")
            seen.add(name)
            out[names.index(name)] = float(value)
        missing = [n for n in names if n not in seen]
        if missing:
            raise KeyError(f"missing parameters {missing}")
        return Parameters(out)

    def as_slice(self) -> np.ndarray:
        return self._v

    def __len__(self) -> int:
        return int(self._v.shape[0])


class ParameterError(KeyError):
    """``ParameterError`` / ``ParameterOrderError``: UnknownParameter, DuplicateParameter, MissingParameters,
    WidthMismatch (parameter_order.rs:118-160)."""


class ParameterOrder:
    """``ParameterOrder::with_model(&model, source_names)`` (parameters.rs:104-165; plan: parameter_order.rs:12-116).

    ``permutation()[model_index] = source_index``: the source column that feeds each model-order slot."""

    def __init__(self, permutation, identity: bool):
        self._perm = [int(p) for p in permutation]
        self._identity = bool(identity)

    @staticmethod
    def with_model(model, source_names) -> "ParameterOrder":
        names = list(model.params)
        if not names:
            raise ParameterError("model declares no parameter names")
        index = {n: i for i, n in enumerate(names)}
        perm = [-1] * len(names)
        width = 0
        for src in source_names:
            src = str(src)
            if src not in index:
                raise ParameterError(f"UnknownParameter: '{src}'; available: {names}")
            if perm[index[src]] != -1:
                raise ParameterError(f"DuplicateParameter: '{src}'")
            perm[index[src]] = width
            width += 1
        missing = [n for i, n in enumerate(names) if perm[i] == -1]
        if missing:
            raise ParameterError(f"MissingParameters: {missing}")
        return ParameterOrder(perm, all(i == p for i, p in enumerate(perm)))

    def values(self, source_values) -> np.ndarray:
        """One dense support point from source order into model order (parameters.rs:119-123)."""
        v = np.ascontiguousarray(source_values, dtype=np.float64).reshape(-1)
        if v.shape[0] != self.width():
            raise ParameterError(f"WidthMismatch: expected {self.width()}, got {v.shape[0]}")
        return v if self._identity else v[self._perm]

    def parameters(self, source_values) -> Parameters:
        return Parameters(self.values(source_values))

    def matrix(self, source_values) -> np.ndarray:
        """A support-point matrix (rows = support points) into model order, column by column (parameters.rs:125-145);
        the result is the row-major ``theta`` the library takes."""
        m = np.asarray(source_values, dtype=np.float64)
        if m.ndim != 2 or m.shape[1] != self.width():
            raise ParameterError(f"WidthMismatch: expected {self.width()}, got {m.shape[1] if m.ndim == 2 else m.shape}")
        return np.ascontiguousarray(m if self._identity else m[:, self._perm])

    def permutation(self):
        return list(self._perm)

    def width(self) -> int:
        return len(self._perm)

    def is_identity(self) -> bool:
        return self._identity
