"""``Parameters``: one support point in model order.

Mirror of src/parameters.rs:51,74-102 — a dense ``Vec<f64>`` in the model's
declared parameter order, with named ingress through the model's metadata.
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
