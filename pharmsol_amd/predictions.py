"""Result containers: ``Prediction`` / ``SubjectPredictions`` / ``PopulationPredictions``.

Mirror of src/simulator/likelihood/prediction.rs:18-27,105-125 and subject.rs:19-21,63-78,105-165.
The device writes only ``pred`` (and, on request, the state amounts: ``pmx_predict_state_device``); time / observation /
outeq / errorpoly / occasion / censoring are re-attached here from the subject (what ``Observation::to_prediction``
does, src/data/event.rs:698-711).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from .data import Censor, Observation, Subject


@dataclass
class Prediction:
    """prediction.rs:18-27"""
    time: float
    observation: Optional[float]
    prediction: float
    outeq: int
    occasion: int
    errorpoly: Optional[object] = None      # the observation's own ErrorPoly, if it carries one
    state: List[float] = field(default_factory=list)  # the state vector at the observation (filled on request)
    censoring: Censor = Censor.NONE

    def prediction_error(self) -> Optional[float]:
        """prediction - observation (prediction.rs:62-65)"""
        return None if self.observation is None else self.prediction - self.observation


class SubjectPredictions:
    def __init__(self, predictions: List[Prediction]):
        self._p = predictions

    @staticmethod
    def from_flat(subject: Subject, model, pred: np.ndarray, states: Optional[np.ndarray] = None) -> "SubjectPredictions":
        """``pred``: this subject's prediction rows; ``states``: optional ``[rows, nstates]`` amounts at the same rows."""
        out: List[Prediction] = []
        row = 0
        for occ in subject.occasions:
            # predictions are emitted in (processed) event order; without lag the order is
            # the stored sort order (equation/mod.rs:500-512)
            for ev in occ.events:
                if isinstance(ev, Observation):
                    out.append(Prediction(ev.time, ev.value, float(pred[row]), model.resolve_output_label(ev.outeq),
                                          occ.index, getattr(ev, "errorpoly", None),
                                          [float(v) for v in states[row]] if states is not None else [],
                                          getattr(ev, "censoring", Censor.NONE)))
                    row += 1
        assert row == pred.shape[0]
        return SubjectPredictions(out)

    def predictions(self) -> List[Prediction]:
        return self._p

    def get_predictions(self) -> List[Prediction]:
        """``Predictions::get_predictions`` (subject.rs:31-33)"""
        return list(self._p)

    def flat_predictions(self) -> List[float]:
        """subject.rs:145-148"""
        return [p.prediction for p in self._p]

    def flat_times(self) -> List[float]:
        return [p.time for p in self._p]

    def flat_observations(self) -> List[Optional[float]]:
        return [p.observation for p in self._p]

    def squared_error(self) -> float:
        """``Predictions::squared_error`` (subject.rs:24-29)"""
        return float(sum((p.observation - p.prediction) ** 2 for p in self._p if p.observation is not None))

    def __len__(self) -> int:
        return len(self._p)


class PopulationPredictions:
    """``PopulationPredictions { subject_predictions: Array2<SubjectPredictions> }`` (subject.rs:140-165): rows =
    subjects, columns = support points.  Built from ONE device pass (``Equation.population_predictions``)."""

    def __init__(self, subject_predictions: Sequence[Sequence[SubjectPredictions]]):
        self.subject_predictions = [list(r) for r in subject_predictions]

    @property
    def shape(self):
        return (len(self.subject_predictions), len(self.subject_predictions[0]) if self.subject_predictions else 0)

    def __getitem__(self, idx):
        s, p = idx
        return self.subject_predictions[s][p]

    @staticmethod
    def from_matrix(data, model, pred: np.ndarray, obs_off: np.ndarray) -> "PopulationPredictions":
        """``pred[n_observations, n_support]`` + per-subject row offsets -> the reference's container."""
        subjects = data.subjects if hasattr(data, "subjects") else list(data)
        rows = []
        for s, subj in enumerate(subjects):
            r0, r1 = int(obs_off[s]), int(obs_off[s + 1])
            rows.append([SubjectPredictions.from_flat(subj, model, pred[r0:r1, p]) for p in range(pred.shape[1])])
        return PopulationPredictions(rows)
