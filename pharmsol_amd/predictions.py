"""Result containers: ``Prediction`` / ``SubjectPredictions``.

Mirror of src/simulator/likelihood/prediction.rs:18-27 and subject.rs:19-21,105-148.
The device writes only ``pred``; time / observation / outeq / occasion are
re-attached here from the subject (what ``Observation::to_prediction`` does,
src/data/event.rs:698-711).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from .data import Observation, Subject


@dataclass
class Prediction:
    time: float
    observation: Optional[float]
    prediction: float
    outeq: int
    occasion: int


class SubjectPredictions:
    def __init__(self, predictions: List[Prediction]):
        self._p = predictions

    @staticmethod
    def from_flat(subject: Subject, model, pred: np.ndarray) -> "SubjectPredictions":
        out: List[Prediction] = []
        row = 0
        for occ in subject.occasions:
            # predictions are emitted in (processed) event order; without lag the order is
            # the stored sort order (equation/mod.rs:500-512)
            for ev in occ.events:
                if isinstance(ev, Observation):
                    out.append(Prediction(ev.time, ev.value, float(pred[row]), model.resolve_output_label(ev.outeq),
                                          occ.index))
                    row += 1
        assert row == pred.shape[0]
        return SubjectPredictions(out)

    def predictions(self) -> List[Prediction]:
        return self._p

    def flat_predictions(self) -> List[float]:
        """subject.rs:145-148"""
        return [p.prediction for p in self._p]

    def flat_times(self) -> List[float]:
        return [p.time for p in self._p]

    def flat_observations(self) -> List[Optional[float]]:
        return [p.observation for p in self._p]

    def __len__(self) -> int:
        return len(self._p)
