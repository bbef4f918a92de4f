"""RMSE / NLPD reducers with the reference's metric names (cglb/backend/metric.py:24-55)."""
from typing import Callable, Dict

import numpy as np


def call_metric_fns(*fns: Callable[[], Dict]) -> Dict[str, float]:
    """Evaluates the callbacks in order and merges their scalar results (later ones win)."""
    merged: Dict[str, float] = {}
    for fn in fns:
        for key, value in fn().items():
            merged[key] = float(np.asarray(value))
    return merged


def rmse_and_lpd_fn(error_logdensity_cb: Callable) -> Callable[[], Dict[str, float]]:
    """error_logdensity_cb() -> ((train_err, test_err), (train_logdensity, test_logdensity))."""

    def reduce() -> Dict[str, float]:
        errors, logdens = error_logdensity_cb()
        out = {}
        for split, err, lpd in zip(("train", "test"), errors, logdens):
            out[f"{split}/rmse"] = float(np.sqrt(np.mean(np.square(np.asarray(err)))))
            out[f"{split}/nlpd"] = float(-np.mean(np.asarray(lpd)))
        return out

    return reduce
