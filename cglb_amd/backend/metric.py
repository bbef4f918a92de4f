"""Metric reducers — mirror of cglb/backend/metric.py:24-55."""
from typing import Callable, Dict

import numpy as np


def call_metric_fns(*fns: Callable[[], Dict]) -> Dict[str, float]:
    out = {}
    for fn in fns:
        out.update({k: float(np.array(v)) for k, v in fn().items()})
    return out


def rmse_and_lpd_fn(error_logdensity_cb: Callable) -> Callable[[], Dict[str, float]]:
    def inner_func() -> Dict[str, float]:
        errs, logdens = error_logdensity_cb()
        (train_errors, test_errors), (train_lds, test_lds) = [np.array(e) for e in errs], [np.array(l) for l in logdens]
        metrics = {
            "train/rmse": np.sqrt(np.mean(train_errors ** 2)),
            "test/rmse": np.sqrt(np.mean(test_errors ** 2)),
            "train/nlpd": -np.mean(train_lds),
            "test/nlpd": -np.mean(test_lds),
        }
        return {k: float(v) for k, v in metrics.items()}

    return inner_func
