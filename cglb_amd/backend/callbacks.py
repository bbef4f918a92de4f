"""Timing and logging callbacks of the training loop.

API-compatible with the reference's `StopWatch` and `Logger` (cglb/backend/callbacks.py:27-62, :76-178) minus the TensorBoard
sink (observability is out of scope).  The in-memory log uses the reference's keys: iteration, elapsed_time,
params, loss, train|test/rmse|nlpd, cg/steps, cg/error and `<stat>-per-feval`."""
from __future__ import annotations

import time
from contextlib import contextmanager
from typing import Callable, Dict, List

_METRIC_PREFIXES = ("train", "test", "cg/", "loss")


class StopWatch:
    """Wall clock that can be paused while metrics are evaluated (callbacks.py:150-151,178 of the reference)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self._t0 = None        # start of the current running stretch
        self._banked = 0.0     # time accumulated in finished stretches
        self._running = False
        self._started = False

    def started(self) -> bool:
        return self._started

    def start(self):
        self._banked, self._t0 = 0.0, time.perf_counter()
        self._running = self._started = True

    def pause(self):
        if self._running:
            self._banked += time.perf_counter() - self._t0
            self._running = False

    def resume(self):
        if self._started and not self._running:
            self._t0, self._running = time.perf_counter(), True

    def get_elapsed_time(self) -> float:
        if not self._started:
            raise RuntimeError("StopWatch was not started")
        return self._banked + (time.perf_counter() - self._t0 if self._running else 0.0)

    def stop(self) -> float:
        total = self.get_elapsed_time()
        self.reset()
        return total


class Logger:
    def __init__(self, logdir: str, metrics_fn: Callable[[], Dict], model_parameters_fn: Callable[[], Dict], holdout_interval: int = 10,
                 include_feval_log: bool = False, verbose: bool = True):
        self.logdir = logdir
        self.holdout_interval = holdout_interval
        self.include_feval_log = include_feval_log
        self.verbose = verbose
        self.counter = 0
        self.timer = StopWatch()
        self._raw_metrics, self._raw_params = metrics_fn, model_parameters_fn
        self._logs: Dict[str, List] = {}

    @property
    def logs(self) -> Dict[str, List]:
        return self._logs

    def model_parameters_fn(self) -> Dict:
        """Parameter snapshot without the (large) inducing points."""
        return {k: v for k, v in self._raw_params().items() if "inducing_point" not in k}

    def metrics_fn(self) -> Dict:
        return {k: v for k, v in self._raw_metrics().items() if k.startswith(_METRIC_PREFIXES)}

    def log(self, **entries):
        for key, value in entries.items():
            self._logs.setdefault(key, []).append(value)

    def log_for_feval(self, **entries):
        """Per objective evaluation statistics (CG steps / residual), recorded only when asked for."""
        if self.include_feval_log:
            self.log(**{key + "-per-feval": value for key, value in entries.items()})

    @contextmanager
    def no_recording(self):
        state = (self.holdout_interval, self.include_feval_log)
        self.holdout_interval, self.include_feval_log = -1, False
        try:
            yield self
        finally:
            self.holdout_interval, self.include_feval_log = state

    def __call__(self, step, *args):
        """Optimizer step callback: every `holdout_interval`-th call evaluates the metrics with the clock paused."""
        iteration, self.counter = self.counter, self.counter + 1
        if self.holdout_interval < 0 or iteration % self.holdout_interval:
            return
        elapsed = self.timer.get_elapsed_time()
        self.timer.pause()
        try:
            metrics = self.metrics_fn()
            if self.verbose:
                print(f"{iteration} - loss={metrics['loss']:.4f}", flush=True)
            self.log(iteration=iteration, elapsed_time=elapsed, params=self.model_parameters_fn(), **metrics)
        finally:
            self.timer.resume()
