"""StopWatch and Logger — mirror of cglb/backend/callbacks.py:27-62, :76-178 without the TensorBoard sink
(observability is out of scope); the in-memory log keys are the reference's:
iteration, elapsed_time, params, loss, train|test/rmse|nlpd, cg/steps, cg/error, <stat>-per-feval."""
import time
from contextlib import contextmanager
from typing import Callable, Dict

import numpy as np


class StopWatch:
    def __init__(self):
        self._start_time = None
        self._pause_time = None
        self._total_paused_time = None

    def started(self) -> bool:
        return self._start_time is not None

    def start(self):
        self._start_time = time.time()
        self._total_paused_time = 0.0

    def pause(self):
        self._pause_time = time.time()

    def resume(self):
        self._total_paused_time += time.time() - self._pause_time
        self._pause_time = None

    def reset(self):
        self._start_time = None
        self._pause_time = None
        self._total_paused_time = None

    def get_elapsed_time(self):
        return time.time() - self._start_time - self._total_paused_time

    def stop(self):
        total = self.get_elapsed_time()
        self.reset()
        return total


class Logger:
    def __init__(self, logdir: str, metrics_fn: Callable, model_parameters_fn: Callable, holdout_interval: int = 10,
                 include_feval_log: bool = False, verbose: bool = True):
        self.holdout_interval = holdout_interval
        self.logdir = logdir
        self._metrics_fn = metrics_fn
        self._model_parameters_fn = model_parameters_fn
        self._logs = {}
        self.counter = 0
        self.include_feval_log = include_feval_log
        self.verbose = verbose
        self.timer = StopWatch()

    @property
    def logs(self) -> Dict:
        return self._logs

    def model_parameters_fn(self) -> Dict[str, np.ndarray]:
        return {k: v for k, v in self._model_parameters_fn().items() if "inducing_point" not in k}

    def metrics_fn(self) -> Dict[str, np.ndarray]:
        prefixes = ["train", "test", "cg/", "loss"]
        return {k: v for k, v in self._metrics_fn().items() if any(k.startswith(p) for p in prefixes)}

    def log(self, **kwargs):
        for k, v in kwargs.items():
            self._logs.setdefault(k, []).append(v)

    def log_for_feval(self, **kwargs):
        if self.include_feval_log:
            self.log(**{f"{k}-per-feval": v for k, v in kwargs.items()})

    @contextmanager
    def no_recording(self):
        saved = (self.holdout_interval, self.include_feval_log)
        self.holdout_interval, self.include_feval_log = -1, False
        try:
            yield
        finally:
            self.holdout_interval, self.include_feval_log = saved

    def __call__(self, step, *args):
        iteration = self.counter
        self.counter += 1
        if self.holdout_interval < 0:
            return
        if (iteration % self.holdout_interval) == 0:
            elapsed_time = self.timer.get_elapsed_time()
            self.timer.pause()
            params = self.model_parameters_fn()
            metrics = self.metrics_fn()
            if self.verbose:
                print(f"{iteration} - loss={metrics['loss']:.4f}", flush=True)
            self.log(iteration=iteration, elapsed_time=elapsed_time, params=params, **metrics)
            self.timer.resume()
