"""SciPy L-BFGS-B driver over torch parameters.

Public contract follows the reference's `Scipy` helper (cglb/backend/pytorch/optimizer.py:20-98): `minimize(closure,
variables, method, step_callback, **scipy_kwargs)` hands SciPy a float64 `(loss, flat gradient)` pair obtained with
`torch.autograd.grad(loss, variables)` (:95-98), writes every trial point back into the parameters before the closure runs
(:34-39) and calls `step_callback(step, variables, values)` once per accepted iterate (:51-61).  The implementation is
organised around a flat view of the parameter list computed once per call.
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Sequence

import numpy as np
import scipy.optimize
import torch


class _FlatView:
    """Offsets/shapes of a parameter list inside one flat float64 vector."""

    def __init__(self, tensors: Sequence[torch.Tensor]):
        self.tensors = tuple(tensors)
        self.shapes = [tuple(t.shape) for t in self.tensors]
        self.sizes = [int(np.prod(s)) if len(s) else 1 for s in self.shapes]
        self.offsets = np.concatenate([[0], np.cumsum(self.sizes)]).astype(int)

    def gather(self, tensors: Iterable[torch.Tensor]) -> torch.Tensor:
        # parameters may live on different devices (the hypers on the host, a trainable v on the GPU): pack on the host
        return torch.cat([t.detach().reshape(-1).to("cpu", torch.float64) for t in tensors])

    def split(self, flat: torch.Tensor) -> List[torch.Tensor]:
        return [flat[a:b].to(t.dtype).reshape(s) for a, b, s, t in zip(self.offsets[:-1], self.offsets[1:], self.shapes, self.tensors)]

    def write(self, values: Sequence[torch.Tensor]) -> None:
        if len(values) != len(self.tensors):
            raise ValueError("to_tensors and values should have same length")
        for dst, src in zip(self.tensors, values):
            dst.data = src.to(dst.device)


class Scipy:
    def minimize(self, closure: Callable[[], torch.Tensor], variables, method: str = "L-BFGS-B", step_callback=None, **scipy_kwargs):
        view = _FlatView(variables)
        x0 = view.gather(view.tensors).detach().cpu().numpy().astype(np.float64)

        def objective(x: np.ndarray):
            view.write(view.split(torch.from_numpy(np.array(x, dtype=np.float64))))
            loss, grads = _compute_loss_and_gradients(closure, view.tensors)
            return float(loss.detach().cpu()), view.gather(grads).detach().cpu().numpy().astype(np.float64)

        if step_callback is not None:
            counter = iter(range(1 << 62))

            def on_step(x: np.ndarray):
                step_callback(next(counter), view.tensors, view.split(torch.from_numpy(np.array(x, dtype=np.float64))))

            scipy_kwargs["callback"] = on_step
        return scipy.optimize.minimize(objective, x0, jac=True, method=method, **scipy_kwargs)

    # the reference exposes these helpers as static methods; kept for callers that use them
    @staticmethod
    def pack(tensors) -> torch.Tensor:
        return _FlatView(tensors).gather(tensors)

    @staticmethod
    def unpack(to_tensors, from_vector) -> List[torch.Tensor]:
        return _FlatView(to_tensors).split(from_vector)

    @staticmethod
    def assign(to_tensors, values) -> None:
        _FlatView(to_tensors).write(values)

    @staticmethod
    def to_numpy(tensor) -> np.ndarray:
        return tensor.detach().cpu().numpy()


def _compute_loss_and_gradients(loss_closure, variables):
    loss = loss_closure()
    return loss, torch.autograd.grad(loss, variables)
