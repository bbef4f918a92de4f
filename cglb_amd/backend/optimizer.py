"""SciPy L-BFGS-B wrapper over torch parameters — mirror of cglb/backend/pytorch/optimizer.py:20-98
(pack/unpack/assign and `torch.autograd.grad(loss, variables)`; SciPy receives float64 loss and flat gradient)."""
import numpy as np
import scipy.optimize
import torch


class Scipy:
    def minimize(self, closure, variables, method="L-BFGS-B", step_callback=None, **scipy_kwargs):
        variables = tuple(variables)
        init_vals = self.to_numpy(self.pack(variables))
        func = self.eval_func(closure, variables)
        if step_callback is not None:
            scipy_kwargs.update(dict(callback=self.callback_func(variables, step_callback)))
        return scipy.optimize.minimize(func, init_vals, jac=True, method=method, **scipy_kwargs)

    @classmethod
    def eval_func(cls, closure, variables):
        def _eval(x):
            values = cls.unpack(variables, torch.from_numpy(np.asarray(x, dtype=np.float64)))
            cls.assign(variables, values)
            loss, grads = _compute_loss_and_gradients(closure, variables)
            return (loss.cpu().detach().numpy().astype(np.float64), cls.pack(grads).cpu().detach().numpy().astype(np.float64))

        return _eval

    @classmethod
    def callback_func(cls, variables, step_callback):
        step = 0

        def _callback(x):
            nonlocal step
            values = cls.unpack(variables, torch.from_numpy(np.asarray(x, dtype=np.float64)))
            step_callback(step, variables, values)
            step += 1

        return _callback

    @staticmethod
    def pack(tensors):
        return torch.cat([torch.flatten(t) for t in tensors], axis=0)

    @staticmethod
    def to_numpy(tensor):
        return tensor.detach().cpu().numpy()

    @staticmethod
    def unpack(to_tensors, from_vector):
        s, values = 0, []
        for target in to_tensors:
            size = int(np.prod(tuple(target.shape))) if target.ndim > 0 else 1
            values.append(torch.reshape(from_vector[s: s + size].type(target.dtype), tuple(target.shape)))
            s += size
        return values

    @staticmethod
    def assign(to_tensors, values):
        if len(to_tensors) != len(values):
            raise ValueError("to_tensors and values should have same length")
        for target, value in zip(to_tensors, values):
            target.data = value


def _compute_loss_and_gradients(loss_closure, variables):
    loss = loss_closure()
    grads = torch.autograd.grad(loss, variables)
    return loss, grads
