"""Backend facade and registry with the new key "hip".

Same surface as the reference facade (cglb/backend/backend.py:34-115): a `Backend` class whose classmethods forward to an
`interface` module, and a `BACKENDS` registry the CLI indexes with `-b`.  The forwarding methods are generated from the list
of interface functions instead of being written out one by one.  Plugging into the reference:
`BACKENDS["hip"] = cglb_amd.backend.Hip`."""
from abc import ABC, abstractmethod

from . import interface as _hip_interface

__all__ = ["Backend", "Hip", "BACKENDS"]

#: interface functions reachable as Backend.<name>(...)  (backend.py:40-91 of the reference)
FORWARDED = ("configure_backend", "create_kernel", "create_model", "model_parameters", "optimize", "save", "load", "metrics_fn",
             "set_default_float", "get_default_float_str", "get_default_float")


class Backend(ABC):
    @staticmethod
    @abstractmethod
    def interface():
        """The module that implements the backend."""

    @classmethod
    def set_default_jitter(cls, float_type: str):
        # Cholesky jitter by precision: 1e-5 for fp32, 1e-6 otherwise (reference backend.py:77-79)
        return cls.interface().set_default_jitter({"fp32": 1e-5}.get(float_type, 1e-6))


def _forward(name):
    def method(cls, *args, **kwargs):
        return getattr(cls.interface(), name)(*args, **kwargs)

    method.__name__ = name
    method.__doc__ = f"Forwards to `interface().{name}`."
    return classmethod(method)


for _name in FORWARDED:
    setattr(Backend, _name, _forward(_name))


class Hip(Backend):
    @staticmethod
    def interface():
        return _hip_interface


BACKENDS = {"hip": Hip}
