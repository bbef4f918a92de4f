"""Backend facade and registry — mirror of cglb/backend/backend.py:34-115 with the new key "hip".

Plugging into the reference: `BACKENDS["hip"] = cglb_amd.backend.Hip`; the CLI then selects it
with `-b hip` exactly like `-b torch`."""
from abc import ABC, abstractmethod
from typing import Dict, Tuple

import numpy as np

from . import interface as _hip_interface

Data = Tuple[np.ndarray, np.ndarray]
Dataset = Tuple[Data, Data]

__all__ = ["Backend", "Hip", "BACKENDS"]


class Backend(ABC):
    @staticmethod
    @abstractmethod
    def interface():
        pass

    @classmethod
    def configure_backend(cls, **kwargs):
        return cls.interface().configure_backend(**kwargs)

    @classmethod
    def create_kernel(cls, cfg, data: Data):
        return cls.interface().create_kernel(cfg, data)

    @classmethod
    def create_model(cls, model_cfg, data: Data):
        return cls.interface().create_model(model_cfg, data)

    @classmethod
    def model_parameters(cls, model) -> Dict[str, np.ndarray]:
        return cls.interface().model_parameters(model)

    @classmethod
    def optimize(cls, model, dataset: Dataset, num_steps: int, logger, optimizer: str):
        return cls.interface().optimize(model, dataset, num_steps, logger, optimizer)

    @classmethod
    def save(cls, model, logdir: str):
        return cls.interface().save(model, logdir)

    @classmethod
    def load(cls, model, filepath: str):
        return cls.interface().load(model, filepath)

    @classmethod
    def metrics_fn(cls, model, dataset_bundle):
        return cls.interface().metrics_fn(model, dataset_bundle)

    @classmethod
    def set_default_float(cls, float_type: str):
        return cls.interface().set_default_float(float_type)

    @classmethod
    def set_default_jitter(cls, float_type: str):
        value = 1e-5 if float_type == "fp32" else 1e-6  # backend.py:77-79
        return cls.interface().set_default_jitter(value)

    @classmethod
    def get_default_float_str(cls):
        return cls.interface().get_default_float_str()

    @classmethod
    def get_default_float(cls):
        return cls.interface().get_default_float()


class Hip(Backend):
    @staticmethod
    def interface():
        return _hip_interface


BACKENDS = {"hip": Hip}
