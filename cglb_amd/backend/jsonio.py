"""JSON artefacts in the encoding the reference writes them in.

The reference dumps `model.json` (pytorch/interface.py:546-551), `results.json` and `logs.json` (cglb_experiments/cli.py:100-109)
with the third-party `json_tricks` (requirements.txt, un-pinned, absent here) and reads them back with `json_tricks.load`
(cglb_experiments/plotting.py:223-230).  This module restates the published json_tricks 3.x numpy encoding so that those readers
get the same keys, dtypes and shapes from files written here:

    ndarray       -> {"__ndarray__": nested lists (a bare number for 0-d), "dtype": "float64", "shape": [..], "Corder": true}
                     ("Corder" only for arrays of two or more dimensions)
    numpy scalar  -> plain JSON number

and the matching decoder (the `__ndarray__` object hook).  Parity with json_tricks itself is unpinned (the package cannot be
imported in the build container); tests/test_backend_host.py holds the schema derived from the reference's writer lines."""
from __future__ import annotations

import json
from typing import Any, IO

import numpy as np


def _encode(obj: Any) -> Any:
    if isinstance(obj, dict):
        return {str(k): _encode(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_encode(v) for v in obj]
    if isinstance(obj, np.ndarray):
        out = {"__ndarray__": obj.tolist(), "dtype": str(obj.dtype), "shape": list(obj.shape)}
        if obj.ndim > 1:
            out["Corder"] = bool(obj.flags["C_CONTIGUOUS"])
        return out
    if isinstance(obj, np.generic):
        return obj.item()
    if hasattr(obj, "detach") and hasattr(obj, "cpu"):  # torch tensor: the reference converts to numpy before dumping
        return _encode(obj.detach().cpu().numpy())
    return obj


def _decode_hook(dct: dict) -> Any:
    if "__ndarray__" in dct:
        order = "C" if dct.get("Corder", True) else "F"
        arr = np.asarray(dct["__ndarray__"], dtype=dct.get("dtype", None), order=order)
        shape = dct.get("shape")
        return arr.reshape(shape) if shape is not None else arr
    return dct


def dumps(obj: Any, **kwargs) -> str:
    return json.dumps(_encode(obj), **kwargs)


def dump(obj: Any, file: IO[str], **kwargs) -> None:
    json.dump(_encode(obj), file, **kwargs)


def loads(text: str) -> Any:
    return json.loads(text, object_hook=_decode_hook)


def load(file) -> Any:
    if isinstance(file, (str, bytes)) or hasattr(file, "__fspath__"):
        with open(file) as f:
            return json.load(f, object_hook=_decode_hook)
    return json.load(file, object_hook=_decode_hook)
