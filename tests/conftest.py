import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    _ensure_built()


def _ensure_built():
    """Build the native pieces if a fresh checkout has none yet (hipcc cross-compiles without a GPU; the built .so files
    travel to the GPU box with the snapshot).  Same recipe as __graft_entry__.build()."""
    import subprocess
    hip_so = os.path.join(ROOT, "cglb_amd", "lib", "libcglb_hip.so")
    orc_so = os.path.join(ROOT, "oracle", "_build", "libcglb_oracle.so")
    if not os.path.exists(hip_so):
        subprocess.run(["make", "-j", str(min(8, os.cpu_count() or 1)), "-C", os.path.join(ROOT, "cglb_amd", "csrc")], check=True,
                       capture_output=True)
    if not os.path.exists(orc_so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    return g


def golden_hypers(g):
    from oracle import cglb_oracle as orc
    return orc.Hypers(lengthscales=g["lengthscales"].copy(), variance=float(g["variance"]), noise=float(g["noise"]),
                      mean=float(g["mean"]), Z=g["Z"].copy(), jitter=float(g["jitter"]))
