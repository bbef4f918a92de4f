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


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    return g


def golden_hypers(g):
    from oracle import cglb_oracle as orc
    return orc.Hypers(lengthscales=g["lengthscales"].copy(), variance=float(g["variance"]), noise=float(g["noise"]),
                      mean=float(g["mean"]), Z=g["Z"].copy(), jitter=float(g["jitter"]))
