"""Deterministic synthetic regression sets for benchmarks and examples.

The reference loads UCI sets through robustgp_experiments (cglb_experiments/datasets.py:47-76: z-normalise,
67/33 split); those need a network download, so the benchmarks use this generator of the same shape:
X ~ N(0,1)^{N x D} (already unit variance like datasets.py:35-39), y = sin(X a) + 0.1 eps z-normalised,
Z = first M rows of a seeded permutation of X (stand-in for robustgp's greedy conditional-variance init,
cglb/backend/config.py:62-65)."""
from __future__ import annotations

import math

import numpy as np


def synthetic_problem(N: int, D: int, M: int, seed: int = 0, dtype=np.float64):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    a = rng.standard_normal(D) / math.sqrt(D)
    y = np.sin(X @ a) + 0.1 * rng.standard_normal(N)
    y = (y - y.mean()) / y.std()
    perm = rng.permutation(N)
    Z = X[perm[:M]].copy()
    return X.astype(dtype), y.astype(dtype), Z.astype(dtype)


def reference_init_hypers(D: int):
    """cglb/backend/config.py:74-76 (variance=1, lengthscales=1) and :104-107 (noise=1); mean 0."""
    return dict(lengthscales=np.ones(D), variance=1.0, noise=1.0, mean=0.0)


def trained_like_hypers(D: int):
    """A point typical of a trained model: longer lengthscales, small noise (harder system, more CG steps)."""
    return dict(lengthscales=np.full(D, 1.5), variance=1.0, noise=0.05, mean=0.0)
