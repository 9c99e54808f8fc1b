"""Seeded synthetic folds. The reference's samplers (user/sample.py:49-254) draw an unseeded Latin
hypercube and unseeded noise, and its fold split shuffles without a seed (data/storage.py:184,195), so no reference run is
reproducible; benchmarks and tests here use this seeded stand-in of the same shape: inputs uniform then probit-normalised
exactly as Normalization.apply_to does (data/storage.py:476-483), output a smooth function with a relevance hierarchy,
z-scored, plus Gaussian noise, re-standardised."""
from __future__ import annotations

from typing import Tuple

import numpy as np
import scipy.stats

UNIFORM_MARGIN = 1.0e-12       # data/storage.py:448-449


def synthetic_fold(N: int, M: int, k: int = 0, l: int = 0, noise: float = 0.04) -> Tuple[np.ndarray, np.ndarray]:
    """Normalised training inputs X (N, M) and one output column y (N,) for fold ``k``, output ``l``."""
    rng = np.random.Generator(np.random.PCG64(20240807 + 1000 * k + l))
    U = rng.random((N, M))
    X = scipy.stats.norm.ppf(np.clip(U, UNIFORM_MARGIN, 1 - UNIFORM_MARGIN))
    f = np.zeros(N)
    for m in range(M):
        f += np.sin(2 * np.pi * U[:, m]) / (m + 1)
    if M > 1:
        f += 0.5 * U[:, 0] * U[:, 1]
    f = (f - f.mean()) / f.std()
    y = f + noise * rng.standard_normal(N)
    y = (y - y.mean()) / y.std()
    return np.ascontiguousarray(X), np.ascontiguousarray(y)


def bench_hyper(M: int) -> Tuple[np.ndarray, float, float]:
    """Fixed hyper-parameters for kernel-level benchmarks: ell_m = 0.5 + 3.5 m/(M-1), variance 1, noise 1.6e-3."""
    return 0.5 + 3.5 * np.arange(M) / max(M - 1, 1), 1.0, 1.6e-3
