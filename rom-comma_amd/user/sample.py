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


def synthetic_cv_fold(N: int, M: int, k: int, K: int = 8, l: int = 0, noise: float = 0.04) -> Tuple[np.ndarray, np.ndarray]:
    """Training rows of fold ``k`` of a K-fold split of ONE seeded synthetic dataset, sized so that every fold trains on
    exactly N rows: the dataset has N + h rows, h = ceil(N/(K-1)), and fold k leaves out the k-th block of h rows (the last
    block is shifted back to fit) -- what ``Repository.into_K_folds(-K)`` produces (data/storage.py:162-204, blocks instead of the reference's
    unseeded shuffle). Folds overlap in (K-2)/(K-1) of their rows, as real cross-validation folds do, so their fits cost
    about the same; normalisation uses the whole dataset, like the repository-level normalization.csv."""
    if not (0 <= k < K) or K < 2:
        raise ValueError(f'need 0 <= k < K and K >= 2, got k={k}, K={K}')
    held = -(-N // (K - 1))
    X, y = synthetic_fold(N + held, M, k=0, l=l, noise=noise)
    start = min(k * held, N)
    keep = np.ones(N + held, dtype=bool)
    keep[start:start + held] = False
    return np.ascontiguousarray(X[keep]), np.ascontiguousarray(y[keep])


def synthetic_outputs(N: int, M: int, L: int, k: int = 0, noise: float = 0.04) -> Tuple[np.ndarray, np.ndarray]:
    """One design X (N, M) shared by L outputs Y (N, L) (BASELINE configs[3]: independent-output GPs on the same inputs).
    Output l rotates the relevance hierarchy: f_l = sum_m sin(2 pi U_{(m+l) mod M}) / (m+1) + 0.5 U_l U_{l+1}."""
    rng = np.random.Generator(np.random.PCG64(20240807 + 1000 * k))
    U = rng.random((N, M))
    X = scipy.stats.norm.ppf(np.clip(U, UNIFORM_MARGIN, 1 - UNIFORM_MARGIN))
    Y = np.empty((N, L))
    for l in range(L):
        f = np.zeros(N)
        for m in range(M):
            f += np.sin(2 * np.pi * U[:, (m + l) % M]) / (m + 1)
        if M > 1:
            f += 0.5 * U[:, l % M] * U[:, (l + 1) % M]
        f = (f - f.mean()) / f.std()
        y = f + noise * np.random.Generator(np.random.PCG64(20240807 + 1000 * k + 7 * (l + 1))).standard_normal(N)
        Y[:, l] = (y - y.mean()) / y.std()
    return np.ascontiguousarray(X), np.ascontiguousarray(Y)
