"""The reference's own test functions with PUBLISHED analytic Sobol indices -- known answers for the whole path (fit + closed-form Sobol).

romcomma/user/functions.py:126-152 takes them from SALib (not installed here; its formulas are the published ones restated below):
  ISHIGAMI 'standard'   f = sin x1 + A sin^2 x2 + B x3^4 sin x1,  x = -pi + 2 pi u,  A = 7, B = 0.1         (functions.py:126,144)
  SOBOL_G  'weak5_2'    g = prod_i ((1 + alpha) |2 u_i - 1|^alpha + a_i) / (1 + a_i),  a = (3, 6, 9, 18, 27), alpha = 2   (functions.py:127,150)
with u ~ U(0, 1)^M.  The reference feeds its GPs z = Phi^-1(u) (data/storage.py Normalization) and integrates the GP against
N(0, I): a monotone map per input leaves every Sobol index unchanged, so the GP's closed-form indices estimate the analytic ones.
Test infrastructure only."""
from typing import Dict, Tuple

import numpy as np
from scipy.special import ndtri


def ishigami(u: np.ndarray, A: float = 7.0, B: float = 0.1) -> np.ndarray:
    x = -np.pi + 2.0 * np.pi * u
    return np.sin(x[:, 0]) + A * np.sin(x[:, 1]) ** 2 + B * x[:, 2] ** 4 * np.sin(x[:, 0])


def ishigami_variances(A: float = 7.0, B: float = 0.1) -> Dict[Tuple[int, ...], float]:
    """Partial variances of the ANOVA decomposition (Ishigami & Homma 1990; Saltelli et al. 2000): V_1, V_2, V_13, everything else 0."""
    pi = np.pi
    return {(0,): 0.5 * (1.0 + B * pi ** 4 / 5.0) ** 2, (1,): A * A / 8.0, (0, 2): B * B * pi ** 8 * (1.0 / 18.0 - 1.0 / 50.0)}


def sobol_g(u: np.ndarray, a=(3.0, 6.0, 9.0, 18.0, 27.0), alpha: float = 2.0) -> np.ndarray:
    a = np.asarray(a, dtype=float)
    return np.prod(((1.0 + alpha) * np.abs(2.0 * u[:, :a.size] - 1.0) ** alpha + a) / (1.0 + a), axis=1)


def sobol_g_variances(a=(3.0, 6.0, 9.0, 18.0, 27.0), alpha: float = 2.0) -> Dict[Tuple[int, ...], float]:
    """V_S = prod_{i in S} V_i with V_i = alpha^2 / ((1 + 2 alpha) (1 + a_i)^2) (Saltelli & Sobol' 1995 for the modified G-function)."""
    a = np.asarray(a, dtype=float)
    Vi = alpha ** 2 / ((1.0 + 2.0 * alpha) * (1.0 + a) ** 2)
    out: Dict[Tuple[int, ...], float] = {}
    for mask in range(1, 1 << a.size):
        S = tuple(i for i in range(a.size) if mask >> i & 1)
        out[S] = float(np.prod(Vi[list(S)]))
    return out


def analytic_indices(partial: Dict[Tuple[int, ...], float], M: int) -> Dict[str, np.ndarray]:
    """The reference's three kinds (gsa/models.py:77-90, 207-214) from the partial variances: first_order[m] = index of input m alone,
    closed[m] = closed index of the inputs 0..m, total[m] = 1 - closed index of the inputs m+1..M-1 (the total index of 0..m)."""
    V = sum(partial.values())

    def closed(members) -> float:
        members = set(members)
        return sum(v for S, v in partial.items() if set(S) <= members) / V

    return {'first_order': np.array([closed([m]) for m in range(M)]),
            'closed': np.array([closed(range(m + 1)) for m in range(M)]),
            'total': np.array([1.0 - closed(range(m + 1, M)) for m in range(M)])}


def sample(function, N: int, M: int, seed: int, noise: float = 0.0) -> Tuple[np.ndarray, np.ndarray]:
    """N points of a Latin hypercube in (0, 1)^M (the reference's DOE, installation_test.py:37), the GP inputs z = Phi^-1(u) and the
    z-scored function values (+ noise-to-signal ratio `noise`, user/sample.py:224-226)."""
    rng = np.random.default_rng(seed)
    u = (np.stack([rng.permutation(N) for _ in range(M)], axis=1) + rng.random((N, M))) / N
    f = function(u)
    y = (f - f.mean()) / f.std()
    if noise > 0.0:
        y = y + noise * rng.standard_normal(N)
        y = (y - y.mean()) / y.std()
    return ndtri(np.clip(u, 1e-12, 1.0 - 1e-12)), y
