"""CPU oracle for the romcomma GP-regression + closed-form Sobol hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product path (``rom-comma_amd``) never routes through here.

PARITY UNPINNED.  The reference (``/root/reference``, romcomma @ 2024_08_07) delegates this path to GPflow/TensorFlow,
neither of which is installed in this image, and it ships no golden vectors, fixtures or assertions for the path
(SURVEY.md section 8c).  This file is therefore a restatement, in NumPy/SciPy fp64, of
  * the reference's own arithmetic where the reference owns it (the Sobol code, written as raw TF ops), and
  * the published GPflow 2.x algorithm at the reference's call sites where GPflow owns it (kernel, LML, predict, optimiser),
cross-checked internally (literal-broadcast transliteration vs closed form, analytic gradient vs finite differences,
mpmath 50-digit evaluation, invariants) in ``tests/test_oracle.py``.

Every function cites the reference lines it follows; paths are relative to ``/root/reference/romcomma/``.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np
import scipy.linalg
import scipy.optimize
import scipy.special

LOG_2PI = math.log(2.0 * math.pi)

# gpr/models.py:62-65 (Likelihood.VARIANCE_FLOOR) and gpr/kernels.py:176 (kernel variance floor).
LIKELIHOOD_VARIANCE_FLOOR = 1.0001e-6
KERNEL_VARIANCE_FLOOR = 1.0005e-6
# GPflow's Gaussian likelihood variance lower bound (variance = 1e-6 + softplus(u)); SURVEY.md section 3.2.
GPFLOW_LIKELIHOOD_LOWER = 1.0e-6


# --------------------------------------------------------------------------------------------------------------------
# Synthetic inputs.  The reference seeds nothing (data/storage.py:184,195; user/sample.py:67,143,182).
# --------------------------------------------------------------------------------------------------------------------

def synthetic_fold(N: int, M: int, k: int = 0, l: int = 0, noise: float = 0.04) -> Tuple[np.ndarray, np.ndarray]:
    """Seeded stand-in for one normalised training fold: X ~ probit(U), y = zscore(zscore(f(U)) + noise*eps).

    Mimics Normalization.apply_to (data/storage.py:476-483): X uniform -> clip(1e-12) -> norm.ppf ; Y z-scored.
    """
    import scipy.stats
    rng = np.random.Generator(np.random.PCG64(20240807 + 1000 * k + l))
    U = rng.random((N, M))
    X = scipy.stats.norm.ppf(np.clip(U, 1e-12, 1 - 1e-12))
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
    """Fixed-theta set for kernel benchmarks: ell_m = 0.5 + 3.5 m/(M-1), var 1.0, noise 1.6e-3."""
    ell = 0.5 + 3.5 * np.arange(M) / max(M - 1, 1)
    return ell, 1.0, 1.6e-3


# --------------------------------------------------------------------------------------------------------------------
# Parameter transforms: GPflow's positive() = softplus; Gaussian likelihood adds a 1e-6 shift.
# --------------------------------------------------------------------------------------------------------------------

def softplus(u):
    u = np.asarray(u, dtype=np.float64)
    return np.logaddexp(0.0, u)


def inv_softplus(x):
    x = np.asarray(x, dtype=np.float64)
    return x + np.log(-np.expm1(-x))


def sigmoid(u):
    return scipy.special.expit(np.asarray(u, dtype=np.float64))


# --------------------------------------------------------------------------------------------------------------------
# Kernel, Cholesky, LML, gradient, predict  (gpr/kernels.py:165-180, gpr/models.py:332-343, 345-384, 427-444 + GPflow)
# --------------------------------------------------------------------------------------------------------------------

def square_distance(A: np.ndarray, B: np.ndarray | None = None) -> np.ndarray:
    """GPflow ``square_distance``: -2 A B^T + |a|^2 + |b|^2, no clamp (SURVEY.md section 7 'quirks')."""
    if B is None:
        s = np.sum(A * A, axis=-1)
        return -2.0 * (A @ A.T) + s[:, None] + s[None, :]
    sa = np.sum(A * A, axis=-1)
    sb = np.sum(B * B, axis=-1)
    return -2.0 * (A @ B.T) + sa[:, None] + sb[None, :]


def gram(X: np.ndarray, ell: np.ndarray, var: float, X2: np.ndarray | None = None) -> np.ndarray:
    """ARD-RBF Gram matrix, gf.kernels.RBF(variance, lengthscales).K (gpr/kernels.py:176; gpr/models.py:435)."""
    ell = np.broadcast_to(np.asarray(ell, dtype=np.float64), (X.shape[1],))
    Z = X / ell
    Z2 = None if X2 is None else X2 / ell
    return var * np.exp(-0.5 * square_distance(Z, Z2))


def noisy_gram(X, ell, var, noise) -> np.ndarray:
    """K + noise I via set_diag (gpr/models.py:435-437)."""
    K = gram(X, ell, var)
    K[np.diag_indices_from(K)] += noise
    return K


def k_cho(X, ell, var, noise) -> np.ndarray:
    """MOGP.K_cho for one independent output: lower Cholesky of K + noise I (gpr/models.py:427-439)."""
    return scipy.linalg.cholesky(noisy_gram(X, ell, var, noise), lower=True, check_finite=False)


def k_inv_y(X, y, ell, var, noise) -> np.ndarray:
    """MOGP.K_inv_Y for one output: cholesky_solve(K_cho, y) (gpr/models.py:441-444)."""
    L = k_cho(X, ell, var, noise)
    return scipy.linalg.cho_solve((L, True), y, check_finite=False)


def lml(X, y, ell, var, noise) -> float:
    """GPflow GPR.log_marginal_likelihood = multivariate_normal(y, 0, chol(K + noise I)) (gpr/models.py:360,365,370):
    -1/2 |L^-1 y|^2 - N/2 log 2pi - sum log L_ii."""
    L = k_cho(X, ell, var, noise)
    w = scipy.linalg.solve_triangular(L, y, lower=True, check_finite=False)
    return float(-0.5 * w @ w - 0.5 * len(y) * LOG_2PI - np.sum(np.log(np.diag(L))))


def lml_and_grad(X, y, ell, var, noise) -> Tuple[float, np.ndarray]:
    """LML and its gradient w.r.t. the CONSTRAINED parameters (ell_1..ell_M [or one isotropic ell], var, noise).

    The reference obtains the gradient from TF autodiff inside gf.optimizers.Scipy (gpr/models.py:359-361); the analytic
    form is SURVEY.md Appendix A:  W = alpha alpha^T - K_n^-1 ;  dLML/dell_m = 1/2 sum W_ij K_ij (x_im-x_jm)^2 / ell_m^3 ;
    dLML/dvar = 1/2 sum W_ij K_ij / var ; dLML/dnoise = 1/2 tr W.
    """
    N, M = X.shape
    ell = np.atleast_1d(np.asarray(ell, dtype=np.float64))
    isotropic = ell.shape[0] == 1 and M > 1
    ell_full = np.broadcast_to(ell, (M,)) if isotropic else ell
    K = gram(X, ell_full, var)
    Kn = K.copy()
    Kn[np.diag_indices(N)] += noise
    L = scipy.linalg.cholesky(Kn, lower=True, check_finite=False)
    w = scipy.linalg.solve_triangular(L, y, lower=True, check_finite=False)
    value = float(-0.5 * w @ w - 0.5 * N * LOG_2PI - np.sum(np.log(np.diag(L))))
    alpha = scipy.linalg.solve_triangular(L, w, lower=True, trans='T', check_finite=False)
    Kinv = scipy.linalg.cho_solve((L, True), np.eye(N), check_finite=False)
    W = np.outer(alpha, alpha) - Kinv
    WK = W * K
    g_ell = np.empty(M)
    for m in range(M):
        d = X[:, m][:, None] - X[:, m][None, :]
        g_ell[m] = 0.5 * np.sum(WK * d * d) / ell_full[m] ** 3
    if isotropic:
        g_ell = np.array([g_ell.sum()])
    g_var = 0.5 * np.sum(WK) / var
    g_noise = 0.5 * np.trace(W)
    return value, np.concatenate([g_ell, [g_var, g_noise]])


def lml_and_grad_blas(X, y, ell, var, noise) -> Tuple[float, np.ndarray]:
    """The same LML and gradient as ``lml_and_grad`` (same formulas) arranged for a many-core host: LAPACK
    potrf + potri for K_n^-1 (N^3 flops in all, multi-threaded) and BLAS-3 products for the gradient sums instead of M
    Python-level passes over N x N temporaries. This is what ``bench.py``'s ``cpu_baseline`` times -- the reference's own CPU
    path (GPflow on TensorFlow: Cholesky + reverse-mode autodiff through it) does strictly more arithmetic, so the baseline is
    conservative. With r = (W o K) 1 and W = alpha alpha^T - K_n^-1:
        sum_ij W_ij K_ij (x_im - x_jm)^2 = 2 sum_i x_im^2 r_i - 2 x_m^T (W o K) x_m ,
        (W o K) V = alpha o (K (alpha o V)) - (K_n^-1 o K) V        for any N x c matrix V (here V = [X, 1]).
    Only the lower triangle of K_n^-1 is formed (dpotri) and used (dsymm)."""
    from scipy.linalg import blas, lapack
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    N, M = X.shape
    ell = np.atleast_1d(np.asarray(ell, dtype=np.float64))
    isotropic = ell.shape[0] == 1 and M > 1
    ell_full = np.broadcast_to(ell, (M,)) if isotropic else ell
    K = gram(X, ell_full, var)
    A = np.array(K, order='F')
    A[np.diag_indices(N)] += noise
    c, info = lapack.dpotrf(A, lower=1, overwrite_a=1)
    if info != 0:
        raise np.linalg.LinAlgError(f'leading minor {info} is not positive definite')
    w = scipy.linalg.solve_triangular(c, y, lower=True, check_finite=False)
    value = float(-0.5 * w @ w - 0.5 * N * LOG_2PI - np.sum(np.log(np.diag(c))))
    alpha = scipy.linalg.solve_triangular(c, w, lower=True, trans='T', check_finite=False)
    Kinv, info = lapack.dpotri(c, lower=1, overwrite_c=1)               # lower triangle of K_n^-1
    trace_Kinv = float(np.trace(Kinv))
    np.multiply(Kinv, K, out=Kinv)                                      # K_n^-1 o K (lower triangle meaningful)
    V = np.concatenate([X, np.ones((N, 1))], axis=1)
    P = K @ (alpha[:, None] * V)                                        # K (alpha o V)
    Q = blas.dsymm(1.0, Kinv, np.asfortranarray(V), side=0, lower=1)    # (K_n^-1 o K) V from the lower triangle
    WKV = alpha[:, None] * P - Q                                        # (W o K) [X, 1]
    r = WKV[:, M]
    quad = np.einsum('im,im->m', X, WKV[:, :M])                         # x_m^T (W o K) x_m
    g_ell = 0.5 * (2.0 * (X * X).T @ r - 2.0 * quad) / ell_full ** 3
    if isotropic:
        g_ell = np.array([g_ell.sum()])
    g_var = 0.5 * np.sum(r) / var
    g_noise = 0.5 * (alpha @ alpha - trace_Kinv)
    return value, np.concatenate([g_ell, [g_var, g_noise]])


def pack_unconstrained(ell, var, noise) -> np.ndarray:
    """theta -> u, GPflow parametrisation: ell, var = softplus(u); noise = 1e-6 + softplus(u)."""
    ell = np.atleast_1d(np.asarray(ell, dtype=np.float64))
    return np.concatenate([inv_softplus(ell), [inv_softplus(var)], [inv_softplus(noise - GPFLOW_LIKELIHOOD_LOWER)]])


def unpack_unconstrained(u: np.ndarray) -> Tuple[np.ndarray, float, float]:
    u = np.asarray(u, dtype=np.float64)
    return softplus(u[:-2]), float(softplus(u[-2])), float(GPFLOW_LIKELIHOOD_LOWER + softplus(u[-1]))


def neg_lml_unconstrained(u: np.ndarray, X, y) -> Tuple[float, np.ndarray]:
    """The objective gf.optimizers.Scipy hands to scipy.optimize.minimize: training_loss = -LML(u), with its gradient."""
    ell, var, noise = unpack_unconstrained(u)
    value, grad = lml_and_grad(X, y, ell, var, noise)
    return -value, -grad * sigmoid(u)


def fit(X, y, ell0, var0=2.0, noise0=0.02, maxiter: int = 5000, gtol: float = 1e-16, callback=None):
    """MOGP.calibrate for one independent output (gpr/models.py:345-373): L-BFGS-B on -LML(u), options maxiter/gtol
    (gpr/models.py:327-330); start point defaults var 2.0, ell 5.0 (gpr/kernels.py:49-50), noise 0.02 (gpr/models.py:52),
    floored as in gpr/models.py:341 and gpr/kernels.py:176."""
    u0 = pack_unconstrained(ell0, max(var0, KERNEL_VARIANCE_FLOOR), max(noise0, LIKELIHOOD_VARIANCE_FLOOR))
    res = scipy.optimize.minimize(neg_lml_unconstrained, u0, args=(X, y), jac=True, method='L-BFGS-B',
                                  options={'maxiter': maxiter, 'gtol': gtol}, callback=callback)
    ell, var, noise = unpack_unconstrained(res.x)
    return {'ell': ell, 'var': var, 'noise': noise, 'lml': -float(res.fun), 'nfev': int(res.nfev), 'nit': int(res.nit),
            'result': res}


def predict(X, y, ell, var, noise, Xs, y_instead_of_f: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """MOGP.predict for one output (gpr/models.py:375-384) = GPflow predict_y/predict_f -> base_conditional:
    A = L^-1 K* ; var = k** - colsum(A^2) (+ noise) ; mean = A^T L^-1 y.  Returns (mean, sqrt(var)): SD, not variance (:384)."""
    L = k_cho(X, ell, var, noise)
    Ks = gram(X, ell, var, Xs)                                  # (N, o)
    A = scipy.linalg.solve_triangular(L, Ks, lower=True, check_finite=False)
    w = scipy.linalg.solve_triangular(L, y, lower=True, check_finite=False)
    mean = A.T @ w
    v = var - np.sum(A * A, axis=0)
    if y_instead_of_f:
        v = v + noise
    return mean, np.sqrt(v)


def check_k_inv_y(X, y, ell, var, noise, Xs) -> float:
    """MOGP.check_K_inv_Y (gpr/models.py:446-463): RMS of k(x,X) . K_inv_Y - predict(x) ; should be ~0."""
    alpha = k_inv_y(X, y, ell, var, noise)
    mean, _ = predict(X, y, ell, var, noise, Xs)
    r = gram(Xs, ell, var, X) @ alpha - mean
    return float(np.sqrt(np.sum(r * r) / len(r)))


# --------------------------------------------------------------------------------------------------------------------
# Closed-form Sobol: literal transliteration of the reference's TF broadcasting (gsa/base.py:52-126, gsa/calibrators.py:49-143)
# --------------------------------------------------------------------------------------------------------------------

class Gaussian:
    """NumPy transliteration of gsa.base.Gaussian (gsa/base.py:52-126), diagonal-variance branch only.
    exponent = -z'z/2, cho_diag = broadcast sqrt(variance); the 2 pi factors are omitted as in the reference."""

    def __init__(self, mean, variance, ordinate=None, LBunch: int = 2):
        variance_cho = np.sqrt(variance)                                    # :107
        ordinate = np.zeros(()) if ordinate is None else ordinate
        if ordinate.shape == mean.shape:                                    # :108-112 (not hit on this path)
            shape = list(ordinate.shape)
            fill = [1, ] * (len(shape) - 1)
            ordinate = np.reshape(ordinate, shape[:-1] + fill + [shape[-1]])
            mean = np.reshape(mean, fill + shape)
        ordinate = ordinate - mean                                          # :113
        insertions = variance_cho.ndim - 1                                  # :115
        insertions -= insertions % LBunch                                   # :116
        for axis in range(insertions, 0, -LBunch):                          # :117-118
            variance_cho = np.expand_dims(variance_cho, axis)
        target = tuple(variance_cho.shape[:-2]) + tuple(ordinate.shape[-2:])
        exponent = ordinate / np.broadcast_to(variance_cho, target)          # :121
        self.exponent = -0.5 * np.einsum('...o,...o->...', exponent, exponent)  # :124
        self.cho_diag = variance_cho                                        # :126

    @property
    def det(self):                                                          # :58-61
        return np.prod(self.cho_diag, axis=-1)

    @property
    def pdf(self):                                                          # :63-66
        return np.exp(self.exponent) / self.det

    def expand_dims(self, axes: Sequence[int]) -> 'Gaussian':               # :68-79
        result = Gaussian.__new__(Gaussian)
        result.exponent, result.cho_diag = self.exponent, self.cho_diag
        for axis in sorted(axes, reverse=True):
            result.exponent = np.expand_dims(result.exponent, axis)
            result.cho_diag = np.expand_dims(result.cho_diag, (axis - 1) if axis < 0 else axis)
        return result

    def __truediv__(self, other: 'Gaussian') -> 'Gaussian':                 # :81-90
        result = Gaussian.__new__(Gaussian)
        result.exponent = self.exponent - other.exponent
        result.cho_diag = self.cho_diag / other.cho_diag
        return result


class LiteralClosedSobol:
    """Op-by-op NumPy transliteration of gsa.calibrators.ClosedSobol for independent GPs (is_F_diagonal=True).

    Memory is O(L^2 N^2 M): use only for small N (tests).  Arguments mirror what the reference reads from the gp
    (gsa/calibrators.py:119-140): X (N,M), K_inv_Y (L,1,N), kernel variance F (1,L) or (L,), lengthscales (L,M).
    """

    def __init__(self, X: np.ndarray, K_inv_Y: np.ndarray, F: np.ndarray, lengthscales: np.ndarray):
        self.X = np.asarray(X, dtype=np.float64)
        self.N, self.M = self.X.shape
        self.K_inv_Y = np.asarray(K_inv_Y, dtype=np.float64)
        self.L = self.K_inv_Y.shape[0]
        self.F = np.reshape(np.asarray(F, dtype=np.float64), (self.L, 1))                    # :134-136
        self.Lambda = np.broadcast_to(np.asarray(lengthscales, dtype=np.float64), (self.L, self.M))   # :140
        lam2 = np.einsum('lM,lM->lM', self.Lambda, self.Lambda)[:, None, :]                   # :105
        plus = tuple(lam2 + j for j in range(3))                                               # :108
        self.Lambda2 = {1: plus, -1: tuple(v ** (-1) for v in plus)}                           # :109
        self._calibrate()

    def _calibrate(self):                                                                      # :82-97
        pre_factor = np.sqrt(np.prod(self.Lambda2[1][0] * self.Lambda2[-1][1], axis=-1)) * self.F
        self.g0 = np.exp(Gaussian(mean=self.X[None, None, ...], variance=self.Lambda2[1][1]).exponent)
        self.g0 = self.g0 * pre_factor[..., None]
        self.g0KY = self.g0 * self.K_inv_Y
        self.g0KY = self.g0KY - np.einsum('lLN->l', self.g0KY)[..., None, None] / float(np.prod(self.g0KY.shape[1:]))
        self.G = np.einsum('lLM,NM->lLNM', self.Lambda2[-1][1], self.X)
        self.Phi = self.Lambda2[-1][1]
        self.V = {0: self._V(self.G, self.Phi)}
        self.V[1] = np.diagonal(self.V[0]).copy()
        V = np.sqrt(self.V[1])
        self.V[2] = np.einsum('l,i->li', V, V)
        self.S = self.V[0] / self.V[2]

    def _V(self, G, Phi):                                                                      # :60-80
        Gamma = 1 - Phi
        Psi = Gamma[:, :, None, None, :] + Gamma[None, None, ...]
        Psi = Psi - np.einsum('lLM,jJM->lLjJM', Gamma, Gamma)
        PsiPhi = np.einsum('lLjJM,lLM->lLjJM', Psi, Phi)
        PhiG = np.expand_dims(np.einsum('lLM,jJnM->lLjJnM', Phi, G), axis=2)
        PhiGauss = Gaussian(mean=G, variance=Phi)
        H = Gaussian(mean=PhiG, variance=PsiPhi, ordinate=G[..., None, None, None, :])
        H = H / PhiGauss.expand_dims([-1, -2, -3])
        return np.einsum('lLN,lLNjJn,jJn->lj', self.g0KY, H.pdf, self.g0KY)

    def marginalize(self, m: Sequence[int]) -> Dict[str, np.ndarray]:                          # :49-58
        G, Phi = self.G[..., m[0]:m[1]], self.Phi[..., m[0]:m[1]]
        result = {'V': self._V(G, Phi)}
        result['S'] = result['V'] / self.V[2]
        return result


# --------------------------------------------------------------------------------------------------------------------
# Closed-form Sobol: O(N M) memory restatement, tiled over (n, n').
# --------------------------------------------------------------------------------------------------------------------

def sobol_prepare(X: np.ndarray, alpha: np.ndarray, F: np.ndarray, lengthscales: np.ndarray):
    """g (L,N) = centred g0*alpha and phi (L,M) = 1/(ell^2+1)  (gsa/calibrators.py:86-92)."""
    X = np.asarray(X, dtype=np.float64)
    alpha = np.atleast_2d(np.asarray(alpha, dtype=np.float64))
    L = alpha.shape[0]
    ell = np.broadcast_to(np.asarray(lengthscales, dtype=np.float64), (L, X.shape[1]))
    F = np.reshape(np.asarray(F, dtype=np.float64), (L,))
    phi = 1.0 / (ell * ell + 1.0)
    pre = F * np.sqrt(np.prod(ell * ell * phi, axis=1))
    g0 = pre[:, None] * np.exp(-0.5 * np.einsum('lm,nm->ln', phi, X * X))
    g = g0 * alpha
    g = g - g.mean(axis=1, keepdims=True)
    return g, phi


def sobol_V_pair(X, g_l, g_j, phi_l, phi_j, slices: Iterable[Sequence[int]], block: int = 1024, rows: Tuple[int, int] | None = None) -> np.ndarray:
    """V_lj for each dim-slice [a,b):  sum_{n,n'} g_l[n] g_j[n'] prod_{m in slice} h_m(n,n') with
    log h_m = -1/2 log(1-a_m) - 1/2 a_m [phi_l x_n^2 + phi_j x_n'^2 - 2 x_n x_n'] / (1-a_m),  a_m = phi_l,m phi_j,m
    (algebraic reduction of gsa/calibrators.py:69-79; for l=j this is SURVEY.md Appendix A).
    ``rows = (r0, r1)`` restricts the n-sum to that stripe (the timing sample of bench.py: the work per row is uniform)."""
    slices = [tuple(int(v) for v in s) for s in slices]
    N, M = X.shape
    r_lo, r_hi = (0, N) if rows is None else rows
    a = phi_l * phi_j
    c0 = -0.5 * np.log1p(-a)
    c2 = a / (1.0 - a)
    rl = -0.5 * c2 * phi_l * X * X            # (N,M) depends on row index n
    rj = -0.5 * c2 * phi_j * X * X            # (N,M) depends on column index n'
    out = np.zeros(len(slices))
    for i0 in range(r_lo, r_hi, block):
        i1 = min(r_hi, i0 + block)
        for j0 in range(0, N, block):
            j1 = min(N, j0 + block)
            t = (c0[None, None, :] + rl[i0:i1, None, :] + rj[None, j0:j1, :]
                 + c2[None, None, :] * X[i0:i1, None, :] * X[None, j0:j1, :])     # (bi,bj,M)
            wgt = g_l[i0:i1, None] * g_j[None, j0:j1]
            for s, (lo, hi) in enumerate(slices):
                out[s] += np.sum(wgt * np.exp(np.sum(t[:, :, lo:hi], axis=-1)))
    return out


class ClosedSobolOracle:
    """Same public surface as the reference ClosedSobol (V dict, S, marginalize) with O(N M) memory."""

    def __init__(self, X, K_inv_Y, F, lengthscales):
        self.X = np.asarray(X, dtype=np.float64)
        self.N, self.M = self.X.shape
        alpha = np.asarray(K_inv_Y, dtype=np.float64).reshape(-1, self.N)
        self.L = alpha.shape[0]
        self.g, self.phi = sobol_prepare(self.X, alpha, F, lengthscales)
        self.V = {0: self._V((0, self.M))}
        self.V[1] = np.diagonal(self.V[0]).copy()
        V = np.sqrt(self.V[1])
        self.V[2] = np.outer(V, V)
        self.S = self.V[0] / self.V[2]

    def _V_many(self, slices) -> np.ndarray:
        out = np.empty((self.L, self.L, len(slices)))
        for l in range(self.L):
            for j in range(self.L):
                out[l, j] = sobol_V_pair(self.X, self.g[l], self.g[j], self.phi[l], self.phi[j], slices)
        return out

    def _V(self, m) -> np.ndarray:
        return self._V_many([m])[..., 0]

    def marginalize(self, m) -> Dict[str, np.ndarray]:
        V = self._V(m)
        return {'V': V, 'S': V / self.V[2]}


# --------------------------------------------------------------------------------------------------------------------
# GSA harness: slices per kind and post-processing (gsa/models.py:77-90, 117-137, 207-214)
# --------------------------------------------------------------------------------------------------------------------

FIRST_ORDER, CLOSED, TOTAL = 1, 2, 3      # GSA.Kind IntEnum auto() values (gsa/models.py:38-42)


def gsa_slices(kind: int, M: int, m: int = -1) -> List[Tuple[int, int]]:
    """GSA._m_dataset (gsa/models.py:77-90)."""
    ms = range(M) if m < 0 else [m]
    if kind == FIRST_ORDER:
        return [(i, i + 1) for i in ms]
    if kind == CLOSED:
        return [(0, i + 1) for i in ms]
    if kind == TOTAL:
        return [(i + 1, M) for i in ms]
    raise ValueError(kind)


def gsa_calibrate(calibrator, kind: int, M: int, m: int = -1) -> Dict[str, np.ndarray]:
    """GSA.calibrate + Sobol._post_calibrate (gsa/models.py:117-137, 207-214): stack per-slice results on a new last axis,
    append the full-model column; TOTAL index = S_full - S_closed(complement)."""
    results: Dict[str, np.ndarray] = {}
    for sl in gsa_slices(kind, M, m):
        r = calibrator.marginalize(sl)
        for key, value in r.items():
            results[key] = value[..., None] if key not in results else np.concatenate([results[key], value[..., None]], axis=-1)
    results['V'] = np.concatenate([results['V'], calibrator.V[0][..., None]], axis=-1)
    if kind == TOTAL:
        results['S'] = calibrator.S[..., None] - results['S']
    results['S'] = np.concatenate([results['S'], calibrator.S[..., None]], axis=-1)
    return results


def all_slices(M: int) -> List[Tuple[int, int]]:
    """The 3M slices of the three kinds followed by the full model [0,M): the 3M+1 quadratic forms per (fold, output)."""
    return gsa_slices(FIRST_ORDER, M) + gsa_slices(CLOSED, M) + gsa_slices(TOTAL, M) + [(0, M)]


def predict_gradient(X, y, ell, var, noise, xs) -> Tuple[np.ndarray, np.ndarray]:
    """MOGP.predict_gradient for one independent output (gpr/models.py:386-415): mean (o, M) = dK^T alpha and
    var (o, o, M, M) = -(L^-1 dK)^T (L^-1 dK) with k(x_O, x_o)/ell_M^2 added on the M == m diagonal (:411-414), where
    dK[N, o, M] = d k(X_N, x_o)/d x_oM (tape.jacobian at :397, analytic here). The reference ignores y_instead_of_f."""
    N, M = X.shape
    o_ = xs.shape[0]
    ell = np.broadcast_to(np.asarray(ell, dtype=np.float64), (M,))
    L = k_cho(X, ell, var, noise)
    alpha = scipy.linalg.cho_solve((L, True), y, check_finite=False)
    K = gram(X, ell, var, xs)                                                   # (N, o)
    dK = -(xs[None, :, :] - X[:, None, :]) / (ell * ell) * K[:, :, None]        # (N, o, M)
    mean = np.einsum('NoM,N->oM', dK, alpha)
    V = scipy.linalg.solve_triangular(L, dK.reshape(N, o_ * M), lower=True, check_finite=False).reshape(N, o_, M)
    cov = -np.einsum('NOM,Nom->OoMm', V, V)
    kxx = gram(xs, ell, var)
    idx = np.arange(M)
    cov[:, :, idx, idx] += kxx[:, :, None] / (ell * ell)[None, None, :]
    return mean, cov
