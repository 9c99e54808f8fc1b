"""CPU oracle for the COVARIANT (dependent multi-output) GP path: the reference's ``romcomma.gpf`` extension of GPflow
(``gpf/kernels.py``, ``gpf/base.py``, ``gpf/likelihoods.py``, ``gpf/models.py``) and the covariant branches of
``gpr/models.py`` and ``gsa/calibrators.py``.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.

PARITY UNPINNED, as for ``gp_oracle.py``: the reference's path needs TensorFlow + GPflow (absent here) and ships no fixtures.
The restatement is pinned internally instead (``tests/test_oracle_mogp.py``): the literal broadcast form of the kernel against
the stacked-points form, the analytic gradient against central finite differences, the literal Sobol transliteration against
the pair closed form, and the L = 1 limit against ``gp_oracle``.

Paths in the citations are relative to ``/root/reference/romcomma/``.  Index convention everywhere: the (LN) axis is
output-major, ``a = l * N + n`` (``gpf/models.py:120``: ``reshape(transpose(Y), [-1, 1])``).
"""
from __future__ import annotations

import math
from typing import Dict, Sequence, Tuple

import numpy as np
import scipy.linalg
import scipy.optimize

from oracle import gp_oracle

LOG_2PI = math.log(2.0 * math.pi)
CHOLESKY_DIAGONAL_LOWER_BOUND = 1.0e-3                      # gpf/base.py:35


# --------------------------------------------------------------------------------------------------------------------
# Kernel (gpf/kernels.py) and likelihood (gpf/likelihoods.py)
# --------------------------------------------------------------------------------------------------------------------

def stack_y(Y: np.ndarray) -> np.ndarray:
    """(N, L) -> (LN,), output-major (gpf/models.py:120)."""
    return np.ascontiguousarray(np.asarray(Y, dtype=np.float64).T).reshape(-1)


def unit_gram_literal(X: np.ndarray, ell: np.ndarray, X2: np.ndarray | None = None) -> np.ndarray:
    """K_unit_variance (gpf/kernels.py:73-82,153-154) written the way GPflow's AnisotropicStationary does it:
    scale(X) = X / lengthscales with lengthscales (L,1,M) -> (L,N,M); difference_matrix -> d (L,N,L,N2,M);
    exp(-1/2 einsum('...M,...M->...', d, d)) -> (L,N,L,N2). O(L^2 N^2 M) memory: small cases only."""
    X = np.asarray(X, dtype=np.float64)
    X2 = X if X2 is None else np.asarray(X2, dtype=np.float64)
    ell = np.asarray(ell, dtype=np.float64)
    L = ell.shape[0]
    sX = X[None, :, :] / ell[:, None, :]                       # (L, N, M)
    sX2 = X2[None, :, :] / ell[:, None, :]                     # (L, N2, M)
    d = sX[:, :, None, None, :] - sX2[None, None, :, :, :]     # (L, N, L, N2, M)
    return np.exp(-0.5 * np.einsum('...M,...M->...', d, d))


def unit_gram(X: np.ndarray, ell: np.ndarray, X2: np.ndarray | None = None) -> np.ndarray:
    """The same numbers from the stacked points u_(l,n) = x_n / ell_l with the |u|^2 + |u'|^2 - 2 u.u' expansion
    (the product's formulation), returned as (LN, LN2)."""
    X = np.asarray(X, dtype=np.float64)
    ell = np.asarray(ell, dtype=np.float64)
    U = (X[None, :, :] / ell[:, None, :]).reshape(-1, X.shape[1])
    U2 = U if X2 is None else (np.asarray(X2, dtype=np.float64)[None, :, :] / ell[:, None, :]).reshape(-1, X.shape[1])
    return np.exp(-0.5 * gp_oracle.square_distance(U, None if X2 is None else U2))


def gram(X, ell, F, X2=None) -> np.ndarray:
    """K_d_apply_variance (gpf/kernels.py:93-104): variance (L,1,L,1) * unit, reshaped (LN, LN2)."""
    X = np.asarray(X, dtype=np.float64)
    F = np.asarray(F, dtype=np.float64)
    L, N = F.shape[0], X.shape[0]
    N2 = N if X2 is None else np.asarray(X2).shape[0]
    E = unit_gram(X, ell, X2).reshape(L, N, L, N2)
    return (F[:, None, :, None] * E).reshape(L * N, L * N2)


def noisy_gram(X, ell, F, Sigma) -> np.ndarray:
    """likelihood.add_to(KXX) (gpf/likelihoods.py:61-64, gpf/base.py:61-69): K + Sigma (x) I_N."""
    N = np.asarray(X).shape[0]
    return gram(X, ell, F) + np.kron(np.asarray(Sigma, dtype=np.float64), np.eye(N))


def initial_noise(noise_variance, L: int) -> np.ndarray:
    """MOGPR.__init__ (gpf/models.py:121-123): the test ``tf.shape(...).numpy != (L, L)`` compares a bound method with a tuple
    and is always true, so whatever is passed is broadcast to (L, L) and then reduced to its diagonal."""
    return np.diag(np.diag(np.broadcast_to(np.asarray(noise_variance, dtype=np.float64), (L, L))))


def k_cho(X, ell, F, Sigma) -> np.ndarray:
    """MOGP.K_cho, covariant branch (gpr/models.py:429-431,439): (LN, LN) lower."""
    return scipy.linalg.cholesky(noisy_gram(X, ell, F, Sigma), lower=True, check_finite=False)


def k_inv_y(X, Y, ell, F, Sigma) -> np.ndarray:
    """MOGP.K_inv_Y, covariant branch (gpr/models.py:441-444): (L, 1, N)."""
    Lc = k_cho(X, ell, F, Sigma)
    a = scipy.linalg.cho_solve((Lc, True), stack_y(Y), check_finite=False)
    return a.reshape(np.asarray(F).shape[0], 1, -1)


def lml(X, Y, ell, F, Sigma) -> float:
    """MOGPR.log_marginal_likelihood (gpf/models.py:73-82): multivariate_normal(Y_stacked, 0, chol(K + Sigma (x) I))."""
    Lc = k_cho(X, ell, F, Sigma)
    y = stack_y(Y)
    w = scipy.linalg.solve_triangular(Lc, y, lower=True, check_finite=False)
    return float(-0.5 * w @ w - 0.5 * len(y) * LOG_2PI - np.sum(np.log(np.diag(Lc))))


def lml_and_grad(X, Y, ell, F, Sigma) -> Tuple[float, np.ndarray, np.ndarray, np.ndarray]:
    """LML and its partial derivatives with every entry of F, ell and Sigma treated as an independent variable:
        W = alpha alpha^T - Kn^-1
        dF[l, j]     = 1/2 sum_{n,n'} W[(l,n),(j,n')] E[(l,n),(j,n')]
        dSigma[l, j] = 1/2 sum_n W[(l,n),(j,n)]
        dell[l, m]   = sum_{a in l} sum_b W_ab K_ab (u_am - u_bm) u_am / ell_lm ,  u_(l,n) = x_n / ell_l
    (the reference gets them from TF autodiff through gf.optimizers.Scipy, gpr/models.py:359-361)."""
    X = np.asarray(X, dtype=np.float64)
    ell = np.asarray(ell, dtype=np.float64)
    F = np.asarray(F, dtype=np.float64)
    N, M = X.shape
    L = F.shape[0]
    E = unit_gram(X, ell)
    K = (F[:, None, :, None] * E.reshape(L, N, L, N)).reshape(L * N, L * N)
    Kn = K + np.kron(np.asarray(Sigma, dtype=np.float64), np.eye(N))
    Lc = scipy.linalg.cholesky(Kn, lower=True, check_finite=False)
    y = stack_y(Y)
    w = scipy.linalg.solve_triangular(Lc, y, lower=True, check_finite=False)
    value = float(-0.5 * w @ w - 0.5 * len(y) * LOG_2PI - np.sum(np.log(np.diag(Lc))))
    alpha = scipy.linalg.solve_triangular(Lc.T, w, lower=False, check_finite=False)
    Kinv = scipy.linalg.cho_solve((Lc, True), np.eye(L * N), check_finite=False)
    W = np.outer(alpha, alpha) - Kinv
    dF = 0.5 * np.einsum('lNjn,lNjn->lj', W.reshape(L, N, L, N), E.reshape(L, N, L, N))
    dSigma = 0.5 * np.einsum('lnjn->lj', W.reshape(L, N, L, N))
    U = (X[None, :, :] / ell[:, None, :]).reshape(L * N, M)
    WK = W * K
    r = WK.sum(axis=1)
    Q = WK @ U
    per_point = U * (U * r[:, None] - Q)                       # sum_b WK_ab (u_am - u_bm) u_am
    dell = per_point.reshape(L, N, M).sum(axis=1) / ell
    return value, dF, dell, dSigma


def predict(X, Y, ell, F, Sigma, Xs, y_instead_of_f: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """MOGP.predict, covariant branch (gpr/models.py:377-379,384) -> (mean (o, L), SD (o, L)).
    MOGPR.predict_f (gpf/models.py:84-113) with full_cov = full_output_cov = False keeps the diagonal of
    Knn - A^T A, A = L^-1 Kmn; predict_y adds diag(Sigma) (gpf/likelihoods.py:77-86, rank-2 branch)."""
    F = np.asarray(F, dtype=np.float64)
    L = F.shape[0]
    Xs = np.asarray(Xs, dtype=np.float64)
    o = Xs.shape[0]
    Lc = k_cho(X, ell, F, Sigma)
    Kmn = gram(X, ell, F, Xs)                                   # (LN, Lo)
    A = scipy.linalg.solve_triangular(Lc, Kmn, lower=True, check_finite=False)
    w = scipy.linalg.solve_triangular(Lc, stack_y(Y), lower=True, check_finite=False)
    mean = (A.T @ w).reshape(L, o).T
    var = (np.repeat(np.diag(F), o) - np.sum(A * A, axis=0)).reshape(L, o).T
    if y_instead_of_f:
        var = var + np.diag(np.asarray(Sigma, dtype=np.float64))[None, :]
    return mean, np.sqrt(var)


def check_k_inv_y(X, Y, ell, F, Sigma, Xs) -> np.ndarray:
    """MOGP.check_K_inv_Y, covariant branch (gpr/models.py:446-463): (L,) RMS of k(x, X) . K_inv_Y - predict(x)."""
    L = np.asarray(F).shape[0]
    o = np.asarray(Xs).shape[0]
    N = np.asarray(X).shape[0]
    KiY = k_inv_y(X, Y, ell, F, Sigma)
    kernel = gram(Xs, ell, F, X).reshape(L, o, L, N)
    r = np.einsum('loLN,LiN->ol', kernel, KiY) - predict(X, Y, ell, F, Sigma, Xs)[0]
    return np.sqrt(np.sum(r * r, axis=0) / o)


def predict_gradient(X, Y, ell, F, Sigma, xs) -> Tuple[np.ndarray, np.ndarray]:
    """MOGP.predict_gradient, covariant branch (gpr/models.py:386-415), with the analytic Jacobian of K(X, x) in place of
    tape.jacobian: mean (o, L, M) and var (o, L, o, L, M, M). As written in the reference, 'LNlOM, LNlom -> OLolMm' (:398) sums the
    Cholesky-solved rows over N only, so the first L of var is the TRAINING output block; that is restated, not corrected."""
    X = np.asarray(X, dtype=np.float64)
    xs = np.asarray(xs, dtype=np.float64)
    ell = np.asarray(ell, dtype=np.float64)
    F = np.asarray(F, dtype=np.float64)
    L, (N, M), o = F.shape[0], X.shape, xs.shape[0]
    Lc = k_cho(X, ell, F, Sigma)
    KiY = k_inv_y(X, Y, ell, F, Sigma)
    U = X[None, :, :] / ell[:, None, :]
    u = xs[None, :, :] / ell[:, None, :]
    d = U[:, :, None, None, :] - u[None, None, :, :, :]                           # (L, N, l, o, M)
    K = F[:, None, :, None] * np.exp(-0.5 * np.einsum('...M,...M->...', d, d))
    dK = K[..., None] * d / ell[None, None, :, None, :]                           # d K[(L,N),(l,o)] / d x_oM   (:393-395)
    mean = np.einsum('LNloM,LiN->olM', dK, KiY)                                   # :396
    V = scipy.linalg.solve_triangular(Lc, dK.reshape(L * N, L * o * M), lower=True, check_finite=False).reshape(L, N, L, o, M)   # :397
    var = -np.einsum('LNlOM,LNlom->OLolMm', V, V)                                 # :398
    Lambda = np.broadcast_to(1.0 / ell, (o, L, M))                                # :388
    dd = u[:, :, None, None, :] - u[None, None, :, :, :]
    kxx = F[:, None, :, None] * np.exp(-0.5 * np.einsum('...M,...M->...', dd, dd))    # kernel(x) as (L, o, L, o)
    ddxxkxx = np.einsum('OLM,olM,LOlo->OLolM', Lambda, Lambda, kxx)               # :399-400
    idx = np.arange(M)
    var[..., idx, idx] += ddxxkxx                                                 # :406
    return mean, var


# --------------------------------------------------------------------------------------------------------------------
# Variance parametrisation (gpf/base.py:32-96) and the fit (gpr/models.py:345-373, gpr/kernels.py:59-70, gpr/models.py:71-80)
# --------------------------------------------------------------------------------------------------------------------

def variance_to_params(V: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Variance.__init__ (gpf/base.py:71-96): (unconstrained Cholesky diagonal, strictly lower triangle row by row)."""
    C = scipy.linalg.cholesky(np.asarray(V, dtype=np.float64), lower=True)
    d = np.diag(C)
    if d.min() <= CHOLESKY_DIAGONAL_LOWER_BOUND:
        raise ValueError('The Cholesky diagonal of a Variance must be strictly greater than 1e-3.')     # gpf/base.py:87-88
    L = C.shape[0]
    lower = np.array([C[i, j] for i in range(1, L) for j in range(i)], dtype=np.float64)                # mask, gpf/base.py:92
    return gp_oracle.inv_softplus(d - CHOLESKY_DIAGONAL_LOWER_BOUND), lower


def params_to_cholesky(u_diag: np.ndarray, lower: np.ndarray) -> np.ndarray:
    """Variance.cholesky (gpf/base.py:42-50): ragged rows of the lower triangle, then set_diag(positive(lower=1e-3)(u))."""
    L = len(u_diag)
    C = np.zeros((L, L))
    k = 0
    for i in range(1, L):
        C[i, :i] = lower[k:k + i]
        k += i
    C[np.diag_indices(L)] = CHOLESKY_DIAGONAL_LOWER_BOUND + gp_oracle.softplus(np.asarray(u_diag, dtype=np.float64))
    return C


def cholesky_chain(dV: np.ndarray, C: np.ndarray, u_diag: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Chain rule V = C C^T for a gradient dV taken with independent entries: dC = (dV + dV^T) C; returns the parts
    belonging to (u_diag, strictly lower triangle)."""
    dC = (dV + dV.T) @ C
    L = C.shape[0]
    lower = np.array([dC[i, j] for i in range(1, L) for j in range(i)], dtype=np.float64)
    return np.diag(dC) * gp_oracle.sigmoid(np.asarray(u_diag, dtype=np.float64)), lower


#: What trains by default in a covariant fit: Kernel.META (gpr/kernels.py:56-57) through Kernel.calibrate (:62-65), and
#: Likelihood.META (gpr/models.py:60) through Likelihood.calibrate (:74-76).
DEFAULT_TRAINABLE = {'kernel_variance': True, 'kernel_covariance': False, 'lengthscales': False,
                     'likelihood_variance': True, 'likelihood_covariance': True}


def fit(X, Y, ell0, F0, Sigma0, trainable: Dict[str, bool] | None = None, maxiter: int = 5000, gtol: float = 1e-16):
    """MOGP.calibrate, covariant branch: L-BFGS-B on -LML over the trainable subset (gpr/models.py:345-367).
    Returns (ell, F, Sigma, lml, scipy result, number of evaluations)."""
    tr = dict(DEFAULT_TRAINABLE)
    tr.update(trainable or {})
    X = np.asarray(X, dtype=np.float64)
    ell0 = np.asarray(ell0, dtype=np.float64)
    L, M = ell0.shape
    kd, kl = variance_to_params(F0)
    nd, nl = variance_to_params(Sigma0)
    ue = gp_oracle.inv_softplus(ell0).reshape(-1)
    state = {'kd': kd, 'kl': kl, 'nd': nd, 'nl': nl, 'ue': ue}
    names = [n for n, on in (('kd', tr['kernel_variance']), ('kl', tr['kernel_covariance']), ('ue', tr['lengthscales']),
                             ('nd', tr['likelihood_variance']), ('nl', tr['likelihood_covariance'])) if on and len(state[n])]
    evals = [0]

    def unpack(u):
        s = dict(state)
        k = 0
        for n in names:
            s[n] = u[k:k + len(state[n])]
            k += len(state[n])
        return s

    def build(s):
        Ck, Cn = params_to_cholesky(s['kd'], s['kl']), params_to_cholesky(s['nd'], s['nl'])
        return gp_oracle.softplus(s['ue']).reshape(L, M), Ck, Cn

    def fun(u):
        s = unpack(u)
        ell, Ck, Cn = build(s)
        evals[0] += 1
        try:
            v, dF, dell, dS = lml_and_grad(X, Y, ell, Ck @ Ck.T, Cn @ Cn.T)
        except np.linalg.LinAlgError:
            return 1e300, np.zeros_like(u)
        g = {}
        g['kd'], g['kl'] = cholesky_chain(dF, Ck, s['kd'])
        g['nd'], g['nl'] = cholesky_chain(dS, Cn, s['nd'])
        g['ue'] = (dell * gp_oracle.sigmoid(s['ue'].reshape(L, M))).reshape(-1)
        return -v, -np.concatenate([g[n] for n in names])

    u0 = np.concatenate([state[n] for n in names])
    res = scipy.optimize.minimize(fun, u0, jac=True, method='L-BFGS-B', options={'maxiter': maxiter, 'gtol': gtol})
    ell, Ck, Cn = build(unpack(res.x))
    F, Sigma = Ck @ Ck.T, Cn @ Cn.T
    return ell, F, Sigma, lml(X, Y, ell, F, Sigma), res, evals[0]


# --------------------------------------------------------------------------------------------------------------------
# Closed-form Sobol with a non-diagonal F (gsa/calibrators.py:99-143 with is_F_diagonal False)
# --------------------------------------------------------------------------------------------------------------------

class LiteralClosedSobolCovariant(gp_oracle.LiteralClosedSobol):
    """The transliteration of gp_oracle.LiteralClosedSobol with the non-diagonal branches of the constructor:
    F stays (L, L) (:133-136 not taken), K_inv_Y is transposed to (1, L, N) (:137-138) and Lambda^2 is the PRODUCT of the
    lengthscales of two outputs, einsum('lM,LM->lLM') (:106-107)."""

    def __init__(self, X: np.ndarray, K_inv_Y: np.ndarray, F: np.ndarray, lengthscales: np.ndarray):
        self.X = np.asarray(X, dtype=np.float64)
        self.N, self.M = self.X.shape
        KiY = np.asarray(K_inv_Y, dtype=np.float64)
        self.L = KiY.shape[0]
        self.F = np.asarray(F, dtype=np.float64).reshape(self.L, self.L)
        self.K_inv_Y = np.transpose(KiY, [1, 0, 2])                                            # :138
        self.Lambda = np.broadcast_to(np.asarray(lengthscales, dtype=np.float64), (self.L, self.M))
        lam2 = np.einsum('lM,LM->lLM', self.Lambda, self.Lambda)                               # :107
        plus = tuple(lam2 + j for j in range(3))
        self.Lambda2 = {1: plus, -1: tuple(v ** (-1) for v in plus)}
        self._calibrate()


def sobol_prepare_covariant(X, K_inv_Y, F, lengthscales):
    """g (L, L, N) and phi (L, L, M) of the pair form: for the ordered output pair p = (l, J),
    phi_p = 1/(ell_l ell_J + 1), g0_p = F_lJ sqrt(prod ell_l ell_J phi_p) exp(-1/2 sum phi_p x^2), g_p = g0_p alpha_J minus the
    mean over (J, N) for each l (gsa/calibrators.py:86-92 with the shapes of the non-diagonal branch)."""
    X = np.asarray(X, dtype=np.float64)
    alpha = np.asarray(K_inv_Y, dtype=np.float64).reshape(-1, X.shape[0])          # (L, N)
    L = alpha.shape[0]
    ell = np.broadcast_to(np.asarray(lengthscales, dtype=np.float64), (L, X.shape[1]))
    F = np.asarray(F, dtype=np.float64).reshape(L, L)
    lam2 = ell[:, None, :] * ell[None, :, :]
    phi = 1.0 / (lam2 + 1.0)
    pre = F * np.sqrt(np.prod(lam2 * phi, axis=-1))
    g0 = pre[..., None] * np.exp(-0.5 * np.einsum('lJm,nm->lJn', phi, X * X))
    g = g0 * alpha[None, :, :]
    g = g - g.sum(axis=(1, 2), keepdims=True) / float(L * X.shape[0])
    return g, phi


def sobol_V_covariant(X, K_inv_Y, F, lengthscales, slices: Sequence[Sequence[int]]) -> np.ndarray:
    """V[s] (L, L) for every slice: V_lj = sum_{L', J'} pair form between the virtual outputs (l, L') and (j, J')
    (the einsum 'lLN, lLNjJn, jJn -> lj' of gsa/calibrators.py:79), each pair evaluated by gp_oracle.sobol_V_pair."""
    g, phi = sobol_prepare_covariant(X, K_inv_Y, F, lengthscales)
    L = g.shape[0]
    out = np.zeros((len(slices), L, L))
    for l in range(L):
        for j in range(L):
            for a in range(L):
                for b in range(L):
                    out[:, l, j] += gp_oracle.sobol_V_pair(X, g[l, a], g[j, b], phi[l, a], phi[j, b], slices)
    return out
