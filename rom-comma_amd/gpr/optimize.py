"""Hyper-parameter fitting for one independent-output GP resident on the GPU.

Replaces ``gf.optimizers.Scipy().minimize(closure=gp.training_loss, variables=gp.trainable_variables, method='L-BFGS-B',
options=meta)`` at reference gpr/models.py:359-361. GPflow hands SciPy the objective -LML(u) and its gradient on the packed
*unconstrained* vector u of the trainable parameters, with theta = softplus(u) for kernel variance and lengthscales and
theta = 1e-6 + softplus(u) for the Gaussian likelihood variance. The same SciPy routine drives the HIP backend here; every
evaluation is one ``rcgp_lml_grad`` call (Gram + Cholesky + L^-1 + fused gradient reduction on the GPU) and the softplus
chain rule stays on the host.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np
import scipy.optimize
import scipy.special

LIKELIHOOD_LOWER = 1.0e-6          # GPflow Gaussian likelihood: variance = 1e-6 + softplus(u)
LIKELIHOOD_VARIANCE_FLOOR = 1.0001e-6     # reference gpr/models.py:62-65, :341
KERNEL_VARIANCE_FLOOR = 1.0005e-6         # reference gpr/kernels.py:176


def softplus(u):
    return np.logaddexp(0.0, np.asarray(u, dtype=np.float64))


def inv_softplus(x):
    x = np.asarray(x, dtype=np.float64)
    return x + np.log(-np.expm1(-x))


def sigmoid(u):
    return scipy.special.expit(np.asarray(u, dtype=np.float64))


def fit_lbfgsb(gp, lengthscales, variance: float, noise: float, is_isotropic: bool = False, train_lengthscales: bool = True,
               train_variance: bool = True, train_noise: bool = True, method: str = 'L-BFGS-B', callback=None,
               **options: Any) -> Dict[str, Any]:
    """Minimise -LML over the trainable hyper-parameters of ``gp`` (a ``romcomma_amd._lib.RcGP``).

    Args:
        lengthscales: start lengthscales, (M,) or a single isotropic value (reference default 5.0, gpr/kernels.py:50).
        variance: start kernel variance (reference default 2.0, gpr/kernels.py:49); floored at 1.0005e-6.
        noise: start likelihood variance (reference default 0.02, gpr/models.py:52); floored at 1.0001e-6.
        is_isotropic: one shared lengthscale (the '.i' models).
        options: SciPy options, reference defaults maxiter=5000, gtol=1e-16 (gpr/models.py:327-330).
    Returns: dict(lengthscales (M,), variance, noise, log_marginal, result (scipy OptimizeResult), nfev).
    """
    M = gp.M
    ell0 = np.broadcast_to(np.asarray(lengthscales, dtype=np.float64).reshape(-1), (1,) if is_isotropic else (M,)).copy()
    variance = max(float(variance), KERNEL_VARIANCE_FLOOR)
    noise = max(float(noise), LIKELIHOOD_VARIANCE_FLOOR)
    n_ell = ell0.shape[0]
    u_all = np.concatenate([inv_softplus(ell0), [inv_softplus(variance)], [inv_softplus(noise - LIKELIHOOD_LOWER)]])
    mask = np.array([train_lengthscales] * n_ell + [train_variance, train_noise])
    state = {'nfev': 0}

    def unpack(u_train):
        u = u_all.copy()
        u[mask] = u_train
        return u, softplus(u[:n_ell]), float(softplus(u[n_ell])), float(LIKELIHOOD_LOWER + softplus(u[n_ell + 1]))

    def objective(u_train):
        u, ell, var, nse = unpack(u_train)
        gp.set_hyper(np.broadcast_to(ell, (M,)), var, nse)
        lml, grad = gp.lml_grad()
        state['nfev'] += 1
        g_ell = np.array([np.sum(grad[:M])]) if is_isotropic else grad[:M]
        g = np.concatenate([g_ell, grad[M:]]) * sigmoid(u)
        return -lml, -g[mask]

    opts = {'maxiter': 5000, 'gtol': 1e-16} | options
    if np.any(mask):
        result = scipy.optimize.minimize(objective, u_all[mask], jac=True, method=method, options=opts, callback=callback)
        u_all[mask] = result.x
    else:
        result = None
    _, ell, var, nse = unpack(u_all[mask])
    gp.set_hyper(np.broadcast_to(ell, (M,)), var, nse)
    log_marginal = gp.lml()
    return {'lengthscales': np.broadcast_to(ell, (M,)).copy(), 'variance': var, 'noise': nse, 'log_marginal': log_marginal,
            'result': result, 'nfev': state['nfev']}
