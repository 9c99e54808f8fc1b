"""Hyper-parameter fitting for one independent-output GP resident on the GPU.

Replaces ``gf.optimizers.Scipy().minimize(closure=gp.training_loss, variables=gp.trainable_variables, method='L-BFGS-B',
options=meta)`` at reference gpr/models.py:359-361. GPflow hands SciPy the objective -LML(u) and its gradient on the packed
*unconstrained* vector u of the trainable parameters, with theta = softplus(u) for kernel variance and lengthscales and
theta = 1e-6 + softplus(u) for the Gaussian likelihood variance. The same SciPy routine drives the HIP backend here; every
evaluation is one ``rcgp_lml_grad`` call (Gram + Cholesky + L^-1 + fused gradient reduction on the GPU) and the softplus
chain rule stays on the host.
"""
from __future__ import annotations

import threading
from typing import Any, Dict, Optional, Sequence

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


# ---------------------------------------------------------------------------------------------------------------------
# Several units at once. The reference fits its outputs and folds one after the other (gpr/models.py:360-361, user/run.py:60-61); here the
# units' optimisers advance in LOCKSTEP: one thread per unit runs the same SciPy routine as above, and their evaluations meet in one
# ``rcgp_lml_grad_batch`` call -- one schedule on the GPU for all of them. Every unit sees exactly the numbers the single-handle call
# would give it (the library guarantees that bit by bit), so each fit follows the path it would follow alone.
# ---------------------------------------------------------------------------------------------------------------------

class _Lockstep:
    """The meeting point of the units' optimiser threads. A thread hands in its point and sleeps; the thread that completes the round
    (every live unit has handed one in) runs the batched evaluation for all of them and wakes the others. A unit whose fit has ended
    leaves, and the rounds go on among the rest."""

    def __init__(self, gps, batch_lml_grad, max_units: int):
        self._gps, self._batch, self._max = list(gps), batch_lml_grad, int(max_units)
        self._cond = threading.Condition()
        self._live = set(range(len(self._gps)))
        self._pending: Dict[int, tuple] = {}
        self._done: Dict[int, Any] = {}
        self.rounds = 0

    def _run_round(self):                                   # (the condition's lock is held)
        units = []
        for u in sorted(self._pending):                      # a point the library refuses (e.g. a lengthscale that underflowed to 0) fails ITS unit
            try:
                self._gps[u].set_hyper(*self._pending[u])
                units.append(u)
            except Exception as failure:
                self._done[u] = failure
        try:
            for i in range(0, len(units), self._max):        # more live units than one call takes: several calls
                part = units[i:i + self._max]
                lml, grad, status = self._batch([self._gps[u] for u in part])
                for k, u in enumerate(part):
                    self._done[u] = (float(lml[k]), np.array(grad[k]), int(status[k]))
        except Exception as failure:                         # a failed call fails every unit of the round
            for u in units:
                self._done.setdefault(u, failure)
        self._pending.clear()
        self.rounds += 1
        self._cond.notify_all()

    def evaluate(self, u: int, theta) -> tuple:
        with self._cond:
            self._pending[u] = theta
            if set(self._pending) == self._live:
                self._run_round()
            else:
                self._cond.wait_for(lambda: u in self._done)
            out = self._done.pop(u)
        if isinstance(out, Exception):
            raise out
        lml, grad, status = out
        if status > 0:
            from romcomma_amd._lib import NotPositiveDefiniteError
            raise NotPositiveDefiniteError(status, f'rcgp_lml_grad_batch: matrix is not positive definite: leading minor {status}')
        return lml, grad

    def leave(self, u: int):
        with self._cond:
            self._live.discard(u)
            self._pending.pop(u, None)
            if self._pending and set(self._pending) == self._live:
                self._run_round()

    def alone(self, call):
        """A single-handle library call made from a unit's thread: not while a round is in flight."""
        with self._cond:
            return call()


class _LockstepUnit:
    """What ``fit_lbfgsb`` sees of one unit: ``set_hyper`` only notes the point, ``lml_grad`` takes it to the meeting point."""

    def __init__(self, lockstep: _Lockstep, u: int, gp):
        self._lockstep, self._u, self._gp, self._theta = lockstep, u, gp, None
        self.M = gp.M

    def set_hyper(self, ell, variance, noise):
        self._theta = (np.array(ell, dtype=np.float64), float(variance), float(noise))

    def lml_grad(self):
        return self._lockstep.evaluate(self._u, self._theta)

    def lml(self):
        def call():
            self._gp.set_hyper(*self._theta)                  # normally the point of the last evaluation: nothing is recomputed
            return self._gp.lml()
        return self._lockstep.alone(call)


def fit_lbfgsb_batch(gps: Sequence[Any], starts: Sequence[Dict[str, Any]], batch_lml_grad=None, max_units: Optional[int] = None,
                     **common: Any) -> list:
    """``fit_lbfgsb`` for several units (``romcomma_amd._lib.RcGP`` of one device, equal M and padded size) at once.

    Args:
        starts: per unit the keyword arguments of ``fit_lbfgsb`` that differ between units (lengthscales, variance, noise, ...).
        common: keyword arguments shared by all units (is_isotropic, train_*, method, SciPy options).
        batch_lml_grad: the batched evaluation, ``romcomma_amd._lib.lml_grad_batch`` by default.
    Returns: per unit the dict ``fit_lbfgsb`` returns -- identical to what a fit of that unit alone returns -- or, for a unit whose
        fit raised (e.g. a matrix that is not positive definite), the exception; the other units are not affected by it.
    """
    if len(gps) != len(starts):
        raise ValueError('one start per unit')
    if batch_lml_grad is None or max_units is None:
        from romcomma_amd import _lib
        batch_lml_grad = batch_lml_grad or _lib.lml_grad_batch
        max_units = max_units or _lib.MAX_BATCH
    if len(gps) == 1:                                         # nothing to meet with
        try:
            return [fit_lbfgsb(gps[0], **(dict(common) | dict(starts[0])))]
        except Exception as failure:
            return [failure]
    lockstep = _Lockstep(gps, batch_lml_grad, max_units)
    results: list = [None] * len(gps)

    def run(u: int):
        try:
            results[u] = fit_lbfgsb(_LockstepUnit(lockstep, u, gps[u]), **(dict(common) | dict(starts[u])))
        except BaseException as failure:                     # (the thread must leave the meeting point whatever happened)
            results[u] = failure
        finally:
            lockstep.leave(u)

    threads = [threading.Thread(target=run, args=(u,), name=f'rcgp-fit-{u}') for u in range(len(gps))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for r in results:
        if isinstance(r, BaseException) and not isinstance(r, Exception):
            raise r                                           # KeyboardInterrupt and the like
    return results


# ---------------------------------------------------------------------------------------------------------------------
# Covariant (dependent-output) GP: the reference's romcomma.gpf.models.MOGPR behind the same gf.optimizers.Scipy call
# ---------------------------------------------------------------------------------------------------------------------

CHOLESKY_DIAGONAL_LOWER_BOUND = 1.0e-3        # gpf/base.py:35


def variance_to_params(V: np.ndarray):
    """gpf.base.Variance.__init__ (gpf/base.py:71-96): the (L, L) matrix is held as its Cholesky factor, the diagonal through
    positive(lower=1e-3) -- returned here unconstrained -- and the strictly lower triangle row by row."""
    C = np.linalg.cholesky(np.asarray(V, dtype=np.float64))
    d = np.diag(C)
    if d.min() <= CHOLESKY_DIAGONAL_LOWER_BOUND:
        raise ValueError(f'The Cholesky diagonal of a Variance must be strictly greater than {CHOLESKY_DIAGONAL_LOWER_BOUND}.')   # :87-88
    L = C.shape[0]
    lower = np.array([C[i, j] for i in range(1, L) for j in range(i)], dtype=np.float64)
    return inv_softplus(d - CHOLESKY_DIAGONAL_LOWER_BOUND), lower


def params_to_cholesky(u_diag: np.ndarray, lower: np.ndarray) -> np.ndarray:
    """gpf.base.Variance.cholesky (gpf/base.py:42-50)."""
    L = len(u_diag)
    C = np.zeros((L, L))
    k = 0
    for i in range(1, L):
        C[i, :i] = lower[k:k + i]
        k += i
    C[np.diag_indices(L)] = CHOLESKY_DIAGONAL_LOWER_BOUND + softplus(u_diag)
    return C


def _cholesky_chain(dV: np.ndarray, C: np.ndarray, u_diag: np.ndarray):
    """d/d(u_diag), d/d(lower) from d/dV taken entry by entry, V = C C^T."""
    dC = (dV + dV.T) @ C
    L = C.shape[0]
    lower = np.array([dC[i, j] for i in range(1, L) for j in range(i)], dtype=np.float64)
    return np.diag(dC) * sigmoid(u_diag), lower


def fit_lbfgsb_mo(gp, lengthscales, F, Sigma, is_isotropic: bool = False, train_kernel_variance: bool = True,
                  train_kernel_covariance: bool = False, train_lengthscales: bool = False, train_likelihood_variance: bool = True,
                  train_likelihood_covariance: bool = True, method: str = 'L-BFGS-B', callback=None, **options: Any) -> Dict[str, Any]:
    """Minimise -LML of a covariant GP (``romcomma_amd._lib.RcMOGP``) over its trainable parameters. The defaults are what
    Kernel.calibrate and Likelihood.calibrate switch on for a covariant GP (gpr/kernels.py:56-65, gpr/models.py:60,74-76): the
    Cholesky diagonal of the kernel variance and the whole Cholesky factor of the likelihood variance; the kernel's lower
    triangle and the lengthscales stay where the independent fit left them.

    Returns dict(lengthscales (L, M), variance (L, L), noise (L, L), log_marginal, result, nfev)."""
    L, M = gp.L, gp.M
    n_ell = 1 if is_isotropic else M
    ell0 = np.broadcast_to(np.asarray(lengthscales, dtype=np.float64).reshape(L, -1)[:, :n_ell], (L, n_ell)).copy()
    kd, kl = variance_to_params(F)
    nd, nl = variance_to_params(Sigma)
    parts = {'kd': kd, 'kl': kl, 'ue': inv_softplus(ell0).reshape(-1), 'nd': nd, 'nl': nl}
    names = [n for n, on in (('kd', train_kernel_variance), ('kl', train_kernel_covariance), ('ue', train_lengthscales),
                             ('nd', train_likelihood_variance), ('nl', train_likelihood_covariance)) if on and len(parts[n])]
    state = {'nfev': 0}

    def unpack(u):
        p = dict(parts)
        k = 0
        for n in names:
            p[n] = u[k:k + len(parts[n])]
            k += len(parts[n])
        return p

    def build(p):
        Ck, Cn = params_to_cholesky(p['kd'], p['kl']), params_to_cholesky(p['nd'], p['nl'])
        ell = np.broadcast_to(softplus(p['ue']).reshape(L, n_ell), (L, M))
        Fm, Sm = Ck @ Ck.T, Cn @ Cn.T
        return ell, Ck, Cn, (Fm + Fm.T) / 2, (Sm + Sm.T) / 2

    def objective(u):
        p = unpack(u)
        ell, Ck, Cn, Fm, Sm = build(p)
        gp.set_hyper(ell, Fm, Sm)
        lml, gF, gell, gS = gp.lml_grad()
        state['nfev'] += 1
        g = {}
        g['kd'], g['kl'] = _cholesky_chain(gF, Ck, p['kd'])
        g['nd'], g['nl'] = _cholesky_chain(gS, Cn, p['nd'])
        g_ell = gell.sum(axis=1, keepdims=True) if is_isotropic else gell
        g['ue'] = (g_ell * sigmoid(p['ue'].reshape(L, n_ell))).reshape(-1)
        return -lml, -np.concatenate([g[n] for n in names])

    opts = {'maxiter': 5000, 'gtol': 1e-16} | options
    u0 = np.concatenate([parts[n] for n in names]) if names else np.zeros(0)
    result = scipy.optimize.minimize(objective, u0, jac=True, method=method, options=opts, callback=callback) if names else None
    ell, Ck, Cn, Fm, Sm = build(unpack(result.x if names else u0))
    gp.set_hyper(ell, Fm, Sm)
    return {'lengthscales': np.array(ell), 'variance': Fm, 'noise': Sm, 'log_marginal': gp.lml(), 'result': result, 'nfev': state['nfev']}
