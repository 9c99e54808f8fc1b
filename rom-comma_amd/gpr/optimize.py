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


class _FitProblem:
    """The optimisation problem of one independent-output GP as GPflow poses it (module docstring): the packed unconstrained vector,
    which of its entries train, the map to the constrained hyper-parameters and the chain rule back. Shared by ``fit_lbfgsb`` (SciPy
    drives it) and the reverse-communication driver of ``fit_lbfgsb_batch``."""

    def __init__(self, M: int, lengthscales, variance: float, noise: float, is_isotropic: bool, train_lengthscales: bool,
                 train_variance: bool, train_noise: bool):
        self.M, self.is_isotropic = M, is_isotropic
        ell0 = np.broadcast_to(np.asarray(lengthscales, dtype=np.float64).reshape(-1), (1,) if is_isotropic else (M,)).copy()
        variance = max(float(variance), KERNEL_VARIANCE_FLOOR)
        noise = max(float(noise), LIKELIHOOD_VARIANCE_FLOOR)
        self.n_ell = ell0.shape[0]
        self.u_all = np.concatenate([inv_softplus(ell0), [inv_softplus(variance)], [inv_softplus(noise - LIKELIHOOD_LOWER)]])
        self.mask = np.array([train_lengthscales] * self.n_ell + [train_variance, train_noise])
        self.nfev = 0

    @property
    def x0(self) -> np.ndarray:
        return self.u_all[self.mask]

    def unpack(self, u_train):
        """(u, lengthscales (n_ell,), variance, noise) at the trainable values ``u_train``."""
        u, n_ell = self.u_all.copy(), self.n_ell
        u[self.mask] = u_train
        return u, softplus(u[:n_ell]), float(softplus(u[n_ell])), float(LIKELIHOOD_LOWER + softplus(u[n_ell + 1]))

    def theta(self, u_train):
        """(u, (lengthscales (M,), variance, noise)): what ``set_hyper`` takes."""
        u, ell, var, nse = self.unpack(u_train)
        return u, (np.broadcast_to(ell, (self.M,)), var, nse)

    def loss_and_gradient(self, u, lml: float, grad: np.ndarray):
        """(-LML, its gradient w.r.t. the trainable unconstrained entries) from the library's LML and constrained-space gradient."""
        M = self.M
        self.nfev += 1
        g_ell = np.array([np.sum(grad[:M])]) if self.is_isotropic else grad[:M]
        g = np.concatenate([g_ell, grad[M:]]) * sigmoid(u)
        return -lml, -g[self.mask]

    def finish(self, gp, result) -> Dict[str, Any]:
        """Set the optimum on the handle (normally the point of the last evaluation: nothing is recomputed) and report the fit."""
        if result is not None:
            self.u_all[self.mask] = result.x
        _, ell, var, nse = self.unpack(self.u_all[self.mask])
        gp.set_hyper(np.broadcast_to(ell, (self.M,)), var, nse)
        log_marginal = gp.lml()
        return {'lengthscales': np.broadcast_to(ell, (self.M,)).copy(), 'variance': var, 'noise': nse, 'log_marginal': log_marginal,
                'result': result, 'nfev': self.nfev}


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
    problem = _FitProblem(gp.M, lengthscales, variance, noise, is_isotropic, train_lengthscales, train_variance, train_noise)

    def objective(u_train):
        u, theta = problem.theta(u_train)
        gp.set_hyper(*theta)
        lml, grad = gp.lml_grad()
        return problem.loss_and_gradient(u, lml, grad)

    opts = {'maxiter': 5000, 'gtol': 1e-16} | options
    result = None
    if np.any(problem.mask):
        result = scipy.optimize.minimize(objective, problem.x0, jac=True, method=method, options=opts, callback=callback)
    return problem.finish(gp, result)


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


# ---- the same lockstep without threads: SciPy's L-BFGS-B core by reverse communication ----------------------------------------------
# scipy.optimize.minimize(method='L-BFGS-B') is a Python loop around the compiled routine ``setulb``, which RETURNS whenever it wants the
# objective at a point (scipy/optimize/_lbfgsb_py.py::_minimize_lbfgsb). Driving that routine directly, one state per unit, lets ONE thread
# advance every unit to its next request and answer all requests with one batched evaluation: the arithmetic -- hence every iterate, count
# and message -- is that of ``minimize``, but a round of 16 small units costs the host ~0.15 ms instead of ~1 ms of sixteen threads
# taking turns under the GIL. ``setulb`` is private to SciPy: the driver is used only where ``_RC_SELF_CHECK`` -- a
# small fit run both ways, bit for bit, the first time it is needed -- has passed on the installed SciPy; otherwise the threads above run.

_RC_OPTIONS = {'maxiter', 'gtol', 'ftol', 'maxcor', 'maxfun', 'maxls'}       # options this driver honours exactly as _minimize_lbfgsb does
_RC_STATE = {'checked': False, 'ok': False}


class _SetulbRun:
    """One unit's L-BFGS-B run: the work arrays of ``_minimize_lbfgsb`` and its loop, cut at the point where the objective is wanted."""

    def __init__(self, x0, maxiter=15000, gtol=1e-5, ftol=2.2204460492503131e-09, maxcor=10, maxfun=15000, maxls=20):
        from scipy.optimize import _lbfgsb_py
        self._py = _lbfgsb_py
        if not maxls > 0:
            raise ValueError('maxls must be positive.')
        self.m, self.maxiter, self.maxfun, self.maxls = int(maxcor), maxiter, maxfun, int(maxls)
        self.pgtol, self.factr = gtol, ftol / np.finfo(float).eps
        x0 = np.asarray(x0).ravel()
        n, m = x0.shape[0], self.m
        self.n = n
        self.nbd, self.low, self.up = np.zeros(n, np.int32), np.zeros(n, np.float64), np.zeros(n, np.float64)
        self.x = np.array(x0, dtype=np.float64)
        self.f, self.g = np.array(0.0, dtype=np.int32), np.zeros((n,), dtype=np.int32)
        self.wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
        self.iwa = np.zeros(3 * n, dtype=np.int32)
        self.task, self.ln_task = np.zeros(2, dtype=np.int32), np.zeros(2, dtype=np.int32)
        self.lsave, self.isave, self.dsave = np.zeros(4, dtype=np.int32), np.zeros(44, dtype=np.int32), np.zeros(29, dtype=np.float64)
        self.nit, self.nfev, self.done = 0, 0, False
        self._cached = None                                   # ScalarFunction evaluates at x0 on construction and serves the first request from it

    def advance(self) -> bool:
        """Run until the objective is wanted at ``self.x`` (True) or the run has ended (False)."""
        while True:
            self.g = self.g.astype(np.float64)
            self._py._lbfgsb.setulb(self.m, self.x, self.low, self.up, self.nbd, self.f, self.g, self.factr, self.pgtol, self.wa, self.iwa,
                                    self.task, self.lsave, self.isave, self.dsave, self.maxls, self.ln_task)
            if self.task[0] == 3:
                return True
            if self.task[0] == 1:
                self.nit += 1
                if self.nit >= self.maxiter:
                    self.task[0], self.task[1] = 5, 504
                elif self.nfev > self.maxfun:
                    self.task[0], self.task[1] = 5, 502
            else:
                self.done = True
                return False

    def answer(self, f: float, g: np.ndarray):
        self.f, self.g = f, g
        self.nfev += 1

    def result(self):
        from scipy.optimize import LbfgsInvHessProduct, OptimizeResult
        if self.task[0] == 4:
            warnflag = 0
        elif self.nfev > self.maxfun or self.nit >= self.maxiter:
            warnflag = 1
        else:
            warnflag = 2
        m, n = self.m, self.n
        s_, y_ = self.wa[0: m * n].reshape(m, n), self.wa[m * n: 2 * m * n].reshape(m, n)
        n_corrs = min(self.isave[30], m)
        message = self._py.status_messages[self.task[0]] + ': ' + self._py.task_messages[self.task[1]]
        return OptimizeResult(fun=self.f, jac=self.g, nfev=self.nfev, njev=self.nfev, nit=self.nit, status=warnflag, message=message, x=self.x,
                              success=(warnflag == 0), hess_inv=LbfgsInvHessProduct(s_[:n_corrs], y_[:n_corrs]))


def _fit_reverse_communication(gps, problems, options, batch_lml_grad, max_units: int, bind=None, release=None) -> list:
    """Every unit's L-BFGS-B run advanced by ONE thread; each round of requests answered by one batched evaluation.

    At most ``max_units`` runs are in flight: where there are more units, the next one STARTS when one in flight has ended (the runs need
    different numbers of evaluations; a batch that waited for its slowest member before the next batch began would idle the slots of the
    rest). ``bind(u)`` -- if given -- returns the device handle of unit ``u`` at that moment (a pool slot just freed, loaded with the unit's
    targets) and ``release(u)`` hands it back; without them ``gps[u]`` is the unit's own handle throughout. A unit's fit is finished
    (optimum set on its handle, LML read) as soon as its run ends, before its handle is released."""
    from collections import deque
    from romcomma_amd._lib import NotPositiveDefiniteError
    n = len(problems)
    gps = list(gps) if gps is not None else [None] * n
    results: list = [None] * n
    runs: Dict[int, _SetulbRun] = {}
    waiting: deque = deque(range(n))
    live: list = []

    def end(u: int, outcome):                                # the run of unit ``u`` is over: its result, its handle back
        try:
            results[u] = problems[u].finish(gps[u], outcome) if not isinstance(outcome, Exception) else outcome
        except Exception as failure:
            results[u] = failure
        if release is not None and gps[u] is not None:
            release(u)
            gps[u] = None

    while live or waiting:
        while waiting and len(live) < max_units:             # start as many waiting units as there are free places
            u = waiting.popleft()
            try:
                if bind is not None:
                    gps[u] = bind(u)
                if not np.any(problems[u].mask):             # nothing trains: the "fit" is the evaluation at the start point
                    end(u, None)
                    continue
                runs[u] = _SetulbRun(problems[u].x0, **options)
                live.append(u)
            except Exception as failure:
                end(u, failure)
        asking, points = [], {}
        for u in live:
            try:
                if runs[u].advance():
                    points[u] = problems[u].theta(np.copy(runs[u].x))
                    gps[u].set_hyper(*points[u][1])          # a point the library refuses fails ITS unit
                    asking.append(u)
                else:
                    end(u, runs[u].result())
            except Exception as failure:
                end(u, failure)
        live = []
        if asking:
            try:
                lml, grad, status = batch_lml_grad([gps[u] for u in asking])
            except Exception as failure:                     # a failed call fails every unit in it
                for u in asking:
                    end(u, failure)
                continue
            for k, u in enumerate(asking):
                if int(status[k]) > 0:
                    end(u, NotPositiveDefiniteError(int(status[k]),
                                                    f'rcgp_lml_grad_batch: matrix is not positive definite: leading minor {int(status[k])}'))
                    continue
                f, g = problems[u].loss_and_gradient(points[u][0], float(lml[k]), np.array(grad[k]))
                runs[u].answer(f, g)
                live.append(u)
    return results


def _reverse_communication_ok() -> bool:
    """Whether the installed SciPy's ``setulb`` can be driven as above: decided once, by running a small problem through
    ``scipy.optimize.minimize`` and through ``_SetulbRun`` and asking for the same iterates, counts, message and printed result."""
    if _RC_STATE['checked']:
        return _RC_STATE['ok']
    _RC_STATE['checked'] = True
    try:
        rng = np.random.default_rng(0)
        A = rng.standard_normal((6, 6))
        A = A @ A.T + np.eye(6)
        b = rng.standard_normal(6)

        def fun(x):                                           # a smooth non-quadratic bowl: several line-search steps per iteration
            r = A @ x - b
            return float(0.5 * r @ r + np.sum(np.cosh(0.3 * x))), A.T @ r + 0.3 * np.sinh(0.3 * x)
        options = {'maxiter': 5000, 'gtol': 1e-16}
        want = scipy.optimize.minimize(fun, np.full(6, 2.0), jac=True, method='L-BFGS-B', options=options)
        run = _SetulbRun(np.full(6, 2.0), **options)
        while run.advance():
            run.answer(*fun(np.copy(run.x)))
        got = run.result()
        _RC_STATE['ok'] = bool(np.array_equal(got.x, want.x) and got.nfev == want.nfev and got.nit == want.nit and got.fun == want.fun and
                               got.message == want.message and got.status == want.status and str(got) == str(want))
    except Exception:
        _RC_STATE['ok'] = False
    return _RC_STATE['ok']


def fit_lbfgsb_batch(gps: Optional[Sequence[Any]], starts: Sequence[Dict[str, Any]], batch_lml_grad=None, max_units: Optional[int] = None,
                     driver: Optional[str] = None, bind=None, release=None, M: Optional[int] = None, **common: Any) -> list:
    """``fit_lbfgsb`` for several units (``romcomma_amd._lib.RcGP`` of one device, equal M and padded size) at once.

    Args:
        gps: one device handle per unit -- or None with ``bind`` / ``release``: FEWER handles than units, ``bind(u)`` returning the handle
            unit ``u`` runs on from the moment it starts (a pool slot loaded with its targets) and ``release(u)`` taking it back when its
            fit has ended; ``M`` = the handles' input dimension.
        starts: per unit the keyword arguments of ``fit_lbfgsb`` that differ between units (lengthscales, variance, noise, ...).
        common: keyword arguments shared by all units (is_isotropic, train_*, method, SciPy options).
        batch_lml_grad: the batched evaluation, ``romcomma_amd._lib.lml_grad_batch`` by default.
        max_units: how many units are in flight at once (default and upper bound: what one batched call takes). With more units than that
            the 'setulb' driver starts the next unit when one in flight has ended; the 'threads' driver takes them group after group.
        driver: 'setulb' (one thread driving SciPy's L-BFGS-B core by reverse communication), 'threads' (one ``minimize`` per unit in a
            thread of its own), or None = the environment variable RCGP_LOCKSTEP, else 'setulb' where it applies -- method L-BFGS-B, no
            callback, only the options ``_RC_OPTIONS``, and the self-check against ``minimize`` passed -- and 'threads' otherwise.
            Both give every unit the fit it has alone.
    Returns: per unit the dict ``fit_lbfgsb`` returns -- identical to what a fit of that unit alone returns -- or, for a unit whose
        fit raised (e.g. a matrix that is not positive definite), the exception; the other units are not affected by it.
    """
    n = len(starts)
    if gps is None:
        if bind is None or M is None:
            raise ValueError('handles, or bind / release and M')
    elif len(gps) != n:
        raise ValueError('one start per unit')
    if batch_lml_grad is None or max_units is None:
        from romcomma_amd import _lib
        batch_lml_grad = batch_lml_grad or _lib.lml_grad_batch
        max_units = min(max_units or _lib.MAX_BATCH, _lib.MAX_BATCH)
    max_units = max(1, int(max_units))
    handle_of = (lambda u: bind(u)) if bind is not None else (lambda u: gps[u])
    give_back = release if (bind is not None and release is not None) else (lambda u: None)

    def one_after_the_other(units) -> list:                   # nothing to meet with
        out = []
        for u in units:
            try:
                gp = handle_of(u)
                try:
                    out.append(fit_lbfgsb(gp, **(dict(common) | dict(starts[u]))))
                finally:
                    give_back(u)
            except Exception as failure:
                out.append(failure)
        return out

    if n == 1 or max_units == 1:
        return one_after_the_other(range(n))
    if driver is None:
        import os
        driver = os.environ.get('RCGP_LOCKSTEP', 'setulb')
    if driver == 'setulb':
        merged = [dict(common) | dict(start) for start in starts]
        keys = ('lengthscales', 'variance', 'noise', 'is_isotropic', 'train_lengthscales', 'train_variance', 'train_noise')
        plain = all(m.get('method', 'L-BFGS-B') == 'L-BFGS-B' and m.get('callback') is None and
                    set(m) - set(keys) - {'method', 'callback'} <= _RC_OPTIONS for m in merged)
        options = [{k: v for k, v in m.items() if k in _RC_OPTIONS} for m in merged]
        if plain and all(o == options[0] for o in options) and _reverse_communication_ok():
            dims = [gp.M for gp in gps] if gps is not None else [int(M)] * n
            problems = [_FitProblem(d, m['lengthscales'], m['variance'], m['noise'], m.get('is_isotropic', False),
                                    m.get('train_lengthscales', True), m.get('train_variance', True), m.get('train_noise', True))
                        for d, m in zip(dims, merged)]
            return _fit_reverse_communication(gps, problems, {'maxiter': 5000, 'gtol': 1e-16} | options[0], batch_lml_grad, max_units,
                                              bind, release)
    results: list = [None] * n
    for first in range(0, n, max_units):                      # the threads meet group by group
        members = list(range(first, min(first + max_units, n)))
        handles: Dict[int, Any] = {}
        for u in members:
            try:
                handles[u] = handle_of(u)
            except Exception as failure:
                results[u] = failure
        members = [u for u in members if u in handles]
        if len(members) == 1:
            try:
                results[members[0]] = fit_lbfgsb(handles[members[0]], **(dict(common) | dict(starts[members[0]])))
            except Exception as failure:
                results[members[0]] = failure
        elif members:
            lockstep = _Lockstep([handles[u] for u in members], batch_lml_grad, max_units)

            def run(k: int, u: int):
                try:
                    results[u] = fit_lbfgsb(_LockstepUnit(lockstep, k, handles[u]), **(dict(common) | dict(starts[u])))
                except BaseException as failure:             # (the thread must leave the meeting point whatever happened)
                    results[u] = failure
                finally:
                    lockstep.leave(k)

            threads = [threading.Thread(target=run, args=(k, u), name=f'rcgp-fit-{u}') for k, u in enumerate(members)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        for u in handles:
            give_back(u)
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
