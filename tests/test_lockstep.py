"""Host logic of the lockstep fit drivers (gpr/optimize.py::fit_lbfgsb_batch) on the CPU: the units' L-BFGS-B runs meet in one batched
evaluation per round (the device calls are answered by the oracle here -- test infrastructure, never the product). What the reference
does in its place: one gf.optimizers.Scipy().minimize per output, one after the other (gpr/models.py:359-361). Two drivers, the same
fits: 'threads' (one scipy.optimize.minimize per unit in a thread of its own) and 'setulb' (one thread driving SciPy's compiled L-BFGS-B
routine by reverse communication)."""
import numpy as np
import pytest

from oracle import gp_oracle as o
from romcomma_amd.gpr.optimize import fit_lbfgsb, fit_lbfgsb_batch


class Unit:
    """The calls fit_lbfgsb makes on a handle, answered by the oracle."""

    def __init__(self, X, y, refuse_after=None):
        self.X, self.y, self.M, self.refuse_after, self.sets = X, y, X.shape[1], refuse_after, 0

    def set_hyper(self, ell, variance, noise):
        self.sets += 1
        if self.refuse_after is not None and self.sets > self.refuse_after:
            raise ValueError('refused')
        self.theta = (np.array(ell), float(variance), float(noise))

    def lml_grad(self):
        return o.lml_and_grad(self.X, self.y, *self.theta)

    def lml(self):
        return o.lml(self.X, self.y, *self.theta)


def batched(calls):
    def call(gps):
        calls.append(len(gps))
        out = [gp.lml_grad() for gp in gps]
        return np.array([a for a, _ in out]), np.array([b for _, b in out]), np.zeros(len(gps), dtype=np.int32)
    return call


START = dict(lengthscales=5.0 * np.ones(3), variance=2.0, noise=0.02)
DRIVERS = pytest.mark.parametrize('driver', ['threads', 'setulb'])


def test_the_setulb_driver_is_in_use_on_this_scipy():
    from romcomma_amd.gpr import optimize
    assert optimize._reverse_communication_ok()                # scipy.optimize.minimize and the bare routine agree bit for bit here


def same_fit(a, b):
    return (a['nfev'] == b['nfev'] and a['log_marginal'] == b['log_marginal'] and np.array_equal(a['lengthscales'], b['lengthscales']) and
            a['variance'] == b['variance'] and a['noise'] == b['noise'] and str(a['result']) == str(b['result']))


@DRIVERS
def test_lockstep_fits_equal_the_fits_alone_with_more_units_than_a_call_takes(driver):
    units = [Unit(*o.synthetic_fold(110 + 3 * k, 3, k=k)) for k in range(5)]
    alone = [fit_lbfgsb(u, **START) for u in units]
    calls = []
    together = fit_lbfgsb_batch(units, [START] * 5, batch_lml_grad=batched(calls), max_units=2, driver=driver)
    assert max(calls) == 2 and min(calls) == 1                 # five live units: calls of 2 + 2 + 1, fewer as units converge
    for a, b in zip(alone, together):
        assert same_fit(a, b)                                  # iterates, counts, message, the printed OptimizeResult that goes into meta.json
    assert len({a['nfev'] for a in alone}) > 1                 # the units leave at different rounds


@DRIVERS
def test_every_kind_of_fit_the_host_asks_for(driver):
    """Isotropic, parameters held fixed, other optimiser options, nothing trainable: each unit of a batch its own variant."""
    variants = [dict(START, lengthscales=5.0, is_isotropic=True), dict(START, train_noise=False), dict(START, train_lengthscales=False),
                dict(START, train_lengthscales=False, train_variance=False, train_noise=False)]
    for options in ({}, {'maxiter': 7}, {'ftol': 1e-4, 'maxcor': 5, 'maxls': 10}):
        units = [Unit(*o.synthetic_fold(90, 3, k=k)) for k in range(4)]
        alone = [fit_lbfgsb(u, **v, **options) for u, v in zip(units, variants)]
        together = fit_lbfgsb_batch(units, variants, batch_lml_grad=batched([]), max_units=8, driver=driver, **options)
        for a, b in zip(alone, together):
            assert same_fit(a, b), options
        assert together[3]['result'] is None and together[3]['nfev'] == 0


def test_options_the_setulb_driver_does_not_know_go_to_the_threads():
    seen = []
    units = [Unit(*o.synthetic_fold(80, 3, k=k)) for k in range(2)]
    together = fit_lbfgsb_batch(units, [START] * 2, batch_lml_grad=batched([]), max_units=8, callback=lambda *a: seen.append(1))
    assert seen and all(isinstance(t, dict) for t in together)  # a callback: scipy.optimize.minimize itself ran (in threads)


@DRIVERS
def test_a_point_the_library_refuses_fails_its_unit_only(driver):
    units = [Unit(*o.synthetic_fold(100, 3, k=k), refuse_after=(5 if k == 1 else None)) for k in range(3)]
    out = fit_lbfgsb_batch(units, [START] * 3, batch_lml_grad=batched([]), max_units=8, driver=driver)
    assert isinstance(out[1], ValueError) and isinstance(out[0], dict) and isinstance(out[2], dict)
    alone = fit_lbfgsb(Unit(*o.synthetic_fold(100, 3, k=2)), **START)
    assert out[2]['nfev'] == alone['nfev'] and out[2]['log_marginal'] == alone['log_marginal']


@DRIVERS
def test_a_failed_batched_call_fails_the_units_of_its_round_and_nobody_hangs(driver):
    units = [Unit(*o.synthetic_fold(90, 3, k=k)) for k in range(3)]
    state = {'n': 0}

    def flaky(gps):
        state['n'] += 1
        if state['n'] == 4:
            raise RuntimeError('device lost')
        return batched([])(gps)
    out = fit_lbfgsb_batch(units, [START] * 3, batch_lml_grad=flaky, max_units=8, driver=driver)
    assert all(isinstance(r, RuntimeError) for r in out)


@DRIVERS
def test_status_words_become_not_positive_definite_errors(driver):
    from romcomma_amd._lib import NotPositiveDefiniteError
    units = [Unit(*o.synthetic_fold(80, 3, k=k)) for k in range(2)]

    def second_unit_singular(gps):
        lml, grad, status = batched([])(gps)
        if len(gps) == 2:
            status[1] = 17
        return lml, grad, status
    out = fit_lbfgsb_batch(units, [START] * 2, batch_lml_grad=second_unit_singular, max_units=8, driver=driver)
    assert isinstance(out[1], NotPositiveDefiniteError) and out[1].k == 17 and isinstance(out[0], dict)


@DRIVERS
def test_more_units_than_handles_take_turns_on_the_handles(driver):
    """Seven units through three handles (bind / release: HipGP.calibrate with more outputs than pool slots): every fit is the fit the
    unit has alone; never more than three handles out; with the setulb driver a waiting unit starts the moment a handle falls free, so
    the device sees fewer calls than with groups of three that each wait for their slowest member."""
    folds = [o.synthetic_fold(100 + 4 * k, 3, k=k) for k in range(7)]
    alone = [fit_lbfgsb(Unit(*fold), **START) for fold in folds]
    handles = [Unit(*folds[0]) for _ in range(3)]
    free, held, most, order = list(range(3)), {}, [0], []

    def bind(u):
        slot = free.pop(0)
        held[u] = slot
        most[0] = max(most[0], len(held))
        order.append(u)
        handles[slot].X, handles[slot].y = folds[u]              # (the pool loads the unit's targets; here the whole fold)
        return handles[slot]

    def release(u):
        free.append(held.pop(u))

    calls = []
    together = fit_lbfgsb_batch(None, [START] * 7, batch_lml_grad=batched(calls), max_units=3, driver=driver, bind=bind, release=release, M=3)
    for a, b in zip(alone, together):
        assert same_fit(a, b)
    assert most[0] == 3 and not held and sorted(free) == [0, 1, 2] and sorted(order) == list(range(7)) and max(calls) <= 3
    if driver == 'setulb':
        evaluations = [a['nfev'] for a in alone]
        grouped = sum(max(evaluations[first:first + 3]) for first in range(0, 7, 3))      # rounds of group-after-group lockstep
        assert len(calls) < grouped and sum(calls) == sum(evaluations)
    with pytest.raises(ValueError):
        fit_lbfgsb_batch(None, [START] * 2, batch_lml_grad=batched([]), bind=bind)         # without M nobody knows the problem's size


def test_a_unit_that_cannot_get_its_handle_fails_alone():
    folds = [o.synthetic_fold(90, 3, k=k) for k in range(4)]
    handle = [Unit(*folds[0]), Unit(*folds[0])]
    free = [0, 1]
    held = {}

    def bind(u):
        if u == 1:
            raise MemoryError('no room')
        held[u] = free.pop(0)
        handle[held[u]].X, handle[held[u]].y = folds[u]
        return handle[held[u]]

    for driver in ('setulb', 'threads'):
        out = fit_lbfgsb_batch(None, [START] * 4, batch_lml_grad=batched([]), max_units=2, driver=driver, bind=bind,
                               release=lambda u: free.append(held.pop(u)), M=3)
        assert isinstance(out[1], MemoryError) and all(isinstance(out[u], dict) for u in (0, 2, 3)) and sorted(free) == [0, 1]
        alone = fit_lbfgsb(Unit(*folds[3]), **START)
        assert same_fit(alone, out[3])
