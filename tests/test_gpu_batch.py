"""GPU tests of the batched entries (rcgp_lml_grad_batch, rcgp_factor_batch): several (fold, output) units in ONE schedule on one GPU, where
the reference walks its outputs and folds one after the other (gpr/models.py:340-342, 360-361; user/run.py:60-61).

What is held: (1) every unit's numbers are BIT-identical to the single-handle call -- whoever shares the launch, whatever the batch size;
(2) parity with the oracle on batches of 2 and 3, units of different N under one padded size included; (3) a unit whose matrix is not
positive definite is reported in its status word while the other units finish; (4) the lockstep L-BFGS-B driver gives every unit the fit
it has alone; (5) HipGP.calibrate on L independent outputs is the same whether the outputs are fitted at once or in turn.
"""
import numpy as np
import pytest

from oracle import gp_oracle as o          # the checker, never the thing measured

pytestmark = pytest.mark.gpu


def _units(gpu, sizes, M, seed_offset=0):
    gps, data = [], []
    for u, N in enumerate(sizes):
        X, y = o.synthetic_fold(N, M, k=seed_offset + u)
        gps.append(gpu.RcGP(X, y))
        data.append((X, y))
    return gps, data


def _thetas(M, n):
    ell, var, noise = o.bench_hyper(M)
    return [(ell * (1.0 + 0.07 * u), var * (1.0 + 0.2 * u), noise * (1.0 + u)) for u in range(n)]


@pytest.mark.parametrize('sizes', [(700, 650, 768), (1500, 1500), (2100, 2050, 2176, 2049, 2100), (5000, 5100, 4999)])
def test_batch_is_bit_identical_to_the_single_handle_call(gpu, sizes):
    """Units of different N under one padded size, batches of every size up to len(sizes): LML, gradient and K_inv_Y equal to the last bit
    to what each handle returns on its own. (The factorisation's schedule, the tile shapes of the L^-1 levels and the launch a unit
    shares do not enter a unit's arithmetic: every tile adds up its k-slabs in one fixed order.) At N = 5000 a batch runs six outer panels of
    512 columns ahead of a 16-block tail while the single call is all tail: two different schedules, the same bits."""
    M = 4
    gps, _ = _units(gpu, sizes, M)
    thetas = _thetas(M, len(sizes))
    other = [(e * 1.03, v, n) for e, v, n in thetas]
    for gp, t in zip(gps, thetas):
        gp.set_hyper(*t)
    single = [gp.lml_grad() for gp in gps]
    alpha = [gp.k_inv_y() for gp in gps]
    for nb in range(1, len(sizes) + 1):
        for gp, t in zip(gps, other):
            gp.set_hyper(*t)
        gpu.lml_grad_batch(gps[:nb])                          # (somewhere else first, so that nothing below is served from a cache)
        for gp, t in zip(gps, thetas):
            gp.set_hyper(*t)
        lml, grad, status = gpu.lml_grad_batch(gps[:nb])
        assert np.all(status == 0)
        for u in range(nb):
            assert lml[u] == single[u][0], (nb, u)
            assert np.array_equal(grad[u], single[u][1]), (nb, u)
            assert np.array_equal(gps[u].k_inv_y(), alpha[u]), (nb, u)      # cached by the batched call: no refactorisation here
    for gp in gps:
        gp.close()


@pytest.mark.parametrize('sizes', [(520, 600), (900, 1000, 1024), (4200, 4224)])       # (the last: a batched schedule with outer panels)
def test_batch_against_oracle(gpu, sizes):
    M = 5
    gps, data = _units(gpu, sizes, M, seed_offset=3)
    thetas = _thetas(M, len(sizes))
    for gp, t in zip(gps, thetas):
        gp.set_hyper(*t)
    lml, grad, status = gpu.lml_grad_batch(gps)
    assert np.all(status == 0)
    for u, ((X, y), t) in enumerate(zip(data, thetas)):
        ref_lml, ref_grad = o.lml_and_grad(X, y, *t)
        assert lml[u] == pytest.approx(ref_lml, rel=1e-10)
        np.testing.assert_allclose(grad[u], ref_grad, rtol=1e-7, atol=1e-9 * np.max(np.abs(ref_grad)))
        np.testing.assert_allclose(gps[u].k_inv_y(), o.k_inv_y(X, y, *t), rtol=0, atol=1e-9 * np.max(np.abs(y)) / t[2])
    # the cached factors serve predict and Sobol of each unit afterwards
    Xs, _ = o.synthetic_fold(40, M, k=11)
    for u, ((X, y), t) in enumerate(zip(data, thetas)):
        mean, sd = gps[u].predict(Xs)
        rmean, rsd = o.predict(X, y, *t, Xs)
        np.testing.assert_allclose(mean, rmean, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(sd, rsd, rtol=1e-6)
    for gp in gps:
        gp.close()


@pytest.mark.parametrize('N,twin', [(900, 700), (3000, 700), (3000, 2900)])
def test_a_unit_that_is_not_positive_definite_leaves_the_others_alone(gpu, N, twin):
    """Duplicate rows with zero noise on unit 1 (TensorFlow raises InvalidArgumentError from tf.linalg.cholesky there, gpr/models.py:439):
    its status word names a leading minor, its numbers are NaN, units 0 and 2 get exactly what they get alone; afterwards the bad unit
    is usable again. N = 3000: the batched schedule runs outer panels ahead of its tail, and the unit fails inside the first panel (row
    700) or inside the tail (row 2900) while the others' panels go on."""
    M = 3
    gps, data = _units(gpu, (N, N, N), M, seed_offset=20)
    X1, y1 = data[1]
    X1 = X1.copy()
    X1[twin] = X1[30]
    gps[1].close()
    gps[1] = gpu.RcGP(X1, y1)
    thetas = _thetas(M, 3)
    thetas[1] = (thetas[1][0], thetas[1][1], 0.0)
    for gp, t in zip(gps, thetas):
        gp.set_hyper(*t)
    alone = [gps[u].lml_grad() for u in (0, 2)]
    for gp, t in zip(gps, thetas):
        gp.set_hyper(t[0] * 1.01, t[1], t[2])
    gpu.lml_grad_batch([gps[0], gps[2]])
    for gp, t in zip(gps, thetas):
        gp.set_hyper(*t)
    lml, grad, status = gpu.lml_grad_batch(gps)
    assert status[0] == 0 and status[2] == 0 and 1 <= status[1] <= N
    assert np.isnan(lml[1]) and np.all(np.isnan(grad[1]))
    assert lml[0] == alone[0][0] and np.array_equal(grad[0], alone[0][1])
    assert lml[2] == alone[1][0] and np.array_equal(grad[2], alone[1][1])
    with pytest.raises(gpu.NotPositiveDefiniteError):
        gps[1].lml()                                           # the failed factor was dropped, not cached
    gps[1].set_hyper(thetas[1][0], thetas[1][1], 1e-2)
    assert np.all(gpu.factor_batch(gps) == 0)
    assert np.isfinite(gps[1].lml())
    for gp in gps:
        gp.close()


def test_batch_arguments_are_checked(gpu):
    M = 3
    gps, _ = _units(gpu, (300, 300), M)
    big, _ = _units(gpu, (600,), M)
    gps[0].set_hyper(*_thetas(M, 1)[0])
    with pytest.raises(gpu.RcgpError, match='hyper-parameters not set'):
        gpu.lml_grad_batch(gps)
    gps[1].set_hyper(*_thetas(M, 1)[0])
    big[0].set_hyper(*_thetas(M, 1)[0])
    with pytest.raises(gpu.RcgpError, match='padded size'):
        gpu.lml_grad_batch([gps[0], big[0]])
    with pytest.raises(gpu.RcgpError, match='twice'):
        gpu.lml_grad_batch([gps[0], gps[0]])
    with pytest.raises(ValueError):
        gpu.lml_grad_batch([])
    for gp in gps + big:
        gp.close()


def test_lockstep_fits_are_the_fits_the_units_have_alone(gpu):
    """Three units fitted at once (one optimiser thread each, their evaluations meeting in rcgp_lml_grad_batch; the unit that converges
    first leaves and the others go on) against the same three fitted one after the other: same evaluation counts, same optimum, bit for
    bit -- and against the oracle's fit from the same start (LML* within 1e-5 relative: SURVEY.md 8c)."""
    from romcomma_amd.gpr.optimize import fit_lbfgsb, fit_lbfgsb_batch
    M = 3
    gps, data = _units(gpu, (400, 420, 390), M, seed_offset=40)
    start = dict(lengthscales=5.0 * np.ones(M), variance=2.0, noise=0.02)
    alone = [fit_lbfgsb(gp, **start) for gp in gps]
    for gp in gps:
        gp.set_hyper(np.ones(M), 1.0, 0.5)
    together = fit_lbfgsb_batch(gps, [start] * 3)
    assert len({fit['nfev'] for fit in alone}) > 1               # the units really finish at different rounds
    for (X, y), a, b in zip(data, alone, together):
        assert a['nfev'] == b['nfev']
        assert np.array_equal(a['lengthscales'], b['lengthscales']) and a['variance'] == b['variance'] and a['noise'] == b['noise']
        assert a['log_marginal'] == b['log_marginal']
        ref = o.fit(X, y, 5.0 * np.ones(M))
        assert b['log_marginal'] == pytest.approx(ref['lml'], rel=1e-5)
    for gp in gps:
        gp.close()


def test_lockstep_fit_with_a_unit_that_fails(gpu, tmp_path):
    """The fold of test_gpu_parity.test_fit_that_runs_into_a_singular_gram_matrix... (its line search reaches a matrix that is not positive
    definite) fitted in lockstep with its sibling folds: it fails with NotPositiveDefiniteError at the same leading minor as alone, at
    whatever round that happens, and every other unit ends exactly where it ends alone."""
    import pandas as pd
    from romcomma_amd.data.storage import Fold, Repository
    from romcomma_amd.gpr.optimize import fit_lbfgsb, fit_lbfgsb_batch
    rng = np.random.default_rng(0)
    U = rng.random((200, 3))
    y = np.sin(2 * np.pi * U[:, 0]) + 0.7 * U[:, 1] ** 2 + 0.05 * U[:, 2] + 0.02 * rng.standard_normal(200)
    columns = pd.MultiIndex.from_tuples([('X', f'X.{m}') for m in range(3)] + [('Y', 'Y.0')])
    repo = Repository.from_df(tmp_path / 'repo', pd.DataFrame(np.concatenate([U, y[:, None]], axis=1), columns=columns)).into_K_folds(-4, seed=5)
    gps = []
    for k in range(4):
        fold = Fold(repo, k)
        gps.append(gpu.RcGP(np.ascontiguousarray(fold.X.values), np.ascontiguousarray(fold.Y.values[:, 0])))
    start = dict(lengthscales=5.0 * np.ones(3), variance=2.0, noise=0.02)

    def alone(gp):
        try:
            return fit_lbfgsb(gp, **start)
        except gpu.NotPositiveDefiniteError as failure:
            return failure
    expected = [alone(gp) for gp in gps]
    assert isinstance(expected[1], gpu.NotPositiveDefiniteError)
    assert any(isinstance(e, dict) for e in expected)
    for gp in gps:
        gp.set_hyper(np.ones(3), 1.0, 0.5)
    got = fit_lbfgsb_batch(gps, [start] * 4)
    for e, g in zip(expected, got):
        if isinstance(e, dict):
            assert isinstance(g, dict) and g['nfev'] == e['nfev'] and g['log_marginal'] == e['log_marginal']
            assert np.array_equal(g['lengthscales'], e['lengthscales'])
        else:
            assert isinstance(g, gpu.NotPositiveDefiniteError) and g.k == e.k
    for gp in gps:
        gp.close()


def test_calibrate_fits_the_outputs_at_once_or_in_turn_alike(gpu, tmp_path):
    """HipGP.calibrate on L = 3 independent outputs with a pool of three units (lockstep, one batched schedule per round) and with one
    unit (the reference's loop, gpr/models.py:360-361): the same kernel/variance.csv, kernel/lengthscales.csv, likelihood/variance.csv and
    likelihood/log_marginal.csv to the last digit, and the same K_inv_Y and predictions from the handles the pool keeps."""
    import pandas as pd
    from romcomma_amd.data.storage import Fold, Repository
    from romcomma_amd.gpr.models import HipGP
    from romcomma_amd.user.sample import synthetic_outputs
    N, M, L = 640, 3, 3
    X, Y = synthetic_outputs(N + 64, M, L)
    columns = pd.MultiIndex.from_tuples([('X', f'X.{m}') for m in range(M)] + [('Y', f'Y.{l}') for l in range(L)])
    results = {}
    for units in (3, 1):
        repo = Repository.from_df(tmp_path / f'repo{units}', pd.DataFrame(np.concatenate([X, Y], axis=1), columns=columns)).into_K_folds(1, is_normalization_applicable=False)
        gp = HipGP('gp', Fold(repo, 0), is_read=False, is_covariant=False, is_isotropic=False, units_per_gpu=units)
        assert gp.pool_size == units
        gp.calibrate()
        frames = {name: pd.read_csv(gp.folder / name, index_col=0).to_numpy() for name in
                  ('kernel/variance.csv', 'kernel/lengthscales.csv', 'likelihood/variance.csv', 'likelihood/log_marginal.csv')}
        test_x = np.ascontiguousarray(gp.fold.test_x.values[:32])
        results[units] = (frames, gp.K_inv_Y, gp.predict(test_x), len(gp._units))
        gp.close()
    (fa, ka, pa, na), (fb, kb, pb, nb_) = results[3], results[1]
    assert na == 3 and nb_ == 1
    for name in fa:
        assert np.array_equal(fa[name], fb[name]), name
    assert np.array_equal(ka, kb)
    assert np.array_equal(pa[0], pb[0]) and np.array_equal(pa[1], pb[1])


def test_run_gpr_over_folds_at_once_writes_what_fold_after_fold_writes(gpu, tmp_path):
    """``run.gpr(units_per_gpu=4)`` on the device: the four folds of a split are calibrated together (HipGP.calibrate_group -> lockstep
    L-BFGS-B -> rcgp_lml_grad_batch; the folds' N differ by one row under one padded size) and tested; every parameter and test file, per
    fold and collected, equals byte for byte what the reference's order -- fold after fold (user/run.py:60-61) -- leaves. The isotropic ->
    anisotropic warm start runs through both."""
    import pandas as pd
    from romcomma_amd.data.storage import Repository
    from romcomma_amd.user import run
    rng = np.random.default_rng(4)
    U = rng.random((1001, 3))
    y = np.sin(2 * np.pi * U[:, 0]) + 0.6 * U[:, 1] ** 2 + 0.1 * U[:, 2] + 0.05 * rng.standard_normal(1001)
    columns = pd.MultiIndex.from_tuples([('X', f'X.{m}') for m in range(3)] + [('Y', 'Y.0')])
    table = pd.DataFrame(np.concatenate([U, y[:, None]], axis=1), columns=columns)
    repos = {}
    before = gpu.stat()['batched_calls']
    for units in (1, 4):
        repos[units] = Repository.from_df(tmp_path / f'units{units}', table).into_K_folds(-4, seed=2)
        assert run.gpr('gpr', repos[units], is_read=False, is_covariant=False, is_isotropic=None, units_per_gpu=units) == ['gpr.v.i', 'gpr.v.a']
        if units == 1:
            assert gpu.stat()['batched_calls'] == before           # one fold after the other: no batched call
    assert gpu.stat()['batched_calls'] > before + 20
    files = [f'{model}/{name}' for model in ('gpr.v.i', 'gpr.v.a') for name in
             ('kernel/lengthscales.csv', 'kernel/variance.csv', 'likelihood/variance.csv', 'likelihood/log_marginal.csv', 'test.csv', 'test_summary.csv')]
    for rel in [f'fold.{k}/{f}' for k in range(4) for f in files] + files:
        assert (repos[1].folder / rel).read_bytes() == (repos[4].folder / rel).read_bytes(), rel


def test_more_outputs_than_pool_slots_take_turns_on_the_slots(gpu, tmp_path):
    """L = 5 outputs on a pool of two handles: the third output starts on the handle of whichever of the first two ends first, and so on
    (fit_lbfgsb_batch with bind / release); the files are those of the one-after-the-other loop, and K_inv_Y / predict afterwards find
    every output on its ``l mod P`` slot again."""
    import pandas as pd
    from romcomma_amd.data.storage import Fold, Repository
    from romcomma_amd.gpr.models import HipGP
    from romcomma_amd.user.sample import synthetic_outputs
    N, M, L = 500, 3, 5
    X, Y = synthetic_outputs(N + 40, M, L)
    columns = pd.MultiIndex.from_tuples([('X', f'X.{m}') for m in range(M)] + [('Y', f'Y.{l}') for l in range(L)])
    results = {}
    for units in (2, 1):
        repo = Repository.from_df(tmp_path / f'repo{units}', pd.DataFrame(np.concatenate([X, Y], axis=1), columns=columns)).into_K_folds(1, is_normalization_applicable=False)
        gp = HipGP('gp', Fold(repo, 0), is_read=False, is_covariant=False, is_isotropic=False, units_per_gpu=units)
        assert gp.pool_size == units and gp.units_at_once == units
        before = gpu.stat()['batched_calls']
        gp.calibrate()
        batched_calls = gpu.stat()['batched_calls'] - before
        frames = {name: pd.read_csv(gp.folder / name, index_col=0).to_numpy() for name in
                  ('kernel/variance.csv', 'kernel/lengthscales.csv', 'likelihood/variance.csv', 'likelihood/log_marginal.csv')}
        meta = (gp.folder / 'meta.json').read_text()
        test_x = np.ascontiguousarray(gp.fold.test_x.values[:16])
        results[units] = (frames, gp.K_inv_Y, gp.predict(test_x), len(gp._units), batched_calls, meta)
        gp.close()
    (fa, ka, pa, na, ca, ma), (fb, kb, pb, nb_, cb, mb) = results[2], results[1]
    assert na == 2 and nb_ == 1 and ca > 20 and cb == 0
    for name in fa:
        assert np.array_equal(fa[name], fb[name]), name
    assert ma == mb                                              # every output's OptimizeResult as printed, in output order
    assert np.array_equal(ka, kb) and np.array_equal(pa[0], pb[0]) and np.array_equal(pa[1], pb[1])


def test_folds_whose_units_overfill_the_gpu_places(gpu, tmp_path):
    """Three folds of L = 3 outputs with four places: ``run.gpr`` opens two folds at a time (six units, four in flight, the other two
    starting as places fall free), then the third; byte for byte the files of fold after fold."""
    import pandas as pd
    from romcomma_amd.data.storage import Repository
    from romcomma_amd.user import run
    rng = np.random.default_rng(8)
    X = rng.random((900, 3))
    Y = np.stack([np.sin(2 * np.pi * X[:, l]) + 0.6 * X[:, (l + 1) % 3] ** 2 + 0.1 * X[:, (l + 2) % 3] for l in range(3)], axis=1)
    Y += 0.05 * rng.standard_normal(Y.shape)
    columns = pd.MultiIndex.from_tuples([('X', f'X.{m}') for m in range(3)] + [('Y', f'Y.{l}') for l in range(3)])
    table = pd.DataFrame(np.concatenate([X, Y], axis=1), columns=columns)
    repos = {}
    for units in (1, 4):
        repos[units] = Repository.from_df(tmp_path / f'units{units}', table).into_K_folds(-3, seed=2)
        before = gpu.stat()['batched_calls']
        assert run.gpr('gpr', repos[units], is_read=False, is_covariant=False, is_isotropic=None, units_per_gpu=units) == ['gpr.v.i', 'gpr.v.a']
        assert (gpu.stat()['batched_calls'] > before + 20) == (units == 4)
    files = [f'{model}/{name}' for model in ('gpr.v.i', 'gpr.v.a') for name in
             ('kernel/lengthscales.csv', 'kernel/variance.csv', 'likelihood/variance.csv', 'likelihood/log_marginal.csv', 'test.csv',
              'test_summary.csv')]
    for rel in [f'fold.{k}/{f}' for k in range(3) for f in files] + [f'fold.{k}/{model}/meta.json' for k in range(3) for model in ('gpr.v.i', 'gpr.v.a')] + files:
        assert (repos[1].folder / rel).read_bytes() == (repos[4].folder / rel).read_bytes(), rel


def test_wide_designs_in_a_batch(gpu):
    """M = 80 > 64 (the chunked Gram / gradient kernels) through the batched entry: bit-identical to the single-handle call and
    equal to the oracle."""
    M = 80
    gps, data = _units(gpu, (300, 330), M, seed_offset=70)
    rng = np.random.default_rng(1)
    thetas = [(rng.uniform(3.0, 9.0, M), 1.1 + 0.2 * u, 0.02) for u in range(2)]
    for gp, t in zip(gps, thetas):
        gp.set_hyper(*t)
    single = [gp.lml_grad() for gp in gps]
    for gp, t in zip(gps, thetas):
        gp.set_hyper(t[0] * 1.01, t[1], t[2])
    gpu.lml_grad_batch(gps)
    for gp, t in zip(gps, thetas):
        gp.set_hyper(*t)
    lml, grad, status = gpu.lml_grad_batch(gps)
    assert np.all(status == 0)
    for u, ((X, y), t) in enumerate(zip(data, thetas)):
        assert lml[u] == single[u][0] and np.array_equal(grad[u], single[u][1])
        ref_lml, ref_grad = o.lml_and_grad(X, y, *t)
        assert lml[u] == pytest.approx(ref_lml, rel=1e-10)
        np.testing.assert_allclose(grad[u], ref_grad, rtol=1e-7, atol=1e-9 * np.max(np.abs(ref_grad)))
    for gp in gps:
        gp.close()
