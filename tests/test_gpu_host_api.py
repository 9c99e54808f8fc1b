"""GPU end-to-end test of the drop-in surface: run.gpr + run.gsa on a small Repository (2 outputs, 2 folds), the reference's
folder layout and file formats, and the in-memory results against the oracle at the fitted hyper-parameters."""
import json
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

from oracle import gp_oracle as o

pytestmark = pytest.mark.gpu


def make_repo(folder: Path, N=240, M=3, L=2, seed=0):
    from romcomma_amd.data.storage import Repository
    rng = np.random.default_rng(seed)
    U = rng.random((N, M))
    Y = np.stack([np.sin(2 * np.pi * U[:, 0]) + (0.5 + l) * U[:, 1] ** 2 + 0.1 * (1 - l) * U[:, 2] + 0.02 * rng.standard_normal(N)
                  for l in range(L)], axis=1)
    columns = pd.MultiIndex.from_tuples([('X', f'X.{m}') for m in range(M)] + [('Y', f'Y.{l}') for l in range(L)])
    return Repository.from_df(folder, pd.DataFrame(np.concatenate([U, Y], axis=1), columns=columns))


def assert_T_matches(T_got, T_ref, V4, W_scale, what=''):
    """T = sqrt(|Q| / V[4]) (gsa/calibrators.py:333-346) is not a primary quantity: Q is a DIFFERENCE of W-sized terms
    (W_mm - 2 V_m W_Mm / V[1] + V_m^2 Q_full) that cancel to a few per cent of their size for a relevant input and to rounding
    noise for an irrelevant one, and the square root then magnifies what is left. north_star's 1e-5 therefore applies where the
    numbers are formed: to W (held to rtol 1e-5 next to this call) and to Q RELATIVE TO THE W SCALE, entry by entry --
    |Q_got - Q_ref| <= 3e-5 max|W| (three W-sized terms) -- which is what a consumer of T can rely on: T to
    3e-5 max|W| / (V4 (T_got + T_ref)), i.e. 1e-5-ish relative wherever T is not itself cancellation noise."""
    Q_got, Q_ref = T_got ** 2 * V4[..., None], T_ref ** 2 * V4[..., None]
    excess = np.abs(Q_got - Q_ref) - 3e-5 * W_scale
    assert np.all(excess <= 0), f'{what}: Q off by up to {np.max(np.abs(Q_got - Q_ref)):.3e} on a W scale of {W_scale:.3e}'


def test_run_gpr_gsa_end_to_end(gpu, tmp_path, monkeypatch):
    from romcomma_amd.data.storage import Fold
    from romcomma_amd.gpr.models import MOGP
    from romcomma_amd.gsa.models import GSA, Sobol
    from romcomma_amd.user import run
    repo = make_repo(tmp_path / 'repo').into_K_folds(-2, seed=3)
    names = run.gpr('gpr', repo, is_read=False, is_covariant=False, is_isotropic=None)
    assert names == ['gpr.v.i', 'gpr.v.a']
    fold = Fold(repo, 0)
    folder = fold.folder / 'gpr.v.a'
    files = sorted(str(p.relative_to(folder)) for p in folder.rglob('*') if p.is_file())
    assert files == ['kernel.csv', 'kernel/lengthscales.csv', 'kernel/variance.csv', 'likelihood/log_marginal.csv', 'likelihood/variance.csv',
                     'meta.json', 'test.csv', 'test_summary.csv']
    meta = json.loads((folder / 'meta.json').read_text())
    assert meta['maxiter'] == 5000 and meta['gtol'] == 1e-16 and 'result' in meta and meta['kernel']['variance'] is True
    assert (repo.folder / 'gpr.v.a' / 'test_summary.csv').exists() and (repo.folder / 'gpr.v.a' / 'kernel' / 'lengthscales.csv').exists()
    iso = MOGP('gpr.v.i', fold, True, False, True)
    assert iso.kernel.data.frames.lengthscales.np.shape == (2, 1)

    gp = MOGP('gpr.v.a', fold, True, False, False)
    ell = gp.kernel.data.frames.lengthscales.np
    var = gp.kernel.data.frames.variance.np[0]
    noise = gp.likelihood.data.frames.variance.np[0]
    lml = gp.likelihood.data.frames.log_marginal.np[0]
    X, Y = gp.X, gp.Y
    assert ell.shape == (2, 3)
    for l in range(2):
        assert lml[l] == pytest.approx(o.lml(X, Y[:, l], ell[l], var[l], noise[l]), rel=1e-8)
        # the optimum found on the GPU is a stationary point of the oracle's objective too
        _, grad = o.lml_and_grad(X, Y[:, l], ell[l], var[l], noise[l])
        assert np.max(np.abs(grad * np.concatenate([ell[l], [var[l], noise[l]]]))) < 2e-2 * abs(lml[l]) ** 0 * 5
    assert np.max(gp.check_K_inv_Y(fold.test_x.values[:20])) < 1e-8
    mean, sd = gp.predict(fold.test_x.values)
    for l in range(2):
        mr, sr = o.predict(X, Y[:, l], ell[l], var[l], noise[l], fold.test_x.values)
        np.testing.assert_allclose(mean[:, l], mr, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(sd[:, l], sr, rtol=1e-7)
    summary = pd.read_csv(folder / 'test_summary.csv', header=[0, 1], index_col=0)
    rmse = np.sqrt(np.mean((fold.test_y.values - mean) ** 2, axis=0))
    np.testing.assert_allclose(summary['RMSE'].values[0], rmse, rtol=1e-5)
    assert np.all(rmse < 0.3)                               # sanity only: out-of-sample error of z-scored outputs, depends on the split (0.14, 0.22 here)
    gp.close()

    gsa_names = run.gsa('gpr', repo, is_covariant=False, is_isotropic=False)
    assert [str(n) for n in gsa_names] == ['gpr.v.a/gsa/first_order', 'gpr.v.a/gsa/closed', 'gpr.v.a/gsa/total']
    S_csv = pd.read_csv(fold.folder / 'gpr.v.a' / 'gsa' / 'total' / 'S.csv', index_col=[0, 1])
    assert list(S_csv.index.names) == ['l.0', 'l.1'] and list(S_csv.columns) == ['0', '1', '2', '3'] and S_csv.shape == (4, 4)
    assert (repo.folder / 'gpr.v.a' / 'gsa' / 'closed' / 'S.csv').exists()

    gp = MOGP('gpr.v.a', fold, True, False, False)
    alpha = np.stack([o.k_inv_y(X, Y[:, l], ell[l], var[l], noise[l]) for l in range(2)])
    ref = o.ClosedSobolOracle(X, alpha[:, None, :], var[None, :], ell)
    for kind, okind in ((GSA.Kind.FIRST_ORDER, o.FIRST_ORDER), (GSA.Kind.CLOSED, o.CLOSED), (GSA.Kind.TOTAL, o.TOTAL)):
        sobol = Sobol(gp, kind)
        sobol.calibrate()
        expect = o.gsa_calibrate(ref, okind, 3)
        np.testing.assert_allclose(sobol.results['S'], expect['S'], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(sobol.results['V'], expect['V'], rtol=1e-6, atol=1e-9 * np.max(np.abs(expect['V'])))
    stored = pd.read_csv(fold.folder / 'gpr.v.a' / 'gsa' / 'first_order' / 'S.csv', index_col=[0, 1]).values
    np.testing.assert_allclose(stored, np.reshape(o.gsa_calibrate(ref, o.FIRST_ORDER, 3)['S'], (4, 4)), atol=6e-7)     # '%.6f'
    first = Sobol(gp, GSA.Kind.FIRST_ORDER, m=1)
    first.calibrate()
    assert first.results['S'].shape == (2, 2, 2) and first.folder.name == 'first_order.1'
    # ---- standard errors (ClosedSobolWithError): partial and full T, all kinds, against the reduced-form oracle
    from oracle.sobol_error_oracle import ClosedSobolWithErrorOracle
    Kc = np.stack([o.k_cho(X, Y[:, l], ell[l], var[l], noise[l]) if False else o.k_cho(X, ell[l], var[l], noise[l]) for l in range(2)])
    device_passes = []
    plain_error_terms = gpu.RcGP.sobol_error_terms
    monkeypatch.setattr(gpu.RcGP, 'sobol_error_terms', lambda self, *args, **kw: device_passes.append(1) or plain_error_terms(self, *args, **kw))
    for partial in (True, False):
        err = ClosedSobolWithErrorOracle(X, alpha[:, None, :], var[None, :], ell, Kc, is_T_partial=partial)
        for kind, okind in ((GSA.Kind.FIRST_ORDER, o.FIRST_ORDER), (GSA.Kind.CLOSED, o.CLOSED), (GSA.Kind.TOTAL, o.TOTAL)):
            sobol = Sobol(gp, kind, is_error_calculated=True, is_T_partial=partial)
            sobol.calibrate()
            per_slice = [err.marginalize(sl) for sl in o.gsa_slices(okind, 3)]
            W = np.stack([r['W'] for r in per_slice], axis=-1)
            T = np.stack([r['T'] for r in per_slice], axis=-1)
            # W = mu_phi_mu - mu_psi_mu cancels heavily for irrelevant inputs: absolute tolerance relative to the largest entry
            np.testing.assert_allclose(sobol.results['W'], W, rtol=1e-5, atol=1e-6 * np.max(np.abs(W)))
            got_T = sobol.results['T']
            W_scale = max(np.max(np.abs(W)), np.max(np.abs(err.W)), np.max(np.abs(getattr(err, 'W_mixed', 0.0))))
            if not partial:                                              # gsa/models.py:211-213: the full-model T appended, and added for TOTAL
                assert_T_matches(got_T[..., -1:], err.T[..., None], err.V[4], W_scale, f'{kind.name} full')
                got_T = got_T[..., :-1] - (got_T[..., -1:] if okind == o.TOTAL else 0.0)
            assert_T_matches(got_T, T, err.V[4], W_scale, f'{kind.name} partial={partial}')
            assert sobol.results['T'].shape == (2, 2, 3 if partial else 4)
            assert (sobol.folder / 'T.csv').exists() and (sobol.folder / 'W.csv').exists()
    # one device pass per output pair serves every slice of every kind, partial or not: the six calibrators above share it through the gp
    # (the reference recomputes all of it per kind: gsa/models.py:205, user/run.py:141-147)
    assert len(device_passes) == 2 * 2
    monkeypatch.undo()
    gp.close()
    names = run.gsa('gpr', repo, is_covariant=False, is_isotropic=False, kinds=GSA.Kind.CLOSED, is_error_calculated=True, is_T_partial=False)
    assert [str(n) for n in names] == ['gpr.v.a/gsa/closed']
    collected = pd.read_csv(repo.folder / 'gpr.v.a' / 'gsa' / 'closed' / 'T.csv')
    assert list(collected.columns[:2]) == ['N', 'fold'] and collected.shape[0] == 2 * 4


def test_predict_gradient(gpu, tmp_path):
    """HipGP.predict_gradient (gpr/models.py:386-415): shapes (o, L, M) and (o, o, L, M, M), values against the oracle, and the
    mean against a finite difference of predict (the gradient GP's mean is the gradient of the posterior mean)."""
    from romcomma_amd.data.storage import Fold
    from romcomma_amd.gpr.models import MOGP
    repo = make_repo(tmp_path / 'repo', N=150, M=3, L=2).into_K_folds(-2, seed=1)
    fold = Fold(repo, 0)
    gp = MOGP('gpr.v.a', fold, False, False, False)
    gp.kernel.data.replace(lengthscales=np.array([[0.9, 1.4, 2.2], [1.1, 0.8, 1.9]]), variance=np.array([[1.2, 0.7]]))
    gp.likelihood.data.replace(variance=np.array([[0.02, 0.01]]))
    gp.kernel._implementation = None
    gp._implementation = None
    x = fold.test_x.values[:7]
    mean, var = gp.predict_gradient(x)
    assert mean.shape == (7, 2, 3) and var.shape == (7, 7, 2, 3, 3)
    for l in range(2):
        ell, v, nse = gp.kernel.data.frames.lengthscales.np[l], gp.kernel.data.frames.variance.np[0, l], gp.likelihood.data.frames.variance.np[0, l]
        m_ref, c_ref = o.predict_gradient(gp.X, gp.Y[:, l], ell, v, nse, x)
        np.testing.assert_allclose(mean[:, l, :], m_ref, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(var[:, :, l, :, :], c_ref, rtol=1e-7, atol=1e-9 * np.max(np.abs(c_ref)))
    h = 1e-5
    for m in range(3):
        xp, xm = x.copy(), x.copy()
        xp[:, m] += h
        xm[:, m] -= h
        fd = (gp.predict(xp, False)[0] - gp.predict(xm, False)[0]) / (2 * h)
        np.testing.assert_allclose(mean[:, :, m], fd, rtol=1e-5, atol=1e-7)
    # more derivative rows than one pass holds (o M > 4096: the reference has no such limit): the rows go through 4096 at a time,
    # and a block of points straddling the chunk boundary must come out exactly as when it is asked for on its own
    rng = np.random.default_rng(2)
    big = rng.normal(size=(1500, 3))                                             # 4500 rows: two chunks, the last one ragged
    handle = gp._select(1)
    m_big, c_big = handle.predict_gradient(big)
    assert m_big.shape == (1500, 3) and c_big.shape == (1500, 3, 1500, 3)
    idx = np.arange(1358, 1372)                                                  # rows 4074 .. 4115
    m_sub, c_sub = handle.predict_gradient(big[idx])
    np.testing.assert_allclose(m_big[idx], m_sub, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(c_big[np.ix_(idx, range(3), idx, range(3))], c_sub, rtol=1e-12, atol=1e-14 * np.max(np.abs(c_sub)))
    m_ref, c_ref = o.predict_gradient(gp.X, gp.Y[:, 1], gp.kernel.data.frames.lengthscales.np[1], gp.kernel.data.frames.variance.np[0, 1],
                                      gp.likelihood.data.frames.variance.np[0, 1], big[idx])
    np.testing.assert_allclose(m_sub, m_ref, rtol=1e-8, atol=1e-10)
    gp.close()


def test_predict_df(gpu, tmp_path):
    """GPR.predict_df (gpr/models.py:202-222): columns (X heading, X.m) | ('Mean', Y.l) | ('SD', Y.l); with is_normalized=False the inputs
    and the mean return to the original units (Normalization.undo_from) and the SD is rescaled without a shift (unscale_Y)."""
    from romcomma_amd.data.storage import Fold
    from romcomma_amd.gpr.models import MOGP
    repo = make_repo(tmp_path / 'repo', N=160, M=3, L=2, seed=3).into_K_folds(-2, seed=4)
    fold = Fold(repo, 1)
    gp = MOGP('gpr.v.a', fold, False, False, False)
    gp.kernel.data.replace(lengthscales=np.array([[1.0, 1.5, 2.5], [1.2, 0.9, 2.0]]), variance=np.array([[1.1, 0.8]]))
    gp.likelihood.data.replace(variance=np.array([[0.03, 0.02]]))
    gp.kernel._implementation = None
    gp._implementation = None
    x = fold.test_x.values[:9]
    mean, sd = gp.predict(x)
    for y_instead_of_f in (True, False):
        df = gp.predict_df(x, y_instead_of_f)
        assert list(df.columns) == [('X', 'X.0'), ('X', 'X.1'), ('X', 'X.2'), ('Mean', 'Y.0'), ('Mean', 'Y.1'), ('SD', 'Y.0'), ('SD', 'Y.1')]
        want_mean, want_sd = gp.predict(x, y_instead_of_f)
        np.testing.assert_array_equal(df['X'].values, x)
        np.testing.assert_array_equal(df['Mean'].values, want_mean)
        np.testing.assert_array_equal(df['SD'].values, want_sd)
    raw = gp.predict_df(x, is_normalized=False)
    stats = fold.normalization.frame.df                                           # rows mean, std, rng, min, max (data/storage.py:547-558)
    import scipy.stats
    X_back = scipy.stats.norm.cdf(x) * stats.loc['rng'].values[:3] + stats.loc['min'].values[:3]
    np.testing.assert_allclose(raw['X'].values, X_back, rtol=1e-12)
    np.testing.assert_allclose(raw['X'].values, fold.normalization.undo_from(fold.test_data.df.iloc[:9])['X'].values, rtol=1e-9)
    np.testing.assert_allclose(raw['Mean'].values, mean * stats.loc['std'].values[3:] + stats.loc['mean'].values[3:], rtol=1e-12)
    np.testing.assert_allclose(raw['SD'].values, sd * stats.loc['std'].values[3:], rtol=1e-12)
    assert list(raw.columns) == list(df.columns)
    gp.close()


def test_covariant_gp_end_to_end(gpu, tmp_path):
    """run.gpr with is_covariant=None: the independent GPs, then the covariant GP warm-started from them (user/run.py:69-84) with
    the reference's default trainables (kernel Cholesky diagonal + likelihood Cholesky factor); stored parameters, LML, predict
    and the Sobol indices (diagonal F, then the full F) against oracle/mogp_oracle.py at the fitted hyper-parameters."""
    from oracle import mogp_oracle as mo
    from romcomma_amd.data.storage import Fold
    from romcomma_amd.gpr.models import MOGP
    from romcomma_amd.gsa.models import GSA, Sobol
    from romcomma_amd.user import run
    repo = make_repo(tmp_path / 'repo', N=200, M=3, L=2, seed=4).into_K_folds(-2, seed=5)
    names = run.gpr('gpr', repo, is_read=False, is_covariant=None, is_isotropic=False)
    assert names == ['gpr.v.a', 'gpr.c.a']
    fold = Fold(repo, 0)
    v = MOGP('gpr.v.a', fold, True, False, False)
    lml_independent = float(np.sum(v.likelihood.data.frames.log_marginal.np))
    ell_v = v.kernel.data.frames.lengthscales.np.copy()
    v.close()
    # as written by calibrate; constructing the model below re-broadcasts the likelihood variance, which keeps only its diagonal
    # and rewrites the file (base/classes.py:72-89 through gpr/models.py:284)
    Sigma_fitted = pd.read_csv(fold.folder / 'gpr.c.a' / 'likelihood' / 'variance.csv', index_col=0).values
    gp = MOGP('gpr.c.a', fold, True, True, False)
    ell = gp.kernel.data.frames.lengthscales.np
    F = gp.kernel.data.frames.variance.np
    Sigma_stored = gp.likelihood.data.frames.variance.np
    assert ell.shape == (2, 3) and F.shape == (2, 2) and Sigma_stored.shape == (2, 2)
    np.testing.assert_array_equal(ell, ell_v)                                  # lengthscales are not trained by default
    assert F[0, 1] == 0.0 and F[1, 0] == 0.0                                   # nor is the kernel covariance
    meta = json.loads((fold.folder / 'gpr.c.a' / 'meta.json').read_text())
    assert meta['kernel']['covariance'] is False and meta['likelihood']['covariance'] is True
    stored_lml = float(gp.likelihood.data.frames.log_marginal.np[0, 0])
    assert Sigma_fitted[0, 1] != 0.0 and Sigma_stored[0, 1] == 0.0
    np.testing.assert_allclose(np.diag(Sigma_stored), np.diag(Sigma_fitted), rtol=1e-12)
    # the stored LML is the value at the full fitted Sigma
    assert stored_lml == pytest.approx(mo.lml(gp.X, gp.Y, ell, F, (Sigma_fitted + Sigma_fitted.T) / 2), rel=1e-8)
    assert stored_lml >= lml_independent - 1e-6 * abs(lml_independent)
    Sigma = np.diag(np.diag(Sigma_stored))
    assert gp.log_marginal_likelihood()[0] == pytest.approx(mo.lml(gp.X, gp.Y, ell, F, Sigma), rel=1e-9)
    x = fold.test_x.values
    mean, sd = gp.predict(x)
    mr, sr = mo.predict(gp.X, gp.Y, ell, F, Sigma, x)
    np.testing.assert_allclose(mean, mr, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(sd, sr, rtol=1e-7)
    assert np.max(gp.check_K_inv_Y(x[:10])) < 1e-8
    assert gp.K_inv_Y.shape == (2, 1, gp.N)
    np.testing.assert_allclose(gp.K_inv_Y, mo.k_inv_y(gp.X, gp.Y, ell, F, Sigma), rtol=1e-6, atol=1e-8)
    assert (fold.folder / 'gpr.c.a' / 'test_summary.csv').exists()
    gmean, gvar = gp.predict_gradient(x[:4])
    rmean, rvar = mo.predict_gradient(gp.X, gp.Y, ell, F, Sigma, x[:4])
    assert gmean.shape == (4, 2, 3) and gvar.shape == (4, 2, 4, 2, 3, 3)
    np.testing.assert_allclose(gmean, rmean, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(gvar, rvar, rtol=1e-6, atol=1e-9 * np.max(np.abs(rvar)))
    # Sobol on the covariant GP: F diagonal by default (kernel covariance untrained), the full F on request
    KiY = mo.k_inv_y(gp.X, gp.Y, ell, F, Sigma)
    sobol = Sobol(gp, GSA.Kind.CLOSED)
    sobol.calibrate()
    alpha = KiY.reshape(2, -1)
    ref = o.ClosedSobolOracle(gp.X, alpha[:, None, :], np.diag(F)[None, :], ell)
    expect = o.gsa_calibrate(ref, o.CLOSED, 3)
    np.testing.assert_allclose(sobol.results['S'], expect['S'], rtol=1e-6, atol=1e-9)
    from romcomma_amd.gsa.calibrators import ClosedSobol
    full = ClosedSobol(gp, is_F_diagonal=False)
    slices = [(0, 3), (0, 1), (1, 2), (2, 3), (1, 3), (0, 2)]
    lit = mo.sobol_V_covariant(gp.X, KiY, F, ell, slices)
    got = full.marginalize_all(slices)['V']
    np.testing.assert_allclose(np.moveaxis(got, -1, 0), lit, rtol=1e-7, atol=1e-10 * np.max(np.abs(lit)))
    # standard errors on the covariant GP (F diagonal): psi_factor solves with the (LN) Cholesky factor, the vector embedded in its
    # output block (gsa/calibrators.py:304-308) -- against the literal transliteration
    from oracle.sobol_error_oracle import LiteralClosedSobolWithError
    Kc = mo.k_cho(gp.X, ell, F, Sigma)
    for partial in (True, False):
        err = LiteralClosedSobolWithError(gp.X, KiY, np.diag(F)[None, :], ell, Kc, is_T_partial=partial)
        sobol = Sobol(gp, GSA.Kind.FIRST_ORDER, is_error_calculated=True, is_T_partial=partial)
        sobol.calibrate()
        per_slice = [err.marginalize(sl) for sl in o.gsa_slices(o.FIRST_ORDER, 3)]
        W = np.stack([r['W'] for r in per_slice], axis=-1)
        T = np.stack([r['T'] for r in per_slice], axis=-1)
        np.testing.assert_allclose(sobol.results['W'], W, rtol=1e-5, atol=1e-6 * np.max(np.abs(W)))
        got_T = sobol.results['T']
        W_full = err.W if partial else err.W.DIAGONAL
        W_scale = max(np.max(np.abs(W)), np.max(np.abs(np.asarray(W_full))), 0.0 if partial else np.max(np.abs(np.asarray(err.W.MIXED))))
        V4 = np.asarray(err.V[4])
        if not partial:
            assert_T_matches(got_T[..., -1:], np.asarray(err.T)[..., None], V4, W_scale, 'covariant full')
            got_T = got_T[..., :-1]
        assert_T_matches(got_T, T, V4, W_scale, f'covariant partial={partial}')
    gp.close()
    gsa_names = run.gsa('gpr', repo, is_covariant=True, is_isotropic=False, kinds=GSA.Kind.FIRST_ORDER)
    assert [str(n) for n in gsa_names] == ['gpr.c.a/gsa/first_order']
    assert (repo.folder / 'gpr.c.a' / 'gsa' / 'first_order' / 'S.csv').exists()


def test_covariant_gp_with_trained_kernel_covariance_and_lengthscales(gpu, tmp_path):
    """Options as the reference takes them (kernel={'covariance': True, 'lengthscales': {'covariant': True}}): every parameter group of
    the covariant GP trains, meta.json records it, and GSA then takes the non-diagonal-F branch on its own (gsa/calibrators.py:129-138).
    Isotropic covariant model: (L,1) lengthscales."""
    from oracle import mogp_oracle as mo
    from romcomma_amd.data.storage import Fold
    from romcomma_amd.gpr.models import MOGP
    from romcomma_amd.gsa.calibrators import ClosedSobol
    from romcomma_amd.gsa.models import GSA, Sobol
    from romcomma_amd.user import run
    repo = make_repo(tmp_path / 'repo', N=160, M=3, L=2, seed=8).into_K_folds(-2, seed=9)
    fold = Fold(repo, 0)
    run.gpr('gpr', fold, is_read=False, is_covariant=False, is_isotropic=False)
    names = run.gpr('gpr', fold, is_read=None, is_covariant=True, is_isotropic=False,
                    kernel={'covariance': True, 'lengthscales': {'covariant': True}})
    assert names == ['gpr.c.a']
    meta = json.loads((fold.folder / 'gpr.c.a' / 'meta.json').read_text())
    assert meta['kernel']['covariance'] is True and meta['kernel']['lengthscales']['covariant'] is True
    v = MOGP('gpr.v.a', fold, True, False, False)
    ell_v = v.kernel.data.frames.lengthscales.np.copy()
    lml_v = float(np.sum(v.likelihood.data.frames.log_marginal.np))
    v.close()
    # The object the reference keeps using after calibrate() is the trained MOGPR with its FULL likelihood covariance
    # (gpr/models.py:359-367; the diagonal reduction happens on construction only, gpf/models.py:121-123): test.csv written by run.gpr
    # straight after the fit and any in-session predict / K_inv_Y use the full fitted Sigma; a re-read model starts from diag(Sigma).
    live = MOGP('live.c.a', fold, False, True, False)
    live.calibrate()
    Sigma_full = np.array(live.likelihood.data.frames.variance.np, dtype=float)
    assert abs(Sigma_full[0, 1]) > 1e-9                                        # the likelihood covariance did train
    ell_l, F_l = np.array(live.kernel.data.frames.lengthscales.np, dtype=float), np.array(live.kernel.data.frames.variance.np, dtype=float)
    xs = fold.test_x.values[:11]
    mean_live, sd_live = live.predict(xs)
    mean_ref, sd_ref = mo.predict(live.X, live.Y, ell_l, (F_l + F_l.T) / 2, (Sigma_full + Sigma_full.T) / 2, xs)
    np.testing.assert_allclose(mean_live, mean_ref, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(sd_live, sd_ref, rtol=1e-7)
    stored_lml = float(live.likelihood.data.frames.log_marginal.np[0, 0])
    assert live.log_marginal_likelihood()[0] == pytest.approx(stored_lml, rel=1e-12)      # the model queried is the model whose LML is stored
    np.testing.assert_allclose(live.K_inv_Y.reshape(-1), mo.k_inv_y(live.X, live.Y, ell_l, (F_l + F_l.T) / 2, Sigma_full).reshape(-1), rtol=1e-6, atol=1e-9)
    mean_diag, _ = mo.predict(live.X, live.Y, ell_l, (F_l + F_l.T) / 2, np.diag(np.diag(Sigma_full)), xs)
    assert np.max(np.abs(mean_diag - mean_ref)) > 1e-9                         # (the two models do differ)
    live.close()
    reread = MOGP('live.c.a', fold, True, True, False)                         # construction diagonalises (and rewrites the csv)
    np.testing.assert_allclose(reread.predict(xs)[0], mean_diag, rtol=1e-7, atol=1e-9)
    reread.close()
    gp = MOGP('gpr.c.a', fold, True, True, False)
    ell, F = gp.kernel.data.frames.lengthscales.np, gp.kernel.data.frames.variance.np
    assert F[0, 1] != 0.0 and F[0, 1] == pytest.approx(F[1, 0], rel=1e-12) and not np.array_equal(ell, ell_v)
    assert float(gp.likelihood.data.frames.log_marginal.np[0, 0]) >= lml_v - 1e-6 * abs(lml_v)
    Sigma = np.diag(np.diag(gp.likelihood.data.frames.variance.np))
    Fs = (F + F.T) / 2
    assert gp.log_marginal_likelihood()[0] == pytest.approx(mo.lml(gp.X, gp.Y, ell, Fs, Sigma), rel=1e-9)
    cal = ClosedSobol(gp)                                                      # is_F_diagonal from meta.json: False
    assert cal.is_F_diagonal is False
    KiY = mo.k_inv_y(gp.X, gp.Y, ell, Fs, Sigma)
    slices = [(0, 3), (0, 1), (1, 2), (2, 3)]
    ref = mo.sobol_V_covariant(gp.X, KiY, Fs, ell, slices)
    np.testing.assert_allclose(np.moveaxis(cal.marginalize_all(slices)['V'], -1, 0), ref, rtol=1e-6, atol=1e-9 * np.max(np.abs(ref)))
    sobol = Sobol(gp, GSA.Kind.TOTAL)
    sobol.calibrate()
    assert sobol.results['S'].shape == (2, 2, 4)
    with pytest.raises(NotImplementedError):                                   # errors need a diagonal F (gsa/calibrators.py:380-381)
        Sobol(gp, GSA.Kind.TOTAL, is_error_calculated=True).calibrate()
    gp.close()
    # isotropic covariant model from scratch
    names = run.gpr('iso', fold, is_read=False, is_covariant=True, is_isotropic=True, kernel={'lengthscales': {'covariant': True}})
    assert names == ['iso.c.i']
    iso = MOGP('iso.c.i', fold, True, True, True)
    assert iso.kernel.data.frames.lengthscales.np.shape == (2, 1) and iso.kernel.data.frames.variance.np.shape == (2, 2)
    ell_i = np.broadcast_to(iso.kernel.data.frames.lengthscales.np, (2, 3))
    Sigma_i = np.diag(np.diag(iso.likelihood.data.frames.variance.np))
    assert iso.log_marginal_likelihood()[0] == pytest.approx(mo.lml(iso.X, iso.Y, ell_i, iso.kernel.data.frames.variance.np, Sigma_i), rel=1e-9)
    iso.close()
