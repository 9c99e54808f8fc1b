"""GPU known-answer tests: the reference's own benchmark functions (user/functions.py:126-152: SALib's Ishigami 'standard' and the modified
Sobol G 'weak5_2') have published analytic Sobol indices; fit + closed-form Sobol on the GPU must reproduce first-order, closed and total
indices (the three kinds of gsa/models.py:77-90) to GP-approximation accuracy, through the C ABI and through the host classes."""
import numpy as np
import pandas as pd
import pytest

import known_functions as kf

pytestmark = pytest.mark.gpu

CASES = {'ishigami': (kf.ishigami, kf.ishigami_variances(), 3), 'ishigami+idle input': (kf.ishigami, kf.ishigami_variances(), 4),
         'sobol_g': (kf.sobol_g, kf.sobol_g_variances(), 5)}


@pytest.mark.parametrize('name', list(CASES))
def test_fit_and_sobol_reproduce_the_published_indices(gpu, name):
    from oracle import gp_oracle as o
    from romcomma_amd import _lib
    from romcomma_amd.gpr.optimize import fit_lbfgsb
    fn, partial, M = CASES[name]
    N = 2048
    X, y = kf.sample(fn, N, M, seed=7)
    want = kf.analytic_indices(partial, M)
    with _lib.RcGP(X, y) as gp:
        # the reference's default flow (user/run.py:75-88): the isotropic model first, the ARD model warm-started from it. (Straight from the
        # defaults the ARD fit of Ishigami at this N stops in a poor local optimum, lengthscales (0.77, 4.2, 2.3) -- on the oracle too, to 9 digits.)
        iso = fit_lbfgsb(gp, 5.0, 2.0, 0.02, is_isotropic=True)
        fit = fit_lbfgsb(gp, np.broadcast_to(iso['lengthscales'], (M,)), iso['variance'], iso['noise'])
        V = gp.sobol_closed(o.all_slices(M))
    S = V / V[-1]
    first, closed, total = S[:M], S[M:2 * M], 1.0 - S[2 * M:3 * M]           # gsa/models.py:207-214: total = S_full - S(complement)
    np.testing.assert_allclose(first, want['first_order'], atol=0.01, err_msg=f'{name}: first order, lengthscales {fit["lengthscales"]}')
    np.testing.assert_allclose(closed, want['closed'], atol=0.01, err_msg=f'{name}: closed')
    np.testing.assert_allclose(total, want['total'], atol=0.01, err_msg=f'{name}: total')


def test_host_classes_reproduce_the_published_indices(gpu, tmp_path):
    """The same through the drop-in surface: Repository -> run.gpr -> run.gsa -> S.csv (Ishigami, one fold's training share of 1536 rows)."""
    from romcomma_amd.data.storage import Fold, Repository
    from romcomma_amd.user import run
    M, N = 3, 2048
    rng = np.random.default_rng(3)
    u = (np.stack([rng.permutation(N) for _ in range(M)], axis=1) + rng.random((N, M))) / N
    columns = pd.MultiIndex.from_tuples([('X', f'X.{m}') for m in range(M)] + [('Y', 'Y.0')])
    repo = Repository.from_df(tmp_path / 'repo', pd.DataFrame(np.concatenate([u, kf.ishigami(u)[:, None]], axis=1), columns=columns))
    repo = repo.into_K_folds(4, seed=1)
    run.gpr('gpr', repo, is_read=False, is_covariant=False, is_isotropic=None)      # isotropic, then ARD warm-started from it
    run.gsa('gpr', repo, is_covariant=False, is_isotropic=False)
    want = kf.analytic_indices(kf.ishigami_variances(), M)
    fold = Fold(repo, 0)
    for kind in ('first_order', 'closed', 'total'):
        S = pd.read_csv(fold.folder / 'gpr.v.a' / 'gsa' / kind / 'S.csv', index_col=[0, 1]).values[0, :M]
        np.testing.assert_allclose(S, want[kind], atol=0.012, err_msg=kind)


def test_standard_errors_are_of_the_size_of_the_actual_errors(gpu, tmp_path):
    """ClosedSobolWithError (gsa/calibrators.py:146-402; SURVEY 8f rank 2 calls the oracle's transliteration of it 'unverified against real
    TF'): on Ishigami the ACTUAL error of every index is known -- the published analytic value is at hand -- so the standard error T the path
    reports can be held against it. Measured: |S - S_analytic| = 0.0004 ... 0.0026 where T = 0.0015 ... 0.0029 (partial), i.e. within 2 T. A
    plausibility check of what T means, not a pin of its arithmetic."""
    from romcomma_amd.data.storage import Fold, Repository
    from romcomma_amd.gpr.models import MOGP
    from romcomma_amd.gsa.models import GSA, Sobol
    from romcomma_amd.user import run
    M, N = 3, 2048
    rng = np.random.default_rng(3)
    u = (np.stack([rng.permutation(N) for _ in range(M)], axis=1) + rng.random((N, M))) / N
    columns = pd.MultiIndex.from_tuples([('X', f'X.{m}') for m in range(M)] + [('Y', 'Y.0')])
    repo = Repository.from_df(tmp_path / 'repo', pd.DataFrame(np.concatenate([u, kf.ishigami(u)[:, None]], axis=1), columns=columns))
    repo = repo.into_K_folds(4, seed=1)
    run.gpr('gpr', repo, is_read=False, is_covariant=False, is_isotropic=None)
    want = kf.analytic_indices(kf.ishigami_variances(), M)
    gp = MOGP('gpr.v.a', Fold(repo, 0), True, False, False)
    try:
        for kind, name in ((GSA.Kind.FIRST_ORDER, 'first_order'), (GSA.Kind.CLOSED, 'closed'), (GSA.Kind.TOTAL, 'total')):
            sobol = Sobol(gp, kind, is_error_calculated=True, is_T_partial=True)
            sobol.calibrate()
            S, T = sobol.results['S'][0, 0, :M], sobol.results['T'][0, 0, :M]
            assert np.all(T >= 0.0) and np.all(T < 0.01), (name, T)
            assert np.all(np.abs(S - want[name]) <= 3.0 * T + 5e-4), (name, S - want[name], T)
    finally:
        gp.close()


def test_cross_output_indices_against_their_analytic_values(gpu, tmp_path):
    """The off-diagonal entries S_lj(S) = Cov(E[f_l | x_S], E[f_j | x_S]) / sqrt(Var f_l Var f_j) that the reference fills for independent GPs
    too (gsa/calibrators.py:79; rows l.0 != l.1 of S.csv) -- for the reference's Ishigami 'standard' (A = 7, B = 0.1) and 'sin' (A = B = 0:
    f = sin x1) outputs (user/functions.py:144-146) they are known in closed form: E[f_sin | x_S] = sin x1 if input 0 is in S, else 0, so
    S_01(S) = (1 + B pi^4 / 5) / 2 / sqrt(Var f_standard / 2) = 0.5603 for every S that contains input 0 and 0 otherwise."""
    from romcomma_amd.data.storage import Fold, Repository
    from romcomma_amd.user import run
    M, N = 3, 2048
    rng = np.random.default_rng(5)
    u = (np.stack([rng.permutation(N) for _ in range(M)], axis=1) + rng.random((N, M))) / N
    Y = np.stack([kf.ishigami(u), kf.ishigami(u, A=0.0, B=0.0)], axis=1)
    Y = (Y - Y.mean(axis=0)) / Y.std(axis=0) + 0.02 * rng.standard_normal(Y.shape)     # a little noise keeps the fit of the one-input output well conditioned
    columns = pd.MultiIndex.from_tuples([('X', f'X.{m}') for m in range(M)] + [('Y', 'Y.0'), ('Y', 'Y.1')])
    repo = Repository.from_df(tmp_path / 'repo', pd.DataFrame(np.concatenate([u, Y], axis=1), columns=columns)).into_K_folds(4, seed=2)
    run.gpr('gpr', repo, is_read=False, is_covariant=False, is_isotropic=None)
    run.gsa('gpr', repo, is_covariant=False, is_isotropic=False)
    B = 0.1
    cross = 0.5 * (1.0 + B * np.pi ** 4 / 5.0) / np.sqrt(sum(kf.ishigami_variances().values()) * 0.5)
    assert cross == pytest.approx(0.5603, abs=1e-4)
    want = {'first_order': np.array([cross, 0.0, 0.0]),                      # S = {m}
            'closed': np.array([cross, cross, cross]),                        # S = {0..m}
            'total': cross - np.array([0.0, 0.0, 0.0])}                       # S_full - S({m+1..}): the complements never contain input 0
    fold = Fold(repo, 0)
    for kind in ('first_order', 'closed', 'total'):
        S = pd.read_csv(fold.folder / 'gpr.v.a' / 'gsa' / kind / 'S.csv', index_col=[0, 1])
        assert [tuple(i) for i in S.index] == [(0, 0), (0, 1), (1, 0), (1, 1)]
        np.testing.assert_allclose(S.values[1, :M], want[kind], atol=0.01, err_msg=f'{kind} (0, 1)')
        np.testing.assert_allclose(S.values[2, :M], want[kind], atol=0.01, err_msg=f'{kind} (1, 0)')
        np.testing.assert_allclose(S.values[3, :M], {'first_order': [1.0, 0.0, 0.0], 'closed': [1.0, 1.0, 1.0], 'total': [1.0, 1.0, 1.0]}[kind],
                                   atol=0.01, err_msg=f'{kind} (1, 1): f = sin x1')
