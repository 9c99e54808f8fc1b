"""CPU tests of the host-side mirror of the reference interface (no GPU, no compute calls into librcgp)."""
import json
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

from romcomma_amd.base.classes import Data, Frame, Model
from romcomma_amd.data.storage import Fold, Normalization, Repository
from romcomma_amd.gpr import optimize
from romcomma_amd.gpr.kernels import RBF, Kernel
from romcomma_amd.user import results
from romcomma_amd.user.sample import synthetic_fold


def make_repo(folder: Path, N=60, M=3, L=2, seed=0) -> Repository:
    rng = np.random.default_rng(seed)
    U = rng.random((N, M)) * 4 - 1
    Y = np.stack([np.sin(U[:, 0]) + 0.3 * U[:, 1] + l * U[:, 2] ** 2 + 0.01 * rng.standard_normal(N) for l in range(L)], axis=1)
    columns = pd.MultiIndex.from_tuples([('X', f'X.{m}') for m in range(M)] + [('Y', f'Y.{l}') for l in range(L)])
    return Repository.from_df(folder, pd.DataFrame(np.concatenate([U, Y], axis=1), columns=columns))


def test_frame_data_model_roundtrip(tmp_path):
    frame = Frame(tmp_path / 'variance', np.array([[1.0, 2.0]]))
    assert (tmp_path / 'variance.csv').exists()                        # suffix appended (base/classes.py:69)
    assert Frame(tmp_path / 'variance').np.tolist() == [[1.0, 2.0]]
    frame.np = np.array([[3.0, 4.0]])                                  # assignment writes through
    assert Frame(tmp_path / 'variance').np.tolist() == [[3.0, 4.0]]
    frame.broadcast_value((2, 2))                                      # (1,L) -> (L,L): diagonal
    assert Frame(tmp_path / 'variance').np.tolist() == [[3.0, 0.0], [0.0, 4.0]]
    with pytest.raises(IndexError):
        frame.broadcast_value((1, 3))
    data = Kernel.Data(tmp_path / 'kernel')
    assert Kernel.Data.fields == ('variance', 'lengthscales')
    assert data.frames.variance.np[0, 0] == 2.0 and data.frames.lengthscales.np[0, 0] == 5.0     # reference defaults
    data.replace(lengthscales=np.full((2, 3), 1.5))
    assert Kernel.Data.read(tmp_path / 'kernel').frames.lengthscales.np.shape == (2, 3)
    copy = Data.copy(tmp_path / 'kernel', tmp_path / 'kernel2')
    assert sorted(p.name for p in copy.iterdir()) == ['lengthscales.csv', 'variance.csv']


def test_kernel_store(tmp_path):
    kernel = RBF(tmp_path / 'k')
    assert RBF.TYPE_IDENTIFIER == 'kernels.RBF'
    assert Kernel.TypeFromIdentifier('kernels.RBF') is RBF
    assert Kernel.TypeFromParameters(Kernel.Data(tmp_path / 'p')) is RBF
    kernel.broadcast_parameters((1, 3), 4)
    assert (kernel.L, kernel.M) == (3, 4) and not kernel.is_covariant
    records = kernel.implementation
    assert len(records) == 3 and records[0]['lengthscales'].shape == (4,) and records[0]['variance'] == 2.0
    with pytest.raises(IndexError):
        kernel.broadcast_parameters((1, 2), 4)                         # shrinking is an error
    assert kernel.calibrate(variance=False)['variance'] is False
    with pytest.raises(TypeError):
        Kernel.TypeFromIdentifier('kernels.Matern')


def test_repository_folds_and_normalization(tmp_path):
    repo = make_repo(tmp_path / 'repo', N=61)
    assert (repo.N, repo.M, repo.L) == (61, 3, 2)
    meta = json.loads((tmp_path / 'repo' / 'meta.json').read_text())
    assert meta['data'] == {'X_heading': 'X', 'Y_heading': 'Y', 'N': 61, 'M': 3, 'L': 2}
    repo.into_K_folds(4, seed=1)
    assert list(repo.folds) == [0, 1, 2, 3, 4]                        # positive K adds the improper fold K
    sizes = [(Fold(repo, k).N, Fold(repo, k).test_data.df.shape[0]) for k in range(4)]
    assert all(n + t == 61 for n, t in sizes) and sorted(t for _, t in sizes) == [15, 15, 15, 16]
    test_rows = np.concatenate([Fold(repo, k).test_data.df.index.values for k in range(4)])
    assert sorted(test_rows) == list(range(61))                       # every row is held out exactly once
    improper = Fold(repo, 4)
    assert improper.N == 61 and improper.test_data.df.shape[0] == 61
    fold = Fold(repo, 0)
    assert fold.meta['k'] == 0
    stats = fold.normalization.frame.df
    assert list(stats.index) == ['mean', 'std', 'rng', 'min', 'max']
    raw = repo.data.df
    np.testing.assert_allclose(stats.loc['std'].values, raw.std().values)                     # pandas ddof = 1
    np.testing.assert_allclose(stats.loc['rng'].values, 2 * np.sqrt(3) * raw.std().values)
    # Y z-scored with the repository statistics; X probit of the clipped uniform map; undo_from inverts apply_to
    rows = fold.data.df.index.values
    np.testing.assert_allclose(fold.Y.values, ((raw.iloc[rows, 3:] - raw.iloc[:, 3:].mean()) / raw.iloc[:, 3:].std()).values)
    back = fold.normalization.undo_from(fold.data.df)
    inside = np.ones(back.shape, dtype=bool)                                    # inputs beyond mean +- sqrt(3) std are clipped: irreversible
    inside[:, :3] = ((raw.iloc[rows, :3] > stats.loc['min'].iloc[:3]) & (raw.iloc[rows, :3] < stats.loc['max'].iloc[:3])).values
    assert inside.mean() > 0.95
    np.testing.assert_allclose(back.values[inside], raw.iloc[rows].values[inside], rtol=1e-9, atol=1e-9)
    assert np.all(np.isfinite(fold.X.values))
    # same seed, same split; K = -4 has no improper fold
    repo2 = make_repo(tmp_path / 'repo2', N=61).into_K_folds(-4, seed=1)
    assert list(repo2.folds) == [0, 1, 2, 3]
    assert list(Fold(repo2, 2).test_data.df.index) == list(Fold(repo, 2).test_data.df.index)
    with pytest.raises(IndexError):
        repo.into_K_folds(100)
    repo.rotate_folds(None)
    np.testing.assert_allclose(Fold(repo, 1).X_rotation, np.eye(3))


def test_y_split(tmp_path):
    repo = make_repo(tmp_path / 'repo')
    repo.Y_split()
    splits = dict(repo.Y_splits)
    assert sorted(splits) == [0, 1]
    assert Repository(splits[1]).L == 1 and Repository(splits[1]).M == 3


def test_parameter_transforms():
    x = np.array([1e-3, 0.5, 5.0, 50.0])
    np.testing.assert_allclose(optimize.softplus(optimize.inv_softplus(x)), x, rtol=1e-12)
    u = np.array([-30.0, -1.0, 0.0, 2.0, 40.0])
    np.testing.assert_allclose(optimize.sigmoid(u), 1 / (1 + np.exp(-u)), rtol=1e-12)
    h = 1e-6
    np.testing.assert_allclose((optimize.softplus(u + h) - optimize.softplus(u - h)) / (2 * h), optimize.sigmoid(u), rtol=1e-6, atol=1e-12)


def test_gsa_slices_and_labels():
    from romcomma_amd.gsa.models import GSA

    class Stub(GSA):
        calibrator = None

        def _post_calibrate(self, calibrator, results):
            return results

        def __init__(self, kind, m, M):
            self.kind, self.meta = kind, {'m': m, 'M': M}
    pairs = lambda kind, m=-1: [tuple(int(v) for v in p) for p in Stub(kind, m, 3)._m_dataset]
    assert pairs(GSA.Kind.FIRST_ORDER) == [(0, 1), (1, 2), (2, 3)]
    assert pairs(GSA.Kind.CLOSED) == [(0, 1), (0, 2), (0, 3)]
    assert pairs(GSA.Kind.TOTAL) == [(1, 3), (2, 3), (3, 3)]
    assert pairs(GSA.Kind.TOTAL, 1) == [(2, 3)]
    assert Stub(GSA.Kind.CLOSED, -1, 3)._m_dataset[0].dtype == np.int32
    assert list(GSA._columns(3, 4, [0, 1, 2])) == [0, 1, 2, 3]
    assert list(GSA._columns(3, 2, [1])) == [1, 3]
    assert GSA._index([2, 2, 4]).names == ['l.0', 'l.1'] and len(GSA._index([2, 2, 4])) == 4
    assert [k.name for k in GSA.ALL_KINDS] == ['FIRST_ORDER', 'CLOSED', 'TOTAL']


def test_collect(tmp_path):
    for k in range(2):
        folder = tmp_path / f'f{k}'
        folder.mkdir()
        pd.DataFrame({'a': [k + 0.1234567, k + 1.0]}).to_csv(folder / 'S.csv')
    results.Collect({'S': {}}, {tmp_path / 'f0': {'fold': 0}, tmp_path / 'f1': {'fold': 1}}).from_folders(tmp_path / 'out', True)
    out = pd.read_csv(tmp_path / 'out' / 'S.csv')
    assert list(out['fold']) == [0, 0, 1, 1]
    assert out['a'].iloc[0] == pytest.approx(0.123457)                 # '%.6f'
    with pytest.raises(FileNotFoundError):
        results.Collect({'missing': {}}, {tmp_path / 'f0': {}}).from_folders(tmp_path / 'out2')
    results.Collect({'missing': {}}, {tmp_path / 'f0': {}}, ignore_missing=True).from_folders(tmp_path / 'out3')


def test_synthetic_generator_matches_oracle_copy():
    from oracle import gp_oracle as o
    a, b = synthetic_fold(100, 4, k=3, l=1), o.synthetic_fold(100, 4, k=3, l=1)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert abs(a[1].mean()) < 1e-12 and a[1].std() == pytest.approx(1.0)


def test_synthetic_cv_folds():
    from romcomma_amd.user.sample import synthetic_cv_fold
    folds = [synthetic_cv_fold(700, 3, k, K=8) for k in range(8)]
    assert all(X.shape == (700, 3) and y.shape == (700,) for X, y in folds)
    full_X, _ = synthetic_fold(800, 3)                                   # 700 + ceil(700/7) rows in the underlying dataset
    assert np.array_equal(folds[0][0], full_X[100:])                     # fold 0 leaves out the first block
    assert np.array_equal(folds[3][0], np.concatenate([full_X[:300], full_X[400:]]))
    assert np.array_equal(folds[7][0], full_X[:700])
    with pytest.raises(ValueError):
        synthetic_cv_fold(700, 3, 8, K=8)


def test_hipgp_stores_and_missing_gpu(tmp_path):
    from romcomma_amd import _lib
    from romcomma_amd.gpr.models import MOGP
    repo = make_repo(tmp_path / 'repo').into_K_folds(-2, seed=0)
    cov = MOGP('gpr.c.a', Fold(repo, 0), False, True, False)           # covariant: (L,L) variances, diagonal to start with
    assert cov.kernel.data.frames.variance.np.shape == (2, 2) and cov.likelihood.data.frames.variance.np.shape == (2, 2)
    assert np.array_equal(cov.kernel.data.frames.variance.np, 2.0 * np.eye(2)) and cov.likelihood.is_covariant and cov.kernel.is_covariant
    lengthscales, variance, noise = cov._hyper_mo()
    assert lengthscales.shape == (2, 3) and np.array_equal(noise, 0.02 * np.eye(2)) and len(cov.kernel.implementation) == 1
    gp = MOGP('gpr.v.a', Fold(repo, 0), False, False, False)           # building the stores needs no GPU
    layout = sorted(str(p.relative_to(gp.folder)) for p in gp.folder.rglob('*.csv'))
    assert layout == ['kernel.csv', 'kernel/lengthscales.csv', 'kernel/variance.csv', 'likelihood/log_marginal.csv', 'likelihood/variance.csv']
    assert Frame(gp.folder / 'kernel').np[0, 0] == 'kernels.RBF'
    assert gp.kernel.data.frames.lengthscales.np.shape == (2, 3) and gp.likelihood.data.frames.variance.np.shape == (1, 2)
    if _lib.device_count() <= 0:
        with pytest.raises(_lib.RcgpError):                            # compute fails loudly without a device
            gp.predict(np.zeros((2, 3)))


def test_bench_fold_schedule():
    """bench.py hands the folds of the 8-fold split round the ranks: one fold per rank and step, timed step s of rank r = fold r + s."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench', str(Path(__file__).resolve().parent.parent / 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.fold_schedule(0, 1, 2, 8) == [7, 0, 1]
    assert bench.fold_schedule(5, 1, 3, 8) == [4, 5, 6, 7]
    for s in range(3):                                                   # every step: the 8 ranks hold 8 different folds
        assert sorted(bench.fold_schedule(r, 1, 3, 8)[1 + s] for r in range(8)) == list(range(8))
    totals = {tuple(sorted(bench.fold_schedule(r, 0, 8, 8))) for r in range(8)}
    assert totals == {tuple(range(8))}                                   # over 8 steps every rank has fitted every fold once


def test_run_plan_and_warm_start_table(tmp_path):
    """user/run.py expands the None options once into an ordered plan (reference behaviour: user/run.py:69-93, 137-147): independent
    before covariant, isotropic before anisotropic, every later model warm-started; names <name>.<c|v>.<i|a>."""
    from romcomma_amd.user import run
    names = lambda *options: [(v.model_name('gpr'), v.start) for v in run._plan(*options)]
    assert names(False, False, False) == [('gpr.v.a', False)]
    assert names(True, True, True) == [('gpr.c.i', True)]
    assert names(False, False, None) == [('gpr.v.i', False), ('gpr.v.a', None)]
    assert names(None, None, False) == [('gpr.v.a', None), ('gpr.c.a', None)]
    assert names(False, None, None) == [('gpr.v.i', False), ('gpr.v.a', None), ('gpr.c.a', None)]       # covariant: anisotropic only
    assert names(True, None, True) == [('gpr.v.i', True), ('gpr.c.i', None)]
    # warm start on a fold folder: own folder > independent model of the same isotropy (covariant only) > isotropic model > fresh
    fold = type('F', (), {'folder': tmp_path})()
    store = lambda model: (tmp_path / model / 'kernel').mkdir(parents=True) or (tmp_path / model / 'kernel' / 'variance.csv').write_text('x')
    cov_a, ind_a, ind_i = run._Variant(True, False, None), run._Variant(False, False, None), run._Variant(False, True, None)
    assert cov_a.relatives('gpr') == ['gpr.v.a', 'gpr.c.i'] and ind_a.relatives('gpr') == ['gpr.v.i']
    assert run._resolve_warm_start(fold, 'gpr', ind_i) is False                       # nothing stored: its only relative is itself
    assert run._resolve_warm_start(fold, 'gpr', cov_a) is False and not (tmp_path / 'gpr.c.a').exists()
    store('gpr.v.i')
    assert run._resolve_warm_start(fold, 'gpr', ind_i) is True
    assert run._resolve_warm_start(fold, 'gpr', ind_a) is True and (tmp_path / 'gpr.v.a' / 'kernel' / 'variance.csv').exists()   # copied from .v.i
    (tmp_path / 'gpr.v.a' / 'kernel' / 'variance.csv').write_text('anisotropic')
    assert run._resolve_warm_start(fold, 'gpr', cov_a) is True
    assert (tmp_path / 'gpr.c.a' / 'kernel' / 'variance.csv').read_text() == 'anisotropic'                # .v.a preferred to .c.i
    (tmp_path / 'gpr.c.a' / 'kernel' / 'variance.csv').write_text('own')
    assert run._resolve_warm_start(fold, 'gpr', cov_a) is True and (tmp_path / 'gpr.c.a' / 'kernel' / 'variance.csv').read_text() == 'own'


def test_fold_deal_is_balanced_and_seeded():
    from romcomma_amd.data.storage import _deal
    label = _deal(61, 4, np.random.default_rng(3))
    assert sorted(np.bincount(label)) == [15, 15, 15, 16] and np.bincount(label)[0] == 16          # fold k < N mod K is the larger kind
    assert all(sorted(label[i:i + 4]) == [0, 1, 2, 3] for i in range(0, 60, 4))                    # every hand of K deals each fold once
    assert np.array_equal(label, _deal(61, 4, np.random.default_rng(3))) and not np.array_equal(label, _deal(61, 4, np.random.default_rng(4)))
    assert list(_deal(5, 1, np.random.default_rng(0))) == [0] * 5


def test_meta_json_is_replaced_atomically(tmp_path):
    repo = make_repo(tmp_path / 'repo')
    repo.meta['K'] = 3
    repo.write_meta()
    assert json.loads((tmp_path / 'repo' / 'meta.json').read_text())['K'] == 3
    assert [p.name for p in (tmp_path / 'repo').iterdir() if p.name.endswith('.tmp')] == []
    assert (tmp_path / 'repo' / 'meta.json').read_text().startswith('{\n        "')                 # indent 8


def test_folds_opened_together_on_one_gpu():
    """run._folds_at_once: equal groups, the fewest that overfill the GPU's places by at most a quarter (profiles/r04_folds_rule.txt)."""
    from romcomma_amd.user.run import _folds_at_once
    assert _folds_at_once(2, 9, 16) == 2            # 18 units on 16 places: one group, two units wait for a place
    assert _folds_at_once(4, 6, 16) == 2            # 24 units: two groups of 12, not 18 + 6
    assert _folds_at_once(6, 5, 16) == 3            # 30 units: 15 + 15
    assert _folds_at_once(8, 3, 16) == 4            # 24 units: 12 + 12, not 15 + 9
    assert _folds_at_once(10, 2, 16) == 10          # 20 units on 16 places
    assert _folds_at_once(5, 1, 2) == 2             # big folds, two places: never more than a quarter over
    assert _folds_at_once(3, 9, 4) == 1             # one fold's outputs already overfill the places: fold after fold
    assert _folds_at_once(1, 3, 16) == 1 and _folds_at_once(7, 1, 1) == 1
