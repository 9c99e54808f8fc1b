"""BASELINE configs[3] (C3): L = 8 independent-output GPs on ONE design, N = 8192, M = 10.

(1) One GPU, one process: the eight outputs in turn on one device handle through ``HipGP`` (``rcgp_set_y`` rotation, the
    reference's ``for gp in self._implementation`` loop, gpr/models.py:360-361) at fixed hyper-parameters -- every output's LML and
    gradient through size-independent properties + finite differences, the Sobol invariants, and the full (L, L) matrices V and S of
    ``ClosedSobol`` with the cross-output entries (gsa/calibrators.py:79).
(2) Outputs on different ranks (two ranks sharing the test box's GPU over gloo; one rank per GPU over RCCL on an 8-GPU node):
    ``Y_splits_sharded`` + ``gpr(shard_folds=False)`` + ``gsa_outputs`` -- the all-gather of (alpha, lengthscales, variance) and of
    the finished rows -- gives the S.csv / V.csv (and T.csv / W.csv) of the single-process run on the L-output repository.
Parity is against the oracle (unpinned) for a sub-sample of entries and through invariants at full size.
"""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

from oracle import gp_oracle as o

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _repo_from_arrays(folder: Path, X: np.ndarray, Y: np.ndarray):
    """An un-normalised repository whose fold holds exactly (X, Y): the design is already probit-normalised (synthetic_outputs)."""
    from romcomma_amd.data.storage import Repository
    M, L = X.shape[1], Y.shape[1]
    columns = pd.MultiIndex.from_tuples([('X', f'X.{m}') for m in range(M)] + [('Y', f'Y.{l}') for l in range(L)])
    return Repository.from_df(folder, pd.DataFrame(np.concatenate([X, Y], axis=1), columns=columns))


def test_c3_eight_outputs_on_one_design(gpu, tmp_path):
    from romcomma_amd.data.storage import Fold
    from romcomma_amd.gpr.kernels import RBF
    from romcomma_amd.gpr.models import MOGP
    from romcomma_amd.gsa.calibrators import ClosedSobol
    from romcomma_amd.user.sample import synthetic_outputs
    N, M, L = 8192, 10, 8
    X, Y = synthetic_outputs(N, M, L)
    repo = _repo_from_arrays(tmp_path / 'c3', X, Y).into_K_folds(1, is_normalization_applicable=False, seed=1)   # fold 1 = improper: all rows
    fold = Fold(repo, 1)
    assert (fold.N, fold.M, fold.L) == (N, M, L)
    rng = np.random.default_rng(3)
    ell = rng.uniform(0.6, 3.5, (L, M))                       # every output its own ARD lengthscales, variance and noise
    var = rng.uniform(0.8, 1.6, (1, L))
    noise = rng.uniform(1.0e-3, 4.0e-3, (1, L))
    params = RBF.Data(fold.folder / 'c3.v.a' / 'kernel', variance=var, lengthscales=ell)
    gp = MOGP('c3.v.a', fold, False, False, False, kernel_parameters=params, likelihood_variance=noise)
    order = np.argsort(fold.X.index.values)                   # the improper fold keeps the repository's row order
    Xf, Yf = gp.X, gp.Y
    np.testing.assert_allclose(Xf[order], X, rtol=1e-12, atol=1e-15)      # (through data.csv: pandas' float parser is good to an ulp)
    np.testing.assert_allclose(Yf[order], Y, rtol=1e-12, atol=1e-15)

    # ---- per output: LML / gradient by properties and finite differences (the eight share a pool of four handles: set_y rotation)
    lml = gp.log_marginal_likelihood()
    assert lml.shape == (L,) and np.all(np.isfinite(lml))
    idx = np.random.default_rng(0).choice(N, 128, replace=False)
    alpha = gp.K_inv_Y                                        # (L, 1, N)
    mean_f, sd_f = gp.predict(Xf[idx], y_instead_of_f=False)
    for l in range(L):                                        # K alpha = y - noise alpha at training points (check_K_inv_Y, gpr/models.py:446-463)
        np.testing.assert_allclose(mean_f[:, l], Yf[idx, l] - noise[0, l] * alpha[l, 0, idx], rtol=0, atol=2e-8 * np.max(np.abs(Yf[:, l])))
        assert np.all(sd_f[:, l] ** 2 <= noise[0, l] * 1.000001)
    assert np.all(gp.check_K_inv_Y(Xf[idx]) < 1e-8)
    assert gp.pool_size == 4 and gp.handle is gp._unit(0)     # N = 8192: four units, two outputs each (output l on unit l mod 4)
    for l in (0, 3, 7):
        h = gp._select(l)
        assert h is gp._unit(l % 4)
        value, grad = h.lml_grad()
        assert value == pytest.approx(lml[l], rel=1e-13)
        for p, step in ((2, 1e-5), (M, 1e-5), (M + 1, 1e-7)):
            def at(delta):
                e, v, n_ = ell[l].copy(), var[0, l], noise[0, l]
                if p < M:
                    e[p] += delta
                elif p == M:
                    v += delta
                else:
                    n_ += delta
                h.set_hyper(e, v, n_)
                return h.lml()
            fd = (at(step) - at(-step)) / (2 * step)
            assert grad[p] == pytest.approx(fd, rel=2e-4), (l, p)
        h.set_hyper(ell[l], var[0, l], noise[0, l])
    # a sub-sampled oracle check of the Gram / solve chain is in test_gpu_parity; here: the LML of one output against LAPACK
    ref_lml = o.lml(Xf, Yf[:, 5], ell[5], var[0, 5], noise[0, 5])
    assert lml[5] == pytest.approx(ref_lml, rel=1e-9)

    # ---- the full (L, L) Sobol matrices
    cal = ClosedSobol(gp)
    V0, S = cal.V[0], cal.S
    assert V0.shape == (L, L) and np.all(np.diag(V0) > 0)
    np.testing.assert_array_equal(V0, V0.T)                   # mirrored exactly
    np.testing.assert_allclose(np.diag(S), 1.0, rtol=0, atol=1e-14)
    assert np.all(np.abs(S) <= 1.0 + 1e-9)                    # a correlation between outputs
    slices = [(m, m + 1) for m in range(M)] + [(0, m + 1) for m in range(M)] + [(m + 1, M) for m in range(M)]
    all_V = cal.marginalize_all(slices)['V']                  # (L, L, 3M), served from the one device pass
    for l in range(L):
        first, closed, comp = all_V[l, l, :M], all_V[l, l, M:2 * M], all_V[l, l, 2 * M:]
        full = V0[l, l]
        assert closed[M - 1] == pytest.approx(full, rel=1e-12) and first[0] == pytest.approx(closed[0], rel=1e-12)
        assert np.all(np.diff(closed) >= -1e-9 * full) and np.all(first <= closed + 1e-9 * full)
        assert np.all(1.0 - comp / full >= first / full - 1e-7)
    # the cross entry from the other side: V_lj computed with output j's design weights and output l passed in equals V_jl
    h2 = gp._select(2)
    V_26 = h2.sobol_cross(ell[6], var[0, 6], alpha[6, 0], [(0, M), (0, 3), (4, 5)])
    h6 = gp._select(6)
    V_62 = h6.sobol_cross(ell[2], var[0, 2], alpha[2, 0], [(0, M), (0, 3), (4, 5)])
    np.testing.assert_allclose(V_26, V_62, rtol=1e-9, atol=1e-13 * abs(V0[2, 2]))
    assert V_26[0] == pytest.approx(V0[2, 6], rel=1e-12)
    # ... and against the oracle's pair form on the full-size design (O(N^2 M) NumPy: one slice, one pair)
    g, phi = o.sobol_prepare(Xf, alpha[[2, 6], 0, :], var[0, [2, 6]], ell[[2, 6]])
    ref = o.sobol_V_pair(Xf, g[0], g[1], phi[0], phi[1], [(0, 3)])
    assert V_26[1] == pytest.approx(ref[0], rel=1e-7)
    gp.close()


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
from pathlib import Path
from romcomma_amd import dist
from romcomma_amd.data.storage import Repository
from romcomma_amd.user import run
os.environ['LOCAL_RANK'] = '0'                       # both ranks on the single GPU of this box
rank, world, _ = dist.init_process_group('gloo')
repo = Repository(Path(sys.argv[2]))
mine = run.Y_splits_sharded(repo)
assert [r.folder.name for r in mine] == ([f'Y.{l}' for l in range(rank, repo.L, world)]), [str(r) for r in mine]
for split in mine:
    split.into_K_folds(-2, seed=5)
    names = run.gpr('gpr', split, is_read=False, is_covariant=False, is_isotropic=False, shard_folds=False)
    assert names == ['gpr.v.a'], names
gsa = run.gsa_outputs('gpr', repo, is_isotropic=False, is_error_calculated=(sys.argv[3] == '1'))
assert [str(n) for n in gsa] == ['gpr.v.a/gsa/first_order', 'gpr.v.a/gsa/closed', 'gpr.v.a/gsa/total'], gsa
dist.barrier()
import torch.distributed as td
td.destroy_process_group()
'''


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _small_repo(folder: Path, N=220, M=3, L=3, seed=0):
    rng = np.random.default_rng(seed)
    U = rng.random((N, M))
    f = np.stack([np.sin(2 * np.pi * U[:, 0]) + 0.7 * U[:, 1] ** 2, np.cos(2 * np.pi * U[:, 1]) + 0.5 * U[:, 2] + 0.3 * U[:, 0],
                  U[:, 0] * U[:, 2] + np.sin(3 * U[:, 1])], axis=1)[:, :L]
    Y = f + 0.03 * rng.standard_normal((N, L))
    return _repo_from_arrays(folder, U, Y)


@pytest.mark.parametrize('with_errors', [False, True])
def test_outputs_on_different_ranks_fill_the_cross_output_entries(gpu, tmp_path, with_errors):
    from romcomma_amd.user import run
    single = _small_repo(tmp_path / 'single').into_K_folds(-2, seed=5)
    run.gpr('gpr', single, is_read=False, is_covariant=False, is_isotropic=False)
    run.gsa('gpr', single, is_covariant=False, is_isotropic=False, is_error_calculated=with_errors)
    multi = _small_repo(tmp_path / 'multi').into_K_folds(-2, seed=5)
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), str(ROOT), str(multi.folder), '1' if with_errors else '0'], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(outs)
    files = ['S.csv', 'V.csv'] + (['T.csv', 'W.csv'] if with_errors else [])
    for k in range(2):
        for kind in ('first_order', 'closed', 'total'):
            for f in files:
                rel = f'fold.{k}/gpr.v.a/gsa/{kind}/{f}'
                a = pd.read_csv(single.folder / rel, index_col=[0, 1])
                b = pd.read_csv(multi.folder / rel, index_col=[0, 1])
                width = 4 if f in ('S.csv', 'V.csv') else 3          # T, W: is_T_partial (the default) appends no full-model column
                assert list(a.columns) == list(b.columns) and a.shape == b.shape == (9, width) and list(a.index) == list(b.index), rel
                # the same fits from the same start on the same data: equal up to the 6 decimals the files carry, cross entries included
                np.testing.assert_allclose(a.values, b.values, rtol=2e-5, atol=3e-6, err_msg=rel)
                off = [i for i, (l0, l1) in enumerate(a.index) if l0 != l1]
                assert np.any(np.abs(b.values[off]) > 1e-4), f'{rel}: cross-output rows are empty'
    collected = pd.read_csv(multi.folder / 'gpr.v.a' / 'gsa' / 'closed' / 'S.csv')
    assert collected.shape[0] == 2 * 9 and list(collected.columns[:2]) == ['N', 'fold']
