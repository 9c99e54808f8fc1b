"""Multi-process path on CPU: world_size 2 over gloo (the GPU job uses RCCL with the same calls). Units shard round-robin,
there is no data-path collective, and the single gather returns every unit's row in unit order on every rank."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, tmp: str):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT))
    from romcomma_amd import dist
    r, w, _ = dist.init_process_group('gloo')
    assert (r, w) == (rank, world) and dist.is_distributed()
    n_units = 5                                           # e.g. 5 (fold, output) units on 2 ranks: 3 + 2
    mine = dist.shard_units(n_units, rank, world)
    assert mine == list(range(rank, n_units, world))
    rows = np.array([[10.0 * u + c for c in range(4)] for u in mine])       # the per-unit result row (indices, LML, ...)
    table = dist.all_gather_rows(rows, n_units, mine)
    expected = np.array([[10.0 * u + c for c in range(4)] for u in range(n_units)])
    assert np.array_equal(table, expected)
    assert dist.max_over_ranks(float(rank + 1)) == float(world)
    dist.barrier()
    # output sharding through the reference's Y_split: rank 0 writes Y.0..Y.2, each rank gets its share
    import pandas as pd
    from romcomma_amd.data.storage import Repository
    from romcomma_amd.user import run
    folder = Path(tmp) / 'repo'
    if rank == 0:
        rng = np.random.default_rng(0)
        columns = pd.MultiIndex.from_tuples([('X', 'X.0'), ('X', 'X.1'), ('Y', 'Y.0'), ('Y', 'Y.1'), ('Y', 'Y.2')])
        Repository.from_df(folder, pd.DataFrame(rng.random((20, 5)), columns=columns))
    dist.barrier()
    mine = run.Y_splits_sharded(Repository(folder))
    assert [r.folder.name for r in mine] == ([f'Y.{l}' for l in range(rank, 3, world)])
    assert all(r.L == 1 and r.M == 2 and r.N == 20 for r in mine)
    dist.barrier()
    np.save(Path(tmp) / f'ok{rank}.npy', table)
    import torch.distributed as td
    td.destroy_process_group()


def test_shard_and_gather_world_size_2(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / 'ok0.npy'), np.load(tmp_path / 'ok1.npy')
    assert np.array_equal(a, b) and a.shape == (5, 4)


def test_single_process_fallbacks():
    from romcomma_amd import dist
    assert dist.shard_units(8, 3, 8) == [3]
    assert dist.shard_units(3, 5, 8) == []
    table = dist.all_gather_rows(np.array([[1.0, 2.0]]), 1, [0])
    assert table.tolist() == [[1.0, 2.0]]
    assert dist.max_over_ranks(2.5) == 2.5


class _StubFrames:
    def __init__(self, value):
        self.np = np.atleast_2d(np.asarray(value, dtype=float))


class _StubGP:
    """The few attributes gpr.sharded.OutputShard reads from a single-output HipGP."""

    def __init__(self, l, N, M):
        rng = np.random.default_rng(100 + l)
        self.N, self.M, self.L, self._is_covariant = N, M, 1, False
        self.K_inv_Y = rng.normal(size=(1, 1, N))
        kernel_frames = type('F', (), {'lengthscales': _StubFrames(rng.uniform(0.5, 2.0, (1, M))), 'variance': _StubFrames([[1.0 + l]])})()
        self.kernel = type('K', (), {'data': type('D', (), {'frames': kernel_frames})()})()
        self.closed = False

    def close(self):
        self.closed = True


def _shard_worker(rank: int, world: int, port: int, tmp: str):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT))
    from romcomma_amd import dist
    from romcomma_amd.gpr.sharded import OutputShard
    dist.init_process_group('gloo')
    L, N, M = 5, 37, 3                                     # five outputs on two ranks: 3 + 2
    owned = dist.shard_units(L, rank, world)
    gps = {l: _StubGP(l, N, M) for l in owned}
    shard = OutputShard(gps, L, Path(tmp) / 'merged' / 'gpr.v.a')
    assert shard.owned_outputs == owned and (shard.N, shard.M, shard.L) == (N, M, L)
    assert shard.is_writer == (rank == 0) and (shard.folder == Path(tmp) / 'merged' / 'gpr.v.a') == (rank == 0)
    for l in range(L):                                     # every rank holds every output's (alpha, lengthscales, variance)
        ref = _StubGP(l, N, M)
        assert np.array_equal(shard.K_inv_Y[l], ref.K_inv_Y.reshape(N)) and np.array_equal(shard.Lambda[l], ref.kernel.data.frames.lengthscales.np[0])
        assert shard.F[l] == 1.0 + l
    block = np.zeros((L, L, 4))
    for l in owned:
        block[l] = 10.0 * l + np.arange(L)[:, None] + 0.1 * np.arange(4)[None, :]
    full = shard.gather_output_rows(block)
    for l in range(L):
        assert np.array_equal(full[l], 10.0 * l + np.arange(L)[:, None] + 0.1 * np.arange(4)[None, :])
    shard.close()
    assert all(gp.closed for gp in gps.values())
    # a failure on one rank reaches all of them instead of leaving the others in the next collective
    try:
        dist.agree_on_failure(ValueError('boom') if rank == 1 else None)
        raised = None
    except ValueError as exc:
        raised = ('own', str(exc))
    except RuntimeError as exc:
        raised = ('other', str(exc))
    assert raised is not None and raised[0] == ('own' if rank == 1 else 'other') and ('boom' in raised[1] or '[1]' in raised[1]), raised
    dist.agree_on_failure(None)                            # nobody failed: nothing happens
    np.save(Path(tmp) / f'shard{rank}.npy', full)
    import torch.distributed as td
    td.destroy_process_group()


def test_output_shard_exchange_and_failure_agreement_world_size_2(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_shard_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert np.array_equal(np.load(tmp_path / 'shard0.npy'), np.load(tmp_path / 'shard1.npy'))


def test_failure_agreement_without_a_process_group():
    from romcomma_amd import dist
    dist.agree_on_failure(None)
    with pytest.raises(KeyError):
        dist.agree_on_failure(KeyError('x'))


# --------------------------------------------------------------------------------------------------------------------
# Eight ranks over gloo on the CPU: the host logic of the one-output-per-GPU configuration (BASELINE configs[3]) and
# of bench.py's N > 1 path with the device calls answered by the oracle (tests/oracle_backend.py -- test infrastructure; the product has
# no CPU path). What runs for real: Y_splits_sharded, the per-rank folds and fits, OutputShard's two all-gathers, gsa_outputs' per-fold
# agreement, rank 0's csv stores and Collect; bench.py's fold schedule, gather table and value formula.
# --------------------------------------------------------------------------------------------------------------------

def _eight_output_repo(folder: Path, N=44, M=3, L=8, seed=0):
    import pandas as pd
    from romcomma_amd.data.storage import Repository
    rng = np.random.default_rng(seed)
    U = rng.random((N, M))
    Y = np.stack([np.sin(2 * np.pi * U[:, l % M]) + 0.4 * (l + 1) / L * U[:, (l + 1) % M] ** 2 + 0.2 * U[:, (l + 2) % M] for l in range(L)], axis=1)
    Y += 0.03 * rng.standard_normal((N, L))
    columns = pd.MultiIndex.from_tuples([('X', f'X.{m}') for m in range(M)] + [('Y', f'Y.{l}') for l in range(L)])
    return Repository.from_df(folder, pd.DataFrame(np.concatenate([U, Y], axis=1), columns=columns))


def _outputs_worker(rank: int, world: int, port: int, repo_folder: str, drop_model_of_rank: int):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / 'tests'))
    import oracle_backend
    oracle_backend.install()
    import shutil
    from romcomma_amd import dist
    from romcomma_amd.data.storage import Repository
    from romcomma_amd.user import run
    dist.init_process_group('gloo')
    repo = Repository(Path(repo_folder))
    mine = run.Y_splits_sharded(repo)
    assert [r.folder.name for r in mine] == [f'Y.{l}' for l in range(rank, repo.L, world)]
    for split in mine:
        split.into_K_folds(-2, seed=5)
        assert run.gpr('gpr', split, is_read=False, is_covariant=False, is_isotropic=False, shard_folds=False) == ['gpr.v.a']
        if rank == drop_model_of_rank:
            shutil.rmtree(split.fold_folder(1) / 'gpr.v.a')          # this rank's stored GP of fold 1 goes missing
    try:
        done = run.gsa_outputs('gpr', repo, is_isotropic=False)
        outcome = 'done:' + ','.join(str(n) for n in done)
    except FileNotFoundError as error:
        outcome = 'own:' + type(error).__name__
    except RuntimeError as error:
        outcome = 'other:' + str(error)
    Path(repo_folder, f'outcome.{rank}').write_text(outcome)
    import torch.distributed as td
    td.destroy_process_group()


def test_gsa_outputs_with_eight_ranks_reproduces_the_single_process_tables(tmp_path):
    import pandas as pd
    import torch.multiprocessing as mp
    sys.path.insert(0, str(ROOT / 'tests'))
    import oracle_backend
    from romcomma_amd import _lib
    from romcomma_amd.user import run
    keep = (_lib.RcGP, _lib.device_count, _lib.lml_grad_batch, _lib.factor_batch)
    oracle_backend.install()
    try:
        single = _eight_output_repo(tmp_path / 'single').into_K_folds(-2, seed=5)
        run.gpr('gpr', single, is_read=False, is_covariant=False, is_isotropic=False)
        run.gsa('gpr', single, is_covariant=False, is_isotropic=False)
    finally:
        _lib.RcGP, _lib.device_count, _lib.lml_grad_batch, _lib.factor_batch = keep
    multi = _eight_output_repo(tmp_path / 'multi').into_K_folds(-2, seed=5)
    mp.spawn(_outputs_worker, args=(8, _free_port(), str(multi.folder), -1), nprocs=8, join=True)
    outcomes = [(multi.folder / f'outcome.{r}').read_text() for r in range(8)]
    assert outcomes == ['done:gpr.v.a/gsa/first_order,gpr.v.a/gsa/closed,gpr.v.a/gsa/total'] * 8, outcomes
    for k in range(2):
        for kind in ('first_order', 'closed', 'total'):
            for name in ('S.csv', 'V.csv'):
                rel = f'fold.{k}/gpr.v.a/gsa/{kind}/{name}'
                a, b = pd.read_csv(single.folder / rel, index_col=[0, 1]), pd.read_csv(multi.folder / rel, index_col=[0, 1])
                assert a.shape == b.shape == (64, 4) and list(a.index) == list(b.index) and list(a.columns) == list(b.columns), rel
                np.testing.assert_allclose(a.values, b.values, rtol=2e-5, atol=3e-6, err_msg=rel)       # the same converged fits; the files carry 6 decimals
                cross = [i for i, (l0, l1) in enumerate(a.index) if l0 != l1]
                assert np.any(np.abs(b.values[cross]) > 1e-4), f'{rel}: the cross-output rows are empty'
    collected = pd.read_csv(multi.folder / 'gpr.v.a' / 'gsa' / 'closed' / 'S.csv')
    assert collected.shape[0] == 2 * 64 and sorted(collected['fold'].unique()) == [0, 1]


def test_gpr_over_folds_at_once_writes_what_fold_after_fold_writes(tmp_path):
    """``run.gpr(units_per_gpu=...)``: the reference walks the folds one after the other (user/run.py:60-61); with several folds per GPU
    their GPs are calibrated together (HipGP.calibrate_group: lockstep L-BFGS-B, one batched evaluation per round). Host logic only
    (device calls answered by the oracle): the isotropic -> anisotropic warm-start plan on five folds of two outputs, groups of 2 + 2 + 1
    folds (four units per call), must leave byte-identical parameter and test files, fold by fold and collected."""
    sys.path.insert(0, str(ROOT / 'tests'))
    import oracle_backend
    from romcomma_amd import _lib
    from romcomma_amd.user import run
    keep = (_lib.RcGP, _lib.device_count, _lib.lml_grad_batch, _lib.factor_batch)
    oracle_backend.install()
    calls = []
    batched = _lib.lml_grad_batch
    _lib.lml_grad_batch = lambda gps: (calls.append(len(gps)), batched(gps))[1]
    try:
        repos = {}
        for units in (1, 4):
            repos[units] = _eight_output_repo(tmp_path / f'units{units}', N=60, L=2).into_K_folds(-5, seed=3)
            names = run.gpr('gpr', repos[units], is_read=False, is_covariant=False, is_isotropic=None, units_per_gpu=units)
            assert names == ['gpr.v.i', 'gpr.v.a']
    finally:
        _lib.RcGP, _lib.device_count, _lib.lml_grad_batch, _lib.factor_batch = keep
    assert max(calls) == 4 and calls.count(4) > 20            # units = 1 makes no batched call; units = 4: two folds x two outputs per round,
                                                              # fewer as units converge and leave
    files = [f'{model}/{name}' for model in ('gpr.v.i', 'gpr.v.a') for name in
             ('kernel/lengthscales.csv', 'kernel/variance.csv', 'likelihood/variance.csv', 'likelihood/log_marginal.csv', 'test.csv', 'test_summary.csv')]
    for rel in [f'fold.{k}/{f}' for k in range(5) for f in files] + files:
        assert (repos[1].folder / rel).read_bytes() == (repos[4].folder / rel).read_bytes(), rel


def test_a_missing_model_on_one_rank_stops_all_eight_before_the_collectives(tmp_path):
    """a rank that cannot read its stored GP must not leave the others inside OutputShard's all-gather -- the ranks agree on
    the local reads of a fold before its first collective. Rank 3 loses its fold-1 model: every rank gets through fold 0 and stops at 1."""
    import torch.multiprocessing as mp
    multi = _eight_output_repo(tmp_path / 'multi', N=30).into_K_folds(-2, seed=5)
    mp.spawn(_outputs_worker, args=(8, _free_port(), str(multi.folder), 3), nprocs=8, join=True)
    outcomes = [(multi.folder / f'outcome.{r}').read_text() for r in range(8)]
    assert outcomes[3].startswith('own:FileNotFoundError'), outcomes
    assert all(out.startswith('other:') and '[3]' in out for r, out in enumerate(outcomes) if r != 3), outcomes
    assert (multi.folder / 'fold.0' / 'gpr.v.a' / 'gsa' / 'closed' / 'S.csv').exists()
    assert not (multi.folder / 'fold.1' / 'gpr.v.a' / 'gsa' / 'closed' / 'S.csv').exists()


def test_bench_eight_ranks_rehearsal_over_gloo():
    """The driver's N = 8 command line (torch.distributed.run, eight ranks) at a toy size: one JSON line from rank 0, whole-job value over
    all eight ranks, and in the last timed step the eight ranks hold the eight different folds of the split (fold (r + s) mod 8)."""
    import json
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1', '--master-port',
           str(_free_port()), str(ROOT / 'tests' / 'bench_cpu_rehearsal.py'), '--gpus', '8', '--steps', '2', '--warmup', '1', '--rows', '64', '--dims', '3',
           '--backend', 'gloo']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(ROOT), env={**os.environ, 'OMP_NUM_THREADS': '1'})
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith('{')]
    assert len(lines) == 1, out.stdout                       # rank 0 alone prints
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak' and 'cpu_baseline' not in d
    assert d['value'] == pytest.approx(8 * 64 * 2 / (d['ms_per_step'] * 2e-3), rel=1e-9)
    assert 'x8' in d['config']['parallelism'] and d['config']['gathered_rows_last_step'] == 8
    assert sorted(d['config']['units_last_step']) == list(range(8))                  # rank r: fold (r + 1) mod 8 in timed step 1
    assert d['config']['units_last_step'] == [(r + 1) % 8 for r in range(8)]
