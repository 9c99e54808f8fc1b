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
