"""Two ranks sharing the one GPU of the test box (gloo rendezvous; on an 8-GPU node the same code runs one rank per GPU over
RCCL): run.gpr / run.gsa shard the folds of a Repository round-robin, rank 0 collects, and the result equals the
single-process run."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
from pathlib import Path
import numpy as np, pandas as pd
from romcomma_amd import dist
from romcomma_amd.data.storage import Repository
from romcomma_amd.user import run
os.environ['LOCAL_RANK'] = '0'                       # both ranks on the single GPU of this box
rank, world, _ = dist.init_process_group('gloo')
repo = Repository(Path(sys.argv[2]))
names = run.gpr('gpr', repo, is_read=False, is_covariant=False, is_isotropic=False)
gsa = run.gsa('gpr', repo, is_covariant=False, is_isotropic=False)
assert names == ['gpr.v.a'], names
dist.barrier()
import torch.distributed as td
td.destroy_process_group()
'''


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def make_repo(folder: Path, N=200, M=3, seed=0):
    from romcomma_amd.data.storage import Repository
    rng = np.random.default_rng(seed)
    U = rng.random((N, M))
    y = np.sin(2 * np.pi * U[:, 0]) + 0.7 * U[:, 1] ** 2 + 0.05 * U[:, 2] + 0.02 * rng.standard_normal(N)
    columns = pd.MultiIndex.from_tuples([('X', f'X.{m}') for m in range(M)] + [('Y', 'Y.0')])
    return Repository.from_df(folder, pd.DataFrame(np.concatenate([U, y[:, None]], axis=1), columns=columns))


def test_two_ranks_shard_folds(gpu, tmp_path):
    from romcomma_amd.user import run
    single = make_repo(tmp_path / 'single').into_K_folds(-4, seed=6)
    run.gpr('gpr', single, is_read=False, is_covariant=False, is_isotropic=False)
    run.gsa('gpr', single, is_covariant=False, is_isotropic=False)
    multi = make_repo(tmp_path / 'multi').into_K_folds(-4, seed=6)
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), str(ROOT), str(multi.folder)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(outs)
    # rank 0 handled folds 0 and 2 (fitted at once: units_per_gpu defaults to a group at this size), rank 1 folds 1 and 3: the timers show it
    assert 'folds [0, 2] gpr.v.a GPR' in outs[0] and 'fold.0 gpr.v.a GSA' in outs[0] and 'fold.2 gpr.v.a GSA' in outs[0] and 'fold.1 ' not in outs[0]
    assert 'folds [1, 3] gpr.v.a GPR' in outs[1] and 'fold.1 gpr.v.a GSA' in outs[1] and 'fold.3 gpr.v.a GSA' in outs[1]
    for rel in ('gpr.v.a/kernel/lengthscales.csv', 'gpr.v.a/likelihood/log_marginal.csv', 'gpr.v.a/gsa/closed/S.csv', 'gpr.v.a/test_summary.csv'):
        a = pd.read_csv(single.folder / rel)
        b = pd.read_csv(multi.folder / rel)
        assert list(a.columns) == list(b.columns) and a.shape == b.shape
        np.testing.assert_allclose(a.select_dtypes('number').values, b.select_dtypes('number').values, rtol=1e-5, atol=2e-6)
