"""One point of the reference's own sweep (benchmark_script.py:33-51: K = -2 folds, M inputs, N samples, L function outputs, GPR then GSA of
all three kinds WITH standard errors) through the drop-in API, timed stage by stage.   python tools/reference_sweep_point.py N M L [errors 0/1] [K]
(RCGP_PKG_ROOT = a directory holding another copy of the package, for A/B runs of the host side.)"""
import contextlib
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import pandas as pd

import os
sys.path.insert(0, os.environ.get('RCGP_PKG_ROOT') or str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib                                      # noqa: E402
from romcomma_amd.data.storage import Repository                   # noqa: E402
from romcomma_amd.gsa.models import GSA                            # noqa: E402
from romcomma_amd.user import run                                   # noqa: E402

N, M, L = (int(a) for a in sys.argv[1:4])
errors = (sys.argv[4] != '0') if len(sys.argv) > 4 else True
K = int(sys.argv[5]) if len(sys.argv) > 5 else -2
rng = np.random.default_rng(1)
U = rng.random((N, M))
Y = np.stack([sum(np.sin(2 * np.pi * U[:, (m + l) % M]) / (m + 1) for m in range(M)) + 0.3 * U[:, l % M] * U[:, (l + 1) % M] for l in range(L)], axis=1)
Y += 0.05 * rng.standard_normal((N, L))
columns = pd.MultiIndex.from_tuples([('X', f'X.{m}') for m in range(M)] + [('Y', f'Y.{l}') for l in range(L)])
with tempfile.TemporaryDirectory() as root, contextlib.redirect_stdout(sys.stderr):
    repo = Repository.from_df(Path(root) / 'repo', pd.DataFrame(np.concatenate([U, Y], axis=1), columns=columns)).into_K_folds(K, seed=0)
    s0 = _lib.stat()
    t0 = time.perf_counter()
    run.gpr('gpr', repo, is_read=None, is_covariant=False, is_isotropic=False)
    t1 = time.perf_counter()
    s1 = _lib.stat()
    run.gsa('gpr', repo, is_covariant=False, is_isotropic=False, kinds=GSA.ALL_KINDS, is_error_calculated=errors)
    t2 = time.perf_counter()
    s2 = _lib.stat()
print(f'N={N} M={M} L={L} K={K} errors={errors}: gpr {t1 - t0:.3f} s ({s1["gradients"] - s0["gradients"]} evaluations, {s1["batched_calls"] - s0["batched_calls"]} batched calls), '
      f'gsa {t2 - t1:.3f} s ({s2["factorisations"] - s1["factorisations"]} factorisations)')
