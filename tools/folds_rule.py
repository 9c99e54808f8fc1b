"""Which folds `run.gpr` opens together on one GPU: the grouping rules of `run._folds_at_once` (environment RCGP_FOLDS_RULE = floor | ceil |
balanced) timed against each other in one process, three repeats after a warm-up pass -> profiles/r04_folds_rule.txt."""
import contextlib, os, sys, tempfile, time
from pathlib import Path
import numpy as np, pandas as pd
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib
from romcomma_amd.data.storage import Repository
from romcomma_amd.user import run
for N, M, L, K in ((800, 5, 5, -6), (1000, 7, 9, -2), (2000, 5, 6, -4), (600, 4, 3, -8), (1500, 5, 4, -5), (3000, 5, 2, -10), (1000, 5, 7, -3)):
    rng = np.random.default_rng(1)
    U = rng.random((N, M))
    Y = np.stack([sum(np.sin(2 * np.pi * U[:, (m + l) % M]) / (m + 1) for m in range(M)) + 0.3 * U[:, l % M] * U[:, (l + 1) % M] for l in range(L)], axis=1)
    Y += 0.05 * rng.standard_normal((N, L))
    columns = pd.MultiIndex.from_tuples([('X', f'X.{m}') for m in range(M)] + [('Y', f'Y.{l}') for l in range(L)])
    out = {}
    for rep in range(4):
        for rule in ('floor', 'balanced'):
            os.environ['RCGP_FOLDS_RULE'] = rule
            with tempfile.TemporaryDirectory() as root, contextlib.redirect_stdout(sys.stderr):
                repo = Repository.from_df(Path(root) / 'repo', pd.DataFrame(np.concatenate([U, Y], axis=1), columns=columns)).into_K_folds(K, seed=0)
                s0 = _lib.stat(); t0 = time.perf_counter()
                try:
                    run.gpr('gpr', repo, is_read=None, is_covariant=False, is_isotropic=False, is_tested=False)
                except Exception as e:
                    print('failed', N, M, L, K, type(e).__name__); break
                t1 = time.perf_counter(); s1 = _lib.stat()
            if rep:
                out.setdefault(rule, []).append((round(t1 - t0, 3), s1['batched_calls'] - s0['batched_calls']))
    print(N, M, L, K, out, flush=True)
